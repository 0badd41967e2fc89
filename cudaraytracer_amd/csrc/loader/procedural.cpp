// Procedural scenes.  makeSoup builds the deep-BIH stress workload: a cloud
// of small independent triangles (centres uniform in [-extent, extent]^3, edge
// vectors uniform in [-edge, edge]^3), one mesh, one object, camera on the z
// axis at 4 * extent, point light at 3 * extent.  The reference's triangle test
// rejects |det| < 0.001 in absolute terms (math.h:140), so a soup only renders
// when edge^2 is well above that: scale the cloud up rather than the triangles
// down.  Every value comes
// from a counter-based hash of (seed, triangle, component), so the scene is
// reproducible anywhere and can be generated in any order.
#include <cmath>
#include <cstring>

#include "../math/math.h"
#include "import_scene.h"

namespace acr
{
	namespace
	{
		// splitmix64 finaliser as a stateless hash
		uint64_t mix(uint64_t x)
		{
			x += 0x9E3779B97F4A7C15ull;
			x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
			x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
			return x ^ (x >> 31);
		}
		// uniform in [-1, 1): 24 random bits
		float unit(uint32_t seed, uint64_t tri, uint32_t comp)
		{
			const uint64_t h = mix(mix(uint64_t(seed) * 0x100000001B3ull + tri) + comp);
			return float(uint32_t(h >> 40)) * (1.0f / 8388608.0f) - 1.0f;
		}
		void setName(char (&dst)[64], const char *s)
		{
			std::memset(dst, 0, 64);
			std::strncpy(dst, s, 63);
		}
	} // namespace

	void makeSoup(uint32_t numTriangles, uint32_t seed, float edge, float extent, ImportScene &out)
	{
		out = ImportScene();
		out.positions.resize(9 * size_t(numTriangles));
		out.normals.resize(9 * size_t(numTriangles));
		out.indices.resize(3 * size_t(numTriangles));
		for (uint64_t t = 0; t < numTriangles; t++) {
			const math::vec3 c = math::vec3(unit(seed, t, 0), unit(seed, t, 1), unit(seed, t, 2)) * extent;
			const math::vec3 e1 = math::vec3(unit(seed, t, 3), unit(seed, t, 4), unit(seed, t, 5)) * edge;
			const math::vec3 e2 = math::vec3(unit(seed, t, 6), unit(seed, t, 7), unit(seed, t, 8)) * edge;
			const math::vec3 p[3] = {c, c + e1, c + e2};
			math::vec3 n = math::cross(e1, e2);
			const float len = math::length(n);
			n = len > 0.0f ? n * (1.0f / len) : math::vec3(0, 0, 1);
			for (int k = 0; k < 3; k++) {
				float *pp = &out.positions[9 * t + 3 * k], *nn = &out.normals[9 * t + 3 * k];
				pp[0] = p[k].x; pp[1] = p[k].y; pp[2] = p[k].z;
				nn[0] = n.x; nn[1] = n.y; nn[2] = n.z;
				out.indices[3 * t + k] = uint32_t(3 * t + k);
			}
		}
		acr_import_mesh m{};
		m.vertex_begin = 0; m.vertex_count = 3 * numTriangles;
		m.face_begin = 0; m.face_count = numTriangles;
		m.material_index = 0;
		out.meshes.push_back(m);

		acr_import_material mat{};
		for (int k = 0; k < 3; k++) { mat.diffuse[k] = 0.8f; mat.specular[k] = 0.0f; mat.ambient[k] = 0.0f; }
		out.materials.push_back(mat);

		const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
		acr_import_node root{};
		root.parent = -1;
		std::memcpy(root.transform, ident, sizeof(ident));
		setName(root.name, "<Root>");
		out.nodes.push_back(root);

		acr_import_node soup{};
		soup.parent = 0;
		std::memcpy(soup.transform, ident, sizeof(ident));
		soup.mesh_begin = 0; soup.mesh_count = 1;
		setName(soup.name, "Soup");
		out.nodes.push_back(soup);
		out.nodeMeshIds.push_back(0);

		acr_import_node lamp{};
		lamp.parent = 0;
		std::memcpy(lamp.transform, ident, sizeof(ident));
		lamp.transform[14] = 3.0f * extent; // translation z
		setName(lamp.name, "Lamp");
		out.nodes.push_back(lamp);
		acr_import_light l{};
		setName(l.name, "Lamp");
		l.type = ACR_LIGHT_POINT;
		for (int k = 0; k < 3; k++) l.ambient[k] = l.diffuse[k] = l.specular[k] = 1.0f;
		l.att_constant = 0.0f; l.att_linear = 1.0f; l.att_quadratic = 0.0f;
		l.inner_cone = l.outer_cone = 6.28318530717958647692f;
		out.lights.push_back(l);

		acr_import_node cam{};
		cam.parent = 0;
		std::memcpy(cam.transform, ident, sizeof(ident));
		cam.transform[14] = 4.0f * extent;
		setName(cam.name, "Camera");
		out.nodes.push_back(cam);
		setName(out.camera.name, "Camera");
		out.camera.aspect = 0.0f;
		out.camera.hfov = 0.25f * 3.1415926535897932384626433832795f;
		out.camera.up[1] = 1.0f;
		out.camera.look_at[2] = -1.0f;
	}
} // namespace acr
