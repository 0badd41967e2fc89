#include "mesh.h"

namespace acr
{
	// geometry.cu:8-48: copy vertices, sum them for the centroid, grow the AABB
	// with glm-order min/max, then build the per-mesh BIH over the faces.
	Mesh::Mesh(const acr_import_scene &scene, const acr_import_mesh &mesh, size_t bihDepth)
		: materialIndex(mesh.material_index), firstPrimitive(mesh.face_begin)
	{
		std::vector<Vertex> vs(mesh.vertex_count);

		math::vec3 sumVertices(0, 0, 0);
		math::vec3 minBound(FLT_MAX, FLT_MAX, FLT_MAX);
		math::vec3 maxBound(-FLT_MAX, -FLT_MAX, -FLT_MAX);

		for (uint32_t i = 0; i < mesh.vertex_count; i++) {
			const size_t at = 3 * size_t(mesh.vertex_begin + i);
			vs[i].position = math::vec3(scene.positions[at], scene.positions[at + 1], scene.positions[at + 2]);
			vs[i].normal = math::vec3(scene.normals[at], scene.normals[at + 1], scene.normals[at + 2]);
			vs[i].color = scene.colors ? Color3(scene.colors[at], scene.colors[at + 1], scene.colors[at + 2]) : Color3(1.0f, 1.0f, 1.0f);
			sumVertices += vs[i].position;
			minBound = math::min(minBound, vs[i].position);
			maxBound = math::max(maxBound, vs[i].position);
		}

		boundingBox.min = minBound;
		boundingBox.max = maxBound;
		centroid = sumVertices / float(mesh.vertex_count);

		vertices = vector<Vertex>(vs);

		std::vector<Face> f(mesh.face_count);
		for (uint32_t i = 0; i < mesh.face_count; i++)
			for (uint32_t j = 0; j < 3; j++)
				f[i].indices[j] = scene.indices[3 * size_t(mesh.face_begin + i) + j];
		faces = BIH<Face>(f, boundingBox, vs.data(), bihDepth);
	}
} // namespace acr
