#include "mesh.h"

#include <cstring>

namespace acr
{
	// What the reference's Mesh(const aiMesh*) produces (geometry.cu:8-48): the vertex array, centroid = sum of the
	// positions / count, the AABB of the positions, and the BIH over the faces with that AABB as its root box.
	// The import scene keeps all meshes' vertices and index triples in shared pools; a mesh is a slice of each.
	Mesh::Mesh(const acr_import_scene &scene, const acr_import_mesh &mesh, size_t bihDepth)
		: materialIndex(mesh.material_index), firstPrimitive(mesh.face_begin)
	{
		const size_t first = 3 * size_t(mesh.vertex_begin);
		const float *position = scene.positions + first;
		const float *normal = scene.normals + first;
		const float *color = scene.colors ? scene.colors + first : nullptr;

		std::vector<Vertex> pool(mesh.vertex_count);
		PointFold fold;
		for (Vertex &v : pool) {
			v.position = math::vec3(position[0], position[1], position[2]);
			v.normal = math::vec3(normal[0], normal[1], normal[2]);
			v.color = color ? Color3(color[0], color[1], color[2]) : Color3(1.0f, 1.0f, 1.0f);
			position += 3;
			normal += 3;
			if (color) color += 3;
			fold.runningFirst(v.position);
		}
		centroid = fold.mean();
		boundingBox = fold.box;

		// a Face is its three vertex numbers, which is what the index pool holds
		static_assert(sizeof(Face) == 3 * sizeof(uint32_t), "Face is an index triple");
		std::vector<Face> triangles(mesh.face_count);
		if (mesh.face_count) std::memcpy(triangles.data(), scene.indices + 3 * size_t(mesh.face_begin), sizeof(Face) * mesh.face_count);

		vertices = vector<Vertex>(pool);
		faces = BIH<Face>(triangles, boundingBox, pool.data(), bihDepth);
	}
} // namespace acr
