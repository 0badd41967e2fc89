// acr::Mesh (reference src/geometry/mesh.h:9-35)
#ifndef ACR_GEOMETRY_MESH_H
#define ACR_GEOMETRY_MESH_H

#include "../utils/bih.h"
#include "../utils/vector.h"
#include "geometry.h"

namespace acr
{
	class Mesh
	{
	public:
		math::vec3 centroid;
		BoundingBox boundingBox;

		Mesh() : materialIndex(0) {}
		// the reference's Mesh(const aiMesh*) (geometry.cu:8-48)
		Mesh(const acr_import_scene &scene, const acr_import_mesh &mesh, size_t bihDepth = 6);

		const vector<Vertex> &getVertices() const { return vertices; }
		const BIH<Face> &getFaces() const { return faces; }
		uint32_t getMaterialIndex() const { return materialIndex; }
		uint32_t getFirstPrimitive() const { return firstPrimitive; }

	private:
		vector<Vertex> vertices;
		BIH<Face> faces;
		uint32_t materialIndex;
		uint32_t firstPrimitive = 0; // number of this mesh's face 0 in the import scene's face list
	};
} // namespace acr

#endif
