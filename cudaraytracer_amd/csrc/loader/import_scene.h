// Owning C++ form of the import-level scene (acr_import_scene in
// include/acr_host.h) and the loaders that produce it.
#ifndef ACR_LOADER_IMPORT_SCENE_H
#define ACR_LOADER_IMPORT_SCENE_H

#include <string>
#include <vector>

#include "../../../include/acr_host.h"

namespace acr
{
	struct ImportScene
	{
		std::vector<acr_import_node> nodes;
		std::vector<uint32_t> nodeMeshIds;
		std::vector<acr_import_mesh> meshes;
		std::vector<float> positions, normals, colors; // colors empty = none
		std::vector<uint32_t> indices;
		std::vector<acr_import_material> materials;
		std::vector<acr_import_light> lights;
		acr_import_camera camera{};

		acr_import_scene view() const
		{
			acr_import_scene v{};
			v.num_nodes = uint32_t(nodes.size()); v.nodes = nodes.data();
			v.num_node_mesh_ids = uint32_t(nodeMeshIds.size()); v.node_mesh_ids = nodeMeshIds.data();
			v.num_meshes = uint32_t(meshes.size()); v.meshes = meshes.data();
			v.num_vertices = uint32_t(positions.size() / 3); v.positions = positions.data(); v.normals = normals.data();
			v.colors = colors.empty() ? nullptr : colors.data();
			v.num_faces = uint32_t(indices.size() / 3); v.indices = indices.data();
			v.num_materials = uint32_t(materials.size()); v.materials = materials.data();
			v.num_lights = uint32_t(lights.size()); v.lights = lights.data();
			v.camera = camera;
			return v;
		}
	};

	// Blender .blend -> ImportScene (loader/blend_import.cpp)
	bool loadBlend(const char *path, ImportScene &out, std::string &error);
	// procedural triangle soup (loader/procedural.cpp)
	void makeSoup(uint32_t numTriangles, uint32_t seed, float edge, float extent, ImportScene &out);
} // namespace acr

#endif
