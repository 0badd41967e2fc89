// Host-side scene construction (reference src/scene/scene.cu:28-202,294-405).
// Every float operation is in the reference's order: these values feed the
// BIH build and the per-object transforms the kernels use, and both must
// come out bit-identical.
#include "scene.h"

#include <cstring>
#include <stdexcept>

#include "../loader/import_scene.h"

namespace acr
{
	namespace
	{
		math::mat4 toMat4(const float *m)
		{
			math::mat4 r;
			for (int c = 0; c < 4; c++)
				r.c[c] = math::vec4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
			return r;
		}
		bool sameName(const char *a, const char *b) { return std::strncmp(a, b, 64) == 0; }
	} // namespace

	Scene::Scene(const Scene::Args &args) : rootIndex(-1)
	{
		std::string error;
		ImportScene imported;
		if (!loadBlend(args.filePath, imported, error))
			throw std::runtime_error("scene import error: " + error); // scene.cu:38-42 prints and exits
		const acr_import_scene v = imported.view();
		loadScene(v, args.sceneBihDepth, args.meshBihDepth);
	}

	Scene::Scene(const acr_import_scene &scene, size_t sceneBihDepth, size_t meshBihDepth) : rootIndex(-1)
	{
		loadScene(scene, sceneBihDepth, meshBihDepth);
	}

	Scene::~Scene() { delete arrays; }

	// scene.cu:54-86
	void Scene::loadScene(const acr_import_scene &scene, size_t sceneBihDepth, size_t meshBihDepth)
	{
		std::string camName;
		loadCamera(scene, camName);

		std::vector<Light> hLights;
		loadLights(scene, hLights);
		loadMaterials(scene);

		std::vector<Mesh> hMeshes;
		loadMeshes(scene, hMeshes, meshBihDepth);

		loadObjects(scene, camName, hLights, hMeshes, sceneBihDepth);

		lights = vector<Light>(hLights);
		meshes = vector<Mesh>(hMeshes);
	}

	void Scene::loadMeshes(const acr_import_scene &scene, std::vector<Mesh> &hMeshes, size_t meshBihDepth)
	{
		hMeshes.clear();
		hMeshes.reserve(scene.num_meshes);
		for (uint32_t i = 0; i < scene.num_meshes; i++)
			hMeshes.emplace_back(scene, scene.meshes[i], meshBihDepth);
	}

	void Scene::loadMaterials(const acr_import_scene &scene)
	{
		std::vector<Material> mats;
		for (uint32_t i = 0; i < scene.num_materials; i++)
			mats.emplace_back(scene.materials[i]);
		materials = vector<Material>(mats);
	}

	void Scene::loadCamera(const acr_import_scene &scene, std::string &camName)
	{
		camera = Camera(scene.camera);
		camName = std::string(scene.camera.name, strnlen(scene.camera.name, 64));
	}

	void Scene::loadLights(const acr_import_scene &scene, std::vector<Light> &hLights)
	{
		hLights.clear();
		for (uint32_t i = 0; i < scene.num_lights; i++)
			hLights.emplace_back(scene.lights[i]);
	}

	// scene.cu:122-152
	void Scene::loadObjects(const acr_import_scene &scene, const std::string &camName, std::vector<Light> &hLights, std::vector<Mesh> &hMeshes,
	                        size_t sceneBihDepth)
	{
		std::vector<std::vector<uint32_t>> childLists(scene.num_nodes);
		for (uint32_t i = 0; i < scene.num_nodes; i++)
			if (scene.nodes[i].parent >= 0) childLists[size_t(scene.nodes[i].parent)].push_back(i);

		std::vector<Object> objs;
		rootIndex = scene.num_nodes ? loadObject(scene, 0, nullptr, objs, camName, hLights, hMeshes, childLists) : -1;

		boundingBox.max = math::vec3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
		boundingBox.min = math::vec3(FLT_MAX, FLT_MAX, FLT_MAX);

		// only nodes that carry meshes take part in the BIH (scene.cu:131-143)
		std::vector<Object> filteredObjs;
		for (const Object &o : objs)
			if (o.meshes.size() > 0) filteredObjs.push_back(o);

		for (size_t i = 0; i < filteredObjs.size(); i++) {
			boundingBox.max = math::max(boundingBox.max, filteredObjs[i].boundingBox.max);
			boundingBox.min = math::min(boundingBox.min, filteredObjs[i].boundingBox.min);
		}

		objects = BIH<Object>(filteredObjs, boundingBox, nullptr, sceneBihDepth);
	}

	// scene.cu:155-202
	int Scene::loadObject(const acr_import_scene &scene, uint32_t nodeIdx, const Object *parent, std::vector<Object> &objs, const std::string &camName,
	                      std::vector<Light> &hLights, std::vector<Mesh> &hMeshes, const std::vector<std::vector<uint32_t>> &childLists)
	{
		const acr_import_node &node = scene.nodes[nodeIdx];
		Object tmp(scene, node, int(objs.size()), parent, hMeshes);
		objs.push_back(tmp);
		const int i = tmp.index;

		// lights and the camera are tied to their nodes by name (scene.cu:174-191);
		// of several lights with one name the first is the one the map keeps (scene.cu:117)
		for (uint32_t li = 0; li < scene.num_lights; li++) {
			if (sameName(scene.lights[li].name, node.name)) {
				Light &l = hLights[li];
				l.position = math::translate(objs[size_t(i)].globalTransform, l.position);
				l.direction = math::translaten(tmp.globalNormalTransform, l.direction);
				break;
			}
		}
		if (std::string(node.name, strnlen(node.name, 64)) == camName) {
			camera.position = math::translate(tmp.globalTransform, camera.position);
			camera.forward = math::translaten(tmp.globalNormalTransform, camera.forward);
			camera.up = math::translaten(tmp.globalNormalTransform, camera.up);
		}

		std::vector<int> children;
		for (uint32_t c : childLists[nodeIdx])
			children.push_back(loadObject(scene, c, &tmp, objs, camName, hLights, hMeshes, childLists));
		objs[size_t(i)].children = vector<int>(children);
		return i;
	}

	// scene.cu:303-330
	Light::Light(const acr_import_light &l)
		: attConstant(l.att_constant), attLinear(l.att_linear), attQuadratic(l.att_quadratic), innerConeAngle(l.inner_cone),
		  outerConeAngle(l.outer_cone)
	{
		position = math::vec3(l.position[0], l.position[1], l.position[2]);
		direction = math::vec3(l.direction[0], l.direction[1], l.direction[2]);
		ambient = Color3(l.ambient[0], l.ambient[1], l.ambient[2]);
		diffuse = Color3(l.diffuse[0], l.diffuse[1], l.diffuse[2]);
		specular = Color3(l.specular[0], l.specular[1], l.specular[2]);
		switch (l.type) {
		case ACR_LIGHT_DIRECTIONAL: type = DIRECTIONAL; break;
		case ACR_LIGHT_SPOT: type = SPOT; break;
		default: type = POINT; break;
		}
	}

	// scene.cu:332-357: the eight corners go through the global transform one by one
	BoundingBox Object::transformBoundingBox(BoundingBox bb) const
	{
		BoundingBox bounds;
		bounds.max = math::vec3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
		bounds.min = math::vec3(FLT_MAX, FLT_MAX, FLT_MAX);
		const float coords[6] = {bb.min.x, bb.min.y, bb.min.z, bb.max.x, bb.max.y, bb.max.z};
		for (int i = 0; i < 8; i++) {
			const int xi = 3 * (i >> 2);
			const int yi = 3 * ((i >> 1) & 1) + 1;
			const int zi = 3 * (i & 1) + 2;
			const math::vec3 localVertex(coords[xi], coords[yi], coords[zi]);
			const math::vec3 globalVertex = math::translate(globalTransform, localVertex);
			bounds.min = math::min(globalVertex, bounds.min);
			bounds.max = math::max(globalVertex, bounds.max);
		}
		return bounds;
	}

	// scene.cu:359-405
	Object::Object(const acr_import_scene &scene, const acr_import_node &node, int idx, const Object *parent, const std::vector<Mesh> &hMeshes)
		: index(idx), parentIndex(parent ? parent->index : -1)
	{
		localTransform = toMat4(node.transform);
		globalTransform = parent ? parent->globalTransform * localTransform : localTransform;
		globalInverseTransform = math::inverse(globalTransform);
		globalNormalTransform = math::transpose(math::inverse(math::toMat3(globalTransform)));
		globalInverseNormalTransform = math::inverse(globalNormalTransform);

		std::vector<int> objMeshes;
		for (uint32_t k = 0; k < node.mesh_count; k++)
			objMeshes.push_back(int(scene.node_mesh_ids[node.mesh_begin + k]));
		centroid = math::vec3(0, 0, 0);
		boundingBox.min = math::vec3(0, 0, 0);
		boundingBox.max = math::vec3(0, 0, 0);

		if (objMeshes.size() > 0) {
			math::vec3 minBound(FLT_MAX, FLT_MAX, FLT_MAX);
			math::vec3 maxBound(-FLT_MAX, -FLT_MAX, -FLT_MAX);
			math::vec3 sumCentroids(0, 0, 0);
			for (size_t i = 0; i < objMeshes.size(); i++) {
				const Mesh &m = hMeshes[size_t(objMeshes[i])];
				sumCentroids += m.centroid;
				minBound = math::min(minBound, m.boundingBox.min);
				maxBound = math::max(maxBound, m.boundingBox.max);
			}
			BoundingBox localBoundingBox;
			localBoundingBox.min = minBound;
			localBoundingBox.max = maxBound;
			boundingBox = transformBoundingBox(localBoundingBox);

			const math::vec3 avgCentroid = sumCentroids / float(objMeshes.size());
			centroid = math::translate(globalTransform, avgCentroid);
		}
		meshes = vector<int>(objMeshes);
	}
} // namespace acr
