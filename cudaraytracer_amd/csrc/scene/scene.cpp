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

	// The node hierarchy becomes the object list (scene.cu:122-202): one Object per node, numbered in depth-first
	// pre-order, children in the order the import lists them; lights and the camera are moved to world space by the
	// node that carries their name; the objects with meshes go into the BIH.
	void Scene::loadObjects(const acr_import_scene &scene, const std::string &camName, std::vector<Light> &hLights, std::vector<Mesh> &hMeshes,
	                        size_t sceneBihDepth)
	{
		// import nodes refer to their parent; the walk needs each node's children
		std::vector<std::vector<uint32_t>> childNodes(scene.num_nodes);
		for (uint32_t n = 0; n < scene.num_nodes; n++)
			if (scene.nodes[n].parent >= 0) childNodes[size_t(scene.nodes[n].parent)].push_back(n);

		std::vector<Object> objs;
		std::vector<uint32_t> nodeOf;               // import node of each object
		std::vector<std::vector<int>> childrenOf;   // object numbers of each object's children
		struct Pending { uint32_t node; int parent; };
		std::vector<Pending> todo;
		if (scene.num_nodes) todo.push_back({0u, -1});
		while (!todo.empty()) {
			const Pending at = todo.back();
			todo.pop_back();
			const int number = int(objs.size());
			math::mat4 parentGlobal; // a copy: the list may move while it grows
			if (at.parent >= 0) parentGlobal = objs[size_t(at.parent)].globalTransform;
			objs.emplace_back(scene, scene.nodes[at.node], number, at.parent, at.parent >= 0 ? &parentGlobal : nullptr, hMeshes);
			nodeOf.push_back(at.node);
			childrenOf.emplace_back();
			if (at.parent >= 0) childrenOf[size_t(at.parent)].push_back(number);
			// last child first onto the stack: the first child is the next node visited
			for (size_t k = childNodes[at.node].size(); k-- > 0;)
				todo.push_back({childNodes[at.node][k], number});
		}
		for (size_t i = 0; i < objs.size(); i++)
			objs[i].children = vector<int>(childrenOf[i]);
		rootIndex = objs.empty() ? -1 : 0;

		// Lights and the camera are tied to nodes by name (scene.cu:174-191).  Of several lights with one name the first
		// is the one the reference's name map keeps (scene.cu:117); nodes are taken in the order they were visited.
		for (size_t i = 0; i < objs.size(); i++) {
			const acr_import_node &node = scene.nodes[nodeOf[i]];
			const Object &o = objs[i];
			for (uint32_t li = 0; li < scene.num_lights; li++) {
				if (!sameName(scene.lights[li].name, node.name)) continue;
				hLights[li].position = math::translate(o.globalTransform, hLights[li].position);
				hLights[li].direction = math::translaten(o.globalNormalTransform, hLights[li].direction);
				break;
			}
			if (std::string(node.name, strnlen(node.name, 64)) == camName) {
				camera.position = math::translate(o.globalTransform, camera.position);
				camera.forward = math::translaten(o.globalNormalTransform, camera.forward);
				camera.up = math::translaten(o.globalNormalTransform, camera.up);
			}
		}

		// only nodes that carry meshes take part in the BIH (scene.cu:131-143); its root box is the union of
		// their world boxes (scene.cu:145-149: max first, then min, running value as first operand)
		std::vector<Object> withMeshes;
		PointFold world;
		for (const Object &o : objs) {
			if (o.meshes.size() == 0) continue;
			withMeshes.push_back(o);
			world.enclose(o.boundingBox);
		}
		boundingBox = world.box;
		objects = BIH<Object>(withMeshes, boundingBox, nullptr, sceneBihDepth);
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

	// World-space AABB of a local-space box (scene.cu:332-357): its eight corners go through the global transform
	// one by one -- x from bit 2 of the corner number, y from bit 1, z from bit 0 -- and are folded with the new
	// corner as first operand of min / max.
	BoundingBox Object::transformBoundingBox(BoundingBox bb) const
	{
		const math::vec3 *end[2] = {&bb.min, &bb.max};
		PointFold fold;
		for (unsigned corner = 0; corner < 8; corner++) {
			const math::vec3 local(end[(corner >> 2) & 1u]->x, end[(corner >> 1) & 1u]->y, end[corner & 1u]->z);
			fold.runningSecond(math::translate(globalTransform, local));
		}
		return fold.box;
	}

	// scene.cu:359-405.  parentGlobal: the parent object's global transform, null for the root.
	Object::Object(const acr_import_scene &scene, const acr_import_node &node, int idx, int parentIdx, const math::mat4 *parentGlobal,
	               const std::vector<Mesh> &hMeshes)
		: index(idx), parentIndex(parentIdx), centroid(0, 0, 0)
	{
		// placement: global = parent's global * local, and the three matrices derived from it (scene.cu:363-367)
		localTransform = toMat4(node.transform);
		globalTransform = parentGlobal ? *parentGlobal * localTransform : localTransform;
		globalInverseTransform = math::inverse(globalTransform);
		globalNormalTransform = math::transpose(math::inverse(math::toMat3(globalTransform)));
		globalInverseNormalTransform = math::inverse(globalNormalTransform);

		// the node's meshes: numbers into the scene's mesh list
		const uint32_t *ids = scene.node_mesh_ids + node.mesh_begin;
		const std::vector<int> mine(ids, ids + node.mesh_count);
		meshes = vector<int>(mine);
		boundingBox.min = boundingBox.max = math::vec3(0, 0, 0);
		if (mine.empty())
			return;

		// extent: the local box around the meshes' boxes, taken to world space corner by corner; centroid: the mean of
		// the meshes' centroids, taken to world space (scene.cu:372-400)
		PointFold local;
		for (int m : mine) {
			local.sum += hMeshes[size_t(m)].centroid;
			local.count++;
			local.enclose(hMeshes[size_t(m)].boundingBox);
		}
		boundingBox = transformBoundingBox(local.box);
		centroid = math::translate(globalTransform, local.mean());
	}
} // namespace acr
