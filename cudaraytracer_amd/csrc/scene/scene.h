// acr::Object, acr::Light, acr::Scene (reference src/scene/scene.h:12-129).
//
// Host half only.  The reference's __device__ members (Scene::intersect,
// lightPoint, pointLightAccum, spotLightAccum, Object::intersect) are the
// GPU kernels of device/acr_kernels.hip; the scene reaches them as the
// structure-of-arrays image produced by Scene::flatten().
#ifndef ACR_SCENE_SCENE_H
#define ACR_SCENE_SCENE_H

#include <string>
#include <vector>

#include "../geometry/mesh.h"
#include "../math/math.h"
#include "camera.h"

namespace acr
{
	class Scene;

	class Object // scene.h:16-50
	{
	public:
		Object() : index(0), parentIndex(-1) {}
		// the reference's Object(const aiNode*, int, Object*, host_vector<Mesh>&) (scene.cu:359-405)
		// parentGlobal: the parent's global transform, null for the root
		Object(const acr_import_scene &scene, const acr_import_node &node, int index, int parentIndex, const math::mat4 *parentGlobal,
		       const std::vector<Mesh> &hMeshes);

		int index;
		int parentIndex;
		math::vec3 centroid;
		BoundingBox boundingBox;

		vector<int> children;
		vector<int> meshes;
		math::mat4 globalTransform;
		math::mat4 localTransform;
		math::mat3 globalNormalTransform;

		math::mat4 globalInverseTransform;
		math::mat3 globalInverseNormalTransform;

		BoundingBox transformBoundingBox(BoundingBox aabb) const;

		const BoundingBox &getBoundingBox(void *) const { return boundingBox; }
		const math::vec3 &getCentroid(void *) const { return centroid; }
	};

	class Light // scene.h:52-80
	{
	public:
		Light() : type(POINT), attConstant(0), attLinear(0), attQuadratic(0), innerConeAngle(0), outerConeAngle(0) {}
		explicit Light(const acr_import_light &l); // scene.cu:303-330

		enum Type
		{
			POINT,
			DIRECTIONAL,
			SPOT
		};

		Type type;

		Color3 ambient;
		Color3 diffuse;
		Color3 specular;

		math::vec3 position;
		math::vec3 direction;

		float attConstant;
		float attLinear;
		float attQuadratic;

		float innerConeAngle;
		float outerConeAngle;
	};

	// Owning storage behind an acr_scene_soa (include/acr_hip.h).
	struct SceneArrays
	{
		std::vector<uint32_t> snodeStart, snodeEnd;
		std::vector<float> snodeLeft, snodeRight;
		std::vector<uint8_t> snodeFlags;
		std::vector<float> objTransform, objInvTransform, objNormal, objInvNormal, objWorldMin, objWorldMax;
		std::vector<uint32_t> objMeshBegin, objMeshCount, objMeshIds;
		std::vector<float> meshMin, meshMax;
		std::vector<uint32_t> meshMaterial, meshDepth, meshNodeBegin, meshTriBegin;
		std::vector<uint32_t> mnodeStart, mnodeEnd;
		std::vector<float> mnodeLeft, mnodeRight;
		std::vector<uint8_t> mnodeFlags;
		std::vector<float> triP0, triP1, triP2, triN0, triN1, triN2;
		std::vector<uint32_t> triPrimId;
		std::vector<float> matAmbient, matDiffuse, matSpecular, matRefractiveIndex;
		std::vector<int32_t> lightType;
		std::vector<float> lightDiffuse, lightPosition, lightDirection, lightAttenuation, lightCone;
		acr_scene_soa view{};
	};

	class Scene // scene.h:82-127
	{
		friend class Object;

	public:
		struct Args
		{
			const char *filePath;
			size_t sceneBihDepth = 6; // MAX_DEPTH of BIH<Object> (bih.h:68)
			size_t meshBihDepth = 6;  // MAX_DEPTH of BIH<Face>
		};

		Scene() : rootIndex(-1) {}
		// loads a .blend file through the built-in reader (the reference calls assimp here, scene.cu:28-46);
		// throws std::runtime_error where the reference prints and exits
		explicit Scene(const Args &args);
		Scene(const acr_import_scene &scene, size_t sceneBihDepth = 6, size_t meshBihDepth = 6);
		~Scene();

		const BIH<Object> &getObjects() const { return objects; }
		// the device-ready image of this scene; built on first use
		const acr_scene_soa &flatten() const;

	private:
		void loadScene(const acr_import_scene &scene, size_t sceneBihDepth, size_t meshBihDepth);
		void loadLights(const acr_import_scene &scene, std::vector<Light> &hLights);
		void loadMaterials(const acr_import_scene &scene);
		void loadCamera(const acr_import_scene &scene, std::string &camName);
		void loadMeshes(const acr_import_scene &scene, std::vector<Mesh> &hMeshes, size_t meshBihDepth);
		void loadObjects(const acr_import_scene &scene, const std::string &camName, std::vector<Light> &hLights, std::vector<Mesh> &hMeshes,
		                 size_t sceneBihDepth);

		BIH<Object> objects; // scene.h:117
		mutable SceneArrays *arrays = nullptr;

	public:
		vector<Material> materials;
		vector<Mesh> meshes;
		vector<Light> lights;

		BoundingBox boundingBox;
		int rootIndex;
		Camera camera;
	};
} // namespace acr

#endif
