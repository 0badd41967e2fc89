// Device-side scene image: one contiguous blob of 16-byte slots in HBM that a
// workgroup can also stage into LDS wholesale.  Built from acr_scene_soa by
// acr_hip_upload_scene, read by the kernels in acr_kernels.hip.
//
// Record formats (one uint4 = one slot; f = float bits, u = uint32):
//   BIH node   1 slot : { f left, f right, u start, u end | axis<<28 | leaf<<30 | leftBigger<<31 }   (bih.h:92-98)
//   object    16 slots: [0]     world box min - margin (3 f), u flags (kObj* below)
//                       [1]     world box max + margin (3 f), 0
//                       [2..5]  globalInverseTransform columns
//                       [6..8]  globalInverseNormalTransform (9 f), u mesh_begin, u mesh_count, 0
//                       [9..12] globalTransform columns
//                       [13..15] globalNormalTransform (9 f), 0, 0, 0
//   mesh       3 slots: { f min.xyz, u material } { f max.xyz, u node_begin } { u tri_begin, u depth, 0, 0 }
//   triangle   3 slots: { a.x a.y a.z e1.x } { e1.y e1.z e2.x e2.y } { e2.z, u prim_id, 0, 0 }
//                       with a = p0, e1 = p1 - p0, e2 = p2 - p0 (math.h:132-133)
//   normals    3 slots: { n0.x n0.y n0.z n1.x } { n1.y n1.z n2.x n2.y } { n2.z 0 0 0 }
//   material   3 slots: { ambient.xyz, f refractiveIndex } { diffuse.xyz, 0 } { specular.xyz, 0 }
//   light      4 slots: { position.xyz, u type } { direction.xyz, f attConstant }
//                       { diffuse.xyz, f attLinear } { f attQuadratic, f innerCone, f outerCone, 0 }
//   obj->mesh id list : packed uint32, 4 per slot
//
// Section order: the small tables first (scene nodes, objects, id list,
// meshes, materials, lights), then the large ones (mesh nodes, triangles,
// normals), so "stage the small tables" and "stage everything" are both a
// copy of a prefix.
#ifndef ACR_DEVICE_SCENE_H
#define ACR_DEVICE_SCENE_H

#include <stdint.h>

namespace acr
{
namespace dev
{
	constexpr uint32_t kNodeLeafBit = 1u << 30;
	constexpr uint32_t kNodeLeftBiggerBit = 1u << 31; // internal node: the left child holds more items than the right (bih.h:269-270)
	constexpr uint32_t kNodeAxisShift = 28;
	constexpr uint32_t kNodeEndMask = (1u << 28) - 1;

	constexpr uint32_t kObjectSlots = 16;
	constexpr uint32_t kObjPrefilter = 1u;  // slots 0-1 hold a usable conservative world box
	constexpr uint32_t kObjInvAffine = 2u;  // globalInverseTransform has bottom row (0,0,0,1): hom.w == 1, the divide is the identity
	constexpr uint32_t kObjFwdAffine = 4u;  // same for globalTransform
	constexpr uint32_t kMeshSlots = 3;
	constexpr uint32_t kTriSlots = 3;
	constexpr uint32_t kNormalSlots = 3;
	// slots allocated (zeroed) behind the blob's last one: kernels fetch short runs of consecutive slots from one
	// address without checking where the table they belong to ends
	constexpr uint32_t kBlobPadSlots = 16;
	constexpr uint32_t kMaterialSlots = 3;
	constexpr uint32_t kLightSlots = 4;

	// ray counters are spread over this many slots of 4 words so that the
	// workgroups' atomic adds do not all land on one address
	constexpr uint32_t kCounterSlots = 2048; // many, so that the waves of a launch do not queue up on a few addresses

	struct SceneHeader
	{
		// slot offsets of each section inside the blob
		uint32_t off_snodes, off_objects, off_objmesh, off_meshes, off_materials, off_lights;
		uint32_t off_mnodes, off_tris, off_normals;
		uint32_t small_slots;   // slots [0, small_slots) hold the small tables
		uint32_t total_slots;
		uint32_t scene_depth, max_mesh_depth;
		uint32_t num_objects, num_meshes, num_materials, num_lights, num_triangles;
		float scene_min[3], scene_max[3];
		// 1: a ray whose direction is the zero vector (glm::refract's answer to total internal reflection, renderer.cu:299)
		// hits nothing in this scene, whatever the traversal does with its infinite 1 / d: every object's matrices are
		// finite, so the local direction is a zero vector too, cross(d, e2) and the determinant (geometry.h:123-146) are
		// zeros for finite edges, and |det| < 0.001 rejects every triangle.  The kernels then skip the traversal.
		uint32_t zero_dir_misses;
	};
} // namespace dev
} // namespace acr

#endif
