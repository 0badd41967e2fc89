// Kernel argument blocks, shared by the kernels (acr_kernels.hip) and the C ABI
// implementation that fills them (acr_hip.cpp).  Passed by value: they take
// the place of the reference's `__constant__ DevParams` (renderer.cu:14-23).
#ifndef ACR_RENDER_PARAMS_H
#define ACR_RENDER_PARAMS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_scene.h"

// A render launch is a grid of blocks of ACR_BLOCK_SLOTS pixel slots (16 x 16 pixels, four 8 x 8 waves); a
// block is rendered by ACR_BLOCK_SLOTS / ACR_BLOCK workgroups of ACR_BLOCK work-items.
#define ACR_BLOCK_SLOTS 256
#ifndef ACR_BLOCK
#define ACR_BLOCK 256
#endif

namespace acr
{
namespace dev
{
	constexpr uint32_t kParallelMaxObjects = 64, kParallelWords = kParallelMaxObjects / 8 + 1;

	struct RenderParams
	{
		SceneHeader scene;
		const uint4 *blob;
		// camera (camera.h:14-16), aspect already set to width/height (renderer.cu:179)
		float cam_aspect, cam_hfov;
		float cam_position[3], cam_forward[3], cam_up[3];
		// the camera-only terms of get_pixel_dir (renderer.cu:197-200) and of the pixel step (renderer.cu:360-361)
		float cam_cr[3], cam_cu[3], cam_dist;
		float inv_width, inv_height;
		// frame
		uint32_t width, height;
		uint32_t blocks_x;          // ceil(width / 16), full-frame mode
		uint64_t seed, frames;
		uint32_t max_bounces;
		// tile-list mode (num_tiles > 0)
		uint32_t tile_w, tile_h, tiles_x, num_tiles;
		uint32_t tiles_into_frame;  // the output buffers are whole frames: a listed tile's pixel (x, y) goes to y * width + x
		const uint32_t *tile_ids;   // device
		// buffers
		int32_t *pixel_values;      // device or null
		float4 *screen;
		uint32_t *primid;           // device or null
		float *hit_t;               // device or null
		unsigned long long *counters; // device [kCounterSlots][4] or null: pixels, shadow rays, bounce rays, primary hits
		uint32_t *tile_rays;          // device [num_tiles] or null
		uint32_t *tile_cycles;        // device [num_tiles] or null
		// Which infinite components of 1 / d a PRIMARY ray may keep in the interval form: four bits per object (bit a:
		// the camera's coordinate on axis a, in the object's local space, equals no mesh box bound or tree plane of
		// the object on that axis), eight objects per word, and in the last word the same for the tree over objects
		// and the objects' world boxes.  All zero: such rays go to the six-plane pass.
		uint32_t parallel_ok[kParallelWords];
		uint32_t lds_stack_slot;    // first LDS slot after the staged scene
		uint32_t lds_cstack_off;    // floats from there to the per-bounce-level direct-term columns
		uint32_t lds_wave_off;      // floats from there to the per-wave task tables of the wave-cooperative kernel
		// Interval traversal of deep mesh trees: the top mesh_glob_levels levels of the waiting-child column live in
		// global memory (mesh_glob[level * mesh_glob_stride + slot], slot = workgroup * ACR_BLOCK + work-item), the
		// levels below them in LDS.  0 levels for trees that fit LDS whole.
		uint32_t mesh_glob_levels, mesh_glob_stride;
		float2 *mesh_glob;
		// Pixels the interval-traversal launch cannot serve: their slot numbers (workgroup * 256 + work-item) are
		// appended here and rendered by the six-plane list pass that follows in the stream.
		uint32_t *redo_count;       // device: entries in redo_list (this frame)
		uint32_t *redo_count_next;  // device: the counter the next frame will use; the list pass zeroes it
		uint32_t *redo_list;        // device [grid * 256]
		uint64_t *path_keys;        // device [width * height] or null: per-pixel traversal-path key (full-frame launches)
		uint32_t *block_cost;        // device [blocks of the launch] or null: wave clock ticks spent per block, summed over its waves
		const uint32_t *block_order; // device [blocks of the frame] or null: which block of 16 x 16 pixels the k-th block of the grid renders (full-frame launches)
	};

	// Wave-cooperative kernel: a wave's task list holds kWaveCap object tests (ACR_WAVE_TASK_SLOTS per lane), and a
	// wave owns waveLdsWords() 4-byte words of LDS: the tasks' results (kWaveCap) and the task list (kWaveCap 32-bit
	// entries).  A lane keeps the objects whose boxes its ray enters as a 64-bit mask, hence at most 64 objects; a
	// list entry also names the triangle and the material of the task's hit in kTaskTriBits and the bits left (18 in all).
#ifndef ACR_WAVE_TASK_SLOTS
#define ACR_WAVE_TASK_SLOTS 2
#endif
	constexpr uint32_t kWaveCap = 64u * ACR_WAVE_TASK_SLOTS;
	constexpr uint32_t kWaveMaxObjects = 64;
	constexpr uint32_t kTaskTriBits = 12;
	constexpr uint32_t kWaveMaxTriangles = 1u << kTaskTriBits, kWaveMaxMaterials = 1u << (18u - kTaskTriBits);
	constexpr uint32_t waveLdsWords() { return 2u * kWaveCap; }
	// slots from one object record to the next in that kernel's LDS copy of the scene (kObjectSlots in the blob)
#ifndef ACR_WAVE_OBJ_STRIDE
#define ACR_WAVE_OBJ_STRIDE 17
#endif
	constexpr uint32_t kWaveObjStride = ACR_WAVE_OBJ_STRIDE;
	static_assert(kWaveObjStride == kObjectSlots || kWaveObjStride == kObjectSlots + 1, "stageSceneSpread moves every record by its number");

	// XCDs the XCD-aware block order is written for (workgroup i of a launch on XCD i % 8)
	constexpr uint32_t kXcdsAssumed = 8;

	// workgroups of the list pass
	constexpr uint32_t kRedoGrid = 64;

	struct TraceParams
	{
		SceneHeader scene;
		const uint4 *blob;
		uint32_t n;
		const float *origins, *dirs, *tmax;
		uint8_t *hit;
		float *t;
		uint32_t *prim;
		float *position, *normal;
		uint32_t lds_stack_slot;
	};
} // namespace dev
} // namespace acr

extern "C" {
hipError_t acr_dev_launch_render(const acr::dev::RenderParams *p, int ldsMode, int fast, uint32_t grid, size_t ldsBytes, hipStream_t stream);
hipError_t acr_dev_launch_iota(int32_t *values, uint32_t n, hipStream_t stream);
hipError_t acr_dev_launch_reciprocal_check(unsigned long long *differ, hipStream_t stream);
hipError_t acr_dev_launch_vote_paths(const uint64_t *keys, int32_t *votes, uint32_t n, int reset, hipStream_t stream);
hipError_t acr_dev_launch_vote_keys(const int32_t *votes, uint64_t *keys, uint32_t n, hipStream_t stream);
// order[0..n) = the numbers 0..n-1, the ones with the largest cost[] first (256 cost classes)
hipError_t acr_dev_launch_order_blocks(uint32_t *cost, uint32_t n, uint32_t *order, hipStream_t stream); // leaves cost[0..n) at zero
// the same, XCD-aware: position i of order[] is taken to run on XCD i % 8, and the blocks of one XCD are neighbours on
// the Z-curve zseq[0..n) (a permutation of 0..n-1), most expensive first
hipError_t acr_dev_launch_order_blocks_regions(uint32_t *cost, uint32_t n, uint32_t *order, const uint32_t *zseq, hipStream_t stream);
hipError_t acr_dev_launch_trace(const acr::dev::TraceParams *p, int ldsMode, int fast, uint32_t grid, size_t ldsBytes, hipStream_t stream);
hipError_t acr_dev_launch_scatter_tiles(const void *tiles, const uint32_t *tileIdsDev, uint32_t numTiles, uint32_t tileW, uint32_t tileH,
                                        uint32_t width, uint32_t height, void *frame, hipStream_t stream);
}

#endif
