// Implementation of the C ABI declared in include/acr_hip.h: context and
// buffer management, scene validation + packing into the device blob
// (device/device_scene.h), and the kernel launches.
//
// There is no CPU fallback anywhere in this file: every compute entry point
// either launches a HIP kernel or returns an error.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/acr_hip.h"
#include "../math/math.h"
#include "../device/device_scene.h"
#include "../device/render_params.h"

using acr::dev::RenderParams;
using acr::dev::SceneHeader;
using acr::dev::TraceParams;

// Scratch of acr_hip_render_static: the list of pixel slots a launch hands to its list pass and the list's two
// counters, in device memory that exists from the moment the library is loaded, so that the call allocates
// nothing.  Each GPU holds its own copy; its address there comes from hipGetSymbolAddress (acr_hip_create).
namespace
{
	struct StaticScratch
	{
		uint32_t redoCount[64]; // [0], [1]: the two counters; the rest keeps the list aligned
		uint32_t redoList[ACR_STATIC_MAX_SLOTS];
	};
}
static __device__ __align__(256) unsigned char scratch[sizeof(StaticScratch)];

struct acr_hip_ctx
{
	int device = 0;
	uint32_t xcds = 0; // XCDs of the device (hipDeviceAttributeNumberOfXccs, or the environment's ACR_XCDS); 0: unknown
	StaticScratch *staticScratch = nullptr; // this device's copy of `scratch`
	hipStream_t ownStream = nullptr;
	hipStream_t stream = nullptr;
	std::string error;

	// scene
	bool haveScene = false;
	SceneHeader header{};
	uint4 *blob = nullptr;
	size_t blobBytes = 0;

	// camera
	bool haveCamera = false;
	acr_camera camera{};

	// context-owned buffers (acr_hip_resize)
	uint32_t width = 0, height = 0;
	float4 *screen = nullptr;
	uint32_t *primid = nullptr;
	float *hitT = nullptr;
	int32_t *pixelValues = nullptr;
	uint64_t *pathKeys = nullptr;
	int32_t *pathVotes = nullptr; // [64][width * height], allocated on first request (acr_hip_path_votes)

	// per-launch scratch
	unsigned long long *counters = nullptr; // device [kCounterSlots][4]
	// Device copies of the tile lists (render / scatter).  A list equal to the previous one is not uploaded again.
	// A new one goes into the next of kTileRing slots: pinned host memory, a device buffer and an event recorded
	// behind the launch that reads the slot, so that a caller which changes its list with every frame (per-frame
	// rebalancing across GPUs) never waits for the stream: the slot it overwrites was last read kTileRing
	// launches ago.
	static constexpr int kTileRing = 4;
	struct TileList
	{
		struct Slot
		{
			uint32_t *dev = nullptr, *pinned = nullptr;
			size_t cap = 0;
			hipEvent_t read = nullptr; // recorded after the last launch that reads dev
			bool inUse = false;
		} slot[kTileRing];
		int current = -1;
		std::vector<uint32_t> host; // contents of slot[current]
		uint32_t *dev() const { return current < 0 ? nullptr : slot[current].dev; }
	} renderTiles, scatterTiles;
	// interval traversal (see acr_kernels.hip): usable for this scene, forced off by the caller, and the
	// buffers through which its launches hand pixels to the six-plane list pass
	// Primary rays parallel to a coordinate plane (a direction component of exactly zero, 1 / d infinite): the interval
	// form serves them too when no plane or box bound they can meet on that axis equals the camera's coordinate there
	// (then no 0 * infinity, hence no NaN, can arise, and infinite slab distances compare like the reference's).
	// Kept per axis, sorted, for the tree over objects and per object (in its local space) for its mesh trees; scenes
	// with more than kParallelMaxObjects objects go without.
	struct AxisValues
	{
		std::vector<float> v[3];
	};
	struct LocalSpace
	{
		float invM[16];
		bool affine;
		AxisValues bounds;
	};
	AxisValues worldBounds;
	std::vector<LocalSpace> localSpaces;
	bool parallelOk = false;       // the tables above are filled
	bool parallelBitsValid = false; // parallelBits belong to the current camera and scene
	uint32_t parallelBits[acr::dev::kParallelWords] = {};
	bool intervalOk = false;
	bool leafOrderOk = false; // near-side-first walks of mesh trees can break exact ties by item position
	bool soloScene = false;   // one object with one mesh: the tree over objects is a single leaf
	bool soloOff = false;     // ACR_SOLO=0: such scenes are rendered by the general kernel (for comparisons)
	int traversalMode = ACR_TRAVERSAL_AUTO;
	uint32_t listAllCount = 0; // ACR_TRAVERSAL_LIST_PASS: source of the copy that sets the list's length
	uint32_t *redoCount = nullptr; // device [2], used alternately by successive frames
	uint32_t *redoList = nullptr;
	size_t redoCap = 0;
	uint32_t redoParity = 0;
	float2 *meshGlob = nullptr;    // global part of the mesh tree's waiting-child column (deep trees only)
	size_t meshGlobBytes = 0;
	uint32_t sceneDepthGiven = 0;  // MAX_DEPTH of the scene tree as uploaded (before unused levels were trimmed)
	// scratch of acr_hip_sort_pixel_map
	// launch order of the blocks of 16 x 16 pixel slots, from measured cost (acr_hip_set_block_order)
	uint32_t blockOrderEvery = 64;   // longest interval between two measurements, in launches; 0 = natural order
	uint32_t blockOrderInterval = 4; // current interval: 4 after a camera move, doubling while the view stays
	uint32_t *blockCost = nullptr, *blockOrder = nullptr;
	// the blocks of the current full-frame size along their Z-curve, for the XCD-aware order of scenes read through the L2
	uint32_t *blockZ = nullptr;
	uint32_t blockZCount = 0, blockZBlocksX = 0;
	std::vector<uint32_t> blockZHost;
	size_t blockCap = 0;
	uint64_t blockKey = 0;      // configuration (frame size, tile list) the order belongs to
	bool blockOrderValid = false;
	bool blockCostZero = false;   // blockCost holds zeros (the sort that reads it leaves it so)
	uint32_t blockOrderAge = 0; // launches since the order was measured
	void *sortTemp = nullptr;
	size_t sortTempBytes = 0;
	uint64_t *sortKeysOut = nullptr;
	int32_t *sortValuesIn = nullptr;
	size_t sortCap = 0;
	hipEvent_t evStart = nullptr, evStop = nullptr;
	bool timed = false;
	bool counted = false; // the last render also counted its rays (ACR_STATS_FRAME / ACR_STATS_ACCUMULATE)
	acr_stats stats{};
	size_t lastSlots = 0; // pixel slots written by the last render
	bool statsStatic = false; // the last render was acr_hip_render_static: its list counter is scratch's
};

namespace
{
	constexpr uint32_t kBlock = ACR_BLOCK;
	constexpr uint32_t kSlots = ACR_BLOCK_SLOTS;
	constexpr uint32_t kMinBlocksForOrder = 1024; // four workgroups per CU are resident at once: a smaller launch starts all its blocks together anyway
	constexpr uint32_t kGroupsPerBlock = kSlots / kBlock; // workgroups per block of 16 x 16 pixel slots
	constexpr uint32_t kBlockRows = kSlots / 16;
	constexpr uint32_t kCounterSlots = acr::dev::kCounterSlots;
	// LDS budget per workgroup for staging: leaves room for 2+ workgroups per CU (160 KiB)
#ifndef ACR_ORDER_INTERVAL_AFTER_MOVE
#define ACR_ORDER_INTERVAL_AFTER_MOVE 4
#endif
	constexpr uint32_t kOrderIntervalAfterMove = ACR_ORDER_INTERVAL_AFTER_MOVE;
	constexpr size_t kLdsBudget = 64 * 1024;

	int fail(acr_hip_ctx *ctx, int code, const char *fmt, ...)
	{
		char buf[512];
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(buf, sizeof(buf), fmt, ap);
		va_end(ap);
		if (ctx) ctx->error = buf;
		return code;
	}

#define ACR_HIP_TRY(ctx, expr)                                                                          \
	do {                                                                                                \
		hipError_t e_ = (expr);                                                                         \
		if (e_ != hipSuccess) return fail(ctx, ACR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));  \
	} while (0)

	uint32_t fbits(float f)
	{
		uint32_t u;
		memcpy(&u, &f, 4);
		return u;
	}

	// Walks the nodes reachable from the root and checks what the kernels rely on:
	// an internal node has both children inside the heap, a leaf's item range is
	// inside [0, itemCount].
	// True when every split plane a traversal can reach lies inside the box of its node (the root box
	// clipped by the ancestors' planes) and the root box has min <= max: the precondition of the interval
	// form of the slab test.  sift() always builds such trees for non-empty item sets.
	bool planesInsideBoxes(const float *rootMin, const float *rootMax, const float *left, const float *right, const uint8_t *flags)
	{
		struct Item { uint32_t i; float mn[3], mx[3]; };
		for (int k = 0; k < 3; k++)
			if (!(rootMin[k] <= rootMax[k]) || !std::isfinite(rootMin[k]) || !std::isfinite(rootMax[k])) return false;
		std::vector<Item> todo(1);
		todo[0].i = 0;
		memcpy(todo[0].mn, rootMin, 12);
		memcpy(todo[0].mx, rootMax, 12);
		while (!todo.empty()) {
			const Item it = todo.back();
			todo.pop_back();
			if (flags[it.i] & ACR_NODE_LEAF) continue;
			const uint32_t a = flags[it.i] & ACR_NODE_AXIS_MASK;
			const float l = left[it.i], r = right[it.i];
			if (!(it.mn[a] <= l && l <= it.mx[a] && it.mn[a] <= r && r <= it.mx[a])) return false;
			Item L = it, R = it;
			L.i = 2 * it.i + 1; L.mx[a] = l;
			R.i = 2 * it.i + 2; R.mn[a] = r;
			todo.push_back(R);
			todo.push_back(L);
		}
		return true;
	}

	// leavesAscend (optional): whether the non-empty item ranges of the leaves, taken in the order the
	// reference's traversal visits them (left child first), ascend without overlap -- then the position of
	// an item in the item array says which of two items the reference tests first.  sift() partitions the
	// item array in place, left part first, so its trees always do.
	bool checkBih(uint32_t depth, const uint32_t *start, const uint32_t *end, const uint8_t *flags, uint32_t itemCount, std::string &why,
	              uint32_t *levelsUsed, bool *leavesAscend = nullptr)
	{
		uint32_t deepest = 0, lastEnd = 0;
		if (leavesAscend) *leavesAscend = true;
		if (depth < 1 || depth > ACR_MAX_BIH_DEPTH) { why = "BIH depth out of range"; return false; }
		const uint64_t count = (1ull << depth) - 1;
		std::vector<uint32_t> todo(1, 0u);
		while (!todo.empty()) {
			const uint32_t i = todo.back();
			todo.pop_back();
			uint32_t level = 0;
			while ((2u << level) - 1 <= i) level++; // node i sits on level floor(log2(i + 1))
			if (level > deepest) deepest = level;
			if (flags[i] & ACR_NODE_LEAF) {
				if (start[i] <= end[i] && end[i] > itemCount) { why = "leaf item range outside the item array"; return false; }
				if (leavesAscend && start[i] < end[i]) {
					if (start[i] < lastEnd) *leavesAscend = false;
					lastEnd = end[i];
				}
				continue;
			}
			if ((flags[i] & ACR_NODE_AXIS_MASK) > 2) { why = "node axis > 2"; return false; }
			if (2ull * i + 2 >= count) { why = "internal node on the last level"; return false; }
			todo.push_back(2 * i + 2);
			todo.push_back(2 * i + 1);
		}
		if (levelsUsed) *levelsUsed = deepest + 1;
		return true;
	}

	// Node i of a heap-ordered tree with `count` packed nodes.
	void putNode(uint4 &dst, const float *left, const float *right, const uint32_t *start, const uint32_t *end, const uint8_t *flags, uint32_t i,
	             uint32_t count)
	{
		dst.x = fbits(left[i]);
		dst.y = fbits(right[i]);
		dst.z = start[i];
		dst.w = (end[i] & acr::dev::kNodeEndMask) | (uint32_t(flags[i] & ACR_NODE_AXIS_MASK) << acr::dev::kNodeAxisShift) |
		        ((flags[i] & ACR_NODE_LEAF) ? acr::dev::kNodeLeafBit : 0u);
		// "left got more than right" (bih.h:269-270), used by the traversal-path key
		const uint32_t l = 2 * i + 1, r = 2 * i + 2;
		if (!(flags[i] & ACR_NODE_LEAF) && r < count && (end[l] - start[l]) > (end[r] - start[r])) dst.w |= acr::dev::kNodeLeftBiggerBit;
	}

	struct LaunchPlan
	{
		int ldsMode;
		uint32_t stackSlot;
		size_t ldsBytes;
		uint32_t cstackOff;       // floats from the first column to the direct-term columns
		uint32_t waveOff;         // floats from the first column to the per-wave task tables; 0: the launch has none
		uint32_t wordsOff;        // floats from the first column to one word per work-item behind the columns (sharedWords)
		uint32_t meshGlobLevels;  // top levels of the mesh tree's waiting-child column kept in global memory
	};

	// Decides how much of the scene a workgroup stages into LDS and how the per-lane columns are laid out.
	// splitDeepColumns: false for traceKernel, which keeps every level of the waiting-child column in LDS and
	// needs room for the six-plane form's columns next to them.
	// waveTasks: also lay out the per-wave task tables of the wave-cooperative kernel, if the scene then still fits
	// LDS whole (plan.waveOff != 0 says it did).
	LaunchPlan planLds(const SceneHeader &h, uint32_t colorLevels, bool splitDeepColumns, bool waveTasks = false, bool fourWaves = false, bool sharedWords = false)
	{
		// Per tree level and lane: two floats for the interval traversal's waiting-child column, one for the
		// six-plane traversal's undo column (the list pass runs in the same LDS block).
		const size_t perLevel = size_t(2) * kBlock * 4;
		const size_t sixPlaneFloats = size_t(h.scene_depth) + h.max_mesh_depth;
		auto layout = [&](uint32_t meshLdsLevels, size_t &columnFloats) {
			const size_t intervalFloats = size_t(2) * (h.scene_depth + meshLdsLevels);
			// render launches run one form at a time in a workgroup and share the region; traceKernel runs both side by side
			columnFloats = (splitDeepColumns ? (intervalFloats > sixPlaneFloats ? intervalFloats : sixPlaneFloats) : intervalFloats + sixPlaneFloats) * kBlock;
			return (columnFloats + size_t(3) * colorLevels * kBlock) * 4;
		};
		LaunchPlan p;
		p.waveOff = 0;
		p.wordsOff = 0;
		p.meshGlobLevels = 0;
		size_t columnFloats = 0;
		size_t stackBytes = layout(h.max_mesh_depth, columnFloats);
		p.cstackOff = uint32_t(columnFloats);
		// the whole scene in LDS: kernels compiled for four waves per SIMD, i.e. four workgroups per CU
		const size_t waveBytes = size_t(kBlock / 64) * acr::dev::waveLdsWords() * 4;
		const uint32_t spread = h.num_objects * (acr::dev::kWaveObjStride - acr::dev::kObjectSlots);
		if (waveTasks && h.num_objects <= acr::dev::kWaveMaxObjects && h.num_triangles <= acr::dev::kWaveMaxTriangles &&
		    h.num_materials <= acr::dev::kWaveMaxMaterials && size_t(h.total_slots + spread) * 16 + stackBytes + waveBytes <= kLdsBudget) {
			p.ldsMode = 2;
			p.stackSlot = h.total_slots + spread; // that kernel stages the object records one slot further apart (kWaveObjStride)
			p.ldsBytes = size_t(p.stackSlot) * 16 + stackBytes + waveBytes;
			p.waveOff = uint32_t(stackBytes / 4);
			return p;
		}
		if (size_t(h.total_slots) * 16 + stackBytes <= kLdsBudget) {
			p.ldsMode = 2;
			p.stackSlot = h.total_slots;
			p.ldsBytes = size_t(p.stackSlot) * 16 + stackBytes;
			return p;
		}
		// Trees and triangles in HBM: those kernels are compiled for three waves per SIMD (no register spills), so
		// three workgroups share a CU's 160 KiB.  A deep mesh tree keeps as many of its deepest levels' waiting-child
		// entries in LDS as that allows (every level of a 22-level tree); the levels above go to global memory.
#ifndef ACR_HBM_LDS_KB
#define ACR_HBM_LDS_KB 52
#endif
		// fourWaves: the kernel for one deep mesh, compiled for four waves per SIMD (four workgroups share the CU's LDS)
#ifndef ACR_SOLO_LDS_KB
#define ACR_SOLO_LDS_KB 40
#endif
		const size_t kHbmBudget = size_t(fourWaves ? ACR_SOLO_LDS_KB : ACR_HBM_LDS_KB) * 1024;
		uint32_t meshLdsLevels = h.max_mesh_depth;
		if (splitDeepColumns) {
			// sharedWords: one more word per work-item behind the columns (the shared walk of a scene of several objects)
			const size_t fixed = size_t(3) * colorLevels * kBlock * 4 + size_t(h.small_slots) * 16 + (sharedWords ? size_t(kBlock) * 4 : 0);
			const size_t room = kHbmBudget > fixed ? kHbmBudget - fixed : 0;
			const size_t fit = room / perLevel;
			if (fit < size_t(h.scene_depth) + meshLdsLevels) {
				meshLdsLevels = fit > size_t(h.scene_depth) + 4 ? uint32_t(fit - h.scene_depth) : 4u;
				if (meshLdsLevels > h.max_mesh_depth) meshLdsLevels = h.max_mesh_depth;
			}
		}
		stackBytes = layout(meshLdsLevels, columnFloats);
		p.cstackOff = uint32_t(columnFloats);
		p.meshGlobLevels = h.max_mesh_depth - meshLdsLevels;
		if (size_t(h.small_slots) * 16 + stackBytes <= (splitDeepColumns ? kHbmBudget : kLdsBudget)) {
			p.ldsMode = 1;
			p.stackSlot = h.small_slots;
		} else {
			p.ldsMode = 0;
			p.stackSlot = 0;
		}
		p.ldsBytes = size_t(p.stackSlot) * 16 + stackBytes;
		if (sharedWords) {
			p.wordsOff = uint32_t(stackBytes / 4);
			p.ldsBytes += size_t(kBlock) * 4;
		}
		return p;
	}
	// Makes list.dev() hold `ids`.  A list equal to the previous one costs a memcmp; a new one is queued into the
	// next ring slot with an asynchronous copy from pinned memory (see TileList).  The caller records the slot's
	// event after the launch that reads it (tileListRead).
	int syncTileList(acr_hip_ctx *ctx, acr_hip_ctx::TileList &list, const uint32_t *ids, uint32_t n)
	{
		if (list.current >= 0 && list.host.size() == n && memcmp(list.host.data(), ids, sizeof(uint32_t) * n) == 0) return ACR_OK;
		const int next = (list.current + 1) % acr_hip_ctx::kTileRing;
		acr_hip_ctx::TileList::Slot &s = list.slot[next];
		if (!s.read) ACR_HIP_TRY(ctx, hipEventCreateWithFlags(&s.read, hipEventDisableTiming));
		if (s.inUse) ACR_HIP_TRY(ctx, hipEventSynchronize(s.read)); // a launch kTileRing lists ago: long over
		if (s.cap < n) {
			(void)hipFree(s.dev);
			(void)hipHostFree(s.pinned);
			s.dev = s.pinned = nullptr;
			s.cap = 0;
			const size_t cap = size_t(n) + n / 2 + 64; // lists of a rebalanced split vary in length: grow with room
			ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s.dev), sizeof(uint32_t) * cap));
			ACR_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&s.pinned), sizeof(uint32_t) * cap, hipHostMallocDefault));
			s.cap = cap;
		}
		memcpy(s.pinned, ids, sizeof(uint32_t) * n);
		ACR_HIP_TRY(ctx, hipMemcpyAsync(s.dev, s.pinned, sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream));
		list.host.assign(ids, ids + n);
		list.current = next;
		s.inUse = true;
		// until the first reader is queued the copy itself is what must finish before the slot is reused
		ACR_HIP_TRY(ctx, hipEventRecord(s.read, ctx->stream));
		return ACR_OK;
	}
	int tileListRead(acr_hip_ctx *ctx, acr_hip_ctx::TileList &list)
	{
		if (list.current >= 0) ACR_HIP_TRY(ctx, hipEventRecord(list.slot[list.current].read, ctx->stream));
		return ACR_OK;
	}
} // namespace

extern "C" {

int acr_hip_create(int device_ordinal, acr_hip_ctx **out)
{
	if (!out) return ACR_ERR_INVALID;
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || device_ordinal < 0 || device_ordinal >= count) return ACR_ERR_HIP;
	acr_hip_ctx *ctx = new acr_hip_ctx();
	ctx->device = device_ordinal;
	if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking) != hipSuccess ||
	    hipMalloc(reinterpret_cast<void **>(&ctx->counters), kCounterSlots * 4 * sizeof(unsigned long long)) != hipSuccess ||
	    hipMalloc(reinterpret_cast<void **>(&ctx->redoCount), 2 * sizeof(uint32_t)) != hipSuccess ||
	    hipMemset(ctx->redoCount, 0, 2 * sizeof(uint32_t)) != hipSuccess ||
	    hipMemset(ctx->counters, 0, kCounterSlots * 4 * sizeof(unsigned long long)) != hipSuccess ||
	    hipEventCreate(&ctx->evStart) != hipSuccess || hipEventCreate(&ctx->evStop) != hipSuccess) {
		acr_hip_destroy(ctx);
		return ACR_ERR_HIP;
	}
	ctx->stream = ctx->ownStream;
	// the device is current: this is its copy of the array
	void *sp = nullptr;
	if (hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp) {
		acr_hip_destroy(ctx);
		return ACR_ERR_HIP;
	}
	ctx->staticScratch = static_cast<StaticScratch *>(sp);
	// The XCD-aware block order assumes that workgroup i of a launch runs on XCD i % 8 (an unpartitioned MI300 / MI355
	// class device): anything else keeps the plain order by cost.
	int xccs = 0;
	if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device_ordinal) == hipSuccess && xccs > 0) ctx->xcds = uint32_t(xccs);
	if (const char *env = getenv("ACR_XCDS")) ctx->xcds = uint32_t(strtoul(env, nullptr, 10));
	*out = ctx;
	return ACR_OK;
}

void acr_hip_destroy(acr_hip_ctx *ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	if (ctx->ownStream) (void)hipStreamSynchronize(ctx->ownStream);
	(void)hipFree(ctx->blob);
	(void)hipFree(ctx->screen);
	(void)hipFree(ctx->primid);
	(void)hipFree(ctx->hitT);
	(void)hipFree(ctx->pixelValues);
	(void)hipFree(ctx->pathKeys);
	(void)hipFree(ctx->pathVotes);
	(void)hipFree(ctx->counters);
	(void)hipFree(ctx->redoCount);
	(void)hipFree(ctx->redoList);
	(void)hipFree(ctx->meshGlob);
	(void)hipFree(ctx->blockCost);
	(void)hipFree(ctx->blockOrder);
	(void)hipFree(ctx->blockZ);
	(void)hipFree(ctx->sortTemp);
	(void)hipFree(ctx->sortKeysOut);
	(void)hipFree(ctx->sortValuesIn);
	for (acr_hip_ctx::TileList *l : {&ctx->renderTiles, &ctx->scatterTiles})
		for (auto &sl : l->slot) {
			(void)hipFree(sl.dev);
			(void)hipHostFree(sl.pinned);
			if (sl.read) (void)hipEventDestroy(sl.read);
		}
	if (ctx->evStart) (void)hipEventDestroy(ctx->evStart);
	if (ctx->evStop) (void)hipEventDestroy(ctx->evStop);
	if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
	delete ctx;
}

const char *acr_hip_last_error(const acr_hip_ctx *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int acr_hip_set_stream(acr_hip_ctx *ctx, void *hip_stream)
{
	if (!ctx) return ACR_ERR_INVALID;
	if (ctx->stream != static_cast<hipStream_t>(hip_stream)) {
		// tile lists queued on the old stream are not ordered against the new one: have them uploaded again
		ctx->renderTiles.host.clear();
		ctx->scatterTiles.host.clear();
	}
	ctx->stream = static_cast<hipStream_t>(hip_stream);
	return ACR_OK;
}

int acr_hip_use_own_stream(acr_hip_ctx *ctx)
{
	if (!ctx) return ACR_ERR_INVALID;
	if (ctx->stream != ctx->ownStream) {
		ctx->renderTiles.host.clear();
		ctx->scatterTiles.host.clear();
	}
	ctx->stream = ctx->ownStream;
	return ACR_OK;
}

int acr_hip_set_block_order(acr_hip_ctx *ctx, uint32_t every)
{
	if (!ctx) return ACR_ERR_INVALID;
	ctx->blockOrderEvery = every;
	ctx->blockOrderInterval = 4;
	ctx->blockOrderValid = false;
	return ACR_OK;
}

int acr_hip_set_traversal(acr_hip_ctx *ctx, int mode)
{
	if (!ctx || (mode != ACR_TRAVERSAL_AUTO && mode != ACR_TRAVERSAL_SIX_PLANE && mode != ACR_TRAVERSAL_LIST_PASS && mode != ACR_TRAVERSAL_NESTED))
		return fail(ctx, ACR_ERR_INVALID, "unknown traversal mode");
	ctx->traversalMode = mode;
	return ACR_OK;
}

int acr_hip_sync(acr_hip_ctx *ctx)
{
	if (!ctx) return ACR_ERR_INVALID;
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return ACR_OK;
}

int acr_hip_selftest_reciprocal(acr_hip_ctx *ctx, uint64_t *mismatches)
{
	if (!ctx || !mismatches) return fail(ctx, ACR_ERR_INVALID, "null argument");
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	unsigned long long *differ = nullptr;
	ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&differ), sizeof(unsigned long long)));
	hipError_t e = hipMemsetAsync(differ, 0, sizeof(unsigned long long), ctx->stream);
	if (e == hipSuccess) e = acr_dev_launch_reciprocal_check(differ, ctx->stream);
	unsigned long long count = 0;
	if (e == hipSuccess) e = hipMemcpyAsync(&count, differ, sizeof(count), hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(differ);
	ACR_HIP_TRY(ctx, e);
	*mismatches = count;
	return ACR_OK;
}

// The kernels' reciprocal shortcut (acr::math::reciprocalNewton) is the division's result on gfx950; whether it also is
// on the GPU at hand is checked once per device and process, before the first scene is uploaded to it.
static int checkReciprocalOnce(acr_hip_ctx *ctx)
{
#ifdef ACR_NO_RCP
	(void)ctx;
	return ACR_OK; // built with divisions throughout: nothing to check
#else
	static std::mutex lock;
	static int verdict[64] = {}; // per device: 0 not run, 1 good, 2 bad
	const int d = ctx->device;
	if (d < 0 || d >= 64) return ACR_OK;
	std::lock_guard<std::mutex> g(lock);
	if (verdict[d] == 0) {
		if (const char *env = getenv("ACR_SKIP_RCP_SELFTEST"); env && env[0] == '1') {
			verdict[d] = 1;
		} else {
			uint64_t bad = 0;
			const int rc = acr_hip_selftest_reciprocal(ctx, &bad);
			if (rc != ACR_OK) return rc;
			verdict[d] = bad == 0 ? 1 : 2;
		}
	}
	if (verdict[d] == 2)
		return fail(ctx, ACR_ERR_UNSUPPORTED, "this GPU's reciprocal approximation plus one Newton step is not the correctly rounded 1 / x the kernels "
		            "were verified with (acr_hip_selftest_reciprocal): rebuild the library with -DACR_NO_RCP");
	return ACR_OK;
#endif
}

int acr_hip_upload_scene(acr_hip_ctx *ctx, const acr_scene_soa *s)
{
	if (!ctx || !s) return fail(ctx, ACR_ERR_INVALID, "null argument");
	if (const int rc = checkReciprocalOnce(ctx); rc != ACR_OK) return rc;
	std::string why;
	if (s->num_objects >= (1u << 28) || s->num_triangles >= (1u << 28) || s->num_mesh_nodes >= (1u << 28))
		return fail(ctx, ACR_ERR_UNSUPPORTED, "scene too large for 28-bit item indices");
	if (!s->snode_start || !s->snode_end || !s->snode_left || !s->snode_right || !s->snode_flags)
		return fail(ctx, ACR_ERR_INVALID, "scene BIH arrays missing");
	// Only the levels a traversal can reach are packed: a tree built with MAX_DEPTH 6 over two
	// triangles uses 2 of its 6 levels, and LDS space is what limits residency.
	uint32_t sceneLevels = 1;
	if (!checkBih(s->scene_depth, s->snode_start, s->snode_end, s->snode_flags, s->num_objects, why, &sceneLevels))
		return fail(ctx, ACR_ERR_INVALID, "scene BIH: %s", why.c_str());
	if (s->num_objects && (!s->obj_transform || !s->obj_inv_transform || !s->obj_normal_transform || !s->obj_inv_normal_transform ||
	                       !s->obj_mesh_begin || !s->obj_mesh_count))
		return fail(ctx, ACR_ERR_INVALID, "object arrays missing");
	for (uint32_t i = 0; i < s->num_objects; i++) {
		if (uint64_t(s->obj_mesh_begin[i]) + s->obj_mesh_count[i] > s->num_obj_mesh_ids)
			return fail(ctx, ACR_ERR_INVALID, "object %u: mesh list outside obj_mesh_ids", i);
	}
	for (uint32_t i = 0; i < s->num_obj_mesh_ids; i++)
		if (s->obj_mesh_ids[i] >= s->num_meshes) return fail(ctx, ACR_ERR_INVALID, "obj_mesh_ids[%u] out of range", i);
	uint32_t maxMeshDepth = 1;
	std::vector<uint32_t> meshLevels(s->num_meshes, 1), meshNodeAt(s->num_meshes, 0);
	uint32_t packedMeshNodes = 0;
	bool leafOrderOk = true; // item position == the reference's test order, in every mesh tree
	for (uint32_t m = 0; m < s->num_meshes; m++) {
		const uint32_t d = s->mesh_depth[m];
		if (d < 1 || d > ACR_MAX_BIH_DEPTH) return fail(ctx, ACR_ERR_UNSUPPORTED, "mesh %u: BIH depth %u", m, d);
		const uint64_t count = (1ull << d) - 1;
		if (uint64_t(s->mesh_node_begin[m]) + count > s->num_mesh_nodes) return fail(ctx, ACR_ERR_INVALID, "mesh %u: nodes outside the pool", m);
		const uint32_t nb = s->mesh_node_begin[m];
		const uint32_t triBegin = s->mesh_tri_begin[m];
		if (triBegin > s->num_triangles) return fail(ctx, ACR_ERR_INVALID, "mesh %u: tri_begin out of range", m);
		bool ascend = true;
		if (!checkBih(d, s->mnode_start + nb, s->mnode_end + nb, s->mnode_flags + nb, s->num_triangles - triBegin, why, &meshLevels[m], &ascend))
			return fail(ctx, ACR_ERR_INVALID, "mesh %u BIH: %s", m, why.c_str());
		leafOrderOk = leafOrderOk && ascend;
		if (s->mesh_material[m] >= s->num_materials) return fail(ctx, ACR_ERR_INVALID, "mesh %u: material index out of range", m);
		if (meshLevels[m] > maxMeshDepth) maxMeshDepth = meshLevels[m];
		meshNodeAt[m] = packedMeshNodes;
		packedMeshNodes += (1u << meshLevels[m]) - 1;
	}
	for (uint32_t i = 0; i < s->num_lights; i++)
		if (s->light_type[i] < 0 || s->light_type[i] > 2) return fail(ctx, ACR_ERR_INVALID, "light %u: unknown type", i);
	// does the scene qualify for the interval traversal?
	bool intervalOk = planesInsideBoxes(s->scene_min, s->scene_max, s->snode_left, s->snode_right, s->snode_flags);
	for (uint32_t m = 0; m < s->num_meshes && intervalOk; m++) {
		const uint32_t nb = s->mesh_node_begin[m];
		intervalOk = planesInsideBoxes(s->mesh_min + 3 * size_t(m), s->mesh_max + 3 * size_t(m), s->mnode_left + nb, s->mnode_right + nb, s->mnode_flags + nb);
	}

	// ... and are its triangles small enough for the triangle test's reciprocal?  (|det| <= |e1| |rd| |e2| must stay
	// below 2^126 for normalised directions, see myIntersectRayTriangleEdges<NEWTON>; finite edges, |e1| |e2| < 2^120)
	for (uint32_t i = 0; i < s->num_triangles && intervalOk; i++) {
		const float *a = s->tri_p0 + 3 * size_t(i), *b = s->tri_p1 + 3 * size_t(i), *c = s->tri_p2 + 3 * size_t(i);
		double l1 = 0, l2 = 0;
		for (int k = 0; k < 3; k++) {
			const double e1 = double(b[k] - a[k]), e2 = double(c[k] - a[k]); // the edges as the blob holds them (fp32 differences)
			l1 += e1 * e1;
			l2 += e2 * e2;
		}
		if (!(l1 * l2 < 0x1p240)) intervalOk = false; // also catches NaN and infinite edges
	}

	// ---- pack ----
	SceneHeader h{};
	const uint32_t snodeCount = (1u << sceneLevels) - 1;
	uint32_t at = 0;
	h.off_snodes = at; at += snodeCount;
	h.off_objects = at; at += s->num_objects * acr::dev::kObjectSlots;
	h.off_objmesh = at; at += (s->num_obj_mesh_ids + 3) / 4;
	h.off_meshes = at; at += s->num_meshes * acr::dev::kMeshSlots;
	h.off_materials = at; at += s->num_materials * acr::dev::kMaterialSlots;
	h.off_lights = at; at += s->num_lights * acr::dev::kLightSlots;
	h.small_slots = at;
	h.off_mnodes = at; at += packedMeshNodes;
	h.off_tris = at; at += s->num_triangles * acr::dev::kTriSlots;
	h.off_normals = at; at += s->num_triangles * acr::dev::kNormalSlots;
	h.total_slots = at;
	h.scene_depth = sceneLevels;
	h.max_mesh_depth = maxMeshDepth;
	h.num_objects = s->num_objects;
	h.num_meshes = s->num_meshes;
	h.num_materials = s->num_materials;
	h.num_lights = s->num_lights;
	h.num_triangles = s->num_triangles;
	memcpy(h.scene_min, s->scene_min, 12);
	memcpy(h.scene_max, s->scene_max, 12);
	{
		// see SceneHeader::zero_dir_misses: finite triangle edges (checked above for intervalOk) and finite matrices
		bool finite = intervalOk;
		for (uint32_t i = 0; i < s->num_objects && finite; i++) {
			for (int k = 0; k < 16; k++) finite = finite && std::isfinite(s->obj_transform[16 * size_t(i) + k]) && std::isfinite(s->obj_inv_transform[16 * size_t(i) + k]);
			for (int k = 0; k < 9; k++) finite = finite && std::isfinite(s->obj_normal_transform[9 * size_t(i) + k]) && std::isfinite(s->obj_inv_normal_transform[9 * size_t(i) + k]);
		}
		h.zero_dir_misses = finite ? 1u : 0u;
	}

	std::vector<uint4> blob(size_t(h.total_slots) + acr::dev::kBlobPadSlots, make_uint4(0, 0, 0, 0)); // see kBlobPadSlots
	for (uint32_t i = 0; i < snodeCount; i++)
		putNode(blob[h.off_snodes + i], s->snode_left, s->snode_right, s->snode_start, s->snode_end, s->snode_flags, i, snodeCount);
	// margin of the conservative world boxes: three to four orders of magnitude above fp32 rounding at scene scale
	double diag2 = 0;
	for (int k = 0; k < 3; k++) diag2 += double(s->scene_max[k] - s->scene_min[k]) * double(s->scene_max[k] - s->scene_min[k]);
	const float margin = std::isfinite(diag2) ? float(std::fmax(2e-3, 1e-3 * std::sqrt(diag2))) : 0.0f;
	for (uint32_t i = 0; i < s->num_objects; i++) {
		uint32_t *w = reinterpret_cast<uint32_t *>(&blob[h.off_objects + i * acr::dev::kObjectSlots]);
		const float *M = s->obj_transform + 16 * size_t(i), *MI = s->obj_inv_transform + 16 * size_t(i);
		uint32_t flags = 0;
		auto affine = [](const float *m) { return m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f; };
		if (affine(MI)) flags |= acr::dev::kObjInvAffine;
		if (affine(M)) flags |= acr::dev::kObjFwdAffine;
		if (s->obj_world_min && s->obj_world_max && margin > 0.0f && affine(M) && affine(MI)) {
			// the skip is only valid where the reference's local ray is the true image of the world ray:
			// rotation times uniform scale (the direction goes through transpose(mat3(M)), scene.cu:366-367,413)
			double g[3][3], tr = 0;
			bool ok = true;
			for (int a = 0; a < 3; a++)
				for (int b = 0; b < 3; b++) {
					g[a][b] = 0;
					for (int k = 0; k < 3; k++) g[a][b] += double(M[4 * a + k]) * double(M[4 * b + k]);
				}
			tr = (g[0][0] + g[1][1] + g[2][2]) / 3.0;
			ok = std::isfinite(tr) && tr > 0;
			for (int a = 0; a < 3 && ok; a++)
				for (int b = 0; b < 3 && ok; b++)
					ok = std::fabs(g[a][b] - (a == b ? tr : 0.0)) <= 1e-4 * tr;
			float mn[3], mx[3];
			for (int k = 0; k < 3 && ok; k++) {
				mn[k] = s->obj_world_min[3 * size_t(i) + k] - margin;
				mx[k] = s->obj_world_max[3 * size_t(i) + k] + margin;
				ok = std::isfinite(mn[k]) && std::isfinite(mx[k]) && mn[k] < mx[k];
			}
			if (ok) {
				flags |= acr::dev::kObjPrefilter;
				for (int k = 0; k < 3; k++) { w[k] = fbits(mn[k]); w[4 + k] = fbits(mx[k]); }
			}
		}
		w[3] = flags;
		// the object's first mesh (0 for an object without meshes: a valid record number when there are meshes at all,
		// and its record is then read without being used)
		w[7] = s->obj_mesh_count[i] ? s->obj_mesh_ids[s->obj_mesh_begin[i]] : 0u;
		memcpy(w + 8, MI, 64);
		memcpy(w + 24, s->obj_inv_normal_transform + 9 * size_t(i), 36);
		w[33] = s->obj_mesh_begin[i];
		w[34] = s->obj_mesh_count[i];
		memcpy(w + 36, M, 64);
		memcpy(w + 52, s->obj_normal_transform + 9 * size_t(i), 36);
	}
	if (s->num_obj_mesh_ids) memcpy(&blob[h.off_objmesh], s->obj_mesh_ids, 4 * size_t(s->num_obj_mesh_ids));
	for (uint32_t m = 0; m < s->num_meshes; m++) {
		uint4 *q = &blob[h.off_meshes + m * acr::dev::kMeshSlots];
		q[0] = make_uint4(fbits(s->mesh_min[3 * m]), fbits(s->mesh_min[3 * m + 1]), fbits(s->mesh_min[3 * m + 2]), s->mesh_material[m]);
		q[1] = make_uint4(fbits(s->mesh_max[3 * m]), fbits(s->mesh_max[3 * m + 1]), fbits(s->mesh_max[3 * m + 2]), meshNodeAt[m]);
		q[2] = make_uint4(s->mesh_tri_begin[m], meshLevels[m], 0, 0);
	}
	for (uint32_t i = 0; i < s->num_materials; i++) {
		uint4 *q = &blob[h.off_materials + i * acr::dev::kMaterialSlots];
		q[0] = make_uint4(fbits(s->mat_ambient[3 * i]), fbits(s->mat_ambient[3 * i + 1]), fbits(s->mat_ambient[3 * i + 2]), fbits(s->mat_refractive_index[i]));
		q[1] = make_uint4(fbits(s->mat_diffuse[3 * i]), fbits(s->mat_diffuse[3 * i + 1]), fbits(s->mat_diffuse[3 * i + 2]), 0);
		q[2] = make_uint4(fbits(s->mat_specular[3 * i]), fbits(s->mat_specular[3 * i + 1]), fbits(s->mat_specular[3 * i + 2]), 0);
	}
	for (uint32_t i = 0; i < s->num_lights; i++) {
		uint4 *q = &blob[h.off_lights + i * acr::dev::kLightSlots];
		const float *pos = s->light_position + 3 * i, *dir = s->light_direction + 3 * i, *dif = s->light_diffuse + 3 * i;
		const float *att = s->light_attenuation + 3 * i, *cone = s->light_cone + 2 * i;
		q[0] = make_uint4(fbits(pos[0]), fbits(pos[1]), fbits(pos[2]), uint32_t(s->light_type[i]));
		q[1] = make_uint4(fbits(dir[0]), fbits(dir[1]), fbits(dir[2]), fbits(att[0]));
		q[2] = make_uint4(fbits(dif[0]), fbits(dif[1]), fbits(dif[2]), fbits(att[1]));
		q[3] = make_uint4(fbits(att[2]), fbits(cone[0]), fbits(cone[1]), 0);
	}
	for (uint32_t m = 0; m < s->num_meshes; m++) {
		const uint32_t nb = s->mesh_node_begin[m], count = (1u << meshLevels[m]) - 1;
		for (uint32_t i = 0; i < count; i++)
			putNode(blob[h.off_mnodes + meshNodeAt[m] + i], s->mnode_left + nb, s->mnode_right + nb, s->mnode_start + nb, s->mnode_end + nb,
			        s->mnode_flags + nb, i, count);
	}
	for (uint32_t i = 0; i < s->num_triangles; i++) {
		const float *a = s->tri_p0 + 3 * size_t(i), *b = s->tri_p1 + 3 * size_t(i), *c = s->tri_p2 + 3 * size_t(i);
		// e1 = b - a, e2 = c - a: the two subtractions of math.h:132-133, done once here
		const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
		const float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
		uint4 *q = &blob[h.off_tris + size_t(i) * acr::dev::kTriSlots];
		q[0] = make_uint4(fbits(a[0]), fbits(a[1]), fbits(a[2]), fbits(e1[0]));
		q[1] = make_uint4(fbits(e1[1]), fbits(e1[2]), fbits(e2[0]), fbits(e2[1]));
		q[2] = make_uint4(fbits(e2[2]), s->tri_prim_id ? s->tri_prim_id[i] : i, 0, 0);
		const float *n0 = s->tri_n0 + 3 * size_t(i), *n1 = s->tri_n1 + 3 * size_t(i), *n2 = s->tri_n2 + 3 * size_t(i);
		uint4 *n = &blob[h.off_normals + size_t(i) * acr::dev::kNormalSlots];
		n[0] = make_uint4(fbits(n0[0]), fbits(n0[1]), fbits(n0[2]), fbits(n1[0]));
		n[1] = make_uint4(fbits(n1[1]), fbits(n1[2]), fbits(n2[0]), fbits(n2[1]));
		n[2] = make_uint4(fbits(n2[2]), 0, 0, 0);
	}

	// what a primary ray parallel to a coordinate plane must not start on (see acr_hip_ctx::worldBounds)
	ctx->parallelOk = false;
	ctx->parallelBitsValid = false;
	ctx->worldBounds = acr_hip_ctx::AxisValues();
	ctx->localSpaces.clear();
	if (intervalOk && s->num_objects <= acr::dev::kParallelMaxObjects) {
		auto planes = [](acr_hip_ctx::AxisValues &out, const float *left, const float *right, const uint8_t *flags, uint32_t count) {
			for (uint32_t i = 0; i < count; i++) {
				if (flags[i] & ACR_NODE_LEAF) continue;
				const int a = flags[i] & ACR_NODE_AXIS_MASK;
				if (a > 2) continue;
				out.v[a].push_back(left[i]);
				out.v[a].push_back(right[i]);
			}
		};
		for (int k = 0; k < 3; k++) {
			ctx->worldBounds.v[k].push_back(s->scene_min[k]);
			ctx->worldBounds.v[k].push_back(s->scene_max[k]);
		}
		planes(ctx->worldBounds, s->snode_left, s->snode_right, s->snode_flags, snodeCount);
		ctx->localSpaces.resize(s->num_objects);
		for (uint32_t i = 0; i < s->num_objects; i++) {
			const uint32_t *w = reinterpret_cast<const uint32_t *>(&blob[h.off_objects + i * acr::dev::kObjectSlots]);
			if (w[3] & acr::dev::kObjPrefilter)
				for (int k = 0; k < 3; k++) {
					float lo, hi;
					memcpy(&lo, w + k, 4);
					memcpy(&hi, w + 4 + k, 4);
					ctx->worldBounds.v[k].push_back(lo);
					ctx->worldBounds.v[k].push_back(hi);
				}
			acr_hip_ctx::LocalSpace &ls = ctx->localSpaces[i];
			memcpy(ls.invM, w + 8, 64);
			ls.affine = (w[3] & acr::dev::kObjInvAffine) != 0;
			for (uint32_t k = 0; k < s->obj_mesh_count[i]; k++) {
				const uint32_t m = s->obj_mesh_ids[s->obj_mesh_begin[i] + k];
				for (int a = 0; a < 3; a++) {
					ls.bounds.v[a].push_back(s->mesh_min[3 * size_t(m) + a]);
					ls.bounds.v[a].push_back(s->mesh_max[3 * size_t(m) + a]);
				}
				const uint32_t nb = s->mesh_node_begin[m];
				planes(ls.bounds, s->mnode_left + nb, s->mnode_right + nb, s->mnode_flags + nb, (1u << meshLevels[m]) - 1);
			}
		}
		// sorted, each value once: the per-camera look-up is a binary search whatever the scene's size
		auto tidy = [](acr_hip_ctx::AxisValues &b) {
			for (std::vector<float> &v : b.v) {
				std::sort(v.begin(), v.end());
				v.erase(std::unique(v.begin(), v.end()), v.end());
				v.shrink_to_fit();
			}
		};
		tidy(ctx->worldBounds);
		for (acr_hip_ctx::LocalSpace &ls : ctx->localSpaces) tidy(ls.bounds);
		ctx->parallelOk = true;
	}

	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (ctx->blob) {
		ACR_HIP_TRY(ctx, hipFree(ctx->blob));
		ctx->blob = nullptr;
	}
	ctx->haveScene = false;
	ctx->blobBytes = blob.size() * sizeof(uint4);
	ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->blob), ctx->blobBytes));
	ACR_HIP_TRY(ctx, hipMemcpy(ctx->blob, blob.data(), ctx->blobBytes, hipMemcpyHostToDevice));
	ctx->header = h;
	ctx->intervalOk = intervalOk;
	ctx->leafOrderOk = leafOrderOk;
	ctx->soloScene = s->num_objects == 1 && s->obj_mesh_count[0] == 1 && sceneLevels == 1;
	{
		const char *e = getenv("ACR_SOLO");
		ctx->soloOff = e && e[0] == '0';
	}
	ctx->sceneDepthGiven = s->scene_depth;
	ctx->haveScene = true;
	return ACR_OK;
}

int acr_hip_set_camera(acr_hip_ctx *ctx, const acr_camera *camera)
{
	if (!ctx || !camera) return fail(ctx, ACR_ERR_INVALID, "null argument");
	// a moved camera keeps the block order (neighbouring views cost alike) but has it re-measured soon
	if (!ctx->haveCamera || memcmp(&ctx->camera, camera, sizeof(*camera)) != 0) ctx->blockOrderInterval = kOrderIntervalAfterMove;
	ctx->camera = *camera;
	ctx->haveCamera = true;
	ctx->parallelBitsValid = false;
	return ACR_OK;
}

int acr_hip_get_camera(const acr_hip_ctx *ctx, acr_camera *camera)
{
	if (!ctx || !camera || !ctx->haveCamera) return ACR_ERR_INVALID;
	*camera = ctx->camera;
	return ACR_OK;
}

int acr_hip_resize(acr_hip_ctx *ctx, uint32_t width, uint32_t height)
{
	if (!ctx || width == 0 || height == 0 || uint64_t(width) * height >= (1ull << 31)) return fail(ctx, ACR_ERR_INVALID, "bad frame size");
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	(void)hipFree(ctx->screen); ctx->screen = nullptr;
	(void)hipFree(ctx->primid); ctx->primid = nullptr;
	(void)hipFree(ctx->hitT); ctx->hitT = nullptr;
	(void)hipFree(ctx->pixelValues); ctx->pixelValues = nullptr;
	(void)hipFree(ctx->pathKeys); ctx->pathKeys = nullptr;
	(void)hipFree(ctx->pathVotes); ctx->pathVotes = nullptr;
	ctx->width = ctx->height = 0;
	const size_t n = size_t(width) * height;
	ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->screen), n * sizeof(float4)));
	ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->primid), n * sizeof(uint32_t)));
	ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->hitT), n * sizeof(float)));
	ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->pixelValues), n * sizeof(int32_t)));
	ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->pathKeys), n * sizeof(uint64_t)));
	ACR_HIP_TRY(ctx, acr_dev_launch_iota(ctx->pixelValues, uint32_t(n), ctx->stream));
	ACR_HIP_TRY(ctx, hipMemsetAsync(ctx->pathKeys, 0, n * sizeof(uint64_t), ctx->stream));
	ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ACR_HIP_TRY(ctx, hipMemset(ctx->screen, 0, n * sizeof(float4)));
	ctx->width = width;
	ctx->height = height;
	return ACR_OK;
}

} // extern "C"

// staticOnly: the launch may use nothing but what exists already -- the context's buffers, the caller's, and
// this device's copy of `scratch`; whatever would need an allocation is refused.
static int renderImpl(acr_hip_ctx *ctx, const acr_frame *f, void *screen_dev, uint32_t *primid_dev, float *hit_t_dev, bool staticOnly)
{
	if (!ctx || !f) return fail(ctx, ACR_ERR_INVALID, "null argument");
	if (!ctx->haveScene || !ctx->haveCamera) return fail(ctx, ACR_ERR_NO_SCENE, "render before scene and camera were set");
	if (f->width == 0 || f->height == 0 || uint64_t(f->width) * f->height >= (1ull << 31)) return fail(ctx, ACR_ERR_INVALID, "bad frame size");
	if (f->max_bounces > ACR_MAX_BOUNCES) return fail(ctx, ACR_ERR_UNSUPPORTED, "max_bounces %u > %d", f->max_bounces, ACR_MAX_BOUNCES);
	const bool tiled = f->num_tiles != 0;
	const bool ownBuffers = screen_dev == nullptr;
	if (staticOnly && (tiled || f->tile_rays_dev || f->tile_cycles_dev))
		return fail(ctx, ACR_ERR_UNSUPPORTED, "acr_hip_render_static renders whole frames only (a tile list needs a device copy)");
	if (ownBuffers && (tiled || ctx->width != f->width || ctx->height != f->height))
		return fail(ctx, ACR_ERR_INVALID, "context-owned framebuffer needs acr_hip_resize(width,height) and a full-frame render");
	uint32_t tilesX = 0, grid = 0;
	if (tiled) {
		if (!f->tile_ids || f->tile_w == 0 || f->tile_h == 0 || (f->tile_w & 15u) || (f->tile_h & 15u))
			return fail(ctx, ACR_ERR_INVALID, "tile list needs tile_ids and tile sizes that are multiples of 16");
		if (f->pixel_values_dev) return fail(ctx, ACR_ERR_INVALID, "a tile list cannot be combined with a pixel permutation");
		tilesX = (f->width + f->tile_w - 1) / f->tile_w;
		const uint32_t tilesY = (f->height + f->tile_h - 1) / f->tile_h;
		for (uint32_t i = 0; i < f->num_tiles; i++)
			if (f->tile_ids[i] >= tilesX * tilesY) return fail(ctx, ACR_ERR_INVALID, "tile_ids[%u] out of range", i);
		const uint64_t g = uint64_t(f->num_tiles) * (f->tile_w / 16) * (f->tile_h / 16) * kGroupsPerBlock;
		if (g >= (1ull << 31)) return fail(ctx, ACR_ERR_INVALID, "too many tiles");
		grid = uint32_t(g);
		if (f->path_keys_dev) return fail(ctx, ACR_ERR_INVALID, "traversal-path keys need a full-frame launch");
	} else {
		if (f->tile_rays_dev || f->tile_cycles_dev) return fail(ctx, ACR_ERR_INVALID, "per-tile costs need a tile list");
		// A traversal takes 2 * MAX_SIZE bits of the Path's length (bih.h:293-297): from MAX_DEPTH 6 on that is more
		// than the 64 bits of the key, so only the primary ray's traversal reaches it, which is what the kernels record.
		if (f->path_keys_dev && ctx->sceneDepthGiven < 6)
			return fail(ctx, ACR_ERR_UNSUPPORTED, "traversal-path keys need a scene BIH of MAX_DEPTH >= 6 (got %u)", ctx->sceneDepthGiven);
		grid = ((f->width + 15) / 16) * ((f->height + kBlockRows - 1) / kBlockRows) * kGroupsPerBlock;
	}

	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (tiled) {
		const int rc = syncTileList(ctx, ctx->renderTiles, f->tile_ids, f->num_tiles);
		if (rc != ACR_OK) return rc;
	}

	RenderParams p{};
	p.scene = ctx->header;
	p.blob = ctx->blob;
	p.cam_aspect = float(f->width) / float(f->height); // renderer.cu:179
	p.cam_hfov = ctx->camera.horizontal_fov;
	memcpy(p.cam_position, ctx->camera.position, 12);
	memcpy(p.cam_forward, ctx->camera.forward, 12);
	memcpy(p.cam_up, ctx->camera.up, 12);
	{
		// renderer.cu:197-200,360-361 with the operations of csrc/math/math.h (this file is built with -ffp-contract=off)
		using namespace acr::math;
		const vec3 dir(p.cam_forward[0], p.cam_forward[1], p.cam_forward[2]);
		const vec3 up(p.cam_up[0], p.cam_up[1], p.cam_up[2]);
		const vec3 cR = cross(dir, up);
		const vec3 cU = cross(cR, dir);
		p.cam_cr[0] = cR.x; p.cam_cr[1] = cR.y; p.cam_cr[2] = cR.z;
		p.cam_cu[0] = cU.x; p.cam_cu[1] = cU.y; p.cam_cu[2] = cU.z;
		p.cam_dist = fastertanfull(p.cam_hfov / 2.0f);
		p.inv_width = 1.0f / int(f->width);
		p.inv_height = 1.0f / int(f->height);
	}
	if (ctx->parallelOk && !ctx->parallelBitsValid) {
		using namespace acr::math;
		auto clear = [](const acr_hip_ctx::AxisValues &b, const vec3 &o) {
			uint32_t bits = 0;
			for (int a = 0; a < 3; a++) {
				bool hit;
				// the values are sorted (upload): equal to o[a] iff neither is less than the other; +0 and -0 count as equal, as with ==
				hit = std::binary_search(b.v[a].begin(), b.v[a].end(), o[a]);
				if (!hit) bits |= 1u << a;
			}
			return bits;
		};
		const vec3 cam(ctx->camera.position[0], ctx->camera.position[1], ctx->camera.position[2]);
		memset(ctx->parallelBits, 0, sizeof(ctx->parallelBits));
		ctx->parallelBits[acr::dev::kParallelWords - 1] = clear(ctx->worldBounds, cam);
		for (size_t i = 0; i < ctx->localSpaces.size(); i++) {
			const acr_hip_ctx::LocalSpace &ls = ctx->localSpaces[i];
			mat4 m;
			for (int j = 0; j < 4; j++) m.c[j] = vec4(ls.invM[4 * j], ls.invM[4 * j + 1], ls.invM[4 * j + 2], ls.invM[4 * j + 3]);
			const vec3 lo = ls.affine ? translateAffine(m, cam) : translate(m, cam); // as objectIntersect forms it
			ctx->parallelBits[i >> 3] |= clear(ls.bounds, lo) << (4 * (i & 7));
		}
		ctx->parallelBitsValid = true;
	}
	if (ctx->parallelOk) memcpy(p.parallel_ok, ctx->parallelBits, sizeof(p.parallel_ok));
	p.width = f->width;
	p.height = f->height;
	p.blocks_x = (f->width + 15) / 16;
	p.seed = f->seed;
	p.frames = f->frames;
	p.max_bounces = f->max_bounces;
	p.tile_w = f->tile_w;
	p.tile_h = f->tile_h;
	p.tiles_x = tilesX;
	p.num_tiles = f->num_tiles;
	p.tiles_into_frame = (tiled && f->tiles_into_frame) ? 1u : 0u;
	p.tile_ids = tiled ? ctx->renderTiles.dev() : nullptr;
	p.pixel_values = f->pixel_values_dev;
	p.path_keys = f->path_keys_dev;
	// Launch order of the blocks: the caller's table, else the context's own (measured every blockOrderEvery-th
	// launch of a configuration; small launches do not fill the GPU and are left alone).
	if (tiled && f->block_order_dev) return fail(ctx, ACR_ERR_INVALID, "block_order_dev needs a full-frame launch");
	const uint32_t numBlocks = grid / kGroupsPerBlock;
	// the context's own order lives in tables it allocates: not for the launch that may allocate nothing
	const bool ownOrder = !staticOnly && ctx->blockOrderEvery != 0 && !f->block_order_dev && numBlocks >= kMinBlocksForOrder;
	bool measureBlocks = false;
	p.block_order = f->block_order_dev;
	if (ownOrder) {
		// FNV-1a over what decides which pixels a block holds
		uint64_t key = 1469598103934665603ull;
		auto mix = [&key](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
		mix(f->width); mix(f->height); mix(f->num_tiles); mix(f->tile_w); mix(f->tile_h);
		for (uint32_t i = 0; i < f->num_tiles; i++) mix(f->tile_ids[i]);
		// a configuration is measured from its second launch in a row on: a caller that changes its tile list
		// with every launch (per-frame rebalancing across GPUs) would pay for tables it never uses
		const bool repeated = key == ctx->blockKey;
		if (!repeated) {
			ctx->blockKey = key;
			ctx->blockOrderValid = false;
		}
		if (ctx->blockCap < numBlocks) {
			ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			for (uint32_t **b : {&ctx->blockCost, &ctx->blockOrder, &ctx->blockZ}) {
				(void)hipFree(*b);
				*b = nullptr;
			}
			ctx->blockCap = 0;
			ctx->blockZCount = 0;
			for (uint32_t **b : {&ctx->blockCost, &ctx->blockOrder, &ctx->blockZ})
				ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(b), size_t(numBlocks) * sizeof(uint32_t)));
			ctx->blockCap = numBlocks;
			ctx->blockOrderValid = false;
			ctx->blockCostZero = false;
		}
		const uint32_t interval = ctx->blockOrderInterval < ctx->blockOrderEvery ? ctx->blockOrderInterval : ctx->blockOrderEvery;
		measureBlocks = repeated && (!ctx->blockOrderValid || ctx->blockOrderAge + 1 >= interval);
		if (ctx->blockOrderValid) p.block_order = ctx->blockOrder;
		if (measureBlocks) {
			if (!ctx->blockCostZero) ACR_HIP_TRY(ctx, hipMemsetAsync(ctx->blockCost, 0, size_t(ctx->blockCap) * sizeof(uint32_t), ctx->stream));
			ctx->blockCostZero = false; // until the sort behind this launch has run
			p.block_cost = ctx->blockCost;
		}
	}
	p.screen = ownBuffers ? ctx->screen : static_cast<float4 *>(screen_dev);
	p.primid = primid_dev ? primid_dev : (ownBuffers ? ctx->primid : nullptr);
	p.hit_t = hit_t_dev ? hit_t_dev : (ownBuffers ? ctx->hitT : nullptr);
	const bool countRays = f->collect_stats == ACR_STATS_FRAME || f->collect_stats == ACR_STATS_ACCUMULATE;
	p.counters = countRays ? ctx->counters : nullptr;
	p.tile_rays = f->tile_rays_dev;
	p.tile_cycles = f->tile_cycles_dev;
	if (p.tile_rays) ACR_HIP_TRY(ctx, hipMemsetAsync(p.tile_rays, 0, sizeof(uint32_t) * f->num_tiles, ctx->stream));
	if (p.tile_cycles) ACR_HIP_TRY(ctx, hipMemsetAsync(p.tile_cycles, 0, sizeof(uint32_t) * f->num_tiles, ctx->stream));

	// Frames with bounces on scenes that fit LDS are rendered by the wave-cooperative kernel (object tests as tasks
	// compacted across the wavefront), if its tables fit as well.
	const bool wantWave = ctx->intervalOk && ctx->traversalMode == ACR_TRAVERSAL_AUTO && f->max_bounces > 1 && !f->path_keys_dev &&
	                      uint64_t(grid) * kBlock < (1ull << 32); // that kernel numbers its pixel slots in 32 bits
	// One deep mesh read from HBM and no bounces: the kernel whose lanes share their walks through the mesh tree
	// (if the scene turns out not to fit LDS whole, which the plan says).
	const bool wantSolo = ctx->intervalOk && ctx->leafOrderOk && ctx->traversalMode == ACR_TRAVERSAL_AUTO && ctx->soloScene && !ctx->soloOff &&
	                      f->max_bounces >= 1 && !f->path_keys_dev;
	// (the variant of that kernel for frames without bounces is compiled for four waves per SIMD, and its columns then
	// get a quarter of a CU's LDS, the levels above them global memory: not for the launch that may allocate nothing)
	// Several objects whose trees are read from HBM: the same kernel, every lane walking the tree over objects for
	// itself and the wave taking the objects asked for one after the other.
	const bool wantMulti = !wantSolo && ctx->intervalOk && ctx->leafOrderOk && ctx->traversalMode == ACR_TRAVERSAL_AUTO && ctx->header.num_objects >= 1 &&
	                       !ctx->soloOff && f->max_bounces >= 1 && !f->path_keys_dev;
	const LaunchPlan plan = planLds(ctx->header, f->max_bounces > 1 ? f->max_bounces - 1 : 0, true, wantWave, wantSolo && f->max_bounces == 1 && !staticOnly, wantMulti);
	p.lds_stack_slot = plan.stackSlot;
	p.lds_cstack_off = plan.cstackOff;
	p.lds_wave_off = plan.waveOff ? plan.waveOff : plan.wordsOff;
	// the interval form walks mesh trees that are not LDS-resident near side first, which needs leafOrderOk
	const bool fast = ctx->intervalOk && ctx->traversalMode != ACR_TRAVERSAL_SIX_PLANE && (plan.ldsMode == 2 || ctx->leafOrderOk);
	const bool listOnly = fast && ctx->traversalMode == ACR_TRAVERSAL_LIST_PASS;
	const bool waveKernel = fast && plan.waveOff != 0;
	const bool soloKernel = fast && wantSolo && plan.ldsMode != 2;
	const bool multiKernel = fast && wantMulti && !listOnly && plan.ldsMode != 2;
	if (listOnly && (f->path_keys_dev || f->tile_cycles_dev))
		return fail(ctx, ACR_ERR_UNSUPPORTED, "ACR_TRAVERSAL_LIST_PASS renders frames only (no path keys, no tile clock ticks)");
	if (staticOnly && fast && plan.meshGlobLevels)
		return fail(ctx, ACR_ERR_UNSUPPORTED, "acr_hip_render_static: this scene's mesh trees are %u levels deeper than the waiting-child column "
		            "kept in LDS, and the part in global memory is allocated per launch size", plan.meshGlobLevels);
	if (staticOnly && fast && size_t(grid) * kBlock > size_t(ACR_STATIC_MAX_SLOTS))
		return fail(ctx, ACR_ERR_UNSUPPORTED, "acr_hip_render_static: %zu pixel slots, the static list holds %u (frames up to 1920 x 1088)",
		            size_t(grid) * kBlock, unsigned(ACR_STATIC_MAX_SLOTS));
	if (fast && plan.meshGlobLevels) {
		// one column per work-item of the larger of the two launches that use it (the list pass has its own fixed grid)
		const size_t slots = size_t(grid > acr::dev::kRedoGrid ? grid : acr::dev::kRedoGrid) * kBlock;
		if (slots >= (1ull << 32)) return fail(ctx, ACR_ERR_UNSUPPORTED, "launch too large for a mesh tree this deep");
		const size_t need = slots * plan.meshGlobLevels * sizeof(float2);
		if (ctx->meshGlobBytes < need) {
			ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			(void)hipFree(ctx->meshGlob);
			ctx->meshGlob = nullptr;
			ctx->meshGlobBytes = 0;
			ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->meshGlob), need));
			ctx->meshGlobBytes = need;
		}
		p.mesh_glob = ctx->meshGlob;
		p.mesh_glob_levels = plan.meshGlobLevels;
		p.mesh_glob_stride = uint32_t(slots);
	}
	if (fast && staticOnly) {
		// Every call on this GPU shares the array and finds what the last one left: both counters start at zero
		// here instead of relying on the previous list pass, which may have been another context's.
		StaticScratch *sc = ctx->staticScratch;
		ACR_HIP_TRY(ctx, hipMemsetAsync(sc->redoCount, 0, 2 * sizeof(uint32_t), ctx->stream));
		p.redo_list = sc->redoList;
		p.redo_count = sc->redoCount;
		p.redo_count_next = sc->redoCount + 1;
	} else if (fast) {
		const size_t slots = size_t(grid) * kBlock;
		if (ctx->redoCap < slots) {
			ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			(void)hipFree(ctx->redoList);
			ctx->redoList = nullptr;
			ctx->redoCap = 0;
			ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->redoList), slots * sizeof(uint32_t)));
			ctx->redoCap = slots;
		}
		p.redo_list = ctx->redoList;
		p.redo_count = ctx->redoCount + ctx->redoParity;
		p.redo_count_next = ctx->redoCount + (ctx->redoParity ^ 1u);
	}

	ctx->timed = f->collect_stats != 0;
	ctx->counted = countRays;
	if (ctx->timed) {
		if (f->collect_stats == ACR_STATS_FRAME)
			ACR_HIP_TRY(ctx, hipMemsetAsync(ctx->counters, 0, kCounterSlots * 4 * sizeof(unsigned long long), ctx->stream));
		ACR_HIP_TRY(ctx, hipEventRecord(ctx->evStart, ctx->stream));
	}
	if (listOnly) {
		// every slot of the launch goes onto the list; only the list pass runs
		const uint32_t slots = grid * kBlock;
		ACR_HIP_TRY(ctx, acr_dev_launch_iota(reinterpret_cast<int32_t *>(p.redo_list), slots, ctx->stream));
		ctx->listAllCount = slots; // the copy below may read it after this call returned
		ACR_HIP_TRY(ctx, hipMemcpyAsync(p.redo_count, &ctx->listAllCount, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		p.block_order = nullptr;
		p.block_cost = nullptr;
		measureBlocks = false;
	}
	ACR_HIP_TRY(ctx, acr_dev_launch_render(&p, plan.ldsMode, listOnly ? 2 : (waveKernel ? 3 : (soloKernel ? 4 : (multiKernel ? 5 : (fast ? 1 : 0)))), grid, plan.ldsBytes, ctx->stream));
	// the list pass that was just queued zeroes the other counter: only now does the next frame switch to it
	if (fast && !staticOnly) ctx->redoParity ^= 1u;
	if (tiled) {
		const int rc = tileListRead(ctx, ctx->renderTiles);
		if (rc != ACR_OK) return rc;
	}
	ctx->statsStatic = staticOnly;
	if (ctx->timed) ACR_HIP_TRY(ctx, hipEventRecord(ctx->evStop, ctx->stream));
	const uint32_t orderUsed = f->block_order_dev ? ACR_BLOCK_ORDER_CALLER : (p.block_order ? ACR_BLOCK_ORDER_MEASURED : ACR_BLOCK_ORDER_NATURAL);
	if (ownOrder) {
		if (measureBlocks) {
			// most expensive first; the launch that was just queued has read the old table by the time this runs
			// Scenes whose trees and triangles are read through the L2 get the XCD-aware table: the blocks one XCD
			// renders are neighbours in the frame (10 M-triangle soup: 2.55 -> 2.46 ms per frame; a scene resident in
			// LDS has nothing in the L2 to share and loses balance by it: Cornell 1080p 0.151 -> 0.170 ms).
			// A list of full-width bands of 16 rows (the multi-GPU split) numbers its blocks row by row like a frame
			// of its own; other tile lists keep the plain order.
			const uint32_t bx = !tiled ? (f->width + 15) / 16 : ((f->tile_h == 16 && f->tile_w >= f->width) ? f->tile_w >> 4 : 0u);
			if (bx != 0 && plan.ldsMode != 2 && ctx->xcds == acr::dev::kXcdsAssumed) {
				if (ctx->blockZCount != numBlocks || ctx->blockZBlocksX != bx) {
					auto spread = [](uint32_t v) { // bit k of v to bit 2k
						uint64_t r = 0;
						for (uint32_t k = 0; k < 32; k++) r |= uint64_t((v >> k) & 1u) << (2 * k);
						return r;
					};
					// the host copy below may still be the source of the last upload: a new frame size is rare enough to wait
					ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
					std::vector<std::pair<uint64_t, uint32_t>> keyed(numBlocks);
					for (uint32_t b = 0; b < numBlocks; b++) keyed[b] = {spread(b % bx) | (spread(b / bx) << 1), b};
					std::sort(keyed.begin(), keyed.end());
					ctx->blockZHost.resize(numBlocks);
					for (uint32_t b = 0; b < numBlocks; b++) ctx->blockZHost[b] = keyed[b].second;
					ACR_HIP_TRY(ctx, hipMemcpyAsync(ctx->blockZ, ctx->blockZHost.data(), size_t(numBlocks) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
					ctx->blockZCount = numBlocks;
					ctx->blockZBlocksX = bx;
				}
				ACR_HIP_TRY(ctx, acr_dev_launch_order_blocks_regions(ctx->blockCost, numBlocks, ctx->blockOrder, ctx->blockZ, ctx->stream));
			} else {
				ACR_HIP_TRY(ctx, acr_dev_launch_order_blocks(ctx->blockCost, numBlocks, ctx->blockOrder, ctx->stream));
			}
			ctx->blockCostZero = true;
			ctx->blockOrderValid = true;
			ctx->blockOrderAge = 0;
			if (ctx->blockOrderInterval < (1u << 30)) ctx->blockOrderInterval *= 2;
		} else {
			ctx->blockOrderAge++;
		}
	}
	ctx->stats = acr_stats{};
	ctx->stats.block_order = orderUsed;
	ctx->stats.lds_bytes = uint32_t(plan.ldsBytes);
	ctx->stats.scene_in_lds = uint32_t(plan.ldsMode);
	ctx->stats.grid_blocks = grid;
	ctx->stats.interval_traversal = fast ? 1u : 0u;
	ctx->stats.near_first = (fast && !listOnly && plan.ldsMode != 2) ? 1u : 0u;
	ctx->stats.wave_tasks = waveKernel ? 1u : 0u;
	ctx->stats.shared_walk = (soloKernel || multiKernel) ? 1u : 0u;
	ctx->lastSlots = tiled ? size_t(f->num_tiles) * f->tile_w * f->tile_h : size_t(f->width) * f->height;
	return ACR_OK;
}

extern "C" {

int acr_hip_render(acr_hip_ctx *ctx, const acr_frame *f, void *screen_dev, uint32_t *primid_dev, float *hit_t_dev)
{
	return renderImpl(ctx, f, screen_dev, primid_dev, hit_t_dev, false);
}

int acr_hip_render_static(acr_hip_ctx *ctx, const acr_frame *f, void *screen_dev, uint32_t *primid_dev, float *hit_t_dev)
{
	return renderImpl(ctx, f, screen_dev, primid_dev, hit_t_dev, true);
}

int acr_hip_trace_rays(acr_hip_ctx *ctx, uint32_t n, const float *origins, const float *dirs, const float *tmax,
                       uint8_t *hit, float *t, uint32_t *prim, float *position, float *normal)
{
	if (!ctx || !origins || !dirs) return fail(ctx, ACR_ERR_INVALID, "null argument");
	if (!ctx->haveScene) return fail(ctx, ACR_ERR_NO_SCENE, "trace before scene upload");
	if (n == 0) return ACR_OK;
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	const size_t v3 = size_t(n) * 12;
	char *base = nullptr;
	// one allocation: origins, dirs, tmax | t, prim, position, normal, hit
	const size_t total = v3 * 2 + size_t(n) * 4 + size_t(n) * 4 + size_t(n) * 4 + v3 * 2 + n;
	ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&base), total));
	float *dO = reinterpret_cast<float *>(base);
	float *dD = reinterpret_cast<float *>(base + v3);
	float *dTmax = reinterpret_cast<float *>(base + 2 * v3);
	float *dT = dTmax + n;
	uint32_t *dPrim = reinterpret_cast<uint32_t *>(dT + n);
	float *dPos = reinterpret_cast<float *>(dPrim + n);
	float *dNrm = reinterpret_cast<float *>(reinterpret_cast<char *>(dPos) + v3);
	uint8_t *dHit = reinterpret_cast<uint8_t *>(reinterpret_cast<char *>(dNrm) + v3);
	int rc = ACR_OK;
	hipError_t e = hipMemcpyAsync(dO, origins, v3, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dD, dirs, v3, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess && tmax) e = hipMemcpyAsync(dTmax, tmax, size_t(n) * 4, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) {
		TraceParams p{};
		p.scene = ctx->header;
		p.blob = ctx->blob;
		p.n = n;
		p.origins = dO; p.dirs = dD; p.tmax = tmax ? dTmax : nullptr;
		p.hit = dHit; p.t = dT; p.prim = dPrim; p.position = dPos; p.normal = dNrm;
		const LaunchPlan plan = planLds(ctx->header, 0, false);
		p.lds_stack_slot = plan.stackSlot;
		e = acr_dev_launch_trace(&p, plan.ldsMode, (ctx->intervalOk && ctx->traversalMode == ACR_TRAVERSAL_AUTO && (plan.ldsMode == 2 || ctx->leafOrderOk)) ? 1 : 0, (n + kBlock - 1) / kBlock, plan.ldsBytes, ctx->stream);
	}
	if (e == hipSuccess && hit) e = hipMemcpyAsync(hit, dHit, n, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess && t) e = hipMemcpyAsync(t, dT, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess && prim) e = hipMemcpyAsync(prim, dPrim, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess && position) e = hipMemcpyAsync(position, dPos, v3, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess && normal) e = hipMemcpyAsync(normal, dNrm, v3, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) rc = fail(ctx, ACR_ERR_HIP, "trace_rays: %s", hipGetErrorString(e));
	(void)hipFree(base);
	return rc;
}

int acr_hip_read_framebuffer(acr_hip_ctx *ctx, float *rgba, size_t num_floats)
{
	if (!ctx || !rgba || !ctx->screen) return fail(ctx, ACR_ERR_INVALID, "no context-owned framebuffer");
	if (num_floats != size_t(ctx->width) * ctx->height * 4) return fail(ctx, ACR_ERR_INVALID, "size mismatch");
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, hipMemcpyAsync(rgba, ctx->screen, num_floats * 4, hipMemcpyDeviceToHost, ctx->stream));
	ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return ACR_OK;
}

int acr_hip_read_primid(acr_hip_ctx *ctx, uint32_t *prim, size_t count)
{
	if (!ctx || !prim || !ctx->primid) return fail(ctx, ACR_ERR_INVALID, "no context-owned primitive-id buffer");
	if (count != size_t(ctx->width) * ctx->height) return fail(ctx, ACR_ERR_INVALID, "size mismatch");
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, hipMemcpyAsync(prim, ctx->primid, count * 4, hipMemcpyDeviceToHost, ctx->stream));
	ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return ACR_OK;
}

int acr_hip_read_hit_t(acr_hip_ctx *ctx, float *t, size_t count)
{
	if (!ctx || !t || !ctx->hitT) return fail(ctx, ACR_ERR_INVALID, "no context-owned hit-distance buffer");
	if (count != size_t(ctx->width) * ctx->height) return fail(ctx, ACR_ERR_INVALID, "size mismatch");
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, hipMemcpyAsync(t, ctx->hitT, count * 4, hipMemcpyDeviceToHost, ctx->stream));
	ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return ACR_OK;
}

int acr_hip_read_stats(acr_hip_ctx *ctx, acr_stats *stats)
{
	if (!ctx || !stats) return fail(ctx, ACR_ERR_INVALID, "null argument");
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (ctx->timed) {
		if (ctx->counted) {
			static_assert(kCounterSlots * 4 * sizeof(unsigned long long) <= 65536, "read into a stack buffer");
			unsigned long long slots[kCounterSlots * 4];
			ACR_HIP_TRY(ctx, hipMemcpy(slots, ctx->counters, sizeof(slots), hipMemcpyDeviceToHost));
			unsigned long long c[4] = {0, 0, 0, 0};
			for (uint32_t k = 0; k < kCounterSlots * 4; k++) c[k & 3] += slots[k];
			ctx->stats.rays_primary = c[0];
			ctx->stats.rays_shadow = c[1];
			ctx->stats.rays_bounce = c[2];
			ctx->stats.primary_hits = c[3];
		}
		float ms = 0;
		ACR_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop));
		ctx->stats.kernel_ms = ms;
		if (ctx->stats.interval_traversal) {
			// the counter of the last frame is the one the next frame will NOT use; it is zeroed one frame later
			uint32_t n = 0;
			const uint32_t *counter = ctx->statsStatic ? ctx->staticScratch->redoCount : ctx->redoCount + (ctx->redoParity ^ 1u);
			ACR_HIP_TRY(ctx, hipMemcpy(&n, counter, sizeof(n), hipMemcpyDeviceToHost));
			ctx->stats.six_plane_pixels = n;
		}
	}
	*stats = ctx->stats;
	return ACR_OK;
}

void *acr_hip_framebuffer(acr_hip_ctx *ctx) { return ctx ? ctx->screen : nullptr; }
int32_t *acr_hip_pixel_map(acr_hip_ctx *ctx) { return ctx ? ctx->pixelValues : nullptr; }
uint64_t *acr_hip_path_keys(acr_hip_ctx *ctx) { return ctx ? ctx->pathKeys : nullptr; }

int32_t *acr_hip_path_votes(acr_hip_ctx *ctx)
{
	if (!ctx || !ctx->width) return nullptr;
	if (!ctx->pathVotes) {
		const size_t bytes = size_t(64) * ctx->width * ctx->height * sizeof(int32_t);
		if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&ctx->pathVotes), bytes) != hipSuccess) {
			ctx->pathVotes = nullptr;
			fail(ctx, ACR_ERR_HIP, "no memory for %zu bytes of path votes", bytes);
			return nullptr;
		}
		if (hipMemsetAsync(ctx->pathVotes, 0, bytes, ctx->stream) != hipSuccess) {
			// the header promises a zeroed table: a retry must allocate and zero again, not find this one
			(void)hipFree(ctx->pathVotes);
			ctx->pathVotes = nullptr;
			fail(ctx, ACR_ERR_HIP, "could not zero %zu bytes of path votes", bytes);
			return nullptr;
		}
	}
	return ctx->pathVotes;
}

int acr_hip_vote_paths(acr_hip_ctx *ctx, const uint64_t *path_keys_dev, int32_t *votes_dev, uint32_t num_pixels, int reset)
{
	if (!ctx || !path_keys_dev || !votes_dev) return fail(ctx, ACR_ERR_INVALID, "null argument");
	if (num_pixels == 0) return ACR_OK;
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, acr_dev_launch_vote_paths(path_keys_dev, votes_dev, num_pixels, reset, ctx->stream));
	return ACR_OK;
}

int acr_hip_vote_keys(acr_hip_ctx *ctx, const int32_t *votes_dev, uint64_t *path_keys_dev, uint32_t num_pixels)
{
	if (!ctx || !path_keys_dev || !votes_dev) return fail(ctx, ACR_ERR_INVALID, "null argument");
	if (num_pixels == 0) return ACR_OK;
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ACR_HIP_TRY(ctx, acr_dev_launch_vote_keys(votes_dev, path_keys_dev, num_pixels, ctx->stream));
	return ACR_OK;
}

int acr_hip_sort_pixel_map(acr_hip_ctx *ctx, const uint64_t *path_keys_dev, int32_t *pixel_values_dev, uint32_t num_pixels)
{
	if (!ctx || !path_keys_dev || !pixel_values_dev) return fail(ctx, ACR_ERR_INVALID, "null argument");
	if (num_pixels == 0) return ACR_OK;
	if (num_pixels >= (1u << 31)) return fail(ctx, ACR_ERR_INVALID, "too many pixels");
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	size_t need = 0;
	ACR_HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, need, path_keys_dev, ctx->sortKeysOut, ctx->sortValuesIn, pixel_values_dev, int(num_pixels),
	                                                    0, 64, ctx->stream));
	if (ctx->sortCap < num_pixels || ctx->sortTempBytes < need) {
		ACR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		(void)hipFree(ctx->sortTemp); ctx->sortTemp = nullptr;
		(void)hipFree(ctx->sortKeysOut); ctx->sortKeysOut = nullptr;
		(void)hipFree(ctx->sortValuesIn); ctx->sortValuesIn = nullptr;
		ctx->sortCap = 0;
		ctx->sortTempBytes = 0;
		ACR_HIP_TRY(ctx, hipMalloc(&ctx->sortTemp, need));
		ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sortKeysOut), size_t(num_pixels) * sizeof(uint64_t)));
		ACR_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sortValuesIn), size_t(num_pixels) * sizeof(int32_t)));
		ctx->sortCap = num_pixels;
		ctx->sortTempBytes = need;
	}
	// thrust::sequence(pixelValues) (renderer.cu:137), then stable_sort_by_key(pixelKeys, pixelValues) (renderer.cu:460):
	// an LSD radix sort is stable
	ACR_HIP_TRY(ctx, acr_dev_launch_iota(ctx->sortValuesIn, num_pixels, ctx->stream));
	size_t bytes = ctx->sortTempBytes;
	ACR_HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->sortTemp, bytes, path_keys_dev, ctx->sortKeysOut, ctx->sortValuesIn, pixel_values_dev,
	                                                    int(num_pixels), 0, 64, ctx->stream));
	return ACR_OK;
}

int acr_hip_scatter_tiles(acr_hip_ctx *ctx, const void *tiles_dev, const uint32_t *tile_ids, uint32_t num_tiles,
                          uint32_t tile_w, uint32_t tile_h, uint32_t width, uint32_t height, void *frame_dev)
{
	if (!ctx || !tiles_dev || !tile_ids || !frame_dev || tile_w == 0 || tile_h == 0 || width == 0 || height == 0)
		return fail(ctx, ACR_ERR_INVALID, "null or zero argument");
	if (num_tiles == 0) return ACR_OK;
	const uint32_t tilesX = (width + tile_w - 1) / tile_w, tilesY = (height + tile_h - 1) / tile_h;
	for (uint32_t i = 0; i < num_tiles; i++)
		if (tile_ids[i] >= tilesX * tilesY) return fail(ctx, ACR_ERR_INVALID, "tile_ids[%u] out of range", i);
	ACR_HIP_TRY(ctx, hipSetDevice(ctx->device));
	const int rc = syncTileList(ctx, ctx->scatterTiles, tile_ids, num_tiles);
	if (rc != ACR_OK) return rc;
	ACR_HIP_TRY(ctx, acr_dev_launch_scatter_tiles(tiles_dev, ctx->scatterTiles.dev(), num_tiles, tile_w, tile_h, width, height, frame_dev, ctx->stream));
	return tileListRead(ctx, ctx->scatterTiles);
}

} // extern "C"
