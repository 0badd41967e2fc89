// Render kernels of the MI355X (gfx950, wave64) renderer core.
//
// One work-item per pixel slot; 256-thread workgroups of four waves, each wave
// an 8x8 pixel square so that the lanes of a wave traverse alike.  The scene
// blob (device_scene.h) is staged into LDS when it fits, so that after the
// staging copy the only HBM traffic of a frame is the framebuffer.
//
// Everything here follows the reference's arithmetic operation for operation
// (citations are file:line relative to the reference root); what differs is
// the machinery around it:
//   * BIH traversal keeps no box stack (the reference pushes whole clipped
//     boxes, bih.h:214-215,256,262).  Two forms of the same traversal:
//       - interval form (the default): a lane carries max(near) / min(far) of
//         its current box, two floats, and the right child that waits while the
//         left subtree is walked is two floats in a per-level LDS column; see
//         the comment above `struct Interval` for why every decision is the
//         reference's;
//       - six-plane form: the slab test compare for compare (geometry.h:44-76)
//         on the six distances (bound - o) * invD kept in registers, one of
//         them replaced per descent and logged in an LDS "undo" column.  It
//         serves the pixels whose rays the interval form cannot (list pass),
//         and whole frames on request.
//     A bitmask remembers at which levels the right child passed its test and
//     is still to be visited (left before right, deepest pending first, leaf
//     items ascending; bih.h:253-288).
//   * rayColor's template recursion (renderer.cu:232-324) is a loop; the
//     per-level direct terms are kept in an LDS column and summed from the
//     deepest level outwards, the order the recursion adds them.
//   * Shadow rays stop at the first object that proves occlusion; see
//     sceneIntersect for why that cannot change the result.
//   * Optionally the per-pixel traversal-path key of the reference's adaptive
//     assignment (bih.h:13-27,268-274) is recorded (KEYS).
//   * Mesh trees that are read from HBM (the deep ones) are walked near side
//     first, with exact ties and order-dependent cases settled the reference's
//     way; see "Near-side-first traversal" below.
//   * Which block of 16 x 16 pixels a workgroup renders comes from a table
//     (RenderParams::block_order), so that the host can start the expensive
//     blocks first; renderKernel also measures the time per block for it.
//   * renderKernel has a variant for max_bounces == 1 (ONE), the counterpart
//     of the reference's compile-time MAX_BOUNCES (renderer.cu:10).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math).
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>

#include "../math/math.h"
#include "../math/rng.h"
#include "device_scene.h"
#include "render_params.h"

using namespace acr::math;
using acr::dev::SceneHeader;

namespace acr
{
namespace dev
{

constexpr int kBlock = ACR_BLOCK;          // work-items per workgroup
constexpr uint32_t kSlots = ACR_BLOCK_SLOTS; // pixel slots per block of the launch grid: 16 x 16 pixels, kSlots / kBlock workgroups
static_assert(ACR_BLOCK_SLOTS % ACR_BLOCK == 0 && ACR_BLOCK % 64 == 0, "a block of pixel slots is a whole number of workgroups of whole waves");
// resident waves per SIMD the render kernel is compiled for (sets its VGPR budget: 512 / waves)
#ifndef ACR_RENDER_WAVES
#define ACR_RENDER_WAVES 4
#endif
// the same for scenes whose trees and triangles are read from HBM: three, four and five waves run equally fast there
// (the length of the slowest waves bounds those frames), and three leave the kernel enough registers to spill nothing
// and a workgroup enough LDS for every level of a deep tree's waiting-child column
#ifndef ACR_RENDER_WAVES_HBM
#define ACR_RENDER_WAVES_HBM 3
#endif

// waves per SIMD the wave-cooperative kernel is compiled for
#ifndef ACR_WAVE_KERNEL_WAVES
#define ACR_WAVE_KERNEL_WAVES 4
#endif

// triangle records of an HBM-resident leaf fetched per memory round trip
#ifndef ACR_TRIS_PER_TRIP
#define ACR_TRIS_PER_TRIP 5
#endif

extern __shared__ uint4 g_lds[];

// The launch parameters, read again from the kernel argument segment at the place of the call.  A field the kernel
// reads at its start and needs only at its end otherwise sits in a scalar register through every loop; there are more
// such fields than registers, and the compiler parks them in lanes of a vector register (a vector instruction per
// move).  What is read through this reference is fetched by scalar loads right here instead.
// This lane's number in its wave, computed where it is asked for (the same number kept from a kernel's start to its
// end costs a register through every loop, or a spill).
__device__ __forceinline__ uint32_t laneNow()
{
	uint32_t l;
	asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
	return l;
}

typedef const __attribute__((address_space(4))) RenderParams LateParams;
__device__ __forceinline__ LateParams &lateParams()
{
	const uint64_t a = reinterpret_cast<uint64_t>(__builtin_amdgcn_kernarg_segment_ptr());
	uint32_t lo = uint32_t(a), hi = uint32_t(a >> 32);
	// through vector registers, so that this also works where the lanes of the wave have gone separate ways
	asm volatile("" : "+v"(lo), "+v"(hi));
	lo = __builtin_amdgcn_readfirstlane(lo);
	hi = __builtin_amdgcn_readfirstlane(hi);
	return *reinterpret_cast<LateParams *>(uint64_t(lo) | (uint64_t(hi) << 32));
}

// The rare way into the interval form for a PRIMARY ray that SlabArgs::initInterval turned away: 1 / d by division,
// and every component either finite or infinite on an axis where the launch vouches that no bound the ray can meet
// equals its origin's coordinate (RenderParams::parallel_ok; `space`: an object number, or kParallelMaxObjects for
// the tree over objects).  Then no 0 * infinity arises, the infinite slab distances compare as the reference's do, and
// the interval form's decisions are the six-plane form's.  Defined after SlabArgs.
struct SlabArgs;
__device__ __forceinline__ bool parallelRayQualifies(SlabArgs &sa, const struct Ray &r, uint32_t space);

__device__ __forceinline__ float asF(uint32_t u) { return __uint_as_float(u); }

// Scene memory view.  SMALL/LARGE say whether that part of the blob was
// staged into LDS at the same slot offsets.  NEAR_FIRST: the interval form walks mesh trees near side first
// (see bihTraverseNear); the default is to do so exactly when the trees live in HBM, i.e. when they are deep.
// OBJ_STRIDE: slots from one object's record to the next.  kObjectSlots in the blob; the wave-cooperative kernel stages
// the records kObjectSlots + 1 apart (stageSceneSpread), so that lanes reading the same word of different objects'
// records, which is what its object tests do, do not all meet in the same LDS banks (records 256 bytes apart do).
template <bool SMALL_LDS, bool LARGE_LDS, bool NEAR_FIRST = !LARGE_LDS, uint32_t OBJ_STRIDE = kObjectSlots>
struct SceneMem
{
	static constexpr bool kLargeInLds = LARGE_LDS;
	static constexpr bool kNearFirst = NEAR_FIRST;
	static constexpr uint32_t kObjStride = OBJ_STRIDE;
	const uint4 *__restrict__ g;
	__device__ __forceinline__ uint4 small(uint32_t slot) const
	{
		if constexpr (SMALL_LDS) return g_lds[slot];
		else return g[slot];
	}
	__device__ __forceinline__ uint4 large(uint32_t slot) const
	{
		if constexpr (LARGE_LDS) return g_lds[slot];
		else return g[slot];
	}
	// N consecutive slots.  From global memory that is one address and N constant offsets (slot + k formed in 32 bits
	// would need an address computation of its own for every k); the blob is followed by kBlobPadSlots slots, so a run
	// that starts inside it may end behind it.
	template <int N>
	__device__ __forceinline__ void largeRun(uint32_t slot, uint4 (&out)[N]) const
	{
		if constexpr (LARGE_LDS) {
#pragma unroll
			for (int k = 0; k < N; k++) out[k] = g_lds[slot + k];
		} else {
			const uint4 *__restrict__ q = g + size_t(slot);
#pragma unroll
			for (int k = 0; k < N; k++) out[k] = q[k];
		}
	}
};

struct Ray { vec3 o, d; };

// Rays that hit nothing whatever a traversal does with them, so that the interval kernels need not hand their pixels to
// the six-plane pass (whose answer is the same, and slow to get on a deep tree):
//  * a direction with a NaN component (randomHemi and refract produce them from degenerate normals): the local direction
//    normalize(M d) is NaN in every component for any matrix, cross(d, e2) and the determinant are NaN, no test of
//    myIntersectRayTriangle (math.h:123-146) is true for a NaN, so it returns true with t = NaN, and `t < info.t`
//    (geometry.h:123-146) is false: no triangle of no object is ever taken;
//  * the zero vector (+0 or -0 in every component), in scenes that vouch for it: SceneHeader::zero_dir_misses.
// Asked where such rays can come from: the continuation rays of rayColor (renderer.cu:262-311) and a caller's own rays
// (traceKernel); primary rays and shadow rays are normalised differences of points.
__device__ __forceinline__ bool hitsNothing(const SceneHeader &sc, const Ray &r)
{
	if (r.d.x != r.d.x || r.d.y != r.d.y || r.d.z != r.d.z)
		return true;
	return sc.zero_dir_misses && r.d.x == 0.0f && r.d.y == 0.0f && r.d.z == 0.0f;
}

// Per-ray constants of the slab test: geometry.h:78-86 / bih.h:223-228
struct SlabArgs
{
	vec3 o, invD;
	__device__ __forceinline__ void init(const Ray &r)
	{
		o = r.o;
		invD = 1.0f / r.d;
	}
	// The interval form's init: 1 / d without the division (acr::math::reciprocalNewton), and whether the ray
	// qualifies for that form.  A component of 1 / d that comes out as a normal number is the quotient, bit for bit;
	// one that does not (a zero, denormal, huge, infinite or NaN direction component) disqualifies the ray, as an
	// infinite 1 / d always did: its pixel is rendered by the six-plane pass, which divides.
	__device__ __forceinline__ bool initInterval(const Ray &r)
	{
		o = r.o;
		invD = vec3(acr::math::reciprocalNewton(r.d.x), acr::math::reciprocalNewton(r.d.y), acr::math::reciprocalNewton(r.d.z));
		const float inf = __builtin_inff();
		return __builtin_isnormal(invD.x) && __builtin_isnormal(invD.y) && __builtin_isnormal(invD.z) && fabsf(o.x) < inf && fabsf(o.y) < inf &&
		       fabsf(o.z) < inf;
	}
	// the sign array of geometry.h:84-86, compared when needed instead of kept in three more registers
	__device__ __forceinline__ bool sx() const { return invD.x < 0; }
	__device__ __forceinline__ bool sy() const { return invD.y < 0; }
	__device__ __forceinline__ bool sz() const { return invD.z < 0; }
	__device__ __forceinline__ bool sign(uint32_t axis) const { return invD[int(axis)] < 0; }
	// (bound - o) * invD on one axis: the per-plane term of geometry.h:52-66
	__device__ __forceinline__ float plane(uint32_t axis, float bound) const { return (bound - o[int(axis)]) * invD[int(axis)]; }
};

__device__ __forceinline__ bool parallelRayQualifies(SlabArgs &sa, const Ray &r, uint32_t space)
{
	sa.init(r);
	LateParams &q = lateParams();
	uint32_t ok = 0;
	if (space < kParallelMaxObjects) ok = (q.parallel_ok[space >> 3] >> (4u * (space & 7u))) & 7u;
	else if (space == kParallelMaxObjects) ok = q.parallel_ok[kParallelWords - 1] & 7u;
	const float inf = __builtin_inff();
	bool good = fabsf(sa.o.x) < inf && fabsf(sa.o.y) < inf && fabsf(sa.o.z) < inf;
#pragma unroll
	for (int a = 0; a < 3; a++) {
		const float v = fabsf(sa.invD[a]);
		good = good && (v < inf || (v == inf && ((ok >> a) & 1u) != 0));
	}
	return good;
}

// Slab distances of a box, near plane / far plane per axis:
// near.x = (bounds[sign.x].x - o.x) * invD.x, far.x = (bounds[1-sign.x].x - o.x) * invD.x
struct SlabBox
{
	vec3 nearT, farT;
	__device__ __forceinline__ void init(const SlabArgs &a, const vec3 &mn, const vec3 &mx)
	{
		const vec3 tmn = (mn - a.o) * a.invD;
		const vec3 tmx = (mx - a.o) * a.invD;
		nearT = vec3(a.sx() ? tmx.x : tmn.x, a.sy() ? tmx.y : tmn.y, a.sz() ? tmx.z : tmn.z);
		farT = vec3(a.sx() ? tmn.x : tmx.x, a.sy() ? tmn.y : tmx.y, a.sz() ? tmn.z : tmx.z);
	}
};

// geometry.h:50-75 on precomputed plane distances; compare-and-select exactly
// as written so that NaNs (0 * inf) take the same branches.
__device__ __forceinline__ bool slabTest(const vec3 &nearT, const vec3 &farT, float tcur)
{
	float tmin = nearT.x, tmax = farT.x;
	const float tymin = nearT.y, tymax = farT.y;
	if ((tmin > tymax) || (tymin > tmax))
		return false;
	if (tymin > tmin)
		tmin = tymin;
	if (tymax < tmax)
		tmax = tymax;
	const float tzmin = nearT.z, tzmax = farT.z;
	if ((tmin > tzmax) || (tzmin > tmax))
		return false;
	if (tzmin > tmin)
		tmin = tzmin;
	if (tzmax < tmax)
		tmax = tzmax;
	return tmin < tcur && tmax > epsilon();
}

// Path of renderer.cu:341 (bih.h:13-27) as designed: two bits per internal node visited, appended from the
// most significant end; what no longer fits is dropped.
struct PathRec
{
	uint64_t path;
	uint32_t length;
	// bih.h:269-274: 00 neither child hit, 11 both, 10 the child with more items, 01 the other one
	__device__ __forceinline__ void append(bool hitL, bool hitR, bool lBigger)
	{
		length += 2;
		if (length <= 64) {
			const uint64_t flag = lBigger ? ((uint64_t(hitL) << 1) | uint64_t(hitR)) : ((uint64_t(hitR) << 1) | uint64_t(hitL));
			path |= flag << (64 - length);
		}
	}
};

__device__ __forceinline__ vec3 withComp(const vec3 &v, uint32_t axis, float f)
{
	return vec3(axis == 0 ? f : v.x, axis == 1 ? f : v.y, axis == 2 ? f : v.z);
}

// bih.h:209-299.  loadNode(i) returns node i's slot; leaf(start, end) tests the
// leaf's items against the ray and returns true to abandon the traversal
// (used by shadow rays only); tcur() is the current info.t the slab tests
// compare against (bih.h:247-248 pass `info`).  `undo` points at this lane's
// column: undo[level * kBlock].
template <bool KEYS = false, class LoadNode, class Leaf, class TCur>
__device__ __forceinline__ void bihTraverse(LoadNode loadNode, uint32_t maxDepth, const SlabArgs &sa, const SlabBox &root,
                                            float *undo, Leaf leaf, TCur tcur, PathRec *path = nullptr)
{
	uint32_t idx = 0, level = 0, pending = 0;
	uint64_t axes = 0; // 2 bits per level: split axis of the path node at that level
	vec3 nearT = root.nearT, farT = root.farT;

	for (;;) {
		const uint4 n = loadNode(idx);
		// a node on the last level has no children in the heap (bih.h:131-134)
		const bool isLeaf = (n.w & kNodeLeafBit) != 0 || level + 1 >= maxDepth;
		if (!isLeaf) {
			const uint32_t axis = (n.w >> kNodeAxisShift) & 3u;
			const bool sg = sa.sign(axis);
			// lBB.max[axis] = left, rBB.min[axis] = right (bih.h:244-245)
			const float tl = sa.plane(axis, asF(n.x));
			const float tr = sa.plane(axis, asF(n.y));
			// max is the far plane when the direction is positive, else the near plane
			const vec3 lNear = sg ? withComp(nearT, axis, tl) : nearT;
			const vec3 lFar = sg ? farT : withComp(farT, axis, tl);
			const vec3 rNear = sg ? nearT : withComp(nearT, axis, tr);
			const vec3 rFar = sg ? withComp(farT, axis, tr) : farT;
			const float tc = tcur();
			const bool hitL = slabTest(lNear, lFar, tc);
			const bool hitR = slabTest(rNear, rFar, tc);
			if constexpr (KEYS) path->append(hitL, hitR, (n.w & kNodeLeftBiggerBit) != 0);
			if (hitL || hitR) {
				// left first; a right child that also hit waits (bih.h:253-264)
				if (hitL && hitR) pending |= 1u << level;
				const bool replacedNear = hitL ? sg : !sg;
				undo[level * kBlock] = replacedNear ? nearT[int(axis)] : farT[int(axis)];
				axes = (axes & ~(3ull << (2 * level))) | (uint64_t(axis) << (2 * level));
				nearT = hitL ? lNear : rNear;
				farT = hitL ? lFar : rFar;
				idx = 2 * idx + (hitL ? 1 : 2);
				level++;
				continue;
			}
		} else {
			const uint32_t start = n.z, end = n.w & kNodeEndMask;
			if (leaf(start, end))
				return;
		}
		// pop: deepest level whose right child is still pending
		if (pending == 0)
			return;
		const uint32_t l = 31u - uint32_t(__clz(int(pending)));
		pending &= ~(1u << l);
		while (level > l) {
			const bool wasLeft = (idx & 1u) != 0;
			idx = (idx - 1) >> 1;
			level--;
			const uint32_t a = uint32_t(axes >> (2 * level)) & 3u;
			const bool sg = sa.sign(a);
			const float v = undo[level * kBlock];
			const bool replacedNear = wasLeft ? sg : !sg;
			if (replacedNear) nearT = withComp(nearT, a, v);
			else farT = withComp(farT, a, v);
		}
		{
			const uint4 pn = loadNode(idx);
			const uint32_t a = (pn.w >> kNodeAxisShift) & 3u;
			const bool sg = sa.sign(a);
			const float tr = sa.plane(a, asF(pn.y));
			// min is the near plane when the direction is positive
			undo[level * kBlock] = sg ? farT[int(a)] : nearT[int(a)];
			if (sg) farT = withComp(farT, a, tr);
			else nearT = withComp(nearT, a, tr);
			idx = 2 * idx + 2;
			level++;
		}
	}
}


// ---- interval form of the same traversal -----------------------------------
//
// When every component of the ray origin and of 1/d is finite and every box
// handed to the slab test has min <= max on each axis, no plane distance
// (bound - o) * invD can be NaN and near[i] <= far[i] holds on each axis
// (subtraction and multiplication by a non-negative value round
// monotonically).  geometry.h:50-75 then rejects exactly when some near[i]
// exceeds some far[j], i.e. it accepts iff
//     max(near) <= min(far)  &&  max(near) < info.t  &&  min(far) > eps.
// A BIH child box differs from its parent's in ONE plane, and that plane lies
// inside the parent's box (the host checks this for every reachable node at
// upload), so the child's max(near) / min(far) are max(parent's, plane) /
// min(parent's, plane): a lane carries two floats per box instead of six, and
// the right child that waits while the left subtree is walked is remembered
// as its two floats in an LDS column indexed by tree level.  Decisions are
// identical to the six-plane form, so visit order and results are too.
// Rays that do not qualify (a zero or denormal direction component, i.e. an
// infinite 1/d) make the pixel leave this kernel; the six-plane kernel
// renders it afterwards (renderListKernel).

struct Interval
{
	float tmin, tmax;
};
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// [max(near), min(far)] of a box with min <= max, for a finite ray
__device__ __forceinline__ Interval boxInterval(const SlabArgs &a, const vec3 &mn, const vec3 &mx)
{
	const vec3 tmn = (mn - a.o) * a.invD;
	const vec3 tmx = (mx - a.o) * a.invD;
	// The near plane of an axis is the min-side one for a positive direction, the max-side one for a negative one
	// (geometry.h:52-66).  With min <= max and nothing NaN, multiplying by invD keeps or reverses the order of the
	// two distances accordingly, so near = the smaller of the two and far = the larger: same values, without the
	// three sign tests and six selects.
	Interval r;
	r.tmin = fmax3(__builtin_fminf(tmn.x, tmx.x), __builtin_fminf(tmn.y, tmx.y), __builtin_fminf(tmn.z, tmx.z));
	r.tmax = fmin3(__builtin_fmaxf(tmn.x, tmx.x), __builtin_fmaxf(tmn.y, tmx.y), __builtin_fmaxf(tmn.z, tmx.z));
	return r;
}
__device__ __forceinline__ bool intervalHit(float tmin, float tmax, float tcur) { return tmin <= tmax && tmin < tcur && tmax > epsilon(); }

// The waiting-child column of one tree and lane: one float2 per tree level.  The deepest `ldsLevels` levels,
// where a traversal spends nearly all of its pushes and pops, live in LDS (entry for level l at
// lds[(l - globLevels) * kBlock]); the levels above them, which a ray touches at most once each, live in a
// global-memory column (glob[l * globStride], neighbouring lanes in neighbouring addresses).  globLevels is 0
// for every tree shallow enough for LDS, which is all trees but the deep single-mesh ones.
struct LevelColumn
{
	float2 *lds;
	float2 *glob;
	uint32_t globLevels, globStride;
	__device__ __forceinline__ void store(uint32_t level, float2 v) const
	{
		if (level >= globLevels) {
			lds[(level - globLevels) * kBlock] = v;
		} else {
			glob[size_t(level) * globStride] = v;
			// keeps the two stores apart: merged into one store through a selected pointer they become a flat store,
			// which waits on both the LDS and the vector memory counter
			asm volatile("");
		}
	}
	__device__ __forceinline__ float2 load(uint32_t level) const
	{
		float2 v;
		if (level >= globLevels) {
			v = lds[(level - globLevels) * kBlock];
		} else {
			v = glob[size_t(level) * globStride];
			asm volatile("" : "+v"(v.x), "+v"(v.y)); // two loads, not one flat load through a selected pointer (see store)
		}
		return v;
	}
};

// bih.h:209-299 on intervals.  `stack` is this lane's waiting-child column.
// Loop shape: an inner loop that only descends (a lane leaves it at a leaf, or when its traversal has
// nothing left), then the leaf's items, then the pop.  Lanes of a wave therefore do their node steps
// together and their item tests together instead of interleaving the two.
template <bool KEYS = false, class LoadNode, class Leaf, class TCur>
__device__ __forceinline__ void bihTraverseFast(LoadNode loadNode, const SlabArgs &sa, Interval box, const LevelColumn &stack, Leaf leaf, TCur tcur,
                                                PathRec *path = nullptr)
{
	uint32_t idx = 0, level = 0, pending = 0;
	float tminB = box.tmin, tmaxB = box.tmax;
	for (;;) {
		uint4 n = loadNode(idx);
		bool alive = true;
		while ((n.w & kNodeLeafBit) == 0) {
			// both children are requested before this node's arithmetic: whichever the walk takes is at hand when
			// the decision is made (an internal node has both; they are neighbours in memory)
			const uint4 childL = loadNode(2 * idx + 1), childR = loadNode(2 * idx + 2);
			const uint32_t axis = (n.w >> kNodeAxisShift) & 3u;
			const float oa = sa.o[int(axis)], ia = sa.invD[int(axis)];
			const bool sg = ia < 0;
			const float tl = (asF(n.x) - oa) * ia; // lBB.max[axis] = left  (bih.h:244)
			const float tr = (asF(n.y) - oa) * ia; // rBB.min[axis] = right (bih.h:245)
			// The child on the ray's near side keeps the parent's tmin and is cut at its far plane;
			// the other keeps the parent's tmax and is cut at its near plane.  Left is the near-side
			// child for a positive direction.
			const float tCutFar = sg ? tr : tl, tCutNear = sg ? tl : tr;
			const float nearSideMax = tCutFar < tmaxB ? tCutFar : tmaxB;
			const float farSideMin = tCutNear > tminB ? tCutNear : tminB;
			const float tc = tcur();
			const bool hitNearSide = intervalHit(tminB, nearSideMax, tc);
			const bool hitFarSide = intervalHit(farSideMin, tmaxB, tc);
			const bool hitL = sg ? hitFarSide : hitNearSide;
			const bool hitR = sg ? hitNearSide : hitFarSide;
			if constexpr (KEYS) path->append(hitL, hitR, (n.w & kNodeLeftBiggerBit) != 0);
			if (hitL || hitR) {
				// left first; a right child that also hit waits (bih.h:253-264)
				const bool takeFarSide = hitL ? sg : !sg;
				if (hitL && hitR) {
					pending |= 1u << level;
					stack.store(level, takeFarSide ? make_float2(tminB, nearSideMax) : make_float2(farSideMin, tmaxB));
				}
				if (takeFarSide) tminB = farSideMin;
				else tmaxB = nearSideMax;
				idx = 2 * idx + (hitL ? 1 : 2);
				level++;
				n = hitL ? childL : childR;
			} else {
				// neither child: resume at the deepest level whose right child is still waiting
				if (pending == 0) {
					alive = false;
					break;
				}
				const uint32_t l = 31u - uint32_t(__clz(int(pending)));
				pending &= ~(1u << l);
				const float2 b = stack.load(l);
				tminB = b.x;
				tmaxB = b.y;
				idx = 2 * (((idx + 1) >> (level - l)) - 1) + 2; // right child of this path's node on level l
				level = l + 1;
				n = loadNode(idx);
			}
		}
		if (!alive)
			return;
		if (leaf(n.z, n.w & kNodeEndMask))
			return;
		if (pending == 0)
			return;
		const uint32_t l = 31u - uint32_t(__clz(int(pending)));
		pending &= ~(1u << l);
		const float2 b = stack.load(l);
		tminB = b.x;
		tmaxB = b.y;
		idx = 2 * (((idx + 1) >> (level - l)) - 1) + 2;
		level = l + 1;
	}
}

#ifdef ACR_WALK_DEBUG
// experiment builds only (make variant NAME=walkdbg DEFS=-DACR_WALK_DEBUG): how full the wave is in each part of the
// near-side-first walk.  Sums over all walks of a run: [0] walks, [1] lane passes through the descent loop's body (two
// node steps each), [2] the same counted once per wave (each lane adds 2^20 / lanes present), [3] [4] likewise for the
// trips of the leaf loop (ACR_TRIS_PER_TRIP records each), [5] triangle tests made, [6] [7] likewise for the rounds
// (descent, leaf, resumption).  Lanes present / 64 in a part = [1] / (64 * [2] / 2^20) and so on.
__device__ unsigned long long g_walkDebug[16];
__device__ __forceinline__ void walkDebugAdd(unsigned long long *acc, int lanePart, int wavePart)
{
	acc[lanePart] += 1ull;
	acc[wavePart] += (1ull << 20) / (unsigned long long)__popcll(__ballot(true));
}
#endif

// Near-side-first traversal, for the deep trees of scenes that live in HBM.
//
// bih.h:253-264 always walks the left child first, whatever the ray's direction; a deep tree then spends most of
// its node visits and item tests behind the hit it ends up returning.  Visiting the child on the ray's near side
// first, and dropping a waiting child once the current hit distance lies in front of it, ends with the nearest
// hit W of the tree.  The reference's order ends with W too unless it drops a box b on the way to W's leaf:
// that takes an earlier-found hit X with tmin(b) >= t_X (the drop, geometry.h:75) and t_X > t_W (else W would
// not be the nearest), hence tmin(b) > t_W -- the box would have to start behind a hit it contains, which only
// rounding can produce.  tmin never decreases along a path, so tmin(W's leaf) >= tmin(b), and these are the
// same fp32 values here and there.  So: if no other hit X has t_W < t_X <= tmin(W's leaf), both orders
// return W.  The
// traversal reports the nearest hit that lost (`second`) and W's leaf tmin; the caller hands the pixel to the
// reference-order pass when second <= tmin(W's leaf), which takes a double coincidence within rounding.  An
// exact tie (common: an ulp of t is ~1e-6 of the scene) needs no such pass: the reference gives it to the hit it
// finds first, and that is the one with the lower item number.
// For X to be seen at all, boxes are only dropped when they start more than kNearPruneSlack behind the current
// hit (far above the rounding that can put a box's tmin behind a hit inside it).
constexpr float kNearPruneSlack = 1.0f + 1.0f / 32768.0f;

// Resume at the deepest waiting child that can still hold a hit as near as `tc`; the waiting child is the
// sibling of the child this path took on that level.
__device__ __forceinline__ bool popNear(const LevelColumn &stack, uint32_t &pending, uint32_t &idx, uint32_t &level, float &tminB, float &tmaxB, float tc)
{
	while (pending != 0) {
		const uint32_t l = 31u - uint32_t(__clz(int(pending)));
		pending &= ~(1u << l);
		const float2 b = stack.load(l);
		if (b.x < tc) {
			tminB = b.x;
			tmaxB = b.y;
			idx = (((idx + 1) >> (level - l - 1)) ^ 1u) - 1u;
			level = l + 1;
			return true;
		}
	}
	return false;
}

// bihTraverseFast visiting the child on the ray's near side first; `tcur` returns the slackened hit distance and
// `leaf` also gets the leaf box's tmin.
//
// These trees live in HBM and a ray's walk is a chain of dependent reads, one memory round trip after the other;
// what a frame of such a scene takes is the length of its longest chains.  The heap order makes every descendant's
// address a function of the node number alone, so a step requests the node's two children AND its four
// grandchildren (two runs of consecutive records)
// before it does the node's arithmetic, and then takes two levels for one round trip (three, with the eight
// great-grandchildren on top, measured slower: 3.25 against 3.03 ms on the 10 M-triangle soup; so did requesting, while a
// leaf's triangles are on their way, the subtree the walk will most likely resume at: 3.13 ms).  nodeCount: records of this tree (requests beyond the last one are moved onto it; what
// they return is never used).
// loadRun(i, out): the records i, i + 1, ... of the tree into out[].
// The walk is made of three pieces, so that a kernel whose lanes walk together (soloWalk) can look at the wave between
// them: nearDescend (down to a leaf), the leaf's items, popNear (up to the next waiting child).

// Where a near-side-first walk stands: node, its level, the levels with a waiting child, the ray's interval in the node's box.
struct NearWalk
{
	uint32_t idx, level, pending;
	float tminB, tmaxB;
};

// One node step (bih.h:241-264 on intervals, near side first): 0 / 1 = the walk goes on with the left / right child,
// 2 = neither child can hold a hit nearer than tc.
__device__ __forceinline__ uint32_t nearNodeStep(const uint4 &n, const SlabArgs &sa, const LevelColumn &stack, float tc, NearWalk &w)
{
	const uint32_t axis = (n.w >> kNodeAxisShift) & 3u;
	const float oa = sa.o[int(axis)], ia = sa.invD[int(axis)];
	const bool sg = ia < 0;
	const float tl = (asF(n.x) - oa) * ia;
	const float tr = (asF(n.y) - oa) * ia;
	const float tCutFar = sg ? tr : tl, tCutNear = sg ? tl : tr;
	const float nearSideMax = tCutFar < w.tmaxB ? tCutFar : w.tmaxB;
	const float farSideMin = tCutNear > w.tminB ? tCutNear : w.tminB;
	const bool hitNearSide = intervalHit(w.tminB, nearSideMax, tc);
	const bool hitFarSide = intervalHit(farSideMin, w.tmaxB, tc);
	if (!(hitNearSide || hitFarSide))
		return 2u;
	if (hitNearSide && hitFarSide) {
		w.pending |= 1u << w.level;
		stack.store(w.level, make_float2(farSideMin, w.tmaxB));
	}
	if (hitNearSide) w.tmaxB = nearSideMax;
	else w.tminB = farSideMin;
	const bool goLeft = hitNearSide != sg; // the near-side child is the left one for a positive direction
	w.idx = 2 * w.idx + (goLeft ? 1 : 2);
	w.level++;
	return goLeft ? 0u : 1u;
}

// From node w.idx (nothing of its subtree at hand) down to a leaf, two levels per memory round trip: 0 = w.idx is a
// leaf (its record in n), 1 = neither child of the node reached can hold a nearer hit, 2 = MAX_PASSES round trips made
// and still on the way down (only with MAX_PASSES != 0).
// The two children are neighbours in memory and so are the four grandchildren: one address each.  A node on the
// last two levels has no grandchildren (or children): the run is moved to the end of the tree, and nothing of it is used.
template <uint32_t MAX_PASSES = 0, class LoadNode, class LoadRun, class TCur>
__device__ __forceinline__ uint32_t nearDescend(LoadNode loadNode, LoadRun loadRun, uint32_t nodeCount, const SlabArgs &sa, const LevelColumn &stack, TCur tcur,
                                                NearWalk &w, uint4 &n
#ifdef ACR_WALK_DEBUG
                                                , unsigned long long *dbg
#endif
)
{
	const uint32_t lastPair = nodeCount >= 2u ? nodeCount - 2u : 0u, lastQuad = nodeCount >= 4u ? nodeCount - 4u : 0u;
	n = loadNode(w.idx);
	uint32_t passes = 0;
	for (;;) {
		if constexpr (MAX_PASSES != 0) {
			if (passes++ == MAX_PASSES)
				return 2u;
		}
#ifdef ACR_WALK_DEBUG
		walkDebugAdd(dbg, 1, 2);
#endif
		// (single variables: an array that is indexed by the two decisions below ends up in scratch memory)
		uint4 c0, c1, g0, g1, g2, g3;
		{
			uint4 c[2], g[4];
			loadRun(min(2u * w.idx + 1u, lastPair), c);
			loadRun(min(4u * w.idx + 3u, lastQuad), g);
			c0 = c[0], c1 = c[1];
			g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
		}
		if ((n.w & kNodeLeafBit) != 0)
			return 0u;
		const uint32_t k = nearNodeStep(n, sa, stack, tcur(), w);
		if (k == 2u)
			return 1u;
		n = k ? c1 : c0;
		if ((n.w & kNodeLeafBit) != 0)
			return 0u;
		const uint32_t j = nearNodeStep(n, sa, stack, tcur(), w);
		if (j == 2u)
			return 1u;
		n = k ? (j ? g3 : g2) : (j ? g1 : g0);
	}
}

// One round of the walk: from node w.idx down to a leaf (or to a node neither of whose children can hold a nearer hit),
// the leaf's items, and the resumption at the deepest waiting child that still can.  Returns false when nothing waits
// any more: the walk is over.  bihTraverseNear is rounds until then.
template <class LoadNode, class LoadRun, class Leaf, class TCur>
__device__ __forceinline__ bool nearWalkRound(LoadNode loadNode, LoadRun loadRun, uint32_t nodeCount, const SlabArgs &sa, const LevelColumn &stack, Leaf leaf,
                                              TCur tcur, NearWalk &w
#ifdef ACR_WALK_DEBUG
                                              , unsigned long long *dbg
#endif
)
{
#ifdef ACR_WALK_DEBUG
	walkDebugAdd(dbg, 6, 7);
#endif
	uint4 n;
#ifdef ACR_WALK_DEBUG
	const uint32_t how = nearDescend(loadNode, loadRun, nodeCount, sa, stack, tcur, w, n, dbg);
#else
	const uint32_t how = nearDescend(loadNode, loadRun, nodeCount, sa, stack, tcur, w, n);
#endif
	if (how == 0u) {
		if (leaf(n.z, n.w & kNodeEndMask, w.tminB))
			return false;
	}
	return popNear(stack, w.pending, w.idx, w.level, w.tminB, w.tmaxB, tcur());
}

#ifdef ACR_WALK_DEBUG
// adds one walk's (or one wave-cooperative walk's lane's) sums to the totals
__device__ __forceinline__ void walkDebugFlush(const unsigned long long *a, bool histogram)
{
	for (int k = 0; k < 8; k++)
		if (k != 3 && k != 4 && k != 5 && a[k]) atomicAdd(&g_walkDebug[k], a[k]);
	if (histogram) {
		// [8]: walks of fewer than 8 rounds, [8 + b]: of 2^(b+2) .. 2^(b+3) - 1 rounds (b = 7: 512 and more)
		const int b = 31 - __clz(int(a[6] | 1ull)) - 2;
		atomicAdd(&g_walkDebug[8 + (b < 0 ? 0 : b < 7 ? b : 7)], 1ull);
	}
}
#endif

template <bool KEYS = false, class LoadNode, class LoadRun, class Leaf, class TCur>
__device__ __forceinline__ void bihTraverseNear(LoadNode loadNode, LoadRun loadRun, uint32_t nodeCount, const SlabArgs &sa, Interval box, const LevelColumn &stack,
                                                Leaf leaf, TCur tcur, PathRec *path = nullptr)
{
	NearWalk w = {0u, 0u, 0u, box.tmin, box.tmax};
#ifdef ACR_WALK_DEBUG
	unsigned long long dbg[8] = {1, 0, 0, 0, 0, 0, 0, 0};
	while (nearWalkRound(loadNode, loadRun, nodeCount, sa, stack, leaf, tcur, w, dbg)) {
	}
	walkDebugFlush(dbg, true);
#else
	while (nearWalkRound(loadNode, loadRun, nodeCount, sa, stack, leaf, tcur, w)) {
	}
#endif
}

// HitInfo (geometry.h:22-27) for the world-space query, plus the primitive id
struct WorldHit
{
	float t;
	vec3 position, normal;
	uint32_t material, prim;
};

// closest hit inside one object, in the object's local space (tmpInfo of scene.cu:418-424)
struct LocalHit
{
	float t;
	vec3 bary;
	uint32_t tri; // slot index of the triangle record
	uint32_t material;
};

template <class Mem>
__device__ __forceinline__ mat4 loadMat4(const Mem &mem, uint32_t slot)
{
	mat4 m;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const uint4 q = mem.small(slot + j);
		m.c[j] = vec4(asF(q.x), asF(q.y), asF(q.z), asF(q.w));
	}
	return m;
}
template <class Mem>
__device__ __forceinline__ mat3 loadMat3(const Mem &mem, uint32_t slot, uint32_t &w9, uint32_t &w10)
{
	const uint4 q0 = mem.small(slot), q1 = mem.small(slot + 1), q2 = mem.small(slot + 2);
	mat3 m;
	m.c[0] = vec3(asF(q0.x), asF(q0.y), asF(q0.z));
	m.c[1] = vec3(asF(q0.w), asF(q1.x), asF(q1.y));
	m.c[2] = vec3(asF(q1.z), asF(q1.w), asF(q2.x));
	w9 = q2.y;
	w10 = q2.z;
	return m;
}

// This lane's per-level LDS columns of the scene tree and of the mesh tree (floats for the six-plane form's undo
// log, float2 for the interval form), and where the mesh column continues in global memory.
struct Columns
{
	float *scene, *mesh;
	float2 *meshGlob;
	uint32_t meshGlobLevels, meshGlobStride;
};

// what a near-side-first traversal needs to know whether the visiting order could have mattered
struct OrderCheck
{
	float second;   // distance of the nearest hit that lost against the current one
	float leafTmin; // tmin of the leaf box the current hit was found in
};

// Face::intersect (geometry.h:123-146) for the items [start, end) of a leaf
template <bool NEAR = false, bool NEWTON = false, uint32_t TRIP = ACR_TRIS_PER_TRIP, class Mem>
__device__ __forceinline__ void leafTriangles(const Mem &mem, uint32_t triBase, uint32_t triFirst, uint32_t start, uint32_t end, const Ray &lr,
                                              struct LocalHit &tmp, bool &intersected, OrderCheck *order = nullptr, float leafTmin = 0.0f);

// Mesh::intersect (geometry.cu:52-61) for a local-space ray
// A mesh's record: box, material, where its nodes and triangles start, tree depth.
struct MeshRecord
{
	uint4 q0, q1, q2;
};
template <class Mem>
__device__ __forceinline__ MeshRecord loadMeshRecord(const Mem &mem, const SceneHeader &sc, uint32_t meshId)
{
	const uint32_t ms = sc.off_meshes + meshId * kMeshSlots;
	MeshRecord m;
	m.q0 = mem.small(ms);
	m.q1 = mem.small(ms + 1);
	m.q2 = mem.small(ms + 2);
	return m;
}

template <bool FAST, class Mem>
__device__ __forceinline__ bool meshIntersect(const Mem &mem, const SceneHeader &sc, const MeshRecord &rec, const Ray &lr,
                                              const SlabArgs &sa, LocalHit &tmp, const Columns &col, bool &bail)
{
	constexpr bool NEAR = FAST && Mem::kNearFirst;
	const uint4 q0 = rec.q0, q1 = rec.q1, q2 = rec.q2;
	const vec3 bmin(asF(q0.x), asF(q0.y), asF(q0.z)), bmax(asF(q1.x), asF(q1.y), asF(q1.z));
	const uint32_t nodeBase = sc.off_mnodes + q1.w;
	const uint32_t triBase = sc.off_tris + q2.x * kTriSlots;
	bool intersected = false;
	auto leaf = [&](uint32_t start, uint32_t end) {
		leafTriangles<false, FAST>(mem, triBase, q2.x, start, end, lr, tmp, intersected);
		return false;
	};
	auto loadNode = [&](uint32_t i) { return mem.large(nodeBase + i); };
	auto tcur = [&]() { return tmp.t; };
	// boundingBox.intersect(r, info): the BIH root box is this same box (geometry.cu:47)
	if constexpr (FAST) {
		const Interval root = boxInterval(sa, bmin, bmax);
		if (!intervalHit(root.tmin, root.tmax, tmp.t))
			return false;
		const LevelColumn lc = {reinterpret_cast<float2 *>(col.mesh), col.meshGlob, col.meshGlobLevels, col.meshGlobStride};
		if constexpr (NEAR) {
			OrderCheck order = {FLT_MAX, 0.0f};
			auto leafNear = [&](uint32_t start, uint32_t end, float leafTmin) {
				leafTriangles<true, true>(mem, triBase, q2.x, start, end, lr, tmp, intersected, &order, leafTmin);
				return false;
			};
			auto tslack = [&]() { return tmp.t * kNearPruneSlack; };
			auto loadRun = [&](uint32_t i, auto &out) { mem.largeRun(nodeBase + i, out); };
			bihTraverseNear(loadNode, loadRun, (1u << q2.y) - 1u, sa, root, lc, leafNear, tslack);
			// order.second == FLT_MAX: nothing lost in this mesh, and the hit (if any) replaced no earlier mesh's
			if (order.second < FLT_MAX && order.second <= order.leafTmin) bail = true;
		} else {
			bihTraverseFast(loadNode, sa, root, lc, leaf, tcur);
		}
	} else {
		SlabBox root;
		root.init(sa, bmin, bmax);
		if (!slabTest(root.nearT, root.farT, tmp.t))
			return false;
		bihTraverse(loadNode, q2.y, sa, root, col.mesh, leaf, tcur);
	}
	if (intersected) {
		tmp.material = q0.w;
		return true;
	}
	return false;
}

template <bool NEAR, bool NEWTON, uint32_t TRIP, class Mem>
__device__ __forceinline__ void leafTriangles(const Mem &mem, uint32_t triBase, uint32_t triFirst, uint32_t start, uint32_t end, const Ray &lr,
                                              LocalHit &tmp, bool &intersected, OrderCheck *order, float leafTmin)
{
	auto test = [&](uint32_t i, const uint4 &t0, const uint4 &t1, const uint4 &t2) {
		const vec3 a(asF(t0.x), asF(t0.y), asF(t0.z));
		const vec3 e1(asF(t0.w), asF(t1.x), asF(t1.y));
		const vec3 e2(asF(t1.z), asF(t1.w), asF(t2.x));
		vec3 b;
		float t;
		if (myIntersectRayTriangleEdges<NEWTON>(lr.o, lr.d, a, e1, e2, b, t)) {
			bool take = t < tmp.t;
			if constexpr (NEAR) {
				// An exact tie goes to the hit the reference finds first: items are numbered in the reference's
				// visiting order (checked at upload), and a tie with an earlier mesh's hit stays with that mesh.
				const bool tie = t == tmp.t;
				if (take) order->second = tmp.t;
				else if (!tie && t < order->second) order->second = t;
				take = take || (tie && intersected && i + triFirst < tmp.tri);
				if (take) order->leafTmin = leafTmin;
			}
			if (take) {
				tmp.t = t;
				tmp.bary = b;
				tmp.tri = i + triFirst;
				intersected = true;
			}
		}
	};
	if constexpr (Mem::kLargeInLds) {
		for (uint32_t i = start; i < end; i++) {
			const uint32_t ts = triBase + i * kTriSlots;
			test(i, mem.large(ts), mem.large(ts + 1), mem.large(ts + 2));
		}
	} else {
		// Triangles in HBM: a leaf's tests are a chain of memory round trips, so TRIP records (ACR_TRIS_PER_TRIP unless the caller says otherwise) are fetched
		// per trip (the first one again where the leaf has no more) and tested in item order (10 M-triangle soup, 4.8 triangles
		// per leaf on average: 3.06 ms with three per trip, 3.00 with four, 2.94 with five, 2.96 with six).
		// A leaf's records follow one another in memory: one address for the trip, and past the leaf's end whatever
		// follows (read, never tested).  Every record is requested before the first test starts.
		static_assert(TRIP * kTriSlots <= kBlobPadSlots + kTriSlots, "a trip may read past the last triangle");
		for (uint32_t i = start; i < end; i += TRIP) {
#ifdef ACR_WALK_DEBUG
			if constexpr (NEAR) {
				unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
				walkDebugAdd(acc, 3, 4);
				atomicAdd(&g_walkDebug[3], acc[3]);
				atomicAdd(&g_walkDebug[4], acc[4]);
				atomicAdd(&g_walkDebug[5], (unsigned long long)(end - i < TRIP ? end - i : TRIP));
			}
#endif
			uint4 r[TRIP * kTriSlots];
			mem.largeRun(triBase + i * kTriSlots, r);
			__builtin_amdgcn_sched_barrier(0);
			test(i, r[0], r[1], r[2]);
#pragma unroll
			for (uint32_t k = 1; k < TRIP; k++)
				if (i + k < end) test(i + k, r[k * kTriSlots], r[k * kTriSlots + 1], r[k * kTriSlots + 2]);
		}
	}
}

// Object::intersect (scene.cu:407-440).  FULL = false skips what a shadow ray
// never reads (normal, primitive id); info.t is computed identically.
//
// Before the reference's work starts, the ray is tested against the object's
// world-space box grown by a margin (device_scene.h, kObjPrefilter).  A ray
// that misses the grown box misses the object's local mesh boxes by a distance
// three to four orders of magnitude above fp32 rounding, so the reference's
// own mesh box test (geometry.cu:55) would fail for every mesh and the call
// would return false with `info` untouched: returning false here is the same
// result without the two matrix transforms.  The host sets the flag only for
// rotation * uniform scale transforms, where the reference's local ray is the
// true image of the world ray.
// `tLimit` also drops an object whose grown box the ray enters no nearer than
// tLimit (a slab distance, i.e. in units of the ray parameter): the world-space
// t of any hit in it (scene.cu:431) exceeds the world distance of that entry
// point by at least the margin, so with tLimit = info.t / |d| it could not
// replace info (scene.cu:432).  The call would still return true there, so the
// caller passes that limit only when the return value can no longer matter (a
// shadow ray, or `intersected` already set), FLT_MAX otherwise.  |d| is 1 for
// primary and shadow rays, which are normalised; bounce directions are not
// (math.h:188-192), so the caller measures them.
// FAST: interval traversal; sets `bail` and returns when the local ray does not qualify for it.
// PREFILTER = false: the caller has made the world-box test already (saWorld and tLimit are not read).
// winner: if given, receives the triangle and the material of the object's closest hit (LocalHit::tri, ::material).
// PARALLEL: the instantiation a primary ray is traced with again after the interval form turned it away for an
// infinite 1 / d (parallelRayQualifies); kept apart from the common one so that the common one carries nothing for it.
template <bool FULL, bool FAST, bool PREFILTER = true, bool PARALLEL = false, class Mem>
__device__ __forceinline__ bool objectIntersect(const Mem &mem, const SceneHeader &sc, uint32_t obj, const Ray &r, const SlabArgs &saWorld,
                                                WorldHit &info, const Columns &col, bool &bail, float tLimit, uint2 *winner = nullptr)
{
	const uint32_t os = sc.off_objects + obj * Mem::kObjStride;
	const uint4 b0 = mem.small(os), b1 = mem.small(os + 1);
	const uint32_t flags = b0.w;
	if (PREFILTER && (flags & kObjPrefilter)) {
		const vec3 wmin(asF(b0.x), asF(b0.y), asF(b0.z)), wmax(asF(b1.x), asF(b1.y), asF(b1.z));
		if constexpr (FAST) {
			const Interval wb = boxInterval(saWorld, wmin, wmax);
			if (!intervalHit(wb.tmin, wb.tmax, tLimit))
				return false;
		} else {
			SlabBox wb;
			wb.init(saWorld, wmin, wmax);
			if (!slabTest(wb.nearT, wb.farT, tLimit))
				return false;
		}
	}
	// an object's first mesh is named in its box record: the list of mesh numbers (one more LDS round trip on the
	// way to the mesh) is only read for further meshes
	const uint32_t firstMesh = b1.w;
	Ray lr;
	uint32_t meshBegin, meshCount;
	{
		const mat4 invM = loadMat4(mem, os + 2);
		const mat3 invN = loadMat3(mem, os + 6, meshBegin, meshCount);
		// hom.w is exactly 1 for an affine matrix and x / 1 == x (math.h:174-178)
		lr.o = (flags & kObjInvAffine) ? translateAffine(invM, r.o) : translate(invM, r.o);
		lr.d = translaten(invN, r.d);
	}
	SlabArgs sa;
	if constexpr (FAST) {
		if (!(PARALLEL ? parallelRayQualifies(sa, lr, obj) : sa.initInterval(lr))) {
			bail = true;
			return false;
		}
	} else {
		sa.init(lr);
	}

	bool intersected = false;
	LocalHit tmp;
	tmp.t = FLT_MAX;
	tmp.tri = 0;
	tmp.material = 0;
	tmp.bary = vec3(0, 0, 0);
	for (uint32_t k = 0; k < meshCount; k++) {
		uint32_t meshId = firstMesh;
		if (k != 0) {
			const uint32_t idSlot = sc.off_objmesh + ((meshBegin + k) >> 2);
			const uint4 ids = mem.small(idSlot);
			const uint32_t sel = (meshBegin + k) & 3u;
			meshId = sel == 0 ? ids.x : (sel == 1 ? ids.y : (sel == 2 ? ids.z : ids.w));
		}
		intersected |= meshIntersect<FAST>(mem, sc, loadMeshRecord(mem, sc, meshId), lr, sa, tmp, col, bail);
	}

	if (intersected) {
		if (winner) *winner = make_uint2(tmp.tri, tmp.material);
		// info.point.position = r.o + r.d * t in local space (geometry.h:140), then to world (scene.cu:428)
		const vec3 pl = lr.o + lr.d * tmp.t;
		const mat4 M = loadMat4(mem, os + 9);
		const vec3 pw = (flags & kObjFwdAffine) ? translateAffine(M, pl) : translate(M, pl);
		const float t = length(pw - r.o);
		if (t < info.t) {
			info.t = t;
			if constexpr (FULL) {
				info.position = pw;
				info.material = tmp.material;
				const uint32_t ns = sc.off_normals + tmp.tri * kNormalSlots;
				const uint4 n0 = mem.large(ns), n1 = mem.large(ns + 1), n2 = mem.large(ns + 2);
				const vec3 na(asF(n0.x), asF(n0.y), asF(n0.z));
				const vec3 nb(asF(n0.w), asF(n1.x), asF(n1.y));
				const vec3 nc(asF(n1.z), asF(n1.w), asF(n2.x));
				const vec3 nl = tmp.bary.x * na + tmp.bary.y * nb + tmp.bary.z * nc; // geometry.h:141
				uint32_t u0, u1;
				const mat3 N = loadMat3(mem, os + 13, u0, u1);
				info.normal = translaten(N, nl); // scene.cu:429
				info.prim = mem.large(sc.off_tris + tmp.tri * kTriSlots + 2).y;
			}
		}
	}
	return intersected;
}

// What objectIntersect<true> returns for a ray whose closest hit inside object `obj` is known to be triangle `tri`
// (an earlier objectIntersect<false> of the same ray said so): the same local ray, that one triangle's test, and
// the hit's world position, normal, material and primitive id -- without the walk through the mesh trees.
template <class Mem>
__device__ __forceinline__ void objectHitDetails(const Mem &mem, const SceneHeader &sc, uint32_t obj, uint32_t tri, uint32_t material, const Ray &r,
                                                 WorldHit &info)
{
	const uint32_t os = sc.off_objects + obj * Mem::kObjStride;
	const uint32_t flags = mem.small(os).w;
	Ray lr;
	{
		uint32_t meshBegin, meshCount;
		const mat4 invM = loadMat4(mem, os + 2);
		const mat3 invN = loadMat3(mem, os + 6, meshBegin, meshCount);
		lr.o = (flags & kObjInvAffine) ? translateAffine(invM, r.o) : translate(invM, r.o);
		lr.d = translaten(invN, r.d);
	}
	const uint32_t ts = sc.off_tris + tri * kTriSlots;
	const uint4 t0 = mem.large(ts), t1 = mem.large(ts + 1), t2 = mem.large(ts + 2);
	const vec3 a(asF(t0.x), asF(t0.y), asF(t0.z));
	const vec3 e1(asF(t0.w), asF(t1.x), asF(t1.y));
	const vec3 e2(asF(t1.z), asF(t1.w), asF(t2.x));
	vec3 bary(0, 0, 0);
	float tl = 0.0f;
	(void)myIntersectRayTriangleEdges<true>(lr.o, lr.d, a, e1, e2, bary, tl); // it hit before: same operands, same result
	const vec3 pl = lr.o + lr.d * tl;
	const mat4 M = loadMat4(mem, os + 9);
	const vec3 pw = (flags & kObjFwdAffine) ? translateAffine(M, pl) : translate(M, pl);
	info.t = length(pw - r.o);
	info.position = pw;
	info.material = material;
	const uint32_t ns = sc.off_normals + tri * kNormalSlots;
	const uint4 n0 = mem.large(ns), n1 = mem.large(ns + 1), n2 = mem.large(ns + 2);
	const vec3 na(asF(n0.x), asF(n0.y), asF(n0.z));
	const vec3 nb(asF(n0.w), asF(n1.x), asF(n1.y));
	const vec3 nc(asF(n1.z), asF(n1.w), asF(n2.x));
	const vec3 nl = bary.x * na + bary.y * nb + bary.z * nc; // geometry.h:141
	uint32_t u0, u1;
	const mat3 N = loadMat3(mem, os + 13, u0, u1);
	info.normal = translaten(N, nl); // scene.cu:429
	info.prim = t2.y;
}

// Scene::intersect (scene.cu:204-207): BIH<Object> traversal in world space.
//
// SHADOW: the callers at scene.cu:226,266 only ask whether
// `info.t + eps < t` after the traversal, with info.t preset to t.  info.t
// only ever decreases, and x -> x + eps is monotonic in binary32, so as soon
// as one object leaves info.t + eps < t the answer is fixed: the remaining
// objects can be skipped without changing it.  (The boolean `intersected` the
// reference also tests is implied: info.t can only drop below t through a
// hit.)  Returns that occlusion flag for SHADOW, `intersected` otherwise.
template <bool SHADOW, bool FAST, bool KEYS = false, bool PARALLEL = false, class Mem>
__device__ __forceinline__ bool sceneIntersect(const Mem &mem, const SceneHeader &sc, const Ray &r, WorldHit &info, float lightT,
                                               const Columns &col, bool &bail, PathRec *path = nullptr, float distPerT = 0.0f)
{
	SlabArgs sa;
	if constexpr (FAST) {
		if (!(PARALLEL ? parallelRayQualifies(sa, r, kParallelMaxObjects) : sa.initInterval(r))) {
			bail = true;
			return false;
		}
	} else {
		sa.init(r);
	}
	const vec3 smin(sc.scene_min[0], sc.scene_min[1], sc.scene_min[2]), smax(sc.scene_max[0], sc.scene_max[1], sc.scene_max[2]);
	bool intersected = false;
	bool occluded = false;
	auto loadNode = [&](uint32_t i) { return mem.small(sc.off_snodes + i); };
	auto leaf = [&](uint32_t start, uint32_t end) {
		for (uint32_t i = start; i < end; i++) {
			intersected |= objectIntersect<!SHADOW, FAST, true, PARALLEL>(mem, sc, i, r, sa, info, col, bail,
			                                                              (distPerT != 0.0f && (SHADOW || intersected)) ? info.t * distPerT : FLT_MAX);
			if constexpr (FAST) {
				if (bail) return true;
			}
			if constexpr (SHADOW) {
				if (info.t + epsilon() < lightT) {
					occluded = true;
					return true;
				}
			}
		}
		return false;
	};
	auto tcur = [&]() { return info.t; };
	if constexpr (FAST) {
		const LevelColumn lc = {reinterpret_cast<float2 *>(col.scene), nullptr, 0u, 0u};
		bihTraverseFast<KEYS>(loadNode, sa, boxInterval(sa, smin, smax), lc, leaf, tcur, path);
	} else {
		SlabBox root;
		root.init(sa, smin, smax);
		bihTraverse<KEYS>(loadNode, sc.scene_depth, sa, root, col.scene, leaf, tcur, path);
	}
	return SHADOW ? occluded : intersected;
}

struct Light
{
	uint32_t type;
	vec3 position, direction, diffuse;
	float attConstant, attLinear, attQuadratic, innerCone, outerCone;
};
template <class Mem>
__device__ __forceinline__ Light loadLight(const Mem &mem, const SceneHeader &sc, uint32_t i)
{
	const uint32_t s = sc.off_lights + i * kLightSlots;
	const uint4 a = mem.small(s), b = mem.small(s + 1), c = mem.small(s + 2), d = mem.small(s + 3);
	Light l;
	l.position = vec3(asF(a.x), asF(a.y), asF(a.z));
	l.type = a.w;
	l.direction = vec3(asF(b.x), asF(b.y), asF(b.z));
	l.attConstant = asF(b.w);
	l.diffuse = vec3(asF(c.x), asF(c.y), asF(c.z));
	l.attLinear = asF(c.w);
	l.attQuadratic = asF(d.x);
	l.innerCone = asF(d.y);
	l.outerCone = asF(d.z);
	return l;
}

struct Counters
{
	uint32_t shadow, bounce, primaryHits;
};

// Scene::lightPoint (scene.cu:273-292) with pointLightAccum (209-231) and spotLightAccum (234-271)
template <bool FAST, class Mem>
__device__ __forceinline__ vec3 lightPoint(const Mem &mem, const SceneHeader &sc, const vec3 &pos, const vec3 &norm, RandState &state,
                                           Counters &cnt, const Columns &col, bool &bail)
{
	vec3 light(0, 0, 0);
	for (uint32_t i = 0; i < sc.num_lights; i++) {
		const Light l = loadLight(mem, sc, i);
		vec3 dir, c;
		float t;
		bool trace = true;
		if (l.type == 0u) {
			dir = (l.position + 0.5f * randNorm(state)) - pos;
			t = length(dir);
			dir = normalize(dir);
			const float cosTheta = max(dot(dir, norm), 0.0f);
			c = cosTheta / (l.attConstant + (l.attLinear + l.attQuadratic * t) * t) * l.diffuse;
			if (length(c) < epsilon()) trace = false;
		} else {
			dir = l.position - pos;
			t = length(dir);
			dir = normalize(dir);
			const float cosTheta = max(dot(dir, norm), 0.0f);
			const float cosLight = dot(-dir, normalize(l.direction));
			const float lightTheta = acosf(cosLight);
			const float inner = l.outerCone - 1.57079632679f;
			const float outer = l.innerCone;
			const float falloff_distance = outer - inner;
			if (fabsf(lightTheta) >= outer / 2.0f) {
				c = vec3(0, 0, 0);
				trace = false;
			} else {
				c = cosTheta / (l.attConstant + (l.attLinear + l.attQuadratic * t) * t) * l.diffuse;
				if (length(c) < epsilon()) {
					trace = false;
				} else {
					const float val = (outer / 2.0f - fabsf(lightTheta));
					const float falloff = val / (falloff_distance / 2.0f);
					if (double(fabsf(lightTheta)) >= double(inner) / 2.0)
						c *= falloff;
				}
			}
		}
		if (trace) {
			Ray sr;
			sr.o = pos;
			sr.d = dir;
			WorldHit sh;
			sh.t = t;
			cnt.shadow++;
			if (sceneIntersect<true, FAST>(mem, sc, sr, sh, t, col, bail, nullptr, 1.0f))
				c = vec3(0, 0, 0);
			if constexpr (FAST) {
				if (bail) return light;
			}
		}
		light += c;
	}
	return light;
}

// get_pixel_dir (renderer.cu:184-206).  Its camera-only part -- cR = cross(dir, up), cU = cross(cR, dir),
// dist = fastertanfull(hFOV / 2) (renderer.cu:197-200) -- is the same for every pixel and is evaluated once on
// the host with the same single-operation arithmetic (cameraBasis in render_params.h); the per-pixel
// expression keeps the reference's association.
__device__ __forceinline__ vec3 getPixelDir(const RenderParams &p, float ni, float nj)
{
	const vec3 dir(p.cam_forward[0], p.cam_forward[1], p.cam_forward[2]);
	const float AR = p.cam_aspect;
	const vec3 cR(p.cam_cr[0], p.cam_cr[1], p.cam_cr[2]);
	const vec3 cU(p.cam_cu[0], p.cam_cu[1], p.cam_cu[2]);
	const float dist = p.cam_dist;
	return normalize(dir + dist * (nj * cU + AR * ni * cR));
}

__device__ __forceinline__ uint32_t waveSum(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v += __shfl_xor(v, off, 64);
	return v;
}

// Copies blob slots [0, n) into LDS at the same offsets, 16 B per lane per step.
__device__ __forceinline__ void stageScene(const uint4 *__restrict__ g, uint32_t n)
{
	for (uint32_t i = threadIdx.x; i < n; i += kBlock)
		g_lds[i] = g[i];
	__syncthreads();
}

// The same copy with the object records kObjectSlots + 1 slots apart: blob slot i goes to LDS slot i before the object
// table, to its place in the spread table inside it, and num_objects slots further on behind it.  spreadHeader() is the
// scene header that describes the copy.
__device__ __forceinline__ void stageSceneSpread(const uint4 *__restrict__ g, const SceneHeader &sc)
{
	const uint32_t first = sc.off_objects, end = sc.off_objects + sc.num_objects * kObjectSlots;
	for (uint32_t i = threadIdx.x; i < sc.total_slots; i += kBlock) {
		constexpr uint32_t extra = kWaveObjStride - kObjectSlots; // 1 (or 0 in a build that compares the two layouts)
		uint32_t dst = i;
		if (i >= end) dst = i + extra * sc.num_objects;
		else if (i >= first) dst = i + extra * ((i - first) / kObjectSlots);
		g_lds[dst] = g[i];
	}
	__syncthreads();
}
__device__ __forceinline__ SceneHeader spreadHeader(const SceneHeader &in)
{
	SceneHeader sc = in;
	const uint32_t n = in.num_objects * (kWaveObjStride - kObjectSlots);
	sc.off_objmesh += n; sc.off_meshes += n; sc.off_materials += n; sc.off_lights += n;
	sc.off_mnodes += n; sc.off_tris += n; sc.off_normals += n;
	sc.small_slots += n;
	sc.total_slots += n;
	return sc;
}

// What one pixel slot reports back to the kernel around it.
struct SlotResult
{
	bool active;        // the slot maps to a pixel inside the frame (renderer.cu:335-338)
	bool bail;          // FAST only: a ray of this pixel needs the six-plane traversal; nothing was written
	uint32_t tileNo;    // position in the tile list (tile mode)
	Counters cnt;
};

// scatterTrace (renderer.cu:326-388) for one pixel slot, with rayColor<N> (renderer.cu:232-324) unrolled into a loop.
// (vblock, vtid) name the slot: workgroup vblock of the frame's launch grid, work-item vtid in it.  The LDS
// columns belong to the executing work-item (threadIdx.x), which need not be vtid.
// KEYS: also records the pixel's traversal-path key (the reference's per-pixel Path, renderer.cu:341,372,380,386).
// ONE: the launch has max_bounces == 1 (rayColor<1>: primary ray, shadow rays, no continuation), known at compile
// time as MAX_BOUNCES is in the reference (renderer.cu:10), so that nothing of the bounce loop is carried.
template <bool FAST, bool KEYS, bool ONE = false, class Mem>
__device__ __forceinline__ SlotResult renderSlot(const RenderParams &p, const Mem &mem, uint32_t vblock, uint32_t vtid)
{
	const uint32_t maxBounces = ONE ? 1u : p.max_bounces;
	const SceneHeader &sc = p.scene;
	float *ldsF = reinterpret_cast<float *>(g_lds + p.lds_stack_slot);
	Columns col;
	col.scene = ldsF + (FAST ? 2 : 1) * threadIdx.x;
	col.mesh = col.scene + (FAST ? 2 : 1) * sc.scene_depth * kBlock;
	// a scene staged whole into LDS has no tree deep enough for the global part: a compile-time zero there
	col.meshGlobLevels = (FAST && !Mem::kLargeInLds) ? p.mesh_glob_levels : 0u;
	col.meshGlobStride = p.mesh_glob_stride;
	col.meshGlob = p.mesh_glob + (size_t(blockIdx.x) * kBlock + threadIdx.x);
	float *cstack = ldsF + p.lds_cstack_off + threadIdx.x;

	// which pixel slot: 16x16 pixels per workgroup, 8x8 per wave
	const uint32_t lane = vtid & 63u, wave = vtid >> 6;
	const uint32_t lx = (lane & 7u) + ((wave & 1u) << 3), ly = (lane >> 3) + ((wave >> 1) << 3);
	uint32_t oldx, oldy;
	size_t outSlot;
	SlotResult res;
	res.tileNo = 0;
	res.bail = false;
	res.cnt = {0, 0, 0};
	const bool tiled = p.num_tiles != 0;
	if (tiled) {
		const uint32_t bpx = p.tile_w >> 4, bpt = bpx * (p.tile_h >> 4);
		const uint32_t tileNo = vblock / bpt, sub = vblock % bpt;
		const uint32_t tile = p.tile_ids[tileNo];
		res.tileNo = tileNo;
		const uint32_t inx = ((sub % bpx) << 4) + lx, iny = ((sub / bpx) << 4) + ly;
		oldx = (tile % p.tiles_x) * p.tile_w + inx;
		oldy = (tile / p.tiles_x) * p.tile_h + iny;
		outSlot = p.tiles_into_frame ? size_t(oldy) * p.width + oldx : size_t(tileNo) * p.tile_w * p.tile_h + size_t(iny) * p.tile_w + inx;
	} else {
		oldx = (vblock % p.blocks_x) * 16u + lx;
		oldy = (vblock / p.blocks_x) * (kSlots / 16u) + ly;
		outSlot = 0;
	}

	Counters &cnt = res.cnt;
	res.active = oldx < p.width && oldy < p.height; // renderer.cu:335-338
	if (!res.active)
		return res;

	const int width = int(p.width);
	const int oldIndex = int(oldx) + width * int(oldy);
	int index, x, y;
	if (p.frames == 0 || !p.pixel_values) { // renderer.cu:344-352
		if (p.pixel_values) p.pixel_values[oldIndex] = oldIndex;
		index = oldIndex;
		// index % width and index / width (renderer.cu:354-355) of oldx + width * oldy with oldx < width
		x = int(oldx);
		y = int(oldy);
	} else {
		index = p.pixel_values[oldIndex];
		x = index % width;
		y = index / width;
	}
	if (!tiled) outSlot = size_t(index);

	RandState state;
	randInit(state, uint64_t(int64_t(index)) + p.seed); // renderer.cu:358

	const float dx = p.inv_width;  // 1.0f / width, 1.0f / height (renderer.cu:360-361), divided once on the host
	const float dy = p.inv_height;
	const float i = 2.0f * (float(x) + randUniform(state)) * dx - 1.0f;
	const float j = 2.0f * (float(y) + randUniform(state)) * dy - 1.0f;

	Ray r;
	r.o = vec3(p.cam_position[0], p.cam_position[1], p.cam_position[2]);
	r.d = getPixelDir(p, i, -j);

	// rayColor<max_bounces>(r, Color3(1,1,1), ...)
	vec3 source(1, 1, 1);
	vec4 tail(0, 0, 0, 1);
	uint32_t level = 0;
	bool bail = false;
	PathRec pathRec;
	pathRec.path = 0;
	pathRec.length = 0;
	// stored early so that they need no registers later; a pixel that bails is written again by the list pass
	if (p.primid) p.primid[outSlot] = 0xFFFFFFFFu;
	if (p.hit_t) p.hit_t[outSlot] = FLT_MAX;
	for (;;) {
		if (level == maxBounces) // rayColor<0> (renderer.cu:318-324)
			break;
		WorldHit info;
		info.t = FLT_MAX;
		info.material = 0;
		info.prim = 0xFFFFFFFFu;
		info.position = vec3(0, 0, 0);
		info.normal = vec3(0, 0, 0);
		if (level > 0) cnt.bounce++;
		bool hit;
		if constexpr (KEYS) {
			// Only the first Scene::intersect of a pixel can reach the key: every traversal pads the Path's length
			// to 2 * MAX_SIZE bits (bih.h:293-297), at least 126 for the depths the key is offered for.
			if (level == 0) hit = sceneIntersect<false, FAST, true>(mem, sc, r, info, 0.0f, col, bail, &pathRec, 1.0f);
			else if (FAST && hitsNothing(sc, r)) hit = false;
			else hit = sceneIntersect<false, FAST>(mem, sc, r, info, 0.0f, col, bail);
		} else {
			// slab distances per unit of world distance: 1 for the normalised primary ray, 1 / |d| for a bounce ray
			if (FAST && level > 0 && hitsNothing(sc, r)) hit = false;
			else hit = sceneIntersect<false, FAST>(mem, sc, r, info, 0.0f, col, bail, nullptr, level == 0 ? 1.0f : 1.0f / length(r.d));
			if constexpr (FAST) {
				if (bail && level == 0) {
					// a primary ray with an infinite 1 / d, most likely: once more, through the instantiation that admits it
					// where the launch vouches for the axis (nothing has been written or counted yet)
					bail = false;
					info.t = FLT_MAX;
					hit = sceneIntersect<false, true, false, true>(mem, sc, r, info, 0.0f, col, bail, nullptr, 1.0f);
				}
			}
		}
		if (FAST && bail) break;
		if (!hit)
			break; // renderer.cu:315
		if (level == 0) {
			LateParams &q = lateParams();
			if (q.primid) q.primid[outSlot] = info.prim;
			if (q.hit_t) q.hit_t[outSlot] = info.t;
			cnt.primaryHits++;
		}
		// direct illum (renderer.cu:247-248); the material record is fetched after the shadow rays
		const vec3 lightSum = lightPoint<FAST>(mem, sc, info.position, info.normal, state, cnt, col, bail);
		if (FAST && bail) break;
		const uint32_t msl = sc.off_materials + info.material * kMaterialSlots;
		const uint4 m0 = mem.small(msl), m1 = mem.small(msl + 1), m2 = mem.small(msl + 2);
		const vec3 ambient(asF(m0.x), asF(m0.y), asF(m0.z));
		const float refractiveIndex = asF(m0.w);
		const vec3 diffuse(asF(m1.x), asF(m1.y), asF(m1.z));
		const vec3 specular(asF(m2.x), asF(m2.y), asF(m2.z));
		const vec3 c = ambient + diffuse * lightSum;

		// indirect illum.  nSource is uninitialised in the reference's
		// translucent branch (renderer.cu:251,286-311); defined as `source` here.
		vec3 nSource = source;
		Ray nr;
		nr.o = info.position;
		nr.d = vec3(0, 0, 0);
		const bool last = level + 1 == maxBounces; // the next level returns (0,0,0,1) whatever the ray
		if (refractiveIndex == 0) {
			const vec3 cd = source * diffuse;
			const vec3 cs = source * specular;
			const float Pd = compMax(cd) / compMax(source);
			const float Ps = compMax(cs) / compMax(source);
			const float P = randUniform(state);
			if (P < Pd) {
				if (!last) {
					nSource = cd * (1 / Pd);
					nr.d = randomHemi(info.normal, state);
				}
			} else if (P < Pd + Ps) {
				if (!last) {
					nSource = cs * (1 / Ps);
					nr.d = reflect(r.d, info.normal);
				}
			} else {
				tail = vec4(c, 1); // absorbed (renderer.cu:282)
				break;
			}
		} else if (!last) {
			float n1, n2;
			vec3 norm = info.normal;
			if (dot(r.d, norm) < 0) {
				n1 = refractiveIndex;
				n2 = 1;
			} else {
				n1 = 1;
				n2 = refractiveIndex;
				norm *= -1.0f;
			}
			const float R = rSchlick2(r.d, norm, n1, n2);
			if (randUniform(state) >= R)
				nr.d = refract(r.d, norm, n1 / n2);
			else
				nr.d = reflect(r.d, norm);
		}
		if (last) { // Color4(c,1) + rayColor<0>() = Color4(c,1) + (0,0,0,1)
			tail = vec4(c, 1) + vec4(0, 0, 0, 1);
			break;
		}
		// Color4(c,1) + rayColor<N-1>(...): remember c, add after the deeper levels
		cstack[(3 * level + 0) * kBlock] = c.x;
		cstack[(3 * level + 1) * kBlock] = c.y;
		cstack[(3 * level + 2) * kBlock] = c.z;
		level++;
		r = nr;
		source = nSource;
	}
	if (FAST && bail) {
		// nothing of this pixel is kept: the six-plane pass renders it from the start
		res.bail = true;
		res.cnt = {0, 0, 0};
		return res;
	}
	vec4 contribution = tail;
	while (level > 0) {
		level--;
		const vec4 cl(cstack[(3 * level + 0) * kBlock], cstack[(3 * level + 1) * kBlock], cstack[(3 * level + 2) * kBlock], 1.0f);
		contribution = cl + contribution;
	}
	LateParams &q = lateParams();
	if constexpr (KEYS) q.path_keys[index] = pathRec.path; // keyed by pixel, so that keys and pixels stay together when sorted

	// average (renderer.cu:375-387)
	float4 *px = q.screen + outSlot;
	if (q.frames > 0) {
		const float4 old = *px;
		vec4 s(old.x, old.y, old.z, old.w);
		s = s * (float(q.frames) / float(q.frames + 1));
		s = s + (contribution / float(q.frames + 1));
		*px = make_float4(s.x, s.y, s.z, s.w);
	} else {
		*px = make_float4(contribution.x, contribution.y, contribution.z, contribution.w);
	}
	return res;
}


// ---- object tests as tasks compacted across the wavefront ----------------------------------------------------
//
// With incoherent rays (everything after the first bounce) the lanes of a wave reach the expensive part of a
// traversal -- Object::intersect past its box test: two matrix transforms, four divides, a square root, the mesh
// tree, the triangles -- at different moments of their walks through the tree over objects, and the wave runs
// that code once for every such moment with a few lanes live.  What an object test returns does not depend on
// the walk: scene.cu:407-440 starts from tmpInfo.t = FLT_MAX and compares with info.t only at its end
// (scene.cu:432).  So a wave traces its 64 rays in three steps:
//   A  every lane tests its ray against the grown world box of every object (the test objectIntersect makes
//      first; a ray that fails it would get `false` and an untouched info from the reference, see there) and
//      writes a task (lane, object) for each box it enters into a list in LDS, behind the tasks of the lanes
//      before it (a prefix sum over the lanes' counts): the list comes out compact.
//   B  lane j takes tasks j, j + 64, ...: it fetches that ray out of the owning lane's registers, runs the object
//      test from FLT_MAX and stores `intersected` and the world distance.  Lanes whose own path has ended, or
//      whose ray has no candidates, work on other lanes' tasks.  (A list sorted by object, which makes
//      neighbouring lanes test the same object, costs more to build than it saves: 4.07 against 3.94 ms at 4K.)
//   C  every lane walks the tree over objects for its own ray exactly as bih.h:209-299 does -- same slab
//      tests against the current info.t, same order -- and where the reference calls Object::intersect it
//      looks the result up: info is replaced iff t < info.t (scene.cu:432), `intersected` accumulates.
// For a closest-hit query the winning object is tested once more for position, normal and primitive id.
// A task list holds kWaveCap entries; when the rays of a wave have more candidates than that, steps B and C are
// made for one group of consecutive lanes after the other (all lanes work on every group's tasks).

#ifdef ACR_WAVE_DEBUG
// experiment builds only: wave traces, rays, tasks, tasks beyond the list, rounds of step B
__device__ unsigned long long g_waveDebug[16];
#define g_dbgAcc wl.dbg
#endif
// a task: lane (6 bits) | object (6) | kTaskBail | kTaskHit | triangle of the object's closest hit (kTaskTriBits) | its material
constexpr uint32_t kTaskHit = 0x2000u, kTaskBail = 0x1000u, kTaskTriShift = 14u;

// LDS of one wave for the steps above; offsets in 4-byte words from the wave's base
struct WaveLds
{
	float *results;     // [kWaveCap] world distance of each task's hit (FLT_MAX: info would not be replaced)
	uint32_t *queue;    // [kWaveCap] lane | object << 6, then | kTaskHit | kTaskBail | (triangle | material << kTaskTriBits) << kTaskTriShift
#ifdef ACR_WAVE_DEBUG
	uint64_t *dbg;      // per-lane accumulators; lane 0's are added to g_waveDebug when the wave ends
#endif
};
__device__ __forceinline__ void waveSync()
{
	// the lanes of a wave exchange data through LDS: keep the compiler from moving LDS accesses across, and make
	// the accesses issued so far complete (LDS serves a wave's instructions in order)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Scene::intersect for the rays of a wave, called by all 64 lanes together (uniform control flow).  `active`: this
// lane has a ray.  SHADOW: returns the occlusion flag of scene.cu:226,266 for a ray whose info.t starts at lightT;
// otherwise returns `intersected` and fills info (position, normal, material, prim of the closest hit).
template <bool SHADOW, class Mem>
__device__ __forceinline__ bool waveSceneIntersect(const Mem &mem, const SceneHeader &sc, const WaveLds &wl, bool active, const Ray &r, WorldHit &info,
                                                   float lightT, const Columns &col, bool &bailOut)
{
	bool bail = false;
	bool testBail = false; // the one flag every object test below reports through
	const uint32_t lane = threadIdx.x & 63u;
	SlabArgs sa;
	sa.o = vec3(0, 0, 0);
	sa.invD = vec3(1, 1, 1);
	bool part = active;
	if (part && !sa.initInterval(r)) {
		bail = true;
		part = false;
	}
#ifdef ACR_WAVE_DEBUG
	uint64_t dbgT0 = __builtin_readcyclecounter();
#define ACR_DBG_PHASE(n) { const uint64_t now_ = __builtin_readcyclecounter(); g_dbgAcc[n] += now_ - dbgT0; dbgT0 = now_; }
#else
#define ACR_DBG_PHASE(n)
#endif
	// step A: which objects' grown boxes does this lane's ray enter?  Object k + 1's box is requested before
	// object k's is tested.
	const float tLimit = SHADOW ? lightT : FLT_MAX; // info.t as the traversal starts; it only shrinks afterwards
	uint64_t candMask = 0;
	{
		uint4 nb0 = mem.small(sc.off_objects), nb1 = mem.small(sc.off_objects + 1); // slots the staging copy has filled even for an empty scene
		for (uint32_t k = 0; k < sc.num_objects; k++) {
			const uint4 b0 = nb0, b1 = nb1;
			{
				// after the last object the same record once more: read, never used
				const uint32_t osn = sc.off_objects + (k + 1 < sc.num_objects ? k + 1 : k) * Mem::kObjStride;
				nb0 = mem.small(osn);
				nb1 = mem.small(osn + 1);
			}
			const vec3 wmin(asF(b0.x), asF(b0.y), asF(b0.z)), wmax(asF(b1.x), asF(b1.y), asF(b1.z));
			const Interval wb = boxInterval(sa, wmin, wmax);
			const bool cand = part && (!(b0.w & kObjPrefilter) || intervalHit(wb.tmin, wb.tmax, tLimit));
			candMask |= cand ? 1ull << k : 0ull;
		}
	}
	// Tasks of the lanes before and including this one.  A task list holds kWaveCap entries: the lanes are served in
	// groups of consecutive lanes whose tasks fit (one group, nearly always; a lane alone always fits, as there
	// are at most 64 objects).
	uint32_t incl = uint32_t(__popcll(candMask));
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t v = __shfl_up(incl, d, 64);
		incl += lane >= uint32_t(d) ? v : 0u;
	}
	ACR_DBG_PHASE(5)
	// how a walk of step C ended early: one variable for both reasons (two flags set to `true` on two ways out of
	// the same loop end up as one store through a selected address, which pins both flags in scratch memory)
	constexpr uint32_t kEndOccluded = 1u, kEndBail = 2u;
	bool intersected = false;
	uint32_t ended = 0;
	uint32_t best = 0;
	uint32_t groupLo = 0, groupBase = 0;
	for (;;) {
		const bool inGroup = lane >= groupLo && incl - groupBase <= kWaveCap;
		const uint32_t groupHi = groupLo + uint32_t(__popcll(__ballot(inGroup)));
		waveSync(); // the previous group's (or call's) step C has read the tables
		// the group's task list, lane after lane: every lane writes its own candidates behind those of the lanes before it
		const uint32_t myBase = incl - uint32_t(__popcll(candMask)) - groupBase;
		const uint32_t queued = uint32_t(__shfl(incl, int(groupHi) - 1, 64)) - groupBase;
		if (inGroup) {
			uint32_t at = myBase;
			for (uint64_t m = candMask; m != 0; m &= m - 1)
				wl.queue[at++] = lane | (uint32_t(__builtin_ctzll(m)) << 6);
		}
#ifdef ACR_WAVE_DEBUG
		{
			const uint64_t pb = __ballot(part && inGroup);
			g_dbgAcc[0] += groupLo == 0 ? 1 : 0;
			g_dbgAcc[1] += __popcll(pb);
			g_dbgAcc[2] += queued;
			g_dbgAcc[3] += groupLo == 0 ? 0 : 1;
			g_dbgAcc[4] += (queued + 63) / 64;
		}
#endif
		waveSync();
		ACR_DBG_PHASE(5)
		// step B: the ray of task (lane rl, object obj) comes straight out of lane rl's registers
		for (uint32_t base = 0; base < queued; base += 64u) {
			const uint32_t task = base + lane;
			const bool valid = task < queued;
			const uint32_t e = valid ? wl.queue[task] : 0u;
			const int rl = int(e & 63u);
			Ray tr;
			tr.o = vec3(__shfl(r.o.x, rl, 64), __shfl(r.o.y, rl, 64), __shfl(r.o.z, rl, 64));
			tr.d = vec3(__shfl(r.d.x, rl, 64), __shfl(r.d.y, rl, 64), __shfl(r.d.z, rl, 64));
			if (valid) {
				WorldHit th;
				th.t = FLT_MAX;
				testBail = false;
				uint2 won = make_uint2(0u, 0u);
				const bool hit = objectIntersect<false, true, false>(mem, sc, (e >> 6) & 63u, tr, sa, th, col, testBail, FLT_MAX, SHADOW ? nullptr : &won);
				wl.results[task] = th.t;
				wl.queue[task] = e | (hit ? kTaskHit : 0u) | (testBail ? kTaskBail : 0u) | ((won.x | (won.y << kTaskTriBits)) << kTaskTriShift);
			}
		}
		waveSync();
		ACR_DBG_PHASE(6)
		// step C
		if (part && inGroup) {
			const vec3 smin(sc.scene_min[0], sc.scene_min[1], sc.scene_min[2]), smax(sc.scene_max[0], sc.scene_max[1], sc.scene_max[2]);
			auto loadNode = [&](uint32_t i) { return mem.small(sc.off_snodes + i); };
			auto leaf = [&](uint32_t start, uint32_t end) {
				for (uint32_t i = start; i < end; i++) {
					if (!((candMask >> i) & 1ull))
						continue; // the ray misses the object's grown box: Object::intersect returns false, info untouched
					const uint32_t slot = myBase + uint32_t(__popcll(candMask & ((1ull << i) - 1ull)));
					const uint32_t e = wl.queue[slot];
					const bool hit = (e & kTaskHit) != 0;
					const float t = wl.results[slot];
					if (e & kTaskBail) {
						ended = kEndBail;
						return true;
					}
					intersected |= hit;
					if (hit && t < info.t) { // scene.cu:432
						info.t = t;
						best = e;
					}
					if constexpr (SHADOW) {
						if (info.t + epsilon() < lightT) {
							ended = kEndOccluded;
							return true;
						}
					}
				}
				return false;
			};
			auto tcur = [&]() { return info.t; };
			const LevelColumn lc = {reinterpret_cast<float2 *>(col.scene), nullptr, 0u, 0u};
			bihTraverseFast(loadNode, sa, boxInterval(sa, smin, smax), lc, leaf, tcur);
		}
		if (groupHi >= 64u)
			break;
		groupBase = uint32_t(__shfl(incl, int(groupHi) - 1, 64));
		groupLo = groupHi;
	}
	ACR_DBG_PHASE(7)
	bail = bail || ended == kEndBail;
	if constexpr (SHADOW) {
		bailOut = bailOut || bail;
		return ended == kEndOccluded;
	} else {
		// the closest hit once more, this time for everything a shaded hit needs
		if (part && !bail && intersected && info.t < FLT_MAX) {
			// (step B has left the winning task's triangle and material in its list entry)
			const uint32_t won = best >> kTaskTriShift;
			objectHitDetails(mem, sc, (best >> 6) & 63u, won & ((1u << kTaskTriBits) - 1u), won >> kTaskTriBits, r, info);
		}
		ACR_DBG_PHASE(8)
		bailOut = bailOut || bail;
		return intersected;
	}
}

// scatterTrace (renderer.cu:326-388) for the 64 pixel slots of a wave together: renderSlot with every
// Scene::intersect made by the wave as a whole (waveSceneIntersect), hence in uniform control flow: all lanes
// walk the bounce levels and the lights in step, each with its own `alive` flag.
template <class Mem>
__device__ __forceinline__ SlotResult renderSlotWave(const RenderParams &p, const SceneHeader &sc, const Mem &mem, uint32_t vblock, uint32_t vtid)
{
	const uint32_t maxBounces = p.max_bounces;
	float *ldsF = reinterpret_cast<float *>(g_lds + p.lds_stack_slot);
	Columns col;
	col.scene = ldsF + 2 * threadIdx.x;
	col.mesh = col.scene + 2 * sc.scene_depth * kBlock;
	col.meshGlobLevels = 0u;
	col.meshGlobStride = 0u;
	col.meshGlob = nullptr;
	// this lane's direct-term column; its address is formed where it is used (see slotNow below)
	auto cstackNow = [&]() {
		uint32_t t = threadIdx.x;
		asm volatile("" : "+v"(t));
		return ldsF + p.lds_cstack_off + t;
	};
	WaveLds wl;
	{
		float *base = ldsF + p.lds_wave_off + (threadIdx.x >> 6) * waveLdsWords();
		wl.results = base;
		wl.queue = reinterpret_cast<uint32_t *>(wl.results + kWaveCap);
	}
#ifdef ACR_WAVE_DEBUG
	uint64_t dbgAcc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	wl.dbg = dbgAcc;
	struct DbgFlush
	{
		uint64_t *a;
		__device__ ~DbgFlush()
		{
			if ((threadIdx.x & 63u) == 0)
				for (int k = 0; k < 9; k++) atomicAdd(&g_waveDebug[k], (unsigned long long)a[k]);
		}
	} dbgFlush{dbgAcc};
#endif

	const uint32_t lane = vtid & 63u, wave = vtid >> 6;
	const uint32_t lx = (lane & 7u) + ((wave & 1u) << 3), ly = (lane >> 3) + ((wave >> 1) << 3);
	SlotResult res;
	res.tileNo = 0;
	res.bail = false;
	res.cnt = {0, 0, 0};
	uint32_t oldx, oldy;
	// Where this lane's pixel lies in the output buffers.  Kept as one 32-bit number, and the 64-bit addresses are
	// formed where they are used (behind an empty asm statement, or the compiler computes them up front and keeps
	// six registers busy through every traversal).
	uint32_t outSlot = 0;
	auto slotNow = [&]() {
		uint32_t v = outSlot;
		asm volatile("" : "+v"(v));
		return size_t(v);
	};
	const bool tiled = p.num_tiles != 0;
	if (tiled) {
		const uint32_t bpx = p.tile_w >> 4, bpt = bpx * (p.tile_h >> 4);
		const uint32_t tileNo = vblock / bpt, sub = vblock % bpt;
		const uint32_t tile = p.tile_ids[tileNo];
		res.tileNo = tileNo;
		const uint32_t inx = ((sub % bpx) << 4) + lx, iny = ((sub / bpx) << 4) + ly;
		oldx = (tile % p.tiles_x) * p.tile_w + inx;
		oldy = (tile / p.tiles_x) * p.tile_h + iny;
		outSlot = p.tiles_into_frame ? oldy * p.width + oldx : tileNo * p.tile_w * p.tile_h + iny * p.tile_w + inx; // below 2^32: the launch's slot count is, and so is the frame's
	} else {
		oldx = (vblock % p.blocks_x) * 16u + lx;
		oldy = (vblock / p.blocks_x) * (kSlots / 16u) + ly;
	}
	// ray counts of this lane in one register: shadow rays in bits 0-19, bounce rays in 20-27, the primary hit in 28
	uint32_t counts = 0;
	res.active = oldx < p.width && oldy < p.height; // renderer.cu:335-338
	bool alive = res.active;

	int index = 0, x = 0, y = 0;
	if (alive) {
		const int width = int(p.width);
		const int oldIndex = int(oldx) + width * int(oldy);
		if (p.frames == 0 || !p.pixel_values) { // renderer.cu:344-352
			if (p.pixel_values) p.pixel_values[oldIndex] = oldIndex;
			index = oldIndex;
			x = int(oldx);
			y = int(oldy);
		} else {
			index = p.pixel_values[oldIndex];
			x = index % width;
			y = index / width;
		}
		if (!tiled) outSlot = uint32_t(index);
	}
	RandState state;
	randInit(state, uint64_t(int64_t(index)) + p.seed); // renderer.cu:358
	const float i = 2.0f * (float(x) + randUniform(state)) * p.inv_width - 1.0f;
	const float j = 2.0f * (float(y) + randUniform(state)) * p.inv_height - 1.0f;
	Ray r;
	r.o = vec3(p.cam_position[0], p.cam_position[1], p.cam_position[2]);
	r.d = getPixelDir(p, i, -j);

	vec3 source(1, 1, 1);
	uint32_t kept = 0; // direct terms of this lane waiting in cstack
	bool bail = false;
	// A path's pixel is written on the level it ends on: Color4(c_0,1) + (Color4(c_1,1) + (... + tail)), the
	// recursion's association, then the running mean of renderer.cu:375-387.
	auto writePixel = [&](vec4 contribution) {
		float *cstack = cstackNow();
		while (kept > 0) {
			kept--;
			const vec4 cl(cstack[(3 * kept + 0) * kBlock], cstack[(3 * kept + 1) * kBlock], cstack[(3 * kept + 2) * kBlock], 1.0f);
			contribution = cl + contribution;
		}
		LateParams &q = lateParams();
		float4 *px = q.screen + slotNow();
		if (q.frames > 0) {
			const float4 old = *px;
			vec4 s(old.x, old.y, old.z, old.w);
			s = s * (float(q.frames) / float(q.frames + 1));
			s = s + (contribution / float(q.frames + 1));
			*px = make_float4(s.x, s.y, s.z, s.w);
		} else {
			*px = make_float4(contribution.x, contribution.y, contribution.z, contribution.w);
		}
	};
	if (alive) {
		if (p.primid) p.primid[slotNow()] = 0xFFFFFFFFu;
		if (p.hit_t) p.hit_t[slotNow()] = FLT_MAX;
	}
	for (uint32_t level = 0; level < maxBounces; level++) { // rayColor<0> (renderer.cu:318-324) is the initial `tail`
		if (!__any(alive))
			break;
		WorldHit info;
		info.t = FLT_MAX;
		info.material = 0;
		info.prim = 0xFFFFFFFFu;
		info.position = vec3(0, 0, 0);
		info.normal = vec3(0, 0, 0);
		if (alive && level > 0) counts += 1u << 20;
		// The primary rays of a wave's 8 x 8 pixels, and the shadow rays of their hits, travel together: each lane walks
		// for itself, as in renderSlot (its lanes reach the same objects at the same time anyway).  From the first bounce
		// on the rays of a wave have nothing in common and the object tests are handed out across it.
		const bool coherent = level == 0;
		bool hit = false;
		if (coherent) {
			if (alive) {
				hit = sceneIntersect<false, true>(mem, sc, r, info, 0.0f, col, bail, nullptr, 1.0f);
				if (bail) { // see renderSlot
					bail = false;
					info.t = FLT_MAX;
					hit = sceneIntersect<false, true, false, true>(mem, sc, r, info, 0.0f, col, bail, nullptr, 1.0f);
				}
			}
		} else {
			hit = waveSceneIntersect<false>(mem, sc, wl, alive && !hitsNothing(sc, r), r, info, 0.0f, col, bail);
		}
		bool ended = false;
		vec4 tail(0, 0, 0, 1);
		if (alive && (bail || !hit)) { // a miss: renderer.cu:315
			alive = false;
			ended = !bail;
		}
		if (alive && level == 0) {
			LateParams &q = lateParams();
			if (q.primid) q.primid[slotNow()] = info.prim;
			if (q.hit_t) q.hit_t[slotNow()] = info.t;
			counts |= 1u << 28;
		}
		// direct illum (renderer.cu:247-248): Scene::lightPoint (scene.cu:273-292), the lights in step across the wave
		vec3 lightSum(0, 0, 0);
		if (coherent) {
			if (alive) {
				Counters lc = {0, 0, 0};
				lightSum = lightPoint<true>(mem, sc, info.position, info.normal, state, lc, col, bail);
				counts += lc.shadow;
				if (bail) alive = false;
			}
		} else
		for (uint32_t li = 0; li < sc.num_lights; li++) {
			vec3 dir(0, 0, 1), c(0, 0, 0);
			float t = 0.0f;
			bool trace = false;
			if (alive) {
				const Light l = loadLight(mem, sc, li);
				trace = true;
				if (l.type == 0u) { // pointLightAccum, scene.cu:209-231
					dir = (l.position + 0.5f * randNorm(state)) - info.position;
					t = length(dir);
					dir = normalize(dir);
					const float cosTheta = max(dot(dir, info.normal), 0.0f);
					c = cosTheta / (l.attConstant + (l.attLinear + l.attQuadratic * t) * t) * l.diffuse;
					if (length(c) < epsilon()) trace = false;
				} else { // spotLightAccum, scene.cu:234-271
					dir = l.position - info.position;
					t = length(dir);
					dir = normalize(dir);
					const float cosTheta = max(dot(dir, info.normal), 0.0f);
					const float cosLight = dot(-dir, normalize(l.direction));
					const float lightTheta = acosf(cosLight);
					const float inner = l.outerCone - 1.57079632679f;
					const float outer = l.innerCone;
					const float falloff_distance = outer - inner;
					if (fabsf(lightTheta) >= outer / 2.0f) {
						c = vec3(0, 0, 0);
						trace = false;
					} else {
						c = cosTheta / (l.attConstant + (l.attLinear + l.attQuadratic * t) * t) * l.diffuse;
						if (length(c) < epsilon()) {
							trace = false;
						} else {
							const float val = (outer / 2.0f - fabsf(lightTheta));
							const float falloff = val / (falloff_distance / 2.0f);
							if (double(fabsf(lightTheta)) >= double(inner) / 2.0)
								c *= falloff;
						}
					}
				}
				if (trace) counts++;
			}
			Ray sr;
			sr.o = info.position;
			sr.d = dir;
			WorldHit sh;
			sh.t = t;
			if (waveSceneIntersect<true>(mem, sc, wl, trace, sr, sh, t, col, bail))
				c = vec3(0, 0, 0);
			if (alive && bail) alive = false;
			if (alive) lightSum += c;
		}
		if (alive) {
			const uint32_t msl = sc.off_materials + info.material * kMaterialSlots;
			const uint4 m0 = mem.small(msl), m1 = mem.small(msl + 1), m2 = mem.small(msl + 2);
			const vec3 ambient(asF(m0.x), asF(m0.y), asF(m0.z));
			const float refractiveIndex = asF(m0.w);
			const vec3 diffuse(asF(m1.x), asF(m1.y), asF(m1.z));
			const vec3 specular(asF(m2.x), asF(m2.y), asF(m2.z));
			const vec3 c = ambient + diffuse * lightSum;
			// indirect illum; nSource is uninitialised in the reference's translucent branch (renderer.cu:251,286-311):
			// defined as `source` here
			vec3 nSource = source;
			Ray nr;
			nr.o = info.position;
			nr.d = vec3(0, 0, 0);
			const bool last = level + 1 == maxBounces; // the next level returns (0,0,0,1) whatever the ray
			bool absorbed = false;
			if (refractiveIndex == 0) {
				const vec3 cd = source * diffuse;
				const vec3 cs = source * specular;
				const float Pd = compMax(cd) / compMax(source);
				const float Ps = compMax(cs) / compMax(source);
				const float P = randUniform(state);
				if (P < Pd) {
					if (!last) {
						nSource = cd * (1 / Pd);
						nr.d = randomHemi(info.normal, state);
					}
				} else if (P < Pd + Ps) {
					if (!last) {
						nSource = cs * (1 / Ps);
						nr.d = reflect(r.d, info.normal);
					}
				} else {
					absorbed = true; // renderer.cu:282
				}
			} else if (!last) {
				float n1, n2;
				vec3 norm = info.normal;
				if (dot(r.d, norm) < 0) {
					n1 = refractiveIndex;
					n2 = 1;
				} else {
					n1 = 1;
					n2 = refractiveIndex;
					norm *= -1.0f;
				}
				const float R = rSchlick2(r.d, norm, n1, n2);
				if (randUniform(state) >= R)
					nr.d = refract(r.d, norm, n1 / n2);
				else
					nr.d = reflect(r.d, norm);
			}
			if (absorbed) {
				tail = vec4(c, 1);
				alive = false;
				ended = true;
			} else if (last) { // Color4(c,1) + rayColor<0>() = Color4(c,1) + (0,0,0,1)
				tail = vec4(c, 1) + vec4(0, 0, 0, 1);
				alive = false;
				ended = true;
			} else {
				// Color4(c,1) + rayColor<N-1>(...): remember c, add after the deeper levels
				float *cstack = cstackNow();
				cstack[(3 * level + 0) * kBlock] = c.x;
				cstack[(3 * level + 1) * kBlock] = c.y;
				cstack[(3 * level + 2) * kBlock] = c.z;
				kept = level + 1;
				r = nr;
				source = nSource;
			}
		}
		if (ended) writePixel(tail);
	}
	if (alive) writePixel(vec4(0, 0, 0, 1)); // max_bounces == 0: rayColor<0> alone
	if (bail) {
		// nothing of this pixel is kept: the six-plane pass renders it from the start
		res.bail = true;
		return res;
	}
	res.cnt = {counts & 0xFFFFFu, (counts >> 20) & 0xFFu, counts >> 28};
	return res;
}

// Adds one wave's ray counts to the per-tile and per-launch counters.
// Position in the tile list of block `vblock` of a tiled launch (0 for a full-frame launch): what renderSlot computes
// at its start, once more for the kernel's last lines instead of kept in a register until then.
template <class Params>
__device__ __forceinline__ uint32_t tileOfBlock(const Params &p, uint32_t vblock)
{
	if (p.num_tiles == 0)
		return 0u;
	return vblock / ((p.tile_w >> 4) * (p.tile_h >> 4));
}

template <class Params>
__device__ __forceinline__ void reportCounts(const Params &p, const SlotResult &res, uint32_t vblock, uint32_t tileNo)
{
	const uint32_t lane = laneNow();
	const bool counted = res.active && !res.bail;
	if (p.num_tiles != 0 && p.tile_rays) {
		// a wave lies inside one 16x16 block, hence inside one tile
		const uint32_t rays = waveSum((counted ? 1u : 0u) + res.cnt.shadow + res.cnt.bounce);
		if (lane == 0 && rays) atomicAdd(&p.tile_rays[tileNo], rays);
	}
	if (p.counters) {
		const uint32_t act = waveSum(counted ? 1u : 0u);
		const uint32_t sh = waveSum(res.cnt.shadow), bo = waveSum(res.cnt.bounce), ph = waveSum(res.cnt.primaryHits);
		if (lane == 0 && (act | sh | bo | ph)) {
			unsigned long long *slot = p.counters + 4 * (vblock % kCounterSlots);
			atomicAdd(&slot[0], (unsigned long long)act);
			atomicAdd(&slot[1], (unsigned long long)sh);
			atomicAdd(&slot[2], (unsigned long long)bo);
			atomicAdd(&slot[3], (unsigned long long)ph);
		}
	}
}

// One work-item per pixel slot of the frame (or of the listed tiles).
// FAST: interval traversal; pixels it cannot serve are appended to p.redo_list for renderListKernel.
template <bool SMALL_LDS, bool LARGE_LDS, bool FAST, bool KEYS, bool ONE = false>
__global__ void __launch_bounds__(kBlock, LARGE_LDS ? ACR_RENDER_WAVES : ACR_RENDER_WAVES_HBM) renderKernel(const RenderParams p)
{
	const SceneHeader &sc = p.scene;
	SceneMem<SMALL_LDS, LARGE_LDS> mem;
	mem.g = p.blob;
	if constexpr (LARGE_LDS) stageScene(p.blob, sc.total_slots);
	else if constexpr (SMALL_LDS) stageScene(p.blob, sc.small_slots);

	const bool tiled = p.num_tiles != 0;
	const uint64_t clock0 = ((tiled && p.tile_cycles) || p.block_cost) ? __builtin_readcyclecounter() : 0;
	// this workgroup's part of its block of pixel slots
	const uint32_t firstSlot = blockIdx.x * uint32_t(kBlock);
	const uint32_t gblock = firstSlot / kSlots, vtid = firstSlot % kSlots + threadIdx.x;
	// Workgroups start in grid order, so the order table decides which blocks start first: with the expensive ones
	// first the launch does not end on a few long workgroups while the rest of the GPU is idle.
	const uint32_t vblock = p.block_order ? p.block_order[gblock] : gblock;
	const SlotResult res = renderSlot<FAST, KEYS, ONE>(p, mem, vblock, vtid);
	LateParams &q = lateParams();
	if constexpr (FAST) {
		if (res.bail) {
			const uint32_t at = atomicAdd(q.redo_count, 1u);
			q.redo_list[at] = vblock * kSlots + vtid;
		}
	}
	const bool firstLane = laneNow() == 0;
	const uint32_t tileNo = tileOfBlock(q, vblock);
	if (q.num_tiles != 0 && q.tile_cycles && firstLane)
		atomicAdd(&q.tile_cycles[tileNo], uint32_t((__builtin_readcyclecounter() - clock0) >> 6));
	if (q.block_cost && firstLane)
		atomicAdd(&q.block_cost[vblock], uint32_t((__builtin_readcyclecounter() - clock0) >> 6));
	reportCounts(q, res, vblock, tileNo);
}

// ---- one deep mesh in HBM: the lanes of a wave share their walks ------------------------------------------------
//
// A ray's walk through a deep mesh tree in HBM is a chain of dependent reads, and the walks of neighbouring pixels
// differ widely in length (10 M-triangle soup: 14 rounds on average, one ray in 200 more than 128, and a wave lasts as
// long as the longest of its 64 walks: on average 29 % of the lanes are still walking in a round).  Here a lane without
// a walk takes over a waiting child of a lane that has some: the subtree below it is walked by the helper with its own
// waiting-child column, and what it finds is handed to the ray's lane when the helper is through.
//
// What a ray ends up with is what bihTraverseNear ends up with.  The nearest hit W (ties to the lower item number) does
// not depend on who visits which subtree, as long as a subtree is only dropped when it starts more than kNearPruneSlack
// behind a hit of this ray; every walker drops against its own hits and the nearest one any walker of the ray has
// reported (one word per ray in LDS).  The order check
// (OrderCheck) needs the nearest hit that lost: every hit any of the walkers saw and that is not W or an exact tie with
// it.  Walkers that know less drop less and see more, so that distance can only come out nearer than in the walk by one
// lane, which lists a pixel more, never one fewer; and a hit X with t_X <= tmin(W's leaf) lies in no subtree anybody
// drops (its boxes start in front of every hit of the ray).
//
// Measured on the 10 M-triangle soup at 1920 x 1080 (one lane per ray: 2.37 ms per frame), each with the others at
// their defaults: frames are bit-identical for every setting.
#ifndef ACR_SOLO_MAX_PASSES
#define ACR_SOLO_MAX_PASSES 3 // steps of two levels a lane goes down in one round (0: down to a leaf); 1.19 ms with 2, 1.09 with 3, 1.10 with 4
#endif
#ifndef ACR_SOLO_TRIS_PER_TRIP
#define ACR_SOLO_TRIS_PER_TRIP 4 // ACR_TRIS_PER_TRIP of this kernel: with four it fits the registers of four waves per SIMD (1.12 ms with 3 and with 5)
#endif
#ifndef ACR_MULTI_WAVES
#define ACR_MULTI_WAVES 3 // the same for the form for several objects (168 VGPRs and 84 bytes of scratch per lane; with two waves and no scratch a fifth slower)
#endif
#ifndef ACR_SOLO_WAVES
#define ACR_SOLO_WAVES 4 // waves per SIMD the kernel is compiled for: 1.21 ms with three (and 52 KB of LDS columns), 1.12 with four (40 KB)
#endif
#ifndef ACR_SOLO_GIVE_REPS
#define ACR_SOLO_GIVE_REPS 2 // waiting children a lane may hand over between two rounds: 1.10 ms with 1, 1.09 with 2 and with 3
#endif
#ifndef ACR_SOLO_MIN_IDLE
#define ACR_SOLO_MIN_IDLE 8 // lanes without work before waiting children are handed over: 1.12 / 1.11 / 1.09 / 1.12 / 1.19 ms with 2 / 4 / 8 / 16 / 24
#endif

// position of the r-th set bit of mask (r < number of set bits)
__device__ __forceinline__ uint32_t nthSetBit(uint64_t mask, uint32_t r)
{
	const uint32_t lo = uint32_t(mask), hi = uint32_t(mask >> 32);
	const uint32_t c = uint32_t(__popc(lo));
	uint32_t base = 0, m = lo;
	if (r >= c) {
		r -= c;
		base = 32;
		m = hi;
	}
#pragma unroll
	for (uint32_t s = 16; s >= 1; s >>= 1) {
		const uint32_t low = m & ((1u << s) - 1u);
		const uint32_t c2 = uint32_t(__popc(low));
		if (r >= c2) {
			r -= c2;
			base += s;
			m >>= s;
		} else {
			m = low;
		}
	}
	return base;
}

// what the walks of one ray have found so far (LocalHit and OrderCheck of meshIntersect)
struct SoloFound
{
	float t;
	vec3 bary;
	uint32_t tri;
	bool any;
	float second, leafTmin;
};

// b's hits seen after a's, by the rules leafTriangles<NEAR> applies hit by hit
__device__ __forceinline__ void soloMerge(SoloFound &a, float t, const vec3 &bary, uint32_t tri, float second, float leafTmin)
{
	bool take = t < a.t;
	const bool tie = t == a.t;
	float s = second < a.second ? second : a.second;
	if (take) s = a.t < s ? a.t : s;
	else if (!tie && t < s) s = t;
	take = take || (tie && a.any && tri < a.tri);
	a.second = s;
	if (take) {
		a.t = t;
		a.bary = bary;
		a.tri = tri;
		a.leafTmin = leafTmin;
		a.any = true;
	}
}

__device__ __forceinline__ float shflF(float v, uint32_t src) { return __shfl(v, int(src), 64); }

// Mesh::intersect (geometry.cu:52-61) below the root box test for the rays of a wave, called by all 64 lanes together.
// wants: this lane has a ray (lr, sa; root: its interval in the mesh's box, which it hits).  R: what the ray found
// (R.any: something nearer than tStart).
template <class Mem>
__device__ __forceinline__ void soloWalk(const Mem &mem, uint32_t nodeBase, uint32_t nodeCount, uint32_t triBase, uint32_t triFirst, const LevelColumn &stack,
                                         bool wants, const Ray &lr, const SlabArgs &sa, const Interval &root, SoloFound &R, uint32_t *nearest,
                                         float tStart = FLT_MAX)
{
	const uint32_t lane = laneNow();
	const uint64_t below = (1ull << lane) - 1ull;
	R.t = tStart; // tmpInfo.t as Mesh::intersect finds it: an earlier mesh of the object may have left a hit (scene.cu:418-424)
	R.bary = vec3(0, 0, 0);
	R.tri = 0;
	R.any = false;
	R.second = FLT_MAX;
	R.leafTmin = 0.0f;
	// the walk in hand: whose ray, the ray, where the walk is, what it has found
	bool working = wants;
	uint32_t owner = lane;
	Ray jr = lr;
	vec3 jinv = sa.invD;
	NearWalk w = {0u, 0u, 0u, root.tmin, root.tmax};
	LocalHit jt;
	jt.t = FLT_MAX;
	jt.tri = 0;
	jt.material = 0;
	jt.bary = vec3(0, 0, 0);
	bool jany = false;
	OrderCheck jorder = {FLT_MAX, 0.0f};
#ifdef ACR_WALK_DEBUG
	unsigned long long dbg[8] = {wants ? 1ull : 0ull, 0, 0, 0, 0, 0, 0, 0};
#endif
	waveSync();
	nearest[lane] = __float_as_uint(tStart);
	waveSync();
	for (;;) {
		uint64_t busy = __ballot(working);
		if (busy == 0)
			break;
		// the nearest hit any walker of the ray has found so far (nearest[]: one word per lane of the wave, in LDS;
		// hit distances are positive, and positive floats order as their bit patterns do)
		float known = __uint_as_float(nearest[owner]);
		for (uint32_t rep = 0; rep < uint32_t(ACR_SOLO_GIVE_REPS); rep++) {
			if (rep != 0) busy = __ballot(working);
			const uint32_t nIdle = 64u - uint32_t(__popcll(busy));
			if (nIdle < uint32_t(ACR_SOLO_MIN_IDLE))
				break;
			const bool canGive = working && w.pending != 0u;
			const uint64_t donors = __ballot(canGive);
			const uint32_t nd = min(uint32_t(__popcll(donors)), nIdle);
			if (nd == 0u)
				break;
			// the k-th lane with waiting children gives one to the k-th lane without work: the deepest one, which the
			// giver would have come to next and which is nearest the ray's origin (giving the one nearest the root, the
			// largest piece and the one least likely to be needed, costs a fifth more rounds: 1.70 against 1.44 ms
			// when both were measured)
			float gmin = 0.0f, gmax = 0.0f;
			uint32_t gidx = 0xFFFFFFFFu, glevel = 0u;
			if (canGive && uint32_t(__popcll(donors & below)) < nd) {
				const uint32_t l = 31u - uint32_t(__clz(int(w.pending)));
				w.pending &= ~(1u << l);
				const float2 b = stack.load(l);
				const float near = jt.t < known ? jt.t : known;
				if (b.x < near * kNearPruneSlack) { // else nobody needs it any more (popNear)
					gmin = b.x;
					gmax = b.y;
					gidx = (((w.idx + 1u) >> (w.level - l - 1u)) ^ 1u) - 1u; // the sibling of this path's node on level l + 1
					glevel = l + 1u;
				}
			}
			const uint32_t rank = uint32_t(__popcll(~busy & below));
			const bool takes = !working && rank < nd;
			const uint32_t src = takes ? nthSetBit(donors, rank) : lane;
			const uint32_t ridx = uint32_t(__shfl(int(gidx), int(src), 64));
			const uint32_t rlevel = uint32_t(__shfl(int(glevel), int(src), 64));
			const float rmin = shflF(gmin, src), rmax = shflF(gmax, src);
			const uint32_t rowner = uint32_t(__shfl(int(owner), int(src), 64));
			const float rknown = shflF(known, src);
			const vec3 ro(shflF(jr.o.x, src), shflF(jr.o.y, src), shflF(jr.o.z, src));
			const vec3 rd(shflF(jr.d.x, src), shflF(jr.d.y, src), shflF(jr.d.z, src));
			const vec3 ri(shflF(jinv.x, src), shflF(jinv.y, src), shflF(jinv.z, src));
			if (takes && ridx != 0xFFFFFFFFu) {
				working = true;
				owner = rowner;
				known = rknown;
				jr.o = ro;
				jr.d = rd;
				jinv = ri;
				w = {ridx, rlevel, 0u, rmin, rmax};
				jt.t = FLT_MAX;
				jt.tri = 0;
				jt.bary = vec3(0, 0, 0);
				jany = false;
				jorder = {FLT_MAX, 0.0f};
			}
		}
		const bool worked = working;
		if (working) {
			SlabArgs jsa;
			jsa.o = jr.o;
			jsa.invD = jinv;
			auto loadNode = [&](uint32_t i) { return mem.large(nodeBase + i); };
			auto loadRun = [&](uint32_t i, auto &out) { mem.largeRun(nodeBase + i, out); };
			auto tslack = [&]() { return (jt.t < known ? jt.t : known) * kNearPruneSlack; };
			// One round: down to a leaf, the leaf's triangles, and up to the deepest waiting child that can still hold a
			// nearer hit.  A lane that is still on its way down after ACR_SOLO_MAX_PASSES memory round trips goes on in
			// the next round instead of keeping the rest of the wave waiting in this one.
#ifdef ACR_WALK_DEBUG
			walkDebugAdd(dbg, 6, 7);
#endif
			uint4 n;
#ifdef ACR_WALK_DEBUG
			const uint32_t how = nearDescend<ACR_SOLO_MAX_PASSES>(loadNode, loadRun, nodeCount, jsa, stack, tslack, w, n, dbg);
#else
			const uint32_t how = nearDescend<ACR_SOLO_MAX_PASSES>(loadNode, loadRun, nodeCount, jsa, stack, tslack, w, n);
#endif
			if (how == 0u) leafTriangles<true, true, ACR_SOLO_TRIS_PER_TRIP>(mem, triBase, triFirst, n.z, n.w & kNodeEndMask, jr, jt, jany, &jorder, w.tminB);
			if (how != 2u) working = popNear(stack, w.pending, w.idx, w.level, w.tminB, w.tmaxB, tslack());
		}
		if (worked && jany) atomicMin(&nearest[owner], __float_as_uint(jt.t));
		// a walk that is through hands what it found to the ray's lane
		const bool through = worked && !working;
		if (through && owner == lane && jany) soloMerge(R, jt.t, jt.bary, jt.tri, jorder.second, jorder.leafTmin);
		uint64_t m = __ballot(through && owner != lane && jany);
		while (m != 0) {
			const int l = __ffsll((unsigned long long)m) - 1;
			m &= m - 1ull;
			const uint32_t o = uint32_t(__builtin_amdgcn_readlane(int(owner), l));
			const float ht = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(jt.t), l));
			const vec3 hb(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(jt.bary.x), l)),
			              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(jt.bary.y), l)),
			              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(jt.bary.z), l)));
			const uint32_t htri = uint32_t(__builtin_amdgcn_readlane(int(jt.tri), l));
			const float hs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(jorder.second), l));
			const float hl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(jorder.leafTmin), l));
			if (lane == o) soloMerge(R, ht, hb, htri, hs, hl);
		}
		if (through) owner = lane;
	}
#ifdef ACR_WALK_DEBUG
	walkDebugFlush(dbg, false);
#endif
}

// renderSlot<true, false, ONE> for a scene of one object with one mesh whose tree lives in HBM (the tree over objects is
// then a single leaf, which Scene::intersect visits without a test: bih.h:209-299 never tests the root box): what a
// pixel's rays compute is renderSlot's, operation for operation, but the 64 lanes of a wave go through it in step and
// walk the mesh tree together (soloWalk).  Every lane of the wave makes every call, with or without a ray.
template <bool ONE, bool MULTI, class Mem>
__device__ __forceinline__ SlotResult renderSlotSolo(const RenderParams &p, const Mem &mem, uint32_t vblock, uint32_t vtid)
{
	const SceneHeader &sc = p.scene;
	float *ldsF = reinterpret_cast<float *>(g_lds + p.lds_stack_slot);
	// this lane's waiting-child column of the mesh tree, where renderSlot has it
	const LevelColumn stack = {reinterpret_cast<float2 *>(ldsF + 2 * threadIdx.x + 2 * sc.scene_depth * kBlock),
	                           p.mesh_glob + (size_t(blockIdx.x) * kBlock + threadIdx.x), p.mesh_glob_levels, p.mesh_glob_stride};

	// One word per lane of this wave for soloWalk: in the column of the tree over objects where nothing walks that
	// tree (one object), behind the columns otherwise (the host has made room: planLds).
	uint32_t *nearest = reinterpret_cast<uint32_t *>(MULTI ? ldsF + p.lds_wave_off : ldsF) + (threadIdx.x & ~63u);
	// the one object and its one mesh (records are read where they are used; these few numbers stay)
	const uint32_t os = sc.off_objects;
	const uint32_t meshId = mem.small(os + 1).w;
	uint32_t nodeBase, nodeCount, triBase, triFirst;
	{
		const MeshRecord mr = loadMeshRecord(mem, sc, meshId);
		nodeBase = __builtin_amdgcn_readfirstlane(sc.off_mnodes + mr.q1.w);
		nodeCount = __builtin_amdgcn_readfirstlane((1u << mr.q2.y) - 1u);
		triFirst = __builtin_amdgcn_readfirstlane(mr.q2.x);
		triBase = sc.off_tris + triFirst * kTriSlots;
	}

	// which pixel slot: 16x16 pixels per workgroup, 8x8 per wave
	const uint32_t lane = vtid & 63u, wave = vtid >> 6;
	const uint32_t lx = (lane & 7u) + ((wave & 1u) << 3), ly = (lane >> 3) + ((wave >> 1) << 3);
	uint32_t oldx, oldy;
	size_t outSlot;
	SlotResult res;
	res.tileNo = 0;
	res.bail = false;
	res.cnt = {0, 0, 0};
	const bool tiled = p.num_tiles != 0;
	if (tiled) {
		const uint32_t bpx = p.tile_w >> 4, bpt = bpx * (p.tile_h >> 4);
		const uint32_t tileNo = vblock / bpt, sub = vblock % bpt;
		const uint32_t tile = p.tile_ids[tileNo];
		res.tileNo = tileNo;
		const uint32_t inx = ((sub % bpx) << 4) + lx, iny = ((sub / bpx) << 4) + ly;
		oldx = (tile % p.tiles_x) * p.tile_w + inx;
		oldy = (tile / p.tiles_x) * p.tile_h + iny;
		outSlot = p.tiles_into_frame ? size_t(oldy) * p.width + oldx : size_t(tileNo) * p.tile_w * p.tile_h + size_t(iny) * p.tile_w + inx;
	} else {
		oldx = (vblock % p.blocks_x) * 16u + lx;
		oldy = (vblock / p.blocks_x) * (kSlots / 16u) + ly;
		outSlot = 0;
	}
	res.active = oldx < p.width && oldy < p.height; // renderer.cu:335-338

	Ray lr; // the ray in hand, in the object's space
	lr.o = vec3(0, 0, 0);
	lr.d = vec3(0, 0, 1);
	SlabArgs sa;
	sa.o = lr.o;
	sa.invD = vec3(1, 1, 1);
	Interval root = {0.0f, 0.0f};
	// Object::intersect (scene.cu:407-440) up to the walk of its mesh tree, as objectIntersect and meshIntersect make it
	// for world ray r: 0 = the object cannot be hit (`false`, info untouched), 1 = walk the tree, 2 = the interval form
	// cannot serve the ray.  `parallel`: the primary ray's second way in (parallelRayQualifies).
	auto enterObject = [&](const Ray &r, const SlabArgs &saWorld, float tLimit, bool parallel) -> uint32_t {
		const uint4 b0 = mem.small(os), b1 = mem.small(os + 1);
		const uint32_t flags = b0.w;
		if (flags & kObjPrefilter) {
			const vec3 wmin(asF(b0.x), asF(b0.y), asF(b0.z)), wmax(asF(b1.x), asF(b1.y), asF(b1.z));
			const Interval wb = boxInterval(saWorld, wmin, wmax);
			if (!intervalHit(wb.tmin, wb.tmax, tLimit))
				return 0u;
		}
		{
			uint32_t meshBegin, meshCount;
			const mat4 invM = loadMat4(mem, os + 2);
			const mat3 invN = loadMat3(mem, os + 6, meshBegin, meshCount);
			lr.o = (flags & kObjInvAffine) ? translateAffine(invM, r.o) : translate(invM, r.o);
			lr.d = translaten(invN, r.d);
		}
		if (parallel || !sa.initInterval(lr)) {
			if (!parallel || !parallelRayQualifies(sa, lr, 0u))
				return 2u;
		}
		const MeshRecord mr = loadMeshRecord(mem, sc, meshId);
		const vec3 bmin(asF(mr.q0.x), asF(mr.q0.y), asF(mr.q0.z)), bmax(asF(mr.q1.x), asF(mr.q1.y), asF(mr.q1.z));
		root = boxInterval(sa, bmin, bmax);
		if (!intervalHit(root.tmin, root.tmax, FLT_MAX))
			return 0u;
		return 1u;
	};
	// the hit of a finished walk in world space (scene.cu:426-431): position and distance from the world ray's origin
	auto worldHit = [&](const vec3 &origin, float tLocal, vec3 &pw) -> float {
		const uint32_t flags = mem.small(os).w;
		const vec3 pl = lr.o + lr.d * tLocal;
		const mat4 M = loadMat4(mem, os + 9);
		pw = (flags & kObjFwdAffine) ? translateAffine(M, pl) : translate(M, pl);
		return length(pw - origin);
	};

	bool listed = false; // the list pass renders the pixel from the start

	// MULTI: Scene::intersect (scene.cu:204-207) for a scene of several objects.  Every lane walks the tree over objects
	// for its own ray, in the reference's order and with info.t as it stands (bihTraverseFast, cut where it reaches an
	// object); whenever lanes stand before an object, the wave takes the objects asked for one after the other, and for
	// each of them the lanes that asked go through Object::intersect together: the object's meshes one after the other,
	// every mesh tree walked by the whole wave (soloWalk).  Which objects a ray visits, in which order and with which
	// info.t is then the reference's, which matters: a bounce ray's direction is not normalised, and the reference compares
	// its slab distances with world distances all the same.  A ray the interval form cannot serve sends its pixel to the
	// list pass (no second way in for primary rays here).
	struct Best
	{
		float t;       // info.t
		bool anyHit;   // `intersected`
		bool replaced; // info holds a hit: (obj, tri, material)
		uint32_t obj, tri, material;
	};
	constexpr uint32_t kNoObject = 0xFFFFFFFFu;
	auto traceObjects = [&](bool &active, const Ray &r, bool shadow, float lightT, float distPerT, bool continuation, Best &best) {
		best.t = shadow ? lightT : FLT_MAX;
		best.anyHit = false;
		best.replaced = false;
		best.obj = best.tri = best.material = 0u;
		SlabArgs saWorld;
		saWorld.o = vec3(0, 0, 0);
		saWorld.invD = vec3(1, 1, 1);
		// where this lane's walk of the tree over objects stands
		uint32_t sIdx = 0, sLevel = 0, sPending = 0, item = 0, itemEnd = 0;
		float sMin = 0.0f, sMax = 0.0f;
		bool needPop = false, walking = false;
		if (active && !(continuation && hitsNothing(sc, r))) {
			if (!saWorld.initInterval(r)) {
#ifdef ACR_WALK_DEBUG
				atomicAdd(&g_walkDebug[12], 1ull);
				if (__builtin_isnan(r.d.x) || __builtin_isnan(r.d.y) || __builtin_isnan(r.d.z)) atomicAdd(&g_walkDebug[15], 1ull);
#endif
				listed = true;
				active = false;
			} else {
				const vec3 smin(sc.scene_min[0], sc.scene_min[1], sc.scene_min[2]), smax(sc.scene_max[0], sc.scene_max[1], sc.scene_max[2]);
				const Interval box = boxInterval(saWorld, smin, smax);
				sMin = box.tmin;
				sMax = box.tmax;
				walking = true;
			}
		}
		const LevelColumn lc = {reinterpret_cast<float2 *>(ldsF + 2 * threadIdx.x), nullptr, 0u, 0u};
		auto loadSceneNode = [&](uint32_t i) { return mem.small(sc.off_snodes + i); };
		// bihTraverseFast up to the next object it hands to its leaf function (kNoObject: the walk is over)
		auto nextObject = [&]() -> uint32_t {
			for (;;) {
				if (item < itemEnd)
					return item++;
				if (needPop) {
					if (sPending == 0)
						return kNoObject;
					const uint32_t l = 31u - uint32_t(__clz(int(sPending)));
					sPending &= ~(1u << l);
					const float2 b = lc.load(l);
					sMin = b.x;
					sMax = b.y;
					sIdx = 2 * (((sIdx + 1) >> (sLevel - l)) - 1) + 2;
					sLevel = l + 1;
					needPop = false;
				}
				uint4 n = loadSceneNode(sIdx);
				bool over = false;
				while ((n.w & kNodeLeafBit) == 0) {
					const uint4 childL = loadSceneNode(2 * sIdx + 1), childR = loadSceneNode(2 * sIdx + 2);
					const uint32_t axis = (n.w >> kNodeAxisShift) & 3u;
					const float oa = saWorld.o[int(axis)], ia = saWorld.invD[int(axis)];
					const bool sg = ia < 0;
					const float tl = (asF(n.x) - oa) * ia;
					const float tr = (asF(n.y) - oa) * ia;
					const float tCutFar = sg ? tr : tl, tCutNear = sg ? tl : tr;
					const float nearSideMax = tCutFar < sMax ? tCutFar : sMax;
					const float farSideMin = tCutNear > sMin ? tCutNear : sMin;
					const bool hitNearSide = intervalHit(sMin, nearSideMax, best.t);
					const bool hitFarSide = intervalHit(farSideMin, sMax, best.t);
					const bool hitL = sg ? hitFarSide : hitNearSide;
					const bool hitR = sg ? hitNearSide : hitFarSide;
					if (hitL || hitR) {
						const bool takeFarSide = hitL ? sg : !sg;
						if (hitL && hitR) {
							sPending |= 1u << sLevel;
							lc.store(sLevel, takeFarSide ? make_float2(sMin, nearSideMax) : make_float2(farSideMin, sMax));
						}
						if (takeFarSide) sMin = farSideMin;
						else sMax = nearSideMax;
						sIdx = 2 * sIdx + (hitL ? 1 : 2);
						sLevel++;
						n = hitL ? childL : childR;
					} else {
						if (sPending == 0) {
							over = true;
							break;
						}
						const uint32_t l = 31u - uint32_t(__clz(int(sPending)));
						sPending &= ~(1u << l);
						const float2 b = lc.load(l);
						sMin = b.x;
						sMax = b.y;
						sIdx = 2 * (((sIdx + 1) >> (sLevel - l)) - 1) + 2;
						sLevel = l + 1;
						n = loadSceneNode(sIdx);
					}
				}
				if (over)
					return kNoObject;
				item = n.z;
				itemEnd = n.w & kNodeEndMask;
				needPop = true;
			}
		};
		for (;;) {
			uint32_t want = kNoObject;
			if (walking) {
				want = nextObject();
				if (want == kNoObject) walking = false;
			}
			uint64_t asking = __ballot(want != kNoObject);
			if (asking == 0)
				break;
			while (asking != 0) {
				const uint32_t obj = uint32_t(__builtin_amdgcn_readlane(int(want), __ffsll((unsigned long long)asking) - 1));
				const bool mine = want == obj;
				asking &= ~__ballot(mine);
				const uint32_t objSlot = sc.off_objects + obj * kObjectSlots;
				bool enter = false;
				if (mine) {
					// see objectIntersect for the limit
					const float tLimit = (shadow || best.anyHit) ? best.t * distPerT : FLT_MAX;
					const uint4 b0 = mem.small(objSlot), b1 = mem.small(objSlot + 1);
					const uint32_t flags = b0.w;
					enter = true;
					if (flags & kObjPrefilter) {
						const vec3 wmin(asF(b0.x), asF(b0.y), asF(b0.z)), wmax(asF(b1.x), asF(b1.y), asF(b1.z));
						const Interval wb = boxInterval(saWorld, wmin, wmax);
						if (!intervalHit(wb.tmin, wb.tmax, tLimit)) enter = false;
					}
					if (enter) {
						uint32_t u0, u1;
						const mat4 invM = loadMat4(mem, objSlot + 2);
						const mat3 invN = loadMat3(mem, objSlot + 6, u0, u1);
						lr.o = (flags & kObjInvAffine) ? translateAffine(invM, r.o) : translate(invM, r.o);
						lr.d = translaten(invN, r.d);
						if (!sa.initInterval(lr)) {
#ifdef ACR_WALK_DEBUG
							atomicAdd(&g_walkDebug[13], 1ull);
#endif
							listed = true;
							active = walking = false;
							enter = false;
						}
					}
				}
				// the object's meshes (the same for every lane: scalar registers)
				const uint32_t firstMesh = __builtin_amdgcn_readfirstlane(mem.small(objSlot + 1).w);
				const uint4 counts = mem.small(objSlot + 8);
				const uint32_t meshBegin = __builtin_amdgcn_readfirstlane(counts.y), meshCount = __builtin_amdgcn_readfirstlane(counts.z);
				// tmpInfo of scene.cu:418-424
				float objT = FLT_MAX;
				uint32_t objTri = 0u, objMaterial = 0u;
				bool objAny = false;
				for (uint32_t k = 0; k < meshCount; k++) {
					uint32_t meshNo = firstMesh;
					if (k != 0) {
						const uint4 ids = mem.small(sc.off_objmesh + ((meshBegin + k) >> 2));
						const uint32_t sel = (meshBegin + k) & 3u;
						meshNo = __builtin_amdgcn_readfirstlane(sel == 0 ? ids.x : (sel == 1 ? ids.y : (sel == 2 ? ids.z : ids.w)));
					}
					const MeshRecord mr = loadMeshRecord(mem, sc, meshNo);
					const uint32_t mNodeBase = __builtin_amdgcn_readfirstlane(sc.off_mnodes + mr.q1.w);
					const uint32_t mNodeCount = __builtin_amdgcn_readfirstlane((1u << mr.q2.y) - 1u);
					const uint32_t mTriFirst = __builtin_amdgcn_readfirstlane(mr.q2.x);
					bool walks = enter;
					if (walks) {
						const vec3 bmin(asF(mr.q0.x), asF(mr.q0.y), asF(mr.q0.z)), bmax(asF(mr.q1.x), asF(mr.q1.y), asF(mr.q1.z));
						root = boxInterval(sa, bmin, bmax);
						walks = intervalHit(root.tmin, root.tmax, objT); // geometry.cu:55
					}
					SoloFound R;
					soloWalk(mem, mNodeBase, mNodeCount, sc.off_tris + mTriFirst * kTriSlots, mTriFirst, stack, walks, lr, sa, root, R, nearest, objT);
					if (walks) {
						if (R.second < FLT_MAX && R.second <= R.leafTmin) { // the visiting order could have mattered
#ifdef ACR_WALK_DEBUG
							atomicAdd(&g_walkDebug[14], 1ull);
#endif
							listed = true;
							active = walking = false;
							enter = false;
						} else if (R.any) {
							objT = R.t;
							objTri = R.tri;
							objMaterial = mr.q0.w;
							objAny = true;
						}
					}
				}
				if (enter && objAny) {
					// info.point.position in local space (geometry.h:140), to world (scene.cu:428), its distance (scene.cu:431)
					const uint32_t flags = mem.small(objSlot).w;
					const vec3 pl = lr.o + lr.d * objT;
					const mat4 M = loadMat4(mem, objSlot + 9);
					const vec3 pw = (flags & kObjFwdAffine) ? translateAffine(M, pl) : translate(M, pl);
					const float t = length(pw - r.o);
					if (t < best.t) { // scene.cu:432
						best.t = t;
						best.replaced = true;
						best.obj = obj;
						best.tri = objTri;
						best.material = objMaterial;
					}
					best.anyHit = true;
					if (shadow && best.t + epsilon() < lightT) walking = false; // scene.cu:226,266: the answer is fixed
				}
			}
		}
	};

	// ---- scatterTrace up to the primary ray (renderer.cu:326-372) ----
	const uint32_t maxBounces = ONE ? 1u : p.max_bounces;
	float *cstack = ldsF + p.lds_cstack_off + threadIdx.x;
	bool alive = res.active; // the pixel's path goes on
	int index = 0;
	RandState state;
	state.v0 = state.v1 = state.v2 = state.v3 = state.v4 = state.d = 0;
	Ray r; // the path's ray at this level, in world space
	r.o = vec3(p.cam_position[0], p.cam_position[1], p.cam_position[2]);
	r.d = vec3(0, 0, 1);
	vec3 source(1, 1, 1);
	vec4 tail(0, 0, 0, 1); // what the deepest level returns; a miss: renderer.cu:315
	uint32_t kept = 0;     // levels whose direct term waits in cstack
	if (alive) {
		const int width = int(p.width);
		const int oldIndex = int(oldx) + width * int(oldy);
		int x, y;
		if (p.frames == 0 || !p.pixel_values) { // renderer.cu:344-352
			if (p.pixel_values) p.pixel_values[oldIndex] = oldIndex;
			index = oldIndex;
			x = int(oldx);
			y = int(oldy);
		} else {
			index = p.pixel_values[oldIndex];
			x = index % width;
			y = index / width;
		}
		if (!tiled) outSlot = size_t(index);
		randInit(state, uint64_t(int64_t(index)) + p.seed); // renderer.cu:358
		const float i = 2.0f * (float(x) + randUniform(state)) * p.inv_width - 1.0f;
		const float j = 2.0f * (float(y) + randUniform(state)) * p.inv_height - 1.0f;
		r.d = getPixelDir(p, i, -j);
		if (p.primid) p.primid[outSlot] = 0xFFFFFFFFu;
		if (p.hit_t) p.hit_t[outSlot] = FLT_MAX;
	}

	// ---- rayColor<max_bounces> (renderer.cu:232-324), one level after the other for the whole wave ----
	for (uint32_t level = 0; level < maxBounces; level++) {
		if (!ONE && __ballot(alive) == 0)
			break;
		vec3 position(0, 0, 0), normal(0, 0, 0), lightSum(0, 0, 0);
		uint32_t material = 0;
		SoloFound R;
		if constexpr (MULTI) {
			if (alive && level > 0) res.cnt.bounce++;
			Best best;
			// slab distances per unit of world distance: 1 for the normalised primary ray, 1 / |d| for a bounce ray
			traceObjects(alive, r, false, 0.0f, level == 0 ? 1.0f : 1.0f / length(r.d), level > 0, best);
			if (alive && !best.anyHit) alive = false; // a miss: renderer.cu:315
			if (alive) {
				WorldHit info;
				info.t = FLT_MAX;
				info.material = 0;
				info.prim = 0xFFFFFFFFu;
				info.position = vec3(0, 0, 0);
				info.normal = vec3(0, 0, 0);
				if (best.replaced) objectHitDetails(mem, sc, best.obj, best.tri, best.material, r, info);
				position = info.position;
				normal = info.normal;
				material = info.material;
				if (level == 0) {
					LateParams &q = lateParams();
					if (q.primid) q.primid[outSlot] = info.prim;
					if (q.hit_t) q.hit_t[outSlot] = info.t;
					res.cnt.primaryHits++;
				}
			}
		} else {
		uint32_t enter = 0u;
		if (alive) {
			if (level > 0) res.cnt.bounce++;
			// Scene::intersect: the tree over objects is one leaf holding the one object
			SlabArgs saWorld;
			enter = 2u;
			if (!ONE && level > 0 && hitsNothing(sc, r)) enter = 0u;
			else if (saWorld.initInterval(r)) enter = enterObject(r, saWorld, FLT_MAX, false);
			if (enter == 2u && level == 0) // once more, the way renderSlot tries a primary ray again
				enter = parallelRayQualifies(saWorld, r, kParallelMaxObjects) ? enterObject(r, saWorld, FLT_MAX, true) : 2u;
			if (enter == 2u) listed = true;
			if (enter != 1u) alive = false; // listed, or a miss: renderer.cu:315
		}
		soloWalk(mem, nodeBase, nodeCount, triBase, triFirst, stack, alive, lr, sa, root, R, nearest);
		if (alive) {
			if (R.second < FLT_MAX && R.second <= R.leafTmin) { // the visiting order could have mattered
				listed = true;
				alive = false;
			} else if (!R.any) {
				alive = false;
			} else {
				vec3 pw;
				const float t = worldHit(r.o, R.t, pw);
				// info as Scene::intersect leaves it: replaced iff t < info.t (scene.cu:432), which started at FLT_MAX
				uint32_t prim = 0xFFFFFFFFu;
				float infoT = FLT_MAX;
				if (t < infoT) {
					infoT = t;
					position = pw;
					material = loadMeshRecord(mem, sc, meshId).q0.w;
					const uint32_t ns = sc.off_normals + R.tri * kNormalSlots;
					const uint4 n0 = mem.large(ns), n1 = mem.large(ns + 1), n2 = mem.large(ns + 2);
					const vec3 na(asF(n0.x), asF(n0.y), asF(n0.z));
					const vec3 nb(asF(n0.w), asF(n1.x), asF(n1.y));
					const vec3 nc(asF(n1.z), asF(n1.w), asF(n2.x));
					const vec3 nl = R.bary.x * na + R.bary.y * nb + R.bary.z * nc; // geometry.h:141
					uint32_t u0, u1;
					const mat3 N = loadMat3(mem, os + 13, u0, u1);
					normal = translaten(N, nl); // scene.cu:429
					prim = mem.large(sc.off_tris + R.tri * kTriSlots + 2).y;
				}
				if (level == 0) {
					LateParams &q = lateParams();
					if (q.primid) q.primid[outSlot] = prim;
					if (q.hit_t) q.hit_t[outSlot] = infoT;
					res.cnt.primaryHits++;
				}
			}
		}
		}

		// Scene::lightPoint (scene.cu:273-292), one light after the other for the whole wave
		for (uint32_t li = 0; li < sc.num_lights; li++) {
			vec3 c(0, 0, 0);
			float lightT = 0.0f;
			uint32_t enterS = 0u;
			Ray shadowRay; // MULTI
			shadowRay.o = vec3(0, 0, 0);
			shadowRay.d = vec3(0, 0, 1);
			if (alive) {
				const Light l = loadLight(mem, sc, li);
				vec3 dir;
				bool trace = true;
				if (l.type == 0u) { // pointLightAccum, scene.cu:209-231
					dir = (l.position + 0.5f * randNorm(state)) - position;
					lightT = length(dir);
					dir = normalize(dir);
					const float cosTheta = max(dot(dir, normal), 0.0f);
					c = cosTheta / (l.attConstant + (l.attLinear + l.attQuadratic * lightT) * lightT) * l.diffuse;
					if (length(c) < epsilon()) trace = false;
				} else { // spotLightAccum, scene.cu:234-271
					dir = l.position - position;
					lightT = length(dir);
					dir = normalize(dir);
					const float cosTheta = max(dot(dir, normal), 0.0f);
					const float cosLight = dot(-dir, normalize(l.direction));
					const float lightTheta = acosf(cosLight);
					const float inner = l.outerCone - 1.57079632679f;
					const float outer = l.innerCone;
					const float falloff_distance = outer - inner;
					if (fabsf(lightTheta) >= outer / 2.0f) {
						c = vec3(0, 0, 0);
						trace = false;
					} else {
						c = cosTheta / (l.attConstant + (l.attLinear + l.attQuadratic * lightT) * lightT) * l.diffuse;
						if (length(c) < epsilon()) {
							trace = false;
						} else {
							const float val = (outer / 2.0f - fabsf(lightTheta));
							const float falloff = val / (falloff_distance / 2.0f);
							if (double(fabsf(lightTheta)) >= double(inner) / 2.0)
								c *= falloff;
						}
					}
				}
				if (trace) {
					res.cnt.shadow++;
					Ray sr;
					sr.o = position;
					sr.d = dir;
					if constexpr (MULTI) {
						shadowRay = sr;
						enterS = 3u; // traced below, by the whole wave
					} else {
						SlabArgs saWorld;
						enterS = saWorld.initInterval(sr) ? enterObject(sr, saWorld, lightT, false) : 2u;
						if (enterS == 2u) {
							listed = true;
							alive = false;
						}
					}
				}
			}
			if constexpr (MULTI) {
				bool traced = alive && enterS == 3u;
				Best best;
				traceObjects(traced, shadowRay, true, lightT, 1.0f, false, best);
				if (alive && enterS == 3u) {
					if (!traced) alive = false; // listed
					else if (best.t + epsilon() < lightT) c = vec3(0, 0, 0); // scene.cu:226,266
				}
				if (alive) lightSum += c; // scene.cu:290
			} else {
			soloWalk(mem, nodeBase, nodeCount, triBase, triFirst, stack, alive && enterS == 1u, lr, sa, root, R, nearest);
			if (alive) {
				if (enterS == 1u) {
					if (R.second < FLT_MAX && R.second <= R.leafTmin) {
						listed = true;
						alive = false;
					} else {
						// scene.cu:226,266
						float infoT = lightT;
						if (R.any) {
							vec3 pw;
							const float t = worldHit(position, R.t, pw);
							if (t < infoT) infoT = t;
						}
						if (infoT + epsilon() < lightT) c = vec3(0, 0, 0);
					}
				}
				if (alive) lightSum += c; // scene.cu:290
			}
			}
		}

		if (alive) {
			// the rest of this level (renderer.cu:247-313)
			const uint32_t msl = sc.off_materials + material * kMaterialSlots;
			const uint4 m0 = mem.small(msl), m1 = mem.small(msl + 1), m2 = mem.small(msl + 2);
			const vec3 ambient(asF(m0.x), asF(m0.y), asF(m0.z));
			const float refractiveIndex = asF(m0.w);
			const vec3 diffuse(asF(m1.x), asF(m1.y), asF(m1.z));
			const vec3 specular(asF(m2.x), asF(m2.y), asF(m2.z));
			const vec3 c = ambient + diffuse * lightSum;
			// nSource is uninitialised in the reference's translucent branch (renderer.cu:251,286-311); `source` here, as in renderSlot
			vec3 nSource = source;
			Ray nr;
			nr.o = position;
			nr.d = vec3(0, 0, 0);
			const bool last = ONE || level + 1 == maxBounces; // the next level returns (0,0,0,1) whatever the ray
			bool absorbed = false;
			if (refractiveIndex == 0) {
				const vec3 cd = source * diffuse;
				const vec3 cs = source * specular;
				const float Pd = compMax(cd) / compMax(source);
				const float Ps = compMax(cs) / compMax(source);
				const float P = randUniform(state);
				if (P < Pd) {
					if (!last) {
						nSource = cd * (1 / Pd);
						nr.d = randomHemi(normal, state);
					}
				} else if (P < Pd + Ps) {
					if (!last) {
						nSource = cs * (1 / Ps);
						nr.d = reflect(r.d, normal);
					}
				} else {
					absorbed = true; // renderer.cu:282
				}
			} else if (!last) {
				float n1, n2;
				vec3 norm = normal;
				if (dot(r.d, norm) < 0) {
					n1 = refractiveIndex;
					n2 = 1;
				} else {
					n1 = 1;
					n2 = refractiveIndex;
					norm *= -1.0f;
				}
				const float Rs = rSchlick2(r.d, norm, n1, n2);
				if (randUniform(state) >= Rs)
					nr.d = refract(r.d, norm, n1 / n2);
				else
					nr.d = reflect(r.d, norm);
			}
			if (absorbed) {
				tail = vec4(c, 1);
				alive = false;
			} else if (last) { // Color4(c,1) + rayColor<0>() = Color4(c,1) + (0,0,0,1)
				tail = vec4(c, 1) + vec4(0, 0, 0, 1);
				alive = false;
			} else {
				// Color4(c,1) + rayColor<N-1>(...): remember c, add after the deeper levels
				cstack[(3 * level + 0) * kBlock] = c.x;
				cstack[(3 * level + 1) * kBlock] = c.y;
				cstack[(3 * level + 2) * kBlock] = c.z;
				kept = level + 1;
				r = nr;
				source = nSource;
			}
		}
	}

	if (listed) {
		// nothing of this pixel is kept: the list pass renders it from the start
		res.bail = true;
		res.cnt = {0, 0, 0};
		return res;
	}
	if (!res.active)
		return res;
	vec4 contribution = tail;
	if constexpr (!ONE) {
		while (kept > 0) {
			kept--;
			const vec4 cl(cstack[(3 * kept + 0) * kBlock], cstack[(3 * kept + 1) * kBlock], cstack[(3 * kept + 2) * kBlock], 1.0f);
			contribution = cl + contribution;
		}
	}
	// average (renderer.cu:375-387)
	LateParams &q = lateParams();
	float4 *px = q.screen + outSlot;
	if (q.frames > 0) {
		const float4 old = *px;
		vec4 s(old.x, old.y, old.z, old.w);
		s = s * (float(q.frames) / float(q.frames + 1));
		s = s + (contribution / float(q.frames + 1));
		*px = make_float4(s.x, s.y, s.z, s.w);
	} else {
		*px = make_float4(contribution.x, contribution.y, contribution.z, contribution.w);
	}
	return res;
}

// renderKernel<SMALL_LDS, false, true, false, true> for a scene of one object with one mesh: same grid, same block
// order, same list pass afterwards.
template <bool SMALL_LDS, bool ONE, bool MULTI = false>
__global__ void __launch_bounds__(kBlock, ONE ? ACR_SOLO_WAVES : MULTI ? ACR_MULTI_WAVES : ACR_RENDER_WAVES_HBM) renderSoloKernel(const RenderParams p)
{
	const SceneHeader &sc = p.scene;
	SceneMem<SMALL_LDS, false> mem;
	mem.g = p.blob;
	if constexpr (SMALL_LDS) stageScene(p.blob, sc.small_slots);
	const bool tiled = p.num_tiles != 0;
	const uint64_t clock0 = ((tiled && p.tile_cycles) || p.block_cost) ? __builtin_readcyclecounter() : 0;
	const uint32_t firstSlot = blockIdx.x * uint32_t(kBlock);
	const uint32_t gblock = firstSlot / kSlots, vtid = firstSlot % kSlots + threadIdx.x;
	const uint32_t vblock = p.block_order ? p.block_order[gblock] : gblock;
	const SlotResult res = renderSlotSolo<ONE, MULTI>(p, mem, vblock, vtid);
	LateParams &q = lateParams();
	if (res.bail) {
		const uint32_t at = atomicAdd(q.redo_count, 1u);
		q.redo_list[at] = vblock * kSlots + vtid;
	}
	const bool firstLane = laneNow() == 0;
	const uint32_t tileNo = tileOfBlock(q, vblock);
	if (q.num_tiles != 0 && q.tile_cycles && firstLane)
		atomicAdd(&q.tile_cycles[tileNo], uint32_t((__builtin_readcyclecounter() - clock0) >> 6));
	if (q.block_cost && firstLane)
		atomicAdd(&q.block_cost[vblock], uint32_t((__builtin_readcyclecounter() - clock0) >> 6));
	reportCounts(q, res, vblock, tileNo);
}

// renderKernel for scenes resident in LDS and frames with bounces: the waves trace their rays together
// (renderSlotWave).  Same grid, same block order, same list pass afterwards.
__global__ void __launch_bounds__(kBlock, ACR_WAVE_KERNEL_WAVES) renderWaveKernel(const RenderParams p)
{
	// this kernel's own copy of the scene: object records 17 slots apart (the host has made room: planLds)
	SceneMem<true, true, false, kWaveObjStride> mem;
	mem.g = p.blob;
	stageSceneSpread(p.blob, p.scene);
	const SceneHeader sc = spreadHeader(p.scene);
	const bool tiled = p.num_tiles != 0;
	const uint64_t clock0 = ((tiled && p.tile_cycles) || p.block_cost) ? __builtin_readcyclecounter() : 0;
	const uint32_t firstSlot = blockIdx.x * uint32_t(kBlock);
	const uint32_t gblock = firstSlot / kSlots, vtid = firstSlot % kSlots + threadIdx.x;
	const uint32_t vblock = p.block_order ? p.block_order[gblock] : gblock;
#ifdef ACR_WAVE_DEBUG
	const uint64_t dbgK0 = __builtin_readcyclecounter();
#endif
	const SlotResult res = renderSlotWave(p, sc, mem, vblock, vtid);
#ifdef ACR_WAVE_DEBUG
	if ((threadIdx.x & 63u) == 0) atomicAdd(&g_waveDebug[9], (unsigned long long)(__builtin_readcyclecounter() - dbgK0));
#endif
	LateParams &q = lateParams();
	if (res.bail) {
		const uint32_t at = atomicAdd(q.redo_count, 1u);
		q.redo_list[at] = vblock * kSlots + vtid;
	}
	const bool firstLane = laneNow() == 0;
	const uint32_t tileNo = tileOfBlock(q, vblock);
	if (q.num_tiles != 0 && q.tile_cycles && firstLane)
		atomicAdd(&q.tile_cycles[tileNo], uint32_t((__builtin_readcyclecounter() - clock0) >> 6));
	if (q.block_cost && firstLane)
		atomicAdd(&q.block_cost[vblock], uint32_t((__builtin_readcyclecounter() - clock0) >> 6));
	reportCounts(q, res, vblock, tileNo);
}

// Renders the pixel slots listed by a FAST launch with the six-plane traversal: a fixed small grid that
// strides over the list (usually empty).  Also zeroes the counter the NEXT frame's FAST launch appends with.
template <bool SMALL_LDS, bool LARGE_LDS, bool KEYS>
__global__ void __launch_bounds__(kBlock) renderListKernel(const RenderParams p)
{
	// scenes walked near side first list pixels for two reasons; the others only for non-finite rays
	constexpr bool ORDERED_FIRST = SceneMem<SMALL_LDS, LARGE_LDS>::kNearFirst;
	const uint32_t count = *p.redo_count;
	if (blockIdx.x == 0 && threadIdx.x == 0) *p.redo_count_next = 0;
	if (count == 0) return;
	const SceneHeader &sc = p.scene;
	SceneMem<SMALL_LDS, LARGE_LDS, false> mem; // always the reference's visiting order here
	mem.g = p.blob;
	if constexpr (LARGE_LDS) stageScene(p.blob, sc.total_slots);
	else if constexpr (SMALL_LDS) stageScene(p.blob, sc.small_slots);
	// Entry k goes to workgroup k % gridDim, work-item k / gridDim: a short list is spread one pixel per workgroup
	// (a pixel alone in its wave is the fastest it can be), a long one fills the workgroups evenly.
	for (uint32_t base = blockIdx.x; base < count; base += gridDim.x * kBlock) {
		const uint32_t k = base + gridDim.x * threadIdx.x;
		SlotResult res;
		res.active = false;
		res.bail = false;
		res.tileNo = 0;
		res.cnt = {0, 0, 0};
		uint32_t vblock = 0;
		if (k < count) {
			const uint32_t slot = p.redo_list[k];
			vblock = slot / kSlots;
			// a pixel listed because the visiting order could matter qualifies for the interval form in the
			// reference's order; one listed for its ray fails again and gets the six-plane form below
			if constexpr (ORDERED_FIRST) res = renderSlot<true, KEYS>(p, mem, vblock, slot % kSlots);
		}
		if constexpr (ORDERED_FIRST) __syncthreads(); // the two forms lay their LDS columns out differently
		if (k < count && (!ORDERED_FIRST || res.bail)) {
			const uint32_t slot = p.redo_list[k];
			res = renderSlot<false, KEYS>(p, mem, vblock, slot % kSlots);
		}
		if constexpr (ORDERED_FIRST) __syncthreads();
		// lanes of one wave may belong to different tiles and blocks here: count per lane
		if (res.active) {
			if (p.num_tiles != 0 && p.tile_rays) atomicAdd(&p.tile_rays[res.tileNo], 1u + res.cnt.shadow + res.cnt.bounce);
			if (p.counters) {
				unsigned long long *c = p.counters + 4 * (vblock % kCounterSlots);
				atomicAdd(&c[0], 1ull);
				atomicAdd(&c[1], (unsigned long long)res.cnt.shadow);
				atomicAdd(&c[2], (unsigned long long)res.cnt.bounce);
				atomicAdd(&c[3], (unsigned long long)res.cnt.primaryHits);
			}
		}
	}
}

// Scene::intersect for caller-supplied rays, one ray per work-item.  FAST: interval traversal for the
// rays that qualify, six-plane traversal for the others, in the same launch.
template <bool SMALL_LDS, bool LARGE_LDS, bool FAST>
__global__ void __launch_bounds__(kBlock) traceKernel(const TraceParams p)
{
	const SceneHeader &sc = p.scene;
	SceneMem<SMALL_LDS, LARGE_LDS> mem;
	mem.g = p.blob;
	if constexpr (LARGE_LDS) stageScene(p.blob, sc.total_slots);
	else if constexpr (SMALL_LDS) stageScene(p.blob, sc.small_slots);
	float *ldsF = reinterpret_cast<float *>(g_lds + p.lds_stack_slot);

	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= p.n) return;
	Ray r;
	r.o = vec3(p.origins[3 * i], p.origins[3 * i + 1], p.origins[3 * i + 2]);
	r.d = vec3(p.dirs[3 * i], p.dirs[3 * i + 1], p.dirs[3 * i + 2]);
	WorldHit info;
	bool h = false, bail = !FAST;
	if constexpr (FAST) {
		info.t = p.tmax ? p.tmax[i] : FLT_MAX;
		info.material = 0;
		info.prim = 0xFFFFFFFFu;
		info.position = vec3(0, 0, 0);
		info.normal = vec3(0, 0, 0);
		const Columns col = {ldsF + 2 * threadIdx.x, ldsF + 2 * threadIdx.x + 2 * sc.scene_depth * kBlock, nullptr, 0u, 0u};
		if (!hitsNothing(sc, r)) h = sceneIntersect<false, true>(mem, sc, r, info, 0.0f, col, bail);
	}
	if (bail) {
		info.t = p.tmax ? p.tmax[i] : FLT_MAX;
		info.material = 0;
		info.prim = 0xFFFFFFFFu;
		info.position = vec3(0, 0, 0);
		info.normal = vec3(0, 0, 0);
		bool unused = false;
		// Other waves of this workgroup may still be in the interval form: the undo columns of the six-plane form
		// get their own LDS region behind the waiting-child columns (FAST), or start at the front (not FAST).
		float *base = ldsF + (FAST ? 2 * (sc.scene_depth + sc.max_mesh_depth) * kBlock : 0) + threadIdx.x;
		const Columns col = {base, base + sc.scene_depth * kBlock, nullptr, 0u, 0u};
		h = sceneIntersect<false, false>(mem, sc, r, info, 0.0f, col, unused);
	}
	p.hit[i] = h ? 1 : 0;
	p.t[i] = info.t;
	p.prim[i] = info.prim;
	p.position[3 * i] = info.position.x; p.position[3 * i + 1] = info.position.y; p.position[3 * i + 2] = info.position.z;
	p.normal[3 * i] = info.normal.x; p.normal[3 * i + 1] = info.normal.y; p.normal[3 * i + 2] = info.normal.z;
}

// Launch order of a frame's blocks from their measured cost, most expensive first: one workgroup, a counting
// sort into 256 cost classes (linear in cost / largest cost).  The order inside a class is whatever the atomics
// make it; frames do not depend on the order at all.  Every wave counts into bins of its own: one LDS atomic per
// element, contended only inside the wave (most blocks of a frame fall into the same few classes).
constexpr uint32_t kOrderThreads = 1024, kOrderWaves = kOrderThreads / 64;
__global__ void __launch_bounds__(kOrderThreads) orderBlocksKernel(uint32_t *__restrict__ cost, uint32_t n, uint32_t *__restrict__ order)
{
	__shared__ uint32_t bins[kOrderWaves][256];
	__shared__ uint32_t largest;
	__shared__ uint32_t waveSum[4];
	for (uint32_t k = threadIdx.x; k < kOrderWaves * 256u; k += kOrderThreads) (&bins[0][0])[k] = 0;
	if (threadIdx.x == 0) largest = 1;
	__syncthreads();
	// eight elements per step, requested together (one by one, a pass is a chain of dependent round trips)
	auto forEach = [&](auto body) {
		for (uint32_t base = 0; base < n; base += 8u * kOrderThreads) {
			uint32_t c[8];
#pragma unroll
			for (uint32_t j = 0; j < 8; j++) {
				const uint32_t i = base + j * kOrderThreads + threadIdx.x;
				c[j] = cost[i < n ? i : n - 1u]; // unconditional: a load under a branch is waited for before the next is requested
			}
#pragma unroll
			for (uint32_t j = 0; j < 8; j++) {
				const uint32_t i = base + j * kOrderThreads + threadIdx.x;
				if (i < n) body(i, c[j]);
			}
		}
	};
	uint32_t m = 0;
	forEach([&](uint32_t, uint32_t c) { m = c > m ? c : m; });
	atomicMax(&largest, m);
	__syncthreads();
	const float scale = 255.0f / float(largest);
	auto classOf = [scale](uint32_t c) {
		const uint32_t k = uint32_t(float(c) * scale);
		return 255u - (k > 255u ? 255u : k);
	};
	const uint32_t wave = threadIdx.x >> 6;
	forEach([&](uint32_t, uint32_t c) { atomicAdd(&bins[wave][classOf(c)], 1u); });
	__syncthreads();
	// class k of wave w starts behind the classes before k (all waves) and class k of the waves before w
	uint32_t total = 0, incl = 0;
	if (threadIdx.x < 256) {
		for (uint32_t w = 0; w < kOrderWaves; w++) {
			const uint32_t c = bins[w][threadIdx.x];
			bins[w][threadIdx.x] = total;
			total += c;
		}
		// prefix sum of the class totals over the 256 classes: a scan inside each of the four waves, then the waves' sums
		incl = total;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t v = __shfl_up(incl, d, 64);
			incl += (threadIdx.x & 63u) >= uint32_t(d) ? v : 0u;
		}
		if ((threadIdx.x & 63u) == 63u) waveSum[threadIdx.x >> 6] = incl;
	}
	__syncthreads();
	if (threadIdx.x < 256) {
		uint32_t before = 0;
		for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += waveSum[w];
		const uint32_t start = before + incl - total;
		for (uint32_t w = 0; w < kOrderWaves; w++) bins[w][threadIdx.x] += start;
	}
	__syncthreads();
	// the same elements go to the same wave as in the counting pass
	// ... and the costs are left at zero for the next measuring launch (which adds to them): no memset in between
	forEach([&](uint32_t i, uint32_t c) {
		order[atomicAdd(&bins[wave][classOf(c)], 1u)] = i;
		cost[i] = 0;
	});
}

// The same table for scenes whose trees and triangles are read through the L2: XCD-aware.  Workgroups are dealt
// round-robin over the GPU's XCDs, so table position i is rendered on XCD i % kXcds, and every XCD has an L2 of its
// own.  The blocks' Z-curve (zseq, made by the host) is cut into kXcds runs of equal measured cost, run g is sorted
// most expensive first and laid out on positions g, g + kXcds, ...: the workgroups that share an L2 then walk
// neighbouring parts of the trees at the same time.  A run with more blocks than its XCD has positions (cheap
// blocks) hands its cheapest ones to the positions that the runs with few, expensive blocks leave free at their
// XCDs' ends.  One workgroup; thread t owns a stretch of the Z-curve.  Leaves cost[] at zero like orderBlocksKernel.
constexpr uint32_t kXcds = kXcdsAssumed;
__global__ void __launch_bounds__(kOrderThreads) orderBlocksRegionsKernel(uint32_t *__restrict__ cost, uint32_t n, uint32_t *__restrict__ order,
                                                                          const uint32_t *__restrict__ zseq)
{
	__shared__ unsigned long long partial[kOrderThreads];
	__shared__ uint32_t bins[kXcds][256];
	__shared__ uint32_t largest;
	__shared__ uint32_t runCount[kXcds], overflowStart[kXcds + 1], freeStart[kXcds + 1];
	for (uint32_t k = threadIdx.x; k < kXcds * 256u; k += kOrderThreads) (&bins[0][0])[k] = 0;
	if (threadIdx.x == 0) largest = 1;
	__syncthreads();
	const uint32_t chunk = (n + kOrderThreads - 1u) / kOrderThreads;
	const uint32_t lo = threadIdx.x * chunk < n ? threadIdx.x * chunk : n;
	const uint32_t hi = lo + chunk < n ? lo + chunk : n;
	unsigned long long sum = 0;
	uint32_t m = 0;
	for (uint32_t i = lo; i < hi; i++) {
		const uint32_t c = cost[zseq[i]];
		sum += c;
		m = c > m ? c : m;
	}
	partial[threadIdx.x] = sum;
	atomicMax(&largest, m);
	__syncthreads();
	unsigned long long before = 0, total = 0;
	for (uint32_t t = 0; t < kOrderThreads; t++) {
		const unsigned long long v = partial[t];
		before += t < threadIdx.x ? v : 0ull;
		total += v;
	}
	if (total == 0) total = 1;
	const float scale = 255.0f / float(largest);
	auto classOf = [scale](uint32_t c) {
		const uint32_t k = uint32_t(float(c) * scale);
		return 255u - (k > 255u ? 255u : k);
	};
	// the run of a block: where the middle of its cost lies in the whole, in eighths
	auto runOf = [total](unsigned long long prefix, uint32_t c) {
		const uint32_t g = uint32_t((kXcds * (prefix + c / 2u)) / total);
		return g < kXcds ? g : kXcds - 1u;
	};
	{
		unsigned long long prefix = before;
		for (uint32_t i = lo; i < hi; i++) {
			const uint32_t c = cost[zseq[i]];
			atomicAdd(&bins[runOf(prefix, c)][classOf(c)], 1u);
			prefix += c;
		}
	}
	__syncthreads();
	if (threadIdx.x < kXcds) {
		uint32_t at = 0;
		for (uint32_t k = 0; k < 256u; k++) {
			const uint32_t c = bins[threadIdx.x][k];
			bins[threadIdx.x][k] = at;
			at += c;
		}
		runCount[threadIdx.x] = at;
	}
	__syncthreads();
	auto slotsOf = [n](uint32_t g) { return g < n ? (n - g + kXcds - 1u) / kXcds : 0u; }; // positions g, g + kXcds, ... below n
	if (threadIdx.x == 0) {
		uint32_t ov = 0, fr = 0;
		for (uint32_t g = 0; g < kXcds; g++) {
			overflowStart[g] = ov;
			freeStart[g] = fr;
			const uint32_t have = runCount[g], room = slotsOf(g);
			ov += have > room ? have - room : 0u;
			fr += room > have ? room - have : 0u;
		}
		overflowStart[kXcds] = ov;
		freeStart[kXcds] = fr;
	}
	__syncthreads();
	{
		unsigned long long prefix = before;
		for (uint32_t i = lo; i < hi; i++) {
			const uint32_t b = zseq[i];
			const uint32_t c = cost[b];
			const uint32_t g = runOf(prefix, c);
			prefix += c;
			const uint32_t r = atomicAdd(&bins[g][classOf(c)], 1u);
			uint32_t pos;
			if (r < slotsOf(g)) {
				pos = kXcds * r + g;
			} else {
				const uint32_t ov = overflowStart[g] + (r - slotsOf(g));
				uint32_t h = 0;
				while (h + 1u < kXcds && freeStart[h + 1u] <= ov) h++;
				pos = kXcds * (runCount[h] + (ov - freeStart[h])) + h;
			}
			order[pos] = b;
			cost[b] = 0;
		}
	}
}

// compact tiles -> full frame (last step of the multi-GPU gather)
__global__ void __launch_bounds__(kBlock) scatterTilesKernel(const float4 *__restrict__ tiles, const uint32_t *__restrict__ tileIds,
                                                             uint32_t tileW, uint32_t tileH, uint32_t tilesX, uint32_t width, uint32_t height,
                                                             float4 *__restrict__ frame)
{
	const uint32_t tileNo = blockIdx.x;
	const uint32_t tile = tileIds[tileNo];
	const uint32_t x0 = (tile % tilesX) * tileW, y0 = (tile / tilesX) * tileH;
	const uint32_t npx = tileW * tileH;
	for (uint32_t k = threadIdx.x; k < npx; k += kBlock) {
		const uint32_t x = x0 + k % tileW, y = y0 + k / tileW;
		if (x < width && y < height)
			frame[size_t(y) * width + x] = tiles[size_t(tileNo) * npx + k];
	}
}

} // namespace dev
} // namespace acr

// ---- launchers used by the C ABI implementation (acr_hip.cpp) --------------

namespace
{
	// The dynamic-LDS limit of a kernel is raised once per (device, kernel) and only when a launch needs more than
	// any earlier one: the attribute call costs a few microseconds of host time, which a 0.2 ms frame notices.
	constexpr int kMaxDevices = 64;
	template <class P>
	hipError_t launchWithLds(void (*k)(const P), const P &p, uint32_t grid, size_t ldsBytes, hipStream_t stream)
	{
		static std::mutex lock;
		static std::unordered_map<const void *, size_t> granted[kMaxDevices];
		int dev = 0;
		hipError_t e = hipGetDevice(&dev);
		if (e != hipSuccess) return e;
		const void *fn = reinterpret_cast<const void *>(k);
		bool raise = true;
		if (dev >= 0 && dev < kMaxDevices) {
			std::lock_guard<std::mutex> g(lock);
			auto it = granted[dev].find(fn);
			raise = it == granted[dev].end() || it->second < ldsBytes;
		}
		if (raise) {
			e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes));
			if (e != hipSuccess) return e;
			if (dev >= 0 && dev < kMaxDevices) {
				std::lock_guard<std::mutex> g(lock);
				granted[dev][fn] = ldsBytes;
			}
		}
		(void)hipGetLastError(); // an earlier, unrelated failure of some HIP call in this thread is not this launch's
		hipLaunchKernelGGL(k, dim3(grid), dim3(acr::dev::kBlock), ldsBytes, stream, p);
		return hipGetLastError();
	}
} // namespace

// fast == 1: interval traversal for the whole grid, then the list pass over whatever it handed back
// (p->redo_* must be set).  fast == 3: the same with the wave-cooperative kernel (scene whole in LDS, no keys).  fast == 4: the same with the kernel for one deep mesh (renderSoloKernel; the caller has checked the scene and the frame); fast == 5: that kernel's form for up to 64 objects.  fast == 2: the list pass alone (the caller listed the slots).  fast == 0: six-plane
// traversal for the whole grid.  p->path_keys selects the
// variants that also record the per-pixel traversal-path key.
template <bool KEYS>
static hipError_t launchRender(const acr::dev::RenderParams *p, int ldsMode, int fast, uint32_t grid, size_t ldsBytes, hipStream_t stream)
{
	using namespace acr::dev;
	typedef void (*K)(const RenderParams);
	if (!fast) {
		K k = ldsMode == 2 ? renderKernel<true, true, false, KEYS> : (ldsMode == 1 ? renderKernel<true, false, false, KEYS> : renderKernel<false, false, false, KEYS>);
		return launchWithLds(k, *p, grid, ldsBytes, stream);
	}
	K kl = ldsMode == 2 ? renderListKernel<true, true, KEYS> : (ldsMode == 1 ? renderListKernel<true, false, KEYS> : renderListKernel<false, false, KEYS>);
	if (fast == 2) return launchWithLds(kl, *p, kRedoGrid, ldsBytes, stream); // the caller has listed every slot
	K k = ldsMode == 2 ? renderKernel<true, true, true, KEYS> : (ldsMode == 1 ? renderKernel<true, false, true, KEYS> : renderKernel<false, false, true, KEYS>);
	if constexpr (!KEYS) {
		if (fast == 3) k = renderWaveKernel; // ldsMode 2 only
		if (p->max_bounces == 1)
			k = ldsMode == 2 ? renderKernel<true, true, true, false, true> : (ldsMode == 1 ? renderKernel<true, false, true, false, true> : renderKernel<false, false, true, false, true>);
		if (fast == 4) { // one object, one mesh, trees in HBM
			if (p->max_bounces == 1) k = ldsMode == 1 ? renderSoloKernel<true, true> : renderSoloKernel<false, true>;
			else k = ldsMode == 1 ? renderSoloKernel<true, false> : renderSoloKernel<false, false>;
		}
		if (fast == 5) k = ldsMode == 1 ? renderSoloKernel<true, false, true> : renderSoloKernel<false, false, true>; // up to 64 objects, trees in HBM
	}
	hipError_t e = launchWithLds(k, *p, grid, ldsBytes, stream);
	if (e != hipSuccess) return e;
	return launchWithLds(kl, *p, kRedoGrid, ldsBytes, stream);
}

extern "C" hipError_t acr_dev_launch_order_blocks_regions(uint32_t *cost, uint32_t n, uint32_t *order, const uint32_t *zseq, hipStream_t stream)
{
	hipLaunchKernelGGL(acr::dev::orderBlocksRegionsKernel, dim3(1), dim3(acr::dev::kOrderThreads), 0, stream, cost, n, order, zseq);
	return hipGetLastError();
}

extern "C" hipError_t acr_dev_launch_render(const acr::dev::RenderParams *p, int ldsMode, int fast, uint32_t grid, size_t ldsBytes, hipStream_t stream)
{
	return p->path_keys ? launchRender<true>(p, ldsMode, fast, grid, ldsBytes, stream) : launchRender<false>(p, ldsMode, fast, grid, ldsBytes, stream);
}

// PathAggregate (bih.h:29-66) for every pixel, the counters plane by plane (votes[bit * n + pixel]) so that the
// work-items of a wave touch consecutive addresses.  reset: PathAggregate::set (renderer.cu:386), else ::add
// (renderer.cu:380); counter i belongs to bit i of the path, so that an aggregate of one path converts back to it.
__global__ void __launch_bounds__(256) votePathsKernel(const uint64_t *__restrict__ keys, int32_t *__restrict__ votes, uint32_t n, int reset)
{
	const uint32_t px = blockIdx.x * 256u + threadIdx.x;
	if (px >= n) return;
	uint64_t bits = keys[px];
#pragma unroll 8
	for (uint32_t i = 0; i < 64; i++) {
		const int32_t vote = (bits & 1u) ? -1 : 1;
		int32_t *w = votes + size_t(i) * n + px;
		*w = reset ? vote : *w + vote;
		bits >>= 1;
	}
}
// PathAggregate::operator uint64_t (bih.h:55-65): bit i set where counter i is negative
__global__ void __launch_bounds__(256) voteKeysKernel(const int32_t *__restrict__ votes, uint64_t *__restrict__ keys, uint32_t n)
{
	const uint32_t px = blockIdx.x * 256u + threadIdx.x;
	if (px >= n) return;
	uint64_t path = 0;
#pragma unroll 8
	for (uint32_t i = 0; i < 64; i++)
		path |= uint64_t(votes[size_t(i) * n + px] < 0) << i;
	keys[px] = path;
}
extern "C" hipError_t acr_dev_launch_vote_paths(const uint64_t *keys, int32_t *votes, uint32_t n, int reset, hipStream_t stream)
{
	hipLaunchKernelGGL(votePathsKernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, votes, n, reset);
	return hipGetLastError();
}
extern "C" hipError_t acr_dev_launch_vote_keys(const int32_t *votes, uint64_t *keys, uint32_t n, hipStream_t stream)
{
	hipLaunchKernelGGL(voteKeysKernel, dim3((n + 255) / 256), dim3(256), 0, stream, votes, keys, n);
	return hipGetLastError();
}

// values[i] = i (thrust::sequence at renderer.cu:137)
__global__ void __launch_bounds__(256) iotaKernel(int32_t *values, uint32_t n)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) values[i] = int32_t(i);
}
extern "C" hipError_t acr_dev_launch_order_blocks(uint32_t *cost, uint32_t n, uint32_t *order, hipStream_t stream)
{
	hipLaunchKernelGGL(acr::dev::orderBlocksKernel, dim3(1), dim3(acr::dev::kOrderThreads), 0, stream, cost, n, order);
	return hipGetLastError();
}

// The reciprocals the kernels take without dividing -- acr::math::reciprocal, SlabArgs::initInterval and the
// determinant's in myIntersectRayTriangleEdges -- against the division they stand for, on 2^24 argument patterns per
// launch (the patterns whose top byte is `hi`).  Counts the patterns for which a result that the kernels would use is
// neither the division's bits nor, like it, a NaN; and those inside [2^-126, 2^126] that would be sent to the slow path.
__global__ void __launch_bounds__(256) reciprocalCheckKernel(unsigned long long *differ, uint32_t hi)
{
	const uint32_t bits = (hi << 24) | (blockIdx.x << 8) | threadIdx.x;
	const float x = __uint_as_float(bits);
	const float want = 1.0f / x;
	auto same = [](float p, float q) { return __float_as_uint(p) == __float_as_uint(q) || (p != p && q != q); };
	bool ok = same(want, acr::math::reciprocal(x));
	// the three-component form, with the argument in each place next to two harmless ones
	const bool inRange = fabsf(x) >= 0x1p-126f && fabsf(x) <= 0x1p126f;
	acr::dev::Ray r;
	r.o = vec3(0, 0, 0);
	acr::dev::SlabArgs a;
	r.d = vec3(x, 1.0f, -4.0f);
	bool q = a.initInterval(r);
	ok = ok && (!q || (same(want, a.invD.x) && same(a.invD.y, 1.0f) && same(a.invD.z, -0.25f))) && (q || !inRange);
	r.d = vec3(0.5f, x, 8.0f);
	q = a.initInterval(r);
	ok = ok && (!q || (same(want, a.invD.y) && same(a.invD.x, 2.0f) && same(a.invD.z, 0.125f))) && (q || !inRange);
	r.d = vec3(-2.0f, 0.25f, x);
	q = a.initInterval(r);
	ok = ok && (!q || (same(want, a.invD.z) && same(a.invD.x, -0.5f) && same(a.invD.y, 4.0f))) && (q || !inRange);
	// the range a determinant can have in the interval-form kernels
	if (fabsf(x) >= acr::math::epsilon() && fabsf(x) <= 0x1p126f) ok = ok && same(want, acr::math::reciprocalNewton(x));
	if (!ok) atomicAdd(differ, 1ull);
}

extern "C" hipError_t acr_dev_launch_reciprocal_check(unsigned long long *differ, hipStream_t stream)
{
	for (uint32_t hi = 0; hi < 256; hi++) {
		hipLaunchKernelGGL(reciprocalCheckKernel, dim3(1u << 16), dim3(256), 0, stream, differ, hi);
		const hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

extern "C" hipError_t acr_dev_launch_iota(int32_t *values, uint32_t n, hipStream_t stream)
{
	hipLaunchKernelGGL(iotaKernel, dim3((n + 255) / 256), dim3(256), 0, stream, values, n);
	return hipGetLastError();
}

extern "C" hipError_t acr_dev_launch_trace(const acr::dev::TraceParams *p, int ldsMode, int fast, uint32_t grid, size_t ldsBytes, hipStream_t stream)
{
	using namespace acr::dev;
	typedef void (*K)(const TraceParams);
	K k;
	if (fast) k = ldsMode == 2 ? traceKernel<true, true, true> : (ldsMode == 1 ? traceKernel<true, false, true> : traceKernel<false, false, true>);
	else k = ldsMode == 2 ? traceKernel<true, true, false> : (ldsMode == 1 ? traceKernel<true, false, false> : traceKernel<false, false, false>);
	return launchWithLds(k, *p, grid, ldsBytes, stream);
}

extern "C" hipError_t acr_dev_launch_scatter_tiles(const void *tiles, const uint32_t *tileIdsDev, uint32_t numTiles, uint32_t tileW, uint32_t tileH,
                                                   uint32_t width, uint32_t height, void *frame, hipStream_t stream)
{
	using namespace acr::dev;
	const uint32_t tilesX = (width + tileW - 1) / tileW;
	hipLaunchKernelGGL(scatterTilesKernel, dim3(numTiles), dim3(kBlock), 0, stream, static_cast<const float4 *>(tiles), tileIdsDev, tileW, tileH,
	                   tilesX, width, height, static_cast<float4 *>(frame));
	return hipGetLastError();
}

#ifdef ACR_WALK_DEBUG
extern "C" int acr_dev_walk_debug(unsigned long long *out, int reset)
{
	unsigned long long z[16] = {0};
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(acr::dev::g_walkDebug), sizeof(z)) != hipSuccess) return 1;
	if (reset && hipMemcpyToSymbol(HIP_SYMBOL(acr::dev::g_walkDebug), z, sizeof(z)) != hipSuccess) return 1;
	return 0;
}
#endif

#ifdef ACR_WAVE_DEBUG
extern "C" int acr_dev_wave_debug(unsigned long long *out, int reset)
{
	unsigned long long z[16] = {0};
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(acr::dev::g_waveDebug), sizeof(z)) != hipSuccess) return 1;
	if (reset && hipMemcpyToSymbol(HIP_SYMBOL(acr::dev::g_waveDebug), z, sizeof(z)) != hipSuccess) return 1;
	return 0;
}
#endif
