// acr_hip_group_*: one frame across several GPUs, driven by one host thread through the C ABI of include/acr_hip.h.
//
// The reference renders on one device (`maxNumDevices = 1`, renderer.cu:74); this is the multi-GPU form of the same
// render loop (renderer.cu:397-423) for a C or C++ host.  Pixels are independent, so a frame needs no exchange while
// it renders: the scene is replicated, the frame is cut into full-width bands of 16 rows, member i renders a
// contiguous run of bands into a frame-shaped buffer of its own (acr_frame::tiles_into_frame), and the pieces travel
// to the first member's GPU -- peer-to-peer copies, each sender over its own xGMI link -- when the caller asks for
// the frame.  Which bands a member renders follows the rules of the reference's adaptive assignment
// (renderer.cu:343-355,457-461,165): an explicit table (the kernel's tile list), rebuilt on a cadence from costs
// measured while rendering (wave clock ticks per band, acr_frame::tile_cycles_dev), reset to the uniform split when
// the camera moves.
//
// Progressive accumulation (frames > 0, renderer.cu:375-387) keeps every band's running mean with the member that
// owns the band: nothing moves per frame; when a rebalance changes a band's owner its running mean moves with it.
//
// Built into libacr_hip.so; uses nothing of the single-context implementation but its public entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/acr_hip.h"

namespace
{
	constexpr uint32_t kBandRows = 16;
	constexpr int kDepth = 2;      // frame buffers per member: frames that overwrite (frames == 0) alternate between them
	constexpr uint32_t kCostLag = 2; // costs measured on frame k decide the split from frame k + kCostLag on

	struct Member
	{
		int device = 0;
		acr_hip_ctx *ctx = nullptr;
		hipStream_t render = nullptr, copy = nullptr;
		float4 *frame[kDepth] = {nullptr, nullptr};
		hipEvent_t rendered = nullptr;          // behind the member's last render (and whatever it queued before it)
		hipEvent_t copied[kDepth] = {nullptr, nullptr}; // behind the last copy that read this member's frame[k] (gather or move)
		bool copyPending[kDepth] = {false, false};
		uint32_t *costDev = nullptr;            // [bands] clock ticks per band of the member's last measuring frame, in list order
		uint32_t *costHost = nullptr;           // pinned
		hipEvent_t costReady = nullptr;
		std::vector<uint32_t> tiles;            // the member's bands for the frame being queued
	};

	struct CostUpdate
	{
		uint64_t applyAt;                // frame number from which the split built from these costs is used
		std::vector<uint32_t> begin;     // the split the costs were measured with
	};
}

struct acr_hip_group
{
	std::vector<Member> m;
	std::string error;
	uint32_t width = 0, height = 0, bands = 0;
	bool haveScene = false, haveCamera = false;
	acr_camera camera{};
	std::vector<uint32_t> begin;      // member i renders bands [begin[i], begin[i + 1])
	std::vector<uint32_t> owned;      // the split the running means in frame[cur] are distributed by (empty: none)
	std::vector<float> cost;          // per band, of the last measured frame (empty: uniform)
	std::vector<float> weights;       // share of the total cost per member (empty: equal shares)
	uint32_t rebalanceEvery = 16;
	uint32_t sinceRebalance = 0;
	uint64_t frameNo = 0;             // frames queued so far
	int cur = 0;                      // frame buffer of the last queued frame
	bool gathered = false;            // the first member's frame[cur] holds every member's bands of the last queued frame
	bool rendered = false;
	std::vector<CostUpdate> pending;  // at most one measurement in flight
	bool costInFlight = false;
};

namespace
{
	int gfail(acr_hip_group *g, int code, const char *fmt, ...)
	{
		char buf[512];
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(buf, sizeof(buf), fmt, ap);
		va_end(ap);
		if (g) g->error = buf;
		return code;
	}

#define ACR_GRP_TRY(g, expr)                                                                             \
	do {                                                                                                 \
		hipError_t e_ = (expr);                                                                          \
		if (e_ != hipSuccess) return gfail(g, ACR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
	} while (0)

	// a member's call failed: its message becomes the group's
	int memberFail(acr_hip_group *g, uint32_t i, int rc, const char *what)
	{
		return gfail(g, rc, "member %u (GPU %d): %s: %s", i, g->m[i].device, what, acr_hip_last_error(g->m[i].ctx));
	}

	// Cuts the bands 0..n-1 into contiguous runs, one per member, whose summed costs are in the ratio of the members'
	// weights; no member with a share is left without a band while there are enough bands.
	void partitionBands(const std::vector<float> &cost, uint32_t n, const std::vector<float> &weights, uint32_t parts, std::vector<uint32_t> &begin)
	{
		begin.assign(parts + 1, 0);
		auto costAt = [&](uint32_t b) { return cost.empty() ? 1.0 : double(cost[b] > 0 ? cost[b] : 0); };
		auto weightOf = [&](uint32_t k) { return weights.empty() ? 1.0 : double(weights[k]); };
		double total = 0, wsum = 0;
		for (uint32_t b = 0; b < n; b++) total += costAt(b);
		for (uint32_t k = 0; k < parts; k++) wsum += weightOf(k);
		uint32_t at = 0;
		double run = 0, wrun = 0;
		for (uint32_t k = 1; k < parts; k++) {
			wrun += weightOf(k - 1);
			const double target = wsum > 0 ? total * wrun / wsum : 0;
			// stop before the band that would overshoot the target by more than it undershoots
			while (at < n && run + costAt(at) * 0.5 <= target) run += costAt(at++);
			if (n >= parts) {
				if (weightOf(k - 1) > 0)
					while (at < begin[k - 1] + 1) run += costAt(at++);
				uint32_t follow = 0; // members after this cut that have a share
				for (uint32_t j = k; j < parts; j++) follow += weightOf(j) > 0 ? 1u : 0u;
				while (at > n - follow) run -= costAt(--at);
			}
			begin[k] = at;
		}
		begin[parts] = n;
	}

	// pixel range [first, last) of bands [b0, b1) in a frame-shaped buffer (the last band may be cut by the frame's bottom edge)
	void bandPixels(const acr_hip_group *g, uint32_t b0, uint32_t b1, size_t &first, size_t &last)
	{
		const size_t rowPx = g->width, total = size_t(g->width) * g->height;
		first = std::min(size_t(b0) * kBandRows * rowPx, total);
		last = std::min(size_t(b1) * kBandRows * rowPx, total);
	}

	// Copies the pixels of bands [b0, b1) from member `from`'s frame[slot] to member `to`'s frame[slot], on `stream`
	// (which belongs to the sender's device: the sender pushes over its own link).
	hipError_t copyBands(acr_hip_group *g, uint32_t from, uint32_t to, int slot, uint32_t b0, uint32_t b1, hipStream_t stream)
	{
		size_t first, last;
		bandPixels(g, b0, b1, first, last);
		if (last <= first) return hipSuccess;
		const Member &s = g->m[from], &d = g->m[to];
		const size_t bytes = (last - first) * sizeof(float4);
		if (s.device == d.device) return hipMemcpyAsync(d.frame[slot] + first, s.frame[slot] + first, bytes, hipMemcpyDeviceToDevice, stream);
		return hipMemcpyPeerAsync(d.frame[slot] + first, d.device, s.frame[slot] + first, s.device, bytes, stream);
	}

	void freeMember(Member &mb)
	{
		if (!mb.ctx) { // creation stopped before this member (or at it: then its device may not even exist)
			(void)hipGetLastError();
			mb = Member();
			return;
		}
		(void)hipSetDevice(mb.device);
		if (mb.render) (void)hipStreamSynchronize(mb.render);
		if (mb.copy) (void)hipStreamSynchronize(mb.copy);
		if (mb.ctx) acr_hip_destroy(mb.ctx);
		for (int k = 0; k < kDepth; k++) {
			(void)hipFree(mb.frame[k]);
			if (mb.copied[k]) (void)hipEventDestroy(mb.copied[k]);
		}
		(void)hipFree(mb.costDev);
		(void)hipHostFree(mb.costHost);
		if (mb.rendered) (void)hipEventDestroy(mb.rendered);
		if (mb.costReady) (void)hipEventDestroy(mb.costReady);
		if (mb.render) (void)hipStreamDestroy(mb.render);
		if (mb.copy) (void)hipStreamDestroy(mb.copy);
		mb = Member();
	}

	// Applies the measured costs that are due at the frame about to be queued; returns whether the split changed.
	int applyCosts(acr_hip_group *g)
	{
		if (g->pending.empty() || g->pending.front().applyAt > g->frameNo) return ACR_OK;
		const CostUpdate u = g->pending.front();
		g->pending.clear();
		g->costInFlight = false;
		std::vector<float> cost(g->bands, 0.0f);
		for (uint32_t i = 0; i < g->m.size(); i++) {
			Member &mb = g->m[i];
			const uint32_t b0 = u.begin[i], b1 = u.begin[i + 1];
			if (b1 == b0) continue;
			ACR_GRP_TRY(g, hipSetDevice(mb.device));
			ACR_GRP_TRY(g, hipEventSynchronize(mb.costReady)); // queued kCostLag frames ago: long over
			for (uint32_t b = b0; b < b1; b++) cost[b] = float(mb.costHost[b - b0]);
		}
		g->cost = cost;
		partitionBands(g->cost, g->bands, g->weights, uint32_t(g->m.size()), g->begin);
		return ACR_OK;
	}

	// The running means in frame[cur] are distributed by g->owned, the next frame will be rendered with g->begin:
	// every band that changes owner travels from the old owner's buffer to the new owner's.  The copies are queued on
	// the old owners' render streams (behind their last frame); the new owners' next frames wait for them.
	int moveRunningMeans(acr_hip_group *g)
	{
		const uint32_t n = uint32_t(g->m.size());
		// a gather may still be reading or writing these buffers
		for (uint32_t i = 0; i < n; i++) {
			Member &mb = g->m[i];
			if (!mb.copyPending[g->cur]) continue;
			for (uint32_t j = 0; j < n; j++) {
				ACR_GRP_TRY(g, hipSetDevice(g->m[j].device));
				ACR_GRP_TRY(g, hipStreamWaitEvent(g->m[j].render, mb.copied[g->cur], 0));
			}
			mb.copyPending[g->cur] = false;
		}
		std::vector<bool> sent(n, false);
		for (uint32_t i = 0; i < n; i++) {     // old owner
			for (uint32_t j = 0; j < n; j++) { // new owner
				if (i == j) continue;
				const uint32_t b0 = std::max(g->owned[i], g->begin[j]), b1 = std::min(g->owned[i + 1], g->begin[j + 1]);
				if (b1 <= b0) continue;
				ACR_GRP_TRY(g, hipSetDevice(g->m[i].device));
				ACR_GRP_TRY(g, copyBands(g, i, j, g->cur, b0, b1, g->m[i].render));
				sent[i] = true;
			}
		}
		for (uint32_t i = 0; i < n; i++) {
			if (!sent[i]) continue;
			ACR_GRP_TRY(g, hipSetDevice(g->m[i].device));
			ACR_GRP_TRY(g, hipEventRecord(g->m[i].rendered, g->m[i].render));
			for (uint32_t j = 0; j < n; j++) {
				if (j == i) continue;
				ACR_GRP_TRY(g, hipSetDevice(g->m[j].device));
				ACR_GRP_TRY(g, hipStreamWaitEvent(g->m[j].render, g->m[i].rendered, 0));
			}
		}
		return ACR_OK;
	}

	// Queues the copies that bring every other member's bands of the last frame into the first member's frame[cur].
	int queueGather(acr_hip_group *g)
	{
		if (g->gathered || !g->rendered) return ACR_OK;
		const int slot = g->cur;
		const std::vector<uint32_t> &split = g->owned;
		for (uint32_t i = 1; i < g->m.size(); i++) {
			Member &mb = g->m[i];
			if (split[i + 1] == split[i]) continue;
			ACR_GRP_TRY(g, hipSetDevice(mb.device));
			ACR_GRP_TRY(g, hipStreamWaitEvent(mb.copy, mb.rendered, 0));
			ACR_GRP_TRY(g, copyBands(g, i, 0, slot, split[i], split[i + 1], mb.copy));
			ACR_GRP_TRY(g, hipEventRecord(mb.copied[slot], mb.copy));
			mb.copyPending[slot] = true;
		}
		g->gathered = true;
		return ACR_OK;
	}
} // namespace

extern "C" {

int acr_hip_group_create(uint32_t n, const int *devices, acr_hip_group **out)
{
	if (!out) return ACR_ERR_INVALID;
	*out = nullptr;
	if (n == 0 || n > 64 || !devices) return ACR_ERR_INVALID;
	acr_hip_group *g = new acr_hip_group();
	g->m.resize(n);
	for (uint32_t i = 0; i < n; i++) {
		Member &mb = g->m[i];
		mb.device = devices[i];
		bool ok = acr_hip_create(devices[i], &mb.ctx) == ACR_OK && mb.ctx && hipSetDevice(devices[i]) == hipSuccess &&
		          hipStreamCreateWithFlags(&mb.render, hipStreamNonBlocking) == hipSuccess &&
		          hipStreamCreateWithFlags(&mb.copy, hipStreamNonBlocking) == hipSuccess &&
		          hipEventCreateWithFlags(&mb.rendered, hipEventDisableTiming) == hipSuccess &&
		          hipEventCreateWithFlags(&mb.costReady, hipEventDisableTiming) == hipSuccess;
		for (int k = 0; k < kDepth && ok; k++) ok = hipEventCreateWithFlags(&mb.copied[k], hipEventDisableTiming) == hipSuccess;
		if (ok) ok = acr_hip_set_stream(mb.ctx, mb.render) == ACR_OK;
		if (!ok) {
			acr_hip_group_destroy(g);
			return ACR_ERR_HIP;
		}
	}
	// the pieces travel peer to peer: let every member's GPU write into the first member's, and the members into one
	// another's for running means that change owner (a refusal is not an error: the runtime then stages such copies)
	for (uint32_t i = 0; i < n; i++)
		for (uint32_t j = 0; j < n; j++) {
			if (devices[i] == devices[j]) continue;
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can && hipSetDevice(devices[i]) == hipSuccess) {
				const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
				if (e != hipSuccess) (void)hipGetLastError(); // already enabled, or refused
			}
		}
	*out = g;
	return ACR_OK;
}

void acr_hip_group_destroy(acr_hip_group *g)
{
	if (!g) return;
	for (Member &mb : g->m) freeMember(mb);
	delete g;
}

const char *acr_hip_group_last_error(const acr_hip_group *g) { return g ? g->error.c_str() : "null group"; }
uint32_t acr_hip_group_size(const acr_hip_group *g) { return g ? uint32_t(g->m.size()) : 0u; }
acr_hip_ctx *acr_hip_group_context(acr_hip_group *g, uint32_t i) { return (g && i < g->m.size()) ? g->m[i].ctx : nullptr; }

int acr_hip_group_sync(acr_hip_group *g)
{
	if (!g) return ACR_ERR_INVALID;
	for (Member &mb : g->m) {
		ACR_GRP_TRY(g, hipSetDevice(mb.device));
		ACR_GRP_TRY(g, hipStreamSynchronize(mb.render));
		ACR_GRP_TRY(g, hipStreamSynchronize(mb.copy));
		for (int k = 0; k < kDepth; k++) mb.copyPending[k] = false;
	}
	return ACR_OK;
}

int acr_hip_group_upload_scene(acr_hip_group *g, const acr_scene_soa *scene)
{
	if (!g || !scene) return gfail(g, ACR_ERR_INVALID, "null argument");
	for (uint32_t i = 0; i < g->m.size(); i++) {
		const int rc = acr_hip_upload_scene(g->m[i].ctx, scene);
		if (rc != ACR_OK) return memberFail(g, i, rc, "acr_hip_upload_scene");
	}
	g->haveScene = true;
	g->cost.clear(); // another scene: no history applies
	if (g->bands) partitionBands(g->cost, g->bands, g->weights, uint32_t(g->m.size()), g->begin);
	return ACR_OK;
}

int acr_hip_group_set_camera(acr_hip_group *g, const acr_camera *camera)
{
	if (!g || !camera) return gfail(g, ACR_ERR_INVALID, "null argument");
	const bool moved = !g->haveCamera || memcmp(&g->camera, camera, sizeof(*camera)) != 0;
	for (uint32_t i = 0; i < g->m.size(); i++) {
		const int rc = acr_hip_set_camera(g->m[i].ctx, camera);
		if (rc != ACR_OK) return memberFail(g, i, rc, "acr_hip_set_camera");
	}
	g->camera = *camera;
	g->haveCamera = true;
	if (moved) {
		// no history applies to the new view (renderer.cu:165,344-348): uniform costs again; costs still on their way
		// belong to the old view; the members' shares are a property of the machine and stay
		g->cost.clear();
		g->pending.clear();
		g->costInFlight = false;
		g->sinceRebalance = 0;
		// (running means of the old view are not moved to the new split: the caller's next frame overwrites them, and
		// if it does not -- frames > 0 after a move -- acr_hip_group_render moves them like after any other rebalance)
		if (g->bands) partitionBands(g->cost, g->bands, g->weights, uint32_t(g->m.size()), g->begin);
	}
	return ACR_OK;
}

int acr_hip_group_get_camera(const acr_hip_group *g, acr_camera *camera)
{
	if (!g || !camera || !g->haveCamera) return ACR_ERR_INVALID;
	*camera = g->camera;
	return ACR_OK;
}

int acr_hip_group_resize(acr_hip_group *g, uint32_t width, uint32_t height)
{
	if (!g || width == 0 || height == 0 || uint64_t(width) * height >= (1ull << 31)) return gfail(g, ACR_ERR_INVALID, "bad frame size");
	const int rc = acr_hip_group_sync(g);
	if (rc != ACR_OK) return rc;
	const uint32_t bands = (height + kBandRows - 1) / kBandRows;
	const size_t bytes = size_t(width) * height * sizeof(float4);
	for (Member &mb : g->m) {
		ACR_GRP_TRY(g, hipSetDevice(mb.device));
		for (int k = 0; k < kDepth; k++) {
			(void)hipFree(mb.frame[k]);
			mb.frame[k] = nullptr;
		}
		(void)hipFree(mb.costDev);
		(void)hipHostFree(mb.costHost);
		mb.costDev = mb.costHost = nullptr;
		for (int k = 0; k < kDepth; k++) {
			ACR_GRP_TRY(g, hipMalloc(reinterpret_cast<void **>(&mb.frame[k]), bytes));
			ACR_GRP_TRY(g, hipMemset(mb.frame[k], 0, bytes));
		}
		ACR_GRP_TRY(g, hipMalloc(reinterpret_cast<void **>(&mb.costDev), sizeof(uint32_t) * bands));
		ACR_GRP_TRY(g, hipHostMalloc(reinterpret_cast<void **>(&mb.costHost), sizeof(uint32_t) * bands, hipHostMallocDefault));
	}
	g->width = width;
	g->height = height;
	g->bands = bands;
	g->cost.clear();
	g->pending.clear();
	g->costInFlight = false;
	g->owned.clear();
	g->rendered = g->gathered = false;
	g->sinceRebalance = 0;
	partitionBands(g->cost, g->bands, g->weights, uint32_t(g->m.size()), g->begin);
	return ACR_OK;
}

int acr_hip_group_set_balance(acr_hip_group *g, uint32_t rebalance_every, const float *weights)
{
	if (!g) return ACR_ERR_INVALID;
	if (weights) {
		double sum = 0;
		for (size_t i = 0; i < g->m.size(); i++) {
			if (!(weights[i] >= 0)) return gfail(g, ACR_ERR_INVALID, "weights must be non-negative numbers");
			sum += weights[i];
		}
		if (!(sum > 0)) return gfail(g, ACR_ERR_INVALID, "at least one member needs a share");
		g->weights.assign(weights, weights + g->m.size());
	} else {
		g->weights.clear();
	}
	g->rebalanceEvery = rebalance_every;
	g->sinceRebalance = 0;
	if (g->bands) partitionBands(g->cost, g->bands, g->weights, uint32_t(g->m.size()), g->begin);
	return ACR_OK;
}

int acr_hip_group_split(const acr_hip_group *g, uint32_t *begin, uint32_t *num_bands)
{
	if (!g || !g->bands) return ACR_ERR_INVALID;
	if (begin) memcpy(begin, g->begin.data(), sizeof(uint32_t) * g->begin.size());
	if (num_bands) *num_bands = g->bands;
	return ACR_OK;
}

int acr_hip_group_render(acr_hip_group *g, const acr_frame *f, int gather)
{
	if (!g || !f) return gfail(g, ACR_ERR_INVALID, "null argument");
	if (!g->haveScene || !g->haveCamera) return gfail(g, ACR_ERR_NO_SCENE, "render before scene and camera were set");
	if (!g->bands || f->width != g->width || f->height != g->height)
		return gfail(g, ACR_ERR_INVALID, "the frame's size must be that of acr_hip_group_resize");
	if (f->num_tiles || f->tile_ids || f->pixel_values_dev || f->tile_rays_dev || f->tile_cycles_dev || f->path_keys_dev || f->block_order_dev)
		return gfail(g, ACR_ERR_INVALID, "a group cuts the frame itself: tile list, pixel table, per-tile costs, path keys and block order must be unset");
	const uint32_t n = uint32_t(g->m.size());
	const bool accumulate = f->frames > 0;

	int rc = applyCosts(g);
	if (rc != ACR_OK) return rc;
	if (accumulate) {
		if (!g->rendered) return gfail(g, ACR_ERR_INVALID, "frames > 0 needs an earlier frame to average with");
		if (g->owned != g->begin) {
			rc = moveRunningMeans(g);
			if (rc != ACR_OK) return rc;
		}
	} else {
		g->cur = (g->cur + 1) % kDepth; // the previous frame's pieces may still be travelling out of the other buffer
	}
	const int slot = g->cur;
	g->owned = g->begin;

	// measure this frame's cost per band?
	const bool measure = n > 1 && g->rebalanceEvery != 0 && !g->costInFlight && g->sinceRebalance + 1 >= g->rebalanceEvery;

	// what last touched the buffers this frame is rendered into: copies out of them (every member's own), and for the
	// first member also the other members' copies INTO its buffer (under whatever split that frame had)
	for (uint32_t i = 0; i < n; i++) {
		Member &mb = g->m[i];
		if (!mb.copyPending[slot]) continue;
		ACR_GRP_TRY(g, hipSetDevice(mb.device));
		ACR_GRP_TRY(g, hipStreamWaitEvent(mb.render, mb.copied[slot], 0));
		if (i != 0) {
			ACR_GRP_TRY(g, hipSetDevice(g->m[0].device));
			ACR_GRP_TRY(g, hipStreamWaitEvent(g->m[0].render, mb.copied[slot], 0));
		}
		mb.copyPending[slot] = false;
	}

	const uint32_t tileW = (g->width + 15u) & ~15u; // one tile per band, as wide as the frame
	for (uint32_t i = 0; i < n; i++) {
		Member &mb = g->m[i];
		const uint32_t b0 = g->begin[i], b1 = g->begin[i + 1];
		if (b1 == b0) continue; // (an empty tile list would mean "the whole frame" to acr_hip_render)
		mb.tiles.resize(b1 - b0);
		for (uint32_t b = b0; b < b1; b++) mb.tiles[b - b0] = b;
		acr_frame mf = *f;
		mf.tile_w = tileW;
		mf.tile_h = kBandRows;
		mf.num_tiles = b1 - b0;
		mf.tile_ids = mb.tiles.data();
		mf.tiles_into_frame = 1;
		mf.tile_cycles_dev = measure ? mb.costDev : nullptr;
		rc = acr_hip_render(mb.ctx, &mf, mb.frame[slot], nullptr, nullptr);
		if (rc != ACR_OK) return memberFail(g, i, rc, "acr_hip_render");
		ACR_GRP_TRY(g, hipSetDevice(mb.device));
		if (measure) {
			ACR_GRP_TRY(g, hipMemcpyAsync(mb.costHost, mb.costDev, sizeof(uint32_t) * (b1 - b0), hipMemcpyDeviceToHost, mb.render));
			ACR_GRP_TRY(g, hipEventRecord(mb.costReady, mb.render));
		}
		ACR_GRP_TRY(g, hipEventRecord(mb.rendered, mb.render));
	}
	if (measure) {
		g->pending.push_back(CostUpdate{g->frameNo + kCostLag, g->begin});
		g->costInFlight = true;
		g->sinceRebalance = 0;
	} else {
		g->sinceRebalance++;
	}
	g->frameNo++;
	g->rendered = true;
	g->gathered = false;
	if (gather) return queueGather(g);
	return ACR_OK;
}

int acr_hip_group_frame(acr_hip_group *g, void **frame_dev)
{
	if (!g || !frame_dev) return gfail(g, ACR_ERR_INVALID, "null argument");
	*frame_dev = nullptr;
	if (!g->rendered) return gfail(g, ACR_ERR_INVALID, "no frame has been rendered");
	int rc = queueGather(g);
	if (rc != ACR_OK) return rc;
	const int slot = g->cur;
	for (uint32_t i = 0; i < g->m.size(); i++) {
		Member &mb = g->m[i];
		ACR_GRP_TRY(g, hipSetDevice(mb.device));
		if (i == 0) ACR_GRP_TRY(g, hipStreamSynchronize(mb.render)); // the first member's own bands
		if (mb.copyPending[slot]) {
			ACR_GRP_TRY(g, hipEventSynchronize(mb.copied[slot]));
			mb.copyPending[slot] = false;
		}
	}
	*frame_dev = g->m[0].frame[slot];
	return ACR_OK;
}

int acr_hip_group_read_framebuffer(acr_hip_group *g, float *rgba, size_t num_floats)
{
	if (!g || !rgba) return gfail(g, ACR_ERR_INVALID, "null argument");
	if (num_floats != size_t(g->width) * g->height * 4) return gfail(g, ACR_ERR_INVALID, "size mismatch");
	void *dev = nullptr;
	const int rc = acr_hip_group_frame(g, &dev);
	if (rc != ACR_OK) return rc;
	ACR_GRP_TRY(g, hipSetDevice(g->m[0].device));
	ACR_GRP_TRY(g, hipMemcpy(rgba, dev, num_floats * sizeof(float), hipMemcpyDeviceToHost));
	return ACR_OK;
}

int acr_hip_group_read_stats(acr_hip_group *g, acr_stats *stats)
{
	if (!g || !stats) return gfail(g, ACR_ERR_INVALID, "null argument");
	acr_stats sum{};
	for (uint32_t i = 0; i < g->m.size(); i++) {
		if (g->owned.size() == g->m.size() + 1 && g->owned[i + 1] == g->owned[i]) continue; // rendered nothing of the last frame
		acr_stats s{};
		const int rc = acr_hip_read_stats(g->m[i].ctx, &s);
		if (rc != ACR_OK) return memberFail(g, i, rc, "acr_hip_read_stats");
		sum.rays_primary += s.rays_primary;
		sum.rays_shadow += s.rays_shadow;
		sum.rays_bounce += s.rays_bounce;
		sum.primary_hits += s.primary_hits;
		sum.six_plane_pixels += s.six_plane_pixels;
		sum.grid_blocks += s.grid_blocks;
		if (s.kernel_ms > sum.kernel_ms) sum.kernel_ms = s.kernel_ms; // the members render side by side
		sum.lds_bytes = s.lds_bytes;
		sum.scene_in_lds = s.scene_in_lds;
		sum.interval_traversal = s.interval_traversal;
		sum.block_order = s.block_order;
		sum.near_first = s.near_first;
		sum.wave_tasks = s.wave_tasks;
		sum.shared_walk = s.shared_walk;
	}
	*stats = sum;
	return ACR_OK;
}

} // extern "C"
