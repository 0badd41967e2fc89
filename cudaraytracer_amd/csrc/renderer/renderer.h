// acr::Renderer (reference src/renderer/renderer.h:15-53), headless.
//
// Same public calls -- Renderer(Args), loadScene(scene), render(),
// moveCamera(pos, dir) -- minus the window: where the reference creates a
// GLUT window, a GL pixel buffer and maps it into CUDA (renderer.cu:38-128,
// 410,426), this one owns a context of the HIP core (include/acr_hip.h) whose
// framebuffer stands in for the pixel buffer.  Display is replaced by
// readFramebuffer()/save().
#ifndef ACR_RENDERER_RENDERER_H
#define ACR_RENDERER_RENDERER_H

#include <string>
#include <vector>

#include "../../../include/acr_hip.h"
#include "../math/math.h"
#include "../scene/scene.h"

namespace acr
{
	// the camera arithmetic of Renderer::moveCamera (renderer.cu:152-161) on its own
	Camera movedCamera(const Camera &camera, const math::vec2 &pos, const math::vec2 &dir);

	class Renderer
	{
	public:
		struct Args
		{
			const char *title;
			math::u32vec2 pos;
			math::u32vec3 dim; // width, height, samples: resolution = dim.xy * dim.z (renderer.cu:42-44)
			int device = 0;    // GPU ordinal (the reference takes the first GL-capable device, renderer.cu:76,91)
			// More than one entry: render every frame across these GPUs (the reference stops at maxNumDevices = 1,
			// renderer.cu:74): an acr_hip_group, bands of the frame per GPU, the frame assembled on devices[0] when it
			// is read.  One entry is the same as `device`.  A GPU may be named more than once.
			std::vector<int> devices;
		};

		explicit Renderer(const Args &args);
		~Renderer();
		Renderer(const Renderer &) = delete;
		Renderer &operator=(const Renderer &) = delete;

		void loadScene(const Scene &scene); // renderer.cu:168-182
		void render();                      // renderer.cu:397-423
		void moveCamera(const math::vec2 &pos, const math::vec2 &dir); // renderer.cu:145-166

		// what the reference hard-codes: MAX_BOUNCES (renderer.cu:10) and rand() as the per-frame seed (renderer.cu:421)
		void setMaxBounces(uint32_t n) { maxBounces = n; }
		void setFixedSeed(long long seed) { fixedSeed = seed; } // < 0: rand()
		void setCollectStats(bool on) { collectStats = on; }
		// The reference's adaptive assignment (renderer.cu:343-355,391-392,457-461; disabled at its v0, so off by
		// default): every frame's traversal-path keys are voted into the per-pixel aggregates (bih.h:29-66), every `every`
		// frames without a camera move the pixel table is re-sorted by the aggregates' majority keys; a move resets
		// table and aggregates.
		void setAdaptiveAssignment(bool on, uint32_t every = 60) { adaptive = on; pixelMapFrameMod = every ? every : 60; }
		// Several GPUs: how often the bands are re-dealt from measured costs (the adaptive-assignment cadence applied to
		// GPUs; default 16 frames, 0 = never), and whether every frame is assembled on the first GPU right away (for a
		// caller that shows every frame, as the reference does) or only when it is read (default).
		void setBalance(uint32_t rebalanceEvery);
		void setGatherEveryFrame(bool on) { gatherEveryFrame = on; }
		uint32_t numDevices() const;
		void setCamera(const Camera &cam); // places the camera directly; restarts accumulation like a move
		Camera getCamera() const;

		uint32_t width() const { return dim.x; }
		uint32_t height() const { return dim.y; }
		uint64_t framesWithoutMove() const { return framesNoMove; }
		acr_hip_ctx *context() { return ctx; }     // one GPU; nullptr when rendering across several
		acr_hip_group *group() { return grp; }     // several GPUs; nullptr otherwise

		void readFramebuffer(std::vector<float> &rgba);
		void readPrimitiveIds(std::vector<uint32_t> &ids);
		void readHitDistances(std::vector<float> &t);
		acr_stats stats();
		void save(const std::string &path); // .ppm (8-bit, clamped to [0,1]) or .pfm (float)

	private:
		void check(int rc, const char *what);

		const char *title;
		math::u32vec3 dim;
		uint64_t framesNoMove;
		uint32_t maxBounces = 5;
		long long fixedSeed = -1;
		bool collectStats = false;
		bool adaptive = false;
		uint32_t pixelMapFrameMod = 60; // renderer.cu:392
		acr_hip_ctx *ctx = nullptr;
		acr_hip_group *grp = nullptr;
		bool gatherEveryFrame = false;
	};
} // namespace acr

#endif
