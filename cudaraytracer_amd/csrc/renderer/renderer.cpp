#include "renderer.h"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace acr
{
	void Renderer::check(int rc, const char *what)
	{
		if (rc != ACR_OK)
			throw std::runtime_error(std::string(what) + ": " + (grp ? acr_hip_group_last_error(grp) : (ctx ? acr_hip_last_error(ctx) : "no context")) +
			                         " (code " + std::to_string(rc) + ")");
	}

	// renderer.cu:31-138 without the window, GL objects and pixel-key arrays
	Renderer::Renderer(const Renderer::Args &args) : title(args.title), dim(args.dim), framesNoMove(0)
	{
		dim.x *= dim.z;
		dim.y *= dim.z;
		dim.z = 1;
		if (args.devices.size() > 1) {
			// the reference's device loop keeps the first device (renderer.cu:73-104); this one keeps them all
			if (acr_hip_group_create(uint32_t(args.devices.size()), args.devices.data(), &grp) != ACR_OK)
				throw std::runtime_error("acr_hip_group_create failed: a GPU of the list is not usable");
			check(acr_hip_group_resize(grp, dim.x, dim.y), "acr_hip_group_resize");
			return;
		}
		const int device = args.devices.empty() ? args.device : args.devices[0];
		const int rc = acr_hip_create(device, &ctx);
		if (rc != ACR_OK)
			throw std::runtime_error("acr_hip_create failed: no usable GPU " + std::to_string(device)); // renderer.cu:77-104 exits here
		check(acr_hip_resize(ctx, dim.x, dim.y), "acr_hip_resize");
	}

	Renderer::~Renderer()
	{
		acr_hip_group_destroy(grp);
		acr_hip_destroy(ctx);
	}

	uint32_t Renderer::numDevices() const { return grp ? acr_hip_group_size(grp) : 1u; }

	void Renderer::setBalance(uint32_t rebalanceEvery)
	{
		if (grp) check(acr_hip_group_set_balance(grp, rebalanceEvery, nullptr), "acr_hip_group_set_balance");
	}

	// renderer.cu:168-182
	void Renderer::loadScene(const Scene &scene)
	{
		if (grp) check(acr_hip_group_upload_scene(grp, &scene.flatten()), "acr_hip_group_upload_scene");
		else check(acr_hip_upload_scene(ctx, &scene.flatten()), "acr_hip_upload_scene");
		Camera cam = scene.camera;
		cam.aspectRatio = float(dim.x) / float(dim.y);
		setCamera(cam);
	}

	void Renderer::setCamera(const Camera &cam)
	{
		acr_camera c{};
		c.aspect_ratio = cam.aspectRatio;
		c.horizontal_fov = cam.horizontalFOV;
		const math::vec3 *src[3] = {&cam.position, &cam.forward, &cam.up};
		float *dst[3] = {c.position, c.forward, c.up};
		for (int k = 0; k < 3; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
		if (grp) check(acr_hip_group_set_camera(grp, &c), "acr_hip_group_set_camera");
		else check(acr_hip_set_camera(ctx, &c), "acr_hip_set_camera");
		framesNoMove = 0;
	}

	Camera Renderer::getCamera() const
	{
		acr_camera c{};
		if ((grp ? acr_hip_group_get_camera(grp, &c) : acr_hip_get_camera(ctx, &c)) != ACR_OK) throw std::runtime_error("no camera set");
		Camera cam;
		cam.aspectRatio = c.aspect_ratio;
		cam.horizontalFOV = c.horizontal_fov;
		cam.position = math::vec3(c.position[0], c.position[1], c.position[2]);
		cam.forward = math::vec3(c.forward[0], c.forward[1], c.forward[2]);
		cam.up = math::vec3(c.up[0], c.up[1], c.up[2]);
		return cam;
	}

	// renderer.cu:152-161: yaw about up, pitch about right, then translate
	Camera movedCamera(const Camera &camera, const math::vec2 &pos, const math::vec2 &dir)
	{
		Camera cam = camera;
		cam.forward = math::rotate(cam.forward, dir.x, cam.up);
		const math::vec3 right = math::cross(cam.forward, cam.up);
		cam.forward = math::rotate(cam.forward, dir.y, right);
		cam.up = math::rotate(cam.up, dir.y, right);
		const math::vec3 delta = cam.forward * pos.y + right * pos.x;
		cam.position += delta;
		return cam;
	}

	// renderer.cu:145-166: read the camera back, move it, write it, restart accumulation
	void Renderer::moveCamera(const math::vec2 &pos, const math::vec2 &dir)
	{
		setCamera(movedCamera(getCamera(), pos, dir));
	}

	// renderer.cu:397-423
	void Renderer::render()
	{
		acr_frame f{};
		f.width = dim.x;
		f.height = dim.y;
		f.seed = fixedSeed >= 0 ? uint64_t(fixedSeed) + framesNoMove : uint64_t(rand());
		f.frames = framesNoMove;
		f.max_bounces = maxBounces;
		f.collect_stats = collectStats ? 1u : 0u;
		if (grp) {
			// the same frame, its bands dealt to the GPUs; the running mean of a band stays with the GPU that owns it
			if (adaptive) throw std::runtime_error("the pixel-level adaptive assignment is a single-GPU feature: the group assigns bands to GPUs by the same rules");
			check(acr_hip_group_render(grp, &f, gatherEveryFrame ? 1 : 0), "acr_hip_group_render");
			check(acr_hip_group_sync(grp), "acr_hip_group_sync"); // cudaDeviceSynchronize (renderer.cu:422), on every GPU
			framesNoMove++;
			return;
		}
		const bool resort = adaptive && (framesNoMove + 1) % pixelMapFrameMod == 0; // renderer.cu:457
		int32_t *votes = nullptr;
		if (adaptive) {
			f.pixel_values_dev = acr_hip_pixel_map(ctx);                // renderer.cu:343-355
			f.path_keys_dev = acr_hip_path_keys(ctx);                   // the Path of renderer.cu:341,372
			votes = acr_hip_path_votes(ctx);
			if (!votes) check(ACR_ERR_HIP, "acr_hip_path_votes");
		}
		check(acr_hip_render(ctx, &f, nullptr, nullptr, nullptr), "acr_hip_render");
		if (adaptive) // paths[...].set(path) on the first frame after a move, .add(path) afterwards (renderer.cu:380,386)
			check(acr_hip_vote_paths(ctx, acr_hip_path_keys(ctx), votes, dim.x * dim.y, framesNoMove == 0), "acr_hip_vote_paths");
		if (resort) { // renderer.cu:459-460: the aggregates as keys, then the sort
			check(acr_hip_vote_keys(ctx, votes, acr_hip_path_keys(ctx), dim.x * dim.y), "acr_hip_vote_keys");
			check(acr_hip_sort_pixel_map(ctx, acr_hip_path_keys(ctx), acr_hip_pixel_map(ctx), dim.x * dim.y), "acr_hip_sort_pixel_map");
		}
		check(acr_hip_sync(ctx), "acr_hip_sync"); // cudaDeviceSynchronize (renderer.cu:422)
		framesNoMove++;
	}

	void Renderer::readFramebuffer(std::vector<float> &rgba)
	{
		rgba.resize(size_t(dim.x) * dim.y * 4);
		if (grp) {
			check(acr_hip_group_read_framebuffer(grp, rgba.data(), rgba.size()), "acr_hip_group_read_framebuffer");
			return;
		}
		check(acr_hip_read_framebuffer(ctx, rgba.data(), rgba.size()), "acr_hip_read_framebuffer");
	}
	void Renderer::readPrimitiveIds(std::vector<uint32_t> &ids)
	{
		if (grp) throw std::runtime_error("primitive ids are not kept when rendering across several GPUs");
		ids.resize(size_t(dim.x) * dim.y);
		check(acr_hip_read_primid(ctx, ids.data(), ids.size()), "acr_hip_read_primid");
	}
	void Renderer::readHitDistances(std::vector<float> &t)
	{
		if (grp) throw std::runtime_error("hit distances are not kept when rendering across several GPUs");
		t.resize(size_t(dim.x) * dim.y);
		check(acr_hip_read_hit_t(ctx, t.data(), t.size()), "acr_hip_read_hit_t");
	}
	acr_stats Renderer::stats()
	{
		acr_stats s{};
		if (grp) {
			check(acr_hip_group_read_stats(grp, &s), "acr_hip_group_read_stats");
			return s;
		}
		check(acr_hip_read_stats(ctx, &s), "acr_hip_read_stats");
		return s;
	}

	void Renderer::save(const std::string &path)
	{
		std::vector<float> rgba;
		readFramebuffer(rgba);
		FILE *fp = std::fopen(path.c_str(), "wb");
		if (!fp) throw std::runtime_error("cannot write " + path);
		const bool pfm = path.size() > 4 && path.substr(path.size() - 4) == ".pfm";
		if (pfm) {
			// PFM stores rows bottom-up; image row 0 is the top row here
			std::fprintf(fp, "PF\n%u %u\n-1.0\n", dim.x, dim.y);
			for (uint32_t y = dim.y; y-- > 0;)
				for (uint32_t x = 0; x < dim.x; x++)
					std::fwrite(&rgba[4 * (size_t(y) * dim.x + x)], 4, 3, fp);
		} else {
			std::fprintf(fp, "P6\n%u %u\n255\n", dim.x, dim.y);
			for (size_t i = 0; i < size_t(dim.x) * dim.y; i++) {
				unsigned char px[3];
				for (int k = 0; k < 3; k++) {
					float v = rgba[4 * i + k];
					v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
					px[k] = (unsigned char)(v * 255.0f + 0.5f);
				}
				std::fwrite(px, 1, 3, fp);
			}
		}
		std::fclose(fp);
	}
} // namespace acr
