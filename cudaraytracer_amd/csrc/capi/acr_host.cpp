// C view of the C++ host side (include/acr_host.h).
#include "../../../include/acr_host.h"

#include <cstdio>
#include <cstring>
#include <algorithm>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "../loader/import_scene.h"
#include "../math/math.h"
#include "../math/rng.h"
#include "../renderer/renderer.h"
#include "../scene/scene.h"

using namespace acr;

struct acr_host_import
{
	ImportScene scene;
	acr_import_scene view;
};
struct acr_host_scene
{
	Scene *scene = nullptr;
};
struct acr_host_renderer
{
	Renderer *renderer = nullptr;
	std::string error;
};

namespace
{
	void setErr(char *err, size_t n, const std::string &msg)
	{
		if (err && n) {
			std::snprintf(err, n, "%s", msg.c_str());
		}
	}
	template <class F>
	int guarded(acr_host_renderer *r, F f)
	{
		if (!r || !r->renderer) return ACR_ERR_INVALID;
		try {
			f();
			return ACR_OK;
		} catch (const std::exception &e) {
			r->error = e.what();
			return ACR_ERR_HIP;
		}
	}
} // namespace

extern "C" {

acr_host_import *acr_host_import_blend(const char *path, char *err, size_t err_len)
{
	acr_host_import *imp = new acr_host_import();
	std::string e;
	if (!path || !loadBlend(path, imp->scene, e)) {
		setErr(err, err_len, path ? e : "null path");
		delete imp;
		return nullptr;
	}
	imp->view = imp->scene.view();
	return imp;
}

acr_host_import *acr_host_import_soup(uint32_t num_triangles, uint32_t seed, float edge, float extent)
{
	acr_host_import *imp = new acr_host_import();
	makeSoup(num_triangles, seed, edge, extent, imp->scene);
	imp->view = imp->scene.view();
	return imp;
}

const acr_import_scene *acr_host_import_view(const acr_host_import *imp) { return imp ? &imp->view : nullptr; }
void acr_host_import_destroy(acr_host_import *imp) { delete imp; }

acr_host_scene *acr_host_scene_create(const acr_import_scene *imp, uint32_t scene_bih_depth, uint32_t mesh_bih_depth, char *err, size_t err_len)
{
	if (!imp || scene_bih_depth < 1 || scene_bih_depth > ACR_MAX_BIH_DEPTH || mesh_bih_depth < 1 || mesh_bih_depth > ACR_MAX_BIH_DEPTH) {
		setErr(err, err_len, "null scene or BIH depth out of range");
		return nullptr;
	}
	try {
		acr_host_scene *s = new acr_host_scene();
		s->scene = new Scene(*imp, scene_bih_depth, mesh_bih_depth);
		return s;
	} catch (const std::exception &e) {
		setErr(err, err_len, e.what());
		return nullptr;
	}
}

acr_host_scene *acr_host_scene_load(const char *blend_path, uint32_t scene_bih_depth, uint32_t mesh_bih_depth, char *err, size_t err_len)
{
	if (!blend_path || scene_bih_depth < 1 || scene_bih_depth > ACR_MAX_BIH_DEPTH || mesh_bih_depth < 1 || mesh_bih_depth > ACR_MAX_BIH_DEPTH) {
		setErr(err, err_len, "null path or BIH depth out of range");
		return nullptr;
	}
	try {
		Scene::Args args;
		args.filePath = blend_path;
		args.sceneBihDepth = scene_bih_depth;
		args.meshBihDepth = mesh_bih_depth;
		acr_host_scene *s = new acr_host_scene();
		s->scene = new Scene(args);
		return s;
	} catch (const std::exception &e) {
		setErr(err, err_len, e.what());
		return nullptr;
	}
}

void acr_host_scene_destroy(acr_host_scene *scene)
{
	if (!scene) return;
	delete scene->scene;
	delete scene;
}

int acr_host_scene_soa(const acr_host_scene *scene, acr_scene_soa *out)
{
	if (!scene || !scene->scene || !out) return ACR_ERR_INVALID;
	*out = scene->scene->flatten();
	return ACR_OK;
}

int acr_host_scene_camera(const acr_host_scene *scene, acr_camera *out)
{
	if (!scene || !scene->scene || !out) return ACR_ERR_INVALID;
	const Camera &c = scene->scene->camera;
	out->aspect_ratio = c.aspectRatio;
	out->horizontal_fov = c.horizontalFOV;
	out->position[0] = c.position.x; out->position[1] = c.position.y; out->position[2] = c.position.z;
	out->forward[0] = c.forward.x; out->forward[1] = c.forward.y; out->forward[2] = c.forward.z;
	out->up[0] = c.up.x; out->up[1] = c.up.y; out->up[2] = c.up.z;
	return ACR_OK;
}

int acr_host_scene_object_order(const acr_host_scene *scene, int32_t *order, uint32_t count)
{
	if (!scene || !scene->scene || !order) return ACR_ERR_INVALID;
	const vector<Object> &objs = scene->scene->getObjects().items();
	if (count != objs.size()) return ACR_ERR_INVALID;
	for (uint32_t i = 0; i < count; i++) order[i] = objs[i].index;
	return ACR_OK;
}

int acr_host_scene_face_order(const acr_host_scene *scene, uint32_t mesh, uint32_t *order, uint32_t count)
{
	if (!scene || !scene->scene || !order || mesh >= scene->scene->meshes.size()) return ACR_ERR_INVALID;
	const std::vector<uint32_t> &o = scene->scene->meshes[mesh].getFaces().itemOrder();
	if (count != o.size()) return ACR_ERR_INVALID;
	std::memcpy(order, o.data(), 4 * size_t(count));
	return ACR_OK;
}

int acr_host_scene_bih_text(const acr_host_scene *scene, uint32_t mesh, char *buf, size_t buf_len)
{
	if (!scene || !scene->scene || !buf || !buf_len) return ACR_ERR_INVALID;
	std::string s;
	if (mesh == UINT32_MAX) s = scene->scene->getObjects().toString();
	else if (mesh < scene->scene->meshes.size()) s = scene->scene->meshes[mesh].getFaces().toString();
	else return ACR_ERR_INVALID;
	std::snprintf(buf, buf_len, "%s", s.c_str());
	return ACR_OK;
}

int acr_host_scene_bih_defined(const acr_host_scene *scene, uint32_t mesh, uint8_t *defined, uint32_t count)
{
	if (!scene || !scene->scene || !defined) return ACR_ERR_INVALID;
	if (mesh == UINT32_MAX) {
		const auto &n = scene->scene->getObjects().nodes();
		if (count != n.size()) return ACR_ERR_INVALID;
		for (uint32_t i = 0; i < count; i++) defined[i] = n[i].defined;
		return ACR_OK;
	}
	if (mesh >= scene->scene->meshes.size()) return ACR_ERR_INVALID;
	const auto &n = scene->scene->meshes[mesh].getFaces().nodes();
	if (count != n.size()) return ACR_ERR_INVALID;
	for (uint32_t i = 0; i < count; i++) defined[i] = n[i].defined;
	return ACR_OK;
}

acr_host_renderer *acr_host_renderer_create(uint32_t width, uint32_t height, uint32_t samples, int device, char *err, size_t err_len)
{
	try {
		Renderer::Args args;
		args.title = "acr";
		args.pos = {0, 0};
		args.dim = {width, height, samples ? samples : 1};
		args.device = device;
		acr_host_renderer *r = new acr_host_renderer();
		r->renderer = new Renderer(args);
		return r;
	} catch (const std::exception &e) {
		setErr(err, err_len, e.what());
		return nullptr;
	}
}

acr_host_renderer *acr_host_renderer_create_multi(uint32_t width, uint32_t height, uint32_t samples, const int *devices, uint32_t num_devices,
                                                  uint32_t rebalance_every, char *err, size_t err_len)
{
	try {
		if (!devices || num_devices == 0) throw std::runtime_error("no devices given");
		Renderer::Args args;
		args.title = "acr";
		args.pos = {0, 0};
		args.dim = {width, height, samples ? samples : 1};
		args.devices.assign(devices, devices + num_devices);
		acr_host_renderer *r = new acr_host_renderer();
		r->renderer = new Renderer(args);
		r->renderer->setBalance(rebalance_every);
		return r;
	} catch (const std::exception &e) {
		setErr(err, err_len, e.what());
		return nullptr;
	}
}

void acr_host_renderer_destroy(acr_host_renderer *r)
{
	if (!r) return;
	delete r->renderer;
	delete r;
}

const char *acr_host_renderer_error(const acr_host_renderer *r) { return r ? r->error.c_str() : "null renderer"; }

int acr_host_renderer_load_scene(acr_host_renderer *r, const acr_host_scene *scene)
{
	if (!scene || !scene->scene) return ACR_ERR_INVALID;
	return guarded(r, [&] { r->renderer->loadScene(*scene->scene); });
}

int acr_host_renderer_move_camera(acr_host_renderer *r, const float pos[2], const float dir[2])
{
	if (!pos || !dir) return ACR_ERR_INVALID;
	return guarded(r, [&] { r->renderer->moveCamera(math::vec2(pos[0], pos[1]), math::vec2(dir[0], dir[1])); });
}

int acr_host_renderer_set_camera(acr_host_renderer *r, const acr_camera *cam)
{
	if (!cam) return ACR_ERR_INVALID;
	return guarded(r, [&] {
		Camera c;
		c.aspectRatio = cam->aspect_ratio;
		c.horizontalFOV = cam->horizontal_fov;
		c.position = math::vec3(cam->position[0], cam->position[1], cam->position[2]);
		c.forward = math::vec3(cam->forward[0], cam->forward[1], cam->forward[2]);
		c.up = math::vec3(cam->up[0], cam->up[1], cam->up[2]);
		r->renderer->setCamera(c);
	});
}

int acr_host_renderer_get_camera(const acr_host_renderer *r, acr_camera *cam)
{
	if (!r || !r->renderer || !cam) return ACR_ERR_INVALID;
	try {
		const Camera c = r->renderer->getCamera();
		cam->aspect_ratio = c.aspectRatio;
		cam->horizontal_fov = c.horizontalFOV;
		cam->position[0] = c.position.x; cam->position[1] = c.position.y; cam->position[2] = c.position.z;
		cam->forward[0] = c.forward.x; cam->forward[1] = c.forward.y; cam->forward[2] = c.forward.z;
		cam->up[0] = c.up.x; cam->up[1] = c.up.y; cam->up[2] = c.up.z;
		return ACR_OK;
	} catch (const std::exception &) {
		return ACR_ERR_NO_SCENE;
	}
}

int acr_host_move_camera(acr_camera *cam, const float pos[2], const float dir[2])
{
	if (!cam || !pos || !dir) return ACR_ERR_INVALID;
	Camera c;
	c.aspectRatio = cam->aspect_ratio;
	c.horizontalFOV = cam->horizontal_fov;
	c.position = math::vec3(cam->position[0], cam->position[1], cam->position[2]);
	c.forward = math::vec3(cam->forward[0], cam->forward[1], cam->forward[2]);
	c.up = math::vec3(cam->up[0], cam->up[1], cam->up[2]);
	c = movedCamera(c, math::vec2(pos[0], pos[1]), math::vec2(dir[0], dir[1]));
	cam->position[0] = c.position.x; cam->position[1] = c.position.y; cam->position[2] = c.position.z;
	cam->forward[0] = c.forward.x; cam->forward[1] = c.forward.y; cam->forward[2] = c.forward.z;
	cam->up[0] = c.up.x; cam->up[1] = c.up.y; cam->up[2] = c.up.z;
	return ACR_OK;
}

int acr_host_renderer_configure(acr_host_renderer *r, uint32_t max_bounces, int64_t fixed_seed, uint32_t collect_stats)
{
	return guarded(r, [&] {
		r->renderer->setMaxBounces(max_bounces);
		r->renderer->setFixedSeed(fixed_seed);
		r->renderer->setCollectStats(collect_stats != 0);
	});
}

int acr_host_renderer_set_adaptive(acr_host_renderer *r, uint32_t on, uint32_t every)
{
	return guarded(r, [&] { r->renderer->setAdaptiveAssignment(on != 0, every); });
}

int acr_host_renderer_render(acr_host_renderer *r)
{
	return guarded(r, [&] { r->renderer->render(); });
}

uint64_t acr_host_renderer_frames_no_move(const acr_host_renderer *r) { return r && r->renderer ? r->renderer->framesWithoutMove() : 0; }

int acr_host_renderer_read(acr_host_renderer *r, float *rgba, size_t num_floats, uint32_t *primid, float *hit_t)
{
	return guarded(r, [&] {
		acr_hip_ctx *ctx = r->renderer->context();
		const size_t n = size_t(r->renderer->width()) * r->renderer->height();
		if (!ctx) { // several GPUs: the frame is assembled on the first one by the read
			if (primid || hit_t) throw std::runtime_error("primitive ids and hit distances are not kept when rendering across several GPUs");
			if (num_floats != n * 4) throw std::runtime_error("size mismatch");
			std::vector<float> tmp;
			r->renderer->readFramebuffer(tmp);
			if (rgba) memcpy(rgba, tmp.data(), tmp.size() * sizeof(float));
			return;
		}
		if (rgba && acr_hip_read_framebuffer(ctx, rgba, num_floats) != ACR_OK) throw std::runtime_error(acr_hip_last_error(ctx));
		if (primid && acr_hip_read_primid(ctx, primid, n) != ACR_OK) throw std::runtime_error(acr_hip_last_error(ctx));
		if (hit_t && acr_hip_read_hit_t(ctx, hit_t, n) != ACR_OK) throw std::runtime_error(acr_hip_last_error(ctx));
	});
}

int acr_host_renderer_stats(acr_host_renderer *r, acr_stats *stats)
{
	if (!stats) return ACR_ERR_INVALID;
	return guarded(r, [&] { *stats = r->renderer->stats(); });
}

acr_hip_ctx *acr_host_renderer_ctx(acr_host_renderer *r) { return r && r->renderer ? r->renderer->context() : nullptr; }

int acr_host_renderer_save(acr_host_renderer *r, const char *path)
{
	if (!path) return ACR_ERR_INVALID;
	return guarded(r, [&] { r->renderer->save(path); });
}

// Contiguous runs along `order` whose summed costs are in the ratio of `weights` (equal when weights is null):
// part k ends at the first tile where the running cost reaches the sum of the first k+1 weights' share of the total.
int acr_host_partition_tiles_weighted(const uint32_t *order, const float *cost, uint32_t num_tiles, uint32_t parts, const float *weights,
                                      uint32_t *begin_out)
{
	if (!order || !begin_out || parts == 0) return ACR_ERR_INVALID;
	double total = 0, wsum = 0;
	for (uint32_t i = 0; i < num_tiles; i++) {
		if (order[i] >= num_tiles) return ACR_ERR_INVALID;
		total += cost ? double(cost[order[i]] > 0 ? cost[order[i]] : 0) : 1.0;
	}
	for (uint32_t k = 0; k < parts; k++) {
		if (weights && !(weights[k] >= 0)) return ACR_ERR_INVALID; // negative or NaN
		wsum += weights ? double(weights[k]) : 1.0;
	}
	if (!(wsum > 0)) return ACR_ERR_INVALID;
	begin_out[0] = 0;
	uint32_t at = 0;
	double run = 0, wrun = 0;
	for (uint32_t k = 1; k < parts; k++) {
		wrun += weights ? double(weights[k - 1]) : 1.0;
		const double target = total * wrun / wsum;
		auto costAt = [&](uint32_t i) { return cost ? double(cost[order[i]] > 0 ? cost[order[i]] : 0) : 1.0; };
		while (at < num_tiles) {
			const double c = costAt(at);
			// stop before the tile that would overshoot the target by more than it undershoots
			if (run + c * 0.5 > target) break;
			run += c;
			at++;
		}
		if (num_tiles >= parts) {
			// no part with a share is left without a tile (a rank with nothing to render would sit out the frame),
			// and enough tiles stay for the parts that follow
			const bool hasShare = !weights || weights[k - 1] > 0;
			while (hasShare && at < begin_out[k - 1] + 1) run += costAt(at++);
			uint32_t follow = 0; // parts after this cut that have a share
			for (uint32_t j = k; j < parts; j++) follow += (!weights || weights[j] > 0) ? 1u : 0u;
			while (at > num_tiles - follow) run -= costAt(--at);
		}
		begin_out[k] = at;
	}
	begin_out[parts] = num_tiles;
	return ACR_OK;
}

int acr_host_partition_tiles(const uint32_t *order, const float *cost, uint32_t num_tiles, uint32_t parts, uint32_t *begin_out)
{
	return acr_host_partition_tiles_weighted(order, cost, num_tiles, parts, nullptr, begin_out);
}

int acr_host_tile_order(uint32_t tiles_x, uint32_t tiles_y, uint32_t *order_out)
{
	if (!order_out || tiles_x == 0 || tiles_y == 0) return ACR_ERR_INVALID;
	auto spread = [](uint32_t v) {
		uint64_t x = v;
		x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
		x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
		x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
		x = (x | (x << 2)) & 0x3333333333333333ull;
		x = (x | (x << 1)) & 0x5555555555555555ull;
		return x;
	};
	std::vector<std::pair<uint64_t, uint32_t>> keyed;
	keyed.reserve(size_t(tiles_x) * tiles_y);
	for (uint32_t y = 0; y < tiles_y; y++)
		for (uint32_t x = 0; x < tiles_x; x++)
			keyed.emplace_back(spread(x) | (spread(y) << 1), y * tiles_x + x);
	std::sort(keyed.begin(), keyed.end());
	for (size_t i = 0; i < keyed.size(); i++) order_out[i] = keyed[i].second;
	return ACR_OK;
}

int acr_host_kat(const char *fn, const float *in, float *out)
{
	using namespace acr::math;
	if (!fn || !in || !out) return -1;
	const std::string f(fn);
	auto v3 = [&](int at) { return vec3(in[at], in[at + 1], in[at + 2]); };
	auto put3 = [&](const vec3 &v) { out[0] = v.x; out[1] = v.y; out[2] = v.z; return 3; };
	auto m4 = [&](int at) { mat4 m; for (int c = 0; c < 4; c++) m.c[c] = vec4(in[at + 4 * c], in[at + 4 * c + 1], in[at + 4 * c + 2], in[at + 4 * c + 3]); return m; };
	auto m3 = [&](int at) { mat3 m; for (int c = 0; c < 3; c++) m.c[c] = vec3(in[at + 3 * c], in[at + 3 * c + 1], in[at + 3 * c + 2]); return m; };
	auto putm4 = [&](const mat4 &m, int at) { for (int c = 0; c < 4; c++) { out[at + 4 * c] = m.c[c].x; out[at + 4 * c + 1] = m.c[c].y; out[at + 4 * c + 2] = m.c[c].z; out[at + 4 * c + 3] = m.c[c].w; } };
	auto putm3 = [&](const mat3 &m, int at) { for (int c = 0; c < 3; c++) { out[at + 3 * c] = m.c[c].x; out[at + 3 * c + 1] = m.c[c].y; out[at + 3 * c + 2] = m.c[c].z; } };
	if (f == "dot3") { out[0] = dot(v3(0), v3(3)); return 1; }
	if (f == "cross3") return put3(cross(v3(0), v3(3)));
	if (f == "length3") { out[0] = length(v3(0)); return 1; }
	if (f == "normalize3") return put3(normalize(v3(0)));
	if (f == "reflect3") return put3(reflect(v3(0), v3(3)));
	if (f == "refract3") return put3(refract(v3(0), v3(3), in[6]));
	if (f == "min3") return put3(min(v3(0), v3(3)));
	if (f == "max3") return put3(max(v3(0), v3(3)));
	if (f == "compmax3") { out[0] = compMax(v3(0)); return 1; }
	if (f == "div3s") return put3(v3(0) / in[3]);
	if (f == "rdiv3") return put3(1.0f / v3(0));
	if (f == "m4mulv") { const vec4 r = m4(0) * vec4(in[16], in[17], in[18], in[19]); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w; return 4; }
	if (f == "m4mul") { putm4(m4(0) * m4(16), 0); return 16; }
	if (f == "m4inverse") { putm4(inverse(m4(0)), 0); return 16; }
	if (f == "m3mulv") return put3(m3(0) * v3(9));
	if (f == "m3inverse") { putm3(inverse(m3(0)), 0); return 9; }
	if (f == "normal_chain") {
		const mat3 N = transpose(inverse(toMat3(m4(0))));
		putm3(N, 0);
		putm3(inverse(N), 9);
		return 18;
	}
	if (f == "translate") return put3(translate(m4(0), v3(16)));
	if (f == "translaten") return put3(translaten(m3(0), v3(9)));
	if (f == "rotate3") return put3(rotate(v3(0), in[3], v3(4)));
	if (f == "pi") { out[0] = pi(); out[1] = 2 * pi(); return 2; }
	return -1;
}

float acr_host_fastersin(float x) { return acr::math::fastersin(x); }
float acr_host_fastercos(float x) { return acr::math::fastercos(x); }
float acr_host_fastertanfull(float x) { return acr::math::fastertanfull(x); }
void acr_host_rng_uniforms(uint64_t seed, uint32_t n, float *out)
{
	acr::math::RandState s;
	acr::math::randInit(s, seed);
	for (uint32_t i = 0; i < n; i++) out[i] = acr::math::randUniform(s);
}

} // extern "C"
