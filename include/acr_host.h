/*
 * acr_host.h -- C view of the C++ host side (libacr_host.so).
 *
 * The host side keeps the reference's C++ API (acr::Scene, acr::Camera,
 * acr::Material, acr::Light, acr::Mesh, acr::Object, acr::BIH<T,MAX_DEPTH>,
 * acr::Renderer; reference src/scene/scene.h, src/geometry/mesh.h,
 * src/utils/bih.h, src/renderer/renderer.h) and reaches the GPU only through
 * include/acr_hip.h.  This header lets non-C++ callers (the Python tests and
 * bench) drive those classes, and defines the import-level scene that stands
 * where assimp's aiScene stands in the reference (src/scene/scene.cu:28-46).
 *
 * "file:line" citations are relative to the reference checkout's root.
 */
#ifndef ACR_HOST_H
#define ACR_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "acr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- import-level scene: the fields of aiScene the reference reads ------ */

typedef struct acr_import_node {      /* aiNode: scene.cu:155-202,359-369 */
	int32_t parent;                   /* -1 for the root; nodes are stored in depth-first pre-order */
	float transform[16];              /* mTransformation as a glm mat4: column-major [col][row] (scene.cu:17-26) */
	uint32_t mesh_begin, mesh_count;  /* mMeshes: range in node_mesh_ids */
	char name[64];                    /* mName */
} acr_import_node;

typedef struct acr_import_mesh {      /* aiMesh: geometry.cu:8-48 */
	uint32_t vertex_begin, vertex_count;  /* range in positions/normals/colors */
	uint32_t face_begin, face_count;      /* range in indices (3 per face, mesh-local vertex numbers) */
	uint32_t material_index;
	uint32_t reserved;
} acr_import_mesh;

typedef struct acr_import_material {  /* aiMaterial: materials.cpp:5-17 */
	float diffuse[3], ambient[3], specular[3];
	float refracti;
} acr_import_material;

typedef struct acr_import_light {     /* aiLight: scene.cu:303-330 */
	char name[64];
	int32_t type;                     /* ACR_LIGHT_* */
	float position[3], direction[3];
	float ambient[3], diffuse[3], specular[3];
	float att_constant, att_linear, att_quadratic;
	float inner_cone, outer_cone;
} acr_import_light;

typedef struct acr_import_camera {    /* aiCamera: scene.cu:294-301 */
	char name[64];
	float aspect, hfov;
	float position[3], up[3], look_at[3];
} acr_import_camera;

typedef struct acr_import_scene {
	uint32_t num_nodes;          const acr_import_node *nodes;
	uint32_t num_node_mesh_ids;  const uint32_t *node_mesh_ids;
	uint32_t num_meshes;         const acr_import_mesh *meshes;
	uint32_t num_vertices;       const float *positions; const float *normals; const float *colors; /* colors may be NULL: 1.0 (geometry.cu:24) */
	uint32_t num_faces;          const uint32_t *indices;
	uint32_t num_materials;      const acr_import_material *materials;
	uint32_t num_lights;         const acr_import_light *lights;
	acr_import_camera camera;
} acr_import_scene;

/* Owning holder of an import-level scene produced by a loader. */
typedef struct acr_host_import acr_host_import;

/* Reads a Blender .blend file (uncompressed, little-endian, 64-bit) the way
 * the reference's assimp call does (scene.cu:30-35: triangulate, join
 * identical vertices).  err receives a message on failure. */
acr_host_import *acr_host_import_blend(const char *path, char *err, size_t err_len);
/* Procedural triangle soup: num_triangles small triangles with centres
 * uniform in [-extent,extent]^3 and edge vectors uniform in [-edge,edge]^3,
 * one mesh, one object, one material, one point light at z = 3 * extent,
 * camera at z = 4 * extent looking at the origin.  Deterministic in
 * (num_triangles, seed, edge, extent).  The reference's triangle test drops
 * triangles with |det| < 0.001 (math.h:140), i.e. roughly edge^2 < 0.001. */
acr_host_import *acr_host_import_soup(uint32_t num_triangles, uint32_t seed, float edge, float extent);
const acr_import_scene *acr_host_import_view(const acr_host_import *imp);
void acr_host_import_destroy(acr_host_import *imp);

/* ---- acr::Scene ---------------------------------------------------------- */

typedef struct acr_host_scene acr_host_scene;

/* Scene::Scene(args) + loadScene (scene.cu:28-86).  The depths are the
 * MAX_DEPTH template arguments of BIH<Object> and BIH<Face> (bih.h:68);
 * the reference uses 6 for both. */
acr_host_scene *acr_host_scene_create(const acr_import_scene *imp, uint32_t scene_bih_depth, uint32_t mesh_bih_depth, char *err, size_t err_len);
acr_host_scene *acr_host_scene_load(const char *blend_path, uint32_t scene_bih_depth, uint32_t mesh_bih_depth, char *err, size_t err_len);
void acr_host_scene_destroy(acr_host_scene *scene);

/* Device-ready arrays of the scene; the pointers stay valid until the scene is destroyed. */
int acr_host_scene_soa(const acr_host_scene *scene, acr_scene_soa *out);
int acr_host_scene_camera(const acr_host_scene *scene, acr_camera *out);
/* original index (Object::index / face number) of the i-th item after BIH construction */
int acr_host_scene_object_order(const acr_host_scene *scene, int32_t *order, uint32_t count);
int acr_host_scene_face_order(const acr_host_scene *scene, uint32_t mesh, uint32_t *order, uint32_t count);
/* the BIH printout of bih.h:301-314 for the scene BIH (mesh == UINT32_MAX) or one mesh */
int acr_host_scene_bih_text(const acr_host_scene *scene, uint32_t mesh, char *buf, size_t buf_len);
/* which node fields sift() defined, per node: bit0 start/end, bit1 left/right/axis, bit2 isLeaf */
int acr_host_scene_bih_defined(const acr_host_scene *scene, uint32_t mesh, uint8_t *defined, uint32_t count);

/* ---- acr::Renderer -------------------------------------------------------- */

typedef struct acr_host_renderer acr_host_renderer;

/* Renderer::Renderer(Args{title,pos,dim}) without the window: dim = (width, height, samples),
 * resolution = dim.xy * dim.z (renderer.cu:42-44).  Binds GPU `device`. */
acr_host_renderer *acr_host_renderer_create(uint32_t width, uint32_t height, uint32_t samples, int device, char *err, size_t err_len);
/* the same renderer over several GPUs (Renderer::Args::devices; an acr_hip_group underneath): every frame's bands are
 * dealt to the GPUs and re-dealt from measured costs every rebalance_every frames (0: never), the frame is assembled
 * on devices[0] when it is read.  The reference stops at one device (renderer.cu:74). */
acr_host_renderer *acr_host_renderer_create_multi(uint32_t width, uint32_t height, uint32_t samples, const int *devices, uint32_t num_devices,
                                                  uint32_t rebalance_every, char *err, size_t err_len);
void acr_host_renderer_destroy(acr_host_renderer *r);
const char *acr_host_renderer_error(const acr_host_renderer *r);
int acr_host_renderer_load_scene(acr_host_renderer *r, const acr_host_scene *scene);       /* renderer.cu:168-182 */
int acr_host_renderer_move_camera(acr_host_renderer *r, const float pos[2], const float dir[2]); /* renderer.cu:145-166 */
/* the camera arithmetic of moveCamera alone (renderer.cu:152-161), no GPU involved */
int acr_host_move_camera(acr_camera *cam, const float pos[2], const float dir[2]);
int acr_host_renderer_set_camera(acr_host_renderer *r, const acr_camera *cam);              /* direct placement; resets accumulation */
int acr_host_renderer_get_camera(const acr_host_renderer *r, acr_camera *cam);
/* MAX_BOUNCES (renderer.cu:10) and the rand() seed source (application.cpp:76, renderer.cu:421):
 * fixed_seed >= 0 makes every frame use seed = fixed_seed + frame number instead of rand(). */
int acr_host_renderer_configure(acr_host_renderer *r, uint32_t max_bounces, int64_t fixed_seed, uint32_t collect_stats);
/* adaptive assignment of pixels to work-items (renderer.cu:343-355,457-461): off by default, like the reference at v0 */
int acr_host_renderer_set_adaptive(acr_host_renderer *r, uint32_t on, uint32_t every);
int acr_host_renderer_render(acr_host_renderer *r);                                          /* renderer.cu:397-423, waits like :422 */
uint64_t acr_host_renderer_frames_no_move(const acr_host_renderer *r);
int acr_host_renderer_read(acr_host_renderer *r, float *rgba, size_t num_floats, uint32_t *primid, float *hit_t);
int acr_host_renderer_stats(acr_host_renderer *r, acr_stats *stats);
acr_hip_ctx *acr_host_renderer_ctx(acr_host_renderer *r);
/* writes the framebuffer as binary PPM (8-bit, clamped) or PFM (float) by extension */
int acr_host_renderer_save(acr_host_renderer *r, const char *path);

/* ---- tile assignment across GPUs ------------------------------------------ */

/* Splits tiles (visited in `order`, a space-filling-curve permutation of
 * 0..num_tiles-1) into `parts` contiguous runs whose summed cost is as equal
 * as possible.  cost == NULL means uniform cost: the reset state after a
 * camera move (renderer.cu:165,344-348).  begin_out gets parts+1 offsets into `order`. */
int acr_host_partition_tiles(const uint32_t *order, const float *cost, uint32_t num_tiles, uint32_t parts, uint32_t *begin_out);
/* The same with unequal shares: part k gets weights[k] / sum(weights) of the total cost (weights == NULL: equal
 * shares).  Used when the parts do not work at the same speed: a GPU that also has to send its tiles to GPU 0
 * finishes fewer of them per second than GPU 0 itself. */
int acr_host_partition_tiles_weighted(const uint32_t *order, const float *cost, uint32_t num_tiles, uint32_t parts, const float *weights,
                                      uint32_t *begin_out);
/* Morton (Z-curve) visiting order of a tiles_x * tiles_y grid. */
int acr_host_tile_order(uint32_t tiles_x, uint32_t tiles_y, uint32_t *order_out);

/* ---- unit access for tests -------------------------------------------------- */

/* evaluates one record of tests/golden/glm_kat.json with acr::math; returns the output count or -1 */
int acr_host_kat(const char *fn, const float *in, float *out);
float acr_host_fastersin(float x);
float acr_host_fastercos(float x);
float acr_host_fastertanfull(float x);
void acr_host_rng_uniforms(uint64_t seed, uint32_t n, float *out);

#ifdef __cplusplus
}
#endif
#endif /* ACR_HOST_H */
