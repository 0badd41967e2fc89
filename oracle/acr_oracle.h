/*
 * acr_oracle.h -- CPU oracle for the CudaRaytracer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing that ships (the HIP library, the C++
 * host mirror, the Python package) may include, link or call this.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED (except the vector/matrix primitives): the reference ships
 * no tests, golden vectors or fixtures for this path, and its own sources do
 * not build in this image without writing stand-ins for CUDA, cuRAND, Thrust
 * and assimp headers.  This file is therefore a line-by-line restatement of
 * the reference algorithm from its source text.  The one part that IS pinned
 * is the GLM 0.9.6.3 arithmetic (vendored header-only inside the reference):
 * oracle/ref_glm_kat.cpp compiles against those headers where they lie and
 * emits known-answer vectors that tests/ check this file against.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout's root).
 */
#ifndef ACR_ORACLE_H
#define ACR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- import-level scene (what assimp's aiScene hands scene.cu:54) --------
 * Same memory layout as the product's interchange struct so that a test can
 * feed one packed buffer to both sides; declared independently on purpose. */
typedef struct orc_import_node {
	int32_t parent;          /* -1 for the root; nodes are in DFS pre-order */
	float transform[16];     /* local transform, column-major [col][row] (glm layout after scene.cu:17-26) */
	uint32_t mesh_begin;     /* into node_mesh_ids */
	uint32_t mesh_count;
	char name[64];
} orc_import_node;

typedef struct orc_import_mesh {
	uint32_t vertex_begin, vertex_count; /* into positions/normals/colors (3 floats per vertex) */
	uint32_t face_begin, face_count;     /* into indices (3 per face, mesh-local vertex numbers) */
	uint32_t material_index;
	uint32_t reserved;
} orc_import_mesh;

typedef struct orc_import_material {
	float diffuse[3], ambient[3], specular[3];
	float refracti;
} orc_import_material;

typedef struct orc_import_light {
	char name[64];
	int32_t type;            /* 0 POINT, 1 DIRECTIONAL, 2 SPOT (scene.h:58-63) */
	float position[3], direction[3];
	float ambient[3], diffuse[3], specular[3];
	float att_constant, att_linear, att_quadratic;
	float inner_cone, outer_cone;
} orc_import_light;

typedef struct orc_import_camera {
	char name[64];
	float aspect, hfov;
	float position[3], up[3], look_at[3];
} orc_import_camera;

typedef struct orc_import_scene {
	uint32_t num_nodes;          const orc_import_node *nodes;
	uint32_t num_node_mesh_ids;  const uint32_t *node_mesh_ids;
	uint32_t num_meshes;         const orc_import_mesh *meshes;
	uint32_t num_vertices;       const float *positions; const float *normals; const float *colors; /* colors may be NULL -> 1.0 */
	uint32_t num_faces;          const uint32_t *indices;
	uint32_t num_materials;      const orc_import_material *materials;
	uint32_t num_lights;         const orc_import_light *lights;
	orc_import_camera camera;
} orc_import_scene;

/* ---- built scene ------------------------------------------------------- */
typedef struct orc_scene orc_scene;

typedef struct orc_camera {
	float aspect, hfov;
	float position[3], forward[3], up[3];
} orc_camera;

/* One BIH node as the reference stores it (bih.h:92-98), flags unpacked.
 * `defined` says which fields sift() actually wrote: bit0 start/end,
 * bit1 left/right/axis, bit2 isLeaf.  Undefined fields are zero here. */
typedef struct orc_bih_node {
	uint32_t start, end;
	float left, right;
	uint8_t axis, is_leaf, defined, pad;
} orc_bih_node;

typedef struct orc_counters {
	uint64_t rays_primary, rays_shadow, rays_bounce;
	uint64_t primary_hits;
	uint64_t node_visits;     /* internal-node visits, both BIH levels */
	uint64_t slab_tests;      /* BoundingBox::intersect calls (2 per internal node + 1 per mesh) */
	uint64_t object_visits;   /* Object::intersect calls */
	uint64_t mesh_visits;     /* Mesh::intersect calls */
	uint64_t triangle_tests;  /* Face::intersect calls */
	uint64_t object_hits;     /* Object::intersect calls whose local ray hit something (scene.cu:426-437 runs) */
	uint64_t mesh_box_hits;   /* Mesh::intersect calls that passed the mesh box test (geometry.cu:55) */
	uint64_t order_doubts;    /* near-side-first order only: mesh traversals redone in the reference's order */
} orc_counters;

typedef struct orc_frame {
	uint32_t width, height;
	uint64_t seed;            /* the `seed` argument of scatterTrace (renderer.cu:327) */
	uint64_t frames;          /* the `frames` argument: 0 overwrites, >0 running mean */
	int32_t max_bounces;      /* N of rayColor<N> (renderer.cu:10); reference default 5 */
	int32_t num_threads;      /* OpenMP threads; <=0 -> runtime default */
	uint32_t row_begin, row_end; /* render rows [row_begin,row_end); 0,0 -> all rows */
} orc_frame;

/* Per-pixel record of the FIRST Scene::intersect of rayColor<MAX> (the primary ray). */
typedef struct orc_aov {
	uint32_t *prim;      /* global primitive id, 0xFFFFFFFF on miss */
	float *t;            /* world-space t (scene.cu:431), FLT_MAX on miss */
	uint32_t *material;  /* materialIndex, 0xFFFFFFFF on miss */
	float *position;     /* 3 per pixel */
	float *normal;       /* 3 per pixel */
	uint64_t *path_key;  /* the pixel's Path::path after rayColor (renderer.cu:341,372; bih.h:13-27,268-274) */
} orc_aov;

orc_scene *orc_scene_create(const orc_import_scene *imp, uint32_t scene_bih_depth, uint32_t mesh_bih_depth);
void orc_scene_destroy(orc_scene *s);

void orc_scene_camera(const orc_scene *s, orc_camera *out);      /* camera after scene.cu:185-191 */
uint32_t orc_scene_num_objects(const orc_scene *s);              /* filtered objects (scene.cu:140-143) */
uint32_t orc_scene_num_meshes(const orc_scene *s);
uint32_t orc_scene_num_lights(const orc_scene *s);
uint32_t orc_scene_num_triangles(const orc_scene *s);
void orc_scene_bounds(const orc_scene *s, float mn[3], float mx[3]);
/* scene BIH: nodes[(1<<depth)-1]; object_order[i] = Object::index of the i-th object after sift */
void orc_scene_bih(const orc_scene *s, orc_bih_node *nodes, int32_t *object_order);
/* per object (BIH order): 16+16+9+9 floats: globalTransform, globalInverseTransform,
 * globalNormalTransform, globalInverseNormalTransform; then centroid(3), bbox min(3), max(3) */
void orc_scene_object(const orc_scene *s, uint32_t i, float out[59]);
/* per mesh: nodes[(1<<depth)-1]; face_order[i] = original face number of the i-th face after sift */
void orc_mesh_bih(const orc_scene *s, uint32_t mesh, orc_bih_node *nodes, uint32_t *face_order);
void orc_mesh_info(const orc_scene *s, uint32_t mesh, float centroid[3], float mn[3], float mx[3], uint32_t *num_faces, uint32_t *material);
/* light after scene.cu:174-183: type, position(3), direction(3), diffuse(3), att(3), cones(2) */
void orc_scene_light(const orc_scene *s, uint32_t i, int32_t *type, float out[14]);

/* scatterTrace over the whole image (renderer.cu:326-388).  screen is W*H*4 floats,
 * read when frames>0.  pixel_map may be NULL (identity).  Returns 0 on success. */
int orc_render(const orc_scene *s, const orc_camera *cam, const orc_frame *f,
               float *screen, const int32_t *pixel_map, const orc_aov *aov, orc_counters *counters);

/* NOT in the reference.  near_first != 0: mesh trees are walked near side first, as the GPU path walks deep
 * trees; results must not change (that is what the tests check), the node / slab / triangle counters become those
 * of that order.  Default 0: the reference's order (bih.h:253-264). */
void orc_scene_set_mesh_order(orc_scene *s, int near_first);

/* Closest-hit query for arbitrary rays (Scene::intersect, scene.cu:204). tmax preset per ray. */
void orc_trace_rays(const orc_scene *s, uint32_t n, const float *origins, const float *dirs,
                    const float *tmax, uint8_t *hit, float *t, uint32_t *prim, float *position, float *normal);

/* ---- unit entry points for known-answer tests --------------------------- */
float orc_fastersin(float x);
float orc_fastercos(float x);
float orc_fastertanfull(float x);
int orc_ray_triangle(const float ro[3], const float rd[3], const float a[3], const float b[3], const float c[3],
                     float bary[3], float *t);
int orc_slab(const float mn[3], const float mx[3], const float ro[3], const float rd[3], float tcur);
void orc_rng_uniforms(uint64_t seed, uint32_t n, float *out);       /* curand_init(seed,0,0) + n curand_uniform */
void orc_translate(const float m[16], const float v[3], float out[3]);
void orc_translaten(const float m[9], const float v[3], float out[3]);
void orc_mat4_inverse(const float m[16], float out[16]);
void orc_mat3_inverse(const float m[9], float out[9]);
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);
void orc_normalize(const float v[3], float out[3]);
void orc_reflect(const float i[3], const float n[3], float out[3]);
void orc_refract(const float i[3], const float n[3], float eta, float out[3]);
void orc_rotate(const float v[3], float angle, const float axis[3], float out[3]);
/* Renderer::moveCamera (renderer.cu:145-166) */
void orc_move_camera(orc_camera *cam, const float pos[2], const float dir[2]);
/* PathAggregate (bih.h:29-66) for every pixel: 64 signed vote counters per pixel, votes[bit * n + pixel].
 * reset != 0: PathAggregate::set (the frames == 0 branch, renderer.cu:386), counter = path bit ? -1 : +1;
 * otherwise PathAggregate::add (renderer.cu:380), counter += path bit ? -1 : +1. */
void orc_path_votes(const uint64_t *path_keys, int32_t *votes, uint32_t n, int reset);
/* PathAggregate::operator uint64_t (bih.h:55-65): bit i of the key is set where counter i is negative */
void orc_vote_keys(const int32_t *votes, uint64_t *keys, uint32_t n);
/* known-answer dispatcher: evaluates one record of tests/golden/glm_kat.json */
int orc_kat(const char *fn, const float *in, float *out);

#ifdef __cplusplus
}
#endif
#endif
