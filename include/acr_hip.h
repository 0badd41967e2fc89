/*
 * acr_hip.h -- C ABI of the MI355X renderer core (libacr_hip.so).
 *
 * This is the drop-in boundary for the reference's one host->device crossing:
 * the `scatterTrace<<<grid,block>>>(screen, seed, frames)` launch and the
 * `__constant__ DevParams` / `acr::vector<T>` uploads that feed it
 * (reference src/renderer/renderer.cu:14-23,168-182,326-327,417-422;
 * src/utils/vector.h:46-55).  Everything below the boundary -- ray
 * generation, BIH traversal, triangle tests, shading -- runs in hand-written
 * HIP kernels for gfx950; everything above it (scene loading, BIH build,
 * camera handling) stays host C++.
 *
 * Plain C: opaque context, plain pointers and sizes, int status codes
 * (0 = ok).  No call exits the process; acr_hip_last_error() gives the text
 * of the last failure on a context.  A context is bound to one GPU and is
 * not thread-safe: the caller serialises calls on it, like the reference's
 * single GLUT thread (renderer.cu:25-29).
 *
 * All "file:line" citations are relative to the reference checkout's root.
 */
#ifndef ACR_HIP_H
#define ACR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACR_OK                 0
#define ACR_ERR_INVALID        1   /* bad argument / inconsistent scene description */
#define ACR_ERR_HIP            2   /* a HIP runtime call failed; see acr_hip_last_error */
#define ACR_ERR_NO_SCENE       3   /* render before acr_hip_upload_scene / acr_hip_set_camera */
#define ACR_ERR_UNSUPPORTED    4   /* e.g. BIH depth or bounce count beyond the compiled limits */

#define ACR_MAX_BIH_DEPTH      30  /* node and item counts must stay below 2^28 */
#define ACR_MAX_BOUNCES        16

#define ACR_LIGHT_POINT        0   /* scene.h:58-63 */
#define ACR_LIGHT_DIRECTIONAL  1
#define ACR_LIGHT_SPOT         2

#define ACR_NODE_AXIS_MASK     0x3u  /* node flag byte: bits 0-1 axis, bit 2 isLeaf (bih.h:96-97) */
#define ACR_NODE_LEAF          0x4u

#define ACR_PRIM_NONE          0xFFFFFFFFu

#define ACR_TRAVERSAL_AUTO      0  /* interval form of the slab test where it is provably identical, six-plane form elsewhere */
#define ACR_TRAVERSAL_SIX_PLANE 1  /* the slab test exactly as written (geometry.h:44-76) for every ray */
#define ACR_TRAVERSAL_LIST_PASS 2  /* every pixel through the pass that normally renders only the listed ones: the interval form in the
                                     reference's visiting order, the six-plane form for the rays it cannot serve; slow, for tests */
#define ACR_TRAVERSAL_NESTED    3  /* as AUTO, but every frame goes through the kernel in which a lane traces its pixel's rays on its own: AUTO hands
                                     the object tests of frames with bounces on LDS-resident scenes out across the wavefront, and lets the lanes
                                     of a wavefront share their walks through mesh trees resident in HBM */

typedef struct acr_hip_ctx acr_hip_ctx;

/*
 * Device-ready scene, structure-of-arrays.  Replaces the byte image of
 * `acr::Scene` in DevParams (renderer.cu:16,171) and the device arrays the
 * reference's acr::vector constructors upload (scene.cu:82-83,98,151;
 * geometry.cu:37,47; bih.h:79).  All pointers are HOST pointers; the call
 * copies what it needs.  Matrices are column-major, m[col*R + row], the glm
 * layout the reference stores.
 *
 * BIH node arrays are the reference's implicit heap (children of i are 2i+1,
 * 2i+2; bih.h:105-115), (1 << depth) - 1 entries, fields as in bih.h:92-98.
 * Fields the reference's sift() leaves indeterminate (bih.h:139-143 and
 * nodes it never reaches) may hold anything; the kernels never read them.
 */
typedef struct acr_scene_soa {
	/* BIH over objects: Scene::objects, BIH<Object> (scene.h:117) */
	uint32_t scene_depth;
	float scene_min[3], scene_max[3];          /* Scene::boundingBox (scene.cu:145-149) */
	const uint32_t *snode_start, *snode_end;
	const float *snode_left, *snode_right;
	const uint8_t *snode_flags;

	/* objects in BIH order (after sift's permutation) */
	uint32_t num_objects;
	const float *obj_transform;                /* 16 each: globalTransform               (scene.cu:364) */
	const float *obj_inv_transform;            /* 16 each: globalInverseTransform        (scene.cu:365) */
	const float *obj_normal_transform;         /*  9 each: globalNormalTransform         (scene.cu:366) */
	const float *obj_inv_normal_transform;     /*  9 each: globalInverseNormalTransform  (scene.cu:367) */
	const float *obj_world_min, *obj_world_max; /* 3 each: Object::boundingBox, world space (scene.cu:392); may be NULL.
	                                             * Only used to skip objects a ray misses by a wide margin; results do not depend on it. */
	const uint32_t *obj_mesh_begin;            /* into obj_mesh_ids */
	const uint32_t *obj_mesh_count;            /* Object::meshes.size() (scene.cu:404) */
	uint32_t num_obj_mesh_ids;
	const uint32_t *obj_mesh_ids;

	/* meshes: acr::Mesh (mesh.h:11-33) */
	uint32_t num_meshes;
	const float *mesh_min, *mesh_max;          /* 3 each: Mesh::boundingBox (geometry.cu:31-32), also the BIH root box (geometry.cu:47) */
	const uint32_t *mesh_material;             /* Mesh::materialIndex */
	const uint32_t *mesh_depth;                /* MAX_DEPTH of this mesh's BIH<Face> */
	const uint32_t *mesh_node_begin;           /* first node of this mesh in the mnode_* pools */
	const uint32_t *mesh_tri_begin;            /* first triangle of this mesh in tri_*; node start/end are mesh-local */
	uint32_t num_mesh_nodes;
	const uint32_t *mnode_start, *mnode_end;
	const float *mnode_left, *mnode_right;
	const uint8_t *mnode_flags;

	/* triangles in per-mesh BIH order, vertex data pre-gathered through
	 * Face::indices (geometry.h:126-133): 3 floats per triangle per array */
	uint32_t num_triangles;
	const float *tri_p0, *tri_p1, *tri_p2;     /* Vertex::position of indices[0..2] */
	const float *tri_n0, *tri_n1, *tri_n2;     /* Vertex::normal   of indices[0..2] */
	const uint32_t *tri_prim_id;               /* caller-defined primitive id reported per pixel */

	/* materials: acr::Material (materials.h:12-18) */
	uint32_t num_materials;
	const float *mat_ambient, *mat_diffuse, *mat_specular;   /* 3 each */
	const float *mat_refractive_index;

	/* lights: acr::Light (scene.h:52-80) after the node transform (scene.cu:174-183) */
	uint32_t num_lights;
	const int32_t *light_type;
	const float *light_diffuse, *light_position, *light_direction;   /* 3 each */
	const float *light_attenuation;            /* 3 each: constant, linear, quadratic */
	const float *light_cone;                   /* 2 each: innerConeAngle, outerConeAngle */
} acr_scene_soa;

/* acr::Camera (camera.h:9-17).  aspect_ratio is overridden by width/height at
 * render time, as Renderer::loadScene does (renderer.cu:179). */
typedef struct acr_camera {
	float aspect_ratio;
	float horizontal_fov;
	float position[3], forward[3], up[3];
} acr_camera;

/* One launch of the path (renderer.cu:417-423). */
typedef struct acr_frame {
	uint32_t width, height;        /* DevParams::width/height (renderer.cu:172-173) */
	uint64_t seed;                 /* scatterTrace's `seed`: rand() in the reference (renderer.cu:421), explicit here */
	uint64_t frames;               /* scatterTrace's `frames` = framesNoMove: 0 overwrites, >0 running mean (renderer.cu:375-387) */
	uint32_t max_bounces;          /* MAX_BOUNCES (renderer.cu:10); reference value 5 */
	/* Optional tile list for multi-GPU partitioning: when num_tiles > 0 only
	 * those tiles are rendered and `screen` is the compact buffer
	 * [num_tiles][tile_h][tile_w] float4.  Tile t covers pixels
	 * x in [ (t % tiles_x) * tile_w, +tile_w ), y likewise with t / tiles_x,
	 * tiles_x = ceil(width / tile_w).  num_tiles == 0 renders the whole frame
	 * into [height][width] float4. */
	uint32_t tile_w, tile_h;
	uint32_t num_tiles;
	const uint32_t *tile_ids;      /* host pointer */
	/* Optional pixel permutation, DevParams::pixelValues (renderer.cu:19,343-355):
	 * device pointer to width*height ints or NULL for identity.  Reset to
	 * identity by the kernel when frames == 0, as in the reference. */
	int32_t *pixel_values_dev;
	uint32_t collect_stats;        /* ACR_STATS_*: count rays and time the kernel (acr_hip_read_stats) */
	/* Optional per-tile cost feedback for the next frame's tile assignment
	 * (the measured counterpart of the reference's per-pixel path statistics,
	 * renderer.cu:380,386,457-461): device uint32[num_tiles] each, zeroed by the
	 * call and filled by the kernel, entry i for tile_ids[i].  tile_rays_dev
	 * counts the rays traced for the tile (deterministic); tile_cycles_dev sums
	 * the shader-clock cycles its wavefronts ran, divided by 64 (timing, varies
	 * from run to run).  Either may be NULL.  Tile mode only. */
	uint32_t *tile_rays_dev;
	uint32_t *tile_cycles_dev;
	/* Optional per-pixel traversal-path key: device uint64[width*height] or NULL.  Entry i receives the Path of
	 * pixel i as the reference designed it (bih.h:13-27,268-274: two bits per internal node the primary ray visits
	 * in the tree over objects, most significant first: 00 neither child hit, 11 both, 10 the child with more
	 * items, 01 the other; renderer.cu:341,372).  Full-frame launches on scenes whose object tree was built with
	 * MAX_DEPTH >= 6.  Input of acr_hip_sort_pixel_map. */
	uint64_t *path_keys_dev;
	/* Optional, full-frame launches only: device array with a permutation of the frame's blocks of 16 x 16
	 * pixels (block b covers columns 16 * (b % ceil(width / 16)) ..., rows 16 * (b / ceil(width / 16)) ...):
	 * the k-th workgroup group of the launch renders block block_order_dev[k].  Results do not depend on it;
	 * it lets the caller choose which blocks start first (see acr_hip_set_block_order).  NULL: the context's order. */
	const uint32_t *block_order_dev;
	/* Tile list only: where a listed tile's pixels go.  0: `screen` is the compact buffer described above.
	 * Non-zero: `screen` (and primid_dev / hit_t_dev) are whole frames, [height][width], and every listed tile is
	 * written in place, pixel (x, y) at y * width + x; pixels of tiles that are not listed are left alone.  This is
	 * what lets several contexts (GPUs) each keep the running mean of their own tiles in a frame-shaped buffer
	 * (acr_hip_group_render), for any width and tile shape. */
	uint32_t tiles_into_frame;
} acr_frame;

/* acr_frame::collect_stats */
#define ACR_STATS_OFF        0
#define ACR_STATS_FRAME      1  /* counters of this launch alone */
#define ACR_STATS_ACCUMULATE 2  /* counters keep what earlier launches added: totals of a run of frames with one read at its end */
#define ACR_STATS_TIME       3  /* only the HIP-event time of the launch (acr_stats::kernel_ms): the kernel carries no counting code */

typedef struct acr_stats {
	uint64_t rays_primary, rays_shadow, rays_bounce;   /* rays actually traced by the last render */
	uint64_t primary_hits;
	float kernel_ms;               /* HIP-event time of the render kernel on the context's stream */
	uint32_t lds_bytes;            /* dynamic LDS per workgroup of the last launch */
	uint32_t scene_in_lds;         /* 2 whole scene staged in LDS, 1 only the small tables, 0 none */
	uint32_t grid_blocks;
	uint32_t interval_traversal;   /* 1 when the last launch used the interval form (ACR_TRAVERSAL_AUTO on a qualifying scene) */
	uint32_t six_plane_pixels;     /* pixels of that launch which were handed to the reference-order list pass (needs collect_stats) */
	uint32_t block_order;          /* order in which that launch started its blocks of 16 x 16 pixel slots: ACR_BLOCK_ORDER_* */
	uint32_t near_first;           /* 1 when that launch walked the mesh trees near side first (interval form, trees resident in HBM) */
	uint32_t wave_tasks;           /* 1 when that launch handed its object tests out across each wavefront (frames with bounces, scene resident in LDS) */
	uint32_t shared_walk;          /* 1 when the lanes of a wavefront shared their walks through the mesh tree (mesh trees resident in HBM, the tree over objects in LDS) */
} acr_stats;

/* ---- lifetime ---------------------------------------------------------- */

/* Binds a context to GPU `device_ordinal` and creates its stream.
 * Replaces the device selection in Renderer::Renderer (renderer.cu:73-104). */
int acr_hip_create(int device_ordinal, acr_hip_ctx **out);
void acr_hip_destroy(acr_hip_ctx *ctx);
const char *acr_hip_last_error(const acr_hip_ctx *ctx);

/* Use an existing hipStream_t (e.g. the caller framework's current stream)
 * for every later launch and copy.  NULL is HIP's default stream, as in any
 * HIP call.  acr_hip_use_own_stream goes back to the context's own stream. */
int acr_hip_set_stream(acr_hip_ctx *ctx, void *hip_stream);
int acr_hip_use_own_stream(acr_hip_ctx *ctx);
int acr_hip_sync(acr_hip_ctx *ctx);            /* cudaDeviceSynchronize at renderer.cu:422 */

/* Self-test with no counterpart in the reference.  The kernels replace the divisions 1.0f / x of the reference's
 * arithmetic (1 / ray direction, geometry.h:80; 1 / determinant, math.h:143; inversesqrt, GLM func_exponential.inl:149-153)
 * by the hardware's reciprocal approximation and one Newton step wherever 2^-126 <= |x| <= 2^126, which returns the
 * correctly rounded quotient there.  This call checks that claim on the GPU the context is bound to, for every one
 * of the 2^32 binary32 argument patterns, against the division itself: *mismatches is the number of patterns for which
 * the two differ (neither equal bits nor both NaN); anything but 0 means this GPU is not one the kernels were
 * verified for.  Takes a fraction of a second; synchronises the context's stream. */
int acr_hip_selftest_reciprocal(acr_hip_ctx *ctx, uint64_t *mismatches);

/* Launch order of the blocks of 16 x 16 pixel slots.  The GPU starts workgroups in grid order and a launch ends
 * with its last workgroups, so which blocks come last matters: if they are expensive ones, most of the GPU idles
 * while they finish.  With every > 0 (default 64) the context measures the time per block of a configuration
 * (frame size, tile list) on its second launch in a row, 4 launches later, then 8, 16 ... up to `every` launches later,
 * and starts the blocks of the launches in between most expensive first; acr_hip_set_camera with a different
 * camera keeps the table and puts the interval back to 4.  This is the reference's adaptive assignment (a table
 * consulted by the kernel, rebuilt from measured history on a cadence, reconsidered when the camera moves:
 * renderer.cu:343-355,457-461,165) applied to the order of workgroups; frames do not depend on it.
 * For full-frame launches (and lists of full-width bands of 16 rows) of a scene whose trees are read from device
 * memory the table is also XCD-aware: the
 * blocks one XCD's workgroups render are neighbours in the frame (runs of the blocks' Z-curve of equal measured
 * cost, most expensive first inside each), so that the workgroups sharing an L2 walk the same parts of the trees.
 * every == 0: natural order.  acr_frame::block_order_dev overrides it for one launch. */
#define ACR_BLOCK_ORDER_NATURAL  0
#define ACR_BLOCK_ORDER_MEASURED 1
#define ACR_BLOCK_ORDER_CALLER   2
int acr_hip_set_block_order(acr_hip_ctx *ctx, uint32_t every);

/* Chooses how BoundingBox::intersect (geometry.h:44-76) is evaluated during BIH traversal
 * (bih.h:241-248).  Results are identical either way; ACR_TRAVERSAL_SIX_PLANE and ACR_TRAVERSAL_LIST_PASS
 * exist so that tests can check exactly that.  Default ACR_TRAVERSAL_AUTO. */
int acr_hip_set_traversal(acr_hip_ctx *ctx, int mode);

/* ---- scene, camera, buffers -------------------------------------------- */

/* Validates and uploads the scene (replaces the acr::vector uploads and the
 * Scene part of cudaMemcpyToSymbol, renderer.cu:171,181). */
int acr_hip_upload_scene(acr_hip_ctx *ctx, const acr_scene_soa *scene);

/* Camera part of DevParams: written by Renderer::loadScene / moveCamera
 * (renderer.cu:163,181), read back by moveCamera (renderer.cu:148). */
int acr_hip_set_camera(acr_hip_ctx *ctx, const acr_camera *camera);
int acr_hip_get_camera(const acr_hip_ctx *ctx, acr_camera *camera);

/* Allocates the context-owned framebuffer (W*H float4, the reference's PBO,
 * renderer.cu:111), primitive-id / hit-distance buffers and the pixel map
 * (renderer.cu:133,137). */
int acr_hip_resize(acr_hip_ctx *ctx, uint32_t width, uint32_t height);

/* ---- the hot path ------------------------------------------------------- */

/* Launches the render kernel for one frame (renderer.cu:417-421) on the
 * context's stream and returns without waiting.
 *   screen_dev : device float4 buffer (see acr_frame); NULL = context-owned
 *   primid_dev : optional device uint32 per pixel slot, primitive id of the
 *                primary hit or ACR_PRIM_NONE; NULL = context-owned if sized
 *   hit_t_dev  : optional device float per pixel slot, world-space t of the
 *                primary hit (scene.cu:431) or FLT_MAX; NULL = context-owned */
int acr_hip_render(acr_hip_ctx *ctx, const acr_frame *frame, void *screen_dev, uint32_t *primid_dev, float *hit_t_dev);

/* acr_hip_render for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference (its renderer allocates everything
 * once, in Renderer::Renderer).  Same arguments, same kernels and bit-identical frames; the call itself
 * allocates and frees nothing.  The list of pixel slots the launch hands to its list pass, which acr_hip_render
 * keeps in a buffer sized per launch, lives in a fixed array in the library's device image
 * (4 * ACR_STATIC_MAX_SLOTS + 256 bytes per GPU, held until the process ends).  Limits, each refused with
 * ACR_ERR_UNSUPPORTED and a message: whole frames only (no tile list, no per-tile costs); at most
 * ACR_STATIC_MAX_SLOTS pixel slots, i.e. ceil(width / 16) * ceil(height / 16) * 256 (1920 x 1080 fits); no scene
 * whose mesh trees are too deep for the LDS part of the waiting-child column (about 14 levels); the context's
 * measured block order is not used (natural order, or acr_frame::block_order_dev).  The framebuffer is the
 * caller's, or the context's after acr_hip_resize.  All contexts on one GPU share the array: calls that
 * overlap in time on one GPU (two contexts, or two streams) race on it and must be serialised by the caller. */
#define ACR_STATIC_MAX_SLOTS (1u << 21)
int acr_hip_render_static(acr_hip_ctx *ctx, const acr_frame *frame, void *screen_dev, uint32_t *primid_dev, float *hit_t_dev);

/* Closest-hit queries for caller-supplied rays: Scene::intersect (scene.cu:204-207)
 * with info.t preset to tmax[i].  Host arrays in and out; synchronous. */
int acr_hip_trace_rays(acr_hip_ctx *ctx, uint32_t n, const float *origins, const float *dirs, const float *tmax,
                       uint8_t *hit, float *t, uint32_t *prim, float *position, float *normal);

/* ---- results (synchronous copies to host memory) ------------------------ */
int acr_hip_read_framebuffer(acr_hip_ctx *ctx, float *rgba, size_t num_floats);
int acr_hip_read_primid(acr_hip_ctx *ctx, uint32_t *prim, size_t count);
int acr_hip_read_hit_t(acr_hip_ctx *ctx, float *t, size_t count);
int acr_hip_read_stats(acr_hip_ctx *ctx, acr_stats *stats);
/* device address of the context-owned framebuffer (NULL before acr_hip_resize) */
void *acr_hip_framebuffer(acr_hip_ctx *ctx);
/* device addresses of the context-owned pixel table (width*height ints, the identity after acr_hip_resize;
 * DevParams::pixelValues, renderer.cu:19,133,137) and traversal-path keys (width*height uint64); NULL before
 * acr_hip_resize.  Pass them as acr_frame::pixel_values_dev / path_keys_dev and to acr_hip_sort_pixel_map. */
int32_t *acr_hip_pixel_map(acr_hip_ctx *ctx);
uint64_t *acr_hip_path_keys(acr_hip_ctx *ctx);

/* The reference's adaptive assignment step (renderer.cu:131-137,457-461; disabled at its v0): fills
 * pixel_values_dev[0..num_pixels) with the pixel numbers 0..num_pixels-1 stably sorted by their traversal-path
 * key (thrust::sequence + thrust::stable_sort_by_key), so that neighbouring work-items of later frames
 * (frames > 0, acr_frame::pixel_values_dev) render pixels whose primary rays traverse alike.  Keys are indexed
 * by pixel, so a pixel's history follows it through the sort.  Device pointers; asynchronous on the
 * context's stream. */
int acr_hip_sort_pixel_map(acr_hip_ctx *ctx, const uint64_t *path_keys_dev, int32_t *pixel_values_dev, uint32_t num_pixels);

/* The reference's per-pixel vote table, PathAggregate (bih.h:29-66; 64 signed counters per aggregate, one per bit
 * of a path key; all bodies commented out at its v0): the history of traversal paths that the adaptive assignment
 * sorts by.  acr_hip_vote_paths with reset != 0 is PathAggregate::set for every pixel (the frames == 0 branch,
 * renderer.cu:386: counter i = bit i of the pixel's key ? -1 : +1), with reset == 0 PathAggregate::add
 * (renderer.cu:380: counter i += ...).  acr_hip_vote_keys is the conversion back to a key (bih.h:55-65: bit i set where
 * counter i is negative, i.e. where most frames had it set), the input of acr_hip_sort_pixel_map where the reference
 * copies pixelKeyData (renderer.cu:459).  Counters are device int32 [64][num_pixels] (votes[bit * num_pixels + pixel]),
 * indexed by pixel like the keys.  acr_hip_path_votes returns the context-owned table for the size of the last
 * acr_hip_resize, allocated (256 bytes per pixel) and zeroed on its first call; NULL before acr_hip_resize or when
 * there is no memory for it.  Device pointers; asynchronous on the context's stream. */
int32_t *acr_hip_path_votes(acr_hip_ctx *ctx);
int acr_hip_vote_paths(acr_hip_ctx *ctx, const uint64_t *path_keys_dev, int32_t *votes_dev, uint32_t num_pixels, int reset);
int acr_hip_vote_keys(acr_hip_ctx *ctx, const int32_t *votes_dev, uint64_t *path_keys_dev, uint32_t num_pixels);

/* Copies compact tiles (as written by a tiled acr_hip_render) into a full
 * [height][width] float4 frame on the same device: the last step after the
 * finished tiles of every GPU have been gathered to GPU 0. */
int acr_hip_scatter_tiles(acr_hip_ctx *ctx, const void *tiles_dev, const uint32_t *tile_ids, uint32_t num_tiles,
                          uint32_t tile_w, uint32_t tile_h, uint32_t width, uint32_t height, void *frame_dev);

/* ---- several GPUs behind one host thread ---------------------------------
 *
 * The reference selects ONE device (`maxNumDevices = 1`, renderer.cu:73-104) and its render() launches one kernel
 * on it (renderer.cu:397-423).  A group is that renderer over N devices: N contexts driven by the calling thread,
 * the scene replicated on each, the frame cut into full-width bands of 16 rows, member i rendering a contiguous run
 * of bands into a frame-shaped buffer on its own GPU, the pieces gathered on the FIRST member's GPU by peer-to-peer
 * copies (each sender over its own xGMI link) when the frame is asked for.  Which bands a member renders follows
 * the reference's adaptive-assignment rules (renderer.cu:343-355,457-461,165): an explicit table consulted by the
 * kernel (its tile list), rebuilt every `rebalance_every` frames from costs measured while rendering (wave clock
 * ticks per band), reset to the uniform split when the camera moves.  A device may be named more than once (several
 * members on one GPU: useful on a one-GPU machine, to rehearse).  Status codes and error text as for a context;
 * calls on a group are not thread-safe. */
typedef struct acr_hip_group acr_hip_group;

/* devices[0..n): GPU ordinals, member 0's GPU is where frames are assembled.  Replaces the device selection of
 * Renderer::Renderer (renderer.cu:73-104) for n devices. */
int acr_hip_group_create(uint32_t n, const int *devices, acr_hip_group **out);
void acr_hip_group_destroy(acr_hip_group *grp);
const char *acr_hip_group_last_error(const acr_hip_group *grp);
uint32_t acr_hip_group_size(const acr_hip_group *grp);
/* member i's context, for per-context switches (acr_hip_set_traversal, acr_hip_set_block_order); the group owns
 * it, its stream and its buffers: do not render, resize or destroy through it */
acr_hip_ctx *acr_hip_group_context(acr_hip_group *grp, uint32_t i);

/* acr_hip_upload_scene / acr_hip_set_camera / acr_hip_get_camera / acr_hip_resize for every member (renderer.cu:
 * 168-182,145-166,107-137).  A camera that differs from the last one also resets the band costs to uniform. */
int acr_hip_group_upload_scene(acr_hip_group *grp, const acr_scene_soa *scene);
int acr_hip_group_set_camera(acr_hip_group *grp, const acr_camera *camera);
int acr_hip_group_get_camera(const acr_hip_group *grp, acr_camera *camera);
int acr_hip_group_resize(acr_hip_group *grp, uint32_t width, uint32_t height);

/* rebalance_every: the split is rebuilt from costs measured every this many frames (default 16; 0: never; the costs
 * of frame k decide the split from frame k + 2 on, so that no frame waits for a measurement).  weights: share of the
 * frame's cost per member, n numbers, or NULL for equal shares (default): e.g. a smaller share for member 0 when
 * every frame is gathered, since its GPU also takes in the other members' pieces. */
int acr_hip_group_set_balance(acr_hip_group *grp, uint32_t rebalance_every, const float *weights);
/* begin[0..n]: member i renders bands [begin[i], begin[i + 1]) of the next frame; *num_bands = ceil(height / 16) */
int acr_hip_group_split(const acr_hip_group *grp, uint32_t *begin, uint32_t *num_bands);

/* One frame (renderer.cu:417-423) across the members; returns without waiting.  `frame` as for acr_hip_render with
 * width / height of acr_hip_group_resize, seed, frames, max_bounces and collect_stats; the tile list, pixel table,
 * per-tile costs, path keys and block order must be unset.
 *   frames == 0: the frame overwrites (what the reference renders after a camera move).  Frames alternate between
 *                two buffers per member, so the pieces of frame k may still travel while frame k + 1 renders.
 *   frames  > 0: frame k is averaged into the running mean (renderer.cu:375-387).  Every member keeps the running
 *                mean of the bands it owns; nothing is exchanged; when a rebalance moves a band to another member
 *                its running mean travels from the old owner to the new one before the next frame.
 *   gather != 0: also queue the copies that assemble this frame on member 0's GPU (otherwise that happens when
 *                acr_hip_group_frame is called). */
int acr_hip_group_render(acr_hip_group *grp, const acr_frame *frame, int gather);
/* Assembles the last frame on member 0's GPU if that has not been queued yet, waits for it, and returns its device
 * address ([height][width] float4).  The buffer is member 0's own: the next frame with frames > 0 updates member 0's
 * bands in it, the next but one with frames == 0 overwrites it. */
int acr_hip_group_frame(acr_hip_group *grp, void **frame_dev);
int acr_hip_group_read_framebuffer(acr_hip_group *grp, float *rgba, size_t num_floats);
/* ray counters of the members' last launches added up (collect_stats), kernel_ms the longest of them */
int acr_hip_group_read_stats(acr_hip_group *grp, acr_stats *stats);
int acr_hip_group_sync(acr_hip_group *grp);

#ifdef __cplusplus
}
#endif
#endif /* ACR_HIP_H */
