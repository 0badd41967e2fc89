"""Python mirror of the host-side API: Scene, Renderer (the reference's
acr::Scene / acr::Renderer, src/scene/scene.h:82-127, src/renderer/renderer.h:18-51)
over libacr_host.so, and HipCore, a thin object over the C ABI of
include/acr_hip.h for callers that drive the GPU core directly."""
import ctypes as C

import numpy as np

from . import _capi
from .scene_io import ImportData


def _err_buf():
    return C.create_string_buffer(512)


class Scene:
    """acr::Scene.  Build from a .blend path (Scene::Args{filePath}), an
    ImportData, or a C import handle; BIH depths default to the reference's 6."""

    def __init__(self, file_path=None, import_data=None, import_view=None, scene_bih_depth=6, mesh_bih_depth=6):
        self._lib = _capi.host_lib()
        self._h = None
        err = _err_buf()
        if file_path is not None:
            self._h = self._lib.acr_host_scene_load(str(file_path).encode(), scene_bih_depth, mesh_bih_depth, err, len(err))
        else:
            keep = None
            if import_data is not None:
                import_view, keep = import_data.pack()
            if import_view is None:
                raise ValueError("Scene needs file_path, import_data or import_view")
            self._h = self._lib.acr_host_scene_create(C.byref(import_view), scene_bih_depth, mesh_bih_depth, err, len(err))
            del keep
        if not self._h:
            raise RuntimeError("scene construction failed: " + err.value.decode())
        self.soa = _capi.SceneSoA()
        if self._lib.acr_host_scene_soa(self._h, C.byref(self.soa)) != 0:
            raise RuntimeError("flatten failed")
        cam = _capi.Camera()
        self._lib.acr_host_scene_camera(self._h, C.byref(cam))
        self.camera = cam

    def close(self):
        if self._h:
            self._lib.acr_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # numpy views of the flattened arrays (valid while the scene lives)
    def _arr(self, ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, (n,)).view(dtype)

    def scene_nodes(self):
        n = (1 << self.soa.scene_depth) - 1
        s = self.soa
        return dict(start=self._arr(s.snode_start, n, np.uint32), end=self._arr(s.snode_end, n, np.uint32),
                    left=self._arr(s.snode_left, n, np.float32), right=self._arr(s.snode_right, n, np.float32),
                    flags=self._arr(s.snode_flags, n, np.uint8), defined=self.bih_defined(None))

    def mesh_nodes(self, mesh):
        s = self.soa
        depth = int(s.mesh_depth[mesh])
        n = (1 << depth) - 1
        b = int(s.mesh_node_begin[mesh])
        tot = s.num_mesh_nodes
        return dict(start=self._arr(s.mnode_start, tot, np.uint32)[b:b + n], end=self._arr(s.mnode_end, tot, np.uint32)[b:b + n],
                    left=self._arr(s.mnode_left, tot, np.float32)[b:b + n], right=self._arr(s.mnode_right, tot, np.float32)[b:b + n],
                    flags=self._arr(s.mnode_flags, tot, np.uint8)[b:b + n], defined=self.bih_defined(mesh))

    def bih_defined(self, mesh):
        if mesh is None:
            n, key = (1 << self.soa.scene_depth) - 1, 0xFFFFFFFF
        else:
            n, key = (1 << int(self.soa.mesh_depth[mesh])) - 1, mesh
        out = np.zeros(n, dtype=np.uint8)
        rc = self._lib.acr_host_scene_bih_defined(self._h, key, out.ctypes.data_as(_capi.c_u8_p), n)
        if rc != 0:
            raise RuntimeError("bih_defined failed")
        return out

    def object_order(self):
        out = np.zeros(self.soa.num_objects, dtype=np.int32)
        if self.soa.num_objects:
            self._lib.acr_host_scene_object_order(self._h, out.ctypes.data_as(_capi.c_i32_p), len(out))
        return out

    def face_order(self, mesh, count):
        out = np.zeros(count, dtype=np.uint32)
        if count:
            rc = self._lib.acr_host_scene_face_order(self._h, mesh, out.ctypes.data_as(_capi.c_u32_p), count)
            if rc != 0:
                raise RuntimeError("face_order failed")
        return out

    def bih_text(self, mesh=None):
        buf = C.create_string_buffer(1 << 16)
        self._lib.acr_host_scene_bih_text(self._h, 0xFFFFFFFF if mesh is None else mesh, buf, len(buf))
        return buf.value.decode()


class HipCore:
    """One context of the GPU core (include/acr_hip.h).  Raises on any error;
    there is no CPU path behind it."""

    def __init__(self, device=0):
        self._lib = _capi.hip_lib()
        h = C.c_void_p()
        rc = self._lib.acr_hip_create(device, C.byref(h))
        if rc != 0 or not h:
            raise RuntimeError(f"acr_hip_create(device={device}) failed with code {rc}: no usable GPU")
        self._h = h

    def close(self):
        if self._h:
            self._lib.acr_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._lib.acr_hip_last_error(self._h).decode()}")

    def upload_scene(self, soa):
        self._check(self._lib.acr_hip_upload_scene(self._h, C.byref(soa)), "acr_hip_upload_scene")

    def set_camera(self, cam):
        self._check(self._lib.acr_hip_set_camera(self._h, C.byref(cam)), "acr_hip_set_camera")

    def set_stream(self, stream_ptr):
        """hipStream_t as an integer; 0 / None is HIP's default stream"""
        self._check(self._lib.acr_hip_set_stream(self._h, C.c_void_p(stream_ptr or 0)), "acr_hip_set_stream")

    def use_own_stream(self):
        self._check(self._lib.acr_hip_use_own_stream(self._h), "acr_hip_use_own_stream")

    def set_traversal(self, six_plane_only=False, list_pass=False, nested=False):
        """Default: interval form of the slab test wherever it is provably identical.  six_plane_only: the
        six-plane form as written in the reference for every ray.  list_pass: every pixel through the pass that
        normally renders only the listed ones (reference visiting order; slow, for tests)."""
        mode = 3 if nested else (2 if list_pass else (1 if six_plane_only else 0))
        self._check(self._lib.acr_hip_set_traversal(self._h, mode), "acr_hip_set_traversal")

    def set_block_order(self, every):
        """every > 0 (default 64): blocks of 16 x 16 pixels start most expensive first, re-measured at intervals
        growing from 4 to `every` launches while the camera stays; 0: natural order.  Frames do not depend on it."""
        self._check(self._lib.acr_hip_set_block_order(self._h, every), "acr_hip_set_block_order")

    def resize(self, w, h):
        self._check(self._lib.acr_hip_resize(self._h, w, h), "acr_hip_resize")
        self._w, self._hh = w, h

    def render(self, width, height, seed=0, frames=0, max_bounces=1, collect_stats=False, screen_dev=None,
               primid_dev=None, hit_t_dev=None, tile_ids=None, tile_w=0, tile_h=0, pixel_values_dev=None,
               tile_rays_dev=None, tile_cycles_dev=None, path_keys_dev=None, block_order_dev=None, static=False,
               tiles_into_frame=False):
        """acr_hip_render; static=True calls acr_hip_render_static, the variant that allocates nothing."""
        f = _capi.Frame()
        f.width, f.height, f.seed, f.frames, f.max_bounces = width, height, seed, frames, max_bounces
        f.collect_stats = int(collect_stats)  # False/True, or 2: counters accumulate over launches (ACR_STATS_ACCUMULATE)
        keep = None
        if tile_ids is not None:
            keep = np.ascontiguousarray(tile_ids, dtype=np.uint32)
            f.tile_w, f.tile_h, f.num_tiles = tile_w, tile_h, len(keep)
            f.tile_ids = keep.ctypes.data_as(_capi.c_u32_p)
        f.pixel_values_dev = pixel_values_dev
        f.tile_rays_dev = tile_rays_dev
        f.tile_cycles_dev = tile_cycles_dev
        f.path_keys_dev = path_keys_dev
        f.block_order_dev = block_order_dev
        f.tiles_into_frame = 1 if tiles_into_frame else 0
        if static:
            self._check(self._lib.acr_hip_render_static(self._h, C.byref(f), screen_dev, primid_dev, hit_t_dev), "acr_hip_render_static")
        else:
            self._check(self._lib.acr_hip_render(self._h, C.byref(f), screen_dev, primid_dev, hit_t_dev), "acr_hip_render")

    def sync(self):
        self._check(self._lib.acr_hip_sync(self._h), "acr_hip_sync")

    def selftest_reciprocal(self):
        """acr_hip_selftest_reciprocal: argument patterns (of all 2^32) for which the kernels' reciprocal differs from 1.0f / x."""
        n = C.c_uint64(0)
        self._check(self._lib.acr_hip_selftest_reciprocal(self._h, C.byref(n)), "acr_hip_selftest_reciprocal")
        return int(n.value)

    def read_framebuffer(self):
        out = np.zeros((self._hh, self._w, 4), dtype=np.float32)
        self._check(self._lib.acr_hip_read_framebuffer(self._h, out.ctypes.data_as(_capi.c_float_p), out.size), "read_framebuffer")
        return out

    def read_primid(self):
        out = np.zeros((self._hh, self._w), dtype=np.uint32)
        self._check(self._lib.acr_hip_read_primid(self._h, out.ctypes.data_as(_capi.c_u32_p), out.size), "read_primid")
        return out

    def read_hit_t(self):
        out = np.zeros((self._hh, self._w), dtype=np.float32)
        self._check(self._lib.acr_hip_read_hit_t(self._h, out.ctypes.data_as(_capi.c_float_p), out.size), "read_hit_t")
        return out

    def read_stats(self):
        s = _capi.Stats()
        self._check(self._lib.acr_hip_read_stats(self._h, C.byref(s)), "read_stats")
        return s

    def framebuffer_ptr(self):
        return self._lib.acr_hip_framebuffer(self._h)

    def trace_rays(self, origins, dirs, tmax=None):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = len(o)
        tm = None if tmax is None else np.ascontiguousarray(tmax, dtype=np.float32)
        hit = np.zeros(n, np.uint8); t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32)
        pos = np.zeros((n, 3), np.float32); nrm = np.zeros((n, 3), np.float32)
        fp = _capi.c_float_p
        self._check(self._lib.acr_hip_trace_rays(self._h, n, o.ctypes.data_as(fp), d.ctypes.data_as(fp),
                                                 None if tm is None else tm.ctypes.data_as(fp),
                                                 hit.ctypes.data_as(_capi.c_u8_p), t.ctypes.data_as(fp),
                                                 prim.ctypes.data_as(_capi.c_u32_p), pos.ctypes.data_as(fp),
                                                 nrm.ctypes.data_as(fp)), "acr_hip_trace_rays")
        return dict(hit=hit, t=t, prim=prim, position=pos, normal=nrm)

    def sort_pixel_map(self, path_keys_dev, pixel_values_dev, num_pixels):
        """pixel_values <- pixel numbers stably sorted by traversal-path key (device pointers, asynchronous)"""
        self._check(self._lib.acr_hip_sort_pixel_map(self._h, path_keys_dev, pixel_values_dev, num_pixels), "acr_hip_sort_pixel_map")

    def vote_paths(self, path_keys_dev, votes_dev, num_pixels, reset):
        """the reference's PathAggregate::set (reset) / ::add for every pixel (device pointers, asynchronous)"""
        self._check(self._lib.acr_hip_vote_paths(self._h, path_keys_dev, votes_dev, num_pixels, 1 if reset else 0), "acr_hip_vote_paths")

    def vote_keys(self, votes_dev, path_keys_dev, num_pixels):
        """majority key of every pixel's vote table (device pointers, asynchronous)"""
        self._check(self._lib.acr_hip_vote_keys(self._h, votes_dev, path_keys_dev, num_pixels), "acr_hip_vote_keys")

    def scatter_tiles(self, tiles_dev, tile_ids, tile_w, tile_h, width, height, frame_dev):
        ids = np.ascontiguousarray(tile_ids, dtype=np.uint32)
        self._check(self._lib.acr_hip_scatter_tiles(self._h, tiles_dev, ids.ctypes.data_as(_capi.c_u32_p), len(ids),
                                                    tile_w, tile_h, width, height, frame_dev), "acr_hip_scatter_tiles")


class HipGroup:
    """acr_hip_group (include/acr_hip.h): one frame across several GPUs from one host thread.  devices: GPU
    ordinals, one per member (a GPU may be named more than once); frames are assembled on the first one."""

    def __init__(self, devices):
        self._lib = _capi.hip_lib()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self._lib.acr_hip_group_create(len(devices), devs, C.byref(h))
        if rc != 0 or not h:
            raise RuntimeError(f"acr_hip_group_create(devices={list(devices)}) failed with code {rc}: no usable GPU")
        self._h = h
        self.size = len(devices)
        self._w = self._hh = 0

    def close(self):
        if self._h:
            self._lib.acr_hip_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._lib.acr_hip_group_last_error(self._h).decode()}")

    def upload_scene(self, soa):
        self._check(self._lib.acr_hip_group_upload_scene(self._h, C.byref(soa)), "acr_hip_group_upload_scene")

    def set_camera(self, cam):
        self._check(self._lib.acr_hip_group_set_camera(self._h, C.byref(cam)), "acr_hip_group_set_camera")

    def resize(self, w, h):
        self._check(self._lib.acr_hip_group_resize(self._h, w, h), "acr_hip_group_resize")
        self._w, self._hh = w, h

    def set_balance(self, rebalance_every=16, weights=None):
        wp = None
        if weights is not None:
            weights = np.ascontiguousarray(weights, dtype=np.float32)
            if len(weights) != self.size:
                raise ValueError("one weight per member")
            wp = weights.ctypes.data_as(_capi.c_float_p)
        self._check(self._lib.acr_hip_group_set_balance(self._h, rebalance_every, wp), "acr_hip_group_set_balance")

    def split(self):
        """begin offsets: member i renders bands [begin[i], begin[i + 1]) of the next frame"""
        begin = np.zeros(self.size + 1, dtype=np.uint32)
        n = C.c_uint32()
        self._check(self._lib.acr_hip_group_split(self._h, begin.ctypes.data_as(_capi.c_u32_p), C.byref(n)), "acr_hip_group_split")
        return [int(x) for x in begin], int(n.value)

    def member_set_traversal(self, i, mode):
        ctx = self._lib.acr_hip_group_context(self._h, i)
        if self._lib.acr_hip_set_traversal(ctx, mode) != 0:
            raise RuntimeError("acr_hip_set_traversal failed")

    def render(self, seed=0, frames=0, max_bounces=1, collect_stats=0, gather=False):
        f = _capi.Frame()
        f.width, f.height, f.seed, f.frames, f.max_bounces = self._w, self._hh, seed, frames, max_bounces
        f.collect_stats = int(collect_stats)
        self._check(self._lib.acr_hip_group_render(self._h, C.byref(f), 1 if gather else 0), "acr_hip_group_render")

    def frame_ptr(self):
        p = C.c_void_p()
        self._check(self._lib.acr_hip_group_frame(self._h, C.byref(p)), "acr_hip_group_frame")
        return p.value

    def read_framebuffer(self):
        out = np.zeros((self._hh, self._w, 4), dtype=np.float32)
        self._check(self._lib.acr_hip_group_read_framebuffer(self._h, out.ctypes.data_as(_capi.c_float_p), out.size), "acr_hip_group_read_framebuffer")
        return out

    def read_stats(self):
        s = _capi.Stats()
        self._check(self._lib.acr_hip_group_read_stats(self._h, C.byref(s)), "acr_hip_group_read_stats")
        return s

    def sync(self):
        self._check(self._lib.acr_hip_group_sync(self._h), "acr_hip_group_sync")


class Renderer:
    """acr::Renderer without the window: Renderer(Args{dim}), loadScene, render, moveCamera."""

    def __init__(self, width, height, samples=1, device=0, devices=None, rebalance_every=16):
        """devices: a list of GPU ordinals renders every frame across them (Renderer::Args::devices)"""
        self._lib = _capi.host_lib()
        err = _err_buf()
        if devices is not None and len(devices) > 1:
            devs = (C.c_int * len(devices))(*devices)
            self._h = self._lib.acr_host_renderer_create_multi(width, height, samples, devs, len(devices), rebalance_every, err, len(err))
        else:
            self._h = self._lib.acr_host_renderer_create(width, height, samples, devices[0] if devices else device, err, len(err))
        if not self._h:
            raise RuntimeError("Renderer: " + err.value.decode())
        self.width, self.height = width * samples, height * samples

    def close(self):
        if self._h:
            self._lib.acr_host_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._lib.acr_host_renderer_error(self._h).decode()}")

    def configure(self, max_bounces=5, fixed_seed=-1, collect_stats=False):
        self._check(self._lib.acr_host_renderer_configure(self._h, max_bounces, fixed_seed, 1 if collect_stats else 0), "configure")

    def set_adaptive(self, on, every=60):
        """The reference's adaptive assignment (off by default, as at the reference's v0): re-sort the pixel
        table by traversal-path key every `every` frames without a camera move."""
        self._check(self._lib.acr_host_renderer_set_adaptive(self._h, 1 if on else 0, every), "setAdaptiveAssignment")

    def load_scene(self, scene):
        self._check(self._lib.acr_host_renderer_load_scene(self._h, scene.handle), "loadScene")

    def render(self):
        self._check(self._lib.acr_host_renderer_render(self._h), "render")

    def move_camera(self, pos, dir):
        p = (C.c_float * 2)(*pos)
        d = (C.c_float * 2)(*dir)
        self._check(self._lib.acr_host_renderer_move_camera(self._h, p, d), "moveCamera")

    def set_camera(self, cam):
        self._check(self._lib.acr_host_renderer_set_camera(self._h, C.byref(cam)), "setCamera")

    def get_camera(self):
        cam = _capi.Camera()
        self._check(self._lib.acr_host_renderer_get_camera(self._h, C.byref(cam)), "getCamera")
        return cam

    @property
    def frames_no_move(self):
        return int(self._lib.acr_host_renderer_frames_no_move(self._h))

    def read(self, primid=False, hit_t=False):
        rgba = np.zeros((self.height, self.width, 4), dtype=np.float32)
        pid = np.zeros((self.height, self.width), dtype=np.uint32) if primid else None
        ht = np.zeros((self.height, self.width), dtype=np.float32) if hit_t else None
        self._check(self._lib.acr_host_renderer_read(self._h, rgba.ctypes.data_as(_capi.c_float_p), rgba.size,
                                                     None if pid is None else pid.ctypes.data_as(_capi.c_u32_p),
                                                     None if ht is None else ht.ctypes.data_as(_capi.c_float_p)), "read")
        return rgba, pid, ht

    def stats(self):
        s = _capi.Stats()
        self._check(self._lib.acr_host_renderer_stats(self._h, C.byref(s)), "stats")
        return s

    def save(self, path):
        self._check(self._lib.acr_host_renderer_save(self._h, str(path).encode()), "save")
