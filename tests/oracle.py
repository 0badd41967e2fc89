"""ctypes bindings of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never by the product package."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(HERE, "..", "oracle", "liboracle.so")

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)


class OrcCamera(C.Structure):
    _fields_ = [("aspect", C.c_float), ("hfov", C.c_float), ("position", C.c_float * 3), ("forward", C.c_float * 3),
                ("up", C.c_float * 3)]


class OrcNode(C.Structure):
    _fields_ = [("start", C.c_uint32), ("end", C.c_uint32), ("left", C.c_float), ("right", C.c_float),
                ("axis", C.c_uint8), ("is_leaf", C.c_uint8), ("defined", C.c_uint8), ("pad", C.c_uint8)]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_primary", "rays_shadow", "rays_bounce", "primary_hits", "node_visits",
                                          "slab_tests", "object_visits", "mesh_visits", "triangle_tests", "object_hits", "mesh_box_hits",
                                          "order_doubts")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OrcFrame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("seed", C.c_uint64), ("frames", C.c_uint64),
                ("max_bounces", C.c_int32), ("num_threads", C.c_int32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32)]


class OrcAov(C.Structure):
    _fields_ = [("prim", c_u32_p), ("t", c_float_p), ("material", c_u32_p), ("position", c_float_p), ("normal", c_float_p),
                ("path_key", C.POINTER(C.c_uint64))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise OSError(f"{_LIB} missing: run `make -C oracle`")
        L = C.CDLL(_LIB)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_camera.argtypes = [C.c_void_p, C.POINTER(OrcCamera)]
        for n in ("orc_scene_num_objects", "orc_scene_num_meshes", "orc_scene_num_lights", "orc_scene_num_triangles"):
            getattr(L, n).restype = C.c_uint32
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_scene_bounds.argtypes = [C.c_void_p, c_float_p, c_float_p]
        L.orc_scene_bih.argtypes = [C.c_void_p, C.POINTER(OrcNode), C.POINTER(C.c_int32)]
        L.orc_scene_object.argtypes = [C.c_void_p, C.c_uint32, c_float_p]
        L.orc_mesh_bih.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(OrcNode), c_u32_p]
        L.orc_mesh_info.argtypes = [C.c_void_p, C.c_uint32, c_float_p, c_float_p, c_float_p, c_u32_p, c_u32_p]
        L.orc_scene_light.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), c_float_p]
        L.orc_scene_set_mesh_order.restype = None
        L.orc_scene_set_mesh_order.argtypes = [C.c_void_p, C.c_int]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(OrcCamera), C.POINTER(OrcFrame), c_float_p, C.POINTER(C.c_int32),
                                 C.POINTER(OrcAov), C.POINTER(OrcCounters)]
        L.orc_trace_rays.argtypes = [C.c_void_p, C.c_uint32, c_float_p, c_float_p, c_float_p, C.POINTER(C.c_uint8), c_float_p,
                                     c_u32_p, c_float_p, c_float_p]
        for n in ("orc_fastersin", "orc_fastercos", "orc_fastertanfull"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        L.orc_ray_triangle.restype = C.c_int
        L.orc_ray_triangle.argtypes = [c_float_p] * 6 + [c_float_p]
        L.orc_slab.restype = C.c_int
        L.orc_slab.argtypes = [c_float_p] * 4 + [C.c_float]
        L.orc_rng_uniforms.argtypes = [C.c_uint64, C.c_uint32, c_float_p]
        L.orc_move_camera.argtypes = [C.POINTER(OrcCamera), c_float_p, c_float_p]
        L.orc_path_votes.restype = None
        L.orc_path_votes.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_uint32, C.c_int]
        L.orc_vote_keys.restype = None
        L.orc_vote_keys.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.c_uint32]
        L.orc_kat.restype = C.c_int
        L.orc_kat.argtypes = [C.c_char_p, c_float_p, c_float_p]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(c_float_p)


class OracleScene:
    """Scene built by the oracle from a packed import scene (same struct layout as the product's)."""

    def __init__(self, import_view, scene_depth=6, mesh_depth=6):
        self.L = lib()
        self.scene_depth, self.mesh_depth = scene_depth, mesh_depth
        self.h = self.L.orc_scene_create(C.addressof(import_view), scene_depth, mesh_depth)
        if not self.h:
            raise RuntimeError("orc_scene_create failed")
        self.camera = OrcCamera()
        self.L.orc_scene_camera(self.h, C.byref(self.camera))

    def set_mesh_order(self, near_first):
        """near_first: mesh trees are walked near side first (NOT the reference's order; results must not change)."""
        self.L.orc_scene_set_mesh_order(self.h, 1 if near_first else 0)

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_objects(self):
        return self.L.orc_scene_num_objects(self.h)

    @property
    def num_meshes(self):
        return self.L.orc_scene_num_meshes(self.h)

    @property
    def num_lights(self):
        return self.L.orc_scene_num_lights(self.h)

    @property
    def num_triangles(self):
        return self.L.orc_scene_num_triangles(self.h)

    def bounds(self):
        mn = np.zeros(3, np.float32); mx = np.zeros(3, np.float32)
        self.L.orc_scene_bounds(self.h, _fp(mn), _fp(mx))
        return mn, mx

    @staticmethod
    def _nodes(arr):
        n = len(arr)
        out = dict(start=np.zeros(n, np.uint32), end=np.zeros(n, np.uint32), left=np.zeros(n, np.float32),
                   right=np.zeros(n, np.float32), axis=np.zeros(n, np.uint8), is_leaf=np.zeros(n, np.uint8),
                   defined=np.zeros(n, np.uint8))
        raw = np.frombuffer(arr, dtype=np.dtype([("start", "<u4"), ("end", "<u4"), ("left", "<f4"), ("right", "<f4"),
                                                 ("axis", "u1"), ("is_leaf", "u1"), ("defined", "u1"), ("pad", "u1")]))
        for k in out:
            out[k] = raw[k].copy()
        return out

    def scene_bih(self):
        n = (1 << self.scene_depth) - 1
        nodes = (OrcNode * n)()
        order = np.zeros(max(self.num_objects, 1), np.int32)
        self.L.orc_scene_bih(self.h, nodes, order.ctypes.data_as(C.POINTER(C.c_int32)))
        return self._nodes(nodes), order[:self.num_objects]

    def mesh_bih(self, mesh):
        n = (1 << self.mesh_depth) - 1
        nodes = (OrcNode * n)()
        nf = C.c_uint32()
        self.L.orc_mesh_info(self.h, mesh, None, None, None, C.byref(nf), None)
        order = np.zeros(max(nf.value, 1), np.uint32)
        self.L.orc_mesh_bih(self.h, mesh, nodes, order.ctypes.data_as(c_u32_p))
        return self._nodes(nodes), order[:nf.value]

    def mesh_info(self, mesh):
        c = np.zeros(3, np.float32); mn = np.zeros(3, np.float32); mx = np.zeros(3, np.float32)
        nf = C.c_uint32(); mat = C.c_uint32()
        self.L.orc_mesh_info(self.h, mesh, _fp(c), _fp(mn), _fp(mx), C.byref(nf), C.byref(mat))
        return dict(centroid=c, min=mn, max=mx, num_faces=nf.value, material=mat.value)

    def object(self, i):
        out = np.zeros(59, np.float32)
        self.L.orc_scene_object(self.h, i, _fp(out))
        return dict(transform=out[0:16], inv_transform=out[16:32], normal=out[32:41], inv_normal=out[41:50],
                    centroid=out[50:53], min=out[53:56], max=out[56:59])

    def light(self, i):
        t = C.c_int32()
        out = np.zeros(14, np.float32)
        self.L.orc_scene_light(self.h, i, C.byref(t), _fp(out))
        return dict(type=t.value, position=out[0:3], direction=out[3:6], diffuse=out[6:9], att=out[9:12], cones=out[12:14])

    def render(self, width, height, seed=0, frames=0, max_bounces=1, threads=0, screen=None, camera=None, aov=True,
               row_begin=0, row_end=0, pixel_map=None):
        f = OrcFrame(width, height, seed, frames, max_bounces, threads, row_begin, row_end)
        if screen is None:
            screen = np.zeros((height, width, 4), np.float32)
        n = width * height
        res = dict(screen=screen)
        a = None
        if aov:
            res.update(prim=np.full(n, 0xFFFFFFFF, np.uint32), t=np.full(n, np.finfo(np.float32).max, np.float32),
                       material=np.full(n, 0xFFFFFFFF, np.uint32), position=np.zeros((n, 3), np.float32),
                       normal=np.zeros((n, 3), np.float32), path_key=np.zeros(n, np.uint64))
            a = OrcAov(res["prim"].ctypes.data_as(c_u32_p), _fp(res["t"]), res["material"].ctypes.data_as(c_u32_p),
                       _fp(res["position"]), _fp(res["normal"]), res["path_key"].ctypes.data_as(C.POINTER(C.c_uint64)))
        cnt = OrcCounters()
        cam = camera if camera is not None else self.camera
        pm = None if pixel_map is None else pixel_map.ctypes.data_as(C.POINTER(C.c_int32))
        rc = self.L.orc_render(self.h, C.byref(cam), C.byref(f), _fp(screen), pm, None if a is None else C.byref(a), C.byref(cnt))
        if rc != 0:
            raise RuntimeError("orc_render failed")
        res["counters"] = cnt.as_dict()
        if aov:
            res["prim"] = res["prim"].reshape(height, width)
            res["t"] = res["t"].reshape(height, width)
            res["path_key"] = res["path_key"].reshape(height, width)
        return res

    def trace_rays(self, origins, dirs, tmax=None):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(o)
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        hit = np.zeros(n, np.uint8); t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32)
        pos = np.zeros((n, 3), np.float32); nrm = np.zeros((n, 3), np.float32)
        self.L.orc_trace_rays(self.h, n, _fp(o), _fp(d), None if tm is None else _fp(tm), hit.ctypes.data_as(C.POINTER(C.c_uint8)),
                              _fp(t), prim.ctypes.data_as(c_u32_p), _fp(pos), _fp(nrm))
        return dict(hit=hit, t=t, prim=prim, position=pos, normal=nrm)


class PathVotes:
    """The reference's per-pixel PathAggregate table (bih.h:29-66) on the CPU: votes[bit][pixel]."""

    def __init__(self, n):
        self.n = n
        self.votes = np.zeros((64, n), np.int32)

    def vote(self, path_keys, reset):
        k = np.ascontiguousarray(path_keys, np.uint64).reshape(-1)
        assert len(k) == self.n
        lib().orc_path_votes(k.ctypes.data_as(C.POINTER(C.c_uint64)), self.votes.ctypes.data_as(C.POINTER(C.c_int32)), self.n, 1 if reset else 0)

    def keys(self):
        out = np.zeros(self.n, np.uint64)
        lib().orc_vote_keys(self.votes.ctypes.data_as(C.POINTER(C.c_int32)), out.ctypes.data_as(C.POINTER(C.c_uint64)), self.n)
        return out
