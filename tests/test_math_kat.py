"""Known-answer tests: the oracle and the product's acr::math against vectors
produced from the reference's vendored GLM 0.9.6.3 headers
(tests/golden/glm_kat.json, made by oracle/ref_glm_kat.cpp), and the
reference's own scalar helpers (src/math/math.h) oracle vs product."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "glm_kat.json")) as f:
        d = json.load(f)
    assert d["glm_version"] == "0.9.6.3"
    return d["cases"]


def _run(fn, cases):
    bad = []
    for c in cases:
        i = np.array(c["in"] if c["in"] else [0], dtype=np.uint32).view(np.float32)
        o = np.zeros(32, np.float32)
        n = fn(c["fn"].encode(), i.ctypes.data_as(C.POINTER(C.c_float)), o.ctypes.data_as(C.POINTER(C.c_float)))
        exp = np.array(c["out"], dtype=np.uint32)
        if n != len(exp) or not np.array_equal(o[:n].view(np.uint32), exp):
            bad.append(c["fn"])
    return bad


def test_oracle_matches_glm_vectors(orc, kat):
    assert len(kat) > 400
    assert _run(orc.lib().orc_kat, kat) == []


def test_product_math_matches_glm_vectors(kat):
    from cudaraytracer_amd import _capi
    assert _run(_capi.host_lib().acr_host_kat, kat) == []


def test_fast_trig_oracle_equals_product(orc):
    from cudaraytracer_amd import _capi
    L, H = orc.lib(), _capi.host_lib()
    xs = np.concatenate([np.linspace(-7, 7, 20001, dtype=np.float32), np.float32([0.0, -0.0, 1e-30, 3.1415927, 6.2831855, 0.39269909])])
    for x in xs:
        x = float(x)
        for a, b in ((L.orc_fastersin, H.acr_host_fastersin), (L.orc_fastercos, H.acr_host_fastercos), (L.orc_fastertanfull, H.acr_host_fastertanfull)):
            ra, rb = np.float32(a(x)), np.float32(b(x))
            assert ra.view(np.uint32) == rb.view(np.uint32) or (np.isnan(ra) and np.isnan(rb)), x


def test_fast_trig_is_a_sane_approximation(orc):
    """math.h:31-75 are approximations valid on [-pi, pi]: sin within 1e-3 and cos within 7e-3 of libm there."""
    L = orc.lib()
    for x in np.linspace(-3.14, 3.14, 629):
        assert abs(L.orc_fastersin(float(x)) - np.sin(x)) < 1e-3
        assert abs(L.orc_fastercos(float(x)) - np.cos(x)) < 7e-3
    # half field of view used by the shipped scenes: tan(pi/8)
    assert abs(L.orc_fastertanfull(float(np.float32(np.pi / 8))) - np.tan(np.pi / 8)) < 5e-3


def test_rng_stream_oracle_equals_product_and_range(orc):
    from cudaraytracer_amd import _capi
    L, H = orc.lib(), _capi.host_lib()
    for seed in (0, 1, 12345, 2**31 - 1, 2**32 + 17, 2**63 + 5):
        a = np.zeros(4096, np.float32)
        b = np.zeros(4096, np.float32)
        L.orc_rng_uniforms(seed, len(a), a.ctypes.data_as(C.POINTER(C.c_float)))
        H.acr_host_rng_uniforms(seed, len(b), b.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert a.min() > 0.0 and a.max() <= 1.0  # curand_uniform's (0,1]
        assert 0.45 < a.mean() < 0.55


def test_rng_neighbouring_seeds_decorrelated(orc):
    """per-pixel seeds are index + seed (renderer.cu:358): adjacent seeds must not give adjacent streams"""
    L = orc.lib()
    first = []
    for seed in range(1000, 1256):
        a = np.zeros(2, np.float32)
        L.orc_rng_uniforms(seed, 2, a.ctypes.data_as(C.POINTER(C.c_float)))
        first.append(a.copy())
    first = np.array(first)
    assert len(np.unique(first[:, 0])) == 256
    assert abs(np.corrcoef(first[:-1, 0], first[1:, 0])[0, 1]) < 0.25


def test_ray_triangle_known_answers(orc):
    """math.h:126-172: hand-checked cases incl. the epsilon rejections"""
    L = orc.lib()
    f3 = lambda *v: np.array(v, np.float32)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    a, b, c = f3(0, 0, 0), f3(1, 0, 0), f3(0, 1, 0)
    bary = np.zeros(3, np.float32)
    t = C.c_float()

    def hit(ro, rd):
        return L.orc_ray_triangle(P(ro), P(rd), P(a), P(b), P(c), P(bary), C.byref(t))
    assert hit(f3(0.25, 0.25, 1), f3(0, 0, -1)) == 1 and t.value == 1.0
    assert np.allclose(bary, [0.5, 0.25, 0.25])
    assert hit(f3(0.25, 0.25, -1), f3(0, 0, 1)) == 1  # two-sided
    assert hit(f3(0.25, 0.25, 1), f3(0, 0, 1)) == 0   # behind
    assert hit(f3(0.25, 0.25, 0.0005), f3(0, 0, -1)) == 0  # t < eps
    assert hit(f3(2, 2, 1), f3(0, 0, -1)) == 0        # outside
    assert hit(f3(0.25, 0.25, 1), f3(1, 0, 0)) == 0   # parallel: |det| < eps
    assert hit(f3(0, 0, 1), f3(0, 0, -1)) == 1        # exactly on a vertex: bary (1,0,0) accepted


def test_slab_known_answers(orc):
    """geometry.h:44-88 incl. zero direction components (inf/NaN paths) and zero-thickness boxes"""
    L = orc.lib()
    f3 = lambda *v: np.array(v, np.float32)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    big = float(np.finfo(np.float32).max)
    mn, mx = f3(-1, -1, -1), f3(1, 1, 1)
    assert L.orc_slab(P(mn), P(mx), P(f3(0, 0, 5)), P(f3(0, 0, -1)), big) == 1   # axis-aligned, d has zeros
    assert L.orc_slab(P(mn), P(mx), P(f3(2, 0, 5)), P(f3(0, 0, -1)), big) == 0   # misses beside the box
    assert L.orc_slab(P(mn), P(mx), P(f3(0, 0, 5)), P(f3(0, 0, 1)), big) == 0    # box behind: tmax < eps
    assert L.orc_slab(P(mn), P(mx), P(f3(0, 0, 5)), P(f3(0, 0, -1)), 3.0) == 0   # tmin (4) >= current t
    assert L.orc_slab(P(mn), P(mx), P(f3(0, 0, 0)), P(f3(0.6, 0.8, 0)), big) == 1  # origin inside
    flat_mn, flat_mx = f3(-1, -1, 0), f3(1, 1, 0)
    assert L.orc_slab(P(flat_mn), P(flat_mx), P(f3(0, 0, 5)), P(f3(0, 0, -1)), big) == 1  # zero-thickness box
    # origin exactly on the max-x plane with d.x == 0: tmax.x = (1 - 1) * inf = NaN.  Every comparison with NaN
    # is false, so the NaN is never replaced (geometry.h:58-73 use if/assign, not fmin/fmax) and the final
    # `tmax > eps` fails: a ray sliding exactly along a box face misses that box.
    assert L.orc_slab(P(mn), P(mx), P(f3(1, 0, 5)), P(f3(0, 0, -1)), big) == 0
    # on the min-x plane the NaN lands in tmin instead, where `tymin > tmin` is false too, and `tmin < t` fails
    assert L.orc_slab(P(mn), P(mx), P(f3(-1, 0, 5)), P(f3(0, 0, -1)), big) == 0
