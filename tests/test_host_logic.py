"""Host-side logic that needs no GPU: camera motion, tile order and
cost-balanced tile partition."""
import ctypes as C

import numpy as np
import pytest

from cudaraytracer_amd import _capi
from cudaraytracer_amd import distributed as D


def _cam(pos, fwd, up):
    c = _capi.Camera()
    c.aspect_ratio, c.horizontal_fov = 1.5, 0.7853982
    c.position = (C.c_float * 3)(*pos)
    c.forward = (C.c_float * 3)(*fwd)
    c.up = (C.c_float * 3)(*up)
    return c


def test_move_camera_product_equals_oracle(orc):
    """Renderer::moveCamera arithmetic (renderer.cu:152-161): C++ host vs oracle, bit for bit, along a path of moves."""
    H, L = _capi.host_lib(), orc.lib()
    pc = _cam([-2.78, -7.674, 2.73], [0, 1, 0], [0, 0, 1])
    oc = orc.OrcCamera(1.5, 0.7853982, pc.position, pc.forward, pc.up)
    rng = np.random.default_rng(3)
    for _ in range(50):
        pos = (C.c_float * 2)(*rng.uniform(-0.5, 0.5, 2))
        dr = (C.c_float * 2)(*rng.uniform(-0.05, 0.05, 2))
        assert H.acr_host_move_camera(C.byref(pc), pos, dr) == 0
        L.orc_move_camera(C.byref(oc), pos, dr)
        for k in ("position", "forward", "up"):
            a = np.array(list(getattr(pc, k)), np.float32).view(np.uint32)
            b = np.array(list(getattr(oc, k)), np.float32).view(np.uint32)
            assert np.array_equal(a, b), k


def test_move_camera_geometry():
    H = _capi.host_lib()
    c = _cam([0, 0, 0], [0, 0, -1], [0, 1, 0])
    # forward by 2: position follows the view direction
    H.acr_host_move_camera(C.byref(c), (C.c_float * 2)(0, 2), (C.c_float * 2)(0, 0))
    assert np.allclose(list(c.position), [0, 0, -2], atol=1e-6)
    # strafe right by 1: right = forward x up = +x
    H.acr_host_move_camera(C.byref(c), (C.c_float * 2)(1, 0), (C.c_float * 2)(0, 0))
    assert np.allclose(list(c.position), [1, 0, -2], atol=1e-6)
    # yaw by +90 degrees about up turns -z into -x
    H.acr_host_move_camera(C.byref(c), (C.c_float * 2)(0, 0), (C.c_float * 2)(np.pi / 2, 0))
    assert np.allclose(list(c.forward), [-1, 0, 0], atol=1e-6) and np.allclose(list(c.up), [0, 1, 0], atol=1e-6)
    assert H.acr_host_move_camera(None, None, None) != 0


@pytest.mark.parametrize("tx,ty", [(1, 1), (4, 4), (30, 68), (7, 3), (33, 1)])
def test_tile_order_is_a_permutation_with_locality(tx, ty):
    o = D.tile_order(tx, ty)
    assert sorted(o.tolist()) == list(range(tx * ty))
    if tx >= 4 and ty >= 4:
        # consecutive tiles of a Z-curve are close: mean grid distance well below a row-major scan's wrap-around
        x, y = o % tx, o // tx
        step = np.abs(np.diff(x.astype(int))) + np.abs(np.diff(y.astype(int)))
        assert step.mean() < 3.0


def test_partition_uniform_and_weighted():
    order = D.tile_order(16, 8)
    b = D.partition(order, None, 4)
    assert b.tolist() == [0, 32, 64, 96, 128]
    b1 = D.partition(order, None, 1)
    assert b1.tolist() == [0, 128]
    # all the cost in the first quarter of the curve: the cuts move into it
    cost = np.ones(128, np.float32) * 0.01
    cost[order[:32]] = 10.0
    b2 = D.partition(order, cost, 4)
    assert b2[0] == 0 and b2[-1] == 128 and np.all(np.diff(b2.astype(int)) >= 0)
    sums = [cost[order[b2[k]:b2[k + 1]]].sum() for k in range(4)]
    assert max(sums) / (sum(sums) / 4) < 1.2
    # more parts than tiles: trailing parts are empty, nothing is lost
    b3 = D.partition(D.tile_order(2, 1), None, 4)
    assert b3[0] == 0 and b3[-1] == 2 and np.all(np.diff(b3.astype(int)) >= 0)
    # zero / negative / NaN-free costs are clamped, not trusted
    b4 = D.partition(order, np.zeros(128, np.float32), 4)
    assert b4[0] == 0 and b4[-1] == 128
    # no part with a share is left without a tile, wherever the cost sits and however small the share
    bands = np.arange(68, dtype=np.uint32)
    for hot in (0, 33, 67):
        c = np.full(68, 1e-3, np.float32)
        c[hot] = 1e6
        for w in (None, [0.9, 0.03, 0.03, 0.04], [0.97, 0.01, 0.01, 0.01]):
            b5 = D.partition(bands, c, 4, w).astype(int)
            assert b5[0] == 0 and b5[-1] == 68 and np.all(np.diff(b5) >= 1), (hot, w, b5)
    b6 = D.partition(bands, None, 4, [1.0, 0.0, 1.0, 1.0]).astype(int)   # a part with no share may stay empty
    assert b6[0] == 0 and b6[-1] == 68 and b6[2] - b6[1] <= 1 and np.all(np.diff(b6) >= 0)


def test_partition_rejects_bad_input():
    H = _capi.host_lib()
    order = np.array([0, 1, 5], np.uint32)  # 5 is not a tile of a 3-tile frame
    begin = np.zeros(3, np.uint32)
    assert H.acr_host_partition_tiles(order.ctypes.data_as(_capi.c_u32_p), None, 3, 2, begin.ctypes.data_as(_capi.c_u32_p)) != 0
    assert H.acr_host_partition_tiles(None, None, 3, 2, begin.ctypes.data_as(_capi.c_u32_p)) != 0
    assert H.acr_host_tile_order(0, 4, begin.ctypes.data_as(_capi.c_u32_p)) != 0


def test_tiled_frame_reset_and_update():
    p = D.TiledFrame(1920, 1080, 8, 64, 16)
    assert p.tiles_x == 30 and p.tiles_y == 68 and p.num_tiles == 2040
    sizes = [len(p.tiles_of(r)) for r in range(8)]
    assert sum(sizes) == 2040 and max(sizes) - min(sizes) <= 1
    assert sorted(np.concatenate([p.tiles_of(r) for r in range(8)]).tolist()) == list(range(2040))
    cost = np.ones(2040, np.float32)
    cost[: 2040 // 2] = 3.0
    p.update(cost)
    loads = [cost[p.tiles_of(r)].sum() for r in range(8)]
    assert max(loads) / (sum(loads) / 8) < 1.05
    p.reset()
    assert [len(p.tiles_of(r)) for r in range(8)] == sizes


def test_weighted_partition_of_tiles():
    """shares in the ratio of the weights; equal weights reproduce the unweighted split; bad weights are refused"""
    from cudaraytracer_amd import _capi
    from cudaraytracer_amd import distributed as D
    order = np.arange(100, dtype=np.uint32)
    assert list(D.partition(order, None, 4)) == [0, 25, 50, 75, 100]
    assert list(D.partition(order, None, 4, [1, 1, 1, 1])) == [0, 25, 50, 75, 100]
    assert list(D.partition(order, None, 4, [4, 2, 2, 2])) == [0, 40, 60, 80, 100]
    assert list(D.partition(order, None, 2, [1, 0])) == [0, 100, 100]
    cost = np.ones(100, np.float32)
    cost[:50] = 3.0  # the first half of the tiles is three times as expensive
    b = D.partition(order, cost, 2, [3, 1])  # 150 of the 200 cost units for part 0
    assert list(b) == [0, 50, 100]
    H = _capi.host_lib()
    begin = np.zeros(3, np.uint32)
    neg = np.array([1, -1], np.float32)
    assert H.acr_host_partition_tiles_weighted(order.ctypes.data_as(_capi.c_u32_p), None, 100, 2, neg.ctypes.data_as(_capi.c_float_p),
                                               begin.ctypes.data_as(_capi.c_u32_p)) != 0
    zero = np.zeros(2, np.float32)
    assert H.acr_host_partition_tiles_weighted(order.ctypes.data_as(_capi.c_u32_p), None, 100, 2, zero.ctypes.data_as(_capi.c_float_p),
                                               begin.ctypes.data_as(_capi.c_u32_p)) != 0


def test_shares_from_measured_times():
    """solve_shares: equal time per frame for every rank; rank 0 (which sends nothing) gets the largest piece when
    sending costs more than rendering; one measurement = proportional model, two = fixed cost + cost per band"""
    from cudaraytracer_amd.distributed import solve_shares

    def measure(bands, render_fixed=5e-6, render_per=3.2e-6, send_fixed=20e-6, send_per=9.7e-6):
        bands = np.asarray(bands, float)
        t = np.zeros((len(bands), 2))
        t[:, 0] = render_fixed + render_per * bands
        t[:, 1] = send_fixed + send_per * bands
        t[0, 1] = 0.0
        return bands, t

    def frame_times(w, n=68):
        b, t = measure(w * n)
        per = np.maximum(t[:, 0], t[:, 1])
        per[0] = t[0, 0]
        return per

    for world in (2, 4, 8):
        first = measure([68.0 / world] * world)
        w1 = solve_shares([first], 68)
        assert abs(w1.sum() - 1) < 1e-9 and w1[0] > w1[1] and np.allclose(w1[1:], w1[1])
        second = measure(np.round(w1 * 68))
        w2 = solve_shares([first, second], 68)
        per = frame_times(w2)
        assert per.max() / per.min() < 1.02           # the affine model is exact here: every rank takes the same time
        assert frame_times(w2).max() < frame_times(np.full(world, 1.0 / world)).max()   # better than equal shares
    # a rank that all but cannot send is held at one band (and the caller is warned), the others share the frame
    import pytest
    b, t = measure([17.0] * 4)
    t[2, 1] = 10.0
    with pytest.warns(RuntimeWarning, match="less than one band"):
        w = solve_shares([(b, t)], 68)
    assert 0.9 / 68 < w[2] < 1.1 / 68 and abs(w.sum() - 1) < 1e-9 and w[2] == w.min()
