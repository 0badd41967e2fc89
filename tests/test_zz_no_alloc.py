"""acr_hip_render_static, the entry point that allocates nothing, against acr_hip_render: same frames bit for
bit, on two calls in a row, the second with a smaller frame; also with every pixel sent through the list pass,
which is what uses the static list, and on a second context of the same GPU (the array is shared)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(core, w, h, static, **kw):
    core.resize(w, h)
    core.render(w, h, collect_stats=True, static=static, **kw)
    core.sync()
    return core.read_framebuffer(), core.read_primid(), core.read_hit_t(), core.read_stats()


def _same(a, b):
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert (a[3].rays_primary, a[3].rays_shadow, a[3].rays_bounce, a[3].primary_hits) == (
        b[3].rays_primary, b[3].rays_shadow, b[3].rays_bounce, b[3].primary_hits)


def test_render_static_equals_render(acr, built):
    b = built["cornell-box-cycles"]
    core = acr.HipCore(0)
    other = acr.HipCore(0)
    try:
        for c in (core, other):
            c.upload_scene(b.scene.soa)
            c.set_camera(b.scene.camera)
        # two calls in a row, the second smaller; the oracle as a third witness on the small one
        big = _frames(core, 1920, 1080, True, seed=11, frames=0, max_bounces=1)
        small = _frames(core, 200, 120, True, seed=12, frames=0, max_bounces=3)
        _same(big, _frames(core, 1920, 1080, False, seed=11, frames=0, max_bounces=1))
        _same(small, _frames(core, 200, 120, False, seed=12, frames=0, max_bounces=3))
        ref = b.oracle.render(200, 120, seed=12, frames=0, max_bounces=3)
        assert np.array_equal(small[1], ref["prim"]) and np.array_equal(small[0].view(np.uint32), ref["screen"].view(np.uint32))
        # every slot through the static list (the list pass), large then small, then a normal call again
        core.set_traversal(list_pass=True)
        try:
            l1 = _frames(core, 640, 360, True, seed=13, frames=0, max_bounces=2)
            l2 = _frames(core, 96, 64, True, seed=14, frames=0, max_bounces=2)
        finally:
            core.set_traversal()
        assert l1[3].six_plane_pixels >= 640 * 360 and l2[3].six_plane_pixels >= 96 * 64
        _same(l1, _frames(core, 640, 360, False, seed=13, frames=0, max_bounces=2))
        _same(l2, _frames(core, 96, 64, False, seed=14, frames=0, max_bounces=2))
        # another context of the same GPU finds what the last call left in the shared array
        o1 = _frames(other, 320, 200, True, seed=15, frames=0, max_bounces=2)
        _same(o1, _frames(core, 320, 200, False, seed=15, frames=0, max_bounces=2))
        # accumulation over frames through the static entry point
        core.resize(160, 90)
        other.resize(160, 90)
        for f in range(4):
            core.render(160, 90, seed=50 + f, frames=f, max_bounces=2, static=True)
            other.render(160, 90, seed=50 + f, frames=f, max_bounces=2)
        core.sync()
        other.sync()
        assert np.array_equal(core.read_framebuffer().view(np.uint32), other.read_framebuffer().view(np.uint32))
    finally:
        core.close()
        other.close()


def test_render_static_allocates_nothing(acr, built):
    """The property the entry point exists for: a warmed context's static calls leave the device's free memory
    where it was -- frames of new sizes, new bounce counts and a scene switch included (each of which makes the
    allocating entry point grow or create a buffer; shown on a second context as the control)."""
    import torch
    b = built["cornell-box-cycles"]
    core = acr.HipCore(0)
    other = acr.HipCore(0)
    screen = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
    ptr = C.c_void_p(screen.data_ptr())
    try:
        for c in (core, other):
            c.upload_scene(b.scene.soa)
            c.set_camera(b.scene.camera)
        core.render(64, 64, seed=1, max_bounces=1, static=True, screen_dev=ptr)  # loads the kernels' code objects
        core.render(64, 64, seed=1, max_bounces=3, static=True, screen_dev=ptr)
        core.sync()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        for k, (w, h, bounces) in enumerate([(1920, 1080, 1), (640, 360, 4), (1280, 720, 2), (33, 17, 5)]):
            core.render(w, h, seed=20 + k, frames=0, max_bounces=bounces, static=True, screen_dev=ptr)
            core.render(w, h, seed=30 + k, frames=1, max_bounces=bounces, static=True, screen_dev=ptr)
        core.sync()
        assert torch.cuda.mem_get_info(0)[0] == free0, "acr_hip_render_static changed the device's free memory"
        # control: the allocating entry point does take memory for the same frames (its list of pixel slots)
        other.render(1920, 1080, seed=20, frames=0, max_bounces=1, screen_dev=ptr)
        other.sync()
        assert torch.cuda.mem_get_info(0)[0] < free0
    finally:
        core.close()
        other.close()
        del screen


def test_render_static_refuses_what_would_allocate(acr, orc, built):
    from conftest import Built, soup_data
    b = built["cornell-box-cycles"]
    core = acr.HipCore(0)
    try:
        core.upload_scene(b.scene.soa)
        core.set_camera(b.scene.camera)
        core.resize(64, 64)
        with pytest.raises(RuntimeError, match="whole frames only"):
            core.render(64, 64, tile_ids=np.array([0], np.uint32), tile_w=16, tile_h=16, static=True,
                        screen_dev=C.c_void_p(core.framebuffer_ptr()))
        with pytest.raises(RuntimeError, match="pixel slots"):
            core.render(3840, 2160, static=True, screen_dev=C.c_void_p(core.framebuffer_ptr()))
        deep = Built(acr, orc, soup_data(acr, 200000, seed=4, edge=0.12, extent=2.0), mesh_depth=18)
        core.upload_scene(deep.scene.soa)
        core.set_camera(deep.scene.camera)
        core.render(64, 64, static=True, collect_stats=True)  # one bounce level: every level of the waiting-child column fits LDS
        core.sync()
        assert core.read_stats().shared_walk == 1  # one deep mesh: the lanes share their walks here too
        fixed = core.read_framebuffer().copy()
        core.render(64, 64)
        core.sync()
        assert np.array_equal(core.read_framebuffer().view(np.uint32), fixed.view(np.uint32))
        with pytest.raises(RuntimeError, match="levels deeper"):
            core.render(64, 64, max_bounces=12, static=True)  # the direct-term columns of 11 levels push part of it out
        core.render(64, 64, max_bounces=12)  # the allocating entry point still serves it
        core.sync()
    finally:
        core.close()
