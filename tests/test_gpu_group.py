"""acr_hip_group_*: one frame across several contexts driven by one host thread, through the C ABI.

The box the tests run on has one GPU, so every member of a group sits on device 0: the members still have their
own contexts, streams, frame buffers and tile lists, and the pieces still travel between the members' buffers,
which is everything the group does; only the link is missing.  The bar is the single-context path's: frames bit
for bit, against one context rendering the same frames and against the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _single(acr, b, w, h, frames, bounces, camera=None):
    """the same sequence of frames on one context: list of (seed, frames) -> framebuffer"""
    core = acr.HipCore(0)
    try:
        core.upload_scene(b.scene.soa)
        core.set_camera(camera if camera is not None else b.scene.camera)
        core.resize(w, h)
        for seed, f in frames:
            core.render(w, h, seed=seed, frames=f, max_bounces=bounces)
        core.sync()
        return core.read_framebuffer()
    finally:
        core.close()


def _oracle(b, w, h, frames, bounces):
    screen = None
    for seed, f in frames:
        screen = b.oracle.render(w, h, seed=seed, frames=f, max_bounces=bounces, screen=screen)["screen"]
    return screen


@pytest.mark.parametrize("members,w,h", [(3, 640, 360), (2, 333, 187), (5, 256, 96)])
def test_accumulation_across_members_with_a_rebalance_in_the_middle(acr, built, members, w, h):
    """frames 0..5 with the camera still (renderer.cu:375-387): every member keeps the running mean of its bands,
    the split is rebuilt from measured costs while the mean is running (the moved bands' means travel to their new
    owners), the frame is assembled only when it is read"""
    b = built["cornell-box-cycles"]
    g = acr.HipGroup([0] * members)
    try:
        g.upload_scene(b.scene.soa)
        g.set_camera(b.scene.camera)
        g.resize(w, h)
        g.set_balance(rebalance_every=2)
        frames = [(100 + f, f) for f in range(6)]
        splits = []
        for seed, f in frames:
            splits.append(g.split()[0])
            g.render(seed=seed, frames=f, max_bounces=2)
        got = g.read_framebuffer()
        nbands = g.split()[1]
        assert nbands == (h + 15) // 16
        uniform = splits[0]
        assert uniform[0] == 0 and uniform[-1] == nbands and max(np.diff(uniform)) - min(np.diff(uniform)) <= 1
        if nbands >= 2 * members:
            assert any(s != uniform for s in splits[1:]), "the measured costs never moved a band: %s" % splits
        ref = _single(acr, b, w, h, frames, 2)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(got.view(np.uint32), _oracle(b, w, h, frames, 2).view(np.uint32))
        # reading again exchanges nothing new and gives the same frame; two more frames continue the same mean
        assert np.array_equal(g.read_framebuffer().view(np.uint32), got.view(np.uint32))
        more = [(106, 6), (107, 7)]
        for seed, f in more:
            g.render(seed=seed, frames=f, max_bounces=2)
        assert np.array_equal(g.read_framebuffer().view(np.uint32), _single(acr, b, w, h, frames + more, 2).view(np.uint32))
        # the camera moves: uniform split again, the mean restarts with a frame that overwrites
        cam = type(b.scene.camera)()
        C.memmove(C.byref(cam), C.byref(b.scene.camera), C.sizeof(cam))
        cam.position[0] += 0.25
        g.set_camera(cam)
        assert g.split()[0] == uniform
        after = [(200, 0), (201, 1), (202, 2)]
        for seed, f in after:
            g.render(seed=seed, frames=f, max_bounces=2)
        assert np.array_equal(g.read_framebuffer().view(np.uint32), _single(acr, b, w, h, after, 2, camera=cam).view(np.uint32))
    finally:
        g.close()


def test_every_frame_gathered_while_the_next_one_renders(acr, built):
    """frames == 0 on every frame (a moving camera): frames alternate between two buffers per member, the pieces of
    frame k travel while frame k + 1 renders, and the frame that is read is the last one queued"""
    b = built["cornell-box-cycles"]
    w, h = 480, 270
    g = acr.HipGroup([0, 0, 0, 0])
    try:
        g.upload_scene(b.scene.soa)
        g.set_camera(b.scene.camera)
        g.resize(w, h)
        g.set_balance(rebalance_every=1)
        seen = {}
        for k in range(7):
            g.render(seed=300 + k, frames=0, max_bounces=1, gather=True)
            if k in (2, 6):
                seen[k] = g.read_framebuffer()
        for k, img in seen.items():
            assert np.array_equal(img.view(np.uint32), _single(acr, b, w, h, [(300 + k, 0)], 1).view(np.uint32)), k
        # the assembled frame's device address, for callers that go on working on the GPU
        assert g.frame_ptr()
        assert np.array_equal(g.read_framebuffer().view(np.uint32), seen[6].view(np.uint32))
    finally:
        g.close()


def test_group_on_scenes_that_list_pixels_and_on_trees_read_from_device_memory(acr, orc, scene_data):
    """bands written in place (acr_frame::tiles_into_frame) by every kind of launch: pixels handed to the six-plane
    list pass (an object with a singular transform), the wave-cooperative kernel (bounces), and mesh trees walked
    near side first from device memory by the lanes of a wave together (a soup; a scene of several large objects), each
    accumulating over frames with the split re-dealt in between"""
    import copy
    from conftest import Built, instanced_cloud_scene, soup_data
    d = copy.deepcopy(scene_data["cornell-box-cycles"])
    k = [n["name"] for n in d.nodes].index("short_box")
    m = np.array(d.nodes[k]["transform"], dtype=np.float32).reshape(4, 4)
    m[2, :3] = 0.0  # the image of the local z axis: the object is squashed flat, rays that reach it are not finite in its space
    d.nodes[k]["transform"] = [float(v) for v in m.reshape(-1)]
    cases = [(Built(acr, orc, d), 320, 180, 3, 3),
             (Built(acr, orc, soup_data(acr, 3000, seed=2, edge=0.3, extent=1.5), mesh_depth=10), 200, 150, 1, 2),
             (Built(acr, orc, soup_data(acr, 60000, seed=3, edge=0.12, extent=2.0), mesh_depth=14), 256, 160, 2, 4),
             (Built(acr, orc, instanced_cloud_scene(acr, 8), mesh_depth=14), 240, 144, 3, 3)]  # several objects in device memory
    for b, w, h, bounces, members in cases:
        g = acr.HipGroup([0] * members)
        try:
            g.upload_scene(b.scene.soa)
            g.set_camera(b.scene.camera)
            g.resize(w, h)
            g.set_balance(rebalance_every=1)
            frames = [(40 + f, f) for f in range(5)]
            for seed, f in frames:
                g.render(seed=seed, frames=f, max_bounces=bounces, collect_stats=1)
            st = g.read_stats()
            got = g.read_framebuffer()
            assert np.array_equal(got.view(np.uint32), _oracle(b, w, h, frames, bounces).view(np.uint32))
            c = b.oracle.render(w, h, seed=frames[-1][0], frames=0, max_bounces=bounces)["counters"]
            assert (st.rays_primary, st.rays_shadow, st.rays_bounce) == (c["rays_primary"], c["rays_shadow"], c["rays_bounce"])
        finally:
            g.close()


def test_ray_counters_and_weights(acr, built):
    b = built["cornell-box-cycles"]
    w, h = 320, 180
    g = acr.HipGroup([0, 0, 0])
    try:
        g.upload_scene(b.scene.soa)
        g.set_camera(b.scene.camera)
        g.resize(w, h)
        g.set_balance(rebalance_every=0, weights=[0.5, 0.25, 0.25])
        begin, nbands = g.split()
        assert begin[1] - begin[0] > begin[2] - begin[1] and begin[-1] == nbands
        g.render(seed=9, frames=0, max_bounces=3, collect_stats=1)
        st = g.read_stats()
        ref = b.oracle.render(w, h, seed=9, frames=0, max_bounces=3)
        c = ref["counters"]
        assert (st.rays_primary, st.rays_shadow, st.rays_bounce, st.primary_hits) == (
            c["rays_primary"], c["rays_shadow"], c["rays_bounce"], c["primary_hits"])
        assert np.array_equal(g.read_framebuffer().view(np.uint32), ref["screen"].view(np.uint32))
        # a member without a share renders nothing and the frame is still whole
        g.set_balance(rebalance_every=0, weights=[1.0, 0.0, 1.0])
        begin, _ = g.split()
        assert begin[2] == begin[1]
        g.render(seed=10, frames=0, max_bounces=1)
        assert np.array_equal(g.read_framebuffer().view(np.uint32), b.oracle.render(w, h, seed=10, frames=0, max_bounces=1)["screen"].view(np.uint32))
    finally:
        g.close()


def test_group_error_paths(acr, built):
    b = built["cornell-box-cycles"]
    with pytest.raises(RuntimeError, match="no usable GPU"):
        acr.HipGroup([0, 9999])
    g = acr.HipGroup([0, 0])
    try:
        with pytest.raises(RuntimeError, match="scene and camera"):
            g.resize(64, 64)
            g.render(seed=1)
        g.upload_scene(b.scene.soa)
        g.set_camera(b.scene.camera)
        with pytest.raises(RuntimeError, match="earlier frame"):
            g.render(seed=1, frames=3)
        from cudaraytracer_amd import _capi
        f = _capi.Frame()
        f.width, f.height, f.max_bounces = 64, 64, 1
        ids = np.zeros(1, np.uint32)
        f.num_tiles, f.tile_w, f.tile_h, f.tile_ids = 1, 16, 16, ids.ctypes.data_as(_capi.c_u32_p)
        assert _capi.hip_lib().acr_hip_group_render(g._h, C.byref(f), 0) != 0
        f = _capi.Frame()
        f.width, f.height, f.max_bounces = 128, 64, 1
        assert _capi.hip_lib().acr_hip_group_render(g._h, C.byref(f), 0) != 0  # not the size of the last resize
        g.render(seed=1)
        g.sync()
    finally:
        g.close()


def test_renderer_across_devices(acr, built):
    """acr::Renderer with Args::devices: the reference's render loop (loadScene, render x n, moveCamera, render) over
    a group; frames equal the one-GPU renderer's"""
    b = built["cornell-box-cycles"]
    w, h = 400, 240

    def run(devices):
        r = acr.Renderer(w, h, samples=1, devices=devices, rebalance_every=2)
        try:
            r.configure(max_bounces=2, fixed_seed=77)
            r.load_scene(b.scene)
            for _ in range(5):
                r.render()
            a = r.read()[0]
            r.move_camera((0.3, -0.2), (0.05, 0.02))
            for _ in range(3):
                r.render()
            assert r.frames_no_move == 3
            return a, r.read()[0]
        finally:
            r.close()

    one = run([0])
    many = run([0, 0, 0])
    assert np.array_equal(one[0].view(np.uint32), many[0].view(np.uint32))
    assert np.array_equal(one[1].view(np.uint32), many[1].view(np.uint32))


def test_tiles_into_frame_through_one_context(acr, built):
    """acr_frame::tiles_into_frame on its own: listed tiles are written in place into a frame-shaped buffer,
    unlisted pixels are left alone, for a width that is not a multiple of the tile width"""
    import torch
    b = built["cornell-box-cycles"]
    w, h = 300, 170
    core = acr.HipCore(0)
    try:
        core.upload_scene(b.scene.soa)
        core.set_camera(b.scene.camera)
        ref = b.oracle.render(w, h, seed=4, frames=0, max_bounces=2)["screen"]
        frame = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
        tw, th = 64, 32
        tx, ty = (w + tw - 1) // tw, (h + th - 1) // th
        ids = np.array([t for t in range(tx * ty) if t % 2 == 0][::-1], np.uint32)
        core.set_stream(torch.cuda.current_stream().cuda_stream)
        core.render(w, h, seed=4, frames=0, max_bounces=2, tile_ids=ids, tile_w=tw, tile_h=th, tiles_into_frame=True,
                    screen_dev=C.c_void_p(frame.data_ptr()))
        core.sync()
        got = frame.cpu().numpy()
        for t in range(tx * ty):
            x0, y0 = (t % tx) * tw, (t // tx) * th
            piece = got[y0:y0 + th, x0:x0 + tw]
            if t % 2 == 0:
                assert np.array_equal(piece.view(np.uint32), ref[y0:y0 + th, x0:x0 + tw].view(np.uint32)), t
            else:
                assert np.all(piece == -1.0), t
    finally:
        core.use_own_stream()
        core.close()
