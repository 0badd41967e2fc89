"""GPU parity: the HIP path, called through the C ABI of include/acr_hip.h,
against the CPU oracle on the same scenes, cameras and seeds.

Bar: primitive ids, hit distances and RGBA all bit-exact (the radiance
tolerance the task allows, 1e-4 relative, is not needed: both sides perform the
same IEEE binary32 operations in the same order)."""
import ctypes as C

import numpy as np
import pytest

from conftest import Built, duplicate_triangle_scene, instanced_cloud_scene, random_instanced_scene, soup_data

pytestmark = pytest.mark.gpu

FMAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def core(acr):
    c = acr.HipCore(0)
    yield c
    c.close()


def load(core, b):
    core.upload_scene(b.scene.soa)
    core.set_camera(b.scene.camera)


def render(core, w, h, **kw):
    core.resize(w, h)
    core.render(w, h, collect_stats=True, **kw)
    core.sync()
    return core.read_framebuffer(), core.read_primid(), core.read_hit_t(), core.read_stats()


def assert_same(img, pid, ht, st, ref):
    assert np.array_equal(pid, ref["prim"])
    assert np.array_equal(ht.view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(img.view(np.uint32), ref["screen"].view(np.uint32))
    c = ref["counters"]
    assert (st.rays_primary, st.rays_shadow, st.rays_bounce, st.primary_hits) == (
        c["rays_primary"], c["rays_shadow"], c["rays_bounce"], c["primary_hits"])


@pytest.mark.parametrize("name,w,h,bounces", [
    ("triangles", 640, 480, 1),          # configuration 1 of the benchmark list
    ("triangles", 33, 17, 5),            # ragged size: partial workgroups on both edges
    ("plane", 320, 240, 1),
    ("plane", 200, 120, 3),
    ("cornell-box-cycles", 480, 270, 1),
    ("cornell-box-cycles", 240, 136, 5),  # the reference's MAX_BOUNCES
    ("cornell-box-cycles", 1, 1, 2),
])
def test_render_matches_oracle(core, built, name, w, h, bounces):
    b = built[name]
    load(core, b)
    img, pid, ht, st = render(core, w, h, seed=7, frames=0, max_bounces=bounces)
    assert st.scene_in_lds == 2  # these scenes are staged whole into LDS
    assert_same(img, pid, ht, st, b.oracle.render(w, h, seed=7, frames=0, max_bounces=bounces))


def test_golden_vectors_on_gpu(core, built):
    """the committed oracle vectors, so the GPU is also held to fixed data, not only to a live oracle"""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_renders.npz"))
    for name in ("triangles", "plane", "cornell-box-cycles"):
        load(core, built[name])
        for bounces in (1, 5):
            img, pid, ht, _ = render(core, 96, 72, seed=7, frames=0, max_bounces=bounces)
            key = f"{name}/b{bounces}"
            assert np.array_equal(pid, gold[key + "/prim"])
            assert np.array_equal(ht.view(np.uint32), gold[key + "/t"])
            assert np.array_equal(img.view(np.uint32), gold[key + "/rgba"])


def test_cornell_1080p_full_size(core, built):
    """configuration 2 at full size, every pixel compared"""
    b = built["cornell-box-cycles"]
    load(core, b)
    img, pid, ht, st = render(core, 1920, 1080, seed=12345, frames=0, max_bounces=1)
    ref = b.oracle.render(1920, 1080, seed=12345, frames=0, max_bounces=1)
    assert_same(img, pid, ht, st, ref)
    mse = float(np.mean((img[..., :3].astype(np.float64) - ref["screen"][..., :3]) ** 2))
    assert mse == 0.0  # PSNR is infinite
    # size-independent properties
    miss = pid == 0xFFFFFFFF
    assert np.all(ht[miss] == FMAX) and np.all(img[miss][:, :3] == 0) and np.all(img[..., 3][miss] == 1)
    assert np.isfinite(img).all() and pid[~miss].max() < 72
    assert 0.54 < (~miss).mean() < 0.56


def test_progressive_accumulation(core, built):
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h = 160, 90
    core.resize(w, h)
    screen = None
    for f in range(6):
        core.render(w, h, seed=100 + f, frames=f, max_bounces=2)
        ref = b.oracle.render(w, h, seed=100 + f, frames=f, max_bounces=2, screen=screen)
        screen = ref["screen"]
    core.sync()
    assert np.array_equal(core.read_framebuffer().view(np.uint32), screen.view(np.uint32))


def test_tile_list_and_scatter(core, built, acr):
    """tiles rendered out of order into a compact buffer, then scattered, equal the full frame;
    per-tile ray counts add up to the oracle's total"""
    import torch
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h, tw, th = 200, 120, 64, 16
    tx, ty = (w + tw - 1) // tw, (h + th - 1) // th
    ids = np.random.default_rng(1).permutation(tx * ty).astype(np.uint32)
    halves = [ids[: len(ids) // 3], ids[len(ids) // 3:]]
    frame = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    total_rays = 0
    try:
        for part in halves:
            tiles = torch.zeros((len(part), th * tw, 4), dtype=torch.float32, device="cuda")
            rays = torch.zeros(len(part), dtype=torch.int32, device="cuda")
            cyc = torch.zeros(len(part), dtype=torch.int32, device="cuda")
            core.render(w, h, seed=3, frames=0, max_bounces=2, tile_ids=part, tile_w=tw, tile_h=th,
                        screen_dev=C.c_void_p(tiles.data_ptr()), tile_rays_dev=C.c_void_p(rays.data_ptr()),
                        tile_cycles_dev=C.c_void_p(cyc.data_ptr()))
            core.scatter_tiles(C.c_void_p(tiles.data_ptr()), part, tw, th, w, h, C.c_void_p(frame.data_ptr()))
            core.sync()
            total_rays += int(rays.sum().item())
            assert int(cyc.min().item()) > 0
    finally:
        core.use_own_stream()
    ref = b.oracle.render(w, h, seed=3, frames=0, max_bounces=2)
    assert np.array_equal(frame.cpu().numpy().view(np.uint32), ref["screen"].view(np.uint32))
    c = ref["counters"]
    assert total_rays == c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]


def test_pixel_permutation(core, built):
    """DevParams::pixelValues: frame 0 resets the map to identity, later frames follow it"""
    import torch
    b = built["triangles"]
    load(core, b)
    w, h = 64, 48
    core.resize(w, h)
    pm = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    core.render(w, h, seed=9, frames=0, max_bounces=1, pixel_values_dev=C.c_void_p(pm.data_ptr()))
    core.sync()
    assert np.array_equal(pm.cpu().numpy(), np.arange(w * h, dtype=np.int32))
    perm = np.random.default_rng(0).permutation(w * h).astype(np.int32)
    pm.copy_(torch.from_numpy(perm))
    torch.cuda.synchronize()
    core.render(w, h, seed=4, frames=1, max_bounces=1, pixel_values_dev=C.c_void_p(pm.data_ptr()))
    core.sync()
    base = b.oracle.render(w, h, seed=9, frames=0, max_bounces=1)["screen"]
    ref = b.oracle.render(w, h, seed=4, frames=1, max_bounces=1, screen=base, pixel_map=perm)["screen"]
    assert np.array_equal(core.read_framebuffer().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("name", ["triangles", "plane", "cornell-box-cycles"])
def test_trace_rays_matches_oracle(core, built, name):
    """Scene::intersect for arbitrary rays, including directions with zero components (inf / NaN slab terms)"""
    b = built[name]
    load(core, b)
    rng = np.random.default_rng(5)
    n = 4096
    mn, mx = b.oracle.bounds()
    ctr, ext = (mn + mx) / 2, (mx - mn).max() + 1.0
    org = (ctr + rng.uniform(-1, 1, (n, 3)) * ext).astype(np.float32)
    tgt = (ctr + rng.uniform(-0.5, 0.5, (n, 3)) * (mx - mn)).astype(np.float32)
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    # axis-aligned and planar directions
    d[:300] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 300)] * rng.choice([-1, 1], (300, 1)).astype(np.float32)
    d[300:600, 1] = 0
    tmax = np.full(n, FMAX, np.float32)
    tmax[600:900] = rng.uniform(0.5, 6, 300).astype(np.float32)
    g = core.trace_rays(org, d, tmax)
    r = b.oracle.trace_rays(org, d, tmax)
    assert np.array_equal(g["hit"], r["hit"]) and np.array_equal(g["prim"], r["prim"])
    assert np.array_equal(g["t"].view(np.uint32), r["t"].view(np.uint32))
    h = r["prim"] != 0xFFFFFFFF
    assert np.array_equal(g["position"][h].view(np.uint32), r["position"][h].view(np.uint32))
    assert np.array_equal(g["normal"][h].view(np.uint32), r["normal"][h].view(np.uint32))
    assert h.sum() > 100


@pytest.mark.parametrize("ntri,depth,mode", [(150, 6, 2), (3000, 10, 1), (60000, 14, 0)])
def test_soup_deeper_bih_and_lds_modes(core, acr, orc, ntri, depth, mode):
    """procedural soup at larger BIH depths: scene fully in LDS / only the small tables / nothing staged"""
    b = Built(acr, orc, soup_data(acr, ntri, seed=4, edge=0.08 if ntri > 1000 else 0.3), mesh_depth=depth)
    load(core, b)
    w, h = 192, 108
    img, pid, ht, st = render(core, w, h, seed=11, frames=0, max_bounces=2)
    if mode != 0:
        assert st.scene_in_lds == mode
    assert_same(img, pid, ht, st, b.oracle.render(w, h, seed=11, frames=0, max_bounces=2))
    assert (pid != 0xFFFFFFFF).mean() > 0.01


def test_translucent_material_path(core, acr, orc, scene_data):
    """refractiveIndex != 0 (renderer.cu:286-311), unused by the shipped scenes: glass cubes in plane.blend"""
    import copy
    d = copy.deepcopy(scene_data["plane"])
    d.materials[d.meshes[0]["material"]]["refracti"] = 1.5
    b = Built(acr, orc, d)
    load(core, b)
    img, pid, ht, st = render(core, 160, 120, seed=2, frames=0, max_bounces=5)
    assert_same(img, pid, ht, st, b.oracle.render(160, 120, seed=2, frames=0, max_bounces=5))
    assert st.rays_bounce > 0


def test_reciprocal_shortcut_is_the_division_for_every_argument(core):
    """The kernels form 1 / direction, 1 / determinant and inversesqrt with the hardware reciprocal plus one Newton step
    where |x| lies in [2^-126, 2^126], and with the division elsewhere.  All 2^32 argument patterns against 1.0f / x
    on this GPU: no pattern may differ (this is what lets the frames stay bit-identical to the oracle, whose CPU
    division is correctly rounded)."""
    assert core.selftest_reciprocal() == 0


def test_primary_rays_parallel_to_a_coordinate_plane(core, built, acr, orc):
    """A primary ray through the middle column or row of an axis-aligned view has a direction component of exactly
    zero, hence an infinite 1 / d.  The interval form serves it when the launch can vouch that none of the bounds it
    meets on that axis equals the camera's coordinate there (Cornell: it can); otherwise, and always under
    six_plane_only, the pixel goes to the six-plane pass.  Frames equal the oracle's either way."""
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h = 1920, 1080
    served = 0
    for seed in range(100, 130):
        img, pid, ht, st = render(core, w, h, seed=seed, frames=0, max_bounces=1)
        if seed in (100, 109, 117, 121, 122, 125, 126, 129):
            assert_same(img, pid, ht, st, b.oracle.render(w, h, seed=seed, frames=0, max_bounces=1))
        served += st.six_plane_pixels == 0
    # Without this, 7 of these 30 frames list a pixel (seeds 109, 117, 121, 122, 125, 126, 129).  The camera stands in the
    # middle of the box, x = -2.78, where several mesh trees have a plane: rays of the middle column (d.x == 0) stay
    # with the six-plane pass, those of the middle row (d.z == 0) are served.
    assert served >= 26
    # a camera that sits exactly on a plane of the tree over objects: nothing is vouched for on that axis
    tri = built["triangles"]
    load(core, tri)
    for seed in range(20):
        img, pid, ht, st = render(core, 640, 480, seed=seed, frames=0, max_bounces=1)
        assert_same(img, pid, ht, st, tri.oracle.render(640, 480, seed=seed, frames=0, max_bounces=1))


def test_parallel_primary_rays_on_a_scene_read_from_hbm(core, acr, orc):
    """The same for a scene whose mesh tree (16 383 nodes, depth 14) is read from HBM and walked near side first: the
    plane values are kept sorted per axis, so the launch vouches for a tree of any size.  With an axis-aligned camera
    in front of the soup these seeds each send one primary ray with a direction component of exactly zero; they used
    to go to the six-plane pass (whole-tree walks in the reference's order) and are now served by the interval form.
    Frames, ids and distances equal the six-plane kernel's, and the oracle's."""
    b = Built(acr, orc, soup_data(acr, 60000, seed=9, edge=0.1, extent=1.0), mesh_depth=14)
    load(core, b)
    w, h = 1920, 1080
    for seed in (205, 209, 215, 301, 306):
        img, pid, ht, st = render(core, w, h, seed=seed, frames=0, max_bounces=1)
        assert st.near_first == 1 and st.six_plane_pixels == 0
        core.set_traversal(six_plane_only=True)
        try:
            img6, pid6, ht6, st6 = render(core, w, h, seed=seed, frames=0, max_bounces=1)
        finally:
            core.set_traversal(six_plane_only=False)
        assert st6.interval_traversal == 0
        assert np.array_equal(img.view(np.uint32), img6.view(np.uint32)) and np.array_equal(pid, pid6)
        assert np.array_equal(ht.view(np.uint32), ht6.view(np.uint32))
        if seed in (205, 306):
            assert_same(img, pid, ht, st, b.oracle.render(w, h, seed=seed, frames=0, max_bounces=1))


def test_scene_with_a_gigantic_triangle_keeps_the_division(core, acr, orc):
    """The interval-form kernels take the determinant's reciprocal without dividing, which is the quotient only up to
    |det| = 2^126; a scene is given to them only if |e1| |e2| < 2^120 for every triangle.  One triangle with edges of
    6e18 behind two hundred small ones: the frame must come from the six-plane kernels and equal the oracle's."""
    from conftest import duplicate_triangle_scene
    data = duplicate_triangle_scene(acr, 100, seed=11)
    m = data.meshes[0]
    big = np.array([[-3e18, -3e18, -2.0], [3e18, -3e18, -2.0], [0.0, 3e18, -2.0]], np.float32)
    m["positions"] = np.concatenate([m["positions"], big])
    m["normals"] = np.concatenate([m["normals"], np.tile(np.array([[0, 0, 1]], np.float32), (3, 1))])
    n0 = len(m["positions"]) - 3
    m["indices"] = np.concatenate([m["indices"], np.array([[n0, n0 + 1, n0 + 2]], np.uint32)])
    b = Built(acr, orc, data)
    load(core, b)
    ref = b.oracle.render(128, 96, seed=5, frames=0, max_bounces=2)
    img, pid, ht, st = render(core, 128, 96, seed=5, frames=0, max_bounces=2)
    assert st.interval_traversal == 0
    assert_same(img, pid, ht, st, ref)
    assert (pid != 0xFFFFFFFF).mean() > 0.9  # the big triangle fills the background


def test_empty_scene_and_no_lights(core, acr, orc, scene_data):
    import copy
    d = copy.deepcopy(scene_data["triangles"])
    d.lights = []
    b = Built(acr, orc, d)
    load(core, b)
    img, pid, ht, st = render(core, 64, 48, seed=1, frames=0, max_bounces=2)
    assert_same(img, pid, ht, st, b.oracle.render(64, 48, seed=1, frames=0, max_bounces=2))
    assert st.rays_shadow == 0
    for n in d.nodes:
        n["meshes"] = []
    b2 = Built(acr, orc, d)
    load(core, b2)
    img, pid, ht, st = render(core, 64, 48, seed=1, frames=0, max_bounces=2)
    assert np.all(pid == 0xFFFFFFFF) and np.all(img[..., :3] == 0) and np.all(img[..., 3] == 1)


def test_error_paths(acr, built):
    c = acr.HipCore(0)
    try:
        with pytest.raises(RuntimeError, match="before scene"):
            c.render(16, 16)
        load(c, built["triangles"])
        with pytest.raises(RuntimeError, match="resize"):
            c.render(16, 16)  # no framebuffer yet
        c.resize(16, 16)
        with pytest.raises(RuntimeError, match="max_bounces"):
            c.render(16, 16, max_bounces=17)
        with pytest.raises(RuntimeError, match="multiples of 16"):
            c.render(16, 16, tile_ids=np.array([0], np.uint32), tile_w=8, tile_h=8, screen_dev=C.c_void_p(c.framebuffer_ptr()))
        with pytest.raises(RuntimeError, match="out of range"):
            c.render(16, 16, tile_ids=np.array([5], np.uint32), tile_w=16, tile_h=16, screen_dev=C.c_void_p(c.framebuffer_ptr()))
        # a corrupted BIH (internal node on the last level) is rejected at upload, not discovered by the kernel
        s = built["triangles"].scene.soa
        from cudaraytracer_amd import _capi
        bad = _capi.SceneSoA()
        C.memmove(C.byref(bad), C.byref(s), C.sizeof(s))
        flags = np.zeros((1 << s.scene_depth) - 1, np.uint8)
        bad.snode_flags = flags.ctypes.data_as(_capi.c_u8_p)
        with pytest.raises(RuntimeError, match="last level"):
            c.upload_scene(bad)
        c.render(16, 16, max_bounces=1)  # the previous, valid scene is still loaded
        c.sync()
    finally:
        c.close()
    with pytest.raises(RuntimeError, match="no usable GPU"):
        acr.HipCore(9999)


def test_spot_and_directional_lights(core, acr, orc, scene_data):
    """DIRECTIONAL and SPOT lights both go through spotLightAccum (scene.cu:234-271,282-285), which calls acos.
    No shipped scene has such lights.  acosf comes from glibc on the oracle side and from the device math
    library on the GPU, so radiance is held to the 1e-4 relative tolerance instead of bit equality;
    geometry (primitive ids, hit distances) stays exact.

    Where the tolerance is exceeded is checked too, on a frame without bounces (so that a pixel's only shaded
    point is the primary hit the oracle reports): lightTheta enters the result through the cone test
    `|lightTheta| >= outer / 2` (scene.cu:249) and through falloff = (outer / 2 - |lightTheta|) / (falloff_distance / 2)
    (scene.cu:257-261), whose relative change for a last-place difference in lightTheta is ulp / (outer / 2 - |lightTheta|):
    above 1e-4 only within about a milliradian inside the cone's edge, or exactly on it.  Every pixel out of
    tolerance must lie there for one of the two lights."""
    import copy
    d = copy.deepcopy(scene_data["cornell-box-cycles"])
    d.lights[0].update(type=2, direction=[0.0, 0.0, -1.0], cones=[2.0, 2.5])   # spot: half-angle 1.0 rad, soft edge
    d.lights[1].update(type=1, direction=[0.3, 0.2, -1.0], cones=[2.4, 3.0])   # "directional": same formula
    b = Built(acr, orc, d)
    load(core, b)
    w, h = 320, 180
    img, pid, ht, st = render(core, w, h, seed=5, frames=0, max_bounces=3)
    ref = b.oracle.render(w, h, seed=5, frames=0, max_bounces=3)
    assert np.array_equal(pid, ref["prim"]) and np.array_equal(ht.view(np.uint32), ref["t"].view(np.uint32))
    a, r = img[..., :3].astype(np.float64), ref["screen"][..., :3].astype(np.float64)
    rel = np.abs(a - r) / np.maximum(np.abs(r), 1e-3)
    # a one-ulp acos difference can flip a cone-edge test for an isolated pixel; allow a handful
    assert (rel > 1e-4).mean() < 1e-4
    assert np.array_equal(img[..., 3], ref["screen"][..., 3])
    lit = r.sum(axis=2) > 0
    assert 0.05 < lit.mean() < 0.9  # the cone lights part of the box, not all of it

    # without bounces, at a size where the cone edges cross many pixels: the set of pixels out of tolerance
    w, h = 960, 540
    f32 = np.float32
    total_bad = 0
    for seed in (5, 6):
        img, pid, ht, st = render(core, w, h, seed=seed, frames=0, max_bounces=1)
        ref = b.oracle.render(w, h, seed=seed, frames=0, max_bounces=1)
        assert np.array_equal(pid, ref["prim"]) and np.array_equal(ht.view(np.uint32), ref["t"].view(np.uint32))
        a, r = img[..., :3].astype(np.float64), ref["screen"][..., :3].astype(np.float64)
        rel = (np.abs(a - r) / np.maximum(np.abs(r), 1e-3)).max(axis=2)
        bad = rel > 1e-4
        total_bad += int(bad.sum())
        pos = ref["position"].reshape(h, w, 3)
        margin = np.full((h, w), np.inf)
        for k in range(b.oracle.num_lights):
            l = b.oracle.light(k)
            assert l["type"] in (1, 2)
            # lightTheta as scene.cu:236-246 forms it, in binary32, operation for operation
            dv = (l["position"][None, None, :] - pos).astype(f32)
            inv = (f32(1) / np.sqrt((dv[..., 0] * dv[..., 0] + dv[..., 1] * dv[..., 1] + dv[..., 2] * dv[..., 2]).astype(f32))).astype(f32)
            dv = (dv * inv[..., None]).astype(f32)
            ld = l["direction"].astype(f32)
            ld = (ld * (f32(1) / np.sqrt(f32(ld[0] * ld[0] + ld[1] * ld[1]) + f32(ld[2] * ld[2])))).astype(f32)
            cos_light = ((-dv[..., 0]) * ld[0] + (-dv[..., 1]) * ld[1] + (-dv[..., 2]) * ld[2]).astype(f32)
            theta = np.abs(np.arccos(np.clip(cos_light, -1, 1).astype(f32)).astype(np.float64))
            outer_half = float(f32(l["cones"][0]) / f32(2))  # `outer` is innerConeAngle (scene.cu:245)
            # inside the edge by less than two milliradians, or outside it by no more than a few last places of theta
            inside = outer_half - theta
            near_edge = (inside < 2e-3) & (inside > -8 * np.spacing(f32(outer_half)))
            margin = np.where(near_edge, 0.0, margin)
        off_edge = bad & (margin != 0.0) & (pid != 0xFFFFFFFF)
        assert not off_edge.any(), "pixels out of tolerance away from every cone edge: %s" % np.argwhere(off_edge)[:8].tolist()
        assert not (bad & (pid == 0xFFFFFFFF)).any()
        assert bad.mean() < 1e-3
    # (informative: how many pixels the two acosf implementations actually disagree on)
    print("spot / directional lights: %d of %d pixels beyond 1e-4, all within 2 mrad of a cone edge" % (total_bad, 2 * w * h))


def test_interval_and_six_plane_traversal_agree(core, built):
    """the interval form of the slab test (default) and the six-plane form as written in geometry.h:44-76
    must give the same frame bit for bit; both equal the oracle"""
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h = 384, 216
    ref = b.oracle.render(w, h, seed=21, frames=0, max_bounces=4)
    try:
        core.set_traversal(six_plane_only=True)
        img6, pid6, ht6, st6 = render(core, w, h, seed=21, frames=0, max_bounces=4)
        assert st6.interval_traversal == 0
        assert_same(img6, pid6, ht6, st6, ref)
    finally:
        core.set_traversal(six_plane_only=False)
    img, pid, ht, st = render(core, w, h, seed=21, frames=0, max_bounces=4)
    assert st.interval_traversal == 1 and st.six_plane_pixels == 0
    assert_same(img, pid, ht, st, ref)


def test_pixels_handed_to_the_six_plane_pass(core, acr, orc, scene_data):
    """An object squashed flat (scale 0 along one axis) has a singular transform: rays that reach it get
    non-finite local rays, which the interval traversal must not touch.  Those pixels go through the list
    pass; the frame still equals the oracle's, pixel for pixel, including tiles and ray counts."""
    import copy
    d = copy.deepcopy(scene_data["cornell-box-cycles"])
    names = [n["name"] for n in d.nodes]
    k = names.index("short_box")
    m = np.array(d.nodes[k]["transform"], dtype=np.float32).reshape(4, 4)  # column-major: m[col][row]
    m[2, :3] = 0.0  # third column: the image of the local z axis
    d.nodes[k]["transform"] = [float(v) for v in m.reshape(-1)]
    b = Built(acr, orc, d)
    load(core, b)
    w, h = 320, 180
    ref = b.oracle.render(w, h, seed=3, frames=0, max_bounces=3)
    img, pid, ht, st = render(core, w, h, seed=3, frames=0, max_bounces=3)
    assert st.interval_traversal == 1
    assert 0 < st.six_plane_pixels <= w * h
    assert_same(img, pid, ht, st, ref)
    # second frame on the same context: the alternate counter is in use and starts from zero
    img2, pid2, ht2, st2 = render(core, w, h, seed=3, frames=0, max_bounces=3)
    assert st2.six_plane_pixels == st.six_plane_pixels
    assert np.array_equal(img2.view(np.uint32), img.view(np.uint32))
    # tile mode with per-tile ray counts
    import torch
    tw, th = 64, 16
    tx, ty = (w + tw - 1) // tw, (h + th - 1) // th
    ids = np.arange(tx * ty, dtype=np.uint32)[::-1].copy()
    tiles = torch.zeros((len(ids), th * tw, 4), dtype=torch.float32, device="cuda")
    rays = torch.zeros(len(ids), dtype=torch.int32, device="cuda")
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        core.render(w, h, seed=3, frames=0, max_bounces=3, tile_ids=ids, tile_w=tw, tile_h=th,
                    screen_dev=C.c_void_p(tiles.data_ptr()), tile_rays_dev=C.c_void_p(rays.data_ptr()))
        core.sync()
    finally:
        core.use_own_stream()
    c = ref["counters"]
    assert int(rays.sum().item()) == c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]
    t = tiles.cpu().numpy().reshape(len(ids), th, tw, 4)
    for n, tid in enumerate(ids):
        x0, y0 = (int(tid) % tx) * tw, (int(tid) // tx) * th
        ww, hh = min(tw, w - x0), min(th, h - y0)
        assert np.array_equal(t[n, :hh, :ww].view(np.uint32), ref["screen"][y0:y0 + hh, x0:x0 + ww].view(np.uint32))


def test_flythrough_cameras(core, built):
    """frames of the 120-frame camera flythrough (every frame renders with frames = 0, as after moveCamera)"""
    from cudaraytracer_amd import camera_path
    b = built["cornell-box-cycles"]
    load(core, b)
    cams = camera_path.flythrough(b.scene.camera, 120)
    assert len(cams) == 120
    w, h = 240, 135
    try:
        for k in (0, 37, 119):
            core.set_camera(cams[k])
            img, pid, ht, st = render(core, w, h, seed=500 + k, frames=0, max_bounces=2)
            oc = orc_camera(cams[k])
            assert_same(img, pid, ht, st, b.oracle.render(w, h, seed=500 + k, frames=0, max_bounces=2, camera=oc))
    finally:
        core.set_camera(b.scene.camera)


def orc_camera(cam):
    import oracle
    oc = oracle.OrcCamera()
    C.memmove(C.byref(oc), C.byref(cam), C.sizeof(oc))
    return oc


def test_soup_10m_triangles_full_size(core, acr, orc):
    """The deep-BIH workload at its full size: 10 M triangles, mesh BIH depth 22, 1920x1080.  The oracle is
    compared on 128 rows spread over the frame (every pixel of them); the whole frame is held to properties that do not
    depend on size: both forms of the slab test give identical frames, a tile list reproduces the same
    pixels, misses are exactly (0,0,0,1), ids and distances are in range, ray counts are consistent."""
    w, h, depth = 1920, 1080, 22
    data = soup_data(acr, 10_000_000, seed=1, edge=0.1, extent=10.0)
    view, keep = data.pack()
    scene = acr.Scene(import_view=view, mesh_bih_depth=depth)
    core.upload_scene(scene.soa)
    core.set_camera(scene.camera)
    img, pid, ht, st = render(core, w, h, seed=9, frames=0, max_bounces=1)
    assert st.scene_in_lds == 1 and st.interval_traversal == 1
    miss = pid == 0xFFFFFFFF
    assert 0.2 < (~miss).mean() < 0.9
    assert np.all(ht[miss] == FMAX) and np.all(img[miss][:, :3] == 0) and np.all(img[miss][:, 3] == 1)
    assert np.isin(img[~miss][:, 3], (1, 2)).all()  # absorbed at the hit, or continued into rayColor<0> (renderer.cu:282,312)
    assert pid[~miss].max() < 10_000_000 and np.isfinite(ht[~miss]).all() and (ht[~miss] > 0).all()
    assert np.isfinite(img).all() and (img[..., :3] >= 0).all()
    assert st.rays_primary == w * h and st.primary_hits == int((~miss).sum()) and 0 < st.rays_shadow <= st.primary_hits
    # the slab test as written in the reference, for every ray
    try:
        core.set_traversal(six_plane_only=True)
        img6, pid6, ht6, st6 = render(core, w, h, seed=9, frames=0, max_bounces=1)
    finally:
        core.set_traversal(six_plane_only=False)
    assert st6.interval_traversal == 0
    assert np.array_equal(pid6, pid) and np.array_equal(ht6.view(np.uint32), ht.view(np.uint32))
    assert np.array_equal(img6.view(np.uint32), img.view(np.uint32))
    assert (st6.rays_shadow, st6.primary_hits) == (st.rays_shadow, st.primary_hits)
    # a band of rows against the oracle
    # 16 bands of 8 rows spread over the frame (128 rows, every pixel of them) against the oracle
    o = orc.OracleScene(view, 6, depth)
    rows = 0
    for k in range(16):
        r0 = 4 + 67 * k
        r1 = r0 + 8
        ref = o.render(w, h, seed=9, frames=0, max_bounces=1, row_begin=r0, row_end=r1)
        assert np.array_equal(pid[r0:r1], ref["prim"][r0:r1]), "rows %d..%d" % (r0, r1)
        assert np.array_equal(ht[r0:r1].view(np.uint32), ref["t"][r0:r1].view(np.uint32)), "rows %d..%d" % (r0, r1)
        assert np.array_equal(img[r0:r1].view(np.uint32), ref["screen"][r0:r1].view(np.uint32)), "rows %d..%d" % (r0, r1)
        rows += r1 - r0
    assert rows == 128 and r1 <= h
    o.close()
    # tiles out of order give the same pixels
    import torch
    tw, th = 64, 16
    tx = (w + tw - 1) // tw
    ids = np.array([5 * tx + 3, 40 * tx + 29, 2, 67 * tx + 1], dtype=np.uint32)  # the last one is cut by the frame's bottom edge
    tiles = torch.zeros((len(ids), th * tw, 4), dtype=torch.float32, device="cuda")
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        core.render(w, h, seed=9, frames=0, max_bounces=1, tile_ids=ids, tile_w=tw, tile_h=th, screen_dev=C.c_void_p(tiles.data_ptr()))
        core.sync()
    finally:
        core.use_own_stream()
    t = tiles.cpu().numpy().reshape(len(ids), th, tw, 4)
    for n, tid in enumerate(ids):
        x0, y0 = (int(tid) % tx) * tw, (int(tid) // tx) * th
        ww, hh = min(tw, w - x0), min(th, h - y0)
        assert np.array_equal(t[n, :hh, :ww].view(np.uint32), img[y0:y0 + hh, x0:x0 + ww].view(np.uint32))


def test_traversal_path_keys_and_sorted_pixel_map(core, built):
    """The reference's adaptive assignment, as designed (bih.h:13-27,268-274; renderer.cu:131-137,343-355,457-461):
    frame 0 records a 64-bit traversal-path key per pixel, the pixel numbers are stably sorted by key, later
    frames run through the sorted table.  Keys, table and the frames rendered through it all equal the oracle's."""
    import torch
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h = 320, 180
    n = w * h
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    table = torch.zeros(n, dtype=torch.int32, device="cuda")
    core.resize(w, h)
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for six_plane in (True, False):
            core.set_traversal(six_plane_only=six_plane)
            keys.zero_()
            core.render(w, h, seed=40, frames=0, max_bounces=3, path_keys_dev=C.c_void_p(keys.data_ptr()),
                        pixel_values_dev=C.c_void_p(table.data_ptr()))
            core.sync()
            ref0 = b.oracle.render(w, h, seed=40, frames=0, max_bounces=3)
            k = keys.cpu().numpy().view(np.uint64)
            assert np.array_equal(k, ref0["path_key"].reshape(-1))
            assert np.array_equal(core.read_framebuffer().view(np.uint32), ref0["screen"].view(np.uint32))
            assert np.array_equal(table.cpu().numpy(), np.arange(n, dtype=np.int32))  # frames == 0 resets the table (renderer.cu:344-348)
        assert len(np.unique(k)) > 8  # the keys do tell pixels apart
        core.sort_pixel_map(C.c_void_p(keys.data_ptr()), C.c_void_p(table.data_ptr()), n)
        core.sync()
        perm = table.cpu().numpy()
        assert np.array_equal(perm, np.argsort(k, kind="stable").astype(np.int32))
        # frames 1..3 through the sorted table
        screen = ref0["screen"]
        for f in (1, 2, 3):
            core.render(w, h, seed=40 + f, frames=f, max_bounces=3, pixel_values_dev=C.c_void_p(table.data_ptr()))
            screen = b.oracle.render(w, h, seed=40 + f, frames=f, max_bounces=3, screen=screen, pixel_map=perm)["screen"]
        core.sync()
        assert np.array_equal(core.read_framebuffer().view(np.uint32), screen.view(np.uint32))
        assert np.array_equal(table.cpu().numpy(), perm)  # frames > 0 leave the table alone
    finally:
        core.set_traversal(six_plane_only=False)
        core.use_own_stream()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_instanced_scenes(core, acr, orc, seed):
    """40 objects instancing a few meshes: nested parents, uniform and non-uniform scales (the conservative world-box
    skip is off for the latter), one or two meshes per object, mirror-like and translucent materials, three lights,
    scene-tree leaves that hold several objects.  Both forms of the slab test, 4 bounces."""
    b = Built(acr, orc, random_instanced_scene(acr, seed))
    load(core, b)
    w, h = 256, 144
    ref = b.oracle.render(w, h, seed=seed, frames=0, max_bounces=4)
    assert ref["counters"]["primary_hits"] > 0.1 * w * h and ref["counters"]["rays_bounce"] > 0
    try:
        for six_plane in (False, True):
            core.set_traversal(six_plane_only=six_plane)
            img, pid, ht, st = render(core, w, h, seed=seed, frames=0, max_bounces=4)
            assert_same(img, pid, ht, st, ref)
    finally:
        core.set_traversal(six_plane_only=False)
    # raw Scene::intersect queries from inside the cloud, in every direction
    rng = np.random.default_rng(seed)
    o = rng.uniform(-4, 4, (2000, 3)).astype(np.float32)
    d = rng.normal(size=(2000, 3)).astype(np.float32)
    d[::7, 0] = 0.0  # some axis-parallel components
    got, want = core.trace_rays(o, d), b.oracle.trace_rays(o, d)
    for k in ("hit", "prim"):
        assert np.array_equal(got[k], want[k])
    for k in ("t", "position", "normal"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32))


def test_random_instanced_scene_modes(core, acr, orc):
    """the same kind of scene through the other launch shapes: accumulation, traversal-path keys, a tile list"""
    import torch
    b = Built(acr, orc, random_instanced_scene(acr, 4))
    load(core, b)
    w, h = 208, 112
    n = w * h
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    core.resize(w, h)
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        screen = None
        for f in range(3):
            core.render(w, h, seed=90 + f, frames=f, max_bounces=3, path_keys_dev=C.c_void_p(keys.data_ptr()) if f == 0 else None)
            ref = b.oracle.render(w, h, seed=90 + f, frames=f, max_bounces=3, screen=screen)
            screen = ref["screen"]
            if f == 0:
                core.sync()
                assert np.array_equal(keys.cpu().numpy().view(np.uint64), ref["path_key"].reshape(-1))
        core.sync()
        assert np.array_equal(core.read_framebuffer().view(np.uint32), screen.view(np.uint32))
        tw, th = 64, 16
        tx, ty = (w + tw - 1) // tw, (h + th - 1) // th
        ids = np.random.default_rng(0).permutation(tx * ty).astype(np.uint32)
        tiles = torch.zeros((len(ids), th * tw, 4), dtype=torch.float32, device="cuda")
        rays = torch.zeros(len(ids), dtype=torch.int32, device="cuda")
        core.render(w, h, seed=90, frames=0, max_bounces=3, tile_ids=ids, tile_w=tw, tile_h=th,
                    screen_dev=C.c_void_p(tiles.data_ptr()), tile_rays_dev=C.c_void_p(rays.data_ptr()))
        core.sync()
    finally:
        core.use_own_stream()
    ref = b.oracle.render(w, h, seed=90, frames=0, max_bounces=3)
    c = ref["counters"]
    assert int(rays.sum().item()) == c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]
    t = tiles.cpu().numpy().reshape(len(ids), th, tw, 4)
    for k, tid in enumerate(ids):
        x0, y0 = (int(tid) % tx) * tw, (int(tid) // tx) * th
        ww, hh = min(tw, w - x0), min(th, h - y0)
        assert np.array_equal(t[k, :hh, :ww].view(np.uint32), ref["screen"][y0:y0 + hh, x0:x0 + ww].view(np.uint32))


@pytest.mark.parametrize("ntri,depth,extent,edge", [(60000, 14, 1.0, 0.1), (400000, 17, 3.0, 0.15)])
def test_near_side_first_mesh_order(core, acr, orc, ntri, depth, extent, edge):
    """Trees that are not LDS-resident are walked near side first; frames and raw ray queries equal the oracle's
    (which walks them in the reference's order) and the six-plane kernel's, and no pixel needed the
    reference-order pass."""
    b = Built(acr, orc, soup_data(acr, ntri, seed=9, edge=edge, extent=extent), mesh_depth=depth)
    load(core, b)
    w, h = 256, 144
    ref = b.oracle.render(w, h, seed=21, frames=0, max_bounces=3)
    img, pid, ht, st = render(core, w, h, seed=21, frames=0, max_bounces=3)
    assert st.near_first == 1 and st.interval_traversal == 1 and st.scene_in_lds != 2
    assert_same(img, pid, ht, st, ref)
    assert (pid != 0xFFFFFFFF).mean() > 0.1
    core.set_traversal(six_plane_only=True)
    try:
        img6, pid6, ht6, st6 = render(core, w, h, seed=21, frames=0, max_bounces=3)
    finally:
        core.set_traversal(six_plane_only=False)
    assert st6.near_first == 0 and st6.interval_traversal == 0
    assert np.array_equal(img6.view(np.uint32), img.view(np.uint32)) and np.array_equal(pid6, pid)
    rng = np.random.default_rng(3)
    n = 20000
    org = rng.uniform(-1.5 * extent, 1.5 * extent, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    g, r = core.trace_rays(org, d), b.oracle.trace_rays(org, d)
    assert np.array_equal(g["hit"], r["hit"]) and np.array_equal(g["prim"], r["prim"])
    assert np.array_equal(g["t"].view(np.uint32), r["t"].view(np.uint32))
    assert r["hit"].sum() > n // 20


def test_near_side_first_exact_ties(core, acr, orc):
    """Every hit is an exact tie between two copies of a triangle (30 000 triangles, each twice, depth-14 tree
    in HBM): the copy the reference tests first must win, whatever the visiting order."""
    b = Built(acr, orc, duplicate_triangle_scene(acr, 30000, spread=0.05), mesh_depth=14)
    load(core, b)
    w, h = 256, 256
    ref = b.oracle.render(w, h, seed=4, frames=0, max_bounces=2)
    img, pid, ht, st = render(core, w, h, seed=4, frames=0, max_bounces=2)
    assert st.near_first == 1
    assert_same(img, pid, ht, st, ref)
    assert (pid != 0xFFFFFFFF).mean() > 0.3
    # the winners are spread over both halves of the item array: position in the tree's item order decides, not the copy
    assert 0.2 < (pid[pid != 0xFFFFFFFF] < 30000).mean() < 0.8


def test_trees_whose_leaves_do_not_ascend_get_the_six_plane_kernel(core, acr, orc):
    """The near-side-first walk breaks exact ties by item position, which needs the leaves' item ranges to ascend in
    left-first order.  A tree without that property (two sibling leaves' ranges swapped by hand) must be rendered by
    the six-plane kernel, in the order as written."""
    b = Built(acr, orc, soup_data(acr, 3000, seed=4, edge=0.08), mesh_depth=10)
    s = b.scene.soa
    n = (1 << int(s.mesh_depth[0])) - 1
    nb = int(s.mesh_node_begin[0])
    flags = np.ctypeslib.as_array(s.mnode_flags, (s.num_mesh_nodes,))[nb:nb + n]
    start = np.ctypeslib.as_array(s.mnode_start, (s.num_mesh_nodes,))[nb:nb + n]
    end = np.ctypeslib.as_array(s.mnode_end, (s.num_mesh_nodes,))[nb:nb + n]
    load(core, b)
    _, _, _, st = render(core, 64, 64, seed=1, frames=0, max_bounces=1)
    assert st.near_first == 1
    # two sibling leaves (children 2i+1, 2i+2 of a reachable internal node), both non-empty
    pair = None
    todo = [0]
    while todo and pair is None:
        i = todo.pop()
        if flags[i] & 4:  # leaf
            continue
        l, r = 2 * i + 1, 2 * i + 2
        if (flags[l] & 4) and (flags[r] & 4) and start[l] < end[l] and start[r] < end[r]:
            pair = (l, r)
        todo += [r, l]
    assert pair is not None
    l, r = pair
    keep = (int(start[l]), int(end[l]), int(start[r]), int(end[r]))
    try:
        start[l], end[l], start[r], end[r] = keep[2], keep[3], keep[0], keep[1]
        core.upload_scene(s)
        img, pid, ht, st = render(core, 64, 64, seed=1, frames=0, max_bounces=1)
        assert st.near_first == 0 and st.interval_traversal == 0
    finally:
        start[l], end[l], start[r], end[r] = keep
        core.upload_scene(s)


def test_block_launch_order_does_not_change_frames(core, built, acr):
    """The context starts the blocks of a launch most expensive first once it has measured them; a caller's table
    can do the same.  Frames, ids and ray counts are those of the natural order (and the oracle's)."""
    import torch
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h = 640, 480   # 1200 blocks of 16 x 16: above the size from which the order is used
    ref = b.oracle.render(w, h, seed=5, frames=0, max_bounces=3)
    core.set_block_order(0)
    try:
        img0, pid0, ht0, st0 = render(core, w, h, seed=5, frames=0, max_bounces=3)
        assert st0.block_order == 0
        assert_same(img0, pid0, ht0, st0, ref)
    finally:
        core.set_block_order(4)
    seen = []
    for k in range(14):  # launch 1 (the second of this configuration) measures, 2.. use the order; measured again 4, then 8 launches later
        img, pid, ht, st = render(core, w, h, seed=5, frames=0, max_bounces=3)
        seen.append(st.block_order)
        assert_same(img, pid, ht, st, ref)
    assert seen == [0, 0] + [1] * 12
    # a caller's table: reverse order
    nb = ((w + 15) // 16) * ((h + 15) // 16)
    order = torch.arange(nb - 1, -1, -1, dtype=torch.int32, device="cuda:0")
    core.resize(w, h)
    core.render(w, h, seed=5, frames=0, max_bounces=3, collect_stats=True, block_order_dev=C.c_void_p(order.data_ptr()))
    core.sync()
    st = core.read_stats()
    assert st.block_order == 2
    assert_same(core.read_framebuffer(), core.read_primid(), core.read_hit_t(), st, ref)
    # accumulation over frames under the measured order
    acc = b.oracle.render(w, h, seed=7, frames=0, max_bounces=2)["screen"]
    core.render(w, h, seed=7, frames=0, max_bounces=2)
    for f in (1, 2):
        acc = b.oracle.render(w, h, seed=7 + f, frames=f, max_bounces=2, screen=acc)["screen"]
        core.render(w, h, seed=7 + f, frames=f, max_bounces=2)
    core.sync()
    assert np.array_equal(core.read_framebuffer().view(np.uint32), acc.view(np.uint32))
    # a table cannot be combined with a tile list
    with pytest.raises(Exception):
        core.render(w, h, seed=5, frames=0, max_bounces=1, tile_ids=np.arange(4, dtype=np.uint32), tile_w=64, tile_h=16,
                    block_order_dev=C.c_void_p(order.data_ptr()))
    core.set_block_order(64)


def test_block_launch_order_of_a_scene_read_through_the_l2(core, acr, orc):
    """A scene whose trees are not LDS-resident gets the XCD-aware order table (runs of the blocks' Z-curve of
    equal measured cost, one per XCD, most expensive first inside, overflow into the positions other runs leave
    free).  The table must hold every block exactly once whatever the costs: frames of a size with cut-off edge
    blocks, a new seed on every launch so that a block left out would show, each equal to the oracle's; then
    another size on the same context (the Z-curve is rebuilt)."""
    b = Built(acr, orc, soup_data(acr, 60000, seed=9, edge=0.1, extent=1.0), mesh_depth=14)
    load(core, b)
    core.set_block_order(4)
    try:
        for (w, h), launches in (((650, 470), 8), ((1000, 330), 4)):   # 41 x 30 and 63 x 21 blocks of 16 x 16
            seen = []
            for k in range(launches):
                ref = b.oracle.render(w, h, seed=30 + k, frames=0, max_bounces=1)
                img, pid, ht, st = render(core, w, h, seed=30 + k, frames=0, max_bounces=1)
                assert st.scene_in_lds != 2
                seen.append(st.block_order)
                assert_same(img, pid, ht, st, ref)
            assert seen == [0, 0] + [1] * (launches - 2)
        # a list of full-width bands (the multi-GPU split) gets the same kind of table: 20 bands of 64 blocks
        import torch
        w, h = 1024, 320
        ids = np.arange(h // 16, dtype=np.uint32)
        tiles = torch.zeros((len(ids), 16 * w, 4), dtype=torch.float32, device="cuda")
        core.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            for k in range(5):
                core.render(w, h, seed=50 + k, frames=0, max_bounces=1, tile_ids=ids, tile_w=w, tile_h=16,
                            screen_dev=C.c_void_p(tiles.data_ptr()), collect_stats=True)
                core.sync()
                assert core.read_stats().block_order == (1 if k >= 2 else 0)
                ref = b.oracle.render(w, h, seed=50 + k, frames=0, max_bounces=1, aov=False)
                assert np.array_equal(tiles.cpu().numpy().reshape(h, w, 4).view(np.uint32), ref["screen"].view(np.uint32))
        finally:
            core.use_own_stream()
    finally:
        core.set_block_order(64)


def test_accumulating_ray_counters(core, built):
    """collect_stats = 2 keeps what earlier launches counted: three frames, one read, the oracle's three totals."""
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h = 320, 180
    core.resize(w, h)
    want = [0, 0, 0, 0]
    for k, seed in enumerate((3, 4, 5)):
        core.render(w, h, seed=seed, frames=0, max_bounces=2, collect_stats=1 if k == 0 else 2)
        c = b.oracle.render(w, h, seed=seed, frames=0, max_bounces=2, aov=False)["counters"]
        for i, n in enumerate(("rays_primary", "rays_shadow", "rays_bounce", "primary_hits")):
            want[i] += c[n]
    st = core.read_stats()
    assert [st.rays_primary, st.rays_shadow, st.rays_bounce, st.primary_hits] == want
    core.render(w, h, seed=5, frames=0, max_bounces=2, collect_stats=1)   # 1 starts from zero again
    st = core.read_stats()
    assert st.rays_primary == w * h


def test_band_renderer_on_one_gpu(core, built, acr):
    """DistributedRenderer over the HIP core without a process group (one rank): frames rendered band by band into
    frame-shaped memory equal the full-frame render; per-band costs come back as clock ticks or as ray counts."""
    import torch
    from cudaraytracer_amd import distributed as D
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h = 640, 360
    dev = torch.device("cuda", 0)
    seeds = iter(range(100, 200))
    state = {}

    def frame_args():
        state["seed"] = next(seeds)
        return dict(seed=state["seed"], frames=0, max_bounces=2)

    bands = D.hip_render_tiles(core, w, h, w, 16, frame_args, cost_kind="cycles")
    dr = D.DistributedRenderer(w, h, bands, dev, tile_w=w, tile_h=16, rebalance_every=2)
    assert dr.bands and dr.world == 1
    for k in range(4):
        frame = dr.render_frame()
        dr.finish()
        core.sync()
        ref = b.oracle.render(w, h, seed=state["seed"], frames=0, max_bounces=2, aov=False)
        assert np.array_equal(frame.cpu().numpy().view(np.uint32), ref["screen"].view(np.uint32))
    ticks = dr.last_cost.cpu().numpy()
    bands.cost_kind = "rays"
    dr.rebalance_every = 0
    dr.render_frame(want_cost=True)
    core.sync()
    rays = int(dr.rays_last_frame().item())
    c = b.oracle.render(w, h, seed=state["seed"], frames=0, max_bounces=2, aov=False)["counters"]
    assert rays == c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]
    assert calibrate_is_noop(dr)
    # the camera stands still: the same renderer averaging frame k into the running mean of its bands (frames = k)
    acc = D.DistributedRenderer(w, h, bands, dev, tile_w=w, tile_h=16, rebalance_every=2, accumulate=True)
    bands.cost_kind = "cycles"
    screen, used = None, []
    for k in range(5):
        acc.render_frame()
        used.append(state["seed"])
        screen = b.oracle.render(w, h, seed=state["seed"], frames=k, max_bounces=2, aov=False, screen=screen)["screen"]
    frame = acc.finish()
    core.sync()
    assert np.array_equal(frame.cpu().numpy().view(np.uint32), screen.view(np.uint32))
    acc.camera_moved()
    acc.render_frame()
    core.sync()
    ref = b.oracle.render(w, h, seed=state["seed"], frames=0, max_bounces=2, aov=False)["screen"]
    assert np.array_equal(acc.finish().cpu().numpy().view(np.uint32), ref.view(np.uint32))
    core.use_own_stream()  # hip_render_tiles moved the shared context onto torch's stream


def calibrate_is_noop(dr):
    return dr.calibrate() is None


@pytest.mark.parametrize("ntri,depth", [(60000, 14), (150, 6)])
def test_whole_frames_through_the_list_pass(core, acr, orc, ntri, depth):
    """The pass that normally renders only the listed pixels (reference visiting order in the interval form, then
    the six-plane form) renders whole frames identical to the main kernel's and the oracle's: for a scene whose
    trees the main kernel walks near side first, and for an LDS-resident one."""
    b = Built(acr, orc, soup_data(acr, ntri, seed=6, edge=0.1 if ntri > 1000 else 0.3), mesh_depth=depth)
    load(core, b)
    w, h = 160, 96
    ref = b.oracle.render(w, h, seed=8, frames=0, max_bounces=3)
    img, pid, ht, st = render(core, w, h, seed=8, frames=0, max_bounces=3)
    assert_same(img, pid, ht, st, ref)
    core.set_traversal(list_pass=True)
    try:
        img2, pid2, ht2, st2 = render(core, w, h, seed=8, frames=0, max_bounces=3)
    finally:
        core.set_traversal()
    assert st2.interval_traversal == 1 and st2.near_first == 0
    assert st2.six_plane_pixels >= w * h   # every slot of the launch was listed
    assert_same(img2, pid2, ht2, st2, ref)
    assert (pid2 != 0xFFFFFFFF).mean() > 0.01


def test_list_pass_on_a_small_frame_of_a_deep_tree(core, acr, orc):
    """A frame of fewer workgroups than the list pass runs (96 x 96: 36 against its 64) on a mesh tree deep
    enough, with enough bounce levels' direct-term columns next to it in LDS, that the top levels of the
    waiting-child column live in global memory: the list pass indexes that column by its own workgroup numbers,
    so the column has to be sized for the larger of the two launches."""
    b = Built(acr, orc, soup_data(acr, 200000, seed=4, edge=0.12, extent=2.0), mesh_depth=18)
    load(core, b)
    w, h = 96, 96
    ref = b.oracle.render(w, h, seed=3, frames=0, max_bounces=12)
    img, pid, ht, st = render(core, w, h, seed=3, frames=0, max_bounces=12)
    assert st.scene_in_lds < 2 and st.near_first == 1 and st.grid_blocks < 64
    assert_same(img, pid, ht, st, ref)
    core.set_traversal(list_pass=True)
    try:
        img2, pid2, ht2, st2 = render(core, w, h, seed=3, frames=0, max_bounces=12)
    finally:
        core.set_traversal()
    assert st2.six_plane_pixels >= w * h
    assert_same(img2, pid2, ht2, st2, ref)
    assert (pid2 != 0xFFFFFFFF).mean() > 0.05


def test_cornell_4k_four_bounces_full_size(core, built):
    """The 4K, four-bounce configuration at its full size: 3840 x 2160, max_bounces 4, every pixel against the
    oracle; then the properties that do not depend on size: the wave-cooperative kernel (object tests handed out
    across each wavefront), the kernel in which every lane makes its own, and the six-plane form give one and
    the same frame, and a tile list reproduces the same pixels."""
    import torch
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h, bounces, seed = 3840, 2160, 4, 12345
    ref = b.oracle.render(w, h, seed=seed, frames=0, max_bounces=bounces)
    img, pid, ht, st = render(core, w, h, seed=seed, frames=0, max_bounces=bounces)
    assert st.scene_in_lds == 2 and st.interval_traversal == 1 and st.wave_tasks == 1
    assert_same(img, pid, ht, st, ref)
    del ref
    miss = pid == 0xFFFFFFFF
    assert np.all(ht[miss] == FMAX) and np.all(img[miss][:, :3] == 0) and np.all(img[miss][:, 3] == 1)
    assert np.isfinite(img).all() and pid[~miss].max() < 72 and 0.54 < (~miss).mean() < 0.56
    assert img[..., 3].max() <= bounces + 1 and st.rays_bounce > st.primary_hits  # most paths go on after the first hit
    for mode in (dict(nested=True), dict(six_plane_only=True)):
        core.set_traversal(**mode)
        try:
            img2, pid2, ht2, st2 = render(core, w, h, seed=seed, frames=0, max_bounces=bounces)
        finally:
            core.set_traversal()
        assert st2.wave_tasks == 0 and st2.interval_traversal == (0 if "six_plane_only" in mode else 1)
        assert np.array_equal(pid2, pid) and np.array_equal(ht2.view(np.uint32), ht.view(np.uint32))
        assert np.array_equal(img2.view(np.uint32), img.view(np.uint32))
        assert (st2.rays_shadow, st2.rays_bounce, st2.primary_hits) == (st.rays_shadow, st.rays_bounce, st.primary_hits)
    tw, th = 3840, 16  # full-width bands, the multi-GPU layout
    ids = np.array([134, 0, 77, 40], dtype=np.uint32)
    tiles = torch.zeros((len(ids), th * tw, 4), dtype=torch.float32, device="cuda")
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        core.render(w, h, seed=seed, frames=0, max_bounces=bounces, tile_ids=ids, tile_w=tw, tile_h=th, screen_dev=C.c_void_p(tiles.data_ptr()))
        core.sync()
    finally:
        core.use_own_stream()
    t = tiles.cpu().numpy().reshape(len(ids), th, tw, 4)
    for n, tid in enumerate(ids):
        y0 = int(tid) * th
        assert np.array_equal(t[n].view(np.uint32), img[y0:y0 + th].view(np.uint32))


def test_flythrough_camera_full_size(core, built):
    """one camera of the 120-frame flythrough at the size the flythrough is benchmarked at, every pixel"""
    from cudaraytracer_amd import camera_path
    b = built["cornell-box-cycles"]
    load(core, b)
    cams = camera_path.flythrough(b.scene.camera, 120)
    try:
        core.set_camera(cams[90])
        img, pid, ht, st = render(core, 1920, 1080, seed=777, frames=0, max_bounces=1)
        assert_same(img, pid, ht, st, b.oracle.render(1920, 1080, seed=777, frames=0, max_bounces=1, camera=orc_camera(cams[90])))
        assert 0.3 < (pid != 0xFFFFFFFF).mean() < 0.9
    finally:
        core.set_camera(b.scene.camera)


def test_path_vote_table(core, built, orc):
    """The per-pixel vote table of the reference's adaptive assignment (PathAggregate, bih.h:29-66) on the GPU against
    the oracle's: real path keys of accumulating Cornell frames, set on the first frame, added on the others; the
    counters and the majority keys are equal bit for bit, and so is the pixel table sorted by them."""
    import torch
    b = built["cornell-box-cycles"]
    load(core, b)
    w, h = 192, 108
    n = w * h
    core.resize(w, h)
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    votes = torch.full((64, n), 12345, dtype=torch.int32, device="cuda")  # whatever was there: set() overwrites
    major = torch.zeros(n, dtype=torch.int64, device="cuda")
    table = torch.zeros(n, dtype=torch.int32, device="cuda")
    ref_votes = orc.PathVotes(n)
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for f in range(5):
            core.render(w, h, seed=900 + f, frames=f, max_bounces=2, path_keys_dev=C.c_void_p(keys.data_ptr()))
            core.vote_paths(C.c_void_p(keys.data_ptr()), C.c_void_p(votes.data_ptr()), n, reset=(f == 0))
            ref = b.oracle.render(w, h, seed=900 + f, frames=f, max_bounces=2)
            ref_votes.vote(ref["path_key"], reset=(f == 0))
            core.sync()
            assert np.array_equal(keys.cpu().numpy().view(np.uint64), ref["path_key"].reshape(-1))
            assert np.array_equal(votes.cpu().numpy(), ref_votes.votes)
        core.vote_keys(C.c_void_p(votes.data_ptr()), C.c_void_p(major.data_ptr()), n)
        core.sort_pixel_map(C.c_void_p(major.data_ptr()), C.c_void_p(table.data_ptr()), n)
        core.sync()
    finally:
        core.use_own_stream()
    want = ref_votes.keys()
    assert np.array_equal(major.cpu().numpy().view(np.uint64), want)
    assert len(np.unique(want)) > 4   # the votes tell pixels apart
    assert np.array_equal(table.cpu().numpy(), np.argsort(want, kind="stable").astype(np.int32))
    # the context-owned table: allocated on request for the current size, zeroed
    p = core._lib.acr_hip_path_votes(core._h)
    assert p
    assert core._lib.acr_hip_path_votes(core._h) == p   # the same table on every call


def test_scene_with_more_materials_than_a_task_entry_names(core, acr, orc):
    """A task of the wave-cooperative kernel carries its hit's material in six bits: a scene with more than 64
    materials is rendered by the kernel in which every lane makes its own object tests, with the same frames."""
    from conftest import duplicate_triangle_scene
    data = duplicate_triangle_scene(acr, 60, seed=3)
    data.materials = [dict(diffuse=[0.1 + 0.01 * k, 0.8, 0.5], ambient=[0, 0, 0], specular=[0.3, 0.3, 0.3], refracti=0.0) for k in range(70)]
    data.meshes[0]["material"] = 69
    b = Built(acr, orc, data)
    load(core, b)
    ref = b.oracle.render(160, 96, seed=8, frames=0, max_bounces=3)
    img, pid, ht, st = render(core, 160, 96, seed=8, frames=0, max_bounces=3)
    assert st.wave_tasks == 0 and st.scene_in_lds == 2 and st.interval_traversal == 1
    assert_same(img, pid, ht, st, ref)
    data.materials = data.materials[:64]
    data.meshes[0]["material"] = 63
    b = Built(acr, orc, data)
    load(core, b)
    ref = b.oracle.render(160, 96, seed=8, frames=0, max_bounces=3)
    img, pid, ht, st = render(core, 160, 96, seed=8, frames=0, max_bounces=3)
    assert st.wave_tasks == 1
    assert_same(img, pid, ht, st, ref)


def test_wave_tasks_with_more_candidates_than_the_list_holds(core, acr, orc):
    """Thirty overlapping objects around the origin: nearly every ray enters nearly every object's box, so a wave has
    many times the 128 tasks its list holds and serves its lanes group by group (down to a lane at a time).  Frames
    equal the oracle's and the nested kernel's; translucent and mirror-like materials are among the objects."""
    d = random_instanced_scene(acr, seed=11, n_objects=30)
    for n in d.nodes:
        if n["name"].startswith("Obj"):
            m = np.array(n["transform"], np.float32).reshape(4, 4)
            m[3, :3] *= 0.3           # pull every object towards its parent's origin: the boxes pile up
            n["transform"] = [float(v) for v in m.reshape(-1)]
    b = Built(acr, orc, d)
    load(core, b)
    w, h = 200, 120
    ref = b.oracle.render(w, h, seed=31, frames=0, max_bounces=4)
    img, pid, ht, st = render(core, w, h, seed=31, frames=0, max_bounces=4)
    assert st.wave_tasks == 1 and st.scene_in_lds == 2
    assert_same(img, pid, ht, st, ref)
    assert (pid != 0xFFFFFFFF).mean() > 0.05 and st.rays_bounce > 1000
    core.set_traversal(nested=True)
    try:
        img2, pid2, ht2, st2 = render(core, w, h, seed=31, frames=0, max_bounces=4)
    finally:
        core.set_traversal()
    assert st2.wave_tasks == 0
    assert_same(img2, pid2, ht2, st2, ref)
    # accumulation through the wave kernel
    core.resize(w, h)
    screen = None
    for f in range(3):
        core.render(w, h, seed=40 + f, frames=f, max_bounces=3)
        screen = b.oracle.render(w, h, seed=40 + f, frames=f, max_bounces=3, screen=screen)["screen"]
    core.sync()
    assert np.array_equal(core.read_framebuffer().view(np.uint32), screen.view(np.uint32))


def _point_lights(data, positions):
    """adds point lights at `positions` to an ImportData, so that a pixel has several shadow rays"""
    for k, pos in enumerate(positions):
        at = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, float(pos[0]), float(pos[1]), float(pos[2]), 1]  # column-major translation
        data.nodes.append(dict(name="Extra%d" % k, parent=0, transform=at, meshes=[]))
        data.lights.append(dict(name="Extra%d" % k, type=0, position=[0, 0, 0], direction=[0, 0, -1], ambient=[2, 2, 2], diffuse=[2, 2, 2],
                                specular=[2, 2, 2], att=[0, 1, 0], cones=[6.28, 6.28]))
    return data


@pytest.mark.parametrize("ntri,depth,extent,edge", [(3000, 10, 1.0, 0.08), (60000, 14, 1.0, 0.1), (400000, 17, 3.0, 0.15)])
@pytest.mark.parametrize("bounces", [1, 4])
def test_shared_walks_of_one_deep_mesh(core, acr, orc, ntri, depth, extent, edge, bounces):
    """One object with one mesh resident in HBM: the lanes of a wavefront share their walks through the mesh tree,
    in frames without bounces (max_bounces 1, a kernel variant of its own) and with them.  Frames equal the oracle's
    and the one-lane-per-ray kernel's bit for bit: full frames of sizes that are and are not multiples of the 16 x 16
    blocks, running means over several frames, a tile list, ray counters."""
    b = Built(acr, orc, soup_data(acr, ntri, seed=9, edge=edge, extent=extent), mesh_depth=depth)
    load(core, b)
    for (w, h) in ((256, 144), (203, 77)):
        ref = b.oracle.render(w, h, seed=31, frames=0, max_bounces=bounces)
        img, pid, ht, st = render(core, w, h, seed=31, frames=0, max_bounces=bounces)
        assert st.shared_walk == 1 and st.near_first == 1 and st.scene_in_lds == 1  # the small tables of one object always fit
        assert_same(img, pid, ht, st, ref)
        assert (pid != 0xFFFFFFFF).mean() > 0.05
        core.set_traversal(nested=True)
        try:
            img1, pid1, ht1, st1 = render(core, w, h, seed=31, frames=0, max_bounces=bounces)
        finally:
            core.set_traversal()
        assert st1.shared_walk == 0 and st1.near_first == 1
        assert np.array_equal(img1.view(np.uint32), img.view(np.uint32)) and np.array_equal(pid1, pid)
        assert np.array_equal(ht1.view(np.uint32), ht.view(np.uint32))
        assert (st1.rays_primary, st1.rays_shadow, st1.rays_bounce, st1.primary_hits) == (st.rays_primary, st.rays_shadow, st.rays_bounce, st.primary_hits)
        assert (st.rays_bounce > 0) == (bounces > 1)
    # running mean of four frames (renderer.cu:375-387)
    w, h = 203, 77
    core.resize(w, h)
    acc = None
    for k in range(4):
        acc = b.oracle.render(w, h, seed=40 + k, frames=k, max_bounces=bounces, screen=acc)["screen"]
        core.render(w, h, seed=40 + k, frames=k, max_bounces=bounces, collect_stats=True)
        core.sync()
        assert core.read_stats().shared_walk == 1
        assert np.array_equal(core.read_framebuffer().view(np.uint32), acc.view(np.uint32)), "frame %d" % k
    # a tile list, tiles out of order, one cut by the frame's edge
    import torch
    w, h, tw, th = 256, 144, 64, 16
    ref = b.oracle.render(w, h, seed=31, frames=0, max_bounces=bounces)
    tx = (w + tw - 1) // tw
    ids = np.array([2 * tx + 3, 8 * tx + 1, 0, 5 * tx + 2], dtype=np.uint32)
    tiles = torch.zeros((len(ids), th * tw, 4), dtype=torch.float32, device="cuda")
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        core.render(w, h, seed=31, frames=0, max_bounces=bounces, tile_ids=ids, tile_w=tw, tile_h=th, screen_dev=C.c_void_p(tiles.data_ptr()))
        core.sync()
        assert core.read_stats().shared_walk == 1
    finally:
        core.use_own_stream()
    t = tiles.cpu().numpy().reshape(len(ids), th, tw, 4)
    for k, tid in enumerate(ids):
        x0, y0 = (int(tid) % tx) * tw, (int(tid) // tx) * th
        ww, hh = min(tw, w - x0), min(th, h - y0)
        assert np.array_equal(t[k, :hh, :ww].view(np.uint32), ref["screen"][y0:y0 + hh, x0:x0 + ww].view(np.uint32))


def test_shared_walks_exact_ties_and_several_lights(core, acr, orc):
    """The shared walk on the scene where every hit is an exact tie between two copies of a triangle (the copy the
    reference tests first must win whoever finds which), and on a soup lit by three lights (three shadow rays a pixel)."""
    b = Built(acr, orc, duplicate_triangle_scene(acr, 30000, spread=0.05), mesh_depth=14)
    load(core, b)
    w, h = 256, 256
    ref = b.oracle.render(w, h, seed=4, frames=0, max_bounces=1)
    img, pid, ht, st = render(core, w, h, seed=4, frames=0, max_bounces=1)
    assert st.shared_walk == 1
    assert_same(img, pid, ht, st, ref)
    assert 0.2 < (pid[pid != 0xFFFFFFFF] < 30000).mean() < 0.8
    data = soup_data(acr, 60000, seed=5, edge=0.1, extent=1.0)
    b = Built(acr, orc, _point_lights(data, [(3.0, 2.0, 4.0), (-3.0, 1.0, 3.0)]), mesh_depth=14)
    load(core, b)
    w, h = 240, 135
    for bounces in (1, 3):
        ref = b.oracle.render(w, h, seed=8, frames=0, max_bounces=bounces)
        img, pid, ht, st = render(core, w, h, seed=8, frames=0, max_bounces=bounces)
        assert st.shared_walk == 1
        assert_same(img, pid, ht, st, ref)
        assert st.rays_shadow > st.primary_hits  # more than one shadow ray a hit on average


@pytest.mark.parametrize("seed,depth", [(3, 12), (8, 14)])
def test_shared_walks_of_several_objects(core, acr, orc, seed, depth):
    """Several objects instancing large meshes (trees in HBM; rotations, non-uniform scales, nested parents, objects of
    two meshes, translucent and mirror-like materials, two lights): the wave goes through the objects one after the
    other and walks every mesh tree together.  Frames, primitive ids, distances and ray counters equal the oracle's and
    the one-lane-per-ray kernel's, with 1 and 4 levels, over accumulated frames and through a tile list."""
    import torch
    b = Built(acr, orc, instanced_cloud_scene(acr, seed), mesh_depth=depth)
    load(core, b)
    w, h = 240, 135
    for bounces in (1, 4):
        ref = b.oracle.render(w, h, seed=17, frames=0, max_bounces=bounces)
        img, pid, ht, st = render(core, w, h, seed=17, frames=0, max_bounces=bounces)
        assert st.shared_walk == 1 and st.near_first == 1 and st.scene_in_lds == 1
        assert_same(img, pid, ht, st, ref)
        assert (pid != 0xFFFFFFFF).mean() > 0.2
        core.set_traversal(nested=True)
        try:
            img1, pid1, ht1, st1 = render(core, w, h, seed=17, frames=0, max_bounces=bounces)
        finally:
            core.set_traversal()
        assert st1.shared_walk == 0
        assert np.array_equal(img1.view(np.uint32), img.view(np.uint32)) and np.array_equal(pid1, pid)
    core.resize(w, h)
    acc = None
    for k in range(3):
        acc = b.oracle.render(w, h, seed=50 + k, frames=k, max_bounces=3, screen=acc)["screen"]
        core.render(w, h, seed=50 + k, frames=k, max_bounces=3)
    core.sync()
    assert np.array_equal(core.read_framebuffer().view(np.uint32), acc.view(np.uint32))
    ref = b.oracle.render(w, h, seed=17, frames=0, max_bounces=2)
    tw, th = 64, 16
    tx, ty = (w + tw - 1) // tw, (h + th - 1) // th
    ids = np.random.default_rng(1).permutation(tx * ty).astype(np.uint32)[:11]
    tiles = torch.zeros((len(ids), th * tw, 4), dtype=torch.float32, device="cuda")
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        core.render(w, h, seed=17, frames=0, max_bounces=2, tile_ids=ids, tile_w=tw, tile_h=th, screen_dev=C.c_void_p(tiles.data_ptr()))
        core.sync()
        assert core.read_stats().shared_walk == 1
    finally:
        core.use_own_stream()
    t = tiles.cpu().numpy().reshape(len(ids), th, tw, 4)
    for k, tid in enumerate(ids):
        x0, y0 = (int(tid) % tx) * tw, (int(tid) // tx) * th
        ww, hh = min(tw, w - x0), min(th, h - y0)
        assert np.array_equal(t[k, :hh, :ww].view(np.uint32), ref["screen"][y0:y0 + hh, x0:x0 + ww].view(np.uint32))
    # raw ray queries go through the one-lane-per-ray walk: the same hits
    rng = np.random.default_rng(3)
    n = 20000
    org = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    g, r = core.trace_rays(org, d), b.oracle.trace_rays(org, d)
    assert np.array_equal(g["hit"], r["hit"]) and np.array_equal(g["prim"], r["prim"])
    assert np.array_equal(g["t"].view(np.uint32), r["t"].view(np.uint32))


def test_rays_with_nan_and_zero_directions(core, built, acr, orc):
    """Rays whose direction has a NaN component, or is the zero vector (what refract returns on total internal
    reflection, renderer.cu:299), hit nothing in the reference whatever its traversal does with them; the interval
    kernels say so without a traversal.  Raw ray queries on a scene resident in LDS, on a one-mesh scene and on a scene of
    several objects in HBM against the oracle, which walks the trees with such rays as the reference does; and frames
    with bounces of the cloud scene, whose degenerate normals produce NaN directions by the thousand, without listed
    pixels."""
    rng = np.random.default_rng(11)
    scenes = [built["cornell-box-cycles"], Built(acr, orc, soup_data(acr, 60000, seed=9, edge=0.1, extent=1.0), mesh_depth=14),
              Built(acr, orc, instanced_cloud_scene(acr, 8), mesh_depth=14)]
    for b in scenes:
        load(core, b)
        mn, mx = b.oracle.bounds()
        n = 3000
        org = rng.uniform(mn, mx, (n, 3)).astype(np.float32)  # inside the scene's box: every slab term is there to go wrong
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[:500] = 0.0
        d[250:500, 1] = -0.0
        for k in range(3):
            d[500 + 300 * k:800 + 300 * k, k] = np.nan
        d[1400:1600] = np.nan
        d[1600:1700, 0] = np.nan
        d[1600:1700, 2] = 0.0
        g, r = core.trace_rays(org, d), b.oracle.trace_rays(org, d)
        assert np.array_equal(g["hit"], r["hit"]) and np.array_equal(g["prim"], r["prim"])
        assert np.array_equal(g["t"].view(np.uint32), r["t"].view(np.uint32))
        assert not r["hit"][:1700].any() and r["hit"][1700:].sum() > 100
    b = scenes[2]
    w, h = 240, 135
    ref = b.oracle.render(w, h, seed=5, frames=0, max_bounces=4)
    for nested in (False, True):
        core.set_traversal(nested=nested)
        try:
            img, pid, ht, st = render(core, w, h, seed=5, frames=0, max_bounces=4)
        finally:
            core.set_traversal()
        assert_same(img, pid, ht, st, ref)
        assert st.six_plane_pixels <= 2


def test_shared_walks_under_spot_and_directional_lights(core, acr, orc):
    """The shared-walk kernels carry their own copy of spotLightAccum (scene.cu:234-271): a one-mesh scene and a scene
    of several objects in HBM under a spot light and a "directional" one, with 1 and 3 levels.  Geometry is held to bit
    equality with the oracle, radiance to the 1e-4 tolerance that acosf needs (libm against the device library, see
    test_spot_and_directional_lights); and the frames equal the one-lane-per-ray kernel's bit for bit, which calls the
    same device acosf."""
    for data in (soup_data(acr, 60000, seed=9, edge=0.1, extent=1.0), instanced_cloud_scene(acr, 8)):
        l0 = data.lights[0]
        l0.update(type=2, direction=[0.1, -0.2, -1.0], cones=[1.2, 1.6])
        extra = dict(l0)
        extra.update(name="Dir", type=1, direction=[0.3, 0.2, -1.0], cones=[2.4, 3.0])
        data.nodes.append(dict(name="Dir", parent=0, transform=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -2.0, 3.0, 6.0, 1], meshes=[]))
        data.lights.append(extra)
        b = Built(acr, orc, data, mesh_depth=14)
        load(core, b)
        w, h = 256, 144
        for bounces in (1, 3):
            img, pid, ht, st = render(core, w, h, seed=5, frames=0, max_bounces=bounces)
            assert st.shared_walk == 1
            ref = b.oracle.render(w, h, seed=5, frames=0, max_bounces=bounces)
            assert np.array_equal(pid, ref["prim"]) and np.array_equal(ht.view(np.uint32), ref["t"].view(np.uint32))
            a, r = img[..., :3].astype(np.float64), ref["screen"][..., :3].astype(np.float64)
            rel = np.abs(a - r) / np.maximum(np.abs(r), 1e-3)
            assert (rel > 1e-4).mean() < 1e-3
            assert np.array_equal(img[..., 3], ref["screen"][..., 3])
            assert (r.sum(axis=2) > 0).mean() > 0.01
            core.set_traversal(nested=True)
            try:
                img1, pid1, ht1, st1 = render(core, w, h, seed=5, frames=0, max_bounces=bounces)
            finally:
                core.set_traversal()
            assert st1.shared_walk == 0
            assert np.array_equal(img1.view(np.uint32), img.view(np.uint32))
            assert (st1.rays_shadow, st1.rays_bounce) == (st.rays_shadow, st.rays_bounce)


@pytest.mark.parametrize("several", [False, True])
def test_shared_walks_follow_the_pixel_table(core, acr, orc, several):
    """DevParams::pixelValues (renderer.cu:344-355) through the shared-walk kernels: frame 0 resets the table to the
    identity, later frames follow it, on a one-mesh scene and on a scene of several objects in HBM."""
    import torch
    data = instanced_cloud_scene(acr, 8) if several else soup_data(acr, 60000, seed=9, edge=0.1, extent=1.0)
    b = Built(acr, orc, data, mesh_depth=14)
    load(core, b)
    w, h = 112, 80
    core.resize(w, h)
    for bounces in (1, 3):
        pm = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        core.render(w, h, seed=9, frames=0, max_bounces=bounces, pixel_values_dev=C.c_void_p(pm.data_ptr()), collect_stats=True)
        core.sync()
        assert core.read_stats().shared_walk == 1
        assert np.array_equal(pm.cpu().numpy(), np.arange(w * h, dtype=np.int32))
        perm = np.random.default_rng(0).permutation(w * h).astype(np.int32)
        pm.copy_(torch.from_numpy(perm))
        torch.cuda.synchronize()
        core.render(w, h, seed=4, frames=1, max_bounces=bounces, pixel_values_dev=C.c_void_p(pm.data_ptr()))
        core.sync()
        base = b.oracle.render(w, h, seed=9, frames=0, max_bounces=bounces)["screen"]
        ref = b.oracle.render(w, h, seed=4, frames=1, max_bounces=bounces, screen=base, pixel_map=perm)["screen"]
        assert np.array_equal(core.read_framebuffer().view(np.uint32), ref.view(np.uint32))
