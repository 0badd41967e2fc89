"""The oracle renderer on the CPU: its own regression vectors, the statistics
recorded from the reference's code, and structural properties of the path."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FMAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "oracle_renders.npz"))


@pytest.mark.parametrize("name", ["triangles", "plane", "cornell-box-cycles"])
@pytest.mark.parametrize("bounces", [1, 5])
def test_oracle_regression_vectors(built, gold, name, bounces):
    r = built[name].oracle.render(96, 72, seed=7, frames=0, max_bounces=bounces)
    key = f"{name}/b{bounces}"
    assert np.array_equal(r["prim"], gold[key + "/prim"])
    assert np.array_equal(r["t"].view(np.uint32), gold[key + "/t"])
    assert np.array_equal(r["screen"].view(np.uint32), gold[key + "/rgba"])


def test_oracle_is_thread_count_independent(built):
    o = built["cornell-box-cycles"].oracle
    a = o.render(160, 90, seed=3, max_bounces=3, threads=1)
    b = o.render(160, 90, seed=3, max_bounces=3, threads=8)
    assert np.array_equal(a["screen"].view(np.uint32), b["screen"].view(np.uint32))
    assert a["counters"] == b["counters"]


def test_reference_statistics_of_the_cornell_config(built):
    """Per-ray work measured with the reference's own traversal code during the
    survey of the reference (Cornell box, pi/4 field of view, primary + shadow):
    14.6 slab tests, 5.6 object visits and 4.4 triangle tests per ray, 55 %
    primary hit rate, 0.84 shadow rays per primary ray.  The random jitter
    differs, so these are checked as statistics at a quarter of the resolution."""
    c = built["cornell-box-cycles"].oracle.render(960, 540, seed=12345, max_bounces=1, aov=False)["counters"]
    rays = c["rays_primary"] + c["rays_shadow"]
    assert abs(c["slab_tests"] / rays - 14.6) < 0.1
    assert abs(c["object_visits"] / rays - 5.6) < 0.1
    assert abs(c["triangle_tests"] / rays - 4.4) < 0.1
    assert abs(c["primary_hits"] / c["rays_primary"] - 0.55) < 0.005
    assert abs(c["rays_shadow"] / c["rays_primary"] - 0.835) < 0.01
    assert c["rays_bounce"] == 0


def test_committed_counters_match(built):
    with open(os.path.join(HERE, "golden", "oracle_counters.json")) as f:
        gold = json.load(f)
    g = gold["C1 triangles 640x480 primary+shadow"]
    c = built["triangles"].oracle.render(640, 480, seed=g["seed"], max_bounces=1, aov=False)["counters"]
    for k, v in g.items():
        if k in c:
            assert c[k] == v, k
    assert c["order_doubts"] == 0  # the reference's order was used


def test_primary_hits_agree_with_trace_rays_geometry(built):
    """AOVs of the render equal a direct Scene::intersect of axis-aligned and random rays: hit flag, t, ids."""
    b = built["triangles"]
    o = b.oracle
    # camera at (0,0,5) looking down -z; the two triangles tile the square [-1,1]^2 at z = 0
    xs = np.linspace(-0.9, 0.9, 7, dtype=np.float32)
    org = np.array([[x, y, 5.0] for x in xs for y in xs], np.float32)
    dr = np.tile(np.float32([0, 0, -1]), (len(org), 1))
    r = o.trace_rays(org, dr)
    assert r["hit"].all() and np.allclose(r["t"], 5.0)
    assert np.allclose(r["position"][:, 2], 0.0, atol=1e-6) and np.allclose(np.abs(r["normal"][:, 2]), 1.0, atol=1e-4)
    # shooting away from the scene misses and leaves t at its preset
    r2 = o.trace_rays(org, -dr)
    assert not r2["hit"].any() and (r2["t"] == FMAX).all() and (r2["prim"] == 0xFFFFFFFF).all()
    # tmax below the hit distance: the slab tests cut the traversal, nothing is reported
    r3 = o.trace_rays(org, dr, tmax=np.full(len(org), 2.0, np.float32))
    assert not r3["hit"].any() and (r3["t"] == 2.0).all()


def test_accumulation_is_a_running_mean(built):
    """frames > 0 blends with the previous buffer: s*(f/(f+1)) + c/(f+1) (renderer.cu:375-379)"""
    o = built["cornell-box-cycles"].oracle
    f0 = o.render(64, 36, seed=10, frames=0, max_bounces=2)["screen"].copy()
    c1 = o.render(64, 36, seed=11, frames=0, max_bounces=2)["screen"].copy()
    acc = o.render(64, 36, seed=11, frames=1, max_bounces=2, screen=f0.copy())["screen"]
    expect = f0 * np.float32(1.0 / 2.0) + c1 / np.float32(2.0)
    assert np.array_equal(acc.view(np.uint32), expect.astype(np.float32).view(np.uint32))


def test_alpha_counts_recursion_levels_and_radiance_is_finite(built):
    o = built["cornell-box-cycles"].oracle
    r = o.render(96, 54, seed=5, max_bounces=5)
    a = r["screen"][..., 3]
    assert a.min() >= 1 and a.max() <= 6 and np.all(a == np.round(a))
    miss = r["prim"] == 0xFFFFFFFF
    assert np.all(a[miss] == 1) and np.all(r["screen"][miss][:, :3] == 0)
    assert np.isfinite(r["screen"]).all() and r["screen"][..., :3].min() >= 0


def test_zero_bounces_is_black(built):
    r = built["triangles"].oracle.render(32, 24, seed=1, max_bounces=0)
    assert np.all(r["screen"][..., :3] == 0) and np.all(r["screen"][..., 3] == 1)
    assert r["counters"]["rays_shadow"] == 0 and r["counters"]["rays_bounce"] == 0


def test_row_range_rendering_equals_full_frame(built):
    o = built["plane"].oracle
    full = o.render(80, 60, seed=2, max_bounces=2)["screen"]
    part = np.zeros_like(full)
    o.render(80, 60, seed=2, max_bounces=2, screen=part, row_begin=0, row_end=25)
    o.render(80, 60, seed=2, max_bounces=2, screen=part, row_begin=25, row_end=60)
    assert np.array_equal(full.view(np.uint32), part.view(np.uint32))


def test_pixel_map_permutes_work_not_pixels(built):
    """pixelValues (renderer.cu:343-355): slot s renders pixel map[s]; any permutation gives the same image.
    The map only applies when frames > 0 (frame 0 resets it to the identity)."""
    o = built["triangles"].oracle
    w, h = 48, 36
    base = o.render(w, h, seed=9, frames=0, max_bounces=1)["screen"].copy()
    perm = np.random.default_rng(0).permutation(w * h).astype(np.int32)
    a = o.render(w, h, seed=4, frames=1, max_bounces=1, screen=base.copy())["screen"]
    b = o.render(w, h, seed=4, frames=1, max_bounces=1, screen=base.copy(), pixel_map=perm)["screen"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_path_aggregate_votes(orc):
    """PathAggregate (bih.h:29-66) as designed: an aggregate set from one path converts back to that path; adding
    paths makes every bit a majority vote, a tie (counter 0) reads as 0; set starts over."""
    rng = np.random.default_rng(3)
    n = 500
    v = orc.PathVotes(n)
    first = rng.integers(0, 2 ** 63, n, dtype=np.uint64) | (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63))
    v.vote(first, reset=True)
    assert np.array_equal(v.keys(), first)
    assert set(np.unique(v.votes)) <= {-1, 1}
    frames = [first]
    for _ in range(4):
        k = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
        frames.append(k)
        v.vote(k, reset=False)
    ones = sum(((f[None, :] >> np.arange(64, dtype=np.uint64)[:, None]) & np.uint64(1)).astype(np.int64) for f in frames)
    assert np.array_equal(v.votes, (5 - 2 * ones).astype(np.int32))
    want = np.zeros(n, np.uint64)
    for i in range(64):
        want |= (ones[i] >= 3).astype(np.uint64) << np.uint64(i)
    assert np.array_equal(v.keys(), want)
    v.vote(frames[1], reset=False)  # six frames: three against three is no majority
    ones6 = ones + ((frames[1][None, :] >> np.arange(64, dtype=np.uint64)[:, None]) & np.uint64(1)).astype(np.int64)
    tie = (ones6 == 3)
    assert tie.any() and not ((v.keys()[None, :] >> np.arange(64, dtype=np.uint64)[:, None]) & np.uint64(1)).astype(bool)[tie].any()
    v.vote(frames[2], reset=True)
    assert np.array_equal(v.keys(), frames[2])
