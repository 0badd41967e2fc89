"""The near-side-first visiting order of mesh trees (what the GPU path uses for deep trees) returns what the
reference's left-first order returns: checked on the CPU with the oracle walking the same trees both ways.

The near-side-first walk is NOT part of the reference; the oracle carries it to count that order's work and
for this check (oracle/acr_oracle.c, mesh_bih_intersect_near)."""
import numpy as np
import pytest

from conftest import duplicate_triangle_scene, random_instanced_scene, soup_data
from oracle import OracleScene


def both_orders(o, w, h, **kw):
    o.set_mesh_order(False)
    a = o.render(w, h, **kw)
    o.set_mesh_order(True)
    b = o.render(w, h, **kw)
    o.set_mesh_order(False)
    return a, b


def same(a, b):
    assert np.array_equal(a["screen"].view(np.uint32), b["screen"].view(np.uint32))
    for k in ("prim", "t", "material", "position", "normal", "path_key"):
        assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k],
                              b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), k
    for k in ("rays_primary", "rays_shadow", "rays_bounce", "primary_hits", "object_visits", "object_hits"):
        assert a["counters"][k] == b["counters"][k], k


@pytest.mark.parametrize("ntri,depth,extent,edge", [(2000, 8, 1.0, 0.2), (60000, 14, 1.0, 0.1), (200000, 16, 3.0, 0.15)])
def test_soup_frames_do_not_depend_on_the_order(acr, orc, ntri, depth, extent, edge):
    data = soup_data(acr, ntri, seed=3, edge=edge, extent=extent)
    view, keep = data.pack()
    o = OracleScene(view, 6, depth)
    try:
        for bounces in (1, 3):
            a, b = both_orders(o, 192, 128, seed=11 + bounces, max_bounces=bounces)
            same(a, b)
            assert a["counters"]["primary_hits"] > 192 * 128 // 10
            # the point of the order: far fewer triangle tests and node visits in a deep tree
            if depth >= 14:
                assert b["counters"]["triangle_tests"] < 0.7 * a["counters"]["triangle_tests"]
                assert b["counters"]["node_visits"] < a["counters"]["node_visits"]
    finally:
        o.close()


def test_exact_ties_go_to_the_item_the_reference_finds_first(acr, orc):
    """Coplanar duplicates: every hit is an exact tie between two triangles; the lower item position must win
    in both orders.  200 random triangles, each present twice, in a depth-8 tree."""
    data = duplicate_triangle_scene(acr, 200)
    view, keep = data.pack()
    o = OracleScene(view, 6, 8)
    try:
        a, b = both_orders(o, 160, 160, seed=2, max_bounces=2)
        same(a, b)
        assert a["counters"]["primary_hits"] > 1000
    finally:
        o.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_instanced_scenes_do_not_depend_on_the_order(acr, orc, seed):
    view, keep = random_instanced_scene(acr, seed).pack()
    o = OracleScene(view, 6, 6)
    try:
        a, b = both_orders(o, 160, 120, seed=seed, max_bounces=4)
        same(a, b)
    finally:
        o.close()
