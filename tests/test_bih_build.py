"""Host BIH build (reference src/utils/bih.h:129-207) and scene construction
(src/scene/scene.cu:122-152,332-405; src/geometry/geometry.cu:8-48): the
product's C++ build against the oracle's restatement, bit for bit, plus the
tree printouts the reference's own operator<< produced for the shipped scenes."""
import json
import os

import numpy as np
import pytest

from conftest import Built, random_instanced_scene, soup_data

HERE = os.path.dirname(os.path.abspath(__file__))


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_nodes_equal(on, pn):
    d = on["defined"]
    assert np.array_equal(d, pn["defined"])
    se = (d & 1) != 0
    lr = (d & 2) != 0
    lf = (d & 4) != 0
    assert np.array_equal(on["start"][se], pn["start"][se]) and np.array_equal(on["end"][se], pn["end"][se])
    assert np.array_equal(_bits(on["left"])[lr], _bits(pn["left"])[lr])
    assert np.array_equal(_bits(on["right"])[lr], _bits(pn["right"])[lr])
    assert np.array_equal(on["axis"][lr], (pn["flags"] & 3)[lr])
    assert np.array_equal(on["is_leaf"][lf], ((pn["flags"] >> 2) & 1)[lf])


def check_scene(b):
    o, s = b.oracle, b.scene
    on, oord = o.scene_bih()
    assert_nodes_equal(on, s.scene_nodes())
    assert np.array_equal(oord, s.object_order())
    mn, mx = o.bounds()
    assert np.array_equal(_bits(mn), _bits(np.array(list(s.soa.scene_min), np.float32)))
    assert np.array_equal(_bits(mx), _bits(np.array(list(s.soa.scene_max), np.float32)))
    assert o.num_objects == s.soa.num_objects and o.num_meshes == s.soa.num_meshes
    assert o.num_triangles == s.soa.num_triangles
    for m in range(o.num_meshes):
        mnodes, mord = o.mesh_bih(m)
        assert_nodes_equal(mnodes, s.mesh_nodes(m))
        assert np.array_equal(mord, s.face_order(m, len(mord)))
        info = o.mesh_info(m)
        assert np.array_equal(_bits(info["min"]), _bits(np.ctypeslib.as_array(s.soa.mesh_min, (3 * o.num_meshes,))[3 * m:3 * m + 3]))
        assert np.array_equal(_bits(info["max"]), _bits(np.ctypeslib.as_array(s.soa.mesh_max, (3 * o.num_meshes,))[3 * m:3 * m + 3]))
        assert info["material"] == s.soa.mesh_material[m]
    n = o.num_objects
    if n:
        T = np.ctypeslib.as_array(s.soa.obj_transform, (16 * n,))
        TI = np.ctypeslib.as_array(s.soa.obj_inv_transform, (16 * n,))
        N = np.ctypeslib.as_array(s.soa.obj_normal_transform, (9 * n,))
        NI = np.ctypeslib.as_array(s.soa.obj_inv_normal_transform, (9 * n,))
        for i in range(n):
            ob = o.object(i)
            assert np.array_equal(_bits(ob["transform"]), _bits(T[16 * i:16 * i + 16]))
            assert np.array_equal(_bits(ob["inv_transform"]), _bits(TI[16 * i:16 * i + 16]))
            assert np.array_equal(_bits(ob["normal"]), _bits(N[9 * i:9 * i + 9]))
            assert np.array_equal(_bits(ob["inv_normal"]), _bits(NI[9 * i:9 * i + 9]))
    for i in range(o.num_lights):
        l = o.light(i)
        assert l["type"] == s.soa.light_type[i]
        assert np.array_equal(_bits(l["position"]), _bits(np.ctypeslib.as_array(s.soa.light_position, (3 * o.num_lights,))[3 * i:3 * i + 3]))
    oc, pc = o.camera, s.camera
    for k in ("position", "forward", "up"):
        assert np.array_equal(_bits(np.array(list(getattr(oc, k)), np.float32)), _bits(np.array(list(getattr(pc, k)), np.float32)))


@pytest.mark.parametrize("name", ["triangles", "plane", "cornell-box-cycles"])
def test_shipped_scenes_product_equals_oracle(built, name):
    check_scene(built[name])


@pytest.mark.parametrize("ntri,depth", [(1, 6), (2, 1), (7, 3), (100, 6), (1000, 6), (5000, 10), (20000, 14),
                                        (150000, 16)])  # the last one is large enough for the product's task-parallel build
def test_soup_product_equals_oracle(acr, orc, ntri, depth):
    b = Built(acr, orc, soup_data(acr, ntri, seed=ntri + depth), scene_depth=6, mesh_depth=depth)
    check_scene(b)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_instanced_scenes_product_equals_oracle(acr, orc, seed):
    """nested parents, non-uniform scales, several meshes per object, 40 objects in a depth-6 tree"""
    check_scene(Built(acr, orc, random_instanced_scene(acr, seed)))


def test_reference_bih_printouts(built):
    """Tree printouts recorded from the reference's own BIH code (bih.h:301-314) for the shipped scenes."""
    with open(os.path.join(HERE, "golden", "reference_bih_prints.json")) as f:
        gold = json.load(f)
    norm = lambda t: " ".join(t.replace("\t", "").replace("[", " [").split()).replace(": [", ":[").replace(" :", ":")
    for name, g in gold["scene_bih"].items():
        assert norm(built[name].scene.bih_text()) == norm(g), name
    c = built["cornell-box-cycles"].scene
    checked = 0
    for m in range(c.soa.num_meshes):
        txt = norm(c.bih_text(m))
        if txt.startswith("0:[0:2]") or txt.startswith("0:[0:10]"):  # the mesh sizes the record covers
            assert txt in [norm(x) for x in gold["cornell_mesh_bih_allowed"]], txt
            checked += 1
    assert checked == 12
    mn, mx = gold["cornell_world_aabb"]["min"], gold["cornell_world_aabb"]["max"]
    assert np.allclose(list(c.soa.scene_min), mn, atol=1e-3) and np.allclose(list(c.soa.scene_max), mx, atol=1e-3)


def test_leaves_partition_items_and_order_is_a_permutation(acr, orc):
    b = Built(acr, orc, soup_data(acr, 3000, seed=9), mesh_depth=9)
    s = b.scene
    nodes = s.mesh_nodes(0)
    order = s.face_order(0, 3000)
    assert sorted(order.tolist()) == list(range(3000))
    # reachable leaves, visited left to right, tile [0, n) exactly once
    covered = []
    stack = [0]
    while stack:
        i = stack.pop()
        if (nodes["flags"][i] >> 2) & 1:
            covered.append((int(nodes["start"][i]), int(nodes["end"][i])))
        else:
            stack.append(2 * i + 2)
            stack.append(2 * i + 1)
    at = 0
    for a, e in covered:
        assert a == at and e >= a
        at = e
    assert at == 3000


def test_empty_and_single_item_scenes(acr, orc):
    d = soup_data(acr, 1)
    b = Built(acr, orc, d)
    check_scene(b)
    nodes = b.scene.mesh_nodes(0)
    # one triangle: the root splits into [0:1] and an empty child, so the root itself is the leaf (bih.h:203-205)
    assert (nodes["flags"][0] >> 2) & 1
    # a scene whose only mesh node is removed: no objects at all
    d.nodes[1]["meshes"] = []
    b2 = Built(acr, orc, d)
    assert b2.scene.soa.num_objects == 0 and b2.oracle.num_objects == 0
    check_scene(b2)
