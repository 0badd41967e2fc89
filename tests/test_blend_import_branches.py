"""Branches of the .blend import the shipped scenes do not reach (the reference's counterparts: node hierarchy and
light types at src/scene/scene.cu:155-202,303-330, materials at src/materials/materials.cpp:5-17, assimp's
triangulation and per-material mesh split behind src/scene/scene.cu:30-35): parent chains, SUN and SPOT lamps,
several material slots on one mesh, n-gons, the default material, merged corners.  Inputs are .blend files written
by tests/blend_writer.py.  The conventions checked here are this loader's stated decisions (assimp is absent from
the reference tree): parity unpinned."""
import numpy as np
import pytest

from blend_writer import LA_LOCAL, LA_SPOT, LA_SUN, OB_CAMERA, OB_EMPTY, OB_LAMP, OB_MESH, BlendFile, trs

UP = (0.0, 0.0, 1.0)


def quad(bf, name, mats=(), slots=(0,), z=0.0):
    v = [(-1, -1, z), (1, -1, z), (1, 1, z), (-1, 1, z)]
    return bf.mesh(name, v, [UP] * 4, [((0, 1, 2, 3), s) for s in slots], mats)


def mat4(node):
    return np.array(node["transform"], np.float32).reshape(4, 4).T  # column-major -> usual sense


def names(d):
    return [n["name"] for n in d.nodes]


def test_parent_chain_gives_local_transforms_and_the_scene_restores_the_world_ones(acr, tmp_path):
    bf = BlendFile()
    A = trs((1, 2, 3), (0, 0, 1), 0.7, (2, 2, 2))
    B = trs((-4, 0.5, 1), (1, 1, 0), -1.1, (0.5, 0.5, 0.5))
    Cm = trs((0.25, -3, 2), (0, 1, 0), 2.0)
    me = quad(bf, "Q")
    a = bf.object("A", OB_EMPTY, 0, A)
    b = bf.object("B", OB_MESH, me, B, parent=a)
    bf.object("C", OB_MESH, me, Cm, parent=b)
    bf.object("Cam", OB_CAMERA, bf.camera("Cam"), trs((0, 0, 10)))
    d = acr.load_blend(bf.write(str(tmp_path / "chain.blend")))
    nm = names(d)
    assert nm[0] == "<BlenderRoot>" and d.nodes[0]["parent"] == -1 and np.array_equal(mat4(d.nodes[0]), np.eye(4, dtype=np.float32))
    ia, ib, ic = nm.index("A"), nm.index("B"), nm.index("C")
    assert d.nodes[ia]["parent"] == 0 and d.nodes[ib]["parent"] == ia and d.nodes[ic]["parent"] == ib
    # a node's transform is inverse(parent obmat) * obmat; top-level objects keep their obmat
    assert np.array_equal(mat4(d.nodes[ia]), A)
    assert np.allclose(mat4(d.nodes[ib]), np.linalg.inv(A.astype(np.float64)) @ B, atol=2e-6)
    assert np.allclose(mat4(d.nodes[ic]), np.linalg.inv(B.astype(np.float64)) @ Cm, atol=2e-6)
    # the scene multiplies the chain back together (scene.cu:363-364): the objects' global transforms are the obmats
    scene = acr.Scene(import_data=d)
    s = scene.soa
    assert s.num_objects == 2  # the two mesh objects; the empty and the camera carry no meshes
    got = [np.ctypeslib.as_array(s.obj_transform, (s.num_objects * 16,))[16 * k:16 * k + 16].reshape(4, 4).T for k in range(2)]
    for want in (B, Cm):
        assert any(np.allclose(g, want, atol=1e-5) for g in got)
    # both instance the one mesh
    assert d.nodes[ib]["meshes"] == d.nodes[ic]["meshes"] or len(d.meshes) == 2


def test_sun_and_spot_lamps(acr, tmp_path):
    bf = BlendFile()
    quad(bf, "unused")
    bf.object("Floor", OB_MESH, quad(bf, "F"), trs())
    bf.object("Sun", OB_LAMP, bf.lamp("Sun", LA_SUN, (1.0, 0.5, 0.25), 2.0), trs((0, 0, 5), (1, 0, 0), np.pi / 2))
    bf.object("Spot", OB_LAMP, bf.lamp("Spot", LA_SPOT, (0.2, 0.4, 0.6), 10.0, spotsize=0.9), trs((3, 0, 4)))
    bf.object("Bulb", OB_LAMP, bf.lamp("Bulb", LA_LOCAL, (1, 1, 1), 3.0), trs((-2, 1, 2)))
    bf.object("Cam", OB_CAMERA, bf.camera("Cam"), trs((0, 0, 10)))
    d = acr.load_blend(bf.write(str(tmp_path / "lamps.blend")))
    by = {l["name"]: l for l in d.lights}
    assert by["Sun"]["type"] == 1 and by["Spot"]["type"] == 2 and by["Bulb"]["type"] == 0
    assert np.allclose(by["Sun"]["diffuse"], [2.0, 1.0, 0.5]) and np.allclose(by["Spot"]["diffuse"], [2.0, 4.0, 6.0]) and np.allclose(by["Bulb"]["diffuse"], 3.0)
    assert by["Sun"]["direction"] == [0, 0, -1] and by["Spot"]["direction"] == [0, 0, -1]
    assert np.allclose(by["Spot"]["cones"], 0.9) and np.allclose(by["Sun"]["cones"], 2 * np.pi) and np.allclose(by["Bulb"]["cones"], 2 * np.pi)
    assert all(l["att"] == [0.0, 1.0, 0.0] and l["position"] == [0, 0, 0] for l in d.lights)
    # the scene moves lights to world space through the node of the same name (scene.cu:174-183)
    scene = acr.Scene(import_data=d)
    s = scene.soa
    assert s.num_lights == 3
    order = [l["name"] for l in d.lights]
    pos = np.ctypeslib.as_array(s.light_position, (9,)).reshape(3, 3)
    direction = np.ctypeslib.as_array(s.light_direction, (9,)).reshape(3, 3)
    types = list(np.ctypeslib.as_array(s.light_type, (3,)))
    assert types == [by[n]["type"] for n in order]
    assert np.allclose(pos[order.index("Sun")], [0, 0, 5], atol=1e-6) and np.allclose(pos[order.index("Spot")], [3, 0, 4], atol=1e-6)
    assert np.allclose(direction[order.index("Sun")], [0, 1, 0], atol=1e-6)    # (0,0,-1) turned a quarter about x
    assert np.allclose(direction[order.index("Spot")], [0, 0, -1], atol=1e-6)


def test_material_slots_split_a_mesh_and_number_materials_by_first_use(acr, tmp_path):
    bf = BlendFile()
    red = bf.material("Red", (0.8, 0.1, 0.1), spec=(0.3, 0.3, 0.3), amb=(0.01, 0.02, 0.03))
    blue = bf.material("Blue", (0.1, 0.1, 0.9))
    v = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (2, 0, 0), (2, 1, 0)]
    # slot 1 (blue) is used by the first polygon, slot 0 (red) by the second: meshes come in ascending slot order
    me = bf.mesh("Two", v, [UP] * 6, [((1, 4, 5, 2), 1), ((0, 1, 2, 3), 0)], (red, blue))
    bf.object("Two", OB_MESH, me, trs())
    bf.object("Bare", OB_MESH, quad(bf, "Bare"), trs((5, 0, 0)))       # no material slots at all
    bf.object("Empty", OB_MESH, quad(bf, "EmptySlot", (0,)), trs((9, 0, 0)))  # one slot that holds no material
    bf.object("Cam", OB_CAMERA, bf.camera("Cam"), trs((0, 0, 10)))
    d = acr.load_blend(bf.write(str(tmp_path / "mats.blend")))
    two = d.nodes[names(d).index("Two")]["meshes"]
    assert len(two) == 2
    m0, m1 = d.meshes[two[0]], d.meshes[two[1]]
    assert np.allclose(d.materials[m0["material"]]["diffuse"], [0.8, 0.1, 0.1]) and np.allclose(d.materials[m1["material"]]["diffuse"], [0.1, 0.1, 0.9])
    assert m0["material"] == 0 and m1["material"] == 1   # numbered in order of first use
    assert np.allclose(d.materials[0]["specular"], 0.3) and np.allclose(d.materials[0]["ambient"], [0.01, 0.02, 0.03]) and d.materials[0]["refracti"] == 0.0
    assert len(m0["indices"]) == 2 and len(m1["indices"]) == 2 and len(m0["positions"]) == 4 and len(m1["positions"]) == 4
    assert m0["positions"][:, 0].max() == 1.0 and m1["positions"][:, 0].min() == 1.0   # each mesh holds only its own corners
    # meshes without a material share one default grey
    bare = d.meshes[d.nodes[names(d).index("Bare")]["meshes"][0]]["material"]
    empty = d.meshes[d.nodes[names(d).index("Empty")]["meshes"][0]]["material"]
    assert bare == empty == 2 and len(d.materials) == 3
    assert np.allclose(d.materials[2]["diffuse"], 0.6) and np.allclose(d.materials[2]["specular"], 0.6) and np.allclose(d.materials[2]["ambient"], 0.0)


def test_ngons_fan_out_and_equal_corners_merge(acr, tmp_path):
    bf = BlendFile()
    ang = np.linspace(0, 2 * np.pi, 5, endpoint=False)
    pent = [(float(np.cos(a)), float(np.sin(a)), 0.0) for a in ang]
    # a pentagon, a triangle sharing two of its corners, and a degenerate two-corner polygon that is skipped
    me = bf.mesh("P", pent + [(3.0, 0.0, 0.0)], [UP] * 6, [((0, 1, 2, 3, 4), 0), ((0, 5, 1), 0), ((0, 1), 0)])
    bf.object("P", OB_MESH, me, trs())
    # the same position listed twice with equal normals merges, with different normals it does not
    me2 = bf.mesh("D", [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 0), (0, 0, 0)], [UP, UP, UP, UP, (1.0, 0.0, 0.0)],
                  [((0, 1, 2), 0), ((3, 2, 1), 0), ((4, 1, 2), 0)])
    bf.object("D", OB_MESH, me2, trs((4, 0, 0)))
    bf.object("Cam", OB_CAMERA, bf.camera("Cam"), trs((0, 0, 10)))
    d = acr.load_blend(bf.write(str(tmp_path / "ngon.blend")))
    p = d.meshes[d.nodes[names(d).index("P")]["meshes"][0]]
    assert p["indices"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 5, 1]]   # fan from the first corner; corners numbered as met
    assert len(p["positions"]) == 6 and np.allclose(p["positions"][:5], pent, atol=1e-7) and np.allclose(p["normals"], [UP] * 6, atol=1e-4)
    q = d.meshes[d.nodes[names(d).index("D")]["meshes"][0]]
    assert len(q["positions"]) == 4 and q["indices"].tolist() == [[0, 1, 2], [0, 2, 1], [3, 1, 2]]
    # and the whole thing renders through the host scene build
    scene = acr.Scene(import_data=d)
    assert scene.soa.num_triangles == 7 and scene.soa.num_objects == 2


def test_a_scene_without_a_camera_is_refused(acr, tmp_path):
    bf = BlendFile()
    bf.object("Q", OB_MESH, quad(bf, "Q"), trs())
    with pytest.raises(RuntimeError, match="camera"):
        acr.load_blend(bf.write(str(tmp_path / "nocam.blend")))


def test_second_camera_is_ignored_and_lamp_under_a_parent_lands_in_world_space(acr, tmp_path):
    bf = BlendFile()
    rig = bf.object("Rig", OB_EMPTY, 0, trs((10, 0, 0), (0, 0, 1), np.pi / 2))
    bf.object("Key", OB_LAMP, bf.lamp("Key", LA_LOCAL, (1, 1, 1), 1.0), trs((10, 2, 0), (0, 0, 1), np.pi / 2), parent=rig)
    bf.object("Q", OB_MESH, quad(bf, "Q"), trs())
    bf.object("CamA", OB_CAMERA, bf.camera("CamA"), trs((0, -8, 1), (1, 0, 0), np.pi / 2), parent=rig)
    bf.object("CamB", OB_CAMERA, bf.camera("CamB"), trs((0, 0, 10)))
    d = acr.load_blend(bf.write(str(tmp_path / "rig.blend")))
    assert d.camera["name"] == "CamA" and d.camera["position"] == [0, 0, 0] and d.camera["look_at"] == [0, 0, -1] and d.camera["up"] == [0, 1, 0]
    assert abs(d.camera["hfov"] - np.pi / 4) < 1e-7
    scene = acr.Scene(import_data=d)
    cam = scene.camera
    assert np.allclose(list(cam.position), [0, -8, 1], atol=1e-5)
    assert np.allclose(list(cam.forward), [0, 1, 0], atol=1e-5) and np.allclose(list(cam.up), [0, 0, 1], atol=1e-5)
    pos = np.ctypeslib.as_array(scene.soa.light_position, (3,))
    assert np.allclose(pos, [10, 2, 0], atol=1e-5)
