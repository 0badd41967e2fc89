"""The built-in .blend loader (the replacement for the reference's assimp call,
src/scene/scene.cu:30-35).  The decoded scenes are committed as fixtures; where
the reference checkout is present the loader is re-run on its scene files and
must reproduce the fixtures bit for bit.  Import conventions themselves are
parity-unpinned: assimp is absent from the reference tree."""
import json
import os

import numpy as np
import pytest

from conftest import REFERENCE, SCENES, fixture_path

have_ref = os.path.isdir(os.path.join(REFERENCE, "scenes"))


@pytest.mark.skipif(not have_ref, reason="reference checkout not present")
@pytest.mark.parametrize("name", SCENES)
def test_loader_reproduces_fixture(acr, name):
    d = acr.load_blend(os.path.join(REFERENCE, "scenes", name + ".blend"))
    with open(fixture_path(name)) as f:
        assert json.loads(d.to_json()) == json.load(f)


@pytest.mark.skipif(not have_ref, reason="reference checkout not present")
def test_scene_args_file_path_constructor(acr):
    """Scene::Args{filePath} path of the C++ API equals building from the decoded import scene."""
    a = acr.Scene(file_path=os.path.join(REFERENCE, "scenes", "cornell-box-cycles.blend"))
    b = acr.Scene(import_data=acr.ImportData.load(fixture_path("cornell-box-cycles")))
    assert a.bih_text() == b.bih_text()
    assert a.soa.num_triangles == b.soa.num_triangles == 72
    n = 72 * 3
    assert np.array_equal(np.ctypeslib.as_array(a.soa.tri_p0, (n,)).view(np.uint32), np.ctypeslib.as_array(b.soa.tri_p0, (n,)).view(np.uint32))


def test_loader_errors(acr, tmp_path):
    with pytest.raises(RuntimeError):
        acr.load_blend(tmp_path / "missing.blend")
    p = tmp_path / "junk.blend"
    p.write_bytes(b"not a blend file" * 10)
    with pytest.raises(RuntimeError, match="blend"):
        acr.load_blend(p)
    q = tmp_path / "be.blend"
    q.write_bytes(b"BLENDER_V274" + b"\0" * 64)
    with pytest.raises(RuntimeError, match="64-bit little-endian"):
        acr.load_blend(q)
    with pytest.raises(RuntimeError):
        acr.Scene(file_path=str(p))


def test_fixture_contents(scene_data):
    """What the survey of the reference's scene files recorded about them."""
    t, p, c = (scene_data[n] for n in SCENES)
    assert [len(m["indices"]) for m in t.meshes] == [1, 1]
    assert sum(len(m["indices"]) for m in p.meshes) == 26
    assert sum(len(m["indices"]) for m in c.meshes) == 72 and sum(len(m["positions"]) for m in c.meshes) == 128
    assert len(c.meshes) == 14 and len(c.materials) == 5 and len(c.lights) == 2
    assert all(l["type"] == 0 and np.allclose(l["diffuse"], 2.0) for l in c.lights)
    assert np.allclose(t.nodes[[n["name"] for n in t.nodes].index("Camera")]["transform"][12:15], [0, 0, 5], atol=1e-6)
    assert abs(c.camera["hfov"] - np.pi / 4) < 1e-6
    # meshes without a material share the default grey one
    assert np.allclose(p.materials[p.meshes[0]["material"]]["diffuse"], 0.6)


def test_fixture_json_roundtrip_is_bit_exact(acr, scene_data):
    d = scene_data["cornell-box-cycles"]
    e = acr.ImportData.from_json(d.to_json())
    for a, b in zip(d.meshes, e.meshes):
        assert np.array_equal(a["positions"].view(np.uint32), b["positions"].view(np.uint32))
        assert np.array_equal(a["normals"].view(np.uint32), b["normals"].view(np.uint32))
        assert np.array_equal(a["indices"], b["indices"])
