import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def acr():
    import cudaraytracer_amd
    return cudaraytracer_amd


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


def fixture_path(name):
    return os.path.join(HERE, "golden", "scenes", name + ".json")


SCENES = ("triangles", "plane", "cornell-box-cycles")


@pytest.fixture(scope="session")
def scene_data(acr):
    """name -> ImportData of the committed scene fixtures"""
    return {n: acr.ImportData.load(fixture_path(n)) for n in SCENES}


class Built:
    """One import scene built by both sides: product Scene and oracle scene."""

    def __init__(self, acr, orc, data, scene_depth=6, mesh_depth=6):
        self.view, self.keep = data.pack()
        self.scene = acr.Scene(import_view=self.view, scene_bih_depth=scene_depth, mesh_bih_depth=mesh_depth)
        self.oracle = orc.OracleScene(self.view, scene_depth, mesh_depth)


@pytest.fixture(scope="session")
def built(acr, orc, scene_data):
    return {n: Built(acr, orc, d) for n, d in scene_data.items()}


def soup_data(acr, ntri, seed=1, edge=0.05, extent=1.0):
    """Procedural triangle soup as ImportData (through the product's generator)."""
    import ctypes as C
    from cudaraytracer_amd import _capi
    lib = _capi.host_lib()
    h = lib.acr_host_import_soup(ntri, seed, C.c_float(edge), C.c_float(extent))
    try:
        return acr.ImportData.from_view(lib.acr_host_import_view(h).contents)
    finally:
        lib.acr_host_import_destroy(h)


def random_instanced_scene(acr, seed, n_objects=40):
    """A scene the shipped ones do not resemble: many objects instancing a few small meshes under random
    rotations, uniform and non-uniform scales, nested parents, one or two meshes per object, diffuse, mirror-like
    and translucent materials, three point lights.  Returns ImportData."""
    import numpy as np
    rng = np.random.default_rng(seed)
    d = acr.ImportData()

    def mesh(positions, faces, material):
        p = np.asarray(positions, np.float32)
        n = p / np.maximum(np.linalg.norm(p, axis=1, keepdims=True), 1e-6)  # smooth normals: direction from the mesh origin
        d.meshes.append(dict(positions=p, normals=n.astype(np.float32), indices=np.asarray(faces, np.uint32), material=material))

    cube_p = [(x, y, z) for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]
    cube_f = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    mesh(cube_p, cube_f, 0)
    mesh([(1, 1, 1), (-1, -1, 1), (-1, 1, -1), (1, -1, -1)], [(0, 1, 2), (0, 3, 1), (0, 2, 3), (1, 3, 2)], 1)
    mesh([(-1, -1, 0.1), (1, -1, 0.1), (1, 1, 0.1), (-1, 1, 0.1)], [(0, 1, 2), (0, 2, 3)], 2)
    sp = rng.uniform(-1, 1, (90, 3))
    mesh(sp, np.arange(90).reshape(30, 3), 3)
    mesh(np.asarray(cube_p, np.float32) * (0.5, 1.5, 0.7), cube_f, 2)

    d.materials = [dict(diffuse=[0.8, 0.6, 0.3], ambient=[0.02, 0.02, 0.02], specular=[0.2, 0.2, 0.2], refracti=0.0),
                   dict(diffuse=[0.1, 0.1, 0.1], ambient=[0, 0, 0], specular=[0.9, 0.9, 0.9], refracti=0.0),
                   dict(diffuse=[0.3, 0.5, 0.9], ambient=[0, 0.01, 0.02], specular=[0.3, 0.3, 0.3], refracti=1.5),
                   dict(diffuse=[0.4, 0.9, 0.4], ambient=[0, 0, 0], specular=[0.0, 0.0, 0.0], refracti=0.0)]

    def trs(t, axis, angle, s):
        axis = axis / np.linalg.norm(axis)
        x, y, z = axis
        c, si = np.cos(angle), np.sin(angle)
        r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * si, x * z * (1 - c) + y * si],
                      [y * x * (1 - c) + z * si, c + y * y * (1 - c), y * z * (1 - c) - x * si],
                      [z * x * (1 - c) - y * si, z * y * (1 - c) + x * si, c + z * z * (1 - c)]])
        m = np.eye(4)
        m[:3, :3] = r * np.asarray(s)[None, :]
        m[:3, 3] = t
        return [float(np.float32(v)) for v in m.T.reshape(-1)]  # column-major

    ident = trs([0, 0, 0], np.array([0, 0, 1.0]), 0.0, [1, 1, 1])
    d.nodes.append(dict(name="<Root>", parent=-1, transform=ident, meshes=[]))
    for k in range(n_objects):
        parent = 0 if (k < 4 or rng.random() < 0.7) else int(rng.integers(1, k + 1))
        scale = [float(rng.uniform(0.3, 0.9))] * 3 if rng.random() < 0.6 else list(rng.uniform(0.3, 1.2, 3))
        spread = 5.0 if parent == 0 else 1.5
        t = rng.uniform(-spread, spread, 3)
        meshes = [int(rng.integers(0, len(d.meshes)))]
        if rng.random() < 0.3:
            meshes.append(int(rng.integers(0, len(d.meshes))))
        d.nodes.append(dict(name=f"Obj{k}", parent=parent, transform=trs(t, rng.normal(size=3), rng.uniform(0, 6.28), scale), meshes=meshes))
    lights = [([6.0, 5.0, 8.0], 6.0), ([-7.0, 2.0, 6.0], 5.0), ([0.5, -6.0, 4.0], 4.0)]
    for k, (pos, e) in enumerate(lights):
        d.nodes.append(dict(name=f"L{k}", parent=0, transform=trs(pos, np.array([0, 0, 1.0]), 0.0, [1, 1, 1]), meshes=[]))
        d.lights.append(dict(name=f"L{k}", type=0, position=[0, 0, 0], direction=[0, 0, -1], ambient=[e, e, e], diffuse=[e, e, e],
                             specular=[e, e, e], att=[0.0, 1.0, 0.0], cones=[6.2831855, 6.2831855]))
    d.nodes.append(dict(name="Cam", parent=0, transform=trs([0.0, 0.0, 11.0], np.array([0, 0, 1.0]), 0.0, [1, 1, 1]), meshes=[]))
    d.camera = dict(name="Cam", aspect=0.0, hfov=0.7853982, position=[0, 0, 0], up=[0, 1, 0], look_at=[0, 0, -1])
    return d


def instanced_cloud_scene(acr, seed, n_objects=10, tris=(30000, 50000, 8000)):
    """Several objects, each instancing one or two of a few LARGE meshes (clouds of random triangles: their trees do
    not fit LDS), under random rotations, uniform and non-uniform scales and nested parents; diffuse, mirror-like and
    translucent materials, two point lights.  Returns ImportData."""
    import numpy as np
    rng = np.random.default_rng(seed)
    d = acr.ImportData()
    for k, n in enumerate(tris):
        c = rng.uniform(-1, 1, (n, 1, 3))
        p = (c + rng.uniform(-0.12, 0.12, (n, 3, 3))).astype(np.float32).reshape(-1, 3)
        nrm = rng.normal(size=p.shape)
        nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
        d.meshes.append(dict(positions=p, normals=nrm, indices=np.arange(len(p), dtype=np.uint32).reshape(-1, 3), material=k % 3))
    d.materials = [dict(diffuse=[0.8, 0.6, 0.3], ambient=[0.02, 0.02, 0.02], specular=[0.1, 0.1, 0.1], refracti=0.0),
                   dict(diffuse=[0.1, 0.1, 0.1], ambient=[0, 0, 0], specular=[0.9, 0.9, 0.9], refracti=0.0),
                   dict(diffuse=[0.3, 0.5, 0.9], ambient=[0, 0.01, 0.02], specular=[0.3, 0.3, 0.3], refracti=1.5)]

    def trs(t, axis, angle, s):
        axis = axis / np.linalg.norm(axis)
        x, y, z = axis
        c, si = np.cos(angle), np.sin(angle)
        r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * si, x * z * (1 - c) + y * si],
                      [y * x * (1 - c) + z * si, c + y * y * (1 - c), y * z * (1 - c) - x * si],
                      [z * x * (1 - c) - y * si, z * y * (1 - c) + x * si, c + z * z * (1 - c)]])
        m = np.eye(4)
        m[:3, :3] = r * np.asarray(s)[None, :]
        m[:3, 3] = t
        return [float(np.float32(v)) for v in m.T.reshape(-1)]  # column-major

    ident = trs([0, 0, 0], np.array([0, 0, 1.0]), 0.0, [1, 1, 1])
    d.nodes.append(dict(name="<Root>", parent=-1, transform=ident, meshes=[]))
    for k in range(n_objects):
        parent = 0 if (k < 3 or rng.random() < 0.7) else int(rng.integers(1, k + 1))
        scale = [float(rng.uniform(0.6, 1.4))] * 3 if rng.random() < 0.6 else list(rng.uniform(0.5, 1.5, 3))
        spread = 3.5 if parent == 0 else 1.0
        t = rng.uniform(-spread, spread, 3)
        meshes = [int(rng.integers(0, len(d.meshes)))]
        if rng.random() < 0.3:
            meshes.append(int(rng.integers(0, len(d.meshes))))
        d.nodes.append(dict(name=f"Obj{k}", parent=parent, transform=trs(t, rng.normal(size=3), rng.uniform(0, 6.28), scale), meshes=meshes))
    for k, (pos, e) in enumerate([([6.0, 5.0, 8.0], 6.0), ([-7.0, 2.0, 6.0], 5.0)]):
        d.nodes.append(dict(name=f"L{k}", parent=0, transform=trs(pos, np.array([0, 0, 1.0]), 0.0, [1, 1, 1]), meshes=[]))
        d.lights.append(dict(name=f"L{k}", type=0, position=[0, 0, 0], direction=[0, 0, -1], ambient=[e, e, e], diffuse=[e, e, e],
                             specular=[e, e, e], att=[0.0, 1.0, 0.0], cones=[6.2831855, 6.2831855]))
    d.nodes.append(dict(name="Cam", parent=0, transform=trs([0.0, 0.0, 12.0], np.array([0, 0, 1.0]), 0.0, [1, 1, 1]), meshes=[]))
    d.camera = dict(name="Cam", aspect=0.0, hfov=0.7853982, position=[0, 0, 0], up=[0, 1, 0], look_at=[0, 0, -1])
    return d


def duplicate_triangle_scene(acr, n, seed=5, spread=0.3):
    """n random triangles, each present twice (the copies in reverse order) in ONE mesh: every hit is an exact tie
    between two items of the mesh tree, usually in different leaves.  One point light and the camera at (0, 0, 5)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    data = acr.ImportData()
    c = rng.uniform(-1, 1, (n, 1, 3))
    tri = (c + rng.uniform(-spread, spread, (n, 3, 3))).astype(np.float32)
    tri = np.concatenate([tri, tri[::-1]], axis=0)
    p = tri.reshape(-1, 3)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (len(p), 1))
    data.meshes.append(dict(positions=p, normals=nrm, indices=np.arange(len(p), dtype=np.uint32).reshape(-1, 3), material=0))
    data.materials = [dict(diffuse=[0.8, 0.8, 0.8], ambient=[0, 0, 0], specular=[0, 0, 0], refracti=0.0)]
    ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    at5 = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 5, 1]  # column-major translation to (0, 0, 5)
    data.nodes = [dict(name="<Root>", parent=-1, transform=ident, meshes=[]),
                  dict(name="Obj", parent=0, transform=ident, meshes=[0]),
                  dict(name="L0", parent=0, transform=at5, meshes=[]),
                  dict(name="Cam", parent=0, transform=at5, meshes=[])]
    data.lights = [dict(name="L0", type=0, position=[0, 0, 0], direction=[0, 0, -1], ambient=[3, 3, 3], diffuse=[3, 3, 3],
                        specular=[3, 3, 3], att=[0, 1, 0], cones=[6.28, 6.28])]
    data.camera = dict(name="Cam", aspect=0.0, hfov=0.6, position=[0, 0, 0], up=[0, 1, 0], look_at=[0, 0, -1])
    return data
