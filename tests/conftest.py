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
