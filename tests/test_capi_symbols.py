"""The shared libraries load on a machine without a GPU and export every
symbol their headers declare.  No compute entry point is called here."""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def declared(header, prefix):
    with open(os.path.join(ROOT, "include", header)) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    from cudaraytracer_amd import _capi
    lib = _capi.hip_lib()
    names = declared("acr_hip.h", "acr_hip_")
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_capi.HIP_SYMBOLS) == names  # the Python binding covers exactly the header


def test_host_library_exports_every_declared_symbol():
    from cudaraytracer_amd import _capi
    lib = _capi.host_lib()
    names = declared("acr_host.h", "acr_host_")
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_capi.HOST_SYMBOLS) == names


def test_struct_sizes_match_the_c_headers():
    """ctypes mirrors vs sizeof in C, checked by compiling a tiny program against the headers."""
    import subprocess
    import tempfile
    from cudaraytracer_amd import _capi
    src = r'''
#include <stdio.h>
#include "acr_host.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(acr_scene_soa), sizeof(acr_camera), sizeof(acr_frame), sizeof(acr_stats),
         sizeof(acr_import_node), sizeof(acr_import_mesh), sizeof(acr_import_material), sizeof(acr_import_light),
         sizeof(acr_import_camera), sizeof(acr_import_scene));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "s.c"), "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "s"), os.path.join(d, "s.c")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    mirrors = [_capi.SceneSoA, _capi.Camera, _capi.Frame, _capi.Stats, _capi.ImportNode, _capi.ImportMesh,
               _capi.ImportMaterial, _capi.ImportLight, _capi.ImportCamera, _capi.ImportScene]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_oracle_import_struct_layout_equals_product(orc):
    """The oracle declares its own import structs; a test feeds one packed buffer to both sides."""
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "acr_host.h"
#include "acr_oracle.h"
int main(void) {
  int ok = sizeof(acr_import_scene) == sizeof(orc_import_scene) && sizeof(acr_import_node) == sizeof(orc_import_node)
    && sizeof(acr_import_mesh) == sizeof(orc_import_mesh) && sizeof(acr_import_material) == sizeof(orc_import_material)
    && sizeof(acr_import_light) == sizeof(orc_import_light) && sizeof(acr_import_camera) == sizeof(orc_import_camera)
    && offsetof(acr_import_scene, camera) == offsetof(orc_import_scene, camera)
    && offsetof(acr_import_scene, indices) == offsetof(orc_import_scene, indices)
    && offsetof(acr_import_light, inner_cone) == offsetof(orc_import_light, inner_cone)
    && offsetof(acr_import_node, name) == offsetof(orc_import_node, name);
  printf("%d\n", ok); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "s.c"), "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), "-o",
                               os.path.join(d, "s"), os.path.join(d, "s.c")])
        assert subprocess.check_output([os.path.join(d, "s")]).strip() == b"1"


def test_product_does_not_reference_the_oracle():
    """Nothing under the package or include/ may import, include or link the oracle."""
    bad = []
    for base in ("cudaraytracer_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".h", ".cpp", ".hip", "Makefile")):
                    with open(os.path.join(dp, fn), errors="ignore") as f:
                        t = f.read()
                    if re.search(r"acr_oracle|liboracle|import oracle|orc_", t):
                        bad.append(os.path.join(dp, fn))
    assert bad == []
    import subprocess
    for so in ("libacr_hip.so", "libacr_host.so"):
        out = subprocess.check_output(["ldd", os.path.join(ROOT, "cudaraytracer_amd", so)]).decode()
        assert "oracle" not in out
