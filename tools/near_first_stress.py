#!/usr/bin/env python3
"""Stress of the near-side-first walk: many soups, seeds and bounce counts, GPU frames against the oracle's
(reference order); prints mismatching pixels and how many pixels went to the reference-order pass."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cudaraytracer_amd as acr
from cudaraytracer_amd import _capi
import oracle

def soup(n, seed, edge, extent):
    lib = _capi.host_lib()
    h = lib.acr_host_import_soup(n, seed, C.c_float(edge), C.c_float(extent))
    try:
        return acr.ImportData.from_view(lib.acr_host_import_view(h).contents)
    finally:
        lib.acr_host_import_destroy(h)

core = acr.HipCore(0)
w, h = 640, 360
core.resize(w, h)
total_px = bad = listed = hits = frames = 0
for k, (n, depth, edge, extent) in enumerate([(50000, 13, 0.2, 2.0), (200000, 16, 0.15, 3.0), (500000, 18, 0.12, 4.0),
                                              (1000000, 19, 0.1, 5.0), (300000, 17, 0.4, 3.0)]):
    data = soup(n, 10 + k, edge, extent)
    view, keep = data.pack()
    scene = acr.Scene(import_view=view, mesh_bih_depth=depth)
    o = oracle.OracleScene(view, 6, depth)
    core.upload_scene(scene.soa); core.set_camera(scene.camera)
    for seed in range(6):
        for bounces in (1, 3):
            core.render(w, h, seed=1000 * k + seed, frames=0, max_bounces=bounces, collect_stats=1)
            core.sync()
            st = core.read_stats()
            assert st.near_first == 1
            ref = o.render(w, h, seed=1000 * k + seed, frames=0, max_bounces=bounces, threads=16)
            img, pid = core.read_framebuffer(), core.read_primid()
            bad += int((img.view(np.uint32) != ref["screen"].view(np.uint32)).any(axis=-1).sum()) + int((pid != ref["prim"]).sum())
            listed += st.six_plane_pixels
            hits += int((pid != 0xFFFFFFFF).sum())
            total_px += w * h
            frames += 1
    o.close()
    print("soup %d depth %d: frames so far %d, mismatching pixels %d, listed pixels %d, primary hits %d" % (n, depth, frames, bad, listed, hits), flush=True)
print("frames %d pixels %d mismatching %d listed %d" % (frames, total_px, bad, listed))
core.close()
sys.exit(1 if bad else 0)
