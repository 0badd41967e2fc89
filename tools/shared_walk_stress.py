#!/usr/bin/env python3
"""Stress of the walk shared by the lanes of a wave on scenes of several objects: clouds of random triangles instanced
under random transforms (tests/conftest.py: instanced_cloud_scene), several sizes, seeds and bounce counts; GPU frames
against the oracle's and against the one-lane-per-ray kernel's.  Prints mismatching pixels and listed pixels."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cudaraytracer_amd as acr
import oracle
from conftest import instanced_cloud_scene

core = acr.HipCore(0)
w, h = 480, 270
core.resize(w, h)
bad = bad1 = listed = frames = 0
SCENES = [(20000, 13), (100000, 16), (400000, 18)]
if len(sys.argv) > 1:  # more scenes: <count> random (triangles, depth, objects) combinations
    rng = np.random.default_rng(7)
    SCENES = [(int(rng.integers(3000, 150000)), int(rng.integers(10, 18))) for _ in range(int(sys.argv[1]))]
for k, (n, depth) in enumerate(SCENES):
    data = instanced_cloud_scene(acr, 20 + k, n_objects=3 + (5 * k) % 17, tris=(n, 2 * n, n // 2))
    view, keep = data.pack()
    scene = acr.Scene(import_view=view, mesh_bih_depth=depth)
    o = oracle.OracleScene(view, 6, depth)
    core.upload_scene(scene.soa); core.set_camera(scene.camera)
    for seed in range(3):
        for bounces in (1, 2, 4):
            ref = o.render(w, h, seed=100 * k + seed, frames=0, max_bounces=bounces, threads=16)
            for nested in (False, True):
                core.set_traversal(nested=nested)
                core.render(w, h, seed=100 * k + seed, frames=0, max_bounces=bounces, collect_stats=1)
                core.sync()
                st = core.read_stats()
                assert st.shared_walk == (0 if nested else 1)
                img, pid = core.read_framebuffer(), core.read_primid()
                m = int((img.view(np.uint32) != ref["screen"].view(np.uint32)).any(axis=-1).sum()) + int((pid != ref["prim"]).sum())
                if nested: bad1 += m
                else:
                    bad += m
                    listed += st.six_plane_pixels
            core.set_traversal()
            frames += 1
            print("clouds %d depth %d seed %d bounces %d: mismatching pixels so far shared %d one-lane %d, listed %d" % (n, depth, seed, bounces, bad, bad1, listed), flush=True)
    o.close()
print("frames %d mismatching shared %d one-lane %d listed %d" % (frames, bad, bad1, listed))
core.close()
sys.exit(1 if bad or bad1 else 0)
