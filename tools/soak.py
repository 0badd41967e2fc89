#!/usr/bin/env python3
"""Soak run: many frames back to back with camera moves in between (block-order cadence, double-buffered list
counters, accumulation), every 2500th frame compared with the oracle bit for bit."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cudaraytracer_amd as acr
from cudaraytracer_amd import camera_path
import oracle

data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-cycles.json"))
view, keep = data.pack()
scene = acr.Scene(import_view=view)
o = oracle.OracleScene(view)
core = acr.HipCore(0)
core.upload_scene(scene.soa)
w, h = 1920, 1080
core.resize(w, h)
cams = camera_path.flythrough(scene.camera, 40)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
t0 = time.time()
bad = checks = 0
frames = 0
for k in range(n):
    if k % 100 == 0:
        cam = cams[(k // 100) % len(cams)]
        core.set_camera(cam)
        frames = 0
    bounces = 1 if (k // 100) % 3 else 2
    core.render(w, h, seed=k, frames=frames, max_bounces=bounces)
    frames += 1
    if k % 2500 == 2499:
        core.sync()
        img = core.read_framebuffer()
        # the oracle repeats this camera's accumulation run
        first = k - (frames - 1)
        oc = oracle.OrcCamera(); C.memmove(C.byref(oc), C.byref(cam), C.sizeof(oc))
        screen = None
        for f in range(frames):
            screen = o.render(w, h, seed=first + f, frames=f, max_bounces=bounces, camera=oc, screen=screen, aov=False, threads=16)["screen"]
        d = int((img.view(np.uint32) != screen.view(np.uint32)).sum())
        bad += d; checks += 1
        print("frame %d: accumulation of %d frames, differing words %d" % (k, frames, d), flush=True)
core.sync()
print("%d frames in %.1f s, %d checks, differing words %d" % (n, time.time() - t0, checks, bad))
core.close()
sys.exit(1 if bad else 0)
