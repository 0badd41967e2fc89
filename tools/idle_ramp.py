#!/usr/bin/env python3
"""Per-launch times of Cornell 1080p frames (HIP events around every launch) in five bursts: the first launches of
the process, a burst right after a synchronize, one after 50 ms of idling, one more after a synchronize, and 200
launches.  Shows how long the GPU takes to come back up to speed after an idle period (a synchronize alone does not
slow the next launches down), which is what `bench.py --settle-ms` is for.   python tools/idle_ramp.py  (on the GPU box)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracer_amd as acr
data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-cycles.json"))
view, keep = data.pack()
scene = acr.Scene(import_view=view)
core = acr.HipCore(0)
core.upload_scene(scene.soa)
w, h = 1920, 1080
core.resize(w, h)
core.set_camera(scene.camera)
core.set_stream(torch.cuda.current_stream().cuda_stream)
def burst(n, seed):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for k, (a, b) in enumerate(evs):
        a.record(); core.render(w, h, seed=seed + k, frames=0, max_bounces=1); b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    ms = [a.elapsed_time(b) for a, b in evs]
    tot = evs[0][0].elapsed_time(evs[-1][1]) / n
    return ms, wall, tot
for name, n, pause in (("first", 80, 0), ("after sync", 40, 0), ("after 50 ms sleep", 40, 0.05), ("after sync again", 40, 0), ("200", 200, 0)):
    if pause: time.sleep(pause)
    ms, wall, tot = burst(n, 1000)
    print(name, "wall/frame %.4f gpu span/frame %.4f mean %.4f" % (wall, tot, np.mean(ms)))
    print("   ", " ".join("%.3f" % x for x in ms[:40]))
