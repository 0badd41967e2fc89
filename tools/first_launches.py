#!/usr/bin/env python3
"""Times the first launches of a configuration one by one (HIP events on torch's stream): natural order, the
measuring launch, the launches with the measured order, the next measuring launches."""
import ctypes as C, os, sys
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
orders = []
for k, (a, b) in enumerate(evs):
    a.record()
    core.render(w, h, seed=100 + k, frames=0, max_bounces=1)
    b.record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in evs]
print(" ".join("%.3f" % x for x in ms))
