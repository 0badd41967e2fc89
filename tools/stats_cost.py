#!/usr/bin/env python3
"""How much slower is a launch that counts rays (collect_stats) than one that does not?  Back-to-back frames, HIP events."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracer_amd as acr
data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-cycles.json"))
view, keep = data.pack()
scene = acr.Scene(import_view=view)
core = acr.HipCore(0)
core.upload_scene(scene.soa); core.set_camera(scene.camera)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st); core.set_stream(st.cuda_stream)
screen = torch.zeros((1080, 1920, 4), dtype=torch.float32, device=dev)
ptr = C.c_void_p(screen.data_ptr())
for mode in (0, 2, 0, 2):
    for k in range(30):
        core.render(1920, 1080, seed=k, frames=0, max_bounces=1, screen_dev=ptr, collect_stats=mode)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(100):
        core.render(1920, 1080, seed=100 + k, frames=0, max_bounces=1, screen_dev=ptr, collect_stats=mode)
    b.record(); torch.cuda.synchronize()
    print("collect_stats", mode, "ms/frame", round(a.elapsed_time(b) / 100, 4))
core.close()
