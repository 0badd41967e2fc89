#!/usr/bin/env python3
"""Host-side time to ISSUE one frame of a rank's share (an eighth of a 1080p Cornell frame: 9 bands), layer by layer:
the C ABI call alone, HipCore.render, the band callable of distributed.py, DistributedRenderer.render_frame.
The GPU work per call is about 20 us, so with the host slower than that the loop time is the host's.

    python tools/host_issue.py [bands]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracer_amd as acr  # noqa: E402
from cudaraytracer_amd import _capi  # noqa: E402
from cudaraytracer_amd import distributed as D  # noqa: E402

W, H = 1920, 1080
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 9
data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-cycles.json"))
view, keep = data.pack()
scene = acr.Scene(import_view=view)
core = acr.HipCore(0)
core.upload_scene(scene.soa)
core.set_camera(scene.camera)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
core.set_stream(stream.cuda_stream)
frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
ids = np.arange(20, 20 + NB, dtype=np.uint32)
out = frame.view(-1, 4)[20 * 16 * W:(20 + NB) * 16 * W]
N = 2000


def timed(fn, label):
    for k in range(50):
        fn(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        fn(k)
    t_issue = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / N * 1e6
    print("%-44s issue %6.1f us/frame   with the GPU drained %6.1f us/frame" % (label, t_issue, t_all))


# (c) the C ABI alone
f = _capi.Frame()
f.width, f.height, f.max_bounces = W, H, 1
f.tile_w, f.tile_h, f.num_tiles, f.tile_ids = W, 16, NB, ids.ctypes.data_as(_capi.c_u32_p)
lib = _capi.hip_lib()
ptr = C.c_void_p(out.data_ptr())


def raw(k):
    f.seed, f.frames = k, k
    lib.acr_hip_render(core._h, C.byref(f), ptr, None, None)


timed(raw, "acr_hip_render (ctypes, prebuilt acr_frame)")
timed(lambda k: core.render(W, H, seed=k, frames=k, max_bounces=1, tile_ids=ids, tile_w=W, tile_h=16, screen_dev=ptr), "HipCore.render")
fn = D.hip_render_tiles(core, W, H, W, 16, lambda: dict(seed=1, frames=0, max_bounces=1))
timed(lambda k: fn(ids, out, None, frames=k), "distributed.hip_render_tiles callable")


class OneOfEight(D.DistributedRenderer):
    """the split an eighth of the frame's bands would give this rank, without a process group"""
    pass


dr = D.DistributedRenderer(W, H, fn, dev, tile_w=W, tile_h=16, rebalance_every=16, accumulate=True)
dr.plan.begin = np.array([20, 20 + NB], dtype=np.uint32)  # this rank's run
dr.plan.update = lambda cost: None
timed(lambda k: dr.render_frame(), "DistributedRenderer.render_frame (accumulate)")
core.close()
