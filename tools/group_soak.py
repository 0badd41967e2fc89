#!/usr/bin/env python3
"""Soak of acr_hip_group_* on one GPU: a group of several contexts and one single context render the same long
sequence of frames (running means with the camera still, a camera move every so often, frames that overwrite in
between, gathers at odd moments, the split re-dealt every few frames); at every checkpoint the assembled frame must
equal the single context's bit for bit.  What it is after is ordering mistakes between the members' render and copy
streams, which show up rarely and only under load.

    python tools/group_soak.py [--members 4] [--frames 3000] [--size 1920x1080] [--bounces 1]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracer_amd as acr  # noqa: E402
from cudaraytracer_amd import camera_path  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=4)
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--bounces", type=int, default=1)
    ap.add_argument("--rebalance", type=int, default=3)
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-cycles.json"))
    view, keep = data.pack()
    scene = acr.Scene(import_view=view)
    cams = camera_path.flythrough(scene.camera, 24)
    g = acr.HipGroup([0] * args.members)
    one = acr.HipCore(0)
    for x in (g, one):
        x.upload_scene(scene.soa)
        x.set_camera(scene.camera)
        x.resize(w, h)
    g.set_balance(rebalance_every=args.rebalance)
    rng = np.random.default_rng(1)
    still, cam_no, checks, splits = 0, 0, 0, set()
    t0 = time.perf_counter()
    for k in range(args.frames):
        r = rng.random()
        if r < 0.02:  # the camera moves
            cam_no = (cam_no + 1) % len(cams)
            g.set_camera(cams[cam_no])
            one.set_camera(cams[cam_no])
            still = 0
        elif r < 0.10:  # a frame that overwrites although the camera stands still
            still = 0
        gather = rng.random() < 0.3
        g.render(seed=1000 + k, frames=still, max_bounces=args.bounces, gather=gather)
        one.render(w, h, seed=1000 + k, frames=still, max_bounces=args.bounces)
        still += 1
        splits.add(tuple(g.split()[0]))
        if rng.random() < 0.03 or k == args.frames - 1:
            a = g.read_framebuffer()
            one.sync()
            b = one.read_framebuffer()
            if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                rows = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=(1, 2)))[0]
                print("MISMATCH at frame %d (still %d): %d rows differ, first %s; split %s" % (k, still, len(rows), rows[:8].tolist(), g.split()[0]))
                sys.exit(1)
            checks += 1
    print("group soak ok: %d members, %d frames of %dx%d, %d checkpoints bit-identical, %d different splits, %.1f s"
          % (args.members, args.frames, w, h, checks, len(splits), time.perf_counter() - t0))
    g.close()
    one.close()


if __name__ == "__main__":
    main()
