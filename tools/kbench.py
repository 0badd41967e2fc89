#!/usr/bin/env python3
"""Kernel experiment bench: times the render kernel of ONE build of libacr_hip.so and prints a digest of
the frame it produces, so that variants of the kernel can be compared for speed and for identical output.

    python tools/kbench.py [--lib path/to/libacr_hip.so] [--scene cornell-box-cycles] [--size 1920x1080]
                           [--bounces 1] [--steps 100] [--soup N --depth D]

One process per variant (the library is bound before anything else loads the default one).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--block-order", choices=["natural", "heavy-first", "reverse", "xcd-regions"], default="natural")
    ap.add_argument("--scene", default="cornell-box-cycles")
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--bounces", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--soup", type=int, default=0)
    ap.add_argument("--clouds", type=int, default=0, help="ten objects instancing three clouds of N, 2 N and N / 2 random triangles (tests/conftest.py: instanced_cloud_scene)")
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--edge", type=float, default=0.1)
    ap.add_argument("--extent", type=float, default=10.0)
    ap.add_argument("--aov", action="store_true", help="also write primitive ids and hit distances")
    ap.add_argument("--tag", default="")
    ap.add_argument("--nested", action="store_true", help="ACR_TRAVERSAL_NESTED: every lane makes its own object tests, also in frames with bounces")
    ap.add_argument("--nolights", action="store_true", help="drop the scene's lights (primary rays only)")
    ap.add_argument("--accumulate", choices=["identity", "sorted"], default=None,
                    help="time frames > 0 (running mean) through the identity pixel table or through the table sorted by traversal-path key")
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))

    from cudaraytracer_amd import _capi
    if args.lib:
        lib = C.CDLL(os.path.abspath(args.lib), mode=C.RTLD_GLOBAL)
        # an older build may lack entry points this benchmark does not call
        have = {k: v for k, v in _capi.HIP_SYMBOLS.items() if hasattr(lib, k)}
        _capi._hip = _capi._bind(lib, have)
    import cudaraytracer_amd as acr

    t0 = time.perf_counter()
    if args.clouds:
        from conftest import instanced_cloud_scene
        data = instanced_cloud_scene(acr, 3, tris=(args.clouds, 2 * args.clouds, args.clouds // 2))
        if args.nolights:
            data.lights = []
        view, keep = data.pack()
        scene = acr.Scene(import_view=view, mesh_bih_depth=args.depth)
    elif args.soup:
        from conftest import soup_data
        data = soup_data(acr, args.soup, seed=1, edge=args.edge, extent=args.extent)
        if args.nolights:
            data.lights = []
        view, keep = data.pack()
        scene = acr.Scene(import_view=view, mesh_bih_depth=args.depth)
    else:
        data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", args.scene + ".json"))
        if args.nolights:
            data.lights = []
        view, keep = data.pack()
        scene = acr.Scene(import_view=view)
    t_build = time.perf_counter() - t0
    core = acr.HipCore(0)
    t0 = time.perf_counter()
    core.upload_scene(scene.soa)
    t_upload = time.perf_counter() - t0
    core.set_camera(scene.camera)
    if args.nested:
        core.set_traversal(nested=True)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    core.set_stream(stream.cuda_stream)
    screen = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    prim = torch.zeros((h, w), dtype=torch.int32, device=dev) if args.aov else None
    ht = torch.zeros((h, w), dtype=torch.float32, device=dev) if args.aov else None
    kw = dict(screen_dev=C.c_void_p(screen.data_ptr()))
    if args.aov:
        kw.update(primid_dev=C.cast(prim.data_ptr(), _capi.c_u32_p), hit_t_dev=C.cast(ht.data_ptr(), _capi.c_float_p))

    core.render(w, h, seed=777, frames=0, max_bounces=args.bounces, collect_stats=True, **kw)
    st = core.read_stats()
    torch.cuda.synchronize()
    digest = hashlib.sha1(screen.cpu().numpy().tobytes()).hexdigest()[:16]
    if args.aov:
        digest += "/" + hashlib.sha1(prim.cpu().numpy().tobytes() + ht.cpu().numpy().tobytes()).hexdigest()[:16]
    rays = st.rays_primary + st.rays_shadow + st.rays_bounce

    if args.block_order != "natural":
        # per-block cost of this frame from a tile-list launch with 16 x 16 tiles, then the order
        bx, by = (w + 15) // 16, (h + 15) // 16
        nb = bx * by
        cyc = torch.zeros(nb, dtype=torch.int32, device=dev)
        tiles = torch.zeros((nb, 256, 4), dtype=torch.float32, device=dev)
        core.render(w, h, seed=777, frames=0, max_bounces=args.bounces, tile_ids=np.arange(nb, dtype=np.uint32), tile_w=16, tile_h=16,
                    screen_dev=C.c_void_p(tiles.data_ptr()), tile_cycles_dev=C.c_void_p(cyc.data_ptr()))
        core.sync()
        cost = cyc.cpu().numpy().astype(np.int64)
        if args.block_order == "xcd-regions":
            # Workgroups are dealt round-robin over the 8 XCDs, so table position i is rendered on XCD i % 8: give every
            # XCD a compact region of the image (a run of the blocks' Z-curve holding an eighth of the measured cost),
            # most expensive first inside it, so that the workgroups that share an L2 walk neighbouring parts of the trees.
            ys, xs = np.divmod(np.arange(nb), bx)
            def spread(v):
                v = v.astype(np.uint64)
                r = np.zeros_like(v)
                for bit in range(16):
                    r |= ((v >> np.uint64(bit)) & np.uint64(1)) << np.uint64(2 * bit)
                return r
            z = np.argsort(spread(xs) | (spread(ys) << np.uint64(1)), kind="stable")
            cum = np.cumsum(cost[z]) - cost[z] * 0.5
            grp = np.minimum((8 * cum / max(1, cost.sum())).astype(np.int64), 7)
            lists = [list(z[grp == g][np.argsort(-cost[z[grp == g]], kind="stable")]) for g in range(8)]
            heads = [0] * 8
            out = []
            while len(out) < nb:
                for g in range(8):
                    if len(out) >= nb:
                        break
                    if heads[g] >= len(lists[g]):
                        g = max(range(8), key=lambda k: len(lists[k]) - heads[k])  # this XCD's region is done: help the one with most left
                    out.append(lists[g][heads[g]])
                    heads[g] += 1
            order = np.array(out, dtype=np.int64)
            assert len(np.unique(order)) == nb
        else:
            order = np.argsort(-cost, kind="stable") if args.block_order == "heavy-first" else np.arange(nb)[::-1].copy()
        order_dev = torch.from_numpy(order.astype(np.int32)).to(dev)
        kw["block_order_dev"] = C.c_void_p(order_dev.data_ptr())
        core.render(w, h, seed=777, frames=0, max_bounces=args.bounces, **kw)
        torch.cuda.synchronize()
        digest += "|" + hashlib.sha1(screen.cpu().numpy().tobytes()).hexdigest()[:16]
        print("block cost: min %d median %d max %d" % (cost.min(), np.median(cost), cost.max()), file=sys.stderr)
    frames_of = lambda k: 0
    if args.accumulate:
        n = w * h
        keys = torch.zeros(n, dtype=torch.int64, device=dev)
        table = torch.arange(n, dtype=torch.int32, device=dev)
        kw["pixel_values_dev"] = C.c_void_p(table.data_ptr())
        core.render(w, h, seed=777, frames=0, max_bounces=args.bounces, path_keys_dev=C.c_void_p(keys.data_ptr()), **kw)
        if args.accumulate == "sorted":
            t0 = time.perf_counter()
            core.sort_pixel_map(C.c_void_p(keys.data_ptr()), C.c_void_p(table.data_ptr()), n)
            torch.cuda.synchronize()
            print("sort ms (first call, with allocation):", 1e3 * (time.perf_counter() - t0), file=sys.stderr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            core.sort_pixel_map(C.c_void_p(keys.data_ptr()), C.c_void_p(table.data_ptr()), n)
            e1.record()
            torch.cuda.synchronize()
            print("sort ms:", e0.elapsed_time(e1), file=sys.stderr)
        frames_of = lambda k: 1 + k
    for k in range(5):
        core.render(w, h, seed=1000 + k, frames=frames_of(k), max_bounces=args.bounces, **kw)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for k, (a, b) in enumerate(evs):
        a.record()
        core.render(w, h, seed=2000 + k, frames=frames_of(5 + k), max_bounces=args.bounces, **kw)
        b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in evs])
    out = dict(tag=args.tag or (args.lib or "default"), scene=("clouds%d/d%d" % (args.clouds, args.depth)) if args.clouds else ("soup%d/d%d" % (args.soup, args.depth)) if args.soup else args.scene,
               size=args.size, bounces=args.bounces, ms_mean=float(ms.mean()), ms_min=float(ms.min()), ms_median=float(np.median(ms)),
               rays=int(rays), grays_per_s=rays / (float(np.median(ms)) * 1e-3) / 1e9, digest=digest,
               six_plane_pixels=st.six_plane_pixels, wave_tasks=st.wave_tasks, shared_walk=st.shared_walk, lds_bytes=st.lds_bytes, lds_mode=st.scene_in_lds, build_s=round(t_build, 2), upload_s=round(t_upload, 2))
    print(json.dumps(out), flush=True)
    wdbg = getattr(_capi.hip_lib(), "acr_dev_walk_debug", None)
    if wdbg is not None:  # experiment builds (-DACR_WALK_DEBUG): how full the wave is in each part of the near-side-first walk
        v = (C.c_ulonglong * 16)()
        wdbg(v, 1)
        one = float(1 << 20)
        print("walk debug: walks %d | descent loop: %.1f passes per walk, lanes present %.3f | leaf trips: %.2f per walk, lanes present %.3f, "
              "%.2f of %s records tested per trip | rounds: %.1f per walk, lanes present %.3f"
              % (v[0], v[1] / max(v[0], 1), v[1] / max(64 * v[2] / one, 1e-9), v[3] / max(v[0], 1), v[3] / max(64 * v[4] / one, 1e-9),
                 v[5] / max(v[3], 1), "ACR_TRIS_PER_TRIP", v[6] / max(v[0], 1), v[6] / max(64 * v[7] / one, 1e-9)), file=sys.stderr)
        print("walk debug: pixels listed by the shared walk of several objects: world ray %d (NaN direction %d), local ray %d, visiting order %d" % (v[12], v[15], v[13], v[14]), file=sys.stderr)
        print("walk debug: share of walks with <8, 8-15, 16-31, ..., 256-511, 512+ rounds: " + " ".join("%.4f" % (v[8 + b] / max(v[0], 1)) for b in range(8)), file=sys.stderr)
    dbg = getattr(_capi.hip_lib(), "acr_dev_wave_debug", None)
    if dbg is not None:  # experiment builds (-DACR_WAVE_DEBUG): totals over all launches of this run
        v = (C.c_ulonglong * 16)()
        dbg(v, 1)
        print("wave debug: traces %d rays %d tasks %d overflow %d rounds %d | rays/trace %.1f tasks/ray %.2f rounds/trace %.2f"
              % (v[0], v[1], v[2], v[3], v[4], v[1] / max(v[0], 1), v[2] / max(v[1], 1), v[4] / max(v[0], 1)), file=sys.stderr)
        tot = max(v[9], 1)
        print("wave clocks: A %.3f B %.3f C %.3f winner %.3f rest %.3f of the waves' time; per trace A %.0f B %.0f C %.0f winner %.0f ticks"
              % (v[5] / tot, v[6] / tot, v[7] / tot, v[8] / tot, 1 - (v[5] + v[6] + v[7] + v[8]) / tot,
                 v[5] / max(v[0], 1), v[6] / max(v[0], 1), v[7] / max(v[0], 1), v[8] / max(v[0], 1)), file=sys.stderr)
    core.close()


if __name__ == "__main__":
    main()
