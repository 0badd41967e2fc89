#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame on the Cornell box at 1920x1080,
primary + shadow rays (max_bounces = 1), on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in-process.  N > 1 expects to be launched once per GPU by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment): the image is cut into tiles, each rank renders a contiguous run
of tiles, rank 0 gathers them over RCCL and assembles the frame.

A step is one frame: a fresh seed, frames = 0 (what the reference renders
after every camera move).  Rank 0 prints one JSON line.  On one GPU the line
also carries the roofline of the render kernel (timed with HIP events on the
stream it is launched on) and a CPU baseline: the oracle (this repository's
CPU restatement of the reference algorithm) timed on the host cores on a
bounded sample of the same workload, which doubles as a parity check of one
frame.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cudaraytracer_amd as acr  # noqa: E402
from cudaraytracer_amd import distributed as D  # noqa: E402

WIDTH, HEIGHT, MAX_BOUNCES = 1920, 1080, 1
SCENE = "cornell-box-cycles"
TILE_W, TILE_H = 64, 16
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, spec
FP32_VECTOR_PEAK_TFLOPS = 157.3
BYTES_PER_PIXEL = 16         # one float4 framebuffer write per pixel: the path's only compulsory HBM traffic


def algorithmic_flops(c):
    """Arithmetic the reference algorithm performs for the counted work: slab test 12 flop, mesh box test
    3 divides on top, object entry 56, object hit exit 65, triangle test 53, direct-light term 30 per light
    and shaded hit, primary-ray generation 60."""
    lights = 2
    return (12 * c["slab_tests"] + 3 * c["mesh_visits"] + 56 * c["object_visits"] + 65 * c["object_hits"] +
            53 * c["triangle_tests"] + 30 * lights * c["primary_hits"] + 60 * c["rays_primary"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (WORLD_SIZE is {world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # ACR_BENCH_REHEARSAL=1 runs the multi-rank path on ONE GPU (every rank on device 0, gloo, tiles staged
    # through the host): a functional rehearsal for boxes with a single GPU, never a measurement.
    rehearsal = os.environ.get("ACR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", SCENE + ".json"))
    view, keep = data.pack()
    scene = acr.Scene(import_view=view)
    core = acr.HipCore(local_rank)
    core.upload_scene(scene.soa)
    core.set_camera(scene.camera)
    # one explicit stream for kernels, torch events and RCCL ordering
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    core.set_stream(stream.cuda_stream)

    seed0 = 12345
    step_no = [0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = None
    if world == 1:
        screen = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=device)
        ptr = C.c_void_p(screen.data_ptr())

        def step():
            core.render(WIDTH, HEIGHT, seed=seed0 + step_no[0], frames=0, max_bounces=MAX_BOUNCES, screen_dev=ptr)
            step_no[0] += 1

        for _ in range(args.warmup):
            step()
        barrier()
        first_timed = step_no[0]
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for a, b in evs:
            a.record()
            step()
            b.record()
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        # rays of the timed frames, counted afterwards so that the timed launches carry no instrumentation
        rays = 0
        for k in range(args.steps):
            core.render(WIDTH, HEIGHT, seed=seed0 + first_timed + k, frames=0, max_bounces=MAX_BOUNCES, screen_dev=ptr, collect_stats=True)
            st = core.read_stats()
            rays += st.rays_primary + st.rays_shadow + st.rays_bounce
        lds_bytes, lds_mode = st.lds_bytes, st.scene_in_lds
    else:
        def frame_args():
            a = dict(seed=seed0 + step_no[0], frames=0, max_bounces=MAX_BOUNCES)
            step_no[0] += 1
            return a
        dr = D.DistributedRenderer(WIDTH, HEIGHT, D.hip_render_tiles(core, WIDTH, HEIGHT, TILE_W, TILE_H, frame_args), device,
                                   tile_w=TILE_W, tile_h=TILE_H, scatter=D.hip_scatter(core, WIDTH, HEIGHT, TILE_W, TILE_H),
                                   rebalance_every=16, stage_through_host=rehearsal)
        ray_sums = []
        for _ in range(args.warmup):
            dr.render_frame()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dr.render_frame()
            ray_sums.append(dr.rays_last_frame())  # device scalar, no sync
        barrier()
        elapsed = time.perf_counter() - t0
        local = torch.stack(ray_sums).double().sum().reshape(1)
        if rehearsal:
            local = local.cpu()
        dist.all_reduce(local)  # each rank counted the rays of its own tiles
        rays = int(local.item())
        lds_bytes = lds_mode = None
        if rank == 0 and os.environ.get("ACR_BENCH_CHECK") == "1":
            # functional check of the assembled frame against a single full-frame render of the same seed
            last_seed = seed0 + step_no[0] - 1
            full = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=device)
            core.render(WIDTH, HEIGHT, seed=last_seed, frames=0, max_bounces=MAX_BOUNCES, screen_dev=C.c_void_p(full.data_ptr()))
            torch.cuda.synchronize()
            same = bool(torch.equal(full.view(torch.int32), dr.frame.view(torch.int32)))
            print(f"[check] assembled frame equals single-GPU frame: {same}; split {[int(x) for x in dr.plan.begin]}", file=sys.stderr)
            if not same:
                raise SystemExit("assembled frame differs from the full-frame render")

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        out = {
            "metric": "Mrays/s + ms/frame, Cornell box 1080p, at 1/2/4/8 MI355X",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "ms_per_frame": ms,
            "rays_per_frame": rays / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "reference scene cornell-box-cycles (decoded fixture in tests/golden/scenes), one jitter seed per frame",
            "config": {"workload": "C2: cornell-box-cycles 1920x1080, primary + shadow rays (max_bounces 1), frames=0 every step",
                       "width": WIDTH, "height": HEIGHT, "max_bounces": MAX_BOUNCES,
                       "parallelism": "single GPU, full frame" if world == 1 else f"{TILE_W}x{TILE_H} tiles over {world} GPUs, cost-balanced Z-curve runs, RCCL gather to rank 0"},
        }
        if world == 1:
            achieved = BYTES_PER_PIXEL * WIDTH * HEIGHT / (kernel_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "renderKernel<true,true>", "kernel_ms": kernel_ms,
                               "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * WIDTH * HEIGHT,
                               "note": "scene staged in LDS (%d B per workgroup, mode %d); the kernel is VALU/latency-bound, see roofline_valu" % (lds_bytes, lds_mode)}
            if not args.no_cpu_baseline:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle  # CPU checker and reported baseline; never part of the timed GPU path
                o = oracle.OracleScene(view)
                threads = min(16, os.cpu_count() or 1)
                frames, cpu_rays, t_cpu, counters = 0, 0, 0.0, None
                while t_cpu < args.cpu_seconds:
                    tc = time.perf_counter()
                    r = o.render(WIDTH, HEIGHT, seed=seed0 + frames, frames=0, max_bounces=MAX_BOUNCES, threads=threads, aov=(frames == 0))
                    t_cpu += time.perf_counter() - tc
                    c = r["counters"]
                    cpu_rays += c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]
                    if frames == 0:
                        counters, ref = c, r
                    frames += 1
                tc = time.perf_counter()
                r1 = o.render(WIDTH, HEIGHT // 4, seed=seed0, frames=0, max_bounces=MAX_BOUNCES, threads=1, aov=False)["counters"]
                t1 = time.perf_counter() - tc
                out["cpu_baseline"] = {"value": cpu_rays / t_cpu / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                                       "sample": f"{frames} frames of the same workload ({t_cpu:.1f} s)",
                                       "ms_per_frame": 1e3 * t_cpu / frames,
                                       "value_1core": (r1["rays_primary"] + r1["rays_shadow"]) / t1 / 1e6,
                                       "host_cpus": os.cpu_count()}
                # VALU-side roofline from the oracle's work counters of frame seed0
                fl = algorithmic_flops(counters)
                rays0 = counters["rays_primary"] + counters["rays_shadow"] + counters["rays_bounce"]
                tf = fl / (kernel_ms * 1e-3) / 1e12
                out["roofline_valu"] = {"bound": "valu_fp32", "achieved": tf, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": tf / FP32_VECTOR_PEAK_TFLOPS, "algorithmic_flops_per_ray": fl / rays0}
                # parity of one frame: GPU vs oracle, every pixel
                core.resize(WIDTH, HEIGHT)
                core.render(WIDTH, HEIGHT, seed=seed0, frames=0, max_bounces=MAX_BOUNCES)
                core.sync()
                img, pid = core.read_framebuffer(), core.read_primid()
                diff = img[..., :3].astype(np.float64) - ref["screen"][..., :3]
                mse = float(np.mean(diff ** 2))
                denom = np.maximum(np.abs(ref["screen"][..., :3]), 1e-6)
                out["parity"] = {"primid_mismatches": int((pid != ref["prim"]).sum()),
                                 "rgba_bit_mismatches": int((img.view(np.uint32) != ref["screen"].view(np.uint32)).sum()),
                                 "max_rel_err": float(np.max(np.abs(diff) / denom)),
                                 "psnr_db": None if mse == 0 else float(10 * np.log10(1.0 / mse)),
                                 "psnr_note": "null means identical images (infinite PSNR)"}
        print(json.dumps(out), flush=True)

    core.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
