#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame on the Cornell box at 1920x1080,
primary + shadow rays (max_bounces = 1), on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5]

N = 1 runs in-process.  N > 1 expects to be launched once per GPU by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment): the image is cut into tiles, each rank renders a contiguous run
of tiles, rank 0 gathers them over RCCL and assembles the frame.

A step is one frame with a fresh seed.  --mode still (the default) is the
reference's own protocol, the one its only instrumentation times ("Time for 200
frames", renderer.cu:399-406): the camera stands still and frame k is averaged
into the framebuffer with frames = k (renderer.cu:375-387); across GPUs every
rank keeps the running mean of its own bands and the pieces travel to GPU 0
when the frame is read, once, at the end of the timed region (inside it).
--mode moving renders every step with frames = 0, as after a camera move
(renderer.cu:165), and across GPUs assembles every frame on GPU 0; c5, the
flythrough, always runs that way.  Rank 0 prints one JSON line.  On one GPU the W
warm-up steps are preceded by --settle-ms (default 30) of untimed frames that
bring the GPU's clocks up from the idle state the scene set-up left them in;
the line reports their number as settle_frames.  On one GPU the line
also carries the roofline of the render kernel (timed with HIP events on the
stream it is launched on) and a CPU baseline: the oracle (this repository's
CPU restatement of the reference algorithm) timed on the host cores on a
bounded sample of the same workload, which doubles as a parity check.

Workloads (the default, c2, is the one the headline metric is quoted on):
  c2  cornell-box-cycles 1920x1080, primary + shadow rays
  c3  cornell-box-cycles 3840x2160, max_bounces 4
  c4  procedural soup of 10 M triangles (mesh BIH depth 22), 1920x1080, primary + shadow rays
  c5  c2 along a 120-frame camera flythrough (one camera per step, assignment rebalanced every frame)
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
from cudaraytracer_amd import _capi  # noqa: E402
from cudaraytracer_amd import camera_path  # noqa: E402
from cudaraytracer_amd import distributed as D  # noqa: E402

TILE_H = 16                  # multi-GPU: full-width bands of 16 rows (a rank's run of bands is contiguous in the frame)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, spec
FP32_VECTOR_PEAK_TFLOPS = 157.3
BYTES_PER_PIXEL = 16         # one float4 framebuffer write per pixel
NODE_BYTES, TRIANGLE_BYTES = 20, 36   # reference record sizes: BIH node (bih.h:92-98), three vertex positions

WORKLOADS = {
    "c2": dict(scene="cornell-box-cycles", width=1920, height=1080, bounces=1,
               text="C2: cornell-box-cycles 1920x1080, primary + shadow rays (max_bounces 1)"),
    "c3": dict(scene="cornell-box-cycles", width=3840, height=2160, bounces=4,
               text="C3: cornell-box-cycles 3840x2160, max_bounces 4"),
    "c4": dict(soup=10_000_000, depth=22, edge=0.1, extent=10.0, width=1920, height=1080, bounces=1,
               text="C4: procedural soup, 10 M triangles (centres in [-10,10]^3, edges in [-0.1,0.1]^3, seed 1), mesh BIH depth 22, "
                    "1920x1080, primary + shadow rays"),
    "c5": dict(scene="cornell-box-cycles", width=1920, height=1080, bounces=1, flythrough=120,
               text="C5: cornell-box-cycles 1920x1080, primary + shadow rays, 120-frame camera flythrough (+-30 degrees), one camera per step"),
}


def algorithmic_flops(c, lights):
    """Arithmetic the reference algorithm performs for the counted work: slab test 12 flop, mesh box test
    3 divides on top, object entry 56, object hit exit 65, triangle test 53, direct-light term 30 per light
    and shaded hit, primary-ray generation 60."""
    return (12 * c["slab_tests"] + 3 * c["mesh_visits"] + 56 * c["object_visits"] + 65 * c["object_hits"] +
            53 * c["triangle_tests"] + 30 * lights * c["primary_hits"] + 60 * c["rays_primary"])


def soup_import(n, edge, extent):
    lib = _capi.host_lib()
    h = lib.acr_host_import_soup(n, 1, C.c_float(edge), C.c_float(extent))
    try:
        return acr.ImportData.from_view(lib.acr_host_import_view(h).contents)
    finally:
        lib.acr_host_import_destroy(h)


class Workload:
    """One workload set up on this rank: scene built, uploaded, camera(s) ready."""

    def __init__(self, name, local_rank, block_order_every=None):
        self.name = name
        self.wl = wl = WORKLOADS[name]
        self.width, self.height, self.bounces = wl["width"], wl["height"], wl["bounces"]
        if "soup" in wl:
            data = soup_import(wl["soup"], wl["edge"], wl["extent"])
            self.view, self.keep = data.pack()
            self.scene = acr.Scene(import_view=self.view, mesh_bih_depth=wl["depth"])
            self.scene_depths = (6, wl["depth"])
            self.data_text = "procedural scene from the host library's generator (seed 1), one jitter seed per frame"
        else:
            data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", wl["scene"] + ".json"))
            self.view, self.keep = data.pack()
            self.scene = acr.Scene(import_view=self.view)
            self.scene_depths = (6, 6)
            self.data_text = "reference scene %s (decoded fixture in tests/golden/scenes), one jitter seed per frame" % wl["scene"]
        self.num_lights = self.scene.soa.num_lights
        self.core = acr.HipCore(local_rank)
        self.core.upload_scene(self.scene.soa)
        self.core.set_camera(self.scene.camera)
        if block_order_every is not None:
            self.core.set_block_order(block_order_every)
        self.cameras = camera_path.flythrough(self.scene.camera, wl["flythrough"]) if "flythrough" in wl else None

    def camera_of(self, step):
        return self.cameras[step % len(self.cameras)] if self.cameras else self.scene.camera

    def close(self):
        self.core.close()
        self.scene.close()


SEED0 = 12345


def run_multi(w, device, world, rank, steps, warmup, rehearsal, barrier, still):
    """The frame split over the ranks of the process group: `warmup` untimed frames, `steps` timed ones between
    barriers, then the rays of the timed frames (counted afterwards) and, on rank 0, the last assembled frame
    against one full-frame render of the same seed and camera.  Returns a dict; elapsed is this rank's."""
    core, WIDTH, HEIGHT, MAX_BOUNCES = w.core, w.width, w.height, w.bounces
    step_no = [0]

    def frame_args():
        if w.cameras:
            core.set_camera(w.camera_of(step_no[0]))
        # still: the renderer passes the frame count of its running mean; this entry is what a moving camera renders with
        a = dict(seed=SEED0 + step_no[0], frames=0, max_bounces=MAX_BOUNCES)
        step_no[0] += 1
        return a
    TILE_W = WIDTH
    render_bands = D.hip_render_tiles(core, WIDTH, HEIGHT, TILE_W, TILE_H, frame_args, cost_kind="cycles")
    dr = D.DistributedRenderer(WIDTH, HEIGHT, render_bands, device,
                               tile_w=TILE_W, tile_h=TILE_H, scatter=D.hip_scatter(core, WIDTH, HEIGHT, TILE_W, TILE_H),
                               rebalance_every=1 if w.cameras else 16, stage_through_host=rehearsal, accumulate=still)
    # Shares of the frame from measured speed (untimed, before the warm-up): a rank that has to send its piece
    # over one xGMI link finishes fewer bands per second than rank 0, which only renders.  Three times: the
    # later calls fit each phase's fixed cost and cost per band through two measurements at different shares.
    cal = None
    if not still and os.environ.get("ACR_BENCH_EQUAL_SHARES") != "1":
        dr.render_frame()
        dr.finish()
        for _ in range(3):
            cal = dr.calibrate(frames=3)
        if rank == 0 and cal:
            print("[shares %s] render ms %s send ms %s shares %s" % (w.name, [round(1e3 * float(x), 3) for x in cal["render_s"]],
                  [round(1e3 * float(x), 3) for x in cal["send_s"]], [round(float(x), 3) for x in cal["weights"]]), file=sys.stderr)
    if still:
        dr.camera_moved()  # the running mean starts here
    first_frame = step_no[0]
    if still:
        # untimed frames before the warm-up, the counterpart of the single-GPU run's settle frames: three rebalance
        # periods, so that the split has followed the measured costs and every pair of ranks that hands a band's
        # running mean to another has exchanged something once (a pair's first exchange sets its channel up)
        for _ in range(3 * 16):
            dr.render_frame()
    for _ in range(warmup):
        dr.render_frame()
    dr.finish()
    barrier()
    first_timed = step_no[0]
    t0 = time.perf_counter()
    for _ in range(steps):
        dr.render_frame()
    dr.finish()  # still: this is where the pieces travel to GPU 0, once
    barrier()
    elapsed = time.perf_counter() - t0
    # functional check, outside the timed region: the last assembled frame against one full-frame render
    same = None
    if rank == 0:
        last = step_no[0] - 1
        full = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=device)
        core.set_camera(w.camera_of(last))
        # still: the same running mean on one GPU, frame by frame from the first warm-up frame on
        for k in (range(first_frame, last + 1) if still else (last, )):
            core.render(WIDTH, HEIGHT, seed=SEED0 + k, frames=k - first_frame if still else 0, max_bounces=MAX_BOUNCES,
                        screen_dev=C.c_void_p(full.data_ptr()))
        torch.cuda.synchronize()
        same = bool(torch.equal(full.view(torch.int32), dr.frame.view(torch.int32)))
        print(f"[check {w.name}] assembled frame equals single-GPU frame: {same}; split {[int(x) for x in dr.plan.begin]}", file=sys.stderr)
        if not same:
            rows = torch.nonzero((full.view(torch.int32) != dr.frame.view(torch.int32)).any(dim=2).any(dim=1)).flatten().tolist()
            print(f"[check {w.name}] differing rows: {len(rows)}, first {rows[:8]}, last {rows[-8:]}; split used by the frame {dr.last_split}", file=sys.stderr)
        del full
    # rays of the timed frames, counted afterwards: the same seeds and cameras once more with per-tile ray
    # counts switched on (the total does not depend on how the frame is split)
    ray_sums = []
    dr.settle()
    if still:
        dr.camera_moved()
    dr.rebalance_every = 0
    render_bands.cost_kind = "rays"
    step_no[0] = first_timed
    for _ in range(steps):
        dr.render_frame(want_cost=True)
        ray_sums.append(dr.rays_last_frame())
    dr.finish()
    local = torch.stack(ray_sums).double().sum().reshape(1)
    if rehearsal:
        local = local.cpu()
    dist.all_reduce(local)  # each rank counted the rays of its own tiles
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return dict(elapsed=float(t.item()), rays=int(local.item()), same=same,
                shares=None if dr.plan.weights is None else [round(float(x), 4) for x in dr.plan.weights])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--block-order-every", type=int, default=None, help="longest interval between block-order measurements (library default 64; 0 = natural order)")
    ap.add_argument("--settle-ms", type=float, default=30.0, help="one GPU: render untimed frames for this long before the warm-up, so that the "
                    "GPU's clocks have come up from idle (0 = none)")
    ap.add_argument("--mode", choices=("still", "moving"), default="still",
                    help="still: the camera stands still, frame k is averaged into the framebuffer with frames = k (the reference's "
                         "render loop); moving: every step renders with frames = 0, as after a camera move")
    ap.add_argument("--lib", default=None, help="another build of libacr_hip.so (make variant ...), e.g. the one with fused multiply-adds that "
                    "measures what bit-exactness costs; the line then says which library it ran")
    ap.add_argument("--no-also", action="store_true", help="several GPUs: do not run the compute-heavy workloads (c3, c4) after the headline one")
    args = ap.parse_args()
    if args.lib:
        # bound before anything else loads the default build (the host library resolves its acr_hip_* calls to this one)
        _capi._hip = _capi._bind(C.CDLL(os.path.abspath(args.lib), mode=C.RTLD_GLOBAL), _capi.HIP_SYMBOLS)
    wl = WORKLOADS[args.workload]
    WIDTH, HEIGHT, MAX_BOUNCES = wl["width"], wl["height"], wl["bounces"]
    heavy = args.workload in ("c3", "c4")
    still = args.mode == "still" and "flythrough" not in wl  # a flythrough moves the camera on every frame
    if args.steps is None:
        args.steps = 30 if heavy else 200
    if args.warmup is None:
        args.warmup = 3 if heavy else 20

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

    w = Workload(args.workload, local_rank, args.block_order_every)
    scene, view, core, cameras, num_lights, scene_depths, data_text = w.scene, w.view, w.core, w.cameras, w.num_lights, w.scene_depths, w.data_text
    camera_of = w.camera_of
    # one explicit stream for kernels, torch events and RCCL ordering
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    core.set_stream(stream.cuda_stream)

    seed0 = SEED0
    step_no = [0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = None
    extra = {}
    multi = None
    also = []
    if world == 1:
        screen = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=device)
        ptr = C.c_void_p(screen.data_ptr())

        def step(mode_still=still, stats=0):
            if cameras:
                core.set_camera(camera_of(step_no[0]))
            core.render(WIDTH, HEIGHT, seed=seed0 + step_no[0], frames=step_no[0] if mode_still else 0, max_bounces=MAX_BOUNCES, screen_dev=ptr,
                        collect_stats=stats)
            step_no[0] += 1

        # The GPU idles while the host builds and uploads the scene, and after an idle period its clocks take some
        # 20 ms of rendering to come back up (Cornell 1080p frames launched after 50 ms of idling: 0.160 ms each at
        # first, 0.154 after 40 frames, 0.146 after 120; a synchronize alone does not cause it).  A renderer in use
        # is never in that state, so untimed frames are rendered for SETTLE_MS before the W warm-up steps; the
        # line reports how many (settle_frames).  They use seeds of their own and leave step_no alone.
        settle_frames = 0
        t_settle = time.perf_counter()
        while args.settle_ms > 0 and 1e3 * (time.perf_counter() - t_settle) < args.settle_ms:
            for k in range(10):
                core.render(WIDTH, HEIGHT, seed=seed0 - 1 - settle_frames, frames=0, max_bounces=MAX_BOUNCES, screen_dev=ptr)
                settle_frames += 1
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        barrier()
        first_timed = step_no[0]
        # the timed region: K asynchronous launches back to back, nothing else on the stream, one barrier at the end
        t0 = time.perf_counter()
        for k in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        last_timed = step_no[0]

        # ---- further passes over the same kind of step, none of them part of `value` ----
        extra = {}
        n_ev = max(args.steps, 10)
        # (a) every step between two HIP events on the stream the kernels run on: per-step times, their mean and median
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for a, b in evs:
            a.record()
            step()
            b.record()
        barrier()
        per_step = [a.elapsed_time(b) for a, b in evs]
        extra["step_ms"] = float(np.mean(per_step))
        extra["ms_per_step_median"] = float(np.median(per_step))
        extra["step_ms_samples"] = n_ev
        # (b) the reference's protocol: the host waits for every frame before it starts the next (renderer.cu:421-422)
        t1 = time.perf_counter()
        for k in range(args.steps):
            step()
            torch.cuda.synchronize()
        extra["ms_per_frame_synced"] = 1e3 * (time.perf_counter() - t1) / args.steps
        # (c) the render kernel's own time: the library's event pair around its launch (acr_stats.kernel_ms), no counting code in the kernel
        own = []
        for k in range(max(10, min(args.steps, 50))):
            # each sample is the last of ten launches queued back to back: reading the events waits for the stream, and a
            # kernel launched onto an idle GPU is not the kernel of a running renderer
            for _ in range(9):
                step()
            step(stats=3)
            own.append(core.read_stats().kernel_ms)
        kernel_ms = float(np.mean(own))
        extra["kernel_ms_samples"] = len(own)
        extra["kernel_ms_median"] = float(np.median(own))
        # (d) the other mode, timed like the main region
        if not cameras:
            for _ in range(max(args.warmup, 40)):  # the passes above waited for the stream after every frame
                step(mode_still=not still)
            barrier()
            t2 = time.perf_counter()
            for k in range(args.steps):
                step(mode_still=not still)
            barrier()
            extra["ms_per_step_moving" if still else "ms_per_step_still"] = 1e3 * (time.perf_counter() - t2) / args.steps
        step_no[0] = last_timed
        # rays of the timed frames, counted afterwards so that the timed launches carry no instrumentation: the
        # same seeds and cameras back to back, the counters accumulating on the device, one read at the end
        for k in range(args.steps):
            if cameras:
                core.set_camera(camera_of(first_timed + k))
            core.render(WIDTH, HEIGHT, seed=seed0 + first_timed + k, frames=first_timed + k if still else 0, max_bounces=MAX_BOUNCES, screen_dev=ptr,
                        collect_stats=1 if k == 0 else 2)
        st = core.read_stats()
        rays = st.rays_primary + st.rays_shadow + st.rays_bounce
        lds_bytes, lds_mode, interval, near_first, wave_tasks = st.lds_bytes, st.scene_in_lds, st.interval_traversal, st.near_first, st.wave_tasks
        shared_walk = st.shared_walk
    else:
        multi = run_multi(w, device, world, rank, args.steps, args.warmup, rehearsal, barrier, still)
        elapsed, rays = multi["elapsed"], multi["rays"]
        if multi["same"] is False:
            raise SystemExit("assembled frame differs from the full-frame render")
        # The headline workload is link-bound beyond one GPU (33 MB per frame into GPU 0 against a 0.16 ms render);
        # the compute-heavy workloads are the ones a tile split can speed up: a short run of each, reported beside it.
        if not args.no_also:
            for name in ("c3", "c4"):
                if name == args.workload:
                    continue
                rec = {"workload": WORKLOADS[name]["text"]}
                tb = time.perf_counter()
                # Building the scene (the 10 M-triangle soup most of all) is where one rank alone can fail, and it has no
                # collective in it: the ranks agree on whether all of them got through before any enters run_multi,
                # whose collectives a missing rank would leave the others waiting in.
                w2, err = None, None
                try:
                    w2 = Workload(name, local_rank, args.block_order_every)
                    w2.core.set_stream(stream.cuda_stream)
                except Exception as e:
                    err = "%s: %s" % (type(e).__name__, e)
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device="cpu" if rehearsal else device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    k2 = 20
                    m2 = run_multi(w2, device, world, rank, k2, 3, rehearsal, barrier, still)
                    rec.update({"steps": k2, "ms_per_frame": 1e3 * m2["elapsed"] / k2, "Mrays_per_s": m2["rays"] / m2["elapsed"] / 1e6,
                                "rays_per_frame": m2["rays"] / k2, "shares": m2["shares"], "assembled_equals_single_gpu": m2["same"],
                                "setup_and_run_s": round(time.perf_counter() - tb, 1)})
                else:
                    rec["error"] = err or "another rank could not set this workload up"
                if w2 is not None:
                    w2.close()
                also.append(rec)

    # (several GPUs: run_multi has already taken the maximum of the ranks' times)

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        out = {
            "metric": "Mrays/s + ms/frame, Cornell box 1080p, at 1/2/4/8 MI355X",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_frames": settle_frames if world == 1 else None,  # untimed frames before the warm-up that bring the GPU's clocks up from idle (--settle-ms)
            "ms_per_step": ms,
            "ms_per_frame": ms,
            "mode": "still" if still else "moving",
            **({"library": os.path.relpath(os.path.abspath(args.lib), ROOT)} if args.lib else {}),
            **(extra if world == 1 else {}),
            "rays_per_frame": rays / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": data_text,
            "config": {"workload": wl["text"] + (", camera still: frame k averaged in with frames = k" if still else ", frames=0 every step"), "width": WIDTH, "height": HEIGHT, "max_bounces": MAX_BOUNCES,
                       "parallelism": "single GPU, full frame" if world == 1 else
                       (f"full-width bands of {TILE_H} rows over {world} GPUs, contiguous runs balanced by measured cost; every rank keeps the running mean of its own bands, "
                                        "the pieces travel to GPU 0 over RCCL once, when the frame is read at the end of the timed region (inside it)" if still else
                                        f"full-width bands of {TILE_H} rows over {world} GPUs, contiguous runs balanced by measured cost and by each rank's measured speed (render + send), received in place on rank 0 over RCCL, two frames in flight")},
        }
        if multi is not None:
            if multi["shares"] is not None:
                out["config"]["shares"] = multi["shares"]
            out["assembled_equals_single_gpu"] = multi["same"]
            if also:
                out["also"] = also
        if world == 1:
            tf_ = lambda v: "true" if v else "false"
            kernel = "renderWaveKernel" if wave_tasks else ("renderSoloKernel<%s, %s, false>" % (tf_(lds_mode >= 1), tf_(MAX_BOUNCES == 1))) if shared_walk else "renderKernel<%s, %s, %s, false, %s>" % (
                tf_(lds_mode >= 1), tf_(lds_mode == 2), tf_(interval), tf_(interval and MAX_BOUNCES == 1))
            frame_bytes = BYTES_PER_PIXEL * WIDTH * HEIGHT
            # The roofline that binds: with the scene resident in LDS a frame's only HBM traffic is the framebuffer
            # and the kernel is bound by fp32 vector issue (divergence); with trees and triangles in HBM it is the
            # memory system.  `roofline` is the binding one, the other is reported beside it.
            roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "kernel": kernel, "kernel_ms": kernel_ms, "step_ms": extra["step_ms"], "algorithmic_bytes_per_launch": frame_bytes + (frame_bytes if still else 0),
                    "note": "LDS per workgroup %d B, staging mode %d (2 = whole scene resident in LDS, 1 = small tables only)" % (lds_bytes, lds_mode)}
            # HBM bytes per launch measured with rocprofv3 counter passes of this same command, when a summary is committed
            tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                roof["traffic"] = tj.get("hbm_bytes_per_launch")
                roof["traffic_source"] = tj.get("source")
            out["roofline"] = roof
            valu = None
            if not args.no_cpu_baseline:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle  # CPU checker and reported baseline; never part of the timed GPU path
                o = oracle.OracleScene(view, *scene_depths)
                threads = min(16, os.cpu_count() or 1)
                cam0 = None
                if cameras:
                    cam0 = oracle.OrcCamera()
                    C.memmove(C.byref(cam0), C.byref(camera_of(0)), C.sizeof(cam0))
                    core.set_camera(camera_of(0))
                # bounded sample: whole frames while they are cheap, else a band of rows in the middle of the frame
                t_probe = None
                for _ in range(2):  # the first pass also pages the scene in; the second one is the estimate
                    tc = time.perf_counter()
                    o.render(WIDTH, HEIGHT, seed=seed0, frames=0, max_bounces=MAX_BOUNCES, threads=threads, aov=False, camera=cam0,
                             row_begin=HEIGHT // 2 - 16, row_end=HEIGHT // 2 + 16)
                    t_probe = time.perf_counter() - tc
                rows_affordable = int(32 * args.cpu_seconds / max(t_probe, 1e-4))
                whole = rows_affordable >= HEIGHT
                frames_done, cpu_rays, t_cpu, counters, ref = 0, 0, 0.0, None, None
                if whole:
                    r0, r1 = 0, HEIGHT
                    while t_cpu < args.cpu_seconds:
                        tc = time.perf_counter()
                        r = o.render(WIDTH, HEIGHT, seed=seed0 + frames_done, frames=0, max_bounces=MAX_BOUNCES, threads=threads,
                                     aov=(frames_done == 0), camera=cam0)
                        t_cpu += time.perf_counter() - tc
                        c = r["counters"]
                        cpu_rays += c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]
                        if frames_done == 0:
                            counters, ref = c, r
                        frames_done += 1
                    sample = f"{frames_done} frames of the same workload ({t_cpu:.1f} s)"
                else:
                    nrows = max(4, min(HEIGHT, rows_affordable) // 4 * 4)
                    r0 = (HEIGHT - nrows) // 2
                    r1 = r0 + nrows
                    tc = time.perf_counter()
                    ref = o.render(WIDTH, HEIGHT, seed=seed0, frames=0, max_bounces=MAX_BOUNCES, threads=threads, aov=True, camera=cam0,
                                   row_begin=r0, row_end=r1)
                    t_cpu = time.perf_counter() - tc
                    counters = ref["counters"]
                    cpu_rays = counters["rays_primary"] + counters["rays_shadow"] + counters["rays_bounce"]
                    sample = f"rows {r0}..{r1 - 1} of one {WIDTH}x{HEIGHT} frame of the same workload ({t_cpu:.1f} s)"
                tc = time.perf_counter()
                rr = max(1, (r1 - r0) // 16)
                m0 = (r0 + r1 - rr) // 2  # rows from the middle of the sample: the top rows of a frame are mostly background
                c1 = o.render(WIDTH, HEIGHT, seed=seed0, frames=0, max_bounces=MAX_BOUNCES, threads=1, aov=False, camera=cam0,
                              row_begin=m0, row_end=m0 + rr)["counters"]
                t1 = time.perf_counter() - tc
                out["cpu_baseline"] = {"value": cpu_rays / t_cpu / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                                       "sample": sample, "value_1core": (c1["rays_primary"] + c1["rays_shadow"] + c1["rays_bounce"]) / t1 / 1e6,
                                       "host_cpus": os.cpu_count()}
                # algorithmic work per ray from the oracle's counters of the sample; when the kernel walks the mesh
                # trees near side first, from the oracle walking them in that order over the same sample
                rays0 = counters["rays_primary"] + counters["rays_shadow"] + counters["rays_bounce"]
                if near_first:
                    o.set_mesh_order(True)
                    counters = o.render(WIDTH, HEIGHT, seed=seed0, frames=0, max_bounces=MAX_BOUNCES, threads=threads, aov=False,
                                        camera=cam0, row_begin=r0, row_end=r1)["counters"]
                    o.set_mesh_order(False)
                    roof["note"] += "; mesh trees walked near side first: work counters are the oracle's for that order"
                fl = algorithmic_flops(counters, num_lights)
                tf = fl / rays0 * (rays / args.steps) / (kernel_ms * 1e-3) / 1e12
                valu = {"bound": "valu_fp32", "achieved": tf, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / FP32_VECTOR_PEAK_TFLOPS, "traffic": None, "algorithmic_flops_per_ray": fl / rays0,
                        "kernel": kernel, "kernel_ms": kernel_ms, "step_ms": extra["step_ms"],
                        "frac_of_no_fma_ceiling": tf / (FP32_VECTOR_PEAK_TFLOPS / 2),
                        "frac_of_unpacked_no_fma_ceiling": tf / (FP32_VECTOR_PEAK_TFLOPS / 4),
                        "note": "peak counts an FMA as two flops and two packed operations per lane and cycle (v_pk_fma_f32); these "
                                "kernels are built with -ffp-contract=off (the checker's arithmetic), so no FMA issues and %.1f TFLOP/s "
                                "is the most they could reach with every operation packed, %.1f with the one-operation-per-lane "
                                "instructions a divergent traversal issues"
                                % (FP32_VECTOR_PEAK_TFLOPS / 2, FP32_VECTOR_PEAK_TFLOPS / 4)}
                if lds_mode != 2:
                    # nodes and triangles are read from HBM: count them at the reference's record sizes
                    per_ray = (NODE_BYTES * counters["node_visits"] + TRIANGLE_BYTES * counters["triangle_tests"]) / rays0
                    roof["algorithmic_bytes_per_ray"] = per_ray
                    roof["algorithmic_bytes_per_launch"] += per_ray * rays / args.steps
                # parity: GPU vs oracle on the sampled rows, every pixel
                core.resize(WIDTH, HEIGHT)
                core.render(WIDTH, HEIGHT, seed=seed0, frames=0, max_bounces=MAX_BOUNCES)
                core.sync()
                img, pid, ht = core.read_framebuffer()[r0:r1], core.read_primid()[r0:r1], core.read_hit_t()[r0:r1]
                rs, rp, rt = ref["screen"][r0:r1], ref["prim"].reshape(HEIGHT, WIDTH)[r0:r1], ref["t"].reshape(HEIGHT, WIDTH)[r0:r1]
                diff = img[..., :3].astype(np.float64) - rs[..., :3]
                mse = float(np.mean(diff ** 2))
                rel = np.abs(diff) / np.maximum(np.abs(rs[..., :3]), 1e-3)  # relative to the value, to 1e-3 at least (radiance is O(1))
                # a differing primitive id is an epsilon-tie when the two hits lie within 1e-5 of each other along the ray
                # (two surfaces a rounding apart), anything else is a different answer
                mism = pid != rp
                with np.errstate(invalid="ignore", over="ignore"):
                    close = np.abs(ht.astype(np.float64) - rt) <= 1e-5 * np.maximum(np.abs(rt.astype(np.float64)), 1e-6)
                out["parity"] = {"rows": [r0, r1], "pixels": int(pid.size), "primid_mismatches": int(mism.sum()),
                                 "primid_mismatches_epsilon_ties": int((mism & close).sum()),
                                 "primid_mismatches_other": int((mism & ~close).sum()),
                                 "rgba_bit_mismatches": int((img.view(np.uint32) != rs.view(np.uint32)).sum()),
                                 "pixels_beyond_1e-4_rel": int((rel.max(axis=2) > 1e-4).sum()),
                                 "max_rel_err": float(np.max(rel)),
                                 "psnr_db": None if mse == 0 else float(10 * np.log10(1.0 / mse)),
                                 "psnr_note": "null means identical images (infinite PSNR); relative errors are taken against max(|reference|, 1e-3)"}
            if lds_mode != 2 and roof["traffic"]:
                # what HBM can have served is what the counters saw leave it; the algorithmic bytes (record sizes of the
                # reference x visits) are mostly served by L2 and are reported as such
                roof["achieved"] = roof["traffic"] / (kernel_ms * 1e-3) / 1e9
                roof["algorithmic_l2_served"] = {"achieved": roof["algorithmic_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9, "unit": "GB/s",
                                                 "frac_of_hbm_peak": roof["algorithmic_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            else:
                roof["achieved"] = roof["algorithmic_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            if lds_mode == 2 and valu is not None:
                valu["traffic"] = roof["traffic"]
                out["roofline"], out["roofline_hbm"] = valu, roof
            elif valu is not None:
                out["roofline_valu"] = valu
        print(json.dumps(out), flush=True)

    core.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
