"""The multi-GPU frame assembly, rehearsed with two CPU processes over gloo:
each rank renders its tiles (the oracle stands in for the GPU core here),
rank 0 gathers and assembles, and the result must equal one full-frame render.
Covers both layouts (Z-ordered tiles with a scatter pass; full-width bands
received in place with two frames in flight), the uniform split, cost feedback
and a camera-move reset."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, TW, TH = 176, 100, 32, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, layout):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cudaraytracer_amd as acr
    import oracle
    from cudaraytracer_amd import distributed as D

    data = acr.ImportData.load(os.path.join(HERE, "golden", "scenes", "cornell-box-cycles.json"))
    view, keep = data.pack()
    o = oracle.OracleScene(view)
    state = dict(seed=21)
    rendered = []

    tw = W if layout == "bands" else TW

    def render_tiles(tile_ids, out, cost):
        # stand-in for the GPU core: render the frame on the CPU, cut out this rank's tiles in the compact
        # layout [tile][row][column] the kernel writes (`out` may end before the last tile's last rows)
        if state.get("slow_rank", -1) == rank:
            import time
            time.sleep(0.5)
        r = o.render(W, H, seed=state["seed"], frames=0, max_bounces=2, threads=2)
        img = r["screen"]
        tiles_x = (W + tw - 1) // tw
        buf = np.zeros((len(tile_ids), TH, tw, 4), np.float32)
        c = np.zeros(len(tile_ids), np.float32)
        for k, t in enumerate(tile_ids):
            x0, y0 = (int(t) % tiles_x) * tw, (int(t) // tiles_x) * TH
            w, h = min(tw, W - x0), min(TH, H - y0)
            buf[k, :h, :w] = img[y0:y0 + h, x0:x0 + w]
            c[k] = 1.0 + float((r["prim"][y0:y0 + h, x0:x0 + w] != 0xFFFFFFFF).sum())
        flat = out.view(-1, 4)
        flat.copy_(torch.from_numpy(buf.reshape(-1, 4)[:flat.shape[0]]))
        if cost is not None:
            cost.copy_(torch.from_numpy(c.astype(np.int32)))
        rendered.append(len(tile_ids))

    dr = D.DistributedRenderer(W, H, render_tiles, torch.device("cpu"), tile_w=tw, tile_h=TH)
    assert dr.bands == (layout == "bands")
    results = {}
    splits = []
    # frames 0 .. lag: uniform split until the first frame's costs arrive (they decide the split `cost_lag` frames
    # later); then the camera moves: uniform again
    seeds = list(range(21, 21 + dr.cost_lag + 2))
    for step, seed in enumerate(seeds):
        state["seed"] = seed
        if step == len(seeds) - 1:
            dr.camera_moved()  # back to the uniform split
        dr.render_frame()
        splits.append(dr.last_split)
        frame = dr.finish()
        if rank == 0:
            ref = o.render(W, H, seed=seed, frames=0, max_bounces=2, threads=2)["screen"]
            results[f"equal{step}"] = bool(np.array_equal(frame.numpy().view(np.uint32), ref.view(np.uint32)))
    results["steps"] = len(seeds)
    results["lag"] = dr.cost_lag
    # every rank derived the same splits
    every = [None] * world
    dist.all_gather_object(every, splits)
    assert all(e == every[0] for e in every)
    # frames started back to back: the exchange of one frame is still in flight while the next one renders
    for seed in (31, 32, 33):
        state["seed"] = seed
        dr.render_frame()
    frame = dr.finish()
    if rank == 0:
        ref = o.render(W, H, seed=33, frames=0, max_bounces=2, threads=2)["screen"]
        results["equal_pipelined"] = bool(np.array_equal(frame.numpy().view(np.uint32), ref.view(np.uint32)))
        if dr.depth > 1:
            prev = dr.frames[(dr.frame_no - 2) % dr.depth]
            ref = o.render(W, H, seed=32, frames=0, max_bounces=2, threads=2)["screen"]
            results["equal_pipelined"] &= bool(np.array_equal(prev.numpy().view(np.uint32), ref.view(np.uint32)))
    if layout == "bands":
        # shares from measured speed: rank 1 is made slow, must get the smaller share, every rank derives the
        # same split, and the frame assembled under it is still the full-frame render
        state["slow_rank"] = 1
        cal = dr.calibrate(frames=2)
        state["slow_rank"] = -1
        begins = [None] * world
        dist.all_gather_object(begins, [int(x) for x in dr.plan.begin])
        assert all(b == begins[0] for b in begins)
        state["seed"] = 41
        dr.camera_moved()  # costs back to uniform; the shares stay
        dr.render_frame()
        frame = dr.finish()
        if rank == 0:
            ref = o.render(W, H, seed=41, frames=0, max_bounces=2, threads=2)["screen"]
            results["equal_weighted"] = bool(np.array_equal(frame.numpy().view(np.uint32), ref.view(np.uint32)))
            results["weights"] = [float(x) for x in cal["weights"]]
            results["weighted_split"] = [int(x) for x in dr.plan.begin]
    if rank == 0:
        results["splits"] = splits
        results["rendered"] = rendered
        torch.save(results, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,layout", [(2, "tiles"), (2, "bands"), (4, "bands")])
def test_frame_assembly_across_ranks(tmp_path, world, layout):
    out = str(tmp_path / "res.pt")
    mp.start_processes(_worker, args=(world, _free_port(), out, layout), nprocs=world, join=True, start_method="spawn")
    res = torch.load(out)
    steps, lag = res["steps"], res["lag"]
    assert all(res[f"equal{k}"] for k in range(steps)) and res["equal_pipelined"]
    splits = res["splits"]
    n = splits[0][-1]
    uniform = splits[0]
    assert uniform[0] == 0 and uniform[-1] == n and len(uniform) == world + 1
    assert max(np.diff(uniform)) - min(np.diff(uniform)) <= 1          # equal shares at the start
    assert all(s == uniform for s in splits[:lag])                      # until the first frame's costs arrive
    assert splits[lag] != uniform and splits[lag][0] == 0 and splits[lag][-1] == n   # rebalanced from the first frame's costs
    assert splits[-1] == uniform                                        # and uniform again after the camera move
    if layout == "bands":
        assert res["equal_weighted"]
        w = res["weights"]
        assert abs(sum(w) - 1.0) < 1e-5 and w[1] < w[0]
        ws = res["weighted_split"]
        assert ws[0] == 0 and ws[-1] == n and (ws[2] - ws[1]) == min(np.diff(ws)) and (ws[2] - ws[1]) < (ws[1] - ws[0])   # the slow rank renders the fewest bands


def _accumulate_worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cudaraytracer_amd as acr
    import oracle
    from cudaraytracer_amd import distributed as D

    data = acr.ImportData.load(os.path.join(HERE, "golden", "scenes", "cornell-box-cycles.json"))
    view, keep = data.pack()
    o = oracle.OracleScene(view)
    state = dict(seed=0)
    calls = []

    def render_bands(tile_ids, out, cost, frames):
        # stand-in for the GPU core: the oracle averages frame `frames` into the rows of this rank's bands, whose
        # running mean `out` holds (renderer.cu:375-387)
        calls.append((len(tile_ids), frames))
        if len(tile_ids) == 0:
            return
        assert list(tile_ids) == list(range(int(tile_ids[0]), int(tile_ids[0]) + len(tile_ids)))
        r0, r1 = int(tile_ids[0]) * TH, min((int(tile_ids[-1]) + 1) * TH, H)
        screen = np.zeros((H, W, 4), np.float32)
        screen[r0:r1] = out.numpy().reshape(r1 - r0, W, 4)
        r = o.render(W, H, seed=state["seed"], frames=frames, max_bounces=2, threads=2, screen=screen, row_begin=r0, row_end=r1)
        out.copy_(torch.from_numpy(r["screen"][r0:r1].reshape(-1, 4)))
        if cost is not None:
            c = [1 + int((r["prim"][b * TH:min((b + 1) * TH, H)] != 0xFFFFFFFF).sum()) for b in tile_ids]
            cost.copy_(torch.tensor(c, dtype=torch.int32))

    dr = D.DistributedRenderer(W, H, render_bands, torch.device("cpu"), tile_w=W, tile_h=TH, rebalance_every=2, accumulate=True)
    assert dr.bands and dr.depth == 1
    results = {}

    def reference(seeds):
        screen = None
        for f, seed in enumerate(seeds):
            screen = o.render(W, H, seed=seed, frames=f, max_bounces=2, threads=2, screen=screen)["screen"]
        return screen

    # frames 0..5 with the camera still; the costs measured on frames 1 and 3 rebuild the split one frame later, so
    # running means change owner in the middle of the accumulation
    seeds = [100 + f for f in range(6)]
    splits = []
    for seed in seeds:
        state["seed"] = seed
        dr.render_frame()
        splits.append(dr.last_split)
    frame = dr.finish()
    if rank == 0:
        results["equal_still"] = bool(np.array_equal(frame.numpy().view(np.uint32), reference(seeds).view(np.uint32)))
    every = [None] * world
    dist.all_gather_object(every, splits)
    assert all(e == every[0] for e in every)
    # a second read without a new frame exchanges nothing more and returns the same frame
    again = dr.finish()
    if rank == 0:
        results["equal_again"] = bool(np.array_equal(again.numpy().view(np.uint32), frame.numpy().view(np.uint32)))
    # two more frames on top (the gather wrote into rank 0's copy of the others' bands, never into anyone's own bands)
    for seed in (106, 107):
        state["seed"] = seed
        dr.render_frame()
    frame = dr.finish()
    if rank == 0:
        results["equal_eight"] = bool(np.array_equal(frame.numpy().view(np.uint32), reference(seeds + [106, 107]).view(np.uint32)))
    # the camera moves: the mean restarts, the split is uniform again
    dr.camera_moved()
    dr.gather_every = 1
    exchanges = []
    plain_exchange = dr._exchange
    dr._exchange = lambda slot: (exchanges.append(dr.frame_no), plain_exchange(slot))
    for seed in (200, 201, 202):
        state["seed"] = seed
        dr.render_frame()
        splits.append(dr.last_split)
    frame = dr.finish()
    # gather_every = 1: the pieces travelled after every frame, and finish() had nothing left to exchange
    assert len(exchanges) == 3
    if rank == 0:
        results["equal_moved"] = bool(np.array_equal(frame.numpy().view(np.uint32), reference([200, 201, 202]).view(np.uint32)))
    if rank == 0:
        results["splits"] = splits
        results["frames_passed"] = [c[1] for c in calls]
        torch.save(results, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_progressive_accumulation_across_ranks(tmp_path, world):
    """frames > 0 across ranks: every rank keeps the running mean of its bands, a rebalance in the middle moves
    running means between ranks, the gather happens when the frame is read; bit-identical to one process
    accumulating the same frames"""
    out = str(tmp_path / "res.pt")
    mp.start_processes(_accumulate_worker, args=(world, _free_port(), out), nprocs=world, join=True, start_method="spawn")
    res = torch.load(out)
    assert res["equal_still"] and res["equal_again"] and res["equal_eight"] and res["equal_moved"]
    splits = res["splits"]
    uniform = splits[0]
    assert splits[1] == uniform
    assert any(s != uniform for s in splits[2:6])   # bands changed owner while frames were being averaged
    assert splits[6] == uniform                     # uniform again after the camera move (frames 106, 107 are not in the list)
    assert res["frames_passed"] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2]


def test_single_rank_path_without_process_group():
    """world size 1: no communication at all, same assembly code."""
    sys.path.insert(0, HERE)
    import cudaraytracer_amd as acr
    import oracle
    from cudaraytracer_amd import distributed as D
    data = acr.ImportData.load(os.path.join(HERE, "golden", "scenes", "triangles.json"))
    view, keep = data.pack()
    o = oracle.OracleScene(view)
    ref = o.render(W, H, seed=5, max_bounces=1)["screen"]
    tiles_x = (W + TW - 1) // TW

    def render_tiles(tile_ids, out, cost):
        buf = np.zeros((len(tile_ids), TH, TW, 4), np.float32)
        for k, t in enumerate(tile_ids):
            x0, y0 = (int(t) % tiles_x) * TW, (int(t) // tiles_x) * TH
            w, h = min(TW, W - x0), min(TH, H - y0)
            buf[k, :h, :w] = ref[y0:y0 + h, x0:x0 + w]
        out.copy_(torch.from_numpy(buf.reshape(len(tile_ids), TH * TW, 4)))
        if cost is not None:
            cost.fill_(1)

    dr = D.DistributedRenderer(W, H, render_tiles, torch.device("cpu"), tile_w=TW, tile_h=TH)
    dr.render_frame()
    assert np.array_equal(dr.finish().numpy().view(np.uint32), ref.view(np.uint32))
