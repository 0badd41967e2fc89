"""One frame across several GPUs: one process per GPU, the scene replicated on
each, the image cut into tiles, each rank rendering a contiguous run of tiles
along a Z-curve, the finished tiles gathered to rank 0.

The assignment follows the rules of the reference's adaptive assignment
(src/renderer/renderer.cu:343-355,457-461; src/utils/bih.h:29-66), applied to
tiles and GPUs instead of pixels and threads: the work goes through an
explicit table, the table is rebuilt from costs measured on earlier frames,
and a camera move resets it to the uniform split (renderer.cu:165,344-348).

Pixels are independent, so a frame needs no exchange while it renders.  Per
frame there are two small steps over torch.distributed (RCCL on GPUs, gloo on
CPU): an all-reduce of the per-tile cost vector (4 bytes per tile) so that
every rank derives the same next assignment, and point-to-point sends of each
rank's finished tiles to rank 0 (grouped isend/irecv: a gather with unequal
counts; each sender has its own xGMI link to GPU 0).

The renderer itself is passed in as a callable so that the same code runs on
CPU ranks in tests:

    render_tiles(tile_ids, out, cost) -> None
        tile_ids : 1-D int array of the tiles this rank renders, in order
        out      : torch tensor [len(tile_ids), tile_h * tile_w, 4] float32 to fill
        cost     : torch tensor [len(tile_ids)] float32 to fill (any positive cost measure)
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _capi


def tile_grid(width, height, tile_w, tile_h):
    return (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h


def tile_order(tiles_x, tiles_y):
    """Z-curve visiting order of the tile grid (host library)."""
    out = np.zeros(tiles_x * tiles_y, dtype=np.uint32)
    rc = _capi.host_lib().acr_host_tile_order(tiles_x, tiles_y, out.ctypes.data_as(_capi.c_u32_p))
    if rc != 0:
        raise RuntimeError("acr_host_tile_order failed")
    return out


def partition(order, cost, parts):
    """Cuts `order` into `parts` contiguous runs of near-equal summed cost
    (cost None = uniform).  Returns parts+1 offsets into `order`."""
    order = np.ascontiguousarray(order, dtype=np.uint32)
    begin = np.zeros(parts + 1, dtype=np.uint32)
    cp = None
    if cost is not None:
        cost = np.ascontiguousarray(cost, dtype=np.float32)
        cp = cost.ctypes.data_as(_capi.c_float_p)
    rc = _capi.host_lib().acr_host_partition_tiles(order.ctypes.data_as(_capi.c_u32_p), cp, len(order), parts,
                                                   begin.ctypes.data_as(_capi.c_u32_p))
    if rc != 0:
        raise RuntimeError("acr_host_partition_tiles failed")
    return begin


class TiledFrame:
    """Tile assignment state shared (by construction, not by communication) by all ranks."""

    def __init__(self, width, height, world_size, tile_w=64, tile_h=16):
        if tile_w % 16 or tile_h % 16:
            raise ValueError("tile sizes must be multiples of 16")
        self.width, self.height, self.world = width, height, world_size
        self.tile_w, self.tile_h = tile_w, tile_h
        self.tiles_x, self.tiles_y = tile_grid(width, height, tile_w, tile_h)
        self.num_tiles = self.tiles_x * self.tiles_y
        self.order = tile_order(self.tiles_x, self.tiles_y)
        self.cost = None           # per-tile cost of the last frame, indexed by tile id
        self.begin = partition(self.order, None, world_size)

    def reset(self):
        """Camera moved and no history applies: back to the uniform split."""
        self.cost = None
        self.begin = partition(self.order, None, self.world)

    def update(self, cost):
        """New per-tile costs (every rank passes the same vector)."""
        self.cost = np.asarray(cost, dtype=np.float32).copy()
        self.begin = partition(self.order, self.cost, self.world)

    def tiles_of(self, rank):
        return self.order[int(self.begin[rank]):int(self.begin[rank + 1])]


class DistributedRenderer:
    """Renders frames across the ranks of the default process group and assembles them on rank 0."""

    def __init__(self, width, height, render_tiles, device, tile_w=64, tile_h=16, scatter=None, rebalance=True, cost_lag=0):
        """scatter(tiles, tile_ids, frame): writes compact tiles into the full frame tensor
        [height, width, 4] on rank 0; None selects a torch implementation.
        cost_lag: a frame's measured costs steer the assignment `cost_lag` frames later (every rank
        applies them at the same frame, so the assignment stays identical everywhere); a lag of 2 lets
        the device-to-host copy of the cost vector finish behind the next frame instead of stalling it."""
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.device = device
        self.render_tiles = render_tiles
        self.scatter = scatter or self._scatter_torch
        self.rebalance = rebalance
        self.cost_lag = cost_lag
        self._pending = []  # (host tensor, event or None) per frame, oldest first
        self.plan = TiledFrame(width, height, self.world, tile_w, tile_h)
        self.frame = torch.zeros((height, width, 4), dtype=torch.float32, device=device) if self.rank == 0 else None
        self.tile_px = tile_w * tile_h

    def camera_moved(self):
        self._pending.clear()
        self.plan.reset()

    def _scatter_torch(self, tiles, tile_ids, frame):
        p = self.plan
        t = tiles.view(len(tile_ids), p.tile_h, p.tile_w, 4)
        for k, tid in enumerate(tile_ids):
            x0, y0 = (int(tid) % p.tiles_x) * p.tile_w, (int(tid) // p.tiles_x) * p.tile_h
            w, h = min(p.tile_w, p.width - x0), min(p.tile_h, p.height - y0)
            frame[y0:y0 + h, x0:x0 + w] = t[k, :h, :w]

    def render_frame(self):
        """One frame.  Returns the assembled frame tensor on rank 0, None elsewhere."""
        p = self.plan
        mine = np.ascontiguousarray(p.tiles_of(self.rank), dtype=np.uint32)
        out = torch.empty((len(mine), self.tile_px, 4), dtype=torch.float32, device=self.device)
        cost = torch.zeros(len(mine), dtype=torch.float32, device=self.device)
        if len(mine):
            self.render_tiles(mine, out, cost)

        if self.world > 1:
            # gather with unequal counts: rank 0 receives every other rank's run of tiles
            ops, staged = [], {}
            if self.rank == 0:
                for r in range(1, self.world):
                    n = int(p.begin[r + 1]) - int(p.begin[r])
                    if n:
                        staged[r] = torch.empty((n, self.tile_px, 4), dtype=torch.float32, device=self.device)
                        ops.append(dist.P2POp(dist.irecv, staged[r], r))
            elif len(mine):
                ops.append(dist.P2POp(dist.isend, out, 0))
            works = dist.batch_isend_irecv(ops) if ops else []
            # cost feedback: every rank contributes its tiles' costs, all ranks get the full vector
            full_cost = torch.zeros(p.num_tiles, dtype=torch.float32, device=self.device)
            if len(mine):
                full_cost[torch.as_tensor(mine.astype(np.int64), device=self.device)] = cost
            dist.all_reduce(full_cost)
            for w in works:
                w.wait()
        else:
            staged = {}
            full_cost = torch.zeros(p.num_tiles, dtype=torch.float32, device=self.device)
            if len(mine):
                full_cost[torch.as_tensor(mine.astype(np.int64), device=self.device)] = cost

        if self.rank == 0:
            if len(mine):
                self.scatter(out, mine, self.frame)
            for r, buf in staged.items():
                self.scatter(buf, np.ascontiguousarray(p.tiles_of(r), dtype=np.uint32), self.frame)
        self.last_cost = full_cost  # device tensor: per-tile cost of this frame (ray counts with the HIP core)
        if self.rebalance:
            if full_cost.is_cuda:
                host = torch.empty(full_cost.shape, dtype=full_cost.dtype, pin_memory=True)
                host.copy_(full_cost, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:
                host, ev = full_cost.clone(), None
            self._pending.append((host, ev))
            if len(self._pending) > self.cost_lag:
                host, ev = self._pending.pop(0)
                if ev is not None:
                    ev.synchronize()
                p.update(host.numpy())
        return self.frame


def hip_render_tiles(core, width, height, tile_w, tile_h, frame_args):
    """render_tiles callable over a HipCore: renders straight into the torch
    tensors' device memory through the C ABI (ray counts as the tile cost)."""
    def fn(tile_ids, out, cost):
        rays = torch.zeros(len(tile_ids), dtype=torch.int32, device=out.device)
        args = frame_args() if callable(frame_args) else dict(frame_args)
        core.render(width, height, tile_ids=tile_ids, tile_w=tile_w, tile_h=tile_h, screen_dev=C.c_void_p(out.data_ptr()),
                    tile_rays_dev=C.c_void_p(rays.data_ptr()), **args)
        cost.copy_(rays.to(torch.float32))
    return fn


def hip_scatter(core, width, height, tile_w, tile_h):
    def fn(tiles, tile_ids, frame):
        core.scatter_tiles(C.c_void_p(tiles.data_ptr()), tile_ids, tile_w, tile_h, width, height, C.c_void_p(frame.data_ptr()))
    return fn
