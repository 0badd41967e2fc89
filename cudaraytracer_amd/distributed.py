"""One frame across several GPUs: one process per GPU, the scene replicated on
each, the image cut into tiles, each rank rendering a contiguous run of tiles
along a Z-curve, the finished tiles gathered to rank 0.

The assignment follows the rules of the reference's adaptive assignment
(src/renderer/renderer.cu:343-355,457-461; src/utils/bih.h:29-66), applied to
tiles and GPUs instead of pixels and threads: the work goes through an
explicit table, the table is rebuilt from costs measured on earlier frames,
and a camera move resets it to the uniform split (renderer.cu:165,344-348).

Pixels are independent, so a frame needs no exchange while it renders.  Over
torch.distributed (RCCL on GPUs, gloo on CPU) there are two steps: per frame,
point-to-point sends of each rank's finished piece to rank 0 (grouped
isend/irecv: a gather with unequal counts; each sender has its own xGMI link to
GPU 0); and, every `rebalance_every` frames, an all-reduce of the per-tile cost
vector (4 bytes per tile) so that every rank derives the same next assignment.

The renderer itself is passed in as a callable so that the same code runs on
CPU ranks in tests:

    render_tiles(tile_ids, out, cost) -> None
        tile_ids : 1-D int array of the tiles this rank renders, in order; may be empty (the call is made on every
                   frame so that state the callable keeps per frame, like seeds and cameras, advances on all ranks)
        out      : torch tensor [len(tile_ids), tile_h * tile_w, 4] float32 to fill
        cost     : torch tensor [len(tile_ids)] int32 to fill (any positive cost measure), or None when
                   this frame's costs are not needed
    With accumulate=True the call carries a fourth, keyword argument, frames=k: the number of frames already
    averaged into `out`, which then holds the running mean of those frames on entry (k == 0: overwrite) and
    of one more frame on return (renderer.cu:375-387).
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


def partition(order, cost, parts, weights=None):
    """Cuts `order` into `parts` contiguous runs whose summed costs are in the ratio of `weights` (None = equal
    shares; cost None = uniform cost).  Returns parts+1 offsets into `order`."""
    order = np.ascontiguousarray(order, dtype=np.uint32)
    begin = np.zeros(parts + 1, dtype=np.uint32)
    cp = wp = None
    if cost is not None:
        cost = np.ascontiguousarray(cost, dtype=np.float32)
        cp = cost.ctypes.data_as(_capi.c_float_p)
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float32)
        if len(weights) != parts:
            raise ValueError("one weight per part")
        wp = weights.ctypes.data_as(_capi.c_float_p)
    rc = _capi.host_lib().acr_host_partition_tiles_weighted(order.ctypes.data_as(_capi.c_u32_p), cp, len(order), parts, wp,
                                                            begin.ctypes.data_as(_capi.c_u32_p))
    if rc != 0:
        raise RuntimeError("acr_host_partition_tiles_weighted failed")
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
        # full-width bands are visited top to bottom (a run of them is one contiguous piece of the frame);
        # smaller tiles along a Z-curve (a run of them is a compact region)
        self.bands = self.tiles_x == 1 and tile_w == width
        self.order = np.arange(self.num_tiles, dtype=np.uint32) if self.bands else tile_order(self.tiles_x, self.tiles_y)
        self.cost = None           # per-tile cost of the last frame, indexed by tile id
        self.weights = None        # share of the total cost each rank gets (None = equal shares)
        self.begin = partition(self.order, None, world_size)

    def reset(self):
        """Camera moved and no history applies: back to uniform costs (the ranks' shares are a property of
        the machine, not of the view, and stay)."""
        self.cost = None
        self.begin = partition(self.order, None, self.world, self.weights)

    def update(self, cost):
        """New per-tile costs (every rank passes the same vector)."""
        self.cost = np.asarray(cost, dtype=np.float32).copy()
        self.begin = partition(self.order, self.cost, self.world, self.weights)

    def set_weights(self, weights):
        """Unequal shares (every rank passes the same vector): rank r gets weights[r] / sum of the total cost."""
        self.weights = None if weights is None else np.asarray(weights, dtype=np.float32).copy()
        self.begin = partition(self.order, self.cost, self.world, self.weights)

    def tiles_of(self, rank):
        return self.order[int(self.begin[rank]):int(self.begin[rank + 1])]


def solve_shares(measurements, num_tiles):
    """Shares of the frame (one weight per rank, summing to 1) that make every rank need the same time per frame.

    measurements: the last one or two (bands per rank, times[rank] = [render seconds, send seconds]) taken at
    different shares.  With frames in flight rank 0 needs its render time per frame and every other rank the
    larger of its render and its send time.  One measurement: time is taken as proportional to the share.  Two:
    each phase of each rank is modelled as fixed cost + cost per band (fitted through the two points, clamped to
    non-negative values; a rank whose share did not change keeps the proportional model), and the common frame
    time T is found by bisection so that the shares the ranks can do in T add up to the frame."""
    share, times = measurements[-1]
    world = len(share)
    fixed = np.zeros((world, 2))
    per_band = times / np.maximum(share, 1.0)[:, None]
    if len(measurements) > 1:
        share0, times0 = measurements[0]
        for r in range(world):
            ds = share[r] - share0[r]
            if abs(ds) < 1.0:
                continue
            for ph in range(2):
                c = (times[r, ph] - times0[r, ph]) / ds
                o = times[r, ph] - c * share[r]
                if c > 0 and o >= 0:
                    per_band[r, ph], fixed[r, ph] = c, o
    per_band = np.maximum(per_band, 1e-12)

    def bands_within(t):
        can = (t - fixed) / per_band                      # bands each phase gets through in t
        b = np.minimum(can[:, 0], can[:, 1])
        b[0] = can[0, 0]                                  # rank 0 sends nothing
        return np.clip(b, 0.0, float(num_tiles))

    lo, hi = 0.0, float(np.max(fixed) + np.max(per_band) * num_tiles)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if bands_within(mid).sum() >= num_tiles:
            hi = mid
        else:
            lo = mid
    b = bands_within(hi)
    # No rank is left without work: a rank the model would give less than one band (a rank far slower than the
    # others, or a measurement taken while something else used the device or the link) still gets one, and the
    # caller is told, since frames will then take as long as that rank's band.
    if num_tiles >= world and (b < 1.0).any():
        import warnings
        warnings.warn("multi-GPU shares: the measured speeds give rank(s) %s less than one band of the frame; they are held at one band "
                      "(render s %s, send s %s)" % ([int(r) for r in np.nonzero(b < 1.0)[0]], np.round(times[:, 0], 6).tolist(),
                                                    np.round(times[:, 1], 6).tolist()), RuntimeWarning, stacklevel=2)
        b = np.maximum(b, 1.0)
    return b / b.sum()


class DistributedRenderer:
    """Renders frames across the ranks of the default process group and assembles them on rank 0.

    Two layouts, chosen by the tile shape:

    * bands (tile_w == width, the default when the width is a multiple of 16): a tile is a full-width band of
      tile_h rows and a rank's run of bands is one contiguous piece of the frame.  Every rank renders straight
      into frame-shaped memory, rank 0 receives each other rank's piece straight into the final frame: no
      staging buffer and no scatter pass.  Frames are double-buffered and the exchange of frame k is only
      waited for when its buffer comes up again, so it overlaps the rendering of frame k + 1.
    * tiles (anything else): tile buffers are kept in Z-curve order; the runs of the ranks, concatenated in
      rank order, are the whole curve whatever the split, so rank 0 owns one buffer [num_tiles][tile pixels]
      float4, receives rank r's run into slice [begin[r], begin[r+1]) and one scatter writes the frame.

    Per-tile costs live in one [num_tiles] vector in visiting order."""

    def __init__(self, width, height, render_tiles, device, tile_w=None, tile_h=16, scatter=None, rebalance_every=1,
                 stage_through_host=False, depth=2, accumulate=False):
        """scatter(tiles, tile_ids, frame): writes compact tiles into the full frame tensor
        [height, width, 4] on rank 0; None selects a torch implementation (tiles layout only).
        rebalance_every: the assignment is rebuilt from the measured costs every this many frames (0 = never);
        the reference re-sorts its pixel table every 60 still frames (renderer.cu:392,457).
        stage_through_host: move tiles and costs through CPU tensors for the exchange (for a process group
        whose backend cannot carry device tensors, e.g. gloo when rehearsing the multi-rank path on one GPU).
        depth: frame buffers in flight (bands layout).
        accumulate: progressive accumulation while the camera stands still, the reference's render loop
        (renderer.cu:375-387,423): frame k is averaged into the running mean of the frames before it.  Bands layout
        only.  Every rank keeps the running mean of the bands it owns, in one frame buffer; nothing is exchanged
        per frame; finish() gathers the pieces on rank 0 (only then, or every `gather_every` frames if that
        attribute is set); when a rebalance moves bands to another rank their running means travel from the old
        owner to the new one; camera_moved() restarts the mean (the next frame overwrites, as frames == 0 does)."""
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.device = device
        self.render_tiles = render_tiles
        self.scatter = scatter or self._scatter_torch
        self.rebalance_every = rebalance_every
        self.stage_through_host = stage_through_host
        if tile_w is None:
            tile_w = width if width % 16 == 0 else 64
        self.plan = TiledFrame(width, height, self.world, tile_w, tile_h)
        self.tile_px = tile_w * tile_h
        self.width, self.height = width, height
        n = self.plan.num_tiles
        self.bands = self.plan.bands
        self.accumulate = bool(accumulate)
        if self.accumulate and not self.bands:
            raise ValueError("accumulation across ranks needs the bands layout (a width that is a multiple of 16, tile_w == width)")
        self.depth = (1 if self.accumulate else max(1, depth)) if self.bands else 1
        self.still_frames = 0       # accumulate: frames averaged into the running means so far
        self.gather_every = 0       # accumulate: also gather on rank 0 every this many frames (0: only in finish())
        self._gathered_at = -1      # frame_no whose pieces rank 0 holds
        self._owned = None          # accumulate: the split (begin offsets) the running means are distributed by
        # bands: every rank renders into frame-shaped buffers (rank 0's are the assembled frames);
        # tiles: rank 0 holds every tile, the others their own run (sized for the largest possible run)
        self.frames = [torch.zeros((height, width, 4), dtype=torch.float32, device=device)
                       for _ in range(self.depth if self.bands else (1 if self.rank == 0 else 0))]
        self.tiles = None if self.bands else torch.zeros((n, self.tile_px, 4), dtype=torch.float32, device=device)
        self.cost = torch.zeros(n, dtype=torch.int32, device=device)  # visiting order; only this rank's run is ever non-zero
        self.pending = [[] for _ in range(self.depth)]  # exchange handles still in flight, per frame buffer
        self.frame_no = 0
        self.frames_since_rebalance = 0
        self.last_cost = self.cost
        self._rays_last = torch.zeros((), dtype=torch.int64, device=device)
        self._exchange_off = False
        self._last_slot = 0
        self.calibration = None
        self._calibrations = []  # (bands per rank, [render s, send s] per rank) of each calibrate() call
        # Costs measured on frame k decide the split from frame k + cost_lag on: their sum over the ranks and the copy
        # to the host are queued behind frame k and only waited for when frame k + cost_lag starts, so that a split
        # rebuilt every frame (camera flythrough) keeps `depth` frames in flight.  Every rank applies the update
        # at the same frame number, hence the same split everywhere.
        self.cost_lag = self.depth if self.bands else 1
        self._cost_updates = []  # (frame number to apply at, host tensor, event / work)
        self.last_split = None

    @property
    def frame(self):
        """The buffer of the most recent frame on rank 0 (complete after finish())."""
        if not self.frames:
            return None
        return self.frames[(self.frame_no - 1) % len(self.frames)] if self.frame_no else self.frames[0]

    def camera_moved(self):
        """No history applies to the new view: uniform costs again, and costs still on their way belong to the old view."""
        self.frames_since_rebalance = 0
        self._drop_cost_updates()
        self.plan.reset()
        self.still_frames = 0  # the next frame overwrites: no running mean needs to follow the new split
        self._owned = None

    def _drop_cost_updates(self):
        for _, _, ready in self._cost_updates:
            if ready is not None:
                ready.synchronize() if hasattr(ready, "synchronize") else ready.wait()
        self._cost_updates = []

    def _apply_cost_updates(self):
        """Rebuilds the split from every cost vector that is due at this frame number."""
        p = self.plan
        while self._cost_updates and self._cost_updates[0][0] <= self.frame_no:
            _, host, ready = self._cost_updates.pop(0)
            if ready is not None:
                ready.synchronize() if hasattr(ready, "synchronize") else ready.wait()
            by_tile = np.zeros(p.num_tiles, dtype=np.float32)
            by_tile[p.order] = host.numpy()
            p.update(by_tile)

    def finish(self):
        """Waits for every exchange still in flight; returns the most recent frame on rank 0.  With accumulation
        this is where the ranks' pieces travel to rank 0, unless they already have for this frame."""
        if self.accumulate and self.world > 1 and self.frame_no and self._gathered_at != self.frame_no:
            self._exchange(0)
            self._gathered_at = self.frame_no
        for slot in range(self.depth):
            for w in self.pending[slot]:
                w.wait()
            self.pending[slot] = []
        return self.frame if self.rank == 0 else None

    def settle(self):
        """Applies every cost update still on its way, whatever frame it was due at (all ranks must call it at the
        same point): for callers that want the split the measured costs lead to before they go on."""
        for k in range(len(self._cost_updates)):
            self._cost_updates[k] = (self.frame_no, ) + self._cost_updates[k][1:]
        self._apply_cost_updates()

    def _scatter_torch(self, tiles, tile_ids, frame):
        p = self.plan
        t = tiles.view(len(tile_ids), p.tile_h, p.tile_w, 4)
        for k, tid in enumerate(tile_ids):
            x0, y0 = (int(tid) % p.tiles_x) * p.tile_w, (int(tid) // p.tiles_x) * p.tile_h
            w, h = min(p.tile_w, p.width - x0), min(p.tile_h, p.height - y0)
            frame[y0:y0 + h, x0:x0 + w] = t[k, :h, :w]

    def _band_pixels(self, b0, b1):
        """flat pixel range of bands [b0, b1) inside the frame (the last band may be cut by the frame's bottom edge)"""
        return b0 * self.tile_px, min(b1 * self.tile_px, self.width * self.height)

    def _exchange(self, slot):
        """Bands layout: posts the point-to-point exchange of the frame in buffer `slot` (rank 0 receives every
        other rank's piece in place, the others send theirs); the handles go to self.pending[slot]."""
        begin = self._owned if (self.accumulate and self._owned is not None) else self.plan.begin
        flat = self.frames[slot].view(-1, 4)
        px0, px1 = self._band_pixels(int(begin[self.rank]), int(begin[self.rank + 1]))
        ops, staged = [], {}
        if self.rank == 0:
            for r in range(1, self.world):
                q0, q1 = self._band_pixels(int(begin[r]), int(begin[r + 1]))
                if q1 > q0:
                    dst = flat[q0:q1]
                    if self.stage_through_host:
                        staged[r] = (torch.empty(dst.shape, dtype=dst.dtype), dst)
                        dst = staged[r][0]
                    ops.append(dist.P2POp(dist.irecv, dst, r))
        elif px1 > px0:
            out = flat[px0:px1]
            ops.append(dist.P2POp(dist.isend, out.cpu() if self.stage_through_host else out, 0))
        works = dist.batch_isend_irecv(ops) if ops else []
        if staged:
            for w in works:
                w.wait()
            works = []
            for buf, dst in staged.values():
                dst.copy_(buf)
        self.pending[slot] = works

    def _move_running_means(self, old, new):
        """accumulate: the split changed from `old` to `new` (begin offsets) while running means exist: the bands
        that change owner travel from the old owner's frame buffer to the new owner's, point to point."""
        flat = self.frames[0].view(-1, 4)
        ops, staged = [], []
        for i in range(self.world):          # old owner
            for j in range(self.world):      # new owner
                if i == j or self.rank not in (i, j):
                    continue
                b0, b1 = max(int(old[i]), int(new[j])), min(int(old[i + 1]), int(new[j + 1]))
                if b1 <= b0:
                    continue
                q0, q1 = self._band_pixels(b0, b1)
                if q1 <= q0:
                    continue
                piece = flat[q0:q1]
                if self.rank == i:
                    ops.append(dist.P2POp(dist.isend, piece.cpu() if self.stage_through_host else piece, j))
                else:
                    dst = piece
                    if self.stage_through_host:
                        dst = torch.empty(piece.shape, dtype=piece.dtype)
                        staged.append((dst, piece))
                    ops.append(dist.P2POp(dist.irecv, dst, i))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for buf, dst in staged:
            dst.copy_(buf)

    def calibrate(self, frames=3):
        """Measures how fast each rank gets its share done and sets the ranks' shares accordingly.

        With frames in flight a rank other than 0 finishes a frame every max(render, send) and rank 0 every
        render (it sends nothing); at kilohertz frame rates the send over one xGMI link takes longer than the
        rendering, so equal shares would make the link the frame time.  Each of `frames` frames is run without
        overlap -- barrier, render, wait, exchange, wait -- every rank times its two phases, the times are shared
        with one all-reduce, and solve_shares() turns them into shares (call it two or three times: from the
        second call on it has two measurements at different shares and can tell each phase's fixed cost from its
        cost per band).  Bands layout only; a no-op for one rank.  The frames rendered here are ordinary frames."""
        if self.world == 1 or not self.bands or self.accumulate:
            return None  # (with accumulation nothing is sent per frame: the ranks' shares stay equal)
        import time
        on_gpu = torch.device(self.device).type == "cuda"

        def wait_all():
            for slot in range(self.depth):
                for w in self.pending[slot]:
                    w.wait()
                self.pending[slot] = []
            if on_gpu:
                torch.cuda.synchronize()

        keep_every, self.rebalance_every = self.rebalance_every, 0
        t_render = t_send = 0.0
        for _ in range(max(1, frames)):
            wait_all()
            dist.barrier()
            t0 = time.perf_counter()
            self._exchange_off = True
            self.render_frame()
            self._exchange_off = False
            wait_all()
            t1 = time.perf_counter()
            dist.barrier()
            t2 = time.perf_counter()
            self._exchange(self._last_slot)
            wait_all()
            t3 = time.perf_counter()
            t_render += t1 - t0
            t_send += t3 - t2
        self.rebalance_every = keep_every
        times = torch.zeros((self.world, 2), dtype=torch.float64)
        times[self.rank, 0], times[self.rank, 1] = t_render, t_send
        if not self.stage_through_host:
            times = times.to(self.device)
        dist.all_reduce(times)
        times = times.cpu().numpy()
        p = self.plan
        share = np.array([int(p.begin[r + 1]) - int(p.begin[r]) for r in range(self.world)], dtype=np.float64)
        times /= max(1, frames)
        self._calibrations.append((share, times))
        weights = solve_shares(self._calibrations[-2:], p.num_tiles)
        p.set_weights(weights)
        self.calibration = dict(render_s=times[:, 0], send_s=times[:, 1], weights=p.weights.copy())
        return self.calibration

    def render_frame(self, want_cost=False):
        """One frame.  Returns the frame tensor on rank 0 (None elsewhere); with the bands layout its
        exchange may still be in flight until finish() or until `depth` more frames were started.
        Per-tile costs are only measured on the frames whose costs are used: the one before a rebalance,
        or when want_cost is set (render_tiles gets cost=None otherwise)."""
        p = self.plan
        self._apply_cost_updates()
        self.last_split = [int(x) for x in p.begin]  # the split this frame is rendered with
        b0, b1 = int(p.begin[self.rank]), int(p.begin[self.rank + 1])
        rebalance_now = bool(self.rebalance_every) and self.frames_since_rebalance + 1 >= self.rebalance_every
        cost = self.cost[b0:b1] if (want_cost or rebalance_now) else None
        slot = self.frame_no % self.depth
        for w in self.pending[slot]:  # this buffer's previous exchange must be over before it is overwritten
            w.wait()
        self.pending[slot] = []

        if self.bands and self.accumulate:
            if self.still_frames > 0 and self._owned is not None and self.world > 1 and list(self._owned) != list(p.begin):
                self._move_running_means(self._owned, p.begin)
            self._owned = np.array(p.begin, copy=True)
            flat = self.frames[0].view(-1, 4)
            px0, px1 = self._band_pixels(b0, b1)
            self.render_tiles(p.order[b0:b1], flat[px0:px1], cost, frames=self.still_frames)
            self.still_frames += 1
            self._last_slot = 0
            if self.world > 1 and self.gather_every and self.still_frames % self.gather_every == 0:
                self._exchange(0)
                self._gathered_at = self.frame_no + 1
        elif self.bands:
            flat = self.frames[slot].view(-1, 4)
            px0, px1 = self._band_pixels(b0, b1)
            out = flat[px0:px1]
            self.render_tiles(p.order[b0:b1], out, cost)  # also with an empty run: the callable keeps per-frame state (seeds, cameras)
            self._last_slot = slot
            if self.world > 1 and not self._exchange_off:
                self._exchange(slot)
        else:
            # rank 0 keeps its run at its Z position; other ranks use the front of their buffer
            at = b0 if self.rank == 0 else 0
            out = self.tiles[at:at + (b1 - b0)]
            self.render_tiles(p.order[b0:b1], out, cost)
            if self.world > 1:
                ops, staged = [], {}
                if self.rank == 0:
                    for r in range(1, self.world):
                        rb0, rb1 = int(p.begin[r]), int(p.begin[r + 1])
                        if rb1 > rb0:
                            dst = self.tiles[rb0:rb1]
                            if self.stage_through_host:
                                staged[r] = torch.empty(dst.shape, dtype=dst.dtype)
                                dst = staged[r]
                            ops.append(dist.P2POp(dist.irecv, dst, r))
                elif b1 > b0:
                    ops.append(dist.P2POp(dist.isend, out.cpu() if self.stage_through_host else out, 0))
                works = dist.batch_isend_irecv(ops) if ops else []
                for w in works:
                    w.wait()
                for r, buf in staged.items():
                    self.tiles[int(p.begin[r]):int(p.begin[r + 1])].copy_(buf)
            if self.rank == 0:
                self.scatter(self.tiles, p.order, self.frames[0])

        self.frame_no += 1
        self.last_cost = self.cost
        self._rays_last = cost.sum() if (cost is not None and b1 > b0) else None
        self.frames_since_rebalance += 1
        if rebalance_now:
            self.frames_since_rebalance = 0
            full = self.cost.to(torch.float32)  # a copy: the run's entries are zeroed below
            ready = None
            if self.world > 1:
                if self.stage_through_host:
                    full = full.cpu()
                dist.all_reduce(full)  # every rank contributes its run; all ranks get the whole vector (stream-ordered on GPUs)
            if full.is_cuda:
                host = torch.empty(full.shape, dtype=full.dtype, pin_memory=True)
                host.copy_(full, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record()
            else:
                host = full
            if b1 > b0:
                self.cost[b0:b1].zero_()  # the next run may start elsewhere: leave no stale costs behind
            self._cost_updates.append((self.frame_no + self.cost_lag - 1, host, ready))
        return self.frame if self.rank == 0 else None

    def rays_last_frame(self):
        """Device scalar: rays this rank traced in the last frame (sum of its per-tile costs when the cost is a
        ray count).  Needs a frame rendered with costs (want_cost, or a frame that precedes a rebalance)."""
        if self._rays_last is None:
            return torch.zeros((), dtype=torch.int64, device=self.device)
        return self._rays_last


def hip_render_tiles(core, width, height, tile_w, tile_h, frame_args, cost_kind="cycles"):
    """render_tiles callable over a HipCore: renders straight into the torch tensors' device memory through
    the C ABI.  The per-tile cost is `fn.cost_kind` (an attribute the caller may change between frames):
    "cycles", the wave clock ticks spent in the tile -- what balancing wants, since a ray through the floor of the
    Cornell box costs ten times one to its ceiling -- or "rays", the number of rays traced, for counting."""
    def fn(tile_ids, out, cost, frames=None):
        args = frame_args() if callable(frame_args) else dict(frame_args)
        if frames is not None:
            args["frames"] = frames  # DistributedRenderer(accumulate=True): `out` holds the running mean of that many frames
        elif args.get("frames", 0) != 0:
            # the running mean reads the pixel's previous value (renderer.cu:375-387): with frame buffers in
            # rotation and bands that change owner between frames that value is another frame's, or another rank's
            raise ValueError("frames > 0 across GPUs needs DistributedRenderer(accumulate=True), which keeps every band's running mean "
                             "with its owner; without it every frame renders with frames == 0 (as after a camera move)")
        if len(tile_ids) == 0:
            return  # nothing of this frame is this rank's (an empty list would mean "the whole frame" to the C ABI)
        ptr = None if cost is None else C.c_void_p(cost.data_ptr())
        # the launch goes onto torch's current stream of the output's device: the exchange (RCCL) and the torch
        # operations on `out` and `cost` that follow are ordered against that stream
        core.set_stream(torch.cuda.current_stream(out.device).cuda_stream)
        core.render(width, height, tile_ids=tile_ids, tile_w=tile_w, tile_h=tile_h, screen_dev=C.c_void_p(out.data_ptr()),
                    tile_rays_dev=ptr if fn.cost_kind == "rays" else None,
                    tile_cycles_dev=ptr if fn.cost_kind == "cycles" else None, **args)
    if cost_kind not in ("cycles", "rays"):
        raise ValueError("cost_kind is 'cycles' or 'rays'")
    fn.cost_kind = cost_kind
    return fn


def hip_scatter(core, width, height, tile_w, tile_h):
    def fn(tiles, tile_ids, frame):
        core.scatter_tiles(C.c_void_p(tiles.data_ptr()), tile_ids, tile_w, tile_h, width, height, C.c_void_p(frame.data_ptr()))
    return fn
