"""Tile partitioning of one frame across GPUs: one process per GPU, scene replicated,
only framebuffer slices exchanged.

Every pixel of the reference's loop is independent (Raytracer/main.cpp:262-277 keeps no
cross-pixel state), so the frame shards without any data-path collective except the
final gather of the finished slices.  Rows are dealt out in `band_rows`-row bands
round-robin (band b goes to rank b % world): cost varies strongly with image height
(sky rows finish at once, floor rows cast shadow and reflection rays), and interleaving
evens it out where contiguous slabs would not.
"""
import numpy as np

from . import RowSet, band_rowset, rows_per_part

DEFAULT_BAND_ROWS = 16


def local_rows(H, band_rows, part, n_parts):
    """Image rows rendered by `part`, in local-buffer order (host mirror of rt_rowset)."""
    rs = band_rowset(H, band_rows, part, n_parts)
    r = np.arange(rs.n_rows)
    return (rs.first_band + (r // rs.band_rows) * rs.band_stride) * rs.band_rows + r % rs.band_rows


def deinterleave_numpy(gathered, H, band_rows):
    """Host reference of rt_deinterleave_bands: gathered[part, rows_per_part, W, ...] -> [H, W, ...]."""
    n_parts = gathered.shape[0]
    out = np.zeros((H,) + gathered.shape[2:], gathered.dtype)
    for p in range(n_parts):
        ys = local_rows(H, band_rows, p, n_parts)
        out[ys] = gathered[p, : len(ys)]
    return out


def reduce_over_ranks(vec, dist=None):
    """What a multi-rank measurement reports from one vector of per-rank numbers: (max over ranks,
    sum over ranks), as two tensors like `vec`.  Times take the max (the frame is done when the last
    rank is), ray counts the sum.  With dist None (one process) both are `vec` itself."""
    if dist is None:
        return vec, vec
    mx = vec.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = vec.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    return mx, sm


class HostStagedGather:
    """all_gather_into_tensor for device tensors over a process group that only moves host memory (gloo): the slice
    goes through the host, ordered on the current stream.  For rehearsing the partitioned job where RCCL cannot run
    (several ranks on one GPU); pass it as `dist` to PartitionedRenderer."""

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch

    def all_gather_into_tensor(self, out, inp):
        torch = self.torch
        torch.cuda.current_stream().synchronize()           # the slice is complete before it leaves the device
        host_out = torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(host_out, inp.cpu())
        out.copy_(host_out, non_blocking=False)             # enqueued on the current stream, ahead of the de-interleave


class Frame:
    """One enqueued frame.  `wait()` orders the caller's current stream behind it and returns the tensor
    (the full frame on rank 0, this rank's slice elsewhere)."""

    def __init__(self, tensor, stream, torch):
        self.tensor, self.stream, self.torch = tensor, stream, torch

    def wait(self):
        if self.stream is not None:
            cur = self.torch.cuda.current_stream(self.tensor.device)
            if cur.cuda_stream != self.stream.cuda_stream:
                cur.wait_stream(self.stream)
        return self.tensor


class _Slot:
    """What one frame in flight owns: a frame slot of a context (workspace, captured graph), a stream, its buffers."""

    def __init__(self, ctx, index, stream, local, gathered, frame):
        self.ctx, self.index, self.stream, self.local, self.gathered, self.frame = ctx, index, stream, local, gathered, frame


class PartitionedRenderer:
    """Renders this rank's bands and gathers the RGB8 slices with torch.distributed (backend "nccl",
    which is RCCL, on GPUs).  Buffers are allocated once.

    Stream order on a GPU: the library's render and de-interleave calls and the collective of one frame all
    run on ONE stream owned by this object (a non-default torch stream, so its handle is never 0 — the ABI
    reads a null stream as the context's private stream, which nothing else is ordered with).
    render_async() makes that stream wait for the caller's current stream and returns a Frame; Frame.wait()
    makes the caller's current stream wait for the frame, so whoever consumes the tensor there sees the
    finished frame without any device-wide synchronisation.  render() is the two together.

    Frames in flight: `frames_in_flight` = N gives N slots, each a frame slot of the context
    (rt_render_rows_device_slot: own workspace and captured graph over ONE scene and hierarchy) with its own
    stream and buffers; frames go to the slots round-robin, and a caller that waits for a frame only when it
    needs it (render_async now, Frame.wait() a few frames later) keeps N frames in flight.  (`ctx` may also be a
    sequence of contexts holding the same scene: one slot per context.)
    The slots' collectives share one further stream, so they run one at a time and in frame order on every rank.
    One rank's share of a small frame has fewer rays than the GPU has lanes and six dependent traversal
    launches, each as long as its longest rays: a second and third frame fill what one leaves idle (an
    eighth of a 1080p frame: 0.74 ms per frame alone, 0.39 ms with four in flight).

    `render_rows(rowset, out)` and `deinterleave(gathered, frame)` are the two device operations; the
    defaults call the slot's context (rt_render_rows_device, rt_deinterleave_bands).  Passing other callables
    (and a CPU device) runs the very same partition / gather / reassembly logic without a GPU."""

    def __init__(self, ctx, W, H, max_depth, rank=0, world=1, band_rows=DEFAULT_BAND_ROWS, device=None,
                 render_rows=None, deinterleave=None, dist=None, frames_in_flight=1):
        import torch
        self.torch = torch
        if isinstance(ctx, (list, tuple)):
            ctxs = [(c, 0) for c in ctx]
        else:
            from . import MAX_FRAME_SLOTS
            if not 1 <= int(frames_in_flight) <= MAX_FRAME_SLOTS:
                raise ValueError("frames_in_flight must be 1..%d (the context's frame slots)" % MAX_FRAME_SLOTS)
            ctxs = [(ctx, k) for k in range(int(frames_in_flight))]
        if not ctxs:
            raise ValueError("PartitionedRenderer needs at least one context")
        self.ctx, self.W, self.H, self.max_depth = ctxs[0][0], W, H, max_depth
        self.rank, self.world, self.band_rows = rank, world, band_rows
        self.rowset = band_rowset(H, band_rows, rank, world) if world > 1 else RowSet(0, 1, 1, H)
        self.cap_rows = rows_per_part(H, band_rows, world) if world > 1 else H
        dev = device if device is not None else torch.device("cuda", ctxs[0][0].device)
        self.device = torch.device(dev)
        self.on_gpu = self.device.type == "cuda"
        if world > 1 and dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.slots = []
        for c, index in ctxs:
            self.slots.append(_Slot(
                c, index, torch.cuda.Stream(device=self.device) if self.on_gpu else None,
                torch.zeros((self.cap_rows, W, 3), dtype=torch.uint8, device=self.device),
                torch.empty((world * self.cap_rows, W, 3), dtype=torch.uint8, device=self.device) if world > 1 else None,
                torch.empty((H, W, 3), dtype=torch.uint8, device=self.device) if (world > 1 and rank == 0) else None))
        self._next = 0
        self.stream = self.slots[0].stream            # the first slot's, for callers that work on it directly
        # Every collective of this object runs on ONE stream, in the order the frames were asked for: operations of one
        # communicator must not run side by side, and every rank must issue them in the same order.  With a single slot
        # that stream is the slot's own.
        self.gather_stream = (torch.cuda.Stream(device=self.device) if len(self.slots) > 1 else self.stream) if (self.on_gpu and world > 1) else None
        self.local, self.gathered, self.frame = self.slots[0].local, self.slots[0].gathered, self.slots[0].frame
        self._render_rows, self._deinterleave = render_rows, deinterleave

    @property
    def frames_in_flight(self):
        return len(self.slots)

    # ---- the default device operations (C ABI)
    @staticmethod
    def _handle(slot):
        h = slot.stream.cuda_stream
        if not h:
            raise RuntimeError("PartitionedRenderer needs a non-default stream")
        return h

    def _frame(self, slot):
        if self._render_rows is not None:
            self._render_rows(self.rowset, slot.local)
        else:
            slot.ctx.render_rows_device(self.W, self.H, self.rowset, self.max_depth, d_rgb8=slot.local.data_ptr(),
                                        stream=self._handle(slot), slot=slot.index)
        if self.world == 1:
            return slot.local
        # on a GPU the collective takes its order from the current stream
        if self.gather_stream is None or self.gather_stream is slot.stream:
            self.dist.all_gather_into_tensor(slot.gathered, slot.local)
        else:
            self.gather_stream.wait_stream(slot.stream)               # the slice is rendered
            with self.torch.cuda.stream(self.gather_stream):
                self.dist.all_gather_into_tensor(slot.gathered, slot.local)
            slot.stream.wait_stream(self.gather_stream)               # the gathered slices are there
        if self.rank == 0:
            if self._deinterleave is not None:
                self._deinterleave(slot.gathered, slot.frame)
            else:
                slot.ctx.deinterleave_bands(slot.gathered.data_ptr(), slot.frame.data_ptr(), self.W, self.H, self.band_rows,
                                            self.world, self.cap_rows, 3, stream=self._handle(slot))
            return slot.frame
        return slot.local

    def render_async(self, slot=None):
        """Enqueue one frame on the next slot (or on slot number `slot`); returns its Frame."""
        if slot is None:
            slot, self._next = self._next, (self._next + 1) % len(self.slots)
        sl = self.slots[slot]
        if not self.on_gpu:
            return Frame(self._frame(sl), None, self.torch)
        torch = self.torch
        caller = torch.cuda.current_stream(self.device)
        if caller.cuda_stream == sl.stream.cuda_stream:        # the caller already works on this slot's stream: nothing to order
            return Frame(self._frame(sl), None, torch)
        sl.stream.wait_stream(caller)            # earlier readers/writers of the slot's buffers on the caller's stream come first
        with torch.cuda.stream(sl.stream):
            out = self._frame(sl)
        return Frame(out, sl.stream, torch)

    def render(self, slot=None):
        """Enqueue one frame; returns the tensor holding the full frame on rank 0 (this rank's
        slice elsewhere), ordered on the caller's current stream."""
        return self.render_async(slot).wait()
