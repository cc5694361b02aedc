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


class PartitionedRenderer:
    """Renders this rank's bands and gathers the RGB8 slices with torch.distributed (backend "nccl",
    which is RCCL, on GPUs).  Buffers are allocated once.

    Stream order on a GPU: the library's render and de-interleave calls and the collective all run on
    ONE stream owned by this object (a non-default torch stream, so its handle is never 0 — the ABI
    reads a null stream as the context's private stream, which nothing else is ordered with).  render()
    makes that stream wait for the caller's current stream first and the caller's current stream wait
    for it last, so whoever consumes the returned tensor on the current stream sees the finished frame
    without any device-wide synchronisation.

    `render_rows(rowset, out)` and `deinterleave(gathered, frame)` are the two device operations; the
    defaults call the context (rt_render_rows_device, rt_deinterleave_bands).  Passing other callables
    (and a CPU device) runs the very same partition / gather / reassembly logic without a GPU."""

    def __init__(self, ctx, W, H, max_depth, rank=0, world=1, band_rows=DEFAULT_BAND_ROWS, device=None,
                 render_rows=None, deinterleave=None, dist=None):
        import torch
        self.torch = torch
        self.ctx, self.W, self.H, self.max_depth = ctx, W, H, max_depth
        self.rank, self.world, self.band_rows = rank, world, band_rows
        self.rowset = band_rowset(H, band_rows, rank, world) if world > 1 else RowSet(0, 1, 1, H)
        self.cap_rows = rows_per_part(H, band_rows, world) if world > 1 else H
        dev = device if device is not None else torch.device("cuda", ctx.device)
        self.device = torch.device(dev)
        self.on_gpu = self.device.type == "cuda"
        if world > 1 and dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.local = torch.zeros((self.cap_rows, W, 3), dtype=torch.uint8, device=self.device)
        self.gathered = torch.empty((world * self.cap_rows, W, 3), dtype=torch.uint8, device=self.device) if world > 1 else None
        self.frame = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device) if (world > 1 and rank == 0) else None
        self._render_rows = render_rows if render_rows is not None else self._ctx_render_rows
        self._deinterleave = deinterleave if deinterleave is not None else self._ctx_deinterleave

    # ---- the default device operations (C ABI)
    def _handle(self):
        h = self.stream.cuda_stream
        if not h:
            raise RuntimeError("PartitionedRenderer needs a non-default stream")
        return h

    def _ctx_render_rows(self, rowset, out):
        self.ctx.render_rows_device(self.W, self.H, rowset, self.max_depth, d_rgb8=out.data_ptr(), stream=self._handle())

    def _ctx_deinterleave(self, gathered, frame):
        self.ctx.deinterleave_bands(gathered.data_ptr(), frame.data_ptr(), self.W, self.H, self.band_rows, self.world,
                                    self.cap_rows, 3, stream=self._handle())

    def _frame(self):
        self._render_rows(self.rowset, self.local)
        if self.world == 1:
            return self.local
        # on a GPU the collective takes its order from the current stream (set by render() to self.stream)
        self.dist.all_gather_into_tensor(self.gathered, self.local)
        if self.rank == 0:
            self._deinterleave(self.gathered, self.frame)
            return self.frame
        return self.local

    def render(self):
        """Enqueue one frame; returns the tensor holding the full frame on rank 0 (this rank's
        slice elsewhere), ordered on the caller's current stream."""
        if not self.on_gpu:
            return self._frame()
        torch = self.torch
        caller = torch.cuda.current_stream(self.device)
        if caller.cuda_stream == self.stream.cuda_stream:      # the caller already works on this object's stream: nothing to order
            return self._frame()
        self.stream.wait_stream(caller)          # earlier readers/writers of our buffers on the caller's stream come first
        with torch.cuda.stream(self.stream):
            out = self._frame()
        caller.wait_stream(self.stream)
        return out
