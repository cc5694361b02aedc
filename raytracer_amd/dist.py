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


class PartitionedRenderer:
    """Renders this rank's bands on its GPU and gathers the RGB8 slices with RCCL
    (torch.distributed backend "nccl").  Buffers are allocated once."""

    def __init__(self, ctx, W, H, max_depth, rank=0, world=1, band_rows=DEFAULT_BAND_ROWS, device=None):
        import torch
        self.torch = torch
        self.ctx, self.W, self.H, self.max_depth = ctx, W, H, max_depth
        self.rank, self.world, self.band_rows = rank, world, band_rows
        self.rowset = band_rowset(H, band_rows, rank, world) if world > 1 else RowSet(0, 1, 1, H)
        self.cap_rows = rows_per_part(H, band_rows, world) if world > 1 else H
        dev = device if device is not None else torch.device("cuda", ctx.device)
        self.local = torch.zeros((self.cap_rows, W, 3), dtype=torch.uint8, device=dev)
        self.gathered = torch.empty((world * self.cap_rows, W, 3), dtype=torch.uint8, device=dev) if world > 1 else None
        self.frame = torch.empty((H, W, 3), dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None

    def render(self):
        """Enqueue one frame on torch's current stream; returns the tensor holding the full
        frame on rank 0 (this rank's slice elsewhere)."""
        torch = self.torch
        stream = torch.cuda.current_stream().cuda_stream
        self.ctx.render_rows_device(self.W, self.H, self.rowset, self.max_depth, d_rgb8=self.local.data_ptr(), stream=stream)
        if self.world == 1:
            return self.local
        import torch.distributed as dist
        dist.all_gather_into_tensor(self.gathered, self.local)
        if self.rank == 0:
            self.ctx.deinterleave_bands(self.gathered.data_ptr(), self.frame.data_ptr(), self.W, self.H, self.band_rows,
                                        self.world, self.cap_rows, 3, stream=stream)
            return self.frame
        return self.local
