"""The N > 1 path on CPU: gloo ranks each construct the product's PartitionedRenderer and call
its render() — the same partition, all-gather and reassembly code the GPU job runs — with the
CPU oracle injected as the renderer of a rank's rows and the host mirror of
rt_deinterleave_bands as the reassembly.  Rank 0's frame must equal the single-process frame.
The reduction bench.py applies to its per-rank numbers runs over the same process group."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import cube_spec, ocam, rt, to_oracle
from raytracer_amd.dist import PartitionedRenderer, deinterleave_numpy, local_rows, reduce_over_ranks

DEPTH = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, W, H, band, q, slots=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cam = rt.Camera(pitch=8.0 / W)
        sc = to_oracle(cube_spec(mirror=False, mesh_reflect=0.5))
        rays = [0]

        def render_rows(rowset, out):
            # this rank's rows in local-buffer order, exactly the mapping rt_rowset documents
            r = np.arange(rowset.n_rows)
            ys = (rowset.first_band + (r // rowset.band_rows) * rowset.band_stride) * rowset.band_rows + r % rowset.band_rows
            assert np.array_equal(ys, local_rows(H, band, rank, world))
            buf = out.numpy()
            for k, y in enumerate(ys):
                f = sc.render(W, H, y0=int(y), y1=int(y) + 1, max_depth=DEPTH, cam=ocam(cam), want=("rgb8",))
                buf[k] = f["rgb8"][0]
                rays[0] += sum(f["counts"].values())

        def deinterleave(gathered, frame):
            cap = rt.rows_per_part(H, band, world)
            frame.copy_(torch.from_numpy(deinterleave_numpy(gathered.numpy().reshape(world, cap, W, 3), H, band)))

        pr = PartitionedRenderer(None, W, H, DEPTH, rank, world, band, torch.device("cpu"),
                                 render_rows=render_rows, deinterleave=deinterleave, dist=dist, frames_in_flight=slots)
        assert pr.frames_in_flight == slots
        if slots > 1:
            # frames in flight: three frames asked for before the first is looked at go to slots 0, 1, 0, each with
            # buffers of its own; every one is the whole frame
            frames = [pr.render_async() for _ in range(3)]
            outs = [f.wait() for f in frames]
            assert outs[0] is outs[2] and outs[0] is not outs[1]
            assert torch.equal(outs[0], outs[1])
            rays[0] = 0                      # the reduction below is about one frame's rays
        out = pr.render(slot=0) if slots > 1 else pr.render()
        vec = torch.tensor([float(rank + 1), float(rays[0])], dtype=torch.float64)
        mx, sm = reduce_over_ranks(vec, dist)
        if rank == 0:
            assert out is pr.frame
            q.put((out.numpy().copy(), float(mx[0]), float(sm[1])))
        else:
            assert out is pr.local
        dist.barrier()
    finally:
        dist.destroy_process_group()


# (world, W, H, band): the last two have H that is not a multiple of band * world, and the 4-rank case
# leaves ranks with different numbers of rows (H = 70, 8-row bands: 9 bands, the last one 6 rows)
@pytest.mark.parametrize("world,W,H,band,slots", [(2, 96, 64, 8, 1), (2, 96, 70, 8, 1), (4, 64, 70, 8, 1), (2, 64, 40, 8, 2)])
@pytest.mark.timeout(300)
def test_partitioned_renderer_over_gloo_reassembles_the_frame(world, W, H, band, slots):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, band, q, slots)) for r in range(world)]
    for p in procs:
        p.start()
    frame, t_max, rays_sum = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cam = rt.Camera(pitch=8.0 / W)
    want = to_oracle(cube_spec(mirror=False, mesh_reflect=0.5)).render(W, H, max_depth=DEPTH, cam=ocam(cam), want=("rgb8",))
    assert np.array_equal(frame, want["rgb8"])
    assert t_max == float(world)                              # MAX over ranks
    assert rays_sum == float(sum(want["counts"].values()))    # SUM over ranks: the parts cast exactly the frame's rays


def test_single_process_renderer_and_reduction_need_no_process_group():
    W, H = 40, 20
    calls = []
    pr = PartitionedRenderer(None, W, H, 1, 0, 1, 8, torch.device("cpu"), render_rows=lambda rs, out: calls.append((rs.first_band, rs.band_rows, rs.band_stride, rs.n_rows)))
    assert pr.render() is pr.local and calls == [(0, 1, 1, H)]
    with pytest.raises(ValueError):
        PartitionedRenderer(None, W, H, 1, 0, 1, 8, torch.device("cpu"), frames_in_flight=rt.MAX_FRAME_SLOTS + 1)
    v = torch.tensor([1.0, 2.0], dtype=torch.float64)
    mx, sm = reduce_over_ranks(v, None)
    assert torch.equal(mx, v) and torch.equal(sm, v)
