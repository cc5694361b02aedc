"""The N > 1 path on CPU: two gloo ranks each produce their bands (here with the CPU oracle
standing in for the renderer), all-gather the padded slices exactly as PartitionedRenderer
does, and rank 0 de-interleaves; the result must equal the single-process frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import cube_spec, ocam, rt, to_oracle
from raytracer_amd.dist import deinterleave_numpy, local_rows

W, H, DEPTH, BAND = 96, 70, 2, 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cam = rt.Camera(pitch=8.0 / W)
        sc = to_oracle(cube_spec(mirror=False, mesh_reflect=0.5))
        ys = local_rows(H, BAND, rank, world)
        cap = rt.rows_per_part(H, BAND, world)
        local = np.zeros((cap, W, 3), np.uint8)
        for k, y in enumerate(ys):          # this rank's rows, in local-buffer order
            local[k] = sc.render(W, H, y0=int(y), y1=int(y) + 1, max_depth=DEPTH, cam=ocam(cam), want=("rgb8",))["rgb8"][0]
        gathered = torch.empty((world * cap, W, 3), dtype=torch.uint8)   # concatenated form, as PartitionedRenderer
        dist.all_gather_into_tensor(gathered, torch.from_numpy(local))
        if rank == 0:
            q.put(deinterleave_numpy(gathered.numpy().reshape(world, cap, W, 3), H, BAND))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_band_partition_reassembles_the_frame():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cam = rt.Camera(pitch=8.0 / W)
    want = to_oracle(cube_spec(mirror=False, mesh_reflect=0.5)).render(W, H, max_depth=DEPTH, cam=ocam(cam), want=("rgb8",))["rgb8"]
    assert np.array_equal(frame, want)
