"""Two ranks on ONE GPU: each process builds the scene in its own context, constructs the product's
PartitionedRenderer(world = 2) on cuda:0 and calls render() — the real rt_render_rows_device and
rt_deinterleave_bands kernels, the real stream ordering, the real band partition — with the slices exchanged
over a gloo process group (RCCL refuses two ranks on one device, so the collective is the one part that differs
from the 8-GPU job: the gather is staged through host memory by a shim with torch.distributed's signature).
Rank 0's frame must equal the frame rendered whole, for a camera that changes between frames."""
import os
import socket

import numpy as np
import pytest

from common import atrium_spec, rt, to_product

pytestmark = pytest.mark.gpu

W, H, D, BAND = 320, 200, 3, 16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cams():
    return [rt.Camera(origin=(0.3 * k - 0.3, -1.5, -6.0), pitch=8.0 / W, off_x=-4.0, off_y=-3.4) for k in range(3)]


from raytracer_amd.dist import HostStagedGather as _HostStagedGather


def _worker_in_flight(rank, world, port, q):
    """Two frame slots of one context per rank, a camera per slot, five frames asked for before the first is waited for."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from raytracer_amd.dist import PartitionedRenderer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        with rt.Context(0) as ctx:
            to_product(ctx, atrium_spec(detail=0.02))
            ctx.set_option(rt.OPT_TIMING, 0)
            pr = PartitionedRenderer(ctx, W, H, D, rank, world, BAND, torch.device("cuda", 0), dist=_HostStagedGather(dist, torch),
                                     frames_in_flight=2)
            assert pr.gather_stream is not None and pr.gather_stream is not pr.stream
            cams = _cams()
            with torch.cuda.stream(torch.cuda.Stream()):
                frames = []
                for k in range(5):
                    ctx.set_camera(cams[k % 2])          # slot k % 2 always gets camera k % 2
                    frames.append(pr.render_async())
                outs = [f.wait() for f in frames]
                torch.cuda.current_stream().synchronize()
                if rank == 0:
                    q.put([outs[0].cpu().numpy().copy(), outs[1].cpu().numpy().copy(), outs[4].cpu().numpy().copy()])
            dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from raytracer_amd.dist import PartitionedRenderer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        with rt.Context(0) as ctx:
            to_product(ctx, atrium_spec(detail=0.02))
            ctx.set_option(rt.OPT_TIMING, 0)
            pr = PartitionedRenderer(ctx, W, H, D, rank, world, BAND, torch.device("cuda", 0), dist=_HostStagedGather(dist, torch))
            frames = []
            for cam in _cams():
                ctx.set_camera(cam)
                out = pr.render()
                torch.cuda.current_stream().synchronize()
                if rank == 0:
                    frames.append(out.cpu().numpy().copy())
            if rank == 0:
                q.put(frames)
            dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_with_two_frames_in_flight():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_in_flight, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        frames = q.get(timeout=240)
    finally:
        for p in procs:
            p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    with rt.Context(0) as c:
        to_product(c, atrium_spec(detail=0.02))
        want = []
        for cam in _cams()[:2]:
            c.set_camera(cam)
            want.append(c.render_frame(W, H, max_depth=D, want=("rgb8",))["rgb8"])
    assert np.array_equal(frames[0], want[0]) and np.array_equal(frames[1], want[1]) and np.array_equal(frames[2], want[0])
    assert not np.array_equal(want[0], want[1])


@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu_reassemble_the_frame():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        frames = q.get(timeout=240)
    finally:
        for p in procs:
            p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    with rt.Context(0) as c:
        to_product(c, atrium_spec(detail=0.02))
        for cam, got in zip(_cams(), frames):
            c.set_camera(cam)
            want = c.render_frame(W, H, max_depth=D, want=("rgb8",))["rgb8"]
            assert np.array_equal(got, want)
    assert not np.array_equal(frames[0], frames[1])


@pytest.mark.timeout(400)
def test_bench_partitioned_job_two_ranks_on_one_gpu():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per process), rehearsed on one GPU:
    the gloo backend with host-staged slices stands in for RCCL, everything else is the job's own code — band partition,
    a frame at a time and four frames in flight per rank, the max/sum reduction over ranks, the one JSON line from rank 0.  The rays of the two
    halves must add up to the frame's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--width", "640", "--height", "360"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=300, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    single = json.loads(one.stdout.strip().splitlines()[-1])
    env = dict(os.environ, OMP_NUM_THREADS="2")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--share-device"] + common, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                         # rank 0 alone prints
    d = json.loads(lines[0])
    # one mode at every rank count: a frame at a time is the headline, four frames in flight ride beside it
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["frames_in_flight"] == single["config"]["frames_in_flight"] == 1
    assert d["config"]["rays_per_frame"] == single["config"]["rays_per_frame"]
    assert d["value"] > 0 and abs(d["frame_latency_ms"] - d["ms_per_step"]) < 1e-9 and d["first_frame_ms"] > d["ms_per_step"] and d["order_frame_ms"] > d["ms_per_step"]
    assert d["pipelined"]["frames_in_flight"] == 4 and d["pipelined"]["value"] > 0 and single["pipelined"]["value"] > 0
