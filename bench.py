#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X trace path.

Metric: Mrays/s (primary + shadow + reflect, counted as the reference would
cast them) on the Sponza-class workload at 1920x1080 with hard shadows and 4 reflection
bounces, on N GPUs of one node (frame tile-partitioned, scene replicated, RGB8 slices
gathered with RCCL).  One "step" = one full frame: every traversal, ray-generation
and shading launch of all bounce levels and, for N > 1, the gather + de-interleave.  The scene (triangles,
LBVH) is resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3      # MI355X vector FP32: 256 CU x 128 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0
# algorithmic flops per intersection test (Moller-Trumbore main.cpp:109-129; slab test;
# sphere main.cpp:56-68; plane main.cpp:84-92)
FLOP_TRI, FLOP_BOX, FLOP_ANALYTIC = 46, 20, 17


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sponza", choices=["sponza", "cube", "soup", "atrium2"])
    ap.add_argument("--obj", default="", help="an OBJ file to render in place of the Sponza stand-in (default: $RT_SPONZA_OBJ when it exists)")
    ap.add_argument("--no-extras", action="store_true", help="skip the share table, the eager loop and the second stand-in (N = 1 extras)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--depth", type=int, default=-1)
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--leaf-max", type=int, default=0)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames a rank keeps in flight in the timed loop (one frame slot and stream each); 0 = 1, at every rank count")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (slices staged through host memory) is for rehearsing the partitioned job where RCCL cannot run")
    ap.add_argument("--share-device", action="store_true", help="every rank uses GPU 0 (rehearsal of N ranks on a one-GPU box; RCCL refuses that, use --backend gloo)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra loop with four frames in flight")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch of every frame instead of replaying a hipGraph")
    ap.add_argument("--builder", type=int, default=0, help="0 auto (= 3), 1 Karras LBVH, 2 PLOC, 3 binned SAH")
    ap.add_argument("--lib", default="", help="another build of librt_amd.so to load (A/B runs of kernel variants)")
    ap.add_argument("--pipeline", type=int, default=0, help="0 auto, 1 fused bounce kernel, 2 wavefront stages")
    return ap.parse_args()


def cpu_baseline(wl, threads):
    """Time the CPU restatement of the reference (oracle/, OpenMP over rows, its own BVH) on a
    bounded sample of the same workload: one full frame."""
    from oracle import oracle as O   # checker used only as the reported CPU baseline
    sc = O.scene_from_workload(wl, bvh=True)
    cam = O.camera_from(wl.camera)
    step = 1
    t0 = time.perf_counter()
    r = sc.render(wl.W, wl.H, max_depth=wl.max_depth, cam=cam, use_bvh=True, threads=threads, x_step=step, y_step=step,
                  want=())
    dt = time.perf_counter() - t0
    rays = r["counts"]["primary"] + r["counts"]["shadow"] + r["counts"]["reflect"]
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "host_cpus": os.cpu_count(),       # what the machine has; `cores` is the threads this figure used
           "sample": "one full %dx%d frame (%d rays) through the CPU restatement with its median-split BVH, "
                     "OpenMP over rows" % (wl.W, wl.H, rays), "seconds": dt}
    # the same hierarchy port on ONE thread, like the reference's own single-threaded loop (main.cpp:262-277), on every
    # fourth row and column
    lat1 = 4
    t0 = time.perf_counter()
    r = sc.render(wl.W, wl.H, max_depth=wl.max_depth, cam=cam, use_bvh=True, threads=1, x_step=lat1, y_step=lat1, want=())
    dt = time.perf_counter() - t0
    rays = r["counts"]["primary"] + r["counts"]["shadow"] + r["counts"]["reflect"]
    out["bvh_port_1core"] = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1,
                             "sample": "every %dth row and column (%d rays), the CPU restatement with its BVH on one thread" % (lat1, rays),
                             "seconds": dt}
    # The reference's own algorithm is a linear scan over all primitives: time it single-threaded
    # on a sparse lattice so the number stays bounded.
    lat = 96
    t0 = time.perf_counter()
    r = sc.render(wl.W, wl.H, max_depth=wl.max_depth, cam=cam, use_bvh=False, threads=1, x_step=lat, y_step=lat, want=())
    dt = time.perf_counter() - t0
    rays = r["counts"]["primary"] + r["counts"]["shadow"] + r["counts"]["reflect"]
    out["linear_scan_1core"] = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1,
                                "sample": "every %dth row and column (%d rays), brute-force scan as main.cpp:149-156" % (lat, rays),
                                "seconds": dt}
    return out


def timed_frames(render, sync, steps, warmup):
    for _ in range(warmup):
        render()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        render()
    sync()
    return (time.perf_counter() - t0) / steps * 1e3


def share_table(ctx, n_slots, wl, band_rows, torch, dev, steps=10):
    """One rank's share of the frame, measured on this GPU: part 0 of 1, 2, 4 and 8 parts (16-row bands
    round-robin), graph replay, RGB8 only, ms per frame, with n_slots frames in flight (one frame slot of the context,
    one stream and one buffer each, frames round-robin).  What strong scaling can reach before the gather."""
    import raytracer_amd as rt
    out = {}
    slots = [(torch.zeros((wl.H, wl.W, 3), dtype=torch.uint8, device=dev), torch.cuda.Stream(device=dev)) for _ in range(n_slots)]
    for parts in (1, 2, 4, 8):
        rs = rt.band_rowset(wl.H, band_rows, 0, parts) if parts > 1 else rt.RowSet(0, 1, 1, wl.H)
        turn = [0]

        def render():
            k = turn[0] % n_slots
            turn[0] += 1
            buf, stream = slots[k]
            ctx.render_rows_device(wl.W, wl.H, rs, wl.max_depth, d_rgb8=buf.data_ptr(), stream=stream.cuda_stream, slot=k)

        out[str(parts)] = timed_frames(render, torch.cuda.synchronize, steps * n_slots, 3 * n_slots)
    return out


def host_frame_times(ctx, wl, steps):
    """The drop-in call of the reference's seam (rt_render_frame: the frame lands in a HOST buffer, main.cpp:259, 273, 281),
    8-bit colour only, ms per frame: into an ordinary (pageable) array through the pinned staging buffer, and into a
    buffer from rt_alloc_host_buffer that the device writes directly.  Never the benchmark's value."""
    out = {}
    image = np.zeros((wl.H, wl.W, 3), np.uint8)
    pinned = ctx.alloc_host_buffer((wl.H, wl.W, 3))
    for name, buf in (("host_frame_ms", image), ("host_frame_pinned_ms", pinned)):
        for _ in range(2):
            ctx.render_frame_into(wl.W, wl.H, buf, max_depth=wl.max_depth)
        walls, after = [], []
        for _ in range(steps):
            t0 = time.perf_counter()
            ctx.render_frame_into(wl.W, wl.H, buf, max_depth=wl.max_depth)
            walls.append((time.perf_counter() - t0) * 1e3)
            after.append(ctx.stats()["staging_ms"])
        # a synchronous call per frame: the median (one call in a few dozen takes milliseconds longer on a busy host),
        # with the mean beside it
        out[name] = float(np.median(walls))
        out[name.replace("_ms", "_mean_ms")] = float(np.mean(walls))
        # of which, after the device finished the frame: copies into the staging buffer and on into the array / the one copy
        out[name.replace("_ms", "_handover_ms")] = float(np.median(after))
    ctx.free_host_buffer(pinned)
    return out


def measured_valu(workload, world, resolution):
    """Issue utilisation, mean active lanes and the FMA / MUL / ADD share of the traversal kernels from the committed
    counter passes (profiles/valu_counters.json, written by profiles/derive.py); like the traffic figure it is only
    reported for exactly the kernel sources, workload and frame size it was measured on."""
    from raytracer_amd.build import source_hash
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "valu_counters.json")))
    except Exception:
        return None, "none"
    if rec.get("source_hash") != source_hash():
        return None, "profiles/valu_counters.json is stale (measured on other kernel sources)"
    if world != 1 or workload not in rec or rec[workload].get("resolution") != list(resolution):
        return None, "profiles/valu_counters.json has no entry for this run"
    return rec[workload].get("kernels"), "profiles/valu_counters.json (" + rec[workload].get("source", "") + ")"


def measured_traffic(workload, world, resolution):
    """HBM-side bytes per frame of the traversal kernel from the committed counter passes
    (profiles/hbm_traffic.json, written by profiles/derive.py from rocprofv3 FETCH_SIZE / WRITE_SIZE
    runs).  Counters cannot be read inside this process, so the figure is only reported when the file was
    measured on exactly these kernel sources (its source_hash equals the tree's); otherwise null."""
    from raytracer_amd.build import source_hash
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(tpath))
    except Exception:
        return (None, None), "none"
    if rec.get("source_hash") != source_hash():
        return (None, None), "profiles/hbm_traffic.json is stale (measured on other kernel sources)"
    if world != 1 or workload not in rec or rec[workload].get("resolution") != list(resolution):
        return (None, None), "profiles/hbm_traffic.json has no entry for this run"
    return (rec[workload].get("bytes_per_frame"), rec[workload].get("bytes_per_frame_as_counted")), "profiles/hbm_traffic.json (" + rec[workload].get("source", "") + ")"


def main():
    a = parse()
    if a.lib:
        os.environ["RT_AMD_LIB"] = os.path.abspath(a.lib)
    import torch
    import raytracer_amd as rt
    from raytracer_amd import workloads
    from raytracer_amd.dist import PartitionedRenderer, reduce_over_ranks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with that many ranks" % a.gpus)
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace path has no CPU fallback")
    if a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = gather = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            from raytracer_amd.dist import HostStagedGather
            dist.init_process_group("gloo")
            gather = HostStagedGather(dist, torch)

    kw = {}
    if a.width: kw["W"] = a.width
    if a.height: kw["H"] = a.height
    if a.depth >= 0: kw["max_depth"] = a.depth
    if a.obj:
        kw["obj"] = a.obj
        if a.workload != "sponza" or not os.path.exists(a.obj):
            raise SystemExit("--obj needs an existing file and the sponza workload")
    wl = workloads.by_name(a.workload, **kw)

    def make_context():
        c = rt.Context(local_rank)
        if a.leaf_max:
            c.set_option(rt.OPT_LEAF_MAX, a.leaf_max)
        if a.pipeline:
            c.set_option(rt.OPT_PIPELINE, a.pipeline)
        if a.builder:
            c.set_option(rt.OPT_BUILDER, a.builder)
        if a.no_graph:
            c.set_option(rt.OPT_GRAPH, 0)
        wl.load(c)
        return c

    # ONE mode for every rank count: a frame at a time, as the reference's setup() renders (main.cpp:262-277), so a 1 -> N curve
    # compares like with like.  On one GPU that is also the fastest way to run whole frames (the traversal launches hand out
    # their longest rays first, RT_OPT_CLAIM_ORDER, and leave little for a second frame to fill: four frames in flight are
    # 2 % slower).  A rank's share of a partitioned frame is different: an eighth of 1080p has fewer rays than the GPU has
    # lanes, and four frames in flight (one frame slot, stream and buffer set each) run 1.8x faster.  That rate is reported
    # beside the headline as `pipelined`, at every rank count, never as `value`.
    in_flight = a.frames_in_flight if a.frames_in_flight > 0 else 1
    ctx = make_context()
    t_first = time.perf_counter()
    pr = PartitionedRenderer(ctx, wl.W, wl.H, wl.max_depth, rank, world, a.band_rows, dev, dist=gather, frames_in_flight=in_flight)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # the first frame a new context renders (plain unit order; the first launches load their code), and the second: the
    # layout repeats, so it is preceded by the counting pass that makes the claim order and by the timing runs that pick the
    # camera-ray kernel (RT_OPT_CLAIM_ORDER).  Every later frame of the layout is a `step`.
    with torch.cuda.stream(pr.stream):
        pr.render(slot=0)
        sync_all()
        first_frame_ms = (time.perf_counter() - t_first) * 1e3
        t_second = time.perf_counter()
        pr.render(slot=0)
        sync_all()
        order_frame_ms = (time.perf_counter() - t_second) * 1e3

    # the timed loop runs without per-launch device events, so that an unchanged frame replays as one
    # hipGraph; kernel times come from one more frame with the events on, outside the timed region
    ctx.set_option(rt.OPT_TIMING, 0)

    def one_at_a_time(renderer, steps, warmup):
        # The loop runs on the renderer's own stream (made current here), so a frame costs no cross-stream events.
        with torch.cuda.stream(renderer.stream):
            for _ in range(warmup):
                renderer.render()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(steps):
                renderer.render()
            sync_all()
            return time.perf_counter() - t0

    def frames_in_flight(renderer, n, steps, warmup):
        # Frames go to the slots round-robin; the consumer of a frame (here: nothing but the wait) runs n - 1
        # frames behind the producer, as a display or an encoder would.
        from collections import deque
        pending = deque()

        def step():
            pending.append(renderer.render_async())
            if len(pending) >= n:
                pending.popleft().wait()

        def drain():
            while pending:
                pending.popleft().wait()

        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            for _ in range(max(warmup, 3) * n):      # every slot replays a captured graph from its third frame on
                step()
            drain()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            drain()
            sync_all()
            return time.perf_counter() - t0

    if in_flight == 1:
        dt = one_at_a_time(pr, a.steps, a.warmup)
        frame_latency_ms = dt / a.steps * 1e3
    else:
        dt = frames_in_flight(pr, in_flight, a.steps, a.warmup)
        # the same frames one at a time: what one frame takes from request to result
        frame_latency_ms = one_at_a_time(pr, min(a.steps, 20), 2) / min(a.steps, 20) * 1e3
    pipelined = None
    if in_flight == 1 and not a.no_pipelined and wl.W * wl.H <= 10_000_000:
        pr4 = PartitionedRenderer(ctx, wl.W, wl.H, wl.max_depth, rank, world, a.band_rows, dev, dist=gather, frames_in_flight=4)
        pipelined = frames_in_flight(pr4, 4, a.steps, a.warmup)
    with torch.cuda.stream(pr.stream):
        ctx.set_option(rt.OPT_TIMING, 1)
        pr.render(slot=0)
        sync_all()
    st = ctx.stats()   # one frame of this rank: ray counts + device-event kernel times

    rays_local = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
    vec = torch.tensor([dt, float(rays_local), float(st["rays_primary"]), float(st["rays_shadow"]), float(st["rays_reflect"]),
                        st["ms_trace"], st["ms_frame"], frame_latency_ms, pipelined if pipelined is not None else 0.0, first_frame_ms, order_frame_ms],
                       dtype=torch.float64, device=dev)
    if dist is not None and a.backend != "nccl":
        vec = vec.cpu()
    mx, sm = reduce_over_ranks(vec, dist)        # times: max over ranks; ray counts: sum over ranks
    dt = float(mx[0])
    rays = float(sm[1]); rp, rs, rr = float(sm[2]), float(sm[3]), float(sm[4])
    ms_trace, ms_frame = float(mx[5]), float(mx[6])
    frame_latency_ms, first_frame_ms, order_frame_ms = float(mx[7]), float(mx[9]), float(mx[10])
    pipelined = float(mx[8]) if pipelined is not None else None

    # ---- roofline of the dominant kernel (k_bounce, all levels of one frame), rank 0's share:
    # algorithmic flops from a counting pass outside the timed region, duration from the
    # device events the library records around those launches on the launch stream.
    ctx.set_option(rt.OPT_COUNT_TESTS, 1)
    pr.render(slot=0)
    sync_all()
    cst = ctx.stats()
    ctx.set_option(rt.OPT_COUNT_TESTS, 0)
    # Box tests are counted two ways: `box_tests` charges the four child slots of every node step (what the kernel
    # executes: a slot that holds no child is tested like the others), `box_tests_occupied` only the slots that hold a
    # child.  The fraction published as `frac` is the LOWER one, from the occupied slots; `frac_charged` is the other.
    # (The camera rays go through the packet kernel, where every ray of a tile tests every node the tile visits.)
    flops_charged = cst["box_tests"] * FLOP_BOX + cst["tri_tests"] * FLOP_TRI + cst["analytic_tests"] * FLOP_ANALYTIC
    occupied = cst["box_tests_occupied"] or cst["box_tests"]      # the fused and linear-scan kernels have no empty slots to count
    flops = occupied * FLOP_BOX + cst["tri_tests"] * FLOP_TRI + cst["analytic_tests"] * FLOP_ANALYTIC
    # k_trace (LBVH scenes: traversal alone, timed by its own device events) or k_bounce (small scenes)
    kernel_ms = st["ms_traverse"] if st["ms_traverse"] > 0 else st["ms_trace"]
    achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    achieved_charged = flops_charged / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    valu, valu_source = measured_valu(a.workload, world, (wl.W, wl.H))
    (traffic, traffic_as_counted), traffic_source = measured_traffic(a.workload, world, (wl.W, wl.H))
    compulsory = wl.W * wl.H * 3 + wl.n_tris * (48 + 16) + st["bvh_nodes"] * 64
    roofline = {
        "bound": "valu_fp32",
        "kernel": ("k_trace (closest + any-hit, all %d levels of one frame)" if st["accel"] else
                   "k_bounce (all %d levels of one frame)") % (wl.max_depth + 1),
        "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
        # fabric-side bytes per frame of the traversal kernels: FETCH_SIZE doubled (the gfx950 correction, calibrated on wide
        # streaming reads, not on this kernel's scattered 64-byte fetches) + WRITE_SIZE; and as the counters report them
        "traffic": traffic, "traffic_as_counted": traffic_as_counted, "traffic_source": traffic_source,
        "frac_charged": achieved_charged / PEAK_FP32_TFLOPS, "flops_per_frame_charged": flops_charged,
        "valu": valu, "valu_source": valu_source,
        "flops_per_frame": flops, "box_tests": cst["box_tests"], "box_tests_occupied": occupied, "tri_tests": cst["tri_tests"],
        "analytic_tests": cst["analytic_tests"], "kernel_ms_per_frame": kernel_ms,
        "all_bounce_stages_ms_per_frame": st["ms_trace"],
        "hbm": {"compulsory_bytes_per_frame": compulsory,
                "achieved_GBs": compulsory / (st["ms_frame"] * 1e-3) / 1e9 if st["ms_frame"] > 0 else 0.0,
                "peak_GBs": PEAK_HBM_GBS},
    }

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+shadow+reflect)", "value": rays * a.steps / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name, "triangles": wl.n_tris, "resolution": [wl.W, wl.H], "bounces": wl.max_depth,
                       "accel": "lbvh" if st["accel"] else "linear", "bvh_depth": st["bvh_depth"],
                       "builder": {0: "binned SAH", 1: "Karras LBVH", 2: "PLOC", 3: "binned SAH"}[a.builder] if st["accel"] else "none",
                       "bvh_sah_cost": round(st["bvh_sah_cost"], 2), "bvh_refs": st["bvh_refs"],
                       "leaf_max": st["bvh_leaf_max"], "partition": "%d-row bands round-robin" % a.band_rows if world > 1 else "none",
                       "frames_in_flight": in_flight,
                       # every step redraws one still frame, as the reference's loop does (main.cpp:284-287): the work units are
                       # claimed in the order recorded for this view (RT_OPT_CLAIM_ORDER; `plain_order` is the rate without it)
                       "view": "still", "claim_order": "recorded for this view",
                       "rays_per_frame": {"primary": rp, "shadow": rs, "reflect": rr}},
            "device_ms_per_frame": ms_frame, "lbvh_build_ms": st["ms_build"],
            "roofline": roofline,
        }
        # one frame at a time (what the reference's setup() does, main.cpp:262-277), at every rank count; ms_per_step and
        # value are the rate with config.frames_in_flight frames in flight
        out["frame_latency_ms"] = frame_latency_ms
        out["single_frame_value"] = rays / frame_latency_ms / 1e3
        # the first frame a new context renders (scene resident; plain unit order, code loading) and the second, which makes
        # the claim order (counting pass + kernel choice + the frame); every later frame of the layout is a `step`
        out["first_frame_ms"] = first_frame_ms
        out["order_frame_ms"] = order_frame_ms
        if pipelined is not None:
            out["pipelined"] = {"frames_in_flight": 4, "ms_per_step": pipelined / a.steps * 1e3, "value": rays * a.steps / pipelined / 1e6,
                                "unit": "Mrays/s", "note": "the same job with four frames in flight per rank; not the headline"}
        if world == 1 and not a.no_extras:
            # the same frames with every launch issued one by one (a frame that changes cannot replay a graph)
            ctx.set_option(rt.OPT_TIMING, 0)
            ctx.set_option(rt.OPT_GRAPH, 0)
            with torch.cuda.stream(pr.stream):
                out["eager_ms_per_step"] = timed_frames(pr.render, sync_all, a.steps, 2)
            ctx.set_option(rt.OPT_GRAPH, 0 if a.no_graph else 1)
            # the same frames with the work units handed out in plain order: what a frame takes without the claim order, which
            # is a record of THIS view's longest rays (it is what a moving camera gets: the order's worth fades with the view)
            ctx.set_option(rt.OPT_CLAIM_ORDER, 0)
            with torch.cuda.stream(pr.stream):
                t_plain = timed_frames(pr.render, sync_all, a.steps, 3)
            ctx.set_option(rt.OPT_CLAIM_ORDER, 1)
            out["plain_order"] = {"ms_per_step": t_plain, "value": rays / t_plain / 1e3, "unit": "Mrays/s",
                                  "note": "RT_OPT_CLAIM_ORDER off: units claimed in screen order; not the headline"}
            out.update(host_frame_times(ctx, wl, min(a.steps, 10)))
            if st["accel"]:
                out["share_ms"] = share_table(ctx, 1, wl, a.band_rows, torch, dev)
                out["share_ms_4_in_flight"] = share_table(ctx, 4, wl, a.band_rows, torch, dev)
            if a.workload == "sponza":
                # the second stand-in (irregular triangle sizes, slivers, overlapping cloth) through the same loop
                wl2 = workloads.atrium2(W=wl.W, H=wl.H, max_depth=wl.max_depth)
                with rt.Context(local_rank) as ctx2:
                    wl2.load(ctx2)
                    ctx2.set_option(rt.OPT_TIMING, 0)
                    pr2 = PartitionedRenderer(ctx2, wl2.W, wl2.H, wl2.max_depth, 0, 1, a.band_rows, dev)
                    with torch.cuda.stream(pr2.stream):
                        ms2 = timed_frames(pr2.render, sync_all, a.steps, a.warmup)
                        ctx2.set_option(rt.OPT_TIMING, 1)
                        pr2.render(); sync_all()
                    s2 = ctx2.stats()
                r2 = s2["rays_primary"] + s2["rays_shadow"] + s2["rays_reflect"]
                out["other_workloads"] = {wl2.name: {"value": r2 / ms2 / 1e3, "unit": "Mrays/s", "ms_per_step": ms2,
                                                     "rays_per_frame": r2, "triangles": wl2.n_tris, "lbvh_build_ms": s2["ms_build"]}}
        if world == 1 and not a.no_cpu_baseline:
            threads = a.cpu_threads or min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(wl, threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
