"""profiles/host_frame_probe.py: where the time of rt_render_frame goes for a pageable and a library (pinned) buffer:
wall time per call, the part after the device finished the frame (rt_stats.staging_ms), and the device's own frame time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import workloads

wl = workloads.sponza(W=1920, H=1080)
with rt.Context(0) as ctx:
    wl.load(ctx)
    image = np.zeros((wl.H, wl.W, 3), np.uint8)
    pinned = ctx.alloc_host_buffer((wl.H, wl.W, 3))
    for name, buf in (("pageable", image), ("pinned", pinned), ("pageable", image)):
        for _ in range(4):
            ctx.render_frame_into(wl.W, wl.H, buf, max_depth=wl.max_depth)
        walls, stag = [], []
        for _ in range(30):
            t0 = time.perf_counter()
            ctx.render_frame_into(wl.W, wl.H, buf, max_depth=wl.max_depth)
            walls.append((time.perf_counter() - t0) * 1e3)
            stag.append(ctx.stats()["staging_ms"])
        walls, stag = np.array(walls), np.array(stag)
        print("%-9s wall median %.3f min %.3f max %.3f | after device: median %.3f min %.3f max %.3f | before: %.3f" % (
            name, np.median(walls), walls.min(), walls.max(), np.median(stag), stag.min(), stag.max(), np.median(walls - stag)))
    ctx.free_host_buffer(pinned)
