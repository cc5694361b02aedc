import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, raytracer_amd as rt
from raytracer_amd import workloads
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
kw = {}
if name == "soup":
    kw = dict(W=1920, H=1080, max_depth=4)
wl = workloads.by_name(name, **kw)
for builder in (1, 2):
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_BUILDER, builder)
        ctx.set_option(rt.OPT_COUNT_TESTS, 1)
        ctx.set_option(rt.OPT_TIMING, 1)
        wl.load(ctx)
        out = ctx.render_frame(wl.W, wl.H, max_depth=wl.max_depth)
        s = ctx.stats()
        print(name, "builder", builder, {k: s[k] for k in ("bvh_depth", "ms_build", "box_tests", "tri_tests", "ms_trace", "ms_traverse") if k in s}, flush=True)
