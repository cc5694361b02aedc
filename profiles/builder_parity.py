"""profiles/builder_parity.py [workload]: every hierarchy builder must return the same frame (ids bit-exact, the
same ray counts); prints each builder's SAH cost and traversal time next to it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracer_amd as rt
from raytracer_amd import workloads

name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
wl = workloads.by_name(name, W=W, H=H)
ref = None
for builder in (rt.BUILDER_PLOC, rt.BUILDER_LBVH, rt.BUILDER_SAH):
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_BUILDER, builder)
        ctx.set_option(rt.OPT_TIMING, 1)
        wl.load(ctx)
        for _ in range(3):
            out = ctx.render_frame(W, H, max_depth=wl.max_depth, want=("rgb8", "hit_id"))
        rgb, ids = out["rgb8"], out["hit_id"]
        st = ctx.stats()
    rays = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
    print("builder %d  SAH %.2f  build %.1f ms  traverse %.3f ms  rays %d  depth %d" % (builder, st["bvh_sah_cost"], st["ms_build"], st["ms_traverse"], rays, st["bvh_depth"]), flush=True)
    if ref is None:
        ref = (ids.copy(), rgb.copy(), rays)
    else:
        assert rays == ref[2], "ray counts differ"
        assert np.array_equal(ids, ref[0]), "hit ids differ: %d pixels" % int((ids != ref[0]).sum())
        assert np.array_equal(rgb, ref[1]), "pixels differ"
print("all builders agree")
