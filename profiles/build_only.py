"""profiles/build_only.py BUILDER workload...: build the hierarchy of each workload (no frame is rendered) and print its facts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracer_amd as rt
from raytracer_amd import workloads

builder = int(sys.argv[1])
for name in sys.argv[2:]:
    wl = workloads.by_name(name, W=64, H=36)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_BUILDER, builder)
        wl.load(ctx)
        st = ctx.stats()
    print("%-10s builder %d  refs %8d  SAH %8.2f  depth %3d  build %8.2f ms" % (name, builder, st["bvh_refs"], st["bvh_sah_cost"], st["bvh_depth"], st["ms_build"]), flush=True)
