#!/usr/bin/env python3
"""profiles/coherence.py [--workload sponza] [--lib path]: one counting frame of the bench workload with RT_WF_DEBUG set;
the library prints, per ray kind, how alike the lanes of a wave are during the node steps of k_trace (lanes at the same
node, at the first lane's node, in the first lane's direction octant) to stderr.  Redirect stderr into profiles/."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sponza")
ap.add_argument("--lib", default="")
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
a = ap.parse_args()
if a.lib:
    os.environ["RT_AMD_LIB"] = os.path.abspath(a.lib)
os.environ["RT_WF_DEBUG"] = "1"
import torch
import raytracer_amd as rt
from raytracer_amd import workloads

kw = {}
if a.width: kw["W"] = a.width
if a.height: kw["H"] = a.height
wl = workloads.by_name(a.workload, **kw)
dev = torch.device("cuda", 0)
with rt.Context(0) as ctx:
    wl.load(ctx)
    ctx.set_option(rt.OPT_COUNT_TESTS, 1)
    buf = torch.zeros((wl.H, wl.W, 3), dtype=torch.uint8, device=dev)
    ctx.render_rows_device(wl.W, wl.H, rt.RowSet(0, 1, 1, wl.H), wl.max_depth, d_rgb8=buf.data_ptr())
    st = ctx.stats()
    print("[coherence] %s %dx%d depth %d: rays %d / %d / %d, box tests charged %d, tri tests %d" % (
        wl.name, wl.W, wl.H, wl.max_depth, st["rays_primary"], st["rays_shadow"], st["rays_reflect"], st["box_tests"], st["tri_tests"]),
        file=sys.stderr)
