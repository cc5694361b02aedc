#!/usr/bin/env python3
"""profiles/wave_times.py --lib raytracer_amd/variants/times.so : one plain frame of the bench workload with per-wave clocks
(a -DWF_TIMES=1 build; RT_WF_TIMES makes wavefront_frame print, per traversal launch, when the waves found nothing left
to claim and when they exited) -> stderr."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--workload", default="sponza")
a = ap.parse_args()
os.environ["RT_AMD_LIB"] = os.path.abspath(a.lib)
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
wl = workloads.by_name(a.workload)
with rt.Context(0) as ctx:
    wl.load(ctx)
    ctx.set_option(rt.OPT_GRAPH, 0)
    buf = torch.zeros((wl.H, wl.W, 3), dtype=torch.uint8, device="cuda")
    rs = rt.RowSet(0, 1, 1, wl.H)
    for k in range(3):
        if k == 2:
            os.environ["RT_WF_TIMES"] = "1"
        ctx.render_rows_device(wl.W, wl.H, rs, wl.max_depth, d_rgb8=buf.data_ptr())
        ctx.synchronize()
