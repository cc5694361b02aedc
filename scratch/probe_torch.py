import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import raytracer_amd as rt
from common import golden_cases, to_product, soup_spec
mode = sys.argv[1]
def maps():
    s = set()
    for l in open("/proc/self/maps"):
        if "libamdhip64" in l or "libhsa-runtime" in l:
            s.add(l.split()[-1])
    return sorted(s)
print("before", maps(), flush=True)
if mode == "mesh":
    for i in range(int(sys.argv[2])):
        spec = soup_spec(2000 + i, seed=i)
        with rt.Context(0) as ctx:
            to_product(ctx, spec)
            ctx.set_camera(rt.Camera())
            ctx.render_frame(64, 48, max_depth=2)
elif mode == "cube":
    spec, W, H, b, cam = golden_cases()["cube_refl_d4_200"]
    for i in range(int(sys.argv[2])):
        with rt.Context(0) as ctx:
            to_product(ctx, spec); ctx.set_camera(cam); ctx.render_frame(W, H, max_depth=b)
import torch
try:
    x = torch.zeros(4, device="cuda")
    print("torch ok", flush=True)
except Exception as e:
    print("torch FAIL", e, flush=True)
print("after", maps(), flush=True)
