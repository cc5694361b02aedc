"""profiles/share_frames.py PARTS [frames] [W H]: render part 0 of PARTS (16-row bands round-robin) of the Sponza stand-in
frame a few times, for a kernel trace of one rank's share of a partitioned frame
(rocprofv3 --kernel-trace -- python3 profiles/share_frames.py 8)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
wl = workloads.sponza(W=W, H=H)
ctx = rt.Context(0)
wl.load(ctx)
ctx.set_option(rt.OPT_TIMING, 0)
if os.environ.get("RT_NO_GRAPH"):
    ctx.set_option(rt.OPT_GRAPH, 0)
rs = rt.band_rowset(H, 16, 0, parts) if parts > 1 else rt.RowSet(0, 1, 1, H)
buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
for k in range(frames + 3):
    if k == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.render_rows_device(W, H, rs, wl.max_depth, d_rgb8=buf.data_ptr(), stream=s.cuda_stream)
torch.cuda.synchronize()
print("part 0 of %d: %.3f ms per frame" % (parts, (time.perf_counter() - t0) / frames * 1e3))
