"""profiles/share_probe.py [parts]: one rank's share (part 0 of `parts`, 16-row bands) of the 1080p stand-in frame, one frame at
a time, as bench.py's share table renders it - for a kernel trace of what a small share's frame consists of
(rocprofv3 --kernel-trace --stats -- python3 profiles/share_probe.py 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = workloads.sponza(W=1920, H=1080)
dev = torch.device("cuda", 0)
with rt.Context(0) as ctx:
    wl.load(ctx)
    rs = rt.band_rowset(wl.H, 16, 0, parts) if parts > 1 else rt.RowSet(0, 1, 1, wl.H)
    buf, stream = torch.zeros((wl.H, wl.W, 3), dtype=torch.uint8, device=dev), torch.cuda.Stream(device=dev)

    def frames(n):
        for _ in range(n):
            ctx.render_rows_device(wl.W, wl.H, rs, wl.max_depth, d_rgb8=buf.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()

    frames(5)
    t0 = time.perf_counter()
    frames(30)
    print("part 0 of %d: %.3f ms per frame" % (parts, (time.perf_counter() - t0) / 30 * 1e3))
