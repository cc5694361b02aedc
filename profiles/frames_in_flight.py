"""profiles/frames_in_flight.py PARTS F [frames]: one rank's share (part 0 of PARTS) of the 1080p frame with F frames in flight
(F contexts on F streams, frame i goes to context i % F), ms per frame."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads

parts, F = int(sys.argv[1]), int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 60
W, H = 1920, 1080
wl = workloads.sponza(W=W, H=H)
rs = rt.band_rowset(H, 16, 0, parts) if parts > 1 else rt.RowSet(0, 1, 1, H)
chains = []
dummy = int(os.environ.get("RT_FIF_DUMMY", "0"))        # streams created (and kept) before every chain's stream: shifts which hardware queue it gets
keep = []
for k in range(F):
    keep += [torch.cuda.Stream() for _ in range(dummy)]
    ctx = rt.Context(0)
    wl.load(ctx)
    ctx.set_option(rt.OPT_TIMING, 0)
    if os.environ.get("RT_NO_GRAPH"):
        ctx.set_option(rt.OPT_GRAPH, 0)
    prio = [int(x) for x in os.environ.get("RT_FIF_PRIO", "0").split(",")]
    chains.append((ctx, torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda"), torch.cuda.Stream(priority=prio[k % len(prio)])))
for it in range(frames + 2 * F):
    if it == 2 * F:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx, buf, s = chains[it % F]
    ctx.render_rows_device(W, H, rs, wl.max_depth, d_rgb8=buf.data_ptr(), stream=s.cuda_stream)
torch.cuda.synchronize()
print("part 0 of %d, %d frames in flight: %.3f ms per frame" % (parts, F, (time.perf_counter() - t0) / frames * 1e3))
