"""profiles/concurrent_parts.py K [frames] [W H]: the whole frame as K band-interleaved parts rendered CONCURRENTLY on K
streams of one GPU (one context per part here; every part is its own chain of launches), against the same K parts one
after the other and the undivided frame.  Question: do the tails of one chain's traversal launches fill with the other
chains' rays?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
name = os.environ.get("RT_WORKLOAD", "sponza")
wl = workloads.by_name(name, W=W, H=H)
band = int(os.environ.get("RT_BAND", "16"))


def make(parts):
    out = []
    for k in range(parts):
        ctx = rt.Context(0)
        wl.load(ctx)
        ctx.set_option(rt.OPT_TIMING, 0)
        rs = rt.band_rowset(H, band, k, parts) if parts > 1 else rt.RowSet(0, 1, 1, H)
        buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        out.append((ctx, rs, buf, torch.cuda.Stream()))
    return out


def run(chains, concurrent):
    for it in range(frames + 3):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for ctx, rs, buf, s in chains:
            ctx.render_rows_device(W, H, rs, wl.max_depth, d_rgb8=buf.data_ptr(), stream=s.cuda_stream)
            if not concurrent:
                s.synchronize()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


whole = make(1)
print("%s %dx%d  undivided frame: %.3f ms" % (name, W, H, run(whole, True)), flush=True)
parts = make(K)
print("%d parts one after the other: %.3f ms" % (K, run(parts, False)), flush=True)
print("%d parts concurrently:        %.3f ms" % (K, run(parts, True)), flush=True)
print("undivided again: %.3f ms" % run(whole, True))
