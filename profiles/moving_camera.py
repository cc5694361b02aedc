"""profiles/moving_camera.py [step ...]: the bench workload with the camera moving every frame (no graph replay: every frame differs; the
claim order made for the first view goes stale as the camera moves on), ms per frame with the claim order on and off, and
with a new order made every 50 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("RT_AMD_LIB"):
    pass        # (raytracer_amd loads the library this names: A/B of builds)
import raytracer_amd as rt
from raytracer_amd import workloads

wl = workloads.sponza(W=1920, H=1080)
dev = torch.device("cuda", 0)
buf = torch.zeros((wl.H, wl.W, 3), dtype=torch.uint8, device=dev)
rs = rt.RowSet(0, 1, 1, wl.H)
cam0 = wl.camera


def run(order, renew_every=0, frames=200, step=0.01):
    with rt.Context(0) as ctx:
        wl.load(ctx)
        ctx.set_option(rt.OPT_CLAIM_ORDER, order)
        stream = torch.cuda.Stream(device=dev)
        for k in range(4):                                   # the first view, until its order is made
            ctx.render_rows_device(wl.W, wl.H, rs, wl.max_depth, d_rgb8=buf.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            o = (cam0.origin[0] + step * k, cam0.origin[1], cam0.origin[2] + 0.5 * step * k)
            ctx.set_camera(rt.Camera(origin=o, pitch=cam0.pitch, off_x=cam0.off_x + step * k, off_y=cam0.off_y, plane_z=cam0.plane_z + 0.5 * step * k))
            if renew_every and k and k % renew_every == 0:
                ctx.set_option(rt.OPT_CLAIM_ORDER, 2)
                ctx.render_rows_device(wl.W, wl.H, rs, wl.max_depth, d_rgb8=buf.data_ptr(), stream=stream.cuda_stream)   # (same view twice: the order is made)
            ctx.render_rows_device(wl.W, wl.H, rs, wl.max_depth, d_rgb8=buf.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3


steps = [float(x) for x in sys.argv[1:]] or [0.002, 0.01, 0.05]
for step in steps:
    print("camera step %g per frame over 200 frames: order off %.3f ms, order of the first view %.3f ms, new order every 50 frames %.3f ms (its making included)" % (
        step, run(0, step=step), run(1, step=step), run(1, 50, step=step)), flush=True)
