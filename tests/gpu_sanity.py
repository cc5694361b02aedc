"""Quick GPU-vs-oracle sweep printed as a table (run by hand: python tests/gpu_sanity.py).
Not collected by pytest; the real parity tests are tests/test_gpu_*.py."""
import json
import sys
import time

import numpy as np

from common import O, compare, ocam, rt, to_oracle, to_product, stock_spec, cube_spec, atrium_spec


def run(name, spec, W, H, depth, accel=rt.ACCEL_AUTO, oracle_bvh=False, threads=8, leaf_max=4):
    cam = rt.camera_for_width(W)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, accel)
        ctx.set_option(rt.OPT_LEAF_MAX, leaf_max)
        ctx.set_option(rt.OPT_COUNT_TESTS, 1)
        to_product(ctx, spec)
        ctx.set_camera(cam)
        t0 = time.time()
        g = ctx.render_frame(W, H, max_depth=depth)
        t_gpu = time.time() - t0
        st = ctx.stats()
    sc = to_oracle(spec, bvh=oracle_bvh)
    t0 = time.time()
    r = sc.render(W, H, max_depth=depth, cam=ocam(cam), use_bvh=oracle_bvh, threads=threads)
    t_cpu = time.time() - t0
    c = compare(g, r)
    rays_ok = (st["rays_primary"], st["rays_shadow"], st["rays_reflect"]) == (
        r["counts"]["primary"], r["counts"]["shadow"], r["counts"]["reflect"])
    print(json.dumps(dict(name=name, W=W, H=H, depth=depth, tris=spec.n_tris, accel=st["accel"], bvh_depth=st["bvh_depth"],
                          **c, rays_equal=rays_ok, rays=[st["rays_primary"], st["rays_shadow"], st["rays_reflect"]],
                          ms_frame=round(st["ms_frame"], 3), box=st["box_tests"], tri=st["tri_tests"],
                          t_gpu_call=round(t_gpu, 3), t_cpu=round(t_cpu, 3))), flush=True)
    return c


if __name__ == "__main__":
    which = sys.argv[1:] or ["stock", "cube", "cube_refl", "cube_bvh", "atrium_small"]
    if "stock" in which:
        run("stock", stock_spec(), 800, 600, 3)
    if "cube" in which:
        run("cube", cube_spec(), 512, 512, 1)
    if "cube_refl" in which:
        run("cube_refl", cube_spec(mesh_reflect=0.5), 640, 360, 4)
    if "cube_bvh" in which:
        run("cube_bvh", cube_spec(mesh_reflect=0.5), 640, 360, 4, accel=rt.ACCEL_LBVH, leaf_max=1)
        run("cube_bvh4", cube_spec(mesh_reflect=0.5), 640, 360, 4, accel=rt.ACCEL_LBVH, leaf_max=4)
    if "atrium_small" in which:
        run("atrium_small", atrium_spec(detail=0.02), 480, 270, 4)
    if "atrium_mid" in which:
        run("atrium_mid", atrium_spec(detail=0.2), 480, 270, 4, oracle_bvh=True)
    if "atrium" in which:
        run("atrium", atrium_spec(), 960, 540, 4, oracle_bvh=True)
