"""Full-size configurations: the 1920x1080 Sponza-class frame and the 4M-triangle soup.

Where the CPU oracle is still affordable (its own BVH, 16 threads) the whole frame is
compared; beyond that the checks are size-independent properties of the trace path:
re-rendering is bit-identical, any row partition reassembles to the same frame, the
primary hit plane does not depend on the bounce limit, ray counts obey the reference's
casting rules (main.cpp:170-188, 208-211)."""
import numpy as np
import pytest

from common import O, compare, rt
from raytracer_amd import workloads
from test_gpu_parity import assert_parity, orays, rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sponza():
    wl = workloads.sponza()
    ctx = rt.Context(0)
    wl.load(ctx)
    yield wl, ctx
    ctx.close()


def test_sponza_1080p_equals_cpu_oracle(sponza):
    wl, ctx = sponza
    g = ctx.render_frame(wl.W, wl.H, max_depth=wl.max_depth)
    st = ctx.stats()
    sc = O.scene_from_workload(wl, bvh=True)
    r = sc.render(wl.W, wl.H, max_depth=wl.max_depth, cam=O.camera_from(wl.camera), use_bvh=True, threads=16)
    c = compare(g, r)
    assert_parity(c)
    assert rays(st) == orays(r)
    assert st["accel"] == 1 and st["n_triangles"] == wl.n_tris


def test_sponza_1080p_properties(sponza):
    wl, ctx = sponza
    W, H, D = wl.W, wl.H, wl.max_depth
    a = ctx.render_frame(W, H, max_depth=D)
    sa = ctx.stats()
    b = ctx.render_frame(W, H, max_depth=D)
    for k in ("rgb8", "hit_id", "rgb32"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k          # idempotent
    # any partition gives the same pixels
    cuts = [0, 1, 130, 137, 600, 1079, H]
    parts = [ctx.render_frame(W, H, y0, y1, max_depth=D) for y0, y1 in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate([p["rgb8"] for p in parts]), a["rgb8"])
    assert np.array_equal(np.concatenate([p["hit_id"] for p in parts]), a["hit_id"])
    # casting rules: with no bounces every non-emissive hit casts one shadow ray per light
    z = ctx.render_frame(W, H, max_depth=0)
    sz = ctx.stats()
    assert np.array_equal(z["hit_id"], a["hit_id"])                                 # primary hits ignore the bounce limit
    lights = [i for i, p in enumerate(wl.analytic) if p.is_light]
    lit = (z["hit_id"] >= 0) & ~np.isin(z["hit_id"], lights)
    assert sz["rays_primary"] == W * H and sz["rays_reflect"] == 0
    assert sz["rays_shadow"] == len(lights) * int(lit.sum())
    # more bounces only ever add rays, and level-1 reflections = reflective primary hits
    one = ctx.render_frame(W, H, max_depth=1, want=("hit_id",))
    s1 = ctx.stats()
    assert s1["rays_reflect"] == int(lit.sum())          # every surface in this workload has reflect > 0
    assert sz["rays_shadow"] <= s1["rays_shadow"] <= sa["rays_shadow"] and s1["rays_reflect"] <= sa["rays_reflect"]


def test_soup_4m_triangles():
    wl = workloads.soup(W=1280, H=720, max_depth=8)
    assert wl.n_tris > 4_000_000
    with rt.Context(0) as ctx:
        wl.load(ctx)
        a = ctx.render_frame(wl.W, wl.H, max_depth=wl.max_depth)
        st = ctx.stats()
        b = ctx.render_frame(wl.W, wl.H, max_depth=wl.max_depth, want=("rgb8",))
    assert st["accel"] == 1 and st["bvh_depth"] <= 64 and st["n_triangles"] == wl.n_tris
    assert np.array_equal(a["rgb8"], b["rgb8"])
    sc = O.scene_from_workload(wl, bvh=True)
    lat = 8
    r = sc.render(wl.W, wl.H, max_depth=wl.max_depth, cam=O.camera_from(wl.camera), use_bvh=True, threads=16, x_step=lat,
                  y_step=lat)
    sub = {k: a[k][::lat, ::lat] for k in ("rgb8", "rgb32", "hit_id")}
    rsub = {k: r[k][::lat, ::lat] for k in ("rgb8", "rgb32", "hit_id")}
    assert_parity(compare(sub, rsub))


def test_sponza_4k_in_eight_band_parts(sponza):
    """The tile-partitioned 4K configuration on one GPU: the 3840x2160 frame rendered as the eight
    16-row band sets the eight ranks would take, reassembled on the device, equals the frame rendered
    whole; a pixel lattice of it equals the CPU oracle; the parts cast exactly the frame's rays."""
    import torch
    wl0, ctx = sponza
    wl = workloads.sponza(W=3840, H=2160)
    W, H, D, parts, band = wl.W, wl.H, wl.max_depth, 8, 16
    wl.load(ctx)                                    # same scene, the 4K camera
    try:
        whole = ctx.render_frame(W, H, max_depth=D, want=("rgb8", "hit_id"))
        whole_rays = rays(ctx.stats())
        cap = rt.rows_per_part(H, band, parts)
        g8 = torch.zeros((parts, cap, W, 3), dtype=torch.uint8, device="cuda")
        gid = torch.zeros((parts, cap, W), dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        total = [0, 0, 0]
        for p in range(parts):
            ctx.render_rows_device(W, H, rt.band_rowset(H, band, p, parts), D, g8[p].data_ptr(), gid[p].data_ptr(), None, None, stream)
            torch.cuda.synchronize()
            total = [a + b for a, b in zip(total, rays(ctx.stats()))]
        out8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        outid = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        ctx.deinterleave_bands(g8.data_ptr(), out8.data_ptr(), W, H, band, parts, cap, 3, stream)
        ctx.deinterleave_bands(gid.data_ptr(), outid.data_ptr(), W, H, band, parts, cap, 4, stream)
        torch.cuda.synchronize()
        assert np.array_equal(out8.cpu().numpy(), whole["rgb8"])
        assert np.array_equal(outid.cpu().numpy(), whole["hit_id"])
        assert tuple(total) == whole_rays
        sc = O.scene_from_workload(wl, bvh=True)
        lat = 6
        r = sc.render(W, H, max_depth=D, cam=O.camera_from(wl.camera), use_bvh=True, threads=16, x_step=lat, y_step=lat)
        assert np.array_equal(whole["hit_id"][::lat, ::lat], r["hit_id"][::lat, ::lat])
        d = np.abs(whole["rgb8"][::lat, ::lat].astype(int) - r["rgb8"][::lat, ::lat].astype(int)).max(-1)
        assert (d > 1).sum() == 0 and (d == 0).mean() >= 0.999
    finally:
        wl0.load(ctx)                               # leave the module's scene as the other tests expect it
