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
