"""Parity of the HIP trace path (through the C ABI) against the reference's outputs.

Bars, from the north star: the nearest-hit primitive id equals the reference's on >= 99.9 % of
pixels and RGB differs by at most 1/255 on >= 99.9 % of pixels.  What is asserted here is
stricter wherever the arithmetic allows it: ids and ray counts must be IDENTICAL, RGB8 may
differ by at most 1/255 anywhere, and float RGB may differ only through powf (the device
evaluates pow(x,20) by exact-ish double multiplications, glibc by its own algorithm; both
are within 1 ulp of the true value).
"""
import hashlib
import os
import subprocess
import tempfile

import numpy as np
import pytest

from common import (GOLDEN, O, ROOT, SceneSpec, atrium_spec, compare, cube_spec, golden_cases, lcg_floats, ocam, rt, scenes, soup_spec,
                    stock_analytic, stock_spec, to_oracle, to_product)

pytestmark = pytest.mark.gpu

RGB32_TOL = 4e-6          # absolute, on colours of order 1: a few float ulps from powf
MIN_FRACTION = 0.999      # the north star's pixel fraction


def assert_parity(c, exact_ids=True):
    n = c["pixels"]
    if exact_ids:
        assert c["id_mismatch"] == 0, c
    assert c["id_mismatch"] <= (1 - MIN_FRACTION) * n, c
    assert c["rgb8_over_tol"] == 0, c
    assert c["rgb8_exact"] >= MIN_FRACTION * n, c
    assert c["rgb32_max_abs"] <= RGB32_TOL, c


def render(spec, W, H, depth, cam, accel=rt.ACCEL_AUTO, leaf_max=4, flat=False):
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, accel)
        ctx.set_option(rt.OPT_LEAF_MAX, leaf_max)
        to_product(ctx, spec, flat=flat)
        ctx.set_camera(cam)
        g = ctx.render_frame(W, H, max_depth=depth)
        g["stats"] = ctx.stats()
    return g


def rays(st):
    return (st["rays_primary"], st["rays_shadow"], st["rays_reflect"])


def orays(r):
    return (r["counts"]["primary"], r["counts"]["shadow"], r["counts"]["reflect"])


# ------------------------------------------------------------------ against the reference's own outputs
def test_stock_frame_against_the_unmodified_reference(golden_stock):
    g = render(stock_spec(), 800, 600, 3, rt.Camera())
    want = golden_stock["rgb8"]
    d = np.abs(g["rgb8"].astype(int) - want.astype(int)).max(-1)
    assert (d > 1).sum() == 0
    assert (d == 0).sum() >= MIN_FRACTION * d.size
    assert rays(g["stats"]) == (480000, 1112248, 236124)


@pytest.mark.parametrize("accel,leaf_max,flat", [(rt.ACCEL_LINEAR, 4, False), (rt.ACCEL_LBVH, 1, False),
                                                  (rt.ACCEL_LBVH, 4, True), (rt.ACCEL_LBVH, 8, False)])
@pytest.mark.parametrize("name", sorted(golden_cases().keys()))
def test_golden_scenes(name, accel, leaf_max, flat, golden_scenes):
    spec, W, H, bounces, cam = golden_cases()[name]
    if spec.n_tris < 2 and accel == rt.ACCEL_LBVH and leaf_max != 1:
        pytest.skip("no mesh to build over")
    g = render(spec, W, H, bounces, cam, accel, leaf_max, flat)
    ref = {k: golden_scenes[name + "/" + k] for k in ("rgb8", "rgb32", "hit_id")}
    assert_parity(compare(g, ref))
    assert g["stats"]["accel"] == (1 if (accel == rt.ACCEL_LBVH and spec.n_tris >= 2) else 0)


# ------------------------------------------------------------------ against the oracle, the benchmark configurations
def test_config_cube_512_one_bounce():
    spec = cube_spec(mirror=False, mesh_reflect=0.5)
    cam = rt.camera_for_width(512)
    g = render(spec, 512, 512, 1, cam)
    r = to_oracle(spec).render(512, 512, max_depth=1, cam=ocam(cam), threads=8)
    assert_parity(compare(g, r))
    assert rays(g["stats"]) == orays(r)


def test_config_cube_1080p_four_bounces():
    spec = cube_spec(mirror=False, mesh_reflect=0.5)
    cam = rt.camera_for_width(1920)
    r = to_oracle(spec).render(1920, 1080, max_depth=4, cam=ocam(cam), threads=16)
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        g = render(spec, 1920, 1080, 4, cam, accel)
        assert_parity(compare(g, r))
        assert rays(g["stats"]) == orays(r)


def test_cube_inside_the_mirror_sphere_is_invisible():
    cam = rt.Camera()
    a = render(stock_spec(), 800, 600, 3, cam)
    b = render(cube_spec(mirror=True), 800, 600, 3, cam)
    assert np.array_equal(a["rgb8"], b["rgb8"]) and np.array_equal(a["hit_id"], b["hit_id"])


def test_atrium_small_against_linear_scan():
    """LBVH over ~6k triangles vs the oracle's brute-force scan (the reference's algorithm)."""
    spec = atrium_spec(detail=0.02)
    cam = rt.camera_for_width(240)
    r = to_oracle(spec).render(240, 135, max_depth=4, cam=ocam(cam), threads=16)
    for leaf_max in (1, 4):
        g = render(spec, 240, 135, 4, cam, rt.ACCEL_LBVH, leaf_max)
        assert_parity(compare(g, r))
        assert rays(g["stats"]) == orays(r)
        assert g["stats"]["bvh_depth"] <= 64


def test_sponza_standin_against_cpu_bvh_and_linear_scan_lattice():
    """The 263k-triangle workload at quarter resolution: full frame vs the oracle with its own
    (independent, median-split) BVH, plus a sparse lattice vs the oracle's brute-force scan."""
    from raytracer_amd import workloads
    wl = workloads.sponza(W=960, H=540)
    with rt.Context(0) as ctx:
        wl.load(ctx)
        g = ctx.render_frame(wl.W, wl.H, max_depth=wl.max_depth)
        st = ctx.stats()
    assert st["accel"] == 1
    sc = O.scene_from_workload(wl, bvh=True)
    cam = O.camera_from(wl.camera)
    r = sc.render(wl.W, wl.H, max_depth=wl.max_depth, cam=cam, use_bvh=True, threads=16)
    assert_parity(compare(g, r))
    assert rays(st) == orays(r)
    lat = 48
    b = sc.render(wl.W, wl.H, max_depth=wl.max_depth, cam=cam, use_bvh=False, threads=16, x_step=lat, y_step=lat)
    sub = {k: g[k][::lat, ::lat] for k in ("rgb8", "rgb32", "hit_id")}
    bsub = {k: b[k][::lat, ::lat] for k in ("rgb8", "rgb32", "hit_id")}
    assert_parity(compare(sub, bsub))


# ------------------------------------------------------------------ rows, tiles, bands
def test_row_ranges_equal_slices_of_the_full_frame():
    spec, W, H, bounces, cam = golden_cases()["soup_d3_160"]
    with rt.Context(0) as ctx:
        to_product(ctx, spec)
        ctx.set_camera(cam)
        full = ctx.render_frame(W, H, max_depth=bounces)
        for y0, y1 in [(0, 1), (5, 6), (17, 90), (H - 3, H), (40, 40)]:
            part = ctx.render_frame(W, H, y0, y1, max_depth=bounces)
            for k in ("rgb8", "hit_id", "rgb32"):
                assert np.array_equal(part[k].view(np.uint8), full[k][y0:y1].view(np.uint8)), (k, y0, y1)


@pytest.mark.parametrize("W,H", [(1, 1), (7, 3), (9, 17), (64, 1), (65, 9)])
def test_odd_frame_sizes(W, H):
    spec = cube_spec(mirror=False, mesh_reflect=0.5)
    cam = rt.Camera(pitch=8.0 / max(W, 8))
    g = render(spec, W, H, 2, cam, rt.ACCEL_LBVH)
    r = to_oracle(spec).render(W, H, max_depth=2, cam=ocam(cam))
    assert_parity(compare(g, r))


def test_band_partition_on_device_and_deinterleave():
    import torch
    from raytracer_amd.dist import deinterleave_numpy, local_rows
    spec, W, H, bounces, cam = golden_cases()["cube_refl_d4_200"]
    parts, band = 3, 16
    cap = rt.rows_per_part(H, band, parts)
    with rt.Context(0) as ctx:
        to_product(ctx, spec)
        ctx.set_camera(cam)
        full = ctx.render_frame(W, H, max_depth=bounces)
        gathered8 = torch.zeros((parts, cap, W, 3), dtype=torch.uint8, device="cuda")
        gathered_id = torch.zeros((parts, cap, W), dtype=torch.int32, device="cuda")
        gathered_f = torch.zeros((parts, cap, W, 3), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for p in range(parts):
            rs = rt.band_rowset(H, band, p, parts)
            ctx.render_rows_device(W, H, rs, bounces, gathered8[p].data_ptr(), gathered_id[p].data_ptr(),
                                   gathered_f[p].data_ptr(), None, stream)
        out8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        outid = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        outf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        ctx.deinterleave_bands(gathered8.data_ptr(), out8.data_ptr(), W, H, band, parts, cap, 3, stream)
        ctx.deinterleave_bands(gathered_id.data_ptr(), outid.data_ptr(), W, H, band, parts, cap, 4, stream)
        ctx.deinterleave_bands(gathered_f.data_ptr(), outf.data_ptr(), W, H, band, parts, cap, 12, stream)
        torch.cuda.synchronize()
        assert np.array_equal(out8.cpu().numpy(), full["rgb8"])
        assert np.array_equal(outid.cpu().numpy(), full["hit_id"])
        assert np.array_equal(outf.cpu().numpy().view(np.uint32), full["rgb32"].view(np.uint32))
        assert np.array_equal(deinterleave_numpy(gathered8.cpu().numpy(), H, band), full["rgb8"])
        for p in range(parts):
            ys = local_rows(H, band, p, parts)
            assert np.array_equal(gathered_id[p, : len(ys)].cpu().numpy(), full["hit_id"][ys])


def test_band_partition_of_an_lbvh_scene_matches_the_whole_frame():
    """The staged LBVH pipeline on every rank's band set (what bench.py runs on N GPUs), reassembled,
    equals the single-GPU frame bit for bit; odd height so that the last band is partial."""
    import torch
    spec = atrium_spec(detail=0.02)
    W, H, bounces, parts, band = 200, 150, 3, 4, 16
    cam = rt.camera_for_width(W)
    cap = rt.rows_per_part(H, band, parts)
    with rt.Context(0) as ctx:
        to_product(ctx, spec)
        ctx.set_camera(cam)
        full = ctx.render_frame(W, H, max_depth=bounces)
        full_rays = rays(ctx.stats())
        assert ctx.stats()["accel"] == 1
        g8 = torch.zeros((parts, cap, W, 3), dtype=torch.uint8, device="cuda")
        gid = torch.full((parts, cap, W), -7, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        total = [0, 0, 0]
        for p in range(parts):
            rs = rt.band_rowset(H, band, p, parts)
            ctx.render_rows_device(W, H, rs, bounces, g8[p].data_ptr(), gid[p].data_ptr(), None, None, stream)
            torch.cuda.synchronize()
            total = [a + b for a, b in zip(total, rays(ctx.stats()))]
        out8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        outid = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        ctx.deinterleave_bands(g8.data_ptr(), out8.data_ptr(), W, H, band, parts, cap, 3, stream)
        ctx.deinterleave_bands(gid.data_ptr(), outid.data_ptr(), W, H, band, parts, cap, 4, stream)
        torch.cuda.synchronize()
    assert np.array_equal(out8.cpu().numpy(), full["rgb8"])
    assert np.array_equal(outid.cpu().numpy(), full["hit_id"])
    assert tuple(total) == full_rays                       # the parts cast exactly the frame's rays


def test_graph_replay_of_an_unchanged_frame_and_recapture_after_a_change():
    """With timing off an unchanged frame replays a captured hipGraph: same bytes as the direct launches,
    also after the camera, the depth and the scene change (each is launched directly once, captured on its repeat)."""
    import torch
    spec = atrium_spec(detail=0.02)
    W, H = 160, 120
    rs = rt.RowSet(0, 1, 1, H)
    stream = torch.cuda.current_stream().cuda_stream

    def frame(ctx, depth):
        out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        ids = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        res = []
        for _ in range(4):                                   # direct launches, then the capture, then two replays
            out.zero_(); ids.zero_()
            ctx.render_rows_device(W, H, rs, depth, out.data_ptr(), ids.data_ptr(), None, None, stream)
            torch.cuda.synchronize()
            res.append((out.cpu().numpy().copy(), ids.cpu().numpy().copy(), rays(ctx.stats())))
        for r in res[1:]:
            assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and r[2] == res[0][2]
        return res[0]

    results = {}
    for graph in (0, 1):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_TIMING, 0)
            ctx.set_option(rt.OPT_GRAPH, graph)
            to_product(ctx, spec)
            ctx.set_camera(rt.camera_for_width(W))
            a = frame(ctx, 3)
            ctx.set_camera(rt.Camera(origin=(0.5, 0.0, -5.0), pitch=8.0 / W, off_x=-3.5))
            b = frame(ctx, 3)
            c = frame(ctx, 1)
            to_product(ctx, cube_spec(mirror=False, mesh_reflect=0.5))   # another scene, linear-scan path
            ctx.set_camera(rt.camera_for_width(W))
            d = frame(ctx, 2)
            results[graph] = (a, b, c, d)
    for x, y in zip(results[0], results[1]):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2]
    assert not np.array_equal(results[1][0][0], results[1][1][0])        # the camera move is visible


def test_frame_planes_t_and_normal_of_the_primary_hit():
    """rt_render_frame_planes on a linear-scan scene and on an LBVH scene: colour and ids as rt_render_frame,
    t bit-equal to the oracle's, normals = the Triangle constructor's N (main.cpp:104-107) / the plane's n."""
    for spec, W, H in ((cube_spec(mirror=False, mesh_reflect=0.5), 200, 150), (atrium_spec(detail=0.02, mirror=False), 160, 120)):
        cam = rt.camera_for_width(W)
        with rt.Context(0) as ctx:
            to_product(ctx, spec)
            ctx.set_camera(cam)
            ref = ctx.render_frame(W, H, max_depth=2)
            g = ctx.render_frame_planes(W, H, max_depth=2)
        for k in ("rgb8", "hit_id", "rgb32"):
            assert np.array_equal(g[k].view(np.uint8), ref[k].view(np.uint8)), k
        r = to_oracle(spec).render(W, H, max_depth=2, cam=ocam(cam), threads=8)
        hit = g["hit_id"] >= 0
        assert np.array_equal(g["hit_t"][hit].view(np.uint32), r["hit_t"][hit].view(np.uint32))
        assert (g["normal"][~hit] == 0).all()
        n_an = len(spec.analytic)
        tri = g["hit_id"] >= n_an
        assert tri.sum() > 500
        v = spec.verts[spec.indices.reshape(-1, 3)]                       # (T, 3, 3)
        e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
        n = np.cross(e1.astype(np.float64), e2.astype(np.float64))
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        want = n[g["hit_id"][tri] - n_an]
        assert np.abs(g["normal"][tri] - want).max() < 2e-6
        for i, a in enumerate(spec.analytic):
            if a["kind"] == "plane":
                m = g["hit_id"] == i
                if m.any():
                    assert np.array_equal(g["normal"][m], np.broadcast_to(np.asarray(a["p"], np.float32), (m.sum(), 3)))
        ln = np.linalg.norm(g["normal"][hit].astype(np.float64), axis=1)
        assert np.abs(ln - 1.0).max() < 1e-5


@pytest.mark.parametrize("n_tris", [2, 3, 4, 5, 7, 9, 17])
def test_tiny_meshes_through_every_hierarchy_shape(n_tris):
    """Two to seventeen triangles forced through the hierarchy pipelines (wide nodes with one to four
    children, a root whose children are leaves, every leaf size, both builders, all three pipelines):
    each equals the linear scan bit for bit, and the linear scan equals the oracle."""
    spec = soup_spec(n_tris, seed=100 + n_tris)
    cam = rt.Camera(pitch=0.05)
    W, H, D = 160, 120, 3
    base = render(spec, W, H, D, cam, rt.ACCEL_LINEAR)
    assert_parity(compare(base, to_oracle(spec).render(W, H, max_depth=D, cam=ocam(cam))))
    for builder in (rt.BUILDER_LBVH, rt.BUILDER_PLOC, rt.BUILDER_SAH):
        for leaf_max in (1, 2, 8):
            for pipeline in (rt.PIPELINE_FUSED, rt.PIPELINE_WAVEFRONT):
                with rt.Context(0) as ctx:
                    ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
                    ctx.set_option(rt.OPT_BUILDER, builder)
                    ctx.set_option(rt.OPT_LEAF_MAX, leaf_max)
                    ctx.set_option(rt.OPT_PIPELINE, pipeline)
                    to_product(ctx, spec)
                    ctx.set_camera(cam)
                    g = ctx.render_frame(W, H, max_depth=D)
                    st = ctx.stats()
                assert st["accel"] == 1
                for k in ("rgb8", "hit_id", "rgb32"):
                    assert np.array_equal(g[k].view(np.uint8), base[k].view(np.uint8)), (k, builder, leaf_max, pipeline)
                assert rays(st) == rays(base["stats"])


def _sah_case(kind):
    """Meshes that send the surface-area-heuristic builder down each of its paths."""
    if kind.startswith("soup"):
        return soup_spec(int(kind[4:]), seed=int(kind[4:]))
    base = soup_spec(1, seed=5)
    tri = base.verts[:3]
    if kind == "stack":            # 300 copies of one triangle: every centroid equal, no plane separates anything
        verts = np.tile(tri, (300, 1))
    else:                          # "ladder": spacing grows by powers of two, so every cheapest cut peels off one end
        n = 90
        shift = np.zeros((n, 3), np.float32)
        shift[:, 0] = (2.0 ** (np.arange(n) * 0.25) - 1.0) * 0.01
        verts = (tri[None] * 0.05 + np.array([-3.5, 0.0, 2.0], np.float32) + shift[:, None, :]).reshape(-1, 3)
    n_tris = len(verts) // 3
    return SceneSpec(base.analytic, verts, np.arange(n_tris * 3, dtype=np.int32), mats=base.mats,
                     tri_material=(np.arange(n_tris) % 4).astype(np.int32))


@pytest.mark.parametrize("kind", ["soup63", "soup64", "soup65", "soup66", "soup129", "soup1000", "soup20000", "stack", "ladder"])
def test_sah_builder_on_every_path_equals_linear_scan(kind):
    """The top-down builder has two regimes (rounds over the whole grid above 64 references, one wave per segment
    below) and two kinds of cut (a plane, or the middle of the range when no plane separates the centroids or the tree
    grows too deep).  Sizes either side of the regime boundary, identical triangles and a mesh that peels one
    triangle per cut: each frame equals the linear scan bit for bit, with the same ray counts."""
    spec = _sah_case(kind)
    cam = rt.Camera(pitch=0.05)
    W, H, D = 160, 120, 3
    base = render(spec, W, H, D, cam, rt.ACCEL_LINEAR)
    for leaf_max in (1, 2):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
            ctx.set_option(rt.OPT_BUILDER, rt.BUILDER_SAH)
            ctx.set_option(rt.OPT_LEAF_MAX, leaf_max)
            ctx.set_option(rt.OPT_PIPELINE, rt.PIPELINE_WAVEFRONT)
            to_product(ctx, spec)
            ctx.set_camera(cam)
            g = ctx.render_frame(W, H, max_depth=D)
            st = ctx.stats()
        assert st["accel"] == 1 and 0 < st["bvh_depth"] <= 64 and st["bvh_sah_cost"] > 0
        for k in ("rgb8", "hit_id", "rgb32"):
            assert np.array_equal(g[k].view(np.uint8), base[k].view(np.uint8)), (k, kind, leaf_max)
        assert rays(st) == rays(base["stats"])


@pytest.mark.parametrize("builder", [rt.BUILDER_SAH, rt.BUILDER_PLOC, rt.BUILDER_LBVH])
def test_triangles_with_nan_or_infinite_vertices_never_hit_and_never_break_the_build(builder):
    """A NaN or an infinity in a vertex makes the reference's Moller-Trumbore test fail for every ray (t comes out NaN or
    0, main.cpp:127-128); such triangles must not poison the scene bounds either.  3000 triangles, five of them broken
    (and one duplicated): every builder's frame equals the linear scan bit for bit, and no broken triangle is ever hit."""
    rng = np.random.default_rng(7)
    n = 3000
    c = rng.uniform(-4, 4, (n, 3)); c[:, 2] += 5
    tri = (c[:, None, :] + rng.uniform(-0.2, 0.2, (n, 3, 3))).astype(np.float32)
    broken = [5, 17, 100, 200, 2999]
    tri[5, 1, 0] = np.nan; tri[17, :, :] = np.nan; tri[100, 2, 1] = np.inf; tri[200, 0, 2] = -np.inf; tri[2999, 1, :] = np.inf
    tri[300] = tri[301]
    verts, idx = tri.reshape(-1, 3), np.arange(n * 3, dtype=np.int32)
    W, H = 96, 64
    outs = []
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_ACCEL, accel)
            ctx.set_option(rt.OPT_BUILDER, builder)
            ctx.set_analytic([rt.Sphere((2.0, 5.0, 1.0), 0.1, None, True)])
            ctx.set_mesh(verts, idx, rt.Material(reflect=0.3))
            ctx.set_camera(rt.Camera(pitch=8.0 / W))
            outs.append(ctx.render_frame(W, H, max_depth=2, want=("hit_id", "rgb32")))
            st = ctx.stats()
    assert st["accel"] == 1 and st["bvh_depth"] <= 64
    assert np.array_equal(outs[0]["hit_id"], outs[1]["hit_id"])
    assert np.array_equal(outs[0]["rgb32"].view(np.uint32), outs[1]["rgb32"].view(np.uint32))
    assert not np.isin(outs[0]["hit_id"], [1 + b for b in broken]).any()       # ids: the light sphere is 0, triangle k is 1 + k
    assert (outs[0]["hit_id"] > 0).sum() > 500


# ------------------------------------------------------------------ moving geometry (refit)
def _moved(verts, step):
    """A deformation that changes the scene bounds: a swell along x and a shear in y, growing with step."""
    v = verts.astype(np.float64).copy()
    v[:, 0] *= 1.0 + 0.15 * step
    v[:, 1] += 0.4 * step * np.sin(v[:, 2] * 1.7 + step)
    v[:, 2] += 0.25 * step * np.cos(v[:, 0] * 0.9)
    return v.astype(np.float32)


@pytest.mark.parametrize("builder", [rt.BUILDER_LBVH, rt.BUILDER_PLOC, rt.BUILDER_SAH])
@pytest.mark.parametrize("flat", [False, True])
def test_refit_after_vertex_updates_equals_a_fresh_build_and_the_oracle(builder, flat):
    """rt_update_mesh_vertices / rt_update_triangles keep the tree and refit its boxes: every frame
    after an update equals, bit for bit, the frame of a context that was given the moved vertices from
    the start, and that frame equals the oracle's.  Per-triangle materials survive the update."""
    spec = soup_spec(n_tris=600, seed=21)
    cam = rt.Camera(pitch=0.05)
    W, H, D = 160, 120, 3
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
        ctx.set_option(rt.OPT_BUILDER, builder)
        to_product(ctx, spec, flat=flat)
        ctx.set_camera(cam)
        ctx.render_frame(W, H, max_depth=D)
        nodes0 = ctx.stats()["bvh_nodes"]
        for step in (1, 2, 3):
            v = _moved(spec.verts, step)
            if flat:
                ctx.update_triangles(scenes.flatten(v, spec.indices))
            else:
                ctx.update_mesh_vertices(v)
            g = ctx.render_frame(W, H, max_depth=D)
            st = ctx.stats()
            assert st["accel"] == 1 and st["bvh_nodes"] == nodes0
            moved = SceneSpecCopy(spec, v)
            fresh = render(moved, W, H, D, cam, rt.ACCEL_LBVH, leaf_max=2, flat=flat)
            for k in ("rgb8", "hit_id", "rgb32"):
                assert np.array_equal(g[k].view(np.uint8), fresh[k].view(np.uint8)), (k, step)
            assert rays(st) == rays(fresh["stats"])
            if step == 3:
                r = to_oracle(moved).render(W, H, max_depth=D, cam=ocam(cam))
                assert_parity(compare(g, r))
                assert rays(st) == orays(r)


def SceneSpecCopy(spec, verts):
    from common import SceneSpec
    return SceneSpec(spec.analytic, verts, spec.indices, spec.mesh_mat, spec.mats, spec.tri_material)


def test_vertex_update_of_a_linear_scan_scene_and_count_mismatches():
    """Below the hierarchy threshold the update only reruns the Triangle constructor; wrong counts and
    an update without a mesh are refused with RT_E_STATE and leave the scene as it was."""
    spec = soup_spec(n_tris=12, seed=5)
    cam = rt.Camera(pitch=0.05)
    W, H, D = 96, 64, 2
    with rt.Context(0) as ctx:
        ctx.set_analytic(rt.stock_scene())
        with pytest.raises(rt.RtError) as e:
            ctx.update_triangles(np.zeros((4, 9), np.float32))          # nothing loaded
        assert e.value.code == -4
        to_product(ctx, spec)
        ctx.set_camera(cam)
        before = ctx.render_frame(W, H, max_depth=D)
        assert ctx.stats()["accel"] == 0
        with pytest.raises(rt.RtError) as e:
            ctx.update_mesh_vertices(spec.verts[:-1])
        assert e.value.code == -4
        with pytest.raises(rt.RtError) as e:
            ctx.update_triangles(np.zeros((spec.n_tris + 1, 9), np.float32))
        assert e.value.code == -4
        again = ctx.render_frame(W, H, max_depth=D)
        assert np.array_equal(before["rgb8"], again["rgb8"])
        v = _moved(spec.verts, 2)
        ctx.update_mesh_vertices(v)
        g = ctx.render_frame(W, H, max_depth=D)
    r = to_oracle(SceneSpecCopy(spec, v)).render(W, H, max_depth=D, cam=ocam(cam))
    assert_parity(compare(g, r))


# ------------------------------------------------------------------ opt-in shading normals
def _smooth_scene():
    """A mirror-ish triangle sphere and a diffuse one with their vertex normals, and a few loose
    triangles: half of those with arbitrary corner normals, half with nine zeros (geometric normal)."""
    from common import SceneSpec
    v1, i1, n1 = scenes.uv_sphere((-1.6, -1.0, 1.0), 1.5, 9, 14)
    v2, i2, n2 = scenes.uv_sphere((1.8, -0.5, 2.0), 1.2, 7, 10)
    loose = soup_spec(n_tris=10, seed=77)
    v = np.concatenate([v1, v2, loose.verts])
    i = np.concatenate([i1, i2 + len(v1), loose.indices + len(v1) + len(v2)])
    vn = np.concatenate([scenes.corner_normals(n1, i1), scenes.corner_normals(n2, i2),
                         (lcg_floats(90, 5).reshape(10, 3, 3) - 0.5).astype(np.float32)])
    vn[-5:] = 0.0
    mats = [(0.6, 0.8, 1.0, (0.9, 0.9, 1.0)), (0.0, 1.0, 1.0, (1.0, 0.6, 0.3)), (0.3, 1.0, 0.5, (0.4, 0.9, 0.5))]
    tm = np.concatenate([np.zeros(len(i1) // 3), np.ones(len(i2) // 3), np.full(10, 2)]).astype(np.int32)
    return SceneSpec(stock_analytic(mirror=False), v, i, mats=mats, tri_material=tm), vn


@pytest.mark.parametrize("accel,pipeline", [(rt.ACCEL_LINEAR, 0), (rt.ACCEL_LBVH, rt.PIPELINE_FUSED),
                                            (rt.ACCEL_LBVH, rt.PIPELINE_WAVEFRONT)])
def test_shading_normals_match_the_oracle_rule_in_every_pipeline(accel, pipeline):
    """rt_set_triangle_normals: normals interpolated with Triangle::Raycast's own u, v.  The reference has
    no such feature (parity unpinned); the check is against the oracle's statement of the same rule, to
    the usual bar: ids and ray counts exact, colours within the tolerance.  Turning it off restores the
    reference image bit for bit; vertex updates keep the normals, a new mesh drops them."""
    spec, vn = _smooth_scene()
    cam = rt.Camera(pitch=0.05)
    W, H, D = 160, 120, 3
    osc = to_oracle(spec)
    r_flat = osc.render(W, H, max_depth=D, cam=ocam(cam))
    osc.set_corner_normals(vn)
    r_smooth = osc.render(W, H, max_depth=D, cam=ocam(cam))
    assert (r_flat["rgb8"] != r_smooth["rgb8"]).any(-1).mean() > 0.05
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, accel)
        ctx.set_option(rt.OPT_PIPELINE, pipeline)
        to_product(ctx, spec)
        ctx.set_camera(cam)
        flat = ctx.render_frame(W, H, max_depth=D)
        assert_parity(compare(flat, r_flat))
        ctx.set_triangle_normals(vn)
        g = ctx.render_frame(W, H, max_depth=D)
        assert_parity(compare(g, r_smooth))
        assert rays(ctx.stats()) == orays(r_smooth)
        ctx.update_mesh_vertices(spec.verts)                       # same positions: normals must survive
        g2 = ctx.render_frame(W, H, max_depth=D)
        assert np.array_equal(g2["rgb32"].view(np.uint32), g["rgb32"].view(np.uint32))
        with pytest.raises(rt.RtError) as e:
            ctx.set_triangle_normals(vn[:-1])
        assert e.value.code == -4
        ctx.set_triangle_normals(None)
        off = ctx.render_frame(W, H, max_depth=D)
        assert np.array_equal(off["rgb32"].view(np.uint32), flat["rgb32"].view(np.uint32))
        ctx.set_triangle_normals(vn)
        to_product(ctx, spec)                                      # a new mesh: back to geometric normals
        again = ctx.render_frame(W, H, max_depth=D)
        assert np.array_equal(again["rgb32"].view(np.uint32), flat["rgb32"].view(np.uint32))


def test_frame_planes_report_the_shading_normal_when_set():
    spec, vn = _smooth_scene()
    cam = rt.Camera(pitch=0.05)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
        to_product(ctx, spec)
        ctx.set_camera(cam)
        ctx.set_triangle_normals(vn)
        g = ctx.render_frame_planes(160, 120, max_depth=1)
    first = len(spec.analytic)
    ball = (g["hit_id"] >= first) & (g["hit_id"] < first + 2 * 14 + 7 * 14 * 2)    # the first sphere's triangles
    assert ball.sum() > 200
    n = g["normal"][ball].astype(np.float64)
    assert np.abs(np.linalg.norm(n, axis=1) - 1.0).max() < 1e-5
    # interpolated normals of a sphere mesh vary smoothly: many distinct values, unlike the per-face normal
    assert len(np.unique(np.round(n, 4), axis=0)) > 0.5 * ball.sum()


# ------------------------------------------------------------------ texture coordinates (the uv plane)
def _affine_uv(p):
    """A texture coordinate that is an affine function of position: barycentric interpolation reproduces it,
    so the uv plane of a hit must equal it at the hit point."""
    p = np.asarray(p, np.float64)
    return np.stack([0.11 * p[..., 0] - 0.05 * p[..., 1] + 0.5, 0.07 * p[..., 2] + 0.09 * p[..., 1] + 0.25], -1)


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
def test_uv_plane_interpolates_the_corner_texcoords_at_the_primary_hit(accel):
    """rt_set_triangle_texcoords + the uv plane of rt_render_frame_planes (the reference parses `vt` and drops
    it, so there is no reference output: the check is the interpolation's defining property)."""
    spec, _ = _smooth_scene()
    cam = rt.Camera(pitch=0.05)
    W, H = 160, 120
    first = len(spec.analytic)
    corner_uv = _affine_uv(spec.verts[spec.indices].reshape(-1, 3, 3)).astype(np.float32)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, accel)
        to_product(ctx, spec)
        ctx.set_camera(cam)
        none = ctx.render_frame_planes(W, H, max_depth=1)
        assert not none["uv"].any()                                        # nothing loaded: zeros
        ctx.set_triangle_texcoords(corner_uv)
        g = ctx.render_frame_planes(W, H, max_depth=1)
        with pytest.raises(rt.RtError):
            ctx.set_triangle_texcoords(corner_uv[:-1])
        for k in ("rgb8", "hit_id", "hit_t", "normal"):                     # the image does not depend on them
            assert np.array_equal(g[k].view(np.uint8), none[k].view(np.uint8)), k
        ctx.set_triangle_texcoords(None)
        assert not ctx.render_frame_planes(W, H, max_depth=1)["uv"].any()
    tri = g["hit_id"] >= first
    assert tri.sum() > 2000 and not g["uv"][~tri].any()                     # misses, spheres and the plane: (0, 0)
    ys, xs = np.nonzero(tri)
    o = np.asarray(cam.origin, np.float64)
    d = np.stack([xs * np.float64(np.float32(cam.pitch)) + cam.off_x, ys * np.float64(np.float32(cam.pitch)) + cam.off_y,
                  np.full(len(xs), cam.plane_z)], -1) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hit = o + d * g["hit_t"][tri][:, None].astype(np.float64)
    assert np.abs(g["uv"][tri] - _affine_uv(hit)).max() < 2e-4


# ------------------------------------------------------------------ edge cases and errors
def test_depth_zero_no_lights_no_hits_and_empty_mesh():
    cam = rt.Camera(pitch=0.05)
    spec = stock_spec()
    g = render(spec, 160, 120, 0, cam)
    r = to_oracle(spec).render(160, 120, max_depth=0, cam=ocam(cam))
    assert_parity(compare(g, r))
    assert g["stats"]["rays_reflect"] == 0
    from common import SceneSpec, plane, sphere
    dark = SceneSpec([sphere((0, 0, 0), 2.5, (1.0, 0.0, 1.0, (1, 1, 1))), plane((0, 1, 0), -4.0)])   # no lights at all
    g = render(dark, 160, 120, 3, cam)
    r = to_oracle(dark).render(160, 120, max_depth=3, cam=ocam(cam))
    assert_parity(compare(g, r))
    assert g["stats"]["rays_shadow"] == 0 and g["rgb8"].max() == 0
    away = SceneSpec([sphere((0, 0, -50), 1.0, is_light=True)])        # nothing in view
    g = render(away, 64, 48, 3, cam)
    assert (g["hit_id"] == -1).all() and g["rgb8"].max() == 0


def test_degenerate_and_duplicate_triangles():
    """Zero-area triangles never hit (det = 0, main.cpp:113); coincident triangles tie on t and
    the first in scene order must win (main.cpp:151) even through the LBVH."""
    from common import SceneSpec
    v = np.array([[-1, -1, 2], [1, -1, 2], [0, 1, 2],      # tri 0
                  [-1, -1, 2], [1, -1, 2], [0, 1, 2],      # tri 1 == tri 0
                  [0, 0, 1], [0, 0, 1], [0, 0, 1],         # zero area
                  [-3, -3, 4], [3, -3, 4], [0, 3, 4]], np.float32)
    spec = SceneSpec(stock_spec().analytic[1:3], v, np.arange(12, dtype=np.int32), (0.5, 1.0, 1.0, (1, 1, 1)))
    cam = rt.Camera(pitch=0.05)
    r = to_oracle(spec).render(160, 120, max_depth=2, cam=ocam(cam))
    assert (r["hit_id"] == 2).any() and not (r["hit_id"] == 3).any() and not (r["hit_id"] == 4).any()
    for accel, lm in [(rt.ACCEL_LINEAR, 4), (rt.ACCEL_LBVH, 1), (rt.ACCEL_LBVH, 4)]:
        g = render(spec, 160, 120, 2, cam, accel, lm)
        assert np.array_equal(g["hit_id"], r["hit_id"])
        assert np.array_equal(g["rgb32"].view(np.uint32), r["rgb32"].view(np.uint32))


def test_argument_errors_are_reported_not_executed():
    with rt.Context(0) as ctx:
        with pytest.raises(rt.RtError) as e:
            ctx.render_frame(16, 16, max_depth=1)                    # no scene yet
        assert e.value.code == -4
        with pytest.raises(rt.RtError):
            lit_plane = rt.Plane((0, 1, 0), -4.0)
            lit_plane.is_light = 1
            ctx.set_analytic([lit_plane])                            # main.cpp:178 needs a Sphere
        ctx.set_analytic(rt.stock_scene())
        with pytest.raises(rt.RtError):
            ctx.render_frame(16, 16, max_depth=rt.RT_MAX_DEPTH + 1)
        with pytest.raises(rt.RtError):
            ctx.render_frame(16, 16, 4, 20, max_depth=1)             # rows past the frame
        with pytest.raises(rt.RtError):
            ctx.set_mesh(np.zeros((3, 3), np.float32), np.array([0, 1, 3], np.int32))   # index out of range
        with pytest.raises(rt.RtError):
            ctx.set_mesh(np.zeros((3, 3), np.float32), np.array([0, 1], np.int32))      # not a multiple of 3
        with pytest.raises(rt.RtError):
            ctx.render_rows_device(16, 16, rt.RowSet(0, 4, 2, 12), 1)                   # band set leaves the frame
        assert ctx.render_frame(16, 16, max_depth=1)["rgb8"].shape == (16, 16, 3)       # context still usable
    with pytest.raises(rt.RtError):
        rt.Context(10_000)


def test_per_triangle_materials_and_material_update():
    spec = soup_spec(n_tris=30, seed=3)
    cam = rt.Camera(pitch=0.05)
    r = to_oracle(spec).render(160, 120, max_depth=3, cam=ocam(cam))
    g = render(spec, 160, 120, 3, cam, rt.ACCEL_LBVH)
    assert_parity(compare(g, r))


# ------------------------------------------------------------------ the C++ host
HOST_BIN = os.environ.get("RT_HOST_BIN") or os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")      # (RT_HOST_BIN: another build of the host, e.g. a sanitized one)


def _read_ppm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3), data


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_application_default_frame(golden_stock):
    """The C++ host with no options = the reference program: stock scene, 800x600, depth 3."""
    with tempfile.TemporaryDirectory() as td:
        raw, up = os.path.join(td, "raw.ppm"), os.path.join(td, "up.ppm")
        subprocess.run([HOST_BIN, "--out", raw, "--raw-order"], check=True, capture_output=True, timeout=120)
        img, data = _read_ppm(raw)
        d = np.abs(img.astype(int) - golden_stock["rgb8"].astype(int)).max(-1)
        assert (d > 1).sum() == 0 and (d == 0).sum() >= MIN_FRACTION * d.size
        if (d == 0).all():
            assert hashlib.md5(data).hexdigest() == str(golden_stock["ppm_md5"])
        subprocess.run([HOST_BIN, "--out", up], check=True, capture_output=True, timeout=120)
        assert np.array_equal(_read_ppm(up)[0][::-1], img)           # upright file = buffer rows reversed


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_application_loads_an_obj_and_writes_float_and_id_planes():
    v, q = scenes.cube_quads()
    spec = cube_spec(mirror=False, mesh_reflect=0.5)
    with tempfile.TemporaryDirectory() as td:
        obj, ppm, pfm, ids = (os.path.join(td, n) for n in ("cube.obj", "f.ppm", "f.pfm", "ids.bin"))
        scenes.write_obj(obj, v, q)
        subprocess.run([HOST_BIN, "--obj", obj, "--no-mirror", "--mesh-reflect", "0.5", "--width", "400", "--height", "300",
                        "--depth", "4", "--out", ppm, "--raw-order", "--pfm", pfm, "--ids", ids],
                       check=True, capture_output=True, timeout=120)
        img, _ = _read_ppm(ppm)
        hid = np.fromfile(ids, np.int32).reshape(300, 400)
        raw = open(pfm, "rb").read().split(b"\n", 3)[3]
        f32 = np.frombuffer(raw, np.float32).reshape(300, 400, 3)
    cam = rt.camera_for_width(400)
    r = to_oracle(spec).render(400, 300, max_depth=4, cam=ocam(cam), threads=8)
    assert_parity(compare(dict(rgb8=img, hit_id=hid, rgb32=f32), r))


def _read_exr(path):
    """Reader for the uncompressed scanline EXR files the host writes: {channel: (H, W) array}, top row first."""
    import struct
    b = open(path, "rb").read()
    assert b[:4] == bytes([0x76, 0x2f, 0x31, 0x01]) and b[4] == 2
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        n = struct.unpack_from("<i", b, pos)[0]; pos += 4
        attrs[name] = (typ, b[pos:pos + n]); pos += n
    pos += 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    chans, cp, cl = [], 0, attrs["channels"][1]
    while cl[cp] != 0:
        e = cl.index(b"\0", cp); nm = cl[cp:e].decode(); cp = e + 1
        ptype = struct.unpack_from("<i", cl, cp)[0]; cp += 16
        chans.append((nm, ptype))
    assert [c for c, _ in chans] == sorted(c for c, _ in chans)
    offs = struct.unpack_from("<%dQ" % H, b, pos)
    out = {nm: np.zeros((H, W), np.float32 if pt == 2 else np.uint32) for nm, pt in chans}
    for y in range(H):
        yy, size = struct.unpack_from("<2i", b, offs[y])
        assert yy == y and size == 4 * W * len(chans)
        q = offs[y] + 8
        for nm, pt in chans:
            out[nm][y] = np.frombuffer(b, np.float32 if pt == 2 else np.uint32, W, q)
            q += 4 * W
    return out


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_relaxed_obj_with_mtl_materials_and_exr_output():
    """Materials come from the .mtl (per-triangle table), the frame goes out as EXR: checked against the
    oracle given the same per-triangle materials."""
    from common import SceneSpec, stock_analytic
    v, q = scenes.cube_quads()
    v = np.asarray(v, np.float32)
    mats = [(0.0, 1.0, 1.0, (1.0, 1.0, 1.0)), (0.0, 1.0, 0.5, (0.8, 0.1, 0.1)), (0.9, 1.0, 0.9, (1.0, 1.0, 1.0))]
    with tempfile.TemporaryDirectory() as td:
        obj, exr, ppm = (os.path.join(td, n) for n in ("c.obj", "f.exr", "f.ppm"))
        with open(os.path.join(td, "c.mtl"), "w") as f:
            f.write("newmtl red\nKd 0.8 0.1 0.1\nKs 0.5 0.25 0.1\nillum 2\nnewmtl mirror\nKs 0.9 0.9 0.9\nillum 3\n")
        with open(obj, "w") as f:
            f.write("mtllib c.mtl\r\n")
            for p in v:
                f.write("v   %r %r\t%r\r\n" % (float(p[0]), float(p[1]), float(p[2])))
            for k, quad in enumerate(q):
                if k == 2: f.write("usemtl red\r\n")
                if k == 4: f.write("usemtl mirror\r\n")
                f.write("f  " + "  ".join(str(int(c) + 1) for c in quad) + "\r\n")
        subprocess.run([HOST_BIN, "--obj", obj, "--relaxed", "--no-mirror", "--width", "400", "--height", "300", "--depth", "3",
                        "--out", ppm, "--raw-order", "--exr", exr], check=True, capture_output=True, timeout=120)
        img, _ = _read_ppm(ppm)
        ex = _read_exr(exr)
        assert sorted(ex) == ["B", "G", "N.x", "N.y", "N.z", "R", "Z", "id"]
    idx = np.array([[a, b, c, a, c, d] for a, b, c, d in q], np.int32).reshape(-1)
    tri_mat = np.repeat([0, 0, 1, 1, 2, 2], 2).astype(np.int32)
    spec = SceneSpec(stock_analytic(mirror=False), v, idx, mats=mats, tri_material=tri_mat)
    r = to_oracle(spec).render(400, 300, max_depth=3, cam=ocam(rt.camera_for_width(400)), threads=8)
    f32 = np.stack([ex["R"], ex["G"], ex["B"]], -1)[::-1]             # EXR is top row first
    hid = ex["id"][::-1].astype(np.int64) - 1
    assert_parity(compare(dict(rgb8=img, hit_id=hid.astype(np.int32), rgb32=np.ascontiguousarray(f32)), r))
    assert (hid >= 4).sum() > 1000                                       # the cube is in view
    seen = hid >= 0
    assert np.array_equal(ex["Z"][::-1][seen].view(np.uint32), r["hit_t"][seen].view(np.uint32))
    nrm = np.stack([ex["N.x"], ex["N.y"], ex["N.z"]], -1)[::-1]
    assert np.abs(np.linalg.norm(nrm[seen], axis=1) - 1.0).max() < 1e-5 and (nrm[~seen] == 0).all()


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_multi_frame_draw_loop_rerenders_with_a_moving_camera():
    """--frames N: draw(dt) re-renders N - 1 times; the last frame equals the oracle at the camera of that tick."""
    spec = stock_spec()
    N = 4
    with tempfile.TemporaryDirectory() as td:
        ppm = os.path.join(td, "f.ppm")
        out = subprocess.run([HOST_BIN, "--frames", str(N), "--width", "320", "--height", "240", "--out", ppm, "--raw-order"],
                             check=True, capture_output=True, text=True, timeout=120).stdout
        img, _ = _read_ppm(ppm)
    assert "%d frames" % N in out and "sustained" in out
    # draw(dt) advances a double clock by float(1/60) per tick; the swing is 1.5f * sinf((float)clock * 2.0f)
    clock = float(np.float32(1.0) / np.float32(60.0)) * (N - 1)
    swing = np.float32(1.5) * np.float32(np.sin(np.float32(np.float32(clock) * np.float32(2.0)), dtype=np.float32))
    base = rt.camera_for_width(320)
    cam = rt.Camera(origin=(float(np.float32(base.origin[0]) + swing), base.origin[1], base.origin[2]), pitch=base.pitch,
                    off_x=float(np.float32(base.off_x) + swing), off_y=base.off_y, plane_z=base.plane_z)
    r = to_oracle(spec).render(320, 240, max_depth=3, cam=ocam(cam), threads=8)
    d = np.abs(img.astype(int) - r["rgb8"].astype(int)).max(-1)
    assert (d > 1).mean() < 0.001


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_wobble_moves_the_mesh_between_ticks_through_the_refit():
    """--wobble A: every tick the mesh vertices move and rt_update_mesh_vertices refits the tree.  The
    run reports the refit time; its frame differs from the rigid run's, whose camera is the same.  (What
    a refitted frame must equal is checked at the library level, against a fresh build and the oracle.)"""
    v, i, _ = scenes.atrium_mesh(detail=0.25)
    tris = [tuple(int(x) for x in i[k:k + 3]) for k in range(0, len(i), 3)]
    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, "m.obj")
        scenes.write_obj(obj, v, tris)
        frames = {}
        for name, extra in (("rigid", []), ("wobble", ["--wobble", "0.3"])):
            ppm = os.path.join(td, name + ".ppm")
            out = subprocess.run([HOST_BIN, "--obj", obj, "--no-mirror", "--frames", "3", "--width", "320", "--height", "240",
                                  "--out", ppm, "--raw-order"] + extra, check=True, capture_output=True, text=True, timeout=120).stdout
            frames[name] = _read_ppm(ppm)[0]
            assert ("refit" in out) == (name == "wobble"), out
    assert (frames["rigid"] != frames["wobble"]).any(-1).mean() > 0.02


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_smooth_flag_shades_with_the_vn_normals_of_the_file():
    """--relaxed --smooth: the OBJ's `vn` normals reach rt_set_triangle_normals; the frame equals the
    oracle's with the same corner normals (the float text round-trips exactly at %.9g)."""
    from common import SceneSpec
    v, i, n = scenes.uv_sphere((0.0, -1.0, 1.0), 2.0, 8, 12)
    with tempfile.TemporaryDirectory() as td:
        obj, ppm, ids = (os.path.join(td, x) for x in ("s.obj", "f.ppm", "ids.bin"))
        with open(obj, "w") as f:
            for p in v:
                f.write("v %.9g %.9g %.9g\n" % tuple(p))
            for p in n:
                f.write("vn %.9g %.9g %.9g\n" % tuple(p))
            for k in range(0, len(i), 3):
                f.write("f " + " ".join("%d//%d" % (j + 1, j + 1) for j in i[k:k + 3]) + "\n")
        out = subprocess.run([HOST_BIN, "--obj", obj, "--relaxed", "--smooth", "--no-mirror", "--width", "320", "--height", "240",
                              "--out", ppm, "--raw-order", "--ids", ids], check=True, capture_output=True, text=True, timeout=120).stdout
        img, _ = _read_ppm(ppm)
        hid = np.fromfile(ids, np.int32).reshape(240, 320)
    assert "%d of %d triangles carry vn normals" % (len(i) // 3, len(i) // 3) in out
    spec = SceneSpec(stock_analytic(mirror=False), v, i, (0.0, 1.0, 1.0, (1.0, 1.0, 1.0)))
    osc = to_oracle(spec)
    osc.set_corner_normals(scenes.corner_normals(n, i))
    r = osc.render(320, 240, max_depth=3, cam=ocam(rt.camera_for_width(320)))
    assert np.array_equal(hid, r["hit_id"])
    d = np.abs(img.astype(int) - r["rgb8"].astype(int)).max(-1)
    assert d.max() <= 1 and (d == 0).mean() >= MIN_FRACTION
    assert (hid >= 3).sum() > 2000


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_exr_carries_u_and_v_when_the_obj_has_texcoords():
    """--relaxed --exr on an OBJ with `f v/vt`: the vt coordinates reach rt_set_triangle_texcoords and the EXR
    gains u and v channels, equal to the affine function the file's coordinates were sampled from."""
    v, i, _ = scenes.uv_sphere((0.0, -1.0, 1.0), 2.0, 8, 12)
    uv = _affine_uv(v)
    with tempfile.TemporaryDirectory() as td:
        obj, exr = os.path.join(td, "s.obj"), os.path.join(td, "f.exr")
        with open(obj, "w") as f:
            for p in v:
                f.write("v %.9g %.9g %.9g\n" % tuple(p))
            for t in uv:
                f.write("vt %.9g %.9g\n" % tuple(t))
            for k in range(0, len(i), 3):
                f.write("f " + " ".join("%d/%d" % (j + 1, j + 1) for j in i[k:k + 3]) + "\n")
        subprocess.run([HOST_BIN, "--obj", obj, "--relaxed", "--no-mirror", "--width", "320", "--height", "240", "--exr", exr,
                        "--out", ""], check=True, capture_output=True, timeout=120)
        ch = _read_exr(exr)
    assert "u" in ch and "v" in ch
    tri = ch["id"] >= 3 + 1                                                 # EXR ids are hit id + 1; 3 analytic primitives
    assert tri.sum() > 2000 and not ch["u"][~tri].any() and not ch["v"][~tri].any()
    assert ch["u"][tri].std() > 0.01 and np.isfinite(ch["u"]).all() and np.isfinite(ch["v"]).all()


# ------------------------------------------------------------------ many lights / no lights through the LBVH stages
def _ring_of_lights(n):
    from common import sphere
    out = []
    for k in range(n):
        a = 2.0 * np.pi * k / n
        c = (0.5 + 0.5 * np.cos(3 * a), 0.5 + 0.5 * np.sin(5 * a), 0.6)
        out.append(sphere((4.0 * np.cos(a), 3.0 + 0.5 * np.sin(2 * a), 1.0 + 4.0 * np.sin(a)), 0.05,
                          (0.0, 1.0, 1.0, tuple(0.05 * x for x in c)), True))
    return out


@pytest.mark.parametrize("n_lights", [0, 1, 31, 32, 33, 70])
def test_light_counts_across_the_32_light_chunks(n_lights):
    """Shadow rays are traced 32 lights at a time; counts on and around the chunk edges."""
    from common import SceneSpec, plane
    v, i, m = scenes.atrium_mesh(detail=0.004)
    spec = SceneSpec(_ring_of_lights(n_lights) + [plane((0.0, 1.0, 0.0), -4.5, (0.3, 1.0, 1.0, (1.0, 1.0, 1.0)))], v, i,
                     (0.5, 1.0, 1.0, (1.0, 1.0, 1.0)))
    cam = rt.camera_for_width(96)
    r = to_oracle(spec).render(96, 54, max_depth=2, cam=ocam(cam), threads=16)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
        ctx.set_option(rt.OPT_LEAF_MAX, 4)
        to_product(ctx, spec)
        ctx.set_camera(cam)
        g = ctx.render_frame(96, 54, max_depth=2)
        g["stats"] = ctx.stats()
        # the same layout again: the second frame makes the claim order (one row of work units per traversal launch, several
        # launches per level once there is more than one chunk of lights), the third runs in it
        for _ in range(2):
            again = ctx.render_frame(96, 54, max_depth=2)
            for k in ("rgb8", "hit_id", "rgb32"):
                assert np.array_equal(again[k].view(np.uint8), g[k].view(np.uint8)), k
            assert rays(ctx.stats()) == rays(g["stats"])
    assert g["stats"]["accel"] == 1
    c = compare(g, r)
    assert c["id_mismatch"] == 0 and c["rgb8_over_tol"] == 0, c
    assert c["rgb32_max_abs"] <= RGB32_TOL * max(1, n_lights), c
    assert rays(g["stats"]) == orays(r)


def test_fused_and_staged_pipelines_agree_bit_for_bit():
    """The one-kernel-per-level pipeline and the staged pipeline share their arithmetic (rt_device.h): same bytes,
    same ray counts, at a depth the shading kernel holds in registers and at one it walks through the records
    plane (more than 9 levels)."""
    spec = atrium_spec(detail=0.02)
    cam = rt.camera_for_width(320)
    for depth in (4, 11):
        outs, stats = [], []
        for pipeline in (rt.PIPELINE_FUSED, rt.PIPELINE_WAVEFRONT):
            with rt.Context(0) as ctx:
                ctx.set_option(rt.OPT_PIPELINE, pipeline)
                to_product(ctx, spec)
                ctx.set_camera(cam)
                outs.append(ctx.render_frame(320, 180, max_depth=depth))
                stats.append(ctx.stats())
        for k in ("rgb8", "hit_id", "rgb32"):
            assert np.array_equal(outs[0][k].view(np.uint8), outs[1][k].view(np.uint8)), (k, depth)
        assert rays(stats[0]) == rays(stats[1])
        assert stats[0]["n_launches"] == depth + 2 and stats[1]["n_launches"] > 2 * depth
    with rt.Context(0) as ctx:
        with pytest.raises(rt.RtError):
            ctx.set_option(rt.OPT_PIPELINE, 3)


def test_partitioned_renderer_is_ordered_by_streams_alone():
    """PartitionedRenderer (world = 1) with a camera that changes every frame, on a non-default caller
    stream, read back with stream-ordered copies only: no device-wide synchronisation anywhere between the
    frames.  Every copy must hold ITS frame (a renderer that is not ordered with the caller's stream hands
    back a torn or a stale one)."""
    import torch
    from raytracer_amd.dist import PartitionedRenderer
    spec = atrium_spec(detail=0.02)
    W, H, D = 320, 200, 2
    cams = [rt.Camera(origin=(0.2 * k - 0.6, -1.5 + 0.1 * k, -6.0), pitch=8.0 / W, off_x=-4.0, off_y=-3.4) for k in range(6)]
    with rt.Context(0) as ctx:
        to_product(ctx, spec)
        ctx.set_option(rt.OPT_TIMING, 0)
        pr = PartitionedRenderer(ctx, W, H, D)
        side = torch.cuda.Stream()
        hosts, events = [], []
        with torch.cuda.stream(side):
            for cam in cams:
                ctx.set_camera(cam)
                out = pr.render()
                h = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
                h.copy_(out, non_blocking=True)
                e = torch.cuda.Event()
                e.record(side)
                hosts.append(h); events.append(e)
        for e in events:
            e.synchronize()
        got = [h.numpy().copy() for h in hosts]
        for cam, g in zip(cams, got):
            ctx.set_camera(cam)
            want = ctx.render_frame(W, H, max_depth=D, want=("rgb8",))["rgb8"]
            assert np.array_equal(g, want)
    assert any(not np.array_equal(got[0], g) for g in got[1:])      # the camera did move


def test_frames_in_flight_each_hold_their_own_frame():
    """Three contexts of one scene as three frame slots, a different camera per slot, seven frames enqueued
    before the first is waited for: every Frame must hold the picture of ITS slot's camera (slots own their
    buffers and streams), and the frames of one slot are the same picture every time."""
    import torch
    from raytracer_amd.dist import PartitionedRenderer
    spec = atrium_spec(detail=0.02)
    W, H, D = 320, 200, 2
    cams = [rt.Camera(origin=(0.3 * k - 0.3, -1.5 + 0.2 * k, -6.0), pitch=8.0 / W, off_x=-4.0, off_y=-3.4) for k in range(3)]
    ctxs = [rt.Context(0) for _ in cams]
    try:
        for c, cam in zip(ctxs, cams):
            to_product(c, spec)
            c.set_option(rt.OPT_TIMING, 0)
            c.set_camera(cam)
        pr = PartitionedRenderer(ctxs, W, H, D)
        assert pr.frames_in_flight == 3
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            frames = [pr.render_async() for _ in range(7)]
            hosts = []
            for f in frames:
                h = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
                h.copy_(f.wait(), non_blocking=True)
                hosts.append(h)
            done = torch.cuda.Event()
            done.record(side)
        done.synchronize()
        want = [c.render_frame(W, H, max_depth=D, want=("rgb8",))["rgb8"] for c in ctxs]
        for k, h in enumerate(hosts):
            assert np.array_equal(h.numpy(), want[k % 3]), k
        assert not np.array_equal(want[0], want[1]) and not np.array_equal(want[1], want[2])
        assert np.array_equal(pr.render(slot=1).cpu().numpy(), want[1])
    finally:
        for c in ctxs:
            c.close()


def test_frame_slots_of_one_context_keep_frames_apart():
    """One context, one scene, one hierarchy, three frame slots (rt_render_rows_device_slot): the camera changes before
    every enqueue, seven frames are enqueued on three streams before the first is waited for, and every slot's
    buffer holds the picture of the camera its frames were enqueued with.  A vertex update then has to wait for all
    three slots before it overwrites the triangles they read."""
    import torch
    from raytracer_amd.dist import PartitionedRenderer
    spec = atrium_spec(detail=0.02)
    W, H, D = 320, 200, 2
    cams = [rt.Camera(origin=(0.3 * k - 0.3, -1.5 + 0.2 * k, -6.0), pitch=8.0 / W, off_x=-4.0, off_y=-3.4) for k in range(3)]
    with rt.Context(0) as ctx:
        to_product(ctx, spec)
        ctx.set_option(rt.OPT_TIMING, 0)
        want = []
        for cam in cams:
            ctx.set_camera(cam)
            want.append(ctx.render_frame(W, H, max_depth=D, want=("rgb8",))["rgb8"])
        pr = PartitionedRenderer(ctx, W, H, D, frames_in_flight=3)
        assert pr.frames_in_flight == 3 and [s.index for s in pr.slots] == [0, 1, 2]
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            frames = []
            for k in range(7):
                ctx.set_camera(cams[k % 3])
                frames.append(pr.render_async())
            # moving the mesh must not touch what the seven frames in flight still read
            moved = _moved(spec.verts, 1)
            ctx.update_mesh_vertices(moved)
            hosts = [f.wait().cpu().numpy().copy() for f in frames]
        for k, h in enumerate(hosts):
            assert np.array_equal(h, want[k % 3]), k
        assert not np.array_equal(want[0], want[1]) and not np.array_equal(want[1], want[2])
        # and after the update every slot renders the moved mesh
        ctx.set_camera(cams[0])
        after = [pr.render(slot=k).cpu().numpy().copy() for k in range(3)]
        fresh = ctx.render_frame(W, H, max_depth=D, want=("rgb8",))["rgb8"]
        for a in after:
            assert np.array_equal(a, fresh)
        assert not np.array_equal(fresh, want[0])
        with pytest.raises(rt.RtError):
            ctx.render_rows_device(W, H, rt.RowSet(0, 1, 1, H), D, d_rgb8=pr.slots[0].local.data_ptr(), slot=rt.MAX_FRAME_SLOTS)


def test_hierarchy_option_changed_with_frames_in_flight_on_other_slots():
    """RT_OPT_LEAF_MAX changed between frames: the next frame rebuilds the hierarchy, freeing the nodes and the triangle
    copy that frames already enqueued on other slots still read (and that their captured graphs point at).  The rebuild
    has to wait for every slot first; every frame, before and after, must be the picture of its camera."""
    import torch
    from raytracer_amd.dist import PartitionedRenderer
    spec = atrium_spec(detail=0.02)
    W, H, D = 320, 200, 2
    cams = [rt.Camera(origin=(0.3 * k - 0.3, -1.5 + 0.2 * k, -6.0), pitch=8.0 / W, off_x=-4.0, off_y=-3.4) for k in range(3)]
    with rt.Context(0) as ctx:
        to_product(ctx, spec)
        ctx.set_option(rt.OPT_TIMING, 0)
        want = []
        for cam in cams:
            ctx.set_camera(cam)
            want.append(ctx.render_frame(W, H, max_depth=D, want=("rgb8",))["rgb8"])
        pr = PartitionedRenderer(ctx, W, H, D, frames_in_flight=3)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(2):                       # every slot has replayed its graph at least once
                for k in range(3):
                    ctx.set_camera(cams[k])
                    pr.render_async()
            frames = []
            for k in range(6):
                ctx.set_camera(cams[k % 3])
                if k == 4:
                    ctx.set_option(rt.OPT_LEAF_MAX, 4)     # frames 0..3 are in flight; frame 4 rebuilds
                frames.append(pr.render_async())
            hosts = [f.wait().cpu().numpy().copy() for f in frames]
        assert ctx.stats()["bvh_leaf_max"] == 4
        for k, h in enumerate(hosts):
            assert np.array_equal(h, want[k % 3]), k


def _wires_and_walls_spec():
    """What whole-triangle boxes handle worst: long thin triangles that cross the scene diagonally, a huge
    two-triangle wall and floor, and small triangles scattered in the space their boxes cover."""
    small = soup_spec(n_tris=300, seed=33)
    v = [small.verts * np.array([1.0, 1.0, 1.0], np.float32) * 0.35 + np.array([0.0, -1.0, 1.0], np.float32)]
    idx = [small.indices]
    n0 = len(small.verts)
    extra = []
    for j in range(12):                                       # diagonal wires, 2 cm wide
        a = np.array([-6.0 + j, -3.5, -1.0 + 0.3 * j]); b = np.array([5.0 - 0.5 * j, 3.0, 9.0 - 0.2 * j])
        w = np.array([0.0, 0.02, 0.0])
        extra += [a, b, b + w, a, b + w, a + w]
    extra += [np.array(p, float) for p in ((-8, -4, -2), (-8, -4, 12), (8, -4, 12), (-8, -4, -2), (8, -4, 12), (8, -4, -2))]      # floor
    extra += [np.array(p, float) for p in ((-8, -4, 12), (-8, 6, 12), (8, 6, 12), (-8, -4, 12), (8, 6, 12), (8, -4, 12))]        # back wall
    ev = np.array(extra, np.float32)
    v.append(ev)
    idx.append(np.arange(len(ev), dtype=np.int32) + n0)
    verts, indices = np.concatenate(v), np.concatenate(idx)
    from common import SceneSpec
    mats = [(0.0, 1.0, 1.0, (1.0, 1.0, 1.0)), (0.6, 0.7, 0.5, (0.9, 0.3, 0.2)), (1.0, 0.0, 1.0, (1.0, 1.0, 1.0)), (0.3, 1.0, 0.0, (0.2, 0.8, 0.4))]
    tm = (np.arange(len(indices) // 3) % 4).astype(np.int32)
    a = [sphere_((2.0, 5.0, 1.0), 0.1, True, (0.7, 0.7, 0.9)), sphere_((-2.0, 5.0, -3.0), 0.1, True, (0.9, 0.9, 0.4))]
    return SceneSpec(a, verts, indices, mats=mats, tri_material=tm)


def sphere_(p, r, light, col):
    return dict(kind="sphere", p=p, s=r, mat=(0.0, 1.0, 1.0, col), is_light=light)


@pytest.mark.parametrize("builder", [rt.BUILDER_LBVH, rt.BUILDER_PLOC, rt.BUILDER_SAH])
def test_cut_triangle_references_change_the_tree_and_nothing_else(builder):
    """RT_OPT_SPLIT: large triangles enter the hierarchy as several references with clipped boxes.  The frame (ids,
    colours, ray counts) must equal the uncut hierarchy's and the oracle's brute-force scan; only the tree grows and
    the triangle tests fall.  After a vertex update (references fall back to whole-triangle boxes) the same holds."""
    spec = _wires_and_walls_spec()
    cam = rt.Camera(pitch=0.05)
    W, H, D = 160, 120, 3
    r = to_oracle(spec).render(W, H, max_depth=D, cam=ocam(cam))
    out = {}
    for split in (1, 0, 4):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
            ctx.set_option(rt.OPT_BUILDER, builder)
            ctx.set_option(rt.OPT_SPLIT, split)
            ctx.set_option(rt.OPT_COUNT_TESTS, 1)
            to_product(ctx, spec)
            ctx.set_camera(cam)
            g = ctx.render_frame(W, H, max_depth=D)
            st = ctx.stats()
            assert_parity(compare(g, r))
            assert rays(st) == orays(r)
            out[split] = (g, st)
            if split == 0:
                v = _moved(spec.verts, 1)
                ctx.update_mesh_vertices(v)
                g2 = ctx.render_frame(W, H, max_depth=D)
                st2 = ctx.stats()
                assert st2["bvh_nodes"] == st["bvh_nodes"]
                r2 = to_oracle(SceneSpecCopy(spec, v)).render(W, H, max_depth=D, cam=ocam(cam))
                assert_parity(compare(g2, r2))
                assert rays(st2) == orays(r2)
    n_tris = spec.n_tris
    assert out[1][1]["bvh_nodes"] == n_tris - 1                       # never cut: one reference per triangle
    assert out[0][1]["bvh_nodes"] > out[4][1]["bvh_nodes"] > n_tris - 1
    for k in ("rgb8", "hit_id", "rgb32"):
        assert np.array_equal(out[0][0][k].view(np.uint8), out[1][0][k].view(np.uint8)), k
    assert out[0][1]["tri_tests"] < 0.6 * out[1][1]["tri_tests"]      # what the cut is for
    with rt.Context(0) as ctx:
        with pytest.raises(rt.RtError):
            ctx.set_option(rt.OPT_SPLIT, 33)


def test_supplied_obj_workload_renders_like_the_oracle(tmp_path):
    """bench.py's --obj / RT_SPONZA_OBJ path end to end: a mesh written as OBJ text, read back through the host
    loader's C ABI (libobjhost.so), fitted to the courtyard camera, rendered on the GPU, equals the oracle on the
    very same workload (ids exact, colours within 1/255, ray counts exact)."""
    from raytracer_amd import workloads
    v, i, m = scenes.atrium_mesh(detail=0.01)
    path = str(tmp_path / "sponza.obj")
    scenes.write_obj(path, v * 37.0, i.reshape(-1, 3))          # other units: the workload rescales
    wl = workloads.sponza(W=320, H=180, max_depth=3, obj=path)
    assert wl.name.startswith("sponza.obj(") and wl.n_tris == len(i) // 3
    with rt.Context(0) as ctx:
        wl.load(ctx)
        g = ctx.render_frame(wl.W, wl.H, max_depth=wl.max_depth)
        st = ctx.stats()
    assert st["accel"] == 1
    sc = O.scene_from_workload(wl, bvh=True)
    r = sc.render(wl.W, wl.H, max_depth=wl.max_depth, cam=O.camera_from(wl.camera), use_bvh=True, threads=8)
    assert_parity(compare(g, r))
    assert rays(st) == orays(r)
    assert (g["hit_id"] >= 0).mean() > 0.5


def test_a_scene_of_nothing_but_large_triangles_keeps_its_reference_count_in_budget():
    """Every triangle here is long against the scene, so an uncapped grid would ask for hundreds of cells each; the
    builder coarsens the grid until the cells fit its budget (4 per triangle + 4 M), and the frame still equals the
    uncut hierarchy's and the oracle's."""
    n = 12000
    r = np.random.default_rng(4242).random((n, 3, 3))
    tri = np.empty((n, 3, 3), np.float64)
    tri[:, 0] = r[:, 0] * [10.0, 6.0, 8.0] + [-5.0, -4.0, -1.0]
    tri[:, 1] = tri[:, 0] + (r[:, 1] - 0.5) * [16.0, 10.0, 14.0]          # long edges in every direction
    tri[:, 2] = tri[:, 0] + (r[:, 2] - 0.5) * [0.6, 0.6, 0.6]
    from common import SceneSpec
    spec = SceneSpec([sphere_((2.0, 5.0, 1.0), 0.1, True, (0.7, 0.7, 0.9))], tri.reshape(-1, 3).astype(np.float32),
                     np.arange(3 * n, dtype=np.int32), (0.3, 1.0, 1.0, (0.8, 0.8, 0.8)))
    cam = rt.Camera(pitch=0.1)
    W, H, D = 80, 60, 2
    frames = {}
    for split in (1, 0):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
            ctx.set_option(rt.OPT_SPLIT, split)
            to_product(ctx, spec)
            ctx.set_camera(cam)
            frames[split] = (ctx.render_frame(W, H, max_depth=D), ctx.stats())
            again = ctx.render_frame(W, H, max_depth=D)           # no rebuild between frames: same tree, same bytes
            assert ctx.stats()["bvh_nodes"] == frames[split][1]["bvh_nodes"]
            assert np.array_equal(again["rgb8"], frames[split][0]["rgb8"])
    assert frames[1][1]["bvh_nodes"] == n - 1
    assert n - 1 < frames[0][1]["bvh_nodes"] <= 4 * n + (4 << 20)
    for k in ("rgb8", "hit_id", "rgb32"):
        assert np.array_equal(frames[0][0][k].view(np.uint8), frames[1][0][k].view(np.uint8)), k
    o = to_oracle(spec).render(W, H, max_depth=D, cam=ocam(cam), threads=8)
    assert_parity(compare(frames[0][0], o))
    assert rays(frames[0][1]) == orays(o)


def test_maximum_depth_and_maximum_analytic_count():
    """The stated maxima: RT_MAX_DEPTH = 64 bounce levels (a hall of mirrors: two facing reflective planes, a mirror
    mesh between them, so chains really run 64 deep) through both pipelines, and RT_MAX_ANALYTIC = 1024 analytic
    primitives (512 of them lights: 16 chunks of 32) in one scene; one more of either is refused."""
    from common import SceneSpec, plane, sphere
    mirror = (0.9, 0.3, 1.0, (0.9, 0.9, 1.0))
    a = [plane((0.0, 0.0, 1.0), -6.0, mirror), plane((0.0, 0.0, -1.0), -9.0, mirror), plane((0.0, 1.0, 0.0), -4.0),
         sphere((1.0, 3.0, 2.0), 0.1, (0.0, 1.0, 1.0, (0.9, 0.9, 0.8)), True)]
    small = soup_spec(n_tris=60, seed=17)
    spec = SceneSpec(a, small.verts, small.indices, mats=[(0.8, 0.4, 1.0, (1.0, 1.0, 1.0)), (0.0, 1.0, 1.0, (0.8, 0.2, 0.2))],
                     tri_material=(np.arange(60) % 2).astype(np.int32))
    cam = rt.Camera(pitch=0.1)
    W, H, D = 80, 60, rt.RT_MAX_DEPTH
    r = to_oracle(spec).render(W, H, max_depth=D, cam=ocam(cam), threads=8)
    assert r["counts"]["reflect"] > 10 * W * H          # the chains are deep
    for accel, pipeline in ((rt.ACCEL_LINEAR, 0), (rt.ACCEL_LBVH, rt.PIPELINE_FUSED), (rt.ACCEL_LBVH, rt.PIPELINE_WAVEFRONT)):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_ACCEL, accel)
            ctx.set_option(rt.OPT_PIPELINE, pipeline)
            to_product(ctx, spec)
            ctx.set_camera(cam)
            g = ctx.render_frame(W, H, max_depth=D)
            st = ctx.stats()
            assert_parity(compare(g, r))
            assert rays(st) == orays(r)
            with pytest.raises(rt.RtError) as e:
                ctx.render_frame(W, H, max_depth=D + 1)
            assert e.value.code == -1
    # 1024 analytic primitives: 512 small light spheres on a ring, 511 plain spheres, one floor
    prims = [plane((0.0, 1.0, 0.0), -4.0)]
    for k in range(512):
        t = 2.0 * np.pi * k / 512
        prims.append(sphere((6.0 * np.cos(t), 4.0 + 0.5 * np.sin(5 * t), 4.0 + 6.0 * np.sin(t)), 0.03, (0.0, 1.0, 1.0, (0.004, 0.004, 0.004)), True))
    for k in range(511):
        t = 2.0 * np.pi * k / 511
        prims.append(sphere((3.0 * np.cos(t), -3.5 + 0.002 * k, 4.0 + 3.0 * np.sin(t)), 0.12, (0.2, 1.0, 1.0, (0.6, 0.7, 0.9))))
    assert len(prims) == rt.RT_MAX_ANALYTIC
    big = SceneSpec(prims, small.verts, small.indices, (0.0, 1.0, 1.0, (1.0, 1.0, 1.0)))
    W2, H2 = 48, 36
    r2 = to_oracle(big).render(W2, H2, max_depth=1, cam=ocam(cam), threads=8)
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        g = render(big, W2, H2, 1, cam, accel)
        assert_parity(compare(g, r2))
        assert rays(g["stats"]) == orays(r2)
    with rt.Context(0) as ctx:
        with pytest.raises(rt.RtError) as e:
            ctx.set_analytic([rt.Sphere((0.0, 0.0, 0.0), 1.0)] * (rt.RT_MAX_ANALYTIC + 1))
        assert e.value.code == -5
