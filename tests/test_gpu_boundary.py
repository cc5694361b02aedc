"""The host-buffer side of the boundary (rt_render_frame, the seam of Raytracer/main.cpp:259, 273, 281) and the
versioned statistics call, through the C ABI on the GPU."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from common import atrium_spec, compare, cube_spec, ocam, rt, to_oracle, to_product

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_frame(ctx, W, H, D):
    import torch
    rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    hid = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    rgb32 = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    ctx.render_rows_device(W, H, rt.RowSet(0, 1, 1, H), D, rgb8.data_ptr(), hid.data_ptr(), rgb32.data_ptr(), None,
                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rgb8.cpu().numpy(), hid.cpu().numpy(), rgb32.cpu().numpy()


def test_host_frame_equals_the_device_frame_staged_pinned_and_in_part():
    W, H, D = 1000, 562, 3          # > 1 MiB of 8-bit colour: the staged hand-over goes in pieces
    with rt.Context(0) as ctx:
        to_product(ctx, atrium_spec(detail=0.05))
        ctx.set_camera(rt.Camera(origin=(0.0, -1.5, -6.0), pitch=8.0 / W, off_x=-4.0, off_y=-3.4))
        d8, did, d32 = _device_frame(ctx, W, H, D)
        assert (did >= 0).mean() > 0.5
        # all three planes through the pinned staging buffer
        g = ctx.render_frame(W, H, max_depth=D)
        assert np.array_equal(g["rgb8"], d8) and np.array_equal(g["hit_id"], did)
        assert np.array_equal(g["rgb32"].view(np.uint32), d32.view(np.uint32))
        assert ctx.stats()["staging_ms"] > 0.0
        # the reference's call: the 8-bit image alone, into a buffer the caller keeps (main.cpp:259)
        image = np.full((H, W, 3), 9, np.uint8)
        ctx.render_frame_into(W, H, image, max_depth=D)
        assert np.array_equal(image, d8)
        # a buffer from the library (pinned): the device writes it directly
        pinned = ctx.alloc_host_buffer((H, W, 3))
        pinned[:] = 9
        ctx.render_frame_into(W, H, pinned, max_depth=D)
        assert np.array_equal(pinned, d8)
        # a row range into the middle of that buffer (still inside it), the rest untouched
        pinned[:] = 9
        ctx.render_frame_into(W, H, pinned[100:228], max_depth=D, y0=100, y1=228)
        assert np.array_equal(pinned[100:228], d8[100:228]) and (pinned[:100] == 9).all() and (pinned[228:] == 9).all()
        ctx.free_host_buffer(pinned)
        del pinned
        with pytest.raises(rt.RtError):
            ctx.free_host_buffer(image)
        image[:] = 9
        ctx.render_frame_into(W, H, image, max_depth=D)
        assert np.array_equal(image, d8)
        # the tooling planes come over the same way
        p = ctx.render_frame_planes(W, H, max_depth=D)
        assert np.array_equal(p["rgb8"], d8) and np.array_equal(p["hit_id"], did)
        assert np.isfinite(p["normal"]).all() and (p["hit_t"][did >= 0] < 3e38).all()


def test_stats_for_a_caller_built_against_an_older_header():
    with rt.Context(0) as ctx:
        to_product(ctx, cube_spec(mirror=False, mesh_reflect=0.5))
        ctx.set_camera(rt.Camera(pitch=8.0 / 160))
        ctx.set_option(rt.OPT_COUNT_TESTS, 1)
        ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
        ctx.render_frame(160, 120, max_depth=2)
        full = ctx.stats()
        assert 0 < full["box_tests_occupied"] <= full["box_tests"]
        # a version-1 host's struct ended behind bvh_refs: exactly that prefix is written, nothing behind it
        old_size = rt.Stats.bvh_refs.offset + 4
        buf = (C.c_ubyte * C.sizeof(rt.Stats))(*([0xAB] * C.sizeof(rt.Stats)))
        assert ctx.L.rt_get_stats_sized(ctx._h, C.addressof(buf), old_size) == 0
        got = rt.Stats.from_buffer_copy(buf)
        assert got.rays_primary == full["rays_primary"] and got.bvh_refs == full["bvh_refs"] and got.box_tests == full["box_tests"]
        assert bytes(buf[old_size:]) == b"\xab" * (C.sizeof(rt.Stats) - old_size)
        # a larger struct than the library knows is filled as far as the library's goes
        big = (C.c_ubyte * (C.sizeof(rt.Stats) + 16))(*([0xCD] * (C.sizeof(rt.Stats) + 16)))
        assert ctx.L.rt_get_stats_sized(ctx._h, C.addressof(big), C.sizeof(rt.Stats) + 16) == 0
        assert bytes(big[C.sizeof(rt.Stats):]) == b"\xcd" * 16
        assert ctx.L.rt_get_stats_sized(ctx._h, C.addressof(buf), 0) == -1
        assert ctx.L.rt_get_stats_sized(ctx._h, C.addressof(buf), 6) == -1


GUARD_LIB = os.path.join(ROOT, "raytracer_amd", "variants", "guard.so")

_DROPPED_PUSH = r"""
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
from common import atrium_spec, rt, to_product
with rt.Context(0) as ctx:
    ctx.set_option(rt.OPT_PIPELINE, rt.PIPELINE_FUSED)
    to_product(ctx, atrium_spec(detail=0.05))
    ctx.set_camera(rt.Camera(origin=(0.0, -1.5, -6.0), pitch=8.0 / 320, off_x=-4.0, off_y=-3.4))
    ctx.render_frame(320, 200, max_depth=2)
    try:
        ctx.stats()
        print("NO ERROR")
    except rt.RtError as e:
        print("ERROR", e)
"""


@pytest.mark.skipif(not os.path.exists(GUARD_LIB), reason="the guarded variant is not built")
def test_guarded_build_reports_a_traversal_stack_that_dropped_a_child():
    """The fused kernel drops the far child when its stack is full (the host sizes the stack from the tree, so it never
    is).  A guarded build counts that; with the stack cut to one entry (RT_GUARD_STACK_CAP, read by guarded builds only)
    rt_get_stats must fail, and with the real size it must not."""
    def run(cap):
        env = dict(os.environ, RT_AMD_LIB=GUARD_LIB)
        env.pop("RT_GUARD_STACK_CAP", None)
        if cap:
            env["RT_GUARD_STACK_CAP"] = str(cap)
        return subprocess.run([sys.executable, "-c", _DROPPED_PUSH % {"root": ROOT}], env=env, capture_output=True, text=True,
                              timeout=300).stdout
    assert "NO ERROR" in run(0)
    out = run(1)
    assert "ERROR" in out and "guarded build" in out and "NO ERROR" not in out


HOST_BIN = os.environ.get("RT_HOST_BIN") or os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")      # (RT_HOST_BIN: another build of the host, e.g. a sanitized one)


def _ppm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_on_n_gpus_with_rccl_writes_the_single_context_frame(tmp_path):
    """`raytracer --gpus N`: one process, one context per GPU, bands round-robin, ncclAllGather of the slices, bands put
    back on GPU 0.  This box has one GPU, so N = 1 (a communicator of one, every step of the path still runs); the file
    must equal the single-context frame byte for byte, also with bands that do not divide the height and over the
    frames of the draw(dt) loop.  N > 1 is not measured on hardware here."""
    from raytracer_amd import scenes
    v, q = scenes.cube_quads()
    obj = str(tmp_path / "cube.obj")
    scenes.write_obj(obj, v, q)
    base = [HOST_BIN, "--obj", obj, "--no-mirror", "--mesh-reflect", "0.5", "--width", "400", "--height", "301", "--depth", "3", "--raw-order"]
    one, multi, odd = (str(tmp_path / n) for n in ("one.ppm", "multi.ppm", "odd.ppm"))
    subprocess.run(base + ["--out", one], check=True, capture_output=True, timeout=120)
    out = subprocess.run(base + ["--out", multi, "--gpus", "1"], check=True, capture_output=True, text=True, timeout=120).stdout
    assert "longest device of 1" in out
    subprocess.run(base + ["--out", odd, "--gpus", "1", "--band-rows", "7"], check=True, capture_output=True, timeout=120)
    ref = _ppm(one)
    assert ref.shape == (301, 400, 3) and ref.max() > 0
    assert np.array_equal(_ppm(multi), ref) and np.array_equal(_ppm(odd), ref)
    # the draw(dt) loop: the last of three frames with the camera moving, on both paths
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    subprocess.run(base + ["--out", a, "--frames", "3"], check=True, capture_output=True, timeout=120)
    subprocess.run(base + ["--out", b, "--frames", "3", "--gpus", "1"], check=True, capture_output=True, timeout=120)
    assert np.array_equal(_ppm(a), _ppm(b)) and not np.array_equal(_ppm(a), ref)
    # more GPUs than the box has is an error, not a crash
    import torch
    r = subprocess.run(base + ["--out", multi, "--gpus", str(torch.cuda.device_count() + 1)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rt_create" in r.stderr


def test_claim_order_changes_when_rays_are_traced_never_what_they_hit():
    """RT_OPT_CLAIM_ORDER: the first frame of a layout is preceded by a counting pass, later frames hand their work units out
    longest rays first.  Every plane and every ray count must be what they are with the option off: on the frame the order
    was made for, on a frame with the camera moved (the order is then stale), after asking for a new order, and with test
    counting on."""
    import torch
    W, H, D = 640, 400, 4

    def frame(ctx):
        rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        hid = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        rgb32 = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        ht = torch.zeros((H, W), dtype=torch.float32, device="cuda")
        ctx.render_rows_device(W, H, rt.RowSet(0, 1, 1, H), D, rgb8.data_ptr(), hid.data_ptr(), rgb32.data_ptr(), ht.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = ctx.stats()
        return (rgb8.cpu().numpy(), hid.cpu().numpy(), rgb32.cpu().numpy().view(np.uint32), ht.cpu().numpy().view(np.uint32),
                (st["rays_primary"], st["rays_shadow"], st["rays_reflect"]))

    def same(a, b):
        return all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4])) and a[4] == b[4]

    cams = [rt.Camera(origin=(0.0, -1.5, -6.0), pitch=8.0 / W, off_x=-4.0, off_y=-3.4),
            rt.Camera(origin=(1.0, -1.0, -6.0), pitch=8.0 / W, off_x=-3.0, off_y=-3.0)]
    with rt.Context(0) as off, rt.Context(0) as on:
        for ctx, value in ((off, 0), (on, 1)):
            ctx.set_option(rt.OPT_CLAIM_ORDER, value)
            to_product(ctx, atrium_spec(detail=0.05))
        with pytest.raises(rt.RtError):
            on.set_option(rt.OPT_CLAIM_ORDER, 3)
        ref = []
        for cam in cams:
            off.set_camera(cam)
            ref.append(frame(off))
        assert ref[0][4][1] > 0 and ref[0][4][2] > 0 and not same(ref[0], ref[1])
        on.set_camera(cams[0])
        assert same(frame(on), ref[0])          # the frame that made the order (counting pass + frame)
        assert same(frame(on), ref[0])          # the same frame in the recorded order
        assert same(frame(on), ref[0])          # ... replayed as a graph
        on.set_camera(cams[1])
        assert same(frame(on), ref[1])          # another view under the first view's order
        on.set_option(rt.OPT_CLAIM_ORDER, 2)    # a new order for this view
        assert same(frame(on), ref[1]) and same(frame(on), ref[1])
        on.set_option(rt.OPT_COUNT_TESTS, 1)
        assert same(frame(on), ref[1])
        assert on.stats()["box_tests"] > 0


def test_camera_rays_as_packets_equal_the_per_lane_kernel(monkeypatch):
    """k_trace_packet against k_trace for the camera rays (RT_WF_PACKET forces one of them when a context is created; left
    alone, the faster of the two is picked per layout): hit ids, nearest t, float colour, 8-bit colour and ray counts must
    be identical.  Frame widths that are no multiple of the 8-pixel tile, a camera whose rays leave through all eight
    octants (tiles that straddle an axis take the minimum / maximum form of the box test), a band set of rows, an
    irregular mesh, and a tree of single-triangle leaves."""
    import torch
    from common import soup_spec

    def frame(ctx, W, H, D, rs):
        n = rs.n_rows
        rgb8 = torch.zeros((n, W, 3), dtype=torch.uint8, device="cuda")
        hid = torch.zeros((n, W), dtype=torch.int32, device="cuda")
        rgb32 = torch.zeros((n, W, 3), dtype=torch.float32, device="cuda")
        ht = torch.zeros((n, W), dtype=torch.float32, device="cuda")
        ctx.render_rows_device(W, H, rs, D, rgb8.data_ptr(), hid.data_ptr(), rgb32.data_ptr(), ht.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = ctx.stats()
        return (rgb8.cpu().numpy(), hid.cpu().numpy(), rgb32.cpu().numpy().view(np.uint32), ht.cpu().numpy().view(np.uint32),
                (st["rays_primary"], st["rays_shadow"], st["rays_reflect"]))

    cases = [
        (atrium_spec(detail=0.05), 637, 403, 3, rt.Camera(origin=(0.0, -1.5, -6.0), pitch=8.0 / 637, off_x=-4.0, off_y=-3.4), None, 2),
        (atrium_spec(detail=0.05), 640, 400, 2, rt.Camera(origin=(0.0, 0.0, 2.0), pitch=8.0 / 640, off_x=-4.0, off_y=-2.5, plane_z=3.0), (16, 1, 3), 2),
        (soup_spec(n_tris=3000, seed=5), 322, 241, 2, rt.Camera(pitch=8.0 / 322), None, 1),
    ]
    for spec, W, H, D, cam, bands, leaf_max in cases:
        rs = rt.band_rowset(H, bands[0], bands[1], bands[2]) if bands else rt.RowSet(0, 1, 1, H)
        got = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("RT_WF_PACKET", mode)
            with rt.Context(0) as ctx:
                ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
                ctx.set_option(rt.OPT_LEAF_MAX, leaf_max)
                to_product(ctx, spec)
                ctx.set_camera(cam)
                got[mode] = [frame(ctx, W, H, D, rs) for _ in range(3)]     # plain order, the order-making frame, ordered
                assert ctx.stats()["accel"] == 1
        monkeypatch.delenv("RT_WF_PACKET")
        ref = got["0"][0]
        assert (ref[1] >= 0).mean() > 0.2
        for mode in ("0", "1"):
            for f in got[mode]:
                assert all(np.array_equal(x, y) for x, y in zip(f[:4], ref[:4])) and f[4] == ref[4], (mode, W, H)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RT_FUZZ_SEEDS", "32")))))
def test_random_scenes_through_every_staged_kernel_choice_equal_the_oracle(seed, monkeypatch):
    """Seeded random frames: triangle count, leaf size, depth, frame size (any width and height, not tile multiples), a
    camera anywhere in or around the mesh with its window anywhere (so rays leave through every octant and some run
    parallel to an axis), one to three lights.  The oracle's brute-force scan is the reference; the staged pipeline must
    match it with the camera rays on the per-lane kernel and on packets, on the first frame of the layout, the
    order-making frame and an ordered frame."""
    import torch
    from common import soup_spec, sphere

    r = np.random.default_rng(4000 + seed).random(16)        # (the frame is compared with an oracle frame made here, from the same draw)
    n_tris = int(2 + r[0] ** 3 * 6000)
    W, H = int(1 + r[1] * 150), int(1 + r[2] * 110)
    D = int(r[3] * 6)
    leaf_max = (1, 2, 4, 8)[int(r[4] * 4)]
    spec = soup_spec(n_tris=n_tris, seed=31 * seed + 5)
    if r[5] < 0.5:                                # a third light inside the mesh
        spec.analytic.append(sphere((float(r[6] * 6 - 3), float(r[7] * 4 - 2), float(r[8] * 6)), 0.05, (0.0, 1.0, 1.0, (0.5, 0.9, 0.6)), True))
    if seed % 4 == 3:                             # a single light
        spec.analytic = [a for k, a in enumerate(spec.analytic) if k != 2]
    origin = (float(r[9] * 10 - 5), float(r[10] * 6 - 3), float(r[11] * 10 - 6))
    span = 2.0 + r[12] * 10.0
    pitch = span / max(W, H)
    # a window that straddles the axis through the origin in x and y puts rays with a zero direction component on it
    cam = rt.Camera(origin=origin, pitch=pitch, off_x=origin[0] - pitch * int(W * r[13]), off_y=origin[1] - pitch * int(H * r[14]),
                    plane_z=origin[2] + (1.0 if r[15] < 0.7 else -1.0))
    want = to_oracle(spec).render(W, H, max_depth=D, cam=ocam(cam), threads=16)
    rs = rt.RowSet(0, 1, 1, H)

    def frame(ctx):
        rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        hid = torch.full((H, W), -7, dtype=torch.int32, device="cuda")
        rgb32 = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        ctx.render_rows_device(W, H, rs, D, rgb8.data_ptr(), hid.data_ptr(), rgb32.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = ctx.stats()
        return dict(rgb8=rgb8.cpu().numpy(), hit_id=hid.cpu().numpy(), rgb32=rgb32.cpu().numpy()), (st["rays_primary"], st["rays_shadow"], st["rays_reflect"])

    def orays(o):
        return (o["counts"]["primary"], o["counts"]["shadow"], o["counts"]["reflect"])

    def assert_parity(c):
        # ids identical; 8-bit colour within 1/255 everywhere and equal on >= 99.9 % of the pixels; float colour within powf's ulps
        assert c["id_mismatch"] == 0 and c["rgb8_over_tol"] == 0 and c["rgb8_exact"] >= 0.999 * c["pixels"] and c["rgb32_max_abs"] <= 4e-6, (seed, c)

    first = None
    for mode in ("0", "1"):
        monkeypatch.setenv("RT_WF_PACKET", mode)
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
            ctx.set_option(rt.OPT_PIPELINE, rt.PIPELINE_WAVEFRONT)
            ctx.set_option(rt.OPT_LEAF_MAX, leaf_max)
            to_product(ctx, spec)
            ctx.set_camera(cam)
            for k in range(3):
                g, counts = frame(ctx)
                if first is None:
                    first = g
                    assert_parity(compare(g, want))
                    assert counts == orays(want), (seed, counts, orays(want))
                else:
                    for key in ("rgb8", "hit_id", "rgb32"):
                        assert np.array_equal(g[key].view(np.uint8), first[key].view(np.uint8)), (seed, mode, k, key)
                    assert counts == orays(want)
            assert ctx.stats()["accel"] == 1
    monkeypatch.delenv("RT_WF_PACKET")


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RT_FUZZ_SEEDS", "32")) // 2)))
def test_random_frames_through_the_other_entry_points_equal_the_staged_frame(seed):
    """The same kind of seeded random frame through the ways a frame can be asked for besides the staged whole-frame call:
    the fused kernel (RT_OPT_PIPELINE = 1), the linear scan, the entry point that allocates nothing, a random band
    partition rendered part by part and put together on the device, the host-buffer call, and a second frame slot.
    Every plane and the ray counts equal the staged frame's bit for bit (the staged frame itself is checked against the
    oracle by the test above; here one seed in four is checked against it again)."""
    import torch
    from common import soup_spec

    r = np.random.default_rng(7000 + seed).random(12)
    n_tris = int(2 + r[0] ** 3 * 3000)
    W, H = int(8 + r[1] * 140), int(8 + r[2] * 100)
    D = int(r[3] * 5)
    spec = soup_spec(n_tris=n_tris, seed=17 * seed + 3)
    origin = (float(r[4] * 8 - 4), float(r[5] * 5 - 3), float(r[6] * 8 - 6))
    pitch = (2.0 + r[7] * 8.0) / max(W, H)
    cam = rt.Camera(origin=origin, pitch=pitch, off_x=origin[0] - pitch * int(W * r[8]), off_y=origin[1] - pitch * int(H * r[9]),
                    plane_z=origin[2] + 1.0)
    stream = torch.cuda.current_stream().cuda_stream

    def planes(n_rows):
        return (torch.zeros((n_rows, W, 3), dtype=torch.uint8, device="cuda"), torch.full((n_rows, W), -7, dtype=torch.int32, device="cuda"),
                torch.zeros((n_rows, W, 3), dtype=torch.float32, device="cuda"))

    def frame(ctx, call, rs, **kw):
        p = planes(rs.n_rows)
        call(W, H, rs, D, p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), 0, stream, **kw)
        torch.cuda.synchronize()
        st = ctx.stats()
        return p, (st["rays_primary"], st["rays_shadow"], st["rays_reflect"])

    def same(a, b, what):
        for x, y, name in zip(a, b, ("rgb8", "hit_id", "rgb32")):
            if not torch.equal(x.view(torch.uint8), y.view(torch.uint8)):
                bad = (x != y).reshape(x.shape[0], x.shape[1], -1).any(-1).nonzero()
                raise AssertionError("seed %d, %s: %s differs at %d pixels, first (row, column) %s: %s against %s" % (
                    seed, what, name, len(bad), bad[0].tolist(), x[bad[0][0], bad[0][1]].tolist(), y[bad[0][0], bad[0][1]].tolist()))

    whole = rt.RowSet(0, 1, 1, H)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
        ctx.set_option(rt.OPT_PIPELINE, rt.PIPELINE_WAVEFRONT)
        to_product(ctx, spec)
        ctx.set_camera(cam)
        ref, ref_rays = frame(ctx, ctx.render_rows_device, whole)
        if seed % 4 == 0:
            want = to_oracle(spec).render(W, H, max_depth=D, cam=ocam(cam), threads=16)
            c = compare(dict(rgb8=ref[0].cpu().numpy(), hit_id=ref[1].cpu().numpy(), rgb32=ref[2].cpu().numpy()), want)
            assert c["id_mismatch"] == 0 and c["rgb8_over_tol"] == 0 and c["rgb32_max_abs"] <= 4e-6, (seed, c)
        # a second frame slot of the same context
        got, rays = frame(ctx, ctx.render_rows_device, whole, slot=1)
        same(got, ref, "slot 1"); assert rays == ref_rays
        # the entry point that allocates nothing (its own fused kernels and static scratch)
        got, rays = frame(ctx, ctx.render_rows_device_noalloc, whole)
        same(got, ref, "no-alloc"); assert rays == ref_rays
        # a band partition, part by part, put together on the device
        band, parts = int(1 + r[10] * 12), int(2 + r[11] * 4)
        cap = rt.rows_per_part(H, band, parts)
        gathered = [torch.zeros((parts * cap,) + tuple(p.shape[1:]), dtype=p.dtype, device="cuda") for p in ref]
        total = [0, 0, 0]
        for part in range(parts):
            rs = rt.band_rowset(H, band, part, parts)
            if rs.n_rows == 0:
                continue
            got, rays = frame(ctx, ctx.render_rows_device, rs)
            for g, p in zip(gathered, got):
                g[part * cap: part * cap + rs.n_rows] = p
            total = [a + b for a, b in zip(total, rays)]
        out = planes(H)
        for g, o, bpp in zip(gathered, out, (3, 4, 12)):
            ctx.deinterleave_bands(g.data_ptr(), o.data_ptr(), W, H, band, parts, cap, bpp, stream)
        torch.cuda.synchronize()
        same(out, ref, "bands %d x %d" % (band, parts)); assert tuple(total) == ref_rays
        # the host-buffer call of the reference's seam
        h = ctx.render_frame(W, H, max_depth=D)
        assert np.array_equal(h["rgb8"], ref[0].cpu().numpy()) and np.array_equal(h["hit_id"], ref[1].cpu().numpy())
        assert np.array_equal(h["rgb32"].view(np.uint32), ref[2].cpu().numpy().view(np.uint32))
    for accel, pipeline, what in ((rt.ACCEL_LBVH, rt.PIPELINE_FUSED, "fused kernel"), (rt.ACCEL_LINEAR, rt.PIPELINE_AUTO, "linear scan")):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_ACCEL, accel)
            ctx.set_option(rt.OPT_PIPELINE, pipeline)
            to_product(ctx, spec)
            ctx.set_camera(cam)
            got, rays = frame(ctx, ctx.render_rows_device, whole)
            assert rays == ref_rays
            if accel == rt.ACCEL_LINEAR:
                # The scan tests every triangle against every ray, as the reference does.  A camera ray that grazes the floor
                # plane meets it 1e8 units out; from there the reference's triangle test is rounding noise (origin - vertex
                # keeps no digit of the vertex) and some triangle of the whole list "blocks" the light, while a hierarchy only
                # ever tests triangles near the ray.  The two then differ by the light's contribution at grazing incidence,
                # ~1e-7 in float colour: ids and 8-bit colour identical, float colour within the parity bar.
                assert torch.equal(got[1], ref[1]) and torch.equal(got[0], ref[0]), (seed, what)
                assert (got[2] - ref[2]).abs().max().item() <= 4e-6, (seed, what)
            else:
                same(got, ref, what)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RT_FUZZ_SEEDS", "32")) // 2)))
def test_random_scenes_with_random_spheres_planes_lights_and_builders_equal_the_oracle(seed):
    """Seeded random SCENES rather than random views of one kind of scene: zero to four spheres (some mirrors, some that
    contain the camera), zero to two planes at any angle, 1 .. 40 lights (so the 32-light chunks of the occlusion words and
    shadow queues are crossed), a triangle soup of 0 .. 1500 triangles with four materials, any of the three hierarchy
    builders, leaf size, reference cutting on or off, depth 0 .. 4, both camera-ray kernels left to the library's choice.
    The oracle's brute-force scan is the reference: ids and ray counts identical, colours within the parity bar, on the
    first frame of the layout and on the ordered one."""
    import torch
    from common import SceneSpec, soup_spec, sphere, plane

    g = np.random.default_rng(9000 + seed)
    r = g.random(16)
    n_lights = (1, 2, 3, 5, 31, 32, 33, 40)[int(r[0] * 8)]
    n_tris = int(r[1] ** 2 * 1500) if n_lights <= 5 else int(r[1] ** 2 * 300)       # (the oracle tests every light against every triangle)
    W, H = int(4 + r[2] * 90), int(4 + r[3] * 66)
    D = int(r[4] * 5)
    analytic = []
    for k in range(n_lights):
        p = g.random(3)
        analytic.append(sphere((float(p[0] * 12 - 6), float(p[1] * 6 + 1), float(p[2] * 12 - 4)), 0.1,
                               (0.0, 1.0, 1.0, tuple(float(c) for c in 0.3 + 0.7 * g.random(3) / max(1, n_lights // 2))), True))
    for k in range(int(r[5] * 5)):
        p = g.random(6)
        mirror = p[4] < 0.4
        analytic.append(sphere((float(p[0] * 8 - 4), float(p[1] * 6 - 3), float(p[2] * 8 - 1)), float(0.2 + p[3] ** 3 * 8),
                               (1.0 if mirror else float(p[5] * 0.5), 0.0 if mirror else 1.0, 1.0, (0.9, 0.8, 0.7))))
    for k in range(int(r[6] * 3)):
        p = g.random(5)
        n = np.array([p[0] - 0.5, p[1] + 0.2, p[2] - 0.5])
        n = n / np.linalg.norm(n)
        analytic.append(plane(tuple(float(c) for c in n), float(-2 - p[3] * 4), (float(p[4] * 0.6), 1.0, 1.0, (0.8, 0.8, 0.8))))
    order = g.permutation(len(analytic))               # ids follow the array order: lights anywhere in it
    analytic = [analytic[k] for k in order]
    base = soup_spec(n_tris=max(n_tris, 1), seed=29 * seed + 11)
    spec = SceneSpec(analytic, base.verts if n_tris else None, base.indices if n_tris else None, mats=base.mats if n_tris else None,
                     tri_material=base.tri_material if n_tris else None)
    origin = (float(r[7] * 6 - 3), float(r[8] * 4 - 2), float(r[9] * 6 - 6))
    pitch = (3.0 + r[10] * 8.0) / max(W, H)
    cam = rt.Camera(origin=origin, pitch=pitch, off_x=origin[0] - pitch * W * 0.5 + 0.013, off_y=origin[1] - pitch * H * 0.5 + 0.007,
                    plane_z=origin[2] + 1.0)
    want = to_oracle(spec).render(W, H, max_depth=D, cam=ocam(cam), threads=16)
    builder = (rt.BUILDER_SAH, rt.BUILDER_PLOC, rt.BUILDER_LBVH)[int(r[11] * 3)]
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
        ctx.set_option(rt.OPT_PIPELINE, rt.PIPELINE_WAVEFRONT if r[12] < 0.8 else rt.PIPELINE_FUSED)
        ctx.set_option(rt.OPT_BUILDER, builder)
        ctx.set_option(rt.OPT_LEAF_MAX, (1, 2, 4, 8)[int(r[13] * 4)])
        ctx.set_option(rt.OPT_SPLIT, 1 if r[14] < 0.3 else 0)
        to_product(ctx, spec)
        ctx.set_camera(cam)
        first = None
        for k in range(3):
            f = ctx.render_frame(W, H, max_depth=D)
            st = ctx.stats()
            assert (st["rays_primary"], st["rays_shadow"], st["rays_reflect"]) == (
                want["counts"]["primary"], want["counts"]["shadow"], want["counts"]["reflect"]), (seed, k)
            if first is None:
                first = f
                c = compare(f, want)
                assert c["id_mismatch"] == 0 and c["rgb8_over_tol"] == 0 and c["rgb8_exact"] >= 0.999 * c["pixels"] and c["rgb32_max_abs"] <= 4e-6, (seed, c)
            else:
                for key in ("rgb8", "hit_id", "rgb32"):
                    assert np.array_equal(f[key].view(np.uint8), first[key].view(np.uint8)), (seed, k, key)
        assert st["accel"] == (1 if n_tris >= 2 else 0)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RT_FUZZ_SEEDS", "32")) // 4)))
def test_random_vertex_updates_keep_the_tree_and_equal_the_oracle(seed):
    """A seeded random mesh is moved three times (rt_update_mesh_vertices: small jitter, a jitter as large as the mesh, a
    collapse of every vertex towards one point) while frames of the same layout keep coming (so the claim order and the
    camera-ray kernel chosen before the moves stay in use, stale): the refitted tree's frame equals the oracle's
    brute-force scan over the moved mesh each time - ids and ray counts identical, colours within the parity bar."""
    from common import SceneSpec, soup_spec

    g = np.random.default_rng(12000 + seed)
    r = g.random(8)
    spec = soup_spec(n_tris=int(2 + r[0] ** 2 * 1500), seed=41 * seed + 1)
    W, H, D = int(8 + r[1] * 100), int(8 + r[2] * 70), int(r[3] * 4)
    cam = rt.Camera(pitch=8.0 / max(W, H) * (0.5 + r[4]))
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
        ctx.set_option(rt.OPT_PIPELINE, rt.PIPELINE_WAVEFRONT)
        ctx.set_option(rt.OPT_BUILDER, (rt.BUILDER_SAH, rt.BUILDER_PLOC, rt.BUILDER_LBVH)[int(r[5] * 3)])
        to_product(ctx, spec)
        ctx.set_camera(cam)
        for _ in range(3):                     # plain frame, order-making frame, ordered frame
            ctx.render_frame(W, H, max_depth=D)
        nodes0 = ctx.stats()["bvh_nodes"]
        verts = spec.verts.astype(np.float32)
        for step, scale in enumerate((0.02, 3.0, None)):
            if scale is None:
                verts = (verts * 0.05 + np.array([0.3, -1.0, 2.0], np.float32)).astype(np.float32)
            else:
                verts = (verts + (g.random(verts.shape) - 0.5) * scale).astype(np.float32)
            ctx.update_mesh_vertices(verts)
            got = ctx.render_frame(W, H, max_depth=D)
            st = ctx.stats()
            assert st["accel"] == 1 and st["bvh_nodes"] == nodes0
            moved = SceneSpec(spec.analytic, verts, spec.indices, spec.mesh_mat, spec.mats, spec.tri_material)
            want = to_oracle(moved).render(W, H, max_depth=D, cam=ocam(cam), threads=16)
            c = compare(got, want)
            assert c["id_mismatch"] == 0 and c["rgb8_over_tol"] == 0 and c["rgb32_max_abs"] <= 4e-6, (seed, step, c)
            assert (st["rays_primary"], st["rays_shadow"], st["rays_reflect"]) == (
                want["counts"]["primary"], want["counts"]["shadow"], want["counts"]["reflect"]), (seed, step)


@pytest.mark.parametrize("seed", list(range(max(4, int(os.environ.get("RT_FUZZ_SEEDS", "32")) // 8))))
def test_random_mid_size_frames_equal_the_oracles_own_hierarchy(seed):
    """Frames large enough for many queue regions, thousands of packet tiles and deep trees (300..1000 pixels wide,
    5 000..125 000 small random triangles, depth 1..4), against the oracle with ITS hierarchy (an independent median-split
    BVH; the brute-force scan would take minutes): ids and ray counts identical, colours within the parity bar, on the
    plain frame, the order-making frame and the ordered frame."""
    from common import soup_spec

    r = np.random.default_rng(500 + seed).random(8)
    n_tris = int(5000 + r[0] * 120000)
    W, H, D = int(300 + r[1] * 700), int(200 + r[2] * 500), int(1 + r[3] * 4)
    spec = soup_spec(n_tris=n_tris, seed=seed + 77)
    tri = spec.verts.reshape(-1, 3, 3)
    centre = tri.mean(1, keepdims=True)
    spec.verts = ((tri - centre) * (0.05 + 0.3 * r[4]) + centre).reshape(-1, 3).astype(np.float32)   # small triangles: rays go deep
    cam = rt.Camera(origin=(float(r[5] * 2 - 1), float(r[6] * 2 - 1), -6.0), pitch=8.0 / W)
    want = to_oracle(spec, bvh=True).render(W, H, max_depth=D, cam=ocam(cam), use_bvh=True, threads=16)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.OPT_LEAF_MAX, (1, 2, 4)[seed % 3])
        to_product(ctx, spec)
        ctx.set_camera(cam)
        for k in range(3):
            f = ctx.render_frame(W, H, max_depth=D)
            st = ctx.stats()
            c = compare(f, want)
            assert c["id_mismatch"] == 0 and c["rgb8_over_tol"] == 0 and c["rgb32_max_abs"] <= 4e-6, (seed, k, c)
            assert (st["rays_primary"], st["rays_shadow"], st["rays_reflect"]) == (
                want["counts"]["primary"], want["counts"]["shadow"], want["counts"]["reflect"]), (seed, k)
        assert st["accel"] == 1


def test_the_shims_host_code_under_the_address_and_undefined_behaviour_sanitizers():
    """raytracer_amd/variants/shim_asan (raytracer_amd.build.build_shim_asan: tests/shim_asan_driver.cpp linked with the
    library's sources, host side sanitized): the C ABI's host paths - the hand-over into pageable and library buffers with
    its helper threads, row ranges, all planes, the claim order's counting sort, frame slots with a rebuild while frames
    are pending, band reassembly, the no-alloc entry point, statistics for an older header, vertex updates, refused
    calls, a second context - produce the expected bytes and no sanitizer report."""
    from raytracer_amd import build
    if not os.path.exists(build.SHIM_ASAN):
        pytest.skip("variants/shim_asan is not built (raytracer_amd.build.build_shim_asan)")
    env = dict(os.environ)
    # leaks at exit belong to the HIP runtime; another preloaded library may come before the sanitizer's runtime; the GPU
    # driver maps memory where the sanitizer keeps its shadow gap
    env["ASAN_OPTIONS"] = "detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([build.SHIM_ASAN], capture_output=True, text=True, timeout=300, env=env)
    if r.returncode != 0 and any(w in r.stderr for w in ("ReserveShadowMemoryRange", "Shadow memory range interleaves", "failed to mmap")):
        pytest.skip("the sanitizer's runtime could not set itself up in this process image: " + r.stderr[-300:])
    assert r.returncode == 0 and "shim driver ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_contexts_give_their_device_memory_back():
    """Twenty-five contexts made, used in every allocating way (scene, hierarchy, staged and fused workspaces, four frame
    slots with captured graphs, pinned staging, a library host buffer the caller forgets to free, a refit) and destroyed:
    the device's free memory after the last is what it was after the fifth."""
    import torch
    torch.zeros(1, device="cuda")

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    spec = atrium_spec(detail=0.05)
    W, H = 640, 400
    baseline = None
    for it in range(25):
        with rt.Context(0) as ctx:
            to_product(ctx, spec)
            ctx.set_camera(rt.Camera(origin=(0.0, -1.5, -6.0), pitch=8.0 / W, off_x=-4.0, off_y=-3.4))
            for _ in range(3):
                ctx.render_frame(W, H, max_depth=3)
            buf = ctx.alloc_host_buffer((H, W, 3))          # not freed: the context releases it
            ctx.render_frame_into(W, H, buf, max_depth=3)
            t = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
            for slot in range(4):
                for _ in range(3):
                    ctx.render_rows_device(W, H, rt.RowSet(0, 1, 1, H), 3, d_rgb8=t.data_ptr(), slot=slot)
            ctx.synchronize()
            ctx.set_option(rt.OPT_PIPELINE, rt.PIPELINE_FUSED)
            ctx.render_frame(W // 2, H // 2, max_depth=2)
            ctx.update_mesh_vertices(spec.verts * 1.01)
            ctx.render_frame(W // 2, H // 2, max_depth=2)
            del t
        if it == 4:
            baseline = free_bytes()
    assert baseline - free_bytes() < (8 << 20), (baseline, free_bytes())


def test_contexts_on_four_host_threads_at_once():
    """One host thread per context is the rule; contexts themselves are independent: four threads, each with a context of
    its own and a scene of its own, render frames at the same time (host-buffer frames, device frames, a rebuild, the
    claim order's calibration somewhere in between).  Every thread's frames equal what the same calls give alone."""
    import threading
    from common import soup_spec

    specs = [atrium_spec(detail=0.05), soup_spec(n_tris=900, seed=3), cube_spec(mirror=False, mesh_reflect=0.5), soup_spec(n_tris=5000, seed=8)]
    sizes = [(320, 200, 3), (257, 131, 2), (400, 300, 4), (199, 240, 1)]

    def work(k, out, rounds):
        try:
            W, H, D = sizes[k]
            with rt.Context(0) as ctx:
                ctx.set_option(rt.OPT_ACCEL, rt.ACCEL_LBVH)
                to_product(ctx, specs[k])
                ctx.set_camera(rt.Camera(pitch=8.0 / W))
                frames = []
                for r in range(rounds):
                    if r == rounds // 2:
                        ctx.set_option(rt.OPT_LEAF_MAX, 4)          # a rebuild in the middle
                    f = ctx.render_frame(W, H, max_depth=D)
                    st = ctx.stats()
                    frames.append((f["rgb8"].copy(), f["hit_id"].copy(), f["rgb32"].view(np.uint32).copy(),
                                   (st["rays_primary"], st["rays_shadow"], st["rays_reflect"])))
                out[k] = frames
        except Exception as e:          # (reported by the main thread)
            out[k] = e

    alone = {}
    for k in range(4):
        work(k, alone, 2)
        assert not isinstance(alone[k], Exception), alone[k]
    together = {}
    threads = [threading.Thread(target=work, args=(k, together, 12)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive()
    for k in range(4):
        assert not isinstance(together[k], Exception), together[k]
        ref = alone[k][0]
        for n, f in enumerate(together[k]):
            assert all(np.array_equal(a, b) for a, b in zip(f[:3], ref[:3])) and f[3] == ref[3], (k, n)
