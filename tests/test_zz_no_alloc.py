"""rt_render_rows_device_noalloc against rt_render_rows_device through the C ABI: the entry point that
allocates nothing must produce the same bytes (ids, 8-bit and float colour, nearest t) and the same ray
counts, on two calls in a row that share the static scratch array, the second with a smaller input; and
must refuse, with RT_E_LIMIT, a frame its scratch cannot hold."""
import numpy as np
import pytest

from common import atrium_spec, cube_spec, rt, to_product

pytestmark = pytest.mark.gpu


def _both(ctx, W, H, D, rowset=None):
    import torch
    rs = rowset if rowset is not None else rt.RowSet(0, 1, 1, H)
    n = rs.n_rows
    out = {}
    for name, call in (("alloc", ctx.render_rows_device), ("noalloc", ctx.render_rows_device_noalloc)):
        rgb8 = torch.full((n, W, 3), 7, dtype=torch.uint8, device="cuda")
        hid = torch.full((n, W), -5, dtype=torch.int32, device="cuda")
        rgb32 = torch.full((n, W, 3), -1.0, dtype=torch.float32, device="cuda")
        ht = torch.full((n, W), -1.0, dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        call(W, H, rs, D, rgb8.data_ptr(), hid.data_ptr(), rgb32.data_ptr(), ht.data_ptr(), stream)
        torch.cuda.synchronize()
        st = ctx.stats()
        out[name] = dict(rgb8=rgb8.cpu().numpy(), hit_id=hid.cpu().numpy(), rgb32=rgb32.cpu().numpy(), hit_t=ht.cpu().numpy(),
                         rays=(st["rays_primary"], st["rays_shadow"], st["rays_reflect"]))
    return out["alloc"], out["noalloc"]


def _same(a, b):
    for k in ("rgb8", "hit_id", "rgb32", "hit_t"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    assert a["rays"] == b["rays"]


def test_noalloc_entry_point_equals_the_allocating_one_twice_in_a_row():
    with rt.Context(0) as ctx:
        # first call: an LBVH scene (the staged pipeline on the allocating side), most of the scratch in use
        to_product(ctx, atrium_spec(detail=0.05))
        ctx.set_camera(rt.Camera(origin=(0.0, -1.5, -6.0), pitch=8.0 / 640, off_x=-4.0, off_y=-3.4))
        a, b = _both(ctx, 640, 400, 4)
        assert ctx.stats()["accel"] == 1
        _same(a, b)
        assert (a["hit_id"] >= 0).mean() > 0.5 and a["rays"][2] > 0
        # second call, smaller input, same scratch: a band set of a smaller frame
        a, b = _both(ctx, 200, 150, 2, rt.band_rowset(150, 16, 1, 2))
        _same(a, b)
    with rt.Context(0) as ctx:
        # a linear-scan scene through another context on the same GPU (the scratch array is per GPU, not per context)
        to_product(ctx, cube_spec(mirror=False, mesh_reflect=0.5))
        ctx.set_camera(rt.Camera(pitch=8.0 / 320))
        a, b = _both(ctx, 320, 240, 3)
        assert ctx.stats()["accel"] == 0
        _same(a, b)


def test_noalloc_entry_point_refuses_what_its_scratch_cannot_hold():
    import torch
    with rt.Context(0) as ctx:
        to_product(ctx, cube_spec(mirror=False, mesh_reflect=0.5))
        rgb8 = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda")
        with pytest.raises(rt.RtError) as e:
            ctx.render_rows_device_noalloc(1920, 1080, rt.RowSet(0, 1, 1, 1080), 4, rgb8.data_ptr())
        assert e.value.code == -5 and "scratch" in str(e.value)
        # the same frame in row sets that fit: 8 slabs of 135 rows
        want = ctx.render_frame(1920, 1080, max_depth=4, want=("rgb8",))["rgb8"]
        for k in range(8):
            ctx.render_rows_device_noalloc(1920, 1080, rt.RowSet(135 * k, 1, 1, 135), 4, rgb8[135 * k:].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(rgb8.cpu().numpy(), want)
