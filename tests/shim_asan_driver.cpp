// tests/shim_asan_driver.cpp: a C++ caller of the C ABI (include/rt_shim.h) that is linked with the library's own sources,
// host side built with the address and undefined-behaviour sanitizers (the device code is built as always).  It walks the
// host-side paths that juggle buffers, vectors and threads - the hand-over of a frame into pageable and library buffers
// with its helper threads, row ranges, all planes, the claim order's counting sort, frame slots, scene changes with
// frames pending, vertex updates, band reassembly, statistics for an older header - and checks every result against the
// first frame.  Any sanitizer report ends the process with a non-zero code.
#include "rt_shim.h"

#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

static unsigned long long g_s = 0x9e3779b97f4a7c15ull;
static float frand() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (float)((g_s >> 11) & 0xffffff) / 16777216.0f; }

#define CHECK(call)                                                                                 \
    do {                                                                                            \
        int rc__ = (call);                                                                          \
        if (rc__ != RT_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, rt_last_error()); return 1; } \
    } while (0)
#define EXPECT(cond)                                                                                \
    do {                                                                                            \
        if (!(cond)) { fprintf(stderr, "line %d: %s is false\n", __LINE__, #cond); return 1; }      \
    } while (0)

int main()
{
    EXPECT(rt_abi_version() == RT_ABI_VERSION);
    rt_ctx *c = rt_create(0);
    if (!c) { fprintf(stderr, "rt_create: %s\n", rt_last_error()); return 1; }
    // two lights, a floor, a mirror sphere; a few thousand random triangles
    std::vector<rt_analytic> prims(4);
    memset(prims.data(), 0, sizeof(rt_analytic) * prims.size());
    const rt_material white = { 0.0f, 1.0f, 1.0f, { 1.0f, 1.0f, 1.0f } }, mirror = { 1.0f, 0.0f, 1.0f, { 1.0f, 1.0f, 1.0f } };
    prims[0] = { RT_SPHERE, 1, white, { 2.0f, 5.0f, 1.0f }, 0.1f };
    prims[1] = { RT_PLANE, 0, white, { 0.0f, 1.0f, 0.0f }, -4.0f };
    prims[2] = { RT_SPHERE, 1, white, { -2.0f, 5.0f, -3.0f }, 0.1f };
    prims[3] = { RT_SPHERE, 0, mirror, { 0.0f, -2.0f, 2.0f }, 1.5f };
    CHECK(rt_set_analytic(c, prims.data(), (int)prims.size()));
    const int n_tris = 6000;
    std::vector<float> verts((size_t)n_tris * 9);
    std::vector<int32_t> idx((size_t)n_tris * 3);
    for (int t = 0; t < n_tris; t++) {
        const float cx = frand() * 8 - 4, cy = frand() * 6 - 4, cz = frand() * 8 - 1;
        for (int k = 0; k < 3; k++) {
            verts[(size_t)t * 9 + k * 3 + 0] = cx + (frand() - 0.5f) * 0.8f;
            verts[(size_t)t * 9 + k * 3 + 1] = cy + (frand() - 0.5f) * 0.8f;
            verts[(size_t)t * 9 + k * 3 + 2] = cz + (frand() - 0.5f) * 0.8f;
            idx[(size_t)t * 3 + k] = t * 3 + k;
        }
    }
    CHECK(rt_set_option(c, RT_OPT_ACCEL, 2));
    CHECK(rt_set_mesh(c, verts.data(), n_tris * 3, idx.data(), n_tris * 3, &white));
    std::vector<rt_material> mats = { white, mirror, { 0.5f, 0.8f, 0.3f, { 0.9f, 0.4f, 0.2f } } };
    std::vector<int32_t> tm(n_tris);
    for (int t = 0; t < n_tris; t++) tm[t] = t % 3;
    CHECK(rt_set_triangle_materials(c, mats.data(), (int)mats.size(), tm.data(), n_tris));
    const int W = 333, H = 207, D = 3;
    const rt_camera cam = { { 0.0f, 0.0f, -5.0f }, 8.0f / W, -4.0f, -2.5f, 0.0f };
    CHECK(rt_set_camera(c, &cam));

    const size_t px = (size_t)W * H;
    std::vector<uint8_t> ref8(px * 3), got8(px * 3);
    std::vector<int32_t> refid(px), gotid(px);
    std::vector<float> ref32(px * 3), got32(px * 3);
    // ---- whole frames into pageable memory: plain, the order-making frame (counting sort on the host), ordered, replayed
    CHECK(rt_render_frame(c, W, H, 0, H, D, ref8.data(), refid.data(), ref32.data()));
    rt_stats st0;
    CHECK(rt_get_stats(c, &st0));
    EXPECT(st0.accel == 1 && st0.rays_shadow > 0 && st0.rays_reflect > 0);
    for (int k = 0; k < 4; k++) {
        CHECK(rt_render_frame(c, W, H, 0, H, D, got8.data(), gotid.data(), got32.data()));
        EXPECT(got8 == ref8 && gotid == refid && memcmp(got32.data(), ref32.data(), px * 12) == 0);
    }
    // only some planes; a row range into the middle of a buffer; rows at the very end of a buffer
    std::fill(got8.begin(), got8.end(), 0);
    CHECK(rt_render_frame(c, W, H, 0, H, D, got8.data(), nullptr, nullptr));
    EXPECT(got8 == ref8);
    {
        const int y0 = 50, y1 = 121;
        std::vector<uint8_t> part((size_t)(y1 - y0) * W * 3);
        std::vector<int32_t> partid((size_t)(y1 - y0) * W);
        CHECK(rt_render_frame(c, W, H, y0, y1, D, part.data(), partid.data(), nullptr));
        EXPECT(memcmp(part.data(), ref8.data() + (size_t)y0 * W * 3, part.size()) == 0);
        EXPECT(memcmp(partid.data(), refid.data() + (size_t)y0 * W, partid.size() * 4) == 0);
    }
    // ---- all six planes
    {
        std::vector<float> t(px), n(px * 3), uv(px * 2);
        rt_frame_planes pl = { got8.data(), gotid.data(), got32.data(), t.data(), n.data(), uv.data() };
        CHECK(rt_render_frame_planes(c, W, H, 0, H, D, &pl));
        EXPECT(got8 == ref8 && gotid == refid);
    }
    // ---- a library buffer: whole frame, then a row range inside it; a large frame through the pieces and helper threads
    {
        uint8_t *pin = (uint8_t *)rt_alloc_host_buffer(c, px * 3 + 64);
        EXPECT(pin != nullptr);
        CHECK(rt_render_frame(c, W, H, 0, H, D, pin + 64, nullptr, nullptr));
        EXPECT(memcmp(pin + 64, ref8.data(), px * 3) == 0);
        CHECK(rt_render_frame(c, W, H, 10, 20, D, pin, nullptr, nullptr));
        EXPECT(memcmp(pin, ref8.data() + (size_t)10 * W * 3, (size_t)10 * W * 3) == 0);
        CHECK(rt_free_host_buffer(c, pin));
        EXPECT(rt_free_host_buffer(c, pin) != RT_OK);
        const int BW = 1600, BH = 900;
        std::vector<uint8_t> a((size_t)BW * BH * 3), b((size_t)BW * BH * 3);
        std::vector<float> f32((size_t)BW * BH * 3);
        CHECK(rt_render_frame(c, BW, BH, 0, BH, 2, a.data(), nullptr, f32.data()));
        CHECK(rt_render_frame(c, BW, BH, 0, BH, 2, b.data(), nullptr, nullptr));
        EXPECT(a == b);
    }
    // ---- device buffers: two frame slots in flight on two streams, a band partition reassembled on the device, no-alloc entry
    {
        hipStream_t s0, s1;
        EXPECT(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess);
        uint8_t *d0, *d1, *dg, *df;
        const int band = 7, parts = 3;
        const int n_bands = (H + band - 1) / band, cap = ((n_bands + parts - 1) / parts) * band;
        EXPECT(hipMalloc((void **)&d0, px * 3) == hipSuccess && hipMalloc((void **)&d1, px * 3) == hipSuccess);
        EXPECT(hipMalloc((void **)&dg, (size_t)parts * cap * W * 3) == hipSuccess && hipMalloc((void **)&df, px * 3) == hipSuccess);
        const rt_rowset whole = { 0, 1, 1, H };
        for (int k = 0; k < 6; k++) {
            CHECK(rt_render_rows_device_slot(c, 0, W, H, &whole, D, d0, nullptr, nullptr, nullptr, s0));
            CHECK(rt_render_rows_device_slot(c, 1, W, H, &whole, D, d1, nullptr, nullptr, nullptr, s1));
        }
        // a hierarchy option changed while both slots have frames pending (the rebuild waits for them)
        CHECK(rt_set_option(c, RT_OPT_LEAF_MAX, 4));
        CHECK(rt_render_rows_device_slot(c, 1, W, H, &whole, D, d1, nullptr, nullptr, nullptr, s1));
        CHECK(rt_synchronize(c));
        EXPECT(hipMemcpy(got8.data(), d0, px * 3, hipMemcpyDeviceToHost) == hipSuccess && got8 == ref8);
        EXPECT(hipMemcpy(got8.data(), d1, px * 3, hipMemcpyDeviceToHost) == hipSuccess && got8 == ref8);
        EXPECT(hipMemset(dg, 0, (size_t)parts * cap * W * 3) == hipSuccess);
        for (int p = 0; p < parts; p++) {
            int mine = 0;
            for (int bnd = p; bnd < n_bands; bnd += parts) mine += (bnd + 1) * band <= H ? band : H - bnd * band;
            const rt_rowset rs = { p, band, parts, mine };
            CHECK(rt_render_rows_device(c, W, H, &rs, D, dg + (size_t)p * cap * W * 3, nullptr, nullptr, nullptr, s0));
        }
        CHECK(rt_deinterleave_bands(c, dg, df, W, H, band, parts, cap, 3, s0));
        CHECK(rt_synchronize(c));
        EXPECT(hipStreamSynchronize(s0) == hipSuccess);
        EXPECT(hipMemcpy(got8.data(), df, px * 3, hipMemcpyDeviceToHost) == hipSuccess && got8 == ref8);
        CHECK(rt_render_rows_device_noalloc(c, W, H, &whole, D, d0, nullptr, nullptr, nullptr, s0));
        EXPECT(hipStreamSynchronize(s0) == hipSuccess);
        EXPECT(hipMemcpy(got8.data(), d0, px * 3, hipMemcpyDeviceToHost) == hipSuccess && got8 == ref8);
        (void)hipFree(d0); (void)hipFree(d1); (void)hipFree(dg); (void)hipFree(df);
        (void)hipStreamDestroy(s0); (void)hipStreamDestroy(s1);
    }
    // ---- statistics for a caller with the first version of the struct; test counting; bad arguments
    {
        unsigned char old[104 + 16];
        memset(old, 0xab, sizeof old);
        CHECK(rt_get_stats_sized(c, (rt_stats *)old, 104));
        EXPECT(old[104] == 0xab && old[119] == 0xab);
        EXPECT(rt_get_stats_sized(c, (rt_stats *)old, 0) != RT_OK && rt_get_stats_sized(c, (rt_stats *)old, 3) != RT_OK);
        CHECK(rt_set_option(c, RT_OPT_COUNT_TESTS, 1));
        CHECK(rt_render_frame(c, W, H, 0, H, D, got8.data(), nullptr, nullptr));
        rt_stats st;
        CHECK(rt_get_stats(c, &st));
        EXPECT(got8 == ref8 && st.box_tests > 0 && st.box_tests_occupied <= st.box_tests && st.tri_tests > 0);
        CHECK(rt_set_option(c, RT_OPT_COUNT_TESTS, 0));
        EXPECT(rt_render_frame(c, W, H, 5, 3, D, got8.data(), nullptr, nullptr) != RT_OK);
        EXPECT(rt_render_frame(c, 0, H, 0, H, D, got8.data(), nullptr, nullptr) != RT_OK);
        EXPECT(rt_set_option(c, RT_OPT_CLAIM_ORDER, 7) != RT_OK);
    }
    // ---- moved vertices (refit), then a smaller mesh, then no mesh at all
    {
        for (size_t k = 0; k < verts.size(); k++) verts[k] += (frand() - 0.5f) * 0.05f;
        CHECK(rt_update_mesh_vertices(c, verts.data(), n_tris * 3));
        CHECK(rt_render_frame(c, W, H, 0, H, D, got8.data(), gotid.data(), nullptr));
        EXPECT(rt_update_mesh_vertices(c, verts.data(), n_tris * 3 - 3) != RT_OK);
        CHECK(rt_set_triangles(c, verts.data(), 40, &mirror));
        CHECK(rt_render_frame(c, W, H, 0, H, D, got8.data(), gotid.data(), nullptr));
        CHECK(rt_render_frame(c, W, H, 0, H, D, got8.data(), gotid.data(), nullptr));
        // a mesh with an index outside its vertices is refused and leaves no mesh behind
        const int32_t bad_idx[3] = { 0, 1, n_tris * 3 };
        EXPECT(rt_set_mesh(c, verts.data(), n_tris * 3, bad_idx, 3, &white) == RT_E_INVALID);
        EXPECT(rt_update_mesh_vertices(c, verts.data(), n_tris * 3) == RT_E_STATE);
        CHECK(rt_set_triangles(c, verts.data(), 0, &white));
        CHECK(rt_render_frame(c, W, H, 0, H, D, got8.data(), gotid.data(), nullptr));
        rt_stats st;
        CHECK(rt_get_stats(c, &st));
        EXPECT(st.n_triangles == 0 && st.accel == 0);
    }
    rt_destroy(c);
    // a second context after the first is gone (per-device state of the library survives a context)
    c = rt_create(0);
    EXPECT(c != nullptr);
    CHECK(rt_set_analytic(c, prims.data(), (int)prims.size()));
    CHECK(rt_render_frame(c, 64, 48, 0, 48, 1, got8.data(), nullptr, nullptr));
    rt_destroy(c);
    printf("shim driver ok\n");
    return 0;
}
