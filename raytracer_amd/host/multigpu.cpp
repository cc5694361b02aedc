// multigpu.cpp — see multigpu.h.  RCCL is called directly (ncclCommInitAll: one process, one communicator rank
// per device); no collective besides the one all-gather of the 8-bit slices.
#include "multigpu.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>

struct MultiGpuFrame {
    std::vector<rt_ctx *> ctx;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<uint8_t *> slice, gathered;     // per device: its rendered bands; every device's bands, rank-major
    std::vector<rt_rowset> rows;
    uint8_t *frame = nullptr;                   // device 0: the frame in image order
    int W = 0, H = 0, band_rows = 16, rows_per_part = 0;
    size_t slice_bytes = 0;
};

namespace {

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return false;
}

bool nccl_ok(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return true;
    fprintf(stderr, "%s: %s\n", what, ncclGetErrorString(r));
    return false;
}

}  // namespace

MultiGpuFrame *multigpu_create(const std::vector<rt_ctx *> &contexts, const std::vector<int> &devices, int W, int H, int band_rows)
{
    const int n = (int)contexts.size();
    if (n < 1 || devices.size() != contexts.size() || W <= 0 || H <= 0 || band_rows < 1) return nullptr;
    MultiGpuFrame *f = new MultiGpuFrame;
    f->ctx = contexts; f->dev = devices; f->W = W; f->H = H; f->band_rows = band_rows;
    const int n_bands = (H + band_rows - 1) / band_rows;
    f->rows_per_part = ((n_bands + n - 1) / n) * band_rows;      // every part padded to the same row count: equal-sized gather
    f->slice_bytes = (size_t)3 * W * f->rows_per_part;
    f->comm.assign(n, nullptr); f->stream.assign(n, nullptr); f->slice.assign(n, nullptr); f->gathered.assign(n, nullptr);
    f->rows.resize(n);
    bool ok = nccl_ok(ncclCommInitAll(f->comm.data(), n, f->dev.data()), "ncclCommInitAll");
    for (int k = 0; k < n && ok; k++) {
        int mine = 0;                                             // rows of the bands k, k + n, k + 2n, ...
        for (int b = k; b < n_bands; b += n) mine += (b + 1) * band_rows <= H ? band_rows : H - b * band_rows;
        f->rows[k] = { k, band_rows, n, mine };
        ok = hip_ok(hipSetDevice(f->dev[k]), "hipSetDevice") && hip_ok(hipStreamCreateWithFlags(&f->stream[k], hipStreamNonBlocking), "hipStreamCreate") &&
             hip_ok(hipMalloc((void **)&f->slice[k], f->slice_bytes), "hipMalloc(slice)") &&
             hip_ok(hipMalloc((void **)&f->gathered[k], f->slice_bytes * n), "hipMalloc(gather)") &&
             hip_ok(hipMemsetAsync(f->slice[k], 0, f->slice_bytes, f->stream[k]), "hipMemset(slice)");
    }
    if (ok) ok = hip_ok(hipSetDevice(f->dev[0]), "hipSetDevice") && hip_ok(hipMalloc((void **)&f->frame, (size_t)3 * W * H), "hipMalloc(frame)");
    if (!ok) {
        multigpu_destroy(f);
        return nullptr;
    }
    return f;
}

bool multigpu_render(MultiGpuFrame *f, int max_depth, uint8_t *rgb8, uint64_t rays[3], float *device_ms)
{
    const int n = (int)f->ctx.size();
    // every device renders its bands on its own stream; the calls only enqueue
    for (int k = 0; k < n; k++) {
        if (f->rows[k].n_rows == 0) continue;
        if (rt_render_rows_device(f->ctx[k], f->W, f->H, &f->rows[k], max_depth, f->slice[k], nullptr, nullptr, nullptr, f->stream[k]) != RT_OK) {
            fprintf(stderr, "rt_render_rows_device on device %d: %s\n", f->dev[k], rt_last_error());
            return false;
        }
    }
    // one grouped all-gather of the slices, each rank on the stream its frame was rendered on
    if (!nccl_ok(ncclGroupStart(), "ncclGroupStart")) return false;
    for (int k = 0; k < n; k++)
        if (!nccl_ok(ncclAllGather(f->slice[k], f->gathered[k], f->slice_bytes, ncclUint8, f->comm[k], f->stream[k]), "ncclAllGather")) return false;
    if (!nccl_ok(ncclGroupEnd(), "ncclGroupEnd")) return false;
    // device 0: bands back into image order, then the frame to the host
    if (rt_deinterleave_bands(f->ctx[0], f->gathered[0], f->frame, f->W, f->H, f->band_rows, n, f->rows_per_part, 3, f->stream[0]) != RT_OK) {
        fprintf(stderr, "rt_deinterleave_bands: %s\n", rt_last_error());
        return false;
    }
    if (!hip_ok(hipSetDevice(f->dev[0]), "hipSetDevice") ||
        !hip_ok(hipMemcpyAsync(rgb8, f->frame, (size_t)3 * f->W * f->H, hipMemcpyDeviceToHost, f->stream[0]), "frame to host"))
        return false;
    for (int k = 0; k < n; k++)
        if (!hip_ok(hipSetDevice(f->dev[k]), "hipSetDevice") || !hip_ok(hipStreamSynchronize(f->stream[k]), "hipStreamSynchronize")) return false;
    rays[0] = rays[1] = rays[2] = 0;
    *device_ms = 0.f;
    for (int k = 0; k < n; k++) {
        if (f->rows[k].n_rows == 0) continue;
        rt_stats st;
        if (rt_get_stats(f->ctx[k], &st) != RT_OK) {
            fprintf(stderr, "rt_get_stats on device %d: %s\n", f->dev[k], rt_last_error());
            return false;
        }
        rays[0] += st.rays_primary; rays[1] += st.rays_shadow; rays[2] += st.rays_reflect;
        if (st.ms_frame > *device_ms) *device_ms = st.ms_frame;
    }
    return true;
}

void multigpu_destroy(MultiGpuFrame *f)
{
    if (!f) return;
    for (size_t k = 0; k < f->ctx.size(); k++) {
        (void)hipSetDevice(f->dev[k]);
        if (f->stream[k]) (void)hipStreamSynchronize(f->stream[k]);
        if (f->comm[k]) (void)ncclCommDestroy(f->comm[k]);
        if (f->slice[k]) (void)hipFree(f->slice[k]);
        if (f->gathered[k]) (void)hipFree(f->gathered[k]);
        if (f->stream[k]) (void)hipStreamDestroy(f->stream[k]);
    }
    if (f->frame) { (void)hipSetDevice(f->dev[0]); (void)hipFree(f->frame); }
    delete f;
}
