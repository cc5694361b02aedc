// rt_shim.hip — the C ABI declared in include/rt_shim.h: context, scene upload,
// workspace management and the render-frame entry points.  No CPU fallback lives
// here: every entry point either runs the HIP kernels or returns an error.
#include "rt_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <new>
#include <thread>

#include <mutex>

// Scratch of rt_render_rows_device_noalloc: one copy per GPU that loads this library, held until the process
// ends.  In host code the array's NAME is only a handle for hipGetSymbolAddress: the device address of a GPU's
// copy is looked up once per device (scratch_for_device) and that pointer is what kernels and memsets receive.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int RT_MAX_DEVICES = 64;
unsigned char *g_scratch_ptr[RT_MAX_DEVICES];     // [device ordinal] -> address of that device's `scratch`
std::mutex g_scratch_mutex;

// With `device` current: the address of this device's copy of the scratch array (looked up on first use).
hipError_t scratch_for_device(int device, unsigned char **out)
{
    *out = nullptr;
    if (device < 0 || device >= RT_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = (unsigned char *)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

}  // namespace

// What one frame in flight owns.
struct FrameSlot {
    FrameWorkspace ws;
    // replay of the last frame's launches (RT_OPT_GRAPH)
    hipGraphExec_t graph_exec = nullptr;
    std::vector<unsigned char> graph_key;
    std::vector<unsigned char> last_key;     // the previous frame's key, captured or not
    // last frame
    bool frame_valid = false;
    int last_levels = 0, last_launches = 0, last_accel = 0;
    bool last_wavefront = false;
    bool last_timed = false;                 // the last frame recorded its device events
    uint64_t last_primary = 0;
    FrameCounters *last_counters = nullptr;  // where the last frame counted its rays (the workspace, or the scratch array)
    float last_staging_ms = 0.f;             // host time of the last frame's device-to-host hand-over (host-buffer entry points)
    // Renders are asynchronous and may run on a caller's stream: frame_done is recorded behind every enqueued
    // frame, and everything that rewrites what a frame reads (scene, hierarchy, workspace, the captured graph)
    // waits for it first.
    hipEvent_t frame_done = nullptr;
    bool frame_pending = false;
    hipStream_t last_stream = nullptr;
};

constexpr int HOST_PIECES = 8;     // most pieces a plane of a host hand-over is cut into (piece k+1 flies while piece k is copied on)
constexpr int HOST_PIECES_DEFAULT = 4;   // measured on a 6.2 MB plane: 1 piece 0.29 ms after the frame, 2: 0.25, 3-4: 0.23, 6: 0.26, 8: 0.28 (each piece costs ~15 us)
constexpr int HOST_PLANES_MAX = 6; // planes of rt_frame_planes

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // scene
    DevAnalytic *d_analytic = nullptr;
    int32_t *d_lights = nullptr;
    int n_analytic = 0, n_lights = 0;
    float4 *d_tri_mt = nullptr, *d_tri_n = nullptr;
    int n_tris = 0;
    int32_t *d_indices = nullptr;   // kept from rt_set_mesh for rt_update_mesh_vertices
    float4 *d_tri_vn = nullptr;     // opt-in shading normals, 3 per triangle (rt_set_triangle_normals)
    float2 *d_tri_vt = nullptr;     // texture coordinates, 3 per triangle (rt_set_triangle_texcoords)
    void *d_stage = nullptr;        // upload staging of rt_update_*: kept between calls (one per tick in an animation)
    size_t stage_cap = 0;
    int n_mesh_verts = 0;
    DevMaterial *d_mats = nullptr;
    int n_mats = 0;
    LbvhResult bvh;
    // options
    int opt_accel = 0, opt_count = 0, opt_leaf_max = 2, opt_timing = 1, opt_pipeline = 0, opt_builder = 0, opt_split = 0;
    DevCamera cam;
    int opt_graph = 1;                       // replay of an unchanged frame's launches (only with timing and test counting off)
    int opt_claim_order = getenv("RT_CLAIM_ORDER") ? atoi(getenv("RT_CLAIM_ORDER")) : 1;   // RT_OPT_CLAIM_ORDER (the variable sets the default, for A/B runs)
    unsigned long long scene_version = 0;    // bumped when primitives are replaced (not when they move)
    int opt_camera_packet = getenv("RT_WF_PACKET") ? atoi(getenv("RT_WF_PACKET")) : -1;   // -1: whichever was faster when the claim order was made; 0 / 1 force
    unsigned long long state_version = 0;   // bumped by anything that changes device pointers or the scene
    unsigned char *scratch_ptr = nullptr;    // this device's copy of the no-alloc scratch array (set by rt_create)
    // Host hand-over of rt_render_frame / rt_render_frame_planes: a pinned staging buffer the device copies into at
    // full PCIe rate, in pieces, each copied on into the caller's (pageable) buffer while the next piece is in flight.
    // Buffers from rt_alloc_host_buffer are pinned themselves and written by the device directly.
    unsigned char *h_stage = nullptr;
    size_t h_stage_cap = 0;
    hipEvent_t h_events[HOST_PLANES_MAX * HOST_PIECES] = {};
    struct PinnedRange { unsigned char *p; size_t bytes; };
    std::vector<PinnedRange> pinned;
    // Frames in flight: every slot has a workspace, a captured graph and a completion event of its own, so frames of
    // different slots may run side by side on different streams; f is the slot of the frame at hand (and afterwards
    // of the last one, which rt_get_stats reports).
    FrameSlot slots[RT_MAX_FRAME_SLOTS];
    FrameSlot *f = &slots[0];
};

namespace {

void default_camera(DevCamera *c)   // main.cpp:267-269
{
    c->ox = 0.0f; c->oy = 0.0f; c->oz = -5.0f;
    c->pitch = 0.01f; c->off_x = -4.0f; c->off_y = -4.0f; c->plane_z = 0.0f;
}

DevMaterial to_dev(const rt_material *m)
{
    DevMaterial d;
    memset(&d, 0, sizeof d);
    if (m) {
        d.reflect = m->reflect; d.diffuse = m->diffuse; d.spec = m->spec;
        d.cr = m->color[0]; d.cg = m->color[1]; d.cb = m->color[2];
    } else {   // Material(), main.cpp:34
        d.reflect = 0.0f; d.diffuse = 1.0f; d.spec = 1.0f;
        d.cr = d.cg = d.cb = 1.0f;
    }
    return d;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Wait for everything this context has enqueued, on its own stream and on any caller stream a frame went to.
void quiesce(rt_ctx *c)
{
    (void)hipStreamSynchronize(c->stream);
    for (FrameSlot &f : c->slots) {
        if (f.frame_pending && f.frame_done) (void)hipEventSynchronize(f.frame_done);
        f.frame_pending = false;
    }
}

void drop_graph(FrameSlot &f)
{
    if (f.graph_exec && f.frame_pending && f.frame_done) (void)hipEventSynchronize(f.frame_done);   // it may still be replaying
    if (f.graph_exec) (void)hipGraphExecDestroy(f.graph_exec);
    f.graph_exec = nullptr;
    f.graph_key.clear();
}

void drop_graphs(rt_ctx *c)
{
    for (FrameSlot &f : c->slots) drop_graph(f);
}


void free_triangles(rt_ctx *c)
{
    c->state_version++;
    c->scene_version++;
    if (c->d_tri_mt) (void)hipFree(c->d_tri_mt);
    if (c->d_tri_n) (void)hipFree(c->d_tri_n);
    if (c->d_mats) (void)hipFree(c->d_mats);
    if (c->d_indices) (void)hipFree(c->d_indices);
    if (c->d_tri_vn) (void)hipFree(c->d_tri_vn);
    c->d_tri_vn = nullptr;
    if (c->d_tri_vt) (void)hipFree(c->d_tri_vt);
    c->d_tri_vt = nullptr;
    c->d_tri_mt = c->d_tri_n = nullptr;
    c->d_mats = nullptr;
    c->d_indices = nullptr;
    c->n_mesh_verts = 0;
    c->n_tris = 0;
    c->n_mats = 0;
    lbvh_free(&c->bvh);
}

void free_workspace(FrameWorkspace &ws)
{
    void *ptrs[] = { ws.records, ws.hit_t, ws.queue_o[0], ws.queue_o[1], ws.queue_d[0], ws.queue_d[1], ws.counters,
                     ws.out_rgb8, ws.out_id, ws.out_rgb32, ws.out_normal, ws.out_uv, ws.sq_o, ws.rq_counts[0], ws.rq_counts[1],
                     ws.sq_counts, ws.overflow, ws.p_hits, ws.p_occ, ws.fq_counts[0], ws.fq_counts[1], ws.unit_cost, ws.unit_order };
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : ws.events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ws.trav_events) (void)hipEventDestroy(e);
    ws = FrameWorkspace();
}

int ensure_workspace(rt_ctx *c, size_t pixels, size_t slots, int levels, int pipe, bool host_outputs)
{
    const bool wavefront = pipe == 2;
    FrameWorkspace &ws = c->f->ws;
    // (hipFree of a buffer waits for the device by itself; the frees below are the only way a buffer goes away)
    // frame counters + claim cursors: one block, cleared by one memset per frame
    {
        const int chunks = (c->n_lights + 31) / 32;
        const size_t need = COUNTERS_PAD + (wavefront ? sizeof(unsigned int) * wavefront_cursor_words(levels, chunks) : 0);
        if (need > ws.zero_cap) {
            if (ws.counters) (void)hipFree(ws.counters);
            ws.counters = nullptr; ws.trace_cursors = nullptr; ws.zero_cap = 0;
            if (hipMalloc((void **)&ws.counters, need) != hipSuccess) return fail(RT_E_NOMEM, "hipMalloc(counters and claim cursors) failed");
            ws.zero_cap = need;
            ws.trace_cursors = (unsigned int *)((char *)ws.counters + COUNTERS_PAD);
        }
    }
    // ray queues are indexed by slot (the frame padded to whole 8x8 tiles, rounded up to whole regions)
    const size_t qslots = wavefront_regions(slots > pixels ? slots : pixels) * 256;
    if (qslots > ws.pixels_cap) {
        void *old[] = { ws.hit_t, ws.queue_o[0], ws.queue_o[1], ws.queue_d[0], ws.queue_d[1], ws.fq_counts[0], ws.fq_counts[1] };
        for (void *p : old)
            if (p) (void)hipFree(p);
        ws.hit_t = nullptr; ws.fq_counts[0] = ws.fq_counts[1] = nullptr;
        ws.queue_o[0] = ws.queue_o[1] = ws.queue_d[0] = ws.queue_d[1] = nullptr;
        ws.pixels_cap = 0;
        const size_t pc = qslots;
        hipError_t e = hipMalloc((void **)&ws.hit_t, sizeof(float) * pc);
        for (int k = 0; k < 2 && e == hipSuccess; k++) {
            e = hipMalloc((void **)&ws.queue_o[k], sizeof(float4) * pc);
            if (e == hipSuccess) e = hipMalloc((void **)&ws.queue_d[k], sizeof(float4) * pc);
            if (e == hipSuccess) e = hipMalloc((void **)&ws.fq_counts[k], sizeof(unsigned int) * (pc / 64 + 1));
        }
        if (e != hipSuccess) return fail(RT_E_NOMEM, std::string("frame workspace: ") + hipGetErrorString(e));
        ws.pixels_cap = pc;
    }
    if (levels > ws.levels_cap) ws.levels_cap = levels;
    // The per-level colour records (16 bytes per pixel and level) are only read by the fused pipeline and by staged frames
    // too deep for the shading kernel's registers: a 7680x4320 frame of 9 levels would otherwise hold 4.8 GB for nothing.
    const bool need_records = pipe == 1 || levels > SHADE_REG_LEVELS;
    const size_t rec = need_records ? pixels * (size_t)levels : 0;
    if (rec > ws.records_cap) {
        if (ws.records) (void)hipFree(ws.records);
        ws.records = nullptr; ws.records_cap = 0;
        if (hipMalloc((void **)&ws.records, sizeof(float4) * rec) != hipSuccess) return fail(RT_E_NOMEM, "hipMalloc(level records) failed");
        ws.records_cap = rec;
    }
    if (wavefront) {
        // slot space: the frame padded to whole 8x8 tiles (>= pixels, also bounds every ray queue)
        const int n_chunks = (c->n_lights + 31) / 32, segs = c->n_lights < 32 ? c->n_lights : 32;
        std::string err;
        const int cus = device_num_cus(&err);
        if (cus <= 0) return fail(RT_E_DEVICE, err);
        if (qslots > ws.slots_cap || n_chunks > ws.chunks_cap || segs > ws.seg_cap) {
            void *old[] = { ws.sq_o, ws.rq_counts[0], ws.rq_counts[1], ws.sq_counts };
            for (void *p : old)
                if (p) (void)hipFree(p);
            ws.sq_o = nullptr;
            ws.rq_counts[0] = ws.rq_counts[1] = ws.sq_counts = nullptr;
            const size_t sc = qslots > ws.slots_cap ? qslots : ws.slots_cap;
            const int cc = n_chunks > ws.chunks_cap ? n_chunks : ws.chunks_cap, gc = segs > ws.seg_cap ? segs : ws.seg_cap;
            const size_t regions = wavefront_regions(sc);
            ws.slots_cap = 0; ws.chunks_cap = 0; ws.seg_cap = 0;
            hipError_t e = hipMalloc((void **)&ws.sq_o, sizeof(float4) * sc * (size_t)(gc ? gc : 1));
            if (e == hipSuccess) e = hipMalloc((void **)&ws.rq_counts[0], sizeof(unsigned int) * regions);
            if (e == hipSuccess) e = hipMalloc((void **)&ws.rq_counts[1], sizeof(unsigned int) * regions);
            if (e == hipSuccess) e = hipMalloc((void **)&ws.sq_counts, sizeof(unsigned int) * regions * (size_t)(gc ? gc : 1));
            if (e != hipSuccess) return fail(RT_E_NOMEM, std::string("wavefront workspace: ") + hipGetErrorString(e));
            ws.slots_cap = sc; ws.chunks_cap = cc; ws.seg_cap = gc;
        }
        // claim-order arrays: one row of units per traversal launch
        const size_t stride = wavefront_unit_stride(slots > pixels ? slots : pixels, c->n_lights);
        const int n_launch = wavefront_unit_launches(levels, n_chunks);
        if (stride * (size_t)n_launch > ws.unit_stride * (size_t)ws.unit_launches || stride != ws.unit_stride) {
            if (ws.unit_cost) (void)hipFree(ws.unit_cost);
            if (ws.unit_order) (void)hipFree(ws.unit_order);
            ws.unit_cost = ws.unit_order = nullptr; ws.unit_stride = 0; ws.unit_launches = 0; ws.order_valid = false;
            const size_t n = stride * (size_t)n_launch;
            if (hipMalloc((void **)&ws.unit_cost, sizeof(unsigned int) * n) != hipSuccess ||
                hipMalloc((void **)&ws.unit_order, sizeof(unsigned int) * n) != hipSuccess)
                return fail(RT_E_NOMEM, "hipMalloc(claim order) failed");
            ws.unit_stride = stride; ws.unit_launches = n_launch;
        }
        const size_t ovf = wavefront_overflow_ints(cus);
        if (ovf > ws.overflow_cap) {
            if (ws.overflow) (void)hipFree(ws.overflow);
            ws.overflow = nullptr; ws.overflow_cap = 0;
            if (hipMalloc((void **)&ws.overflow, sizeof(int) * ovf) != hipSuccess)
                return fail(RT_E_NOMEM, "hipMalloc(stack overflow columns) failed");
            ws.overflow_cap = ovf;
        }
    }
    if (pipe == 2) {
        // [level][pixel] hits and [level][chunk][pixel] occlusion words
        const size_t n_chunks = (size_t)((c->n_lights + 31) / 32 > 0 ? (c->n_lights + 31) / 32 : 1);
        const size_t need = pixels * (size_t)levels * n_chunks;
        if (need > ws.p_cap) {
            if (ws.p_hits) (void)hipFree(ws.p_hits);
            if (ws.p_occ) (void)hipFree(ws.p_occ);
            ws.p_hits = nullptr; ws.p_occ = nullptr; ws.p_cap = 0;
            // both planes get `need` elements so that one capacity covers any later (levels, chunks) split
            hipError_t e = hipMalloc((void **)&ws.p_hits, sizeof(float2) * need);
            if (e == hipSuccess) e = hipMalloc((void **)&ws.p_occ, sizeof(unsigned int) * need);
            if (e != hipSuccess) return fail(RT_E_NOMEM, std::string("hit planes: ") + hipGetErrorString(e));
            ws.p_cap = need;
        }
    }
    if (host_outputs && pixels > ws.out_cap) {
        void *old[] = { ws.out_rgb8, ws.out_id, ws.out_rgb32, ws.out_normal, ws.out_uv };
        for (void *p : old)
            if (p) (void)hipFree(p);
        ws.out_rgb8 = nullptr; ws.out_id = nullptr; ws.out_rgb32 = nullptr; ws.out_normal = nullptr; ws.out_uv = nullptr; ws.out_cap = 0;
        hipError_t e = hipMalloc((void **)&ws.out_rgb8, 3 * pixels);
        if (e == hipSuccess) e = hipMalloc((void **)&ws.out_id, sizeof(int32_t) * pixels);
        if (e == hipSuccess) e = hipMalloc((void **)&ws.out_rgb32, sizeof(float) * 3 * pixels);
        if (e == hipSuccess) e = hipMalloc((void **)&ws.out_normal, sizeof(float) * 3 * pixels);
        if (e == hipSuccess) e = hipMalloc((void **)&ws.out_uv, sizeof(float) * 2 * pixels);
        if (e != hipSuccess) return fail(RT_E_NOMEM, std::string("output staging: ") + hipGetErrorString(e));
        ws.out_cap = pixels;
    }
    return RT_OK;
}

// Cells per axis a large triangle may be cut into when the hierarchy is built (RT_OPT_SPLIT; 0 = the default).
int split_cells(const rt_ctx *c) { return c->opt_split == 0 ? 32 : c->opt_split; }

// Decide linear scan vs LBVH for this frame and make sure the LBVH exists when wanted.
int ensure_accel(rt_ctx *c, int *accel)
{
    bool want = c->opt_accel == 2 || (c->opt_accel == 0 && c->n_tris > RT_BRUTE_MAX);
    if (c->n_tris < 2) want = false;
    if (want && (!c->bvh.nodes || c->bvh.leaf_max != c->opt_leaf_max || c->bvh.builder != c->opt_builder || c->bvh.split != split_cells(c))) {
        // A rebuild frees the nodes and triangle copy that frames of other slots may still be reading, and a captured
        // graph holds their addresses: wait for every slot and drop the graphs first, as the scene-changing calls do.
        if (c->bvh.nodes) {
            quiesce(c);
            drop_graphs(c);
            c->state_version++;
        }
        std::string err;
        int rc = lbvh_build(c->d_tri_mt, c->n_tris, c->opt_leaf_max, c->opt_builder, split_cells(c), c->stream, &c->bvh, &err);
        if (rc != RT_OK) return fail(rc, err);
        c->bvh.builder = c->opt_builder;
        if (c->bvh.depth > 64) {
            // do not keep a hierarchy no kernel may traverse: the next call must fail the same way
            lbvh_free(&c->bvh);
            return fail(RT_E_LIMIT, "LBVH deeper than the 64-entry traversal stack");
        }
    }
    *accel = want ? 1 : 0;
    return RT_OK;
}

int finish_triangles(rt_ctx *c, int n_tris, const rt_material *mat)
{
    DevMaterial m = to_dev(mat);
    if (hipMalloc((void **)&c->d_mats, sizeof(DevMaterial)) != hipSuccess) return fail(RT_E_NOMEM, "hipMalloc(mats)");
    if (hipMemcpyAsync(c->d_mats, &m, sizeof m, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return fail(RT_E_DEVICE, "upload material");
    c->n_mats = 1;
    c->n_tris = n_tris;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(RT_E_DEVICE, "triangle set-up failed");
    int accel = 0;
    return ensure_accel(c, &accel);
}

int validate_rows(int W, int H, const rt_rowset *r)
{
    if (W <= 0 || H <= 0) return fail(RT_E_INVALID, "W and H must be positive");
    if (!r || r->band_rows < 1 || r->band_stride < 1 || r->first_band < 0 || r->n_rows < 0)
        return fail(RT_E_INVALID, "bad rowset");
    if (r->n_rows > 0) {
        long long last = r->n_rows - 1;
        long long y = ((long long)r->first_band + (last / r->band_rows) * r->band_stride) * r->band_rows + last % r->band_rows;
        if (y >= H) return fail(RT_E_INVALID, "rowset reaches past the last image row");
    }
    if ((long long)W * r->n_rows >= (1ll << 31)) return fail(RT_E_LIMIT, "more than 2^31 pixels in one call");
    return RT_OK;
}

DevScene device_scene(const rt_ctx *c, int accel)
{
    DevScene sc;
    memset(&sc, 0, sizeof sc);      // padding included: the struct is compared bytewise as part of the captured graph's key
    sc.analytic = c->d_analytic; sc.lights = c->d_lights;
    sc.tri_mt = c->d_tri_mt; sc.tri_n = c->d_tri_n; sc.tri_vn = c->d_tri_vn; sc.tri_vt = c->d_tri_vt; sc.mats = c->d_mats;
    sc.nodes = c->bvh.nodes; sc.wnodes = c->bvh.wnodes; sc.fnodes = c->bvh.fnodes; sc.grid = c->bvh.grid; sc.bvh_tri = c->bvh.bvh_tri;
    sc.n_analytic = c->n_analytic; sc.n_lights = c->n_lights; sc.n_tris = c->n_tris; sc.accel = accel;
    sc.n_refs = c->bvh.n_refs;
    return sc;
}

int render_common(rt_ctx *c, int W, int H, const rt_rowset *rows, int max_depth, void *d_rgb8, void *d_hit_id,
                  void *d_rgb32, void *d_hit_t, hipStream_t stream, bool host_outputs, bool host_aov = false, int slot = 0,
                  unsigned host_want = 7u /* host_outputs: bit 0 rgb8, 1 hit_id, 2 rgb32 */)
{
    if (slot < 0 || slot >= RT_MAX_FRAME_SLOTS) return fail(RT_E_INVALID, "frame slot out of range");
    if (int rc = validate_rows(W, H, rows)) return rc;
    if (max_depth < 0 || max_depth > RT_MAX_DEPTH) return fail(RT_E_INVALID, "max_depth out of range");
    if (c->n_analytic == 0 && c->n_tris == 0) return fail(RT_E_STATE, "no scene set");
    c->f = &c->slots[slot];
    c->f->frame_valid = false;
    const size_t pixels = (size_t)W * (size_t)rows->n_rows;
    if (pixels == 0) return RT_OK;
    int accel = 0;
    if (int rc = ensure_accel(c, &accel)) return rc;
    const int pipe = accel == 1 ? lbvh_pipeline(c->opt_pipeline, c->n_lights) : 1;
    const bool wavefront = pipe != 1;
    const size_t slots = (size_t)((W + 7) / 8) * (size_t)((rows->n_rows + 7) / 8) * 64;
    if (int rc = ensure_workspace(c, pixels, slots, max_depth + 1, pipe, host_outputs)) return rc;
    // one workspace serves every frame of this slot: a frame on another stream than the slot's last one waits for it
    if (c->f->frame_pending && stream != c->f->last_stream && hipStreamWaitEvent(stream, c->f->frame_done, 0) != hipSuccess)
        return fail(RT_E_DEVICE, "hipStreamWaitEvent failed");

    const DevScene sc = device_scene(c, accel);

    FrameParams fp;
    memset(&fp, 0, sizeof fp);      // padding included, as for the scene
    fp.W = W; fp.H = H; fp.rows = *rows; fp.max_depth = max_depth;
    fp.count_tests = c->opt_count; fp.timing = c->opt_timing;
    fp.stack_depth = accel ? c->bvh.depth : 0;
    fp.wide_depth = accel ? c->bvh.wide_depth : 0;
    fp.pipeline = c->opt_pipeline;
    fp.d_normal = nullptr;
    fp.d_uv = nullptr;
    if (host_outputs) {     // only the planes the caller asked for are produced
        d_rgb8 = (host_want & 1u) ? c->f->ws.out_rgb8 : nullptr;
        d_hit_id = (host_want & 2u) ? c->f->ws.out_id : nullptr;
        d_rgb32 = (host_want & 4u) ? c->f->ws.out_rgb32 : nullptr;
    }
    if (host_aov) {
        if (!c->f->ws.out_normal || c->f->ws.out_cap < pixels) return fail(RT_E_STATE, "plane staging missing");
        d_hit_t = c->f->ws.hit_t;
        fp.d_normal = c->f->ws.out_normal;
        fp.d_uv = c->f->ws.out_uv;
    }
    std::string err;
    int launches = 0;
    int rc = RT_OK;
    if (c->opt_claim_order == 2) {          // asked for a new order: every slot makes one at its next frame
        for (FrameSlot &f : c->slots) f.ws.order_valid = false;
        c->opt_claim_order = 1;
    }
    if (wavefront && c->opt_claim_order && c->f->ws.unit_cost) {
        // The claim order of this frame layout: made once, by a counting frame that records every unit's longest ray, and
        // used by every later frame of the layout (a moved camera or refitted mesh keeps it: it only decides WHEN a ray
        // is traced).  A new mesh or new analytic primitives (scene_version) make a new one.
        struct OrderKey { int W, depth, n_lights, n_analytic, n_tris; rt_rowset rows; unsigned long long scene; } ok;
        memset(&ok, 0, sizeof ok);
        ok.W = W; ok.depth = max_depth; ok.n_lights = c->n_lights; ok.n_analytic = c->n_analytic; ok.n_tris = c->n_tris; ok.rows = *rows;
        ok.scene = c->scene_version;
        FrameWorkspace &ws = c->f->ws;
        const bool have = ws.order_valid && ws.order_key.size() == sizeof ok && memcmp(ws.order_key.data(), &ok, sizeof ok) == 0;
        // An order is made when a layout is asked for the second time in a row (like the graph capture below): a caller
        // that renders one frame, or never the same layout twice, pays nothing for it and runs in plain unit order.
        const bool repeats = ws.seen_key.size() == sizeof ok && memcmp(ws.seen_key.data(), &ok, sizeof ok) == 0;
        ws.seen_key.assign((const unsigned char *)&ok, (const unsigned char *)&ok + sizeof ok);
        if (!have) ws.order_valid = false;
        if (!have && repeats) {
            FrameParams cal = fp;
            cal.count_tests = 2; cal.timing = 0; cal.track_units = 1; cal.use_order = 0;
            const size_t n = ws.unit_stride * (size_t)ws.unit_launches;
            if (hipMemsetAsync(ws.unit_cost, 0, sizeof(unsigned int) * n, stream) != hipSuccess) return fail(RT_E_DEVICE, "clear unit costs");
            int cal_launches = 0;
            rc = trace_frame(sc, c->cam, cal, ws, d_rgb8, d_hit_id, d_rgb32, d_hit_t, stream, &cal_launches, &err);
            if (rc != RT_OK) return fail(rc, err);
            std::vector<unsigned int> cost(n), order(n);
            if (hipMemcpyAsync(cost.data(), ws.unit_cost, sizeof(unsigned int) * n, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess)
                return fail(RT_E_DEVICE, "read unit costs");
            // per launch: units by falling cost, equal costs by rising index (a counting sort; costs are step counts).  Units
            // past the launch's own count cost nothing and therefore come last: the first n_units entries of a row are a
            // permutation of that launch's units.
            constexpr unsigned int BUCKETS = 4096;
            std::vector<size_t> start(BUCKETS + 1);
            for (int l = 0; l < ws.unit_launches; l++) {
                unsigned int *o = order.data() + (size_t)l * ws.unit_stride;
                const unsigned int *cs = cost.data() + (size_t)l * ws.unit_stride;
                std::fill(start.begin(), start.end(), (size_t)0);
                auto bucket = [](unsigned int v) { return BUCKETS - 1 - (v < BUCKETS ? v : BUCKETS - 1); };     // high cost first
                for (size_t k = 0; k < ws.unit_stride; k++) start[bucket(cs[k]) + 1]++;
                for (unsigned int b = 0; b < BUCKETS; b++) start[b + 1] += start[b];
                for (size_t k = 0; k < ws.unit_stride; k++) o[start[bucket(cs[k])]++] = (unsigned int)k;
            }
            if (hipMemcpyAsync(ws.unit_order, order.data(), sizeof(unsigned int) * n, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess)
                return fail(RT_E_DEVICE, "upload claim order");
            ws.order_key.assign((const unsigned char *)&ok, (const unsigned char *)&ok + sizeof ok);
            ws.order_valid = true;
            // Which kernel for the camera rays?  Packets (one wave walks the tree once for the 64 rays of a tile) win where
            // tiles see large surfaces and lose where they see slivers and wires; both are timed here, once, on this very
            // frame (second run of each: the first loads the code), and the faster one serves the layout.
            ws.camera_packet = 0;
            if (c->opt_camera_packet < 0 && wavefront_packet_possible(sc, fp)) {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
                    if (e0) (void)hipEventDestroy(e0);
                    return fail(RT_E_DEVICE, "hipEventCreate failed");
                }
                float ms[2] = { 0.f, 0.f };
                FrameParams cam_only = fp;
                cam_only.count_tests = 0; cam_only.timing = 0; cam_only.use_order = 1; cam_only.camera_only = 1;
                for (int mode = 0; mode < 2 && rc == RT_OK; mode++) {
                    cam_only.camera_packet = mode;
                    for (int run = 0; run < 2 && rc == RT_OK; run++) {
                        int n_l = 0;
                        if (run == 1 && hipEventRecord(e0, stream) != hipSuccess) rc = RT_E_DEVICE;
                        if (rc == RT_OK) rc = trace_frame(sc, c->cam, cam_only, ws, d_rgb8, d_hit_id, d_rgb32, d_hit_t, stream, &n_l, &err);
                        if (rc == RT_OK && run == 1 && (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                                                        hipEventElapsedTime(&ms[mode], e0, e1) != hipSuccess))
                            rc = RT_E_DEVICE;
                    }
                }
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                if (rc != RT_OK) return fail(rc, err.empty() ? "timing the camera-ray kernels failed" : err);
                ws.camera_packet = ms[1] < ms[0] ? 1 : 0;
                if (getenv("RT_WF_DEBUG")) fprintf(stderr, "[rt] camera rays: per-lane %.3f ms, packets %.3f ms -> %s\n", ms[0], ms[1], ws.camera_packet ? "packets" : "per-lane");
            }
        }
        fp.use_order = ws.order_valid ? 1 : 0;
    }
    fp.camera_packet = c->opt_camera_packet >= 0 ? c->opt_camera_packet : ((wavefront && fp.use_order) ? c->f->ws.camera_packet : 0);
    const bool want_graph = c->opt_graph && !fp.timing && !fp.count_tests;
    if (want_graph) {
        // The launches of a frame depend only on these values: when none changed since the last frame,
        // replay the captured graph instead of issuing a dozen launches one by one.
        struct Key {
            FrameParams fp; DevScene sc; DevCamera cam; void *out[4]; void *ws[16]; unsigned long long version;
        } key;
        memset(&key, 0, sizeof key);
        memcpy(&key.fp, &fp, sizeof fp); memcpy(&key.sc, &sc, sizeof sc); key.cam = c->cam;
        key.out[0] = d_rgb8; key.out[1] = d_hit_id; key.out[2] = d_rgb32; key.out[3] = d_hit_t;
        void *wsp[16] = { c->f->ws.records, c->f->ws.queue_o[0], c->f->ws.queue_o[1], c->f->ws.queue_d[0], c->f->ws.queue_d[1], c->f->ws.counters,
                          c->f->ws.p_hits, c->f->ws.p_occ, c->f->ws.sq_o, c->f->ws.unit_order, c->f->ws.trace_cursors, c->f->ws.overflow,
                          c->f->ws.rq_counts[0], c->f->ws.rq_counts[1], c->f->ws.sq_counts, c->f->ws.hit_t };
        memcpy(key.ws, wsp, sizeof wsp);
        key.version = c->state_version;
        const bool same = c->f->graph_exec && c->f->graph_key.size() == sizeof key && memcmp(c->f->graph_key.data(), &key, sizeof key) == 0;
        // A frame that differs from the last one (moving camera, moving geometry) is launched directly; the
        // capture is made when the same frame is asked for a second time in a row, so that a changing frame
        // never pays for a capture it will not replay.
        const bool repeat = c->f->last_key.size() == sizeof key && memcmp(c->f->last_key.data(), &key, sizeof key) == 0;
        c->f->last_key.assign((const unsigned char *)&key, (const unsigned char *)&key + sizeof key);
        if (!same) {
            drop_graph(*c->f);
            hipGraph_t graph = nullptr;
            if (repeat && hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                rc = trace_frame(sc, c->cam, fp, c->f->ws, d_rgb8, d_hit_id, d_rgb32, d_hit_t, c->stream, &launches, &err);
                const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
                if (rc == RT_OK && ee == hipSuccess && graph &&
                    hipGraphInstantiate(&c->f->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    c->f->graph_key.assign((const unsigned char *)&key, (const unsigned char *)&key + sizeof key);
                    c->f->last_launches = launches;
                } else {
                    c->f->graph_exec = nullptr;
                }
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
            }
            if (rc != RT_OK) return fail(rc, err);
        }
        if (c->f->graph_exec) {
            if (hipGraphLaunch(c->f->graph_exec, stream) != hipSuccess) {
                drop_graph(*c->f);
                return fail(RT_E_DEVICE, "hipGraphLaunch failed");
            }
            launches = c->f->last_launches;
        } else {
            rc = trace_frame(sc, c->cam, fp, c->f->ws, d_rgb8, d_hit_id, d_rgb32, d_hit_t, stream, &launches, &err);
        }
    } else {
        rc = trace_frame(sc, c->cam, fp, c->f->ws, d_rgb8, d_hit_id, d_rgb32, d_hit_t, stream, &launches, &err);
    }
    if (rc != RT_OK) return fail(rc, err);
    if (hipEventRecord(c->f->frame_done, stream) != hipSuccess) return fail(RT_E_DEVICE, "hipEventRecord failed");
    c->f->frame_pending = true;
    c->f->last_stream = stream;
    c->f->last_counters = c->f->ws.counters;
    c->f->last_staging_ms = 0.f;
    c->f->last_timed = fp.timing != 0;
    c->f->frame_valid = true;
    c->f->last_levels = max_depth + 1;
    c->f->last_launches = launches;
    c->f->last_accel = accel;
    c->f->last_wavefront = wavefront;
    c->f->last_primary = pixels;
    return RT_OK;
}

// One plane of a host hand-over.
struct HostPlane { void *dst; const void *d_src; size_t bytes; };

bool host_range_pinned(const rt_ctx *c, const void *p, size_t bytes)
{
    const unsigned char *q = (const unsigned char *)p;
    for (const rt_ctx::PinnedRange &r : c->pinned)
        if (q >= r.p && q + bytes <= r.p + r.bytes) return true;
    return false;
}

// Device planes into the caller's host buffers, behind the frame on the context's stream; returns when they are there.
// A plane inside a buffer of rt_alloc_host_buffer is written by the device directly.  Any other buffer is
// pageable: a device copy into it would go through the runtime's own bounce buffer at a fraction of the bus rate, so
// such a plane goes into the context's pinned staging buffer in a few pieces (HOST_PIECES_DEFAULT), an event behind each, and the host
// copies piece k on into the caller's buffer while the later pieces are still in flight.
int hand_over(rt_ctx *c, const HostPlane *planes, int n_planes)
{
    if (n_planes > HOST_PLANES_MAX) return fail(RT_E_INVALID, "too many planes in one hand-over");
    size_t staged = 0;
    for (int k = 0; k < n_planes; k++)
        if (planes[k].dst && planes[k].bytes && !host_range_pinned(c, planes[k].dst, planes[k].bytes)) staged += (planes[k].bytes + 255) & ~(size_t)255;
    if (staged > c->h_stage_cap) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr; c->h_stage_cap = 0;
        if (hipHostMalloc((void **)&c->h_stage, staged, hipHostMallocDefault) != hipSuccess)
            return fail(RT_E_NOMEM, "pinned staging buffer for the host hand-over");
        c->h_stage_cap = staged;
    }
    struct Piece { void *dst; const unsigned char *src; size_t bytes; hipEvent_t ev; } pieces[HOST_PLANES_MAX * HOST_PIECES];
    int n_pieces = 0;
    size_t base = 0;
    hipError_t e = hipSuccess;
    for (int k = 0; k < n_planes && e == hipSuccess; k++) {
        const HostPlane &pl = planes[k];
        if (!pl.dst || !pl.bytes) continue;
        if (host_range_pinned(c, pl.dst, pl.bytes)) {
            e = hipMemcpyAsync(pl.dst, pl.d_src, pl.bytes, hipMemcpyDeviceToHost, c->stream);
            continue;
        }
        // small planes go in one piece; pieces are multiples of 4 KiB
        static const int pieces_env = getenv("RT_HOST_PIECES") ? atoi(getenv("RT_HOST_PIECES")) : 0;   // (measurements)
        const int want = pl.bytes >= ((size_t)1 << 20) ? (pieces_env >= 1 && pieces_env <= HOST_PIECES ? pieces_env : HOST_PIECES_DEFAULT) : 1;
        const size_t step = ((pl.bytes + want - 1) / want + 4095) & ~(size_t)4095;
        for (size_t off = 0; off < pl.bytes && e == hipSuccess; off += step) {
            const size_t n = off + step <= pl.bytes ? step : pl.bytes - off;
            hipEvent_t &ev = c->h_events[n_pieces];
            if (!ev) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipMemcpyAsync(c->h_stage + base + off, (const unsigned char *)pl.d_src + off, n, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventRecord(ev, c->stream);
            pieces[n_pieces++] = { (unsigned char *)pl.dst + off, c->h_stage + base + off, n, ev };
        }
        base += (pl.bytes + 255) & ~(size_t)255;
    }
    // One host thread copies ~9 GB/s, a fraction of what the bus delivers.  A large hand-over is therefore copied on by
    // helper threads: this thread only waits for the pieces' events, in order, and publishes how many pieces have landed;
    // every helper copies its share of each landed piece.  The helpers make no HIP call and are started before the frame has finished
    // rendering, so their start-up costs nothing; until the first piece lands they sleep in short naps.
    constexpr int HELPERS = 4;
    bool threaded = e == hipSuccess && staged >= ((size_t)2 << 20);
    std::atomic<int> landed{ 0 };
    std::atomic<bool> give_up{ false };
    std::thread helpers[HELPERS];
    int started = 0;
    if (threaded)
        for (int k = 0; k < HELPERS; k++) {
            try {
                helpers[k] = std::thread([&pieces, &landed, &give_up, n_pieces, k] {
                    for (int q = 0; q < n_pieces; q++) {
                        while (landed.load(std::memory_order_acquire) <= q) {
                            if (give_up.load(std::memory_order_relaxed)) return;
                            if (landed.load(std::memory_order_relaxed) == 0) std::this_thread::sleep_for(std::chrono::microseconds(30));
                            else std::this_thread::yield();
                        }
                        // every helper takes its quarter of every piece (whole 4 KiB pages), so the last piece - the only one
                        // whose copy nothing hides - is copied at all helpers' rate
                        const size_t part = ((pieces[q].bytes + HELPERS - 1) / HELPERS + 4095) & ~(size_t)4095;
                        const size_t lo = (size_t)k * part;
                        if (lo < pieces[q].bytes) {
                            const size_t n = lo + part <= pieces[q].bytes ? part : pieces[q].bytes - lo;
                            memcpy((unsigned char *)pieces[q].dst + lo, pieces[q].src + lo, n);
                        }
                    }
                });
                started++;
            } catch (...) {
                break;      // no thread to be had: the helpers that did start are sent home, this thread copies everything
            }
        }
    if (threaded && started < HELPERS) {
        give_up.store(true);
        for (int k = 0; k < started; k++) helpers[k].join();
        give_up.store(false);
        threaded = false;
    }
    // the frame itself first (the clock of staging_ms starts when the device has finished rendering)
    if (e == hipSuccess && c->f->frame_done) e = hipEventSynchronize(c->f->frame_done);
    const auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < n_pieces && e == hipSuccess; q++) {
        e = hipEventSynchronize(pieces[q].ev);
        if (e != hipSuccess) break;
        if (threaded) landed.store(q + 1, std::memory_order_release);
        else memcpy(pieces[q].dst, pieces[q].src, pieces[q].bytes);
    }
    if (threaded) {
        if (e != hipSuccess) give_up.store(true);
        for (std::thread &t : helpers) t.join();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);     // nothing of this call may still be writing the staging buffer
        return fail(RT_E_DEVICE, std::string("frame readback: ") + hipGetErrorString(e));
    }
    c->f->last_staging_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

}  // namespace

// No C++ exception leaves the library: a caller of a C ABI cannot catch one.  Every entry point with a body that can throw
// (std::vector, std::string, std::thread) is a function-try-block that turns the exception into an error code.  The texts are
// short enough for the string's own buffer, so reporting "out of memory" allocates nothing.
#define RT_ABI_CATCH_INT                                                                       \
    catch (const std::bad_alloc &) { g_err = "out of memory"; return RT_E_NOMEM; }             \
    catch (...) { g_err = "C++ exception"; return RT_E_DEVICE; }
#define RT_ABI_CATCH_PTR                                                                       \
    catch (const std::bad_alloc &) { g_err = "out of memory"; return nullptr; }                \
    catch (...) { g_err = "C++ exception"; return nullptr; }
#define RT_ABI_CATCH_VOID catch (...) { }

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char *rt_last_error(void) { return g_err.c_str(); }

rt_ctx *rt_create(int device_ordinal)
try {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fail(RT_E_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
        return nullptr;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        fail(RT_E_INVALID, "device ordinal out of range");
        return nullptr;
    }
    rt_ctx *c = new (std::nothrow) rt_ctx();
    if (!c) { fail(RT_E_NOMEM, "out of host memory"); return nullptr; }
    c->device = device_ordinal;
    DeviceGuard g(c->device);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        fail(RT_E_DEVICE, "hipStreamCreate failed");
        delete c;
        return nullptr;
    }
    {
        std::string err;
        const hipError_t se = scratch_for_device(c->device, &c->scratch_ptr);
        if (se != hipSuccess || device_num_cus(&err) <= 0) {
            fail(RT_E_DEVICE, se != hipSuccess ? std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(se) : err);
            (void)hipStreamDestroy(c->stream);
            delete c;
            return nullptr;
        }
    }
    for (FrameSlot &f : c->slots) {
        if (hipEventCreateWithFlags(&f.frame_done, hipEventDisableTiming) != hipSuccess) {
            fail(RT_E_DEVICE, "hipEventCreate failed");
            for (FrameSlot &g2 : c->slots)
                if (g2.frame_done) (void)hipEventDestroy(g2.frame_done);
            (void)hipStreamDestroy(c->stream);
            delete c;
            return nullptr;
        }
    }
    default_camera(&c->cam);
    return c;
} RT_ABI_CATCH_PTR

void rt_destroy(rt_ctx *c)
try {
    if (!c) return;
    DeviceGuard g(c->device);
    quiesce(c);
    drop_graphs(c);
    free_triangles(c);
    for (FrameSlot &f : c->slots) free_workspace(f.ws);
    if (c->d_analytic) (void)hipFree(c->d_analytic);
    if (c->d_lights) (void)hipFree(c->d_lights);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (hipEvent_t ev : c->h_events)
        if (ev) (void)hipEventDestroy(ev);
    for (const rt_ctx::PinnedRange &r : c->pinned) (void)hipHostFree(r.p);
    for (FrameSlot &f : c->slots)
        if (f.frame_done) (void)hipEventDestroy(f.frame_done);
    (void)hipStreamDestroy(c->stream);
    delete c;
} RT_ABI_CATCH_VOID

int rt_set_option(rt_ctx *c, int option, int value)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    switch (option) {
    case RT_OPT_ACCEL:
        if (value < 0 || value > 2) return fail(RT_E_INVALID, "RT_OPT_ACCEL takes 0, 1 or 2");
        c->opt_accel = value;
        return RT_OK;
    case RT_OPT_COUNT_TESTS: c->opt_count = value ? 1 : 0; return RT_OK;
    case RT_OPT_LEAF_MAX:
        if (value < 1 || value > 8) return fail(RT_E_INVALID, "RT_OPT_LEAF_MAX takes 1..8");
        c->opt_leaf_max = value;
        return RT_OK;
    case RT_OPT_TIMING: c->opt_timing = value ? 1 : 0; return RT_OK;
    case RT_OPT_BUILDER:
        if (value < 0 || value > 3) return fail(RT_E_INVALID, "RT_OPT_BUILDER takes 0..3");
        c->opt_builder = value;
        return RT_OK;
    case RT_OPT_SPLIT:
        if (value < 0 || value > 32) return fail(RT_E_INVALID, "RT_OPT_SPLIT takes 0..32");
        c->opt_split = value;
        return RT_OK;
    case RT_OPT_GRAPH: c->opt_graph = value ? 1 : 0; drop_graphs(c); return RT_OK;
    case RT_OPT_CLAIM_ORDER:
        if (value < 0 || value > 2) return fail(RT_E_INVALID, "RT_OPT_CLAIM_ORDER takes 0, 1 or 2");
        c->opt_claim_order = value;
        return RT_OK;
    case RT_OPT_PIPELINE:
        if (value < 0 || value > 2) return fail(RT_E_INVALID, "RT_OPT_PIPELINE takes 0, 1 or 2");
        c->opt_pipeline = value;
        return RT_OK;
    default: return fail(RT_E_INVALID, "unknown option");
    }
} RT_ABI_CATCH_INT

int rt_set_analytic(rt_ctx *c, const rt_analytic *prims, int n)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (n < 0 || (n > 0 && !prims)) return fail(RT_E_INVALID, "bad analytic array");
    if (n > RT_MAX_ANALYTIC) return fail(RT_E_LIMIT, "more than RT_MAX_ANALYTIC analytic primitives");
    std::vector<DevAnalytic> host((size_t)n);
    std::vector<int32_t> lights;
    for (int i = 0; i < n; i++) {
        const rt_analytic &p = prims[i];
        if (p.kind != RT_SPHERE && p.kind != RT_PLANE) return fail(RT_E_INVALID, "unknown analytic kind");
        if (p.is_light && p.kind != RT_SPHERE)
            return fail(RT_E_INVALID, "only spheres can be lights (the reference reads a light's position through Sphere*)");
        DevAnalytic d;
        memset(&d, 0, sizeof d);
        d.px = p.p[0]; d.py = p.p[1]; d.pz = p.p[2]; d.s = p.s;
        d.s2 = p.s * p.s;   // Sphere::radiusSq, main.cpp:53
        d.kind = p.kind; d.is_light = p.is_light ? 1 : 0;
        d.reflect = p.mat.reflect; d.diffuse = p.mat.diffuse; d.spec = p.mat.spec;
        d.cr = p.mat.color[0]; d.cg = p.mat.color[1]; d.cb = p.mat.color[2];
        host[(size_t)i] = d;
        if (p.is_light) lights.push_back(i);
    }
    DeviceGuard g(c->device);
    quiesce(c);
    c->scene_version++;
    if (c->d_analytic) (void)hipFree(c->d_analytic);
    if (c->d_lights) (void)hipFree(c->d_lights);
    c->d_analytic = nullptr; c->d_lights = nullptr; c->n_analytic = 0; c->n_lights = 0;
    if (hipMalloc((void **)&c->d_analytic, sizeof(DevAnalytic) * (size_t)(n ? n : 1)) != hipSuccess ||
        hipMalloc((void **)&c->d_lights, sizeof(int32_t) * (lights.size() ? lights.size() : 1)) != hipSuccess)
        return fail(RT_E_NOMEM, "hipMalloc(analytic) failed");
    if (n && hipMemcpy(c->d_analytic, host.data(), sizeof(DevAnalytic) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess)
        return fail(RT_E_DEVICE, "upload analytic primitives");
    if (!lights.empty() &&
        hipMemcpy(c->d_lights, lights.data(), sizeof(int32_t) * lights.size(), hipMemcpyHostToDevice) != hipSuccess)
        return fail(RT_E_DEVICE, "upload light list");
    c->n_analytic = n;
    c->n_lights = (int)lights.size();
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_set_triangles(rt_ctx *c, const float *tri_verts, int n_tris, const rt_material *mat)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (n_tris < 0 || (n_tris > 0 && !tri_verts)) return fail(RT_E_INVALID, "bad triangle array");
    if ((long long)n_tris * 8 >= (1ll << 31)) return fail(RT_E_LIMIT, "too many triangles");
    DeviceGuard g(c->device);
    quiesce(c);
    free_triangles(c);
    if (n_tris == 0) return RT_OK;
    float *d_v = nullptr;
    if (hipMalloc((void **)&d_v, sizeof(float) * 9 * (size_t)n_tris) != hipSuccess ||
        hipMalloc((void **)&c->d_tri_mt, sizeof(float4) * 3 * (size_t)n_tris) != hipSuccess ||
        hipMalloc((void **)&c->d_tri_n, sizeof(float4) * (size_t)n_tris) != hipSuccess) {
        if (d_v) (void)hipFree(d_v);
        free_triangles(c);
        return fail(RT_E_NOMEM, "hipMalloc(triangles) failed");
    }
    int rc = RT_OK;
    if (hipMemcpyAsync(d_v, tri_verts, sizeof(float) * 9 * (size_t)n_tris, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        rc = fail(RT_E_DEVICE, "upload triangle vertices");
    if (rc == RT_OK && tri_setup_flat(d_v, n_tris, c->d_tri_mt, c->d_tri_n, 0, c->stream) != RT_OK)
        rc = fail(RT_E_DEVICE, "triangle set-up launch failed");
    if (rc == RT_OK) rc = finish_triangles(c, n_tris, mat);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_v);
    if (rc != RT_OK) free_triangles(c);
    return rc;
} RT_ABI_CATCH_INT

int rt_set_mesh(rt_ctx *c, const float *verts, int n_verts, const int32_t *indices, int n_indices, const rt_material *mat)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (n_verts < 0 || n_indices < 0 || (n_verts > 0 && !verts) || (n_indices > 0 && !indices))
        return fail(RT_E_INVALID, "bad mesh arrays");
    // main.cpp:228 steps i += 3 and reads inds[i+1], inds[i+2]: a count that is not a
    // multiple of 3 reads past the vector there; refuse it here.
    if (n_indices % 3) return fail(RT_E_INVALID, "index count is not a multiple of 3");
    const int n_tris = n_indices / 3;
    if ((long long)n_tris * 8 >= (1ll << 31)) return fail(RT_E_LIMIT, "too many triangles");
    DeviceGuard g(c->device);
    quiesce(c);
    free_triangles(c);
    if (n_tris == 0) return RT_OK;
    if (n_verts == 0) return fail(RT_E_INVALID, "indices without vertices");
    float *d_v = nullptr;
    int32_t *d_i = nullptr;
    int *d_bad = nullptr;
    hipError_t e = hipMalloc((void **)&d_v, sizeof(float) * 3 * (size_t)n_verts);
    if (e == hipSuccess) e = hipMalloc((void **)&d_i, sizeof(int32_t) * (size_t)n_indices);
    if (e == hipSuccess) e = hipMalloc((void **)&d_bad, sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_tri_mt, sizeof(float4) * 3 * (size_t)n_tris);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_tri_n, sizeof(float4) * (size_t)n_tris);
    int rc = RT_OK;
    if (e != hipSuccess) rc = fail(RT_E_NOMEM, "hipMalloc(mesh) failed");
    int bad = 0;
    if (rc == RT_OK) {
        e = hipMemcpyAsync(d_v, verts, sizeof(float) * 3 * (size_t)n_verts, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_i, indices, sizeof(int32_t) * (size_t)n_indices, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), c->stream);
        if (e != hipSuccess) rc = fail(RT_E_DEVICE, "upload mesh");
    }
    if (rc == RT_OK && tri_setup_indexed(d_v, n_verts, d_i, n_tris, c->d_tri_mt, c->d_tri_n, 0, d_bad, c->stream) != RT_OK)
        rc = fail(RT_E_DEVICE, "triangle set-up launch failed");
    if (rc == RT_OK && hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        rc = fail(RT_E_DEVICE, "read index check");
    if (rc == RT_OK) rc = finish_triangles(c, n_tris, mat);
    if (rc == RT_OK && bad) rc = fail(RT_E_INVALID, "vertex index out of range");
    (void)hipStreamSynchronize(c->stream);
    if (d_v) (void)hipFree(d_v);
    if (d_bad) (void)hipFree(d_bad);
    if (rc != RT_OK) {
        if (d_i) (void)hipFree(d_i);
        free_triangles(c);
    } else {
        c->d_indices = d_i;
        c->n_mesh_verts = n_verts;
    }
    return rc;
} RT_ABI_CATCH_INT

namespace {

void *stage(rt_ctx *c, size_t bytes)
{
    if (bytes > c->stage_cap) {
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr; c->stage_cap = 0;
        if (hipMalloc(&c->d_stage, bytes) != hipSuccess) return nullptr;
        c->stage_cap = bytes;
    }
    return c->d_stage;
}

// After the Triangle constructor has run again over moved vertices: refit the hierarchy if one exists
// (or build it if this frame's settings want one and none is there).
int after_vertex_update(rt_ctx *c)
{
    c->state_version++;
    for (FrameSlot &f : c->slots) f.frame_valid = false;
    if (c->bvh.wnodes && c->bvh.leaf_max == c->opt_leaf_max && c->bvh.builder == c->opt_builder && c->bvh.split == split_cells(c)) {
        std::string err;
        int rc = lbvh_refit(c->d_tri_mt, c->n_tris, c->stream, &c->bvh, &err);
        if (rc != RT_OK) return fail(rc, err);
        return RT_OK;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(RT_E_DEVICE, "triangle set-up failed");
    int accel = 0;
    return ensure_accel(c, &accel);
}

}  // namespace

int rt_update_triangles(rt_ctx *c, const float *tri_verts, int n_tris)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (!tri_verts) return fail(RT_E_INVALID, "null vertex array");
    if (c->n_tris == 0 || n_tris != c->n_tris) return fail(RT_E_STATE, "triangle count does not match the loaded triangles");
    DeviceGuard g(c->device);
    quiesce(c);
    float *d_v = (float *)stage(c, sizeof(float) * 9 * (size_t)n_tris);
    if (!d_v) return fail(RT_E_NOMEM, "hipMalloc(vertices) failed");
    int rc = RT_OK;
    if (hipMemcpyAsync(d_v, tri_verts, sizeof(float) * 9 * (size_t)n_tris, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        rc = fail(RT_E_DEVICE, "upload triangle vertices");
    if (rc == RT_OK && tri_setup_flat(d_v, n_tris, c->d_tri_mt, c->d_tri_n, -1, c->stream) != RT_OK)
        rc = fail(RT_E_DEVICE, "triangle set-up launch failed");
    if (rc == RT_OK) rc = after_vertex_update(c);
    (void)hipStreamSynchronize(c->stream);
    return rc;
} RT_ABI_CATCH_INT

int rt_update_mesh_vertices(rt_ctx *c, const float *verts, int n_verts)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (!verts) return fail(RT_E_INVALID, "null vertex array");
    if (!c->d_indices || c->n_tris == 0) return fail(RT_E_STATE, "no indexed mesh loaded (rt_set_mesh)");
    if (n_verts != c->n_mesh_verts) return fail(RT_E_STATE, "vertex count does not match the loaded mesh");
    DeviceGuard g(c->device);
    quiesce(c);
    // staging: the vertices, then one word for the index check of the set-up kernel
    const size_t vbytes = (sizeof(float) * 3 * (size_t)n_verts + 15) & ~(size_t)15;
    float *d_v = (float *)stage(c, vbytes + sizeof(int));
    if (!d_v) return fail(RT_E_NOMEM, "hipMalloc(vertices) failed");
    int *d_bad = (int *)((char *)d_v + vbytes);
    int rc = RT_OK;
    if (hipMemcpyAsync(d_v, verts, sizeof(float) * 3 * (size_t)n_verts, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemsetAsync(d_bad, 0, sizeof(int), c->stream) != hipSuccess)
        rc = fail(RT_E_DEVICE, "upload vertices");
    // the indices were range-checked against this vertex count when the mesh was loaded
    if (rc == RT_OK && tri_setup_indexed(d_v, n_verts, c->d_indices, c->n_tris, c->d_tri_mt, c->d_tri_n, -1, d_bad, c->stream) != RT_OK)
        rc = fail(RT_E_DEVICE, "triangle set-up launch failed");
    if (rc == RT_OK) rc = after_vertex_update(c);
    (void)hipStreamSynchronize(c->stream);
    return rc;
} RT_ABI_CATCH_INT

int rt_set_triangle_materials(rt_ctx *c, const rt_material *mats, int n_mats, const int32_t *tri_material, int n_tris)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (!mats || n_mats < 1 || !tri_material) return fail(RT_E_INVALID, "bad material arrays");
    if (n_tris != c->n_tris || c->n_tris == 0) return fail(RT_E_STATE, "triangle count does not match the loaded mesh");
    for (int k = 0; k < n_tris; k++)
        if (tri_material[k] < 0 || tri_material[k] >= n_mats) return fail(RT_E_INVALID, "material index out of range");
    DeviceGuard g(c->device);
    quiesce(c);
    std::vector<DevMaterial> hm((size_t)n_mats);
    for (int k = 0; k < n_mats; k++) hm[(size_t)k] = to_dev(&mats[k]);
    // tri_n.w carries the material index: patch it in place.
    std::vector<float4> hn((size_t)n_tris);
    if (hipMemcpy(hn.data(), c->d_tri_n, sizeof(float4) * (size_t)n_tris, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(RT_E_DEVICE, "read normals");
    for (int k = 0; k < n_tris; k++) memcpy(&hn[(size_t)k].w, &tri_material[k], sizeof(int32_t));
    DevMaterial *nm = nullptr;
    if (hipMalloc((void **)&nm, sizeof(DevMaterial) * (size_t)n_mats) != hipSuccess) return fail(RT_E_NOMEM, "hipMalloc(mats)");
    if (hipMemcpy(nm, hm.data(), sizeof(DevMaterial) * (size_t)n_mats, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_tri_n, hn.data(), sizeof(float4) * (size_t)n_tris, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(nm);
        return fail(RT_E_DEVICE, "upload materials");
    }
    if (c->d_mats) (void)hipFree(c->d_mats);
    c->d_mats = nm;
    c->n_mats = n_mats;
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_set_triangle_normals(rt_ctx *c, const float *corner_normals, int n_tris)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    DeviceGuard g(c->device);
    quiesce(c);
    if (!corner_normals) {                      // back to the reference's geometric normals
        if (c->d_tri_vn) (void)hipFree(c->d_tri_vn);
        c->d_tri_vn = nullptr;
        c->state_version++;
        return RT_OK;
    }
    if (n_tris != c->n_tris || c->n_tris == 0) return fail(RT_E_STATE, "triangle count does not match the loaded mesh");
    std::vector<float4> h(3 * (size_t)n_tris);
    for (size_t k = 0; k < 3 * (size_t)n_tris; k++)
        h[k] = make_float4(corner_normals[3 * k], corner_normals[3 * k + 1], corner_normals[3 * k + 2], 0.0f);
    float4 *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(float4) * h.size()) != hipSuccess) return fail(RT_E_NOMEM, "hipMalloc(normals)");
    if (hipMemcpy(d, h.data(), sizeof(float4) * h.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return fail(RT_E_DEVICE, "upload normals");
    }
    if (c->d_tri_vn) (void)hipFree(c->d_tri_vn);
    c->d_tri_vn = d;
    c->state_version++;
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_set_triangle_texcoords(rt_ctx *c, const float *corner_uv, int n_tris)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    DeviceGuard g(c->device);
    quiesce(c);
    if (!corner_uv) {
        if (c->d_tri_vt) (void)hipFree(c->d_tri_vt);
        c->d_tri_vt = nullptr;
        c->state_version++;
        return RT_OK;
    }
    if (n_tris != c->n_tris || c->n_tris == 0) return fail(RT_E_STATE, "triangle count does not match the loaded mesh");
    float2 *d = nullptr;
    const size_t bytes = sizeof(float2) * 3 * (size_t)n_tris;
    if (hipMalloc((void **)&d, bytes) != hipSuccess) return fail(RT_E_NOMEM, "hipMalloc(texcoords)");
    if (hipMemcpy(d, corner_uv, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return fail(RT_E_DEVICE, "upload texcoords");
    }
    if (c->d_tri_vt) (void)hipFree(c->d_tri_vt);
    c->d_tri_vt = d;
    c->state_version++;
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_set_camera(rt_ctx *c, const rt_camera *cam)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (!cam) { default_camera(&c->cam); return RT_OK; }
    c->cam.ox = cam->origin[0]; c->cam.oy = cam->origin[1]; c->cam.oz = cam->origin[2];
    c->cam.pitch = cam->pitch; c->cam.off_x = cam->off_x; c->cam.off_y = cam->off_y; c->cam.plane_z = cam->plane_z;
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_render_frame(rt_ctx *c, int W, int H, int y0, int y1, int max_depth, uint8_t *rgb8, int32_t *hit_id, float *rgb32)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (y0 < 0 || y1 < y0 || y1 > H) return fail(RT_E_INVALID, "bad row range");
    DeviceGuard g(c->device);
    rt_rowset rows = { y0, 1, 1, y1 - y0 };
    const unsigned want = (rgb8 ? 1u : 0u) | (hit_id ? 2u : 0u) | (rgb32 ? 4u : 0u);
    int rc = render_common(c, W, H, &rows, max_depth, nullptr, nullptr, nullptr, nullptr, c->stream, true, false, 0, want);
    if (rc != RT_OK) return rc;
    const size_t pixels = (size_t)W * (size_t)(y1 - y0);
    if (!pixels) return RT_OK;
    const HostPlane planes[3] = { { rgb8, c->f->ws.out_rgb8, 3 * pixels }, { hit_id, c->f->ws.out_id, sizeof(int32_t) * pixels },
                                  { rgb32, c->f->ws.out_rgb32, sizeof(float) * 3 * pixels } };
    return hand_over(c, planes, 3);
} RT_ABI_CATCH_INT

int rt_render_frame_planes(rt_ctx *c, int W, int H, int y0, int y1, int max_depth, const rt_frame_planes *out)
try {
    if (!c || !out) return fail(RT_E_INVALID, "null argument");
    if (y0 < 0 || y1 < y0 || y1 > H) return fail(RT_E_INVALID, "bad row range");
    DeviceGuard g(c->device);
    rt_rowset rows = { y0, 1, 1, y1 - y0 };
    const unsigned want = (out->rgb8 ? 1u : 0u) | (out->hit_id ? 2u : 0u) | (out->rgb32 ? 4u : 0u);
    int rc = render_common(c, W, H, &rows, max_depth, nullptr, nullptr, nullptr, nullptr, c->stream, true, true, 0, want);
    if (rc != RT_OK) return rc;
    const size_t pixels = (size_t)W * (size_t)(y1 - y0);
    if (!pixels) return RT_OK;
    const HostPlane planes[6] = { { out->rgb8, c->f->ws.out_rgb8, 3 * pixels }, { out->hit_id, c->f->ws.out_id, sizeof(int32_t) * pixels },
                                  { out->rgb32, c->f->ws.out_rgb32, sizeof(float) * 3 * pixels }, { out->hit_t, c->f->ws.hit_t, sizeof(float) * pixels },
                                  { out->normal, c->f->ws.out_normal, sizeof(float) * 3 * pixels }, { out->uv, c->f->ws.out_uv, sizeof(float) * 2 * pixels } };
    return hand_over(c, planes, 6);
} RT_ABI_CATCH_INT

void *rt_alloc_host_buffer(rt_ctx *c, uint64_t bytes)
try {
    if (!c || bytes == 0) { fail(RT_E_INVALID, "rt_alloc_host_buffer: null context or no bytes"); return nullptr; }
    DeviceGuard g(c->device);
    void *p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess || !p) {
        (void)hipGetLastError();
        fail(RT_E_NOMEM, "rt_alloc_host_buffer: hipHostMalloc failed");
        return nullptr;
    }
    c->pinned.push_back({ (unsigned char *)p, (size_t)bytes });
    return p;
} RT_ABI_CATCH_PTR

int rt_free_host_buffer(rt_ctx *c, void *buffer)
try {
    if (!c || !buffer) return fail(RT_E_INVALID, "rt_free_host_buffer: null argument");
    DeviceGuard g(c->device);
    for (size_t k = 0; k < c->pinned.size(); k++)
        if (c->pinned[k].p == (unsigned char *)buffer) {
            quiesce(c);
            (void)hipHostFree(buffer);
            c->pinned.erase(c->pinned.begin() + (long)k);
            return RT_OK;
        }
    return fail(RT_E_STATE, "rt_free_host_buffer: not a buffer of rt_alloc_host_buffer");
} RT_ABI_CATCH_INT

int rt_render_rows_device(rt_ctx *c, int W, int H, const rt_rowset *rows, int max_depth, void *d_rgb8, void *d_hit_id,
                          void *d_rgb32, void *d_hit_t, void *stream)
try {
    return rt_render_rows_device_slot(c, 0, W, H, rows, max_depth, d_rgb8, d_hit_id, d_rgb32, d_hit_t, stream);
} RT_ABI_CATCH_INT

int rt_render_rows_device_slot(rt_ctx *c, int slot, int W, int H, const rt_rowset *rows, int max_depth, void *d_rgb8, void *d_hit_id,
                               void *d_rgb32, void *d_hit_t, void *stream)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    DeviceGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return render_common(c, W, H, rows, max_depth, d_rgb8, d_hit_id, d_rgb32, d_hit_t, s, false, false, slot);
} RT_ABI_CATCH_INT

// The frame of rt_render_rows_device through the one-kernel-per-bounce-level pipeline, with every intermediate
// buffer (ray queues, per-level records, counters) carved out of this GPU's static scratch array: no hipMalloc,
// no hipFree, no event or graph creation.  Calls that overlap on one GPU share the array and race.
int rt_render_rows_device_noalloc(rt_ctx *c, int W, int H, const rt_rowset *rows, int max_depth, void *d_rgb8, void *d_hit_id,
                                  void *d_rgb32, void *d_hit_t, void *stream)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    DeviceGuard g(c->device);
    if (int rc = validate_rows(W, H, rows)) return rc;
    if (max_depth < 0 || max_depth > RT_MAX_DEPTH) return fail(RT_E_INVALID, "max_depth out of range");
    if (c->n_analytic == 0 && c->n_tris == 0) return fail(RT_E_STATE, "no scene set");
    if (!c->scratch_ptr) return fail(RT_E_STATE, "no scratch array on this device");
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    c->f = &c->slots[0];
    c->f->frame_valid = false;
    const size_t pixels = (size_t)W * (size_t)rows->n_rows;
    if (pixels == 0) return RT_OK;
    // the hierarchy is built when the mesh is set; building it here would allocate
    bool want = c->opt_accel == 2 || (c->opt_accel == 0 && c->n_tris > RT_BRUTE_MAX);
    if (c->n_tris < 2) want = false;
    if (want && (!c->bvh.nodes || c->bvh.leaf_max != c->opt_leaf_max || c->bvh.builder != c->opt_builder || c->bvh.split != split_cells(c)))
        return fail(RT_E_STATE, "the hierarchy for the current options is not built: set the mesh after the options, or render once "
                                "through rt_render_rows_device");
    const int accel = want ? 1 : 0;

    // ---- carve the scratch array: counters | records[levels][pixels] | 2 x (origins, directions)[qslots] | 2 x batch counts
    const size_t slots = (size_t)((W + 7) / 8) * (size_t)((rows->n_rows + 7) / 8) * 64;
    const size_t qslots = wavefront_regions(slots > pixels ? slots : pixels) * 256;
    const size_t levels = (size_t)max_depth + 1;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += up(bytes); return at; };
    const size_t o_counters = take(sizeof(FrameCounters));
    const size_t o_records = take(sizeof(float4) * pixels * levels);
    size_t o_qo[2], o_qd[2], o_fq[2];
    for (int k = 0; k < 2; k++) {
        o_qo[k] = take(sizeof(float4) * qslots);
        o_qd[k] = take(sizeof(float4) * qslots);
        o_fq[k] = take(sizeof(unsigned int) * (qslots / 64 + 1));
    }
    if (off > (size_t)RT_NOALLOC_SCRATCH_BYTES)
        return fail(RT_E_LIMIT, "frame needs " + std::to_string(off) + " bytes of scratch (" + std::to_string(pixels) + " pixels x " +
                                    std::to_string(levels) + " levels); rt_render_rows_device_noalloc holds " +
                                    std::to_string((size_t)RT_NOALLOC_SCRATCH_BYTES) + ": render fewer rows per call or use rt_render_rows_device");
    unsigned char *base = c->scratch_ptr;
    FrameWorkspace ws;
    ws.counters = (FrameCounters *)(base + o_counters);
    ws.records = (float4 *)(base + o_records);
    for (int k = 0; k < 2; k++) {
        ws.queue_o[k] = (float4 *)(base + o_qo[k]);
        ws.queue_d[k] = (float4 *)(base + o_qd[k]);
        ws.fq_counts[k] = (unsigned int *)(base + o_fq[k]);
    }
    ws.pixels_cap = qslots;
    ws.levels_cap = (int)levels;

    const DevScene sc = device_scene(c, accel);
    FrameParams fp;
    fp.W = W; fp.H = H; fp.rows = *rows; fp.max_depth = max_depth;
    fp.count_tests = 0; fp.timing = 0;             // counting builds and device events are the allocating entry point's
    fp.stack_depth = accel ? c->bvh.depth : 0;
    fp.wide_depth = accel ? c->bvh.wide_depth : 0;
    fp.pipeline = 1;                               // one fused kernel per bounce level: its whole workspace is the carve above
    fp.d_normal = nullptr; fp.d_uv = nullptr;
    // the array keeps the last call's bytes: the counters are cleared by trace_frame's memset, every batch count
    // and every record a level reads was written by the level before it in this same frame
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && c->f->frame_pending && s != c->f->last_stream && hipStreamWaitEvent(s, c->f->frame_done, 0) != hipSuccess)
        return fail(RT_E_DEVICE, "hipStreamWaitEvent failed");
    std::string err;
    int launches = 0;
    const int rc = trace_frame(sc, c->cam, fp, ws, d_rgb8, d_hit_id, d_rgb32, d_hit_t, s, &launches, &err);
    if (rc != RT_OK) return fail(rc, err);
    if (!capturing) {
        if (hipEventRecord(c->f->frame_done, s) != hipSuccess) return fail(RT_E_DEVICE, "hipEventRecord failed");
        c->f->frame_pending = true;
        c->f->last_stream = s;
    }
    c->f->last_counters = ws.counters;
    c->f->last_staging_ms = 0.f;
    c->f->last_timed = false;
    c->f->frame_valid = true;
    c->f->last_levels = max_depth + 1;
    c->f->last_launches = launches;
    c->f->last_accel = accel;
    c->f->last_wavefront = false;
    c->f->last_primary = pixels;
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_deinterleave_bands(rt_ctx *c, const void *d_src, void *d_dst, int W, int H, int band_rows, int n_parts,
                          int rows_per_part, int bytes_per_pixel, void *stream)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    if (!d_src || !d_dst || W <= 0 || H <= 0 || band_rows < 1 || n_parts < 1 || rows_per_part < 0)
        return fail(RT_E_INVALID, "bad de-interleave arguments");
    if (bytes_per_pixel != 3 && bytes_per_pixel != 4 && bytes_per_pixel != 12)
        return fail(RT_E_INVALID, "bytes_per_pixel must be 3, 4 or 12");
    // every image row must exist inside its part
    const int n_bands = (H + band_rows - 1) / band_rows;
    const int bands_per_part = (n_bands + n_parts - 1) / n_parts;
    if (bands_per_part * band_rows > rows_per_part) return fail(RT_E_INVALID, "rows_per_part too small for this split");
    DeviceGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (trace_deinterleave(d_src, d_dst, W, H, band_rows, n_parts, rows_per_part, bytes_per_pixel, s) != RT_OK)
        return fail(RT_E_DEVICE, "de-interleave launch failed");
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_synchronize(rt_ctx *c)
try {
    if (!c) return fail(RT_E_INVALID, "null context");
    DeviceGuard g(c->device);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(RT_E_DEVICE, hipGetErrorString(e));
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_get_stats_sized(rt_ctx *c, rt_stats *out, uint64_t size)
try {
    if (!c || !out) return fail(RT_E_INVALID, "null argument");
    if (size == 0 || size % 4 != 0) return fail(RT_E_INVALID, "rt_get_stats_sized: size must be a positive multiple of 4");
    rt_stats full;
    const int rc = rt_get_stats(c, &full);
    if (rc != RT_OK) return rc;
    memcpy(out, &full, size < sizeof full ? (size_t)size : sizeof full);
    return RT_OK;
} RT_ABI_CATCH_INT

int rt_get_stats(rt_ctx *c, rt_stats *out)
try {
    if (!c || !out) return fail(RT_E_INVALID, "null argument");
    DeviceGuard g(c->device);
    memset(out, 0, sizeof *out);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(RT_E_DEVICE, hipGetErrorString(e));
    out->n_analytic = c->n_analytic;
    out->n_triangles = c->n_tris;
    out->bvh_nodes = c->bvh.n_nodes;
    out->bvh_depth = c->bvh.depth;
    out->bvh_leaf_max = c->bvh.leaf_max;
    out->ms_build = c->bvh.ms_build;
    out->bvh_sah_cost = c->bvh.sah_cost;
    out->bvh_refs = c->bvh.n_refs;
    if (!c->f->frame_valid) return RT_OK;
    FrameCounters fc;
    if (hipMemcpy(&fc, c->f->last_counters ? c->f->last_counters : c->f->ws.counters, sizeof fc, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(RT_E_DEVICE, "read frame counters");
    if (fc.guard_hits)      // only a guarded build (-DRT_GUARD=1) ever counts these
        return fail(RT_E_DEVICE, "guarded build: " + std::to_string(fc.guard_hits) + " out-of-range indices were caught in the last frame (the first: index " +
                                     std::to_string(fc.guard_value) + " of " + std::to_string(fc.guard_limit) + " at source line " + std::to_string(fc.guard_line) +
                                     "; line 0 = a traversal loop ran into its iteration bound)");
    out->rays_primary = c->f->last_primary;
    out->rays_shadow = fc.rays_shadow;
    out->rays_reflect = fc.rays_reflect;
    out->box_tests = fc.box_tests;
    out->box_tests_occupied = fc.box_slots;
    out->staging_ms = c->f->last_staging_ms;
    out->tri_tests = fc.tri_tests;
    out->analytic_tests = fc.analytic_tests;
    if (c->opt_count && getenv("RT_WF_DEBUG") && fc.node_iters)
        fprintf(stderr, "[rt] traversal: node steps %llu (%.1f lanes; %.1f without a ray, %.1f waiting at a leaf), leaf steps %llu (%.1f lanes), refills %llu (%.1f rays)\n",
                fc.node_iters, (double)fc.node_lanes / (double)fc.node_iters, (double)fc.node_idle / (double)fc.node_iters,
                (double)fc.node_leafwait / (double)fc.node_iters, fc.leaf_iters,
                fc.leaf_iters ? (double)fc.leaf_lanes / (double)fc.leaf_iters : 0.0, fc.refill_events,
                fc.refill_events ? (double)fc.refill_lanes / (double)fc.refill_events : 0.0);
    if (c->opt_count && getenv("RT_WF_DEBUG") && fc.node_iters) {
        // how alike the lanes of a wave are during the node steps (FrameCounters::coh)
        static const char *kinds[3] = { "camera", "shadow", "reflection" };
        fprintf(stderr, "[rt] box tests on occupied child slots: %llu of %llu charged\n", fc.box_slots, fc.box_tests);
        for (int k = 0; k < 3; k++) {
            unsigned long long tot[4] = { 0, 0, 0, 0 };
            for (int b = 0; b < 16; b++)
                for (int j = 0; j < 4; j++) tot[j] += fc.coh[k][b][j];
            if (!tot[0]) continue;
            fprintf(stderr, "[rt] coherence %-10s lane-steps %llu: lanes of the wave at the same node %.2f, at the first lane's node %.1f %%, in the first lane's octant %.1f %%\n",
                    kinds[k], tot[0], (double)tot[1] / (double)tot[0], 100.0 * (double)tot[2] / (double)tot[0], 100.0 * (double)tot[3] / (double)tot[0]);
            fprintf(stderr, "[rt]   by the ray's node-step index (share of lane-steps %% | same-node lanes | first lane's node %% | octant %%):\n");
            for (int b = 0; b < 16; b++) {
                const unsigned long long *q = fc.coh[k][b];
                if (!q[0]) continue;
                fprintf(stderr, "[rt]   %2d%s %5.1f | %5.2f | %5.1f | %5.1f\n", b, b == 15 ? "+" : " ", 100.0 * (double)q[0] / (double)tot[0],
                        (double)q[1] / (double)q[0], 100.0 * (double)q[2] / (double)q[0], 100.0 * (double)q[3] / (double)q[0]);
            }
        }
    }
    if (c->opt_count && getenv("RT_WF_DEBUG") && fc.max_steps) {
        fprintf(stderr, "[rt] steps per ray: max %u; rays by bit length of the step count:", fc.max_steps);
        for (int k = 0; k < 16; k++) fprintf(stderr, " %llu", fc.step_hist[k]);
        fprintf(stderr, "\n");
    }
    out->n_launches = c->f->last_launches;
    out->accel = c->f->last_accel;
    const int ne = c->f->ws.n_events;
    if (c->f->last_timed && ne >= 3 && (int)c->f->ws.events.size() >= ne) {
        // events: [start] [after each traversal stage] ... [after the final fold]
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->f->ws.events[0], c->f->ws.events[(size_t)ne - 2]) == hipSuccess) out->ms_trace = ms;
        if (hipEventElapsedTime(&ms, c->f->ws.events[(size_t)ne - 2], c->f->ws.events[(size_t)ne - 1]) == hipSuccess) out->ms_resolve = ms;
        if (hipEventElapsedTime(&ms, c->f->ws.events[0], c->f->ws.events[(size_t)ne - 1]) == hipSuccess) out->ms_frame = ms;
        out->ms_traverse = out->ms_trace;
        if (c->f->last_wavefront && c->f->ws.trav_pairs > 0 && (int)c->f->ws.trav_events.size() >= 2 * c->f->ws.trav_pairs) {
            float tv = 0.f;
            for (int k = 0; k < c->f->ws.trav_pairs; k++)
                if (hipEventElapsedTime(&ms, c->f->ws.trav_events[2 * (size_t)k], c->f->ws.trav_events[2 * (size_t)k + 1]) == hipSuccess) tv += ms;
            out->ms_traverse = tv;
        }
    }
    return RT_OK;
} RT_ABI_CATCH_INT

}  // extern "C"
