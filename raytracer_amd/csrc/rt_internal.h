// rt_internal.h — device data layout and host-side plumbing shared by the
// translation units of librt_amd.so (rt_shim.hip, trace.hip, wavefront.hip, shade.hip, lbvh.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/rt_shim.h"

// The trace arithmetic must round exactly as the reference's scalar C++ does
// (x86-64 SSE, no fused multiply-add), so contraction is off for every
// expression in these files; the build also passes -ffp-contract=off.
#pragma clang fp contract(off)

// ------------------------------------------------------------------ device records

// Analytic primitive (sphere or plane), 64 B, read wave-uniformly.
struct DevAnalytic {
    float px, py, pz, s;          // sphere: centre, radius | plane: normal, offset
    float s2;                     // sphere: radius*radius (main.cpp:53)
    int32_t kind, is_light, pad0;
    float reflect, diffuse, spec, pad1;
    float cr, cg, cb, pad2;
};

// Material table entry, 32 B.
struct DevMaterial {
    float reflect, diffuse, spec, pad0;
    float cr, cg, cb, pad1;
};

// LBVH traversal node, 64 B: both children's boxes live in the parent so one
// 64-byte fetch decides both.  child >= 0: node index; child < 0: leaf,
// ~child = first_sorted_triangle * 8 + (count - 1).
struct BvhNode {
    float4 xy0;   // child0: lo.x, hi.x, lo.y, hi.y
    float4 xy1;   // child1: lo.x, hi.x, lo.y, hi.y
    float4 z01;   // child0: lo.z, hi.z ; child1: lo.z, hi.z
    int4 ch;      // child0, child1, unused, unused
};

// Four children per node, 64 B, boxes rounded outward onto a 15-bit lattice over the root box (each 16-bit half of a
// word is 0x8000 | q; world = base + q * step per axis, QuantGrid): one dependent fetch where the binary tree needs two.
// An unused slot holds an inverted box (never entered).
struct BvhNode4 {
    uint4 a;      // child0 x, y, z, child1 x          (each word lo | hi << 16)
    uint4 b;      // child1 y, z, child2 x, y
    uint4 c;      // child2 z, child3 x, y, z
    uint4 d;      // child references 0..3 (inner: wide node index; negative: leaf, as in BvhNode)
};

// The same four children as floats, 128 B, for the packet kernel: a wave that walks one node for all its rays loads
// it through the scalar cache straight into scalar registers, where nothing is left to decode.  Every plane holds
// f = 1 + q / 32768 of its lattice coordinate q, the value the packed node's 16-bit halves decode to, so both
// node forms give bit-identical box tests with the same per-ray constants (slab_ray).
struct BvhNode4F {
    float lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4];
    int ref[4];
    int pad[4];
};

// The first WIDE_TOP_NODES entries of the wide node array are the top WIDE_TOP_LEVELS levels as a complete
// 4-ary tree (unused slots zero); the traversal kernel copies them into LDS.
constexpr int WIDE_TOP_LEVELS = 4;
constexpr int WIDE_TOP_NODES = 1 + 4 + 16 + 64;

struct QuantGrid {
    float base[3], step[3];
};

// Everything a trace kernel needs about the scene (passed by value).
struct DevScene {
    const DevAnalytic *analytic;   // [n_analytic], scene order
    const int32_t *lights;         // [n_lights] analytic ids with is_light, scene order
    const float4 *tri_mt;          // [n_tris*3] original order: (v1,·) (e1,·) (e2,·)
    const float4 *tri_n;           // [n_tris] (N.xyz, material index as int bits)
    const float4 *tri_vn;          // [n_tris*3] shading normals at v1, v2, v3, or null: geometric normals (the reference)
    const float2 *tri_vt;          // [n_tris*3] texture coordinates at v1, v2, v3, or null (only the uv plane reads them)
    const DevMaterial *mats;       // [n_mats]
    const BvhNode *nodes;          // LBVH, root = 0
    const BvhNode4 *wnodes;        // the tree collapsed to four children per node
    const BvhNode4F *fnodes;       // the same nodes with float planes (same indices)
    QuantGrid grid;
    const float4 *bvh_tri;         // [n_tris*3] Morton order: (v1, original tri index bits) (e1,·) (e2,·)
    int32_t n_analytic, n_lights, n_tris, accel;
    int32_t n_refs;                // entries of bvh_tri: triangle references (>= n_tris when large triangles were cut)
};

struct DevCamera {
    float ox, oy, oz;
    float pitch, off_x, off_y, plane_z;
};

// Per-frame device counters (zeroed by one hipMemsetAsync per frame).
struct FrameCounters {
    unsigned int queue_count[RT_MAX_DEPTH + 2];   // rays queued for each level
    unsigned int cursor[RT_MAX_DEPTH + 2];        // work-fetch cursor of each level
    unsigned long long rays_shadow, rays_reflect;
    unsigned long long box_tests, tri_tests, analytic_tests;
    // RT_OPT_COUNT_TESTS only: wave iterations of each traversal step and the lanes that took part
    unsigned long long node_iters, node_lanes, leaf_iters, leaf_lanes, refill_events, refill_lanes;
    unsigned long long node_idle, node_leafwait;   // lanes that sat out the node steps: without a ray / waiting at a leaf
    unsigned long long step_hist[16];             // rays by traversal steps taken: bin = bit length of the step count
    unsigned int max_steps, pad_steps;
    unsigned int guard_hits, guard_line;         // RT_GUARD builds: indices found outside their array (0 otherwise); source line of one of them
    // RT_OPT_COUNT_TESTS only.  box_slots: box tests on child slots that hold a child (box_tests charges four per node
    // step).  coh[kind][bin][k]: how alike the lanes of a wave are during the node steps, by ray kind (0 camera,
    // 1 shadow, 2 reflection) and by the ray's own node-step index (bin 15 = fifteenth step and later): k = 0 lanes at a
    // node, 1 the sum over those lanes of how many lanes of the wave (same kind) stand at the SAME node, 2 lanes
    // at the node the first such lane holds, 3 lanes whose direction-sign octant is the first such lane's.
    unsigned long long box_slots;
    unsigned long long coh[3][16][4];
    unsigned long long guard_value, guard_limit; // RT_GUARD builds: the index caught at guard_line and the size of its array
};

// ------------------------------------------------------------------ host-side state

struct LbvhTopology;   // lbvh.hip: what the build keeps so that lbvh_refit can follow moving vertices

struct LbvhResult {
    LbvhTopology *topo = nullptr;
    BvhNode *nodes = nullptr;      // [max(n_tris-1,1)]
    BvhNode4 *wnodes = nullptr;    // [WIDE_TOP_NODES + max(n_tris-1,1)] (about a quarter used)
    BvhNode4F *fnodes = nullptr;   // same entries, float planes
    QuantGrid grid = {};
    float4 *bvh_tri = nullptr;     // [n_tris*3]
    int n_nodes = 0, depth = 0, wide_depth = 0, leaf_max = 0, builder = 0;
    int n_refs = 0, split = 0;     // triangle references under the hierarchy; cells per axis a triangle was allowed to be cut into
    float ms_build = 0.f;
    float sah_cost = 0.f;          // surface-area-heuristic cost of the collapsed binary tree (node cost 1, triangle cost 1)
};

// lbvh.hip: builds the LBVH over tri_mt (device) on `stream`.  Returns RT_* code.
int lbvh_build(const float4 *tri_mt, int n_tris, int leaf_max, int builder, int split_cells, hipStream_t stream, LbvhResult *out,
               std::string *err);
void lbvh_free(LbvhResult *r);
// lbvh.hip: the same tree over moved vertices (same triangle count and order as the build).
int lbvh_refit(const float4 *tri_mt, int n_tris, hipStream_t stream, LbvhResult *out, std::string *err);

// lbvh.hip: Triangle constructor (main.cpp:104-107) on the device.
int tri_setup_flat(const float *d_tri_verts, int n_tris, float4 *tri_mt, float4 *tri_n, int mat_index,
                   hipStream_t stream);
int tri_setup_indexed(const float *d_verts, int n_verts, const int32_t *d_indices, int n_tris, float4 *tri_mt,
                      float4 *tri_n, int mat_index, int *d_bad_index_flag, hipStream_t stream);

struct FrameParams {
    int W, H;
    rt_rowset rows;
    int max_depth;
    int count_tests;               // 1: count tests and lane statistics (slow kernels); 2: only record the claim units' costs (track_units), no picture
    int timing;
    int stack_depth;               // LBVH depth bound used to pick the traversal variant
    int wide_depth;                // depth of the 4-wide tree
    int pipeline;                  // 0 auto, 1 fused bounce kernel, 2 wavefront stages
    int track_units;               // staged pipeline: record every claim unit's longest ray (with count_tests; FrameWorkspace::unit_cost)
    int use_order;                 // staged pipeline: claim the units in the order of FrameWorkspace::unit_order
    int camera_packet;             // staged pipeline: 1 = the camera rays go through the packet kernel (one wave walks the tree for a tile)
    int camera_only;               // staged pipeline: enqueue the camera rays' traversal launch and nothing else (timing runs)
    float *d_normal;               // optional device plane: the primary hit's normal, 3 floats per pixel (0 on a miss)
    float *d_uv;                   // optional device plane: the primary hit's texture coordinate, 2 floats per pixel
};

struct FrameWorkspace {
    float4 *records = nullptr;     // [(max_depth+1)][n_pixels]: (local colour, hit id bits)
    float *hit_t = nullptr;        // [n_pixels] level-0 nearest t
    float4 *queue_o[2] = { nullptr, nullptr };   // ray queues: (origin, pixel index bits)
    float4 *queue_d[2] = { nullptr, nullptr };   //             (direction, unused)
    FrameCounters *counters = nullptr;   // head of the block that one memset clears per frame; the claim cursors follow it
    size_t zero_cap = 0;                 // bytes allocated in that block
    size_t pixels_cap = 0;
    int levels_cap = 0;
    // wavefront pipeline (LBVH scenes)
    float4 *sq_o = nullptr;   // [seg_cap][slots_cap] shadow rays (hit point, pixel), one segment per light of a chunk
    unsigned int *rq_counts[2] = { nullptr, nullptr };   // [regions] rays per region of each ray queue
    unsigned int *fq_counts[2] = { nullptr, nullptr };   // fused kernel: rays per 64-slot batch of each ray queue
    unsigned int *sq_counts = nullptr;         // [seg_cap][regions] shadow rays per segment and region
    unsigned int *trace_cursors = nullptr;     // per-frame unit claim cursors, one row per k_trace launch (inside the counters' block)
    int *overflow = nullptr;       // traversal stack spill columns
    size_t slots_cap = 0, overflow_cap = 0;
    int chunks_cap = 0, seg_cap = 0;
    // Claim order of the traversal launches.  A launch ends with its longest rays; when those are claimed last, the chip
    // idles behind them.  unit_cost[launch][unit] = steps of the longest ray of each 64-ray claim unit, recorded by a
    // counting frame; unit_order[launch][k] = the unit claimed k-th: longest first.  A frame with the same layout (frame
    // size, rows, depth, lights) claims in that order, whatever moved in between: the order only decides when a ray
    // is traced, never what it hits.
    unsigned int *unit_cost = nullptr, *unit_order = nullptr;
    size_t unit_stride = 0;        // entries per launch
    int unit_launches = 0;         // launches the arrays have room for
    bool order_valid = false;
    int camera_packet = 0;         // which camera-ray kernel was faster for this layout when the order was made
    std::vector<unsigned char> order_key;   // the frame layout unit_order was made for
    std::vector<unsigned char> seen_key;    // the layout of the previous frame (an order is only made for a layout that repeats)
    // hit planes of the LBVH pipelines
    float2 *p_hits = nullptr;      // [levels][pixels] nearest (t, id bits)
    unsigned int *p_occ = nullptr; // [levels][chunks][pixels] occlusion bit per light, 32 lights per chunk
    size_t p_cap = 0;              // elements allocated in p_hits / p_occ
    size_t records_cap = 0;        // float4 elements allocated in records
    // staging outputs for the host-buffer entry point
    uint8_t *out_rgb8 = nullptr; int32_t *out_id = nullptr; float *out_rgb32 = nullptr; float *out_normal = nullptr; float *out_uv = nullptr;
    size_t out_cap = 0;
    std::vector<hipEvent_t> events;        // frame start, after each traversal stage, after the final fold
    int n_events = 0;                      // events recorded by the last frame
    std::vector<hipEvent_t> trav_events;   // begin/end pairs around the traversal launches of the last frame
    int trav_pairs = 0;
};

// Bytes from the frame counters to the claim cursors that share their allocation (and their per-frame memset).
constexpr size_t COUNTERS_PAD = (sizeof(FrameCounters) + 255) & ~(size_t)255;

// trace.hip: enqueue one frame.  Outputs are device pointers (may be null).
int trace_frame(const DevScene &scene, const DevCamera &cam, const FrameParams &fp, FrameWorkspace &ws,
                void *d_rgb8, void *d_hit_id, void *d_rgb32, void *d_hit_t, hipStream_t stream,
                int *n_launches, std::string *err);
// wavefront.hip: the LBVH pipeline of one frame's bounce levels (records only; trace_frame resolves).
int wavefront_trace_grid(int num_cus);
size_t wavefront_overflow_ints(int num_cus);
size_t wavefront_regions(size_t slots);
size_t wavefront_cursor_words(int levels, int n_chunks);
size_t wavefront_unit_stride(size_t slots, int n_lights);   // claim units of the largest traversal launch of a frame
int wavefront_unit_launches(int levels, int n_chunks);      // traversal launches a frame may have
bool wavefront_packet_possible(const DevScene &scene, const FrameParams &fp);   // the packet kernel can walk this hierarchy
int wavefront_frame(const DevScene &scene, const DevCamera &cam, const FrameParams &fp, FrameWorkspace &ws, void *d_rgb8,
                    void *d_hit_id, void *d_rgb32, void *d_hit_t, hipStream_t stream, int num_cus,
                    const std::function<int()> &stamp, int *n_launches, std::string *err);
// shade.hip: local colour of every level, fold and outputs from the [level][pixel] hit and occlusion planes
int shade_all_launch(const DevScene &scene, const DevCamera &cam, const FrameParams &fp, const float2 *hits,
                     const unsigned int *occ, int n_chunks, float4 *records, FrameCounters *counters, void *d_rgb8, void *d_hit_id,
                     void *d_rgb32, void *d_hit_t, hipStream_t stream, int num_cus, std::string *err);
// Which LBVH pipeline a frame uses: 1 fused (one kernel per bounce level), 2 staged (traversal / generation / shading).
inline int lbvh_pipeline(int option, int /*n_lights*/) { return option == 1 ? 1 : 2; }
// Levels up to which the staged pipeline's shading keeps a pixel's level colours in registers (no records plane).
constexpr int SHADE_REG_LEVELS = 9;
int device_num_cus(std::string *err);   // trace.hip: CU count of the current device (cached), < 0 on error
int trace_deinterleave(const void *src, void *dst, int W, int H, int band_rows, int n_parts, int rows_per_part,
                       int bpp, hipStream_t stream);

#define RT_HIP_CHECK(expr, errp)                                                              \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            if (errp) *(errp) = std::string(#expr) + ": " + hipGetErrorString(e__);           \
            return RT_E_DEVICE;                                                               \
        }                                                                                     \
    } while (0)
