// lbvh.hip — on-device triangle set-up and LBVH construction for gfx950.
//
// The reference has no acceleration structure: raytrace() scans every primitive
// (Raytracer/main.cpp:149-156, 181-188).  The LBVH built here only PRUNES that
// scan.  Every triangle a ray reaches still goes through the reference's
// Möller–Trumbore arithmetic (main.cpp:109-129) on the same v1/e1/e2 values the
// reference's constructor computes (main.cpp:104-107), so hit ids and distances
// stay bit-equal as long as no padded box culls a triangle the test would accept.
//
// Pipeline (all kernels, no host geometry work):
//   k_tri_setup_*   Triangle constructor: v1, e1, e2, N
//   k_bounds        scene bounds (wave + block reduce, ordered-int atomics)
//   k_split_*       triangle references: a large triangle is cut by a grid into references with clipped boxes
//   binary tree over the references, one of
//     k_sah_*       top-down binned surface-area-heuristic sweep (the default), checked by k_tree_check
//     k_ploc_*      agglomeration over the Morton order (k_morton + 8-pass radix sort first)
//     k_karras      Karras 2012 radix tree from the sorted Morton codes
//   k_node_depth    depth of every internal node (parent walk)
//   k_refit_level   bottom-up box union, one launch per depth (no in-kernel hand-offs)
//   k_emit          64-byte float nodes with both children's padded boxes (fused pipeline); subtrees
//                   of <= leaf_max references collapse into one leaf
//   k_wide_*, k_emit_wide   the tree collapsed to four children per node on a 15-bit lattice (staged pipeline)
//   k_gather_tris   triangles in leaf order
#include "rt_internal.h"

#include <float.h>

#include <algorithm>

#include <new>

namespace {

constexpr int SORT_TILE = 1024;   // keys per wave-sized block in the radix sort

__device__ inline unsigned int float_to_ordered(float f)
{
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ordered_to_float(unsigned int u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// ------------------------------------------------------------------ Triangle constructor (main.cpp:104-107)
__device__ inline void tri_ctor(float ax, float ay, float az, float bx, float by, float bz, float cx, float cy,
                                float cz, int i, int mat_index, float4 *tri_mt, float4 *tri_n)
{
    // e1 = v2 - v1, e2 = v3 - v1                                   (maths.h:132)
    float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    // N = e1.cross(e2).normalize()                                 (maths.h:165-172, 158-162)
    float nx = e1y * e2z - e1z * e2y;
    float ny = e1z * e2x - e1x * e2z;
    float nz = e1x * e2y - e1y * e2x;
    float len = sqrtf(nx * nx + ny * ny + nz * nz);
    tri_mt[3 * (size_t)i + 0] = make_float4(ax, ay, az, 0.f);
    tri_mt[3 * (size_t)i + 1] = make_float4(e1x, e1y, e1z, 0.f);
    tri_mt[3 * (size_t)i + 2] = make_float4(e2x, e2y, e2z, 0.f);
    const float w = mat_index < 0 ? tri_n[i].w : __int_as_float(mat_index);   // < 0: keep the stored material index
    tri_n[i] = make_float4(nx / len, ny / len, nz / len, w);
}

__global__ void k_tri_setup_flat(const float *__restrict__ v, int n, float4 *tri_mt, float4 *tri_n, int mat_index)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = v + 9 * (size_t)i;
    tri_ctor(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], i, mat_index, tri_mt, tri_n);
}

// main.cpp:228-231: new Triangle(verts[inds[i]], verts[inds[i+1]], verts[inds[i+2]])
__global__ void k_tri_setup_indexed(const float *__restrict__ verts, int n_verts, const int32_t *__restrict__ idx,
                                    int n, float4 *tri_mt, float4 *tri_n, int mat_index, int *bad)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int i0 = idx[3 * (size_t)i], i1 = idx[3 * (size_t)i + 1], i2 = idx[3 * (size_t)i + 2];
    if ((unsigned)i0 >= (unsigned)n_verts || (unsigned)i1 >= (unsigned)n_verts || (unsigned)i2 >= (unsigned)n_verts) {
        *bad = 1;   // the reference would read out of bounds here; reported as RT_E_INVALID
        i0 = i1 = i2 = 0;
    }
    const float *a = verts + 3 * (size_t)i0, *b = verts + 3 * (size_t)i1, *c = verts + 3 * (size_t)i2;
    tri_ctor(a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], i, mat_index, tri_mt, tri_n);
}

// ------------------------------------------------------------------ bounds + Morton codes
// The box of a triangle is the box of the three points the intersector actually
// sees: v1, v1+e1, v1+e2.
// A triangle with a NaN or an infinity among the values the intersector reads (v1, e1, e2) can never be hit (its
// Moller-Trumbore t comes out as NaN or 0, never > 1e-4), but it would poison the scene bounds, the Morton codes and
// every box union above it.  For the builder it is a point at the origin; the intersector still sees the caller's values.
__device__ inline bool tri_corners(const float4 *__restrict__ tri_mt, int i, float v[3][3])
{
    const float4 a = tri_mt[3 * (size_t)i], e1 = tri_mt[3 * (size_t)i + 1], e2 = tri_mt[3 * (size_t)i + 2];
    v[0][0] = a.x; v[0][1] = a.y; v[0][2] = a.z;
    v[1][0] = a.x + e1.x; v[1][1] = a.y + e1.y; v[1][2] = a.z + e1.z;
    v[2][0] = a.x + e2.x; v[2][1] = a.y + e2.y; v[2][2] = a.z + e2.z;
    const float w[9] = { a.x, a.y, a.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z };
    bool finite = true;
    for (int k = 0; k < 9; k++) finite = finite && fabsf(w[k]) <= FLT_MAX;
    if (!finite)
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 3; k++) v[c][k] = 0.f;
    return finite;
}

__device__ inline void tri_box(const float4 *__restrict__ tri_mt, int i, float lo[3], float hi[3])
{
    float v[3][3];
    (void)tri_corners(tri_mt, i, v);
    for (int k = 0; k < 3; k++) {
        lo[k] = fminf(v[0][k], fminf(v[1][k], v[2][k]));
        hi[k] = fmaxf(v[0][k], fmaxf(v[1][k], v[2][k]));
    }
}

// Grid-stride: every thread folds many triangles, a wave reduces by shuffles, the block by LDS, and one lane per
// block issues the six atomics (atomics to one cache line retire at about 90 per microsecond on this chip: one set
// per wave of triangles made this kernel half of a refit).
__global__ __launch_bounds__(256) void k_bounds(const float4 *__restrict__ tri_mt, int n, unsigned int *bounds /*[6] ordered: lo xyz, hi xyz*/)
{
    __shared__ float s_lo[4][3], s_hi[4][3];
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float l[3], h[3];
        tri_box(tri_mt, i, l, h);
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], l[k]); hi[k] = fmaxf(hi[k], h[k]); }
    }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; k++) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = s_lo[0][k], b = s_hi[0][k];
        for (int w = 1; w < 4; w++) { a = fminf(a, s_lo[w][k]); b = fmaxf(b, s_hi[w][k]); }
        atomicMin(&bounds[k], float_to_ordered(a));
        atomicMax(&bounds[3 + k], float_to_ordered(b));
    }
}

// ------------------------------------------------------------------ triangle references (split large triangles)
// A hierarchy over whole-triangle boxes handles one shape badly: a triangle that is long or large against its
// neighbours, and not aligned with an axis, has a box that is mostly empty space, and every ray through that space
// pays a triangle test.  The hierarchy is therefore built over REFERENCES: a triangle whose box is longer than
// 1/SPLIT_RES of the scene is cut by a grid of cells along its two longest box axes (at most `max_cells` per axis),
// and every cell the triangle reaches becomes one reference whose box is the box of the triangle clipped to the cell.
// A reference still points to the whole triangle and the intersector runs on the whole triangle
// (Raytracer/main.cpp:109-129), so hit ids and distances do not change: a triangle found through two references gives
// the same (t, id) twice.  Ordinary triangles get exactly one reference with their own box.
#ifndef LBVH_SPLIT_RES
#define LBVH_SPLIT_RES 64.0f
#endif
constexpr float SPLIT_RES = LBVH_SPLIT_RES;

struct SplitGrid {
    int n[3];              // cells per axis (1 = axis not cut)
    float lo[3], hi[3];    // the triangle's box
    float v[3][3];         // the three corners the intersector sees: v1, v1+e1, v1+e2
};

__device__ inline SplitGrid split_grid(const float4 *__restrict__ tri_mt, int i, const unsigned int *__restrict__ bounds, int max_cells)
{
    SplitGrid g;
    (void)tri_corners(tri_mt, i, g.v);
    float ext = 0.f;
    for (int k = 0; k < 3; k++) ext = fmaxf(ext, ordered_to_float(bounds[3 + k]) - ordered_to_float(bounds[k]));
    const float h = ext / SPLIT_RES;
    int smallest = 0;
    for (int k = 0; k < 3; k++) {
        g.lo[k] = fminf(g.v[0][k], fminf(g.v[1][k], g.v[2][k]));
        g.hi[k] = fmaxf(g.v[0][k], fmaxf(g.v[1][k], g.v[2][k]));
        const float e = g.hi[k] - g.lo[k];
        g.n[k] = 1;
        if (max_cells > 1 && h > 0.f && e > h && e < FLT_MAX) g.n[k] = min(max_cells, (int)ceilf(e / h));     // (a NaN or infinite extent is never cut)
        if (g.n[k] < g.n[smallest]) smallest = k;
    }
    g.n[smallest] = 1;     // a triangle is flat: two axes of cells reach every part of it
    return g;
}

// Boundary j of axis k's cells; the same expression from both sides, so neighbouring cells share it exactly.
__device__ inline float split_bound(const SplitGrid &g, int k, int j)
{
    return j <= 0 ? g.lo[k] : j >= g.n[k] ? g.hi[k] : g.lo[k] + (g.hi[k] - g.lo[k]) * ((float)j / (float)g.n[k]);
}

// Sutherland-Hodgman against one axis plane, closed half-space (points on the plane are kept).
__device__ inline int clip_plane(const float (*p)[3], int np, int ax, float bound, bool keep_ge, float (*q)[3])
{
    int nq = 0;
    for (int i = 0; i < np; i++) {
        const float *a = p[i], *b = p[i + 1 == np ? 0 : i + 1];
        const bool ia = keep_ge ? a[ax] >= bound : a[ax] <= bound, ib = keep_ge ? b[ax] >= bound : b[ax] <= bound;
        if (ia) { q[nq][0] = a[0]; q[nq][1] = a[1]; q[nq][2] = a[2]; nq++; }
        if (ia != ib) {
            const float t = (bound - a[ax]) / (b[ax] - a[ax]);
            for (int k = 0; k < 3; k++) q[nq][k] = k == ax ? bound : a[k] + (b[k] - a[k]) * t;
            nq++;
        }
    }
    return nq;
}

// Box of the triangle inside cell (c[0], c[1], c[2]); false when the triangle does not reach the cell.
__device__ inline bool split_cell_box(const SplitGrid &g, const int c[3], float lo[3], float hi[3])
{
    float p[10][3], q[10][3];
    int np = 3;
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) p[i][k] = g.v[i][k];
    for (int k = 0; k < 3; k++) {
        if (g.n[k] == 1) continue;
        np = clip_plane(p, np, k, split_bound(g, k, c[k]), true, q);
        if (np == 0) return false;
        np = clip_plane(q, np, k, split_bound(g, k, c[k] + 1), false, p);
        if (np == 0) return false;
    }
    for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
    for (int i = 0; i < np; i++)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], p[i][k]); hi[k] = fmaxf(hi[k], p[i][k]); }
    for (int k = 0; k < 3; k++) {        // never outside the triangle's own box or the cell
        lo[k] = fmaxf(lo[k], g.n[k] > 1 ? split_bound(g, k, c[k]) : g.lo[k]);
        hi[k] = fminf(hi[k], g.n[k] > 1 ? split_bound(g, k, c[k] + 1) : g.hi[k]);
    }
    return true;
}

// counts[i] = cells of triangle i's grid (1 for a triangle that is not cut)
__global__ void k_split_count(const float4 *__restrict__ tri_mt, int n, const unsigned int *__restrict__ bounds, int max_cells,
                              unsigned int *counts)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SplitGrid g = split_grid(tri_mt, i, bounds, max_cells);
    counts[i] = (unsigned)(g.n[0] * g.n[1] * g.n[2]);
}

// Sum of the cell counts in 64 bits (the 32-bit scan that follows must not wrap): wave and block reduce, one atomic per block.
__global__ __launch_bounds__(256) void k_sum_counts(const unsigned int *__restrict__ counts, int n, unsigned long long *total)
{
    __shared__ unsigned long long s_sum[4];
    unsigned long long v = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v += counts[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

// One thread per cell of every triangle's grid (cell_off = exclusive scan of the counts, n_cells their sum): the box of
// the triangle inside the cell and whether the triangle reaches it at all.  A triangle's corner always lies in one of
// its (closed) cells, so every triangle keeps at least one reference.
__global__ void k_split_cells(const float4 *__restrict__ tri_mt, int n, const unsigned int *__restrict__ bounds, int max_cells,
                              const unsigned int *__restrict__ cell_off, unsigned int n_cells, unsigned int *cell_tri, float4 *cell_lo,
                              float4 *cell_hi, unsigned int *flags)
{
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_cells) return;
    // the triangle whose range of cells holds idx: the last i with cell_off[i] <= idx
    int lo_i = 0, hi_i = n - 1;
    while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i + 1) >> 1;
        if (cell_off[mid] <= idx) lo_i = mid; else hi_i = mid - 1;
    }
    const int i = lo_i;
    const SplitGrid g = split_grid(tri_mt, i, bounds, max_cells);
    const int j = (int)(idx - cell_off[i]);
    float lo[3] = { g.lo[0], g.lo[1], g.lo[2] }, hi[3] = { g.hi[0], g.hi[1], g.hi[2] };
    bool keep = true;
    if (g.n[0] * g.n[1] * g.n[2] > 1) {
        const int c[3] = { j % g.n[0], (j / g.n[0]) % g.n[1], j / (g.n[0] * g.n[1]) };
        keep = split_cell_box(g, c, lo, hi);
    }
    cell_tri[idx] = (unsigned)i;
    cell_lo[idx] = make_float4(lo[0], lo[1], lo[2], 0.f);
    cell_hi[idx] = make_float4(hi[0], hi[1], hi[2], 0.f);
    flags[idx] = keep ? 1u : 0u;
}

// The cells a triangle reaches, in order, become the references (dst = exclusive scan of the flags).
__global__ void k_split_compact(const unsigned int *__restrict__ cell_tri, const float4 *__restrict__ cell_lo, const float4 *__restrict__ cell_hi,
                                const unsigned int *__restrict__ dst, unsigned int n_cells, unsigned int m, unsigned int *ref_tri, float4 *ref_lo,
                                float4 *ref_hi)
{
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_cells) return;
    const unsigned int at = dst[idx], next = idx + 1 < n_cells ? dst[idx + 1] : m;
    if (next == at || at >= m) return;          // flag was 0
    ref_tri[at] = cell_tri[idx];
    ref_lo[at] = cell_lo[idx];
    ref_hi[at] = cell_hi[idx];
}

__global__ void k_ref_identity(int n, unsigned int *ref_tri)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) ref_tri[r] = (unsigned)r;
}

// The references of a refit: the triangles moved, so every reference falls back to its whole triangle's box.
__global__ void k_ref_boxes_whole(const float4 *__restrict__ tri_mt, const unsigned int *__restrict__ ref_tri, int m, float4 *ref_lo, float4 *ref_hi)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    float lo[3], hi[3];
    tri_box(tri_mt, (int)ref_tri[r], lo, hi);
    ref_lo[r] = make_float4(lo[0], lo[1], lo[2], 0.f);
    ref_hi[r] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

__device__ inline unsigned long long expand21(unsigned long long v)
{
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ void k_morton(const float4 *__restrict__ ref_lo, const float4 *__restrict__ ref_hi, int n, const unsigned int *__restrict__ bounds,
                         unsigned long long *keys, unsigned int *vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 l4 = ref_lo[i], h4 = ref_hi[i];
    const float lo[3] = { l4.x, l4.y, l4.z }, hi[3] = { h4.x, h4.y, h4.z };
    unsigned long long code = 0;
    // one cell size for all three axes (the longest scene extent): cubic cells split better than
    // cells stretched to each axis' own extent
    float ext = 0.f;
    for (int k = 0; k < 3; k++) ext = fmaxf(ext, ordered_to_float(bounds[3 + k]) - ordered_to_float(bounds[k]));
    for (int k = 0; k < 3; k++) {
        float slo = ordered_to_float(bounds[k]);
        float c = 0.5f * (lo[k] + hi[k]);
        float u = ext > 0.f ? (c - slo) / ext : 0.f;
        float q = fminf(fmaxf(u * 2097152.0f, 0.f), 2097151.0f);
        code |= expand21((unsigned long long)q) << (2 - k);
    }
    keys[i] = code;
    vals[i] = (unsigned)i;
}

// ------------------------------------------------------------------ radix sort (LSD, 8 bits per pass, stable)
// One 64-lane block per tile of SORT_TILE keys.  hist layout: [digit][tile] so an
// exclusive scan of the flat array yields global scatter bases.
__global__ __launch_bounds__(64) void k_sort_hist(const unsigned long long *__restrict__ keys, int n, int shift,
                                                  unsigned int *hist, int n_tiles)
{
    __shared__ unsigned int h[256];
    int lane = threadIdx.x, tile = blockIdx.x;
    for (int d = lane; d < 256; d += 64) h[d] = 0;
    __syncthreads();
    int base = tile * SORT_TILE;
    for (int r = 0; r < SORT_TILE / 64; r++) {
        int i = base + r * 64 + lane;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = lane; d < 256; d += 64) hist[(size_t)d * n_tiles + tile] = h[d];
}

// Exclusive scan of `m` counters by a single 1024-thread block (m is at most a few million).
// total (optional) receives the sum of all inputs.
template <typename T>
__global__ __launch_bounds__(1024) void k_scan_exclusive(T *data, int m, T *total)
{
    __shared__ T wave_sum[16];
    __shared__ T carry_s;
    int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 1024 * 4) {
        int i0 = base + tid * 4;
        T v[4], s = 0;
        for (int k = 0; k < 4; k++) { v[k] = (i0 + k < m) ? data[i0 + k] : (T)0; s += v[k]; }
        T incl = s;
        for (int off = 1; off < 64; off <<= 1) {
            T t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wave_sum[w] = incl;
        __syncthreads();
        T wprefix = 0;
        for (int k = 0; k < w; k++) wprefix += wave_sum[k];
        T carry = carry_s;
        T excl = carry + wprefix + incl - s;
        for (int k = 0; k < 4; k++) { if (i0 + k < m) data[i0 + k] = excl; excl += v[k]; }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wprefix + incl;
        __syncthreads();
    }
    if (total && tid == 0) *total = carry_s;
}

// The same scan over many blocks, for inputs where one block would take hundreds of microseconds:
//   k_scan_partials   sum of every 4096-element chunk
//   k_scan_exclusive  (one block) scan of the chunk sums, grand total
//   k_scan_chunks     scan inside every chunk, offset by its chunk's prefix
constexpr int SCAN_CHUNK = 4096;

template <typename T>
__global__ __launch_bounds__(1024) void k_scan_partials(const T *__restrict__ data, int m, T *partial)
{
    __shared__ T wave_sum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i0 = blockIdx.x * SCAN_CHUNK + tid * 4;
    T s = 0;
    for (int k = 0; k < 4; k++) s += (i0 + k < m) ? data[i0 + k] : (T)0;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) wave_sum[w] = s;
    __syncthreads();
    if (tid == 0) {
        T t = 0;
        for (int k = 0; k < 16; k++) t += wave_sum[k];
        partial[blockIdx.x] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void k_scan_chunks(T *data, int m, const T *__restrict__ partial_scanned)
{
    __shared__ T wave_sum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i0 = blockIdx.x * SCAN_CHUNK + tid * 4;
    T v[4], s = 0;
    for (int k = 0; k < 4; k++) { v[k] = (i0 + k < m) ? data[i0 + k] : (T)0; s += v[k]; }
    T incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        T t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sum[w] = incl;
    __syncthreads();
    T wprefix = 0;
    for (int k = 0; k < w; k++) wprefix += wave_sum[k];
    T excl = partial_scanned[blockIdx.x] + wprefix + incl - s;
    for (int k = 0; k < 4; k++) { if (i0 + k < m) data[i0 + k] = excl; excl += v[k]; }
}

__global__ __launch_bounds__(64) void k_sort_scatter(const unsigned long long *__restrict__ keys_in,
                                                     const unsigned int *__restrict__ vals_in, int n, int shift,
                                                     const unsigned int *__restrict__ hist_scanned, int n_tiles,
                                                     unsigned long long *keys_out, unsigned int *vals_out)
{
    __shared__ unsigned int running[256];
    int lane = threadIdx.x, tile = blockIdx.x;
    for (int d = lane; d < 256; d += 64) running[d] = hist_scanned[(size_t)d * n_tiles + tile];
    __syncthreads();
    int base = tile * SORT_TILE;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int r = 0; r < SORT_TILE / 64; r++) {
        int i = base + r * 64 + lane;
        bool valid = i < n;
        unsigned long long key = valid ? keys_in[i] : ~0ull;
        unsigned int digit = (unsigned)(key >> shift) & 255u;
        // lanes holding the same digit (stable rank = position among them)
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < 8; b++) {
            unsigned long long m = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? m : ~m;
        }
        unsigned int rank = __popcll(peers & lt_mask);
        unsigned int start = 0;
        int leader = __ffsll((long long)peers) - 1;
        if (valid && rank == 0) {
            start = running[digit];
            running[digit] = start + (unsigned)__popcll(peers);
        }
        start = __shfl(start, leader < 0 ? 0 : leader);
        if (valid) {
            unsigned int dst = start + rank;
            keys_out[dst] = key;
            vals_out[dst] = vals_in[i];
        }
        __syncthreads();   // one wave per block: orders the LDS updates of consecutive rounds
    }
}

// ------------------------------------------------------------------ Karras 2012
__device__ inline int delta(const unsigned long long *__restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Node ids: internal 0..n-2, leaf k = (n-1)+k.
__global__ void k_karras(const unsigned long long *__restrict__ keys, int n, int2 *children, int2 *range, int *parent)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + (d < 0 ? d : 0);
    int first = i < j ? i : j, last = i < j ? j : i;
    int left = (first == gamma) ? (n - 1) + gamma : gamma;
    int right = (last == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
    children[i] = make_int2(left, right);
    range[i] = make_int2(first, last);
    parent[left] = i;
    parent[right] = i;
    if (i == 0) parent[0] = -1;
}


// ------------------------------------------------------------------ PLOC (parallel locally-ordered clustering)
// Bottom-up agglomeration over the Morton-ordered clusters (Meister & Bittner 2018): every
// cluster looks PLOC_R places to either side for the neighbour whose merged box has the
// smallest surface area; mutual nearest neighbours merge; the cluster array is compacted and
// the search repeats.  The result is markedly closer to a surface-area-heuristic tree than
// the Karras hierarchy (about a quarter fewer box tests per ray on the courtyard scene).
// Node ids as elsewhere: leaves (n-1)+sorted position, inner nodes 0..n-2 with the root = 0
// (inner ids are handed out downward from n-2 in creation order, so the last merge is node 0).
#ifndef PLOC_RADIUS
#define PLOC_RADIUS 16
#endif
constexpr int PLOC_R = PLOC_RADIUS;

__global__ void k_ploc_init(int n, int *cid, int *size)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    cid[j] = (n - 1) + j;
    size[(n - 1) + j] = 1;
}

__global__ __launch_bounds__(256) void k_ploc_nn(const int *__restrict__ cid, int C, const float4 *__restrict__ box_lo,
                                                 const float4 *__restrict__ box_hi, int *nn)
{
    __shared__ float4 s_lo[256 + 2 * PLOC_R], s_hi[256 + 2 * PLOC_R];
    const int base = blockIdx.x * 256 - PLOC_R;
    for (int t = threadIdx.x; t < 256 + 2 * PLOC_R; t += 256) {
        int k = base + t;
        if (k >= 0 && k < C) {
            int id = cid[k];
            s_lo[t] = box_lo[id];
            s_hi[t] = box_hi[id];
        }
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C) return;
    const float4 lo = s_lo[threadIdx.x + PLOC_R], hi = s_hi[threadIdx.x + PLOC_R];
    float best = FLT_MAX;
    int bj = -1;
    for (int dj = -PLOC_R; dj <= PLOC_R; dj++) {          // ascending index + strict '<': lowest index wins ties,
        const int k = i + dj;                             // which guarantees at least one mutual pair per round
        if (dj == 0 || k < 0 || k >= C) continue;
        const float4 l2 = s_lo[threadIdx.x + PLOC_R + dj], h2 = s_hi[threadIdx.x + PLOC_R + dj];
        const float dx = fmaxf(hi.x, h2.x) - fminf(lo.x, l2.x), dy = fmaxf(hi.y, h2.y) - fminf(lo.y, l2.y),
                    dz = fmaxf(hi.z, h2.z) - fminf(lo.z, l2.z);
        const float a = dx * dy + dy * dz + dz * dx;
        if (a < best) { best = a; bj = k; }
    }
    nn[i] = bj;
}

// flags[i]: low word 1 if cluster i survives the round, high word 1 if it is the lower half of a merging pair
__global__ void k_ploc_flags(const int *__restrict__ nn, int C, unsigned long long *flags)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    const bool merge = mutual && i < j, removed = mutual && i > j;
    flags[i] = (removed ? 0ull : 1ull) | (merge ? (1ull << 32) : 0ull);
}

__global__ void k_ploc_apply(const int *__restrict__ cid_in, const int *__restrict__ nn, const unsigned long long *__restrict__ scanned,
                             int C, int n, int created_before, int *cid_out, int2 *children, int *parent, int *size,
                             float4 *box_lo, float4 *box_hi)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    if (mutual && i > j) return;                          // absorbed by its partner
    const unsigned long long sc = scanned[i];
    const int pos = (int)(unsigned int)sc;
    int id = cid_in[i];
    if (mutual) {
        const int a = id, b = cid_in[j];
        const int node = (n - 2) - (created_before + (int)(sc >> 32));
        children[node] = make_int2(a, b);
        parent[a] = node;
        parent[b] = node;
        size[node] = size[a] + size[b];
        const float4 la = box_lo[a], lb = box_lo[b], ha = box_hi[a], hb = box_hi[b];
        box_lo[node] = make_float4(fminf(la.x, lb.x), fminf(la.y, lb.y), fminf(la.z, lb.z), 0.f);
        box_hi[node] = make_float4(fmaxf(ha.x, hb.x), fmaxf(ha.y, hb.y), fmaxf(ha.z, hb.z), 0.f);
        id = node;
    }
    cid_out[pos] = id;
}

// Depth-first leaf order, one tree level per launch: a node's subtree occupies
// [offset, offset + size) of the leaf order; the left child starts at offset, the right after it.
__global__ void k_dfs_level(const int2 *__restrict__ children, const int *__restrict__ depth, const int *__restrict__ size,
                            int n_internal, int level, int *offset)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal || depth[i] != level) return;
    const int2 c = children[i];
    const int off = offset[i];
    offset[c.x] = off;
    offset[c.y] = off + size[c.x];
}

__global__ void k_dfs_finish(const int *__restrict__ offset, const int *__restrict__ size, int n, int2 *range, int *leafpos,
                             int *order)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n - 1) range[i] = make_int2(offset[i], offset[i] + size[i] - 1);
    if (i < n) {
        const int p = offset[(n - 1) + i];
        leafpos[i] = p;
        order[p] = i;
    }
}

__global__ __launch_bounds__(256) void k_node_depth(const int *__restrict__ parent, int n_internal, int *depth, int *max_depth)
{
    int deepest = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_internal; i += gridDim.x * blockDim.x) {
        int d = 0;
        for (int p = parent[i]; p >= 0 && d <= n_internal; p = parent[p]) d++;     // the bound ends the walk on a malformed tree
        depth[i] = d;
        deepest = max(deepest, d);
    }
    for (int off = 32; off > 0; off >>= 1) deepest = max(deepest, __shfl_xor(deepest, off));
    if ((threadIdx.x & 63) == 0) atomicMax(max_depth, deepest);     // one per wave of a fixed grid, not one per node
}

// Leaf boxes: the intersector's triangle, padded so that rounding in the
// Möller–Trumbore test (which can accept a ray passing slightly outside the exact
// triangle) never lets the slab test cull a triangle the reference would hit.
__global__ void k_leaf_boxes(const float4 *__restrict__ ref_lo, const float4 *__restrict__ ref_hi, const unsigned int *__restrict__ sorted_vals, int n,
                             const unsigned int *__restrict__ bounds, float4 *box_lo, float4 *box_hi)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float4 l4 = ref_lo[sorted_vals[k]], h4 = ref_hi[sorted_vals[k]];
    float lo[3] = { l4.x, l4.y, l4.z }, hi[3] = { h4.x, h4.y, h4.z };
    float scene_abs = 0.f;
    for (int a = 0; a < 3; a++)
        scene_abs = fmaxf(scene_abs, fmaxf(fabsf(ordered_to_float(bounds[a])), fabsf(ordered_to_float(bounds[3 + a]))));
    for (int a = 0; a < 3; a++) {
        float pad = 2e-5f * scene_abs + 1e-4f * (hi[a] - lo[a]) + 1e-6f;
        lo[a] -= pad;
        hi[a] += pad;
    }
    box_lo[(n - 1) + k] = make_float4(lo[0], lo[1], lo[2], 0.f);
    box_hi[(n - 1) + k] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

__global__ void k_refit_level(const int2 *__restrict__ children, const int *__restrict__ depth, int n_internal, int level,
                              float4 *box_lo, float4 *box_hi)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal || depth[i] != level) return;
    int2 c = children[i];
    float4 a = box_lo[c.x], b = box_lo[c.y], p = box_hi[c.x], q = box_hi[c.y];
    box_lo[i] = make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), 0.f);
    box_hi[i] = make_float4(fmaxf(p.x, q.x), fmaxf(p.y, q.y), fmaxf(p.z, q.z), 0.f);
}

// Depth-first (preorder) numbering of the nodes that traversal can reach, so that a node's first
// child is the next node in memory (4 lattice nodes share a 128-byte line) and every subtree is
// contiguous.  live[i] = reachable inner nodes in the subtree of i, bottom-up one level per launch;
// pre[i] = position, top-down.
__global__ void k_live_level(const int2 *__restrict__ children, const int *__restrict__ depth, const int2 *__restrict__ range,
                             int n, int leaf_max, int level, int *live)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || depth[i] != level) return;
    const int2 r = range[i];
    if (i != 0 && r.y - r.x + 1 <= leaf_max) { live[i] = 0; return; }
    const int2 c = children[i];
    live[i] = 1 + (c.x < n - 1 ? live[c.x] : 0) + (c.y < n - 1 ? live[c.y] : 0);
}

// Surface-area-heuristic cost of the collapsed binary tree, the usual measure of a hierarchy's quality:
//   (sum over traversed inner nodes of area(node) + sum over leaves of area(leaf) * triangles in it) / area(root).
// One thread per inner node adds its own term and the terms of the leaves below it; block reduce, one atomic per block.
__global__ __launch_bounds__(256) void k_sah_cost(const int2 *__restrict__ children, const int2 *__restrict__ range, int n, int leaf_max,
                                                  const float4 *__restrict__ box_lo, const float4 *__restrict__ box_hi, double *cost)
{
    __shared__ double s_sum[4];
    auto area = [&](int id) -> double {
        const float4 l = box_lo[id], h = box_hi[id];
        const double dx = (double)h.x - l.x, dy = (double)h.y - l.y, dz = (double)h.z - l.z;
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    };
    double v = 0.0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n - 1) {
        const int2 r = range[i];
        if (i == 0 || r.y - r.x + 1 > leaf_max) {           // a node the traversal visits
            v = area(i);
            const int2 c = children[i];
            const int cc[2] = { c.x, c.y };
            for (int k = 0; k < 2; k++) {
                if (cc[k] >= n - 1) v += area(cc[k]);                                   // a single reference
                else {
                    const int2 rc = range[cc[k]];
                    const int cnt = rc.y - rc.x + 1;
                    if (cnt <= leaf_max) v += area(cc[k]) * (double)cnt;                // a collapsed subtree
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        if (t != 0.0) atomicAdd(cost, t);
    }
}

__global__ void k_pre_level(const int2 *__restrict__ children, const int *__restrict__ depth, int n, int level,
                            const int *__restrict__ live, int *pre)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || depth[i] != level || live[i] == 0) return;
    const int2 c = children[i];
    const int p = pre[i];
    const int lx = c.x < n - 1 ? live[c.x] : 0;
    if (lx) pre[c.x] = p + 1;
    if (c.y < n - 1 && live[c.y]) pre[c.y] = p + 1 + lx;
}

// Traversal nodes.  A child whose sorted range holds <= leaf_max triangles becomes
// a leaf reference ~(first*8 + count-1); nodes inside such a subtree are never
// reached and are left zeroed.
__device__ __forceinline__ int emit_node(int i, const int2 *__restrict__ children, const int2 *__restrict__ range, const int *__restrict__ parent,
                       const int *__restrict__ depth, int n, int leaf_max, const float4 *__restrict__ box_lo,
                       const float4 *__restrict__ box_hi, const int *__restrict__ leafpos, const int *__restrict__ pre,
                       BvhNode *nodes, QuantGrid *grid)
{
    if (i >= n - 1) return 0;
    // 15-bit lattice over the root box (the union of all padded leaf boxes)
    const float4 rlo = box_lo[0], rhi = box_hi[0];
    const float gb[3] = { rlo.x, rlo.y, rlo.z };
    float gs[3] = { (rhi.x - rlo.x) * (1.0f / 32767.0f), (rhi.y - rlo.y) * (1.0f / 32767.0f), (rhi.z - rlo.z) * (1.0f / 32767.0f) };
    for (int k = 0; k < 3; k++)
        if (!(gs[k] > 0.0f)) gs[k] = 1e-30f;
    if (i == 0) {
        for (int k = 0; k < 3; k++) { grid->base[k] = gb[k]; grid->step[k] = gs[k]; }
    }
    if (parent[i] >= 0) {
        // A non-root node spanning <= leaf_max triangles is referenced as a leaf by its
        // parent, or lies inside such a leaf: never traversed.  Ranges nest, so every
        // node spanning more than leaf_max triangles has only live ancestors.
        int2 r = range[i];
        if (r.y - r.x + 1 <= leaf_max) return 0;
    }
    int2 c = children[i];
    int enc[2];
    int cc[2] = { c.x, c.y };
    for (int k = 0; k < 2; k++) {
        if (cc[k] >= n - 1) {
            const int j = cc[k] - (n - 1);
            enc[k] = ~(((leafpos ? leafpos[j] : j) << 3) | 0);
        } else {
            int2 r = range[cc[k]];
            int cnt = r.y - r.x + 1;
            enc[k] = (cnt <= leaf_max) ? ~((r.x << 3) | (cnt - 1)) : pre[cc[k]];
        }
    }
    float4 l0 = box_lo[c.x], h0 = box_hi[c.x], l1 = box_lo[c.y], h1 = box_hi[c.y];
    BvhNode nd;
    nd.xy0 = make_float4(l0.x, h0.x, l0.y, h0.y);
    nd.xy1 = make_float4(l1.x, h1.x, l1.y, h1.y);
    nd.z01 = make_float4(l0.z, h0.z, l1.z, h1.z);
    nd.ch = make_int4(enc[0], enc[1], 0, 0);
    nodes[pre[i]] = nd;
    return depth[i] + 1;
}

// Every lane reaches the reduction: one atomic per wave for the depth of the emitted tree.
__global__ void k_emit(const int2 *__restrict__ children, const int2 *__restrict__ range, const int *__restrict__ parent,
                       const int *__restrict__ depth, int n, int leaf_max, const float4 *__restrict__ box_lo,
                       const float4 *__restrict__ box_hi, const int *__restrict__ leafpos, const int *__restrict__ pre,
                       BvhNode *nodes, QuantGrid *grid, int *live_depth)
{
    int d = emit_node(blockIdx.x * blockDim.x + threadIdx.x, children, range, parent, depth, n, leaf_max, box_lo, box_hi, leafpos, pre,
                      nodes, grid);
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(live_depth, d);
}


// ---- 4-wide nodes.  A wide node is rooted at a binary node; its children are found by opening, up to
// twice, the inner child with the largest surface area (the one a ray is most likely to enter), so that big
// boxes are split further and small ones stay whole.  WIDE_COLLAPSE 0 restores the fixed rule (every binary
// node at an even depth is a wide root, its children are its grandchildren).  The choice is made once, at
// build time, and kept in wkids: a refit moves boxes, and the tree must not change under it.
// root[i] = 1 + wide depth if binary node i is a wide root, else 0; wkids[i] = its children (binary node ids,
// >= n-1: a single triangle; -1: unused slot).  wide[i] = wide roots in the subtree of i (bottom-up),
// pre4[i] = preorder position among the wide roots (top-down).
#ifndef WIDE_COLLAPSE
#define WIDE_COLLAPSE 1
#endif

__device__ __forceinline__ bool wide_is_leaf(int c, const int2 *range, int n, int leaf_max)
{
    if (c >= n - 1) return true;
    const int2 rg = range[c];
    return rg.y - rg.x + 1 <= leaf_max;
}

__device__ __forceinline__ float box_area(int c, const float4 *box_lo, const float4 *box_hi)
{
    const float4 l = box_lo[c], h = box_hi[c];
    const float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z;
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_wide_mark_level(const int2 *__restrict__ children, const int2 *__restrict__ range, const int *__restrict__ depth,
                                  const int *__restrict__ live, const float4 *__restrict__ box_lo, const float4 *__restrict__ box_hi,
                                  int n, int leaf_max, int level, int *root, int4 *wkids)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || depth[i] != level || live[i] == 0 || root[i] == 0) return;
    int who[4] = { children[i].x, children[i].y, -1, -1 };
    int m = 2;
#if WIDE_COLLAPSE
    while (m < 4) {
        int best = -1;
        float best_area = -1.0f;
        for (int k = 0; k < m; k++) {
            if (wide_is_leaf(who[k], range, n, leaf_max)) continue;
            const float a = box_area(who[k], box_lo, box_hi);
            if (a > best_area) { best_area = a; best = k; }
        }
        if (best < 0) break;
        const int2 g = children[who[best]];
        for (int k = m; k > best + 1; k--) who[k] = who[k - 1];     // keep the slot order of the binary tree
        who[best] = g.x;
        who[best + 1] = g.y;
        m++;
    }
#else
    {
        int out[4], q = 0;
        for (int k = 0; k < 2; k++) {
            if (wide_is_leaf(who[k], range, n, leaf_max)) { out[q++] = who[k]; continue; }
            const int2 g = children[who[k]];
            out[q++] = g.x;
            out[q++] = g.y;
        }
        m = q;
        for (int k = 0; k < 4; k++) who[k] = k < m ? out[k] : -1;
    }
#endif
    wkids[i] = make_int4(who[0], who[1], who[2], who[3]);
    for (int k = 0; k < m; k++)
        if (!wide_is_leaf(who[k], range, n, leaf_max)) root[who[k]] = root[i] + 1;
}

__global__ void k_wide_live_level(const int2 *__restrict__ children, const int *__restrict__ depth, const int *__restrict__ live,
                                  const int *__restrict__ root, int n, int level, int *wide)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || depth[i] != level) return;
    if (live[i] == 0) { wide[i] = 0; return; }
    const int2 c = children[i];
    wide[i] = (root[i] ? 1 : 0) + (c.x < n - 1 ? wide[c.x] : 0) + (c.y < n - 1 ? wide[c.y] : 0);
}

__global__ void k_wide_pre_level(const int2 *__restrict__ children, const int *__restrict__ depth, const int *__restrict__ live,
                                 const int *__restrict__ wide, const int *__restrict__ root, int n, int level, int *pre4)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || depth[i] != level || live[i] == 0) return;
    const int2 c = children[i];
    const int first = pre4[i] + (root[i] ? 1 : 0);
    if (c.x < n - 1 && live[c.x]) pre4[c.x] = first;
    if (c.y < n - 1 && live[c.y]) pre4[c.y] = first + (c.x < n - 1 ? wide[c.x] : 0);
}

// The top of the wide tree (wide depths 0..WIDE_TOP_LEVELS-1) is numbered as a complete 4-ary tree
// (root 0, children of slot s at 4s+1..4s+4) so that it sits at the start of the node array, where the
// traversal kernel can keep a copy in LDS; every other wide node follows in preorder.
__device__ __forceinline__ int wide_index(int i, const int *root, const int *slot, const int *pre4)
{
    return root[i] - 1 < WIDE_TOP_LEVELS ? slot[i] : WIDE_TOP_NODES + pre4[i];
}

__global__ void k_wide_slots(const int2 *__restrict__ range, const int *__restrict__ live, const int *__restrict__ root,
                             const int4 *__restrict__ wkids, int n, int leaf_max, int wide_level, int *slot)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || live[i] == 0 || root[i] != wide_level + 1) return;
    const int4 w = wkids[i];
    const int who[4] = { w.x, w.y, w.z, w.w };
    const int s = slot[i];
    for (int k = 0; k < 4; k++)
        if (who[k] >= 0 && !wide_is_leaf(who[k], range, n, leaf_max)) slot[who[k]] = 4 * s + 1 + k;
}

// A box rounded OUTWARD onto the 15-bit lattice, one word per axis: (0x8000 | lo) | (0x8000 | hi) << 16.
__device__ void lattice_box(const float4 lo, const float4 hi, const float *gb, const float *gs, unsigned int *w)
{
    const float l[3] = { lo.x, lo.y, lo.z }, h[3] = { hi.x, hi.y, hi.z };
    for (int k = 0; k < 3; k++) {
        double ql = floor(((double)l[k] - (double)gb[k]) / (double)gs[k]);
        double qh = ceil(((double)h[k] - (double)gb[k]) / (double)gs[k]);
        ql = fmin(fmax(ql, 0.0), 32767.0);
        qh = fmin(fmax(qh, 0.0), 32767.0);
        w[k] = (0x8000u | (unsigned int)ql) | ((0x8000u | (unsigned int)qh) << 16);
    }
}

__device__ __forceinline__ int emit_wide_node(int i, const int2 *__restrict__ range, const int *__restrict__ live, const int *__restrict__ root,
                            const int4 *__restrict__ wkids, int n, int leaf_max, const float4 *__restrict__ box_lo,
                            const float4 *__restrict__ box_hi, const int *__restrict__ leafpos, const int *__restrict__ pre4,
                            const int *__restrict__ slot, BvhNode4 *wnodes, BvhNode4F *fnodes)
{
    if (i >= n - 1 || live[i] == 0 || root[i] == 0) return 0;
    const float4 rlo = box_lo[0], rhi = box_hi[0];
    const float gb[3] = { rlo.x, rlo.y, rlo.z };
    float gs[3] = { (rhi.x - rlo.x) * (1.0f / 32767.0f), (rhi.y - rlo.y) * (1.0f / 32767.0f), (rhi.z - rlo.z) * (1.0f / 32767.0f) };
    for (int k = 0; k < 3; k++)
        if (!(gs[k] > 0.0f)) gs[k] = 1e-30f;
    // children in slot order: binary node (or single-triangle) ids and their references
    const int4 wk = wkids[i];
    const int who[4] = { wk.x, wk.y, wk.z, wk.w };
    int ref[4];
    unsigned int w[12];
    for (int k = 0; k < 4; k++) {
        const int c = who[k];
        if (c < 0) {
            w[3 * k] = w[3 * k + 1] = w[3 * k + 2] = 0x8000ffffu;     // lo = 32767 > hi = 0: never entered
            ref[k] = 0x7fffffff;
            continue;
        }
        if (c >= n - 1) {
            const int j = c - (n - 1);
            ref[k] = ~(((leafpos ? leafpos[j] : j) << 3) | 0);
        } else {
            const int2 rg = range[c];
            const int cnt = rg.y - rg.x + 1;
            ref[k] = cnt <= leaf_max ? ~((rg.x << 3) | (cnt - 1)) : wide_index(c, root, slot, pre4);
        }
        lattice_box(box_lo[c], box_hi[c], gb, gs, w + 3 * k);
    }
    BvhNode4 nd;
    nd.a = make_uint4(w[0], w[1], w[2], w[3]);
    nd.b = make_uint4(w[4], w[5], w[6], w[7]);
    nd.c = make_uint4(w[8], w[9], w[10], w[11]);
    nd.d = make_uint4((unsigned)ref[0], (unsigned)ref[1], (unsigned)ref[2], (unsigned)ref[3]);
    const int at = wide_index(i, root, slot, pre4);
    wnodes[at] = nd;
    if (fnodes) {
        // the float form: every plane as the value its 16-bit half decodes to (0x3f000000 | half << 8 = 1 + q / 32768)
        BvhNode4F fn;
        for (int k = 0; k < 4; k++) {
            fn.lo_x[k] = __uint_as_float(0x3f000000u | (w[3 * k] & 0xffffu) << 8);
            fn.hi_x[k] = __uint_as_float(0x3f000000u | (w[3 * k] >> 16) << 8);
            fn.lo_y[k] = __uint_as_float(0x3f000000u | (w[3 * k + 1] & 0xffffu) << 8);
            fn.hi_y[k] = __uint_as_float(0x3f000000u | (w[3 * k + 1] >> 16) << 8);
            fn.lo_z[k] = __uint_as_float(0x3f000000u | (w[3 * k + 2] & 0xffffu) << 8);
            fn.hi_z[k] = __uint_as_float(0x3f000000u | (w[3 * k + 2] >> 16) << 8);
            fn.ref[k] = ref[k];
            fn.pad[k] = 0;
        }
        fnodes[at] = fn;
    }
    return root[i];
}

__global__ void k_emit_wide(const int2 *__restrict__ range, const int *__restrict__ live, const int *__restrict__ root,
                            const int4 *__restrict__ wkids, int n, int leaf_max, const float4 *__restrict__ box_lo,
                            const float4 *__restrict__ box_hi, const int *__restrict__ leafpos, const int *__restrict__ pre4,
                            const int *__restrict__ slot, BvhNode4 *wnodes, BvhNode4F *fnodes, int *wide_depth)
{
    int d = emit_wide_node(blockIdx.x * blockDim.x + threadIdx.x, range, live, root, wkids, n, leaf_max, box_lo, box_hi, leafpos, pre4,
                           slot, wnodes, fnodes);
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(wide_depth, d);
}


__global__ void k_gather_tris(const float4 *__restrict__ tri_mt, const unsigned int *__restrict__ ref_tri, const unsigned int *__restrict__ sorted_vals,
                              const int *__restrict__ order, int n, float4 *bvh_tri)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    unsigned int src = ref_tri[sorted_vals[order ? order[k] : k]];
    float4 a = tri_mt[3 * (size_t)src], e1 = tri_mt[3 * (size_t)src + 1], e2 = tri_mt[3 * (size_t)src + 2];
    a.w = __int_as_float((int)src);
    bvh_tri[3 * (size_t)k] = a;
    bvh_tri[3 * (size_t)k + 1] = e1;
    bvh_tri[3 * (size_t)k + 2] = e2;
}

__global__ void k_init_bounds(unsigned int *bounds)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;
}

// Is this a binary tree over leaves 0..n-1 in order?  Every inner node's children must name it as their parent and
// split its leaf range without gap or overlap, the root must span everything, and the order array must hold every
// reference once.  bad[0] counts violations.
__global__ void k_tree_check(const int2 *__restrict__ children, const int2 *__restrict__ range, const int *__restrict__ parent,
                             const unsigned int *__restrict__ order, int n, int *seen, int *bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned int r = order[i];
        if (r >= (unsigned)n || atomicAdd(&seen[r], 1) != 0) atomicAdd(bad, 1);
    }
    if (i >= n - 1) return;
    const int2 c = children[i], rg = range[i];
    int fails = 0;
    if (i == 0 && (parent[0] != -1 || rg.x != 0 || rg.y != n - 1)) fails++;
    if (rg.x < 0 || rg.y >= n || rg.x >= rg.y) fails++;
    int2 sub[2];
    const int kid[2] = { c.x, c.y };
    for (int k = 0; k < 2; k++) {
        if (kid[k] <= 0 || kid[k] > 2 * n - 2 || kid[k] == i) { fails++; sub[k] = make_int2(0, 0); continue; }
        if (parent[kid[k]] != i) fails++;
        sub[k] = kid[k] >= n - 1 ? make_int2(kid[k] - (n - 1), kid[k] - (n - 1)) : range[kid[k]];
    }
    if (sub[0].x != rg.x || sub[0].y + 1 != sub[1].x || sub[1].y != rg.y) fails++;
    if (fails) atomicAdd(bad, fails);
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)); }
    T *release() { T *q = p; p = nullptr; return q; }
};

inline int blocks_for(int n, int bs) { return (n + bs - 1) / bs; }

// Exclusive scan of data[0..m) on `stream`; scratch holds blocks_for(m, SCAN_CHUNK) elements.  Small inputs take
// the one-block kernel (one launch), large ones the three-launch form.
template <typename T>
void scan_exclusive(T *data, int m, T *total, T *scratch, hipStream_t stream)
{
    if (m <= 8 * SCAN_CHUNK || !scratch) {
        hipLaunchKernelGGL(k_scan_exclusive<T>, dim3(1), dim3(1024), 0, stream, data, m, total);
        return;
    }
    const int chunks = blocks_for(m, SCAN_CHUNK);
    hipLaunchKernelGGL(k_scan_partials<T>, dim3(chunks), dim3(1024), 0, stream, (const T *)data, m, scratch);
    hipLaunchKernelGGL(k_scan_exclusive<T>, dim3(1), dim3(1024), 0, stream, scratch, chunks, total);
    hipLaunchKernelGGL(k_scan_chunks<T>, dim3(chunks), dim3(1024), 0, stream, data, m, (const T *)scratch);
}

}  // namespace

// What a refit needs from the build: the binary tree over the sorted triangles and the numbering of
// the emitted nodes.  None of it depends on where the vertices are, only on which triangle sits where.
struct LbvhTopology {
    int n = 0, max_depth = 0;          // n = references (leaves of the binary tree)
    int n_tris = 0;
    bool ploc = false;
    unsigned int *bounds = nullptr, *vals = nullptr;
    unsigned int *ref_tri = nullptr;   // [n] triangle of every reference
    float4 *ref_lo = nullptr, *ref_hi = nullptr;   // [n] reference boxes (unpadded)
    int2 *children = nullptr, *range = nullptr;
    int *parent = nullptr, *depth = nullptr, *scalars = nullptr, *leafpos = nullptr, *order = nullptr;
    int *live = nullptr, *pre = nullptr, *pre4 = nullptr, *slot = nullptr, *root = nullptr;
    int4 *wkids = nullptr;
    float4 *box_lo = nullptr, *box_hi = nullptr;
    QuantGrid *grid = nullptr;
    ~LbvhTopology()
    {
        void *ptrs[] = { bounds, vals, children, range, parent, depth, scalars, leafpos, order, live, pre, pre4, slot, root, wkids, box_lo, box_hi, grid,
                         ref_tri, ref_lo, ref_hi };
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
    }
};

int tri_setup_flat(const float *d_tri_verts, int n_tris, float4 *tri_mt, float4 *tri_n, int mat_index, hipStream_t stream)
{
    if (n_tris <= 0) return RT_OK;
    hipLaunchKernelGGL(k_tri_setup_flat, dim3(blocks_for(n_tris, 256)), dim3(256), 0, stream, d_tri_verts, n_tris, tri_mt,
                       tri_n, mat_index);
    return hipGetLastError() == hipSuccess ? RT_OK : RT_E_DEVICE;
}

int tri_setup_indexed(const float *d_verts, int n_verts, const int32_t *d_indices, int n_tris, float4 *tri_mt,
                      float4 *tri_n, int mat_index, int *d_bad_index_flag, hipStream_t stream)
{
    if (n_tris <= 0) return RT_OK;
    hipLaunchKernelGGL(k_tri_setup_indexed, dim3(blocks_for(n_tris, 256)), dim3(256), 0, stream, d_verts, n_verts,
                       d_indices, n_tris, tri_mt, tri_n, mat_index, d_bad_index_flag);
    return hipGetLastError() == hipSuccess ? RT_OK : RT_E_DEVICE;
}

void lbvh_free(LbvhResult *r)
{
    if (r->nodes) (void)hipFree(r->nodes);
    if (r->wnodes) (void)hipFree(r->wnodes);
    if (r->fnodes) (void)hipFree(r->fnodes);
    if (r->bvh_tri) (void)hipFree(r->bvh_tri);
    delete r->topo;
    *r = LbvhResult();
}

namespace {

// The two timing events of a build or refit, destroyed on every way out.
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t create()
    {
        hipError_t e = hipEventCreate(&a);
        if (e == hipSuccess) e = hipEventCreate(&b);
        return e;
    }
    ~EventPair()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

int lbvh_build_impl(const float4 *tri_mt, int n_tris, int leaf_max, int builder, int split_cells, hipStream_t stream, LbvhResult *out,
                    std::string *err);

}  // namespace

// Every failure leaves `out` empty: a half-built hierarchy (arrays allocated, depth and leaf size unset) is never kept.
int lbvh_build(const float4 *tri_mt, int n, int leaf_max, int builder, int split_cells, hipStream_t stream, LbvhResult *out,
               std::string *err)
{
    lbvh_free(out);
    const int rc = lbvh_build_impl(tri_mt, n, leaf_max, builder, split_cells, stream, out, err);
    if (rc != RT_OK) lbvh_free(out);
    return rc;
}

namespace {

// ------------------------------------------------------------------ top-down binned SAH (builder 3)
// The tree a surface-area-heuristic sweep gives, built on the device.  Every open node owns a contiguous
// range ("segment") of one order array.  A segment longer than SAH_SMALL references is split by the whole grid, one
// tree level per round: every reference drops its box into one of SAH_BINS bins per axis (atomics, collected per block
// in LDS for every segment the block meets), one wave per segment sweeps the 3 x (SAH_BINS-1) planes for the
// cheapest cut, a scan of the "goes left" flags places every reference (a stable partition), and the children's
// centroid bounds are gathered on the way.  A segment of at most SAH_SMALL references is finished by ONE wave, down to
// single references, out of LDS.  Node ids: a node that cuts its range after position g ("gamma") is inner node g --
// every gap between neighbouring leaves is cut by exactly one node, so the ids are unique in 0..n-2 -- except that the
// root is node 0 and the node cutting after position 0 takes the root's gamma instead.  Leaf k is (n-1)+k.
constexpr int SAH_BINS = 32;
constexpr int SAH_SMALL = 64;
constexpr int SAH_SMALL_BINS = 16;
constexpr int SAH_DEPTH_CAP = 60;      // a node at depth d over s references is cut in the middle once d + ceil(log2 s) reaches this

struct SahSeg { int first, last, parent, side, depth; };
struct SahBin { unsigned int count, lo[3], hi[3]; };
struct SahCut { int axis, bin, n_left, slot_left, slot_right; };     // axis < 0: cut in the middle of the range

__device__ __forceinline__ int sah_bin_of(float c, float lo, float hi, int nb)
{
    const float ext = hi - lo;
    if (!(ext > 0.f)) return 0;
    const int b = (int)((c - lo) * ((float)nb / ext));
    return b < 0 ? 0 : (b >= nb ? nb - 1 : b);
}

__device__ __forceinline__ int ceil_log2(int v) { return v <= 1 ? 0 : 32 - __clz(v - 1); }

__device__ __forceinline__ float half_area3(const float *l, const float *h)
{
    const float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
    return dx * dy + dy * dz + dz * dx;
}

// ord[p] = p, every position in segment 0 (or in none when the whole input is one small segment); centroid bounds of
// the root: every thread folds many references, a wave reduces by shuffles, the block by LDS, one lane per block issues
// the six atomics (they land in one cache line, where atomics retire at about 90 per microsecond).
__global__ __launch_bounds__(256) void k_sah_init(const float4 *__restrict__ ref_lo, const float4 *__restrict__ ref_hi, int n, int root_slot,
                                                  unsigned int *ord, int *seg_of, unsigned int *cb)
{
    __shared__ float s_lo[4][3], s_hi[4][3];
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        ord[p] = (unsigned)p;
        seg_of[p] = root_slot;
        const float4 l = ref_lo[p], h = ref_hi[p];
        const float c[3] = { 0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z) };
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], c[k]); hi[k] = fmaxf(hi[k], c[k]); }
    }
    for (int k = 0; k < 3; k++)
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; k++) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = s_lo[0][k], b = s_hi[0][k];
        for (int w = 1; w < 4; w++) { a = fminf(a, s_lo[w][k]); b = fmaxf(b, s_hi[w][k]); }
        atomicMin(&cb[k], float_to_ordered(a));
        atomicMax(&cb[3 + k], float_to_ordered(b));
    }
}

__global__ void k_sah_clear(SahBin *bins, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    SahBin b;
    b.count = 0;
    for (int k = 0; k < 3; k++) { b.lo[k] = 0xffffffffu; b.hi[k] = 0u; }
    bins[i] = b;
}

__device__ __forceinline__ void sah_bin_add(SahBin *b, const float4 l, const float4 h)
{
    atomicAdd(&b->count, 1u);
    atomicMin(&b->lo[0], float_to_ordered(l.x)); atomicMin(&b->lo[1], float_to_ordered(l.y)); atomicMin(&b->lo[2], float_to_ordered(l.z));
    atomicMax(&b->hi[0], float_to_ordered(h.x)); atomicMax(&b->hi[1], float_to_ordered(h.y)); atomicMax(&b->hi[2], float_to_ordered(h.z));
}

// Which of the block's open segments a position belongs to.  Segments are contiguous and longer than SAH_SMALL, so
// the 256 positions of a block meet at most SAH_BLOCK_SEGS of them; the k-th one in position order gets local index k
// and s_slot[k] = its slot in the round's segment list.  Returns -1 for a position in no open segment; a local index
// >= SAH_BLOCK_SEGS (cannot happen) sends the caller to the global arrays.  Two barriers; every thread must call it.
constexpr int SAH_BLOCK_SEGS = 256 / (SAH_SMALL + 1) + 2;

__device__ __forceinline__ int sah_local_segment(const int *__restrict__ seg_of, int p, int n, int n_act, int *s_wave, int *s_slot, int &seg,
                                                 int &n_local)
{
    const int t = threadIdx.x;
    int s = -1, prev = -2;
    if (p < n) {
        s = seg_of[p];
        if (s >= n_act) s = -1;
        if (t > 0) {
            prev = seg_of[p - 1];
            if (prev >= n_act) prev = -1;
        }
    }
    const bool head = s >= 0 && s != prev;
    const unsigned long long mask = __ballot(head);
    if ((t & 63) == 0) s_wave[t >> 6] = (int)__popcll(mask);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 4; w++) {
        const int c = s_wave[w];
        if (w < (t >> 6)) before += c;
        total += c;
    }
    const int local = before + (int)__popcll(mask & ((2ull << (t & 63)) - 1ull)) - 1;
    if (head && local < SAH_BLOCK_SEGS) s_slot[local] = s;
    __syncthreads();
    seg = s;
    n_local = total < SAH_BLOCK_SEGS ? total : SAH_BLOCK_SEGS;
    return s >= 0 ? local : -1;
}

// Every reference of an open segment into one bin per axis.  The block collects in LDS and adds each non-empty bin to
// the segment's global bins once (direct atomics from every thread ran at ~24 G per second and made this kernel most
// of the build).
__global__ __launch_bounds__(256) void k_sah_bin(const unsigned int *__restrict__ ord, const int *__restrict__ seg_of, int n,
                                                 const float4 *__restrict__ ref_lo, const float4 *__restrict__ ref_hi,
                                                 const unsigned int *__restrict__ cb, int n_act, int chunks, SahBin *bins)
{
    __shared__ SahBin s_bin[SAH_BLOCK_SEGS * 3 * SAH_BINS];
    __shared__ int s_wave[4], s_slot[SAH_BLOCK_SEGS];
    // A block takes `chunks` runs of 256 positions.  While they all lie in one segment (the first rounds of a large
    // input) it collects them all before it touches the global bins: fewer adds to the few cache lines every block wants.
    const int first = blockIdx.x * chunks * 256;
    if (first >= n) return;
    const int last = min(first + chunks * 256, n) - 1;
    const int s_first = seg_of[first];
    const bool one_segment = chunks > 1 && s_first >= 0 && s_first < n_act && s_first == seg_of[last];
    for (int chunk = 0; chunk < chunks; chunk++) {
        const int p = first + chunk * 256 + threadIdx.x;
        if (first + chunk * 256 >= n) break;
        if (!one_segment || chunk == 0) {
            __syncthreads();                                      // the flush of the chunk before has read s_bin
            for (int e = threadIdx.x; e < SAH_BLOCK_SEGS * 3 * SAH_BINS; e += 256) {
                SahBin b;
                b.count = 0;
                for (int k = 0; k < 3; k++) { b.lo[k] = 0xffffffffu; b.hi[k] = 0u; }
                s_bin[e] = b;
            }
        }
        int s, n_local, local;
        if (one_segment) {
            if (chunk == 0) {
                if (threadIdx.x == 0) s_slot[0] = s_first;
                __syncthreads();
            }
            s = s_first; n_local = 1; local = p < n ? 0 : -1;
        } else {
            local = sah_local_segment(seg_of, p, n, n_act, s_wave, s_slot, s, n_local);     // its barriers also order the clearing above
        }
        if (local >= 0) {
            const unsigned int r = ord[p];
            const float4 l = ref_lo[r], h = ref_hi[r];
            const float c[3] = { 0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z) };
            for (int a = 0; a < 3; a++) {
                const int b = sah_bin_of(c[a], ordered_to_float(cb[6 * s + a]), ordered_to_float(cb[6 * s + 3 + a]), SAH_BINS);
                sah_bin_add(local < SAH_BLOCK_SEGS ? &s_bin[(local * 3 + a) * SAH_BINS + b] : &bins[((size_t)s * 3 + a) * SAH_BINS + b], l, h);
            }
        }
        if (one_segment && chunk + 1 < chunks && first + (chunk + 1) * 256 < n) continue;       // keep collecting
        __syncthreads();
        for (int e = threadIdx.x; e < n_local * 3 * SAH_BINS; e += 256) {
            const SahBin b = s_bin[e];
            if (!b.count) continue;
            SahBin *g = &bins[(size_t)s_slot[e / (3 * SAH_BINS)] * 3 * SAH_BINS + e % (3 * SAH_BINS)];
            atomicAdd(&g->count, b.count);
            for (int k = 0; k < 3; k++) { atomicMin(&g->lo[k], b.lo[k]); atomicMax(&g->hi[k], b.hi[k]); }
        }
    }
}

// The id of the node that cuts after position gamma; the root announces its own gamma through *root_gamma.
__device__ __forceinline__ int sah_node_id(int parent, int gamma, int *root_gamma)
{
    if (parent < 0) { *root_gamma = gamma; return 0; }
    return gamma == 0 ? *root_gamma : gamma;
}

// Links node `id` under its parent, records its range, and sends its two children on: a single reference becomes a
// leaf, a short range goes to the list for the wave builder, a long one into the next round (returns its slot there).
__device__ int sah_child(int id, int side, int first, int last, int depth, int n, int2 *children, int *parent, SahSeg *small, int *n_small,
                         SahSeg *next, int *n_next, int next_cap, unsigned int *cb_next)
{
    const int size = last - first + 1;
    if (size == 1) {
        const int leaf = (n - 1) + first;
        ((int *)(children + id))[side] = leaf;
        parent[leaf] = id;
        return -1;
    }
    SahSeg sg;
    sg.first = first; sg.last = last; sg.parent = id; sg.side = side; sg.depth = depth;
    if (size <= SAH_SMALL) {
        small[atomicAdd(n_small, 1)] = sg;
        return -1;
    }
    const int k = atomicAdd(n_next, 1);
    if (k >= next_cap) return -1;            // cannot happen (disjoint ranges longer than SAH_SMALL); the host checks the count
    next[k] = sg;
    for (int a = 0; a < 3; a++) { cb_next[6 * k + a] = 0xffffffffu; cb_next[6 * k + 3 + a] = 0u; }
    return k;
}

// One wave per open segment.  Lane b holds bin b of the axis at hand; inclusive scans by shuffles give the box and
// count of bins 0..b (from below) and b..31 (from above), lane b < 31 prices the plane between bins b and b + 1, a
// shuffle reduction keeps the cheapest (lowest axis, then lowest bin on a tie).
__global__ __launch_bounds__(64) void k_sah_split(const SahSeg *__restrict__ act, int n_act, const SahBin *__restrict__ bins, int n, int2 *children,
                                                  int2 *range, int *parent, int *scalars /*0 root gamma, 1 segments of the next round, 2 small segments*/,
                                                  SahSeg *small, SahSeg *next, int next_cap, unsigned int *cb_next, SahCut *cut)
{
    static_assert(SAH_BINS == 32, "one half-wave of lanes per axis sweep");
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= n_act) return;
    const SahSeg seg = act[s];
    const int size = seg.last - seg.first + 1;
    float best = FLT_MAX;
    int best_key = 0x7fffffff, best_left = 0;           // key = axis * 32 + bin
    if (seg.depth + ceil_log2(size) < SAH_DEPTH_CAP) {
        const int b = lane & 31;
        const bool up = lane < 32;                      // lanes 0..31 scan from bin 0 upward, lanes 32..63 from bin 31 downward
        for (int a = 0; a < 3; a++) {
            const SahBin q = bins[((size_t)s * 3 + a) * SAH_BINS + (up ? b : 31 - b)];
            float l[3], h[3];
            int c = (int)q.count;
            for (int k = 0; k < 3; k++) {
                l[k] = q.count ? ordered_to_float(q.lo[k]) : FLT_MAX;
                h[k] = q.count ? ordered_to_float(q.hi[k]) : -FLT_MAX;
            }
            for (int off = 1; off < 32; off <<= 1) {            // inclusive scan inside each half-wave
                const int oc = __shfl_up(c, off, 32);
                float ol[3], oh[3];
                for (int k = 0; k < 3; k++) { ol[k] = __shfl_up(l[k], off, 32); oh[k] = __shfl_up(h[k], off, 32); }
                if (b >= off) {
                    c += oc;
                    for (int k = 0; k < 3; k++) { l[k] = fminf(l[k], ol[k]); h[k] = fmaxf(h[k], oh[k]); }
                }
            }
            const float part = c > 0 ? half_area3(l, h) * (float)c : 0.f;
            // plane after bin b: lower part = upward scan at b (lane b), upper part = downward scan at 30 - b (lane 32 + 30 - b)
            const float upper = __shfl(part, 32 + (30 - b < 0 ? 0 : 30 - b));
            if (up && b < 31 && c > 0 && c < size) {
                const float cost = part + upper;
                if (cost < best) { best = cost; best_key = a * 32 + b; best_left = c; }
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float oc = __shfl_xor(best, off);
            const int ok = __shfl_xor(best_key, off), ol = __shfl_xor(best_left, off);
            if (oc < best || (oc == best && ok < best_key)) { best = oc; best_key = ok; best_left = ol; }
        }
    }
    if (lane != 0) return;
    const int axis = best == FLT_MAX ? -1 : best_key >> 5, bin = best == FLT_MAX ? 0 : best_key & 31;
    const int n_left = best == FLT_MAX ? size / 2 : best_left;
    const int gamma = seg.first + n_left - 1;
    const int id = sah_node_id(seg.parent, gamma, scalars);
    parent[id] = seg.parent;
    range[id] = make_int2(seg.first, seg.last);
    if (seg.parent >= 0) ((int *)(children + seg.parent))[seg.side] = id;
    SahCut c;
    c.axis = axis; c.bin = bin; c.n_left = n_left;
    c.slot_left = sah_child(id, 0, seg.first, gamma, seg.depth + 1, n, children, parent, small, scalars + 2, next, scalars + 1, next_cap, cb_next);
    c.slot_right = sah_child(id, 1, gamma + 1, seg.last, seg.depth + 1, n, children, parent, small, scalars + 2, next, scalars + 1, next_cap, cb_next);
    cut[s] = c;
}

__global__ void k_sah_flag(const unsigned int *__restrict__ ord, const int *__restrict__ seg_of, int n, const float4 *__restrict__ ref_lo,
                           const float4 *__restrict__ ref_hi, const SahSeg *__restrict__ act, int n_act, const unsigned int *__restrict__ cb,
                           const SahCut *__restrict__ cut, unsigned int *flag, unsigned int *scan)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int s = seg_of[p];
    unsigned int left = 0;
    if (s >= 0 && s < n_act) {
        const SahCut c = cut[s];
        if (c.axis < 0) {
            left = p - act[s].first < c.n_left;
        } else {
            const unsigned int r = ord[p];
            const float4 l = ref_lo[r], h = ref_hi[r];
            const float cen = c.axis == 0 ? 0.5f * (l.x + h.x) : (c.axis == 1 ? 0.5f * (l.y + h.y) : 0.5f * (l.z + h.z));
            left = sah_bin_of(cen, ordered_to_float(cb[6 * s + c.axis]), ordered_to_float(cb[6 * s + 3 + c.axis]), SAH_BINS) <= c.bin;
        }
    }
    flag[p] = left;
    scan[p] = left;
}

// scan = exclusive prefix sums of flag.  The partition is stable: both halves keep their order.  The children's
// centroid bounds are collected per block in LDS (two children per open segment the block meets) and added once.
__global__ __launch_bounds__(256) void k_sah_scatter(const unsigned int *__restrict__ ord, const int *__restrict__ seg_of, int n,
                                                     const float4 *__restrict__ ref_lo, const float4 *__restrict__ ref_hi,
                                                     const SahSeg *__restrict__ act, int n_act, const SahCut *__restrict__ cut,
                                                     const unsigned int *__restrict__ flag, const unsigned int *__restrict__ scan,
                                                     unsigned int *ord_next, int *seg_next, unsigned int *cb_next)
{
    __shared__ unsigned int s_cb[SAH_BLOCK_SEGS * 2 * 6];
    __shared__ int s_wave[4], s_slot[SAH_BLOCK_SEGS];
    if (threadIdx.x < SAH_BLOCK_SEGS * 12) s_cb[threadIdx.x] = threadIdx.x % 6 < 3 ? 0xffffffffu : 0u;
    const int p = blockIdx.x * 256 + threadIdx.x;
    int s, n_local;
    const int local = sah_local_segment(seg_of, p, n, n_act, s_wave, s_slot, s, n_local);
    if (p < n) {
        const unsigned int r = ord[p];
        if (local < 0) {
            ord_next[p] = r;
            seg_next[p] = -1;
        } else {
            const SahSeg seg = act[s];
            const SahCut c = cut[s];
            const int left_rank = (int)(scan[p] - scan[seg.first]);
            const bool left = flag[p] != 0;
            const int dest = left ? seg.first + left_rank : seg.first + c.n_left + (p - seg.first - left_rank);
            const int slot = left ? c.slot_left : c.slot_right;
            ord_next[dest] = r;
            seg_next[dest] = slot;
            if (slot >= 0) {
                const float4 l = ref_lo[r], h = ref_hi[r];
                const unsigned int cl[3] = { float_to_ordered(0.5f * (l.x + h.x)), float_to_ordered(0.5f * (l.y + h.y)), float_to_ordered(0.5f * (l.z + h.z)) };
                unsigned int *dst = local < SAH_BLOCK_SEGS ? &s_cb[(local * 2 + (left ? 0 : 1)) * 6] : cb_next + 6 * (size_t)slot;
                for (int a = 0; a < 3; a++) { atomicMin(&dst[a], cl[a]); atomicMax(&dst[3 + a], cl[a]); }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < n_local * 12) {
        const int k = threadIdx.x % 6, side = (threadIdx.x / 6) & 1;
        const SahCut c = cut[s_slot[threadIdx.x / 12]];
        const int slot = side == 0 ? c.slot_left : c.slot_right;
        const unsigned int v = s_cb[threadIdx.x];
        if (slot >= 0 && v != (k < 3 ? 0xffffffffu : 0u)) {
            if (k < 3) atomicMin(&cb_next[6 * (size_t)slot + k], v);
            else atomicMax(&cb_next[6 * (size_t)slot + k], v);
        }
    }
}

// One wave finishes one short segment.  Position = lane; the boxes live in LDS and move with the partition.  Lane j
// prices plane j % (SAH_SMALL_BINS-1) of axis j / (SAH_SMALL_BINS-1) by walking the segment, the cheapest plane wins
// (lowest lane on a tie), ballots give the stable partition.  Open sub-ranges wait on a stack in LDS.
__global__ __launch_bounds__(64) void k_sah_small(const SahSeg *__restrict__ small, int n, unsigned int *ord, const float4 *__restrict__ ref_lo,
                                                  const float4 *__restrict__ ref_hi, int *scalars, int2 *children, int2 *range, int *parent)
{
    __shared__ float s_lo[3][64], s_hi[3][64];
    __shared__ unsigned int s_ref[64];
    __shared__ int s_stack[64][5];
    constexpr int PLANES = SAH_SMALL_BINS - 1;
    const int lane = threadIdx.x;
    const SahSeg seg = small[blockIdx.x];
    const int F = seg.first, m = seg.last - seg.first + 1;
    if (m < 2 || m > 64 || F < 0 || seg.last >= n) return;
    if (lane < m) {
        const unsigned int r = ord[F + lane];
        const float4 l = ref_lo[r], h = ref_hi[r];
        s_ref[lane] = r;
        s_lo[0][lane] = l.x; s_lo[1][lane] = l.y; s_lo[2][lane] = l.z;
        s_hi[0][lane] = h.x; s_hi[1][lane] = h.y; s_hi[2][lane] = h.z;
    }
    if (lane == 0) { s_stack[0][0] = 0; s_stack[0][1] = m - 1; s_stack[0][2] = seg.parent; s_stack[0][3] = seg.side; s_stack[0][4] = seg.depth; }
    int sp = 1;
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    while (sp > 0) {
        sp--;
        const int f = s_stack[sp][0], l = s_stack[sp][1], par = s_stack[sp][2], side = s_stack[sp][3], depth = s_stack[sp][4];
        const int size = l - f + 1;
        const bool in = lane >= f && lane <= l;
        float cmin[3], cmax[3];
        for (int a = 0; a < 3; a++) {
            const float c = in ? 0.5f * (s_lo[a][lane] + s_hi[a][lane]) : 0.f;
            float lo = in ? c : FLT_MAX, hi = in ? c : -FLT_MAX;
            for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
            cmin[a] = lo; cmax[a] = hi;
        }
        float cost = FLT_MAX;
        if (lane < 3 * PLANES && depth + ceil_log2(size) < SAH_DEPTH_CAP) {
            const int a = lane / PLANES, k = lane % PLANES;
            const float alo = a == 0 ? cmin[0] : (a == 1 ? cmin[1] : cmin[2]), ahi = a == 0 ? cmax[0] : (a == 1 ? cmax[1] : cmax[2]);
            float ll[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, lh[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            float rl[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, rh[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            int nl = 0;
            for (int i = f; i <= l; i++) {
                const bool left = sah_bin_of(0.5f * (s_lo[a][i] + s_hi[a][i]), alo, ahi, SAH_SMALL_BINS) <= k;
                nl += left ? 1 : 0;
                for (int q = 0; q < 3; q++) {
                    const float bl = s_lo[q][i], bh = s_hi[q][i];
                    ll[q] = fminf(ll[q], left ? bl : FLT_MAX); lh[q] = fmaxf(lh[q], left ? bh : -FLT_MAX);
                    rl[q] = fminf(rl[q], left ? FLT_MAX : bl); rh[q] = fmaxf(rh[q], left ? -FLT_MAX : bh);
                }
            }
            if (nl > 0 && nl < size) cost = half_area3(ll, lh) * (float)nl + half_area3(rl, rh) * (float)(size - nl);
        }
        float best = cost;
        int who = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const float oc = __shfl_xor(best, off);
            const int ow = __shfl_xor(who, off);
            if (oc < best || (oc == best && ow < who)) { best = oc; who = ow; }
        }
        bool left;
        if (best == FLT_MAX) {
            left = lane - f < size / 2;
        } else {
            const int a = who / PLANES, k = who % PLANES;
            const float alo = a == 0 ? cmin[0] : (a == 1 ? cmin[1] : cmin[2]), ahi = a == 0 ? cmax[0] : (a == 1 ? cmax[1] : cmax[2]);
            left = in && sah_bin_of(0.5f * (s_lo[a][lane & 63] + s_hi[a][lane & 63]), alo, ahi, SAH_SMALL_BINS) <= k;
        }
        const unsigned long long mask = __ballot(in && left);
        const int n_left = __popcll(mask), left_rank = __popcll(mask & lt_mask);
        const int dest = left ? f + left_rank : f + n_left + (lane - f - left_rank);
        float ml[3], mh[3];
        unsigned int mr = 0;
        if (in) {
            for (int q = 0; q < 3; q++) { ml[q] = s_lo[q][lane]; mh[q] = s_hi[q][lane]; }
            mr = s_ref[lane];
        }
        __syncthreads();
        if (in) {
            for (int q = 0; q < 3; q++) { s_lo[q][dest] = ml[q]; s_hi[q][dest] = mh[q]; }
            s_ref[dest] = mr;
        }
        if (lane == 0) {
            const int gamma = F + f + n_left - 1;
            const int id = sah_node_id(par, gamma, scalars);
            parent[id] = par;
            range[id] = make_int2(F + f, F + l);
            if (par >= 0) ((int *)(children + par))[side] = id;
            for (int sd = 0; sd < 2; sd++) {
                const int cf = sd == 0 ? f : f + n_left, cl = sd == 0 ? f + n_left - 1 : l;
                if (cf == cl) {
                    const int leaf = (n - 1) + F + cf;
                    ((int *)(children + id))[sd] = leaf;
                    parent[leaf] = id;
                } else {
                    const int t = sp + (sd == 1 && n_left > 1 ? 1 : 0);
                    s_stack[t][0] = cf; s_stack[t][1] = cl; s_stack[t][2] = id; s_stack[t][3] = sd; s_stack[t][4] = depth + 1;
                }
            }
        }
        const bool open_l = n_left > 1, open_r = size - n_left > 1;
        sp += (open_l ? 1 : 0) + (open_r ? 1 : 0);
        __syncthreads();
    }
    if (lane < m) ord[F + lane] = s_ref[lane];
}

int lbvh_build_impl(const float4 *tri_mt, int n_tris, int leaf_max, int builder, int split_cells, hipStream_t stream, LbvhResult *out,
                    std::string *err)
{
    if (n_tris < 2) {
        if (err) *err = "lbvh_build needs at least 2 triangles";
        return RT_E_INVALID;
    }
    if (leaf_max < 1 || leaf_max > 8) {
        if (err) *err = "leaf_max must be 1..8";
        return RT_E_INVALID;
    }
    if (split_cells < 1 || split_cells > 32) {
        if (err) *err = "split cells per axis must be 1..32";
        return RT_E_INVALID;
    }
    const int split_requested = split_cells;     // what the caller asked for (the grid below may be made coarser to fit a budget)
    EventPair ev;
    RT_HIP_CHECK(ev.create(), err);
    const hipEvent_t ev0 = ev.a, ev1 = ev.b;
    RT_HIP_CHECK(hipEventRecord(ev0, stream), err);

    // ---- scene bounds, then the references: one per triangle, more for the triangles that are cut
    DevBuf<unsigned int> bounds, counts, total, ref_tri;
    DevBuf<float4> ref_lo, ref_hi;
    if (bounds.alloc(6) != hipSuccess || counts.alloc(n_tris) != hipSuccess || total.alloc(1) != hipSuccess) {
        if (err) *err = "lbvh_build allocation: reference counts";
        return RT_E_NOMEM;
    }
    const int GT = blocks_for(n_tris, 256);
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, bounds.p);
    hipLaunchKernelGGL(k_bounds, dim3(GT < 256 ? GT : 256), dim3(256), 0, stream, tri_mt, n_tris, bounds.p);
    unsigned int m_refs = (unsigned)n_tris;
    if (split_cells > 1) {
        DevBuf<unsigned int> scan_tmp, cell_tri, flags;
        DevBuf<float4> cell_lo, cell_hi;
        if (scan_tmp.alloc((size_t)blocks_for(n_tris, 4096) + 1) != hipSuccess) {
            if (err) *err = "lbvh_build allocation: reference scan";
            return RT_E_NOMEM;
        }
        // A scene made of nothing but large triangles would ask for up to 1024 cells each: the grid is halved until
        // the cells fit a budget of 4 per triangle (plus 4 M), which also keeps every 32-bit count below 2^30.
        DevBuf<unsigned long long> total64;
        if (total64.alloc(1) != hipSuccess) {
            if (err) *err = "lbvh_build allocation: reference total";
            return RT_E_NOMEM;
        }
        const unsigned long long budget = std::min<unsigned long long>((unsigned long long)n_tris * 4ull + (4ull << 20), (1ull << 30) - 1);
        unsigned long long cells64 = 0;
        for (;;) {
            RT_HIP_CHECK(hipMemsetAsync(total64.p, 0, sizeof(unsigned long long), stream), err);
            hipLaunchKernelGGL(k_split_count, dim3(GT), dim3(256), 0, stream, tri_mt, n_tris, bounds.p, split_cells, counts.p);
            hipLaunchKernelGGL(k_sum_counts, dim3(GT < 1024 ? GT : 1024), dim3(256), 0, stream, (const unsigned int *)counts.p, n_tris, total64.p);
            RT_HIP_CHECK(hipMemcpyAsync(&cells64, total64.p, sizeof cells64, hipMemcpyDeviceToHost, stream), err);
            RT_HIP_CHECK(hipStreamSynchronize(stream), err);
            if (cells64 <= budget || split_cells <= 1) break;
            split_cells = split_cells / 2;
        }
        if (cells64 < (unsigned long long)n_tris || cells64 > budget) {
            if (err) *err = "triangle split grid out of range";
            return cells64 < (unsigned long long)n_tris ? RT_E_DEVICE : RT_E_LIMIT;
        }
        const unsigned int n_cells = (unsigned int)cells64;
        if (n_cells > (unsigned)n_tris) scan_exclusive<unsigned int>(counts.p, n_tris, total.p, scan_tmp.p, stream);
        if (n_cells > (unsigned)n_tris) {            // some triangle is cut
            DevBuf<unsigned int> scan_tmp2;
            if (cell_tri.alloc(n_cells) != hipSuccess || flags.alloc(n_cells) != hipSuccess || cell_lo.alloc(n_cells) != hipSuccess ||
                cell_hi.alloc(n_cells) != hipSuccess || scan_tmp2.alloc((size_t)blocks_for((int)n_cells, 4096) + 1) != hipSuccess) {
                if (err) *err = "lbvh_build allocation: split cells";
                return RT_E_NOMEM;
            }
            const int GC = blocks_for((int)n_cells, 256);
            hipLaunchKernelGGL(k_split_cells, dim3(GC), dim3(256), 0, stream, tri_mt, n_tris, bounds.p, split_cells, counts.p, n_cells, cell_tri.p,
                               cell_lo.p, cell_hi.p, flags.p);
            scan_exclusive<unsigned int>(flags.p, (int)n_cells, total.p, scan_tmp2.p, stream);
            RT_HIP_CHECK(hipMemcpyAsync(&m_refs, total.p, sizeof m_refs, hipMemcpyDeviceToHost, stream), err);
            RT_HIP_CHECK(hipStreamSynchronize(stream), err);
            if (m_refs < (unsigned)n_tris || m_refs > n_cells) {
                if (err) *err = "reference count out of range";
                return RT_E_DEVICE;
            }
            if (ref_tri.alloc(m_refs) != hipSuccess || ref_lo.alloc(m_refs) != hipSuccess || ref_hi.alloc(m_refs) != hipSuccess) {
                if (err) *err = "lbvh_build allocation: references";
                return RT_E_NOMEM;
            }
            hipLaunchKernelGGL(k_split_compact, dim3(GC), dim3(256), 0, stream, cell_tri.p, cell_lo.p, cell_hi.p, flags.p, n_cells, m_refs,
                               ref_tri.p, ref_lo.p, ref_hi.p);
            RT_HIP_CHECK(hipStreamSynchronize(stream), err);      // the cell arrays go out of scope here
        }
    }
    if ((long long)m_refs * 8 >= (1ll << 31)) {
        if (err) *err = "reference count beyond the leaf encoding (2^28)";
        return RT_E_LIMIT;
    }
    if (((long long)m_refs + WIDE_TOP_NODES) * (long long)sizeof(BvhNode4) >= (1ll << 32)) {
        if (err) *err = "reference count beyond the 32-bit byte offsets of the wide node array (2^26)";
        return RT_E_LIMIT;
    }
    const int n = (int)m_refs;          // leaves of the hierarchy from here on
    if (!ref_tri.p) {                   // nothing was cut: reference i is triangle i with its own box
        if (ref_tri.alloc(n) != hipSuccess || ref_lo.alloc(n) != hipSuccess || ref_hi.alloc(n) != hipSuccess) {
            if (err) *err = "lbvh_build allocation: references";
            return RT_E_NOMEM;
        }
        hipLaunchKernelGGL(k_ref_identity, dim3(GT), dim3(256), 0, stream, n, ref_tri.p);
        hipLaunchKernelGGL(k_ref_boxes_whole, dim3(GT), dim3(256), 0, stream, tri_mt, (const unsigned int *)ref_tri.p, n, ref_lo.p, ref_hi.p);
    }

    const int n_tiles = blocks_for(n, SORT_TILE);
    DevBuf<unsigned int> vals[2], hist;
    DevBuf<unsigned long long> keys[2];
    DevBuf<int2> children, range;
    DevBuf<int> parent, depth, scalars;
    DevBuf<float4> box_lo, box_hi;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(vals[0].alloc(n)); A(vals[1].alloc(n)); A(keys[0].alloc(n)); A(keys[1].alloc(n));
    A(hist.alloc((size_t)256 * n_tiles)); A(children.alloc(n)); A(range.alloc(n)); A(parent.alloc(2 * (size_t)n));
    A(depth.alloc(n)); A(scalars.alloc(4)); A(box_lo.alloc(2 * (size_t)n)); A(box_hi.alloc(2 * (size_t)n));
    DevBuf<QuantGrid> grid;
    A(grid.alloc(1));
    DevBuf<unsigned int> scan_u32;               // chunk sums of the multi-block scans
    DevBuf<unsigned long long> scan_u64;
    A(scan_u32.alloc((size_t)blocks_for(256 * n_tiles, 4096) + 1));
    A(scan_u64.alloc((size_t)blocks_for(n, 4096) + 1));
    A(hipMalloc((void **)&out->nodes, sizeof(BvhNode) * (size_t)(n - 1)));
    A(hipMalloc((void **)&out->wnodes, sizeof(BvhNode4) * ((size_t)(n - 1) + WIDE_TOP_NODES)));
    A(hipMalloc((void **)&out->fnodes, sizeof(BvhNode4F) * ((size_t)(n - 1) + WIDE_TOP_NODES)));
    A(hipMalloc((void **)&out->bvh_tri, sizeof(float4) * 3 * (size_t)n));
    if (e != hipSuccess) {
        if (err) *err = std::string("lbvh_build allocation: ") + hipGetErrorString(e);
        lbvh_free(out);
        return RT_E_NOMEM;
    }
    const int B = 256, G = blocks_for(n, B);
    int cur = 0;
    const bool sah = builder == 0 || builder == 3;
    if (sah) {
        // top-down binned SAH on the device: rounds over the long segments, then one wave per short segment
        const int cap = n / (SAH_SMALL + 1) + 2, small_cap = n / 2 + 2;
        DevBuf<int> seg_of[2], sc;
        DevBuf<unsigned int> flag, scan, scan_tmp, cb[2];
        DevBuf<SahSeg> act[2], small;
        DevBuf<SahBin> bins;
        DevBuf<SahCut> cut;
        hipError_t e3 = hipSuccess;
        auto A3 = [&](hipError_t r) { if (e3 == hipSuccess) e3 = r; };
        A3(seg_of[0].alloc(n)); A3(seg_of[1].alloc(n)); A3(sc.alloc(4)); A3(flag.alloc(n)); A3(scan.alloc(n));
        A3(scan_tmp.alloc((size_t)blocks_for(n, SCAN_CHUNK) + 1)); A3(cb[0].alloc(6 * (size_t)cap)); A3(cb[1].alloc(6 * (size_t)cap));
        A3(act[0].alloc(cap)); A3(act[1].alloc(cap)); A3(small.alloc(small_cap)); A3(bins.alloc((size_t)cap * 3 * SAH_BINS)); A3(cut.alloc(cap));
        if (e3 != hipSuccess) {
            if (err) *err = std::string("SAH builder allocation: ") + hipGetErrorString(e3);
            return RT_E_NOMEM;
        }
        RT_HIP_CHECK(hipMemsetAsync(sc.p, 0, 4 * sizeof(int), stream), err);
        hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, cb[0].p);
        SahSeg root_seg;
        root_seg.first = 0; root_seg.last = n - 1; root_seg.parent = -1; root_seg.side = 0; root_seg.depth = 0;
        int n_act = n > SAH_SMALL ? 1 : 0, n_small = n_act ? 0 : 1;
        RT_HIP_CHECK(hipMemcpyAsync(n_act ? act[0].p : small.p, &root_seg, sizeof root_seg, hipMemcpyHostToDevice, stream), err);
        if (!n_act) {
            RT_HIP_CHECK(hipMemcpyAsync(sc.p + 2, &n_small, sizeof(int), hipMemcpyHostToDevice, stream), err);
        }
        hipLaunchKernelGGL(k_sah_init, dim3(G < 512 ? G : 512), dim3(256), 0, stream, ref_lo.p, ref_hi.p, n, n_act ? 0 : -1, vals[0].p, seg_of[0].p,
                           cb[0].p);
        int a_cur = 0;
        const int bin_chunks = std::max(1, std::min(8, n >> 18));        // runs of 256 positions per binning block
        for (int round = 0; n_act > 0; round++) {
            if (round > 64) {
                if (err) *err = "SAH builder: the rounds do not end";
                return RT_E_DEVICE;
            }
            hipLaunchKernelGGL(k_sah_clear, dim3(blocks_for(n_act * 3 * SAH_BINS, B)), dim3(B), 0, stream, bins.p, n_act * 3 * SAH_BINS);
            hipLaunchKernelGGL(k_sah_bin, dim3(blocks_for(n, 256 * bin_chunks)), dim3(256), 0, stream, vals[cur].p, seg_of[a_cur].p, n, ref_lo.p, ref_hi.p,
                               cb[a_cur].p, n_act, bin_chunks, bins.p);
            RT_HIP_CHECK(hipMemsetAsync(sc.p + 1, 0, sizeof(int), stream), err);
            hipLaunchKernelGGL(k_sah_split, dim3(n_act), dim3(64), 0, stream, act[a_cur].p, n_act, bins.p, n, children.p, range.p,
                               parent.p, sc.p, small.p, act[a_cur ^ 1].p, cap, cb[a_cur ^ 1].p, cut.p);
            hipLaunchKernelGGL(k_sah_flag, dim3(G), dim3(B), 0, stream, vals[cur].p, seg_of[a_cur].p, n, ref_lo.p, ref_hi.p, act[a_cur].p, n_act,
                               cb[a_cur].p, cut.p, flag.p, scan.p);
            scan_exclusive<unsigned int>(scan.p, n, (unsigned int *)nullptr, scan_tmp.p, stream);
            hipLaunchKernelGGL(k_sah_scatter, dim3(G), dim3(256), 0, stream, vals[cur].p, seg_of[a_cur].p, n, ref_lo.p, ref_hi.p, act[a_cur].p, n_act,
                               cut.p, flag.p, scan.p, vals[cur ^ 1].p, seg_of[a_cur ^ 1].p, cb[a_cur ^ 1].p);
            int counts2[2] = { 0, 0 };
            RT_HIP_CHECK(hipMemcpyAsync(counts2, sc.p + 1, 2 * sizeof(int), hipMemcpyDeviceToHost, stream), err);
            RT_HIP_CHECK(hipStreamSynchronize(stream), err);
            if (counts2[0] < 0 || counts2[0] > cap || counts2[1] < 0 || counts2[1] > small_cap) {
                if (err) *err = "SAH builder: segment count out of range";
                return RT_E_DEVICE;
            }
            n_act = counts2[0];
            n_small = counts2[1];
            cur ^= 1;
            a_cur ^= 1;
        }
        if (n_small > 0)
            hipLaunchKernelGGL(k_sah_small, dim3(n_small), dim3(64), 0, stream, small.p, n, vals[cur].p, ref_lo.p, ref_hi.p, sc.p, children.p,
                               range.p, parent.p);
        // a malformed tree would send the level-by-level passes below (and every ray later) astray: check it
        RT_HIP_CHECK(hipMemsetAsync(depth.p, 0, sizeof(int) * (size_t)n, stream), err);
        hipLaunchKernelGGL(k_tree_check, dim3(G), dim3(B), 0, stream, children.p, range.p, parent.p, vals[cur].p, n, depth.p, sc.p + 3);
        int bad = 0;
        RT_HIP_CHECK(hipGetLastError(), err);
        RT_HIP_CHECK(hipMemcpyAsync(&bad, sc.p + 3, sizeof(int), hipMemcpyDeviceToHost, stream), err);
        RT_HIP_CHECK(hipStreamSynchronize(stream), err);       // also: the work arrays go out of scope
        if (bad) {
            if (err) *err = "SAH builder: the tree failed its check (" + std::to_string(bad) + " violations)";
            return RT_E_DEVICE;
        }
    } else {
        hipLaunchKernelGGL(k_morton, dim3(G), dim3(B), 0, stream, ref_lo.p, ref_hi.p, n, bounds.p, keys[0].p, vals[0].p);
    }
    for (int pass = 0; pass < 8 && !sah; pass++) {
        int shift = pass * 8;
        hipLaunchKernelGGL(k_sort_hist, dim3(n_tiles), dim3(64), 0, stream, keys[cur].p, n, shift, hist.p, n_tiles);
        scan_exclusive<unsigned int>(hist.p, 256 * n_tiles, (unsigned int *)nullptr, scan_u32.p, stream);
        hipLaunchKernelGGL(k_sort_scatter, dim3(n_tiles), dim3(64), 0, stream, keys[cur].p, vals[cur].p, n, shift, hist.p,
                           n_tiles, keys[cur ^ 1].p, vals[cur ^ 1].p);
        cur ^= 1;
    }
    RT_HIP_CHECK(hipMemsetAsync(scalars.p, 0, 4 * sizeof(int), stream), err);
    RT_HIP_CHECK(hipMemsetAsync(out->nodes, 0, sizeof(BvhNode) * (size_t)(n - 1), stream), err);
    RT_HIP_CHECK(hipMemsetAsync(out->wnodes, 0, sizeof(BvhNode4) * ((size_t)(n - 1) + WIDE_TOP_NODES), stream), err);
    RT_HIP_CHECK(hipMemsetAsync(out->fnodes, 0, sizeof(BvhNode4F) * ((size_t)(n - 1) + WIDE_TOP_NODES), stream), err);
    const bool ploc = builder != 1 && !sah;
    int max_depth = 0;
    DevBuf<int> cid[2], nn, size, offset, leafpos, order;
    DevBuf<unsigned long long> flags, totals;
    hipLaunchKernelGGL(k_leaf_boxes, dim3(G), dim3(B), 0, stream, ref_lo.p, ref_hi.p, vals[cur].p, n, bounds.p, box_lo.p, box_hi.p);
    if (!ploc) {
        if (!sah)
            hipLaunchKernelGGL(k_karras, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, keys[cur].p, n, children.p, range.p,
                               parent.p);
        hipLaunchKernelGGL(k_node_depth, dim3(min(blocks_for(n - 1, 256), 2048)), dim3(256), 0, stream, parent.p, n - 1, depth.p, scalars.p);
        RT_HIP_CHECK(hipMemcpyAsync(&max_depth, scalars.p, sizeof(int), hipMemcpyDeviceToHost, stream), err);
        RT_HIP_CHECK(hipStreamSynchronize(stream), err);
        if (max_depth > 4096) {
            if (err) *err = "hierarchy deeper than 4096 levels";
            return RT_E_LIMIT;
        }
        for (int level = max_depth; level >= 0; level--)
            hipLaunchKernelGGL(k_refit_level, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, depth.p, n - 1, level,
                               box_lo.p, box_hi.p);
    } else {
        hipError_t e2 = hipSuccess;
        auto A2 = [&](hipError_t r) { if (e2 == hipSuccess) e2 = r; };
        A2(cid[0].alloc(n)); A2(cid[1].alloc(n)); A2(nn.alloc(n)); A2(size.alloc(2 * (size_t)n)); A2(offset.alloc(2 * (size_t)n));
        A2(leafpos.alloc(n)); A2(order.alloc(n)); A2(flags.alloc(n)); A2(totals.alloc(1));
        if (e2 != hipSuccess) {
            if (err) *err = std::string("PLOC allocation: ") + hipGetErrorString(e2);
            lbvh_free(out);
            return RT_E_NOMEM;
        }
        RT_HIP_CHECK(hipMemsetAsync(parent.p, 0xff, sizeof(int) * 2 * (size_t)n, stream), err);
        RT_HIP_CHECK(hipMemsetAsync(offset.p, 0, sizeof(int) * 2 * (size_t)n, stream), err);
        hipLaunchKernelGGL(k_ploc_init, dim3(G), dim3(B), 0, stream, n, cid[0].p, size.p);
        int C = n, created = 0, cc = 0;
        while (C > 1) {
            const int g = blocks_for(C, B);
            hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(256), 0, stream, cid[cc].p, C, box_lo.p, box_hi.p, nn.p);
            hipLaunchKernelGGL(k_ploc_flags, dim3(g), dim3(B), 0, stream, nn.p, C, flags.p);
            scan_exclusive<unsigned long long>(flags.p, C, totals.p, scan_u64.p, stream);
            hipLaunchKernelGGL(k_ploc_apply, dim3(g), dim3(B), 0, stream, cid[cc].p, nn.p, flags.p, C, n, created, cid[cc ^ 1].p,
                               children.p, parent.p, size.p, box_lo.p, box_hi.p);
            unsigned long long tot = 0;
            RT_HIP_CHECK(hipMemcpyAsync(&tot, totals.p, sizeof tot, hipMemcpyDeviceToHost, stream), err);
            RT_HIP_CHECK(hipStreamSynchronize(stream), err);
            const int kept = (int)(unsigned int)tot, merged = (int)(tot >> 32);
            if (merged <= 0 || kept != C - merged) {
                if (err) *err = "PLOC made no progress";
                lbvh_free(out);
                return RT_E_DEVICE;
            }
            created += merged;
            C = kept;
            cc ^= 1;
        }
        hipLaunchKernelGGL(k_node_depth, dim3(min(blocks_for(n - 1, 256), 2048)), dim3(256), 0, stream, parent.p, n - 1, depth.p, scalars.p);
        RT_HIP_CHECK(hipMemcpyAsync(&max_depth, scalars.p, sizeof(int), hipMemcpyDeviceToHost, stream), err);
        RT_HIP_CHECK(hipStreamSynchronize(stream), err);
        for (int level = 0; level <= max_depth; level++)
            hipLaunchKernelGGL(k_dfs_level, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, depth.p, size.p, n - 1, level,
                               offset.p);
        hipLaunchKernelGGL(k_dfs_finish, dim3(G), dim3(B), 0, stream, offset.p, size.p, n, range.p, leafpos.p, order.p);
    }
    DevBuf<int> live, pre;
    if (live.alloc(n) != hipSuccess || pre.alloc(n) != hipSuccess) {
        if (err) *err = "node order allocation failed";
        lbvh_free(out);
        return RT_E_NOMEM;
    }
    RT_HIP_CHECK(hipMemsetAsync(pre.p, 0, sizeof(int) * (size_t)n, stream), err);
    for (int level = max_depth; level >= 0; level--)
        hipLaunchKernelGGL(k_live_level, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, depth.p, range.p, n, leaf_max,
                           level, live.p);
    for (int level = 0; level <= max_depth; level++)
        hipLaunchKernelGGL(k_pre_level, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, depth.p, n, level, live.p, pre.p);
    hipLaunchKernelGGL(k_emit, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, range.p, parent.p, depth.p, n,
                       leaf_max, box_lo.p, box_hi.p, ploc ? leafpos.p : (const int *)nullptr, pre.p, out->nodes,
                       grid.p, scalars.p + 1);
    DevBuf<double> sah_acc;
    if (sah_acc.alloc(1) == hipSuccess) {
        RT_HIP_CHECK(hipMemsetAsync(sah_acc.p, 0, sizeof(double), stream), err);
        hipLaunchKernelGGL(k_sah_cost, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, range.p, n, leaf_max, box_lo.p, box_hi.p, sah_acc.p);
    }
    DevBuf<int> wide, pre4, root, slot;
    DevBuf<int4> wkids;
    if (wide.alloc(n) != hipSuccess || pre4.alloc(n) != hipSuccess || root.alloc(n) != hipSuccess || slot.alloc(n) != hipSuccess ||
        wkids.alloc(n) != hipSuccess) {
        if (err) *err = "wide node allocation failed";
        lbvh_free(out);
        return RT_E_NOMEM;
    }
    RT_HIP_CHECK(hipMemsetAsync(pre4.p, 0, sizeof(int) * (size_t)n, stream), err);
    RT_HIP_CHECK(hipMemsetAsync(slot.p, 0, sizeof(int) * (size_t)n, stream), err);
    RT_HIP_CHECK(hipMemsetAsync(root.p, 0, sizeof(int) * (size_t)n, stream), err);
    const int one = 1;
    RT_HIP_CHECK(hipMemcpyAsync(root.p, &one, sizeof(int), hipMemcpyHostToDevice, stream), err);     // binary node 0 is the wide root
    for (int level = 0; level <= max_depth; level++)
        hipLaunchKernelGGL(k_wide_mark_level, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, range.p, depth.p, live.p,
                           box_lo.p, box_hi.p, n, leaf_max, level, root.p, wkids.p);
    for (int level = max_depth; level >= 0; level--)
        hipLaunchKernelGGL(k_wide_live_level, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, depth.p, live.p, root.p, n,
                           level, wide.p);
    for (int level = 0; level <= max_depth; level++)
        hipLaunchKernelGGL(k_wide_pre_level, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, children.p, depth.p, live.p, wide.p,
                           root.p, n, level, pre4.p);
    for (int wl = 0; wl + 1 < WIDE_TOP_LEVELS; wl++)
        hipLaunchKernelGGL(k_wide_slots, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, range.p, live.p, root.p, wkids.p, n, leaf_max,
                           wl, slot.p);
    hipLaunchKernelGGL(k_emit_wide, dim3(blocks_for(n - 1, B)), dim3(B), 0, stream, range.p, live.p, root.p, wkids.p, n, leaf_max,
                       box_lo.p, box_hi.p, ploc ? leafpos.p : (const int *)nullptr, pre4.p, slot.p, out->wnodes, out->fnodes, scalars.p + 2);
    hipLaunchKernelGGL(k_gather_tris, dim3(G), dim3(B), 0, stream, tri_mt, ref_tri.p, vals[cur].p, ploc ? order.p : (const int *)nullptr, n,
                       out->bvh_tri);
    RT_HIP_CHECK(hipGetLastError(), err);
    RT_HIP_CHECK(hipEventRecord(ev1, stream), err);
    int live_depth = 0, wide_depth = 0;
    RT_HIP_CHECK(hipMemcpyAsync(&live_depth, scalars.p + 1, sizeof(int), hipMemcpyDeviceToHost, stream), err);
    RT_HIP_CHECK(hipMemcpyAsync(&wide_depth, scalars.p + 2, sizeof(int), hipMemcpyDeviceToHost, stream), err);
    RT_HIP_CHECK(hipMemcpyAsync(&out->grid, grid.p, sizeof(QuantGrid), hipMemcpyDeviceToHost, stream), err);
    double sah_sum = 0.0;
    float4 root_lo = make_float4(0.f, 0.f, 0.f, 0.f), root_hi = root_lo;
    if (sah_acc.p) {
        RT_HIP_CHECK(hipMemcpyAsync(&sah_sum, sah_acc.p, sizeof(double), hipMemcpyDeviceToHost, stream), err);
        RT_HIP_CHECK(hipMemcpyAsync(&root_lo, box_lo.p, sizeof(float4), hipMemcpyDeviceToHost, stream), err);
        RT_HIP_CHECK(hipMemcpyAsync(&root_hi, box_hi.p, sizeof(float4), hipMemcpyDeviceToHost, stream), err);
    }
    RT_HIP_CHECK(hipStreamSynchronize(stream), err);
    {
        const double dx = (double)root_hi.x - root_lo.x, dy = (double)root_hi.y - root_lo.y, dz = (double)root_hi.z - root_lo.z;
        const double a = 2.0 * (dx * dy + dy * dz + dz * dx);
        out->sah_cost = a > 0.0 ? (float)(sah_sum / a) : 0.f;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    out->n_nodes = n - 1;
    out->n_refs = n;
    out->split = split_requested;
    out->depth = live_depth;
    out->wide_depth = wide_depth;
    out->leaf_max = leaf_max;
    out->ms_build = ms;
    LbvhTopology *t = new (std::nothrow) LbvhTopology;
    if (t) {
        t->n = n; t->n_tris = n_tris; t->max_depth = max_depth; t->ploc = ploc;
        t->ref_tri = ref_tri.release(); t->ref_lo = ref_lo.release(); t->ref_hi = ref_hi.release();
        t->bounds = bounds.release(); t->vals = vals[cur].release();
        t->children = children.release(); t->range = range.release(); t->parent = parent.release();
        t->depth = depth.release(); t->scalars = scalars.release();
        t->leafpos = leafpos.release(); t->order = order.release();
        t->live = live.release(); t->pre = pre.release(); t->pre4 = pre4.release(); t->slot = slot.release(); t->root = root.release(); t->wkids = wkids.release();
        t->box_lo = box_lo.release(); t->box_hi = box_hi.release(); t->grid = grid.release();
    }
    out->topo = t;
    return RT_OK;
}

}  // namespace

// New vertex positions under the same tree: leaf boxes, bottom-up unions, then the node arrays and the
// triangle copy are written again.  The lattice follows the new root box.  Which triangles share a leaf
// does not change, so a large deformation costs traversal steps, never correctness.
int lbvh_refit(const float4 *tri_mt, int n_tris, hipStream_t stream, LbvhResult *out, std::string *err)
{
    LbvhTopology *t = out->topo;
    if (!t || !out->wnodes || t->n_tris != n_tris) {
        if (err) *err = "lbvh_refit: no hierarchy over this many triangles";
        return RT_E_STATE;
    }
    EventPair ev;
    RT_HIP_CHECK(ev.create(), err);
    const hipEvent_t ev0 = ev.a, ev1 = ev.b;
    RT_HIP_CHECK(hipEventRecord(ev0, stream), err);
    const int n = t->n;
    const int B = 256, G = blocks_for(n, B), GI = blocks_for(n - 1, B), GT = blocks_for(n_tris, B);
    const int *leafpos = t->ploc ? t->leafpos : (const int *)nullptr;
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, t->bounds);
    hipLaunchKernelGGL(k_bounds, dim3(GT < 256 ? GT : 256), dim3(256), 0, stream, tri_mt, n_tris, t->bounds);
    // the cells a triangle was cut by belong to where it was: after a move every reference takes its whole triangle's box
    hipLaunchKernelGGL(k_ref_boxes_whole, dim3(G), dim3(B), 0, stream, tri_mt, t->ref_tri, n, t->ref_lo, t->ref_hi);
    hipLaunchKernelGGL(k_leaf_boxes, dim3(G), dim3(B), 0, stream, t->ref_lo, t->ref_hi, t->vals, n, t->bounds, t->box_lo, t->box_hi);
    for (int level = t->max_depth; level >= 0; level--)
        hipLaunchKernelGGL(k_refit_level, dim3(GI), dim3(B), 0, stream, t->children, t->depth, n - 1, level, t->box_lo, t->box_hi);
    hipLaunchKernelGGL(k_emit, dim3(GI), dim3(B), 0, stream, t->children, t->range, t->parent, t->depth, n, out->leaf_max,
                       t->box_lo, t->box_hi, leafpos, t->pre, out->nodes, t->grid, t->scalars + 1);
    hipLaunchKernelGGL(k_emit_wide, dim3(GI), dim3(B), 0, stream, t->range, t->live, t->root, t->wkids, n, out->leaf_max, t->box_lo,
                       t->box_hi, leafpos, t->pre4, t->slot, out->wnodes, out->fnodes, t->scalars + 2);
    hipLaunchKernelGGL(k_gather_tris, dim3(G), dim3(B), 0, stream, tri_mt, t->ref_tri, t->vals, t->ploc ? t->order : (const int *)nullptr, n,
                       out->bvh_tri);
    RT_HIP_CHECK(hipGetLastError(), err);
    RT_HIP_CHECK(hipEventRecord(ev1, stream), err);
    RT_HIP_CHECK(hipMemcpyAsync(&out->grid, t->grid, sizeof(QuantGrid), hipMemcpyDeviceToHost, stream), err);
    RT_HIP_CHECK(hipStreamSynchronize(stream), err);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    out->ms_build = ms;
    return RT_OK;
}
