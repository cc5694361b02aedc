// trace.hip — the per-pixel trace path as wave64 HIP kernels for gfx950.
//
// What it replaces (Raytracer/main.cpp): the pixel loop 262-277 and everything it
// calls — camera ray 267-270, raytrace() 144-217 with Sphere/Plane/Triangle
// Raycast 56-68 / 84-92 / 109-129, hard shadows 177-188, N·L diffuse 191-195,
// pow-20 specular 198-204, mirror recursion 208-213, clamp + 8-bit store 139-142, 273.
//
// Shape of the computation on the GPU:
//   * The recursion becomes one launch per bounce level.  k_bounce<level> takes a
//     compacted queue of live rays (level 0 generates camera rays in registers),
//     finds the nearest primitive, casts the shadow rays, evaluates the local
//     shading term and writes one 16-byte record (local colour, hit id) per pixel
//     and level.  Rays that reflect are appended to the next level's queue with a
//     wave ballot + one atomic per wave.
//   * k_resolve folds the per-level records back together deepest-first,
//     col_k = local_k + (col_{k+1} * Mc_k) * kr_k, which is exactly the order in which
//     the reference's recursion unwinds (main.cpp:211-212), then clamps/quantises.
//   * Waves are persistent: each wave pulls 64-ray batches from a per-level cursor, so
//     launch size does not depend on queue lengths known only on the device.
//
// Float discipline: no contraction, IEEE divide/sqrt, operand order as in maths.h,
// so every value except powf is bit-equal to the reference's x86 results.
#include "rt_device.h"
#include <stdlib.h>

namespace {


template <bool COUNT>
__device__ __forceinline__ void brute_nearest(const DevScene &sc, F3 o, F3 d, float &best_t, int &best_id, Tally &tl)
{
    for (int k = 0; k < sc.n_tris; k++) {
        F3 v1 = f3(sc.tri_mt[3 * k]), e1 = f3(sc.tri_mt[3 * k + 1]), e2 = f3(sc.tri_mt[3 * k + 2]);
        float t;
        if (tri_raycast(v1, e1, e2, o, d, t) && best_t > t) { best_t = t; best_id = sc.n_analytic + k; }
    }
    if (COUNT) tl.tri += (unsigned)sc.n_tris;
}

// Everything a frame expects to find zero (frame counters, claim cursors), in one launch.
__global__ void k_zero_words(unsigned int *p, unsigned int n)
{
    for (unsigned int i = blockIdx.x * blockDim.x * 4u + threadIdx.x; i < n; i += gridDim.x * blockDim.x * 4u) {
        p[i] = 0u;
        if (i + blockDim.x < n) p[i + blockDim.x] = 0u;
        if (i + 2u * blockDim.x < n) p[i + 2u * blockDim.x] = 0u;
        if (i + 3u * blockDim.x < n) p[i + 3u * blockDim.x] = 0u;
    }
}

// stack: this lane's column of the block's LDS stack, element k at stack[k*BLOCK].
template <bool COUNT>
__device__ __forceinline__ void bvh_nearest(const DevScene &sc, F3 o, F3 d, bool live, float &best_t, int &best_id,
                                            int *stack, int stack_cap, Tally &tl, FrameCounters *counters)
{
    F3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int sp = 1;
    stack[0] = NODE_SENTINEL;
    int node = live ? 0 : NODE_SENTINEL;
    while (node != NODE_SENTINEL) {
        if (node >= 0) {
            const BvhNode *np = sc.nodes + node;
            float4 xy0 = np->xy0, xy1 = np->xy1, z01 = np->z01;
            int4 ch = np->ch;
            float tn0, tn1;
            bool h0 = slab(xy0.x, xy0.y, xy0.z, xy0.w, z01.x, z01.y, o, inv, best_t, tn0);
            bool h1 = slab(xy1.x, xy1.y, xy1.z, xy1.w, z01.z, z01.w, o, inv, best_t, tn1);
            if (COUNT) tl.box += 2;
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                int nearc = swap ? ch.y : ch.x, farc = swap ? ch.x : ch.y;
                // The host sizes the stack from the tree's depth, so this never drops a child; a guarded build says so if it does.
                if (sp < stack_cap) { stack[sp * BLOCK] = farc; sp++; }
                else RT_GUARD_EVENT(counters, sp, stack_cap);
                node = nearc;
                continue;
            }
            if (h0) { node = ch.x; continue; }
            if (h1) { node = ch.y; continue; }
        } else {
            int leaf = ~node, first = leaf >> 3, cnt = (leaf & 7) + 1;
            for (int k = 0; k < cnt; k++) {
                float4 a = sc.bvh_tri[3 * (size_t)(first + k)];
                F3 e1 = f3(sc.bvh_tri[3 * (size_t)(first + k) + 1]), e2 = f3(sc.bvh_tri[3 * (size_t)(first + k) + 2]);
                float t;
                if (tri_raycast(f3(a), e1, e2, o, d, t)) {
                    int id = sc.n_analytic + __float_as_int(a.w);
                    if (best_t > t || (best_t == t && id < best_id)) { best_t = t; best_id = id; }
                }
            }
            if (COUNT) tl.tri += (unsigned)cnt;
        }
        sp--;
        node = stack[sp * BLOCK];
    }
}

// ------------------------------------------------------------------ shadow query (main.cpp:181-188)
// True when ANY non-light primitive's Raycast accepts the ray, at any distance.
template <bool COUNT>
__device__ __forceinline__ bool brute_occluded(const DevScene &sc, F3 o, F3 d, Tally &tl)
{
    for (int k = 0; k < sc.n_tris; k++) {
        F3 v1 = f3(sc.tri_mt[3 * k]), e1 = f3(sc.tri_mt[3 * k + 1]), e2 = f3(sc.tri_mt[3 * k + 2]);
        float t;
        if (COUNT) tl.tri++;
        if (tri_raycast(v1, e1, e2, o, d, t)) return true;
    }
    return false;
}

template <bool COUNT>
__device__ __forceinline__ bool bvh_occluded(const DevScene &sc, F3 o, F3 d, bool live, int *stack, int stack_cap, Tally &tl,
                                             FrameCounters *counters)
{
    F3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int sp = 1;
    stack[0] = NODE_SENTINEL;
    int node = live ? 0 : NODE_SENTINEL;
    bool occ = false;
    while (node != NODE_SENTINEL) {
        if (node >= 0) {
            const BvhNode *np = sc.nodes + node;
            float4 xy0 = np->xy0, xy1 = np->xy1, z01 = np->z01;
            int4 ch = np->ch;
            float tn0, tn1;
            bool h0 = slab(xy0.x, xy0.y, xy0.z, xy0.w, z01.x, z01.y, o, inv, FLT_MAX, tn0);
            bool h1 = slab(xy1.x, xy1.y, xy1.z, xy1.w, z01.z, z01.w, o, inv, FLT_MAX, tn1);
            if (COUNT) tl.box += 2;
            if (h0 && h1) {
                if (sp < stack_cap) { stack[sp * BLOCK] = ch.y; sp++; }
                else RT_GUARD_EVENT(counters, sp, stack_cap);
                node = ch.x;
                continue;
            }
            if (h0) { node = ch.x; continue; }
            if (h1) { node = ch.y; continue; }
        } else {
            int leaf = ~node, first = leaf >> 3, cnt = (leaf & 7) + 1;
            for (int k = 0; k < cnt; k++) {
                F3 v1 = f3(sc.bvh_tri[3 * (size_t)(first + k)]);
                F3 e1 = f3(sc.bvh_tri[3 * (size_t)(first + k) + 1]), e2 = f3(sc.bvh_tri[3 * (size_t)(first + k) + 2]);
                float t;
                if (tri_raycast(v1, e1, e2, o, d, t)) occ = true;
            }
            if (COUNT) tl.tri += (unsigned)cnt;
            if (occ) break;
        }
        sp--;
        node = stack[sp * BLOCK];
    }
    return occ;
}

// ------------------------------------------------------------------ bounce kernel
struct BounceArgs {
    DevScene sc;
    DevCamera cam;
    int W, n_rows, n_pixels;
    rt_rowset rows;
    int level, max_depth;
    unsigned int n_slots;            // level 0: padded 8x8-tile slot count
    float4 *records;                 // [level][pixel]
    float *hit_t;                    // level 0 only, may be null
    float *normal;                   // level 0 only, may be null
    float *uv;                       // level 0 only, may be null
    const float4 *q_in_o, *q_in_d;   // level > 0
    float4 *q_out_o, *q_out_d;
    const unsigned int *counts_in;    // level > 0: rays in each 64-slot batch of the input queue
    unsigned int *counts_out;         // rays this launch leaves in each batch of the output queue
    unsigned int n_batches;
    FrameCounters *counters;
    int stack_cap;
};

template <int ACCEL, bool PRIMARY, bool COUNT, int STACK>
__global__ __launch_bounds__(BLOCK) void k_bounce(BounceArgs A)
{
    __shared__ int s_stack[(ACCEL ? (STACK + 1) : 1) * BLOCK];
    int *stack = s_stack + threadIdx.x;
    const DevScene &sc = A.sc;
    const int lane = threadIdx.x & 63;
    Tally tl = { 0u, 0u, 0u };
    // Batches of 64 slots are dealt to the waves statically, and the rays a batch spawns go to the same
    // batch of the next level's queue (dense from its first slot, with a count): no cursor and no queue
    // tail to contend for — atomics on one address retire at ~90 per microsecond on this chip, which
    // would cost more than the whole linear scan of a small scene.
    const unsigned int wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), n_waves = gridDim.x * (BLOCK / 64);
    const bool more = A.level < A.max_depth;
    const int stack_cap = (A.stack_cap < STACK ? A.stack_cap : STACK) + 1;    // never past the LDS column, whatever the host passed

    for (unsigned int batch = wave_id; batch < A.n_batches; batch += n_waves) {
        const unsigned int cnt = PRIMARY ? 64u : A.counts_in[batch];
        if (cnt == 0u) {
            if (more && lane == 0) A.counts_out[batch] = 0u;
            continue;
        }
        const unsigned int slot = batch * 64u + (unsigned)lane;

        // ---- this lane's ray
        bool live;
        int pixel;
        F3 o, d;
        if (PRIMARY) {
            const int tiles_per_row = (A.W + 7) >> 3;
            const int tile = (int)(slot >> 6), w = (int)(slot & 63u);
            const int x = (tile % tiles_per_row) * 8 + (w & 7);
            const int r = (tile / tiles_per_row) * 8 + (w >> 3);
            live = x < A.W && r < A.n_rows;
            pixel = r * A.W + x;
            const int y = (A.rows.first_band + (r / A.rows.band_rows) * A.rows.band_stride) * A.rows.band_rows +
                          r % A.rows.band_rows;
            o = mk(A.cam.ox, A.cam.oy, A.cam.oz);                               // main.cpp:267
            float offsetX = (float)x * A.cam.pitch + A.cam.off_x;               // main.cpp:268
            float offsetY = (float)y * A.cam.pitch + A.cam.off_y;               // main.cpp:269
            d = normalize(sub(mk(offsetX, offsetY, A.cam.plane_z), o));         // main.cpp:270
        } else {
            live = (unsigned)lane < cnt;
            float4 qo = live ? A.q_in_o[slot] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 qd = live ? A.q_in_d[slot] : make_float4(0.f, 0.f, 1.f, 0.f);
            o = f3(qo);
            d = f3(qd);
            pixel = __float_as_int(qo.w);
        }

        // ---- nearest hit
        float best_t = FLT_MAX;
        int best_id = -1;
        if (live) analytic_nearest<COUNT>(sc, o, d, best_t, best_id, tl);
        if (ACCEL == 1) {
            bvh_nearest<COUNT>(sc, o, d, live, best_t, best_id, stack, stack_cap, tl, A.counters);
        } else if (live) {
            brute_nearest<COUNT>(sc, o, d, best_t, best_id, tl);
        }

        // ---- surface (main.cpp:163-164)
        const bool hit = live && best_id >= 0;
        F3 pos = mk(0.f, 0.f, 0.f);
        Surface s;
        s.N = mk(0.f, 0.f, 1.f); s.Mc = mk(0.f, 0.f, 0.f); s.kr = 0.f; s.kd = 0.f; s.ks = 0.f; s.emissive = false;
        if (hit) {
            pos = add(o, muls(d, best_t));                                      // main.cpp:163
            s = surface_at(sc, best_id, pos, o, d);                             // main.cpp:164
        }

        // ---- local shading (main.cpp:166-205)
        F3 col = mk(0.f, 0.f, 0.f);
        const bool lit = hit && !s.emissive;
        if (hit && s.emissive) col = s.Mc;                                      // main.cpp:166-167
        for (int li = 0; li < sc.n_lights; li++) {                              // main.cpp:170-173
            const DevAnalytic lt = sc.analytic[sc.lights[li]];
            F3 L = mk(0.f, 0.f, 1.f), so = pos;
            if (lit) {
                L = normalize(sub(mk(lt.px, lt.py, lt.pz), pos));               // main.cpp:178
                so = add(pos, muls(L, 0.01f));                                  // main.cpp:179
            }
            bool occ = false;
            if (lit) occ = analytic_occluded<COUNT>(sc, so, L, tl);
            if (ACCEL == 1) {
                bool go = lit && !occ;
                if (bvh_occluded<COUNT>(sc, so, L, go, stack, stack_cap, tl, A.counters)) occ = true;
            } else if (lit && !occ) {
                occ = brute_occluded<COUNT>(sc, so, L, tl);
            }
            if (lit) col = add_light(col, s, L, mk(lt.cr, lt.cg, lt.cb), d, occ ? 0.0f : 1.0f);   // main.cpp:175, 185, 191-204
        }

        // ---- record + reflection ray (main.cpp:208-211)
        if (live) {
            A.records[(size_t)A.level * (size_t)A.n_pixels + (size_t)pixel] =
                make_float4(col.x, col.y, col.z, __int_as_float(best_id));
            if (PRIMARY && A.hit_t) A.hit_t[pixel] = best_t;
            if (PRIMARY && A.normal) {
                const F3 Nn = hit ? s.N : mk(0.f, 0.f, 0.f);
                A.normal[3 * (size_t)pixel] = Nn.x; A.normal[3 * (size_t)pixel + 1] = Nn.y; A.normal[3 * (size_t)pixel + 2] = Nn.z;
            }
            if (PRIMARY && A.uv) {
                const float2 tc = hit ? hit_uv(sc, best_id, o, d) : make_float2(0.f, 0.f);
                A.uv[2 * (size_t)pixel] = tc.x; A.uv[2 * (size_t)pixel + 1] = tc.y;
            }
        }
        const bool spawn = lit && s.kr > 0.0f && more;
        const unsigned long long mask = __ballot(spawn);
        if (spawn) {
            const F3 R = mirror_dir(d, s.N);                                    // main.cpp:210
            const unsigned int dst = batch * 64u + (unsigned)__popcll(mask & lanemask_lt());
            A.q_out_o[dst] = make_float4(pos.x, pos.y, pos.z, __int_as_float(pixel));       // origin is pos itself
            A.q_out_d[dst] = make_float4(R.x, R.y, R.z, 0.f);
        }
        if (more && lane == 0) A.counts_out[batch] = (unsigned)__popcll(mask);
    }

    // ---- test counters (COUNT builds): wave reduce, one atomic per wave.  Ray counts come from k_resolve.
    if (COUNT) {
        unsigned int vals[3] = { tl.box, tl.tri, tl.analytic };
        for (int k = 0; k < 3; k++) {
            unsigned long long v = vals[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && v)
                atomicAdd(k == 0 ? &A.counters->box_tests : k == 1 ? &A.counters->tri_tests : &A.counters->analytic_tests, v);
        }
    }
}

// ------------------------------------------------------------------ resolve
struct ResolveArgs {
    DevScene sc;
    int n_pixels, max_depth;
    const float4 *records;
    uint8_t *rgb8;      // may be null
    int32_t *hit_id;    // may be null
    float *rgb32;       // may be null
    FrameCounters *counters;
};

__global__ __launch_bounds__(BLOCK) void k_resolve(ResolveArgs A)
{
    __shared__ unsigned int s_stat[2];
    const DevScene &sc = A.sc;
    if (threadIdx.x < 2) s_stat[threadIdx.x] = 0;
    __syncthreads();
    unsigned int n_shadow = 0, n_reflect = 0;
    for (size_t p = (size_t)blockIdx.x * BLOCK + threadIdx.x; p < (size_t)A.n_pixels; p += (size_t)gridDim.x * BLOCK) {
        // Find the deepest level that exists for this pixel: a level k+1 record exists
        // iff the level-k hit reflected (same test as the spawn in k_bounce).  Every hit that is
        // not a light cast one shadow ray per light (main.cpp:170-188).
        int last = 0;
        int id0 = -1;
        for (int k = 0;; k++) {
            float4 r = A.records[(size_t)k * (size_t)A.n_pixels + p];
            int id = __float_as_int(r.w);
            if (k == 0) id0 = id;
            last = k;
            if (id < 0) break;
            float kr;
            bool emissive;
            if (id < sc.n_analytic) { kr = sc.analytic[id].reflect; emissive = sc.analytic[id].is_light != 0; }
            else { kr = sc.mats[__float_as_int(sc.tri_n[id - sc.n_analytic].w)].reflect; emissive = false; }
            if (emissive) break;
            n_shadow += (unsigned)sc.n_lights;
            if (!(kr > 0.0f) || k >= A.max_depth) break;
            n_reflect++;
        }
        float4 r = A.records[(size_t)last * (size_t)A.n_pixels + p];
        F3 col = mk(r.x, r.y, r.z);
        for (int k = last - 1; k >= 0; k--) {
            float4 rk = A.records[(size_t)k * (size_t)A.n_pixels + p];
            int id = __float_as_int(rk.w);
            F3 Mc;
            float kr;
            if (id < sc.n_analytic) {
                const DevAnalytic a = sc.analytic[id];
                Mc = mk(a.cr, a.cg, a.cb); kr = a.reflect;
            } else {
                const DevMaterial m = sc.mats[__float_as_int(sc.tri_n[id - sc.n_analytic].w)];
                Mc = mk(m.cr, m.cg, m.cb); kr = m.reflect;
            }
            col = add(mk(rk.x, rk.y, rk.z), muls(mulv(col, Mc), kr));               // main.cpp:212
        }
        if (A.rgb32) { A.rgb32[3 * p] = col.x; A.rgb32[3 * p + 1] = col.y; A.rgb32[3 * p + 2] = col.z; }
        if (A.rgb8) {                                                               // main.cpp:273
            A.rgb8[3 * p] = quantise8(col.x);
            A.rgb8[3 * p + 1] = quantise8(col.y);
            A.rgb8[3 * p + 2] = quantise8(col.z);
        }
        if (A.hit_id) A.hit_id[p] = id0;
    }
    // ray counters: wave reduce -> LDS -> one atomic per block and counter
    unsigned int vals[2] = { n_shadow, n_reflect };
    for (int k = 0; k < 2; k++) {
        unsigned int v = vals[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_stat[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_stat[threadIdx.x])
        atomicAdd(threadIdx.x == 0 ? &A.counters->rays_shadow : &A.counters->rays_reflect, (unsigned long long)s_stat[threadIdx.x]);
}

// ------------------------------------------------------------------ band de-interleave (after the gather)
__global__ void k_deinterleave(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int W, int H, int band_rows,
                               int n_parts, int rows_per_part, int bpp)
{
    // one thread per destination byte of the W x H image
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t row_bytes = (size_t)W * bpp;
    if (i >= row_bytes * (size_t)H) return;
    int y = (int)(i / row_bytes);
    size_t xb = i % row_bytes;
    int band = y / band_rows, part = band % n_parts, local_band = band / n_parts;
    int r = local_band * band_rows + y % band_rows;
    dst[i] = src[((size_t)part * rows_per_part + r) * row_bytes + xb];
}

// The same scatter 16 bytes per thread, for rows whose length and addresses are multiples of 16 (every row of a part and
// of the image then starts 16-byte aligned): one 128-bit load and store instead of sixteen byte accesses.
__global__ void k_deinterleave16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int row_vecs, int H, int band_rows,
                                 int n_parts, int rows_per_part)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)row_vecs * (size_t)H) return;
    int y = (int)(i / (size_t)row_vecs);
    size_t xv = i % (size_t)row_vecs;
    int band = y / band_rows, part = band % n_parts, local_band = band / n_parts;
    int r = local_band * band_rows + y % band_rows;
    dst[i] = src[((size_t)part * rows_per_part + r) * (size_t)row_vecs + xv];
}

template <int ACCEL, bool PRIMARY, bool COUNT, int STACK>
void launch_bounce(const BounceArgs &a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((k_bounce<ACCEL, PRIMARY, COUNT, STACK>), dim3(grid), dim3(BLOCK), 0, stream, a);
}

template <int ACCEL, int STACK>
void launch_bounce_sel(const BounceArgs &a, bool primary, bool count, int grid, hipStream_t stream)
{
    if (primary) {
        if (count) launch_bounce<ACCEL, true, true, STACK>(a, grid, stream);
        else launch_bounce<ACCEL, true, false, STACK>(a, grid, stream);
    } else {
        if (count) launch_bounce<ACCEL, false, true, STACK>(a, grid, stream);
        else launch_bounce<ACCEL, false, false, STACK>(a, grid, stream);
    }
}

}  // namespace

int g_num_cus = 0;

int device_num_cus(std::string *err)
{
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        RT_HIP_CHECK(hipGetDevice(&dev), err);
        RT_HIP_CHECK(hipGetDeviceProperties(&prop, dev), err);
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_num_cus;
}

int trace_frame(const DevScene &scene, const DevCamera &cam, const FrameParams &fp, FrameWorkspace &ws, void *d_rgb8,
                void *d_hit_id, void *d_rgb32, void *d_hit_t, hipStream_t stream, int *n_launches, std::string *err)
{
    const int n_rows = fp.rows.n_rows;
    const int n_pixels = fp.W * n_rows;
    if (device_num_cus(err) <= 0) return RT_E_DEVICE;
    const int pipe = scene.accel == 1 ? lbvh_pipeline(fp.pipeline, scene.n_lights) : 1;
    const bool wavefront = pipe == 2;
    {
        // one memset for everything a frame expects to find zero: the counters and, for the staged pipeline, the
        // claim cursors of all its traversal launches (they sit behind the counters in one allocation)
        size_t clear = (sizeof(FrameCounters) + 15) & ~(size_t)15;
        if (wavefront) clear = COUNTERS_PAD + sizeof(unsigned int) * wavefront_cursor_words(fp.max_depth + 1, (scene.n_lights + 31) / 32);
        // (one launch of our own: hipMemsetAsync splits a range like this one into three fill kernels, ~14 us where a
        // rank's eighth of a frame takes 650)
        const unsigned int words = (unsigned int)(clear / 4);
        hipLaunchKernelGGL(k_zero_words, dim3((words + 1023) / 1024 < 256 ? (words + 1023) / 1024 : 256), dim3(256), 0, stream,
                           (unsigned int *)ws.counters, words);
    }

    BounceArgs a;
    a.sc = scene;
    a.cam = cam;
    a.W = fp.W; a.n_rows = n_rows; a.n_pixels = n_pixels;
    a.rows = fp.rows;
    a.max_depth = fp.max_depth;
    const int tiles = ((fp.W + 7) / 8) * ((n_rows + 7) / 8);
    a.n_slots = (unsigned)tiles * 64u;
    a.records = ws.records;
    a.hit_t = (float *)d_hit_t;
    a.normal = fp.d_normal;
    a.uv = fp.d_uv;
    a.counters = ws.counters;
    a.stack_cap = fp.stack_depth;
#if RT_GUARD
    // guarded builds only: a deliberately short stack, to show that a dropped child is reported
    if (const char *e = getenv("RT_GUARD_STACK_CAP")) a.stack_cap = atoi(e) > 0 ? atoi(e) : a.stack_cap;
#endif

    const bool timing = fp.timing != 0;
    size_t ev = 0;
    auto stamp = [&]() -> int {
        if (!timing) return RT_OK;
        if (ev >= ws.events.size()) {
            hipEvent_t e;
            RT_HIP_CHECK(hipEventCreate(&e), err);
            ws.events.push_back(e);
        }
        RT_HIP_CHECK(hipEventRecord(ws.events[ev++], stream), err);
        return RT_OK;
    };

    int launches = 0;
    const int full_grid = g_num_cus * 8;
    if (int rc = stamp()) return rc;
    if (wavefront) {
        std::function<int()> st = stamp;
        if (int rc = wavefront_frame(scene, cam, fp, ws, d_rgb8, d_hit_id, d_rgb32, d_hit_t, stream, g_num_cus, st, &launches, err))
            return rc;
        ws.n_events = (int)ev;
        if (n_launches) *n_launches = launches;
        return RT_OK;
    }
    for (int level = 0; level <= fp.max_depth && !wavefront; level++) {
        a.level = level;
        a.q_in_o = ws.queue_o[level & 1];
        a.q_in_d = ws.queue_d[level & 1];
        a.q_out_o = ws.queue_o[(level + 1) & 1];
        a.q_out_d = ws.queue_d[(level + 1) & 1];
        a.counts_in = ws.fq_counts[level & 1];
        a.counts_out = ws.fq_counts[(level + 1) & 1];
        a.n_batches = a.n_slots / 64u;
        int grid = full_grid;
        const int need = (int)((a.n_slots + BLOCK - 1) / BLOCK);
        if (need < grid) grid = need;
        const bool primary = level == 0, count = fp.count_tests != 0;
        if (scene.accel == 1) {
            if (fp.stack_depth <= 32) launch_bounce_sel<1, 32>(a, primary, count, grid, stream);
            else launch_bounce_sel<1, 64>(a, primary, count, grid, stream);
        } else {
            launch_bounce_sel<0, 1>(a, primary, count, grid, stream);
        }
        launches++;
        if (int rc = stamp()) return rc;
    }
    ResolveArgs r;
    r.sc = scene;
    r.n_pixels = n_pixels;
    r.max_depth = fp.max_depth;
    r.records = ws.records;
    r.rgb8 = (uint8_t *)d_rgb8;
    r.hit_id = (int32_t *)d_hit_id;
    r.rgb32 = (float *)d_rgb32;
    r.counters = ws.counters;
    int rgrid = (n_pixels + BLOCK - 1) / BLOCK;
    if (rgrid > g_num_cus * 8) rgrid = g_num_cus * 8;
    hipLaunchKernelGGL(k_resolve, dim3(rgrid), dim3(BLOCK), 0, stream, r);
    launches++;
    if (int rc = stamp()) return rc;
    RT_HIP_CHECK(hipGetLastError(), err);
    ws.n_events = (int)ev;
    if (n_launches) *n_launches = launches;
    return RT_OK;
}

int trace_deinterleave(const void *src, void *dst, int W, int H, int band_rows, int n_parts, int rows_per_part, int bpp,
                       hipStream_t stream)
{
    size_t total = (size_t)W * H * bpp;
    if (total == 0) return RT_OK;
    const size_t row_bytes = (size_t)W * bpp;
    if (row_bytes % 16 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0) {
        const size_t vecs = total / 16;
        hipLaunchKernelGGL(k_deinterleave16, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, stream, (const uint4 *)src, (uint4 *)dst,
                           (int)(row_bytes / 16), H, band_rows, n_parts, rows_per_part);
        return hipGetLastError() == hipSuccess ? RT_OK : RT_E_DEVICE;
    }
    hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const uint8_t *)src,
                       (uint8_t *)dst, W, H, band_rows, n_parts, rows_per_part, bpp);
    return hipGetLastError() == hipSuccess ? RT_OK : RT_E_DEVICE;
}
