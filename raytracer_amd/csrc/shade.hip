// shade.hip — the last kernel of an LBVH frame: every level's local colour, the fold and the outputs.
//
// The traversal stages (wavefront.hip) leave, per pixel and bounce level, the nearest hit (t, id) and one
// occlusion bit per light.  k_shade_all runs one thread per pixel: it walks the pixel's chain of hits, recomputes
// each ray exactly as k_shadow_gen spawned it, shades every level in the reference's accumulation order
// (Raytracer/main.cpp:166-205), folds the levels deepest-first the way the recursion unwinds (main.cpp:212), then
// clamps and quantises (main.cpp:139-142, 273).
#include "rt_device.h"

namespace {

// ------------------------------------------------------------------ shading of every level + fold
struct ShadeAllArgs {
    DevScene sc;
    DevCamera cam;
    int W, n_rows, n_pixels, max_depth;
    rt_rowset rows;
    const float2 *hits;               // [level][pixel]
    const unsigned int *occ;          // [level][chunk][pixel], 32 lights per chunk
    int n_chunks;
    float4 *records;                  // [level][pixel] scratch: (local colour, hit id bits)
    uint8_t *rgb8; int32_t *hit_id; float *rgb32; float *hit_t; float *normal; float *uv;
    FrameCounters *counters;
};

// One level of one pixel's chain: the local colour of main.cpp:159-205 for the ray (o, d) whose nearest hit is
// stored in plane k.  Returns whether the chain goes on (main.cpp:208) and, if so, the next ray (pos, R).
struct LevelOut { F3 col, pos, R, Mc; float kr; int id; bool spawn; };

// h, occ0: the level's hit and its first 32 occlusion bits, loaded by the caller (all levels' at once, ahead of the walk).
__device__ __forceinline__ LevelOut shade_level(const ShadeAllArgs &A, size_t p, int k, F3 o, F3 d, int &id0, unsigned int &n_shadow, float2 h,
                                                unsigned int occ0)
{
    const DevScene &sc = A.sc;
    LevelOut out;
    const float t = h.x;
    int id = __float_as_int(h.y);
    if (id >= sc.n_analytic + sc.n_tris) id = -1;      // never index scene arrays with an id outside the scene
    if (k == 0) {
        id0 = id;
        if (A.hit_t) A.hit_t[p] = t;
        if (A.normal && id < 0) { A.normal[3 * p] = 0.f; A.normal[3 * p + 1] = 0.f; A.normal[3 * p + 2] = 0.f; }
        if (A.uv) {
            const float2 tc = id >= 0 ? hit_uv(sc, id, o, d) : make_float2(0.f, 0.f);
            A.uv[2 * p] = tc.x; A.uv[2 * p + 1] = tc.y;
        }
    }
    F3 col = mk(0.f, 0.f, 0.f);                                                   // main.cpp:159-160
    out.spawn = false;
    out.pos = o; out.R = d;
    out.Mc = mk(0.f, 0.f, 0.f); out.kr = 0.f;
    if (id >= 0) {
        const F3 pos = add(o, muls(d, t));                                        // main.cpp:163
        const Surface s = surface_at(sc, id, pos, o, d);                          // main.cpp:164
        out.pos = pos; out.Mc = s.Mc; out.kr = s.kr;
        if (k == 0 && A.normal) { A.normal[3 * p] = s.N.x; A.normal[3 * p + 1] = s.N.y; A.normal[3 * p + 2] = s.N.z; }
        if (s.emissive) {
            col = s.Mc;                                                           // main.cpp:166-167
        } else {
            n_shadow += (unsigned)sc.n_lights;
            unsigned int bits = 0;
            for (int li = 0; li < sc.n_lights; li++) {                            // main.cpp:170-173
                if ((li & 31) == 0) bits = li == 0 ? occ0 : A.occ[((size_t)k * A.n_chunks + (li >> 5)) * (size_t)A.n_pixels + p];
                const DevAnalytic lt_ = sc.analytic[sc.lights[li]];
                const F3 L = normalize(sub(mk(lt_.px, lt_.py, lt_.pz), pos));     // main.cpp:178
                const float shade = ((bits >> (li & 31)) & 1u) ? 0.0f : 1.0f;     // main.cpp:175, 185
                col = add_light(col, s, L, mk(lt_.cr, lt_.cg, lt_.cb), d, shade);                    // main.cpp:191-204
            }
            out.spawn = s.kr > 0.0f && k < A.max_depth;                           // main.cpp:208
            if (out.spawn) out.R = mirror_dir(d, s.N);
        }
    }
    out.col = col; out.id = id;
    return out;
}

// REG_LEVELS > 0: the frame has at most that many levels and a pixel's local colours stay in registers
// between the walk down its chain and the fold back up; 0: any depth, through the records plane.
template <int REG_LEVELS>
__global__ __launch_bounds__(BLOCK) void k_shade_all(ShadeAllArgs A)
{
    __shared__ unsigned int s_stat[2];
    const DevScene &sc = A.sc;
    if (threadIdx.x < 2) s_stat[threadIdx.x] = 0;
    __syncthreads();
    unsigned int n_shadow = 0, n_reflect = 0;
    for (size_t p = (size_t)blockIdx.x * BLOCK + threadIdx.x; p < (size_t)A.n_pixels; p += (size_t)gridDim.x * BLOCK) {
        const int x = (int)(p % (size_t)A.W), lr = (int)(p / (size_t)A.W);
        const int y = (A.rows.first_band + (lr / A.rows.band_rows) * A.rows.band_stride) * A.rows.band_rows + lr % A.rows.band_rows;
        F3 o = mk(A.cam.ox, A.cam.oy, A.cam.oz);                                          // main.cpp:267
        const float offsetX = (float)x * A.cam.pitch + A.cam.off_x;                      // main.cpp:268
        const float offsetY = (float)y * A.cam.pitch + A.cam.off_y;                      // main.cpp:269
        F3 d = normalize(sub(mk(offsetX, offsetY, A.cam.plane_z), o));                    // main.cpp:270
        int last = 0, id0 = -1;
        F3 col;
        if (REG_LEVELS > 0) {
            constexpr int NL = REG_LEVELS > 0 ? REG_LEVELS : 1;
            F3 lc[NL], lm[NL];
            float lk[NL];
            bool alive = true;
            // Every level's hit and occlusion word of this pixel, requested at once: the walk below would otherwise wait
            // for them one level at a time, each behind the previous level's arithmetic.  (Planes of levels the chain does
            // not reach hold an earlier frame's values: read, never used.)
            float2 hk[NL];
            unsigned int ok[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const bool in = k <= A.max_depth;
                hk[k] = in ? A.hits[(size_t)k * (size_t)A.n_pixels + p] : make_float2(FLT_MAX, __int_as_float(-1));
                ok[k] = (in && sc.n_lights > 0) ? A.occ[(size_t)k * A.n_chunks * (size_t)A.n_pixels + p] : 0u;
            }
#pragma unroll
            for (int k = 0; k < NL; k++) {
                lc[k] = mk(0.f, 0.f, 0.f); lm[k] = mk(0.f, 0.f, 0.f); lk[k] = 0.f;
                if (alive) {
                    const LevelOut L = shade_level(A, p, k, o, d, id0, n_shadow, hk[k], ok[k]);
                    lc[k] = L.col; lm[k] = L.Mc; lk[k] = L.kr;
                    last = k;
                    alive = L.spawn;
                    if (alive) { n_reflect++; o = L.pos; d = L.R; }                       // main.cpp:211
                }
            }
            // fold deepest-first: col_k = local_k + (col_{k+1} * Mc_k) * kr_k           (main.cpp:212)
            col = mk(0.f, 0.f, 0.f);
#pragma unroll
            for (int k = NL - 1; k >= 0; k--) {
                if (k == last) col = lc[k];
                else if (k < last) col = add(lc[k], muls(mulv(col, lm[k]), lk[k]));
            }
        } else {
            for (int k = 0;; k++) {
                const LevelOut L = shade_level(A, p, k, o, d, id0, n_shadow, A.hits[(size_t)k * (size_t)A.n_pixels + p],
                                               sc.n_lights > 0 ? A.occ[(size_t)k * A.n_chunks * (size_t)A.n_pixels + p] : 0u);
                last = k;
                A.records[(size_t)k * (size_t)A.n_pixels + p] = make_float4(L.col.x, L.col.y, L.col.z, __int_as_float(L.id));
                if (!L.spawn) break;
                n_reflect++;
                o = L.pos; d = L.R;                                                       // main.cpp:211
            }
            // fold deepest-first: col_k = local_k + (col_{k+1} * Mc_k) * kr_k           (main.cpp:212)
            const float4 rl = A.records[(size_t)last * (size_t)A.n_pixels + p];
            col = mk(rl.x, rl.y, rl.z);
            for (int k = last - 1; k >= 0; k--) {
                const float4 rk = A.records[(size_t)k * (size_t)A.n_pixels + p];
                const int id = __float_as_int(rk.w);
                F3 Mc;
                float kr;
                if (id < sc.n_analytic) {
                    const DevAnalytic a = sc.analytic[id];
                    Mc = mk(a.cr, a.cg, a.cb); kr = a.reflect;
                } else {
                    const DevMaterial m = sc.mats[__float_as_int(sc.tri_n[id - sc.n_analytic].w)];
                    Mc = mk(m.cr, m.cg, m.cb); kr = m.reflect;
                }
                col = add(mk(rk.x, rk.y, rk.z), muls(mulv(col, Mc), kr));
            }
        }
        if (A.rgb32) { A.rgb32[3 * p] = col.x; A.rgb32[3 * p + 1] = col.y; A.rgb32[3 * p + 2] = col.z; }
        if (A.rgb8) { A.rgb8[3 * p] = quantise8(col.x); A.rgb8[3 * p + 1] = quantise8(col.y); A.rgb8[3 * p + 2] = quantise8(col.z); }
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

}  // namespace

int shade_all_launch(const DevScene &scene, const DevCamera &cam, const FrameParams &fp, const float2 *hits,
                     const unsigned int *occ, int n_chunks, float4 *records, FrameCounters *counters, void *d_rgb8, void *d_hit_id,
                     void *d_rgb32, void *d_hit_t, hipStream_t stream, int num_cus, std::string *err)
{
    ShadeAllArgs s;
    s.sc = scene; s.cam = cam; s.W = fp.W; s.n_rows = fp.rows.n_rows; s.n_pixels = fp.W * fp.rows.n_rows; s.max_depth = fp.max_depth;
    s.rows = fp.rows;
    s.hits = hits; s.occ = occ; s.n_chunks = n_chunks; s.records = records;
    s.rgb8 = (uint8_t *)d_rgb8; s.hit_id = (int32_t *)d_hit_id; s.rgb32 = (float *)d_rgb32; s.hit_t = (float *)d_hit_t;
    s.normal = fp.d_normal;
    s.uv = fp.d_uv;
    s.counters = counters;
    int sgrid = (s.n_pixels + BLOCK - 1) / BLOCK;
    if (sgrid > num_cus * 8) sgrid = num_cus * 8;
    if (fp.max_depth + 1 <= 5) hipLaunchKernelGGL(k_shade_all<5>, dim3(sgrid), dim3(BLOCK), 0, stream, s);
    else if (fp.max_depth + 1 <= SHADE_REG_LEVELS) hipLaunchKernelGGL(k_shade_all<SHADE_REG_LEVELS>, dim3(sgrid), dim3(BLOCK), 0, stream, s);
    else hipLaunchKernelGGL(k_shade_all<0>, dim3(sgrid), dim3(BLOCK), 0, stream, s);
    RT_HIP_CHECK(hipGetLastError(), err);
    return RT_OK;
}

