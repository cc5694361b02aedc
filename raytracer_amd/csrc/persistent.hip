// persistent.hip — the whole ray tree of a frame in ONE traversal launch.
//
// The staged pipeline in wavefront.hip launches one traversal kernel per bounce level and ray
// kind; every launch ends with a drain in which each wave waits for the longest of its last
// rays, and at 1080p those drains are a third of the frame.  Here a frame is two launches:
//
//   k_stream     persistent waves; every lane owns one ray of any kind.  A lane whose nearest-hit
//                ray (Raytracer/main.cpp:147-156) has finished evaluates the hit point, pushes the
//                reflection ray (main.cpp:208-211) and one shadow ray per light (main.cpp:178-188,
//                analytic occluders tested on the spot) onto its WAVE's private ray stack, and idle
//                lanes refill from that stack before the wave claims new camera rays.  Rays never
//                cross waves, so there are no queue atomics; results are the nearest hit per
//                (level, pixel) and one occlusion bit per (level, pixel, light).
//   k_shade_all  one thread per pixel walks its chain of hits, recomputes each ray exactly as
//                k_stream generated it, shades every level in the reference's accumulation order
//                (main.cpp:166-205) and folds the levels deepest-first the way the recursion
//                unwinds (main.cpp:212), then clamps and quantises (main.cpp:139-142, 273).
//
// The private stack is LIFO with the reflection ray pushed BELOW the shadow rays of the same hit:
// shadow rays spawn nothing, so a wave never holds more than one hit's worth of rays per lane,
// 64 * (lights + 1) entries.  Scenes with more than 32 lights use the staged pipeline.
#include "rt_traverse.h"

#include <stdlib.h>

namespace {

constexpr int PS_LDS_STACK = 16;          // traversal stack entries per lane held in LDS
constexpr int PS_OVF = 56;                // further entries per lane in global memory
constexpr unsigned SENT_U = 0x7fffffffu;  // NODE_SENTINEL as unsigned: "(unsigned)node < SENT_U" == inner node
#ifndef PS_REFILL_MIN
#define PS_REFILL_MIN 16
#endif
#ifndef PS_WAVES_EU
#define PS_WAVES_EU 6
#endif
constexpr int PS_SHARDS = 64;             // camera-ray claim cursors, each on its own 128-byte line
constexpr int PS_CURSOR_STRIDE = 32;
constexpr int PS_MAX_PROBES = 8;
constexpr int PS_PARK = 128;              // parking list entries per wave (64 suffice; see persistent_stack_entries)
constexpr int INFO_SHADOW = 1 << 8;       // info word: level | shadow flag | light << 16

struct StreamArgs {
    DevScene sc;
    DevCamera cam;
    int W, n_rows, n_pixels, max_depth;
    rt_rowset rows;
    unsigned int n_slots;             // camera-ray slots (the frame padded to whole 8x8 tiles)
    float2 *hits;                     // [level][pixel] nearest (t, id bits)
    unsigned int *occ;                // [level][pixel] occlusion bit per light
    float4 *ls_o, *ls_d;              // [wave][ls_cap] private ray stacks: (origin, pixel) (direction, info)
    unsigned int ls_cap;
    unsigned int *cursors;            // [PS_SHARDS * PS_CURSOR_STRIDE], zeroed per frame
    int *overflow;                    // [PS_OVF][total_lanes]
    unsigned int total_lanes;
    FrameCounters *counters;
};

// main.cpp:210 — the mirror direction, evaluated as (N*2.0f)*(d.N)
__device__ __forceinline__ F3 mirror_dir(F3 d, F3 N) { return sub(d, muls(muls(N, 2.0f), dot(d, N))); }

template <bool COUNT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(PS_WAVES_EU, PS_WAVES_EU))) void k_stream(StreamArgs A)
{
    __shared__ int s_stack[PS_LDS_STACK * BLOCK];
    const int lane = threadIdx.x & 63;
    int *stack = s_stack + threadIdx.x;                                   // element k at stack[k * BLOCK]
    int *ovf = A.overflow + (size_t)blockIdx.x * BLOCK + threadIdx.x;     // element k at ovf[k * total_lanes]
    const DevScene &sc = A.sc;
    const unsigned long long lt = lanemask_lt();
    const size_t wave_id = (size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    float4 *ls_o = A.ls_o + wave_id * A.ls_cap, *ls_d = A.ls_d + wave_id * A.ls_cap;
    float4 *pk_o = ls_o + (A.ls_cap - PS_PARK), *pk_d = ls_d + (A.ls_cap - PS_PARK);   // parking list: the last PS_PARK entries
    const unsigned int stack_cap = A.ls_cap - PS_PARK;
    unsigned int ls_n = 0, park_n = 0;                                    // entries on the stack / parked (uniform)

    // ---- per-lane ray state
    int node = NODE_SENTINEL, sp = 0, best_id = -1, pix = 0, info = 0;
    bool gen_pending = false;          // a finished nearest-hit ray whose hit has not been expanded yet
    F3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);
    F3 inv = mk(0.f, 0.f, 1.f), oi = mk(0.f, 0.f, 0.f);
    uint3 sel_n = make_uint3(SEL_LO, SEL_LO, SEL_LO), sel_f = make_uint3(SEL_HI, SEL_HI, SEL_HI);
    float best_t = FLT_MAX;
    Tally tl = { 0u, 0u, 0u };
    unsigned int dbg[6] = { 0u, 0u, 0u, 0u, 0u, 0u };   // COUNT: node iters/lanes, leaf iters/lanes, refills/lanes
    unsigned int steps = 0;                             // COUNT: traversal steps of the current ray

    // ---- camera-ray claims (uniform): 64-slot units dealt round-robin to the shards
    unsigned int cur = 0, end = 0;
    const int n_shards = (int)min((unsigned)PS_SHARDS, gridDim.x);
    const int max_probes = min(PS_MAX_PROBES, n_shards);
    int shard = (int)(blockIdx.x % (unsigned)n_shards), probes = 0;
    const unsigned int n_units = A.n_slots >> 6;
    bool exhausted = n_units == 0;

    auto push = [&](int v) {
        if (sp < PS_LDS_STACK) stack[sp * BLOCK] = v;
        else ovf[(size_t)(sp - PS_LDS_STACK) * A.total_lanes] = v;
        sp++;
    };
    auto pop = [&]() {
        sp--;
        if (sp < PS_LDS_STACK) {
            node = stack[sp * BLOCK];
        } else {
            // explicit global load: as two plain C++ loads hipcc merges the LDS and the global branch
            // into one flat_load, and every pop then pays the flat path's latency
            const int *p = ovf + (size_t)(sp - PS_LDS_STACK) * A.total_lanes;
            int v;
            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            node = v;
        }
        if (node == NODE_SENTINEL && !(info & INFO_SHADOW)) gen_pending = true;
        if (COUNT && node == NODE_SENTINEL) {
            atomicAdd(&A.counters->step_hist[min(15, 32 - __clz((int)steps))], 1ull);
            atomicMax(&A.counters->max_steps, steps);
        }
    };
    auto start_ray = [&](bool shadow) {
        best_t = FLT_MAX; best_id = -1;
        if (COUNT) steps = 0;
        if (!shadow) analytic_nearest<COUNT>(sc, o, d, best_t, best_id, tl);
        slab_ray(sc.grid, o, d, inv, oi, sel_n, sel_f);
        stack[0] = NODE_SENTINEL;
        sp = 1;
        node = 0;
    };

    for (;;) {
        const unsigned long long idle0 = __ballot(node == NODE_SENTINEL);
        if (idle0 != 0ull && (__popcll(idle0) >= PS_REFILL_MIN || idle0 == ~0ull)) {
            // ---- (1) park finished nearest-hit rays: the hit goes to the hit plane, the ray to the wave's
            // parking list; the lane is free at once
            unsigned long long m = __ballot(gen_pending);
            if (m != 0ull) {
                // misses are not parked: nothing follows from them (main.cpp:159-160)
                const unsigned long long mh = __ballot(gen_pending && best_id >= 0);
                if (gen_pending) {
                    const int level = info & 0xff;
                    A.hits[(size_t)level * (size_t)A.n_pixels + (size_t)pix] = make_float2(best_t, __int_as_float(best_id));
                    const unsigned int k = park_n + (unsigned)__popcll(mh & lt);
                    if (best_id >= 0 && k < (unsigned)PS_PARK) {
                        pk_o[k] = make_float4(o.x, o.y, o.z, __int_as_float(pix));
                        pk_d[k] = make_float4(d.x, d.y, d.z, __int_as_float(level));
                    }
                    gen_pending = false;
                }
                park_n = min(park_n + (unsigned)__popcll(mh), (unsigned)PS_PARK);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            const unsigned int n_idle0 = (unsigned)__popcll(__ballot(node == NODE_SENTINEL));
            // ---- (2) expand parked hits at FULL wave width once a batch is complete, or as soon as the
            // ray stack cannot feed the idle lanes (so camera rays are only claimed when both are empty):
            // reflection ray (main.cpp:208-211) below one shadow ray per light (main.cpp:178-188)
            if (park_n >= 64u || (park_n > 0u && ls_n < n_idle0)) {
                const unsigned int n_take = min(park_n, 64u);
                bool lit = false, spawn = false;
                F3 pos = mk(0.f, 0.f, 0.f), R = mk(0.f, 0.f, 1.f);
                int ppix = 0, plevel = 0;
                if ((unsigned)lane < n_take) {
                    const unsigned int e = park_n - 1u - (unsigned)lane;
                    const float4 qo = pk_o[e], qd = pk_d[e];
                    ppix = __float_as_int(qo.w); plevel = __float_as_int(qd.w);
                    const float2 h = A.hits[(size_t)plevel * (size_t)A.n_pixels + (size_t)ppix];
                    const int id = __float_as_int(h.y);
                    if (id >= 0 && id < sc.n_analytic + sc.n_tris) {
                        pos = add(f3(qo), muls(f3(qd), h.x));                             // main.cpp:163
                        const Surface sf = surface_at(sc, id, pos, f3(qo), f3(qd));       // main.cpp:164
                        lit = !sf.emissive;                                               // main.cpp:166
                        spawn = lit && sf.kr > 0.0f && plevel < A.max_depth;              // main.cpp:208
                        if (spawn) R = mirror_dir(f3(qd), sf.N);
                    }
                }
                park_n -= n_take;
                m = __ballot(spawn);
                if (spawn) {
                    const unsigned int k = ls_n + (unsigned)__popcll(m & lt);
                    if (k < stack_cap) {
                        ls_o[k] = make_float4(pos.x, pos.y, pos.z, __int_as_float(ppix));  // origin is pos itself (main.cpp:211)
                        ls_d[k] = make_float4(R.x, R.y, R.z, __int_as_float(plevel + 1));
                    }
                }
                ls_n = min(ls_n + (unsigned)__popcll(m), stack_cap);
                unsigned int bits = 0;
                for (int li = 0; li < sc.n_lights; li++) {                                // main.cpp:170-173
                    const DevAnalytic lt_ = sc.analytic[sc.lights[li]];
                    bool want = false;
                    F3 L = mk(0.f, 0.f, 1.f), so = pos;
                    if (lit) {
                        L = normalize(sub(mk(lt_.px, lt_.py, lt_.pz), pos));               // main.cpp:178
                        so = add(pos, muls(L, 0.01f));                                    // main.cpp:179
                        if (analytic_occluded<COUNT>(sc, so, L, tl)) bits |= 1u << li;
                        else want = true;
                    }
                    m = __ballot(want);
                    if (want) {
                        const unsigned int k = ls_n + (unsigned)__popcll(m & lt);
                        if (k < stack_cap) {
                            ls_o[k] = make_float4(so.x, so.y, so.z, __int_as_float(ppix));
                            ls_d[k] = make_float4(L.x, L.y, L.z, __int_as_float(plevel | INFO_SHADOW | (li << 16)));
                        }
                    }
                    ls_n = min(ls_n + (unsigned)__popcll(m), stack_cap);
                }
                if (lit) A.occ[(size_t)plevel * (size_t)A.n_pixels + (size_t)ppix] = bits;
                // the stack entries and the occlusion words must have landed before lanes of this
                // wave read them back or add bits to them
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            // ---- (3) refill idle lanes from the private stack, newest first
            const unsigned long long idle = __ballot(node == NODE_SENTINEL);
            const unsigned int n_idle = (unsigned)__popcll(idle), rank = (unsigned)__popcll(idle & lt);
            const unsigned int n_local = min(n_idle, ls_n);
            if (node == NODE_SENTINEL && rank < n_local) {
                const unsigned int k = ls_n - 1u - rank;
                const float4 qo = ls_o[k], qd = ls_d[k];
                o = f3(qo); d = f3(qd);
                pix = __float_as_int(qo.w);
                info = __float_as_int(qd.w);
                start_ray((info & INFO_SHADOW) != 0);
            }
            ls_n -= n_local;
            if (COUNT && n_local) { dbg[4]++; dbg[5] += n_local; }
            // ---- (4) camera rays for the lanes still idle (stack and parking list are empty now)
            if (n_idle > n_local && !exhausted) {
                if (cur >= end) {
                    unsigned int k = 0;
                    if (lane == 0) k = atomicAdd(&A.cursors[shard * PS_CURSOR_STRIDE], 1u);
                    k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
                    const unsigned long long unit = (unsigned long long)k * (unsigned)n_shards + (unsigned)shard;
                    if (unit >= n_units) {
                        shard = shard + 1 == n_shards ? 0 : shard + 1;
                        if (++probes >= max_probes) exhausted = true;
                    } else {
                        cur = (unsigned)unit << 6;
                        end = cur + 64u;
                    }
                }
                if (cur < end) {
                    const unsigned int my = cur + (rank - n_local);
                    if (node == NODE_SENTINEL && rank >= n_local && my < end) {
                        const PixelRay pr = primary_ray(A.cam, A.W, A.n_rows, A.rows, my);
                        if (pr.live) {
                            o = pr.o; d = pr.d; pix = pr.pixel; info = 0;
                            start_ray(false);
                        }
                    }
                    if (COUNT) { dbg[4]++; dbg[5] += min(n_idle - n_local, end - cur); }
                    cur = min(end, cur + (n_idle - n_local));
                }
            }
        }

        const unsigned long long m_node = __ballot((unsigned)node < SENT_U);
        const unsigned long long m_leaf = __ballot(node < 0);
        if ((m_node | m_leaf) == 0ull) {
            if (exhausted && ls_n == 0u && park_n == 0u && __ballot(gen_pending) == 0ull) break;
            continue;
        }
        const bool do_node = __popcll(m_node) >= __popcll(m_leaf);
        if (COUNT && do_node) { dbg[0]++; dbg[1] += (unsigned)__popcll(m_node); }
        if (COUNT && !do_node) { dbg[2]++; dbg[3] += (unsigned)__popcll(m_leaf); }
        if (do_node) {
            // ---- inner-node step: both children's lattice boxes from one 32-byte node
            if ((unsigned)node < SENT_U) {
                if (COUNT) steps++;
                const BvhNodeQ *np = sc.qnodes + node;
                const uint4 qa = np->a, qb = np->b;
                const int2 ch = make_int2((int)qb.z, (int)qb.w);
                float n0, n1;
                bool h0, h1;
                node_test(qa, qb, inv, oi, sel_n, sel_f, best_t, n0, n1, h0, h1);     // shadow rays keep best_t = FLT_MAX
                if (COUNT) tl.box += 2;
                if (h0 && h1) {
                    const bool swap = n1 < n0;
                    push(swap ? ch.x : ch.y);
                    node = swap ? ch.y : ch.x;
                } else if (h0) {
                    node = ch.x;
                } else if (h1) {
                    node = ch.y;
                } else {
                    pop();
                }
            }
        } else {
            // ---- leaf step: every triangle of the leaf through the reference's Moller-Trumbore test
            if (node < 0) {
                if (COUNT) steps++;
                const int leaf = ~node;
                int tri_i = leaf >> 3, tri_left = (leaf & 7) + 1;
                const bool shadow = (info & INFO_SHADOW) != 0;
                bool blocked = false;
                for (;;) {
                    const float4 v1 = sc.bvh_tri[3 * (size_t)tri_i];
                    const F3 e1 = f3(sc.bvh_tri[3 * (size_t)tri_i + 1]), e2 = f3(sc.bvh_tri[3 * (size_t)tri_i + 2]);
                    float t;
                    const bool h = tri_raycast(f3(v1), e1, e2, o, d, t);
                    if (COUNT) tl.tri += 1;
                    tri_i++;
                    tri_left--;
                    if (h) {
                        if (shadow) { blocked = true; break; }     // main.cpp:183-186: any hit blocks the light
                        const int id = sc.n_analytic + __float_as_int(v1.w);
                        if (best_t > t || (best_t == t && id < best_id)) { best_t = t; best_id = id; }
                    }
                    if (tri_left == 0) break;
                }
                if (blocked) {
                    atomicOr(&A.occ[(size_t)(info & 0xff) * (size_t)A.n_pixels + (size_t)pix], 1u << (info >> 16));
                    node = NODE_SENTINEL;
                } else {
                    pop();
                }
            }
        }
    }

    if (COUNT) {
        unsigned int vals[3] = { tl.box, tl.tri, tl.analytic };
        for (int k = 0; k < 3; k++) {
            unsigned long long v = vals[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && v)
                atomicAdd(k == 0 ? &A.counters->box_tests : k == 1 ? &A.counters->tri_tests : &A.counters->analytic_tests, v);
        }
        if (lane == 0) {
            unsigned long long *dst = &A.counters->node_iters;
            for (int k = 0; k < 6; k++)
                if (dbg[k]) atomicAdd(dst + k, (unsigned long long)dbg[k]);
        }
    }
}

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

// (char)(clamp01(c)*255.0f) as the reference's Debug build evaluates it (main.cpp:139-142, 273):
// truncate toward zero, keep the low byte; NaN passes the clamp and converts to the low byte of INT_MIN = 0.
__device__ __forceinline__ uint8_t quantise8(float c)
{
    const float f = (c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c)) * 255.0f;
    return (f == f) ? (uint8_t)(int)f : (uint8_t)0;
}

// One level of one pixel's chain: the local colour of main.cpp:159-205 for the ray (o, d) whose nearest hit is
// stored in plane k.  Returns whether the chain goes on (main.cpp:208) and, if so, the next ray (pos, R).
struct LevelOut { F3 col, pos, R, Mc; float kr; int id; bool spawn; };

__device__ __forceinline__ LevelOut shade_level(const ShadeAllArgs &A, size_t p, int k, F3 o, F3 d, int &id0, unsigned int &n_shadow)
{
    const DevScene &sc = A.sc;
    LevelOut out;
    const float2 h = A.hits[(size_t)k * (size_t)A.n_pixels + p];
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
                if ((li & 31) == 0) bits = A.occ[((size_t)k * A.n_chunks + (li >> 5)) * (size_t)A.n_pixels + p];
                const DevAnalytic lt_ = sc.analytic[sc.lights[li]];
                const F3 L = normalize(sub(mk(lt_.px, lt_.py, lt_.pz), pos));     // main.cpp:178
                const float shade = ((bits >> (li & 31)) & 1u) ? 0.0f : 1.0f;     // main.cpp:175, 185
                const F3 Lc = mk(lt_.cr, lt_.cg, lt_.cb);
                if (s.kd > 0.0f) {                                                // main.cpp:191-195
                    const float diffuse = dot(s.N, L) * s.kd;
                    col = add(col, muls(muls(mulv(Lc, s.Mc), diffuse), shade));
                }
                if (s.ks > 0.0f) {                                                // main.cpp:198-204
                    const F3 Rl = sub(L, muls(muls(s.N, dot(L, s.N)), 2.0f));
                    const float dt = dot(d, Rl);
                    if (dt > 0.0f)
                        col = add(col, muls(muls(muls(mulv(Lc, s.Mc), pow20(dt)), s.ks), shade));
                }
            }
            out.spawn = s.kr > 0.0f && k < A.max_depth;                           // main.cpp:208
            if (out.spawn) out.R = mirror_dir(d, s.N);
        }
    }
    out.col = col; out.id = id;
    return out;
}

#ifndef SHADE_REG9
#define SHADE_REG9 1
#endif
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
#pragma unroll
            for (int k = 0; k < NL; k++) {
                lc[k] = mk(0.f, 0.f, 0.f); lm[k] = mk(0.f, 0.f, 0.f); lk[k] = 0.f;
                if (alive) {
                    const LevelOut L = shade_level(A, p, k, o, d, id0, n_shadow);
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
                const LevelOut L = shade_level(A, p, k, o, d, id0, n_shadow);
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

int blocks_per_cu()
{
    static int per_cu = 0;
    if (per_cu == 0) {
        const char *e = getenv("RT_WF_BLOCKS_PER_CU");
        per_cu = e ? atoi(e) : PS_WAVES_EU;
        if (per_cu < 1 || per_cu > 8) per_cu = PS_WAVES_EU;
    }
    return per_cu;
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
    else if (SHADE_REG9 && fp.max_depth + 1 <= 9) hipLaunchKernelGGL(k_shade_all<9>, dim3(sgrid), dim3(BLOCK), 0, stream, s);
    else hipLaunchKernelGGL(k_shade_all<0>, dim3(sgrid), dim3(BLOCK), 0, stream, s);
    RT_HIP_CHECK(hipGetLastError(), err);
    return RT_OK;
}

int persistent_max_lights(void) { return 32; }
int persistent_grid(int num_cus) { return num_cus * blocks_per_cu(); }
// entries of one wave's private ray stack: one hit's worth of rays per lane (doubled as a margin;
// pushes past the end are dropped rather than written) plus the parking list
size_t persistent_stack_entries(int n_lights) { return (size_t)2 * 64 * (size_t)(n_lights + 1) + PS_PARK; }
size_t persistent_overflow_ints(int num_cus) { return (size_t)PS_OVF * (size_t)persistent_grid(num_cus) * BLOCK; }
size_t persistent_cursor_words(void) { return (size_t)PS_SHARDS * PS_CURSOR_STRIDE; }

int persistent_frame(const DevScene &scene, const DevCamera &cam, const FrameParams &fp, FrameWorkspace &ws, void *d_rgb8,
                     void *d_hit_id, void *d_rgb32, void *d_hit_t, hipStream_t stream, int num_cus,
                     const std::function<int()> &stamp, int *n_launches, std::string *err)
{
    const int n_rows = fp.rows.n_rows;
    const int n_pixels = fp.W * n_rows;
    const int tiles = ((fp.W + 7) / 8) * ((n_rows + 7) / 8);
    if (fp.stack_depth + 1 > PS_LDS_STACK + PS_OVF) {
        if (err) *err = "LBVH deeper than the traversal stack";
        return RT_E_LIMIT;
    }
    if (scene.n_lights > persistent_max_lights()) {
        if (err) *err = "the single-launch pipeline handles at most 32 lights";
        return RT_E_LIMIT;
    }
    RT_HIP_CHECK(hipMemsetAsync(ws.trace_cursors, 0, sizeof(unsigned int) * persistent_cursor_words(), stream), err);

    StreamArgs t;
    t.sc = scene; t.cam = cam; t.W = fp.W; t.n_rows = n_rows; t.n_pixels = n_pixels; t.max_depth = fp.max_depth;
    t.rows = fp.rows;
    t.n_slots = (unsigned)tiles * 64u;
    t.hits = ws.p_hits; t.occ = ws.p_occ;
    t.ls_o = ws.ls_o; t.ls_d = ws.ls_d; t.ls_cap = (unsigned)persistent_stack_entries(scene.n_lights);
    t.cursors = ws.trace_cursors;
    t.overflow = ws.overflow;
    t.counters = ws.counters;
    int grid = persistent_grid(num_cus);
    const int need = (tiles + 3) / 4;          // one 64-slot unit per wave is the least that keeps a wave busy
    if (need < grid) grid = need;
    t.total_lanes = (unsigned)persistent_grid(num_cus) * BLOCK;

    size_t te = 0;
    auto tstamp = [&]() -> int {
        if (!fp.timing) return RT_OK;
        if (te >= ws.trav_events.size()) {
            hipEvent_t e;
            RT_HIP_CHECK(hipEventCreate(&e), err);
            ws.trav_events.push_back(e);
        }
        RT_HIP_CHECK(hipEventRecord(ws.trav_events[te++], stream), err);
        return RT_OK;
    };
    ws.trav_pairs = 0;
    if (int rc = tstamp()) return rc;
    if (fp.count_tests) hipLaunchKernelGGL(k_stream<true>, dim3(grid), dim3(BLOCK), 0, stream, t);
    else hipLaunchKernelGGL(k_stream<false>, dim3(grid), dim3(BLOCK), 0, stream, t);
    if (int rc = tstamp()) return rc;
    ws.trav_pairs = (int)(te / 2);
    if (int rc = stamp()) return rc;

    if (int rc = shade_all_launch(scene, cam, fp, ws.p_hits, ws.p_occ, 1, ws.records, ws.counters, d_rgb8, d_hit_id, d_rgb32,
                                  d_hit_t, stream, num_cus, err))
        return rc;
    if (int rc = stamp()) return rc;
    RT_HIP_CHECK(hipGetLastError(), err);
    *n_launches += 2;
    return RT_OK;
}
