// wavefront.hip — the LBVH trace pipeline: traversal separated from ray generation and shading.
//
// The fused kernel in trace.hip keeps a whole ray's life in one thread; on LBVH scenes
// its wave64 lanes spend most of their time masked off (rays end at different times,
// node and leaf work serialise).  Here the reference's recursion (Raytracer/main.cpp:144-217)
// is run level by level over flat ray queues:
//
//   k_trace<nearest, camera rays>                 nearest primitive per ray (main.cpp:147-156)
//   for every level k:
//     k_shadow_gen   hit point; the level-k+1 reflection ray (main.cpp:208-211) into the next
//                    queue; per light: L, shadow-ray origin, analytic occluders
//                    (main.cpp:178-188), survivors into one queue segment per light
//     k_trace<mixed> ONE launch for the shadow rays of level k (any triangle at any distance
//                    -> occlusion bit) and the nearest hits of level k+1: neither needs the other
//   k_shade_all      (shade.hip) per pixel: every level's local colour in the reference's
//                    accumulation order (main.cpp:166-205), the fold, clamp and quantise
//
// Hits and occlusion bits live in planes indexed [level][pixel], so nothing has to be shaded
// between the traversal launches.
//
// Queues are REGIONED: slot space is cut into regions of 256 slots; the block that
// processes input region r writes its surviving rays densely at the start of output
// region r and stores the count.  Compaction is therefore block-local (ballot +
// popcount + a 4-entry LDS prefix) and needs no global atomic; a same-address atomic
// retires only ~90 per microsecond on this chip, which would otherwise cap the frame.
//
// k_trace is the hot kernel.  Waves are persistent and every LANE owns a ray: a lane
// whose ray has finished goes idle and, once enough lanes of the wave are idle, they are
// refilled from the wave's local range of rays; the wave claims a new 64-ray unit (one
// atomic on a cursor sharded 64 ways, each on its own cache line, with stealing) only when
// that range is empty.  Lanes are either at an inner node or at a leaf; each iteration the
// wave runs the one step that more of its lanes are waiting for (ballot + popcount vote).
// Inner nodes are 4-wide (BvhNode4): one dependent 64-byte fetch per two binary levels.
// The per-lane traversal stack lives in LDS (16 entries, column-per-lane: conflict-free)
// and overflows to a global column for the rare deeper path.
#include "rt_traverse.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

namespace {

#ifndef WF_TIMES
#define WF_TIMES 0         // 1: a diagnostic build whose plain kernels also record per-wave clocks (see RT_WF_TIMES below)
#endif
#ifndef WF_STACK
#define WF_STACK 16
#endif
constexpr int WF_LDS_STACK = WF_STACK;    // stack entries per lane held in LDS
constexpr int WF_OVF = 56;                // further entries per lane in global memory
constexpr unsigned SENT_U = 0x7fffffffu;  // NODE_SENTINEL as unsigned: "(unsigned)node < SENT_U" == inner node
#ifndef WF_REFILL_MIN
#define WF_REFILL_MIN 16
#endif
#ifndef WF_UNIT
#define WF_UNIT 64
#endif
#ifndef WF_CLOSEST_FIRST
#define WF_CLOSEST_FIRST 1
#endif
#ifndef WF_MERGE
#define WF_MERGE 1         // trace level k's shadow rays and level k+1's nearest-hit rays in one launch
#endif
#ifndef WF_SETASIDE
#define WF_SETASIDE 0      // 1: a shadow lane that reaches a leaf during node steps keeps it in a register and walks on (see the node step)
#endif
#ifndef WF_SETASIDE_VOTE
#define WF_SETASIDE_VOTE 0 // 1: lanes that hold a set-aside leaf also vote for the leaf step
#endif
#ifndef WF_WAVES_EU
#define WF_WAVES_EU 8      // register budget and resident blocks per CU of k_trace
#endif
#ifndef WF_WAVES_EU_MIXED
#define WF_WAVES_EU_MIXED 7   // the mixed-kind kernel carries more state: 7 waves avoid spills in its loop
#endif
constexpr int REFILL_MIN = WF_REFILL_MIN; // refill once this many lanes of a wave are idle
constexpr int REGION = BLOCK;             // slots per queue region = one producer block iteration
constexpr int UNIT = WF_UNIT;             // rays per claimed work unit
constexpr int UPR = REGION / UNIT;        // units per region
constexpr int SHARDS = 64;                // claim cursors per launch
constexpr int CURSOR_STRIDE = 32;         // words between cursors: each on its own 128-byte line, because atomics
                                          // to one line serialise (~90 per microsecond) wherever in the line they land
constexpr int MAX_PROBES = 8;             // shards a wave tries (its own first) before it stops claiming

struct TraceArgs {
    DevScene sc;
    DevCamera cam;
    int W, n_rows;
    rt_rowset rows;
    const float4 *q_o, *q_d;        // closest, level > 0: ray queue; any-hit and mixed: q_o = shadow queue (segment-major: hit point, pixel)
    const float4 *q2_o, *q2_d;      // mixed: the next level's ray queue, traced for the nearest hit
    float2 *hits;                   // nearest hit: (t, id bits) per PIXEL, the plane of the level being traced
    unsigned int *occ;              // any-hit: occ[pixel] |= 1 << bit, the plane of (level, light chunk)
    const unsigned int *counts;     // [segment][region] rays per region (null at level 0: regions are full)
    const unsigned int *counts2;    // mixed: [region] rays per region of q2
    unsigned int *cursors;          // [SHARDS * CURSOR_STRIDE] unit claim cursors of this launch (zeroed per frame)
    unsigned int n_slots;           // level 0: slot count known on the host
    int n_segments, n_regions;
    unsigned int seg_stride;        // queue entries between segments
    int light0;                     // any-hit and mixed: the light of segment 0 (index into sc.lights)
    int *overflow;                  // [WF_OVF][total_lanes]
    unsigned int total_lanes;
    FrameCounters *counters;
    // array sizes, read by guarded builds only
    unsigned long long q_cap, q2_cap;   // entries of q_o/q_d and q2_o/q2_d
    unsigned int n_pix;                 // entries of the hit and occlusion planes
    unsigned int n_wnodes;              // entries of sc.wnodes
    unsigned long long *wave_times; // diagnostic (RT_WF_TIMES): [wave][3] 100 MHz clock at start, when nothing was left to claim, at exit
    const unsigned int *order;      // [n_units] the unit the k-th claim takes (longest rays first), or null: unit k
    unsigned int *cost;             // [n_units] COUNT builds: steps of the longest ray of every unit (atomic max), or null
};

// MODE 0: nearest hit of every ray of one queue (or of the camera rays, PRIMARY).
// MODE 1: any-hit of the shadow rays of n_segments light segments.
// MODE 2: both in one launch — the shadow rays of level k and the nearest-hit rays of level k+1 do
//         not depend on each other, and one launch has one drain instead of two.  A lane's kind is
//         the sign of `bit` (>= 0: shadow ray of that light).
template <int MODE, bool PRIMARY, int COUNTING>
__global__ __launch_bounds__(BLOCK)
__attribute__((amdgpu_waves_per_eu(MODE == 2 ? WF_WAVES_EU_MIXED : WF_WAVES_EU, MODE == 2 ? WF_WAVES_EU_MIXED : WF_WAVES_EU)))
void k_trace(TraceArgs A)
{
    // COUNTING 1: every statistic (tests, lane occupancy, coherence tables) and the units' costs; 2: the units' costs only
    // (the pass that makes a claim order: as fast as a plain launch); 0: neither.
    constexpr bool COUNT = COUNTING == 1, COST = COUNTING != 0;
    constexpr bool ANYHIT = MODE == 1;
    __shared__ int s_stack[WF_LDS_STACK * BLOCK];
    const int lane = threadIdx.x & 63;
    int *stack = s_stack + threadIdx.x;                                   // element k at stack[k * BLOCK]
    int *ovf = A.overflow + (size_t)blockIdx.x * BLOCK + threadIdx.x;     // element k at ovf[k * total_lanes]
    const DevScene &sc = A.sc;
    const unsigned long long lt = lanemask_lt();

    // ---- per-lane ray state
    int node = NODE_SENTINEL, sp = 0, tri_i = 0, tri_left = 0, best_id = -1, bit = MODE == 0 ? -1 : 0;
    unsigned int slot = 0;             // the ray's pixel: index into the hit / occlusion plane
    F3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f), inv = mk(0.f, 0.f, 1.f), oi = mk(0.f, 0.f, 0.f);
    uint3 sel_n = make_uint3(SEL_LO, SEL_LO, SEL_LO), sel_f = make_uint3(SEL_HI, SEL_HI, SEL_HI);
    float best_t = FLT_MAX;
    Tally tl = { 0u, 0u, 0u };
    unsigned int dbg[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };   // COUNT: node iters/lanes, leaf iters/lanes, refills/lanes, idle / leaf-waiting lanes at node steps
    unsigned int box_slots = 0, nsteps = 0;                     // COUNT: box tests on occupied child slots; node steps of this lane's ray
    unsigned int cur_unit = 0, my_unit = 0, ray_steps = 0;      // COST: the unit being dealt out; this lane's ray's unit and steps so far
    __shared__ unsigned long long s_coh[COUNT ? 3 * 16 * 4 : 1];
    if (COUNT) {
        for (int k = threadIdx.x; k < 3 * 16 * 4; k += BLOCK) s_coh[k] = 0ull;
        __syncthreads();
    }
#if WF_SETASIDE
    int stash = 0;                     // a shadow lane's set-aside leaf reference (negative), 0 = none
#endif
#if RT_GUARD
    unsigned int guard_iters = 0;      // guarded builds bound the loop: a slip in the lane state machine ends in an error, not a hang
#endif

    // ---- per-wave fetch state (uniform): local range [cur, end) of queue positions in segment `seg`
    unsigned int cur = 0, end = 0;
    int seg = 0;
    // Every shard needs an owner that drains it to the end: with fewer blocks than SHARDS only
    // gridDim.x shards are used.
    const int n_shards = (int)min((unsigned)SHARDS, gridDim.x);
    const int max_probes = min(MAX_PROBES, n_shards);
    int shard = (int)(blockIdx.x % (unsigned)n_shards), probes = 0;
    const unsigned int n_units = (unsigned)(A.n_segments + (MODE == 2 ? 1 : 0)) * (unsigned)A.n_regions * UPR;
    bool exhausted = n_units == 0;
    unsigned long long t_start = 0, t_exhaust = 0;
    if ((COUNT || WF_TIMES) && A.wave_times) t_start = t_exhaust = wall_clock64();

    auto push = [&](int v) {
        if (sp < WF_LDS_STACK) stack[sp * BLOCK] = v;
        else ovf[(size_t)RTG(sp - WF_LDS_STACK, WF_OVF) * A.total_lanes] = v;
        sp++;
    };
    auto pop = [&]() {
        sp--;
        if (sp < WF_LDS_STACK) {
            node = stack[sp * BLOCK];
        } else {
            // The rare deep entry comes back through an explicit global load: left as two plain C++
            // loads, hipcc merges the LDS and the global branch into ONE flat_load on a generic
            // pointer, and every pop then pays the flat path's latency and waits.
            const int *p = ovf + (size_t)RTG(sp - WF_LDS_STACK, WF_OVF) * A.total_lanes;
            int v;
            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            node = v;
        }
        if (node == NODE_SENTINEL && (MODE == 0 || (MODE == 2 && bit < 0))) A.hits[RTG(slot, A.n_pix)] = make_float2(best_t, __int_as_float(best_id));
    };

    for (;;) {
#if RT_GUARD
        // Far beyond what a frame asks of one wave (the 1080p frame averages about a thousand iterations per wave).
        if (++guard_iters > (1u << 22)) {
            if (lane == 0) atomicAdd(&A.counters->guard_hits, 1u);
            break;
        }
#endif
        if (COST && A.cost && node == NODE_SENTINEL && ray_steps != 0u) {
            // a ray has ended since the last look: its unit remembers its longest ray
            atomicMax(&A.cost[RTG(my_unit, n_units)], ray_steps);
            ray_steps = 0u;
        }
        // ---- refill idle lanes from the local range; claim a new unit when it is empty
        const unsigned long long idle = __ballot(node == NODE_SENTINEL);
        if (!exhausted && idle != 0ull && (__popcll(idle) >= REFILL_MIN || idle == ~0ull)) {
            while (cur >= end && !exhausted) {
                // units are dealt to shards round-robin (unit = k*n_shards + shard); a wave drains its own
                // shard to the end, then steals from the next few; an empty unit is followed by the next claim at once
                unsigned int k = 0;
                if (lane == 0) k = atomicAdd(&A.cursors[shard * CURSOR_STRIDE], 1u);
                k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
                const unsigned long long unit = (unsigned long long)k * (unsigned)n_shards + (unsigned)shard;
                if (unit >= n_units) {
                    shard = shard + 1 == n_shards ? 0 : shard + 1;
                    if (++probes >= max_probes) {
                        exhausted = true;
                        if ((COUNT || WF_TIMES) && A.wave_times) t_exhaust = wall_clock64();
                    }
                    continue;
                }
                unsigned int u = (unsigned)unit;
                if (A.order) u = (unsigned)RTG(A.order[u], n_units);     // the claim order of the launch: units with long rays first
                if (COST) cur_unit = u;
                const unsigned int per_seg = (unsigned)A.n_regions * UPR;
                seg = (int)(u / per_seg);
#if WF_CLOSEST_FIRST
                // mixed launch: the nearest-hit rays are claimed first, the shadow rays fill the end of the launch
                if (MODE == 2) seg = seg == 0 ? A.n_segments : seg - 1;
#endif
                const unsigned int rem = u % per_seg, region = rem / UPR, sub = rem % UPR;
                unsigned int cnt;
                if (PRIMARY) {
                    const unsigned int r0 = region * REGION;
                    cnt = A.n_slots > r0 ? min((unsigned)REGION, A.n_slots - r0) : 0u;
                } else if (MODE == 2 && seg == A.n_segments) {
                    cnt = A.counts2[region];
                } else {
                    cnt = A.counts[(size_t)seg * A.n_regions + region];
                }
                cur = region * REGION + sub * UNIT;
                end = region * REGION + min(sub * UNIT + UNIT, cnt);
            }
            const unsigned int my = cur + (unsigned)__popcll(idle & lt);
            if (COUNT && cur < end) { dbg[4]++; dbg[5] += min((unsigned)__popcll(idle), end - cur); }
            if (node == NODE_SENTINEL && my < end) {
                bool live = true;
                const bool unit_any = MODE == 1 || (MODE == 2 && seg < A.n_segments);     // wave-uniform
                if (unit_any) {
                    const size_t q = (size_t)RTG((size_t)seg * A.seg_stride + my, A.q_cap);
                    const float4 qo = A.q_o[q];
                    slot = (unsigned)__float_as_int(qo.w);
                    // A queued shadow ray is its hit point and pixel (16 bytes, half of what the ray itself would take through
                    // the queue); the ray is formed here exactly as k_shadow_gen and k_shade_all form it (main.cpp:178-179).
                    const DevAnalytic lt_ = sc.analytic[RTG(sc.lights[RTG(A.light0 + seg, sc.n_lights)], sc.n_analytic)];
                    const F3 pos = f3(qo);
                    const F3 L = normalize(sub(mk(lt_.px, lt_.py, lt_.pz), pos));
                    o = add(pos, muls(L, 0.01f)); d = L;
                    bit = seg;
                } else if (PRIMARY) {
                    const PixelRay pr = primary_ray(A.cam, A.W, A.n_rows, A.rows, my);
                    o = pr.o; d = pr.d; live = pr.live;
                    slot = (unsigned)pr.pixel;
                } else {
                    const size_t q = (size_t)RTG(my, MODE == 2 ? A.q2_cap : A.q_cap);
                    const float4 qo = (MODE == 2 ? A.q2_o : A.q_o)[q], qd = (MODE == 2 ? A.q2_d : A.q_d)[q];
                    o = f3(qo); d = f3(qd);
                    slot = (unsigned)__float_as_int(qo.w);
                    if (MODE == 2) bit = -1;
                }
                if (live) {
                    best_t = FLT_MAX; best_id = -1;
                    if (PRIMARY) {
                        analytic_nearest<COUNT>(sc, o, d, best_t, best_id, tl);
                    } else if (!unit_any) {
                        // a queued ray's nearest sphere / plane was found where the ray was made (k_shadow_gen, all 64 lanes
                        // at work there) and waits in the hit plane this lane writes its result to
                        const float2 h0 = A.hits[RTG(slot, A.n_pix)];
                        best_t = h0.x; best_id = __float_as_int(h0.y);
                        if (COUNT) tl.analytic += (unsigned)sc.n_analytic;
                    }
                    slab_ray(sc.grid, o, d, inv, oi, sel_n, sel_f);
                    stack[0] = NODE_SENTINEL;
                    sp = 1;
                    tri_left = 0;
                    node = 0;
#if WF_SETASIDE
                    stash = 0;
#endif
                    if (COUNT) nsteps = 0;
                    if (COST) { my_unit = cur_unit; ray_steps = 1u; }
                }
            }
            cur = min(end, cur + (unsigned)__popcll(idle));
        }

        const unsigned long long m_node = __ballot((unsigned)node < SENT_U);
        const unsigned long long m_leaf = __ballot(node < 0);
        if ((m_node | m_leaf) == 0ull) {
            if (exhausted) break;
            continue;
        }

        // the step more lanes are waiting for
#if WF_SETASIDE && WF_SETASIDE_VOTE
        const bool do_node = (int)__popcll(m_node) >= (int)__popcll(__ballot(node < 0 || stash != 0));
#else
        const bool do_node = (int)__popcll(m_node) >= (int)__popcll(m_leaf);
#endif
        const bool do_leaf = !do_node;
        if (COUNT && do_node) { dbg[0]++; dbg[1] += (unsigned)__popcll(m_node); dbg[6] += 64u - (unsigned)__popcll(m_node | m_leaf); dbg[7] += (unsigned)__popcll(m_leaf); }
        if (COUNT && do_leaf) { dbg[2]++; dbg[3] += (unsigned)__popcll(m_leaf); }
        if (COUNT && do_node) {
            // How alike are the lanes of this node step?  Per ray kind: the lanes at the same node as each lane, the
            // lanes at the first lane's node, the lanes in the first lane's direction octant.
            const bool at = (unsigned)node < SENT_U;
            const int kind = PRIMARY ? 0 : ((MODE == 1 || (MODE == 2 && bit >= 0)) ? 1 : 2);
            const int oct = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
            for (int kd = 0; kd < 3; kd++) {
                const bool mine = at && kind == kd;
                unsigned long long rest = __ballot(mine);
                if (rest == 0ull) continue;
                const int first = __ffsll(rest) - 1;
                const int first_node = __shfl(node, first), first_oct = __shfl(oct, first);
                unsigned int group = 0;
                while (rest != 0ull) {
                    const int nd = __shfl(node, __ffsll(rest) - 1);
                    const unsigned long long g = __ballot(mine && node == nd);
                    if (mine && node == nd) group = (unsigned)__popcll(g);
                    rest &= ~g;
                }
                if (mine) {
                    unsigned long long *c = s_coh + (kd * 16 + (int)min(nsteps, 15u)) * 4;
                    atomicAdd(c, 1ull);
                    atomicAdd(c + 1, (unsigned long long)group);
                    if (node == first_node) atomicAdd(c + 2, 1ull);
                    if (oct == first_oct) atomicAdd(c + 3, 1ull);
                }
            }
            if (at) nsteps++;
        }
        if (COST && ((do_node && (unsigned)node < SENT_U) || (do_leaf && node < 0))) ray_steps++;
        if (do_node) {
            // ---- inner-node step: four children per 64-byte node
            if ((unsigned)node < SENT_U) {
                const float tmax = ANYHIT ? FLT_MAX : best_t;      // mixed: a shadow lane's best_t stays FLT_MAX
                // four children from one 64-byte node; the lattice boxes contain the padded float boxes
                uint4 qa, qb, qc, qd;
                {
                    // 32-bit byte offset from the uniform base: one shift instead of 64-bit address arithmetic
                    // (the builder keeps the wide node array below 4 GiB)
                    const uint4 *np = (const uint4 *)((const char *)sc.wnodes + (unsigned)RTG(node, A.n_wnodes) * 64u);
                    qa = np[0]; qb = np[1]; qc = np[2]; qd = np[3];
                }
                // Entry distance and hit flag of each child in ONE integer key: the bits of the (non-negative) entry
                // distance, with the sign bit set when the box is hit, so that as signed integers hit children sort
                // before missed ones and among themselves by distance.  The flag comes from the sign of far - near
                // (a subtraction and one three-input bit operation, both full-rate instructions) instead of a
                // compare and a select.  Unused slots hold an inverted box and always miss.
                int dk[4];
                int rk[4] = { (int)qd.x, (int)qd.y, (int)qd.z, (int)qd.w };
                {
                    const unsigned int wx[4] = { qa.x, qa.w, qb.z, qc.y }, wy[4] = { qa.y, qb.x, qb.w, qc.z },
                                       wz[4] = { qa.z, qb.y, qc.x, qc.w };
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const V2 x = plane_pair(wx[k], sel_n.x, sel_f.x, inv.x, oi.x), y = plane_pair(wy[k], sel_n.y, sel_f.y, inv.y, oi.y),
                                 z = plane_pair(wz[k], sel_n.z, sel_f.z, inv.z, oi.z);
                        const float nk = fmaxf(__builtin_fmaxf(__builtin_fmaxf(x.x, y.x), z.x), 0.0f);
                        const float fk = fminf(__builtin_fminf(__builtin_fminf(x.y, y.y), z.y), tmax);
                        // bit 31 from ~(fk - nk), bits 30..0 from nk: one v_bitop3_b32 (truth table 0x72 = c ? ~b : a).  nk's own
                        // sign bit is dropped: max(.., 0.0f) may hand back -0.0f, which would read as a hit at the far end.
                        dk[k] = __builtin_amdgcn_bitop3_b32(__float_as_int(nk), __float_as_int(fk - nk), (int)0x80000000, 0x72);
                    }
                }
                if (COUNT) {
                    tl.box += 4;
#pragma unroll
                    for (int k = 0; k < 4; k++) box_slots += rk[k] != NODE_SENTINEL ? 1u : 0u;
                }
#define WF_CSWAP(a, b)                                                          \
                {                                                               \
                    const bool sw = dk[b] < dk[a];                              \
                    const int td = sw ? dk[b] : dk[a];                          \
                    const int tr = sw ? rk[b] : rk[a];                          \
                    dk[b] = sw ? dk[a] : dk[b];                                 \
                    rk[b] = sw ? rk[a] : rk[b];                                 \
                    dk[a] = td;                                                 \
                    rk[a] = tr;                                                 \
                }
                // nearest child to slot 0; the other hits are pushed in slot order
                WF_CSWAP(0, 1) WF_CSWAP(2, 3) WF_CSWAP(0, 2)
#undef WF_CSWAP
                if (dk[0] >= 0) {
                    pop();
                } else if (sp + 3 <= WF_LDS_STACK) {
                    // branch-free push: every slot is stored, a miss lands on the position the next
                    // hit (or a later push) overwrites
                    const int c3 = (int)((unsigned)dk[3] >> 31), c2 = (int)((unsigned)dk[2] >> 31), c1 = (int)((unsigned)dk[1] >> 31);
                    stack[sp * BLOCK] = rk[3];
                    stack[(sp + c3) * BLOCK] = rk[2];
                    stack[(sp + c3 + c2) * BLOCK] = rk[1];
                    sp += c3 + c2 + c1;
                    node = rk[0];
                } else {
                    if (dk[3] < 0) push(rk[3]);
                    if (dk[2] < 0) push(rk[2]);
                    if (dk[1] < 0) push(rk[1]);
                    node = rk[0];
                }
#if WF_SETASIDE
                // Any-hit rays need no order (main.cpp:181-188 stops at the first primitive that accepts the ray,
                // whichever it is).  A shadow lane that arrives at a leaf while the wave goes on with node steps
                // therefore keeps the leaf in a register and takes the next node off its stack; the leaf is tested
                // in the wave's next leaf step.  Only while the stack holds more than the end-of-ray mark; and when
                // the mark comes off the stack with a leaf still set aside, that leaf becomes the lane's node and
                // the mark goes back (stack[0] still holds it), so the ray ends only after the leaf was tested.
                if (MODE == 1 || (MODE == 2 && bit >= 0)) {
                    if (node == NODE_SENTINEL) {
                        if (stash != 0) { node = stash; stash = 0; sp = 1; }
                    } else if (node < 0 && stash == 0 && sp > 1) {
                        stash = node;
                        pop();
                    }
                }
#endif
            }
        }
#if WF_SETASIDE
        if (do_leaf) {
            // ---- leaf step with set-aside leaves: a lane tests the leaf it stands at, a shadow lane then the leaf it
            // set aside (whether or not it stands at one); after the step no lane holds a set-aside leaf.
            const bool shadow_lane = MODE == 1 || (MODE == 2 && bit >= 0);
            const int at_leaf = node < 0 ? node : 0;
            int ref = at_leaf != 0 ? at_leaf : stash, ref2 = at_leaf != 0 ? stash : 0;
            stash = 0;
            while (ref != 0) {
                const int leaf = ~ref;
                tri_i = leaf >> 3;
                tri_left = (leaf & 7) + 1;
                for (;;) {
                    const float4 *tp = (const float4 *)((const char *)sc.bvh_tri + (unsigned)RTG(tri_i, sc.n_refs) * 48u);
                    const float4 v1 = tp[0];
                    const F3 e1 = f3(tp[1]), e2 = f3(tp[2]);
                    float t;
                    const bool h = tri_raycast(f3(v1), e1, e2, o, d, t);
                    if (COUNT) tl.tri += 1;
                    tri_i++;
                    tri_left--;
                    if (shadow_lane) {
                        if (h) {                                       // main.cpp:183-186: any hit blocks the light
                            atomicOr(&A.occ[RTG(slot, A.n_pix)], 1u << (RTG(bit, 32) & 31));
                            node = NODE_SENTINEL;
                            tri_left = 0;
                            ref2 = 0;
                        }
                    } else if (h) {
                        const int id = sc.n_analytic + __float_as_int(v1.w);
                        if (best_t > t || (best_t == t && id < best_id)) { best_t = t; best_id = id; }
                    }
                    if (tri_left == 0) break;
                }
                ref = ref2;
                ref2 = 0;
            }
            if (at_leaf != 0 && node != NODE_SENTINEL) pop();
        }
#else
        if (do_leaf) {
            // ---- leaf step: every triangle of the leaf through the reference's Moller-Trumbore test
            if (node < 0) {
                if (tri_left == 0) {
                    const int leaf = ~node;
                    tri_i = leaf >> 3;
                    tri_left = (leaf & 7) + 1;
                }
                for (;;) {
                // 32-bit byte offset (48 bytes per triangle; the builder's triangle limit keeps it below 4 GiB)
                const float4 *tp = (const float4 *)((const char *)sc.bvh_tri + (unsigned)RTG(tri_i, sc.n_refs) * 48u);
                const float4 v1 = tp[0];
                const F3 e1 = f3(tp[1]), e2 = f3(tp[2]);
                float t;
                const bool h = tri_raycast(f3(v1), e1, e2, o, d, t);
                if (COUNT) tl.tri += 1;
                tri_i++;
                tri_left--;
                if (MODE == 1 || (MODE == 2 && bit >= 0)) {
                    if (h) {                                       // main.cpp:183-186: any hit blocks the light
                        atomicOr(&A.occ[RTG(slot, A.n_pix)], 1u << (RTG(bit, 32) & 31));
                        node = NODE_SENTINEL;
                        tri_left = 0;
                    }
                } else if (h) {
                    const int id = sc.n_analytic + __float_as_int(v1.w);
                    if (best_t > t || (best_t == t && id < best_id)) { best_t = t; best_id = id; }
                }
                if (node == NODE_SENTINEL || tri_left == 0) break;
                }
                if (node != NODE_SENTINEL && tri_left == 0) pop();
            }
        }
#endif
    }

    if ((COUNT || WF_TIMES) && A.wave_times && lane == 0) {
        unsigned long long *w = A.wave_times + 3 * ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
        w[0] = t_start; w[1] = t_exhaust; w[2] = wall_clock64();
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
            for (int k = 0; k < 8; k++)
                if (dbg[k]) atomicAdd(dst + k, (unsigned long long)dbg[k]);
        }
        {
            unsigned long long v = box_slots;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && v) atomicAdd(&A.counters->box_slots, v);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 3 * 16 * 4; k += BLOCK)
            if (s_coh[k]) atomicAdd(&A.counters->coh[0][0][0] + k, s_coh[k]);
    }
}

// ------------------------------------------------------------------ packet traversal of the camera rays
// The 64 camera rays of an 8x8 pixel tile (main.cpp:262-272) start at one point and leave through neighbouring pixels:
// during the node steps of k_trace most lanes of such a wave stand at the same node as their neighbours.  Here the
// wave walks the hierarchy ONCE for its 64 rays: one node at a time, wave-uniform; the node comes through the scalar
// cache as floats (BvhNode4F) into scalar registers, every lane tests its own ray against the four boxes with scalar
// operands (nothing to decode, no per-lane stack, no per-lane sort), a child is entered when ANY lane's ray enters it,
// in the order of the first such lane's entry distances, and the stack is one vector register (lane k holds entry k).
// A leaf's triangles come through the scalar cache too and every lane runs the reference's own test on them
// (main.cpp:109-129).  The hierarchy only prunes and ties go to the lowest id explicitly, so the hits are those of
// k_trace bit for bit; the box arithmetic is the same fused multiply-add on the same values.
constexpr int PK_STACK = 64;
typedef float PkF16 __attribute__((ext_vector_type(16)));
typedef float PkF4 __attribute__((ext_vector_type(4)));
#ifndef PK_VECTOR_LOADS
#define PK_VECTOR_LOADS 0
#endif
#if PK_VECTOR_LOADS
// the same wave-uniform addresses through the vector memory path: every lane receives the same 16 bytes
typedef const PkF4 *ConstF4P;
#else
// loads through the constant address space with a wave-uniform address: scalar loads (s_load_dwordx16 / x4)
typedef const __attribute__((address_space(4))) PkF4 *ConstF4P;
#endif

#ifndef PK_WAVES
#define PK_WAVES 8
#endif
template <int COUNTING>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(PK_WAVES, PK_WAVES))) void k_trace_packet(TraceArgs A)
{
    constexpr bool COUNT = COUNTING == 1, COST = COUNTING != 0;      // as in k_trace
    static_assert(UNIT <= 64, "a packet is one wave");
    const int lane = threadIdx.x & 63;
    const DevScene &sc = A.sc;
    Tally tl = { 0u, 0u, 0u };
    unsigned int box_slots = 0, n_node_steps = 0, n_node_lanes = 0, n_leaf_steps = 0, n_leaf_lanes = 0;
    const int n_shards = (int)min((unsigned)SHARDS, gridDim.x);
    const int max_probes = min(MAX_PROBES, n_shards);
    int shard = (int)(blockIdx.x % (unsigned)n_shards), probes = 0;
    const unsigned int n_units = (unsigned)A.n_regions * UPR;
#if RT_GUARD
    unsigned int guard_iters = 0;
#endif
    while (n_units != 0) {
        // ---- claim one unit of 64 slots = one 8x8 pixel tile (same cursors and stealing as k_trace)
        unsigned int k = 0;
        if (lane == 0) k = atomicAdd(&A.cursors[shard * CURSOR_STRIDE], 1u);
        k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        const unsigned long long unit = (unsigned long long)k * (unsigned)n_shards + (unsigned)shard;
        if (unit >= n_units) {
            shard = shard + 1 == n_shards ? 0 : shard + 1;
            if (++probes >= max_probes) break;
            continue;
        }
        unsigned int u = (unsigned)unit;
        if (A.order) u = (unsigned)RTG(A.order[u], n_units);
        const unsigned int region = u / UPR, sub = u % UPR, r0 = region * REGION;
        unsigned int packet_steps = 0;
        const unsigned int cnt = A.n_slots > r0 ? min((unsigned)REGION, A.n_slots - r0) : 0u;
        const unsigned int cur = r0 + sub * UNIT, end = r0 + min(sub * UNIT + UNIT, cnt);
        if (cur >= end) continue;
        const unsigned int my = cur + (unsigned)lane;
        bool live = my < end;
        const PixelRay pr = primary_ray(A.cam, A.W, A.n_rows, A.rows, live ? my : cur);
        live = live && pr.live;
        const F3 o = pr.o, d = pr.d;
        float best_t = FLT_MAX;
        int best_id = -1;
        if (live) analytic_nearest<COUNT>(sc, o, d, best_t, best_id, tl);
        F3 inv, oi;
        uint3 sel_n, sel_f;
        slab_ray(sc.grid, o, d, inv, oi, sel_n, sel_f);
        const unsigned long long alive = __ballot(live);
        if (alive == 0ull) continue;
        // When every ray of the tile points into the same octant, near and far planes are picked once per wave.
        const unsigned long long mx = __ballot(live && inv.x < 0.0f), my_ = __ballot(live && inv.y < 0.0f), mz = __ballot(live && inv.z < 0.0f);
        const bool uni = (mx == 0ull || mx == alive) && (my_ == 0ull || my_ == alive) && (mz == 0ull || mz == alive);
        const int ix = mx != 0ull ? 1 : 0, iy = my_ != 0ull ? 1 : 0, iz = mz != 0ull ? 1 : 0;    // which plane of a pair comes first

        int stackv = 0;          // the packet's stack: lane k holds entry k
        int sp = 0, node = 0;    // wave-uniform
        for (;;) {
#if RT_GUARD
            if (++guard_iters > (1u << 22)) {
                if (lane == 0) atomicAdd(&A.counters->guard_hits, 1u);
                break;
            }
#endif
            if (node >= 0) {
                // BvhNode4F: lo_x hi_x lo_y hi_y lo_z hi_z ref (pad is not read), 16 bytes each, at a wave-uniform address
                ConstF4P np = (ConstF4P)(unsigned long long)(sc.fnodes + (unsigned)RTG(node, A.n_wnodes));
                const PkF4 rf = np[6];
                unsigned int key[4];
                const int ref[4] = { __float_as_int(rf.x), __float_as_int(rf.y), __float_as_int(rf.z), __float_as_int(rf.w) };
                float tn[4], tf[4];
                if (uni) {
                    // the planes every ray meets first / last: picked once for the wave, by address
                    const PkF4 ax = np[ix], bx = np[1 - ix], ay = np[2 + iy], by = np[3 - iy], az = np[4 + iz], bz = np[5 - iz];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        tn[c] = fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(ax[c], inv.x, oi.x), __builtin_fmaf(ay[c], inv.y, oi.y)),
                                                      __builtin_fmaf(az[c], inv.z, oi.z)), 0.0f);
                        tf[c] = fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaf(bx[c], inv.x, oi.x), __builtin_fmaf(by[c], inv.y, oi.y)),
                                                      __builtin_fmaf(bz[c], inv.z, oi.z)), best_t);
                    }
                } else {
                    const PkF4 lx = np[0], hx = np[1], ly = np[2], hy = np[3], lz = np[4], hz = np[5];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float x0 = __builtin_fmaf(lx[c], inv.x, oi.x), x1 = __builtin_fmaf(hx[c], inv.x, oi.x);
                        const float y0 = __builtin_fmaf(ly[c], inv.y, oi.y), y1 = __builtin_fmaf(hy[c], inv.y, oi.y);
                        const float z0 = __builtin_fmaf(lz[c], inv.z, oi.z), z1 = __builtin_fmaf(hz[c], inv.z, oi.z);
                        tn[c] = fmaxf(__builtin_fmaxf(__builtin_fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1)), 0.0f);
                        tf[c] = fminf(__builtin_fminf(__builtin_fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)), best_t);
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    // an unused slot is never entered (its inverted box only says so while near and far planes are
                    // picked by the direction signs; the minimum / maximum form above would turn it round)
                    const unsigned long long m = ref[c] == NODE_SENTINEL ? 0ull : (__ballot(tn[c] <= tf[c]) & alive);
                    key[c] = 0xffffffffu;
                    if (m != 0ull) key[c] = (unsigned)__builtin_amdgcn_readlane(__float_as_int(tn[c]), __ffsll((long long)m) - 1);
                    if (COUNT) box_slots += (live && ref[c] != NODE_SENTINEL) ? 1u : 0u;
                }
                if (COUNT) { tl.box += live ? 4u : 0u; n_node_steps++; n_node_lanes += (unsigned)__popcll(alive); }
                if (COST) packet_steps++;
                // children by the first entering lane's entry distance (non-negative floats order like their bits); all scalar
#define PK_CSWAP(a, b)                                                          \
                if (key[b] < key[a]) {                                          \
                    const unsigned int tk = key[a]; key[a] = key[b]; key[b] = tk; \
                    const int tr = ref_[a]; ref_[a] = ref_[b]; ref_[b] = tr;    \
                }
                int ref_[4] = { ref[0], ref[1], ref[2], ref[3] };
                PK_CSWAP(0, 1) PK_CSWAP(2, 3) PK_CSWAP(0, 2) PK_CSWAP(1, 3) PK_CSWAP(1, 2)
#undef PK_CSWAP
                if (key[0] == 0xffffffffu) {
                    if (sp == 0) break;
                    sp--;
                    node = __builtin_amdgcn_readlane(stackv, sp);
                } else {
                    // farthest first, so that the nearest of them comes off first
#pragma unroll
                    for (int c = 3; c >= 1; c--)
                        if (key[c] != 0xffffffffu) {
                            if (lane == (int)RTG(sp, PK_STACK)) stackv = ref_[c];
                            sp++;
                        }
                    node = ref_[0];
                }
            } else {
                // ---- leaf: every triangle through the reference's test, for every ray of the tile
                const int leaf = ~node;
                int ti = leaf >> 3;
                const int n_t = (leaf & 7) + 1;
                for (int j = 0; j < n_t; j++, ti++) {
                    ConstF4P tp = (ConstF4P)(unsigned long long)((const char *)sc.bvh_tri + (unsigned)RTG(ti, sc.n_refs) * 48u);
                    const PkF4 v1 = tp[0], e1 = tp[1], e2 = tp[2];
                    float t;
                    if (live && tri_raycast(mk(v1.x, v1.y, v1.z), mk(e1.x, e1.y, e1.z), mk(e2.x, e2.y, e2.z), o, d, t)) {
                        const int id = sc.n_analytic + __float_as_int(v1.w);
                        if (best_t > t || (best_t == t && id < best_id)) { best_t = t; best_id = id; }
                    }
                    if (COUNT && live) tl.tri += 1;
                }
                if (COUNT) { n_leaf_steps++; n_leaf_lanes += (unsigned)__popcll(alive); }
                if (COST) packet_steps++;
                if (sp == 0) break;
                sp--;
                node = __builtin_amdgcn_readlane(stackv, sp);
            }
        }
        if (live) A.hits[RTG((unsigned)pr.pixel, A.n_pix)] = make_float2(best_t, __int_as_float(best_id));
        if (COST && A.cost && lane == 0) A.cost[RTG(u, n_units)] = packet_steps;
    }
    if (COUNT) {
        unsigned int vals[4] = { tl.box, tl.tri, tl.analytic, box_slots };
        for (int c = 0; c < 4; c++) {
            unsigned long long v = vals[c];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && v)
                atomicAdd(c == 0 ? &A.counters->box_tests : c == 1 ? &A.counters->tri_tests : c == 2 ? &A.counters->analytic_tests : &A.counters->box_slots, v);
        }
        if (lane == 0) {
            if (n_node_steps) { atomicAdd(&A.counters->node_iters, (unsigned long long)n_node_steps); atomicAdd(&A.counters->node_lanes, (unsigned long long)n_node_lanes); }
            if (n_leaf_steps) { atomicAdd(&A.counters->leaf_iters, (unsigned long long)n_leaf_steps); atomicAdd(&A.counters->leaf_lanes, (unsigned long long)n_leaf_lanes); }
        }
    }
}

// ------------------------------------------------------------------ shadow-ray generation and shading
struct ShadeArgs {
    DevScene sc;
    DevCamera cam;
    int W, n_rows, n_pixels;
    rt_rowset rows;
    int level, max_depth;
    unsigned int n_slots;             // level 0 slot count
    int n_regions;
    const unsigned int *in_counts;    // level > 0: rays per region of this level's queue
    const float4 *q_o, *q_d;          // this level's rays
    const float2 *hits;               // [pixel] nearest hits of this level
    unsigned int *occ;                // [pixel] occlusion bits of (this level, this chunk)
    unsigned int slots_cap;
    // k_shadow_gen
    int chunk;
    float4 *sq_o;                     // [segment][slots_cap] (hit point, pixel)
    unsigned int *sq_counts;          // [segment][region]
    float4 *nq_o, *nq_d;              // next level's rays
    float2 *next_hits;                // [pixel] hit plane of the next level: seeded with each new ray's nearest sphere / plane
    unsigned int *out_counts;         // next level's rays per region
    FrameCounters *counters;
};

struct SlotRay { F3 o, d; int pixel; bool live; float t; int id; };

template <bool PRIMARY>
__device__ __forceinline__ unsigned int region_count(const ShadeArgs &A, int region)
{
    if (PRIMARY) {
        const unsigned int r0 = (unsigned)region * REGION;
        return A.n_slots > r0 ? min((unsigned)REGION, A.n_slots - r0) : 0u;
    }
    return A.in_counts[region];
}

// A region's queue entries as they lie in memory, asked for one region ahead of their use: a block walks its regions in a
// loop whose body is a chain of dependent loads (count -> ray -> hit -> surface -> material), and the first two links
// of the next region's chain cost nothing while the current region is worked on.  Entries past the region's count are
// read too (they lie inside the queue) and ignored.
struct SlotFetch { float4 qo, qd; unsigned int cnt; };

template <bool PRIMARY>
__device__ __forceinline__ SlotFetch fetch_slot(const ShadeArgs &A, int region)
{
    SlotFetch f;
    f.qo = f.qd = make_float4(0.f, 0.f, 1.f, 0.f);
    f.cnt = 0;
    if (region < A.n_regions) {
        f.cnt = region_count<PRIMARY>(A, region);
        const unsigned int slot = (unsigned)region * REGION + threadIdx.x;
        if (!PRIMARY && slot < A.slots_cap) { f.qo = A.q_o[RTG(slot, A.slots_cap)]; f.qd = A.q_d[RTG(slot, A.slots_cap)]; }
    }
    return f;
}

// The third link of the chain, asked for once the entry it depends on has landed (in the middle of the previous
// region's work): the nearest hit of the entry's ray, from the hit plane of its pixel.
template <bool PRIMARY>
__device__ __forceinline__ float2 fetch_hit(const ShadeArgs &A, int region, const SlotFetch &f)
{
    float2 h = make_float2(FLT_MAX, __int_as_float(-1));
    if (region < A.n_regions && threadIdx.x < f.cnt) {
        int pixel = __float_as_int(f.qo.w);
        bool live = true;
        if (PRIMARY) {
            const PixelRay pr = primary_ray(A.cam, A.W, A.n_rows, A.rows, (unsigned)region * REGION + threadIdx.x);
            pixel = pr.pixel; live = pr.live;
        }
        if (live) h = A.hits[RTG(pixel, A.n_pixels)];
    }
    return h;
}

template <bool PRIMARY>
__device__ __forceinline__ SlotRay load_slot(const ShadeArgs &A, unsigned int slot, const SlotFetch &f, float2 h)
{
    SlotRay r;
    r.live = threadIdx.x < f.cnt;
    r.pixel = 0; r.t = FLT_MAX; r.id = -1;
    r.o = mk(0.f, 0.f, 0.f); r.d = mk(0.f, 0.f, 1.f);
    if (r.live) {
        if (PRIMARY) {
            const PixelRay pr = primary_ray(A.cam, A.W, A.n_rows, A.rows, slot);
            r.o = pr.o; r.d = pr.d; r.pixel = pr.pixel; r.live = pr.live;
        } else {
            r.o = f3(f.qo); r.d = f3(f.qd); r.pixel = __float_as_int(f.qo.w);
        }
    }
    if (r.live) {
        r.pixel = (int)RTG(r.pixel, A.n_pixels);
        r.t = h.x; r.id = __float_as_int(h.y);
        // never index scene arrays with an id outside the scene
        if (r.id >= A.sc.n_analytic + A.sc.n_tris) r.id = -1;
    }
    return r;
}

template <bool PRIMARY>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_shadow_gen(ShadeArgs A)
{
    // [list][wave] rays each wave of the block contributes to every output list of the region: lists 0..31 = the shadow
    // queues of the chunk's lights, list 32 = the next level's ray queue.  All lists are ranked behind ONE barrier
    // (one barrier per list made a region wait seven times where it now waits twice).
    constexpr int WAVES = BLOCK / 64, LISTS = 33;
    __shared__ unsigned int s_cnt[LISTS][WAVES];
    const DevScene &sc = A.sc;
    Tally tl = { 0u, 0u, 0u };
    const int l0 = A.chunk * 32, l1 = min(sc.n_lights, l0 + 32);
    const int wave = threadIdx.x >> 6;
    const unsigned long long lt = lanemask_lt();
    SlotFetch ahead = fetch_slot<PRIMARY>(A, (int)blockIdx.x);
    float2 hit_ahead = fetch_hit<PRIMARY>(A, (int)blockIdx.x, ahead);
    for (int region = blockIdx.x; region < A.n_regions; region += gridDim.x) {
        const unsigned int slot = (unsigned)region * REGION + threadIdx.x;
        const SlotRay r = load_slot<PRIMARY>(A, slot, ahead, hit_ahead);
        ahead = fetch_slot<PRIMARY>(A, region + (int)gridDim.x);
        bool lit = r.live && r.id >= 0;
        if (lit && r.id < sc.n_analytic && sc.analytic[r.id].is_light) lit = false;       // main.cpp:166
        const F3 pos = add(r.o, muls(r.d, r.t));                                          // main.cpp:163
        // the next level's ray (main.cpp:208-211) leaves here, before the shadow rays are traced:
        // it does not depend on them, and the two kinds of ray share the next traversal launch
        bool spawn = false;
        F3 R = r.d;
        if (A.chunk == 0 && lit && A.level < A.max_depth) {
            const Surface s = surface_at(sc, r.id, pos, r.o, r.d);                        // main.cpp:164
            spawn = s.kr > 0.0f;                                                          // main.cpp:208
            if (spawn) R = mirror_dir(r.d, s.N);                                          // main.cpp:210
        }
        hit_ahead = fetch_hit<PRIMARY>(A, region + (int)gridDim.x, ahead);                // (its entry has landed by now)
        // per light: blocked by a sphere or plane already (bit set), or a shadow ray to be traced (want)
        unsigned int bits = 0, want = 0;
        for (int li = l0; li < l1; li++) {                                                // main.cpp:170-173
            if (lit) {
                const DevAnalytic lt_ = sc.analytic[sc.lights[li]];
                const F3 L = normalize(sub(mk(lt_.px, lt_.py, lt_.pz), pos));             // main.cpp:178
                const F3 so = add(pos, muls(L, 0.01f));                                   // main.cpp:179
                if (analytic_occluded<false>(sc, so, L, tl)) bits |= 1u << (li - l0);
                else want |= 1u << (li - l0);
            }
        }
        // ---- ranks: every wave's count per list, one barrier, then each ray's place = waves before + lanes before
        const unsigned long long m_spawn = __ballot(spawn);
        if ((threadIdx.x & 63) == 0) s_cnt[32][wave] = (unsigned)__popcll(m_spawn);
        for (int li = l0; li < l1; li++) {
            const unsigned long long m = __ballot((want >> (li - l0)) & 1u);
            if ((threadIdx.x & 63) == 0) s_cnt[li - l0][wave] = (unsigned)__popcll(m);
        }
        __syncthreads();
        if (A.chunk == 0) {
            unsigned int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                const unsigned int c = s_cnt[32][w];
                if (w < wave) before += c;
                total += c;
            }
            if (spawn) {
                const unsigned int rank = before + (unsigned)__popcll(m_spawn & lt);
                const size_t dst = (size_t)RTG((size_t)region * REGION + rank, A.slots_cap);
                A.nq_o[dst] = make_float4(pos.x, pos.y, pos.z, __int_as_float(r.pixel));  // origin is pos itself (main.cpp:211)
                A.nq_d[dst] = make_float4(R.x, R.y, R.z, 0.f);
                // the new ray's nearest analytic primitive (the first part of main.cpp:147-156), here with every lane busy
                // instead of in the traversal's refill with a third of them: the traversal starts from this (t, id)
                float bt = FLT_MAX;
                int bid = -1;
                analytic_nearest<false>(sc, pos, R, bt, bid, tl);
                A.next_hits[r.pixel] = make_float2(bt, __int_as_float(bid));
            }
            if (threadIdx.x == 0) A.out_counts[region] = total;
        }
        for (int li = l0; li < l1; li++) {
            const bool mine = (want >> (li - l0)) & 1u;
            const unsigned long long m = __ballot(mine);
            unsigned int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                const unsigned int c = s_cnt[li - l0][w];
                if (w < wave) before += c;
                total += c;
            }
            if (mine) {
                const unsigned int rank = before + (unsigned)__popcll(m & lt);
                const size_t q = (size_t)(li - l0) * A.slots_cap + (size_t)RTG((size_t)region * REGION + rank, A.slots_cap);
                A.sq_o[q] = make_float4(pos.x, pos.y, pos.z, __int_as_float(r.pixel));    // the traversal forms L and the origin again
            }
            if (threadIdx.x == 0) A.sq_counts[(size_t)(li - l0) * A.n_regions + region] = total;
        }
        if (r.live && l1 > l0) A.occ[r.pixel] = bits;
        __syncthreads();        // the counts are read; the next region may overwrite them
    }
}

template <int MODE, bool PRIMARY>
void launch_trace(const TraceArgs &a, int counting, int grid, hipStream_t stream)
{
    if (counting == 1) hipLaunchKernelGGL((k_trace<MODE, PRIMARY, 1>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else if (counting == 2) hipLaunchKernelGGL((k_trace<MODE, PRIMARY, 2>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((k_trace<MODE, PRIMARY, 0>), dim3(grid), dim3(BLOCK), 0, stream, a);
}

}  // namespace

// Persistent traversal blocks per CU (the LDS/VGPR residency limit).  RT_WF_BLOCKS_PER_CU
// overrides it for occupancy experiments.
int wavefront_trace_grid(int num_cus)
{
    static int per_cu = 0;
    if (per_cu == 0) {
        const char *e = getenv("RT_WF_BLOCKS_PER_CU");
        per_cu = e ? atoi(e) : WF_WAVES_EU;
        if (per_cu < 1 || per_cu > 8) per_cu = WF_WAVES_EU;
    }
    return num_cus * per_cu;
}
size_t wavefront_overflow_ints(int num_cus) { return (size_t)WF_OVF * (size_t)wavefront_trace_grid(num_cus) * BLOCK; }
size_t wavefront_regions(size_t slots) { return (slots + REGION - 1) / REGION; }
// unsigned ints of per-frame claim cursors: one SHARDS-wide row per k_trace launch
size_t wavefront_cursor_words(int levels, int n_chunks)
{
    return (size_t)SHARDS * CURSOR_STRIDE * (size_t)levels * (size_t)(1 + n_chunks);
}
size_t wavefront_unit_stride(size_t slots, int n_lights)
{
    return (size_t)((n_lights < 32 ? n_lights : 32) + 1) * wavefront_regions(slots) * UPR;
}
int wavefront_unit_launches(int levels, int n_chunks) { return levels * (1 + n_chunks) + 1; }
bool wavefront_packet_possible(const DevScene &scene, const FrameParams &fp) { return scene.fnodes && 3 * fp.wide_depth + 1 <= PK_STACK; }

int wavefront_frame(const DevScene &scene, const DevCamera &cam, const FrameParams &fp, FrameWorkspace &ws, void *d_rgb8,
                    void *d_hit_id, void *d_rgb32, void *d_hit_t, hipStream_t stream, int num_cus,
                    const std::function<int()> &stamp, int *n_launches, std::string *err)
{
    const int n_rows = fp.rows.n_rows;
    const int n_pixels = fp.W * n_rows;
    const int tiles = ((fp.W + 7) / 8) * ((n_rows + 7) / 8);
    const unsigned int n_slots = (unsigned)tiles * 64u;
    const int n_regions = (int)wavefront_regions(n_slots);
    const int n_chunks = (scene.n_lights + 31) / 32;
    const int trace_grid = wavefront_trace_grid(num_cus);
    const bool count = fp.count_tests == 1;                              // every statistic
    const int counting = fp.count_tests == 2 ? 2 : (count ? 1 : 0);      // 2: the claim units' costs only (FrameParams::count_tests)
    if (3 * fp.wide_depth + 1 > WF_LDS_STACK + WF_OVF) {
        if (err) *err = "LBVH deeper than the traversal stack";
        return RT_E_LIMIT;
    }
    // (the claim cursors were cleared with the frame counters: trace_frame)

    TraceArgs t;
    t.sc = scene; t.cam = cam; t.W = fp.W; t.n_rows = n_rows; t.rows = fp.rows;
    t.hits = ws.p_hits;
    t.n_slots = n_slots; t.n_regions = n_regions;
    t.overflow = ws.overflow; t.total_lanes = (unsigned)trace_grid * BLOCK;
    t.counters = ws.counters;
    t.n_pix = (unsigned)n_pixels;
    t.n_wnodes = (unsigned)(WIDE_TOP_NODES + (scene.n_refs > 1 ? scene.n_refs - 1 : 1));
    t.q_cap = t.q2_cap = (unsigned long long)ws.pixels_cap;
    t.light0 = 0;

    ShadeArgs s;
    s.sc = scene; s.cam = cam; s.W = fp.W; s.n_rows = n_rows; s.n_pixels = n_pixels; s.rows = fp.rows;
    s.max_depth = fp.max_depth;
    s.n_slots = n_slots; s.n_regions = n_regions;
    s.hits = ws.p_hits;
    s.occ = ws.p_occ; s.slots_cap = (unsigned)ws.slots_cap;
    s.sq_o = ws.sq_o; s.sq_counts = ws.sq_counts;
    s.counters = ws.counters;
    // hit and occlusion planes are indexed [level][pixel] and [level][chunk][pixel]: k_shade_all reads
    // every level of a pixel at the end of the frame
    const int occ_chunks = n_chunks > 0 ? n_chunks : 1;
    auto hits_of = [&](int level) { return ws.p_hits + (size_t)level * (size_t)n_pixels; };
    auto occ_of = [&](int level, int chunk) { return ws.p_occ + ((size_t)level * occ_chunks + chunk) * (size_t)n_pixels; };

    const int shade_grid = n_regions < num_cus * 8 ? n_regions : num_cus * 8;
    unsigned int *cursor = ws.trace_cursors;
    int launches = 0;
    // Diagnostic: with RT_WF_TIMES set (and test counting on, or a -DWF_TIMES=1 build for undisturbed kernels)
    // every wave of every traversal launch records when it started, when it found nothing left to claim and
    // when it exited; a summary per launch goes to stderr.
    const size_t tw_stride = 3 * (size_t)trace_grid * (BLOCK / 64);
    unsigned long long *d_times = nullptr;
    std::vector<int> tw_modes;
    if ((count || WF_TIMES) && getenv("RT_WF_TIMES")) {
        const size_t n = tw_stride * (size_t)(2 * (fp.max_depth + 2) * (n_chunks + 1));
        if (hipMalloc((void **)&d_times, sizeof(unsigned long long) * n) == hipSuccess) (void)hipMemsetAsync(d_times, 0, sizeof(unsigned long long) * n, stream);
        else d_times = nullptr;
    }
    // device events around every traversal launch (for rt_stats.ms_traverse)
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
    // Level k's shadow rays and level k+1's nearest-hit rays are both known once k_shadow_gen(k) has
    // run, and neither needs the other: they share one traversal launch (WF_MERGE), so a frame of D
    // bounces has D + 2 traversal drains instead of 2 D + 2.
    const bool merge = WF_MERGE != 0;
    // camera rays as wave packets when the caller found them faster for this layout; the packet's stack is one register
    const bool packet = fp.camera_packet != 0 && wavefront_packet_possible(scene, fp);
    auto launch_gen = [&](int level, int chunk) {
        s.level = level; s.chunk = chunk;
        s.q_o = ws.queue_o[level & 1]; s.q_d = ws.queue_d[level & 1];
        s.in_counts = ws.rq_counts[level & 1];
        s.hits = hits_of(level);
        s.occ = occ_of(level, chunk);
        s.nq_o = ws.queue_o[(level + 1) & 1]; s.nq_d = ws.queue_d[(level + 1) & 1];
        s.out_counts = ws.rq_counts[(level + 1) & 1];
        s.next_hits = hits_of(level < fp.max_depth ? level + 1 : level);      // (nothing spawns at the last level)
        if (level == 0) hipLaunchKernelGGL(k_shadow_gen<true>, dim3(shade_grid), dim3(BLOCK), 0, stream, s);
        else hipLaunchKernelGGL(k_shadow_gen<false>, dim3(shade_grid), dim3(BLOCK), 0, stream, s);
        launches++;
    };
    // mode 0: nearest hit of `level`; 1: shadow rays of (level, chunk); 2: shadow rays of (level, chunk) + nearest hit of level + 1
    int launch_index = 0;
    auto launch_traversal = [&](int mode, int level, int chunk) -> int {
        t.cursors = cursor; cursor += SHARDS * CURSOR_STRIDE;
        const bool units_ok = ws.unit_cost && launch_index < ws.unit_launches;
        t.cost = (fp.track_units && counting && units_ok) ? ws.unit_cost + (size_t)launch_index * ws.unit_stride : nullptr;
        t.order = (fp.use_order && units_ok) ? ws.unit_order + (size_t)launch_index * ws.unit_stride : nullptr;
        launch_index++;
        const int close_level = mode == 2 ? level + 1 : level;
        t.hits = hits_of(close_level);
        if (mode == 0) {
            t.q_o = ws.queue_o[level & 1]; t.q_d = ws.queue_d[level & 1];
            t.q_cap = (unsigned long long)ws.pixels_cap;
            t.counts = ws.rq_counts[level & 1];
            t.occ = nullptr;
            t.n_segments = 1; t.seg_stride = 0;
        } else {
            t.q_o = ws.sq_o; t.q_d = nullptr;
            t.q_cap = (unsigned long long)ws.slots_cap * (unsigned long long)(ws.seg_cap ? ws.seg_cap : 1);
            t.counts = ws.sq_counts;
            t.occ = occ_of(level, chunk);
            t.n_segments = (chunk == n_chunks - 1) ? scene.n_lights - 32 * chunk : 32;
            t.light0 = 32 * chunk;
            t.seg_stride = (unsigned)ws.slots_cap;
            t.q2_o = ws.queue_o[(level + 1) & 1]; t.q2_d = ws.queue_d[(level + 1) & 1];
            t.counts2 = ws.rq_counts[(level + 1) & 1];
        }
        const bool primary = mode == 0 && level == 0;
        int grid = (primary && n_regions * UPR < trace_grid * (BLOCK / 64)) ? (n_regions * UPR + 3) / 4 : trace_grid;
        if (mode == 2 && grid > num_cus * WF_WAVES_EU_MIXED) grid = num_cus * WF_WAVES_EU_MIXED;   // resident blocks only
        t.wave_times = d_times ? d_times + tw_stride * tw_modes.size() : nullptr;
        tw_modes.push_back(mode);
        if (int rc = tstamp()) return rc;
        if (primary && packet) {
            // one wave per 8x8 tile: 8 waves per SIMD fit (few vector registers, no LDS)
            const int pgrid = num_cus * PK_WAVES;
            if (counting == 1) hipLaunchKernelGGL(k_trace_packet<1>, dim3(pgrid), dim3(BLOCK), 0, stream, t);
            else if (counting == 2) hipLaunchKernelGGL(k_trace_packet<2>, dim3(pgrid), dim3(BLOCK), 0, stream, t);
            else hipLaunchKernelGGL(k_trace_packet<0>, dim3(pgrid), dim3(BLOCK), 0, stream, t);
        } else if (primary) launch_trace<0, true>(t, counting, grid, stream);
        else if (mode == 0) launch_trace<0, false>(t, counting, grid, stream);
        else if (mode == 1) launch_trace<1, false>(t, counting, grid, stream);
        else launch_trace<2, false>(t, counting, grid, stream);
        launches++;
        return tstamp();
    };
    if (int rc = launch_traversal(0, 0, 0)) return rc;
    if (fp.camera_only) {
        RT_HIP_CHECK(hipGetLastError(), err);
        *n_launches += launches;
        return RT_OK;
    }
    for (int level = 0; level <= fp.max_depth; level++) {
        const bool more = level < fp.max_depth;
        // chunk 0 also spawns the next level's rays; it runs even without lights
        launch_gen(level, 0);
        bool next_traced = false;
        if (n_chunks > 0) {
            if (merge && more) {
                if (int rc = launch_traversal(2, level, 0)) return rc;
                next_traced = true;
            } else {
                if (int rc = launch_traversal(1, level, 0)) return rc;
            }
        }
        for (int chunk = 1; chunk < n_chunks; chunk++) {
            launch_gen(level, chunk);
            if (int rc = launch_traversal(1, level, chunk)) return rc;
        }
        if (int rc = stamp()) return rc;
        if (more && !next_traced)
            if (int rc = launch_traversal(0, level + 1, 0)) return rc;
    }
    RT_HIP_CHECK(hipGetLastError(), err);
    if (d_times) {
        std::vector<unsigned long long> h(tw_stride * tw_modes.size());
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), d_times, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        (void)hipFree(d_times);
        for (size_t L = 0; L < tw_modes.size(); L++) {
            std::vector<double> ex, en;
            unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t w = 0; w < tw_stride / 3; w++) {
                const unsigned long long *q = &h[L * tw_stride + 3 * w];
                if (q[2] == 0) continue;
                t0 = q[0] < t0 ? q[0] : t0;
                t1 = q[2] > t1 ? q[2] : t1;
            }
            for (size_t w = 0; w < tw_stride / 3; w++) {
                const unsigned long long *q = &h[L * tw_stride + 3 * w];
                if (q[2] == 0) continue;
                ex.push_back((double)(q[1] - t0) * 0.01);      // 100 MHz clock -> microseconds
                en.push_back((double)(q[2] - t0) * 0.01);
            }
            std::sort(ex.begin(), ex.end());
            std::sort(en.begin(), en.end());
            if (ex.empty()) continue;
            auto pct = [](const std::vector<double> &v, double p) { return v[(size_t)(p * (double)(v.size() - 1))]; };
            fprintf(stderr, "[rt] launch %zu mode %d: %zu waves, span %.1f us; nothing left to claim at p10 %.1f p50 %.1f p90 %.1f max %.1f; exit at p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f\n",
                    L, tw_modes[L], ex.size(), (double)(t1 - t0) * 0.01, pct(ex, 0.1), pct(ex, 0.5), pct(ex, 0.9), ex.back(), pct(en, 0.1),
                    pct(en, 0.5), pct(en, 0.9), pct(en, 0.99), en.back());
        }
    }
    ws.trav_pairs = (int)(te / 2);
    if (counting == 2) {        // a pass made for the units' costs: its picture is not wanted
        *n_launches += launches;
        return RT_OK;
    }
    // ---- every level's local colour, the fold and the outputs, one thread per pixel
    if (int rc = shade_all_launch(scene, cam, fp, ws.p_hits, ws.p_occ, occ_chunks, ws.records, ws.counters, d_rgb8, d_hit_id,
                                  d_rgb32, d_hit_t, stream, num_cus, err))
        return rc;
    launches++;
    if (int rc = stamp()) return rc;
    *n_launches += launches;
    return RT_OK;
}
