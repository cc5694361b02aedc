// rt_device.h — device-side arithmetic shared by the trace kernels: the reference's vec3
// operators, intersectors and camera ray, written operation for operation as
// Raytracer/maths.h and Raytracer/main.cpp evaluate them (no contraction, IEEE divide/sqrt).
#pragma once

#include "rt_internal.h"

#include <float.h>

namespace {

constexpr int BLOCK = 256;
constexpr int NODE_SENTINEL = 0x7fffffff;

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 mk(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 f3(float4 v) { return mk(v.x, v.y, v.z); }
__device__ __forceinline__ F3 muls(F3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }            // maths.h:102
__device__ __forceinline__ F3 add(F3 a, F3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }           // maths.h:110
__device__ __forceinline__ F3 mulv(F3 a, F3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }          // maths.h:117
__device__ __forceinline__ F3 sub(F3 a, F3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }           // maths.h:132
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // maths.h:140
__device__ __forceinline__ F3 normalize(F3 a)                                                           // maths.h:158-162
{
    float len = sqrtf(dot(a, a));
    return mk(a.x / len, a.y / len, a.z / len);
}
__device__ __forceinline__ F3 cross(F3 a, F3 b)                                                         // maths.h:165-172
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// powf(x, 20.0f) for x > 0 (main.cpp:203).  Five double multiplications carry a
// relative error below 1e-15, so the float result is the correctly rounded power
// except within that distance of a rounding boundary.
__device__ __forceinline__ float pow20(float x)
{
    double a = (double)x;
    double a2 = a * a, a4 = a2 * a2, a5 = a4 * a, a10 = a5 * a5;
    return (float)(a10 * a10);
}

// ------------------------------------------------------------------ intersectors
__device__ __forceinline__ bool sphere_raycast(const DevAnalytic &s, F3 o, F3 d, float &t)              // main.cpp:56-68
{
    F3 l = sub(mk(s.px, s.py, s.pz), o);
    float distToCenter = dot(l, d);
    if (distToCenter < 0.0f) return false;
    float distToIntersectSq = dot(l, l) - distToCenter * distToCenter;
    if (distToIntersectSq > s.s2) return false;
    t = distToCenter - sqrtf(s.s2 - distToIntersectSq);
    return true;
}

__device__ __forceinline__ bool plane_raycast(const DevAnalytic &p, F3 o, F3 d, float &t)               // main.cpp:84-92
{
    F3 n = mk(p.px, p.py, p.pz);
    float ldotn = dot(n, d);
    if (ldotn == 0.0f) return false;
    t = (p.s - dot(n, o)) / ldotn;
    return t > 0.0f;
}

__device__ __forceinline__ bool tri_raycast(F3 v1, F3 e1, F3 e2, F3 o, F3 d, float &t)                  // main.cpp:109-129
{
    F3 P = cross(d, e2);
    float det = dot(e1, P);
    if (det > -0.0001f && det < 0.0001f) return false;
    float invdet = 1.0f / det;
    F3 T = sub(o, v1);
    float u = dot(T, P) * invdet;
    if (u < 0.0f || u > 1.0f) return false;
    F3 Q = cross(T, e1);
    float v = dot(d, Q) * invdet;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(e2, Q) * invdet;
    return t > 0.0001f;
}

// Guarded builds (-DRT_GUARD=1) are for trying kernel changes: every index that comes out of device data (queue
// positions from region counts, node and triangle references, pixels carried by rays, stack spill slots) is checked
// against its array first.  A bad one is replaced by 0 and counted in FrameCounters::guard_hits, which rt_get_stats
// reports as an error, instead of faulting the card.  In a normal build RTG(i, limit) is just (i).
#ifndef RT_GUARD
#define RT_GUARD 0
#endif
#if RT_GUARD
__device__ __forceinline__ unsigned long long rt_guard(unsigned long long i, unsigned long long limit, FrameCounters *c, unsigned int line)
{
    if (i < limit) return i;
    if (atomicAdd(&c->guard_hits, 1u) == 0u) {      // the first one caught says where to look
        c->guard_line = line;
        c->guard_value = i;
        c->guard_limit = limit;
    }
    return 0ull;
}
#define RTG(i, limit) rt_guard((unsigned long long)(i), (unsigned long long)(limit), A.counters, (unsigned int)__LINE__)
// something a normal build does silently although it would change the image (a traversal stack that is full drops the
// far child): counted like a caught index, so rt_get_stats fails
#define RT_GUARD_EVENT(counters, value, limit) (void)rt_guard((unsigned long long)(value), (unsigned long long)(limit), (counters), (unsigned int)__LINE__)
#else
#define RTG(i, limit) (i)
#define RT_GUARD_EVENT(counters, value, limit) ((void)0)
#endif

struct Tally { unsigned box, tri, analytic; };

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// ------------------------------------------------------------------ nearest hit (main.cpp:147-156)
// `nearest > t` in scene order keeps the lowest id among equal distances; out of
// order (LBVH) the same winner is selected explicitly by (t, id).
template <bool COUNT>
__device__ __forceinline__ void analytic_nearest(const DevScene &sc, F3 o, F3 d, float &best_t, int &best_id, Tally &tl)
{
    for (int i = 0; i < sc.n_analytic; i++) {
        const DevAnalytic a = sc.analytic[i];
        float t;
        bool h = (a.kind == RT_SPHERE) ? sphere_raycast(a, o, d, t) : plane_raycast(a, o, d, t);
        if (h && best_t > t) { best_t = t; best_id = i; }
    }
    if (COUNT) tl.analytic += (unsigned)sc.n_analytic;
}

// Slab test of one child box against the ray interval [0, tmax].
__device__ __forceinline__ bool slab(float lox, float hix, float loy, float hiy, float loz, float hiz, F3 o, F3 inv,
                                     float tmax, float &tnear)
{
    float ax = (lox - o.x) * inv.x, bx = (hix - o.x) * inv.x;
    float ay = (loy - o.y) * inv.y, by = (hiy - o.y) * inv.y;
    float az = (loz - o.z) * inv.z, bz = (hiz - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = tn;
    return tn <= tf;
}

// True when ANY non-light analytic primitive accepts the ray, at any distance (main.cpp:181-188).
template <bool COUNT>
__device__ __forceinline__ bool analytic_occluded(const DevScene &sc, F3 o, F3 d, Tally &tl)
{
    for (int i = 0; i < sc.n_analytic; i++) {
        const DevAnalytic a = sc.analytic[i];
        if (a.is_light) continue;
        float t;
        if (COUNT) tl.analytic++;
        if ((a.kind == RT_SPHERE) ? sphere_raycast(a, o, d, t) : plane_raycast(a, o, d, t)) return true;
    }
    return false;
}

// Camera ray of one pixel (main.cpp:267-270).  Slots walk the frame in 8x8 pixel tiles so a
// wave's 64 primary rays form a compact bundle; returns false for slots outside the frame.
struct PixelRay { F3 o, d; int pixel; bool live; };

__device__ __forceinline__ PixelRay primary_ray(const DevCamera &cam, int W, int n_rows, const rt_rowset &rows, unsigned slot)
{
    PixelRay r;
    const int tiles_per_row = (W + 7) >> 3;
    const int tile = (int)(slot >> 6), w = (int)(slot & 63u);
    const int x = (tile % tiles_per_row) * 8 + (w & 7);
    const int lr = (tile / tiles_per_row) * 8 + (w >> 3);
    r.live = x < W && lr < n_rows;
    r.pixel = lr * W + x;
    const int y = (rows.first_band + (lr / rows.band_rows) * rows.band_stride) * rows.band_rows + lr % rows.band_rows;
    r.o = mk(cam.ox, cam.oy, cam.oz);                                        // main.cpp:267
    float offsetX = (float)x * cam.pitch + cam.off_x;                        // main.cpp:268
    float offsetY = (float)y * cam.pitch + cam.off_y;                        // main.cpp:269
    r.d = normalize(sub(mk(offsetX, offsetY, cam.plane_z), r.o));            // main.cpp:270
    return r;
}

// Opt-in shading normal of a triangle (the reference parses `vn` and never uses it, objloader.cpp:56-63,
// main.cpp:222-226): the corner normals weighted by the barycentrics of Triangle::Raycast (main.cpp:110-124),
// N = normalize(n1*(1-u-v) + n2*u + n3*v), all in the reference's vec3 operators.  A triangle whose three
// corner normals are all zero keeps its geometric normal.
__device__ __forceinline__ F3 smooth_normal(const DevScene &sc, int tri, F3 o, F3 d, F3 flat)
{
    const F3 n1 = f3(sc.tri_vn[3 * (size_t)tri]), n2 = f3(sc.tri_vn[3 * (size_t)tri + 1]), n3 = f3(sc.tri_vn[3 * (size_t)tri + 2]);
    if (n1.x == 0.0f && n1.y == 0.0f && n1.z == 0.0f && n2.x == 0.0f && n2.y == 0.0f && n2.z == 0.0f && n3.x == 0.0f &&
        n3.y == 0.0f && n3.z == 0.0f)
        return flat;
    const F3 v1 = f3(sc.tri_mt[3 * (size_t)tri]), e1 = f3(sc.tri_mt[3 * (size_t)tri + 1]), e2 = f3(sc.tri_mt[3 * (size_t)tri + 2]);
    const F3 P = cross(d, e2);                          // main.cpp:111
    const float invdet = 1.0f / dot(e1, P);             // main.cpp:112, 116
    const F3 T = sub(o, v1);                            // main.cpp:118
    const float u = dot(T, P) * invdet;                 // main.cpp:119
    const F3 Q = cross(T, e1);                          // main.cpp:123
    const float v = dot(d, Q) * invdet;                 // main.cpp:124
    const float w = 1.0f - u - v;
    return normalize(add(add(muls(n1, w), muls(n2, u)), muls(n3, v)));
}

// Texture coordinate of a triangle hit (the reference parses `vt` and never uses it, objloader.cpp:48-55):
// the corner coordinates weighted by the barycentrics of Triangle::Raycast (main.cpp:110-124).  (0, 0) for
// a miss, an analytic primitive, or when no coordinates are loaded.
__device__ __forceinline__ float2 hit_uv(const DevScene &sc, int id, F3 o, F3 d)
{
    if (!sc.tri_vt || id < sc.n_analytic) return make_float2(0.f, 0.f);
    const int tri = id - sc.n_analytic;
    const F3 v1 = f3(sc.tri_mt[3 * (size_t)tri]), e1 = f3(sc.tri_mt[3 * (size_t)tri + 1]), e2 = f3(sc.tri_mt[3 * (size_t)tri + 2]);
    const F3 P = cross(d, e2);                          // main.cpp:111
    const float invdet = 1.0f / dot(e1, P);             // main.cpp:112, 116
    const F3 T = sub(o, v1);                            // main.cpp:118
    const float u = dot(T, P) * invdet;                 // main.cpp:119
    const F3 Q = cross(T, e1);                          // main.cpp:123
    const float v = dot(d, Q) * invdet;                 // main.cpp:124
    const float w = 1.0f - u - v;
    const float2 t1 = sc.tri_vt[3 * (size_t)tri], t2 = sc.tri_vt[3 * (size_t)tri + 1], t3 = sc.tri_vt[3 * (size_t)tri + 2];
    return make_float2(t1.x * w + t2.x * u + t3.x * v, t1.y * w + t2.y * u + t3.y * v);
}

// Surface data of a hit primitive: normal (main.cpp:70-73, 94-97, 131-134) and material.  (o, d) is the
// ray that hit; only the opt-in shading normals read it.
struct Surface { F3 N, Mc; float kr, kd, ks; bool emissive; };

__device__ __forceinline__ Surface surface_at(const DevScene &sc, int id, F3 pos, F3 o, F3 d)
{
    Surface s;
    if (id < sc.n_analytic) {
        const DevAnalytic a = sc.analytic[id];
        s.N = (a.kind == RT_SPHERE) ? normalize(sub(pos, mk(a.px, a.py, a.pz))) : mk(a.px, a.py, a.pz);
        s.Mc = mk(a.cr, a.cg, a.cb);
        s.kr = a.reflect; s.kd = a.diffuse; s.ks = a.spec;
        s.emissive = a.is_light != 0;
    } else {
        float4 tn = sc.tri_n[id - sc.n_analytic];
        s.N = f3(tn);
        if (sc.tri_vn) s.N = smooth_normal(sc, id - sc.n_analytic, o, d, s.N);
        const DevMaterial m = sc.mats[__float_as_int(tn.w)];
        s.Mc = mk(m.cr, m.cg, m.cb);
        s.kr = m.reflect; s.kd = m.diffuse; s.ks = m.spec;
        s.emissive = false;
    }
    return s;
}

// main.cpp:210 — the mirror direction, evaluated as (N*2.0f)*(d.N)
__device__ __forceinline__ F3 mirror_dir(F3 d, F3 N) { return sub(d, muls(muls(N, 2.0f), dot(d, N))); }

// One light's contribution to a hit's local colour (main.cpp:191-204): un-clamped N.L diffuse, then the pow-20
// specular of the incoming ray direction d against L mirrored about N; both are evaluated and then multiplied by
// shade (1, or 0 when the shadow ray was blocked, main.cpp:175, 185), in the reference's operand order.
__device__ __forceinline__ F3 add_light(F3 col, const Surface &s, F3 L, F3 Lc, F3 d, float shade)
{
    if (s.kd > 0.0f) {                                                           // main.cpp:191-195
        const float diffuse = dot(s.N, L) * s.kd;
        col = add(col, muls(muls(mulv(Lc, s.Mc), diffuse), shade));
    }
    if (s.ks > 0.0f) {                                                           // main.cpp:198-204
        const F3 R = sub(L, muls(muls(s.N, dot(L, s.N)), 2.0f));
        const float dt = dot(d, R);
        if (dt > 0.0f) col = add(col, muls(muls(muls(mulv(Lc, s.Mc), pow20(dt)), s.ks), shade));
    }
    return col;
}

// (char)(clamp01(c)*255.0f) as the reference's Debug build evaluates it (main.cpp:139-142, 273):
// truncate toward zero, keep the low byte; NaN passes the clamp and converts to the low byte of INT_MIN = 0.
__device__ __forceinline__ uint8_t quantise8(float c)
{
    const float f = (c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c)) * 255.0f;
    return (f == f) ? (uint8_t)(int)f : (uint8_t)0;
}

}  // namespace
