// rt_traverse.h — ray-versus-lattice-box arithmetic shared by the LBVH traversal kernels
// (wavefront.hip, persistent.hip).  The box tests only prune: every surviving triangle runs
// the reference's own intersector (Raytracer/main.cpp:109-129), so nothing here has to round
// like the reference; it has to be conservative, and the lattice boxes contain the padded
// float boxes.
#pragma once

#include "rt_device.h"

namespace {

// v_perm_b32 selectors that lift one 16-bit half of a packed plane pair into float bits
// 0x3f000000 | half << 8 (bit 15 of every stored half is set, so this is 1 + q/32768): byte 3 from the
// constant operand, bytes 2-1 from the half, byte 0 zero.
constexpr unsigned int SEL_LO = 0x0401000cu, SEL_HI = 0x0403020cu, PERM_EXP = 0x3f3f3f3fu;
constexpr float INV_MAX = 1e30f;          // stands in for 1/0 so that lo*inv - o*inv stays finite

typedef float V2 __attribute__((ext_vector_type(2)));

// Entry (x) and exit (y) parameter of the ray against one axis' plane pair, packed lo | hi << 16 on
// the lattice: sel_n / sel_f pick the half the ray meets first / last, and t = f*s + b with
// f = 1 + q/32768 read straight from the permuted bits.
__device__ __forceinline__ V2 plane_pair(unsigned int w, unsigned int sel_n, unsigned int sel_f, float s, float b)
{
    const V2 f = { __uint_as_float(__builtin_amdgcn_perm(PERM_EXP, w, sel_n)),
                   __uint_as_float(__builtin_amdgcn_perm(PERM_EXP, w, sel_f)) };
    const V2 sv = { s, s }, bv = { b, b };
    return __builtin_elementwise_fma(f, sv, bv);
}

// Per-ray constants of the box test: t = f*inv + oi per axis, and the selectors that order each
// plane pair by the sign of the direction.  (Separate values rather than one struct: a struct that
// is written through a reference stays in memory instead of registers.)
__device__ __forceinline__ void slab_ray(const QuantGrid &g, F3 o, F3 d, F3 &inv_out, F3 &oi_out, uint3 &sel_n, uint3 &sel_f)
{
    // 1/d clamped to a finite value: a zero component must not turn lo*inv - o*inv into inf - inf.
    // The hardware reciprocal (1 ulp) is enough here: these values only place the ray against padded boxes.
    F3 inv = mk(fminf(fmaxf(__builtin_amdgcn_rcpf(d.x), -INV_MAX), INV_MAX), fminf(fmaxf(__builtin_amdgcn_rcpf(d.y), -INV_MAX), INV_MAX),
                fminf(fmaxf(__builtin_amdgcn_rcpf(d.z), -INV_MAX), INV_MAX));
    sel_n = make_uint3(inv.x < 0.0f ? SEL_HI : SEL_LO, inv.y < 0.0f ? SEL_HI : SEL_LO, inv.z < 0.0f ? SEL_HI : SEL_LO);
    sel_f = make_uint3(inv.x < 0.0f ? SEL_LO : SEL_HI, inv.y < 0.0f ? SEL_LO : SEL_HI, inv.z < 0.0f ? SEL_LO : SEL_HI);
    // lattice plane q (read as f = 1 + q/32768) maps to t = f*(32768*step*inv) + ((base - o)*inv - 32768*step*inv)
    const F3 oi = mk((g.base[0] - o.x) * inv.x, (g.base[1] - o.y) * inv.y, (g.base[2] - o.z) * inv.z);
    inv = mk(g.step[0] * 32768.0f * inv.x, g.step[1] * 32768.0f * inv.y, g.step[2] * 32768.0f * inv.z);
    inv_out = inv;
    oi_out = mk(oi.x - inv.x, oi.y - inv.y, oi.z - inv.z);
}

}  // namespace
