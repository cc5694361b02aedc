// valu_ceiling.hip — how many vector instructions one SIMD of gfx950 issues per cycle, measured per opcode.
//
// The traversal kernel's roofline argument needs the issue ceiling of a SIMD that holds w resident waves of
// independent VALU work, for the opcodes its node step is made of.  Every wave runs ITER x 32 independent
// instructions of one kind (inline assembly on eight accumulators, so neither the compiler nor a dependency
// decides the rate) and stamps the shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) around
// the loop.  Grids of w blocks of 256 threads per CU put w waves on every SIMD (w <= 4: one round; the table
// also prints the kernel's own duration, which is what counts when blocks run in more than one round).
//
//   cycles per instruction on one SIMD = kernel cycles / (instructions per wave * waves per SIMD)
//
// Build: hipcc --offload-arch=gfx950 -O3 profiles/valu_ceiling.hip -o profiles/valu_ceiling
// Run:   profiles/valu_ceiling            (prints the table kept in profiles/valu_ceiling.txt)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITER = 4096;
typedef float V2 __attribute__((ext_vector_type(2)));

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

// One kernel per opcode: T is the accumulator type (float or V2), OP(k) the instruction on accumulator k with
// operands %8 (a) and %9 (b).
#define DEFINE_KERNEL(NAME, T, OP, ...)                                                                           \
    __global__ __launch_bounds__(256) void NAME(T a, T b, unsigned long long *stamps, float *sink)               \
    {                                                                                                             \
        T acc[8];                                                                                                 \
        for (int k = 0; k < 8; k++) acc[k] = a * (float)(threadIdx.x + k);                                        \
        asm volatile("s_mov_b64 vcc, 0x5555\n\ts_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");                                                          \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();       \
        for (int it = 0; it < ITER; it++)                                                                         \
            asm volatile(REP32(OP)                                                                                \
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),   \
                           "+v"(acc[7])                                                                           \
                         : "v"(a), "v"(b)                                                                         \
                         : __VA_ARGS__);                                                                           \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();       \
        T s = acc[0];                                                                                             \
        for (int k = 1; k < 8; k++) s += acc[k];                                                                  \
        if (*(float *)&s == 12345.678f) sink[0] = 1.0f;                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                            \
            const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);                                         \
            stamps[2 * w] = c1 - c0;                                                                              \
            stamps[2 * w + 1] = r1 - r0;                                                                          \
        }                                                                                                         \
    }

#define OP_FMA(k)     "v_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
#define OP_MUL(k)     "v_mul_f32 %" #k ", %8, %" #k "\n\t"
#define OP_ADD(k)     "v_add_f32 %" #k ", %8, %" #k "\n\t"
#define OP_PKFMA(k)   "v_pk_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
#define OP_PKMUL(k)   "v_pk_mul_f32 %" #k ", %8, %" #k "\n\t"
#define OP_PERM(k)    "v_perm_b32 %" #k ", %8, %" #k ", %9\n\t"
#define OP_MAX(k)     "v_max_f32 %" #k ", %8, %" #k "\n\t"
#define OP_MIN(k)     "v_min_f32 %" #k ", %8, %" #k "\n\t"
#define OP_MAX3(k)    "v_max3_f32 %" #k ", %8, %" #k ", %9\n\t"
#define OP_MED3(k)    "v_med3_f32 %" #k ", %8, %" #k ", %9\n\t"
#define OP_MINU(k)    "v_min_u32 %" #k ", %8, %" #k "\n\t"
#define OP_AND(k)     "v_and_b32 %" #k ", %8, %" #k "\n\t"
#define OP_BFI(k)     "v_bfi_b32 %" #k ", %8, %" #k ", %9\n\t"
#define OP_LSHL(k)    "v_lshlrev_b32 %" #k ", 3, %" #k "\n\t"
#define OP_ADDU(k)    "v_add_u32 %" #k ", %8, %" #k "\n\t"
#define OP_MOV(k)     "v_mov_b32 %" #k ", %8\n\t"
#define OP_CNDM(k)    "v_cndmask_b32_e32 %" #k ", %8, %" #k ", vcc\n\t"
#define OP_CNDM64(k)  "v_cndmask_b32_e64 %" #k ", %8, %" #k ", s[20:21]\n\t"
#define OP_CMP(k)     "v_cmp_lt_f32_e32 vcc, %8, %" #k "\n\t"
#define OP_CMPSEL(k)  "v_cmp_lt_f32_e32 vcc, %8, %" #k "\n\tv_cndmask_b32_e32 %" #k ", %9, %" #k ", vcc\n\t"
#define OP_RCP(k)     "v_rcp_f32 %" #k ", %" #k "\n\t"
#define OP_SQRT(k)    "v_sqrt_f32 %" #k ", %" #k "\n\t"

#define OP_FMAMIXLO(k) "v_fma_mix_f32 %" #k ", %8, %9, %" #k " op_sel_hi:[1,0,0]\n\t"
#define OP_FMAMIXHI(k) "v_fma_mix_f32 %" #k ", %8, %9, %" #k " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#define OP_CVTUB1(k)  "v_cvt_f32_ubyte1 %" #k ", %" #k "\n\t"
#define OP_CVTF16(k)  "v_cvt_f32_f16 %" #k ", %" #k "\n\t"
#define OP_CVTF16HI(k) "v_cvt_f32_f16_sdwa %" #k ", %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
#define OP_CVTBF16(k) "v_cvt_f32_bf16 %" #k ", %" #k "\n\t"
#define OP_CVTU32(k)  "v_cvt_f32_u32 %" #k ", %" #k "\n\t"
#define OP_MIN3U(k)   "v_min3_u32 %" #k ", %8, %" #k ", %9\n\t"
#define OP_MAXU(k)    "v_max_u32 %" #k ", %8, %" #k "\n\t"
#define OP_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n\t"
#define OP_LSHLOR(k)  "v_lshl_or_b32 %" #k ", %" #k ", 8, %8\n\t"
#define OP_ANDOR(k)   "v_and_or_b32 %" #k ", %" #k ", %8, %9\n\t"
#define OP_BFE(k)     "v_bfe_u32 %" #k ", %" #k ", 8, 8\n\t"
#define OP_SUB(k)     "v_sub_f32 %" #k ", %8, %" #k "\n\t"
#define OP_OR(k)      "v_or_b32 %" #k ", %8, %" #k "\n\t"
#define OP_XOR(k)     "v_xor_b32 %" #k ", %8, %" #k "\n\t"
#define OP_MADU24(k)  "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n\t"
#define OP_BITOP3(k)  "v_bitop3_b32 %" #k ", %" #k ", %8, %9 bitop3:0xca\n\t"
#define OP_PKFMAH(k)  "v_pk_fma_f16 %" #k ", %8, %9, %" #k "\n\t"
#define OP_PKMAXH(k)  "v_pk_max_f16 %" #k ", %8, %" #k "\n\t"
#define OP_MULSDWA(k) "v_mul_f32_sdwa %" #k ", %8, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define OP_CMPU(k)    "v_cmp_lt_u32_e32 vcc, %8, %" #k "\n\t"
#define OP_ADDCO(k)   "v_add_co_u32_e32 %" #k ", vcc, %8, %" #k "\n\t"
#define OP_MULLO(k)   "v_mul_lo_u32 %" #k ", %8, %" #k "\n\t"
DEFINE_KERNEL(k_fmamix_lo, float, OP_FMAMIXLO, "memory")
DEFINE_KERNEL(k_fmamix_hi, float, OP_FMAMIXHI, "memory")
DEFINE_KERNEL(k_cvt_ubyte1, float, OP_CVTUB1, "memory")
DEFINE_KERNEL(k_cvt_f16, float, OP_CVTF16, "memory")
DEFINE_KERNEL(k_cvt_f16_hi, float, OP_CVTF16HI, "memory")
DEFINE_KERNEL(k_cvt_bf16, float, OP_CVTBF16, "memory")
DEFINE_KERNEL(k_cvt_u32, float, OP_CVTU32, "memory")
DEFINE_KERNEL(k_min3u, float, OP_MIN3U, "memory")
DEFINE_KERNEL(k_maxu, float, OP_MAXU, "memory")
DEFINE_KERNEL(k_lshl_add, float, OP_LSHLADD, "memory")
DEFINE_KERNEL(k_lshl_or, float, OP_LSHLOR, "memory")
DEFINE_KERNEL(k_and_or, float, OP_ANDOR, "memory")
DEFINE_KERNEL(k_bfe, float, OP_BFE, "memory")
DEFINE_KERNEL(k_sub, float, OP_SUB, "memory")
DEFINE_KERNEL(k_or, float, OP_OR, "memory")
DEFINE_KERNEL(k_xor, float, OP_XOR, "memory")
DEFINE_KERNEL(k_mad_u24, float, OP_MADU24, "memory")
DEFINE_KERNEL(k_bitop3, float, OP_BITOP3, "memory")
DEFINE_KERNEL(k_pk_fma_f16, float, OP_PKFMAH, "memory")
DEFINE_KERNEL(k_pk_max_f16, float, OP_PKMAXH, "memory")
DEFINE_KERNEL(k_mul_sdwa, float, OP_MULSDWA, "memory")
DEFINE_KERNEL(k_cmp_u32, float, OP_CMPU, "vcc")
DEFINE_KERNEL(k_add_co, float, OP_ADDCO, "vcc")
DEFINE_KERNEL(k_mul_lo, float, OP_MULLO, "memory")
DEFINE_KERNEL(k_fma, float, OP_FMA, "memory")
DEFINE_KERNEL(k_mul, float, OP_MUL, "memory")
DEFINE_KERNEL(k_add, float, OP_ADD, "memory")
DEFINE_KERNEL(k_pkfma, V2, OP_PKFMA, "memory")
DEFINE_KERNEL(k_pkmul, V2, OP_PKMUL, "memory")
DEFINE_KERNEL(k_perm, float, OP_PERM, "memory")
DEFINE_KERNEL(k_max, float, OP_MAX, "memory")
DEFINE_KERNEL(k_min, float, OP_MIN, "memory")
DEFINE_KERNEL(k_max3, float, OP_MAX3, "memory")
DEFINE_KERNEL(k_med3, float, OP_MED3, "memory")
DEFINE_KERNEL(k_minu, float, OP_MINU, "memory")
DEFINE_KERNEL(k_and, float, OP_AND, "memory")
DEFINE_KERNEL(k_bfi, float, OP_BFI, "memory")
DEFINE_KERNEL(k_lshl, float, OP_LSHL, "memory")
DEFINE_KERNEL(k_addu, float, OP_ADDU, "memory")
DEFINE_KERNEL(k_mov, float, OP_MOV, "memory")
DEFINE_KERNEL(k_cndm, float, OP_CNDM, "memory")
DEFINE_KERNEL(k_cndm64, float, OP_CNDM64, "s20", "s21")
DEFINE_KERNEL(k_cmp, float, OP_CMP, "vcc")
DEFINE_KERNEL(k_cmpsel, float, OP_CMPSEL, "vcc")
DEFINE_KERNEL(k_rcp, float, OP_RCP, "memory")
DEFINE_KERNEL(k_sqrt, float, OP_SQRT, "memory")

template <typename T, typename K>
int run(const char *name, K kernel, int cus, int per_op)
{
    unsigned long long *d_st = nullptr;
    float *d_sink = nullptr;
    CHECK(hipMalloc((void **)&d_st, sizeof(unsigned long long) * 2 * 4 * 8 * (size_t)cus));
    CHECK(hipMalloc((void **)&d_sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    T a, b;
    { float fa = 1.0001f, fb = 0.5f; a = T(fa); b = T(fb); }
    for (int w : { 1, 2, 8 }) {
        const int grid = cus * w;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, a, b, d_st, d_sink);   // warm-up
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, a, b, d_st, d_sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> st(2 * 4 * (size_t)grid);
        CHECK(hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
        std::vector<double> cyc, ghz;
        for (size_t k = 0; k < st.size() / 2; k++) {
            cyc.push_back((double)st[2 * k]);
            if (st[2 * k + 1]) ghz.push_back((double)st[2 * k] / ((double)st[2 * k + 1] * 10.0));   // cycles per 10 ns tick -> GHz
        }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double med = cyc[cyc.size() / 2], clk = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
        const double insts = (double)ITER * 32.0 * per_op;
        const double kernel_cycles = ms * 1e-3 * clk * 1e9;
        printf("%-18s waves/SIMD %d  wave: %6.2f cycles per instruction  kernel: %6.2f SIMD-cycles per instruction (%.3f per cycle)  clock %4.2f GHz  kernel %7.3f ms\n",
               name, w, med / insts, kernel_cycles / (insts * w), insts * w / kernel_cycles, clk, ms);
    }
    (void)hipFree(d_st); (void)hipFree(d_sink);
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs, clock %d MHz; %d x 32 independent instructions per wave, w blocks of 256 threads per CU\n", p.gcnArchName, cus, p.clockRate / 1000, ITER);
#define RUN(T, K, N) if (run<T>(#K, K, cus, N)) return 1;
    RUN(float, k_fma, 1) RUN(float, k_mul, 1) RUN(float, k_add, 1) RUN(V2, k_pkfma, 1) RUN(V2, k_pkmul, 1)
    RUN(float, k_perm, 1) RUN(float, k_max, 1) RUN(float, k_min, 1) RUN(float, k_max3, 1) RUN(float, k_med3, 1)
    RUN(float, k_minu, 1) RUN(float, k_and, 1) RUN(float, k_bfi, 1) RUN(float, k_lshl, 1) RUN(float, k_addu, 1) RUN(float, k_mov, 1)
    RUN(float, k_fmamix_lo, 1) RUN(float, k_fmamix_hi, 1) RUN(float, k_cvt_ubyte1, 1) RUN(float, k_cvt_f16, 1) RUN(float, k_cvt_f16_hi, 1)
    RUN(float, k_cvt_bf16, 1) RUN(float, k_cvt_u32, 1) RUN(float, k_min3u, 1) RUN(float, k_maxu, 1) RUN(float, k_lshl_add, 1) RUN(float, k_lshl_or, 1)
    RUN(float, k_and_or, 1) RUN(float, k_bfe, 1) RUN(float, k_sub, 1) RUN(float, k_or, 1) RUN(float, k_xor, 1) RUN(float, k_mad_u24, 1)
    RUN(float, k_bitop3, 1) RUN(float, k_pk_fma_f16, 1) RUN(float, k_pk_max_f16, 1) RUN(float, k_mul_sdwa, 1) RUN(float, k_cmp_u32, 1)
    RUN(float, k_add_co, 1) RUN(float, k_mul_lo, 1)
    RUN(float, k_cndm, 1) RUN(float, k_cndm64, 1) RUN(float, k_cmp, 1) RUN(float, k_cmpsel, 2) RUN(float, k_rcp, 1) RUN(float, k_sqrt, 1)
    return 0;
}
