// Issue rate of vector instructions on one SIMD of gfx950, by instruction and by the number of waves that share the SIMD.
// One workgroup of 256 x W threads per CU puts W waves on each of its four SIMDs (W = 8: two workgroups of 1024). After a
// barrier every wave runs a loop of 64 copies of one instruction (or short sequence) over eight independent accumulators
// (or over ONE, "dep": a dependent chain) between two reads of the shader clock, and records which SIMD it ran on. For
// every SIMD the program then divides the cycles between the first start and the last end of its waves by the
// instructions they issued. No memory traffic inside the timed loop; the kernels store only their two timestamps. No
// instruction in the loops writes scc (the compiler keeps the loop test there).
//
// build: hipcc -O2 --offload-arch=gfx950 -o build_exp/valu_rate tools/valu_rate.hip        run: build_exp/valu_rate [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

struct WaveRec {
    unsigned long long t0, t1;
    uint32_t hwId, xcc;
};

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

__device__ inline void stamp(WaveRec *out, unsigned long long t0, unsigned long long t1) {
    if ((threadIdx.x & 63) == 0) {
        WaveRec r;
        r.t0 = t0;
        r.t1 = t1;
        r.hwId = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_ID
        r.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // XCC_ID, low four bits
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
    }
}

// L(i): the instruction on accumulator i; %8 and %9 are two more vector inputs, %10 a scalar one
#define KERNEL(NAME, T, L)                                                                                              \
    __global__ __launch_bounds__(1024) void NAME(WaveRec *out, int iters, T seed, T bIn, float sIn) {                    \
        T a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed, b = bIn, c = bIn;     \
        asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(sIn), "v"(sIn + (float)(threadIdx.x & 1)) : "vcc");               \
        __syncthreads();                                                                                                \
        const unsigned long long t0 = clock64();                                                                        \
        for (int i = 0; i < iters; ++i)                                                                                 \
            asm volatile(REP8(L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7))                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                         : "v"(b), "v"(c), "s"(sIn));                                                                   \
        const unsigned long long t1 = clock64();                                                                        \
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));                      \
        stamp(out, t0, t1);                                                                                             \
    }
// the same with vcc written inside the loop
#define KERNEL_VCC(NAME, T, L)                                                                                          \
    __global__ __launch_bounds__(1024) void NAME(WaveRec *out, int iters, T seed, T bIn, float sIn) {                    \
        T a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed, b = bIn, c = bIn;     \
        __syncthreads();                                                                                                \
        const unsigned long long t0 = clock64();                                                                        \
        for (int i = 0; i < iters; ++i)                                                                                 \
            asm volatile(REP8(L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7))                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                         : "v"(b), "v"(c), "s"(sIn)                                                                     \
                         : "vcc", "s20", "s21");                                                                                      \
        const unsigned long long t1 = clock64();                                                                        \
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));                      \
        stamp(out, t0, t1);                                                                                             \
    }

#define L_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define L_FMA_DEP(i) "v_fma_f32 %0, %0, %8, %9\n"
#define L_FMA_S(i) "v_fma_f32 %" #i ", %" #i ", %8, %10\n"
#define L_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define L_MUL_E64(i) "v_mul_f32_e64 %" #i ", %" #i ", %8\n"
#define L_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define L_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define L_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define L_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define L_ADD3U(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define L_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %9\n"
#define L_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define L_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define L_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define L_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define L_CMP_SGPR(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n"
#define L_CMP_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define L_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define L_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define L_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define L_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define L_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define L_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define L_DIVFIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define L_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define L_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define L_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define L_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define L_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define L_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define L_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define L_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
#define L_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %8\n"
#define L_FMA_ADDU(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\nv_add_u32 %" #i ", %" #i ", %8\n"
#define L_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define L_SUB(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define L_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define L_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define L_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define L_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define L_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define L_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 8\n"
#define L_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define L_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define L_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define L_CVT64(i) "v_cvt_f64_f32 %" #i ", %8\n"
#define L_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define L_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", -1, %" #i "\n"
#define L_READLANE(i) "v_readlane_b32 s20, %" #i ", 3\n"
#define L_ADD_ABS(i) "v_add_f32_e64 %" #i ", |%" #i "|, %8\n"
#define L_MUL_NEG(i) "v_mul_f32_e64 %" #i ", -%" #i ", %8\n"
#define L_MUL_SGPR(i) "v_mul_f32 %" #i ", %10, %" #i "\n"
#define L_MUL_LIT(i) "v_mul_f32 %" #i ", 0x3f8ccccd, %" #i "\n"
#define L_FMA_MUL(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\nv_mul_f32 %" #i ", %" #i ", %8\n"
#define L_CMP_MUL(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_mul_f32 %" #i ", %" #i ", %8\n"
#define L_CNDMASK_SGPR(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define L_CMP_CND4(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %0, %0, %9, vcc\nv_cndmask_b32 %1, %1, %9, vcc\nv_cndmask_b32 %2, %2, %9, vcc\nv_cndmask_b32 %3, %3, %9, vcc\n"
#define L_CMPS_CND4(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\nv_cndmask_b32_e64 %0, %0, %9, s[20:21]\nv_cndmask_b32_e64 %1, %1, %9, s[20:21]\nv_cndmask_b32_e64 %2, %2, %9, s[20:21]\nv_cndmask_b32_e64 %3, %3, %9, s[20:21]\n"
#define L_CMP_NOP_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_mul_f32 %4, %4, %8\nv_mul_f32 %5, %5, %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define L_NOP(i) "s_nop 0\n"

KERNEL(k_fma, float, L_FMA)
KERNEL(k_fma_dep, float, L_FMA_DEP)
KERNEL(k_fma_sgpr, float, L_FMA_S)
KERNEL(k_mul, float, L_MUL)
KERNEL(k_mul_e64, float, L_MUL_E64)
KERNEL(k_add, float, L_ADD)
KERNEL(k_max, float, L_MAX)
KERNEL(k_max3, float, L_MAX3)
KERNEL(k_add_u32, float, L_ADDU)
KERNEL(k_add3_u32, float, L_ADD3U)
KERNEL(k_lshl_add_u32, float, L_LSHLADD)
KERNEL(k_xor, float, L_XOR)
KERNEL(k_mov, float, L_MOV)
KERNEL(k_cndmask, float, L_CNDMASK)
KERNEL_VCC(k_cmp_vcc, float, L_CMP)
KERNEL_VCC(k_cmp_sgpr, float, L_CMP_SGPR)
KERNEL_VCC(k_cmp_then_cndmask, float, L_CMP_CND)
KERNEL(k_mul_lo_u32, float, L_MULLO)
KERNEL(k_mul_u32_u24, float, L_MUL24)
KERNEL(k_rcp, float, L_RCP)
KERNEL(k_sqrt, float, L_SQRT)
KERNEL_VCC(k_div_scale, float, L_DIVSCALE)
KERNEL(k_div_fmas, float, L_DIVFMAS)
KERNEL(k_div_fixup, float, L_DIVFIXUP)
KERNEL(k_cvt_f32_u32, float, L_CVT)
KERNEL(k_pk_fma, v2f, L_PKFMA)
KERNEL(k_pk_mul, v2f, L_PKMUL)
KERNEL(k_pk_add, v2f, L_PKADD)
KERNEL(k_fma_f64, double, L_FMA64)
KERNEL(k_mul_f64, double, L_MUL64)
KERNEL(k_add_f64, double, L_ADD64)
__global__ __launch_bounds__(1024) void k_mad_u64_u32(WaveRec *out, int iters, unsigned long long seed, unsigned long long bIn, float sIn) {
    unsigned long long a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;
    const uint32_t b = (uint32_t)bIn + threadIdx.x, c = (uint32_t)bIn * 3u;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
        asm volatile(REP8(L_MAD64(0) L_MAD64(1) L_MAD64(2) L_MAD64(3) L_MAD64(4) L_MAD64(5) L_MAD64(6) L_MAD64(7))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c)
                     : "vcc");
    const unsigned long long t1 = clock64();
    asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    stamp(out, t0, t1);
}
KERNEL(k_lshl_add_u64, unsigned long long, L_LSHLADD64)
KERNEL(k_fma_then_add_u32, float, L_FMA_ADDU)
KERNEL(k_fmac, float, L_FMAC)
KERNEL(k_sub, float, L_SUB)
KERNEL(k_min, float, L_MIN)
KERNEL(k_and, float, L_AND)
KERNEL(k_or, float, L_OR)
KERNEL(k_lshlrev, float, L_LSHL)
KERNEL(k_lshrrev, float, L_LSHR)
KERNEL(k_bfe_u32, float, L_BFE)
KERNEL(k_mul_hi_u32, float, L_MULHI)
KERNEL(k_mad_u32_u24, float, L_MAD24)
KERNEL(k_alignbit, float, L_ALIGNBIT)
KERNEL(k_floor, float, L_FLOOR)
KERNEL(k_mbcnt, float, L_MBCNT)
KERNEL_VCC(k_readlane, float, L_READLANE)
KERNEL(k_add_abs_e64, float, L_ADD_ABS)
KERNEL(k_mul_neg_e64, float, L_MUL_NEG)
KERNEL(k_mul_sgpr, float, L_MUL_SGPR)
KERNEL(k_mul_literal, float, L_MUL_LIT)
KERNEL(k_fma_then_mul, float, L_FMA_MUL)
KERNEL_VCC(k_cmp_then_mul, float, L_CMP_MUL)
KERNEL_VCC(k_cmp_then_4_cndmask, float, L_CMP_CND4)
KERNEL_VCC(k_cmp_sgpr_4_cndmask, float, L_CMPS_CND4)
KERNEL_VCC(k_cmp_2mul_cndmask, float, L_CMP_NOP_CND)
__global__ __launch_bounds__(1024) void k_cndmask_sgpr(WaveRec *out, int iters, float seed, float bIn, float sIn) {
    float a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed, b = bIn, c = bIn;
    asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" ::"v"(sIn), "v"(sIn + (float)(threadIdx.x & 1)) : "s20", "s21");
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
        asm volatile(REP8(L_CNDMASK_SGPR(0) L_CNDMASK_SGPR(1) L_CNDMASK_SGPR(2) L_CNDMASK_SGPR(3) L_CNDMASK_SGPR(4) L_CNDMASK_SGPR(5) L_CNDMASK_SGPR(6) L_CNDMASK_SGPR(7))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
    const unsigned long long t1 = clock64();
    asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    stamp(out, t0, t1);
}
KERNEL_VCC(k_s_nop, float, L_NOP)

// 64-bit accumulators of the kernels above need 64-bit scalars for the "s" operand too; the float one is unused there
template <typename T> struct Init { static T seed() { return (T)1.0001f; } static T b() { return (T)0.9999f; } };
template <> struct Init<v2f> { static v2f seed() { return v2f{1.0001f, 1.0002f}; } static v2f b() { return v2f{0.9999f, 0.9998f}; } };
template <> struct Init<unsigned long long> { static unsigned long long seed() { return 12345; } static unsigned long long b() { return 77; } };

struct Result { double perWave, perSimd; int simds, wavesMax; };

template <typename T, typename K> static Result run(K kernel, int wavesPerSimd, int iters, int perLine, int nCU, WaveRec *dOut, std::vector<WaveRec> &h) {
    const int perBlock = wavesPerSimd > 4 ? 4 : wavesPerSimd;   // waves per SIMD that one workgroup brings
    const int blocks = nCU * (wavesPerSimd / perBlock);
    CHECK(hipMemset(dOut, 0, sizeof(WaveRec) * nCU * 8 * 4));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256 * perBlock), 0, 0, dOut, iters, Init<T>::seed(), Init<T>::b(), 0.5f);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    h.resize((size_t)blocks * perBlock * 4);
    CHECK(hipMemcpy(h.data(), dOut, sizeof(WaveRec) * h.size(), hipMemcpyDeviceToHost));
    const double instrPerWave = (double)iters * 64.0 * perLine;
    std::vector<double> perWave;
    struct Acc { unsigned long long lo = ~0ull, hi = 0; int n = 0; };
    std::map<uint64_t, Acc> simd;
    for (const WaveRec &r : h) {
        perWave.push_back((double)(r.t1 - r.t0) / instrPerWave);
        // xcc | se, sh, cu (bits 8..15 of HW_ID) | simd (bits 4..5)
        const uint64_t key = ((uint64_t)r.xcc << 32) | (r.hwId & 0xff30u);
        Acc &a = simd[key];
        a.lo = std::min(a.lo, r.t0);
        a.hi = std::max(a.hi, r.t1);
        a.n++;
    }
    std::sort(perWave.begin(), perWave.end());
    std::vector<double> perSimd;
    int wmax = 0;
    for (auto &kv : simd) {
        perSimd.push_back((double)(kv.second.hi - kv.second.lo) / (instrPerWave * kv.second.n));
        wmax = std::max(wmax, kv.second.n);
    }
    std::sort(perSimd.begin(), perSimd.end());
    return {perWave[perWave.size() / 2], perSimd[perSimd.size() / 2], (int)simd.size(), wmax};
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int nCU = prop.multiProcessorCount;
    printf("# %s, %d CUs, %d iterations of 64 instructions per wave; cycles of the shader clock per instruction:\n", prop.name, nCU, iters);
    printf("# 'wave' = as one wave sees it (median), 'simd' = per instruction issued on the SIMD by all its waves (median over SIMDs)\n");
    WaveRec *dOut;
    CHECK(hipMalloc(&dOut, sizeof(WaveRec) * nCU * 8 * 4));
    std::vector<WaveRec> h;
    const int W[] = {1, 2, 3, 4, 8};
    printf("%-22s", "waves per SIMD:");
    for (int w : W) printf("  %d: wave  simd ", w);
    printf("\n");
#define ROW(NAME, T, PERLINE)                                                          \
    {                                                                                  \
        printf("%-22s", #NAME);                                                        \
        for (int w : W) {                                                              \
            Result r = run<T>(NAME, w, iters, PERLINE, nCU, dOut, h);                  \
            printf("  %7.2f %5.2f%s", r.perWave, r.perSimd, r.wavesMax == w ? " " : "*"); \
        }                                                                              \
        printf("\n");                                                                  \
        fflush(stdout);                                                                \
    }
    ROW(k_fma, float, 1)
    ROW(k_fma_dep, float, 1)
    ROW(k_fma_sgpr, float, 1)
    ROW(k_mul, float, 1)
    ROW(k_mul_e64, float, 1)
    ROW(k_add, float, 1)
    ROW(k_max, float, 1)
    ROW(k_max3, float, 1)
    ROW(k_add_u32, float, 1)
    ROW(k_add3_u32, float, 1)
    ROW(k_lshl_add_u32, float, 1)
    ROW(k_xor, float, 1)
    ROW(k_mov, float, 1)
    ROW(k_cndmask, float, 1)
    ROW(k_cmp_vcc, float, 1)
    ROW(k_cmp_sgpr, float, 1)
    ROW(k_cmp_then_cndmask, float, 2)
    ROW(k_mul_lo_u32, float, 1)
    ROW(k_mul_u32_u24, float, 1)
    ROW(k_rcp, float, 1)
    ROW(k_sqrt, float, 1)
    ROW(k_div_scale, float, 1)
    ROW(k_div_fmas, float, 1)
    ROW(k_div_fixup, float, 1)
    ROW(k_cvt_f32_u32, float, 1)
    ROW(k_pk_fma, v2f, 1)
    ROW(k_pk_mul, v2f, 1)
    ROW(k_pk_add, v2f, 1)
    ROW(k_fma_f64, double, 1)
    ROW(k_mul_f64, double, 1)
    ROW(k_add_f64, double, 1)
    ROW(k_mad_u64_u32, unsigned long long, 1)
    ROW(k_lshl_add_u64, unsigned long long, 1)
    ROW(k_fma_then_add_u32, float, 2)
    ROW(k_fmac, float, 1)
    ROW(k_sub, float, 1)
    ROW(k_min, float, 1)
    ROW(k_and, float, 1)
    ROW(k_or, float, 1)
    ROW(k_lshlrev, float, 1)
    ROW(k_lshrrev, float, 1)
    ROW(k_bfe_u32, float, 1)
    ROW(k_mul_hi_u32, float, 1)
    ROW(k_mad_u32_u24, float, 1)
    ROW(k_alignbit, float, 1)
    ROW(k_floor, float, 1)
    ROW(k_mbcnt, float, 1)
    ROW(k_readlane, float, 1)
    ROW(k_add_abs_e64, float, 1)
    ROW(k_mul_neg_e64, float, 1)
    ROW(k_mul_sgpr, float, 1)
    ROW(k_mul_literal, float, 1)
    ROW(k_fma_then_mul, float, 2)
    ROW(k_cmp_then_mul, float, 2)
    ROW(k_cmp_then_4_cndmask, float, 5)
    ROW(k_cmp_sgpr_4_cndmask, float, 5)
    ROW(k_cmp_2mul_cndmask, float, 4)
    ROW(k_cndmask_sgpr, float, 1)
    ROW(k_s_nop, float, 1)
    {   // the shader clock itself: cycles a wave counts against the time HIP events measure for the same launch
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const int longIters = iters * 40;
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_fma, dim3(nCU * 2), dim3(1024), 0, 0, dOut, longIters, 1.0001f, 0.9999f, 0.5f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        h.resize((size_t)nCU * 2 * 16);
        CHECK(hipMemcpy(h.data(), dOut, sizeof(WaveRec) * h.size(), hipMemcpyDeviceToHost));
        unsigned long long longest = 0;
        for (const WaveRec &r : h) longest = std::max(longest, r.t1 - r.t0);
        printf("# shader clock under this load (8 waves per SIMD of v_fma_f32): %.0f MHz (%llu cycles of the longest wave in a launch of %.3f ms)\n",
               longest / (ms * 1e3), longest, ms);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
    printf("# '*': some SIMD held more waves than that (two workgroups on one CU); the simd column divides by the waves found\n");
    CHECK(hipFree(dOut));
    return 0;
}
