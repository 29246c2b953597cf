// Issue-port / matrix-pipe model of one gfx950 SIMD, measured (round 5): how many cycles do v_exp_f32, v_cvt_pk_bf16_f32,
// v_dot2c_f32_bf16 and v_mfma_f32_32x32x16_bf16 cost a wave, alone and beside other waves of the same SIMD, and do one wave's
// MFMAs overlap ANOTHER wave's VALU work?   tools/bin/issue_probe   (prints cycles per loop body per wave, s_memtime)
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/bin/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#define EXP4(x) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
#define CVT2(d, x) asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5" : "=v"(d[0]), "=v"(d[1]) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]))
#define DOT2(s, d) asm volatile("v_dot2c_f32_bf16 %0, %1, %2\n\tv_dot2c_f32_bf16 %0, %1, %3" : "+v"(s) : "v"(one), "v"(d[0]), "v"(d[1]))
#define FMA4(x) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// mode: what a wave's loop body is; split: odd workgroups run `mode2` instead (role split across the waves of a SIMD)
template <int MODE>
__device__ __forceinline__ void body(f32x16 &acc, f32x16 &acc2, float (&x)[16], unsigned (&d)[8], float &s, const u32x4 &a, const u32x4 &b, unsigned one) {
    if (MODE == 0) { EXP4((x + 0)); EXP4((x + 4)); EXP4((x + 8)); EXP4((x + 12)); }                       // 16 exps
    if (MODE == 1) { MFMA(acc); MFMA(acc); MFMA(acc); MFMA(acc); }                                         // 4 MFMAs, one accumulator
    if (MODE == 2) { MFMA(acc); MFMA(acc2); MFMA(acc); MFMA(acc2); }                                       // 4 MFMAs, two accumulators
    if (MODE == 3) { MFMA(acc); EXP4((x + 0)); MFMA(acc); EXP4((x + 4)); MFMA(acc2); EXP4((x + 8)); MFMA(acc2); EXP4((x + 12)); }    // interleaved
    if (MODE == 4) { MFMA(acc); MFMA(acc); MFMA(acc2); MFMA(acc2); EXP4((x + 0)); EXP4((x + 4)); EXP4((x + 8)); EXP4((x + 12)); }    // burst + burst
    if (MODE == 5) { CVT2((d + 0), (x + 0)); CVT2((d + 2), (x + 4)); CVT2((d + 4), (x + 8)); CVT2((d + 6), (x + 12)); CVT2((d + 0), (x + 0)); CVT2((d + 2), (x + 4)); CVT2((d + 4), (x + 8)); CVT2((d + 6), (x + 12)); }   // 16 cvt_pk
    if (MODE == 6) { DOT2(s, (d + 0)); DOT2(s, (d + 2)); DOT2(s, (d + 4)); DOT2(s, (d + 6)); DOT2(s, (d + 0)); DOT2(s, (d + 2)); DOT2(s, (d + 4)); DOT2(s, (d + 6)); }   // 16 dot2c, one accumulator
    if (MODE == 7) { FMA4((x + 0)); FMA4((x + 4)); FMA4((x + 8)); FMA4((x + 12)); }                       // 16 plain VALU
    if (MODE == 8) {                                                                                        // the hidden-layer block, burst order: 4 MFMA, 16 exp, 8 cvt, 8 dot
        MFMA(acc); MFMA(acc); EXP4((x + 0)); EXP4((x + 4)); EXP4((x + 8)); EXP4((x + 12));
        CVT2((d + 0), (x + 0)); CVT2((d + 2), (x + 4)); CVT2((d + 4), (x + 8)); CVT2((d + 6), (x + 12));
        DOT2(s, (d + 0)); MFMA(acc2); DOT2(s, (d + 2)); DOT2(s, (d + 4)); DOT2(s, (d + 6)); MFMA(acc2);
    }
    if (MODE == 9) {                                                                                        // same multiset, one MFMA per quarter
        MFMA(acc); EXP4((x + 0)); CVT2((d + 0), (x + 0)); DOT2(s, (d + 0));
        MFMA(acc); EXP4((x + 4)); CVT2((d + 2), (x + 4)); DOT2(s, (d + 2));
        MFMA(acc2); EXP4((x + 8)); CVT2((d + 4), (x + 8)); DOT2(s, (d + 4));
        MFMA(acc2); EXP4((x + 12)); CVT2((d + 6), (x + 12)); DOT2(s, (d + 6));
    }
    if (MODE == 10) {                                                                                       // the last layer's block, burst order: 9 + 2 MFMA, 16 exp, 8 cvt, 8 dot
        MFMA(acc); MFMA(acc); MFMA(acc); MFMA(acc); MFMA(acc); MFMA(acc); MFMA(acc); MFMA(acc); MFMA(acc);
        EXP4((x + 0)); EXP4((x + 4)); EXP4((x + 8)); EXP4((x + 12));
        CVT2((d + 0), (x + 0)); CVT2((d + 2), (x + 4)); CVT2((d + 4), (x + 8)); CVT2((d + 6), (x + 12));
        DOT2(s, (d + 0)); MFMA(acc2); DOT2(s, (d + 2)); DOT2(s, (d + 4)); DOT2(s, (d + 6)); MFMA(acc2);
    }
}

template <int MODE, int MODE2>
__global__ __launch_bounds__(256) void k_probe(unsigned long long *out, int iters, int split) {
    extern __shared__ unsigned char pad[];
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    float x[16];
    for (int r = 0; r < 16; ++r) x[r] = -0.001f * (threadIdx.x + r);
    unsigned d[8] = {0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    float s = 0.f;
    const u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    const unsigned one = 0x3f803f80u;
    const bool second = split && (blockIdx.x & 1);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (!second) { for (int it = 0; it < iters; ++it) body<MODE>(acc, acc2, x, d, s, a, b, one); }
    else { for (int it = 0; it < iters; ++it) body<MODE2>(acc, acc2, x, d, s, a, b, one); }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sink = s;
    for (int r = 0; r < 16; ++r) sink += acc[r] + acc2[r] + x[r];
    for (int r = 0; r < 8; ++r) sink += (float)d[r];
    if ((threadIdx.x & 63) == 0) {
        out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
        out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = (unsigned long long)(sink != 12345.f);
    }
}

template <int MODE, int MODE2>
static void run(const char *name, int occ, int split, int per_body) {
    const int iters = 2000, nwg = 256 * occ;
    unsigned long long *d, *h = (unsigned long long *)malloc(16 * 4 * nwg);
    CK(hipMalloc(&d, 16 * 4 * nwg));
    const int lds = (160 * 1024 / occ) - 1024;            // exactly `occ` workgroups (= waves per SIMD) fit a CU
    CK(hipFuncSetAttribute((const void *)k_probe<MODE, MODE2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    k_probe<MODE, MODE2><<<nwg, 256, lds>>>(d, iters, split);
    k_probe<MODE, MODE2><<<nwg, 256, lds>>>(d, iters, split);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, 16 * 4 * nwg, hipMemcpyDeviceToHost));
    double t[2] = {0, 0}; int c[2] = {0, 0};
    for (int w = 0; w < nwg; ++w) for (int k = 0; k < 4; ++k) { const int role = split && (w & 1); t[role] += (double)h[2 * (w * 4 + k)]; c[role]++; }
    if (!split) printf("%-58s %d waves/SIMD: %7.1f cycles per body (%5.1f per instruction-unit)\n", name, occ, t[0] / c[0] / iters, t[0] / c[0] / iters / per_body);
    else printf("%-58s %d waves/SIMD: role A %7.1f, role B %7.1f cycles per body\n", name, occ, t[0] / c[0] / iters, t[1] / c[1] / iters);
    CK(hipFree(d)); free(h);
}

int main() {
    for (int occ : {1, 2, 4}) {
        run<0, 0>("16 v_exp_f32", occ, 0, 16);
        run<7, 7>("16 v_fma_f32", occ, 0, 16);
        run<5, 5>("16 v_cvt_pk_bf16_f32", occ, 0, 16);
        run<6, 6>("16 v_dot2c_f32_bf16 (one accumulator)", occ, 0, 16);
        run<1, 1>("4 MFMA 32x32x16 bf16, one accumulator", occ, 0, 4);
        run<2, 2>("4 MFMA, two accumulators alternating", occ, 0, 4);
        run<3, 3>("4 x (MFMA + 4 exp) interleaved in the wave", occ, 0, 1);
        run<4, 4>("4 MFMA burst then 16 exp", occ, 0, 1);
        run<8, 8>("hidden block, burst order (4 MFMA 16 exp 8 cvt 8 dot)", occ, 0, 1);
        run<9, 9>("hidden block, one MFMA per quarter", occ, 0, 1);
        run<10, 10>("last-layer block, burst order (11 MFMA 16 exp 8 cvt 8 dot)", occ, 0, 1);
    }
    for (int occ : {2, 4}) {
        run<1, 0>("ROLE SPLIT: A = 4 MFMA (one acc), B = 16 exp", occ, 1, 1);
        run<1, 7>("ROLE SPLIT: A = 4 MFMA (one acc), B = 16 fma", occ, 1, 1);
        run<2, 0>("ROLE SPLIT: A = 4 MFMA (two acc), B = 16 exp", occ, 1, 1);
    }
    return 0;
}
