// Do the matrix pipe and the vector ALU of one SIMD overlap ACROSS waves on gfx950?  Each wave loops over
//   MODE 1: 4 dependent-free v_mfma_f32_32x32x16_bf16            (matrix only)
//   MODE 2: 48 VALU (16 v_exp_f32 + 16 v_pk_fma_f32-halves + ...) (vector only)
//   MODE 3: both, the VALU block consuming the MFMA results and feeding the next MFMAs (the attention's chain)
//   MODE 4: both, independent of each other inside the wave
// with 1, 2 or 4 waves per SIMD (blocks of 256 threads = one wave per SIMD; occupancy set by dynamic LDS).
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o build_tmp/ov && build_tmp/ov
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    extern __shared__ char smem[];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.01f); }
    f32x16 s = {}, o = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 3 || MODE == 4) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, (f32x16){}, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s, 0, 0, 0);
        }
        if (MODE == 3) { for (int i = 0; i < 16; ++i) v[i] = s[i]; }
        if (MODE >= 2 && MODE <= 4) {
            const f32x2 sc = {0.25f, 0.25f}, nm = {-1.f, -1.f};
            f32x2 e[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                f32x2 t = (f32x2){v[2 * r], v[2 * r + 1]} * sc + nm;
                e[r] = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            }
            f32x2 c = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
            v[0] += c[0] + c[1];
#pragma unroll
            for (int r = 0; r < 8; ++r) { a[r] = (__bf16)e[r][0]; }
#pragma unroll
            for (int r = 0; r < 8; ++r) { b[r] = (__bf16)e[r][1]; }
            if (MODE != 3) { for (int i = 1; i < 16; ++i) v[i] = e[i >> 1][i & 1]; }
        }
        if (MODE == 1 || MODE == 3 || MODE == 4) {
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, o, 0, 0, 0);
        }
        if (MODE == 1) { asm volatile("" : "+v"(s), "+v"(o)); }
        if (MODE >= 5) {
            // clean: 4 MFMAs on persistent accumulators (MODE 6, 7), 16 v_exp + 16 v_fma on 16 private registers (MODE 5, 6), no data exchange
            // MODE 8, 9: the same four MFMAs as two back-to-back DEPENDENT pairs (s, s, o, o) instead of alternating accumulators
            if (MODE == 8 || MODE == 9) { s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0); s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s, 0, 0, 0); }
            else if (MODE != 5) { s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0); o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, o, 0, 0, 0); }
            if (MODE != 7 && MODE != 9) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
            }
            if (MODE == 8 || MODE == 9) { o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, o, 0, 0, 0); o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, o, 0, 0, 0); }
            else if (MODE != 5) { s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0); o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, o, 0, 0, 0); }
            if (MODE != 7 && MODE != 9) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
            }
        }
    }
    float r = v[0] + o[0] + s[3] + o[7];
    for (int i = 0; i < 16; ++i) r += v[i];
    if (r == 12345.678f) out[0] = r + smem[0];
}
template <int MODE> void run(const char *name, int wps) {
    float *d; hipMalloc(&d, 4);
    const int iters = 4000, blocks = 256 * wps * 4;          // 4 rounds
    const int lds = wps == 4 ? 36 * 1024 : wps == 2 ? 72 * 1024 : 150 * 1024;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, lds>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256, lds>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 rounds x wps waves x iters iterations
    printf("%-34s %d waves/SIMD: %.3f ms -> %.1f ns per wave-iteration per SIMD\n", name, wps, ms, ms * 1e6 / (4.0 * wps * iters));
}
int main() {
    for (int wps : {1, 2, 4}) {
        run<1>("4 MFMA", wps); run<2>("softmax-like VALU", wps); run<3>("MFMA -> VALU -> MFMA (dependent)", wps); run<4>("MFMA + VALU (independent)", wps);
        run<7>("clean: 4 MFMA", wps); run<5>("clean: 16 exp + 48 fma", wps); run<6>("clean: both, independent", wps); run<9>("clean: 4 MFMA as dependent pairs", wps); run<8>("clean: both, dependent MFMA pairs", wps);
    }
    return 0;
}
