// Issue rate of v_exp_f32, v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_cvt_pk_bf16_f32 on gfx950: 8 independent chains per wave, 4 waves per SIMD
// (1024 blocks of 256 threads), cycles per instruction per SIMD from the kernel time.   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-6f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
                if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
                if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 6) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(0x07060302u));
                if (OP == 7) asm volatile("v_dot2_f32_bf16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(0x3f803f80u));
                if (OP == 8) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(a[i]) : "v"(0x3f803f80u));
                if (OP == 9) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
                if (OP == 10) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 11) asm volatile("v_mov_b32 %0, %0" : "+v"(a[i]));
                // mixes: is the transcendental unit a separate port?  (one exp + three fma: sum or max of the parts)
                if (OP == 12) { if ((i & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i])); }
                if (OP == 13) { if ((i & 1) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); else asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i])); }
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int OP> void run(const char *name) {
    float *d; hipMalloc(&d, 4);
    const int iters = 2000, blocks = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 1.f); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters, 1.f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD: blocks * 4 waves / (256 CUs * 4 SIMDs); instructions per wave: iters * 64
    const double waves_per_simd = blocks * 4.0 / 1024.0, inst = iters * 64.0;
    printf("%-22s %.3f ms  -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (waves_per_simd * inst));
}
int main() { run<0>("v_exp_f32"); run<5>("v_rcp_f32"); run<1>("v_fma_f32"); run<2>("v_pk_fma_f32"); run<3>("v_pk_add_f32"); run<4>("v_cvt_pk_bf16_f32");
    run<6>("v_perm_b32"); run<7>("v_dot2_f32_bf16"); run<8>("v_dot2c_f32_bf16"); run<9>("v_exp_f16"); run<10>("v_add_f32"); run<11>("v_mov_b32");
    run<12>("1 exp + 3 fma (avg)"); run<13>("1 exp + 1 pk_fma (avg)"); return 0; }
