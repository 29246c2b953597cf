// Shared device/host helpers for libdiffassemble_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/diffassemble_hip.h"

namespace da {

typedef unsigned short bf16_t;   // raw bfloat16 bits

void set_error(const char *fmt, ...);

#define DA_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            da::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

#define DA_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            da::set_error(__VA_ARGS__);                                                 \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

#define DA_LAUNCH_CHECK()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            da::set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return 3;                                                                   \
        }                                                                               \
    } while (0)

__host__ __device__ inline float bf2f(bf16_t h) {
    union { unsigned u; float f; } v;
    v.u = ((unsigned)h) << 16;
    return v.f;
}
__host__ __device__ inline bf16_t f2bf(float f) {     // round-to-nearest-even
#if defined(__HIP_DEVICE_COMPILE__)
    // device: the hardware conversion (v_cvt_pk_bf16_f32, RNE: the same bits as the integer sequence below for every non-NaN
    // input) -- one instruction where the sequence costs seven; the small-group attention kernels of the training path round
    // ~80 values per lane this way and are bound by their instruction count
    return __builtin_bit_cast(unsigned short, (__bf16)f);
#endif
    union { unsigned u; float f; } v;
    v.f = f;
    unsigned u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ inline float ldf(const float *p) { return *p; }
__device__ inline float ldf(const bf16_t *p) { return bf2f(*p); }
__device__ inline void stf(float *p, float v) { *p = v; }
__device__ inline void stf(bf16_t *p, float v) { *p = f2bf(v); }

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7): ~12 instructions with one v_exp and one
// v_rcp, instead of the ~35-instruction libm erff that bloats unrolled epilogues past the I-cache.
__device__ inline float erf_as(float z) {
    const float a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-a * a * 1.4426950408889634f);
    const float r = fmaf(-poly * t, e, 1.0f);
    return copysignf(r, z);
}
__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
__device__ inline float gelu_grad(float x) {          // d gelu_erf / dx (training path)
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return fmaf(x, pdf, cdf);
}
__device__ inline float apply_act(float x, int act) {
    if (act == DA_ACT_GELU) return gelu_erf(x);
    if (act == DA_ACT_LEAKY02) return x > 0.f ? x : 0.2f * x;
    return x;
}

// Barrier that also covers this wave's outstanding LDS-DMA (global_load_lds).  hipcc's waitcnt
// insertion does NOT reliably put `s_waitcnt vmcnt(0)` in front of a __syncthreads() whose pending
// DMA was issued in a previous loop iteration (observed: attention main loop compiled to
// `s_waitcnt lgkmcnt(0); s_barrier`, tiles were read before they landed under load) -- so say it.
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// s_waitcnt vmcnt(n) for a small runtime n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// internal precision code of the training path's DA_TRAIN_MMA_BF16 mode: fp32 STORAGE everywhere (esize = 4), GEMM operands
// rounded to bf16 on their way into the matrix cores, fp32 accumulation (da_gemm_common.h Mma16<float, true>)
#define DA_PREC_F32_BF16MMA 2
inline size_t esize(int prec) { return prec == DA_PREC_BF16 ? 2 : 4; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace da
