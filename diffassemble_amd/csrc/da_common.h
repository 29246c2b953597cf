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

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ inline float apply_act(float x, int act) {
    if (act == DA_ACT_GELU) return gelu_erf(x);
    if (act == DA_ACT_LEAKY02) return x > 0.f ? x : 0.2f * x;
    return x;
}

inline size_t esize(int prec) { return prec == DA_PREC_BF16 ? 2 : 4; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace da
