// Pieces shared by the two MFMA linear kernels (da_gemm_mfma.hip, da_gemm_astat.hip).
#pragma once
#include "da_common.h"
#include "da_internal.h"

namespace da {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// BFC ("bf16 compute", T = float only): fp32 operands in memory and LDS, rounded to bf16 (RNE) in registers on their way
// into the matrix core, fp32 accumulation -- the training path's DA_TRAIN_MMA_BF16 mode (da_train.hip).  The lane's four
// consecutive fp32 K-values are exactly the four K-slots lane group (lane >> 4) feeds v_mfma_f32_16x16x16_bf16: one MFMA
// (16 cycles) + four packs replace four exact-fp32 16x16x4 MFMAs (128 cycles).
template <typename T, bool BFC = false> struct Mma16;
template <> struct Mma16<float, true> {
    static __device__ __forceinline__ f32x4 run(const u32x4 &x, const u32x4 &y, f32x4 c) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
        typedef __attribute__((ext_vector_type(4))) short s16x4_;
        const f32x4 a = __builtin_bit_cast(f32x4, x), b = __builtin_bit_cast(f32x4, y);
        const bf16x4_ ab = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
        const bf16x4_ bb = {(__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_, ab), __builtin_bit_cast(s16x4_, bb), c, 0, 0, 0);
    }
};
template <> struct Mma16<bf16_t, false> {
    static __device__ __forceinline__ f32x4 run(const u32x4 &x, const u32x4 &y, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    }
};
template <> struct Mma16<float, false> {
    static __device__ __forceinline__ f32x4 run(const u32x4 &x, const u32x4 &y, f32x4 c) {
        const f32x4 a = __builtin_bit_cast(f32x4, x), b = __builtin_bit_cast(f32x4, y);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

// 4 consecutive elements; callers guarantee natural alignment (16 B fp32 / 8 B bf16)
__device__ __forceinline__ void store4(float *dst, const float v[4]) { *(f32x4 *)dst = (f32x4){v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void store4(bf16_t *dst, const float v[4]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *(u32x2 *)dst = __builtin_bit_cast(u32x2, b);
}
__device__ __forceinline__ void load4(const float *src, float v[4]) {
    const f32x4 f = *(const f32x4 *)src;
    v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
}
__device__ __forceinline__ void load4(const bf16_t *src, float v[4]) {
    const u32x2 u = *(const u32x2 *)src;
    v[0] = bf2f((bf16_t)(u[0] & 0xffff)); v[1] = bf2f((bf16_t)(u[0] >> 16));
    v[2] = bf2f((bf16_t)(u[1] & 0xffff)); v[3] = bf2f((bf16_t)(u[1] >> 16));
}

// One K stage (128 bytes of K) of the 128 x 128 tile: A and W tiles share one LDS layout (128 rows x 128 B, 16-byte
// chunks XOR-swizzled by row & 7); waves 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles.  Used by the linear kernel
// (da_gemm_mfma.hip) and the implicit-GEMM group convolution (da_encoder.hip).
template <typename T, bool BFC = false>
__device__ __forceinline__ void mma_block(const unsigned char *sA, const unsigned char *sW, int wm, int wn, int lane,
                                          f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        u32x4 fa[4], fw[4];
        const int c = kk * 4 + (lane >> 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int Ra = wm * 64 + t * 16 + (lane & 15);
            const int Rw = wn * 64 + t * 16 + (lane & 15);
            fa[t] = *(const u32x4 *)(sA + Ra * 128 + ((c ^ (Ra & 7)) << 4));
            fw[t] = *(const u32x4 *)(sW + Rw * 128 + ((c ^ (Rw & 7)) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = Mma16<T, BFC>::run(fw[ni], fa[mi], acc[mi][ni]);    // a lane owns one row, 4 consecutive features
    }
}

struct GemmParams {
    int M, K, Nout;
    const void *A; int lda;
    const void *W; const float *bias;
    int ldw;            // row stride of W in elements (>= K): lets a column slice of a wider weight be used in place
    const void *pre;    // optional [M, Nout] (act dtype, row stride Nout) added BEFORE the activation: a loop-invariant
                        // part of the product computed once (the piece-feature columns of mlp.0)
    int act; const void *res; void *out; int ldo;
    // QKV scatter mode (dense attention layouts)
    int qkv; int HC, C, n_pad; const int32_t *row_map;
    unsigned Cmagic;   // ceil(2^32 / C): f / C == __umulhi(f, Cmagic) for the f < 2^16 that occur here
    int Cv; unsigned Cvmagic;   // Cv > 0: the V block holds heads of Cv channels (folded value heads) and there is no skip block
    void *Q, *Kb, *Vt, *S;
    int nct, nt;     // column tiles of this launch, column tiles per workgroup
    int xcd_groups;  // > 0: 1-D grid, XCD-aware (row tile, column group) mapping with this many column groups
    int ksplit, kchunk;   // ksplit > 1 (XCD-aware grid only): blockIdx.y = reduction split z over [z kchunk, (z + 1) kchunk); out = partial [ksplit][M][ldo]
    unsigned long long *prof;   // DA_GEMM_PROBE builds: per-workgroup cycle breakdown [total, wait, mma, epilogue]
    int debug;   // DA_GEMM_DEBUG bits: 1 = no global stores, 2 = no MFMA, 4 = no DMA (timing experiments only)
};

}  // namespace da
