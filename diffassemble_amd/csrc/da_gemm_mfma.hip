// MFMA linear kernel: out = act(A[M,K] @ W[Nout,K]^T + bias) (+ residual), fp32 accumulate.
//
// Replaces torch.nn.Linear on the path (mlp, the fused Q|K|V|skip projections of the four
// TransformerConv layers, the pose-head hidden layer).  One kernel template for both act dtypes:
//   bf16: v_mfma_f32_16x16x32_bf16  (one MFMA per 64-byte K chunk)
//   fp32: v_mfma_f32_16x16x4_f32    (exact fp32; four MFMAs per 64-byte K chunk, the K index is
//         permuted consistently for both operands so each lane still reads 16 contiguous bytes)
// Tile 128 (rows of A) x 128 (rows of W) x 128 BYTES of K; 4 waves as 2x2, each 64x64 = 4x4 MFMA
// tiles.  Both operands are K-contiguous, so A and W tiles share one LDS layout: 128 rows x 128 B,
// 16-byte chunks XOR-swizzled by (row & 7) -- conflict-free for the ds_write_b128 staging and for
// the ds_read_b128 fragment reads (lane groups of 16 hit 16 distinct 16-B slots of the 256-B bank
// row).  Global->LDS goes through registers: the loads of tile t+1 are issued before the MFMAs of
// tile t and land in LDS after them (one LDS buffer, two barriers per K tile).
//
// Epilogue.  MFMA is issued as (W fragment) x (A fragment), so a lane owns ONE row of `out` and FOUR
// consecutive output features: bias/activation are applied in registers and the store is one
// 8-byte (bf16) / 16-byte (fp32) access.  In QKV mode (dense block-diagonal attention) the fused
// projection is scattered straight into the layouts the attention kernel consumes:
//   Q, K  -> [H][n_pad][C]     head-major, rows at the graph's padded offset (row_map)
//   V     -> [H][C][n_pad]     TRANSPOSED (blocks of V columns issue (A) x (W) so a lane owns four
//                              consecutive nodes of one feature)
//   skip  -> [M][H*C]          row-major
#include "da_common.h"
#include "da_internal.h"

namespace da {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const u32x4 &x, const u32x4 &y, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    }
};
template <> struct Mma16<float> {
    static __device__ __forceinline__ f32x4 run(const u32x4 &x, const u32x4 &y, f32x4 c) {
        const f32x4 a = __builtin_bit_cast(f32x4, x), b = __builtin_bit_cast(f32x4, y);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

__device__ __forceinline__ void store4(float *dst, const float v[4]) {
    if ((((size_t)dst) & 15) == 0) *(float4 *)dst = make_float4(v[0], v[1], v[2], v[3]);
    else { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
}
__device__ __forceinline__ void store4(bf16_t *dst, const float v[4]) {
    if ((((size_t)dst) & 7) == 0) {
        uint2 u;
        u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *(uint2 *)dst = u;
    } else { dst[0] = f2bf(v[0]); dst[1] = f2bf(v[1]); dst[2] = f2bf(v[2]); dst[3] = f2bf(v[3]); }
}
__device__ __forceinline__ void load4(const float *src, float v[4]) {
    if ((((size_t)src) & 15) == 0) { const float4 f = *(const float4 *)src; v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
    else { v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3]; }
}
__device__ __forceinline__ void load4(const bf16_t *src, float v[4]) {
    if ((((size_t)src) & 7) == 0) {
        const uint2 u = *(const uint2 *)src;
        v[0] = bf2f((bf16_t)(u.x & 0xffff)); v[1] = bf2f((bf16_t)(u.x >> 16));
        v[2] = bf2f((bf16_t)(u.y & 0xffff)); v[3] = bf2f((bf16_t)(u.y >> 16));
    } else { v[0] = bf2f(src[0]); v[1] = bf2f(src[1]); v[2] = bf2f(src[2]); v[3] = bf2f(src[3]); }
}

struct GemmParams {
    int M, K, Nout;
    const void *A; int lda;
    const void *W; const float *bias;
    int act; const void *res; void *out; int ldo;
    // QKV scatter mode (dense attention layouts)
    int qkv; int HC, C, n_pad; const int32_t *row_map;
    void *Q, *Kb, *Vt, *S;
};

template <typename T, bool VORIENT>
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
                acc[mi][ni] = VORIENT ? Mma16<T>::run(fa[mi], fw[ni], acc[mi][ni])     // D[node][feature]
                                      : Mma16<T>::run(fw[ni], fa[mi], acc[mi][ni]);    // D[feature][node]
    }
}

template <typename T, bool VORIENT>
__global__ __launch_bounds__(256, 2) void k_gemm_mfma(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 128 * 128];
    unsigned char *sA = smem, *sW = smem + 128 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int row0 = blockIdx.y * 128;
    // QKV mode is issued as two launches: the V column blocks (VORIENT) and everything else
    int cb = blockIdx.x;
    if (p.qkv) {
        const int per = p.HC / 128;
        cb = VORIENT ? cb + 2 * per : (cb < 2 * per ? cb : cb + per);
    }
    const int col0 = cb * 128;
    constexpr int BK = 128 / (int)sizeof(T);
    constexpr bool vorient = VORIENT;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: thread owns 16-byte chunk c of rows r0 + 32 i (i = 0..3) of both tiles
    const int r0 = tid >> 3, c = tid & 7;
    const int loff = r0 * 128 + ((c ^ (r0 & 7)) << 4);            // + i * 4096 for row r0 + 32 i
    const char *Ab = (const char *)p.A + c * 16, *Wb = (const char *)p.W + c * 16;
    const size_t ldaB = (size_t)p.lda * sizeof(T), ldwB = (size_t)p.K * sizeof(T);
    int arow[4], wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        arow[i] = min(row0 + r0 + 32 * i, p.M - 1);
        wrow[i] = min(col0 + r0 + 32 * i, p.Nout - 1);
    }
    u32x4 ra[4], rw[4];
    const int nk = p.K / BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *(const u32x4 *)(Ab + arow[i] * ldaB);
        rw[i] = *(const u32x4 *)(Wb + wrow[i] * ldwB);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4 *)(sA + loff + i * 4096) = ra[i];
            *(u32x4 *)(sW + loff + i * 4096) = rw[i];
        }
        __syncthreads();
        if (kt + 1 < nk) {
            const size_t kb = (size_t)(kt + 1) * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *(const u32x4 *)(Ab + arow[i] * ldaB + kb);
                rw[i] = *(const u32x4 *)(Wb + wrow[i] * ldwB + kb);
            }
        }
        mma_block<T, VORIENT>(sA, sW, wm, wn, lane, acc);
    }

    // ------------------------------------------------------------------ epilogue
    if (!vorient) {
        const int which = p.qkv ? col0 / p.HC : 0;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = row0 + wm * 64 + mi * 16 + (lane & 15);
            if (m >= p.M) continue;
            const int prow = (p.qkv && which < 2) ? p.row_map[m] : 0;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int f0 = col0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
                if (f0 >= p.Nout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[mi][ni][r] + (p.bias ? p.bias[f0 + r] : 0.f), p.act);
                if (!p.qkv) {
                    if (p.res) {
                        float rr[4];
                        load4((const T *)p.res + (size_t)m * p.ldo + f0, rr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rr[r];
                    }
                    store4((T *)p.out + (size_t)m * p.ldo + f0, v);
                } else if (which == 3) {
                    store4((T *)p.S + (size_t)m * p.HC + (f0 - 3 * p.HC), v);
                } else {
                    const int f = f0 - which * p.HC, h = f / p.C, c = f - h * p.C;
                    T *dst = (T *)(which == 0 ? p.Q : p.Kb) + ((size_t)h * p.n_pad + prow) * p.C + c;
                    store4(dst, v);
                }
            }
        }
    } else {
        // V columns: lane owns feature f and four consecutive nodes -> transposed store
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int fcol = col0 + wn * 64 + ni * 16 + (lane & 15);
            if (fcol >= p.Nout) continue;
            const int f = fcol - 2 * p.HC, h = f / p.C, c = f - h * p.C;
            const float b = p.bias ? p.bias[fcol] : 0.f;
            T *vrow = (T *)p.Vt + ((size_t)h * p.C + c) * p.n_pad;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m0 = row0 + wm * 64 + mi * 16 + (lane >> 4) * 4;
                if (m0 >= p.M) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r] + b;
                const int p0 = p.row_map[m0];
                if (m0 + 3 < p.M && p.row_map[m0 + 3] == p0 + 3) {
                    store4(vrow + p0, v);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (m0 + r < p.M) stf(vrow + p.row_map[m0 + r], v[r]);
                }
            }
        }
    }
}

static bool aligned16(const void *p) { return (((size_t)p) & 15) == 0; }

// returns 0 = launched, -1 = shape not supported by this kernel (caller falls back), >0 error
int launch_gemm_mfma(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias,
                     int act, const void *res, void *out, int ldo, const QkvScatter *qs, hipStream_t st) {
    const int es = (int)esize(prec), BK = 128 / es;
    if (M <= 0 || Nout <= 0) return 0;
    if (K % BK != 0 || (Nout & 3) || !aligned16(A) || !aligned16(W) || ((size_t)lda * es) % 16 != 0) return -1;
    GemmParams p;
    p.M = M; p.K = K; p.Nout = Nout; p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.act = act; p.res = res;
    p.out = out; p.ldo = ldo; p.qkv = 0; p.HC = 1; p.C = 1; p.n_pad = 0; p.row_map = nullptr;
    p.Q = p.Kb = p.Vt = p.S = nullptr;
    if (qs) {
        if (qs->HC % 128 != 0 || (qs->C & 3) || Nout != 4 * qs->HC || act != DA_ACT_NONE || res) return -1;
        p.qkv = 1; p.HC = qs->HC; p.C = qs->C; p.n_pad = qs->n_pad; p.row_map = qs->row_map;
        p.Q = qs->Q; p.Kb = qs->K; p.Vt = qs->Vt; p.S = qs->S;
    } else if ((ldo & 3) != 0) {
        return -1;
    }
    dim3 grid((Nout + 127) / 128, (M + 127) / 128);
    if (!qs) {
        if (prec == DA_PREC_BF16) k_gemm_mfma<bf16_t, false><<<grid, 256, 0, st>>>(p);
        else k_gemm_mfma<float, false><<<grid, 256, 0, st>>>(p);
    } else {
        const int per = qs->HC / 128;
        dim3 g_qks(3 * per, grid.y), g_v(per, grid.y);
        if (prec == DA_PREC_BF16) {
            k_gemm_mfma<bf16_t, false><<<g_qks, 256, 0, st>>>(p);
            k_gemm_mfma<bf16_t, true><<<g_v, 256, 0, st>>>(p);
        } else {
            k_gemm_mfma<float, false><<<g_qks, 256, 0, st>>>(p);
            k_gemm_mfma<float, true><<<g_v, 256, 0, st>>>(p);
        }
    }
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
