// Training path of the denoiser: forward with saved activations + backward of every
// live parameter, fp32 throughout (the parity mode; the gradient fixtures of the reference are fp32).
//
// Replaces torch autograd through Eff_GAT.forward_with_feats (efficient_gat.py:121-146) and the PyG
// TransformerConv message/aggregate of Transformer_GNN.py:29-46 / exophormer_gnn.py:161-215.
//
// Attention backward works on the CSR graph for every graph type (complete puzzles, Exphander
// expanders, the exophormer virtual-node edges, multi-edges), with NO atomics: one kernel walks the
// incoming edges of each destination (dq, and the per-(node, head) term D = sum_e p_e dp_e), a second
// walks the outgoing edges of each source (dk, dv) -- the host supplies both CSR orientations.
//   p_e  = exp(a_e - m_i) / (sum + 1e-16)          (m_i, 1/(sum+1e-16) saved by the forward kernel)
//   dp_e = <dO_i, v_j>,  ds_e = p_e (dp_e - D_i)
//   dq_i = scale * sum_e ds_e k_j,  dk_j = scale * sum_e ds_e q_i,  dv_j = sum_e p_e dO_i
// Linear layers: dX = dY @ W runs through the forward MFMA linear kernel with W transposed once per
// step; dW = dY^T @ X is a dedicated fp32-MFMA kernel whose reduction runs over the node dimension,
// split over node chunks with a deterministic second-pass reduction.
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <vector>

#include "da_gemm_common.h"

namespace da {

// dst = gelu(src); dst16 (optional): the same values rounded to bf16 -- the next projection's operand in the q16 mode (its own
// k_cast_h launch folded in; the rounding is of the fp32 value just stored, i.e. the same bits)
__global__ __launch_bounds__(256) void k_gelu_fwd(size_t n, const float *__restrict__ src, float *__restrict__ dst, bf16_t *__restrict__ dst16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = gelu_erf(src[i]);
        dst[i] = v;
        if (dst16) dst16[i] = f2bf(v);
    }
}

// dx = dy * gelu'(pre)   (dx may alias dy)
__global__ __launch_bounds__(256) void k_gelu_bwd(size_t n, const float *__restrict__ pre, const float *dy, float *dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * gelu_grad(pre[i]);
}

// dst = bf16(src), 8 elements per thread and pass (n a multiple of 8; both 16-byte aligned)
__global__ __launch_bounds__(256) void k_cast_h(size_t n8, const float *__restrict__ src, bf16_t *__restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4 *)(src + 8 * i), b = *(const f32x4 *)(src + 8 * i + 4);
        typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_;
        const bf16x8_ v = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
        *(u32x4 *)(dst + 8 * i) = __builtin_bit_cast(u32x4, v);
    }
}
static int cast_h(size_t n, const float *src, bf16_t *dst, hipStream_t st) {
    const size_t n8 = n / 8, blocks = (n8 + 255) / 256;
    k_cast_h<<<(unsigned)(blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks)), 256, 0, st>>>(n8, src, dst);
    DA_LAUNCH_CHECK();
    return 0;
}
// dst[c][r] = src[r][c], rounded to bf16 (the W^T operand of the q16 mode's dX products)
__global__ __launch_bounds__(256) void k_transpose_h(int rows, int cols, const float *__restrict__ src, bf16_t *__restrict__ dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[(size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = f2bf(tile[tx][ty + 8 * k]);
    }
}
// dst[c][r] = src[r][c]
__global__ __launch_bounds__(256) void k_transpose(int rows, int cols, const float *__restrict__ src, float *__restrict__ dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[(size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][ty + 8 * k];
    }
}

// out[c] += sum_m A[m][c]      (bias gradients).  Two deterministic stages: grid (ceil(N/64), chunks)
// column sums over row chunks into `partial[chunk][N]`, then one pass adds the chunks into out.
__global__ __launch_bounds__(256) void k_colsum_partial(int M, int N, const float *__restrict__ A, int lda, int rows_per,
                                                        float *__restrict__ partial) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s = 0.f;
    if (c < N)
        for (int m = m0 + ty; m < m1; m += 4) s += A[(size_t)m * lda + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < N) partial[(size_t)blockIdx.y * N + c] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}
// 64 columns per block, four threads per column: thread (tx, ty) adds the chunks ty, ty + 4, ... in ascending order, the four
// sub-sums are combined as (s0 + s1) + (s2 + s3) -- a fixed order (data-parallel replicas must produce identical bits), a
// quarter of the dependent loads per thread (one thread per column walked all chunks: 15 - 18 us per call, 8 calls per step)
__global__ __launch_bounds__(256) void k_colsum_finish(int chunks, int N, const float *__restrict__ partial, float *out) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < N)
        for (int k = ty; k < chunks; k += 4) s += partial[(size_t)k * N + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < N) out[c] += (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// ---------------------------------------------------------------------------------------------
// dW kernel: C[n][k] (+)= sum_m A[m][n] * B[m][k]   (A = dY [M, N], B = X [M, K], both row-major:
// the reduction runs over ROWS, so both tiles are staged exactly as they lie in memory, [m][col], and
// the fp32 MFMA fragments (one float per lane: row = lane & 15 of the output tile, k = lane >> 4) are
// read with conflict-free 4-byte LDS reads -- no transposes anywhere).
// Tile 64 (n) x 64 (k), 16 rows of m per stage, 4 waves as 2 x 2, v_mfma_f32_16x16x4_f32.
// grid = (ceil(K/64), ceil(N/64), splits); split s reduces rows [s * Mc, (s + 1) * Mc).
__device__ __forceinline__ f32x4 load_row4(const float *base, int ld, int row, int col, int rows, int cols, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= rows) return v;
    const float *p = base + (size_t)row * ld + col;
    if (vec && col + 3 < cols) return *(const f32x4 *)p;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (col + r < cols) v[r] = p[r];
    return v;
}

// BFC: the fp32 tiles are rounded to bf16 in registers and multiplied with v_mfma_f32_16x16x16_bf16 (one MFMA per 16-row
// stage and tile pair instead of four exact-fp32 ones; training's DA_TRAIN_MMA_BF16 mode); lane group g = lane >> 4 takes the
// stage rows g, g + 4, g + 8, g + 12 (any assignment works as long as both operands use it: rows one apart keep the four
// groups on different banks)
template <bool BFC>
__global__ __launch_bounds__(256) void k_gemm_tn(int M, int N, int K, const float *__restrict__ A, int lda,
                                                 const float *__restrict__ B, int ldb, float *C, int ldc,
                                                 float *partial, int Mc) {
    constexpr int LS = 80;                                       // LDS row stride (floats): 4 rows -> banks 0,16,32,48
    __shared__ __attribute__((aligned(16))) float As[16 * LS];
    __shared__ __attribute__((aligned(16))) float Bs[16 * LS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int m_beg = blockIdx.z * Mc, m_end = min(M, m_beg + Mc);
    const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
    const bool vecA = (lda % 4 == 0) && (((size_t)A & 15) == 0), vecB = (ldb % 4 == 0) && (((size_t)B & 15) == 0);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra = load_row4(A, lda, m_beg + lrow, n0 + lc4, m_end, N, vecA);
    f32x4 rb = load_row4(B, ldb, m_beg + lrow, k0 + lc4, m_end, K, vecB);
    for (int m0 = m_beg; m0 < m_end; m0 += 16) {
        *(f32x4 *)(As + lrow * LS + lc4) = ra;
        *(f32x4 *)(Bs + lrow * LS + lc4) = rb;
        __syncthreads();
        if (m0 + 16 < m_end) {
            ra = load_row4(A, lda, m0 + 16 + lrow, n0 + lc4, m_end, N, vecA);
            rb = load_row4(B, ldb, m0 + 16 + lrow, k0 + lc4, m_end, K, vecB);
        }
        if (BFC) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
            typedef __attribute__((ext_vector_type(4))) short s16x4_;
            s16x4_ a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x4_ ta, tb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ta[e] = (__bf16)As[(4 * e + (lane >> 4)) * LS + wr * 32 + i * 16 + (lane & 15)];
                    tb[e] = (__bf16)Bs[(4 * e + (lane >> 4)) * LS + wc * 32 + i * 16 + (lane & 15)];
                }
                a[i] = __builtin_bit_cast(s16x4_, ta);
                b[i] = __builtin_bit_cast(s16x4_, tb);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[(kk + (lane >> 4)) * LS + wr * 32 + i * 16 + (lane & 15)];
                b[i] = Bs[(kk + (lane >> 4)) * LS + wc * 32 + i * 16 + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        }
        __syncthreads();
    }
    float *dst = partial ? partial + (size_t)blockIdx.z * N * K : C;
    const int ldd = partial ? K : ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wr * 32 + i * 16 + 4 * (lane >> 4) + r, k = k0 + wc * 32 + j * 16 + (lane & 15);
                if (n < N && k < K) {
                    if (partial) dst[(size_t)n * ldd + k] = acc[i][j][r];
                    else dst[(size_t)n * ldd + k] += acc[i][j][r];
                }
            }
}

__global__ __launch_bounds__(256) void k_reduce_partial(int splits, int N, int K, const float *__restrict__ partial,
                                                        float *C, int ldc) {
    const size_t NK = (size_t)N * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < NK; i += (size_t)gridDim.x * blockDim.x) {
        // fixed summation order (deterministic), eight independent loads in flight: the split count reaches several
        // hundred for the encoder's weight gradients and one dependent load per iteration was latency-bound
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= splits; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(k + u) * NK + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < splits; ++k) s += partial[(size_t)k * NK + i];
        const size_t n = i / K, kk = i - n * K;
        C[n * ldc + kk] += s;
    }
}

// k_reduce_partial and k_colsum_finish in one launch (launch_gemm_tn_db): items [0, N K) add the split partials into C (when
// partial != null), items [N K, N K + N) add the `splits` row-range sums of a dY column into db -- in k_colsum_finish's order
// ((s0 + s1) + (s2 + s3), s_q = chunks q, q + 4, ... ascending), so the bits are the ones the two-launch form produced.
__global__ __launch_bounds__(256) void k_tn_finish(int splits, int N, int K, const float *__restrict__ partial, float *C, int ldc,
                                                   const float *__restrict__ bpartial, float *db) {
    const size_t NK = partial ? (size_t)N * K : 0, items = NK + (bpartial ? (size_t)N : 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t)gridDim.x * blockDim.x) {
        if (i < NK) {
            const size_t NKs = (size_t)N * K;
            float s = 0.f;
            int k = 0;
            for (; k + 32 <= splits; k += 32) {               // (small outputs are cut into up to 144 row ranges: 32 loads in flight)
                float v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = partial[(size_t)(k + u) * NKs + i];
#pragma unroll
                for (int u = 0; u < 32; ++u) s += v[u];
            }
            for (; k + 8 <= splits; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(k + u) * NKs + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < splits; ++k) s += partial[(size_t)k * NKs + i];
            const size_t n = i / K, kk = i - n * K;
            C[n * ldc + kk] += s;
        } else {
            const size_t c = i - NK;
            float q[4] = {0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 32 <= splits; k += 32) {
                float v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = bpartial[(size_t)(k + u) * N + c];
#pragma unroll
                for (int u = 0; u < 32; ++u) q[u & 3] += v[u];
            }
            for (; k < splits; ++k) q[k & 3] += bpartial[(size_t)k * N + c];
            db[c] += (q[0] + q[1]) + (q[2] + q[3]);
        }
    }
}

constexpr size_t PART_CAP = (size_t)16 << 20;        // floats of split-reduction scratch (64 MB)

int launch_gemm_tn(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                          float *partial, hipStream_t st, bool bfc) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int tn = (N + 63) / 64, tk = (K + 63) / 64;
    long splits = 2048 / ((long)tn * tk);
    const long by_rows = (M + 255) / 256, by_cap = (long)(PART_CAP / ((size_t)N * K));
    splits = splits > by_rows ? by_rows : splits;
    splits = splits > by_cap ? by_cap : splits;
    if (splits < 1) splits = 1;
    int Mc = (int)(((M + splits - 1) / splits + 15) / 16 * 16);
    splits = (M + Mc - 1) / Mc;
    const dim3 grid((unsigned)tk, (unsigned)tn, (unsigned)splits);
    if (splits == 1) {
        if (bfc) k_gemm_tn<true><<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, nullptr, Mc);
        else k_gemm_tn<false><<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, nullptr, Mc);
    } else {
        if (bfc) k_gemm_tn<true><<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, partial, Mc);
        else k_gemm_tn<false><<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, partial, Mc);
        const size_t NK = (size_t)N * K;
        k_reduce_partial<<<(unsigned)((NK + 255) / 256 > 4096 ? 4096 : (NK + 255) / 256), 256, 0, st>>>((int)splits, N, K, partial, C, ldc);
    }
    DA_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// dW + db in ONE pass for the bf16-operand training mode (round 4): C[n][k] += sum_m A[m][n] B[m][k] and, if db,
// db[n] += sum_m A[m][n] (A = dY, B = X, fp32 row-major).  k_gemm_tn<true> above is bound by what its 64 x 64 tiles pull through
// the L2 (every dY column block is re-read K / 64 times, every X column block N / 64 times: 1.36 GB per call at BASELINE
// configuration 5's conv 3) and the bias gradient read dY once more in its own two kernels.  Here: 128 x 128 tiles (half the
// operand traffic), 64 rows of m per stage (one stage of register prefetch: the bytes in flight per workgroup are what hides the load latency at two to three workgroups per CU), the fp32 tiles rounded to bf16 on their way into LDS (row-major [m][col], 8-byte
// stores) and fetched as MFMA fragments by ds_read_b64_tr_b16 (one read per 16 x 16 operand block instead of four scalar reads +
// four conversions); the workgroups of the first k tile also add up their dY columns in fp32 (exact values, fixed order) while
// they stage them.  grid = (ceil(K/128), ceil(N/128), splits).
typedef __attribute__((ext_vector_type(4))) short tn_s16x4;
// 4 consecutive bf16 of a row-major matrix as fp32, zero beyond (rows, cols)
__device__ __forceinline__ f32x4 load_row4_h(const bf16_t *base, int ld, int row, int col, int rows, int cols, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= rows) return v;
    const bf16_t *p = base + (size_t)row * ld + col;
    if (vec && col + 3 < cols) {
        const u32x2 u = *(const u32x2 *)p;
        return (f32x4){bf2f((bf16_t)(u[0] & 0xffff)), bf2f((bf16_t)(u[0] >> 16)), bf2f((bf16_t)(u[1] & 0xffff)), bf2f((bf16_t)(u[1] >> 16))};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (col + r < cols) v[r] = bf2f(p[r]);
    return v;
}
// A16: A (dY) is stored as bf16 (the q16 mode's projection gradient); B16: B (the layer's input X) comes as the bf16 image the forward
// kept of it (q16 mode: the projection's own operand) -- 8 instead of 16 bytes per piece on the kernel's dominant stream (a 128 x 128
// tile pulls 32 KB of fp32 X per 64-row stage against 16 KB of bf16 dY; the kernel moves ~0.5 GB through the L2 per call)
template <bool A16, bool B16 = false>
__global__ __launch_bounds__(256) void k_gemm_tn_db(int M, int N, int K, const void *__restrict__ Av, int lda,
                                                    const void *__restrict__ Bv, int ldb, float *C, int ldc,
                                                    float *partial, float *bpartial, int Mc, int xcd_tk) {
    constexpr int PT = 136;                                      // LDS row pitch (bf16 elements): 272 B
    constexpr int SR = 64, NU = SR / 8;                          // rows of m per stage; 8-byte / 16-byte pieces per thread, operand and stage
    __shared__ __attribute__((aligned(16))) unsigned short As[SR * PT];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[SR * PT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, l15 = lane & 15, lg = lane >> 4;
    // xcd_tk > 0: 1-D grid with the row split = id % splits (splits a multiple of 8: workgroups are dealt round-robin over the
    // 8 XCDs, so ALL tiles of one row range run on one XCD, start together and walk the rows at the same pace -- the operand
    // rows one of them pulls in are L2 hits for the others; with the 3-D grid the tiles of a row range were spread over the XCDs
    // and every one of them fetched its operands from the fabric)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd_tk > 0) {
        const int splits = (M + Mc - 1) / Mc, tile = blockIdx.x / splits;
        bz = blockIdx.x - tile * splits;
        by = tile / xcd_tk;
        bx = tile - by * xcd_tk;
    }
    const int n0 = by * 128, k0 = bx * 128;
    const int m_beg = bz * Mc, m_end = min(M, m_beg + Mc);
    const int lrow = tid >> 5, lc4 = (tid & 31) * 4;             // stage rows lrow + 8 u, columns lc4 ..+3
    const bool vecA = (lda % 4 == 0) && (((size_t)Av & 15) == 0), vecB = (ldb % 4 == 0) && (((size_t)Bv & 15) == 0);
    auto pack = [](const f32x4 &v) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
        const bf16x4_ b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        return __builtin_bit_cast(tn_s16x4, b);
    };
    // A pieces stay in the form they are stored in (bf16: the bits go to LDS untouched; fp32: rounded when they are written there)
    typedef typename std::conditional<A16, tn_s16x4, f32x4>::type RA;
    auto loadA = [&](int row) -> RA {
        if constexpr (A16) {
            const bf16_t *q = (const bf16_t *)Av + (size_t)row * lda + n0 + lc4;
            if (row < m_end && vecA && n0 + lc4 + 3 < N) return *(const tn_s16x4 *)q;
            return pack(load_row4_h((const bf16_t *)Av, lda, row, n0 + lc4, m_end, N, false));
        } else {
            return load_row4((const float *)Av, lda, row, n0 + lc4, m_end, N, vecA);
        }
    };
    typedef typename std::conditional<B16, tn_s16x4, f32x4>::type RB;
    auto loadB = [&](int row) -> RB {
        if constexpr (B16) {
            const bf16_t *q = (const bf16_t *)Bv + (size_t)row * ldb + k0 + lc4;
            if (row < m_end && vecB && k0 + lc4 + 3 < K) return *(const tn_s16x4 *)q;
            return pack(load_row4_h((const bf16_t *)Bv, ldb, row, k0 + lc4, m_end, K, false));
        } else {
            return load_row4((const float *)Bv, ldb, row, k0 + lc4, m_end, K, vecB);
        }
    };
    const bool want_db = bpartial != nullptr && bx == 0;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    RA ra[NU];
    RB rb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        ra[u] = loadA(m_beg + lrow + 8 * u);
        rb[u] = loadB(m_beg + lrow + 8 * u);
    }
    for (int m0 = m_beg; m0 < m_end; m0 += SR) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if constexpr (A16) {
                *(tn_s16x4 *)(As + (lrow + 8 * u) * PT + lc4) = ra[u];
                if (want_db) {
                    const unsigned lo = (unsigned)(unsigned short)ra[u][0] | ((unsigned)(unsigned short)ra[u][1] << 16);
                    const unsigned hi = (unsigned)(unsigned short)ra[u][2] | ((unsigned)(unsigned short)ra[u][3] << 16);
                    csum += (f32x4){__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                                    __uint_as_float(hi & 0xffff0000u)};
                }
            } else {
                *(tn_s16x4 *)(As + (lrow + 8 * u) * PT + lc4) = pack(ra[u]);
                csum += ra[u];
            }
            if constexpr (B16) *(tn_s16x4 *)(Bs + (lrow + 8 * u) * PT + lc4) = rb[u];
            else *(tn_s16x4 *)(Bs + (lrow + 8 * u) * PT + lc4) = pack(rb[u]);
        }
        __syncthreads();
        if (m0 + SR < m_end) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                ra[u] = loadA(m0 + SR + lrow + 8 * u);
                rb[u] = loadB(m0 + SR + lrow + 8 * u);
            }
        }
#pragma unroll
        for (int ks = 0; ks < SR / 16; ++ks) {
            tn_s16x4 a[4], b[4];
            // fragment (k = rows 16 ks + 4 lg ..+3, n = column c0 + l15): lane i' of a 16-lane group supplies the address of
            // row (i' >> 2), columns 4 (i' & 3) ..+3 and receives column i' of the four rows
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4 *)(
                    As + (16 * ks + 4 * lg + (l15 >> 2)) * PT + wr * 64 + i * 16 + 4 * (l15 & 3)));
                b[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4 *)(
                    Bs + (16 * ks + 4 * lg + (l15 >> 2)) * PT + wc * 64 + i * 16 + 4 * (l15 & 3)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float *dst = partial ? partial + (size_t)bz * N * K : C;
    const int ldd = partial ? K : ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wr * 64 + i * 16 + 4 * lg + r, k = k0 + wc * 64 + j * 16 + l15;
                if (n < N && k < K) {
                    if (partial) dst[(size_t)n * ldd + k] = acc[i][j][r];
                    else dst[(size_t)n * ldd + k] += acc[i][j][r];
                }
            }
    if (want_db) {                                              // the eight row groups' sums of each column, fixed order
        float *red = (float *)As;                               // [8][128] floats = 4 KB of the 17 KB tile
        *(f32x4 *)(red + lrow * 128 + lc4) = csum;
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += red[g * 128 + tid];
            bpartial[(size_t)bz * N + n0 + tid] = s;
        }
    }
}

// dW (+ db) of the bf16-operand mode.  bscratch: splits * N floats
int launch_gemm_tn_db(int M, int N, int K, const void *A, int lda, const void *B, int ldb, float *C, int ldc,
                      float *partial, float *db, float *bscratch, hipStream_t st, bool a16, bool b16) {
    DA_REQUIRE(!b16 || a16, "launch_gemm_tn_db: a bf16 X image is only instantiated beside a bf16 dY");
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int tn = (N + 127) / 128, tk = (K + 127) / 128;
    long splits = 640 / ((long)tn * tk);
    // (one- and two-tile outputs -- the head's, pos_mlp's, mlp.2's gradients: row ranges of ONE 64-row stage, so that the launch is
    //  one load round trip deep instead of four: 14 -> ~5 us each at 9 216 rows)
    const long by_rows = tn * tk <= 2 ? (M + 63) / 64 : (M + 255) / 256, by_cap = (long)(PART_CAP / ((size_t)N * K));
    splits = splits > by_rows ? by_rows : splits;
    splits = splits > by_cap ? by_cap : splits;
    if (splits < 1) splits = 1;
    static int xcd_off = -1;
    if (xcd_off < 0) xcd_off = DA_XENV("DA_TN_NO_XCD_MAP", 0) ? 1 : 0;
    // several tiles and enough rows: exactly 8 (or 16) row ranges, one (two) per XCD (see the kernel)
    const bool xcd = !xcd_off && tn * tk >= 8 && by_rows >= 8 && by_cap >= 8;
    if (xcd) splits = (tn * tk <= 12 && by_rows >= 32 && by_cap >= 32) ? 32 : (tn * tk <= 24 && by_rows >= 16 && by_cap >= 16) ? 16 : 8;
    int Mc = (int)(((M + splits - 1) / splits + 7) / 8 * 8);      // (a last partial stage is zero-filled: no multiple of the stage needed)
    const long want = splits;
    splits = (M + Mc - 1) / Mc;
    const bool xmap = xcd && splits == want;
    const dim3 grid = xmap ? dim3((unsigned)(tk * tn * splits)) : dim3((unsigned)tk, (unsigned)tn, (unsigned)splits);
    if (a16 && b16) k_gemm_tn_db<true, true><<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, splits == 1 ? nullptr : partial, db ? bscratch : nullptr, Mc,
                                                                xmap ? tk : 0);
    else if (a16) k_gemm_tn_db<true><<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, splits == 1 ? nullptr : partial, db ? bscratch : nullptr, Mc,
                                                   xmap ? tk : 0);
    else k_gemm_tn_db<false><<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, splits == 1 ? nullptr : partial, db ? bscratch : nullptr, Mc,
                                                   xmap ? tk : 0);
    {       // one finishing launch: the split partials into dW and the row-range sums into db (two launches until round 5)
        const size_t NK = splits > 1 ? (size_t)N * K : 0, items = NK + (db ? (size_t)N : 0);
        if (items) k_tn_finish<<<(unsigned)((items + 255) / 256 > 4096 ? 4096 : (items + 255) / 256), 256, 0, st>>>((int)splits, N, K, NK ? partial : nullptr, C, ldc, db ? bscratch : nullptr, db);
    }
    DA_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// bf16 variant of the dW kernel (the encoder's bf16 training mode): C[n][k] (+)= sum_m A[m][n] B[m][k], A / B bf16
// row-major, fp32 accumulation and output.  v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE reduction indices per lane,
// and the reduction index m is the slow one in memory, so the tiles are transposed on their way into LDS: a thread loads
// 16 bytes (8 columns of one row) and writes them as eight 2-byte stores into the [column][m] image (row pitch 72 bytes:
// the two 8-byte fragment reads of a lane are conflict-free across the 32-lane groups, the 2-byte writes of the two
// column groups of a wave land on disjoint banks).  Tile 128 (n) x 128 (k), 32 rows of m per stage, 4 waves as 2 x 2.
// Arbitrary N, K (zero-filled loads, masked stores); lda / ldb multiples of 8, 16-byte aligned bases.
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void k_gemm_tn_bf16(int M, int N, int K, const bf16_t *__restrict__ A, int lda,
                                                      const bf16_t *__restrict__ B, int ldb, float *C, int ldc,
                                                      float *partial, int Mc) {
    constexpr int PITCH = 72;
    __shared__ __attribute__((aligned(16))) unsigned char As[128 * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[128 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int n0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
    const int m_beg = blockIdx.z * Mc, m_end = min(M, m_beg + Mc);
    const int srow = tid & 31, sch = tid >> 5;                   // staging role: row of the stage, 8-column chunk (and + 8)
    auto ldchunk = [&](const bf16_t *P, int ld, int m, int col, int cols) -> u32x4 {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m >= m_end || col >= cols) return v;
        const bf16_t *q = P + (size_t)m * ld + col;
        if (col + 8 <= cols) return *(const u32x4 *)q;
        unsigned short e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8 && col + i < cols; ++i) e[i] = q[i];
        v[0] = e[0] | ((unsigned)e[1] << 16); v[1] = e[2] | ((unsigned)e[3] << 16);
        v[2] = e[4] | ((unsigned)e[5] << 16); v[3] = e[6] | ((unsigned)e[7] << 16);
        return v;
    };
    auto put = [&](unsigned char *S, int chunk, const u32x4 &v) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *(unsigned short *)(S + (8 * chunk + e) * PITCH + srow * 2) = (unsigned short)(v[e >> 1] >> ((e & 1) * 16));
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 ra[2], rb[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        ra[p] = ldchunk(A, lda, m_beg + srow, n0 + 8 * (sch + 8 * p), N);
        rb[p] = ldchunk(B, ldb, m_beg + srow, k0 + 8 * (sch + 8 * p), K);
    }
    for (int m0 = m_beg; m0 < m_end; m0 += 32) {
#pragma unroll
        for (int p = 0; p < 2; ++p) { put(As, sch + 8 * p, ra[p]); put(Bs, sch + 8 * p, rb[p]); }
        __syncthreads();
        if (m0 + 32 < m_end) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                ra[p] = ldchunk(A, lda, m0 + 32 + srow, n0 + 8 * (sch + 8 * p), N);
                rb[p] = ldchunk(B, ldb, m0 + 32 + srow, k0 + 8 * (sch + 8 * p), K);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned char *pa = As + (wr * 64 + i * 32 + (lane & 31)) * PITCH + ks * 32 + (lane >> 5) * 16;
                const unsigned char *pb = Bs + (wc * 64 + i * 32 + (lane & 31)) * PITCH + ks * 32 + (lane >> 5) * 16;
                const uint2 a0 = *(const uint2 *)pa, a1 = *(const uint2 *)(pa + 8), b0 = *(const uint2 *)pb, b1 = *(const uint2 *)(pb + 8);
                a[i] = (u32x4){a0.x, a0.y, a1.x, a1.y};
                b[i] = (u32x4){b0.x, b0.y, b1.x, b1.y};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                                       acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float *dst = partial ? partial + (size_t)blockIdx.z * N * K : C;
    const int ldd = partial ? K : ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), k = k0 + wc * 64 + j * 32 + (lane & 31);
                if (n < N && k < K) {
                    if (partial) dst[(size_t)n * ldd + k] = acc[i][j][r];
                    else dst[(size_t)n * ldd + k] += acc[i][j][r];
                }
            }
}

int launch_gemm_tn_bf16(int M, int N, int K, const bf16_t *A, int lda, const bf16_t *B, int ldb, float *C, int ldc,
                        float *partial, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int tn = (N + 127) / 128, tk = (K + 127) / 128;
    long splits = 2048 / ((long)tn * tk);
    const long by_rows = (M + 511) / 512, by_cap = (long)(PART_CAP / ((size_t)N * K));
    splits = splits > by_rows ? by_rows : splits;
    splits = splits > by_cap ? by_cap : splits;
    if (splits < 1) splits = 1;
    int Mc = (int)(((M + splits - 1) / splits + 31) / 32 * 32);
    splits = (M + Mc - 1) / Mc;
    const dim3 grid((unsigned)tk, (unsigned)tn, (unsigned)splits);
    if (splits == 1) {
        k_gemm_tn_bf16<<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, nullptr, Mc);
    } else {
        k_gemm_tn_bf16<<<grid, 256, 0, st>>>(M, N, K, A, lda, B, ldb, C, ldc, partial, Mc);
        const size_t NK = (size_t)N * K;
        k_reduce_partial<<<(unsigned)((NK + 255) / 256 > 4096 ? 4096 : (NK + 255) / 256), 256, 0, st>>>((int)splits, N, K, partial, C, ldc);
    }
    DA_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Attention backward, destination side: wave per destination node i, lane l holds the EPL = C / 8
// contiguous channels [l * EPL, (l + 1) * EPL) of the H*C-wide rows (8 lanes per head), exactly the
// decomposition of the forward kernel (da_attn_csr.hip).  Writes dq_i and the skip gradient into
// the fused [n, 4 HC] gradient of the projection, and D_i per head.
template <int EPL>
__global__ __launch_bounds__(256) void k_attn_bwd_dst(int n_nodes, const int32_t *__restrict__ row_ptr,
                                                      const int32_t *__restrict__ col_src, int H, int HC,
                                                      const float *__restrict__ qkvs, const float *__restrict__ d_o,
                                                      const float *__restrict__ stats, float *__restrict__ dY4,
                                                      float *__restrict__ Dd, float scale) {
    const int lane = threadIdx.x & 63;
    const int i = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (i >= n_nodes) return;
    const size_t ld = (size_t)4 * HC;
    const int off = lane * EPL, head = lane >> 3;
    float q[EPL], g[EPL], a1[EPL], a2[EPL];
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        q[x] = qkvs[(size_t)i * ld + off + x] * scale;
        g[x] = d_o[(size_t)i * HC + off + x];
        a1[x] = a2[x] = 0.f;
    }
    const float m = stats[((size_t)i * H + head) * 2], inv = stats[((size_t)i * H + head) * 2 + 1];
    float D = 0.f;
    const int beg = row_ptr[i], end = row_ptr[i + 1];
    for (int e = beg; e < end; ++e) {
        const int j = col_src[e];
        const float *kp = qkvs + (size_t)j * ld + HC + off;
        const float *vp = kp + HC;
        float kk[EPL], vv[EPL];
#pragma unroll
        for (int x = 0; x < EPL; ++x) { kk[x] = kp[x]; vv[x] = vp[x]; }
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int x = 0; x < EPL; ++x) { s = fmaf(q[x], kk[x], s); dp = fmaf(g[x], vv[x], dp); }
        s += __shfl_xor(s, 1); dp += __shfl_xor(dp, 1);
        s += __shfl_xor(s, 2); dp += __shfl_xor(dp, 2);
        s += __shfl_xor(s, 4); dp += __shfl_xor(dp, 4);
        const float p = expf(s - m) * inv;
        const float pd = p * dp;
        D += pd;
#pragma unroll
        for (int x = 0; x < EPL; ++x) { a1[x] = fmaf(pd, kk[x], a1[x]); a2[x] = fmaf(p, kk[x], a2[x]); }
    }
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        dY4[(size_t)i * ld + off + x] = (a1[x] - D * a2[x]) * scale;
        dY4[(size_t)i * ld + 3 * (size_t)HC + off + x] = g[x];
    }
    if ((lane & 7) == 0) Dd[(size_t)i * H + head] = D;
}

// Source side: wave per source node j over its OUTGOING edges (out_ptr / out_dst = CSR by source).
template <int EPL>
__global__ __launch_bounds__(256) void k_attn_bwd_src(int n_nodes, const int32_t *__restrict__ out_ptr,
                                                      const int32_t *__restrict__ out_dst, int H, int HC,
                                                      const float *__restrict__ qkvs, const float *__restrict__ d_o,
                                                      const float *__restrict__ stats, const float *__restrict__ Dd,
                                                      float *__restrict__ dY4, float scale) {
    const int lane = threadIdx.x & 63;
    const int j = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (j >= n_nodes) return;
    const size_t ld = (size_t)4 * HC;
    const int off = lane * EPL, head = lane >> 3;
    float kk[EPL], vv[EPL], dk[EPL], dv[EPL];
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        kk[x] = qkvs[(size_t)j * ld + HC + off + x];
        vv[x] = qkvs[(size_t)j * ld + 2 * (size_t)HC + off + x];
        dk[x] = dv[x] = 0.f;
    }
    const int beg = out_ptr[j], end = out_ptr[j + 1];
    for (int e = beg; e < end; ++e) {
        const int i = out_dst[e];
        float q[EPL], g[EPL];
#pragma unroll
        for (int x = 0; x < EPL; ++x) { q[x] = qkvs[(size_t)i * ld + off + x] * scale; g[x] = d_o[(size_t)i * HC + off + x]; }
        const float m = stats[((size_t)i * H + head) * 2], inv = stats[((size_t)i * H + head) * 2 + 1];
        const float D = Dd[(size_t)i * H + head];
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int x = 0; x < EPL; ++x) { s = fmaf(q[x], kk[x], s); dp = fmaf(g[x], vv[x], dp); }
        s += __shfl_xor(s, 1); dp += __shfl_xor(dp, 1);
        s += __shfl_xor(s, 2); dp += __shfl_xor(dp, 2);
        s += __shfl_xor(s, 4); dp += __shfl_xor(dp, 4);
        const float p = expf(s - m) * inv;
        const float ds = p * (dp - D);
#pragma unroll
        for (int x = 0; x < EPL; ++x) { dk[x] = fmaf(ds, q[x], dk[x]); dv[x] = fmaf(p, g[x], dv[x]); }
    }
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        dY4[(size_t)j * ld + HC + off + x] = dk[x];
        dY4[(size_t)j * ld + 2 * (size_t)HC + off + x] = dv[x];
    }
}

static int launch_attn_bwd(const da_graph *g, int H, int C, const float *qkvs, const float *d_o, const float *stats,
                           float *Dd, float *dY4, hipStream_t st) {
    const int n = g->n_nodes, HC = H * C;
    const float scale = 1.0f / sqrtf((float)C);
    const int grid = (int)(((size_t)n * 64 + 255) / 256);
#define DA_BWD_CASE(E)                                                                                              \
    case E:                                                                                                         \
        k_attn_bwd_dst<E><<<grid, 256, 0, st>>>(n, g->row_ptr, g->col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale); \
        k_attn_bwd_src<E><<<grid, 256, 0, st>>>(n, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale); \
        break;
    switch (C / 8) {
        DA_BWD_CASE(1) DA_BWD_CASE(2) DA_BWD_CASE(4) DA_BWD_CASE(8) DA_BWD_CASE(13) DA_BWD_CASE(16) DA_BWD_CASE(18)
        default:
            set_error("attention backward: unsupported head width C=%d", C);
            return 1;
    }
#undef DA_BWD_CASE
    DA_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Small pieces around the embedding
// a[r][o] = b0[o] + sum_k W0[o][k] x[r][k]  (pos_mlp.0 pre-activation), p1 = gelu(a)
__global__ __launch_bounds__(256) void k_pos_hidden(int n, int c_in, const float *__restrict__ x, const float *__restrict__ w0,
                                                    const float *__restrict__ b0, float *__restrict__ a, float *__restrict__ p1) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 16) return;
    const int r = idx >> 4, o = idx & 15;
    float v = b0[o];
    for (int k = 0; k < c_in; ++k) v += w0[o * c_in + k] * x[(size_t)r * c_in + k];
    a[idx] = v;
    p1[idx] = gelu_erf(v);
}

// time_emb.grad[t[r]][c] += dcomb[r][F + 32 + c]
__global__ __launch_bounds__(256) void k_time_scatter(int n, int D, int F, int steps, const int64_t *__restrict__ t,
                                                      const float *__restrict__ dcomb, float *grad) {
    // a thread takes one channel of 16 consecutive rows and adds them up while the timestep stays the same (the nodes of a
    // puzzle share theirs): one atomic per run instead of one per row -- 144 rows hammering one address took 21 us
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, chunk = idx >> 5, c = idx & 31;
    const int r0 = chunk * 16, r1 = min(n, r0 + 16);
    if (r0 >= n) return;
    int64_t cur = -1;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
        int64_t ti = t[r];
        ti = ti < 0 ? 0 : (ti >= steps ? steps - 1 : ti);
        if (ti != cur) {
            if (cur >= 0) atomicAdd(grad + cur * 32 + c, s);
            cur = ti;
            s = 0.f;
        }
        s += dcomb[(size_t)r * D + F + 32 + c];
    }
    if (cur >= 0) atomicAdd(grad + cur * 32 + c, s);
}

__global__ __launch_bounds__(256) void k_copy_cols(int n, int cols, const float *__restrict__ src, int lds, float *__restrict__ dst) {
    const size_t total = (size_t)n * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cols, c = i - r * cols;
        dst[i] = src[r * lds + c];
    }
}

// virt_node_embedding.grad[v][c] += sum_g dh[(g * V + v)][c]   (rows appended as arange(V).repeat(G))
__global__ __launch_bounds__(256) void k_virt_grad(int rows, int V, int D, const float *__restrict__ dh, float *grad) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= V * D) return;
    const int v = idx / D, c = idx - v * D;
    float s = 0.f;
    for (int r = v; r < rows; r += V) s += dh[(size_t)r * D + c];
    grad[idx] += s;
}

// da_train_dense.hip: complete graphs on the matrix cores (grouped GEMMs, attention matrix kept)
size_t dense_pair_floats(const da_graph *g, int H);
int dense_train_prepare(const da_graph *g, int H, long long *poff, int32_t *node_graph, hipStream_t st);
int dense_train_attn_fwd(const da_graph *g, int H, int C, const float *qkvs, const float *res, float *o, float *P,
                         const long long *poff, const int32_t *node_graph, hipStream_t st, bool bfc, bool q16);
int dense_train_attn_bwd(const da_graph *g, int H, int C, const float *qkvs, const float *d_o, const float *P, float *dP,
                         float *dY4, const long long *poff, const int32_t *node_graph, hipStream_t st, bool bfc, bool q16);
bool attn_small_ok(const da_graph *g, int C, bool bfc);      // the one-workgroup-per-group attention kernels take this layer
// hybrid graphs (adjacency-masked grouped GEMMs over the regular edges + CSR remainder, one softmax over both)
int hybrid_train_attn_fwd(const da_graph *g, int H, int C, const float *qkvs, const float *res, float *o, float *P, float *stats,
                          const long long *poff, const int32_t *node_graph, hipStream_t st, bool bfc);
int hybrid_train_attn_bwd(const da_graph *g, int H, int C, const float *qkvs, const float *d_o, const float *P, float *dP,
                          const float *stats, float *Dd, float *dY4, const long long *poff, const int32_t *node_graph, hipStream_t st, bool bfc,
                          const float *o, const float *res);
bool hybrid_flash_ok(const da_graph *g, int C, bool bfc);     // the flash-style kernels (no pair matrix) take this hybrid layer

static bool train_dense_disabled() { return cfg().train_attn == 0; }          // da_config.train_attn (DA_TRAIN_ATTN=0): the edge-list kernels only

static unsigned grid_for(size_t n) { const size_t b = (n + 255) / 256; return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// Every weight image the step needs beside the fp32 parameters, in ONE launch at the top of the forward (round 5; they were 13
// launches of k_cast_h / k_transpose_h / k_transpose spread over the forward and the backward, ~5 us of dependent-launch time
// each): job j of the table = one matrix, mode 0: dst[c][r] = src[r][c] (fp32 W^T for a dX product), 1: the same rounded to bf16
// (q16 mode), 2: dst[r][c] = bf16(src[r][c]) (the bf16 forward operand of the q16 mode).  grid = (tiles of the largest job, jobs).
struct WPrepJob { const float *src; void *dst; int rows, cols, mode, pad; };
struct WPrepTable { WPrepJob job[5 + 2 * DA_MAX_LAYERS]; int n; };
__global__ __launch_bounds__(256) void k_weight_prep(WPrepTable tab) {
    __shared__ float tile[32][33];
    const WPrepJob jb = tab.job[blockIdx.y];
    const int tcols = (jb.cols + 31) / 32, trows = (jb.rows + 31) / 32;
    if ((int)blockIdx.x >= tcols * trows) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int r0 = ((int)blockIdx.x / tcols) * 32, c0 = ((int)blockIdx.x % tcols) * 32;
    if (jb.mode == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            if (r < jb.rows && c < jb.cols) ((bf16_t *)jb.dst)[(size_t)r * jb.cols + c] = f2bf(jb.src[(size_t)r * jb.cols + c]);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < jb.rows && c < jb.cols) tile[ty + 8 * k][tx] = jb.src[(size_t)r * jb.cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < jb.rows && c < jb.cols) {
            if (jb.mode == 1) ((bf16_t *)jb.dst)[(size_t)c * jb.rows + r] = f2bf(tile[tx][ty + 8 * k]);
            else ((float *)jb.dst)[(size_t)c * jb.rows + r] = tile[tx][ty + 8 * k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct TrainWs {
    float *comb_in, *m1pre, *m1, *h0;
    float *qkvs[DA_MAX_LAYERS], *o[DA_MAX_LAYERS], *hact[DA_MAX_LAYERS], *stats[DA_MAX_LAYERS];
    float *f1pre, *f1;
    float *dz, *dh0, *dY4, *dxa, *dxb, *Dd, *dm1, *df1, *dcomb, *wt, *partial, *csum, *pa, *p1, *dp1;
    float *P[DA_MAX_LAYERS], *dP;      // dense path: attention matrices kept per layer, one gradient scratch
    long long *poff;
    int32_t *node_graph;
    // weight images written by the forward's k_weight_prep launch, read by the backward of the same step: W^T of every Linear with
    // a dX product (fp32; bf16 for the convs in the q16 mode) and, q16 mode, the convs' bf16 forward operands
    void *wt_conv[DA_MAX_LAYERS];
    bf16_t *wh_conv[DA_MAX_LAYERS];
    float *partial2, *dY4b;            // side-stream dW products: their own split scratch, and the second projection-gradient buffer (see SideDw)
    bf16_t *x16[DA_MAX_LAYERS];        // q16 mode: bf16 image of every conv's input (the projection's operand in the forward, X of its dW product in the backward)
    float *wt_head1, *wt_head0, *wt_mlp1, *wt_mlp0, *wt_pos1;
    size_t total;
};

struct Dims {
    int nr, n, F, D, hid, H, L, V, c_in, c_out, G;
    int din[DA_MAX_LAYERS], C[DA_MAX_LAYERS], hc[DA_MAX_LAYERS];
    bool gelu_between;
    bool dense;                // complete graphs: grouped-GEMM attention (da_train_dense.hip)
    bool hybrid;               // hybrid graphs: masked grouped GEMMs + CSR remainder (da_train_dense.hip)
    bool bfc;                  // DA_TRAIN_MMA_BF16: GEMM operands rounded to bf16 inside the matrix-core kernels (storage stays fp32)
    bool q16;                  // bf16-operand mode on small complete graphs: the projection buffers Q | K | V | skip and their gradient dY4 are
                               // STORED as bf16 (what autocast(bfloat16) makes of the reference's lin_query / key / value / skip outputs): they
                               // are only ever read as bf16 operands (k_attn_small_*, the dX and dW products), and at BASELINE configuration 5
                               // the last layer's two [n, 4608] buffers alone are 340 MB of fp32 per step
    size_t pair_floats;
};

static int dims_of(const da_weights *w, const da_graph *g, Dims &d, int mma = DA_TRAIN_MMA_FP32) {
    d.bfc = mma == DA_TRAIN_MMA_BF16;
    DA_REQUIRE(w && g, "training: null argument");
    DA_REQUIRE(w->variant == DA_VARIANT_2D, "training: only the 2D denoiser is implemented");
    DA_REQUIRE(w->heads == 8 && w->n_layers >= 2 && w->n_layers <= DA_MAX_LAYERS, "training: bad heads / n_layers");
    d.nr = g->n_real; d.n = g->n_nodes; d.F = w->feat_dim; d.D = w->feat_dim + 64; d.hid = w->hidden; d.H = w->heads;
    d.L = w->n_layers; d.V = w->arch == DA_ARCH_EXOPHORMER ? w->virt_nodes : 0; d.c_in = w->c_in; d.c_out = w->c_out;
    d.gelu_between = w->arch == DA_ARCH_TRANSFORMER;
    d.G = g->n_graphs;
    DA_REQUIRE(d.D % d.H == 0 && (d.D / d.H) % 8 == 0, "training: D / heads must be a multiple of 8");
    for (int l = 0; l < d.L; ++l) {
        d.din[l] = l == 0 ? d.D : 32 * d.H;
        d.C[l] = l == d.L - 1 ? d.D / d.H : 32;
        d.hc[l] = d.C[l] * d.H;
    }
    DA_REQUIRE(g->n_real > 0 && g->n_nodes >= g->n_real, "training: bad graph");
    DA_REQUIRE(d.V == 0 || g->n_nodes == g->n_real + d.V * g->n_graphs, "training: exophormer expects n_nodes = n_real + V*G");
    d.dense = !train_dense_disabled() && g->dense != 0 && g->graph_ptr && d.V == 0 && g->max_graph_nodes > 0;
    d.hybrid = !d.dense && !train_dense_disabled() && g->hybrid && g->mask && g->mask_ptr && g->irr_row_ptr && g->graph_ptr &&
               g->pad_ptr && g->max_graph_nodes > 0;
    // the hybrid training kernels index the adjacency bits by NODE (row i = node - graph_ptr[g]); a plan in the banded slot layout
    // (graph_plan.expander_plan: bits in SLOT space, slot_node != NULL) would be read in the wrong order -- refuse it loudly
    DA_REQUIRE(!(d.hybrid && g->slot_node), "training: hybrid graph in the banded slot layout (slot_node set); train on a natural-layout plan: build_plan(edge_index, batch), expander_plan(..., banded=False) or DA_EXPANDER_LAYOUT=natural");
    DA_REQUIRE(d.dense || d.hybrid || g->row_ptr, "training: this graph walks the edge list but the CSR arrays are missing");
    // hybrid graphs in the bf16-operand mode run flash-style (da_train_dense.hip: k_hyb_*): no pair matrix is ever allocated
    bool flash = d.hybrid && d.bfc;
    for (int l = 0; l < d.L && flash; ++l) flash = hybrid_flash_ok(g, d.C[l], d.bfc);
    d.pair_floats = (d.dense || (d.hybrid && !flash)) ? dense_pair_floats(g, d.H) : 0;
    static int q16_off = -1;
    if (q16_off < 0) {
        q16_off = (DA_XENV("DA_TRAIN_Q16", 1) == 0 || DA_XENV("DA_TRAIN_TN_DB", 1) == 0) ? 1 : 0;
    }
    d.q16 = d.bfc && d.dense && !q16_off;
    for (int l = 0; l < d.L && d.q16; ++l) d.q16 = attn_small_ok(g, d.C[l], true) && d.din[l] % 32 == 0 && d.C[l] % 8 == 0;
    return 0;
}

static TrainWs carve_train(const Dims &d, void *base) {
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float *q = p ? (float *)(p + off) : nullptr;
        off += align_up(floats * 4, 256);
        return q;
    };
    const size_t n = (size_t)d.n + 64, nr = (size_t)d.nr + 64;
    TrainWs w;
    w.comb_in = take(nr * d.D);
    w.m1pre = take(nr * d.hid);
    w.m1 = take(nr * d.hid);
    w.h0 = take(n * d.D);
    int hcmax = 0;
    size_t wmax = (size_t)d.D * d.hid;
    for (int l = 0; l < d.L; ++l) {
        w.qkvs[l] = take(n * 4 * d.hc[l]);
        w.o[l] = take(n * d.hc[l]);
        w.hact[l] = (l < d.L - 1 && d.gelu_between) ? take(n * d.hc[l]) : nullptr;
        w.stats[l] = take(n * d.H * 2);
        hcmax = d.hc[l] > hcmax ? d.hc[l] : hcmax;
        const size_t ws = (size_t)4 * d.hc[l] * d.din[l];
        wmax = ws > wmax ? ws : wmax;
    }
    w.f1pre = take(nr * 32);
    w.f1 = take(nr * 32);
    w.dz = take(n * d.D);
    w.dh0 = take(n * d.D);
    w.dY4 = take(n * 4 * hcmax);
    w.dxa = take(n * d.D);
    w.dxb = take(n * d.D);
    w.Dd = take(n * d.H);
    w.dm1 = take(nr * d.hid);
    w.df1 = take(nr * 32);
    w.dcomb = take(nr * d.D);
    w.wt = take(wmax + 1024);
    w.partial = take(PART_CAP);
    w.csum = take(((size_t)d.n / 128 + 2) * 4 * (size_t)hcmax);
    w.pa = take(nr * 16);
    w.p1 = take(nr * 16);
    w.dp1 = take(nr * 16);
    for (int l = 0; l < DA_MAX_LAYERS; ++l) w.P[l] = nullptr;
    w.dP = nullptr; w.poff = nullptr; w.node_graph = nullptr;
    if (d.dense || d.hybrid) {
        for (int l = 0; l < d.L; ++l) w.P[l] = take(d.pair_floats);
        w.dP = take(d.pair_floats);
        w.poff = (long long *)take(2 * ((size_t)d.G + 2));
        w.node_graph = (int32_t *)take(n);
    }
    for (int l = 0; l < DA_MAX_LAYERS; ++l) { w.wt_conv[l] = nullptr; w.wh_conv[l] = nullptr; w.x16[l] = nullptr; }
    for (int l = 0; l < d.L; ++l) {
        const size_t we = (size_t)4 * d.hc[l] * d.din[l];
        w.wt_conv[l] = take(d.q16 ? (we + 1) / 2 : we);
        w.wh_conv[l] = d.q16 ? (bf16_t *)take((we + 1) / 2) : nullptr;
        w.x16[l] = d.q16 ? (bf16_t *)take((n * d.din[l] + 1) / 2) : nullptr;
    }
    w.partial2 = take(PART_CAP);
    w.dY4b = take(n * 4 * hcmax);
    w.wt_head1 = take((size_t)d.c_out * 32);
    w.wt_head0 = take((size_t)32 * d.D);
    w.wt_mlp1 = take((size_t)d.D * d.hid);
    w.wt_mlp0 = take((size_t)d.hid * d.D);
    w.wt_pos1 = take((size_t)32 * 16);
    w.total = off;
    return w;
}

static int check_fused(const da_weights *w, const Dims &d, const char *what) {
    for (int l = 0; l < d.L; ++l) {
        const size_t blk = (size_t)d.hc[l] * d.din[l];
        DA_REQUIRE(w->conv_wq[l] && w->conv_bq[l], "%s: conv %d pointers missing", what, l);
        DA_REQUIRE(w->conv_wk[l] == w->conv_wq[l] + blk && w->conv_wv[l] == w->conv_wq[l] + 2 * blk &&
                       w->conv_ws[l] == w->conv_wq[l] + 3 * blk && w->conv_bk[l] == w->conv_bq[l] + d.hc[l] &&
                       w->conv_bv[l] == w->conv_bq[l] + 2 * d.hc[l] && w->conv_bs[l] == w->conv_bq[l] + 3 * d.hc[l],
                   "%s: conv %d needs lin_query|key|value|skip contiguous in that order (flat parameter buffer)", what, l);
    }
    return 0;
}

static int q16_cast_on() {
    static int v = -1;
    if (v < 0) v = DA_XENV("DA_TRAIN_Q16_CAST", 1) ? 1 : 0;
    return v;
}

static int weight_prep(const da_weights *w, const Dims &d, TrainWs &ws, hipStream_t st) {
    WPrepTable tab;
    tab.n = 0;
    int maxt = 1;
    auto add = [&](const float *src, void *dst, int rows, int cols, int mode) {
        WPrepJob &j = tab.job[tab.n++];
        j.src = src; j.dst = dst; j.rows = rows; j.cols = cols; j.mode = mode; j.pad = 0;
        const int tiles = ((rows + 31) / 32) * ((cols + 31) / 32);
        maxt = tiles > maxt ? tiles : maxt;
    };
    static_assert(sizeof(WPrepTable) <= 2048, "the job table travels as a kernel argument");
    add(w->head_w1, ws.wt_head1, d.c_out, 32, 0);
    add(w->head_w0, ws.wt_head0, 32, d.D, 0);
    add(w->mlp_w1, ws.wt_mlp1, d.D, d.hid, 0);
    add(w->mlp_w0, ws.wt_mlp0, d.hid, d.D, 0);
    add(w->pos_w1, ws.wt_pos1, 32, 16, 0);
    for (int l = 0; l < d.L; ++l) {
        add(w->conv_wq[l], ws.wt_conv[l], 4 * d.hc[l], d.din[l], d.q16 ? 1 : 0);
        if (d.q16) add(w->conv_wq[l], ws.wh_conv[l], 4 * d.hc[l], d.din[l], 2);
    }
    k_weight_prep<<<dim3((unsigned)maxt, (unsigned)tab.n), 256, 0, st>>>(tab);
    DA_LAUNCH_CHECK();
    return 0;
}

static int gelu_fwd(size_t n, const float *src, float *dst, hipStream_t st, bf16_t *dst16 = nullptr) {
    k_gelu_fwd<<<grid_for(n), 256, 0, st>>>(n, src, dst, dst16);
    DA_LAUNCH_CHECK();
    return 0;
}
static int gelu_bwd(size_t n, const float *pre, const float *dy, float *dx, hipStream_t st) {
    k_gelu_bwd<<<grid_for(n), 256, 0, st>>>(n, pre, dy, dx);
    DA_LAUNCH_CHECK();
    return 0;
}
int launch_transpose_f32(int rows, int cols, const float *src, float *dst, hipStream_t st) {
    k_transpose<<<dim3((cols + 31) / 32, (rows + 31) / 32), 256, 0, st>>>(rows, cols, src, dst);
    DA_LAUNCH_CHECK();
    return 0;
}
int colsum_add(int M, int N, const float *A, int lda, float *out, float *scratch, hipStream_t st) {
    // at most ~32 row chunks (never more than the 128-row chunks the scratch is sized for): chunks * N floats of scratch
    const int rows_per = max(128, ((M + 31) / 32 + 3) / 4 * 4), chunks = (M + rows_per - 1) / rows_per;
    k_colsum_partial<<<dim3((N + 63) / 64, chunks), 256, 0, st>>>(M, N, A, lda, rows_per, scratch);
    k_colsum_finish<<<(N + 63) / 64, 256, 0, st>>>(chunks, N, scratch, out);
    DA_LAUNCH_CHECK();
    return 0;
}

// The weight-gradient products run BESIDE the backward's critical path (round 5).  Only dX feeds the next layer; dW = dY^T X and db
// are leaves.  At BASELINE configuration 5 (9 216 nodes per GPU) every kernel of the step is a few hundred workgroups deep in
// dependent load -> LDS -> MFMA stages and leaves most of the chip idle, so the leaves are issued on a side stream of the library
// (fork: an event on the caller's stream once dY is final; join: at the end of every da_train_backward_stage call, i.e. before the
// caller can read a gradient or start an exchange) and overlap with the dX product, the GELU backward and the next layer's attention
// backward.  What the two streams share is kept apart: the side stream has its own split-partial scratch (partial2; csum is only
// ever used by dW products), and the convs' projection gradient dY4 alternates between two buffers -- layer l's attention backward
// may only overwrite the buffer of layer l + 2 after that layer's dW product has read it (done[l + 2]).  DA_TRAIN_SIDE_DW=0: everything
// on the caller's stream, launch for launch the round-4 order.
struct SideDw {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, done[DA_MAX_LAYERS] = {};
};
// one context per (device, caller stream): two engines driven on different streams of one process must not share events
static SideDw *side_dw(hipStream_t caller) {
    if (!(cfg().train_side_streams & 1)) return nullptr;          // da_config.train_side_streams bit 0
    struct Slot { int dev; hipStream_t caller; SideDw ctx; };
    static std::mutex mu;
    static std::vector<Slot *> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (Slot *sl : slots)
        if (sl->dev == dev && sl->caller == caller) return &sl->ctx;
    if (slots.size() >= 64) return nullptr;                 // (a process that cycles through streams: fall back to the one-stream order)
    Slot *sl = new Slot{dev, caller, SideDw()};
    SideDw &c = sl->ctx;
    bool good = hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) == hipSuccess;
    good = good && hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c.join, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; l < DA_MAX_LAYERS && good; ++l) good = hipEventCreateWithFlags(&c.done[l], hipEventDisableTiming) == hipSuccess;
    if (!good) { delete sl; return nullptr; }
    slots.push_back(sl);
    return &sl->ctx;
}

// Linear backward: dW += dY^T X, db += colsum(dY), and (if dX) dX = dY @ W (+ res); WT = the forward's image of W^T (k_weight_prep:
// fp32, or bf16 when dy16); gelu_pre (optional, same leading dimension as dX): dX *= gelu'(gelu_pre) -- inside the reduction-split
// product's second kernel where that route is taken, as a launch of its own otherwise.  sd: the dW / db launches go to the side
// stream (after an event that says dY is final) and `done`, if given, is recorded behind them
static int linear_bwd(int M, int N, int K, const float *dY, int ldy, const float *X, int ldx, const void *WT,
                      float *dW, float *db, float *dX, int lddx, const float *res, TrainWs &ws, hipStream_t st, bool bfc,
                      bool dy16 = false, const float *gelu_pre = nullptr, const bf16_t *X16 = nullptr, SideDw *sd = nullptr,
                      hipEvent_t done = nullptr) {
    int rc;
    auto gelu_tail = [&]() -> int { return gelu_pre ? gelu_bwd((size_t)M * K, gelu_pre, dX, dX, st) : 0; };      // (every caller's dX is dense: lddx == K)
    hipStream_t sw = st;                                    // stream and split scratch of the dW / db launches
    float *part = ws.partial;
    if (sd) {
        DA_CHECK_HIP(hipEventRecord(sd->fork, st));
        DA_CHECK_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
        sw = sd->s;
        part = ws.partial2;
    }
    static int tn_db = -1;
    if (tn_db < 0) tn_db = DA_XENV("DA_TRAIN_TN_DB", 1) ? 1 : 0;
    if (dy16) {                                             // q16 mode: dY is bf16 (written by k_attn_small_bwd)
        // X16: the bf16 image of X the forward's projection multiplied (dense, leading dimension K) -- the same bits the fp32 route
        // rounds to inside the kernel
        if ((rc = X16 ? launch_gemm_tn_db(M, N, K, dY, ldy, X16, K, dW, K, part, db, ws.csum, sw, true, true)
                      : launch_gemm_tn_db(M, N, K, dY, ldy, X, ldx, dW, K, part, db, ws.csum, sw, true))) return rc;
    } else if (bfc && tn_db) {                              // dW and db from one pass over dY (k_gemm_tn_db)
        if ((rc = launch_gemm_tn_db(M, N, K, dY, ldy, X, ldx, dW, K, part, db, ws.csum, sw))) return rc;
    } else {
        if ((rc = launch_gemm_tn(M, N, K, dY, ldy, X, ldx, dW, K, part, sw, bfc))) return rc;
        if (db && (rc = colsum_add(M, N, dY, ldy, db, ws.csum, sw))) return rc;
    }
    if (sd && done) DA_CHECK_HIP(hipEventRecord(done, sd->s));
    if (!dX) return 0;
    if (dy16) {
        rc = launch_gemm_mfma_splitk(M, N, K, dY, ldy, WT, nullptr, res, dX, lddx, ws.partial, PART_CAP, st, true, gelu_pre);
        if (rc >= 0) return rc;
        rc = launch_gemm_mfma_mixed(true, false, M, N, K, dY, ldy, WT, nullptr, res, dX, lddx, st);
        if (rc < 0) { set_error("training (q16): dX product %d x %d x %d not covered", M, N, K); return 1; }
        return rc ? rc : gelu_tail();
    }
    if (bfc) {                                              // skinny dX with a long reduction: split over the reduction (one launch + a fixed-order sum)
        rc = launch_gemm_mfma_splitk(M, N, K, dY, ldy, WT, nullptr, res, dX, lddx, ws.partial, PART_CAP, st, false, gelu_pre);
        if (rc >= 0) return rc;
    }
    if ((rc = linear(bfc ? DA_PREC_F32_BF16MMA : DA_PREC_F32, M, N, K, dY, ldy, (const float *)WT, nullptr, DA_ACT_NONE, res, dX, lddx, st))) return rc;
    return gelu_tail();
}

// forward Linear of the training path: the bf16-operand mode tries the reduction-split launch first (skinny outputs with a
// long reduction: mlp.0 and the head's first layer leave most CUs idle otherwise)
// act_out (optional, dense [M, Nout] like out): gelu(out) -- written by the split product's second kernel, or by a launch of its own
static int linear_fw(const Dims &d, TrainWs &ws, int M, int K, int Nout, const float *A, int lda, const float *W, const float *bias,
                     float *out, int ldo, hipStream_t st, float *act_out = nullptr) {
    if (d.bfc) {
        const int rc = launch_gemm_mfma_splitk(M, K, Nout, A, lda, W, bias, nullptr, out, ldo, ws.partial, PART_CAP, st, false, nullptr, act_out);
        if (rc >= 0) return rc;
    }
    const int rc = linear(d.bfc ? DA_PREC_F32_BF16MMA : DA_PREC_F32, M, K, Nout, A, lda, W, bias, DA_ACT_NONE, nullptr, out, ldo, st);
    if (rc || !act_out) return rc;
    return gelu_fwd((size_t)M * Nout, out, act_out, st);
}

}  // namespace da

using namespace da;

extern "C" {

size_t da_train_workspace_bytes(const da_weights *w, const da_graph *g) {
    return da_train_workspace_bytes_ex(w, g, DA_TRAIN_MMA_FP32);
}

size_t da_train_workspace_bytes_ex(const da_weights *w, const da_graph *g, int mma_precision) {
    Dims d;
    if (dims_of(w, g, d, mma_precision)) return 0;
    return carve_train(d, nullptr).total;
}

int da_train_forward(const da_weights *w, const da_graph *g, const float *x, const int64_t *t, const float *feats,
                     float *out, void *workspace, size_t workspace_bytes, void *stream) {
    return da_train_forward_ex(w, g, x, t, feats, out, workspace, workspace_bytes, DA_TRAIN_MMA_FP32, stream);
}

int da_train_forward_ex(const da_weights *w, const da_graph *g, const float *x, const int64_t *t, const float *feats,
                        float *out, void *workspace, size_t workspace_bytes, int mma_precision, void *stream) {
    Dims d;
    int rc;
    DA_REQUIRE(mma_precision == DA_TRAIN_MMA_FP32 || mma_precision == DA_TRAIN_MMA_BF16, "da_train_forward: unknown mma_precision %d", mma_precision);
    if ((rc = dims_of(w, g, d, mma_precision))) return rc;
    DA_REQUIRE(x && t && feats && out && workspace, "da_train_forward: null argument");
    if ((rc = check_fused(w, d, "da_train_forward"))) return rc;
    TrainWs ws = carve_train(d, workspace);
    DA_REQUIRE(workspace_bytes >= ws.total, "training workspace too small: %zu < %zu", workspace_bytes, ws.total);
    hipStream_t st = (hipStream_t)stream;
    const int P = DA_PREC_F32, nr = d.nr, n = d.n, D = d.D;
    const int PL = d.bfc ? DA_PREC_F32_BF16MMA : DA_PREC_F32;          // precision code of the linear layers (storage is fp32 either way)
    // W^T / bf16 images of the step (the backward of this forward reads them too): on the library's side stream, beside the
    // feature copy, the embedding and the mlp -- the first reader is conv 0's projection
    SideDw *sd = side_dw(st);
    StreamJoin side_guard;              // every exit below joins the side stream (the error exits through its destructor)
    if (sd) {
        DA_CHECK_HIP(hipEventRecord(sd->fork, st));         // (behind the optimizer step / whatever last wrote the weights on the caller's stream)
        DA_CHECK_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
        side_guard.arm(sd->s, st, sd->join);
    }
    if ((rc = weight_prep(w, d, ws, sd ? sd->s : st))) return rc;
    // (the pair offsets / node -> graph table of the grouped attention kernels: the same side stream, needed by the first attention)
    if (sd && (d.dense || d.hybrid) && (rc = dense_train_prepare(g, d.H, ws.poff, ws.node_graph, sd->s))) return rc;
    hipEvent_t ev_images = sd ? sd->done[0] : nullptr;       // (done[] belongs to the backward; idle during a forward)
    if (sd) DA_CHECK_HIP(hipEventRecord(ev_images, sd->s));
    if ((rc = launch_set_feats(P, nr, d.F, D, feats, ws.comb_in, st))) return rc;
    if ((rc = launch_embed_pos_time(P, nr, d.c_in, d.F, D, x, t, 0, w->steps, w->time_emb, w->pos_w0, w->pos_b0, w->pos_w1,
                                    w->pos_b1, ws.comb_in, st))) return rc;
    if ((rc = linear_fw(d, ws, nr, D, d.hid, ws.comb_in, D, w->mlp_w0, w->mlp_b0, ws.m1pre, d.hid, st, ws.m1))) return rc;
    if ((rc = linear(PL, nr, d.hid, D, ws.m1, d.hid, w->mlp_w1, w->mlp_b1, DA_ACT_NONE, nullptr, ws.h0, D, st))) return rc;
    if (d.V > 0) {
        DA_REQUIRE(w->virt_emb, "exophormer: virt_emb missing");
        if ((rc = launch_set_virtual_rows(P, n - nr, d.V, D, w->virt_emb, ws.h0 + (size_t)nr * D, st))) return rc;
    }
    if (sd) { DA_CHECK_HIP(hipStreamWaitEvent(st, ev_images, 0)); side_guard.armed = false; }      // the weight images are there (nothing else runs on the side stream in a forward)
    const float *xin = ws.h0;
    int ldx = D;
    bool x16_ready = false;           // x16[l] already holds the bf16 image of xin (written by the GELU launch of the layer before)
    for (int l = 0; l < d.L; ++l) {
        const bool last = l == d.L - 1;
        if (d.q16) {                                        // bf16 projection buffer (same allocation, half used)
            // input and weight are rounded to bf16 ONCE, by a pass of their own (into dY4 and wt: scratch of the backward, idle
            // here), and the projection runs on the bf16 kernels of the inference path (W in registers / a 64-deep LDS ring
            // instead of 128 x 128 fp32 tiles re-streamed through the L2 by every column group); the products are the same
            // bf16 x bf16 -> fp32 ones.  DA_TRAIN_Q16_CAST=0: fp32 operands rounded inside the kernel (launch_gemm_mfma_mixed)
            const int cast_on = q16_cast_on();
            rc = -1;
            if (cast_on && ldx == d.din[l] && ((size_t)n * d.din[l]) % 8 == 0) {
                // (the weight's bf16 image comes from k_weight_prep; the input's from the GELU launch that produced it -- x16_ready --
                //  or, for layer 0 and the architectures without a GELU between the layers, from a cast of its own)
                if (!x16_ready && (rc = cast_h((size_t)n * d.din[l], xin, ws.x16[l], st))) return rc;
                x16_ready = true;
                rc = launch_gemm_mfma(DA_PREC_BF16, n, d.din[l], 4 * d.hc[l], ws.x16[l], d.din[l], ws.wh_conv[l], w->conv_bq[l], DA_ACT_NONE,
                                      nullptr, ws.qkvs[l], 4 * d.hc[l], nullptr, st);
            }
            if (rc < 0)
            rc = launch_gemm_mfma_mixed(false, true, n, d.din[l], 4 * d.hc[l], xin, ldx, w->conv_wq[l], w->conv_bq[l], nullptr,
                                        ws.qkvs[l], 4 * d.hc[l], st);
            if (rc < 0) { set_error("training (q16): projection %d x %d x %d not covered", n, d.din[l], 4 * d.hc[l]); return 1; }
            if (rc) return rc;
        } else
        if ((rc = linear(PL, n, d.din[l], 4 * d.hc[l], xin, ldx, w->conv_wq[l], w->conv_bq[l], DA_ACT_NONE, nullptr,
                         ws.qkvs[l], 4 * d.hc[l], st))) return rc;
        if (d.dense) {
            if (l == 0 && !sd && (rc = dense_train_prepare(g, d.H, ws.poff, ws.node_graph, st))) return rc;
            if ((rc = dense_train_attn_fwd(g, d.H, d.C[l], ws.qkvs[l], last ? ws.h0 : nullptr, ws.o[l], ws.P[l], ws.poff,
                                           ws.node_graph, st, d.bfc, d.q16))) return rc;
        } else if (d.hybrid) {
            if (l == 0 && !sd && (rc = dense_train_prepare(g, d.H, ws.poff, ws.node_graph, st))) return rc;
            if ((rc = hybrid_train_attn_fwd(g, d.H, d.C[l], ws.qkvs[l], last ? ws.h0 : nullptr, ws.o[l], ws.P[l], ws.stats[l],
                                            ws.poff, ws.node_graph, st, d.bfc))) return rc;
        } else if ((rc = launch_attn_csr(P, n, g->row_ptr, g->col_src, nullptr, d.H, d.C[l], ws.qkvs[l], last ? ws.h0 : nullptr,
                                         DA_ACT_NONE, ws.o[l], nullptr, ws.stats[l], st))) return rc;
        x16_ready = false;
        if (!last && d.gelu_between) {
            // q16 mode: the same launch leaves the bf16 image of the next projection's input in x16[l + 1] (kept for the backward's dW product)
            const bool to16 = d.q16 && q16_cast_on() && ((size_t)n * d.hc[l]) % 8 == 0 && d.hc[l] == d.din[l + 1];
            if ((rc = gelu_fwd((size_t)n * d.hc[l], ws.o[l], ws.hact[l], st, to16 ? ws.x16[l + 1] : nullptr))) return rc;
            x16_ready = to16;
            xin = ws.hact[l];
        } else {
            xin = ws.o[l];
        }
        ldx = d.hc[l];
    }
    const float *z = ws.o[d.L - 1];                           // conv output + combined (efficient_gat.py:144)
    if ((rc = linear_fw(d, ws, nr, D, 32, z, D, w->head_w0, w->head_b0, ws.f1pre, 32, st, ws.f1))) return rc;
    return launch_head2d(P, nr, d.c_out, ws.f1, w->head_w1, w->head_b1, out, st);
}

int da_train_backward(const da_weights *w, const da_weights *grads, const da_graph *g, const float *x,
                      const int64_t *t, const float *d_out, float *d_feats, void *workspace, size_t workspace_bytes,
                      void *stream) {
    return da_train_backward_ex(w, grads, g, x, t, d_out, d_feats, workspace, workspace_bytes, DA_TRAIN_MMA_FP32, stream);
}

int da_train_backward_ex(const da_weights *w, const da_weights *grads, const da_graph *g, const float *x,
                         const int64_t *t, const float *d_out, float *d_feats, void *workspace, size_t workspace_bytes,
                         int mma_precision, void *stream) {
    return da_train_backward_stage(w, grads, g, x, t, d_out, d_feats, workspace, workspace_bytes, mma_precision, DA_TRAIN_BWD_ALL, stream);
}

// The backward in two halves, cut where the gradients of final_mlp and of every conv but the first are complete (the data-parallel
// exchange of that bucket -- 1.7 M of the 3.2 M parameters -- can then run on a side stream under the largest dW / dX products
// of the step, conv 0's, and the mlp / embedding tail): DA_TRAIN_BWD_EARLY = head, convs L-1 .. 1; DA_TRAIN_BWD_LATE = conv 0,
// virtual-node embedding, mlp, concat pieces.  EARLY then LATE on one stream == DA_TRAIN_BWD_ALL, launch for launch.
int da_train_backward_stage(const da_weights *w, const da_weights *grads, const da_graph *g, const float *x,
                            const int64_t *t, const float *d_out, float *d_feats, void *workspace, size_t workspace_bytes,
                            int mma_precision, int stage, void *stream) {
    Dims d;
    DA_REQUIRE(stage == DA_TRAIN_BWD_ALL || stage == DA_TRAIN_BWD_EARLY || stage == DA_TRAIN_BWD_LATE, "da_train_backward_stage: unknown stage %d", stage);
    const bool do_early = stage != DA_TRAIN_BWD_LATE, do_late = stage != DA_TRAIN_BWD_EARLY;
    int rc;
    DA_REQUIRE(mma_precision == DA_TRAIN_MMA_FP32 || mma_precision == DA_TRAIN_MMA_BF16, "da_train_backward: unknown mma_precision %d", mma_precision);
    if ((rc = dims_of(w, g, d, mma_precision))) return rc;
    DA_REQUIRE(grads && x && t && d_out && workspace, "da_train_backward: null argument");
    DA_REQUIRE(d.dense || (g->out_ptr && (g->out_dst || d.hybrid)), "da_train_backward: the graph needs the by-source CSR (out_ptr / "
               "out_dst; of the remainder edges for hybrid graphs, where out_dst may be empty)");
    if ((rc = check_fused(w, d, "da_train_backward(weights)"))) return rc;
    if ((rc = check_fused(grads, d, "da_train_backward(grads)"))) return rc;
    TrainWs ws = carve_train(d, workspace);
    DA_REQUIRE(workspace_bytes >= ws.total, "training workspace too small: %zu < %zu", workspace_bytes, ws.total);
    hipStream_t st = (hipStream_t)stream;
    const int nr = d.nr, n = d.n, D = d.D, L = d.L;
    auto G = [](const float *p) { return (float *)p; };       // grads: same struct, written by the library

    const bool dh0_copy = n > nr;
    SideDw *sd = side_dw(st);           // null: DA_TRAIN_SIDE_DW=0
    StreamJoin side_guard;              // every exit joins the dW / db launches of the side stream (error exits through its destructor)
    if (sd) side_guard.arm(sd->s, st, sd->join);
    static int dw_x16 = -1;             // DA_TRAIN_DW_X16=1: the convs' dW products read the bf16 image of X (measured slower: 69 vs 63 us at conv 3)
    if (dw_x16 < 0) dw_x16 = DA_XENV("DA_TRAIN_DW_X16", 0) ? 1 : 0;
    if (do_early) {
    // ---- head: final_mlp.2, GELU, final_mlp.0 (efficient_gat.py:145)
    if ((rc = linear_bwd(nr, d.c_out, 32, d_out, d.c_out, ws.f1, 32, ws.wt_head1, G(grads->head_w1), G(grads->head_b1),
                         ws.df1, 32, nullptr, ws, st, d.bfc, false, ws.f1pre, nullptr, sd))) return rc;
    const float *z = ws.o[L - 1];
    if (n > nr) DA_CHECK_HIP(hipMemsetAsync(ws.dz + (size_t)nr * D, 0, (size_t)(n - nr) * D * 4, st));
    if ((rc = linear_bwd(nr, 32, D, ws.df1, 32, z, D, ws.wt_head0, G(grads->head_w0), G(grads->head_b0), ws.dz, D, nullptr,
                         ws, st, d.bfc, false, nullptr, nullptr, sd))) return rc;
    // residual: z = conv_out + h0  ->  both get dz.  Without virtual rows dh0 = dz is not materialised: layer 0's dX product
    // takes dz as its residual operand and writes dh0 (dz is read-only from here on)
    if (dh0_copy) DA_CHECK_HIP(hipMemcpyAsync(ws.dh0, ws.dz, (size_t)n * D * 4, hipMemcpyDeviceToDevice, st));
    }

    // ---- graph transformer layers, last to first (EARLY: L-1 .. 1; LATE: 0, whose incoming gradient is layer 1's dX buffer)
    const float *d_o = do_early ? ws.dz : (L > 1 ? ws.dxa : ws.dz);
    const int l_first = do_early ? L - 1 : 0;
    for (int l = l_first; l >= (do_late ? 0 : 1); --l) {
        const int hc = d.hc[l], din = d.din[l];
        // this layer's projection gradient: the two buffers alternate, and the one about to be overwritten was last read by layer
        // l + 2's dW product on the side stream
        float *dY4 = (sd && ((L - 1 - l) & 1)) ? ws.dY4b : ws.dY4;
        if (sd && l + 2 <= l_first) DA_CHECK_HIP(hipStreamWaitEvent(st, sd->done[l + 2], 0));
        if (d.dense) {
            if ((rc = dense_train_attn_bwd(g, d.H, d.C[l], ws.qkvs[l], d_o, ws.P[l], ws.dP, dY4, ws.poff, ws.node_graph, st, d.bfc, d.q16))) return rc;
        } else if (d.hybrid) {
            if ((rc = hybrid_train_attn_bwd(g, d.H, d.C[l], ws.qkvs[l], d_o, ws.P[l], ws.dP, ws.stats[l], ws.Dd, dY4, ws.poff,
                                            ws.node_graph, st, d.bfc, ws.o[l], l == L - 1 ? ws.h0 : nullptr))) return rc;
        } else if ((rc = launch_attn_bwd(g, d.H, d.C[l], ws.qkvs[l], d_o, ws.stats[l], ws.Dd, dY4, st))) return rc;
        const float *xin = l == 0 ? ws.h0 : (d.gelu_between ? ws.hact[l - 1] : ws.o[l - 1]);
        float *dx = (l & 1) ? ws.dxa : ws.dxb;
        // l == 0: the input is h0, whose gradient also carries the residual branch
        if ((rc = linear_bwd(n, 4 * hc, din, dY4, 4 * hc, xin, din, ws.wt_conv[l], G(grads->conv_wq[l]), G(grads->conv_bq[l]),
                             l == 0 ? ws.dh0 : dx, din, l == 0 ? (dh0_copy ? ws.dh0 : ws.dz) : nullptr, ws, st, d.bfc, d.q16,
                             (l > 0 && d.gelu_between) ? ws.o[l - 1] : nullptr,
                             (dw_x16 && d.q16 && q16_cast_on() && ((size_t)n * din) % 8 == 0) ? ws.x16[l] : nullptr,      // (the forward's condition for writing it)
                             sd, sd ? sd->done[l] : nullptr))) return rc;
        if (l > 0) d_o = dx;
    }
    auto join_side = [&]() -> int {         // the caller's stream continues behind every dW / db launch of this call
        DA_CHECK_HIP(side_guard.join());
        return 0;
    };
    if (!do_late) return join_side();
    // ---- virtual-node embedding (exophormer_gnn.py:169-178)
    if (d.V > 0) {
        DA_REQUIRE(grads->virt_emb, "exophormer: virt_emb gradient pointer missing");
        k_virt_grad<<<(d.V * D + 255) / 256, 256, 0, st>>>(n - nr, d.V, D, ws.dh0 + (size_t)nr * D, G(grads->virt_emb));
        DA_LAUNCH_CHECK();
    }
    // ---- mlp.2, GELU, mlp.0 (efficient_gat.py:135)
    if ((rc = linear_bwd(nr, D, d.hid, ws.dh0, D, ws.m1, d.hid, ws.wt_mlp1, G(grads->mlp_w1), G(grads->mlp_b1), ws.dm1, d.hid,
                         nullptr, ws, st, d.bfc, false, ws.m1pre, nullptr, sd))) return rc;
    if ((rc = linear_bwd(nr, d.hid, D, ws.dm1, d.hid, ws.comb_in, D, ws.wt_mlp0, G(grads->mlp_w0), G(grads->mlp_b0), ws.dcomb, D,
                         nullptr, ws, st, d.bfc, false, nullptr, nullptr, sd))) return rc;
    // ---- concat pieces: [feats | pos | time]
    if (d_feats) {
        k_copy_cols<<<grid_for((size_t)nr * d.F), 256, 0, st>>>(nr, d.F, ws.dcomb, D, d_feats);
        DA_LAUNCH_CHECK();
    }
    k_time_scatter<<<(((nr + 15) / 16) * 32 + 255) / 256, 256, 0, st>>>(nr, D, d.F, w->steps, t, ws.dcomb, G(grads->time_emb));
    DA_LAUNCH_CHECK();
    // pos_mlp (efficient_gat.py:133): Linear(c,16) GELU Linear(16,32); the hidden layer is recomputed
    k_pos_hidden<<<(nr * 16 + 255) / 256, 256, 0, st>>>(nr, d.c_in, x, w->pos_w0, w->pos_b0, ws.pa, ws.p1);
    DA_LAUNCH_CHECK();
    if ((rc = linear_bwd(nr, 32, 16, ws.dcomb + d.F, D, ws.p1, 16, ws.wt_pos1, G(grads->pos_w1), G(grads->pos_b1), ws.dp1, 16,
                         nullptr, ws, st, d.bfc, false, ws.pa, nullptr, sd))) return rc;
    if ((rc = linear_bwd(nr, 16, d.c_in, ws.dp1, 16, x, d.c_in, nullptr, G(grads->pos_w0), G(grads->pos_b0), nullptr, 0,
                         nullptr, ws, st, d.bfc, false, nullptr, nullptr, sd))) return rc;
    return join_side();
}

}  // extern "C"
