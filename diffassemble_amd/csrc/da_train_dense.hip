// Dense (complete-graph) attention for the TRAINING path, fp32, on the matrix cores.
//
// For complete puzzles (da_graph.dense != 0) the per-(graph, head) attention of PyG's TransformerConv is
// a pair of small dense matrix products; with 288 GB of HBM the attention matrix P (G*H*n^2 fp32: 42 MB
// per layer at BASELINE config 5, 0.8 GB at 32 x 900 pieces) is simply KEPT for the backward instead of
// being recomputed.  Everything is one grouped GEMM kernel (group = (graph, head), ragged n_g):
//   forward   S = scale Q K^T  ->  P = softmax rows (PyG: exp(s - max) / (sum + 1e-16); the diagonal is
//             excluded for graphs without self loops)  ->  O = P V + skip (+ residual)
//   backward  dP = dO V^T ; D_i = sum_j P_ij dP_ij ; dS = P o (dP - D) ; dV = P^T dO ;
//             dQ = scale dS K ; dK = scale dS^T Q
// operating directly on the row-major [n, 4HC] projection buffer (Q | K | V | skip) and writing dQ|dK|dV
// into the fused [n, 4HC] gradient, so the dense and the CSR training paths share every other kernel.
// (That is the exact-fp32 route and the route of graphs beyond 160 nodes.  In the bf16-operand mode, groups of up to 160
// nodes -- BASELINE configuration 5 -- take k_attn_small_fwd / k_attn_small_bwd further down instead: the whole attention of a
// (graph, head) in one workgroup, nothing but O kept, the softmax recomputed in the backward.)
//
// k_ggemm: C(m, n) = alpha * sum_k A(m, k) B(k, n) (+ C), 64 x 64 tile, 16 k per stage, 4 waves as 2 x 2,
// v_mfma_f32_16x16x4_f32.  Operands may be transposed views; both tiles are staged k-major ([k][m] and
// [k][n], row stride 80 floats) so the single-float MFMA fragments (row = lane & 15, k = lane >> 4) are
// conflict-free 4-byte LDS reads whatever the memory orientation; the global side always moves 16 bytes
// per lane along the contiguous axis.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "da_gemm_common.h"

namespace da {

struct GOp {
    float *base;
    int kind;        // 0: node matrix  (i, c) -> base[(gp[g] + i) * ld + h * hcol + c]
                     // 1: pair matrix  (i, j) -> base[poff[g] + (h * n_g + i) * ldp_g + j],  ldp_g = round4(n_g)
    int ld, hcol;
};
struct GGemm {
    GOp A, B, C;
    int transA, transB;          // A(m, k) = opA[k][m] if transA ; B(k, n) = opB[n][k] if transB
    int dimM, dimN, dimK;        // 0 = n_g (nodes of the graph), else the literal size (head width)
    float alpha;
    int accumulate;
    int H;
    int bfc;                     // operands rounded to bf16 in registers, v_mfma_f32_16x16x16_bf16 (training's DA_TRAIN_MMA_BF16 mode)
    int epi = 0;                 // k_ggemm_small only: 0 = plain; 1 = row softmax of the product (C = P, PyG: exp(s - max) / (sum + 1e-16),
                                 // diagonal excluded when nodiag); 2 = C = E o (product - rowsum(E o product)) with E = the kept P (dS)
    int nodiag = 0;
    const float *E = nullptr;    // epi == 2: pair matrix P (same layout as C)
    int gs_pitch = 0, gs_rows_a = 0;      // k_ggemm_small: LDS row pitch (elements) and rows reserved for the A image (set by the launcher)
    int pair_bf16 = 0;           // k_ggemm_small: pair matrices (kind 1 operands, E) are stored as bf16 (same element offsets, half the bytes)
    const int32_t *gp;           // [G + 1] node offsets
    const long long *poff;       // [G + 1] pair-matrix offsets (floats)
};

__device__ __forceinline__ const float *op_ptr(const GOp &o, int g, int h, int n_g, const int32_t *gp, const long long *poff,
                                               int &rs) {
    if (o.kind == 0) { rs = o.ld; return o.base + (size_t)gp[g] * o.ld + (size_t)h * o.hcol; }
    const int ldp = (n_g + 3) & ~3;
    rs = ldp;
    return o.base + poff[g] + (size_t)h * n_g * ldp;
}

// 4 consecutive elements along the contiguous axis starting at (r, c) of a [R, Cc] view, zero beyond
__device__ __forceinline__ f32x4 ld4z(const float *p, int rs, int r, int c, int R, int Cc) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r >= R || c >= Cc) return v;
    const float *q = p + (size_t)r * rs + c;
    if (c + 3 < Cc) return *(const f32x4 *)q;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c + e < Cc) v[e] = q[e];
    return v;
}

template <bool BFC>
__global__ __launch_bounds__(256) void k_ggemm(GGemm p) {
    constexpr int LS = 80;
    __shared__ __attribute__((aligned(16))) float As[16 * LS];
    __shared__ __attribute__((aligned(16))) float Bs[16 * LS];
    const int g = blockIdx.z / p.H, h = blockIdx.z - g * p.H;
    const int n_g = p.gp[g + 1] - p.gp[g];
    const int M = p.dimM ? p.dimM : n_g, N = p.dimN ? p.dimN : n_g, K = p.dimK ? p.dimK : n_g;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    if (m0 >= M || n0 >= N) return;
    int rsA, rsB, rsC;
    const float *A = op_ptr(p.A, g, h, n_g, p.gp, p.poff, rsA);
    const float *B = op_ptr(p.B, g, h, n_g, p.gp, p.poff, rsB);
    float *C = (float *)op_ptr(p.C, g, h, n_g, p.gp, p.poff, rsC);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    // staging roles.  contiguous-along-m/n ("wide"): thread = (k = tid >> 4, 4 columns (tid & 15) * 4)
    //                 contiguous-along-k ("deep"): thread = (row = tid & 63, 4 k values (tid >> 6) * 4)
    const int wk = tid >> 4, wx = (tid & 15) * 4, dr = tid & 63, dk = (tid >> 6) * 4;
    auto loadA = [&](int k0) -> f32x4 {
        if (p.transA) return ld4z(A, rsA, k0 + wk, m0 + wx, K, M);          // memory [k][m]
        return ld4z(A, rsA, m0 + dr, k0 + dk, M, K);                         // memory [m][k]
    };
    auto loadB = [&](int k0) -> f32x4 {
        if (!p.transB) return ld4z(B, rsB, k0 + wk, n0 + wx, K, N);         // memory [k][n]
        return ld4z(B, rsB, n0 + dr, k0 + dk, N, K);                         // memory [n][k]
    };
    auto put = [&](float *S, bool wide, const f32x4 &v) {
        if (wide) *(f32x4 *)(S + wk * LS + wx) = v;
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) S[(dk + e) * LS + dr] = v[e];
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra = loadA(0), rb = loadB(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        put(As, p.transA != 0, ra);
        put(Bs, p.transB == 0, rb);
        __syncthreads();
        if (k0 + 16 < K) { ra = loadA(k0 + 16); rb = loadB(k0 + 16); }
        if (BFC) {
            // one v_mfma_f32_16x16x16_bf16 per tile pair and stage: lane group g = lane >> 4 feeds the stage's k rows
            // g, g + 4, g + 8, g + 12 (both operands alike; rows one apart keep the four groups on different banks)
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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + i * 16 + 4 * (lane >> 4) + r, n = n0 + wc * 32 + j * 16 + (lane & 15);
                if (m < M && n < N) {
                    float *d = C + (size_t)m * rsC + n;
                    const float v = p.alpha * acc[i][j][r];
                    *d = p.accumulate ? *d + v : v;
                }
            }
}

// k_ggemm_small (round 4, bf16-operand mode only): the same grouped product for SMALL groups -- every dimension <= 160, i.e. the
// 12 x 12 puzzles of BASELINE configuration 5 (M, N, K in {144, 32}) -- as ONE 10-wave workgroup per (graph, head) instead of a
// 64 x 64 tile per workgroup: k_ggemm cut such a product into nine workgroups of which five are 16 / 64 full and each runs two to
// nine 16-deep stages behind a barrier pair -- launch- and latency-bound whatever the MFMA costs (46.6 -> 41.2 us when the
// operands went to bf16).  Here both operands are converted to bf16 ONCE into LDS, k-contiguous ([m][k] and [n][k], row pitch
// K + 8 elements: 8-byte fragment reads on distinct banks), and the waves walk the 16 x 16 output tiles with
// v_mfma_f32_16x16x16_bf16 straight out of LDS: no barrier after the staging.
constexpr int GS_MAX = 160;
constexpr int GS_WAVES = 10;          // ten waves: one per 16-row band of a 160-row product (nine busy at n = 144)
__global__ __launch_bounds__(64 * GS_WAVES) void k_ggemm_small(GGemm p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short gs_lds[];      // A_s [Mmax][pitch] | B_s [Nmax][pitch]
    const int GS_PITCH = p.gs_pitch;                      // K rounded up to 16, + 8 elements (the launcher sizes the LDS with it)
    unsigned short *As = gs_lds, *Bs = gs_lds + p.gs_rows_a * GS_PITCH;
    const int g = blockIdx.x / p.H, h = blockIdx.x - g * p.H;
    const int n_g = p.gp[g + 1] - p.gp[g];
    const int M = p.dimM ? p.dimM : n_g, N = p.dimN ? p.dimN : n_g, K = p.dimK ? p.dimK : n_g;
    if (M <= 0 || N <= 0) return;
    int rsA, rsB, rsC;
    const float *A = op_ptr(p.A, g, h, n_g, p.gp, p.poff, rsA);
    const float *B = op_ptr(p.B, g, h, n_g, p.gp, p.poff, rsB);
    float *C = (float *)op_ptr(p.C, g, h, n_g, p.gp, p.poff, rsC);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Mp = (M + 15) & ~15, Np = (N + 15) & ~15, Kp = (K + 15) & ~15;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
    typedef __attribute__((ext_vector_type(4))) short s16x4_;
    // stage one operand: X(r, k) for r < R (rows of the LDS image), k < K; `kmajor` = the memory is [k][r] (r contiguous)
    auto stage = [&](unsigned short *S, const float *X, int rs, bool kmajor, int R, int Rp) {
        if (!kmajor) {                      // memory [r][k]: 4 consecutive k per thread -> one 8-byte LDS store
            const int kq = Kp >> 2;
            for (int idx = tid; idx < Rp * kq; idx += 64 * GS_WAVES) {
                const int r = idx / kq, k = (idx - r * kq) * 4;
                const f32x4 v = ld4z(X, rs, r, k, R, K);
                const bf16x4_ b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *(s16x4_ *)(S + r * GS_PITCH + k) = __builtin_bit_cast(s16x4_, b);
            }
        } else {                            // memory [k][r]: 4 consecutive r of one k per thread (one 16-byte load) -> four 2-byte LDS
            const int rq = Rp >> 2;         // stores four rows apart.  (Tried: 4 consecutive k of one r per thread, four 4-byte loads -> one
            for (int idx = tid; idx < Kp * rq; idx += 64 * GS_WAVES) {      // conflict-free 8-byte store: 39.9 vs 28.8 us per launch.)
                const int k = idx / rq, r = (idx - k * rq) * 4;
                const f32x4 v = ld4z(X, rs, k, r, K, R);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const __bf16 b = (__bf16)v[e]; S[(r + e) * GS_PITCH + k] = __builtin_bit_cast(unsigned short, b); }
            }
        }
    };
    // the same from a pair matrix kept in bf16 (pair_bf16: P and dS of the dense small-group path -- they are rounded to bf16 on
    // their way into the MFMA anyway, so storing them rounded changes no product and halves the traffic these launches are bound by)
    auto stage_h = [&](unsigned short *S, const unsigned short *X, int rs, bool kmajor, int R, int Rp) {
        const int q0 = kmajor ? (Rp >> 2) : (Kp >> 2), n0 = kmajor ? Kp : Rp;
        for (int idx = tid; idx < n0 * q0; idx += 64 * GS_WAVES) {
            const int a = idx / q0, b4 = (idx - a * q0) * 4;             // memory row a, 4 consecutive columns from b4
            const int rows = kmajor ? K : R, cols = kmajor ? R : K;
            unsigned short e[4] = {0, 0, 0, 0};
            if (a < rows) {
                const unsigned short *q = X + (size_t)a * rs + b4;
                if (b4 + 3 < cols) { const u32x2 u = *(const u32x2 *)q; e[0] = u[0] & 0xffff; e[1] = u[0] >> 16; e[2] = u[1] & 0xffff; e[3] = u[1] >> 16; }
                else for (int x = 0; x < 4; ++x) if (b4 + x < cols) e[x] = q[x];
            }
            if (!kmajor) *(u32x2 *)(S + a * GS_PITCH + b4) = (u32x2){(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16)};
            else { S[(b4 + 0) * GS_PITCH + a] = e[0]; S[(b4 + 1) * GS_PITCH + a] = e[1]; S[(b4 + 2) * GS_PITCH + a] = e[2]; S[(b4 + 3) * GS_PITCH + a] = e[3]; }
        }
    };
    const bool hA = p.pair_bf16 && p.A.kind == 1, hC = p.pair_bf16 && p.C.kind == 1;
    const size_t pair_off = (size_t)p.poff[g] + (size_t)h * n_g * ((n_g + 3) & ~3);      // element offset of this group's pair matrix
    if (hA) stage_h(As, (const unsigned short *)p.A.base + pair_off, rsA, p.transA != 0, M, Mp);
    else stage(As, A, rsA, p.transA != 0, M, Mp);     // A(m, k) = opA[k][m] if transA
    stage(Bs, B, rsB, p.transB == 0, N, Np);          // B(k, n) = opB[n][k] if transB (k contiguous), else memory [k][n]
    __syncthreads();
    const int tn_n = Np >> 4, tm_n = Mp >> 4;
    if (p.epi == 0) {                                   // (node-matrix outputs only: pair matrices leave through the row epilogues)
        for (int t = wid; t < tm_n * tn_n; t += GS_WAVES) {
            const int tm = t / tn_n, tn = t - tm * tn_n;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const unsigned short *ap = As + (tm * 16 + (lane & 15)) * GS_PITCH + 4 * (lane >> 4);
            const unsigned short *bp = Bs + (tn * 16 + (lane & 15)) * GS_PITCH + 4 * (lane >> 4);
            for (int k0 = 0; k0 < Kp; k0 += 16)
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*(const s16x4_ *)(ap + k0), *(const s16x4_ *)(bp + k0), acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = tm * 16 + 4 * (lane >> 4) + r, n = tn * 16 + (lane & 15);
                if (m < M && n < N) {
                    float *d = C + (size_t)m * rsC + n;
                    const float v = p.alpha * acc[r];
                    *d = p.accumulate ? *d + v : v;
                }
            }
        }
        return;
    }
    // Row epilogues (pair-matrix outputs, N = n_g <= 160): a wave owns WHOLE 16-row bands of the product -- ten 16 x 16 tiles in
    // registers -- so a row's N values sit in the 16 lanes of one lane group (lane & 15 = column inside a tile) times the tiles:
    // row max / sums are four xor-shuffles + a loop over the tiles, and the score / dP matrices never travel to HBM and back
    // through a separate row kernel (k_pair_rows read and wrote 84 - 126 MB per layer at configuration 5).
    constexpr int TN_MAX = GS_MAX / 16;
    const float *E = nullptr;
    if (p.epi == 2) { int rsE; GOp eo = p.C; eo.base = (float *)p.E; E = op_ptr(eo, g, h, n_g, p.gp, p.poff, rsE); }
    for (int tm = wid; tm < tm_n; tm += GS_WAVES) {
        f32x4 acc[TN_MAX];
        const unsigned short *ap = As + (tm * 16 + (lane & 15)) * GS_PITCH + 4 * (lane >> 4);
#pragma unroll
        for (int tn = 0; tn < TN_MAX; ++tn) {
            acc[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tn < tn_n) {
                const unsigned short *bp = Bs + (tn * 16 + (lane & 15)) * GS_PITCH + 4 * (lane >> 4);
                for (int k0 = 0; k0 < Kp; k0 += 16)
                    acc[tn] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*(const s16x4_ *)(ap + k0), *(const s16x4_ *)(bp + k0), acc[tn], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = tm * 16 + 4 * (lane >> 4) + r;                 // this lane's row (shared by its 16-lane group)
            const bool rowok = m < M;
            if (p.epi == 1) {
                float mx = -INFINITY;
#pragma unroll
                for (int tn = 0; tn < TN_MAX; ++tn) {
                    const int n = tn * 16 + (lane & 15);
                    const bool ok = tn < tn_n && n < N && !(p.nodiag && n == m);
                    acc[tn][r] = ok ? p.alpha * acc[tn][r] : -INFINITY;
                    mx = fmaxf(mx, acc[tn][r]);
                }
                for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                float sum = 0.f;
#pragma unroll
                for (int tn = 0; tn < TN_MAX; ++tn) {
                    acc[tn][r] = (mx > -INFINITY) ? expf(acc[tn][r] - mx) : 0.f;      // exp(-inf) = 0 for the excluded entries
                    sum += acc[tn][r];
                }
                for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                const float inv = (mx > -INFINITY) ? 1.0f / (sum + 1e-16f) : 0.f;
#pragma unroll
                for (int tn = 0; tn < TN_MAX; ++tn) {
                    const int n = tn * 16 + (lane & 15);
                    if (rowok && tn < tn_n && n < N) {
                        if (hC) ((unsigned short *)p.C.base + pair_off)[(size_t)m * rsC + n] = f2bf(acc[tn][r] * inv);
                        else C[(size_t)m * rsC + n] = acc[tn][r] * inv;
                    }
                }
            } else {
                float pe[TN_MAX], D = 0.f;
#pragma unroll
                for (int tn = 0; tn < TN_MAX; ++tn) {
                    const int n = tn * 16 + (lane & 15);
                    const bool ok = rowok && tn < tn_n && n < N;
                    pe[tn] = !ok ? 0.f : (p.pair_bf16 ? bf2f(((const unsigned short *)p.E + pair_off)[(size_t)m * rsC + n]) : E[(size_t)m * rsC + n]);
                    acc[tn][r] = ok ? p.alpha * acc[tn][r] : 0.f;
                    D = fmaf(pe[tn], acc[tn][r], D);
                }
                for (int o = 8; o > 0; o >>= 1) D += __shfl_xor(D, o);
#pragma unroll
                for (int tn = 0; tn < TN_MAX; ++tn) {
                    const int n = tn * 16 + (lane & 15);
                    if (rowok && tn < tn_n && n < N) {
                        if (hC) ((unsigned short *)p.C.base + pair_off)[(size_t)m * rsC + n] = f2bf(pe[tn] * (acc[tn][r] - D));
                        else C[(size_t)m * rsC + n] = pe[tn] * (acc[tn][r] - D);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Small groups, bf16-operand mode, second step (round 4): the WHOLE attention of a (graph, head) group in one workgroup, forward
// and backward, flash-style -- no pair matrix ever leaves the CU.  k_ggemm_small above still ran the layer as 2 + 4 launches of
// 30 - 50 us that each staged their operands again and carried P / dS through HBM; here K, V (then Q, dO) of the group are
// converted to bf16 into LDS once, row-major, and everything else happens in registers:
//   forward   a wave owns a 16-query band; its Q rows go from HBM straight into B fragments.  S^T tiles (A = K rows, B = Q rows:
//             the accumulator then holds, per lane, four consecutive KEYS of one query -- the A-operand layout of the next
//             product), row softmax over registers + two shuffles, O = P V with V fragments fetched by ds_read_b64_tr_b16 (the
//             transposition happens in the LDS read: no transposed image is ever built), epilogue adds the skip projection
//             (+ residual): k_init_out is gone too;
//   backward  phase A, wave = query band (Q, dO bands in registers; K, V in LDS): S^T and dP^T tiles, the softmax statistics
//             recomputed (the forward keeps nothing), D_i = sum_j P dP, dS -> dQ = scale dS K (K by transposing reads); (max,
//             1 / sum, D) of the band go to LDS.  Then Q, dO replace K, V in LDS.  Phase B, wave = key band (K, V bands in
//             registers): the same scores in the other orientation (rows = queries), P^T and dS^T as A operands, dV = P^T dO,
//             dK = scale dS^T Q.  The skip gradient (a copy of dO) is written while the dO bands are fetched: k_copy_skip_grad
//             is gone as well.
// Products and roundings are those of the k_ggemm_small route (operands bf16, accumulation fp32, P and dS rounded to bf16 as
// operands).  CT = 16-wide channel tiles (C <= 16 CT, zero-padded); limits: n_g <= 160, C <= 160.  DA_ATTN_SMALL_FUSED=0 keeps
// the grouped-GEMM route.
constexpr int AS_MAXN = 160, AS_WAVES = 10, AS_TN = AS_MAXN / 16;
struct AttnSmall {
    const void *qkvs;            // [n, 4 HC]  Q | K | V | skip   (fp32; bf16 in the q16 instances)
    const float *res;            // forward: residual [n, HC] or null
    float *o;                    // forward: [n, HC]
    const float *d_o;            // backward: [n, HC]
    void *dY4;                   // backward: [n, 4 HC]  dq | dk | dv | d_o   (fp32; bf16 in the q16 instances)
    const int32_t *gp;
    int H, C, HC, nodiag, np;    // np: max_graph_nodes rounded up to 16 (sizes the LDS images)
    float scale;
};
typedef __attribute__((ext_vector_type(4))) __bf16 as_bf16x4;
typedef __attribute__((ext_vector_type(4))) short as_s16x4;

__device__ __forceinline__ as_s16x4 as_pack(float a, float b, float c, float d) {
    const as_bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return __builtin_bit_cast(as_s16x4, v);
}
// Q16 = the projection buffer / its gradient are bf16 (q16 mode of da_train.hip): four consecutive elements at element offset e
template <bool Q16> __device__ __forceinline__ as_s16x4 as_ld_pack(const void *base, size_t e) {
    if (Q16) return *(const as_s16x4 *)((const unsigned short *)base + e);
    const f32x4 v = *(const f32x4 *)((const float *)base + e);
    return as_pack(v[0], v[1], v[2], v[3]);
}
template <bool Q16> __device__ __forceinline__ f32x4 as_ld4(const void *base, size_t e) {
    if (Q16) {
        const u32x2 u = *(const u32x2 *)((const unsigned short *)base + e);
        return (f32x4){bf2f((bf16_t)(u[0] & 0xffff)), bf2f((bf16_t)(u[0] >> 16)), bf2f((bf16_t)(u[1] & 0xffff)), bf2f((bf16_t)(u[1] >> 16))};
    }
    return *(const f32x4 *)((const float *)base + e);
}
template <bool Q16> __device__ __forceinline__ void as_st4(void *base, size_t e, const f32x4 &v) {
    if (Q16) *(as_s16x4 *)((unsigned short *)base + e) = as_pack(v[0], v[1], v[2], v[3]);
    else *(f32x4 *)((float *)base + e) = v;
}
// rows [0, np) x 16 CT columns of a node matrix (memory [row][ld] from element offset e0, C valid columns, n_g valid rows) as a
// row-major bf16 image of pitch 16 CT + 8, zero beyond the valid part
template <int CT, bool Q16>
__device__ __forceinline__ void as_stage(const void *X, size_t e0, int ld, int n_g, int np, int C, unsigned short *R, int tid) {
    constexpr int kq = CT * 4, pr = CT * 16 + 8, NB = CT >= 4 ? 4 : 2;      // NB loads in flight per thread before the first LDS store
    for (int i0 = tid; i0 < np * kq; i0 += 64 * AS_WAVES * NB) {
        as_s16x4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = i0 + u * 64 * AS_WAVES, r = idx / kq, k = (idx - r * kq) * 4;
            v[u] = (as_s16x4){0, 0, 0, 0};
            if (r < n_g && k < C) v[u] = as_ld_pack<Q16>(X, e0 + (size_t)r * ld + k);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = i0 + u * 64 * AS_WAVES, r = idx / kq, k = (idx - r * kq) * 4;
            if (idx < np * kq) *(as_s16x4 *)(R + r * pr + k) = v[u];
        }
    }
}
// NI images staged in ONE load round trip (narrow heads: np * kq <= 2 * 640 pieces per image): every image's pieces are requested
// before the first LDS store (as_stage per image = one global-load round trip per image, the stores of image k waiting in front
// of the loads of image k + 1)
template <int CT, bool Q16, int NI>
__device__ __forceinline__ void as_stage_multi(const void *X, const size_t (&e0)[NI], int ld, int n_g, int np, int C,
                                               unsigned short *const (&R)[NI], int tid) {
    constexpr int kq = CT * 4, pr = CT * 16 + 8, NB = 2;
    for (int i0 = tid; i0 < np * kq; i0 += 64 * AS_WAVES * NB) {
        as_s16x4 v[NI][NB];
#pragma unroll
        for (int im = 0; im < NI; ++im)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int idx = i0 + u * 64 * AS_WAVES, r = idx / kq, k = (idx - r * kq) * 4;
                v[im][u] = (as_s16x4){0, 0, 0, 0};
                if (r < n_g && k < C) v[im][u] = as_ld_pack<Q16>(X, e0[im] + (size_t)r * ld + k);
            }
#pragma unroll
        for (int im = 0; im < NI; ++im)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int idx = i0 + u * 64 * AS_WAVES, r = idx / kq, k = (idx - r * kq) * 4;
                if (idx < np * kq) *(as_s16x4 *)(R[im] + r * pr + k) = v[im][u];
            }
    }
}
// the 16 rows [r0, r0 + 16) as B (or A) fragments straight from memory: lane (l15, lg) takes row r0 + l15, columns 16 kt + 4 lg ..+3
template <int CT, bool Q16>
__device__ __forceinline__ void as_band(const void *X, size_t e0, int ld, int n_g, int C, int r0, int l15, int lg, as_s16x4 (&f)[CT]) {
    const int r = r0 + l15;
#pragma unroll
    for (int kt = 0; kt < CT; ++kt) {
        const int k = kt * 16 + 4 * lg;
        f[kt] = (as_s16x4){0, 0, 0, 0};
        if (r < n_g && k < C) f[kt] = as_ld_pack<Q16>(X, e0 + (size_t)r * ld + k);
    }
}
// B fragment (k = rows r0 + 4 lg ..+3, n = column c0 + l15) of a row-major image by a transposing read: within a 16-lane group
// lane i' supplies the address of row (i' >> 2), columns 4 (i' & 3) ..+3 and receives column i' of the four rows
__device__ __forceinline__ as_s16x4 as_tr(const unsigned short *img, int pr, int r0, int c0, int l15, int lg) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) as_s16x4 *)(img + (r0 + 4 * lg + (l15 >> 2)) * pr + c0 + 4 * (l15 & 3)));
}
#define AS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)

// e^x as one v_exp_f32 (2^(x log2 e): ~2 ulp; libm's expf costs a dozen vector instructions per call and these kernels issue ~80 per
// lane -- they only run in the bf16-operand mode, where P is rounded to 8 bits of mantissa right after)
__device__ __forceinline__ float as_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// scores of one query band against every key, transposed tiles: s[tn][r] = score(query i, key 16 tn + 4 lg + r); softmax over
// the band's rows in place (s becomes exp(s - max)); returns (max, 1 / (sum + 1e-16)) of this lane's query (PyG's softmax)
__device__ __forceinline__ void as_softmax(f32x4 (&s)[AS_TN], int tn_n, int n_g, int i, int lg, float scale, int nodiag,
                                           float &mx, float &inv) {
    mx = -INFINITY;
#pragma unroll
    for (int tn = 0; tn < AS_TN; ++tn) {
        // (wave-uniform cases: a tile beyond the graph; the last tile or a missing diagonal -- per-key guards; everything else is
        //  a plain scale.  The guards on every element were ~240 of a band's ~4 000 instructions)
        if (tn >= tn_n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[tn][r] = -INFINITY;
        } else if (tn == tn_n - 1 || nodiag) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = tn * 16 + 4 * lg + r;
                const bool ok = j < n_g && !(nodiag && j == i);
                s[tn][r] = ok ? scale * s[tn][r] : -INFINITY;
                mx = fmaxf(mx, s[tn][r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[tn][r] = scale * s[tn][r];
                mx = fmaxf(mx, s[tn][r]);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int tn = 0; tn < AS_TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[tn][r] = (mx > -INFINITY) ? as_exp(s[tn][r] - mx) : 0.f;
            sum += s[tn][r];
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    inv = (mx > -INFINITY) ? 1.0f / (sum + 1e-16f) : 0.f;
}

template <int CT, bool Q16>
__global__ __launch_bounds__(64 * AS_WAVES) void k_attn_small_fwd(AttnSmall p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short as_lds[];
    const int g = blockIdx.x / p.H, h = blockIdx.x - g * p.H;
    const int n0 = p.gp[g], n_g = p.gp[g + 1] - n0;
    if (n_g <= 0) return;
    constexpr int pr = CT * 16 + 8;
    const int C = p.C, np = p.np, ld = 4 * p.HC;
    unsigned short *Ks = as_lds, *Vs = Ks + np * pr;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const size_t e0 = (size_t)n0 * ld + h * C;                         // element offset of (node n0, head h) in the projection buffer
    const int tn_n = (n_g + 15) >> 4;
    as_s16x4 qf[CT];
    if (wid < tn_n) as_band<CT, Q16>(p.qkvs, e0, ld, n_g, C, wid * 16, l15, lg, qf);       // (requested first: in flight under the staging)
    if constexpr (CT <= 2) {
        const size_t es[2] = {e0 + p.HC, e0 + 2 * p.HC};
        unsigned short *const im[2] = {Ks, Vs};
        as_stage_multi<CT, Q16, 2>(p.qkvs, es, ld, n_g, np, C, im, tid);
    } else {
        as_stage<CT, Q16>(p.qkvs, e0 + p.HC, ld, n_g, np, C, Ks, tid);
        as_stage<CT, Q16>(p.qkvs, e0 + 2 * p.HC, ld, n_g, np, C, Vs, tid);
    }
    __syncthreads();
    for (int tm = wid; tm < tn_n; tm += AS_WAVES) {                    // (one band per wave: n_g <= 160)
        const int i = tm * 16 + l15;                                   // this lane's query
        f32x4 acc[AS_TN];
#pragma unroll
        for (int tn = 0; tn < AS_TN; ++tn) {
            acc[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tn < tn_n) {
                const unsigned short *kp = Ks + (tn * 16 + l15) * pr + 4 * lg;
#pragma unroll
                for (int kt = 0; kt < CT; ++kt) acc[tn] = AS_MFMA(*(const as_s16x4 *)(kp + kt * 16), qf[kt], acc[tn]);
            }
        }
        float mx, inv;
        as_softmax(acc, tn_n, n_g, i, lg, p.scale, p.nodiag, mx, inv);
        f32x4 oacc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) oacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tn = 0; tn < AS_TN; ++tn) {
            if (tn < tn_n) {
                const as_s16x4 pa = as_pack(acc[tn][0] * inv, acc[tn][1] * inv, acc[tn][2] * inv, acc[tn][3] * inv);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) oacc[ct] = AS_MFMA(as_tr(Vs, pr, tn * 16, ct * 16, l15, lg), pa, oacc[ct]);
            }
        }
        // O^T tiles (A = V^T by the transposing read, B = P^T): a lane holds four consecutive CHANNELS of its query, so the skip
        // projection, the residual and o move as 16-byte (8-byte bf16) pieces, all loads issued before the first use
        // (channel offsets beyond C are clamped to 0 instead of guarded: guarded loads compile to load -> wait round trips)
        if (i < n_g) {
            const size_t es = e0 + 3 * p.HC + (size_t)i * ld;
            const float *rsp = p.res ? p.res + (size_t)(n0 + i) * p.HC + h * C : nullptr;
            float *ob = p.o + (size_t)(n0 + i) * p.HC + h * C + 4 * lg;
            // (144- and 160-wide heads: in two halves -- nine / ten skip + residual + output tiles at once are ~110 registers beside the band's
            //  state under this kernel's 168, which spilled 12 - 26 of them to scratch)
            constexpr int CH = CT > 8 ? (CT + 1) / 2 : CT;
#pragma unroll
            for (int c0 = 0; c0 < CT; c0 += CH) {
                f32x4 sk[CH], rs[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int ct = c0 + k;
                    if (ct < CT) sk[k] = as_ld4<Q16>(p.qkvs, es + (ct * 16 + 4 * lg < C ? ct * 16 + 4 * lg : 0));
                }
                if (rsp) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int ct = c0 + k;
                        if (ct < CT) rs[k] = *(const f32x4 *)(rsp + (ct * 16 + 4 * lg < C ? ct * 16 + 4 * lg : 0));
                    }
#pragma unroll
                    for (int k = 0; k < CH; ++k)
                        if (c0 + k < CT) sk[k] += rs[k];
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int ct = c0 + k;
                    if (ct < CT && ct * 16 + 4 * lg < C) *(f32x4 *)(ob + ct * 16) = oacc[ct] + sk[k];
                }
            }
        }
    }
}

template <int CT, bool Q16, bool A4>
__global__ __launch_bounds__(64 * AS_WAVES) void k_attn_small_bwd(AttnSmall p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short as_lds[];
    const int g = blockIdx.x / p.H, h = blockIdx.x - g * p.H;
    const int n0 = p.gp[g], n_g = p.gp[g + 1] - n0;
    if (n_g <= 0) return;
    constexpr int pr = CT * 16 + 8;
    const int C = p.C, np = p.np, ld = 4 * p.HC;
    // ALL4 (narrow heads: four images are 46 KB at n = 144): K, V, Q AND dO are staged at once -- one load round trip instead of
    // two, no band loads from HBM (the bands are rows of the images), no second staging pass between the phases
    constexpr bool ALL4 = A4;
    unsigned short *I0 = as_lds, *I1 = I0 + np * pr;                  // phase A: K, V ; phase B: Q, dO (ALL4: K, V stay, Q, dO in J0, J1)
    unsigned short *J0 = ALL4 ? I1 + np * pr : I0, *J1 = ALL4 ? J0 + np * pr : I1;
    float *st_m = (float *)(J1 + np * pr), *st_inv = st_m + np, *st_D = st_inv + np;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const size_t e0 = (size_t)n0 * ld + h * C;                         // (node n0, head h) in the projection buffer and in dY4
    const size_t g0 = (size_t)n0 * p.HC + h * C;                       // ... in d_o
    const int tn_n = (n_g + 15) >> 4;
    if constexpr (ALL4) {
        // the dO pieces are requested first and the three projection images behind them: ONE load round trip for all four images
        // (np * kq <= 2 * 640 pieces per image at CT <= 2)
        constexpr int kq = CT * 4;
        static_assert(AS_MAXN * kq <= 2 * 64 * AS_WAVES, "two dO pieces per thread");
        f32x4 dv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * 64 * AS_WAVES, r = idx / kq, k = (idx - r * kq) * 4;
            dv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r < n_g && k < C) dv[u] = *(const f32x4 *)(p.d_o + g0 + (size_t)r * p.HC + k);
        }
        const size_t es[3] = {e0 + p.HC, e0 + 2 * p.HC, e0};
        unsigned short *const im[3] = {I0, I1, J0};
        as_stage_multi<CT, Q16, 3>(p.qkvs, es, ld, n_g, np, C, im, tid);
        // dO image + its copy = the skip projection's gradient (columns 3 HC .. of dY4)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * 64 * AS_WAVES, r = idx / kq, k = (idx - r * kq) * 4;
            if (idx < np * kq) {
                if (r < n_g && k < C) as_st4<Q16>(p.dY4, e0 + 3 * p.HC + (size_t)r * ld + k, dv[u]);
                *(as_s16x4 *)(J1 + r * pr + k) = as_pack(dv[u][0], dv[u][1], dv[u][2], dv[u][3]);
            }
        }
    } else {
        as_stage<CT, Q16>(p.qkvs, e0 + p.HC, ld, n_g, np, C, I0, tid);
        as_stage<CT, Q16>(p.qkvs, e0 + 2 * p.HC, ld, n_g, np, C, I1, tid);
    }
    // ---- phase A: query bands (one per wave)
    {
        as_s16x4 qf[CT], gf[CT];
        if (ALL4) {
            __syncthreads();
            if (wid < tn_n) {
#pragma unroll
                for (int kt = 0; kt < CT; ++kt) {
                    qf[kt] = *(const as_s16x4 *)(J0 + (wid * 16 + l15) * pr + kt * 16 + 4 * lg);
                    gf[kt] = *(const as_s16x4 *)(J1 + (wid * 16 + l15) * pr + kt * 16 + 4 * lg);
                }
            }
        } else if (wid < tn_n) {
            as_band<CT, Q16>(p.qkvs, e0, ld, n_g, C, wid * 16, l15, lg, qf);
            // dO band, and its copy = the skip projection's gradient (columns 3 HC .. of dY4)
            const int r = wid * 16 + l15;
#pragma unroll
            for (int kt = 0; kt < CT; ++kt) {
                const int k = kt * 16 + 4 * lg;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < n_g && k < C) {
                    v = *(const f32x4 *)(p.d_o + g0 + (size_t)r * p.HC + k);
                    as_st4<Q16>(p.dY4, e0 + 3 * p.HC + (size_t)r * ld + k, v);
                }
                gf[kt] = as_pack(v[0], v[1], v[2], v[3]);
            }
        }
        if (!ALL4) __syncthreads();
        if (wid < tn_n) {
            const int tm = wid, i = tm * 16 + l15;
            f32x4 sa[AS_TN], da[AS_TN];
#pragma unroll
            for (int tn = 0; tn < AS_TN; ++tn) {
                sa[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
                da[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (tn < tn_n) {
                    const unsigned short *kp = I0 + (tn * 16 + l15) * pr + 4 * lg, *vp = I1 + (tn * 16 + l15) * pr + 4 * lg;
#pragma unroll
                    for (int kt = 0; kt < CT; ++kt) {
                        sa[tn] = AS_MFMA(*(const as_s16x4 *)(kp + kt * 16), qf[kt], sa[tn]);
                        da[tn] = AS_MFMA(*(const as_s16x4 *)(vp + kt * 16), gf[kt], da[tn]);
                    }
                }
            }
            float mx, inv;
            as_softmax(sa, tn_n, n_g, i, lg, p.scale, p.nodiag, mx, inv);
            float D = 0.f;
#pragma unroll
            for (int tn = 0; tn < AS_TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // P enters the products rounded to bf16 (as the kept P of the grouped-GEMM route was); D uses the same values
                    sa[tn][r] = bf2f(f2bf(sa[tn][r] * inv));
                    D = fmaf(sa[tn][r], da[tn][r], D);
                }
            D += __shfl_xor(D, 16);
            D += __shfl_xor(D, 32);
            if (lg == 0) { st_m[i] = mx; st_inv[i] = inv; st_D[i] = D; }
            f32x4 qacc[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) qacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tn = 0; tn < AS_TN; ++tn) {
                if (tn < tn_n) {
                    const as_s16x4 dsa = as_pack(sa[tn][0] * (da[tn][0] - D), sa[tn][1] * (da[tn][1] - D),
                                                 sa[tn][2] * (da[tn][2] - D), sa[tn][3] * (da[tn][3] - D));
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) qacc[ct] = AS_MFMA(as_tr(I0, pr, tn * 16, ct * 16, l15, lg), dsa, qacc[ct]);
                }
            }
            if (i < n_g) {                                            // dQ^T tiles: four consecutive channels of query i per lane
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    if (ct * 16 + 4 * lg < C) as_st4<Q16>(p.dY4, e0 + (size_t)i * ld + ct * 16 + 4 * lg, p.scale * qacc[ct]);
            }
        }
    }
    __syncthreads();
    // ---- phase B: key bands (one per wave)
    as_s16x4 kf[CT], vf[CT];
    if constexpr (ALL4) {
        if (wid < tn_n) {
#pragma unroll
            for (int kt = 0; kt < CT; ++kt) {
                kf[kt] = *(const as_s16x4 *)(I0 + (wid * 16 + l15) * pr + kt * 16 + 4 * lg);
                vf[kt] = *(const as_s16x4 *)(I1 + (wid * 16 + l15) * pr + kt * 16 + 4 * lg);
            }
        }
    } else {
        as_stage<CT, Q16>(p.qkvs, e0, ld, n_g, np, C, I0, tid);
        as_stage<CT, false>(p.d_o, g0, p.HC, n_g, np, C, I1, tid);
        if (wid < tn_n) {
            as_band<CT, Q16>(p.qkvs, e0 + p.HC, ld, n_g, C, wid * 16, l15, lg, kf);
            as_band<CT, Q16>(p.qkvs, e0 + 2 * p.HC, ld, n_g, C, wid * 16, l15, lg, vf);
        }
        __syncthreads();
    }
    if (wid < tn_n) {
        const int tj = wid, j = tj * 16 + l15;                         // this lane's key
        f32x4 vacc[CT], kacc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { vacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; kacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        for (int ti = 0; ti < tn_n; ++ti) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
            const unsigned short *qp = J0 + (ti * 16 + l15) * pr + 4 * lg, *gpp = J1 + (ti * 16 + l15) * pr + 4 * lg;
#pragma unroll
            for (int kt = 0; kt < CT; ++kt) {
                s4 = AS_MFMA(*(const as_s16x4 *)(qp + kt * 16), kf[kt], s4);
                d4 = AS_MFMA(*(const as_s16x4 *)(gpp + kt * 16), vf[kt], d4);
            }
            const f32x4 m4 = *(const f32x4 *)(st_m + ti * 16 + 4 * lg), i4 = *(const f32x4 *)(st_inv + ti * 16 + 4 * lg),
                        D4 = *(const f32x4 *)(st_D + ti * 16 + 4 * lg);
            float pv[4], dsv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + 4 * lg + r;
                const bool ok = i < n_g && j < n_g && !(p.nodiag && j == i) && m4[r] > -INFINITY;
                pv[r] = ok ? bf2f(f2bf(as_exp(p.scale * s4[r] - m4[r]) * i4[r])) : 0.f;
                dsv[r] = pv[r] * (d4[r] - D4[r]);
            }
            const as_s16x4 pa = as_pack(pv[0], pv[1], pv[2], pv[3]), dsa = as_pack(dsv[0], dsv[1], dsv[2], dsv[3]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                vacc[ct] = AS_MFMA(as_tr(J1, pr, ti * 16, ct * 16, l15, lg), pa, vacc[ct]);
                kacc[ct] = AS_MFMA(as_tr(J0, pr, ti * 16, ct * 16, l15, lg), dsa, kacc[ct]);
            }
        }
        if (j < n_g) {                                                // dK^T, dV^T tiles: four consecutive channels of key j per lane
            const size_t er = e0 + (size_t)j * ld + 4 * lg;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (ct * 16 + 4 * lg < C) {
                    as_st4<Q16>(p.dY4, er + p.HC + ct * 16, p.scale * kacc[ct]);
                    as_st4<Q16>(p.dY4, er + 2 * p.HC + ct * 16, vacc[ct]);
                }
        }
    }
}
#undef AS_MFMA

static int as_tiles(int C) { const int ct = (C + 15) / 16; return ct <= 1 ? 1 : ct <= 2 ? 2 : ct <= 4 ? 4 : ct <= 9 ? 9 : 10; }
bool attn_small_ok(const da_graph *g, int C, bool bfc) {
    static int off = -1;
    if (off < 0) off = DA_XENV("DA_ATTN_SMALL_FUSED", 1) == 0 ? 1 : 0;
    return bfc && !off && g->max_graph_nodes <= AS_MAXN && C <= 160 && C % 4 == 0;
}
template <int CT, bool Q16>
static int attn_small_launch_ct(const da_graph *g, AttnSmall &a, bool bwd, hipStream_t st) {
    constexpr int pr = CT * 16 + 8;
    static int a4_off = -1;
    if (a4_off < 0) a4_off = DA_XENV("DA_ATTN_SMALL_ALL4", 1) == 0 ? 1 : 0;
    const bool a4 = CT <= 2 && !a4_off;                                   // narrow-head backward: four images, one staging pass
    const int lds = ((bwd && a4) ? 4 : 2) * a.np * pr * 2 + (bwd ? 3 * a.np * 4 : 0);
    static bool attr = false;
    if (!attr) {
        const int cap = (CT <= 2 ? 4 : 2) * AS_MAXN * pr * 2 + 3 * AS_MAXN * 4;
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_small_bwd<CT, Q16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        if (CT <= 2) DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_small_bwd<CT, Q16, CT <= 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_small_fwd<CT, Q16>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        attr = true;
    }
    if (bwd && a4) k_attn_small_bwd<CT, Q16, CT <= 2><<<g->n_graphs * a.H, 64 * AS_WAVES, lds, st>>>(a);
    else if (bwd) k_attn_small_bwd<CT, Q16, false><<<g->n_graphs * a.H, 64 * AS_WAVES, lds, st>>>(a);
    else k_attn_small_fwd<CT, Q16><<<g->n_graphs * a.H, 64 * AS_WAVES, lds, st>>>(a);
    DA_LAUNCH_CHECK();
    return 0;
}
template <bool Q16>
static int attn_small_launch_q(const da_graph *g, int C, AttnSmall &a, bool bwd, hipStream_t st) {
    switch (as_tiles(C)) {
    case 1: return attn_small_launch_ct<1, Q16>(g, a, bwd, st);
    case 2: return attn_small_launch_ct<2, Q16>(g, a, bwd, st);
    case 4: return attn_small_launch_ct<4, Q16>(g, a, bwd, st);
    case 9: return attn_small_launch_ct<9, Q16>(g, a, bwd, st);
    default: return attn_small_launch_ct<10, Q16>(g, a, bwd, st);
    }
}
static int attn_small_launch(const da_graph *g, int H, int C, AttnSmall &a, bool bwd, bool q16, hipStream_t st) {
    a.gp = g->graph_ptr; a.H = H; a.C = C; a.HC = H * C; a.nodiag = g->dense == 2;
    a.np = (g->max_graph_nodes + 15) & ~15;
    a.scale = 1.0f / sqrtf((float)C);
    return q16 ? attn_small_launch_q<true>(g, C, a, bwd, st) : attn_small_launch_q<false>(g, C, a, bwd, st);
}

// poff[g] = sum_{g' < g} H * n_g' * round4(n_g')   (one thread; G is a few hundred at most)
__global__ void k_pair_offsets(int G, int H, const int32_t *__restrict__ gp, long long *__restrict__ poff) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        long long o = 0;
        for (int g = 0; g < G; ++g) {
            poff[g] = o;
            const long long n = gp[g + 1] - gp[g];
            o += (long long)H * n * ((n + 3) & ~3LL);
        }
        poff[G] = o;
    }
}

// rows of the pair matrices: one wave per (g, h, i).  mode 0: softmax in place (scores already scaled);
// mode 1: dS = P o (dP - sum_j P dP) written over dP.
__global__ __launch_bounds__(256) void k_pair_rows(int mode, int n_nodes, int H, int nodiag, const int32_t *__restrict__ gp,
                                                   const int32_t *__restrict__ node_graph, const long long *__restrict__ poff,
                                                   float *P, float *dP) {
    const int lane = threadIdx.x & 63;
    // grid-stride over the (node, head) rows: the launch caps the grid (gridsz), a Batch has more rows than that cap
    // covers from 4 097 nodes on (the first version returned for them and left raw scores in P)
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long wv = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wv < (long long)n_nodes * H; wv += n_waves) {
    const int node = (int)(wv / H), h = (int)(wv - (long long)node * H);
    const int g = node_graph[node], n_g = gp[g + 1] - gp[g], i = node - gp[g];
    const int ldp = (n_g + 3) & ~3;
    float *row = P + poff[g] + ((size_t)h * n_g + i) * ldp;
    if (mode == 0) {
        float m = -INFINITY;
        for (int j = lane; j < n_g; j += 64)
            if (!(nodiag && j == i)) m = fmaxf(m, row[j]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float s = 0.f;
        for (int j = lane; j < n_g; j += 64)
            if (!(nodiag && j == i)) s += expf(row[j] - m);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float inv = (m > -INFINITY) ? 1.0f / (s + 1e-16f) : 0.f;
        for (int j = lane; j < n_g; j += 64) row[j] = (nodiag && j == i) ? 0.f : expf(row[j] - m) * inv;
    } else {
        float *drow = dP + poff[g] + ((size_t)h * n_g + i) * ldp;
        float D = 0.f;
        for (int j = lane; j < n_g; j += 64) D = fmaf(row[j], drow[j], D);
        for (int o = 32; o > 0; o >>= 1) D += __shfl_xor(D, o);
        for (int j = lane; j < n_g; j += 64) drow[j] = row[j] * (drow[j] - D);
    }
    }
}

// o[i, :] = skip_i (+ residual_i): the accumulate target of O = P V
__global__ __launch_bounds__(256) void k_init_out(int n, int HC, const float *__restrict__ qkvs, const float *__restrict__ res,
                                                  float *__restrict__ o) {
    const size_t total = (size_t)n * HC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / HC, c = idx - r * HC;
        float v = qkvs[r * 4 * HC + 3 * (size_t)HC + c];
        if (res) v += res[idx];
        o[idx] = v;
    }
}

// dY4[i, 3HC:4HC] = d_o[i, :]  (gradient of the skip projection)
__global__ __launch_bounds__(256) void k_copy_skip_grad(int n, int HC, const float *__restrict__ d_o, float *__restrict__ dY4) {
    const size_t total = (size_t)n * HC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / HC, c = idx - r * HC;
        dY4[r * 4 * HC + 3 * (size_t)HC + c] = d_o[idx];
    }
}

__global__ __launch_bounds__(256) void k_node_graph(int G, const int32_t *__restrict__ gp, int32_t *__restrict__ node_graph) {
    const int g = blockIdx.x;
    if (g >= G) return;
    for (int i = gp[g] + threadIdx.x; i < gp[g + 1]; i += blockDim.x) node_graph[i] = g;
}

static unsigned gridsz(size_t n) { const size_t b = (n + 255) / 256; return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// the one-workgroup-per-group kernel takes a product (bf16-operand mode, every dimension <= 160; DA_GGEMM_SMALL=0: never)
static bool ggemm_small_ok(const GGemm &p, int maxn) {
    const int Mx = p.dimM ? p.dimM : maxn, Nx = p.dimN ? p.dimN : maxn, Kx = p.dimK ? p.dimK : maxn;
    static int small_off = -1;
    if (small_off < 0) small_off = DA_XENV("DA_GGEMM_SMALL", 1) == 0 ? 1 : 0;
    return p.bfc && !small_off && Mx <= GS_MAX && Nx <= GS_MAX && Kx <= GS_MAX;
}
static int ggemm(const GGemm &p, int G, int H, int maxn, hipStream_t st) {
    const int Mx = p.dimM ? p.dimM : maxn, Nx = p.dimN ? p.dimN : maxn;
    if (ggemm_small_ok(p, maxn)) {
        const int Kx = p.dimK ? p.dimK : maxn;
        GGemm q = p;
        q.gs_pitch = ((Kx + 15) & ~15) + 8;
        q.gs_rows_a = (Mx + 15) & ~15;
        const int lds = (q.gs_rows_a + ((Nx + 15) & ~15)) * q.gs_pitch * 2;           // 23 KB (K = 32) ... 97 KB (three 160s)
        static bool attr = false;
        if (!attr) { DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_ggemm_small, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GS_MAX * (GS_MAX + 8) * 2)); attr = true; }
        k_ggemm_small<<<G * H, 64 * GS_WAVES, lds, st>>>(q);
        DA_LAUNCH_CHECK();
        return 0;
    }
    if (p.epi) { set_error("grouped GEMM: row epilogues exist in the small-group kernel only"); return 1; }
    if (p.bfc) k_ggemm<true><<<dim3((Nx + 63) / 64, (Mx + 63) / 64, G * H), 256, 0, st>>>(p);
    else k_ggemm<false><<<dim3((Nx + 63) / 64, (Mx + 63) / 64, G * H), 256, 0, st>>>(p);
    DA_LAUNCH_CHECK();
    return 0;
}

size_t dense_pair_floats(const da_graph *g, int H) {
    // upper bound without reading the device graph_ptr: every graph at most max_graph_nodes
    const size_t n = (size_t)g->max_graph_nodes, ldp = (n + 3) & ~(size_t)3;
    return (size_t)g->n_graphs * H * n * ldp + 64;
}

int dense_train_prepare(const da_graph *g, int H, long long *poff, int32_t *node_graph, hipStream_t st) {
    k_pair_offsets<<<1, 64, 0, st>>>(g->n_graphs, H, g->graph_ptr, poff);
    k_node_graph<<<g->n_graphs, 256, 0, st>>>(g->n_graphs, g->graph_ptr, node_graph);
    DA_LAUNCH_CHECK();
    return 0;
}

// forward: o = softmax(scale q k^T) v + skip (+ res); P kept for the backward
int dense_train_attn_fwd(const da_graph *g, int H, int C, const float *qkvs, const float *res, float *o, float *P,
                         const long long *poff, const int32_t *node_graph, hipStream_t st, bool bfc, bool q16) {
    const int n = g->n_nodes, HC = H * C, G = g->n_graphs, mx = g->max_graph_nodes;
    if (attn_small_ok(g, C, bfc)) {       // small groups: the whole layer in one launch, P never stored (k_attn_small_fwd)
        AttnSmall a{};
        a.qkvs = qkvs; a.res = res; a.o = o;
        return attn_small_launch(g, H, C, a, false, q16, st);
    }
    if (q16) { set_error("training: bf16 projection buffers exist on the small-group attention route only"); return 1; }
    GGemm s;
    s.bfc = bfc;
    s.A = {(float *)qkvs, 0, 4 * HC, C};
    s.B = {(float *)qkvs + HC, 0, 4 * HC, C};
    s.C = {P, 1, 0, 0};
    s.transA = 0; s.transB = 1; s.dimM = 0; s.dimN = 0; s.dimK = C; s.alpha = 1.0f / sqrtf((float)C); s.accumulate = 0;
    s.H = H; s.gp = g->graph_ptr; s.poff = poff;
    int rc;
    const bool fused_rows = ggemm_small_ok(s, mx);       // small groups: the row softmax runs in the product's epilogue
    if (fused_rows) { s.epi = 1; s.nodiag = g->dense == 2; s.pair_bf16 = 1; }
    if ((rc = ggemm(s, G, H, mx, st))) return rc;
    if (!fused_rows) k_pair_rows<<<gridsz((size_t)n * H * 64), 256, 0, st>>>(0, n, H, g->dense == 2, g->graph_ptr, node_graph, poff, P, nullptr);
    k_init_out<<<gridsz((size_t)n * HC), 256, 0, st>>>(n, HC, qkvs, res, o);
    DA_LAUNCH_CHECK();
    GGemm pv;
    pv.bfc = bfc;
    pv.A = {P, 1, 0, 0};
    pv.B = {(float *)qkvs + 2 * HC, 0, 4 * HC, C};
    pv.C = {o, 0, HC, C};
    pv.transA = 0; pv.transB = 0; pv.dimM = 0; pv.dimN = C; pv.dimK = 0; pv.alpha = 1.0f; pv.accumulate = 1;
    pv.H = H; pv.gp = g->graph_ptr; pv.poff = poff;
    pv.pair_bf16 = fused_rows ? 1 : 0;                  // (ggemm_small_ok(pv) == ggemm_small_ok(s): same dimensions)
    return ggemm(pv, G, H, mx, st);
}

// backward: dY4 = [dq | dk | dv | d_o] from d_o [n, HC], the saved P and the projection buffer
int dense_train_attn_bwd(const da_graph *g, int H, int C, const float *qkvs, const float *d_o, const float *P, float *dP,
                         float *dY4, const long long *poff, const int32_t *node_graph, hipStream_t st, bool bfc, bool q16) {
    const int n = g->n_nodes, HC = H * C, G = g->n_graphs, mx = g->max_graph_nodes;
    const float scale = 1.0f / sqrtf((float)C);
    int rc;
    if (attn_small_ok(g, C, bfc)) {       // matches dense_train_attn_fwd's choice: that forward kept no P
        AttnSmall a{};
        a.qkvs = qkvs; a.d_o = d_o; a.dY4 = dY4;
        return attn_small_launch(g, H, C, a, true, q16, st);
    }
    if (q16) { set_error("training: bf16 projection buffers exist on the small-group attention route only"); return 1; }
    GGemm q;
    q.bfc = bfc;
    q.H = H; q.gp = g->graph_ptr; q.poff = poff; q.accumulate = 0;
    // dV = P^T dO
    q.A = {(float *)P, 1, 0, 0}; q.B = {(float *)d_o, 0, HC, C}; q.C = {dY4 + 2 * HC, 0, 4 * HC, C};
    q.transA = 1; q.transB = 0; q.dimM = 0; q.dimN = C; q.dimK = 0; q.alpha = 1.0f;
    q.pair_bf16 = ggemm_small_ok(q, mx) ? 1 : 0;        // the forward of this mode kept P in bf16 (dense_train_attn_fwd)
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    // dP = dO V^T
    q.A = {(float *)d_o, 0, HC, C}; q.B = {(float *)qkvs + 2 * HC, 0, 4 * HC, C}; q.C = {dP, 1, 0, 0};
    q.transA = 0; q.transB = 1; q.dimM = 0; q.dimN = 0; q.dimK = C; q.alpha = 1.0f;
    // dS = P o (dP - rowsum(P o dP)): small groups in the product's epilogue, else in place over dP by the row kernel
    const bool fused_rows = ggemm_small_ok(q, mx);
    if (fused_rows) { q.epi = 2; q.E = P; }
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    q.epi = 0; q.E = nullptr;
    if (!fused_rows) k_pair_rows<<<gridsz((size_t)n * H * 64), 256, 0, st>>>(1, n, H, 0, g->graph_ptr, node_graph, poff, (float *)P, dP);
    DA_LAUNCH_CHECK();
    // dQ = scale dS K
    q.A = {dP, 1, 0, 0}; q.B = {(float *)qkvs + HC, 0, 4 * HC, C}; q.C = {dY4, 0, 4 * HC, C};
    q.transA = 0; q.transB = 0; q.dimM = 0; q.dimN = C; q.dimK = 0; q.alpha = scale;
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    // dK = scale dS^T Q
    q.A = {dP, 1, 0, 0}; q.B = {(float *)qkvs, 0, 4 * HC, C}; q.C = {dY4 + HC, 0, 4 * HC, C};
    q.transA = 1; q.transB = 0; q.dimM = 0; q.dimN = C; q.dimK = 0; q.alpha = scale;
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    k_copy_skip_grad<<<gridsz((size_t)n * HC), 256, 0, st>>>(n, HC, d_o, dY4);
    DA_LAUNCH_CHECK();
    return 0;
}


// =====================================================================================================================
// HYBRID graphs in training (Exphander edges + exophormer virtual nodes, the scripted training configuration,
// singularity/gianscarpe/train_celeba_rot.sh:4-15): the unique real -> real in-graph edges ("regular", one adjacency bit per
// (target, source) pair, da_graph.mask) run through the same grouped matrix-core GEMMs as complete graphs; every other edge
// (virtual nodes, duplicated pairs, cross-graph pairs of the exophormer quirk, exophormer_gnn.py:183-200) stays a small CSR
// by destination (irr_row_ptr / irr_col_src) + by source (out_ptr / out_dst of a hybrid training graph).  ONE softmax per
// (target, head) spans both parts: statistics (max, 1 / (sum + 1e-16)) are computed over the masked dense row AND the row's
// remainder edges, the dense part of P is kept, the remainder weights are recomputed from the statistics wherever needed --
// exactly what the CSR kernels of da_train.hip do for whole graphs.
// rows of the pair matrices + their remainder edges: one wave per (node, head); virtual rows (node >= n_real) have no dense part.
//   mode 0: S (scaled scores) -> P in place (masked entries 0), stats[node, h] = (max, 1 / (sum + 1e-16)) over both parts
//   mode 2: Dd[node, h] = sum_j P_ij dP_ij over the dense part (virtual rows: 0)
//   mode 1: dS = P o (dP - Dd[node, h]) written over dP (Dd = the total over both parts by then)
__global__ __launch_bounds__(256) void k_pair_rows_hybrid(int mode, int row0, int n_nodes, int n_real, int H, int C, const int32_t *__restrict__ gp,
                                                          const int32_t *__restrict__ pad_ptr, const int32_t *__restrict__ node_graph,
                                                          const long long *__restrict__ poff, const unsigned char *__restrict__ mask,
                                                          const long long *__restrict__ mask_ptr, const int32_t *__restrict__ irr_ptr,
                                                          const int32_t *__restrict__ irr_src, const float *__restrict__ qkvs,
                                                          float *P, float *dP, float *__restrict__ stats, float *__restrict__ Dd) {
    const int lane = threadIdx.x & 63;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int HC = H * C;
    const float scale = 1.0f / sqrtf((float)C);
    for (long long wv = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wv < (long long)(n_nodes - row0) * H; wv += n_waves) {      // rows [row0, n_nodes)
        const int node = row0 + (int)(wv / H), h = (int)(wv - (long long)(node - row0) * H);
        const bool real = node < n_real;
        int n_g = 0, i = 0, ldp = 0;
        float *row = nullptr;
        const unsigned char *mrow = nullptr;
        if (real) {
            const int g = node_graph[node];
            n_g = gp[g + 1] - gp[g]; i = node - gp[g]; ldp = (n_g + 3) & ~3;
            row = P + poff[g] + ((size_t)h * n_g + i) * ldp;
            mrow = mask + mask_ptr[g] + (size_t)i * (size_t)((pad_ptr[g + 1] - pad_ptr[g]) >> 3);
        }
        if (mode == 0) {
            // online max / sum over the masked dense row ...
            // (sixteen 64-element groups per pass: all their loads are issued before the first use -- one load per lane in
            // flight made the pass a chain of ~30 dependent round trips per row, 0.85 ms per launch at 16 x 900 pieces)
            float m = -INFINITY, z = 0.f;
            for (int jb = 0; jb < n_g; jb += 1024) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {          // unconditional loads at clamped positions (guarded ones compile to load -> wait chains)
                    const int j0 = jb + 64 * u, j = j0 + lane, jl = n_g - 1;
                    const unsigned long long bits = *(const unsigned long long *)(mrow + (min(j0, jl & ~63) >> 3));
                    const float rv = row[min(j, jl)];
                    v[u] = (j < n_g && ((bits >> lane) & 1ull)) ? rv : -INFINITY;
                }
                float mc = m;
#pragma unroll
                for (int u = 0; u < 16; ++u) mc = fmaxf(mc, v[u]);
                if (mc > -INFINITY) {
                    float zc = 0.f;
#pragma unroll
                    for (int u = 0; u < 16; ++u) zc += v[u] > -INFINITY ? expf(v[u] - mc) : 0.f;
                    z = (m > -INFINITY ? z * expf(m - mc) : 0.f) + zc;
                    m = mc;
                }
            }
            // ... and this row's remainder edges, a LANE per edge (64 at a time; the lane walks the head's C channels): one wave
            // reduction per edge and a dependent load chain per edge made the 900-edge rows of the virtual nodes 1.7 ms launches
            const int eb = irr_ptr[node], ee = irr_ptr[node + 1];
            const float *qp = qkvs + (size_t)node * 4 * HC + (size_t)h * C;
            float mi = -INFINITY, zi = 0.f;
            for (int e = eb + lane; e < ee; e += 64) {
                const float *kp = qkvs + (size_t)irr_src[e] * 4 * HC + HC + (size_t)h * C;
                float s = 0.f;
                for (int c = 0; c < C; c += 4) {
                    const f32x4 qv = *(const f32x4 *)(qp + c), kv = *(const f32x4 *)(kp + c);
                    s = fmaf(qv[0], kv[0], s); s = fmaf(qv[1], kv[1], s); s = fmaf(qv[2], kv[2], s); s = fmaf(qv[3], kv[3], s);
                }
                s *= scale;
                const float mn = fmaxf(mi, s);
                zi = zi * expf(mi - mn) + expf(s - mn);
                mi = mn;
            }
            // the lane's remainder part joins its dense part, then the lanes merge
            {
                const float mn = fmaxf(m, mi);
                if (mn > -INFINITY) { z = (m > -INFINITY ? z * expf(m - mn) : 0.f) + (mi > -INFINITY ? zi * expf(mi - mn) : 0.f); m = mn; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float m2 = __shfl_xor(m, o), z2 = __shfl_xor(z, o);
                const float mn = fmaxf(m, m2);
                if (mn > -INFINITY) { z = (m > -INFINITY ? z * expf(m - mn) : 0.f) + (m2 > -INFINITY ? z2 * expf(m2 - mn) : 0.f); m = mn; }
            }
            const float inv = (m > -INFINITY) ? 1.0f / (z + 1e-16f) : 0.f;
            const float mfin = (m > -INFINITY) ? m : 0.f;
            for (int jb = 0; jb < n_g; jb += 1024) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {          // unconditional loads at clamped positions (guarded ones compile to load -> wait chains)
                    const int j0 = jb + 64 * u, j = j0 + lane, jl = n_g - 1;
                    const unsigned long long bits = *(const unsigned long long *)(mrow + (min(j0, jl & ~63) >> 3));
                    const float rv = row[min(j, jl)];
                    v[u] = (j < n_g && ((bits >> lane) & 1ull)) ? rv : -INFINITY;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int j = jb + 64 * u + lane;
                    if (j < n_g) row[j] = v[u] > -INFINITY ? expf(v[u] - mfin) * inv : 0.f;
                }
            }
            if (lane == 0) { stats[((size_t)node * H + h) * 2] = mfin; stats[((size_t)node * H + h) * 2 + 1] = inv; }
        } else if (mode == 2) {
            float D = 0.f;
            if (real) {
                const float *drow = dP + poff[node_graph[node]] + ((size_t)h * n_g + i) * ldp;
                for (int j = lane; j < n_g; j += 64) D = fmaf(row[j], drow[j], D);
                for (int o = 32; o > 0; o >>= 1) D += __shfl_xor(D, o);
            }
            if (lane == 0) Dd[(size_t)node * H + h] = D;
        } else if (real) {
            float *drow = dP + poff[node_graph[node]] + ((size_t)h * n_g + i) * ldp;
            const float D = Dd[(size_t)node * H + h];
            for (int j = lane; j < n_g; j += 64) drow[j] = row[j] * (drow[j] - D);
        }
    }
}

// Heavy rows of the remainder CSR (round 4).  The three kernels below walk a row's edges one after the other in ONE wave, each
// edge a dependent load chain (index -> K / V or Q / dO row) of about a microsecond: fine for the 8 - 20 remainder edges of a
// piece, 0.8 - 2.1 ms per launch for the exophormer's virtual nodes (900 edges each way; 12 launches per step = half the scripted
// training step).  Every sum over edges is plain (the softmax statistics are known), so a row with more than IRR_HEAVY edges
// is taken by a WORKGROUP of 8 - 16 waves instead (HEAVY instances: one block per row, rows below the threshold return at
// once; the wave-per-row instances skip the rows above it; HEAVY launches cover the rows behind the real nodes only -- the
// virtual nodes -- and the wave-per-row instances keep every real row whatever its degree: a grid over all rows spent
// 80 - 280 us per launch dispatching empty workgroups): wave w takes edges beg + w, beg + w + IRR_NW, ..., the partial sums
// meet in LDS and are added in wave order -- deterministic, no atomics.
constexpr int IRR_HEAVY = 96;
template <int EPL> struct IrrU { static constexpr int v = EPL <= 4 ? 4 : 2; };          // edges of a wave in flight per trip (round 6)
template <int EPL> struct IrrNW { static constexpr int v = EPL <= 4 ? 16 : 8; };      // (eight at C = 144: the pipelined loop wants more than the 128 registers of a 1024-thread block)     // waves of a heavy row's workgroup (LDS: v x 64 x (EPL + 1) floats)

// o[i, :] += sum over i's remainder edges of p_e v_src, p_e = exp(s_e - m_i) inv_i from the combined statistics.  Wave per
// destination, lane = EPL contiguous channels of the H*C-wide rows (8 lanes per head), as the CSR kernels of da_train.hip.
template <int EPL, bool HEAVY>
__global__ __launch_bounds__(HEAVY ? 64 * IrrNW<EPL>::v : 256) void k_attn_irr_fwd(int n_nodes, int n_real, const int32_t *__restrict__ irr_ptr, const int32_t *__restrict__ irr_src,
                                                      int H, int HC, const float *__restrict__ qkvs, const float *__restrict__ stats,
                                                      float *__restrict__ o, float scale, int row0) {
    constexpr int IRR_NW = IrrNW<EPL>::v;
    __shared__ float red[HEAVY ? IRR_NW : 1][HEAVY ? 64 * EPL : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = HEAVY ? n_real + (int)blockIdx.x : row0 + (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);      // (n_nodes = the end of the launch's row range)
    if (i >= n_nodes) return;
    const int beg = irr_ptr[i], end = irr_ptr[i + 1];
    if (HEAVY ? end - beg <= IRR_HEAVY : (beg == end || (i >= n_real && end - beg > IRR_HEAVY))) return;
    const size_t ld = (size_t)4 * HC;
    const int off = lane * EPL, head = lane >> 3;
    float q[EPL], acc[EPL];
#pragma unroll
    for (int x = 0; x < EPL; ++x) { q[x] = qkvs[(size_t)i * ld + off + x] * scale; acc[x] = 0.f; }
    const float m = stats[((size_t)i * H + head) * 2], inv = stats[((size_t)i * H + head) * 2 + 1];
    // software pipeline: the rows of edge e + 1 are requested before edge e is consumed, its index one edge earlier still (the
    // loop was two dependent round trips per edge: index -> rows)
    // (round 6: U edges of the wave per trip -- their rows are requested together, the next trip's rows before this trip is consumed and
    //  its indices a trip earlier still: a trip is one memory round trip for U edges.  Same edge order per wave as before: same sums.)
    constexpr int ST = HEAVY ? IRR_NW : 1, U = IrrU<EPL>::v;
    int e = HEAVY ? beg + wv : beg;
    float kn[U][EPL], vn[U][EPL];
    int s1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        s1[u] = 0;
        if (e + u * ST < end) {
            const float *kp = qkvs + (size_t)irr_src[e + u * ST] * ld + HC + off;
#pragma unroll
            for (int x = 0; x < EPL; ++x) { kn[u][x] = kp[x]; vn[u][x] = kp[HC + x]; }
            if (e + (U + u) * ST < end) s1[u] = irr_src[e + (U + u) * ST];
        }
    }
    for (; e < end; e += U * ST) {
        float kk[U][EPL], vv[U][EPL];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int x = 0; x < EPL; ++x) { kk[u][x] = kn[u][x]; vv[u][x] = vn[u][x]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + (U + u) * ST < end) {
                const float *kp = qkvs + (size_t)s1[u] * ld + HC + off;
#pragma unroll
                for (int x = 0; x < EPL; ++x) { kn[u][x] = kp[x]; vn[u][x] = kp[HC + x]; }
                if (e + (2 * U + u) * ST < end) s1[u] = irr_src[e + (2 * U + u) * ST];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u * ST >= end) break;
            float s = 0.f;
#pragma unroll
            for (int x = 0; x < EPL; ++x) s = fmaf(q[x], kk[u][x], s);
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            const float pe = expf(s - m) * inv;
#pragma unroll
            for (int x = 0; x < EPL; ++x) acc[x] = fmaf(pe, vv[u][x], acc[x]);
        }
    }
    if (HEAVY) {
#pragma unroll
        for (int x = 0; x < EPL; ++x) red[wv][x * 64 + lane] = acc[x];
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int x = 0; x < EPL; ++x) {
            float t = red[0][x * 64 + lane];
            for (int w = 1; w < IRR_NW; ++w) t += red[w][x * 64 + lane];
            acc[x] = t;
        }
    }
#pragma unroll
    for (int x = 0; x < EPL; ++x) o[(size_t)i * HC + off + x] += acc[x];
}

// backward over the remainder edges, destination side: D_i total = Dd[i] (dense part, on entry) + sum_e p_e dp_e; writes
// D_i total back, dq_i of the remainder edges into dY4 (the dense dQ GEMM ACCUMULATES on top afterwards) and the skip gradient.
template <int EPL, bool HEAVY>
__global__ __launch_bounds__(HEAVY ? 64 * IrrNW<EPL>::v : 256) void k_attn_irr_bwd_dst(int n_nodes, int n_real, const int32_t *__restrict__ irr_ptr, const int32_t *__restrict__ irr_src,
                                                          int H, int HC, const float *__restrict__ qkvs, const float *__restrict__ d_o,
                                                          const float *__restrict__ stats, float *__restrict__ dY4, float *__restrict__ Dd,
                                                          float scale, int dtotal, int row0) {
    // dtotal: Dd already holds the row's TOTAL D (k_hyb_rowdot, the flash-style route): used as it is, not written back
    constexpr int IRR_NW = IrrNW<EPL>::v;
    __shared__ float red[HEAVY ? IRR_NW : 1][HEAVY ? 64 * (EPL + 1) : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = HEAVY ? n_real + (int)blockIdx.x : row0 + (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);      // (n_nodes = the end of the launch's row range)
    if (i >= n_nodes) return;
    const int beg = irr_ptr[i], end = irr_ptr[i + 1];
    if (HEAVY ? end - beg <= IRR_HEAVY : (i >= n_real && end - beg > IRR_HEAVY)) return;      // (rows without edges still get their skip gradient)
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
    constexpr int ST = HEAVY ? IRR_NW : 1, U = IrrU<EPL>::v;                 // (software pipeline as in k_attn_irr_fwd: U edges per trip)
    int e = HEAVY ? beg + wv : beg;
    float kn[U][EPL], vn[U][EPL];
    int s1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        s1[u] = 0;
        if (e + u * ST < end) {
            const float *kp = qkvs + (size_t)irr_src[e + u * ST] * ld + HC + off;
#pragma unroll
            for (int x = 0; x < EPL; ++x) { kn[u][x] = kp[x]; vn[u][x] = kp[HC + x]; }
            if (e + (U + u) * ST < end) s1[u] = irr_src[e + (U + u) * ST];
        }
    }
    for (; e < end; e += U * ST) {
        float kk[U][EPL], vv[U][EPL];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int x = 0; x < EPL; ++x) { kk[u][x] = kn[u][x]; vv[u][x] = vn[u][x]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + (U + u) * ST < end) {
                const float *kp = qkvs + (size_t)s1[u] * ld + HC + off;
#pragma unroll
                for (int x = 0; x < EPL; ++x) { kn[u][x] = kp[x]; vn[u][x] = kp[HC + x]; }
                if (e + (2 * U + u) * ST < end) s1[u] = irr_src[e + (2 * U + u) * ST];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u * ST >= end) break;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int x = 0; x < EPL; ++x) { s = fmaf(q[x], kk[u][x], s); dp = fmaf(g[x], vv[u][x], dp); }
            s += __shfl_xor(s, 1); dp += __shfl_xor(dp, 1);
            s += __shfl_xor(s, 2); dp += __shfl_xor(dp, 2);
            s += __shfl_xor(s, 4); dp += __shfl_xor(dp, 4);
            const float pe = expf(s - m) * inv, pd = pe * dp;
            D += pd;
#pragma unroll
            for (int x = 0; x < EPL; ++x) { a1[x] = fmaf(pd, kk[u][x], a1[x]); a2[x] = fmaf(pe, kk[u][x], a2[x]); }
        }
    }
    if (HEAVY) {                                            // two rounds through the same LDS: (a1, D), then a2
#pragma unroll
        for (int x = 0; x < EPL; ++x) red[wv][x * 64 + lane] = a1[x];
        red[wv][EPL * 64 + lane] = D;
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int x = 0; x < EPL; ++x) {
                float t = red[0][x * 64 + lane];
                for (int w = 1; w < IRR_NW; ++w) t += red[w][x * 64 + lane];
                a1[x] = t;
            }
            float t = red[0][EPL * 64 + lane];
            for (int w = 1; w < IRR_NW; ++w) t += red[w][EPL * 64 + lane];
            D = t;
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < EPL; ++x) red[wv][x * 64 + lane] = a2[x];
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int x = 0; x < EPL; ++x) {
            float t = red[0][x * 64 + lane];
            for (int w = 1; w < IRR_NW; ++w) t += red[w][x * 64 + lane];
            a2[x] = t;
        }
    }
    const float Dt = dtotal ? Dd[(size_t)i * H + head] : D + Dd[(size_t)i * H + head];
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        dY4[(size_t)i * ld + off + x] = (a1[x] - Dt * a2[x]) * scale;
        dY4[(size_t)i * ld + 3 * (size_t)HC + off + x] = g[x];
    }
    if (!dtotal && (lane & 7) == 0) Dd[(size_t)i * H + head] = Dt;          // (the head's eight lanes read it above, in lockstep)
}

// source side over the remainder edges (CSR by source): dk_j, dv_j ADDED to what the dense GEMMs wrote
template <int EPL, bool HEAVY>
__global__ __launch_bounds__(HEAVY ? 64 * IrrNW<EPL>::v : 256) void k_attn_irr_bwd_src(int n_nodes, int n_real, const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_dst,
                                                          int H, int HC, const float *__restrict__ qkvs, const float *__restrict__ d_o,
                                                          const float *__restrict__ stats, const float *__restrict__ Dd,
                                                          float *__restrict__ dY4, float scale, int row0) {
    constexpr int IRR_NW = IrrNW<EPL>::v;
    __shared__ float red[HEAVY ? IRR_NW : 1][HEAVY ? 64 * EPL : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = HEAVY ? n_real + (int)blockIdx.x : row0 + (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);      // (n_nodes = the end of the launch's row range)
    if (j >= n_nodes) return;
    const int beg = out_ptr[j], end = out_ptr[j + 1];
    if (HEAVY ? end - beg <= IRR_HEAVY : (beg == end || (j >= n_real && end - beg > IRR_HEAVY))) return;
    const size_t ld = (size_t)4 * HC;
    const int off = lane * EPL, head = lane >> 3;
    float kk[EPL], vv[EPL], dk[EPL], dv[EPL];
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        kk[x] = qkvs[(size_t)j * ld + HC + off + x];
        vv[x] = qkvs[(size_t)j * ld + 2 * (size_t)HC + off + x];
        dk[x] = dv[x] = 0.f;
    }
    constexpr int ST = HEAVY ? IRR_NW : 1, U = IrrU<EPL>::v;                 // (software pipeline as in k_attn_irr_fwd: U edges per trip)
    int e = HEAVY ? beg + wv : beg;
    float qn[U][EPL], gn[U][EPL], mn[U], invn[U], Dn[U];
    int s1[U];
    auto fetch = [&](int u, int i) {
#pragma unroll
        for (int x = 0; x < EPL; ++x) { qn[u][x] = qkvs[(size_t)i * ld + off + x]; gn[u][x] = d_o[(size_t)i * HC + off + x]; }
        mn[u] = stats[((size_t)i * H + head) * 2]; invn[u] = stats[((size_t)i * H + head) * 2 + 1];
        Dn[u] = Dd[(size_t)i * H + head];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) {
        s1[u] = 0; mn[u] = invn[u] = Dn[u] = 0.f;
        if (e + u * ST < end) {
            fetch(u, out_dst[e + u * ST]);
            if (e + (U + u) * ST < end) s1[u] = out_dst[e + (U + u) * ST];
        }
    }
    for (; e < end; e += U * ST) {
        float q[U][EPL], g[U][EPL], m[U], inv[U], D[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int x = 0; x < EPL; ++x) { q[u][x] = qn[u][x] * scale; g[u][x] = gn[u][x]; }
            m[u] = mn[u]; inv[u] = invn[u]; D[u] = Dn[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + (U + u) * ST < end) {
                fetch(u, s1[u]);
                if (e + (2 * U + u) * ST < end) s1[u] = out_dst[e + (2 * U + u) * ST];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u * ST >= end) break;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int x = 0; x < EPL; ++x) { s = fmaf(q[u][x], kk[x], s); dp = fmaf(g[u][x], vv[x], dp); }
            s += __shfl_xor(s, 1); dp += __shfl_xor(dp, 1);
            s += __shfl_xor(s, 2); dp += __shfl_xor(dp, 2);
            s += __shfl_xor(s, 4); dp += __shfl_xor(dp, 4);
            const float pe = expf(s - m[u]) * inv[u], ds = pe * (dp - D[u]);
#pragma unroll
            for (int x = 0; x < EPL; ++x) { dk[x] = fmaf(ds, q[u][x], dk[x]); dv[x] = fmaf(pe, g[u][x], dv[x]); }
        }
    }
    if (HEAVY) {                                            // two rounds through the same LDS: dk, then dv
#pragma unroll
        for (int x = 0; x < EPL; ++x) red[wv][x * 64 + lane] = dk[x];
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int x = 0; x < EPL; ++x) {
                float t = red[0][x * 64 + lane];
                for (int w = 1; w < IRR_NW; ++w) t += red[w][x * 64 + lane];
                dk[x] = t;
            }
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < EPL; ++x) red[wv][x * 64 + lane] = dv[x];
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int x = 0; x < EPL; ++x) {
            float t = red[0][x * 64 + lane];
            for (int w = 1; w < IRR_NW; ++w) t += red[w][x * 64 + lane];
            dv[x] = t;
        }
    }
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        dY4[(size_t)j * ld + HC + off + x] += dk[x];
        dY4[(size_t)j * ld + 2 * (size_t)HC + off + x] += dv[x];
    }
}

// zero the dk | dv columns of rows [r0, r1) (virtual rows: only the remainder kernels write them)
__global__ __launch_bounds__(256) void k_zero_kv_grad(int r0, int r1, int HC, float *__restrict__ dY4) {
    const size_t total = (size_t)(r1 - r0) * 2 * HC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / (2 * HC), c = idx - r * 2 * HC;
        dY4[((size_t)r0 + r) * 4 * HC + HC + c] = 0.f;
    }
}


// =====================================================================================================================
// HYBRID graphs, bf16-operand mode, FLASH-STYLE (round 5).  The route above keeps the masked pair matrices of a (graph, head)
// in HBM -- 422 MB per layer in fp32 at 16 puzzles of 900 pieces, crossed ~14 times per layer (S, P, dP, dS written and read by
// five grouped GEMMs and three row passes): 15 of every 16 microseconds of the scripted training step (singularity/gianscarpe/
// train_celeba_rot.sh:4-15; exophormer_gnn.py:161-215 under spatial_diffusion.py:432-483) moved pair matrices.  Here no [n, n]
// tensor exists: the k_attn_small_* scheme (bf16 row-major LDS images, the wave's own 16-row band in registers as B fragments,
// transposed tiles so that a lane holds four consecutive keys of ONE query, transposing LDS reads for the second product) tiled
// over the other dimension in chunks of 128 rows, with the adjacency bits of the regular edges applied to the score tiles:
//   k_hyb_fwd     query bands.  Pass 1: scores only -> (max, sum) of the masked dense row, then the row's remainder edges join
//                 (fp32 dot products, as everywhere on the remainder side) -> the statistics (max, 1 / (sum + 1e-16)) of PyG's
//                 softmax over BOTH parts, kept for the backward.  Pass 2: scores again, P = exp(s - max) inv (bf16 operand),
//                 O = P V, + skip (+ residual).  The remainder edges' own contributions are added by k_attn_irr_fwd as before.
//   k_hyb_rowdot  D_i = sum_c dO_ic attn_ic (attn = o - skip - residual): the softmax backward's row term over ALL edges of the
//                 row -- dense and remainder -- without touching either edge set.
//   k_hyb_bwd_q   query bands: S, dP = dO V^T, dS = P o (dP - D), dQ += scale dS K (on top of the remainder's dq).
//   k_hyb_bwd_kv  key bands against query chunks (Q, dO images + the queries' statistics and adjacency bytes in LDS):
//                 dV = P^T dO, dK = scale dS^T Q.
// Products and roundings are those of the bf16-operand mode (operands bf16, P and dS rounded to bf16 as operands, fp32
// accumulation, fp32 storage of everything that reaches HBM).  Virtual rows (no dense part) keep their kernels.
constexpr int HF_WAVES = 8, HF_ROWS = 16 * HF_WAVES, HF_KC = 128, HF_TN = HF_KC / 16, HF_NT = 64 * HF_WAVES;
struct HybFlash {
    const float *qkvs;           // [n, 4 HC]  Q | K | V | skip
    const float *res;            // forward / rowdot: residual [n, HC] or null
    float *o;                    // forward: out; rowdot: in
    const float *d_o;            // backward: [n, HC]
    float *dY4;                  // backward: [n, 4 HC]  dq | dk | dv | d_o
    float *stats;                // [n, H, 2]: forward writes the real rows', backward reads
    float *Dd;                   // [n, H]: rowdot writes, backward reads
    const int32_t *gp, *pad_ptr;
    const unsigned char *mask;
    const long long *mask_ptr;
    const int32_t *irr_ptr, *irr_src;
    int H, C, HC, nrt, n_nodes;  // nrt: 128-row tiles per graph (of the largest graph)
    float scale;
};
#define AS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)
// exp through the hardware's exp2 (one multiply + v_exp_f32; ~1e-6 relative): these kernels evaluate four exponentials per
// (query, key) pair and step, libm's expf is ~10 instructions each
__device__ __forceinline__ float hf_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// rows [0, HF_KC) of a node matrix from element offset e0 (row stride ld; `valid` rows exist, C columns) as a row-major bf16 image
// of pitch 16 CT + 8, zero beyond the valid part
template <int CT>
__device__ __forceinline__ void hf_stage(const float *X, size_t e0, int ld, int valid, int C, unsigned short *R, int tid) {
    constexpr int kq = CT * 4, pr = CT * 16 + 8, NB = CT >= 4 ? 3 : 2;
#pragma unroll 1
    for (int i0 = tid; i0 < HF_KC * kq; i0 += HF_NT * NB) {
        as_s16x4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = i0 + u * HF_NT, r = idx / kq, k = (idx - r * kq) * 4;
            v[u] = (as_s16x4){0, 0, 0, 0};
            if (idx < HF_KC * kq && r < valid && k < C) v[u] = as_ld_pack<false>(X, e0 + (size_t)r * ld + k);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = i0 + u * HF_NT, r = idx / kq, k = (idx - r * kq) * 4;
            if (idx < HF_KC * kq) *(as_s16x4 *)(R + r * pr + k) = v[u];
        }
    }
}
// adjacency nibble of (this lane's query row, keys k0 + 16 tn + 4 lg ..+3) out of the row's 128 bits of the chunk
__device__ __forceinline__ unsigned hf_nib(unsigned long long b0, unsigned long long b1, int tn, int lg) {
    const unsigned long long w = tn < 4 ? b0 : b1;
    return (unsigned)(w >> (((tn & 3) << 4) + 4 * lg)) & 15u;
}

template <int CT>
__global__ __launch_bounds__(HF_NT) void k_hyb_fwd(HybFlash p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short hf_lds[];
    constexpr int pr = CT * 16 + 8;
    unsigned short *Ks = hf_lds, *Vs = Ks + HF_KC * pr;
    const int rt = blockIdx.x % p.nrt, gh = blockIdx.x / p.nrt, h = gh % p.H, g = gh / p.H;
    const int n0 = p.gp[g], n_g = p.gp[g + 1] - n0, r0 = rt * HF_ROWS;
    if (r0 >= n_g) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int C = p.C, ld = 4 * p.HC;
    const size_t e0 = (size_t)n0 * ld + h * C;
    const int band0 = r0 + wid * 16, i = band0 + l15;
    const bool band_on = band0 < n_g, row_on = i < n_g;
    as_s16x4 qf[CT];
    as_band<CT, false>(p.qkvs, e0, ld, n_g, C, band0, l15, lg, qf);
    const unsigned char *mrow = p.mask + p.mask_ptr[g] + (size_t)min(i, n_g - 1) * (size_t)((p.pad_ptr[g + 1] - p.pad_ptr[g]) >> 3);
    const int nkc = (n_g + HF_KC - 1) / HF_KC;
    // ---- pass 1: statistics of the masked dense row
    float m = -INFINITY, z = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
        const int k0 = kc * HF_KC;
        __syncthreads();
        hf_stage<CT>(p.qkvs, e0 + p.HC + (size_t)k0 * ld, ld, n_g - k0, C, Ks, tid);
        const unsigned long long b0 = *(const unsigned long long *)(mrow + (k0 >> 3)), b1 = *(const unsigned long long *)(mrow + (k0 >> 3) + 8);
        __syncthreads();
        if (band_on) {
#pragma unroll
            for (int tn = 0; tn < HF_TN; ++tn) {
                if (k0 + tn * 16 < n_g) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const unsigned short *kp = Ks + (tn * 16 + l15) * pr + 4 * lg;
#pragma unroll
                    for (int kt = 0; kt < CT; ++kt) acc = AS_MFMA(*(const as_s16x4 *)(kp + kt * 16), qf[kt], acc);
                    const unsigned nib = hf_nib(b0, b1, tn, lg);
                    float sv[4], mc = m;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (k0 + tn * 16 + 4 * lg + r < n_g) && ((nib >> r) & 1u);
                        sv[r] = ok ? p.scale * acc[r] : -INFINITY;
                        mc = fmaxf(mc, sv[r]);
                    }
                    if (mc > -INFINITY) {
                        float zc = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) zc += sv[r] > -INFINITY ? hf_exp(sv[r] - mc) : 0.f;
                        z = (m > -INFINITY ? z * hf_exp(m - mc) : 0.f) + zc;
                        m = mc;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float m2 = __shfl_xor(m, o), z2 = __shfl_xor(z, o);
        const float mn = fmaxf(m, m2);
        if (mn > -INFINITY) { z = (m > -INFINITY ? z * hf_exp(m - mn) : 0.f) + (m2 > -INFINITY ? z2 * hf_exp(m2 - mn) : 0.f); m = mn; }
    }
    // ... and the row's remainder edges (fp32 dot products; the four lanes of a query split the channels)
    if (band_on && row_on) {
        const int node = n0 + i, cq = C >> 2;
        const float *qp = p.qkvs + (size_t)node * ld + (size_t)h * C + lg * cq;
        for (int e = p.irr_ptr[node]; e < p.irr_ptr[node + 1]; ++e) {
            const float *kp = p.qkvs + (size_t)p.irr_src[e] * ld + p.HC + (size_t)h * C + lg * cq;
            float sc = 0.f;
            for (int c = 0; c < cq; c += 4) {
                const f32x4 qv = *(const f32x4 *)(qp + c), kv = *(const f32x4 *)(kp + c);
                sc = fmaf(qv[0], kv[0], sc); sc = fmaf(qv[1], kv[1], sc); sc = fmaf(qv[2], kv[2], sc); sc = fmaf(qv[3], kv[3], sc);
            }
            sc += __shfl_xor(sc, 16);
            sc += __shfl_xor(sc, 32);
            sc *= p.scale;
            const float mn = fmaxf(m, sc);
            z = (m > -INFINITY ? z * hf_exp(m - mn) : 0.f) + hf_exp(sc - mn);
            m = mn;
        }
    }
    const float inv = (m > -INFINITY) ? 1.0f / (z + 1e-16f) : 0.f, mf = (m > -INFINITY) ? m : 0.f;
    if (band_on && row_on && lg == 0) {
        p.stats[((size_t)(n0 + i) * p.H + h) * 2] = mf;
        p.stats[((size_t)(n0 + i) * p.H + h) * 2 + 1] = inv;
    }
    // ---- pass 2: O = P V over the regular edges
    f32x4 oacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) oacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < nkc; ++kc) {
        const int k0 = kc * HF_KC;
        __syncthreads();
        hf_stage<CT>(p.qkvs, e0 + p.HC + (size_t)k0 * ld, ld, n_g - k0, C, Ks, tid);
        hf_stage<CT>(p.qkvs, e0 + 2 * p.HC + (size_t)k0 * ld, ld, n_g - k0, C, Vs, tid);
        const unsigned long long b0 = *(const unsigned long long *)(mrow + (k0 >> 3)), b1 = *(const unsigned long long *)(mrow + (k0 >> 3) + 8);
        __syncthreads();
        if (band_on) {
#pragma unroll
            for (int tn = 0; tn < HF_TN; ++tn) {
                if (k0 + tn * 16 < n_g) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const unsigned short *kp = Ks + (tn * 16 + l15) * pr + 4 * lg;
#pragma unroll
                    for (int kt = 0; kt < CT; ++kt) acc = AS_MFMA(*(const as_s16x4 *)(kp + kt * 16), qf[kt], acc);
                    const unsigned nib = hf_nib(b0, b1, tn, lg);
                    float pv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (k0 + tn * 16 + 4 * lg + r < n_g) && ((nib >> r) & 1u) && inv > 0.f;
                        pv[r] = ok ? hf_exp(p.scale * acc[r] - mf) * inv : 0.f;
                    }
                    const as_s16x4 pa = as_pack(pv[0], pv[1], pv[2], pv[3]);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) oacc[ct] = AS_MFMA(as_tr(Vs, pr, tn * 16, ct * 16, l15, lg), pa, oacc[ct]);
                }
            }
        }
    }
    if (band_on && row_on) {
        const size_t es = e0 + 3 * p.HC + (size_t)i * ld;
        f32x4 sk[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) sk[ct] = *(const f32x4 *)(p.qkvs + es + (ct * 16 + 4 * lg < C ? ct * 16 + 4 * lg : 0));
        if (p.res) {
            const float *rsp = p.res + (size_t)(n0 + i) * p.HC + h * C;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) sk[ct] += *(const f32x4 *)(rsp + (ct * 16 + 4 * lg < C ? ct * 16 + 4 * lg : 0));
        }
        float *ob = p.o + (size_t)(n0 + i) * p.HC + h * C + 4 * lg;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            if (ct * 16 + 4 * lg < C) *(f32x4 *)(ob + ct * 16) = oacc[ct] + sk[ct];
    }
}

// D[node, h] = sum_c dO o attn,  attn = o - skip - residual  (every row, virtual ones included).  One wave per node, a lane owns
// four consecutive channels (coalesced 16-byte loads): C = 32 -- the 8 heads are the wave's 8-lane groups, one pass; other widths --
// one pass per head over its C / 4 lanes, whole-wave shuffle sum.
__global__ __launch_bounds__(256) void k_hyb_rowdot(HybFlash p) {
    const int lane = threadIdx.x & 63;
    const size_t node = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (node >= (size_t)p.n_nodes) return;
    const float *g = p.d_o + node * p.HC, *o = p.o + node * p.HC, *sk = p.qkvs + node * 4 * (size_t)p.HC + 3 * (size_t)p.HC;
    const float *rs = p.res ? p.res + node * p.HC : nullptr;
    auto part = [&](int c, bool on) {
        float D = 0.f;
        if (on) {
            const f32x4 gv = *(const f32x4 *)(g + c), ov = *(const f32x4 *)(o + c), sv = *(const f32x4 *)(sk + c);
            f32x4 a = ov - sv;
            if (rs) a -= *(const f32x4 *)(rs + c);
            D = gv[0] * a[0] + gv[1] * a[1] + gv[2] * a[2] + gv[3] * a[3];
        }
        return D;
    };
    if (p.C == 32 && p.H == 8) {
        float D = part(4 * lane, true);
        D += __shfl_xor(D, 1); D += __shfl_xor(D, 2); D += __shfl_xor(D, 4);
        if ((lane & 7) == 0) p.Dd[node * 8 + (lane >> 3)] = D;
        return;
    }
    for (int h = 0; h < p.H; ++h) {                              // C <= 256
        float D = part(h * p.C + 4 * lane, 4 * lane < p.C);
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) D += __shfl_xor(D, of);
        if (lane == 0) p.Dd[node * p.H + h] = D;
    }
}

template <int CT>
__global__ __launch_bounds__(HF_NT) void k_hyb_bwd_q(HybFlash p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short hf_lds[];
    constexpr int pr = CT * 16 + 8;
    unsigned short *Ks = hf_lds, *Vs = Ks + HF_KC * pr;
    const int rt = blockIdx.x % p.nrt, gh = blockIdx.x / p.nrt, h = gh % p.H, g = gh / p.H;
    const int n0 = p.gp[g], n_g = p.gp[g + 1] - n0, r0 = rt * HF_ROWS;
    if (r0 >= n_g) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int C = p.C, ld = 4 * p.HC;
    const size_t e0 = (size_t)n0 * ld + h * C, g0 = (size_t)n0 * p.HC + h * C;
    const int band0 = r0 + wid * 16, i = band0 + l15;
    const bool band_on = band0 < n_g, row_on = i < n_g;
    as_s16x4 qf[CT], gf[CT];
    as_band<CT, false>(p.qkvs, e0, ld, n_g, C, band0, l15, lg, qf);
    as_band<CT, false>(p.d_o, g0, p.HC, n_g, C, band0, l15, lg, gf);
    const int ic = min(i, n_g - 1);
    const unsigned char *mrow = p.mask + p.mask_ptr[g] + (size_t)ic * (size_t)((p.pad_ptr[g + 1] - p.pad_ptr[g]) >> 3);
    const float mf = p.stats[((size_t)(n0 + ic) * p.H + h) * 2], inv = row_on ? p.stats[((size_t)(n0 + ic) * p.H + h) * 2 + 1] : 0.f;
    const float D = p.Dd[(size_t)(n0 + ic) * p.H + h];
    const int nkc = (n_g + HF_KC - 1) / HF_KC;
    f32x4 qacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) qacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < nkc; ++kc) {
        const int k0 = kc * HF_KC;
        __syncthreads();
        hf_stage<CT>(p.qkvs, e0 + p.HC + (size_t)k0 * ld, ld, n_g - k0, C, Ks, tid);
        hf_stage<CT>(p.qkvs, e0 + 2 * p.HC + (size_t)k0 * ld, ld, n_g - k0, C, Vs, tid);
        const unsigned long long b0 = *(const unsigned long long *)(mrow + (k0 >> 3)), b1 = *(const unsigned long long *)(mrow + (k0 >> 3) + 8);
        __syncthreads();
        if (band_on) {
#pragma unroll
            for (int tn = 0; tn < HF_TN; ++tn) {
                if (k0 + tn * 16 < n_g) {
                    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
                    const unsigned short *kp = Ks + (tn * 16 + l15) * pr + 4 * lg, *vp = Vs + (tn * 16 + l15) * pr + 4 * lg;
#pragma unroll
                    for (int kt = 0; kt < CT; ++kt) {
                        sa = AS_MFMA(*(const as_s16x4 *)(kp + kt * 16), qf[kt], sa);
                        da = AS_MFMA(*(const as_s16x4 *)(vp + kt * 16), gf[kt], da);
                    }
                    const unsigned nib = hf_nib(b0, b1, tn, lg);
                    float ds[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (k0 + tn * 16 + 4 * lg + r < n_g) && ((nib >> r) & 1u) && inv > 0.f;
                        const float pv = ok ? bf2f(f2bf(hf_exp(p.scale * sa[r] - mf) * inv)) : 0.f;
                        ds[r] = pv * (da[r] - D);
                    }
                    const as_s16x4 dsa = as_pack(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) qacc[ct] = AS_MFMA(as_tr(Ks, pr, tn * 16, ct * 16, l15, lg), dsa, qacc[ct]);
                }
            }
        }
    }
    if (band_on && row_on) {                  // on top of the remainder edges' dq (k_attn_irr_bwd_dst wrote it)
        float *dq = p.dY4 + e0 + (size_t)i * ld + 4 * lg;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            if (ct * 16 + 4 * lg < C) *(f32x4 *)(dq + ct * 16) += p.scale * qacc[ct];
    }
}

template <int CT>
__global__ __launch_bounds__(HF_NT) void k_hyb_bwd_kv(HybFlash p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short hf_lds[];
    constexpr int pr = CT * 16 + 8;
    unsigned short *Qs = hf_lds, *Gs = Qs + HF_KC * pr;
    float *st_m = (float *)(Gs + HF_KC * pr), *st_inv = st_m + HF_KC, *st_D = st_inv + HF_KC;
    unsigned char *mt = (unsigned char *)(st_D + HF_KC);              // [HF_KC queries][16 bytes: the 128 keys of this workgroup]
    const int rt = blockIdx.x % p.nrt, gh = blockIdx.x / p.nrt, h = gh % p.H, g = gh / p.H;
    const int n0 = p.gp[g], n_g = p.gp[g + 1] - n0, r0 = rt * HF_ROWS;
    if (r0 >= n_g) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int C = p.C, ld = 4 * p.HC;
    const size_t e0 = (size_t)n0 * ld + h * C, g0 = (size_t)n0 * p.HC + h * C;
    const int band0 = r0 + wid * 16, j = band0 + l15;                 // this lane's key
    const bool band_on = band0 < n_g;
    as_s16x4 kf[CT], vf[CT];
    as_band<CT, false>(p.qkvs, e0 + p.HC, ld, n_g, C, band0, l15, lg, kf);
    as_band<CT, false>(p.qkvs, e0 + 2 * p.HC, ld, n_g, C, band0, l15, lg, vf);
    const size_t mstride = (size_t)((p.pad_ptr[g + 1] - p.pad_ptr[g]) >> 3);
    const unsigned char *mbase = p.mask + p.mask_ptr[g] + (r0 >> 3);
    const int nqc = (n_g + HF_KC - 1) / HF_KC;
    f32x4 vacc[CT], kacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { vacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; kacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int qc = 0; qc < nqc; ++qc) {
        const int q0 = qc * HF_KC;
        __syncthreads();
        hf_stage<CT>(p.qkvs, e0 + (size_t)q0 * ld, ld, n_g - q0, C, Qs, tid);
        hf_stage<CT>(p.d_o, g0 + (size_t)q0 * p.HC, p.HC, n_g - q0, C, Gs, tid);
        if (tid < HF_KC) {
            const int iq = min(q0 + tid, n_g - 1);
            const bool on = q0 + tid < n_g;
            st_m[tid] = p.stats[((size_t)(n0 + iq) * p.H + h) * 2];
            st_inv[tid] = on ? p.stats[((size_t)(n0 + iq) * p.H + h) * 2 + 1] : 0.f;
            st_D[tid] = p.Dd[(size_t)(n0 + iq) * p.H + h];
            const unsigned long long *mr = (const unsigned long long *)(mbase + (size_t)iq * mstride);
            ((unsigned long long *)mt)[2 * tid] = mr[0];
            ((unsigned long long *)mt)[2 * tid + 1] = mr[1];
        }
        __syncthreads();
        if (band_on) {
#pragma unroll 2
            for (int ti = 0; ti < HF_TN; ++ti) {
                if (q0 + ti * 16 < n_g) {
                    f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
                    const unsigned short *qp = Qs + (ti * 16 + l15) * pr + 4 * lg, *gpp = Gs + (ti * 16 + l15) * pr + 4 * lg;
#pragma unroll
                    for (int kt = 0; kt < CT; ++kt) {
                        s4 = AS_MFMA(*(const as_s16x4 *)(qp + kt * 16), kf[kt], s4);
                        d4 = AS_MFMA(*(const as_s16x4 *)(gpp + kt * 16), vf[kt], d4);
                    }
                    const f32x4 m4 = *(const f32x4 *)(st_m + ti * 16 + 4 * lg), i4 = *(const f32x4 *)(st_inv + ti * 16 + 4 * lg),
                                D4 = *(const f32x4 *)(st_D + ti * 16 + 4 * lg);
                    float pv[4], dsv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned bit = (mt[(ti * 16 + 4 * lg + r) * 16 + ((wid * 16 + l15) >> 3)] >> (l15 & 7)) & 1u;
                        const bool ok = bit && j < n_g && i4[r] > 0.f;        // (queries beyond n_g carry inv = 0)
                        pv[r] = ok ? bf2f(f2bf(hf_exp(p.scale * s4[r] - m4[r]) * i4[r])) : 0.f;
                        dsv[r] = pv[r] * (d4[r] - D4[r]);
                    }
                    const as_s16x4 pa = as_pack(pv[0], pv[1], pv[2], pv[3]), dsa = as_pack(dsv[0], dsv[1], dsv[2], dsv[3]);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        vacc[ct] = AS_MFMA(as_tr(Gs, pr, ti * 16, ct * 16, l15, lg), pa, vacc[ct]);
                        kacc[ct] = AS_MFMA(as_tr(Qs, pr, ti * 16, ct * 16, l15, lg), dsa, kacc[ct]);
                    }
                }
            }
        }
    }
    if (band_on && j < n_g) {
        float *er = p.dY4 + e0 + (size_t)j * ld + 4 * lg;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            if (ct * 16 + 4 * lg < C) {
                *(f32x4 *)(er + p.HC + ct * 16) = p.scale * kacc[ct];
                *(f32x4 *)(er + 2 * p.HC + ct * 16) = vacc[ct];
            }
    }
}
#undef AS_MFMA

bool hybrid_flash_ok(const da_graph *g, int C, bool bfc) {
    const bool off = cfg().train_attn < 2;          // da_config.train_attn (DA_TRAIN_ATTN=1): hybrid graphs through the pair matrices
    return bfc && !off && (C == 32 || C == 144) && g->hybrid && !g->slot_node;
}
template <int CT>
static int hyb_flash_launch(int which, const HybFlash &a, int G, hipStream_t st) {
    constexpr int pr = CT * 16 + 8;
    const int lds = 2 * HF_KC * pr * 2 + (which == 2 ? 3 * HF_KC * 4 + HF_KC * 16 : 0);
    static bool attr[16] = {};
    int dev = 0;
    DA_CHECK_HIP(hipGetDevice(&dev));
    if (!attr[dev & 15]) {
        const int cap = 2 * HF_KC * pr * 2 + 3 * HF_KC * 4 + HF_KC * 16;
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_hyb_fwd<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_hyb_bwd_q<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_hyb_bwd_kv<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        attr[dev & 15] = true;
    }
    const int grid = G * a.H * a.nrt;
    if (which == 0) k_hyb_fwd<CT><<<grid, HF_NT, lds, st>>>(a);
    else if (which == 1) k_hyb_bwd_q<CT><<<grid, HF_NT, lds, st>>>(a);
    else k_hyb_bwd_kv<CT><<<grid, HF_NT, lds, st>>>(a);
    DA_LAUNCH_CHECK();
    return 0;
}
static HybFlash hyb_flash_params(const da_graph *g, int H, int C, const float *qkvs) {
    HybFlash a;
    a.qkvs = qkvs; a.res = nullptr; a.o = nullptr; a.d_o = nullptr; a.dY4 = nullptr; a.stats = nullptr; a.Dd = nullptr;
    a.gp = g->graph_ptr; a.pad_ptr = g->pad_ptr; a.mask = g->mask; a.mask_ptr = (const long long *)g->mask_ptr;
    a.irr_ptr = g->irr_row_ptr; a.irr_src = g->irr_col_src;
    a.H = H; a.C = C; a.HC = H * C; a.nrt = (g->max_graph_nodes + HF_ROWS - 1) / HF_ROWS; a.n_nodes = g->n_nodes;
    a.scale = 1.0f / sqrtf((float)C);
    return a;
}

#define DA_HYB_SWITCH(C, STMT4, STMT18)                                                     \
    switch ((C) / 8) {                                                                      \
        case 4: STMT4; break;                                                               \
        case 18: STMT18; break;                                                             \
        default: set_error("hybrid training attention: unsupported head width C=%d", (C)); return 1; \
    }

// Side stream of the flash-style hybrid route (round 5): everything that concerns the VIRTUAL rows -- their statistics, their
// initial output, their remainder edges (heavy rows by workgroups, the virtual nodes of small puzzles by waves), in the backward
// their dq and the dk | dv they collect as sources -- depends on the projections (and, in the backward, on k_hyb_rowdot) only, not
// on the real rows' flash kernels, and is a chain of small launches (64 virtual rows in the reference's scripted Batches).  It runs
// on a stream of the library beside the real rows' chain; fork and join by events, one context per (device, caller stream).
// DA_HYB_SIDE=0: one stream, the round-4 launch order.
struct HybSide { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static HybSide *hyb_side(hipStream_t caller) {
    if (!(cfg().train_side_streams & 2)) return nullptr;          // da_config.train_side_streams bit 1
    struct Slot { int dev; hipStream_t caller; HybSide ctx; };
    static std::mutex mu;
    static std::vector<Slot *> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (Slot *sl : slots)
        if (sl->dev == dev && sl->caller == caller) return &sl->ctx;
    if (slots.size() >= 64) return nullptr;
    Slot *sl = new Slot{dev, caller, HybSide()};
    HybSide &c = sl->ctx;
    const bool good = hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&c.join, hipEventDisableTiming) == hipSuccess;
    if (!good) { delete sl; return nullptr; }
    slots.push_back(sl);
    return &sl->ctx;
}

// forward: o = softmax over (regular edges U remainder edges) . v + skip (+ res); P (dense part) and stats kept
int hybrid_train_attn_fwd(const da_graph *g, int H, int C, const float *qkvs, const float *res, float *o, float *P, float *stats,
                          const long long *poff, const int32_t *node_graph, hipStream_t st, bool bfc) {
    const int n = g->n_nodes, nr = g->n_real, HC = H * C, G = g->n_graphs, mx = g->max_graph_nodes;
    if (hybrid_flash_ok(g, C, bfc)) {
        // flash-style route: no pair matrix (P is not touched and may be null)
        int rc;
        HybFlash a = hyb_flash_params(g, H, C, qkvs);
        a.res = res; a.o = o; a.stats = stats;
        const float scale_f = 1.0f / sqrtf((float)C);
        HybSide *sd = n > nr ? hyb_side(st) : nullptr;
        hipStream_t sv = sd ? sd->s : st;                     // the virtual rows' chain
        StreamJoin side_guard;                               // error exits below still join the virtual rows' stream
        if (sd) {
            DA_CHECK_HIP(hipEventRecord(sd->fork, st));       // (the projections are there)
            DA_CHECK_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
            side_guard.arm(sd->s, st, sd->join);
        }
        auto virtual_chain = [&]() -> int {
            // virtual rows: statistics over their (remainder-only) edges, o = skip (+ residual), then their edges -- rows with more
            // than IRR_HEAVY edges by workgroups, the others (the virtual nodes of small puzzles) by waves
            k_pair_rows_hybrid<<<gridsz((size_t)(n - nr) * H * 64), 256, 0, sv>>>(0, nr, n, nr, H, C, g->graph_ptr, g->pad_ptr, node_graph, poff, g->mask,
                                                                               (const long long *)g->mask_ptr, g->irr_row_ptr, g->irr_col_src, qkvs, nullptr,
                                                                               nullptr, stats, nullptr);
            k_init_out<<<gridsz((size_t)(n - nr) * HC), 256, 0, sv>>>(n - nr, HC, qkvs + (size_t)nr * 4 * HC, res ? res + (size_t)nr * HC : nullptr, o + (size_t)nr * HC);
            const int grid_v = (int)(((size_t)(n - nr) * 64 + 255) / 256);
            DA_HYB_SWITCH(C, (k_attn_irr_fwd<4, false><<<grid_v, 256, 0, sv>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale_f, nr)),
                          (k_attn_irr_fwd<18, false><<<grid_v, 256, 0, sv>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale_f, nr)))
            DA_HYB_SWITCH(C, (k_attn_irr_fwd<4, true><<<n - nr, 64 * IrrNW<4>::v, 0, sv>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale_f, 0)),
                          (k_attn_irr_fwd<18, true><<<n - nr, 64 * IrrNW<18>::v, 0, sv>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale_f, 0)))
            DA_LAUNCH_CHECK();
            return 0;
        };
        if (sd && (rc = virtual_chain())) return rc;          // side stream: enqueued first, runs beside the real rows' kernels
        if ((rc = C == 32 ? hyb_flash_launch<2>(0, a, G, st) : hyb_flash_launch<9>(0, a, G, st))) return rc;
        if (!sd && n > nr && (rc = virtual_chain())) return rc;
        const int grid_r = (int)(((size_t)nr * 64 + 255) / 256);      // the real rows' remainder edges (rows [0, nr))
        DA_HYB_SWITCH(C, (k_attn_irr_fwd<4, false><<<grid_r, 256, 0, st>>>(nr, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale_f, 0)),
                      (k_attn_irr_fwd<18, false><<<grid_r, 256, 0, st>>>(nr, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale_f, 0)))
        DA_LAUNCH_CHECK();
        if (sd) {
            DA_CHECK_HIP(side_guard.join());
        }
        return 0;
    }
    GGemm s;
    s.bfc = bfc;
    s.A = {(float *)qkvs, 0, 4 * HC, C};
    s.B = {(float *)qkvs + HC, 0, 4 * HC, C};
    s.C = {P, 1, 0, 0};
    s.transA = 0; s.transB = 1; s.dimM = 0; s.dimN = 0; s.dimK = C; s.alpha = 1.0f / sqrtf((float)C); s.accumulate = 0;
    s.H = H; s.gp = g->graph_ptr; s.poff = poff;
    int rc;
    if ((rc = ggemm(s, G, H, mx, st))) return rc;
    k_pair_rows_hybrid<<<gridsz((size_t)n * H * 64), 256, 0, st>>>(0, 0, n, nr, H, C, g->graph_ptr, g->pad_ptr, node_graph, poff, g->mask,
                                                                    (const long long *)g->mask_ptr, g->irr_row_ptr, g->irr_col_src, qkvs, P,
                                                                    nullptr, stats, nullptr);
    k_init_out<<<gridsz((size_t)n * HC), 256, 0, st>>>(n, HC, qkvs, res, o);
    DA_LAUNCH_CHECK();
    GGemm pv;
    pv.bfc = bfc;
    pv.A = {P, 1, 0, 0};
    pv.B = {(float *)qkvs + 2 * HC, 0, 4 * HC, C};
    pv.C = {o, 0, HC, C};
    pv.transA = 0; pv.transB = 0; pv.dimM = 0; pv.dimN = C; pv.dimK = 0; pv.alpha = 1.0f; pv.accumulate = 1;
    pv.H = H; pv.gp = g->graph_ptr; pv.poff = poff;
    if ((rc = ggemm(pv, G, H, mx, st))) return rc;
    const float scale = 1.0f / sqrtf((float)C);
    const int grid = (int)(((size_t)n * 64 + 255) / 256);
    DA_HYB_SWITCH(C, (k_attn_irr_fwd<4, false><<<grid, 256, 0, st>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale, 0)),
                  (k_attn_irr_fwd<18, false><<<grid, 256, 0, st>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale, 0)))
    if (n > nr) DA_HYB_SWITCH(C, (k_attn_irr_fwd<4, true><<<n - nr, 64 * IrrNW<4>::v, 0, st>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale, 0)),
                  (k_attn_irr_fwd<18, true><<<n - nr, 64 * IrrNW<18>::v, 0, st>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, stats, o, scale, 0)))
    DA_LAUNCH_CHECK();
    return 0;
}

// backward: dY4 = [dq | dk | dv | d_o]
int hybrid_train_attn_bwd(const da_graph *g, int H, int C, const float *qkvs, const float *d_o, const float *P, float *dP,
                          const float *stats, float *Dd, float *dY4, const long long *poff, const int32_t *node_graph, hipStream_t st, bool bfc,
                          const float *o, const float *res) {
    const int n = g->n_nodes, nr = g->n_real, HC = H * C, G = g->n_graphs, mx = g->max_graph_nodes;
    const float scale = 1.0f / sqrtf((float)C);
    const int grid = (int)(((size_t)n * 64 + 255) / 256);
    int rc;
    if (hybrid_flash_ok(g, C, bfc)) {
        // flash-style route (o = the layer's forward output, res = its residual operand: D = rowsum(dO o (o - skip - res)))
        constexpr int DTOT = 1;
        HybFlash a = hyb_flash_params(g, H, C, qkvs);
        a.res = res; a.o = (float *)o; a.d_o = d_o; a.dY4 = dY4; a.stats = (float *)stats; a.Dd = Dd;
        k_hyb_rowdot<<<(unsigned)(((size_t)n * 64 + 255) / 256), 256, 0, st>>>(a);
        DA_LAUNCH_CHECK();
        HybSide *sd = n > nr ? hyb_side(st) : nullptr;
        hipStream_t sv = sd ? sd->s : st;
        StreamJoin side_guard;                               // error exits below still join the virtual rows' stream
        if (sd) {
            DA_CHECK_HIP(hipEventRecord(sd->fork, st));       // (D of every row is there)
            DA_CHECK_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
            side_guard.arm(sd->s, st, sd->join);
        }
        const int grid_r = (int)(((size_t)nr * 64 + 255) / 256), grid_v = (int)(((size_t)(n - nr) * 64 + 255) / 256);
        // the virtual rows' chain: dk | dv start from zero; as destinations their dq (and skip gradient), as sources the dk | dv of
        // their edges into the real rows -- none of it touches a real row's gradient
        auto virtual_chain = [&]() -> int {
            k_zero_kv_grad<<<gridsz((size_t)(n - nr) * 2 * HC), 256, 0, sv>>>(nr, n, HC, dY4);
            DA_HYB_SWITCH(C, (k_attn_irr_bwd_dst<4, false><<<grid_v, 256, 0, sv>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, nr)),
                          (k_attn_irr_bwd_dst<18, false><<<grid_v, 256, 0, sv>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, nr)))
            DA_HYB_SWITCH(C, (k_attn_irr_bwd_dst<4, true><<<n - nr, 64 * IrrNW<4>::v, 0, sv>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, 0)),
                          (k_attn_irr_bwd_dst<18, true><<<n - nr, 64 * IrrNW<18>::v, 0, sv>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, 0)))
            DA_HYB_SWITCH(C, (k_attn_irr_bwd_src<4, false><<<grid_v, 256, 0, sv>>>(n, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, nr)),
                          (k_attn_irr_bwd_src<18, false><<<grid_v, 256, 0, sv>>>(n, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, nr)))
            DA_HYB_SWITCH(C, (k_attn_irr_bwd_src<4, true><<<n - nr, 64 * IrrNW<4>::v, 0, sv>>>(n, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, 0)),
                          (k_attn_irr_bwd_src<18, true><<<n - nr, 64 * IrrNW<18>::v, 0, sv>>>(n, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, 0)))
            DA_LAUNCH_CHECK();
            return 0;
        };
        if (sd && (rc = virtual_chain())) return rc;
        // the real rows' chain.  Remainder, destination side: dq of the remainder edges (D is already the row's total), skip gradient
        DA_HYB_SWITCH(C, (k_attn_irr_bwd_dst<4, false><<<grid_r, 256, 0, st>>>(nr, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, 0)),
                      (k_attn_irr_bwd_dst<18, false><<<grid_r, 256, 0, st>>>(nr, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, 0)))
        DA_LAUNCH_CHECK();
        if ((rc = C == 32 ? hyb_flash_launch<2>(1, a, G, st) : hyb_flash_launch<9>(1, a, G, st))) return rc;      // dQ += (regular edges)
        if ((rc = C == 32 ? hyb_flash_launch<2>(2, a, G, st) : hyb_flash_launch<9>(2, a, G, st))) return rc;      // dK, dV (regular edges)
        DA_HYB_SWITCH(C, (k_attn_irr_bwd_src<4, false><<<grid_r, 256, 0, st>>>(nr, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, 0)),
                      (k_attn_irr_bwd_src<18, false><<<grid_r, 256, 0, st>>>(nr, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, 0)))
        DA_LAUNCH_CHECK();
        if (!sd && n > nr && (rc = virtual_chain())) return rc;
        if (sd) {
            DA_CHECK_HIP(side_guard.join());
        }
        return 0;
    }
    constexpr int DTOT = 0;
    GGemm q;
    q.bfc = bfc;
    q.H = H; q.gp = g->graph_ptr; q.poff = poff; q.accumulate = 0;
    // dV (regular edges) = P^T dO ; virtual rows' dk | dv start from zero
    q.A = {(float *)P, 1, 0, 0}; q.B = {(float *)d_o, 0, HC, C}; q.C = {dY4 + 2 * HC, 0, 4 * HC, C};
    q.transA = 1; q.transB = 0; q.dimM = 0; q.dimN = C; q.dimK = 0; q.alpha = 1.0f;
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    if (n > nr) { k_zero_kv_grad<<<gridsz((size_t)(n - nr) * 2 * HC), 256, 0, st>>>(nr, n, HC, dY4); DA_LAUNCH_CHECK(); }
    // dP = dO V^T ; D (dense part)
    q.A = {(float *)d_o, 0, HC, C}; q.B = {(float *)qkvs + 2 * HC, 0, 4 * HC, C}; q.C = {dP, 1, 0, 0};
    q.transA = 0; q.transB = 1; q.dimM = 0; q.dimN = 0; q.dimK = C; q.alpha = 1.0f;
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    k_pair_rows_hybrid<<<gridsz((size_t)n * H * 64), 256, 0, st>>>(2, 0, n, nr, H, C, g->graph_ptr, g->pad_ptr, node_graph, poff, g->mask,
                                                                    (const long long *)g->mask_ptr, g->irr_row_ptr, g->irr_col_src, qkvs,
                                                                    (float *)P, dP, nullptr, Dd);
    // remainder, destination side: D total, dq of the remainder edges, skip gradient
    DA_HYB_SWITCH(C, (k_attn_irr_bwd_dst<4, false><<<grid, 256, 0, st>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, 0)),
                  (k_attn_irr_bwd_dst<18, false><<<grid, 256, 0, st>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, 0)))
    if (n > nr) DA_HYB_SWITCH(C, (k_attn_irr_bwd_dst<4, true><<<n - nr, 64 * IrrNW<4>::v, 0, st>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, 0)),
                  (k_attn_irr_bwd_dst<18, true><<<n - nr, 64 * IrrNW<18>::v, 0, st>>>(n, nr, g->irr_row_ptr, g->irr_col_src, H, HC, qkvs, d_o, stats, dY4, Dd, scale, DTOT, 0)))
    // dS = P o (dP - D) over the regular edges
    k_pair_rows_hybrid<<<gridsz((size_t)n * H * 64), 256, 0, st>>>(1, 0, n, nr, H, C, g->graph_ptr, g->pad_ptr, node_graph, poff, g->mask,
                                                                    (const long long *)g->mask_ptr, g->irr_row_ptr, g->irr_col_src, qkvs,
                                                                    (float *)P, dP, nullptr, Dd);
    DA_LAUNCH_CHECK();
    // dQ += scale dS K (on top of the remainder's dq) ; dK = scale dS^T Q
    q.A = {dP, 1, 0, 0}; q.B = {(float *)qkvs + HC, 0, 4 * HC, C}; q.C = {dY4, 0, 4 * HC, C};
    q.transA = 0; q.transB = 0; q.dimM = 0; q.dimN = C; q.dimK = 0; q.alpha = scale; q.accumulate = 1;
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    q.A = {dP, 1, 0, 0}; q.B = {(float *)qkvs, 0, 4 * HC, C}; q.C = {dY4 + HC, 0, 4 * HC, C};
    q.transA = 1; q.transB = 0; q.dimM = 0; q.dimN = C; q.dimK = 0; q.alpha = scale; q.accumulate = 0;
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    // remainder, source side: += dk, dv
    DA_HYB_SWITCH(C, (k_attn_irr_bwd_src<4, false><<<grid, 256, 0, st>>>(n, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, 0)),
                  (k_attn_irr_bwd_src<18, false><<<grid, 256, 0, st>>>(n, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, 0)))
    if (n > nr) DA_HYB_SWITCH(C, (k_attn_irr_bwd_src<4, true><<<n - nr, 64 * IrrNW<4>::v, 0, st>>>(n, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, 0)),
                  (k_attn_irr_bwd_src<18, true><<<n - nr, 64 * IrrNW<18>::v, 0, st>>>(n, nr, g->out_ptr, g->out_dst, H, HC, qkvs, d_o, stats, Dd, dY4, scale, 0)))
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
