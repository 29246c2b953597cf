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
//
// k_ggemm: C(m, n) = alpha * sum_k A(m, k) B(k, n) (+ C), 64 x 64 tile, 16 k per stage, 4 waves as 2 x 2,
// v_mfma_f32_16x16x4_f32.  Operands may be transposed views; both tiles are staged k-major ([k][m] and
// [k][n], row stride 80 floats) so the single-float MFMA fragments (row = lane & 15, k = lane >> 4) are
// conflict-free 4-byte LDS reads whatever the memory orientation; the global side always moves 16 bytes
// per lane along the contiguous axis.
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

static int ggemm(const GGemm &p, int G, int H, int maxn, hipStream_t st) {
    const int Mx = p.dimM ? p.dimM : maxn, Nx = p.dimN ? p.dimN : maxn;
    k_ggemm<<<dim3((Nx + 63) / 64, (Mx + 63) / 64, G * H), 256, 0, st>>>(p);
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
                         const long long *poff, const int32_t *node_graph, hipStream_t st) {
    const int n = g->n_nodes, HC = H * C, G = g->n_graphs, mx = g->max_graph_nodes;
    GGemm s;
    s.A = {(float *)qkvs, 0, 4 * HC, C};
    s.B = {(float *)qkvs + HC, 0, 4 * HC, C};
    s.C = {P, 1, 0, 0};
    s.transA = 0; s.transB = 1; s.dimM = 0; s.dimN = 0; s.dimK = C; s.alpha = 1.0f / sqrtf((float)C); s.accumulate = 0;
    s.H = H; s.gp = g->graph_ptr; s.poff = poff;
    int rc;
    if ((rc = ggemm(s, G, H, mx, st))) return rc;
    k_pair_rows<<<gridsz((size_t)n * H * 64), 256, 0, st>>>(0, n, H, g->dense == 2, g->graph_ptr, node_graph, poff, P, nullptr);
    k_init_out<<<gridsz((size_t)n * HC), 256, 0, st>>>(n, HC, qkvs, res, o);
    DA_LAUNCH_CHECK();
    GGemm pv;
    pv.A = {P, 1, 0, 0};
    pv.B = {(float *)qkvs + 2 * HC, 0, 4 * HC, C};
    pv.C = {o, 0, HC, C};
    pv.transA = 0; pv.transB = 0; pv.dimM = 0; pv.dimN = C; pv.dimK = 0; pv.alpha = 1.0f; pv.accumulate = 1;
    pv.H = H; pv.gp = g->graph_ptr; pv.poff = poff;
    return ggemm(pv, G, H, mx, st);
}

// backward: dY4 = [dq | dk | dv | d_o] from d_o [n, HC], the saved P and the projection buffer
int dense_train_attn_bwd(const da_graph *g, int H, int C, const float *qkvs, const float *d_o, const float *P, float *dP,
                         float *dY4, const long long *poff, const int32_t *node_graph, hipStream_t st) {
    const int n = g->n_nodes, HC = H * C, G = g->n_graphs, mx = g->max_graph_nodes;
    const float scale = 1.0f / sqrtf((float)C);
    int rc;
    GGemm q;
    q.H = H; q.gp = g->graph_ptr; q.poff = poff; q.accumulate = 0;
    // dV = P^T dO
    q.A = {(float *)P, 1, 0, 0}; q.B = {(float *)d_o, 0, HC, C}; q.C = {dY4 + 2 * HC, 0, 4 * HC, C};
    q.transA = 1; q.transB = 0; q.dimM = 0; q.dimN = C; q.dimK = 0; q.alpha = 1.0f;
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    // dP = dO V^T
    q.A = {(float *)d_o, 0, HC, C}; q.B = {(float *)qkvs + 2 * HC, 0, 4 * HC, C}; q.C = {dP, 1, 0, 0};
    q.transA = 0; q.transB = 1; q.dimM = 0; q.dimN = 0; q.dimK = C; q.alpha = 1.0f;
    if ((rc = ggemm(q, G, H, mx, st))) return rc;
    // dS = P o (dP - rowsum(P o dP)), in place over dP
    k_pair_rows<<<gridsz((size_t)n * H * 64), 256, 0, st>>>(1, n, H, 0, g->graph_ptr, node_graph, poff, (float *)P, dP);
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

}  // namespace da
