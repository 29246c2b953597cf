// Sparse graph attention: one PyG TransformerConv message/aggregate pass over a CSR-by-
// destination graph (Transformer_GNN.py:32,38; exophormer_gnn.py:203,205).
//
// One 64-lane wavefront owns one destination node and ALL heads: lane l holds the
// contiguous slice [l*EPL, (l+1)*EPL) of the H*C-wide q/k/v rows (EPL = C/8, so the 8 lanes
// [8h, 8h+8) are head h).  Per incoming edge the wave reads one K row and one V row of the
// source node -- two fully coalesced H*C*sizeof(T) segments -- reduces the per-head dot
// product with three xor-shuffles inside the 8-lane group, and folds it into an online
// softmax (running max / sum per head, replicated across the group) and the V accumulator.
// Multi-edges are separate softmax terms and nodes without incoming edges yield 0 + skip,
// exactly as torch_geometric.utils.softmax + scatter-add do (oracle/pyg_restatement.py).
// HBM-bound: algorithmic bytes per edge = 2*H*C*sizeof(T) (+ 4 B index).
#include "da_common.h"
#include "da_internal.h"

namespace da {

typedef __attribute__((ext_vector_type(4))) unsigned int csr_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int csr_u32x2;

// EPL contiguous channels of one lane as floats, fetched with the widest loads the slice allows (a lane's
// slice starts at a multiple of its own size inside a 16-byte aligned row: 36-byte slices of a 144-wide bf16
// head are 4-byte aligned, 8-byte slices of a 32-wide one 8-byte aligned, ...).  One 2-byte load per channel
// made a K + V row cost 36 load instructions per edge.
template <typename T, int EPL>
__device__ __forceinline__ void ld_row(const T *p, float (&v)[EPL]) {
    constexpr int B = EPL * (int)sizeof(T), NW = (B + 3) / 4;
    if constexpr (B % 4 != 0) {                 // odd bf16 slices (EPL = 1, 13): one load per channel
#pragma unroll
        for (int i = 0; i < EPL; ++i) v[i] = ldf(p + i);
        return;
    }
    unsigned w[NW];
    if constexpr (B % 16 == 0) {
#pragma unroll
        for (int i = 0; i < B / 16; ++i) { const csr_u32x4 t = *((const csr_u32x4 *)p + i); w[4 * i] = t[0]; w[4 * i + 1] = t[1]; w[4 * i + 2] = t[2]; w[4 * i + 3] = t[3]; }
    } else if constexpr (B % 8 == 0) {
#pragma unroll
        for (int i = 0; i < B / 8; ++i) { const csr_u32x2 t = *((const csr_u32x2 *)p + i); w[2 * i] = t[0]; w[2 * i + 1] = t[1]; }
    } else if constexpr (B % 12 == 0) {
        // 36-byte slices of a 144-wide bf16 head (only 4-byte aligned): three 12-byte loads (global_load_dwordx3) instead of nine 4-byte ones --
        // a K + V row used to cost a wave 18 vector-memory instructions of 256 bytes each (round 6, profiles/r06/r06_pmc_gather_calibration.txt)
        struct __attribute__((packed, aligned(4))) U3 { unsigned a, b, c; };
#pragma unroll
        for (int i = 0; i < B / 12; ++i) { const U3 t = *((const U3 *)p + i); w[3 * i] = t.a; w[3 * i + 1] = t.b; w[3 * i + 2] = t.c; }
    } else {
        struct __attribute__((packed, aligned(4))) U4 { unsigned a, b, c, d; };
#pragma unroll
        for (int i = 0; i < B / 16; ++i) { const U4 t = *((const U4 *)p + i); w[4 * i] = t.a; w[4 * i + 1] = t.b; w[4 * i + 2] = t.c; w[4 * i + 3] = t.d; }
#pragma unroll
        for (int i = 4 * (B / 16); i < NW; ++i) w[i] = *((const unsigned *)p + i);
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) v[i] = __builtin_bit_cast(float, w[i]);
    } else {
#pragma unroll
        for (int i = 0; i < EPL / 2; ++i) { v[2 * i] = bf2f((bf16_t)(w[i] & 0xffff)); v[2 * i + 1] = bf2f((bf16_t)(w[i] >> 16)); }
    }
}

// A lane's slice of a row AS LOADED (bf16: two channels per register), converted element by element at its use: rows in flight cost half the
// registers of their float form, so twice as many edges fit in flight (round 6).  fp32 rows are what they were.
template <typename T, int EPL> struct RowRegs {
    static constexpr int NW = (EPL * (int)sizeof(T) + 3) / 4;
    unsigned w[NW];
    __device__ __forceinline__ void load(const T *p) {
        constexpr int B = EPL * (int)sizeof(T);
        if constexpr (B % 16 == 0) {
#pragma unroll
            for (int i = 0; i < B / 16; ++i) { const csr_u32x4 t = *((const csr_u32x4 *)p + i); w[4 * i] = t[0]; w[4 * i + 1] = t[1]; w[4 * i + 2] = t[2]; w[4 * i + 3] = t[3]; }
        } else if constexpr (B % 8 == 0) {
#pragma unroll
            for (int i = 0; i < B / 8; ++i) { const csr_u32x2 t = *((const csr_u32x2 *)p + i); w[2 * i] = t[0]; w[2 * i + 1] = t[1]; }
        } else if constexpr (B % 12 == 0) {
            struct __attribute__((packed, aligned(4))) U3 { unsigned a, b, c; };
#pragma unroll
            for (int i = 0; i < B / 12; ++i) { const U3 t = *((const U3 *)p + i); w[3 * i] = t.a; w[3 * i + 1] = t.b; w[3 * i + 2] = t.c; }
        } else if constexpr (B % 4 == 0) {
            // other 4-byte aligned slices (52 bytes: 26 bf16 channels of a 104-wide head at four lanes per head): 16-byte loads that are only
            // 4-byte aligned (the hardware takes them) + the rest as dwords
            struct __attribute__((packed, aligned(4))) U4 { unsigned a, b, c, d; };
#pragma unroll
            for (int i = 0; i < B / 16; ++i) { const U4 t = *((const U4 *)p + i); w[4 * i] = t.a; w[4 * i + 1] = t.b; w[4 * i + 2] = t.c; w[4 * i + 3] = t.d; }
#pragma unroll
            for (int i = 4 * (B / 16); i < NW; ++i) w[i] = *((const unsigned *)p + i);
        } else {                                // odd bf16 slices (EPL = 1, 13): one load per channel, kept as floats' bit patterns is not possible -- two per word
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const unsigned lo = ((const unsigned short *)p)[2 * i];
                const unsigned hi = 2 * i + 1 < EPL ? ((const unsigned short *)p)[2 * i + 1] : 0u;
                w[i] = lo | (hi << 16);
            }
        }
    }
    __device__ __forceinline__ float get(int x) const {
        if constexpr (sizeof(T) == 4) return __builtin_bit_cast(float, w[x]);
        else return bf2f((bf16_t)((x & 1) ? (w[x >> 1] >> 16) : (w[x >> 1] & 0xffff)));
    }
};

template <typename T, int EPL>
__global__ __launch_bounds__(256) void k_attn_csr(int n_nodes, const int32_t *__restrict__ row_ptr,
                                                  const int32_t *__restrict__ col_src,
                                                  const int32_t *__restrict__ edge_id, int H, int HC,
                                                  const T *__restrict__ qkvs, const T *__restrict__ residual,
                                                  int act, T *__restrict__ out, float *__restrict__ alpha,
                                                  float *__restrict__ stats, float scale) {
    const int lane = threadIdx.x & 63;
    const int i = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (i >= n_nodes) return;
    const size_t ld = (size_t)4 * HC;
    const int off = lane * EPL;
    float q[EPL], acc[EPL];
    const T *qp = qkvs + (size_t)i * ld + off;
    ld_row<T, EPL>(qp, q);                          // (wide loads, as for the K / V rows: the per-channel form was 18 two-byte loads at C = 144)
    // the epilogue's skip (+ residual) slices are requested HERE, with the query row: one memory round trip less at the end of every row
    // (narrow heads only: at C = 144 the 18 - 36 registers they would hold through the edge loop cost a resident wave)
    constexpr bool EARLY = EPL <= 8;
    RowRegs<T, EPL> skr, rsr;
    if (EARLY) {
        skr.load(qkvs + (size_t)i * ld + 3 * (size_t)HC + off);
        if (residual) rsr.load(residual + (size_t)i * HC + off);
    }
#pragma unroll
    for (int x = 0; x < EPL; ++x) { q[x] *= scale; acc[x] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const int beg = row_ptr[i], end = row_ptr[i + 1];
    const int head = lane >> 3;
    // The walk used to be one dependent chain per edge -- index load, then the K / V row loads it addresses, then the softmax update -- a
    // latency chain (0.157 of the HBM peak with the whole K | V table in the Infinity Cache).  Now: (1) the row's source
    // indices arrive by ONE coalesced load per 64 edges (lane e holds edge e's source; v_readlane hands it to the wave as a scalar, so
    // the row addresses are SGPR base + per-lane offset); (2) the K and V rows of U edges are requested together, before any is consumed;
    // (3) the U scores enter the running softmax in one update (one rescale per U edges instead of one per edge).
    constexpr int U = EPL <= 4 ? 8 : (EPL <= 8 ? (sizeof(T) == 2 ? 8 : 4) : (sizeof(T) == 2 ? 4 : 2));
    for (int e0 = beg; e0 < end; e0 += 64) {
        const int cnt = min(64, end - e0);
        const int myj = col_src[e0 + min(lane, cnt - 1)];
        for (int u0 = 0; u0 < cnt; u0 += U) {
            RowRegs<T, EPL> kk[U], vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = __builtin_amdgcn_readlane(myj, min(u0 + u, cnt - 1));          // (edges past the end re-read the last one)
                const T *kp = qkvs + (size_t)j * ld + HC + off;
                kk[u].load(kp);
                vv[u].load(kp + HC);
            }
            float sc[U];
            float mn = m;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float s = 0.f;
#pragma unroll
                for (int x = 0; x < EPL; ++x) s = fmaf(q[x], kk[u].get(x), s);
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                const bool ok = u0 + u < cnt;                                                  // wave-uniform
                if (ok && alpha && (lane & 7) == 0) {
                    const size_t eid = edge_id ? (size_t)edge_id[e0 + u0 + u] : (size_t)(e0 + u0 + u);
                    alpha[eid * H + head] = s;        // raw score; normalised in the second pass
                }
                sc[u] = ok ? s : -INFINITY;
                mn = fmaxf(mn, sc[u]);
            }
            const float corr = expf(m - mn);           // (m = -inf on the first group: 0; mn is finite from the first real edge on)
            l *= corr;
#pragma unroll
            for (int x = 0; x < EPL; ++x) acc[x] *= corr;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pe = expf(sc[u] - mn);     // (padding: exp(-inf) = 0)
                l += pe;
#pragma unroll
                for (int x = 0; x < EPL; ++x) acc[x] = fmaf(pe, vv[u].get(x), acc[x]);
            }
            m = mn;
        }
    }
    const float inv = (end > beg) ? 1.0f / (l + 1e-16f) : 0.f;
    if (stats && (lane & 7) == 0) {               // training: softmax statistics for the backward kernels
        stats[((size_t)i * H + head) * 2] = m;
        stats[((size_t)i * H + head) * 2 + 1] = inv;
    }
    T *op = out + (size_t)i * HC + off;
    if (!EARLY) {
        skr.load(qkvs + (size_t)i * ld + 3 * (size_t)HC + off);
        if (residual) rsr.load(residual + (size_t)i * HC + off);
    }
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        float v = acc[x] * inv + skr.get(x);
        if (residual) v += rsr.get(x);
        stf(op + x, apply_act(v, act));
    }
    if (alpha && (lane & 7) == 0) {
        for (int e = beg; e < end; ++e) {
            const size_t eid = edge_id ? (size_t)edge_id[e] : (size_t)e;
            const float s = alpha[eid * H + head];
            alpha[eid * H + head] = expf(s - m) * inv;
        }
    }
}

// C = 32 (the hidden layers): TWO destination rows per wave.  A 32-wide head's K row is 64 bytes: with one row per wave a lane moves 8 bytes per load
// and an edge costs the wave two 512-byte instructions; here half a wave owns a destination (four lanes per head, eight channels = 16 bytes per lane in
// bf16), so one instruction fetches the K (or V) rows of TWO edges -- half the vector-memory instructions per edge, 1 KB each -- and the per-head
// dot product closes over four lanes (two shuffles).  The two halves walk their own edge lists (their lengths differ: the wave runs to the longer
// one, the shorter half's surplus trips are masked); a half's source indices come from its own coalesced index load through ds_bpermute.
// Same arithmetic per edge group as k_attn_csr (one softmax update per U edges).
// EPL = C / 4 channels per lane.  Round 6, later: also the 104-wide heads of the 3D variant (D = 832: BASELINE configuration 4's last layer) -- at one
// row per wave their 13-channel lane slices are 26 bytes, an odd number of bf16 pairs, fetched one 2-byte load per channel (26 instructions per
// K + V row); at four lanes per head the slices are 52 bytes = three 16-byte loads and a dword.
template <typename T, int EPL>
__global__ __launch_bounds__(256) void k_attn_csr2(int n_nodes, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_src,
                                                   const int32_t *__restrict__ edge_id, int H, int HC, const T *__restrict__ qkvs,
                                                   const T *__restrict__ residual, int act, T *__restrict__ out, float *__restrict__ alpha,
                                                   float *__restrict__ stats, float scale) {
    constexpr int U = EPL <= 8 ? (sizeof(T) == 2 ? 8 : 4) : (sizeof(T) == 2 ? 4 : 2);
    const int lane = threadIdx.x & 63, hl = lane & 31, r = lane >> 5;
    const int wave = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int i = 2 * wave + r;
    if (2 * wave >= n_nodes) return;
    const bool row_on = i < n_nodes;
    const int ic = row_on ? i : n_nodes - 1;
    const size_t ld = (size_t)4 * HC;
    const int off = hl * EPL, head = hl >> 2;
    float q[EPL], acc[EPL];
    ld_row<T, EPL>(qkvs + (size_t)ic * ld + off, q);
    RowRegs<T, EPL> skr, rsr;
    skr.load(qkvs + (size_t)ic * ld + 3 * (size_t)HC + off);
    if (residual) rsr.load(residual + (size_t)ic * HC + off);
#pragma unroll
    for (int x = 0; x < EPL; ++x) { q[x] *= scale; acc[x] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const int beg = row_ptr[ic], end = row_on ? row_ptr[ic + 1] : beg;
    const int deg = end - beg;
    const int degmax = __builtin_amdgcn_readfirstlane(max(deg, __shfl_xor(deg, 32)));
    for (int e0 = 0; e0 < degmax; e0 += 32) {
        const int cnt = min(32, max(deg - e0, 0));                              // this half's edges in the chunk (uniform within the half)
        const int cmax = __builtin_amdgcn_readfirstlane(max(cnt, __shfl_xor(cnt, 32)));
        const int myj = cnt > 0 ? col_src[beg + e0 + min(hl, cnt - 1)] : 0;
        for (int u0 = 0; u0 < cmax; u0 += U) {
            RowRegs<T, EPL> kk[U], vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = __shfl(myj, r * 32 + min(u0 + u, max(cnt - 1, 0)));           // (edges past the half's end re-read its last one; an empty half row 0)
                const T *kp = qkvs + (size_t)j * ld + HC + off;
                kk[u].load(kp);
                vv[u].load(kp + HC);
            }
            float sc[U];
            float mn = m;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float s = 0.f;
#pragma unroll
                for (int x = 0; x < EPL; ++x) s = fmaf(q[x], kk[u].get(x), s);
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                const bool ok = u0 + u < cnt;
                if (ok && alpha && (hl & 3) == 0) {
                    const size_t eid = edge_id ? (size_t)edge_id[beg + e0 + u0 + u] : (size_t)(beg + e0 + u0 + u);
                    alpha[eid * H + head] = s;        // raw score; normalised in the second pass
                }
                sc[u] = ok ? s : -INFINITY;
                mn = fmaxf(mn, sc[u]);
            }
            // (a half without an edge so far keeps m = mn = -inf: exp(-inf - -inf) would be NaN -- its state is all zero, skip the update)
            const bool live = mn > -INFINITY;
            const float corr = live ? expf(m - mn) : 1.f;
            l *= corr;
#pragma unroll
            for (int x = 0; x < EPL; ++x) acc[x] *= corr;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pe = live ? expf(sc[u] - mn) : 0.f;
                l += pe;
#pragma unroll
                for (int x = 0; x < EPL; ++x) acc[x] = fmaf(pe, vv[u].get(x), acc[x]);
            }
            m = mn;
        }
    }
    if (!row_on) return;
    const float inv = (end > beg) ? 1.0f / (l + 1e-16f) : 0.f;
    if (stats && (hl & 3) == 0) {
        stats[((size_t)i * H + head) * 2] = m;
        stats[((size_t)i * H + head) * 2 + 1] = inv;
    }
    T *op = out + (size_t)i * HC + off;
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        float v = acc[x] * inv + skr.get(x);
        if (residual) v += rsr.get(x);
        stf(op + x, apply_act(v, act));
    }
    if (alpha && (hl & 3) == 0) {
        for (int e = beg; e < end; ++e) {
            const size_t eid = edge_id ? (size_t)edge_id[e] : (size_t)e;
            const float s = alpha[eid * H + head];
            alpha[eid * H + head] = expf(s - m) * inv;
        }
    }
}

template <typename T>
static int launch_t(int n_nodes, const int32_t *row_ptr, const int32_t *col_src, const int32_t *edge_id, int H,
                    int C, const T *qkvs, const T *residual, int act, T *out, float *alpha, float *stats, hipStream_t st) {
    const int HC = H * C;
    const float scale = 1.0f / sqrtf((float)C);
    const int grid = (int)(((size_t)n_nodes * 64 + 255) / 256);
#define DA_CSR_CASE(E)                                                                                   \
    case E:                                                                                              \
        k_attn_csr<T, E><<<grid, 256, 0, st>>>(n_nodes, row_ptr, col_src, edge_id, H, HC, qkvs, residual, act, \
                                               out, alpha, stats, scale);                                \
        break;
    if ((C == 32 || C == 104) && H == 8 && DA_XENV("DA_CSR_TWO_ROWS", 1)) {          // two destination rows per wave (k_attn_csr2)
        const int grid2 = (int)(((size_t)((n_nodes + 1) / 2) * 64 + 255) / 256);
        if (C == 32) k_attn_csr2<T, 8><<<grid2, 256, 0, st>>>(n_nodes, row_ptr, col_src, edge_id, H, HC, qkvs, residual, act, out, alpha, stats, scale);
        else k_attn_csr2<T, 26><<<grid2, 256, 0, st>>>(n_nodes, row_ptr, col_src, edge_id, H, HC, qkvs, residual, act, out, alpha, stats, scale);
        DA_LAUNCH_CHECK();
        return 0;
    }
    switch (C / 8) {
        DA_CSR_CASE(1) DA_CSR_CASE(2) DA_CSR_CASE(4) DA_CSR_CASE(8) DA_CSR_CASE(13) DA_CSR_CASE(16) DA_CSR_CASE(18)
        default:
            set_error("da_attn_csr: unsupported head width C=%d", C);
            return 1;
    }
#undef DA_CSR_CASE
    DA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Hybrid mode, the rows the masked MFMA attention does not own.  Real nodes are finished inside
// k_attn_dense<.., MASKED> (its epilogue folds their few remainder edges into the same softmax); the virtual
// nodes of the exophormer arch (exophormer_gnn.py:183-200) have NO regular edge and are attended here over
// the remainder CSR, reading Q / K / V from the head-major padded layouts the projection scattered.
constexpr int HEAVY_WAVES = 16;

// Rows with a long remainder list -- the exophormer virtual nodes: ~n_g incoming edges each, most of them
// the duplicated virtual->virtual pairs -- get a whole 16-wave workgroup: wave w walks edges w, w+16, ...,
// the 16 partial softmax states are merged through LDS.  (One wave per row made the 8 virtual rows of a
// graph the critical path of the whole layer: 560 us at any batch size.)
template <typename T, int EPL, bool WIDE = false>
__global__ __launch_bounds__(1024) void k_attn_csr_cont_heavy(int n_nodes, int n_real, const int32_t *__restrict__ row_ptr,
                                                              const int32_t *__restrict__ col_src,
                                                              const int32_t *__restrict__ row_map, int H, int C,
                                                              size_t n_pad, const T *__restrict__ Q,
                                                              const T *__restrict__ K, const T *__restrict__ V,
                                                              const T *__restrict__ skip, const T *__restrict__ residual,
                                                              int act, T *__restrict__ out, float scale,
                                                              const float *__restrict__ mult) {
    extern __shared__ float hsm[];                       // [HEAVY_WAVES][64 * EPL] acc, then [HEAVY_WAVES][64][2] (m, l)
    const int i = n_real + blockIdx.x;
    if (i >= n_nodes) return;
    const int beg = row_ptr[i], end = row_ptr[i + 1];
    // (round 6: light rows too -- sixteen waves with a handful of edges each; they used to return here and leave the row to a second launch,
    //  k_attn_csr_cont, 5 - 6 us per layer on the critical path of the scripted Batches)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int head = lane >> 3, sub = (lane & 7) * EPL;
    const size_t hb = (size_t)head * n_pad;
    const size_t si = hb + (size_t)row_map[i];
    float q[EPL], acc[EPL];
    float m = -INFINITY, l = 0.f;
    ld_row<T, EPL>(Q + si * C + sub, q);
#pragma unroll
    for (int x = 0; x < EPL; ++x) { q[x] *= scale; acc[x] = 0.f; }
    // four edges per trip: their index -> slot -> K / V row loads are independent, so one round of memory
    // latency serves four softmax updates (the walk used to be one dependent chain per edge)
    // (C = 144 in fp32: four edges' K and V rows would be 144 registers beside q and acc under this kernel's 128 -- two there, no spills)
    // (bf16 rows stay packed in flight: RowRegs.  WIDE -- eight edges in flight at C = 32, experiments build, DA_CONT_WIDE=1: 106 registers instead of
    //  70, and this kernel runs BESIDE the masked attention of the real rows: measured on configuration 3, see launch_cont_t)
    constexpr int U = (EPL > 8 && sizeof(T) == 4) ? 2 : ((WIDE && EPL <= 4 && sizeof(T) == 2) ? 8 : 4);
    // (round 6) the trip's source slots -- col_src -> row_map: two dependent index loads -- are fetched ONE TRIP AHEAD, under the K / V loads and the
    // softmax updates of the current trip: a trip is then one memory round trip (the rows), not three
    size_t sjn[U];
    bool okn[U];
    float wgtn[U];
    auto fetch_idx = [&](int e0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * HEAVY_WAVES;
            okn[u] = e < end;
            sjn[u] = hb + (size_t)row_map[col_src[okn[u] ? e : beg]];
            wgtn[u] = mult ? mult[okn[u] ? e : beg] : 1.0f;
        }
    };
    constexpr bool PFI = EPL <= 8;          // (144-wide heads: no registers left for a second set of indices under this kernel's 128)
    if (PFI && beg + wv < end) fetch_idx(beg + wv);
    for (int e0 = beg + wv; e0 < end; e0 += HEAVY_WAVES * U) {
        size_t sj[U];
        bool ok[U];
        float wgt[U];                                        // multiplicity of the (aggregated) edge, 1 without `mult`
        if (!PFI) fetch_idx(e0);
#pragma unroll
        for (int u = 0; u < U; ++u) { sj[u] = sjn[u]; ok[u] = okn[u]; wgt[u] = wgtn[u]; }
        RowRegs<T, EPL> kk[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { kk[u].load(K + sj[u] * C + sub); vv[u].load(V + sj[u] * C + sub); }
        if (PFI && e0 + HEAVY_WAVES * U < end) fetch_idx(e0 + HEAVY_WAVES * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;                            // wave-uniform
            float s = 0.f;
#pragma unroll
            for (int x = 0; x < EPL; ++x) s = fmaf(q[x], kk[u].get(x), s);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            const float mn = fmaxf(m, s);
            const float corr = expf(m - mn);
            const float pr = expf(s - mn) * wgt[u];
            l = l * corr + pr;
#pragma unroll
            for (int x = 0; x < EPL; ++x) acc[x] = fmaf(pr, vv[u].get(x), acc[x] * corr);
            m = mn;
        }
    }
    float *sacc = hsm + (size_t)wv * 64 * EPL, *sml = hsm + (size_t)HEAVY_WAVES * 64 * EPL;
#pragma unroll
    for (int x = 0; x < EPL; ++x) sacc[lane * EPL + x] = acc[x];
    sml[(wv * 64 + lane) * 2] = m;
    sml[(wv * 64 + lane) * 2 + 1] = l;
    __syncthreads();
    if (wv != 0) return;
    float M = -INFINITY;
    for (int w = 0; w < HEAVY_WAVES; ++w) M = fmaxf(M, sml[(w * 64 + lane) * 2]);
    float L = 0.f;
#pragma unroll
    for (int x = 0; x < EPL; ++x) acc[x] = 0.f;
    for (int w = 0; w < HEAVY_WAVES; ++w) {
        const float mw = sml[(w * 64 + lane) * 2], lw = sml[(w * 64 + lane) * 2 + 1];
        if (!(lw > 0.f)) continue;
        const float f = expf(mw - M);
        L = fmaf(lw, f, L);
#pragma unroll
        for (int x = 0; x < EPL; ++x) acc[x] = fmaf(hsm[(size_t)w * 64 * EPL + lane * EPL + x], f, acc[x]);
    }
    const float inv = L > 0.f ? 1.0f / (L + 1e-16f) : 0.f;
    const size_t o = (size_t)i * H * C + (size_t)head * C + sub;
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
        float v = acc[x] * inv + ldf(skip + o + x);
        if (residual) v += ldf(residual + o + x);
        stf(out + o + x, apply_act(v, act));
    }
}

template <typename T>
static int launch_cont_t(int n_nodes, int n_real, const int32_t *rp, const int32_t *cs, const int32_t *row_map, int H, int C,
                         int n_pad, const DenseLayout &L, const T *residual, int act, T *out, hipStream_t st, const float *mult) {
    if (n_nodes <= n_real) return 0;
    const float scale = L.q_prescaled ? 0.6931471805599453f : 1.0f / sqrtf((float)C);      // pre-scaled Q: q . k is in log2 units
#ifdef DA_EXPERIMENTS
    constexpr bool DA_CONT_WIDE_BUILT = true;
#else
    constexpr bool DA_CONT_WIDE_BUILT = false;
#endif
    [[maybe_unused]] const bool wide_ = DA_XENV("DA_CONT_WIDE", 0) != 0;
#define DA_CONT_CASE(E)                                                                                          \
    case E:                                                                                                      \
        if (n_nodes > n_real) {                                                                                  \
            const int lds = HEAVY_WAVES * 64 * (E + 2) * 4;                                                      \
            static bool attr = false;                                                                            \
            if (!attr && lds > 48 * 1024) {                                                                      \
                DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_csr_cont_heavy<T, E>,                      \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));              \
                attr = true;                                                                                     \
            }                                                                                                    \
            if (wide_ && E == 4)                                                                                 \
                k_attn_csr_cont_heavy<T, E, DA_CONT_WIDE_BUILT><<<n_nodes - n_real, 1024, lds, st>>>(            \
                    n_nodes, n_real, rp, cs, row_map, H, C, (size_t)n_pad, (const T *)L.Q, (const T *)L.K,       \
                    (const T *)L.Vt, (const T *)L.S, residual, act, out, scale, mult);                           \
            else                                                                                                 \
            k_attn_csr_cont_heavy<T, E><<<n_nodes - n_real, 1024, lds, st>>>(                                    \
                n_nodes, n_real, rp, cs, row_map, H, C, (size_t)n_pad, (const T *)L.Q, (const T *)L.K,           \
                (const T *)L.Vt, (const T *)L.S, residual, act, out, scale, mult);                               \
        }                                                                                                        \
        break;
    switch (C / 8) {
        DA_CONT_CASE(4) DA_CONT_CASE(18)
        default:
            set_error("da_attn_csr_cont: unsupported head width C=%d", C);
            return 1;
    }
#undef DA_CONT_CASE
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_attn_csr_cont(int prec, int n_nodes, int n_real, const int32_t *irr_row_ptr, const int32_t *irr_col_src,
                         const int32_t *row_map, int heads, int C, int n_pad, const DenseLayout &L,
                         const void *residual, int act, void *out, hipStream_t st, const float *mult) {
    if (n_nodes <= 0) return 0;
    DA_REQUIRE(heads == 8 && C % 8 == 0, "da_attn_csr_cont: heads must be 8 and C a multiple of 8");
    if (prec == DA_PREC_BF16)
        return launch_cont_t<bf16_t>(n_nodes, n_real, irr_row_ptr, irr_col_src, row_map, heads, C, n_pad, L,
                                     (const bf16_t *)residual, act, (bf16_t *)out, st, mult);
    return launch_cont_t<float>(n_nodes, n_real, irr_row_ptr, irr_col_src, row_map, heads, C, n_pad, L,
                                (const float *)residual, act, (float *)out, st, mult);
}

// ---- TINY COMPLETE graphs (at most 32 pieces: the 3D variant's 20-fragment objects, BASELINE configuration 4) at head widths the matrix-core
// kernels do not take (C = 104, D = 832).  As an edge list such a layer re-reads every K | V row once per destination -- 20 times -- out of the
// L2: 333 MB per launch for 17 MB of rows, 46 us.  Here one workgroup owns a graph: its K | V rows are staged in LDS once, half a wave per
// destination (four lanes per head, C / 4 channels per lane: k_attn_csr2's layout and arithmetic -- groups of U keys, one softmax update per group),
// keys = every node of the graph (self loops iff !nodiag).  Same PyG TransformerConv row as k_attn_csr (reference efficient_gat_3d.py -> the
// backbone's last layer, Transformer_GNN.py:38-46).
template <typename T, int EPL>
__global__ __launch_bounds__(256) void k_attn_tiny(const int32_t *__restrict__ graph_ptr, int nodiag, int H, int HC, const T *__restrict__ qkvs,
                                                   const T *__restrict__ residual, int act, T *__restrict__ out, float scale) {
    constexpr int U = EPL <= 8 ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    T *kv = (T *)tsm;                                   // [n][2 HC]: K row | V row of every node of the graph
    const int g = blockIdx.x, node0 = graph_ptr[g], n = graph_ptr[g + 1] - node0;
    if (n <= 0) return;
    const size_t ld = (size_t)4 * HC;
    {
        const int cpr = 2 * HC * (int)sizeof(T) / 16;   // 16-byte chunks per staged row
        // eight chunks of a thread requested together (one chunk per trip was a chain of n cpr / 256 = 16 - 26 dependent round trips: 30 of the
        // kernel's first 46 us)
        for (int base = threadIdx.x; base < n * cpr; base += 256 * 8) {
            csr_u32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = min(base + 256 * u, n * cpr - 1);
                const int r_ = idx / cpr, c_ = idx - r_ * cpr;
                t[u] = *(const csr_u32x4 *)((const unsigned char *)(qkvs + (size_t)(node0 + r_) * ld + HC) + (size_t)c_ * 16);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (base + 256 * u < n * cpr) ((csr_u32x4 *)tsm)[base + 256 * u] = t[u];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, hl = lane & 31, hw = (int)(threadIdx.x >> 6) * 2 + (lane >> 5);
    const int off = hl * EPL, head = hl >> 2;
    (void)head; (void)H;
    for (int r = hw; r < n; r += 8) {                    // (the two halves of a wave run their own trip counts: every shuffle below stays inside four lanes)
        const size_t i = (size_t)(node0 + r);
        float q[EPL], acc[EPL];
        ld_row<T, EPL>(qkvs + i * ld + off, q);
        RowRegs<T, EPL> skr, rsr, qr;
        qr.load(qkvs + i * ld + off);          // bf16: the query slice as loaded -- scores by v_dot2_f32_bf16 on the packed words (half the instructions of unpack + fma; the kernel is VALU-bound)
        skr.load(qkvs + i * ld + 3 * (size_t)HC + off);
        if (residual) rsr.load(residual + i * HC + off);
#pragma unroll
        for (int x = 0; x < EPL; ++x) { q[x] *= scale; acc[x] = 0.f; }
        float m = -INFINITY, l = 0.f;
        for (int j0 = 0; j0 < n; j0 += U) {
            RowRegs<T, EPL> kk[U], vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const T *kp = kv + (size_t)min(j0 + u, n - 1) * 2 * HC + off;
                kk[u].load(kp);
                vv[u].load(kp + HC);
            }
            float sc[U];
            float mn = m;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float s_ = 0.f;
                if constexpr (sizeof(T) == 2 && EPL % 2 == 0) {
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
#pragma unroll
                    for (int x = 0; x < EPL / 2; ++x)
                        s_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_, qr.w[x]), __builtin_bit_cast(bf16x2_, kk[u].w[x]), s_, false);
                    s_ *= scale;
                } else {
#pragma unroll
                    for (int x = 0; x < EPL; ++x) s_ = fmaf(q[x], kk[u].get(x), s_);
                }
                s_ += __shfl_xor(s_, 1);
                s_ += __shfl_xor(s_, 2);
                const bool ok = j0 + u < n && !(nodiag && j0 + u == r);
                sc[u] = ok ? s_ : -INFINITY;
                mn = fmaxf(mn, sc[u]);
            }
            const bool live = mn > -INFINITY;                // (nothing but masked keys so far: the state stays zero)
            const float corr = live ? expf(m - mn) : 1.f;
            l *= corr;
#pragma unroll
            for (int x = 0; x < EPL; ++x) acc[x] *= corr;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pe = live ? expf(sc[u] - mn) : 0.f;
                l += pe;
#pragma unroll
                for (int x = 0; x < EPL; ++x) acc[x] = fmaf(pe, vv[u].get(x), acc[x]);
            }
            m = mn;
        }
        const float inv = l > 0.f ? 1.0f / (l + 1e-16f) : 0.f;          // (a 1-piece graph without self loop: no key, the row is its skip)
        T *op = out + i * HC + off;
#pragma unroll
        for (int x = 0; x < EPL; ++x) {
            float v = acc[x] * inv + skr.get(x);
            if (residual) v += rsr.get(x);
            stf(op + x, apply_act(v, act));
        }
    }
}

// returns 0 = launched, -1 = shape not covered (the caller walks the edge list)
int launch_attn_tiny(int prec, int n_graphs, int max_graph_nodes, const int32_t *graph_ptr, int nodiag, int heads, int C, const void *qkvs,
                     const void *residual, int act, void *out, hipStream_t st) {
    if (n_graphs <= 0) return 0;
    const size_t lds = (size_t)max_graph_nodes * 2 * heads * C * esize(prec);
    if (heads != 8 || C != 104 || max_graph_nodes > 32 || lds > (size_t)160 * 1024 || !graph_ptr || !DA_XENV("DA_ATTN_TINY", 1)) return -1;
    const float scale = 1.0f / sqrtf((float)C);
    const int HC = heads * C;
#define DA_TINY(TT)                                                                                                                    \
    do {                                                                                                                               \
        static bool attr = false;                                                                                                      \
        if (!attr) { DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_tiny<TT, 26>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
        k_attn_tiny<TT, 26><<<n_graphs, 256, lds, st>>>(graph_ptr, nodiag, heads, HC, (const TT *)qkvs, (const TT *)residual, act, (TT *)out, scale); \
    } while (0)
    if (prec == DA_PREC_BF16) DA_TINY(bf16_t); else DA_TINY(float);
#undef DA_TINY
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_attn_csr(int prec, int n_nodes, const int32_t *row_ptr, const int32_t *col_src, const int32_t *edge_id,
                    int heads, int C, const void *qkvs, const void *residual, int act, void *out, float *alpha,
                    float *stats, hipStream_t st) {
    if (n_nodes <= 0) return 0;
    DA_REQUIRE(heads == 8, "da_attn_csr: heads must be 8 (got %d)", heads);
    DA_REQUIRE(C % 8 == 0, "da_attn_csr: C %% 8 != 0 (C=%d)", C);
    if (prec == DA_PREC_BF16)
        return launch_t<bf16_t>(n_nodes, row_ptr, col_src, edge_id, heads, C, (const bf16_t *)qkvs,
                                (const bf16_t *)residual, act, (bf16_t *)out, alpha, stats, st);
    return launch_t<float>(n_nodes, row_ptr, col_src, edge_id, heads, C, (const float *)qkvs,
                           (const float *)residual, act, (float *)out, alpha, stats, st);
}

}  // namespace da
