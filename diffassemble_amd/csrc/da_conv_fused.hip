// EXPERIMENTS BUILD ONLY (DA_EXPERIMENTS=1 python __graft_entry__.py; __graft_entry__.EXPERIMENT_SOURCES): projection + attention of a hidden conv in one kernel: a tie at 256 workgroups, a loss below (round 2).
// One hidden TransformerConv layer (Transformer_GNN.py:32,38 -> PyG TransformerConv, C = 32 channels per head)
// on COMPLETE graphs as ONE kernel: the fused Q | K | V | skip projection of a (graph, head) AND its
// attention, with nothing but the layer's input x and its output touching HBM.
//
//   out[i, 32h:32h+32] = act( softmax_j(q_i . k_j / sqrt(32)) v_j + skip_i ),   [q|k|v|skip]_i = W_h x_i + b_h
//
// Why (rocprof, 32 puzzles of 900 pieces, bf16): the two-kernel form spends 35 us per layer in the projection
// GEMM (M = 28 800, K = 256, N = 1024: 59 MB of head-major Q / K / V / skip written and read back) and 61 us in
// the attention, whose 128-query workgroups re-stream K_h / V_h eight times.  Here one workgroup owns one
// (graph, head): K_h and V_h^T of all n <= 992 keys are projected ONCE into LDS (n x 64 B + 32 x 2n B, ~120 KB at
// n = 900) and stay there; every wave then walks 32-query slabs: Q^T and skip^T of the slab straight out of the
// matrix cores into the operand / accumulator registers the attention needs, 29 blocks of 32 keys against the
// resident K / V^T, epilogue in registers.  HBM traffic per layer: x (read through L2 by the 8 heads of a graph,
// which share an XCD) and out.
//
// Register-level layouts (v_mfma_f32_32x32x16_bf16: A lane l = row l&31, 8 k-values 8(l>>5)..; B lane l = column
// l&31, same k-values; D lane l = column l&31, register r = row (r&3) + 8(r>>2) + 4(l>>5)):
//   K^T = W_k x^T      A = W_k rows (channel), B = x rows (node)  -> lane (node, half) holds channels
//                      ch(r) = (r&3) + 8(r>>2) + 4 half.  Registers 8t..8t+7 are written as ONE 16-byte chunk
//                      (2t + half) of K's LDS row: the channel order inside a row is whatever the S^T MFMA of the
//                      attention consumes, because
//   Q^T = W_q x^T      comes out in the same layout, and its registers 8t..8t+7 (as bf16) ARE the B operand of
//                      k-step t of S^T = K Q^T (the dot product only needs K and Q to agree on the order).
//   V   = x W_v^T      A = x rows (node), B = W_v rows (channel) -> lane (channel, half) holds 4-node runs, written
//                      as 8-byte pieces of row `channel` of V^T [32][n]: the PV operand is then a plain 16-byte read.
//   skip^T = W_s x^T   same layout as O^T = V^T P^T: added in registers.
// x fragments are loaded from global memory in operand shape (16 B per lane per k-step; x_g is L2 resident), the four
// 32 x KIN weight blocks of the head sit in LDS in fragment order (W_k, W_v during the projection phase, W_q, W_s
// during the attention phase: same 2 x 16 KB region).  Softmax as in da_attn_dense.hip: MFMA row rho is fed key
// pi(rho) so that a lane holds 16 consecutive keys of one query, lane-local running max / sum, rescale only when
// the max moves by more than 2^8, exp2 with the scale folded in.
//
// bf16 only (fp32 K / V do not fit in LDS: the fp32 parity mode keeps the two-kernel path), H = 8, C = 32,
// KIN in {128, 256}.
#include <stdlib.h>

#include "da_common.h"
#include "da_internal.h"

namespace da {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct ConvFusedParams {
    const bf16_t *x;            // [N][ldx]
    int ldx;
    const bf16_t *W;            // [4 * 256][KIN]: rows Q | K | V | skip, feature f = 32 h + c inside a block
    const float *bias;          // [4 * 256]
    bf16_t *out;                // [N][256]
    const int32_t *graph_ptr;
    int n_graphs, act, nodiag;
    float sc;                   // log2(e) / sqrt(32)
    int debug;                  // DA_FUSED_PROBE builds (tools/fused_probe.py): 1 no phase 2, 2 no softmax, 4 no PV, 8 no QK,
                                // 16 no x loads in phase 2, 32 one key block, 64 no phase 1
};

#ifdef DA_FUSED_PROBE
#define DA_FDBG(...) __VA_ARGS__
#else
#define DA_FDBG(...)
#endif

__device__ __forceinline__ f32x16 mfma(const u32x4 &a, const u32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ u32x4 pack8(const f32x16 &v, int r0) {
    bf16x8 b;
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = (__bf16)v[r0 + e];
    return __builtin_bit_cast(u32x4, b);
}

// LDS regions: K [n32][64 B] (16-byte chunks XOR-swizzled by (key >> 2) & 3), V^T [32][VS] (VS = 2 n32 + 16: an odd
// number of 16-byte slots), W fragments [2][KS][64 lanes][16 B].
// Occupancy: ONE workgroup per CU (the K / V^T images of a 900-piece graph are 120 KB), so the waves per SIMD come from
// the workgroup itself: 16 waves (4 per SIMD) at <= 128 VGPRs -- the attention loop is bound by VALU issue and
// dependency latency (ablation, tools/fused_probe.py), which is what more resident waves hide.  A wave owns at most
// MAXS = 2 query slabs (launch_conv_fused picks the wave count so that this holds).
template <int KIN>
__global__ __launch_bounds__(1024, 4) void k_conv_fused(ConvFusedParams p) {
    constexpr int KS = KIN / 16, MAXS = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // all 8 heads of a graph on ONE XCD (workgroup b runs on XCD b % 8): they share x_g in its L2
    const int nblk = gridDim.x;
    const int bid = (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8);
    const int g = bid >> 3, h = bid & 7;
    const int node0 = p.graph_ptr[g], n_g = p.graph_ptr[g + 1] - node0;
    if (n_g <= 0) return;
    const int nslab = (n_g + 31) >> 5, n32 = nslab * 32;
    const int VS = 2 * n32 + 16;
    unsigned char *sK = smem;
    unsigned char *sV = smem + (size_t)n32 * 64;
    unsigned char *sW = sV + (size_t)32 * VS;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), NW = blockDim.x >> 6;

    // ---- weight blocks of this head into LDS in fragment order: piece (m, s, lane') = W_m[ch = lane' & 31][16 s + 8 (lane' >> 5) ..+8]
    auto stage_w = [&](int blk0, int blk1) {
        for (int it = tid; it < 2 * KS * 64; it += blockDim.x) {
            const int m = it / (KS * 64), r = it - m * (KS * 64), s = r >> 6, l2 = r & 63;
            const int blk = m ? blk1 : blk0;
            const bf16_t *src = p.W + ((size_t)blk * 256 + 32 * h + (l2 & 31)) * KIN + 16 * s + 8 * (l2 >> 5);
            *(u32x4 *)(sW + (size_t)it * 16) = *(const u32x4 *)src;
        }
    };
    // bias of the 16 channels a lane holds in the transposed (channel x node) layouts, as an accumulator initialiser
    auto bias16 = [&](int blk) {
        f32x16 b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = *(const f32x4 *)(p.bias + blk * 256 + 32 * h + 8 * j + 4 * half);
            b[4 * j] = v[0]; b[4 * j + 1] = v[1]; b[4 * j + 2] = v[2]; b[4 * j + 3] = v[3];
        }
        return b;
    };
    // Two projections of one 32-node slab against the two weight blocks currently in LDS, accumulated onto a0 / a1:
    //   a0 += W_0 . x^T (channel x node);  a1 += second_node_major ? x . W_1^T (node x channel) : W_1 . x^T.
    // x fragments come straight from global memory in operand shape -- lane (node, half), k-step s -> x[node][16 s + 8 half
    // ..+8] -- eight k-steps (32 VGPRs) at a time.
    auto project = [&](int slab, f32x16 &a0, f32x16 &a1, bool second_node_major) {
        const int node = min(32 * slab + i, n_g - 1);
        const bf16_t *row = p.x + (size_t)(node0 + node) * p.ldx + 8 * half;
        // the weight fragments are the same for every slab: without this the compiler hoists all 2 x KS reads (128 VGPRs)
        // out of the slab loop and spills
        unsigned lo = (unsigned)lane * 16u;
        asm volatile("" : "+v"(lo));
#pragma unroll 1
        for (int hk = 0; hk < KS; hk += 8) {
            // (the address is laundered so that the second half's loads are not hoisted above the first half's MFMAs:
            // 64 VGPRs of x fragments at once do not fit beside the accumulators at 4 waves per SIMD)
            const bf16_t *rowh = row + 16 * hk;
            asm volatile("" : "+v"(rowh));
            u32x4 xf[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) xf[s] = *(const u32x4 *)(rowh + 16 * s);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 w0 = *(const u32x4 *)(sW + (size_t)(hk + s) * 1024 + lo);
                const u32x4 w1 = *(const u32x4 *)(sW + (size_t)(KS + hk + s) * 1024 + lo);
                a0 = mfma(w0, xf[s], a0);
                a1 = second_node_major ? mfma(xf[s], w1, a1) : mfma(w1, xf[s], a1);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep the W fragment reads from piling up in registers
            }
        }
    };

    // =========================== phase 1: K_h and V_h^T of the whole graph into LDS ===========================
    stage_w(1, 2);
    __syncthreads();
    DA_FDBG(if (!(p.debug & 64)))
    {
        const float bv = p.bias[2 * 256 + 32 * h + i];
        for (int slab = wid; slab < nslab; slab += NW) {
            f32x16 aK = bias16(1), aV;
#pragma unroll
            for (int r = 0; r < 16; ++r) aV[r] = bv;
            project(slab, aK, aV, true);           // K^T[ch][node], V[node][ch]
            // K: this lane = node 32 slab + i; rows beyond the graph are zero (their scores are masked anyway)
            const int key = 32 * slab + i;
            const bool kin = key < n_g;
#pragma unroll
            for (int r = 0; r < 16; ++r) aK[r] = kin ? aK[r] : 0.f;
            const int sw = (key >> 2) & 3;
            *(u32x4 *)(sK + (size_t)key * 64 + (((0 + half) ^ sw) << 4)) = pack8(aK, 0);
            *(u32x4 *)(sK + (size_t)key * 64 + (((2 + half) ^ sw) << 4)) = pack8(aK, 8);
            // V^T: this lane = channel i; register r = node 32 slab + (r&3) + 8 (r>>2) + 4 half; zero beyond the graph
            // (a masked probability is exactly 0, and 0 x garbage must stay 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nd = 32 * slab + 8 * j + 4 * half;
                bf16x4 b;
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = (__bf16)((nd + e < n_g) ? aV[4 * j + e] : 0.f);
                *(u32x2 *)(sV + (size_t)i * VS + (size_t)nd * 2) = __builtin_bit_cast(u32x2, b);
            }
        }
    }
    __syncthreads();
    stage_w(0, 3);
    __syncthreads();

    // ====== phase 1b: Q^T (-> B operand of S^T, kept in registers) and skip^T of this wave's slabs.  skip goes to the
    // OUTPUT rows as bf16 right away (the two-kernel path rounds it to bf16 too) and is read back by the same lanes in the
    // epilogue: 16 registers per slab that the attention loop does not have to carry.
    u32x4 qf[MAXS][2];                           // bf16: registers 8t..8t+7 of the (channel x node) accumulator
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
        const int slab = wid + j * NW;
        if (slab < nslab) {
            f32x16 aQ = bias16(0), aS = bias16(3);
            project(slab, aQ, aS, false);
            qf[j][0] = pack8(aQ, 0); qf[j][1] = pack8(aQ, 8);
            const int qidx = 32 * slab + i;
            if (qidx < n_g) {
                bf16_t *dst = p.out + (size_t)(node0 + qidx) * 256 + 32 * h + 4 * half;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    bf16x4 b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = (__bf16)aS[4 * jj + e];
                    *(u32x2 *)(dst + 8 * jj) = __builtin_bit_cast(u32x2, b);
                }
            }
        }
    }
    DA_FDBG(if (p.debug & 1) return;)

    // =========================== phase 2: attention of this wave's query slabs ===========================
    const int pi_i = (i & 3) + 4 * ((i >> 3) & 3) + 16 * ((i >> 2) & 1);       // key fed to MFMA row i
    // a 32-key block is 2 KB of K rows; (32 kb + pi_i) >> 2 == pi_i >> 2 (mod 4): the swizzle does not depend on kb
    const unsigned char *kp0 = sK + (size_t)pi_i * 64 + (((0 + half) ^ ((pi_i >> 2) & 3)) << 4);
    const unsigned char *kp1 = sK + (size_t)pi_i * 64 + (((2 + half) ^ ((pi_i >> 2) & 3)) << 4);
    const unsigned char *vrow = sV + (size_t)i * VS + (size_t)(16 * half) * 2;
    constexpr float THR = 16384.0f;              // block sums above this re-centre the running max (see below)
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
        const int slab = wid + j * NW;
        if (slab >= nslab) break;
        const int qidx = 32 * slab + i;                   // this lane's query (index inside the graph)
        f32x16 O;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] = 0.f;
        // Online softmax WITHOUT a per-block max: p = exp2((s - m) sc) is formed against the running reference m
        // directly; only when a block's sum says the reference is stale (first block: m = -1e30 gives +inf; later: some
        // score more than ~2^10 above m) is the block's true max taken and O / l rescaled.  Softmax is shift invariant,
        // so any m works as long as nothing overflows: the max tree (8 v_max3 + compare per block) leaves the common path.
        float m = -1e30f, l = 0.f;
        int nkb = nslab;
        DA_FDBG(if (p.debug & 32) nkb = 1;)
        for (int kb = 0; kb < nkb; ++kb) {
            const u32x4 k0 = *(const u32x4 *)(kp0 + (size_t)kb * 2048);
            const u32x4 k1 = *(const u32x4 *)(kp1 + (size_t)kb * 2048);
            const u32x4 v0 = *(const u32x4 *)(vrow + (size_t)kb * 64);
            const u32x4 v1 = *(const u32x4 *)(vrow + (size_t)kb * 64 + 16);
            f32x16 s = mfma(k0, qf[j][0], (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
            s = mfma(k1, qf[j][1], s);
            // this lane holds keys 32 kb + 16 half + r, r = 0..15, of query qidx
            const int kbase = 32 * kb + 16 * half;
            if (32 * kb + 32 > n_g || (p.nodiag && kb == slab)) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + r >= n_g || (p.nodiag && kbase + r == qidx)) s[r] = -INFINITY;
            }
            float ms = m * p.sc;
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[r] = __builtin_amdgcn_exp2f(fmaf(s[r], p.sc, -ms));
            float bsum = ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7])) +
                         (((pr[8] + pr[9]) + (pr[10] + pr[11])) + ((pr[12] + pr[13]) + (pr[14] + pr[15])));
            if (__any(!(bsum < THR))) {
                // rare: re-centre on this block's max (both halves of a query agree through the cross-half exchange)
                const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]);
                const float a2 = fmaxf(fmaxf(s[6], s[7]), s[8]), a3 = fmaxf(fmaxf(s[9], s[10]), s[11]);
                const float a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
                const float mloc = fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), s[15]));
                const float mnew = fmaxf(m, fmaxf(mloc, __shfl_xor(mloc, 32)));     // >= -1e30: finite
                const float corr = __builtin_amdgcn_exp2f((m - mnew) * p.sc);
                m = mnew;
                l *= corr;
#pragma unroll
                for (int r = 0; r < 16; ++r) O[r] *= corr;
                ms = m * p.sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] = __builtin_amdgcn_exp2f(fmaf(s[r], p.sc, -ms));
                bsum = ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7])) +
                       (((pr[8] + pr[9]) + (pr[10] + pr[11])) + ((pr[12] + pr[13]) + (pr[14] + pr[15])));
            }
            l += bsum;
            bf16x8 pf0, pf1;
#pragma unroll
            for (int e = 0; e < 8; ++e) { pf0[e] = (__bf16)pr[e]; pf1[e] = (__bf16)pr[8 + e]; }
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf0, O, 0, 0, 0);
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, O, 0, 0, 0);
        }
        // ---- epilogue in registers: normalise (PyG: sum + 1e-16), + skip, activation, 8-byte stores
        const float lt = l + __shfl_xor(l, 32);
        const float inv = lt > 0.f ? 1.0f / (lt + 1e-16f) : 0.f;
        if (qidx < n_g) {
            bf16_t *dst = p.out + (size_t)(node0 + qidx) * 256 + 32 * h + 4 * half;
            u32x2 skv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) skv[jj] = *(const u32x2 *)(dst + 8 * jj);     // skip^T, written in phase 1b by this lane
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bf16x4 sv = __builtin_bit_cast(bf16x4, skv[jj]);
                bf16x4 b;
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = (__bf16)apply_act(fmaf(O[4 * jj + e], inv, (float)sv[e]), p.act);
                *(u32x2 *)(dst + 8 * jj) = __builtin_bit_cast(u32x2, b);
            }
        }
    }
}

// OPT-IN (DA_CONV_FUSED=1).  Measured on MI355X, 32 puzzles of 900 pieces, bf16, per hidden layer: this kernel 94-96 us
// against 35 us (projection GEMM) + 61 us (k_attn_dense) = 96 us for the two-kernel path -- a tie at 256 workgroups and
// a loss below (8 puzzles: 82 vs 40 us; one workgroup per (graph, head) cannot fill the chip).  Ablation
// (tools/fused_probe.py) and SQ counters (profiles/r02/pmc_fused_vs_two_kernel.txt): 41 us go to the projection phases,
// which wait on operand-shaped global loads of x (2/3 of the wave cycles in s_waitcnt), 53 us to the attention loop, which
// executes the same 19 M VALU instructions as k_attn_dense but with one workgroup per CU.  Kept as the measured record
// of VERDICT r01 direction 4(iii); the default path is the two-kernel one.
static bool conv_fused_disabled() { return DA_XENV("DA_CONV_FUSED", 0) == 0; }

size_t conv_fused_lds_bytes(int max_graph_nodes, int kin) {
    const size_t n32 = (size_t)((max_graph_nodes + 31) / 32) * 32;
    return n32 * 64 + 32 * (2 * n32 + 16) + (size_t)2 * (kin / 16) * 1024;
}

bool conv_fused_applicable(int prec, int heads, int C, int kin, int max_graph_nodes, int ldo) {
    return !conv_fused_disabled() && prec == DA_PREC_BF16 && heads == 8 && C == 32 && (kin == 128 || kin == 256) && ldo == 256 &&
           max_graph_nodes > 0 && conv_fused_lds_bytes(max_graph_nodes, kin) <= 160 * 1024;
}

// returns 0 = launched, -1 = configuration not supported (caller takes the two-kernel path)
int launch_conv_fused(int prec, int heads, int C, int kin, int n_graphs, int max_graph_nodes, const int32_t *graph_ptr,
                      int nodiag, const void *x, int ldx, const void *W, const float *bias, int act, void *out, int ldo,
                      hipStream_t st) {
    if (!conv_fused_applicable(prec, heads, C, kin, max_graph_nodes, ldo) || n_graphs <= 0) return -1;
    const size_t lds = conv_fused_lds_bytes(max_graph_nodes, kin);
    ConvFusedParams p;
    p.x = (const bf16_t *)x; p.ldx = ldx; p.W = (const bf16_t *)W; p.bias = bias; p.out = (bf16_t *)out;
    p.graph_ptr = graph_ptr; p.n_graphs = n_graphs; p.act = act; p.nodiag = nodiag;
    p.sc = 1.4426950408889634f / sqrtf(32.0f);
    { const char *e = DA_XENV_LIVE("DA_FUSED_DEBUG"); p.debug = e ? atoi(e) : 0; }
    const int nblk = n_graphs * 8;
    // waves per workgroup: enough that no wave owns more than two 32-query slabs (the kernel keeps a slab's Q^T / skip^T
    // fragments in registers), 16 for the 900-piece graphs = 4 per SIMD
    const int nslab_max = (max_graph_nodes + 31) / 32;
    const int threads = nslab_max > 16 ? 1024 : (nslab_max > 8 ? 512 : 256);
    if (kin == 256) {
        static bool attr = false;
        if (!attr) { DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_conv_fused<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
        k_conv_fused<256><<<nblk, threads, lds, st>>>(p);
    } else {
        static bool attr = false;
        if (!attr) { DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_conv_fused<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
        k_conv_fused<128><<<nblk, threads, lds, st>>>(p);
    }
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
