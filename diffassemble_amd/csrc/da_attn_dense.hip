// Dense block-diagonal graph attention on the matrix cores: one PyG TransformerConv attention
// (Transformer_GNN.py:32,38) for batches whose graphs are COMPLETE (every piece attends to every
// piece of its puzzle, with or without self loops) -- the fully-connected case of the reference's
// rotation / translation datasets (puzzle_dataset.py:279-289,609-614).
//
//   out[i, h*C:(h+1)*C] = act( sum_j softmax_j(q_i.k_j / sqrt(C)) v_j + skip_i (+ residual_i) )
//
// flash-style: the n x n score matrix never exists; per (graph, head, 128-query tile) workgroup the
// K / V^T tiles stream through LDS and each of the 4 waves owns 32 queries.
//   S^T = K_tile . Q^T      (32 keys x 32 queries per MFMA chain; A = K rows from LDS, B = Q rows
//                            held in registers for the whole kernel)
//   each lane then holds ONE query column: running max / sum are lane-local plus a single
//   cross-half exchange, and P needs NO data movement to become the B operand of
//   O^T += V^T_tile . P^T   (A = V^T rows from LDS: V is produced already transposed by the
//                            projection GEMM, so these are plain 8/16-byte LDS reads)
// with a consistent permutation of the k-slots (key (r&3)+8(r>>2)+4*half <-> slot) on both
// operands.  bf16: v_mfma_f32_32x32x16_bf16; fp32 parity mode: v_mfma_f32_32x32x2_f32 (exact).
// LDS rows are padded to an odd number of 16-B (K, read b128) / 8-B (V^T bf16, read b64) slots,
// which makes every fragment read bank-conflict free.  Global -> LDS is register staged: tile t+1
// is in flight while tile t is multiplied.  Workgroup ids are remapped so that the 8 query tiles of
// one (graph, head) run back to back on ONE XCD and share its L2 copy of K / V^T.
#include "da_common.h"
#include "da_internal.h"

namespace da {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct AttnDenseParams {
    const void *Q, *K, *Vt, *S;     // [H][n_pad][C], [H][n_pad][C], [H][C][n_pad], [N][H*C]
    const void *res;                // [N][H*C] or null
    void *out;                      // [N][H*C]
    const int32_t *graph_ptr, *pad_ptr;
    int n_pad, H, n_graphs, nqt, act, nodiag;
    float sc;                       // log2(e) / sqrt(C)
};

template <typename T, int C> struct Cfg {
    static constexpr int ES = (int)sizeof(T);
    static constexpr int ROWB = C * ES;                       // bytes of one K / Q row
    static constexpr int NCH = ROWB / 32;                     // 32-byte K-dim chunks
    static constexpr int RS = ROWB + (((ROWB / 16) & 1) ? 0 : 16);
    static constexpr int BKEYS = ES == 2 ? 64 : 32;           // keys per LDS tile
    static constexpr int KB = BKEYS / 32;
    static constexpr int VROWB = BKEYS * ES;                  // 128 bytes
    static constexpr int RSV = ES == 2 ? VROWB + 8 : VROWB + 16;
    static constexpr int NCB = (C + 31) / 32;
    static constexpr int KBYTES = BKEYS * RS, VBYTES = C * RSV;
    static constexpr int NPK = BKEYS * ROWB / 16, NPV = C * (VROWB / 16);
    static constexpr int RPK = (NPK + 255) / 256, RPV = (NPV + 255) / 256;
    static_assert(ROWB % 32 == 0, "head width must be a multiple of 32 bytes");
};

// ---- S^T += Kfrag . Qfrag over one 32-byte chunk
__device__ __forceinline__ f32x16 mma_chunk(bf16_t, const u32x4 &a, const u32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma_chunk(float, const u32x4 &a, const u32x4 &b, f32x16 c) {
    const f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(x[0], y[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(x[1], y[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(x[2], y[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(x[3], y[3], c, 0, 0, 0);
    return c;
}

// ---- O^T[cb] += V^T rows . P^T for one 32-key block.  p[16] are this lane's probabilities.
template <int RSV>
__device__ __forceinline__ f32x16 mma_pv(bf16_t, const unsigned char *vrow, int kb, int half, const float (&p)[16], f32x16 o) {
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
        const int e0 = kb * 32 + 16 * mm + 4 * half;
        const u32x2 lo = *(const u32x2 *)(vrow + e0 * 2);
        const u32x2 hi = *(const u32x2 *)(vrow + (e0 + 8) * 2);
        const u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)p[8 * mm + e];
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf, o, 0, 0, 0);
    }
    return o;
}
template <int RSV>
__device__ __forceinline__ f32x16 mma_pv(float, const unsigned char *vrow, int kb, int half, const float (&p)[16], f32x16 o) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const f32x4 v4 = *(const f32x4 *)(vrow + (kb * 32 + 8 * jj + 4 * half) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_32x32x2f32(v4[e], p[4 * jj + e], o, 0, 0, 0);
    }
    return o;
}

__device__ __forceinline__ void ld4(const float *s, float v[4]) { const float4 f = *(const float4 *)s; v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
__device__ __forceinline__ void ld4(const bf16_t *s, float v[4]) {
    const uint2 u = *(const uint2 *)s;
    v[0] = bf2f((bf16_t)(u.x & 0xffff)); v[1] = bf2f((bf16_t)(u.x >> 16));
    v[2] = bf2f((bf16_t)(u.y & 0xffff)); v[3] = bf2f((bf16_t)(u.y >> 16));
}
__device__ __forceinline__ void st4(float *d, const float v[4]) { *(float4 *)d = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void st4(bf16_t *d, const float v[4]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *(uint2 *)d = __builtin_bit_cast(uint2, b);
}

template <typename T, int C>
__global__ __launch_bounds__(256, 2) void k_attn_dense(AttnDenseParams p) {
    using CF = Cfg<T, C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sK = smem, *sV = smem + CF::KBYTES;

    // XCD-aware remap: hardware places workgroup b on XCD b % 8; give XCD x head x of every graph and
    // walk the query tiles of one (graph, head) consecutively.
    const int bid = blockIdx.x;
    const int h = bid & 7, s_ = bid >> 3;
    const int qt = s_ % p.nqt, g = s_ / p.nqt;
    const int node0 = p.graph_ptr[g], n_g = p.graph_ptr[g + 1] - node0, pad0 = p.pad_ptr[g];
    if (qt * 128 >= n_g) return;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, half = lane >> 5;
    const int q0 = qt * 128 + wid * 32;
    const bool wave_on = q0 < n_g;
    const int HC = p.H * C;
    const size_t np = (size_t)p.n_pad;

    // Q fragments of this wave's 32 queries stay in registers
    u32x4 qf[CF::NCH];
    {
        const unsigned char *qrow = (const unsigned char *)p.Q + ((size_t)h * np + pad0 + min(q0, n_g - 1) / 32 * 32 + i) * CF::ROWB;
#pragma unroll
        for (int ch = 0; ch < CF::NCH; ++ch) qf[ch] = *(const u32x4 *)(qrow + ch * 32 + half * 16);
    }

    f32x16 O[CF::NCB];
#pragma unroll
    for (int cb = 0; cb < CF::NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[cb][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const unsigned char *Kg = (const unsigned char *)p.K + ((size_t)h * np + pad0) * CF::ROWB;
    const unsigned char *Vg = (const unsigned char *)p.Vt + ((size_t)h * C * np + pad0) * CF::ES;
    u32x4 rk[CF::RPK], rv[CF::RPV];
    auto gload = [&](int kt) {
#pragma unroll
        for (int x = 0; x < CF::RPK; ++x) {
            const int pi = tid + x * 256;
            if (pi < CF::NPK) rk[x] = *(const u32x4 *)(Kg + (size_t)kt * CF::BKEYS * CF::ROWB + (size_t)pi * 16);
        }
#pragma unroll
        for (int x = 0; x < CF::RPV; ++x) {
            const int pi = tid + x * 256;
            if (pi < CF::NPV) {
                const int c = pi >> 3, c16 = pi & 7;
                rv[x] = *(const u32x4 *)(Vg + ((size_t)c * np + (size_t)kt * CF::BKEYS) * CF::ES + c16 * 16);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int x = 0; x < CF::RPK; ++x) {
            const int pi = tid + x * 256;
            if (pi < CF::NPK) {
                const int row = pi / (CF::ROWB / 16), c16 = pi - row * (CF::ROWB / 16);
                *(u32x4 *)(sK + row * CF::RS + c16 * 16) = rk[x];
            }
        }
#pragma unroll
        for (int x = 0; x < CF::RPV; ++x) {
            const int pi = tid + x * 256;
            if (pi < CF::NPV) {
                const int c = pi >> 3, c16 = pi & 7;
                unsigned char *d = sV + c * CF::RSV + c16 * 16;
                if (CF::ES == 2) {                 // rows are only 8-byte aligned: two b64 writes
                    *(u32x2 *)d = (u32x2){rv[x][0], rv[x][1]};
                    *(u32x2 *)(d + 8) = (u32x2){rv[x][2], rv[x][3]};
                } else {
                    *(u32x4 *)d = rv[x];
                }
            }
        }
    };

    const int nkt = (n_g + CF::BKEYS - 1) / CF::BKEYS;
    const int qidx = q0 + i;                     // this lane's query (index inside the graph)
    gload(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (kt + 1 < nkt) gload(kt + 1);
        if (!wave_on) continue;
#pragma unroll
        for (int kb = 0; kb < CF::KB; ++kb) {
            const int key0 = kt * CF::BKEYS + kb * 32;
            if (key0 >= n_g) break;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const unsigned char *krow = sK + (kb * 32 + i) * CF::RS + half * 16;
#pragma unroll
            for (int ch = 0; ch < CF::NCH; ++ch) s = mma_chunk(T(), *(const u32x4 *)(krow + ch * 32), qf[ch], s);
            // mask padded keys (last tile) and the diagonal (graphs without self loops)
            const bool tail = key0 + 32 > n_g;
            const bool diag = p.nodiag && key0 < q0 + 32 && key0 + 32 > q0;
            if (tail || diag) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (kidx >= n_g || (p.nodiag && kidx == qidx)) s[r] = -INFINITY;
                }
            }
            float mloc = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            float mnew = fmaxf(m, mloc);
            if (mnew == -INFINITY) mnew = 0.f;               // nothing but masked keys so far
            const float corr = __builtin_amdgcn_exp2f((m - mnew) * p.sc);
            m = mnew;
            const float ms = mnew * p.sc;
            float pr[16], psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { pr[r] = __builtin_amdgcn_exp2f(fmaf(s[r], p.sc, -ms)); psum += pr[r]; }
            l = fmaf(l, corr, psum);
#pragma unroll
            for (int cb = 0; cb < CF::NCB; ++cb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) O[cb][r] *= corr;
                const int cr = min(cb * 32 + i, C - 1);
                O[cb] = mma_pv<CF::RSV>(T(), sV + cr * CF::RSV, kb, half, pr, O[cb]);
            }
        }
    }
    if (!wave_on || qidx >= n_g) return;

    // epilogue: normalise (PyG: sum + 1e-16), + skip (+ residual), activation, store
    const float lt = l + __shfl_xor(l, 32);
    const float inv = lt > 0.f ? 1.0f / (lt + 1e-16f) : 0.f;
    const size_t orow = ((size_t)node0 + qidx) * HC + (size_t)h * C;
    const T *sp = (const T *)p.S + orow;
    const T *rp = p.res ? (const T *)p.res + orow : nullptr;
    T *op = (T *)p.out + orow;
#pragma unroll
    for (int cb = 0; cb < CF::NCB; ++cb) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int c0 = cb * 32 + 8 * jj + 4 * half;
            if (c0 >= C) continue;
            float v[4], sk[4];
            ld4(sp + c0, sk);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(O[cb][4 * jj + e], inv, sk[e]);
            if (rp) {
                ld4(rp + c0, sk);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += sk[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
            st4(op + c0, v);
        }
    }
}

template <typename T, int C>
static int launch_tc(const AttnDenseParams &p, int nblocks, hipStream_t st) {
    using CF = Cfg<T, C>;
    const int lds = CF::KBYTES + CF::VBYTES;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_dense<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    k_attn_dense<T, C><<<nblocks, 256, lds, st>>>(p);
    DA_LAUNCH_CHECK();
    return 0;
}

// returns 0 = launched, -1 = configuration not supported (caller uses the CSR kernel)
int launch_attn_dense(int prec, const DenseLayout &L, int heads, int C, int n_graphs, int max_graph_nodes,
                      const int32_t *graph_ptr, const int32_t *pad_ptr, int nodiag, const void *res, int act,
                      void *out, hipStream_t st) {
    if (heads != 8 || (C != 32 && C != 144)) return -1;
    AttnDenseParams p;
    p.Q = L.Q; p.K = L.K; p.Vt = L.Vt; p.S = L.S; p.res = res; p.out = out;
    p.graph_ptr = graph_ptr; p.pad_ptr = pad_ptr; p.n_pad = L.n_pad; p.H = heads; p.n_graphs = n_graphs;
    p.nqt = (max_graph_nodes + 127) / 128; p.act = act; p.nodiag = nodiag;
    p.sc = 1.4426950408889634f / sqrtf((float)C);
    const int nblocks = p.nqt * heads * n_graphs;
    if (nblocks <= 0) return 0;
    if (prec == DA_PREC_BF16) return C == 32 ? launch_tc<bf16_t, 32>(p, nblocks, st) : launch_tc<bf16_t, 144>(p, nblocks, st);
    return C == 32 ? launch_tc<float, 32>(p, nblocks, st) : launch_tc<float, 144>(p, nblocks, st);
}

}  // namespace da
