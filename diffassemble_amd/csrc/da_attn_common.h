// Pieces shared by the dense block-diagonal attention kernels (da_attn_dense.hip: k_attn_dense / k_attn_dense2, the
// referenced and FAST softmax paths; da_attn_opt.hip: the optimistic kernels): parameter block, LDS tile geometry, MFMA
// chunk, transposing LDS read, row load / store helpers, counted vmcnt wait.
#pragma once
#include "da_common.h"
#include "da_internal.h"

namespace da {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct AttnDenseParams {
    const void *Q, *K, *Vt, *S;     // [H][n_pad][C] each (Vt holds V ROW-major since the tr_b16 rewrite), [N][H*C]
    const void *res;                // [N][H*C] or null
    void *out;                      // [N][H*C]
    const int32_t *graph_ptr, *pad_ptr;
    int n_pad, H, n_graphs, nqt, act, nodiag;       // nqt = query tiles per graph (set by the launcher: depends on the waves per workgroup)
    int max_nodes;                                  // largest graph of the batch
    float sc;                       // log2(e) / sqrt(C)
    unsigned long long *prof;       // DA_ATTN_PROBE: per-workgroup cycle breakdown of wave 0
    int fast;                       // Q pre-scaled (sc == 1): start every wave in the shift-free softmax mode (see k_attn_dense)
    int debug;                      // DA_ATTN_DEBUG bits (timing experiments): 1 no DMA, 2 no softmax, 4 no PV, 8 no QK
    int force_gen;                  // DA_ATTN_FORCE_GEN=1 (tests): the shift-free kernels start in their running-max fallback mode
    // hybrid (MASKED) mode: adjacency bits of the regular edges; the remainder edges are folded in by the epilogue
    const unsigned char *mask;      // rows of graph g at mask_ptr[g], row stride (pad_ptr[g+1] - pad_ptr[g]) / 8 bytes
    const long long *mask_ptr;
    const int32_t *irr_row_ptr;     // remainder edges (virtual nodes, duplicates, cross-graph pairs): CSR by destination
    const int32_t *irr_col_src;
    const int32_t *row_map;         // node -> padded slot (for the sources of remainder edges)
    const int32_t *slot_node;       // hybrid, banded layout: padded slot -> node (null: slot pad0 + i holds node node0 + i)
    const unsigned char *blk_class; // hybrid: class of every (32-query slab, 32-key block) of a graph: 0 empty, 1 partial, 2 full (or null)
    const long long *blk_class_ptr; // [G + 1] byte offsets of the graphs' class tables
    int blk_class_stride;           // bytes per slab row of a class table
    const int32_t *rm_meta;         // hybrid: [n_pad][4] per-slot { remainder begin, end, slot of the first remainder source, node } (or null)
    void *fold_out;                 // CV != C: [H][n_rows][CV] normalised per-head outputs in the activation dtype (no skip / activation here)
    int n_rows;
    // k_attn_res<.., 64> (the layer's projection in the kernel's prologue: no projection kernel ran; Q / K / Vt / S above are WRITTEN by the
    // kernel -- K | V only into its LDS image): the layer's input rows, the per-head weight fragments of pack_w_qs and the four bias blocks
    // (Q's carries the softmax scale like its weights)
    // k_attn_optt<32, false, true> (hidden layers of hybrid graphs): the rows the masked attention does NOT own -- the exophormer's virtual nodes
    // (exophormer_gnn.py:183-200) -- ride the same launch as v_rows * v_split extra workgroups behind the attention's (launch_optt): no second
    // kernel, no second stream.  v_row_ptr / v_col_src / v_mult: their remainder CSR by destination over all nodes (da_graph.agg_* or irr_*),
    // v_part / v_cnt: per-Batch scratch (partial softmax states of a row's workgroups; arrival counters, zero between launches).
    int v_rows, v_split, v_n_real;
    const int32_t *v_row_ptr, *v_col_src;
    const float *v_mult;
    float *v_part;                  // [v_rows][v_split][64 lanes][6]: m, l, acc[4]
    unsigned *v_cnt;                // [v_rows]
    const void *x;                  // [N][ldx] bf16 (null: Q / K / V / S come from memory)
    int ldx, kin;                   // kin = reduction length (128 / 256)
    const void *wqs;
    const float *bias_q, *bias_k, *bias_v, *bias_s;   // [H * 32] each
};

template <typename T, int C, int CV = C, int BK = 0> struct Cfg {
    static constexpr int ES = (int)sizeof(T);
    static constexpr int ROWB = C * ES;                       // bytes of one K / Q row
    static constexpr int ROWBV = CV * ES;                     // bytes of one V row (CV != C: value heads folded with the
                                                              // next linear layer, see launch_attn_dense)
    static constexpr int NCH = ROWB / 32;                     // 32-byte K-dim chunks
    static constexpr int RS = ROWB + (((ROWB / 16) & 1) ? 0 : 16);   // odd number of 16-B slots
    static constexpr int KSPR = RS / 16;                      // LDS slots per K row
    static constexpr int KVALID = ROWB / 16;                  // of which carry data
    static constexpr int BKEYS = BK ? BK : (ES == 2 ? 64 : 32);   // keys per LDS tile (BK = 0: the default of the element size)
    static constexpr int KB = BKEYS / 32;
    // V rows (row-major, like K).  bf16: the PV operand is fetched with ds_read_b64_tr_b16, whose 16-lane
    // groups read [4 keys][16 channels] blocks; two groups share an LDS cycle, and their 8 x 32-byte
    // pieces tile all 64 banks exactly when the row stride is 64 (mod 256) bytes.  fp32: scalar reads,
    // the two 32-lane halves sit 16 rows apart -> stride 32 (mod 64) bytes keeps them on disjoint banks.
    static constexpr int RSV = ES == 2 ? ((ROWBV - 64 + 255) / 256 * 256 + 64) : ((ROWBV - 32 + 63) / 64 * 64 + 32);
    static constexpr int KVALIDV = ROWBV / 16;
    static constexpr int VSPR = RSV / 16;
    static constexpr int NCB = (CV + 31) / 32;
    static constexpr int NIK = (BKEYS * KSPR + 63) / 64;      // DMA instructions (1 KB each) per tile
    static constexpr int NIV = (BKEYS * VSPR + 63) / 64;
    static constexpr int NI = NIK + NIV;
    static constexpr int MAXI = (NI + 3) / 4;                 // per wave
    static constexpr int KBYTES = NIK * 1024, VBYTES = NIV * 1024, STAGE = KBYTES + VBYTES;
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

// ---- PV operand fetch.  bf16: two transposing reads give this lane 8 consecutive keys of ONE channel
// (the A operand of v_mfma_f32_32x32x16_bf16) out of the row-major [key][channel] tile: lane i' of a
// 16-lane group supplies the address of key (i' >> 2), channels 4 (i' & 3)..+3, and receives channel i'
// of keys 0..3 (measured with tools/tr_probe.hip).
// Issued as inline asm: hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of the builtin form, i.e.
// it waits for the LDS-DMA of the NEXT tile (just issued) before every V fetch and serialises the
// pipeline.  The asm form is invisible to that pass, so the result must be fenced by hand: tr_fence()
// (s_waitcnt lgkmcnt(0) carrying the fragment registers as operands) before the first MFMA that uses it.
__device__ __forceinline__ u32x2 tr_read(unsigned lds_byte_addr, int imm) {      // imm folds to a constant after unrolling
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(imm));
    return r;
}
// fp32: O^T[cb] += V rows . P^T for one 32-key block with exact-fp32 MFMAs; vcol points at key 0 of the
// block, this lane's channel; p[16] = this lane's probabilities for keys 16*half + 0..15.
template <int RSV>
__device__ __forceinline__ f32x16 mma_pv_f32(const unsigned char *vcol, int half, const float (&p)[16], f32x16 o) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = *(const float *)(vcol + (16 * half + e) * RSV);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(v, p[e], o, 0, 0, 0);
    }
    return o;
}

__device__ __forceinline__ void ld4(const float *s, float v[4]) { const f32x4 f = *(const f32x4 *)s; v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3]; }
__device__ __forceinline__ void ld4(const bf16_t *s, float v[4]) {
    const u32x2 u = *(const u32x2 *)s;
    v[0] = bf2f((bf16_t)(u[0] & 0xffff)); v[1] = bf2f((bf16_t)(u[0] >> 16));
    v[2] = bf2f((bf16_t)(u[1] & 0xffff)); v[3] = bf2f((bf16_t)(u[1] >> 16));
}
__device__ __forceinline__ void st4(float *d, const float v[4]) { *(f32x4 *)d = (f32x4){v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void st4(bf16_t *d, const float v[4]) {
    const bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *(u32x2 *)d = __builtin_bit_cast(u32x2, b);
}

// one 16-byte chunk: 4 fp32 or 8 bf16
__device__ __forceinline__ void unpack_chunk(float, const u32x4 &u, float (&v)[4]) {
    const f32x4 f = __builtin_bit_cast(f32x4, u);
    v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
}
__device__ __forceinline__ void unpack_chunk(bf16_t, const u32x4 &u, float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = bf2f((bf16_t)(u[k] & 0xffff)); v[2 * k + 1] = bf2f((bf16_t)(u[k] >> 16)); }
}
__device__ __forceinline__ void ldc(const float *s, float (&v)[4]) { const f32x4 f = *(const f32x4 *)s; v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3]; }
__device__ __forceinline__ void ldc(const bf16_t *s, float (&v)[8]) {
    const u32x4 u = *(const u32x4 *)s;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = bf2f((bf16_t)(u[k] & 0xffff)); v[2 * k + 1] = bf2f((bf16_t)(u[k] >> 16)); }
}
__device__ __forceinline__ void stc(float *d, const float (&v)[4]) { *(f32x4 *)d = (f32x4){v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void stc(bf16_t *d, const float (&v)[8]) {
    bf16x8 b;
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = (__bf16)v[k];
    *(u32x4 *)d = __builtin_bit_cast(u32x4, b);
}

// NST = stages of the K / V ring.  An LDS-DMA takes ~2 us from issue to landing under load (measured), longer than a
// wave spends on one 64-key tile of the C = 32 layers: with two stages (one tile in flight) every tile waited for its own
// DMA and the kernel ran at the DMA latency (15 tiles x ~2 us per workgroup, 8 workgroups per CU in two rounds = the
// measured 55-61 us).  NST - 1 tiles are kept in flight instead, waited for with a COUNTED vmcnt.
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define DA_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        DA_VMCNT_CASE(1) DA_VMCNT_CASE(2) DA_VMCNT_CASE(3) DA_VMCNT_CASE(4) DA_VMCNT_CASE(5) DA_VMCNT_CASE(6) DA_VMCNT_CASE(7) DA_VMCNT_CASE(8)
        DA_VMCNT_CASE(9) DA_VMCNT_CASE(10) DA_VMCNT_CASE(11) DA_VMCNT_CASE(12) DA_VMCNT_CASE(13) DA_VMCNT_CASE(14) DA_VMCNT_CASE(15) DA_VMCNT_CASE(16)
        DA_VMCNT_CASE(17) DA_VMCNT_CASE(18) DA_VMCNT_CASE(19) DA_VMCNT_CASE(20) DA_VMCNT_CASE(21) DA_VMCNT_CASE(22) DA_VMCNT_CASE(23) DA_VMCNT_CASE(24)
#undef DA_VMCNT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // 0, or more than the cases cover: wait for everything
    }
}

// An UN-SHIFTED softmax state (weights exp2(s), reference 0) handed to the running-max recurrence: re-reference it by the
// exact power of two e = floor(log2(row sum)), so that the sum restarts in [1, 2) -- from there on PyG's `+ 1e-16` is as
// invisible as it is in the reference (where it is added to sum exp(a - max) >= 1), whatever the scores' offset.
__device__ __forceinline__ float pow2_floor_exp(float lq) { return (float)((int)((__builtin_bit_cast(unsigned, lq) >> 23) & 0xffu) - 127); }

// Hybrid graphs, remainder edges of one query tile (k_attn_dense<MASKED>, k_attn_optt<MASKED>): the rows staged in LDS
// (`so`, RSOF floats per query: CO un-normalised channels, then the row's softmax reference in nat and its sum) hold the
// state of the masked attention over the regular edges; each query's few remaining incoming edges -- from virtual nodes,
// duplicated pairs -- continue the same online softmax here, 8 lanes per query, before the rows are normalised.  A lane owns
// the 16-byte chunks sub, sub + 8, ... of the rows (vector loads); the CSR metadata of the four queries a lane group serves
// (rm_*) was fetched at kernel start, and the Q / K / V rows of all four first edges are requested before any is consumed:
// the pass used to be a chain of ~5 dependent global loads per query.  The caller synchronises the workgroup afterwards.
template <typename T, typename CFG, int CO, int RSOF>
__device__ __forceinline__ void remainder_edges(const AttnDenseParams &p, float *so, int h, size_t np, int pad0, int n_g, int qtile0,
                                                int wid, int lane, bool wave_on, const int (&rm_beg)[4], const int (&rm_end)[4],
                                                const int (&rm_slot)[4]) {
    constexpr int NCK = CFG::ROWB / 16, MAXT = (NCK + 7) / 8, EPK = 16 / CFG::ES;
    constexpr int NCKV = CFG::ROWBV / 16, MAXTV = (NCKV + 7) / 8;       // V rows may be narrower (folded heads)
    const int sub = lane & 7;
    const float scale = p.sc * 0.6931471805599453f;          // 1 / sqrt(C)
    auto load_row = [&](const void *base, size_t row, u32x4 (&dst)[MAXT]) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int ck = sub + 8 * t;
            dst[t] = ck < NCK ? *(const u32x4 *)((const unsigned char *)base + row * CFG::ROWB + ck * 16) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto load_vrow = [&](size_t row, u32x4 (&dst)[MAXTV]) {
#pragma unroll
        for (int t = 0; t < MAXTV; ++t) {
            const int ck = sub + 8 * t;
            dst[t] = ck < NCKV ? *(const u32x4 *)((const unsigned char *)p.Vt + row * CFG::ROWBV + ck * 16) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    // rows whose first-edge operands are requested together: all four, or two at a time for the 144-wide heads (three 16-byte pieces per Q / K
    // row and lane: four rows' Q, K and V are 112 registers -- the masked last-layer kernel spilled 20 of its 168 to scratch over them)
    constexpr int RG = MAXT >= 3 ? 2 : 4;
    if (wave_on) {
#pragma unroll
      for (int r0 = 0; r0 < 4; r0 += RG) {
        u32x4 qr[4][MAXT], kr[4][MAXT], vr[4][MAXTV];
#pragma unroll
        for (int r = r0; r < r0 + RG; ++r) {
            const int qg = min(qtile0 + wid * 32 + (lane >> 3) + 8 * r, n_g - 1);
            load_row(p.Q, (size_t)h * np + pad0 + qg, qr[r]);
            load_row(p.K, (size_t)h * np + rm_slot[r], kr[r]);
            load_vrow((size_t)h * np + rm_slot[r], vr[r]);
        }
#pragma unroll
        for (int r = r0; r < r0 + RG; ++r) {
            if (rm_end[r] <= rm_beg[r]) continue;
            const int ql = wid * 32 + (lane >> 3) + 8 * r;
            float *orow = so + ql * RSOF;
            float mm = orow[CO], ll = orow[CO + 1];
            if (!(ll > 0.f)) { mm = -INFINITY; ll = 0.f; }
            float qv[MAXT][EPK], acc[MAXTV][EPK];
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                unpack_chunk(T(), qr[r][t], qv[t]);
#pragma unroll
                for (int x = 0; x < EPK; ++x) qv[t][x] *= scale;
            }
#pragma unroll
            for (int t = 0; t < MAXTV; ++t)
#pragma unroll
                for (int x = 0; x < EPK; ++x) acc[t][x] = (sub + 8 * t < NCKV) ? orow[(sub + 8 * t) * EPK + x] : 0.f;
            for (int e = rm_beg[r]; e < rm_end[r]; ++e) {
                u32x4 k2[MAXT], v2[MAXTV];
                if (e > rm_beg[r]) {                              // beyond the prefetched first edge (rare)
                    const size_t sj = (size_t)h * np + (size_t)p.row_map[p.irr_col_src[e]];
                    load_row(p.K, sj, k2);
                    load_vrow(sj, v2);
                } else {
#pragma unroll
                    for (int t = 0; t < MAXT; ++t) k2[t] = kr[r][t];
#pragma unroll
                    for (int t = 0; t < MAXTV; ++t) v2[t] = vr[r][t];
                }
                float sc_ = 0.f;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    float kk[EPK];
                    unpack_chunk(T(), k2[t], kk);
#pragma unroll
                    for (int x = 0; x < EPK; ++x) sc_ = fmaf(qv[t][x], kk[x], sc_);
                }
                sc_ += __shfl_xor(sc_, 1);
                sc_ += __shfl_xor(sc_, 2);
                sc_ += __shfl_xor(sc_, 4);
                const float mn = fmaxf(mm, sc_);
                const float corr = expf(mm - mn), pe = expf(sc_ - mn);
                ll = ll * corr + pe;
#pragma unroll
                for (int t = 0; t < MAXTV; ++t) {
                    float vv[EPK];
                    unpack_chunk(T(), v2[t], vv);
#pragma unroll
                    for (int x = 0; x < EPK; ++x) acc[t][x] = fmaf(pe, vv[x], acc[t][x] * corr);
                }
                mm = mn;
            }
#pragma unroll
            for (int t = 0; t < MAXTV; ++t)
                if (sub + 8 * t < NCKV) {
#pragma unroll
                    for (int x = 0; x < EPK; ++x) orow[(sub + 8 * t) * EPK + x] = acc[t][x];
                }
            if (sub == 0) { orow[CO] = mm; orow[CO + 1] = ll; }
        }
      }
    }
}

// K tile geometry of these kernels.  C = 144: Cfg's padded rows (288 B + one 16-byte pad slot: an odd number of slots keeps the
// fragment reads conflict-free).  C = 32 (round 4): rows of exactly 64 B, their four 16-byte slots XOR-SWIZZLED instead of padded --
// slot s of key row r lives at position s ^ f(r), f(r) = bit 2 of r + 2 * bit 4 of r: the four rows a 16-lane group of a fragment
// read meets in one 64-byte quarter of the 256-byte bank window (r, r + 4, r + 16, r + 20 under the pi permutation) get four
// different positions.  The swizzle is applied to the LDS-DMA's per-lane SOURCE address, as in the GEMM kernels.  It saves the
// 1 KB of pad slots per 64-key stage (and one DMA instruction per tile): with the adjacency-word slots of the masked instance a
// stage is 9 KB again and four workgroups fit a CU -- at three, the 2048 workgroups of a 32-puzzle launch needed three rounds
// instead of two (measured: 92 us against 55 us for the un-masked kernel, with the masking itself costing nothing).
template <int C, int BK = 64> struct OptK {
    using CF = Cfg<bf16_t, C, 32, BK>;
    static constexpr bool SWZ = C == 32;
    static constexpr int RS = SWZ ? CF::ROWB : CF::RS, KSPR = RS / 16;
    static constexpr int NIK = (CF::BKEYS * KSPR + 63) / 64, NI = NIK + CF::NIV;
    static constexpr int KBYTES = NIK * 1024, STAGE = KBYTES + CF::VBYTES;
    static __device__ __forceinline__ int f(int row) { return SWZ ? (((row >> 2) & 1) | (((row >> 4) & 1) << 1)) : 0; }
};


// da_attn_opt.hip: the optimistic kernels (bf16, Q pre-scaled, 32-wide value heads); 0 = launched, -1 = shape not covered
int launch_attn_opt(const AttnDenseParams &p, int C, hipStream_t st);
bool attn_opt_took_virtual_rows();          // whether the last launch_attn_opt of this thread carried p.v_rows virtual rows in its grid
int attn_opt_counters(unsigned long long *out2, int reset);

}  // namespace da
