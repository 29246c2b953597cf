// Optimistic-softmax block-diagonal attention on the matrix cores (bf16, Q PRE-SCALED by log2(e) / sqrt(C), 32-wide value
// heads): one PyG TransformerConv attention (reference call sites backbones/Transformer_GNN.py:32,38;
// backbones/exophormer_gnn.py:203,205) per launch, for
//   k_attn_optt<32, false, false>   hidden layers on complete graphs (the class with the largest share of the sampling step);
//                                   Batches of 512 .. 1216-piece graphs go to k_attn_res below (K / V resident in LDS) since round 5
//   k_attn_optt<144, true, false>   the last layer on complete graphs, value heads folded with final_mlp.0
//   k_attn_optt<32, false, true>    hidden layers of hybrid graphs (adjacency-masked: Exphander + exophormer, config 3)
//   k_attn_optt<144, true, true>    their folded last layer
// Same LDS image, DMA ring, fragment layouts and epilogues as k_attn_dense (da_attn_dense.hip; geometry in da_attn_common.h).
// Written around what the round-3 measurements say bounds these kernels -- the SIMD's vector issue port (v_exp_f32 ~13-16
// cycles, packs ~6.4, every MFMA ~12 cycles of the same port; profiles/r03/NOTES.md) -- and, for C = 144, the
// register budget (the FAST mode of k_attn_dense keeps scores AND exponentials alive for its per-block fallback: 176 VGPRs,
// two waves per SIMD where round 2 had three).  Per score only the exponential and the pack are left on that port:
//   * OPTIMISTIC softmax: p = exp2(s) with no reference, no per-block test, in place.  fp32 / bf16 carry 8 exponent bits,
//     so this is exact whenever the row sums stay inside [2^-60, 2^100]; that is VERIFIED once, after the last key block,
//     on the final row sums (an overflow anywhere shows up there as inf / NaN, total underflow as 0).  A workgroup whose
//     check fails -- logits beyond +-41 nat -- re-runs its tile with the classic running-max recurrence (same loop, `gen`
//     switched on: row max, rescale, shift).  MASKED: a row sum of exactly 0 is legitimate when the row has no regular
//     edge (the OR of its adjacency words says so); with edges it means total underflow and fails the check;
//   * the row sums are eight v_dot2_f32_bf16 of the packed P against (1, 1) per block: they sum exactly the bf16 values the
//     PV product weighs with, at half the instruction count of fp32 adds (row sums on the matrix pipe -- two more MFMAs per
//     block against an all-ones operand -- measured slower: 113.6 vs 100.5 us per layer);
//   * an optimistic pass normalises by 1 / sum, WITHOUT PyG's + 1e-16: the sum is un-shifted there (anything inside the
//     window) while the reference adds its epsilon to sum exp(a - max) >= 1, where it is below fp32 resolution.  The GEN
//     pass keeps its sum >= 1 and PyG's formula;
//   * MASKED: the adjacency bits enter as the initial value of the S^T accumulator (0 / -inf from a 16-entry LDS table,
//     k_attn_dense's scheme); exp2(-inf) = 0 needs no special case.  The adjacency words ride the K / V ring: one more
//     LDS-DMA instruction per wave and tile (4 bytes per lane: the 32 bits of the lane's query row for one of the tile's two
//     key blocks) into a wave-private 256-byte slot of the stage, read back with two ds_read_u16 after the tile's wait.  As
//     ordinary global loads they cost the compiler's own `s_waitcnt vmcnt(0)` in front of every tile (k_attn_dense<MASKED>:
//     the DMA ring is drained every tile); as inline-asm loads the register allocator, which cannot know that their
//     destination is written asynchronously, reused it for DMA addresses (a memory fault, found the hard way).  The state
//     handed to the remainder-edge epilogue is re-referenced to (ln(sum), 1): rows normalised by their dense part, so the
//     epilogue's own online softmax starts from a sum of 1 whatever the logits' offset.
#include <stdlib.h>

#include <type_traits>

#include "da_attn_common.h"
#include "da_attn_res_asm.inc"

namespace da {

// DA_OPT_PROBE builds (tools/build_ab.sh probe da_attn_opt.hip "-fno-slp-vectorize -DDA_OPT_PROBE"): cycle breakdown of every
// wave-0 of the classic (non-pipelined) tile loop into DA_OPT_PROF_PTR: [0] whole kernel, [1] waiting for the tile (vmcnt +
// barrier), [2] issuing the DMA, [3] K-fragment reads until their data is there, [4] the rest of the block (QK, softmax, PV),
// [5] epilogue, [6] blocks, [7] 1.  (s_memtime; every stamp drains lgkmcnt, i.e. the V reads are not overlapped in this build.)
#ifdef DA_OPT_PROBE
#define DA_OPB(...) __VA_ARGS__
#else
#define DA_OPB(...)
#endif

// da_debug_counters: workgroups whose optimistic pass failed its verification ([0] complete graphs, [1] adjacency-masked)
__device__ unsigned long long g_opt_fallbacks[2];

// ---- the virtual rows of the exophormer arch as extra workgroups of the masked hidden-layer launch (round 6).
// A virtual node has no regular edge: its row is an online softmax over its remainder CSR (every real node of some graphs, the other virtual
// nodes, duplicates merged with multiplicities), 32-wide heads: lane = (head, 4-channel slice), a row's edges dealt out over the 4 * v_split
// waves of its v_split workgroups, four edges' K / V rows in flight per wave, their slots fetched one trip ahead (k_attn_csr_cont_heavy's walk).
// The partial states of a workgroup's waves meet in LDS, those of a row's workgroups in p.v_part: the workgroup that arrives last (p.v_cnt)
// merges them IN INDEX ORDER (deterministic), adds skip, applies the activation and resets the counter.  As a second kernel on the same stream
// these rows were 14 of a scripted Batch's 28 us per hidden layer (every kernel there sits at its launch floor); on a second stream they
// needed a fork and a join per layer.  Reference: the same TransformerConv rows (exophormer_gnn.py:203-205 over the edges of :183-200).
__device__ __forceinline__ void virt_rows_block(const AttnDenseParams &p, int vb, unsigned char *smem) {
    constexpr int EPL = 4, U = 4, C = 32;
    const int row = vb / p.v_split, part = vb - row * p.v_split;
    const int i = p.v_n_real + row;
    const int beg = p.v_row_ptr[i], end = p.v_row_ptr[i + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int head = lane >> 3, sub = (lane & 7) * EPL;
    const size_t hb = (size_t)head * (size_t)p.n_pad;
    const bf16_t *Q = (const bf16_t *)p.Q, *K = (const bf16_t *)p.K, *V = (const bf16_t *)p.Vt;
    const float scale = 0.6931471805599453f;          // Q is pre-scaled by log2(e) / sqrt(C): q . k is in log2 units
    float q[EPL], acc[EPL];
    {
        const u32x2 qq = *(const u32x2 *)(Q + (hb + (size_t)p.row_map[i]) * C + sub);
        q[0] = bf2f((bf16_t)(qq[0] & 0xffff)) * scale; q[1] = bf2f((bf16_t)(qq[0] >> 16)) * scale;
        q[2] = bf2f((bf16_t)(qq[1] & 0xffff)) * scale; q[3] = bf2f((bf16_t)(qq[1] >> 16)) * scale;
    }
#pragma unroll
    for (int x = 0; x < EPL; ++x) acc[x] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int NWT = 4 * p.v_split, gw = part * 4 + wv;
    size_t sjn[U];
    bool okn[U];
    float wgtn[U];
    auto fetch_idx = [&](int e0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * NWT;
            okn[u] = e < end;
            sjn[u] = hb + (size_t)p.row_map[p.v_col_src[okn[u] ? e : beg]];
            wgtn[u] = p.v_mult ? p.v_mult[okn[u] ? e : beg] : 1.0f;
        }
    };
    if (beg + gw < end) fetch_idx(beg + gw);
    for (int e0 = beg + gw; e0 < end; e0 += NWT * U) {
        size_t sj[U];
        bool ok[U];
        float wgt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { sj[u] = sjn[u]; ok[u] = okn[u]; wgt[u] = wgtn[u]; }
        u32x2 kk[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { kk[u] = *(const u32x2 *)(K + sj[u] * C + sub); vv[u] = *(const u32x2 *)(V + sj[u] * C + sub); }
        if (e0 + NWT * U < end) fetch_idx(e0 + NWT * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;                            // wave-uniform
            float s_ = q[0] * bf2f((bf16_t)(kk[u][0] & 0xffff));
            s_ = fmaf(q[1], bf2f((bf16_t)(kk[u][0] >> 16)), s_);
            s_ = fmaf(q[2], bf2f((bf16_t)(kk[u][1] & 0xffff)), s_);
            s_ = fmaf(q[3], bf2f((bf16_t)(kk[u][1] >> 16)), s_);
            s_ += __shfl_xor(s_, 1);
            s_ += __shfl_xor(s_, 2);
            s_ += __shfl_xor(s_, 4);
            const float mn = fmaxf(m, s_);
            const float corr = expf(m - mn);
            const float pr = expf(s_ - mn) * wgt[u];
            l = l * corr + pr;
            acc[0] = fmaf(pr, bf2f((bf16_t)(vv[u][0] & 0xffff)), acc[0] * corr);
            acc[1] = fmaf(pr, bf2f((bf16_t)(vv[u][0] >> 16)), acc[1] * corr);
            acc[2] = fmaf(pr, bf2f((bf16_t)(vv[u][1] & 0xffff)), acc[2] * corr);
            acc[3] = fmaf(pr, bf2f((bf16_t)(vv[u][1] >> 16)), acc[3] * corr);
            m = mn;
        }
    }
    // ---- the workgroup's four waves: (m, l, acc) per lane through LDS, merged by wave 0 in wave order
    float *sh = (float *)smem;                               // [4 waves][64 lanes][6]
    {
        float *o_ = sh + ((size_t)wv * 64 + lane) * 6;
        o_[0] = m; o_[1] = l; o_[2] = acc[0]; o_[3] = acc[1]; o_[4] = acc[2]; o_[5] = acc[3];
    }
    __syncthreads();
    if (wv != 0) return;
    auto merge = [&](auto get, int cnt, float &M, float &L, float (&a)[EPL]) {          // get(w, k): field k of partial state w of this lane
        M = -INFINITY;
        for (int w = 0; w < cnt; ++w) M = fmaxf(M, get(w, 0));
        L = 0.f;
#pragma unroll
        for (int x = 0; x < EPL; ++x) a[x] = 0.f;
        for (int w = 0; w < cnt; ++w) {
            const float lw = get(w, 1);
            if (!(lw > 0.f)) continue;
            const float f = expf(get(w, 0) - M);
            L = fmaf(lw, f, L);
#pragma unroll
            for (int x = 0; x < EPL; ++x) a[x] = fmaf(get(w, 2 + x), f, a[x]);
        }
    };
    float M, L;
    merge([&](int w, int k) { return sh[((size_t)w * 64 + lane) * 6 + k]; }, 4, M, L, acc);
    if (p.v_split > 1) {
        // (no fence: an agent-scope fence is an L2 write-back + invalidate on this chip, and two thousand workgroups issuing one each cost the
        //  layer 130 us -- measured.  The partial states go to the L2 as relaxed agent-scope atomics, the wait below lets them arrive there, the
        //  arrival counter lives there, and the last workgroup reads them from there.)
        float *mine = p.v_part + (((size_t)row * p.v_split + part) * 64 + lane) * 6;
        const float st6[6] = {M, L, acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
        for (int k = 0; k < 6; ++k) __hip_atomic_store(mine + k, st6[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(p.v_cnt + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        if ((int)old != p.v_split - 1) return;               // somebody else arrives last
        const float *all = p.v_part + ((size_t)row * p.v_split * 64 + lane) * 6;
        merge([&](int w, int k) { return __hip_atomic_load(all + (size_t)w * 64 * 6 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, p.v_split, M, L, acc);
        if (lane == 0) __hip_atomic_store(p.v_cnt + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // ready for the next launch
    }
    const float inv = L > 0.f ? 1.0f / (L + 1e-16f) : 0.f;
    const size_t o = (size_t)i * (size_t)(p.H * C) + (size_t)head * C + sub;
    const u32x2 sk = *(const u32x2 *)((const bf16_t *)p.S + o);
    float v4[4] = {acc[0] * inv + bf2f((bf16_t)(sk[0] & 0xffff)), acc[1] * inv + bf2f((bf16_t)(sk[0] >> 16)),
                   acc[2] * inv + bf2f((bf16_t)(sk[1] & 0xffff)), acc[3] * inv + bf2f((bf16_t)(sk[1] >> 16))};
#pragma unroll
    for (int x = 0; x < 4; ++x) stf((bf16_t *)p.out + o + x, apply_act(v4[x], p.act));
}

// (OptK -- the K tile geometry of these kernels -- lives in da_attn_common.h)
// BK = keys per ring stage (64: two 32-key blocks per stage and barrier; 32: one -- half the bytes per stage, so that a
// deeper ring fits the same LDS); VAR = instruction-mix experiments (bit 0: row sums as f32 adds of the un-rounded
// exponentials instead of v_dot2c on the packed P; bit 1: P packed by TRUNCATION (one v_perm_b32 per pair) instead of
// v_cvt_pk_bf16_f32 -- the truncation bias cancels in the normalisation because the row sums weigh the same truncated values).
// NWV = waves (32-query slabs) per workgroup: 4 (128-query tiles) or 5 (160-query tiles: a 144-piece puzzle is ONE tile instead of a
// full one and a 16-query rest, a 900-piece puzzle six tiles / 30 slabs instead of eight / 32 for its 29; un-masked instances).
template <int C, bool FOLD, bool MASKED, int NST, int MINB, int BK = 64, int VAR = 0, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, MINB) void k_attn_optt(AttnDenseParams p) {
    using T = bf16_t;
    constexpr int CV = 32, NW = NWV, QT = 32 * NWV, NT = 64 * NWV;
    static_assert(NWV == 4 || (!MASKED && NWV <= 8), "the adjacency-word slots and class rows are laid out for four waves");
    using CF = Cfg<T, C, CV, BK>;
    using KG = OptK<C, BK>;
    static_assert(!MASKED || BK == 64, "the adjacency words and class rows are laid out for 64-key tiles");
    static_assert((VAR & 3) != 3, "truncated P needs the row sums of the truncated values");
    constexpr bool PIPE = (VAR & 64) != 0;         // software-pipelined optimistic pass (see there)
    constexpr bool SDMA = PIPE || !(VAR & 256);    // K / V pieces issued in the scalar-base DMA form (default since round 5; VAR bit 256 = the per-lane-address form)
    static_assert(!PIPE || (!MASKED && NST >= 3 && !(VAR & 63)), "the pipelined pass serves the un-masked instances, on a ring of >= 3 stages");
    static_assert(BK <= 64, "key tiles beyond 64 rows read past the graphs' 64-row slot padding");
    // VAR bits 2..5 are timing ablations (wrong results): 4 no exponentials, 8 one QK product per block (all K reads kept),
    // 16 K fragments read once per workgroup, 32 no PV products
    static_assert(CF::NCB == 1, "one 32-channel value block");
    constexpr int MAXI = (KG::NI + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int vfirst = 0;          // workgroups in FRONT of the attention's own: the virtual rows (virt_rows_block above) -- their latency chains run under the attention
    if constexpr (C == 32 && !FOLD && MASKED && NWV == 4) {
        if (p.v_rows > 0) {
            vfirst = (p.v_rows * p.v_split + 7) & ~7;          // (a multiple of eight: the attention's head -> XCD map keeps its phase)
            if ((int)blockIdx.x < vfirst) {
                if ((int)blockIdx.x < p.v_rows * p.v_split) virt_rows_block(p, (int)blockIdx.x, smem);
                return;
            }
        }
    }
    DA_OPB(const unsigned long long pb_start = __builtin_readcyclecounter(), pb_wall0 = wall_clock64();)
    constexpr int MSTAGE = KG::STAGE + (MASKED ? 1024 : 0);     // MASKED: + the four waves' adjacency-word slots (256 B each)
    int *flags = (int *)(smem + NST * MSTAGE);                  // one word per wave: "my optimistic pass failed"; [4 .. 11]: the waves' tile masks
    float *mlut = (float *)(smem + NST * MSTAGE + 64);          // MASKED: nibble -> four accumulator initial values
    unsigned char *lcls = smem + NST * MSTAGE + 64 + 256;       // MASKED: the four waves' block-class rows (128 B each)
    if (MASKED && threadIdx.x < 64) {
        const int e = threadIdx.x >> 2, b = threadIdx.x & 3;
        mlut[threadIdx.x] = ((e >> b) & 1) ? 0.f : -INFINITY;
    }
    // (visible to every wave after the first barrier of the tile loop)

    // XCD-aware remap (as k_attn_dense): XCD x takes head x of every graph, the query tiles of one (graph, head) run back to back on it
    const int bid = (int)blockIdx.x - vfirst;
    const int h = bid & 7, s_ = bid >> 3;
    const int qt = s_ % p.nqt, g = s_ / p.nqt;
    const int node0 = p.graph_ptr[g], n_g = p.graph_ptr[g + 1] - node0, pad0 = p.pad_ptr[g];
    if (qt * QT >= n_g) return;
    DA_OPB(const unsigned long long pb_p1 = __builtin_readcyclecounter();)

    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = qt * QT + wid * 32;
    const bool wave_on = q0 < n_g;
    const int HC = p.H * C;
    const size_t np = (size_t)p.n_pad;

    u32x4 qf[CF::NCH];
    {
        const unsigned char *qrow = (const unsigned char *)p.Q + ((size_t)h * np + pad0 + min(q0, n_g - 1) / 32 * 32 + i) * CF::ROWB;
#pragma unroll
        for (int ch = 0; ch < CF::NCH; ++ch) qf[ch] = *(const u32x4 *)(qrow + ch * 32 + half * 16);
    }
    DA_OPB(asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[CF::NCH - 1])); const unsigned long long pb_p2 = __builtin_readcyclecounter();)
    const unsigned char *Kg = (const unsigned char *)p.K + ((size_t)h * np + pad0) * CF::ROWB;
    const unsigned char *Vg = (const unsigned char *)p.Vt + ((size_t)h * np + pad0) * CF::ROWBV;
    unsigned soff[MAXI];
#pragma unroll
    for (int x = 0; x < MAXI; ++x) {
        const int q = wid + NW * x;
        unsigned o = 0;
        if (q < KG::NIK) {
            const int s = q * 64 + lane, row = s / KG::KSPR, col = s - row * KG::KSPR;
            if (KG::SWZ) o = (unsigned)(row * CF::ROWB + (col ^ KG::f(row)) * 16);          // LDS position col holds slot col ^ f(row)
            else if (row < CF::BKEYS && col < CF::KVALID) o = (unsigned)(row * CF::ROWB + col * 16);
        } else {
            const int s = (q - KG::NIK) * 64 + lane, row = s / CF::VSPR, col = s - row * CF::VSPR;
            if (row < CF::BKEYS && col < CF::KVALIDV) o = (unsigned)(row * CF::ROWBV + col * 16);
        }
        soff[x] = o;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned char *mrow4 = nullptr;       // MASKED: this lane's source of the adjacency-word DMA (set below)
    auto issue = [&](int kt, int stage) {
        unsigned char *sb = smem + stage * MSTAGE;
        const unsigned char *kb_ = Kg + (size_t)kt * CF::BKEYS * CF::ROWB;
        const unsigned char *vb_ = Vg + (size_t)kt * CF::BKEYS * CF::ROWBV;
        if constexpr (SDMA) {
            // "scalar base + 32-bit lane offset" form (the GEMM producers' idiom): the tile's base is wave-uniform, the lane
            // offsets never change -- no 64-bit per-lane address to rebuild per piece (or to spill: a reloaded address register makes
            // the compiler put `s_waitcnt vmcnt(0)` in front of the DMA, which drains the ring).  Round-5 probe, last layer: 9.7 k
            // -> 4.8 k cycles of DMA issue per workgroup, 184 -> 176 us per 64-puzzle launch, the sampling step -1.7 ... -2.4 %.
            const unsigned sbl = lds0 + (unsigned)(stage * MSTAGE);
#pragma unroll
            for (int x = 0; x < MAXI; ++x) {
                const int q = wid + NW * x;
                if (NW * x + NW - 1 < KG::NI || q < KG::NI)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(sbl + q * 1024), "v"(soff[x]),
                                 "s"(q < KG::NIK ? kb_ : vb_)
                                 : "memory");
            }
        } else {
#pragma unroll
            for (int x = 0; x < MAXI; ++x) {
                const int q = wid + NW * x;
                if (NW * x + NW - 1 < KG::NI || q < KG::NI) {
                    const unsigned char *src = (q < KG::NIK ? kb_ : vb_) + soff[x];
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(sb + q * 1024), 16, 0, 0);
                }
            }
        }
        if (MASKED)         // lane (i, half) fetches the 32 adjacency bits of query row i for key block `half` of the tile
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(mrow4 + 8 * (size_t)kt),
                                             (__attribute__((address_space(3))) void *)(sb + KG::STAGE + wid * 256), 4, 0, 0);
    };
    const int myn = (KG::NI - wid + NW - 1) / NW + (MASKED ? 1 : 0);      // DMA instructions this wave issues per tile
    const int nkt = (n_g + CF::BKEYS - 1) / CF::BKEYS;
    const int qidx = q0 + i;
    const int pi_i = (i & 3) + 4 * ((i >> 3) & 3) + 16 * ((i >> 2) & 1);
    // byte offsets of this lane's K fragments (chunk ch = logical slot 2 ch + half of key row pi_i) inside a 32-key block
    int kfo[CF::NCH];
#pragma unroll
    for (int ch = 0; ch < CF::NCH; ++ch) kfo[ch] = pi_i * KG::RS + (KG::SWZ ? (((2 * ch + half) ^ KG::f(pi_i)) * 16) : (ch * 32 + half * 16));
    const int li = lane & 15;
    const int vbase = KG::KBYTES + (16 * half + (li >> 2)) * CF::RSV + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;

    // MASKED: this lane's row of the adjacency bit matrix (bit j = edge j -> this query); the 8 bytes of key tile kt sit at
    // mrow + 8 kt (rows are 8-byte aligned: the padded slot count is a multiple of 64); waves without queries fetch the
    // graph's last row (every wave issues the same number of DMA instructions per tile).  Remainder-edge metadata of the four
    // queries this 8-lane group finishes in the epilogue.
    // Slots vs nodes.  Rows of Q / K / V and of the adjacency matrix live in SLOT space (the padded per-graph row ranges the
    // projection scatters into through row_map).  Normally slot pad0 + i holds node node0 + i; an expander plan in the banded
    // layout (graph_plan.expander_plan) orders a graph's slots by the nodes' POSITIONS in the generator's permutation, which
    // turns the adjacency into a circulant band (whole 32 x 32 blocks empty or full), and hands slot_node[] to find the node
    // -- skip / output rows, remainder edges -- of a slot.
    auto node_of = [&](int ql) { return (MASKED && p.slot_node) ? p.slot_node[pad0 + ql] : node0 + ql; };       // ql < n_g
    const unsigned char *mrow = nullptr;
    int rm_beg[4] = {0, 0, 0, 0}, rm_end[4] = {0, 0, 0, 0}, rm_slot[4] = {0, 0, 0, 0};
    int my_node = node0 + min(qidx, n_g - 1);       // MASKED: the node this lane's query slot holds (for the epilogue's row addresses)
    if (MASKED) {
        mrow = p.mask + p.mask_ptr[g] + (size_t)min(qidx, n_g - 1) * (size_t)((p.pad_ptr[g + 1] - pad0) >> 3);
        if (wave_on && p.rm_meta) {
            // one 16-byte record per query (da_graph.rm_meta), none of it needed before the epilogue: the loads stay in flight
            // under the key loop.  (Walking slot_node -> irr_row_ptr -> irr_col_src -> row_map here, and waiting for it, put four
            // dependent L2 round trips in front of every workgroup: ~14 of the ~19 us a workgroup lived at d = 90.)
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            const i32x4 *meta = (const i32x4 *)p.rm_meta + pad0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qg = qt * QT + wid * 32 + (lane >> 3) + 8 * r;
                const i32x4 mt = meta[min(qg, n_g - 1)];
                rm_beg[r] = qg < n_g ? mt[0] : 0;
                rm_end[r] = qg < n_g ? mt[1] : 0;
                rm_slot[r] = mt[2];
            }
            my_node = ((const int *)(meta + min(qidx, n_g - 1)))[3];
        } else if (wave_on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qg = qt * QT + wid * 32 + (lane >> 3) + 8 * r;
                if (qg < n_g) {
                    const int nd = node_of(qg);
                    rm_beg[r] = p.irr_row_ptr[nd];
                    rm_end[r] = p.irr_row_ptr[nd + 1];
                }
                rm_slot[r] = pad0;                                       // any valid slot when there is no edge
                if (rm_end[r] > rm_beg[r]) rm_slot[r] = p.row_map[p.irr_col_src[rm_beg[r]]];
            }
            my_node = node_of(min(qidx, n_g - 1));
        }
    }
    if (MASKED) mrow4 = mrow + 4 * half;

    // MASKED with block classes (da_graph.blk_class: per 32-query slab and 32-key block 0 = no edge, 1 = some, 2 = all): a wave
    // skips the empty blocks of its slab outright, runs the full ones without the adjacency words, and the workgroup walks only
    // the key tiles in which at least one of its slabs has an edge (d = 539 of 900: 63 % of the tiles' blocks survive, 14 % of
    // them partial; d = 90: 13 %).  Without classes every block counts as partial.
    const unsigned char *crow = nullptr;        // this wave's class row (wave-uniform address)
    unsigned long long tmask = 0;                // key tiles the workgroup needs (bit kt)
    int ntl = nkt;                               // ... and their number
    if (MASKED) {
        tmask = nkt >= 64 ? ~0ull : ((1ull << nkt) - 1ull);
        if (p.blk_class) {
            // behind the graph's class rows the plan keeps one 64-bit word per query tile: the key tiles in which some slab
            // of the tile has an edge (ONE scalar load; computing it here cost two barrier rounds in front of the first DMA)
            const int nsl = (p.pad_ptr[g + 1] - pad0) >> 5;
            const unsigned char *tab = p.blk_class + p.blk_class_ptr[g];
            tmask &= ((const unsigned long long *)(tab + (size_t)nsl * (size_t)p.blk_class_stride))[qt];
            ntl = __builtin_popcountll(tmask);
            crow = lcls + wid * 128;             // filled below, once the first DMA stages are on their way
        }
    }
    // the j-th tile of the walk: j itself, or the j-th set bit of tmask (cursor masks, scalar arithmetic)
    auto next_tile = [&](unsigned long long &rem, int seq) {
        if (!MASKED) return seq;
        const int k = __builtin_ctzll(rem);
        rem &= rem - 1ull;
        return k;
    };

    DA_OPB(unsigned long long pb_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pb_t = __builtin_readcyclecounter(); const unsigned long long pb_p3 = pb_t;)
    f32x16 O;
    float ls = 0.f;               // this lane's share of the row sum (16 of the block's 32 keys)
    float ls2 = 0.f;              // VAR & 1: second chain of the row sum
    float m = 0.f;                // GEN mode only: running row max (log2 units)
    unsigned anym = 0;            // MASKED: OR of this lane's adjacency words
    bool gen = p.force_gen != 0;  // false: optimistic pass
    // epilogue operands requested EARLY (un-masked, un-folded instances whose epilogue is one pass over the tile: the hidden layers):
    // the skip (+ residual) rows of this tile are loaded right behind the key loop, so that their latency passes under the
    // verification barrier and the O staging instead of being waited for in the epilogue (round-5 probe: 4.3 k cycles of epilogue
    // per workgroup, most of it these loads' round trip)
    constexpr int EPC_ = 8, CPR_ = C / EPC_, NBP = 3;
    // (measured, round 5: hidden layers 55.7 / 56.1 / 56.6 us with it against 56.8 / 55.4 / 56.2 without, the sampling step 0.7044 /
    //  0.7069 / 0.7045 ms against 0.7019 / 0.6991 / 0.7028 -- no gain, nine more registers: opt-in, VAR bit 128)
    constexpr bool EPI_PRE = (VAR & 128) && !FOLD && !MASKED && !(VAR & 64) && (QT * CPR_ <= NT * NBP);
    u32x4 skv_p[EPI_PRE ? NBP : 1], rsv_p[EPI_PRE ? NBP : 1];
    for (int attempt = 0; attempt < 2; ++attempt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] = 0.f;
        ls = 0.f;
        ls2 = 0.f;
        m = -1e30f;
        anym = 0;
        unsigned long long rem_cur = tmask, rem_pf = tmask;
        if constexpr (PIPE) {
            if (!gen) {
                // ---- software-pipelined optimistic pass (un-masked instances).  Measured (round 5, timing ablations in
                // DESIGN.md): on one SIMD the matrix pipe's time and the vector port's time ADD UP across waves -- a wave that
                // issues its QK chain back to back parks the next MFMA in the shared issue stage until the pipe frees, and
                // nobody's exponentials issue meanwhile (removing one MFMA from either kernel saves ~44 cycles per block and
                // wave, whatever the other waves do).  So the cover has to come from the wave's OWN stream: the QK chain of
                // block b + 1 is issued one MFMA at a time with the exponentials / packs of block b behind each, the two PV
                // products of block b with the row-sum dots behind them.  Two score tiles alive (SA / SB, roles swap every
                // block; static names, the loop is unrolled by two).  Ring protocol of this pass: `acquire(kt)` runs at the top of
                // the step that still needs V of tile kt - 1, so the stage it may refill is tile kt - 2's: NST - 2 tiles in flight.
                const int nblk = (n_g + 31) >> 5;
#pragma unroll
                for (int st = 0; st < NST - 2; ++st)
                    if (st < nkt) issue(st, st);
                auto acquire = [&](int kt) {
                    const int younger = min(nkt - 1 - kt, NST - 3);
                    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else wait_vmcnt(younger * myn);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (kt + NST - 2 < nkt) issue(kt + NST - 2, (kt + NST - 2) % NST);
                };
                auto load_k = [&](int blk, u32x4(&kf)[CF::NCH]) {
                    const unsigned char *base = smem + ((blk / CF::KB) % NST) * MSTAGE + (blk % CF::KB) * 32 * KG::RS;
#pragma unroll
                    for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(base + kfo[ch]);
                };
                constexpr int UPS = (8 + CF::NCH - 1) / CF::NCH;          // exp / pack units (two scores each) behind one QK product
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};
                // one block: finish block b out of Sc (softmax, PV) while the scores of block b + 1 are produced into Sn
                auto step = [&](auto last_tag, f32x16 &Sc, f32x16 &Sn, int b) {
                    constexpr bool LAST = decltype(last_tag)::value;
                    DA_OPB(pb_t = __builtin_readcyclecounter();)
                    if (!LAST && (b + 1) % CF::KB == 0) acquire((b + 1) / CF::KB);
                    DA_OPB({ const unsigned long long t_ = __builtin_readcyclecounter(); pb_[1] += t_ - pb_t; pb_t = t_; })
                    if (!wave_on) return;
                    u32x4 kf[CF::NCH];
                    if (!LAST) load_k(b + 1, kf);
                    u32x2 vlo[2], vhi[2];
                    const unsigned vb = lds0 + (unsigned)(((b / CF::KB) % NST) * MSTAGE + vbase + (b % CF::KB) * 32 * CF::RSV);
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        vlo[mm] = tr_read(vb, (8 * mm) * CF::RSV);
                        vhi[mm] = tr_read(vb, (8 * mm + 4) * CF::RSV);
                    }
                    {
                        const int key0 = b * 32, kbase = key0 + 16 * half;
                        const bool tail = key0 + 32 > n_g;
                        const bool diag = p.nodiag && key0 < q0 + 32 && key0 + 32 > q0;
                        if (tail || diag) {             // rare (the last block, the diagonal block): a real branch -- the asm keeps the
                            asm volatile("" ::: "memory");      // compiler from if-converting it into 16 compares + selects per block
                            const int lim = n_g - kbase, dg = p.nodiag ? qidx - kbase : -1;      // (two values per block, compared with the
#pragma unroll                                                                                       //  constants r: nothing hoisted into registers)
                            for (int r = 0; r < 16; ++r)
                                if (r >= lim || r == dg) Sc[r] = -INFINITY;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 pf0, pf1;
                    int unit = 0;
                    auto units = [&](int n) {
#pragma unroll
                        for (int u = 0; u < n; ++u, ++unit)
                            if (unit < 8) {
                                const float x0 = __builtin_amdgcn_exp2f(Sc[2 * unit]), x1 = __builtin_amdgcn_exp2f(Sc[2 * unit + 1]);
                                if (unit < 4) { pf0[2 * unit] = (__bf16)x0; pf0[2 * unit + 1] = (__bf16)x1; }
                                else { pf1[2 * unit - 8] = (__bf16)x0; pf1[2 * unit - 7] = (__bf16)x1; }
                            }
                    };
                    auto dots = [&](const bf16x8 &pf) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bf16x2 pa = {pf[2 * e], pf[2 * e + 1]};
                            ls = __builtin_amdgcn_fdot2_f32_bf16(pa, one2, ls, false);
                        }
                    };
                    if (!LAST) {
#pragma unroll
                        for (int ch = 0; ch < CF::NCH; ++ch) {
                            if (ch == 0) {
                                f32x16 z;
#pragma unroll
                                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                                Sn = mma_chunk(T(), kf[0], qf[0], z);
                            } else {
                                Sn = mma_chunk(T(), kf[ch], qf[ch], Sn);
                            }
                            units(UPS);
                            if (CF::NCH > 8 && ch == CF::NCH - 1) dots(pf0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        units(8);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]));
                    const u32x4 v0 = {vlo[0][0], vlo[0][1], vhi[0][0], vhi[0][1]};
                    const u32x4 v1 = {vlo[1][0], vlo[1][1], vhi[1][0], vhi[1][1]};
                    O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf0, O, 0, 0, 0);
                    if (!(CF::NCH > 8) || LAST) dots(pf0);
                    __builtin_amdgcn_sched_barrier(0);
                    O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, O, 0, 0, 0);
                    dots(pf1);
                    __builtin_amdgcn_sched_barrier(0);
                    DA_OPB({ const unsigned long long t_ = __builtin_readcyclecounter(); pb_[4] += t_ - pb_t; pb_t = t_; pb_[6] += 1; })
                };
                f32x16 SA, SB;
                acquire(0);
                if (wave_on) {
                    u32x4 kf[CF::NCH];
                    load_k(0, kf);
#pragma unroll
                    for (int r = 0; r < 16; ++r) SA[r] = 0.f;
#pragma unroll
                    for (int ch = 0; ch < CF::NCH; ++ch) SA = mma_chunk(T(), kf[ch], qf[ch], SA);
                }
                const std::false_type mid;
                const std::true_type last;
                int b = 0;
#pragma unroll 1
                for (; b + 2 < nblk; b += 2) {
                    step(mid, SA, SB, b);
                    step(mid, SB, SA, b + 1);
                }
                if (b + 2 == nblk) {
                    step(mid, SA, SB, b);
                    step(last, SB, SA, b + 1);
                } else {
                    step(last, SA, SB, b);
                }
            }
        }
        if (!PIPE || gen) {
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < ntl) issue(next_tile(rem_pf, st), st);
        if (MASKED && crow && attempt == 0) {
            // the wave's class row -> its private LDS slot (read back per tile as a broadcast ds_read; a per-tile GLOBAL load of
            // it put a compiler-managed vmcnt(0) in front of every tile and drained the DMA ring)
            // (waves whose slab lies beyond the graph's last slab stage the last slab's row: the table has one row per 32 slots)
            const int nsl_ = (p.pad_ptr[g + 1] - pad0) >> 5;
            const unsigned char *grow = p.blk_class + p.blk_class_ptr[g] + (size_t)min(qt * 4 + wid, nsl_ - 1) * (size_t)p.blk_class_stride;
            if (lane < 32) ((unsigned *)crow)[lane] = (4 * lane < p.blk_class_stride) ? ((const unsigned *)grow)[lane] : 0u;
        }
        for (int j = 0; j < ntl; ++j) {
            const int kt = next_tile(rem_cur, j);
            DA_OPB(pb_t = __builtin_readcyclecounter();)
            {
                // tiles that may stay in flight behind this one (each is `myn` operations of this wave; myn is LO or LO + 1)
                constexpr int LO = KG::NI / NW + (MASKED ? 1 : 0);
                const int younger = min(ntl - 1 - j, NST - 2);
                if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (younger == 1) { if (myn == LO) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LO) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LO + 1) : "memory"); }
                else if (younger == 2 || NST <= 4) { if (myn == LO) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LO) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LO + 2) : "memory"); }
                else wait_vmcnt(younger * myn);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            DA_OPB({ const unsigned long long t_ = __builtin_readcyclecounter(); pb_[1] += t_ - pb_t; pb_t = t_; })
            if (j + NST - 1 < ntl && !(VAR & 2048)) issue((VAR & 4096) ? 0 : next_tile(rem_pf, j + NST - 1), (j + NST - 1) % NST);      // (bits 2048 / 4096: timing ablations -- no DMA beyond the prologue / every tile re-fetches tile 0)
            DA_OPB({ const unsigned long long t_ = __builtin_readcyclecounter(); pb_[2] += t_ - pb_t; pb_t = t_; })
            if (!wave_on) continue;
            // classes of this slab's two blocks in the tile (1 = partial when the plan has no class table)
            const unsigned cls2 = (MASKED && crow) ? (unsigned)__builtin_amdgcn_readfirstlane((int)*(const unsigned short *)(crow + 2 * kt)) : 0x0101u;
            const unsigned char *stg = smem + (j % NST) * MSTAGE;
            // VAR bit 512 (narrow heads): the K fragments of BOTH blocks of the tile are requested at its top, so that the second
            // block's LDS round trip passes under the first block's arithmetic (the probe: ~210 cycles of K-read wait per block)
            constexpr bool KPRE = (VAR & 512) != 0 && CF::KB == 2 && CF::NCH <= 2 && !MASKED;
            u32x4 kfa[KPRE ? CF::KB : 1][CF::NCH];
            if constexpr (KPRE) {
#pragma unroll
                for (int kb = 0; kb < CF::KB; ++kb)
#pragma unroll
                    for (int ch = 0; ch < CF::NCH; ++ch) kfa[kb][ch] = *(const u32x4 *)(stg + kfo[ch] + kb * 32 * KG::RS);
            }
#pragma unroll
            for (int kb = 0; kb < CF::KB; ++kb) {
                const int key0 = kt * CF::BKEYS + kb * 32;
                if (key0 >= n_g) break;
                const unsigned cls = MASKED ? ((cls2 >> (8 * kb)) & 3u) : 1u;      // wave-uniform
                if (MASKED && cls == 0u) continue;                                  // no edge between this slab and these keys
                u32x4 kf[CF::NCH];
#pragma unroll
                for (int ch = 0; ch < CF::NCH; ++ch) {
                    if constexpr (KPRE) kf[ch] = kfa[kb][ch];
                    else if (!(VAR & 16) || (j == 0 && kb == 0)) kf[ch] = *(const u32x4 *)(stg + kfo[ch] + kb * 32 * KG::RS);
                }
                if (!(VAR & 1024)) __builtin_amdgcn_sched_barrier(0);      // (bit 1024: let the compiler count lgkmcnt down fragment by fragment)
                DA_OPB({ asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[CF::NCH - 1])); const unsigned long long t_ = __builtin_readcyclecounter(); pb_[3] += t_ - pb_t; pb_t = t_; })
                f32x16 s;
                if (MASKED && cls == 2u) {                                           // every pair of the block is an edge
                    anym = 1u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = 0.f;
                } else if (MASKED) {
                    // this lane's 16 keys of the block: half `half` of the dword lane (i, kb) fetched for the tile
                    const unsigned mw = *(const unsigned short *)(stg + KG::STAGE + wid * 256 + (kb * 32 + i) * 4 + 2 * half);
                    anym |= mw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 b4 = *(const f32x4 *)((const unsigned char *)mlut + (((mw >> (4 * j)) & 15u) << 4));
                        s[4 * j] = b4[0]; s[4 * j + 1] = b4[1]; s[4 * j + 2] = b4[2]; s[4 * j + 3] = b4[3];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = 0.f;
                }
#pragma unroll
                for (int ch = 0; ch < CF::NCH; ++ch) {
                    if (!(VAR & 8) || ch == 0) s = mma_chunk(T(), kf[ch], qf[ch], s);
                    else asm volatile("" ::"v"(kf[ch]));          // ablation: the reads stay, the products go
                }
                u32x2 vlo[2], vhi[2];
                const unsigned vb = lds0 + (unsigned)((j % NST) * MSTAGE + vbase + kb * 32 * CF::RSV);
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    vlo[mm] = tr_read(vb, (8 * mm) * CF::RSV);
                    vhi[mm] = tr_read(vb, (8 * mm + 4) * CF::RSV);
                }
                if (!MASKED) {                      // MASKED: non-edges already sit at -inf (bits beyond n_g and on a missing diagonal are 0)
                    const int kbase = key0 + 16 * half;
                    const bool tail = key0 + 32 > n_g;
                    const bool diag = p.nodiag && key0 < q0 + 32 && key0 + 32 > q0;
                    if (tail || diag) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kbase + r >= n_g || (p.nodiag && kbase + r == qidx)) s[r] = -INFINITY;
                    }
                }
                if (gen) {
                    // classic recurrence: new reference = max(old, block max); state rescaled when it moves
                    const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]);
                    const float a2 = fmaxf(fmaxf(s[6], s[7]), s[8]), a3 = fmaxf(fmaxf(s[9], s[10]), s[11]);
                    const float a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
                    const float mloc = fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), s[15]));
                    const float mnew = fmaxf(m, fmaxf(mloc, __shfl_xor(mloc, 32)));      // >= -1e30: finite
                    if (__any(mnew > m)) {
                        const float corr = __builtin_amdgcn_exp2f(m - mnew);
#pragma unroll
                        for (int r = 0; r < 16; ++r) O[r] *= corr;
                        ls *= corr;
                        ls2 *= corr;
                        m = mnew;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] -= m;
                }
                bf16x8 pf0, pf1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {               // pair by pair, so that at most two exponentials wait for their pack
                    const float x0 = (VAR & 4) ? s[2 * e] : __builtin_amdgcn_exp2f(s[2 * e]), x1 = (VAR & 4) ? s[2 * e + 1] : __builtin_amdgcn_exp2f(s[2 * e + 1]);
                    const float y0 = (VAR & 4) ? s[8 + 2 * e] : __builtin_amdgcn_exp2f(s[8 + 2 * e]), y1 = (VAR & 4) ? s[8 + 2 * e + 1] : __builtin_amdgcn_exp2f(s[8 + 2 * e + 1]);
                    if (VAR & 2) {
                        u32x4 a = __builtin_bit_cast(u32x4, pf0), b = __builtin_bit_cast(u32x4, pf1);
                        a[e] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
                        b[e] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, y1), __builtin_bit_cast(unsigned, y0), 0x07060302u);
                        pf0 = __builtin_bit_cast(bf16x8, a); pf1 = __builtin_bit_cast(bf16x8, b);
                    } else {
                        pf0[2 * e] = (__bf16)x0; pf0[2 * e + 1] = (__bf16)x1;
                        pf1[2 * e] = (__bf16)y0; pf1[2 * e + 1] = (__bf16)y1;
                    }
                    if (VAR & 1) {                          // plain adds: this file is built with -fno-slp-vectorize (SLP would pack them into v_pk_add_f32)
                        ls += x0 + x1;                      // (inline-asm adds here read the exponentials before the transcendental unit
                        ls2 += y0 + y1;                     //  had written them: the compiler cannot see the hazard inside asm)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]));
                const u32x4 v0 = {vlo[0][0], vlo[0][1], vhi[0][0], vhi[0][1]};
                const u32x4 v1 = {vlo[1][0], vlo[1][1], vhi[1][0], vhi[1][1]};
                if (VAR & 32) {                              // ablation: no PV products
                    asm volatile("" ::"v"(v0), "v"(v1), "v"(pf0), "v"(pf1));
                } else {
                    O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf0, O, 0, 0, 0);
                    O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, O, 0, 0, 0);
                }
                if (!(VAR & 1)) {
                    // row sum of the bf16-rounded p (what the PV product weighs with): eight v_dot2_f32_bf16 against (1, 1)
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                    const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16x2 pa = {pf0[2 * e], pf0[2 * e + 1]}, pb = {pf1[2 * e], pf1[2 * e + 1]};
                        ls = __builtin_amdgcn_fdot2_f32_bf16(pa, one2, ls, false);
                        ls = __builtin_amdgcn_fdot2_f32_bf16(pb, one2, ls, false);
                    }
                }
                DA_OPB({ asm volatile("" : "+v"(O), "+v"(ls)); const unsigned long long t_ = __builtin_readcyclecounter(); pb_[4] += t_ - pb_t; pb_t = t_; pb_[6] += 1; })
            }
        }
        }           // (!PIPE || gen)
        if (VAR & 1) ls += ls2;
        if constexpr (EPI_PRE) {
            {       // (every attempt loads them again: kept across a re-run they would be live through the key loop -- 24 spilled registers)
                const int nq_ = min(QT, n_g - qt * QT);
#pragma unroll
                for (int k = 0; k < NBP; ++k) {
                    // (fully defined on every path -- items beyond the tile read its first row: a partial definition would keep the
                    //  previous attempt's values alive through the key loop)
                    const int it = min(tid + NT * k, nq_ * CPR_ - 1);
                    const int q = it / CPR_, ch = it - q * CPR_;
                    const size_t off = (size_t)(node0 + qt * QT + q) * HC + (size_t)h * C + ch * EPC_;
                    skv_p[k] = *(const u32x4 *)((const T *)p.S + off);
                    rsv_p[k] = p.res ? *(const u32x4 *)((const T *)p.res + off) : (u32x4){0u, 0u, 0u, 0u};
                }
            }
        }
        if (gen) break;
        // ---- verification of the optimistic pass (workgroup-uniform verdict: the waves share the K / V stream)
        const float lt0 = ls + __shfl_xor(ls, 32);
        bool ok = lt0 > 8.673617379884035e-19f && lt0 < 1.2676506002282294e30f;        // 2^-60, 2^100; NaN fails
        if (MASKED) {
            const unsigned had = anym | (unsigned)__shfl_xor((int)anym, 32);
            ok = ok || (lt0 == 0.f && had == 0u);                                        // a row without regular edges
        }
        const bool bad = wave_on && __any(!ok && qidx < n_g);
        if (lane == 0) flags[wid] = bad ? 1 : 0;            // (nothing else lives in flags[0 .. 3]; the ring was drained by the last tile's vmcnt(0))
        __syncthreads();                                    // every wave has left the key loop: the ring's LDS is free from here on
        bool redo = false;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) redo = redo || flags[w_] != 0;
        if (!redo) break;
        __syncthreads();
        gen = true;
        if (tid == 0) atomicAdd(&g_opt_fallbacks[MASKED ? 1 : 0], 1ull);
    }

    // ---- epilogue
    DA_OPB(const unsigned long long pb_ep = __builtin_readcyclecounter();)
    DA_OPB(auto pb_out = [&]() { if (p.prof && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); unsigned long long *o_ = p.prof + 16 * (size_t)blockIdx.x;
                                 o_[8] = pb_wall0; o_[9] = wall_clock64(); o_[12] = pb_p1 - pb_start; o_[13] = pb_p2 - pb_p1; o_[14] = pb_p3 - pb_p2; o_[10] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) /* XCC_ID */; o_[11] = __smid();
                                 o_[0] = t_ - pb_start; o_[1] = pb_[1]; o_[2] = pb_[2]; o_[3] = pb_[3]; o_[4] = pb_[4]; o_[5] = t_ - pb_ep; o_[6] = pb_[6]; o_[7] = 1; } };)
    const float lt = ls + __shfl_xor(ls, 32);
    const float inv = lt > 0.f ? 1.0f / (lt + ((gen && !MASKED) ? 1e-16f : 0.f)) : 0.f;      // (see the header: no epsilon on an un-shifted sum;
                                                                                              //  MASKED rows are normalised again after their remainder edges)
    if (FOLD && !MASKED) {
        // folded value heads: CV-wide normalised rows per head, summed over heads by the tail kernel -- no skip, no
        // activation, straight from the accumulator layout
        if (wave_on && qidx < n_g) {
            T *dst = (T *)p.fold_out + ((size_t)h * p.n_rows + node0 + qidx) * CV;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int c0 = 8 * jj + 4 * half;
                const float v4[4] = {O[4 * jj] * inv, O[4 * jj + 1] * inv, O[4 * jj + 2] * inv, O[4 * jj + 3] * inv};
                st4(dst + c0, v4);
            }
        }
        DA_OPB(pb_out();)
        return;
    }
    constexpr int CO = CV, RSOF = CO + 4;
    static_assert(QT * RSOF * 4 <= NST * MSTAGE, "O staging must fit in the K/V ring");
    float *so = (float *)smem;
    if (gen) dma_barrier();       // (an optimistic pass went through the verification barrier: nobody reads the ring any more; workgroup-uniform)
    if (wave_on) {
        float *orow = so + (wid * 32 + i) * RSOF;
        if (MASKED && half == 0) {
            // state for the remainder edges, re-referenced to (ln(sum), 1): reference in nat, rows already normalised by
            // their dense part; and the row's node, for the output addresses below
            orow[CO] = lt > 0.f ? (gen ? m : 0.f) * 0.6931471805599453f + __logf(lt) : 0.f;
            orow[CO + 1] = lt > 0.f ? 1.0f : 0.f;
            orow[CO + 2] = __builtin_bit_cast(float, my_node);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int c0 = 8 * jj + 4 * half;
            *(f32x4 *)(orow + c0) = (f32x4){O[4 * jj] * inv, O[4 * jj + 1] * inv, O[4 * jj + 2] * inv, O[4 * jj + 3] * inv};
        }
    }
    dma_barrier();
    const int nq = min(QT, n_g - qt * QT);
    if (MASKED) {
        remainder_edges<T, CF, CO, RSOF>(p, so, h, np, pad0, n_g, qt * QT, wid, lane, wave_on, rm_beg, rm_end, rm_slot);
        __syncthreads();
    }
    if (FOLD) {                   // MASKED + folded value heads: normalised per-head rows for the tail kernel
        constexpr int CQ = CV / 4;
        for (int it = tid; it < nq * CQ; it += NT) {
            const int q = it / CQ, ch = it - q * CQ;
            const float lr = so[q * RSOF + CO + 1];
            const float ir = lr > 0.f ? 1.0f / (lr + 1e-16f) : 0.f;
            const f32x4 a = *(const f32x4 *)(so + q * RSOF + ch * 4);
            const float v4[4] = {a[0] * ir, a[1] * ir, a[2] * ir, a[3] * ir};
            st4((T *)p.fold_out + ((size_t)h * p.n_rows + __builtin_bit_cast(int, so[q * RSOF + CO + 2])) * CV + ch * 4, v4);
        }
        return;
    }
    constexpr int EPC = 8, CPR = C / EPC;
    constexpr int NB = 3;
    for (int it0 = tid; it0 < nq * CPR; it0 += NT * NB) {
        u32x4 skv[NB], rsv[NB];
        if constexpr (EPI_PRE) {
#pragma unroll
            for (int k = 0; k < NB; ++k) { skv[k] = skv_p[k]; rsv[k] = rsv_p[k]; }
        } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int it = it0 + NT * k;
            if (it < nq * CPR) {
                const int q = it / CPR, ch = it - q * CPR;
                const size_t off = (size_t)(MASKED ? __builtin_bit_cast(int, so[q * RSOF + CO + 2]) : node0 + qt * QT + q) * HC + (size_t)h * C + ch * EPC;
                skv[k] = *(const u32x4 *)((const T *)p.S + off);
                if (p.res) rsv[k] = *(const u32x4 *)((const T *)p.res + off);
            }
        }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int it = it0 + NT * k;
            if (it < nq * CPR) {
                const int q = it / CPR, ch = it - q * CPR;
                const size_t off = (size_t)(MASKED ? __builtin_bit_cast(int, so[q * RSOF + CO + 2]) : node0 + qt * QT + q) * HC + (size_t)h * C + ch * EPC;
                const float *src = so + q * RSOF + ch * EPC;
                float v[EPC], sk[EPC];
                const f32x4 a = *(const f32x4 *)src, b2 = *(const f32x4 *)(src + 4);
                v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b2[0]; v[5] = b2[1]; v[6] = b2[2]; v[7] = b2[3];
                if (MASKED) {                                     // rows carry the sum over dense part + remainder edges
                    const float lr = so[q * RSOF + CO + 1];
                    const float ir = lr > 0.f ? 1.0f / (lr + 1e-16f) : 0.f;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] *= ir;
                }
                unpack_chunk(T(), skv[k], sk);
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] += sk[e];
                if (p.res) {
                    unpack_chunk(T(), rsv[k], sk);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] += sk[e];
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = apply_act(v[e], p.act);
                stc((T *)p.out + off, v);
            }
        }
    }
    DA_OPB(pb_out();)
}

static long long g_virt_launches = 0;      // da_debug_counters [DA_DBG_VIRT_IN_LAUNCH]
long long attn_virt_launches(int reset) {
    return reset ? __atomic_exchange_n(&g_virt_launches, 0ll, __ATOMIC_RELAXED) : __atomic_load_n(&g_virt_launches, __ATOMIC_RELAXED);
}
static thread_local bool t_virt_taken = false;
bool attn_opt_took_virtual_rows() { return t_virt_taken; }

template <int C, bool FOLD, bool MASKED, int NST, int MINB, int BK = 64, int VAR = 0, int NWV = 4>
static int launch_optt(AttnDenseParams p, hipStream_t st) {
    const int lds = NST * (OptK<C, BK>::STAGE + (MASKED ? 1024 : 0)) + 64 + (MASKED ? 256 + 512 : 0);
    static bool attr_done[16] = {};           // per device: the attribute belongs to the device's copy of the function
    int dev = 0;
    DA_CHECK_HIP(hipGetDevice(&dev));
    if (lds > 48 * 1024 && !attr_done[dev & 15]) {
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_optt<C, FOLD, MASKED, NST, MINB, BK, VAR, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done[dev & 15] = true;
    }
    p.nqt = (p.max_nodes + 32 * NWV - 1) / (32 * NWV);
    DA_OPB({ const char *e = DA_XENV_LIVE("DA_OPT_PROF_PTR"); p.prof = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; })
    int grid = p.nqt * p.H * p.n_graphs;
    t_virt_taken = false;
    if constexpr (C == 32 && !FOLD && MASKED && NWV == 4) {
        if (p.v_rows > 0) {
            // workgroups per virtual row: four waves each; sixteen waves per row as the stand-alone kernel had, thirty-two for large Batches (rows
            // with ~900 sources each)
            p.v_split = p.v_n_real >= 8192 ? DA_VIRT_SPLIT_MAX : 4;
            grid += (p.v_rows * p.v_split + 7) & ~7;
            t_virt_taken = true;
            __atomic_fetch_add(&g_virt_launches, 1ll, __ATOMIC_RELAXED);
        }
    } else {
        p.v_rows = 0;
    }
    k_attn_optt<C, FOLD, MASKED, NST, MINB, BK, VAR, NWV><<<grid, 64 * NWV, lds, st>>>(p);
    DA_LAUNCH_CHECK();
    return 0;
}

// ---- K / V-RESIDENT instance for the hidden layers of LARGE complete graphs (round 5, last session).
// What the round's probes measured on k_attn_optt<32> (profiles/r05/NOTES.md): the arithmetic of a 32 x 32 block is under
// half of a wave's time; the rest is a barrier + tile wait per 64 keys, DMA issue, a 30 % half-empty tail per launch, and a
// prologue / epilogue per 128-query tile -- latency at WORKGROUP granularity.  A head of a 900-piece puzzle is 2 x 57.6 KB of K and
// V: it FITS the CU's 160 KB of LDS.  So: ONE workgroup of sixteen waves per (graph, head); the whole K | V of the head is requested
// once (8 x fewer L2 -> LDS bytes than one ring per query tile), in the stage image the ring kernels use; ONE barrier, when
// everything has landed (nothing is ever overwritten: there is no "release" side) -- a barrier of sixteen waves is a convoy, its
// partners share SIMDs, so every further one would re-align waves the matrix pipe and the vector port had just pulled apart
// (measured: one barrier per two tiles 58 us per 32-puzzle launch, three barriers 54, one 52); from there on every wave runs
// 32-query slabs over ALL keys with no barrier, no DMA, no tile wait: a flat loop over 32-key blocks with the NEXT block's K
// fragments requested behind this block's QK chain (they land under the exponentials).  Slabs: a wave's first is its own, the rest
// are drawn from a counter in LDS -- the SIMD's arbiter favours its oldest waves (per-wave stamps, tools/bin/attn_bench_probe
// PROBE3=1: with fixed slabs the waves of one SIMD finished their first slab after 22 k ... 70 k cycles and the last quarter of the
// kernel ran on one or two waves per SIMD).  The optimistic softmax is verified per WAVE (a wave whose row sums left the window
// re-runs its own slab in the running-max mode; nobody else waits), and the epilogue goes straight from the accumulator layout to
// global memory (lane (i, half) owns channels 8 jj + 4 half .. + 3 of query i: 8-byte skip loads and stores) -- no LDS staging,
// no workgroup barrier per slab.  32 puzzles x 8 heads = 256 workgroups = one per CU.  Reference: the same TransformerConv
// attention (backbones/Transformer_GNN.py:32).
template <int NWV, bool QUEUE, bool KPF, int XV = 0>
__global__ __launch_bounds__(64 * NWV, NWV >= 16 ? 1 : ((NWV > 5 || (XV & 8)) ? 4 : 5)) void k_attn_res(AttnDenseParams p) {
    // XV (experiments, A/B through DA_ATTN_RES_PH): bit 0 = two PV accumulators (keys 0 .. 15 / 16 .. 31 of every block: no product waits for
    // the one before it on the same registers); bit 1 = the younger half of the waves at priority 1 (the arbiter favours old waves)
    // bit 4 (16) = PROGRESSIVE LANDING (round 6): no all-or-nothing barrier behind the 120 KB request.  The request order is already
    // tile by tile (wave w holds piece w & 7 of tiles w >> 3, (w >> 3) + 2, ...: tiles complete in order, one per ~1 k cycles, while a wave consumes
    // one per ~2.9 k), so a wave may start on tile 0 as soon as tile 0 is there.  Per 64-key stage an LDS word counts the pieces that have landed:
    // a wave posts piece k of its own after `s_waitcnt vmcnt(pieces issued after it)` -- on a schedule that runs AHEAD of every wave's
    // consumption (at tile t it posts its pieces of tiles <= 3 t + 1, which landed long before anybody needs them), so that nobody waits on a
    // slow partner -- and polls the word of tile t + 1 before it requests that tile's first K fragments.  The only barrier left orders the
    // zeroing of the words before the first post (at kernel start, with the requests in flight).
    // bit 5 (32) = ADJACENCY-MASKED (round 6): the hidden layers of hybrid (Exphander) graphs in the resident form.  Per
    // (32-query slab, 32-key block): class 0 (no edge) blocks are skipped, class 2 (all edges) run unmasked, class 1 blocks take this lane's 16
    // adjacency bits (one 2-byte load, requested one block ahead) as the 0 / -inf initial value of the score accumulator, exactly as
    // k_attn_optt<MASKED> does; a slab's classes are two 64-bit ballots (lane b fetches the class of block b).  The remainder edges of a query (the exophormer's
    // virtual -> real edge, duplicates) are folded in REGISTERS behind the key loop -- score from this lane's Q fragments and the source's K
    // row, the value row's channels this lane owns -- in the same un-shifted (or, in the running-max pass, shifted) softmax state, so the
    // per-wave verification covers them and no LDS staging is needed; output rows go to the slot's node (rm_meta / slot_node).
    // bit 6 (64) = THE LAYER'S PROJECTION IN THE PROLOGUE (round 6; + bit 7 (128): reduction length 128 instead of 256): no projection kernel runs in
    // front of this one.  The workgroup of a (graph, head) projects that head's Q | K | V | skip columns of ITS graph itself, on the matrix pipe,
    // while nothing else is going on in the CU: wave w takes the 32-node slabs w and w + 16.  x rows come from global memory in operand shape
    // (lane (node, half), k-step s: x[node][16 s + 8 half .. + 7]), the head's four 32 x KIN weight blocks are A operands out of LDS (pack_w_qs
    // image: MFMA row a = channel pi(a), bits 2 and 3 swapped, so registers 8 t .. 8 t + 7 of a lane are the 16-byte chunk 2 t + half of its
    // node's row -- in every layout that follows).  The Q and skip weights sit in the last stages of the image, which slabs >= 16 fill last: rows to p.Q / p.S in the
    // projection kernels' own layouts -- the loop below reads them back as it always did; a wave's first slab keeps its Q fragments in
    // registers -- then K and V straight into the resident LDS image (K chunks swizzled, V row-major), where the DMA burst used to put them.
    // Three barriers, all in front of the key loop.  Same 16-step MFMA chain, bias after the reduction, same rounding as the projection
    // kernels: every row is the two-kernel path's bit for bit.  (A wave projecting the NEXT slab's Q and skip between two slabs instead -- the
    // first form of this variant -- costs ~5 us per slab: the loop is bound by each wave's own latency chain, and a second chain of global
    // loads and LDS reads per slab extends it; profiles/r06/r06_qsf_*.log, tools/scratch/withdrawn/r06_qsf_on_the_fly_k_attn_res.diff.)
    // bit 8 (256) = SOFTWARE-PIPELINED KEY LOOP (round 6): the steady-state blocks of an optimistic pass run as generated
    // inline asm on pinned registers, one statement per pair of blocks (tools/gen_attn_res_asm.py -> da_attn_res_asm.inc): the score product of
    // block b + 1 is issued in front of the exponentials of block b, inside ONE wave.  First block, the blocks with a masked tail, graphs without
    // self loops and the running-max pass keep the compiler's loop below.
    constexpr bool O2 = (XV & 1) != 0, PROG = (XV & 16) != 0, MSK = (XV & 32) != 0, QSF = (XV & 64) != 0, PIPE2 = (XV & 256) != 0;
    static_assert(!PIPE2 || (NWV == 16 && KPF && !PROG && !MSK && !O2), "pipelined key loop: the sixteen-wave instances on complete graphs");
    static_assert(!QSF || (NWV == 16 && KPF && QUEUE && !PROG && !MSK && !O2), "projection in the prologue: the sixteen-wave default instance");
    static_assert(!MSK || (NWV == 16 && KPF && !PROG && !O2), "masked resident instance: sixteen waves, K fragments one block ahead");
    static_assert(!PROG || (NWV == 16 && KPF), "progressive landing: the sixteen-wave instance (two waves per piece index), K fragments one block ahead");
    using T = bf16_t;
    constexpr int C = 32, CV = 32;
    using CF = Cfg<T, C, CV, 64>;
    using KG = OptK<C, 64>;
    static_assert(KG::NI == 8, "eight 1 KB pieces per 64-key stage");
    // NWV = 16: the large graphs described above.  NWV = 5 / 8 (SMALL graphs: 129 .. 160 / 161 .. 256 pieces, one slab per wave, no queue): the
    // same kernel as "one workgroup per (graph, head) of a 12 x 12 / 16 x 16 puzzle" -- the ring kernel spends two 128-query tiles on a
    // 144-piece graph (the second with 16 queries) and a prologue, three barriers and an LDS-staged epilogue on five blocks of work; here four
    // workgroups of five waves share a CU, each fetches its 18 KB once and every wave does its five blocks and its own epilogue.
    static_assert(CF::ROWB == 64 && CF::ROWBV == 64 && KG::RS == 64 && CF::RSV == 64 && KG::KBYTES == 4096, "one tile stride for K and V; 2 KB per 32-key block");
    constexpr int TPW = NWV >= 8 ? NWV / 8 : 1;               // (NWV a multiple of 8) an octet of waves fetches every TPW-th tile
    constexpr int STAGE = KG::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DA_OPB(unsigned long long pb_[8] = {__builtin_readcyclecounter(), 0, 0, 0, 0, 0, 0, 0};)
    // head = XCD, as in the ring kernels; QSF: the eight heads of a graph on ONE XCD (workgroup b runs on XCD b % 8) -- they all read the graph's
    // x rows, which then cross the fabric once instead of eight times (K | V of a (graph, head) are read by one workgroup wherever it runs)
    const int bid = (XV & 64) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int h = bid & 7, g = bid >> 3;
    // Batches whose graphs all share one padded slot size (every benched Batch): the slot offset and the tile count follow from the kernel
    // arguments alone, so the DMA below does not wait for the graph table (one dependent L2 round trip at the head of every workgroup)
    const int npg = (p.max_nodes + 63) & ~63;
    const bool upad = (long long)p.n_pad == (long long)p.n_graphs * npg;
    const int pad0 = upad ? g * npg : p.pad_ptr[g];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HC = p.H * C;
    const size_t np = (size_t)p.n_pad;
    const unsigned char *Qg = (const unsigned char *)p.Q + ((size_t)h * np + pad0) * CF::ROWB;
    const unsigned char *Kg = (const unsigned char *)p.K + ((size_t)h * np + pad0) * CF::ROWB;
    const unsigned char *Vg = (const unsigned char *)p.Vt + ((size_t)h * np + pad0) * CF::ROWBV;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned *qctr = (unsigned *)(smem + ((p.max_nodes + 63) >> 6) * STAGE);
    volatile unsigned *land = qctr + 4;         // PROG: [nkt] pieces landed per stage (launch_res sizes the LDS for them)
    float *mlut = (float *)(qctr + 36);         // MSK: nibble -> four accumulator initial values (0 / -inf), 256 B
    if (MSK && tid < 64) mlut[tid] = (((tid >> 2) >> (tid & 3)) & 1) ? 0.f : -INFINITY;
    // QSF: [2][KS][64 lanes][16 B] weight fragments (K block, V block) + 128 bias floats behind the words above (launch_res sizes the LDS); the
    // Q and skip blocks' fragments start out at the head of the (still empty) K | V image
    constexpr int KS = QSF ? ((XV & 128) ? 8 : 16) : 1;          // k-steps of the projection: reduction length 256 (XV = 64) or 128 (XV = 64 + 128)
    [[maybe_unused]] unsigned char *wimg = (unsigned char *)qctr + 512;
    [[maybe_unused]] float *biasl = (float *)(wimg + 2 * KS * 1024);
    if constexpr (QSF) {
        // pack_w_qs image of head h: blocks Q, K, V, skip, KS KB each
        const unsigned char *wsrc = (const unsigned char *)p.wqs + (size_t)h * 4 * KS * 1024;
        const unsigned wl = lds0 + (unsigned)(wimg - smem);
        const int qs_off = ((p.max_nodes + 63) >> 6) * STAGE - 2 * KS * 1024;          // the Q / skip blocks: the last 2 KS KB of the (still empty) image
        for (int j = wid; j < 4 * KS; j += NWV) {
            const int blk = j / KS, s_ = j - blk * KS;           // wave-uniform
            const unsigned dst = (blk == 1 || blk == 2) ? wl + (unsigned)(((blk - 1) * KS + s_) * 1024) : lds0 + (unsigned)(qs_off + ((blk ? 1 : 0) * KS + s_) * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"((unsigned)lane * 16u), "s"(wsrc + (size_t)j * 1024) : "memory");
        }
        if (tid < 128) biasl[tid] = (tid < 32 ? p.bias_q : (tid < 64 ? p.bias_k : (tid < 96 ? p.bias_v : p.bias_s)))[32 * h + (tid & 31)];
    }
    if (QUEUE && tid == 0) { *qctr = 0u; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }      // (nobody draws before the landing barrier)
    if (PROG && tid < 32) { land[tid] = 0u; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    int node0 = 0, n_g = 0;
    if (!upad) { node0 = p.graph_ptr[g]; n_g = p.graph_ptr[g + 1] - node0; }
    const int nkt = upad ? (npg >> 6) : ((n_g + 63) >> 6);          // tiles to fetch (upad: the whole slot -- its rows beyond n_g are readable padding)
    // PROG: the first slab's Q fragments are requested FIRST (the oldest vector-memory operations of the wave: every counted wait on a DMA piece
    // below covers them) and by asm loads -- a load the compiler tracks would get ITS `vmcnt(0)` in front of the first QK product, i.e. a wait
    // for all 120 KB.  (Slab = wave index < 16 <= slabs of any graph this instance takes: no clamp, no graph table needed.)
    u32x4 qf0[2] = {};
    if constexpr (PROG) {
        const unsigned char *qrow = Qg + (size_t)(wid * 32 + i) * CF::ROWB + half * 16;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:32" : "=&v"(qf0[0]), "=&v"(qf0[1]) : "v"(qrow) : "memory");
    }

    // ---- the whole K | V of the head: wave w issues piece (w & 7) of tiles (w >> 3), (w >> 3) + TPW, ... (scalar base + lane offset form)
    if constexpr (QSF) {
        // (no DMA of K | V: the prologue below projects them into the image)
    } else if constexpr (NWV % 8 != 0) {
        // any number of waves: the 8 nkt pieces dealt out round-robin (the lane offset depends on the piece: recomputed per instruction)
        for (int idx = wid; idx < nkt * 8; idx += NWV) {
            const int kt = idx >> 3, pq = idx & 7;          // wave-uniform
            unsigned so_;
            if (pq < KG::NIK) {
                const int s = pq * 64 + lane, row = s / KG::KSPR, col = s - row * KG::KSPR;
                so_ = (unsigned)(row * CF::ROWB + (col ^ KG::f(row)) * 16);
            } else {
                const int s = (pq - KG::NIK) * 64 + lane, row = s / CF::VSPR, col = s - row * CF::VSPR;
                so_ = (unsigned)(row * CF::ROWBV + col * 16);
            }
            const unsigned long long sa = (unsigned long long)(size_t)((pq < KG::NIK ? Kg : Vg) + (size_t)kt * 64 * CF::ROWB);
            // (wave-uniform by construction; said explicitly, the address lands in scalar registers)
            const unsigned long long su = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa) |
                                          ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32)) << 32);
            const unsigned ml = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(kt * STAGE + pq * 1024)));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ml), "v"(so_), "s"((const unsigned char *)(size_t)su) : "memory");
        }
    } else {
        const int pq = wid & 7, t0 = wid >> 3;
        unsigned so_;
        if (pq < KG::NIK) {
            const int s = pq * 64 + lane, row = s / KG::KSPR, col = s - row * KG::KSPR;
            so_ = (unsigned)(row * CF::ROWB + (col ^ KG::f(row)) * 16);
        } else {
            const int s = (pq - KG::NIK) * 64 + lane, row = s / CF::VSPR, col = s - row * CF::VSPR;
            so_ = (unsigned)(row * CF::ROWBV + col * 16);
        }
        const unsigned char *src0 = pq < KG::NIK ? Kg : Vg;
        for (int kt = t0; kt < nkt; kt += TPW) {
            const unsigned char *src = src0 + (size_t)kt * 64 * CF::ROWB;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds0 + (unsigned)(kt * STAGE + pq * 1024)), "v"(so_), "s"(src)
                         : "memory");
        }
    }
    DA_OPB(pb_[1] = __builtin_readcyclecounter();)
    if (upad) {          // (behind the DMA: the asm's memory clobber keeps these loads here)
        if constexpr (PROG) {
            // a SCALAR load by hand: the compiler's own choice here is a vector load, whose `vmcnt(0)` would wait for all 120 KB
            typedef __attribute__((ext_vector_type(2))) int i32x2_;
            i32x2_ gp2;
            const int32_t *gpp = p.graph_ptr + g;
            asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gp2) : "s"(gpp) : "memory");
            node0 = gp2[0]; n_g = gp2[1] - gp2[0];
        } else { node0 = p.graph_ptr[g]; n_g = p.graph_ptr[g + 1] - node0; }
    }
    const int nslab = (n_g + 31) >> 5;      // 32-query slabs = 32-key blocks
    int slab = wid;
    u32x4 qf[CF::NCH];
    auto load_q = [&](int sl, u32x4(&dst)[CF::NCH]) {
        const unsigned char *qrow = Qg + (size_t)(max(min(sl, nslab - 1), 0) * 32 + i) * CF::ROWB;     // (rows of the 64-row slot padding are readable)
#pragma unroll
        for (int ch = 0; ch < CF::NCH; ++ch) dst[ch] = *(const u32x4 *)(qrow + ch * 32 + half * 16);
    };
    static_assert(!PROG || CF::NCH == 2, "progressive landing: two Q fragments per lane");
    // ---- QSF: the projection prologue (see the header of the kernel)
    [[maybe_unused]] u32x4 xq[KS];          // a slab's x rows, every k-step (one memory round trip per slab)
    [[maybe_unused]] auto x_issue = [&](int sl) {
        const int nd = max(min(sl * 32 + i, n_g - 1), 0);          // (rows behind the graph repeat its last row: finite K / V rows, never stored Q / skip rows)
        const unsigned char *row = (const unsigned char *)p.x + ((size_t)(node0 + nd) * (size_t)p.ldx + 8 * half) * 2;
        // (p.debug == 77, timing only, WRONG results: the same bytes as 1 KB-contiguous loads, as if x were stored fragment-major)
        if (p.debug == 77) row = (const unsigned char *)p.x + ((size_t)(node0 + max(min(sl * 32, n_g - 32), 0)) * (size_t)p.ldx) * 2 + lane * 16;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) xq[s_] = *(const u32x4 *)(row + s_ * (p.debug == 77 ? 1024 : 32));
    };
    // two chains over the slab's x fragments: a0 = W0 x^T, a1 = W1 x^T (channel x node); weight fragments one k-step ahead of their products
    [[maybe_unused]] auto chain2 = [&](const unsigned char *w0b, const unsigned char *w1b, f32x16 &a0, f32x16 &a1) {
        unsigned lo = (unsigned)lane * 16u;
        asm volatile("" : "+v"(lo));          // (keeps the weight-fragment reads where they are used: hoisted they are 128 registers)
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
        u32x4 wa0 = *(const u32x4 *)(w0b + lo), wa1 = *(const u32x4 *)(w1b + lo);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            u32x4 wn0 = wa0, wn1 = wa1;
            if (s_ + 1 < KS) { wn0 = *(const u32x4 *)(w0b + (size_t)(s_ + 1) * 1024 + lo); wn1 = *(const u32x4 *)(w1b + (size_t)(s_ + 1) * 1024 + lo); }
            __builtin_amdgcn_sched_barrier(0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa0), __builtin_bit_cast(bf16x8, xq[s_]), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa1), __builtin_bit_cast(bf16x8, xq[s_]), a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wa0 = wn0; wa1 = wn1;
        }
    };
    // chunk t (of this lane's two: 2 t + half) of a projected row: accumulator registers 8 t .. 8 t + 7 + bias, rounded to bf16
    [[maybe_unused]] auto chunk_of = [&](const f32x16 &a, int blk, int t) {
        const f32x4 b0 = *(const f32x4 *)(biasl + 32 * blk + 16 * t + 8 * half), b1 = *(const f32x4 *)(biasl + 32 * blk + 16 * t + 8 * half + 4);
        bf16x8 q8;
#pragma unroll
        for (int e = 0; e < 4; ++e) { q8[e] = (__bf16)(a[8 * t + e] + b0[e]); q8[4 + e] = (__bf16)(a[8 * t + 4 + e] + b1[e]); }
        return __builtin_bit_cast(u32x4, q8);
    };
    // Q and skip rows of slab sl (xq holds it) -> memory, in the projection kernels' layouts; keep = this wave's first slab: its Q fragments stay in qf
    [[maybe_unused]] auto project_qs = [&](int sl, bool keep) {
        f32x16 aQ, aS;
        const unsigned char *wq_ = smem + (size_t)(((p.max_nodes + 63) >> 6) * STAGE - 2 * KS * 1024);
        chain2(wq_, wq_ + (size_t)KS * 1024, aQ, aS);
        const int row = sl * 32 + i;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4 qc = chunk_of(aQ, 0, t), sc_ = chunk_of(aS, 3, t);
            if (keep) qf[t] = qc;
            if (row < n_g) {
                *(u32x4 *)(const_cast<unsigned char *>(Qg) + (size_t)row * CF::ROWB + (2 * t + half) * 16) = qc;
                *(u32x4 *)((unsigned char *)p.S + ((size_t)(node0 + row) * HC + (size_t)h * C) * 2 + (2 * t + half) * 16) = sc_;
            }
        }
    };
    // K and V rows of slab sl (xq holds it) -> the resident image (stage sl / 2: K rows with the chunk swizzle of the DMA image, V rows behind them)
    [[maybe_unused]] auto project_kv = [&](int sl) {
        f32x16 aK, aV;
        chain2(wimg, wimg + (size_t)KS * 1024, aK, aV);
        const int rt = (sl & 1) * 32 + i;          // row inside the 64-key stage
        unsigned char *st_ = smem + (size_t)(sl >> 1) * STAGE + (size_t)rt * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            *(u32x4 *)(st_ + (((2 * t + half) ^ KG::f(rt)) << 4)) = chunk_of(aK, 1, t);
            *(u32x4 *)(st_ + KG::KBYTES + ((2 * t + half) << 4)) = chunk_of(aV, 2, t);
        }
    };
    if constexpr (QSF) x_issue(slab);
    else if constexpr (!PROG) load_q(slab, qf);    else if constexpr (!PROG) load_q(slab, qf);
    // PROG: this wave's pieces (tiles t0p, t0p + 2, ...: nmine of them, issued in that order, with nothing older than them outstanding but the two
    // Q loads above and nothing younger) and how many it has posted
    [[maybe_unused]] const int t0p = wid >> 3, nmine = nkt > t0p ? (nkt - t0p + 1) >> 1 : 0;
    [[maybe_unused]] int posted = 0;
    [[maybe_unused]] auto wait_le = [&](int n) {           // s_waitcnt vmcnt(n), n a run-time count (<= 9: ten pieces of nineteen stages)
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        }
    };
    // post every piece of mine whose tile index is <= tmax: piece k has landed once at most the nmine - 1 - k pieces issued after it are outstanding
    [[maybe_unused]] auto post_upto = [&](int tmax) {
        while (posted < nmine && t0p + 2 * posted <= tmax) {
            wait_le(nmine - 1 - posted);
            if (lane == 0) __hip_atomic_fetch_add((unsigned *)land + t0p + 2 * posted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++posted;
        }
    };
    // (asm read: a volatile access makes the compiler wait for EVERY outstanding memory operation -- vmcnt(0): all 120 KB -- in front of it)
    [[maybe_unused]] auto await_tile = [&](int t) {
        const unsigned a_ = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)(land + t);
        unsigned v_;
        do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v_) : "v"(a_) : "memory");
            if (__builtin_amdgcn_readfirstlane((int)v_) >= 8) break;
            __builtin_amdgcn_s_sleep(1);
        } while (true);
    };
    if constexpr (PROG) {
        __builtin_amdgcn_s_barrier();          // the landing words are zero (the requests are in flight; nobody has waited for anything yet)
        post_upto(slab < nslab ? 1 : (1 << 30));                // (a wave without a slab still owes all its pieces)
        if (nmine == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (every wave of this instance holds at least one piece of tile 0 or 1: the wait above covered the Q loads, which are older)
        qf[0] = qf0[0]; qf[1] = qf0[1];
        asm volatile("" : "+v"(qf[0]), "+v"(qf[1]));
        if (slab < nslab) await_tile(0);
    } else {
    if constexpr (QSF) {
        const int slabB = slab + NWV;
        const bool hasA = slab < nslab, hasB = slabB < nslab;
        // weight fragments, bias words and slab A's x rows are there; barrier 1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // slab A (= wave index: stages 0 .. 7 of the image) completely, then Q / skip of slab B; every x row is read ONCE per workgroup
        if (hasA) { project_qs(slab, true); project_kv(slab); }
        if (hasB) { x_issue(slabB); project_qs(slabB, false); }
        __builtin_amdgcn_s_barrier();          // 2: nobody reads the Q / skip weights (in the last stages of the K | V image) any more
        if (hasB) project_kv(slabB);
        // the Q / skip rows have reached the L2, the K | V rows the LDS; barrier 3, the last one of the kernel
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
    // the first slab's Q fragments travel behind the DMA; one wait for both, then the only barrier of the kernel
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[CF::NCH - 1]));
    __builtin_amdgcn_s_barrier();
    }
    }
    DA_OPB(pb_[6] = __builtin_readcyclecounter();)

    const int pi_i = (i & 3) + 4 * ((i >> 3) & 3) + 16 * ((i >> 2) & 1);
    int kfo[CF::NCH];
#pragma unroll
    for (int ch = 0; ch < CF::NCH; ++ch) kfo[ch] = pi_i * KG::RS + (((2 * ch + half) ^ KG::f(pi_i)) * 16);
    const int li = lane & 15;
    const unsigned vbase = lds0 + (unsigned)(KG::KBYTES + (16 * half + (li >> 2)) * CF::RSV + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2);

    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};
    f32x16 O, Ob;
    float ls, m;
    if (XV & 2) { if (wid >= NWV / 2) __builtin_amdgcn_s_setprio(1); }
    // byte offset of 32-key block b inside the K (or, + KBYTES, the V) image: stage b / 2, block b % 2
    auto boff = [&](int b) { return (b >> 1) * STAGE + (b & 1) * 2048; };

    for (int round = 0; slab < nslab; ++round) {
        const int q0 = slab * 32, qidx = q0 + i;
        bool gen = p.force_gen != 0;
        auto pass = [&](const bool first) {          // first (PROG): this wave's first walk over the keys -- tiles may still be on their way
#pragma unroll
            for (int r = 0; r < 16; ++r) { O[r] = 0.f; if (O2) Ob[r] = 0.f; }
            ls = 0.f;
            m = -1e30f;
            u32x4 kf[CF::NCH];
#pragma unroll
            for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(smem + kfo[ch]);
            int b_first = 0;                      // PIPE2: the compiler's loop starts here, with this block's scores already in sA
            [[maybe_unused]] f32x16 sA;
            if constexpr (PIPE2) {
                static_assert(!PIPE2 || CF::NCH == 2, "two K fragments per block");
                const int nfull = n_g >> 5;       // blocks without a masked key
                if (!gen && !p.nodiag && nfull >= 3) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sA[r] = 0.f;
#pragma unroll
                    for (int ch = 0; ch < CF::NCH; ++ch) sA = mma_chunk(T(), kf[ch], qf[ch], sA);          // scores of block 0
                    u32x4 kfA0 = *(const u32x4 *)(smem + boff(1) + kfo[0]), kfA1 = *(const u32x4 *)(smem + boff(1) + kfo[1]);
                    const unsigned ones = 0x3f803f80u;
                    // (sA comes out of a compiler MFMA and the statement's first vector reads of it are invisible to the compiler: 16 wait states by hand)
                    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sA), "+v"(kfA0), "+v"(kfA1));
                    int b = 0;
                    for (; b + 1 < nfull && b + 3 < nslab; b += 2) {
                        const unsigned kad0 = lds0 + (unsigned)(boff(b + 2) + kfo[0]), kad1 = lds0 + (unsigned)(boff(b + 2) + kfo[1]);
                        const unsigned vad = vbase + (unsigned)boff(b);
                        constexpr int ABL = (XV >> 9) & 7;          // (the generator's timing ablations, WRONG results: DA_ATTN_RES_PIPE = 2 .. 6)
                        if constexpr (ABL == 1) DA_RES_PAIR_NOWAIT();
                        else if constexpr (ABL == 2) DA_RES_PAIR_NOVALU();
                        else if constexpr (ABL == 3) DA_RES_PAIR_NOMFMA();
                        else if constexpr (ABL == 4) DA_RES_PAIR_NOREAD();
                        else if constexpr (ABL == 5) DA_RES_PAIR_HALFEXP();
                        else DA_RES_PAIR();
                    }
                    // K fragments of block b + 1 are still on their way; sA / O were last written by asm MFMAs
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(sA), "+v"(kfA0), "+v"(kfA1), "+v"(O), "+v"(ls));
                    kf[0] = kfA0; kf[1] = kfA1;
                    b_first = b;
                }
            }
            for (int b = b_first; b < nslab; ++b) {
                const int key0 = b * 32;
                const int bo = boff(b);
                const bool given = PIPE2 && b_first > 0 && b == b_first;          // scores in sA, kf already holds block b + 1
                if (PROG && first && !(b & 1)) post_upto(3 * (b >> 1) + 1);          // ahead of everybody's consumption (see the header)
                if (!KPF && b > 0) {
#pragma unroll
                    for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(smem + bo + kfo[ch]);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x16 s;
                if (given) {
                    if constexpr (PIPE2) s = sA;
                } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                for (int ch = 0; ch < CF::NCH; ++ch) s = mma_chunk(T(), kf[ch], qf[ch], s);
                }
                asm volatile("" : "+v"(s));         // (the V reads and the next K fragments stay BEHIND the QK chain: the IR-level sinking of the products put
                                                    //  the reads in front, where the chain's lgkmcnt wait covers them too)
                u32x2 vlo[2], vhi[2];
                const unsigned vb = vbase + (unsigned)bo;
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    vlo[mm] = tr_read(vb, (8 * mm) * CF::RSV);
                    vhi[mm] = tr_read(vb, (8 * mm + 4) * CF::RSV);
                }
                if (KPF && b + 1 < nslab && !given) {       // next block's K fragments: they land under this block's exponentials
                    if (PROG && first && (b & 1)) await_tile((b + 1) >> 1);          // (a new tile: all eight pieces there?)
                    const int bn = boff(b + 1);
#pragma unroll
                    for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(smem + bn + kfo[ch]);
                }
                {
                    const int kbase = key0 + 16 * half;
                    const bool tail = key0 + 32 > n_g;
                    const bool diag = p.nodiag && key0 == q0;
                    if (tail || diag) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kbase + r >= n_g || (p.nodiag && kbase + r == qidx)) s[r] = -INFINITY;
                    }
                }
                if (gen) {
                    const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]);
                    const float a2 = fmaxf(fmaxf(s[6], s[7]), s[8]), a3 = fmaxf(fmaxf(s[9], s[10]), s[11]);
                    const float a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
                    const float mloc = fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), s[15]));
                    const float mnew = fmaxf(m, fmaxf(mloc, __shfl_xor(mloc, 32)));
                    if (__any(mnew > m)) {
                        const float corr = __builtin_amdgcn_exp2f(m - mnew);
#pragma unroll
                        for (int r = 0; r < 16; ++r) { O[r] *= corr; if (O2) Ob[r] *= corr; }
                        ls *= corr;
                        m = mnew;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] -= m;
                }
                bf16x8 pf0, pf1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = __builtin_amdgcn_exp2f(s[2 * e]), x1 = __builtin_amdgcn_exp2f(s[2 * e + 1]);
                    const float y0 = __builtin_amdgcn_exp2f(s[8 + 2 * e]), y1 = __builtin_amdgcn_exp2f(s[8 + 2 * e + 1]);
                    pf0[2 * e] = (__bf16)x0; pf0[2 * e + 1] = (__bf16)x1;
                    pf1[2 * e] = (__bf16)y0; pf1[2 * e + 1] = (__bf16)y1;
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]));
                const u32x4 v0 = {vlo[0][0], vlo[0][1], vhi[0][0], vhi[0][1]};
                const u32x4 v1 = {vlo[1][0], vlo[1][1], vhi[1][0], vhi[1][1]};
                O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf0, O, 0, 0, 0);
                if (O2) Ob = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, Ob, 0, 0, 0);
                else O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, O, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2 pa = {pf0[2 * e], pf0[2 * e + 1]}, pb = {pf1[2 * e], pf1[2 * e + 1]};
                    ls = __builtin_amdgcn_fdot2_f32_bf16(pa, one2, ls, false);
                    ls = __builtin_amdgcn_fdot2_f32_bf16(pb, one2, ls, false);
                }
            }
        };
        // ---- MSK: this slab's block classes, this lane's adjacency row and its query's remainder record
        [[maybe_unused]] unsigned long long nzm = 0, fullm = 0;          // blocks with some edge / with every edge
        [[maybe_unused]] const unsigned char *mrow2 = nullptr;           // this lane's 16 bits of block b: *(ushort *)(mrow2 + 4 b)
        [[maybe_unused]] int rbeg = 0, rend = 0, rslot = 0, my_node = node0 + min(qidx, n_g - 1);
        [[maybe_unused]] unsigned anym = 0;                               // this lane saw an edge (any adjacency bit, a full block, a remainder edge)
        if constexpr (MSK) {
            const int slots = p.pad_ptr[g + 1] - pad0;
            unsigned cl = 1u;                                             // (no class table: every block partial)
            if (p.blk_class && lane < nslab) cl = (p.blk_class + p.blk_class_ptr[g] + (size_t)slab * (size_t)p.blk_class_stride)[lane];
            if (lane >= nslab) cl = 0u;
            nzm = __ballot(cl != 0u);
            fullm = __ballot(cl == 2u);
            mrow2 = p.mask + p.mask_ptr[g] + (size_t)min(qidx, n_g - 1) * (size_t)(slots >> 3) + 2 * half;
            if (p.rm_meta) {
                typedef __attribute__((ext_vector_type(4))) int i32x4;
                const i32x4 mt = ((const i32x4 *)p.rm_meta + pad0)[min(qidx, n_g - 1)];
                rbeg = qidx < n_g ? mt[0] : 0; rend = qidx < n_g ? mt[1] : 0; rslot = mt[2]; my_node = mt[3];
            } else {
                my_node = p.slot_node ? p.slot_node[pad0 + min(qidx, n_g - 1)] : node0 + min(qidx, n_g - 1);
                if (qidx < n_g) { rbeg = p.irr_row_ptr[my_node]; rend = p.irr_row_ptr[my_node + 1]; }
                rslot = rend > rbeg ? p.row_map[p.irr_col_src[rbeg]] : pad0;
            }
        }
        auto pass_masked = [&]() {
#pragma unroll
            for (int r = 0; r < 16; ++r) O[r] = 0.f;
            ls = 0.f;
            m = -1e30f;
            unsigned long long rem = nzm;
            if (rem) {
                int b = __builtin_ctzll(rem);
                rem &= rem - 1ull;
                u32x4 kf[CF::NCH];
#pragma unroll
                for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(smem + boff(b) + kfo[ch]);
                unsigned mw_n = ((fullm >> b) & 1ull) ? 0xffffu : *(const unsigned short *)(mrow2 + 4 * b);
                while (true) {
                    const int bo = boff(b);
                    const bool full = (fullm >> b) & 1ull;                 // wave-uniform
                    const unsigned mw = mw_n;
                    anym |= mw;
                    __builtin_amdgcn_sched_barrier(0);
                    f32x16 s;
                    if (full) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[r] = 0.f;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 b4 = *(const f32x4 *)((const unsigned char *)mlut + (((mw >> (4 * j)) & 15u) << 4));
                            s[4 * j] = b4[0]; s[4 * j + 1] = b4[1]; s[4 * j + 2] = b4[2]; s[4 * j + 3] = b4[3];
                        }
                    }
#pragma unroll
                    for (int ch = 0; ch < CF::NCH; ++ch) s = mma_chunk(T(), kf[ch], qf[ch], s);
                    asm volatile("" : "+v"(s));
                    u32x2 vlo[2], vhi[2];
                    const unsigned vb = vbase + (unsigned)bo;
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        vlo[mm] = tr_read(vb, (8 * mm) * CF::RSV);
                        vhi[mm] = tr_read(vb, (8 * mm + 4) * CF::RSV);
                    }
                    const bool last = rem == 0ull;
                    int bn = b;
                    if (!last) {                 // the next block with an edge: its K fragments (and adjacency bits) land under this block's exponentials
                        bn = __builtin_ctzll(rem);
                        rem &= rem - 1ull;
                        const int bon = boff(bn);
#pragma unroll
                        for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(smem + bon + kfo[ch]);
                        mw_n = ((fullm >> bn) & 1ull) ? 0xffffu : *(const unsigned short *)(mrow2 + 4 * bn);
                    }
                    if (gen) {
                        const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]);
                        const float a2 = fmaxf(fmaxf(s[6], s[7]), s[8]), a3 = fmaxf(fmaxf(s[9], s[10]), s[11]);
                        const float a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
                        const float mloc = fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), s[15]));
                        const float mnew = fmaxf(m, fmaxf(mloc, __shfl_xor(mloc, 32)));
                        if (__any(mnew > m)) {
                            const float corr = __builtin_amdgcn_exp2f(m - mnew);
#pragma unroll
                            for (int r = 0; r < 16; ++r) O[r] *= corr;
                            ls *= corr;
                            m = mnew;
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[r] -= m;
                    }
                    bf16x8 pf0, pf1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = __builtin_amdgcn_exp2f(s[2 * e]), x1 = __builtin_amdgcn_exp2f(s[2 * e + 1]);
                        const float y0 = __builtin_amdgcn_exp2f(s[8 + 2 * e]), y1 = __builtin_amdgcn_exp2f(s[8 + 2 * e + 1]);
                        pf0[2 * e] = (__bf16)x0; pf0[2 * e + 1] = (__bf16)x1;
                        pf1[2 * e] = (__bf16)y0; pf1[2 * e + 1] = (__bf16)y1;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]));
                    const u32x4 v0 = {vlo[0][0], vlo[0][1], vhi[0][0], vhi[0][1]};
                    const u32x4 v1 = {vlo[1][0], vlo[1][1], vhi[1][0], vhi[1][1]};
                    O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf0, O, 0, 0, 0);
                    O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, O, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16x2 pa = {pf0[2 * e], pf0[2 * e + 1]}, pb = {pf1[2 * e], pf1[2 * e + 1]};
                        ls = __builtin_amdgcn_fdot2_f32_bf16(pa, one2, ls, false);
                        ls = __builtin_amdgcn_fdot2_f32_bf16(pb, one2, ls, false);
                    }
                    if (last) break;
                    b = bn;
                }
            }
            // ---- the query's remainder edges, in the same softmax state.  Lane (i, half) holds channels 16 ch + 8 half .. + 7 of its query in
            // qf[ch] (the score's other half comes from lane i of the other half-wave) and channels 8 jj + 4 half .. + 3 of the output in O.
            int nrem = rend - rbeg;                                        // the wave walks as many rounds as its longest list
#pragma unroll
            for (int o_ = 32; o_ > 0; o_ >>= 1) nrem = max(nrem, __shfl_xor(nrem, o_));
            nrem = __builtin_amdgcn_readfirstlane(nrem);
            for (int k = 0; k < nrem; ++k) {
                const bool on = rbeg + k < rend;
                size_t sj = (size_t)h * np + (size_t)(k == 0 ? rslot : p.row_map[p.irr_col_src[on ? rbeg + k : max(rbeg, 0)]]);
                const unsigned char *kr = (const unsigned char *)p.K + sj * CF::ROWB, *vr = (const unsigned char *)p.Vt + sj * CF::ROWBV;
                float sc_ = 0.f;
#pragma unroll
                for (int ch = 0; ch < CF::NCH; ++ch) {
                    const u32x4 kk = *(const u32x4 *)(kr + ch * 32 + half * 16);
#pragma unroll
                    for (int w_ = 0; w_ < 4; ++w_) {
                        sc_ = fmaf(bf2f((bf16_t)(qf[ch][w_] & 0xffff)), bf2f((bf16_t)(kk[w_] & 0xffff)), sc_);
                        sc_ = fmaf(bf2f((bf16_t)(qf[ch][w_] >> 16)), bf2f((bf16_t)(kk[w_] >> 16)), sc_);
                    }
                }
                sc_ += __shfl_xor(sc_, 32);                               // log2 units (Q is pre-scaled)
                u32x2 vv[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) vv[jj] = *(const u32x2 *)(vr + (8 * jj + 4 * half) * 2);
                if (gen) {
                    const float mnew = on ? fmaxf(m, sc_) : m;
                    if (mnew > m) {
                        const float corr = __builtin_amdgcn_exp2f(m - mnew);
#pragma unroll
                        for (int r = 0; r < 16; ++r) O[r] *= corr;
                        ls *= corr;
                        m = mnew;
                    }
                    sc_ -= m;
                }
                const float pe = on ? __builtin_amdgcn_exp2f(sc_) : 0.f;
                anym |= on ? 1u : 0u;
                if (half == 0) ls += pe;                                  // (the row sum is the two half-waves' shares added up)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    O[4 * jj] = fmaf(pe, bf2f((bf16_t)(vv[jj][0] & 0xffff)), O[4 * jj]);
                    O[4 * jj + 1] = fmaf(pe, bf2f((bf16_t)(vv[jj][0] >> 16)), O[4 * jj + 1]);
                    O[4 * jj + 2] = fmaf(pe, bf2f((bf16_t)(vv[jj][1] & 0xffff)), O[4 * jj + 2]);
                    O[4 * jj + 3] = fmaf(pe, bf2f((bf16_t)(vv[jj][1] >> 16)), O[4 * jj + 3]);
                }
            }
        };
        if constexpr (MSK) pass_masked(); else
        pass(PROG && round == 0);
        if (PROG && round == 0) post_upto(1 << 30);          // (every piece posted before this wave issues another vector-memory operation)
        if (O2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) O[r] += Ob[r];
        }
        DA_OPB({ asm volatile("" : "+v"(O), "+v"(ls)); if (round < 2) pb_[2 + 2 * round] = __builtin_readcyclecounter(); })
        if (!gen) {
            // verification of the optimistic pass, per WAVE (the waves share nothing but the resident tiles)
            const float lt0 = ls + __shfl_xor(ls, 32);
            bool ok = lt0 > 8.673617379884035e-19f && lt0 < 1.2676506002282294e30f;        // 2^-60, 2^100; NaN fails
            if (MSK) ok = ok || (lt0 == 0.f && (anym | (unsigned)__shfl_xor((int)anym, 32)) == 0u);          // a row without a single edge
            if (__any(!ok && qidx < n_g)) {
                gen = true;
                if (lane == 0) atomicAdd(&g_opt_fallbacks[MSK ? 1 : 0], 1ull);
                if constexpr (MSK) pass_masked(); else
                pass(false);
                if (O2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[r] += Ob[r];
                }
            }
        }
        // the next slab (drawn from the queue) and its Q fragments, which travel under this slab's epilogue
        int nxt = slab + NWV;
        if (QUEUE) {
            unsigned t_ = 0;
            if (lane == 0) t_ = __hip_atomic_fetch_add(qctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nxt = NWV + __builtin_amdgcn_readfirstlane((int)t_);
        }
        [[maybe_unused]] u32x4 qn[CF::NCH];
        const bool more = nxt < nslab;
        if (more) load_q(nxt, qn);
        const float lt = ls + __shfl_xor(ls, 32);
        const float inv = lt > 0.f ? 1.0f / (lt + (gen ? 1e-16f : 0.f)) : 0.f;      // (no epsilon on an un-shifted sum: see the header)
        if (qidx < n_g) {
            const size_t roff = (size_t)(MSK ? my_node : node0 + qidx) * HC + (size_t)h * C + 4 * half;
            u32x2 sk[4], rs[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                sk[jj] = *(const u32x2 *)((const T *)p.S + roff + 8 * jj);
                rs[jj] = p.res ? *(const u32x2 *)((const T *)p.res + roff + 8 * jj) : (u32x2){0u, 0u};
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                // (products rounded on their own, as the ring kernel's are on their way through LDS: no contraction with the skip add)
                float v4[4] = {__fmul_rn(O[4 * jj], inv), __fmul_rn(O[4 * jj + 1], inv), __fmul_rn(O[4 * jj + 2], inv), __fmul_rn(O[4 * jj + 3], inv)};
                // (skip first, then the residual: the ring kernel's order -- the two kernels give bit-identical rows)
                v4[0] += bf2f((bf16_t)(sk[jj][0] & 0xffff)); v4[1] += bf2f((bf16_t)(sk[jj][0] >> 16));
                v4[2] += bf2f((bf16_t)(sk[jj][1] & 0xffff)); v4[3] += bf2f((bf16_t)(sk[jj][1] >> 16));
                if (p.res) {
                    v4[0] += bf2f((bf16_t)(rs[jj][0] & 0xffff)); v4[1] += bf2f((bf16_t)(rs[jj][0] >> 16));
                    v4[2] += bf2f((bf16_t)(rs[jj][1] & 0xffff)); v4[3] += bf2f((bf16_t)(rs[jj][1] >> 16));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = apply_act(v4[e], p.act);
                st4((T *)p.out + roff + 8 * jj, v4);
            }
        }
        if (more) {
#pragma unroll
            for (int ch = 0; ch < CF::NCH; ++ch) qf[ch] = qn[ch];
            // (a use the compiler sees: its wait for these loads belongs HERE -- left to the first QK product it would sit inside the key loop)
            asm volatile("" : "+v"(qf[0]), "+v"(qf[CF::NCH - 1]));
        }
        slab = nxt;
        DA_OPB({ if (round < 2) pb_[3 + 2 * round] = __builtin_readcyclecounter(); })
    }
    DA_OPB({ if (p.prof && lane == 0) { pb_[7] = __builtin_readcyclecounter(); unsigned long long *o_ = p.prof + ((size_t)blockIdx.x * NWV + wid) * 8;
             for (int k = 0; k < 8; ++k) o_[k] = pb_[k]; } })
}

static long long g_res_launches = 0;       // da_debug_counters [DA_DBG_RES_LAUNCHES]: launches of k_attn_res (tests: "the resident kernel took this layer")
long long attn_res_launches(int reset) {
    return reset ? __atomic_exchange_n(&g_res_launches, 0ll, __ATOMIC_RELAXED) : __atomic_load_n(&g_res_launches, __ATOMIC_RELAXED);
}

// projection in the prologue (k_attn_res<16, true, true, 64 (+ 128)>): LDS = the K | V tiles + 512 B of words + two weight blocks + 128 bias floats
// (the other two weight blocks start out inside the K | V image: it must hold them -- 2 kin / 16 KB <= four 8 KB stages at kin = 256)
static size_t attn_res_qsf_lds(int max_nodes, int kin) { return (size_t)((max_nodes + 63) >> 6) * OptK<32, 64>::STAGE + 512 + (size_t)2 * (kin >> 4) * 1024 + 512; }
static bool attn_res_takes(int max_nodes, int n_pad, int n_graphs) {          // launch_attn_opt's test for the resident kernel (complete graphs)
    return cfg().attn_level >= 2 && DA_XENV("DA_OPT_HID", 0) == 0 && max_nodes >= DA_XENV("DA_ATTN_RES_MIN", 512) && max_nodes <= 19 * 64 &&
           (long long)n_pad * 2 >= (long long)n_graphs * max_nodes;
}
bool attn_res_qsf_shape_ok(int max_nodes, int n_pad, int n_graphs, int kin) {
#ifndef DA_EXPERIMENTS
    return false;          // (the instance exists in the experiments build only)
#endif
    if (!DA_XENV("DA_ATTN_RES_QSF", 0) || (kin != 128 && kin != 256)) return false;
    if (DA_XENV("DA_ATTN_RES_PH", 1) != 1) return false;
    return attn_res_takes(max_nodes, n_pad, n_graphs) && attn_res_qsf_lds(max_nodes, kin) <= (size_t)160 * 1024 &&
           (size_t)((max_nodes + 63) >> 6) * OptK<32, 64>::STAGE >= (size_t)8 * OptK<32, 64>::STAGE + (size_t)2 * (kin >> 4) * 1024;
}

// Fragment-major image of a layer's projection weights for k_attn_res<.., 64>: packed[((h * 4 + m) * KS + s) * 64 + lane] = the 16 bytes
// W[m * hc + 32 h + pi(lane & 31)][16 s + 8 (lane >> 5) .. + 7], m = Q, K, V, skip; pi(a) = a with bits 2 and 3 swapped (MFMA row a of the channel x node
// product then leaves a lane's registers 8 t .. 8 t + 7 holding channels 16 t + 8 half .. + 7: the 16-byte chunk 2 t + half of the node's row)
__global__ void k_pack_w_qs(int heads, int kin, int hc, const bf16_t *__restrict__ W, u32x4 *__restrict__ out) {
    const int KS = kin >> 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)heads * 4 * KS * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), s = (int)((idx >> 6) % KS), hm = (int)((idx >> 6) / KS), m = hm & 3, h = hm >> 2;
    const int a = lane & 31, row = (a & 0x13) | (((a >> 2) & 1) << 3) | (((a >> 3) & 1) << 2);
    out[idx] = *(const u32x4 *)(W + ((size_t)m * hc + 32 * h + row) * kin + 16 * s + 8 * (lane >> 5));
}
size_t w_qs_bytes(int heads, int kin) { return (size_t)heads * 4 * kin * 64; }
int pack_w_qs(int heads, int kin, int hc, const void *wd, void *packed, hipStream_t st) {
    const size_t total = w_qs_bytes(heads, kin) / 16;
    k_pack_w_qs<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(heads, kin, hc, (const bf16_t *)wd, (u32x4 *)packed);
    DA_LAUNCH_CHECK();
    return 0;
}

template <int NWV, bool QUEUE, bool KPF, int XV = 0>
static int launch_res(const AttnDenseParams &p, hipStream_t st) {
    int lds = ((p.max_nodes + 63) >> 6) * OptK<32, 64>::STAGE + 16 + 128 + 256;          // + the queue word + the landing words (XV & 16) + the mask table (XV & 32)
    if (XV & 64) lds = (int)attn_res_qsf_lds(p.max_nodes, (XV & 128) ? 128 : 256);
    static bool attr_done[16] = {};
    int dev = 0;
    DA_CHECK_HIP(hipGetDevice(&dev));
    if (!attr_done[dev & 15]) {
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_res<NWV, QUEUE, KPF, XV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev & 15] = true;
    }
    AttnDenseParams q = p;
    DA_OPB({ const char *e = DA_XENV_LIVE("DA_OPT_PROF_PTR"); q.prof = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; })
    k_attn_res<NWV, QUEUE, KPF, XV><<<p.H * p.n_graphs, 64 * NWV, lds, st>>>(q);
    DA_LAUNCH_CHECK();
    __atomic_fetch_add(&g_res_launches, 1ll, __ATOMIC_RELAXED);
    return 0;
}

// bf16, Q pre-scaled, heads of 32 value channels: C = 32 (hidden layers) or C = 144 with p.fold_out (folded last layer);
// p.mask selects the adjacency-masked instances.  Returns 0 = launched, -1 = shape not covered.
// The PRODUCT build holds one instance per class -- hidden ring kernel, hidden resident kernel, last layer, and their two masked
// forms; the alternatives that lost their whole-graph A/B (numbered DA_OPT_HID / DA_OPT_LAST / DA_ATTN_RES_PH variants, the
// small-graph resident forms, the ablation builds) exist in the EXPERIMENTS build only (da_config.h).
int launch_attn_opt(const AttnDenseParams &p, int C, hipStream_t st) {
    const bool fold = p.fold_out != nullptr, masked = p.mask != nullptr;
    t_virt_taken = false;
    if (masked && p.max_nodes > 4096) return -1;          // the masked walk keeps its key tiles in a 64-bit mask
    if (masked && p.blk_class && p.blk_class_stride > 128) return -1;      // class rows are staged in 128-byte LDS slots
    if (C == 32 && !fold) {
        if (masked) {
#ifdef DA_EXPERIMENTS
            if (DA_XENV("DA_OPT_MASKED_VAR", 0) == 30) return launch_optt<32, false, true, 4, 4, 64, 256>(p, st);
#endif
#ifdef DA_EXPERIMENTS
            // OPT-IN (DA_ATTN_RES_MASKED=1): the K / V-resident form with adjacency masks (round 6, k_attn_res<16, true, true, 32>).  Parity-clean
            // (the whole hybrid suite at 900 pieces), and measured in one process, ten interleaved pairs (profiles/r06/r06_masked_resident_config3_ab.log):
            // configuration 3 at d = 90 -1.2 % (10 of 10), at the scripted d = 539 +7.2 % (0 of 10) -- its one sixteen-wave workgroup per CU leaves
            // no room for the virtual rows' kernel (k_attn_csr_cont_heavy: 16 waves x 128 registers), which the four-wave ring kernel runs beside.
            if (cfg().attn_level >= 2 && DA_XENV("DA_ATTN_RES_MASKED", 0) && p.max_nodes >= 512 && p.max_nodes <= 19 * 64 - 64 &&
                (long long)p.n_pad * 2 >= (long long)p.n_graphs * p.max_nodes)
                return launch_res<16, true, true, 32>(p, st);
#endif
            return launch_optt<32, false, true, 4, 4>(p, st);
        }
        [[maybe_unused]] const int v = DA_XENV("DA_OPT_HID", 0);
        // large complete graphs: the K / V-resident kernel (da_config.attn_level < 2 keeps the ring kernel; 512 = smallest "largest graph"
        // it takes; the Batch's mean slot size must be at least half of its largest graph's -- a ragged Batch of small puzzles with one
        // large one would leave most workgroups of sixteen waves with one or two slabs)
        {
            const int res = cfg().attn_level >= 2, res_min = DA_XENV("DA_ATTN_RES_MIN", 512);
#ifdef DA_EXPERIMENTS
            // OPT-IN (DA_ATTN_RES_SMALL=1; 2 = the 105-VGPR build): measured on 12 x 12 puzzles -- 35.5 -> 32.4 us per 256-puzzle launch alone
            // (58.0 against 63.6 at 512), and configuration 2's step 0.6370 / 0.6315 -> 0.6407 / 0.6370 ms: faster alone, not in the two-branch step
            const int res_small = DA_XENV("DA_ATTN_RES_SMALL", 0);
            if (res && res_small && v == 0 && p.max_nodes > 128 && p.max_nodes <= 160 && (long long)p.n_pad * 2 >= (long long)p.n_graphs * p.max_nodes)
                return res_small == 2 ? launch_res<5, false, true, 8>(p, st)        // (2: 105 VGPRs, no spill, three workgroups per CU)
                                      : launch_res<5, false, true>(p, st);          // 12 x 12 puzzles: five slabs, five waves, four workgroups per CU (96 VGPRs)
#endif
            if (res && v == 0 && p.max_nodes >= res_min && p.max_nodes <= 19 * 64 /* 19 stages + the queue word <= 160 KB */ && (long long)p.n_pad * 2 >= (long long)p.n_graphs * p.max_nodes) {
#ifdef DA_EXPERIMENTS
                // OPT-IN (DA_ATTN_RES_QSF=1): the layer's projection in this kernel's prologue.  Bit-identical to the two-kernel path; measured on the
                // headline (profiles/r06/r06_prologue_projection_*.log): the three hidden layers' kernels 294 -> 287 us per branch-step, the STEP
                // +2.5 % (0.668 -> 0.686 ms) -- in the two-graph loop the projection kernels already run under the other branch's attention, the
                // longer attention kernel does not; with x loads as 1 KB-contiguous instructions (timing probe, DA_QSF_FAKE_FM=1) 268 us and -1.2 %.
                if (p.x) {
                    DA_REQUIRE(attn_res_qsf_shape_ok(p.max_nodes, p.n_pad, p.n_graphs, p.kin) && p.wqs && p.bias_q && p.bias_k && p.bias_v && p.bias_s,
                               "k_attn_res: projection in the prologue requested for a shape it does not take");
                    return p.kin == 256 ? launch_res<16, true, true, 64>(p, st) : launch_res<16, true, true, 64 + 128>(p, st);
                }
#endif
#ifdef DA_EXPERIMENTS
                switch (DA_XENV("DA_ATTN_RES_PH", 1)) {          // A/B switches; default 1
                    case 3: return launch_res<16, false, true>(p, st);         // fixed slabs (wave, wave + 16)
                    case 4: return launch_res<16, true, false>(p, st);         // no K-fragment prefetch
                    case 5: return launch_res<16, true, true, 1>(p, st);       // two PV accumulators
                    case 6: return launch_res<16, true, true, 2>(p, st);       // younger half of the waves at priority 1
                    case 7: return launch_res<16, true, true, 3>(p, st);
                    case 8: return launch_res<16, true, true, 16>(p, st);      // progressive landing (round 6)
                    default: break;
                }
#endif
#ifdef DA_EXPERIMENTS
                // OPT-IN (DA_ATTN_RES_PIPE=1): the software-pipelined key loop (generated asm); 2 .. 6 are its timing ablations (WRONG results)
                switch (DA_XENV("DA_ATTN_RES_PIPE", 0)) {
                    case 1: return launch_res<16, true, true, 256>(p, st);
                    case 2: return launch_res<16, true, true, 256 + 512 * 1>(p, st);
                    case 3: return launch_res<16, true, true, 256 + 512 * 2>(p, st);
                    case 4: return launch_res<16, true, true, 256 + 512 * 3>(p, st);
                    case 5: return launch_res<16, true, true, 256 + 512 * 4>(p, st);
                    case 6: return launch_res<16, true, true, 256 + 512 * 5>(p, st);
                    default: break;
                }
#endif
                return launch_res<16, true, true>(p, st);
            }
        }
#ifdef DA_EXPERIMENTS
        switch (v) {
            case 1: return launch_optt<32, false, false, 4, 4, 64, 1>(p, st);
            case 2: return launch_optt<32, false, false, 4, 4, 64, 2>(p, st);
            case 3: return launch_optt<32, false, false, 6, 4, 32, 0>(p, st);
            case 10: return launch_optt<32, false, false, 4, 4, 64, 64>(p, st);
            case 11: return launch_optt<32, false, false, 5, 4, 64, 64>(p, st);
            case 12: return launch_optt<32, false, false, 6, 4, 32, 64>(p, st);
            case 20: return launch_optt<32, false, false, 4, 4, 64, 128>(p, st);      // skip / residual rows requested before the verification barrier
            case 30: return launch_optt<32, false, false, 4, 4, 64, 256>(p, st);      // DMA issued in the per-lane-address form (round 4's)
            case 40: return launch_optt<32, false, false, 3, 5, 64, 0>(p, st);        // five workgroups per CU on a three-stage ring
            case 50: return launch_optt<32, false, false, 4, 4, 64, 512>(p, st);      // both blocks' K fragments requested at the top of the tile
            case 60: return launch_optt<32, false, false, 4, 4, 64, 1024>(p, st);     // no scheduling barrier between the K reads and the QK chain
            case 70: return launch_optt<32, false, false, 4, 4, 64, 0, 5>(p, st);     // five waves (160 queries) per workgroup
#ifdef DA_ATTN_ABLATE
            case 201: return launch_optt<32, false, false, 4, 4, 64, 2048>(p, st);
            case 202: return launch_optt<32, false, false, 4, 4, 64, 4096>(p, st);
            case 104: return launch_optt<32, false, false, 4, 4, 64, 4>(p, st);
            case 108: return launch_optt<32, false, false, 4, 4, 64, 8>(p, st);
            case 116: return launch_optt<32, false, false, 4, 4, 64, 16>(p, st);
            case 132: return launch_optt<32, false, false, 4, 4, 64, 32>(p, st);
            case 160: return launch_optt<32, false, false, 4, 4, 64, 60>(p, st);
#endif
            default: break;
        }
#endif
        return launch_optt<32, false, false, 4, 4>(p, st);
    }
    // (two ring stages at three workgroups per CU; measured at the end of round 4: three stages at two workgroups per CU 189 - 191 us
    // against 178 in the harness at 64 puzzles, 99 against 92 at 32, the sampling step 0.717 against 0.703 ms -- occupancy, not ring depth)
    if (C == 144 && fold) {
        if (masked) {
#ifdef DA_EXPERIMENTS
            if (DA_XENV("DA_OPT_MASKED_VAR", 0) == 30) return launch_optt<144, true, true, 2, 3, 64, 256>(p, st);
#endif
            return launch_optt<144, true, true, 2, 3>(p, st);
        }
#ifdef DA_EXPERIMENTS
        switch (DA_XENV("DA_OPT_LAST", 0)) {
            case 1: return launch_optt<144, true, false, 4, 3, 32, 0>(p, st);
            case 2: return launch_optt<144, true, false, 3, 3, 32, 0>(p, st);
            case 3: return launch_optt<144, true, false, 3, 4, 32, 0>(p, st);
            case 4: return launch_optt<144, true, false, 2, 3, 64, 1>(p, st);
            case 5: return launch_optt<144, true, false, 4, 3, 32, 1>(p, st);
            case 6: return launch_optt<144, true, false, 2, 3, 64, 2>(p, st);
            case 7: return launch_optt<144, true, false, 4, 3, 32, 2>(p, st);
            case 10: return launch_optt<144, true, false, 4, 3, 32, 64>(p, st);
            case 11: return launch_optt<144, true, false, 3, 2, 64, 64>(p, st);
            case 12: return launch_optt<144, true, false, 3, 3, 32, 64>(p, st);
            case 30: return launch_optt<144, true, false, 2, 3, 64, 256>(p, st);       // DMA issued in the per-lane-address form (round 4's)
            case 60: return launch_optt<144, true, false, 2, 3, 64, 1024>(p, st);      // no scheduling barrier between the K reads and the QK chain
            case 70: return launch_optt<144, true, false, 2, 3, 64, 0, 5>(p, st);      // five waves (160 queries) per workgroup
#ifdef DA_ATTN_ABLATE
            case 201: return launch_optt<144, true, false, 2, 3, 64, 2048>(p, st);
            case 202: return launch_optt<144, true, false, 2, 3, 64, 4096>(p, st);
            case 104: return launch_optt<144, true, false, 2, 3, 64, 4>(p, st);
            case 108: return launch_optt<144, true, false, 2, 3, 64, 8>(p, st);
            case 116: return launch_optt<144, true, false, 2, 3, 64, 16>(p, st);
            case 132: return launch_optt<144, true, false, 2, 3, 64, 32>(p, st);
            case 124: return launch_optt<144, true, false, 2, 3, 64, 24>(p, st);
            case 128: return launch_optt<144, true, false, 2, 3, 64, 28>(p, st);
            case 160: return launch_optt<144, true, false, 2, 3, 64, 60>(p, st);
#endif
            default: break;
        }
#endif
        return launch_optt<144, true, false, 2, 3>(p, st);
    }
    return -1;
}

int attn_opt_counters(unsigned long long *out2, int reset) {
    DA_CHECK_HIP(hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_opt_fallbacks), 2 * sizeof(unsigned long long)));
    if (reset) { const unsigned long long z[2] = {0, 0}; DA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_opt_fallbacks), z, sizeof(z))); }
    return 0;
}

}  // namespace da
