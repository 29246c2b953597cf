// Dense block-diagonal graph attention on the matrix cores: one PyG TransformerConv attention
// (Transformer_GNN.py:32,38) for batches whose graphs are COMPLETE (every piece attends to every
// piece of its puzzle, with or without self loops) -- the fully-connected case of the reference's
// rotation / translation datasets (puzzle_dataset.py:279-289,609-614).
//
//   out[i, h*C:(h+1)*C] = act( sum_j softmax_j(q_i.k_j / sqrt(C)) v_j + skip_i (+ residual_i) )
//
// flash-style: the n x n score matrix never exists; per (graph, head, 128-query tile) workgroup the
// K / V tiles stream through LDS and each of the 4 waves owns 32 queries.
//   S^T = K_tile . Q^T      32 keys x 32 queries per MFMA chain; A = K rows from LDS, B = Q rows held
//                           in registers for the whole kernel.  MFMA row rho of a 32-key block is fed
//                           key pi(rho) = (rho&3) + 4((rho>>3)&3) + 16((rho>>2)&1), so that with the
//                           32x32 accumulator layout lane (q, half) ends up holding the 16
//                           CONSECUTIVE keys 16*half .. 16*half+15 of query q.
//   softmax                 running max / sum are lane-local plus one cross-half exchange; the
//                           accumulator rescale is skipped while the running max grows by < 2^8.
//   O^T += V^T_tile . P^T   P needs NO data movement to become the B operand.  The A operand wants 8
//                           consecutive KEYS of one channel per lane, while V lies row-major ([key][C],
//                           the layout the projection GEMM can write with full cache lines): the
//                           transposition happens on the LDS read, ds_read_b64_tr_b16 (bf16), two per
//                           MFMA, fetched two channel blocks ahead of their MFMAs (a wave may only have
//                           15 LDS operations in flight).
// bf16: v_mfma_f32_32x32x16_bf16; fp32 parity mode: v_mfma_f32_32x32x2_f32 (exact fp32).
// LDS rows are padded: K to an odd number of 16-byte slots (conflict-free ds_read_b128), V to a stride of
// 64 (mod 256) bytes (the 16-lane groups of a transposing read tile the 64 banks exactly).  Global -> LDS
// is LDS-DMA (global_load_lds_dwordx4): the padded image is produced by per-lane SOURCE addresses (pad
// slots re-load a dummy piece), two stages, tile t+1 in flight under the MFMAs of tile t, one barrier per
// tile.  Workgroup ids are remapped so that the query tiles of one (graph, head) run back to back on ONE
// XCD and share its L2 copy of K / V.
#include <stdlib.h>

#include "da_attn_common.h"

namespace da {

// DA_ATTN_PROBE builds keep the ablation switches of tools/attn_probe.py (DA_ATTN_DEBUG env)
#ifdef DA_ATTN_PROBE
#define DA_ATTN_DBG(...) __VA_ARGS__
#define DA_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define DA_ATTN_DBG(...)
#define DA_TICK(var)
#endif

// Fallback bookkeeping of the shift-free softmax paths (tests assert that the branches they aim at really ran;
// da_debug_counters): [1] k_attn_dense waves that left FAST mode ([0] / [2] are filled from da_attn_opt.hip's counters).
// Only touched inside the (rare) fallback branch.
__device__ unsigned long long g_attn_fallbacks[4];

// (two workgroups per CU -- 256 registers per wave -- except the fp32 instances with 144-wide heads AND values (the un-folded last layer of the
// parity mode): their accumulators alone are 144 registers; at 256 they spilled 55 / 134 registers to scratch, at one workgroup per CU none)
template <typename T, int C, bool MASKED, int CV, int NST, int NW>
__global__ __launch_bounds__(64 * NW, (sizeof(T) == 4 && C == 144 && CV == 144) ? 1 : 2) void k_attn_dense(AttnDenseParams p) {
    using CF = Cfg<T, C, CV>;
    // NW waves of 32 queries per workgroup: 4 in production; 5 and 8 exist for the A/B record (see attn_nw)
    constexpr int QT = 32 * NW, NT = 64 * NW, MAXI = (CF::NI + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // NST stages x (K | V)

    // MASKED: the adjacency bits enter as the INITIAL VALUE of the S^T accumulator -- 0 for an edge, -inf for no edge -- so
    // masking costs no VALU pass over the scores (the bit test + select pair per score doubled the softmax's instruction
    // count: 128 vs 54 us per C = 32 layer).  The 16 floats a lane needs for a 32-key block come from a 16-entry table in
    // LDS, one 16-byte read per nibble of its 16-bit mask word: entry i = {bit0 ? 0 : -inf, ..., bit3 ? 0 : -inf}.
    float *mlut = (float *)(smem + NST * CF::STAGE);
    if (MASKED && threadIdx.x < 64) {
        const int e = threadIdx.x >> 2, b = threadIdx.x & 3;
        mlut[threadIdx.x] = ((e >> b) & 1) ? 0.f : -INFINITY;
    }
    // (visible to every wave after the first barrier of the tile loop)

    // XCD-aware remap: hardware places workgroup b on XCD b % 8; give XCD x head x of every graph and
    // walk the query tiles of one (graph, head) consecutively.
    const int bid = blockIdx.x;
    const int h = bid & 7, s_ = bid >> 3;
    const int qt = s_ % p.nqt, g = s_ / p.nqt;
    const int node0 = p.graph_ptr[g], n_g = p.graph_ptr[g + 1] - node0, pad0 = p.pad_ptr[g];
    if (qt * QT >= n_g) return;

    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = qt * QT + wid * 32;
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
    float m = -1e30f, l = 0.f;               // finite reference (see the softmax below); only the slow path moves it
    bool fast = !MASKED && p.fast == 1;     // shift-free mode (wave-uniform, one way out); p.fast == 2: DA_ATTN_FORCE_GEN

    // ---- LDS-DMA plan: instruction q (1 KB) of a tile is issued by wave q % NW; lane -> slot q*64+lane
    const unsigned char *Kg = (const unsigned char *)p.K + ((size_t)h * np + pad0) * CF::ROWB;
    const unsigned char *Vg = (const unsigned char *)p.Vt + ((size_t)h * np + pad0) * CF::ROWBV;
    unsigned soff[MAXI];
#pragma unroll
    for (int x = 0; x < MAXI; ++x) {
        const int q = wid + NW * x;
        unsigned o = 0;
        if (q < CF::NIK) {
            const int s = q * 64 + lane, row = s / CF::KSPR, col = s - row * CF::KSPR;
            if (row < CF::BKEYS && col < CF::KVALID) o = (unsigned)(row * CF::ROWB + col * 16);
        } else {
            const int s = (q - CF::NIK) * 64 + lane, row = s / CF::VSPR, col = s - row * CF::VSPR;
            if (row < CF::BKEYS && col < CF::KVALIDV) o = (unsigned)(row * CF::ROWBV + col * 16);
        }
        soff[x] = o;
    }
    auto issue = [&](int kt, int stage) {
        unsigned char *sb = smem + stage * CF::STAGE;
        const unsigned char *kb_ = Kg + (size_t)kt * CF::BKEYS * CF::ROWB;
        const unsigned char *vb_ = Vg + (size_t)kt * CF::BKEYS * CF::ROWBV;
#pragma unroll
        for (int x = 0; x < MAXI; ++x) {
            const int q = wid + NW * x;
            if (NW * x + NW - 1 < CF::NI || q < CF::NI) {            // compile-time true except for a partial last round
                const unsigned char *src = (q < CF::NIK ? kb_ : vb_) + soff[x];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sb + q * 1024), 16, 0, 0);
            }
        }
    };

    const int myn = (CF::NI - wid + NW - 1) / NW;             // q = wid, wid + NW, ... < NI
    int nkt = (n_g + CF::BKEYS - 1) / CF::BKEYS;
    DA_ATTN_DBG(if (p.debug & 32) nkt = 1;)
    const int qidx = q0 + i;                     // this lane's query (index inside the graph)
    const int pi_i = (i & 3) + 4 * ((i >> 3) & 3) + 16 * ((i >> 2) & 1);     // key fed to MFMA row i
    // MASKED: this lane's row of the adjacency bit matrix (bit j = edge j -> this query)
    const unsigned char *mrow = nullptr;
    if (MASKED) mrow = p.mask + p.mask_ptr[g] + (size_t)min(qidx, n_g - 1) * (size_t)((p.pad_ptr[g + 1] - pad0) >> 3);
    // MASKED: slot -> node (banded expander plans order a graph's slots by position; see k_attn_optt in da_attn_opt.hip)
    auto node_of = [&](int ql) { return (MASKED && p.slot_node) ? p.slot_node[pad0 + ql] : node0 + ql; };
    // MASKED: remainder-edge metadata of the four queries this 8-lane group finishes in the epilogue
    int rm_beg[4] = {0, 0, 0, 0}, rm_end[4] = {0, 0, 0, 0}, rm_slot[4] = {0, 0, 0, 0};
    if (MASKED && wave_on) {
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
    }
    // LDS byte offsets of this lane's fragments inside a stage (constant over the whole kernel)
    const int koff = pi_i * CF::RS + half * 16;
    // V fragment base inside a stage.  bf16 (transposing reads): key 16*half + (li >> 2), channel
    // 16*((lane >> 4) & 1) + 4*(li & 3) with li = lane & 15; fp32: key 0, channel i.
    const int li = lane & 15;
    const int vbase = CF::ES == 2 ? CF::KBYTES + (16 * half + (li >> 2)) * CF::RSV + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2
                                  : CF::KBYTES + i * 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned mw_nxt[CF::KB];
#pragma unroll
    for (int kb = 0; kb < CF::KB; ++kb) mw_nxt[kb] = 0;
    if (MASKED && wave_on) {
#pragma unroll
        for (int kb = 0; kb < CF::KB; ++kb) mw_nxt[kb] = *(const unsigned short *)(mrow + ((kb * 32 + 16 * half) >> 3));
    }
    DA_ATTN_DBG(unsigned long long c_bar = 0, c_iss = 0, c_qk = 0, c_sm = 0, c_pv = 0;)
    DA_TICK(t_start);
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nkt) issue(st, st);
    for (int kt = 0; kt < nkt; ++kt) {
        DA_TICK(t0_);
        // tile kt must have landed; the younger tiles already issued (at most NST - 2 of them) may stay in flight.  Loads
        // retire in order, so "at most k * nmine of my VMEM operations outstanding" implies tile kt is in LDS; anything
        // else this wave has pending (MASKED: adjacency words) only makes the wait conservative.
        {
            // wave `wid` issues `myn` DMA instructions per tile, so younger * myn outstanding operations is exact
            wait_vmcnt(min(nkt - 1 - kt, NST - 2) * myn);
            // raw barrier: __syncthreads() would drain the DMA still in flight (its fence carries vmcnt(0))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // everyone's share landed + everyone left the slot refilled below
        }
        DA_TICK(t1_);
        if (kt + NST - 1 < nkt DA_ATTN_DBG(&& !(p.debug & 1))) issue(kt + NST - 1, (kt + NST - 1) % NST);
        DA_TICK(t2_);
        DA_ATTN_DBG(c_bar += t1_ - t0_; c_iss += t2_ - t1_;)
        if (!wave_on) continue;
        const unsigned char *stg = smem + (kt % NST) * CF::STAGE;
        // MASKED: adjacency words of THIS tile were requested during the previous one; request the next tile's now
        unsigned mw_cur[CF::KB];
        if (MASKED) {
#pragma unroll
            for (int kb = 0; kb < CF::KB; ++kb) mw_cur[kb] = mw_nxt[kb];
            if (kt + 1 < nkt) {
#pragma unroll
                for (int kb = 0; kb < CF::KB; ++kb)
                    mw_nxt[kb] = *(const unsigned short *)(mrow + (((kt + 1) * CF::BKEYS + kb * 32 + 16 * half) >> 3));
            }
        }
#pragma unroll
        for (int kb = 0; kb < CF::KB; ++kb) {
            const int key0 = kt * CF::BKEYS + kb * 32;
            if (key0 >= n_g) break;
            DA_TICK(tb_);
            // ---- all LDS fragment reads of this 32-key block are issued up front (K for QK^T now, V
            // for PV later): the compiler otherwise pairs every two reads with a full lgkmcnt(0) wait
            // and the MFMA chain idles ~100 cycles per pair
            const unsigned mw = MASKED ? mw_cur[kb] : 0u;        // fetched one tile ahead (see the top of the loop)
            u32x4 kf[CF::NCH];
#pragma unroll
            for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(stg + koff + kb * 32 * CF::RS + ch * 32);
            __builtin_amdgcn_sched_barrier(0);       // keep the K read batch ahead of the MFMA chain
            f32x16 s;
            if (MASKED) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 b4 = *(const f32x4 *)((const unsigned char *)mlut + (((mw >> (4 * j)) & 15u) << 4));
                    s[4 * j] = b4[0]; s[4 * j + 1] = b4[1]; s[4 * j + 2] = b4[2]; s[4 * j + 3] = b4[3];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
            }
            DA_ATTN_DBG(if (!(p.debug & 8)))
            {
#pragma unroll
                for (int ch = 0; ch < CF::NCH; ++ch) s = mma_chunk(T(), kf[ch], qf[ch], s);
            }
            DA_ATTN_DBG(asm volatile("" :: "v"(s[0]), "v"(s[15]));)
            DA_TICK(t3_);
            // V fragments of the first channel blocks: issued behind the QK^T chain, they land under the softmax
            u32x2 vlo[CF::ES == 2 ? CF::NCB : 1][2], vhi[CF::ES == 2 ? CF::NCB : 1][2];
            const unsigned vb = lds0 + (unsigned)((kt % NST) * CF::STAGE + vbase + kb * 32 * CF::RSV);
            if (CF::ES == 2) {
                // only the first two channel blocks now; the rest are fetched two blocks ahead inside the PV
                // loop: a wave may have 15 LDS operations in flight (lgkmcnt is 4 bits) and the 9 K reads +
                // 20 V reads of a C = 144 block stalled the issue stream when they were all queued here
#pragma unroll
                for (int cb = 0; cb < (CF::NCB < 2 ? CF::NCB : 2); ++cb) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        vlo[cb][mm] = tr_read(vb, cb * 64 + (8 * mm) * CF::RSV);
                        vhi[cb][mm] = tr_read(vb, cb * 64 + (8 * mm + 4) * CF::RSV);
                    }
                }
            }
            // this lane now holds keys key0 + 16*half + r, r = 0..15, of query qidx.
            // mask padded keys (last tile) and the diagonal (graphs without self loops)
            const int kbase = key0 + 16 * half;
            if (MASKED) {                       // non-edges already sit at -inf (accumulator initial value; bits beyond n_g are 0)
            } else {
                const bool tail = key0 + 32 > n_g;
                const bool diag = p.nodiag && key0 < q0 + 32 && key0 + 32 > q0;
                if (tail || diag) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kbase + r >= n_g || (p.nodiag && kbase + r == qidx)) s[r] = -INFINITY;
                }
            }
            float pr[16];
            DA_ATTN_DBG(if (p.debug & 2) { _Pragma("unroll") for (int r = 0; r < 16; ++r) pr[r] = s[r]; } else)
            {
                // Online softmax WITHOUT a per-block max: p = exp2((s - m) sc) is formed against the running reference m
                // directly; only when a block's sum says the reference is stale -- first block: m = -1e30 gives +inf;
                // later: some score more than ~2^10 above m -- is the block's true max taken (lane-local v_max3 tree + one
                // cross-half exchange) and O / l rescaled.  Softmax is shift invariant, so any m works as long as nothing
                // overflows; p <= 2^14 keeps every accumulator far from the fp32 range.  The kernel is bound by instruction
                // issue (rocprof: SQ_ACTIVE_INST_ANY summed over the 4 waves of a SIMD > its cycles), so the 11-instruction
                // max tree leaves the common path and the scale / sum run as packed-f32 pairs (v_pk_fma_f32, v_pk_add_f32).
                typedef __attribute__((ext_vector_type(2))) float f32x2;
                const f32x2 sc2 = {p.sc, p.sc};
                f32x2 e2[8];
                auto exp_block = [&](float ms_) {
                    const f32x2 nm = {-ms_, -ms_};
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const f32x2 t = (f32x2){s[2 * r], s[2 * r + 1]} * sc2 + nm;
                        e2[r] = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    }
                    const f32x2 a = (e2[0] + e2[1]) + (e2[2] + e2[3]), b = (e2[4] + e2[5]) + (e2[6] + e2[7]);
                    const f32x2 c = a + b;
                    return c[0] + c[1];
                };
                // FAST mode (complete graphs, Q pre-scaled so that s is in log2 units): p = exp2(s) with NO reference at all.
                // The shift only guards the exponent range, and fp32 / bf16 carry 8 exponent bits: while the running sum
                // stays inside [2^-60, 2^60] nothing overflows or vanishes.  The block sum -- needed anyway -- is the test;
                // the first block that leaves the range sends the wave to the referenced path below for good (its scores
                // are still intact; what was accumulated so far is relative to the reference 0).
                float bsum;
                bool redo = !fast;
                if (fast) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) e2[r] = (f32x2){__builtin_amdgcn_exp2f(s[2 * r]), __builtin_amdgcn_exp2f(s[2 * r + 1])};
                    const f32x2 a = (e2[0] + e2[1]) + (e2[2] + e2[3]), b = (e2[4] + e2[5]) + (e2[6] + e2[7]);
                    const f32x2 c = a + b;
                    bsum = c[0] + c[1];
                    if (__any(!(bsum < 1.152921504606847e18f) || !(l + bsum > 8.673617379884035e-19f))) {
                        fast = false;
                        redo = true;
                        // what was accumulated so far is relative to the reference 0: move it to the reference
                        // e = floor(log2(row sum)) (exact scaling), i.e. sum in [1, 2); nothing accumulated: start afresh
                        const float lq = l + __shfl_xor(l, 32);
                        const float e_ = lq > 0.f ? pow2_floor_exp(lq) : 0.f;
                        const float c_ = __builtin_amdgcn_exp2f(-e_);
                        m = lq > 0.f ? e_ : -1e30f;                     // (FAST mode implies sc == 1: m is in the scores' units)
                        l *= c_;
#pragma unroll
                        for (int cb = 0; cb < CF::NCB; ++cb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) O[cb][r] *= c_;
                        if (lane == 0) atomicAdd(&g_attn_fallbacks[1], 1ull);
                    }
                }
                if (redo) bsum = exp_block(m * p.sc);
                if (redo && __any(!(bsum < 16384.0f))) {
                    const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]);
                    const float a2 = fmaxf(fmaxf(s[6], s[7]), s[8]), a3 = fmaxf(fmaxf(s[9], s[10]), s[11]);
                    const float a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
                    const float mloc = fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), s[15]));
                    const float mnew = fmaxf(m, fmaxf(mloc, __shfl_xor(mloc, 32)));        // >= -1e30: finite
                    const float corr = __builtin_amdgcn_exp2f((m - mnew) * p.sc);
                    m = mnew;
                    l *= corr;
#pragma unroll
                    for (int cb = 0; cb < CF::NCB; ++cb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) O[cb][r] *= corr;
                    bsum = exp_block(m * p.sc);
                }
                l += bsum;
#pragma unroll
                for (int r = 0; r < 8; ++r) { pr[2 * r] = e2[r][0]; pr[2 * r + 1] = e2[r][1]; }
            }
            DA_ATTN_DBG(asm volatile("" :: "v"(pr[0]), "v"(pr[15]));)
            DA_TICK(t4_);
            DA_ATTN_DBG(if (!(p.debug & 4)))
            {
                if (CF::ES == 2) {
                    bf16x8 pf0, pf1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pf0[e] = (__bf16)pr[e]; pf1[e] = (__bf16)pr[8 + e]; }
#pragma unroll
                    for (int cb = 0; cb < CF::NCB; ++cb) {
                        if (cb + 2 < CF::NCB) {
#pragma unroll
                            for (int mm = 0; mm < 2; ++mm) {
                                vlo[cb + 2][mm] = tr_read(vb, (cb + 2) * 64 + (8 * mm) * CF::RSV);
                                vhi[cb + 2][mm] = tr_read(vb, (cb + 2) * 64 + (8 * mm + 4) * CF::RSV);
                            }
                        }
                        // fence of the asm reads of block cb: LDS returns in order, so "at most N newer
                        // operations outstanding" (N = 4 reads per block still in flight behind it) is enough
                        if (cb + 2 < CF::NCB)
                            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vlo[cb][0]), "+v"(vhi[cb][0]), "+v"(vlo[cb][1]), "+v"(vhi[cb][1]));
                        else if (cb + 1 < CF::NCB)
                            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vlo[cb][0]), "+v"(vhi[cb][0]), "+v"(vlo[cb][1]), "+v"(vhi[cb][1]));
                        else
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[cb][0]), "+v"(vhi[cb][0]), "+v"(vlo[cb][1]), "+v"(vhi[cb][1]));
                        const u32x4 v0 = {vlo[cb][0][0], vlo[cb][0][1], vhi[cb][0][0], vhi[cb][0][1]};
                        const u32x4 v1 = {vlo[cb][1][0], vlo[cb][1][1], vhi[cb][1][0], vhi[cb][1][1]};
                        O[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf0, O[cb], 0, 0, 0);
                        O[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, O[cb], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int cb = 0; cb < CF::NCB; ++cb)
                        O[cb] = mma_pv_f32<CF::RSV>(stg + vbase + cb * 128 + kb * 32 * CF::RSV, half, pr, O[cb]);
                }
            }
            DA_ATTN_DBG(asm volatile("" :: "v"(O[0][0]), "v"(O[CF::NCB - 1][15]));)
            DA_TICK(t5_);
            DA_ATTN_DBG(c_qk += t3_ - tb_; c_sm += t4_ - t3_; c_pv += t5_ - t4_;)
        }
    }
    DA_TICK(t_loop_end);
    // ---- epilogue: normalise (PyG: sum + 1e-16) and stage O through LDS as [query][c] fp32, then all
    // 256 threads stream whole output rows: 16-byte coalesced reads of skip (+ residual), activation,
    // 16-byte coalesced stores.  (Written straight from the accumulator layout every access is an
    // 8-byte piece in one of 32 different rows: that cost 25 % of the kernel.)
    // PyG normalises by (sum exp(a - max) + 1e-16), a sum >= 1 where the epsilon is below fp32 resolution.  A wave still
    // in FAST mode holds the UN-SHIFTED sum (any size inside [2^-60, 2^65]): there the epsilon must not be added (it would
    // be 1e-16 at the wrong scale: rows whose logits all sit near -38 nat came out 2e-3 low, round-3 verdict); the referenced
    // path keeps its sum >= 1 and PyG's formula.
    const float lt = l + __shfl_xor(l, 32);
    const float eps_l = fast ? 0.f : 1e-16f;
    const float inv = MASKED ? 1.0f : (lt > 0.f ? 1.0f / (lt + eps_l) : 0.f);
    if (CV != C && !MASKED) {
        // folded value heads: the caller projected V with the next linear layer's weight block of this head
        // (softmax(QK^T) (V W^T) == (softmax(QK^T) V) W^T), so the output is CV wide and goes out normalised,
        // per head, for the tail kernel to sum over heads -- no skip, no activation, no LDS staging
        static_assert(CV == C || CF::NCB == 1, "folded value heads are one 32-channel block");
        if (wave_on && qidx < n_g) {
            const float invf = inv;
            T *dst = (T *)p.fold_out + ((size_t)h * p.n_rows + node0 + qidx) * CV;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int c0 = 8 * jj + 4 * half;
                if (c0 < CV) {
                    const float v4[4] = {O[0][4 * jj] * invf, O[0][4 * jj + 1] * invf, O[0][4 * jj + 2] * invf, O[0][4 * jj + 3] * invf};
                    st4(dst + c0, v4);
                }
            }
        }
        return;
    }
    constexpr int CO = CV;                                        // width of the staged rows (= C unless the value heads are folded)
    constexpr int RSOF = CO + 4;                                  // floats per staged row (16-B aligned, odd # of 16-B slots)
    static_assert(QT * RSOF * 4 <= NST * CF::STAGE, "O staging must fit in the K/V ring");
    float *so = (float *)smem;
    dma_barrier();                                              // ring no longer read by anyone
    DA_ATTN_DBG(if (p.debug & 16) return;)
    if (wave_on) {
        float *orow = so + (wid * 32 + i) * RSOF;
        if (MASKED && half == 0) {            // partial softmax state rides in the row's 4 spare floats
            orow[CO] = (m > -1e29f) ? m * (p.sc * 0.6931471805599453f) : 0.f;
            orow[CO + 1] = lt;
        }
#pragma unroll
        for (int cb = 0; cb < CF::NCB; ++cb) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int c0 = cb * 32 + 8 * jj + 4 * half;
                if (c0 >= CO) continue;
                *(f32x4 *)(orow + c0) = (f32x4){O[cb][4 * jj] * inv, O[cb][4 * jj + 1] * inv, O[cb][4 * jj + 2] * inv,
                                               O[cb][4 * jj + 3] * inv};
            }
        }
    }
    dma_barrier();
    constexpr int EPC = 16 / CF::ES, CPR = C / EPC;              // elements per 16-B chunk, chunks per row
    const int nq = min(QT, n_g - qt * QT);                      // valid queries of this tile
    if (MASKED) {
        // Remainder edges of this tile's queries (hybrid mode): the rows staged above hold the UN-normalised
        // sum of the masked attention with (max, sum) in their spare floats; each query's few remaining
        // incoming edges -- from virtual nodes, duplicated pairs -- continue the same online softmax right
        // here, 8 lanes per query, before the rows are normalised below.  A lane owns the 16-byte chunks
        // sub, sub + 8, ... of the C-wide rows (vector loads); the CSR metadata of the four queries a lane
        // group serves (rm_*) was fetched at kernel start, and the Q / K / V rows of all four first edges are
        // requested before any is consumed: the pass used to be a chain of ~5 dependent global loads per query.
        remainder_edges<T, CF, CO, RSOF>(p, so, h, np, pad0, n_g, qt * QT, wid, lane, wave_on, rm_beg, rm_end, rm_slot);
        __syncthreads();
    }
    if (CV != C) {                // MASKED + folded value heads: normalised per-head rows for the tail kernel
        constexpr int CQ = CV / 4;
        for (int it = tid; it < nq * CQ; it += NT) {
            const int q = it / CQ, ch = it - q * CQ;
            const float lr = so[q * RSOF + CO + 1];
            const float ir = lr > 0.f ? 1.0f / (lr + 1e-16f) : 0.f;
            const f32x4 a = *(const f32x4 *)(so + q * RSOF + ch * 4);
            const float v4[4] = {a[0] * ir, a[1] * ir, a[2] * ir, a[3] * ir};
            st4((T *)p.fold_out + ((size_t)h * p.n_rows + node_of(qt * QT + q)) * CV + ch * 4, v4);
        }
        return;
    }
    // batches of NB chunks per thread: all skip / residual loads of a batch are in flight before the
    // first one is consumed (a rolled load -> add -> store loop pays one L2/HBM latency per chunk)
    constexpr int NB = 3;
    for (int it0 = tid; it0 < nq * CPR; it0 += NT * NB) {
        u32x4 skv[NB], rsv[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int it = it0 + NT * k;
            if (it < nq * CPR) {
                const int q = it / CPR, ch = it - q * CPR;
                const size_t off = (size_t)node_of(qt * QT + q) * HC + (size_t)h * C + ch * EPC;
                skv[k] = *(const u32x4 *)((const T *)p.S + off);
                if (p.res) rsv[k] = *(const u32x4 *)((const T *)p.res + off);
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int it = it0 + NT * k;
            if (it < nq * CPR) {
                const int q = it / CPR, ch = it - q * CPR;
                const size_t off = (size_t)node_of(qt * QT + q) * HC + (size_t)h * C + ch * EPC;
                const float *src = so + q * RSOF + ch * EPC;
                float v[EPC], sk[EPC];
                {
                    const f32x4 a = *(const f32x4 *)src;
                    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
                    if (EPC == 8) { const f32x4 b2 = *(const f32x4 *)(src + 4); v[4] = b2[0]; v[5] = b2[1]; v[6] = b2[2]; v[7] = b2[3]; }
                }
                if (MASKED) {                                     // rows were staged un-normalised
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
    DA_ATTN_DBG(if (p.prof && tid == 0) { DA_TICK(t_end_); unsigned long long *o = p.prof + 8 * blockIdx.x; o[0] = t_end_ - t_start; o[1] = c_bar; o[2] = c_iss; o[3] = c_qk; o[4] = c_sm; o[5] = c_pv; o[6] = t_end_ - t_loop_end; o[7] = 1; })
}

// ---------------------------------------------------------------------------------------------------------------------
// k_attn_dense2: the C = 32 bf16 instance on complete graphs (the three hidden layers of the 2D arch = the class with the
// largest share of the sampling step) with TWO 32-query slabs per wave.  Why: SQ counters and the ablation runs of
// tools/attn_ablate.py say the one-slab kernel is bound by the instruction stream as a whole -- per 32 keys x 32 queries a
// wave issues ~45 VALU + 4 MFMA next to ~45 "skeleton" instructions (K / V fragment reads, DMA issue, counted waits, barrier,
// tail / diagonal tests, loop control), and the parts add up instead of overlapping across the 4 resident waves of a SIMD
// (tools/overlap_probe.hip: even a clean MFMA -> VALU -> MFMA chain only overlaps ~35 %).  With two slabs every K / V
// fragment read from LDS feeds the MFMAs of 64 queries, the skeleton is paid once per 2 x 1024 scores, a workgroup
// (4 waves, 256 queries) streams each K / V tile once for twice the queries, and inside a wave the matrix work of one slab
// sits next to the softmax of the other.  MEASURED: no faster (168 VGPRs = 3 waves per SIMD; see attn2_env below) -- the
// skeleton was not what bounds the kernel.  Slabs w and w + 4 of the workgroup's eight belong to wave w, so the last tile of
// a 900-piece puzzle (5 slabs) still gives every wave work (2, 1, 1, 1); one-slab waves run the NS = 1 body.
// Same LDS image, DMA ring, fragment layouts, softmax and epilogue arithmetic as k_attn_dense<bf16_t, 32, false, 32, 4>:
// results are bit-identical to it (tests/test_gpu_parity.py::test_attn_dense2_matches_one_slab_kernel).
#ifdef DA_EXPERIMENTS          // (k_attn_dense2 .. launch_attn_dense2: the experiments build only)
template <int NS> struct SlabTag { static constexpr int value = NS; };

__global__ __launch_bounds__(256, 2) void k_attn_dense2(AttnDenseParams p) {
    using T = bf16_t;
    constexpr int C = 32, NST = 4, QT = 256;
    using CF = Cfg<T, C, C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int bid = blockIdx.x;
    const int h = bid & 7, s_ = bid >> 3;
    const int qt = s_ % p.nqt, g = s_ / p.nqt;                       // p.nqt counts 256-query tiles here
    const int node0 = p.graph_ptr[g], n_g = p.graph_ptr[g + 1] - node0, pad0 = p.pad_ptr[g];
    if (qt * QT >= n_g) return;

    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nslab = min(8, (n_g - qt * QT + 31) >> 5);             // 32-query slabs of this workgroup
    const bool wave_on = wid < nslab, two = wid + 4 < nslab;
    const int HC = p.H * C;
    const size_t np = (size_t)p.n_pad;

    // Q fragments of both slabs stay in registers; rows beyond the graph are zeroed (their scores stay finite)
    int q0[2], qidx[2];
    u32x4 qf[2][CF::NCH];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        q0[sl] = qt * QT + (wid + 4 * sl) * 32;
        qidx[sl] = q0[sl] + i;
        const unsigned char *qrow = (const unsigned char *)p.Q + ((size_t)h * np + pad0 + min(q0[sl], n_g - 1) / 32 * 32 + i) * CF::ROWB;
#pragma unroll
        for (int ch = 0; ch < CF::NCH; ++ch) {
            qf[sl][ch] = *(const u32x4 *)(qrow + ch * 32 + half * 16);
            if (qidx[sl] >= n_g) qf[sl][ch] = (u32x4){0u, 0u, 0u, 0u};
        }
    }

    // ---- LDS-DMA plan (as in k_attn_dense): instruction q (1 KB) of a tile is issued by wave q % 4
    const unsigned char *Kg = (const unsigned char *)p.K + ((size_t)h * np + pad0) * CF::ROWB;
    const unsigned char *Vg = (const unsigned char *)p.Vt + ((size_t)h * np + pad0) * CF::ROWBV;
    unsigned soff[CF::MAXI];
#pragma unroll
    for (int x = 0; x < CF::MAXI; ++x) {
        const int q = wid + 4 * x;
        unsigned o = 0;
        if (q < CF::NIK) {
            const int s = q * 64 + lane, row = s / CF::KSPR, col = s - row * CF::KSPR;
            if (row < CF::BKEYS && col < CF::KVALID) o = (unsigned)(row * CF::ROWB + col * 16);
        } else {
            const int s = (q - CF::NIK) * 64 + lane, row = s / CF::VSPR, col = s - row * CF::VSPR;
            if (row < CF::BKEYS && col < CF::KVALIDV) o = (unsigned)(row * CF::ROWBV + col * 16);
        }
        soff[x] = o;
    }
    auto issue = [&](int kt, int stage) {
        unsigned char *sb = smem + stage * CF::STAGE;
        const unsigned char *kb_ = Kg + (size_t)kt * CF::BKEYS * CF::ROWB;
        const unsigned char *vb_ = Vg + (size_t)kt * CF::BKEYS * CF::ROWBV;
#pragma unroll
        for (int x = 0; x < CF::MAXI; ++x) {
            const int q = wid + 4 * x;
            if (4 * x + 3 < CF::NI || q < CF::NI) {
                const unsigned char *src = (q < CF::NIK ? kb_ : vb_) + soff[x];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sb + q * 1024), 16, 0, 0);
            }
        }
    };

    const int nkt = (n_g + CF::BKEYS - 1) / CF::BKEYS;
    const int pi_i = (i & 3) + 4 * ((i >> 3) & 3) + 16 * ((i >> 2) & 1);     // key fed to MFMA row i
    const int koff = pi_i * CF::RS + half * 16;
    const int li = lane & 15;
    const int vbase = CF::KBYTES + (16 * half + (li >> 2)) * CF::RSV + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

    constexpr int RSOF = C + 4;                                  // floats per staged output row
    static_assert(QT * RSOF * 4 <= NST * CF::STAGE, "O staging must fit in the K/V ring");
    float *so = (float *)smem;

    auto run = [&](auto tag) {
        constexpr int NS = decltype(tag)::value;
        f32x16 O[NS];
        float m[NS], l[NS];
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) O[sl][r] = 0.f;
            m[sl] = -1e30f; l[sl] = 0.f;
        }
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nkt) issue(st, st);
        for (int kt = 0; kt < nkt; ++kt) {
            {
                constexpr int PERW = CF::NI / 4;
                const int younger = min(nkt - 1 - kt, NST - 2);
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PERW) : "memory");
                else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PERW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (kt + NST - 1 < nkt) issue(kt + NST - 1, (kt + NST - 1) % NST);
            if (!wave_on) continue;
            const unsigned char *stg = smem + (kt % NST) * CF::STAGE;
#pragma unroll
            for (int kb = 0; kb < CF::KB; ++kb) {
                const int key0 = kt * CF::BKEYS + kb * 32;
                if (key0 >= n_g) break;
                u32x4 kf[CF::NCH];
#pragma unroll
                for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(stg + koff + kb * 32 * CF::RS + ch * 32);
                __builtin_amdgcn_sched_barrier(0);
                f32x16 s[NS];
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[sl][r] = 0.f;
#pragma unroll
                    for (int ch = 0; ch < CF::NCH; ++ch) s[sl] = mma_chunk(T(), kf[ch], qf[sl][ch], s[sl]);
                }
                // V fragments of this block (shared by the slabs): issued behind the QK^T chains, they land under the softmax
                u32x2 vlo[2], vhi[2];
                const unsigned vb = lds0 + (unsigned)((kt % NST) * CF::STAGE + vbase + kb * 32 * CF::RSV);
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    vlo[mm] = tr_read(vb, (8 * mm) * CF::RSV);
                    vhi[mm] = tr_read(vb, (8 * mm + 4) * CF::RSV);
                }
                const int kbase = key0 + 16 * half;
                const bool tail = key0 + 32 > n_g;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const bool diag = p.nodiag && key0 < q0[sl] + 32 && key0 + 32 > q0[sl];
                    if (tail || diag) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kbase + r >= n_g || (p.nodiag && kbase + r == qidx[sl])) s[sl][r] = -INFINITY;
                    }
                    // online softmax against the running reference m (see k_attn_dense): no per-block max on the common path
                    typedef __attribute__((ext_vector_type(2))) float f32x2;
                    const f32x2 sc2 = {p.sc, p.sc};
                    f32x2 e2[8];
                    auto exp_block = [&](float ms_) {
                        const f32x2 nm = {-ms_, -ms_};
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const f32x2 t = (f32x2){s[sl][2 * r], s[sl][2 * r + 1]} * sc2 + nm;
                            e2[r] = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                        }
                        const f32x2 a = (e2[0] + e2[1]) + (e2[2] + e2[3]), b = (e2[4] + e2[5]) + (e2[6] + e2[7]);
                        const f32x2 c = a + b;
                        return c[0] + c[1];
                    };
                    float bsum = exp_block(m[sl] * p.sc);
                    if (__any(!(bsum < 16384.0f))) {
                        const float a0 = fmaxf(fmaxf(s[sl][0], s[sl][1]), s[sl][2]), a1 = fmaxf(fmaxf(s[sl][3], s[sl][4]), s[sl][5]);
                        const float a2 = fmaxf(fmaxf(s[sl][6], s[sl][7]), s[sl][8]), a3 = fmaxf(fmaxf(s[sl][9], s[sl][10]), s[sl][11]);
                        const float a4 = fmaxf(fmaxf(s[sl][12], s[sl][13]), s[sl][14]);
                        const float mloc = fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), s[sl][15]));
                        const float mnew = fmaxf(m[sl], fmaxf(mloc, __shfl_xor(mloc, 32)));
                        const float corr = __builtin_amdgcn_exp2f((m[sl] - mnew) * p.sc);
                        m[sl] = mnew;
                        l[sl] *= corr;
#pragma unroll
                        for (int r = 0; r < 16; ++r) O[sl][r] *= corr;
                        bsum = exp_block(m[sl] * p.sc);
                    }
                    l[sl] += bsum;
                    bf16x8 pf0, pf1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pf0[2 * e] = (__bf16)e2[e][0]; pf0[2 * e + 1] = (__bf16)e2[e][1];
                        pf1[2 * e] = (__bf16)e2[4 + e][0]; pf1[2 * e + 1] = (__bf16)e2[4 + e][1];
                    }
                    if (sl == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]));
                    const u32x4 v0 = {vlo[0][0], vlo[0][1], vhi[0][0], vhi[0][1]};
                    const u32x4 v1 = {vlo[1][0], vlo[1][1], vhi[1][0], vhi[1][1]};
                    O[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf0, O[sl], 0, 0, 0);
                    O[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, O[sl], 0, 0, 0);
                    if (sl + 1 < NS) __builtin_amdgcn_sched_barrier(0);      // slab 0's PV is in the matrix pipe under slab 1's softmax
                }
            }
        }
        // ---- normalise (PyG: sum + 1e-16) and stage O as [query][c] fp32 rows
        dma_barrier();                                              // ring no longer read by anyone
        if (wave_on) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const float lt = l[sl] + __shfl_xor(l[sl], 32);
                const float inv = lt > 0.f ? 1.0f / (lt + 1e-16f) : 0.f;
                float *orow = so + ((wid + 4 * sl) * 32 + i) * RSOF;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int c0 = 8 * jj + 4 * half;
                    *(f32x4 *)(orow + c0) = (f32x4){O[sl][4 * jj] * inv, O[sl][4 * jj + 1] * inv, O[sl][4 * jj + 2] * inv, O[sl][4 * jj + 3] * inv};
                }
            }
        }
    };
    if (two) run(SlabTag<2>());
    else run(SlabTag<1>());
    dma_barrier();

    // ---- all 256 threads stream whole output rows: + skip (+ residual), activation, 16-byte coalesced stores
    constexpr int EPC = 8, CPR = C / EPC;
    const int nq = min(QT, n_g - qt * QT);
    constexpr int NB = 3;
    for (int it0 = tid; it0 < nq * CPR; it0 += 256 * NB) {
        u32x4 skv[NB], rsv[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int it = it0 + 256 * k;
            if (it < nq * CPR) {
                const int q = it / CPR, ch = it - q * CPR;
                const size_t off = ((size_t)node0 + qt * QT + q) * HC + (size_t)h * C + ch * EPC;
                skv[k] = *(const u32x4 *)((const T *)p.S + off);
                if (p.res) rsv[k] = *(const u32x4 *)((const T *)p.res + off);
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int it = it0 + 256 * k;
            if (it < nq * CPR) {
                const int q = it / CPR, ch = it - q * CPR;
                const size_t off = ((size_t)node0 + qt * QT + q) * HC + (size_t)h * C + ch * EPC;
                const float *src = so + q * RSOF + ch * EPC;
                float v[EPC], sk[EPC];
                const f32x4 a = *(const f32x4 *)src, b2 = *(const f32x4 *)(src + 4);
                v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b2[0]; v[5] = b2[1]; v[6] = b2[2]; v[7] = b2[3];
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
}

#endif      // DA_EXPERIMENTS (k_attn_dense2)

// da_config.attn_level = 0 (DA_ATTN_LEVEL=0): every layer on k_attn_dense (the general kernel: any graph, both precisions) -- the
// tests' way to cover it at shapes the optimistic kernels would take.  DA_ATTN_OPT_LAST / _MASKED (experiments build): per class.
static int attn_opt_last_env() { return DA_XENV("DA_ATTN_OPT_LAST", 1); }
static int attn_opt_masked_env() { return DA_XENV("DA_ATTN_OPT_MASKED", 1); }
static int attn_opt_env() { return cfg().attn_level >= 1; }
#ifdef DA_EXPERIMENTS
// opt-in (DA_ATTN2=1): measured EQUAL to the one-slab kernel at 64 puzzles (173 vs 174 us per conv) and slower at 32
// (95.5 vs 89.1) -- kept as the record of the experiment, see DESIGN.md
static int attn2_env() { return DA_XENV("DA_ATTN2", 0); }
static int launch_attn_dense2(AttnDenseParams p, int heads, int n_graphs, int max_graph_nodes, hipStream_t st) {
    using CF = Cfg<bf16_t, 32, 32>;
    p.nqt = (max_graph_nodes + 255) / 256;
    const int nblocks = p.nqt * heads * n_graphs;
    k_attn_dense2<<<nblocks, 256, 4 * CF::STAGE, st>>>(p);
    DA_LAUNCH_CHECK();
    return 0;
}
#endif

template <typename T, int C, bool MASKED, int CV, int NST, int NW>
static int launch_tcmn(AttnDenseParams p, hipStream_t st) {
    using CF = Cfg<T, C, CV>;
    int lds = NST * CF::STAGE + (MASKED ? 256 : 0);
    DA_ATTN_DBG({ lds += DA_XENV("DA_ATTN_LDS_PAD", 0); })      // occupancy experiments
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_dense<T, C, MASKED, CV, NST, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    p.nqt = (p.max_nodes + 32 * NW - 1) / (32 * NW);
    k_attn_dense<T, C, MASKED, CV, NST, NW><<<p.nqt * p.H * p.n_graphs, 64 * NW, lds, st>>>(p);
    DA_LAUNCH_CHECK();
    return 0;
}
// waves per workgroup of the C = 32 instances.  Four.  DA_ATTN_NW=5 / 8 select the other shapes for A/B runs; measured at 64
// puzzles of 900 pieces (sustained, us per conv = projection + attention): 4 waves 143.3 - 145.3; 5 waves (29 slabs =
// 6 x 5 - 1 instead of 8 x 4 - 3 idle wave slots, each K / V tile streamed for 160 queries) 190.5 -- the odd wave of every
// workgroup lands on a SIMD that already holds one of its waves; 8 waves (256 queries, two workgroups per CU) 148.4 - 148.8.
[[maybe_unused]] static int attn_nw(int) {
    const int env = DA_XENV("DA_ATTN_NW", 0);
    return (env == 5 || env == 8) ? env : 4;
}
// ring depth per instance: the C = 32 layers (9 KB stages, 4 workgroups per CU) take four stages; the C = 144 layer's
// stages are 23 - 47 KB, where a third stage costs a resident workgroup: two by default, DA_ATTN_STAGES=3 for A/B runs of
// the folded bf16 instance
[[maybe_unused]] static int attn_stages_env() { return DA_XENV("DA_ATTN_STAGES", 0); }
template <typename T, int C, bool MASKED, int CV>
static int launch_tcm(const AttnDenseParams &p, hipStream_t st) {
    if constexpr (C == 32) {
        // (tried: three stages under a 96-VGPR budget = five workgroups per CU instead of four: 25 spilled registers, 184 vs
        // 167 us for the three hidden layers)
#ifdef DA_EXPERIMENTS
        if (attn_nw(p.max_nodes) == 5) return launch_tcmn<T, C, MASKED, CV, 4, 5>(p, st);
        if (attn_nw(p.max_nodes) == 8) return launch_tcmn<T, C, MASKED, CV, 4, 8>(p, st);
#endif
        return launch_tcmn<T, C, MASKED, CV, 4, 4>(p, st);
    } else if constexpr (sizeof(T) == 2 && CV == 32 && !MASKED) {
#ifdef DA_EXPERIMENTS
        if (attn_stages_env() == 3) return launch_tcmn<T, C, MASKED, CV, 3, 4>(p, st);
#endif
        return launch_tcmn<T, C, MASKED, CV, 2, 4>(p, st);
    } else {
        return launch_tcmn<T, C, MASKED, CV, 2, 4>(p, st);
    }
}
template <typename T, int C>
static int launch_tc(const AttnDenseParams &p, hipStream_t st) {
    return p.mask ? launch_tcm<T, C, true, C>(p, st) : launch_tcm<T, C, false, C>(p, st);
}

// DA_ATTN_DUAL=0: the folded last layer on k_attn_dense instead of the two-slab kernel (da_attn_dual.hip), for A/B runs
// k_attn_dual (da_attn_dual.hip) for the folded last layer is OPT-IN (DA_ATTN_DUAL=1).  Alone it is the faster kernel (harness,
// four boxes: 221 - 225 us against 224 - 230 for k_attn_dense's FAST path), inside the step it is the slower choice on every
// box measured at the end of round 3 (A/B/A/B, five boxes: 0.729 - 0.768 ms per step without it, 0.749 - 0.780 with it, -2 % each
// time): the step runs at the package power cap, and what the denser kernel saves it takes back, and more, from the
// kernels that follow it.  (Earlier in the round, with the lighter round-2 kernels on the hidden layers, the same switch measured
// +3.4 % for the dual kernel.)
[[maybe_unused]] static int attn_dual_env() { return DA_XENV("DA_ATTN_DUAL", 0); }
// returns 0 = launched, -1 = configuration not supported (caller uses the CSR kernel)
int launch_attn_dense(int prec, const DenseLayout &L, int heads, int C, int n_graphs, int max_graph_nodes,
                      const int32_t *graph_ptr, const int32_t *pad_ptr, int nodiag, const void *res, int act,
                      void *out, hipStream_t st, const DenseMask *mk, const DenseFold *fold) {
    if (heads != 8 || (C != 32 && C != 144)) return -1;
    if ((size_t)C * (size_t)L.n_pad * esize(prec) >= ((size_t)1 << 31)) return -1;      // 32-bit DMA offsets
    AttnDenseParams p;
    p.Q = L.Q; p.K = L.K; p.Vt = L.Vt; p.S = L.S; p.res = res; p.out = out;
    p.graph_ptr = graph_ptr; p.pad_ptr = pad_ptr; p.n_pad = L.n_pad; p.H = heads; p.n_graphs = n_graphs;
    p.nqt = 0; p.max_nodes = max_graph_nodes; p.act = act; p.nodiag = nodiag;
    p.x = L.x; p.ldx = L.ldx; p.kin = L.kin; p.wqs = L.wqs;
    p.bias_q = L.bias; p.bias_k = L.bias ? L.bias + heads * C : nullptr; p.bias_v = L.bias ? L.bias + 2 * heads * C : nullptr; p.bias_s = L.bias ? L.bias + 3 * heads * C : nullptr;
    p.fast = L.q_prescaled ? 1 : 0;
    if (DA_XENV("DA_ATTN_NO_FAST", 0)) p.fast = 0;
    p.sc = L.q_prescaled ? 1.0f : 1.4426950408889634f / sqrtf((float)C);      // pre-scaled Q: the scores already are in log2 units
    p.mask = mk ? mk->mask : nullptr; p.mask_ptr = mk ? (const long long *)mk->mask_ptr : nullptr;
    p.irr_row_ptr = mk ? mk->irr_row_ptr : nullptr; p.irr_col_src = mk ? mk->irr_col_src : nullptr;
    p.row_map = mk ? mk->row_map : nullptr;
    p.slot_node = mk ? mk->slot_node : nullptr;
    p.blk_class = mk ? mk->blk_class : nullptr; p.blk_class_ptr = mk ? (const long long *)mk->blk_class_ptr : nullptr;
    p.blk_class_stride = mk ? mk->blk_class_stride : 0;
    p.rm_meta = mk ? mk->rm_meta : nullptr;
    const bool virt = mk && mk->v_rows > 0 && mk->v_row_ptr && mk->v_col_src && mk->v_part && mk->v_cnt && mk->v_taken && !fold && !res;
    p.v_rows = virt ? mk->v_rows : 0; p.v_split = 0; p.v_n_real = virt ? mk->v_n_real : 0;
    p.v_row_ptr = virt ? mk->v_row_ptr : nullptr; p.v_col_src = virt ? mk->v_col_src : nullptr; p.v_mult = virt ? mk->v_mult : nullptr;
    p.v_part = virt ? mk->v_part : nullptr; p.v_cnt = virt ? mk->v_cnt : nullptr;
    if (mk && mk->v_taken) *mk->v_taken = 0;
    { const char *e = DA_XENV_LIVE("DA_ATTN_DEBUG"); p.debug = e ? atoi(e) : 0; }
    if (p.x && DA_XENV("DA_QSF_FAKE_FM", 0)) p.debug = 77;
    { const int fg = DA_XENV("DA_ATTN_FORCE_GEN", 0) ? 1 : 0; p.force_gen = fg; if (fg) p.fast = p.fast ? 2 : 0; }
    { const char *e = DA_XENV_LIVE("DA_ATTN_PROF_PTR"); p.prof = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
    if (n_graphs <= 0 || max_graph_nodes <= 0) return 0;
    p.fold_out = nullptr; p.n_rows = 0;
    if (fold) {           // value heads folded to 32 channels (last layer of the 2D transformer arch)
        if (C != 144 || fold->cv != 32) return -1;
        // DA_ATTN_LAST_FAST=0: the last layer on the referenced (round-2) recurrence while the hidden layers keep k_attn_opt (A/B)
        if (!DA_XENV("DA_ATTN_LAST_FAST", 1)) p.fast = 0;
        p.fold_out = fold->out; p.n_rows = fold->n_rows;
#ifdef DA_EXPERIMENTS
        if (!mk && prec == DA_PREC_BF16 && L.q_prescaled && attn_dual_env() DA_ATTN_DBG(&& !p.debug && !p.prof)) {
            const int r2 = launch_attn_dual(L, heads, C, n_graphs, max_graph_nodes, graph_ptr, pad_ptr, nodiag, act, out, fold, st);
            if (r2 >= 0) return r2;
        }
#endif
        // bf16 with pre-scaled Q: the optimistic kernels (da_attn_opt.hip); DA_ATTN_OPT_LAST=0 / DA_ATTN_OPT_MASKED=0 keep
        // this layer on k_attn_dense (A/B runs)
        if (prec == DA_PREC_BF16 && L.q_prescaled && p.fast && attn_opt_env() && attn_opt_last_env() && (!p.mask || attn_opt_masked_env())
            DA_ATTN_DBG(&& !p.debug && !p.prof)) {
            const int ro = launch_attn_opt(p, C, st);
            if (ro >= 0) return ro;                        // (-1: shape not covered, e.g. masked graphs beyond 4096 pieces)
        }
        if (mk) return prec == DA_PREC_BF16 ? launch_tcm<bf16_t, 144, true, 32>(p, st) : launch_tcm<float, 144, true, 32>(p, st);
        return prec == DA_PREC_BF16 ? launch_tcm<bf16_t, 144, false, 32>(p, st) : launch_tcm<float, 144, false, 32>(p, st);
    }
    if (prec == DA_PREC_BF16 && C == 32 && L.q_prescaled && p.fast && attn_opt_env() && (!p.mask || attn_opt_masked_env()) DA_ATTN_DBG(&& !p.debug && !p.prof)) {
        const int ro = launch_attn_opt(p, C, st);
        if (ro == 0 && p.v_rows > 0 && attn_opt_took_virtual_rows()) *mk->v_taken = 1;
        if (ro >= 0) return ro;
    }
    p.v_rows = 0;          // (the other kernels do not carry the virtual rows: the caller launches launch_attn_csr_cont)
    DA_REQUIRE(!p.x, "launch_attn_dense: projection in the prologue requested for a layer the resident kernel does not take");
#ifdef DA_EXPERIMENTS
    if (prec == DA_PREC_BF16 && C == 32 && !p.mask && attn2_env() DA_ATTN_DBG(&& !p.debug && !p.prof))
        return launch_attn_dense2(p, heads, n_graphs, max_graph_nodes, st);
#endif
    if (prec == DA_PREC_BF16) return C == 32 ? launch_tc<bf16_t, 32>(p, st) : launch_tc<bf16_t, 144>(p, st);
    return C == 32 ? launch_tc<float, 32>(p, st) : launch_tc<float, 144>(p, st);
}

// The conditions under which launch_attn_dense hands a hidden layer to k_attn_res<.., 64> (da_api.hip asks INSTEAD of launching the layer's
// projection kernel).  launch_attn_dense refuses (DA_REQUIRE) a layout with x set that does not end up there.
bool attn_res_qsf_shape_ok(int max_nodes, int n_pad, int n_graphs, int kin);          // da_attn_opt.hip
bool attn_qsf_applicable(int prec, int heads, int C, int kin, int n_graphs, int max_graph_nodes, int n_pad, int q_prescaled) {
    if (prec != DA_PREC_BF16 || heads != 8 || C != 32 || !q_prescaled || !attn_opt_env()) return false;
    if (DA_XENV("DA_ATTN_NO_FAST", 0) DA_ATTN_DBG(|| DA_XENV_LIVE("DA_ATTN_DEBUG") || DA_XENV_LIVE("DA_ATTN_PROF_PTR"))) return false;
    if ((size_t)C * (size_t)n_pad * 2 >= ((size_t)1 << 31)) return false;
    return attn_res_qsf_shape_ok(max_graph_nodes, n_pad, n_graphs, kin);
}

// da_debug_counters: [0] k_attn_opt workgroups re-run in GEN mode, [1] k_attn_dense waves that left FAST mode, since the last reset
int attn_dense_counters(unsigned long long *out4, int reset) {
    DA_CHECK_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_attn_fallbacks), 4 * sizeof(unsigned long long)));
    if (reset) { const unsigned long long z[4] = {0, 0, 0, 0}; DA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_fallbacks), z, sizeof(z))); }
    unsigned long long o2[2] = {0, 0};
    const int rc = attn_opt_counters(o2, reset);
    if (rc) return rc;
    out4[0] = o2[0];        // k_attn_optt workgroups re-run in GEN mode: complete graphs ...
    out4[2] = o2[1];        // ... and adjacency-masked
    return 0;
}

}  // namespace da
