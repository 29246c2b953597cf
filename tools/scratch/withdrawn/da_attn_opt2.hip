// Optimistic-softmax block-diagonal attention with TWO 32-query slabs per wave (bf16, Q pre-scaled, 32-wide value heads, complete
// graphs): the un-masked instances of k_attn_optt (da_attn_opt.hip; reference call sites backbones/Transformer_GNN.py:32,38) on
// 256-query tiles.  Why (DESIGN.md "Measured, round 5"): a wave of k_attn_optt<32> spends ~1 320 cycles per 32-key block of which
// 164 wait for the tile + barrier, 44 issue DMA, 238 wait for the K fragments out of LDS and ~200 are loop control -- costs paid
// per BLOCK AND WAVE, whatever the number of queries the wave serves.  Here every K / V fragment a wave fetches from LDS feeds two
// MFMA chains (two slabs), every barrier / tile wait / DMA round covers twice the scores, and the K / V stream of a (graph, head)
// crosses L2 -> LDS four times instead of eight.  The softmax is k_attn_optt's: p = exp2(s) un-shifted, verified once on the final
// row sums, a workgroup whose check fails re-runs its tile with the running-max recurrence (GEN).  Epilogues are register-direct
// (no staging through the ring's LDS, no barrier): the accumulator layout gives a lane four runs of four consecutive channels of
// its query's row (8 bytes each); skip / residual rows are fetched in the same shape.
// A graph's slabs are split EVENLY over its tiles (29 slabs of a 900-piece puzzle: 7 + 7 + 7 + 8); wave w of a tile serves slabs
// w and w + 4 of it (the second only where it exists: a wave-uniform branch picks the one-slab body).
#include <stdlib.h>

#include <type_traits>

#include "da_attn_common.h"

namespace da {

template <int C, bool FOLD, int NST, int MINB>
__global__ __launch_bounds__(256, MINB) void k_attn_opt2(AttnDenseParams p) {
    using T = bf16_t;
    constexpr int CV = 32, NW = 4, BK = 64;
    static_assert(FOLD || C == CV, "un-folded instances write C-wide heads");
    using CF = Cfg<T, C, CV, BK>;
    using KG = OptK<C, BK>;
    static_assert(CF::NCB == 1, "one 32-channel value block");
    constexpr int MAXI = (KG::NI + NW - 1) / NW;
    constexpr int MSTAGE = KG::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *flags = (int *)(smem + NST * MSTAGE);                  // one word per wave: "my optimistic pass failed"

    // XCD-aware remap (as k_attn_optt): XCD x takes head x of every graph, the query tiles of one (graph, head) run back to back on it
    const int bid = blockIdx.x;
    const int h = bid & 7, s_ = bid >> 3;
    const int qt = s_ % p.nqt, g = s_ / p.nqt;
    const int node0 = p.graph_ptr[g], n_g = p.graph_ptr[g + 1] - node0, pad0 = p.pad_ptr[g];
    const int ns = (n_g + 31) >> 5, ntile = (ns + 7) >> 3;
    if (qt >= ntile) return;
    const int sl0 = (qt * ns) / ntile, sl1 = ((qt + 1) * ns) / ntile;          // this tile's slabs: [sl0, sl1), at most eight

    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qsA = sl0 + wid, qsB = sl0 + 4 + wid;
    const bool onA = qsA < sl1, onB = qsB < sl1;
    const int q0A = qsA * 32, q0B = qsB * 32;
    const int qiA = q0A + i, qiB = q0B + i;
    const int HC = p.H * C;
    const size_t np = (size_t)p.n_pad;

    u32x4 qfA[CF::NCH], qfB[CF::NCH];
    {
        const unsigned char *qbase = (const unsigned char *)p.Q + ((size_t)h * np + pad0) * CF::ROWB;
        const unsigned char *ra = qbase + (size_t)(min(qsA, ns - 1) * 32 + i) * CF::ROWB, *rb = qbase + (size_t)(min(qsB, ns - 1) * 32 + i) * CF::ROWB;
#pragma unroll
        for (int ch = 0; ch < CF::NCH; ++ch) {
            qfA[ch] = *(const u32x4 *)(ra + ch * 32 + half * 16);
            qfB[ch] = *(const u32x4 *)(rb + ch * 32 + half * 16);
        }
    }
    const unsigned char *Kg = (const unsigned char *)p.K + ((size_t)h * np + pad0) * CF::ROWB;
    const unsigned char *Vg = (const unsigned char *)p.Vt + ((size_t)h * np + pad0) * CF::ROWBV;
    unsigned soff[MAXI];
#pragma unroll
    for (int x = 0; x < MAXI; ++x) {
        const int q = wid + NW * x;
        unsigned o = 0;
        if (q < KG::NIK) {
            const int s = q * 64 + lane, row = s / KG::KSPR, col = s - row * KG::KSPR;
            if (KG::SWZ) o = (unsigned)(row * CF::ROWB + (col ^ KG::f(row)) * 16);
            else if (row < CF::BKEYS && col < CF::KVALID) o = (unsigned)(row * CF::ROWB + col * 16);
        } else {
            const int s = (q - KG::NIK) * 64 + lane, row = s / CF::VSPR, col = s - row * CF::VSPR;
            if (row < CF::BKEYS && col < CF::KVALIDV) o = (unsigned)(row * CF::ROWBV + col * 16);
        }
        soff[x] = o;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // one tile = 64 keys of K and V into stage `stage`: scalar base + 32-bit lane offset form (see k_attn_optt)
    auto issue = [&](int kt, int stage) {
        const unsigned char *kb_ = Kg + (size_t)kt * CF::BKEYS * CF::ROWB;
        const unsigned char *vb_ = Vg + (size_t)kt * CF::BKEYS * CF::ROWBV;
        const unsigned sbl = lds0 + (unsigned)(stage * MSTAGE);
#pragma unroll
        for (int x = 0; x < MAXI; ++x) {
            const int q = wid + NW * x;
            if (NW * x + NW - 1 < KG::NI || q < KG::NI)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(sbl + q * 1024), "v"(soff[x]),
                             "s"(q < KG::NIK ? kb_ : vb_)
                             : "memory");
        }
    };
    const int myn = (KG::NI - wid + NW - 1) / NW;               // DMA instructions this wave issues per tile
    const int nkt = (n_g + CF::BKEYS - 1) / CF::BKEYS;
    const int pi_i = (i & 3) + 4 * ((i >> 3) & 3) + 16 * ((i >> 2) & 1);
    int kfo[CF::NCH];
#pragma unroll
    for (int ch = 0; ch < CF::NCH; ++ch) kfo[ch] = pi_i * KG::RS + (KG::SWZ ? (((2 * ch + half) ^ KG::f(pi_i)) * 16) : (ch * 32 + half * 16));
    const int li = lane & 15;
    const int vbase = KG::KBYTES + (16 * half + (li >> 2)) * CF::RSV + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;

    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};

    f32x16 OA, OB;
    float lsA = 0.f, lsB = 0.f;       // this lane's share of the row sums (16 of a block's 32 keys)
    float mA = 0.f, mB = 0.f;         // GEN mode only: running row max (log2 units)
    bool gen = p.force_gen != 0;

    // softmax + PV of one slab's 32 x 32 scores (s: this lane's 16 keys of query row i); v0 / v1: the block's V fragments
    auto finish = [&](f32x16 &s, f32x16 &O, float &ls, float &m, const u32x4 &v0, const u32x4 &v1) {
        if (gen) {
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
        }
        O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf0, O, 0, 0, 0);
        O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf1, O, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16x2 pa = {pf0[2 * e], pf0[2 * e + 1]}, pb = {pf1[2 * e], pf1[2 * e + 1]};
            ls = __builtin_amdgcn_fdot2_f32_bf16(pa, one2, ls, false);
            ls = __builtin_amdgcn_fdot2_f32_bf16(pb, one2, ls, false);
        }
    };
    auto edge_mask = [&](f32x16 &s, int key0, int q0, int qidx) {        // the last block's keys beyond the graph; a missing diagonal
        const bool tail = key0 + 32 > n_g;
        const bool diag = p.nodiag && key0 < q0 + 32 && key0 + 32 > q0;
        if (tail || diag) {
            const int kbase = key0 + 16 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + r >= n_g || (p.nodiag && kbase + r == qidx)) s[r] = -INFINITY;
        }
    };
    // one 32-key block for both slabs (DUAL) or for slab A alone
    auto block = [&](auto dual_tag, int stage, int kb, int key0) {
        constexpr bool DUAL = decltype(dual_tag)::value;
        const unsigned char *stg = smem + stage * MSTAGE;
        u32x4 kf[CF::NCH];
#pragma unroll
        for (int ch = 0; ch < CF::NCH; ++ch) kf[ch] = *(const u32x4 *)(stg + kfo[ch] + kb * 32 * KG::RS);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sA, sB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
#pragma unroll
        for (int ch = 0; ch < CF::NCH; ++ch) {
            sA = mma_chunk(T(), kf[ch], qfA[ch], sA);
            if (DUAL) sB = mma_chunk(T(), kf[ch], qfB[ch], sB);
        }
        u32x2 vlo[2], vhi[2];
        const unsigned vb = lds0 + (unsigned)(stage * MSTAGE + vbase + kb * 32 * CF::RSV);
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
            vlo[mm] = tr_read(vb, (8 * mm) * CF::RSV);
            vhi[mm] = tr_read(vb, (8 * mm + 4) * CF::RSV);
        }
        edge_mask(sA, key0, q0A, qiA);
        if (DUAL) edge_mask(sB, key0, q0B, qiB);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]));
        const u32x4 v0 = {vlo[0][0], vlo[0][1], vhi[0][0], vhi[0][1]};
        const u32x4 v1 = {vlo[1][0], vlo[1][1], vhi[1][0], vhi[1][1]};
        finish(sA, OA, lsA, mA, v0, v1);
        if (DUAL) finish(sB, OB, lsB, mB, v0, v1);
    };

    for (int attempt = 0; attempt < 2; ++attempt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { OA[r] = 0.f; OB[r] = 0.f; }
        lsA = lsB = 0.f;
        mA = mB = -1e30f;
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nkt) issue(st, st);
        for (int j = 0; j < nkt; ++j) {
            {
                // tiles that may stay in flight behind this one (each is `myn` operations of this wave; myn is LO or LO + 1)
                constexpr int LO = KG::NI / NW;
                const int younger = min(nkt - 1 - j, NST - 2);
                if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (younger == 1) { if (myn == LO) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LO) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LO + 1) : "memory"); }
                else if (younger == 2 || NST <= 4) { if (myn == LO) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LO) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LO + 2) : "memory"); }
                else wait_vmcnt(younger * myn);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (j + NST - 1 < nkt) issue(j + NST - 1, (j + NST - 1) % NST);
            if (!onA) continue;
#pragma unroll
            for (int kb = 0; kb < CF::KB; ++kb) {
                const int key0 = j * CF::BKEYS + kb * 32;
                if (key0 >= n_g) break;
                if (onB) block(std::true_type(), j % NST, kb, key0);
                else block(std::false_type(), j % NST, kb, key0);
            }
        }
        if (gen) break;
        // ---- verification of the optimistic pass (workgroup-uniform verdict: the waves share the K / V stream)
        const float ltA = lsA + __shfl_xor(lsA, 32), ltB = lsB + __shfl_xor(lsB, 32);
        const bool okA = ltA > 8.673617379884035e-19f && ltA < 1.2676506002282294e30f;        // 2^-60, 2^100; NaN fails
        const bool okB = ltB > 8.673617379884035e-19f && ltB < 1.2676506002282294e30f;
        const bool bad = (onA && __any(!okA && qiA < n_g)) || (onB && __any(!okB && qiB < n_g));
        if (lane == 0) flags[wid] = bad ? 1 : 0;
        __syncthreads();                                    // every wave has left the key loop: the ring's LDS is free from here on
        bool redo = false;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) redo = redo || flags[w_] != 0;
        if (!redo) break;
        __syncthreads();
        gen = true;
        if (tid == 0 && p.fb_ctr) atomicAdd(p.fb_ctr, 1ull);
    }

    // ---- epilogue, straight from the accumulator layout: lane (i, half) holds channels 8 jj + 4 half + 0 .. 3 (jj = 0 .. 3) of query row i
    auto epilogue = [&](const f32x16 &O, float ls, int qidx) {
        const float lt = ls + __shfl_xor(ls, 32);
        const float inv = lt > 0.f ? 1.0f / (lt + (gen ? 1e-16f : 0.f)) : 0.f;      // (no epsilon on an un-shifted sum: see k_attn_optt's header)
        if (qidx >= n_g) return;
        if constexpr (FOLD) {
            T *dst = (T *)p.fold_out + ((size_t)h * p.n_rows + node0 + qidx) * CV;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float v4[4] = {O[4 * jj] * inv, O[4 * jj + 1] * inv, O[4 * jj + 2] * inv, O[4 * jj + 3] * inv};
                st4(dst + 8 * jj + 4 * half, v4);
            }
        } else {
            const size_t off = (size_t)(node0 + qidx) * HC + (size_t)h * C + 4 * half;
            u32x2 sk[4], rs[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                sk[jj] = *(const u32x2 *)((const T *)p.S + off + 8 * jj);
                if (p.res) rs[jj] = *(const u32x2 *)((const T *)p.res + off + 8 * jj);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float v4[4] = {O[4 * jj] * inv, O[4 * jj + 1] * inv, O[4 * jj + 2] * inv, O[4 * jj + 3] * inv};
                v4[0] += bf2f((bf16_t)(sk[jj][0] & 0xffff)); v4[1] += bf2f((bf16_t)(sk[jj][0] >> 16));
                v4[2] += bf2f((bf16_t)(sk[jj][1] & 0xffff)); v4[3] += bf2f((bf16_t)(sk[jj][1] >> 16));
                if (p.res) {
                    v4[0] += bf2f((bf16_t)(rs[jj][0] & 0xffff)); v4[1] += bf2f((bf16_t)(rs[jj][0] >> 16));
                    v4[2] += bf2f((bf16_t)(rs[jj][1] & 0xffff)); v4[3] += bf2f((bf16_t)(rs[jj][1] >> 16));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = apply_act(v4[e], p.act);
                st4((T *)p.out + off + 8 * jj, v4);
            }
        }
    };
    if (onA) epilogue(OA, lsA, qiA);
    if (onB) epilogue(OB, lsB, qiB);
}

template <int C, bool FOLD, int NST, int MINB>
static int launch_opt2t(AttnDenseParams p, hipStream_t st) {
    const int lds = NST * OptK<C, 64>::STAGE + 64;
    static bool attr_done[16] = {};           // per device: the attribute belongs to the device's copy of the function
    int dev = 0;
    DA_CHECK_HIP(hipGetDevice(&dev));
    if (lds > 48 * 1024 && !attr_done[dev & 15]) {
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_attn_opt2<C, FOLD, NST, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done[dev & 15] = true;
    }
    p.nqt = ((p.max_nodes + 31) / 32 + 7) / 8;
    p.fb_ctr = attn_opt_fallback_counters();
    k_attn_opt2<C, FOLD, NST, MINB><<<p.nqt * p.H * p.n_graphs, 256, lds, st>>>(p);
    DA_LAUNCH_CHECK();
    return 0;
}

// variant = the DA_OPT_HID / DA_OPT_LAST value (80 ...); -1 = shape / variant not covered
int launch_attn_opt2(const AttnDenseParams &p, int C, int variant, hipStream_t st) {
    if (p.mask) return -1;
    const bool fold = p.fold_out != nullptr;
    if (C == 32 && !fold) {
        switch (variant) {
            case 80: return launch_opt2t<32, false, 4, 3>(p, st);
            case 81: return launch_opt2t<32, false, 4, 4>(p, st);
            case 82: return launch_opt2t<32, false, 3, 4>(p, st);
            case 83: return launch_opt2t<32, false, 6, 3>(p, st);
            default: return -1;
        }
    }
    if (C == 144 && fold) {
        switch (variant) {
            case 80: return launch_opt2t<144, true, 2, 2>(p, st);
            case 81: return launch_opt2t<144, true, 3, 2>(p, st);
            default: return -1;
        }
    }
    return -1;
}

}  // namespace da
