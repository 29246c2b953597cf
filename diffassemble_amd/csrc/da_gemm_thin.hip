// EXPERIMENTS BUILD ONLY (DA_EXPERIMENTS=1 python __graft_entry__.py; __graft_entry__.EXPERIMENT_SOURCES): four-wave projection kernel co-resident with k_attn_res: co-runs as designed, the step loses 3 - 5 % (round 6, profiles/r06).
// CO-RESIDENT projection kernel: the fused Q | K | V (| skip) projections sized to what one workgroup of the K / V-resident
// attention kernel (k_attn_res: sixteen waves x 96 VGPRs, 120 KB of LDS) leaves free on a CU -- 128 VGPRs per SIMD lane, 40 KB
// of LDS and four wave slots per SIMD -- so that, in the two-stream pair loop, the projections of one half Batch run INSIDE the
// CUs that the other half Batch's hidden-layer attention occupies (the exp-bound attention leaves the matrix
// pipe 79 % and HBM 80 % idle; the store-bound projection leaves the vector unit idle).  None of the other projection kernels can
// do that: k_gemm_xpanel takes the whole register file (8 waves x 256 VGPRs), k_gemm_wreg / wreg2 nine / five waves of 152 / 243.
//
// Same arithmetic as k_gemm_wreg (W-in-registers, D^T[col][node] = W . x^T on v_mfma_f32_32x32x16_bf16, k-steps in ascending order
// into one zero-initialised accumulator, + bias in fp32, one rounding): outputs are BIT-IDENTICAL to the other projection kernels.
//   * workgroup = FOUR waves, one per SIMD: three consumers (32 output columns each, W fragments = 64 VGPRs at K = 256) and one
//     producer.  VGPRs <= 128 (amdgpu_num_vgpr), so the workgroup needs 128 of a SIMD's 512 registers.
//   * A arrives by LDS-DMA in STAGES of 32 rows x 128 K (8 KB) through a four-stage ring (32 KB): at K = 256 a tile is two stages,
//     three stages (24 KB) in flight while one is multiplied.  One barrier per stage (8 MFMAs per consumer).
//   * the producer wave issues every DMA and is the only wave that waits on vmcnt (counted); consumers only ever issue stores
//     (gfx950 counts loads and stores in one counter and they return out of order with respect to each other).
//   * register-direct epilogue (the W rows are permuted so that lane (node, half) ends with 16 CONSECUTIVE columns of its node:
//     two 16-byte stores); the bias of the wave's 32 columns sits in LDS (128 B per consumer), not in registers.
//   * 96 output columns per workgroup, ceil(Nout / 96) column groups; the XCD-aware map of k_gemm_wreg keeps the column groups
//     that walk the same rows on one XCD, so A crosses the fabric once per row chunk and the other groups hit that XCD's L2.
// LDS: 4 x 8 KB ring + 4 x 256 B row slots + 3 x 128 B bias = 34 176 B.
#include <stdlib.h>

#include "da_gemm_common.h"

namespace da {

typedef __attribute__((ext_vector_type(16))) float f32x16t;

template <int N> __device__ __forceinline__ void thin_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void thin_dma16(unsigned lds_addr, unsigned voff, const void *sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void thin_dma4(unsigned lds_addr, unsigned voff, const void *sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

constexpr int THIN_NSTG = 4;            // ring stages
constexpr int THIN_STGB = 32 * 256;     // bytes of one stage: 32 rows x 128 K bf16
constexpr int THIN_LDS = THIN_NSTG * THIN_STGB + 4 * 256 + 3 * 128;

template <int KIN, bool QKV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_gemm_thin(GemmParams p, int tiles_per_wg, int ncg, int cpx) {
    constexpr int SPT = KIN / 128;               // stages per 32-row tile
    constexpr int NSTG = THIN_NSTG, STGB = THIN_STGB;
    constexpr int PER = 9;                       // vector-memory instructions the producer issues per stage: 8 x 1 KB of A + the row slots
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *ring = smem;                                  // [NSTG][STGB]
    unsigned char *slots = smem + NSTG * STGB;                   // [4][256 B]: padded-row slot of a tile's 32 nodes
    float *bias_l = (float *)(slots + 4 * 256);                  // [3][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nrt = (p.M + 31) >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cgi = slot % ncg, chunk = xcd * cpx + slot / ncg;
    const int t0 = chunk * tiles_per_wg, t1 = min(t0 + tiles_per_wg, nrt);
    const int ntile = t1 - t0;
    if (ntile <= 0) return;
    const int nstage = ntile * SPT;

    if (wid == 3) {
        // ------------------------------------------------ producer
        const int rsub = lane >> 4, pc = lane & 15;               // row inside a 4-row DMA instruction, physical 16-byte chunk
        const unsigned ring_lds = (unsigned)(size_t)ring, slots_lds = (unsigned)(size_t)slots;
        unsigned voff[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = q * 4 + rsub;
            voff[q] = (unsigned)row * (unsigned)p.lda * 2u + (unsigned)(pc ^ (row & 15)) * 16u;
        }
        const unsigned voff_rm = (unsigned)min(lane, 31) * 4u;
        auto issue = [&](int sj) {
            const int ti = sj / SPT, ks = sj % SPT;
            const int row0 = (t0 + ti) * 32;
            const unsigned buf = ring_lds + (unsigned)(sj % NSTG) * STGB;
            const char *base = (const char *)p.A + (size_t)row0 * (size_t)p.lda * 2 + (size_t)ks * 256;
            if (row0 + 32 <= p.M) {
#pragma unroll
                for (int q = 0; q < 8; ++q) thin_dma16(buf + q * 1024, voff[q], base);
                thin_dma4(slots_lds + (unsigned)(ti & 3) * 256, voff_rm, QKV ? (const void *)(p.row_map + row0) : (const void *)base);
                return;
            }
            // last, partial tile: rows past the end re-read row M - 1
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = q * 4 + rsub;
                thin_dma16(buf + q * 1024, (unsigned)(min(row0 + row, p.M - 1) - row0) * (unsigned)p.lda * 2u + (unsigned)(pc ^ (row & 15)) * 16u, base);
            }
            thin_dma4(slots_lds + (unsigned)(ti & 3) * 256, (unsigned)(min(row0 + min(lane, 31), p.M - 1) - row0) * 4u,
                      QKV ? (const void *)(p.row_map + row0) : (const void *)base);
        };
        const int pre = min(NSTG - 1, nstage);
        for (int sj = 0; sj < pre; ++sj) issue(sj);
        __builtin_amdgcn_s_barrier();                             // (pairs with the consumers' bias barrier)
        for (int j = 0; j < nstage; ++j) {
            const int younger = min(nstage, j + NSTG - 1) - 1 - j;   // stages after j that may stay in flight: 0 .. NSTG - 2
            if (younger >= 2) thin_wait_vmcnt<2 * PER>();
            else if (younger == 1) thin_wait_vmcnt<PER>();
            else thin_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                         // stage j is readable; everyone is done with stage j - 1
            if (j + NSTG - 1 < nstage) issue(j + NSTG - 1);       // into the slot of stage j - 1
        }
        return;
    }

    // ---------------------------------------------------- consumers
    const int i32 = lane & 31, half = lane >> 5;
    const int col0 = cgi * 96 + wid * 32;
    const bool active = col0 < p.Nout;
    // W rows permuted (k_gemm_wreg's DIRECT form): MFMA row 8 j + 4 h + i of the A operand is fed W column col0 + 16 h + 4 j + i, so the
    // accumulator of lane (node, half) holds the 16 consecutive output columns col0 + 16 half + (0 .. 15) of its node
    constexpr int KS = KIN / 16;
    u32x4 wf[KS];
    {
        const int wsel = 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
        const char *wrow = (const char *)p.W + (size_t)min(col0 + wsel, p.Nout - 1) * (size_t)p.ldw * 2 + half * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[s] = *(const u32x4 *)(wrow + s * 32);
    }
    if (lane < 32) bias_l[wid * 32 + lane] = (p.bias && active) ? p.bias[min(col0 + lane, p.Nout - 1)] : 0.f;
    const int colc = col0 + 16 * half;
    bf16_t *dbase;
    size_t rstride;
    bool use_slot = false;
    if (!QKV) {
        dbase = (bf16_t *)p.out + colc;
        rstride = (size_t)p.ldo;
    } else {
        const int which = min(colc / p.HC, 3), f = colc - which * p.HC;
        if (which == 2 && p.Cv > 0) {
            const int h = f / p.Cv, c = f - h * p.Cv;
            dbase = (bf16_t *)p.Vt + (size_t)h * p.n_pad * p.Cv + c;
            rstride = (size_t)p.Cv;
            use_slot = true;
        } else if (which == 3) {
            dbase = (bf16_t *)p.S + f;
            rstride = (size_t)p.HC;
        } else {
            const int h = f / p.C, c = f - h * p.C;
            dbase = (bf16_t *)(which == 0 ? p.Q : (which == 1 ? p.Kb : p.Vt)) + (size_t)h * p.n_pad * p.C + c;
            rstride = (size_t)p.C;
            use_slot = true;
        }
    }
    const float *bz = bias_l + wid * 32 + 16 * half;
    // the W / bias loads are waited for here, once: inside the loop this wave only ever issues stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // bias rows visible (the producer joins after its first issues)
    for (int ti = 0; ti < ntile; ++ti) {
        f32x16t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < SPT; ++ks) {
            const int j = ti * SPT + ks;
            __builtin_amdgcn_s_barrier();                         // producer: stage j has landed
            if (active) {
                const unsigned char *buf = ring + (j % NSTG) * STGB + i32 * 256;
                // x fragments four k-steps (128 cycles of the matrix pipe) ahead of their MFMAs, through four rotating registers
                u32x4 xa[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) xa[s] = *(const u32x4 *)(buf + (((2 * s + half) ^ (i32 & 15)) << 4));
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ks * 8 + s]), __builtin_bit_cast(bf16x8, xa[s & 3]), acc, 0, 0, 0);
                    if (s + 4 < 8) xa[s & 3] = *(const u32x4 *)(buf + (((2 * (s + 4) + half) ^ (i32 & 15)) << 4));
                }
            }
        }
        if (!active) continue;
        bf16x8 o0, o1;
        {
            const f32x4 b0 = *(const f32x4 *)(bz), b1 = *(const f32x4 *)(bz + 4), b2 = *(const f32x4 *)(bz + 8), b3 = *(const f32x4 *)(bz + 12);
            const float bv[16] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o0[e] = (__bf16)(acc[e] + bv[e]);
                o1[e] = (__bf16)(acc[8 + e] + bv[8 + e]);
            }
        }
        const int m = (t0 + ti) * 32 + i32;
        if (m < p.M) {
            const size_t ridx = use_slot ? (size_t)((const int32_t *)(slots + (ti & 3) * 256))[i32] : (size_t)m;
            bf16_t *dst = dbase + ridx * rstride;
            *(u32x4 *)dst = __builtin_bit_cast(u32x4, o0);
            *(u32x4 *)(dst + 8) = __builtin_bit_cast(u32x4, o1);
        }
    }
}

// DA_GEMM_THIN: 1 = the projections of the sampling step take this kernel (set by the pair loop's co-residency mode); DA_THIN_TPW = row
// tiles per workgroup (default 12).  returns 0 = launched, -1 = not applicable
int launch_gemm_thin(int prec, const GemmParams &p0, const QkvScatter *qs, int act, hipStream_t st) {
    if (prec != DA_PREC_BF16 || act != DA_ACT_NONE) return -1;
    GemmParams p = p0;
    if ((p.K != 128 && p.K != 256) || p.pre || p.res || p.M < 2048 || (p.Nout & 31)) return -1;
    if (qs) {
        // a lane's 16 consecutive columns must stay inside one column block and one head
        if ((qs->HC & 31) || (qs->C & 15) || (qs->Cv & 15)) return -1;
    } else if ((p.ldo & 7) || (((size_t)p.out) & 15)) {
        return -1;
    }
    static int tpw = -1;
    if (tpw < 0) { tpw = DA_XENV("DA_THIN_TPW", 12); if (tpw < 1) tpw = 1; }
    const int nrt = (p.M + 31) / 32;
    const int ncg = (p.Nout + 95) / 96;
    int cpx = (nrt + 8 * tpw - 1) / (8 * tpw);
    cpx = cpx < 1 ? 1 : cpx;
    const int nchunk = 8 * cpx;
    const int tiles = (nrt + nchunk - 1) / nchunk;
    const dim3 grid((unsigned)(nchunk * ncg));
    if (qs) {
        if (p.K == 256) k_gemm_thin<256, true><<<grid, 256, THIN_LDS, st>>>(p, tiles, ncg, cpx);
        else k_gemm_thin<128, true><<<grid, 256, THIN_LDS, st>>>(p, tiles, ncg, cpx);
    } else {
        if (p.K == 256) k_gemm_thin<256, false><<<grid, 256, THIN_LDS, st>>>(p, tiles, ncg, cpx);
        else k_gemm_thin<128, false><<<grid, 256, THIN_LDS, st>>>(p, tiles, ncg, cpx);
    }
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
