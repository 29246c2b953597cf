// W-in-registers MFMA linear kernel for SHORT reductions and TALL inputs (bf16, K in {128, 256}, M >> Nout):
// the fused Q | K | V (| skip) projections, e.g. conv 3 of the 2D denoiser: M = 28 800 nodes, K = 256, Nout = 2560.
//
// The A-stationary kernel (da_gemm_astat.hip) keeps a 128-row panel of A in LDS and streams W tiles through a
// barrier every 8 MFMAs per wave; measured 435-484 TFLOP/s on these shapes, far from every roof.
// Here the roles are swapped and the barrier count drops 4x while the LDS bytes per FLOP drop 2x:
//   * every consumer wave keeps ITS 32 output columns of W -- 32 x K bf16 = 64 VGPRs at K = 256 -- in registers for
//     the whole kernel, as the A operand of v_mfma_f32_32x32x16_bf16 (D^T[col][node] = W . x^T);
//   * the workgroup (8 consumer waves = 256 columns) walks its share of the 32-row tiles of A: a tile (32 x K,
//     16 KB) arrives by LDS-DMA into a 4-slot ring, every consumer multiplies the SAME tile against its own columns
//     (one 16-byte LDS read per MFMA, conflict-free through an XOR swizzle applied on the DMA source address);
//   * a ninth wave is the PRODUCER: it issues all DMA of a tile (17 instructions: 16 KB of A + the padded-row slots
//     of the 32 nodes), waits for them with COUNTED vmcnt and joins the one barrier per tile.  Consumers therefore
//     never wait on vmcnt: their only vector-memory operations inside the loop are the output stores, which stay in
//     flight for as long as HBM needs (gfx950's vmcnt counts loads and stores together and the two return out of
//     order with respect to each other, so a wave that mixes both can only ever wait for "everything");
//   * epilogue per wave: the 32 x 32 accumulator goes through a wave-private fp32 LDS strip (no workgroup barrier)
//     and leaves as 16-byte stores, bias / activation applied on the row-major side in fp32 (one rounding).
// QKV scatter mode writes Q / K / V head-major at the padded row slots and skip row-major, like the other kernels.
#include <stdlib.h>

#include "da_gemm_common.h"

namespace da {

typedef __attribute__((ext_vector_type(16))) float f32x16w;

#ifndef DA_WREG_NSTG
#define DA_WREG_NSTG 4
#endif
#ifdef DA_WREG_PROBE
// timeline of workgroup (0, 0): consumer waves 0 and 4 (one SIMD) and the producer, first 64 tiles, 4 stamps per tile
#define DA_WTICK(w, t, k) do { if (p.prof && blockIdx.x == 0 && (t) < 64) p.prof[((w) * 64 + (t)) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define DA_WTICK(w, t, k)
#endif
template <int N> __device__ __forceinline__ void wreg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The producer wave of both kernels.  Its instruction stream shares a SIMD's issue port with two consumer waves' MFMAs, so it
// must be SHORT: per DMA instruction one s_mov to M0 and the load itself, in the "SGPR base + 32-bit VGPR offset" form -- the
// per-lane offsets inside a tile never change (16 VGPRs, computed once) and the tile's base is a scalar.  (Round 3 timeline,
// conv 3 at 57 600 rows: with a 64-bit per-lane address rebuilt for every instruction -- 7 VALU, three of them quarter-rate --
// issuing one tile took 1840 cycles (3400 beside the output stores) and the eight consumer waves, done after 1400, waited at
// the barrier for it: tile period 2350 / 3500 cycles against 1024 of MFMA work per SIMD.)
__device__ __forceinline__ void wreg_dma16(unsigned lds_addr, unsigned voff, const void *sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void wreg_dma4(unsigned lds_addr, unsigned voff, const void *sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

template <int KIN, bool QKV, int NSTG>
__device__ __forceinline__ void wreg_producer(const GemmParams &p, unsigned char *ring, unsigned char *slots, int t0, int ntile, int lane) {
    constexpr int ROWB = KIN * 2, TILEB = 32 * ROWB, NDMA = TILEB / 1024, CPR = ROWB / 16, PER = NDMA + 1;
    const int rsub = lane / CPR, pc = lane % CPR;                 // row inside a DMA instruction, physical chunk
    const unsigned ring_lds = (unsigned)(size_t)ring, slots_lds = (unsigned)(size_t)slots;
    unsigned voff[NDMA];
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
        const int row = q * (64 / CPR) + rsub;                    // row inside the tile
        const int lc = pc ^ (row & 15);                           // logical chunk this LDS slot holds
        voff[q] = (unsigned)row * (unsigned)p.lda * 2u + (unsigned)lc * 16u;
    }
    const unsigned voff_rm = (unsigned)min(lane, 31) * 4u;
    auto issue = [&](int ti) {
        if (p.debug & 1024) return;
        const int row0 = (t0 + ti) * 32;
        const unsigned buf = ring_lds + (unsigned)(ti % NSTG) * TILEB;
        if (row0 + 32 <= p.M) {
            const char *base = (const char *)p.A + (size_t)row0 * (size_t)p.lda * 2;
#pragma unroll
            for (int q = 0; q < NDMA; ++q) wreg_dma16(buf + q * 1024, voff[q], base);
            // padded-row slots of the tile's nodes (QKV scatter); one 4-byte piece per lane, lanes >= 32 re-load row 31
            wreg_dma4(slots_lds + (unsigned)(ti % NSTG) * 256, voff_rm, QKV ? (const void *)(p.row_map + row0) : (const void *)base);
            return;
        }
        // the last, partial tile: rows past the end re-read row M - 1 (M0 is only ever written by these asm statements: no
        // compiler-generated LDS-DMA in this function, whose M0 bookkeeping they would bypass)
        const char *base = (const char *)p.A + (size_t)row0 * (size_t)p.lda * 2;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const int row = q * (64 / CPR) + rsub;
            const int lc = pc ^ (row & 15);
            wreg_dma16(buf + q * 1024, (unsigned)(min(row0 + row, p.M - 1) - row0) * (unsigned)p.lda * 2u + (unsigned)lc * 16u, base);
        }
        wreg_dma4(slots_lds + (unsigned)(ti % NSTG) * 256, (unsigned)(min(row0 + min(lane, 31), p.M - 1) - row0) * 4u,
                  QKV ? (const void *)(p.row_map + row0) : (const void *)base);
    };
    const int pre = min(NSTG - 1, ntile);
    for (int ti = 0; ti < pre; ++ti) issue(ti);
    for (int i = 0; i < ntile; ++i) {
        const int issued = min(ntile, i + NSTG - 1);              // tiles 0 .. issued-1 are in flight or landed
        const int younger = issued - 1 - i;                       // tiles after i that may stay in flight: 0 .. NSTG - 2
        static_assert((NSTG - 2) * PER <= 63, "vmcnt is a 6-bit counter");
        DA_WTICK(2, i, 0);
        if (NSTG >= 5 && younger >= 3) wreg_wait_vmcnt<(NSTG >= 5 ? 3 : 0) * PER>();
        else if (younger >= 2) wreg_wait_vmcnt<2 * PER>();
        else if (younger == 1) wreg_wait_vmcnt<PER>();
        else wreg_wait_vmcnt<0>();
        DA_WTICK(2, i, 1);
        __builtin_amdgcn_s_barrier();                             // tile i is readable; everyone is done with tile i - 1
        DA_WTICK(2, i, 2);
        if (i + NSTG - 1 < ntile) issue(i + NSTG - 1);            // into the slot of tile i - 1
        DA_WTICK(2, i, 3);
    }
}

template <int KIN, bool QKV, int ACT, bool DIRECT>
__global__ __launch_bounds__(576) void k_gemm_wreg(GemmParams p, int tiles_per_wg, int ncg, int cpx) {
    constexpr int KS = KIN / 16;                 // k-steps of 16
    constexpr int ROWB = KIN * 2;                // bytes of one A row
    constexpr int TILEB = 32 * ROWB;             // one 32-row tile
    constexpr int NDMA = TILEB / 1024;           // 1 KB DMA instructions per tile
    constexpr int CPR = ROWB / 16;               // 16-byte chunks per row (32 / 16)
    constexpr int NSTG = DA_WREG_NSTG;
    constexpr int SROW = 144;                    // strip row: 32 fp32 + 16 B pad
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *ring = smem;                                  // [NSTG][TILEB]
    unsigned char *slots = smem + NSTG * TILEB;                  // [NSTG][256 B]: padded-row slot of the tile's 32 nodes
    unsigned char *strips = slots + NSTG * 256;                  // [8][32][SROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nrt = (p.M + 31) >> 5;
    // XCD-aware map (workgroup L runs on XCD L % 8): the ncg column groups that walk the SAME rows sit on one XCD, so a tile of A
    // comes over the fabric once per row chunk and the other ncg - 1 readers hit that XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cgi = slot % ncg, chunk = xcd * cpx + slot / ncg;
    const int t0 = chunk * tiles_per_wg, t1 = min(t0 + tiles_per_wg, nrt);
    const int ntile = t1 - t0;
    if (ntile <= 0) return;

    if (wid == 8) {
        wreg_producer<KIN, QKV, NSTG>(p, ring, slots, t0, ntile, lane);
        return;
    }

    // ---------------------------------------------------- consumer waves
    const int i32 = lane & 31, half = lane >> 5;
    const int col0 = cgi * 256 + wid * 32;
    const bool active = col0 < p.Nout;
    // this wave's 32 columns of W as A-operand fragments: lane (col, half), k-step s -> W[col][16 s + 8 half ..+8]
    // DIRECT: MFMA row m of the A operand is fed W column col0 + pi(m), pi(8 j + 4 h + i) = 16 h + 4 j + i, so that the
    // accumulator of lane (node, half) -- rows 8 j + 4 half + i in registers 4 j + i -- is the 16 CONSECUTIVE output
    // columns col0 + 16 half + (0 .. 15) of its node: 32 bytes of bf16 that leave as two 16-byte stores straight from
    // registers.
    u32x4 wf[KS];
    {
        const int wsel = DIRECT ? 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3) : i32;
        const char *wrow = (const char *)p.W + (size_t)min(col0 + wsel, p.Nout - 1) * (size_t)p.ldw * 2 + half * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[s] = *(const u32x4 *)(wrow + s * 32);
    }
    // row-major side of the epilogue: this lane stores the 16-byte chunk `ch` (8 columns) of rows rr and rr + 16
    // (DIRECT: its node's columns colc .. colc + 15)
    const int ch = lane & 3, rr = lane >> 2;
    const int colc = DIRECT ? col0 + 16 * half : col0 + 8 * ch;
    float bz[8];
    f32x16w bzv;
    if (DIRECT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) bzv[e] = (p.bias && active) ? p.bias[min(colc + e, p.Nout - 1)] : 0.f;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bz[e] = (p.bias && active) ? p.bias[min(colc + e, p.Nout - 1)] : 0.f;
    }
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
    unsigned char *strip = strips + wid * (32 * SROW);
    // wait for the W / bias loads here, once: inside the loop this wave only ever issues stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < ntile; ++i) {
        __builtin_amdgcn_s_barrier();                             // producer: tile i has landed
        if (!active) continue;
        if ((wid & 3) == 0) DA_WTICK(wid >> 2, i, 0);
        const unsigned char *buf = ring + (i % NSTG) * TILEB + i32 * ROWB;
        f32x16w acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // x fragments one group of four k-steps ahead of their MFMAs (a dependent 16-byte LDS read in front of every
        // MFMA leaves the matrix pipe idle for the read latency); sched_barrier keeps hipcc from sinking the reads again
        constexpr int PF = 4, NG = KS / PF;
        u32x4 xa[2][PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) xa[0][s] = *(const u32x4 *)(buf + (((2 * s + half) ^ (i32 & 15)) << 4));
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) {
#pragma unroll
                for (int s = 0; s < PF; ++s)
                    xa[(gq + 1) & 1][s] = *(const u32x4 *)(buf + (((2 * ((gq + 1) * PF + s) + half) ^ (i32 & 15)) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < PF; ++s)
                if (!(p.debug & 512)) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[gq * PF + s]),
                                                              __builtin_bit_cast(bf16x8, xa[gq & 1][s]), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((wid & 3) == 0) DA_WTICK(wid >> 2, i, 1);
        if (DIRECT) {
            bf16x8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o0[e] = (__bf16)apply_act(acc[e] + bzv[e], ACT);
                o1[e] = (__bf16)apply_act(acc[8 + e] + bzv[8 + e], ACT);
            }
            const int m = (t0 + i) * 32 + i32;
            if (m < p.M && !(p.debug & 256)) {
                const size_t ridx = use_slot ? (size_t)((const int32_t *)(slots + (i % NSTG) * 256))[i32] : (size_t)m;
                bf16_t *dst = dbase + ridx * rstride;
                *(u32x4 *)dst = __builtin_bit_cast(u32x4, o0);
                *(u32x4 *)(dst + 8) = __builtin_bit_cast(u32x4, o1);
            }
            if ((wid & 3) == 0) DA_WTICK(wid >> 2, i, 2);
            continue;
        }
        // D^T[col][node]: lane (node = i32, half) holds columns 8 j + 4 half + (0..3) -> strip[node][col] fp32
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *(f32x4 *)(strip + i32 * SROW + (8 * j + 4 * half) * 4) = (f32x4){acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
        const int row0 = (t0 + i) * 32;
        const int32_t *sl = (const int32_t *)(slots + (i % NSTG) * 256);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = rr + 16 * k;
            const f32x4 a = *(const f32x4 *)(strip + row * SROW + ch * 32);
            const f32x4 b = *(const f32x4 *)(strip + row * SROW + ch * 32 + 16);
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)apply_act(v[e] + bz[e], ACT);
            const int m = row0 + row;
            if (m < p.M && !(p.debug & 256)) {
                const size_t ridx = use_slot ? (size_t)sl[row] : (size_t)m;
                *(u32x4 *)(dbase + ridx * rstride) = __builtin_bit_cast(u32x4, o);   // (non-temporal stores: 243 -> 313 us for the four projections, and the attention kernels that read Q / K / V next lose 5 %)
            }
        }
    }
}

template <int KIN, bool QKV, int ACT, bool DIRECT>
__global__ __launch_bounds__(320) void k_gemm_wreg2(GemmParams p, int tiles_per_wg, int ncg, int cpx) {
    constexpr int KS = KIN / 16;                 // k-steps of 16
    constexpr int ROWB = KIN * 2;                // bytes of one A row
    constexpr int TILEB = 32 * ROWB;             // one 32-row tile
    constexpr int NDMA = TILEB / 1024;           // 1 KB DMA instructions per tile
    constexpr int CPR = ROWB / 16;               // 16-byte chunks per row (32 / 16)
    constexpr int NSTG = DA_WREG_NSTG;
    constexpr int SROW = 272;                    // strip row: 64 fp32 + 16 B pad
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *ring = smem;                                  // [NSTG][TILEB]
    unsigned char *slots = smem + NSTG * TILEB;                  // [NSTG][256 B]: padded-row slot of the tile's 32 nodes
    unsigned char *strips = slots + NSTG * 256;                  // [4][32][SROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nrt = (p.M + 31) >> 5;
    // XCD-aware map (workgroup L runs on XCD L % 8): the ncg column groups that walk the SAME rows sit on one XCD, so a tile of A
    // comes over the fabric once per row chunk and the other ncg - 1 readers hit that XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cgi = slot % ncg, chunk = xcd * cpx + slot / ncg;
    const int t0 = chunk * tiles_per_wg, t1 = min(t0 + tiles_per_wg, nrt);
    const int ntile = t1 - t0;
    if (ntile <= 0) return;

    if (wid == 4) {
        wreg_producer<KIN, QKV, NSTG>(p, ring, slots, t0, ntile, lane);
        return;
    }

    // ---------------------------------------------------- consumer waves
    const int i32 = lane & 31, half = lane >> 5;
    const int col0 = cgi * 256 + wid * 64;
    const bool active = col0 < p.Nout;
    // this wave's 32 columns of W as A-operand fragments: lane (col, half), k-step s -> W[col][16 s + 8 half ..+8]
    // DIRECT (see k_gemm_wreg): MFMA row 8 j + 4 h + i of column block cb is fed W column col0 + 32 h + 16 cb + 4 j + i, so lane
    // (node, half) ends up with the 32 consecutive columns col0 + 32 half + (0 .. 31) of its node -- 64 bytes, four 16-byte stores
    u32x4 wf[2][KS];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int wsel = DIRECT ? 32 * ((i32 >> 2) & 1) + 16 * cb + 4 * (i32 >> 3) + (i32 & 3) : cb * 32 + i32;
        const char *wrow = (const char *)p.W + (size_t)min(col0 + wsel, p.Nout - 1) * (size_t)p.ldw * 2 + half * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[cb][s] = *(const u32x4 *)(wrow + s * 32);
    }
    // row-major side of the epilogue: this lane stores the 16-byte chunk `ch` (8 columns) of rows rr and rr + 16
    // (DIRECT: its node's columns colc .. colc + 31)
    const int ch = lane & 7, rr = lane >> 3;
    const int colc = DIRECT ? col0 + 32 * half : col0 + 8 * ch;
    float bz[8];
    f32x16w bzv, bzv2;
    if (DIRECT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            bzv[e] = (p.bias && active) ? p.bias[min(colc + e, p.Nout - 1)] : 0.f;
            bzv2[e] = (p.bias && active) ? p.bias[min(colc + 16 + e, p.Nout - 1)] : 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bz[e] = (p.bias && active) ? p.bias[min(colc + e, p.Nout - 1)] : 0.f;
    }
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
    unsigned char *strip = strips + wid * (32 * SROW);
    // wait for the W / bias loads here, once: inside the loop this wave only ever issues stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Software pipeline (one consumer wave per SIMD: nobody else hides this wave's epilogue): the MFMAs of tile i are issued
    // first and run on the matrix pipe while the wave converts and stores tile i - 1 out of its LDS strip; only then are
    // the accumulators of tile i written to the strip.  The padded-row slots of a tile are copied to registers when it
    // lands (the producer refills that ring entry one barrier later).
    int slot_prev[4] = {0, 0, 0, 0}, slot_cur[4] = {0, 0, 0, 0};
    auto drain = [&](int it) {                                    // strip (tile `it`) -> bf16 -> global
        const int row0 = (t0 + it) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = rr + 8 * k;
            const f32x4 a = *(const f32x4 *)(strip + row * SROW + ch * 32);
            const f32x4 b = *(const f32x4 *)(strip + row * SROW + ch * 32 + 16);
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)apply_act(v[e] + bz[e], ACT);
            const int m = row0 + row;
            if (m < p.M && !(p.debug & 256)) {
                const size_t ridx = use_slot ? (size_t)slot_prev[k] : (size_t)m;
                *(u32x4 *)(dbase + ridx * rstride) = __builtin_bit_cast(u32x4, o);   // (non-temporal stores: 243 -> 313 us for the four projections, and the attention kernels that read Q / K / V next lose 5 %)
            }
        }
    };
    for (int i = 0; i < ntile; ++i) {
        __builtin_amdgcn_s_barrier();                             // producer: tile i has landed
        if (!active) continue;
        const unsigned char *buf = ring + (i % NSTG) * TILEB + i32 * ROWB;
        if (QKV && !DIRECT) {
            const int32_t *sl = (const int32_t *)(slots + (i % NSTG) * 256);
#pragma unroll
            for (int k = 0; k < 4; ++k) slot_cur[k] = sl[rr + 8 * k];
        }
        f32x16w acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
        // x fragments one group of four k-steps ahead of their MFMAs (a dependent 16-byte LDS read in front of every
        // MFMA leaves the matrix pipe idle for the read latency); sched_barrier keeps hipcc from sinking the reads again
        constexpr int PF = 4, NG = KS / PF;
        u32x4 xa[2][PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) xa[0][s] = *(const u32x4 *)(buf + (((2 * s + half) ^ (i32 & 15)) << 4));
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) {
#pragma unroll
                for (int s = 0; s < PF; ++s)
                    xa[(gq + 1) & 1][s] = *(const u32x4 *)(buf + (((2 * ((gq + 1) * PF + s) + half) ^ (i32 & 15)) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[0][gq * PF + s]),
                                                              __builtin_bit_cast(bf16x8, xa[gq & 1][s]), acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[1][gq * PF + s]),
                                                               __builtin_bit_cast(bf16x8, xa[gq & 1][s]), acc2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DIRECT) {
            bf16x8 o[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[0][e] = (__bf16)apply_act(acc[e] + bzv[e], ACT);
                o[1][e] = (__bf16)apply_act(acc[8 + e] + bzv[8 + e], ACT);
                o[2][e] = (__bf16)apply_act(acc2[e] + bzv2[e], ACT);
                o[3][e] = (__bf16)apply_act(acc2[8 + e] + bzv2[8 + e], ACT);
            }
            const int m = (t0 + i) * 32 + i32;
            if (m < p.M && !(p.debug & 256)) {
                const size_t ridx = use_slot ? (size_t)((const int32_t *)(slots + (i % NSTG) * 256))[i32] : (size_t)m;
                bf16_t *dst = dbase + ridx * rstride;
#pragma unroll
                for (int e = 0; e < 4; ++e) *(u32x4 *)(dst + 8 * e) = __builtin_bit_cast(u32x4, o[e]);
            }
            continue;
        }
        if (i > 0) drain(i - 1);                                  // under the matrix pipe's work on tile i
        __builtin_amdgcn_sched_barrier(0);
        // D^T[col][node]: lane (node = i32, half) holds columns 8 j + 4 half + (0..3) -> strip[node][col] fp32
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(f32x4 *)(strip + i32 * SROW + (8 * j + 4 * half) * 4) = (f32x4){acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
            *(f32x4 *)(strip + i32 * SROW + 128 + (8 * j + 4 * half) * 4) = (f32x4){acc2[4 * j], acc2[4 * j + 1], acc2[4 * j + 2], acc2[4 * j + 3]};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) slot_prev[k] = slot_cur[k];
    }
    if (!DIRECT && active && ntile > 0) drain(ntile - 1);
}

static bool wreg_disabled() { return DA_XENV("DA_DISABLE_WREG", 0) != 0; }

// returns 0 = launched, -1 = not applicable (caller falls back to the A-stationary / generic kernels)
int launch_gemm_wreg(int prec, const GemmParams &p0, const QkvScatter *qs, int act, hipStream_t st) {
    if (wreg_disabled() || prec != DA_PREC_BF16) return -1;
    GemmParams p = p0;
    // measured (28 800 rows): faster than the A-stationary kernel from ~1100 output columns up (K = 256: 3456 columns 77 vs 90 us,
    // 4608: 97 vs 119; 1024: 39 vs 36 -- too few column groups to fill the chip), K = 128 x 1152: 23.5 vs 26 us
    // Two variants.  k_gemm_wreg: 8 consumer waves x 32 columns (two waves per SIMD cover each other's epilogues): the wide
    // projections (Nout >= 1100: conv 3, 110 vs 115 us).  k_gemm_wreg2: 4 consumer waves x 64 columns -- every x fragment
    // read from LDS feeds two MFMAs, half the LDS reads and barriers per FLOP -- for 512 <= Nout < 1100, where it replaces
    // the A-stationary kernel (57 600 rows: K = 256 x 1024 columns 70.5 -> 48.9 us, K = 128 x 1024 53.6 -> 38.6 us; the
    // A-stationary kernel re-streams all of W through LDS for every 128-row panel).  DA_WREG2=0 / =1 force one of them.
    const int v2mode = DA_XENV("DA_WREG2", -1);
    const bool v2 = v2mode == 1 || (v2mode == -1 && p.Nout < 1100);
    if ((p.K != 128 && p.K != 256) || p.pre || p.res || p.M < 4096 || (p.Nout & (v2 ? 63 : 31)) || p.Nout < (v2 ? 512 : 1100)) return -1;
    // register-direct epilogue (DA_WREG_DIRECT=1; default the LDS-strip epilogue: 125 vs 139 us on conv 3, whose 64-byte row pieces it merges): a lane's 16 (32) consecutive columns must stay
    // inside one column block and one head
    const int dmode = DA_XENV("DA_WREG_DIRECT", 0);
    const int lw = v2 ? 32 : 16;
    bool direct = dmode == 1;
    if (qs && ((qs->HC % lw) || (qs->C % lw) || (qs->Cv % lw))) direct = false;
    if (qs) {
        // a wave's 16-byte output chunk (8 columns) must not straddle a column block or a head
        if ((qs->HC & 31) || (qs->C & 7) || (qs->Cv & 7) || act != DA_ACT_NONE) return -1;
    } else if ((p.ldo & 7) || (((size_t)p.out) & 15)) {
        return -1;
    }
    const int nrt = (p.M + 31) / 32;
    const int ncg = (p.Nout + 255) / 256;
    // row chunks: cpx per XCD (32 CUs each, one workgroup per CU), every chunk crossed with all ncg column groups on that XCD
    int cpx = 32 / ncg;
    cpx = cpx < 1 ? 1 : cpx;
    while (cpx > 1 && (size_t)8 * (cpx - 1) >= (size_t)nrt) --cpx;
    const int nchunk = 8 * cpx;
    const int tiles = (nrt + nchunk - 1) / nchunk;
    const dim3 grid((unsigned)(nchunk * ncg));
#define DA_WREG(KK, QQ, AA)                                                                                              \
    do {                                                                                                                  \
        if (direct) DA_WREG_D(KK, QQ, AA, true); else DA_WREG_D(KK, QQ, AA, false);                                       \
    } while (0)
#define DA_WREG_D(KK, QQ, AA, DD)                                                                                         \
    do {                                                                                                                  \
        if (v2) {                                                                                                         \
            constexpr int lds2 = DA_WREG_NSTG * 32 * KK * 2 + DA_WREG_NSTG * 256 + 4 * 32 * 272;                          \
            static bool attr2 = false;                                                                                    \
            if (!attr2) {                                                                                                 \
                DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_gemm_wreg2<KK, QQ, AA, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2)); \
                attr2 = true;                                                                                             \
            }                                                                                                             \
            k_gemm_wreg2<KK, QQ, AA, DD><<<grid, 320, lds2, st>>>(p, tiles, ncg, cpx);                                                  \
            break;                                                                                                        \
        }                                                                                                                 \
        constexpr int lds = DA_WREG_NSTG * 32 * KK * 2 + DA_WREG_NSTG * 256 + 8 * 32 * 144;                                                     \
        static bool attr = false;                                                                                         \
        if (!attr) {                                                                                                      \
            DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_gemm_wreg<KK, QQ, AA, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            attr = true;                                                                                                  \
        }                                                                                                                 \
        k_gemm_wreg<KK, QQ, AA, DD><<<grid, 576, lds, st>>>(p, tiles, ncg, cpx);                                                        \
    } while (0)
    if (qs) {
        if (p.K == 256) DA_WREG(256, true, DA_ACT_NONE); else DA_WREG(128, true, DA_ACT_NONE);
    } else if (act == DA_ACT_GELU) {
        if (p.K == 256) DA_WREG(256, false, DA_ACT_GELU); else DA_WREG(128, false, DA_ACT_GELU);
    } else if (act == DA_ACT_NONE) {
        if (p.K == 256) DA_WREG(256, false, DA_ACT_NONE); else DA_WREG(128, false, DA_ACT_NONE);
    } else {
        return -1;
    }
#undef DA_WREG
#undef DA_WREG_D
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
