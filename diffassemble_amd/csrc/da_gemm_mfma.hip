// MFMA linear kernel: out = act(A[M,K] @ W[Nout,K]^T + bias) (+ residual), fp32 accumulate.
//
// Replaces torch.nn.Linear on the path (mlp, the fused Q|K|V|skip projections of the four
// TransformerConv layers, the pose-head hidden layer).  One kernel template for both act dtypes:
//   bf16: v_mfma_f32_16x16x32_bf16  (one MFMA per 64-byte K chunk)
//   fp32: v_mfma_f32_16x16x4_f32    (exact fp32; four MFMAs per 64-byte K chunk, the K index is
//         permuted consistently for both operands so each lane still reads 16 contiguous bytes)
// Tile 128 (rows of A) x 128 (rows of W) x 128 BYTES of K; 4 waves as 2x2, each 64x64 = 4x4 MFMA
// tiles.  Both operands are K-contiguous, so A and W tiles share one LDS layout: 128 rows x 128 B,
// 16-byte chunks XOR-swizzled by (row & 7) -- conflict-free for the ds_write_b128 staging and for
// the ds_read_b128 fragment reads (lane groups of 16 hit 16 distinct 16-B slots of the 256-B bank
// row).  Global->LDS is LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write): two
// stages, the DMA of tile t+1 is in flight under the MFMAs of tile t, one barrier per K tile.  The
// DMA destination is lane-linear, so the swizzle is applied to the per-lane SOURCE address.
//
// Epilogue.  MFMA is issued as (W fragment) x (A fragment), so a lane owns ONE row of `out` and FOUR
// consecutive output features: bias/activation are applied in registers, the tile is staged through
// LDS and leaves as coalesced 16-byte stores of whole rows.  In QKV mode (dense block-diagonal attention) the fused
// projection is scattered straight into the layouts the attention kernel consumes:
//   Q, K, V -> [H][n_pad][C]   head-major, rows at the graph's padded offset (row_map)
//   skip    -> [M][H*C]        row-major
// (V used to leave TRANSPOSED for the attention's PV operand; its 64..256-byte runs at unaligned offsets
// were partial-line writes that cost ~90 us per step -- the attention now transposes on the LDS read
// side with ds_read_b64_tr_b16 instead.)
#include <stdlib.h>

#include "da_gemm_common.h"

namespace da {

// Padded-row positions of the rows this thread stores in the epilogue, fetched ONCE per workgroup (a
// workgroup keeps its 128 rows for all its column tiles): a row_map load inside the epilogue is a
// dependent global load on the critical path of every tile (measured: +60 % on the conv-3 projection).
template <int PASSES, int NIT>
struct RowSlots {
    int q[PASSES][NIT];     // Q/K/V tiles: slot of row (tid + 256 it) / CPR + pass * ROWS
};

template <typename T, int ACT, int PASSES, int NIT>
__device__ __forceinline__ void mfma_epilogue(const GemmParams &p, const f32x4 (&acc)[4][4], const float (&bz)[4][4],
                                              unsigned char *stg, const RowSlots<PASSES, NIT> &rs, int row0, int col0,
                                              int which, int wm, int wn, int lane, int tid) {
    constexpr int ES = (int)sizeof(T), EPC = 16 / ES;          // elements per 16-byte chunk
    constexpr int RSO = 128 * ES + 16;                          // padded LDS row of the staged tile
    constexpr int ROWS = ES == 4 ? 32 : 64;                     // staged rows per pass (fits one ring slot)
    constexpr int CPR = 128 * ES / 16;                          // 16-byte chunks per staged row
    static_assert(ROWS * RSO <= 32768 && PASSES == 128 / ROWS && NIT == ROWS * CPR / 256, "epilogue staging");
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        dma_barrier();                                        // slot free / previous pass read out
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int rg = wm * 64 + mi * 16 + (lane & 15);  // row of the 128 x 128 image
                if (rg / ROWS != pass) continue;                // wave-uniform (16-row groups never straddle)
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r] + bz[ni][r];
                if (p.pre) {
                    const int m = row0 + wm * 64 + mi * 16 + (lane & 15);
                    const int f0 = col0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
                    if (m < p.M && f0 + 3 < p.Nout) {
                        float pp[4];
                        load4((const T *)p.pre + (size_t)m * p.Nout + f0, pp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += pp[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], ACT);
                if (p.res) {
                    const int m = row0 + wm * 64 + mi * 16 + (lane & 15);
                    const int f0 = col0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
                    if (m < p.M && f0 + 3 < p.Nout) {
                        float rr[4];
                        load4((const T *)p.res + (size_t)m * p.ldo + f0, rr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rr[r];
                    }
                }
                store4((T *)(stg + (rg % ROWS) * RSO) + wn * 64 + ni * 16 + (lane >> 4) * 4, v);
            }
        }
        dma_barrier();
        u32x4 val[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, row = idx / CPR, ch = idx - row * CPR;
            val[it] = *(const u32x4 *)(stg + row * RSO + ch * 16);
        }
        if (p.debug & 1) continue;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, row = idx / CPR, ch = idx - row * CPR;
            int m = row0 + row + pass * ROWS;
            if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(m));          // (see `slot` below)
            const int col = col0 + ch * EPC;
            if (m >= p.M || col >= p.Nout) continue;
            T *dst;
            int slot = rs.q[pass][it];
            // (fp32: the compiler hoists the 64-bit row offsets of all 16 slots out of the column-tile loop and then spills them -- 26 registers of
            //  the Q | K | V scatter instance; opaque here, they are recomputed per tile: three integer operations)
            if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(slot));
            if (!p.qkv) dst = (T *)p.out + (size_t)m * p.ldo + col;
            else if (which == 3) dst = (T *)p.S + (size_t)m * p.HC + (col - 3 * p.HC);
            else {
                const int f = col - which * p.HC;
                if (which == 2 && p.Cv > 0) {
                    const int h = (int)__umulhi((unsigned)f, p.Cvmagic), c = f - h * p.Cv;
                    dst = (T *)p.Vt + ((size_t)h * p.n_pad + slot) * p.Cv + c;
                } else {
                const int h = (int)__umulhi((unsigned)f, p.Cmagic), c = f - h * p.C;
                dst = (T *)(which == 0 ? p.Q : (which == 1 ? p.Kb : p.Vt)) + ((size_t)h * p.n_pad + slot) * p.C + c;
                }
            }
            *(u32x4 *)dst = val[it];
        }
    }
}

// TO: element type of out / res / pre (default: the operand type T).  T = float + BFC with TO = bf16_t, and T = bf16_t with
// TO = float, are the two mixed forms of the training path's bf16 projection buffers (da_train.hip, q16 mode).
template <typename T, bool QKV, int ACT, bool BFC = false, typename TO = T>
__global__ __launch_bounds__(256, 2) void k_gemm_mfma(GemmParams p) {
    // two stages x (A tile 16 KB + W tile 16 KB), filled by LDS-DMA (global_load_lds_dwordx4).  A
    // workgroup owns one 128-row tile of A and walks `nt` consecutive 128-column tiles of W as ONE
    // continuous stream of K stages, so the DMA of the next column tile's first stage is already in
    // flight while the current tile's epilogue runs (the epilogue stages through the ring slot that
    // was consumed last).  In QKV mode the walk covers Q | K | V | skip column blocks in one launch.
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * 128 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (row tile, column group).  With a long reduction (K = 1152) a row tile of A is 295 KB and
    // every column group re-streams it: the groups of one row tile must therefore run on the SAME XCD at
    // the same time so that only the first of them goes to HBM (measured before: FETCH_SIZE = 8.5x the
    // size of A on the conv-0 projection).  Workgroups are dispatched round-robin over the 8 XCDs, so
    // XCD x = id % 8 takes row tiles x, x + 8, ... and walks (row tile, column group) pairs column-fastest.
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_groups > 0) {
        const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
        by = x + 8 * (s / p.xcd_groups);
        bx = s % p.xcd_groups;
        if (by * 128 >= p.M) return;
    }
    const int row0 = by * 128;
    constexpr int ES = (int)sizeof(T), EPC = 16 / ES, BK = 128 / ES;
    const int zs = p.ksplit > 1 ? (int)blockIdx.y : 0;                       // reduction split (XCD-aware 1-D grid: blockIdx.y is free)
    const size_t koff = (size_t)zs * p.kchunk * ES;
    constexpr int EO = (int)sizeof(TO);
    constexpr int ROWS = EO == 4 ? 32 : 64, PASSES = 128 / ROWS, CPR = 128 * EO / 16, NIT = ROWS * CPR / 256;
    const int t_beg = bx * p.nt, t_end = min(t_beg + p.nt, p.nct);

    // Staging by LDS-DMA: wave w fills rows [32w, 32w+32) of both tiles, 8 rows (1 KB) per instruction.
    // The DMA writes lane l at (wave-uniform base) + 16 l, i.e. row (l >> 3), slot (l & 7); the XOR
    // swizzle is therefore applied on the SOURCE: slot s of row r holds logical chunk s ^ (r & 7).
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    const char *Wb = (const char *)p.W + lc * 16 + koff;
    const size_t ldaB = (size_t)p.lda * sizeof(T), ldwB = (size_t)p.ldw * sizeof(T);
    const char *ap[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        ap[j] = (const char *)p.A + lc * 16 + koff + (size_t)min(row0 + 32 * wid + 8 * j + lr, p.M - 1) * ldaB;
    const int nk = (p.ksplit > 1 ? p.kchunk : p.K) / BK;
    const int S = (t_end - t_beg) * nk;                       // stages of this workgroup
    auto issue = [&](int s) {
        const int ti = s / nk, kt = s - ti * nk;
        const int c0 = (t_beg + ti) * 128;
        unsigned char *sa = smem + (s & 1) * 32768 + (32 * wid) * 128, *sw = sa + 16384;
        const size_t kb = (size_t)kt * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char *wsrc = Wb + (size_t)min(c0 + 32 * wid + 8 * j + lr, p.Nout - 1) * ldwB + kb;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ap[j] + kb),
                                             (__attribute__((address_space(3))) void *)(sa + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)wsrc,
                                             (__attribute__((address_space(3))) void *)(sw + j * 1024), 16, 0, 0);
        }
    };
    if (S > 0 && !(p.debug & 4)) issue(0);

    RowSlots<PASSES, NIT> rs;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass)
#pragma unroll
        for (int it = 0; it < NIT; ++it) rs.q[pass][it] = 0;
    if (QKV) {
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int m = row0 + (tid + it * 256) / CPR + pass * ROWS;
                if (m < p.M) rs.q[pass][it] = p.row_map[m];
            }
    }

    for (int ti = 0; ti < t_end - t_beg; ++ti) {
        const int col0 = (t_beg + ti) * 128;
        const int which = QKV ? col0 / p.HC : 0;
        // bias of this lane's output features, fetched under the MFMAs (the epilogue wants them in
        // registers: dependent scalar loads there cost microseconds per tile)
        float bz[4][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int f0 = col0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
            if (p.bias && f0 + 3 < p.Nout) { const f32x4 b4 = *(const f32x4 *)(p.bias + f0); bz[ni][0] = b4[0]; bz[ni][1] = b4[1]; bz[ni][2] = b4[2]; bz[ni][3] = b4[3]; }
            else { bz[ni][0] = bz[ni][1] = bz[ni][2] = bz[ni][3] = 0.f; }
        }
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt) {
            const int s = ti * nk + kt;
            dma_barrier();                   // own DMA landed (vmcnt(0)) + everyone done with the other slot
            if (s + 1 < S && !(p.debug & 4)) issue(s + 1);
            const unsigned char *sA = smem + (s & 1) * 32768;
            if (!(p.debug & 2)) mma_block<T, BFC>(sA, sA + 16384, wm, wn, lane, acc);
        }

        // -------------------------------------------------------------- epilogue of this column tile
        // Registers -> LDS (bias / activation / residual applied on the way) -> coalesced 16-byte global
        // stores of whole tile rows.  The accumulator layout gives a lane 4 consecutive features of one
        // row (or, for V columns, 4 consecutive nodes of one feature): 8-byte pieces scattered over 16
        // rows per instruction, ~1 TB/s if written directly.  Staging area = the ring slot just consumed.
        unsigned char *stg = smem + ((ti * nk + nk - 1) & 1) * 32768;
        if (p.ksplit > 1) {                                   // this split's partial image (bias / residual are added by the reduction)
            GemmParams q = p;
            q.out = (char *)p.out + (size_t)zs * p.M * p.ldo * EO;
            mfma_epilogue<TO, ACT, PASSES, NIT>(q, acc, bz, stg, rs, row0, col0, which, wm, wn, lane, tid);
        } else
        mfma_epilogue<TO, ACT, PASSES, NIT>(p, acc, bz, stg, rs, row0, col0, which, wm, wn, lane, tid);
    }
}

static bool aligned16(const void *p) { return (((size_t)p) & 15) == 0; }

int launch_gemm_astat(int prec, const GemmParams &p0, const QkvScatter *qs, int act, hipStream_t st);   // da_gemm_astat.hip
int launch_gemm_wreg(int prec, const GemmParams &p0, const QkvScatter *qs, int act, hipStream_t st);    // da_gemm_wreg.hip
int launch_gemm_xpanel(int prec, const GemmParams &p0, const QkvScatter *qs, int act, const void *wpacked, hipStream_t st);   // da_gemm_xpanel.hip
int launch_gemm_thin(int prec, const GemmParams &p0, const QkvScatter *qs, int act, hipStream_t st);    // da_gemm_thin.hip

// DA_GEMM_THIN=1 (gemm_thin_set: the same switch at run time, tools/corun_probe.hip): tall short-reduction products take the co-resident
// four-wave kernel of da_gemm_thin.hip
static int g_thin = -1;
void gemm_thin_set(int v) { g_thin = v; }
[[maybe_unused]] static bool thin_on(int Nout) {          // 1 = every tall projection, 2 = the 1024-column ones only (the folded last layer keeps its kernel)
    if (g_thin < 0) g_thin = DA_XENV("DA_GEMM_THIN", 0);
    return g_thin == 1 || (g_thin == 2 && Nout < 1100);
}

// returns 0 = launched, -1 = shape not supported by this kernel (caller falls back), >0 error
int launch_gemm_mfma(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias,
                     int act, const void *res, void *out, int ldo, const QkvScatter *qs, hipStream_t st, int ldw,
                     const void *pre, const void *wpacked) {
    // DA_PREC_F32_BF16MMA (training): fp32 storage, operands rounded to bf16 inside the matrix-core kernel (Mma16<float, true>)
    const bool bfc = prec == DA_PREC_F32_BF16MMA;
    if (bfc && (qs || act != DA_ACT_NONE || pre)) return -1;
    const int es = (int)esize(prec), BK = 128 / es;
    if (M <= 0 || Nout <= 0) return 0;
    if (ldw <= 0) ldw = K;
    if (K % BK != 0 || (Nout % (16 / es)) || !aligned16(A) || !aligned16(W) || ((size_t)lda * es) % 16 != 0 ||
        ((size_t)ldw * es) % 16 != 0 || (pre && (!aligned16(pre) || qs))) return -1;
    GemmParams p;
    p.M = M; p.K = K; p.Nout = Nout; p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.act = act; p.res = res;
    p.ldw = ldw; p.pre = pre; p.Cv = 0; p.Cvmagic = 0;
    p.out = out; p.ldo = ldo; p.qkv = 0; p.HC = 1; p.C = 1; p.n_pad = 0; p.row_map = nullptr; p.Cmagic = 0;
    p.xcd_groups = 0;
    p.ksplit = 0; p.kchunk = 0;
    p.Q = p.Kb = p.Vt = p.S = nullptr;
    { const char *e = DA_XENV_LIVE("DA_GEMM_PROF_PTR"); p.prof = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
    { const char *e = DA_XENV_LIVE("DA_GEMM_DEBUG"); p.debug = e ? atoi(e) : 0; }
    if (qs) {
        const int nexp = qs->blocks > 0 ? qs->blocks * qs->HC : (qs->Cv > 0 ? 2 * qs->HC + (qs->HC / qs->C) * qs->Cv : 4 * qs->HC);
        if (qs->HC % 128 != 0 || (qs->C & 7) || (qs->Cv & 7) || Nout != nexp || Nout % 128 != 0 || act != DA_ACT_NONE || res) return -1;
        p.qkv = 1; p.HC = qs->HC; p.C = qs->C; p.n_pad = qs->n_pad; p.row_map = qs->row_map;
        p.Cv = qs->Cv;
        p.Cvmagic = qs->Cv > 0 ? (unsigned)((((unsigned long long)1 << 32) + (unsigned)qs->Cv - 1) / (unsigned)qs->Cv) : 0;
        p.Cmagic = (unsigned)((((unsigned long long)1 << 32) + (unsigned)qs->C - 1) / (unsigned)qs->C);
        p.Q = qs->Q; p.Kb = qs->K; p.Vt = qs->Vt; p.S = qs->S;
    } else if (((size_t)ldo * es) % 16 != 0 || !aligned16(out) || (res && !aligned16(res))) {
        return -1;
    }
    if (!bfc) {   // short reductions: A-stationary kernel (da_gemm_astat.hip)
        const bool off = DA_XENV("DA_DISABLE_ASTAT", 0) != 0;
#ifdef DA_EXPERIMENTS
        if (K * es <= 512 && thin_on(Nout)) {          // the co-resident four-wave kernel (da_gemm_thin.hip): measured, lost (profiles/r06)
            const int rt = launch_gemm_thin(prec, p, qs, act, st);
            if (rt >= 0) return rt;
        }
#endif
        if (K * es <= 512 && wpacked) {
            // tall inputs at the benched batch sizes: row panel of A in LDS, pre-packed W fragments double-buffered in registers
            const int rx = launch_gemm_xpanel(prec, p, qs, act, wpacked, st);
            if (rx >= 0) return rx;
        }
        if (K * es <= 512) {
            // tall inputs: W in registers, A tiles streamed by a producer wave (da_gemm_wreg.hip)
            const int rw = launch_gemm_wreg(prec, p, qs, act, st);
            if (rw >= 0) return rw;
        }
        if (!off && K * es <= 512) {
            const int rc = launch_gemm_astat(prec, p, qs, act, st);
            if (rc >= 0) return rc;
        }
    }
    const int nrt = (M + 127) / 128;
    // Column tiles per workgroup: 256 CUs x 2 resident workgroups = 512 slots.  Use as many column
    // groups as keep the whole grid co-resident (one round, no tail) -- each workgroup then streams
    // its share of the column tiles back to back.
    const bool xcd_off = DA_XENV("DA_GEMM_NO_XCD_MAP", 0) != 0;
    auto plan2 = [&](int nct) {
        p.xcd_groups = 0;
        if (!xcd_off && nct > 1 && (size_t)K * es > 512) {
            // long reduction: one column tile per workgroup, column groups of a row tile co-resident on one XCD
            p.nct = nct; p.nt = 1; p.xcd_groups = nct;
            return dim3((unsigned)(8 * ((nrt + 7) / 8) * nct), 1u);
        }
        int groups = 512 / nrt;
        groups = groups > nct ? nct : (groups < 1 ? 1 : groups);
        int ntile = (nct + groups - 1) / groups;
        groups = (nct + ntile - 1) / ntile;
        p.nct = nct; p.nt = ntile;
        return dim3((unsigned)groups, (unsigned)nrt);
    };
#define DA_GEMM_LAUNCH(TT, VO, AC, GRID) k_gemm_mfma<TT, VO, AC><<<GRID, 256, 0, st>>>(p)
#define DA_GEMM_ACT(TT, GRID)                                            \
    do {                                                                  \
        if (act == DA_ACT_GELU) DA_GEMM_LAUNCH(TT, false, DA_ACT_GELU, GRID);        \
        else if (act == DA_ACT_LEAKY02) DA_GEMM_LAUNCH(TT, false, DA_ACT_LEAKY02, GRID); \
        else DA_GEMM_LAUNCH(TT, false, DA_ACT_NONE, GRID);                \
    } while (0)
    if (!qs) {
        const dim3 grid = plan2((Nout + 127) / 128);
        if (prec == DA_PREC_BF16) DA_GEMM_ACT(bf16_t, grid);
        else if (bfc) k_gemm_mfma<float, false, DA_ACT_NONE, true><<<grid, 256, 0, st>>>(p);
        else DA_GEMM_ACT(float, grid);
    } else {
        const dim3 g = plan2(Nout / 128);                        // Q | K | V (| skip) column blocks, one launch
        if (prec == DA_PREC_BF16) DA_GEMM_LAUNCH(bf16_t, true, DA_ACT_NONE, g);
        else DA_GEMM_LAUNCH(float, true, DA_ACT_NONE, g);
    }
#undef DA_GEMM_ACT
#undef DA_GEMM_LAUNCH
    DA_LAUNCH_CHECK();
    return 0;
}

// out[m][c] = sum_z partial[z][m][c] (+ bias[c]) (+ res[m][c]): fixed order
// gelu_pre (backward: the product is dX of a Linear that follows a GELU): out = sum * gelu'(gelu_pre) -- the separate k_gelu_bwd launch
// of the training path folded in (a dependent ~5 us launch per site; same arithmetic, same order).  act_out (forward: the Linear
// feeds a GELU): out = sum (the pre-activation backward needs) and act_out = gelu(sum).  Both share out's leading dimension.
__global__ __launch_bounds__(256) void k_splitk_reduce(int splits, int M, int N, const float *__restrict__ partial,
                                                       const float *__restrict__ bias, const float *res, float *out, int ldo,
                                                       const float *gelu_pre, float *act_out) {
    const size_t MN = (size_t)M * N, n4 = MN / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = *(const f32x4 *)(partial + 4 * i);
        for (int z = 1; z < splits; ++z) s += *(const f32x4 *)(partial + (size_t)z * MN + 4 * i);
        const size_t m = 4 * i / N, c = 4 * i - m * N;
        if (bias) s += *(const f32x4 *)(bias + c);
        if (res) s += *(const f32x4 *)(res + m * ldo + c);
        if (gelu_pre) {
            const f32x4 pr = *(const f32x4 *)(gelu_pre + m * ldo + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] * gelu_grad(pr[e]);
        }
        *(f32x4 *)(out + m * ldo + c) = s;
        if (act_out) {
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = gelu_erf(s[e]);
            *(f32x4 *)(act_out + m * ldo + c) = a;
        }
    }
}

// Skinny outputs with a long reduction in the bf16-operand training mode (dX of the conv projections: Nout = 256, K = 1024 ...
// 4608 at BASELINE configuration 5): 72 row tiles x 2 column tiles leave 44 % of the CUs without a workgroup.  The reduction is
// cut into `splits` pieces that run as separate workgroups of ONE launch (partial images [split][M][Nout], fp32) and a second
// kernel adds them in a fixed order together with bias and residual.  Returns -1 when the shape does not qualify.
int launch_gemm_mfma_splitk(int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias, const void *res,
                            void *out, int ldo, float *partial, size_t partial_floats, hipStream_t st, bool in16, const float *gelu_pre,
                            float *act_out) {
    const int es = in16 ? 2 : 4;                             // in16: A and W are bf16 (out / res / partial stay fp32)
    const bool off = DA_XENV("DA_GEMM_SPLITK", 1) == 0;
    const int nrt = (M + 127) / 128, nct = (Nout + 127) / 128;
    if (off || !partial || 8 * ((nrt + 7) / 8) * nct > 320 || K < 1024 || K % 128 != 0 || Nout % 4 != 0 || ldo % 4 != 0 ||
        !aligned16(A) || !aligned16(W) || !aligned16(out) || (res && !aligned16(res)) || (bias && !aligned16(bias)) ||
        ((size_t)lda * es) % 16 != 0 || ((size_t)K * es) % 16 != 0)
        return -1;
    // as many splits (<= 4) as keep the launch inside ONE round of the 512 resident workgroups (2 per CU): a second round with a
    // few workgroups costs a whole extra pass over the split's stages
    const int base = 8 * ((nrt + 7) / 8) * nct;
    int smax = DA_XENV("DA_GEMM_SPLITK_MAX", 4);              // (8 measured: 1.204 vs 1.205 ms at configuration 5 -- no gain)
    smax = smax < 2 ? 2 : (smax > 8 ? 8 : smax);
    int splits = smax;
    while (splits > 1 && (K % (splits * (128 / es)) != 0 || (size_t)splits * M * Nout > partial_floats || (base * splits > 512 && splits > 2))) --splits;
    if (splits < 2) return -1;
    GemmParams p;
    p.M = M; p.K = K; p.Nout = Nout; p.A = A; p.lda = lda; p.W = W; p.bias = nullptr; p.act = DA_ACT_NONE; p.res = nullptr;
    p.ldw = K; p.pre = nullptr; p.Cv = 0; p.Cvmagic = 0;
    p.out = partial; p.ldo = Nout; p.qkv = 0; p.HC = 1; p.C = 1; p.n_pad = 0; p.row_map = nullptr; p.Cmagic = 0;
    p.Q = p.Kb = p.Vt = p.S = nullptr;
    p.prof = nullptr; p.debug = 0;
    p.nct = nct; p.nt = 1; p.xcd_groups = nct;
    p.ksplit = splits; p.kchunk = K / splits;
    const dim3 grid((unsigned)(8 * ((nrt + 7) / 8) * nct), (unsigned)splits);
    if (in16) k_gemm_mfma<bf16_t, false, DA_ACT_NONE, false, float><<<grid, 256, 0, st>>>(p);
    else k_gemm_mfma<float, false, DA_ACT_NONE, true><<<grid, 256, 0, st>>>(p);
    const size_t n4 = (size_t)M * Nout / 4;
    k_splitk_reduce<<<(unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256), 256, 0, st>>>(splits, M, Nout, partial, bias,
                                                                                                   (const float *)res, (float *)out, ldo, gelu_pre, act_out);
    DA_LAUNCH_CHECK();
    return 0;
}

// The mixed forms of the training path's q16 mode (plain row-major out, no activation):
//   in16 = false, out16 = true : fp32 A and W (operands rounded to bf16 in registers), bf16 out  (conv projections -> bf16 Q | K | V | skip)
//   in16 = true,  out16 = false: bf16 A and W, fp32 out (+ fp32 res)                           (dX from the bf16 projection gradient)
// Returns -1 when the shape / alignment is not covered.
int launch_gemm_mfma_mixed(bool in16, bool out16, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias,
                           const void *res, void *out, int ldo, hipStream_t st) {
    if (in16 == out16) return -1;
    const int es = in16 ? 2 : 4, eo = out16 ? 2 : 4, BK = 128 / es;
    if (M <= 0 || Nout <= 0) return 0;
    if (K % BK != 0 || (Nout % (16 / eo)) || !aligned16(A) || !aligned16(W) || ((size_t)lda * es) % 16 != 0 || ((size_t)K * es) % 16 != 0 ||
        ((size_t)ldo * eo) % 16 != 0 || !aligned16(out) || (res && !aligned16(res)) || (bias && !aligned16(bias)))
        return -1;
    GemmParams p;
    p.M = M; p.K = K; p.Nout = Nout; p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.act = DA_ACT_NONE; p.res = res;
    p.ldw = K; p.pre = nullptr; p.Cv = 0; p.Cvmagic = 0;
    p.out = out; p.ldo = ldo; p.qkv = 0; p.HC = 1; p.C = 1; p.n_pad = 0; p.row_map = nullptr; p.Cmagic = 0;
    p.Q = p.Kb = p.Vt = p.S = nullptr;
    p.prof = nullptr; p.debug = 0; p.ksplit = 0; p.kchunk = 0; p.xcd_groups = 0;
    const int nrt = (M + 127) / 128, nct = (Nout + 127) / 128;
    dim3 grid;
    if (nct > 1 && (size_t)K * es > 512) {                  // long reduction: column groups of a row tile co-resident on one XCD
        p.nct = nct; p.nt = 1; p.xcd_groups = nct;
        grid = dim3((unsigned)(8 * ((nrt + 7) / 8) * nct), 1u);
    } else {
        int groups = 512 / nrt;
        groups = groups > nct ? nct : (groups < 1 ? 1 : groups);
        const int ntile = (nct + groups - 1) / groups;
        groups = (nct + ntile - 1) / ntile;
        p.nct = nct; p.nt = ntile;
        grid = dim3((unsigned)groups, (unsigned)nrt);
    }
    if (in16) k_gemm_mfma<bf16_t, false, DA_ACT_NONE, false, float><<<grid, 256, 0, st>>>(p);
    else k_gemm_mfma<float, false, DA_ACT_NONE, true, bf16_t><<<grid, 256, 0, st>>>(p);
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
