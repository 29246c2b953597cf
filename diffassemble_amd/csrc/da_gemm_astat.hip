// A-stationary MFMA linear kernel for SHORT reductions (K * sizeof(T) <= 512 bytes: the K = 256
// fused Q|K|V|skip projections of convs 1-3 and mlp.2), where the generic 128x128 kernel re-reads its
// A tile for every column tile and the fused projection's outputs are 4-18x larger than its inputs:
// measured there, TCC hit rate 47 % (64 resident workgroups x 64 KB of A tile = the whole 4 MB L2 of
// an XCD) and ~1 GB of LDS-DMA traffic per launch for 17 MB of compulsory input.
//
// Here a workgroup (8 waves) loads its 128-row A panel ONCE (all of K, <= 64 KB of LDS) and streams
// only W: one 128-column x 64-byte... 128-byte K-chunk tile (16 KB) per stage through a 4-slot ring,
// three stages in flight (counted s_waitcnt vmcnt, never a drain: loads retire in order, so younger
// stores only make the counted wait conservative).  Wave grid 4 (rows) x 2 (columns): 32 x 64 outputs per wave = 2 x 4
// MFMA tiles; same XOR-swizzled 128-byte-row LDS images as da_gemm_mfma.hip.  The epilogue needs no
// workgroup barrier: every wave stages its own 16 x 64 sub-tiles through a private 2.3 KB LDS strip
// and writes 128-byte contiguous row segments.
#include <stdlib.h>

#include "da_gemm_common.h"

namespace da {

#ifdef DA_GEMM_PROBE
#define DA_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define DA_PROBE(...) __VA_ARGS__
#else
#define DA_TICK(var)
#define DA_PROBE(...)
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Wave-private epilogue of one 32 x 64 (per wave) output tile: bias / activation / residual in
// registers, 16-row strips through this wave's own LDS strip, 128-byte contiguous global stores.
// Padded-row slots (row_map) of the rows a lane stores in the epilogue, fetched once per workgroup:
// the rows of an A-stationary workgroup never change, and a row_map load inside the epilogue is a
// dependent global load in front of every store (measured +60 % on the conv-3 projection).
struct AstatSlots {
    int q[2][2];        // Q/K/V tiles: slot of row wm*32 + mi*16 + ((lane + 64 k) >> 3)
};

template <typename T>
__device__ __forceinline__ AstatSlots astat_load_slots(const GemmParams &p, int row0, int wm, int lane) {
    AstatSlots rs;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = row0 + wm * 32 + mi * 16 + ((lane + 64 * k) >> 3);
            rs.q[mi][k] = (p.qkv && m < p.M) ? p.row_map[m] : 0;
        }
    return rs;
}

template <typename T, int ACT>
__device__ __forceinline__ void astat_epilogue(const GemmParams &p, const f32x4 (&acc)[2][4], const float (&bz)[4][4],
                                               unsigned char *stg, const AstatSlots &rs, int row0, int col0,
                                               int which, int wm, int wn, int lane) {
    constexpr int ES = (int)sizeof(T), EPC = 16 / ES;
    // strip = 16 nodes x 128 bytes of features; fp32 needs two passes over the wave's 64 columns
    constexpr int CPP = 128 / ES;                          // columns per pass: 64 (bf16) / 32 (fp32)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ps = 0; ps < 64 / CPP; ++ps) {
#pragma unroll
            for (int nj = 0; nj < CPP / 16; ++nj) {
                const int ni = ps * (CPP / 16) + nj;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r] + bz[ni][r];
                if (p.pre) {
                    const int m = row0 + wm * 32 + mi * 16 + (lane & 15);
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
                    const int m = row0 + wm * 32 + mi * 16 + (lane & 15);
                    const int f0 = col0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
                    if (m < p.M && f0 + 3 < p.Nout) {
                        float rr[4];
                        load4((const T *)p.res + (size_t)m * p.ldo + f0, rr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rr[r];
                    }
                }
                store4((T *)(stg + (lane & 15) * 144) + nj * 16 + (lane >> 4) * 4, v);
            }
            // 16 rows x 8 chunks of 16 bytes: two chunks per lane, 128 contiguous bytes per row
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int id = lane + 64 * k, row = id >> 3, ch = id & 7;
                const u32x4 val = *(const u32x4 *)(stg + row * 144 + ch * 16);
                const int m = row0 + wm * 32 + mi * 16 + row;
                const int col = col0 + wn * 64 + ps * CPP + ch * EPC;
                if (m >= p.M || col >= p.Nout) continue;
                if (p.qkv && (p.debug & (which == 3 ? 64 : 32))) continue;
                T *dst;
                if (!p.qkv) dst = (T *)p.out + (size_t)m * p.ldo + col;
                else if (which == 3) dst = (T *)p.S + (size_t)m * p.HC + (col - 3 * p.HC);
                else {
                    const int f = col - which * p.HC;
                        if (which == 2 && p.Cv > 0) {
                            const int h = (int)__umulhi((unsigned)f, p.Cvmagic), c = f - h * p.Cv;
                            dst = (T *)p.Vt + ((size_t)h * p.n_pad + rs.q[mi][k]) * p.Cv + c;
                        } else {
                        const int h = (int)__umulhi((unsigned)f, p.Cmagic), c = f - h * p.C;
                    dst = (T *)(which == 0 ? p.Q : (which == 1 ? p.Kb : p.Vt)) + ((size_t)h * p.n_pad + rs.q[mi][k]) * p.C + c;
                        }
                }
                *(u32x4 *)dst = val;
            }
        }
    }
}

template <typename T, int ACT>
__global__ __launch_bounds__(512, 2) void k_gemm_astat(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = (int)sizeof(T), EPC = 16 / ES, BK = 128 / ES;
    const int nk = p.K / BK;                                     // 1..4 resident K chunks of A
    unsigned char *sA = smem;                                    // [nk][128 rows][128 B]
    unsigned char *sW = smem + nk * 16384;                       // 4-slot ring of [128 rows][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;                       // 4 x 2 waves: rows 32 wm, cols 64 wn
    unsigned char *stg = smem + nk * 16384 + 65536 + wid * 2304; // private strip: 16 rows x 144 B
    const int row0 = blockIdx.y * 128;
    const AstatSlots rs = astat_load_slots<T>(p, row0, wm, lane);
    const int t_beg = blockIdx.x * p.nt, t_end = min(t_beg + p.nt, p.nct);
    auto colblock = [&](int t) { return t; };
    const int S = (t_end - t_beg) * nk;                          // W stages of this workgroup
    if (S <= 0) return;

    // LDS-DMA lane mapping (lane -> row lr, swizzled 16-byte chunk lc), as in da_gemm_mfma.hip
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    const size_t ldaB = (size_t)p.lda * ES, ldwB = (size_t)p.ldw * ES;
    // A panel: nk x 16 instructions of 8 rows; wave w issues instructions w and w + 8 of every chunk
    {
        const char *a0 = (const char *)p.A + lc * 16 + (size_t)min(row0 + 8 * wid + lr, p.M - 1) * ldaB;
        const char *a1 = (const char *)p.A + lc * 16 + (size_t)min(row0 + 8 * (wid + 8) + lr, p.M - 1) * ldaB;
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a0 + (size_t)kt * 128),
                                             (__attribute__((address_space(3))) void *)(sA + kt * 16384 + wid * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a1 + (size_t)kt * 128),
                                             (__attribute__((address_space(3))) void *)(sA + kt * 16384 + (wid + 8) * 1024), 16, 0, 0);
        }
    }
    const char *Wb = (const char *)p.W + lc * 16;
    auto issue = [&](int s) {                                    // 2 DMA instructions per wave per stage
        const int ti = s / nk, kt = s - ti * nk;
        const int c0 = colblock(t_beg + ti) * 128;
        unsigned char *dst = sW + (s & 3) * 16384;
        const size_t kb = (size_t)kt * 128;
        const char *w0 = Wb + (size_t)min(c0 + 8 * wid + lr, p.Nout - 1) * ldwB + kb;
        const char *w1 = Wb + (size_t)min(c0 + 8 * (wid + 8) + lr, p.Nout - 1) * ldwB + kb;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)w0,
                                         (__attribute__((address_space(3))) void *)(dst + wid * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)w1,
                                         (__attribute__((address_space(3))) void *)(dst + (wid + 8) * 1024), 16, 0, 0);
    };
    DA_TICK(t_start);
    DA_PROBE(unsigned long long c_wait = 0, c_mma = 0, c_epi = 0;)
    issue(0);
    if (S > 1) issue(1);
    if (S > 2) issue(2);

    // bias of this lane's output features, fetched ONE TILE AHEAD so that it is never the youngest
    // outstanding load at a counted wait
    auto load_bias = [&](int ti, float (&bz)[4][4]) {
        const int col0 = colblock(t_beg + ti) * 128;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int f0 = col0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
            if (p.bias && f0 + 3 < p.Nout) { const f32x4 b4 = *(const f32x4 *)(p.bias + f0); bz[ni][0] = b4[0]; bz[ni][1] = b4[1]; bz[ni][2] = b4[2]; bz[ni][3] = b4[3]; }
            else { bz[ni][0] = bz[ni][1] = bz[ni][2] = bz[ni][3] = 0.f; }
        }
    };
    float bz[4][4], bzn[4][4];
    load_bias(0, bz);
    for (int ti = 0; ti < t_end - t_beg; ++ti) {
        const int col0 = colblock(t_beg + ti) * 128;
        const int which = p.qkv ? col0 / p.HC : 0;
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt) {
            const int s = ti * nk + kt;
            // stage s must have landed; stages s+1, s+2 (2 DMA instructions each per wave) may stay in
            // flight.  The bias loads above and the epilogue's stores also count, hence the drains.
            DA_TICK(t_a);
            // Counted wait: loads (incl. LDS-DMA) retire in order, so "at most 2*ahead VMEM ops outstanding"
            // implies stage s has landed no matter how many younger stores / bias loads are pending (they
            // only add to the counter: conservative, never early).  No drain -> the epilogue's stores of
            // the previous tile stay in flight under this tile's MFMAs.
            const int ahead = min(2, S - 1 - s);
            if (ahead == 0) wait_vmcnt<0>();
            else if (ahead == 1) wait_vmcnt<2>();
            else wait_vmcnt<4>();
            if (kt == (nk > 1 ? 1 : 0) && ti + 1 < t_end - t_beg) load_bias(ti + 1, bzn);
            __syncthreads();                     // everyone's share landed; slot (s + 3) & 3 is free again
            DA_TICK(t_b);
            if (s + 3 < S) issue(s + 3);
            const unsigned char *a = sA + kt * 16384, *w = sW + (s & 3) * 16384;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 fa[2], fw[4];
                const int c = kk * 4 + (lane >> 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int R = wm * 32 + t * 16 + (lane & 15);
                    fa[t] = *(const u32x4 *)(a + R * 128 + ((c ^ (R & 7)) << 4));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int R = wn * 64 + t * 16 + (lane & 15);
                    fw[t] = *(const u32x4 *)(w + R * 128 + ((c ^ (R & 7)) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = Mma16<T>::run(fw[ni], fa[mi], acc[mi][ni]);
            }
            DA_PROBE(asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][3])); { DA_TICK(t_c); c_wait += t_b - t_a; c_mma += t_c - t_b; })
        }
        DA_TICK(t_e0);

        astat_epilogue<T, ACT>(p, acc, bz, stg, rs, row0, col0, which, wm, wn, lane);
#pragma unroll
        for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 4; ++b_) bz[a_][b_] = bzn[a_][b_];
        DA_PROBE({ DA_TICK(t_e1); c_epi += t_e1 - t_e0; })
    }
    DA_PROBE(if (p.prof && tid == 0) { DA_TICK(t_end_); unsigned long long *o = p.prof + 4 * (blockIdx.y * gridDim.x + blockIdx.x); o[0] = t_end_ - t_start; o[1] = c_wait; o[2] = c_mma; o[3] = c_epi; })
}

// ---------------------------------------------------------------------------------------------
// Register-staged variant (the default): W tiles travel global -> VGPR -> LDS with plain loads
// issued TA column tiles (TA * NK stages = 128 KB per CU) ahead of their use.  Why: measured with the
// phase probe, LDS-DMA requests take ~2.3 us from issue to landing under load, so the three 16 KB
// stages the LDS budget allows in flight sustain only ~23 GB/s per CU; plain loads keep their data in
// the (much larger) register file instead, and hipcc counts their vmcnt by itself.
template <typename T, bool QKV, int ACT, int NK>
__global__ __launch_bounds__(512, 2) void k_gemm_astat_rs(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = (int)sizeof(T), TA = 2;
    unsigned char *sA = smem;                                    // [NK][128 rows][128 B]
    unsigned char *sW = smem + NK * 16384;                       // 2 slots of [128 rows][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    unsigned char *stg = smem + NK * 16384 + 32768 + wid * 2304;
    const int row0 = blockIdx.y * 128;
    const AstatSlots rs = astat_load_slots<T>(p, row0, wm, lane);
    const int t_beg = blockIdx.x * p.nt, t_end = min(t_beg + p.nt, p.nct);
    const int ntile = t_end - t_beg;
    auto colblock = [&](int t) { return t; };             // QKV mode: one launch walks Q | K | V | skip tiles
    if (ntile <= 0) return;

    // staging role of this thread: 16-byte chunk c of rows r0 and r0 + 64 of a [128][128 B] tile
    const int r0 = tid >> 3, c = tid & 7;
    const int lo0 = r0 * 128 + ((c ^ (r0 & 7)) << 4), lo1 = lo0 + 64 * 128;     // (r0 + 64) & 7 == r0 & 7
    const size_t ldaB = (size_t)p.lda * ES, ldwB = (size_t)p.ldw * ES;
    {   // A panel, once
        const char *a0 = (const char *)p.A + c * 16 + (size_t)min(row0 + r0, p.M - 1) * ldaB;
        const char *a1 = (const char *)p.A + c * 16 + (size_t)min(row0 + r0 + 64, p.M - 1) * ldaB;
        u32x4 ra[NK][2];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) { ra[kt][0] = *(const u32x4 *)(a0 + kt * 128); ra[kt][1] = *(const u32x4 *)(a1 + kt * 128); }
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) { *(u32x4 *)(sA + kt * 16384 + lo0) = ra[kt][0]; *(u32x4 *)(sA + kt * 16384 + lo1) = ra[kt][1]; }
    }
    const char *Wb = (const char *)p.W + c * 16;
    u32x4 wr[TA][NK][2];                                         // TA column tiles of W in flight
    auto wload = [&](int ti, u32x4 (&dst)[NK][2]) {
        if (ti < ntile) {
            const int c0 = colblock(t_beg + ti) * 128;
            const char *w0 = Wb + (size_t)min(c0 + r0, p.Nout - 1) * ldwB;
            const char *w1 = Wb + (size_t)min(c0 + r0 + 64, p.Nout - 1) * ldwB;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) { dst[kt][0] = *(const u32x4 *)(w0 + kt * 128); dst[kt][1] = *(const u32x4 *)(w1 + kt * 128); }
        }
    };
#pragma unroll
    for (int u = 0; u < TA; ++u) wload(u, wr[u]);
    auto load_bias = [&](int ti, float (&bz)[4][4]) {
        const int col0 = colblock(t_beg + min(ti, ntile - 1)) * 128;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int f0 = col0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
            if (p.bias && f0 + 3 < p.Nout) { const f32x4 b4 = *(const f32x4 *)(p.bias + f0); bz[ni][0] = b4[0]; bz[ni][1] = b4[1]; bz[ni][2] = b4[2]; bz[ni][3] = b4[3]; }
            else { bz[ni][0] = bz[ni][1] = bz[ni][2] = bz[ni][3] = 0.f; }
        }
    };
    // stage 0 of tile 0 into slot 0
    *(u32x4 *)(sW + lo0) = wr[0][0][0];
    *(u32x4 *)(sW + lo1) = wr[0][0][1];

    for (int tb = 0; tb < ntile; tb += TA) {
#pragma unroll
        for (int u = 0; u < TA; ++u) {
            const int ti = tb + u;
            if (ti >= ntile) break;
            const int col0 = colblock(t_beg + ti) * 128;
            const int which = QKV ? col0 / p.HC : 0;
            float bz[4][4];
            load_bias(ti, bz);
            f32x4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                const int s = ti * NK + kt;                         // global stage index; slot s & 1
                __syncthreads();                                    // stage s is in LDS; slot (s + 1) & 1 is free
                // hand the NEXT stage to LDS and immediately re-use its registers for the stage TA tiles later
                if (kt + 1 < NK) {
                    *(u32x4 *)(sW + ((s + 1) & 1) * 16384 + lo0) = wr[u][kt + 1][0];
                    *(u32x4 *)(sW + ((s + 1) & 1) * 16384 + lo1) = wr[u][kt + 1][1];
                } else if (ti + 1 < ntile) {
                    *(u32x4 *)(sW + ((s + 1) & 1) * 16384 + lo0) = wr[(u + 1) % TA][0][0];
                    *(u32x4 *)(sW + ((s + 1) & 1) * 16384 + lo1) = wr[(u + 1) % TA][0][1];
                }
                if (kt == NK - 1) wload(ti + TA, wr[u]);            // all of wr[u] has been handed over by now
                const unsigned char *a = sA + kt * 16384, *w = sW + (s & 1) * 16384;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4 fa[2], fw[4];
                    const int cc = kk * 4 + (lane >> 4);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int R = wm * 32 + t * 16 + (lane & 15);
                        fa[t] = *(const u32x4 *)(a + R * 128 + ((cc ^ (R & 7)) << 4));
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int R = wn * 64 + t * 16 + (lane & 15);
                        fw[t] = *(const u32x4 *)(w + R * 128 + ((cc ^ (R & 7)) << 4));
                    }
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Mma16<T>::run(fw[ni], fa[mi], acc[mi][ni]);
                }
            }
            astat_epilogue<T, ACT>(p, acc, bz, stg, rs, row0, col0, which, wm, wn, lane);
        }
    }
}

// returns 0 = launched, -1 = not applicable
int launch_gemm_astat(int prec, const GemmParams &p0, const QkvScatter *qs, int act, hipStream_t st) {
    GemmParams p = p0;
    const int es = (int)esize(prec), BK = 128 / es;
    const int nk = p.K / BK;
    if (nk < 1 || nk > 4 || p.K % BK) return -1;
    const int nrt = (p.M + 127) / 128;
    const int lds = nk * 16384 + 65536 + 8 * 2304;
    auto plan = [&](int nct) {                                    // one resident workgroup per CU
        int groups = 256 / nrt;
        groups = groups > nct ? nct : (groups < 1 ? 1 : groups);
        int ntile = (nct + groups - 1) / groups;
        groups = (nct + ntile - 1) / ntile;
        p.nct = nct; p.nt = ntile;
        return dim3((unsigned)groups, (unsigned)nrt);
    };
    const int use_dma = DA_XENV("DA_ASTAT_DMA", 0);
    const bool rs = !use_dma && (nk == 2 || nk == 4);
    const int lds_rs = nk * 16384 + 32768 + 8 * 2304;
#define DA_ASTAT_RS(TT, VO, AC, NKK, GRID)                                                                 \
    do {                                                                                                   \
        static bool attr = false;                                                                          \
        if (!attr) {                                                                                       \
            DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_gemm_astat_rs<TT, VO, AC, NKK>,               \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, NKK * 16384 + 32768 + 8 * 2304)); \
            attr = true;                                                                                   \
        }                                                                                                  \
        k_gemm_astat_rs<TT, VO, AC, NKK><<<GRID, 512, lds_rs, st>>>(p);                                    \
    } while (0)
#define DA_ASTAT_LAUNCH(TT, VO, AC, GRID)                                                                 \
    do {                                                                                                   \
        if (rs && nk == 4) { DA_ASTAT_RS(TT, false, AC, 4, GRID); break; }                                 \
        if (rs && nk == 2) { DA_ASTAT_RS(TT, false, AC, 2, GRID); break; }                                 \
        static bool attr = false;                                                                          \
        if (!attr) {                                                                                       \
            DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_gemm_astat<TT, AC>,                       \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384 + 65536 + 8 * 2304)); \
            attr = true;                                                                                   \
        }                                                                                                  \
        k_gemm_astat<TT, AC><<<GRID, 512, lds, st>>>(p);                                               \
    } while (0)
#define DA_ASTAT_ACT(TT, GRID)                                                \
    do {                                                                       \
        if (act == DA_ACT_GELU) DA_ASTAT_LAUNCH(TT, false, DA_ACT_GELU, GRID); \
        else if (act == DA_ACT_LEAKY02) DA_ASTAT_LAUNCH(TT, false, DA_ACT_LEAKY02, GRID); \
        else DA_ASTAT_LAUNCH(TT, false, DA_ACT_NONE, GRID);                    \
    } while (0)
    if (!qs) {
        const dim3 grid = plan((p.Nout + 127) / 128);
        if (prec == DA_PREC_BF16) DA_ASTAT_ACT(bf16_t, grid);
        else DA_ASTAT_ACT(float, grid);
    } else if (rs) {
        const dim3 g = plan(p.Nout / 128);                           // Q | K | V (| skip) tiles in ONE launch
        if (prec == DA_PREC_BF16) { if (nk == 4) DA_ASTAT_RS(bf16_t, true, DA_ACT_NONE, 4, g); else DA_ASTAT_RS(bf16_t, true, DA_ACT_NONE, 2, g); }
        else { if (nk == 4) DA_ASTAT_RS(float, true, DA_ACT_NONE, 4, g); else DA_ASTAT_RS(float, true, DA_ACT_NONE, 2, g); }
    } else {
        return -1;             // QKV scatter is only built into the register-staged kernel; the caller falls back
    }
#undef DA_ASTAT_ACT
#undef DA_ASTAT_LAUNCH
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
