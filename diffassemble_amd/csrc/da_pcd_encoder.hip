// 3D piece encoder: the reference's vector-neuron DGCNN over fragment point clouds, inference
// (eval-mode BatchNorm), plus the nearest-neighbour distances of the part-accuracy metric.
//
// Replaces (paths under /root/reference/puzzle_diff/model/):
//   backbones/vnn/vn_dgcnn.py:34-74     VN_DGCNN.forward: 3 x (kNN graph feature -> VNLinearLeakyReLU [x2] -> mean over
//                                       the 20 neighbours), cat -> conv6 -> mean over points -> [P, 768] (or linear0, inv)
//   backbones/vnn/vn_dgcnn.py:84-120    get_graph_feature / knn: a dense [N, N] pairwise matrix + topk(20) per cloud,
//                                       then a [P, 2C, 3, N, 20] gather (device hard-coded to 'cuda', :94)
//   backbones/vnn/vn_layers.py:50-91    VNLinearLeakyReLU (map_to_feat, VNBatchNorm :133-154, map_to_dir, projection)
//   chamfer_distance.py:148-149         pytorch3d knn_points(K = 1) both ways (utils_3d.py:1089-1129 calc_part_acc)
// The reference materialises the [P, N, N] distance matrix, the [P, 2C, 3, N, 20] edge tensor and ~10 elementwise
// temporaries of that size per layer (fp32; 20 x 42 x 3 floats per point and layer).  Here:
//   * k_pcd_knn / k_pcd_knn64_mfma keep a 16- / 32-query x N slab of the distance matrix in LDS (the 63-dimensional stages
//     compute it on the matrix cores in fp32) and select the 20 nearest per query there (select_set: a lower bound of the
//     threshold from the lanes' maxima, the few keys above it compacted and ranked): the matrix never reaches HBM, only the
//     int32 neighbour lists do;
//   * the first layer of a stage is linear in cat(x_j - x_i, x_i), so it is evaluated PER POINT once
//     (k_pcd_premap: A = W[:, :C] x, U = (W[:, C:] - W[:, :C]) x, for the feature and the direction maps) and an edge
//     costs two vector adds: p = A_j + U_i.  20x fewer multiplies than the reference's per-edge matmul;
//   * k_pcd_edge (one thread per point) walks the 20 neighbours, applies BatchNorm-of-the-norm + the vector leaky projection, the stage's second VN layer (weights through scalar
//     loads) and the mean over neighbours in registers;
//   * k_pcd_conv6 fuses the concat, conv6, its activation and the mean over points (wave reduction -> per-block
//     partials, summed in a fixed order: deterministic).
// All arithmetic is fp32 on the vector ALU: the neighbour selection is discrete, and the whole encoder runs once per
// sampling loop (spatial_diffusion_3d_test_double_diffusion.py:700), not per step.
#include <float.h>

#include "da_internal.h"

namespace da {

constexpr int KNN = DA_PCD_K, VC = DA_PCD_C, VROW = DA_PCD_ROW, V3 = VC * 3;   // 20 neighbours, 21 channels, 64-float rows
constexpr float VN_EPS = 1e-6f;                                               // vn_layers.py:11
typedef float f32x2 __attribute__((ext_vector_type(2)));

// -------------------------------------------------------------------------------------------------------------
// Wave-wide maximum through DPP (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast15 / row_bcast31): no LDS
// round trips.  Every lane returns the maximum of the 64 inputs.
__device__ __forceinline__ float wave_max(float v) {
#define DA_DPP_MAX(ctrl, rmask)                                                                                          \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false)))
    DA_DPP_MAX(0x111, 0xf);
    DA_DPP_MAX(0x112, 0xf);
    DA_DPP_MAX(0x114, 0xf);
    DA_DPP_MAX(0x118, 0xf);
    DA_DPP_MAX(0x142, 0xa);
    DA_DPP_MAX(0x143, 0xc);
#undef DA_DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The k largest scores of one LDS row (Npad = 64 NR entries at most), k <= 64, by k rounds of a wave-wide arg-max over
// register-resident scores (lane l holds entries l, l + 64, ...).  Ties go to the lower index.  Lane t returns the
// index of rank t.  (VALU-issue bound, ~7 NR instructions per round.  Reading the winning lane's registers into SGPRs
// and patching one slot under a uniform branch has fewer vector instructions but ran 25 % slower: v_readlane -> SALU
// compare chains stall.)
template <int NR>
__device__ __forceinline__ int select_topk(const float *row, int Npad, int k, int lane) {
    float v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = r * 64 < Npad ? row[r * 64 + lane] : -INFINITY;
    int res = 0;
    for (int t = 0; t < k; ++t) {
        float best = v[0];
#pragma unroll
        for (int r = 1; r < NR; ++r) best = fmaxf(best, v[r]);
        const float wmax = wave_max(best);
        unsigned long long tied = __ballot(best == wmax);
        int wl = __builtin_ctzll(tied);
        if (tied & (tied - 1)) {                       // several lanes hold the maximum: the lowest INDEX wins
            int br = 0;
#pragma unroll
            for (int r = NR - 1; r >= 0; --r) br = v[r] == wmax ? r : br;
            const int bj = br * 64 + lane;
            int wbj = __builtin_amdgcn_readlane(bj, wl);
            for (tied &= tied - 1; tied; tied &= tied - 1) {
                const int l2 = __builtin_ctzll(tied), b2 = __builtin_amdgcn_readlane(bj, l2);
                if (b2 < wbj) { wbj = b2; wl = l2; }
            }
        }
        bool done = lane != wl;
        int wr = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool hit = !done && v[r] == wmax;
            wr = hit ? r : wr;
            v[r] = hit ? -INFINITY : v[r];
            done = done || hit;
        }
        const int wj = __builtin_amdgcn_readlane(wr, wl) * 64 + wl;
        res = lane == t ? wj : res;
    }
    return res;
}

// The same k entries as an UNORDERED set (what the encoder needs: it pools over the neighbours), 3x fewer vector
// instructions: the scores become order-preserving 32-bit keys, the k-th largest key tau is built bit by bit (32 rounds
// of "how many keys >= candidate": one v_cmp per register, the counting is s_bcnt1 on the ballot masks), then the keys
// > tau and the first k - count(> tau) keys == tau (lowest indices first, the ordered variant's tie rule) are compacted
// into dst[0..k) through mbcnt prefix counts.  The rounds are bound by the SCALAR unit (s_bcnt1 + s_add per register, one
// scalar instruction per cycle per CU): bracketing tau between the k-th largest lane maximum and the maximum (fewer
// 16-register rounds, 32 one-register rounds more) and searching two queries of a wave jointly both measured SLOWER (+13 %).
template <int NR>
__device__ __forceinline__ void select_set_keys(const unsigned (&u)[NR], int N, int k, int lane, int32_t *dst) {
    unsigned tau = 0;
    bool exact = false;                                 // exactly k keys >= tau: the set is known, stop refining
    for (int bit = 31; bit >= 0 && !exact; --bit) {
        const unsigned cand = tau | (1u << bit);
        int c = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) c += __builtin_popcountll(__ballot(u[r] >= cand));
        tau = c >= k ? cand : tau;
        exact = c == k;
    }
    int g = k;
    if (!exact) {                                       // tau is the k-th largest key and it repeats
        g = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) g += __builtin_popcountll(__ballot(u[r] > tau));
    }
    const int need = k - g;
    int og = 0, oe = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool gt = exact ? u[r] >= tau : u[r] > tau, eq = !exact && u[r] == tau;
        const unsigned long long mg = __ballot(gt), me = __ballot(eq);
        const int pg = og + __builtin_amdgcn_mbcnt_hi((unsigned)(mg >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mg, 0));
        const int pe = oe + __builtin_amdgcn_mbcnt_hi((unsigned)(me >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)me, 0));
        const int j = min(r * 64 + lane, N - 1);
        if (gt) dst[pg] = j;
        else if (eq && pe < need) dst[g + pe] = j;
        og += __builtin_popcountll(mg);
        oe += __builtin_popcountll(me);
    }
}

// The selection the encoder runs (round 3).  The full-width search above costs 32 rounds x NR compares + NR scalar bit
// counts; almost all of it is spent on keys that are nowhere near the top.  Here a cheap lower bound of the threshold
// comes first: every lane's largest key (NR - 1 v_max), and a value L that at least k of those 64 lane maxima reach (a
// bit search over ONE register per lane) -- so at least k keys are >= L, and for keys spread over the lanes at random about
// 1.2 k - 2 k of them are.  The keys >= L are compacted (in index order) into the row's own LDS space, which is dead once the
// keys sit in registers, and ranked against each other there: candidate p's rank = the number of candidates with a larger
// key, or an equal key and a lower index -- the tie rule of the ordered variant; ranks < k are the answer, written at
// dst[rank], i.e. in decreasing order.  More than 128 candidates (keys bunched in few lanes): the full search.
template <int NR>
__device__ __forceinline__ void select_set(float *row, int Npad, int N, int k, int lane, int32_t *dst) {
    unsigned u[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const unsigned b = __builtin_bit_cast(unsigned, r * 64 < Npad ? row[r * 64 + lane] : -INFINITY);
        u[r] = b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
    }
    if (Npad < 256) {                                   // rows shorter than the 256-entry scratch: full search
        select_set_keys<NR>(u, N, k, lane, dst);
        return;
    }
    unsigned m = u[0];
#pragma unroll
    for (int r = 1; r < NR; ++r) m = max(m, u[r]);
    unsigned L = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = L | (1u << bit);
        const int c = __builtin_popcountll(__ballot(m >= cand));
        if (c >= k) {
            L = cand;
            if (c <= k + 2) break;                      // (nearly) as tight as lane maxima get
        }
    }
    unsigned *ck = (unsigned *)row;
    int *ci = (int *)row + 128;
    int ct = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool ge = u[r] >= L;
        const unsigned long long mk = __ballot(ge);
        const int pos = ct + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
        if (ge && pos < 128) { ck[pos] = u[r]; ci[pos] = r * 64 + lane; }
        ct += __builtin_popcountll(mk);
    }
    if (ct > 128) {
        select_set_keys<NR>(u, N, k, lane, dst);
        return;
    }
    const unsigned c0 = lane < ct ? ck[lane] : 0u, c1 = lane + 64 < ct ? ck[lane + 64] : 0u;
    const int i0 = ci[lane], i1 = ci[lane + 64];
    int r0 = 0, r1 = 0;
    if (ct <= 64) {
        for (int j = 0; j < ct; ++j) {
            const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)c0, j);
            r0 += (kj > c0 || (kj == c0 && j < lane)) ? 1 : 0;
        }
    } else {
        for (int j = 0; j < ct; ++j) {
            const unsigned kj = ck[j];
            r0 += (kj > c0 || (kj == c0 && j < lane)) ? 1 : 0;
            r1 += (kj > c1 || (kj == c1 && j < lane + 64)) ? 1 : 0;
        }
    }
    if (lane < ct && r0 < k) dst[r0] = min(i0, N - 1);
    if (lane + 64 < ct && r1 < k) dst[r1] = min(i1, N - 1);
}

// Same selection for rows that do not fit the registers: the scores stay in LDS, the winner is overwritten.
__device__ __forceinline__ int select_topk_lds(volatile float *row, int Npad, int k, int lane) {
    int res = 0;
    for (int t = 0; t < k; ++t) {
        float best = -INFINITY;
        int bj = lane;
        for (int j = lane; j < Npad; j += 64) {
            const float x = row[j];
            if (x > best) { best = x; bj = j; }
        }
        const float wmax = wave_max(best);
        unsigned long long tied = __ballot(best == wmax);
        int wl = __builtin_ctzll(tied), wbj = __builtin_amdgcn_readlane(bj, wl);
        for (tied &= tied - 1; tied; tied &= tied - 1) {
            const int l2 = __builtin_ctzll(tied), b2 = __builtin_amdgcn_readlane(bj, l2);
            if (b2 < wbj) { wbj = b2; wl = l2; }
        }
        if (lane == wl) row[wbj] = -INFINITY;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        res = lane == t ? wbj : res;
    }
    return res;
}

// k nearest neighbours (self included) in an F-dimensional feature space, per cloud.
// grid (ceil(N / QB), clouds), 256 threads.  LDS: the QB squared query norms and QB x Npad scores.
// score(i, j) = -|xi|^2 + 2 xi.xj - |xj|^2, the reference's formula (vn_dgcnn.py:115-117); the k LARGEST are kept,
// in decreasing order (topk), ties broken towards the lower index.  Non-finite scores rank last.
template <int F, int NT = 256>
__global__ __launch_bounds__(NT) void k_pcd_knn(const float *__restrict__ X, int ldx, int N, int QB, int Npad, int k,
                                                int ordered, int32_t *__restrict__ idx) {
    extern __shared__ float smem[];
    float *qxx = smem, *score = qxx + ((QB + 3) & ~3);
    const int cloud = blockIdx.y, q0 = blockIdx.x * QB, tid = threadIdx.x;
    const float *Xc = X + (size_t)cloud * N * ldx;
    if (tid < QB) {
        const float *qr = Xc + (size_t)min(q0 + tid, N - 1) * ldx;
        float s = 0.f;
        for (int f = 0; f < (F == 3 ? 3 : F); ++f) s += qr[f] * qr[f];
        qxx[tid] = s;
    }
    __syncthreads();
    // Scores of this thread's candidates against the block's queries.  A query row is wave-uniform: it comes through
    // the scalar cache into SGPRs (no LDS traffic), the candidate row sits in VGPRs, the dot product is packed fp32 FMAs.
    for (int j = tid; j < Npad; j += NT) {
        if (j < N) {
            if constexpr (F == 64) {
                f32x2 c[32];
#pragma unroll
                for (int f = 0; f < 32; f += 2) {
                    const float4 v = *(const float4 *)(Xc + (size_t)j * ldx + 2 * f);
                    c[f] = f32x2{v.x, v.y}; c[f + 1] = f32x2{v.z, v.w};
                }
                f32x2 xx2 = {0.f, 0.f};
#pragma unroll
                for (int f = 0; f < 32; ++f) xx2 = __builtin_elementwise_fma(c[f], c[f], xx2);
                const float xxj = xx2.x + xx2.y;
#pragma unroll 2
                for (int q = 0; q < QB; ++q) {
                    const f32x2 *qr = (const f32x2 *)(Xc + (size_t)min(q0 + q, N - 1) * ldx);
                    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                    for (int f = 0; f < 32; f += 2) {
                        a0 = __builtin_elementwise_fma(qr[f], c[f], a0);
                        a1 = __builtin_elementwise_fma(qr[f + 1], c[f + 1], a1);
                    }
                    const float dot = (a0.x + a0.y) + (a1.x + a1.y);
                    const float sc = (-qxx[q] - (-2.f * dot)) - xxj;
                    score[q * Npad + j] = fabsf(sc) <= FLT_MAX ? sc + 0.f : -FLT_MAX;  // NaN / inf rank last; -0 -> +0 (bit compares)
                }
            } else {
                const float c0 = Xc[(size_t)j * ldx], c1 = Xc[(size_t)j * ldx + 1], c2 = Xc[(size_t)j * ldx + 2];
                const float xxj = c0 * c0 + c1 * c1 + c2 * c2;
                for (int q = 0; q < QB; ++q) {
                    const float *qr = Xc + (size_t)min(q0 + q, N - 1) * ldx;
                    const float dot = qr[0] * c0 + qr[1] * c1 + qr[2] * c2;
                    const float sc = (-qxx[q] - (-2.f * dot)) - xxj;
                    score[q * Npad + j] = fabsf(sc) <= FLT_MAX ? sc + 0.f : -FLT_MAX;
                }
            }
        } else {
            for (int q = 0; q < QB; ++q) score[q * Npad + j] = -INFINITY;          // padding: never before a real point
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int q = wave; q < QB && q0 + q < N; q += NT / 64) {
        float *row = score + q * Npad;
        int32_t *dst = idx + ((size_t)cloud * N + q0 + q) * k;
        if (!ordered && Npad <= 1024) {
            if (Npad <= 256) select_set<4>(row, Npad, N, k, lane, dst);
            else if (Npad <= 512) select_set<8>(row, Npad, N, k, lane, dst);
            else select_set<16>(row, Npad, N, k, lane, dst);
            continue;
        }
        int r;
        if (Npad <= 256) r = select_topk<4>(row, Npad, k, lane);
        else if (Npad <= 512) r = select_topk<8>(row, Npad, k, lane);
        else if (Npad <= 1024) r = select_topk<16>(row, Npad, k, lane);
        else r = select_topk_lds(row, Npad, k, lane);
        if (lane < k) dst[lane] = min(r, N - 1);
    }
}

// The same neighbour search for the 63-dimensional stages with the Gram matrix on the matrix cores (round 3): the scores of a
// block are a [32 queries] x [N candidates] x 64 product; on the vector ALU the query row streams through SGPRs, one s_load
// burst per query with nothing to hide it behind at two waves per SIMD.  v_mfma_f32_32x32x2_f32 (full fp32 products and sums)
// does a 32-candidate x 32-query tile in 32 instructions.  640 clouds of 1000 points: k_pcd_knn<64> 2.49 ms -> 1.97 ms here;
// by ablation 0.3 ms is loading and 0.75 ms the MFMA phase (floor 0.62 ms: 82 GFLOP at the 137 TFLOP/s fp32 matrix rate), the
// rest the selection.  The squared norms come from the kernel that produced the rows (k_pcd_edge writes |row|^2 next to them):
// computing them here, per block, re-read the whole cloud once more and cost 0.7 ms.  Operands: the MFMA's k pair of step s is (s, 32 + s), so lane (row, kk)
// holds the CONTIGUOUS half [32 kk, 32 kk + 32) of its row -- eight 16-byte loads; A = candidates, B = queries, so that a
// lane ends up with 4 x 4 consecutive candidates of ONE query and writes them as four 16-byte LDS stores (row stride
// Npad + 4 floats: conflict-free).  grid (ceil(N / 32), clouds), 512 threads, one block per CU (the 32 x N slab is 129 KB at
// N = 1000); the selection that follows is select_set, one query per wave at a time.
typedef float f32x16k __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(1024) void k_pcd_knn64_mfma(const float *__restrict__ X, const float *__restrict__ XN, int N, int Npad, int k,
                                                         int32_t *__restrict__ idx) {
    extern __shared__ float smem[];
    const int RS = Npad + 4;                              // score row stride (floats)
    float *score = smem, *cxx = score + 32 * RS, *qxx = cxx + Npad;
    const int cloud = blockIdx.y, q0 = blockIdx.x * 32, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *Xc = X + (size_t)cloud * N * VROW, *XNc = XN + (size_t)cloud * N;
    // squared norms of the candidates and of the block's queries: written next to the rows by the kernel that produced them
    for (int j = tid; j < Npad + 32; j += 1024) {
        if (j < Npad) cxx[j] = XNc[min(j, N - 1)];
        else qxx[j - Npad] = XNc[min(q0 + j - Npad, N - 1)];
    }
    const int rl = lane & 31, kk = lane >> 5;
    float qr[32];
    {
        const float4 *r = (const float4 *)(Xc + (size_t)min(q0 + rl, N - 1) * VROW + 32 * kk);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float4 v = r[e]; qr[4 * e] = v.x; qr[4 * e + 1] = v.y; qr[4 * e + 2] = v.z; qr[4 * e + 3] = v.w; }
    }
    auto load_tile = [&](float4 (&cr)[8], int c0) {
        const float4 *r = (const float4 *)(Xc + (size_t)min(c0 + rl, N - 1) * VROW + 32 * kk);
#pragma unroll
        for (int e = 0; e < 8; ++e) cr[e] = r[e];
    };
    __syncthreads();
    const float nq = -qxx[rl];
    auto do_tile = [&](const float4 (&cr)[8], int c0) {
        f32x16k acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[e].x, qr[4 * e], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[e].y, qr[4 * e + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[e].z, qr[4 * e + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[e].w, qr[4 * e + 3], acc, 0, 0, 0);
        }
        // lane (query rl, half kk) holds candidates c0 + 8 a + 4 kk + b in acc[4 a + b]
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int cb = c0 + 8 * a + 4 * kk;
            const float4 cx = *(const float4 *)(cxx + cb);
            float o[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float sc = (nq - (-2.f * acc[4 * a + b])) - (b == 0 ? cx.x : b == 1 ? cx.y : b == 2 ? cx.z : cx.w);
                o[b] = cb + b < N ? (fabsf(sc) <= FLT_MAX ? sc + 0.f : -FLT_MAX) : -INFINITY;   // NaN / inf rank last, the padding after them
            }
            *(float4 *)(score + rl * RS + cb) = float4{o[0], o[1], o[2], o[3]};
        }
    };
    // the wave's candidate tiles (at most two below 1 024 candidates), the FIRST HALF of the next tile's rows in flight under the current tile's
    // MFMAs: a whole second tile (round 3's form) needs 32 more registers than this kernel's 128 hold -- five were spilled to scratch
    float4 ca[8], nb[4];
    int c0 = wave * 32;
    if (c0 < Npad) load_tile(ca, c0);
    for (; c0 < Npad; c0 += 512) {
        const bool more = c0 + 512 < Npad;
        const float4 *rn = (const float4 *)(Xc + (size_t)min(c0 + 512 + rl, N - 1) * VROW + 32 * kk);
        if (more) {
#pragma unroll
            for (int e = 0; e < 4; ++e) nb[e] = rn[e];
        }
        do_tile(ca, c0);
        if (more) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ca[e] = nb[e]; ca[4 + e] = rn[4 + e]; }
        }
    }
    __syncthreads();
    for (int q = wave; q < 32 && q0 + q < N; q += 16) {
        float *row = score + q * RS;
        int32_t *dst = idx + ((size_t)cloud * N + q0 + q) * k;
        if (Npad <= 256) select_set<4>(row, Npad, N, k, lane, dst);
        else if (Npad <= 512) select_set<8>(row, Npad, N, k, lane, dst);
        else select_set<16>(row, Npad, N, k, lane, dst);
    }
}

// -------------------------------------------------------------------------------------------------------------
// Per-point maps of a stage's first VN layer.  X row: C channels x 3 (ldx floats per point).  T row (256 floats):
// [A | Ad | U | Ud], 64-float segments of 21 x 3 values: A = Wf[:, :C] x, Ad = Wd[:, :C] x, U = (Wf[:, C:] - Wf[:, :C]) x,
// Ud likewise.  Wm = [4][21][C] holds the four maps (host-packed).  One thread per point.
template <int C>
__global__ __launch_bounds__(256) void k_pcd_premap(const float *__restrict__ X, int ldx, const float *__restrict__ Wm,
                                                    long long total, float *__restrict__ T) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    float x[C * 3];
#pragma unroll
    for (int e = 0; e < C * 3; ++e) x[e] = X[p * ldx + e];
    float *t = T + p * 4 * VROW;
    // four output channels (12 floats = three 16-byte stores) at a time: a thread owns a whole 1 KB row of T, so every
    // store instruction touches 64 different lines -- 4-byte stores made this kernel store-bound (0.67 -> 0.34 ms)
    for (int m = 0; m < 4; ++m) {
#pragma unroll 1
        for (int o0 = 0; o0 < 24; o0 += 4) {
            float a[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = o0 + q < VC ? o0 + q : VC - 1;          // rows 21..23 of the padded segment: recomputed, masked below
                const float *w = Wm + (m * VC + o) * C;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float wv = w[c];
                    a0 += wv * x[c * 3]; a1 += wv * x[c * 3 + 1]; a2 += wv * x[c * 3 + 2];
                }
                const bool on = o0 + q < VC;
                a[q * 3] = on ? a0 : 0.f; a[q * 3 + 1] = on ? a1 : 0.f; a[q * 3 + 2] = on ? a2 : 0.f;
            }
            if (o0 < 20) {
#pragma unroll
                for (int e = 0; e < 12; e += 4) *(float4 *)(t + m * VROW + o0 * 3 + e) = float4{a[e], a[e + 1], a[e + 2], a[e + 3]};
            } else {                                                   // channel 20 + the pad: floats 60..63
                *(float4 *)(t + m * VROW + 60) = float4{a[0], a[1], a[2], 0.f};
            }
        }
    }
}

// VNBatchNorm on the norm (eval: norm * scale + shift) + the vector leaky projection, vn_layers.py:80-91.  Branch-free;
// v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the IEEE sequences: ~1e-7 relative, far inside the parity tolerance.
__device__ __forceinline__ void vn_act(float &p0, float &p1, float &p2, float d0, float d1, float d2, float scale, float shift) {
    const float norm = __builtin_amdgcn_sqrtf(p0 * p0 + p1 * p1 + p2 * p2) + VN_EPS;
    const float f = (norm * scale + shift) * __builtin_amdgcn_rcpf(norm);
    p0 *= f; p1 *= f; p2 *= f;
    const float dot = p0 * d0 + p1 * d1 + p2 * d2;
    // 0.2 p + 0.8 (p - [dot < 0] dot / (|d|^2 + eps) d)
    const float c = dot < 0.f ? 0.8f * dot * __builtin_amdgcn_rcpf(d0 * d0 + d1 * d1 + d2 * d2 + VN_EPS) : 0.f;
    p0 -= c * d0; p1 -= c * d1; p2 -= c * d2;
}

// One stage's edge work: for point i and each neighbour j: h = act(A_j + U_i; Ad_j + Ud_i); optionally the second
// VN layer (21 -> 21); mean over the k neighbours -> Xout row (64 floats, 63 used, pad = 0).  One thread per point.
// wb = [2][21 o][22] (feature map, direction map; rows zero-padded to 11 pairs) then [2][21] (scale, shift) of the
// second layer.  The second layer is packed fp32 over channel pairs: (even, odd partial sums) += (w[o][c], w[o][c + 1]) *
// (h[c][k], h[c + 1][k]); a weight pair is one 64-bit scalar operand and streams through SGPRs.
template <bool HAS_B>
__global__ __launch_bounds__(128, 1) void k_pcd_edge(const float *__restrict__ T, const int32_t *__restrict__ idx,
                                                     const float *__restrict__ bn_a, const float *__restrict__ wb, int N,
                                                     int clouds, float *__restrict__ Xout, float *__restrict__ xn) {
    // (Round 3, tried and withdrawn: requesting chunk c of neighbour r + 1 into the registers chunk c of neighbour r was just read
    // from -- 460 VGPRs, 32 gathers in flight per thread.  1.77 vs 2.03 ms per 640 000 points on one kind of box of the pool,
    // 2.91 vs 2.23 ms on another: 32 MB of outstanding gathers chip-wide is more than the slower boxes' memory side digests.)
    // (Round 4, tried and withdrawn: the second layer's 966 weights staged in LDS and read from there with broadcast reads instead of
    // streaming through SGPRs -- the round-3 analysis blamed the 64 s_load_dwordx16 per neighbour: 3.03 vs 2.24 ms per 640 000
    // points; a scalar weight pair is a free operand of v_pk_fma_f32, an LDS one costs a VGPR pair and an LDS instruction each.)
    // The point's own U / Ud rows stay in 126 registers (re-reading them per neighbour doubled the divergent 16-byte
    // loads the kernel is bound by: 5.4 ms vs 2.3 ms per 640 000 points); one wave per SIMD, overflow into AGPRs.
    constexpr bool PIN_U = true;
    // XCD-aware block map (workgroup L runs on XCD L % 8): the blocks of ONE cloud share an XCD, so the cloud's A | Ad rows
    // (512 B x N = 512 KB at N = 1000), which its N x 20 gathers hit at random, are filled into one L2 once instead of into
    // all eight
    const int nb = (N + 127) >> 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cloud = (slot / nb) * 8 + xcd, i = (slot % nb) * 128 + (int)threadIdx.x;
    if (cloud >= clouds || i >= N) return;
    const long long base = (long long)cloud * N, p = base + i;
    float acc[V3];
#pragma unroll
    for (int e = 0; e < V3; ++e) acc[e] = 0.f;
    float4 uu[PIN_U ? 16 : 1], ud[PIN_U ? 16 : 1];
    if constexpr (PIN_U) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            uu[e] = *(const float4 *)(T + p * 4 * VROW + 2 * VROW + 4 * e);
            ud[e] = *(const float4 *)(T + p * 4 * VROW + 3 * VROW + 4 * e);
        }
    }
#pragma unroll 1
    for (int r = 0; r < KNN; ++r) {
        const int j = idx[p * KNN + r];
        const float *tj = T + (base + j) * 4 * VROW;
        int toff = 0, woff = 0;                           // opaque zeros: keep the U / Ud loads and the weight loads
        asm volatile("" : "+v"(toff));                    // inside the loop (not hoisted into ~1000 live registers)
        asm volatile("" : "+s"(woff));
        const float *ti = T + p * 4 * VROW + 2 * VROW + toff;
        const float *wl = wb + woff;
        // h[k][c / 2] holds the channel pair (c, c + 1) of component k: the operand layout of the packed second layer
        f32x2 h[3][VC / 2 + 1];
        h[0][VC / 2] = h[1][VC / 2] = h[2][VC / 2] = f32x2{0.f, 0.f};
        // first layer, four channels (12 floats = three 16-byte loads per operand) at a time; channel 20 + the pad last
#pragma unroll
        for (int e0 = 0; e0 < VROW; e0 += 12) {
            float a[12], ad[12];
#pragma unroll
            for (int e = 0; e < 12 && e0 + e < VROW; e += 4) {
                const float4 x = *(const float4 *)(tj + e0 + e), xd = *(const float4 *)(tj + VROW + e0 + e);
                float4 y, yd;
                if constexpr (PIN_U) { y = uu[(e0 + e) / 4]; yd = ud[(e0 + e) / 4]; }
                else { y = *(const float4 *)(ti + e0 + e); yd = *(const float4 *)(ti + VROW + e0 + e); }
                a[e] = x.x + y.x; a[e + 1] = x.y + y.y; a[e + 2] = x.z + y.z; a[e + 3] = x.w + y.w;
                ad[e] = xd.x + yd.x; ad[e + 1] = xd.y + yd.y; ad[e + 2] = xd.z + yd.z; ad[e + 3] = xd.w + yd.w;
            }
#pragma unroll
            for (int cc = 0; cc < 4 && e0 / 3 + cc < VC; ++cc) {
                const int c = e0 / 3 + cc;
                float p0 = a[cc * 3], p1 = a[cc * 3 + 1], p2 = a[cc * 3 + 2];
                vn_act(p0, p1, p2, ad[cc * 3], ad[cc * 3 + 1], ad[cc * 3 + 2], bn_a[c], bn_a[VC + c]);
                if constexpr (!HAS_B) { acc[c * 3] += p0; acc[c * 3 + 1] += p1; acc[c * 3 + 2] += p2; }
                else if (c & 1) { h[0][c / 2].y = p0; h[1][c / 2].y = p1; h[2][c / 2].y = p2; }
                else { h[0][c / 2].x = p0; h[1][c / 2].x = p1; h[2][c / 2].x = p2; }
            }
            __builtin_amdgcn_sched_barrier(0);            // do not hoist every chunk's loads to the top (VGPR budget)
        }
        if constexpr (HAS_B) {
            constexpr int WR = VC / 2 + 1;                 // 11 weight pairs per (zero-padded) row of 22
            const f32x2 *wf = (const f32x2 *)wl, *wd = wf + VC * WR;
            const float *bn = wl + 4 * VC * WR;
#pragma unroll
            for (int o = 0; o < VC; ++o) {
                f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, s2 = {0.f, 0.f}, t0 = {0.f, 0.f}, t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < WR; ++c) {
                    const f32x2 a = wf[o * WR + c], b = wd[o * WR + c];
                    s0 = __builtin_elementwise_fma(a, h[0][c], s0); s1 = __builtin_elementwise_fma(a, h[1][c], s1);
                    s2 = __builtin_elementwise_fma(a, h[2][c], s2);
                    t0 = __builtin_elementwise_fma(b, h[0][c], t0); t1 = __builtin_elementwise_fma(b, h[1][c], t1);
                    t2 = __builtin_elementwise_fma(b, h[2][c], t2);
                }
                float p0 = s0.x + s0.y, p1 = s1.x + s1.y, p2 = s2.x + s2.y;
                vn_act(p0, p1, p2, t0.x + t0.y, t1.x + t1.y, t2.x + t2.y, bn[o], bn[VC + o]);
                acc[o * 3] += p0; acc[o * 3 + 1] += p1; acc[o * 3 + 2] += p2;
            }
        }
    }
    float *xo = Xout + p * VROW;
    float nn = 0.f;                                       // |row|^2 for the next stage's neighbour search (k_pcd_knn64_mfma)
#pragma unroll
    for (int e = 0; e < V3; ++e) { const float v = acc[e] / (float)KNN; xo[e] = v; nn += v * v; }
    xo[V3] = 0.f;
    if (xn) xn[p] = nn;
}

// conv6 (63 -> feat channels, ONE shared direction) over cat(x1, x2, x3), its activation, and the sum over the
// block's points.  w6 = [feat][63] feature map, [63] direction map, [2][feat] scale / shift.
// grid (ceil(N / 256), clouds); partial[cloud][block][feat * 3].
__global__ __launch_bounds__(256) void k_pcd_conv6(const float *__restrict__ X1, const float *__restrict__ X2,
                                                   const float *__restrict__ X3, const float *__restrict__ w6, int feat,
                                                   int N, float *__restrict__ partial) {
    __shared__ float red[4][3 * 256];
    const int cloud = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = blockIdx.x * 256 + tid;
    const bool on = i < N;
    const long long p = (long long)cloud * N + min(i, N - 1);
    float f[3 * V3];
#pragma unroll
    for (int e = 0; e < V3; ++e) { f[e] = X1[p * VROW + e]; f[V3 + e] = X2[p * VROW + e]; f[2 * V3 + e] = X3[p * VROW + e]; }
    const float *wd = w6 + (size_t)feat * V3, *sc = wd + V3, *sh = sc + feat;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
    for (int c = 0; c < V3; ++c) { const float w = wd[c]; d0 += w * f[c * 3]; d1 += w * f[c * 3 + 1]; d2 += w * f[c * 3 + 2]; }
#pragma unroll 1
    for (int o = 0; o < feat; ++o) {
        const float *w = w6 + (size_t)o * V3;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int c = 0; c < V3; ++c) { const float wv = w[c]; p0 += wv * f[c * 3]; p1 += wv * f[c * 3 + 1]; p2 += wv * f[c * 3 + 2]; }
        vn_act(p0, p1, p2, d0, d1, d2, sc[o], sh[o]);
        if (!on) p0 = p1 = p2 = 0.f;
#pragma unroll
        for (int off = 32; off; off >>= 1) { p0 += __shfl_xor(p0, off); p1 += __shfl_xor(p1, off); p2 += __shfl_xor(p2, off); }
        if (lane == 0) { red[wave][o * 3] = p0; red[wave][o * 3 + 1] = p1; red[wave][o * 3 + 2] = p2; }
    }
    __syncthreads();
    float *dst = partial + ((size_t)cloud * gridDim.x + blockIdx.x) * feat * 3;
    for (int e = tid; e < feat * 3; e += 256) dst[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// Mean over points (fixed-order sum of the block partials), then vn_dgcnn.py:62-74: cat(x, mean(x)) pooled over the
// points is the pooled map twice -> out[cloud][(c, k)] for c < 2 feat;  inv: mean over the 2 feat channels of
// linear0(x[c, :]) = linear0(mean_c x[c, :]) -> out[cloud][2 feat].
__global__ __launch_bounds__(256) void k_pcd_final(const float *__restrict__ partial, int nblk, int feat, int N,
                                                   const float *__restrict__ lin0, int inv, float *__restrict__ out, int ldo) {
    __shared__ float m[3 * 256];
    __shared__ float m3[3];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < feat * 3; e += 256) {
        float s = 0.f;
        for (int b = 0; b < nblk; ++b) s += partial[((size_t)cloud * nblk + b) * feat * 3 + e];
        m[e] = s / (float)N;
    }
    __syncthreads();
    float *o = out + (size_t)cloud * ldo;
    if (!inv) {
        for (int e = tid; e < feat * 3; e += 256) { o[e] = m[e]; o[feat * 3 + e] = m[e]; }
        return;
    }
    if (tid < 3) {
        float s = 0.f;
        for (int c = 0; c < feat; ++c) s += m[c * 3 + tid];
        m3[tid] = (s + s) / (float)(2 * feat);
    }
    __syncthreads();
    const float *b0 = lin0 + (size_t)2 * feat * 3;
    for (int e = tid; e < 2 * feat; e += 256) o[e] = lin0[e * 3] * m3[0] + lin0[e * 3 + 1] * m3[1] + lin0[e * 3 + 2] * m3[2] + b0[e];
}

// For every point of a: the squared distance to its nearest point of b (same cloud index).  grid (ceil(N/256), clouds).
__global__ __launch_bounds__(256) void k_nearest_sq(const float *__restrict__ a, const float *__restrict__ b, int N, int M,
                                                    float *__restrict__ out) {
    __shared__ float tile[3 * 1024];
    const int cloud = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    const float *ap = a + ((size_t)cloud * N + min(i, N - 1)) * 3;
    const float ax = ap[0], ay = ap[1], az = ap[2];
    float best = FLT_MAX;
    for (int m0 = 0; m0 < M; m0 += 1024) {
        const int cnt = min(1024, M - m0);
        __syncthreads();
        for (int e = tid; e < cnt * 3; e += 256) tile[e] = b[((size_t)cloud * M + m0) * 3 + e];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float dx = ax - tile[j * 3], dy = ay - tile[j * 3 + 1], dz = az - tile[j * 3 + 2];
            best = fminf(best, dx * dx + dy * dy + dz * dz);
        }
    }
    if (i < N) out[(size_t)cloud * N + i] = best;
}

static int knn_launch(int clouds, int N, int dim, const float *x, int ldx, int k, int ordered, int32_t *idx, hipStream_t st,
                      const float *xn = nullptr) {
    const int F = dim <= 3 ? 3 : 64;
    const int Npad = (N + 63) & ~63;
    static int mfma_knn = -1;
    if (mfma_knn < 0) mfma_knn = DA_XENV("DA_PCD_KNN_VALU", 0) ? 0 : 1;
    if (F == 64 && !ordered && ldx == VROW && Npad <= 1024 && mfma_knn && xn) {
        const size_t lds = (size_t)(32 * (Npad + 4) + Npad + 32) * sizeof(float);
        static bool attrm = false;
        if (!attrm) { DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_pcd_knn64_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)); attrm = true; }
        k_pcd_knn64_mfma<<<dim3((N + 31) / 32, clouds), 1024, lds, st>>>(x, xn, N, Npad, k, idx);
        DA_LAUNCH_CHECK();
        return 0;
    }
    int QB = 32;
    auto bytes = [&](int qb) { return (size_t)(((qb + 3) & ~3) + (size_t)qb * Npad) * sizeof(float); };
    // (one 1024-thread block of 32 queries per CU instead of two 256-thread blocks of 16: 1.11 vs 0.93 ms for the 3-D stage)
    while (QB > 1 && bytes(QB) > 64 * 1024) QB >>= 1;          // two blocks per CU
    DA_REQUIRE(bytes(QB) <= 160 * 1024 - 512, "kNN: %d points per cloud do not fit the LDS slab", N);
    const dim3 grid((N + QB - 1) / QB, clouds);
    // the dynamic-LDS ceiling is raised ONCE per kernel (not per launch: the launches must stay stream-capturable)
    static bool attr3 = false, attr64 = false;
    constexpr int LDS_MAX = 160 * 1024 - 512;
    if (F == 3) {
        if (!attr3) { DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_pcd_knn<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX)); attr3 = true; }
        k_pcd_knn<3><<<grid, 256, bytes(QB), st>>>(x, ldx, N, QB, Npad, k, ordered, idx);
    } else {
        if (!attr64) { DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_pcd_knn<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX)); attr64 = true; }
        k_pcd_knn<64><<<grid, 256, bytes(QB), st>>>(x, ldx, N, QB, Npad, k, ordered, idx);
    }
    DA_LAUNCH_CHECK();
    return 0;
}

// side stream of da_pcd_encoder_forward (one per device, created on first use)
struct PcdSide { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static PcdSide *pcd_side() {
    static PcdSide ctx[16];
    static bool ok[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    PcdSide &c = ctx[dev & 15];
    if (!ok[dev & 15]) {
        if (hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess) return nullptr;
        ok[dev & 15] = true;
    }
    return &c;
}

}  // namespace da

using namespace da;

extern "C" {

int da_knn(int n_clouds, int n_points, int dim, const float *x, int ldx, int k, int32_t *idx, void *stream) {
    DA_REQUIRE(x && idx, "da_knn: null argument");
    DA_REQUIRE(n_clouds > 0 && n_points >= k && k > 0 && k <= 64, "da_knn: need n_points >= k, 0 < k <= 64 (got %d, %d)", n_points, k);
    DA_REQUIRE(dim == 3 ? ldx >= 3 : (dim > 3 && dim <= 64 && ldx == 64), "da_knn: dim 3 (ldx >= 3) or 4..64 with zero-padded 64-float rows");
    return knn_launch(n_clouds, n_points, dim, x, ldx, k, 1, idx, (hipStream_t)stream);
}

int da_nearest_sq(int n_clouds, int n, int m, const float *a, const float *b, float *d_ab, float *d_ba, void *stream) {
    DA_REQUIRE(a && b && (d_ab || d_ba), "da_nearest_sq: null argument");
    DA_REQUIRE(n_clouds > 0 && n > 0 && m > 0, "da_nearest_sq: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (d_ab) { k_nearest_sq<<<dim3((n + 255) / 256, n_clouds), 256, 0, st>>>(a, b, n, m, d_ab); DA_LAUNCH_CHECK(); }
    if (d_ba) { k_nearest_sq<<<dim3((m + 255) / 256, n_clouds), 256, 0, st>>>(b, a, m, n, d_ba); DA_LAUNCH_CHECK(); }
    return 0;
}

static size_t pcd_ws_one(int n_points, int chunk, int feat_dim) {          // buffers of ONE chunk in flight
    const size_t pts = (size_t)chunk * n_points, nblk = (n_points + 255) / 256;
    return align_up(pts * VROW * 4, 256) * 3 + align_up(pts * 4 * VROW * 4, 256) + align_up(pts * KNN * 4, 256) +
           align_up((size_t)chunk * nblk * feat_dim * 3 * 4, 256) + 2 * align_up(pts * 4, 256);
}
size_t da_pcd_encoder_workspace_bytes(int n_points, int chunk, int feat_dim) {
    // one chunk, or two half-chunks side by side (da_pcd_encoder_forward's two-stream schedule): whichever is larger
    const size_t one = pcd_ws_one(n_points, chunk, feat_dim), two = chunk >= 2 ? 2 * align_up(pcd_ws_one(n_points, chunk / 2, feat_dim), 256) : 0;
    return one > two ? one : two;
}

int da_pcd_encoder_forward(const da_pcd_encoder_weights *w, int n_parts, int n_points, const float *points, int inv,
                           float *out, int ld_out, void *workspace, size_t workspace_bytes, int chunk, void *stream) {
    DA_REQUIRE(w && points && out && workspace, "da_pcd_encoder_forward: null argument");
    DA_REQUIRE(n_parts > 0 && chunk > 0 && n_points >= KNN, "da_pcd_encoder_forward: need >= %d points per fragment", KNN);
    const int feat = w->feat_dim;
    DA_REQUIRE(feat > 0 && feat <= 256, "da_pcd_encoder_forward: feat_dim %d outside 1..256", feat);
    DA_REQUIRE(!inv || w->linear0, "da_pcd_encoder_forward: the invariant output needs linear0");
    DA_REQUIRE(ld_out >= (inv ? 2 * feat : 6 * feat), "da_pcd_encoder_forward: ld_out too small");
    for (int s = 0; s < DA_PCD_STAGES; ++s)
        DA_REQUIRE(w->premap[s] && w->bn_a[s] && (s == 2 || w->conv_b[s]), "da_pcd_encoder_forward: stage %d weights missing", s);
    DA_REQUIRE(w->conv6, "da_pcd_encoder_forward: conv6 weights missing");
    DA_REQUIRE(workspace_bytes >= da_pcd_encoder_workspace_bytes(n_points, chunk, feat), "da_pcd_encoder_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // Two half-chunks on two streams (round 5).  Fragments [p0, p0 + h) run on the caller's stream and the next h on a side stream
    // of the library, each in its own half of the workspace (h = chunk / 2; the side stream forks behind an event on `stream` and
    // is joined before the call returns).  Measured gain: 1.2 % (12.09 -> 11.94 ms per 640 x 1000 points) -- the two big kernels
    // cannot share a CU (the neighbour search's 1024-thread blocks hold all of its vector registers and 129 KB of its LDS), so
    // only the tails and the small kernels overlap.  DA_PCD_TWO_STREAMS=0, fewer than 64 fragments, or a workspace that does not
    // hold two halves: one stream, whole chunks.
    static int two_off = -1;
    if (two_off < 0) two_off = DA_XENV("DA_PCD_TWO_STREAMS", 1) == 0 ? 1 : 0;
    const int half = chunk / 2;
    PcdSide *sd = nullptr;
    size_t sub_bytes = 0;
    if (!two_off && n_parts >= 64 && half >= 16) {
        sub_bytes = align_up(pcd_ws_one(n_points, half, feat), 256);
        if (2 * sub_bytes <= workspace_bytes) sd = pcd_side();
    }
    const int step = sd ? half : chunk;
    if (sd) {
        DA_CHECK_HIP(hipEventRecord(sd->fork, st));
        DA_CHECK_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
    }
    int lane_id = 0;
    for (int p0 = 0; p0 < n_parts; p0 += step, lane_id ^= 1) {
        const int B = n_parts - p0 < step ? n_parts - p0 : step;
        hipStream_t sq = (sd && lane_id) ? sd->s : st;
        char *base = (char *)workspace + ((sd && lane_id) ? sub_bytes : 0);
        const size_t cp = (size_t)step * n_points;
        float *X[3];
        for (int s = 0; s < 3; ++s) X[s] = (float *)(base + s * align_up(cp * VROW * 4, 256));
        float *T = (float *)(base + 3 * align_up(cp * VROW * 4, 256));
        int32_t *idx = (int32_t *)((char *)T + align_up(cp * 4 * VROW * 4, 256));
        float *partial = (float *)((char *)idx + align_up(cp * KNN * 4, 256));
        float *xnorm[2];                                      // |row|^2 of X[0] / X[1], written by the edge kernel of that stage
        xnorm[0] = (float *)((char *)partial + align_up((size_t)step * ((n_points + 255) / 256) * feat * 3 * 4, 256));
        xnorm[1] = (float *)((char *)xnorm[0] + align_up(cp * 4, 256));
        const int nblk = (n_points + 255) / 256;
        const long long total = (long long)B * n_points;
        const float *pts = points + (size_t)p0 * n_points * 3;
        for (int s = 0; s < 3; ++s) {
            const float *xin = s == 0 ? pts : X[s - 1];
            const int ldx = s == 0 ? 3 : VROW;
            int rc = knn_launch(B, n_points, s == 0 ? 3 : V3, xin, ldx, KNN, 0, idx, sq, s == 0 ? nullptr : xnorm[s - 1]);      // the pooling is order-free
            if (rc) return rc;
            const int nb = (int)((total + 255) / 256);
            if (s == 0) k_pcd_premap<1><<<nb, 256, 0, sq>>>(xin, ldx, w->premap[s], total, T);
            else k_pcd_premap<VC><<<nb, 256, 0, sq>>>(xin, ldx, w->premap[s], total, T);
            DA_LAUNCH_CHECK();
            const int ne = ((B + 7) / 8) * 8 * ((n_points + 127) / 128);
            float *xno = s < 2 ? xnorm[s] : nullptr;
            if (w->conv_b[s]) k_pcd_edge<true><<<ne, 128, 0, sq>>>(T, idx, w->bn_a[s], w->conv_b[s], n_points, B, X[s], xno);
            else k_pcd_edge<false><<<ne, 128, 0, sq>>>(T, idx, w->bn_a[s], nullptr, n_points, B, X[s], xno);
            DA_LAUNCH_CHECK();
        }
        k_pcd_conv6<<<dim3(nblk, B), 256, 0, sq>>>(X[0], X[1], X[2], w->conv6, feat, n_points, partial);
        DA_LAUNCH_CHECK();
        k_pcd_final<<<B, 256, 0, sq>>>(partial, nblk, feat, n_points, w->linear0, inv, out + (size_t)p0 * ld_out, ld_out);
        DA_LAUNCH_CHECK();
    }
    if (sd) {
        DA_CHECK_HIP(hipEventRecord(sd->join, sd->s));
        DA_CHECK_HIP(hipStreamWaitEvent(st, sd->join, 0));
    }
    return 0;
}

}  // extern "C"
