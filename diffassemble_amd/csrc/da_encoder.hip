// 2D piece encoder: the reference's P4 (quarter-turn) group-equivariant ResNet-18 over 32x32 piece crops,
// inference (eval-mode BatchNorm), SURVEY.md 8f rank 2.
//
// Replaces (paths under /root/reference/puzzle_diff/model/backbones/):
//   efficient_gat.py:149-189            Eff_GAT.visual_features for model='resnet18equiv'
//   resnet_equivariant.py:14-38,70-112  BasicBlock / ResNet.forward (ResNet18 = [2,2,2,2], planes 32,64,64,128)
//   groupy/gconv/pytorch_gconv/splitgconv2d.py:15-22,70-92   trans_filter (index gather) + F.conv2d
// The reference rebuilds the rotated filter bank with a python index gather on EVERY forward and runs 20
// conv2d + 20 BatchNorm3d + ReLU/add launches over [B, C*4, H, W] fp32.  Here:
//   * the filter bank (rotations + stabilizer shifts), the eval-mode BatchNorm scale and the NHWC re-layout
//     are folded into packed weights ONCE per checkpoint (host: diffassemble_amd/encoder.py);
//   * activations live as zero-haloed NHWC  [B][H+2][W+2][C*4]  (act dtype), so a 3x3 tap is a constant
//     byte offset and no bounds test exists anywhere;
//   * every group convolution is ONE implicit-GEMM MFMA launch (k_conv_mfma: 128 pixels x 128 channels x
//     128 bytes of K per stage, LDS-DMA double buffer, the K walk runs tap-major over the 9 taps) with the
//     folded BatchNorm bias, the residual add and the ReLU in its epilogue;
//   * the 3-channel stem is a VALU kernel (K = 27), the two 544-wide output linears run through the MFMA
//     linear kernel straight on the haloed NHWC maps (halo columns of the packed weights are zero).
// Patches are processed in chunks so the working set of the 32x32 stage stays near the Infinity Cache.
#include <stdlib.h>

#include "da_gemm_common.h"

namespace da {

struct ConvParams {
    const void *X; int Cin, Hpi, Wpi;      // haloed input [B][Hpi][Wpi][Cin]
    const void *W; const float *bias;      // [Cout][taps * Cin] act dtype (K = tap-major, channel-minor), [Cout] fp32
    const void *res; void *Y;              // haloed [B][Ho + 2][Wo + 2][Cout]; res may be NULL
    int Cout, lgHo, lgWo, stride, taps, relu;
    int M;                                  // B * Ho * Wo output pixels
    int nct, lgcpt;                         // column tiles; log2(K stages per tap)
    int nvirt;                              // virtual tiles: 8 * ceil(pixel tiles / 8) * nct
    int nrt8;                               // pixel tiles per XCD: ceil(pixel tiles / 8)
    int debug;                              // DA_ENCODER_DEBUG bits (builds with -DDA_ENCODER_PROBE only; timing experiments): 1 = no A DMA, 2 = no W DMA, 4 = no MFMA
    long long tap0;                         // element offset of tap 0 from the pixel's base: 0 (3x3, pad 1) | (Wpi + 1) * Cin (1x1, pad 0)
};

__device__ __forceinline__ u32x4 add_relu8(u32x4 a, u32x4 b, bool relu, float) {          // fp32: 4 values
    f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] += y[i]; if (relu) x[i] = fmaxf(x[i], 0.f); }
    return __builtin_bit_cast(u32x4, x);
}
__device__ __forceinline__ u32x4 add_relu8(u32x4 a, u32x4 b, bool relu, bf16_t) {         // bf16: 8 values
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo = __builtin_bit_cast(float, a[i] << 16) + __builtin_bit_cast(float, b[i] << 16);
        float hi = __builtin_bit_cast(float, a[i] & 0xffff0000u) + __builtin_bit_cast(float, b[i] & 0xffff0000u);
        if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
        const bf16x2 pk = {(__bf16)lo, (__bf16)hi};              // v_cvt_pk_bf16_f32 (round to nearest even)
        o[i] = __builtin_bit_cast(unsigned, pk);
    }
    return o;
}

// Implicit-GEMM group convolution: 128 pixels x 128 channels per tile, 4 waves as 2 x 2 (64 x 64 each), K walked
// tap-major in 128-byte stages.
//
// What bounds it (ablation on the 32x32 layers, DA_ENCODER_DEBUG, two-slot version): DMA stream alone 325 us,
// MFMA + fragment reads alone 241 us, everything 351 us -- the K walk waits on the LATENCY of the single DMA
// stage it has in flight (a stage is ~0.45 us of MFMA; a round trip through L2 / MALL / HBM under load is
// longer), not on the matrix pipe.  Hence the asymmetric LDS ring: the activation tile (first touches come
// from HBM) has THREE 16 KB slots and is fetched two stages ahead, the weight tile (always L2-hot) two slots,
// one stage ahead: 80 KB per workgroup = exactly two workgroups per CU in the 160 KB LDS.  vmcnt retires in
// order, so W(s+1) is issued BEFORE A(s+2) and the wait at the top of a stage is vmcnt(4): everything but
// the four youngest DMA instructions (= A(s+2)) has landed.
// Tried and dropped: 256 x 128 tiles with 128 x 64 wave tiles (3/4 of the LDS bytes per FLOP but one wave per
// SIMD: 391 us), 8-wave 256-pixel tiles (370 us), a "row slab" variant for the stride-1 layers that fetches the
// input rows of one filter row once, three stages ahead, and reads the three kx operands from it at a
// one-pixel offset (1/3 of the activation DMA: 329 vs 331 us -- what is left is the weight tile, one stage
// ahead; a third weight slot does not fit twice per CU), starting half of the workgroups out of phase (+-0),
// 256 x 128 tiles with 128 x 64 wave tiles on 64-byte stages in a three-slot ring at two workgroups per CU (3/4
// of the LDS and DMA bytes per FLOP, both operands two stages ahead: 352 vs 340 us).  With the DMA compiled out
// the kernel runs at 1 280 - 1 400 TFLOP/s, which is also what hipBLASLt reaches on this box (1 239): that, not
// 2.5 PFLOP/s, is the practical ceiling of the matrix pipe here; the DMA stream costs the remaining 25 - 30 %.
template <typename T>
__global__ __launch_bounds__(256, 2) void k_conv_mfma(ConvParams p) {
    constexpr int SA = 16384, OFFW = 3 * SA;                 // A ring: 3 x 16 KB, W ring: 2 x 16 KB behind it
    __shared__ __attribute__((aligned(16))) unsigned char smem[5 * SA];
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int ES = (int)sizeof(T), EPC = 16 / ES, BK = 128 / ES;
    constexpr int ROWS = ES == 4 ? 32 : 64, PASSES = 128 / ROWS, CPR = 128 * ES / 16, NIT = ROWS * CPR / 256;
    constexpr int RSO = 128 * ES + 16;
    static_assert(ROWS * RSO <= 2 * SA, "epilogue staging lives in A slots 1-2");
    constexpr bool PREFETCH = ES == 2;           // fp32 (parity mode): 64 more registers would spill; fetched in the epilogue
    const int Ho = 1 << p.lgHo, Wo = 1 << p.lgWo;
    const int K = p.taps * p.Cin;
    const int nk = p.taps << p.lgcpt;
    const bool relu = p.relu != 0, has_res = p.res != nullptr;
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;

    // PERSISTENT workgroups: 2 per CU, each walks virtual tiles v, v + grid, ...  A tile's first stage is put in
    // flight BEFORE the previous tile's epilogue and its stores drain under the next tile's K walk.
    // Virtual tile -> (pixel tile, channel tile), XCD-aware (workgroups are dispatched round-robin over the 8 XCDs and
    // the grid is a multiple of 8 * nct, so a workgroup's XCD and channel tile never change): the channel tiles of
    // one pixel tile sit on the same XCD, and every XCD owns a CONTIGUOUS range of pixel tiles, so the 64 tiles
    // its workgroups hold at any time are neighbours -- adjacent row bands of the same pieces share their halo
    // rows in that XCD's L2 (with tiles dealt round-robin FETCH_SIZE was 1.8x the compulsory input bytes: every
    // band's two halo rows were fetched from HBM once per XCD).
    const int V = p.nvirt;
    auto tile_of = [&](int v) { return (v & 7) * p.nrt8 + (v >> 3) / p.nct; };      // XCD v & 7 owns a CONTIGUOUS range of pixel tiles
    auto valid = [&](int v) { return tile_of(v) * 128 < p.M; };
    int v = blockIdx.x;
    while (v < V && !valid(v)) v += gridDim.x;
    if (v >= V) return;
    const int col0 = ((v >> 3) % p.nct) * 128;

    unsigned ap[4], wp[4];                   // byte offsets from p.X / p.W (a chunk's maps stay below 4 GB, see launch_conv)
    int row0 = 0;
    auto setup = [&](int vv) {
        row0 = tile_of(vv) * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = min(row0 + 32 * wid + 8 * j + lr, p.M - 1);
            const int b = m >> (p.lgHo + p.lgWo), y = (m >> p.lgWo) & (Ho - 1), x = m & (Wo - 1);
            const size_t px = ((size_t)b * p.Hpi + (size_t)(y * p.stride)) * p.Wpi + (size_t)(x * p.stride);
            ap[j] = (unsigned)((px * p.Cin + (size_t)p.tap0) * ES + lc * 16);
        }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j)
        wp[j] = (unsigned)((size_t)min(col0 + 32 * wid + 8 * j + lr, p.Cout - 1) * K * ES + lc * 16);
    // LDS-DMA: wave w fills rows [32w, 32w+32) of a tile, 8 rows (1 KB) per instruction; the destination is
    // lane-linear, so the XOR swizzle (16-byte chunk ^ (row & 7)) is applied to the per-lane SOURCE address
    auto issueA = [&](int s) {
        const int tap = s >> p.lgcpt, cb = s - (tap << p.lgcpt);
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;              // tap / 3 for tap < 9
        const char *xa = (const char *)p.X + ((size_t)(ky * p.Wpi + kx) * p.Cin + (size_t)cb * BK) * ES;
        unsigned char *sa = smem + (s % 3) * SA + (32 * wid) * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#ifdef DA_ENCODER_PROBE
            if (!(p.debug & 1))
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xa + ap[j]),
                                             (__attribute__((address_space(3))) void *)(sa + j * 1024), 16, 0, 0);
    };
    auto issueW = [&](int s) {
        const char *xw = (const char *)p.W + (size_t)s * 128;
        unsigned char *sw = smem + OFFW + (s & 1) * SA + (32 * wid) * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#ifdef DA_ENCODER_PROBE
            if (!(p.debug & 2))
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xw + wp[j]),
                                             (__attribute__((address_space(3))) void *)(sw + j * 1024), 16, 0, 0);
    };
    // output offsets / residual tile in the layout of the coalesced store phase; the residual is fetched at the
    // START of a tile and sits in registers under its whole K walk
    unsigned ooff[PASSES][NIT];              // element offsets into p.Y / p.res
    u32x4 rv[PASSES][NIT];
    int orow0 = 0;
    auto fetch_res = [&]() {
        orow0 = row0;
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * 256, row = idx / CPR, ch = idx - row * CPR;
                const int m = min(row0 + row + pass * ROWS, p.M - 1), col = col0 + ch * EPC;
                const int b = m >> (p.lgHo + p.lgWo), y = (m >> p.lgWo) & (Ho - 1), x = m & (Wo - 1);
                ooff[pass][it] = (unsigned)((((size_t)b * (Ho + 2) + (y + 1)) * (Wo + 2) + (x + 1)) * p.Cout + col);
                rv[pass][it] = (u32x4){0u, 0u, 0u, 0u};
                if (PREFETCH && has_res) rv[pass][it] = *(const u32x4 *)((const T *)p.res + ooff[pass][it]);
            }
    };
    float bz[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const f32x4 b4 = *(const f32x4 *)(p.bias + col0 + wn * 64 + ni * 16 + (lane >> 4) * 4);
        bz[ni][0] = b4[0]; bz[ni][1] = b4[1]; bz[ni][2] = b4[2]; bz[ni][3] = b4[3];
    }
    constexpr int NRES = PASSES * NIT;

    setup(v);
    issueW(0);
    issueA(0);
    for (;;) {
        // VMEM order at this point: W(0), A(0) [, epilogue stores of the previous tile]; now A(1), residual
        if (nk > 1) issueA(1);
        fetch_res();
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int s = 0; s < nk; ++s) {
            // A(s), W(s) landed; the youngest DMA group -- A(s+1), four instructions -- may stay in flight, and at
            // s == 0 so may the residual loads issued behind it
            if (s == 0 && PREFETCH && has_res) {
                if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES + 4) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES) : "memory");
            } else if (s + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                         // + everyone is done with stage s - 1 (its slots are refilled now)
            if (s + 1 < nk) issueW(s + 1);
            if (s + 2 < nk) issueA(s + 2);
#ifdef DA_ENCODER_PROBE
            if (!(p.debug & 4))
#endif
            mma_block<T>(smem + (s % 3) * SA, smem + OFFW + (s & 1) * SA, wm, wn, lane, acc);
        }

        int vn = v + gridDim.x;
        while (vn < V && !valid(vn)) vn += gridDim.x;
        const bool more = vn < V;
        __syncthreads();                             // every slot is free
        if (more) {
            setup(vn);                               // every DMA of this tile has been issued: ap[] is free
            issueW(0);
            issueA(0);                               // A slot 0, W slot 0; the staging area below is A slots 1-2
        }

        // epilogue: (+ folded BatchNorm bias) -> LDS -> coalesced 16-byte stores of whole pixel rows, the
        // residual is added (and the ReLU applied) on the coalesced side.  Plain barriers: the DMA in flight is
        // the next tile's stage 0, which must NOT be waited for here.
        unsigned char *stg = smem + SA;
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            if (pass > 0) __syncthreads();
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int rg = wm * 64 + mi * 16 + (lane & 15);
                    if (rg / ROWS != pass) continue;                // wave-uniform
                    float vv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vv[r] = acc[mi][ni][r] + bz[ni][r];
                        if (relu && !has_res) vv[r] = fmaxf(vv[r], 0.f);
                    }
                    store4((T *)(stg + (rg % ROWS) * RSO) + wn * 64 + ni * 16 + (lane >> 4) * 4, vv);
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * 256, row = idx / CPR, ch = idx - row * CPR;
                if (orow0 + row + pass * ROWS >= p.M) continue;
                u32x4 val = *(const u32x4 *)(stg + row * RSO + ch * 16);
                if (has_res) val = add_relu8(val, PREFETCH ? rv[pass][it] : *(const u32x4 *)((const T *)p.res + ooff[pass][it]), relu, T());
                *(u32x4 *)((T *)p.Y + ooff[pass][it]) = val;
            }
        }
        if (!more) break;
        v = vn;
        __syncthreads();                             // staging area (A slots 1-2) read out before A(1) lands in it
    }
}

// Stem: normalise (efficient_gat.py:150), P4ConvZ2(3 -> 32 planes x 4 rotations, 3x3, pad 1) with the
// BatchNorm folded, ReLU (resnet_equivariant.py:96) -> haloed NHWC [B][34][34][128].  One workgroup per
// patch; a lane owns 4 output channels (its 108 weights stay in registers) and 32 lanes cover one pixel, so
// every store instruction writes two whole pixels (2 x 256 / 512 B contiguous).
template <typename T>
__global__ __launch_bounds__(256) void k_enc_stem(const float *__restrict__ patches, const float *__restrict__ w,
                                                  const float *__restrict__ bias, T *__restrict__ Y, int relu) {
    __shared__ float win[3 * 34 * 34];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float mean[3] = {0.4850f, 0.4560f, 0.4060f}, sd[3] = {0.2290f, 0.2240f, 0.2250f};
    for (int i = tid; i < 3 * 34 * 34; i += 256) {
        const int c = i / 1156, r = i - c * 1156, y = r / 34, x = r - y * 34;
        float v = 0.f;                                          // the conv pads the NORMALISED image with zeros
        if (y >= 1 && y <= 32 && x >= 1 && x <= 32)
            v = (patches[(((size_t)b * 3 + c) * 32 + (y - 1)) * 32 + (x - 1)] - mean[c]) / sd[c];
        win[i] = v;
    }
    const int l32 = tid & 31, sub = tid >> 5;                    // 8 pixels in flight per iteration
    float wr[4][27], bz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bz[q] = bias[4 * l32 + q];
#pragma unroll
        for (int t = 0; t < 27; ++t) wr[q][t] = w[(4 * l32 + q) * 27 + t];
    }
    __syncthreads();
    for (int it = 0; it < 128; ++it) {
        const int pix = it * 8 + sub, y = pix >> 5, x = pix & 31;
        float acc[4] = {bz[0], bz[1], bz[2], bz[3]};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = win[c * 1156 + (y + ky) * 34 + x + kx];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(v, wr[q][c * 9 + ky * 3 + kx], acc[q]);
                }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = relu ? fmaxf(acc[q], 0.f) : acc[q];
        store4(Y + (((size_t)b * 34 + (y + 1)) * 34 + (x + 1)) * 128 + 4 * l32, acc);
    }
}

static int lg2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

int launch_conv(int prec, int B, const void *X, int Cin, int Hi, const void *W, const float *bias, const void *res,
                       void *Y, int Cout, int ksize, int stride, int relu, hipStream_t st) {
    const int es = (int)esize(prec), BK = 128 / es;
    ConvParams p;
    const int Ho = Hi / stride;
    p.X = X; p.Cin = Cin; p.Hpi = Hi + 2; p.Wpi = Hi + 2; p.W = W; p.bias = bias; p.res = res; p.Y = Y;
    p.Cout = Cout; p.lgHo = lg2(Ho); p.lgWo = p.lgHo; p.stride = stride; p.taps = ksize * ksize; p.relu = relu;
    p.M = B * Ho * Ho; p.nct = Cout / 128; p.lgcpt = lg2(Cin / BK);
    p.tap0 = ksize == 3 ? 0 : (long long)(p.Wpi + 1) * Cin;
    DA_REQUIRE((size_t)B * (Hi + 2) * (Hi + 2) * Cin * es < ((size_t)1 << 32) && (size_t)B * (Ho + 2) * (Ho + 2) * Cout * es < ((size_t)1 << 32),
               "encoder conv: a chunk's feature map must stay below 4 GB (32-bit offsets): use a smaller chunk");
    DA_REQUIRE(Cin % BK == 0 && (1 << p.lgcpt) == Cin / BK && Cout % 128 == 0 && (1 << p.lgHo) == Ho && (ksize == 1 || ksize == 3),
               "encoder conv: unsupported geometry (Cin %d Cout %d H %d k %d)", Cin, Cout, Hi, ksize);
    p.debug = DA_XENV("DA_ENCODER_DEBUG", 0);
    const int nrt = (p.M + 127) / 128;
    p.nrt8 = (nrt + 7) / 8;
    p.nvirt = 8 * p.nrt8 * p.nct;
    int per_cu = DA_XENV("DA_ENCODER_WG_PER_CU", 2);
    if (per_cu < 1) per_cu = 1000000;
    // persistent: at most 256 CUs x 2 resident workgroups, a multiple of 8 * nct (see the kernel)
    long long cap = (long long)256 * per_cu / (8 * p.nct) * (8 * p.nct);
    if (cap < 8 * p.nct) cap = 8 * p.nct;
    const unsigned grid = (unsigned)(p.nvirt < cap ? p.nvirt : cap);
    if (prec == DA_PREC_BF16) k_conv_mfma<bf16_t><<<grid, 256, 0, st>>>(p);
    else k_conv_mfma<float><<<grid, 256, 0, st>>>(p);
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_enc_stem(int prec, int B, const float *patches, const float *w, const float *bias, void *Y, int relu, hipStream_t st) {
    if (prec == DA_PREC_BF16) k_enc_stem<bf16_t><<<B, 256, 0, st>>>(patches, w, bias, (bf16_t *)Y, relu);
    else k_enc_stem<float><<<B, 256, 0, st>>>(patches, w, bias, (float *)Y, relu);
    DA_LAUNCH_CHECK();
    return 0;
}

// elements per patch of the haloed maps
constexpr size_t P32 = 34 * 34 * 128, P16 = 18 * 18 * 256, P8 = 10 * 10 * 256, P4 = 6 * 6 * 512;

}  // namespace da

using namespace da;

extern "C" {

size_t da_encoder_workspace_bytes(int precision, int n_patches, int chunk) {
    if (n_patches <= 0 || chunk <= 0) return 0;
    const size_t es = esize(precision);
    return align_up((size_t)chunk * 3 * (P32 + P16 + P8 + P4) * es, 256) + align_up((size_t)n_patches * (P8 + P4) * es, 256);
}

int da_encoder_forward(int precision, const da_encoder_weights *w, int n_patches, const float *patches, void *feats,
                       int ld_feats, void *workspace, size_t workspace_bytes, int chunk, int zero_workspace, void *stream) {
    DA_REQUIRE(precision == DA_PREC_F32 || precision == DA_PREC_BF16, "da_encoder_forward: bad precision");
    DA_REQUIRE(w && patches && feats && workspace, "da_encoder_forward: null argument");
    DA_REQUIRE(n_patches > 0 && chunk > 0 && ld_feats >= DA_ENCODER_FEATS, "da_encoder_forward: bad sizes");
    DA_REQUIRE(w->n_convs == DA_ENCODER_CONVS, "da_encoder_forward: expected %d packed convolutions", DA_ENCODER_CONVS);
    DA_REQUIRE(workspace_bytes >= da_encoder_workspace_bytes(precision, n_patches, chunk), "da_encoder_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t es = esize(precision);
    // the halo of every map must be zero; the kernels only ever write interiors, so one fill per workspace is enough
    if (zero_workspace) DA_CHECK_HIP(hipMemsetAsync(workspace, 0, da_encoder_workspace_bytes(precision, n_patches, chunk), st));
    char *base = (char *)workspace;
    char *a32 = base, *b32 = a32 + (size_t)chunk * P32 * es, *c32 = b32 + (size_t)chunk * P32 * es;
    char *t16 = c32 + (size_t)chunk * P32 * es, *s16 = t16 + (size_t)chunk * P16 * es, *y16 = s16 + (size_t)chunk * P16 * es;
    char *t8 = y16 + (size_t)chunk * P16 * es, *s8 = t8 + (size_t)chunk * P8 * es, *y8 = s8 + (size_t)chunk * P8 * es;
    char *t4 = y8 + (size_t)chunk * P8 * es, *s4 = t4 + (size_t)chunk * P4 * es, *y4 = s4 + (size_t)chunk * P4 * es;
    char *out3 = base + align_up((size_t)chunk * 3 * (P32 + P16 + P8 + P4) * es, 256);
    char *out4 = out3 + (size_t)n_patches * P8 * es;
#define CONV(i, B, X, Cin, Hi, RES, Y, Cout, k, s, relu)                                                              \
    do {                                                                                                              \
        const int rc_ = launch_conv(precision, B, X, Cin, Hi, w->conv_w[i], w->conv_b[i], RES, Y, Cout, k, s, relu, st); \
        if (rc_) return rc_;                                                                                          \
    } while (0)
    for (int p0 = 0; p0 < n_patches; p0 += chunk) {
        const int B = n_patches - p0 < chunk ? n_patches - p0 : chunk;
        const float *px = patches + (size_t)p0 * 3 * 32 * 32;
        if (precision == DA_PREC_BF16) k_enc_stem<bf16_t><<<B, 256, 0, st>>>(px, w->stem_w, w->stem_b, (bf16_t *)a32, 1);
        else k_enc_stem<float><<<B, 256, 0, st>>>(px, w->stem_w, w->stem_b, (float *)a32, 1);
        DA_LAUNCH_CHECK();
        char *o3 = out3 + (size_t)p0 * P8 * es, *o4 = out4 + (size_t)p0 * P4 * es;
        // layer1 (32 planes, 32x32); packed order = state-dict order: conv1, conv2 (, shortcut) per block
        CONV(0, B, a32, 128, 32, nullptr, b32, 128, 3, 1, 1);
        CONV(1, B, b32, 128, 32, a32, c32, 128, 3, 1, 1);
        CONV(2, B, c32, 128, 32, nullptr, b32, 128, 3, 1, 1);
        CONV(3, B, b32, 128, 32, c32, a32, 128, 3, 1, 1);
        // layer2 (64 planes, 16x16)
        CONV(4, B, a32, 128, 32, nullptr, t16, 256, 3, 2, 1);
        CONV(6, B, a32, 128, 32, nullptr, s16, 256, 1, 2, 0);
        CONV(5, B, t16, 256, 16, s16, y16, 256, 3, 1, 1);
        CONV(7, B, y16, 256, 16, nullptr, t16, 256, 3, 1, 1);
        CONV(8, B, t16, 256, 16, y16, s16, 256, 3, 1, 1);
        // layer3 (64 planes, 8x8) -> out3
        CONV(9, B, s16, 256, 16, nullptr, t8, 256, 3, 2, 1);
        CONV(11, B, s16, 256, 16, nullptr, s8, 256, 1, 2, 0);
        CONV(10, B, t8, 256, 8, s8, y8, 256, 3, 1, 1);
        CONV(12, B, y8, 256, 8, nullptr, t8, 256, 3, 1, 1);
        CONV(13, B, t8, 256, 8, y8, o3, 256, 3, 1, 1);
        // layer4 (128 planes, 4x4) -> out4
        CONV(14, B, o3, 256, 8, nullptr, t4, 512, 3, 2, 1);
        CONV(16, B, o3, 256, 8, nullptr, s4, 512, 1, 2, 0);
        CONV(15, B, t4, 512, 4, s4, y4, 512, 3, 1, 1);
        CONV(17, B, y4, 512, 4, nullptr, t4, 512, 3, 1, 1);
        CONV(18, B, t4, 512, 4, y4, o4, 512, 3, 1, 1);
    }
#undef CONV
    // linear1 / linear2 (resnet_equivariant.py:107-108) over all patches, written side by side:
    // feats[:, 0:544] | feats[:, 544:1088]  (efficient_gat.py:188 cat)
    int rc = launch_gemm_mfma(precision, n_patches, (int)P8, 544, out3, (int)P8, w->lin1_w, w->lin1_b, DA_ACT_NONE, nullptr,
                              feats, ld_feats, nullptr, st, 0, nullptr);
    DA_REQUIRE(rc == 0, "da_encoder_forward: linear1 launch failed (%d)", rc);
    rc = launch_gemm_mfma(precision, n_patches, (int)P4, 544, out4, (int)P4, w->lin2_w, w->lin2_b, DA_ACT_NONE, nullptr,
                          (char *)feats + 544 * es, ld_feats, nullptr, st, 0, nullptr);
    DA_REQUIRE(rc == 0, "da_encoder_forward: linear2 launch failed (%d)", rc);
    return 0;
}

}  // extern "C"
