// Training path of the 2D piece encoder ("every step in training", spatial_diffusion.py:450):
// the primitives a TRAINING-mode forward and the backward of the reference's P4 ResNet-18 are built from, fp32, over the
// same zero-haloed NHWC maps [B][H+2][H+2][planes*4] as the inference kernels (da_encoder.hip).  The network walk
// (BasicBlock order, buffer pool, gradient accumulation) is host logic in diffassemble_amd/encoder_train.py.
//
// Replaces, under /root/reference/puzzle_diff/model/backbones/:
//   resnet_equivariant.py:14-38,93-112     BasicBlock / ResNet.forward in train() mode + torch autograd through it
//   nn.BatchNorm3d(planes) on [B, planes, 4, H, W] (resnet_equivariant.py:22,26,31,75): batch statistics per plane over
//                                           (B, 4, H, W), biased variance; backward of the normalisation
//   groupy/gconv/pytorch_gconv/splitgconv2d.py:70-92   F.conv2d of the rotated filter bank (forward, dgrad, wgrad) and
//                                           the index gather trans_filter (:15-22), whose backward is a 4-way gather-sum
// How the backward maps onto existing kernels (no new matrix kernel):
//   * forward convolution, unfused: k_conv_mfma with the raw filter bank, zero bias, no ReLU (da_enc_conv);
//   * dgrad of a stride-1 convolution = the same kernel on dY with the taps flipped and the channel roles swapped
//     (host-packed [Cin4][8 - tap][Cout4]); its `res` input accumulates gradients arriving over other paths;
//   * a stride-2 convolution is the stride-1 one sampled at even pixels: dY is zero-stuffed to the input resolution
//     (k_enc_upsample2) and both dgrad and wgrad treat the layer as stride 1;
//   * wgrad: per tap, dBank[o][tap][c] = sum over pixels dY[q][o] X[q + offset(tap)][c].  Because dY's halo is ZERO, the
//     sum may run over ALL haloed positions q, i.e. it is ONE plain TN GEMM per tap on the two maps as they lie in
//     memory (X shifted by a constant element offset): k_gemm_tn (fp32 MFMA, split over rows), no im2col;
//   * filter bank -> parameter gradient: every parameter entry feeds exactly 4 bank entries (k_bank_grad, table built once).
// BatchNorm reductions are two-stage and deterministic (per-block fp32 partials over <= 4096 pixels, summed in double).
#include "da_internal.h"

namespace da {

constexpr float ENC_BN_EPS = 1e-5f;           // nn.BatchNorm3d default (resnet_equivariant.py:22)
constexpr int BN_PIX = 4096;                  // interior pixels per reduction block

struct MapGeom { int B, H, C4; };

// four consecutive channels of a map stored as fp32 or bf16 (the bf16 training mode keeps activations and activation
// gradients in bf16; all arithmetic stays fp32)
__device__ __forceinline__ float4 ld4(const float *p, size_t o) { return *(const float4 *)(p + o); }
__device__ __forceinline__ float4 ld4(const bf16_t *p, size_t o) {
    const uint2 v = *(const uint2 *)(p + o);
    return float4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                  __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
}
__device__ __forceinline__ void st4(float *p, size_t o, const float4 &v) { *(float4 *)(p + o) = v; }
__device__ __forceinline__ void st4(bf16_t *p, size_t o, const float4 &v) {
    uint2 u;
    u.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    u.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    *(uint2 *)(p + o) = u;
}            // haloed map [B][H + 2][H + 2][C4], interior H x H; plane = channel >> 2

__device__ __forceinline__ size_t pix_off(const MapGeom &g, long long ip) {      // ip: interior pixel index over (b, y, x)
    const int HH = g.H * g.H;
    const long long b = ip / HH;
    const int r = (int)(ip - b * HH), y = r / g.H, x = r - y * g.H;
    return (((size_t)b * (g.H + 2) + (y + 1)) * (g.H + 2) + (x + 1)) * g.C4;
}

// Block reduction of up to NV values per thread across the block's pixel lanes -> partial[block][plane][NV] (double).
// 256 threads = (256 / planes) pixel lanes x planes; thread (lane_p, plane) has already summed its pixels.
template <int NV>
__device__ __forceinline__ void block_plane_reduce(const float (&v)[NV], int planes, double *partial) {
    __shared__ float red[256 * NV];
    const int tid = threadIdx.x, plane = tid % planes;
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i * 256 + tid] = v[i];
    __syncthreads();
    if (tid < planes) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
            for (int l = plane; l < 256; l += planes) s += (double)red[i * 256 + l];
            partial[((size_t)blockIdx.x * planes + plane) * NV + i] = s;
        }
    }
}

// sum and sum of squares per plane over a block's interior pixels
template <typename T>
__global__ __launch_bounds__(256) void k_enc_bn_stats(MapGeom g, const T *__restrict__ Y, double *__restrict__ partial) {
    const int planes = g.C4 >> 2, lanes = 256 / planes, tid = threadIdx.x, plane = tid % planes, lp = tid / planes;
    const long long total = (long long)g.B * g.H * g.H, p0 = (long long)blockIdx.x * BN_PIX, p1 = min(total, p0 + BN_PIX);
    float v[2] = {0.f, 0.f};
#pragma unroll 4
    for (long long ip = p0 + lp; ip < p1; ip += lanes) {
        const float4 y = ld4(Y, pix_off(g, ip) + 4 * plane);
        v[0] += (y.x + y.y) + (y.z + y.w);
        v[1] += (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
    }
    block_plane_reduce<2>(v, planes, partial);
}

// partial[nblk][planes][2] -> mean, biased variance (computed in double)
__global__ __launch_bounds__(128) void k_enc_bn_finish(int nblk, int planes, double count, const double *__restrict__ partial,
                                                       float *__restrict__ mean, float *__restrict__ var) {
    const int p = threadIdx.x;
    if (p >= planes) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < nblk; ++b) { s += partial[((size_t)b * planes + p) * 2]; ss += partial[((size_t)b * planes + p) * 2 + 1]; }
    const double m = s / count;
    mean[p] = (float)m;
    var[p] = (float)fmax(ss / count - m * m, 0.0);
}

// z = (y - mean) * (rsqrt(var + eps) * gamma) + beta [+ res] [ReLU], interior only (the halo of Z stays zero)
template <typename T>
__global__ __launch_bounds__(256) void k_enc_bn_apply(MapGeom g, const T *__restrict__ Y, const float *__restrict__ mean,
                                                      const float *__restrict__ var, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, const T *res, int relu, T *Z) {
    const int planes = g.C4 >> 2;
    const long long total = (long long)g.B * g.H * g.H * planes;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int plane = (int)(i % planes);
        const size_t o = pix_off(g, i / planes) + 4 * plane;
        const float mu = mean[plane], sc = rsqrtf(var[plane] + ENC_BN_EPS) * gamma[plane], sh = beta[plane];
        float4 y = ld4(Y, o);
        y.x = (y.x - mu) * sc + sh; y.y = (y.y - mu) * sc + sh; y.z = (y.z - mu) * sc + sh; y.w = (y.w - mu) * sc + sh;
        if (res) { const float4 r = ld4(res, o); y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w; }
        if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        st4(Z, o, y);
    }
}

// backward reductions per plane: s1 = sum g, s2 = sum g * xhat, with g = dZ masked by the unit's ReLU (Z > 0)
template <typename T>
__global__ __launch_bounds__(256) void k_enc_bn_bwd_reduce(MapGeom g, const T *__restrict__ dZ, const T *__restrict__ Z,
                                                           const T *__restrict__ Y, const float *__restrict__ mean,
                                                           const float *__restrict__ var, int relu, double *__restrict__ partial) {
    const int planes = g.C4 >> 2, lanes = 256 / planes, tid = threadIdx.x, plane = tid % planes, lp = tid / planes;
    const long long total = (long long)g.B * g.H * g.H, p0 = (long long)blockIdx.x * BN_PIX, p1 = min(total, p0 + BN_PIX);
    const float mu = mean[plane], rstd = rsqrtf(var[plane] + ENC_BN_EPS);
    float v[2] = {0.f, 0.f};
#pragma unroll 4
    for (long long ip = p0 + lp; ip < p1; ip += lanes) {
        const size_t o = pix_off(g, ip) + 4 * plane;
        float4 d = ld4(dZ, o);
        const float4 y = ld4(Y, o);
        if (relu) {
            const float4 z = ld4(Z, o);
            d.x = z.x > 0.f ? d.x : 0.f; d.y = z.y > 0.f ? d.y : 0.f; d.z = z.z > 0.f ? d.z : 0.f; d.w = z.w > 0.f ? d.w : 0.f;
        }
        v[0] += (d.x + d.y) + (d.z + d.w);
        v[1] += (d.x * ((y.x - mu) * rstd) + d.y * ((y.y - mu) * rstd)) + (d.z * ((y.z - mu) * rstd) + d.w * ((y.w - mu) * rstd));
    }
    block_plane_reduce<2>(v, planes, partial);
}

// s1 / s2 per plane (double sums of the partials); dbeta += s1, dgamma += s2
__global__ __launch_bounds__(128) void k_enc_bn_bwd_finish(int nblk, int planes, const double *__restrict__ partial,
                                                           float *__restrict__ s12, float *dgamma, float *dbeta) {
    const int p = threadIdx.x;
    if (p >= planes) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) { s1 += partial[((size_t)b * planes + p) * 2]; s2 += partial[((size_t)b * planes + p) * 2 + 1]; }
    s12[p] = (float)s1; s12[planes + p] = (float)s2;
    dbeta[p] += (float)s1; dgamma[p] += (float)s2;
}

// dY = gamma * rstd * (g - s1 / n - xhat * s2 / n);  dRes = g (the gradient of the tensor added before the ReLU), optional
template <typename T>
__global__ __launch_bounds__(256) void k_enc_bn_bwd_apply(MapGeom g, const T *dZ, const T *__restrict__ Z,
                                                          const T *__restrict__ Y, const float *__restrict__ mean,
                                                          const float *__restrict__ var, const float *__restrict__ gamma,
                                                          const float *__restrict__ s12, float inv_n, int relu, T *dY,
                                                          T *dRes) {
    const int planes = g.C4 >> 2;
    const long long total = (long long)g.B * g.H * g.H * planes;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int plane = (int)(i % planes);
        const size_t o = pix_off(g, i / planes) + 4 * plane;
        const float mu = mean[plane], rstd = rsqrtf(var[plane] + ENC_BN_EPS), k = gamma[plane] * rstd;
        const float m1 = s12[plane] * inv_n, m2 = s12[planes + plane] * inv_n;
        float4 d = ld4(dZ, o);
        const float4 y = ld4(Y, o);
        if (relu) {
            const float4 z = ld4(Z, o);
            d.x = z.x > 0.f ? d.x : 0.f; d.y = z.y > 0.f ? d.y : 0.f; d.z = z.z > 0.f ? d.z : 0.f; d.w = z.w > 0.f ? d.w : 0.f;
        }
        if (dRes) st4(dRes, o, d);
        float4 r;
        r.x = k * (d.x - m1 - (y.x - mu) * rstd * m2); r.y = k * (d.y - m1 - (y.y - mu) * rstd * m2);
        r.z = k * (d.z - m1 - (y.z - mu) * rstd * m2); r.w = k * (d.w - m1 - (y.w - mu) * rstd * m2);
        st4(dY, o, r);
    }
}

// zero-stuffing: Up (interior 2H x 2H) gets S(i, j) at interior (2i, 2j) and zeros elsewhere (halo untouched = zero)
template <typename T>
__global__ __launch_bounds__(256) void k_enc_upsample2(MapGeom g /* of the SMALL map */, const T *__restrict__ S, T *__restrict__ Up) {
    const int q4 = g.C4 >> 2, H2 = 2 * g.H;
    const long long total = (long long)g.B * H2 * H2 * q4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % q4);
        const long long ip = i / q4;
        const long long b = ip / ((long long)H2 * H2);
        const int r = (int)(ip - b * H2 * H2), y = r / H2, x = r - y * H2;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(y & 1) && !(x & 1))
            v = ld4(S, (((size_t)b * (g.H + 2) + (y / 2 + 1)) * (g.H + 2) + (x / 2 + 1)) * g.C4 + 4 * c);
        st4(Up, (((size_t)b * (H2 + 2) + (y + 1)) * (H2 + 2) + (x + 1)) * g.C4 + 4 * c, v);
    }
}

// im2col of the NORMALISED 3-channel crop for the stem's wgrad: cols [B][34][34][32], taps c*9 + ky*3 + kx in 0..26 at
// interior pixels, zero elsewhere (the conv pads the normalised image with zeros, efficient_gat.py:150)
template <typename T>
__global__ __launch_bounds__(256) void k_enc_stem_im2col(int B, const float *__restrict__ patches, T *__restrict__ cols) {
    const float mean[3] = {0.4850f, 0.4560f, 0.4060f}, sd[3] = {0.2290f, 0.2240f, 0.2250f};
    const long long total = (long long)B * 34 * 34 * 32;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int t = (int)(i & 31);
        const long long q = i >> 5;
        const int b = (int)(q / 1156), r = (int)(q - (long long)b * 1156), yh = r / 34, xh = r - yh * 34;
        float v = 0.f;
        if (t < 27 && yh >= 1 && yh <= 32 && xh >= 1 && xh <= 32) {
            const int c = t / 9, ky = (t - c * 9) / 3, kx = t - c * 9 - ky * 3;
            const int y = yh - 1 + ky - 1, x = xh - 1 + kx - 1;
            if (y >= 0 && y < 32 && x >= 0 && x < 32) v = (patches[(((size_t)b * 3 + c) * 32 + y) * 32 + x] - mean[c]) / sd[c];
        }
        stf(cols + i, v);
    }
}

// dW[i] (+)= sum_r dBank[table[i][r]], r < 4: the backward of trans_filter's index gather
__global__ __launch_bounds__(256) void k_enc_bank_grad(int n, const int32_t *__restrict__ table, const float *__restrict__ dbank,
                                                       float *__restrict__ dW) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 t = *(const int4 *)(table + 4 * (size_t)i);
    dW[i] += (dbank[t.x] + dbank[t.y]) + (dbank[t.z] + dbank[t.w]);
}

static unsigned grid_for(long long total) { const long long b = (total + 255) / 256; return (unsigned)(b < 65536 ? b : 65536); }

}  // namespace da

using namespace da;

extern "C" {

size_t da_enc_train_scratch_bytes(int n_patches) {
    const size_t nblk = ((size_t)n_patches * 1024 + BN_PIX - 1) / BN_PIX;
    return align_up(nblk * 128 * 2 * sizeof(double), 256) + 256 * sizeof(float) + ((size_t)16 << 20) * sizeof(float);
}

int da_enc_conv(int precision, int B, const void *X, int Cin, int Hi, const void *W, const float *bias, const void *res, void *Y,
                int Cout, int ksize, int stride, int relu, void *stream) {
    DA_REQUIRE(X && W && bias && Y && B > 0, "da_enc_conv: null argument");
    return launch_conv(precision, B, X, Cin, Hi, W, bias, res, Y, Cout, ksize, stride, relu, (hipStream_t)stream);
}

int da_enc_stem(int precision, int B, const float *patches, const float *w, const float *bias, void *Y, int relu, void *stream) {
    DA_REQUIRE(patches && w && bias && Y && B > 0, "da_enc_stem: null argument");
    return launch_enc_stem(precision, B, patches, w, bias, Y, relu, (hipStream_t)stream);
}

int da_enc_stem_im2col(int precision, int B, const float *patches, void *cols, void *stream) {
    DA_REQUIRE(patches && cols && B > 0, "da_enc_stem_im2col: null argument");
    if (precision == DA_PREC_BF16) k_enc_stem_im2col<bf16_t><<<grid_for((long long)B * 1156 * 32), 256, 0, (hipStream_t)stream>>>(B, patches, (bf16_t *)cols);
    else k_enc_stem_im2col<float><<<grid_for((long long)B * 1156 * 32), 256, 0, (hipStream_t)stream>>>(B, patches, (float *)cols);
    DA_LAUNCH_CHECK();
    return 0;
}

static int geom_ok(int B, int H, int C4) { return B > 0 && H > 0 && C4 >= 128 && C4 <= 512 && C4 % 128 == 0; }

int da_enc_bn_stats(int precision, int B, int H, int C4, const void *Y, float *mean, float *var, void *scratch, void *stream) {
    DA_REQUIRE(Y && mean && var && scratch && geom_ok(B, H, C4), "da_enc_bn_stats: bad argument");
    const MapGeom g{B, H, C4};
    const long long total = (long long)B * H * H;
    const int nblk = (int)((total + BN_PIX - 1) / BN_PIX);
    hipStream_t st = (hipStream_t)stream;
    if (precision == DA_PREC_BF16) k_enc_bn_stats<bf16_t><<<nblk, 256, 0, st>>>(g, (const bf16_t *)Y, (double *)scratch);
    else k_enc_bn_stats<float><<<nblk, 256, 0, st>>>(g, (const float *)Y, (double *)scratch);
    k_enc_bn_finish<<<1, 128, 0, st>>>(nblk, C4 / 4, (double)total * 4.0, (const double *)scratch, mean, var);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_enc_bn_apply(int precision, int B, int H, int C4, const void *Y, const float *mean, const float *var, const float *gamma,
                    const float *beta, const void *res, int relu, void *Z, void *stream) {
    DA_REQUIRE(Y && mean && var && gamma && beta && Z && geom_ok(B, H, C4), "da_enc_bn_apply: bad argument");
    const MapGeom g{B, H, C4};
    const unsigned grid = grid_for((long long)B * H * H * (C4 / 4));
    if (precision == DA_PREC_BF16)
        k_enc_bn_apply<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const bf16_t *)Y, mean, var, gamma, beta, (const bf16_t *)res, relu, (bf16_t *)Z);
    else
        k_enc_bn_apply<float><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const float *)Y, mean, var, gamma, beta, (const float *)res, relu, (float *)Z);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_enc_bn_backward(int precision, int B, int H, int C4, const void *dZ, const void *Z, const void *Y, const float *mean,
                       const float *var, const float *gamma, int relu, float *dgamma, float *dbeta, void *dY, void *dRes,
                       void *scratch, void *stream) {
    DA_REQUIRE(dZ && Y && mean && var && gamma && dgamma && dbeta && dY && scratch && (Z || !relu) && geom_ok(B, H, C4),
               "da_enc_bn_backward: bad argument");
    const MapGeom g{B, H, C4};
    const long long total = (long long)B * H * H;
    const int nblk = (int)((total + BN_PIX - 1) / BN_PIX), planes = C4 / 4;
    hipStream_t st = (hipStream_t)stream;
    double *partial = (double *)scratch;
    float *s12 = (float *)((char *)scratch + align_up((size_t)nblk * 128 * 2 * sizeof(double), 256));
    const float inv_n = (float)(1.0 / ((double)total * 4.0));
    if (precision == DA_PREC_BF16) {
        typedef const bf16_t *P;
        k_enc_bn_bwd_reduce<bf16_t><<<nblk, 256, 0, st>>>(g, (P)dZ, (P)Z, (P)Y, mean, var, relu, partial);
        k_enc_bn_bwd_finish<<<1, 128, 0, st>>>(nblk, planes, partial, s12, dgamma, dbeta);
        k_enc_bn_bwd_apply<bf16_t><<<grid_for(total * planes), 256, 0, st>>>(g, (P)dZ, (P)Z, (P)Y, mean, var, gamma, s12, inv_n, relu,
                                                                            (bf16_t *)dY, (bf16_t *)dRes);
    } else {
        typedef const float *P;
        k_enc_bn_bwd_reduce<float><<<nblk, 256, 0, st>>>(g, (P)dZ, (P)Z, (P)Y, mean, var, relu, partial);
        k_enc_bn_bwd_finish<<<1, 128, 0, st>>>(nblk, planes, partial, s12, dgamma, dbeta);
        k_enc_bn_bwd_apply<float><<<grid_for(total * planes), 256, 0, st>>>(g, (P)dZ, (P)Z, (P)Y, mean, var, gamma, s12, inv_n, relu,
                                                                           (float *)dY, (float *)dRes);
    }
    DA_LAUNCH_CHECK();
    return 0;
}

int da_enc_upsample2(int precision, int B, int H, int C4, const void *S, void *Up, void *stream) {
    DA_REQUIRE(S && Up && geom_ok(B, H, C4), "da_enc_upsample2: bad argument");
    const MapGeom g{B, H, C4};
    const unsigned grid = grid_for((long long)B * 4 * H * H * (C4 / 4));
    if (precision == DA_PREC_BF16) k_enc_upsample2<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const bf16_t *)S, (bf16_t *)Up);
    else k_enc_upsample2<float><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const float *)S, (float *)Up);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_gemm_tn_f32(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, void *scratch,
                   void *stream) {
    DA_REQUIRE(A && B && C && scratch && M > 0 && N > 0 && K > 0, "da_gemm_tn_f32: bad argument");
    return launch_gemm_tn(M, N, K, A, lda, B, ldb, C, ldc, (float *)scratch, (hipStream_t)stream);
}

int da_gemm_tn_bf16(int M, int N, int K, const void *A, int lda, const void *B, int ldb, float *C, int ldc, void *scratch,
                    void *stream) {
    DA_REQUIRE(A && B && C && scratch && M > 0 && N > 0 && K > 0, "da_gemm_tn_bf16: bad argument");
    DA_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ((size_t)A & 15) == 0 && ((size_t)B & 15) == 0, "da_gemm_tn_bf16: operands must be 16-byte aligned with row strides that are multiples of 8");
    return launch_gemm_tn_bf16(M, N, K, (const bf16_t *)A, lda, (const bf16_t *)B, ldb, C, ldc, (float *)scratch, (hipStream_t)stream);
}

int da_colsum_f32(int M, int N, const float *A, int lda, float *out, void *scratch, void *stream) {
    DA_REQUIRE(A && out && scratch && M > 0 && N > 0, "da_colsum_f32: bad argument");
    return colsum_add(M, N, A, lda, out, (float *)scratch, (hipStream_t)stream);
}

int da_enc_bank_grad(int n, const int32_t *table, const float *dbank, float *dW, void *stream) {
    DA_REQUIRE(table && dbank && dW && n > 0, "da_enc_bank_grad: bad argument");
    k_enc_bank_grad<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, table, dbank, dW);
    DA_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
