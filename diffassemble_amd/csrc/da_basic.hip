// Small / simple kernels of the denoiser path: feature staging, pose+timestep embedding,
// the reference-grade (VALU, fp32-accumulate) linear kernel, the 2D pose head and the
// DDIM / DDPM update.  All are HBM- or latency-bound helpers around the two hot kernels
// (MFMA linear: da_gemm_mfma.hip, attention: da_attn_*.hip).
#include "da_common.h"
#include "da_internal.h"

namespace da {

// ------------------------------------------------------------------------------------------
// comb_in[r, 0:F] = feats[r, :]  (once per Batch; features are loop invariant)
template <typename T>
__global__ __launch_bounds__(256) void k_set_feats(int n, int F, int D, const float *__restrict__ feats,
                                                   T *__restrict__ comb_in) {
    size_t total = (size_t)n * F;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx / F;
        int c = (int)(idx - r * F);
        stf(comb_in + r * D + c, feats[idx]);
    }
}

// dst[r, :] = virt_emb[r % V, :] for the V*G appended virtual rows (exophormer_gnn.py:169-178:
// index vector arange(V).repeat(G))
template <typename T>
__global__ __launch_bounds__(256) void k_set_virtual_rows(int rows, int V, int D, const T *__restrict__ emb,
                                                          T *__restrict__ dst) {
    size_t total = (size_t)rows * D;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx / D;
        int c = (int)(idx - r * D);
        dst[idx] = emb[(size_t)(r % V) * D + c];
    }
}

// comb_in[r, F:F+32] = pos_mlp(x[r]) ; comb_in[r, F+32:F+64] = time_emb[t[r]]
// (efficient_gat.py:131-134).  A wave works through NPW consecutive nodes with its weight rows in registers (lane j < 16:
// row j of the first linear, lane o < 32: row o of the second): one wave per node spent its life on launch, 16 weight
// loads per lane and a workgroup barrier (57 600 waves, 23.7 us per step at 64 puzzles).  The 16 hidden activations cross
// lanes through a wave-private LDS row; same operation order as before.
// Workgroups behind the embedding's own (vs.rows > 0): the exophormer's virtual rows' conv-0 projections into the head-major buffers
// (k_scatter_virtual's job, a launch of its own on the critical path of small Batches otherwise).
template <typename T>
__global__ __launch_bounds__(256) void k_embed_pos_time(int n, int NPW, int c_in, int F, int D, const float *__restrict__ x,
                                                        const int64_t *__restrict__ t, int64_t t_scalar, int steps,
                                                        const float *__restrict__ time_emb,
                                                        const float *__restrict__ w0, const float *__restrict__ b0,
                                                        const float *__restrict__ w1, const float *__restrict__ b1,
                                                        T *__restrict__ comb_in, int embed_blocks, VirtScatter vs) {
    __shared__ float hid[4][16];
    if ((int)blockIdx.x >= embed_blocks) {
        const int HC = vs.H * vs.C;
        const size_t total = (size_t)vs.rows * 4 * HC, stride = (size_t)(gridDim.x - embed_blocks) * blockDim.x;
        for (size_t idx = (size_t)(blockIdx.x - embed_blocks) * blockDim.x + threadIdx.x; idx < total; idx += stride) {
            const size_t r = idx / (4 * (size_t)HC);
            const int col = (int)(idx - r * 4 * HC), which = col / HC, f = col - which * HC;
            const T v = ((const T *)vs.src)[(size_t)(r % vs.V) * 4 * HC + col];
            const size_t node = (size_t)vs.n_real + r;
            if (which == 3) { ((T *)vs.S)[node * HC + f] = v; continue; }
            const int h = f / vs.C, c = f - h * vs.C;
            T *dstb = (T *)(which == 0 ? vs.Q : (which == 1 ? vs.K : vs.Vt));
            dstb[((size_t)h * (size_t)vs.n_pad + (size_t)vs.row_map[node]) * vs.C + c] = v;
        }
        return;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wv) * NPW;
    if (r0 >= n) return;
    float w0r[8], w1r[16];
    float b0r = 0.f, b1r = 0.f;
    if (lane < 16) {
        b0r = b0[lane];
#pragma unroll
        for (int k = 0; k < 8; ++k) w0r[k] = k < c_in ? w0[lane * c_in + k] : 0.f;
    }
    if (lane < 32) {
        b1r = b1[lane];
#pragma unroll
        for (int k = 0; k < 16; ++k) w1r[k] = w1[lane * 16 + k];
    }
    for (int i = 0; i < NPW; ++i) {
        const int r = r0 + i;
        if (r >= n) break;
        if (lane < 16) {
            float a = b0r;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += k < c_in ? w0r[k] * x[(size_t)r * c_in + k] : 0.f;
            hid[wv][lane] = gelu_erf(a);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the row is this wave's own: no workgroup barrier
        T *dst = comb_in + (size_t)r * D + F;
        if (lane < 32) {
            float a = b1r;
#pragma unroll
            for (int k = 0; k < 16; ++k) a += w1r[k] * hid[wv][k];
            stf(dst + lane, a);
        } else {
            int64_t ti = t ? t[r] : t_scalar;
            ti = ti < 0 ? 0 : (ti >= steps ? steps - 1 : ti);
            stf(dst + lane, time_emb[ti * 32 + (lane - 32)]);
        }
        __builtin_amdgcn_wave_barrier();                             // hid[wv] is rewritten by the next node
    }
}

// ------------------------------------------------------------------------------------------
// Reference-grade linear: out = act(A @ W^T + bias) + residual.  64x64 tile, BK = 16,
// fp32 accumulate on the VALU.  Used for odd shapes and as the parity cross-check of the
// MFMA kernel.
template <typename T>
__global__ __launch_bounds__(256) void k_gemm_simple(int M, int K, int Nout, const T *__restrict__ A, int lda,
                                                     const T *__restrict__ W, const float *__restrict__ bias,
                                                     int act, const T *__restrict__ res, T *__restrict__ out,
                                                     int ldo) {
    __shared__ float As[16][68];
    __shared__ float Ws[16][68];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int ar = row0 + lr, wr = col0 + lr;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + lk + i;
            As[lk + i][lr] = (ar < M && k < K) ? ldf(A + (size_t)ar * lda + k) : 0.f;
            Ws[lk + i][lr] = (wr < Nout && k < K) ? ldf(W + (size_t)wr * K + k) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Ws[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + ty * 4 + i;
        if (r >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = col0 + tx * 4 + j;
            if (c >= Nout) continue;
            float v = acc[i][j] + (bias ? bias[c] : 0.f);
            v = apply_act(v, act);
            if (res) v += ldf(res + (size_t)r * ldo + c);
            stf(out + (size_t)r * ldo + c, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// 2D head tail: out[r, :] = W2 @ hh[r, :32] + b2  (final_mlp.2, efficient_gat.py:88-92,145)
template <typename T>
__global__ __launch_bounds__(256) void k_head2d(int n, int c_out, const T *__restrict__ hh,
                                                const float *__restrict__ w2, const float *__restrict__ b2,
                                                float *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * c_out) return;
    const int r = idx / c_out, c = idx - r * c_out;
    float a = b2[c];
#pragma unroll
    for (int k = 0; k < 32; ++k) a += w2[c * 32 + k] * ldf(hh + (size_t)r * 32 + k);
    out[idx] = a;
}

// ------------------------------------------------------------------------------------------
// DDIM update, p_sample_ddim spatial_diffusion.py:555-566,603-627 (fp32, op order kept).
__global__ __launch_bounds__(256) void k_ddim2d(DeviceSchedule s, int mean_type, int n, int c,
                                                const float *__restrict__ x, const float *__restrict__ mo,
                                                const int64_t *__restrict__ t, int64_t t_scalar, int ratio,
                                                int prev_all_nonneg, float eta, const float *__restrict__ noise,
                                                float *__restrict__ x_prev) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * c) return;
    const int r = idx / c;
    const int64_t ti = t ? t[r] : t_scalar;
    x_prev[idx] = ddim2d_value(s, mean_type, ti, ratio, prev_all_nonneg, eta, x[idx], mo[idx], (eta > 0.f && noise) ? noise[idx] : 0.f);
}

// DDPM update, p_sample_ddpm spatial_diffusion.py:485-510.
__global__ __launch_bounds__(256) void k_ddpm2d(DeviceSchedule s, int n, int c, const float *__restrict__ x,
                                                const float *__restrict__ mo, const int64_t *__restrict__ t,
                                                int64_t t_scalar, const float *__restrict__ noise,
                                                float *__restrict__ x_prev) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * c) return;
    const int r = idx / c;
    int64_t ti = t ? t[r] : t_scalar;
    ti = ti < 0 ? 0 : (ti >= s.steps ? s.steps - 1 : ti);
    float mean = s.sqrt_recip_alphas[ti] * (x[idx] - s.betas[ti] * mo[idx] / s.sqrt_one_minus_alphas_cumprod[ti]);
    if (noise) mean += sqrtf(s.posterior_variance[ti]) * noise[idx];
    x_prev[idx] = mean;
}

// fp32 -> act dtype conversion (weight packing)
template <typename T>
__global__ __launch_bounds__(256) void k_convert(size_t n, const float *__restrict__ src, T *__restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        stf(dst + i, src[i]);
}

// ------------------------------------------------------------------------------------------ launchers
static inline int grid_for(size_t total, int block = 256, int cap = 4096) {
    size_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

template <typename T>
static int launch_set_feats_t(int n, int F, int D, const float *feats, void *comb_in, hipStream_t st) {
    if (n <= 0) return 0;
    k_set_feats<T><<<grid_for((size_t)n * F), 256, 0, st>>>(n, F, D, feats, (T *)comb_in);
    DA_LAUNCH_CHECK();
    return 0;
}
int launch_set_feats(int prec, int n, int F, int D, const float *feats, void *comb_in, hipStream_t st) {
    return prec == DA_PREC_BF16 ? launch_set_feats_t<bf16_t>(n, F, D, feats, comb_in, st)
                                : launch_set_feats_t<float>(n, F, D, feats, comb_in, st);
}

int launch_set_virtual_rows(int prec, int rows, int V, int D, const void *emb, void *dst, hipStream_t st) {
    if (rows <= 0) return 0;
    if (prec == DA_PREC_BF16)
        k_set_virtual_rows<bf16_t><<<grid_for((size_t)rows * D), 256, 0, st>>>(rows, V, D, (const bf16_t *)emb, (bf16_t *)dst);
    else
        k_set_virtual_rows<float><<<grid_for((size_t)rows * D), 256, 0, st>>>(rows, V, D, (const float *)emb, (float *)dst);
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_embed_pos_time(int prec, int n, int c_in, int F, int D, const float *x, const int64_t *t, int64_t t_scalar,
                          int steps, const float *time_emb, const float *w0, const float *b0, const float *w1,
                          const float *b1, void *comb_in, hipStream_t st, const VirtScatter *vsp) {
    if (n <= 0) return 0;
    if (c_in > 8) { set_error("launch_embed_pos_time: c_in %d > 8", c_in); return 2; }
    VirtScatter vs;
    if (vsp && vsp->rows > 0) vs = *vsp;
    const size_t vtotal = (size_t)vs.rows * 4 * vs.H * vs.C;
    const int vblocks = vs.rows > 0 ? (int)((vtotal + 255) / 256 > 1024 ? 1024 : (vtotal + 255) / 256) : 0;
    const int npw_env = DA_XENV("DA_EMBED_NPW", 0);
    // nodes per wave: one while the batch is too small to cover the chip otherwise, eight from 8 192 nodes up
    const int npw = npw_env > 0 ? npw_env : (n >= 8192 ? 8 : 1), grid = (n + 4 * npw - 1) / (4 * npw);
    if (prec == DA_PREC_BF16)
        k_embed_pos_time<bf16_t><<<grid + vblocks, 256, 0, st>>>(n, npw, c_in, F, D, x, t, t_scalar, steps, time_emb, w0, b0, w1, b1, (bf16_t *)comb_in, grid, vs);
    else
        k_embed_pos_time<float><<<grid + vblocks, 256, 0, st>>>(n, npw, c_in, F, D, x, t, t_scalar, steps, time_emb, w0, b0, w1, b1, (float *)comb_in, grid, vs);
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_simple(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias,
                       int act, const void *res, void *out, int ldo, hipStream_t st) {
    if (M <= 0 || Nout <= 0) return 0;
    dim3 grid((Nout + 63) / 64, (M + 63) / 64);
    if (prec == DA_PREC_BF16)
        k_gemm_simple<bf16_t><<<grid, 256, 0, st>>>(M, K, Nout, (const bf16_t *)A, lda, (const bf16_t *)W, bias, act, (const bf16_t *)res, (bf16_t *)out, ldo);
    else
        k_gemm_simple<float><<<grid, 256, 0, st>>>(M, K, Nout, (const float *)A, lda, (const float *)W, bias, act, (const float *)res, (float *)out, ldo);
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_head2d(int prec, int n, int c_out, const void *hh, const float *w2, const float *b2, float *out, hipStream_t st) {
    if (n <= 0) return 0;
    const int grid = (n * c_out + 255) / 256;
    if (prec == DA_PREC_BF16)
        k_head2d<bf16_t><<<grid, 256, 0, st>>>(n, c_out, (const bf16_t *)hh, w2, b2, out);
    else
        k_head2d<float><<<grid, 256, 0, st>>>(n, c_out, (const float *)hh, w2, b2, out);
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_ddim2d(const DeviceSchedule &s, int mean_type, int n, int c, const float *x, const float *mo,
                  const int64_t *t, int64_t t_scalar, int ratio, int prev_all_nonneg, float eta, const float *noise,
                  float *x_prev, hipStream_t st) {
    if (n <= 0) return 0;
    k_ddim2d<<<(n * c + 255) / 256, 256, 0, st>>>(s, mean_type, n, c, x, mo, t, t_scalar, ratio, prev_all_nonneg, eta, noise, x_prev);
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_ddpm2d(const DeviceSchedule &s, int n, int c, const float *x, const float *mo, const int64_t *t,
                  int64_t t_scalar, const float *noise, float *x_prev, hipStream_t st) {
    if (n <= 0) return 0;
    k_ddpm2d<<<(n * c + 255) / 256, 256, 0, st>>>(s, n, c, x, mo, t, t_scalar, noise, x_prev);
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_convert(int prec, size_t n, const float *src, void *dst, hipStream_t st) {
    if (n == 0) return 0;
    if (prec == DA_PREC_BF16)
        k_convert<bf16_t><<<grid_for(n), 256, 0, st>>>(n, src, (bf16_t *)dst);
    else
        k_convert<float><<<grid_for(n), 256, 0, st>>>(n, src, (float *)dst);
    DA_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------
// C[m][n] = sum_k A[m * lda + k] * B[k * ldb + n] (+ bias[m]) -- naive fp32, only used at create time to
// compose small weight matrices (a few MFLOP).
__global__ __launch_bounds__(256) void k_mm_nn_f32(int M, int N, int K, const float *__restrict__ A, int lda,
                                                   const float *__restrict__ B, int ldb, const float *__restrict__ bias,
                                                   float *__restrict__ Cm, int ldc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int m = idx / N, n = idx - m * N;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * lda + k], B[(size_t)k * ldb + n], acc);
    Cm[(size_t)m * ldc + n] = acc + (bias ? bias[m] : 0.f);
}

int launch_mm_nn_f32(int M, int N, int K, const float *A, int lda, const float *B, int ldb, const float *bias, float *Cm,
                     int ldc, hipStream_t st) {
    k_mm_nn_f32<<<(M * N + 255) / 256, 256, 0, st>>>(M, N, K, A, lda, B, ldb, bias, Cm, ldc);
    DA_LAUNCH_CHECK();
    return 0;
}

// Tail of the folded last layer (2D): f = GELU(sum_h Pz[h][n][:] + pre[n][:]) ; out[n] = W2 f + b2
// (final_mlp: efficient_gat.py:144-146 with the value heads and the residual already projected to 32 wide).
// 32 rows per workgroup, 8 lanes per row with four channels each: every load is 16 bytes (the per-head partial outputs are
// what this kernel exists to read -- 59 MB at 64 puzzles while they were fp32; they are kept in the activation dtype now,
// like the attention output they replace -- and one float per thread left it latency-bound at 2.2 TB/s).  The sums run
// in the same order as before (pre, then heads 0 .. H-1; k ascending), so the outputs are bit-identical.
__device__ __forceinline__ void ld4f(const float *p, float (&v)[4]) {
    const float4 f = *(const float4 *)p;
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}
__device__ __forceinline__ void ld4f(const bf16_t *p, float (&v)[4]) {
    const uint2 u = *(const uint2 *)p;
    v[0] = bf2f((bf16_t)(u.x & 0xffff)); v[1] = bf2f((bf16_t)(u.x >> 16));
    v[2] = bf2f((bf16_t)(u.y & 0xffff)); v[3] = bf2f((bf16_t)(u.y >> 16));
}
template <typename T>
__global__ __launch_bounds__(256) void k_head_fold(int n, int H, int c_out, const T *__restrict__ pz, const T *__restrict__ pre,
                                                   const float *__restrict__ w2, const float *__restrict__ b2,
                                                   float *__restrict__ out, DdimFuse df) {
    __shared__ float f[32][33];
    const int q = threadIdx.x & 7, lr = threadIdx.x >> 3;
    const int r = blockIdx.x * 32 + lr;
    if (r < n) {
        float a[4], v[4];
        ld4f(pre + (size_t)r * 32 + 4 * q, a);
        for (int h = 0; h < H; ++h) {
            ld4f(pz + ((size_t)h * n + r) * 32 + 4 * q, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) f[lr][4 * q + e] = gelu_erf(a[e]);
    }
    __syncthreads();
    // second linear: thread -> (row, output channel), c_out <= 8
    const int lr2 = threadIdx.x / c_out, j = threadIdx.x - lr2 * c_out;
    const int r2 = blockIdx.x * 32 + lr2;
    if (lr2 < 32 && r2 < n) {
        float a = b2[j];
#pragma unroll
        for (int k = 0; k < 32; ++k) a = fmaf(w2[j * 32 + k], f[lr2][k], a);
        out[(size_t)r2 * c_out + j] = a;
        // sampling loop: the DDIM update of this element right here (x_prev never aliases x: the loop ping-pongs)
        if (df.x_prev)
            df.x_prev[(size_t)r2 * c_out + j] = ddim2d_value(df.s, df.mean_type, df.t, df.ratio, df.prev_all_nonneg, 0.f,
                                                             df.x[(size_t)r2 * c_out + j], a, 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// The whole folded tail of the 2D transformer arch as ONE kernel (bf16 mode, hidden 128, conv input 256, 8 heads):
//   pre[n]  = Wh . h[n] + bh  +  Ws . xin[n] + bs          (mlp.2 share and skip share of final_mlp.0, 32 wide)
//   f[n]    = GELU(pre[n] + sum_h Pz[h][n])                (Pz: per-head outputs of the folded last attention)
//   out[n]  = W2 f[n] + b2   (+ the DDIM update of the sampling loop)
// It used to be two skinny GEMMs (N = 32: 15.4 + 9.9 us at 64 puzzles, 1.9 TB/s) + k_head_fold (14 us); the tail only
// has to read h (14.7 MB), xin (29.5 MB) and Pz (14.7 MB).  One wave per 32 rows, transposed product D^T = W . X^T on
// v_mfma_f32_32x32x16_bf16: A = the 32 x 384 weight block, held in 96 registers for the wave's lifetime (K chunk c: lane
// (m, g) holds W[m][16 c + 8 g .. + 7]); B = the rows themselves, one 16-byte global load per lane and chunk (lane (n, g):
// row n, the same 8 columns) -- all 24 loads of a slab are in flight before the first MFMA.  The accumulator leaves
// lane (n, g) with channels (r & 3) + 8 (r >> 2) + 4 g, r = 0..15, of row n: the GELU is lane-local, the 32 -> c_out
// product is two half sums joined by one cross-half exchange.
typedef __attribute__((ext_vector_type(8))) __bf16 tl_bf16x8;
typedef __attribute__((ext_vector_type(16))) float tl_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int tl_u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <bool NEXT>
__global__ __launch_bounds__(64) void k_tail_fused(int n, int H, int c_out, const bf16_t *h, const bf16_t *__restrict__ xin,
                                                   int ldx, const bf16_t *__restrict__ wh, const float *__restrict__ bh,
                                                   const bf16_t *__restrict__ wsk, const float *__restrict__ bsk,
                                                   const bf16_t *__restrict__ pz, const float *__restrict__ w2,
                                                   const float *__restrict__ b2, float *__restrict__ out, DdimFuse df, int slabs_per_wave) {
    constexpr int KH = 128, KX = 256, NCH = (KH + KX) / 16;
    const int lane = threadIdx.x, m = lane & 31, g = lane >> 5;
    tl_u32x4 wf[NCH];
#pragma unroll
    for (int c = 0; c < KH / 16; ++c) wf[c] = *(const tl_u32x4 *)(wh + (size_t)m * KH + 16 * c + 8 * g);
#pragma unroll
    for (int c = 0; c < KX / 16; ++c) wf[KH / 16 + c] = *(const tl_u32x4 *)(wsk + (size_t)m * KX + 16 * c + 8 * g);
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int ch = (r & 3) + 8 * (r >> 2) + 4 * g; bias[r] = bh[ch] + bsk[ch]; }
    for (int sl = 0; sl < slabs_per_wave; ++sl) {
        const int row0 = (blockIdx.x * slabs_per_wave + sl) * 32;
        if (row0 >= n) break;
        const int row = min(row0 + m, n - 1);                        // rows beyond n: computed, not stored
        tl_u32x4 xf[NCH];
#pragma unroll
        for (int c = 0; c < KH / 16; ++c) xf[c] = *(const tl_u32x4 *)(h + (size_t)row * KH + 16 * c + 8 * g);
#pragma unroll
        for (int c = 0; c < KX / 16; ++c) xf[KH / 16 + c] = *(const tl_u32x4 *)(xin + (size_t)row * ldx + 16 * c + 8 * g);
        // per-head partial outputs of this lane's 16 channels: 4 pieces of 4 consecutive channels per head
        uint2 pzv[8][4];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pzv[hh][q] = hh < H ? *(const uint2 *)(pz + ((size_t)hh * n + row) * 32 + 8 * q + 4 * g) : make_uint2(0u, 0u);
        tl_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias[r];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tl_bf16x8, wf[c]), __builtin_bit_cast(tl_bf16x8, xf[c]), acc, 0, 0, 0);
        float f[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
#pragma unroll
            for (int hh = 0; hh < 8; ++hh) {
                const uint2 u = pzv[hh][q];
                a[0] += bf2f((bf16_t)(u.x & 0xffff)); a[1] += bf2f((bf16_t)(u.x >> 16));
                a[2] += bf2f((bf16_t)(u.y & 0xffff)); a[3] += bf2f((bf16_t)(u.y >> 16));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * q + e] = gelu_erf(a[e]);
        }
        // second linear: this lane's 16 channels, then the other half's
        float xn[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};            // NEXT: this row's updated sample (both halves hold it)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j >= c_out) break;
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k0 = (r & 3) + 8 * (r >> 2);
                const float wlo = w2[j * 32 + k0], whi = w2[j * 32 + k0 + 4];        // uniform: scalar loads
                part = fmaf(g ? whi : wlo, f[r], part);
            }
            const float a = b2[j] + part + __shfl_xor(part, 32);
            const size_t o = (size_t)row * c_out + j;
            if (df.x_prev) xn[j] = ddim2d_value(df.s, df.mean_type, df.t, df.ratio, df.prev_all_nonneg, 0.f, df.x[o], a, 0.f);
            if (g == 0 && row0 + m < n) {
                out[o] = a;
                if (df.x_prev) df.x_prev[o] = xn[j];
            }
        }
        if (NEXT) {
            // ---- the next step's h rows (efficient_gat.py:131-135 with the feature part of mlp.0 hoisted): pose MLP in fp32 exactly as
            // k_embed_pos_time does it (hidden = GELU(W0 x + b0), pos = W1 hidden + b1), pos | time_emb[t'] rounded to bf16 (the concat
            // buffer's dtype), then h = GELU(Wp [pos | time] + feat_proj) on the matrix cores, transposed as above.
            const int cin = df.nx_cin;
            float hm[8];                    // hidden units 8 g .. 8 g + 7 of this row; the other half computes the other eight
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float a = g ? df.nx_b0[8 + kk] : df.nx_b0[kk];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < cin) a += (g ? df.nx_w0[(8 + kk) * cin + j] : df.nx_w0[kk * cin + j]) * xn[j];
                hm[kk] = gelu_erf(a);
            }
            float hid[16];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float ho = __shfl_xor(hm[kk], 32);
                hid[kk] = g ? ho : hm[kk];
                hid[8 + kk] = g ? hm[kk] : ho;
            }
            long long ti = df.nx_t;
            ti = ti < 0 ? 0 : (ti >= df.nx_steps ? df.nx_steps - 1 : ti);
            tl_u32x4 bx[4];                 // B operand: chunk c = columns 16 c + 8 g .. + 7 of [pos(32) | time(32)] of this row
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned pk[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    unsigned short two[2];
#pragma unroll
                    for (int z = 0; z < 2; ++z) {
                        const int o = 16 * c + 8 * g + 2 * e2 + z;
                        const f32x4_t *wr = (const f32x4_t *)(df.nx_w1 + (size_t)o * 16);
                        float a = df.nx_b1[o];
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const f32x4_t w4 = wr[q4];
                            a += w4[0] * hid[4 * q4]; a += w4[1] * hid[4 * q4 + 1]; a += w4[2] * hid[4 * q4 + 2]; a += w4[3] * hid[4 * q4 + 3];
                        }
                        two[z] = f2bf(a);
                    }
                    pk[e2] = (unsigned)two[0] | ((unsigned)two[1] << 16);
                }
                bx[c] = (tl_u32x4){pk[0], pk[1], pk[2], pk[3]};
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4_t *tr = (const f32x4_t *)(df.nx_time_emb + (size_t)ti * 32 + 16 * c + 8 * g);
                const f32x4_t t0 = tr[0], t1 = tr[1];
                bx[2 + c] = (tl_u32x4){(unsigned)f2bf(t0[0]) | ((unsigned)f2bf(t0[1]) << 16), (unsigned)f2bf(t0[2]) | ((unsigned)f2bf(t0[3]) << 16),
                                       (unsigned)f2bf(t1[0]) | ((unsigned)f2bf(t1[1]) << 16), (unsigned)f2bf(t1[2]) | ((unsigned)f2bf(t1[3]) << 16)};
            }
            const bf16_t *wp = (const bf16_t *)df.nx_wp;
            const bf16_t *fp = (const bf16_t *)df.nx_feat_proj + (size_t)row * 128;
            bf16_t *hn = (bf16_t *)df.nx_h + (size_t)row * 128;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                tl_u32x4 wa[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) wa[c] = *(const tl_u32x4 *)(wp + (size_t)(32 * mt + m) * df.nx_ldw + 16 * c + 8 * g);
                uint2 fpv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) fpv[q] = *(const uint2 *)(fp + 32 * mt + 8 * q + 4 * g);
                tl_f32x16 ac;
#pragma unroll
                for (int r = 0; r < 16; ++r) ac[r] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tl_bf16x8, wa[c]), __builtin_bit_cast(tl_bf16x8, bx[c]), ac, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4] = {ac[4 * q] + bf2f((bf16_t)(fpv[q].x & 0xffff)), ac[4 * q + 1] + bf2f((bf16_t)(fpv[q].x >> 16)),
                                  ac[4 * q + 2] + bf2f((bf16_t)(fpv[q].y & 0xffff)), ac[4 * q + 3] + bf2f((bf16_t)(fpv[q].y >> 16))};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    if (row0 + m < n)
                        *(uint2 *)(hn + 32 * mt + 8 * q + 4 * g) = make_uint2((unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16),
                                                                              (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16));
                }
            }
        }
    }
}

// returns 0 = launched, -1 = shape / precision not covered (the caller runs the three-kernel tail)
int launch_tail_fused(int prec, int n, int H, int c_out, int hidden, int din, const void *h, const void *xin, int ldx, const void *wh,
                      const float *bh, const void *wsk, const float *bsk, const void *pz, const float *w2, const float *b2, float *out,
                      hipStream_t st, DdimFuse *dfp) {
    const bool off = (cfg().disable_folds & DA_FOLD_TAIL) != 0;
    if (off || prec != DA_PREC_BF16 || hidden != 128 || din != 256 || H > 8 || c_out > 8 || (ldx & 7)) return -1;
    if (n <= 0) return 0;
    DdimFuse df;
    if (dfp) df = *dfp; else { df = DdimFuse(); df.x = nullptr; df.x_prev = nullptr; }
    // slabs per wave: as many waves as the chip has SIMDs (1 024, one resident wave each at 256 VGPRs) run in ONE round and
    // load the weight block once (57 600 rows: 2 slabs per wave 24.6 - 25.2 us, 1 slab 27.3 - 27.6, 4 slabs 26.9)
    const int slabs = (n + 31) / 32;
    int spw = (slabs + 1023) / 1024;
    spw = spw < 1 ? 1 : (spw > 4 ? 4 : spw);
    // the next step's embedding + mlp.0 in the same kernel (DdimFuse::nx_*): requested by enqueue_loop (da_api.hip) -- since the round's very last
    // session by default for Batches whose largest graph has >= 512 pieces (DA_STEP_AUTO; together with the row-panel projections -2.5 ... -3.3 % on
    // 100-iteration loops of the headline Batch, profiles/r05/r05_step_auto_ab.log), DA_TAIL_NEXT=1 / 0 forces it on / off.  First measured (A/B on
    // one box): parity-clean (tests/test_gpu_tail_next.py) and two launches fewer per step, but not faster -- headline 0.6786 / 0.6890 ms
    // without, 0.6788 / 0.6845 with; one-branch loop 0.7175 -> 0.7154; configuration 2 (512 x 144 pieces) 0.6346 -> 0.6428: the work moves into a
    // kernel that runs one wave per SIMD at 255 VGPRs (64 more GELUs and 16 MFMAs per lane and slab), which costs what the launches cost.
    const bool nx_off = cfg().tail_next == 0;        // the request itself (nx_on) is decided in enqueue_loop
    const bool next = dfp && df.x_prev && df.nx_on && !nx_off && df.nx_cin <= 8 && df.nx_cin == c_out && df.nx_h == h;
    if (next)
        k_tail_fused<true><<<(slabs + spw - 1) / spw, 64, 0, st>>>(n, H, c_out, (const bf16_t *)h, (const bf16_t *)xin, ldx, (const bf16_t *)wh, bh,
                                                                   (const bf16_t *)wsk, bsk, (const bf16_t *)pz, w2, b2, out, df, spw);
    else
        k_tail_fused<false><<<(slabs + spw - 1) / spw, 64, 0, st>>>(n, H, c_out, (const bf16_t *)h, (const bf16_t *)xin, ldx, (const bf16_t *)wh, bh,
                                                                    (const bf16_t *)wsk, bsk, (const bf16_t *)pz, w2, b2, out, df, spw);
    DA_LAUNCH_CHECK();
    if (next) dfp->nx_done = 1;
    return 0;
}

int launch_head_fold(int prec, int n, int H, int c_out, const void *pz, const void *pre, const float *w2, const float *b2,
                     float *out, hipStream_t st, const DdimFuse *dfp) {
    if (n <= 0) return 0;
    DdimFuse df;
    if (dfp) df = *dfp; else { df = DdimFuse(); df.x = nullptr; df.x_prev = nullptr; }
    if (c_out > 8) { set_error("launch_head_fold: c_out %d > 8", c_out); return 2; }
    if (prec == DA_PREC_BF16) k_head_fold<bf16_t><<<(n + 31) / 32, 256, 0, st>>>(n, H, c_out, (const bf16_t *)pz, (const bf16_t *)pre, w2, b2, out, df);
    else k_head_fold<float><<<(n + 31) / 32, 256, 0, st>>>(n, H, c_out, (const float *)pz, (const float *)pre, w2, b2, out, df);
    DA_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------
// exophormer + folded mlp.2: the conv-0 projections of the virtual rows are constants of the checkpoint
// (virt_emb . Wcat0^T + b, [V, 4*HC]); every step they are placed behind the real rows' projections.
//   dense / hybrid layouts: Q, K, V head-major at the row's padded slot, skip row-major
//   CSR layout: row-major qkvs[n][4*HC]
template <typename T>
__global__ __launch_bounds__(256) void k_scatter_virtual(int rows, int V, int H, int C, const T *__restrict__ src, int n_real,
                                                         const int32_t *__restrict__ row_map, size_t n_pad, T *Q, T *K, T *Vt,
                                                         T *S, T *qkvs) {
    const int HC = H * C;
    const size_t total = (size_t)rows * 4 * HC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / (4 * (size_t)HC);
        const int col = (int)(idx - r * 4 * HC), which = col / HC, f = col - which * HC;
        const T v = src[(size_t)(r % V) * 4 * HC + col];
        const size_t node = (size_t)n_real + r;
        if (qkvs) { qkvs[node * 4 * HC + col] = v; continue; }
        if (which == 3) { S[node * HC + f] = v; continue; }
        const int h = f / C, c = f - h * C;
        T *dstb = which == 0 ? Q : (which == 1 ? K : Vt);
        dstb[((size_t)h * n_pad + (size_t)row_map[node]) * C + c] = v;
    }
}

int launch_scatter_virtual(int prec, int rows, int V, int H, int C, const void *src, int n_real, const int32_t *row_map,
                           int n_pad, void *Q, void *K, void *Vt, void *S, void *qkvs, hipStream_t st) {
    if (rows <= 0) return 0;
    const size_t total = (size_t)rows * 4 * H * C;
    const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (prec == DA_PREC_BF16)
        k_scatter_virtual<bf16_t><<<grid, 256, 0, st>>>(rows, V, H, C, (const bf16_t *)src, n_real, row_map, (size_t)n_pad,
                                                        (bf16_t *)Q, (bf16_t *)K, (bf16_t *)Vt, (bf16_t *)S, (bf16_t *)qkvs);
    else
        k_scatter_virtual<float><<<grid, 256, 0, st>>>(rows, V, H, C, (const float *)src, n_real, row_map, (size_t)n_pad,
                                                       (float *)Q, (float *)K, (float *)Vt, (float *)S, (float *)qkvs);
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
