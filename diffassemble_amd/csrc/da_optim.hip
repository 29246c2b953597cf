// Fused Adafactor step over the flat parameter / gradient buffers of the training engine.
//
// Replaces, for the live denoiser parameters, the optimizer the reference configures
// (spatial_diffusion.py:701-705: transformers.optimization.Adafactor(self.parameters()) with its
// defaults: relative step, scale_parameter, eps = (1e-30, 1e-3), clip_threshold 1, decay_rate -0.8, no
// first moment, no weight decay).  The torch implementation issues ~20 tiny kernels and two host syncs
// per parameter (~700 launches, 6 ms per step); here the whole step is FOUR launches driven by device
// tables, every reduction is two-stage and deterministic (all data-parallel ranks must apply
// bit-identical updates to stay in sync), nothing returns to the host.
//
// Per parameter p with gradient g (fp32), step t:
//   rho = min(1e-2, 1/sqrt(t)),  lr = max(eps2, rms(p)) * rho,  beta = 1 - t^decay
//   matrices [R, C] (factored second moment):
//     row_r <- beta row_r + (1-beta) mean_c(g^2 + eps1),  col_c <- beta col_c + (1-beta) mean_r(g^2 + eps1)
//     u = g * rsqrt(row_r / mean(row)) * rsqrt(col_c)
//   vectors:  v <- beta v + (1-beta)(g^2 + eps1),  u = g * rsqrt(v)
//   p <- p - lr * u / max(1, rms(u) / clip)
#include <math.h>

#include "da_common.h"

namespace da {

struct AfParam {            // one entry per parameter tensor (device table)
    long long off;          // offset (floats) into flat / flat_grad
    int rows, cols;         // matrices: [rows, cols]; vectors: rows = 1, cols = numel
    int factored;           // 1 = matrix
    long long row_off, col_off;   // offsets into the state buffer (matrices: row[R], col[C]; vectors: v[numel] at row_off)
    int blk0, nblk;         // its blocks in the block table
    long long colpart_off;  // scratch: [nblk][cols] column partial sums (matrices)
};
struct AfBlock { int pid, row0, nrows; };     // matrices: a chunk of rows; vectors: a chunk of `nrows` ELEMENTS from row0
constexpr int AF_RMAX = 512;                  // rows of a block whose factors phases C / D keep in LDS
constexpr int AF_CPL = 20;                    // columns per LANE in phase A: matrices up to 1280 columns

__device__ __forceinline__ float block_sum(float v, float *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    return s;
}

// Phase A: per block -- sum p^2, row EMA (matrices: final), column partial sums; vectors: v EMA + sum u^2
__global__ __launch_bounds__(256) void k_af_a(const AfParam *__restrict__ P, const AfBlock *__restrict__ B, const float *__restrict__ flat,
                                              const float *__restrict__ grad, float *__restrict__ state, float *__restrict__ colpart,
                                              float *__restrict__ part_p2, float *__restrict__ part_u2, float beta, float eps1) {
    __shared__ float red[4];
    const AfBlock b = B[blockIdx.x];
    const AfParam p = P[b.pid];
    const float *w = flat + p.off, *g = grad + p.off;
    float sp = 0.f, su = 0.f;
    if (p.factored) {
        // wave w takes rows row0 + w, row0 + w + 4, ...; lane l the columns l, l + 64, ... : row sums are
        // wave reductions (no block barrier per row), column sums stay in registers until the end
        __shared__ float colred[4][64 * AF_CPL];
        float *row = state + p.row_off;
        float *cp = colpart + p.colpart_off + (size_t)(blockIdx.x - p.blk0) * p.cols;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        float ca[AF_CPL];
#pragma unroll
        for (int k = 0; k < AF_CPL; ++k) ca[k] = 0.f;
        // (round 5: the columns this lane really has -- ncl <= AF_CPL -- and TWO rows of the wave in flight: a row was a chain of
        //  load -> reduce -> read-modify-write of its EMA, eight rows per wave one after the other = the kernel's 29 us)
        const int ncl = (p.cols - lane + 63) / 64;
        const int rend = b.row0 + b.nrows;
        for (int r = b.row0 + wv; r < rend; r += 8) {
            const bool two = r + 4 < rend;
            const float *g0 = g + (size_t)r * p.cols + lane, *w0 = w + (size_t)r * p.cols + lane;
            const size_t d1 = two ? (size_t)4 * p.cols : 0;
            const float re0 = lane == 0 ? row[r] : 0.f, re1 = (lane == 0 && two) ? row[r + 4] : 0.f;
            float sr0 = 0.f, sr1 = 0.f;
#pragma unroll
            for (int k = 0; k < AF_CPL; ++k) {
                if (k < ncl) {
                    const float ga = g0[64 * k], wa = w0[64 * k], gb = g0[d1 + 64 * k], wb = w0[d1 + 64 * k];
                    const float qa = fmaf(ga, ga, eps1), qb = fmaf(gb, gb, eps1);
                    sr0 += qa;
                    ca[k] += qa;
                    sp = fmaf(wa, wa, sp);
                    if (two) { sr1 += qb; ca[k] += qb; sp = fmaf(wb, wb, sp); }
                }
            }
            for (int o = 32; o > 0; o >>= 1) { sr0 += __shfl_xor(sr0, o); sr1 += __shfl_xor(sr1, o); }
            if (lane == 0) {
                row[r] = beta * re0 + (1.0f - beta) * (sr0 / (float)p.cols);
                if (two) row[r + 4] = beta * re1 + (1.0f - beta) * (sr1 / (float)p.cols);
            }
        }
#pragma unroll
        for (int k = 0; k < AF_CPL; ++k) colred[wv][lane + 64 * k] = ca[k];
        __syncthreads();
        for (int c = threadIdx.x; c < p.cols; c += 256) cp[c] = (colred[0][c] + colred[1][c]) + (colred[2][c] + colred[3][c]);
    } else {
        float *v = state + p.row_off;
        for (int i = b.row0 + threadIdx.x; i < b.row0 + b.nrows; i += 256) {
            const float gv = g[i], wv = w[i];
            const float vv = beta * v[i] + (1.0f - beta) * fmaf(gv, gv, eps1);
            v[i] = vv;
            const float u = gv * rsqrtf(vv);
            su = fmaf(u, u, su);
            sp = fmaf(wv, wv, sp);
        }
    }
    sp = block_sum(sp, red);
    su = block_sum(su, red);
    if (threadIdx.x == 0) { part_p2[blockIdx.x] = sp; part_u2[blockIdx.x] = su; }
}

// Phase B: grid (parameter, 64-column chunk) -- chunk 0 of a parameter: rms(p) -> lr and the mean of the row EMA; every chunk: the
// column EMA of its 64 columns, four threads per column (thread (tx, ty) adds the row blocks' partials ty, ty + 4, ... in ascending
// order, the four sub-sums are combined as (s0 + s1) + (s2 + s3): a fixed order -- data-parallel replicas must apply identical
// bits).  Round 5: one block per parameter walked up to 144 partials per column in 18 dependent rounds of eight loads and set the
// launch's duration alone (18 us of the step's 95 us of Adafactor at BASELINE configuration 5).
constexpr int AF_BCH = 64;
__global__ __launch_bounds__(256) void k_af_b(const AfParam *__restrict__ P, float *__restrict__ state, const float *__restrict__ colpart,
                                              const float *__restrict__ part_p2, float *__restrict__ scal /* [n][4]: lr, row_mean, -, - */,
                                              float beta, float rho, float eps2) {
    __shared__ float red[4];
    __shared__ float csub[4][AF_BCH];
    const AfParam p = P[blockIdx.x];
    const int c0 = blockIdx.y * AF_BCH;
    if (blockIdx.y > 0 && (!p.factored || c0 >= p.cols)) return;
    if (p.factored && c0 < p.cols) {
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, c = c0 + tx;
        float t = 0.f;
        if (c < p.cols) {
            const float *cp = colpart + p.colpart_off + c;
            int k = ty;
            for (; k + 28 < p.nblk; k += 32) {              // eight of this thread's partials in flight
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = cp[(size_t)(k + 4 * u) * p.cols];
#pragma unroll
                for (int u = 0; u < 8; ++u) t += v[u];
            }
            for (; k < p.nblk; k += 4) t += cp[(size_t)k * p.cols];
        }
        csub[ty][tx] = t;
        __syncthreads();
        if (ty == 0 && c < p.cols) {
            float *col = state + p.col_off;
            const float tot = (csub[0][tx] + csub[1][tx]) + (csub[2][tx] + csub[3][tx]);
            col[c] = beta * col[c] + (1.0f - beta) * (tot / (float)p.rows);
        }
    }
    if (blockIdx.y > 0) return;
    float s = 0.f;
    for (int k = threadIdx.x; k < p.nblk; k += 256) s += part_p2[p.blk0 + k];
    s = block_sum(s, red);
    const float numel = (float)p.rows * (float)p.cols;
    const float rms = sqrtf(s) / sqrtf(numel);
    float rm = 0.f;
    if (p.factored) {
        const float *row = state + p.row_off;
        for (int r = threadIdx.x; r < p.rows; r += 256) rm += row[r];
        rm = block_sum(rm, red) / (float)p.rows;
    }
    if (threadIdx.x == 0) {
        scal[blockIdx.x * 4 + 0] = fmaxf(eps2, rms) * rho;
        scal[blockIdx.x * 4 + 1] = rm;
    }
}

// Phase C (matrices): sum u^2 per block
__global__ __launch_bounds__(256) void k_af_c(const AfParam *__restrict__ P, const AfBlock *__restrict__ B, const float *__restrict__ grad,
                                              const float *__restrict__ state, const float *__restrict__ scal, float *__restrict__ part_u2) {
    __shared__ float red[4];
    const AfBlock b = B[blockIdx.x];
    const AfParam p = P[b.pid];
    if (!p.factored) return;                  // vectors were done in phase A (uniform per block)
    const float *g = grad + p.off, *row = state + p.row_off, *col = state + p.col_off;
    const float rm = scal[b.pid * 4 + 1];
    float su = 0.f;
    // (round 5: the block's row factors once into LDS, then a thread walks ITS columns down the rows with eight loads in flight --
    //  the row loop was a chain of "row EMA -> rsqrt -> that row's loads", 32 rows one after the other = the kernel's 23 us)
    __shared__ float rfs[AF_RMAX];
    if (b.nrows <= AF_RMAX) {
        for (int r = threadIdx.x; r < b.nrows; r += 256) rfs[r] = rsqrtf(row[b.row0 + r] / rm);
        __syncthreads();
        for (int c = threadIdx.x; c < p.cols; c += 256) {
            const float cf = rsqrtf(col[c]);
            const float *gc = g + (size_t)b.row0 * p.cols + c;
            int r = 0;
            for (; r + 8 <= b.nrows; r += 8) {
                float gv[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) gv[x] = gc[(size_t)(r + x) * p.cols];
#pragma unroll
                for (int x = 0; x < 8; ++x) { const float u = gv[x] * (rfs[r + x] * cf); su = fmaf(u, u, su); }
            }
            for (; r < b.nrows; ++r) { const float u = gc[(size_t)r * p.cols] * (rfs[r] * cf); su = fmaf(u, u, su); }
        }
    } else {
    for (int r = b.row0; r < b.row0 + b.nrows; ++r) {
        const float rf = rsqrtf(row[r] / rm);
        for (int c = threadIdx.x; c < p.cols; c += 256) {
            const float u = g[(size_t)r * p.cols + c] * (rf * rsqrtf(col[c]));
            su = fmaf(u, u, su);
        }
    }
    }
    su = block_sum(su, red);
    if (threadIdx.x == 0) part_u2[blockIdx.x] = su;
}

// Phase D: clip by rms(u), apply
__global__ __launch_bounds__(256) void k_af_d(const AfParam *__restrict__ P, const AfBlock *__restrict__ B, float *__restrict__ flat,
                                              const float *__restrict__ grad, const float *__restrict__ state, const float *__restrict__ scal,
                                              const float *__restrict__ part_u2, float clip) {
    __shared__ float red[4];
    const AfBlock b = B[blockIdx.x];
    const AfParam p = P[b.pid];
    float s = 0.f;
    for (int k = threadIdx.x; k < p.nblk; k += 256) s += part_u2[p.blk0 + k];
    s = block_sum(s, red);
    const float numel = (float)p.rows * (float)p.cols;
    const float rms_u = sqrtf(s) / sqrtf(numel);
    const float step = scal[b.pid * 4 + 0] / fmaxf(1.0f, rms_u / clip);
    float *w = flat + p.off;
    const float *g = grad + p.off;
    if (p.factored) {
        const float *row = state + p.row_off, *col = state + p.col_off;
        const float rm = scal[b.pid * 4 + 1];
        __shared__ float rfs[AF_RMAX];
        if (b.nrows <= AF_RMAX) {                       // (as phase C: row factors in LDS, eight rows of a column in flight)
            for (int r = threadIdx.x; r < b.nrows; r += 256) rfs[r] = rsqrtf(row[b.row0 + r] / rm);
            __syncthreads();
            for (int c = threadIdx.x; c < p.cols; c += 256) {
                const float cf = rsqrtf(col[c]);
                const size_t i0 = (size_t)b.row0 * p.cols + c;
                int r = 0;
                for (; r + 8 <= b.nrows; r += 8) {
                    float gv[8], wv[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) { gv[x] = g[i0 + (size_t)(r + x) * p.cols]; wv[x] = w[i0 + (size_t)(r + x) * p.cols]; }
#pragma unroll
                    for (int x = 0; x < 8; ++x) w[i0 + (size_t)(r + x) * p.cols] = wv[x] - step * (gv[x] * (rfs[r + x] * cf));
                }
                for (; r < b.nrows; ++r) { const size_t i = i0 + (size_t)r * p.cols; w[i] -= step * (g[i] * (rfs[r] * cf)); }
            }
        } else {
        for (int r = b.row0; r < b.row0 + b.nrows; ++r) {
            const float rf = rsqrtf(row[r] / rm);
            for (int c = threadIdx.x; c < p.cols; c += 256) {
                const size_t i = (size_t)r * p.cols + c;
                w[i] -= step * (g[i] * (rf * rsqrtf(col[c])));
            }
        }
        }
    } else {
        const float *v = state + p.row_off;
        for (int i = b.row0 + threadIdx.x; i < b.row0 + b.nrows; i += 256) w[i] -= step * (g[i] * rsqrtf(v[i]));
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The elementwise glue of p_losses (spatial_diffusion.py:421-430, 432-483): q_sample and the loss, each as ONE launch (the torch forms
// were ~6 + ~5 launches of 4 - 5 us between the optimizer step and the forward: VERDICT r05 item 8).
// q_sample: x_noisy = extract(sqrt_alphas_cumprod, t) * x_start + extract(sqrt_one_minus_alphas_cumprod, t) * noise -- two rounded products,
// one rounded sum, the reference's order: bit-identical to the torch expression.
__global__ __launch_bounds__(256) void k_q_sample(int steps, size_t n, int c, const float *__restrict__ sa, const float *__restrict__ sb,
                                                  const float *__restrict__ x, const float *__restrict__ nz, const int64_t *__restrict__ t,
                                                  float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * (size_t)c) return;
    long long ti = t[i / c];
    ti = ti < 0 ? 0 : (ti >= steps ? steps - 1 : ti);
    out[i] = __fadd_rn(__fmul_rn(sa[ti], x[i]), __fmul_rn(sb[ti], nz[i]));
}
// loss = mean over the n elements of l(target - pred), d_pred = d loss / d pred, ONE workgroup (n = N c is a few 10^4): fixed summation order
// (thread-strided partial sums, then a tree in LDS) -- deterministic, data-parallel replicas agree bit for bit.
//   kind 0: l1 (|d|), 1: l2 (d^2), 2: smooth-l1 / Huber with beta = 1 (F.smooth_l1_loss: 0.5 d^2 below 1, |d| - 0.5 above)
__global__ __launch_bounds__(1024) void k_loss_grad(int kind, size_t n, const float *__restrict__ target, const float *__restrict__ pred,
                                                    float *__restrict__ loss, float *__restrict__ d_pred) {
    __shared__ float red[1024];
    const float inv_n = 1.0f / (float)n;
    float s = 0.f;
    auto one = [&](size_t i, float tv, float pv) {
        const float d = tv - pv, a = fabsf(d);
        float l, g;                                     // g = d l / d (target - pred)
        if (kind == 0) { l = a; g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
        else if (kind == 1) { l = d * d; g = 2.f * d; }
        else if (a < 1.f) { l = 0.5f * d * d; g = d; }
        else { l = a - 0.5f; g = d > 0.f ? 1.f : -1.f; }
        s += l;
        d_pred[i] = -g * inv_n;
    };
    // eight of a thread's elements requested together (36 dependent round trips at n = 36 864 were the kernel's 22 us), summed in the same order
    size_t i = threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
        float tv[8], pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { tv[u] = target[i + (size_t)u * 1024]; pv[u] = pred[i + (size_t)u * 1024]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) one(i + (size_t)u * 1024, tv[u], pv[u]);
    }
    for (; i < n; i += 1024) one(i, target[i], pred[i]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] * inv_n;
}

}  // namespace da

using namespace da;

extern "C" {

int da_q_sample(int steps, int n, int c, const float *sqrt_alphas_cumprod, const float *sqrt_one_minus_alphas_cumprod, const float *x_start,
                const float *noise, const int64_t *t, float *x_noisy, void *stream) {
    DA_REQUIRE(steps > 0 && c > 0 && sqrt_alphas_cumprod && sqrt_one_minus_alphas_cumprod && x_start && noise && t && x_noisy, "da_q_sample: null argument");
    if (n <= 0) return 0;
    const size_t tot = (size_t)n * c;
    k_q_sample<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(steps, (size_t)n, c, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod,
                                                                               x_start, noise, t, x_noisy);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_loss_grad(int kind, size_t n, const float *target, const float *pred, float *loss, float *d_pred, void *stream) {
    DA_REQUIRE(kind >= 0 && kind <= 2 && n > 0 && target && pred && loss && d_pred, "da_loss_grad: bad argument");
    k_loss_grad<<<1, 1024, 0, (hipStream_t)stream>>>(kind, n, target, pred, loss, d_pred);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_adafactor_step(int n_params, const void *param_table, int n_blocks, const void *block_table, float *flat,
                      const float *flat_grad, float *state, float *scratch, size_t scratch_floats, int step, float eps1,
                      float eps2, float clip_threshold, float decay_rate, void *stream) {
    DA_REQUIRE(n_params > 0 && n_blocks > 0 && param_table && block_table && flat && flat_grad && state && scratch,
               "da_adafactor_step: null argument");
    DA_REQUIRE(step >= 1, "da_adafactor_step: step counts from 1");
    static_assert(sizeof(AfParam) == 56 && sizeof(AfBlock) == 12, "table layouts are part of the ABI (see _lib.py)");
    hipStream_t st = (hipStream_t)stream;
    const AfParam *P = (const AfParam *)param_table;
    const AfBlock *B = (const AfBlock *)block_table;
    // scratch: [n_blocks] p^2 partials | [n_blocks] u^2 partials | [n_params][4] scalars | column partials
    const size_t head = 2 * (size_t)n_blocks + 4 * (size_t)n_params;
    DA_REQUIRE(scratch_floats >= head, "da_adafactor_step: scratch too small");
    float *part_p2 = scratch, *part_u2 = scratch + n_blocks, *scal = scratch + 2 * (size_t)n_blocks, *colpart = scratch + head;
    const float beta = (float)(1.0 - pow((double)step, (double)decay_rate));
    const float rho = (float)fmin(1e-2, 1.0 / sqrt((double)step));
    k_af_a<<<n_blocks, 256, 0, st>>>(P, B, flat, flat_grad, state, colpart, part_p2, part_u2, beta, eps1);
    k_af_b<<<dim3((unsigned)n_params, (64 * AF_CPL + AF_BCH - 1) / AF_BCH), 256, 0, st>>>(P, state, colpart, part_p2, scal, beta, rho, eps2);
    k_af_c<<<n_blocks, 256, 0, st>>>(P, B, flat_grad, state, scal, part_u2);
    k_af_d<<<n_blocks, 256, 0, st>>>(P, B, flat, flat_grad, state, scal, part_u2, clip_threshold);
    DA_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
