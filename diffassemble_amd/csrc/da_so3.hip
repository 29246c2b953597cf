// 3D (Breaking-Bad) tail of the path: the two pose heads of Eff_GAT_3d
// (efficient_gat_3d.py:207-219: mlp_t / mlp_r second Linear, matrix_exp(vec2skew(r)),
// matrix_to_quaternion, L2 normalise) and the SE(3) DDIM update
// (spatial_diffusion_3d_test_double_diffusion.py:595-685 with so3_scale / log_rmat of
// utils_3d.py:1018-1061).  A few dozen flops per piece: one thread per piece, fp32.
//
// matrix_exp of a skew matrix is evaluated in closed form (Rodrigues); torch.matrix_exp uses
// a scaled Taylor/Pade series -- equal to ~1e-7.  log_rmat's NaN branch (rotation by exactly
// pi, where the reference falls back to torch.linalg.eigh) takes the axis from (R + I)/2
// instead; the two agree up to the sign of the axis, which is undetermined at pi.
#include "da_common.h"
#include "da_internal.h"

namespace da {

struct M3 { float m[9]; };

__device__ inline M3 rodrigues(float v0, float v1, float v2) {   // exp(vec2skew(v))
    const float th2 = v0 * v0 + v1 * v1 + v2 * v2;
    const float th = sqrtf(th2);
    float a, b;
    if (th < 1e-4f) {
        a = 1.0f - th2 * (1.0f / 6.0f);
        b = 0.5f - th2 * (1.0f / 24.0f);
    } else {
        a = sinf(th) / th;
        const float h = sinf(0.5f * th) / (0.5f * th);
        b = 0.5f * h * h;
    }
    M3 r;
    // I + a*S + b*(v v^T - th2 I), S = [[0,-v2,v1],[v2,0,-v0],[-v1,v0,0]]
    r.m[0] = 1.0f + b * (v0 * v0 - th2); r.m[1] = -a * v2 + b * v0 * v1;     r.m[2] = a * v1 + b * v0 * v2;
    r.m[3] = a * v2 + b * v0 * v1;       r.m[4] = 1.0f + b * (v1 * v1 - th2); r.m[5] = -a * v0 + b * v1 * v2;
    r.m[6] = -a * v1 + b * v0 * v2;      r.m[7] = a * v0 + b * v1 * v2;      r.m[8] = 1.0f + b * (v2 * v2 - th2);
    return r;
}

__device__ inline void mat_to_quat(const M3 &R, float q[4]) {    // pytorch3d matrix_to_quaternion
    const float m00 = R.m[0], m01 = R.m[1], m02 = R.m[2], m10 = R.m[3], m11 = R.m[4], m12 = R.m[5],
                m20 = R.m[6], m21 = R.m[7], m22 = R.m[8];
    float qa[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
    int best = 0;
    for (int i = 0; i < 4; ++i) qa[i] = qa[i] > 0.f ? sqrtf(qa[i]) : 0.f;
    for (int i = 1; i < 4; ++i) if (qa[i] > qa[best]) best = i;      // argmax: first maximum
    float c[4];
    if (best == 0) { c[0] = qa[0] * qa[0]; c[1] = m21 - m12; c[2] = m02 - m20; c[3] = m10 - m01; }
    else if (best == 1) { c[0] = m21 - m12; c[1] = qa[1] * qa[1]; c[2] = m10 + m01; c[3] = m02 + m20; }
    else if (best == 2) { c[0] = m02 - m20; c[1] = m10 + m01; c[2] = qa[2] * qa[2]; c[3] = m12 + m21; }
    else { c[0] = m10 - m01; c[1] = m20 + m02; c[2] = m21 + m12; c[3] = qa[3] * qa[3]; }
    const float den = 2.0f * fmaxf(qa[best], 0.1f);
    const float sgn = (c[0] / den) < 0.f ? -1.f : 1.f;              // standardize_quaternion
    for (int i = 0; i < 4; ++i) q[i] = sgn * (c[i] / den);
}

__device__ inline M3 quat_to_mat(const float q[4]) {             // pytorch3d quaternion_to_matrix
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    M3 o;
    o.m[0] = 1 - two_s * (j * j + k * k); o.m[1] = two_s * (i * j - k * r);     o.m[2] = two_s * (i * k + j * r);
    o.m[3] = two_s * (i * j + k * r);     o.m[4] = 1 - two_s * (i * i + k * k); o.m[5] = two_s * (j * k - i * r);
    o.m[6] = two_s * (i * k - j * r);     o.m[7] = two_s * (j * k + i * r);     o.m[8] = 1 - two_s * (i * i + j * j);
    return o;
}

__device__ inline M3 so3_scale(const M3 &R, float sc) {          // matrix_exp(sc * log_rmat(R))
    // log_rmat, utils_3d.py:1018-1046
    const float k0 = R.m[7] - R.m[5];      // skew[2,1]
    const float k1 = -(R.m[6] - R.m[2]);   // -skew[2,0]
    const float k2 = R.m[3] - R.m[1];      // skew[1,0]
    const float s_angle = sqrtf(k0 * k0 + k1 * k1 + k2 * k2) * 0.5f;
    const float c_angle = (R.m[0] + R.m[4] + R.m[8] - 1.0f) * 0.5f;
    const float angle = atan2f(s_angle, c_angle);
    float w0, w1, w2;
    if (angle == 0.0f) {
        w0 = w1 = w2 = 0.f;
    } else if (s_angle == 0.0f) {          // rotation by pi: axis from (R + I) / 2
        float d0 = (R.m[0] + 1.f) * 0.5f, d1 = (R.m[4] + 1.f) * 0.5f, d2 = (R.m[8] + 1.f) * 0.5f;
        float a0, a1, a2;
        if (d0 >= d1 && d0 >= d2) { a0 = sqrtf(fmaxf(d0, 0.f)); a1 = (R.m[1] + R.m[3]) * 0.25f / a0; a2 = (R.m[2] + R.m[6]) * 0.25f / a0; }
        else if (d1 >= d2) { a1 = sqrtf(fmaxf(d1, 0.f)); a0 = (R.m[1] + R.m[3]) * 0.25f / a1; a2 = (R.m[5] + R.m[7]) * 0.25f / a1; }
        else { a2 = sqrtf(fmaxf(d2, 0.f)); a0 = (R.m[2] + R.m[6]) * 0.25f / a2; a1 = (R.m[5] + R.m[7]) * 0.25f / a2; }
        w0 = angle * a0; w1 = angle * a1; w2 = angle * a2;
    } else {
        const float scale = angle / (2.0f * s_angle);
        w0 = scale * k0; w1 = scale * k1; w2 = scale * k2;     // skew2vec(scale * skew)
    }
    return rodrigues(sc * w0, sc * w1, sc * w2);
}

__device__ inline M3 matmul(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[i * 3 + j] = A.m[i * 3] * B.m[j] + A.m[i * 3 + 1] * B.m[3 + j] + A.m[i * 3 + 2] * B.m[6 + j];
    return C;
}
__device__ inline M3 transpose(const M3 &A) {
    M3 C;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i * 3 + j] = A.m[j * 3 + i];
    return C;
}

// One wave per piece: 6 dot products of length 256 (mlp_t.2 / mlp_r.2), then lane 0 finishes.
template <typename T>
__global__ __launch_bounds__(256) void k_head3d(int n, const T *__restrict__ hh, const float *__restrict__ wt,
                                                const float *__restrict__ bt, const float *__restrict__ wr,
                                                const float *__restrict__ br, float *__restrict__ out7,
                                                float *__restrict__ pre) {
    const int lane = threadIdx.x & 63;
    const int r = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (r >= n) return;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    const T *ht = hh + (size_t)r * 512, *hr = ht + 256;
    for (int k = lane; k < 256; k += 64) {
        const float a = ldf(ht + k), b = ldf(hr + k);
#pragma unroll
        for (int c = 0; c < 3; ++c) { acc[c] = fmaf(wt[c * 256 + k], a, acc[c]); acc[3 + c] = fmaf(wr[c * 256 + k], b, acc[3 + c]); }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
        for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
    if (lane != 0) return;
    const float t0 = acc[0] + bt[0], t1 = acc[1] + bt[1], t2 = acc[2] + bt[2];
    const float r0 = acc[3] + br[0], r1 = acc[4] + br[1], r2 = acc[5] + br[2];
    if (pre) { float *p = pre + (size_t)r * 6; p[0] = r0; p[1] = r1; p[2] = r2; p[3] = t0; p[4] = t1; p[5] = t2; }
    float q[4];
    mat_to_quat(rodrigues(r0, r1, r2), q);
    const float nrm = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);  // F.normalize eps
    float *o = out7 + (size_t)r * 7;
    o[0] = q[0] / nrm; o[1] = q[1] / nrm; o[2] = q[2] / nrm; o[3] = q[3] / nrm;
    o[4] = t0; o[5] = t1; o[6] = t2;
}

__global__ __launch_bounds__(64) void k_ddim3d(DeviceSchedule s, int mean_type, int n, const float *__restrict__ x,
                                               const float *__restrict__ mo, const int64_t *__restrict__ t,
                                               int64_t t_scalar, int ratio, int prev_all_nonneg,
                                               float *__restrict__ x_prev) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int64_t ti = t ? t[r] : t_scalar;
    ti = ti < 0 ? 0 : (ti >= s.steps ? s.steps - 1 : ti);
    const int64_t tp = ti - ratio;
    const float ap = s.alphas_cumprod[ti];
    const float ap_prev = (prev_all_nonneg && tp >= 0) ? s.alphas_cumprod[tp] : 1.0f;
    const float beta = 1.0f - ap;
    float xv[7], x0[7];
    for (int c = 0; c < 7; ++c) {
        xv[c] = x[(size_t)r * 7 + c];
        const float m = mo[(size_t)r * 7 + c];
        x0[c] = mean_type == DA_MEAN_START_X ? m : (xv[c] - sqrtf(beta) * m) / sqrtf(ap);
    }
    const float sr = s.sqrt_recip_alphas_cumprod[ti], srm1 = s.sqrt_recipm1_alphas_cumprod[ti];
    const float sq_prev = sqrtf(ap_prev), sq_dir = sqrtf(1.0f - ap_prev);
    float *o = x_prev + (size_t)r * 7;
    for (int c = 4; c < 7; ++c) {
        const float eps = (sr * xv[c] - x0[c]) / srm1;
        o[c] = sq_prev * x0[c] + sq_dir * eps;
    }
    const M3 Rx = quat_to_mat(xv), R0 = quat_to_mat(x0);
    const M3 xt_term = so3_scale(Rx, sr / srm1);
    const M3 x0_term = so3_scale(R0, 1.0f / srm1);
    float qe[4];
    mat_to_quat(matmul(xt_term, transpose(x0_term)), qe);
    const M3 dir = so3_scale(quat_to_mat(qe), sq_dir);
    float qp[4];
    mat_to_quat(matmul(so3_scale(R0, sq_prev), dir), qp);
    o[0] = qp[0]; o[1] = qp[1]; o[2] = qp[2]; o[3] = qp[3];
}

int launch_head3d(int prec, int n, const void *hh, const float *wt, const float *bt, const float *wr, const float *br,
                  float *out7, float *pre_head, hipStream_t st) {
    if (n <= 0) return 0;
    const int grid = (int)(((size_t)n * 64 + 255) / 256);
    if (prec == DA_PREC_BF16)
        k_head3d<bf16_t><<<grid, 256, 0, st>>>(n, (const bf16_t *)hh, wt, bt, wr, br, out7, pre_head);
    else
        k_head3d<float><<<grid, 256, 0, st>>>(n, (const float *)hh, wt, bt, wr, br, out7, pre_head);
    DA_LAUNCH_CHECK();
    return 0;
}

int launch_ddim3d(const DeviceSchedule &s, int mean_type, int n, const float *x, const float *mo, const int64_t *t,
                  int64_t t_scalar, int ratio, int prev_all_nonneg, float *x_prev, hipStream_t st) {
    if (n <= 0) return 0;
    k_ddim3d<<<(n + 63) / 64, 64, 0, st>>>(s, mean_type, n, x, mo, t, t_scalar, ratio, prev_all_nonneg, x_prev);
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace da
