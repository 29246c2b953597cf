// Greedy position assignment on the device: the step that turns predicted poses into the
// reported accuracy right after every sampling loop.
//
// Replaces greedy_cost_assignment, spatial_diffusion.py:179-216 -- a TorchScript `while` loop that takes the
// globally smallest remaining distance of an n x m matrix, retires its row and column, and pays a
// `mask.nonzero()` + `.item()` host sync per assignment (900 syncs over 810 000 elements for a 30 x 30
// puzzle, twice per puzzle).  Here: one workgroup per puzzle, no host round trips.  Each row keeps its
// smallest free column (value, index); an iteration is (a) a workgroup argmin over the row minima,
// (b) retire that row and column, (c) rows whose cached minimum was the retired column rescan their free
// columns -- a wave per row.  Ties resolve like the reference's `dist[mask].min()`: first in row-major order
// (smallest row, then smallest column).  Distances are sqrt(dx*dx + dy*dy) in fp32, never materialised.
#include "da_common.h"

namespace da {

__device__ __forceinline__ float dist2(const float *a, const float *b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1];
    const float d = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    // a diverged sample (NaN / Inf poses) must still be assignable: +inf is the "retired" marker of rmin and a NaN
    // never wins a comparison, so non-finite distances rank as the largest finite value (the reference returns
    // some wrong assignment in that case; here the row simply goes last, and nothing indexes out of LDS)
    return d <= 3.402823466e38f ? d : 3.402823466e38f;
}

// lexicographic (value, index) minimum
__device__ __forceinline__ void lexmin(float &v, int &i, float v2, int i2) {
    if (v2 < v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

__global__ __launch_bounds__(1024) void k_greedy_assign(const float *__restrict__ pos1, int ld1, const float *__restrict__ pos2,
                                                        int ld2, const int32_t *__restrict__ ptr1,
                                                        const int32_t *__restrict__ ptr2, long long *__restrict__ out) {
    extern __shared__ float sm[];
    const int g = blockIdx.x;
    const int r0 = ptr1[g], n = ptr1[g + 1] - r0, c0 = ptr2[g], m = ptr2[g + 1] - c0;
    float *p2 = sm;                                    // [m][2]
    float *rmin = p2 + 2 * m;                          // [n] smallest free distance of the row (+inf once retired)
    int *rarg = (int *)(rmin + n);                     // [n] its column
    int *cused = rarg + n;                             // [m]
    float *redv = (float *)(cused + m);                // [16]
    int *redi = (int *)(redv + 16);                    // [16]
    int *bcast = redi + 16;                            // [2]: chosen row, chosen column
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    for (int j = tid; j < m; j += blockDim.x) {
        p2[2 * j] = pos2[(size_t)(c0 + j) * ld2];
        p2[2 * j + 1] = pos2[(size_t)(c0 + j) * ld2 + 1];
        cused[j] = 0;
    }
    __syncthreads();
    auto rescan = [&](int i) {                         // one wave: smallest free column of row i
        const float a[2] = {pos1[(size_t)(r0 + i) * ld1], pos1[(size_t)(r0 + i) * ld1 + 1]};
        float v = INFINITY;
        int c = 0x7fffffff;
        for (int j = lane; j < m; j += 64)
            if (!cused[j]) lexmin(v, c, dist2(a, p2 + 2 * j), j);
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(v, o);
            const int c2 = __shfl_xor(c, o);
            lexmin(v, c, v2, c2);
        }
        if (lane == 0) { rmin[i] = v; rarg[i] = c; }
    };
    for (int i = wv; i < n; i += nw) rescan(i);
    __syncthreads();
    const int total = n < m ? n : m;
    for (int k = 0; k < total; ++k) {
        // (a) argmin over rows, ties -> smallest row (its cached column is already the smallest of the row)
        float v = INFINITY;
        int i = 0x7fffffff;
        for (int r = tid; r < n; r += blockDim.x) lexmin(v, i, rmin[r], r);
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(v, o);
            const int i2 = __shfl_xor(i, o);
            lexmin(v, i, v2, i2);
        }
        if (lane == 0) { redv[wv] = v; redi[wv] = i; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < nw; ++w) lexmin(v, i, redv[w], redi[w]);
            if (i >= n) {                                        // cannot happen with finite rmin; never index past LDS
                i = 0;
                while (i < n - 1 && !(rmin[i] < INFINITY)) ++i;
            }
            int j = rarg[i];
            j = j < 0 ? 0 : (j >= m ? m - 1 : j);
            bcast[0] = i; bcast[1] = j;
            out[((size_t)r0 + k) * 3] = i;
            out[((size_t)r0 + k) * 3 + 1] = j;
            out[((size_t)r0 + k) * 3 + 2] = (long long)v;          // the reference stores it in an int64 tensor
            rmin[i] = INFINITY;
            cused[j] = 1;
        }
        __syncthreads();
        // (c) rows that pointed at the retired column look again
        const int jr = bcast[1];
        for (int r = wv; r < n; r += nw)
            if (rmin[r] < INFINITY && rarg[r] == jr) rescan(r);
        __syncthreads();
    }
}

}  // namespace da

using namespace da;

extern "C" int da_greedy_assign(int n_puzzles, const float *pos1, int ld1, const float *pos2, int ld2,
                                const int32_t *ptr1, const int32_t *ptr2, int max_n, int max_m, long long *out,
                                void *stream) {
    DA_REQUIRE(n_puzzles > 0 && pos1 && pos2 && ptr1 && ptr2 && out, "da_greedy_assign: null argument");
    DA_REQUIRE(ld1 >= 2 && ld2 >= 2 && max_n > 0 && max_m > 0, "da_greedy_assign: bad sizes");
    const size_t lds = (size_t)(2 * max_m + 2 * max_n + max_m + 64) * 4;
    DA_REQUIRE(lds <= 160 * 1024, "da_greedy_assign: puzzle too large for one workgroup (%d x %d)", max_n, max_m);
    static bool attr = false;
    if (!attr && lds > 48 * 1024) {
        DA_CHECK_HIP(hipFuncSetAttribute((const void *)k_greedy_assign, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    k_greedy_assign<<<n_puzzles, 1024, lds, (hipStream_t)stream>>>(pos1, ld1, pos2, ld2, ptr1, ptr2, out);
    DA_LAUNCH_CHECK();
    return 0;
}
