// Exphander graphs on the device: the adjacency bit rows the masked attention reads, straight from
// the permutations the reference's generator draws (/root/reference/puzzle_diff/dataset/puzzle_dataset.py:115-152:
// nodes[p] <-> nodes[(p - k) mod n], k = 1 .. d / 2, plus nodes[p] <-> nodes[p + n / 2] for odd d, symmetrised).  In
// position space the graph is a circulant band: bit (i, j) = [cyclic distance of pos(i), pos(j) in [1, d / 2]  or  = n / 2].
// Two launches per Batch (inverse permutation, bit rows); everything else of an expander plan depends on the Batch SHAPE
// only and is cached by the host (diffassemble_amd/graph_plan.py expander_plan).
#include "da_internal.h"

namespace da {

__global__ __launch_bounds__(256) void k_exp_pos(long long total, int n, const long long *__restrict__ perms, int32_t *__restrict__ pos) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long g = i / n;
    const long long node = perms[i];
    if (node >= 0 && node < n) pos[g * n + node] = (int32_t)(i - g * n);
}

// one thread per mask byte: graph g, target row i, sources 8 b .. 8 b + 7
__global__ __launch_bounds__(256) void k_exp_mask(long long total, int n, int row_bytes, int reps, int odd,
                                                  const int32_t *__restrict__ pos, unsigned char *__restrict__ mask) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int b = (int)(t % row_bytes);
    const long long gi = t / row_bytes;                 // g * n + i
    const long long g = gi / n;
    const int pi = pos[gi];
    unsigned v = 0;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        const int j = 8 * b + bit;
        if (j < n) {
            int dist = pi - pos[g * n + j];
            dist += dist < 0 ? n : 0;
            const int cd = min(dist, n - dist);
            if ((cd >= 1 && cd <= reps) || (odd && 2 * cd == n)) v |= 1u << bit;
        }
    }
    mask[t] = (unsigned char)v;
}

}  // namespace da

using namespace da;

extern "C" int da_expander_mask(int n_graphs, int n, int degree, const int64_t *perms, int32_t *pos, int row_bytes,
                                unsigned char *mask, void *stream) {
    DA_REQUIRE(perms && pos && mask && n_graphs > 0 && n > 0 && degree > 0 && degree < n, "da_expander_mask: bad argument");
    DA_REQUIRE(row_bytes * 8 >= n, "da_expander_mask: row stride %d bytes cannot hold %d sources", row_bytes, n);
    hipStream_t st = (hipStream_t)stream;
    const long long np = (long long)n_graphs * n, nb = np * row_bytes;
    k_exp_pos<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(np, n, (const long long *)perms, pos);
    k_exp_mask<<<(unsigned)((nb + 255) / 256), 256, 0, st>>>(nb, n, row_bytes, degree / 2, degree & 1, pos, mask);
    DA_LAUNCH_CHECK();
    return 0;
}
