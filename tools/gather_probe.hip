// FETCH_SIZE calibration for the edge-list attention's access pattern (MI355X_MICROARCH.md: the x2 correction of FETCH_SIZE is
// measured for 16 B / lane streaming reads; "calibrate on a known byte count in your own access pattern"): a wave reads random
// ROWS of a table larger than the Infinity Cache the way k_attn_csr does -- 512-byte rows as one 8-byte load per lane (the hidden
// layers' K / V rows) or 2304-byte rows as nine 4-byte loads per lane at a 36-byte lane stride (the last layer's) -- a known number
// of bytes.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; prints the bytes the kernel really asked for.
//   tools/bin/gather_probe <mode 0|1> [table_MiB] [reads_per_wave]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const unsigned char *tab, const int *idx, int per_wave, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int *my = idx + wave * per_wave;
    unsigned acc = 0;
    for (int r = 0; r < per_wave; r += 4) {           // four independent rows in flight, as the kernel keeps
        unsigned t[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = my[r + u];
            if (MODE == 0) {
                const uint2 v = *(const uint2 *)(tab + (size_t)row * 512 + lane * 8);
                t[u] = v.x ^ v.y;
            } else {
                const unsigned *p = (const unsigned *)(tab + (size_t)row * 2304 + lane * 36);
#pragma unroll
                for (int k = 0; k < 9; ++k) t[u] ^= p[k];
            }
        }
        acc ^= t[0] ^ t[1] ^ t[2] ^ t[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t mib = argc > 2 ? atol(argv[2]) : 1024;
    const int per_wave = argc > 3 ? atoi(argv[3]) : 256;
    const int rowb = mode == 0 ? 512 : 2304;
    const size_t rows = mib * 1024 * 1024 / rowb;
    const int nwg = 256 * 8, nwave = nwg * 4;
    unsigned char *tab; int *idx; unsigned *sink;
    CK(hipMalloc(&tab, rows * rowb)); CK(hipMemset(tab, 1, rows * rowb));
    std::vector<int> h((size_t)nwave * per_wave);
    unsigned s = 777;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (int)((s >> 4) % rows); }
    CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&sink, 4));
    for (int it = 0; it < 3; ++it) {
        if (mode == 0) k_gather<0><<<nwg, 256>>>(tab, idx, per_wave, sink); else k_gather<1><<<nwg, 256>>>(tab, idx, per_wave, sink);
    }
    CK(hipDeviceSynchronize());
    printf("mode %d: %zu-MiB table of %d-byte rows, %d waves x %d random rows -> %.1f MB of row bytes per launch (+ %.1f MB of indices)\n", mode, mib, rowb, nwave,
           per_wave, (double)nwave * per_wave * rowb / 1e6, (double)nwave * per_wave * 4 / 1e6);
    return 0;
}
