// Store-pattern probe (MI355X): how fast can 256 workgroups write an [M, N] bf16 matrix when the bytes leave in the
// shapes the projection kernels use?   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/bin/store_probe
//   mode 0: linear -- every workgroup streams a contiguous slab, 16 B per lane, 1 KB per wave-instruction
//   mode 1: strip  -- k_gemm_wreg's LDS-strip epilogue: a wave owns 32 columns (64 B per row); one instruction = 16 rows x 64 B
//   mode 2: direct -- register-direct epilogue: lane (row, half) writes 2 x 16 B at row * stride + 32 * half
//   mode 3: strip2 -- k_gemm_wreg2's: a wave owns 64 columns; one instruction = 8 rows x 128 B
//   mode 5: strip + LDS-DMA loads -- mode 1, and every wave also pulls 2 KB of a [M, 256] bf16 input tile into LDS per tile (16 KB per
//           workgroup and tile, the traffic of k_gemm_wreg's producer), at most 8 vector-memory instructions in flight per wave
//   mode 6: the loads of mode 5 alone
//   mode 4: wide   -- a workgroup's 8 waves together own 256 columns, wave w writes rows 4 w .. 4 w + 3 of the tile, 512 B per row
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void k_store(unsigned short *out, int M, int N, int mode, int tiles_per_wg, const char *xin, int ncg, int cpx) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[4][16384];
    const int lane = threadIdx.x & 63, wid = (threadIdx.x >> 6) & 7, wset = threadIdx.x >> 9, nset = blockDim.x >> 9;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    const int nrt = M / 32;
    if (mode == 0) {
        const size_t total = (size_t)M * N * 2, per = total / (gridDim.x * gridDim.y);
        const size_t b0 = (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * per;
        for (size_t o = (size_t)threadIdx.x * 16; o < per; o += 512 * 16) *(u32x4 *)((char *)out + b0 + o) = v;
        return;
    }
    int cgi = blockIdx.x, chunk = blockIdx.y;
    if (cpx > 0) {                              // XCD-aware: the ncg column groups of a row chunk on one XCD
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        cgi = slot % ncg; chunk = xcd * cpx + slot / ncg;
    }
    const int t0 = chunk * tiles_per_wg, t1 = min(t0 + tiles_per_wg, nrt);
    const size_t rs = (size_t)N * 2;
    if (mode == 15 || mode == 16) {
        // mode 12 with FOUR accumulators carried across tiles, used round-robin: the registers a store reads are not written
        // again for three tiles (does a backed-up store hold up the MFMA that overwrites its data registers?); 16 = without the stores
        typedef __attribute__((ext_vector_type(16))) float f32x16;
        typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
        f32x16 acc[4] = {{0}, {0}, {0}, {0}};
        bf16x8 a = __builtin_bit_cast(bf16x8, v), b = __builtin_bit_cast(bf16x8, v);
        const size_t rs = (size_t)N * 2;
        for (int t = t0 + wset; t < t1; t += 4 * nset) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + u * nset;
                if (tt >= t1) break;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
                char *q = (char *)out + (size_t)tt * 32 * rs + (size_t)cgi * 512 + (size_t)(lane & 31) * rs + wid * 64 + (lane >> 5) * 32;
                u32x4 o0 = {__builtin_bit_cast(unsigned, acc[u][0]), __builtin_bit_cast(unsigned, acc[u][1]), __builtin_bit_cast(unsigned, acc[u][2]), __builtin_bit_cast(unsigned, acc[u][3])};
                u32x4 o1 = {__builtin_bit_cast(unsigned, acc[u][4]), __builtin_bit_cast(unsigned, acc[u][5]), __builtin_bit_cast(unsigned, acc[u][6]), __builtin_bit_cast(unsigned, acc[u][7])};
                if (mode == 15) { *(u32x4 *)q = o0; *(u32x4 *)(q + 16) = o1; }
            }
        }
        if (mode == 16) { asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); }
        return;
    }
    for (int t = t0 + wset; t < t1; t += nset) {
        char *tile = (char *)out + (size_t)t * 32 * rs + (size_t)cgi * 512;
        if (mode == 1) {
            for (int k = 0; k < 2; ++k) *(u32x4 *)(tile + (size_t)((lane >> 2) + 16 * k) * rs + wid * 64 + (lane & 3) * 16) = v;
        } else if (mode == 2) {
            char *p = tile + (size_t)(lane & 31) * rs + wid * 64 + (lane >> 5) * 32;
            *(u32x4 *)p = v; *(u32x4 *)(p + 16) = v;
        } else if (mode == 3) {
            if (wid < 4) for (int k = 0; k < 4; ++k) *(u32x4 *)(tile + (size_t)((lane >> 3) + 8 * k) * rs + wid * 128 + (lane & 7) * 16) = v;
        } else if (mode == 5 || mode == 6 || mode == 11) {
            // mode 11: mode 5 with every load served by the same 64 KB (always cache hits)
            const char *src = xin + (size_t)(mode == 11 ? (t & 3) : t) * 16384 + wid * 2048 + lane * 16;
            for (int k = 0; k < 2; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * 1024),
                                                 (__attribute__((address_space(3))) void *)(&ring[t & 3][wid * 2048 + k * 1024]), 16, 0, 0);
            if (mode != 6) for (int k = 0; k < 2; ++k) *(u32x4 *)(tile + (size_t)((lane >> 2) + 16 * k) * rs + wid * 64 + (lane & 3) * 16) = v;
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (mode == 7) {
            // split roles: waves 0 .. 3 load (4 KB each, they wait for their own loads only), waves 4 .. 7 store (two waves' columns each, never wait)
            if (wid < 4) {
                const char *src = xin + (size_t)t * 16384 + wid * 4096 + lane * 16;
                for (int k = 0; k < 4; ++k)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * 1024),
                                                     (__attribute__((address_space(3))) void *)(&ring[t & 3][wid * 4096 + k * 1024]), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                for (int k = 0; k < 4; ++k) *(u32x4 *)(tile + (size_t)((lane >> 3) + 8 * k) * rs + (wid - 4) * 128 + (lane & 7) * 16) = v;
            }
        } else if (mode == 8) {
            // mode 5 with plain loads into registers
            const char *src = xin + (size_t)t * 16384 + wid * 2048 + lane * 16;
            u32x4 r0 = *(const u32x4 *)src, r1 = *(const u32x4 *)(src + 1024);
            for (int k = 0; k < 2; ++k) *(u32x4 *)(tile + (size_t)((lane >> 2) + 16 * k) * rs + wid * 64 + (lane & 3) * 16) = v;
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("" ::"v"(r0), "v"(r1));
        } else if (mode == 9) {
            // mode 5, 32 instructions in flight per wave
            const char *src = xin + (size_t)t * 16384 + wid * 2048 + lane * 16;
            for (int k = 0; k < 2; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * 1024),
                                                 (__attribute__((address_space(3))) void *)(&ring[t & 3][wid * 2048 + k * 1024]), 16, 0, 0);
            for (int k = 0; k < 2; ++k) *(u32x4 *)(tile + (size_t)((lane >> 2) + 16 * k) * rs + wid * 64 + (lane & 3) * 16) = v;
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        } else if (mode == 10) {
            // mode 7 with one loader wave (16 KB per tile, k_gemm_wreg's producer) and seven storing waves that never wait
            if (wid == 0) {
                const char *src = xin + (size_t)t * 16384 + lane * 16;
                for (int k = 0; k < 16; ++k)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * 1024),
                                                     (__attribute__((address_space(3))) void *)(&ring[t & 3][k * 1024]), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            } else if (wid < 5) {
                for (int k = 0; k < 4; ++k) *(u32x4 *)(tile + (size_t)((lane >> 3) + 8 * k) * rs + (wid - 1) * 128 + (lane & 7) * 16) = v;
            }
        } else if (mode == 12 || mode == 13 || mode == 14) {
            // arithmetic + stores, no loads: 16 dependent MFMAs per tile and wave, then (12, 14) the two direct stores of mode 2;
            // 13 = the MFMAs alone; 14 = the stores of tile t issued after the MFMAs of tile t + 1 were queued
            typedef __attribute__((ext_vector_type(16))) float f32x16;
            typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
            f32x16 acc = {0};
            bf16x8 a = __builtin_bit_cast(bf16x8, v), b = __builtin_bit_cast(bf16x8, v);
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            u32x4 o0 = {__builtin_bit_cast(unsigned, acc[0]), __builtin_bit_cast(unsigned, acc[1]), __builtin_bit_cast(unsigned, acc[2]), __builtin_bit_cast(unsigned, acc[3])};
            u32x4 o1 = {__builtin_bit_cast(unsigned, acc[4]), __builtin_bit_cast(unsigned, acc[5]), __builtin_bit_cast(unsigned, acc[6]), __builtin_bit_cast(unsigned, acc[7])};
            char *q = tile + (size_t)(lane & 31) * rs + wid * 64 + (lane >> 5) * 32;
            if (mode != 13) { *(u32x4 *)q = o0; *(u32x4 *)(q + 16) = o1; }
            else asm volatile("" ::"v"(o0), "v"(o1));
        } else if (mode == 4) {
            for (int k = 0; k < 2; ++k) *(u32x4 *)(tile + (size_t)(4 * wid + 2 * k + (lane >> 5)) * rs + (lane & 31) * 16) = v;
        }
    }
}

__global__ __launch_bounds__(512) void k_store_cu(char *out, unsigned per, int waves) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wid >= waves) return;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    char *base = out + (size_t)blockIdx.x * per;
    for (unsigned o = (wid * 64 + lane) * 16; o < per; o += waves * 1024) *(u32x4 *)(base + o) = v;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 57600, N = argc > 2 ? atoi(argv[2]) : 2560;
    unsigned short *out; CK(hipMalloc(&out, (size_t)M * N * 2));
    char *xin; CK(hipMalloc(&xin, (size_t)M * 512)); CK(hipMemset(xin, 1, (size_t)M * 512));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int ncg = N / 256, nrt = M / 32;
    int nchunk = 256 / ncg; const int tiles = (nrt + nchunk - 1) / nchunk; nchunk = (nrt + tiles - 1) / tiles;
    const int xcdmap = argc > 3 ? atoi(argv[3]) : 0;
    int cpx = 0;
    dim3 grid(ncg, nchunk);
    int tiles_k = tiles;
    if (xcdmap) { cpx = 32 / ncg; tiles_k = (nrt + 8 * cpx - 1) / (8 * cpx); grid = dim3(8 * cpx * ncg); }
    const int threads = argc > 4 ? atoi(argv[4]) : 512;
    const int only = argc > 5 ? atoi(argv[5]) : -1;
    for (int mode = 0; mode < 17; ++mode) {
        if (only >= 0 && mode != only) continue;
        for (int r = 0; r < 3; ++r) k_store<<<grid, threads>>>(out, M, N, mode, tiles_k, xin, ncg, cpx);
        CK(hipDeviceSynchronize());
        const int iters = getenv("STORE_PROBE_ITERS") ? atoi(getenv("STORE_PROBE_ITERS")) : 20;   // long loops: power sampling
        CK(hipEventRecord(a));
        for (int r = 0; r < iters; ++r) k_store<<<grid, threads>>>(out, M, N, mode, tiles_k, xin, ncg, cpx);
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("M=%d N=%d mode %d grid %dx%d: %.1f us  %.2f TB/s\n", M, N, mode, ncg, nchunk, ms / iters * 1e3, (double)M * N * 2 / (ms / iters * 1e-3) / 1e12);
    }
    // per-CU store rate when the chip is NOT saturated: `nwg` workgroups (one per CU), each streaming 4 MB in the linear shape,
    // with 8 / 4 / 2 / 1 waves per workgroup active
    if (argc > 6) for (int nwg : {8, 32, 256}) for (int waves : {8, 2, 1}) {
        const size_t bytes = (size_t)nwg * (1u << 20);
        for (int r = 0; r < 2; ++r) k_store_cu<<<nwg, 512>>>((char *)out, 1u << 20, waves);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int r = 0; r < 10; ++r) k_store_cu<<<nwg, 512>>>((char *)out, 1u << 20, waves);
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double Bps = (double)bytes / (ms / 10 * 1e-3);
        printf("%3d workgroups x %d storing waves: %.2f TB/s = %.1f GB/s per CU = %.1f B/clk/CU at 2.1 GHz\n", nwg, waves, Bps / 1e12, Bps / nwg / 1e9, Bps / nwg / 2.1e9);
    }
    return 0;
}
