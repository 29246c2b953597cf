// Effective shader clock under load, from OUTSIDE the workload: one wave of one workgroup spins for <seconds> and samples the
// shader-cycle counter (s_memtime) against the constant 100 MHz real-time counter (s_memrealtime) every <period_ms>; run it as a
// second process next to the workload (it occupies one wave slot of one CU).  rocm-smi's sclk is a requested P-state, not what a
// power-capped chip sustains (MI355X_MICROARCH.md "DVFS give-back").
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/bin/clock_probe ;  tools/bin/clock_probe 6 250
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_probe(unsigned long long *out, int nsamp, unsigned long long period_ticks) {
    if (threadIdx.x != 0) return;
    unsigned long long r0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int s = 0; s < nsamp; ++s) {
        unsigned long long r;
        do { __builtin_amdgcn_s_sleep(127); r = wall_clock64(); } while (r - r0 < period_ticks);
        const unsigned long long c = __builtin_readcyclecounter();
        out[2 * s] = c - c0; out[2 * s + 1] = r - r0;
        c0 = c; r0 = r;
    }
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0, period_ms = argc > 2 ? atof(argv[2]) : 250.0;
    const int nsamp = (int)(seconds * 1000.0 / period_ms);
    unsigned long long *d, *h = (unsigned long long *)malloc(16 * nsamp);
    hipMalloc(&d, 16 * nsamp);
    k_probe<<<1, 64>>>(d, nsamp, (unsigned long long)(period_ms * 1e5));
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16 * nsamp, hipMemcpyDeviceToHost);
    printf("effective shader clock (MHz) per %.0f ms:", period_ms);
    for (int s = 0; s < nsamp; ++s) printf(" %.0f", (double)h[2 * s] / ((double)h[2 * s + 1] / 100.0));
    printf("\n");
    return 0;
}
