#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ void k(const unsigned short *in, unsigned short *out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, i = lane & 15;
    // each 16-lane group reads a [4 rows][16 cols] block: lane i' supplies the address of row (i' >> 2), cols 4 (i' & 3)
    const int row = grp * 4 + (i >> 2), col = 4 * (i & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + row * 64 + col));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short h[64 * 64], o[256];
    for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = r * 100 + c;
    unsigned short *d, *e;
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    return 0;
}
