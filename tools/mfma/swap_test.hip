#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o0, unsigned *o1) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o0[threadIdx.x] = r[0];
    o1[threadIdx.x] = r[1];
}
int main() {
    unsigned *d0, *d1, h0[64], h1[64];
    hipMalloc(&d0, 256); hipMalloc(&d1, 256);
    k<<<1, 64>>>(d0, d1);
    hipMemcpy(h0, d0, 256, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, 256, hipMemcpyDeviceToHost);
    printf("r0: lane0=%u lane1=%u lane32=%u lane33=%u\n", h0[0], h0[1], h0[32], h0[33]);
    printf("r1: lane0=%u lane1=%u lane32=%u lane33=%u\n", h1[0], h1[1], h1[32], h1[33]);
    return 0;
}
