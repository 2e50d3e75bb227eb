// permlane_lab.hip -- do v_permlane16_swap / v_permlane32_swap give lane ^ 16 / lane ^ 32 the way dev/linesearch.inc uses them?  (checked against __shfl_xor)
// build: hipcc --offload-arch=gfx950 -O3 -o permlane_lab.bin permlane_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned l = threadIdx.x & 63, v = threadIdx.x * 2654435761u + 12345u;
    auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const unsigned x32 = (l & 32) ? r32[0] : r32[1], x16 = (l & 16) ? r16[0] : r16[1];
    o[threadIdx.x * 4 + 0] = x32; o[threadIdx.x * 4 + 1] = (unsigned)__shfl_xor((int)v, 32, 64);
    o[threadIdx.x * 4 + 2] = x16; o[threadIdx.x * 4 + 3] = (unsigned)__shfl_xor((int)v, 16, 64);
}
int main() {
    unsigned *d, h[256 * 4]; (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad32 = 0, bad16 = 0;
    for (int i = 0; i < 256; i++) { bad32 += h[4 * i] != h[4 * i + 1]; bad16 += h[4 * i + 2] != h[4 * i + 3]; }
    printf("permlane32_swap as lane^32: %d mismatches; permlane16_swap as lane^16: %d mismatches (of 256)\n", bad32, bad16);
    return bad32 + bad16 ? 1 : 0;
}
