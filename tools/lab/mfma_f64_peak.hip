// fp64 MFMA peak on this chip: back-to-back v_mfma_f64_16x16x4_f64 on independent accumulators, W waves per SIMD on every CU.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/lab/mfma_f64_peak.hip -o tools/lab/mfma_f64_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dvec4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_peak(int iters, double *out) {
    dvec4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (dvec4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}
int main() {
    double *out; hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", ncu, prop.clockRate);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int iters = 20000, nacc = 8;
        const int blocks = ncu, threads = 64 * 4 * wps;
        hipLaunchKernelGGL(k_peak<8>, dim3(blocks), dim3(threads), 0, 0, 100, out);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_peak<8>, dim3(blocks), dim3(threads), 0, 0, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)blocks * 4 * wps * iters * nacc;          // MFMAs
            const double fl = mf * 2048.0;
            printf("waves/SIMD %d: %.3f ms  %.1f TF/s  (%.1f ns per MFMA per SIMD)\n", wps, ms, fl / ms / 1e9, ms * 1e6 / ((double)iters * nacc * wps));
        }
    }
    // one CU only: per-MFMA cycles without chip-wide power effects
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak<8>, dim3(1), dim3(256), 0, 0, 20000, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("one CU, 1 wave/SIMD: %.1f ns per MFMA\n", ms * 1e6 / (20000.0 * 8));
    return 0;
}
