// Phase times of k_ldl_diag_blocked on one 64x64 SPD block (wall_clock64, 10 ns ticks) and its average launch duration.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DDG_TIMING -I include -I qpdo_amd/csrc
//        tools/lab/diag_lab.hip -o tools/lab/diag_lab.bin -L/opt/rocm/lib -lrccl
#include "../../qpdo_amd/csrc/qpdo_dev.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
int main() {
    const int ld = 64;
    std::vector<double> A(64 * 64);
    for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) A[i + j * 64] = (i == j ? 80.0 : 0.0) + std::cos(0.37 * (i + 1) * (j + 1) + 0.11 * (i + j));
    for (int i = 0; i < 64; i++) for (int j = 0; j < i; j++) A[j + i * 64] = A[i + j * 64];
    double *K, *Dg, *Li, *LiT;
    (void)hipMalloc(&K, 64 * 64 * 8); (void)hipMalloc(&Dg, 64 * 8); (void)hipMalloc(&Li, 64 * 64 * 8); (void)hipMalloc(&LiT, 64 * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipMemcpy(K, A.data(), 64 * 64 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_ldl_diag_blocked, dim3(1), dim3(256), 0, 0, K, ld, 0, Dg, Li, LiT);
        (void)hipDeviceSynchronize();
    }
    long long t[32];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_dg_t), sizeof(t));
    const char *nm[16] = {"load issued", "load done", "sub0", "pan0", "upd0", "sub1", "pan1", "upd1", "sub2", "pan2", "upd2", "sub3", "-", "-", "inverse", "stores issued"};
    for (int i = 1; i < 16; i++) if (nm[i][0] != '-') { int pr = i - 1; while (nm[pr][0] == '-') pr--; printf("%-14s %+7.2f us\n", nm[i], (t[i] - t[pr]) * 0.01); }
    printf("total in kernel %.2f us\n", (t[15] - t[0]) * 0.01);
    // check: L D L' = A
    std::vector<double> L(64 * 64), D(64);
    (void)hipMemcpy(L.data(), K, 64 * 64 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(D.data(), Dg, 64 * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 64; i++) for (int j = 0; j <= i; j++) {
        double s = 0; for (int k = 0; k <= j; k++) s += (k == i ? 1.0 : L[i + k * 64]) * D[k] * (k == j ? 1.0 : L[j + k * 64]);
        err = fmax(err, fabs(s - A[i + j * 64]));
    }
    printf("max |L D L' - A| = %.3e\n", err);
    { std::vector<double> Lv(64 * 64), Iv(64 * 64); (void)hipMemcpy(Iv.data(), Li, 64 * 64 * 8, hipMemcpyDeviceToHost);
      unsigned long long h = 1469598103934665603ull; auto mix = [&](const double *p, int n) { for (int i = 0; i < n; i++) { unsigned long long b; memcpy(&b, p + i, 8); h = (h ^ b) * 1099511628211ull; } };
      for (int i = 0; i < 64; i++) for (int j = 0; j < i; j++) mix(&L[i + j * 64], 1);
      mix(D.data(), 64); mix(Iv.data(), 64 * 64); printf("hash of L, D, L^-1: %016llx\n", h); }
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 200; rep++) hipLaunchKernelGGL(k_ldl_diag_blocked, dim3(1), dim3(256), 0, 0, K, ld, 0, Dg, Li, LiT);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("back-to-back launches: %.2f us each\n", ms * 1e3 / 200);
    return 0;
}
