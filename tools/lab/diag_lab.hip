// The two diagonal-block kernels on one 64x64 SPD block: phase times (wall_clock64, 10 ns ticks; -DDG_TIMING), accuracy of
// L D L' = A and of L^-1, average launch duration back to back.  VER 1: columns published through LDS (rounds 2-4); VER 2: 64 x 16
// sub-panels eliminated in registers with v_readlane broadcasts (round 5).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DDG_TIMING -I include -I qpdo_amd/csrc
//        tools/lab/diag_lab.hip -o tools/lab/diag_lab.bin -L/opt/rocm/lib -lrccl
#include "../../qpdo_amd/csrc/qpdo_dev.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
template <int VER>
static void run(const char *name, const std::vector<double> &A, double *K, double *Dg, double *Li, double *LiT) {
    const int ld = 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipMemcpy(K, A.data(), 64 * 64 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_ldl_diag_blocked_v<VER>, dim3(1), dim3(256), 0, 0, K, ld, 0, Dg, Li, LiT);
        (void)hipDeviceSynchronize();
    }
    printf("== %s\n", name);
#ifdef DG_TIMING
    long long t[32];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_dg_t), sizeof(t));
    printf("load %.2f us, factor + inverse %.2f us, stores %.2f us, total in kernel %.2f us\n", (t[1] - t[0]) * 0.01, (t[14] - t[1]) * 0.01, (t[15] - t[14]) * 0.01, (t[15] - t[0]) * 0.01);
    if (VER == 2) {
        long long pr = t[1];
        for (int s = 0; s < 4; s++) { printf("  sub-panel %d: wave 0 done %+.2f us, barrier passed %+.2f us", s, (t[16 + 2 * s] - pr) * 0.01, (t[17 + 2 * s] - pr) * 0.01); pr = t[17 + 2 * s];
            if (s < 3) { printf(", update %+.2f us\n", (t[4 + 3 * s] - pr) * 0.01); pr = t[4 + 3 * s]; } else printf("\n"); }
        printf("  inverse tail: %+.2f us, %+.2f us\n", (t[24] - pr) * 0.01, (t[25] - t[24]) * 0.01);
    }
#endif
    std::vector<double> L(64 * 64), D(64), Iv(64 * 64), It(64 * 64);
    (void)hipMemcpy(L.data(), K, 64 * 64 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(D.data(), Dg, 64 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(Iv.data(), Li, 64 * 64 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(It.data(), LiT, 64 * 64 * 8, hipMemcpyDeviceToHost);
    double err = 0, erri = 0, errt = 0;
    for (int i = 0; i < 64; i++) for (int j = 0; j <= i; j++) {
        double s = 0; for (int k = 0; k <= j; k++) s += (k == i ? 1.0 : L[i + k * 64]) * D[k] * (k == j ? 1.0 : L[j + k * 64]);
        err = fmax(err, fabs(s - A[i + j * 64]));
    }
    // Li[c*64 + r] = (L^-1)[r][c]: L * L^-1 = I
    for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) {
        double s = 0; for (int k = 0; k < 64; k++) { const double lik = (k == i) ? 1.0 : (k < i ? L[i + k * 64] : 0.0); s += lik * Iv[j * 64 + k]; }
        erri = fmax(erri, fabs(s - (i == j ? 1.0 : 0.0)));
        errt = fmax(errt, fabs(It[i * 64 + j] - Iv[j * 64 + i]));       // LiT[r*64 + c] = (L^-1)[r][c]
    }
    printf("max |L D L' - A| = %.3e   max |L L^-1 - I| = %.3e   max |LinvT - Linv'| = %.3e\n", err, erri, errt);
    (void)hipMemcpy(K, A.data(), 64 * 64 * 8, hipMemcpyHostToDevice);
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 200; rep++) hipLaunchKernelGGL(k_ldl_diag_blocked_v<VER>, dim3(1), dim3(256), 0, 0, K, ld, 0, Dg, Li, LiT);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("back-to-back launches: %.2f us each\n", ms * 1e3 / 200);
}
int main() {
    std::vector<double> A(64 * 64);
    for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) A[i + j * 64] = (i == j ? 80.0 : 0.0) + std::cos(0.37 * (i + 1) * (j + 1) + 0.11 * (i + j));
    for (int i = 0; i < 64; i++) for (int j = 0; j < i; j++) A[j + i * 64] = A[i + j * 64];
    double *K, *Dg, *Li, *LiT;
    (void)hipMalloc(&K, 64 * 64 * 8); (void)hipMalloc(&Dg, 64 * 8); (void)hipMalloc(&Li, 64 * 64 * 8); (void)hipMalloc(&LiT, 64 * 64 * 8);
    run<1>("VER 1 (LDS column publication)", A, K, Dg, Li, LiT);
    run<2>("VER 2 (register sub-panels, readlane broadcasts)", A, K, Dg, Li, LiT);
    return 0;
}
