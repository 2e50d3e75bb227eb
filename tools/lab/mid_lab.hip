// k_mid_factor (one-launch LDL' of a mid-size dense SPD matrix + forward solve) followed by the chained backward solve, against a host
// solve; timing of the pair back to back.  usage: mid_lab.bin [n ...]
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I include -I qpdo_amd/csrc
//        tools/lab/mid_lab.hip -o tools/lab/mid_lab.bin -L/opt/rocm/lib -lrccl
#include "../../qpdo_amd/csrc/qpdo_dev.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
static double frand(unsigned long long &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((double)(s >> 11) / 9007199254740992.0) * 2.0 - 1.0; }
static int run(int n) {
    const int ld = (n + 63) / 64 * 64, nb = ld / 64;
    if (nb > MID_MAX_NB) { printf("n = %d: too large\n", n); return 0; }
    unsigned long long seed = 12345 + n;
    // K = B B' / n + (1 + i/n) I with B n x 32 random: SPD, full, moderately conditioned; identity padding
    std::vector<double> B((size_t)n * 32), K((size_t)ld * ld, 0.0), b(ld, 0.0);
    for (auto &v : B) v = frand(seed);
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k < 32; k++) s += B[(size_t)i * 32 + k] * B[(size_t)j * 32 + k]; K[i + (size_t)j * ld] = K[j + (size_t)i * ld] = s / 8.0 + (i == j ? 1.0 + (double)i / n : 0.0); }
    for (int i = n; i < ld; i++) K[i + (size_t)i * ld] = 1.0;
    for (int i = 0; i < n; i++) b[i] = frand(seed);
    // host solve (Cholesky-free LDL' in place on a copy, lower)
    const bool host_ref = n <= 3000;
    std::vector<double> A(host_ref ? K : std::vector<double>(1)), x(b);
    if (host_ref) {
    for (int k = 0; k < ld; k++) { const double d = A[k + (size_t)k * ld]; for (int i = k + 1; i < ld; i++) A[i + (size_t)k * ld] /= d;
        for (int j = k + 1; j < ld; j++) { const double t = A[j + (size_t)k * ld] * d; if (t != 0.0) for (int i = j; i < ld; i++) A[i + (size_t)j * ld] -= A[i + (size_t)k * ld] * t; } }
    for (int k = 0; k < ld; k++) for (int i = k + 1; i < ld; i++) x[i] -= A[i + (size_t)k * ld] * x[k];
    for (int k = 0; k < ld; k++) x[k] /= A[k + (size_t)k * ld];
    for (int k = ld - 1; k >= 0; k--) for (int i = k + 1; i < ld; i++) x[k] -= A[i + (size_t)k * ld] * x[i];
    }
    double *dK, *dK0, *Dg, *Li, *LiT, *rhs, *z, *y, *xs, *Cp, *Dinv; unsigned int *flags; Ctrl *ctrl;
    (void)hipMalloc(&dK, (size_t)ld * ld * 8); (void)hipMalloc(&dK0, (size_t)ld * ld * 8); (void)hipMalloc(&Dg, ld * 8); (void)hipMalloc(&Li, (size_t)nb * 4096 * 8); (void)hipMalloc(&LiT, (size_t)nb * 4096 * 8);
    (void)hipMalloc(&Cp, (size_t)nb * 4096 * 8); (void)hipMalloc(&Dinv, ld * 8); (void)hipMalloc(&rhs, ld * 8); (void)hipMalloc(&z, ld * 8); (void)hipMalloc(&y, ld * 8); (void)hipMalloc(&xs, ld * 8); (void)hipMalloc(&flags, (size_t)(nb + 1) * nb * 4); (void)hipMalloc(&ctrl, sizeof(Ctrl));
    (void)hipMemset(flags, 0, (size_t)(nb + 1) * nb * 4); (void)hipMemset(ctrl, 0, sizeof(Ctrl));
    (void)hipMemcpy(dK0, K.data(), (size_t)ld * ld * 8, hipMemcpyHostToDevice); (void)hipMemcpy(rhs, b.data(), ld * 8, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mid_factor), hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS_DOUBLES * 8);
    const int grid = nb * (nb + 1) / 2 + nb;
    unsigned int epoch = 0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms_tot = 0.f; const int reps = n > 3000 ? 5 : 20;
    for (int rep = 0; rep < reps + 2; rep++) {
        (void)hipMemcpy(dK, dK0, (size_t)ld * ld * 8, hipMemcpyDeviceToDevice);
        hipLaunchKernelGGL(k_fill_sentinel, dim3(4), dim3(256), 0, 0, ld, z, xs);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_mid_factor, dim3(grid), dim3(256), MID_LDS_DOUBLES * 8, 0, dK, n, ld, nb, Dg, Li, LiT, (const double *)rhs, z, y, flags, ++epoch, ctrl, Cp, Dinv);
        hipLaunchKernelGGL(k_ldl_chain<false>, dim3(1, nb), dim3(256), 0, 0, (const double *)dK, ld, nb, (const double *)LiT, (const double *)Dg, (const double *)y, xs, (double *)nullptr, 0, ctrl);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) ms_tot += ms;
    }
    // the backward chain alone (sentinel refill + chain per repetition; the refill is a ~2 us launch)
    float ms_chain = 0.f;
    {
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < reps; rep++) {
            hipLaunchKernelGGL(k_fill_sentinel, dim3(4), dim3(256), 0, 0, ld, xs, xs);
            hipLaunchKernelGGL(k_ldl_chain<false>, dim3(1, nb), dim3(256), 0, 0, (const double *)dK, ld, nb, (const double *)LiT, (const double *)Dg, (const double *)y, xs, (double *)nullptr, 0, ctrl);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms_chain, e0, e1);
    }
    hipError_t err = hipDeviceSynchronize();
#ifdef MID_TIMING
    if (nb <= 5) {
        static long long t[512][16];
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_mid_t), sizeof(t));
        long long t0 = t[0][0];
        int w = 0;
        for (int j = 0; j < nb; j++) for (int i = j; i <= nb; i++, w++) {
            auto us = [&](int k) { return (t[w][k] - t0) * 0.01; };
            if (i == nb) continue;
            if (j) printf("     (%d,%d) last k: flag %.2f loads landed %.2f barrier %.2f staged %.2f\n", i, j, us(1), us(13), us(14), us(2));
            if (i == j) printf("  diag (%d,%d): start %.2f | last k: flag %.2f staged %.2f mfma %.2f | T ready %.2f factored %.2f published %.2f\n", i, j, us(0), j ? us(1) : 0.0, j ? us(2) : 0.0, j ? us(3) : 0.0, us(4), us(5), us(6));
            else printf("  tile (%d,%d): start %.2f | last k: flag %.2f staged %.2f mfma %.2f | C ready %.2f diag flag %.2f Linv staged %.2f mfma %.2f stores issued %.2f published %.2f\n", i, j, us(0), j ? us(1) : 0.0, j ? us(2) : 0.0, j ? us(3) : 0.0, us(7), us(8), us(9), us(10), us(11), us(12));
        }
    }
#endif
    std::vector<double> xd(ld); Ctrl hc;
    (void)hipMemcpy(xd.data(), xs, ld * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hc, ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost);
    double e = 0, nx = 0; for (int i = 0; i < n; i++) { e = fmax(e, fabs(xd[i] - x[i])); nx = fmax(nx, fabs(x[i])); }
    if (!host_ref) e = -1.0;
    // residual of the device solution
    double rmax = 0; for (int i = 0; i < n; i++) { double s = -b[i]; for (int j = 0; j < n; j++) s += K[i + (size_t)j * ld] * xd[j]; rmax = fmax(rmax, fabs(s)); }
    printf("n = %4d (nb = %2d, %3d workgroups): factor + forward + backward %.1f us (refill + backward chain alone %.1f us)   max |x - x_host| = %.2e (|x| %.2e)   |K x - b| = %.2e   chain_err %d   %s\n",
           n, nb, grid, ms_tot * 1e3 / reps, ms_chain * 1e3 / reps, e, nx, rmax, hc.cnt[C_CHAIN_ERR], hipGetErrorString(err));
    (void)hipFree(dK); (void)hipFree(dK0); (void)hipFree(Dg); (void)hipFree(Li); (void)hipFree(LiT); (void)hipFree(rhs); (void)hipFree(z); (void)hipFree(y); (void)hipFree(xs); (void)hipFree(flags); (void)hipFree(ctrl);
    return 0;
}
int main(int argc, char **argv) {
    if (argc > 1) { for (int a = 1; a < argc; a++) run(atoi(argv[a])); return 0; }
    const int ns[] = {64, 130, 200, 300, 500, 1000, 1344};
    for (int n : ns) run(n);
    return 0;
}
