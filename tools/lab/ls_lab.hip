// ls_lab.hip -- phase times of k_ls_small (the one-launch linesearch of small problems) on random candidates.  usage: ls_lab.bin [2m ...]
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DLS_TIMING -I include -I qpdo_amd/csrc
//        tools/lab/ls_lab.hip -o tools/lab/ls_lab.bin -L/opt/rocm/lib -lrccl
#include "../../qpdo_amd/csrc/qpdo_dev.hip"
#include <cstdio>
#include <vector>
static double frand(unsigned long long &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((double)(s >> 11) / 9007199254740992.0) * 2.0 - 1.0; }
int main(int argc, char **argv) {
    std::vector<int> sizes; for (int i = 1; i < argc; i++) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {200, 1200, 2000, 4000};
    for (int M2 : sizes) {
        unsigned long long seed = 99 + M2;
        std::vector<double> dl(M2), al(M2);
        for (int i = 0; i < M2; i++) { dl[i] = frand(seed); al[i] = frand(seed); }
        double *ddl, *dal, *part; u64 *key; u32 *idx; Ctrl *ctrl;
        (void)hipMalloc(&ddl, M2 * 8); (void)hipMalloc(&dal, M2 * 8); (void)hipMalloc(&part, (size_t)P_COUNT * PGRID * 8); (void)hipMalloc(&key, M2 * 8); (void)hipMalloc(&idx, M2 * 4); (void)hipMalloc(&ctrl, sizeof(Ctrl));
        (void)hipMemcpy(ddl, dl.data(), M2 * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dal, al.data(), M2 * 8, hipMemcpyHostToDevice);
        (void)hipMemset(part, 0, (size_t)P_COUNT * PGRID * 8);
        const int g = (M2 + 255) / 256;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0.f;
        for (int rep = 0; rep < 5; rep++) {
            (void)hipMemset(ctrl, 0, sizeof(Ctrl));
            hipLaunchKernelGGL(k_ls_prep_raw, dim3(g), dim3(256), 0, 0, M2, (const double *)ddl, (const double *)dal, key, idx, part + P_A0 * PGRID, part + P_B0 * PGRID, ctrl);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_ls_small, dim3(1), dim3(1024), 0, 0, ctrl, (const double *)part, g, 1, (const u64 *)key, (const double *)ddl, (const double *)dal, M2);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        Ctrl h; (void)hipMemcpy(&h, ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost);
        long long t[16]; (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ls_t), sizeof(t));
        printf("2m = %5d: nL %d, launch %.1f us | loads + compaction %.2f | sort %.2f | six sums %.2f | items %.2f | sub-block scan %.2f | search %.2f | (in kernel %.2f us)  tau %.6g\n",
               M2, h.cnt[C_NL], ms * 1e3, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01, (t[6] - t[0]) * 0.01, h.val[V_TAU]);
    }
    return 0;
}
