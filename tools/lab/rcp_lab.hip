// accuracy of v_rcp_f64 alone and with one / two Newton steps against IEEE division (random doubles in [1e-3, 1e3])
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *e, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const double d = x[i], t = 1.0 / d;
    double r0 = __builtin_amdgcn_rcp(d);
    double r1 = fma(fma(-d, r0, 1.0), r0, r0);
    double r2 = fma(fma(-d, r1, 1.0), r1, r1);
    e[3 * i] = fabs(r0 - t) / t; e[3 * i + 1] = fabs(r1 - t) / t; e[3 * i + 2] = fabs(r2 - t) / t;
}
int main() {
    const int n = 1 << 20; double *x, *e; (void)hipMallocManaged(&x, n * 8); (void)hipMallocManaged(&e, 3 * n * 8);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (double)(s >> 11) / 9007199254740992.0; x[i] = pow(10.0, 6.0 * u - 3.0) * ((s & 1) ? 1 : -1); }
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, e, n); (void)hipDeviceSynchronize();
    double m[3] = {0, 0, 0}; for (int i = 0; i < n; i++) for (int q = 0; q < 3; q++) m[q] = fmax(m[q], e[3 * i + q]);
    printf("max relative error: rcp %.3e   +1 Newton %.3e   +2 Newton %.3e   (eps = 1.1e-16)\n", m[0], m[1], m[2]);
    return 0;
}
