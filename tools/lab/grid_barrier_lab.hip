// Cost of a grid-wide barrier inside a persistent kernel on MI355X (round 4: would a persistent inner-CG kernel beat three launches per
// iteration at cache-resident sizes?).  Variants: one device-scope counter; one counter per XCD (blockIdx % 8) + a top-level counter.
// Every spin is bounded: a barrier that does not complete sets a flag and every block leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct Bar { unsigned cnt[64]; unsigned xcd[8][16]; unsigned top[16]; unsigned gen[16]; int fail; };
__device__ __forceinline__ bool spin_until(volatile unsigned *p, unsigned target, int *fail) {
    for (long i = 0; i < (1L << 22); i++) { if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true; if (*(volatile int *)fail) return false; __builtin_amdgcn_s_sleep(1); }
    *(volatile int *)fail = 1; return false;
}
// flat: every block adds 1 to one counter and waits for it to reach (round+1)*nblocks
__device__ bool barrier_flat(Bar *b, unsigned round) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(&b->cnt[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        ok = spin_until(&b->cnt[0], (round + 1) * gridDim.x, &b->fail);
        __threadfence();
    }
    __syncthreads();
    return ok;
}
// two-level: blocks of one "group" (blockIdx % 8: the XCD a block is dispatched to, round-robin) count on their own line; the last arriver
// of a group bumps the top counter; everyone waits on the top counter
__device__ bool barrier_2lvl(Bar *b, unsigned round) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = blockIdx.x & 7, per = (gridDim.x + 7 - g) / 8;
        const unsigned old = __hip_atomic_fetch_add(&b->xcd[g][0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == (round + 1) * per) __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned ngroups = gridDim.x < 8 ? gridDim.x : 8;
        ok = spin_until(&b->top[0], (round + 1) * ngroups, &b->fail);
        __threadfence();
    }
    __syncthreads();
    return ok;
}
template <int V>
__global__ void k_bar(Bar *b, int rounds, double *work, int wn) {
    double acc = 0.0;
    for (int r = 0; r < rounds; r++) {
        // a little work between barriers: each block touches its slice (so that the barrier really orders memory traffic)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < wn; i += gridDim.x * blockDim.x) acc += work[i];
        const bool ok = V == 0 ? barrier_flat(b, (unsigned)r) : barrier_2lvl(b, (unsigned)r);
        if (!ok) break;
    }
    if (acc == 12345.678) work[0] = acc;
}
template <int V>
static void run(const char *name, int grid, int threads, int rounds, int wn, Bar *b, double *work) {
    CK(hipMemset(b, 0, sizeof(Bar)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_bar<V>), dim3(grid), dim3(threads), 0, 0, b, 10, work, wn); CK(hipDeviceSynchronize());
    CK(hipMemset(b, 0, sizeof(Bar)));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_bar<V>), dim3(grid), dim3(threads), 0, 0, b, rounds, work, wn);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Bar hb; CK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost));
    printf("%-28s grid %4d x %4d, %d rounds, work %7d doubles: %.2f us per round%s\n", name, grid, threads, rounds, wn, ms * 1e3 / rounds, hb.fail ? "  ** BARRIER TIMED OUT **" : ""); fflush(stdout);
}
int main() {
    Bar *b; double *work; CK(hipMalloc(&b, sizeof(Bar))); CK(hipMalloc(&work, 8 << 20)); CK(hipMemset(work, 0, 8 << 20));
    for (int grid : {64, 128, 256}) {
        run<0>("flat counter", grid, 256, 2000, 0, b, work);
        run<1>("per-XCD + top counter", grid, 256, 2000, 0, b, work);
        run<0>("flat counter", grid, 1024, 2000, 0, b, work);
        run<1>("per-XCD + top counter", grid, 1024, 2000, 0, b, work);
        run<1>("per-XCD + top, 1M doubles", grid, 1024, 2000, 1 << 20, b, work);
    }
    return 0;
}
