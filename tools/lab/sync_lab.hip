// sync_lab.hip -- how long does one "device decides, host looks, host launches again" round trip take?  (lab, not product)
// The multi-kernel route reads its control block back once per pass (dev/host_core.inc read_ctrl: hipMemcpyAsync D2H into pinned memory +
// hipStreamSynchronize); at C1 size that is ~35 read-backs in a 3.6 ms solve.  Variants:
//   A  kernel -> hipMemcpyAsync(pinned <- device, 512 B) -> hipStreamSynchronize                      (what read_ctrl does)
//   B  kernel -> second one-wave kernel that copies the block into pinned memory and bumps a sequence word there -> host spins on it
//   C  the kernel itself writes the block + sequence word into pinned memory -> host spins
//   D  kernel -> hipEventRecord + hipEventSynchronize, block read from pinned memory written by the kernel (no copy engine / copy kernel)
// Each loop: N round trips, every one followed by the launch of the next kernel; wall per round trip.
// build: hipcc -O3 --offload-arch=gfx950 -o sync_lab.bin sync_lab.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Blk { unsigned long long w[63]; unsigned long long seq; };
__global__ void k_work(Blk *dev, unsigned long long it) {          // stands for the last kernel of a pass: fills the device block
    if (threadIdx.x < 63) dev->w[threadIdx.x] = it * 64 + threadIdx.x;
}
__global__ void k_pub(const Blk *dev, Blk *host, unsigned long long it) {
    if (threadIdx.x < 63) host->w[threadIdx.x] = dev->w[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store(&host->seq, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
__global__ void k_work_pub(Blk *dev, Blk *host, unsigned long long it) {
    if (threadIdx.x < 63) { dev->w[threadIdx.x] = it * 64 + threadIdx.x; host->w[threadIdx.x] = it * 64 + threadIdx.x; }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store(&host->seq, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2000;
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Blk *dev, *host; CHK(hipMalloc(&dev, sizeof(Blk))); CHK(hipHostMalloc(&host, sizeof(Blk), hipHostMallocDefault));
    memset(host, 0, sizeof(Blk));
    hipEvent_t ev; CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    volatile unsigned long long *seq = &host->seq;
    unsigned long long it = 0, bad = 0;
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now_us();
        for (int i = 0; i < N; i++) {
            it++;
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, dev, it);
            CHK(hipMemcpyAsync(host, dev, sizeof(Blk) - 8, hipMemcpyDeviceToHost, s));
            CHK(hipStreamSynchronize(s));
            if (host->w[5] != it * 64 + 5) bad++;
        }
        double tA = (now_us() - t0) / N;
        t0 = now_us();
        for (int i = 0; i < N; i++) {
            it++;
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, dev, it);
            hipLaunchKernelGGL(k_pub, dim3(1), dim3(64), 0, s, (const Blk *)dev, host, it);
            long spins = 0;
            while (*seq != it) { if (++spins > 2000000000L) { printf("B: stuck\n"); return 2; } }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if (host->w[5] != it * 64 + 5) bad++;
        }
        double tB = (now_us() - t0) / N;
        t0 = now_us();
        for (int i = 0; i < N; i++) {
            it++;
            hipLaunchKernelGGL(k_work_pub, dim3(1), dim3(64), 0, s, dev, host, it);
            long spins = 0;
            while (*seq != it) { if (++spins > 2000000000L) { printf("C: stuck\n"); return 2; } }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if (host->w[5] != it * 64 + 5) bad++;
        }
        double tC = (now_us() - t0) / N;
        t0 = now_us();
        for (int i = 0; i < N; i++) {
            it++;
            hipLaunchKernelGGL(k_work_pub, dim3(1), dim3(64), 0, s, dev, host, it);
            CHK(hipEventRecord(ev, s));
            CHK(hipEventSynchronize(ev));
            if (host->w[5] != it * 64 + 5) bad++;
        }
        double tD = (now_us() - t0) / N;
        // and for scale: the same kernel launched back to back without looking (launch throughput)
        t0 = now_us();
        for (int i = 0; i < N; i++) { it++; hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, dev, it); }
        CHK(hipStreamSynchronize(s));
        double tE = (now_us() - t0) / N;
        printf("rep %d: per round trip  A memcpy+streamsync %.2f us | B pub kernel + host spin %.2f us | C fused pub + host spin %.2f us | D fused pub + event sync %.2f us | (launch only %.2f us)  mismatches %llu\n",
               rep, tA, tB, tC, tD, tE, bad);
    }
    return bad ? 3 : 0;
}
