// qpdo_dev.hip -- HIP backend for gfx950 (MI355X, CDNA4): every per-iteration
// operation of the reference solver's hot path as hand-written kernels.
//
// State is resident in HBM for the lifetime of the workspace; the host driver
// (qpdo_api.c) only sees a 200-byte control block per pass.  Matrices are held
// gather-style: CSR(A) for A*x, CSR(A') (= the caller's CSC of A) for A'*y and a
// full symmetric CSR(Q) for Q*x, so no product needs atomics and every result is
// reproducible run to run.  Elementwise arithmetic keeps the reference's
// operation order and the file is compiled with -ffp-contract=off, so vector
// kernels are bit-identical to the CPU oracle; only reductions (dot products,
// row sums) differ in summation order.
//
// Wave = 64 lanes throughout.  No CUDA compatibility paths.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>
#include <thread>

#include <rccl/rccl.h>

#include "qpdo_dev.h"
#include "pass_decision.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define QPDO_INFTY_D 1e20
static const int BLK = 256;        // threads per block: 4 waves
static const int PGRID = 1024;     // max blocks of any kernel that emits per-block partial sums
static const u64 KEY_SENTINEL = 0xFFFFFFFFFFFFFFFFull;

static thread_local char g_err[256] = "";
static int set_err(hipError_t e, const char *what, int line) {
    snprintf(g_err, sizeof(g_err), "%s (line %d): %s", what, line, hipGetErrorString(e));
    return (int)e ? (int)e : -1;
}
#define HIPCHK(call)                                              \
    do {                                                          \
        hipError_t e__ = (call);                                  \
        if (e__ != hipSuccess) return set_err(e__, #call, __LINE__); \
    } while (0)

#include "dev/state.inc"
#include "dev/helpers.inc"
#include "dev/spmv.inc"
#include "dev/vector.inc"
#include "dev/pcg_kernels.inc"
#include "dev/linesearch.inc"
#include "dev/dense_kernels.inc"
#include "dev/mid_kernels.inc"
#include "dev/transpose.inc"
#include "dev/band.inc"
#include "dev/host_core.inc"

extern "C" {
#include "dev/host_setup.inc"
#include "dev/host_pcg.inc"
#include "dev/host_dense.inc"
#include "dev/host_band.inc"
#include "dev/host_step.inc"
}  // extern "C"
