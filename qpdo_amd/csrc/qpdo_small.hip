// qpdo_small.hip -- fused solver for batches of small QPs on gfx950: ONE workgroup runs the complete
// qpdo_setup (Ruiz scaling) + qpdo_warm_start + qpdo_solve loop of ONE problem in a single kernel launch
// (BASELINE.json configs[2]: thousands of MPC-sized QPs, no collective).  A batch is a grid of such
// workgroups; nothing returns to the host between passes.
//
// The arithmetic follows the reference's operation order step by step (the same restatement the CPU
// oracle uses): row sums in ascending column order, 4-way grouped dot products (src/lin_alg.c:59-71),
// natural-order left-looking LDL' of Q + sigma_f I + A' diag(d) A, column-oriented triangular solves,
// stable sort of the linesearch breakpoints followed by the sequential walk of src/linesearch.c:126-157.
// With -ffp-contract=off the results are therefore bit-identical to the oracle, not merely close.
//
// Work split inside the workgroup: elementwise passes and row sums are thread-parallel, reductions whose
// order matters (dots, the breakpoint walk) run on one lane, the factorization is parallel over rows with
// one barrier per column.  All vectors and the dense K (n x n) live in global memory (L2-resident at these
// sizes); LDS holds the sort keys and reduction scratch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <vector>
#include <thread>
#include <mutex>
#include <memory>

#include "qpdo.h"
#include "qpdo_amd_ext.h"
#include "qpdo_dev.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define SM_THREADS 512
#define SM_MAX_M 1024                    // 2m breakpoints are sorted in LDS
#define SM_MAX_N 1024
#define SM_INFTY 1e20
#define SM_MAX_RANK_UPDATE 100

struct SmallQP {
    int n, m;
    const int *Arp, *Aci; double *Aval;      // CSR(A), m rows, column-sorted
    const int *Trp, *Tci; double *Tval;      // CSR(A'), n rows
    const int *Qrp, *Qci; double *Qval;      // full symmetric CSR(Q)
    double *q, *l, *u;                       // scaled in place
    const double *x0, *y0;                   // warm start or NULL
    double *nv;                              // n-vector workspace (NV_COUNT * n)
    double *mv;                              // m-vector workspace (MV_COUNT * m)
    double *lsv;                             // 2 * (2m): ls_delta, ls_alpha
    int *iv;                                 // 3 * m ints: active, active_old, changed
    int *tpos;                               // nnz(A) ints: for entry q of CSR(A') (column j, row r) the slot of column j in row r of CSR(A)
    double *K;                               // n * n, column-major, lower
    double c_const;
    // results
    double *sol_x, *sol_y, *cert_dx, *cert_dy;
    QPDOInfo info;
    long newton_passes, factor_count;
    long long *prof;                         // optional: per-phase wall-clock ticks (diagnostic runs only)
    struct SmallRes *res;                    // resident mode (latency kernel only): see SmallRes; NULL in a batch
    unsigned batch_vec_off;                  // a batch through the latency kernel: byte offset of the vector workspace inside the dynamic LDS (0: global memory)
};
// ---- resident mode (qpdo_solve of ONE small workspace through the latency kernel, qdev_small_resident_*): the matrices, q, l, u are the
// workspace's own device arrays, already scaled by qpdo_setup (bit-identical to the oracle's scaling), so the kernel skips its Ruiz
// phase and takes D, E, c from the workspace; x0 / y0 are the workspace's warm-started x, y (already scaled by qpdo_warm_start).
struct SmallRes {
    const double *rD, *rDinv, *rE, *rEinv; double r_c, r_cinv;
    // mode 0: the whole of warm start (from zero, or from x0 / y0) + solve in one launch -- qpdo_solve with its AUTOMATIC warm start
    //         (qpdo.c:312-314), where nothing can come between the two.
    // mode 1: an explicit qpdo_warm_start alone: x, x_bar, Qx, Ax, y, y_bar, A'y, mu, 1/sqrt(mu) in the oracle's operation order into the
    //         workspace's vectors (st_*), the objective of qpdo.c:257 into ws_objective; the caller may then qpdo_update_q / _bounds --
    //         those act on the workspace's vectors exactly as in the reference (qpdo.c:522-586) -- before
    // mode 2: the solve loop from the workspace's state (st_* are read instead of a warm start).
    int mode;
    double *state_x, *state_Qx;              // modes 0, 2 out: final x and Qx into the workspace's vectors (qpdo_update_q reads them, qpdo.c:549-586); = st_x, st_Qx
    double *st_xbar, *st_Ax, *st_y, *st_ybar, *st_Aty, *st_mu, *st_isq;
    double ws_objective;
    unsigned lds_vec_off;                    // byte offset inside the dynamic LDS of the item's vector workspace (nv, mv, iv), 0: it stays in global memory
    double *out_x, *out_y;                   // out: the internal (scaled) iterates, the host mirrors work->x / work->y (y after termination.c:85)
    QPDOAmdTraceRec *trace; long trace_cap, ntrace;      // optional per-pass trace (pinned host memory), records written / capacity
    double sigma_end, tau_end;
};
enum { NV_X = 0, NV_XBAR, NV_QX, NV_ATY, NV_DF, NV_RD, NV_RDI, NV_RHS, NV_DX, NV_QDX, NV_ATDY, NV_D, NV_DINV, NV_T, NV_COUNT };
enum { MV_Y = 0, MV_YBAR, MV_AX, MV_MU, MV_ISQ, MV_W, MV_RP, MV_RPOLD, MV_RPI, MV_DY, MV_ADX, MV_DW, MV_E, MV_EINV, MV_ATS, MV_T, MV_DWF, MV_COUNT };

// diagnostic phase timer: lane 0 adds wall-clock ticks (100 MHz) to a buffer no other code reads
#define PH(k) do { if (P.prof && threadIdx.x == 0) { const long long t_ = wall_clock64(); P.prof[k] += t_ - tph; tph = t_; } } while (0)
enum { PH_RESID = 0, PH_OUTER, PH_PREP, PH_ASM, PH_FACTOR, PH_SOLVE, PH_SPMV, PH_LS, PH_UPDATE, PH_LS_DOTS, PH_LS_SORT, PH_LS_JSUM, PH_LS_WALK, PH_CYC /* shader cycles of the solve loop (s_memtime) */, PH_WALL /* its 100 MHz ticks */, PH_COUNT };   // (the last four: inside PH_LS)
#define FOR_T(i, N) for (int i = threadIdx.x; i < (N); i += blockDim.x)
#define SYNC __syncthreads()

__device__ __forceinline__ double s_abs(double x) { return x < 0 ? -x : x; }
__device__ __forceinline__ double s_max(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double s_min(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double s_mid(double a, double lo, double hi) { return s_max(lo, s_min(a, hi)); }

// Sequential left fold  acc = (..((acc + v[0]) + v[1]) + ..) + v[count-1]  -- the additions of a one-lane loop in index order, hence the
// same bits -- executed by wave 0: its 64 lanes LOAD 64 elements at a time and the fold walks them with v_readlane (register to
// register, ~16 cycles per element) instead of paying a dependent LDS round trip per element on one lane (measured on the batch
// kernel's slowest items: 84 us -> 7 us for the 2m-element sums of a linesearch).  Call from all 64 lanes of wave 0.
typedef __attribute__((address_space(3))) double lds_f64;      // a pointer KNOWN to address the workgroup's LDS: ds_read / ds_write, not flat
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ double rl64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// (+0.0) + v[0] + v[1] + ... in index order on one lane (or on every lane alike; lanes may differ in v and count): the next eight
// elements are loaded while eight are added, so an addition waits for the previous addition only -- ~10 cycles per element against ~30
// for the readlane fold below and ~100 for a loop that loads what it adds.  The last count % 8 elements are loaded together and padded
// with +0.0, which never changes a sum that started at +0.0 (such a sum is never -0).
__device__ __forceinline__ double seq_fold(const lds_f64 *v, int count) {
    double acc = 0.0;
    int g = 0;
    if (count >= 8) {
        double a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7];
#pragma nounroll
        for (g = 8; g + 8 <= count; g += 8) {
            const double b0 = v[g], b1 = v[g + 1], b2 = v[g + 2], b3 = v[g + 3], b4 = v[g + 4], b5 = v[g + 5], b6 = v[g + 6], b7 = v[g + 7];
            acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        }
        acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
    }
    const int rem = count - g;
    if (rem > 0) {
        double t[7];
#pragma unroll
        for (int u = 0; u < 7; u++) t[u] = v[g + (u < rem ? u : 0)];
#pragma unroll
        for (int u = 0; u < 7; u++) acc += (u < rem ? t[u] : 0.0);
    }
    return acc;
}
__device__ __forceinline__ double wave0_fold(const double *v, int count, double acc) {
    const int lane = threadIdx.x & 63;
#pragma nounroll
    for (int c = 0; c < count; c += 64) {
        const double x = (c + lane < count) ? v[c + lane] : 0.0;
        const int left = count - c;
        if (left >= 64) {
#pragma unroll
            for (int l = 0; l < 64; l++) acc += rl64(x, l);
        } else {
#pragma unroll
            for (int l = 0; l < 64; l++) if (l < left) acc += rl64(x, l);
        }
    }
    return acc;
}

// block-wide max of non-negative values (order independent => exact); result to every thread
__device__ double blk_max(double v, double *sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_down(v, o, 64); v = t > v ? t : v; }
    SYNC;
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    SYNC;
    double t = sm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) t = sm[i] > t ? sm[i] : t;
    return t;
}
__device__ int blk_sum_int(int v, int *sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    SYNC;
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    SYNC;
    int t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += sm[i];
    return t;
}
// four maxima / three integer sums with ONE pair of barriers (max and integer addition are exact and order independent)
__device__ void blk_max4(double &a, double &b, double &c, double &d, double *sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_down(a, o, 64); a = t > a ? t : a; t = __shfl_down(b, o, 64); b = t > b ? t : b;
        t = __shfl_down(c, o, 64); c = t > c ? t : c; t = __shfl_down(d, o, 64); d = t > d ? t : d;
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;     // nw <= 8: sm[0 .. 31]
    SYNC;
    if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[8 + w] = b; sm[16 + w] = c; sm[24 + w] = d; }
    SYNC;
    a = sm[0]; b = sm[8]; c = sm[16]; d = sm[24];
    for (int i = 1; i < nw; i++) { a = sm[i] > a ? sm[i] : a; b = sm[8 + i] > b ? sm[8 + i] : b; c = sm[16 + i] > c ? sm[16 + i] : c; d = sm[24 + i] > d ? sm[24 + i] : d; }
    SYNC;                                           // (sm is reused right away by the caller's next reduction)
}
__device__ void blk_sum_int3(int &a, int &b, int &c, int *sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c += __shfl_down(c, o, 64); }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    SYNC;
    if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[8 + w] = b; sm[16 + w] = c; }
    SYNC;
    a = 0; b = 0; c = 0;
    for (int i = 0; i < nw; i++) { a += sm[i]; b += sm[8 + i]; c += sm[16 + i]; }
    SYNC;
}
// The same reductions with ONE barrier (round 4, the solve loop's): the partial results go to one of two banks of a scratch of their own,
// taken in turn.  A wave can write bank(k+2) = bank(k) only after the barrier of reduction k+1, which every wave passes after it has
// read bank(k) -- so neither the barrier before the partials are written nor the one after they are read is needed.  post / read can be
// called apart (work that needs no result of the reduction, and other barriers, may come between them).
struct RedBank { lds_f64 *red; int rb; };
__device__ __forceinline__ lds_f64 *red_next(RedBank &B) { lds_f64 *R = B.red + 32 * B.rb; B.rb ^= 1; return R; }
__device__ __forceinline__ void max4_post(double a, double b, double c, double d, lds_f64 *R) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_down(a, o, 64); a = t > a ? t : a; t = __shfl_down(b, o, 64); b = t > b ? t : b;
        t = __shfl_down(c, o, 64); c = t > c ? t : c; t = __shfl_down(d, o, 64); d = t > d ? t : d;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { R[w] = a; R[8 + w] = b; R[16 + w] = c; R[24 + w] = d; }
}
__device__ __forceinline__ void max4_read(double &a, double &b, double &c, double &d, const lds_f64 *R) {
    const int nw = blockDim.x >> 6;
    a = R[0]; b = R[8]; c = R[16]; d = R[24];
    for (int i = 1; i < nw; i++) { const double ta = R[i], tb = R[8 + i], tc = R[16 + i], td = R[24 + i]; a = ta > a ? ta : a; b = tb > b ? tb : b; c = tc > c ? tc : c; d = td > d ? td : d; }
}
__device__ __forceinline__ void blk_max4_1b(double &a, double &b, double &c, double &d, RedBank &B) {
    lds_f64 *R = red_next(B);
    max4_post(a, b, c, d, R);
    SYNC;
    max4_read(a, b, c, d, R);
}
__device__ __forceinline__ void blk_sum_int3_1b(int &a, int &b, int &c, RedBank &B) {
    __attribute__((address_space(3))) int *R = (__attribute__((address_space(3))) int *)red_next(B);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c += __shfl_down(c, o, 64); }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { R[w] = a; R[8 + w] = b; R[16 + w] = c; }
    SYNC;
    a = 0; b = 0; c = 0;
    for (int i = 0; i < nw; i++) { a += R[i]; b += R[8 + i]; c += R[16 + i]; }
}
// (+0.0) + the NONZERO elements of v[0 .. count) in index order, by one wave (call from all 64 lanes of it): an element that is +0.0 or
// -0.0 never changes a running sum that starts at +0.0 (x + (+-0) = x, and such a sum is never -0), so leaving the zeros out gives the
// bits of the sum over all.  The nonzero ones are first moved to the front of v, in order (ballot + prefix count per 64-element chunk;
// a chunk is in registers before anything is written at or below it), then added by seq_fold.  v is destroyed.
__device__ __forceinline__ double wave_fold_nonzero(lds_f64 *v, int count) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    double x = lane < count ? v[lane] : 0.0;
#pragma nounroll
    for (int c = 0; c < count; c += 64) {
        const double xc = x;
        const int inx = c + 64 + lane;
        x = inx < count ? v[inx] : 0.0;
        const bool nz = xc != 0.0;
        const u64 mask = __ballot(nz);
        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (nz) v[pos] = xc;
        base += __popcll(mask);
    }
    return seq_fold(v, base);
}
// inf-norm of a vector, or of a .* b when b != NULL (lin_alg.c:107-140: NaN never wins)
__device__ double norm_inf(const double *a, const double *b, int n, double *sm) {
    double mx = 0.0;
    FOR_T(i, n) { double s = s_abs(b ? a[i] * b[i] : a[i]); mx = s > mx ? s : mx; }
    return blk_max(mx, sm);
}
// lin_alg.c:59-71: prod += (a0b0 + a1b1 + a2b2 + a3b3) per 4-block, then the scalar tail.  The bracketed group
// sums are independent, so all threads form them (into gbuf, LDS); one lane then adds the groups and the tail in
// the reference's order - the result is bit-identical to the sequential loop.
__device__ double dot_seq(const double *a, const double *b, int n, double *sm, double *gbuf) {
    const int ng = n >> 2;
    SYNC;
    FOR_T(g, ng) { const int i = 4 * g; gbuf[g] = (a[i] * b[i] + a[i + 1] * b[i + 1] + a[i + 2] * b[i + 2] + a[i + 3] * b[i + 3]); }
    FOR_T(t, n - 4 * ng) gbuf[ng + t] = a[4 * ng + t] * b[4 * ng + t];
    SYNC;
    // the groups, then the tail, added in order (gbuf holds them back to back).  Short sums (n/4 or m/4 elements) stay on one lane -- the
    // compiler pipelines these independent loads well, measured 1.5 us per dot against 2.9 us folded by the wave; long ones are folded
    const int cnt = ng + (n - 4 * ng);
    if (cnt >= 256) {
        if (threadIdx.x < 64) { const double prod = wave0_fold(gbuf, cnt, 0.0); if (threadIdx.x == 0) sm[16] = prod; }
    } else if (threadIdx.x == 0) {
        double prod = 0.0;
        for (int g = 0; g < cnt; g++) prod += gbuf[g];
        sm[16] = prod;
    }
    SYNC;
    return sm[16];
}
// y = M x, CSR, one thread per row, ascending columns (== the column-ordered CSC product of the reference)
__device__ void spmv_rows(int nrows, const int *rp, const int *ci, const double *val, const double *x, double *y) {
    FOR_T(r, nrows) {
        double s = 0.0;
        for (int k = rp[r]; k < rp[r + 1]; k++) s += val[k] * x[ci[k]];
        y[r] = s;
    }
}

// The same products with the LATENCY taken out: one thread per row walks its row as a chain of dependent loads (index -> x gather) from
// L2-resident global memory -- ~36 hops for a row of A' at n = 120, m = 360: 5-6 us per product, four products per Newton pass.  Staged:
// every thread forms val[k] * x[ci[k]] for a slice of ALL entries (one hop, fully parallel) into an LDS scratch (the union region, free at
// every call site), then thread r adds row r's products from LDS in ascending k -- the multiplications and the additions of spmv_rows
// in the same order, so the same bits.  Falls back to spmv_rows when the entries do not fit.  One extra barrier inside; the caller
// synchronises before (x complete) and after (y complete) as for spmv_rows.
__device__ void spmv_rows_staged(int nrows, const int *rp, const int *ci, const double *val, const double *x, double *y, double *scr, int cap) {
    const int nnz = __builtin_amdgcn_readfirstlane(rp[nrows]);
    if (nnz > cap) { spmv_rows(nrows, rp, ci, val, x, y); return; }
    FOR_T(k, nnz) scr[k] = val[k] * x[ci[k]];
    SYNC;
    const lds_f64 *sl = (const lds_f64 *)scr;       // (the union region: LDS)
    FOR_T(r, nrows) { const int k0 = rp[r]; y[r] = seq_fold(sl + k0, rp[r + 1] - k0); }
}
// two products of the same vector (Q x and A x) with one barrier between the phases
__device__ void spmv2_rows_staged(int n1, const int *rp1, const int *ci1, const double *val1, double *y1, int n2, const int *rp2, const int *ci2,
                                  const double *val2, double *y2, const double *x, double *scr, int cap) {
    const int nnz1 = __builtin_amdgcn_readfirstlane(rp1[n1]), nnz2 = __builtin_amdgcn_readfirstlane(rp2[n2]);
    if (nnz1 + nnz2 > cap) { spmv_rows(n1, rp1, ci1, val1, x, y1); spmv_rows(n2, rp2, ci2, val2, x, y2); return; }
    double *s2 = scr + nnz1;
    FOR_T(k, nnz1) scr[k] = val1[k] * x[ci1[k]];
    FOR_T(k, nnz2) s2[k] = val2[k] * x[ci2[k]];
    SYNC;
    const lds_f64 *sl1 = (const lds_f64 *)scr, *sl2 = (const lds_f64 *)s2;
    FOR_T(r, n1) { const int k0 = rp1[r]; y1[r] = seq_fold(sl1 + k0, rp1[r + 1] - k0); }
    FOR_T(r, n2) { const int k0 = rp2[r]; y2[r] = seq_fold(sl2 + k0, rp2[r + 1] - k0); }
}

// ---- Ruiz + cost scaling (scaling.c:24-91) ----------------------------------------------------------
__device__ void small_scale(SmallQP &P, int iters, double *D, double *Dinv, double *E, double *Einv, double *tn, double *tm,
                            double &c, double &cinv, double *sm) {
    const int n = P.n, m = P.m;
    FOR_T(i, n) D[i] = 1.0;
    FOR_T(i, m) E[i] = 1.0;
    SYNC;
    for (int it = 0; it < iters; it++) {
        FOR_T(j, n) { double mx = 0.0; for (int k = P.Trp[j]; k < P.Trp[j + 1]; k++) mx = s_max(s_abs(P.Tval[k]), mx); tn[j] = mx; }
        FOR_T(i, m) { double mx = 0.0; for (int k = P.Arp[i]; k < P.Arp[i + 1]; k++) mx = s_max(s_abs(P.Aval[k]), mx); tm[i] = mx; }
        SYNC;
        FOR_T(j, n) { double v = tn[j]; v = v < 1e-9 ? 1.0 : v; v = sqrt(v); v = 1.0 / v; tn[j] = v; D[j] = D[j] * v; }
        FOR_T(i, m) { double v = tm[i]; v = v < 1e-9 ? 1.0 : v; v = sqrt(v); v = 1.0 / v; tm[i] = v; E[i] = E[i] * v; }
        SYNC;
        FOR_T(i, m) for (int k = P.Arp[i]; k < P.Arp[i + 1]; k++) { double v = P.Aval[k]; v = v * tm[i]; v = v * tn[P.Aci[k]]; P.Aval[k] = v; }
        FOR_T(j, n) for (int k = P.Trp[j]; k < P.Trp[j + 1]; k++) { double v = P.Tval[k]; v = v * tm[P.Tci[k]]; v = v * tn[j]; P.Tval[k] = v; }
        SYNC;
    }
    FOR_T(r, n) for (int k = P.Qrp[r]; k < P.Qrp[r + 1]; k++) {
        const int cc = P.Qci[k];
        const double t = r >= cc ? D[cc] * D[r] : D[r] * D[cc];
        P.Qval[k] *= t;
    }
    FOR_T(j, n) P.q[j] = D[j] * P.q[j];
    SYNC;
    const double nq = norm_inf(P.q, nullptr, n, sm);          // Qx = 0 at setup
    c = 1 / s_max(1.0, nq);
    FOR_T(j, n) P.q[j] *= c;
    FOR_T(r, n) for (int k = P.Qrp[r]; k < P.Qrp[r + 1]; k++) P.Qval[k] *= c;
    FOR_T(j, n) Dinv[j] = 1.0 / D[j];
    FOR_T(i, m) Einv[i] = 1.0 / E[i];
    cinv = 1.0 / c;
    FOR_T(i, m) { P.l[i] = E[i] * P.l[i]; P.u[i] = E[i] * P.u[i]; }
    SYNC;
}

// ---- dense system -----------------------------------------------------------------------------------------
// K is addressed as K[koff(j) + i] (i >= j).  When the packed lower triangle fits in LDS it lives there
// (koff = j*n - j(j+1)/2), otherwise in global memory as a full column-major square (koff = j*n).
struct KView {
    double *K; int n; int packed;
    __device__ __forceinline__ size_t off(int j) const { return packed ? (size_t)j * n - (size_t)j * (j + 1) / 2 : (size_t)j * n; }
    __device__ __forceinline__ double &at(int i, int j) const { return K[off(j) + i]; }
};
// Assembly in the oracle's order -- Q first, then the rows r of A ascending, sigma_f last -- WITHOUT a barrier per row: the
// contributions to one entry K(i,j) must be added in ascending r, and they are when ONE thread owns the entry for the whole row loop.
// Column j belongs to the four ADJACENT lanes 4j..4j+3 of one wave (lane `part` takes every fourth entry of a row's tail): they walk column j of A (= row j
// of CSR(A'), rows ascending), and for every weighted row r adds (A_rj d_r) A_ri for the columns i >= j of that row it is responsible
// for -- the products and the order of the former row-by-row loop (one barrier per weighted row: ~120 of them, 16 % of a pass), so
// the same bits.  (An entry K(i,j) may be touched by different lanes of the quartet in different rows; they are lanes of ONE wave
// running the same loop in lockstep, and a wave's LDS operations execute in program order, so the additions still land in row order.)  tpos[q] is the slot of column j inside row r of CSR(A) for entry q of CSR(A') (built once per item, below).
__device__ void small_build_tpos(SmallQP &P) {
    FOR_T(j, P.n) for (int q = P.Trp[j]; q < P.Trp[j + 1]; q++) {
        const int r = P.Tci[q];
        int lo = P.Arp[r], hi = P.Arp[r + 1] - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.Aci[mid] < j) lo = mid + 1; else hi = mid; }
        P.tpos[q] = lo;
    }
}
__device__ void small_assemble(SmallQP &P, const KView &kv, const double *dw, double sigma_f, const int *rp_s, double *d_s) {
    const int n = P.n, m = P.m;
    const size_t tot = kv.packed ? (size_t)n * (n + 1) / 2 : (size_t)n * n;
    for (size_t i = threadIdx.x; i < tot; i += blockDim.x) kv.K[i] = 0.0;
    FOR_T(r, m) d_s[r] = dw[r];                    // weights of this pass into LDS
    SYNC;
    FOR_T(r, n) for (int k = P.Qrp[r]; k < P.Qrp[r + 1]; k++) { const int cc = P.Qci[k]; if (r >= cc) kv.at(r, cc) += P.Qval[k]; }
    SYNC;
    // (round 4) The loop "row index -> weight -> position -> value -> first tail entry" is a chain of dependent loads from L2 per
    // weighted row; four rows at a time their loads go out hop by hop (scalars, not arrays: indexed arrays went to scratch memory), the
    // additions then run in the old order -- row by row, entry by entry -- so the same bits.  24 -> 21.8 us per assembly at n = 120
    // (what remains is the chain of read-modify-writes itself; prefetching more entries per row spilled registers and lost).
    for (int jp = threadIdx.x; jp < 4 * n; jp += blockDim.x) {
        const int j = jp >> 2, part = jp & 3;
        const int q1 = P.Trp[j + 1];
        const size_t offj = kv.off(j);
        for (int q0 = P.Trp[j]; q0 < q1; q0 += 4) {
#define ASM_L1(u) const int qq##u = q0 + u < q1 ? q0 + u : q0; const int rr##u = P.Tci[qq##u], sp##u = P.tpos[qq##u];
#define ASM_L2(u) const double wg##u = d_s[rr##u]; const int ee##u = rp_s[rr##u + 1];
#define ASM_L3(u) const double av##u = P.Aval[sp##u]; const int b##u = sp##u + part; const int bc##u = b##u < ee##u ? b##u : sp##u; \
                  const int c##u = P.Aci[bc##u]; const double a##u = P.Aval[bc##u]; const double vj##u = av##u * wg##u;
#define ASM_L4(u) if (q0 + u < q1 && wg##u != 0.0) { if (b##u < ee##u) kv.K[offj + c##u] += vj##u * a##u; \
                      for (int bb = b##u + 4; bb < ee##u; bb += 4) kv.K[offj + P.Aci[bb]] += vj##u * P.Aval[bb]; }
            ASM_L1(0) ASM_L1(1) ASM_L1(2) ASM_L1(3)
            ASM_L2(0) ASM_L2(1) ASM_L2(2) ASM_L2(3)
            ASM_L3(0) ASM_L3(1) ASM_L3(2) ASM_L3(3)
            ASM_L4(0) ASM_L4(1) ASM_L4(2) ASM_L4(3)
#undef ASM_L1
#undef ASM_L2
#undef ASM_L3
#undef ASM_L4
        }
    }
    SYNC;
    FOR_T(j, n) kv.at(j, j) += sigma_f;
    SYNC;
}
// left-looking by columns, natural order, no pivoting; K holds unit-lower L below and D on the diagonal
// PACKED: K is the packed lower triangle (in LDS); element (i, k) sits at off(k) + i with off(k+1) = off(k) + n - k - 1,
// so the inner loop walks the row with an incremental 32-bit offset.  Separate instantiations keep the LDS
// pointer provenance visible to the compiler (ds_read instead of flat loads).
template <bool PACKED>
__device__ __forceinline__ void small_factor_t(int n, double *__restrict__ K, double *__restrict__ lcol, double *__restrict__ tcol,
                                               double *__restrict__ lcol1, double *__restrict__ tcol1, int k_start = 0, int offk_start = 0) {
    // Right-looking (outer-product) LDL'.  Element (i,j) receives  -= l_ik * (l_jk d_k)  at step k, i.e. the same
    // subtractions in the same ascending-k order as the left-looking reference loop - bit-identical results - but
    // every step updates the whole trailing triangle, so all waves of the workgroup have work.
    // (The reference skips k when l_jk == 0; subtracting x * 0 leaves an element unchanged bit for bit.)
    // TWO columns per barrier pair: the kernel is bound by the latency of its barrier-separated steps, not by arithmetic.  Every
    // thread applies step k to its entry of column k+1 itself -- the subtraction the trailing update would have made, with
    // l_{k+1,k} and d_{k+1} recomputed by every thread from the same LDS values, so the same bits -- and the trailing update then
    // makes the subtractions of step k and of step k+1 one after the other.
    const int li = threadIdx.x & 63, wj = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int offk = offk_start;                          // off(k)
    int k = k_start;
    for (; k + 1 < n; k += 2) {
        const int offk1 = offk + (PACKED ? n - k - 1 : n);            // off(k+1)
        const double dk = K[offk + k], inv = 1.0 / dk;
        const double lk1 = K[offk + k + 1] * inv;                     // l_{k+1,k}
        const double tk1 = lk1 * dk;                                  // its product with d_k: tcol[k+1] of step k
        const double dk1 = K[offk1 + k + 1] - lk1 * tk1;              // K(k+1,k+1) after step k = d_{k+1}
        const double inv1 = 1.0 / dk1;
        // (K(k+1,k) and K(k+1,k+1) are inputs of every thread above: they are overwritten after the barrier, by one thread)
        for (int i = k + 2 + (int)threadIdx.x; i < n; i += blockDim.x) {
            const double l0 = K[offk + i] * inv;
            K[offk + i] = l0; lcol[i] = l0; tcol[i] = l0 * dk;
            const double v1 = K[offk1 + i] - l0 * tk1;                // column k+1 after step k
            const double l1 = v1 * inv1;
            K[offk1 + i] = l1; lcol1[i] = l1; tcol1[i] = l1 * dk1;
        }
        SYNC;
        if (threadIdx.x == 0) { K[offk + k + 1] = lk1; K[offk1 + k + 1] = dk1; }
        // four columns of the trailing triangle per trip: the LDS reads of all four are issued before the first
        // dependent multiply (each element still receives exactly  K(i,j) -= l_ik * (l_jk d_k), k then k+1:  same bits)
        for (int j0 = k + 2 + wj; j0 < n; j0 += 4 * nw) {
            int jj[4], off[4]; double tj0[4], tj1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                jj[u] = j0 + u * nw;
                const int jc = jj[u] < n ? jj[u] : n - 1;
                tj0[u] = tcol[jc]; tj1[u] = tcol1[jc];
                off[u] = PACKED ? jc * n - (jc * (jc + 1)) / 2 : jc * n;
            }
            for (int i = k + 2 + li; i < n; i += 64) {
                const double a0 = lcol[i], a1 = lcol1[i];
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = (jj[u] < n && i >= jj[u]) ? K[off[u] + i] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; u++) if (jj[u] < n && i >= jj[u]) { const double w = v[u] - a0 * tj0[u]; K[off[u] + i] = w - a1 * tj1[u]; }
            }
        }
        SYNC;
        offk = offk1 + (PACKED ? n - k - 2 : n);
    }
    if (k < n) {                                    // odd n: the last column has nothing below it
        SYNC;
    }
}
// FOUR columns per barrier pair (round 3).  Every thread factors the leading 4 x 4 block itself, from the same LDS values and with the
// operations the two-column steps k, k+2 would have made, and carries its rows through all four columns; the trailing update then
// makes the subtractions of steps k .. k+3 one after the other.  Per element: the same subtractions in the same ascending-k order
// -- the same bits as the two-column version (and as the left-looking reference loop) -- with half the barriers and half the
// read-modify-write traffic on K.  F: 8n doubles of LDS (l and l*d of the four columns).  The last n mod 4 columns go through the
// two-column code.
template <bool PACKED>
__device__ __forceinline__ void small_factor4_t(int n, double *__restrict__ K, double *__restrict__ F) {
    double *L0 = F, *L1 = F + n, *L2 = F + 2 * (size_t)n, *L3 = F + 3 * (size_t)n;
    double *T0 = F + 4 * (size_t)n, *T1 = F + 5 * (size_t)n, *T2 = F + 6 * (size_t)n, *T3 = F + 7 * (size_t)n;
    const int li = threadIdx.x & 63, wj = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int off0 = 0, k = 0;
    for (; k + 3 < n; k += 4) {
        const int off1 = off0 + (PACKED ? n - k - 1 : n), off2 = off1 + (PACKED ? n - k - 2 : n), off3 = off2 + (PACKED ? n - k - 3 : n);
        const double d0 = K[off0 + k], inv0 = 1.0 / d0;
        const double l10 = K[off0 + k + 1] * inv0, l20 = K[off0 + k + 2] * inv0, l30 = K[off0 + k + 3] * inv0;
        const double t10 = l10 * d0, t20 = l20 * d0, t30 = l30 * d0;
        const double d1 = K[off1 + k + 1] - l10 * t10, inv1 = 1.0 / d1;
        const double l21 = (K[off1 + k + 2] - l20 * t10) * inv1, l31 = (K[off1 + k + 3] - l30 * t10) * inv1;
        const double t21 = l21 * d1, t31 = l31 * d1;
        const double d2 = (K[off2 + k + 2] - l20 * t20) - l21 * t21, inv2 = 1.0 / d2;
        const double l32 = ((K[off2 + k + 3] - l30 * t20) - l31 * t21) * inv2, t32 = l32 * d2;
        const double d3 = ((K[off3 + k + 3] - l30 * t30) - l31 * t31) - l32 * t32, inv3 = 1.0 / d3;
        for (int i = k + 4 + (int)threadIdx.x; i < n; i += blockDim.x) {
            const double l0 = K[off0 + i] * inv0;
            const double v1 = K[off1 + i] - l0 * t10;
            const double l1 = v1 * inv1;
            const double v2 = (K[off2 + i] - l0 * t20) - l1 * t21;
            const double l2 = v2 * inv2;
            const double v3 = ((K[off3 + i] - l0 * t30) - l1 * t31) - l2 * t32;
            const double l3 = v3 * inv3;
            K[off0 + i] = l0; K[off1 + i] = l1; K[off2 + i] = l2; K[off3 + i] = l3;
            L0[i] = l0; L1[i] = l1; L2[i] = l2; L3[i] = l3;
            T0[i] = l0 * d0; T1[i] = l1 * d1; T2[i] = l2 * d2; T3[i] = l3 * d3;
        }
        SYNC;
        if (threadIdx.x == 0) {                       // (inputs of every thread above: overwritten after the barrier)
            K[off0 + k + 1] = l10; K[off0 + k + 2] = l20; K[off0 + k + 3] = l30;
            K[off1 + k + 1] = d1;  K[off1 + k + 2] = l21; K[off1 + k + 3] = l31;
            K[off2 + k + 2] = d2;  K[off2 + k + 3] = l32; K[off3 + k + 3] = d3;
        }
        for (int j0 = k + 4 + wj; j0 < n; j0 += 2 * nw) {
            int jj[2], off[2]; double tj[2][4];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                jj[u] = j0 + u * nw;
                const int jc = jj[u] < n ? jj[u] : n - 1;
                tj[u][0] = T0[jc]; tj[u][1] = T1[jc]; tj[u][2] = T2[jc]; tj[u][3] = T3[jc];
                off[u] = PACKED ? jc * n - (jc * (jc + 1)) / 2 : jc * n;
            }
            for (int i = k + 4 + li; i < n; i += 64) {
                const double a0 = L0[i], a1 = L1[i], a2 = L2[i], a3 = L3[i];
                double v[2];
#pragma unroll
                for (int u = 0; u < 2; u++) v[u] = (jj[u] < n && i >= jj[u]) ? K[off[u] + i] : 0.0;
#pragma unroll
                for (int u = 0; u < 2; u++) if (jj[u] < n && i >= jj[u]) {
                    double w = v[u] - a0 * tj[u][0];
                    w = w - a1 * tj[u][1];
                    w = w - a2 * tj[u][2];
                    K[off[u] + i] = w - a3 * tj[u][3];
                }
            }
        }
        SYNC;
        off0 = off3 + (PACKED ? n - k - 4 : n);
    }
    if (k < n) small_factor_t<PACKED>(n, K, L0, T0, L1, T1, k, off0);
}
// The four-column factorization with LOOK-AHEAD (round 4; K packed in LDS; NR = rows per lane and trip -- 2 was measured slower: spills).  In small_factor4_t every thread factors the leading 4 x 4
// block itself -- four dependent IEEE divisions, ~200 fp64 instructions, in all eight waves: ~2000 issue cycles per SIMD and step, on the
// critical path between two barriers -- and the trailing update waits for it.  Here wave 0 alone works one step ahead: during step k it
// applies step k to columns k+4 .. k+7 (all rows), factors their 4 x 4 block, scales the four columns and leaves l and l*d of step k+4 in
// the OTHER of two buffers, while waves 1 .. 7 apply step k to the columns from k+8 on; one barrier per step.  Every element receives the
// subtractions of steps k, k+1, ... in ascending order with the operands of small_factor4_t -- which wave makes them and when is all
// that changes -- so the same bits.  F: 16n doubles (two buffers of l and l*d of four columns).  The last n mod 4 columns go through the
// two-column code.
template <int NR>
__device__ __forceinline__ void factor4_block_and_scale(int n, int k, lds_f64 *K, int off0, lds_f64 *Fn, int lane) {
    // columns k .. k+3 have received every earlier step: factor the 4 x 4 block (every lane alike), scale rows >= k+4 (lane by lane)
    lds_f64 *L0 = Fn, *L1 = Fn + n, *L2 = Fn + 2 * (size_t)n, *L3 = Fn + 3 * (size_t)n;
    lds_f64 *T0 = Fn + 4 * (size_t)n, *T1 = Fn + 5 * (size_t)n, *T2 = Fn + 6 * (size_t)n, *T3 = Fn + 7 * (size_t)n;
    const int off1 = off0 + (n - k - 1), off2 = off1 + (n - k - 2), off3 = off2 + (n - k - 3);
    const double d0 = K[off0 + k], inv0 = 1.0 / d0;
    const double l10 = K[off0 + k + 1] * inv0, l20 = K[off0 + k + 2] * inv0, l30 = K[off0 + k + 3] * inv0;
    const double t10 = l10 * d0, t20 = l20 * d0, t30 = l30 * d0;
    const double d1 = K[off1 + k + 1] - l10 * t10, inv1 = 1.0 / d1;
    const double l21 = (K[off1 + k + 2] - l20 * t10) * inv1, l31 = (K[off1 + k + 3] - l30 * t10) * inv1;
    const double t21 = l21 * d1, t31 = l31 * d1;
    const double d2 = (K[off2 + k + 2] - l20 * t20) - l21 * t21, inv2 = 1.0 / d2;
    const double l32 = ((K[off2 + k + 3] - l30 * t20) - l31 * t21) * inv2, t32 = l32 * d2;
    const double d3 = ((K[off3 + k + 3] - l30 * t30) - l31 * t31) - l32 * t32;
    const double inv3 = 1.0 / d3;
    if constexpr (NR == 1) {
        for (int i = k + 4 + lane; i < n; i += 64) {
            const double l0 = K[off0 + i] * inv0;
            const double v1 = K[off1 + i] - l0 * t10;
            const double l1 = v1 * inv1;
            const double v2 = (K[off2 + i] - l0 * t20) - l1 * t21;
            const double l2 = v2 * inv2;
            const double v3 = ((K[off3 + i] - l0 * t30) - l1 * t31) - l2 * t32;
            const double l3 = v3 * inv3;
            K[off0 + i] = l0; K[off1 + i] = l1; K[off2 + i] = l2; K[off3 + i] = l3;
            L0[i] = l0; L1[i] = l1; L2[i] = l2; L3[i] = l3;
            T0[i] = l0 * d0; T1[i] = l1 * d1; T2[i] = l2 * d2; T3[i] = l3 * d3;
        }
    } else
    // two rows per lane and trip (i, i + 64): two independent chains of ten dependent operations side by side
    for (int i = k + 4 + lane; i < n; i += 128) {
        const int ib = i + 64 < n ? i + 64 : i;       // (clamped: the second row's results are dropped when it does not exist)
        const double ka0 = K[off0 + i], ka1 = K[off1 + i], ka2 = K[off2 + i], ka3 = K[off3 + i];
        const double kb0 = K[off0 + ib], kb1 = K[off1 + ib], kb2 = K[off2 + ib], kb3 = K[off3 + ib];
        const double l0 = ka0 * inv0, m0 = kb0 * inv0;
        const double v1 = ka1 - l0 * t10, u1 = kb1 - m0 * t10;
        const double l1 = v1 * inv1, m1 = u1 * inv1;
        const double v2 = (ka2 - l0 * t20) - l1 * t21, u2 = (kb2 - m0 * t20) - m1 * t21;
        const double l2 = v2 * inv2, m2 = u2 * inv2;
        const double v3 = ((ka3 - l0 * t30) - l1 * t31) - l2 * t32, u3 = ((kb3 - m0 * t30) - m1 * t31) - m2 * t32;
        const double l3 = v3 * inv3, m3 = u3 * inv3;
        K[off0 + i] = l0; K[off1 + i] = l1; K[off2 + i] = l2; K[off3 + i] = l3;
        L0[i] = l0; L1[i] = l1; L2[i] = l2; L3[i] = l3;
        T0[i] = l0 * d0; T1[i] = l1 * d1; T2[i] = l2 * d2; T3[i] = l3 * d3;
        if (ib != i) {
            K[off0 + ib] = m0; K[off1 + ib] = m1; K[off2 + ib] = m2; K[off3 + ib] = m3;
            L0[ib] = m0; L1[ib] = m1; L2[ib] = m2; L3[ib] = m3;
            T0[ib] = m0 * d0; T1[ib] = m1 * d1; T2[ib] = m2 * d2; T3[ib] = m3 * d3;
        }
    }
    if (lane == 0) {                                  // (read by every lane above: the wave's LDS operations execute in program order)
        K[off0 + k + 1] = l10; K[off0 + k + 2] = l20; K[off0 + k + 3] = l30;
        K[off1 + k + 1] = d1;  K[off1 + k + 2] = l21; K[off1 + k + 3] = l31;
        K[off2 + k + 2] = d2;  K[off2 + k + 3] = l32; K[off3 + k + 3] = d3;
    }
}
// step k applied to NC columns jj[u] (rows from row0 on, this wave's lanes): K(i,j) -= l_i0 t_j0, -= l_i1 t_j1, -= l_i2 t_j2, -= l_i3 t_j3
template <int NC, int NR>
__device__ __forceinline__ void factor4_apply(int n, lds_f64 *K, const lds_f64 *Fc, const int (&jj)[NC], int row0, int lane) {
    const lds_f64 *L0 = Fc, *L1 = Fc + n, *L2 = Fc + 2 * (size_t)n, *L3 = Fc + 3 * (size_t)n;
    const lds_f64 *T0 = Fc + 4 * (size_t)n, *T1 = Fc + 5 * (size_t)n, *T2 = Fc + 6 * (size_t)n, *T3 = Fc + 7 * (size_t)n;
    int off[NC]; double tj[NC][4];
#pragma unroll
    for (int u = 0; u < NC; u++) {
        const int jc = jj[u] < n ? jj[u] : n - 1;
        tj[u][0] = T0[jc]; tj[u][1] = T1[jc]; tj[u][2] = T2[jc]; tj[u][3] = T3[jc];
        off[u] = jc * n - (jc * (jc + 1)) / 2;
    }
    // NR rows per lane and trip (i, i + 64, ...): their loads are issued together and their chains run side by side
    for (int i0 = row0 + lane; i0 < n; i0 += 64 * NR) {
        int ir[NR]; double a[NR][4], v[NR][NC];
#pragma unroll
        for (int r = 0; r < NR; r++) { const int i = i0 + 64 * r; ir[r] = i < n ? i : i0; }      // (clamped: a row that does not exist stores nothing)
#pragma unroll
        for (int r = 0; r < NR; r++) { a[r][0] = L0[ir[r]]; a[r][1] = L1[ir[r]]; a[r][2] = L2[ir[r]]; a[r][3] = L3[ir[r]]; }
#pragma unroll
        for (int r = 0; r < NR; r++)
#pragma unroll
            for (int u = 0; u < NC; u++) v[r][u] = K[off[u] + ir[r]];      // (jc <= n-1, i <= n-1: inside the packed array whatever the guard says)
#pragma unroll
        for (int r = 0; r < NR; r++)
#pragma unroll
            for (int u = 0; u < NC; u++) if (jj[u] < n && ir[r] >= jj[u] && (r == 0 || i0 + 64 * r < n)) {
                double w = v[r][u] - a[r][0] * tj[u][0];
                w = w - a[r][1] * tj[u][1];
                w = w - a[r][2] * tj[u][2];
                K[off[u] + ir[r]] = w - a[r][3] * tj[u][3];
            }
    }
}
template <int NR>
__device__ __forceinline__ void small_factor4_la(int n, double *Kg, double *Fg) {
    lds_f64 *K = (lds_f64 *)Kg, *F = (lds_f64 *)Fg;
    const int li = threadIdx.x & 63, wj = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int off0 = 0, k = 0, cur = 0;
    if (n >= 4 && wj == 0) factor4_block_and_scale<NR>(n, 0, K, 0, F, li);
    SYNC;
    for (; k + 3 < n; k += 4) {
        const lds_f64 *Fc = F + (size_t)cur * 8 * n; lds_f64 *Fn = F + (size_t)(cur ^ 1) * 8 * n;
        const int off4 = off0 + (n - k - 1) + (n - k - 2) + (n - k - 3) + (n - k - 4);       // off(k+4)
        const bool ahead = (k + 7 < n) && nw > 1;     // a whole next block: wave 0 takes it one step ahead
        if (ahead && wj == 0) {
            const int jj[4] = { k + 4, k + 5, k + 6, k + 7 };
            factor4_apply<4, NR>(n, K, Fc, jj, k + 4, li);
            factor4_block_and_scale<NR>(n, k + 4, K, off4, Fn, li);
        } else {
            const int first = ahead ? k + 8 : k + 4, w0 = ahead ? wj - 1 : wj, nwt = ahead ? nw - 1 : nw;
            for (int j0 = first + w0; j0 < n; j0 += 4 * nwt) {
                const int jj[4] = { j0, j0 + nwt, j0 + 2 * nwt, j0 + 3 * nwt };
                factor4_apply<4, NR>(n, K, Fc, jj, k + 4, li);
            }
        }
        SYNC;
        if (!ahead && k + 7 < n) {                     // (one wave only: nobody worked ahead)
            if (wj == 0) factor4_block_and_scale<NR>(n, k + 4, K, off4, Fn, li);
            SYNC;
        }
        off0 = off4; cur ^= 1;
    }
    if (k < n) small_factor_t<true>(n, Kg, Fg, Fg + 4 * (size_t)n, Fg + n, Fg + 5 * (size_t)n, k, off0);
}
// x lives in LDS (xs) for the duration of the solve.  Two columns per barrier, as in the factorization: every thread applies step j
// to entry j+1 itself (x_{j+1} = xs[j+1] - L(j+1,j) x_j, the subtraction the column sweep would have made), then each entry receives
// the subtractions of column j and of column j+1 in that order -- the operations of the one-column loop, so the same bits.
// Both triangular solves by ONE wave without a barrier (n <= 256, K packed in LDS): lane l keeps rows l, l + 64, ... of x in registers,
// x_j is broadcast with v_readlane, every row receives its subtractions in ascending (descending) j -- the operations of the column-
// oriented loops below, element for element, so the same bits -- and the L entries of the next four steps are loaded before they are
// needed (their addresses do not depend on x).  120 + 120 dependent steps of ~30 cycles instead of 120 workgroup barriers.
// (round 4) The factor is addressed as LDS (ds_read; through the generic pointer every load was a flat_load inside its own exec-mask branch,
// and the wave waited ~250 cycles for the four steps' entries before each group), every load is unconditional at a clamped address
// (the guard sits on the update), and the entries of the NEXT group are in flight while a group is applied.
template <int RPL, int GS>
__device__ __forceinline__ void small_ldl_solve_wave(int n, const lds_f64 *__restrict__ Kp, const double *__restrict__ b, double *__restrict__ xout) {
    const int lane = threadIdx.x & 63;
    double x[RPL]; int offi[RPL], icl[RPL];
#pragma unroll
    for (int r = 0; r < RPL; r++) { const int i = lane + 64 * r; x[r] = i < n ? b[i] : 0.0; const int ic = i < n ? i : n - 1; icl[r] = ic; offi[r] = ic * n - (ic * (ic + 1)) / 2; }
    double LA[GS][RPL], LB[GS][RPL];
    // L z = b: step j eliminates x_j from the rows below;  L(i,j) sits at off(j) + i
#define FWD_LOAD(L, j0_) do { _Pragma("unroll") for (int u = 0; u < GS; u++) { const int j = (j0_) + u; const int jc = j < n ? j : n - 1; const int offj = jc * n - (jc * (jc + 1)) / 2; \
        _Pragma("unroll") for (int r = 0; r < RPL; r++) L[u][r] = Kp[offj + icl[r]]; } } while (0)
#define FWD_STEPS(L, s_, jl_, j0_) do { _Pragma("unroll") for (int u = 0; u < GS; u++) { const int j = (j0_) + u; \
        if (j < n - 1) { const double xj = rl64(x[s_], (jl_) + u); \
            _Pragma("unroll") for (int r = 0; r < RPL; r++) { const int i = lane + 64 * r; if (i > j && i < n) x[r] = x[r] - L[u][r] * xj; } } } } while (0)
    FWD_LOAD(LA, 0);
#pragma unroll
    for (int s = 0; s < RPL; s++) {
        for (int jl = 0; jl < 64; jl += 2 * GS) {
            const int j0 = 64 * s + jl;
            if (j0 >= n - 1) break;
            FWD_LOAD(LB, j0 + GS);
            FWD_STEPS(LA, s, jl, j0);
            FWD_LOAD(LA, j0 + 2 * GS);
            FWD_STEPS(LB, s, jl + GS, j0 + GS);
        }
    }
#undef FWD_LOAD
#undef FWD_STEPS
#pragma unroll
    for (int r = 0; r < RPL; r++) { const int i = lane + 64 * r; const double dg = Kp[offi[r] + icl[r]]; if (i < n) x[r] = x[r] / dg; }
    // L' x = z: step j (descending) eliminates x_j from the rows above;  L(j,i) sits at off(i) + j
#define BWD_LOAD(L, j0_) do { _Pragma("unroll") for (int u = 0; u < GS; u++) { const int j = (j0_) - u; const int jc = j < 0 ? 0 : (j < n ? j : n - 1); \
        _Pragma("unroll") for (int r = 0; r < RPL; r++) L[u][r] = Kp[offi[r] + jc]; } } while (0)
#define BWD_STEPS(L, s_, jl_, j0_) do { _Pragma("unroll") for (int u = 0; u < GS; u++) { const int j = (j0_) - u; \
        if (j >= 1 && j <= n - 1) { const double xj = rl64(x[s_], (jl_) - u); \
            _Pragma("unroll") for (int r = 0; r < RPL; r++) { const int i = lane + 64 * r; if (i < j) x[r] = x[r] - L[u][r] * xj; } } } } while (0)
    BWD_LOAD(LA, 64 * RPL - 1);
#pragma unroll
    for (int s = RPL - 1; s >= 0; s--) {
        for (int jl = 63; jl >= 0; jl -= 2 * GS) {
            const int j0 = 64 * s + jl;
            if (j0 < 1) break;
            BWD_LOAD(LB, j0 - GS);
            if (j0 - (GS - 1) <= n - 1) BWD_STEPS(LA, s, jl, j0);
            BWD_LOAD(LA, j0 - 2 * GS);
            if (j0 - (2 * GS - 1) <= n - 1) BWD_STEPS(LB, s, jl - GS, j0 - GS);
        }
    }
#undef BWD_LOAD
#undef BWD_STEPS
#pragma unroll
    for (int r = 0; r < RPL; r++) { const int i = lane + 64 * r; if (i < n) xout[i] = x[r]; }
}
__device__ void small_ldl_solve(SmallQP &P, const KView &kv, const double *b, double *xout, double *xs) {
    const int n = P.n;
    if (kv.packed && n <= 256) {
        if (threadIdx.x < 64) {
            const lds_f64 *Kl = (const lds_f64 *)kv.K;             // packed => the factor is in the workgroup's LDS
            if (n <= 64) small_ldl_solve_wave<1, 4>(n, Kl, b, xout);
            else if (n <= 128) small_ldl_solve_wave<2, 4>(n, Kl, b, xout);
            else if (n <= 192) small_ldl_solve_wave<3, 2>(n, Kl, b, xout);
            else small_ldl_solve_wave<4, 2>(n, Kl, b, xout);
        }
        SYNC;
        return;
    }
    FOR_T(i, n) xs[i] = b[i];
    SYNC;
    for (int j = 0; j + 1 < n; j += 2) {          // L z = b  (x_j is final when the loop reaches it)
        const double xj = xs[j];
        const double xj1 = xs[j + 1] - kv.at(j + 1, j) * xj;
        FOR_T(ii, n - j - 2) { const int i = j + 2 + ii; const double t = xs[i] - kv.at(i, j) * xj; xs[i] = t - kv.at(i, j + 1) * xj1; }
        SYNC;
        if (threadIdx.x == 0) xs[j + 1] = xj1;    // (an input of every thread above; read again only after the barrier below the loop)
    }
    SYNC;
    FOR_T(j, n) xs[j] /= kv.at(j, j);
    SYNC;
    for (int j = n - 1; j >= 1; j -= 2) {         // L' x = z, column oriented: x_j is final, eliminate it from the rows above
        const double xj = xs[j];
        const double xj1 = xs[j - 1] - kv.at(j, j - 1) * xj;
        FOR_T(i, j - 1) { const double t = xs[i] - kv.at(j, i) * xj; xs[i] = t - kv.at(j - 1, i) * xj1; }
        SYNC;
        if (threadIdx.x == 0) xs[j - 1] = xj1;
    }
    SYNC;
    FOR_T(i, n) xout[i] = xs[i];
    SYNC;
}

// ---- linesearch (linesearch.c:8-158) -------------------------------------------------------------------
// Bitonic network on (key, index) with the elements in REGISTERS: thread t holds elements t + 512 r, r < R (R = np2 / 512, at least 1).
// A compare-exchange whose partner is at distance j >= 512 is inside the thread, 64 <= j < 512 goes through LDS (the output arrays
// double as the exchange buffer: write, barrier, read, barrier), j < 64 is a lane shuffle without a barrier -- 45 of the 55 stages at
// np2 = 1024.  The network and every comparison are those of the all-LDS version it replaces (one barrier per stage): the same
// permutation, i.e. the stable order by (t, index) of the reference's qsort.  Ends with the sorted arrays in skey / sidx.
// PRE: the first M2 (key, index) pairs are already in skey / sidx (a compacted subset of the candidates); the rest is padding.
template <int R, bool PRE = false>
__device__ __forceinline__ void small_sort_regs(int np2, int M2, const lds_f64 *ls_alpha, const lds_f64 *ls_delta, lds_u64 *skey, lds_u32 *sidx) {
    u64 key[R]; u32 idx[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = (int)threadIdx.x + SM_THREADS * r;
        u64 kv = ~0ull; u32 iv = (u32)e;
        if constexpr (PRE) { if (e < M2) { kv = skey[e]; iv = sidx[e]; } }
        else if (e < M2) { const double t = ls_alpha[e] / ls_delta[e]; if (t > 0) kv = (u64)__double_as_longlong(t); }
        key[r] = kv; idx[r] = iv;
    }
    if constexpr (PRE) SYNC;                            // (the arrays double as the exchange buffer below)
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= SM_THREADS) {                     // partner in this thread: elements r and r ^ (j / 512)
                const int dr = j / SM_THREADS;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((r & dr) == 0 && (r | dr) < R) {
                        const int e = (int)threadIdx.x + SM_THREADS * r;
                        const bool up = ((e & k) == 0);
                        const bool gt = (key[r] > key[r | dr]) || (key[r] == key[r | dr] && idx[r] > idx[r | dr]);
                        if (gt == up) { const u64 tk = key[r]; key[r] = key[r | dr]; key[r | dr] = tk; const u32 ti = idx[r]; idx[r] = idx[r | dr]; idx[r | dr] = ti; }
                    }
                }
            } else {
                u64 pk[R]; u32 pi[R];
                if (j >= 64) {
#pragma unroll
                    for (int r = 0; r < R; r++) { const int e = (int)threadIdx.x + SM_THREADS * r; if (e < np2) { skey[e] = key[r]; sidx[e] = idx[r]; } }
                    SYNC;
#pragma unroll
                    for (int r = 0; r < R; r++) { const int e = (int)threadIdx.x + SM_THREADS * r; if (e < np2) { pk[r] = skey[e ^ j]; pi[r] = sidx[e ^ j]; } else { pk[r] = key[r]; pi[r] = idx[r]; } }
                    SYNC;
                } else {
#pragma unroll
                    for (int r = 0; r < R; r++) { pk[r] = (u64)__shfl_xor((long long)key[r], j, 64); pi[r] = (u32)__shfl_xor((int)idx[r], j, 64); }
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int e = (int)threadIdx.x + SM_THREADS * r;
                    const bool mine_gt = (key[r] > pk[r]) || (key[r] == pk[r] && idx[r] > pi[r]);
                    const bool up = ((e & k) == 0), lower = ((e & j) == 0);
                    const bool swap = lower ? (mine_gt == up) : (mine_gt != up);      // pair (i, i^j), i lower: exchanged iff (elem i > elem i^j) == up
                    if (swap) { key[r] = pk[r]; idx[r] = pi[r]; }
                }
            }
        }
#pragma unroll
    for (int r = 0; r < R; r++) { const int e = (int)threadIdx.x + SM_THREADS * r; if (e < np2) { skey[e] = key[r]; sidx[e] = idx[r]; } }
    SYNC;
}
// (its own function, not inlined: the kernel is held to 128 VGPRs, and inlined the register pressure of the sort and of the unrolled
// folds below pushed spills into the factorization and solve loops -- measured: triangular solves 2x slower)
__device__ __forceinline__ double small_linesearch_impl(SmallQP &P, double *V[], double *ls_delta_, double *ls_alpha_, unsigned char *jflag_, double *tm, double *sm_, u64 *skey_, u32 *sidx_, double *gbuf_, double *nscr_) {
    const int n = P.n, m = P.m, M2 = 2 * m;
    // every scratch array of the linesearch is in the workgroup's LDS; said so, the out-of-line copy of the batch kernel reads and writes
    // them with ds instructions instead of flat ones (two counters to wait for, twice the latency)
    lds_f64 *ls_delta = (lds_f64 *)ls_delta_, *ls_alpha = (lds_f64 *)ls_alpha_, *sm = (lds_f64 *)sm_, *gbuf = (lds_f64 *)gbuf_, *nscr = (lds_f64 *)nscr_;
    lds_u8 *jflag = (lds_u8 *)jflag_; lds_u64 *skey = (lds_u64 *)skey_; lds_u32 *sidx = (lds_u32 *)sidx_;
    long long tls = P.prof ? wall_clock64() : 0;
#define PHL(k) do { if (P.prof && threadIdx.x == 0) { const long long t_ = wall_clock64(); P.prof[k] += t_ - tls; tls = t_; } } while (0)
    double *dy = V[MV_DY], *mu = V[MV_MU], *isq = V[MV_ISQ], *w = V[MV_W], *y = V[MV_Y], *Adx = V[MV_ADX];
    // Round 4: the whole linesearch of a pass that stops at its FIRST breakpoint -- 98.5 % of the passes of an instance that crawls towards
    // eps for thousands of passes, 4 % of an ordinary solve's (counted on the C3 batch with an instrumented oracle) -- is two phases and
    // two barriers; the sort (15-17 us of the former 34) runs only when the walk goes on.  linesearch.c:120-127 looks at the sorted
    // breakpoints only through t[0] before the walk starts, and t[0] is the MINIMUM of the candidates: a minimum needs no order.
    // Phase A, every thread: the bracketed 4-groups of the four dot products of linesearch.c:19-25 (lin_alg.c:59-71), delta / alpha
    // (linesearch.c:27-40), the L / P / J flags of its two elements (linesearch.c:84-106) and the smallest positive t it has seen (the
    // bits of a positive double order like an integer).  temp_m = 0.5 (dy .* mu) is recomputed where a group needs it: the same product,
    // the same bits as the stored copy.  The group arrays live outside the union region (solve scratch, gbuf) and in the sort's key
    // array, all idle here, so that delta / alpha can be written in the same phase.
    const double *dxv = P.nv + (size_t)NV_DX * n, *qdx = P.nv + (size_t)NV_QDX * n, *dfv = P.nv + (size_t)NV_DF * n;
    const int ngm = m >> 2, cm = ngm + (m - 4 * ngm), ngn = n >> 2, cn = ngn + (n - 4 * ngn);
    lds_f64 *G0 = gbuf, *G1 = nscr, *G2 = (lds_f64 *)sidx, *G3 = nscr + cn;    // cm <= m/4 + 3 (gbuf; index array: >= m doubles), 2 cn <= n/2 + 6 <= 8n (nscr)
    FOR_T(g, ngm) { const int i = 4 * g;
        double t0 = dy[i] * mu[i], t1 = dy[i + 1] * mu[i + 1], t2 = dy[i + 2] * mu[i + 2], t3 = dy[i + 3] * mu[i + 3];
        t0 = t0 * 0.5; t1 = t1 * 0.5; t2 = t2 * 0.5; t3 = t3 * 0.5;
        G0[g] = (dy[i] * t0 + dy[i + 1] * t1 + dy[i + 2] * t2 + dy[i + 3] * t3);
        G2[g] = (y[i] * t0 + y[i + 1] * t1 + y[i + 2] * t2 + y[i + 3] * t3); }
    FOR_T(t, m - 4 * ngm) { const int i = 4 * ngm + t; double s = dy[i] * mu[i]; s = s * 0.5; G0[ngm + t] = dy[i] * s; G2[ngm + t] = y[i] * s; }
    FOR_T(g, ngn) { const int i = 4 * g;
        G1[g] = (dxv[i] * qdx[i] + dxv[i + 1] * qdx[i + 1] + dxv[i + 2] * qdx[i + 2] + dxv[i + 3] * qdx[i + 3]);
        G3[g] = (dxv[i] * dfv[i] + dxv[i + 1] * dfv[i + 1] + dxv[i + 2] * dfv[i + 2] + dxv[i + 3] * dfv[i + 3]); }
    FOR_T(t, n - 4 * ngn) { G1[ngn + t] = dxv[4 * ngn + t] * qdx[4 * ngn + t]; G3[ngn + t] = dxv[4 * ngn + t] * dfv[4 * ngn + t]; }
    u64 kmin = ~0ull;
    FOR_T(i, m) {
        double s = dy[i] * mu[i]; s = s * 0.5; tm[i] = s;
        double c0 = Adx[i] - s; c0 = c0 * isq[i];
        const double d1 = c0, d0 = c0 * -1.0, a0 = (w[i] - P.l[i]) * isq[i], a1 = (P.u[i] - w[i]) * isq[i];
        ls_delta[i + m] = d1; ls_delta[i] = d0; ls_alpha[i] = a0; ls_alpha[i + m] = a1;
        const double q0 = a0 / d0, q1 = a1 / d1;
        const int L0 = q0 > 0, L1 = q1 > 0, P0 = d0 > 0, P1 = d1 > 0;
        jflag[i] = (unsigned char)(L0 | (((P0 + L0) == 1) << 1)); jflag[i + m] = (unsigned char)(L1 | (((P1 + L1) == 1) << 1));
        if (L0) { const u64 k = (u64)__double_as_longlong(q0); kmin = k < kmin ? k : kmin; }
        if (L1) { const u64 k = (u64)__double_as_longlong(q1); kmin = k < kmin ? k : kmin; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 t = (u64)__shfl_down((long long)kmin, o, 64); kmin = t < kmin ? t : kmin; }
    if ((threadIdx.x & 63) == 0) ((lds_u64 *)sm)[threadIdx.x >> 6] = kmin;         // sm[0 .. 7]
    SYNC;
    PHL(PH_LS_DOTS);
    // Phase B, three waves side by side: lanes 0..3 of wave 0 each add the groups and the tail of ONE dot product in order; wave 1 the sum
    // of delta^2 over J and the count of L, wave 2 the sum of delta*alpha over J (vec_prod_ind, linesearch.c:190-199: sequential, index
    // order).  Only the elements IN J are added: each of the two waves first compacts its products, in index order (ballot + prefix
    // count per 64-element chunk), into its half of the sort's key array, then adds them with the loads running ahead of the additions.
    // J holds one element per active row -- a fifth of the 2m -- and never both elements of a row while l <= u; the former fold over all
    // 2m (+0.0 for the others, 30 cycles per readlane step) took 9.5 us.
    if (threadIdx.x < 4) {
        const lds_f64 *G = threadIdx.x == 0 ? G0 : threadIdx.x == 1 ? G1 : threadIdx.x == 2 ? G2 : G3;
        sm[24 + threadIdx.x] = seq_fold(G, (threadIdx.x & 1) ? cn : cm);
    } else if (threadIdx.x >= 64 && threadIdx.x < 192) {
        const int lane = threadIdx.x & 63, second = (threadIdx.x >> 6) - 1;
        lds_f64 *cb = (lds_f64 *)skey + (size_t)second * m;
        double acc = 0.0; int nLc = 0, base = 0;
        unsigned char f = lane < M2 ? jflag[lane] : 0;
        double dl = lane < M2 ? ls_delta[lane] : 0.0, al = lane < M2 ? ls_alpha[lane] : 0.0;
#pragma nounroll
        for (int c = 0; c < M2; c += 64) {
            const unsigned char fc = f;
            const double v = second ? dl * al : dl * dl;
            const int inx = c + 64 + lane;                                        // the next chunk is in flight while this one is placed
            f = inx < M2 ? jflag[inx] : 0; dl = inx < M2 ? ls_delta[inx] : 0.0; al = inx < M2 ? ls_alpha[inx] : 0.0;
            const bool inJ = (fc & 2) != 0;
            const u64 mask = __ballot(inJ);
            const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (inJ && pos < m) cb[pos] = v;
            base += __popcll(mask);
            if (!second) nLc += __popcll(__ballot(fc & 1));
        }
        if (base <= m) acc = seq_fold(cb, base);
        else {                                       // (bounds with l > u somewhere: more than m elements in J -- add them lane by lane)
#pragma nounroll
            for (int c = 0; c < M2; c += 64) {
                const int i = c + lane;
                const unsigned char fc = i < M2 ? jflag[i] : 0;
                const double d_ = i < M2 ? ls_delta[i] : 0.0, a_ = i < M2 ? ls_alpha[i] : 0.0;
                const double v = second ? d_ * a_ : d_ * d_;
                u64 mask = __ballot((fc & 2) != 0);
                while (mask) { const int l = __builtin_ctzll(mask); acc += rl64(v, l); mask &= mask - 1; }
            }
        }
        if (lane == 0) { sm[28 + (second ? 0 : 1)] = acc; if (!second) sm[30] = (double)nLc; }      // sm[28]: delta*alpha, sm[29]: delta^2
    }
    SYNC;
    PHL(PH_LS_JSUM);
    // every thread: eta, beta, a, b and the first breakpoint -- the same operations on the same values, so the same decision everywhere
    double eta = sm[24]; eta += sm[25]; eta *= 0.5;
    double beta = sm[26]; beta += sm[27]; beta *= 0.5;
    const double sa = sm[29], sb = sm[28];
    const int nL = (int)sm[30];
    double a = eta + sa, b = beta - sb;
    u64 k0 = ((lds_u64 *)sm)[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) { const u64 t = ((lds_u64 *)sm)[i]; k0 = t < k0 ? t : k0; }
    if (nL == 0 || b + a * __longlong_as_double((long long)k0) > 0) {               // linesearch.c:116-127
        PHL(PH_LS_WALK);
        return -b / a;                                 // (the caller's next barrier comes before anything writes sm or the scratch again)
    }
    int np2 = 1; while (np2 < M2) np2 <<= 1;
    // The walk goes on.  It rarely goes far: counted on the C3 batch (instrumented oracle, 4682 walks), 30 breakpoints on average of 435
    // candidates, and in 95 % of the walks it ends below 1.5 tau0, tau0 = -b/a the root before any breakpoint is crossed.  So first the
    // candidates with t <= 1.5 tau0 alone (97 on average): compacted in index order, sorted by (t, index) -- a 64 / 128 / 256-element
    // network instead of the 1024-element one -- and walked; they are exactly the first c elements of the full order, so a walk that stops
    // inside them has seen what the full sort would have shown it.  One that runs off their end starts again on the full sort below.
    if (np2 >= 1024) {
        const double tlim = 1.5 * (-b / a);
        if (tlim > 0.0 && tlim < 1e300) {
            const u64 klim = (u64)__double_as_longlong(tlim);
            const int per0 = (M2 + SM_THREADS - 1) / SM_THREADS, wv = threadIdx.x >> 6;
            u64 kk[4]; bool in[4]; int pre[4];
            __attribute__((address_space(3))) int *cnts = (__attribute__((address_space(3))) int *)gbuf;      // 4 x 8 wave counts (the group sums are spent)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int e = (int)threadIdx.x + SM_THREADS * r;
                kk[r] = ~0ull; in[r] = false;
                if (r < per0 && e < M2 && (jflag[e] & 1)) { const double t = ls_alpha[e] / ls_delta[e]; kk[r] = (u64)__double_as_longlong(t); in[r] = kk[r] <= klim; }
                const u64 mask = __ballot(in[r]);
                pre[r] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if ((threadIdx.x & 63) == 0) cnts[8 * r + wv] = __popcll(mask);
            }
            SYNC;
            int c = 0, off[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int r = 0; r < 4; r++)
                for (int w2 = 0; w2 < SM_THREADS / 64; w2++) { if (w2 == wv) off[r] = c; c += cnts[8 * r + w2]; }
            if (c <= 256) {
#pragma unroll
                for (int r = 0; r < 4; r++) if (in[r]) { skey[off[r] + pre[r]] = kk[r]; sidx[off[r] + pre[r]] = (u32)((int)threadIdx.x + SM_THREADS * r); }
                SYNC;
                const int np2p = c <= 64 ? 64 : c <= 128 ? 128 : 256;
                small_sort_regs<1, true>(np2p, c, ls_alpha, ls_delta, skey, sidx);
                lds_f64 *D2 = (lds_f64 *)(skey + 256), *A2 = (lds_f64 *)(skey + 512);       // (np2 >= 1024 keys: room behind the subset)
                if ((int)threadIdx.x < c) { const u32 iz = sidx[threadIdx.x]; D2[threadIdx.x] = ls_delta[iz]; A2[threadIdx.x] = ls_alpha[iz]; }
                SYNC;
                if (threadIdx.x == 0) {
                    double aw = a, bw = b;
                    int i = 0; bool found = false;
#pragma nounroll
                    while (i < c - 1 && !found) {
                        const int left = c - 1 - i;
                        double dl[4], al[4], tn[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) if (u < left) { dl[u] = D2[i + u]; al[u] = A2[i + u]; tn[u] = __longlong_as_double((long long)skey[i + u + 1]); }
#pragma unroll
                        for (int u = 0; u < 4; u++) if (u < left && !found) {
                            if (dl[u] > 0) { aw = aw + dl[u] * dl[u]; bw = bw - dl[u] * al[u]; } else { aw = aw - dl[u] * dl[u]; bw = bw + dl[u] * al[u]; }
                            i++;
                            if (bw + aw * tn[u] > 0) found = true;
                        }
                    }
                    if (!found && c == nL) {              // the subset is all there is: the last breakpoint, linesearch.c:148-157
                        const double dl = D2[i], al = A2[i];
                        if (dl > 0) { aw = aw + dl * dl; bw = bw - dl * al; } else { aw = aw - dl * dl; bw = bw + dl * al; }
                        found = true;
                    }
                    sm[16] = found ? 1.0 : 0.0; sm[17] = -bw / aw;
                }
                SYNC;
                if (sm[16] != 0.0) { PHL(PH_LS_SORT); return sm[17]; }
                SYNC;                                     // (sm[16] is read by everyone before the full path may write it again)
            }
        }
    }
    // stable order by (t, index) of ALL candidates = the reference's qsort
    if (np2 <= SM_THREADS) small_sort_regs<1>(np2, M2, ls_alpha, ls_delta, skey, sidx);
    else if (np2 == 2 * SM_THREADS) small_sort_regs<2>(np2, M2, ls_alpha, ls_delta, skey, sidx);
    else small_sort_regs<4>(np2, M2, ls_alpha, ls_delta, skey, sidx);
    PHL(PH_LS_SORT);
    // The walk runs on one lane at the latency of its dependent LDS reads unless the reads are taken out of the dependence chain: it reads
    // (delta, alpha) from arrays that every thread has first rearranged INTO the sorted order (in place, through registers), so that it
    // can load ahead.
    const int per = (M2 + (int)blockDim.x - 1) / (int)blockDim.x;          // <= 4 sorted positions per thread (2m <= 2048)
    double gd[4], ga[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int e = (int)threadIdx.x + r * (int)blockDim.x;
        gd[r] = 0.0; ga[r] = 0.0;
        if (r < per && e < M2) { const u32 iz = sidx[e]; if (iz < (u32)M2) { gd[r] = ls_delta[iz]; ga[r] = ls_alpha[iz]; } }
    }
    SYNC;                                             // everyone is done with the index-ordered arrays
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int e = (int)threadIdx.x + r * (int)blockDim.x;
        if (r < per && e < M2) { ls_delta[e] = gd[r]; ls_alpha[e] = ga[r]; }
    }
    SYNC;
    if (threadIdx.x == 0) {
        double tau;
        int i = 0; bool found = false;
#pragma nounroll
        while (i < nL - 1 && !found) {
            const int left = nL - 1 - i;
            double dl[4], al[4], tn[4];
#pragma unroll
            for (int u = 0; u < 4; u++) if (u < left) { dl[u] = ls_delta[i + u]; al[u] = ls_alpha[i + u]; tn[u] = __longlong_as_double((long long)skey[i + u + 1]); }
#pragma unroll
            for (int u = 0; u < 4; u++) if (u < left && !found) {
                if (dl[u] > 0) { a = a + dl[u] * dl[u]; b = b - dl[u] * al[u]; } else { a = a - dl[u] * dl[u]; b = b + dl[u] * al[u]; }
                i++;
                if (b + a * tn[u] > 0) found = true;
            }
        }
        if (!found) {
            const double dl = ls_delta[i], al = ls_alpha[i];
            if (dl > 0) { a = a + dl * dl; b = b - dl * al; } else { a = a - dl * dl; b = b + dl * al; }
        }
        tau = -b / a;
        sm[17] = tau;
    }
    PHL(PH_LS_WALK);
#undef PHL
    SYNC;
    return sm[17];
}
// The batch kernel calls it out of line (see above); the latency variant, with twice the registers, inlines it (no pointer tables in scratch).
__device__ __attribute__((noinline)) double small_linesearch(SmallQP &P, double *V[], double *ls_delta, double *ls_alpha, unsigned char *jflag, double *tm, double *sm, u64 *skey, u32 *sidx, double *gbuf, double *nscr) {
    return small_linesearch_impl(P, V, ls_delta, ls_alpha, jflag, tm, sm, skey, sidx, gbuf, nscr);
}

__device__ void small_status(QPDOInfo &info, long st) {
    info.status_val = st;
    const char *s = "unrecognised status value";
    switch (st) {
        case QPDO_SOLVED: s = "solved"; break;
        case QPDO_PRIMAL_INFEASIBLE: s = "primal infeasible"; break;
        case QPDO_DUAL_INFEASIBLE: s = "dual infeasible"; break;
        case QPDO_MAX_ITER_REACHED: s = "maximum iterations reached"; break;
        case QPDO_MAX_TIME_REACHED: s = "max time exceeded"; break;
        case QPDO_NON_CVX: s = "unrecognised status value"; break;
        case QPDO_UNSOLVED: s = "unsolved"; break;
        case QPDO_ERROR: s = "error"; break;
    }
    int i = 0;
    for (; s[i] && i < 31; i++) info.status[i] = s[i];
    info.status[i] = 0;
}

// compute_objective (iteration.c:185-221), one lane: sum of (0.5 (Qx - sigma x) + q) x in the 4-way grouped order of the reference
__device__ __forceinline__ double small_objective(int n, int prox, double sigma, const double *Qx, const double *x, const double *q) {
    double obj = 0; int i = 0;
    if (prox) {
        if (n >= 4) for (; i <= n - 4; i += 4)
            obj += (0.5 * (Qx[i] - x[i] * sigma) + q[i]) * x[i] + (0.5 * (Qx[i + 1] - x[i + 1] * sigma) + q[i + 1]) * x[i + 1] +
                   (0.5 * (Qx[i + 2] - x[i + 2] * sigma) + q[i + 2]) * x[i + 2] + (0.5 * (Qx[i + 3] - x[i + 3] * sigma) + q[i + 3]) * x[i + 3];
        for (; i < n; i++) obj += (0.5 * (Qx[i] - sigma * x[i]) + q[i]) * x[i];
    } else {
        if (n >= 4) for (; i <= n - 4; i += 4)
            obj += (0.5 * Qx[i] + q[i]) * x[i] + (0.5 * Qx[i + 1] + q[i + 1]) * x[i + 1] + (0.5 * Qx[i + 2] + q[i + 2]) * x[i + 2] +
                   (0.5 * Qx[i + 3] + q[i + 3]) * x[i + 3];
        for (; i < n; i++) obj += (0.5 * Qx[i] + q[i]) * x[i];
    }
    return obj;
}
template <class T>
__device__ __forceinline__ T *uni_ptr(T *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
// ---- the whole solve of one QP by one workgroup ----------------------------------------------------------
// LAT = 0: the batch kernel (held to 128 VGPRs so that two workgroups share a CU); LAT = 1: the latency variant for ONE workspace
// (qdev_small_resident_solve): the same code with the whole register file of a CU's SIMDs to itself (no spills) -- same operations
// in the same order, so the same bits.
template <int LAT>
__device__ __forceinline__ void small_solve_body(SmallQP *probs, int count, const QPDOSettings &st, int kflags) {
    const int klds_ok = kflags & 1, ucap = kflags >> 1;          // bit 0: the packed factor lives in LDS; the rest: doubles in the union region U
    __shared__ double sm[32];
    __shared__ double red_scr[64];                    // two banks of reduction partials (RedBank)
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    // dynamic LDS: [xs: n][colbuf: n][tk: 2n][12n more for the four-column factorization with look-ahead][gbuf][d_s: m][rp_s: m+1][U], U = one region shared by the packed factor K
    // (if it fits) and the linesearch scratch (delta, alpha, sort keys, sort indices, flags).  The linesearch of a
    // pass runs after the pass's solve, so it may overwrite K: the factor is then rebuilt in the next pass instead
    // of being reused when the weights did not change -- the same bits, a little more work -- and the workgroup needs
    // ~67 KB instead of ~104 KB at n = 120, m = 360: two workgroups per CU instead of one.
    if ((int)blockIdx.x >= count) return;
    // Every field of the item's descriptor is the same for all threads, but loaded from global memory it lands in VECTOR registers, and
    // so does every pointer derived from it (~45 of them are live across the solve loop: two thirds of the 128-VGPR budget).  Passing
    // each field through v_readfirstlane tells the compiler it is uniform: the descriptor and the address arithmetic move to SGPRs.
    SmallQP &Pg = probs[blockIdx.x];                  // (outputs are written through this one)
    SmallQP P;
    P.n = __builtin_amdgcn_readfirstlane(Pg.n); P.m = __builtin_amdgcn_readfirstlane(Pg.m);
#define UNI_PTR(f) P.f = uni_ptr(Pg.f)
    UNI_PTR(Arp); UNI_PTR(Aci); UNI_PTR(Aval); UNI_PTR(Trp); UNI_PTR(Tci); UNI_PTR(Tval); UNI_PTR(Qrp); UNI_PTR(Qci); UNI_PTR(Qval);
    UNI_PTR(q); UNI_PTR(l); UNI_PTR(u); UNI_PTR(x0); UNI_PTR(y0); UNI_PTR(nv); UNI_PTR(mv); UNI_PTR(lsv); UNI_PTR(iv); UNI_PTR(tpos); UNI_PTR(K);
    UNI_PTR(sol_x); UNI_PTR(sol_y); UNI_PTR(cert_dx); UNI_PTR(cert_dy); UNI_PTR(prof);
    // resident-mode extras (compiled out of the batch kernel: its register budget is spoken for)
    SmallRes *Rg = nullptr; int resident = 0;
    const double *rD = nullptr, *rDinv = nullptr, *rE = nullptr, *rEinv = nullptr; double *state_x = nullptr, *state_Qx = nullptr;
    QPDOAmdTraceRec *trace = nullptr; long trace_cap = 0; double *out_x = nullptr, *out_y = nullptr; int mode = 0;
    if constexpr (LAT) {
        Rg = uni_ptr(Pg.res);
        if (Rg) {
            resident = 1;
            rD = uni_ptr(Rg->rD); rDinv = uni_ptr(Rg->rDinv); rE = uni_ptr(Rg->rE); rEinv = uni_ptr(Rg->rEinv);
            state_x = uni_ptr(Rg->state_x); state_Qx = uni_ptr(Rg->state_Qx); trace = uni_ptr(Rg->trace); trace_cap = Rg->trace_cap;
            out_x = uni_ptr(Rg->out_x); out_y = uni_ptr(Rg->out_y);
            mode = __builtin_amdgcn_readfirstlane(Rg->mode);
            // One workspace has the CU to itself: its ~30 work vectors move from global memory (L2-resident, but every barrier that follows a
            // vector update waits for the stores' round trip) into LDS when they fit beside the factor.  Same operations on the same values.
        }
        const unsigned voff = (unsigned)__builtin_amdgcn_readfirstlane((int)(Rg ? Rg->lds_vec_off : Pg.batch_vec_off));
        if (voff) {
            P.nv = uni_ptr((double *)((char *)dyn + voff));
            P.mv = P.nv + (size_t)NV_COUNT * P.n;
            P.iv = (int *)(P.mv + (size_t)MV_COUNT * P.m);
        }
    }
#undef UNI_PTR
    P.c_const = Pg.c_const;
    const int n = P.n, m = P.m;
    // info->setup_time / solve_time / run_time of this item (reference PROFILING build, qpdo.c:79-82,327-329,461-464) and the
    // max_time limit (qpdo.c:441-447): the 100 MHz wall clock, read by one lane and broadcast so that every decision taken from
    // it is uniform over the workgroup.  Time never enters the arithmetic.
    const long long t_begin = wall_clock64();
    const bool timed = st.max_time < 1e19;
    double *NVp[NV_COUNT]; double *V[MV_COUNT];
    for (int i = 0; i < NV_COUNT; i++) NVp[i] = P.nv + (size_t)i * n;
    for (int i = 0; i < MV_COUNT; i++) V[i] = P.mv + (size_t)i * m;
    double *x = NVp[NV_X], *xbar = NVp[NV_XBAR], *Qx = NVp[NV_QX], *Aty = NVp[NV_ATY], *df = NVp[NV_DF], *res_dual = NVp[NV_RD],
           *res_dual_in = NVp[NV_RDI], *rhs = NVp[NV_RHS], *dx = NVp[NV_DX], *Qdx = NVp[NV_QDX], *Atdy = NVp[NV_ATDY], *D = NVp[NV_D],
           *Dinv = NVp[NV_DINV], *tn = NVp[NV_T];
    double *y = V[MV_Y], *ybar = V[MV_YBAR], *Ax = V[MV_AX], *mu = V[MV_MU], *isq = V[MV_ISQ], *w = V[MV_W], *res_prim = V[MV_RP],
           *res_prim_old = V[MV_RPOLD], *res_prim_in = V[MV_RPI], *dy = V[MV_DY], *Adx = V[MV_ADX], *dw = V[MV_DW], *E = V[MV_E],
           *Einv = V[MV_EINV], *ats = V[MV_ATS], *tm = V[MV_T], *dwf = V[MV_DWF];
    int *active = P.iv, *active_old = P.iv + m, *changed = P.iv + 2 * m;
    double *xs = dyn, *colbuf = dyn + n, *tk = dyn + 2 * (size_t)n, *gbuf = dyn + (LAT ? 16 : 8) * (size_t)n;   // dyn .. dyn + 8n: l and l*d of four columns during a factorization; the latency kernel: two such buffers (look-ahead)
    double *d_s = gbuf + (((size_t)(n > m ? n : m) / 4 + 4 + 1) & ~(size_t)1);
    int *rp_s = (int *)(d_s + m);
    double *Klds = (double *)(rp_s + (((size_t)m + 1 + 3) & ~(size_t)3));          // start of U
    double *ls_delta = Klds, *ls_alpha = ls_delta + 2 * (size_t)m;
    size_t np2s = 1; while (np2s < 2 * (size_t)m) np2s <<= 1;                         // the bitonic sort pads 2m to a power of two
    u64 *skey = (u64 *)(ls_alpha + 2 * (size_t)m);
    u32 *sidx = (u32 *)(skey + np2s);
    unsigned char *jflag = (unsigned char *)(sidx + np2s);
    KView kv; kv.n = n; kv.packed = klds_ok; kv.K = klds_ok ? Klds : P.K;
    const int scaled = st.scaling > 0, prox = (int)st.proximal;
    double sc_c = 1.0, sc_cinv = 1.0;

    FOR_T(r, m + 1) rp_s[r] = P.Arp[r];
    small_build_tpos(P);
    // ---- setup: workspace zero + scaling (qpdo.c:49-212) ----
    FOR_T(i, NV_COUNT * n) P.nv[i] = 0.0;
    FOR_T(i, MV_COUNT * m) P.mv[i] = 0.0;
    FOR_T(i, 3 * m) P.iv[i] = 0;
    SYNC;
    if (scaled && !resident) small_scale(P, (int)st.scaling, D, Dinv, E, Einv, tn, tm, sc_c, sc_cinv, sm);
    else if (scaled) {                              // resident: qpdo_setup's scaling (Ruiz + cost, scaling.c:24-91), as installed in the workspace
        FOR_T(j, n) { D[j] = rD[j]; Dinv[j] = rDinv[j]; }
        FOR_T(i, m) { E[i] = rE[i]; Einv[i] = rEinv[i]; }
        sc_c = Rg->r_c; sc_cinv = Rg->r_cinv;
        SYNC;
    }
    const bool ws_scaled = scaled;

    // ---- warm start (qpdo.c:217-299) + initialize_mu (iteration.c:98-122) ----
    double sigma = st.sigma_init;
    if (LAT && mode == 2) {                         // the workspace's state, as qpdo_warm_start (mode 1) and any qpdo_update_* since left it
        // (the descriptor lives in pinned host memory: its pointer fields are read ONCE, not per loop iteration)
        const double *g_xbar = uni_ptr(Rg->st_xbar), *g_Aty = uni_ptr(Rg->st_Aty), *g_y = uni_ptr(Rg->st_y), *g_ybar = uni_ptr(Rg->st_ybar),
                     *g_Ax = uni_ptr(Rg->st_Ax), *g_mu = uni_ptr(Rg->st_mu), *g_isq = uni_ptr(Rg->st_isq);
        FOR_T(j, n) { x[j] = state_x[j]; xbar[j] = g_xbar[j]; Qx[j] = state_Qx[j]; Aty[j] = g_Aty[j]; }
        FOR_T(i, m) { y[i] = g_y[i]; ybar[i] = g_ybar[i]; Ax[i] = g_Ax[i]; mu[i] = g_mu[i]; isq[i] = g_isq[i]; }
        SYNC;
    } else {
    if (P.x0) {
        FOR_T(i, n) { double v = P.x0[i]; if (ws_scaled) v = v * Dinv[i]; x[i] = v; xbar[i] = v; }
        SYNC;
        spmv_rows(n, P.Qrp, P.Qci, P.Qval, x, Qdx);
        spmv_rows(m, P.Arp, P.Aci, P.Aval, x, Ax);
        SYNC;
        FOR_T(i, n) Qx[i] = prox ? Qdx[i] + sigma * x[i] : Qdx[i];
        SYNC;
    }
    if (P.y0) {
        FOR_T(i, m) { double v = P.y0[i]; if (ws_scaled) { v = v * Einv[i]; v = v * sc_c; } y[i] = v; ybar[i] = v; }
        SYNC;
        spmv_rows(n, P.Trp, P.Tci, P.Tval, y, Aty);
        SYNC;
    }
    {
        const double f = 0.5 * dot_seq(x, Qx, n, sm, gbuf) + dot_seq(P.q, x, n, sm, gbuf);
        FOR_T(i, m) {
            const double r = Ax[i] - s_mid(Ax[i], P.l[i], P.u[i]);
            const double v = s_max(1e-3, s_min(1e3, 0.1 * s_max(1, 0.5 * r * r) / s_max(1, s_abs(f))));
            mu[i] = v;
            double s = sqrt(v); isq[i] = 1.0 / s;
        }
        SYNC;
    }
    }
    if (LAT && mode == 1) {                         // an explicit qpdo_warm_start: hand the state to the workspace and stop
        double *g_xbar = uni_ptr(Rg->st_xbar), *g_Aty = uni_ptr(Rg->st_Aty), *g_y = uni_ptr(Rg->st_y), *g_ybar = uni_ptr(Rg->st_ybar),
               *g_Ax = uni_ptr(Rg->st_Ax), *g_mu = uni_ptr(Rg->st_mu), *g_isq = uni_ptr(Rg->st_isq);
        FOR_T(j, n) { state_x[j] = x[j]; g_xbar[j] = xbar[j]; state_Qx[j] = Qx[j]; g_Aty[j] = Aty[j]; }
        FOR_T(i, m) { g_y[i] = y[i]; g_ybar[i] = ybar[i]; g_Ax[i] = Ax[i]; g_mu[i] = mu[i]; g_isq[i] = isq[i]; }
        if (threadIdx.x == 0) {
            double obj = small_objective(n, prox, sigma, Qx, x, P.q);     // qpdo.c:257 (compute_objective on the warm-started x)
            if (scaled) obj *= sc_cinv;
            Rg->ws_objective = obj + P.c_const;
        }
        return;
    }
    const double isq_mu_min = 1 / sqrt(st.mu_min);

    // ---- solve loop (qpdo.c:304-476) ----
    double eps_in = st.eps_abs_in, tau = 0.0, sigma_f = 0.0;
    sigma = st.sigma_init;
    int reset_newton = 1, factor_valid = 0, lds_has_factor = 0, last_branch = -1; double last_sigma_f = -1.0; long nrestore = 0;
    int have_fact = 0; double fact_sigma_f = -1.0;
    long iter = 0, oter = 0, iter_old = 0, status = QPDO_UNSOLVED, newton = 0, nfactor = 0;
    double rpn = 0, rdn = 0, rpin = 0, rdin = 0;
    long long tph = P.prof ? wall_clock64() : 0;
    const long long t_solve = wall_clock64();
    RedBank RB; RB.red = (lds_f64 *)red_scr; RB.rb = 0;
    const long long cyc0 = P.prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    for (iter = 0; iter < st.max_iter; iter++) {
        // outer + inner residuals (iteration.c:30-93) and their four norms: every thread takes the maxima over the entries it has just
        // formed, one reduction with one barrier follows (round 4; was: residuals, barrier, a second pass over them, three more barriers)
        {
            double m1 = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0;
            FOR_T(i, m) {
                const double ax = Ax[i], yi = y[i];
                double t;
                if (scaled) { t = E[i] * yi; t = t * sc_cinv; t = E[i] * t; t = ax + t; } else t = ax + yi;
                const double z = s_mid(t, P.l[i], P.u[i]);
                const double rp = ax - z;
                res_prim[i] = rp;
                const double wi = ax + mu[i] * (ybar[i] - 0.5 * yi);
                w[i] = wi;
                const double zin = s_mid(wi, P.l[i], P.u[i]);
                const double rpi = ax + mu[i] * (ybar[i] - yi) - zin;
                res_prim_in[i] = rpi;
                const double e = scaled ? Einv[i] : 1.0;
                const double a1 = s_abs(scaled ? rp * e : rp), a3 = s_abs(scaled ? rpi * e : rpi);
                m1 = a1 > m1 ? a1 : m1; m3 = a3 > m3 ? a3 : m3;
            }
            FOR_T(j, n) {
                const double df0 = Qx[j] + P.q[j], aty = Aty[j];
                double rd, dfi;
                if (prox) { rd = df0 + (-sigma) * x[j]; rd = rd + aty; dfi = df0 + (-sigma) * xbar[j]; } else { rd = df0 + aty; dfi = df0; }
                const double rdi = dfi + aty;
                df[j] = dfi; res_dual[j] = rd; res_dual_in[j] = rdi;
                const double e = scaled ? Dinv[j] : 1.0;
                const double a2 = s_abs(scaled ? rd * e : rd), a4 = s_abs(scaled ? rdi * e : rdi);
                m2 = a2 > m2 ? a2 : m2; m4 = a4 > m4 ? a4 : m4;
            }
            blk_max4_1b(m1, m2, m3, m4, RB);
            rpn = m1; rdn = m2; rpin = m3; rdin = m4;
            if (scaled) { rdn *= sc_cinv; rdin *= sc_cinv; }
        }
        // per-pass trace record (the fields of the host loop's record, qpdo_api.c; resident mode only): lane 0, into pinned host memory
        QPDOAmdTraceRec *tr = (LAT && trace && iter < trace_cap && threadIdx.x == 0) ? trace + iter : nullptr;
        if (tr) {
            tr->kind = 2; tr->n_active = 0; tr->n_enter = 0; tr->n_leave = 0; tr->factor_branch = -1; tr->lin_iters = 0; tr->tau = 0.0;
            tr->res_prim = rpn; tr->res_dual = rdn; tr->res_prim_in = rpin; tr->res_dual_in = rdin; tr->sigma = sigma; tr->eps_in = eps_in;
        }
        if ((rpn > SM_INFTY) || (rdn > SM_INFTY)) { status = QPDO_NON_CVX; break; }
        if ((rpn <= st.eps_abs) && (rdn <= st.eps_abs)) { status = QPDO_SOLVED; break; }
        const int inner_opt = (rpin <= eps_in) && (rdin <= eps_in);
        PH(PH_RESID);
        if (((iter > iter_old + 1) && inner_opt) || (iter == iter_old + st.inner_max_iter)) {
            if (tr) tr->kind = 1;
            // The outer update in a dozen barrier-separated phases instead of thirty (round 4): both infeasibility tests are evaluated
            // TOGETHER and looked at once -- every quantity of termination.c:97-216 and iteration.c:127-180 with the operations and the
            // summation orders of the step-by-step version (each test in its own block, each norm its own reduction), so the same bits:
            // maxima are order free, the two sequential sums run side by side on two waves, and everything the dual test computes before
            // the primal test has spoken goes to scratch (dxs) until the reference would have written it.
            {
                const bool chk = iter < iter_old + st.inner_max_iter;
                const bool do_p = chk && st.eps_prim_inf > 0, do_d = chk && st.eps_dual_inf > 0;
                const bool do_mu = (oter > 0) && (rpn > st.eps_abs);
                double *dxs = tn;                                   // x - x_bar (the scaling's scratch vector is free after setup)
                double eps_p = 0.0, eps_d = 0.0, rn = 0.0;
                {   // dy = y - y_bar, dx = x - x_bar with || E dy ||, || D dx || taken as they are formed, and || res_prim || (update_mu's own
                    // norm, iteration.c:130): one reduction, one barrier
                    double a = 0.0, bq = 0.0, c = 0.0, d4 = 0.0;
                    if (do_p) { FOR_T(i, m) { const double dv = y[i] - ybar[i]; dy[i] = dv; const double v = s_abs(scaled ? dv * E[i] : dv); a = v > a ? v : a; } }
                    if (do_d) { FOR_T(j, n) { const double dv = x[j] - xbar[j]; dxs[j] = dv; const double v = s_abs(scaled ? dv * D[j] : dv); bq = v > bq ? v : bq; } }
                    if (do_mu) { FOR_T(k, m) { const double v = s_abs(res_prim[k]); c = v > c ? v : c; } }
                    blk_max4_1b(a, bq, c, d4, RB);
                    eps_p = st.eps_prim_inf * a; eps_d = st.eps_dual_inf * bq; rn = c;
                }
                const bool act_p = do_p && eps_p != 0, act_d = do_d && eps_d != 0;
                // A' dy, Q dx, A dx: the products of all three staged in ONE phase when the union region holds them (spmv_rows_staged)
                {
                    const int nzT = __builtin_amdgcn_readfirstlane(P.Trp[n]), nzQ = __builtin_amdgcn_readfirstlane(P.Qrp[n]), nzA = __builtin_amdgcn_readfirstlane(rp_s[m]);
                    if (act_p && act_d && nzT + nzQ + nzA <= ucap) {
                        double *s1 = Klds, *s2 = s1 + nzT, *s3 = s2 + nzQ;
                        FOR_T(k, nzT) s1[k] = P.Tval[k] * dy[P.Tci[k]];
                        FOR_T(k, nzQ) s2[k] = P.Qval[k] * dxs[P.Qci[k]];
                        FOR_T(k, nzA) s3[k] = P.Aval[k] * dxs[P.Aci[k]];
                        SYNC;
                        const lds_f64 *l1 = (const lds_f64 *)s1, *l2 = (const lds_f64 *)s2, *l3 = (const lds_f64 *)s3;
                        FOR_T(r, n) { const int k0 = P.Trp[r]; Atdy[r] = seq_fold(l1 + k0, P.Trp[r + 1] - k0); }
                        FOR_T(r, n) { const int k0 = P.Qrp[r]; Qdx[r] = seq_fold(l2 + k0, P.Qrp[r + 1] - k0); }
                        FOR_T(r, m) { const int k0 = rp_s[r]; Adx[r] = seq_fold(l3 + k0, rp_s[r + 1] - k0); }
                        SYNC;                                     // (the staged products sit where the bound terms are written next)
                    } else {
                        if (act_p) { spmv_rows_staged(n, P.Trp, P.Tci, P.Tval, dy, Atdy, Klds, ucap); SYNC; }
                        if (act_d) { spmv2_rows_staged(n, P.Qrp, P.Qci, P.Qval, Qdx, m, rp_s, P.Aci, P.Aval, Adx, dxs, Klds, ucap); SYNC; }
                    }
                }
                double mx_at = 0.0, mx_q = 0.0, viol_f = 0.0;
                if (act_p) {
                    FOR_T(j, n) { double v = Atdy[j]; if (scaled) { v = Dinv[j] * v; Atdy[j] = v; } v = s_abs(v); mx_at = v > mx_at ? v : mx_at; }
                    FOR_T(i, m) {                                   // the 2m bound terms, added in the reference's order below
                        const double e = scaled ? E[i] : 1.0;
                        ls_delta[2 * i] = (P.u[i] < e * SM_INFTY) ? P.u[i] * s_max(dy[i], 0) : 0;
                        ls_delta[2 * i + 1] = (P.l[i] > -e * SM_INFTY) ? P.l[i] * s_min(dy[i], 0) : 0;
                    }
                }
                const int ngq = n >> 2, cntq = ngq + (n - 4 * ngq);
                if (act_d) {
                    FOR_T(k, m) {
                        double v = Adx[k]; const double e = scaled ? E[k] : 1.0;
                        if (scaled) { v = Einv[k] * v; Adx[k] = v; }
                        if ((P.u[k] < e * SM_INFTY && v >= eps_d) || (P.l[k] > -e * SM_INFTY && v <= -eps_d)) viol_f = 1.0;
                    }
                    FOR_T(j, n) { double qv = Qdx[j]; if (prox) qv = qv + (-sigma * tau) * dxs[j]; Qdx[j] = qv; qv = s_abs(qv); mx_q = qv > mx_q ? qv : mx_q; }
                    // q'dx: the 4-groups of lin_alg.c:59-71 in parallel, added in order below
                    FOR_T(g, ngq) { const int i = 4 * g; gbuf[g] = (P.q[i] * dxs[i] + P.q[i + 1] * dxs[i + 1] + P.q[i + 2] * dxs[i + 2] + P.q[i + 3] * dxs[i + 3]); }
                    FOR_T(t, n - 4 * ngq) gbuf[ngq + t] = P.q[4 * ngq + t] * dxs[4 * ngq + t];
                }
                // the three maxima of this phase are posted before the barrier that publishes the bound terms and read after the one that
                // publishes the two sequential sums: the sum of the bound terms on wave 0 -- only the NONZERO ones, at most one per row with
                // dy != 0, moved to the front in order (wave_fold_nonzero; the fold over all 2m took 9 us) -- and q'dx on wave 1
                lds_f64 *Rm = red_next(RB);
                max4_post(mx_at, mx_q, viol_f, 0.0, Rm);
                SYNC;
                if (act_p && threadIdx.x < 64) { const double oobs = wave_fold_nonzero((lds_f64 *)ls_delta, 2 * m); if (threadIdx.x == 0) sm[18] = oobs; }
                if (act_d && threadIdx.x >= 64 && threadIdx.x < 128) { const double prod = seq_fold((const lds_f64 *)gbuf, cntq); if (threadIdx.x == 64) sm[19] = prod; }
                SYNC;
                const double oob = sm[18], qdx = sm[19];
                double nat, nq, viol_m, d4;
                max4_read(nat, nq, viol_m, d4, Rm);
                if (act_p && (nat <= eps_p) && (oob <= -eps_p)) {
                    status = QPDO_PRIMAL_INFEASIBLE;
                    if (scaled) { FOR_T(i, m) { double v = dy[i] * sc_cinv; dy[i] = E[i] * v; } }
                    SYNC;
                    break;
                }
                if (do_d) { FOR_T(j, n) dx[j] = dxs[j]; }            // (the reference's dx from here on: termination.c:166)
                if (act_d && viol_m == 0.0) {
                    const double cc = scaled ? sc_c : 1.0;
                    if ((nq <= cc * eps_d) && (qdx <= -cc * eps_d)) {
                        status = QPDO_DUAL_INFEASIBLE;
                        SYNC;
                        if (scaled) { FOR_T(j, n) dx[j] = D[j] * dx[j]; }
                        SYNC;
                        break;
                    }
                }
                // shift the estimates, update_mu (iteration.c:127-168), update_sigma (:173-180), keep res_prim: one elementwise phase
                const double sigma_before = sigma;
                const bool sig_upd = prox && (oter > 0) && (rdn > st.eps_abs) && (sigma > st.sigma_min);
                if (sig_upd) sigma = s_max(sigma * st.sigma_upd, st.sigma_min);
                FOR_T(j, n) { xbar[j] = x[j]; if (sig_upd) Qx[j] = Qx[j] + (sigma - sigma_before) * x[j]; }
                int cnt = 0;
                FOR_T(k, m) {
                    ybar[k] = y[k];
                    if (do_mu) {
                        int ch = 0;
                        if (s_abs(res_prim[k]) > s_max(st.eps_abs, st.theta * s_abs(res_prim_old[k]))) {
                            double mu_factor = 1.0 / s_min(1.0, st.delta * rn / s_abs(res_prim[k]));
                            const double mu_new = mu[k] / mu_factor;
                            if (mu_new >= st.mu_min) {
                                if (mu[k] != mu_new) ch = 1;
                                mu[k] = mu_new; mu_factor = sqrt(mu_factor); isq[k] = mu_factor * isq[k]; ats[k] = mu_factor;
                            } else {
                                if (mu[k] != st.mu_min) ch = 1;
                                mu[k] = st.mu_min; ats[k] = isq_mu_min / isq[k]; isq[k] = isq_mu_min;
                            }
                        } else ats[k] = 1.0;
                        changed[k] = ch; cnt += ch;
                    }
                    res_prim_old[k] = res_prim[k];
                }
                if (do_mu) {
                    { int c2 = 0, c3 = 0; blk_sum_int3_1b(cnt, c2, c3, RB); }
                    if ((prox && sigma_before > st.sigma_min) || (cnt > 0.25 * SM_MAX_RANK_UPDATE)) reset_newton = 1;
                    else if (cnt > 0) {
                        FOR_T(k, m) if (changed[k]) { const double s = sqrt(1 - 1 / (ats[k] * ats[k])); const double col = isq[k] * s; dw[k] += col * col; }
                        factor_valid = 0;
                    }
                }
                if (sig_upd) reset_newton = 1;
                SYNC;
                if (chk) eps_in = s_max(st.rho * eps_in, 0.1 * st.eps_abs);
            }
            oter++; iter_old = iter;
            PH(PH_OUTER);
        } else {
            if (st.reset_newton_iter > 0 && (iter % st.reset_newton_iter == 0)) reset_newton = 1;
            // active set, enter / leave (newton.c:96-126)
            int na = 0, ne = 0, nl = 0;
            FOR_T(i, m) {
                const int act = (w[i] <= P.l[i]) || (w[i] >= P.u[i]);
                active[i] = act; na += act; ne += (act && !active_old[i]); nl += (!act && active_old[i]);
            }
            blk_sum_int3_1b(na, ne, nl, RB);
            int branch;
            if ((reset_newton && na) || (ne + nl) > SM_MAX_RANK_UPDATE) { reset_newton = 0; branch = 0; }
            else if (na) branch = 1; else branch = 2;
            if (branch == 0 || branch == 2) sigma_f = prox ? sigma : 0.0;
            FOR_T(i, m) {
                const double wgt = isq[i] * isq[i];
                if (branch == 0) dw[i] = active[i] ? wgt : 0.0;
                else if (branch == 1) { if (active[i] && !active_old[i]) dw[i] += wgt; else if (!active[i] && active_old[i]) dw[i] -= wgt; }
                else dw[i] = 0.0;
                double t = res_prim_in[i] / mu[i];
                if (!active[i]) t *= 2;
                dy[i] = t;
            }
            if (branch == 0) factor_valid = 0;
            else if (branch == 1) { if (ne + nl > 0) factor_valid = 0; }
            else if (!(last_branch == 2 && last_sigma_f == sigma_f)) factor_valid = 0;
            SYNC;
            spmv_rows_staged(n, P.Trp, P.Tci, P.Tval, dy, Atdy, Klds, ucap);
            FOR_T(j, n) rhs[j] = -res_dual_in[j] - Atdy[j];      // (entry j of the product was written by this very thread: no barrier between)
            SYNC;
            PH(PH_PREP);
            // The factor is a function of (sigma_f, d) alone.  The reference refactors whenever reset_newton is set -- after EVERY outer
            // update while sigma > sigma_min (iteration.c:161-162) -- even when neither the weights nor sigma moved; an instance
            // that stalls just above eps_abs alternates Newton pass / outer update like that up to max_iter, each time factoring
            // the matrix it factored before.  If the weights and sigma_f carry the very bits of the last factorization, its
            // result is reused: the same bits as factoring again.
            if (!factor_valid && have_fact && sigma_f == fact_sigma_f) {
                int diff = 0;
                FOR_T(i, m) diff |= (__double_as_longlong(dw[i]) != __double_as_longlong(dwf[i]));
                { int c2 = 0, c3 = 0; blk_sum_int3_1b(diff, c2, c3, RB); }
                if (!diff) factor_valid = 1;
            }
            if (!factor_valid) {
                FOR_T(i, m) dwf[i] = dw[i];
                fact_sigma_f = sigma_f; have_fact = 1;
                small_assemble(P, kv, dw, sigma_f, rp_s, d_s); PH(PH_ASM);
                if (klds_ok) {
                    // look-ahead in the latency kernel only (80 -> 71 us at n = 120); in the wide kernel, two workgroups per CU at 128 VGPRs, it gains
                    // nothing at max_iter 300 and its registers cost the stalled passes 5 % (measured both ways)
                    if constexpr (LAT) small_factor4_la<1>(n, Klds, dyn); else small_factor4_t<true>(n, Klds, dyn);
                } else small_factor4_t<false>(n, P.K, dyn);
                PH(PH_FACTOR); factor_valid = 1; nfactor++;
                // The factor shares its LDS region with the linesearch scratch, so it does not survive the pass.  A copy in the
                // item's global K buffer (58 KB at n = 120, L2-resident) lets the next passes RESTORE it while (sigma_f, d) stay
                // what they are -- the reference's factor reuse (newton.c:21-33: nothing to do when no row enters or leaves) --
                // instead of re-assembling and re-factoring to the same bits.  An instance that crawls towards eps for thousands
                // of passes with a fixed active set spends most of its time there.
                if (klds_ok) { const int tot = n * (n + 1) / 2; FOR_T(i, tot) P.K[i] = Klds[i]; lds_has_factor = 1; SYNC; }
            } else if (klds_ok && !lds_has_factor) {
                const int tot = n * (n + 1) / 2; FOR_T(i, tot) Klds[i] = P.K[i]; lds_has_factor = 1; nrestore++; SYNC;
                PH(PH_FACTOR);
            }
            last_branch = branch; last_sigma_f = sigma_f;
            small_ldl_solve(P, kv, rhs, dx, xs);
            PH(PH_SOLVE);
            spmv2_rows_staged(n, P.Qrp, P.Qci, P.Qval, Qdx, m, rp_s, P.Aci, P.Aval, Adx, dx, Klds, ucap);
            if (prox) { FOR_T(j, n) Qdx[j] = Qdx[j] + sigma * dx[j]; }      // (own entries of both products: no barrier between)
            FOR_T(i, m) { if (active[i]) dy[i] += (Adx[i] / mu[i]); active_old[i] = active[i]; }
            SYNC;
            spmv_rows_staged(n, P.Trp, P.Tci, P.Tval, dy, Atdy, Klds, ucap);
            SYNC;
            PH(PH_SPMV);
            if constexpr (LAT) tau = small_linesearch_impl(P, V, ls_delta, ls_alpha, jflag, tm, sm, skey, sidx, gbuf, dyn);
            else tau = small_linesearch(P, V, ls_delta, ls_alpha, jflag, tm, sm, skey, sidx, gbuf, dyn);
            lds_has_factor = 0;                      // the scratch above lives in the factor's LDS region
            PH(PH_LS);
            FOR_T(j, n) { x[j] = x[j] + tau * dx[j]; Qx[j] = Qx[j] + tau * Qdx[j]; Aty[j] = Aty[j] + tau * Atdy[j]; }
            FOR_T(i, m) { y[i] = y[i] + tau * dy[i]; Ax[i] = Ax[i] + tau * Adx[i]; }
            SYNC;
            newton++;
            if (tr) { tr->kind = 0; tr->n_active = na; tr->n_enter = ne; tr->n_leave = nl; tr->factor_branch = branch; tr->tau = tau; }
            PH(PH_UPDATE);
        }
        if (timed) {                                  // qpdo.c:441-447: checked at the end of every pass, iter is not advanced
            SYNC;
            if (threadIdx.x == 0) sm[18] = (double)(wall_clock64() - t_begin) * 1e-8;
            SYNC;
            if (sm[18] > st.max_time) { status = QPDO_MAX_TIME_REACHED; break; }
        }
    }
    if (status == QPDO_UNSOLVED) status = QPDO_MAX_ITER_REACHED;
    if (P.prof && threadIdx.x == 0) { P.prof[PH_CYC] = (long long)__builtin_amdgcn_s_memtime() - cyc0; P.prof[PH_WALL] = wall_clock64() - t_solve; }
    // store_solution (termination.c:82-92) + objective (iteration.c:185-221)
    FOR_T(j, n) P.sol_x[j] = scaled ? x[j] * D[j] : x[j];
    FOR_T(i, m) { if (scaled) { const double v = y[i] * sc_cinv; y[i] = v; P.sol_y[i] = v * E[i]; } else P.sol_y[i] = y[i]; }
    FOR_T(j, n) P.cert_dx[j] = dx[j];
    FOR_T(i, m) P.cert_dy[i] = dy[i];
    if (LAT && state_x) { FOR_T(j, n) { state_x[j] = x[j]; state_Qx[j] = Qx[j]; out_x[j] = x[j]; } FOR_T(i, m) out_y[i] = y[i]; }
    SYNC;
    if (threadIdx.x == 0) {
        double obj = small_objective(n, prox, sigma, Qx, x, P.q);
        if (scaled) obj *= sc_cinv;
        obj += P.c_const;
        Pg.info.iterations = iter; Pg.info.oterations = oter;
        Pg.info.res_prim_norm = rpn; Pg.info.res_dual_norm = rdn; Pg.info.res_prim_in_norm = rpin; Pg.info.res_dual_in_norm = rdin;
        Pg.info.objective = obj;
        const long long t_end = wall_clock64();
        Pg.info.setup_time = (double)(t_solve - t_begin) * 1e-8;          // scaling + warm start + initialize_mu of this item
        Pg.info.solve_time = (double)(t_end - t_solve) * 1e-8;
        Pg.info.run_time = Pg.info.setup_time + Pg.info.solve_time;
        small_status(Pg.info, status);
        Pg.newton_passes = newton; Pg.factor_count = nfactor; (void)nrestore;
        if constexpr (LAT) {
            if (Rg) {
                Rg->sigma_end = sigma; Rg->tau_end = tau;
                Rg->ntrace = (status == QPDO_MAX_ITER_REACHED) ? iter : iter + 1;     // a pass that ended the loop by `break` left a record too
            }
        }
    }
}
__global__ __launch_bounds__(SM_THREADS, 4) void k_small_solve(SmallQP *probs, int count, QPDOSettings st, int klds_ok) {
    small_solve_body<0>(probs, count, st, klds_ok);
}
__global__ __launch_bounds__(SM_THREADS, 2) void k_small_solve_lat(SmallQP *probs, int count, QPDOSettings st, int klds_ok) {
    small_solve_body<1>(probs, count, st, klds_ok);
}

// ================================================================================================
// host side: pack a batch into one arena, one upload, one launch, one download
// ================================================================================================
static thread_local char s_err[256] = "";
#define SHIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { snprintf(s_err, sizeof(s_err), "%s: %s", #call, hipGetErrorString(e__)); rc = -1; goto done; } } while (0)

static inline long long idx_at(const void *a, int itype, long long k) { return itype == 0 ? (long long)((const int *)a)[k] : (long long)((const long long *)a)[k]; }
// CSC -> the kernel's CSR images (A, A' = the CSC arrays narrowed, the full symmetric Q) straight into caller-provided arrays (the pinned
// staging buffer of a batch): no per-item vectors, no second copy.  `next` / the temporaries are per-thread scratch that only grows.
// Row i of the full Q = (lower stored) CSR row i of the stored triangle [columns <= i] followed by the stored column i below the
// diagonal, (upper stored) the stored column i above the diagonal followed by CSR row i [columns >= i]; same rule as the host driver's
// sym_to_full_csr (qpdo_api.c).
static void csc_to_csr32_raw(const cholmod_sparse *M, int *rp, int *ci, double *val, std::vector<int> &next) {
    const long long nr = (long long)M->nrow, nc = (long long)M->ncol, nnz = idx_at(M->p, M->itype, nc);
    for (long long i = 0; i <= nr; i++) rp[i] = 0;
    for (long long k = 0; k < nnz; k++) rp[idx_at(M->i, M->itype, k) + 1]++;
    for (long long i = 0; i < nr; i++) rp[i + 1] += rp[i];
    if ((long long)next.size() < nr + 1) next.resize((size_t)nr + 1);
    for (long long i = 0; i < nr; i++) next[(size_t)i] = rp[i];
    const double *x = (const double *)M->x;
    for (long long j = 0; j < nc; j++)
        for (long long k = idx_at(M->p, M->itype, j); k < idx_at(M->p, M->itype, j + 1); k++) { const int s2 = next[(size_t)idx_at(M->i, M->itype, k)]++; ci[s2] = (int)j; val[s2] = x[k]; }
}
static void csc_as_csrT32_raw(const cholmod_sparse *M, int *rp, int *ci, double *val) {
    const long long nc = (long long)M->ncol, nnz = idx_at(M->p, M->itype, nc);
    for (long long j = 0; j <= nc; j++) rp[j] = (int)idx_at(M->p, M->itype, j);
    for (long long k = 0; k < nnz; k++) ci[k] = (int)idx_at(M->i, M->itype, k);
    if (nnz) memcpy(val, M->x, (size_t)nnz * 8);
}
static long long sym_full_nnz(const cholmod_sparse *Q) {              // entries of the full symmetric matrix (sym_full32's count)
    const long long n = (long long)Q->ncol, nnz = idx_at(Q->p, Q->itype, n);
    const int st = Q->stype;
    if (st == 0) return nnz;
    long long c = 0;
    for (long long j = 0; j < n; j++)
        for (long long k = idx_at(Q->p, Q->itype, j); k < idx_at(Q->p, Q->itype, j + 1); k++) {
            const long long i = idx_at(Q->i, Q->itype, k);
            if (st < 0) c += (i >= j) + (i > j); else c += (i <= j) + (i < j);      // the stored triangle once, its strict part mirrored
        }
    return c;
}
struct ConvScratch { std::vector<int> next, Rrp, Rci; std::vector<double> Rval; };
static void sym_full32_raw(const cholmod_sparse *Q, int *rp, int *ci, double *val, ConvScratch &W) {
    const int st = Q->stype;
    if (st == 0) { csc_as_csrT32_raw(Q, rp, ci, val); return; }
    const long long n = (long long)Q->ncol, nnzs = idx_at(Q->p, Q->itype, n);
    if ((long long)W.Rrp.size() < n + 1) W.Rrp.resize((size_t)n + 1);
    if ((long long)W.Rci.size() < nnzs + 1) { W.Rci.resize((size_t)nnzs + 1); W.Rval.resize((size_t)nnzs + 1); }
    csc_to_csr32_raw(Q, W.Rrp.data(), W.Rci.data(), W.Rval.data(), W.next);
    const int *Rrp = W.Rrp.data(), *Rci = W.Rci.data(); const double *Rval = W.Rval.data();
    const double *x = (const double *)Q->x;
    auto keep_csr = [&](long long i, long long j) { return st < 0 ? j <= i : j >= i; };
    auto keep_mir = [&](long long i, long long j) { return st < 0 ? i > j : i < j; };
    int s2 = 0;
    for (long long i = 0; i < n; i++) {                                  // (the rule and the order of sym_full32)
        rp[i] = s2;
        const long long b0 = idx_at(Q->p, Q->itype, i), e0 = idx_at(Q->p, Q->itype, i + 1);
        if (st < 0) {
            for (int k = Rrp[i]; k < Rrp[i + 1]; k++) if (keep_csr(i, Rci[k])) { ci[s2] = Rci[k]; val[s2] = Rval[k]; s2++; }
            for (long long k = b0; k < e0; k++) { const long long r = idx_at(Q->i, Q->itype, k); if (keep_mir(r, i)) { ci[s2] = (int)r; val[s2] = x[k]; s2++; } }
        } else {
            for (long long k = b0; k < e0; k++) { const long long r = idx_at(Q->i, Q->itype, k); if (keep_mir(r, i)) { ci[s2] = (int)r; val[s2] = x[k]; s2++; } }
            for (int k = Rrp[i]; k < Rrp[i + 1]; k++) if (keep_csr(i, Rci[k])) { ci[s2] = Rci[k]; val[s2] = Rval[k]; s2++; }
        }
    }
    rp[n] = s2;
}
// One in-flight batch owns a SLOT: a stream, a device arena, pinned host staging for the inputs, the outputs and the per-item
// control structs, and the per-item conversion buffers -- all grow-only and kept between batches (hipMalloc/hipFree of ~0.5 GB
// cost ~0.1 s per batch; releasing the host buffers another 0.1 s).  qdev_small_batch uses one static slot; a batch STREAM
// (qdev_small_stream_*) owns `depth` slots, so that batch i+1 is packed, uploaded and started while the slowest workgroups of
// batch i still run: a launch is as slow as its slowest item (an instance that never reaches eps runs max_iter passes on one
// workgroup), and with one batch at a time the other CUs idle behind it.
struct Lay { size_t Arp, Aci, Aval, Trp, Tci, Tval, Qrp, Qci, Qval, q, l, u, x0, y0, nv, mv, lsv, iv, tpos, K, solx, soly, dx, dy; size_t nnzA, nnzQ; };
struct SmallSlot {
    int device = -1;
    hipStream_t stream = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
    char *arena = nullptr; size_t arena_cap = 0;
    SmallQP *dprobs = nullptr; size_t dprobs_cap = 0;
    char *hin = nullptr, *hout = nullptr; size_t hin_cap = 0, hout_cap = 0;      // pinned: the copies must not block the host
    SmallQP *hp = nullptr; size_t hp_cap = 0;                                   // pinned image of the per-item structs
    long long *dprof = nullptr;
    std::vector<Lay> lay;
    // the batch in flight
    long count = 0; QPDOAmdBatchItem *items = nullptr; size_t upload_bytes = 0, out_bytes = 0; bool busy = false; long ticket = -1;
    double kernel_s = 0.0;
};
// run f(i) for i in [0, count) on up to 16 host threads
template <class F>
static void parallel_items(long count, F f) {
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 4;
    long T = hw > 16 ? 16 : (long)hw;
    if (const char *e = getenv("QPDO_SETUP_THREADS")) { const long v = atol(e); if (v > 0) T = v; }
    if (T > count / 32 + 1) T = count / 32 + 1;
    if (T <= 1) { for (long i = 0; i < count; i++) f(i); return; }
    std::vector<std::thread> th;
    for (long t = 0; t < T; t++) th.emplace_back([=]() { for (long i = t; i < count; i += T) f(i); });
    for (auto &x : th) x.join();
}
static void slot_release(SmallSlot &S) {
    if (S.device >= 0) (void)hipSetDevice(S.device);
    if (S.stream) (void)hipStreamSynchronize(S.stream);
    if (S.dprobs) (void)hipFree(S.dprobs);
    if (S.dprof) (void)hipFree(S.dprof);
    if (S.arena) (void)hipFree(S.arena);
    if (S.hin) (void)hipHostFree(S.hin);
    if (S.hout) (void)hipHostFree(S.hout);
    if (S.hp) (void)hipHostFree(S.hp);
    if (S.ev0) (void)hipEventDestroy(S.ev0);
    if (S.ev1) (void)hipEventDestroy(S.ev1);
    if (S.stream) (void)hipStreamDestroy(S.stream);
    S = SmallSlot();
}
static int pinned_reserve(char **buf, size_t *cap, size_t need) {
    if (*cap >= need && *buf) return 0;
    if (*buf) (void)hipHostFree(*buf);
    *buf = nullptr; *cap = 0;
    const size_t want = need + need / 8 + 4096;
    if (hipHostMalloc((void **)buf, want, hipHostMallocDefault) != hipSuccess) { *buf = nullptr; return -1; }
    *cap = want;
    return 0;
}
// dynamic LDS of a workgroup whose largest item has (nmax, mmax): the fixed part + the union region U (packed factor K if it fits,
// else the linesearch scratch alone; the kernel's layout, k_small_solve).  klds_ok (optional out): 1 when K lives in LDS; passing
// NULL sizes the K-in-global-memory layout.
static const size_t SMALL_LDS_BUDGET = 160 * 1024 - 1024;                // static LDS: reduction scratch only
static size_t small_lds_bytes(size_t nmax, size_t mmax, int *klds_ok, size_t *union_bytes = nullptr, bool lat = false) {
    size_t lds = (lat ? 16 : 8) * nmax * 8 + ((((nmax > mmax ? nmax : mmax) / 4 + 4 + 1) & ~(size_t)1) * 8) + mmax * 8 + (((mmax + 1 + 3) & ~(size_t)3) * 4);
    const size_t kbytes = nmax * (nmax + 1) / 2 * 8;
    size_t np2 = 1; while (np2 < 2 * mmax) np2 <<= 1;
    size_t lsbytes = 2 * mmax * (8 + 8) + np2 * (8 + 4) + ((2 * mmax + 15) & ~(size_t)15);   // delta, alpha, keys, indices, flags
    { const size_t gdots = 8 * (2 * (mmax / 4 + 4) + 2 * (nmax / 4 + 4)); if (gdots > lsbytes) lsbytes = gdots; }   // the group arrays of the four dot products live there too
    const int ok = klds_ok && (lds + (kbytes > lsbytes ? kbytes : lsbytes) <= SMALL_LDS_BUDGET);
    if (klds_ok) *klds_ok = ok;
    const size_t ub = (ok && kbytes > lsbytes) ? kbytes : lsbytes;
    if (union_bytes) *union_bytes = ub;
    return lds + ub;
}
static volatile double s_last_kernel_s = 0.0;   // (a statistic: written by whichever batch finished last; an aligned 8-byte store)
//          // HIP-event duration of the last finished k_small_solve launch (bench.py's latency statement)

// pack the batch into the slot's staging buffer, upload, launch, enqueue the downloads: returns without waiting for the GPU
static int slot_submit(SmallSlot &S, int device, long count, QPDOAmdBatchItem *items, const QPDOSettings *settings, bool one_at_a_time) {
    int rc = 0;
    const bool tprof = getenv("QPDO_SMALL_PROF") && !strcmp(getenv("QPDO_SMALL_PROF"), "2");
    auto now = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    double tp0 = now();
    auto lap = [&](const char *what) { if (tprof) { const double t = now(); fprintf(stderr, "[qpdo_small host] %-22s %.3f s\n", what, t - tp0); tp0 = t; } };
    size_t total = 0;
    auto reserve = [&](size_t bytes) { size_t o = total; total += (bytes + 255) & ~(size_t)255; return o; };
    if (S.device >= 0 && S.device != device) slot_release(S);
    S.device = device;
    if (S.lay.size() < (size_t)count) S.lay.resize((size_t)count);
    std::vector<Lay> &lay = S.lay;
    // device arena: [inputs of all items][outputs of all items][scratch]; only the inputs are uploaded and only the
    // outputs come back.  The per-item conversions and the copies into the staging buffer run on host threads.
    parallel_items(count, [&](long i) { const QPDOData *d = items[i].data; Lay &L = lay[(size_t)i]; L.nnzA = (size_t)idx_at(d->A->p, d->A->itype, (long long)d->A->ncol); L.nnzQ = (size_t)sym_full_nnz(d->Q); });
    for (long i = 0; i < count; i++) {
        const QPDOData *d = items[i].data; Lay &L = lay[(size_t)i];
        const size_t n = d->n, m = d->m;
        L.Arp = reserve((m + 1) * 4); L.Aci = reserve(L.nnzA * 4 + 4); L.Aval = reserve(L.nnzA * 8 + 8);
        L.Trp = reserve((n + 1) * 4); L.Tci = reserve(L.nnzA * 4 + 4); L.Tval = reserve(L.nnzA * 8 + 8);
        L.Qrp = reserve((n + 1) * 4); L.Qci = reserve(L.nnzQ * 4 + 4); L.Qval = reserve(L.nnzQ * 8 + 8);
        L.q = reserve(n * 8); L.l = reserve(m * 8 + 8); L.u = reserve(m * 8 + 8);
        L.x0 = items[i].x0 ? reserve(n * 8) : (size_t)-1; L.y0 = items[i].y0 ? reserve(m * 8 + 8) : (size_t)-1;
    }
    lap("conversions");
    const size_t upload_bytes = total;       // inputs
    for (long i = 0; i < count; i++) {
        const QPDOData *d = items[i].data; Lay &L = lay[(size_t)i];
        const size_t n = d->n, m = d->m;
        L.solx = reserve(n * 8); L.soly = reserve(m * 8 + 8); L.dx = reserve(n * 8); L.dy = reserve(m * 8 + 8);
    }
    const size_t out_bytes = total - upload_bytes;
    for (long i = 0; i < count; i++) {
        const QPDOData *d = items[i].data; Lay &L = lay[(size_t)i];
        const size_t n = d->n, m = d->m;
        L.nv = reserve((size_t)NV_COUNT * n * 8); L.mv = reserve((size_t)MV_COUNT * m * 8 + 8); L.lsv = reserve(4 * m * 8 + 8);
        L.iv = reserve(3 * m * 4 + 4); L.tpos = reserve(L.nnzA * 4 + 4); L.K = reserve(n * n * 8);
    }
    char *harena = nullptr, *dbase = nullptr; SmallQP *hp = nullptr;
    SHIP(hipSetDevice(device));
    if (pinned_reserve(&S.hin, &S.hin_cap, upload_bytes ? upload_bytes : 1) || pinned_reserve(&S.hout, &S.hout_cap, out_bytes ? out_bytes : 1) ||
        pinned_reserve((char **)&S.hp, &S.hp_cap, (size_t)count * sizeof(SmallQP))) { snprintf(s_err, sizeof(s_err), "pinned host staging allocation failed"); return -1; }
    harena = S.hin; hp = S.hp;
    if (!S.stream) SHIP(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    if (!S.ev0) { SHIP(hipEventCreate(&S.ev0)); SHIP(hipEventCreate(&S.ev1)); }
    if (S.arena && S.arena_cap < total) { (void)hipFree(S.arena); S.arena = nullptr; S.arena_cap = 0; }
    if (!S.arena) { SHIP(hipMalloc((void **)&S.arena, total)); S.arena_cap = total; }
    if (S.dprobs && S.dprobs_cap < (size_t)count) { (void)hipFree(S.dprobs); S.dprobs = nullptr; S.dprobs_cap = 0; }
    if (!S.dprobs) { SHIP(hipMalloc((void **)&S.dprobs, (size_t)count * sizeof(SmallQP))); S.dprobs_cap = (size_t)count; }
    dbase = S.arena;
    lap("hipMalloc");
    {   // staging fill and upload in eight slices of the batch: the copy engine moves slice g while the host threads pack slice g + 1
        // (0.5 GB at the C3 batch: ~10 ms of PCIe that used to start only when the whole staging buffer was packed)
        const long G = count >= 1024 ? 8 : 1;
        for (long g = 0; g < G; g++) {
            const long i0 = count * g / G, i1 = count * (g + 1) / G;
            if (i1 <= i0) continue;
            parallel_items(i1 - i0, [&](long k) {
                const long i = i0 + k;
                const QPDOData *d = items[i].data; Lay &L = lay[(size_t)i];
                const size_t n = d->n, m = d->m;
                char *h = harena;
                const size_t reg_end = (i + 1 < count) ? lay[(size_t)i + 1].Arp : upload_bytes;     // this item's input region, padding included
                // conversions straight into the staging buffer; the slack behind every array (4 / 8 bytes + alignment) is zeroed like the
                // memset of the whole region used to do it
                static thread_local ConvScratch W;
                auto tail0 = [&](size_t off, size_t bytes, size_t next_off) { if (next_off > off + bytes) memset(h + off + bytes, 0, next_off - off - bytes); };
                csc_to_csr32_raw(d->A, (int *)(h + L.Arp), (int *)(h + L.Aci), (double *)(h + L.Aval), W.next);
                tail0(L.Arp, (m + 1) * 4, L.Aci); tail0(L.Aci, L.nnzA * 4, L.Aval); tail0(L.Aval, L.nnzA * 8, L.Trp);
                csc_as_csrT32_raw(d->A, (int *)(h + L.Trp), (int *)(h + L.Tci), (double *)(h + L.Tval));
                tail0(L.Trp, (n + 1) * 4, L.Tci); tail0(L.Tci, L.nnzA * 4, L.Tval); tail0(L.Tval, L.nnzA * 8, L.Qrp);
                sym_full32_raw(d->Q, (int *)(h + L.Qrp), (int *)(h + L.Qci), (double *)(h + L.Qval), W);
                tail0(L.Qrp, (n + 1) * 4, L.Qci); tail0(L.Qci, L.nnzQ * 4, L.Qval); tail0(L.Qval, L.nnzQ * 8, L.q);
                tail0(L.q, n * 8, L.l); tail0(L.l, m * 8, L.u);
                {   // u, then the optional x0 / y0, up to the end of this item's region
                    size_t off = L.u, bytes = m * 8;
                    if (L.x0 != (size_t)-1) { tail0(off, bytes, L.x0); off = L.x0; bytes = n * 8; }
                    if (L.y0 != (size_t)-1) { tail0(off, bytes, L.y0); off = L.y0; bytes = m * 8; }
                    tail0(off, bytes, reg_end);
                }
                memcpy(h + L.q, d->q, n * 8); if (m) { memcpy(h + L.l, d->l, m * 8); memcpy(h + L.u, d->u, m * 8); }
                if (items[i].x0) memcpy(h + L.x0, items[i].x0, n * 8);
                if (items[i].y0 && m) memcpy(h + L.y0, items[i].y0, m * 8);
            });
            const size_t b0 = lay[(size_t)i0].Arp, b1 = (i1 < count) ? lay[(size_t)i1].Arp : upload_bytes;
            if (b1 > b0) SHIP(hipMemcpyAsync(dbase + b0, harena + b0, b1 - b0, hipMemcpyHostToDevice, S.stream));
        }
    }
    lap("staging fill + upload enqueue");
    if (tprof) { SHIP(hipStreamSynchronize(S.stream)); lap("upload"); }
    for (long i = 0; i < count; i++) {
        const QPDOData *d = items[i].data; Lay &L = lay[(size_t)i]; SmallQP &p = hp[(size_t)i];
        memset(&p, 0, sizeof(p));
        p.n = (int)d->n; p.m = (int)d->m; p.c_const = d->c;
        p.Arp = (const int *)(dbase + L.Arp); p.Aci = (const int *)(dbase + L.Aci); p.Aval = (double *)(dbase + L.Aval);
        p.Trp = (const int *)(dbase + L.Trp); p.Tci = (const int *)(dbase + L.Tci); p.Tval = (double *)(dbase + L.Tval);
        p.Qrp = (const int *)(dbase + L.Qrp); p.Qci = (const int *)(dbase + L.Qci); p.Qval = (double *)(dbase + L.Qval);
        p.q = (double *)(dbase + L.q); p.l = (double *)(dbase + L.l); p.u = (double *)(dbase + L.u);
        p.x0 = items[i].x0 ? (const double *)(dbase + L.x0) : nullptr; p.y0 = items[i].y0 ? (const double *)(dbase + L.y0) : nullptr;
        p.nv = (double *)(dbase + L.nv); p.mv = (double *)(dbase + L.mv); p.lsv = (double *)(dbase + L.lsv); p.iv = (int *)(dbase + L.iv); p.tpos = (int *)(dbase + L.tpos);
        p.K = (double *)(dbase + L.K);
        p.sol_x = (double *)(dbase + L.solx); p.sol_y = (double *)(dbase + L.soly); p.cert_dx = (double *)(dbase + L.dx); p.cert_dy = (double *)(dbase + L.dy);
    }
    {
        const char *pf = getenv("QPDO_SMALL_PROF");
        if (S.dprof) { (void)hipFree(S.dprof); S.dprof = nullptr; }
        if (pf && !strcmp(pf, "1")) {
            SHIP(hipMalloc((void **)&S.dprof, (size_t)count * PH_COUNT * sizeof(long long)));
            SHIP(hipMemsetAsync(S.dprof, 0, (size_t)count * PH_COUNT * sizeof(long long), S.stream));
            for (long i = 0; i < count; i++) hp[(size_t)i].prof = S.dprof + i * PH_COUNT;
        }
    }
    {
        size_t nmax = 1, mmax = 0;
        for (long i = 0; i < count; i++) { if (items[i].data->n > nmax) nmax = items[i].data->n; if (items[i].data->m > mmax) mmax = items[i].data->m; }
        const size_t budget = SMALL_LDS_BUDGET;
        int klds_ok = 0; size_t ubytes = 0;
        size_t lds = small_lds_bytes(nmax, mmax, &klds_ok, &ubytes);
        if (const char *kg = getenv("QPDO_SMALL_K_GLOBAL")) { if (atoi(kg) && klds_ok) { klds_ok = 0; lds = small_lds_bytes(nmax, mmax, nullptr, &ubytes); } }      // occupancy experiments
        int klds_lat = 0; size_t ub_lat = 0;
        const size_t lds_lat = small_lds_bytes(nmax, mmax, &klds_lat, &ub_lat, true);           // the latency kernel's layout (two factor buffers)
        SHIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_small_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(budget)));
        if (const char *pad = getenv("QPDO_SMALL_LDS_MIN")) { const size_t v = (size_t)atol(pad); if (v > lds && v <= budget) lds = v; }   // occupancy experiments
        const int kflags = klds_ok | ((int)(ubytes / 8) << 1);       // (the LDS_MIN padding experiment below the launch only grows the tail)
        // Which kernel (QPDO_SMALL_BATCH_KERNEL=wide|lat overrides).  "lat": the latency variant -- one workgroup per CU with 256 VGPRs and
        // the item's ~30 work vectors in LDS beside the factor; "wide": two workgroups per CU at 128 VGPRs, vectors in global memory.  Same
        // operations on the same values either way.  The wide kernel has the throughput (4096 C3 items at max_iter 300: 0.064 against
        // 0.087 s), the latency kernel the faster single item (an item that runs all 10000 passes: 0.27 against 0.33 s).  A batch that is
        // solved one at a time under a large pass limit is as slow as its slowest item: it takes the latency kernel; batches of a STREAM
        // overlap their stragglers with the next batches' ordinary items: they take the wide kernel.
        const size_t voff = (lds_lat + 15) & ~(size_t)15;
        const size_t vbytes = ((size_t)NV_COUNT * nmax + (size_t)MV_COUNT * mmax) * 8 + 3 * mmax * 4 + 16;
        const char *bk = getenv("QPDO_SMALL_BATCH_KERNEL");
        const bool lat_fits = klds_ok && klds_lat && ub_lat == ubytes && voff + vbytes <= budget;
        const bool use_lat = lat_fits && !(bk && !strcmp(bk, "wide")) && (bk ? !strcmp(bk, "lat") : (one_at_a_time && (count <= 256 || settings->max_iter >= 1000)));
        if (use_lat) {
            for (long i = 0; i < count; i++) hp[(size_t)i].batch_vec_off = (unsigned)voff;
            SHIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_small_solve_lat), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(budget)));
        }
        SHIP(hipMemcpyAsync(S.dprobs, hp, (size_t)count * sizeof(SmallQP), hipMemcpyHostToDevice, S.stream));
        SHIP(hipEventRecord(S.ev0, S.stream));
        if (use_lat) hipLaunchKernelGGL(k_small_solve_lat, dim3((unsigned)count), dim3(SM_THREADS), voff + vbytes, S.stream, S.dprobs, (int)count, *settings, kflags);
        else hipLaunchKernelGGL(k_small_solve, dim3((unsigned)count), dim3(SM_THREADS), lds, S.stream, S.dprobs, (int)count, *settings, kflags);
        SHIP(hipEventRecord(S.ev1, S.stream));
    }
    SHIP(hipGetLastError());
    // The downloads are NOT enqueued here: a device-to-host copy waiting for this kernel would sit at the head of the copy
    // engine's in-order ring for as long as the slowest workgroup runs, and the upload of the next batch (another stream, the
    // same engine) would queue behind it -- measured: batches submitted 20 ms apart then ran strictly one after the other.
    // slot_finish issues them once the kernel has completed.
    S.count = count; S.items = items; S.upload_bytes = upload_bytes; S.out_bytes = out_bytes; S.busy = true;
    lap("enqueue");
done:
    // a failure after work was enqueued (the upload, the launch): the pinned staging and the arena must not be repacked, regrown or
    // freed by the next submit while that work is still in flight
    if (rc && S.stream) (void)hipStreamSynchronize(S.stream);
    return rc;
}
// wait for the slot's batch and hand the results to its items
static int slot_finish(SmallSlot &S) {
    int rc = 0;
    if (!S.busy) return 0;
    const long count = S.count; QPDOAmdBatchItem *items = S.items; const size_t upload_bytes = S.upload_bytes;
    SHIP(hipSetDevice(S.device));
    SHIP(hipEventSynchronize(S.ev1));
    SHIP(hipMemcpyAsync(S.hp, S.dprobs, (size_t)count * sizeof(SmallQP), hipMemcpyDeviceToHost, S.stream));
    SHIP(hipMemcpyAsync(S.hout, S.arena + upload_bytes, S.out_bytes, hipMemcpyDeviceToHost, S.stream));
    SHIP(hipStreamSynchronize(S.stream));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, S.ev0, S.ev1) == hipSuccess) { S.kernel_s = (double)ms * 1e-3; s_last_kernel_s = S.kernel_s; } }
    if (S.dprof) {   // diagnostic: phase shares of the longest-running item
        std::vector<long long> hpf((size_t)count * PH_COUNT);
        SHIP(hipMemcpy(hpf.data(), S.dprof, hpf.size() * sizeof(long long), hipMemcpyDeviceToHost));
        long best = 0; long long bt = -1;
        for (long i = 0; i < count; i++) { long long t = 0; for (int k = 0; k < PH_COUNT; k++) t += hpf[(size_t)i * PH_COUNT + k]; if (t > bt) { bt = t; best = i; } }
        static const char *nm[PH_COUNT] = {"resid", "outer", "prep", "assemble", "factor", "solve", "spmv", "linesearch", "update", "(ls:dots", "ls:sort", "ls:jsum", "ls:walk)", "cycles", "ticks"};
        fprintf(stderr, "[qpdo_small prof] item %ld, %ld passes, ticks(100MHz):", best, (long)S.hp[(size_t)best].info.iterations);
        for (int k = 0; k < PH_COUNT; k++) fprintf(stderr, " %s=%.1fms", nm[k], hpf[(size_t)best * PH_COUNT + k] * 1e-5);
        fprintf(stderr, "\n");
    }
    parallel_items(count, [&](long i) {
        const QPDOData *d = items[i].data; Lay &L = S.lay[(size_t)i];
        const size_t n = d->n, m = d->m;
        items[i].info = S.hp[(size_t)i].info;
        const long stv = items[i].info.status_val;
        const double *sx = (const double *)(S.hout + (L.solx - upload_bytes)), *sy = (const double *)(S.hout + (L.soly - upload_bytes));
        const bool infeasible = (stv == QPDO_PRIMAL_INFEASIBLE) || (stv == QPDO_DUAL_INFEASIBLE);
        if (items[i].x) for (size_t k = 0; k < n; k++) items[i].x[k] = infeasible ? NAN : sx[k];
        if (items[i].y) for (size_t k = 0; k < m; k++) items[i].y[k] = infeasible ? NAN : sy[k];
    });
done:
    if (rc && S.stream) (void)hipStreamSynchronize(S.stream);
    return rc;                                     // (S.busy is cleared by the caller, under the lock that guards the slots)
}

struct SmallStream { int device; std::vector<SmallSlot> slots; long next_ticket = 0; std::mutex mu; };
static std::mutex s_slot0_mu;
static SmallSlot s_slot0;                      // the one-batch-at-a-time entry point's slot

extern "C" {

const char *qdev_small_last_error(void) { return s_err; }
double qdev_small_last_kernel_seconds(void) { return s_last_kernel_s; }

// 1 if every item fits the fused kernel
int qdev_small_eligible(long count, const void *items_) {
    const QPDOAmdBatchItem *items = (const QPDOAmdBatchItem *)items_;
    for (long i = 0; i < count; i++) {
        const QPDOData *d = items[i].data;
        if (!d || !d->Q || !d->A) return 0;
        if (d->n < 1 || d->n > SM_MAX_N || d->m > SM_MAX_M) return 0;
        // the checks qpdo_setup makes before it touches a matrix (qpdo_api.c sparse_ok + dimensions): anything else goes to
        // the generic path, which rejects it with a message instead of indexing out of bounds here
        const cholmod_sparse *Ms[2] = {d->Q, d->A};
        for (int k = 0; k < 2; k++) {
            const cholmod_sparse *M = Ms[k];
            if (!M->p || (M->itype != 0 && M->itype != 2) || M->xtype != 1 || M->dtype != 0) return 0;
            if (!M->packed && M->nz) return 0;
            const long long nnz = idx_at(M->p, M->itype, (long long)M->ncol);
            if (nnz < 0 || nnz >= 2147483647LL || (nnz > 0 && (!M->i || !M->x))) return 0;
        }
        if (d->Q->nrow != d->n || d->Q->ncol != d->n || d->A->nrow != d->m || d->A->ncol != d->n) return 0;
        if (!d->q || (d->m > 0 && (!d->l || !d->u))) return 0;
    }
    return 1;
}

// Solve all items with the fused kernel on `device`, one batch at a time.  Returns 0 on success.
int qdev_small_batch(int device, long count, void *items_, const void *settings_) {
    std::lock_guard<std::mutex> lock(s_slot0_mu);
    int rc = slot_submit(s_slot0, device, count, (QPDOAmdBatchItem *)items_, (const QPDOSettings *)settings_, true);
    if (rc == 0) rc = slot_finish(s_slot0);
    s_slot0.busy = false;
    return rc;
}

// ---- batch stream: up to `depth` batches in flight on `device`, each on its own HIP stream -------------------------------
void *qdev_small_stream_create(int device, int depth) {
    if (depth < 1) depth = 1;
    if (depth > 64) depth = 64;
    SmallStream *T = new SmallStream();
    T->device = device; T->slots.resize((size_t)depth);
    return T;
}
// Packs and starts the batch; returns a ticket (>= 0) without waiting for the GPU, or -1.  The items (and everything they
// point to) must stay valid and untouched until qdev_small_stream_wait(ticket) returns.  If all slots are busy the oldest
// batch in flight must be waited for first: the call fails rather than blocking behind it.
long qdev_small_stream_submit(void *h, long count, void *items_, const void *settings_) {
    SmallStream *T = (SmallStream *)h;
    std::lock_guard<std::mutex> lock(T->mu);
    for (SmallSlot &S : T->slots) {
        if (S.busy) continue;
        if (slot_submit(S, T->device, count, (QPDOAmdBatchItem *)items_, (const QPDOSettings *)settings_, false) != 0) { S.busy = false; return -1; }
        S.ticket = T->next_ticket++;
        return S.ticket;
    }
    snprintf(s_err, sizeof(s_err), "batch stream: all %zu slots are in flight (wait for a ticket first)", T->slots.size());
    return -1;
}
// Blocks until the batch of `ticket` is complete and its items hold their results.  0 ok, -1 error / unknown ticket.
int qdev_small_stream_wait(void *h, long ticket, double *kernel_seconds) {
    SmallStream *T = (SmallStream *)h;
    SmallSlot *S = nullptr;
    { std::lock_guard<std::mutex> lock(T->mu); for (SmallSlot &c : T->slots) if (c.busy && c.ticket == ticket) S = &c; }
    if (!S) { snprintf(s_err, sizeof(s_err), "batch stream: ticket %ld is not in flight", ticket); return -1; }
    const int rc = slot_finish(*S);                // (outside the lock: other tickets may be submitted / waited for meanwhile)
    if (kernel_seconds) *kernel_seconds = S->kernel_s;
    { std::lock_guard<std::mutex> lock(T->mu); S->busy = false; }
    return rc;
}
// ---- resident mode: qpdo_solve of ONE small workspace through the latency variant of the fused kernel ------------------------------
// Everything the kernel reads per solve is already on the device (the workspace's own arrays); its descriptor and every output live in
// pinned host memory that the kernel reads / writes directly: one launch + one stream synchronisation per qpdo_solve, no copies.
struct SmallResident {
    int device = 0; hipStream_t stream = nullptr; int n = 0, m = 0;
    char *arena = nullptr;                             // device scratch of the one item: nv, mv, lsv, iv, tpos, K
    struct HostBlock { SmallQP p; SmallRes r; } *hb = nullptr;      // pinned
    double *hout = nullptr;                            // pinned: sol_x(n) sol_y(m) x(n) y(m) dx(n) dy(m)
    QPDOAmdTraceRec *htrace = nullptr; long trace_cap = 0;           // pinned
    size_t lds = 0; int klds_ok = 0, kflags = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t o_nv = 0, o_mv = 0, o_lsv = 0, o_iv = 0, o_tpos = 0, o_K = 0;
    unsigned lds_vec_off = 0;
    double *h_x0 = nullptr, *h_y0 = nullptr;           // pinned copies of an explicit warm start's vectors (inside hout)
    long long *dprof = nullptr;                        // QPDO_SMALL_PROF=1: in-kernel phase ticks
};
int qdev_small_resident_fits(int32_t n, int32_t m) {
    if (n < 1 || n > SM_MAX_N || m < 0 || m > SM_MAX_M) return 0;
    int ok = 0;
    (void)small_lds_bytes((size_t)n, (size_t)m, &ok, nullptr, true);
    return ok;                                          // the packed factor must live in LDS: beyond that one workgroup is the wrong shape
}
void qdev_small_resident_destroy(void *h) {
    SmallResident *R = (SmallResident *)h;
    if (!R) return;
    (void)hipSetDevice(R->device);
    if (R->stream) (void)hipStreamSynchronize(R->stream);
    if (R->arena) (void)hipFree(R->arena);
    if (R->hb) (void)hipHostFree(R->hb);
    if (R->hout) (void)hipHostFree(R->hout);
    if (R->htrace) (void)hipHostFree(R->htrace);
    if (R->dprof) (void)hipFree(R->dprof);
    if (R->ev0) (void)hipEventDestroy(R->ev0);
    if (R->ev1) (void)hipEventDestroy(R->ev1);
    delete R;
}
void *qdev_small_resident_create(const QdevSmallView *v, long trace_cap) {
    int rc = 0;
    SmallResident *R = new SmallResident();
    R->device = v->device; R->stream = (hipStream_t)v->stream; R->n = v->n; R->m = v->m;
    const size_t n = (size_t)v->n, m = (size_t)v->m;
    size_t total = 0;
    auto reserve = [&](size_t bytes) { size_t o = total; total += (bytes + 255) & ~(size_t)255; return o; };
    int nnzA = 0;
    SHIP(hipSetDevice(R->device));
    SHIP(hipMemcpy(&nnzA, v->Arp + m, sizeof(int), hipMemcpyDeviceToHost));
    R->o_nv = reserve((size_t)NV_COUNT * n * 8); R->o_mv = reserve((size_t)MV_COUNT * m * 8 + 8); R->o_lsv = reserve(4 * m * 8 + 8);
    R->o_iv = reserve(3 * m * 4 + 4); R->o_tpos = reserve((size_t)nnzA * 4 + 4); R->o_K = reserve(n * n * 8);
    SHIP(hipMalloc((void **)&R->arena, total));
    SHIP(hipHostMalloc((void **)&R->hb, sizeof(*R->hb), hipHostMallocDefault));
    SHIP(hipHostMalloc((void **)&R->hout, (4 * n + 4 * m + 8) * 8, hipHostMallocDefault));
    R->h_x0 = R->hout + (3 * n + 3 * m + 6); R->h_y0 = R->h_x0 + n;
    if (trace_cap < 1) trace_cap = 1;
    SHIP(hipHostMalloc((void **)&R->htrace, (size_t)trace_cap * sizeof(QPDOAmdTraceRec), hipHostMallocDefault));
    R->trace_cap = trace_cap;
    SHIP(hipEventCreate(&R->ev0)); SHIP(hipEventCreate(&R->ev1));
    { size_t ub = 0; R->lds = small_lds_bytes(n, m, &R->klds_ok, &ub, true); R->kflags = R->klds_ok | ((int)(ub / 8) << 1); }
    {   // vectors into LDS when they fit beside everything else (QPDO_SMALL_VEC_LDS=0: keep them in global memory)
        const size_t off = (R->lds + 15) & ~(size_t)15;
        const size_t vbytes = ((size_t)NV_COUNT * n + (size_t)MV_COUNT * m) * 8 + 3 * m * 4 + 16;
        const char *ve = getenv("QPDO_SMALL_VEC_LDS");
        R->lds_vec_off = 0;
        if (R->klds_ok && off + vbytes <= SMALL_LDS_BUDGET && !(ve && atoi(ve) == 0)) { R->lds_vec_off = (unsigned)off; R->lds = off + vbytes; }
    }
    if (!R->klds_ok) { snprintf(s_err, sizeof(s_err), "resident fused solve: the packed factor of n = %d does not fit in LDS", v->n); rc = -1; goto done; }
    SHIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_small_solve_lat), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_BUDGET));
done:
    if (rc) { qdev_small_resident_destroy(R); return nullptr; }
    return R;
}
// fills the pinned descriptor of the one item for a launch in `mode` (SmallRes)
static void resident_fill(SmallResident *R, const QdevSmallView *v, int mode, double c_const) {
    const size_t n = (size_t)R->n, m = (size_t)R->m;
    SmallQP &p = R->hb->p; SmallRes &r = R->hb->r;
    double *o = R->hout;
    double *h_solx = o, *h_soly = h_solx + n, *h_x = h_soly + m + 1, *h_y = h_x + n, *h_dx = h_y + m + 1, *h_dy = h_dx + n;
    memset(&p, 0, sizeof(p)); memset(&r, 0, sizeof(r));
    p.n = (int)n; p.m = (int)m; p.c_const = c_const;
    p.Arp = v->Arp; p.Aci = v->Aci; p.Aval = const_cast<double *>(v->Aval);            // (read only here: the kernel's Ruiz phase, the one writer, is skipped)
    p.Trp = v->Trp; p.Tci = v->Tci; p.Tval = const_cast<double *>(v->Tval);
    p.Qrp = v->Qrp; p.Qci = v->Qci; p.Qval = const_cast<double *>(v->Qval);
    p.q = const_cast<double *>(v->q); p.l = const_cast<double *>(v->l); p.u = const_cast<double *>(v->u);
    p.nv = (double *)(R->arena + R->o_nv); p.mv = (double *)(R->arena + R->o_mv); p.lsv = (double *)(R->arena + R->o_lsv);
    p.iv = (int *)(R->arena + R->o_iv); p.tpos = (int *)(R->arena + R->o_tpos); p.K = (double *)(R->arena + R->o_K);
    p.sol_x = h_solx; p.sol_y = h_soly; p.cert_dx = h_dx; p.cert_dy = h_dy;
    p.res = &R->hb->r;
    p.prof = R->dprof;
    r.mode = mode;
    r.rD = v->D; r.rDinv = v->Dinv; r.rE = v->E; r.rEinv = v->Einv; r.r_c = v->c; r.r_cinv = v->cinv;
    r.state_x = v->st_x; r.state_Qx = v->st_Qx; r.st_xbar = v->st_xbar; r.st_Aty = v->st_Aty;
    r.st_y = v->st_y; r.st_ybar = v->st_ybar; r.st_Ax = v->st_Ax; r.st_mu = v->st_mu; r.st_isq = v->st_isq;
    r.out_x = h_x; r.out_y = h_y;
    r.trace = R->htrace; r.trace_cap = R->trace_cap;
    r.lds_vec_off = R->lds_vec_off;
}
int qdev_small_resident_warm_start(void *h, const QdevSmallView *v, const void *settings_, const double *x_ws, const double *y_ws, double c_const,
                                   double *objective) {
    int rc = 0;
    SmallResident *R = (SmallResident *)h;
    const QPDOSettings *settings = (const QPDOSettings *)settings_;
    const size_t n = (size_t)R->n, m = (size_t)R->m;
    SHIP(hipSetDevice(R->device));
    resident_fill(R, v, 1, c_const);
    if (x_ws) { memcpy(R->h_x0, x_ws, n * 8); R->hb->p.x0 = R->h_x0; }         // pinned copies: the caller may release its vectors on return
    if (y_ws && m) { memcpy(R->h_y0, y_ws, m * 8); R->hb->p.y0 = R->h_y0; }
    hipLaunchKernelGGL(k_small_solve_lat, dim3(1), dim3(SM_THREADS), R->lds, R->stream, &R->hb->p, 1, *settings, R->kflags);
    SHIP(hipGetLastError());
    SHIP(hipStreamSynchronize(R->stream));
    *objective = R->hb->r.ws_objective;
done:
    return rc;
}
int qdev_small_resident_solve(void *h, const QdevSmallView *v, const void *settings_, int from_state, double c_const, QdevSmallResult *out) {
    int rc = 0;
    SmallResident *R = (SmallResident *)h;
    const QPDOSettings *settings = (const QPDOSettings *)settings_;
    const size_t n = (size_t)R->n, m = (size_t)R->m;
    SmallQP &p = R->hb->p; SmallRes &r = R->hb->r;
    double *o = R->hout;
    double *h_solx = o, *h_soly = h_solx + n, *h_x = h_soly + m + 1, *h_y = h_x + n, *h_dx = h_y + m + 1, *h_dy = h_dx + n;
    const char *pf = getenv("QPDO_SMALL_PROF");
    const bool prof = pf && !strcmp(pf, "1");
    SHIP(hipSetDevice(R->device));
    if (settings->max_iter > R->trace_cap && R->trace_cap < (1L << 20)) {            // grow the trace with max_iter (bounded: 1 M records = 104 MB)
        long cap = settings->max_iter < (1L << 20) ? (long)settings->max_iter : (1L << 20);
        SHIP(hipStreamSynchronize(R->stream));
        (void)hipHostFree(R->htrace); R->htrace = nullptr; R->trace_cap = 0;
        SHIP(hipHostMalloc((void **)&R->htrace, (size_t)cap * sizeof(QPDOAmdTraceRec), hipHostMallocDefault));
        R->trace_cap = cap;
    }
    if (prof && !R->dprof) SHIP(hipMalloc((void **)&R->dprof, PH_COUNT * sizeof(long long)));
    if (R->dprof) SHIP(hipMemsetAsync(R->dprof, 0, PH_COUNT * sizeof(long long), R->stream));
    resident_fill(R, v, from_state ? 2 : 0, c_const);
    if (!prof) p.prof = nullptr;
    SHIP(hipEventRecord(R->ev0, R->stream));
    hipLaunchKernelGGL(k_small_solve_lat, dim3(1), dim3(SM_THREADS), R->lds, R->stream, &R->hb->p, 1, *settings, R->kflags);
    SHIP(hipGetLastError());
    SHIP(hipEventRecord(R->ev1, R->stream));
    SHIP(hipStreamSynchronize(R->stream));
    { float ms = 0.f; out->kernel_seconds = (hipEventElapsedTime(&ms, R->ev0, R->ev1) == hipSuccess) ? (double)ms * 1e-3 : 0.0; }
    if (prof) {
        long long hpf[PH_COUNT];
        SHIP(hipMemcpy(hpf, R->dprof, sizeof(hpf), hipMemcpyDeviceToHost));
        static const char *nm[PH_COUNT] = {"resid", "outer", "prep", "assemble", "factor", "solve", "spmv", "linesearch", "update", "(ls:dots", "ls:sort", "ls:jsum", "ls:walk)", "", ""};
        fprintf(stderr, "[qpdo_small resident prof] n=%d m=%d, %ld passes, %ld factorizations, kernel %.3f ms, in-kernel clock %.0f MHz; us:", R->n, R->m, (long)p.info.iterations, p.factor_count,
                out->kernel_seconds * 1e3, hpf[PH_WALL] > 0 ? 100.0 * (double)hpf[PH_CYC] / (double)hpf[PH_WALL] : 0.0);
        for (int k = 0; k < PH_CYC; k++) fprintf(stderr, " %s=%.0f", nm[k], hpf[k] * 1e-2);
        fprintf(stderr, "\n");
    }
    *(QPDOInfo *)out->info = p.info;
    out->newton_passes = p.newton_passes; out->factor_count = p.factor_count;
    out->ntrace = r.ntrace < R->trace_cap ? r.ntrace : R->trace_cap; out->trace = R->htrace;
    out->sol_x = h_solx; out->sol_y = h_soly; out->x = h_x; out->y = h_y; out->dx = h_dx; out->dy = h_dy;
    out->sigma_end = r.sigma_end; out->tau_end = r.tau_end;
done:
    return rc;
}

void qdev_small_stream_destroy(void *h) {
    SmallStream *T = (SmallStream *)h;
    if (!T) return;
    for (SmallSlot &S : T->slots) { if (S.busy) { (void)slot_finish(S); S.busy = false; } slot_release(S); }
    delete T;
}

}  // extern "C"
