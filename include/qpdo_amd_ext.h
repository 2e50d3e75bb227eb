/*
 * qpdo_amd_ext.h -- extensions of libqpdo_amd.so beyond the reference API.
 *
 * Nothing here is needed to use the library as a drop-in for the reference;
 * these entry points exist for measurement (bench.py), for the parity tests of
 * single kernels, and for device selection.  All are plain C-ABI.
 *
 * Environment variables read once per qpdo_setup:
 *   QPDO_DEVICE      HIP device ordinal (default: LOCAL_RANK if set, else 0)
 *   QPDO_LINSOLVE    "pcg" | "dense" | "band" | "auto" (default auto: the BAND direct solver when the Newton matrix is banded -- half-bandwidth
 *                    of Q + A'A <= 127, chain-structured QPs -- and n >= 2048; otherwise dense LDL' for n <= QPDO_DENSE_MAX_N = 12288, PCG
 *                    above; "band" on a matrix that is not banded makes qpdo_setup fail with a message; the dense solver accepts
 *                    n <= 40000 -- its assembly tiles the LDS accumulator, QPDO_DENSE_ASM_TILE rows at a time -- and is also the rescue of
 *                    a PCG solve that cannot converge up to that order)
 *   QPDO_HYBRID      where the dense solver is selected automatically and n >= 8192, every solve starts with PCG and switches to the dense
 *                    factor at the first Newton pass that needs more than 450 PCG iterations (default since round 4; same per-pass integers;
 *                    QPDOAmdStats.hybrid_pcg_passes counts the PCG passes; every numerical PCG failure hands the pass to the dense factor).  "0": off; "1": on from n = 4096; "<budget>" > 1: on from n = 4096 with that budget
 *   QPDO_DENSE_MID   "0": the dense factorization as the multi-launch blocked pipeline of rounds 1-4 instead of ONE launch of tile-owning
 *                    workgroups (k_mid_factor, the default at every order since round 5: n = 1e4 8.4 ms against 13.3 ms).  The
 *                    look-ahead / outer-panel / syrk knobs below act on the multi-launch path only
 *   QPDO_CTRL_PUBLISH "0": the per-pass read-back of the control block as hipMemcpyAsync + hipStreamSynchronize instead of a kernel that
 *                    writes the block and a sequence word into coherent pinned memory while the host spins (bounded; default since round 5:
 *                    15.7 -> 9.9 us per read-back).  Only the transport differs: the same bits
 *   QPDO_FUSE_RESID  "0": the deferred Newton step's five axpys and the read-back's publication as launches of their own instead of
 *                    inside the residual launch (dense / band routes; the same bits)
 *   QPDO_LS_SMALL    "0": the linesearch through the radix-sort kernels (28 launches) at every size instead of ONE launch for 2m <= 8192
 *                    (read at qpdo_setup; the same tau bits: tests/test_gpu_parity.py)
 *   QPDO_FUSE_OUTER  "0": the outer-update sequences (infeasibility tests, mu update, shifting the estimates) as their separate kernels and
 *                    device copies (27 launches) instead of 10 (the same bits); also: Q dx and A dx of a Newton step as two launches
 *   QPDO_LAUNCH_AHEAD "0": the Newton step of a pass is launched after the host has read the pass's norms and decided, instead of behind
 *                    the residual launch with the decision formed on the device (mid-size dense route, n < 9000, 2m <= 8192; DESIGN.md
 *                    section 5; read at qpdo_setup; QPDOAmdStats.ahead_steps / ahead_skips).  The same kernels in the same order: the same bits,
 *                    except on a pass whose factor the host-first path would keep (it is refactored: the same matrix)
 *   QPDO_DENSE_LOWRANK  "0": refactor on every weight change, "1": low-rank update of the kept dense factor (default: from n = 9000 up)
 *   QPDO_DENSE_LOOKAHEAD "0": factor on one stream, "1": overlap the next panel with the trailing update (default: from n = 7000 up)
 *   QPDO_DENSE_RESERVE_CUS  CUs left out of the trailing-update stream's mask (default 32; 0 = no mask)
 *   QPDO_DENSE_SOLVE "steps": per-block-step triangular solve kernels instead of the one-launch chained solves
 *   Experiment knobs of the dense factor (every setting leaves the same factor bits):
 *   QPDO_DENSE_OUTER (block columns per outer panel, default 4), QPDO_SYRK_KC (16 | 32), QPDO_SYRK_SWZ ("0": 2-D tile grid instead of
 *   the XCD-aware order), QPDO_DENSE_FPANEL ("1": a whole outer panel in one
 *   launch; slower at present)
 *   QPDO_SETUP_THREADS  host threads of the CSC -> CSR conversions in qpdo_setup (default min(16, cores)); QPDO_SETUP_PROF=1 prints phase times
 *   QPDO_SLAB_TPR    lanes per row segment of the slab SpMV (8 | 16 | 32, default 16)
 *   QPDO_SPMV        "slab" | "plain" (default: LDS-staged slab kernel for matrices >= 192 MB)
 *   QPDO_DEFLATE     "0" disables the heavy-row deflation of the PCG preconditioner
 *   QPDO_IDX16       "0" disables the 16-bit slab-local column indices
 *   QPDO_PCG_SCHUR   "0" disables / "1" forces the Schur-complement mode of the PCG (default: automatic)
 *   QPDO_PCG_INNER_F32  "1": the Schur mode's inner (preconditioner) solve streams an fp32 copy of the compact matrix values;
 *                    vectors, accumulation and the outer CG on the exact K stay fp64 (opt-in, default off)
 *   QPDO_PCG_TOL     relative residual tolerance of the Jacobi-PCG solve (default 1e-12)
 *   QPDO_PCG_ABS     factor f of the absolute stopping rule of the PCG solve: stop when the residual, in the unscaled inf-norm of
 *                    the reference's inner dual residual, is <= f * eps_abs (default 1e-5; 0: relative rule only; proximal only)
 *   QPDO_PCG_MAXIT   PCG iteration cap per Newton step (default 100000)
 *   QPDO_SMALL_FUSED "0": never route qpdo_solve through the fused one-launch kernel (default: workspaces with n <= QPDO_SMALL_FUSED_MAX_N
 *                    = 160, m <= 1024 whose packed Newton matrix fits one workgroup's LDS, verbose = 0, no explicit QPDO_LINSOLVE)
 *   QPDO_SMALL_BATCH_KERNEL  "wide" | "lat": launch shape of a fused batch (qpdo_amd_solve_batch / batch_stream).  Default: a batch solved one at a
 *                    time takes the latency kernel (one workgroup per CU, work vectors in LDS) when it has <= 256 items or max_iter >= 1000 -- it is
 *                    as slow as its slowest item --, batches of a stream the wide one (two workgroups per CU: the throughput).  Same bits either way
 *   QPDO_FIX_STATUS_RESET  "1": reset info->status_val at the start of qpdo_solve
 *                    (the reference does not: src/qpdo.c:451-453 vs :200)
 */
#ifndef QPDO_AMD_EXT_H
#define QPDO_AMD_EXT_H

#include "qpdo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one record per loop pass of qpdo_solve (reference src/qpdo.c:343-449) */
typedef struct {
    long   kind;            /* 0 Newton step, 1 outer update, 2 terminated in this pass */
    long   n_active, n_enter, n_leave;
    long   factor_branch;   /* 0 full, 1 rank update, 2 Q only, -1 n/a (src/newton.c:21-33) */
    long   lin_iters;       /* PCG iterations of this pass (0 for the dense solver)       */
    double tau;
    double res_prim, res_dual, res_prim_in, res_dual_in;
    double sigma, eps_in;
} QPDOAmdTraceRec;

typedef struct {
    long   newton_passes;   /* passes that ran update_iterate (iterations - oterations - final) */
    long   lin_iters;       /* PCG iterations, all passes                                  */
    long   spmv_calls;      /* SpMV launches in the last solve                             */
    double spmv_alg_bytes;  /* sum over those launches of 12 nnz + 4(rows+1) + 8 rows + 8 cols */
    long   factor_count;    /* dense LDL' factorizations                                   */
    long   linsolve;        /* 0 pcg, 1 dense, 2 fused small-problem kernel, 3 band direct  */
    double spmv_Q_avg_s;    /* HIP-event average duration of the sampled Q SpMV inside PCG */
    long   spmv_Q_samples;
    double spmv_Ac_time_s;  /* Schur-mode inner solves: summed HIP-event time of the sampled A_c products ...          */
    double spmv_Ac_bytes;   /* ... and their summed algorithmic bytes (compact matrix, size changes per pass)          */
    long   spmv_Ac_samples;
    long   schur_passes;    /* Newton passes solved by the Schur-complement mode of the PCG                            */
    long   lowrank_solves;  /* dense solves through the low-rank update of the kept factor (cholmod_interface.c:57-93) */
    long   lowrank_cols;    /* rows that entered the low-rank set (one multi-RHS solve column each)                 */
    long   lowrank_sweeps;  /* refinement sweeps of the low-rank solves (one kept-factor solve + 3 SpMV each)        */
    long   lowrank_rejects; /* low-rank solves abandoned for a refactorization (ill-conditioned downdate)          */
    long   pcg_soft_accepts;/* PCG solves that stopped at the iteration cap or stagnated and were accepted because their relative
                             * residual was <= 1e-8; a worse or NaN residual ends qpdo_solve with status QPDO_ERROR instead      */
    long   collectives;     /* all-reduces issued by a row-partitioned solve (0 on one GPU)                                        */
    long   inner_solves, inner_steps, inner_collectives;   /* Schur mode: inner (preconditioner) solves, the iterations launched for them,
                             * and the all-reduces they issued: exactly one per launched iteration plus one per solve              */
    long   chain_fallbacks; /* dense triangular solves redone with the stepwise kernels (dense LDL')                  */
    double pcg_max_relres;  /* largest ||r||/||rhs|| a PCG solve of the last qpdo_solve ended with (tolerance QPDO_PCG_TOL)      */
    long   pcg_dense_fallbacks; /* PCG solves that could not converge (relative residual > 1e-8: e.g. settings->proximal = 0 on a singular
                             * Q + A'DA) and were redone by the dense LDL' solver, which the rest of that qpdo_solve then keeps (n <= 40000) */
    long   fused_solves;    /* qpdo_solve calls of this workspace that ran as ONE launch of the fused small-problem kernel (then linsolve = 2:
                             * in-LDS natural-order LDL' in the oracle's operation order, factor_count = its factorizations)                  */
    double fused_kernel_s;  /* HIP-event duration of that launch in the last qpdo_solve (0 if it took the generic path)                     */
    /* round 5 (appended: the members above keep their offsets) */
    long   pcg_rescues;     /* PCG solves that could not converge where no dense factor is possible (n > 40000 or a row partition) and were rescued */
    long   pcg_rescue_kinds;/* bit 0: the band direct solver took over (banded Newton matrix); bit 1: the pass was redone by plain Jacobi-PCG
                             * (Schur mode and deflation off, four times the iteration cap)                                                */
    long   hybrid_pcg_passes;/* hybrid PCG -> dense (default from n = 8192): Newton passes solved by PCG before the dense factor took over --
                             * `linsolve` reads 1 for such a workspace although its first passes ran PCG                                    */
    long   band_fallbacks;  /* band factorizations that met a non-positive or non-finite pivot and were redone by the dense solver / PCG   */
    long   onelaunch_factors;/* dense factorizations that ran as ONE launch of the tile-dataflow kernel (k_mid_factor, the default)          */
    long   ahead_steps;     /* Newton steps launched ahead of the host's decision (mid-size dense route, QPDO_LAUNCH_AHEAD) that ran ...   */
    long   ahead_skips;     /* ... and passes whose launched-ahead step left at once because the pass was an outer update or the last one  */
} QPDOAmdStats;

int  qpdo_amd_device_count(void);
/* What a loop pass of qpdo_solve is, from its residual norms and active-set counts (reference src/termination.c:11-30, src/qpdo.c:361-363,
 * src/newton.c:21-33): the one function behind both the host loop and the residual launch's decision on the launch-ahead route
 * (qpdo_amd/csrc/pass_decision.h).  Pure host arithmetic (no device needed): exported for the CPU tests.  res_dual / res_dual_in: already
 * multiplied by cinv.  allow_outer: iter > iter_old + 1; force_outer: iter == iter_old + inner_max_iter; n_change: n_enter + n_leave.
 * Out: the solve ends with QPDO_NON_CVX / QPDO_SOLVED, the pass is an outer update, else a Newton step with factorization branch 0 | 1 | 2. */
int  qpdo_amd_pass_decision(double res_prim, double res_dual, double res_prim_in, double res_dual_in, double eps_abs, double eps_in, int allow_outer,
                            int force_outer, int reset_newton, int n_active, int n_change, int *ends_nc, int *ends_ok, int *outer, int *branch);
const char *qpdo_amd_last_error(void);
int  qpdo_amd_get_stats(const QPDOWorkspace *work, QPDOAmdStats *out);
/* trace of the last qpdo_solve; pointer stays valid until the next solve / cleanup */
int  qpdo_amd_get_trace(const QPDOWorkspace *work, const QPDOAmdTraceRec **recs, long *count);
int  qpdo_amd_sync(QPDOWorkspace *work);

/* HIP-event timing of the SpMV kernel on the workspace's own (scaled) matrices.
 * which: 0 = A (CSR m x n), 1 = A' (CSR n x m), 2 = Q (full symmetric CSR). */
int  qpdo_amd_bench_spmv(QPDOWorkspace *work, int which, int reps, double *avg_seconds, double *alg_bytes);
/* HIP-event timing of the dense LDL' factorization (n <= QPDO_DENSE_MAX_N) with the workspace's current weights; *check
 * (optional) receives the relative residual of one solve with the fresh factor */
int  qpdo_amd_bench_dense_factor(QPDOWorkspace *work, int reps, double *avg_seconds, double *check);
/* y = M v on the device, host in/out (parity tests) */
int  qpdo_amd_spmv(QPDOWorkspace *work, int which, const double *v, double *y);
/* root of eta t + beta + delta'[delta t - alpha]_+ over 2m breakpoints (reference
 * src/linesearch.c:74-158) on the device, host in/out (parity tests) */
int  qpdo_amd_linesearch(QPDOWorkspace *work, double eta, double beta, const double *delta,
                         const double *alpha, double *tau);
/* copy a device-resident vector to the host: 0 x, 1 Qx, 2 y, 3 mu, 4 d (factor weights),
 * 5 dx, 6 dy, 7 Ax, 8 Aty, 9 l, 10 u, 11 ybar, 12 xbar, 13 w (of the last loop pass that ran) */
int  qpdo_amd_download(QPDOWorkspace *work, int which, double *dst);

/* ---- one large QP row-partitioned over G GPUs (BASELINE.json configs[3]) --------------------------------
 * One process per GPU.  Every rank calls the normal API with the SAME full problem; the library keeps rows
 * [rank*ceil(m/G), ...) of A (and the matching slices of A' and, for the PCG operator, of Q) on its GPU and
 * replicates the vectors.  The exchange step is one sum all-reduce of an n-vector per A' / K product: RCCL on
 * the solver's stream (rccl_unique_id: 128 bytes created by rank 0 with qpdo_amd_dist_unique_id and sent to
 * the others by the caller), or - for tests on a single GPU - a host callback `fn` (e.g. gloo).
 * Applies to workspaces set up afterwards in this process; world = 1 switches it off.  All ranks obtain the
 * same status, counts and solution. */
typedef void (*qpdo_amd_allreduce_fn)(void *ctx, double *buf, long count, int op /* 0 sum, 1 max */);
int  qpdo_amd_dist_config(int rank, int world, const void *rccl_unique_id, qpdo_amd_allreduce_fn fn, void *ctx);
int  qpdo_amd_dist_unique_id(void *out128);

/* ---- batch of independent QPs (BASELINE.json configs[2]: MPC-sized problems, no collective) --------------
 * Default path (every item has n, m <= 1024 and passes the checks of qpdo_setup): ONE launch of the fused kernel
 * k_small_solve, one workgroup per item running the item's qpdo_setup (Ruiz scaling), qpdo_warm_start and the whole
 * qpdo_solve loop, including settings->max_time (QPDO_MAX_TIME_REACHED, checked at the end of every pass as in the
 * reference, src/qpdo.c:441-447) and the three PROFILING times of QPDOInfo (per item, from the device wall clock).
 * Otherwise, or with QPDO_BATCH=threads: every item goes through the entry points above on `nthreads` host threads, each
 * with its own workspace and HIP stream on the device of this process (QPDO_DEVICE).  Across GPUs the caller shards
 * the item list over processes (item b -> GPU b mod G; qpdo_amd/solver.py shard_indices).  x (n) and y (m) receive the
 * solution (NaN for infeasible statuses, as the reference's mex gateway does), info the final QPDOInfo. */
typedef struct {
    const QPDOData *data;       /* problem (caller-owned, read only)                     */
    const c_float  *x0, *y0;    /* optional warm start (NULL: cold)                      */
    c_float        *x, *y;      /* out: solution, caller-allocated, length n / m         */
    QPDOInfo        info;       /* out                                                    */
} QPDOAmdBatchItem;
/* returns the number of items whose setup failed (their info.status_val is QPDO_ERROR) */
long qpdo_amd_solve_batch(long count, QPDOAmdBatchItem *items, const QPDOSettings *settings, int nthreads);
/* HIP-event duration of the fused kernel launch of the last qpdo_amd_solve_batch on this process (0 if it took the threaded path) */
double qpdo_amd_batch_kernel_seconds(void);

/* ---- STREAMED batches (BASELINE.json configs[2]: "batch of 4096 MPC-sized QPs ... streamed") ----------------------------
 * A fused-kernel launch is as slow as its slowest item: an instance that never reaches eps_abs (in the reference either)
 * holds one workgroup for max_iter passes while the other CUs idle.  A batch stream keeps up to `depth` batches in flight,
 * each on its own HIP stream with its own device arena and pinned staging: submit packs, uploads and launches batch i+1 and
 * returns; its workgroups fill the CUs that the stragglers of batch i do not occupy.  Per item the arithmetic is that of
 * qpdo_amd_solve_batch (same kernel): results do not depend on what else is in flight.
 * submit: ticket >= 0, or -1 (invalid settings / data, an item that does not fit the fused kernel -- use
 * qpdo_amd_solve_batch for those --, no free slot: wait for the oldest ticket first).  The items and everything they point
 * to must stay valid and untouched until wait(ticket) has returned; wait fills x, y and info of every item and returns 0
 * (-1: device error or unknown ticket), optionally the HIP-event duration of that batch's kernel. */
typedef struct QPDOAmdBatchStream_ QPDOAmdBatchStream;
QPDOAmdBatchStream *qpdo_amd_batch_stream_create(int depth);
long qpdo_amd_batch_stream_submit(QPDOAmdBatchStream *stream, long count, QPDOAmdBatchItem *items, const QPDOSettings *settings);
int  qpdo_amd_batch_stream_wait(QPDOAmdBatchStream *stream, long ticket, double *kernel_seconds);
void qpdo_amd_batch_stream_destroy(QPDOAmdBatchStream *stream);

#ifdef __cplusplus
}
#endif
#endif
