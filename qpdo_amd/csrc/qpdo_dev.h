/*
 * qpdo_dev.h -- thin C-ABI between the C host driver (qpdo_api.c) and the HIP
 * backend (qpdo_dev.hip).  Plain pointers and sizes only.  Every call returns 0
 * on success and a non-zero HIP error code otherwise; the driver maps failures
 * to QPDO_ERROR / NULL.
 *
 * Each entry point corresponds to one phase of the reference's hot path; the
 * reference lines it replaces are cited at the definition in qpdo_dev.hip.
 */
#ifndef QPDO_DEV_H
#define QPDO_DEV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct QpdoDev QpdoDev;

/* host-side CSR triple handed to the device at setup (int32 indices) */
typedef struct {
    int32_t nrows, ncols;
    int64_t nnz;
    const int32_t *rp;    /* nrows+1 */
    const int32_t *ci;    /* nnz */
    const double  *val;   /* nnz */
} QdevCsr;

/* norms and counts produced by one residual pass */
typedef struct {
    double res_prim, res_dual, res_prim_in, res_dual_in;   /* inf-norms, un-scaled by Einv / Dinv */
    int32_t n_active, n_enter, n_leave;
    /* a Newton step whose read-back was deferred (qdev_set_deferred_step) completes with the NEXT residual pass: its step length */
    int32_t prev_step_done; double prev_tau;
    /* launch-ahead (qdev_residuals_ahead): 1 = a Newton step was enqueued behind this residual pass; went = the device let it run, with
     * factorization branch `ahead_branch` -- the caller's own decision must agree (qdev_newton_step checks) */
    int32_t ahead_enqueued, ahead_went, ahead_branch;
} QdevResid;
/* what the device needs to form the caller's decision for this pass (qpdo_api.c qpdo_solve) */
typedef struct {
    int32_t allow_outer;        /* iter > iter_old + 1                          */
    int32_t force_outer;        /* iter == iter_old + inner_max_iter            */
    int32_t reset_newton;       /* the flag as the inner branch would see it (including the periodic reset of this iter) */
    int32_t max_rank;           /* QPDO_MAX_RANK_UPDATE (newton.c:23)           */
    double eps_abs, eps_in, infty /* QPDO_INFTY */;
} QdevAhead;

/* per-solve device statistics (extension; see include/qpdo_amd_ext.h) */
typedef struct {
    int64_t newton_passes;
    int64_t lin_iters;          /* PCG iterations, summed                    */
    int64_t spmv_calls;         /* SpMV launches                             */
    int64_t spmv_bytes;         /* algorithmic bytes moved by those launches */
    int64_t factor_count;       /* dense LDL' factorizations                 */
    int32_t linsolve;           /* 0 pcg, 1 dense                            */
    int64_t lowrank_solves;     /* dense solves that used the low-rank update of the kept factor */
    int64_t lowrank_cols;       /* rows that received a low-rank slot (multi-RHS solve columns)   */
    int64_t lowrank_sweeps;     /* refinement sweeps of those solves (each one kept-factor solve + 3 SpMV) */
    int64_t lowrank_rejects;    /* low-rank solves abandoned for a refactorization (tiny pivot)   */
    int64_t pcg_soft_accepts;   /* PCG solves that hit the iteration cap / stagnated but were accepted (rel. residual <= 1e-8) */
    int64_t collectives;        /* all-reduces issued (row-partitioned solves)                                   */
    int64_t inner_solves, inner_steps, inner_collectives;   /* Schur mode: inner solves, their iterations launched, their all-reduces */
    int64_t chain_fallbacks;    /* dense solves redone stepwise after a polled triangular solve lost a producer */
    double  pcg_max_relres;     /* largest relative residual ||r|| / ||rhs|| any PCG solve of the last qpdo_solve ended with   */
    int64_t pcg_dense_fallbacks;/* PCG solves that did not converge and were redone by the dense solver */
    int64_t pcg_rescues;        /* PCG solves that did not converge where no dense factor is possible (n > 40000, row partition) and a rescue succeeded */
    int64_t pcg_rescue_kinds;   /* which: bit 0 band direct solver took over, bit 1 plain Jacobi retry (Schur mode / deflation off, 4x the cap) */
    int64_t hybrid_pcg_passes;  /* hybrid PCG -> dense: Newton passes that were solved by PCG before the dense factor took over */
    int64_t band_fallbacks;     /* band factorizations that met a non-positive / non-finite pivot and were redone by another solver */
    int64_t onelaunch_factors;  /* dense factorizations through the one-launch tile-dataflow kernel (k_mid_factor) */
    int64_t ahead_steps;        /* Newton steps that were enqueued behind their pass's residual launch and went ahead on the device's own decision */
    int64_t ahead_skips;        /* passes whose launched-ahead step left at once (outer update, termination, lost producer) */
} QdevStats;

int qdev_device_count(void);
const char *qdev_last_error(void);

/* Ar = CSR(A) m x n, At = CSR(A') n x m, Qf = full symmetric CSR(Q) n x n.  q,l,u unscaled. */
int qdev_create(QpdoDev **out, int device, int32_t n, int32_t m,
                const QdevCsr *Ar, const QdevCsr *At, const QdevCsr *Qf,
                const double *q, const double *l, const double *u);
/* The same from the caller's CSC arrays (cholmod_sparse, int32 or int64 indices): they are uploaded as they are -- the CSC arrays of A ARE
 * CSR(A') -- and CSR(A) and the full symmetric CSR(Q) are built on the device (stable radix transposition, dev/transpose.inc).  One GPU. */
typedef struct {
    int32_t nrows, ncols; int64_t nnz;
    int itype;            /* 0: int32 indices, 2: int64 (CHOLMOD_INT / CHOLMOD_LONG) */
    const void *p, *i; const double *x;
    int stype;            /* Q: -1 lower stored, +1 upper stored, 0 full; A: 0 */
} QdevCsc;
int qdev_create_csc(QpdoDev **out, int device, int32_t n, int32_t m, const QdevCsc *A, const QdevCsc *Q,
                    const double *q, const double *l, const double *u);
/* row partition of one large QP (one process per GPU).  fn != NULL: host-callback all-reduce (tests);
 * fn == NULL and world > 1: RCCL with the 128-byte unique id of rank 0. */
typedef void (*qdev_allreduce_fn)(void *ctx, double *buf, long count, int op /*0 sum, 1 max*/);
typedef struct {
    int rank, world;
    int32_t m0, mloc;        /* rows of A held by this rank                      */
    int32_t n0, nloc;        /* rows of Q applied by this rank inside PCG        */
    qdev_allreduce_fn fn; void *ctx;
    unsigned char nccl_id[128];
    int force;               /* world == 1 but run the collective code path anyway (exercises the RCCL branch on one GPU) */
} QdevDist;
/* Ar: local rows (mloc x n); At: local columns (n x mloc, column indices local); Qf: full Q; Qs: rows [n0, n0+nloc) of Q */
int qdev_create_dist(QpdoDev **out, int device, int32_t n, int32_t m, const QdevCsr *Ar, const QdevCsr *At, const QdevCsr *Qf,
                     const QdevCsr *Qs, const double *q, const double *l, const double *u, const QdevDist *dist);
int qdev_rccl_unique_id(void *out128);
void qdev_destroy(QpdoDev *d);
int qdev_sync(QpdoDev *d);

/* Ruiz equilibration + cost scaling (reference scaling.c:24-91).  Runs `iters` more
 * iterations on the current data; D,E (host, out) receive this call's factors, c_out the
 * cost scale.  Qx_nonzero: use the workspace Qx in the cost norm (update_settings path). */
int qdev_scale_data(QpdoDev *d, int iters, int use_Qx, double *D_host, double *E_host, double *c_out);
/* install total scaling vectors (after update_settings composed them on the host) */
int qdev_set_scaling(QpdoDev *d, int scaled, const double *D, const double *Dinv,
                     const double *E, const double *Einv, double c, double cinv);
int qdev_upload_bounds(QpdoDev *d, const double *l, const double *u);   /* already scaled */
int qdev_upload_q(QpdoDev *d, const double *q);                          /* already scaled */
int qdev_download_q(QpdoDev *d, double *q);
int qdev_scale_Q_values(QpdoDev *d, double factor);                      /* qpdo.c:566-572 */

/* warm start (qpdo.c:217-299): x_ws / y_ws may be NULL.  Returns objective. */
int qdev_warm_start(QpdoDev *d, const double *x_ws, const double *y_ws, int proximal,
                    double sigma, double mu_min, double c_const, double *objective);

/* begin a solve: clear old active set (qpdo.c:321-322) */
int qdev_begin_solve(QpdoDev *d);

/* outer + inner residuals, norms, active-set counts (iteration.c:30-93, termination.c:35-77,
 * newton.c:96-126) */
int qdev_residuals(QpdoDev *d, int proximal, double sigma, QdevResid *out);
/* The same, with this pass's Newton step launched AHEAD of the host's decision (mid-size dense workspaces: the device idles ~20 us per
 * pass while the host reads the norms and launches the step's first kernels): the residual launch forms the decision itself
 * (vector.inc SpecArgs), the step's kernels are enqueued behind it guarded by that decision, and the host reads the control block
 * while they run.  Falls back to qdev_residuals when the workspace is not on that route (ahead == NULL, or see spec_route_ok). */
int qdev_residuals_ahead(QpdoDev *d, int proximal, double sigma, const QdevAhead *ahead, QdevResid *out);

/* one Newton step (iteration.c:11-25): factor-state update, direction, exact linesearch,
 * iterate update.  branch: 0 full (d = active/mu), 1 rank update (enter/leave), 2 Q only. */
int qdev_newton_step(QpdoDev *d, int branch, int n_changed /* n_enter + n_leave */, int proximal, double sigma,
                     double *tau_out, int *lin_iters_out);

/* outer-update helpers */
int qdev_primal_infeasibility(QpdoDev *d, double eps_prim_inf, int *is_infeasible);  /* termination.c:97-151 */
int qdev_dual_infeasibility(QpdoDev *d, int proximal, double sigma, double tau,
                            double eps_dual_inf, int *is_infeasible);                /* termination.c:156-216 */
int qdev_shift_estimates(QpdoDev *d);                                                /* qpdo.c:396-397 */
/* update_mu (iteration.c:127-168): returns number of changed rows; if !reset_decided_by_sigma and
 * 0 < n_changed <= 25 the caller invokes qdev_mu_changed_update. */
int qdev_update_mu(QpdoDev *d, double eps_abs, double theta, double delta, double mu_min,
                   double isq_mu_min, int *n_changed);
int qdev_mu_changed_update(QpdoDev *d);                                              /* cholmod_interface.c:77-93 */
int qdev_update_sigma(QpdoDev *d, double sigma_new, double sigma_old);               /* iteration.c:173-180 */
int qdev_save_res_prim(QpdoDev *d);                                                  /* qpdo.c:425 */

/* objective (iteration.c:185-221) and final solution (termination.c:82-92) */
int qdev_objective(QpdoDev *d, int proximal, double sigma, double c_const, double *objective);
int qdev_store_solution(QpdoDev *d, double *sol_x, double *sol_y, double *x, double *y,
                        double *dx, double *dy);
/* both at the end of qpdo_solve: one device-to-host copy (pinned) and one synchronisation instead of six copies + two */
int qdev_store_solution_obj(QpdoDev *d, int proximal, double sigma, double c_const, double *sol_x, double *sol_y, double *x, double *y,
                            double *dx, double *dy, double *objective);
/* qdpo_update_q support (qpdo.c:549-586): Qx -= sigma x ; returns ||q + cinv Qx||inf etc. on host */
int qdev_download_vec(QpdoDev *d, int which, double *dst);   /* 0 x, 1 Qx, 2 y */
int qdev_upload_vec(QpdoDev *d, int which, const double *src);

/* configuration (environment-driven in the driver) */
int qdev_configure(QpdoDev *d, int linsolve /*0 pcg,1 dense,-1 auto*/, double pcg_tol, int pcg_maxit);
/* the linear solves stop at pcg_tol relative OR at 1e-5 eps_abs in the reference's unscaled inf-norm of the dual residual */
int qdev_set_eps_abs(QpdoDev *d, double eps_abs);
int qdev_update_q_scaled(QpdoDev *d, const double *q_unscaled, int proximal, double sigma_old, double sigma_new, double c_old,
                         double *c_new, double *cinv_new, double *norm_q);
/* Deferred read-back of the Newton step (dense solver, one GPU): qdev_newton_step returns without synchronising; its step length and the
 * lost-producer latch of the chained solves arrive with the control block of the next qdev_residuals (or qdev_finish_step): one host
 * synchronisation per loop pass instead of two.  tau_out of qdev_newton_step is then NaN. */
int qdev_dual_infeasibility_and_mu(QpdoDev *d, int do_dinf, int proximal, double sigma, double tau, double eps_dual_inf, int do_mu,
                                   double eps_abs, double theta, double delta, double mu_min, double isq_mu_min, int *is_infeasible, int *n_changed);
int qdev_set_deferred_step(QpdoDev *d, int on);
int qdev_finish_step(QpdoDev *d, int *had_pending, double *tau);
int qdev_get_stats(QpdoDev *d, QdevStats *out);
int qdev_reset_stats(QpdoDev *d);
/* HIP-event average of the Q SpMV sampled once per PCG batch during the last solve */
int qdev_get_spmv_sample(QpdoDev *d, double *avg_seconds, long *samples);
/* HIP-event samples of the A_c product inside the Schur-mode inner solves: summed seconds and algorithmic bytes */
int qdev_get_ac_sample(QpdoDev *d, double *seconds_sum, double *bytes_sum, long *samples, long *schur_passes);

/* micro-benchmark of the dominant kernel on the workspace's own matrices, timed with HIP
 * events on the backend stream.  which: 0 A (CSR m x n), 1 A' (CSR n x m), 2 Q.
 * Returns average seconds per launch and the algorithmic bytes of one launch. */
int qdev_bench_spmv(QpdoDev *d, int which, int reps, double *avg_seconds, double *alg_bytes);
/* dense LDL' factorization with the current weights, timed with HIP events; optional residual check of one solve */
int qdev_bench_dense_factor(QpdoDev *d, int reps, double *avg_seconds, double *check);
/* standalone SpMV for parity tests: y = M v */
int qdev_spmv(QpdoDev *d, int which, const double *v_host, double *y_host);
/* standalone piecewise-affine linesearch for parity tests (2m entries) */
int qdev_linesearch(QpdoDev *d, double eta, double beta, const double *delta, const double *alpha,
                    double *tau);

/* fused one-workgroup-per-QP solver for batches of small problems (qpdo_small.hip) */
struct QPDOAmdBatchItem_;
int qdev_small_eligible(long count, const void *items);
int qdev_small_batch(int device, long count, void *items, const void *settings);
const char *qdev_small_last_error(void);
double qdev_small_last_kernel_seconds(void);
/* batch stream: up to `depth` fused-kernel batches in flight, each on its own HIP stream */
void *qdev_small_stream_create(int device, int depth);
long  qdev_small_stream_submit(void *stream, long count, void *items, const void *settings);
int   qdev_small_stream_wait(void *stream, long ticket, double *kernel_seconds);
void  qdev_small_stream_destroy(void *stream);

/* ---- ONE small workspace through the fused kernel (the default path of qpdo_solve for problems whose whole state fits one
 * workgroup's LDS): the kernel runs on the workspace's own device arrays -- matrices, q, l, u as qpdo_setup scaled them, the warm start
 * as qpdo_warm_start left it -- and keeps the oracle's operation order, so its results are bit-identical to the oracle's. */
typedef struct {
    int device; void *stream;                            /* the workspace's HIP stream (hipStream_t)                        */
    int32_t n, m;
    const int32_t *Arp, *Aci; const double *Aval;        /* CSR(A)  m x n, rows column-sorted, scaled                       */
    const int32_t *Trp, *Tci; const double *Tval;        /* CSR(A') n x m                                                   */
    const int32_t *Qrp, *Qci; const double *Qval;        /* full symmetric CSR(Q), scaled (incl. the cost scaling c)        */
    const double *q, *l, *u;                             /* scaled                                                          */
    int scaled; const double *D, *Dinv, *E, *Einv; double c, cinv;
    /* the workspace's iterate state: written by an explicit qpdo_warm_start (mode 1), adjusted by qpdo_update_q, read by the solve that
     * follows (mode 2); every solve writes its final x and Qx back (qpdo_update_q reads them, qpdo.c:556-560) */
    double *st_x, *st_xbar, *st_Qx, *st_Aty, *st_y, *st_ybar, *st_Ax, *st_mu, *st_isq;
} QdevSmallView;
typedef struct {
    void *info;                                          /* QPDOInfo *: iterations, oterations, norms, objective, status    */
    long newton_passes, factor_count, ntrace;
    const void *trace;                                   /* QPDOAmdTraceRec[ntrace], valid until the next solve             */
    const double *sol_x, *sol_y, *x, *y, *dx, *dy;       /* pinned host arrays owned by the handle, valid until the next solve */
    double sigma_end, tau_end, kernel_seconds;
} QdevSmallResult;
int   qdev_small_view(QpdoDev *d, QdevSmallView *out);
int   qdev_small_resident_fits(int32_t n, int32_t m);
void *qdev_small_resident_create(const QdevSmallView *v, long trace_cap);
/* an explicit qpdo_warm_start (qpdo.c:217-299) in the oracle's operation order: x_ws / y_ws are the caller's (unscaled) vectors or NULL */
int   qdev_small_resident_warm_start(void *handle, const QdevSmallView *v, const void *settings /* QPDOSettings */, const double *x_ws,
                                     const double *y_ws, double c_const, double *objective);
/* from_state = 0: cold start (qpdo_solve's automatic warm start with NULLs) and solve in one launch; 1: solve from the workspace's state */
int   qdev_small_resident_solve(void *handle, const QdevSmallView *v, const void *settings /* QPDOSettings */, int from_state,
                                double c_const, QdevSmallResult *out);
void  qdev_small_resident_destroy(void *handle);

#ifdef __cplusplus
}
#endif
#endif
