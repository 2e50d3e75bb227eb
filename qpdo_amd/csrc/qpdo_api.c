/*
 * qpdo_api.c -- C host driver behind the qpdo.h C-ABI (the drop-in boundary).
 *
 * Mirrors the reference's public functions (src/qpdo.c) in name, argument
 * meaning, ownership and error behaviour.  The driver keeps only scalars and
 * the small host mirrors documented in include/qpdo.h; every vector and matrix
 * operation is a HIP kernel reached through the thin C-ABI of qpdo_dev.h.
 * There is no CPU fallback: if the device backend cannot be created,
 * qpdo_setup returns NULL.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <omp.h>
#include <pthread.h>

#include "qpdo.h"
#include "qpdo_amd_ext.h"
#include "qpdo_dev.h"
#include "pass_decision.h"

struct QPDO_TIMER { struct timespec tic, toc; };   /* reference include/util.h:98-104 */

struct QPDOBackend {
    QpdoDev *dev;
    int reset_newton;             /* reference chol->reset_newton (types.h:132) */
    int fix_status_reset;
    QPDOAmdTraceRec *trace; long ntrace, captrace;
    long newton_passes;
    /* small problems (the packed Newton matrix fits one workgroup's LDS, n <= QPDO_SMALL_FUSED_MAX_N): qpdo_solve runs as ONE launch of
     * the fused kernel on the workspace's own device arrays (qdev_small_resident_*), in the oracle's operation order */
    void *small;                  /* resident fused-solve handle, or NULL: generic multi-kernel path */
    int ws_state;                 /* the workspace's device vectors hold the warm-started state of the pending solve (0: qpdo_solve's automatic
                                   * cold start of a fused-path workspace, which is part of the solve's launch) */
    int auto_ws;                  /* qpdo_warm_start is being called by qpdo_solve itself (qpdo.c:312-314) */
    long fused_solves, fused_factor_count; double fused_kernel_s; int last_fused;
};

#define c_max(a, b) (((a) > (b)) ? (a) : (b))
#define c_min(a, b) (((a) < (b)) ? (a) : (b))
#define c_absval(x) (((x) < 0) ? -(x) : (x))

#define QPDO_PRINT(...) do { printf(__VA_ARGS__); fflush(stdout); } while (0)
#define QPDO_EPRINT(...) do { printf("ERROR in %s: ", __func__); printf(__VA_ARGS__); printf("\n"); fflush(stdout); } while (0)

/* ---- timing (reference src/util.c:245-264) ---------------------------------- */
static void tic(QPDOTimer *t) { clock_gettime(CLOCK_MONOTONIC, &t->tic); }
static c_float toc(QPDOTimer *t) {
    clock_gettime(CLOCK_MONOTONIC, &t->toc);
    return (c_float)(t->toc.tv_sec - t->tic.tv_sec) + 1e-9 * (c_float)(t->toc.tv_nsec - t->tic.tv_nsec);
}

/* ---- status strings (reference src/util.c:50-91) ------------------------------ */
static void update_status(QPDOInfo *info, c_int status_val) {
    const char *s;
    info->status_val = status_val;
    switch (status_val) {
        case QPDO_SOLVED: s = "solved"; break;
        case QPDO_DUAL_TERMINATED: s = "dual terminated"; break;
        case QPDO_PRIMAL_INFEASIBLE: s = "primal infeasible"; break;
        case QPDO_DUAL_INFEASIBLE: s = "dual infeasible"; break;
        case QPDO_PRIMAL_DUAL_INFEASIBLE: s = "primal-dual infeasible"; break;
        case QPDO_MAX_TIME_REACHED: s = "max time exceeded"; break;
        case QPDO_MAX_ITER_REACHED: s = "maximum iterations reached"; break;
        case QPDO_UNSOLVED: s = "unsolved"; break;
        case QPDO_ERROR: s = "error"; break;
        default: s = "unrecognised status value"; break;
    }
    strncpy(info->status, s, sizeof(info->status) - 1);
    info->status[sizeof(info->status) - 1] = '\0';
}

/* ---- validation (reference src/validate.c:9-170) ------------------------------- */
static int validate_data(const QPDOData *data) {
    if (!data) { QPDO_EPRINT("Missing data"); return 0; }
    for (size_t j = 0; j < data->m; j++)
        if (data->l[j] > data->u[j]) {
            QPDO_EPRINT("Lower bound at index %d is greater than upper bound: %.4e > %.4e", (int)j, data->l[j], data->u[j]);
            return 0;
        }
    return 1;
}
static int validate_settings(const QPDOSettings *s) {
    if (!s) { QPDO_EPRINT("Missing settings!"); return 0; }
    if (s->max_iter <= 0) { QPDO_EPRINT("max_iter must be positive"); return 0; }
    if (s->inner_max_iter <= 0) { QPDO_EPRINT("inner_max_iter must be positive"); return 0; }
    if (s->eps_abs <= 0) { QPDO_EPRINT("eps_abs must be positive"); return 0; }
    if (s->eps_abs_in <= 0) { QPDO_EPRINT("eps_abs_in must be positive"); return 0; }
    if (s->eps_prim_inf < 0) { QPDO_EPRINT("eps_prim_inf must be nonnegative"); return 0; }
    if (s->eps_dual_inf < 0) { QPDO_EPRINT("eps_dual_inf must be nonnegative"); return 0; }
    if (s->rho <= 0 || s->rho >= 1) { QPDO_EPRINT("rho must be positive and smaller than 1"); return 0; }
    if (s->theta <= 0 || s->theta > 1) { QPDO_EPRINT("theta must be positive and smaller than or equal to 1"); return 0; }
    if (s->delta <= 0 || s->delta >= 1) { QPDO_EPRINT("delta must be positive and smaller than 1"); return 0; }
    if (s->mu_min <= 0) { QPDO_EPRINT("mu_min must be positive"); return 0; }
    if ((s->proximal != 0) && (s->proximal != 1)) { QPDO_EPRINT("proximal must be either 0 or 1"); return 0; }
    if (s->sigma_init <= 0) { QPDO_EPRINT("sigma_init must be positive"); return 0; }
    if (s->sigma_upd <= 0 || s->sigma_upd > 1) { QPDO_EPRINT("sigma_upd must be positive and smaller than or equal to 1"); return 0; }
    if (s->sigma_min > s->sigma_init) { QPDO_EPRINT("sigma_min must be smaller than or equal to sigma_init"); return 0; }
    if (s->scaling < 0) { QPDO_EPRINT("scaling must be nonnegative"); return 0; }
    if (s->verbose < 0) { QPDO_EPRINT("verbose must be nonnegative"); return 0; }
    if (s->print_interval < 0) { QPDO_EPRINT("print_interval must be nonnegative"); return 0; }
    if (s->reset_newton_iter < 0) { QPDO_EPRINT("reset_newton_iter must be nonnegative"); return 0; }
    return 1;
}

/* reference src/qpdo.c:24-44 with include/constants.h:44-69 */
void qpdo_set_default_settings(QPDOSettings *s) {
    s->max_time = QPDO_INFTY;
    s->max_iter = 10000;
    s->inner_max_iter = 1000;
    s->eps_abs = 1e-6;
    s->eps_abs_in = 1e0;
    s->eps_prim_inf = 1e-6;
    s->eps_dual_inf = 1e-6;
    s->rho = 0.1;
    s->theta = 0.25;
    s->delta = 1e-2;
    s->mu_min = 1e-9;
    s->proximal = 1;
    s->sigma_init = 1e-3;
    s->sigma_upd = 1e-1;
    s->sigma_min = 1e-7;
    s->scaling = 10;
    s->verbose = 1;
    s->print_interval = 1;
    s->reset_newton_iter = 1000;
}
static QPDOSettings *copy_settings(const QPDOSettings *s) {   /* src/util.c:21-45 */
    QPDOSettings *n = malloc(sizeof(QPDOSettings));
    if (n) *n = *s;
    return n;
}

/* The batch stream (qpdo_amd_batch_stream_*) keeps up to `depth` fused-kernel launches in flight on separate HIP streams.  The
 * runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams that share a queue serialise: measured
 * on MI355X, 16 batches of 4096 C3 QPs at depth 12: 7.0 k QP/s on 4 queues, 10.1 k on 16.  The variable is read when the HIP
 * runtime initialises (the first HIP call of the process).  This library does NOT touch its host's environment: the caller sets
 * GPU_MAX_HW_QUEUES before its first HIP call (the Python front end does so before it loads the library), and
 * qpdo_amd_batch_stream_create says so once when a deep stream is created without it. */
static void warn_hw_queues_once(int depth) {
    static int warned = 0;
    const char *v = getenv("GPU_MAX_HW_QUEUES");
    const int have = (v && *v) ? atoi(v) : 4;
    if (depth > have && !__sync_lock_test_and_set(&warned, 1))
        fprintf(stderr, "qpdo_amd: batch stream of depth %d on %d HIP hardware queues (GPU_MAX_HW_QUEUES%s): streams that share a queue "
                        "serialise; export GPU_MAX_HW_QUEUES=%d before the process's first HIP call\n", depth, have, (v && *v) ? "" : " unset", depth > 16 ? depth : 16);
}

static int env_int_early(const char *name, int dflt) { const char *v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
static double wall_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
/* ---- matrix intake: CSC (int32 or int64 indices) -> int32 CSR triples ------------- */
typedef struct { int32_t nrows, ncols; int64_t nnz; int32_t *rp, *ci; double *val; } HostCsr;
static void host_csr_free(HostCsr *h) { free(h->rp); free(h->ci); free(h->val); memset(h, 0, sizeof(*h)); }
static inline int64_t idx_at(const void *a, int itype, int64_t k) {
    return itype == 0 ? (int64_t)((const int32_t *)a)[k] : (int64_t)((const int64_t *)a)[k];
}
static int sparse_ok(const cholmod_sparse *M) {
    if (!M || !M->p) return 0;
    if (M->itype != 0 && M->itype != 2) return 0;      /* CHOLMOD_INT / CHOLMOD_LONG */
    if (M->xtype != 1 || M->dtype != 0) return 0;      /* real, double */
    if (!M->packed && M->nz) return 0;                 /* unpacked storage not supported */
    if (M->nrow >= (size_t)INT32_MAX || M->ncol >= (size_t)INT32_MAX) return 0;
    if (idx_at(M->p, M->itype, (int64_t)M->ncol) >= (int64_t)INT32_MAX) return 0;
    return 1;
}
/* release the staging copies on a detached thread (falls back to freeing in place) */
typedef struct { HostCsr h[3]; } FreeJob;
static void *free_job(void *arg) { FreeJob *j = (FreeJob *)arg; for (int i = 0; i < 3; i++) host_csr_free(&j->h[i]); free(j); return NULL; }
static void host_csr_free_async(HostCsr *a, HostCsr *b, HostCsr *c) {
    FreeJob *j = malloc(sizeof(FreeJob));
    pthread_t th; pthread_attr_t at;
    if (j) {
        j->h[0] = *a; j->h[1] = *b; j->h[2] = *c;
        pthread_attr_init(&at); pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        if (pthread_create(&th, &at, free_job, j) == 0) { pthread_attr_destroy(&at); memset(a, 0, sizeof(*a)); memset(b, 0, sizeof(*b)); memset(c, 0, sizeof(*c)); return; }
        pthread_attr_destroy(&at); free(j);
    }
    host_csr_free(a); host_csr_free(b); host_csr_free(c);
}
/* host threads for the conversions (OpenMP; QPDO_SETUP_THREADS, default min(16, cores)) */
static int conv_threads(void) {
    int t = env_int_early("QPDO_SETUP_THREADS", 0);
    if (t <= 0) { t = omp_get_max_threads(); if (t > 16) t = 16; }
    return t < 1 ? 1 : t;
}
/* the CSC arrays of an r x c matrix are the CSR arrays of its c x r transpose */
static int csc_as_csr_of_transpose(const cholmod_sparse *M, HostCsr *out) {
    const int64_t nc = (int64_t)M->ncol, nnz = idx_at(M->p, M->itype, nc);
    memset(out, 0, sizeof(*out));
    out->nrows = (int32_t)M->ncol; out->ncols = (int32_t)M->nrow; out->nnz = nnz;
    out->rp = malloc(((size_t)nc + 1) * sizeof(int32_t));
    out->ci = malloc((size_t)(nnz ? nnz : 1) * sizeof(int32_t));
    out->val = malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
    if (!out->rp || !out->ci || !out->val) { host_csr_free(out); return 0; }
    const int T = conv_threads();
    const double *x = M->x;
#pragma omp parallel num_threads(T) if (nnz > 200000)
    {
#pragma omp for schedule(static) nowait
        for (int64_t j = 0; j <= nc; j++) out->rp[j] = (int32_t)idx_at(M->p, M->itype, j);
#pragma omp for schedule(static)
        for (int64_t k = 0; k < nnz; k++) { out->ci[k] = (int32_t)idx_at(M->i, M->itype, k); out->val[k] = x[k]; }
    }
    return 1;
}
/* CSR of the r x c matrix itself, rows column-sorted.  The columns are cut into T contiguous ranges of equal
 * nnz; thread t counts its range per row, a prefix over (row, t) gives every thread its first slot in each row,
 * and the threads scatter their ranges in column order -- the result is the one a serial counting pass gives. */
static int csc_to_csr(const cholmod_sparse *M, HostCsr *out) {
    const int64_t nr = (int64_t)M->nrow, nc = (int64_t)M->ncol, nnz = idx_at(M->p, M->itype, nc);
    memset(out, 0, sizeof(*out));
    out->nrows = (int32_t)nr; out->ncols = (int32_t)nc; out->nnz = nnz;
    int T = conv_threads();
    if ((nnz < 200000 && env_int_early("QPDO_SETUP_THREADS", 0) <= 0) || (int64_t)T * nr > 400000000LL) T = 1;
    out->rp = calloc((size_t)nr + 1, sizeof(int32_t));
    out->ci = malloc((size_t)(nnz ? nnz : 1) * sizeof(int32_t));
    out->val = malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
    int32_t *cnt = calloc((size_t)T * (size_t)(nr ? nr : 1), sizeof(int32_t));     /* cnt[t*nr + i] */
    int64_t *cut = malloc(((size_t)T + 1) * sizeof(int64_t));                          /* column ranges */
    if (!out->rp || !out->ci || !out->val || !cnt || !cut) { free(cnt); free(cut); host_csr_free(out); return 0; }
    cut[0] = 0; cut[T] = nc;
    for (int t = 1; t < T; t++) {            /* first column whose start offset reaches t/T of the entries */
        const int64_t target = nnz / T * t;
        int64_t lo = cut[t - 1], hi = nc;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (idx_at(M->p, M->itype, mid) < target) lo = mid + 1; else hi = mid; }
        cut[t] = lo;
    }
    const double *x = M->x;
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        int32_t *c = cnt + (size_t)t * (size_t)nr;
        const int64_t kb = idx_at(M->p, M->itype, cut[t]), ke = idx_at(M->p, M->itype, cut[t + 1]);
        for (int64_t k = kb; k < ke; k++) c[idx_at(M->i, M->itype, k)]++;
#pragma omp barrier
#pragma omp for schedule(static)
        for (int64_t i = 0; i < nr; i++) {   /* row totals; cnt becomes the offset of thread t inside row i */
            int32_t run = 0;
            for (int tt = 0; tt < T; tt++) { const int32_t v = cnt[(size_t)tt * (size_t)nr + i]; cnt[(size_t)tt * (size_t)nr + i] = run; run += v; }
            out->rp[i + 1] = run;
        }
#pragma omp single
        for (int64_t i = 0; i < nr; i++) out->rp[i + 1] += out->rp[i];
        for (int64_t j = cut[t]; j < cut[t + 1]; j++) {
            const int64_t b = idx_at(M->p, M->itype, j), e = idx_at(M->p, M->itype, j + 1);
            for (int64_t k = b; k < e; k++) {
                const int64_t i = idx_at(M->i, M->itype, k);
                const int32_t s = out->rp[i] + c[i]++;
                out->ci[s] = (int32_t)j; out->val[s] = x[k];
            }
        }
    }
    free(cnt); free(cut);
    return 1;
}
/* full symmetric CSR of Q from one stored triangle (stype -1 lower, +1 upper) or from full storage (stype 0).
 * Lower stored: row i = CSR(L) row i (columns <= i, ascending) followed by column i of L below the diagonal
 * (ascending rows = ascending columns of the mirror).  Upper stored: column i of U up to the diagonal, then
 * CSR(U) row i beyond it.  Rows come out column-sorted for column-sorted input; entries on the wrong side of
 * the stored triangle are ignored, as CHOLMOD does for stype != 0. */
static int sym_to_full_csr(const cholmod_sparse *Q, HostCsr *out) {
    const int64_t n = (int64_t)Q->ncol;
    const int st = Q->stype;
    if (st == 0) return csc_as_csr_of_transpose(Q, out);      /* symmetric: transpose == itself */
    HostCsr R;                                                /* CSR of the stored matrix (all its entries) */
    if (!csc_to_csr(Q, &R)) return 0;
    memset(out, 0, sizeof(*out));
    out->nrows = out->ncols = (int32_t)n;
    out->rp = calloc((size_t)n + 1, sizeof(int32_t));
    if (!out->rp) { host_csr_free(&R); return 0; }
    const int T = conv_threads();
    const int par = R.nnz > 200000;
    /* an entry (i,j) of the stored matrix counts when it is on the stored side or the diagonal */
#define KEEP_CSR(i, j) ((st < 0) ? ((j) <= (i)) : ((j) >= (i)))      /* CSR row i keeps (i,j): own triangle incl. diagonal */
#define KEEP_MIR(i, j) ((st < 0) ? ((i) > (j)) : ((i) < (j)))        /* CSC column j entry (i,j) mirrors into row j as (j,i) */
#pragma omp parallel for schedule(static) num_threads(T) if (par)
    for (int64_t i = 0; i < n; i++) {
        int32_t c = 0;
        for (int32_t k = R.rp[i]; k < R.rp[i + 1]; k++) c += KEEP_CSR(i, (int64_t)R.ci[k]);
        const int64_t b = idx_at(Q->p, Q->itype, i), e = idx_at(Q->p, Q->itype, i + 1);
        for (int64_t k = b; k < e; k++) c += KEEP_MIR(idx_at(Q->i, Q->itype, k), i);
        out->rp[i + 1] = c;
    }
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) { total += out->rp[i + 1]; if (total >= (int64_t)INT32_MAX) { host_csr_free(&R); host_csr_free(out); return 0; } out->rp[i + 1] = (int32_t)total; }
    out->nnz = total;
    out->ci = malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    out->val = malloc((size_t)(total ? total : 1) * sizeof(double));
    if (!out->ci || !out->val) { host_csr_free(&R); host_csr_free(out); return 0; }
    const double *x = Q->x;
#pragma omp parallel for schedule(static) num_threads(T) if (par)
    for (int64_t i = 0; i < n; i++) {
        int32_t s = out->rp[i];
        const int64_t b = idx_at(Q->p, Q->itype, i), e = idx_at(Q->p, Q->itype, i + 1);
        if (st < 0) {       /* columns <= i from the CSR row, then the mirrored column below the diagonal */
            for (int32_t k = R.rp[i]; k < R.rp[i + 1]; k++) if (KEEP_CSR(i, (int64_t)R.ci[k])) { out->ci[s] = R.ci[k]; out->val[s] = R.val[k]; s++; }
            for (int64_t k = b; k < e; k++) { const int64_t r = idx_at(Q->i, Q->itype, k); if (KEEP_MIR(r, i)) { out->ci[s] = (int32_t)r; out->val[s] = x[k]; s++; } }
        } else {            /* mirrored column above the diagonal, then columns >= i from the CSR row */
            for (int64_t k = b; k < e; k++) { const int64_t r = idx_at(Q->i, Q->itype, k); if (KEEP_MIR(r, i)) { out->ci[s] = (int32_t)r; out->val[s] = x[k]; s++; } }
            for (int32_t k = R.rp[i]; k < R.rp[i + 1]; k++) if (KEEP_CSR(i, (int64_t)R.ci[k])) { out->ci[s] = R.ci[k]; out->val[s] = R.val[k]; s++; }
        }
    }
#undef KEEP_CSR
#undef KEEP_MIR
    host_csr_free(&R);
    return 1;
}

/* ---- small host vector helpers (operation order of reference src/lin_alg.c) --------- */
static c_float vec_norm_inf(const c_float *a, size_t n) {
    c_float mx = 0.0;
    for (size_t i = 0; i < n; i++) { c_float s = c_absval(a[i]); mx = s > mx ? s : mx; }
    return mx;
}
static c_float *vec_dup(const c_float *a, size_t n) {
    c_float *b = malloc((n ? n : 1) * sizeof(c_float));
    if (b && n) memcpy(b, a, n * sizeof(c_float));
    return b;
}

/* Distributed configuration for the workspaces created next in this process (read once, under the lock, by each
 * qpdo_setup).  An RCCL unique id creates exactly ONE communicator: the qpdo_setup that consumes it clears it, and a
 * further distributed setup without a fresh qpdo_amd_dist_config fails with a message instead of hanging in
 * ncclCommInitRank on a spent id.  The host-callback mode has no such limit. */
static QdevDist g_dist = {0, 1, 0, 0, 0, 0, NULL, NULL, {0}, 0};
static int g_dist_has_id = 0, g_dist_id_spent = 0;
static pthread_mutex_t g_dist_mu = PTHREAD_MUTEX_INITIALIZER;
int qpdo_amd_dist_config(int rank, int world, const void *rccl_unique_id, qpdo_amd_allreduce_fn fn, void *ctx) {
    if (world < 1 || rank < 0 || rank >= world) return -1;
    if (world > 1 && !fn && !rccl_unique_id) return -1;
    pthread_mutex_lock(&g_dist_mu);
    memset(&g_dist, 0, sizeof(g_dist));
    g_dist.rank = rank; g_dist.world = world; g_dist.fn = (qdev_allreduce_fn)fn; g_dist.ctx = ctx;
    g_dist_has_id = 0; g_dist_id_spent = 0;
    if (rccl_unique_id && !fn) { memcpy(g_dist.nccl_id, rccl_unique_id, 128); g_dist_has_id = 1; }
    /* world == 1 with a communicator: a forced single-rank partition -- every collective call site runs (on one GPU) */
    g_dist.force = (world == 1 && (fn || rccl_unique_id)) ? 1 : 0;
    pthread_mutex_unlock(&g_dist_mu);
    return 0;
}
/* snapshot for one qpdo_setup; returns 0, or -1 when the RCCL id of the configuration has already been used */
static int dist_take(QdevDist *out) {
    int rc = 0;
    pthread_mutex_lock(&g_dist_mu);
    *out = g_dist;
    if ((g_dist.world > 1 || g_dist.force) && !g_dist.fn) {
        if (!g_dist_has_id || g_dist_id_spent) rc = -1;
        else g_dist_id_spent = 1;
    }
    pthread_mutex_unlock(&g_dist_mu);
    return rc;
}
int qpdo_amd_dist_unique_id(void *out128) { return qdev_rccl_unique_id(out128); }

static int env_int(const char *name, int dflt) { const char *v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
static double env_double(const char *name, double dflt) { const char *v = getenv(name); return (v && *v) ? atof(v) : dflt; }

static QPDOAmdTraceRec *trace_push(struct QPDOBackend *b) {
    if (b->ntrace == b->captrace) {
        long cap = b->captrace ? 2 * b->captrace : 256;
        QPDOAmdTraceRec *t = realloc(b->trace, (size_t)cap * sizeof(*t));
        if (!t) return NULL;
        b->trace = t; b->captrace = cap;
    }
    QPDOAmdTraceRec *r = &b->trace[b->ntrace++];
    memset(r, 0, sizeof(*r));
    r->factor_branch = -1;
    return r;
}

/* apply this call's Ruiz/cost scaling to the host-side q,l,u mirrors and install the totals */
static int install_scaling(QPDOWorkspace *work) {
    size_t n = work->data->n, m = work->data->m;
    QPDOScaling *sc = work->scaling;
    for (size_t i = 0; i < n; i++) sc->Dinv[i] = (c_float)1.0 / sc->D[i];
    for (size_t i = 0; i < m; i++) sc->Einv[i] = (c_float)1.0 / sc->E[i];
    sc->cinv = (c_float)1.0 / sc->c;
    return qdev_set_scaling(work->chol->dev, 1, sc->D, sc->Dinv, sc->E, sc->Einv, sc->c, sc->cinv);
}

/* ---- qpdo_setup (reference src/qpdo.c:49-212) ----------------------------------------- */
QPDOWorkspace *qpdo_setup(const QPDOData *data, const QPDOSettings *settings) {
    if (!validate_data(data)) { QPDO_EPRINT("Data validation returned failure"); return QPDO_NULL; }
    if (!validate_settings(settings)) { QPDO_EPRINT("Settings validation returned failure"); return QPDO_NULL; }
    if (!sparse_ok(data->Q) || !sparse_ok(data->A)) { QPDO_EPRINT("unsupported sparse matrix storage"); return QPDO_NULL; }
    if (data->Q->nrow != data->n || data->Q->ncol != data->n || data->A->nrow != data->m || data->A->ncol != data->n) {
        QPDO_EPRINT("matrix dimensions do not match n, m"); return QPDO_NULL;
    }
    QPDOWorkspace *work = calloc(1, sizeof(QPDOWorkspace));
    if (!work) { QPDO_EPRINT("allocating work failure"); return QPDO_NULL; }
    work->timer = malloc(sizeof(QPDOTimer));
    if (!work->timer) { free(work); return QPDO_NULL; }
    tic(work->timer);

    const size_t n = data->n, m = data->m;
    work->settings = copy_settings(settings);
    work->chol = calloc(1, sizeof(struct QPDOBackend));
    work->data = calloc(1, sizeof(QPDOData));
    work->solution = calloc(1, sizeof(QPDOSolution));
    work->info = calloc(1, sizeof(QPDOInfo));
    if (!work->settings || !work->chol || !work->data || !work->solution || !work->info) goto fail;
    work->sqrt_delta = sqrt(work->settings->delta);
    work->sigma = work->settings->sigma_init;
    work->data->n = n; work->data->m = m; work->data->c = data->c;
    work->data->q = vec_dup(data->q, n);
    work->data->l = vec_dup(data->l, m);
    work->data->u = vec_dup(data->u, m);
    work->x = calloc(n ? n : 1, sizeof(c_float));
    work->y = calloc(m ? m : 1, sizeof(c_float));
    work->dx = calloc(n ? n : 1, sizeof(c_float));
    work->dy = calloc(m ? m : 1, sizeof(c_float));
    work->solution->x = calloc(n ? n : 1, sizeof(c_float));
    work->solution->y = calloc(m ? m : 1, sizeof(c_float));
    if (!work->data->q || !work->data->l || !work->data->u || !work->x || !work->y || !work->dx || !work->dy ||
        !work->solution->x || !work->solution->y) goto fail;
    work->initialized = 0;
    work->n_mu_changed = 0;
    work->chol->reset_newton = 1;
    work->chol->fix_status_reset = env_int("QPDO_FIX_STATUS_RESET", 0);

    int on_device = 0;
    {   /* matrices to the device.  One GPU (the default): the caller's CSC arrays are uploaded as they are -- they ARE CSR(A') -- and
         * CSR(A) and the full symmetric CSR(Q) are built on the device (qdev_create_csc: the same arrays, bit for bit, that the host
         * conversions below produce; QPDO_SETUP_HOST=1 keeps those).  A row-partitioned workspace cuts its slices on the host. */
        QdevDist dd0;
        pthread_mutex_lock(&g_dist_mu); dd0 = g_dist; pthread_mutex_unlock(&g_dist_mu);
        if (dd0.world <= 1 && !dd0.force && !env_int("QPDO_SETUP_HOST", 0)) {
            const int prof = env_int("QPDO_SETUP_PROF", 0);
            const double t0 = wall_now();
            int ndev = qdev_device_count();
            if (ndev <= 0) { QPDO_EPRINT("no HIP device available (this library has no CPU path)"); goto fail; }
            int device = env_int("QPDO_DEVICE", env_int("LOCAL_RANK", 0)) % ndev;
            const cholmod_sparse *A = data->A, *Q = data->Q;
            QdevCsc a = {(int32_t)A->nrow, (int32_t)A->ncol, idx_at(A->p, A->itype, (int64_t)A->ncol), A->itype, A->p, A->i, (const double *)A->x, 0};
            QdevCsc qq = {(int32_t)Q->nrow, (int32_t)Q->ncol, idx_at(Q->p, Q->itype, (int64_t)Q->ncol), Q->itype, Q->p, Q->i, (const double *)Q->x, Q->stype};
            if (qdev_create_csc(&work->chol->dev, device, (int32_t)n, (int32_t)m, &a, &qq, work->data->q, work->data->l, work->data->u)) {
                QPDO_EPRINT("device backend: %s", qdev_last_error()); goto fail;
            }
            if (prof) fprintf(stderr, "[setup] device create     %.3f s (upload of the CSC arrays, transpositions and slab tables on the device)\n", wall_now() - t0);
            const char *ls = getenv("QPDO_LINSOLVE");
            int mode = -1;
            if (ls && !strcmp(ls, "pcg")) mode = 0; else if (ls && !strcmp(ls, "dense")) mode = 1; else if (ls && !strcmp(ls, "band")) mode = 3;
            if (qdev_configure(work->chol->dev, mode, env_double("QPDO_PCG_TOL", 0.0), env_int("QPDO_PCG_MAXIT", 0))) { QPDO_EPRINT("device backend: %s", qdev_last_error()); goto fail; }
            on_device = 1;
        }
    }
    if (!on_device) {   /* host conversions (row-partitioned workspaces; QPDO_SETUP_HOST=1) */
        HostCsr Ar = {0}, At = {0}, Qf = {0};
        const int prof = env_int("QPDO_SETUP_PROF", 0);
        double t0 = wall_now();
        int ok = csc_to_csr(data->A, &Ar);
        if (prof) { fprintf(stderr, "[setup] CSR(A)            %.3f s\n", wall_now() - t0); t0 = wall_now(); }
        ok = ok && csc_as_csr_of_transpose(data->A, &At);
        if (prof) { fprintf(stderr, "[setup] CSR(A') narrowing %.3f s\n", wall_now() - t0); t0 = wall_now(); }
        ok = ok && sym_to_full_csr(data->Q, &Qf);
        if (prof) { fprintf(stderr, "[setup] full CSR(Q)       %.3f s\n", wall_now() - t0); t0 = wall_now(); }
        if (!ok) { host_csr_free(&Ar); host_csr_free(&At); host_csr_free(&Qf); QPDO_EPRINT("matrix conversion failed"); goto fail; }
        QdevCsr a = {Ar.nrows, Ar.ncols, Ar.nnz, Ar.rp, Ar.ci, Ar.val};
        QdevCsr t = {At.nrows, At.ncols, At.nnz, At.rp, At.ci, At.val};
        QdevCsr qf = {Qf.nrows, Qf.ncols, Qf.nnz, Qf.rp, Qf.ci, Qf.val};
        int device = env_int("QPDO_DEVICE", env_int("LOCAL_RANK", 0));
        int ndev = qdev_device_count();
        if (ndev <= 0) {
            host_csr_free(&Ar); host_csr_free(&At); host_csr_free(&Qf);
            QPDO_EPRINT("no HIP device available (this library has no CPU path)"); goto fail;
        }
        device = device % ndev;
        int rc;
        QdevDist dd;
        if (dist_take(&dd)) {
            host_csr_free(&Ar); host_csr_free(&At); host_csr_free(&Qf);
            QPDO_EPRINT("the RCCL unique id of qpdo_amd_dist_config has already created a communicator: call qpdo_amd_dist_config "
                        "with a fresh id before setting up another row-partitioned workspace"); goto fail;
        }
        if (dd.world <= 1 && !dd.force) {
            rc = qdev_create(&work->chol->dev, device, (int32_t)n, (int32_t)m, &a, &t, &qf, work->data->q, work->data->l, work->data->u);
        } else {
            /* row partition: rows [m0, m0+mloc) of A, the same columns of A', rows [n0, n0+nloc) of Q */
            const int64_t rpm = ((int64_t)m + dd.world - 1) / dd.world, rpn = ((int64_t)n + dd.world - 1) / dd.world;
            int64_t m0 = (int64_t)dd.rank * rpm; if (m0 > (int64_t)m) m0 = (int64_t)m;
            int64_t m1 = m0 + rpm; if (m1 > (int64_t)m) m1 = (int64_t)m;
            int64_t n0 = (int64_t)dd.rank * rpn; if (n0 > (int64_t)n) n0 = (int64_t)n;
            int64_t n1 = n0 + rpn; if (n1 > (int64_t)n) n1 = (int64_t)n;
            dd.m0 = (int32_t)m0; dd.mloc = (int32_t)(m1 - m0); dd.n0 = (int32_t)n0; dd.nloc = (int32_t)(n1 - n0);
            int32_t *arp = malloc(((size_t)dd.mloc + 1) * sizeof(int32_t)), *qrp = malloc(((size_t)dd.nloc + 1) * sizeof(int32_t));
            int32_t *trp = malloc(((size_t)n + 1) * sizeof(int32_t));
            int64_t tn = 0;
            for (int64_t j = 0; j < (int64_t)n; j++) for (int32_t k = At.rp[j]; k < At.rp[j + 1]; k++) tn += (At.ci[k] >= m0 && At.ci[k] < m1);
            int32_t *tci = malloc((size_t)(tn ? tn : 1) * sizeof(int32_t)); double *tval = malloc((size_t)(tn ? tn : 1) * sizeof(double));
            if (!arp || !qrp || !trp || !tci || !tval) { free(arp); free(qrp); free(trp); free(tci); free(tval); rc = -1; }
            else {
                for (int32_t i = 0; i <= dd.mloc; i++) arp[i] = Ar.rp[m0 + i] - Ar.rp[m0];
                for (int32_t i = 0; i <= dd.nloc; i++) qrp[i] = Qf.rp[n0 + i] - Qf.rp[n0];
                int64_t pos = 0;
                for (int64_t j = 0; j < (int64_t)n; j++) {
                    trp[j] = (int32_t)pos;
                    for (int32_t k = At.rp[j]; k < At.rp[j + 1]; k++)
                        if (At.ci[k] >= m0 && At.ci[k] < m1) { tci[pos] = At.ci[k] - (int32_t)m0; tval[pos] = At.val[k]; pos++; }
                }
                trp[n] = (int32_t)pos;
                QdevCsr al = {dd.mloc, (int32_t)n, arp[dd.mloc], arp, Ar.ci + Ar.rp[m0], Ar.val + Ar.rp[m0]};
                QdevCsr tl = {(int32_t)n, dd.mloc, pos, trp, tci, tval};
                QdevCsr ql = {dd.nloc, (int32_t)n, qrp[dd.nloc], qrp, Qf.ci + Qf.rp[n0], Qf.val + Qf.rp[n0]};
                rc = qdev_create_dist(&work->chol->dev, device, (int32_t)n, (int32_t)m, &al, &tl, &qf, &ql, work->data->q, work->data->l, work->data->u, &dd);
                free(arp); free(qrp); free(trp); free(tci); free(tval);
            }
        }
        if (prof) { fprintf(stderr, "[setup] device create     %.3f s (upload, slab tables)\n", wall_now() - t0); t0 = wall_now(); }
        host_csr_free_async(&Ar, &At, &Qf);       /* unmapping several GB takes ~0.5 s: off the caller's path */
        if (prof) { fprintf(stderr, "[setup] host free         %.3f s\n", wall_now() - t0); }
        if (rc) { QPDO_EPRINT("device backend: %s", qdev_last_error()); goto fail; }
        const char *ls = getenv("QPDO_LINSOLVE");
        int mode = -1;
        if (ls && !strcmp(ls, "pcg")) mode = 0; else if (ls && !strcmp(ls, "dense")) mode = 1; else if (ls && !strcmp(ls, "band")) mode = 3;
        if (qdev_configure(work->chol->dev, mode, env_double("QPDO_PCG_TOL", 0.0), env_int("QPDO_PCG_MAXIT", 0))) { QPDO_EPRINT("device backend: %s", qdev_last_error()); goto fail; }
    }

    if (settings->scaling) {
        work->scaling = malloc(sizeof(QPDOScaling));
        if (!work->scaling) goto fail;
        work->scaling->D = calloc(n ? n : 1, sizeof(c_float));
        work->scaling->Dinv = calloc(n ? n : 1, sizeof(c_float));
        work->scaling->E = calloc(m ? m : 1, sizeof(c_float));
        work->scaling->Einv = calloc(m ? m : 1, sizeof(c_float));
        if (!work->scaling->D || !work->scaling->Dinv || !work->scaling->E || !work->scaling->Einv) goto fail;
        /* scale_data (scaling.c:24-91): A, Q, q on the device; l, u through the host mirrors */
        const double ts0 = wall_now();
        if (qdev_scale_data(work->chol->dev, (int)settings->scaling, 0, work->scaling->D, work->scaling->E, &work->scaling->c)) goto fail_dev;
        if (env_int("QPDO_SETUP_PROF", 0)) fprintf(stderr, "[setup] device scaling    %.3f s\n", wall_now() - ts0);
        if (install_scaling(work)) goto fail_dev;
        if (qdev_download_q(work->chol->dev, work->data->q)) goto fail_dev;
        for (size_t i = 0; i < m; i++) { work->data->l[i] = work->scaling->E[i] * work->data->l[i]; work->data->u[i] = work->scaling->E[i] * work->data->u[i]; }
        if (qdev_upload_bounds(work->chol->dev, work->data->l, work->data->u)) goto fail_dev;
        {   /* || Dinv q || (qpdo.c:163-165) */
            c_float mx = 0.0;
            for (size_t i = 0; i < n; i++) { c_float s = c_absval(work->scaling->Dinv[i] * work->data->q[i]); mx = s > mx ? s : mx; }
            work->norm_q = mx;
        }
    } else {
        work->scaling = QPDO_NULL;
        if (qdev_set_scaling(work->chol->dev, 0, NULL, NULL, NULL, NULL, 1.0, 1.0)) goto fail_dev;
        work->norm_q = vec_norm_inf(work->data->q, n);
    }
    {   /* small problems: the fused one-launch solve (QPDO_SMALL_FUSED=0 keeps the generic path; an explicit QPDO_LINSOLVE asks for one
         * of the generic path's solvers; a row-partitioned workspace is never small) */
        const char *ls = getenv("QPDO_LINSOLVE");
        const int want = env_int("QPDO_SMALL_FUSED", 1) && !(ls && *ls && strcmp(ls, "auto"));
        if (want && (long)n <= env_int("QPDO_SMALL_FUSED_MAX_N", 160) && qdev_small_resident_fits((int32_t)n, (int32_t)m)) {
            QdevSmallView v;
            if (qdev_small_view(work->chol->dev, &v) == 0) {
                long cap = (long)settings->max_iter < 16384 ? (long)settings->max_iter : 16384;
                work->chol->small = qdev_small_resident_create(&v, cap);      /* NULL: not fatal, the generic path serves the workspace */
            }
        }
    }
    update_status(work->info, QPDO_UNSOLVED);
    work->info->solve_time = 0.0;
    work->info->run_time = 0.0;
    work->info->setup_time = toc(work->timer);
    return work;

fail_dev:
    QPDO_EPRINT("device backend: %s", qdev_last_error());
fail:
    qpdo_cleanup(work);
    return QPDO_NULL;
}

/* ---- qpdo_warm_start (reference src/qpdo.c:217-299) --------------------------------------- */
void qpdo_warm_start(QPDOWorkspace *work, c_float *x_warm_start, c_float *y_warm_start) {
    work->sigma = work->settings->sigma_init;
    if (work->info->status_val != QPDO_UNSOLVED) work->info->setup_time = 0;
    tic(work->timer);
    c_float obj = 0.0;
    if (work->chol->small && work->chol->auto_ws && !work->settings->verbose) {
        /* qpdo_solve's automatic warm start (qpdo.c:312-314, x = y = 0) of a fused-path workspace: nothing can come between it and the
         * solve, so it is part of the solve's one launch -- nothing to do on the device here */
        work->chol->ws_state = 0;
        work->info->objective = 0.0;
        work->sqrt_mu_min = 1 / sqrt(work->settings->mu_min);
        work->initialized = 1;
        work->info->setup_time += toc(work->timer);
        return;
    }
    if (work->chol->small) {
        /* an explicit warm start of a fused-path workspace: x, x_bar, Qx, Ax, y, y_bar, A'y, mu in the ORACLE's operation order (one
         * workgroup, qpdo_small.hip mode 1) into the workspace's device vectors; qpdo_update_q / _bounds may then act on them as in the
         * reference before the solve reads them back (mode 2) */
        QdevSmallView v;
        if (qdev_small_view(work->chol->dev, &v) ||
            qdev_small_resident_warm_start(work->chol->small, &v, work->settings, x_warm_start, y_warm_start, work->data->c, &obj)) {
            QPDO_EPRINT("fused small-problem warm start: %s / %s", qdev_small_last_error(), qdev_last_error()); update_status(work->info, QPDO_ERROR); return;
        }
        work->chol->ws_state = 1;
        work->info->objective = x_warm_start ? obj : 0.0;
        work->sqrt_mu_min = 1 / sqrt(work->settings->mu_min);
        work->initialized = 1;
        work->info->setup_time += toc(work->timer);
        return;
    }
    int rc = qdev_warm_start(work->chol->dev, x_warm_start, y_warm_start, (int)work->settings->proximal, work->sigma,
                             work->settings->mu_min, work->data->c, &obj);
    if (rc) { QPDO_EPRINT("device backend: %s", qdev_last_error()); update_status(work->info, QPDO_ERROR); return; }
    work->chol->ws_state = 1;
    work->info->objective = x_warm_start ? obj : 0.0;
    work->sqrt_mu_min = 1 / sqrt(work->settings->mu_min);      /* iteration.c:121 */
    work->initialized = 1;
    work->info->setup_time += toc(work->timer);
}

/* reference src/util.c:101-117 */
static void print_header(void) {
    QPDO_PRINT("============================================================================\n");
    QPDO_PRINT("===                    QPDO  v0.1  (MI355X-native backend)               ===\n");
    QPDO_PRINT("============================================================================\n");
    QPDO_PRINT("  iter |  objective     r.prim     r.dual |  r.p. in    r.d. in   stepsize | \n");
    QPDO_PRINT("============================================================================\n");
}
static void print_iteration(c_int iter, QPDOWorkspace *work) {
    QPDO_PRINT("%6ld | %+-.3e   %.2e   %.2e | %.2e   %.2e   %.2e | \n", (long)iter, work->info->objective, work->info->res_prim_norm,
               work->info->res_dual_norm, work->info->res_prim_in_norm, work->info->res_dual_in_norm, work->tau);
}
static void print_final_message(QPDOWorkspace *work) {   /* src/util.c:122-173 */
    const char *msg;
    switch (work->info->status_val) {
        case QPDO_SOLVED: msg = "QPDO finished successfully."; break;
        case QPDO_PRIMAL_INFEASIBLE: msg = "QPDO detected a primal infeasible problem."; break;
        case QPDO_DUAL_INFEASIBLE: msg = "QPDO detected a dual infeasible problem."; break;
        case QPDO_PRIMAL_DUAL_INFEASIBLE: msg = "QPDO detected a primal-dual infeasible problem."; break;
        case QPDO_MAX_ITER_REACHED: msg = "QPDO hit the maximum number of iterations."; break;
        case QPDO_MAX_TIME_REACHED: msg = "QPDO exceeded the specified time limit."; break;
        default: msg = "QPDO ended with an unrecognised status."; break;
    }
    QPDO_PRINT("============================================================================\n");
    QPDO_PRINT("| %-72s |\n", msg);
    QPDO_PRINT("| primal residual: %5.4e,                primal tolerance: %5.4e |\n", work->info->res_prim_norm, work->settings->eps_abs);
    QPDO_PRINT("| dual residual  : %5.4e,                dual tolerance  : %5.4e |\n", work->info->res_dual_norm, work->settings->eps_abs);
    QPDO_PRINT("| objective value: %+-5.4e                                             |\n", work->info->objective);
    if (work->info->run_time > 1.0) QPDO_PRINT("| runtime:         %4.2f seconds\n", work->info->run_time);
    else QPDO_PRINT("| runtime:         %4.2f milliseconds\n", work->info->run_time * 1000);
    QPDO_PRINT("============================================================================\n\n");
}

/* ---- qpdo_solve of a small workspace: ONE launch of the fused kernel (qpdo_small.hip) on the workspace's device arrays --------------
 * Same loop (src/qpdo.c:343-449) in the oracle's operation order, so status, counts and iterates carry the oracle's bits.  Called by
 * qpdo_solve after its prologue (header, automatic warm start, eps_in / sigma / reset_newton, status quirk Q1, tic, trace reset). */
static void fused_solve(QPDOWorkspace *work) {
    struct QPDOBackend *be = work->chol;
    QPDOSettings st = *work->settings;
    QdevSmallView v;
    QdevSmallResult res;
    QPDOInfo kinfo;
    memset(&res, 0, sizeof(res)); memset(&kinfo, 0, sizeof(kinfo));
    res.info = &kinfo;
    /* max_time (qpdo.c:441-447) counts from the start of setup: run_time = setup_time + time in the solve; the kernel's clock starts now */
    if (st.max_time < 1e19) st.max_time -= work->info->setup_time;
    const c_int status_before = work->info->status_val;
    if (qdev_small_view(be->dev, &v) ||
        qdev_small_resident_solve(be->small, &v, &st, be->ws_state, work->data->c, &res)) {
        QPDO_EPRINT("fused small-problem solve: %s / %s", qdev_small_last_error(), qdev_last_error());
        update_status(work->info, QPDO_ERROR);
        work->initialized = 0;
        work->info->solve_time = toc(work->timer);
        work->info->run_time = work->info->setup_time + work->info->solve_time;
        return;
    }
    const size_t n = work->data->n, m = work->data->m;
    work->info->iterations = kinfo.iterations; work->info->oterations = kinfo.oterations;
    work->info->res_prim_norm = kinfo.res_prim_norm; work->info->res_dual_norm = kinfo.res_dual_norm;
    work->info->res_prim_in_norm = kinfo.res_prim_in_norm; work->info->res_dual_in_norm = kinfo.res_dual_in_norm;
    work->info->objective = kinfo.objective;
    /* quirk Q1 (qpdo.c:451-453): only an UNSOLVED status is overwritten when the loop runs out of passes */
    if (kinfo.status_val == QPDO_MAX_ITER_REACHED) { if (status_before == QPDO_UNSOLVED || be->fix_status_reset) update_status(work->info, QPDO_MAX_ITER_REACHED); }
    else update_status(work->info, kinfo.status_val);
    work->sigma = res.sigma_end; work->tau = res.tau_end;
    memcpy(work->solution->x, res.sol_x, n * sizeof(c_float)); memcpy(work->x, res.x, n * sizeof(c_float)); memcpy(work->dx, res.dx, n * sizeof(c_float));
    if (m) { memcpy(work->solution->y, res.sol_y, m * sizeof(c_float)); memcpy(work->y, res.y, m * sizeof(c_float)); memcpy(work->dy, res.dy, m * sizeof(c_float)); }
    be->newton_passes = res.newton_passes; be->last_fused = 1; be->fused_solves++; be->fused_factor_count = res.factor_count; be->fused_kernel_s = res.kernel_seconds;
    /* the per-pass trace (pinned records written by the kernel) into the workspace's trace array */
    if (res.ntrace > be->captrace) {
        QPDOAmdTraceRec *t = realloc(be->trace, (size_t)res.ntrace * sizeof(*t));
        if (t) { be->trace = t; be->captrace = res.ntrace; }
    }
    be->ntrace = res.ntrace <= be->captrace ? res.ntrace : be->captrace;
    if (be->ntrace > 0) memcpy(be->trace, res.trace, (size_t)be->ntrace * sizeof(QPDOAmdTraceRec));
    work->initialized = 0;
    work->info->solve_time = toc(work->timer);
    work->info->run_time = work->info->setup_time + work->info->solve_time;
}

#define DEVCALL(call) do { if ((call)) { QPDO_EPRINT("device backend: %s", qdev_last_error()); update_status(work->info, QPDO_ERROR); goto done; } } while (0)

/* ---- qpdo_solve (reference src/qpdo.c:304-476) ---------------------------------------------- */
void qpdo_solve(QPDOWorkspace *work) {
    struct QPDOBackend *be = work->chol;
    QpdoDev *dev = be->dev;
    const QPDOSettings *s;
    if (work->settings->verbose) print_header();
    if (!work->initialized) { be->auto_ws = 1; qpdo_warm_start(work, NULL, NULL); be->auto_ws = 0; }
    s = work->settings;
    const int prox = (int)s->proximal;
    work->eps_in = s->eps_abs_in;
    work->sigma = s->sigma_init;
    be->reset_newton = 1;
    if (be->fix_status_reset) update_status(work->info, QPDO_UNSOLVED);
    tic(work->timer);
    be->ntrace = 0; be->newton_passes = 0; be->last_fused = 0;
    qdev_reset_stats(dev);
    c_int iter = 0, oter = 0, iter_old = 0;
    long tr_pending = -1;           /* index of the trace record whose tau is still in flight (deferred step read-back) */
    c_int last_nchange = -1;        /* n_enter + n_leave of the last Newton pass (-1: none yet / an outer update came after it) */
    if (!work->initialized) goto done;      /* warm start failed on the device */
    if (be->small && !s->verbose) { fused_solve(work); return; }
    if (!be->ws_state) {                    /* (cannot happen: the automatic cold start is skipped only when this solve takes the fused path) */
        c_float obj0 = 0.0;
        DEVCALL(qdev_warm_start(dev, NULL, NULL, prox, work->sigma, s->mu_min, work->data->c, &obj0));
        be->ws_state = 1;
    }
    DEVCALL(qdev_begin_solve(dev));
    DEVCALL(qdev_set_eps_abs(dev, s->eps_abs));
    /* One host synchronisation per loop pass instead of two (dense solver; QPDO_DEFER_STEP=0 switches it off): the Newton step is
     * launched without waiting for it; its step length -- needed only by the trace, the iteration line and is_dual_infeasible -- comes
     * back with the next pass's residual norms.  Printing wants tau on the pass's own line, so verbose solves keep two. */
    DEVCALL(qdev_set_deferred_step(dev, !s->verbose && env_int("QPDO_DEFER_STEP", 1)));

    for (iter = 0; iter < s->max_iter; iter++) {
        QdevResid r;
        /* launch-ahead (qpdo_dev.h qdev_residuals_ahead): on the mid-size dense route this pass's Newton step is enqueued behind the
         * residual launch, which forms the decisions below on the device from what the host knows now; the host still makes them itself
         * from the norms it reads back, and any disagreement ends the solve with QPDO_ERROR */
        QdevAhead ah;
        ah.allow_outer = iter > iter_old + 1; ah.force_outer = (iter == iter_old + s->inner_max_iter);
        ah.reset_newton = be->reset_newton || (s->reset_newton_iter > 0 && (iter % s->reset_newton_iter == 0));
        ah.max_rank = QPDO_MAX_RANK_UPDATE; ah.eps_abs = s->eps_abs; ah.eps_in = work->eps_in; ah.infty = QPDO_INFTY;
        /* (not when the last Newton pass saw no row enter or leave and nothing forces a new factorization: the active set tends to
         * stand still then, the host-first path keeps its factor on such a pass and the launched-ahead step would refactor) */
        const int ahead_ok = !s->verbose && (ah.reset_newton || last_nchange != 0);
        DEVCALL(qdev_residuals_ahead(dev, prox, work->sigma, ahead_ok ? &ah : NULL, &r));
        if (r.prev_step_done) {
            work->tau = r.prev_tau;
            if (tr_pending >= 0 && tr_pending < be->ntrace) be->trace[tr_pending].tau = r.prev_tau;
            tr_pending = -1;
        }
        work->info->res_prim_norm = r.res_prim; work->info->res_dual_norm = r.res_dual;
        work->info->res_prim_in_norm = r.res_prim_in; work->info->res_dual_in_norm = r.res_dual_in;
        QPDOAmdTraceRec *tr = trace_push(be);
        if (tr) {
            tr->kind = 2; tr->res_prim = r.res_prim; tr->res_dual = r.res_dual; tr->res_prim_in = r.res_prim_in;
            tr->res_dual_in = r.res_dual_in; tr->sigma = work->sigma; tr->eps_in = work->eps_in;
        }
        if (s->verbose && (s->print_interval > 0) && (iter % s->print_interval == 0)) {
            DEVCALL(qdev_objective(dev, prox, work->sigma, work->data->c, &work->info->objective));
            print_iteration(iter, work);
        }
        /* check_outer_optimality (termination.c:11-23) */
        /* (pass_decision.h: termination.c:11-30, qpdo.c:361-363, newton.c:21-33 -- the function the residual launch itself evaluates on the
         * launch-ahead route, on the same numbers) */
        const QpdoPassDecision pd = qpdo_pass_decision(r.res_prim, r.res_dual, r.res_prim_in, r.res_dual_in, s->eps_abs, work->eps_in, QPDO_INFTY,
                                                       ah.allow_outer, ah.force_outer, ah.reset_newton, (int)r.n_active, (int)(r.n_enter + r.n_leave),
                                                       QPDO_MAX_RANK_UPDATE);
        const int ends_nc = pd.ends_nc, ends_ok = pd.ends_ok, outer_pass = pd.outer;
        if (r.ahead_went && (ends_nc || ends_ok || outer_pass)) {
            QPDO_EPRINT("launch-ahead: the device started a Newton step in pass %ld, which the host ends or makes an outer update", (long)iter);
            update_status(work->info, QPDO_ERROR); break;
        }
        if (ends_nc) { update_status(work->info, QPDO_NON_CVX); break; }
        if (ends_ok) { update_status(work->info, QPDO_SOLVED); break; }

        if (outer_pass) {
            if (tr) tr->kind = 1;
            if (iter < iter_old + s->inner_max_iter) {
                if (s->eps_prim_inf > 0) {
                    int inf = 0;
                    DEVCALL(qdev_primal_infeasibility(dev, s->eps_prim_inf, &inf));
                    if (inf) { update_status(work->info, QPDO_PRIMAL_INFEASIBLE); break; }
                }
            }
            /* is_dual_infeasible and update_mu (iteration.c:127-168) share one read-back: update_mu is enqueued behind the test and does
             * nothing if the test says "infeasible" (update_mu does not look at x_bar / y_bar, so running it before the estimates are
             * shifted changes nothing) */
            const int do_dinf = (iter < iter_old + s->inner_max_iter) && (s->eps_dual_inf > 0);
            const int do_mu = (oter > 0) && (r.res_prim > s->eps_abs);
            int dinf = 0, nch = 0;
            DEVCALL(qdev_dual_infeasibility_and_mu(dev, do_dinf, prox, work->sigma, work->tau, s->eps_dual_inf, do_mu, s->eps_abs, s->theta,
                                                   s->delta, s->mu_min, work->sqrt_mu_min, &dinf, &nch));
            if (dinf) { update_status(work->info, QPDO_DUAL_INFEASIBLE); break; }
            DEVCALL(qdev_shift_estimates(dev));
            if (do_mu) {
                work->n_mu_changed = nch;
                if ((prox && work->sigma > s->sigma_min) || (nch > 0.25 * QPDO_MAX_RANK_UPDATE)) be->reset_newton = 1;
                else if (nch == 0) { /* nothing */ }
                else DEVCALL(qdev_mu_changed_update(dev));
            }
            if (prox && (oter > 0) && (r.res_dual > s->eps_abs)) {
                /* update_sigma (iteration.c:173-180) */
                if (work->sigma > s->sigma_min) {
                    c_float sigma_old = work->sigma;
                    work->sigma = c_max(work->sigma * s->sigma_upd, s->sigma_min);
                    be->reset_newton = 1;
                    DEVCALL(qdev_update_sigma(dev, work->sigma, sigma_old));
                }
            }
            if (iter < iter_old + s->inner_max_iter) {
                work->eps_in = c_max(s->rho * work->eps_in, 0.1 * s->eps_abs);
                if (s->verbose && (s->print_interval > 0) && (iter % s->print_interval == 0))
                    QPDO_PRINT("%6ld |-------------------------------------------------------------------|\n", (long)iter);
            } else if (s->verbose && (s->print_interval > 0) && (iter % s->print_interval == 0)) {
                QPDO_PRINT("%6ld |--  --  --  --  --  --  --  --  --  --  --  --  --  --  --  --  -- |\n", (long)iter);
            }
            DEVCALL(qdev_save_res_prim(dev));
            oter++;
            iter_old = iter;
            last_nchange = -1;
        } else {
            if (tr) tr->kind = 0;
            /* the reference computes iter % reset_newton_iter, a division by zero for the (valid) setting 0;
             * here 0 means "no periodic refactorization" */
            if (s->reset_newton_iter > 0 && (iter % s->reset_newton_iter == 0)) be->reset_newton = 1;
            /* factorization decision (newton.c:21-33) */
            const int branch = pd.branch;          /* (ah.reset_newton is be->reset_newton after the line above) */
            if (branch == 0) be->reset_newton = 0;
            int lin = 0;
            DEVCALL(qdev_newton_step(dev, branch, r.n_enter + r.n_leave, prox, work->sigma, &work->tau, &lin));
            be->newton_passes++;
            last_nchange = r.n_enter + r.n_leave;
            if (tr && work->tau != work->tau) tr_pending = be->ntrace - 1;        /* NaN: the step's read-back is deferred */
            if (tr) { tr->tau = work->tau; tr->n_active = r.n_active; tr->n_enter = r.n_enter; tr->n_leave = r.n_leave; tr->factor_branch = branch; tr->lin_iters = lin; }
        }
        work->info->run_time = work->info->setup_time + toc(work->timer);
        if (work->info->run_time > s->max_time) { update_status(work->info, QPDO_MAX_TIME_REACHED); break; }
    }
    if (work->info->status_val == QPDO_UNSOLVED) update_status(work->info, QPDO_MAX_ITER_REACHED);

done:
    {   /* a Newton step still in flight (loop left by max_iter / max_time right after it): complete it before the solution is stored */
        int had = 0; c_float tau_last = 0.0;
        if (qdev_finish_step(dev, &had, &tau_last)) { QPDO_EPRINT("device backend: %s", qdev_last_error()); update_status(work->info, QPDO_ERROR); }
        else if (had) { work->tau = tau_last; if (tr_pending >= 0 && tr_pending < be->ntrace) be->trace[tr_pending].tau = tau_last; }
    }
    work->info->iterations = iter;
    work->info->oterations = oter;
    /* store_solution (termination.c:82-92) + host mirrors */
    if (qdev_store_solution_obj(dev, prox, work->sigma, work->data->c, work->solution->x, work->solution->y, work->x, work->y, work->dx, work->dy,
                                &work->info->objective)) {
        QPDO_EPRINT("device backend: %s", qdev_last_error());
        update_status(work->info, QPDO_ERROR);
    }
    work->initialized = 0;
    work->info->solve_time = toc(work->timer);
    work->info->run_time = work->info->setup_time + work->info->solve_time;
    if (work->settings->verbose) {
        if (work->settings->print_interval > 0 && (iter % work->settings->print_interval != 0)) print_iteration(iter, work);
        print_final_message(work);
    }
}

/* ---- qpdo_update_settings (reference src/qpdo.c:481-517) --------------------------------------- */
void qpdo_update_settings(QPDOWorkspace *work, const QPDOSettings *settings) {
    if (!validate_settings(settings)) {
        QPDO_EPRINT("Settings validation returned failure");
        update_status(work->info, QPDO_ERROR);
        return;
    }
    if (work->settings->scaling > settings->scaling) {
        QPDO_EPRINT("Decreasing the number of scaling iterations is not allowed");
        update_status(work->info, QPDO_ERROR);
        return;
    } else if (work->settings->scaling < settings->scaling) {
        if (!work->scaling) {   /* the reference dereferences NULL here (qpdo.c:496-499) */
            QPDO_EPRINT("scaling cannot be enabled after setup");
            update_status(work->info, QPDO_ERROR);
            return;
        }
        size_t n = work->data->n, m = work->data->m;
        c_float *Dsave = vec_dup(work->scaling->D, n), *Esave = vec_dup(work->scaling->E, m);
        c_float c_temp = work->scaling->c, c_new = 1.0;
        if (!Dsave || !Esave || qdev_scale_data(work->chol->dev, (int)(settings->scaling - work->settings->scaling), 1,
                                                 work->scaling->D, work->scaling->E, &c_new)) {
            free(Dsave); free(Esave);
            update_status(work->info, QPDO_ERROR);
            return;
        }
        /* bounds: l,u <- E_new .* l,u on the host mirrors (scaling.c:86-87) */
        for (size_t i = 0; i < m; i++) { work->data->l[i] = work->scaling->E[i] * work->data->l[i]; work->data->u[i] = work->scaling->E[i] * work->data->u[i]; }
        for (size_t i = 0; i < n; i++) work->scaling->D[i] = work->scaling->D[i] * Dsave[i];
        for (size_t i = 0; i < m; i++) work->scaling->E[i] = work->scaling->E[i] * Esave[i];
        work->scaling->c = c_new * c_temp;
        free(Dsave); free(Esave);
        if (install_scaling(work) || qdev_download_q(work->chol->dev, work->data->q) ||
            qdev_upload_bounds(work->chol->dev, work->data->l, work->data->u)) {
            update_status(work->info, QPDO_ERROR);
            return;
        }
    }
    free(work->settings);
    work->settings = copy_settings(settings);
    work->sqrt_delta = sqrt(work->settings->delta);
}

/* ---- qpdo_update_bounds (reference src/qpdo.c:522-544) ------------------------------------------ */
void qpdo_update_bounds(QPDOWorkspace *work, const c_float *l, const c_float *u) {
    size_t m = work->data->m;
    if (l != NULL && u != NULL) {
        for (size_t j = 0; j < m; j++) {
            if (l[j] > u[j]) {
                QPDO_EPRINT("Lower bound at index %d is greater than upper bound: %.4e > %.4e", (int)j, l[j], u[j]);
                update_status(work->info, QPDO_ERROR);
                return;
            }
        }
    }
    if (l != NULL) memcpy(work->data->l, l, m * sizeof(c_float));
    if (u != NULL) memcpy(work->data->u, u, m * sizeof(c_float));
    if (work->settings->scaling) {
        if (l != NULL) for (size_t i = 0; i < m; i++) work->data->l[i] = work->scaling->E[i] * work->data->l[i];
        if (u != NULL) for (size_t i = 0; i < m; i++) work->data->u[i] = work->scaling->E[i] * work->data->u[i];
    }
    if (qdev_upload_bounds(work->chol->dev, l ? work->data->l : NULL, u ? work->data->u : NULL)) update_status(work->info, QPDO_ERROR);
}

/* ---- qpdo_update_q (reference src/qpdo.c:549-586) ------------------------------------------------- */
void qpdo_update_q(QPDOWorkspace *work, const c_float *q) {
    size_t n = work->data->n;
    QpdoDev *dev = work->chol->dev;
    memcpy(work->data->q, q, n * sizeof(c_float));
    if (work->settings->scaling) {
        /* device side (round 3): Qx and x stay in HBM; the host receives the new cost scaling c and norm_q only and keeps its copy
         * of the scaled q (work->data->q) in step by reading it back -- n doubles, instead of 2n down + 5n up before */
        c_float c_new = 1.0, cinv_new = 1.0, nq = 0.0;
        const c_float sigma_new = work->settings->proximal ? work->settings->sigma_init : work->sigma;
        if (qdev_update_q_scaled(dev, q, (int)work->settings->proximal, work->sigma, sigma_new, work->scaling->c, &c_new, &cinv_new, &nq) ||
            qdev_download_q(dev, work->data->q)) { update_status(work->info, QPDO_ERROR); return; }
        work->scaling->c = c_new; work->scaling->cinv = cinv_new;
        if (work->settings->proximal) work->sigma = work->settings->sigma_init;
        work->norm_q = nq;
    } else {
        if (qdev_upload_q(dev, work->data->q)) update_status(work->info, QPDO_ERROR);
        work->norm_q = vec_norm_inf(work->data->q, n);
    }
}

/* ---- qpdo_cleanup (reference src/qpdo.c:591-689) ---------------------------------------------------- */
void qpdo_cleanup(QPDOWorkspace *work) {
    if (!work) return;
    if (work->data) { free(work->data->q); free(work->data->l); free(work->data->u); free(work->data); }
    if (work->scaling) { free(work->scaling->D); free(work->scaling->Dinv); free(work->scaling->E); free(work->scaling->Einv); free(work->scaling); }
    free(work->x); free(work->y); free(work->dx); free(work->dy);
    free(work->settings);
    if (work->chol) { qdev_small_resident_destroy(work->chol->small); qdev_destroy(work->chol->dev); free(work->chol->trace); free(work->chol); }
    if (work->solution) { free(work->solution->x); free(work->solution->y); free(work->solution); }
    free(work->timer);
    free(work->info);
    free(work);
}

/* ---- extensions (include/qpdo_amd_ext.h) --------------------------------------------------------------- */
int qpdo_amd_device_count(void) { return qdev_device_count(); }
int qpdo_amd_pass_decision(double res_prim, double res_dual, double res_prim_in, double res_dual_in, double eps_abs, double eps_in, int allow_outer,
                           int force_outer, int reset_newton, int n_active, int n_change, int *ends_nc, int *ends_ok, int *outer, int *branch) {
    const QpdoPassDecision pd = qpdo_pass_decision(res_prim, res_dual, res_prim_in, res_dual_in, eps_abs, eps_in, QPDO_INFTY, allow_outer, force_outer,
                                                   reset_newton, n_active, n_change, QPDO_MAX_RANK_UPDATE);
    *ends_nc = pd.ends_nc; *ends_ok = pd.ends_ok; *outer = pd.outer; *branch = pd.branch;
    return 0;
}
const char *qpdo_amd_last_error(void) { return qdev_last_error(); }
int qpdo_amd_sync(QPDOWorkspace *work) { return qdev_sync(work->chol->dev); }
int qpdo_amd_get_trace(const QPDOWorkspace *work, const QPDOAmdTraceRec **recs, long *count) {
    *recs = work->chol->trace; *count = work->chol->ntrace; return 0;
}
int qpdo_amd_bench_spmv(QPDOWorkspace *work, int which, int reps, double *avg_seconds, double *alg_bytes) {
    return qdev_bench_spmv(work->chol->dev, which, reps, avg_seconds, alg_bytes);
}
int qpdo_amd_bench_dense_factor(QPDOWorkspace *work, int reps, double *avg_seconds, double *check) {
    return qdev_bench_dense_factor(work->chol->dev, reps, avg_seconds, check);
}
int qpdo_amd_spmv(QPDOWorkspace *work, int which, const double *v, double *y) { return qdev_spmv(work->chol->dev, which, v, y); }
int qpdo_amd_linesearch(QPDOWorkspace *work, double eta, double beta, const double *delta, const double *alpha, double *tau) {
    return qdev_linesearch(work->chol->dev, eta, beta, delta, alpha, tau);
}
int qpdo_amd_download(QPDOWorkspace *work, int which, double *dst) { return qdev_download_vec(work->chol->dev, which, dst); }
int qpdo_amd_get_stats(const QPDOWorkspace *work, QPDOAmdStats *out) {
    QdevStats st;
    double avg = 0.0; long ns = 0;
    qdev_get_stats(work->chol->dev, &st);
    qdev_get_spmv_sample(work->chol->dev, &avg, &ns);
    out->newton_passes = work->chol->newton_passes;
    out->lin_iters = (long)st.lin_iters;
    out->spmv_calls = (long)st.spmv_calls;
    out->spmv_alg_bytes = (double)st.spmv_bytes;
    out->factor_count = (long)st.factor_count;
    out->linsolve = st.linsolve;
    out->spmv_Q_avg_s = avg;
    out->spmv_Q_samples = ns;
    { double ts = 0, bs = 0; long n2 = 0, sp = 0; qdev_get_ac_sample(work->chol->dev, &ts, &bs, &n2, &sp);
      out->spmv_Ac_time_s = ts; out->spmv_Ac_bytes = bs; out->spmv_Ac_samples = n2; out->schur_passes = sp; }
    out->lowrank_solves = (long)st.lowrank_solves;
    out->lowrank_cols = (long)st.lowrank_cols;
    out->lowrank_sweeps = (long)st.lowrank_sweeps;
    out->lowrank_rejects = (long)st.lowrank_rejects;
    out->pcg_soft_accepts = (long)st.pcg_soft_accepts;
    out->chain_fallbacks = (long)st.chain_fallbacks;
    out->collectives = (long)st.collectives; out->inner_solves = (long)st.inner_solves;
    out->inner_steps = (long)st.inner_steps; out->inner_collectives = (long)st.inner_collectives;
    out->pcg_max_relres = st.pcg_max_relres;
    out->pcg_dense_fallbacks = (long)st.pcg_dense_fallbacks;
    out->pcg_rescues = (long)st.pcg_rescues; out->pcg_rescue_kinds = (long)st.pcg_rescue_kinds;
    out->hybrid_pcg_passes = (long)st.hybrid_pcg_passes; out->band_fallbacks = (long)st.band_fallbacks;
    out->onelaunch_factors = (long)st.onelaunch_factors;
    out->ahead_steps = (long)st.ahead_steps; out->ahead_skips = (long)st.ahead_skips;
    out->fused_solves = work->chol->fused_solves;
    out->fused_kernel_s = work->chol->last_fused ? work->chol->fused_kernel_s : 0.0;
    if (work->chol->last_fused) { out->factor_count = work->chol->fused_factor_count; out->linsolve = 2; }
    return 0;
}

/* ---- batch of independent QPs ------------------------------------------------------------------------------ */
typedef struct { long count; QPDOAmdBatchItem *items; const QPDOSettings *settings; long next; long failed; pthread_mutex_t mu; } BatchCtx;
static void *batch_worker(void *arg) {
    BatchCtx *b = (BatchCtx *)arg;
    for (;;) {
        pthread_mutex_lock(&b->mu);
        long i = b->next++;
        pthread_mutex_unlock(&b->mu);
        if (i >= b->count) break;
        QPDOAmdBatchItem *it = &b->items[i];
        const size_t n = it->data->n, m = it->data->m;
        QPDOWorkspace *w = qpdo_setup(it->data, b->settings);
        if (!w) {
            memset(&it->info, 0, sizeof(it->info));
            update_status(&it->info, QPDO_ERROR);
            pthread_mutex_lock(&b->mu); b->failed++; pthread_mutex_unlock(&b->mu);
            continue;
        }
        if (it->x0 || it->y0) qpdo_warm_start(w, (c_float *)it->x0, (c_float *)it->y0);
        qpdo_solve(w);
        it->info = *w->info;
        const int infeasible = (w->info->status_val == QPDO_PRIMAL_INFEASIBLE) || (w->info->status_val == QPDO_DUAL_INFEASIBLE);
        if (it->x) for (size_t k = 0; k < n; k++) it->x[k] = infeasible ? NAN : w->solution->x[k];
        if (it->y) for (size_t k = 0; k < m; k++) it->y[k] = infeasible ? NAN : w->solution->y[k];
        qpdo_cleanup(w);
    }
    return NULL;
}
double qpdo_amd_batch_kernel_seconds(void) { return qdev_small_last_kernel_seconds(); }
long qpdo_amd_solve_batch(long count, QPDOAmdBatchItem *items, const QPDOSettings *settings, int nthreads) {
    if (count <= 0) return 0;
    /* small problems: the fused one-workgroup-per-QP kernel solves the whole batch in one launch
     * (QPDO_BATCH=threads forces the generic multi-kernel path on a host thread pool) */
    const char *mode = getenv("QPDO_BATCH");
    if (!(mode && !strcmp(mode, "threads")) && validate_settings(settings) && qdev_small_eligible(count, items)) {
        long bad = 0;
        for (long i = 0; i < count; i++) if (!validate_data(items[i].data)) bad++;
        if (!bad) {
            int ndev = qdev_device_count();
            if (ndev > 0) {
                int device = env_int("QPDO_DEVICE", env_int("LOCAL_RANK", 0)) % ndev;
                const double tb0 = wall_now();
                const int rcb = qdev_small_batch(device, count, items, settings);
                if (env_int("QPDO_SMALL_PROF", 0) == 2) fprintf(stderr, "[qpdo_small host] qdev_small_batch total   %.3f s\n", wall_now() - tb0);
                if (rcb == 0) return 0;
                QPDO_EPRINT("fused batch kernel failed (%s); using the generic path", qdev_small_last_error());
            }
        }
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nthreads > count) nthreads = (int)count;
    BatchCtx b = {count, items, settings, 0, 0, PTHREAD_MUTEX_INITIALIZER};
    pthread_t th[64];
    int started = 0;
    for (int t = 0; t < nthreads; t++) if (pthread_create(&th[started], NULL, batch_worker, &b) == 0) started++;
    if (started == 0) batch_worker(&b);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    return b.failed;
}

/* streamed batches: see include/qpdo_amd_ext.h */
QPDOAmdBatchStream *qpdo_amd_batch_stream_create(int depth) {
    int ndev = qdev_device_count();
    if (ndev <= 0) { QPDO_EPRINT("no HIP device available (this library has no CPU path)"); return NULL; }
    int device = env_int("QPDO_DEVICE", env_int("LOCAL_RANK", 0)) % ndev;
    warn_hw_queues_once(depth);
    return (QPDOAmdBatchStream *)qdev_small_stream_create(device, depth);
}
long qpdo_amd_batch_stream_submit(QPDOAmdBatchStream *stream, long count, QPDOAmdBatchItem *items, const QPDOSettings *settings) {
    if (!stream || count <= 0 || !items || !settings) return -1;
    if (!validate_settings(settings)) return -1;
    if (!qdev_small_eligible(count, items)) { QPDO_EPRINT("batch stream: an item does not fit the fused kernel (n, m <= 1024); use qpdo_amd_solve_batch"); return -1; }
    for (long i = 0; i < count; i++) if (!validate_data(items[i].data)) return -1;
    const long t = qdev_small_stream_submit(stream, count, items, settings);
    if (t < 0) QPDO_EPRINT("batch stream: %s", qdev_small_last_error());
    return t;
}
int qpdo_amd_batch_stream_wait(QPDOAmdBatchStream *stream, long ticket, double *kernel_seconds) {
    if (!stream) return -1;
    const int rc = qdev_small_stream_wait(stream, ticket, kernel_seconds);
    if (rc) QPDO_EPRINT("batch stream: %s", qdev_small_last_error());
    return rc;
}
void qpdo_amd_batch_stream_destroy(QPDOAmdBatchStream *stream) { qdev_small_stream_destroy(stream); }
