/*
 * qpdo_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A sequential, plain-C CPU restatement of the reference solver aldma/qpdo for
 * the hot path named in BASELINE.json (Newton pass: residuals, semismooth
 * Newton direction, exact breakpoint linesearch, iterate update; plus the
 * driver, parameter updates, infeasibility certificates and Ruiz scaling that
 * decide iteration counts and statuses).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's library.  The product
 * (libqpdo_amd.so) never links, loads or calls it.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference tree).  Vector arithmetic keeps the reference's operation order
 * (including the 4-way grouping of vec_prod) and is compiled with
 * -ffp-contract=off, as the reference's x86-64 -O3 build has no FMA.
 *
 * Third-party arithmetic: the reference delegates SpMV, normal-equation
 * assembly, LDL' factor / rank update / solve to SuiteSparse CHOLMOD 3.0.14
 * (SuiteSparse 5.8.1), which is NOT in the reference tree and not installed
 * here.  This file restates the published algorithms instead:
 *   - sdmult: column-ordered CSC mat-vec, symmetric case using one triangle;
 *   - factor: the reference's factor always equals  Q + sigma_f I + A' diag(d) A
 *     for a weight vector d evolved by full-factor / enter / leave / mu-changed
 *     rules (cholmod_interface.c:35-93).  The oracle tracks (sigma_f, d)
 *     explicitly and factors that matrix with a dense natural-order LDL'
 *     (no pivoting), i.e. rank updates are replaced by their exact-arithmetic
 *     equivalent.  Summation order therefore differs from CHOLMOD's.
 * PARITY PIN: the reference ships only three known-answer checks for this
 * path (examples/infeasibility_tests.m:30,48,75: statuses 1, -3, -4); the
 * oracle reproduces them (tests/test_oracle_kat.py).  Beyond those three,
 * parity against genuine CHOLMOD arithmetic is unpinned (no fixture exists and
 * the reference cannot be built here).
 *
 * A second linear-solver mode (Jacobi-PCG on the same operator) exists for the
 * CPU baseline at sizes where the dense factor does not fit, and to study the
 * sensitivity of iteration counts to inexact solves.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double  f64;
typedef int64_t i64;

#define O_INFTY 1e20            /* constants.h:37-39 */
#define O_MAX_RANK_UPDATE 100   /* constants.h:69 */
#define O_MIN_SCALING 1e-9      /* constants.h:62 */

#define ST_SOLVED 1
#define ST_NON_CVX (-1)
#define ST_PRIMAL_INFEASIBLE (-3)
#define ST_DUAL_INFEASIBLE (-4)
#define ST_MAX_ITER (-5)
#define ST_MAX_TIME (-6)
#define ST_UNSOLVED (-10)
#define ST_ERROR (-99)

/* same member order as QPDOSettings (types.h:96-116), 64-bit ints */
typedef struct {
    f64 max_time;
    i64 max_iter;
    i64 inner_max_iter;
    f64 eps_abs;
    f64 eps_abs_in;
    f64 eps_prim_inf;
    f64 eps_dual_inf;
    f64 rho;
    f64 theta;
    f64 delta;
    f64 mu_min;
    i64 proximal;
    f64 sigma_init;
    f64 sigma_upd;
    f64 sigma_min;
    i64 scaling;
    i64 verbose;
    i64 print_interval;
    i64 reset_newton_iter;
} OracleSettings;

typedef struct {
    i64 nrow, ncol;
    i64 *p, *i;
    f64 *x;
    int stype;          /* 0 general, -1 lower stored, +1 upper stored */
} Csc;

/* one record per loop pass of qpdo_solve (qpdo.c:343-449) */
typedef struct {
    i64 kind;           /* 0 Newton step, 1 outer update, 2 terminated here */
    i64 n_active, n_enter, n_leave;
    i64 factor_branch;  /* 0 full, 1 rank update, 2 Q only, -1 n/a */
    i64 lin_iters;      /* PCG iterations (0 for direct) */
    f64 tau;
    f64 res_prim, res_dual, res_prim_in, res_dual_in;
    f64 sigma, eps_in;
    f64 t_end;          /* seconds from the start of oracle_solve to the end of this pass (cpu_baseline sampling) */
} TraceRec;

typedef struct {
    i64 n, m;
    Csc Q, A;           /* scaled copies */
    /* row-access copies for assembly / PCG: CSR(A) and full symmetric CSR(Q) */
    i64 *Arp, *Aci; f64 *Arx; i64 *Amap;   /* Amap: CSR slot -> CSC slot */
    i64 *Qrp, *Qci, *Qmap;                 /* full symmetric CSR pattern of Q (large n, lower storage only) */
    f64 *q, *l, *u; f64 c;
    OracleSettings s;
    /* scaling */
    int scaled;
    f64 *D, *Dinv, *E, *Einv, sc_c, sc_cinv;
    /* iterate */
    f64 *x, *y, *Ax, *Qx, *Aty, *xbar, *ybar;
    int initialized;
    f64 *temp_m, *temp_n, *temp_2m;
    f64 *mu, *isq;      /* isq = 1/sqrt(mu): the reference's `sqrt_mu` (iteration.c:112-113) */
    f64 isq_mu_min, sigma, tau, eps_in, norm_q;
    f64 *At_scale;
    i64 n_mu_changed;
    f64 *dx, *dy, *Qdx, *Adx, *Atdy;
    f64 *w, *df, *res_prim, *res_dual, *res_prim_old, *res_prim_in, *res_dual_in, *rhs;
    f64 ls_eta, ls_beta, *ls_delta, *ls_alpha;
    f64 *ls_t; i64 *ls_idx; i64 *ls_L, *ls_P, *ls_J;
    /* Newton state */
    int reset_newton;
    i64 *active, *active_old, n_active, *enter, n_enter, *leave, n_leave;
    /* factor state: K = Q + sigma_f I + A' diag(d) A */
    f64 *d, sigma_f; int factor_valid; int factor_dirty;
    f64 *K;             /* dense n*n column-major, lower: L (unit) and D on diagonal */
    int linsolve;       /* 0 dense LDL', 1 Jacobi-PCG */
    f64 pcg_tol; i64 pcg_maxit;
    f64 deadline;       /* absolute CLOCK_MONOTONIC seconds; 0 = none (cpu_baseline sampling only) */
    f64 *pc_r, *pc_z, *pc_p, *pc_Kp, *pc_diag, *pc_t;
    /* compact 32-bit copies streamed by the Jacobi-PCG operator on large instances (K_apply_compact): full-symmetric
     * CSR(Q) with its values laid out in row order, and -- rebuilt for every solve -- the weighted rows (d_i != 0)
     * of A as CSR and the columns of A restricted to those rows */
    int32_t *fq_ci; f64 *fq_x; int fq_valid;
    i64 fa_k, fa_cap; i64 *fa_row, *fa_rp, *fa_cp; int32_t *fa_ci, *fa_ri; f64 *fa_rx, *fa_cx, *fa_t;
    int progress;       /* ORACLE_PROGRESS=1: one line per loop pass on stderr (hours-long fixture runs) */
    /* info */
    i64 iterations, oterations, status_val, newton_passes, lin_iters_total;
    f64 res_prim_norm, res_dual_norm, res_prim_in_norm, res_dual_in_norm, objective;
    f64 setup_time, solve_time, run_time;
    f64 *sol_x, *sol_y;
    TraceRec *trace; i64 ntrace, captrace;
    int fix_status_reset;
} Oracle;

static int oracle_threads(void);
static f64 now_s(void) {
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return (f64)t.tv_sec + 1e-9 * (f64)t.tv_nsec;
}
static f64 *dvec(i64 n) { return (f64 *)calloc((size_t)(n > 0 ? n : 1), sizeof(f64)); }
static i64 *ivec(i64 n) { return (i64 *)calloc((size_t)(n > 0 ? n : 1), sizeof(i64)); }
#define ABSV(a) (((a) < 0) ? -(a) : (a))     /* global_opts.h:135-137 */
#define MAXV(a, b) (((a) > (b)) ? (a) : (b)) /* global_opts.h:139-141 */
#define MINV(a, b) (((a) < (b)) ? (a) : (b)) /* global_opts.h:143-145 */

/* ---- dense vector helpers (lin_alg.c) ------------------------------------ */
/* lin_alg.c:59-71 */
static f64 vec_prod(const f64 *a, const f64 *b, i64 n) {
    f64 prod = 0.0; i64 i = 0;
    if (n >= 4) for (; i <= n - 4; i += 4)
        prod += (a[i]*b[i] + a[i+1]*b[i+1] + a[i+2]*b[i+2] + a[i+3]*b[i+3]);
    for (; i < n; i++) prod += a[i] * b[i];
    return prod;
}
/* lin_alg.c:107-140; max is order independent, NaN never wins a '>' */
static f64 vec_norm_inf(const f64 *a, i64 n) {
    f64 mx = 0.0;
    for (i64 i = 0; i < n; i++) { f64 s = ABSV(a[i]); mx = s > mx ? s : mx; }
    return mx;
}
/* lin_alg.c:93-98 */
static void vec_add_scaled(const f64 *a, const f64 *b, f64 *c, f64 sc, i64 n) {
    for (i64 i = 0; i < n; i++) c[i] = a[i] + sc * b[i];
}
/* lin_alg.c:163-168 */
static void vec_mid(const f64 *a, const f64 *lo, const f64 *hi, f64 *c, i64 n) {
    for (i64 i = 0; i < n; i++) c[i] = MAXV(lo[i], MINV(a[i], hi[i]));
}
static void vec_ew_prod(const f64 *a, const f64 *b, f64 *c, i64 n) { /* lin_alg.c:77-82 */
    for (i64 i = 0; i < n; i++) c[i] = a[i] * b[i];
}
static void vec_scale(f64 *a, f64 sc, i64 n) { /* lin_alg.c:45-50 */
    for (i64 i = 0; i < n; i++) a[i] *= sc;
}

/* ---- sparse kernels (restating cholmod_sdmult as used at
 *      cholmod_interface.c:132-157) ----------------------------------------- */
/* y = M x, CSC, column order; symmetric storage uses the stored triangle only */
static void csc_mv(const Csc *M, const f64 *x, f64 *y) {
    for (i64 i = 0; i < M->nrow; i++) y[i] = 0.0;
    if (M->stype == 0) {
        for (i64 j = 0; j < M->ncol; j++) {
            f64 xj = x[j];
            for (i64 k = M->p[j]; k < M->p[j+1]; k++) y[M->i[k]] += M->x[k] * xj;
        }
    } else {
        for (i64 j = 0; j < M->ncol; j++) {
            f64 xj = x[j];
            for (i64 k = M->p[j]; k < M->p[j+1]; k++) {
                i64 i = M->i[k];
                if (i == j) y[j] += M->x[k] * xj;
                else if ((M->stype < 0 && i > j) || (M->stype > 0 && i < j)) {
                    y[i] += M->x[k] * xj;
                    y[j] += M->x[k] * x[i];
                }
            }
        }
    }
}
/* y = M' x, CSC general */
static void csc_tmv(const Csc *M, const f64 *x, f64 *y) {
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (M->p[M->ncol] > 200000)
    for (i64 j = 0; j < M->ncol; j++) {
        f64 s = 0.0;
        for (i64 k = M->p[j]; k < M->p[j+1]; k++) s += M->x[k] * x[M->i[k]];
        y[j] = s;
    }
}
/* cholmod_interface.c:162-199 */
static void csc_inf_norm_cols(const Csc *M, f64 *E) {
    for (i64 j = 0; j < M->ncol; j++) {
        E[j] = 0.0;
        for (i64 k = M->p[j]; k < M->p[j+1]; k++) E[j] = MAXV(ABSV(M->x[k]), E[j]);
    }
}
static void csc_inf_norm_rows(const Csc *M, f64 *E) {
    for (i64 i = 0; i < M->nrow; i++) E[i] = 0.0;
    for (i64 j = 0; j < M->ncol; j++)
        for (i64 k = M->p[j]; k < M->p[j+1]; k++) {
            i64 i = M->i[k]; E[i] = MAXV(ABSV(M->x[k]), E[i]);
        }
}

static int csc_copy(Csc *dst, i64 nrow, i64 ncol, const i64 *p, const i64 *i, const f64 *x, int stype) {
    i64 nnz = p[ncol];
    dst->nrow = nrow; dst->ncol = ncol; dst->stype = stype;
    dst->p = ivec(ncol + 1); dst->i = ivec(nnz); dst->x = dvec(nnz);
    if (!dst->p || !dst->i || !dst->x) return 0;
    memcpy(dst->p, p, (size_t)(ncol + 1) * sizeof(i64));
    memcpy(dst->i, i, (size_t)nnz * sizeof(i64));
    memcpy(dst->x, x, (size_t)nnz * sizeof(f64));
    return 1;
}

/* build CSR pattern of A once; values are refreshed from the CSC copy */
static void build_csr_A(Oracle *o) {
    i64 m = o->m, n = o->n, nnz = o->A.p[n];
    o->Arp = ivec(m + 1); o->Aci = ivec(nnz); o->Arx = dvec(nnz); o->Amap = ivec(nnz);
    for (i64 k = 0; k < nnz; k++) o->Arp[o->A.i[k] + 1]++;
    for (i64 i = 0; i < m; i++) o->Arp[i+1] += o->Arp[i];
    i64 *next = ivec(m);
    for (i64 i = 0; i < m; i++) next[i] = o->Arp[i];
    for (i64 j = 0; j < n; j++)
        for (i64 k = o->A.p[j]; k < o->A.p[j+1]; k++) {
            i64 s = next[o->A.i[k]]++;
            o->Aci[s] = j; o->Amap[s] = k;
        }
    free(next);
}
static void refresh_csr_A(Oracle *o) {
    i64 nnz = o->A.p[o->n];
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (nnz > 200000)
    for (i64 s = 0; s < nnz; s++) o->Arx[s] = o->A.x[o->Amap[s]];
}

/* Row-gather forms of the two non-transposed products, used so that they can run on all cores.  Both give the
 * SAME BITS as csc_mv: the column-ordered scatter adds the contributions to y_i in ascending column order, and a
 * row gather over column-sorted rows performs the same additions in the same order (for the symmetric case the
 * full row r is [columns j < r from the stored triangle] ++ [stored column r from the diagonal down]). */
static void A_mv(const Oracle *o, const f64 *x, f64 *y) {
    if (!o->Arp || o->A.p[o->n] <= 200000) { csc_mv(&o->A, x, y); return; }
#pragma omp parallel for num_threads(oracle_threads()) schedule(static)
    for (i64 r = 0; r < o->m; r++) {
        f64 s = 0.0;
        for (i64 a = o->Arp[r]; a < o->Arp[r+1]; a++) s += o->Arx[a] * x[o->Aci[a]];
        y[r] = s;
    }
}
static void build_csr_Q(Oracle *o) {
    i64 n = o->n;
    if (o->Q.stype >= 0 || o->Q.p[n] <= 100000) return;
    for (i64 j = 0; j < n; j++)                 /* needs sorted columns */
        for (i64 k = o->Q.p[j] + 1; k < o->Q.p[j+1]; k++) if (o->Q.i[k] <= o->Q.i[k-1]) return;
    i64 *rp = ivec(n + 1);
    for (i64 j = 0; j < n; j++)
        for (i64 k = o->Q.p[j]; k < o->Q.p[j+1]; k++) {
            i64 i = o->Q.i[k];
            if (i > j) { rp[i+1]++; rp[j+1]++; } else if (i == j) rp[j+1]++;
        }
    for (i64 i = 0; i < n; i++) rp[i+1] += rp[i];
    i64 nz = rp[n];
    i64 *ci = ivec(nz), *mp = ivec(nz), *next = ivec(n);
    for (i64 i = 0; i < n; i++) next[i] = rp[i];
    /* pass 1: strictly-lower entries as (row i, column j), columns ascending because j ascends */
    for (i64 j = 0; j < n; j++)
        for (i64 k = o->Q.p[j]; k < o->Q.p[j+1]; k++) {
            i64 i = o->Q.i[k];
            if (i > j) { i64 s = next[i]++; ci[s] = j; mp[s] = k; }
        }
    /* pass 2: stored column j from the diagonal down = row j, columns i >= j ascending */
    for (i64 j = 0; j < n; j++)
        for (i64 k = o->Q.p[j]; k < o->Q.p[j+1]; k++) {
            i64 i = o->Q.i[k];
            if (i >= j) { i64 s = next[j]++; ci[s] = i; mp[s] = k; }
        }
    free(next);
    o->Qrp = rp; o->Qci = ci; o->Qmap = mp;
}
static void Q_mv(const Oracle *o, const f64 *x, f64 *y) {
    if (!o->Qrp) { csc_mv(&o->Q, x, y); return; }
#pragma omp parallel for num_threads(oracle_threads()) schedule(static)
    for (i64 r = 0; r < o->n; r++) {
        f64 s = 0.0;
        for (i64 a = o->Qrp[r]; a < o->Qrp[r+1]; a++) s += o->Q.x[o->Qmap[a]] * x[o->Qci[a]];
        y[r] = s;
    }
}

/* ---- scaling (scaling.c:13-91) -------------------------------------------- */
static void limit_scaling(f64 *D, i64 n) { /* scaling.c:13-18 */
    for (i64 i = 0; i < n; i++) D[i] = D[i] < O_MIN_SCALING ? 1.0 : D[i];
}
static void scale_data(Oracle *o, i64 iters) {
    i64 n = o->n, m = o->m;
    const i64 nnzA = o->A.p[n];
    const int par = nnzA > 200000;       /* max, sqrt, products: elementwise or order-independent => same bits on any thread count */
    f64 *Dt = dvec(n), *Et = dvec(m);
    for (i64 i = 0; i < n; i++) o->D[i] = 1.0;
    for (i64 i = 0; i < m; i++) o->E[i] = 1.0;
    for (i64 it = 0; it < iters; it++) {
        if (par && o->Arp) {
            /* mat_inf_norm_cols / mat_inf_norm_rows (cholmod_interface.c:162-199) over the column / row access copies */
#pragma omp parallel for num_threads(oracle_threads()) schedule(static)
            for (i64 j = 0; j < n; j++) {
                f64 e = 0.0;
                for (i64 k = o->A.p[j]; k < o->A.p[j+1]; k++) e = MAXV(ABSV(o->A.x[k]), e);
                Dt[j] = e;
            }
#pragma omp parallel for num_threads(oracle_threads()) schedule(static)
            for (i64 i = 0; i < m; i++) {
                f64 e = 0.0;
                for (i64 a = o->Arp[i]; a < o->Arp[i+1]; a++) e = MAXV(ABSV(o->A.x[o->Amap[a]]), e);
                Et[i] = e;
            }
        } else {
            csc_inf_norm_cols(&o->A, Dt);
            csc_inf_norm_rows(&o->A, Et);
        }
        limit_scaling(Dt, n); limit_scaling(Et, m);
        for (i64 i = 0; i < n; i++) Dt[i] = 1.0 / sqrt(Dt[i]);
        for (i64 i = 0; i < m; i++) Et[i] = 1.0 / sqrt(Et[i]);
        /* cholmod_scale ROW then COL (scaling.c:56-57): two separately rounded multiplications per entry */
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (par)
        for (i64 j = 0; j < n; j++) {
            for (i64 k = o->A.p[j]; k < o->A.p[j+1]; k++) o->A.x[k] *= Et[o->A.i[k]];
            for (i64 k = o->A.p[j]; k < o->A.p[j+1]; k++) o->A.x[k] *= Dt[j];
        }
        vec_ew_prod(o->D, Dt, o->D, n);
        vec_ew_prod(o->E, Et, o->E, m);
    }
    /* Q <- D Q D (cholmod_scale SYM), q <- D q  (scaling.c:66-69) */
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (par)
    for (i64 j = 0; j < n; j++) {
        f64 t = o->D[j];
        for (i64 k = o->Q.p[j]; k < o->Q.p[j+1]; k++) o->Q.x[k] *= t * o->D[o->Q.i[k]];
    }
    vec_ew_prod(o->D, o->q, o->q, n);
    /* cost scaling (scaling.c:71-79); Qx is whatever the workspace holds */
    vec_add_scaled(o->Qx, o->q, o->temp_n, 1, n);
    o->sc_c = 1 / MAXV(1.0, vec_norm_inf(o->temp_n, n));
    vec_scale(o->q, o->sc_c, n);
    for (i64 k = 0; k < o->Q.p[n]; k++) o->Q.x[k] *= o->sc_c;
    for (i64 i = 0; i < n; i++) o->Dinv[i] = 1.0 / o->D[i];
    for (i64 i = 0; i < m; i++) o->Einv[i] = 1.0 / o->E[i];
    o->sc_cinv = 1.0 / o->sc_c;
    vec_ew_prod(o->E, o->l, o->l, m);
    vec_ew_prod(o->E, o->u, o->u, m);
    free(Dt); free(Et);
}

/* ---- validation (validate.c:9-170) ---------------------------------------- */
static int validate_settings(const OracleSettings *s) {
    if (!s) return 0;
    if (s->max_iter <= 0 || s->inner_max_iter <= 0) return 0;
    if (s->eps_abs <= 0 || s->eps_abs_in <= 0) return 0;
    if (s->eps_prim_inf < 0 || s->eps_dual_inf < 0) return 0;
    if (s->rho <= 0 || s->rho >= 1) return 0;
    if (s->theta <= 0 || s->theta > 1) return 0;
    if (s->delta <= 0 || s->delta >= 1) return 0;
    if (s->mu_min <= 0) return 0;
    if (s->proximal != 0 && s->proximal != 1) return 0;
    if (s->sigma_init <= 0) return 0;
    if (s->sigma_upd <= 0 || s->sigma_upd > 1) return 0;
    if (s->sigma_min > s->sigma_init) return 0;
    if (s->scaling < 0 || s->verbose < 0 || s->print_interval < 0) return 0;
    if (s->reset_newton_iter < 0) return 0;
    return 1;
}

void oracle_default_settings(OracleSettings *s) { /* qpdo.c:24-44, constants.h:44-69 */
    s->max_time = O_INFTY; s->max_iter = 10000; s->inner_max_iter = 1000;
    s->eps_abs = 1e-6; s->eps_abs_in = 1.0; s->eps_prim_inf = 1e-6; s->eps_dual_inf = 1e-6;
    s->rho = 0.1; s->theta = 0.25; s->delta = 1e-2; s->mu_min = 1e-9;
    s->proximal = 1; s->sigma_init = 1e-3; s->sigma_upd = 1e-1; s->sigma_min = 1e-7;
    s->scaling = 10; s->verbose = 1; s->print_interval = 1; s->reset_newton_iter = 1000;
}

void oracle_cleanup(Oracle *o);

/* qpdo.c:49-212 */
Oracle *oracle_setup(i64 n, i64 m,
                     const i64 *Qp, const i64 *Qi, const f64 *Qx, int Qstype,
                     const i64 *Ap, const i64 *Ai, const f64 *Ax,
                     const f64 *q, f64 c, const f64 *l, const f64 *u,
                     const OracleSettings *s) {
    f64 t0 = now_s();
    for (i64 j = 0; j < m; j++) if (l[j] > u[j]) return NULL;   /* validate.c:9-31 */
    if (!validate_settings(s)) return NULL;
    Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
    if (!o) return NULL;
    o->n = n; o->m = m; o->s = *s; o->c = c;
    o->sigma = s->sigma_init;
    if (!csc_copy(&o->Q, n, n, Qp, Qi, Qx, Qstype) || !csc_copy(&o->A, m, n, Ap, Ai, Ax, 0)) {
        oracle_cleanup(o); return NULL;
    }
    o->q = dvec(n); memcpy(o->q, q, (size_t)n * sizeof(f64));
    o->l = dvec(m); memcpy(o->l, l, (size_t)m * sizeof(f64));
    o->u = dvec(m); memcpy(o->u, u, (size_t)m * sizeof(f64));
    o->x = dvec(n); o->y = dvec(m); o->xbar = dvec(n); o->ybar = dvec(m);
    o->Ax = dvec(m); o->Qx = dvec(n); o->Aty = dvec(n);
    o->temp_m = dvec(m); o->temp_n = dvec(n); o->temp_2m = dvec(2*m);
    o->mu = dvec(m); o->isq = dvec(m); o->At_scale = dvec(m);
    o->w = dvec(m); o->res_prim = dvec(m); o->res_prim_old = dvec(m); o->res_dual = dvec(n);
    o->res_prim_in = dvec(m); o->res_dual_in = dvec(n); o->df = dvec(n); o->rhs = dvec(n);
    o->dx = dvec(n); o->dy = dvec(m); o->Qdx = dvec(n); o->Adx = dvec(m); o->Atdy = dvec(n);
    o->ls_delta = dvec(2*m); o->ls_alpha = dvec(2*m); o->ls_t = dvec(2*m);
    o->ls_idx = ivec(2*m); o->ls_L = ivec(2*m); o->ls_P = ivec(2*m); o->ls_J = ivec(2*m);
    o->active = ivec(m); o->active_old = ivec(m); o->enter = ivec(m); o->leave = ivec(m);
    o->d = dvec(m);
    o->sol_x = dvec(n); o->sol_y = dvec(m);
    o->reset_newton = 1;
    o->pcg_tol = 1e-12; o->pcg_maxit = 20000;
    o->progress = getenv("ORACLE_PROGRESS") != NULL;
    build_csr_A(o);                     /* pattern + slot map; values are refreshed after the scaling */
    if (s->scaling) {
        o->scaled = 1;
        o->D = dvec(n); o->Dinv = dvec(n); o->E = dvec(m); o->Einv = dvec(m);
        scale_data(o, s->scaling);
        vec_ew_prod(o->Dinv, o->q, o->temp_n, n);
        o->norm_q = vec_norm_inf(o->temp_n, n);
    } else {
        o->norm_q = vec_norm_inf(o->q, n);
    }
    refresh_csr_A(o); build_csr_Q(o);
    o->status_val = ST_UNSOLVED;
    o->setup_time = now_s() - t0;
    return o;
}

void oracle_set_linsolve(Oracle *o, int mode, f64 tol, i64 maxit) {
    o->linsolve = mode;
    if (tol > 0) o->pcg_tol = tol;
    if (maxit > 0) o->pcg_maxit = maxit;
}
void oracle_set_fix_status_reset(Oracle *o, int on) { o->fix_status_reset = on; }
/* stop PCG (and, through settings.max_time, the solve) `seconds` from now */
void oracle_set_deadline(Oracle *o, f64 seconds) { o->deadline = seconds > 0 ? now_s() + seconds : 0.0; }

/* iteration.c:185-221 */
static f64 compute_objective(Oracle *o) {
    f64 obj = 0; i64 n = o->n, i = 0;
    const f64 *Qx = o->Qx, *x = o->x, *q = o->q; f64 sg = o->sigma;
    if (o->s.proximal) {
        if (n >= 4) for (; i <= n - 4; i += 4)
            obj += (0.5*(Qx[i]   - x[i]  *sg) + q[i]  )*x[i]
                 + (0.5*(Qx[i+1] - x[i+1]*sg) + q[i+1])*x[i+1]
                 + (0.5*(Qx[i+2] - x[i+2]*sg) + q[i+2])*x[i+2]
                 + (0.5*(Qx[i+3] - x[i+3]*sg) + q[i+3])*x[i+3];
        for (; i < n; i++) obj += (0.5*(Qx[i] - sg*x[i]) + q[i])*x[i];
    } else {
        if (n >= 4) for (; i <= n - 4; i += 4)
            obj += (0.5*Qx[i] + q[i])*x[i] + (0.5*Qx[i+1] + q[i+1])*x[i+1]
                 + (0.5*Qx[i+2] + q[i+2])*x[i+2] + (0.5*Qx[i+3] + q[i+3])*x[i+3];
        for (; i < n; i++) obj += (0.5*Qx[i] + q[i])*x[i];
    }
    if (o->scaled) obj *= o->sc_cinv;
    obj += o->c;
    return obj;
}

/* iteration.c:98-122 */
static void initialize_mu(Oracle *o) {
    i64 n = o->n, m = o->m;
    f64 f = 0.5 * vec_prod(o->x, o->Qx, n) + vec_prod(o->q, o->x, n);
    vec_mid(o->Ax, o->l, o->u, o->temp_m, m);
    vec_add_scaled(o->Ax, o->temp_m, o->temp_m, -1, m);
    for (i64 i = 0; i < m; i++) {
        f64 v = 0.1 * MAXV(1, 0.5 * o->temp_m[i] * o->temp_m[i]) / MAXV(1, ABSV(f));
        o->mu[i] = MAXV(1e-3, MINV(1e3, v));
    }
    for (i64 i = 0; i < m; i++) o->isq[i] = sqrt(o->mu[i]);
    for (i64 i = 0; i < m; i++) o->isq[i] = 1.0 / o->isq[i];
    o->isq_mu_min = 1 / sqrt(o->s.mu_min);
}

/* qpdo.c:217-299 */
void oracle_warm_start(Oracle *o, const f64 *xw, const f64 *yw) {
    f64 t0 = now_s();
    i64 n = o->n, m = o->m;
    o->sigma = o->s.sigma_init;
    if (o->status_val != ST_UNSOLVED) o->setup_time = 0;
    if (xw) {
        memcpy(o->x, xw, (size_t)n * sizeof(f64));
        if (o->scaled) vec_ew_prod(o->x, o->Dinv, o->x, n);
        memcpy(o->xbar, o->x, (size_t)n * sizeof(f64));
        memcpy(o->dx, o->x, (size_t)n * sizeof(f64));
        Q_mv(o, o->dx, o->Qdx);
        if (o->s.proximal) vec_add_scaled(o->Qdx, o->x, o->Qx, o->sigma, n);
        else memcpy(o->Qx, o->Qdx, (size_t)n * sizeof(f64));
        A_mv(o, o->dx, o->Adx);
        memcpy(o->Ax, o->Adx, (size_t)m * sizeof(f64));
        o->objective = compute_objective(o);
    } else {
        memset(o->x, 0, (size_t)n * sizeof(f64)); memset(o->xbar, 0, (size_t)n * sizeof(f64));
        memset(o->Qx, 0, (size_t)n * sizeof(f64)); memset(o->Ax, 0, (size_t)m * sizeof(f64));
        o->objective = 0.0;
    }
    if (yw) {
        memcpy(o->y, yw, (size_t)m * sizeof(f64));
        if (o->scaled) { vec_ew_prod(o->y, o->Einv, o->y, m); vec_scale(o->y, o->sc_c, m); }
        memcpy(o->ybar, o->y, (size_t)m * sizeof(f64));
        memcpy(o->dy, o->y, (size_t)m * sizeof(f64));
        csc_tmv(&o->A, o->dy, o->Atdy);
        memcpy(o->Aty, o->Atdy, (size_t)n * sizeof(f64));
    } else {
        memset(o->y, 0, (size_t)m * sizeof(f64)); memset(o->ybar, 0, (size_t)m * sizeof(f64));
        memset(o->Aty, 0, (size_t)n * sizeof(f64));
    }
    initialize_mu(o);
    o->initialized = 1;
    o->setup_time += now_s() - t0;
}

/* iteration.c:30-60 */
static void compute_outer_residuals(Oracle *o) {
    i64 n = o->n, m = o->m;
    if (o->scaled) {
        vec_ew_prod(o->E, o->y, o->temp_m, m);
        vec_scale(o->temp_m, o->sc_cinv, m);
        vec_ew_prod(o->E, o->temp_m, o->temp_m, m);
        vec_add_scaled(o->Ax, o->temp_m, o->temp_m, 1, m);
    } else vec_add_scaled(o->Ax, o->y, o->temp_m, 1, m);
    vec_mid(o->temp_m, o->l, o->u, o->temp_m, m);
    vec_add_scaled(o->Ax, o->temp_m, o->res_prim, -1, m);
    vec_add_scaled(o->Qx, o->q, o->df, 1, n);
    if (o->s.proximal) {
        vec_add_scaled(o->df, o->x, o->res_dual, -o->sigma, n);
        vec_add_scaled(o->res_dual, o->Aty, o->res_dual, 1, n);
    } else vec_add_scaled(o->df, o->Aty, o->res_dual, 1, n);
}
/* termination.c:35-53 */
static void compute_outer_residuals_norm(Oracle *o) {
    i64 n = o->n, m = o->m;
    if (o->scaled) {
        vec_ew_prod(o->Einv, o->res_prim, o->temp_m, m);
        o->res_prim_norm = vec_norm_inf(o->temp_m, m);
        vec_ew_prod(o->Dinv, o->res_dual, o->temp_n, n);
        o->res_dual_norm = vec_norm_inf(o->temp_n, n);
        o->res_dual_norm *= o->sc_cinv;
    } else {
        o->res_prim_norm = vec_norm_inf(o->res_prim, m);
        o->res_dual_norm = vec_norm_inf(o->res_dual, n);
    }
}
/* iteration.c:65-93 */
static void compute_inner_residuals(Oracle *o) {
    i64 n = o->n, m = o->m;
    for (i64 i = 0; i < m; i++) o->w[i] = o->Ax[i] + o->mu[i] * (o->ybar[i] - 0.5 * o->y[i]);
    vec_mid(o->w, o->l, o->u, o->temp_m, m);
    for (i64 i = 0; i < m; i++)
        o->res_prim_in[i] = o->Ax[i] + o->mu[i] * (o->ybar[i] - o->y[i]) - o->temp_m[i];
    if (o->s.proximal) vec_add_scaled(o->df, o->xbar, o->df, -o->sigma, n);
    vec_add_scaled(o->df, o->Aty, o->res_dual_in, 1, n);
}
/* termination.c:58-77 */
static void compute_inner_residuals_norm(Oracle *o) {
    i64 n = o->n, m = o->m;
    if (o->scaled) {
        vec_ew_prod(o->Einv, o->res_prim_in, o->temp_m, m);
        o->res_prim_in_norm = vec_norm_inf(o->temp_m, m);
        vec_ew_prod(o->Dinv, o->res_dual_in, o->temp_n, n);
        o->res_dual_in_norm = vec_norm_inf(o->temp_n, n);
        o->res_dual_in_norm *= o->sc_cinv;
    } else {
        o->res_prim_in_norm = vec_norm_inf(o->res_prim_in, m);
        o->res_dual_in_norm = vec_norm_inf(o->res_dual_in, n);
    }
}

/* ---- linear system K = Q + sigma_f I + A' diag(d) A ----------------------- */
/*
 * Dense LDL', natural order, no pivoting (cholmod_interface.c:107-123: NATURAL, no postorder;
 * result kept as LDL').  Two implementations with IDENTICAL results bit for bit:
 *   ldl_factor_scalar  -- the plain left-looking column loop (the definition);
 *   ldl_factor_blocked -- right-looking over panels of LDL_NB columns, OpenMP over tiles.
 * Every entry (i,j) of both receives  K_ij -= L_ik * (L_jk * D_k)  for k = 0, 1, .., j-1 in ascending
 * order, each product and each subtraction rounded separately (no FMA: -ffp-contract=off), then
 * L_ij = K_ij * (1/D_j); only the order in which *different* entries are visited differs.  The blocked
 * version exists so that the oracle can produce fixtures at production sizes (n = 1e4 .. 3e4) and serve as
 * the all-cores CPU baseline; tests/test_oracle_specs.py checks the bitwise equality.
 */
#define LDL_NB 256
#define LDL_MR 24
#define LDL_NR 8
#define LDL_TR 10     /* micro row blocks per task: 240 rows */
#define LDL_TC 16     /* micro column blocks per task: 128 columns */
static int g_ldl_force_scalar = 0;
/* threads used by the parallel regions of this file: min(cores, 16) unless ORACLE_THREADS or oracle_set_threads says
 * otherwise (the fixtures and tests must not depend on it: every parallel form is bit-identical to its serial loop) */
static int g_threads = 0;
static int oracle_threads(void) {
    if (g_threads <= 0) {
        const char *e = getenv("ORACLE_THREADS");
        int t = (e && *e) ? atoi(e) : 0;
#ifdef _OPENMP
        if (t <= 0) { t = omp_get_num_procs(); if (t > 16) t = 16; }
#else
        t = 1;
#endif
        g_threads = t < 1 ? 1 : t;
    }
    return g_threads;
}
void oracle_set_threads(int t) { g_threads = t; }
int oracle_get_threads(void) { return oracle_threads(); }
void oracle_set_ldl_scalar(int on) { g_ldl_force_scalar = on; }

static void ldl_factor_scalar(f64 *K, i64 n) {
    for (i64 j = 0; j < n; j++) {
        f64 *cj = K + j*n;
        for (i64 k = 0; k < j; k++) {
            f64 ljk = K[j + k*n];
            f64 t = ljk * K[k + k*n];          /* L_jk * D_k */
            const f64 *ck = K + k*n;
            for (i64 i = j; i < n; i++) cj[i] -= ck[i] * t;
        }
        f64 inv = 1.0 / cj[j];
        for (i64 i = j + 1; i < n; i++) cj[i] *= inv;
    }
}

#include <immintrin.h>
/* C[MR x NR] (column-major, leading dimension ldc) -= sum_k a[k][0..MR) * w[k][0..NR), k ascending, one rounding
 * per multiply and per subtract (separate vmulpd / vsubpd: this file is compiled with -ffp-contract=off).
 * a and w are packed k-major.  rows/cols: how much of the tile exists; tri: store only entries with
 * (i0 + r) >= (j0 + c). */
#define LDL_COL(c) \
    b = _mm512_set1_pd(wk[c]); \
    c0##c = _mm512_sub_pd(c0##c, _mm512_mul_pd(a0, b)); \
    c1##c = _mm512_sub_pd(c1##c, _mm512_mul_pd(a1, b)); \
    c2##c = _mm512_sub_pd(c2##c, _mm512_mul_pd(a2, b));
#define LDL_LD(c) c0##c = _mm512_loadu_pd(src + (c)*ld); c1##c = _mm512_loadu_pd(src + (c)*ld + 8); c2##c = _mm512_loadu_pd(src + (c)*ld + 16);
#define LDL_ST(c) _mm512_storeu_pd(dst + (c)*ld, c0##c); _mm512_storeu_pd(dst + (c)*ld + 8, c1##c); _mm512_storeu_pd(dst + (c)*ld + 16, c2##c);
__attribute__((target("avx512f")))
static void ldl_tile_avx512(f64 *C, i64 ldc, const f64 *a, const f64 *w, i64 kk, i64 rows, i64 cols,
                            int tri, i64 i0, i64 j0) {
    __m512d c00, c10, c20, c01, c11, c21, c02, c12, c22, c03, c13, c23, c04, c14, c24, c05, c15, c25,
            c06, c16, c26, c07, c17, c27, a0, a1, a2, b;
    f64 tmp[LDL_MR * LDL_NR];
    int full = (rows == LDL_MR && cols == LDL_NR && !tri);
    const f64 *src; f64 *dst; i64 ld;
    if (full) { src = C; ld = ldc; }
    else {
        memset(tmp, 0, sizeof tmp);
        for (i64 c = 0; c < cols; c++) for (i64 r = 0; r < rows; r++) tmp[c*LDL_MR + r] = C[c*ldc + r];
        src = tmp; ld = LDL_MR;
    }
    LDL_LD(0) LDL_LD(1) LDL_LD(2) LDL_LD(3) LDL_LD(4) LDL_LD(5) LDL_LD(6) LDL_LD(7)
    for (i64 k = 0; k < kk; k++) {
        const f64 *ak = a + k*LDL_MR, *wk = w + k*LDL_NR;
        a0 = _mm512_loadu_pd(ak); a1 = _mm512_loadu_pd(ak + 8); a2 = _mm512_loadu_pd(ak + 16);
        LDL_COL(0) LDL_COL(1) LDL_COL(2) LDL_COL(3) LDL_COL(4) LDL_COL(5) LDL_COL(6) LDL_COL(7)
    }
    if (full) { dst = C; LDL_ST(0) LDL_ST(1) LDL_ST(2) LDL_ST(3) LDL_ST(4) LDL_ST(5) LDL_ST(6) LDL_ST(7) }
    else {
        dst = tmp; LDL_ST(0) LDL_ST(1) LDL_ST(2) LDL_ST(3) LDL_ST(4) LDL_ST(5) LDL_ST(6) LDL_ST(7)
        for (i64 c = 0; c < cols; c++) for (i64 r = 0; r < rows; r++)
            if (!tri || (i0 + r) >= (j0 + c)) C[c*ldc + r] = tmp[c*LDL_MR + r];
    }
}
static void ldl_tile_generic(f64 *C, i64 ldc, const f64 *a, const f64 *w, i64 kk, i64 rows, i64 cols,
                             int tri, i64 i0, i64 j0) {
    for (i64 c = 0; c < cols; c++) {
        i64 r0 = 0;
        if (tri && (j0 + c) > i0) r0 = (j0 + c) - i0;
        for (i64 r = r0; r < rows; r++) {
            f64 v = C[c*ldc + r];
            for (i64 k = 0; k < kk; k++) v -= a[k*LDL_MR + r] * w[k*LDL_NR + c];
            C[c*ldc + r] = v;
        }
    }
}

static void ldl_factor_blocked(f64 *K, i64 n) {
    int use512 = __builtin_cpu_supports("avx512f");
    i64 nrb_max = (n + LDL_MR - 1) / LDL_MR, ncb_max = (n + LDL_NR - 1) / LDL_NR;
    f64 *Apack = (f64 *)malloc((size_t)nrb_max * LDL_MR * LDL_NB * sizeof(f64));
    f64 *Wpack = (f64 *)malloc((size_t)ncb_max * LDL_NR * LDL_NB * sizeof(f64));
    f64 *T = (f64 *)malloc((size_t)LDL_NB * LDL_NB * sizeof(f64));   /* T[j][k] = L_jk * D_k inside the panel */
    for (i64 p0 = 0; p0 < n; p0 += LDL_NB) {
        i64 nb = MINV((i64)LDL_NB, n - p0), p1 = p0 + nb;
        /* 1. diagonal block: the scalar recurrence restricted to rows/columns [p0, p1) */
        for (i64 j = p0; j < p1; j++) {
            f64 *cj = K + j*n;
            for (i64 k = p0; k < j; k++) {
                f64 t = K[j + k*n] * K[k + k*n];
                T[(j - p0)*LDL_NB + (k - p0)] = t;
                const f64 *ck = K + k*n;
                for (i64 i = j; i < p1; i++) cj[i] -= ck[i] * t;
            }
            f64 inv = 1.0 / cj[j];
            for (i64 i = j + 1; i < p1; i++) cj[i] *= inv;
        }
        if (p1 >= n) break;
        /* 2. rows below the panel: the same recurrence row by row (rows are independent) */
        i64 below = n - p1;
#pragma omp parallel for num_threads(oracle_threads()) schedule(dynamic, 1) if (below * nb > 20000)
        for (i64 c0 = 0; c0 < below; c0 += 128) {
            i64 r0 = p1 + c0, r1 = MINV(n, r0 + 128);
            for (i64 j = p0; j < p1; j++) {
                f64 *cj = K + j*n;
                for (i64 k = p0; k < j; k++) {
                    f64 t = T[(j - p0)*LDL_NB + (k - p0)];
                    const f64 *ck = K + k*n;
                    for (i64 i = r0; i < r1; i++) cj[i] -= ck[i] * t;
                }
                f64 inv = 1.0 / cj[j];
                for (i64 i = r0; i < r1; i++) cj[i] *= inv;
            }
        }
        /* 3. pack L (rows below, k-major per block of MR rows) and W = L * D (k-major per block of NR columns) */
        i64 nrb = (below + LDL_MR - 1) / LDL_MR, ncb = (below + LDL_NR - 1) / LDL_NR;
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (below * nb > 20000)
        for (i64 b = 0; b < nrb; b++) {
            f64 *dst = Apack + (size_t)b * LDL_MR * nb;
            i64 r0 = p1 + b*LDL_MR;
            for (i64 k = 0; k < nb; k++)
                for (i64 r = 0; r < LDL_MR; r++)
                    dst[k*LDL_MR + r] = (r0 + r < n) ? K[(r0 + r) + (p0 + k)*n] : 0.0;
        }
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (below * nb > 20000)
        for (i64 b = 0; b < ncb; b++) {
            f64 *dst = Wpack + (size_t)b * LDL_NR * nb;
            i64 c0 = p1 + b*LDL_NR;
            for (i64 k = 0; k < nb; k++)
                for (i64 c = 0; c < LDL_NR; c++)
                    dst[k*LDL_NR + c] = (c0 + c < n) ? K[(c0 + c) + (p0 + k)*n] * K[(p0 + k) + (p0 + k)*n] : 0.0;
        }
        /* 4. trailing update of the lower triangle: tasks of LDL_TR x LDL_TC micro-tiles (MR x NR each); inside a
         *    task the W tile of a column block stays in L1 while the packed rows stream from L2 */
        i64 ntr = (nrb + LDL_TR - 1) / LDL_TR, ntc = (ncb + LDL_TC - 1) / LDL_TC;
#pragma omp parallel for num_threads(oracle_threads()) collapse(2) schedule(dynamic, 1) if (below * below * nb > 200000)
        for (i64 tr = ntr - 1; tr >= 0; tr--)
            for (i64 tc = 0; tc < ntc; tc++) {
                i64 rb0 = tr * LDL_TR, rb1 = MINV(nrb, rb0 + LDL_TR);
                i64 cb0 = tc * LDL_TC, cb1 = MINV(ncb, cb0 + LDL_TC);
                i64 last_row = MINV(n, p1 + rb1*LDL_MR) - 1;
                if (p1 + cb0*LDL_NR > last_row) continue;           /* task entirely above the diagonal */
                for (i64 cb = cb0; cb < cb1; cb++) {
                    i64 j0 = p1 + cb*LDL_NR, cols = MINV((i64)LDL_NR, n - j0);
                    const f64 *w = Wpack + (size_t)cb * LDL_NR * nb;
                    for (i64 rb = rb0; rb < rb1; rb++) {
                        i64 i0 = p1 + rb*LDL_MR, rows = MINV((i64)LDL_MR, n - i0);
                        if (j0 > i0 + rows - 1) continue;            /* tile entirely above the diagonal */
                        int tri = (j0 + cols - 1) > i0;
                        const f64 *a = Apack + (size_t)rb * LDL_MR * nb;
                        f64 *C = K + i0 + j0*n;
                        if (use512) ldl_tile_avx512(C, n, a, w, nb, rows, cols, tri, i0, j0);
                        else ldl_tile_generic(C, n, a, w, nb, rows, cols, tri, i0, j0);
                    }
                }
            }
    }
    free(Apack); free(Wpack); free(T);
}

static void assemble_and_factor(Oracle *o) {
    i64 n = o->n, m = o->m;
    if (!o->K) o->K = (f64 *)malloc((size_t)n * (size_t)n * sizeof(f64));
    f64 *K = o->K;
    /* every entry of K is built by one thread, contributions in the order of the serial loops (Q entry, then the
     * weighted rows of A in ascending row order, then sigma_f): column ranges per thread */
#pragma omp parallel num_threads(oracle_threads()) if (n >= 512)
    {
        i64 nt = 1, tid = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads(); tid = omp_get_thread_num();
#endif
        /* balance the triangle: column j has n-j rows */
        i64 jlo = (i64)((f64)n * (1.0 - sqrt(1.0 - (f64)tid / (f64)nt)));
        i64 jhi = (tid == nt - 1) ? n : (i64)((f64)n * (1.0 - sqrt(1.0 - (f64)(tid + 1) / (f64)nt)));
        if (jlo > n) jlo = n;
        if (jhi > n) jhi = n;
        if (jhi > jlo) memset(K + jlo*n, 0, (size_t)(jhi - jlo) * (size_t)n * sizeof(f64));
        /* lower triangle of Q (either stored triangle, or lower part of full storage) */
        if (o->Q.stype > 0) {
            for (i64 j = 0; j < n; j++)
                for (i64 k = o->Q.p[j]; k < o->Q.p[j+1]; k++) {
                    i64 i = o->Q.i[k];
                    if (i <= j && i >= jlo && i < jhi) K[j + i*n] += o->Q.x[k];
                }
        } else {
            for (i64 j = jlo; j < jhi; j++)
                for (i64 k = o->Q.p[j]; k < o->Q.p[j+1]; k++) {
                    i64 i = o->Q.i[k];
                    if (i >= j) K[i + j*n] += o->Q.x[k];
                }
        }
        for (i64 r = 0; r < m; r++) {
            f64 dr = o->d[r];
            if (dr == 0.0) continue;
            i64 a0 = o->Arp[r], a1 = o->Arp[r+1], lo = a0, hi = a1;
            while (lo < hi) { i64 mid = (lo + hi) / 2; if (o->Aci[mid] < jlo) lo = mid + 1; else hi = mid; }
            for (i64 a = lo; a < a1 && o->Aci[a] < jhi; a++) {
                i64 j = o->Aci[a]; f64 vj = o->Arx[a] * dr;
                for (i64 b = a; b < a1; b++)      /* columns ascending: Aci[b] >= j */
                    K[o->Aci[b] + j*n] += vj * o->Arx[b];
            }
        }
        for (i64 j = jlo; j < jhi; j++) K[j + j*n] += o->sigma_f;
    }
    if (g_ldl_force_scalar || n < 1024) ldl_factor_scalar(K, n);
    else ldl_factor_blocked(K, n);
    o->factor_valid = 1; o->factor_dirty = 0;
}
static void ldl_solve(Oracle *o, const f64 *b, f64 *x) {
    i64 n = o->n; const f64 *K = o->K;
    for (i64 i = 0; i < n; i++) x[i] = b[i];
    for (i64 j = 0; j < n; j++) {               /* L z = b */
        f64 xj = x[j]; const f64 *cj = K + j*n;
        for (i64 i = j + 1; i < n; i++) x[i] -= cj[i] * xj;
    }
    for (i64 j = 0; j < n; j++) x[j] /= K[j + j*n];
    for (i64 j = n - 1; j >= 0; j--) {          /* L' x = z, column oriented: x_j is final, eliminate it above */
        f64 xj = x[j];
        for (i64 i = 0; i < j; i++) x[i] -= K[j + i*n] * xj;
    }
}

/* operator v -> K v using the row-access copies */
static void K_apply(Oracle *o, const f64 *v, f64 *out) {
    i64 n = o->n, m = o->m;
    Q_mv(o, v, out);
    for (i64 i = 0; i < n; i++) out[i] += o->sigma_f * v[i];
    f64 *t = o->pc_t;
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (o->A.p[n] > 200000)
    for (i64 r = 0; r < m; r++) {
        f64 s = 0.0;
        if (o->d[r] != 0.0) {
            for (i64 a = o->Arp[r]; a < o->Arp[r+1]; a++) s += o->Arx[a] * v[o->Aci[a]];
            s *= o->d[r];
        }
        t[r] = s;
    }
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (o->A.p[n] > 200000)
    for (i64 j = 0; j < n; j++) {
        f64 s = 0.0;
        for (i64 k = o->A.p[j]; k < o->A.p[j+1]; k++) s += o->A.x[k] * t[o->A.i[k]];
        out[j] += s;
    }
}
/* ---- compact operator for the hours-long full-size records ------------------------------------------------
 * K_apply streams 16 bytes per stored entry of A (64-bit indices), all of A for the transposed half, and reaches Q
 * through a slot map.  K_apply_compact gives the SAME BITS from a third of the traffic:
 *  - a row r with d_r == 0 has t_r = +0.0 in K_apply, and a term A_rj * (+0.0) = (+-)0.0 never changes a column sum
 *    that starts at +0.0 (round-to-nearest: x + (+-)0 = x for x != 0, (+0) + (+-0) = +0, and an exact cancellation
 *    gives +0, so the running sum is never -0.0): dropping those terms keeps every bit;
 *  - the remaining additions run in the same ascending order (columns of a row / rows of a column);
 *  - Q's values are copied once into row order (same values, same order as Q_mv reads them).
 * tests/test_oracle_specs.py compares both operators bit for bit. */
static int compact_ok(const Oracle *o) {
    return o->Arp && o->Qrp && o->A.p[o->n] > 200000 && o->n < 2000000000 && o->m < 2000000000 && !getenv("ORACLE_NO_COMPACT");
}
static void compact_build(Oracle *o) {
    i64 n = o->n, m = o->m;
    if (!o->fq_valid) {
        i64 nz = o->Qrp[n];
        if (!o->fq_ci) { o->fq_ci = (int32_t *)malloc((size_t)nz * sizeof(int32_t)); o->fq_x = dvec(nz); }
#pragma omp parallel for num_threads(oracle_threads()) schedule(static)
        for (i64 a = 0; a < nz; a++) { o->fq_ci[a] = (int32_t)o->Qci[a]; o->fq_x[a] = o->Q.x[o->Qmap[a]]; }
        o->fq_valid = 1;
    }
    if (!o->fa_row) { o->fa_row = ivec(m); o->fa_rp = ivec(m + 1); o->fa_cp = ivec(n + 1); o->fa_t = dvec(m); }
    i64 *cidx = o->ls_idx;          /* scratch, 2m entries, free between line searches: row -> compact index */
    i64 k = 0;
    o->fa_rp[0] = 0;
    for (i64 r = 0; r < m; r++) {
        cidx[r] = -1;
        if (o->d[r] != 0.0) { cidx[r] = k; o->fa_row[k] = r; o->fa_rp[k+1] = o->fa_rp[k] + (o->Arp[r+1] - o->Arp[r]); k++; }
    }
    o->fa_k = k;
    i64 nz = o->fa_rp[k];
    if (nz > o->fa_cap) {
        free(o->fa_ci); free(o->fa_ri); free(o->fa_rx); free(o->fa_cx);
        o->fa_cap = nz + nz / 8 + 1024;
        o->fa_ci = (int32_t *)malloc((size_t)o->fa_cap * sizeof(int32_t)); o->fa_ri = (int32_t *)malloc((size_t)o->fa_cap * sizeof(int32_t));
        o->fa_rx = dvec(o->fa_cap); o->fa_cx = dvec(o->fa_cap);
    }
#pragma omp parallel for num_threads(oracle_threads()) schedule(static)
    for (i64 c = 0; c < k; c++) {
        i64 r = o->fa_row[c], dst = o->fa_rp[c];
        for (i64 a = o->Arp[r]; a < o->Arp[r+1]; a++, dst++) { o->fa_ci[dst] = (int32_t)o->Aci[a]; o->fa_rx[dst] = o->Arx[a]; }
    }
#pragma omp parallel for num_threads(oracle_threads()) schedule(static)
    for (i64 j = 0; j < n; j++) {
        i64 c = 0;
        for (i64 a = o->A.p[j]; a < o->A.p[j+1]; a++) c += (cidx[o->A.i[a]] >= 0);
        o->fa_cp[j+1] = c;
    }
    o->fa_cp[0] = 0;
    for (i64 j = 0; j < n; j++) o->fa_cp[j+1] += o->fa_cp[j];
#pragma omp parallel for num_threads(oracle_threads()) schedule(static)
    for (i64 j = 0; j < n; j++) {
        i64 dst = o->fa_cp[j];
        for (i64 a = o->A.p[j]; a < o->A.p[j+1]; a++) {
            i64 c = cidx[o->A.i[a]];
            if (c >= 0) { o->fa_ri[dst] = (int32_t)c; o->fa_cx[dst] = o->A.x[a]; dst++; }
        }
    }
}
static void K_apply_compact(Oracle *o, const f64 *v, f64 *out) {
    i64 n = o->n, k = o->fa_k;
    f64 *t = o->fa_t;
    const f64 sf = o->sigma_f;
#pragma omp parallel num_threads(oracle_threads())
    {
#pragma omp for schedule(static) nowait
        for (i64 r = 0; r < n; r++) {
            f64 s = 0.0;
            for (i64 a = o->Qrp[r]; a < o->Qrp[r+1]; a++) s += o->fq_x[a] * v[o->fq_ci[a]];
            out[r] = s + sf * v[r];
        }
#pragma omp for schedule(static)
        for (i64 c = 0; c < k; c++) {
            f64 s = 0.0;
            for (i64 a = o->fa_rp[c]; a < o->fa_rp[c+1]; a++) s += o->fa_rx[a] * v[o->fa_ci[a]];
            t[c] = s * o->d[o->fa_row[c]];
        }
#pragma omp for schedule(static)
        for (i64 j = 0; j < n; j++) {
            f64 s = 0.0;
            for (i64 a = o->fa_cp[j]; a < o->fa_cp[j+1]; a++) s += o->fa_cx[a] * t[o->fa_ri[a]];
            out[j] += s;
        }
    }
}
/* test hook: out = K v through either operator, for the (sigma_f, d) held by the workspace */
void oracle_K_apply(Oracle *o, const f64 *v, f64 *out, int compact) {
    if (!o->pc_t) o->pc_t = dvec(o->m);
    if (compact) { compact_build(o); K_apply_compact(o, v, out); } else K_apply(o, v, out);
}
void oracle_set_factor_state(Oracle *o, f64 sigma_f, const f64 *d) {
    o->sigma_f = sigma_f; memcpy(o->d, d, (size_t)o->m * sizeof(f64));
}
int oracle_compact_ok(const Oracle *o) { return compact_ok(o); }

static i64 pcg_solve(Oracle *o, const f64 *b, f64 *x) {
    i64 n = o->n, m = o->m;
    if (!o->pc_r) {
        o->pc_r = dvec(n); o->pc_z = dvec(n); o->pc_p = dvec(n); o->pc_Kp = dvec(n);
        o->pc_diag = dvec(n); o->pc_t = dvec(m);
    }
    const int compact = compact_ok(o);
    if (compact) compact_build(o);
    f64 *r = o->pc_r, *z = o->pc_z, *p = o->pc_p, *Kp = o->pc_Kp, *dg = o->pc_diag;
    for (i64 j = 0; j < n; j++) dg[j] = o->sigma_f;
    for (i64 j = 0; j < n; j++)
        for (i64 k = o->Q.p[j]; k < o->Q.p[j+1]; k++) if (o->Q.i[k] == j) dg[j] += o->Q.x[k];
#pragma omp parallel for num_threads(oracle_threads()) schedule(static) if (o->A.p[n] > 200000)
    for (i64 j = 0; j < n; j++) {
        f64 s = 0.0;
        for (i64 k = o->A.p[j]; k < o->A.p[j+1]; k++) s += o->A.x[k] * o->A.x[k] * o->d[o->A.i[k]];
        dg[j] += s;
    }
    f64 bnorm = sqrt(vec_prod(b, b, n));
    for (i64 i = 0; i < n; i++) { x[i] = 0.0; r[i] = b[i]; z[i] = r[i] / dg[i]; p[i] = z[i]; }
    if (bnorm == 0.0) return 0;
    f64 rz = vec_prod(r, z, n);
    i64 it = 0;
    for (; it < o->pcg_maxit; it++) {
        if (o->deadline > 0 && now_s() > o->deadline) break;
        if (compact) K_apply_compact(o, p, Kp); else K_apply(o, p, Kp);
        f64 alpha = rz / vec_prod(p, Kp, n);
        for (i64 i = 0; i < n; i++) { x[i] += alpha * p[i]; r[i] -= alpha * Kp[i]; }
        f64 rn = sqrt(vec_prod(r, r, n));
        if (rn <= o->pcg_tol * bnorm) { it++; break; }
        for (i64 i = 0; i < n; i++) z[i] = r[i] / dg[i];
        f64 rz2 = vec_prod(r, z, n);
        f64 beta = rz2 / rz; rz = rz2;
        for (i64 i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
    }
    return it;
}

/* newton.c:96-126 */
static void active_constraints(Oracle *o) {
    o->n_active = 0;
    for (i64 i = 0; i < o->m; i++) {
        if ((o->w[i] <= o->l[i]) || (o->w[i] >= o->u[i])) { o->active[i] = 1; o->n_active++; }
        else o->active[i] = 0;
    }
}
static void enter_leave_constraints(Oracle *o) {
    o->n_enter = 0; o->n_leave = 0;
    for (i64 i = 0; i < o->m; i++) {
        if (o->active[i] && !o->active_old[i]) o->enter[o->n_enter++] = i;
        if (!o->active[i] && o->active_old[i]) o->leave[o->n_leave++] = i;
    }
}

/* newton.c:13-69 with the factor-state rules of cholmod_interface.c:8-72 */
static void newton_direction(Oracle *o, TraceRec *tr) {
    i64 n = o->n, m = o->m;
    active_constraints(o);
    enter_leave_constraints(o);
    if ((o->reset_newton && o->n_active) || (o->n_enter + o->n_leave) > O_MAX_RANK_UPDATE) {
        o->reset_newton = 0;
        /* ldlcholQAtmuA: aat of the scaled columns => weight isq^2 on active rows */
        for (i64 i = 0; i < m; i++) o->d[i] = o->active[i] ? o->isq[i] * o->isq[i] : 0.0;
        o->sigma_f = o->s.proximal ? o->sigma : 0.0;
        o->factor_dirty = 1;
        tr->factor_branch = 0;
    } else if (o->n_active) {
        for (i64 k = 0; k < o->n_enter; k++) { i64 i = o->enter[k]; o->d[i] += o->isq[i] * o->isq[i]; }
        for (i64 k = 0; k < o->n_leave; k++) { i64 i = o->leave[k]; o->d[i] -= o->isq[i] * o->isq[i]; }
        if (o->n_enter || o->n_leave) o->factor_dirty = 1;
        tr->factor_branch = 1;
    } else {
        /* ldlchol(Q): fresh factor of Q + sigma I every such pass; reset flag untouched */
        int changed = (o->sigma_f != (o->s.proximal ? o->sigma : 0.0)) || !o->factor_valid;
        for (i64 i = 0; i < m; i++) { if (o->d[i] != 0.0) changed = 1; o->d[i] = 0.0; }
        o->sigma_f = o->s.proximal ? o->sigma : 0.0;
        if (changed) o->factor_dirty = 1;
        tr->factor_branch = 2;
    }
    for (i64 i = 0; i < m; i++) {
        o->dy[i] = o->res_prim_in[i] / o->mu[i];
        if (!o->active[i]) o->dy[i] *= 2;
    }
    csc_tmv(&o->A, o->dy, o->Atdy);
    for (i64 i = 0; i < n; i++) o->rhs[i] = -o->res_dual_in[i] - o->Atdy[i];
    if (o->linsolve == 0) {
        if (o->factor_dirty || !o->factor_valid) assemble_and_factor(o);
        ldl_solve(o, o->rhs, o->dx);
        tr->lin_iters = 0;
    } else {
        tr->lin_iters = pcg_solve(o, o->rhs, o->dx);
        o->lin_iters_total += tr->lin_iters;
    }
    Q_mv(o, o->dx, o->Qdx);
    if (o->s.proximal) vec_add_scaled(o->Qdx, o->dx, o->Qdx, o->sigma, n);
    A_mv(o, o->dx, o->Adx);
    for (i64 i = 0; i < m; i++) if (o->active[i]) o->dy[i] += (o->Adx[i] / o->mu[i]);
    csc_tmv(&o->A, o->dy, o->Atdy);
    memcpy(o->active_old, o->active, (size_t)m * sizeof(i64));
}

/* stable merge sort of (t, idx) by t: glibc qsort is a stable merge sort for
 * arrays of this size, comparator linesearch.c:204-211 (ties keep index order) */
static void msort(f64 *t, i64 *ix, f64 *tt, i64 *ti, i64 lo, i64 hi) {
    if (hi - lo < 2) return;
    i64 mid = lo + (hi - lo) / 2;
    msort(t, ix, tt, ti, lo, mid); msort(t, ix, tt, ti, mid, hi);
    i64 a = lo, b = mid, k = lo;
    while (a < mid && b < hi) {
        if (t[b] < t[a]) { tt[k] = t[b]; ti[k++] = ix[b++]; }
        else             { tt[k] = t[a]; ti[k++] = ix[a++]; }
    }
    while (a < mid) { tt[k] = t[a]; ti[k++] = ix[a++]; }
    while (b < hi)  { tt[k] = t[b]; ti[k++] = ix[b++]; }
    for (k = lo; k < hi; k++) { t[k] = tt[k]; ix[k] = ti[k]; }
}

/* linesearch.c:74-158 */
static void pwa_linesearch(Oracle *o) {
    i64 M2 = 2 * o->m, nL = 0;
    f64 *dl = o->ls_delta, *al = o->ls_alpha, *t = o->ls_t; i64 *ix = o->ls_idx;
    for (i64 i = 0; i < M2; i++) o->temp_2m[i] = al[i] / dl[i];
    for (i64 i = 0; i < M2; i++) {
        if (o->temp_2m[i] > 0) { o->ls_L[i] = 1; t[nL] = o->temp_2m[i]; ix[nL] = i; nL++; }
        else o->ls_L[i] = 0;
    }
    for (i64 i = 0; i < M2; i++) o->ls_P[i] = dl[i] > 0 ? 1 : 0;
    for (i64 i = 0; i < M2; i++) o->ls_J[i] = ((o->ls_P[i] + o->ls_L[i]) == 1) ? 1 : 0;
    f64 sa = 0.0, sb = 0.0;                       /* vec_prod_ind, linesearch.c:190-199 */
    for (i64 i = 0; i < M2; i++) if (o->ls_J[i]) sa += dl[i] * dl[i];
    for (i64 i = 0; i < M2; i++) if (o->ls_J[i]) sb += dl[i] * al[i];
    f64 a = o->ls_eta + sa, b = o->ls_beta - sb;
    if (nL == 0) { o->tau = -b / a; return; }
    {
        f64 *tt = (f64 *)malloc((size_t)nL * sizeof(f64)); i64 *ti = (i64 *)malloc((size_t)nL * sizeof(i64));
        msort(t, ix, tt, ti, 0, nL);
        free(tt); free(ti);
    }
    if (b + a * t[0] > 0) { o->tau = -b / a; return; }
    i64 i = 0, iz;
    while (i < nL - 1) {
        iz = ix[i];
        if (o->ls_P[iz]) { a = a + dl[iz]*dl[iz]; b = b - dl[iz]*al[iz]; }
        else             { a = a - dl[iz]*dl[iz]; b = b + dl[iz]*al[iz]; }
        i++;
        if (b + a * t[i] > 0) { o->tau = -b / a; return; }
    }
    iz = ix[i];
    if (o->ls_P[iz]) { a = a + dl[iz]*dl[iz]; b = b - dl[iz]*al[iz]; }
    else             { a = a - dl[iz]*dl[iz]; b = b + dl[iz]*al[iz]; }
    o->tau = -b / a;
}
/* linesearch.c:8-69 (psi_one at :42-49 is dead code in the reference and is omitted) */
static void exact_linesearch(Oracle *o) {
    i64 n = o->n, m = o->m;
    vec_ew_prod(o->dy, o->mu, o->temp_m, m);
    vec_scale(o->temp_m, 0.5, m);
    o->ls_eta = vec_prod(o->dy, o->temp_m, m);
    o->ls_eta += vec_prod(o->dx, o->Qdx, n);
    o->ls_eta *= 0.5;
    o->ls_beta = vec_prod(o->y, o->temp_m, m);
    o->ls_beta += vec_prod(o->dx, o->df, n);
    o->ls_beta *= 0.5;
    vec_add_scaled(o->Adx, o->temp_m, o->temp_m, -1, m);
    vec_ew_prod(o->temp_m, o->isq, o->temp_m, m);
    memcpy(o->ls_delta + m, o->temp_m, (size_t)m * sizeof(f64));
    vec_scale(o->temp_m, -1, m);
    memcpy(o->ls_delta, o->temp_m, (size_t)m * sizeof(f64));
    vec_add_scaled(o->w, o->l, o->temp_m, -1, m);
    vec_ew_prod(o->temp_m, o->isq, o->temp_m, m);
    memcpy(o->ls_alpha, o->temp_m, (size_t)m * sizeof(f64));
    vec_add_scaled(o->u, o->w, o->temp_m, -1, m);
    vec_ew_prod(o->temp_m, o->isq, o->temp_m, m);
    memcpy(o->ls_alpha + m, o->temp_m, (size_t)m * sizeof(f64));
    pwa_linesearch(o);
}

/* iteration.c:11-25 */
static void update_iterate(Oracle *o, TraceRec *tr) {
    i64 n = o->n, m = o->m;
    newton_direction(o, tr);
    exact_linesearch(o);
    vec_add_scaled(o->x, o->dx, o->x, o->tau, n);
    vec_add_scaled(o->y, o->dy, o->y, o->tau, m);
    vec_add_scaled(o->Qx, o->Qdx, o->Qx, o->tau, n);
    vec_add_scaled(o->Ax, o->Adx, o->Ax, o->tau, m);
    vec_add_scaled(o->Aty, o->Atdy, o->Aty, o->tau, n);
}

/* iteration.c:127-168 with cholmod_interface.c:77-93 folded into the d-vector */
static void update_mu(Oracle *o) {
    i64 m = o->m;
    o->n_mu_changed = 0;
    f64 rpn = vec_norm_inf(o->res_prim, m);
    i64 *changed = o->enter;
    f64 *As = o->At_scale;
    for (i64 k = 0; k < m; k++) {
        if (ABSV(o->res_prim[k]) > MAXV(o->s.eps_abs, o->s.theta * ABSV(o->res_prim_old[k]))) {
            f64 mu_factor = 1.0 / MINV(1.0, o->s.delta * rpn / ABSV(o->res_prim[k]));
            f64 mu_new = o->mu[k] / mu_factor;
            if (mu_new >= o->s.mu_min) {
                if (o->mu[k] != mu_new) changed[o->n_mu_changed++] = k;
                o->mu[k] = mu_new;
                mu_factor = sqrt(mu_factor);
                o->isq[k] = mu_factor * o->isq[k];
                As[k] = mu_factor;
            } else {
                if (o->mu[k] != o->s.mu_min) changed[o->n_mu_changed++] = k;
                o->mu[k] = o->s.mu_min;
                As[k] = o->isq_mu_min / o->isq[k];
                o->isq[k] = o->isq_mu_min;
            }
        } else As[k] = 1.0;
    }
    if ((o->s.proximal && o->sigma > o->s.sigma_min) || (o->n_mu_changed > 0.25 * O_MAX_RANK_UPDATE)) {
        o->reset_newton = 1;
    } else if (o->n_mu_changed == 0) {
        /* nothing */
    } else {
        /* ldlupdate_mu_changed: factor += (1/mu_new - 1/mu_old) a_k a_k' for every changed k,
         * active or not (cholmod_interface.c:80-91) */
        for (i64 j = 0; j < o->n_mu_changed; j++) {
            i64 k = changed[j];
            f64 s = sqrt(1 - 1 / (As[k] * As[k]));
            f64 col = o->isq[k] * s;              /* scaled column factor */
            o->d[k] += col * col;
        }
        o->factor_dirty = 1;
    }
}
/* iteration.c:173-180 */
static void update_sigma(Oracle *o) {
    if (o->sigma > o->s.sigma_min) {
        f64 old = o->sigma;
        o->sigma = MAXV(o->sigma * o->s.sigma_upd, o->s.sigma_min);
        o->reset_newton = 1;
        vec_add_scaled(o->Qx, o->x, o->Qx, o->sigma - old, o->n);
    }
}

/* termination.c:97-151 */
static int is_primal_infeasible(Oracle *o) {
    i64 n = o->n, m = o->m; f64 eps;
    if (o->scaled) { vec_ew_prod(o->E, o->dy, o->temp_m, m); eps = o->s.eps_prim_inf * vec_norm_inf(o->temp_m, m); }
    else eps = o->s.eps_prim_inf * vec_norm_inf(o->dy, m);
    if (eps == 0) return 0;
    if (o->scaled) vec_ew_prod(o->Dinv, o->Atdy, o->Atdy, n);
    f64 oob = 0;
    if (o->scaled) {
        for (i64 i = 0; i < m; i++) {
            oob += (o->u[i] <  o->E[i] * O_INFTY) ? o->u[i] * MAXV(o->dy[i], 0) : 0;
            oob += (o->l[i] > -o->E[i] * O_INFTY) ? o->l[i] * MINV(o->dy[i], 0) : 0;
        }
    } else {
        for (i64 i = 0; i < m; i++) {
            oob += (o->u[i] <  O_INFTY) ? o->u[i] * MAXV(o->dy[i], 0) : 0;
            oob += (o->l[i] > -O_INFTY) ? o->l[i] * MINV(o->dy[i], 0) : 0;
        }
    }
    if ((vec_norm_inf(o->Atdy, n) <= eps) && (oob <= -eps)) {
        o->status_val = ST_PRIMAL_INFEASIBLE;
        if (o->scaled) { vec_scale(o->dy, o->sc_cinv, m); vec_ew_prod(o->E, o->dy, o->dy, m); }
        return 1;
    }
    return 0;
}
/* termination.c:156-216 */
static int is_dual_infeasible(Oracle *o) {
    i64 n = o->n, m = o->m; f64 eps;
    if (o->scaled) { vec_ew_prod(o->D, o->dx, o->temp_n, n); eps = o->s.eps_dual_inf * vec_norm_inf(o->temp_n, n); }
    else eps = o->s.eps_dual_inf * vec_norm_inf(o->dx, n);
    if (eps == 0) return 0;
    if (o->scaled) {
        vec_ew_prod(o->Einv, o->Adx, o->Adx, m);
        for (i64 k = 0; k < m; k++)
            if ((o->u[k] < o->E[k] * O_INFTY && o->Adx[k] >= eps) || (o->l[k] > -o->E[k] * O_INFTY && o->Adx[k] <= -eps)) return 0;
    } else {
        for (i64 k = 0; k < m; k++)
            if ((o->u[k] < O_INFTY && o->Adx[k] >= eps) || (o->l[k] > -O_INFTY && o->Adx[k] <= -eps)) return 0;
    }
    if (o->s.proximal) vec_add_scaled(o->Qdx, o->dx, o->Qdx, -o->sigma * o->tau, n);
    if (o->scaled) {
        if ((vec_norm_inf(o->Qdx, n) <= o->sc_c * eps) && (vec_prod(o->q, o->dx, n) <= -o->sc_c * eps)) {
            o->status_val = ST_DUAL_INFEASIBLE;
            vec_ew_prod(o->D, o->dx, o->dx, n);
            return 1;
        }
    } else {
        if ((vec_norm_inf(o->Qdx, n) <= eps) && (vec_prod(o->q, o->dx, n) <= -eps)) {
            o->status_val = ST_DUAL_INFEASIBLE; return 1;
        }
    }
    return 0;
}
/* termination.c:82-92 */
static void store_solution(Oracle *o) {
    if (o->scaled) {
        vec_ew_prod(o->x, o->D, o->sol_x, o->n);
        vec_scale(o->y, o->sc_cinv, o->m);
        vec_ew_prod(o->y, o->E, o->sol_y, o->m);
    } else {
        memcpy(o->sol_x, o->x, (size_t)o->n * sizeof(f64));
        memcpy(o->sol_y, o->y, (size_t)o->m * sizeof(f64));
    }
    o->objective = compute_objective(o);
}

static TraceRec *trace_push(Oracle *o) {
    if (o->ntrace == o->captrace) {
        o->captrace = o->captrace ? 2 * o->captrace : 256;
        o->trace = (TraceRec *)realloc(o->trace, (size_t)o->captrace * sizeof(TraceRec));
    }
    TraceRec *t = &o->trace[o->ntrace++];
    memset(t, 0, sizeof(*t)); t->factor_branch = -1;
    return t;
}

/* qpdo.c:304-476 */
void oracle_solve(Oracle *o) {
    if (!o->initialized) oracle_warm_start(o, NULL, NULL);
    i64 n = o->n, m = o->m;
    o->eps_in = o->s.eps_abs_in;
    o->sigma = o->s.sigma_init;
    o->reset_newton = 1;
    for (i64 i = 0; i < m; i++) o->active_old[i] = 0;
    if (o->fix_status_reset) o->status_val = ST_UNSOLVED;
    f64 t0 = now_s();
    o->ntrace = 0; o->newton_passes = 0; o->lin_iters_total = 0;
    i64 iter, oter = 0, iter_old = 0;
    for (iter = 0; iter < o->s.max_iter; iter++) {
        compute_outer_residuals(o);
        compute_outer_residuals_norm(o);
        compute_inner_residuals(o);
        compute_inner_residuals_norm(o);
        TraceRec *tr = trace_push(o);
        tr->res_prim = o->res_prim_norm; tr->res_dual = o->res_dual_norm;
        tr->res_prim_in = o->res_prim_in_norm; tr->res_dual_in = o->res_dual_in_norm;
        tr->sigma = o->sigma; tr->eps_in = o->eps_in; tr->kind = 2;
        /* termination.c:11-23 */
        if ((o->res_prim_norm > O_INFTY) || (o->res_dual_norm > O_INFTY)) { o->status_val = ST_NON_CVX; break; }
        if ((o->res_prim_norm <= o->s.eps_abs) && (o->res_dual_norm <= o->s.eps_abs)) { o->status_val = ST_SOLVED; break; }
        int inner_opt = (o->res_prim_in_norm <= o->eps_in) && (o->res_dual_in_norm <= o->eps_in);
        if (((iter > iter_old + 1) && inner_opt) || (iter == iter_old + o->s.inner_max_iter)) {
            tr->kind = 1;
            if (iter < iter_old + o->s.inner_max_iter) {
                if (o->s.eps_prim_inf > 0) {
                    vec_add_scaled(o->y, o->ybar, o->dy, -1, m);
                    csc_tmv(&o->A, o->dy, o->Atdy);
                    if (is_primal_infeasible(o)) break;
                }
                if (o->s.eps_dual_inf > 0) {
                    vec_add_scaled(o->x, o->xbar, o->dx, -1, n);
                    Q_mv(o, o->dx, o->Qdx);
                    A_mv(o, o->dx, o->Adx);
                    if (is_dual_infeasible(o)) break;
                }
            }
            memcpy(o->xbar, o->x, (size_t)n * sizeof(f64));
            memcpy(o->ybar, o->y, (size_t)m * sizeof(f64));
            if ((oter > 0) && (o->res_prim_norm > o->s.eps_abs)) update_mu(o);
            if (o->s.proximal && (oter > 0) && (o->res_dual_norm > o->s.eps_abs)) update_sigma(o);
            if (iter < iter_old + o->s.inner_max_iter)
                o->eps_in = MAXV(o->s.rho * o->eps_in, 0.1 * o->s.eps_abs);
            memcpy(o->res_prim_old, o->res_prim, (size_t)m * sizeof(f64));
            oter++; iter_old = iter;
        } else {
            tr->kind = 0;
            /* reset_newton_iter == 0 is a modulo-by-zero in the reference (qpdo.c:434);
             * guarded here: 0 means "no periodic refactor" */
            if (o->s.reset_newton_iter > 0 && iter % o->s.reset_newton_iter == 0) o->reset_newton = 1;
            update_iterate(o, tr);
            tr->tau = o->tau; tr->n_active = o->n_active; tr->n_enter = o->n_enter; tr->n_leave = o->n_leave;
            o->newton_passes++;
        }
        o->run_time = o->setup_time + (now_s() - t0);
        tr = &o->trace[o->ntrace - 1]; tr->t_end = now_s() - t0;
        if (o->progress)
            fprintf(stderr, "[oracle] pass %ld kind %d act %ld +%ld -%ld tau %.6e rp %.3e rd %.3e rpi %.3e rdi %.3e lin %ld t %.0f s\n",
                    (long)iter, (int)tr->kind, (long)tr->n_active, (long)tr->n_enter, (long)tr->n_leave, tr->tau, tr->res_prim, tr->res_dual,
                    tr->res_prim_in, tr->res_dual_in, (long)tr->lin_iters, tr->t_end);
        if (o->run_time > o->s.max_time) { o->status_val = ST_MAX_TIME; break; }
        if (o->deadline > 0 && now_s() > o->deadline) { o->status_val = ST_MAX_TIME; break; }   /* cpu_baseline sampling */
    }
    if (o->status_val == ST_UNSOLVED) o->status_val = ST_MAX_ITER;
    o->iterations = iter; o->oterations = oter;
    store_solution(o);
    o->initialized = 0;
    o->solve_time = now_s() - t0;
    o->run_time = o->setup_time + o->solve_time;
}

/* qpdo.c:522-544 */
void oracle_update_bounds(Oracle *o, const f64 *l, const f64 *u) {
    i64 m = o->m;
    if (l && u) for (i64 j = 0; j < m; j++) if (l[j] > u[j]) { o->status_val = ST_ERROR; return; }
    if (l) memcpy(o->l, l, (size_t)m * sizeof(f64));
    if (u) memcpy(o->u, u, (size_t)m * sizeof(f64));
    if (o->scaled) {
        if (l) vec_ew_prod(o->E, o->l, o->l, m);
        if (u) vec_ew_prod(o->E, o->u, o->u, m);
    }
}
/* qpdo.c:549-586 */
void oracle_update_q(Oracle *o, const f64 *q) {
    i64 n = o->n;
    memcpy(o->q, q, (size_t)n * sizeof(f64));
    if (o->scaled) {
        vec_ew_prod(o->D, o->q, o->q, n);
        f64 c_old = o->sc_c;
        if (o->s.proximal) vec_add_scaled(o->Qx, o->x, o->Qx, -o->sigma, n);
        vec_add_scaled(o->q, o->Qx, o->temp_n, o->sc_cinv, n);
        o->sc_c = 1 / MAXV(1.0, vec_norm_inf(o->temp_n, n));
        o->sc_cinv = 1 / o->sc_c;
        vec_scale(o->q, o->sc_c, n);
        f64 f = o->sc_c / c_old;
        for (i64 k = 0; k < o->Q.p[n]; k++) o->Q.x[k] *= f;
        o->fq_valid = 0;
        vec_scale(o->Qx, o->sc_c / c_old, n);
        if (o->s.proximal) {
            o->sigma = o->s.sigma_init;
            vec_add_scaled(o->Qx, o->x, o->Qx, o->sigma, n);
        }
        vec_ew_prod(o->Dinv, o->q, o->temp_n, n);
        o->norm_q = vec_norm_inf(o->temp_n, n);
        o->factor_valid = 0;
    } else o->norm_q = vec_norm_inf(o->q, n);
}
/* qpdo.c:481-517.  scaling == 0 at setup followed by an increase dereferences a
 * NULL scaling struct in the reference; guarded here as an error. */
void oracle_update_settings(Oracle *o, const OracleSettings *s) {
    if (!validate_settings(s)) { o->status_val = ST_ERROR; return; }
    if (o->s.scaling > s->scaling) { o->status_val = ST_ERROR; return; }
    else if (o->s.scaling < s->scaling) {
        if (!o->scaled) { o->status_val = ST_ERROR; return; }
        i64 n = o->n, m = o->m;
        f64 *Dsave = dvec(n), *Esave = dvec(m);
        memcpy(Dsave, o->D, (size_t)n * sizeof(f64)); memcpy(Esave, o->E, (size_t)m * sizeof(f64));
        f64 c_temp = o->sc_c;
        scale_data(o, s->scaling - o->s.scaling);
        vec_ew_prod(o->D, Dsave, o->D, n); vec_ew_prod(o->E, Esave, o->E, m);
        o->sc_c *= c_temp;
        for (i64 i = 0; i < n; i++) o->Dinv[i] = 1.0 / o->D[i];
        for (i64 i = 0; i < m; i++) o->Einv[i] = 1.0 / o->E[i];
        o->sc_cinv = 1 / o->sc_c;
        refresh_csr_A(o); o->factor_valid = 0; o->fq_valid = 0;
        free(Dsave); free(Esave);
    }
    o->s = *s;
}

/* ---- getters --------------------------------------------------------------- */
i64 oracle_info_i(Oracle *o, int which) {
    switch (which) {
        case 0: return o->iterations; case 1: return o->oterations; case 2: return o->status_val;
        case 3: return o->newton_passes; case 4: return o->lin_iters_total; case 5: return o->ntrace;
    }
    return 0;
}
f64 oracle_info_f(Oracle *o, int which) {
    switch (which) {
        case 0: return o->res_prim_norm; case 1: return o->res_dual_norm;
        case 2: return o->res_prim_in_norm; case 3: return o->res_dual_in_norm;
        case 4: return o->objective; case 5: return o->setup_time; case 6: return o->solve_time;
        case 7: return o->run_time; case 8: return o->sigma; case 9: return o->tau;
        case 10: return o->sc_c;
    }
    return 0;
}
const f64 *oracle_vec(Oracle *o, int which) {
    switch (which) {
        case 0: return o->sol_x; case 1: return o->sol_y; case 2: return o->dx; case 3: return o->dy;
        case 4: return o->x; case 5: return o->y; case 6: return o->mu; case 7: return o->D; case 8: return o->E;
        case 9: return o->q; case 10: return o->l; case 11: return o->u; case 12: return o->A.x; case 13: return o->Q.x;
        case 14: return o->Qx; case 15: return o->Ax; case 16: return o->Aty; case 17: return o->d;
        case 18: return o->Qdx; case 19: return o->Adx; case 20: return o->Atdy;
        case 21: return o->res_prim_in; case 22: return o->res_dual_in; case 23: return o->ls_delta; case 24: return o->ls_alpha;
        case 25: return o->xbar; case 26: return o->ybar; case 27: return o->w;
    }
    return NULL;
}
const TraceRec *oracle_trace(Oracle *o) { return o->trace; }

void oracle_cleanup(Oracle *o) {
    if (!o) return;
    free(o->Q.p); free(o->Q.i); free(o->Q.x); free(o->A.p); free(o->A.i); free(o->A.x);
    free(o->Arp); free(o->Aci); free(o->Arx); free(o->Amap);
    free(o->Qrp); free(o->Qci); free(o->Qmap);
    free(o->q); free(o->l); free(o->u); free(o->D); free(o->Dinv); free(o->E); free(o->Einv);
    free(o->x); free(o->y); free(o->Ax); free(o->Qx); free(o->Aty); free(o->xbar); free(o->ybar);
    free(o->temp_m); free(o->temp_n); free(o->temp_2m); free(o->mu); free(o->isq); free(o->At_scale);
    free(o->dx); free(o->dy); free(o->Qdx); free(o->Adx); free(o->Atdy);
    free(o->w); free(o->df); free(o->res_prim); free(o->res_dual); free(o->res_prim_old);
    free(o->res_prim_in); free(o->res_dual_in); free(o->rhs);
    free(o->ls_delta); free(o->ls_alpha); free(o->ls_t); free(o->ls_idx); free(o->ls_L); free(o->ls_P); free(o->ls_J);
    free(o->active); free(o->active_old); free(o->enter); free(o->leave); free(o->d); free(o->K);
    free(o->pc_r); free(o->pc_z); free(o->pc_p); free(o->pc_Kp); free(o->pc_diag); free(o->pc_t);
    free(o->fq_ci); free(o->fq_x); free(o->fa_row); free(o->fa_rp); free(o->fa_cp); free(o->fa_ci); free(o->fa_ri);
    free(o->fa_rx); free(o->fa_cx); free(o->fa_t);
    free(o->sol_x); free(o->sol_y); free(o->trace);
    free(o);
}

/* ---- standalone pieces for unit tests of single kernels -------------------- */
/* y = M x for a CSC matrix (stype as above); transposed if trans != 0 (general only) */
void oracle_csc_mv(i64 nrow, i64 ncol, const i64 *p, const i64 *i, const f64 *x, int stype,
                   int trans, const f64 *v, f64 *out) {
    Csc M; M.nrow = nrow; M.ncol = ncol; M.p = (i64 *)p; M.i = (i64 *)i; M.x = (f64 *)x; M.stype = stype;
    if (trans) csc_tmv(&M, v, out); else csc_mv(&M, v, out);
}
f64 oracle_vec_norm_inf(const f64 *a, i64 n) { return vec_norm_inf(a, n); }
f64 oracle_vec_prod(const f64 *a, const f64 *b, i64 n) { return vec_prod(a, b, n); }
/* in-place dense LDL' of a caller matrix (column-major, lower triangle used): the two implementations */
void oracle_ldl_factor(f64 *K, i64 n, int blocked) { if (blocked) ldl_factor_blocked(K, n); else ldl_factor_scalar(K, n); }
/* piecewise-affine root (linesearch.c:74-158) on caller data: returns tau */
f64 oracle_pwa_linesearch(i64 m, f64 eta, f64 beta, const f64 *delta, const f64 *alpha) {
    Oracle o; memset(&o, 0, sizeof(o));
    o.m = m; o.ls_eta = eta; o.ls_beta = beta;
    o.ls_delta = (f64 *)delta; o.ls_alpha = (f64 *)alpha;
    o.temp_2m = dvec(2*m); o.ls_t = dvec(2*m); o.ls_idx = ivec(2*m);
    o.ls_L = ivec(2*m); o.ls_P = ivec(2*m); o.ls_J = ivec(2*m);
    pwa_linesearch(&o);
    free(o.temp_2m); free(o.ls_t); free(o.ls_idx); free(o.ls_L); free(o.ls_P); free(o.ls_J);
    return o.tau;
}
