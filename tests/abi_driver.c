/*
 * abi_driver.c -- a compiled-C caller of the drop-in boundary (test infrastructure).
 *
 * Includes ONLY include/qpdo.h, links against libqpdo_amd.so, and replays the call sequence of the reference's
 * host program, the MATLAB gateway (reference interfaces/mex/qpdo_mex.c): default_settings (:98-109), setup (:120-170,
 * caller-owned data freed right after), warm_start (:171-192), update_bounds (:193-213), update_q (:214-226),
 * solve (:227-281, including its rule for which of solution / certificates is valid), delete (:110-119) -- on the three
 * known-answer QPs of the reference (examples/infeasibility_tests.m:15-75; MATLAB's sparse() drops explicit zeros and
 * the class clips +-Inf to +-1e20, interfaces/mex/qpdo.m:138-139).
 *
 * The _Static_asserts pin the struct layout of the reference's DLONG + PROFILING build (include/types.h) for every
 * member the gateway reads (qpdo_mex.c:237-280, 342-357) and every member of the structs it fills.
 *
 * usage: abi_driver            -> runs on GPU 0, prints one line per case, exit code 0 iff every expectation holds
 *        abi_driver --no-device -> stops after the first qpdo_setup (used by the CPU sanitizer build: the library must
 *                                  return NULL with "no HIP device" and leak nothing)
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qpdo.h"

/* ---- layout of the public structs (x86-64, DLONG: c_int = long, c_float = double) ------------------------- */
_Static_assert(sizeof(c_int) == 8 && sizeof(c_float) == 8, "DLONG build: 64-bit c_int");
_Static_assert(sizeof(QPDOSettings) == 19 * 8, "QPDOSettings: 19 eight-byte members (types.h:96-116)");
_Static_assert(offsetof(QPDOSettings, max_time) == 0 && offsetof(QPDOSettings, max_iter) == 8 &&
               offsetof(QPDOSettings, eps_abs) == 24 && offsetof(QPDOSettings, proximal) == 88 &&
               offsetof(QPDOSettings, scaling) == 120 && offsetof(QPDOSettings, reset_newton_iter) == 144, "QPDOSettings offsets");
_Static_assert(offsetof(QPDOInfo, iterations) == 0 && offsetof(QPDOInfo, oterations) == 8 && offsetof(QPDOInfo, status) == 16 &&
               offsetof(QPDOInfo, status_val) == 48 && offsetof(QPDOInfo, res_prim_norm) == 56 &&
               offsetof(QPDOInfo, res_dual_norm) == 64 && offsetof(QPDOInfo, res_prim_in_norm) == 72 &&
               offsetof(QPDOInfo, res_dual_in_norm) == 80 && offsetof(QPDOInfo, objective) == 88 &&
               offsetof(QPDOInfo, setup_time) == 96 && offsetof(QPDOInfo, solve_time) == 104 &&
               offsetof(QPDOInfo, run_time) == 112 && sizeof(QPDOInfo) == 120, "QPDOInfo (types.h:53-72, PROFILING)");
_Static_assert(offsetof(QPDOData, n) == 0 && offsetof(QPDOData, m) == 8 && offsetof(QPDOData, Q) == 16 &&
               offsetof(QPDOData, A) == 24 && offsetof(QPDOData, q) == 32 && offsetof(QPDOData, c) == 40 &&
               offsetof(QPDOData, l) == 48 && offsetof(QPDOData, u) == 56 && sizeof(QPDOData) == 64, "QPDOData (types.h:81-90)");
_Static_assert(offsetof(QPDOSolution, x) == 0 && offsetof(QPDOSolution, y) == 8, "QPDOSolution (types.h:27-30)");
_Static_assert(offsetof(QPDOScaling, D) == 0 && offsetof(QPDOScaling, E) == 16 && offsetof(QPDOScaling, c) == 32 &&
               offsetof(QPDOScaling, cinv) == 40, "QPDOScaling (types.h:40-47)");
_Static_assert(sizeof(array_element) == 16, "array_element (types.h:14-17)");
_Static_assert(offsetof(cholmod_sparse, nrow) == 0 && offsetof(cholmod_sparse, p) == 24 && offsetof(cholmod_sparse, i) == 32 &&
               offsetof(cholmod_sparse, x) == 48 && offsetof(cholmod_sparse, stype) == 64 && offsetof(cholmod_sparse, itype) == 68 &&
               offsetof(cholmod_sparse, xtype) == 72 && offsetof(cholmod_sparse, dtype) == 76 && offsetof(cholmod_sparse, sorted) == 80 &&
               offsetof(cholmod_sparse, packed) == 84 && sizeof(cholmod_sparse) == 88, "cholmod_sparse view (CHOLMOD 3.0.x cholmod_core.h)");
/* QPDOWorkspace (types.h:147-224): the members the gateway reads, by the offsets the reference's layout gives them */
_Static_assert(offsetof(QPDOWorkspace, data) == 0 && offsetof(QPDOWorkspace, x) == 8 && offsetof(QPDOWorkspace, y) == 16 &&
               offsetof(QPDOWorkspace, Ax) == 24 && offsetof(QPDOWorkspace, initialized) == 48 &&
               offsetof(QPDOWorkspace, temp_m) == 56 && offsetof(QPDOWorkspace, mu) == 80 &&
               offsetof(QPDOWorkspace, sqrt_mu_min) == 96 && offsetof(QPDOWorkspace, n_mu_changed) == 112 &&
               offsetof(QPDOWorkspace, sigma) == 120 && offsetof(QPDOWorkspace, norm_q) == 136 &&
               offsetof(QPDOWorkspace, xbar) == 144 && offsetof(QPDOWorkspace, dx) == 160 && offsetof(QPDOWorkspace, dy) == 168 &&
               offsetof(QPDOWorkspace, tau) == 176 && offsetof(QPDOWorkspace, Qdx) == 184 && offsetof(QPDOWorkspace, w) == 208 &&
               offsetof(QPDOWorkspace, linsys_rhs) == 272 && offsetof(QPDOWorkspace, res_prim_norm_old) == 280 &&
               offsetof(QPDOWorkspace, ls_eta) == 296 && offsetof(QPDOWorkspace, ls_taus) == 328 &&
               offsetof(QPDOWorkspace, eps_prim) == 360 && offsetof(QPDOWorkspace, eps_in) == 392 &&
               offsetof(QPDOWorkspace, D_temp) == 400 && offsetof(QPDOWorkspace, chol) == 416 &&
               offsetof(QPDOWorkspace, settings) == 424 && offsetof(QPDOWorkspace, scaling) == 432 &&
               offsetof(QPDOWorkspace, solution) == 440 && offsetof(QPDOWorkspace, info) == 448 &&
               offsetof(QPDOWorkspace, timer) == 456 && sizeof(QPDOWorkspace) == 464, "QPDOWorkspace (types.h:147-224, PROFILING)");

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { failures++; printf("  FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

/* the gateway's "solve" marshalling (qpdo_mex.c:227-281) into caller arrays */
static void gateway_solve(QPDOWorkspace *w, double *x, double *y, double *pcert, double *dcert) {
    qpdo_solve(w);
    const size_t n = w->data->n, m = w->data->m;
    const long st = w->info->status_val;
    for (size_t i = 0; i < n; i++) { x[i] = NAN; dcert[i] = NAN; }
    for (size_t i = 0; i < m; i++) { y[i] = NAN; pcert[i] = NAN; }
    if (st != QPDO_PRIMAL_INFEASIBLE && st != QPDO_DUAL_INFEASIBLE) {
        memcpy(x, w->solution->x, n * sizeof(double)); memcpy(y, w->solution->y, m * sizeof(double));
    } else if (st == QPDO_PRIMAL_INFEASIBLE) memcpy(pcert, w->dy, m * sizeof(double));
    else memcpy(dcert, w->dx, n * sizeof(double));
}

/* Q = [1 0; 0 0], q = [1; c], A = [a a; 1 0; 0 1], l = [-inf; 1; 1], u = [0; 3; b]   (infeasibility_tests.m:9-12) */
typedef struct { const char *name; double a, b, c; long expect; } Kat;

static QPDOWorkspace *kat_setup(const Kat *k, int no_device) {
    /* caller-owned containers on the heap, released right after setup as the gateway does (qpdo_mex.c:165-169) */
    QPDOSettings *settings = calloc(1, sizeof(QPDOSettings));
    QPDOData *data = calloc(1, sizeof(QPDOData));
    cholmod_sparse *Q = calloc(1, sizeof(cholmod_sparse)), *A = calloc(1, sizeof(cholmod_sparse));
    long *Qp = malloc(3 * sizeof(long)), *Qi = malloc(1 * sizeof(long)); double *Qx = malloc(1 * sizeof(double));
    long *Ap = malloc(3 * sizeof(long)), *Ai = malloc(4 * sizeof(long)); double *Ax = malloc(4 * sizeof(double));
    double *q = malloc(2 * sizeof(double)), *l = malloc(3 * sizeof(double)), *u = malloc(3 * sizeof(double));
    /* lower triangle of Q, explicit zeros dropped: one entry (0,0) */
    Qp[0] = 0; Qp[1] = 1; Qp[2] = 1; Qi[0] = 0; Qx[0] = 1.0;
    long nz = 0;
    Ap[0] = 0;
    if (k->a != 0.0) { Ai[nz] = 0; Ax[nz++] = k->a; }
    Ai[nz] = 1; Ax[nz++] = 1.0; Ap[1] = nz;
    if (k->a != 0.0) { Ai[nz] = 0; Ax[nz++] = k->a; }
    Ai[nz] = 2; Ax[nz++] = 1.0; Ap[2] = nz;
    Q->nrow = Q->ncol = 2; Q->nzmax = 1; Q->p = Qp; Q->i = Qi; Q->x = Qx; Q->stype = -1; Q->itype = 2; Q->xtype = 1; Q->dtype = 0; Q->sorted = 1; Q->packed = 1;
    A->nrow = 3; A->ncol = 2; A->nzmax = (size_t)nz; A->p = Ap; A->i = Ai; A->x = Ax; A->stype = 0; A->itype = 2; A->xtype = 1; A->dtype = 0; A->sorted = 1; A->packed = 1;
    q[0] = 1.0; q[1] = k->c;
    l[0] = -QPDO_INFTY; l[1] = 1.0; l[2] = 1.0;
    u[0] = 0.0; u[1] = 3.0; u[2] = isinf(k->b) ? QPDO_INFTY : k->b;
    data->n = 2; data->m = 3; data->c = 0; data->Q = Q; data->A = A; data->q = q; data->l = l; data->u = u;
    qpdo_set_default_settings(settings);
    EXPECT(settings->max_iter == 10000 && settings->eps_abs == 1e-6 && settings->scaling == 10 && settings->reset_newton_iter == 1000,
           "default settings (constants.h:44-69)");
    settings->max_iter = 100; settings->verbose = 0;      /* infeasibility_tests.m: defaults + max_iter = 100 */
    QPDOWorkspace *w = qpdo_setup(data, settings);
    free(settings); free(data); free(Q); free(A); free(Qp); free(Qi); free(Qx); free(Ap); free(Ai); free(Ax); free(q); free(l); free(u);
    if (no_device) return w;
    EXPECT(w != NULL, "%s: qpdo_setup returned NULL", k->name);
    return w;
}

/* a seeded mid-size instance for the sanitizer run: exercises the threaded CSC -> CSR conversions of qpdo_setup (both index
 * widths, lower / upper / full storage of Q) before the library notices that there is no device */
static QPDOWorkspace *random_setup(int n, int m, int per_col, int itype, int qstype) {
    unsigned long long st = 88172645463325252ULL;
#define RND() (st ^= st << 13, st ^= st >> 7, st ^= st << 17, st)
    const size_t isz = itype == 2 ? sizeof(long) : sizeof(int);
    const long annz = (long)n * per_col;
    void *Ap = malloc((size_t)(n + 1) * isz), *Ai = malloc((size_t)annz * isz); double *Ax = malloc((size_t)annz * sizeof(double));
    void *Qp = malloc((size_t)(n + 1) * isz), *Qi = malloc((size_t)n * 3 * isz); double *Qx = malloc((size_t)n * 3 * sizeof(double));
#define SETI(a, k, v) do { if (itype == 2) ((long *)(a))[k] = (long)(v); else ((int *)(a))[k] = (int)(v); } while (0)
    long k = 0, kq = 0;
    for (int j = 0; j < n; j++) {
        SETI(Ap, j, k);
        int r = (int)(RND() % (unsigned)(m / per_col));
        for (int e = 0; e < per_col; e++) { SETI(Ai, k, r); Ax[k++] = (double)(RND() % 2001) / 1000.0 - 1.0; r += 1 + (int)(RND() % (unsigned)(m / per_col - 1)); if (r >= m) r = m - 1; }
        SETI(Qp, j, kq);
        /* tridiagonal, diagonally dominant; stored as lower / upper / both */
        if (qstype >= 0 && j > 0) { SETI(Qi, kq, j - 1); Qx[kq++] = -0.5; }
        SETI(Qi, kq, j); Qx[kq++] = 2.0;
        if (qstype <= 0 && j + 1 < n) { SETI(Qi, kq, j + 1); Qx[kq++] = -0.5; }
    }
    SETI(Ap, n, k); SETI(Qp, n, kq);
    /* rows may repeat inside a column after clamping: make them strictly increasing by construction instead */
    for (int j = 0; j < n; j++) {
        long b = itype == 2 ? ((long *)Ap)[j] : ((int *)Ap)[j], e = itype == 2 ? ((long *)Ap)[j + 1] : ((int *)Ap)[j + 1];
        for (long t = b; t < e; t++) SETI(Ai, t, (long)(t - b) * (m / per_col) + (long)((j * 7 + t) % (m / per_col)));
    }
    cholmod_sparse Q = {0}, A = {0};
    Q.nrow = Q.ncol = (size_t)n; Q.nzmax = (size_t)kq; Q.p = Qp; Q.i = Qi; Q.x = Qx; Q.stype = qstype; Q.itype = itype; Q.xtype = 1; Q.sorted = 1; Q.packed = 1;
    A.nrow = (size_t)m; A.ncol = (size_t)n; A.nzmax = (size_t)k; A.p = Ap; A.i = Ai; A.x = Ax; A.itype = itype; A.xtype = 1; A.sorted = 1; A.packed = 1;
    double *q = malloc((size_t)n * sizeof(double)), *l = malloc((size_t)m * sizeof(double)), *u = malloc((size_t)m * sizeof(double));
    for (int i = 0; i < n; i++) q[i] = (double)(RND() % 2001) / 1000.0 - 1.0;
    for (int i = 0; i < m; i++) { l[i] = -1.0; u[i] = 1.0; }
    QPDOData data = {(size_t)n, (size_t)m, &Q, &A, q, 0.0, l, u};
    QPDOSettings s; qpdo_set_default_settings(&s); s.verbose = 0;
    QPDOWorkspace *w = qpdo_setup(&data, &s);
    /* invalid inputs are rejected before any device work: crossed bounds, bad settings (validate.c:9-170) */
    l[0] = 2.0; EXPECT(qpdo_setup(&data, &s) == NULL, "crossed bounds accepted"); l[0] = -1.0;
    s.rho = 1.5; EXPECT(qpdo_setup(&data, &s) == NULL, "rho = 1.5 accepted"); s.rho = 0.1;
    free(Ap); free(Ai); free(Ax); free(Qp); free(Qi); free(Qx); free(q); free(l); free(u);
    return w;
#undef RND
#undef SETI
}

int main(int argc, char **argv) {
    const int no_device = argc > 1 && !strcmp(argv[1], "--no-device");
    const Kat kats[3] = {{"degenerate", 0.0, 3.0, 0.0, QPDO_SOLVED},
                         {"primal_infeasible", 1.0, 3.0, 0.0, QPDO_PRIMAL_INFEASIBLE},
                         {"dual_infeasible", 0.0, INFINITY, -1.0, QPDO_DUAL_INFEASIBLE}};
    if (no_device) {
        QPDOWorkspace *w = kat_setup(&kats[0], 1);
        /* on a box without a GPU the library has no CPU path: NULL.  (With a GPU this mode simply cleans up.) */
        printf("no-device mode: qpdo_setup returned %s\n", w ? "a workspace" : "NULL");
        qpdo_cleanup(w);
        qpdo_cleanup(NULL);                                  /* NULL-safe (qpdo.c:592) */
        for (int v = 0; v < 4; v++) {
            w = random_setup(3000, 6000, 80, v == 3 ? 0 : 2, v == 1 ? 1 : (v == 2 ? 0 : -1));
            printf("no-device mode: random instance (itype %d, Q stype %d): %s\n", v == 3 ? 0 : 2, v == 1 ? 1 : (v == 2 ? 0 : -1), w ? "a workspace" : "NULL");
            qpdo_cleanup(w);
        }
        return failures ? 1 : 0;
    }
    double x[2], y[3], pc[3], dc[2];
    for (int c = 0; c < 3; c++) {
        const Kat *k = &kats[c];
        QPDOWorkspace *w = kat_setup(k, 0);
        if (!w) continue;
        EXPECT(w->data->n == 2 && w->data->m == 3, "%s: workspace dimensions", k->name);
        EXPECT(w->info->status_val == QPDO_UNSOLVED && !strcmp(w->info->status, "unsolved"), "%s: status after setup", k->name);
        gateway_solve(w, x, y, pc, dc);
        printf("%-18s status %3ld (%s), %ld passes (%ld outer), x = [% .6f % .6f]\n", k->name, w->info->status_val, w->info->status,
               w->info->iterations, w->info->oterations, x[0], x[1]);
        EXPECT(w->info->status_val == k->expect, "%s: status %ld, expected %ld (infeasibility_tests.m:30,48,75)", k->name, w->info->status_val, k->expect);
        EXPECT(w->info->run_time >= w->info->solve_time && w->info->solve_time > 0, "%s: PROFILING times", k->name);
        if (k->expect == QPDO_SOLVED) {
            /* min 1/2 x1^2 + x1 s.t. 1<=x1<=3, 1<=x2<=3: x1 = 1; certificates NaN */
            EXPECT(fabs(x[0] - 1.0) <= 1e-5 && x[1] >= 1.0 - 1e-5 && x[1] <= 3.0 + 1e-5, "%s: solution", k->name);
            EXPECT(isnan(pc[0]) && isnan(dc[0]), "%s: certificates must be NaN", k->name);
            /* the MPC-style sequence on the same workspace: warm start, new bounds, new q, re-solve */
            double xw[2] = {x[0], x[1]}, yw[3] = {y[0], y[1], y[2]};
            qpdo_warm_start(w, xw, yw);
            gateway_solve(w, x, y, pc, dc);
            EXPECT(w->info->status_val == QPDO_SOLVED && fabs(x[0] - 1.0) <= 1e-5, "%s: warm-started re-solve", k->name);
            double l2[3] = {-QPDO_INFTY, 2.0, 1.0}, u2[3] = {0.0, 3.0, 3.0};
            qpdo_update_bounds(w, l2, u2);
            gateway_solve(w, x, y, pc, dc);
            EXPECT(w->info->status_val == QPDO_SOLVED && fabs(x[0] - 2.0) <= 1e-5, "%s: after update_bounds x1 = %.8f, expected 2", k->name, x[0]);
            double q2[2] = {-2.5, 0.0};                      /* min 1/2 x1^2 - 2.5 x1 on [2,3]: x1 = 2.5 */
            qpdo_update_q(w, q2);
            gateway_solve(w, x, y, pc, dc);
            EXPECT(w->info->status_val == QPDO_SOLVED && fabs(x[0] - 2.5) <= 1e-5, "%s: after update_q x1 = %.8f, expected 2.5", k->name, x[0]);
            double l3[3] = {0, 5.0, 0}, u3[3] = {0, 4.0, 0};  /* crossed bounds -> QPDO_ERROR, nothing else changes (qpdo.c:526-536) */
            qpdo_update_bounds(w, l3, u3);
            EXPECT(w->info->status_val == QPDO_ERROR, "%s: crossed bounds must give QPDO_ERROR", k->name);
            QPDOSettings s2 = *w->settings;
            s2.eps_abs = 1e-8;
            qpdo_update_settings(w, &s2);
            gateway_solve(w, x, y, pc, dc);
            EXPECT(fabs(x[0] - 2.5) <= 1e-7, "%s: after update_settings(eps_abs=1e-8) x1 = %.10f", k->name, x[0]);
        } else if (k->expect == QPDO_PRIMAL_INFEASIBLE) {
            EXPECT(isnan(x[0]) && isnan(y[0]) && isnan(dc[0]), "%s: x, y, dual cert must be NaN", k->name);
            /* certificate dy: A' dy ~ 0 and u'[dy]+ + l'[dy]- < 0 (infeasibility_tests.m:50-55) */
            const double nrm = fmax(fabs(pc[0]), fmax(fabs(pc[1]), fabs(pc[2])));
            EXPECT(nrm > 0 && fabs(pc[0] + pc[1]) <= 1e-5 * nrm && fabs(pc[0] + pc[2]) <= 1e-5 * nrm, "%s: A' dy = [%g %g]", k->name, pc[0] + pc[1], pc[0] + pc[2]);
        } else {
            EXPECT(isnan(x[0]) && isnan(y[0]) && isnan(pc[0]), "%s: x, y, primal cert must be NaN", k->name);
            /* certificate dx: Q dx = 0, q'dx < 0 (infeasibility_tests.m:77-90): dx along +x2 */
            EXPECT(fabs(dc[0]) <= 1e-5 * fabs(dc[1]) && dc[1] > 0, "%s: dx = [%g %g]", k->name, dc[0], dc[1]);
        }
        qpdo_cleanup(w);
    }
    qpdo_cleanup(NULL);
    printf("%s (%d failures)\n", failures ? "ABI DRIVER FAILED" : "abi driver ok", failures);
    return failures ? 1 : 0;
}
