/*
 * qpdo_gen.c -- seeded synthetic QP generator (host, C, OpenMP).
 *
 * Workload synthesis for tests and bench.py; not part of the solver.  The
 * reference's demo builds its instance with MATLAB sprandsym/sprandn
 * (examples/demo_mex.m:11-15), which cannot be reproduced outside MATLAB, so
 * the instances are defined here by a counter-based generator (splitmix64
 * keyed by (seed, stream, counter)): the same (seed, shape, density) gives the
 * same instance on every machine and at every thread count.
 *
 *   A (m x n, CSC): every column holds K = max(1, round(density*m)) entries, one
 *      per row stratum, values N(0,1).
 *   Q (n x n, lower triangle CSC, stype -1): strict-lower pattern at `density`
 *      by the same stratified rule, values N(0,1)/sqrt(max(1,density*n)),
 *      diagonal Q_jj = sum_i |Q_ij| + 1e-3*U(0,1)  => diagonally dominant, PSD.
 *   q ~ N(0,1).  Bounds: l = A x0 - U(0,1), u = A x0 + U(0,1) with x0 = 0 when
 *      n_eq == 0 (then l = -U, u = +U as in demo_mex.m:14-15); the first n_eq
 *      rows are equalities l = u = A x0 with x0 ~ 0.1 N(0,1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint64_t rnd(uint64_t seed, uint64_t stream, uint64_t ctr) {
    return mix64(mix64(seed * 0x2545f4914f6cdd1dULL + stream) ^ (ctr * 0xd6e8feb86659fd93ULL));
}
static inline double uni(uint64_t seed, uint64_t stream, uint64_t ctr) {   /* (0,1) */
    return ((double)(rnd(seed, stream, ctr) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static inline double nrm(uint64_t seed, uint64_t stream, uint64_t ctr) {
    double u1 = uni(seed, stream, 2 * ctr), u2 = uni(seed, stream, 2 * ctr + 1);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

enum { S_AROW = 1, S_AVAL, S_QROW, S_QVAL, S_QDIAG, S_Q, S_L, S_U, S_X0 };

int64_t qpdo_gen_A_per_col(int64_t m, double density) {
    int64_t K = (int64_t)llround(density * (double)m);
    if (K < 1) K = 1;
    if (K > m) K = m;
    return K;
}
/* number of strict-lower entries in column j of Q */
static inline int64_t q_col_count(int64_t n, int64_t j, double density) {
    int64_t avail = n - 1 - j;
    int64_t K = (int64_t)llround(density * (double)avail);
    if (K > avail) K = avail;
    return K < 0 ? 0 : K;
}
int64_t qpdo_gen_Q_nnz(int64_t n, double density) {
    int64_t t = 0;
    for (int64_t j = 0; j < n; j++) t += 1 + q_col_count(n, j, density);
    return t;
}

/* Fills caller-allocated arrays.  Ap: n+1, Ai/Ax: n*K.  Qp: n+1, Qi/Qx: qpdo_gen_Q_nnz. */
void qpdo_gen_problem(uint64_t seed, int64_t n, int64_t m, double density, int64_t n_eq,
                      int64_t *Ap, int64_t *Ai, double *Ax,
                      int64_t *Qp, int64_t *Qi, double *Qx,
                      double *q, double *l, double *u) {
    const int64_t K = qpdo_gen_A_per_col(m, density);
    for (int64_t j = 0; j <= n; j++) Ap[j] = j * K;
#pragma omp parallel for schedule(static) if (n * K > 200000)
    for (int64_t j = 0; j < n; j++) {
        for (int64_t k = 0; k < K; k++) {
            int64_t lo = (k * m) / K, hi = ((k + 1) * m) / K;     /* stratum [lo,hi) */
            uint64_t ctr = (uint64_t)(j * K + k);
            int64_t r = lo + (int64_t)(rnd(seed, S_AROW, ctr) % (uint64_t)(hi - lo));
            Ai[j * K + k] = r;
            Ax[j * K + k] = nrm(seed, S_AVAL, ctr);
        }
    }
    /* Q pattern + off-diagonal values */
    Qp[0] = 0;
    for (int64_t j = 0; j < n; j++) Qp[j + 1] = Qp[j] + 1 + q_col_count(n, j, density);
    const double s = 1.0 / sqrt(fmax(1.0, density * (double)n));
#pragma omp parallel for schedule(dynamic, 64) if (Qp[n] > 200000)
    for (int64_t j = 0; j < n; j++) {
        int64_t base = Qp[j], Kj = Qp[j + 1] - Qp[j] - 1, avail = n - 1 - j;
        Qi[base] = j; Qx[base] = 0.0;
        for (int64_t k = 0; k < Kj; k++) {
            int64_t lo = (k * avail) / Kj, hi = ((k + 1) * avail) / Kj;
            uint64_t ctr = (uint64_t)(base + 1 + k);
            int64_t r = j + 1 + lo + (int64_t)(rnd(seed, S_QROW, ctr) % (uint64_t)(hi - lo));
            Qi[base + 1 + k] = r;
            Qx[base + 1 + k] = s * nrm(seed, S_QVAL, ctr);
        }
    }
    double *rows = (double *)calloc((size_t)n, sizeof(double));
    for (int64_t j = 0; j < n; j++)
        for (int64_t k = Qp[j] + 1; k < Qp[j + 1]; k++) {
            double a = fabs(Qx[k]);
            rows[Qi[k]] += a; rows[j] += a;
        }
    for (int64_t j = 0; j < n; j++) Qx[Qp[j]] = rows[j] + 1e-3 * uni(seed, S_QDIAG, (uint64_t)j);
    free(rows);
    for (int64_t j = 0; j < n; j++) q[j] = nrm(seed, S_Q, (uint64_t)j);
    /* bounds around A x0 */
    double *Ax0 = (double *)calloc((size_t)m, sizeof(double));
    if (n_eq > 0) {
        for (int64_t j = 0; j < n; j++) {
            double x0 = 0.1 * nrm(seed, S_X0, (uint64_t)j);
            for (int64_t k = Ap[j]; k < Ap[j + 1]; k++) Ax0[Ai[k]] += Ax[k] * x0;
        }
    }
    for (int64_t i = 0; i < m; i++) {
        if (i < n_eq) { l[i] = Ax0[i]; u[i] = Ax0[i]; }
        else {
            l[i] = Ax0[i] - uni(seed, S_L, (uint64_t)i);
            u[i] = Ax0[i] + uni(seed, S_U, (uint64_t)i);
        }
    }
    free(Ax0);
}
