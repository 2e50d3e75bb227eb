/*
 * qpdo.h -- C-ABI of the MI355X-native primal-dual Newton proximal QP engine.
 *
 * This single header is the drop-in boundary.  It declares, with C linkage, the
 * eight entry points a host program binds when it uses the reference solver
 * (reference include/qpdo.h:14-56) and the public structs those entry points
 * exchange (reference include/types.h, include/constants.h).  Struct member
 * names, order and types are kept identical to the reference's DLONG+PROFILING
 * build (the one interfaces/mex/qpdo_make.m:300,331 produces) so that a caller
 * compiled against the reference headers can be re-linked against
 * libqpdo_amd.so without source changes.
 *
 *   min 1/2 x'Qx + q'x + c   s.t.  l <= Ax <= u        (reference README.md:3-10)
 *
 * What differs from the reference, by design:
 *   - All iterates and matrices live in HBM for the lifetime of the workspace.
 *     Only the members documented "host mirror" below are valid host pointers;
 *     every other vector member of QPDOWorkspace is NULL on the host.
 *   - `chol` points to an opaque device backend, not to CHOLMOD objects.
 *   - The matrix inputs are read through a layout-compatible view of
 *     cholmod_sparse (CSC); CHOLMOD itself is neither needed nor linked.
 */
#ifndef QPDO_AMD_QPDO_H
#define QPDO_AMD_QPDO_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar types (reference include/global_opts.h:13-47) ---------------- */
typedef double c_float;
#ifdef QPDO_DINT
typedef int c_int;            /* reference DINT build                        */
#else
typedef long c_int;           /* reference DLONG build: SuiteSparse_long     */
#endif

/* ---- status codes (reference include/constants.h:17-26) ------------------ */
#define QPDO_SOLVED                   (1)
#define QPDO_DUAL_TERMINATED          (2)
#define QPDO_NON_CVX                  (-1)
#define QPDO_PRIMAL_DUAL_INFEASIBLE   (-2)
#define QPDO_PRIMAL_INFEASIBLE        (-3)
#define QPDO_DUAL_INFEASIBLE          (-4)
#define QPDO_MAX_ITER_REACHED         (-5)
#define QPDO_MAX_TIME_REACHED         (-6)
#define QPDO_UNSOLVED                 (-10)
#define QPDO_ERROR                    (-99)

#ifndef QPDO_NULL
#define QPDO_NULL 0
#endif
/* "infinite" bound magnitude (reference include/constants.h:37-39) */
#ifndef QPDO_INFTY
#define QPDO_INFTY ((c_float)1e20)
#endif
/* refactor instead of rank-update above this many entering+leaving rows
 * (reference include/constants.h:69, used at src/newton.c:21)              */
#define QPDO_MAX_RANK_UPDATE 100

/* ---- sparse matrix view ---------------------------------------------------
 * Layout-compatible with CHOLMOD 3.0.x `cholmod_sparse` (compressed-column).
 * If the caller already included <cholmod.h> its own definition is used.
 * Accepted inputs: xtype real (1), dtype double (0), packed, itype 0 (int32
 * indices) or 2 (int64 indices).  Q: stype -1 (lower stored), +1 (upper
 * stored) or 0 (both triangles stored).  A: stype ignored (reference
 * src/qpdo.c:109 forces 0).                                                 */
#ifndef CHOLMOD_H
typedef struct cholmod_sparse_struct {
    size_t nrow;
    size_t ncol;
    size_t nzmax;
    void  *p;        /* column pointers, ncol+1                              */
    void  *i;        /* row indices                                          */
    void  *nz;       /* unused (packed)                                      */
    void  *x;        /* values, double                                       */
    void  *z;        /* unused (real)                                        */
    int    stype;
    int    itype;    /* 0: int32 p/i, 2: int64 p/i                           */
    int    xtype;    /* 1: real                                              */
    int    dtype;    /* 0: double                                            */
    int    sorted;
    int    packed;
} cholmod_sparse;
#endif

/* ---- public structs (reference include/types.h) --------------------------- */

/* linesearch breakpoint record (types.h:14-17) */
typedef struct array_element {
    c_float x;
    size_t  i;
} array_element;

/* types.h:27-30 */
typedef struct {
    c_float *x;      /* primal solution, length n (host)                     */
    c_float *y;      /* dual solution, length m (host)                       */
} QPDOSolution;

typedef struct QPDO_TIMER QPDOTimer;

/* types.h:40-47.  Host mirrors of the Ruiz/cost scaling. */
typedef struct {
    c_float *D;
    c_float *Dinv;
    c_float *E;
    c_float *Einv;
    c_float  c;
    c_float  cinv;
} QPDOScaling;

/* types.h:53-72 (PROFILING members present) */
typedef struct {
    c_int   iterations;        /* loop passes, Newton + outer (qpdo.c:455)   */
    c_int   oterations;        /* outer (proximal) updates                   */
    char    status[32];
    c_int   status_val;
    c_float res_prim_norm;
    c_float res_dual_norm;
    c_float res_prim_in_norm;
    c_float res_dual_in_norm;
    c_float objective;
    c_float setup_time;
    c_float solve_time;
    c_float run_time;
} QPDOInfo;

/* types.h:81-90 */
typedef struct {
    size_t          n;
    size_t          m;
    cholmod_sparse *Q;         /* n x n, CSC                                  */
    cholmod_sparse *A;         /* m x n, CSC                                  */
    c_float        *q;
    c_float         c;
    c_float        *l;
    c_float        *u;
} QPDOData;

/* types.h:96-116 */
typedef struct {
    c_float max_time;
    c_int   max_iter;
    c_int   inner_max_iter;
    c_float eps_abs;
    c_float eps_abs_in;
    c_float eps_prim_inf;
    c_float eps_dual_inf;
    c_float rho;
    c_float theta;
    c_float delta;
    c_float mu_min;
    c_int   proximal;
    c_float sigma_init;
    c_float sigma_upd;
    c_float sigma_min;
    c_int   scaling;
    c_int   verbose;
    c_int   print_interval;
    c_int   reset_newton_iter;
} QPDOSettings;

/* Opaque device backend; takes the place of QPDOCholmod (types.h:121-142). */
typedef struct QPDOBackend QPDOBackend;

/* types.h:147-224.  Members marked [host] are valid host pointers after
 * qpdo_setup / qpdo_solve; all other pointers are NULL on the host because the
 * vector lives in HBM (see DESIGN.md "data layout").                        */
typedef struct {
    QPDOData *data;            /* [host] n, m, c and scaled q,l,u; Q=A=NULL   */

    c_float *x;                /* [host] mirror, synced when qpdo_solve exits */
    c_float *y;                /* [host] mirror, synced when qpdo_solve exits */
    c_float *Ax;
    c_float *Qx;
    c_float *Aty;
    c_int    initialized;

    c_float *temp_m;
    c_float *temp_n;
    c_float *temp_2m;

    c_float *mu;
    c_float *sqrt_mu;
    c_float  sqrt_mu_min;
    c_float  sqrt_delta;
    c_int    n_mu_changed;
    c_float  sigma;
    c_int    sigma_mined;
    c_float  norm_q;

    c_float *xbar;
    c_float *ybar;
    c_float *dx;               /* [host] mirror: dual-infeasibility cert.     */
    c_float *dy;               /* [host] mirror: primal-infeasibility cert.   */
    c_float  tau;
    c_float *Qdx;
    c_float *Adx;
    c_float *Atdy;

    c_float *w;
    c_float *z;
    c_float *df;
    c_float *res_prim;
    c_float *res_dual;
    c_float *res_prim_old;
    c_float *res_prim_in;
    c_float *res_dual_in;
    c_float *linsys_rhs;

    c_float  res_prim_norm_old;
    c_float  res_dual_norm_old;

    c_float  ls_eta;
    c_float  ls_beta;
    c_float *ls_delta;
    c_float *ls_alpha;
    array_element *ls_taus;
    c_int   *ls_idx_L;
    c_int   *ls_idx_P;
    c_int   *ls_idx_J;

    c_float  eps_prim;
    c_float  eps_dual;
    c_float  eps_prim_in;
    c_float  eps_dual_in;
    c_float  eps_in;

    c_float *D_temp;
    c_float *E_temp;

    QPDOBackend  *chol;        /* opaque HIP backend                          */
    QPDOSettings *settings;    /* [host]                                      */
    QPDOScaling  *scaling;     /* [host] NULL when settings->scaling == 0     */
    QPDOSolution *solution;    /* [host]                                      */
    QPDOInfo     *info;        /* [host]                                      */

    QPDOTimer    *timer;       /* [host]                                      */
} QPDOWorkspace;

/* ---- entry points ---------------------------------------------------------
 * Each replaces the reference function of the same name; semantics, argument
 * meaning, ownership (deep copies at setup) and error behaviour follow the
 * cited lines.  HIP failures surface as NULL (setup) or status QPDO_ERROR.   */

/* reference include/qpdo.h:14, src/qpdo.c:24-44 */
void qpdo_set_default_settings(QPDOSettings *settings);

/* reference include/qpdo.h:19-20, src/qpdo.c:49-212.  NULL on invalid data or
 * settings (src/validate.c) or on allocation / device failure.              */
QPDOWorkspace *qpdo_setup(const QPDOData *data, const QPDOSettings *settings);

/* reference include/qpdo.h:25-27, src/qpdo.c:217-299.  NULL x or y => zeros. */
void qpdo_warm_start(QPDOWorkspace *work, c_float *x_warm_start,
                     c_float *y_warm_start);

/* reference include/qpdo.h:32, src/qpdo.c:304-476 */
void qpdo_solve(QPDOWorkspace *work);

/* reference include/qpdo.h:37-38, src/qpdo.c:481-517 */
void qpdo_update_settings(QPDOWorkspace *work, const QPDOSettings *settings);

/* reference include/qpdo.h:43-45, src/qpdo.c:522-544.  NULL => unchanged.    */
void qpdo_update_bounds(QPDOWorkspace *work, const c_float *bmin,
                        const c_float *bmax);

/* reference include/qpdo.h:50, src/qpdo.c:549-586 */
void qpdo_update_q(QPDOWorkspace *work, const c_float *q);

/* reference include/qpdo.h:56, src/qpdo.c:591-689.  NULL-safe.               */
void qpdo_cleanup(QPDOWorkspace *work);

#ifdef __cplusplus
}
#endif
#endif /* QPDO_AMD_QPDO_H */
