/* pass_decision.h -- what a loop pass of qpdo_solve is, decided from the pass's residual norms and active-set counts.
 *
 * ONE definition for the two places that decide: the host loop (qpdo_api.c qpdo_solve, gcc) and the publishing block of the residual
 * launch (dev/vector.inc k_resid_mn, hipcc: the launch-ahead of the Newton step).  Both get the same numbers -- the four inf-norms as the
 * reference compares them (the dual ones already multiplied by cinv: termination.c:45,72), eps_in, the counts -- and so the same answer.
 *
 *   ends_nc   check_outer_optimality, termination.c:11-23 / qpdo.c:352-356: a residual beyond QPDO_INFTY (status QPDO_NON_CVX)
 *   ends_ok   ... both outer residuals within eps_abs (status QPDO_SOLVED)
 *   outer     qpdo.c:361-363: the inner problem is solved (termination.c:28-30) and at least one Newton step was made since the last
 *             outer update, or inner_max_iter passes went by
 *   branch    newton.c:21-33: 0 full factorization, 1 rank update of the kept factor, 2 no active row (Q + sigma I alone)
 */
#ifndef QPDO_PASS_DECISION_H
#define QPDO_PASS_DECISION_H

#if defined(__HIPCC__)
#define QPDO_PD_FN __host__ __device__ static inline
#else
#define QPDO_PD_FN static inline
#endif

typedef struct { int ends_nc, ends_ok, outer, branch; } QpdoPassDecision;

QPDO_PD_FN QpdoPassDecision qpdo_pass_decision(double res_prim, double res_dual, double res_prim_in, double res_dual_in, double eps_abs,
                                               double eps_in, double infty, int allow_outer /* iter > iter_old + 1 */,
                                               int force_outer /* iter == iter_old + inner_max_iter */, int reset_newton, int n_active,
                                               int n_change /* n_enter + n_leave */, int max_rank) {
    QpdoPassDecision p;
    p.ends_nc = (res_prim > infty) || (res_dual > infty);
    p.ends_ok = !p.ends_nc && (res_prim <= eps_abs) && (res_dual <= eps_abs);
    const int inner_opt = (res_prim_in <= eps_in) && (res_dual_in <= eps_in);
    p.outer = (allow_outer && inner_opt) || force_outer;
    p.branch = ((reset_newton && n_active) || n_change > max_rank) ? 0 : (n_active ? 1 : 2);
    return p;
}
#endif
