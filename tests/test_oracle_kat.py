"""Pins the CPU oracle to the reference's only known-answer checks for this path:
examples/infeasibility_tests.m:30,48,75 (status 1 / -3 / -4), settings = defaults with
max_iter = 100 (infeasibility_tests.m:9-12)."""
import numpy as np
import pytest

from oracle import binding as ob
from qpdo_amd import problems


@pytest.mark.parametrize("case", ["degenerate", "primal_infeasible", "dual_infeasible"])
@pytest.mark.parametrize("linsolve", ["dense", "pcg"])
def test_reference_known_answers(case, linsolve):
    p = problems.infeasibility_kat(case)
    s = ob.default_settings(max_iter=p["max_iter"])
    o = ob.OracleSolver(p, s, linsolve=linsolve)
    assert o.ok
    r = o.solve()
    assert r["info"]["status_val"] == p["expected_status"]
    A, l, u = p["A"].toarray(), p["l"], p["u"]
    Q = problems.full_Q(p).toarray()
    if case == "degenerate":
        rp, rd = problems.kkt_residuals(p, r["x"], r["y"])
        assert rp <= 1e-6 and rd <= 1e-6
        assert np.isnan(r["prim_inf_cert"]).all() and np.isnan(r["dual_inf_cert"]).all()
    elif case == "primal_infeasible":
        # certificate quality as printed by infeasibility_tests.m:50-55
        dy = r["prim_inf_cert"]
        nrm = np.abs(dy).max()
        assert nrm > 0
        assert np.abs(A.T @ dy).max() / nrm <= 1e-6 * 10
        fin_u, fin_l = u < 1e20, l > -1e20
        oob = u[fin_u] @ np.maximum(dy[fin_u], 0) + l[fin_l] @ np.minimum(dy[fin_l], 0)
        assert oob / nrm < 0
        assert np.isnan(r["x"]).all() and np.isnan(r["y"]).all()
    else:
        dx = r["dual_inf_cert"]
        nrm = np.abs(dx).max()
        assert nrm > 0
        assert np.abs(Q @ dx).max() / nrm <= 1e-5
        assert (p["q"] @ dx) / nrm < 0
        Adx = A @ dx
        assert np.abs(Adx[(u < 1e20) & (l > -1e20)] / nrm).max() <= 1e-5
        assert (Adx[(u >= 1e20) & (l > -1e20)] / nrm).min() >= -1e-5
    o.close()
