"""The band direct solver (dev/band.inc): chain-structured QPs, whose Newton matrix Q + sigma I + A'DA is banded for every D, are what the
reference solves trivially at any size -- CHOLMOD factors a banded matrix in natural order in O(n b^2) (cholmod_interface.c:35-52,
107-123).  The device equivalent: lower band storage, natural-order LDL' inside the band (one workgroup, the window of b + 4 columns in LDS),
band triangular solves by one wave.  Parity as for the dense direct solver: status, counts and per-pass integers identical to the oracle,
step lengths, norms and iterates within the dense tolerances."""
import numpy as np
import pytest

from helpers import ITERATE_RTOL, assert_same_trace, close_vec
from oracle import binding as ob
from qpdo_amd import problems, solver

pytestmark = pytest.mark.gpu


def oracle_run(p, **st):
    o = ob.OracleSolver(p, ob.default_settings(**st))
    ro = o.solve(); tr = o.trace(); o.close()
    return ro, tr


def check(r, ro, tro, p):
    gi, oi = r["info"], ro["info"]
    assert (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"]), (gi, oi)
    assert_same_trace(r["trace"], tro)
    if oi["status_val"] not in (-3, -4):
        assert close_vec(r["x"], ro["x"], ITERATE_RTOL) and close_vec(r["y"], ro["y"], ITERATE_RTOL)
        rp, rd = problems.kkt_residuals(p, r["x"], r["y"])
        assert abs(rp - gi["res_prim_norm"]) <= 1e-9 and abs(rd - gi["res_dual_norm"]) <= 1e-9


@pytest.mark.parametrize("n,bw,win", [(2501, 1, 2), (2048, 3, 4), (3000, 17, 8), (2203, 64, 65), (2600, 127, 30), (2400, 126, 127)])
def test_band_solver_is_the_default_on_banded_problems_and_matches_the_oracle(n, bw, win, gpu_required, monkeypatch):
    for k in ("QPDO_LINSOLVE",):
        monkeypatch.delenv(k, raising=False)
    p = problems.banded_random_qp(100 + bw, n, bw, win=win)
    r = solver.solve_problem(p, verbose=0)
    assert r["stats"]["linsolve"] == 3 and r["stats"]["factor_count"] > 0 and r["stats"]["lin_iters"] == 0
    ro, tro = oracle_run(p)
    check(r, ro, tro, p)
    # and the dense direct solver of the same device on the same instance: the same trajectory
    monkeypatch.setenv("QPDO_LINSOLVE", "dense")
    rd = solver.solve_problem(p, verbose=0)
    assert rd["stats"]["linsolve"] == 1
    check(rd, ro, tro, p)
    assert ro["info"]["status_val"] == 1
    assert np.abs(r["x"] - rd["x"]).max() <= 1e-9 * max(1.0, np.abs(rd["x"]).max())


def test_band_solver_selection_rules(gpu_required, monkeypatch):
    monkeypatch.delenv("QPDO_LINSOLVE", raising=False)
    # half-bandwidth 128: one too many -> the usual selection
    p = problems.banded_random_qp(7, 2400, 128, win=4)
    assert solver.solve_problem(p, verbose=0, max_iter=5)["stats"]["linsolve"] == 1
    # a random sparse QP is not banded
    assert solver.solve_problem(problems.random_qp(3, 2500, 3000, 0.01), verbose=0, max_iter=5)["stats"]["linsolve"] == 1
    # small banded problems keep the dense solver by default, but take the band solver on request
    p = problems.banded_random_qp(8, 600, 5)
    assert solver.solve_problem(p, verbose=0)["stats"]["linsolve"] == 1
    monkeypatch.setenv("QPDO_LINSOLVE", "band")
    r = solver.solve_problem(p, verbose=0)
    assert r["stats"]["linsolve"] == 3
    ro, tro = oracle_run(p)
    check(r, ro, tro, p)
    # asked for on a matrix that is not banded: setup fails with a message (NULL), it does not silently pick something else
    with pytest.raises(RuntimeError):
        solver.solve_problem(problems.random_qp(3, 2500, 3000, 0.01), verbose=0)


def test_band_solver_sequences_and_settings(gpu_required, monkeypatch):
    """cold solves with default settings, without scaling and without the proximal term (a definite Q keeps K definite): the dense
    tolerances; then a warm-started re-solve after update_q / update_bounds on the same workspace: a 77-92 pass crawl on ill-conditioned
    systems where ANY two direct solvers drift apart in the last digits (measured, tools/band_dev_probe.py: per-pass norms within 1.4e-9 /
    2.6e-9 of the oracle for the band solver, 1.7e-8 / 7e-10 for the dense MFMA solver) -- there the per-pass integers must be identical
    and the final iterate within 1e-8"""
    from helpers import ITERATE_RTOL_PCG, same_trace_counts
    monkeypatch.delenv("QPDO_LINSOLVE", raising=False)
    p = problems.banded_random_qp(21, 2300, 9)
    for st in (dict(), dict(scaling=0), dict(proximal=0)):
        o = ob.OracleSolver(p, ob.default_settings(**st))
        s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0, **st)
        ro, rg = o.solve(), s.solve()
        assert s.stats()["linsolve"] == 3
        rg["trace"] = s.trace(); check(rg, ro, o.trace(), p)
        if st.get("proximal", 1) == 0:
            s.delete(); o.close()
            continue
        o.warm_start(ro["x"], ro["y"]); s.warm_start(ro["x"], ro["y"])
        q2 = p["q"] * 1.1 + 0.05
        o.update_q(q2); s.update_q(q2)
        l2, u2 = p["l"] - 0.05, p["u"] + 0.02
        o.update_bounds(l2, u2); s.update_bounds(l2, u2)
        ro, rg = o.solve(), s.solve()
        gi, oi = rg["info"], ro["info"]
        assert (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
        assert same_trace_counts(s.trace(), o.trace())
        assert close_vec(rg["x"], ro["x"], ITERATE_RTOL_PCG) and close_vec(rg["y"], ro["y"], ITERATE_RTOL_PCG)
        s.delete(); o.close()


def test_band_solver_at_a_size_no_other_direct_solver_reaches(gpu_required, monkeypatch):
    """n = 200 000 (the dense matrix would be 320 GB): solved, KKT residuals recomputed independently; the same instance through PCG for
    the counts (both exact to the tolerances of the solve, so the pass counts agree)"""
    monkeypatch.delenv("QPDO_LINSOLVE", raising=False)
    p = problems.banded_qp(5, 200_000)
    r = solver.solve_problem(p, verbose=0)
    assert r["stats"]["linsolve"] == 3 and r["info"]["status_val"] == 1
    rp, rd = problems.kkt_residuals(p, r["x"], r["y"])
    assert rp <= 1e-6 and rd <= 1e-6 and abs(rp - r["info"]["res_prim_norm"]) <= 1e-9 and abs(rd - r["info"]["res_dual_norm"]) <= 1e-9


def test_singular_banded_matrix_is_latched_and_handed_to_another_solver(gpu_required, monkeypatch):
    """round-4 advisor finding: k_band_factor took reciprocals of its pivots with no check.  banded_qp's Q (second differences) is
    singular along constants; with settings->proximal = 0 and bounds so wide that no row is active in the first pass, the first Newton
    matrix is Q itself: the last pivot is zero.  The band kernel must latch that, the device must skip the pass's iterate update, and
    the pass must be redone by another solver (band_fallbacks counts it) -- the same outcome as asking for the dense solver outright,
    never a silent inf / NaN step."""
    monkeypatch.delenv("QPDO_LINSOLVE", raising=False)
    p = problems.banded_qp(9, 2304, box=1e3, rate=1e3)
    r = solver.solve_problem(p, verbose=0, proximal=0, max_iter=60)
    assert r["stats"]["band_fallbacks"] == 1 and r["stats"]["linsolve"] == 1
    monkeypatch.setenv("QPDO_LINSOLVE", "dense")
    r1 = solver.solve_problem(p, verbose=0, proximal=0, max_iter=60)
    assert r["info"]["status_val"] == r1["info"]["status_val"] and r["info"]["iterations"] == r1["info"]["iterations"]
    # (what the dense solver makes of an exactly singular K is the reference's own behaviour -- CHOLMOD's status is not checked either,
    # cholmod_interface.c:19-29, and vec_norm_inf skips NaN (lin_alg.c:107-140), so a NaN iterate reads as "solved": the point here is
    # that the band path arrives at the SAME outcome through its latch instead of continuing on its own inf / NaN)
    assert close_vec(r["x"], r1["x"], 1e-8) and close_vec(r["y"], r1["y"], 1e-8)
    # with the proximal term (the reference's default) the same instance never meets a bad pivot
    monkeypatch.delenv("QPDO_LINSOLVE")
    r2 = solver.solve_problem(p, verbose=0)
    assert r2["stats"]["band_fallbacks"] == 0 and r2["stats"]["linsolve"] == 3 and r2["info"]["status_val"] == 1


def test_pcg_that_cannot_converge_above_the_dense_limit_is_rescued(gpu_required, monkeypatch):
    """round-4 verdict, item 7: above n = 40 000 a PCG solve that could not converge ended qpdo_solve with QPDO_ERROR -- the one place where
    an instance the reference solves (CHOLMOD factors whatever it is given, cholmod_interface.c:8-30) ended in an error.  n = 50 000,
    chain-structured, forced through PCG with an iteration cap its late Newton systems cannot meet (Jacobi-preconditioned condition ~ 4 /
    sigma): (i) with the rescues switched off: -99 as before; (ii) default: the band direct solver takes over at the first failed pass
    (pcg_rescues = 1, kind bit 0) and the outcome equals the band solver's from the start; (iii) a non-banded instance under a cap its
    accelerated solves cannot meet on some passes: the plain-Jacobi retry (kind bit 1) -- rescued, or an error that names both attempts."""
    p = problems.banded_qp(11, 50_000)
    monkeypatch.delenv("QPDO_LINSOLVE", raising=False)
    ref = solver.solve_problem(p, verbose=0)
    assert ref["stats"]["linsolve"] == 3 and ref["info"]["status_val"] == 1
    monkeypatch.setenv("QPDO_LINSOLVE", "pcg")
    monkeypatch.setenv("QPDO_PCG_MAXIT", "400")
    monkeypatch.setenv("QPDO_PCG_DENSE_FALLBACK", "0")
    r0 = solver.solve_problem(p, verbose=0)
    assert r0["info"]["status_val"] == -99 and "did not converge" in solver.lib().qpdo_amd_last_error().decode()
    monkeypatch.delenv("QPDO_PCG_DENSE_FALLBACK")
    r = solver.solve_problem(p, verbose=0)
    st = r["stats"]
    assert st["pcg_rescues"] == 1 and (st["pcg_rescue_kinds"] & 1) and st["linsolve"] == 3, st
    assert (r["info"]["status_val"], r["info"]["iterations"], r["info"]["oterations"]) == \
           (ref["info"]["status_val"], ref["info"]["iterations"], ref["info"]["oterations"])
    assert close_vec(r["x"], ref["x"], 1e-8) and close_vec(r["y"], ref["y"], 1e-8)
    # (iii) a general sparse K above the dense limit: forced PCG under a cap that its accelerated solves (Schur mode, deflation) miss on
    # some passes -- those passes are solved again by plain Jacobi-PCG with four times the cap
    p2 = problems.random_qp(61, 41000, 9000, 0.0005, 0)
    monkeypatch.delenv("QPDO_PCG_MAXIT")
    full = solver.solve_problem(p2, verbose=0)
    cap = max(t["lin_iters"] for t in full["trace"]) // 2          # (a cap near the maximum is met to 1e-8 and soft-accepted: no failure to rescue)
    monkeypatch.setenv("QPDO_PCG_MAXIT", str(cap))
    r3 = solver.solve_problem(p2, verbose=0)
    assert r3["stats"]["pcg_rescues"] >= 1 and (r3["stats"]["pcg_rescue_kinds"] & 2), r3["stats"]
    if r3["info"]["status_val"] == 1:
        assert r3["info"]["iterations"] == full["info"]["iterations"] and close_vec(r3["x"], full["x"], 1e-7)
    else:
        assert r3["info"]["status_val"] == -99 and "plain-Jacobi retry" in solver.lib().qpdo_amd_last_error().decode()
