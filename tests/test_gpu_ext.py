"""Externally produced fixtures: tests/golden/ext_*.mat in the MATLAB exchange format of qpdo_amd.io (SURVEY section
8(f)4).  A MATLAB user of the reference writes them with tools/reference_fixture.m (the reference's class + mex over
CHOLMOD: interfaces/mex/qpdo.m:50-233, qpdo_mex.c:227-281); the ext_selfcheck_* files committed today come from the CPU
oracle and say so in ref.source.  Every file present is solved through the C-ABI with its stored settings / warm start
and must reproduce the stored answer: status, iterations, oterations IDENTICAL; x, y within ITERATE_RTOL; certificates
within 1e-6 relative (as for the golden KATs); objective within 1e-9 relative."""
import glob
import os

import numpy as np
import pytest

from helpers import ITERATE_RTOL, close_vec
from qpdo_amd import io, problems, solver

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXT = sorted(glob.glob(os.path.join(HERE, "golden", "ext_*.mat")))


def solve_loaded(prob, settings, warm):
    kw = dict(settings or {})
    kw["verbose"] = 0
    s = solver.QPDO().setup(prob["Q"], prob["q"], prob["A"], prob["l"], prob["u"], Qstype=prob["Qstype"], c=prob["c"], **kw)
    if warm is not None:
        s.warm_start(*warm)
    r = s.solve()
    s.delete()
    return r


@pytest.mark.parametrize("path", EXT, ids=[os.path.basename(p)[4:-4] for p in EXT])
def test_external_fixture(path, gpu_required):
    prob, settings, ref, warm = io.load_mat(path)
    assert ref is not None, "fixture without a stored reference answer"
    r = solve_loaded(prob, settings, warm)
    i = r["info"]
    assert (i["status_val"], i["iterations"], i["oterations"]) == (ref["status_val"], ref["iterations"], ref["oterations"]), (ref["source"], i)
    assert close_vec(r["x"], ref["x"], ITERATE_RTOL) and close_vec(r["y"], ref["y"], ITERATE_RTOL), ref["source"]
    if ref["status_val"] == -3:
        c, c0 = r["prim_inf_cert"], ref["prim_inf_cert"]
        assert np.abs(c - c0).max() <= 1e-6 * max(1.0, np.abs(c0).max())
    if ref["status_val"] == -4:
        c, c0 = r["dual_inf_cert"], ref["dual_inf_cert"]
        assert np.abs(c - c0).max() <= 1e-6 * max(1.0, np.abs(c0).max())
    if ref["status_val"] == 1 and "objective" in ref:
        assert abs(i["objective"] - ref["objective"]) <= 1e-9 * max(1.0, abs(ref["objective"]))


@pytest.mark.parametrize("fmt", ["npz", "mat"])
def test_saved_instance_round_trips_through_io_and_solves(fmt, gpu_required, tmp_path):
    """C1 written to disk, read back and solved: the loaded instance must give the very result of the in-memory one
    (same bits: both reach qpdo_setup as the same arrays), and a file written WITH that result must pass the fixture
    check above -- the full loop a MATLAB user of the reference would close from the other side."""
    p = problems.config_qp("C1")
    st = {k: getattr(solver.default_settings(verbose=0, max_iter=200), k) for k in io.SETTING_NAMES}
    r0 = solver.solve_problem(p, verbose=0, max_iter=200)
    path = tmp_path / ("c1." + fmt)
    if fmt == "npz":
        io.save_problem(path, p, st, result=r0)
        q, s2, res = io.load_problem(path)
        ref = dict(x=res["x"], y=res["y"], **res["info"])
        warm = None
    else:
        io.save_mat(path, p, st, result=r0)
        q, s2, ref, warm = io.load_mat(path)
    assert s2["max_iter"] == 200
    r1 = solve_loaded(q, s2, warm)
    assert (r1["info"]["status_val"], r1["info"]["iterations"], r1["info"]["oterations"]) == (ref["status_val"], ref["iterations"], ref["oterations"])
    assert r1["x"].tobytes() == r0["x"].tobytes() and r1["y"].tobytes() == r0["y"].tobytes()
    assert np.array_equal(ref["x"], r0["x"]) and np.array_equal(ref["y"], r0["y"])
