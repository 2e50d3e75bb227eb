"""Writes tests/golden/ext_selfcheck_*.mat: instances in the MATLAB exchange format of qpdo_amd.io.save_mat whose stored
answer comes from the CPU ORACLE (oracle/qpdo_oracle.c), labelled as such in ref.source.  They keep
tests/test_gpu_ext.py from being vacuous while no MATLAB user of the reference has contributed a genuine-CHOLMOD file
(tools/reference_fixture.m produces those: same format, ref.source = 'aldma/qpdo reference ...').

Run (build container):  python tests/golden/make_ext_selfcheck.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402
from qpdo_amd import io, problems         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = "oracle/qpdo_oracle.c (CPU restatement of the reference; NOT a CHOLMOD build)"


def settings_dict(s):
    return {k: getattr(s, k) for k in io.SETTING_NAMES}


def main():
    cases = {
        "c1": (problems.config_qp("C1"), dict(max_iter=200), None),
        "c1b_warm": (problems.config_qp("C1b"), dict(), "warm"),
        "kat_primal_infeasible": (problems.infeasibility_kat("primal_infeasible"), dict(max_iter=100), None),
        "kat_dual_infeasible": (problems.infeasibility_kat("dual_infeasible"), dict(max_iter=100), None),
        "eq_rows": (problems.random_qp(4242, 80, 150, 0.2, 30), dict(), None),
    }
    for name, (p, st, warm) in cases.items():
        s = ob.default_settings(**st)
        o = ob.OracleSolver(p, s)
        w = None
        if warm:
            rng = np.random.default_rng(9)
            w = (0.1 * rng.standard_normal(p["n"]), 0.1 * rng.standard_normal(p["m"]))
            o.warm_start(*w)
        r = o.solve()
        o.close()
        path = os.path.join(HERE, "ext_selfcheck_%s.mat" % name)
        io.save_mat(path, p, settings=settings_dict(s), result=r, warm=w, source=SOURCE)
        print("wrote %s: status %d, %d passes (%d outer)" % (path, r["info"]["status_val"], r["info"]["iterations"], r["info"]["oterations"]))


if __name__ == "__main__":
    main()
