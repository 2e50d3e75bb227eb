"""Per-variant maximum deviation of the band solver's (and the dense solver's) per-pass norms / step lengths from the oracle on one banded
workspace: cold solve and a warm-started re-solve after update_q / update_bounds, for default settings, scaling = 0 and proximal = 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from qpdo_amd import problems, solver
p = problems.banded_random_qp(21, 2300, 9)
for mode in ("band", "dense"):
    os.environ["QPDO_LINSOLVE"] = mode
    for st in (dict(), dict(scaling=0), dict(proximal=0)):
        o = ob.OracleSolver(p, ob.default_settings(**st))
        s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0, **st)
        for phase in ("cold", "resolve"):
            if phase == "resolve":
                o.warm_start(ro["x"], ro["y"]); s.warm_start(ro["x"], ro["y"])
                q2 = p["q"] * 1.1 + 0.05; o.update_q(q2); s.update_q(q2)
                l2, u2 = p["l"] - 0.05, p["u"] + 0.02; o.update_bounds(l2, u2); s.update_bounds(l2, u2)
            ro, rg = o.solve(), s.solve()
            tg, to = s.trace(), o.trace()
            same = len(tg) == len(to) and all(int(a[f]) == int(b[f]) for a, b in zip(tg, to) for f in ("kind", "n_active", "n_enter", "n_leave", "factor_branch"))
            dn = max((abs(a[f] - b[f]) for a, b in zip(tg, to) for f in ("res_prim", "res_dual", "res_prim_in", "res_dual_in")), default=0) if len(tg) == len(to) else -1
            dt = max((abs(a["tau"] - b["tau"]) for a, b in zip(tg, to) if int(b["kind"]) == 0 and max(b["res_prim_in"], b["res_dual_in"]) > 1e-13), default=0) if len(tg) == len(to) else -1
            dx = np.abs(rg["x"] - ro["x"]).max() if ro["info"]["status_val"] == 1 else float("nan")
            print(mode, st, phase, "passes", rg["info"]["iterations"], ro["info"]["iterations"], "ints same", same, "max norm dev %.2e  max tau dev %.2e  |x-xo| %.2e" % (dn, dt, dx), flush=True)
        s.delete(); o.close()
