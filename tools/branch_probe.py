"""Scratch: per-pass branch statistics (rows entering/leaving, refactor flag) of a configuration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
name = sys.argv[1]
p = problems.config_qp(name)
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
t = time.time(); r = s.solve(); print("solve", time.time() - t, r["info"]["status"], r["info"]["iterations"], flush=True)
tr = s.trace()
print("keys", list(tr[0].keys()))
for t_ in tr:
    print({k: (v if not isinstance(v, float) else float("%.3g" % v)) for k, v in t_.items() if k in ("iter", "kind", "n_active", "n_enter", "n_leave", "refactor", "lin_iters", "branch", "sigma")})
