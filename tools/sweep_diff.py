"""Per-pass trace of one instance of the randomized sweep (tests/test_gpu_sweep.py::_instance): oracle | device, first differences marked.
usage: sweep_diff.py i [dense|pcg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import binding as ob
from qpdo_amd import solver
import test_gpu_sweep as T
i = int(sys.argv[1])
os.environ["QPDO_LINSOLVE"] = sys.argv[2] if len(sys.argv) > 2 else "dense"
p, st = T._instance(i)
o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve(); to = o.trace()
r = solver.solve_problem(p, verbose=0, **st); tg = r["trace"]
print("instance", i, "n", p["n"], "m", p["m"], st, "oracle its", ro["info"]["iterations"], ro["info"]["status_val"], "device its", r["info"]["iterations"], r["info"]["status_val"])
shown = 0
for k in range(max(len(to), len(tg))):
    a = to[k] if k < len(to) else None; b = tg[k] if k < len(tg) else None
    def f(t): return None if t is None else (t["kind"], t["n_active"], t["n_enter"], t["n_leave"], t["factor_branch"], "%.10e" % t["tau"], "%.6e" % t["res_prim"], "%.6e" % t["res_dual"], "%.6e" % t["res_prim_in"], "%.6e" % t["res_dual_in"], "%.3e" % t["eps_in"], "%.3e" % t["sigma"])
    d = a is None or b is None or f(a) != f(b)
    if d or k < 2:
        print(k, f(a), "\n  ", f(b), "<<<" if d else ""); shown += d
        if shown > 12: break
