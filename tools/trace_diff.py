"""Scratch: compare the per-pass trace of the device solver with the CPU oracle on one random instance."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qpdo_amd import problems, solver
from oracle import binding as ob
seed, n, m, dens, neq = 29, 200, 400, 0.05, 0
st = dict(reset_newton_iter=3, inner_max_iter=6)
p = problems.random_qp(seed, n, m, dens, neq)
o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve(); to = o.trace()
r = solver.solve_problem(p, verbose=0, **st); tg = r["trace"]
print("oracle its", ro["info"]["iterations"], "gpu its", r["info"]["iterations"], r.get("stats"))
for i in range(max(len(to), len(tg))):
    a = to[i] if i < len(to) else None; b = tg[i] if i < len(tg) else None
    def f(t): return None if t is None else (t["kind"], t["n_active"], t["n_enter"], t["n_leave"], t["factor_branch"], "%.6e" % t["tau"], "%.6e" % t["res_prim_in"], "%.6e" % t["res_dual_in"], "%.3e" % t["eps_in"])
    print(i, f(a), f(b), "" if a is None or b is None or f(a) == f(b) else "<<<")
