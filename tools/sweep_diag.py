"""Scratch: per-instance detail of the randomized parity sweep (tests/test_gpu_sweep.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_sweep import _instance
from oracle import binding as ob
from qpdo_amd import solver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for i in range(N):
    p, st = _instance(i)
    o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve(); oi = dict(ro["info"]); ox, oy = np.array(ro["x"]), np.array(ro["y"]); o.close()
    r = solver.solve_problem(p, verbose=0, **st)
    gi = r["info"]
    def md(a, b):
        a, b = np.asarray(a, float), np.asarray(b, float)
        if np.isnan(b).all() or b.size == 0: return 0.0
        return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    flag = "" if (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"]) else "<<< COUNTS"
    dx, dy = md(r["x"], ox), md(r["y"], oy)
    if not flag and oi["status_val"] not in (-3, -4) and max(dx, dy) > 1e-9: flag = "<<< ITERATES"
    print(i, p["n"], p["m"], st, "status", oi["status_val"], gi["status_val"], "it", oi["iterations"], gi["iterations"], "ot", oi["oterations"], gi["oterations"], "dx %.2e dy %.2e" % (dx, dy), flag, flush=True)
