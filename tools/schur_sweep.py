"""Scratch: wider randomized sweep of the Schur-complement PCG mode against the oracle (not a test: ~2 min)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QPDO_LINSOLVE"] = "pcg"; os.environ["QPDO_PCG_SCHUR"] = "1"
import numpy as np
from oracle import binding as ob
from qpdo_amd import problems, solver
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    rng = np.random.default_rng(900 + i)
    n = int(rng.integers(400, 800)); m = int(rng.integers(600, 2400)); neq = int(rng.integers(0, 150)) if rng.random() < 0.5 else 0
    p = problems.random_qp(9100 + i, n, m, float(rng.choice([0.02, 0.04, 0.08])), neq)
    st = {}
    r_ = rng.random()
    if r_ < 0.25: st["scaling"] = 0
    elif r_ < 0.4: st["proximal"] = 0
    elif r_ < 0.55: st["eps_abs"] = 1e-8
    elif r_ < 0.7: st["mu_min"] = 1e-10
    st["max_iter"] = 300
    o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve(); oi = dict(ro["info"]); ox, oy = np.array(ro["x"]), np.array(ro["y"]); o.close()
    r = solver.solve_problem(p, verbose=0, **st); gi = r["info"]
    same = (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
    err = 0.0 if oi["status_val"] in (-3, -4, -5) else float(max(np.abs(r["x"] - ox).max() / max(1, np.abs(ox).max()), np.abs(r["y"] - oy).max() / max(1, np.abs(oy).max())))
    flag = "" if same and err <= 1e-7 else "<<<"
    bad += bool(flag)
    print(i, n, m, neq, st, "status", oi["status_val"], gi["status_val"], "it", oi["iterations"], gi["iterations"], "schur", r["stats"]["schur_passes"], "err %.1e" % err, flag, flush=True)
print("bad", bad)
