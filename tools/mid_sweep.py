"""One-off parity sweep over mid-size instances (n 200..4000) through the DEFAULT solver selection and the forced PCG path, against the
oracle run on the box: counts, per-pass integer trace, iterates.  usage: mid_sweep.py count [seed0 [nmin nmax]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import close_vec, same_trace_counts
from oracle import binding as ob
from qpdo_amd import problems, solver
cnt = int(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nmin, nmax = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (200, 4000)
for k in list(os.environ):
    if k.startswith("QPDO_"): del os.environ[k]
bad = []; t0 = time.time()
for i in range(seed0, seed0 + cnt):
    rng = np.random.default_rng(50000 + i)
    n = int(rng.integers(nmin, nmax)); m = int(rng.integers(n // 2, 3 * n)); dens = float(rng.choice([0.002, 0.005, 0.01, 0.03])) if n > 800 else float(rng.choice([0.02, 0.05, 0.1]))
    neq = int(rng.integers(0, min(n, m) // 3)) if rng.random() < 0.4 else 0
    p = problems.random_qp(60000 + i, n, m, dens, neq)
    o = ob.OracleSolver(p, ob.default_settings()); ro = o.solve(); oi = dict(ro["info"]); ox, oy = np.array(ro["x"]), np.array(ro["y"]); to = o.trace(); o.close()
    line = "i=%d n=%d m=%d dens=%.3f neq=%d oracle its %d st %d |" % (i, n, m, dens, neq, oi["iterations"], oi["status_val"])
    for mode in ("auto", "pcg"):
        if mode == "pcg": os.environ["QPDO_LINSOLVE"] = "pcg"
        else: os.environ.pop("QPDO_LINSOLVE", None)
        r = solver.solve_problem(p, verbose=0); gi = r["info"]
        same = (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
        if same and oi["status_val"] != -5: same = same_trace_counts(r["trace"], to)
        if same and oi["status_val"] == 1: same = close_vec(r["x"], ox, 1e-7) and close_vec(r["y"], oy, 1e-7)
        line += " %s its %d st %d %s (%.2fs, lr %d, schur %d)" % (mode, gi["iterations"], gi["status_val"], "ok" if same else "MISMATCH", gi["solve_time"], r["stats"]["lowrank_solves"], r["stats"].get("schur_passes", 0))
        if not same: bad.append((i, mode))
    os.environ.pop("QPDO_LINSOLVE", None)
    print(line, "t=%.0fs" % (time.time() - t0), flush=True)
print("mismatches:", bad)
