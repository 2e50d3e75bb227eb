"""One-off wider run of the randomized parity sweep of tests/test_gpu_sweep.py: instances [lo, hi) through the dense path, the PCG path and
the default path (small workspaces: the fused route of qpdo_solve, compared bit for bit) and the fused batch kernel, each against the
oracle (same acceptance rules as the tests).  usage: sweep_more.py lo hi"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import close_vec, same_trace_counts
from oracle import binding as ob
from qpdo_amd import solver
import test_gpu_sweep as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
orc = {}
for i in range(lo, hi):
    p, st = T._instance(i)
    o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve()
    orc[i] = (dict(ro["info"]), np.array(ro["x"]), np.array(ro["y"]), o.trace()); o.close()
print("oracle done %.1f s" % (time.time() - t0), flush=True)
for mode in ("dense", "pcg"):
    os.environ["QPDO_LINSOLVE"] = mode
    bad = []
    for i in range(lo, hi):
        p, st = T._instance(i)
        oi, ox, oy, to = orc[i]
        r = solver.solve_problem(p, verbose=0, **st); gi = r["info"]
        same = (gi["status_val"] == oi["status_val"] and gi["iterations"] == oi["iterations"] and gi["oterations"] == oi["oterations"])
        if oi["status_val"] != -5: same = same and same_trace_counts(r["trace"], to)
        if same and oi["status_val"] not in (-3, -4, -5): same = close_vec(r["x"], ox, 1e-7) and close_vec(r["y"], oy, 1e-7)
        if not same: bad.append((i, p["n"], p["m"], st, oi["status_val"], gi["status_val"], oi["iterations"], gi["iterations"]))
    print(mode, "instances", hi - lo, "mismatches", len(bad), bad[:10], "%.1f s" % (time.time() - t0), flush=True)
os.environ.pop("QPDO_LINSOLVE", None)
# the DEFAULT path (round 4: n <= 160 runs as one launch of the fused kernel on the workspace's arrays): the oracle's bits, per-instance settings
bad = []
for i in range(lo, hi):
    p, st = T._instance(i)
    oi, ox, oy, to = orc[i]
    r = solver.solve_problem(p, verbose=0, **st); gi = r["info"]
    same = (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"]) and r["stats"]["linsolve"] == 2
    same = same and len(r["trace"]) == len(to) and all(g[f] == t_[f] for g, t_ in zip(r["trace"], to) for f in ("kind", "n_active", "n_enter", "n_leave", "factor_branch", "res_prim", "res_dual", "res_prim_in", "res_dual_in", "sigma", "eps_in"))
    if same and oi["status_val"] not in (-3, -4): same = np.array_equal(r["x"], ox) and np.array_equal(r["y"], oy) and gi["objective"] == oi["objective"]
    if not same: bad.append((i, p["n"], p["m"], st, oi["status_val"], gi["status_val"], oi["iterations"], gi["iterations"]))
print("default path (fused route, bit for bit)", "instances", hi - lo, "mismatches", len(bad), bad[:10], "%.1f s" % (time.time() - t0), flush=True)
probs = [T._instance(i)[0] for i in range(lo, hi)]
res, failed = solver.solve_batch(probs, verbose=0, max_iter=300)
bad = []
for i, (p, r) in zip(range(lo, hi), zip(probs, res)):
    o = ob.OracleSolver(p, ob.default_settings(max_iter=300)); ro = o.solve(); oi, ox, oy = dict(ro["info"]), np.array(ro["x"]), np.array(ro["y"]); o.close()
    gi = r["info"]
    ok = (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
    if ok and oi["status_val"] not in (-3, -4):
        ok = np.array_equal(r["x"], ox) and np.array_equal(r["y"], oy) and gi["objective"] == oi["objective"]
    if not ok: bad.append((i, p["n"], p["m"], oi["status_val"], gi["status_val"], oi["iterations"], gi["iterations"]))
print("fused batch: failed", failed, "mismatches", len(bad), bad[:10], "%.1f s" % (time.time() - t0), flush=True)
