"""The fused batch kernel alone over instances [lo, hi) of the randomized sweep of tests/test_gpu_sweep.py (one launch), every item against
the oracle bit for bit: status, counts, x, y, objective, residual norms.  usage: sweep_fused.py lo hi [max_iter]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import binding as ob
from qpdo_amd import solver
import test_gpu_sweep as T
lo, hi = int(sys.argv[1]), int(sys.argv[2]); mi = int(sys.argv[3]) if len(sys.argv) > 3 else 300
t0 = time.time()
probs = [T._instance(i)[0] for i in range(lo, hi)]
res, failed = solver.solve_batch(probs, verbose=0, max_iter=mi)
print("batch done %.1f s" % (time.time() - t0), flush=True)
ob.set_threads(1)
bad = []
for i, (p, r) in zip(range(lo, hi), zip(probs, res)):
    o = ob.OracleSolver(p, ob.default_settings(max_iter=mi)); ro = o.solve(); oi = dict(ro["info"]); o.close()
    gi = r["info"]
    ok = (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
    if ok and oi["status_val"] not in (-3, -4):
        ok = np.array_equal(r["x"], ro["x"]) and np.array_equal(r["y"], ro["y"]) and gi["objective"] == oi["objective"] \
            and gi["res_prim_norm"] == oi["res_prim_norm"] and gi["res_dual_norm"] == oi["res_dual_norm"]
    if not ok: bad.append((i, p["n"], p["m"], oi["status_val"], gi["status_val"], oi["iterations"], gi["iterations"]))
print("fused batch: instances %d failed %d mismatches %d %s  %.1f s" % (hi - lo, failed, len(bad), bad[:10], time.time() - t0), flush=True)
