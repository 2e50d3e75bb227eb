"""Deviation of the default device path from the complete oracle record of C4 (tests/golden/big_C4_full.npz): per-pass tau and norms, final iterate."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import trace_from_npz
from qpdo_amd import problems, solver
z = np.load(os.path.join(ROOT, "tests", "golden", "big_C4_full.npz")); meta = json.loads(str(z["meta"]))
p = problems.config_qp("C4", 0)
r = solver.solve_problem(p, verbose=0)
ref = trace_from_npz(z); got = r["trace"]
print("passes", len(got), len(ref), "status", r["info"]["status_val"], meta["info"]["status_val"])
ints = all(int(g[f]) == int(q[f]) for g, q in zip(got, ref) for f in ("kind", "n_active", "n_enter", "n_leave", "factor_branch"))
dtau = max(abs(g["tau"] - q["tau"]) / max(1.0, abs(q["tau"])) for g, q in zip(got, ref) if int(q["kind"]) == 0)
dn = max(abs(g[f] - q[f]) for g, q in zip(got, ref) for f in ("res_prim", "res_dual", "res_prim_in", "res_dual_in"))
print("integers identical", ints, "max tau dev %.2e" % dtau, "max norm dev (abs) %.2e" % dn)
print("|x-xo|inf %.2e |y-yo|inf %.2e" % (np.abs(r["x"] - z["x"]).max(), np.abs(r["y"] - z["y"]).max()), "objective", r["info"]["objective"], meta["info"]["objective"])
print("KKT", r["info"]["res_prim_norm"], r["info"]["res_dual_norm"], "oracle", meta["info"]["res_prim_norm"], meta["info"]["res_dual_norm"])
print("device lin iters", r["stats"]["lin_iters"], "oracle CG iterations", meta["oracle_lin_iters"], "oracle seconds", meta["oracle_seconds"])
print("per pass (kind, n_active, lin_iters):", [(int(t["kind"]), int(t["n_active"]), int(t["lin_iters"])) for t in got])
