"""Per-field maximum deviation of the dense path's trace from the oracle's, one-launch factorization (default) beside the multi-launch one
(QPDO_DENSE_MID=0), on the instances of test_one_launch_factorization_matches_the_multi_launch_one_and_the_oracle and a few more."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["QPDO_LINSOLVE"] = "dense"; os.environ["QPDO_DENSE_LOWRANK"] = "0"
from oracle import binding as ob
from qpdo_amd import problems, solver
import importlib.util
spec = importlib.util.spec_from_file_location("td", os.path.join(ROOT, "tools", "trace_dev.py"))
def dev(tg, to):
    out = {}
    if len(tg) != len(to): return dict(len=(len(tg), len(to)))
    for k, (g, r) in enumerate(zip(tg, to)):
        for f in ("kind", "n_active", "n_enter", "n_leave", "factor_branch"):
            if int(g[f]) != int(r[f]): out.setdefault("int_mismatch", []).append((k, f))
        if int(r["kind"]) == 0 and max(r["res_prim_in"], r["res_dual_in"]) > 1e-13:
            e = abs(g["tau"] - r["tau"]) / max(1.0, abs(r["tau"]))
            if e > out.get("tau", (0, 0))[0]: out["tau"] = (float(e), k)
        for f in ("res_prim", "res_dual", "res_prim_in", "res_dual_in"):
            ea = abs(g[f] - r[f])
            if ea > out.get(f, (0, 0))[0]: out[f] = (float(ea), k)
    return out
for seed, n, m, dens in ((5, 200, 100, 0.1), (6, 64, 150, 0.2), (7, 360, 500, 0.05), (77, 1500, 2600, 0.01), (31, 300, 600, 0.1), (32, 500, 1000, 0.05), (33, 1000, 2000, 0.02), (41, 600, 1200, 0.02), (43, 1000, 700, 0.03), (51, 2400, 3000, 0.01)):
    p = problems.random_qp(seed, n, m, dens, min(50, m // 4))
    o = ob.OracleSolver(p, ob.default_settings()); ro = o.solve(); to = o.trace(); o.close()
    for mid in ("0", "1"):
        os.environ["QPDO_DENSE_MID"] = mid
        r = solver.solve_problem(p, verbose=0)
        print("n=%d m=%d mid=%s its %d/%d" % (n, m, mid, r["info"]["iterations"], ro["info"]["iterations"]), json.dumps(dev(r["trace"], to)), flush=True)
