"""Diagnostic: per-field maximum deviation of the device trace from the oracle trace, per case and linear solver.
usage: trace_dev.py [big]      (big: also the tests/golden/big_*.npz fixtures with the default solver selection)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import golden_problem, load_golden, trace_from_npz
from oracle import binding as ob
from qpdo_amd import problems, solver


def dev(tg, to):
    out = {}
    if len(tg) != len(to):
        return dict(len=(len(tg), len(to)))
    for k, (g, r) in enumerate(zip(tg, to)):
        for f in ("kind", "n_active", "n_enter", "n_leave", "factor_branch"):
            if int(g[f]) != int(r[f]):
                out.setdefault("int_mismatch", []).append((k, f, int(g[f]), int(r[f])))
        if int(r["kind"]) == 0:
            e = abs(g["tau"] - r["tau"]) / max(1.0, abs(r["tau"]))
            if e > out.get("tau", (0, 0))[0]:
                out["tau"] = (float(e), k)
        for f in ("res_prim", "res_dual", "res_prim_in", "res_dual_in"):
            ea = abs(g[f] - r[f]); er = ea / max(abs(r[f]), 1e-300)
            if ea > out.get(f + "_abs", (0, 0))[0]:
                out[f + "_abs"] = (float(ea), k)
            if min(er, ea / 1e-10) > out.get(f + "_mix", (0, 0))[0]:     # smallest rtol s.t. ea <= rtol*|r| + 1e-10 roughly
                out[f + "_mix"] = (float(min(er, ea / 1e-10)), k)
    return out


cases = []
G = load_golden()
for name in sorted(G):
    cases.append((name, golden_problem(G[name]["spec"]), G[name]["settings"]))
for seed, n, m, dens, neq, st in [(21, 40, 60, 0.2, 0, {}), (22, 150, 300, 0.05, 0, {}), (23, 150, 300, 0.05, 50, {}),
                                  (24, 300, 200, 0.03, 0, dict(scaling=0)), (25, 300, 200, 0.03, 0, dict(proximal=0)),
                                  (28, 500, 1000, 0.02, 0, dict(eps_abs=1e-8)), (29, 200, 400, 0.05, 0, dict(reset_newton_iter=3, inner_max_iter=6))]:
    cases.append(("rand%d" % seed, problems.random_qp(seed, n, m, dens, neq), st))
for name, p, st in cases:
    o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve(); to = o.trace(); o.close()
    for ls in ("dense", "pcg"):
        os.environ["QPDO_LINSOLVE"] = ls
        r = solver.solve_problem(p, verbose=0, **st)
        d = dev(r["trace"], to)
        print(name, ls, "its", r["info"]["iterations"], ro["info"]["iterations"], json.dumps(d), flush=True)
        if name.startswith("kat") and ls == "pcg":
            for k, (g, t) in enumerate(zip(r["trace"], to)):
                print("   ", k, g["kind"], t["kind"], g["n_active"], t["n_active"], g["factor_branch"], t["factor_branch"], g["lin_iters"],
                      "tau %.12g %.12g" % (g["tau"], t["tau"]), "rdi %.6e %.6e" % (g["res_dual_in"], t["res_dual_in"]), "rpi %.6e %.6e" % (g["res_prim_in"], t["res_prim_in"]))
os.environ.pop("QPDO_LINSOLVE", None)
if len(sys.argv) > 1 and sys.argv[1] == "big":
    gd = os.path.join(ROOT, "tests", "golden")
    for f in sorted(os.listdir(gd)):
        if f.startswith("big_") and f.endswith(".npz"):
            z = np.load(os.path.join(gd, f)); meta = json.loads(str(z["meta"]))
            p = golden_problem(meta["spec"])
            import time
            t0 = time.time(); r = solver.solve_problem(p, verbose=0, **meta["settings"]); dt = time.time() - t0
            d = dev(r["trace"], trace_from_npz(z))
            ex = float(np.abs(r["x"] - z["x"]).max() / max(1, np.abs(z["x"]).max())); ey = float(np.abs(r["y"] - z["y"]).max() / max(1, np.abs(z["y"]).max()))
            print(f, "linsolve", r["stats"]["linsolve"], "its", r["info"]["iterations"], meta["info"]["iterations"], "status", r["info"]["status_val"], meta["info"]["status_val"],
                  "ex %.2e ey %.2e" % (ex, ey), "%.1fs" % dt, json.dumps(d), {k: r["stats"][k] for k in ("factor_count", "lowrank_solves", "schur_passes", "lin_iters")}, flush=True)
