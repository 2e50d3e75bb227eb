"""C2 (n=1e4, m=2e4): per-pass cost of the PCG solver against the dense factorization, and the hybrid (QPDO_HYBRID=<budget>: PCG while a pass
stays under the iteration budget, the dense factor from the first pass that exceeds it): time and deviation from the oracle fixture."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import trace_from_npz
from qpdo_amd import problems, solver
z = np.load(os.path.join(ROOT, "tests", "golden", "big_C2.npz")); to = trace_from_npz(z)
p = problems.config_qp("C2")
def run(label, env):
    for k in list(os.environ):
        if k.startswith("QPDO_") and k != "QPDO_DEVICE": del os.environ[k]
    os.environ.update(env)
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    s.solve()
    ts = []
    for _ in range(3):
        t = time.time(); r = s.solve(); solver.lib().qpdo_amd_sync(s._w); ts.append(time.time() - t)
    st, tr = s.stats(), s.trace()
    ints = len(tr) == len(to) and all(int(a[f]) == int(b[f]) for a, b in zip(tr, to) for f in ("kind", "n_active", "n_enter", "n_leave", "factor_branch"))
    dtau = max((abs(a["tau"] - b["tau"]) / max(1.0, abs(b["tau"])) for a, b in zip(tr, to) if int(b["kind"]) == 0 and max(b["res_prim_in"], b["res_dual_in"]) > 1e-13), default=-1) if ints else -1
    dn = max((abs(a[f] - b[f]) / (1e-8 * abs(b[f]) + 1e-9) for a, b in zip(tr, to) for f in ("res_prim", "res_dual", "res_prim_in", "res_dual_in")), default=-1) if ints else -1
    print(label, "time %.3f s (min of 3: %.3f)" % (ts[-1], min(ts)), "passes", r["info"]["iterations"], "factor_count", st["factor_count"], "lowrank", st["lowrank_solves"], "lin_iters", st["lin_iters"],
          "| ints same", ints, "max tau dev %.2e (tol 1e-8)  max norm dev / tol %.2f" % (dtau, dn), "|x-xo| %.2e" % np.abs(r["x"] - z["x"]).max(), flush=True)
    if env.get("QPDO_LINSOLVE") == "pcg":
        print("   lin_iters per Newton pass:", [int(t_["lin_iters"]) for t_ in tr if t_["kind"] == 0])
    s.delete()
run("dense (default)", {})
run("pcg", {"QPDO_LINSOLVE": "pcg"})
for b in (200, 350, 450, 600, 800):
    run("hybrid budget %d" % b, {"QPDO_HYBRID": str(b)})
