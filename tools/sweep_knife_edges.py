"""The sweep instances whose runs end at max_iter with per-pass integers that differ from the oracle's (tests/test_gpu_sweep.py): for each,
the first differing pass, the counts on both sides, the smallest margins min(|w - l|, |u - w|) / max(1, |w|) in the oracle's state
entering that pass, and how far the device's iterate is from the oracle's by then.  usage: sweep_knife_edges.py [lo hi]  (default 0 720)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import device_active_count_consistent, first_integer_mismatch, knife_edge_margins, same_trace_counts
from oracle import binding as ob
from qpdo_amd import solver
from test_gpu_sweep import _instance
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 720)
worst = 0.0
for mode in ("dense", "pcg"):
    os.environ["QPDO_LINSOLVE"] = mode
    for i in range(lo, hi):
        p, st = _instance(i)
        o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve(); to = o.trace(); oi = dict(ro["info"]); o.close()
        if oi["status_val"] != -5:
            continue
        r = solver.solve_problem(p, verbose=0, **st)
        if same_trace_counts(r["trace"], to):
            continue
        k = first_integer_mismatch(r["trace"], to)
        g, t = r["trace"][k], to[k]
        marg = np.sort(knife_edge_margins(p, st, k))[:4]
        # the device's iterate entering pass k against the oracle's
        o2 = ob.OracleSolver(p, ob.default_settings(**dict(st, max_iter=k))); r2o = o2.solve(); o2.close()
        r2 = solver.solve_problem(p, verbose=0, **dict(st, max_iter=k))
        dev = max(float(np.abs(r2["x"] - r2o["x"]).max()) / max(1.0, float(np.abs(r2o["x"]).max())), float(np.abs(r2["y"] - r2o["y"]).max()) / max(1.0, float(np.abs(r2o["y"]).max())))
        cons = device_active_count_consistent(p, st, k) if int(g["kind"]) == 0 else None
        need = max(1, abs(int(g["n_active"]) - int(t["n_active"])))
        worst = max(worst, float(marg[need - 1]))
        print("%s #%d n=%d m=%d %s: first mismatch at pass %d of %d (kind %d/%d, n_active %d/%d, enter %d/%d, leave %d/%d); smallest margins %s; relative |iterate - oracle's| entering that pass %.2e; device count follows from its own w: %s"
              % (mode, i, p["n"], p["m"], {k_: v for k_, v in st.items() if k_ != "max_iter"}, k, len(to), g["kind"], t["kind"], g["n_active"], t["n_active"], g["n_enter"], t["n_enter"],
                 g["n_leave"], t["n_leave"], ["%.2e" % m for m in marg], dev, cons), flush=True)
print("largest margin that had to be accepted: %.2e" % worst)
