"""Pass counts / traces of the production-size fixtures and C4 timing against the factor of the absolute stopping rule of
the Newton solves (QPDO_PCG_ABS x eps_abs; 0 = relative rule only)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import golden_problem, trace_from_npz, same_trace_counts
from qpdo_amd import problems, solver
gd = os.path.join(ROOT, "tests", "golden")
cases = []
for f in ("big_pcg_13k_eq.npz", "big_schur_30k.npz"):
    z = np.load(os.path.join(gd, f)); meta = json.loads(str(z["meta"]))
    cases.append((f, golden_problem(meta["spec"]), meta, trace_from_npz(z), z["x"], z["y"]))
KNOB = os.environ.get("PROBE_KNOB", "QPDO_PCG_ABS")
VALS = [float(v) for v in os.environ.get("PROBE_VALS", "0,1e-5,1e-4,1e-3").split(",")]
for tol in VALS:
    os.environ[KNOB] = "%g" % tol
    for f, p, meta, tr, x, y in cases:
        t0 = time.time(); r = solver.solve_problem(p, verbose=0); dt = time.time() - t0
        oi = meta["info"]
        same = (r["info"]["status_val"], r["info"]["iterations"], r["info"]["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
        print("tol %.0e %-20s %.2fs its %d/%d counts %s trace %s cg %d ex %.1e ey %.1e" % (
            tol, f, dt, r["info"]["iterations"], oi["iterations"], same, same_trace_counts(r["trace"], tr), r["stats"]["lin_iters"],
            np.abs(r["x"] - x).max() / max(1, np.abs(x).max()), np.abs(r["y"] - y).max() / max(1, np.abs(y).max())), flush=True)
p = problems.config_qp("C4")
for tol in VALS:
    os.environ[KNOB] = "%g" % tol
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    t0 = time.time(); r = s.solve(); dt = time.time() - t0
    rp, rd = problems.kkt_residuals(p, r["x"], r["y"])
    print("C4 abs %.0e %.2fs its %d oter %d cg %d kkt %.6e %.6e tr %s" % (tol, dt, r["info"]["iterations"], r["info"]["oterations"], s.stats()["lin_iters"], rp, rd,
          [t["n_active"] for t in s.trace()][-8:]), flush=True)
    s.delete()
