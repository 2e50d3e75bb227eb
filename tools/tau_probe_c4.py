"""C4 time-to-eps, pass counts and inner iterations against the base of the Schur-mode inner tolerance (QPDO_SCHUR_TAU), each value on a
fresh workspace of the same instance, plus agreement of the per-pass integers with the complete oracle record."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import trace_from_npz, same_trace_counts
from qpdo_amd import problems, solver
z = np.load(os.path.join(ROOT, "tests", "golden", "big_C4_full.npz")); ref = trace_from_npz(z)
p = problems.config_qp("C4")
for tau in [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2e-6,3e-6,4e-6,6e-6").split(",")]:
    os.environ["QPDO_SCHUR_TAU"] = "%g" % tau
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    t0 = time.time(); r = s.solve(); dt = time.time() - t0
    print("tau0 %.1e: %.3f s, passes %d (%d outer), status %d, lin iters %d, trace integers %s, |x-xo| %.1e" % (
        tau, dt, r["info"]["iterations"], r["info"]["oterations"], r["info"]["status_val"], s.stats()["lin_iters"],
        same_trace_counts(s.trace(), ref), np.abs(r["x"] - z["x"]).max()), flush=True)
    s.delete()
