"""The reference's MPC use case (src/qpdo.c:217-299,522-586): one workspace, repeated qpdo_update_q + qpdo_warm_start + qpdo_solve
with a slightly moved linear term.  Prints the latency of each call on a C3-size instance (n=120, m=360) and on C1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qpdo_amd import problems, solver
for name in ("C3", "C1"):
    p = problems.config_qp(name)
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    s.solve()
    t = time.time(); r = s.solve(); cold = time.time() - t
    cold_passes = r["info"]["iterations"]
    rng = np.random.default_rng(0)
    rows = []
    for k in range(12):
        q = p["q"] + 1e-2 * rng.standard_normal(p["n"])
        t0 = time.time(); s.update_q(q); t1 = time.time(); s.warm_start(r["x"], r["y"]); t2 = time.time(); r = s.solve(); t3 = time.time()
        rows.append((t1 - t0, t2 - t1, t3 - t2, r["info"]["iterations"], r["info"]["status_val"]))
    a = np.array(rows[2:])
    print("%s: cold solve %.2f ms (%d passes); re-solve medians: update_q %.3f ms, warm_start %.3f ms, solve %.3f ms, passes %s, status %s" % (
        name, cold * 1e3, cold_passes, *(np.median(a[:, :3], axis=0) * 1e3), sorted(set(int(v) for v in a[:, 3])), sorted(set(int(v) for v in a[:, 4]))), flush=True)
    s.delete()
