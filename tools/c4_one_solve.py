"""One cold-start C4 solve (for profilers): prints time, passes, CG iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
p = problems.config_qp("C4")
t0 = time.time()
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
print("C4 setup %.3f s (info.setup_time %.3f s)" % (time.time() - t0, s.info()["setup_time"]))
t0 = time.time(); r = s.solve(); dt = time.time() - t0
print("C4 solve %.3f s passes %d status %d cg %d" % (dt, r["info"]["iterations"], r["info"]["status_val"], s.stats()["lin_iters"]))
