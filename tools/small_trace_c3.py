import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
p = problems.config_qp("C3")
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
for rep in range(3):
    t = time.time(); r = s.solve(); dt = time.time() - t
print("C3 passes", r["info"]["iterations"], "solve %.2f ms" % (dt * 1e3), [t["kind"] for t in r["trace"]], [t["factor_branch"] for t in r["trace"]])
s.delete()
