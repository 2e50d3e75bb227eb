"""One mid-size instance for the kernel timeline (tools/mid_trace.sh): three cold solves through the default path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
which = sys.argv[1] if len(sys.argv) > 1 else "C1"
p = problems.config_qp("C1") if which == "C1" else {"n500": problems.random_qp(32, 500, 1000, 0.05), "n1000": problems.random_qp(33, 1000, 2000, 0.02), "n300": problems.random_qp(31, 300, 600, 0.1), "n2000": problems.random_qp(34, 2000, 4000, 0.01)}[which]
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
for rep in range(3):
    t = time.time(); r = s.solve(); dt = time.time() - t
print(which, "passes", r["info"]["iterations"], "solve %.2f ms" % (dt * 1e3), [t["kind"] for t in r["trace"]], [t["factor_branch"] for t in r["trace"]])
s.delete()
