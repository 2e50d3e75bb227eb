"""Scratch: dense LDL' vs PCG (Schur / deflated Jacobi) over n, m = 2n, 1 % fill (both linear solvers forced)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
for n in ([int(a) for a in sys.argv[1:]] or [1000, 2000, 4000, 6000, 8000, 10000, 12000, 14000, 16000]):
    p = problems.random_qp(500 + n, n, 2 * n, 0.01, 0)
    out = []
    for mode in ("dense", "pcg"):
        if mode == "dense" and n > 30000: out.append("-"); continue
        os.environ["QPDO_LINSOLVE"] = mode
        os.environ["QPDO_DENSE_MAX_N"] = "30000"
        t = time.time(); r = solver.solve_problem(p, verbose=0); dt = time.time() - t
        out.append("%s %.3fs (it %d, %s)" % (mode, r["info"]["solve_time"], r["info"]["iterations"], r["info"]["status"]))
    print(n, " | ".join(out), flush=True)
