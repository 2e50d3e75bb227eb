"""Solve time with and without the low-rank update of the kept dense factor, by size (m = 2n, 1 % fill, at least 8 entries per row).
usage: lowrank_crossover.py n [n ...]   (QPDO_DENSE_LOWRANK=0/1 selects the mode)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
os.environ["QPDO_LINSOLVE"] = "dense"
for n in [int(a) for a in sys.argv[1:]]:
    p = problems.random_qp(11, n, 2 * n, max(0.01, 8.0 / n))
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    best = 1e9
    for rep in range(3):
        t = time.time(); r = s.solve(); best = min(best, time.time() - t)
    st = s.stats()
    print("n=%5d passes %3d status %d  solve %8.2f ms  factors %3d lowrank solves %3d" % (n, r["info"]["iterations"], r["info"]["status_val"], best * 1e3, st["factor_count"], st["lowrank_solves"]), flush=True)
    s.delete()
