"""One solve of a chain-structured QP through the band direct solver (for profilers).  usage: band_one_solve.py [n] [bw]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = problems.banded_random_qp(3, n, bw) if bw else problems.banded_qp(5, n)
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
t0 = time.time(); r = s.solve(); dt = time.time() - t0
st = s.stats()
print("n=%d bw=%d: solve %.3f s passes %d status %d linsolve %d factorizations %d" % (n, bw, dt, r["info"]["iterations"], r["info"]["status_val"], st["linsolve"], st["factor_count"]))
