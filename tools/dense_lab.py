"""Dense LDL' factor timing at C2 (or --n N): solve once (weights of the final pass), then time the factor alone.
usage: dense_lab.py [n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = problems.config_qp("C2") if n == 10000 else problems.random_qp(5, n, 2 * n, 0.01)
os.environ["QPDO_LINSOLVE"] = "dense"
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
t0 = time.time(); r = s.solve(); dt = time.time() - t0
st = s.stats()
t, chk = s.bench_dense_factor(reps)
print("n=%d solve %.3fs its %d factors %d lowrank %d | factor %.3f ms = %.1f TF/s, solve residual %.2e" % (
    p["n"], dt, r["info"]["iterations"], st["factor_count"], st["lowrank_solves"], t * 1e3, p["n"] ** 3 / 3 / t / 1e12, chk), flush=True)
