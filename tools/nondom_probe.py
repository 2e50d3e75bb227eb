"""Scratch: a QP whose Q is NOT diagonally dominant (Q = B B' + 1e-3 I, sparse B): Schur mode vs deflated Jacobi-PCG."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from qpdo_amd import problems, solver
n, m = 16000, 24000
rng = np.random.default_rng(5)
p = problems.random_qp(201, n, m, 0.004, 0)
B = sp.random(n, n, density=0.0006, random_state=7, data_rvs=rng.standard_normal, format="csc")
Qf = (B @ B.T + 1e-3 * sp.identity(n)).tocsc()
p["Q"] = sp.tril(Qf).tocsc(); p["Qstype"] = -1
dq = Qf.diagonal(); off = abs(Qf).sum(axis=1).A.ravel() - dq
print("nnz(Q)", Qf.nnz, "offdiag/diag median %.2f max %.2f" % (np.median(off / dq), (off / dq).max()), flush=True)
for mode in ("0", "1"):
    os.environ["QPDO_PCG_SCHUR"] = mode
    t = time.time(); r = solver.solve_problem(p, verbose=0); dt = time.time() - t
    st = r["stats"]
    print("schur", mode, r["info"]["status"], "it", r["info"]["iterations"], "%.2fs" % dt, "schur passes", st["schur_passes"], "lin", st["lin_iters"], flush=True)
