"""Scratch: mid-size instances through the PCG path (n > dense_max_n) with different constraint mixes; prints status,
counts, KKT residuals and which linear-solver modes ran."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qpdo_amd import problems, solver
cases = [(101, 20000, 40000, 0.004, 0), (102, 20000, 40000, 0.004, 8000), (103, 15000, 60000, 0.003, 0), (104, 30000, 20000, 0.003, 5000),
         (105, 16000, 16000, 0.005, 12000)]
for seed, n, m, dens, neq in cases:
    p = problems.random_qp(seed, n, m, dens, neq)
    t = time.time(); r = solver.solve_problem(p, verbose=0); dt = time.time() - t
    rp, rd = problems.kkt_residuals(p, r["x"], r["y"]) if r["info"]["status_val"] == 1 else (float("nan"),) * 2
    st = r["stats"]; na = [t_["n_active"] for t_ in r["trace"] if t_["kind"] == 0]
    print((seed, n, m, dens, neq), r["info"]["status"], "it", r["info"]["iterations"], "ot", r["info"]["oterations"], "%.2fs" % dt, "kkt %.1e %.1e" % (rp, rd),
          "newton", st["newton_passes"], "schur", st["schur_passes"], "lin", st["lin_iters"], "k_max/n %.2f" % (max(na) / n), flush=True)
