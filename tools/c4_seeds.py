"""Scratch: the C4 instances the ranks of `bench.py --gpus N` solve (index = rank): status, counts, time, KKT."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
for idx in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    p = problems.config_qp("C4", index=idx)
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    t = time.time(); r = s.solve(); dt = time.time() - t
    st = s.stats(); s.delete()
    rp, rd = problems.kkt_residuals(p, r["x"], r["y"]) if r["info"]["status_val"] == 1 else (float("nan"),) * 2
    print("C4 index", idx, r["info"]["status"], "it", r["info"]["iterations"], "ot", r["info"]["oterations"], "%.2f s" % dt, "schur", st["schur_passes"], "of", st["newton_passes"], "kkt %.1e %.1e" % (rp, rd), flush=True)
