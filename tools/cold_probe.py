"""Scratch: is the first solve of a process slower than the second (lazy code-object loading, first-touch)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
p = problems.config_qp(name) if name in problems.CONFIGS else problems.random_qp(201, 16000, 24000, 0.004, 0)
for rep in range(3):
    t = time.time(); s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0); ts = time.time() - t
    t = time.time(); r = s.solve(); dt = time.time() - t
    print(name, "rep", rep, "setup %.3f solve %.3f" % (ts, dt), r["info"]["status"], r["info"]["iterations"], flush=True)
    s.delete()
