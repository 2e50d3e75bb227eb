"""Scratch: time setup/solve at C2 and C4 and print per-pass PCG iteration counts."""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
name = sys.argv[1]; max_time = float(sys.argv[2]) if len(sys.argv) > 2 else 0
t = time.time(); p = problems.config_qp(name); print("gen", time.time() - t, flush=True)
t = time.time(); kw = dict(max_time=max_time) if max_time else {}
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0, **kw); print("setup", time.time() - t, flush=True)
for w in (0, 1, 2):
    tt, b = s.bench_spmv(w, 20); print("spmv", w, "%.3f ms  %.1f GB/s" % (tt * 1e3, b / tt / 1e9), flush=True)
t = time.time(); r = s.solve(); dt = time.time() - t
print("solve", dt, r["info"], s.stats(), flush=True)
print("kkt", problems.kkt_residuals(p, r["x"], r["y"]))
print("lin per pass", [t_["lin_iters"] for t_ in s.trace() if t_["kind"] == 0])
print("nact", [t_["n_active"] for t_ in s.trace() if t_["kind"] == 0])
