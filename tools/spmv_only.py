"""C4 instance, SpMV micro-benchmarks only (for rocprofv3 --pmc passes: few dispatches, known byte counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
p = problems.config_qp(name)
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0, scaling=0)
for w in (0, 1, 2):
    t, b = s.bench_spmv(w, 5)
    print("spmv", w, "alg_bytes", b, "avg_s", t, "GB/s", b / t / 1e9, flush=True)
s.delete()
