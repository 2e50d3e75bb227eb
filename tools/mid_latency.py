"""Cold-solve wall time of mid-size single QPs on the default device path (too large for the fused one-workgroup route) beside the CPU
oracle on this box's host (dense LDL', all cores): where the launch-bound generic path stands between the fused kernel and the large-problem kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob
from qpdo_amd import problems, solver
for name, p in (("C1 n=200 m=100", problems.config_qp("C1")), ("n=300 m=600", problems.random_qp(31, 300, 600, 0.1)), ("n=500 m=1000", problems.random_qp(32, 500, 1000, 0.05)),
                ("n=1000 m=2000", problems.random_qp(33, 1000, 2000, 0.02)), ("n=2000 m=4000", problems.random_qp(34, 2000, 4000, 0.01)), ("n=4000 m=8000", problems.random_qp(35, 4000, 8000, 0.01))):
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    best = 1e9
    for rep in range(3):
        t = time.time(); r = s.solve(); best = min(best, time.time() - t)
    st = s.stats(); s.delete()
    t = time.time(); o = ob.OracleSolver(p, ob.default_settings()); ro = o.solve(); to = time.time() - t; oi = dict(ro["info"]); o.close()
    print(f"{name}: GPU {best*1e3:.2f} ms ({r['info']['iterations']} passes, {best*1e6/max(1,r['info']['iterations']):.0f} us/pass, linsolve {st['linsolve']}, {st['factor_count']} factorizations)"
          f"   CPU oracle setup+solve {to*1e3:.1f} ms ({oi['iterations']} passes, {ob.get_threads()} threads)", flush=True)
