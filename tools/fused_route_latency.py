"""qpdo_solve latency of small workspaces on the fused route (one launch of k_small_solve_lat): wall time of a cold solve, HIP-event
kernel time, and -- with QPDO_SMALL_PROF=1 in the environment -- the in-kernel phase times on stderr.  Extra settings as k=v arguments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
extra = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    extra[k] = float(v) if "." in v or "e" in v else int(v)
cases = [("C3#0", problems.config_qp("C3", 0), {}), ("C1b", problems.config_qp("C1b"), dict(max_iter=200)),
         ("KAT", problems.infeasibility_kat("degenerate"), dict(max_iter=100)), ("n60m180", problems.random_qp(9, 60, 180, 0.1, 20), {}),
         ("n150m400", problems.random_qp(8, 150, 400, 0.05, 50), {}),
         ("C3#322 (never reaches eps: 10000 passes)", problems.config_qp("C3", 322), {})]
for name, p, st in cases:
    st = dict(st, **extra)
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0, **st)
    best = 1e9
    for rep in range(4):
        t = time.time(); r = s.solve(); dt = time.time() - t
        best = min(best, dt)
    i, stt = r["info"], s.stats()
    print(f"{name}: n={p['n']} m={p['m']} linsolve {stt['linsolve']} status {i['status_val']} passes {i['iterations']} ({i['oterations']} outer) factorizations {stt['factor_count']}  "
          f"solve wall {best*1e3:.3f} ms  kernel {stt['fused_kernel_s']*1e3:.3f} ms  per pass {best*1e6/max(1,i['iterations']):.1f} us", flush=True)
    s.delete()
