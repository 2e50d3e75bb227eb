"""Latency of the fused one-workgroup kernel on ONE QP (the candidate fast path of qpdo_solve for small problems): wall time through
the batch entry point with count = 1, HIP-event kernel time, per-pass time, and the in-kernel phase shares (QPDO_SMALL_PROF=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
cases = [("C3#0", problems.config_qp("C3", 0), {}), ("C3#1", problems.config_qp("C3", 1), {}), ("C1b", problems.config_qp("C1b"), dict(max_iter=200)),
         ("KAT", problems.infeasibility_kat("degenerate"), dict(max_iter=100)), ("C1", problems.config_qp("C1"), dict(max_iter=200)),
         ("n60m180", problems.random_qp(7, 60, 180, 0.1, 60), {}), ("n180m500", problems.random_qp(8, 180, 500, 0.05, 100), {})]
for name, p, st in cases:
    B = solver.Batch([p])
    os.environ.pop("QPDO_SMALL_PROF", None)
    best = 1e9
    for rep in range(5):
        t = time.time(); res, failed = B.run(verbose=0, **st); dt = time.time() - t
        best = min(best, dt)
    i = res[0]["info"]
    ks = B.kernel_seconds
    print(f"{name}: n={p['n']} m={p['m']} status {i['status_val']} passes {i['iterations']} ({i['oterations']} outer)  wall {best*1e3:.2f} ms  kernel {ks*1e3:.2f} ms  "
          f"kernel/pass {ks*1e6/max(1,i['iterations']):.1f} us  in-kernel setup {i['setup_time']*1e3:.3f} ms solve {i['solve_time']*1e3:.3f} ms", flush=True)
    os.environ["QPDO_SMALL_PROF"] = "1"
    B.run(verbose=0, **st)
    sys.stderr.flush()
os.environ.pop("QPDO_SMALL_PROF", None)
