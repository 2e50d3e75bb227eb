import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
for name, p, st in [("C1", problems.config_qp("C1"), dict(max_iter=200)), ("C1b", problems.config_qp("C1b"), dict(max_iter=200)),
                    ("C3", problems.config_qp("C3"), {}), ("KAT", problems.infeasibility_kat("degenerate"), dict(max_iter=100))]:
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0, **st)
    for rep in range(3):
        t = time.time(); r = s.solve(); dt = time.time() - t
    i = r["info"]
    print(f"{name}: n={p['n']} m={p['m']} passes {i['iterations']} solve {dt*1e3:.2f} ms  per pass {dt*1e3/max(1,i['iterations']):.3f} ms  setup {i['setup_time']*1e3:.1f} ms", flush=True)
    s.delete()
