import os, sys, time
sys.path.insert(0, os.getcwd())
from qpdo_amd import problems, solver
cases = (("C1", problems.config_qp("C1")), ("n300", problems.random_qp(31, 300, 600, 0.1)), ("n500", problems.random_qp(32, 500, 1000, 0.05)),
         ("n1000", problems.random_qp(33, 1000, 2000, 0.02)), ("n2000", problems.random_qp(34, 2000, 4000, 0.01)))
for rnd in range(2):
  for name, p in cases:
    out = []
    for ahead in ("0", "1"):
        os.environ["QPDO_LAUNCH_AHEAD"] = ahead
        s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
        best = 1e9
        for rep in range(5):
            t = time.time(); r = s.solve(); best = min(best, time.time() - t)
        st = s.stats(); s.delete()
        out.append(f"{best*1e3:.3f} ms ({r['info']['iterations']} it, ahead {st['ahead_steps']}/{st['ahead_skips']})")
    print(name, " | ".join(out), flush=True)
