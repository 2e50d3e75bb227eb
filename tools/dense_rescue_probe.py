"""Dense direct solver above the former LDS limit (n > 18000) and as the rescue of a PCG solve that cannot converge: a chain-structured
(banded) QP through the default path, the forced dense path and with proximal = 0; and the tiled assembly against the untiled one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qpdo_amd import problems, solver

def run(p, label, env=None, **st):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    t = time.time()
    try:
        r = solver.solve_problem(p, verbose=0, **st)
        dt = time.time() - t
        i, s = r["info"], r["stats"]
        kk = problems.kkt_residuals(p, r["x"], r["y"]) if i["status_val"] == 1 else None
        print(f"{label}: status {i['status_val']} passes {i['iterations']} ({i['oterations']} outer) {dt:.2f} s linsolve {s['linsolve']} factor_count {s['factor_count']} "
              f"lin_iters {s['lin_iters']} pcg_dense_fallbacks {s['pcg_dense_fallbacks']} soft {s['pcg_soft_accepts']} maxrel {s['pcg_max_relres']:.1e} kkt {kk}", flush=True)
    except Exception as e:
        r = None
        print(label, "FAILED", repr(e), flush=True)
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    return r

if "tile" in sys.argv[1:] or len(sys.argv) == 1:
    p = problems.random_qp(5, 3000, 5000, 0.02, 100)
    a = run(p, "n=3000 dense, one tile", dict(QPDO_LINSOLVE="dense"))
    b = run(p, "n=3000 dense, tile 512", dict(QPDO_LINSOLVE="dense", QPDO_DENSE_ASM_TILE="512"))
    print("tiled == untiled bits:", np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]) and
          all(ta[k] == tb[k] for ta, tb in zip(a["trace"], b["trace"]) for k in ta), flush=True)
if "band" in sys.argv[1:] or len(sys.argv) == 1:
    for n in (4000, 20000):
        p = problems.banded_qp(1, n)
        run(p, f"banded n={n} default")
        run(p, f"banded n={n} forced dense", dict(QPDO_LINSOLVE="dense"))
        run(p, f"banded n={n} proximal=0 default", None, proximal=0)
