"""C2 (n=1e4, m=2e4): per-pass cost of the PCG solver against the dense factorization -- the data behind a per-pass choice
(PCG while the Newton systems are well conditioned and every pass would refactor anyway, dense LDL' afterwards)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qpdo_amd import problems, solver
p = problems.config_qp(sys.argv[1] if len(sys.argv) > 1 else "C2")
for mode in ("dense", "pcg"):
    os.environ["QPDO_LINSOLVE"] = mode
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    s.solve()
    t = time.time(); r = s.solve(); solver.lib().qpdo_amd_sync(s._w); dt = time.time() - t
    st, tr = s.stats(), s.trace()
    print(mode, "time %.3f s" % dt, "passes", r["info"]["iterations"], "factor_count", st["factor_count"], "lowrank", st["lowrank_solves"], "lin_iters", st["lin_iters"], "schur_passes", st["schur_passes"], flush=True)
    if mode == "pcg":
        print("lin_iters per Newton pass:", [int(t_["lin_iters"]) for t_ in tr if t_["kind"] == 0])
    s.delete()
os.environ.pop("QPDO_LINSOLVE")
for cap in (60, 100, 150, 250, 400):
    os.environ["QPDO_HYBRID_PCG_MAXIT"] = str(cap)
    s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    s.solve()
    t = time.time(); r = s.solve(); solver.lib().qpdo_amd_sync(s._w); dt = time.time() - t
    st = s.stats()
    print("hybrid cap", cap, "time %.3f s" % dt, "passes", r["info"]["iterations"], r["info"]["status_val"], "factor_count", st["factor_count"], "lowrank", st["lowrank_solves"], "lin_iters", st["lin_iters"],
          "fallbacks", st["pcg_dense_fallbacks"], flush=True)
    s.delete()
