"""Scratch: solve a configuration without scaling and save the final factor weights d and sigma (for CPU experiments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qpdo_amd import problems, solver
name = sys.argv[1]
p = problems.config_qp(name)
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0, scaling=0)
r = s.solve()
d = s.download("d"); mu = s.download("mu")
tr = s.trace()
print(r["info"]["status"], r["info"]["iterations"], "sigma", tr[-1]["sigma"], "k", int((d != 0).sum()), "lin", [t["lin_iters"] for t in tr if t["kind"] == 0][-8:])
np.savez_compressed(os.path.join("gpurun_out", "d_%s.npz" % name), d=d, mu=mu, sigma=tr[-1]["sigma"])
