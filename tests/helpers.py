import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_PATH = os.path.join(ROOT, "tests", "golden", "oracle_golden.json")

# stated fp64 tolerances of the parity tests (BASELINE.json north_star: iteration count and status
# bit-identical, iterates and KKT residuals within a stated fp64 tolerance)
ITERATE_RTOL = 1e-9      # |x - x_ref|inf <= ITERATE_RTOL * max(1, |x_ref|inf)   (dense LDL' solver)
ITERATE_RTOL_PCG = 1e-8  # same bound for the Jacobi-PCG solver: its stopping rule is a relative residual of 1e-12 on
                         # systems whose condition number reaches 1e12 late in a solve (weights 1/mu up to 1e9)
KKT_ATOL = 1e-10         # |KKT residual - reference KKT residual| <= KKT_ATOL


def load_golden():
    with open(GOLDEN_PATH) as f:
        return json.load(f)


def golden_problem(spec):
    from qpdo_amd import problems
    if "kat" in spec:
        return problems.infeasibility_kat(spec["kat"])
    if "cfg" in spec:
        return problems.config_qp(spec["cfg"], spec["index"])
    seed, n, m, dens, neq = spec["rand"]
    return problems.random_qp(seed, n, m, dens, neq)


def close_vec(a, b, rtol=ITERATE_RTOL):
    a, b = np.asarray(a, float), np.asarray(b, float)
    if np.isnan(b).all():
        return bool(np.isnan(a).all())
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    return bool(np.abs(a - b).max() <= rtol * scale) if b.size else True
