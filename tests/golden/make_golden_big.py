"""Generates the production-size oracle fixtures tests/golden/big_<name>.npz (build container only; minutes to
hours of CPU on all cores: the oracle's dense LDL' is blocked + OpenMP, bit-identical to its scalar loop).

Each fixture holds the final record and the whole per-pass trace of the oracle (oracle/qpdo_oracle.c, the CPU
restatement of the reference; direct natural-order LDL' = the reference's algorithm, cholmod_interface.c:35-52)
on one seeded instance of the product's own generator, so the -m gpu tests can compare the DEFAULT device
solver selection with it on the GPU box, where neither the reference nor hours of CPU are available:

    status_val, iterations, oterations, newton_passes, objective, the 4 residual norms,
    per pass: kind, n_active, n_enter, n_leave, factor_branch, tau, res_prim, res_dual, res_prim_in,
              res_dual_in, sigma, eps_in
    x, y (unscaled solution)

Run:  python tests/golden/make_golden_big.py [name ...]      (default: all CASES)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402
from qpdo_amd import problems             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (problem spec, settings overrides).  spec: cfg/index (problems.config_qp) or rand=(seed,n,m,density,n_eq)
CASES = {
    # BASELINE.json configs[1] at full size: default device path = dense MFMA LDL' + low-rank updates
    "C2": (dict(cfg="C2", index=0), dict()),
    # above QPDO_DENSE_MAX_N: default device path = PCG, Schur-complement mode, slab SpMV kernels auto-selected
    "schur_30k": (dict(rand=(123456 + 7, 30_000, 60_000, 0.01, 0)), dict()),
    # mid size with equality rows (tie hazard) just above the dense limit
    "pcg_13k_eq": (dict(rand=(123456 + 8, 13_000, 20_000, 0.01, 3000)), dict()),
    # BASELINE configs[3] / the metric's configuration at FULL size (n=1e5, m=2e5), first passes only: the dense factor does not
    # exist at this size (80 GB), so the oracle runs its Jacobi-PCG (tol 1e-12) -- ~0.5 s per CG iteration on 8 cores, hours for
    # the whole solve -- and both sides stop at max_iter: the record pins the per-pass trace and the iterate after those passes
    "C4_first16": (dict(cfg="C4", index=0), dict(max_iter=16)),
    "C4_first32": (dict(cfg="C4", index=0), dict(max_iter=32)),
    "C4_first40": (dict(cfg="C4", index=0), dict(max_iter=40)),
    # the whole solve of the metric's configuration (round 3: the oracle's PCG operator streams compact 32-bit copies,
    # bit-identical to its plain loops -- tests/test_oracle_specs.py -- which makes the 65 passes a matter of hours)
    "C4_full": (dict(cfg="C4", index=0), dict()),
    # round 4: a chain-structured (banded) instance ABOVE the former limit of the dense direct solver (n <= 18000, an LDS accumulator):
    # the forced dense path, the default PCG path and the PCG -> dense rescue are all compared with this record
    "banded_20k": (dict(banded=(1, 20_000, {})), dict()),
    # the same structure without the proximal term (a slightly regularised Q keeps K definite): proximal = 0 drops the absolute stopping
    # rule of the PCG and is where Jacobi-PCG solves have failed before (tests/test_gpu_sweep.py instance 1358)
    "banded_20k_noprox": (dict(banded=(2, 20_000, dict(q_reg=1e-4))), dict(proximal=0)),
}
LINSOLVE = {"C4_first16": "pcg", "C4_first32": "pcg", "C4_first40": "pcg", "C4_full": "pcg"}
TRACE_FIELDS = ["kind", "n_active", "n_enter", "n_leave", "factor_branch", "tau", "res_prim", "res_dual",
                "res_prim_in", "res_dual_in", "sigma", "eps_in", "lin_iters", "t_end"]


def make_problem(spec):
    if "cfg" in spec:
        return problems.config_qp(spec["cfg"], spec["index"])
    if "banded" in spec:
        seed, n, kw = spec["banded"]
        return problems.banded_qp(seed, n, **kw)
    seed, n, m, dens, neq = spec["rand"]
    return problems.random_qp(seed, n, m, dens, neq)


def main():
    names = sys.argv[1:] or list(CASES)
    for name in names:
        spec, st = CASES[name]
        t0 = time.time()
        p = make_problem(spec)
        o = ob.OracleSolver(p, ob.default_settings(**st), linsolve=LINSOLVE.get(name, "dense"), pcg_tol=1e-12, pcg_maxit=50000)
        r = o.solve()
        tr = o.trace()
        i = r["info"]
        meta = dict(spec=spec, settings=st, n=p["n"], m=p["m"],
                    info={k: i[k] for k in ("status_val", "iterations", "oterations", "newton_passes", "objective",
                                            "res_prim_norm", "res_dual_norm", "res_prim_in_norm", "res_dual_in_norm")},
                    oracle_seconds=time.time() - t0, threads=os.cpu_count(), oracle_linsolve=LINSOLVE.get(name, "dense"),
                    oracle_lin_iters=i.get("lin_iters", 0))
        arrays = {f: np.array([t[f] for t in tr]) for f in TRACE_FIELDS}
        path = os.path.join(os.environ.get("GOLDEN_OUT", HERE), "big_%s.npz" % name)
        np.savez_compressed(path, meta=json.dumps(meta), x=r["x"], y=r["y"], **{"tr_" + k: v for k, v in arrays.items()})
        o.close()
        print("wrote %s (%d bytes): status %d, %d passes (%d outer), %.0f s" % (
            path, os.path.getsize(path), i["status_val"], i["iterations"], i["oterations"], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
