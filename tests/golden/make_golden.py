"""Generates tests/golden/oracle_golden.json with the CPU oracle (oracle/qpdo_oracle.c).

The reference itself cannot run here (CHOLMOD absent), so these vectors are outputs of the oracle,
which is pinned to the reference's three known answers (tests/test_oracle_kat.py).  The GPU tests
compare the HIP path against these committed vectors as well as against the live oracle.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402
from qpdo_amd import problems             # noqa: E402

CASES = [
    ("kat_degenerate", dict(kat="degenerate"), dict(max_iter=100)),
    ("kat_primal_infeasible", dict(kat="primal_infeasible"), dict(max_iter=100)),
    ("kat_dual_infeasible", dict(kat="dual_infeasible"), dict(max_iter=100)),
    ("C1", dict(cfg="C1", index=0), dict(max_iter=200)),
    ("C1b", dict(cfg="C1b", index=0), dict(max_iter=200)),
    ("C1_noscale", dict(cfg="C1", index=0), dict(max_iter=200, scaling=0)),
    ("C1_noprox", dict(cfg="C1", index=0), dict(max_iter=200, proximal=0)),
    ("C3_0", dict(cfg="C3", index=0), dict()),
    ("C3_1", dict(cfg="C3", index=1), dict()),
    ("small_100x200", dict(rand=(11, 100, 200, 0.05, 0)), dict()),
]


def make_problem(spec):
    if "kat" in spec:
        return problems.infeasibility_kat(spec["kat"])
    if "cfg" in spec:
        return problems.config_qp(spec["cfg"], spec["index"])
    seed, n, m, dens, neq = spec["rand"]
    return problems.random_qp(seed, n, m, dens, neq)


def main():
    out = {}
    for name, spec, st in CASES:
        p = make_problem(spec)
        o = ob.OracleSolver(p, ob.default_settings(**st))
        r = o.solve()
        tr = o.trace()
        i = r["info"]
        out[name] = dict(
            spec=spec, settings=st,
            status_val=i["status_val"], iterations=i["iterations"], oterations=i["oterations"],
            newton_passes=i["newton_passes"], objective=i["objective"],
            res_prim_norm=i["res_prim_norm"], res_dual_norm=i["res_dual_norm"],
            x=[float(v) for v in r["x"]], y=[float(v) for v in r["y"]],
            prim_inf_cert=[float(v) for v in r["prim_inf_cert"]],
            dual_inf_cert=[float(v) for v in r["dual_inf_cert"]],
            kinds=[t["kind"] for t in tr], n_active=[t["n_active"] for t in tr],
            # the whole per-pass trace (TraceRec of oracle/qpdo_oracle.c): one list per field
            trace={f: [t[f] for t in tr] for f in ("kind", "n_active", "n_enter", "n_leave", "factor_branch", "tau",
                                                    "res_prim", "res_dual", "res_prim_in", "res_dual_in", "sigma", "eps_in")},
        )
        o.close()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, allow_nan=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
