"""Scratch probe: run the HIP path next to the oracle on a few instances and print both."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qpdo_amd import problems, solver
from oracle import binding as ob

def run(name, p, **st):
    so = ob.default_settings(**st)
    o = ob.OracleSolver(p, so); ro = o.solve(); io = ro["info"]
    t = time.time()
    r = solver.solve_problem(p, verbose=0, **st); ig = r["info"]
    dt = time.time() - t
    print(f"{name}: oracle st {io['status_val']} it {io['iterations']} ot {io['oterations']} | "
          f"gpu st {ig['status_val']} it {ig['iterations']} ot {ig['oterations']} lin {r['stats']['lin_iters']} t {dt:.3f}s")
    if ig["status_val"] not in (-3, -4) and io["status_val"] not in (-3, -4):
        print("    |dx|inf", np.abs(r["x"] - ro["x"]).max(), "|dy|inf", np.abs(r["y"] - ro["y"]).max(),
              "kkt", problems.kkt_residuals(p, r["x"], r["y"]))
    to, tg = o.trace(), r["trace"]
    for k in range(min(len(to), len(tg), 60)):
        a, b = to[k], tg[k]
        flag = "" if (a["kind"] == b["kind"] and a["n_active"] == b["n_active"]) else "  <<<<"
        if flag or k < 4:
            print(f"    {k:3d} kind {a['kind']}/{b['kind']} nact {a['n_active']}/{b['n_active']} tau {a['tau']:.6e}/{b['tau']:.6e} "
                  f"rp {a['res_prim']:.3e}/{b['res_prim']:.3e} rd {a['res_dual']:.3e}/{b['res_dual']:.3e}{flag}")

print("devices", solver.device_count())
for case in ["degenerate", "primal_infeasible", "dual_infeasible"]:
    p = problems.infeasibility_kat(case)
    run(case, p, max_iter=100)
for nm in ["C1", "C1b", "C3"]:
    run(nm, problems.config_qp(nm), max_iter=200 if nm.startswith("C1") else 10000)
run("C1-noscale", problems.config_qp("C1"), scaling=0)
run("mid", problems.random_qp(7, 1000, 2000, 0.01))
