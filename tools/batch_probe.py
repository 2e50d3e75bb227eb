"""Scratch: throughput of the threaded batch API on C3-sized QPs, and dense-vs-PCG crossover at mid sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
probs = [problems.config_qp("C3", i) for i in range(nq)]
for nt in (1, 4, 16, 32):
    t = time.time(); res, failed = solver.solve_batch(probs, nthreads=nt, verbose=0); dt = time.time() - t
    newton = sum(r["info"]["iterations"] - r["info"]["oterations"] for r in res)
    print(f"batch {nq} QPs, {nt} threads: {dt:.3f}s  {nq/dt:.1f} QP/s  {newton/dt:.0f} Newton it/s  failed {failed} solved {sum(r['info']['status_val']==1 for r in res)}", flush=True)
for n in (1000, 2000, 4000, 6000):
    p = problems.random_qp(5, n, 2 * n, 0.01)
    for ls in ("dense", "pcg"):
        os.environ["QPDO_LINSOLVE"] = ls
        t = time.time(); r = solver.solve_problem(p, verbose=0); dt = time.time() - t
        print(f"n={n} {ls}: total {dt:.3f}s solve {r['info']['solve_time']:.3f}s it {r['info']['iterations']} lin {r['stats']['lin_iters']}", flush=True)
