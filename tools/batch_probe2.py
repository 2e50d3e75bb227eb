import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 300
t = time.time(); probs = []
for i in range(nq):
    probs.append(problems.config_qp("C3", i))
    if i % 512 == 0: print("gen", i, time.time() - t, flush=True)
print("gen done", time.time() - t, flush=True)
t = time.time(); B = solver.Batch(probs); print("python-side batch image %.3f s" % (time.time() - t), flush=True)
for rep in range(3):
    t = time.time(); res, failed = B.run(verbose=0, max_iter=maxit); dt = time.time() - t
    newton = sum(r["info"]["iterations"] - r["info"]["oterations"] for r in res)
    print(f"fused batch {nq} QPs: {dt:.3f}s  {nq/dt:.1f} QP/s  {newton/dt:.0f} Newton it/s  failed {failed} solved {sum(r['info']['status_val']==1 for r in res)}", flush=True)
