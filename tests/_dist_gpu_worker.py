"""Worker for tests/test_gpu_dist.py.  Default: two (or more) ranks share GPU 0 and the row-partitioned solver exchanges
through a gloo host callback.  QPDO_TEST_DIST_MODE=rccl: one rank per GPU (device = LOCAL_RANK), the exchange is RCCL on
the solver's stream (needs as many GPUs as ranks).  Rank 0 compares with the oracle and prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODE = os.environ.get("QPDO_TEST_DIST_MODE", "host")
os.environ["QPDO_DEVICE"] = os.environ.get("LOCAL_RANK", "0") if MODE == "rccl" else "0"
import numpy as np                              # noqa: E402
import torch.distributed as dist                # noqa: E402
from oracle import binding as ob                # noqa: E402
from qpdo_amd import problems, solver           # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
out = []
for name, p, st in [("C1", problems.config_qp("C1"), dict(max_iter=200)),
                    ("rand_eq", problems.random_qp(23, 150, 300, 0.05, 50), {}),
                    ("kat_pinf", problems.infeasibility_kat("primal_infeasible"), dict(max_iter=100)),
                    ("noscale", problems.random_qp(24, 300, 200, 0.03), dict(scaling=0)),
                    ("schur", problems.random_qp(61, 700, 1400, 0.03, 0), {})]:   # enough active rows for the Schur-complement mode
    assert solver.dist_config(rank, world, mode=MODE) == 0      # (an RCCL unique id is one-shot: a fresh one per workspace)
    r = solver.solve_problem(p, verbose=0, **st)
    if rank == 0:
        o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve()
        ok_counts = (r["info"]["status_val"], r["info"]["iterations"], r["info"]["oterations"]) == \
                    (ro["info"]["status_val"], ro["info"]["iterations"], ro["info"]["oterations"])
        if r["info"]["status_val"] in (-3, -4):
            err = 0.0
        else:
            err = float(max(np.abs(r["x"] - ro["x"]).max(), np.abs(r["y"] - ro["y"]).max()))
        st_ = r["stats"]
        out.append(dict(name=name, ok_counts=bool(ok_counts), err=err, status=r["info"]["status_val"], linsolve=st_["linsolve"],
                        schur_passes=st_["schur_passes"], collectives=st_["collectives"], inner_solves=st_["inner_solves"],
                        inner_steps=st_["inner_steps"], inner_collectives=st_["inner_collectives"]))
        o.close()
    # every rank must hold the same solution
    chk = np.nan_to_num(np.concatenate([r["x"], r["y"]]))
    import torch
    t = torch.from_numpy(chk.copy()); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t2 = torch.from_numpy(chk.copy()); dist.all_reduce(t2, op=dist.ReduceOp.MIN)
    assert torch.equal(t, t2), "ranks disagree on the solution"
if rank == 0:
    print(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
