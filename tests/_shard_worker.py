"""Worker for the batch-sharding tests (BASELINE.json configs[2]: a batch of MPC-sized QPs spread over the GPUs, no
data-path collective).  Rank r builds the shard r::world of a batch of C3 instances, optionally solves it with the
fused batch kernel on GPU 0 (the ranks share the one GPU of the test box), and the per-rank records are gathered
over gloo; rank 0 prints one JSON line.

usage: _shard_worker.py COUNT [solve]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                              # noqa: E402
import torch.distributed as dist                # noqa: E402
from qpdo_amd import problems, solver           # noqa: E402

count = int(sys.argv[1])
do_solve = len(sys.argv) > 2 and sys.argv[2] == "solve"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
made = []


def make(i):
    made.append(i)
    return problems.config_qp("C3", i)


B = solver.shard_batch(count, rank, world, make)
rec = dict(rank=rank, indices=B.indices, made=made, seeds=[int(problems.config_qp("C3", i)["seed"]) for i in B.indices[:2]])
if do_solve:
    os.environ["QPDO_DEVICE"] = "0"
    from oracle import binding as ob
    res, failed = B.run(verbose=0)
    rec["failed"] = failed
    rec["items"] = []
    for i, r in zip(B.indices, res):
        p = problems.config_qp("C3", i)
        o = ob.OracleSolver(p, ob.default_settings()); ro = o.solve(); o.close()
        rec["items"].append(dict(
            index=i, status=r["info"]["status_val"], iterations=r["info"]["iterations"],
            counts_equal=(r["info"]["status_val"], r["info"]["iterations"], r["info"]["oterations"]) ==
                         (ro["info"]["status_val"], ro["info"]["iterations"], ro["info"]["oterations"]),
            bit_identical=bool(np.array_equal(r["x"], ro["x"]) and np.array_equal(r["y"], ro["y"])
                               and r["info"]["objective"] == ro["info"]["objective"]),
            xsum=float(np.sum(r["x"]))))
box = [None] * world
dist.all_gather_object(box, rec)
if rank == 0:
    merged = None
    if do_solve:      # the caller-side merge: global order, nothing missing, nothing twice
        merged = solver.merge_shards(count, [(b["indices"], b["items"]) for b in box])
        merged = [m["index"] for m in merged]
    print(json.dumps(dict(world=world, count=count, ranks=box, merged=merged)))
dist.barrier()
dist.destroy_process_group()
