"""Worker for tests/test_dist_cpu.py: exercises bench.py's rank plumbing (rendezvous over gloo, per-rank
instance selection, barrier, max/sum reductions) without touching a GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from qpdo_amd import problems                  # noqa: E402

rank, world, dist = bench.dist_setup(int(os.environ["WORLD_SIZE"]))
assert os.environ["QPDO_DEVICE"] == os.environ.get("LOCAL_RANK", str(rank))
p = problems.config_qp("C1b", index=rank)      # the shard of this rank: its own independent QP
bench.barrier(dist)
tmax = bench.allreduce(dist, [float(rank + 1)], "max")[0]
tot = bench.allreduce(dist, [10.0 * (rank + 1), float(p["seed"])], "sum")
if rank == 0:
    print(json.dumps(dict(world=world, tmax=tmax, tot=tot, seed0=p["seed"])))
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
