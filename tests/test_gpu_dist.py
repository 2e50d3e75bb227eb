"""Row-partitioned single QP (BASELINE.json configs[3]) on ONE GPU: the ranks are separate processes sharing
the device and exchange through a gloo host callback, which exercises the partition logic, the replicated
control flow and the all-reduce call sites (the RCCL backend uses the same call sites on the stream)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_row_partitioned_solve_matches_oracle(world, gpu_required):
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29590 + world), os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("[{")][-1]
    res = json.loads(line)
    for r in res:
        assert r["ok_counts"], r
        assert r["err"] <= 1e-8, r
        assert r["linsolve"] == 0
    # the larger instance must have gone through the row-partitioned Schur-complement mode: k-vectors partitioned, ONE
    # collective per inner iteration (the n-vector of A_c' u with the three CG scalars appended) plus one per inner solve
    # for the first application of S' 
    sch = [r for r in res if r["name"] == "schur"][0]
    assert sch["schur_passes"] > 0 and sch["inner_steps"] > 0
    assert sch["inner_collectives"] == sch["inner_steps"] + sch["inner_solves"]
    assert sch["collectives"] > sch["inner_collectives"]


def test_sharded_batch_two_processes_bit_identical_to_oracle(gpu_required):
    """BASELINE.json configs[2] across ranks: two processes share the one GPU, rank r solves
    items r::2 of a batch of C3 instances with ONE fused-kernel launch and no collective; every item must carry the
    oracle's status / pass counts and the oracle's bits (x, y, objective), and the merged result must be complete."""
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29597", os.path.join(ROOT, "tests", "_shard_worker.py"), "24", "solve"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["merged"] == list(range(24))
    for k, rk in enumerate(r["ranks"]):
        assert rk["indices"] == list(range(k, 24, 2)) and rk["failed"] == 0
        for it in rk["items"]:
            assert it["counts_equal"] and it["bit_identical"], it


def test_rccl_backend_runs_on_a_single_rank_communicator(gpu_required):
    """The RCCL branch of the exchange (ncclCommInitRank + ncclAllReduce on the solver's stream) cannot be shared by two
    ranks on one GPU, so it is exercised with a forced world-1 communicator: the partition is the whole problem and every
    collective call site of the row-partitioned solver -- Ruiz norms, A x, A'y, the K products, the one all-reduce per
    inner iteration of the Schur-complement mode -- goes through ncclAllReduce.  Results must equal the oracle's, and a
    second workspace on the spent unique id must be refused with a message (not hang)."""
    code = r"""
import json, os, sys
sys.path.insert(0, %r)
os.environ["QPDO_DEVICE"] = "0"
import numpy as np
from oracle import binding as ob
from qpdo_amd import problems, solver
assert solver.dist_config(0, 1, mode="rccl", force=True) == 0
p = problems.random_qp(61, 700, 1400, 0.03, 0)
r = solver.solve_problem(p, verbose=0)
o = ob.OracleSolver(p, ob.default_settings()); ro = o.solve(); o.close()
second = "ok"
try:
    solver.solve_problem(problems.config_qp("C1b"), verbose=0)
except RuntimeError as e:
    second = "refused"
print(json.dumps(dict(counts=[r["info"][k] == ro["info"][k] for k in ("status_val", "iterations", "oterations")],
                      err=float(max(np.abs(r["x"] - ro["x"]).max(), np.abs(r["y"] - ro["y"]).max())),
                      stats={k: r["stats"][k] for k in ("linsolve", "schur_passes", "collectives", "inner_steps", "inner_solves", "inner_collectives")},
                      second=second)))
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert all(r["counts"]) and r["err"] <= 1e-8, r
    st = r["stats"]
    assert st["linsolve"] == 0 and st["schur_passes"] > 0 and st["collectives"] > 0
    assert st["inner_collectives"] == st["inner_steps"] + st["inner_solves"]
    assert r["second"] == "refused"


@pytest.mark.parametrize("workload,extra", [("C2", ["--steps", "3", "--warmup", "1", "--no-cpu-baseline"]),
                                            ("C3", ["--steps", "2", "--warmup", "1", "--max-iter", "300", "--batch-count", "512", "--no-cpu-baseline"])])
def test_bench_contract_with_two_ranks_on_one_gpu(workload, extra, gpu_required):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU; here both ranks land on the one GPU):
    one JSON line from rank 0 with the whole-job aggregate; C2: independent QPs, each rank cycling through two seeded instances;
    C3: the batch sharded over the ranks"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29613" if workload == "C2" else "29614", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["unit"] == "newton_iters/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    if workload == "C2":
        assert d["scaling"] == "weak" and d["config"]["instances_per_rank"] == 2 and all(v == 1 for v in d["status_val"])
        assert d["newton_passes"] > 2 * 3 * 30               # both ranks' passes are in the aggregate
    else:
        assert d["scaling"] == "strong" and d["items"] == 2 * 512 and d["failed"] == 0
