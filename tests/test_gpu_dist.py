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
    # the larger instance must have gone through the row-partitioned Schur-complement mode (global compact index space,
    # two all-reduces per inner iteration)
    assert [r for r in res if r["name"] == "schur"][0]["schur_passes"] > 0


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
