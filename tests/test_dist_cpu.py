"""N > 1 path of bench.py on CPU: two ranks over gloo (127.0.0.1 rendezvous).  The data path has no
collective (independent QPs per rank); what needs covering is the sharding by rank and the
max-over-ranks / sum-over-ranks reductions that form the reported value."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_plumbing_gloo():
    env = dict(os.environ)
    env.pop("QPDO_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "tests", "_dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and r["tmax"] == 2.0
    assert r["tot"][0] == 30.0
    assert r["tot"][1] == 2 * r["seed0"] + 1          # rank 1 solves the next seed: a different instance
