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


def test_batch_shard_split_is_disjoint_complete_and_order_stable():
    """pure host logic of the split: item b -> rank b mod world"""
    from qpdo_amd import solver
    for count in (0, 1, 7, 4096):
        for world in (1, 2, 3, 8):
            shards = [solver.shard_indices(count, r, world) for r in range(world)]
            flat = sorted(i for s in shards for i in s)
            assert flat == list(range(count))                       # complete, disjoint
            assert all(s == sorted(s) for s in shards)              # order-stable
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    import pytest
    with pytest.raises(ValueError):
        solver.shard_indices(4, 2, 2)
    with pytest.raises(ValueError):
        solver.merge_shards(3, [([0, 1], ["a", "b"])])               # item 2 missing
    with pytest.raises(ValueError):
        solver.merge_shards(2, [([0, 1], ["a", "b"]), ([1], ["c"])])  # item 1 twice
    assert solver.merge_shards(3, [([1], ["b"]), ([0, 2], ["a", "c"])]) == ["a", "b", "c"]


def test_two_rank_batch_shards_gloo():
    """world-2 gloo: each rank builds only its own items of a C3 batch (no GPU, no solve): the shards are disjoint,
    complete, in global order, and every rank generated exactly the instances it owns"""
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29579", os.path.join(ROOT, "tests", "_shard_worker.py"), "11"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == 2
    a, b = r["ranks"]
    assert a["indices"] == [0, 2, 4, 6, 8, 10] and b["indices"] == [1, 3, 5, 7, 9]
    assert a["made"] == a["indices"] and b["made"] == b["indices"]
    assert a["seeds"][0] != b["seeds"][0]


def _bare_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "QPDO_DEVICE")}
    return env


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher environment must start two ranks itself (before anything touches HIP), relay rank
    0's single JSON line and report n_gpus == 2 -- the shape of the driver's N = 1 command with N changed.  --launch-check keeps the
    ranks off the GPU: rendezvous, barriers and the max / sum reductions of the timed region only."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C1", "--steps", "3", "--warmup", "1", "--launch-check"],
                         env=_bare_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["sum_of_rank_plus_one"] == 3.0
    assert r["seconds_max_over_ranks"] >= 0.02                # rank 1 slept longer: the max over ranks, not rank 0's own time


def test_bench_gpus_n_refuses_fewer_devices_than_ranks():
    """no GPU in this container: a real (not --launch-check) run with --gpus 2 must exit non-zero with a message instead of
    reporting a one-GPU figure as n_gpus = 2 (or n_gpus = 1 for a command that asked for 2)"""
    env = dict(_bare_env(), HIP_VISIBLE_DEVICES="0")          # one device visible at most
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C1", "--steps", "1"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert "--gpus 2" in out.stderr and "visible" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_rank_count_must_match_gpus_flag():
    """a launcher that started 2 ranks for `--gpus 1` (or the reverse) is an error, not a silently different n_gpus"""
    env = dict(_bare_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "refusing" in out.stderr


def test_bench_adopts_the_launchers_world_size_when_gpus_is_left_at_its_default():
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py` WITHOUT --gpus reports n_gpus = 2 (the default adopts
    WORLD_SIZE); only an explicit --gpus that disagrees with the launcher is an error (the test above)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--workload", "C1", "--steps", "2", "--launch-check"]
    out = subprocess.run(cmd, env=_bare_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["sum_of_rank_plus_one"] == 3.0
