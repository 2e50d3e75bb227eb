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


def test_row_partitioned_solve_over_rccl_with_two_gpus(gpu_required):
    """The same five instances with the exchange over RCCL, one process per GPU (north star: "a single very large QP
    row-partitions A across GPUs with an RCCL all-reduce of A'y over xGMI").  Needs two GPUs: collected everywhere, skipped
    on a one-GPU box -- the day the suite runs on a multi-GPU node this is where ncclAllReduce first sees two ranks."""
    from qpdo_amd import solver
    if solver.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL cannot put two ranks on one device); this box has %d" % solver.device_count())
    env = dict(os.environ, QPDO_TEST_DIST_MODE="rccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29596", os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("[{")][-1])
    for r in res:
        assert r["ok_counts"] and r["err"] <= 1e-8 and r["linsolve"] == 0, r
    sch = [r for r in res if r["name"] == "schur"][0]
    assert sch["schur_passes"] > 0 and sch["inner_collectives"] == sch["inner_steps"] + sch["inner_solves"]


def test_row_partitioned_c4_matches_fixture(gpu_required):
    """BASELINE.json configs[3] at FULL size in its row-partitioned form (reference loop
    src/qpdo.c:343-449): n=1e5, m=2e5, 1 % fill through the partitioned code path -- unfused epilogues behind every
    exchange, partitioned k-vectors in the Schur-complement mode, every collective through ncclAllReduce on the solver's
    stream (forced single-rank communicator: the one GPU of the box) -- against the oracle's record of the WHOLE solve
    (tests/golden/big_C4_full.npz: 65 passes, 9 outer updates; big_C4_first40.npz if that file is absent): status and counts
    identical, per-pass integers identical, tau / norms / final iterate within the PCG tolerances, one collective per inner iteration."""
    code = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
for k in [k for k in os.environ if k.startswith("QPDO_")]: del os.environ[k]
os.environ["QPDO_DEVICE"] = "0"
import numpy as np
from helpers import assert_same_trace, trace_from_npz, close_vec, ITERATE_RTOL_PCG
from qpdo_amd import problems, solver
gold = os.path.join(%r, "tests", "golden")
z = np.load(os.path.join(gold, "big_C4_full.npz" if os.path.exists(os.path.join(gold, "big_C4_full.npz")) else "big_C4_first40.npz"))
meta = json.loads(str(z["meta"]))
assert solver.dist_config(0, 1, mode="rccl", force=True) == 0
p = problems.config_qp("C4", 0)
r = solver.solve_problem(p, verbose=0, **meta["settings"])
gi, oi = r["info"], meta["info"]
assert (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"]), gi
assert_same_trace(r["trace"], trace_from_npz(z), pcg=True)
assert close_vec(r["x"], z["x"], ITERATE_RTOL_PCG) and close_vec(r["y"], z["y"], ITERATE_RTOL_PCG)
st = r["stats"]
print(json.dumps(dict(ok=True, ex=float(np.abs(r["x"] - z["x"]).max()), ey=float(np.abs(r["y"] - z["y"]).max()),
                      stats={k: st[k] for k in ("linsolve", "schur_passes", "collectives", "inner_steps", "inner_solves", "inner_collectives")})))
""" % (ROOT, ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    st = r["stats"]
    assert r["ok"] and st["linsolve"] == 0 and st["schur_passes"] >= 30
    assert st["inner_collectives"] == st["inner_steps"] + st["inner_solves"] and st["collectives"] > st["inner_collectives"]


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


def test_bench_under_torch_distributed_run_with_two_ranks_on_one_gpu(gpu_required):
    """the DRIVER's form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py
    --gpus 2 ...` (static rendezvous: every rank carries TORCHELASTIC_USE_AGENT_STORE and friends).  The row-partition extra starts
    child processes that rendezvous among themselves; they must not inherit the elastic agent's variables (round 4: they did, every
    child became a client of a store nobody hosted, and the extra always ended in its timeout)."""
    import socket
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "QPDO_DEVICE")}
    noise = ("amdgpu.ids", "hostname of the client socket", "[Gloo]", "OMP_NUM_THREADS", "*****")
    # The port is probed and released before the launcher binds it: on a host that runs other jobs somebody else can take it in between.
    # ONLY that rendezvous failure is retried (once, on a fresh port); anything else fails with the ranks' whole stderr.
    rendezvous_trouble = ("EADDRINUSE", "Address already in use", "address already in use", "Connection refused", "connect() timed out",
                          "failed to connect", "RendezvousConnectionError", "client socket has timed out")
    for attempt in (0, 1):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                             env=dict(env, QPDO_BENCH_ROWS_BACKEND="host", QPDO_BENCH_ROWS_PASSES="8", QPDO_BENCH_SHARE_GPU="1", QPDO_DEVICE="0"))
        if out.returncode == 0 or attempt == 1 or not any(t in out.stderr for t in rendezvous_trouble):
            break
    if out.returncode != 0:
        err = "\n".join([l for l in out.stderr.splitlines() if l.strip() and not any(t in l for t in noise)][-120:])
        log_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(log_dir):
            with open(os.path.join(log_dir, "torchrun_bench_test_stderr.txt"), "w") as fh:
                fh.write(out.stdout[-4000:] + "\n---- stderr ----\n" + out.stderr)
        pytest.fail("bench.py under torch.distributed.run ended with code %d (attempt %d)\n%s\n---- stderr ----\n%s" % (out.returncode, attempt, out.stdout[-800:], err),
                    pytrace=False)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    rp = d["other_configs"]["row_partition"]
    assert "error" not in rp, rp
    assert rp["iterations"] == 8 and rp["collectives"] > 0


@pytest.mark.parametrize("workload,extra", [("C2", ["--steps", "3", "--warmup", "1", "--no-cpu-baseline"]),
                                            ("C3", ["--steps", "2", "--warmup", "1", "--max-iter", "300", "--batch-count", "512", "--no-cpu-baseline"])])
def test_bench_contract_with_two_ranks_on_one_gpu(workload, extra, gpu_required):
    """bench.py for N > 1, as the BARE command `python bench.py --gpus 2 ...` with no launcher environment: the script starts its own two
    ranks (here both land on the one GPU: QPDO_BENCH_SHARE_GPU=1; without it fewer than N visible devices is an error) and relays
    one JSON line from rank 0 with the whole-job aggregate; C2: independent QPs, every rank the instance of the N = 1 line;
    C3: the batch sharded over the ranks.  (The driver's torch.distributed.run form reaches the same code past the launcher.)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload] + extra
    # both ranks share the one GPU, where RCCL cannot form a 2-rank communicator: the row-partitioned extra of the default mode
    # exchanges through torch.distributed on host buffers here (a multi-GPU node uses RCCL)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "QPDO_DEVICE")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(env, QPDO_BENCH_ROWS_BACKEND="host", QPDO_BENCH_ROWS_PASSES="8", QPDO_BENCH_SHARE_GPU="1"))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["unit"] == "newton_iters/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    if workload == "C2":
        assert d["scaling"] == "weak" and d["config"]["instances_per_rank"] == 1 and not d["config"]["distinct_instances"] and all(v == 1 for v in d["status_val"])
        assert d["newton_passes"] > 2 * 3 * 30               # both ranks' passes are in the aggregate
        # north_star's second multi-GPU mode rides along in the same run: ONE instance, rows of A partitioned over the ranks
        rp = d["other_configs"]["row_partition"]
        assert "error" not in rp, rp
        assert rp["iterations"] == 8 and rp["seconds"] > 0 and rp["collectives"] > 0 and rp["newton_passes"] >= 6
    else:
        assert d["scaling"] == "strong" and d["items"] == 2 * 512 and d["failed"] == 0
