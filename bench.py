#!/usr/bin/env python3
"""bench.py -- headline benchmark: Newton iterations/sec (+ time-to-eps) of the HIP path on seeded random
sparse QPs, one process per GPU.

Workloads (--workload; BASELINE.json configs):
  C4  one QP n=1e5, m=2e5, 1 % fill per GPU -- the configuration the metric is quoted on (default).  A "step" is one
      full cold-start qpdo_solve to eps_abs = 1e-6 with the reference's default settings.
  C2  one QP n=1e4, m=2e4, 1 % fill per GPU; same meaning of a step.
  C3  a batch of 4096 MPC-sized QPs (n=120, m=360, 120 equality rows) sharded over the ranks (rank r solves items
      r::N with ONE fused-kernel launch, no collective); a step is one pass over the whole batch.
`value` = Newton passes (loop passes that ran update_iterate) of all timed steps of all ranks / wall time (max over
ranks); `ms_per_step` = wall time per step (C4/C2: time-to-eps, inputs already in HBM; setup_s is reported beside it and
`time_to_eps_incl_setup_s` adds it, which is what the reference's info->run_time measures, src/qpdo.c:461-464).
N > 1: independent QPs (different seeds) per rank / disjoint shards of the batch: no data-path collective ("weak");
--partition rows: ONE QP with the rows of A partitioned over the ranks and RCCL all-reduces ("strong").
Launching N > 1: either `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's form: RANK /
WORLD_SIZE come from the launcher) or the bare `python bench.py --gpus N ...`: with no RANK / WORLD_SIZE in the environment the script
starts its N ranks itself (launch_ranks: before anything loads the library or touches HIP), relays rank 0's JSON line and the worst
exit code; fewer than N visible GPUs, or a rank count that differs from --gpus, is an error, never a silently different n_gpus.

Adds to the JSON line:
  roofline     -- HBM roofline of the dominant kernel: algorithmic bytes (12 nnz + 4(rows+1) + 8 rows + 8 cols) over the
                  HIP-event duration sampled live in the timed solves on the solver's stream.
  cpu_baseline -- the CPU oracle (a port: the reference needs CHOLMOD, absent here) MEASURED on this host, all cores of
                  the process's CPU share, on a bounded sample of the same workload (no constants from profiles/).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# N ranks on one node share its host cores: the OpenMP conversions of qpdo_setup and the generator take 1/N of them each (set before
# libgomp loads; untimed either way, but N x all cores oversubscribes the node while every rank sets up at once).
if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "OMP_NUM_THREADS" not in os.environ:
    try:
        _aff = len(os.sched_getaffinity(0))
    except Exception:
        _aff = os.cpu_count() or 1
    os.environ["OMP_NUM_THREADS"] = str(max(1, _aff // int(os.environ["WORLD_SIZE"])))

# The reference never resets info->status_val between solves (quirk Q1, src/qpdo.c:451-453, replicated by default): steps 2..K of a
# workspace would report the first step's status even if they ran out of iterations.  The benchmark switches the quirk off so that
# `status_val` of every step is that step's own outcome (and checks the reported norms against eps_abs besides).
os.environ["QPDO_FIX_STATUS_RESET"] = "1"

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s)
C3_COUNT = 4096           # BASELINE.json configs[2]
MFMA_PROFILE = "r02_c2_dense_mfma_util_wide.json"   # (re-collected whenever k_ldl_syrk changes; unchanged since round 2)
PMC_PROFILE = "r05_pmc_schur_inner_c4.json"     # HBM-traffic counters of the dominant kernel (tools/profile_round.sh writes it)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks of the job (default: WORLD_SIZE when a launcher started this process, otherwise 1)")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--workload", default=os.environ.get("QPDO_BENCH_WORKLOAD", "C4"),
                    help="C4 (n=1e5,m=2e5,1%%: the config the metric is quoted on), C2, C3 (sharded batch), C1 ...")
    ap.add_argument("--max-time", type=float, default=float(os.environ.get("QPDO_BENCH_MAX_TIME", "0")),
                    help="settings.max_time per solve in seconds (0 = reference default, unlimited)")
    ap.add_argument("--max-iter", type=int, default=0, help="settings.max_iter (0 = reference default 10000)")
    ap.add_argument("--partition", default=os.environ.get("QPDO_BENCH_PARTITION", "independent"), choices=["independent", "rows"],
                    help="N > 1: 'independent' = one QP per GPU, no collective (weak scaling, default); 'rows' = ONE QP whose "
                         "rows of A are partitioned over the GPUs with RCCL all-reduces (strong scaling)")
    ap.add_argument("--batch-count", type=int, default=C3_COUNT, help="C3: number of QPs in the whole batch")
    ap.add_argument("--stream-depth", type=int, default=int(os.environ.get("QPDO_BENCH_STREAM_DEPTH", "12")),
                    help="C3: batches in flight on the batch stream (1 = one launch at a time, each step waited for before the next)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="bound of the cpu_baseline sample of the main workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the informational C2 / C3-batch measurements")
    ap.add_argument("--no-mixed-extra", action="store_true", help="skip the informational fp32-inner-preconditioner solve")
    ap.add_argument("--launch-check", action="store_true",
                    help="rank plumbing only: rendezvous, barrier and the two reductions of the timed region, no GPU work (CPU tests of --gpus N)")
    ap.add_argument("--rows-extra-child", action="store_true", help=argparse.SUPPRESS)    # internal: see row_partition_extra
    a = ap.parse_args()
    a.gpus_explicit = a.gpus is not None
    if a.gpus is None:                                      # left at its default: adopt the launcher's world size
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    return a


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("QPDO_DEVICE", str(local))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # control plane only (barrier + max of the wall time): the data path has no collective
        dist_mod.init_process_group(backend=os.environ.get("QPDO_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
        dist = dist_mod
    return rank, world, dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def allreduce(dist, vals, op="sum"):
    if dist is None:
        return list(vals)
    import torch
    t = torch.tensor(list(vals), dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return t.tolist()


# ---- CPU baselines: the oracle timed on this host's cores, bounded samples ----------------------------------------
def host_cores():
    """CPU share of this process: min(affinity mask, cgroup quota); QPDO_BENCH_CORES overrides.  A GPU box shows all 256
    hardware threads in the affinity mask while a one-GPU lease is entitled to 16 of them: without a cgroup quota to
    read, a mask wider than 64 is taken to be such a shared host and the share of one GPU (16) is used."""
    env = os.environ.get("QPDO_BENCH_CORES")
    if env:
        return max(1, int(env))
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            if quota:
                break
        except Exception:
            pass
    if quota:
        return max(1, min(aff, int(quota + 0.5)))
    return aff if aff <= 64 else 16


def cpu_baseline_single(prob, seconds, mode, threads=None):
    """One QP, cold start, reference default settings, all cores (OpenMP in the oracle's products / factor).
    mode 'direct': the reference's own algorithm (natural-order LDL' of Q + sigma I + A'DA, cholmod_interface.c:35-52);
    mode 'pcg': Jacobi-PCG on the same operator, for sizes where the dense factor does not exist (C4: 80 GB).
    The sample is the first `seconds` of the solve; value = Newton passes COMPLETED in it / the time at which the
    last of them ended (early passes have the fewest active rows, so this flatters the CPU)."""
    from oracle import binding as ob
    cores = host_cores() if threads is None else int(threads)
    ob.set_threads(cores)
    t0 = time.time()
    o = ob.OracleSolver(prob, ob.default_settings(), linsolve="dense" if mode == "direct" else "pcg", pcg_tol=1e-12)
    t_setup = time.time() - t0
    o.set_deadline(seconds)
    t0 = time.time()
    o.solve()
    dt = time.time() - t0
    tr, info = o.trace(), o.info()
    o.close()
    finished = info["status_val"] not in (-6,)
    done = [t for t in tr if t["kind"] == 0 and (finished or mode == "direct" or t["t_end"] <= seconds)]
    t_last = done[-1]["t_end"] if done else dt
    cg = info["lin_iters"]
    what = ("natural-order dense LDL' (the reference's direct algorithm), blocked + OpenMP" if mode == "direct"
            else "Jacobi-PCG on the matrix-free Newton operator, OpenMP products (the reference's direct CHOLMOD path would need "
                 "an ~80 GB / 3.3e14-flop factor at this size)")
    return dict(value=(len(done) / t_last) if done else None, unit="newton_iters/s", cores=cores, kind="port",
                sample=("first %.1f s of a cold-start solve of the same instance with the oracle (%s), default settings, %d thread(s): "
                        "%d Newton passes completed by t=%.1f s%s; oracle setup %.1f s not included"
                        % (dt, what, cores, len(done), t_last, ("" if mode == "direct" else ", %d CG iterations (%.2f it/s)" % (cg, cg / dt)), t_setup)),
                newton_passes=len(done), seconds=t_last, solved_within_sample=bool(finished),
                **({} if mode == "direct" else dict(cg_iters_per_s=cg / dt if dt > 0 else None)))


def cpu_baseline_small(prob, threads, reps, **settings):
    """A QP the oracle solves in milliseconds (C1): setup + cold solve, `reps` times, the mean -- what the reference's info->run_time
    covers (src/qpdo.c:461-464)."""
    from oracle import binding as ob
    ob.set_threads(threads)
    ts = []
    for _ in range(reps):
        t0 = time.time()
        o = ob.OracleSolver(prob, ob.default_settings(**settings))
        o.solve()
        ts.append(time.time() - t0)
        info = o.info()
        o.close()
    sec = sum(ts) / len(ts)
    return dict(value=info["newton_passes"] / sec, unit="newton_iters/s", cores=int(threads), kind="port", seconds_per_solve=sec, best_seconds=min(ts),
                iterations=info["iterations"], status_val=info["status_val"],
                sample="setup + cold solve of the same instance by the oracle (dense natural-order LDL'), %d thread%s, mean of %d" % (threads, "" if threads == 1 else "s", reps))


def cpu_baseline_batch(probs, seconds, settings_over):
    """A slice of the batch through the oracle, one QP per host thread at a time (ctypes releases the GIL), all cores.
    Returns QPs/s and Newton passes/s over the QPs finished within `seconds`."""
    from oracle import binding as ob
    cores = host_cores()
    ob.set_threads(1)
    lock = threading.Lock()
    state = dict(next=0, qps=0, newton=0)
    t0 = time.time()

    def work():
        while True:
            with lock:
                i = state["next"]; state["next"] += 1
            if time.time() - t0 > seconds:
                return
            o = ob.OracleSolver(probs[i % len(probs)], ob.default_settings(**settings_over))
            o.solve()
            inf = o.info()
            o.close()
            with lock:
                state["qps"] += 1; state["newton"] += inf["newton_passes"]
    th = [threading.Thread(target=work) for _ in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.time() - t0
    return dict(value=state["newton"] / dt, unit="newton_iters/s", cores=cores, kind="port", qps_per_s=state["qps"] / dt,
                sample="%d solves over a %d-QP slice of the batch through the oracle (dense LDL' per QP, setup + solve), %d host threads, %.1f s" % (state["qps"], len(probs), cores, dt))


# ---- workloads -----------------------------------------------------------------------------------------------------
def run_batch(a, rank, world, dist):
    """C3: the batch sharded over the ranks (item b -> rank b mod world), one fused launch per rank and step."""
    import numpy as np
    from qpdo_amd import problems, solver
    count = a.batch_count
    st = dict(verbose=0)
    if a.max_iter > 0:
        st["max_iter"] = a.max_iter
    t0 = time.time()
    B = solver.shard_batch(count, rank, world, lambda i: problems.config_qp("C3", i))
    t_gen = time.time() - t0
    for _ in range(max(1, a.warmup)):          # at least one: device arena + code objects
        B.run(**st)
    # "streamed" (BASELINE.json configs[2]): with more than one step the steps go through a batch stream, up to --stream-depth of
    # them in flight (step s+1 is packed, uploaded and launched while the slowest workgroups of step s still run); every step is
    # still one pass over the whole batch and every item of every step is waited for inside the timed region.
    depth = max(1, min(a.stream_depth, a.steps))
    images = [B] + [B.twin() for _ in range(depth - 1)] if depth > 1 else [B]
    stream = solver.BatchStream(depth=depth) if depth > 1 else None
    if stream is not None:                     # warm the slots (arenas, pinned staging) outside the timed region
        for t in [stream.submit(img, **dict(st, max_iter=50)) for img in images]:
            stream.wait(t)
    barrier(dist)
    t0 = time.time()
    newton = solved = failed = 0
    kernel_s = 0.0
    t_submit = t_wait = 0.0                     # host seconds inside submit() (pack + upload + launch) and inside wait() (GPU + download)
    if stream is None:
        for _ in range(a.steps):
            _, f = B.run(results=False, **st)
            v = B.info_view()
            failed += f
            kernel_s += B.kernel_seconds
            newton += int((v["iterations"].astype(np.int64) - v["oterations"]).sum())
            solved += int((v["status_val"] == 1).sum())
    else:
        tickets = []
        def collect(tk):
            # every item of the step has been waited for when wait() returns; its outputs are in the image (x, y: img.outs; the info
            # fields: one numpy view over the item array -- building 4096 Python dicts per step cost 25 ms of the 89 ms step)
            nonlocal newton, solved, kernel_s, t_wait
            t, img = tk
            tw = time.time(); _, ks = stream.wait(t, results=False); t_wait += time.time() - tw
            v = img.info_view()
            kernel_s += ks
            newton += int((v["iterations"].astype(np.int64) - v["oterations"]).sum())
            solved += int((v["status_val"] == 1).sum())
        for k in range(a.steps):
            if len(tickets) == depth:
                collect(tickets.pop(0))
            ts = time.time(); tickets.append((stream.submit(images[k % depth], **st), images[k % depth])); t_submit += time.time() - ts
        while tickets:
            collect(tickets.pop(0))
    barrier(dist)
    dt = time.time() - t0
    if stream is not None:
        stream.close()
    dt_max = allreduce(dist, [dt], "max")[0]
    tot_newton, tot_solved, tot_failed, tot_items = allreduce(dist, [newton, solved, failed, len(B.indices) * a.steps], "sum")
    if rank != 0:
        return None
    cfg = problems.CONFIGS["C3"]
    out = {
        "metric": "primal-dual Newton iters/sec (+ time-to-eps) on random sparse QP", "value": tot_newton / dt_max, "unit": "newton_iters/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt_max / max(1, a.steps),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (seeded counter-based generator, qpdo_amd/csrc/qpdo_gen.c)",
        "config": {"workload": "C3: batch of %d MPC-sized QPs (n=%d, m=%d, %d equality rows, density %g) sharded over the GPUs "
                               "(item b -> GPU b mod N), one fused-kernel launch per GPU and step, %s, no collective; cold start, reference "
                               "default settings%s" % (count, cfg["n"], cfg["m"], cfg["n_eq"], cfg["density"],
                                                       ("steps streamed, up to %d in flight" % depth) if depth > 1 else "one step at a time",
                                                       (", max_iter=%d" % a.max_iter) if a.max_iter > 0 else " (max_iter=10000)"),
                   "count": count, "n": cfg["n"], "m": cfg["m"], "stream_depth": depth, "parallelism": "independent QPs, batch sharded over ranks, no collective"},
        "qps_per_s": tot_items / dt_max, "solved": tot_solved, "failed": tot_failed, "items": tot_items, "generate_s": t_gen,
        "kernel_s_per_step_rank0": kernel_s / max(1, a.steps),
        "host_submit_s_per_step_rank0": t_submit / max(1, a.steps), "host_wait_s_per_step_rank0": t_wait / max(1, a.steps),
        "roofline": small_kernel_roofline(newton / max(1, a.steps), kernel_s / max(1, a.steps), 1),       # rank 0's launch
    }
    if world == 1 and not a.no_cpu_baseline:
        sl = [problems.config_qp("C3", i) for i in range(min(count, 256))]
        out["cpu_baseline"] = cpu_baseline_batch(sl, min(a.cpu_seconds, 10.0), {k: v for k, v in st.items() if k != "verbose"})
    return out


def small_kernel_roofline(newton_passes, seconds, n_gpus):
    """seconds: HIP-event duration of the kernel launch.  What bounds k_small_solve: not HBM (the problem data once, ~70 KB per QP) and not LDS bandwidth
    (a few % of the aggregate), but the LATENCY of dependent, barrier-separated steps: a Newton pass walks about
    m + 2n + 55 of them (assembly row by row, one per factor column and one per column of the two triangular solves -- two columns
    share a barrier pair --, the bitonic sort stages);
    bit-identity with the oracle fixes the operation order that makes them dependent.  achieved = ns per step of one
    workgroup (two interleaved per CU), floor = barrier + LDS round trip + a dozen dependent operations."""
    n, m = 120, 360
    steps = m + 2 * n + 55
    wgs = 512 * max(1, n_gpus)                     # workgroups in flight: 2 per CU x 256 CUs per GPU
    ns_per_step = seconds * wgs / max(1.0, newton_passes) / steps * 1e9
    floor = 250.0
    lds_bytes = 8.0 * (n ** 3) + 8.0 * 4 * n * n + 16.0 * 2 * m * 55
    lds_peak = 256 * 128 * 2.4 * max(1, n_gpus)    # GB/s
    return dict(bound="latency", kernel="k_small_solve (one workgroup per QP, whole state in LDS)", achieved=ns_per_step, peak=floor,
                unit="ns per dependent step", frac=floor / ns_per_step if ns_per_step > 0 else None, traffic=None,
                steps_per_newton_pass=steps, ms_per_newton_pass_per_workgroup=ns_per_step * steps * 1e-6,
                lds_bandwidth_frac=(newton_passes * lds_bytes / seconds / 1e9) / lds_peak,
                note="latency-bound; HBM traffic is the problem data once, LDS bandwidth use is a few percent")


MFMA_PROFILE = "r05_c2_mid_factor_mfma_util.json"     # SQ_VALU_MFMA_BUSY_CYCLES of the dense factorization kernel (tools/profile_round.sh writes it)


def dense_roofline(n, t_f, chk, factor_count, onelaunch, load_profile):
    """fp64-MFMA roofline entry of the dense LDL' factorization: n^3 / 3 flops over the HIP-event time of one factorization (the
    workspace's final weights, timed back to back on the solver's stream).  The MFMA-busy share comes from the committed counter
    profile of this round's build when it is there."""
    tf = n ** 3 / 3.0 / t_f / 1e12
    roof = dict(bound="mfma", kernel=("k_mid_factor: the whole LDL' in one launch, one resident workgroup per 64 x 64 tile, v_mfma_f64_16x16x4_f64 products, "
                                      "flag hand-offs" if onelaunch else "dense LDL' factorization (k_ldl_syrk fp64 MFMA trailing update + diag / panel chain)"),
                achieved=tf, peak=78.6, unit="TFLOP/s", frac=tf / 78.6, traffic=None, factor_seconds=t_f, flops_per_factor=n ** 3 / 3.0,
                factor_count=factor_count, onelaunch_factors=onelaunch, solve_residual_check=chk,
                note="n^3/3 flops over the HIP-event time of one factorization")
    prof = load_profile(MFMA_PROFILE)
    ent = (prof or {}).get("kernels", {}).get("k_mid_factor")
    if ent is not None and onelaunch:
        roof["mfma_busy_pct_of_1024_simds"] = ent.get("mfma_util_pct_of_1024_simds")
        roof["mfma_busy_source"] = ("profiles/%s, collected at commit %s: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) summed over the %s "
                                    "k_mid_factor dispatches of a C2 bench run in a separate rocprofv3 --pmc pass" % (MFMA_PROFILE, prof.get("commit"), ent.get("dispatches")))
    return roof


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _kill_group(p):
    """ends exactly the process group this launcher created for child p (start_new_session=True), nothing else"""
    try:
        os.killpg(p.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass


def row_partition_extra(a, rank, world, dist):
    """N > 1, default (independent-QP) mode: a short measurement of north_star's OTHER multi-GPU mode in the same run --
    ONE instance of the workload (seed 0) with the rows of A partitioned over the ranks and an all-reduce of A'y per
    product (RCCL on the solver's stream; QPDO_BENCH_ROWS_BACKEND=host: torch.distributed on host buffers, for ranks that
    share a GPU) -- so that one driver SCALE run captures both modes.  Not part of `value`.
    The RCCL exchange has never run on more than one rank before the first multi-GPU node sees this code (the build pool has one
    GPU per box), so the measurement runs in a CHILD process per rank (`bench.py --rows-extra-child`, started after this rank has
    released its workspaces; the children rendezvous among themselves on a port rank 0 picks): a child that has not finished
    within QPDO_BENCH_ROWS_TIMEOUT seconds (default 300; the clock covers the child's interpreter start, generation and two setups) is killed by its pid / process group and reported as an error, while this
    process -- which never entered that collective -- prints the main line and leaves through its normal barrier with exit code 0."""
    limit = float(os.environ.get("QPDO_BENCH_ROWS_TIMEOUT", "300"))
    box = [_free_port() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    # The children form a process group of their OWN on a fresh port.  Under `python -m torch.distributed.run` every rank carries the
    # elastic agent's variables (TORCHELASTIC_USE_AGENT_STORE=True makes every rank, rank 0 included, a CLIENT of a store the agent
    # hosts -- on the fresh port nobody would): they are not inherited, nor is anything else the launcher set for its own rendezvous.
    drop = ("TORCHELASTIC_", "TORCH_NCCL_", "TORCH_DIST", "GROUP_", "ROLE_", "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "OMP_NUM_THREADS")
    env = {k: v for k, v in os.environ.items() if not k.startswith(drop)}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(box[0]), RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_RANK=os.environ.get("LOCAL_RANK", str(rank)))
    cmd = [sys.executable, os.path.abspath(__file__), "--rows-extra-child", "--gpus", str(world), "--workload", a.workload]
    t0 = time.time()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, cwd=ROOT)
    try:
        so, se = p.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        _kill_group(p)
        so, se = p.communicate()
        return dict(error="timed out after %.0f s: the child process of this rank was killed; the main measurement is unaffected" % limit,
                    stderr_tail=(se or "")[-400:])
    lines = [l for l in (so or "").splitlines() if l.startswith("{")]
    if rank != 0:
        return None
    if p.returncode != 0 or not lines:
        return dict(error="child exited with code %s" % p.returncode, stderr_tail=(se or "")[-600:])
    res = json.loads(lines[-1])
    res["wall_s_incl_process_start"] = time.time() - t0
    return res


def rows_extra_child(a):
    """body of `--rows-extra-child` (one process per rank, see row_partition_extra): the first `passes` loop passes of a cold solve
    of seed 0 (max_iter = passes), once to warm up and once timed, bracketed by barriers; max over ranks; rank 0 prints the JSON."""
    rank, world, dist = dist_setup(a.gpus)
    from qpdo_amd import problems, solver
    passes = int(os.environ.get("QPDO_BENCH_ROWS_PASSES", "16"))
    mode = os.environ.get("QPDO_BENCH_ROWS_BACKEND", "rccl")
    res = dict(workload="%s seed 0, rows of A partitioned over %d ranks, the first %d loop passes of a cold solve (max_iter=%d)" % (a.workload, world, passes, passes),
               backend="RCCL all-reduce on the solver's stream" if mode == "rccl" else "torch.distributed (gloo) on host buffers")
    t0 = time.time()
    prob = problems.config_qp(a.workload, index=0)
    t_generate = time.time() - t0
    times = []
    for rep in range(2):
        if solver.dist_config(rank, world, mode=mode) != 0:          # an RCCL unique id is one-shot: a fresh one per workspace
            raise RuntimeError("qpdo_amd_dist_config failed")
        t0 = time.time()
        s = solver.QPDO().setup(prob["Q"], prob["q"], prob["A"], prob["l"], prob["u"], Qstype=-1, verbose=0, max_iter=passes)
        t_setup = time.time() - t0
        barrier(dist)
        t0 = time.time()
        r = s.solve()
        solver.lib().qpdo_amd_sync(s._w)
        barrier(dist)
        times.append(time.time() - t0)
        stt = s.stats()
        s.delete()
    dt = allreduce(dist, [times[-1]], "max")[0]
    res.update(seconds=dt, setup_s=t_setup, generate_s=t_generate, newton_passes=stt["newton_passes"], newton_iters_per_s=stt["newton_passes"] / dt,
               iterations=r["info"]["iterations"], status_val=r["info"]["status_val"], cg_iters=stt["lin_iters"],
               collectives=stt["collectives"], inner_collectives=stt["inner_collectives"], inner_steps=stt["inner_steps"],
               inner_solves=stt["inner_solves"], schur_passes=stt["schur_passes"], scaling="strong")
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ---- --gpus N without a launcher: this process starts the N ranks itself ---------------------------------------------------
def visible_gpu_count():
    """GPUs this process may use.  The launcher must not create a HIP context (it only starts children), so it does not load
    libqpdo_amd: HIP_/ROCR_VISIBLE_DEVICES when set, otherwise torch.cuda.device_count() (a count, no context on this image)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def launch_ranks(a, argv):
    """`python bench.py --gpus N ...` with no RANK / WORLD_SIZE in the environment: start N copies of this script, one per GPU
    (RANK = LOCAL_RANK = r, rendezvous on 127.0.0.1 and a free port), relay rank 0's JSON line and return the worst exit code.
    Each child is its own session, so a hung run is ended by killing exactly the process groups created here.  Fewer than N
    visible GPUs is an error (QPDO_BENCH_SHARE_GPU=1: let the ranks share what there is -- tests on a one-GPU box)."""
    n = a.gpus
    if not a.launch_check and not os.environ.get("QPDO_BENCH_SHARE_GPU"):
        have = visible_gpu_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, %d visible: refusing to report an N-GPU figure from fewer devices\n" % (n, have))
            return 2
    limit = float(os.environ.get("QPDO_BENCH_LAUNCH_TIMEOUT", "3000"))
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=os.getcwd(), start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def _stop(signum, frame):
        for p in procs:
            _kill_group(p)
        sys.exit(128 + signum)
    signal.signal(signal.SIGTERM, _stop)
    signal.signal(signal.SIGINT, _stop)
    out0 = []
    rd = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    rd.start()
    t0 = time.time()
    worst, failed_at = 0, None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            worst = max((abs(c) for c in codes), default=0)
            break
        bad = [c for c in codes if c not in (None, 0)]
        if bad and failed_at is None:
            failed_at = time.time()
        if (failed_at is not None and time.time() - failed_at > 10.0) or time.time() - t0 > limit:
            why = "a rank failed" if failed_at is not None else "no result after %.0f s" % limit
            sys.stderr.write("bench.py: %s; ending the remaining ranks\n" % why)
            for p in procs:
                if p.poll() is None:
                    _kill_group(p)
            for p in procs:
                p.wait()
            worst = max([abs(c) for c in bad] + [1])
            break
        time.sleep(0.1)
    rd.join(5.0)
    for line in out0:
        sys.stdout.write(line)
    sys.stdout.flush()
    return worst


def main():
    a = parse()
    if a.rows_extra_child:
        return rows_extra_child(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))             # (nothing above this line loads the library or touches HIP)
    rank, world, dist = dist_setup(a.gpus)
    if world != max(1, a.gpus):                             # only an EXPLICIT --gpus can disagree with the launcher (the default adopts WORLD_SIZE)
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks: refusing to report an inconsistent n_gpus\n" % (a.gpus, world))
        sys.exit(2)
    if a.launch_check:
        # the control plane of a timed region without GPU work: barrier, max over ranks of a per-rank time, sum over ranks of a count
        barrier(dist)
        t0 = time.time()
        time.sleep(0.01 * (rank + 1))
        barrier(dist)
        dt_max = allreduce(dist, [time.time() - t0], "max")[0]
        tot = allreduce(dist, [float(rank + 1)], "sum")[0]
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "sum_of_rank_plus_one": tot,
                              "seconds_max_over_ranks": dt_max, "workload": a.workload}), flush=True)
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return
    from qpdo_amd import problems, solver
    if a.workload == "C3":
        out = run_batch(a, rank, world, dist)
        if rank == 0:
            print(json.dumps(out))
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return
    cfg = problems.CONFIGS[a.workload]
    t0 = time.time()
    rows_mode = (a.partition == "rows")
    # Independent QPs on N GPUs.  Default (round 4): EVERY rank solves the instance the N = 1 line is quoted on (seed offset 0), so the
    # work per GPU is exactly the N = 1 work and value(N) / (N value(1)) reads the system alone (host contention, clocks) -- "weak"
    # scaling in the strict sense.  QPDO_BENCH_DISTINCT=1: the eight instances of profiles/r04_c4_eight_instances.txt instead (seed
    # offset = rank; a cold solve takes 50-71 passes and 5.5-8.5 s depending on the seed, so each rank keeps K = min(N, 4) workspaces
    # -- seeds rank, rank+1, .. mod N; 20 GB of HBM each at C4 -- and step s solves workspace s mod K: a balanced sum over the timed
    # steps); either way no collective and no host traffic inside the timed region.
    distinct = os.environ.get("QPDO_BENCH_DISTINCT", "0") not in ("", "0")
    K = 1
    if not rows_mode:
        if os.environ.get("QPDO_BENCH_ROTATE"):
            K = max(1, int(os.environ["QPDO_BENCH_ROTATE"]))      # (tests of the rotation on one GPU; 1 switches it off)
            distinct = True
        elif world > 1 and distinct:
            K = min(world, 4)
    prob = problems.config_qp(a.workload, index=(rank if (distinct and not rows_mode) else 0))
    t_gen = time.time() - t0
    if rows_mode:      # every rank holds the same instance; the library keeps its row slice on the GPU
        # (N = 1: a forced single-rank RCCL communicator -- the partitioned code path, every collective through ncclAllReduce,
        # on one GPU: its time against the default path's is the overhead of the partitioned organisation itself)
        if solver.dist_config(rank, world, mode="rccl", force=(world == 1)) != 0:
            raise RuntimeError("qpdo_amd_dist_config failed")
    st = dict(verbose=0)
    if a.max_time > 0:
        st["max_time"] = a.max_time
    if a.max_iter > 0:
        st["max_iter"] = a.max_iter
    t0 = time.time()
    s = solver.QPDO().setup(prob["Q"], prob["q"], prob["A"], prob["l"], prob["u"], Qstype=-1, **st)
    t_setup = time.time() - t0
    L = solver.lib()
    ws = [(prob, s)]
    for j in range(1, K):
        pj = problems.config_qp(a.workload, index=(rank + j) % max(world, K))
        ws.append((pj, solver.QPDO().setup(pj["Q"], pj["q"], pj["A"], pj["l"], pj["u"], Qstype=-1, **st)))

    for i in range(a.warmup):
        ws[i % K][1].solve()
    for _, sj in ws:
        L.qpdo_amd_sync(sj._w)
    barrier(dist)
    t0 = time.time()
    newton = cg = 0
    iters = oters = 0
    statuses = []
    converged = []
    at_time = at_n = 0.0
    ac_time = ac_bytes = ac_n = 0.0
    schur_passes = 0
    for i in range(a.steps):
        prob_i, s_i = ws[i % K]
        r = s_i.solve()
        stt = s_i.stats()
        newton += stt["newton_passes"]; cg += stt["lin_iters"]
        iters += r["info"]["iterations"]; oters += r["info"]["oterations"]
        statuses.append(r["info"]["status_val"])
        converged.append(bool(r["info"]["res_prim_norm"] <= 1e-6 and r["info"]["res_dual_norm"] <= 1e-6))
        at_time += stt["spmv_Q_avg_s"] * stt["spmv_Q_samples"]; at_n += stt["spmv_Q_samples"]
        ac_time += stt["spmv_Ac_time_s"]; ac_bytes += stt["spmv_Ac_bytes"]; ac_n += stt["spmv_Ac_samples"]; schur_passes += stt["schur_passes"]
    for _, sj in ws:
        L.qpdo_amd_sync(sj._w)
    barrier(dist)
    dt = time.time() - t0
    prob_last = ws[(a.steps - 1) % K][0] if a.steps > 0 else prob
    for _, sj in ws[1:]:
        sj.delete()

    dt_max = allreduce(dist, [dt], "max")[0]
    tot_newton, tot_cg = allreduce(dist, [newton, cg], "sum")
    if rows_mode:      # one QP: every rank counted the same passes
        tot_newton, tot_cg = tot_newton / world, tot_cg / world

    last = r
    rp, rd = problems.kkt_residuals(prob_last, last["x"], last["y"]) if last["info"]["status_val"] not in (-3, -4) else (None, None)
    # roofline of the dominant kernel, live HIP-event samples from the timed solves.  With the Schur-complement mode
    # of the PCG the bulk of the time is the inner solves' A_c product (k_spmv_slab<EpiSchurW>: the k active rows of A,
    # compact index space, k changes per pass, so bytes and time are summed over the samples); otherwise it is the Q
    # product of the PCG operator (k_spmv_slab<EpiPcgQ>).  A back-to-back micro-benchmark of the full-size kernel beside it.
    def load_profile(name):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                return json.load(fh)
        except Exception:
            return None
    if rows_mode:
        # row-partitioned solve: the HIP-event samples of the inner products are not taken across collectives; the figure is the
        # back-to-back micro-benchmark of this rank's slice of A (the same slab kernel the inner products use)
        bench_t, full_bytes = s.bench_spmv(0, reps=20)
        roof = dict(bound="hbm", kernel="k_spmv_slab<EpiStore> on this rank's rows of A (micro-benchmark)", achieved=full_bytes / bench_t / 1e9, peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=full_bytes / bench_t / 1e9 / HBM_PEAK_GBS, traffic=None, alg_bytes_per_launch=full_bytes, avg_launch_s=bench_t,
                    spmv_A_GBs=None, spmv_At_GBs=None)
    elif s.stats()["linsolve"] == 1:
        # dense LDL' (C2): the dominant kernel is the fp64-MFMA trailing update; MFMA utilisation comes from the committed
        # PMC profile, the live figure here is the factor's flop rate from the solve's own statistics
        fc = s.stats()["factor_count"]
        t_f, chk = s.bench_dense_factor(reps=5)           # HIP events on the solver's stream, the factor of the final pass's weights
        tf = cfg["n"] ** 3 / 3.0 / t_f / 1e12
        roof = dense_roofline(cfg["n"], t_f, chk, fc, s.stats().get("onelaunch_factors", 0), load_profile)
    elif ac_n > at_n and ac_time > 0:
        bench_t, full_bytes = s.bench_spmv(0, reps=20)
        achieved = ac_bytes / ac_time / 1e9
        roof = dict(bound="hbm", kernel="k_spmv_slab<EpiSchurW> (w = u/d + A_c t, product 2 of S' u: the active rows of A in the pass's compact index space)",
                    achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=None,
                    alg_bytes_per_launch=ac_bytes / ac_n, avg_launch_s=ac_time / ac_n, samples=int(ac_n),
                    microbench_GBs=full_bytes / bench_t / 1e9, spmv_A_GBs=None, spmv_At_GBs=None)
        if a.workload == "C4":
            # HBM bytes per launch from the PMC counters cannot be collected inside this process: the figure is the average over
            # all real launches of this kernel in separate rocprofv3 --pmc passes over this same command (committed summary)
            pin = load_profile(PMC_PROFILE)
            used_profile = PMC_PROFILE
            if pin is None:                                      # (this round's collection not committed yet: the previous round's, under the same geometry check)
                used_profile = PMC_PROFILE.replace("r05_", "r04_")
                pin = load_profile(used_profile)
            ent = pin.get("k_spmv_slab<EpiSchurW>") if pin is not None else None
            live = ac_bytes / ac_n
            if ent is not None and ent.get("alg_bytes_per_launch_avg") and abs(ent["alg_bytes_per_launch_avg"] - live) <= 0.02 * live:
                roof["traffic"] = ent["traffic_bytes_avg"]
                # the bytes the kernel really moved over the live launch time, beside `achieved` (the ALGORITHMIC 12 B per nonzero over the
                # same time): the kernel streams 16-bit slab-local column indices, 10 B per nonzero, so this figure is the lower one
                roof["achieved_traffic_GBs"] = ent["traffic_bytes_avg"] / (ac_time / ac_n) / 1e9
                roof["achieved_traffic_frac_of_achievable_6300GBs"] = roof["achieved_traffic_GBs"] / 6300.0
                roof["traffic_age"] = pin.get("commit")
                roof["traffic_source"] = ("profiles/%s (collected at commit %s): 2*FETCH_SIZE + WRITE_SIZE averaged over the %d real launches of this "
                                          "kernel in separate rocprofv3 --pmc passes over this command; its launch geometry (%.1f MB algorithmic per "
                                          "launch) matches the live one (%.1f MB)" % (used_profile, pin.get("commit"), ent["real_launches"],
                                                                                   ent["alg_bytes_per_launch_avg"] / 1e6, live / 1e6))
            else:
                roof["traffic_source"] = ("none: profiles/%s is missing, or was collected on a build whose launches of this kernel moved a different "
                                          "number of algorithmic bytes (stale) -- re-collect with tools/profile_round.sh" % PMC_PROFILE)
        if at_n:
            q_t, q_b = s.bench_spmv(2, reps=5)
            roof["spmv_Q_live_GBs"] = q_b / (at_time / at_n) / 1e9
    else:
        bench_t, alg_bytes = s.bench_spmv(2, reps=20)
        live_t = at_time / at_n if at_n else bench_t
        achieved = alg_bytes / live_t / 1e9
        roof = dict(bound="hbm", kernel="k_spmv_slab<EpiPcgQ> (Kp = Q p + sigma p, full symmetric CSR n x n)", achieved=achieved, peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=None, alg_bytes_per_launch=alg_bytes,
                    avg_launch_s=live_t, samples=int(at_n), microbench_GBs=alg_bytes / bench_t / 1e9,
                    spmv_A_GBs=None, spmv_At_GBs=None)
        pmc = load_profile("r01_pmc_spmv_c4.json")
        if pmc is not None and a.workload == "C4":
            roof["traffic"] = pmc["Q  (CSR n x n)"]["traffic_bytes"]
            roof["traffic_source"] = "profiles/r01_pmc_spmv_c4.json (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
    if roof["bound"] == "hbm":
        for which, key in ((0, "spmv_A_GBs"), (1, "spmv_At_GBs")):
            t_, b_ = s.bench_spmv(which, reps=20)
            roof[key] = b_ / t_ / 1e9
    out = None
    all_solved = all(v == 1 for v in statuses) and (all(converged) or a.max_iter > 0 or a.max_time > 0)
    if rank == 0:
        out = {
            "metric": "primal-dual Newton iters/sec (+ time-to-eps) on random sparse QP",
            "value": tot_newton / dt_max if dt_max > 0 else 0.0,
            "unit": "newton_iters/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt_max / max(1, a.steps),
            "higher_is_better": True, "scaling": "strong" if rows_mode else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seeded counter-based generator, qpdo_amd/csrc/qpdo_gen.c)",
            "config": {"workload": "%s: one random sparse QP per GPU, n=%d, m=%d, density=%g, cold start, reference "
                                   "default settings (eps_abs=1e-6, scaling=10)%s" % (
                                       a.workload, cfg["n"], cfg["m"], cfg["density"],
                                       (", max_time=%gs" % a.max_time) if a.max_time > 0 else ""),
                       "n": cfg["n"], "m": cfg["m"], "density": cfg["density"], "linsolve": ("pcg: Jacobi + heavy-row deflation, Schur-complement mode on %d of %d Newton passes" % (schur_passes, newton)) if s.stats()["linsolve"] == 0 else "dense-ldlt",
                       "parallelism": ("one QP, rows of A partitioned over the GPUs, RCCL all-reduce of A'y" if rows_mode
                                       else ("independent QPs per GPU, no collective" + ("; each rank cycles through %d seeded instances, one cold solve per step" % K if K > 1
                                                                                         else ("; rank r solves the instance with seed offset r" if distinct else "; every rank solves the same instance (seed offset 0): the N = 1 work per GPU")))),
                       "instances_per_rank": K, "distinct_instances": bool(distinct and not rows_mode)},
            # time-to-eps: the solve alone (inputs resident in HBM), and with qpdo_setup added -- the reference's info->run_time
            # covers setup + solve (src/qpdo.c:461-464)
            "time_to_eps_s": dt_max / max(1, a.steps) if all_solved else None,
            "time_to_eps_incl_setup_s": (dt_max / max(1, a.steps) + t_setup) if all_solved else None,
            "status_val": statuses, "residuals_within_eps_abs": converged, "iterations": iters, "oterations": oters, "newton_passes": tot_newton,
            "cg_iters": tot_cg, "kkt_prim": rp, "kkt_dual": rd,
            "setup_s": t_setup, "generate_s": t_gen,
            "roofline": roof,
        }
    s.delete()
    if world > 1 and not rows_mode and not a.no_other_configs:
        rp_extra = row_partition_extra(a, rank, world, dist)            # every rank takes part; rank 0 reports
        if rank == 0:
            out.setdefault("other_configs", {})["row_partition"] = rp_extra
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline_single(prob, a.cpu_seconds, "direct" if cfg["n"] <= 20000 else "pcg")
        except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU line
            out["cpu_baseline"] = dict(value=None, unit="newton_iters/s", cores=host_cores(), kind="port", sample="failed: %r" % (e,))
        # the reference is single-threaded (src/qpdo.c has no threading; its only clock is util.c:245-264): the same sample on ONE thread
        try:
            one = cpu_baseline_single(prob, min(a.cpu_seconds, 15.0), "direct" if cfg["n"] <= 20000 else "pcg", threads=1)
            out["cpu_baseline"]["single_thread"] = {k: one.get(k) for k in ("value", "unit", "cores", "kind", "sample", "newton_passes", "seconds", "cg_iters_per_s") if k in one}
        except Exception as e:
            out["cpu_baseline"]["single_thread"] = dict(value=None, cores=1, sample="failed: %r" % (e,))
        if a.workload == "C4":
            # The sample above is the first ~25 s of the oracle's solve: its cheapest passes.  The WHOLE solve of this instance by the
            # oracle is on record (tests/golden/big_C4_full.npz, written by tests/golden/make_golden_big.py on the build container;
            # log: profiles/r03_oracle_C4_full_run.log): it is the figure that compares with time_to_eps_s.
            try:
                import numpy as np
                meta = json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", "big_C4_full.npz"))["meta"]))
                npass, sec = meta["info"]["newton_passes"], meta["oracle_seconds"]
                out["cpu_baseline"]["full_solve"] = dict(
                    value=npass / sec, unit="newton_iters/s", seconds=sec, newton_passes=npass, cg_iters=meta.get("oracle_lin_iters"),
                    cores=meta.get("threads"), kind="port", measured="offline, on the build container (not on this host, not in this run)",
                    note="the complete cold-start solve of this same instance by the oracle (Jacobi-PCG on the matrix-free Newton operator, compact "
                         "32-bit copies, %s threads): %d Newton passes in %.0f s (setup included); the late passes need 1000-29000 CG iterations each, "
                         "which is why the whole solve is ~75x slower per pass than the sampled early passes" % (meta.get("threads"), npass, sec))
            except Exception as e:
                out["cpu_baseline"]["full_solve"] = dict(error=repr(e))
    if rank == 0 and world == 1 and schur_passes and not a.no_mixed_extra:
        # Informational extra, NOT part of `value`: the same solve with the opt-in fp32 copy of the inner preconditioner's
        # matrix values (QPDO_PCG_INNER_F32=1; every vector, accumulation and the outer CG on the exact K stay fp64).
        try:
            os.environ["QPDO_PCG_INNER_F32"] = "1"
            s2 = solver.QPDO().setup(prob["Q"], prob["q"], prob["A"], prob["l"], prob["u"], Qstype=-1, **st)
            t0 = time.time(); r2 = s2.solve(); L.qpdo_amd_sync(s2._w); dt2 = time.time() - t0
            rp2, rd2 = problems.kkt_residuals(prob, r2["x"], r2["y"]) if r2["info"]["status_val"] == 1 else (None, None)
            out["extra_fp32_inner_preconditioner"] = dict(time_to_eps_s=dt2, newton_iters_per_s=s2.stats()["newton_passes"] / dt2,
                                                          status_val=r2["info"]["status_val"], iterations=r2["info"]["iterations"],
                                                          oterations=r2["info"]["oterations"], kkt_prim=rp2, kkt_dual=rd2,
                                                          max_abs_dx_vs_fp64_run=float(abs(r2["x"] - last["x"]).max()),
                                                          note="opt-in; not the measured configuration")
            s2.delete()
        except Exception as e:
            out["extra_fp32_inner_preconditioner"] = dict(error=repr(e))
        finally:
            os.environ.pop("QPDO_PCG_INNER_F32", None)
    if rank == 0 and world == 1 and a.workload == "C4" and not a.no_other_configs:
        # Informational extras, NOT part of `value`: the other single-GPU configurations of BASELINE.json, measured in the
        # same process (configs[1]: one QP n=1e4, m=2e4; configs[2]: 4096 MPC-sized QPs through the fused batch kernel), each
        # with its own measured CPU baseline
        try:
            # qpdo_setup once more in the same process (the first workspace is gone): what is left of setup_s once the code objects are loaded
            # and 20 GB of device memory have been touched; host part: CSC marshalling + OpenMP conversions on `host_threads` threads
            t0 = time.time(); sw = solver.QPDO().setup(prob["Q"], prob["q"], prob["A"], prob["l"], prob["u"], Qstype=-1, **st); out["setup_s_second_workspace"] = time.time() - t0
            sw.delete()
            out["host_threads"] = host_cores()
        except Exception as e:
            out["setup_s_second_workspace"] = None
        try:
            # BASELINE.json configs[0]: examples/demo_mex.m's shape (n = 200, m = 100, density 0.1), seeded generator, max_iter = 200 as in the demo
            pc1 = problems.config_qp("C1")
            sc1 = solver.QPDO().setup(pc1["Q"], pc1["q"], pc1["A"], pc1["l"], pc1["u"], Qstype=-1, verbose=0, max_iter=200)
            best1 = None
            for _ in range(5):
                t0 = time.time(); rc1 = sc1.solve(); d1 = time.time() - t0
                best1 = d1 if best1 is None or d1 < best1 else best1
            stc1 = sc1.stats()
            sc1.delete()
            c1 = dict(workload="n=200, m=100, density 0.1 (examples/demo_mex.m:7-9), cold start, default settings + max_iter=200", cold_solve_ms=1e3 * best1,
                      iterations=rc1["info"]["iterations"], oterations=rc1["info"]["oterations"], status_val=rc1["info"]["status_val"],
                      newton_iters_per_s=stc1["newton_passes"] / best1, factor_count=stc1["factor_count"], onelaunch_factors=stc1.get("onelaunch_factors"),
                      steps_launched_ahead=stc1.get("ahead_steps"), launched_ahead_steps_that_left=stc1.get("ahead_skips"),
                      route="fused one-launch kernel" if stc1["linsolve"] == 2 else "generic path, dense LDL' in one launch per factorization; Newton steps launched ahead of the host's decision")
            if not a.no_cpu_baseline:
                c1["cpu_baseline"] = cpu_baseline_small(pc1, host_cores(), 10, max_iter=200)
                c1["cpu_baseline"]["single_thread"] = cpu_baseline_small(pc1, 1, 10, max_iter=200)
            out.setdefault("other_configs", {})["C1"] = c1
        except Exception as e:
            out.setdefault("other_configs", {})["C1"] = dict(error=repr(e))
        try:
            p2 = problems.config_qp("C2")
            t0 = time.time(); s3 = solver.QPDO().setup(p2["Q"], p2["q"], p2["A"], p2["l"], p2["u"], Qstype=-1, verbose=0); ts = time.time() - t0
            s3.solve()                                                            # warm-up (code objects, allocations)
            t0 = time.time(); r3 = s3.solve(); L.qpdo_amd_sync(s3._w); dt3 = time.time() - t0
            st3 = s3.stats()
            t_f3, _ = s3.bench_dense_factor(reps=5) if st3["linsolve"] == 1 else (None, None)
            s3.delete()
            out.setdefault("other_configs", {})["C2"] = dict(workload="n=10000, m=20000, density 0.01, cold start, default settings", time_to_eps_s=dt3, setup_s=ts,
                                               roofline=(dense_roofline(10000, t_f3, None, st3["factor_count"], st3.get("onelaunch_factors", 0), load_profile) if t_f3 else None),
                                               hybrid_pcg_passes=st3.get("hybrid_pcg_passes"),
                                               newton_iters_per_s=st3["newton_passes"] / dt3, status_val=r3["info"]["status_val"],
                                               iterations=r3["info"]["iterations"], linsolve="dense-ldlt" if st3["linsolve"] == 1 else "pcg",
                                               factor_count=st3["factor_count"], lowrank_solves=st3["lowrank_solves"])
            if not a.no_cpu_baseline:
                out["other_configs"]["C2"]["cpu_baseline"] = cpu_baseline_single(p2, 20.0, "direct")
                try:
                    one2 = cpu_baseline_single(p2, 12.0, "direct", threads=1)
                    out["other_configs"]["C2"]["cpu_baseline"]["single_thread"] = {k: one2.get(k) for k in ("value", "unit", "cores", "kind", "sample", "newton_passes", "seconds")}
                except Exception as e:
                    out["other_configs"]["C2"]["cpu_baseline"]["single_thread"] = dict(value=None, cores=1, sample="failed: %r" % (e,))
            nb = C3_COUNT
            probs = [problems.config_qp("C3", i) for i in range(nb)]
            B = solver.Batch(probs)
            c3 = {}
            for label, kw in (("max_iter_default_10000", {}), ("max_iter_300", dict(max_iter=300))):
                B.run(verbose=0, **kw)                      # warm-up (device arena, code objects)
                import numpy as np
                t0 = time.time(); _, failed = B.run(verbose=0, results=False, **kw); dtb = time.time() - t0      # outputs: B.outs, B.info_view()
                vb = B.info_view()
                nwt = int((vb["iterations"].astype(np.int64) - vb["oterations"]).sum())
                c3[label] = dict(seconds=dtb, kernel_seconds=B.kernel_seconds, qps_per_s=nb / dtb, failed=failed, newton_iters_per_s=nwt / dtb,
                                 solved=int((vb["status_val"] == 1).sum()),
                                 roofline=small_kernel_roofline(nwt, B.kernel_seconds, 1))
            # streamed (configs[2]): 48 consecutive batches at the reference's default settings, up to 12 in flight (24 batches are two
            # fills of the pipeline: 38 k QP/s measured against 46 k with 96)
            try:
                depth_s, nb_s = 12, 48
                imgs = [B] + [B.twin() for _ in range(depth_s - 1)]
                stq = solver.BatchStream(depth=depth_s)
                for t_ in [stq.submit(img, verbose=0, max_iter=50) for img in imgs]:
                    stq.wait(t_)
                t0 = time.time(); tick = []; nwt = nsolved = 0; ksum = 0.0
                def _collect(tk_):
                    nonlocal nwt, nsolved, ksum
                    import numpy as np
                    t_, img_ = tk_
                    _, ks_ = stq.wait(t_, results=False)
                    v_ = img_.info_view()
                    ksum += ks_; nwt += int((v_["iterations"].astype(np.int64) - v_["oterations"]).sum()); nsolved += int((v_["status_val"] == 1).sum())
                for k_ in range(nb_s):
                    if len(tick) == depth_s:
                        _collect(tick.pop(0))
                    tick.append((stq.submit(imgs[k_ % depth_s], verbose=0), imgs[k_ % depth_s]))
                while tick:
                    _collect(tick.pop(0))
                dts = time.time() - t0
                stq.close()
                c3["streamed_max_iter_default_10000"] = dict(batches=nb_s, in_flight=depth_s, seconds=dts, qps_per_s=nb_s * nb / dts, newton_iters_per_s=nwt / dts,
                                                             solved=nsolved, kernel_seconds_summed=ksum,
                                                             note="qpdo_amd_batch_stream_*: %d consecutive batches of %d, every item waited for inside the timed region" % (nb_s, nb))
            except Exception as e:
                c3["streamed_max_iter_default_10000"] = dict(error=repr(e))
            c3["workload"] = "%d QPs n=120, m=360 (120 equality rows), one fused-kernel launch, wall time through the Python class" % nb
            if not a.no_cpu_baseline:
                c3["cpu_baseline"] = cpu_baseline_batch(probs[:256], 8.0, {})
            out["other_configs"]["C3_batch"] = c3
            # one small QP through the drop-in API (qpdo_solve = one launch of the fused kernel): cold-solve latency
            lat = {}
            for name_, pq, stq_ in (("KAT_2x3", problems.infeasibility_kat("degenerate"), dict(max_iter=100)), ("C1b_n50_m100", problems.config_qp("C1b"), {}),
                                    ("C3_size_n120_m360", problems.config_qp("C3", 0), {})):
                sq = solver.QPDO().setup(pq["Q"], pq["q"], pq["A"], pq["l"], pq["u"], Qstype=-1, verbose=0, **stq_)
                best = None
                for _ in range(5):
                    t0 = time.time(); rq = sq.solve(); dq = time.time() - t0
                    best = dq if best is None or dq < best else best
                stt_ = sq.stats()
                lat[name_] = dict(cold_solve_ms=1e3 * best, kernel_ms=1e3 * stt_["fused_kernel_s"], passes=rq["info"]["iterations"], status_val=rq["info"]["status_val"],
                                  route="fused one-launch kernel" if stt_["linsolve"] == 2 else "generic")
                sq.delete()
            out["other_configs"]["small_qp_latency"] = lat
        except Exception as e:
            out.setdefault("other_configs", {})["error"] = repr(e)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
