#!/usr/bin/env python3
"""bench.py -- headline benchmark: Newton iterations/sec (+ time-to-eps) of the HIP path on a seeded
random sparse QP, one process per GPU.

A "step" is one full cold-start qpdo_solve to eps_abs = 1e-6 (reference defaults, include/constants.h)
on the workload; `value` = Newton passes (loop passes that ran update_iterate) of all timed solves of all
ranks / wall time (max over ranks); `ms_per_step` = time-to-eps.  Inputs are resident in HBM before
the timed region starts (qpdo_setup uploads, converts and scales them; that time is reported separately as
setup_s).  N > 1: independent QPs (different seeds) per rank, no data-path collective ("weak").

Adds to the JSON line:
  roofline     -- HBM roofline of the dominant kernel (the Q CSR SpMV inside PCG): algorithmic bytes
                  12 nnz + 4(rows+1) + 8 rows + 8 cols over the HIP-event duration sampled live in the timed solves.
  cpu_baseline -- the CPU oracle (a port; the reference needs CHOLMOD, absent here) timed on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--workload", default=os.environ.get("QPDO_BENCH_WORKLOAD", "C4"),
                    help="C4 (n=1e5,m=2e5,1%%: the config the metric is quoted on), C2, C1 ...")
    ap.add_argument("--max-time", type=float, default=float(os.environ.get("QPDO_BENCH_MAX_TIME", "0")),
                    help="settings.max_time per solve in seconds (0 = reference default, unlimited)")
    ap.add_argument("--partition", default=os.environ.get("QPDO_BENCH_PARTITION", "independent"), choices=["independent", "rows"],
                    help="N > 1: 'independent' = one QP per GPU, no collective (weak scaling, default); 'rows' = ONE QP whose "
                         "rows of A are partitioned over the GPUs with an RCCL all-reduce per A' product (strong scaling)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="bound of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the informational C2 / C3-batch measurements")
    ap.add_argument("--no-mixed-extra", action="store_true", help="skip the informational fp32-inner-preconditioner solve")
    return ap.parse_args()


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


def cpu_baseline(prob, seconds, cg_per_newton, cg_source):
    """Bounded sample on this host's cores: `seconds` of the oracle's Jacobi-PCG on the same matrices
    (scaling off: the CG iteration cost does not depend on the scaling values), single thread.  The oracle's
    solver is plain Jacobi-PCG, so its rate is converted with the iterations per Newton pass that algorithm needs
    (measured on the GPU with the same algorithm), not with the count of the GPU's Schur-complement mode."""
    from oracle import binding as ob
    s = ob.default_settings(scaling=0, max_time=seconds)
    o = ob.OracleSolver(prob, s, linsolve="pcg", pcg_tol=1e-12)
    o.set_deadline(seconds)
    t0 = time.time()
    o.solve()
    dt = time.time() - t0
    info = o.info()
    cg = info["lin_iters"]
    o.close()
    cg_rate = cg / dt if dt > 0 else 0.0
    newton_rate = cg_rate / cg_per_newton if cg_per_newton > 0 else None
    return dict(value=newton_rate, unit="newton_iters/s", cores=1, kind="port",
                sample=("%.1f s of the oracle's Jacobi-PCG on the same instance (scaling off, 1 thread): %d CG iterations "
                        "= %.3f CG it/s, divided by the %.1f Jacobi-CG iterations per Newton pass (%s); the reference's "
                        "own direct CHOLMOD path is not buildable here and would need ~80 GB / 3.3e14 flop per factor at C4"
                        % (dt, cg, cg_rate, cg_per_newton, cg_source)),
                cg_iters_per_s=cg_rate)


def main():
    a = parse()
    rank, world, dist = dist_setup(a.gpus)
    from qpdo_amd import problems, solver
    cfg = problems.CONFIGS[a.workload]
    t0 = time.time()
    rows_mode = (a.partition == "rows" and world > 1)
    prob = problems.config_qp(a.workload, index=0 if rows_mode else rank)
    t_gen = time.time() - t0
    if rows_mode:      # every rank holds the same instance; the library keeps its row slice on the GPU
        if solver.dist_config(rank, world, mode="rccl") != 0:
            raise RuntimeError("qpdo_amd_dist_config failed")
    st = dict(verbose=0)
    if a.max_time > 0:
        st["max_time"] = a.max_time
    t0 = time.time()
    s = solver.QPDO().setup(prob["Q"], prob["q"], prob["A"], prob["l"], prob["u"], Qstype=-1, **st)
    t_setup = time.time() - t0
    L = solver.lib()

    for _ in range(a.warmup):
        s.solve()
    L.qpdo_amd_sync(s._w)
    barrier(dist)
    t0 = time.time()
    newton = cg = 0
    iters = oters = 0
    statuses = []
    at_time = at_n = 0.0
    ac_time = ac_bytes = ac_n = 0.0
    schur_passes = 0
    for _ in range(a.steps):
        r = s.solve()
        stt = s.stats()
        newton += stt["newton_passes"]; cg += stt["lin_iters"]
        iters += r["info"]["iterations"]; oters += r["info"]["oterations"]
        statuses.append(r["info"]["status_val"])
        at_time += stt["spmv_Q_avg_s"] * stt["spmv_Q_samples"]; at_n += stt["spmv_Q_samples"]
        ac_time += stt["spmv_Ac_time_s"]; ac_bytes += stt["spmv_Ac_bytes"]; ac_n += stt["spmv_Ac_samples"]; schur_passes += stt["schur_passes"]
    L.qpdo_amd_sync(s._w)
    barrier(dist)
    dt = time.time() - t0
    dt_max = allreduce(dist, [dt], "max")[0]
    tot_newton, tot_cg = allreduce(dist, [newton, cg], "sum")
    if rows_mode:      # one QP: every rank counted the same passes
        tot_newton, tot_cg = tot_newton / world, tot_cg / world

    last = r
    rp, rd = problems.kkt_residuals(prob, last["x"], last["y"]) if last["info"]["status_val"] not in (-3, -4) else (None, None)
    # roofline of the dominant kernel, live HIP-event samples from the timed solves.  With the Schur-complement mode
    # of the PCG the bulk of the time is the inner solves' A_c product (k_spmv_slab<EpiSchurA>: the k active rows of A,
    # compact index space, k changes per pass, so bytes and time are summed over the samples); otherwise it is the Q
    # product of the PCG operator (k_spmv_slab<EpiPcgQ>).  A back-to-back micro-benchmark of the full-size kernel beside it.
    pmc = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_spmv_c4.json")) as fh:
            pmc = json.load(fh)
    except Exception:
        pass
    if ac_n > at_n and ac_time > 0:
        bench_t, full_bytes = s.bench_spmv(0, reps=20)
        achieved = ac_bytes / ac_time / 1e9
        roof = dict(bound="hbm", kernel="k_spmv_slab<EpiSchurA> (S'p = p/d + A_c t: the active rows of A in the pass's compact index space)",
                    achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=None,
                    alg_bytes_per_launch=ac_bytes / ac_n, avg_launch_s=ac_time / ac_n, samples=int(ac_n),
                    microbench_GBs=full_bytes / bench_t / 1e9, spmv_A_GBs=None, spmv_At_GBs=None)
        if a.workload == "C4":
            try:      # averaged over all real launches of this kernel in a profiled run of this same command
                with open(os.path.join(ROOT, "profiles", "r01_pmc_schur_inner_c4.json")) as fh:
                    pin = json.load(fh)
                roof["traffic"] = pin["k_spmv_slab<EpiSchurA> (A_c product)"]["traffic_bytes_avg"]
                roof["traffic_source"] = ("profiles/r01_pmc_schur_inner_c4.json: 2*FETCH_SIZE + WRITE_SIZE averaged over the real launches of this kernel "
                                          "in separate rocprofv3 --pmc passes over this command")
            except Exception:
                if pmc is not None:
                    ratio = pmc["A  (CSR m x n)"]["traffic_over_alg"]
                    roof["traffic"] = ratio * ac_bytes / ac_n
                    roof["traffic_source"] = "profiles/r01_pmc_spmv_c4.json ratio for the same kernel on the full A, applied to the average compact launch"
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
        # HBM bytes per launch from the PMC counters cannot be collected inside this process; the figure comes from
        # the committed rocprofv3 --pmc passes on the same workload and kernel (profiles/r01_pmc_spmv_c4.json), if present
        if pmc is not None and a.workload == "C4":
            roof["traffic"] = pmc["Q  (CSR n x n)"]["traffic_bytes"]
            roof["traffic_source"] = "profiles/r01_pmc_spmv_c4.json (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
    for which, key in ((0, "spmv_A_GBs"), (1, "spmv_At_GBs")):
        t_, b_ = s.bench_spmv(which, reps=20)
        roof[key] = b_ / t_ / 1e9
    out = None
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
                                       else "independent QPs per GPU, no collective")},
            "time_to_eps_s": dt_max / max(1, a.steps) if all(v == 1 for v in statuses) else None,
            "status_val": statuses, "iterations": iters, "oterations": oters, "newton_passes": tot_newton,
            "cg_iters": tot_cg, "kkt_prim": rp, "kkt_dual": rd,
            "setup_s": t_setup, "generate_s": t_gen,
            "roofline": roof,
        }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cgpn, src = (cg / newton) if newton else 0.0, "this GPU run"
        if schur_passes:      # the GPU count is not the CPU algorithm's: use the committed Jacobi-only GPU run of the same instance
            try:
                with open(os.path.join(ROOT, "profiles", "r01_c4_jacobi_reference.json")) as fh:
                    jr = json.load(fh)
                if jr.get("workload") == a.workload:
                    cgpn, src = jr["cg_iters"] / jr["newton_passes"], "profiles/r01_c4_jacobi_reference.json: deflated Jacobi-PCG on the GPU, same instance"
            except Exception:
                pass
        try:
            out["cpu_baseline"] = cpu_baseline(prob, a.cpu_seconds, cgpn, src)
        except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU line
            out["cpu_baseline"] = dict(value=None, unit="newton_iters/s", cores=1, kind="port", sample="failed: %r" % (e,))
    s.delete()
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
        # same process (configs[1]: one QP n=1e4, m=2e4; configs[2]: 4096 MPC-sized QPs through the fused batch kernel).
        try:
            p2 = problems.config_qp("C2")
            t0 = time.time(); s3 = solver.QPDO().setup(p2["Q"], p2["q"], p2["A"], p2["l"], p2["u"], Qstype=-1, verbose=0); ts = time.time() - t0
            t0 = time.time(); r3 = s3.solve(); L.qpdo_amd_sync(s3._w); dt3 = time.time() - t0
            st3 = s3.stats(); s3.delete()
            out["other_configs"] = {"C2": dict(workload="n=10000, m=20000, density 0.01, cold start, default settings", time_to_eps_s=dt3, setup_s=ts,
                                               newton_iters_per_s=st3["newton_passes"] / dt3, status_val=r3["info"]["status_val"],
                                               iterations=r3["info"]["iterations"], linsolve="dense-ldlt" if st3["linsolve"] == 1 else "pcg",
                                               factor_count=st3["factor_count"], lowrank_solves=st3["lowrank_solves"])}
            nb = 4096
            probs = [problems.config_qp("C3", i) for i in range(nb)]
            B = solver.Batch(probs)
            B.run(verbose=0, max_iter=300)                      # warm-up (device arena, code objects)
            t0 = time.time(); resb, failed = B.run(verbose=0, max_iter=300); dtb = time.time() - t0
            out["other_configs"]["C3_batch"] = dict(workload="%d QPs n=120, m=360 (120 equality rows), one fused-kernel launch, max_iter=300" % nb,
                                                    seconds=dtb, qps_per_s=nb / dtb, failed=failed,
                                                    newton_iters_per_s=sum(r_["info"]["iterations"] - r_["info"]["oterations"] for r_ in resb) / dtb,
                                                    solved=sum(r_["info"]["status_val"] == 1 for r_ in resb))
        except Exception as e:
            out.setdefault("other_configs", {})["error"] = repr(e)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
