"""Randomized parity sweep: many small seeded instances with varied shapes, densities, bound patterns and
settings, every one solved by the HIP path (through the C-ABI) and by the CPU oracle.  Bar as in
test_gpu_parity.py: status, iteration and outer-iteration counts identical, iterates within the stated
tolerance.  The instances are cheap for the oracle (n <= 160), so the sweep runs in well under a minute."""
import numpy as np
import pytest

from helpers import close_vec, device_active_count_consistent, first_integer_mismatch, same_trace_counts
from oracle import binding as ob
from qpdo_amd import problems, solver

pytestmark = pytest.mark.gpu


def _instance(i):
    """deterministic variety: shape, density, equality rows, one-sided / free rows, settings"""
    rng = np.random.default_rng(9000 + i)
    n = int(rng.integers(2, 160))
    m = int(rng.integers(1, 320))
    dens = float(rng.choice([0.02, 0.05, 0.1, 0.3, 0.8]))
    neq = int(rng.integers(0, min(n, m) // 2 + 1)) if rng.random() < 0.5 else 0
    p = problems.random_qp(7000 + i, n, m, dens, neq)
    kind = rng.random(m)
    l, u = p["l"].copy(), p["u"].copy()
    l[(kind < 0.15) & (np.arange(m) >= neq)] = -np.inf          # upper-bounded only
    u[(kind > 0.85) & (np.arange(m) >= neq)] = np.inf           # lower-bounded only
    free = (kind > 0.45) & (kind < 0.5) & (np.arange(m) >= neq)
    l[free], u[free] = -np.inf, np.inf                           # free rows
    # the C API takes bounds already clipped to +-QPDO_INFTY, as the reference's front end does (interfaces/mex/qpdo.m:138-139)
    p["l"], p["u"] = np.maximum(l, -1e20), np.minimum(u, 1e20)
    st = {}
    r = rng.random()
    if r < 0.2: st["scaling"] = 0
    elif r < 0.3: st["scaling"] = 3
    if rng.random() < 0.2: st["proximal"] = 0 if n > 4 else 1
    if rng.random() < 0.2: st["eps_abs"] = float(rng.choice([1e-4, 1e-8]))
    if rng.random() < 0.15: st["reset_newton_iter"] = int(rng.integers(1, 6))
    if rng.random() < 0.15: st["inner_max_iter"] = int(rng.integers(2, 12))
    if rng.random() < 0.15: st["sigma_init"] = float(rng.choice([1e-1, 1e-5]))
    if rng.random() < 0.15: st["mu_min"] = float(rng.choice([1e-6, 1e-12]))
    if rng.random() < 0.1: st["max_iter"] = int(rng.integers(3, 40))
    # Exotic settings (inner_max_iter of a few passes, mu_min 1e-12) make some instances crawl for thousands of passes;
    # after that many passes of a barely contracting iteration the per-pass rounding differences have been amplified
    # to the point where a termination test can flip (seen: 2082 vs 2040 passes).  The sweep bounds every run.
    st.setdefault("max_iter", 300)
    return p, st


@pytest.mark.parametrize("mode", ["dense", "pcg", "pcg-slab"])
def test_randomized_sweep_matches_oracle(mode, gpu_required, monkeypatch):
    monkeypatch.setenv("QPDO_LINSOLVE", "dense" if mode == "dense" else "pcg")
    if mode == "pcg-slab":
        monkeypatch.setenv("QPDO_SPMV", "slab")
    # stated tolerance of the sweep: 1e-7.  Without the proximal term or with tiny mu_min the Newton systems reach
    # condition numbers of 1e10+, which amplifies the (order-of-summation) rounding differences between any two
    # implementations; both device solvers differ from the oracle by the same 7e-9 on the worst instance.
    rtol = 1e-7
    bad = []
    for i in range(120):
        p, st = _instance(i)
        o = ob.OracleSolver(p, ob.default_settings(**st))
        ro = o.solve()
        oi = dict(ro["info"]); ox, oy = np.array(ro["x"]), np.array(ro["y"])
        to = o.trace()
        o.close()
        r = solver.solve_problem(p, verbose=0, **st)
        gi = r["info"]
        same = (gi["status_val"] == oi["status_val"] and gi["iterations"] == oi["iterations"] and gi["oterations"] == oi["oterations"])
        # per pass: kind, n_active, n_enter, n_leave, factor branch.  Not for runs cut off by max_iter: those are the crawling
        # instances of the comment in _instance() (inner_max_iter of 3-4 passes with mu_min = 1e-12), where after a few hundred
        # barely contracting passes a row sitting exactly on a bound flips sides between any two implementations (seen on
        # 3 of the 120 instances, identically for the dense and both PCG solvers) -- they are compared through their counts
        if oi["status_val"] != -5:
            same = same and same_trace_counts(r["trace"], to)
        elif same and not same_trace_counts(r["trace"], to):
            # (round 5) ... but not unconditionally: at the first pass whose integers differ the pass kinds must still agree (a Newton
            # pass on both sides) and the device's count must follow from the device's own w of that pass (helpers.py: what was measured on
            # these runs, and why no margin bound is asserted)
            k = first_integer_mismatch(r["trace"], to)
            same = k is not None and k < len(to) and k < len(r["trace"]) and int(r["trace"][k]["kind"]) == int(to[k]["kind"]) == 0 \
                and device_active_count_consistent(p, st, k)
        # a run stopped by max_iter is compared through its counts only (its iterate is mid-flight, not a solution)
        if same and oi["status_val"] not in (-3, -4, -5):
            same = close_vec(r["x"], ox, rtol) and close_vec(r["y"], oy, rtol)
        if not same:
            bad.append((i, p["n"], p["m"], st, oi["status_val"], gi["status_val"], oi["iterations"], gi["iterations"]))
    assert not bad, bad


def test_pcg_without_proximal_term_falls_back_to_the_dense_solver_not_to_an_error(gpu_required, monkeypatch):
    """Instance 1358 of the wider one-off sweep (profiles/r02_parity_sweep_more2.txt; proximal = 0, inner_max_iter = 2): without the
    proximal term Q + A'DA is singular to working precision there, Jacobi-PCG stagnates above 1e-8 and round 2 ended the solve with
    QPDO_ERROR (-99) where the reference -- a direct factorization, cholmod_interface.c:35-52 -- runs on to max_iter (-5).  The pass is
    now redone by the dense solver and the outcome must be the oracle's."""
    monkeypatch.setenv("QPDO_LINSOLVE", "pcg")
    p, st = _instance(1358)
    assert st.get("proximal") == 0
    o = ob.OracleSolver(p, ob.default_settings(**st)); ro = o.solve(); o.close()
    r = solver.solve_problem(p, verbose=0, **st)
    gi, oi = r["info"], ro["info"]
    assert oi["status_val"] == -5
    assert (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
    assert r["stats"]["pcg_dense_fallbacks"] >= 1


@pytest.mark.parametrize("kernel", ["wide", "lat"])
def test_fused_batch_sweep_is_bit_identical(kernel, gpu_required, monkeypatch):
    """the one-workgroup-per-QP kernel on the same 120 varied instances (one launch): operation order equals the
    oracle's, so x, y, objective and residual norms must match bit for bit (NaN-filled outputs for the certificates).
    Both launch shapes of a batch: "wide" (default: two workgroups per CU, work vectors in global memory) and "lat"
    (QPDO_SMALL_BATCH_KERNEL=lat: the latency kernel qpdo_solve uses, one workgroup per CU, work vectors in LDS)."""
    monkeypatch.setenv("QPDO_SMALL_BATCH_KERNEL", kernel)
    probs = [_instance(i)[0] for i in range(120)]
    res, failed = solver.solve_batch(probs, verbose=0, max_iter=300)
    assert failed == 0
    bad = []
    for i, (p, r) in enumerate(zip(probs, res)):
        o = ob.OracleSolver(p, ob.default_settings(max_iter=300))
        ro = o.solve()
        oi, ox, oy = dict(ro["info"]), np.array(ro["x"]), np.array(ro["y"])
        o.close()
        gi = r["info"]
        ok = (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
        if ok and oi["status_val"] not in (-3, -4):
            ok = np.array_equal(r["x"], ox) and np.array_equal(r["y"], oy) and gi["objective"] == oi["objective"] \
                and gi["res_prim_norm"] == oi["res_prim_norm"] and gi["res_dual_norm"] == oi["res_dual_norm"]
        if not ok:
            bad.append((i, p["n"], p["m"], oi["status_val"], gi["status_val"], oi["iterations"], gi["iterations"]))
    assert not bad, bad


def test_fused_batch_takes_q_in_lower_upper_or_full_storage(gpu_required):
    """qpdo_amd_solve_batch converts every item's CSC matrices into the kernel's CSR images straight into its staging buffer: Q stored as
    its lower triangle (stype -1, what the reference's mex passes), its upper triangle (+1) or in full (0) must give the same image, i.e.
    the same bits out of the kernel"""
    import scipy.sparse as sp
    base = [_instance(i)[0] for i in range(12)]
    def variant(p, st):
        q = dict(p)
        L = sp.csc_matrix(sp.tril(p["Q"]))
        if st == -1: q["Q"] = L
        elif st == 1: q["Q"] = sp.csc_matrix(L.T)
        else:
            F = (L + sp.tril(L, -1).T).tocsc(); F.sort_indices(); q["Q"] = F
        q["Qstype"] = st
        return q
    out = {}
    for st in (-1, 1, 0):
        res, failed = solver.solve_batch([variant(p, st) for p in base], verbose=0, max_iter=300)
        assert failed == 0
        out[st] = res
    for a, b, c in zip(out[-1], out[1], out[0]):
        for r in (b, c):
            assert (r["info"]["status_val"], r["info"]["iterations"]) == (a["info"]["status_val"], a["info"]["iterations"])
            assert np.array_equal(r["x"], a["x"], equal_nan=True) and np.array_equal(r["y"], a["y"], equal_nan=True)
            assert r["info"]["objective"] == a["info"]["objective"] or (np.isnan(r["info"]["objective"]) and np.isnan(a["info"]["objective"]))


def test_fused_batch_degenerate_shapes_are_bit_identical(gpu_required):
    """m = 0 (no constraints), n = 1, a single constraint, a row without entries: the fused kernel's sort layouts, wave folds, quartet
    assembly and one-wave solves at their smallest sizes, against the oracle bit for bit"""
    import scipy.sparse as sp
    probs = []
    p = problems.random_qp(52, 80, 1, 0.1)
    p["A"] = sp.csc_matrix((0, 80)); p["l"] = np.zeros(0); p["u"] = np.zeros(0); p["m"] = 0
    probs.append(p)
    probs.append(problems.random_qp(27, 1, 5, 1.0, 0))
    probs.append(problems.random_qp(26, 64, 1, 0.2, 0))
    probs.append(problems.random_qp(28, 65, 3, 0.2, 1))
    q = problems.random_qp(29, 30, 12, 0.2, 0)
    A = sp.lil_matrix(q["A"]); A[4, :] = 0.0; q["A"] = sp.csc_matrix(A); q["A"].eliminate_zeros()      # an empty row
    probs.append(q)
    res, failed = solver.solve_batch(probs, verbose=0, max_iter=500)
    assert failed == 0
    for p, r in zip(probs, res):
        o = ob.OracleSolver(p, ob.default_settings(max_iter=500))
        ro = o.solve()
        oi = dict(ro["info"])
        o.close()
        gi = r["info"]
        assert (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"]), (p["n"], p["m"])
        if oi["status_val"] not in (-3, -4):
            assert np.array_equal(r["x"], ro["x"]) and np.array_equal(r["y"], ro["y"]) and gi["objective"] == oi["objective"], (p["n"], p["m"])


def test_batch_stream_api_edges(gpu_required, capfd):
    """qpdo_amd_batch_stream_*: an unknown or already collected ticket is an error (not a hang), an item that does not fit the fused
    kernel or invalid settings are refused at submit, a stream destroyed with a batch still in flight completes it first, and a slot
    becomes free again once its ticket has been waited for"""
    import ctypes as C
    L = solver.lib()
    probs = [problems.config_qp("C3", i) for i in range(8)]
    B = solver.Batch(probs)
    st = solver.BatchStream(depth=1)
    t = st.submit(B, verbose=0, max_iter=50)
    ks = C.c_double(0.0)
    assert L.qpdo_amd_batch_stream_wait(st._h, C.c_long(t + 5), C.byref(ks)) == -1          # unknown ticket
    res, _ = st.wait(t)
    assert len(res) == 8 and all(r["info"]["iterations"] > 0 for r in res)
    assert L.qpdo_amd_batch_stream_wait(st._h, C.c_long(t), C.byref(ks)) == -1              # already collected
    t2 = st.submit(B, verbose=0, max_iter=50)                                               # the slot is free again
    assert t2 == t + 1
    st.close()                                                                              # in flight: destroy completes it
    assert all(np.isfinite(x).all() for x, _ in B.outs)
    st = solver.BatchStream(depth=2)
    with pytest.raises(RuntimeError):
        st.submit(B, verbose=0, rho=2.0)                                                    # invalid settings
    big = solver.Batch([problems.random_qp(5, 1100, 40, 0.01)])                             # n > 1024: not for the fused kernel
    with pytest.raises(RuntimeError):
        st.submit(big, verbose=0)
    st.close()
    capfd.readouterr()


@pytest.mark.parametrize("n,m,dens,neq", [(60, 700, 0.1, 50), (100, 1000, 0.05, 0), (30, 1024, 0.2, 10), (250, 513, 0.03, 100), (40, 257, 0.2, 0), (8, 129, 0.5, 2), (700, 300, 0.01, 20), (1024, 64, 0.01, 0)])
def test_fused_batch_is_bit_identical_at_the_breakpoint_counts_that_change_the_sort_layout(n, m, dens, neq, gpu_required):
    """The breakpoint sort of the fused kernel holds 1, 2 or 4 elements per thread (2m <= 512, <= 1024, <= 2048) and its one-lane sums are
    folded by a wave 64 elements at a time: instances on both sides of every layout change (the sweep above stays below m = 320), three
    seeds each, must carry the oracle's bits.  The last two: the largest orders the kernel takes (factor in global memory, n up to 1024) --
    its dynamic LDS must still fit (a layout change in round 4 did not, and the batch quietly took the generic path)."""
    probs = [problems.random_qp(4400 + 7 * k + m, n, m, dens, neq) for k in range(3)]
    res, failed = solver.solve_batch(probs, verbose=0, max_iter=400)
    assert failed == 0
    for p, r in zip(probs, res):
        o = ob.OracleSolver(p, ob.default_settings(max_iter=400))
        ro = o.solve()
        oi = dict(ro["info"])
        o.close()
        gi = r["info"]
        assert (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
        if oi["status_val"] not in (-3, -4):
            assert np.array_equal(r["x"], ro["x"]) and np.array_equal(r["y"], ro["y"]) and gi["objective"] == oi["objective"]
            assert gi["res_prim_norm"] == oi["res_prim_norm"] and gi["res_dual_norm"] == oi["res_dual_norm"]


def test_streamed_overlapping_batches_are_bit_identical(gpu_required):
    """BASELINE.json configs[2] "streamed": three batches in flight at once on a batch stream (qpdo_amd_batch_stream_*) -- the
    same 120 varied instances split 50 / 40 / 30, the first with max_iter large enough that its slow items are still running
    while the next two are packed, uploaded and launched.  What else is in flight must not matter: every item carries the
    oracle's counts and the oracle's bits, exactly as in the one-launch sweep above; tickets can be waited for out of order."""
    probs = [_instance(i)[0] for i in range(120)]
    parts = [probs[:50], probs[50:90], probs[90:]]
    batches = [solver.Batch(pp) for pp in parts]
    st = solver.BatchStream(depth=3)
    tickets = [st.submit(b, verbose=0, max_iter=300) for b in batches]
    with pytest.raises(RuntimeError):
        st.submit(solver.Batch(probs[:2]), verbose=0)          # all three slots busy: refused, not blocked
    out = {}
    for k in (1, 0, 2):
        out[k], ks = st.wait(tickets[k])
        assert ks > 0.0
    res = out[0] + out[1] + out[2]
    # a slot is free again: a second round through the same stream, one batch alone
    # ... together with a twin of it (shared problem data, own output buffers), both in flight at once
    t, t2 = st.submit(batches[2], verbose=0, max_iter=300), st.submit(batches[2].twin(), verbose=0, max_iter=300)
    again, _ = st.wait(t)
    twin, _ = st.wait(t2)
    st.close()
    assert all(np.array_equal(a["x"], b["x"], equal_nan=True) and np.array_equal(a["y"], b["y"], equal_nan=True) for a, b in zip(twin, again))
    assert all(np.array_equal(a["x"], b["x"], equal_nan=True) and a["info"]["iterations"] == b["info"]["iterations"] for a, b in zip(again, out[2]))
    bad = []
    for i, (p, r) in enumerate(zip(probs, res)):
        o = ob.OracleSolver(p, ob.default_settings(max_iter=300))
        ro = o.solve()
        oi, ox, oy = dict(ro["info"]), np.array(ro["x"]), np.array(ro["y"])
        o.close()
        gi = r["info"]
        ok = (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
        if ok and oi["status_val"] not in (-3, -4):
            ok = np.array_equal(r["x"], ox) and np.array_equal(r["y"], oy) and gi["objective"] == oi["objective"] \
                and gi["res_prim_norm"] == oi["res_prim_norm"] and gi["res_dual_norm"] == oi["res_dual_norm"]
        if not ok:
            bad.append((i, p["n"], p["m"], oi["status_val"], gi["status_val"], oi["iterations"], gi["iterations"]))
    assert not bad, bad


@pytest.mark.parametrize("i", range(6))
def test_schur_mode_mid_size_instances_match_oracle(i, gpu_required, monkeypatch):
    """instances large enough (k >= 256 active rows) for the Schur-complement mode of the PCG, with equality rows,
    one-sided rows and perturbed settings, against the oracle's direct solves"""
    monkeypatch.setenv("QPDO_LINSOLVE", "pcg")
    monkeypatch.setenv("QPDO_PCG_SCHUR", "1")
    rng = np.random.default_rng(300 + i)
    n = int(rng.integers(500, 900)); m = int(rng.integers(900, 2000))
    neq = int(rng.integers(0, 120)) if i % 2 else 0
    p = problems.random_qp(8000 + i, n, m, float(rng.choice([0.02, 0.05])), neq)
    kind = rng.random(m)
    l, u = p["l"].copy(), p["u"].copy()
    l[(kind < 0.1) & (np.arange(m) >= neq)] = -1e20
    u[(kind > 0.9) & (np.arange(m) >= neq)] = 1e20
    p["l"], p["u"] = l, u
    st = [dict(), dict(scaling=0), dict(eps_abs=1e-8), dict(proximal=0), dict(sigma_init=1e-1), dict(scaling=3, mu_min=1e-6)][i]
    o = ob.OracleSolver(p, ob.default_settings(**st))
    ro = o.solve()
    oi, ox, oy = dict(ro["info"]), np.array(ro["x"]), np.array(ro["y"])
    to = o.trace()
    o.close()
    r = solver.solve_problem(p, verbose=0, **st)
    gi = r["info"]
    assert (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
    assert same_trace_counts(r["trace"], to)
    assert r["stats"]["schur_passes"] > 0
    assert close_vec(r["x"], ox, 1e-7) and close_vec(r["y"], oy, 1e-7)


@pytest.mark.parametrize("qdiag", [0.0, 1e-3])
def test_schur_mode_on_mid_size_lp_like_instances(qdiag, gpu_required, monkeypatch):
    """Q = 0 (an LP) and a weak diagonal Q at a size where the Schur-complement mode takes over: Dq is then only the proximal sigma
    (1e-3 down to 1e-7), so the inner system A_c A_c' / sigma + D^-1 and the absolute stopping rule work at their extremes"""
    import scipy.sparse as sp
    monkeypatch.setenv("QPDO_LINSOLVE", "pcg")
    monkeypatch.setenv("QPDO_PCG_SCHUR", "1")
    p = problems.random_qp(8101, 700, 1500, 0.03)
    p["Q"] = sp.diags(np.full(700, qdiag)).tocsc() if qdiag > 0 else sp.csc_matrix((700, 700))
    o = ob.OracleSolver(p, ob.default_settings(max_iter=600))
    ro = o.solve()
    oi, ox, oy, to = dict(ro["info"]), np.array(ro["x"]), np.array(ro["y"]), o.trace()
    o.close()
    r = solver.solve_problem(p, verbose=0, max_iter=600)
    gi = r["info"]
    assert (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
    assert same_trace_counts(r["trace"], to)
    if oi["status_val"] == 1:
        assert close_vec(r["x"], ox, 1e-7) and close_vec(r["y"], oy, 1e-7)


def test_whole_c3_batch_of_4096_at_reference_default_settings(gpu_required):
    """BASELINE.json configs[2] at FULL size through the path bench.py measures: 4096 MPC-sized QPs (n = 120, m = 360, 120 equality rows),
    one fused-kernel launch, the reference's default settings (max_iter = 10000: the handful of instances that stall just above eps_abs in
    the reference algorithm itself run all 10000 passes).  Every item: a status the reference can return, and for solved items the
    independently recomputed KKT residuals within eps_abs and equal to the reported norms; a random 64 of them (plus every item that did
    NOT end solved, up to 8): status, counts and the oracle's bits."""
    from qpdo_amd.problems import config_qp, kkt_residuals
    count = 4096
    probs = [config_qp("C3", i) for i in range(count)]
    res, failed = solver.solve_batch(probs, verbose=0)
    assert failed == 0 and len(res) == count
    unsolved = []
    for i, (p, r) in enumerate(zip(probs, res)):
        gi = r["info"]
        assert gi["status_val"] in (1, -3, -4, -5), (i, gi["status_val"])
        if gi["status_val"] == 1:
            rp, rd = kkt_residuals(p, r["x"], r["y"])
            assert rp <= 1e-6 and rd <= 1e-6, (i, rp, rd)
            assert abs(rp - gi["res_prim_norm"]) <= 1e-9 and abs(rd - gi["res_dual_norm"]) <= 1e-9, (i, rp, rd, gi)
            assert gi["iterations"] < 10000
        else:
            unsolved.append(i)
    assert len(unsolved) <= 40, len(unsolved)                 # (12 on this generator: instances that stall at res_prim ~ 1.2e-6, in the oracle too)
    rng = np.random.default_rng(4096)
    pick = sorted(set(rng.choice(count, 64, replace=False).tolist()) | set(unsolved[:8]))
    bad = []
    for i in pick:
        o = ob.OracleSolver(probs[i], ob.default_settings())
        ro = o.solve()
        oi, ox, oy = dict(ro["info"]), np.array(ro["x"]), np.array(ro["y"])
        o.close()
        gi, r = res[i]["info"], res[i]
        ok = (gi["status_val"], gi["iterations"], gi["oterations"]) == (oi["status_val"], oi["iterations"], oi["oterations"])
        if ok and oi["status_val"] not in (-3, -4):
            ok = np.array_equal(r["x"], ox) and np.array_equal(r["y"], oy) and gi["objective"] == oi["objective"] \
                and gi["res_prim_norm"] == oi["res_prim_norm"] and gi["res_dual_norm"] == oi["res_dual_norm"]
        if not ok:
            bad.append((i, oi["status_val"], gi["status_val"], oi["iterations"], gi["iterations"]))
    assert not bad, bad
