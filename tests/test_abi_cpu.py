"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/*.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import has_gpu
from qpdo_amd import _build, problems, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qpdo_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_build.ensure_lib())
    api = declared_symbols("qpdo.h")
    ext = declared_symbols("qpdo_amd_ext.h")
    assert sorted(api) == sorted(solver.API_SYMBOLS)
    assert sorted(ext) == sorted(solver.EXT_SYMBOLS)
    for s in api + ext:
        assert hasattr(L, s), s


def test_default_settings_match_reference_constants():
    """reference include/constants.h:44-69"""
    s = solver.default_settings()
    exp = dict(max_time=1e20, max_iter=10000, inner_max_iter=1000, eps_abs=1e-6, eps_abs_in=1.0, eps_prim_inf=1e-6,
               eps_dual_inf=1e-6, rho=0.1, theta=0.25, delta=1e-2, mu_min=1e-9, proximal=1, sigma_init=1e-3,
               sigma_upd=1e-1, sigma_min=1e-7, scaling=10, verbose=1, print_interval=1, reset_newton_iter=1000)
    for k, v in exp.items():
        assert getattr(s, k) == v, k
    with pytest.raises(KeyError):
        solver.default_settings(not_a_setting=1)


def test_struct_layout_sizes():
    """x86-64 LP64 sizes of the public structs in the DLONG+PROFILING layout"""
    assert C.sizeof(solver.QPDOSettings) == 19 * 8
    assert C.sizeof(solver.QPDOInfo) == 2 * 8 + 32 + 8 + 8 * 8
    assert C.sizeof(solver.QPDOData) == 8 * 8
    assert C.sizeof(solver.CholmodSparse) == 8 * 8 + 6 * 4


def test_ctypes_mirror_has_the_reference_member_offsets():
    """qpdo_amd/solver.py mirrors include/qpdo.h by hand: pin every member offset the C caller's _Static_asserts pin
    (tests/abi_driver.c; reference include/types.h, DLONG + PROFILING layout)"""
    W = solver.QPDOWorkspace
    exp = dict(data=0, x=8, y=16, Ax=24, initialized=48, temp_m=56, mu=80, sqrt_mu_min=96, n_mu_changed=112, sigma=120, norm_q=136,
               xbar=144, dx=160, dy=168, tau=176, Qdx=184, w=208, linsys_rhs=272, res_prim_norm_old=280, ls_eta=296, ls_taus=328,
               eps_prim=360, eps_in=392, D_temp=400, chol=416, settings=424, scaling=432, solution=440, info=448, timer=456)
    for k, v in exp.items():
        assert getattr(W, k).offset == v, k
    assert C.sizeof(W) == 464
    I = solver.QPDOInfo
    for k, v in dict(iterations=0, oterations=8, status=16, status_val=48, res_prim_norm=56, objective=88, setup_time=96, solve_time=104,
                     run_time=112).items():
        assert getattr(I, k).offset == v, k
    S = solver.QPDOSettings
    for k, v in dict(max_time=0, max_iter=8, eps_abs=24, proximal=88, scaling=120, reset_newton_iter=144).items():
        assert getattr(S, k).offset == v, k


def test_setup_returns_null_on_invalid_input(capfd):
    p = problems.config_qp("C1b")
    with pytest.raises(RuntimeError):          # l > u: validate_data (reference src/validate.c:20-28)
        solver.QPDO().setup(p["Q"], p["q"], p["A"], p["u"] + 1.0, p["u"], Qstype=-1, verbose=0)
    with pytest.raises(RuntimeError):          # validate_settings
        solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0, rho=2.0)
    out = capfd.readouterr().out
    assert "Data validation returned failure" in out and "Settings validation returned failure" in out


@pytest.mark.skipif(has_gpu(), reason="only meaningful where no GPU exists")
def test_no_cpu_fallback_without_device(capfd):
    p = problems.config_qp("C1b")
    with pytest.raises(RuntimeError):
        solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
    assert "no HIP device" in capfd.readouterr().out


def test_batch_info_view_reads_the_item_array_in_place():
    """Batch.info_view(): one numpy view over the QPDOAmdBatchItem array the C side fills -- the same memory results() reads item by item"""
    B = solver.Batch([problems.config_qp("C3", i) for i in range(6)])
    for i in range(6):
        inf = B.items[i].info
        inf.iterations, inf.oterations, inf.status_val, inf.objective, inf.res_prim_norm = 10 + i, i, (1 if i % 2 else -5), 0.25 * i, 1e-7 * (i + 1)
    v, r = B.info_view(), B.results()
    assert len(v) == 6
    for i in range(6):
        for f in ("iterations", "oterations", "status_val", "objective", "res_prim_norm", "res_dual_norm", "solve_time"):
            assert v[f][i] == r[i]["info"][f]
    B.items[3].info.iterations = 77                       # a view, not a copy
    assert v["iterations"][3] == 77


def test_generator_is_deterministic_and_psd():
    a, b = problems.random_qp(3, 60, 90, 0.1), problems.random_qp(3, 60, 90, 0.1)
    assert (a["A"] != b["A"]).nnz == 0 and (a["Q"] != b["Q"]).nnz == 0 and np.array_equal(a["q"], b["q"])
    Q = problems.full_Q(a).toarray()
    assert np.allclose(Q, Q.T) and np.linalg.eigvalsh(Q).min() > 0
    assert (a["l"] <= a["u"]).all()
    c3 = problems.config_qp("C3")
    assert (c3["l"][:120] == c3["u"][:120]).all() and (c3["l"][120:] < c3["u"][120:]).all()


def test_problem_io_round_trip(tmp_path):
    from qpdo_amd import io
    p = problems.config_qp("C1b")
    s = solver.default_settings(verbose=0, max_iter=123)
    io.save_problem(tmp_path / "c1b.npz", p, s, result=dict(x=np.ones(p["n"]), y=np.zeros(p["m"]), info=dict(status_val=1, iterations=5, oterations=2)))
    q, st, res = io.load_problem(tmp_path / "c1b.npz")
    assert (q["A"] != p["A"]).nnz == 0 and (q["Q"] != p["Q"]).nnz == 0
    assert np.array_equal(q["q"], p["q"]) and np.array_equal(q["l"], p["l"]) and q["Qstype"] == -1
    assert st["max_iter"] == 123 and st["eps_abs"] == 1e-6 and res["info"]["iterations"] == 5


def test_mat_exchange_format_round_trip_and_selfcheck_fixtures(tmp_path):
    """The MATLAB-loadable twin of the format (io.save_mat / load_mat, scipy.io v5 files): full symmetric Q out, lower
    triangle back; settings struct; warm start; stored answer.  The committed ext_selfcheck_*.mat files must load and
    carry exactly what the oracle computes today on the loaded instance (they are oracle output, labelled as such)."""
    import glob
    from oracle import binding as ob
    from qpdo_amd import io
    p = problems.random_qp(11, 40, 70, 0.2, 8)
    st = {k: getattr(solver.default_settings(verbose=0, max_iter=77), k) for k in io.SETTING_NAMES}
    w = (np.arange(40) * 0.01, np.arange(70) * -0.02)
    io.save_mat(tmp_path / "a.mat", p, st, warm=w)
    q, s2, ref, w2 = io.load_mat(tmp_path / "a.mat")
    assert (q["A"] != p["A"]).nnz == 0 and (q["Q"] != p["Q"]).nnz == 0 and q["Qstype"] == -1
    assert np.array_equal(q["q"], p["q"]) and np.array_equal(q["l"], p["l"]) and np.array_equal(q["u"], p["u"])
    assert s2 == st and ref is None and np.array_equal(w2[0], w[0]) and np.array_equal(w2[1], w[1])
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ext_selfcheck_*.mat")))
    assert len(files) >= 5
    for f in files:
        q, s2, ref, w2 = io.load_mat(f)
        assert "NOT a CHOLMOD build" in ref["source"]
        o = ob.OracleSolver(q, ob.default_settings(**{k: v for k, v in s2.items()}))
        if w2 is not None:
            o.warm_start(*w2)
        r = o.solve()
        o.close()
        assert (r["info"]["status_val"], r["info"]["iterations"], r["info"]["oterations"]) == (ref["status_val"], ref["iterations"], ref["oterations"]), f
        assert np.array_equal(r["x"], ref["x"], equal_nan=True) and np.array_equal(r["y"], ref["y"], equal_nan=True), f


def test_header_compiles_from_c_and_layout_asserts_hold(tmp_path):
    """tests/abi_driver.c includes only include/qpdo.h; its _Static_asserts pin every struct member offset of the
    reference's DLONG + PROFILING layout (include/types.h); -Wall -Werror"""
    import subprocess
    exe = _build.build_abi_driver(str(tmp_path))
    assert os.path.exists(exe)
    if not has_gpu():      # with a GPU the full run is tests/test_gpu_abi.py
        out = subprocess.run([exe, "--no-device"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "qpdo_setup returned NULL" in out.stdout


def test_host_driver_under_address_and_ub_sanitizers(tmp_path):
    """the host path (qpdo_api.c: validation, the threaded CSC -> CSR conversions with both index widths
    and all three storages of Q, clean-up after a failed setup) under AddressSanitizer + UBSan + LeakSanitizer, driven by
    the compiled-C caller up to the point where the library finds no HIP device.  (GPU sanitizers do not exist on this
    pool; device code is covered by the parity tests.)"""
    import subprocess
    exe = _build.build_abi_driver(str(tmp_path), sanitize=True)
    # Sanitizers belong on the CPU build: hide every device from the child so that this test never initialises HIP under
    # ASan / LSan, also when the suite runs on a GPU box (the plain driver does the device run in tests/test_gpu_abi.py).
    # (Where /dev/kfd exists the HSA runtime still starts up to find that nothing is visible and leaks a few objects of its own:
    # leaks whose stack lies in the ROCm runtime libraries are not this library's and are suppressed by name.)
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libhsa-runtime64\nleak:libamdhip64\nleak:librccl\n")
    env = dict(os.environ, QPDO_SETUP_THREADS="5", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               LSAN_OPTIONS="suppressions=%s:print_suppressions=0" % supp,
               HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", CUDA_VISIBLE_DEVICES="-1")
    out = subprocess.run([exe, "--no-device"], capture_output=True, text=True, timeout=300, env=env)
    txt = out.stdout + out.stderr
    assert out.returncode == 0, txt[-3000:]
    assert "no-device mode: qpdo_setup returned NULL" in txt, txt[-3000:]
    assert "AddressSanitizer" not in txt and "runtime error" not in txt and "LeakSanitizer" not in txt, txt[-3000:]
    assert txt.count("random instance") == 4


def test_pass_decision_follows_the_reference_rules():
    """qpdo_amd/csrc/pass_decision.h through its exported wrapper: the ONE function behind the host loop's per-pass decision and the residual
    launch's own (launch-ahead route).  Reference: src/termination.c:11-23 (NON_CVX before SOLVED, strict > QPDO_INFTY, <= eps_abs),
    :28-30 (inner optimality), src/qpdo.c:361-363 (outer update: inner optimum AND a Newton step since the last one, or inner_max_iter
    passes), src/newton.c:21-33 (branch 0 / 1 / 2; QPDO_MAX_RANK_UPDATE = 100).  NaN norms compare false everywhere, as in the reference."""
    L = C.CDLL(_build.ensure_lib())
    f = L.qpdo_amd_pass_decision
    f.argtypes = [C.c_double] * 6 + [C.c_int] * 5 + [C.POINTER(C.c_int)] * 4

    def dec(rp, rd, rpi, rdi, eps_abs=1e-6, eps_in=1e-2, allow=1, force=0, reset=0, nact=5, nch=3):
        o = [C.c_int(-1) for _ in range(4)]
        assert f(rp, rd, rpi, rdi, eps_abs, eps_in, allow, force, reset, nact, nch, *[C.byref(v) for v in o]) == 0
        return tuple(v.value for v in o)           # ends_nc, ends_ok, outer, branch

    def ref(rp, rd, rpi, rdi, eps_abs=1e-6, eps_in=1e-2, allow=1, force=0, reset=0, nact=5, nch=3):
        nc = (rp > 1e20) or (rd > 1e20)
        ok = (not nc) and rp <= eps_abs and rd <= eps_abs
        outer = (allow and rpi <= eps_in and rdi <= eps_in) or force
        branch = 0 if ((reset and nact) or nch > 100) else (1 if nact else 2)
        return (int(nc), int(ok), int(bool(outer)), branch)

    nan, inf = float("nan"), float("inf")
    cases = [dict(rp=1.0, rd=1.0, rpi=1.0, rdi=1.0), dict(rp=1e-7, rd=1e-6, rpi=1.0, rdi=1.0), dict(rp=1e-6, rd=1.0000001e-6, rpi=1.0, rdi=1.0),
             dict(rp=1e21, rd=1e-9, rpi=0.0, rdi=0.0), dict(rp=1e20, rd=1e20, rpi=0.0, rdi=0.0), dict(rp=inf, rd=0.0, rpi=0.0, rdi=0.0),
             dict(rp=nan, rd=nan, rpi=nan, rdi=nan), dict(rp=nan, rd=1e-9, rpi=1e-3, rdi=1e-3),
             dict(rp=1.0, rd=1.0, rpi=1e-2, rdi=1e-2), dict(rp=1.0, rd=1.0, rpi=1e-2, rdi=1.0000001e-2), dict(rp=1.0, rd=1.0, rpi=1e-3, rdi=1e-3, allow=0),
             dict(rp=1.0, rd=1.0, rpi=1.0, rdi=1.0, force=1), dict(rp=1.0, rd=1.0, rpi=1e-3, rdi=1e-3, allow=0, force=1),
             dict(rp=1.0, rd=1.0, rpi=1.0, rdi=1.0, reset=1), dict(rp=1.0, rd=1.0, rpi=1.0, rdi=1.0, reset=1, nact=0),
             dict(rp=1.0, rd=1.0, rpi=1.0, rdi=1.0, nact=0, nch=0), dict(rp=1.0, rd=1.0, rpi=1.0, rdi=1.0, nch=100), dict(rp=1.0, rd=1.0, rpi=1.0, rdi=1.0, nch=101),
             dict(rp=1.0, rd=1.0, rpi=1.0, rdi=1.0, nact=0, nch=101), dict(rp=1e-9, rd=1e-9, rpi=0.0, rdi=0.0, force=1)]
    for c in cases:
        assert dec(**c) == ref(**c), c
    # spot values (not through the mirror above)
    assert dec(1e21, 0.0, 0.0, 0.0)[:2] == (1, 0)                 # NON_CVX wins over SOLVED
    assert dec(1e-6, 1e-6, 1.0, 1.0)[:2] == (0, 1)                # <= eps_abs
    assert dec(nan, nan, nan, nan) == (0, 0, 0, 1)                # a NaN iterate is neither ended nor an outer update: the loop goes on
    assert dec(1.0, 1.0, 1.0, 1.0, reset=1, nact=0)[3] == 2       # the reset flag alone does not force a factorization of an empty active set
