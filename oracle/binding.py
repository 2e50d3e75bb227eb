"""ctypes binding of the CPU oracle (oracle/qpdo_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  The product package (qpdo_amd) never
imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SETTINGS_FIELDS = [
    ("max_time", C.c_double), ("max_iter", C.c_int64), ("inner_max_iter", C.c_int64),
    ("eps_abs", C.c_double), ("eps_abs_in", C.c_double), ("eps_prim_inf", C.c_double),
    ("eps_dual_inf", C.c_double), ("rho", C.c_double), ("theta", C.c_double),
    ("delta", C.c_double), ("mu_min", C.c_double), ("proximal", C.c_int64),
    ("sigma_init", C.c_double), ("sigma_upd", C.c_double), ("sigma_min", C.c_double),
    ("scaling", C.c_int64), ("verbose", C.c_int64), ("print_interval", C.c_int64),
    ("reset_newton_iter", C.c_int64),
]


class OracleSettings(C.Structure):
    _fields_ = SETTINGS_FIELDS


class TraceRec(C.Structure):
    _fields_ = [("kind", C.c_int64), ("n_active", C.c_int64), ("n_enter", C.c_int64),
                ("n_leave", C.c_int64), ("factor_branch", C.c_int64), ("lin_iters", C.c_int64),
                ("tau", C.c_double), ("res_prim", C.c_double), ("res_dual", C.c_double),
                ("res_prim_in", C.c_double), ("res_dual_in", C.c_double),
                ("sigma", C.c_double), ("eps_in", C.c_double), ("t_end", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "qpdo_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
        L.oracle_default_settings.argtypes = [C.POINTER(OracleSettings)]
        L.oracle_setup.restype = C.c_void_p
        L.oracle_setup.argtypes = [C.c_int64, C.c_int64, ip, ip, dp, C.c_int, ip, ip, dp,
                                   dp, C.c_double, dp, dp, C.POINTER(OracleSettings)]
        L.oracle_set_linsolve.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int64]
        L.oracle_set_fix_status_reset.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_deadline.argtypes = [C.c_void_p, C.c_double]
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_get_threads.restype = C.c_int
        L.oracle_warm_start.argtypes = [C.c_void_p, dp, dp]
        L.oracle_solve.argtypes = [C.c_void_p]
        L.oracle_update_bounds.argtypes = [C.c_void_p, dp, dp]
        L.oracle_update_q.argtypes = [C.c_void_p, dp]
        L.oracle_update_settings.argtypes = [C.c_void_p, C.POINTER(OracleSettings)]
        L.oracle_info_i.restype = C.c_int64
        L.oracle_info_i.argtypes = [C.c_void_p, C.c_int]
        L.oracle_info_f.restype = C.c_double
        L.oracle_info_f.argtypes = [C.c_void_p, C.c_int]
        L.oracle_vec.restype = dp
        L.oracle_vec.argtypes = [C.c_void_p, C.c_int]
        L.oracle_trace.restype = C.POINTER(TraceRec)
        L.oracle_trace.argtypes = [C.c_void_p]
        L.oracle_cleanup.argtypes = [C.c_void_p]
        L.oracle_csc_mv.argtypes = [C.c_int64, C.c_int64, ip, ip, dp, C.c_int, C.c_int, dp, dp]
        L.oracle_vec_norm_inf.restype = C.c_double
        L.oracle_vec_norm_inf.argtypes = [dp, C.c_int64]
        L.oracle_vec_prod.restype = C.c_double
        L.oracle_vec_prod.argtypes = [dp, dp, C.c_int64]
        L.oracle_pwa_linesearch.restype = C.c_double
        L.oracle_pwa_linesearch.argtypes = [C.c_int64, C.c_double, C.c_double, dp, dp]
        L.oracle_K_apply.argtypes = [C.c_void_p, dp, dp, C.c_int]
        L.oracle_set_factor_state.argtypes = [C.c_void_p, C.c_double, dp]
        L.oracle_compact_ok.restype = C.c_int
        L.oracle_compact_ok.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def default_settings(**over):
    s = OracleSettings()
    lib().oracle_default_settings(C.byref(s))
    s.verbose = 0
    for k, v in over.items():
        if not hasattr(s, k):
            raise KeyError(k)
        setattr(s, k, v)
    return s


_VEC = dict(sol_x=0, sol_y=1, dx=2, dy=3, x=4, y=5, mu=6, D=7, E=8, q=9, l=10, u=11, Ax_vals=12,
            Qx_vals=13, Qx=14, Ax=15, Aty=16, d=17, Qdx=18, Adx=19, Atdy=20, res_prim_in=21,
            res_dual_in=22, ls_delta=23, ls_alpha=24, xbar=25, ybar=26, w=27)
_VLEN = dict(sol_x="n", sol_y="m", dx="n", dy="m", x="n", y="m", mu="m", D="n", E="m", q="n", l="m",
             u="m", Qx="n", Ax="m", Aty="n", d="m", Qdx="n", Adx="m", Atdy="n", res_prim_in="m",
             res_dual_in="n", ls_delta="2m", ls_alpha="2m", xbar="n", ybar="m", w="m")


class OracleSolver:
    """Workspace-style wrapper with the call sequence of the reference API."""

    def __init__(self, prob, settings=None, linsolve="dense", pcg_tol=0.0, pcg_maxit=0):
        """prob: dict with n, m, Q (scipy CSC, lower triangle or full), Qstype, A (scipy CSC),
        q, l, u, c."""
        L = lib()
        self.n, self.m = int(prob["n"]), int(prob["m"])
        Q, A = prob["Q"].tocsc(), prob["A"].tocsc()
        Q.sort_indices()
        A.sort_indices()
        self._keep = [np.ascontiguousarray(Q.indptr, np.int64), np.ascontiguousarray(Q.indices, np.int64),
                      np.ascontiguousarray(Q.data, np.float64),
                      np.ascontiguousarray(A.indptr, np.int64), np.ascontiguousarray(A.indices, np.int64),
                      np.ascontiguousarray(A.data, np.float64),
                      np.ascontiguousarray(prob["q"], np.float64), np.ascontiguousarray(prob["l"], np.float64),
                      np.ascontiguousarray(prob["u"], np.float64)]
        k = self._keep
        self.settings = settings if settings is not None else default_settings()
        self.h = L.oracle_setup(self.n, self.m, _ip(k[0]), _ip(k[1]), _dp(k[2]), int(prob.get("Qstype", -1)),
                                _ip(k[3]), _ip(k[4]), _dp(k[5]), _dp(k[6]), float(prob.get("c", 0.0)),
                                _dp(k[7]), _dp(k[8]), C.byref(self.settings))
        if self.h:
            L.oracle_set_linsolve(self.h, 0 if linsolve == "dense" else 1, float(pcg_tol), int(pcg_maxit))

    @property
    def ok(self):
        return bool(self.h)

    def warm_start(self, x=None, y=None):
        x = None if x is None else np.ascontiguousarray(x, np.float64)
        y = None if y is None else np.ascontiguousarray(y, np.float64)
        lib().oracle_warm_start(self.h, _dp(x), _dp(y))

    def solve(self):
        lib().oracle_solve(self.h)
        return self.result()

    def update_bounds(self, l=None, u=None):
        l = None if l is None else np.ascontiguousarray(l, np.float64)
        u = None if u is None else np.ascontiguousarray(u, np.float64)
        lib().oracle_update_bounds(self.h, _dp(l), _dp(u))

    def update_q(self, q):
        q = np.ascontiguousarray(q, np.float64)
        lib().oracle_update_q(self.h, _dp(q))

    def update_settings(self, settings):
        lib().oracle_update_settings(self.h, C.byref(settings))
        self.settings = settings

    def set_deadline(self, seconds):
        lib().oracle_set_deadline(self.h, float(seconds))

    def set_fix_status_reset(self, on):
        lib().oracle_set_fix_status_reset(self.h, int(on))

    def K_apply(self, v, sigma_f, d, compact):
        """K v = Q v + sigma_f v + A' (d o (A v)) through the PCG operator (compact=1: the 32-bit compact copies)"""
        v = np.ascontiguousarray(v, np.float64)
        d = np.ascontiguousarray(d, np.float64)
        out = np.zeros(self.n)
        lib().oracle_set_factor_state(self.h, float(sigma_f), _dp(d))
        lib().oracle_K_apply(self.h, _dp(v), _dp(out), int(compact))
        return out

    def compact_ok(self):
        return bool(lib().oracle_compact_ok(self.h))

    def vec(self, name):
        ln = _VLEN[name]
        size = {"n": self.n, "m": self.m, "2m": 2 * self.m}[ln]
        p = lib().oracle_vec(self.h, _VEC[name])
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(size,)).copy()

    def info(self):
        L = lib()
        names_i = ["iterations", "oterations", "status_val", "newton_passes", "lin_iters", "ntrace"]
        names_f = ["res_prim_norm", "res_dual_norm", "res_prim_in_norm", "res_dual_in_norm", "objective",
                   "setup_time", "solve_time", "run_time", "sigma", "tau", "scaling_c"]
        d = {k: int(L.oracle_info_i(self.h, i)) for i, k in enumerate(names_i)}
        d.update({k: float(L.oracle_info_f(self.h, i)) for i, k in enumerate(names_f)})
        return d

    def trace(self):
        n = int(lib().oracle_info_i(self.h, 5))
        t = lib().oracle_trace(self.h)
        out = []
        for i in range(n):
            r = t[i]
            out.append({f: getattr(r, f) for f, _ in TraceRec._fields_})
        return out

    def result(self):
        """Marshalling rule of the reference mex gateway (interfaces/mex/qpdo_mex.c:247-279)."""
        info = self.info()
        st = info["status_val"]
        nan_n, nan_m = np.full(self.n, np.nan), np.full(self.m, np.nan)
        res = dict(info=info, x=nan_n, y=nan_m, prim_inf_cert=nan_m.copy(), dual_inf_cert=nan_n.copy())
        if st not in (-3, -4):
            res["x"], res["y"] = self.vec("sol_x"), self.vec("sol_y")
        elif st == -3:
            res["prim_inf_cert"] = self.vec("dy")
        else:
            res["dual_inf_cert"] = self.vec("dx")
        return res

    def close(self):
        if self.h:
            lib().oracle_cleanup(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def csc_mv(M, v, stype=0, trans=False):
    M = M.tocsc()
    M.sort_indices()
    p, i, x = (np.ascontiguousarray(M.indptr, np.int64), np.ascontiguousarray(M.indices, np.int64),
               np.ascontiguousarray(M.data, np.float64))
    v = np.ascontiguousarray(v, np.float64)
    out = np.zeros(M.shape[1] if trans else M.shape[0])
    lib().oracle_csc_mv(M.shape[0], M.shape[1], _ip(p), _ip(i), _dp(x), int(stype), int(trans), _dp(v), _dp(out))
    return out


def pwa_linesearch(eta, beta, delta, alpha):
    delta = np.ascontiguousarray(delta, np.float64)
    alpha = np.ascontiguousarray(alpha, np.float64)
    return float(lib().oracle_pwa_linesearch(len(delta) // 2, float(eta), float(beta), _dp(delta), _dp(alpha)))


def set_threads(n):
    """threads of the oracle's OpenMP regions (default min(cores, 16); results do not depend on it)"""
    lib().oracle_set_threads(int(n))


def get_threads():
    return int(lib().oracle_get_threads())
