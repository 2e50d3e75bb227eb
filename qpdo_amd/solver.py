"""Python front end over the C-ABI of libqpdo_amd.so.

Plays the role of the reference's MATLAB handle class (interfaces/mex/qpdo.m) and mex gateway
(interfaces/mex/qpdo_mex.c): dimension checks, +-Inf -> +-1e20 clipping (qpdo.m:138-139,215-216),
settings merge with unknown-field rejection (qpdo.m:238-273), and the result marshalling rule of
qpdo_mex.c:247-279 (NaN solution for infeasible statuses, NaN certificates otherwise).

Every numerical call goes through the C entry points declared in include/qpdo.h; there is no
Python or CPU implementation behind this class.
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _build

QPDO_INFTY = 1e20
c_int = C.c_long
c_float = C.c_double
dp = C.POINTER(C.c_double)


class CholmodSparse(C.Structure):
    """Layout of cholmod_sparse (include/qpdo.h)."""
    _fields_ = [("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t), ("p", C.c_void_p),
                ("i", C.c_void_p), ("nz", C.c_void_p), ("x", C.c_void_p), ("z", C.c_void_p),
                ("stype", C.c_int), ("itype", C.c_int), ("xtype", C.c_int), ("dtype", C.c_int),
                ("sorted", C.c_int), ("packed", C.c_int)]


class QPDOSettings(C.Structure):
    _fields_ = [("max_time", c_float), ("max_iter", c_int), ("inner_max_iter", c_int), ("eps_abs", c_float),
                ("eps_abs_in", c_float), ("eps_prim_inf", c_float), ("eps_dual_inf", c_float), ("rho", c_float),
                ("theta", c_float), ("delta", c_float), ("mu_min", c_float), ("proximal", c_int),
                ("sigma_init", c_float), ("sigma_upd", c_float), ("sigma_min", c_float), ("scaling", c_int),
                ("verbose", c_int), ("print_interval", c_int), ("reset_newton_iter", c_int)]


class QPDOData(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("Q", C.POINTER(CholmodSparse)), ("A", C.POINTER(CholmodSparse)),
                ("q", dp), ("c", c_float), ("l", dp), ("u", dp)]


class QPDOInfo(C.Structure):
    _fields_ = [("iterations", c_int), ("oterations", c_int), ("status", C.c_char * 32), ("status_val", c_int),
                ("res_prim_norm", c_float), ("res_dual_norm", c_float), ("res_prim_in_norm", c_float),
                ("res_dual_in_norm", c_float), ("objective", c_float), ("setup_time", c_float),
                ("solve_time", c_float), ("run_time", c_float)]


class QPDOSolution(C.Structure):
    _fields_ = [("x", dp), ("y", dp)]


class QPDOScaling(C.Structure):
    _fields_ = [("D", dp), ("Dinv", dp), ("E", dp), ("Einv", dp), ("c", c_float), ("cinv", c_float)]


class QPDOWorkspace(C.Structure):
    """Member order of QPDOWorkspace in include/qpdo.h (= reference include/types.h:147-224)."""
    _fields_ = [
        ("data", C.POINTER(QPDOData)),
        ("x", dp), ("y", dp), ("Ax", dp), ("Qx", dp), ("Aty", dp), ("initialized", c_int),
        ("temp_m", dp), ("temp_n", dp), ("temp_2m", dp),
        ("mu", dp), ("sqrt_mu", dp), ("sqrt_mu_min", c_float), ("sqrt_delta", c_float), ("n_mu_changed", c_int),
        ("sigma", c_float), ("sigma_mined", c_int), ("norm_q", c_float),
        ("xbar", dp), ("ybar", dp), ("dx", dp), ("dy", dp), ("tau", c_float), ("Qdx", dp), ("Adx", dp), ("Atdy", dp),
        ("w", dp), ("z", dp), ("df", dp), ("res_prim", dp), ("res_dual", dp), ("res_prim_old", dp),
        ("res_prim_in", dp), ("res_dual_in", dp), ("linsys_rhs", dp),
        ("res_prim_norm_old", c_float), ("res_dual_norm_old", c_float),
        ("ls_eta", c_float), ("ls_beta", c_float), ("ls_delta", dp), ("ls_alpha", dp), ("ls_taus", C.c_void_p),
        ("ls_idx_L", C.c_void_p), ("ls_idx_P", C.c_void_p), ("ls_idx_J", C.c_void_p),
        ("eps_prim", c_float), ("eps_dual", c_float), ("eps_prim_in", c_float), ("eps_dual_in", c_float), ("eps_in", c_float),
        ("D_temp", dp), ("E_temp", dp),
        ("chol", C.c_void_p), ("settings", C.POINTER(QPDOSettings)), ("scaling", C.POINTER(QPDOScaling)),
        ("solution", C.POINTER(QPDOSolution)), ("info", C.POINTER(QPDOInfo)), ("timer", C.c_void_p),
    ]


class TraceRec(C.Structure):
    _fields_ = [("kind", C.c_long), ("n_active", C.c_long), ("n_enter", C.c_long), ("n_leave", C.c_long),
                ("factor_branch", C.c_long), ("lin_iters", C.c_long), ("tau", C.c_double), ("res_prim", C.c_double),
                ("res_dual", C.c_double), ("res_prim_in", C.c_double), ("res_dual_in", C.c_double),
                ("sigma", C.c_double), ("eps_in", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("newton_passes", C.c_long), ("lin_iters", C.c_long), ("spmv_calls", C.c_long),
                ("spmv_alg_bytes", C.c_double), ("factor_count", C.c_long), ("linsolve", C.c_long),
                ("spmv_Q_avg_s", C.c_double), ("spmv_Q_samples", C.c_long),
                ("spmv_Ac_time_s", C.c_double), ("spmv_Ac_bytes", C.c_double), ("spmv_Ac_samples", C.c_long),
                ("schur_passes", C.c_long), ("lowrank_solves", C.c_long), ("lowrank_cols", C.c_long), ("lowrank_sweeps", C.c_long),
                ("lowrank_rejects", C.c_long), ("pcg_soft_accepts", C.c_long), ("collectives", C.c_long), ("inner_solves", C.c_long),
                ("inner_steps", C.c_long), ("inner_collectives", C.c_long), ("chain_fallbacks", C.c_long),
                ("pcg_max_relres", C.c_double), ("pcg_dense_fallbacks", C.c_long), ("fused_solves", C.c_long), ("fused_kernel_s", C.c_double),
                ("pcg_rescues", C.c_long), ("pcg_rescue_kinds", C.c_long), ("hybrid_pcg_passes", C.c_long), ("band_fallbacks", C.c_long),
                ("onelaunch_factors", C.c_long), ("ahead_steps", C.c_long), ("ahead_skips", C.c_long)]


API_SYMBOLS = ["qpdo_set_default_settings", "qpdo_setup", "qpdo_warm_start", "qpdo_solve", "qpdo_update_settings",
               "qpdo_update_bounds", "qpdo_update_q", "qpdo_cleanup"]
class BatchItem(C.Structure):
    _fields_ = [("data", C.POINTER(QPDOData)), ("x0", dp), ("y0", dp), ("x", dp), ("y", dp), ("info", QPDOInfo)]


EXT_SYMBOLS = ["qpdo_amd_dist_config", "qpdo_amd_dist_unique_id", "qpdo_amd_solve_batch", "qpdo_amd_batch_kernel_seconds", "qpdo_amd_batch_stream_create",
               "qpdo_amd_batch_stream_submit", "qpdo_amd_batch_stream_wait", "qpdo_amd_batch_stream_destroy", "qpdo_amd_device_count", "qpdo_amd_last_error", "qpdo_amd_get_stats", "qpdo_amd_get_trace",
               "qpdo_amd_sync", "qpdo_amd_pass_decision", "qpdo_amd_bench_spmv", "qpdo_amd_bench_dense_factor", "qpdo_amd_spmv", "qpdo_amd_linesearch", "qpdo_amd_download"]

_lib = None


def lib():
    """Loads libqpdo_amd.so (building it in-tree if needed).  Raises if it is unavailable."""
    global _lib
    if _lib is None:
        # batch streams keep up to `depth` launches in flight on separate HIP streams; two streams that share one of the runtime's
        # GPU_MAX_HW_QUEUES (default 4) hardware queues serialise (7.0 k vs 10.1 k QP/s at depth 12).  The variable is read at the
        # process's first HIP call, so it is the CALLER's to set: this front end does it here, before the library is loaded -- the
        # library itself never changes its host's environment.  No effect if HIP was initialised earlier.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        L = C.CDLL(_build.ensure_lib())
        W = C.POINTER(QPDOWorkspace)
        L.qpdo_set_default_settings.argtypes = [C.POINTER(QPDOSettings)]
        L.qpdo_setup.restype = W
        L.qpdo_setup.argtypes = [C.POINTER(QPDOData), C.POINTER(QPDOSettings)]
        L.qpdo_warm_start.argtypes = [W, dp, dp]
        L.qpdo_solve.argtypes = [W]
        L.qpdo_update_settings.argtypes = [W, C.POINTER(QPDOSettings)]
        L.qpdo_update_bounds.argtypes = [W, dp, dp]
        L.qpdo_update_q.argtypes = [W, dp]
        L.qpdo_cleanup.argtypes = [W]
        L.qpdo_amd_device_count.restype = C.c_int
        L.qpdo_amd_last_error.restype = C.c_char_p
        L.qpdo_amd_get_stats.argtypes = [W, C.POINTER(Stats)]
        L.qpdo_amd_get_trace.argtypes = [W, C.POINTER(C.POINTER(TraceRec)), C.POINTER(C.c_long)]
        L.qpdo_amd_sync.argtypes = [W]
        L.qpdo_amd_bench_spmv.argtypes = [W, C.c_int, C.c_int, dp, dp]
        L.qpdo_amd_spmv.argtypes = [W, C.c_int, dp, dp]
        L.qpdo_amd_bench_dense_factor.argtypes = [W, C.c_int, dp, dp]
        L.qpdo_amd_linesearch.argtypes = [W, C.c_double, C.c_double, dp, dp, dp]
        L.qpdo_amd_download.argtypes = [W, C.c_int, dp]
        L.qpdo_amd_dist_config.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.qpdo_amd_dist_unique_id.argtypes = [C.c_void_p]
        L.qpdo_amd_solve_batch.restype = C.c_long
        L.qpdo_amd_batch_kernel_seconds.restype = C.c_double
        L.qpdo_amd_solve_batch.argtypes = [C.c_long, C.POINTER(BatchItem), C.POINTER(QPDOSettings), C.c_int]
        L.qpdo_amd_batch_stream_create.restype = C.c_void_p
        L.qpdo_amd_batch_stream_create.argtypes = [C.c_int]
        L.qpdo_amd_batch_stream_submit.restype = C.c_long
        L.qpdo_amd_batch_stream_submit.argtypes = [C.c_void_p, C.c_long, C.POINTER(BatchItem), C.POINTER(QPDOSettings)]
        L.qpdo_amd_batch_stream_wait.restype = C.c_int
        L.qpdo_amd_batch_stream_wait.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_double)]
        L.qpdo_amd_batch_stream_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def device_count():
    return int(lib().qpdo_amd_device_count())


def default_settings(**over):
    s = QPDOSettings()
    lib().qpdo_set_default_settings(C.byref(s))
    names = {f for f, _ in QPDOSettings._fields_}
    for k, v in over.items():
        if k not in names:                      # qpdo.m:262-266 rejects unknown fields
            raise KeyError("unrecognized solver setting '%s'" % k)
        setattr(s, k, v)
    return s


def _as_dp(a):
    return None if a is None else a.ctypes.data_as(dp)


def _sparse_view(M, stype, keep, index_dtype=None):
    """cholmod_sparse header over the CSC arrays of M.  index_dtype None: the index type scipy holds (int32 below 2^31 entries), which the
    host driver reads as CHOLMOD_INT -- no 64-bit copy of the index arrays (0.4 s of setup at 2e8 nonzeros); np.int64: CHOLMOD_LONG, the
    reference's DLONG layout."""
    M = sp.csc_matrix(M)
    M.sort_indices()
    if index_dtype is None:
        index_dtype = np.int32 if M.indices.dtype == np.int32 and M.indptr.dtype == np.int32 else np.int64
    p = np.ascontiguousarray(M.indptr, index_dtype)
    i = np.ascontiguousarray(M.indices, index_dtype)
    x = np.ascontiguousarray(M.data, np.float64)
    keep.extend([p, i, x])
    s = CholmodSparse()
    s.nrow, s.ncol, s.nzmax = M.shape[0], M.shape[1], max(1, len(x))
    s.p, s.i, s.x = p.ctypes.data, i.ctypes.data, x.ctypes.data
    s.nz, s.z = None, None
    s.stype, s.itype, s.xtype, s.dtype, s.sorted, s.packed = stype, (2 if index_dtype == np.int64 else 0), 1, 0, 1, 1
    return s


class QPDO:
    """solver = QPDO(); solver.setup(Q, q, A, l, u, **settings); res = solver.solve()"""

    def __init__(self):
        self._w = None
        self.n = self.m = 0

    # qpdo.m:50-160
    def setup(self, Q, q, A, l, u, settings=None, Qstype=None, c=0.0, index_dtype=None, **kw):
        if self._w:
            raise RuntimeError("Solver is already initialized with problem data.")   # qpdo_mex.c:122-124
        A = sp.csc_matrix(A)
        Q = sp.csc_matrix(Q)
        m, n = A.shape
        if Q.shape != (n, n):
            raise ValueError("Q must be n x n with n = number of columns of A")
        q = np.zeros(n) if q is None else np.ascontiguousarray(q, np.float64).ravel()
        l = np.full(m, -QPDO_INFTY) if l is None else np.ascontiguousarray(l, np.float64).ravel()
        u = np.full(m, QPDO_INFTY) if u is None else np.ascontiguousarray(u, np.float64).ravel()
        if len(q) != n or len(l) != m or len(u) != m:
            raise ValueError("incompatible vector dimensions")
        l = np.clip(l, -QPDO_INFTY, QPDO_INFTY)      # qpdo.m:138-139
        u = np.clip(u, -QPDO_INFTY, QPDO_INFTY)
        if Qstype is None:
            Q = sp.tril(Q).tocsc()                     # the mex reads the lower triangle (qpdo_mex.c:150)
            Qstype = -1
        if settings is None:
            settings = default_settings(**kw)
        elif kw:
            for k, v in kw.items():
                if not hasattr(settings, k):
                    raise KeyError("unrecognized solver setting '%s'" % k)
                setattr(settings, k, v)
        keep = [q, l, u]
        Qs, As = _sparse_view(Q, Qstype, keep, index_dtype), _sparse_view(A, 0, keep, index_dtype)
        data = QPDOData()
        data.n, data.m, data.Q, data.A = n, m, C.pointer(Qs), C.pointer(As)
        data.q, data.c, data.l, data.u = _as_dp(q), float(c), _as_dp(l), _as_dp(u)
        w = lib().qpdo_setup(C.byref(data), C.byref(settings))
        if not w:
            raise RuntimeError("Invalid problem setup: %s" % (lib().qpdo_amd_last_error() or b"").decode())
        self._w, self.n, self.m = w, n, m
        return self

    @property
    def work(self):
        return self._w.contents

    def warm_start(self, x=None, y=None):
        x = None if x is None else np.ascontiguousarray(x, np.float64)
        y = None if y is None else np.ascontiguousarray(y, np.float64)
        lib().qpdo_warm_start(self._w, _as_dp(x), _as_dp(y))

    def update_bounds(self, l=None, u=None):
        l = None if l is None else np.clip(np.ascontiguousarray(l, np.float64), -QPDO_INFTY, QPDO_INFTY)   # qpdo.m:215-216
        u = None if u is None else np.clip(np.ascontiguousarray(u, np.float64), -QPDO_INFTY, QPDO_INFTY)
        lib().qpdo_update_bounds(self._w, _as_dp(l), _as_dp(u))

    def update_q(self, q):
        q = np.ascontiguousarray(q, np.float64)
        if len(q) != self.n:
            raise ValueError("q has wrong length")
        lib().qpdo_update_q(self._w, _as_dp(q))

    def update_settings(self, settings=None, **kw):
        s = QPDOSettings()
        C.memmove(C.byref(s), self.work.settings, C.sizeof(QPDOSettings))
        if settings is not None:
            s = settings
        for k, v in kw.items():
            if not hasattr(s, k):
                raise KeyError("unrecognized solver setting '%s'" % k)
            setattr(s, k, v)
        lib().qpdo_update_settings(self._w, C.byref(s))

    def info(self):
        i = self.work.info.contents
        d = {f: getattr(i, f) for f, _ in QPDOInfo._fields_}
        d["status"] = d["status"].decode()
        return d

    def _vec(self, ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0)

    # qpdo_mex.c:227-281
    def solve(self):
        lib().qpdo_solve(self._w)
        info = self.info()
        st = info["status_val"]
        w = self.work
        nan_n, nan_m = np.full(self.n, np.nan), np.full(self.m, np.nan)
        res = dict(info=info, x=nan_n, y=nan_m, prim_inf_cert=nan_m.copy(), dual_inf_cert=nan_n.copy())
        if st not in (-3, -4):
            res["x"] = self._vec(w.solution.contents.x, self.n)
            res["y"] = self._vec(w.solution.contents.y, self.m)
        elif st == -3:
            res["prim_inf_cert"] = self._vec(w.dy, self.m)
        else:
            res["dual_inf_cert"] = self._vec(w.dx, self.n)
        return res

    # extensions
    def stats(self):
        s = Stats()
        lib().qpdo_amd_get_stats(self._w, C.byref(s))
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    def trace(self):
        p, n = C.POINTER(TraceRec)(), C.c_long(0)
        lib().qpdo_amd_get_trace(self._w, C.byref(p), C.byref(n))
        return [{f: getattr(p[i], f) for f, _ in TraceRec._fields_} for i in range(n.value)]

    def bench_spmv(self, which, reps=20):
        t, b = C.c_double(0), C.c_double(0)
        rc = lib().qpdo_amd_bench_spmv(self._w, which, reps, C.byref(t), C.byref(b))
        if rc:
            raise RuntimeError(lib().qpdo_amd_last_error().decode())
        return t.value, b.value

    def bench_dense_factor(self, reps=5, check=True):
        t, c = C.c_double(0), C.c_double(0)
        rc = lib().qpdo_amd_bench_dense_factor(self._w, reps, C.byref(t), C.byref(c) if check else None)
        if rc:
            raise RuntimeError(lib().qpdo_amd_last_error().decode())
        return t.value, c.value

    def spmv(self, which, v):
        v = np.ascontiguousarray(v, np.float64)
        rows = {0: self.m, 1: self.n, 2: self.n}[which]
        y = np.zeros(rows)
        rc = lib().qpdo_amd_spmv(self._w, which, _as_dp(v), _as_dp(y))
        if rc:
            raise RuntimeError(lib().qpdo_amd_last_error().decode())
        return y

    def linesearch(self, eta, beta, delta, alpha):
        delta = np.ascontiguousarray(delta, np.float64)
        alpha = np.ascontiguousarray(alpha, np.float64)
        assert len(delta) == 2 * self.m == len(alpha)
        t = C.c_double(0)
        rc = lib().qpdo_amd_linesearch(self._w, float(eta), float(beta), _as_dp(delta), _as_dp(alpha), C.byref(t))
        if rc:
            raise RuntimeError(lib().qpdo_amd_last_error().decode())
        return t.value

    def download(self, name):
        which, n = {"x": (0, self.n), "Qx": (1, self.n), "y": (2, self.m), "mu": (3, self.m), "d": (4, self.m),
                    "dx": (5, self.n), "dy": (6, self.m), "Ax": (7, self.m), "Aty": (8, self.n),
                    "l": (9, self.m), "u": (10, self.m), "ybar": (11, self.m), "xbar": (12, self.n), "w": (13, self.m)}[name]
        out = np.zeros(n)
        lib().qpdo_amd_download(self._w, which, _as_dp(out))
        return out

    def scaling(self):
        w = self.work
        if not w.scaling:
            return None
        s = w.scaling.contents
        return dict(D=self._vec(s.D, self.n), E=self._vec(s.E, self.m), c=s.c, cinv=s.cinv)

    def delete(self):
        if self._w:
            lib().qpdo_cleanup(self._w)
            self._w = None

    def __del__(self):
        try:
            self.delete()
        except Exception:
            pass


def solve_problem(prob, settings=None, **kw):
    """Convenience: prob dict from qpdo_amd.problems -> result dict (+ stats, trace)."""
    s = QPDO().setup(prob["Q"], prob["q"], prob["A"], prob["l"], prob["u"], settings=settings,
                     Qstype=prob.get("Qstype", -1), c=prob.get("c", 0.0), **kw)
    res = s.solve()
    res["stats"], res["trace"] = s.stats(), s.trace()
    s.delete()
    return res


class Batch:
    """Host-side image of a batch of independent QPs for qpdo_amd_solve_batch (ctypes structs + the arrays they point
    to).  Building it is Python/scipy work; `run()` is the C call alone and may be repeated."""

    def __init__(self, probs, indices=None):
        self.keep, self.items, self.outs = [], (BatchItem * len(probs))(), []
        self.probs = probs
        self.indices = list(range(len(probs))) if indices is None else list(indices)   # global item numbers of this shard
        for i, p in enumerate(probs):
            A, Q = sp.csc_matrix(p["A"]), sp.csc_matrix(p["Q"])
            m, n = A.shape
            q = np.ascontiguousarray(p["q"], np.float64)
            l = np.clip(np.ascontiguousarray(p["l"], np.float64), -QPDO_INFTY, QPDO_INFTY)
            u = np.clip(np.ascontiguousarray(p["u"], np.float64), -QPDO_INFTY, QPDO_INFTY)
            Qs, As = _sparse_view(Q, p.get("Qstype", -1), self.keep), _sparse_view(A, 0, self.keep)
            data = QPDOData()
            data.n, data.m, data.Q, data.A = n, m, C.pointer(Qs), C.pointer(As)
            data.q, data.c, data.l, data.u = _as_dp(q), float(p.get("c", 0.0)), _as_dp(l), _as_dp(u)
            x, y = np.zeros(n), np.zeros(m)
            self.keep.extend([q, l, u, Qs, As, data, x, y])
            self.items[i].data = C.pointer(data)
            self.items[i].x0, self.items[i].y0 = None, None
            self.items[i].x, self.items[i].y = _as_dp(x), _as_dp(y)
            self.outs.append((x, y))

    def twin(self):
        """A second host image of the same batch that shares the (read-only) problem data and has its own output buffers: several
        twins of one Batch can be in flight on a BatchStream at once (building an image from scratch is Python / scipy work)."""
        t = Batch.__new__(Batch)
        t.probs, t.indices = self.probs, list(self.indices)
        t.keep, t.items, t.outs = [self], (BatchItem * len(self.outs))(), []
        for i in range(len(self.outs)):
            n, m = len(self.outs[i][0]), len(self.outs[i][1])
            x, y = np.zeros(n), np.zeros(m)
            t.items[i].data = self.items[i].data
            t.items[i].x0, t.items[i].y0 = self.items[i].x0, self.items[i].y0
            t.items[i].x, t.items[i].y = _as_dp(x), _as_dp(y)
            t.outs.append((x, y))
        return t

    def run(self, settings=None, nthreads=16, results=True, **kw):
        """results=False: skip building the per-item dicts (25 ms for 4096 items); the outputs are in info_view() and outs"""
        if settings is None:
            settings = default_settings(**kw)
        failed = lib().qpdo_amd_solve_batch(len(self.outs), self.items, C.byref(settings), int(nthreads))
        self.kernel_seconds = float(lib().qpdo_amd_batch_kernel_seconds())
        return (self.results() if results else None), int(failed)

    def info_view(self):
        """The items' QPDOInfo fields as ONE numpy structured array over the item array itself (no copy, no per-item Python work):
        info_view()["iterations"], ["oterations"], ["status_val"], ["res_prim_norm"] ... -- what a caller that streams batches
        looks at per batch; x and y of item i are self.outs[i] either way.  results() builds the per-item dicts from the same memory."""
        base = BatchItem.info.offset
        names, formats, offsets = [], [], []
        for f, t in QPDOInfo._fields_:
            if f == "status":
                continue
            names.append(f); formats.append(np.dtype(t)); offsets.append(base + getattr(QPDOInfo, f).offset)
        dt = np.dtype(dict(names=names, formats=formats, offsets=offsets, itemsize=C.sizeof(BatchItem)))
        return np.frombuffer(self.items, dtype=dt, count=len(self.outs))

    def results(self):
        """list of dicts (info, x, y) from the items' output fields (after run(), or after BatchStream.wait)"""
        names = [f for f, _ in QPDOInfo._fields_]
        res = []
        for i, (x, y) in enumerate(self.outs):
            inf = self.items[i].info
            info = {f: getattr(inf, f) for f in names}
            info["status"] = info["status"].decode()
            res.append(dict(info=info, x=x.copy(), y=y.copy()))
        return res


class BatchStream:
    """Up to `depth` fused-kernel batches in flight on this process's GPU (qpdo_amd_batch_stream_*): submit() packs, uploads
    and launches a Batch and returns a ticket without waiting for the GPU; wait(ticket) blocks until that batch is complete
    and returns (results, kernel_seconds).  A Batch object may be in flight only once at a time (its items hold the output
    buffers)."""

    def __init__(self, depth=8):
        self._h = lib().qpdo_amd_batch_stream_create(int(depth))
        if not self._h:
            raise RuntimeError("qpdo_amd_batch_stream_create failed")
        self._inflight = {}

    def submit(self, batch, settings=None, **kw):
        if settings is None:
            settings = default_settings(**kw)
        if any(b is batch for b, _ in self._inflight.values()):
            raise ValueError("this Batch is already in flight")
        t = int(lib().qpdo_amd_batch_stream_submit(self._h, len(batch.outs), batch.items, C.byref(settings)))
        if t < 0:
            raise RuntimeError("qpdo_amd_batch_stream_submit failed: " + lib().qpdo_amd_last_error().decode())
        self._inflight[t] = (batch, settings)
        return t

    def wait(self, ticket, results=True):
        """results=False: skip building the per-item dicts (the outputs are in the Batch: info_view(), outs)"""
        batch, _ = self._inflight.pop(ticket)
        ks = C.c_double(0.0)
        if lib().qpdo_amd_batch_stream_wait(self._h, int(ticket), C.byref(ks)) != 0:
            raise RuntimeError("qpdo_amd_batch_stream_wait failed: " + lib().qpdo_amd_last_error().decode())
        batch.kernel_seconds = float(ks.value)
        return (batch.results() if results else None), float(ks.value)

    def close(self):
        if self._h:
            lib().qpdo_amd_batch_stream_destroy(self._h)
            self._h = None
            self._inflight.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_indices(count, rank, world):
    """Items of a batch of `count` independent QPs owned by `rank` of `world` processes (one per GPU): item b goes to
    GPU b mod world.  Disjoint, complete and order-stable; no data-path collective."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("rank %r not in [0, %r)" % (rank, world))
    return list(range(rank, count, world))


def shard_batch(count, rank, world, make_problem):
    """Batch of this rank's items only: make_problem(i) is called for the owned global indices."""
    idx = shard_indices(count, rank, world)
    return Batch([make_problem(i) for i in idx], indices=idx)


def merge_shards(count, shards):
    """shards: iterable of (indices, results) as returned per rank -> list of `count` results in global order."""
    out = [None] * count
    for idx, res in shards:
        for i, r in zip(idx, res):
            if out[i] is not None:
                raise ValueError("item %d solved twice" % i)
            out[i] = r
    missing = [i for i, r in enumerate(out) if r is None]
    if missing:
        raise ValueError("items not solved: %r" % missing[:8])
    return out


def solve_batch(probs, settings=None, nthreads=16, **kw):
    """Solve independent QPs (dicts from qpdo_amd.problems) concurrently on this process's GPU.
    Returns a list of dicts (info, x, y) and the number of failed items."""
    return Batch(probs).run(settings, nthreads, **kw)


# ---- one large QP row-partitioned over the ranks of a torch.distributed job ---------------------------------
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_long, C.c_int)
_dist_keep = []


def dist_config(rank, world, mode="rccl", group=None, force=False):
    """Partition the rows of A over `world` ranks for the workspaces created next in this process.
    mode 'rccl': all-reduce with RCCL on the solver stream (one GPU per rank); the unique id is created by rank 0
    and broadcast over torch.distributed.  mode 'host': all-reduce through torch.distributed on host buffers
    (gloo) - slow, for tests of the partition logic on a single GPU."""
    L = lib()
    if world <= 1 and force and mode == "rccl":
        # a single-rank RCCL communicator: the partition is the whole problem, but every collective call site of the
        # row-partitioned solver runs through ncclAllReduce on the solver's stream (tests of the RCCL branch on one GPU)
        uid = C.create_string_buffer(128)
        if L.qpdo_amd_dist_unique_id(uid) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        _dist_keep.append(uid)
        return L.qpdo_amd_dist_config(0, 1, C.cast(uid, C.c_void_p), None, None)
    if world <= 1:
        return L.qpdo_amd_dist_config(0, 1, None, None, None)
    import torch
    import torch.distributed as dist
    if mode == "host":
        def _cb(ctx, buf, count, op):
            a = np.ctypeslib.as_array(buf, shape=(count,))
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)
        fn = ALLREDUCE_FN(_cb)
        _dist_keep.append(fn)
        return L.qpdo_amd_dist_config(rank, world, None, C.cast(fn, C.c_void_p), None)
    uid = C.create_string_buffer(128)
    if rank == 0:
        if L.qpdo_amd_dist_unique_id(uid) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
    box = [bytes(uid.raw)]
    dist.broadcast_object_list(box, src=0, group=group)
    uid = C.create_string_buffer(box[0], 128)
    _dist_keep.append(uid)
    return L.qpdo_amd_dist_config(rank, world, C.cast(uid, C.c_void_p), None, None)
