"""On-disk exchange format for QP instances.

One compressed .npz per instance holding exactly what the reference's mex gateway hands to qpdo_setup
(interfaces/mex/qpdo_mex.c:134-150): the CSC arrays of Q (with its stype) and A, the vectors q, l, u, the
constant c and, optionally, the 19 settings.  A MATLAB user of the reference can produce the same arrays with
[i,j,v] = find(...) / sparse(...) and compare results against a genuine CHOLMOD build.
"""
import numpy as np
import scipy.sparse as sp

SETTING_NAMES = ["max_time", "max_iter", "inner_max_iter", "eps_abs", "eps_abs_in", "eps_prim_inf", "eps_dual_inf", "rho",
                 "theta", "delta", "mu_min", "proximal", "sigma_init", "sigma_upd", "sigma_min", "scaling", "verbose",
                 "print_interval", "reset_newton_iter"]


def save_problem(path, prob, settings=None, result=None):
    """prob: dict as produced by qpdo_amd.problems; settings: dict or ctypes settings struct; result: optional dict
    (info, x, y) stored beside the instance as a regression vector."""
    Q, A = sp.csc_matrix(prob["Q"]), sp.csc_matrix(prob["A"])
    Q.sort_indices(); A.sort_indices()
    out = dict(n=np.int64(A.shape[1]), m=np.int64(A.shape[0]), Qstype=np.int64(prob.get("Qstype", -1)), c=np.float64(prob.get("c", 0.0)),
               Qp=Q.indptr.astype(np.int64), Qi=Q.indices.astype(np.int64), Qx=Q.data.astype(np.float64),
               Ap=A.indptr.astype(np.int64), Ai=A.indices.astype(np.int64), Ax=A.data.astype(np.float64),
               q=np.asarray(prob["q"], np.float64), l=np.asarray(prob["l"], np.float64), u=np.asarray(prob["u"], np.float64))
    if settings is not None:
        get = (lambda k: settings[k]) if isinstance(settings, dict) else (lambda k: getattr(settings, k))
        out["settings"] = np.array([float(get(k)) for k in SETTING_NAMES])
    if result is not None:
        out["res_x"], out["res_y"] = np.asarray(result["x"], np.float64), np.asarray(result["y"], np.float64)
        out["res_info"] = np.array([result["info"]["status_val"], result["info"]["iterations"], result["info"]["oterations"]], np.int64)
    np.savez_compressed(path, **out)


def load_problem(path):
    z = np.load(path)
    n, m = int(z["n"]), int(z["m"])
    prob = dict(n=n, m=m, Qstype=int(z["Qstype"]), c=float(z["c"]),
                Q=sp.csc_matrix((z["Qx"], z["Qi"], z["Qp"]), shape=(n, n)), A=sp.csc_matrix((z["Ax"], z["Ai"], z["Ap"]), shape=(m, n)),
                q=z["q"], l=z["l"], u=z["u"])
    settings = None
    if "settings" in z:
        settings = {k: (int(v) if k in ("max_iter", "inner_max_iter", "proximal", "scaling", "verbose", "print_interval", "reset_newton_iter") else float(v))
                    for k, v in zip(SETTING_NAMES, z["settings"])}
    result = None
    if "res_x" in z:
        result = dict(x=z["res_x"], y=z["res_y"], info=dict(zip(["status_val", "iterations", "oterations"], (int(v) for v in z["res_info"]))))
    return prob, settings, result


# ---- MATLAB v5 .mat exchange (scipy.io) ------------------------------------------------------------------------------
# The reference's only front end is MATLAB (interfaces/mex/qpdo.m, qpdo_mex.c:134-150,227-281), so the route by which
# results of a genuine CHOLMOD build can reach tests/golden/ is a .mat file a MATLAB user can load, run and write back:
#   Q (sparse n x n, the FULL symmetric matrix as handed to qpdo.m's setup), q, A (sparse m x n), l, u, c,
#   settings (struct with the 19 fields of types.h:96-116), optionally x0 / y0 (warm start),
#   and -- written by tools/reference_fixture.m or by save_mat(result=...) -- ref: struct with x, y, prim_inf_cert,
#   dual_inf_cert, status_val, iterations, oterations, res_prim_norm, res_dual_norm, objective, source (a string).
_INT_SETTINGS = ("max_iter", "inner_max_iter", "proximal", "scaling", "verbose", "print_interval", "reset_newton_iter")


def _full_symmetric(Q, stype):
    Q = sp.csc_matrix(Q)
    if stype == 0:
        return Q
    T = sp.tril(Q, 0) if stype < 0 else sp.triu(Q, 0)
    return sp.csc_matrix(T + T.T - sp.diags(T.diagonal()))


def save_mat(path, prob, settings=None, result=None, warm=None, source="qpdo_amd"):
    """MATLAB-loadable twin of save_problem.  result: dict(info, x, y[, prim_inf_cert, dual_inf_cert])."""
    import scipy.io
    n, m = int(prob["n"]), int(prob["m"])
    out = dict(Q=_full_symmetric(prob["Q"], int(prob.get("Qstype", -1))).astype(np.float64), A=sp.csc_matrix(prob["A"]).astype(np.float64),
               q=np.asarray(prob["q"], np.float64).reshape(n, 1), l=np.asarray(prob["l"], np.float64).reshape(m, 1),
               u=np.asarray(prob["u"], np.float64).reshape(m, 1), c=np.float64(prob.get("c", 0.0)))
    if settings is not None:
        get = (lambda k: settings[k]) if isinstance(settings, dict) else (lambda k: getattr(settings, k))
        out["settings"] = {k: float(get(k)) for k in SETTING_NAMES}
    if warm is not None:
        out["x0"], out["y0"] = np.asarray(warm[0], np.float64).reshape(n, 1), np.asarray(warm[1], np.float64).reshape(m, 1)
    if result is not None:
        i = result["info"]
        ref = dict(x=np.asarray(result["x"], np.float64).reshape(n, 1), y=np.asarray(result["y"], np.float64).reshape(m, 1), source=str(source))
        for k in ("prim_inf_cert", "dual_inf_cert"):
            if k in result:
                ref[k] = np.asarray(result[k], np.float64).reshape(-1, 1)
        for k in ("status_val", "iterations", "oterations", "res_prim_norm", "res_dual_norm", "objective"):
            if k in i:
                ref[k] = float(i[k])
        out["ref"] = ref
    scipy.io.savemat(str(path), out, do_compression=True, oned_as="column")


def load_mat(path):
    """-> (prob, settings or None, ref or None, warm or None).  Q is returned as its lower triangle with Qstype = -1, which
    is what the mex gateway makes of the matrix (qpdo_mex.c:146-150); +-Inf bounds are clipped by the front end."""
    import scipy.io
    z = scipy.io.loadmat(str(path), squeeze_me=True, struct_as_record=False)
    A = sp.csc_matrix(z["A"])
    m, n = A.shape
    Q = sp.csc_matrix(z["Q"]) if sp.issparse(z["Q"]) else sp.csc_matrix(np.atleast_2d(z["Q"]))
    vec = lambda v, k: np.atleast_1d(np.asarray(v, np.float64)).reshape(k)
    prob = dict(n=n, m=m, Q=sp.csc_matrix(sp.tril(Q, 0)), Qstype=-1, A=A, q=vec(z["q"], n), l=vec(z["l"], m), u=vec(z["u"], m),
                c=float(z["c"]) if "c" in z else 0.0)
    settings = None
    if "settings" in z:
        s = z["settings"]
        settings = {k: (int(round(float(getattr(s, k)))) if k in _INT_SETTINGS else float(getattr(s, k))) for k in SETTING_NAMES if hasattr(s, k)}
    ref = None
    if "ref" in z:
        r = z["ref"]
        ref = {k: getattr(r, k) for k in r._fieldnames}
        for k in ("x", "y", "prim_inf_cert", "dual_inf_cert"):
            if k in ref:
                ref[k] = np.atleast_1d(np.asarray(ref[k], np.float64))
        for k in ("status_val", "iterations", "oterations"):
            if k in ref:
                ref[k] = int(round(float(ref[k])))
        ref["source"] = str(ref.get("source", "unknown"))
    warm = (vec(z["x0"], n), vec(z["y0"], m)) if "x0" in z and "y0" in z else None
    return prob, settings, ref, warm
