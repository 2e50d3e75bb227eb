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
