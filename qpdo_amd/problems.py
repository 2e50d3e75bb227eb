"""Synthetic QP instances for tests and bench.py.

`random_qp` wraps the seeded C generator (csrc/qpdo_gen.c).  `infeasibility_kat`
returns the three explicit 2-variable QPs of the reference's
examples/infeasibility_tests.m:15-68 (data only), with MATLAB's conventions
applied by hand: sparse() drops explicit zeros and the qpdo.m front end clips
infinite bounds to +-1e20 (interfaces/mex/qpdo.m:138-139).
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _build

QPDO_INFTY = 1e20

# (name, shape, density, n_eq, seed offset): BASELINE.json configs / SURVEY.md section 8
CONFIGS = {
    "C1": dict(n=200, m=100, density=0.1, n_eq=0),       # examples/demo_mex.m:7-9
    "C1b": dict(n=50, m=100, density=0.1, n_eq=0),       # BASELINE.json configs[0] shape
    "C2": dict(n=10_000, m=20_000, density=0.01, n_eq=0),
    "C3": dict(n=120, m=360, density=0.1, n_eq=120),
    "C4": dict(n=100_000, m=200_000, density=0.01, n_eq=0),
}
BASE_SEED = 123456  # examples/demo_mex.m:4

_gen = None


def _genlib():
    global _gen
    if _gen is None:
        L = C.CDLL(_build.ensure_gen())
        ip, dp = C.POINTER(C.c_int64), C.POINTER(C.c_double)
        L.qpdo_gen_A_per_col.restype = C.c_int64
        L.qpdo_gen_A_per_col.argtypes = [C.c_int64, C.c_double]
        L.qpdo_gen_Q_nnz.restype = C.c_int64
        L.qpdo_gen_Q_nnz.argtypes = [C.c_int64, C.c_double]
        L.qpdo_gen_problem.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_double, C.c_int64,
                                       ip, ip, dp, ip, ip, dp, dp, dp, dp]
        _gen = L
    return _gen


def random_qp(seed, n, m, density, n_eq=0):
    """Seeded random sparse convex QP.  Q is returned as its lower triangle (Qstype -1)."""
    L = _genlib()
    K = int(L.qpdo_gen_A_per_col(m, density))
    nnzQ = int(L.qpdo_gen_Q_nnz(n, density))
    Ap, Ai, Ax = np.empty(n + 1, np.int64), np.empty(n * K, np.int64), np.empty(n * K, np.float64)
    Qp, Qi, Qx = np.empty(n + 1, np.int64), np.empty(nnzQ, np.int64), np.empty(nnzQ, np.float64)
    q, l, u = np.empty(n), np.empty(m), np.empty(m)
    ip, dp = C.POINTER(C.c_int64), C.POINTER(C.c_double)
    L.qpdo_gen_problem(int(seed), n, m, float(density), int(n_eq),
                       Ap.ctypes.data_as(ip), Ai.ctypes.data_as(ip), Ax.ctypes.data_as(dp),
                       Qp.ctypes.data_as(ip), Qi.ctypes.data_as(ip), Qx.ctypes.data_as(dp),
                       q.ctypes.data_as(dp), l.ctypes.data_as(dp), u.ctypes.data_as(dp))
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(m, n))
    Q = sp.csc_matrix((Qx, Qi, Qp), shape=(n, n))
    return dict(n=n, m=m, Q=Q, Qstype=-1, A=A, q=q, l=l, u=u, c=0.0, seed=int(seed))


def banded_qp(seed, n, n_rate=None, q_scale=1.0, box=1.0, rate=0.05, q_reg=0.0):
    """A chain-structured (MPC / spline-type) QP: Q = second-difference operator (tridiagonal, singular along constants unless
    q_reg > 0), constraints = n box rows -box <= x_i <= box followed by n_rate rate rows |x_{i+1} - x_i| <= rate.  The Newton matrix
    Q + sigma I + A'DA is tridiagonal: trivial for a direct factorization (the reference: CHOLMOD), but its Jacobi-preconditioned
    condition number is ~ 4 / sigma, so a diagonally preconditioned CG stalls late in a solve -- the case that needs a direct solver."""
    rng = np.random.default_rng(seed)
    n_rate = n - 1 if n_rate is None else int(n_rate)
    main = np.full(n, 2.0 + q_reg); main[0] = main[-1] = 1.0 + q_reg
    Q = sp.diags([main, -np.ones(n - 1)], [0, -1], format="csc")            # lower triangle
    rows = np.concatenate([np.arange(n), n + np.arange(n_rate), n + np.arange(n_rate)])
    cols = np.concatenate([np.arange(n), np.arange(n_rate), np.arange(n_rate) + 1])
    vals = np.concatenate([np.ones(n), -np.ones(n_rate), np.ones(n_rate)])
    m = n + n_rate
    A = sp.csc_matrix((vals, (rows, cols)), shape=(m, n))
    A.sort_indices()
    t = np.linspace(0.0, 1.0, n)
    q = q_scale * (np.sin(6.0 * np.pi * t) * 0.02 + 0.002 * rng.standard_normal(n))
    l = np.concatenate([np.full(n, -box), np.full(n_rate, -rate)])
    u = np.concatenate([np.full(n, box), np.full(n_rate, rate)])
    return dict(n=n, m=m, Q=Q, Qstype=-1, A=A, q=q, l=l, u=u, c=0.0, seed=int(seed))


def banded_random_qp(seed, n, bw, n_win=None, win=None):
    """A random chain-structured QP with half-bandwidth `bw`: Q = symmetric band (random off-diagonals, diagonally dominant), constraints =
    n box rows followed by n_win "window" rows with `win` consecutive random coefficients each (column span win - 1 <= bw): the pattern
    of Q + A'DA has half-bandwidth exactly max(bw of Q, win - 1) for every D."""
    rng = np.random.default_rng(seed)
    win = min(bw + 1, 8) if win is None else int(win)
    assert 1 <= win <= bw + 1
    n_win = n // 2 if n_win is None else int(n_win)
    diags, offs = [], []
    tot = np.zeros(n)
    for k in range(1, bw + 1):
        if k > 3 and rng.random() < 0.5 and k != bw:
            continue                                             # (a band with holes; the outermost diagonal is always there)
        v = 0.3 * rng.standard_normal(n - k)
        diags.append(v); offs.append(-k)
        tot[:-k] += np.abs(v); tot[k:] += np.abs(v)
    Q = sp.diags([tot + 0.05 + 0.1 * rng.random(n)] + diags, [0] + offs, format="csc")      # lower triangle
    starts = np.sort(rng.integers(0, n - win + 1, n_win))
    rows = np.concatenate([np.arange(n), n + np.repeat(np.arange(n_win), win)])
    cols = np.concatenate([np.arange(n), (starts[:, None] + np.arange(win)[None, :]).ravel()])
    vals = np.concatenate([np.ones(n), rng.standard_normal(n_win * win)])
    m = n + n_win
    A = sp.csc_matrix((vals, (rows, cols)), shape=(m, n))
    A.sort_indices()
    q = rng.standard_normal(n)
    l = np.concatenate([-rng.random(n), -rng.random(n_win)])
    u = np.concatenate([rng.random(n), rng.random(n_win)])
    neq = n_win // 5
    l[n:n + neq] = 0.0; u[n:n + neq] = 0.0                           # some equality rows (x = 0 stays feasible)
    return dict(n=n, m=m, Q=Q, Qstype=-1, A=A, q=q, l=l, u=u, c=0.0, seed=int(seed))


def config_qp(name, index=0):
    cfg = CONFIGS[name]
    seed = BASE_SEED + 1000 * (list(CONFIGS).index(name) + 1) + index
    return random_qp(seed, cfg["n"], cfg["m"], cfg["density"], cfg["n_eq"])


def full_Q(prob):
    """Dense-free full symmetric Q from the stored triangle."""
    Q = prob["Q"]
    st = prob.get("Qstype", -1)
    if st == 0:
        return Q.tocsc()
    T = sp.tril(Q) if st < 0 else sp.triu(Q)
    return (T + T.T - sp.diags(T.diagonal())).tocsc()


def infeasibility_kat(case):
    """case in {'degenerate','primal_infeasible','dual_infeasible'}; expected status 1 / -3 / -4
    (reference examples/infeasibility_tests.m:30,48,75)."""
    a, b, c, expected = {
        "degenerate": (0.0, 3.0, 0.0, 1),
        "primal_infeasible": (1.0, 3.0, 0.0, -3),
        "dual_infeasible": (0.0, np.inf, -1.0, -4),
    }[case]
    Qd = np.array([[1.0, 0.0], [0.0, 0.0]])
    Ad = np.array([[a, a], [1.0, 0.0], [0.0, 1.0]])
    Q = sp.csc_matrix(sp.tril(sp.csc_matrix(Qd)))
    A = sp.csc_matrix(Ad)
    Q.eliminate_zeros()
    A.eliminate_zeros()
    l = np.clip(np.array([-np.inf, 1.0, 1.0]), -QPDO_INFTY, QPDO_INFTY)
    u = np.clip(np.array([0.0, 3.0, b]), -QPDO_INFTY, QPDO_INFTY)
    return dict(n=2, m=3, Q=Q, Qstype=-1, A=A, q=np.array([1.0, c]), l=l, u=u, c=0.0,
                expected_status=expected, max_iter=100)


def kkt_residuals(prob, x, y):
    """Unscaled outer residuals as the reference's demo recomputes them
    (examples/demo_mex.m:39-40): ||Ax - clip(Ax+y)||inf, ||Qx+q+A'y||inf."""
    A, Qf = prob["A"], full_Q(prob)
    Ax = A @ x
    rp = np.abs(Ax - np.clip(Ax + y, prob["l"], prob["u"])).max() if prob["m"] else 0.0
    rd = np.abs(Qf @ x + prob["q"] + A.T @ y).max() if prob["n"] else 0.0
    return float(rp), float(rd)
