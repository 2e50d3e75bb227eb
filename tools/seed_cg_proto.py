import numpy as np, scipy.sparse as sp, time
rng = np.random.default_rng(0)
n, dens = 6000, 0.01
k = int(0.70 * n)
nz = int(dens * n)
rows = np.repeat(np.arange(k), nz)
cols = np.concatenate([rng.choice(n, nz, replace=False) for _ in range(k)])
A = sp.csr_matrix((rng.standard_normal(k * nz), (rows, cols)), shape=(k, n))
# Ruiz-like row scaling: rows with unit inf-norm
A = sp.diags(1.0 / abs(A).max(axis=1).toarray().ravel()) @ A
# Q: symmetric sparse, diagonally dominant as the generator makes it
Ql = sp.random(n, n, density=dens / 2, random_state=1, data_rvs=rng.standard_normal).tocsr()
Qo = Ql + Ql.T
dq = np.asarray(abs(Qo).sum(axis=1)).ravel() + 1e-3 * rng.random(n)
scale = 1.0 / dq.max()
Q = (Qo + sp.diags(dq)) * scale
sigma = 1e-7
Dq = Q.diagonal() + sigma
d = 10.0 ** rng.uniform(2, 5, k)          # weights 1/mu, spread over 3 decades
At = A.T.tocsr()
def K(v): return Q @ v + sigma * v + At @ (d * (A @ v))
def Sp(u): return u / d + A @ ((At @ u) / Dq)
sdiag = 1.0 / d + np.asarray(A.multiply(A) @ (1.0 / Dq)).ravel()

def inner(b, tol, x0=None, store=None, maxit=2000):
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - Sp(x) if x0 is not None else b.copy()
    bn = np.linalg.norm(b)
    z = r / sdiag; p = z.copy(); rz = r @ z
    it = 0
    while np.linalg.norm(r) > tol * bn and it < maxit:
        s = Sp(p); ps = p @ s
        if store is not None: store.append((p.copy(), s.copy(), ps))
        a = rz / ps
        x += a * p; r -= a * s
        z = r / sdiag; rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
        it += 1
    return x, it

def galerkin(b, basis):
    # basis: list of (p, S'p, p'S'p), S'-conjugate within one solve (approximately); sequential projection = Gram-Schmidt in S' inner product
    x = np.zeros_like(b); r = b.copy()
    for p, s, ps in basis:
        a = (p @ r) / ps
        x += a * p; r -= a * s
    return x

def outer(rhs, tau, mode):
    x = np.zeros(n); r = rhs.copy(); bn = np.linalg.norm(rhs)
    basis = []; inner_its = []
    def Minv(r):
        u = r / Dq
        b = A @ u
        if mode == "plain" or not basis:
            st = basis if mode != "plain" else None
            s, it = inner(b, tau, store=st)
        else:
            x0 = galerkin(b, basis)
            st = basis if mode == "seed_all" else None
            s, it = inner(b, tau, x0=x0, store=st)
        inner_its.append(it)
        return u - (At @ s) / Dq
    z = Minv(r); p = z.copy(); rz = r @ z; o = 0
    while np.linalg.norm(r) > 1e-12 * bn and o < 40:
        Kp = K(p); a = rz / (p @ Kp)
        x += a * p; r -= a * Kp
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2; o += 1
    return o, inner_its

rhs = rng.standard_normal(n)
tau = 3e-6 * min(1.0, 1e4 / d.max())
for mode in ("plain", "seed_first", "seed_all"):
    t0 = time.time()
    o, its = outer(rhs, tau, mode)
    print(mode, "outer", o, "inner total", sum(its), its, "%.1f s" % (time.time() - t0), flush=True)
