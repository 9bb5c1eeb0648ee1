"""FGMRES iteration counts of multiplicative forms of the two-level preconditioner (damped Jacobi before / after the coarse correction) next to the additive one.
Outer method of the two-level solver: flexible GMRES against BiCGStab (operator / preconditioner applications to rtol 1e-10), same set-up as c5_pmg_proto.py.
Would a two-level preconditioner change C5?  P2 fine level, the P1 space on the SAME mesh as the coarse level (prolongation: a vertex DOF takes the vertex
value, an edge DOF the mean of its edge's two vertices), additive form M^-1 = D^-1 + P A1^-1 P^T with the coarse system solved exactly (SuperLU) -- the best
case of the idea -- inside BiCGStab, next to Jacobi-BiCGStab, on C5's operator at reduced sizes.  Both matrices assembled by the device, everything else numpy /
scipy on the host.  Counts fine-level operator applications to rtol 1e-10.  usage: c5_pmg_proto.py [nx ...]"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

sys.path.insert(0, ".")
from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen, workloads


def assemble(nodes, cells, bnd, order):
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    c.set_operator(workloads.c5_operator(capi))
    c.set_forcing(workloads.c5_forcing(c.quadrature_nodes()))
    c.init()
    rp, ci = c.pattern_get()
    A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
    f = c.force()
    dofs, bd, coords = c.dofs_get()
    c.close()
    return A, f, dofs, bd, coords


def bicgstab_prec(A, b, Minv, tol, maxit=20000):
    x, r = np.zeros_like(b), b.copy()
    r0, rho, alpha, omega, v, p, mv, nb = r.copy(), 1.0, 1.0, 1.0, np.zeros_like(b), np.zeros_like(b), 0, np.linalg.norm(b)
    for it in range(maxit):
        rho_new = r0 @ r
        p = r + ((rho_new / rho) * (alpha / omega)) * (p - omega * v) if it else r.copy()
        ph = Minv(p)
        v = A @ ph
        alpha = rho_new / (r0 @ v)
        s = r - alpha * v
        sh = Minv(s)
        t = A @ sh
        mv += 2
        omega = (t @ s) / (t @ t)
        x += alpha * ph + omega * sh
        r, rho = s - omega * t, rho_new
        if np.linalg.norm(r) <= tol * nb:
            return x, mv
    return x, -mv



def fgmres(A, b, Minv, tol, maxit=400):
    """flexible GMRES without restart: one preconditioner + one operator application per iteration, modified Gram-Schmidt, Givens on the Hessenberg column"""
    nb = np.linalg.norm(b)
    V, Z, H = [b / nb], [], np.zeros((maxit + 1, maxit))
    g = np.zeros(maxit + 1); g[0] = nb
    cs, sn = np.zeros(maxit), np.zeros(maxit)
    for j in range(maxit):
        Z.append(Minv(V[j]))
        w = A @ Z[j]
        for i in range(j + 1):
            H[i, j] = V[i] @ w
            w = w - H[i, j] * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V.append(w / H[j + 1, j])
        for i in range(j):
            H[i, j], H[i + 1, j] = cs[i] * H[i, j] + sn[i] * H[i + 1, j], -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
        d = np.hypot(H[j, j], H[j + 1, j])
        cs[j], sn[j] = H[j, j] / d, H[j + 1, j] / d
        H[j, j], H[j + 1, j] = d, 0.0
        g[j + 1], g[j] = -sn[j] * g[j], cs[j] * g[j]
        if abs(g[j + 1]) <= tol * nb:
            break
    k = j + 1
    y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
    x = sum(y[i] * Z[i] for i in range(k))
    return x, k


for nx in [int(a) for a in sys.argv[1:]] or (12, 20):
    t0 = time.time()
    nodes, cells, bnd = meshgen.unit_cube(nx)
    A2, f2, dofs2, bd2, x2 = assemble(nodes, cells, bnd, 2)
    A1, _, dofs1, bd1, x1 = assemble(nodes, cells, bnd, 1)
    n2, n1 = A2.shape[0], A1.shape[0]
    # prolongation from the DOF tables: local DOFs 0..3 of a P2 cell are its vertices (the same local order as the P1 table), 4..9 its edges -- matched to vertex
    # pairs by coordinates (the midpoint)
    key = lambda p: tuple(np.round(p * 4096.0 * 2).astype(np.int64))   # noqa: E731
    rows, cols, vals = [], [], []
    seen = np.zeros(n2, dtype=bool)
    pairs = [(i, j) for i in range(4) for j in range(i + 1, 4)]
    for e in range(cells.shape[0]):
        v1 = dofs1[e]
        for k in range(4):
            d = dofs2[e, k]
            if not seen[d]:
                seen[d] = True
                rows.append(d), cols.append(v1[k]), vals.append(1.0)
        mids = {key(0.5 * (x1[v1[i]] + x1[v1[j]])): (v1[i], v1[j]) for i, j in pairs}
        for k in range(4, 10):
            d = dofs2[e, k]
            if not seen[d]:
                seen[d] = True
                a, b = mids[key(x2[d])]
                rows += [d, d]
                cols += [a, b]
                vals += [0.5, 0.5]
    assert seen.all()
    P = sp.csr_matrix((vals, (rows, cols)), shape=(n2, n1))
    i2, i1 = np.flatnonzero(bd2 == 0), np.flatnonzero(bd1 == 0)
    A2i, f2i, Pi = A2[i2][:, i2].tocsr(), f2[i2], P[i2][:, i1].tocsr()
    A1i = A1[i1][:, i1].tocsc()
    galerkin = (Pi.T @ A2i @ Pi).tocsc()   # (the coarse operator the fine one induces; A1i is the P1 assembly of the same operator)
    dinv = 1.0 / A2i.diagonal()
    out = [f"nx {nx}: {i2.size} P2 / {i1.size} P1 interior DOFs"]
    lu = spl.splu(A1i)
    A1r = A1i.tocsr()
    d1 = 1.0 / A1r.diagonal()
    for cname, csolve in (("exact", lambda rc: lu.solve(rc)), ("inner 1e-1", lambda rc: bicgstab_prec(A1r, rc, lambda r: d1 * r, 1e-1, maxit=2000)[0])):
        cc = lambda r: Pi @ csolve(Pi.T @ r)   # noqa: E731
        res = []
        res.append(("additive", fgmres(A2i, f2i, lambda r: dinv * r + cc(r), 1e-10)[1]))
        for om in (0.5, 0.7, 1.0):
            def smooth_then_coarse(r, om=om):
                z = om * dinv * r
                return z + cc(r - A2i @ z)

            def coarse_then_smooth(r, om=om):
                z = cc(r)
                return z + om * dinv * (r - A2i @ z)

            def both(r, om=om):
                z = om * dinv * r
                z = z + cc(r - A2i @ z)
                return z + om * dinv * (r - A2i @ z)

            res.append((f"smooth({om})->coarse", fgmres(A2i, f2i, smooth_then_coarse, 1e-10)[1]))
            res.append((f"coarse->smooth({om})", fgmres(A2i, f2i, coarse_then_smooth, 1e-10)[1]))
            res.append((f"smooth->coarse->smooth({om})", fgmres(A2i, f2i, both, 1e-10)[1]))
        out.append(f"{cname}: " + ", ".join(f"{a} {b}" for a, b in res))
    print(" | ".join(out), f"[{time.time() - t0:.0f} s]", flush=True)
