"""Would another Krylov method steady or shorten C5's solve?  IDR(s) (van Gijzen & Sonneveld, biorthogonalised form) restated in numpy next to BiCGStab, both on the
Jacobi-scaled interior block of C5's operator as the device assembles it (reduced sizes: the comparison runs on the host).  Counts operator applications to
rtol 1e-10.  Result (profiles/r6_c5_idr_proto.txt): IDR(4) needs 10 - 20 % fewer applications; each of its applications carries ~2 s + 4 vector passes where
BiCGStab's carries 7 -- on C5 (operator 1.65 GB, a vector pass 43 MB) that leaves a few per cent.  Not built.  usage: c5_idr_proto.py [nx ...]"""
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen, workloads


def build(nx):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    c.set_operator(workloads.c5_operator(capi))
    c.set_forcing(workloads.c5_forcing(c.quadrature_nodes()))
    c.init()
    rp, ci = c.pattern_get()
    A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
    f = c.force()
    _, bd, _ = c.dofs_get()
    c.close()
    keep = np.nonzero(bd == 0)[0]
    A, f = A[keep][:, keep].tocsr(), f[keep]
    s = 1.0 / np.sqrt(np.abs(A.diagonal()))
    return (sp.diags(s) @ A @ sp.diags(s)).tocsr(), s * f


def bicgstab(A, b, tol, maxit=20000):
    x, r = np.zeros_like(b), b.copy()
    r0, rho, alpha, omega, v, p, mv, nb = r.copy(), 1.0, 1.0, 1.0, np.zeros_like(b), np.zeros_like(b), 0, np.linalg.norm(b)
    for it in range(maxit):
        rho_new = r0 @ r
        p = r + ((rho_new / rho) * (alpha / omega)) * (p - omega * v) if it else r.copy()
        v = A @ p
        alpha = rho_new / (r0 @ v)
        s = r - alpha * v
        t = A @ s
        mv += 2
        omega = (t @ s) / (t @ t)
        x += alpha * p + omega * s
        r, rho = s - omega * t, rho_new
        if np.linalg.norm(r) <= tol * nb:
            return x, mv
    return x, -mv


def idrs(A, b, s, tol, maxit=40000, seed=1):
    n, rng = b.size, np.random.default_rng(seed)
    P = np.linalg.qr(rng.standard_normal((n, s)))[0]
    x, r, nb, mv = np.zeros(n), b.copy(), np.linalg.norm(b), 0
    G, U, M, om = np.zeros((n, s)), np.zeros((n, s)), np.eye(s), 1.0
    while mv < maxit:
        f = P.T @ r
        for k in range(s):
            c = np.linalg.solve(M[k:, k:], f[k:])
            v = r - G[:, k:] @ c
            U[:, k] = U[:, k:] @ c + om * v
            G[:, k] = A @ U[:, k]
            mv += 1
            for i in range(k):
                al = (P[:, i] @ G[:, k]) / M[i, i]
                G[:, k] -= al * G[:, i]
                U[:, k] -= al * U[:, i]
            M[k:, k] = P[:, k:].T @ G[:, k]
            beta = f[k] / M[k, k]
            r, x = r - beta * G[:, k], x + beta * U[:, k]
            if np.linalg.norm(r) <= tol * nb:
                return x, mv
            if k + 1 < s:
                f[k + 1:] = f[k + 1:] - beta * M[k + 1:, k]
        t = A @ r
        mv += 1
        nt, nv, ts = np.linalg.norm(t), np.linalg.norm(r), t @ r
        rho = abs(ts / (nt * nv))
        om = ts / (nt * nt)
        if rho < 0.7:
            om *= 0.7 / rho
        x, r = x + om * r, r - om * t
        if np.linalg.norm(r) <= tol * nb:
            return x, mv
    return x, -mv


for nx in ([int(a) for a in sys.argv[1:]] or [12, 20, 28]) if __name__ == "__main__" else []:
    A, b = build(nx)
    t0 = time.time()
    _, mvb = bicgstab(A, b, 1e-10)
    out = [f"nx {nx}, {b.size} interior DOFs: BiCGStab {mvb} applications"]
    for s in (2, 4, 8):
        xs, mvs = idrs(A, b, s, 1e-10)
        out.append(f"IDR({s}) {mvs} (true relres {np.linalg.norm(b - A @ xs) / np.linalg.norm(b):.1e})")
    print(" | ".join(out), f"[{time.time() - t0:.0f} s]", flush=True)
