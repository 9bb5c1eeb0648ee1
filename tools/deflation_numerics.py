#!/usr/bin/env python3
"""Would subdomain deflation pay inside the single-launch CG?  (CPU study, numpy / scipy; nothing of the product is touched.)

The persistent CG spends 503 iterations on C3 because Jacobi leaves the low end of the Laplacian's spectrum alone.  Deflating one vector per
workgroup block (D^(1/2) 1 on the block, "DEF1" of Tang / Nabben / Vuik / Erlangga 2009) would fit the kernel's structure -- Z^T y is a sum per
workgroup that can ride in the dot all-gather, E = Z^T A Z is 256 x 256 -- at the price of a second all-gather per iteration (the quadratic form
d^T E^-1 d sits between the operator and alpha): ~34.5 instead of 30 us.  This script counts what it buys, on the bench's own meshes:
  python tools/deflation_numerics.py NX          blocks = 256 chunks of a Morton ordering (what the layout's blocks look like), 256 / 128 / 64 vectors
  python tools/deflation_numerics.py NX cubes    ideal m^3 cubic aggregates
  python tools/deflation_numerics.py NX lin      cubic aggregates x (1, x, y, z)
Result: 1.4 - 1.5 x fewer iterations with block-shaped aggregates, 2.1 x with ideal cubes of the same number -- CG's own superlinear
phase already takes care of the extreme eigenvalues; not enough for the extra gather + set-up (E and its inverse per solve)."""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import meshgen

def assemble(nodes, cells):
    X = nodes[cells]                       # (nc,4,3)
    J = X[:, 1:, :] - X[:, :1, :]          # rows = edges
    det = np.linalg.det(J)
    vol = np.abs(det) / 6.0
    Jinv = np.linalg.inv(J)                # (nc,3,3): columns = gradients of lambda_1..3
    G = np.empty((cells.shape[0], 4, 3))
    G[:, 1:, :] = np.transpose(Jinv, (0, 2, 1))
    G[:, 0, :] = -G[:, 1:, :].sum(axis=1)
    K = np.einsum('cid,cjd->cij', G, G) * vol[:, None, None]
    I = np.repeat(cells, 4, axis=1).ravel(); Jc = np.tile(cells, (1, 4)).ravel()
    n = nodes.shape[0]
    A = sp.coo_matrix((K.ravel(), (I, Jc)), shape=(n, n)).tocsr()
    return A

def morton(coords, bits=10):
    q = np.minimum((coords * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    key = np.zeros(coords.shape[0], dtype=np.int64)
    for b in range(bits):
        for d in range(3):
            key |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return np.argsort(key, kind='stable')

def pcg(A, b, tol, maxit, Zinfo=None):
    n = A.shape[0]
    if Zinfo is not None:
        Z, AZ, Einv = Zinfo
        def P(v):  return v - AZ @ (Einv @ (Z.T @ v))
    else:
        P = lambda v: v
    x = np.zeros(n); r = np.array(P(b), copy=True); p = r.copy(); rr = r @ r; bb = b @ b; it = 0
    r0 = rr
    while it < maxit and rr > tol * tol * r0:
        w = P(A @ p)
        a = rr / (p @ w); x += a * p; r -= a * w
        rn = r @ r; p = r + (rn / rr) * p; rr = rn; it += 1
    if Zinfo is not None:
        x = Z @ (Einv @ (Z.T @ b)) + x - Z @ (Einv @ (AZ.T @ x))
    return x, it

def run(nx, G=256, ks=(256, 128, 64)):
    t = time.time()
    nodes, cells, bnd = meshgen.unit_cube(nx)
    A = assemble(nodes, cells)
    inter = np.flatnonzero(bnd == 0)
    Ai = A[inter][:, inter].tocsr()
    d = Ai.diagonal(); s = 1.0 / np.sqrt(d)
    As = sp.diags(s) @ Ai @ sp.diags(s); As = As.tocsr()
    n = As.shape[0]
    print(f"nx {nx}: {n} interior rows, nnz {As.nnz}, assembled in {time.time()-t:.1f} s", flush=True)
    rng = np.random.default_rng(0)
    _, f = meshgen.manufactured(3)
    # right-hand side: the load of the manufactured forcing, lumped (close enough to the bench's b for counting iterations)
    M = np.bincount(cells.ravel(), weights=np.repeat(np.abs(np.linalg.det(nodes[cells][:,1:,:]-nodes[cells][:,:1,:]))/24.0, 4), minlength=nodes.shape[0])
    fv = f(nodes)
    b = (M * fv)[inter] * s
    t = time.time(); x0, it0 = pcg(As, b, 1e-10, 5000); print(f"  plain Jacobi-PCG: {it0} iterations ({time.time()-t:.1f} s)", flush=True)
    order = morton(nodes[inter])
    wg = np.empty(n, dtype=np.int64); wg[order] = (np.arange(n) * G) // n
    for k in ks:
        agg = wg * k // G
        z = 1.0 / s                                    # D^{1/2} 1
        Z = sp.csr_matrix((z, (np.arange(n), agg)), shape=(n, k))
        AZ = (As @ Z).tocsr()
        AZ.data[np.abs(AZ.data) < 1e-13] = 0; AZ.eliminate_zeros()
        E = (Z.T @ AZ).toarray(); Einv = np.linalg.inv(E)
        rows_touched = np.count_nonzero(np.diff(AZ.indptr))
        t = time.time(); x1, it1 = pcg(As, b, 1e-10, 5000, (Z, AZ, Einv))
        res = np.linalg.norm(As @ x1 - b) / np.linalg.norm(b)
        print(f"  deflated, {k} vectors: {it1} iterations ({time.time()-t:.1f} s), true relres {res:.2e}, |x-x0|/|x0| {np.linalg.norm(x1-x0)/np.linalg.norm(x0):.2e}, "
              f"AZ nnz {AZ.nnz} on {rows_touched} rows ({100.0*rows_touched/n:.0f} %), cond(E) {np.linalg.cond(E):.1f}", flush=True)


def run_cubes(nx, ms=(4, 6, 8, 12)):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    A = assemble(nodes, cells)
    inter = np.flatnonzero(bnd == 0)
    Ai = A[inter][:, inter].tocsr()
    d = Ai.diagonal(); s = 1.0 / np.sqrt(d)
    As = (sp.diags(s) @ Ai @ sp.diags(s)).tocsr()
    n = As.shape[0]
    _, f = meshgen.manufactured(3)
    M = np.bincount(cells.ravel(), weights=np.repeat(np.abs(np.linalg.det(nodes[cells][:,1:,:]-nodes[cells][:,:1,:]))/24.0, 4), minlength=nodes.shape[0])
    b = (M * f(nodes))[inter] * s
    x0, it0 = pcg(As, b, 1e-10, 5000); print(f"nx {nx} plain {it0}")
    X = nodes[inter]
    for m in ms:
        c = np.minimum((X * m).astype(np.int64), m - 1)
        agg = (c[:, 0] * m + c[:, 1]) * m + c[:, 2]
        k = m ** 3
        Z = sp.csr_matrix((1.0 / s, (np.arange(n), agg)), shape=(n, k))
        AZ = (As @ Z).tocsr()
        E = (Z.T @ AZ).toarray(); Einv = np.linalg.inv(E)
        x1, it1 = pcg(As, b, 1e-10, 5000, (Z, AZ, Einv))
        print(f"  cubes {m}^3 = {k} vectors: {it1} iterations", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "cubes": run_cubes(int(sys.argv[1]))
elif len(sys.argv) == 2: run(int(sys.argv[1]))

def run_lin(nx, ms=(4, 6)):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    A = assemble(nodes, cells)
    inter = np.flatnonzero(bnd == 0)
    Ai = A[inter][:, inter].tocsr()
    d = Ai.diagonal(); s = 1.0 / np.sqrt(d)
    As = (sp.diags(s) @ Ai @ sp.diags(s)).tocsr()
    n = As.shape[0]
    _, f = meshgen.manufactured(3)
    M = np.bincount(cells.ravel(), weights=np.repeat(np.abs(np.linalg.det(nodes[cells][:,1:,:]-nodes[cells][:,:1,:]))/24.0, 4), minlength=nodes.shape[0])
    b = (M * f(nodes))[inter] * s
    X = nodes[inter]
    for m in ms:
        c = np.minimum((X * m).astype(np.int64), m - 1)
        agg = (c[:, 0] * m + c[:, 1]) * m + c[:, 2]
        k = m ** 3
        cols = [sp.csr_matrix((1.0 / s, (np.arange(n), agg)), shape=(n, k))]
        ctr = (c + 0.5) / m
        for dd in range(3):
            cols.append(sp.csr_matrix(((X[:, dd] - ctr[:, dd]) / s, (np.arange(n), agg)), shape=(n, k)))
        Z = sp.hstack(cols).tocsr()
        AZ = (As @ Z).tocsr()
        E = (Z.T @ AZ).toarray(); Einv = np.linalg.inv(E)
        x1, it1 = pcg(As, b, 1e-10, 5000, (Z, AZ, Einv))
        print(f"  cubes {m}^3 x (1, x, y, z) = {4*k} vectors: {it1} iterations", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "lin": run_lin(int(sys.argv[1]))
