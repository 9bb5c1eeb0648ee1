#!/usr/bin/env python3
"""Strongly indefinite symmetric operators -Lap u - k^2 u (many negative eigenvalues): FDAPDE_SOLVER_AUTO (CG -> BiCGStab -> GMRES(m)) against scipy's LU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen

def run(dim, nx, k2, m=50):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    _, bd, coords = c.dofs_get()
    c.set_operator(-capi.laplacian() + capi.reaction(-k2))
    c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(0.2 * coords[:, 0]); c.init()
    c.tune("gmres_m", m)
    t0 = time.perf_counter()
    i = c.solve(rtol=1e-10, raise_on_noconv=False)
    dt = time.perf_counter() - t0
    rp, ci = c.pattern_get()
    A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
    ref = spl.spsolve(A.tocsc(), c.force())
    inter = np.flatnonzero(bd == 0)
    ev = np.linalg.eigvalsh(A[inter][:, inter].toarray()) if nd < 5000 else np.array([0.0])
    err = np.linalg.norm(c.solution() - ref) / np.linalg.norm(ref)
    print(f"{dim}-D nx {nx} k2 {k2} m {m} ({nd} DOFs, {int((ev < 0).sum())} negative eigenvalues, min |ev| {np.abs(ev).min():.1e}): conv {i.converged} method {i.method_used} "
          f"iters {i.iters} relres {i.relres:.1e} err-vs-LU {err:.1e} {1e3 * dt:.0f} ms", flush=True)
    c.close()

if __name__ == "__main__":
    for dim, nx, k2, m in ((2, 32, 100, 50), (2, 32, 1000, 50), (2, 32, 5000, 50), (2, 32, 5000, 150), (2, 64, 5000, 50), (2, 64, 5000, 200), (3, 10, 300, 50), (3, 10, 3000, 50), (3, 10, 3000, 200)):
        run(dim, nx, k2, m)
