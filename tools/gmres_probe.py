#!/usr/bin/env python3
"""Advection-dominated operators -Lap u + b . grad u at cell Peclet numbers 15 - 1000: what FDAPDE_SOLVER_AUTO does (CG -> BiCGStab with restarts
-> GMRES(m)) against scipy's sparse LU (the stand-in for the reference's direct solve), and GMRES called directly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen

def run(dim, nx, pe, order=1):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(order)
    _, _, coords = c.dofs_get()
    d = np.array([1.0, 0.5, 0.25])[:dim]
    bmag = 2.0 * pe * nx / np.linalg.norm(d)
    c.set_operator(-capi.laplacian() + capi.advection(bmag * d))
    c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(0.2 * coords[:, 0]); c.init()
    out = []
    for name, kw in (("auto", {}), ("gmres", {"method": capi.SOLVER_GMRES}), ("bicgstab", {"method": capi.SOLVER_BICGSTAB})):
        t0 = time.perf_counter()
        i = c.solve(rtol=1e-10, raise_on_noconv=False, **kw)
        dt = time.perf_counter() - t0
        if name == "auto":
            rp, ci = c.pattern_get()
            A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
            ref = spl.spsolve(A.tocsc(), c.force())
        err = np.linalg.norm(c.solution() - ref) / np.linalg.norm(ref)
        out.append(f"{name}: conv {i.converged} method {i.method_used} iters {i.iters} relres {i.relres:.1e} err-vs-LU {err:.1e} {1e3 * dt:.1f} ms")
    print(f"{dim}-D P{order} nx {nx} Pe {pe} ({nd} DOFs): " + " | ".join(out), flush=True)
    c.close()

if __name__ == "__main__":
    for dim, nx, pe in ((2, 32, 17), (2, 32, 150), (2, 32, 500), (2, 32, 1000), (2, 64, 150), (2, 128, 150), (2, 128, 1000), (3, 10, 150), (3, 10, 1000), (3, 16, 500)):
        run(dim, nx, pe)
