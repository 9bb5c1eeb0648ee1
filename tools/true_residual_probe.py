#!/usr/bin/env python3
"""Recurrence residual against TRUE residual on ill-conditioned systems (no Dirichlet DOF, a domain of 1e-3 .. 1e-6 across: -Lap + c is singular to 1e-6 .. 1e-12):
what `converged = 1` at rtol means there.  The true residual ||b - A u|| / ||b|| is computed with scipy from the matrix the product hands out."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402
import scipy.sparse as sp   # noqa: E402
import scipy.sparse.linalg as spl   # noqa: E402

rng = np.random.default_rng(2)
for dim, nx, order, scale, adv in ((2, 8, 2, 1e-3, False), (2, 8, 2, 1e-5, False), (2, 8, 2, 1e-6, False), (3, 7, 1, 1e-5, True), (2, 30, 1, 1e-6, False)):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    nodes = nodes * scale
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, np.zeros_like(bnd))
    nd = c.dofs_build(order)
    op = -capi.laplacian() + capi.reaction(1.0)
    if adv:
        op = op + capi.advection(np.full(dim, 0.3))
    c.set_operator(op)
    c.set_forcing(rng.standard_normal(c.quadrature_nodes().shape[0]))
    c.init()
    for rtol in (1e-10, 1e-12, 1e-14):
        info = c.solve(rtol=rtol, raise_on_noconv=False)
        u = c.solution()
        rp, ci = c.pattern_get()
        A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
        b = c.force()
        ref = spl.spsolve(A.tocsc(), b)
        true_res = np.linalg.norm(b - A @ u) / np.linalg.norm(b)
        lu_res = np.linalg.norm(b - A @ ref) / np.linalg.norm(b)
        print(f"dim {dim} P{order} nx {nx} domain {scale:g}{' adr' if adv else ''}: rtol {rtol:g}: converged {info.converged} method {info.method_used} iters {info.iters} "
              f"reported relres {info.relres:.2e} TRUE relres {true_res:.2e} (LU's {lu_res:.1e}) | error vs LU {np.linalg.norm(u - ref) / np.linalg.norm(ref):.2e}", flush=True)
    c.close()
