"""Fused-update CG (lazy x) against textbook CG and the single-reduction form on ill-conditioned SPD problems: strongly anisotropic
diffusion, a stretched mesh, and a reaction-dominated operator; iterations, claimed residual and the TRUE residual of the result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
cases = []
nodes, cells, bnd = meshgen.unit_square(200)
cases.append(("anisotropic 1:1e-4, 2-D P1", nodes, cells, bnd, 1, lambda: -capi.diffusion(np.array([[1.0, 0.0], [0.0, 1e-4]]))))
n2 = nodes.copy(); n2[:, 1] *= 1e-2
cases.append(("stretched mesh 1:100, 2-D P2", n2, cells, bnd, 2, lambda: -capi.laplacian()))
nodes3, cells3, bnd3 = meshgen.unit_cube(30)
cases.append(("reaction 1e4, 3-D P2", nodes3, cells3, bnd3, 2, lambda: -capi.laplacian() + capi.reaction(1e4)))
for name, nd_, cl_, b_, order, mkop in cases:
    ctx = capi.Context(0)
    ctx.mesh_upload(nd_, cl_, b_); nd = ctx.dofs_build(order)
    qn = ctx.quadrature_nodes()
    op = mkop()
    ctx.set_operator(op)
    ctx.set_forcing(np.sin(5 * qn[:, 0]) + qn[:, -1]); ctx.set_dirichlet(np.zeros(nd)); ctx.init()
    rp, ci = ctx.pattern_get()
    for rtol in (1e-10, 1e-13):
        line = []
        for m, lazy, lab in ((capi.SOLVER_CG, 0, "CG"), (capi.SOLVER_CG_SR, 0, "CG_SR"), (capi.SOLVER_CG_FUSED, 0, "fused"), (capi.SOLVER_CG_FUSED, 1, "fused+lazy")):
            ctx.tune("cgf_lazy", lazy)
            try:
                info = ctx.solve(method=m, rtol=rtol, maxit=200000)
                conv = info.converged
            except Exception as e:
                info = ctx.info(); conv = 0
            u = ctx.solution()
            A = sp.csr_matrix((ctx.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))   # row-zeroed system after the solve
            f = ctx.force()
            true = np.linalg.norm(A @ u - f) / np.linalg.norm(f)
            line.append(f"{lab} {info.iters} it conv {conv} true res {true:.1e}")
        print(f"{name}, rtol {rtol:g}: " + " | ".join(line))
    ctx.close()
