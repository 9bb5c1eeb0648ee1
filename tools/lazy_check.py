"""cgf_lazy (x updated every second launch) against the eager form: iterations, solve time, solution difference, true residual."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for dim, nx, order in ((3, 119, 1), (3, 30, 2), (2, 300, 1), (2, 90, 2), (3, 9, 1)):
    ctx = capi.Context(0)
    ctx.mesh_upload(*(meshgen.unit_cube(nx) if dim == 3 else meshgen.unit_square(nx))); nd = ctx.dofs_build(order)
    u_exact, f = meshgen.manufactured(dim)
    ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(nd)); ctx.init()
    for rtol in (1e-10, 1e-13):
        out = {}
        for lazy in (0, 1):
            ctx.tune("cgf_lazy", lazy)
            for _ in range(2):
                info = ctx.solve(rtol=rtol)
            out[lazy] = (info.iters, info.t_solve_ms, ctx.solution(), info.relres, info.converged)
        d = np.abs(out[0][2] - out[1][2]).max() / np.abs(out[0][2]).max()
        print(f"dim {dim} nx {nx} P{order} rtol {rtol:g}: eager {out[0][0]} it {out[0][1]:.2f} ms | lazy {out[1][0]} it {out[1][1]:.2f} ms conv {out[1][4]} | rel. solution diff {d:.1e}")
    ctx.close()
