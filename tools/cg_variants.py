import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for nx, order in ((40, 1), (119, 1), (30, 2)):
    ctx = capi.Context(0)
    ctx.mesh_upload(*meshgen.unit_cube(nx)); nd = ctx.dofs_build(order)
    u_exact, f = meshgen.manufactured(3)
    ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(nd)); ctx.init()
    for rtol in (1e-10, 1e-13):
        out = []
        for m in (capi.SOLVER_CG, capi.SOLVER_CG_SR, capi.SOLVER_CG_FUSED):
            for rep in range(2):
                info = ctx.solve(method=m, rtol=rtol)
            _, _, co = ctx.dofs_get()
            out.append(f"m{m}: it {info.iters} {info.t_solve_ms:.2f} ms relres {info.relres:.1e} conv {info.converged} err {np.abs(ctx.solution()-u_exact(co)).max():.2e}")
        print(f"nx {nx} P{order} rtol {rtol}: " + " | ".join(out))
    ctx.close()
