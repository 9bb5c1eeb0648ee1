import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
ctx = capi.Context(0)
ctx.mesh_upload(*meshgen.unit_cube(20)); nd = ctx.dofs_build(1)
u_exact, f = meshgen.manufactured(3)
ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(nd)); ctx.init()
for K in (1, 2, 3, 4, 5, 6, 7, 33, 34, 64, 65):
    sol = {}
    for lazy in (0, 1):
        ctx.tune("cgf_lazy", lazy)
        try:
            info = ctx.solve(rtol=1e-30, maxit=K)
        except capi.FdapdeError as e:
            info = ctx.info()
        sol[lazy] = ctx.solution()
    d = np.abs(sol[0] - sol[1]).max() / np.abs(sol[0]).max()
    print(f"maxit {K}: iters {info.iters}  rel diff eager vs lazy {d:.2e}")
for rtol in (1e-6, 1e-8, 1e-10):
    sol = {}
    for lazy in (0, 1):
        ctx.tune("cgf_lazy", lazy)
        info = ctx.solve(rtol=rtol)
        sol[lazy] = ctx.solution()
    d = np.abs(sol[0] - sol[1]).max() / np.abs(sol[0]).max()
    print(f"rtol {rtol:g}: iters {info.iters}  rel diff eager vs lazy {d:.2e}")
