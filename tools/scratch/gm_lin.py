import sys, numpy as np
sys.path.insert(0, '.')
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
import scipy.sparse as sp, scipy.sparse.linalg as spl
for nx, shift in ((40, 5e4), (120, 5e4), (300, 5e4), (300, 5e5)):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    d = np.array([1.0, 0.5]); pe = 3.0
    c.set_operator(-capi.laplacian() + capi.advection((2.0 * pe * nx / np.linalg.norm(d)) * d))
    c.set_forcing(np.zeros(3 * cells.shape[0])); c.init()
    vals = c.matrix_values(capi.MAT_STIFF) + shift * c.matrix_values(capi.MAT_MASS)
    rp, ci = c.pattern_get()
    K = sp.csr_matrix((vals, ci, rp), shape=(nd, nd)).tocsc()
    c.lin_compute(values=vals, symmetric=False)
    b = np.random.default_rng(5).standard_normal((nd, 2))
    ref = spl.splu(K).solve(b[:, 0])
    for m, name in ((capi.SOLVER_BICGSTAB, "bicg"), (capi.SOLVER_GMRES, "gmres"), (capi.SOLVER_AUTO, "auto")):
        try:
            x, info = c.lin_solve(b, method=m, rtol=1e-10)
            print(nx, shift, name, "iters", info.iters, "relres", info.relres, "conv", info.converged, "used", info.method_used, "err", np.linalg.norm(x[:, 0] - ref) / np.linalg.norm(ref))
        except capi.FdapdeError as e:
            i = c.info()
            print(nx, shift, name, "FAILED", e, "iters", i.iters, "relres", i.relres)
    c.close()
