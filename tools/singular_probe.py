"""-Lap u = 1 WITHOUT a Dirichlet DOF (a singular matrix, the right-hand side outside its range) through the single launch and the multi-launch path:
what the open method reports (success = false: the iterate grows beyond 1e12 |b|) instead of a 'solution' of size 1e14."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen
nodes, cells, bnd = meshgen.unit_square(8)
for persist in (1, 0):
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, np.zeros_like(bnd)); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian()); c.set_forcing(np.full(c.quadrature_nodes().shape[0], 1.0)); c.set_dirichlet(np.full(nd, 0.0)); c.init()
    c.tune("persist", persist)
    i = c.solve(rtol=1e-10, maxit=200, raise_on_noconv=False)
    u = c.solution()
    print("persist", persist, "conv", i.converged, "iters", i.iters, "relres", i.relres, "method", i.method_used, "u range", u.min(), u.max(), "finite", np.isfinite(u).all())
    A = c.matrix_values(capi.MAT_STIFF); print("row sums max", np.abs(np.add.reduceat(A, c.pattern_get()[0][:-1])).max())
