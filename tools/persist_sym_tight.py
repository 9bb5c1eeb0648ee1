"""Symmetric against plain storage of the persistent CG at tight tolerances on an ill-conditioned system (2-D P1, 1.96 M DOFs: 5 000
iterations at rtol 1e-10): the fixed-point accumulators must not cost attainable accuracy or iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "1400"))
nodes, cells, bnd = meshgen.unit_square(nx)
u_exact, f = meshgen.manufactured(2)
c = capi.Context(0)
c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
A_rp, A_ci = c.pattern_get(); A_v = c.matrix_values(capi.MAT_STIFF); b = c.force()
import scipy.sparse as sp
A = sp.csr_matrix((A_v, A_ci, A_rp), shape=(nd, nd))
interior = bnd == 0 if bnd.size == nd else None
for rtol in (1e-10, 1e-12, 1e-13, 1e-14):
    for sym in (0, 1):
        c.tune("persist_sym", sym)
        i = c.solve(rtol=rtol, maxit=40000, raise_on_noconv=False)
        u = c.solution()
        r = (b - A @ u)[interior] if interior is not None else None
        true_rel = np.linalg.norm(r) / np.linalg.norm(b[interior]) if r is not None else float("nan")
        print(f"rtol {rtol:.0e} persist_sym={sym}: converged {i.converged} iters {i.iters} recurrence relres {i.relres:.2e} true relres (unscaled system) {true_rel:.2e} "
              f"error vs analytic {np.abs(u - u_exact(nodes)).max():.3e}", flush=True)
