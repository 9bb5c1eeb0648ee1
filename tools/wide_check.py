"""The wide form of the single launch (24 rows per thread, x in HBM) checked three ways: a moderate system forced into it (fewer workgroups) against the
multi-launch path and its true residual, a system at its own size, and the bits of two runs."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
import scipy.sparse as sp
# (1) a moderate system forced into the wide form by limiting the workgroups: parity with the multi-launch path + true residual
nodes, cells, bnd = meshgen.unit_cube(44)
_, f = meshgen.manufactured(3)
c = capi.Context(0)
c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
_, _, coords = c.dofs_get()
g = 0.1 * coords[:, 0]
c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(g); c.init()
c.tune("persist", 0); i0 = c.solve(rtol=1e-11); u0 = c.solution()
c.tune("persist", 1); c.tune("persist_max_wg", 8)
i1 = c.solve(rtol=1e-11); u1 = c.solution()
i2 = c.solve(rtol=1e-11); u2 = c.solution()
print("layout", c.solver_layout_kind(True), "persistent", i1.persistent, "iters", i0.iters, i1.iters, "diff", np.linalg.norm(u1 - u0) / np.linalg.norm(u0), "repeat identical", np.array_equal(u1, u2))
rp, ci = c.pattern_get()
A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
b = c.force()
print("true residual", np.linalg.norm(A @ u1 - b) / np.linalg.norm(b))
c.close()
