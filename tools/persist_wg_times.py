"""Per-workgroup operator-phase times of the persistent CG on C3 (FDAPDE_DEBUG_PERSIST=1 prints them from the library)."""
import os, sys
os.environ["FDAPDE_DEBUG_PERSIST"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nodes, cells, bnd = meshgen.unit_cube(int(os.environ.get("NX", "119")))
_, f = meshgen.manufactured(3)
c = capi.Context(0)
c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
if os.environ.get('SYM'): c.tune('persist_sym', int(os.environ['SYM']))
c.solve(rtol=1e-10)
i = c.solve(rtol=1e-10, time_spmv=32)
print("iters", i.iters, "us/it", 1e3 * i.t_solve_ms / i.iters)
