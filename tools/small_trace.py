"""A 289-DOF solve three times over, for `rocprofv3 --kernel-trace`: which launches a small solve consists of (one: k_small_front's work and the epilogue are inside the
single launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen
nodes, cells, bnd = meshgen.unit_square(16)
_, f = meshgen.manufactured(2)
c = capi.Context(0)
c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd))
for _ in range(3):
    c.init(); c.solve(rtol=1e-10)
c.synchronize()
print("MARK", flush=True)
c.init()
c.solve(rtol=1e-10)
c.synchronize()
