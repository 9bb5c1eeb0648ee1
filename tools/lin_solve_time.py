#!/usr/bin/env python3
"""fdapde_lin_solve on a factor-once handle (mass matrix of a 2-D P1 space, nx given): wall time per solve, iterations, launch duration;
FDAPDE_DEBUG_TIMING=1 prints where the host side spends it."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen
nodes, cells, bnd = meshgen.unit_square(int(sys.argv[1]))
c = capi.Context(0)
c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
c.set_operator(-capi.laplacian()); qn = c.quadrature_nodes(); c.set_forcing(np.ones(qn.shape[0])); c.set_dirichlet(np.zeros(nd)); c.init()
c.lin_compute(capi.MAT_MASS, symmetric=True)
b = np.random.default_rng(0).standard_normal(nd)
for _ in range(3): c.lin_solve(b)
c.synchronize()
t0 = time.perf_counter()
for _ in range(50): c.lin_solve(b)
c.synchronize()
i = c.info()
print(f"{nd} DOFs: {1e3 * (time.perf_counter() - t0) / 50:.1f} us per solve; {i.iters} iterations, launch {1e3 * i.launch_ms:.1f} us, layout {c.solver_layout_kind(False)}")
