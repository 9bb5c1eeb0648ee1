#!/usr/bin/env python3
"""The wide single launch (24 rows per thread, x in HBM) on a system above the 2.1 M-row cap of the register-resident forms:
us per iteration and phase split for 4 / 6 / 12 passes of a phase loading together, next to the multi-launch path the system took before."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "132"))
nodes, cells, bnd = meshgen.unit_cube(nx)
_, f = meshgen.manufactured(3)
c = capi.Context(0)
c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
del nodes, cells
c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
for name, kv in (("wide, 6 passes together", [("persist", 1), ("persist_wide", 1), ("persist_wide_gj", 6)]), ("wide, 12 together", [("persist_wide_gj", 12)]),
                 ("wide, 4 together", [("persist_wide_gj", 4)]), ("multi-launch", [("persist_wide", 0)])):
    for k, v in kv:
        c.tune(k, v)
    c.solve(rtol=1e-10)
    i = min((c.solve(rtol=1e-10) for _ in range(2)), key=lambda i: i.t_solve_ms)
    lay = c.solver_layout(True)
    us = 1e3 * (i.launch_ms if i.persistent else i.t_solve_ms) / max(i.iters, 1)
    print(f"nx {nx}: {nd} DOFs, {name}: {us:.1f} us per iteration ({i.iters} iterations, persistent {i.persistent}, {c.solver_layout_kind(True)}, layout {lay[2] / 1e6:.0f} MB per iteration "
          f"= {lay[2] / 1e6 / us:.2f} TB/s) | operator phase slowest {1e3 * i.spmv_avg_ms:.1f} mean {1e3 * i.spmv_mean_ms:.1f} us, gather {1e3 * i.gather_avg_ms:.1f} us, update {1e3 * i.update_avg_ms:.1f} us", flush=True)
