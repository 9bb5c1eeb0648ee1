#!/usr/bin/env python3
"""us per iteration of the single-launch CG on 3-D P1 Laplace systems of growing size (what dist._ITER_US_BY_ROWS tabulates: the iteration of one rank's
share of C3 as one launch on all 256 CUs)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402

for nx in (45, 60, 64, 72, 90, 100, 119):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(3)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    best = None
    for _ in range(4):
        i = c.solve(rtol=1e-10)
        t = 1e3 * i.launch_ms / max(i.iters, 1)
        best = t if best is None else min(best, t)
    ni, _, _ = c.solver_layout(True)
    k = c.solver_layout_kind(True)
    print(f"nx {nx}: {nd} DOFs, {ni} interior rows, {i.iters} iterations, {best:.2f} us per iteration, layout {k}", flush=True)
    c.close()
