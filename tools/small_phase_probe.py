#!/usr/bin/env python3
"""Where the iteration of a ONE-workgroup system goes: in-kernel phase stamps (operator / sums / update) of small 2-D systems.  The stamps are
switched off for such systems in production (small_rows); this probe switches them on (small_rows = 0)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402

for nx, order in ((16, 1), (32, 1), (44, 1), (20, 2)):
    nodes, cells, bnd = meshgen.unit_square(nx)
    _, f = meshgen.manufactured(2)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    c.tune("small_rows", 0)
    c.tune("persist_time", 1)
    for _ in range(3):
        i = c.solve(rtol=1e-10)
    k = c.solver_layout_kind(True)
    print(f"2-D P{order} nx {nx}: {nd} DOFs, {i.iters} iterations, launch {1e3 * i.launch_ms:.1f} us = {1e3 * i.launch_ms / max(i.iters, 1):.2f} us/iteration; "
          f"stamps per iteration: operator {1e3 * i.spmv_avg_ms:.2f} us, sums {1e3 * i.gather_avg_ms:.2f} us, update {1e3 * i.update_avg_ms:.2f} us; layout {k}", flush=True)
    c.close()
