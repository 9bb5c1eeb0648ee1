#!/usr/bin/env python3
"""Wall time of PDE::init() and PDE::solve() (fdapde_init / fdapde_solve) on small 2-D systems -- the sizes of the reference's own test
meshes -- after the set-up: where fixed per-call costs matter more than bandwidth."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(nx, order, adr=False):
    nodes, cells, bnd = meshgen.unit_square(nx)
    _, f = meshgen.manufactured(2)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    c.set_operator(-capi.laplacian() + capi.advection((1.0, 0.5)) + capi.reaction(1.0) if adr else -capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    for _ in range(3):
        c.init()
        c.solve(rtol=1e-10)
    c.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        c.init()
    c.synchronize()
    t_init = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        i = c.solve(rtol=1e-10)
    c.synchronize()
    t_solve = (time.perf_counter() - t0) / reps
    print(f"2-D P{order} nx {nx}{' ADR (BiCGStab)' if adr else ''}: {nd} DOFs  init {1e3 * t_init:.3f} ms   solve {1e3 * t_solve:.3f} ms ({i.iters} iterations, launch {1e3 * i.launch_ms:.0f} us, "
          f"G={c.solver_layout_kind(True)['workgroups']})", flush=True)
    c.close()


if __name__ == "__main__":
    for nx, order in ((16, 1), (32, 1), (60, 1), (60, 2), (128, 1)):
        run(nx, order)
    for nx in (16, 32):
        run(nx, 1, adr=True)
