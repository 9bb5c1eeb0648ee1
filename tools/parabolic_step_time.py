#!/usr/bin/env python3
"""Time per step of the parabolic stepper (FEMLinearParabolicSolver::solve, fem_linear_parabolic_solver.h:37-72: one linear solve per time
step) and of repeated fdapde_lin_solve calls on a factor-once handle, over system sizes: where the per-solve overhead outside the kernels
(launches, the host's read of the outcome) matters.  2-D P1 heat equation on the unit square, m steps."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(nx, m):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(capi.dt() - capi.laplacian())
    qn = c.quadrature_nodes()
    times = np.linspace(0.0, 0.1, m + 1)
    c.set_forcing(np.zeros((qn.shape[0], m + 1)))
    _, _, coords = c.dofs_get()
    u0 = np.prod(np.sin(np.pi * coords), axis=1)
    c.init()
    c.solve_parabolic(times, u0, dirichlet=np.zeros((nd, m + 1)))
    c.synchronize()
    t0 = time.perf_counter()
    c.solve_parabolic(times, u0, dirichlet=np.zeros((nd, m + 1)))
    c.synchronize()
    t_par = (time.perf_counter() - t0) / m
    i = c.info()
    # handle: one matrix, many right-hand sides one after the other
    c.set_operator(-capi.laplacian())
    c.set_forcing(np.ones(qn.shape[0]))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    c.lin_compute(capi.MAT_MASS, symmetric=True)
    b = np.random.default_rng(0).standard_normal(nd)
    c.lin_solve(b)
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(m):
        c.lin_solve(b)
    c.synchronize()
    t_lin = (time.perf_counter() - t0) / m
    print(f"2-D P1 nx {nx}: {nd} DOFs  parabolic {1e3 * t_par:.3f} ms per step ({m} steps, last step {i.iters} iterations)   "
          f"handle (mass matrix) {1e3 * t_lin:.3f} ms per solve", flush=True)
    c.close()


if __name__ == "__main__":
    for nx in (16, 60, 128, 256, 512):
        run(nx, 20)
