#!/usr/bin/env python3
"""Jacobi-BiCGStab as ONE launch (kernels_persist_bicg.h) against the multi-launch kernels, same context, same system: iterations, us per
iteration, solve time.  Advection-diffusion-reaction, b = (1, 0.5, 0.25), c = 1."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(dim, nx, order):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    c.set_operator(-capi.laplacian() + capi.advection([1.0, 0.5, 0.25][:dim]) + capi.reaction(1.0))
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    out = {}
    for knob in (0, 1):
        c.tune("persist_bicg", knob)
        c.solve(rtol=1e-10)
        i = c.solve(rtol=1e-10)
        out[knob] = (i.iters, i.t_solve_ms, i.persistent, i.launch_ms)
    lay = c.solver_layout_kind(True)
    (it0, t0, _, _), (it1, t1, p1, l1) = out[0], out[1]
    print(f"{dim}-D P{order} nx {nx}: {nd} DOFs  multi-launch {it0} it, {t0:.2f} ms = {1e3 * t0 / it0:.1f} us/it   single launch ({p1}) {it1} it, "
          f"{t1:.2f} ms = {1e3 * t1 / max(it1, 1):.1f} us/it (launch {l1:.2f} ms)   layout {lay}", flush=True)
    c.close()


if __name__ == "__main__":
    for dim, nx, order in ((2, 250, 1), (2, 708, 1), (2, 1000, 1), (3, 40, 1), (3, 64, 1), (3, 100, 1), (2, 300, 2), (3, 30, 2), (3, 45, 2)):
        run(dim, nx, order)
