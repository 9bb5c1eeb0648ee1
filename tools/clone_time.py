#!/usr/bin/env python3
"""fdapde_ctx_clone: wall time of a clone (what a diverging copy of a PDE handle costs, include/fdapde_hip.hpp) next to the set-up it repeats,
on a small 2-D mesh and at C3's size.   usage: tools/clone_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen
for dim, nx, order in ((2, 60, 1), (2, 60, 2), (3, 24, 1), (3, 119, 1)):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    t0 = time.perf_counter()
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(order)
    t_setup = time.perf_counter() - t0
    c.set_operator(-capi.laplacian()); c.set_forcing(np.ones(c.quadrature_nodes().shape[0])); c.set_dirichlet(np.zeros(nd)); c.init(); c.solve()
    ts = []
    for _ in range(3):
        c.synchronize()
        t0 = time.perf_counter()
        d = c.clone()
        d.synchronize()
        ts.append(time.perf_counter() - t0)
        d.close()
    print(f"{dim}-D P{order} nx {nx}: {nd} DOFs: mesh_upload + dofs_build {1e3 * t_setup:.2f} ms (first of the process includes one-off costs); clone {1e3 * min(ts):.2f} ms "
          f"(all: {' '.join(f'{1e3 * t:.2f}' for t in ts)})", flush=True)
    c.close()
