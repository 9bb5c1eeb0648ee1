#!/usr/bin/env python3
"""Repeated create / init / solve / clone / handle-solve / destroy of contexts on one device: free device memory must not drift (what a long-running
caller of the C ABI relies on).  Prints the free memory after 10, 60 and 120 rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
nodes, cells, bnd = meshgen.unit_cube(12)
_, f = meshgen.manufactured(3)
base = None
for it in range(120):
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1 + it % 2)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0) + (capi.advection(np.array([1., .5, .25])) if it % 3 == 0 else capi.reaction(0.5)))   # (never singular without Dirichlet rows)
    c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init(); c.solve()
    d = c.clone(); d.init(); d.solve(); d.close()
    c.lin_compute(capi.MAT_STIFF)   # (not the mass matrix: the reference's 3-D P2 rule has a negative weight, that mass matrix is singular to rounding)
    try:
        c.lin_solve(np.ones((nd, 3)))
    except capi.FdapdeError as e:
        i = c.info()
        print("round", it, "order", 1 + it % 2, "handle solve failed:", e, "iters", i.iters, "relres", i.relres, "method", i.method_used, "persistent", i.persistent, flush=True)
        raise
    c.close()
    if it in (9, 59, 119):
        fr = free(); base = base or fr
        print(it, "free MB", fr / 2**20, "delta vs round 9:", (base - fr) / 2**20, flush=True)
