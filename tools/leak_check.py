#!/usr/bin/env python3
"""Device memory over repeated context lifetimes that touch every engine unit (space-varying operator, CG and BiCGStab solves, a handle with
several columns, point location, a second mesh on the same context, the parabolic stepper): free memory after each close()."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
nodes, cells, bnd = meshgen.unit_cube(24)
n2, c2, b2 = meshgen.unit_square(100)
base = None
for it in range(12):
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1 + it % 2)
    qn = c.quadrature_nodes(); nq = qn.shape[0]
    c.set_operator(-capi.laplacian() + capi.reaction_field(1.0 + qn[:, 0]))
    c.set_forcing(np.ones(nq)); c.set_dirichlet(np.zeros(nd)); c.init(); c.solve(rtol=1e-9)
    c.set_operator(-capi.laplacian() + capi.advection([1.0, 0.5, 0.25]) + capi.reaction(1.0)); c.init(); c.solve(rtol=1e-9)
    c.lin_compute(capi.MAT_STIFF, symmetric=False); c.lin_solve(np.ones((nd, 5)))   # (the advection-diffusion-reaction matrix just assembled)
    c.eval_pointwise(np.random.default_rng(0).uniform(0.1, 0.9, (1000, 3)))
    c.mesh_upload(n2, c2, b2); nd = c.dofs_build(2); c.set_operator(capi.dt() - capi.laplacian())
    qn = c.quadrature_nodes(); c.set_forcing(np.zeros((qn.shape[0], 4))); c.init()
    _, _, co = c.dofs_get()
    c.solve_parabolic(np.linspace(0, 0.1, 4), np.prod(np.sin(np.pi * co), axis=1), dirichlet=np.zeros((nd, 4)))
    c.eval_pointwise(np.random.default_rng(0).uniform(0.1, 0.9, (1000, 2)))
    c.close()
    f = free_mb()
    if it == 1: base = f
    print(it, round(f, 1), flush=True)
print("drift MB over 10 rounds:", round(base - f, 1))
