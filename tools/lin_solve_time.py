#!/usr/bin/env python3
"""fdapde_lin_solve on a factor-once handle (mass matrix of a 2-D P1 space, nx given), ONE right-hand side per call -- a caller that cannot
batch columns: wall time per solve through the general path (upload, prologue kernels, launch, read-backs) and through the direct launch that
reads b and writes x and its outcome through pinned host memory itself (knob persist_direct); FDAPDE_DEBUG_TIMING=1 prints where the host side
spends it.   usage: tools/lin_solve_time.py <nx> [<nx> ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen
for nx in [int(a) for a in sys.argv[1:]] or [16]:
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian()); qn = c.quadrature_nodes(); c.set_forcing(np.ones(qn.shape[0])); c.set_dirichlet(np.zeros(nd)); c.init()
    c.lin_compute(capi.MAT_MASS, symmetric=True)
    b = np.random.default_rng(0).standard_normal(nd)
    out = {}
    for direct, spin in ((0, 0), (1, 0), (1, 2000)):
        c.tune("persist_direct", direct)
        c.tune("persist_direct_spin_us", spin)
        for _ in range(5): x, i = c.lin_solve(b)
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): x, i = c.lin_solve(b)
        c.synchronize()
        out[(direct, spin)] = (1e6 * (time.perf_counter() - t0) / 200, i.iters, x)
    g, d0, d1 = out[(0, 0)], out[(1, 0)], out[(1, 2000)]
    print(f"{nd} DOFs, layout {c.solver_layout_kind(False)}: general path {g[0]:.1f} us per solve ({g[1]} iterations); direct launch {d0[0]:.1f} us waiting for the "
          f"stream, {d1[0]:.1f} us spinning on the outcome record ({d1[1]} iterations); max |x_direct - x_general| {np.abs(d1[2] - g[2]).max():.2e}")
    c.close()
