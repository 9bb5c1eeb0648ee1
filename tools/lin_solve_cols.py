#!/usr/bin/env python3
"""fdapde_lin_solve with many right-hand sides on a factor-once handle (fdapde::SparseLU::solve(B); SMW and the downstream models call it
with blocks of columns): the columns side by side in one persistent launch (knob persist_cols 1) against one launch per column (0).
2-D P1 mass + stiffness matrix (SPD), n_rhs columns; wall time per column, identical bits expected."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(nx, n_rhs):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.init()
    c.lin_compute(capi.MAT_STIFF, symmetric=True)
    B = np.random.default_rng(0).standard_normal((nd, n_rhs))
    res = {}
    for knob in (0, 1):
        c.tune("persist_cols", knob)
        c.lin_solve(B)
        c.synchronize()
        t0 = time.perf_counter()
        X, info = c.lin_solve(B)
        c.synchronize()
        res[knob] = (time.perf_counter() - t0, X, info.iters)
    same = np.array_equal(res[0][1], res[1][1])
    lay = c.solver_layout_kind(False)
    print(f"2-D P1 nx {nx}: {nd} DOFs, {n_rhs} columns, G={lay['workgroups']}: one by one {1e6 * res[0][0] / n_rhs:.1f} us per column ({res[0][2]} iterations)"
          f"   side by side {1e6 * res[1][0] / n_rhs:.1f} us per column ({res[1][2]} iterations)   identical bits: {same}", flush=True)
    c.close()


if __name__ == "__main__":
    for nx, q in ((16, 64), (30, 64), (60, 64), (128, 32), (256, 16), (60, 3), (16, 200)):
        run(nx, q)
