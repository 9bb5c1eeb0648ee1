#!/usr/bin/env python3
"""Single-launch CG: layouts with late-import workgroups (knob persist_late 1: rows per thread from the row count alone; a workgroup whose
importing rows overflow the second half of its slots fetches its imports before its first pass) against the default (rows per thread
doubled until the importing rows fit the second half).  3-D P1 Laplacian over the sizes where the two differ."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(dim, nx):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    out = []
    for knob in (0, 1, 0, 1):
        c.tune("persist_late", knob)
        c.solve(rtol=1e-10)
        i = c.solve(rtol=1e-10)
        out.append((knob, i.iters, 1e3 * i.launch_ms / max(i.iters, 1), c.solver_layout_kind(True)))
    print(f"{dim}-D nx {nx}: {nd} DOFs  " + "  ".join(f"late={k}: {it} it {us:.2f} us/it R={lay['rows_per_thread']} kind={lay['kind']} sym={lay['sym']}" for k, it, us, lay in out), flush=True)
    c.close()


if __name__ == "__main__":
    for dim, nx in ((3, 40), (3, 56), (3, 64), (3, 72), (3, 80), (3, 90), (3, 100), (2, 708)):
        run(dim, nx)
