#!/usr/bin/env python3
"""fdapde_init with the mass matrix accumulated in the operator's sweep (knob asm_fuse_mass 1) against two sweeps (0): init time on C3
(and a 2-D case), bits of stiff / mass / force compared."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(dim, nx, order=1):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    res = {}
    for knob in (0, 1, 3, 0, 1, 3):
        c.tune("asm_fuse_mass", knob)
        ts = []
        for _ in range(5):
            c.init()
            ts.append(c.info().t_assemble_ms)
        res.setdefault(knob, []).append(float(np.median(ts)))
        if knob not in res.get("bits", {}):
            res.setdefault("bits", {})[knob] = (c.matrix_values(capi.MAT_STIFF), c.matrix_values(capi.MAT_MASS), c.force())
    b0 = res["bits"][0]
    same = all(np.array_equal(x, y) for k in (1, 3) for x, y in zip(b0, res["bits"][k]))
    print(f"{dim}-D P{order} nx {nx}: {nd} DOFs  init two launches {res[0]} ms, default rule {res[1]} ms, second pass in the launch {res[3]} ms, "
          f"identical bits: {same}", flush=True)
    c.close()


if __name__ == "__main__":
    run(3, 119)
    run(2, 708)
    run(3, 40, 2)
