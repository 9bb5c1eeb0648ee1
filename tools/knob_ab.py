#!/usr/bin/env python3
"""A/B of one solver knob on ONE box, alternating: C2, C3 and the wide form's system, single-launch CG.
usage: knob_ab.py <knob> <value_a> <value_b> [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

pkg = load_package()
capi = pkg.capi
from fdapde_core_amd import meshgen   # noqa: E402

knob, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cases = [("C2", 2, 708), ("C3", 3, 119), ("3-D 132^3", 3, 132), ("3-D 64^3", 3, 64)]
if os.environ.get("KNOB_AB_CASES"):
    want = os.environ["KNOB_AB_CASES"].split(",")
    cases = [c for c in cases if c[0] in want]
for name, dim, nx in cases:
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    out = {}
    for v in (va, vb):
        c.tune(knob, v)
        c.solve(rtol=1e-10)
    for r in range(reps):
        for v in (va, vb):
            c.tune(knob, v)
            i = c.solve(rtol=1e-10)
            out.setdefault(v, []).append((i.launch_ms, i.iters, i.relres, i.persistent))
    for v in (va, vb):
        ms = [o[0] for o in out[v]]
        print(f"{name}: {nd} DOFs  {knob}={v}: launch {min(ms):.3f} .. {max(ms):.3f} ms, {out[v][0][1]} iterations = {1e3 * min(ms) / out[v][0][1]:.2f} us/iteration, "
              f"relres {out[v][0][2]:.15e}, persistent {out[v][0][3]}", flush=True)
    c.close()
