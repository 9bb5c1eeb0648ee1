#!/usr/bin/env python3
"""What a single launch beyond 2.1 M rows could gain.  The symmetric storage cannot exist there (16 B of
LDS per row), so the candidate is the PLAIN storage with x in HBM and 24 rows per thread.  Measured here: just below the cap (127^3 = 2.05 M DOFs)
the plain single launch, the symmetric one and the multi-launch path on the same system; just above it (133^3 = 2.35 M DOFs) the multi-launch path
the system takes today.  The plain form's time per row and iteration, plus the 16 B per row the x round trip would add at the plain form's own
streaming rate, extrapolated to the larger system, is the prototype's best case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen


def per_iteration(nx, knobs):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(3)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    del nodes, cells
    c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
    out = {}
    for name, kv in knobs:
        for k, v in kv:
            c.tune(k, v)
        c.solve(rtol=1e-10)
        best = min((c.solve(rtol=1e-10) for _ in range(2)), key=lambda i: i.t_solve_ms)
        lay = c.solver_layout(True)
        out[name] = dict(us=1e3 * (best.launch_ms if best.persistent else best.t_solve_ms) / max(best.iters, 1), iters=best.iters, persistent=best.persistent,
                         rows=lay[0], streamed_mb=lay[2] / 1e6)
        print(f"nx {nx}: {nd} DOFs ({lay[0]} interior rows) {name}: {out[name]['us']:.1f} us per iteration ({best.iters} iterations, persistent {best.persistent}, "
              f"layout streams {lay[2] / 1e6:.0f} MB per iteration)", flush=True)
    c.close()
    return out


below = per_iteration(126, [("symmetric single launch", [("persist", 1), ("persist_sym", 2)]), ("plain single launch", [("persist_sym", 0)]),
                            ("multi-launch", [("persist", 0)])])
above = per_iteration(132, [("multi-launch (what the system takes today)", [("persist", 1), ("persist_sym", 2)])])
p, m = below["plain single launch"], list(above.values())[0]
rate = p["streamed_mb"] / p["us"]   # MB per us the plain launch sustains over a whole iteration (hand-offs and dot gather included)
est = p["us"] * m["rows"] / p["rows"] + 16e-6 * m["rows"] / rate
print(f"plain single launch extrapolated to {m['rows']} rows: {p['us']:.1f} us x {m['rows'] / p['rows']:.3f} + x round trip {16e-6 * m['rows'] / rate:.1f} us = {est:.1f} us "
      f"per iteration against {m['us']:.1f} us on the multi-launch path: {100 * (1 - est / m['us']):.0f} % (best case; a 24-rows-per-thread instantiation keeps 12 loads "
      f"instead of 8 in flight per wavefront and pays for the registers)")
