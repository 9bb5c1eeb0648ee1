#!/usr/bin/env python3
"""The first-call path: mesh handed over -> first solution, phase by phase (wall clock around each C-ABI call, device idle at each mark).

What the reference does inside PDE(...), init() and solve() the first time (lagrangian_basis.h:94-136 enumerate_dofs,
fem_assembler.h:112-117 pattern through setFromTriplets, fem_linear_elliptic_solver.h:38-40 ordering + symbolic analysis) is here
dofs_build + set_forcing + solver_prepare ("set-up"); bench.py times the steps that follow.  FDAPDE_DEBUG_SETUP=1 prints the device
stages of dofs_build from inside the library.

    python tools/first_call.py [c3|c2|c5] [reps]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen   # noqa: E402

CASES = {"c3": (meshgen.unit_cube, 119, 1, 3), "c2": (meshgen.unit_square, 708, 1, 2), "c5": (meshgen.unit_cube, 87, 2, 3)}


def first_call(name, reps=2, quiet=False):
    gen, nx, order, dim = CASES[name]
    nodes, cells, bnd = gen(nx)
    _, f = meshgen.manufactured(dim)
    out = []
    for rep in range(reps):
        ctx = capi.Context(0)
        ph = {}

        def mark(key, t0):
            ctx.synchronize()
            ph[key] = 1e3 * (time.perf_counter() - t0)

        t0 = time.perf_counter(); ctx.mesh_upload(nodes, cells, bnd); mark("mesh_upload", t0)
        t_first = time.perf_counter()
        t0 = time.perf_counter(); nd = ctx.dofs_build(order); mark("dofs_build", t0)
        t0 = time.perf_counter(); qn = ctx.quadrature_nodes(); fq = f(qn); ph["host_sample_forcing_not_counted"] = 1e3 * (time.perf_counter() - t0)
        t_host = time.perf_counter() - t0
        if name == "c5":
            ctx.set_operator(-capi.laplacian() + capi.advection([1.0, 0.5, 0.25]) + capi.reaction(1.0))
        else:
            ctx.set_operator(-capi.laplacian())
        t0 = time.perf_counter(); ctx.set_forcing(fq); mark("set_forcing", t0)
        t0 = time.perf_counter(); ctx.set_dirichlet(np.zeros(nd)); mark("set_dirichlet", t0)
        t0 = time.perf_counter(); ctx.solver_prepare(True); mark("solver_prepare", t0)
        t0 = time.perf_counter(); ctx.init(); mark("init", t0)
        t0 = time.perf_counter(); info = ctx.solve(rtol=1e-10); mark("solve", t0)
        total = 1e3 * (time.perf_counter() - t_first - t_host)
        ph["first_call_ms"] = total
        ph["iters"], ph["persistent"], ph["n_dofs"] = int(info.iters), int(info.persistent), int(nd)
        # the steady state right after (what bench.py times)
        t0 = time.perf_counter(); ctx.init(); ctx.solve(rtol=1e-10); mark("second_step", t0)
        out.append(ph)
        if not quiet:
            print(f"== {name} rep {rep}: " + json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in ph.items()}), flush=True)
        ctx.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for n in which.split(","):
        first_call(n, reps)
