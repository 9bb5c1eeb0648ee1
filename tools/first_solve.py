#!/usr/bin/env python3
"""First-solve latency of small systems: solve #1 (lazy layout build), #2, #3 on the smoke problem and on a few
small meshes.  FDAPDE_DEBUG_TIMING=1 prints where the host side spends it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(name, nodes, cells, bnd, order=1):
    c = capi.Context(0)
    t0 = time.perf_counter()
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    t_build = time.perf_counter() - t0
    c.set_operator(-capi.laplacian())
    c.set_forcing(np.ones(c.sizes()["n_quadrature"] * cells.shape[0]))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    if os.environ.get("COOP") is not None:
        c.tune("persist_coop", int(os.environ["COOP"]))
    out = []
    for k in range(3):
        t0 = time.perf_counter()
        i = c.solve(rtol=1e-10)
        out.append((1e3 * (time.perf_counter() - t0), i.t_solve_ms, i.iters, i.persistent))
    print(f"{name}: {nd} DOFs, dofs_build {1e3 * t_build:.1f} ms; solves (wall ms, device ms, iters, persistent): " +
          "  ".join(f"({w:.2f}, {d:.2f}, {it}, {p})" for w, d, it, p in out), flush=True)
    c.close()


if __name__ == "__main__":
    run("warm-up", *meshgen.unit_square(10))
    run("square 10", *meshgen.unit_square(10))
    run("cube 8 (729)", *meshgen.unit_cube(8))
    run("cube 8 again", *meshgen.unit_cube(8))
    run("square 60", *meshgen.unit_square(60))
    run("cube 30", *meshgen.unit_cube(30))
    run("cube 48 (117k: device builder)", *meshgen.unit_cube(48))
