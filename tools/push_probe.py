#!/usr/bin/env python3
"""Timing probe: 100 iterations of the single-launch CG (method pinned, no fall-back) on C3 / a 3-D 64^3 system / C2: launch time per iteration and the
in-kernel phase stamps.  Run with the regular library and with a probe build (FDAPDE_HIP_LIB) whose imports are contiguous and list-free (wrong values:
the iteration count and the result mean nothing there)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402

for name, dim, nx in (("C2", 2, 708), ("3-D 64^3", 3, 64), ("C3", 3, 119)):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    c.tune("persist_time", 1)
    best = None
    for _ in range(4):
        i = c.solve(method=capi.SOLVER_CG_FUSED, rtol=1e-30, maxit=100, raise_on_noconv=False)
        if i.iters > 0 and (best is None or i.launch_ms / i.iters < best[0]):
            best = (i.launch_ms / i.iters, i.iters, i.spmv_avg_ms, i.spmv_mean_ms, i.gather_avg_ms, i.update_avg_ms, i.persistent)
    print(f"{name}: {nd} DOFs  {1e3 * best[0]:.2f} us/iteration over {best[1]} iterations (persistent {best[6]}); stamps: operator slowest {1e3 * best[2]:.2f} "
          f"mean {1e3 * best[3]:.2f}, gather {1e3 * best[4]:.2f}, update {1e3 * best[5]:.2f}", flush=True)
    c.close()
