#!/usr/bin/env python3
"""fdapde_eval_pointwise (PDE__::eval_basis pointwise, basis/lagrangian_basis.h:203-246 + point location): wall time per call over mesh and
location counts, first and repeated calls."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(dim, nx, order, n_locs):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    locs = np.random.default_rng(1).uniform(0.01, 0.99, size=(n_locs, dim))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = c.eval_pointwise(locs)
        ts.append(time.perf_counter() - t0)
    psi = out[0] if isinstance(out, tuple) else out
    # the C-ABI call alone (the wrapper adds the DOF table download and the scipy matrix)
    import ctypes as C

    flat = np.ascontiguousarray(locs.T).reshape(-1)
    cid = np.zeros(n_locs, dtype=np.int32)
    vals = np.zeros((n_locs, c.sizes()["n_basis"]))
    tc = []
    for _ in range(3):
        t0 = time.perf_counter()
        c._check(c.lib.fdapde_eval_pointwise(c._ctx, C.c_int64(n_locs), capi._dp(flat), capi._ip(cid), capi._dp(vals)))
        tc.append(time.perf_counter() - t0)
    print(f"{dim}-D P{order} nx {nx}: {cells.shape[0]} cells, {nd} DOFs, {n_locs} locations: " + " / ".join(f"{1e3 * t:.1f}" for t in ts) +
          f" ms per call, fdapde_eval_pointwise alone " + " / ".join(f"{1e3 * t:.2f}" for t in tc) + f" ms   (nnz of Psi {psi.nnz})", flush=True)
    c.close()


if __name__ == "__main__":
    for dim, nx, order, n in ((2, 60, 1, 1000), (2, 60, 2, 100000), (2, 708, 1, 1000000), (3, 30, 1, 100000), (3, 80, 1, 1000000)):
        run(dim, nx, order, n)
