"""Per-column cost of the factor-once handle (fdapde::SparseLU::solve, fdaPDE/utils/symbols.h:148-155) by system size: the GPU's Krylov run per column, the
GPU's dense inverse (build once + one product per column; kernels_dense.h) and scipy's SuperLU on the host (factorisation + back-substitution) --
the crossover table of VERDICT r5 item 3.  -> profiles/r6_handle_crossover.txt"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

sys.path.insert(0, ".")
from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen

print(f"{'n':>7} | {'Krylov us/col':>13} | {'dense build ms':>14} {'dense us/col':>12} {'us/col of 64':>12} {'max|I-AX|':>10} | {'LU factor ms':>12} {'LU us/col':>10} | columns to amortise the inversion vs Krylov")
for nx in (16, 32, 45, 64, 90, 128, 256):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.set_forcing(np.ones(3 * cells.shape[0]))
    c.init()
    vals = c.matrix_values(capi.MAT_STIFF)
    rp, ci = c.pattern_get()
    A = sp.csr_matrix((vals, ci, rp), shape=(nd, nd)).tocsc()
    t0 = time.perf_counter()
    lu = spl.splu(A)
    t_fac = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    B = rng.standard_normal((nd, 64))
    reps = 30
    t0 = time.perf_counter()
    for k in range(reps):
        ref = lu.solve(B[:, k % 64])
    t_lu = (time.perf_counter() - t0) / reps
    c.tune("dense_rows", 0)
    c.lin_compute(capi.MAT_STIFF)
    c.lin_solve(B[:, 0])
    t0 = time.perf_counter()
    for k in range(reps):
        x, info = c.lin_solve(B[:, k % 64])
    t_kry = (time.perf_counter() - t0) / reps
    dense = (float("nan"),) * 4
    if nd <= 8192:
        c.tune("dense_rows", 8192)
        c.tune("dense_after", 0)
        c.lin_compute(capi.MAT_STIFF)
        t0 = time.perf_counter()
        x, info = c.lin_solve(B[:, 0])
        t_first = time.perf_counter() - t0
        assert info.method_used == capi.SOLVER_DENSE, info.method_used
        t0 = time.perf_counter()
        for k in range(reps):
            x, info = c.lin_solve(B[:, k % 64])
        t_den = (time.perf_counter() - t0) / reps
        assert np.linalg.norm(x - lu.solve(B[:, (reps - 1) % 64])) <= 1e-9 * np.linalg.norm(x)
        c.lin_solve(B)
        t0 = time.perf_counter()
        for _ in range(5):
            X, _ = c.lin_solve(B)
        t_64 = (time.perf_counter() - t0) / 5 / 64
        dense = (1e3 * (t_first - t_den), 1e6 * t_den, 1e6 * t_64, info.relres)
    amort = dense[0] * 1e3 / max(1e6 * t_kry - dense[1], 1e-9) if nd <= 8192 else float("nan")
    print(f"{nd:7d} | {1e6 * t_kry:13.1f} | {dense[0]:14.2f} {dense[1]:12.1f} {dense[2]:12.1f} {dense[3]:10.1e} | {1e3 * t_fac:12.2f} {1e6 * t_lu:10.1f} | {amort:8.0f}", flush=True)
    c.close()
