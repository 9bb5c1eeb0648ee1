"""fdapde_lin_solve with 8 columns on a C3-size matrix: batched (shared matrix stream) against column by column."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "119")); q = int(os.environ.get("Q", "8"))
ctx = capi.Context(0)
ctx.mesh_upload(*meshgen.unit_cube(nx)); nd = ctx.dofs_build(1)
ctx.set_operator(-capi.laplacian() + capi.reaction(50.0))
ctx.set_forcing(np.zeros(ctx.sizes()["n_quadrature"] * ctx.n_cells)); ctx.init()
ctx.lin_compute(capi.MAT_STIFF, symmetric=True)
B = np.random.default_rng(0).standard_normal((nd, q))
for mode in (1, 0, 1, 0):
    ctx.tune("multi_rhs", mode)
    t = time.time(); X, info = ctx.lin_solve(B, rtol=1e-10); wall = time.time() - t
    print(f"multi_rhs={mode}: {q} columns, n {nd}: device {info.t_solve_ms:.1f} ms (wall {wall*1e3:.0f} ms incl. transfers)  iterations (sum over batches / columns) {info.iters}  relres {info.relres:.1e}")
