"""BASELINE config C2 on one GPU: 2-D P1 Laplacian, 708^2 x 2 = 1 002 528 jittered / diagonal-flipped triangles, 502 681 DOFs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "708"))
nodes, cells, bnd = meshgen.unit_square(nx)
u_exact, f = meshgen.manufactured(2)
ctx = capi.Context(0)
t = time.time(); ctx.mesh_upload(nodes, cells, bnd); nd = ctx.dofs_build(1); ctx.solver_prepare(True); t_setup = time.time() - t
if os.environ.get("PERSIST") == "0": ctx.tune("persist", 0)
ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(nd))
for i in range(3):
    t = time.time(); ctx.init(); info = ctx.solve(rtol=1e-10, time_spmv=32); wall = time.time() - t
_, _, coords = ctx.dofs_get()
err = np.abs(ctx.solution() - u_exact(coords)).max()
s = ctx.sizes()
alg = 12 * s["nnz"] + 4 * (nd + 1) + 16 * nd
print(f"C2 nx={nx}: cells {cells.shape[0]} dofs {nd} nnz {s['nnz']} | setup {t_setup:.2f}s | assemble {info.t_assemble_ms:.3f} ms  solve {info.t_solve_ms:.2f} ms  "
      f"wall {wall * 1e3:.2f} ms  persistent {info.persistent}  iters {info.iters} ({info.t_solve_ms / info.iters * 1e3:.1f} us/iter)  in-CG SpMV {info.spmv_avg_ms * 1e3:.1f} us = "
      f"{alg / (info.spmv_avg_ms * 1e-3) / 1e12:.2f} TB/s | max err {err:.2e} | DOF/s {nd / wall:.3e}")
