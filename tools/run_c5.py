"""BASELINE config C5 on one GPU: 3-D P2 advection-diffusion-reaction (non-symmetric, BiCGStab), 87^3 x 6 = 3 951 018 tetrahedra."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "87"))
t = time.time(); nodes, cells, bnd = meshgen.unit_cube(nx); t_gen = time.time() - t
ctx = capi.Context(0)
t = time.time(); ctx.mesh_upload(nodes, cells, bnd); nd = ctx.dofs_build(2); t_setup = time.time() - t
s = ctx.sizes()
_, bd, coords = ctx.dofs_get()
b = np.array([1.0, 0.5, 0.25]); c = 1.0
pi = np.pi
u = lambda x: np.prod(np.sin(pi * x), axis=1)
def f(x):
    s_, c_ = np.sin(pi * x), np.cos(pi * x)
    grad = np.stack([pi * c_[:, 0] * s_[:, 1] * s_[:, 2], pi * s_[:, 0] * c_[:, 1] * s_[:, 2], pi * s_[:, 0] * s_[:, 1] * c_[:, 2]], axis=1)
    return 3 * pi**2 * u(x) + grad @ b + c * u(x)
qn = ctx.quadrature_nodes()
ctx.set_operator(-capi.laplacian() + capi.advection(b) + capi.reaction(c))
ctx.set_forcing(f(qn)); del qn
ctx.set_dirichlet(np.zeros(nd))
for i in range(2):
    ctx.init(); info = ctx.solve(rtol=1e-10)
sol = ctx.solution()
err = np.abs(sol - u(coords)).max()
print(f"C5 nx={nx}: cells {cells.shape[0]} dofs {nd} nnz {s['nnz']} edges {s['n_edges']} | meshgen {t_gen:.1f}s setup {t_setup:.1f}s | "
      f"assemble {info.t_assemble_ms:.2f} ms  solve {info.t_solve_ms:.2f} ms  iters {info.iters} method {info.method_used} relres {info.relres:.2e} | "
      f"max err {err:.2e} | DOF/s {nd / ((info.t_assemble_ms + info.t_solve_ms) * 1e-3):.3e}")
