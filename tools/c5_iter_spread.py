"""How much the Jacobi-BiCGStab iteration count of C5 (3-D P2 advection-diffusion-reaction, 5.36 M DOFs) moves with rounding, and whether another shadow
residual steadies it (712 -> 752 -> 785 iterations across rounds "with the summation order").  Each sample solves the SAME problem with
the right-hand side scaled by (1 + k 2^-48): the exact solution scales with it, the iteration path differs in its last bits only.
usage: c5_iter_spread.py [nx] [samples]   -> profiles/r6_c5_iter_spread.txt"""
import sys

import numpy as np

sys.path.insert(0, ".")
from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen, workloads

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 87
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nodes, cells, bnd = meshgen.unit_cube(nx)
c = capi.Context(0)
c.mesh_upload(nodes, cells, bnd)
nd = c.dofs_build(2)
del nodes, cells
c.set_operator(workloads.c5_operator(capi))
fq = workloads.c5_forcing(c.quadrature_nodes())
c.set_dirichlet(np.zeros(nd))
c.tune("pmg_auto", 0)   # (this tool is about the Jacobi-BiCGStab stage)
print(f"C5 at nx {nx}: {nd} DOFs; Jacobi-BiCGStab to rtol 1e-10; {samples} right-hand sides differing in the last bits per shadow residual")
for mode, name in ((0, "r0 (as published)"), (1, "pseudo-random"), (2, "r0, entries scaled by (0.5, 1.5)")):
    c.tune("bicg_shadow", mode)
    its, ms = [], []
    for k in range(samples):
        c.set_forcing(fq * (1.0 + k * 2.0**-48))
        c.init()
        info = c.solve(rtol=1e-10, raise_on_noconv=False)
        its.append(int(info.iters) if info.converged else -int(info.iters))
        ms.append(info.t_solve_ms)
    ok = [i for i in its if i > 0]
    spread = (max(ok) - min(ok)) / min(ok) if ok else float("nan")
    print(f"shadow {mode} {name:34s} iterations {its}  spread {100 * spread:5.1f} %  mean solve {np.mean(ms):7.1f} ms", flush=True)
c.close()
