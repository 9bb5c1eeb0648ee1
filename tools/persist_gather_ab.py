"""All-gather knobs of the persistent CG on C2 and C3: polling wavefronts (4 | 1) x pause between polls (0 .. 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for dim, nx in ((2, 708), (3, 119)):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
    c.solve(rtol=1e-10)
    for rep in range(2):
        for waves in (4, 1):
            for sl in (0, 1, 2, 3):
                c.tune("persist_gather_waves", waves); c.tune("persist_poll_sleep", sl)
                i = min((c.solve(rtol=1e-10) for _ in range(3)), key=lambda z: z.t_solve_ms)
                print(f"dim {dim} nx {nx} waves {waves} sleep {sl}: {1e3 * i.t_solve_ms / i.iters:.2f} us/it (operator {1e3 * i.spmv_avg_ms:.2f}, gather {1e3 * i.gather_avg_ms:.2f}) iters {i.iters}", flush=True)
    c.close()
