"""What the automatic choice of the persistent CG's storage does over a range of 3-D / 2-D sizes: us per iteration and streamed bytes
with persist_sym 2 (auto), next to the forced plain (0) and symmetric (1) storage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
cases = [(3, int(a)) for a in os.environ.get("CU", "60,64,68,75,80,90").split(",") if a] + [(2, int(a)) for a in os.environ.get("SQ", "708,750,800").split(",") if a]
for dim, nx in cases:
    nodes, cells, bnd = meshgen.unit_cube(nx) if dim == 3 else meshgen.unit_square(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
    out = []
    for mode in (2, 0, 1):
        c.tune("persist_sym", mode)
        c.solve(rtol=1e-10)
        i = min((c.solve(rtol=1e-10) for _ in range(3)), key=lambda z: z.t_solve_ms)
        out.append(f"{('auto', 'plain', 'sym')[(2, 0, 1).index(mode)]} {1e3 * i.t_solve_ms / i.iters:.2f} us/it ({c.solver_layout()[2] / 1e6:.1f} MB)")
    print(f"dim {dim} nx {nx} dofs {nd}: " + " | ".join(out), flush=True)
    c.close()
