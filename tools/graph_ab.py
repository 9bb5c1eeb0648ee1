"""hipGraph replay of CG chunks (tune use_graph) against plain launches: interleaved, several problem sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for dim, nx in ((2, 60), (2, 708), (3, 40), (3, 119)):
    ctx = capi.Context(0)
    ctx.mesh_upload(*(meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx))); nd = ctx.dofs_build(1)
    u_exact, f = meshgen.manufactured(dim)
    ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(nd)); ctx.init()
    res = {0: [], 1: []}
    sols = {}
    for rnd in range(5):
        for g in (0, 1):
            ctx.tune("use_graph", g)
            info = ctx.solve(rtol=1e-10)
            res[g].append(info.t_solve_ms)
            sols[g] = (ctx.solution(), info.iters)
    same = np.array_equal(sols[0][0], sols[1][0]) and sols[0][1] == sols[1][1]
    print(f"dim {dim} nx {nx}: dofs {nd} iters {info.iters} | plain {np.median(res[0]):.3f} ms  graph {np.median(res[1]):.3f} ms  identical results: {same}")
    ctx.close()
