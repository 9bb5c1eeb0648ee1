import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo")
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
for nx in (28, 87):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(2)
    for bscale in (1.0, 20.0):
        c.set_operator(-capi.laplacian() + capi.advection(bscale * np.array([1.0, 0.5, 0.25])) + capi.reaction(1.0))
        c.set_forcing(workloads.c5_forcing(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
        for sym in (0, 1):
            c.tune("pmg_coarse_sym", sym)
            c.solve(method=capi.SOLVER_PMG, raise_on_noconv=False)
            t0 = time.perf_counter(); i = c.solve(method=capi.SOLVER_PMG, raise_on_noconv=False); ms = 1e3 * (time.perf_counter() - t0)
            print(f"nx {nx} |b| x{bscale:g} coarse_sym {sym}: conv {i.converged} iters {i.iters} relres {i.relres:.1e} {ms:.1f} ms", flush=True)
    c.close()
