import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo")
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
for nx in (250, 500, 800):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(2)
    qn = c.quadrature_nodes()
    for name, op in (("adr", -capi.laplacian() + capi.advection([2.0, 1.0]) + capi.reaction(1.0)), ("lap", -capi.laplacian())):
        c.set_operator(op); c.set_forcing(np.sin(3 * qn[:, 0]) + qn[:, 1]); c.set_dirichlet(np.zeros(nd)); c.init()
        c.tune("pmg_auto", 0)
        res = []
        for m, mn in ((capi.SOLVER_PMG, "pmg"), (capi.SOLVER_AUTO, "jacobi")):
            c.solve(method=m, raise_on_noconv=False)
            t0 = time.perf_counter(); i = c.solve(method=m, raise_on_noconv=False); ms = 1e3 * (time.perf_counter() - t0)
            res.append(f"{mn}: conv {i.converged} method {i.method_used} iters {i.iters} {ms:.1f} ms")
            if mn == "pmg": u = c.solution()
            else: res.append(f"diff {float(np.abs(c.solution() - u).max()):.1e}")
        print(f"2-D P2 nx {nx} ({nd} DOFs) {name}: " + " | ".join(res), flush=True)
    c.close()
