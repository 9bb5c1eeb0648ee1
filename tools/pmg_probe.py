"""FDAPDE_SOLVER_PMG (two-level: P2 fine level, P1 coarse level on the same mesh) next to the open method on C5's operator: tools/pmg_probe.py [nx ...]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
for nx in [int(a) for a in sys.argv[1:]] or (8, 16, 28):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(2)
    _, bd, coords = c.dofs_get()
    c.set_operator(workloads.c5_operator(capi))
    c.set_forcing(workloads.c5_forcing(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    out = []
    c.tune("pmg_auto", 0)   # ("jacobi": the open method's Jacobi-preconditioned stages -- what it takes below 1 M DOFs)
    for name, method in (("pmg", capi.SOLVER_PMG), ("jacobi", capi.SOLVER_AUTO)):
        info = c.solve(method=method, rtol=1e-10, raise_on_noconv=False)   # (the first call: set-up included)
        t0 = time.perf_counter(); info = c.solve(method=method, rtol=1e-10, raise_on_noconv=False); ms = 1e3 * (time.perf_counter() - t0)
        u = c.solution()
        err = float(np.abs(u - np.prod(np.sin(np.pi * coords), axis=1)).max())
        out.append(f"{name}: conv {info.converged} method {info.method_used} iters {info.iters} relres {info.relres:.1e} err-vs-analytic {err:.2e} {ms:.1f} ms")
        if name == "pmg": u_pmg = u
        else: out.append(f"max |u_pmg - u_jacobi| {float(np.abs(u_pmg - u).max()):.1e}")
    print(f"nx {nx}, {nd} DOFs: " + " | ".join(out), flush=True)
    c.close()
