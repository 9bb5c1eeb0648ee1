"""FDAPDE_SOLVER_PMG (two-level: P2 fine level, P1 coarse level on the same mesh) next to the open method on C5's operator: tools/pmg_probe.py [2d|2dsym] [nx ...]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
two_d = len(sys.argv) > 1 and sys.argv[1] in ("2d", "2dsym")   # 2d / 2dsym: the unit square, -Lap + b . grad + 1 (b = (1, 0.5)) / -Lap, forcing of the same analytic solution
sym = len(sys.argv) > 1 and sys.argv[1] == "2dsym"
for nx in [int(a) for a in sys.argv[(2 if two_d else 1):]] or (8, 16, 28):
    nodes, cells, bnd = meshgen.unit_square(nx) if two_d else meshgen.unit_cube(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(2)
    _, bd, coords = c.dofs_get()
    if two_d:
        q = c.quadrature_nodes()
        sx, sy, cx, cy = np.sin(np.pi * q[:, 0]), np.sin(np.pi * q[:, 1]), np.cos(np.pi * q[:, 0]), np.cos(np.pi * q[:, 1])
        c.set_operator(-capi.laplacian() if sym else -capi.laplacian() + capi.advection(np.array([1.0, 0.5])) + capi.reaction(1.0))
        c.set_forcing(2 * np.pi ** 2 * sx * sy if sym else (2 * np.pi ** 2 + 1.0) * sx * sy + np.pi * (cx * sy + 0.5 * sx * cy))
    else:
        c.set_operator(workloads.c5_operator(capi))
        c.set_forcing(workloads.c5_forcing(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    for kv in filter(None, os.environ.get("PMG_PROBE_TUNE", "").split(",")):   # e.g. PMG_PROBE_TUNE=pmg_inner_maxit=100,pmg_inner_tol_exp=2
        c.tune(kv.split("=")[0], int(kv.split("=")[1]))
    out = []
    c.tune("pmg_auto", 0)   # ("jacobi": the open method's Jacobi-preconditioned stages -- what it takes below 300 k DOFs)
    for name, method in (("pmg", capi.SOLVER_PMG), ("jacobi", capi.SOLVER_AUTO)):
        info = c.solve(method=method, rtol=1e-10, raise_on_noconv=False)   # (the first call: set-up included)
        ms = 1e9
        for _ in range(3):   # (best of three: a one-off of 30 - 80 ms now and then lands on the second call of a context)
            t0 = time.perf_counter(); info = c.solve(method=method, rtol=1e-10, raise_on_noconv=False); ms = min(ms, 1e3 * (time.perf_counter() - t0))
        u = c.solution()
        err = float(np.abs(u - np.prod(np.sin(np.pi * coords), axis=1)).max())
        out.append(f"{name}: conv {info.converged} method {info.method_used} iters {info.iters} relres {info.relres:.1e} err-vs-analytic {err:.2e} {ms:.1f} ms")
        if name == "pmg": u_pmg = u
        else: out.append(f"max |u_pmg - u_jacobi| {float(np.abs(u_pmg - u).max()):.1e}")
    print(f"nx {nx}, {nd} DOFs: " + " | ".join(out), flush=True)
    c.close()
