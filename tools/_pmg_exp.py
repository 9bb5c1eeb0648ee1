import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo")
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
nx = 64
nodes, cells, bnd = meshgen.unit_cube(nx)
c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(2)
_, bd, coords = c.dofs_get(); qn = c.quadrature_nodes()
times = np.linspace(0.0, 0.1, 6)
c.set_operator(capi.dt() - capi.laplacian() + capi.advection([1.0, 0.5, 0.25]))
c.set_forcing(np.stack([np.sin(2.0 * qn[:, 0]) * (1.0 + t) for t in times], axis=1)); c.init()
u0 = np.prod(np.sin(np.pi * coords), axis=1); G = np.zeros((nd, times.size))
res = []
for auto in (1, 0):
    c.tune("pmg_auto", auto)
    c.solve_parabolic(times, u0, G)
    t0 = time.perf_counter(); sol, info = c.solve_parabolic(times, u0, G); ms = 1e3 * (time.perf_counter() - t0)
    res.append(f"pmg_auto {auto}: method {info.method_used} conv {info.converged} iters {info.iters} wall {ms:.0f} ms device {info.t_solve_ms:.0f} ms")
    if auto: s1 = sol
print(f"3-D P2 nx {nx} ({nd} DOFs), 5 implicit Euler steps: " + " | ".join(res) + f" | max diff {float(np.abs(sol - s1).max()):.1e}", flush=True)
