"""The implicit-Euler stepper on a large order-2 space through the two-level solver and through the Jacobi-preconditioned stages: C5's operator, 3-D, `steps` steps of
dt, forcing constant in time, homogeneous Dirichlet data, zero initial condition.  tools/pmg_parabolic_probe.py [nx] [steps] [dt]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
nodes, cells, bnd = meshgen.unit_cube(nx)
c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(2)
c.set_operator(workloads.c5_operator(capi))
f = workloads.c5_forcing(c.quadrature_nodes())
c.set_forcing(np.tile(f[:, None], (1, steps + 1)))
times = dt * np.arange(steps + 1)
g = np.zeros((nd, steps + 1))
c.init()
res = {}
for name, auto in (("two-level", 1), ("jacobi", 0)):
    c.tune("pmg_auto", auto)
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); u, info = c.solve_parabolic(times, np.zeros(nd), dirichlet=g, rtol=1e-10); best = min(best, 1e3 * (time.perf_counter() - t0))
    res[name] = u
    print(f"nx {nx}, {nd} DOFs, {steps} steps of {dt}: {name}: method {info.method_used} converged {info.converged} iterations (all steps) {info.iters} worst relres {info.relres:.1e}, best of 2 {best:.1f} ms (host columns in and out included)", flush=True)
print(f"max |u_two_level - u_jacobi| over all columns {float(np.abs(res['two-level'] - res['jacobi']).max()):.1e} (max |u| {float(np.abs(res['jacobi']).max()):.2e})")
c.close()
