"""How much of the two-level solver's run-to-run difference is the trajectory (BiCGStab around inexact coarse solves): C5's operator, the forcing perturbed at the 1e-3
level by a seeded random field, outer iterations / coarse iterations / ms with the fine operator through the CSR kernel and through the blocked-ELL SpMV.
tools/pmg_spread_probe.py [nx] [samples]"""
import sys, os, time, re, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 44
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nodes, cells, bnd = meshgen.unit_cube(nx)
c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(2)
c.set_operator(workloads.c5_operator(capi))
f0 = workloads.c5_forcing(c.quadrature_nodes())
c.set_dirichlet(np.zeros(nd))
rng = np.random.default_rng(5)
rows = {0: [], 1: []}
for s in range(ns):
    c.set_forcing(f0 * (1.0 + 1e-3 * rng.standard_normal(f0.shape)))
    c.init()
    for blocked in (0, 1):
        c.tune("pmg_blocked", blocked)
        c.solve(method=capi.SOLVER_PMG, rtol=1e-10, raise_on_noconv=False)
        t0 = time.perf_counter(); info = c.solve(method=capi.SOLVER_PMG, rtol=1e-10, raise_on_noconv=False); ms = 1e3 * (time.perf_counter() - t0)
        rows[blocked].append((int(info.iters), ms, int(info.converged)))
for blocked in (0, 1):
    it = np.array([r[0] for r in rows[blocked]]); ms = np.array([r[1] for r in rows[blocked]])
    print(f"nx {nx} ({nd} DOFs), fine operator {'blocked-ELL' if blocked else 'CSR'}: outer iterations {it.tolist()} mean {it.mean():.1f}; ms {np.round(ms, 1).tolist()} mean {ms.mean():.1f}; all converged {all(r[2] for r in rows[blocked])}", flush=True)
c.close()
