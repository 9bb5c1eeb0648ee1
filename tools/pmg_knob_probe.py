"""The two-level solver's inner-solve budget on C5's operator: tools/pmg_knob_probe.py [nx] [budget] -- the fine operator through the CSR kernel / the blocked-ELL SpMV (knob pmg_blocked), or with `budget` per (inner tolerance exponent, inner maxit): outer iterations, ms, and
(FDAPDE_DEBUG_SETUP on stderr) the coarse solves' iteration total."""
import sys, os, time, numpy as np
os.environ["FDAPDE_DEBUG_SETUP"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 87
nodes, cells, bnd = meshgen.unit_cube(nx)
c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(2)
c.set_operator(workloads.c5_operator(capi))
c.set_forcing(workloads.c5_forcing(c.quadrature_nodes()))
c.set_dirichlet(np.zeros(nd))
c.init()
full = len(sys.argv) > 2 and sys.argv[2] == "budget"
for tol_exp, maxit, blocked in (((2, 1000, 1), (1, 1000, 1), (3, 1000, 1), (2, 10, 1), (2, 20, 1), (2, 30, 1), (1, 20, 1), (6, 1000, 1)) if full else ((2, 1000, 0), (2, 1000, 1), (6, 1000, 0), (6, 1000, 1), (3, 1000, 0), (3, 1000, 1))):
    c.tune("pmg_inner_tol_exp", tol_exp); c.tune("pmg_inner_maxit", maxit); c.tune("pmg_blocked", blocked)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); info = c.solve(method=capi.SOLVER_PMG, rtol=1e-10, raise_on_noconv=False); best = min(best, 1e3 * (time.perf_counter() - t0))
    print(f"RESULT inner rtol 1e-{tol_exp} maxit {maxit} blocked fine operator {blocked}: conv {info.converged} outer {info.iters} relres {info.relres:.1e} best of 3 {best:.1f} ms", flush=True)
    sys.stderr.flush()
c.close()
