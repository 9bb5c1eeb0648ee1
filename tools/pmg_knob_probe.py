"""The two-level solver's inner-solve budget on C5's operator: tools/pmg_knob_probe.py [nx] [cycle|outer|blocked|budget|one tol_exp maxit blocked outer] -- outer: flexible GMRES against BiCGStab per inner tolerance (knob pmg_outer); blocked: the fine operator through the CSR kernel / the blocked-ELL SpMV (knob pmg_blocked); budget: BiCGStab per (inner tolerance exponent, inner maxit): outer iterations, ms, and
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
names = ("FGMRES (additive)", "BiCGStab (additive)", "FGMRES (V(1,1) cycle)")
mode = sys.argv[2] if len(sys.argv) > 2 else "cycle"
# (inner tolerance exponent, inner maxit, fine operator blocked, outer method: 0 flexible GMRES / 1 BiCGStab)
grid = {"budget": ((2, 1000, 1, 1), (1, 1000, 1, 1), (3, 1000, 1, 1), (2, 10, 1, 1), (2, 20, 1, 1), (2, 30, 1, 1), (1, 20, 1, 1), (6, 1000, 1, 1)),
        "blocked": ((2, 1000, 0, 1), (2, 1000, 1, 1), (6, 1000, 0, 1), (6, 1000, 1, 1), (3, 1000, 0, 1), (3, 1000, 1, 1)),
        "one": (tuple(int(a) for a in sys.argv[3:7]),) if len(sys.argv) >= 7 else (),
        "outer": ((2, 1000, 1, 1), (2, 1000, 1, 0), (1, 1000, 1, 1), (1, 1000, 1, 0), (1, 12, 1, 0), (1, 8, 1, 0), (6, 1000, 1, 0)),
        # (outer 2: flexible GMRES around a V(1,1) cycle -- knob pmg_smooth --, 0: around the additive form)
        "cycle": ((1, 1000, 1, 0), (1, 1000, 1, 2), (2, 1000, 1, 2), (1, 16, 1, 2), (1, 1000, 0, 2), (6, 1000, 1, 2))}[mode]
for tol_exp, maxit, blocked, outer in grid:
    c.tune("pmg_inner_tol_exp", tol_exp); c.tune("pmg_inner_maxit", maxit); c.tune("pmg_blocked", blocked); c.tune("pmg_outer", 1 if outer == 1 else 0)
    c.tune("pmg_smooth", 1 if outer == 2 else 0)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); info = c.solve(method=capi.SOLVER_PMG, rtol=1e-10, raise_on_noconv=False); best = min(best, 1e3 * (time.perf_counter() - t0))
    print(f"RESULT inner rtol 1e-{tol_exp} maxit {maxit} blocked fine operator {blocked} outer method {names[outer]}: conv {info.converged} outer {info.iters} relres {info.relres:.1e} best of 3 {best:.1f} ms", flush=True)
    sys.stderr.flush()
c.close()
