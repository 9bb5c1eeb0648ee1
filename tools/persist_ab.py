"""Single-launch CG (persist 1) against the multi-launch fused-update CG (persist 0) over a range of sizes: us per iteration,
iterations, and the in-kernel phase split (operator application incl. import wait / all-gather of the dots / vector update)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
cases = [(2, int(a)) for a in os.environ.get("SQ", "100,250,500,708,1000").split(",") if a] + \
        [(3, int(a)) for a in os.environ.get("CU", "30,60,90").split(",") if a]
for dim, nx in cases:
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
    s = c.sizes(); alg = 12 * s["nnz"] + 4 * (nd + 1) + 16 * nd
    for knob in (0, 1):
        c.tune("persist", knob)
        c.solve(rtol=1e-10)
        best = None
        for rep in range(3):
            i = c.solve(rtol=1e-10, time_spmv=32)
            if best is None or i.t_solve_ms < best.t_solve_ms: best = i
        i = best
        print(f"dim {dim} nx {nx} dofs {nd} persist={knob} ran_persistent={i.persistent}: solve {i.t_solve_ms:.2f} ms, {i.iters} it, "
              f"{1e3 * i.t_solve_ms / max(i.iters, 1):.2f} us/it | operator phase {1e3 * i.spmv_avg_ms:.2f} us = {alg / max(i.spmv_avg_ms * 1e-3, 1e-12) / 1e9:.0f} GB/s algorithmic"
              + (f" (mean over workgroups {1e3 * i.spmv_mean_ms:.2f} us) | gather {1e3 * i.gather_avg_ms:.2f} us update {1e3 * i.update_avg_ms:.2f} us" if i.persistent else ""), flush=True)
    c.close()
