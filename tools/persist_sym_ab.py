"""Symmetric storage of the persistent CG (tune persist_sym 1: in-block pairs stored once, transposed products through fixed-point LDS
accumulators) against the plain storage (persist_sym 0) on the same systems: us per iteration, iterations, phase split, bytes of the
ELL blocks, run-to-run bitwise reproducibility and the difference between the two solutions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
cases = [(2, int(a)) for a in os.environ.get("SQ", "708,1000,1400").split(",") if a] + \
        [(3, int(a)) for a in os.environ.get("CU", "60,90,119").split(",") if a]
for dim, nx in cases:
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(int(os.environ.get("ORDER", "1")))
    c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
    s = c.sizes(); alg = 12 * s["nnz"] + 4 * (nd + 1) + 16 * nd
    sols = {}
    for knob in (0, 1):
        c.tune("persist_sym", knob)
        c.solve(rtol=1e-10)
        u0 = c.solution().copy()
        best = None
        for rep in range(3):
            i = c.solve(rtol=1e-10, time_spmv=32)
            if best is None or i.t_solve_ms < best.t_solve_ms: best = i
        same = np.array_equal(u0, c.solution())
        sols[knob] = c.solution().copy()
        lay = c.solver_layout()
        i = best
        print(f"dim {dim} nx {nx} dofs {nd} persist_sym={knob} persistent={i.persistent} layout={lay}: solve {i.t_solve_ms:.2f} ms, {i.iters} it, "
              f"{1e3 * i.t_solve_ms / max(i.iters, 1):.2f} us/it | operator phase max {1e3 * i.spmv_avg_ms:.2f} us mean {1e3 * i.spmv_mean_ms:.2f} us "
              f"({alg / max(i.spmv_avg_ms * 1e-3, 1e-12) / 1e9:.0f} GB/s algorithmic) | gather {1e3 * i.gather_avg_ms:.2f} us update {1e3 * i.update_avg_ms:.2f} us | "
              f"bitwise reproducible {same}", flush=True)
    d = np.abs(sols[0] - sols[1]).max() / max(np.abs(sols[0]).max(), 1e-300)
    print(f"   max |u_sym - u_plain| / max |u| = {d:.3e}", flush=True)
    c.close()
