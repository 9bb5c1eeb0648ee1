"""Knob sweep of the multi-launch SpMV at the size above the single launch (3-D P1, nx = 200: 8.1 M DOFs): fdapde_bench_spmv (the SpMV kernel the
multi-launch CG runs, HIP-event timed) under the tuning knobs -- the priced A/B behind extra.large_8p1M.  usage: large_ab.py [nx]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nodes, cells, bnd = meshgen.unit_cube(nx)
_, f = meshgen.manufactured(3)


def variant(label, knobs):
    """a FRESH context per variant (knobs set before the layout is built: no state of an earlier variant leaks into the next)"""
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian())
    c.set_forcing(fq)
    c.set_dirichlet(np.zeros(nd))
    try:
        for k, v in knobs:
            c.tune(k, v)
        c.solver_prepare(True)
        c.init()
        c.solve(rtol=1e-1)   # (prepares the scaled system the benchmark kernel reads)
        ms, _ = c.bench_spmv(reps=40)
        _, _, st = c.solver_layout(True)
        print(f"{label:34s} {1e3 * ms:8.1f} us   {st / 1e6:8.1f} MB   {st / (ms * 1e-3) / 1e12:5.2f} TB/s   frac {st / (ms * 1e-3) / 8e12:5.3f}", flush=True)
    except capi.FdapdeError as e:
        print(f"{label}: {e}", flush=True)
    c.close()


c0 = capi.Context(0)
c0.mesh_upload(nodes, cells, bnd)
c0.dofs_build(1)
fq = f(c0.quadrature_nodes())
c0.close()
print(f"3-D P1 Laplacian, nx {nx}: SpMV of the multi-launch CG (fdapde_bench_spmv, 40 launches, HIP events), bytes = the kernel's own layout")
variant("default", [])
for key, vals in (("spmv_team", (8,)), ("spmv_unroll", (2, 8)), ("spmv_ntv", (0, 1)), ("spmv_c16", (0,)), ("spmv_deep", (1,)), ("spmv_bpx", (64, 512)),
                  ("spmv_variant", (0,)), ("blocked", (2,))):
    for v in vals:
        variant(f"{key} = {v}", [(key, v)])
variant("spmv_team = 8, spmv_unroll = 8", [("spmv_team", 8), ("spmv_unroll", 8)])
