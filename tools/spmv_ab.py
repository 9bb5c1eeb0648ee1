"""Interleaved A/B of SpMV launch configurations inside ONE process (medians over rounds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "119"))
ctx = capi.Context(0)
dim = int(os.environ.get("DIM", "3"))
ctx.mesh_upload(*(meshgen.unit_cube(nx) if dim == 3 else meshgen.unit_square(nx)))
ctx.dofs_build(int(os.environ.get("ORDER", "1")))
ctx.assemble_operator(capi.MAT_STIFF, -capi.laplacian())
if os.environ.get("SOLVE"):      # time the solver's compact Jacobi-scaled matrix (what CG streams) instead of stiff()
    u_exact, f = meshgen.manufactured(dim)
    ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(ctx.sizes()["n_dofs"]))
    ctx.init(); print("solve:", ctx.solve(rtol=1e-10).iters, "iterations")
configs = [dict(c.split("=") for c in a.split(",")) for a in sys.argv[1:]] or [{"spmv_variant": "2"}]
res = {i: [] for i in range(len(configs))}
defaults = {"spmv_variant": 2, "spmv_team": int(os.environ.get("TEAM", "8")), "spmv_unroll": 4, "spmv_bpx": 192, "spmv_ablate": 0, "spmv_c16": 1, "spmv_deep": 0, "spmv_ntv": -1}
for rnd in range(7):
    for i, cfg in enumerate(configs):
        full = dict(defaults); full.update({k: int(v) for k, v in cfg.items()})
        for k, v in full.items():
            ctx.tune(k, v)
        ms, by = ctx.bench_spmv(reps=100)
        res[i].append(ms * 1e3)
for i, cfg in enumerate(configs):
    r = np.array(res[i])
    print(f"{cfg}: median {np.median(r):.2f} us  min {r.min():.2f}  max {r.max():.2f}  -> {by/np.median(r)/1e3/8000:.3f} of 8 TB/s")
