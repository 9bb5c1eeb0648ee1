"""Interleaved A/B of k_cgf_update widths (and other solve-level knobs) on the C3 solve inside ONE process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "119"))
ctx = capi.Context(0)
ctx.mesh_upload(*meshgen.unit_cube(nx)); nd = ctx.dofs_build(1)
u_exact, f = meshgen.manufactured(3)
ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(nd)); ctx.init()
configs = [dict(c.split("=") for c in a.split(",")) for a in sys.argv[1:]] or [{"cgf_v": "4"}]
defaults = {"cgf_v": 8, "cgf_band": 1, "cgf_nt": 7, "cgf_lazy": 1, "spmv_bpx": 192, "cgf_split": 0}
res = {i: [] for i in range(len(configs))}
for rnd in range(5):
    for i, cfg in enumerate(configs):
        full = dict(defaults); full.update({k: int(v) for k, v in cfg.items()})
        for k, v in full.items():
            ctx.tune(k, v)
        info = ctx.solve(rtol=1e-10)
        res[i].append(info.t_solve_ms)
for i, cfg in enumerate(configs):
    r = np.array(res[i])
    print(f"{cfg}: solve median {np.median(r):.3f} ms  min {r.min():.3f}  ({info.iters} iterations)")
