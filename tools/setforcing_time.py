"""Cost of fdapde_set_forcing on C3 (host permutation + upload + k_visit_load_coeffs) -- what a caller pays per new forcing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
ctx = capi.Context(0)
ctx.mesh_upload(*meshgen.unit_cube(int(os.environ.get("NX", "119")))); nd = ctx.dofs_build(1)
u_exact, f = meshgen.manufactured(3)
fq = f(ctx.quadrature_nodes())
for _ in range(3):
    t = time.perf_counter(); ctx.set_forcing(fq); ctx.synchronize(); print(f"set_forcing: {(time.perf_counter() - t) * 1e3:.1f} ms")
