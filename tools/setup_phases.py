"""Wall time of the set-up phases (FDAPDE_DEBUG_SETUP=1 prints them from the library) for C3 / C2 / C5."""
import os, sys, time
os.environ["FDAPDE_DEBUG_SETUP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for name, gen, nx, order in (("C3", meshgen.unit_cube, 119, 1), ("C2", meshgen.unit_square, 708, 1), ("C5", meshgen.unit_cube, 87, 2)):
    if os.environ.get("ONLY") and os.environ["ONLY"] != name: continue
    nodes, cells, bnd = gen(nx)
    for rep in range(2):
        ctx = capi.Context(0)
        t0 = time.perf_counter(); ctx.mesh_upload(nodes, cells, bnd); t1 = time.perf_counter()
        nd = ctx.dofs_build(order); t2 = time.perf_counter()
        ctx.solver_prepare(True); t3 = time.perf_counter()
        print(f"== {name} rep {rep}: mesh_upload {1e3*(t1-t0):.1f} ms, dofs_build {1e3*(t2-t1):.1f} ms, solver_prepare {1e3*(t3-t2):.1f} ms", file=sys.stderr, flush=True)
        ctx.close()
