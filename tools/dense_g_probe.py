import sys, os, time, numpy as np
sys.path.insert(0, ".")
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for nx in (16, 32, 45):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0)); c.set_forcing(np.ones(3 * cells.shape[0])); c.init()
    c.tune("dense_after", 0)
    b = np.ones(nd)
    ts = []
    for _ in range(3):
        c.lin_compute(capi.MAT_STIFF)
        t0 = time.perf_counter(); c.lin_solve(b); ts.append(time.perf_counter() - t0)
    print(os.environ.get("FDAPDE_DENSE_ROWS_PER_WG"), nd, "build+solve ms", [round(1e3 * t, 2) for t in ts], flush=True)
    c.close()
