"""Inversion time of the dense path by size (fdapde_lin_compute + first column with dense_after = 0): tools/dense_size_probe.py [nx ...]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for nx in [int(a) for a in sys.argv[1:]] or (16, 32, 45, 64):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0)); c.set_forcing(np.ones(3 * cells.shape[0])); c.init()
    c.tune("dense_after", 0); c.tune("dense_rows", 8192)
    b = np.ones(nd)
    ts = []
    for _ in range(3):
        c.lin_compute(capi.MAT_STIFF)
        t0 = time.perf_counter(); x = c.lin_solve(b); ts.append(time.perf_counter() - t0)
    print(nd, "DOFs: inversion + first column ms", [round(1e3 * t, 2) for t in ts], "method", c.info().method_used, flush=True)
    c.close()
