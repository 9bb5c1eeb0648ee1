"""64 columns at once through the dense inverse: the pinned block across PCIe by DMA (knob dense_bulk 1) against the kernels reading / writing it
themselves: tools/dense_cols_probe.py [nx ...]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for nx in [int(a) for a in sys.argv[1:]] or (16, 32, 45, 64):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0)); c.set_forcing(np.ones(c.quadrature_nodes().shape[0])); c.init()
    c.tune("dense_after", 0)
    c.lin_compute(capi.MAT_STIFF)
    B = np.random.default_rng(1).standard_normal((nd, 64))
    row = []
    for g in (1, 0):
        c.tune("dense_bulk", g)
        for _ in range(3): X, info = c.lin_solve(B)
        t0 = time.perf_counter()
        for _ in range(10): X, info = c.lin_solve(B)
        row.append((g, round(1e6 * (time.perf_counter() - t0) / 10, 1)))
        if g == 1: X1 = X
    print(nd, "DOFs, 64 columns, us per call (dense_bulk, us):", row, "max diff", float(np.abs(X - X1).max()), flush=True)
    c.close()
