"""One column through the dense inverse at the reference's small sizes, by knob: tools/dense_column_probe.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for nx in (16, 21):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0)); c.set_forcing(np.ones(c.quadrature_nodes().shape[0])); c.init()
    c.tune("dense_after", 0)
    c.lin_compute(capi.MAT_STIFF)
    b = np.ones(nd)
    ref, _ = c.lin_solve(b)
    row = []
    for knob, val in (("dense_hostb", 0), ("dense_hostb", 1), ("dense_direct", 1)):
        c.tune("dense_hostb", 0); c.tune("dense_direct", 0); c.tune(knob, val)
        for _ in range(20): x, info = c.lin_solve(b)
        t0 = time.perf_counter()
        for _ in range(400): x, info = c.lin_solve(b)
        row.append((knob, val, round(1e6 * (time.perf_counter() - t0) / 400, 1), float(np.abs(x - ref).max())))
    print(nd, "DOFs, us per column:", row, flush=True)
    c.close()
