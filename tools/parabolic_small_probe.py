"""The parabolic stepper at the reference's own sizes (101 time points, K = M / dt + A): wall time of fdapde_solve_parabolic by size and knob.
tools/parabolic_small_probe.py [nx ...]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
for nx in [int(a) for a in sys.argv[1:]] or (16, 32, 45):
    nodes, cells, bnd = meshgen.unit_square(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    times = np.linspace(0.0, 1.0, 101)
    qn = c.quadrature_nodes()
    _, _, coords = c.dofs_get()
    c.set_operator(capi.dt() - capi.laplacian())
    c.set_forcing(np.stack([np.sin(np.pi * qn[:, 0]) * np.cos(t) for t in times], axis=1))
    c.init()
    u0 = np.sin(np.pi * coords[:, 0]) * np.sin(np.pi * coords[:, 1])
    G = np.zeros((nd, times.size))
    row = []
    for fold in (1, 0):
        c.tune("dense_fold", fold)
        c.solve_parabolic(times, u0, G)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); sol, info = c.solve_parabolic(times, u0, G); ts.append(time.perf_counter() - t0)
        row.append((fold, round(1e3 * min(ts), 2), round(info.t_solve_ms, 2), int(info.method_used)))
    print(nd, "DOFs, 101 points: (dense_fold, wall ms, device ms, method)", row, flush=True)
    c.close()
