import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen
for dim, nx, order in ((3, 60, 1), (2, 256, 2)):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0); c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(order)
    c.set_operator(capi.dt() - capi.laplacian())
    nq = c.quadrature_nodes().shape[0]
    for m in (1, 2, 21):
        t0 = time.perf_counter(); c.set_forcing(np.ones((nq, m)) if m > 1 else np.ones(nq)); t_sf = time.perf_counter() - t0
        c.init(); c.synchronize()
        t0 = time.perf_counter(); c.init(); c.synchronize(); t_init = time.perf_counter() - t0
        t0 = time.perf_counter(); F = c.force(m); t_f = time.perf_counter() - t0
        print(f"{dim}-D P{order} nx {nx}: {cells.shape[0]} cells, {m} forcing columns: set_forcing {1e3*t_sf:.1f} ms, init {1e3*t_init:.2f} ms, force() {1e3*t_f:.1f} ms", flush=True)
    c.close()
