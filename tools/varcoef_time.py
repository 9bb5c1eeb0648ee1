import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package
capi = load_package().capi
from fdapde_core_amd import meshgen
def run(dim, nx, order):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(order)
    for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]:   # e.g. TUNE=asm_items=0
        c.tune(k, int(v))
    qn = c.quadrature_nodes()
    nq = qn.shape[0]
    f = np.ones(nq)
    K = np.tile(np.eye(dim).reshape(-1), (nq, 1)) * (1.0 + 0.1 * qn[:, :1])
    b = np.tile(np.array([1.0, 0.5, 0.25])[:dim], (nq, 1))
    cq = 1.0 + qn[:, 0]
    for name, op in (("const", -capi.laplacian() + capi.advection(np.array([1.0, 0.5, 0.25])[:dim]) + capi.reaction(1.0)),
                     ("reaction_field", -capi.laplacian() + capi.reaction_field(cq)),
                     ("all fields", capi.diffusion_field(K) * -1.0 if False else (-capi.diffusion_field(K)) + capi.advection_field(b) + capi.reaction_field(cq))):
        t0 = time.perf_counter(); c.set_operator(op); t_set = time.perf_counter() - t0
        c.set_forcing(f); c.set_dirichlet(np.zeros(nd))
        c.init(); c.synchronize()
        t0 = time.perf_counter(); c.init(); c.synchronize(); t_init = time.perf_counter() - t0
        print(f"{dim}-D P{order} nx {nx} ({cells.shape[0]} cells): {name}: set_operator {1e3*t_set:.1f} ms, init {1e3*t_init:.2f} ms", flush=True)
    c.close()
run(3, 60, 1)
run(3, 87, 2)
