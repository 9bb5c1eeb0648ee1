"""Where fdapde_init's time goes on C3: stiffness values alone (fdapde_assemble_operator), init without forcing, init with forcing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "119"))
ctx = capi.Context(0)
ctx.mesh_upload(*meshgen.unit_cube(nx)); nd = ctx.dofs_build(1)
u_exact, f = meshgen.manufactured(3)
ctx.set_operator(-capi.laplacian()); ctx.set_dirichlet(np.zeros(nd))

def timed(fn, reps=20):
    fn(); ctx.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t) / reps * 1e3

print(f"assemble_operator(stiff) alone : {timed(lambda: ctx.assemble_operator(capi.MAT_STIFF, -capi.laplacian())):.3f} ms")
print(f"assemble_operator(mass) alone  : {timed(lambda: ctx.assemble_operator(capi.MAT_MASS, capi.reaction(1.0))):.3f} ms")
ctx.set_forcing(None)
ctx.init(); print(f"init without forcing           : {np.median([ctx.init() or ctx.info().t_assemble_ms for _ in range(5)]):.3f} ms")
ctx.set_forcing(f(ctx.quadrature_nodes()))
ctx.init(); print(f"init with forcing              : {np.median([ctx.init() or ctx.info().t_assemble_ms for _ in range(5)]):.3f} ms")
