"""Assembly timing (t_assemble_ms of fdapde_init = stiff+force+mass) under FDAPDE_ASM_* env knobs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "119"))
ctx = capi.Context(0)
ctx.mesh_upload(*meshgen.unit_cube(nx))
nd = ctx.dofs_build(1)
ctx.set_operator(-capi.laplacian())
ctx.set_forcing(np.ones(4 * ctx.n_cells))
ts = []
for i in range(5):
    ctx.init()
    ts.append(ctx.info().t_assemble_ms)
print({k: v for k, v in os.environ.items() if k.startswith("FDAPDE")}, "init ms:", [round(t, 3) for t in ts])
