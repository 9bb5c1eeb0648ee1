"""C3 `init` (stiff + force + mass) with the forcing as per-visit load coefficients computed inside init (asm_fq_block 1: coefficient
kernel + coalesced stream) against the sweep gathering the cell's samples itself (asm_fq_block 0: no extra kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "119"))
nodes, cells, bnd = meshgen.unit_cube(nx)
_, f = meshgen.manufactured(3)
ctx = capi.Context(0)
ctx.mesh_upload(nodes, cells, bnd); nd = ctx.dofs_build(1)
ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(nd))
for rep in range(2):
    for knob in (1, 0):
        ctx.tune("asm_fq_block", knob)
        ts = []
        for i in range(6):
            ctx.init(); ts.append(ctx.info().t_assemble_ms)
        print(f"asm_fq_block={knob}: init {np.median(ts[1:]):.3f} ms (min {min(ts[1:]):.3f})")
fr = ctx.force()
ctx.tune("asm_fq_block", 1); ctx.init()
print("force agrees:", np.abs(ctx.force() - fr).max())
