"""Assembly only, at C5's size (3-D P2 advection-diffusion-reaction, 87^3 x 6 tetrahedra) or, with ORDER=1 NX=119, at C3's: fdapde_init a few
times, nothing else -- what tools/profile_asm.sh wraps in rocprofv3 passes.  FDAPDE_DEBUG_ASM=1 prints the launch configuration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen


def main():
    order = int(os.environ.get("ORDER", "2"))
    nx = int(os.environ.get("NX", "87" if order == 2 else "119"))
    reps = int(os.environ.get("REPS", "3"))
    nodes, cells, bnd = meshgen.unit_cube(nx)
    ctx = capi.Context(0)
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(order)
    s = ctx.sizes()
    op = -capi.laplacian() + capi.advection(np.array([1.0, 0.5, 0.25])) + capi.reaction(1.0) if order == 2 else -capi.laplacian()
    ctx.set_operator(op)
    qn = ctx.quadrature_nodes()
    ctx.set_forcing(np.prod(np.sin(np.pi * qn), axis=1))
    del qn
    for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]:
        ctx.tune(k, int(v))
    ts = []
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.init()
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    info = ctx.info()
    chk = float(np.abs(ctx.matrix_values(capi.MAT_STIFF)).sum()), float(np.abs(ctx.matrix_values(capi.MAT_MASS)).sum()), float(np.abs(ctx.force()).sum())
    print(f"asm order {order} nx {nx}: cells {cells.shape[0]} dofs {nd} nnz {s['nnz']} | init device {info.t_assemble_ms:.3f} ms, wall {min(ts):.3f} ms "
          f"(all: {' '.join(f'{t:.3f}' for t in ts)}) | checksums stiff {chk[0]:.17g} mass {chk[1]:.17g} force {chk[2]:.17g}")


if __name__ == "__main__":
    main()
