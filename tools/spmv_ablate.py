"""Diagnostic: back-to-back SpMV timing (fdapde_bench_spmv) under the FDAPDE_SPMV_* environment knobs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "119"))
nodes, cells, bnd = meshgen.unit_cube(nx)
ctx = capi.Context(0)
ctx.mesh_upload(nodes, cells, bnd)
nd = ctx.dofs_build(1)
ctx.assemble_operator(capi.MAT_STIFF, -capi.laplacian())
ms, by = ctx.bench_spmv(reps=200)
print(f"env={ {k: v for k, v in os.environ.items() if k.startswith('FDAPDE')} } spmv {ms*1e3:.2f} us  {by/ms/1e6:.0f} GB/s ({by/ms/1e6/8000:.3f} of 8 TB/s)")
