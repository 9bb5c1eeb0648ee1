"""A/B of the assembly forms on C3 (3-D P1) and C5 (3-D P2): the row-owner sweep (default) against the element-wise scatter forms --
ATOMIC (lane per (cell,row), slot search, fp64 atomics), COLOURED (the same, one launch per colour), PARTITIONED (one workgroup per cell
partition, colours walked inside it, atomics only on rows shared between partitions, slot map streamed) and WAVE (one wavefront per
element, lane = (i, j, q), one launch per colour, slot map streamed; P2 in passes of 8 (i, j) pairs x 8 node lanes).  Device time of the stiffness values alone
(fdapde_assemble_operator is synchronous: events around it), median of 5 after one warm-up (which also pays the one-off index work of a
variant).  Writes its numbers to OUT=<path> (profiles/r2_asm_ab.json, profiles/r4_asm_ab.json)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
res = {}
for name, nx, order in (("C3", int(os.environ.get("NX3", "119")), 1), ("C5", int(os.environ.get("NX5", "87")), 2)):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    ctx = capi.Context(0)
    ctx.mesh_upload(nodes, cells, bnd); nd = ctx.dofs_build(order)
    op = -capi.laplacian() if order == 1 else workloads.c5_operator(capi)
    variants = [("rows", capi.ASSEMBLY_ROWS), ("atomic", capi.ASSEMBLY_ATOMIC), ("coloured", capi.ASSEMBLY_COLOURED),
                ("partitioned", capi.ASSEMBLY_PARTITIONED), ("wave", capi.ASSEMBLY_WAVE)]   # (round 4: the wavefront-per-element form takes P2 too)
    ref = None
    res[name] = {"cells": int(cells.shape[0]), "dofs": int(nd), "nnz": int(ctx.sizes()["nnz"]), "operator": "-laplacian" if order == 1 else "-laplacian + advection + reaction", "ms": {}}
    for vname, v in variants:
        t0 = time.perf_counter(); ctx.assemble_operator(capi.MAT_STIFF, op, v); t_first = time.perf_counter() - t0
        ts = []
        for i in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); ctx.assemble_operator(capi.MAT_STIFF, op, v); ts.append(1e3 * (time.perf_counter() - t0))
        vals = ctx.matrix_values(capi.MAT_STIFF)
        if ref is None: ref = vals
        err = float(np.abs(vals - ref).max() / np.abs(ref).max())
        res[name]["ms"][vname] = {"median_ms": float(np.median(ts)), "min_ms": float(min(ts)), "first_call_s": t_first, "rel_diff_vs_rows": err}
        print(name, vname, res[name]["ms"][vname], flush=True)
    ctx.close()
if os.environ.get("OUT"):
    json.dump(res, open(os.environ["OUT"], "w"), indent=1)
