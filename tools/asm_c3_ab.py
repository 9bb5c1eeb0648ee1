#!/usr/bin/env python3
"""A/B of the P1 row-owner sweep on C3: fdapde_init (stiff + force + mass) with
  * the mass matrix in ONE pass with the operator (two accumulator ranges: asm_fuse_mass 2), as a second pass of the same launch (1: what the default
    rule picks on C3) and as a launch of its own (0);
  * the forcing samples from the block-cell copy (asm_fq_bc 1, default), gathered by cell id through the L2 (0), or reduced to one load coefficient per
    visit slot inside every init (asm_fq_block 1).
Median of 7 inits each (device time of the launch(es)), bits of stiff / mass / force compared with the default's.  -> JSON on stdout."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def main():
    nx = int(os.environ.get("NX", "119"))
    nodes, cells, bnd = meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(3)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian())
    fq = f(c.quadrature_nodes())
    out = {"workload": f"C3: 3-D P1, {nx}^3 x 6 = {cells.shape[0]} tetrahedra, {nd} DOFs; fdapde_init = stiff + force + mass", "variants": []}
    ref = None
    for name, fuse, fq_block, fq_bc in (("default: mass as second pass of the launch, forcing samples from the block-cell copy", 1, 0, 1),
                                        ("mass in ONE pass with the operator (two accumulator ranges in LDS: one block per CU instead of three)", 2, 0, 1),
                                        ("mass as a launch of its own", 0, 0, 1),
                                        ("forcing samples gathered by cell id through the L2 (no block-cell copy)", 1, 0, 0),
                                        ("one pass + forcing gathered by cell id", 2, 0, 0),
                                        ("forcing as one load coefficient per visit slot (k_visit_load_coeffs inside every init)", 1, 1, 1)):
        c.tune("asm_fuse_mass", fuse), c.tune("asm_fq_block", fq_block), c.tune("asm_fq_bc", fq_bc)
        c.set_forcing(fq)   # (the forcing's device layouts follow the knobs)
        ts = []
        for _ in range(8):
            c.init()
            ts.append(c.info().t_assemble_ms)
        bits = (c.matrix_values(capi.MAT_STIFF), c.matrix_values(capi.MAT_MASS), c.force())
        if ref is None:
            ref = bits
        same = all(np.array_equal(x, y) for x, y in zip(ref, bits))
        out["variants"].append({"variant": name, "asm_fuse_mass": fuse, "asm_fq_block": fq_block, "asm_fq_bc": fq_bc, "init_ms_median": float(np.median(ts[1:])),
                                "init_ms_min": float(min(ts[1:])), "identical_bits_with_default": bool(same)})
    c.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
