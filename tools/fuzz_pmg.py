#!/usr/bin/env python3
"""Randomised check of FDAPDE_SOLVER_PMG (csrc/eng_pmg.hip: order-2 space, the P1 space of the same mesh as coarse level) against scipy's sparse LU of the
system the product itself hands out (stiff() / force() after the solve: the reference's row-zeroed matrix, fem_solver_base.h:142-155).  Random dimension, mesh
size and jitter, operator (constant terms or coefficient fields; symmetric / advection-diffusion-reaction), Dirichlet data (none / zero / non-zero / on a random
part of the boundary); every case by name and through the open method with the switch-over size lowered, twice (new data through the same coarse level).
usage: fuzz_pmg.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402
import scipy.sparse as sp   # noqa: E402
import scipy.sparse.linalg as spl   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
worst, fails, most_iters = 0.0, 0, 0
for case in range(n_cases):
    dim = int(rng.integers(2, 4))
    nx = int(rng.integers(3, 36)) if dim == 2 else int(rng.integers(2, 10))
    jit = 0.2
    for _try in range(50):
        try:
            nodes, cells, bnd = (meshgen.unit_square(nx, seed=int(rng.integers(1, 1 << 30)), jitter=jit) if dim == 2
                                 else meshgen.unit_cube(nx, seed=int(rng.integers(1, 1 << 30)), jitter=jit))
            break
        except AssertionError:
            continue
    bc = rng.choice(["none", "zero", "nonzero", "partial"])
    if bc == "partial":
        keep = (rng.uniform(0, 1, bnd.shape[0]) < rng.uniform(0.05, 0.8)) & (bnd != 0)
        if keep.any():
            bnd = keep.astype(bnd.dtype)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd if bc != "none" else np.zeros_like(bnd))
    nd = c.dofs_build(2)
    _, bdofs, coords = c.dofs_get()
    qn = c.quadrature_nodes()
    nq = qn.shape[0]
    kind = rng.choice(["reaction", "adr", "diffusion", "fields"])
    if kind == "reaction":
        op = -capi.laplacian() + capi.reaction(float(rng.uniform(0.1, 5.0)))
    elif kind == "diffusion":
        op = -capi.diffusion(np.eye(dim) + 0.3 * np.diag(rng.uniform(0, 1, dim))) + capi.reaction(float(rng.uniform(0.1, 2.0)))
    elif kind == "adr":
        op = -capi.laplacian() + capi.advection(rng.uniform(-3.0, 3.0, dim)) + capi.reaction(float(rng.uniform(0.1, 2.0)))
    else:
        K = np.zeros((nq, dim * dim))
        for a in range(dim):
            K[:, a * dim + a] = 1.0 + rng.uniform(0, 1) * np.sin(3.0 * qn[:, a]) ** 2
        op = -capi.diffusion_field(K) + capi.advection_field(np.stack([rng.uniform(-2, 2) * (1.0 + qn[:, (a + 1) % dim]) for a in range(dim)], axis=1)) \
            + capi.reaction_field(0.2 + rng.uniform(0, 4) * qn[:, 0] ** 2)
    if os.environ.get("FUZZ_PMG_SETUP_CHECK"):
        c.tune("pmg_setup_check", 1)   # (the device-built transfer tables against the host loops)
    if os.environ.get("FUZZ_PMG_BLOCKED"):
        c.tune("pmg_blocked", int(os.environ["FUZZ_PMG_BLOCKED"]))   # (A/B of the fine operator's form)
    c.set_operator(op)
    for rep in range(2):
        c.set_forcing(rng.standard_normal(nq))
        if bc == "zero":
            c.set_dirichlet(np.zeros(nd))
        elif bc in ("nonzero", "partial"):
            c.set_dirichlet(coords @ rng.uniform(-1, 1, dim) + 0.3)
        c.init()
        c.tune("pmg_auto_rows", 1000000 if rep == 0 else 50)
        c.tune("pmg_auto_first_rows", 1000000 if rep == 0 else 50)
        info = c.solve(method=capi.SOLVER_PMG if rep == 0 else capi.SOLVER_AUTO, rtol=1e-11, raise_on_noconv=False)
        u = c.solution()
        rp, ci = c.pattern_get()
        A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
        ref = spl.spsolve(A.tocsc(), c.force())
        err = float(np.linalg.norm(u - ref) / max(np.linalg.norm(ref), 1e-300))
        ok = info.converged == 1 and err <= 1e-7 and (rep == 1 or info.method_used == capi.SOLVER_PMG)
        worst = max(worst, err if info.converged else 0.0)
        if info.method_used == capi.SOLVER_PMG:
            most_iters = max(most_iters, int(info.iters))
        if ok and info.method_used == capi.SOLVER_PMG and info.iters > 60:
            print(f"slow case {case} rep {rep}: dim {dim} nx {nx} {nd} DOFs {kind} bc {bc}: iters {info.iters} relres {info.relres:.1e} err {err:.1e}", flush=True)
        if not ok:
            fails += 1
            print(f"FAIL case {case} rep {rep}: dim {dim} nx {nx} {nd} DOFs {kind} bc {bc}: conv {info.converged} method {info.method_used} iters {info.iters} relres {info.relres:.1e} err {err:.1e}", flush=True)
    c.close()
    if (case + 1) % 20 == 0:
        print(f"... {case + 1} cases, worst relative error so far {worst:.2e}, most outer iterations {most_iters}, failures {fails}", flush=True)
print(f"{n_cases} cases x 2 solves: worst relative error against scipy LU {worst:.2e}, most outer iterations of the two-level solver {most_iters}, failures {fails}")
sys.exit(1 if fails else 0)
