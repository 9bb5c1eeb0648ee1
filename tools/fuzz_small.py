#!/usr/bin/env python3
"""Randomised check of fdapde_solve on small systems (the sizes of the reference's own meshes: what k_small_front, the one-workgroup launches and the
single-launch BiCGStab serve) against scipy's sparse LU of the system the product itself hands out (stiff() / force() after the solve: Dirichlet rows zeroed,
unit diagonal -- the reference's matrix, fem_solver_base.h:142-155).  Random mesh size, dimension, order, operator (symmetric / advection-diffusion-reaction),
Dirichlet data (none / zero / non-zero), forcing; every case solved three times (first solve of a layout, then the fused front, then after new data).
usage: fuzz_small.py [cases] [seed] [mid|tiny]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402
import scipy.sparse as sp   # noqa: E402
import scipy.sparse.linalg as spl   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
TINY = len(sys.argv) > 3 and sys.argv[3] == "tiny"   # 1 - 3 cells per axis: a handful of DOFs, most or all of them on the boundary
MID = len(sys.argv) > 3 and sys.argv[3] == "mid"   # systems of 3 000 - 60 000 DOFs: several workgroups, plain / symmetric storage, resident / streaming
worst = 0.0
fails = 0
for case in range(n_cases):
    dim = int(rng.integers(2, 4))
    order = int(rng.integers(1, 3))
    nx = int(rng.integers(3, 40)) if dim == 2 else int(rng.integers(2, 11))
    if MID:
        nx = int(rng.integers(60, 240)) if dim == 2 else int(rng.integers(14, 36))
    if TINY:
        nx = int(rng.integers(1, 4))
    if order == 2 and not TINY:
        nx = max(2, nx // 2)
    jit = float(os.environ.get("FUZZ_JITTER", "0.2"))   # (0.45: triangles close to degenerate; tetrahedra invert beyond ~0.25)
    for _try in range(50):   # (the generator refuses a jitter that inverts a cell: another seed then)
        try:
            nodes, cells, bnd = (meshgen.unit_square(nx, seed=int(rng.integers(1, 1 << 30)), jitter=jit) if dim == 2
                                 else meshgen.unit_cube(nx, seed=int(rng.integers(1, 1 << 30)), jitter=min(jit, 0.25)))
            break
        except AssertionError:
            jit = 0.5 * (jit + 0.2)   # (a large mesh at a large jitter always has an inverted cell somewhere: towards the default)
            continue
    if os.environ.get("FUZZ_SCALE"):   # the domain anywhere between a micron and ten kilometres across
        nodes = nodes * float(10.0 ** rng.uniform(-6, 4))
    kind = rng.choice(["laplace", "reaction", "adr", "diffusion"])
    bc = rng.choice(["none", "zero", "nonzero"])
    if os.environ.get("FUZZ_SCALE") and bc == "none":
        bc = "nonzero"   # (without a Dirichlet DOF the condition number goes with the domain size: -Lap + c on a micron-sized domain is singular to 1e-12, and
                         #  an iterative solve to rtol is then kappa x rtol away from the LU solution -- conditioning, not what this run looks for)
    if bc == "none" and kind == "laplace":
        kind = "reaction"   # (pure Neumann Laplace is singular)
    if bc != "none" and os.environ.get("FUZZ_PARTIAL_BC") and rng.integers(0, 2):   # Dirichlet data on a random part of the boundary (at least one node), Neumann elsewhere
        keep = (rng.uniform(0, 1, bnd.shape[0]) < rng.uniform(0.05, 0.8)) & (bnd != 0)
        if keep.any():
            bnd = keep.astype(bnd.dtype)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd if bc != "none" else np.zeros_like(bnd))
    nd = c.dofs_build(order)
    _, bdofs, coords = c.dofs_get()
    if kind == "laplace":
        op = -capi.laplacian()
    elif kind == "reaction":
        op = -capi.laplacian() + capi.reaction(float(rng.uniform(0.1, 5.0)))
    elif kind == "diffusion":
        K = np.eye(dim) + 0.3 * np.diag(rng.uniform(0, 1, dim))
        op = -capi.diffusion(K) + capi.reaction(float(rng.uniform(0.1, 2.0)))
    else:
        op = -capi.laplacian() + capi.advection(rng.uniform(-1.5, 1.5, dim)) + capi.reaction(float(rng.uniform(0.1, 2.0)))
    if os.environ.get("FUZZ_SCALE"):
        op = float(10.0 ** rng.uniform(-4, 4)) * op
    c.set_operator(op)
    qn = c.quadrature_nodes()
    c.set_forcing(rng.standard_normal(qn.shape[0]))
    if bc == "zero":
        c.set_dirichlet(np.zeros(nd))
    elif bc == "nonzero":
        c.set_dirichlet(coords @ rng.uniform(-1, 1, dim) + 0.3)
    c.init()
    for rep in range(3):
        if rep == 2:   # new data through the same layout
            c.set_forcing(rng.standard_normal(qn.shape[0]))
            if bc == "nonzero":
                c.set_dirichlet(coords @ rng.uniform(-1, 1, dim) - 0.1)
            c.init()
        method = capi.SOLVER_AUTO
        if os.environ.get("FUZZ_METHODS"):   # a method named by the caller instead of the open one
            sym = kind != "adr"
            method = int(rng.choice([capi.SOLVER_AUTO, capi.SOLVER_CG, capi.SOLVER_CG_SR, capi.SOLVER_CG_FUSED, capi.SOLVER_BICGSTAB, capi.SOLVER_GMRES] if sym
                                    else [capi.SOLVER_AUTO, capi.SOLVER_BICGSTAB, capi.SOLVER_GMRES]))
        try:
            info = c.solve(method=method, rtol=1e-12, raise_on_noconv=False)
        except Exception as e:   # noqa: BLE001
            fails += 1
            print(f"ERROR case {case} rep {rep}: dim {dim} P{order} nx {nx} {nd} DOFs ({int((bdofs == 0).sum())} interior) {kind} bc {bc}: {e}", flush=True)
            break
        u = c.solution()
        rp, ci = c.pattern_get()
        A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
        ref = spl.spsolve(A.tocsc(), c.force())
        err = np.linalg.norm(u - ref) / max(np.linalg.norm(ref), 1e-300)
        worst = max(worst, err)
        ok = info.converged == 1 and err <= 1e-8
        if not ok:
            fails += 1
            print(f"FAIL case {case} rep {rep}: dim {dim} P{order} nx {nx} {nd} DOFs {kind} bc {bc} asked {method}: converged {info.converged} method {info.method_used} "
                  f"iters {info.iters} persistent {info.persistent} err {err:.3e}", flush=True)
    if case % 20 == 19:
        print(f"... {case + 1} cases, worst relative error so far {worst:.2e}, failures {fails}", flush=True)
    c.close()
print(f"{n_cases} cases x 3 solves: worst relative error against scipy LU {worst:.2e}, failures {fails}")
sys.exit(1 if fails else 0)
