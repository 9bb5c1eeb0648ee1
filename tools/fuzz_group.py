#!/usr/bin/env python3
"""Randomised check of the multi-device context (fdapde_ctx_create_multi, csrc/eng_group.hip) against a single-device context holding the same problem:
random dimension, order, mesh, number of "devices" (2 - 5, all GPU 0), form (left to the context / pinned), operator -- constant and SPACE-VARYING leaves
(the per-quadrature-node data are dealt to the ranks by cell id) --, forcing, Dirichlet data (none / zero / non-zero / on a random part of the boundary
through fdapde_dofs_set_boundary), then a random walk over the entry points: init, solve, getters, assemble_operator into the mass slot, spmv, lump, the
factor-once handle with given values, implicit Euler steps, new data, clone.  Everything compared with the single-device answer (<= 1e-8 solutions,
<= 1e-12 entries).  usage: fuzz_group.py [cases] [seed]"""
import os
import sys

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # (ranks sharing ONE GPU: a hardware queue per launch, see tests/conftest.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
fails, worst = 0, 0.0


def relerr(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def check(what, a, b, tol, case):
    global fails, worst
    e = relerr(np.asarray(a, dtype=float), np.asarray(b, dtype=float))
    worst = max(worst, e if tol >= 1e-9 else 0.0)
    if not (e <= tol):
        fails += 1
        print(f"case {case}: {what}: {e:.3e} > {tol:g}", flush=True)


for case in range(n_cases):
    dim = int(rng.integers(2, 4))
    order = int(rng.integers(1, 3))
    nx = int(rng.integers(6, 26)) if dim == 2 else int(rng.integers(4, 10))
    if order == 2:
        nx = max(4, nx // 2 + 2)
    nodes, cells, bnd = (meshgen.unit_square(nx, seed=int(rng.integers(1, 1 << 30))) if dim == 2 else meshgen.unit_cube(nx, seed=int(rng.integers(1, 1 << 30))))
    n_dev = int(rng.integers(2, 6))
    form = int(rng.integers(-1, 2))   # -1: left to the context
    one, grp = capi.Context(0), capi.Context(devices=[0] * n_dev)
    for c in (one, grp):
        c.mesh_upload(nodes, cells, bnd)
        nd = c.dofs_build(order)
    if form >= 0:
        grp.tune("group_form", form)
    dofs, bd, coords = one.dofs_get()
    qn = one.quadrature_nodes()
    nq_rows = qn.shape[0]
    bc = rng.choice(["none", "zero", "nonzero", "partial"])
    if bc == "partial":
        keep = (rng.uniform(0, 1, nd) < rng.uniform(0.2, 0.9)) & (bd != 0)
        if not keep.any():
            keep = bd != 0
        for c in (one, grp):
            c.dofs_set_boundary(keep.astype(np.uint8))
    kind = rng.choice(["lap", "lap+r", "adr", "diff_field", "react_field", "adv_field", "K"])
    if bc == "none" and kind in ("lap", "diff_field", "K"):
        kind = "lap+r"
    bvec = rng.uniform(-2, 2, dim)
    Kc = np.eye(dim) + 0.3 * np.diag(rng.uniform(0, 1, dim))
    Kq = np.tile(np.eye(dim).reshape(1, -1), (nq_rows, 1)) * (1.0 + 0.5 * np.sin(3 * qn[:, :1]))
    cq = 0.5 + qn[:, 0] ** 2
    bq = np.stack([1.0 + qn[:, 0], -0.5 + 0 * qn[:, 0], 0.3 * qn[:, -1]][:dim], axis=1)

    def mkop():
        if kind == "lap":
            return -capi.laplacian()
        if kind == "lap+r":
            return -capi.laplacian() + capi.reaction(float(0.7))
        if kind == "adr":
            return -capi.laplacian() + capi.advection(bvec) + capi.reaction(1.0)
        if kind == "diff_field":
            return -capi.diffusion_field(Kq) + capi.reaction(0.2)
        if kind == "react_field":
            return -capi.laplacian() + capi.reaction_field(cq)
        if kind == "adv_field":
            return -capi.laplacian() + capi.advection_field(bq) + capi.reaction(0.5)
        return -capi.diffusion(Kc) + capi.reaction(0.1)

    fq = np.sin(2 * qn[:, 0]) + qn[:, -1] ** 2
    g = None if bc == "none" else (np.zeros(nd) if bc == "zero" else 0.3 * coords[:, 0] - 0.2 * coords[:, -1])
    for c in (one, grp):
        c.set_operator(mkop())
        c.set_forcing(fq)
        c.set_dirichlet(g)
        c.init()
    tag = f"{case} (dim {dim} order {order} nx {nx} devices {n_dev} form {form} op {kind} bc {bc})"
    scale = max(1.0, np.abs(one.matrix_values(capi.MAT_STIFF)).max())
    check("stiff after init", grp.matrix_values(capi.MAT_STIFF) / scale, one.matrix_values(capi.MAT_STIFF) / scale, 1e-12, tag)
    check("mass after init", grp.matrix_values(capi.MAT_MASS), one.matrix_values(capi.MAT_MASS), 1e-12, tag)
    check("force after init", grp.force(), one.force(), 1e-12, tag)
    try:
        i1 = one.solve(rtol=1e-12, raise_on_noconv=False)
        ig = grp.solve(rtol=1e-12, raise_on_noconv=False)
    except capi.FdapdeError as e:
        fails += 1
        print(f"case {tag}: solve raised {e}", flush=True)
        one.close(), grp.close()
        continue
    if i1.converged != 1 or ig.converged != 1:
        if i1.converged != ig.converged:
            # (a chaotic BiCGStab may need its later stages on one side only: the multi-device context has no GMRES / dense stage -- report, do not fail)
            print(f"case {tag}: converged single {i1.converged} (method {i1.method_used}) / multi {ig.converged} (method {ig.method_used})", flush=True)
        one.close(), grp.close()
        continue
    check("solution", grp.solution(), one.solution(), 1e-8, tag)
    check("stiff after solve", grp.matrix_values(capi.MAT_STIFF) / scale, one.matrix_values(capi.MAT_STIFF) / scale, 1e-12, tag)
    check("force after solve", grp.force(), one.force(), 1e-12, tag)
    for _ in range(int(rng.integers(1, 5))):
        act = rng.choice(["spmv", "lump", "assemble", "handle", "parabolic", "newdata", "clone"])
        if act == "spmv":
            x = rng.standard_normal(nd)
            w = int(rng.integers(0, 2))
            check("spmv", grp.spmv(w, x), one.spmv(w, x), 1e-12, tag)
        elif act == "lump":
            check("lump", grp.lump(capi.MAT_MASS), one.lump(capi.MAT_MASS), 1e-12, tag)
        elif act == "assemble":
            op2 = capi.reaction_field(cq) if rng.integers(0, 2) else capi.reaction(2.0) - 0.1 * capi.laplacian()
            for c in (one, grp):
                c.assemble_operator(capi.MAT_MASS, op2)
            check("assemble_operator into the mass slot", grp.matrix_values(capi.MAT_MASS), one.matrix_values(capi.MAT_MASS), 1e-12, tag)
            for c in (one, grp):
                c.assemble_operator(capi.MAT_MASS, capi.reaction(1.0))
        elif act == "handle":
            for c in (one, grp):
                c.init()
            vals = one.matrix_values(capi.MAT_STIFF) + 3.0 * one.matrix_values(capi.MAT_MASS)
            sym = kind in ("lap", "lap+r", "diff_field", "react_field", "K")
            B = rng.standard_normal((nd, int(rng.integers(1, 4))))
            out = []
            for c in (one, grp):
                c.lin_compute(values=vals, symmetric=sym)
                X, info = c.lin_solve(B, rtol=1e-12)
                out.append(X)
            check("handle", out[1], out[0], 1e-8, tag)
        elif act == "parabolic":
            times = np.linspace(0.0, 0.05, int(rng.integers(3, 6)))
            F = np.stack([fq * (1.0 + t) for t in times], axis=1)
            G = None if bc == "none" else np.stack([(g if g is not None else np.zeros(nd)) * (1.0 + t) for t in times], axis=1)
            u0 = np.cos(coords[:, 0]) if G is None else G[:, 0]
            sols = []
            for c in (one, grp):
                c.set_operator(capi.dt() + mkop())
                c.set_forcing(F)
                c.init()
                sol, info = c.solve_parabolic(times, u0, G, rtol=1e-12)
                sols.append(sol)
                c.set_operator(mkop())
                c.set_forcing(fq)
                c.init()
            check("parabolic", sols[1], sols[0], 1e-8, tag)
        elif act == "newdata":
            fq2 = fq * rng.uniform(0.5, 2.0)
            for c in (one, grp):
                c.set_forcing(fq2)
                c.init()
                c.solve(rtol=1e-12, raise_on_noconv=False)
            check("solution after new forcing", grp.solution(), one.solution(), 1e-8, tag)
        else:
            for c in (one, grp):
                c.init()
                c.solve(rtol=1e-12, raise_on_noconv=False)
            twin = grp.clone()
            check("clone: solution", twin.solution(), grp.solution(), 0.0, tag)
            check("clone: stiff", twin.matrix_values(capi.MAT_STIFF), grp.matrix_values(capi.MAT_STIFF), 0.0, tag)
            twin.solve(rtol=1e-12, raise_on_noconv=False)
            check("clone: solves without init", twin.solution(), one.solution(), 1e-8, tag)
            twin.close()
    one.close(), grp.close()
print(f"cases {n_cases}  worst relative error {worst:.2e}  failures {fails}")
sys.exit(1 if fails else 0)
