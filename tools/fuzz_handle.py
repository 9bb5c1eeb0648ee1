#!/usr/bin/env python3
"""Randomised check of the factor-once handle (fdapde_lin_compute / fdapde_lin_solve: fdapde::SparseLU, utils/symbols.h:133-160) and of the parabolic
stepper (fem_linear_parabolic_solver.h:37-72) against scipy: random mesh (2-D / 3-D, P1 / P2, a few hundred to a few thousand DOFs), matrix of the handle =
mass, stiff + mass, or an advection-diffusion-reaction operator (non-symmetric), 1 / 3 / 70 right-hand sides (the single zero-copy launch, columns side by
side, more columns than one launch takes), solved twice (in place the second time); implicit Euler over 2-6 steps with / without Dirichlet data.
usage: fuzz_handle.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402
import scipy.sparse as sp   # noqa: E402
import scipy.sparse.linalg as spl   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 13)
worst = 0.0
fails = 0


def csr(c, which, nd):
    rp, ci = c.pattern_get()
    return sp.csr_matrix((c.matrix_values(which), ci, rp), shape=(nd, nd))


for case in range(n_cases):
    dim = int(rng.integers(2, 4))
    order = int(rng.integers(1, 3))
    nx = int(rng.integers(4, 48)) if dim == 2 else int(rng.integers(3, 12))
    if order == 2:
        nx = max(2, nx // 2)
    nodes, cells, bnd = meshgen.unit_square(nx, seed=int(rng.integers(1, 1 << 30))) if dim == 2 else meshgen.unit_cube(nx, seed=int(rng.integers(1, 1 << 30)))
    what = rng.choice(["mass", "stiff_plus_mass", "adr", "parabolic"])
    if what == "mass" and dim == 3 and order == 2:
        what = "stiff_plus_mass"   # (the reference's 3-D P2 mass matrix is singular to rounding -- its 5-point rule has a negative weight: DESIGN 4.4)
    c = capi.Context(0)
    tag = f"case {case}: dim {dim} P{order} nx {nx} {what}"
    if os.environ.get("FUZZ_VERBOSE"):
        print("start", tag, flush=True)
    try:
        if what == "parabolic":
            use_bc = bool(rng.integers(0, 2))
            c.mesh_upload(nodes, cells, bnd if use_bc else np.zeros_like(bnd))
            nd = c.dofs_build(order)
            _, bdofs, coords = c.dofs_get()
            m = int(rng.integers(2, 7))
            c.set_operator(capi.dt() - capi.laplacian() + capi.reaction(float(rng.uniform(0.0, 2.0))))
            qn = c.quadrature_nodes()
            times = np.linspace(0.0, 0.05 * m, m + 1)
            c.set_forcing(rng.standard_normal((qn.shape[0], m + 1)))
            u0 = np.prod(np.sin(np.pi * coords), axis=1) + (0.0 if use_bc else 0.3)
            g = None
            if use_bc:
                g = np.tile((coords @ rng.uniform(-1, 1, dim))[:, None], (1, m + 1)) * np.linspace(0.0, 1.0, m + 1)[None, :]
                u0 = np.where(bdofs != 0, g[:, 0], u0)
            c.init()
            sol, info = c.solve_parabolic(times, u0, dirichlet=g, rtol=1e-12)
            # the reference's stepping with scipy: (M / dt + A) u_{k+1} = M u_k / dt + f_{k+1}, Dirichlet rows replaced
            A = csr(c, capi.MAT_STIFF, nd)
            M = csr(c, capi.MAT_MASS, nd)
            F = c.force().reshape(m + 1, nd).T   # (one column per time point)
            dt = times[1] - times[0]
            K = (M / dt + A).tolil()
            bd = np.flatnonzero(bdofs != 0) if use_bc else np.array([], dtype=int)
            for i in bd:
                K.rows[i], K.data[i] = [int(i)], [1.0]
            lu = spl.splu(K.tocsc())
            u = u0.copy()
            err = 0.0
            for k in range(m):
                rhs = M @ u / dt + F[:, k + 1]
                if use_bc:
                    rhs[bd] = g[bd, k + 1]
                u = lu.solve(rhs)
                err = max(err, np.linalg.norm(sol[:, k + 1] - u) / max(np.linalg.norm(u), 1e-300))
            ok = err <= 1e-8
            tag += f" steps {m} bc {use_bc} {nd} DOFs"
        else:
            c.mesh_upload(nodes, cells, bnd)
            nd = c.dofs_build(order)
            sym = what != "adr"
            if what == "adr":
                c.set_operator(-capi.laplacian() + capi.advection(rng.uniform(-1.5, 1.5, dim)) + capi.reaction(float(rng.uniform(0.5, 2.0))))
            else:
                c.set_operator(-capi.laplacian() + capi.reaction(1.0))
            c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
            c.init()
            if what == "mass":
                c.lin_compute(capi.MAT_MASS, symmetric=True)
                A = csr(c, capi.MAT_MASS, nd)
            else:
                c.lin_compute(capi.MAT_STIFF, symmetric=sym)
                A = csr(c, capi.MAT_STIFF, nd)
            lu = spl.splu(A.tocsc())
            err = 0.0
            for n_rhs in (1, 3, 70):
                B = rng.standard_normal((nd, n_rhs))
                if n_rhs == 3:
                    B[:, 1] = 0.0   # a zero column
                X, info = c.lin_solve(B[:, 0] if n_rhs == 1 else B, rtol=1e-12)
                X = X.reshape(nd, -1)
                ref = lu.solve(B)
                err = max(err, np.linalg.norm(X - ref.reshape(nd, -1)) / max(np.linalg.norm(ref), 1e-300))
            ok = err <= 1e-8
            tag += f" {nd} DOFs"
        worst = max(worst, err)
        if not ok:
            fails += 1
            print(f"FAIL {tag}: err {err:.3e}", flush=True)
    except Exception as e:   # noqa: BLE001
        fails += 1
        print(f"ERROR {tag}: {e}", flush=True)
    if case % 10 == 9:
        print(f"... {case + 1} cases, worst relative error so far {worst:.2e}, failures {fails}", flush=True)
    c.close()
print(f"{n_cases} cases: worst relative error against scipy LU {worst:.2e}, failures {fails}")
sys.exit(1 if fails else 0)
