#!/usr/bin/env python3
"""Random SEQUENCES of boundary calls on one context -- new operator, new forcing, new Dirichlet data, another order, another mesh, init, solve, a handle solve in
between, a clone taking over -- and after every solve the same problem once more on a FRESH context: whatever the long-lived context cached (coefficient
slots, row statistics, solver layouts and their column tables, the remembered CG breakdown, scaled copies) must not show in the answer (1e-9).
usage: fuzz_sequence.py [sequences] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402

# FUZZ_SEQ_PMG=1: the long-lived context's open method takes the two-level solver (eng_pmg.hip) for every order-2 system -- its coarse level, the coarse operator,
# the blocked-ELL fill and the damping it caches must follow every change --, the fresh context keeps the Jacobi-preconditioned stages
PMG = bool(os.environ.get("FUZZ_SEQ_PMG"))
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)


class Problem:
    def __init__(self):
        self.new_mesh()

    def new_mesh(self):
        self.dim = int(rng.integers(2, 4))
        self.nx = int(rng.integers(3, 20)) if self.dim == 2 else int(rng.integers(2, 7))
        self.seed = int(rng.integers(1, 1 << 30))
        self.order = int(rng.integers(1, 3))
        self.new_operator()
        self.fseed = int(rng.integers(1, 1 << 30))
        self.gseed = int(rng.integers(1, 1 << 30))
        self.gkind = rng.choice(["zero", "nonzero"])

    def mesh(self):
        return meshgen.unit_square(self.nx, seed=self.seed) if self.dim == 2 else meshgen.unit_cube(self.nx, seed=self.seed)

    def new_operator(self):
        self.okind = rng.choice(["lap", "reac", "adr", "diff", "diffn", "difff", "reacfield", "indef"])
        self.oseed = int(rng.integers(1, 1 << 30))

    def operator(self, rows):
        r = np.random.default_rng(self.oseed)
        N = self.dim
        if self.okind == "lap":
            return -capi.laplacian()
        if self.okind == "reac":
            return -capi.laplacian() + capi.reaction(float(r.uniform(0.1, 3.0)))
        if self.okind == "adr":
            return -capi.laplacian() + capi.advection(r.uniform(-1, 1, N)) + capi.reaction(float(r.uniform(0.1, 2.0)))
        if self.okind == "diff":
            L = r.uniform(-0.3, 0.3, (N, N))
            return -capi.diffusion(L @ L.T + np.eye(N)) + capi.reaction(float(r.uniform(0.0, 1.0)))
        if self.okind == "diffn":   # not symmetric, no advection: the reference's mirrored lower triangle
            return -capi.diffusion(np.eye(N) + r.uniform(-0.3, 0.3, (N, N))) + capi.reaction(float(r.uniform(0.0, 1.0)))
        if self.okind == "difff":
            L = r.uniform(-0.3, 0.3, (rows, N, N))
            K = np.einsum("qij,qkj->qik", L, L) + np.eye(N)[None]
            return -capi.diffusion_field(K.reshape(rows, N * N)) + capi.reaction(float(r.uniform(0.0, 1.0)))
        if self.okind == "reacfield":
            return -capi.laplacian() + capi.reaction_field(r.uniform(0.1, 2.0, rows))
        return -capi.laplacian() + capi.reaction(-float(r.uniform(5.0, 30.0)))   # symmetric indefinite (CG breaks down, BiCGStab takes over)

    def apply(self, c, what):
        """bring context c up to this problem; what: which parts changed ("all" for a fresh context)"""
        if what in ("all", "mesh"):
            nodes, cells, bnd = self.mesh()
            c.mesh_upload(nodes, cells, bnd)
        if what in ("all", "mesh", "order"):
            self.nd = c.dofs_build(self.order)
        rows = c.quadrature_nodes().shape[0]
        if what in ("all", "mesh", "order", "operator"):
            c.set_operator(self.operator(rows))
        if what in ("all", "mesh", "order", "forcing"):
            c.set_forcing(np.random.default_rng(self.fseed).standard_normal(rows))
        if what in ("all", "mesh", "order", "dirichlet"):
            _, _, coords = c.dofs_get()
            g = np.zeros(self.nd) if self.gkind == "zero" else coords @ np.random.default_rng(self.gseed).uniform(-1, 1, self.dim) + 0.1
            c.set_dirichlet(g)


fails = 0
checks = 0
pmg_solves = 0
for s in range(n_seq):
    p = Problem()
    c = capi.Context(0)
    p.apply(c, "all")
    if PMG:
        c.tune("pmg_auto_rows", 30), c.tune("pmg_auto_first_rows", 30), c.tune("pmg_setup_check", 1)
    log = ["all"]
    for step in range(int(rng.integers(6, 16))):
        act = rng.choice(["operator", "forcing", "dirichlet", "order", "mesh", "solve", "solve", "solve", "handle", "clone", "parabolic"])
        log.append(act)
        try:
            if act == "operator":
                p.new_operator(); p.apply(c, "operator")
            elif act == "forcing":
                p.fseed = int(rng.integers(1, 1 << 30)); p.apply(c, "forcing")
            elif act == "dirichlet":
                p.gseed = int(rng.integers(1, 1 << 30)); p.gkind = rng.choice(["zero", "nonzero"]); p.apply(c, "dirichlet")
            elif act == "order":
                p.order = 3 - p.order; p.apply(c, "order")
            elif act == "mesh":
                p.new_mesh(); p.apply(c, "mesh")
            elif act == "handle":
                c.init()
                c.lin_compute(capi.MAT_MASS, symmetric=True)
                if not (p.dim == 3 and p.order == 2):
                    c.lin_solve(np.random.default_rng(1).standard_normal(p.nd), rtol=1e-10)
            elif act == "parabolic":   # a time-dependent problem on the same space in between (forcing columns, M / dt + A), then back
                mt = int(rng.integers(2, 5))
                times = np.linspace(0.0, 0.1, mt + 1)
                rows = c.quadrature_nodes().shape[0]
                _, _, coords = c.dofs_get()
                F = np.random.default_rng(p.fseed).standard_normal((rows, mt + 1))
                u0 = np.prod(np.sin(np.pi * coords), axis=1)
                res = []
                for ctx in (c, capi.Context(0)):
                    if ctx is not c:
                        p.apply(ctx, "all")
                    ctx.set_operator(capi.dt() - capi.laplacian() + capi.reaction(0.5))
                    ctx.set_forcing(F)
                    ctx.init()
                    sol, _ = ctx.solve_parabolic(times, u0, dirichlet=np.zeros((p.nd, mt + 1)), rtol=1e-12)
                    res.append(sol)
                    if ctx is not c:
                        ctx.close()
                checks += 1
                err = np.linalg.norm(res[0] - res[1]) / max(np.linalg.norm(res[1]), 1e-300)
                if err > 1e-9:
                    fails += 1
                    print(f"FAIL sequence {s} after {log}: parabolic err {err:.3e}", flush=True)
                p.apply(c, "operator"); p.apply(c, "forcing"); p.apply(c, "dirichlet")
            elif act == "clone":
                d = c.clone()
                c.close()
                c = d
                if PMG:
                    c.tune("pmg_auto_rows", 30), c.tune("pmg_auto_first_rows", 30), c.tune("pmg_setup_check", 1)
            else:
                c.init()
                info = c.solve(rtol=1e-12, raise_on_noconv=False)
                u = c.solution()
                f = capi.Context(0)
                p.apply(f, "all")
                f.init()
                info_f = f.solve(rtol=1e-12, raise_on_noconv=False)
                uf = f.solution()
                f.close()
                checks += 1
                pmg_solves += 1 if info.method_used == capi.SOLVER_PMG else 0
                err = np.linalg.norm(u - uf) / max(np.linalg.norm(uf), 1e-300)
                if info.converged != info_f.converged or (info.converged == 1 and err > 1e-9):
                    fails += 1
                    print(f"FAIL sequence {s} after {log}: dim {p.dim} P{p.order} nx {p.nx} {p.okind} g {p.gkind}: converged {info.converged}/{info_f.converged} "
                          f"method {info.method_used}/{info_f.method_used} iters {info.iters}/{info_f.iters} err {err:.3e}", flush=True)
        except Exception as e:   # noqa: BLE001
            fails += 1
            print(f"ERROR sequence {s} after {log}: dim {p.dim} P{p.order} nx {p.nx} {p.okind}: {e}", flush=True)
            break
    c.close()
    if s % 10 == 9:
        print(f"... {s + 1} sequences, {checks} solves compared with a fresh context, failures {fails}", flush=True)
print(f"{n_seq} sequences, {checks} solves compared with a fresh context" + (f" ({pmg_solves} of them through the two-level solver)" if PMG else "") + f", failures {fails}")
sys.exit(1 if fails else 0)
