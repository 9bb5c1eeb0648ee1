"""Blocked-ELL SpMV (k_spmv_blocked: x staged in LDS per block of ~4096 rows) against the CSR kernel on the compact pattern
(k_spmv_team2) inside the multi-launch Krylov solves: C5 (P2 ADR, BiCGStab) and C3 with the persistent CG switched off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen, workloads
for name in os.environ.get("CASES", "C5,C3").split(","):
    if name == "C5":
        nodes, cells, bnd = meshgen.unit_cube(int(os.environ.get("NX5", "87")))
        order, op, f = 2, workloads.c5_operator(capi), workloads.c5_forcing
    else:
        nodes, cells, bnd = meshgen.unit_cube(int(os.environ.get("NX3", "119")))
        order, op, f = 1, -capi.laplacian(), meshgen.manufactured(3)[1]
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(order)
    c.tune("persist", 0)
    c.tune("pmg_auto", 0)   # (the Krylov stage's SpMV layouts are what this tool compares)
    c.set_operator(op); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
    s = c.sizes(); alg = 12 * s["nnz"] + 4 * (nd + 1) + 16 * nd
    sols = {}
    for knob in (2, 0, 2, 0):   # 2 = blocked also for short rows
        c.tune("blocked", knob)
        t0 = time.perf_counter(); i = c.solve(rtol=1e-10, time_spmv=32); first = time.perf_counter() - t0
        i = min((c.solve(rtol=1e-10, time_spmv=32) for _ in range(2)), key=lambda z: z.t_solve_ms)
        sols[knob] = c.solution()
        print(f"{name} dofs {nd} blocked={knob}: solve {i.t_solve_ms:.2f} ms, {i.iters} it, {1e3 * i.t_solve_ms / i.iters:.1f} us/it, SpMV {1e3 * i.spmv_avg_ms:.1f} us = "
              f"{alg / (i.spmv_avg_ms * 1e-3) / 1e12:.2f} TB/s algorithmic (first call incl. layout {first:.2f} s) method {i.method_used}", flush=True)
    print(f"{name}: solutions agree to {np.abs(sols[2] - sols[0]).max():.2e}", flush=True)
    c.close()
