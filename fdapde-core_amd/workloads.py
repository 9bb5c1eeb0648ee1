"""The BASELINE.json configurations other than the headline one, as callable single-GPU runs (bench.py's `extra` results, the
full-size tests and tools/run_c2.py / run_c5.py share them).

  C2: 2-D P1 Laplacian, 708^2 x 2 = 1 002 528 jittered / diagonal-flipped triangles, 502 681 DOFs, Jacobi-PCG
  C5: 3-D P2 advection-diffusion-reaction  -Lap u + b.grad u + c u = f,  b = (1, 0.5, 0.25), c = 1, 87^3 x 6 = 3 951 018
      tetrahedra, 5 359 375 DOFs, Jacobi-BiCGStab.  3-D P2 numbering is build-defined (the reference does not compile
      LagrangianBasis<Triangulation<3,3>, 2>, lagrangian_basis.h:111-123): parity of C5 numbers is "unpinned" by construction.
"""
from __future__ import annotations

import time

import numpy as np

C5_B = np.array([1.0, 0.5, 0.25])
C5_C = 1.0


def c5_exact(x):
    return np.prod(np.sin(np.pi * x), axis=1)


def c5_forcing(x):
    pi = np.pi
    s_, c_ = np.sin(pi * x), np.cos(pi * x)
    u = np.prod(s_, axis=1)
    grad = np.stack([pi * c_[:, 0] * s_[:, 1] * s_[:, 2], pi * s_[:, 0] * c_[:, 1] * s_[:, 2], pi * s_[:, 0] * s_[:, 1] * c_[:, 2]], axis=1)
    return 3 * pi**2 * u + grad @ C5_B + C5_C * u


def c5_operator(capi):
    return -capi.laplacian() + capi.advection(C5_B) + capi.reaction(C5_C)


def device_arrays(ctx, capi):
    """what the device path holds for the current problem, for parity_against_oracle: solution() of the last solve, then stiff_ / mass_ / force_ as
    fdapde_init leaves them (init again: after a Dirichlet solve stiff() is the row-zeroed matrix, as in the reference)"""
    u = ctx.solution()
    ctx.init()
    rp, ci = ctx.pattern_get()
    dofs, bnd, _ = ctx.dofs_get()
    return {"u": u, "rowptr": rp, "colidx": ci, "dofs": dofs, "boundary": bnd, "stiff": ctx.matrix_values(capi.MAT_STIFF),
            "mass": ctx.matrix_values(capi.MAT_MASS), "force": ctx.force()}


def _timed_steps(ctx, steps, warmup, time_spmv, rtol):
    for _ in range(warmup):
        ctx.init()
        ctx.solve(rtol=rtol)
    ctx.synchronize()
    t0 = time.perf_counter()
    infos = []
    for _ in range(steps):
        ctx.init()
        infos.append(ctx.solve(rtol=rtol, time_spmv=time_spmv))
    ctx.synchronize()
    return (time.perf_counter() - t0) / steps, infos


def _summary(ctx, nd, wall, infos, u_exact, hbm_peak_gbps):
    """figures of one configuration.  Rates: `streamed` = the bytes the kernel's own layout moves per operator application (what an
    HBM fraction can be quoted on), `effective` = the algorithmic CSR bytes 12 nnz + 4 (n + 1) + 16 n over the same time (may exceed
    the peak where the layout stores less than CSR does; not a roofline fraction)"""
    info = infos[-1]
    s = ctx.sizes()
    alg = 12.0 * s["nnz"] + 4.0 * (nd + 1) + 16.0 * nd
    ni, nzi, streamed = ctx.solver_layout(True)
    iters = max(int(info.iters), 1)
    _, _, coords = ctx.dofs_get()
    err = float(np.abs(ctx.solution() - u_exact(coords)).max())
    out = {
        "dofs": int(nd), "nnz": int(s["nnz"]), "dof_per_s": nd / wall, "ms_per_step": 1e3 * wall,
        "t_assemble_ms": float(np.mean([i.t_assemble_ms for i in infos])), "t_solve_ms": float(np.mean([i.t_solve_ms for i in infos])),
        "iterations": int(info.iters), "us_per_iteration": 1e3 * float(info.t_solve_ms) / iters,
        "method": int(info.method_used), "relres": float(info.relres), "max_abs_error_vs_analytic": err,
        "persistent": int(info.persistent), "interior_rows": int(ni), "interior_nnz": int(nzi),
        "algorithmic_bytes_per_application": alg, "streamed_bytes_per_application": streamed,
        "layout": ctx.solver_layout_kind(True),
    }
    if info.persistent and out["layout"]["kind"] == 3:
        # ONE launch, blocks resident in LDS: an iteration moves exchanged granules only and is bound by two hand-off latencies (neighbour
        # import, dot all-gather), not by HBM -- no HBM fraction is quoted; effective_gbps = the CSR operator over the launch time, served from LDS
        launch_ms = float(np.mean([i.launch_ms for i in infos]))
        out.update(bound="latency", launch_ms=launch_ms, us_per_iteration_in_launch=1e3 * launch_ms / iters,
                   effective_gbps=alg * iters / (launch_ms * 1e-3) / 1e9,
                   operator_phase_us=1e3 * float(np.mean([i.spmv_avg_ms for i in infos])),
                   gather_avg_us=1e3 * float(np.mean([i.gather_avg_ms for i in infos])),
                   update_avg_us=1e3 * float(np.mean([i.update_avg_ms for i in infos])))
        return out
    if info.persistent:   # ONE launch per solve: rates over the launch duration (HIP events around the dispatch)
        launch_ms = float(np.mean([i.launch_ms for i in infos]))
        out.update(launch_ms=launch_ms, us_per_iteration_in_launch=1e3 * launch_ms / iters,
                   streamed_gbps=streamed * iters / (launch_ms * 1e-3) / 1e9, effective_gbps=alg * iters / (launch_ms * 1e-3) / 1e9,
                   operator_phase_us=1e3 * float(np.mean([i.spmv_avg_ms for i in infos])),
                   gather_avg_us=1e3 * float(np.mean([i.gather_avg_ms for i in infos])),
                   update_avg_us=1e3 * float(np.mean([i.update_avg_ms for i in infos])))
    else:                 # multi-launch Krylov iteration: the event-timed SpMV launches
        spmv_ms = float(np.mean([i.spmv_avg_ms for i in infos]))
        out.update(spmv_avg_us=1e3 * spmv_ms, spmv_timed=int(info.spmv_timed))
        if spmv_ms > 0:
            out.update(streamed_gbps=streamed / (spmv_ms * 1e-3) / 1e9, effective_gbps=alg / (spmv_ms * 1e-3) / 1e9)
    if "streamed_gbps" in out:
        out["bound"] = "hbm"
        out["frac"] = out["streamed_gbps"] / hbm_peak_gbps   # physical: layout bytes / time / peak
        out["effective_frac"] = out["effective_gbps"] / hbm_peak_gbps
    return out


def run_c2(capi, meshgen, nx=708, steps=2, warmup=1, time_spmv=32, rtol=1e-10, device=0, hbm_peak_gbps=8000.0, keep_arrays=False):
    nodes, cells, bnd = meshgen.unit_square(nx)
    u_exact, f = meshgen.manufactured(2)
    ctx = capi.Context(device)
    t0 = time.perf_counter()
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(1)
    ctx.solver_prepare(True)
    t_setup = time.perf_counter() - t0
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    ctx.set_dirichlet(np.zeros(nd))
    wall, infos = _timed_steps(ctx, steps, warmup, time_spmv, rtol)
    out = _summary(ctx, nd, wall, infos, u_exact, hbm_peak_gbps)
    arrays = device_arrays(ctx, capi) if keep_arrays else None   # (what the timed path produced, for the caller's oracle comparison; untimed)
    if out["persistent"]:
        # the multi-launch path (the HBM / L2 streaming kernels) on the same context, for comparison
        ctx.tune("persist", 0)
        wall_m, infos_m = _timed_steps(ctx, 1, 1, time_spmv, rtol)
        m = _summary(ctx, nd, wall_m, infos_m, u_exact, hbm_peak_gbps)
        out["multi_launch"] = {k: m[k] for k in ("dof_per_s", "ms_per_step", "iterations", "us_per_iteration", "spmv_avg_us", "streamed_gbps",
                                                 "effective_gbps", "frac", "effective_frac") if k in m}
        ctx.tune("persist", 1)
    out.update(workload=f"C2: 2-D P1 Laplacian, {nx}^2 x 2 = {cells.shape[0]} triangles, jitter 0.2h, diagonals flipped, ids permuted",
               cells=int(cells.shape[0]), t_setup_s=t_setup)
    ctx.close()
    if keep_arrays:
        out["_arrays"] = arrays   # (popped by the caller: not part of the record)
    return out


def run_c5(capi, meshgen, nx=87, steps=3, warmup=1, time_spmv=16, rtol=1e-10, device=0, hbm_peak_gbps=8000.0):
    nodes, cells, bnd = meshgen.unit_cube(nx)
    ctx = capi.Context(device)
    t0 = time.perf_counter()
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(2)
    ctx.solver_prepare(True)
    t_setup = time.perf_counter() - t0
    n_cells = int(cells.shape[0])
    del nodes, cells
    ctx.set_operator(c5_operator(capi))
    ctx.set_forcing(c5_forcing(ctx.quadrature_nodes()))
    ctx.set_dirichlet(np.zeros(nd))
    # the open method as a caller gets it: from 300 k DOFs on an order-2 system with constant coefficients goes to the two-level solver (eng_pmg.hip: the P1
    # space of the same mesh as the coarse level) -- its first call builds that level (untimed here, like every set-up; reported)
    t0 = time.perf_counter()
    ctx.init()
    first = ctx.solve(rtol=rtol)
    t_first = time.perf_counter() - t0
    two_level = None
    if first.method_used == capi.SOLVER_PMG:
        wall2, infos2 = _timed_steps(ctx, steps, warmup, 0, rtol)
        _, _, coords = ctx.dofs_get()
        two_level = {"dof_per_s": nd / wall2, "ms_per_step": 1e3 * wall2, "iterations": int(infos2[-1].iters), "iterations_per_step": [int(i.iters) for i in infos2],
                     "fine_operator_applications": 3 * int(infos2[-1].iters) + 3, "method": int(infos2[-1].method_used), "relres_true": float(infos2[-1].relres),
                     "t_assemble_ms": float(np.mean([i.t_assemble_ms for i in infos2])), "t_solve_ms": float(np.mean([i.t_solve_ms for i in infos2])),
                     "max_abs_error_vs_analytic": float(np.abs(ctx.solution() - c5_exact(coords)).max()), "first_call_s_with_coarse_level_setup": t_first,
                     "note": "flexible GMRES, right-preconditioned by a V(1,1) cycle: damped Jacobi, P A1^-1 P^T (A1: the same operator assembled on the P1 space of the "
                             "mesh, solved to 1e-1 by the single-launch BiCGStab of a context of its own), damped Jacobi; three fine operator applications (blocked-ELL "
                             "SpMV on A D^-1) and one coarse solve per iteration; iterations do not grow with the mesh (17-20 from 16 k to 5.4 M DOFs)"}
        u_two = ctx.solution()
        # the same step with the operator handed over again before every fdapde_init: a NEW matrix epoch -- the coarse operator is assembled again, the blocked-ELL
        # layout filled again, the damping estimated again (what a caller pays whose operator really changes from solve to solve)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.set_operator(c5_operator(capi))
            ctx.init()
            ctx.solve(rtol=rtol)
        ctx.synchronize()
        two_level["ms_per_step_with_the_operator_set_again"] = 1e3 * (time.perf_counter() - t0) / steps
        two_level["epoch_note"] = ("ms_per_step: fdapde_init + fdapde_solve with the operator unchanged -- the row-owner sweep reproduces the matrix bit for bit, so what the "
                                   "solver derived from it (coarse operator, blocked-ELL fill, Jacobi damping) is kept; ms_per_step_with_the_operator_set_again: all of that redone")
        ctx.tune("pmg_auto", 0)   # ... and the Jacobi-preconditioned stage it replaces, on the same context: what follows is that record
    wall, infos = _timed_steps(ctx, steps, warmup, time_spmv, rtol)
    out = _summary(ctx, nd, wall, infos, c5_exact, hbm_peak_gbps)
    if two_level is not None:
        two_level["max_abs_diff_vs_jacobi_bicgstab"] = float(np.abs(ctx.solution() - u_two).max())
        jac = {k: out[k] for k in ("dof_per_s", "ms_per_step", "t_solve_ms", "iterations", "us_per_iteration", "method")}
        out["jacobi_bicgstab"] = jac
        out["two_level"] = two_level
        out["dof_per_s"], out["ms_per_step"] = two_level["dof_per_s"], two_level["ms_per_step"]   # the headline of this entry: the open method
        out["headline_method"] = int(two_level["method"])
        out["record_note"] = ("dof_per_s / ms_per_step: the open method (two_level); every other top-level key (iterations, us_per_iteration, spmv_avg_us, frac, "
                              "traffic, layout ...) describes the Jacobi-BiCGStab stage it replaces at this size, as in earlier records (jacobi_bicgstab has its rates)")
    out.update(workload=f"C5: 3-D P2 advection-diffusion-reaction, {nx}^3 x 6 = {n_cells} tetrahedra, b = (1, 0.5, 0.25), c = 1; "
                        "3-D P2 numbering build-defined (parity unpinned)",
               cells=n_cells, t_setup_s=t_setup, iterations_per_step=[int(i.iters) for i in infos],
               # BiCGStab's path to 1e-10 is chaotic in the last bits of its scalars: the SAME problem with its right-hand side scaled by (1 + k 2^-48),
               # k = 0 .. 5, takes 659 - 785 iterations (tools/c5_iter_spread.py -> profiles/r6_c5_iter_spread.txt; other shadow residuals spread wider).
               # The figure of a record is one draw from that distribution; the steps of one run repeat it bit for bit.
               iteration_spread_note="659-785 iterations over six right-hand sides differing in the last bits (profiles/r6_c5_iter_spread.txt)")
    try:   # HBM bytes per SpMV launch from the committed counter passes of this workload (tools/profile_c5.sh), labelled: not this run
        import json
        import os

        pj = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r5_c5_spmv_pmc.json")))
        if nx == 87 and not out["persistent"] and pj.get("hbm_bytes_per_launch") and out.get("spmv_avg_us"):
            out["traffic"] = float(pj["hbm_bytes_per_launch"])
            out["traffic_frac"] = out["traffic"] / (out["spmv_avg_us"] * 1e-6) / 1e9 / hbm_peak_gbps
            out["traffic_source"] = ("profiles/r5_c5_spmv_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE of k_spmv_blocked, separate rocprofv3 --pmc passes of an "
                                     "earlier run of this workload (not this run)")
    except Exception:
        pass
    ctx.close()
    return out


def run_wide(capi, meshgen, nx=132, steps=2, warmup=1, rtol=1e-10, device=0, hbm_peak_gbps=8000.0):
    """The headline problem one size up -- (nx + 1)^3 = 2.35 M DOFs at nx = 132 -- where the single launch runs in its WIDE form (24 rows per
    thread, plain storage, x in HBM between the iterations: kernels_persist.h).  Its layout streams ~370 MB per iteration: more
    than the 256 MiB Infinity Cache holds, so the fraction quoted here is on an HBM-resident stream (C3's 146 MB per iteration is not)."""
    nodes, cells, bnd = meshgen.unit_cube(nx)
    u_exact, f = meshgen.manufactured(3)
    ctx = capi.Context(device)
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(1)
    n_cells = int(cells.shape[0])
    del nodes, cells
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    ctx.set_dirichlet(np.zeros(nd))
    ctx.solver_prepare(True)
    wall, infos = _timed_steps(ctx, steps, warmup, 0, rtol)
    out = _summary(ctx, nd, wall, infos, u_exact, hbm_peak_gbps)
    out.update(workload=f"3-D P1 Laplacian, {nx}^3 x 6 = {n_cells} tetrahedra, {nd} DOFs (C3's problem one size up): the wide form of the single launch",
               cells=n_cells, infinity_cache_bytes=256 * 1024 * 1024,
               residency="streamed bytes per iteration exceed the 256 MiB Infinity Cache: an HBM-resident stream")
    ctx.close()
    return out


def run_large(capi, meshgen, nx=200, steps=2, warmup=1, time_spmv=24, rtol=1e-10, device=0, hbm_peak_gbps=8000.0):
    """The regime ABOVE one launch: C3's problem at (nx + 1)^3 = 8.1 M DOFs (nx = 200: 48 M tetrahedra) -- more rows than the single launch holds (3.1 M on
    256 CUs), so the solve is the multi-launch Jacobi-PCG: k_spmv_blocked / k_spmv_team2 + the fused vector update per iteration, ~1.6 GB of CSR per
    operator application streaming from HBM.  The only regime where north_star's "HBM-bound CSR SpMV inside CG" is literally what runs; the reference's own
    solve has no size cap but memory (fem_linear_elliptic_solver.h:38-47)."""
    t0 = time.perf_counter()
    nodes, cells, bnd = meshgen.unit_cube(nx)
    t_gen = time.perf_counter() - t0
    u_exact, f = meshgen.manufactured(3)
    ctx = capi.Context(device)
    t0 = time.perf_counter()
    ctx.mesh_upload(nodes, cells, bnd)
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    nd = ctx.dofs_build(1)
    ctx.solver_prepare(True)
    t_setup = time.perf_counter() - t0
    n_cells = int(cells.shape[0])
    del nodes, cells
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    ctx.set_dirichlet(np.zeros(nd))
    wall, infos = _timed_steps(ctx, steps, warmup, time_spmv, rtol)
    out = _summary(ctx, nd, wall, infos, u_exact, hbm_peak_gbps)
    info = infos[-1]
    # the whole iteration against its minimal traffic: one pass over the layout + the vector passes of the fused update (x, r, p read and written, y read)
    it_us = 1e3 * float(info.t_solve_ms) / max(int(info.iters), 1)
    vec_bytes = 7.0 * 8.0 * out["interior_rows"]
    out.update(workload=f"3-D P1 Laplacian, {nx}^3 x 6 = {n_cells} tetrahedra, {nd} DOFs: above the single launch (multi-launch Jacobi-PCG)", cells=n_cells,
               t_meshgen_s=t_gen, t_mesh_upload_s=t_up, t_setup_s=t_setup,
               iteration_gbps=(out["streamed_bytes_per_application"] + vec_bytes) / (it_us * 1e-6) / 1e9,
               iteration_frac=(out["streamed_bytes_per_application"] + vec_bytes) / (it_us * 1e-6) / 1e9 / hbm_peak_gbps,
               iteration_bytes_note="layout bytes of one operator application + 7 vector passes of 8 n bytes (fused update: x, r, p read + written, y read)")
    ctx.close()
    return out


def run_group(capi, meshgen, nx=119, n_dev=2, steps=2, warmup=1, rtol=1e-10):
    """The multi-device context (fdapde_ctx_create_multi) on C3's mesh with what this box has: n_dev "devices" dealt round-robin over the real ones (ONE GPU:
    all of them GPU 0, its CUs shared out -- a plumbing and set-up-cost record, no scaling figure).  What it shows: the split behind the one-object
    interface costs tens of milliseconds on the device (dist.py's numpy partitioner: 16.3 s), and the sharded solve is the same Krylov iteration."""
    n_real = max(int(capi.load().fdapde_device_count()), 1)
    devices = [r % n_real for r in range(n_dev)]
    nodes, cells, bnd = meshgen.unit_cube(nx)
    u_exact, f = meshgen.manufactured(3)
    ctx = capi.Context(devices=devices)
    ctx.mesh_upload(nodes, cells, bnd)
    t0 = time.perf_counter()
    nd = ctx.dofs_build(1)
    t_build = time.perf_counter() - t0
    n_cells = int(cells.shape[0])
    del nodes, cells
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    ctx.set_dirichlet(np.zeros(nd))
    wall, infos = _timed_steps(ctx, steps, warmup, 0, rtol)
    info = infos[-1]
    _, _, coords = ctx.dofs_get()
    err = float(np.abs(ctx.solution() - u_exact(coords)).max())
    d = ctx.devices()
    ctx.close()
    return {"workload": f"C3's mesh ({nx}^3 x 6 = {n_cells} tetrahedra, {nd} DOFs) through ONE multi-device context", "devices": devices, "real_devices": n_real,
            "form": {0: "row-distributed", 1: "element partition"}.get(d["form"], str(d["form"])), "dofs_build_ms": 1e3 * t_build,
            "t_partition_ms": d["t_partition_ms"], "t_rank_setup_ms": d["t_rank_setup_ms"], "ms_per_step": 1e3 * wall, "dof_per_s": nd / wall,
            "iterations": int(info.iters), "us_per_iteration": 1e3 * float(info.t_solve_ms) / max(int(info.iters), 1), "persistent": int(info.persistent),
            "relres": float(info.relres), "max_abs_error_vs_analytic": err,
            "note": ("the devices are ONE GPU named several times: set-up cost and plumbing only, no scaling figure" if n_real < n_dev else "one rank per GPU")}


def _read_fixture_csv(path, dtype):
    """the reference's CSV dialect (utils/IO/csv_reader.h:75-117): a header row, first column = row index, quotes stripped"""
    rows = []
    with open(path) as fh:
        next(fh)
        for line in fh:
            parts = [t.strip().strip('"') for t in line.strip().split(",")]
            if len(parts) > 1:
                rows.append(parts[1:])
    return np.array(rows, dtype=float).astype(dtype)


def load_fixture_mesh(directory):
    """points / elements / boundary of one of the reference's test meshes (test/src/utils/mesh_loader.h:62-84: 1-based -> 0-based)"""
    import os

    nodes = _read_fixture_csv(os.path.join(directory, "points.csv"), float)
    cells = _read_fixture_csv(os.path.join(directory, "elements.csv"), np.int32) - 1
    bnd = _read_fixture_csv(os.path.join(directory, "boundary.csv"), np.uint8).reshape(-1)
    return np.ascontiguousarray(nodes), np.ascontiguousarray(cells.astype(np.int32)), np.ascontiguousarray(bnd)


def run_c1(capi, golden_mesh_dir, names=("unit_square_16", "unit_square_32"), reps=50, rtol=1e-10, device=0):
    """BASELINE config C1: the reference's own unit_square_16 / _32 fixtures (289 / 1 089 DOFs): wall time per PDE::init(), per PDE::solve()
    and per column of the factor-once handle (one by one, and 64 side by side) -- the sizes the reference is used at, where fixed per-call
    costs matter and bandwidth does not"""
    import os

    out = {}
    for name in names:
        nodes, cells, bnd = load_fixture_mesh(os.path.join(golden_mesh_dir, name))
        ctx = capi.Context(device)
        ctx.mesh_upload(nodes, cells, bnd)
        nd = ctx.dofs_build(1)
        qn = ctx.quadrature_nodes()
        ctx.set_operator(-capi.laplacian())
        ctx.set_forcing(2 * np.pi**2 * np.sin(np.pi * qn[:, 0]) * np.sin(np.pi * qn[:, 1]))
        ctx.set_dirichlet(np.zeros(nd))
        for _ in range(3):
            ctx.init()
            info = ctx.solve(rtol=rtol)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.init()
        ctx.synchronize()
        t_init = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            info = ctx.solve(rtol=rtol)
        ctx.synchronize()
        t_solve = (time.perf_counter() - t0) / reps
        _, _, coords = ctx.dofs_get()
        err = float(np.abs(ctx.solution() - np.sin(np.pi * coords[:, 0]) * np.sin(np.pi * coords[:, 1])).max())
        # the handle downstream models hold (fdapde::SparseLU: compute once, solve many): -Lap + mass, no Dirichlet rows
        ctx.set_operator(-capi.laplacian() + capi.reaction(1.0))
        ctx.init()
        ctx.lin_compute(capi.MAT_STIFF)
        rng = np.random.default_rng(0)
        B = rng.standard_normal((nd, 64))
        # the handle as its users drive it: column after column against ONE matrix.  The first columns are Krylov runs; once they have cost half an
        # inversion the handle inverts (rent or buy) and a column is one product.  Reported: the Krylov column, how many of them it took, what
        # the inversion cost, and the steady state after it.
        t0 = time.perf_counter()
        _, hinfo = ctx.lin_solve(B[:, 0], rtol=rtol)
        _, hinfo = ctx.lin_solve(B[:, 1], rtol=rtol)
        t_krylov_col = (time.perf_counter() - t0) / 2
        n_before, t_build = 2, 0.0
        while hinfo.method_used != 6 and n_before < 400:
            t0 = time.perf_counter()
            _, hinfo = ctx.lin_solve(B[:, n_before % 64], rtol=rtol)
            t_build = time.perf_counter() - t0   # (the call that switched carries the inversion)
            n_before += 1
        ctx.lin_solve(B, rtol=rtol)
        t0 = time.perf_counter()
        for k in range(reps):
            ctx.lin_solve(B[:, k % 64], rtol=rtol)
        t_col = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.lin_solve(B, rtol=rtol)
        t_cols = (time.perf_counter() - t0) / 5 / 64
        # the parabolic loop at the reference's own size (fem_pde_test.cpp:222-368's shape): 101 time points, K = M / dt + A fixed over the steps
        times = np.linspace(0.0, 1.0, 101)
        ctx.set_operator(capi.dt() - capi.laplacian())
        ctx.set_forcing(np.stack([np.sin(np.pi * qn[:, 0]) * np.cos(t) for t in times], axis=1))
        ctx.init()
        u0 = np.sin(np.pi * coords[:, 0]) * np.sin(np.pi * coords[:, 1])
        G = np.zeros((nd, times.size))
        ctx.solve_parabolic(times, u0, G, rtol=rtol)
        t0 = time.perf_counter()
        _, pinfo = ctx.solve_parabolic(times, u0, G, rtol=rtol)
        t_par = time.perf_counter() - t0
        out[name] = {"dofs": int(nd), "cells": int(cells.shape[0]), "init_ms": 1e3 * t_init, "solve_ms": 1e3 * t_solve, "iterations": int(info.iters),
                     "persistent": int(info.persistent), "max_abs_error_vs_analytic": err,
                     "handle_solve_one_column_ms": 1e3 * t_col, "handle_solve_per_column_of_64_ms": 1e3 * t_cols,
                     "handle_method": int(hinfo.method_used),   # 6 = the dense inverse (kernels_dense.h), taken by a handle that has solved many columns
                     "handle_krylov_column_ms": 1e3 * t_krylov_col, "handle_columns_before_the_inverse": int(n_before) - 1,
                     "handle_inversion_ms": 1e3 * t_build if hinfo.method_used == 6 else None,
                     "parabolic_101_points_ms": 1e3 * t_par, "parabolic_method": int(pinfo.method_used)}
        ctx.close()
    return out
