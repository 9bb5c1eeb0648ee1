#!/usr/bin/env python3
"""bench.py -- assemble + solve of the 3-D P1 Laplacian (BASELINE.json: metric, configs[2] "C3").

A step = one pass of the hot path over the resident mesh: FEMSolverBase::init (stiffness + forcing + mass assembly)
followed by PDE::solve (Dirichlet reduction + Jacobi-PCG to rtol 1e-10), all on the device.  Inputs (mesh, forcing
samples, Dirichlet data) are resident in HBM when the timed region starts; nothing but the solver's convergence flag
crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nx 119]

N = 1: the whole C3 mesh (119^3 x 6 = 10 110 954 tetrahedra, 1 728 000 DOFs) on one MI355X.
N > 1 (launched by torch.distributed.run, one rank per GPU): the same mesh, element-partitioned over the ranks with
an RCCL all-reduce of the interface DOF contributions per operator application (strong scaling).

Prints ONE JSON line (rank 0).  `roofline` is the CSR SpMV inside CG: algorithmic bytes 12 nnz + 4 (n+1) + 16 n per
launch over the average launch duration measured with HIP events on the solver's stream during the timed steps.
`cpu_baseline` is the CPU oracle (oracle/fem_oracle.c, the single-threaded port of the reference algorithm) timed on the
same workload at the same size (--cpu-nx 119); `cpu_baseline_all_cores` is the same restatement with OpenMP on every host core
(oracle/fem_oracle_mt.c: BASELINE.md's "CPU-best" column).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
RTOL = 1e-10


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=119, help="cubes per axis of the C3 mesh (119 = BASELINE size)")
    ap.add_argument("--cpu-nx", type=int, default=119,
                    help="cubes per axis of the CPU-baseline sample (119 = the GPU line's own workload: ~25 s on 1 core + ~6 s on all cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary single-GPU results (BASELINE configs C2 and C5) appended as `extra`")
    ap.add_argument("--time-spmv", type=int, default=32,
                    help="SpMV launches per step timed with dispatch-attached HIP events (each costs a ~6 us bubble)")
    return ap.parse_args()


def cpu_baseline(nx):
    """The oracle (kind 'port') on a bounded sample: same generator, same operator, same solver and tolerance."""
    from fdapde_loader import load_package
    from oracle import oracle as o

    load_package()
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_cube(nx)
    m = o.Mesh(nodes, cells, bnd)
    _, f = meshgen.manufactured(3)
    dofs, b, nd, _ = o.enumerate_dofs(m, 1)
    fq = f(o.quadrature_nodes(m, 1))
    t0 = time.perf_counter()
    A = o.assemble_operator(m, 1, dofs, nd, -o.laplacian())
    rhs = o.assemble_forcing(m, 1, dofs, nd, fq)
    Mm = o.assemble_operator(m, 1, dofs, nd, o.reaction(1.0))
    t1 = time.perf_counter()
    u, it, rr, rc = o.pcg(A, rhs, b, np.zeros(nd), rtol=RTOL, maxit=100000)
    t2 = time.perf_counter()
    assert rc == 0
    faithful = {
        "value": nd / (t2 - t0), "unit": "DOF/s", "cores": 1, "kind": "port",
        "sample": f"3-D P1 Laplacian, {nx}^3 x 6 = {m.n_cells} tetrahedra, {nd} DOFs, same generator/operator/rtol; "
                  f"assemble {t1 - t0:.2f} s + Jacobi-PCG {t2 - t1:.2f} s ({it} iterations); oracle/fem_oracle.c at -O2",
    }
    # BASELINE.md "CPU-best": the same restatement on all host cores (coloured OpenMP assembly into the prebuilt pattern +
    # row-parallel PCG, oracle/fem_oracle_mt.c at -O3 -march=native); pattern and colouring are set-up, as on the GPU
    best = None
    try:
        o.mt_set_threads(o.usable_cpus())   # affinity mask capped by the cgroup CPU quota
        colouring = o.mt_colour_cells(dofs, nd)
        o.mt_assemble(m, 1, dofs, nd, -o.laplacian(), A, colouring, fq)   # warm-up: thread pool, page faults
        t3 = time.perf_counter()
        A2, rhs2 = o.mt_assemble(m, 1, dofs, nd, -o.laplacian(), A, colouring, fq)
        o.mt_assemble(m, 1, dofs, nd, o.reaction(1.0), Mm, colouring)
        t4 = time.perf_counter()
        u2, it2, rr2, rc2 = o.mt_pcg(A2, rhs2, b, np.zeros(nd), rtol=RTOL, maxit=100000)
        t5 = time.perf_counter()
        assert rc2 == 0 and np.abs(u2 - u).max() <= 1e-8 * max(1.0, np.abs(u).max())
        best = {
            "value": nd / (t5 - t3), "unit": "DOF/s", "cores": o.mt_threads(), "kind": "port, OpenMP",
            "sample": f"same sample; coloured assembly {t4 - t3:.3f} s + row-parallel Jacobi-PCG {t5 - t4:.3f} s ({it2} iterations) on "
                      f"{o.mt_threads()} threads (= the CPUs this container may use: {os.cpu_count()} visible, cgroup quota applied); "
                      "oracle/fem_oracle_mt.c at -O3 -march=native; reference DOF numbering (ids as generated)",
        }
    except Exception as e:   # the all-cores column is an extra: never let it take the bench line down
        best = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    del Mm
    return faithful, best


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    # FDAPDE_BENCH_BACKEND=gloo: plumbing check of the N > 1 leg on a box with fewer GPUs than ranks (ranks share devices,
    # host-staged all-reduce instead of RCCL).  Never used for reported numbers.
    backend = os.environ.get("FDAPDE_BENCH_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    device_index = local_rank % n_dev
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(device_index)
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    from fdapde_loader import load_package

    pkg = load_package()
    from fdapde_core_amd import capi, meshgen

    if capi.load().fdapde_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    t_gen = time.perf_counter()
    nodes, cells, bnd = meshgen.unit_cube(args.nx)
    t_gen = time.perf_counter() - t_gen
    n_cells_total = int(cells.shape[0])
    u_exact, f = meshgen.manufactured(3)

    if world == 1:
        ctx = capi.Context(device=device_index)
        ctx.mesh_upload(nodes, cells, bnd)
        n_dofs = ctx.dofs_build(1)
        sizes = ctx.sizes()
        qn = ctx.quadrature_nodes()
        ctx.set_operator(-capi.laplacian())
        ctx.set_forcing(f(qn))
        ctx.set_dirichlet(np.zeros(n_dofs))
        del qn
        t0 = time.perf_counter()
        ctx.solver_prepare(True)   # set-up: compact solver pattern + 16-bit column codes for this boundary mask (host work + upload)
        t_prep = time.perf_counter() - t0

        def step(time_spmv=0):
            ctx.init()
            return ctx.solve(rtol=RTOL, time_spmv=time_spmv)

        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        infos = [step(args.time_spmv) for _ in range(args.steps)]
        barrier()
        elapsed = time.perf_counter() - t0
        info = infos[-1]
        spmv_ms = float(np.mean([i.spmv_avg_ms for i in infos]))
        t_asm = float(np.mean([i.t_assemble_ms for i in infos]))
        t_sol = float(np.mean([i.t_solve_ms for i in infos]))
        u = ctx.solution()
        _, _, coords = ctx.dofs_get()
        err = float(np.abs(u - u_exact(coords)).max())
        setup_ms = ctx.info().t_setup_ms + 1e3 * t_prep
        _, alg_bytes = ctx.bench_spmv(reps=1)
        n_int, nnz_int, streamed_bytes = ctx.solver_layout(True)
        parallelism = "1 GPU"
        total_dofs = n_dofs
        ctx.close()
        del nodes, cells, bnd
    else:
        from fdapde_core_amd import dist as fdist

        res = fdist.bench_partitioned(capi, nodes, cells, bnd, f, u_exact, rank, world, device_index, args, barrier, RTOL, backend)
        if rank != 0:
            return
        (elapsed, info, spmv_ms, t_asm, t_sol, err, setup_ms, alg_bytes, sizes, total_dofs, parallelism) = res
        n_int = nnz_int = streamed_bytes = None

    if rank != 0:
        return
    # HBM bytes per SpMV launch: NOT measured in this run (PMC counters need rocprofv3 around the process) -- taken from the
    # committed PMC passes of the same workload and code (tools/profile_gpu.sh -> profiles/spmv_pmc.json), and labelled so
    traffic, traffic_source = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "spmv_pmc.json")))
        if world == 1 and pj.get("nx") == args.nx:
            if getattr(info, "persistent", 0) and pj.get("persist_hbm_bytes_per_solve"):
                # one dispatch per solve: counter bytes of the whole launch / iterations (the operator application is all it streams)
                traffic = pj["persist_hbm_bytes_per_solve"] / max(int(pj.get("persist_iterations", info.iters)), 1)
            elif not getattr(info, "persistent", 0):
                traffic = pj.get("hbm_bytes_per_launch")
            traffic_source = "profiles/spmv_pmc.json (rocprofv3 --pmc passes of an earlier run of this workload; not this run)"
    except Exception:
        pass
    ms_per_step = 1e3 * elapsed / args.steps
    achieved = alg_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    out = {
        "metric": "DOF/s assemble+solve, 3D P1 Laplacian; SpMV achieved HBM GB/s vs peak",
        "value": total_dofs * args.steps / elapsed,
        "unit": "DOF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"C3: 3-D P1 Laplacian, [0,1]^3, {args.nx}^3 x 6 Kuhn tetrahedra = {n_cells_total} cells, "
                        f"{total_dofs} DOFs, nnz {sizes['nnz']}, jitter 0.2h, ids permuted, seed 12345; "
                        "u = sin(pi x) sin(pi y) sin(pi z), homogeneous Dirichlet; init (stiff+force+mass) + Jacobi-PCG rtol 1e-10",
            "parallelism": parallelism,
            "cg_iterations": int(info.iters),
            "relres": float(info.relres),
            "t_assemble_ms": t_asm,
            "t_solve_ms": t_sol,
            "t_setup_ms_untimed": setup_ms,
            "t_meshgen_s_untimed": t_gen,
            "max_abs_error_vs_analytic": err,
            "spmv_launches_timed_per_step": int(info.spmv_timed),
            "persistent_launch": int(getattr(info, "persistent", 0)),
            "us_per_iteration": 1e3 * t_sol / max(int(info.iters), 1),
            "operator_phase_mean_us": 1e3 * float(getattr(info, "spmv_mean_ms", 0.0)),
            "allgather_phase_us": 1e3 * float(getattr(info, "gather_avg_ms", 0.0)),
            "update_phase_us": 1e3 * float(getattr(info, "update_avg_ms", 0.0)),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("k_cg_persist operator phase (the whole CG is ONE launch: x, r, p in registers, the matrix streams once per iteration; "
                       "avg_launch_ms = per-iteration SpMV + neighbour-import phase of the slowest workgroup, stamped in the kernel)"
                       if getattr(info, "persistent", 0) else "k_spmv_team2 (CSR SpMV fused with p.Ap and Ap.Ap inside CG)"),
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": spmv_ms,
        },
    }
    if n_int is not None and spmv_ms > 0:
        # the CG streams the Dirichlet-reduced interior block, not the full CSR the caller sees: the same figure on that operator,
        # and the rate of the bytes the kernel really streams (compact layout: no diagonal, 16-bit column codes)
        alg_int = 12.0 * nnz_int + 4.0 * (n_int + 1) + 16.0 * n_int
        r = out["roofline"]
        r["algorithmic_bytes_interior"] = alg_int
        r["achieved_interior"] = alg_int / (spmv_ms * 1e-3) / 1e9
        r["frac_interior"] = r["achieved_interior"] / HBM_PEAK_GBPS
        r["streamed_bytes_per_launch"] = streamed_bytes
        r["streamed_gbps"] = streamed_bytes / (spmv_ms * 1e-3) / 1e9
        r["interior_rows"], r["interior_nnz"] = n_int, nnz_int
    if world == 1 and not args.no_extra:
        # secondary results, after the C3 line's own timed region: the other single-GPU BASELINE configurations
        from fdapde_core_amd import workloads

        extra = {}
        for name, fn in (("c2", workloads.run_c2), ("c5", workloads.run_c5)):
            try:
                extra[name] = fn(capi, meshgen, device=device_index, hbm_peak_gbps=HBM_PEAK_GBPS)
            except Exception as e:   # never let a secondary result take the bench line down
                extra[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["extra"] = extra
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["cpu_baseline_all_cores"] = cpu_baseline(args.cpu_nx)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
