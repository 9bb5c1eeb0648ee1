#!/usr/bin/env python3
"""bench.py -- assemble + solve of the 3-D P1 Laplacian (BASELINE.json: metric, configs[2] "C3").

A step = one pass of the hot path over the resident mesh: FEMSolverBase::init (stiffness + forcing + mass assembly)
followed by PDE::solve (Dirichlet reduction + Jacobi-PCG to rtol 1e-10), all on the device.  Inputs (mesh, forcing
samples, Dirichlet data) are resident in HBM when the timed region starts; nothing but the solver's convergence flag
crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nx 119]

N = 1: the whole C3 mesh (119^3 x 6 = 10 110 954 tetrahedra, 1 728 000 DOFs) on one MI355X.
N > 1: the same mesh, element-partitioned over N ranks, one per GPU (strong scaling).  Started either by
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) or plainly as `python bench.py --gpus N ...`: with WORLD_SIZE unset the script starts its own N rank processes --
fresh children, decided before anything touches the GPU -- and rank 0's JSON line is the output.

A rank process holds ONE HIP / RCCL stack: it loads libfdapde_hip.so (bound to /opt/rocm's runtime) and nothing else that
touches the GPU -- no torch.  The 128-byte RCCL id of the library's communicator travels through a rendezvous directory on the
node's file system, the barriers and the max-over-ranks of the timing are all-reduces of that communicator
(fdapde_comm_allreduce).  FDAPDE_BENCH_BACKEND=gloo (plumbing checks on a box with fewer GPUs than ranks: ranks share
devices, host-staged transports over torch.distributed/gloo, imported AFTER the library) is never used for reported numbers.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel -- on one GPU the single launch that runs the whole CG
(k_cg_persist): bytes it streams per launch (its layout x iterations) over the launch duration measured with HIP events on the
solver's stream; `effective_*` are the same time on the algorithmic bytes of the CSR operator the caller sees
(12 nnz + 4 (n+1) + 16 n per application).  `cpu_baseline` is the CPU oracle (oracle/fem_oracle.c, the single-threaded port of the
reference algorithm) timed on the same workload at the same size; `cpu_baseline_all_cores` the same restatement with OpenMP.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
RTOL = 1e-10


# seed of the generator per mesh size: 12345 wherever that jitter leaves every tetrahedron upright; at nx = 189 it inverts one (the generator asserts):
# the next seed that does not
MESH_SEEDS = {189: 12346}


def weak_nx(n_gpus):
    """cubes per axis of the weak-scaling mesh: (nx + 1)^3 nodes ~ n_gpus x C3's 120^3"""
    return int(round((n_gpus * 120.0**3) ** (1.0 / 3.0))) - 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=None, help="cubes per axis of the mesh (default: 119 = C3's BASELINE size; 87 for --workload c5)")
    ap.add_argument("--workload", choices=("c3", "c5"), default="c3",
                    help="c3 (default): the headline configuration.  c5 (N > 1 only): BASELINE config C5 -- 3-D P2 advection-diffusion-reaction, "
                         "Jacobi-BiCGStab -- across the ranks in the row-distributed form")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong (default) = the C3 mesh split over the N GPUs; weak = the mesh grown with N so that every GPU keeps about C3's "
                         "1.73 M rows (nx = round((N 120^3)^(1/3)) - 1: 150 / 189 / 239 cubes per axis at N = 2 / 4 / 8)")
    ap.add_argument("--cpu-nx", type=int, default=119,
                    help="cubes per axis of the CPU-baseline sample (119 = the GPU line's own workload: ~25 s on 1 core + ~6 s on all cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary single-GPU results (BASELINE configs C2 and C5) appended as `extra`")
    ap.add_argument("--time-spmv", type=int, default=32,
                    help="multi-launch path: SpMV launches per step timed with dispatch-attached HIP events (each costs a ~6 us bubble)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the script starts its own rank processes (nothing has touched the GPU in this process)
# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    return _spawn_ranks(args.gpus, sys.argv[1:], {}, timeout=0)


class FileRendezvous:
    """Small blobs between the rank processes of ONE node through a directory: put = write + atomic rename, get = poll."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        d = os.environ.get("FDAPDE_BENCH_RDZV")   # (set by _spawn_ranks: a fresh mkdtemp directory, removed by the spawner)
        self.own = not d
        if not d:
            # under torch.distributed.run all ranks are children of ONE agent process: the directory is named after that process INSTANCE --
            # pid + its start time in clock ticks since boot (field 22 of /proc/<pid>/stat): no earlier run can have had the same pair, so no
            # stale 'form', 'rccl_id' or 'problem.N' of a previous job is ever read (same port, default run id and a recycled pid included)
            ppid = os.getppid()
            try:
                with open(f"/proc/{ppid}/stat") as f:
                    born = f.read().rsplit(")", 1)[1].split()[19]
            except Exception:
                born = "0"
            d = os.path.join("/tmp", f"fdapde_rdzv_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{ppid}_{born}")
        self.dir = d
        os.makedirs(d, exist_ok=True)

    def consume(self, key):
        """get + delete: blobs addressed to ONE rank (its sub-mesh: ~25 MB per rank at C3 size) do not outlive their reader"""
        data = self.get(key)
        try:
            os.remove(self.path(key))
        except OSError:
            pass
        return data

    def finish(self, timeout=120.0):
        """every rank says goodbye; rank 0 waits for all of them and removes the directory it named itself (a directory handed in by
        _spawn_ranks is removed by the spawner)"""
        import shutil

        try:
            self.put(f"bye.{self.rank}", b"1")
            if self.rank == 0 and self.own:
                for r in range(self.world):
                    try:
                        self.get(f"bye.{r}", timeout=timeout)
                    except TimeoutError:
                        break   # (a rank that died: clean up anyway)
                shutil.rmtree(self.dir, ignore_errors=True)
        except OSError:
            pass

    def path(self, key):
        return os.path.join(self.dir, key)

    def put(self, key, data: bytes):
        tmp = self.path(f".{key}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, self.path(key))

    def get(self, key, timeout=1800.0) -> bytes:
        t0 = time.time()
        p = self.path(key)
        while not os.path.exists(p):
            if time.time() - t0 > timeout:
                raise TimeoutError(f"rank {self.rank}: rendezvous key {key} did not appear in {self.dir}")
            time.sleep(0.01)
        with open(p, "rb") as f:
            return f.read()

    def barrier(self, name):
        self.put(f"{name}.{self.rank}", b"1")
        for r in range(self.world):
            self.get(f"{name}.{r}")


# ---------------------------------------------------------------------------------------------------------------------
# Oracle parity AT THE HEADLINE SIZES, in the record.  The CPU legs below already run the oracle -- the reference's
# algorithm restated (oracle/fem_oracle.c) -- on the very workloads the GPU lines are quoted on; what the device path produced for the same
# inputs is compared with it entry by entry here.  Bars: SURVEY 8(d).  Test infrastructure only: runs after and outside every timed region.
# ---------------------------------------------------------------------------------------------------------------------
PARITY_BARS = {"pattern": "bit-exact", "dof_table": "bit-exact", "boundary_dofs": "bit-exact",
               "matrix_entries": 1e-12, "force": 1e-12, "solution_rel_l2": 1e-8}


def parity_against_oracle(gpu, dofs, bnd, A, Mm, rhs, u, what_u):
    """gpu: arrays of the device path (pattern, dof table, boundary DOFs, stiff_ / mass_ / force_ right after init, solution()); the rest: the
    oracle's for the same mesh.  -> dict with every figure, its bar, and `ok`"""
    import numpy as np

    def rel_max(a, b):
        return float(np.abs(a - b).max() / max(1.0, float(np.abs(b).max())))

    r = {"bars": PARITY_BARS}
    r["dof_table_equal"] = bool(np.array_equal(gpu["dofs"], dofs))
    r["boundary_dofs_equal"] = bool(np.array_equal(gpu["boundary"] != 0, np.asarray(bnd) != 0))
    r["pattern_equal"] = bool(np.array_equal(gpu["rowptr"], A.rowptr) and np.array_equal(gpu["colidx"], A.colidx))
    same = r["pattern_equal"]
    r["stiff_max_abs_diff_over_max1_Amax"] = rel_max(gpu["stiff"], A.values) if same else None
    r["mass_max_abs_diff_over_max1_Mmax"] = rel_max(gpu["mass"], Mm.values) if same and Mm is not None else None
    r["force_max_abs_diff_over_max1_fmax"] = rel_max(gpu["force"], rhs)
    r["solution_rel_l2"] = float(np.linalg.norm(gpu["u"] - u) / np.linalg.norm(u))
    r["solution_reference"] = what_u
    r["entries_compared"] = int(A.values.size) * (2 if Mm is not None else 1) + int(rhs.size)
    r["ok"] = bool(r["dof_table_equal"] and r["boundary_dofs_equal"] and same and
                   r["stiff_max_abs_diff_over_max1_Amax"] <= PARITY_BARS["matrix_entries"] and
                   (Mm is None or r["mass_max_abs_diff_over_max1_Mmax"] <= PARITY_BARS["matrix_entries"]) and
                   r["force_max_abs_diff_over_max1_fmax"] <= PARITY_BARS["force"] and r["solution_rel_l2"] <= PARITY_BARS["solution_rel_l2"])
    return r


def cpu_baseline(nx, gpu=None):
    """The oracle (kind 'port') on a bounded sample: same generator, same operator, same solver and tolerance."""
    import numpy as np

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
    parity = None
    if gpu is not None:   # entry-level parity of the headline configuration (same generator, same size: the GPU line's own workload)
        try:
            parity = parity_against_oracle(gpu, dofs, b, A, Mm, rhs, u, f"oracle Jacobi-PCG, rtol {RTOL:g} ({it} iterations)")
            parity["workload"] = f"C3 at {nx}^3 x 6 = {m.n_cells} tetrahedra, {nd} DOFs: fem_assembler.h:52-121 restated, entry by entry"
        except Exception as e:
            parity = {"ok": False, "error": f"{type(e).__name__}: {e}"[:300]}
    del Mm
    return faithful, best, parity


def cpu_columns(c5_nx=24, c2_gpu=None):
    """CPU figures beside the secondary results (SURVEY 8d plan item 1): the oracle's assembly (the reference's algorithm, 1 core) followed by a DIRECT
    solve -- scipy's SuperLU standing in for Eigen::SparseLU -- for C1 and C2, where the reference's own solve is feasible; for C5 the oracle's
    assembly + Jacobi-BiCGStab at a stated reduced size (LU of a 3-D P2 system: DNF by memory at full size)."""
    import numpy as np
    import scipy.sparse.linalg as spla

    from fdapde_loader import load_package
    from oracle import oracle as o

    load_package()
    from fdapde_core_amd import meshgen, workloads

    out = {}

    def direct(m, order, fq, g, keep_raw=False):
        dofs, bnd, nd, _ = o.enumerate_dofs(m, order)
        t0 = time.perf_counter()
        A = o.assemble_operator(m, order, dofs, nd, -o.laplacian())
        b = o.assemble_forcing(m, order, dofs, nd, fq)
        Mm = o.assemble_operator(m, order, dofs, nd, o.reaction(1.0))
        t1 = time.perf_counter()
        raw = (dofs, bnd, A.copy(), Mm, b.copy()) if keep_raw else None   # (untimed copies: set_dirichlet works in place)
        t1b = time.perf_counter()
        o.set_dirichlet(A, b, bnd, g(nd))
        lu = spla.splu(A.to_scipy().tocsc())
        t2 = time.perf_counter()
        u = lu.solve(b)
        t3 = time.perf_counter()
        return nd, t1 - t0, t2 - t1b, t3 - t2, u, raw

    try:   # C1: the reference's own fixtures
        c1 = {}
        for name in ("unit_square_16", "unit_square_32"):
            m = o.load_mesh(os.path.join(ROOT, "tests", "golden", "mesh", name))
            qn = o.quadrature_nodes(m, 1)
            fq = 2 * np.pi**2 * np.sin(np.pi * qn[:, 0]) * np.sin(np.pi * qn[:, 1])
            best = None
            for _ in range(5):
                r = direct(m, 1, fq, lambda nd: np.zeros(nd))
                best = r if best is None or sum(r[1:4]) < sum(best[1:4]) else best
            nd, ta, tf, ts, _, _ = best
            # the parabolic loop the way the reference runs it (fem_linear_parabolic_solver.h:41,56-68): one factorisation, a back-substitution per step
            t_par = None
            try:
                dofs_, bnd_, nd_, _ = o.enumerate_dofs(m, 1)
                A_ = o.assemble_operator(m, 1, dofs_, nd_, -o.laplacian()).to_scipy()
                M_ = o.assemble_operator(m, 1, dofs_, nd_, o.reaction(1.0)).to_scipy()
                rhs_ = np.ones(nd_)
                best_par = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    K_ = (M_ * 100.0 + A_).tocsc()
                    lu_ = spla.splu(K_)
                    u_ = np.zeros(nd_)
                    for _s in range(100):
                        u_ = lu_.solve(M_ @ u_ * 100.0 + rhs_)
                    t1 = time.perf_counter() - t0
                    best_par = t1 if best_par is None or t1 < best_par else best_par
                t_par = 1e3 * best_par
            except Exception:
                pass
            c1[name] = {"dofs": int(nd), "init_ms": 1e3 * ta, "solve_ms": 1e3 * (tf + ts), "factorise_ms": 1e3 * tf, "solve_one_column_ms": 1e3 * ts,
                        "parabolic_101_points_ms": t_par,
                        "cores": 1, "kind": "port + scipy SuperLU", "note": "oracle assembly (stiff + force + mass) | Dirichlet rows + splu + one solve; best of 5; "
                        "parabolic: K = M / dt + A factorised once, 100 steps of (M u / dt + f, back-substitution), best of 3"}
        out["c1"] = c1
    except Exception as e:
        out["c1"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    try:   # C2 at full size
        nodes, cells, bnd = meshgen.unit_square(708)
        m = o.Mesh(nodes, cells, bnd)
        _, f = meshgen.manufactured(2)
        nd, ta, tf, ts, u_lu, raw = direct(m, 1, f(o.quadrature_nodes(m, 1)), lambda nd: np.zeros(nd), keep_raw=c2_gpu is not None)
        if c2_gpu is not None:   # C2 at full size against the reference's actual algorithm: oracle assembly + a sparse LU (fem_linear_elliptic_solver.h:38-47)
            try:
                out["c2_parity"] = parity_against_oracle(c2_gpu, raw[0], raw[1], raw[2], raw[3], raw[4], u_lu, "oracle assembly + scipy SuperLU (direct)")
                out["c2_parity"]["workload"] = f"C2 at full size: {m.n_cells} triangles, {nd} DOFs"
            except Exception as e:
                out["c2_parity"] = {"ok": False, "error": f"{type(e).__name__}: {e}"[:300]}
            del raw
        out["c2"] = {"value": nd / (ta + tf + ts), "unit": "DOF/s", "cores": 1, "kind": "port + scipy SuperLU",
                     "sample": f"C2 at full size ({m.n_cells} triangles, {nd} DOFs): oracle assembly {ta:.2f} s + sparse LU {tf:.2f} s + solve {ts:.3f} s"}
    except Exception as e:
        out["c2"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    try:   # C5 at a reduced size
        nodes, cells, bnd = meshgen.unit_cube(c5_nx)
        m = o.Mesh(nodes, cells, bnd)
        dofs, b_dofs, nd, _ = o.enumerate_dofs(m, 2)
        fq = workloads.c5_forcing(o.quadrature_nodes(m, 2))
        op = -o.laplacian() + o.advection(workloads.C5_B) + o.reaction(workloads.C5_C)
        t0 = time.perf_counter()
        A = o.assemble_operator(m, 2, dofs, nd, op)
        rhs = o.assemble_forcing(m, 2, dofs, nd, fq)
        o.assemble_operator(m, 2, dofs, nd, o.reaction(1.0))
        t1 = time.perf_counter()
        u, it, rr, rc = o.bicgstab(A, rhs, b_dofs, np.zeros(nd), rtol=RTOL, maxit=100000)
        t2 = time.perf_counter()
        out["c5"] = {"value": nd / (t2 - t0), "unit": "DOF/s", "cores": 1, "kind": "port",
                     "sample": f"C5's problem at {c5_nx}^3 x 6 = {m.n_cells} tetrahedra, {nd} DOFs (full size: 87^3 x 6): oracle assembly {t1 - t0:.2f} s + "
                               f"Jacobi-BiCGStab {t2 - t1:.2f} s ({it} iterations, rc {rc}); a sparse LU of the full-size system does not fit the host's memory"}
    except Exception as e:
        out["c5"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


SOLVE_KERNEL_SOURCES = ("kernels_persist.h", "kernels_persist_bicg.h", "kernels_spmv.h", "kernels_krylov.h", "kernels_reduce.h", "persist_engine.hip",
                        "host_persist.cpp", "dev_persist.hip")


def solve_kernel_sha16():
    """digest of the sources of the solve's kernels and of the layout they stream: a committed counter pass (profiles/spmv_pmc.json) carries the digest
    it was collected on, so that a stale file after a kernel change does not go unnoticed"""
    import hashlib

    h = hashlib.sha256()
    for name in SOLVE_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "fdapde-core_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def roofline_of(info_list, alg_bytes, streamed_bytes, nx, world, n_int=None, nnz_int=None, layout=None):
    """The dominant kernel's roofline entry.  `frac` is PHYSICAL: the bytes the kernel's layout streams (and, where a committed PMC
    pass of the same workload exists, the counter bytes next to it as `traffic`) over the launch duration; the algorithmic bytes of
    the CSR operator the caller sees over the same time are `effective_*` (they exceed the physical figure where the layout stores
    less than the CSR form: symmetric storage, 16-bit column codes, no diagonal)."""
    import numpy as np

    info = info_list[-1]
    persistent = int(getattr(info, "persistent", 0))
    iters = max(int(info.iters), 1)
    r = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    alg_full = float(alg_bytes)
    if n_int is not None and nnz_int is not None:   # the solve runs on the interior block A_II (Dirichlet rows and columns eliminated): ITS CSR bytes
        alg_bytes = 12.0 * nnz_int + 4.0 * (n_int + 1) + 16.0 * n_int
        r["algorithmic_bytes_full_operator_per_application"] = alg_full
    if persistent:
        launch_ms = float(np.mean([i.launch_ms for i in info_list]))
        per_launch = float(streamed_bytes) * iters if streamed_bytes else None
        r.update({
            "kernel": "k_cg_persist: ONE launch runs the whole Jacobi-PCG (x, r, p in registers; the matrix blocks stream once per iteration, "
                      "neighbour entries and dot records cross through granule boards)",
            "avg_launch_ms": launch_ms, "iterations_per_launch": iters,
            "launch_timing": "HIP events recorded on the solver's stream right before and after the dispatch, averaged over the timed steps",
            "streamed_bytes_per_launch": per_launch,
            "streamed_bytes_per_iteration": float(streamed_bytes) if streamed_bytes else None,
            "algorithmic_bytes_per_launch": float(alg_bytes) * iters,
            "algorithmic_bytes_per_iteration": float(alg_bytes),
        })
        achieved = per_launch / (launch_ms * 1e-3) / 1e9 if per_launch and launch_ms > 0 else 0.0
        eff = float(alg_bytes) * iters / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        r["phase_stamps_us_per_iteration"] = {
            "operator_slowest_workgroup": 1e3 * float(info.spmv_avg_ms), "operator_mean": 1e3 * float(info.spmv_mean_ms),
            "allgather": 1e3 * float(info.gather_avg_ms), "update": 1e3 * float(info.update_avg_ms),
            "note": "s_memrealtime stamps inside the kernel (diagnostic; not what achieved / frac are computed from)"}
    else:
        spmv_ms = float(np.mean([i.spmv_avg_ms for i in info_list]))
        r.update({
            "kernel": "SpMV of the multi-launch Krylov iteration (k_spmv_team2 / k_spmv_blocked, fused with the dot products)",
            "avg_launch_ms": spmv_ms, "launches_timed_per_step": int(info.spmv_timed),
            "launch_timing": "HIP events attached to the timed dispatches on the solver's stream",
            "streamed_bytes_per_launch": float(streamed_bytes) if streamed_bytes else None,
            "algorithmic_bytes_per_launch": float(alg_bytes),
        })
        achieved = float(streamed_bytes if streamed_bytes else alg_bytes) / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
        eff = float(alg_bytes) / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    r["achieved"], r["frac"] = achieved, achieved / HBM_PEAK_GBPS
    r["achieved_source"] = "bytes of the kernel's own layout (fdapde_solver_layout) / measured launch duration"
    if layout is not None:
        r["layout"] = layout
        if layout.get("kind") == 3:   # (reduced sizes only: C3 streams) blocks resident in LDS -- the iteration is two hand-off latencies, no HBM stream
            r["bound"], r["achieved"], r["frac"] = "latency", None, None
            r["achieved_source"] = "matrix blocks resident in LDS for the whole launch: no HBM fraction applies"
    r["effective_gbps"], r["effective_frac"] = eff, eff / HBM_PEAK_GBPS
    r["effective_note"] = ("algorithmic CSR bytes (SURVEY 8d: 12 nnz + 4 (n+1) + 16 n per application) of the interior block the solve runs on, over the same "
                           "time; not a roofline fraction")
    if streamed_bytes and persistent:
        ic = 256 * 1024 * 1024
        r["residency"] = (f"{streamed_bytes / 1e6:.0f} MB streamed per iteration " + ("< " if streamed_bytes < ic else ">= ") + "the 256 MiB Infinity Cache: " +
                          ("the stream is Infinity-Cache resident from the second iteration on, so `frac` (of the HBM peak) is nominal for this size; the "
                           "HBM-resident evidence is extra.wide_2p35M" if streamed_bytes < ic else "an HBM-resident stream"))
    # HBM bytes per launch from counters: NOT measured in this run (PMC needs rocprofv3 around the process) -- taken from the committed
    # PMC passes of the same workload (tools/profile_gpu.sh -> profiles/spmv_pmc.json) when they match it, and labelled so
    r["traffic"], r["traffic_source"] = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "spmv_pmc.json")))
        if world == 1 and pj.get("nx") == nx:
            if persistent and pj.get("persist_hbm_bytes_per_solve") and abs(int(pj.get("persist_iterations", 0)) - iters) <= 2:
                r["traffic"] = float(pj["persist_hbm_bytes_per_solve"])
                r["traffic_frac"] = r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            elif not persistent and pj.get("hbm_bytes_per_launch"):
                r["traffic"] = float(pj["hbm_bytes_per_launch"])
            if r["traffic"] is not None:
                r["traffic_head"] = pj.get("head")   # the commit the counter passes ran on
                r["traffic_kernel_sha16"] = pj.get("kernel_source_sha16")
                try:
                    r["traffic_stale"] = pj.get("kernel_source_sha16") != solve_kernel_sha16()   # True: the solve's kernel sources changed since
                except OSError:
                    r["traffic_stale"] = None
                r["traffic_source"] = ("profiles/spmv_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE of this kernel, separate rocprofv3 --pmc passes of an "
                                       "earlier run of this workload (not this run); FETCH_SIZE counts Infinity-Cache hits")
    except Exception:
        pass
    if n_int is not None:
        r["interior_rows"], r["interior_nnz"] = int(n_int), int(nnz_int)
    return r


def run_single(args):
    import numpy as np

    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi, meshgen

    if capi.load().fdapde_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    device_index = int(os.environ.get("LOCAL_RANK", "0")) % max(int(capi.load().fdapde_device_count()), 1)
    t_gen = time.perf_counter()
    nodes, cells, bnd = meshgen.unit_cube(args.nx)
    t_gen = time.perf_counter() - t_gen
    n_cells_total = int(cells.shape[0])
    u_exact, f = meshgen.manufactured(3)

    def timed(fn):   # wall clock of one C-ABI call, the device idle when it ends
        t0 = time.perf_counter()
        r = fn()
        ctx.synchronize()
        return r, 1e3 * (time.perf_counter() - t0)

    t0 = time.perf_counter()
    ctx = capi.Context(device=device_index)
    t_ctx = 1e3 * (time.perf_counter() - t0)
    _, t_mesh_upload = timed(lambda: ctx.mesh_upload(nodes, cells, bnd))   # host copy + device copy of the mesh: the mesh is resident from here on
    n_dofs, t_dofs_build = timed(lambda: ctx.dofs_build(1))
    sizes = ctx.sizes()
    qn = ctx.quadrature_nodes()
    ctx.set_operator(-capi.laplacian())
    fq = f(qn)
    ctx.synchronize()
    _, t_set_forcing = timed(lambda: ctx.set_forcing(fq))   # upload of the samples + their re-layout in block-cell order (set-up for a static forcing; reported, untimed)
    _, t_set_dirichlet = timed(lambda: ctx.set_dirichlet(np.zeros(n_dofs)))
    del qn, fq
    _, t_prep = timed(lambda: ctx.solver_prepare(True))   # set-up: the solver's layout for this boundary mask
    t_prep *= 1e-3

    def step(time_spmv=0):
        ctx.init()
        return ctx.solve(rtol=RTOL, time_spmv=time_spmv)

    # the FIRST call of the path in this process: mesh resident on the device -> first solution.  What the reference does inside PDE(...), init()
    # and solve() the first time (lagrangian_basis.h:94-136 enumerate_dofs, fem_assembler.h:112-117 pattern, fem_linear_elliptic_solver.h:38-40
    # ordering + symbolic analysis) is dofs_build + set_forcing + set_dirichlet + solver_prepare here; the first init + solve is warm-up step 1
    first_call = {"mesh_upload_ms_not_counted": t_mesh_upload, "ctx_create_ms_not_counted": t_ctx, "dofs_build_ms": t_dofs_build,
                  "set_forcing_ms": t_set_forcing, "set_dirichlet_ms": t_set_dirichlet, "solver_prepare_ms": 1e3 * t_prep}
    for w in range(args.warmup):
        if w == 0:
            _, first_call["first_init_ms"] = timed(ctx.init)
            _, first_call["first_solve_ms"] = timed(lambda: ctx.solve(rtol=RTOL))
        else:
            step()
    first_call_ms = (sum(v for k, v in first_call.items() if not k.endswith("not_counted")) if args.warmup > 0 else None)
    ctx.synchronize()
    t0 = time.perf_counter()
    infos = [step(args.time_spmv) for _ in range(args.steps)]
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    info = infos[-1]
    t_asm = float(np.mean([i.t_assemble_ms for i in infos]))
    t_sol = float(np.mean([i.t_solve_ms for i in infos]))
    u = ctx.solution()
    _, _, coords = ctx.dofs_get()
    err = float(np.abs(u - u_exact(coords)).max())
    setup_ms = ctx.info().t_setup_ms + 1e3 * t_prep
    _, alg_bytes = ctx.bench_spmv(reps=1)
    n_int, nnz_int, streamed_bytes = ctx.solver_layout(True)
    layout = ctx.solver_layout_kind(True)
    want_parity = not args.no_cpu_baseline and args.cpu_nx == args.nx   # (the CPU leg runs the oracle on this very workload: compare entry by entry)
    c3_arrays = None
    if want_parity:
        from fdapde_core_amd import workloads as _w

        c3_arrays = _w.device_arrays(ctx, capi)
    ctx.close()
    del nodes, cells, bnd
    out = {
        "metric": "DOF/s assemble+solve, 3D P1 Laplacian; SpMV achieved HBM GB/s vs peak",
        "value": n_dofs * args.steps / elapsed,
        "unit": "DOF/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",   # the same mesh whatever N (bench.py --gpus N partitions it)
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"C3: 3-D P1 Laplacian, [0,1]^3, {args.nx}^3 x 6 Kuhn tetrahedra = {n_cells_total} cells, "
                        f"{n_dofs} DOFs, nnz {sizes['nnz']}, jitter 0.2h, ids permuted, seed 12345; "
                        "u = sin(pi x) sin(pi y) sin(pi z), homogeneous Dirichlet; init (stiff+force+mass) + Jacobi-PCG rtol 1e-10",
            "parallelism": "1 GPU",
            "cg_iterations": int(info.iters),
            "relres": float(info.relres),
            "t_assemble_ms": t_asm,
            "t_solve_ms": t_sol,
            "t_setup_ms_untimed": setup_ms,
            "t_set_forcing_ms_untimed": t_set_forcing,
            # mesh resident on the device -> first solution, in THIS (cold) process: dofs_build + set_forcing + set_dirichlet + solver_prepare + the
            # first init + the first solve; the timed steps that follow are the steady state
            "first_call_ms": first_call_ms,
            "first_call_phases_ms": first_call,
            "t_meshgen_s_untimed": t_gen,
            "max_abs_error_vs_analytic": err,
            "persistent_launch": int(getattr(info, "persistent", 0)),
            "us_per_iteration": 1e3 * t_sol / max(int(info.iters), 1),
        },
        "roofline": roofline_of(infos, alg_bytes, streamed_bytes, args.nx, 1, n_int, nnz_int, layout),
    }
    if not args.no_extra:
        # secondary results, after the C3 line's own timed region: the other single-GPU BASELINE configurations
        from fdapde_core_amd import workloads

        extra = {}
        # (the wide run BEFORE C5: measured right after C5's seconds of full-HBM BiCGStab the same launch took 101 instead of 74 us per iteration)
        for name, fn in (("c2", workloads.run_c2), ("wide_2p35M", workloads.run_wide), ("large_8p1M", workloads.run_large), ("c5", workloads.run_c5)):
            try:
                kw = {"keep_arrays": True} if name == "c2" and not args.no_cpu_baseline else {}
                extra[name] = fn(capi, meshgen, device=device_index, hbm_peak_gbps=HBM_PEAK_GBPS, **kw)
            except Exception as e:   # never let a secondary result take the bench line down
                extra[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:
            extra["c1"] = workloads.run_c1(capi, os.path.join(ROOT, "tests", "golden", "mesh"), device=device_index)
        except Exception as e:
            extra["c1"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:   # the mesh sharded behind the one-object interface (fdapde_ctx_create_multi): what the split costs, with the devices this box has
            extra["multi_device_context"] = workloads.run_group(capi, meshgen, nx=args.nx)
        except Exception as e:
            extra["multi_device_context"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["extra"] = extra
    c2_arrays = out.get("extra", {}).get("c2", {}).pop("_arrays", None) if isinstance(out.get("extra", {}).get("c2"), dict) else None
    parity_ok = True
    if not args.no_cpu_baseline:
        out["cpu_baseline"], out["cpu_baseline_all_cores"], c3_parity = cpu_baseline(args.cpu_nx, c3_arrays)
        del c3_arrays
        # entry-level parity with the oracle at the sizes the numbers are quoted on (bars: SURVEY 8d); a missed bar fails the bench (exit code 3)
        out["parity"] = {"c3": c3_parity if c3_parity is not None else
                         {"ok": None, "skipped": f"the CPU sample (--cpu-nx {args.cpu_nx}) is not this line's workload (--nx {args.nx})"}}
        if not args.no_extra:   # CPU columns beside the secondary results: direct solves where the reference's own solve is feasible (C1, C2)
            cols = cpu_columns(c2_gpu=c2_arrays)
            for name in ("c1", "c2", "c5"):
                if isinstance(out["extra"].get(name), dict):
                    out["extra"][name]["cpu"] = cols.get(name)
            if "c2_parity" in cols:
                out["parity"]["c2"] = cols["c2_parity"]
        parity_ok = all(v.get("ok") is not False for v in out["parity"].values())
        out["parity"]["ok"] = parity_ok
    print(json.dumps(out), flush=True)
    if not parity_ok:
        print("bench.py: the device path misses a parity bar against the oracle (see `parity` in the line above)", file=sys.stderr)
        raise SystemExit(3)


def _spawn_ranks(world, argv_extra, extra_env, timeout):
    """N fresh rank processes of this script (nothing of the caller's GPU state is inherited: they are new programs); -> exit code"""
    import socket
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rdzv = tempfile.mkdtemp(prefix="fdapde_rdzv_")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FDAPDE_BENCH_RDZV=rdzv, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        for k in [k for k in env if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")]:
            env.pop(k)   # (children of a rank started by torch.distributed.run must not look for that launcher's store)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv_extra, env=env, stdout=subprocess.DEVNULL if extra_env.get("FDAPDE_BENCH_CANARY") else None))
    rc, t0 = 0, time.time()
    try:
        pending = dict(enumerate(procs))
        while pending:
            for r, p in list(pending.items()):
                code = p.poll()
                if code is None:
                    continue
                del pending[r]
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the others", file=sys.stderr)
                    for q in pending.values():
                        q.terminate()
            if timeout and time.time() - t0 > timeout and pending:
                rc = rc or 124
                print("bench.py: rank processes timed out; stopping them", file=sys.stderr)
                for q in pending.values():
                    q.terminate()
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=5)
                except subprocess.TimeoutExpired:
                    p.kill()
        import shutil

        shutil.rmtree(rdzv, ignore_errors=True)
    return rc


def _rank_placement(capi, world):
    """(devices visible, transport backend, ranks per device) of a multi-rank job -- ONE rule for the canary's ranks and the real ones"""
    n_dev = int(capi.load().fdapde_device_count())
    if n_dev < 1:
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    backend = os.environ.get("FDAPDE_BENCH_BACKEND", "rccl")
    if backend == "nccl":
        backend = "rccl"
    if backend == "rccl" and n_dev < world:
        raise SystemExit(f"--gpus {world} but this node shows {n_dev} HIP devices (RCCL needs one device per rank; "
                         "FDAPDE_BENCH_BACKEND=gloo runs the plumbing with ranks sharing devices)")
    return n_dev, backend, (world + n_dev - 1) // n_dev


def run_canary(rank, world, local_rank):
    """one rank of the canary job (see dist.canary): exit code 0 = the row-distributed solve works between these devices"""
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi
    from fdapde_core_amd import dist as fdist

    try:
        n_dev, backend, share = _rank_placement(capi, world)
    except SystemExit:
        raise SystemExit(1)
    rdzv = FileRendezvous(rank, world)
    raise SystemExit(fdist.canary(capi, rdzv, rank, world, local_rank % n_dev, backend, share))


def choose_form(rdzv, rank, world):
    """row-distributed persistent launches where they work between the node's devices, the RCCL exchange otherwise.  Decided by rank 0
    BEFORE any rank touches a GPU, by running the canary job in processes of its own (FDAPDE_BENCH_EXCHANGE = rowdist | peers | dense
    forces a form)."""
    form = os.environ.get("FDAPDE_BENCH_EXCHANGE", "auto")
    if form != "auto":
        return form, f"not run (FDAPDE_BENCH_EXCHANGE={form} forces the form)"
    if rank == 0:
        rc = _spawn_ranks(world, ["--gpus", str(world)], {"FDAPDE_BENCH_CANARY": "1"}, timeout=600)
        if rc != 0:
            print(f"bench.py: the row-distributed canary job failed (exit code {rc}); using the RCCL neighbour exchange", file=sys.stderr)
        verdict = "passed: a row-distributed solve between these devices converged on every rank" if rc == 0 else f"FAILED (exit code {rc}): fell back to the RCCL neighbour exchange"
        rdzv.put("form", (("rowdist" if rc == 0 else "peers") + "\n" + verdict).encode())
    form, verdict = rdzv.get("form").decode().split("\n", 1)
    return form, verdict


def run_ranks(args, rank, world, local_rank):
    rdzv = FileRendezvous(rank, world)
    form, canary_verdict = choose_form(rdzv, rank, world)   # (may run a job of its own on the GPUs: nothing of this process has touched them yet)
    from fdapde_loader import load_package

    load_package()   # the library first: the process binds to /opt/rocm's HIP runtime, and RCCL is taken from the same installation
    from fdapde_core_amd import capi
    from fdapde_core_amd import dist as fdist

    n_dev, backend, share = _rank_placement(capi, world)
    out = fdist.bench_partitioned(capi, rdzv, rank, world, local_rank % n_dev, args, RTOL, backend, form, share)
    rdzv.finish()
    if rank != 0:
        return
    res = out
    info = res["info"]
    line = {
        "metric": ("DOF/s assemble+solve, 3D P2 advection-diffusion-reaction (BASELINE config C5)" if args.workload == "c5" else
                   "DOF/s assemble+solve, 3D P1 Laplacian; SpMV achieved HBM GB/s vs peak"),
        "value": res["total_dofs"] * args.steps / res["elapsed"],
        "unit": "DOF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * res["elapsed"] / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"C5: 3-D P2 advection-diffusion-reaction, b = (1, 0.5, 0.25), c = 1, {args.nx}^3 x 6 = {res['n_cells_total']} tetrahedra, "
                         f"{res['total_dofs']} DOFs; init + Jacobi-BiCGStab rtol 1e-10; 3-D P2 numbering build-defined (parity unpinned)"
                         if args.workload == "c5" else
                         ("C3" if args.nx == 119 else f"C3's problem grown to {world} GPUs (weak scaling: ~1.73 M rows per GPU)" if args.scaling == "weak" else "C3's problem") +
                         f": 3-D P1 Laplacian, [0,1]^3, {args.nx}^3 x 6 Kuhn tetrahedra = {res['n_cells_total']} cells, "
                         f"{res['total_dofs']} DOFs, jitter 0.2h, ids permuted, seed {args.mesh_seed}; "
                         "u = sin(pi x) sin(pi y) sin(pi z), homogeneous Dirichlet; init (stiff+force+mass) + Jacobi-PCG rtol 1e-10"),
            "parallelism": res["parallelism"],
            "cg_iterations": int(info.iters),
            "relres": float(info.relres),
            "t_assemble_ms": res["t_asm"],
            "t_solve_ms": res["t_sol"],
            "t_setup_ms_untimed": res["setup_ms"],
            # whole mesh on this rank's device -> its own sub-mesh, by the library's device-side partitioner (fdapde_partition_build), slowest rank; every
            # rank generates the mesh itself (t_meshgen: numpy, the bench's own generator -- not part of the path)
            "t_partition_s_untimed": res["t_partition"],
            "t_meshgen_s_untimed": res.get("t_meshgen"),
            "max_abs_error_vs_analytic": res["err"],
            "persistent_launch": int(getattr(info, "persistent", 0)),
            "us_per_iteration": 1e3 * res["t_sol"] / max(int(info.iters), 1),
            "transport": res["transport"],
            "exchange_form": res["form"],
            # what the collectives really ran over: ncclCommCount of the library's communicator (the registered world size under the gloo plumbing)
            "comm_ranks": res["comm_ranks"],
            "devices_visible": n_dev,
            "ranks_per_device": share,
            "canary": canary_verdict,
            "fallback": res.get("fallback"),
            # in-kernel phase stamps (persistent launches): per iteration, the SLOWEST rank's figure and the mean over the ranks
            "phase_stamps_us_per_iteration": res["phases"],
            # what this record should show on `world` real MI355X (dist.predict_c3): both forms, from single-GPU measurements + an
            # ASSUMED xGMI hop.  On a shared device (ranks_per_device > 1) the prediction does not apply: the ranks split one GPU's CUs
            "predicted_us_per_iteration": ({f: fdist.predict_c3(world, f, iterations=max(int(info.iters), 1)) for f in ("rowdist", "peers")}
                                           if args.workload == "c3" and args.nx == 119 else
                                           {f: fdist.predict_weak(world, f, args.nx, iterations=max(int(info.iters), 1)) for f in ("rowdist", "peers")}
                                           if args.workload == "c3" and args.scaling == "weak" else None),
        },
        "roofline": roofline_of(res["infos"], res["alg_bytes"], res["streamed_bytes"], args.nx, world),
    }
    line["extra"] = {"rccl_neighbour_exchange": res.get("other")}   # north_star's own form ("RCCL all-reduce of halo DOF contributions") next to the chosen one
    line["roofline"]["note"] = "the largest rank-local operator"
    if res["form"] == "rowdist":   # a rank's share of C3 is (nearly) resident in its LDS: the iteration is hand-off latency, not an HBM stream
        line["roofline"].update(bound="latency", achieved=None, frac=None,
                                achieved_source="row-distributed launches: per-rank blocks of the matrix are resident in LDS or stream from the rank's own "
                                                "Infinity Cache; an iteration is bound by the two in-kernel hand-offs (neighbour entries, dot all-gather over all "
                                                "ranks' workgroups), no HBM fraction is quoted")
    print(json.dumps(line), flush=True)


def main():
    # cross-process hipIpc on this pool needs the dmabuf IPC mode; set BEFORE anything loads the HIP runtime, and in every kind of rank process
    # (self-launched, canary, started by torch.distributed.run), so that the canary's verdict holds for the ranks that follow it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # extra.multi_device_context names the box's one GPU several times: its ranks' persistent launches need a hardware queue each (the runtime multiplexes
    # a process's streams onto 4 by default; real devices have their own) -- asked for before anything initialises the runtime; no effect on the other results
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    args = parse()
    if args.nx is None:
        args.nx = 87 if args.workload == "c5" else 119
        if args.scaling == "weak" and args.gpus > 1 and args.workload == "c3":
            args.nx = weak_nx(args.gpus)
    args.mesh_seed = MESH_SEEDS.get(args.nx, 12345)
    if args.workload == "c5" and args.gpus == 1:
        raise SystemExit("--workload c5 is the multi-GPU form of C5 (--gpus N > 1); on one GPU C5 is reported as `extra.c5` of the default run")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it plainly (python bench.py --gpus {args.gpus}) or with "
                         f"torch.distributed.run --nproc-per-node {args.gpus}")
    if world == 1:
        run_single(args)
    elif os.environ.get("FDAPDE_BENCH_CANARY"):
        run_canary(rank, world, local_rank)
    else:
        run_ranks(args, rank, world, local_rank)


if __name__ == "__main__":
    main()
