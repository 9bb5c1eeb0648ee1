"""The device-side element-partitioned solve end to end, against the single-domain solve of the same mesh.
  * ranks sharing one GPU, a host-staged gloo all-reduce in place of RCCL (which cannot put two ranks on one device): runs on
    every GPU box;
  * one rank per GPU over the library's own RCCL communicator (the product configuration, BASELINE config C4's transport): runs
    whenever the box has >= 2 GPUs, skipped otherwise;
  * bench.py --gpus 2 under torch.distributed.run, exactly as the driver launches it (>= 2 GPUs).
See tests/dist_gpu_worker.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    # through the library, not torch: importing torch here would load the ROCm runtime bundled with the wheel into a process that
    # may already hold the system's (another test file loaded libfdapde_hip.so first), and RCCL then finds no device
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi

    return int(capi.load().fdapde_device_count())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run_ranks(world, nx, case, transport, exchange="peers"):
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(r), str(world), port, str(nx), case,
                               transport, exchange], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out}"
        assert f"rank {r}: ok" in out


@pytest.mark.parametrize("world,nx,case", [(2, 12, "p1"), (3, 10, "p1"), (2, 8, "p2"), (3, 20, "sq2"), (2, 10, "adr1"), (3, 7, "adr2"),
                                           (2, 9, "parab"), (2, 9, "handle")])
def test_partitioned_device_solve_matches_single_domain(world, nx, case):
    """neighbour-only exchange (fdapde_halo_setup_peers): per-peer packed segments, contributions summed in rank order"""
    _run_ranks(world, nx, case, "shared")


@pytest.mark.parametrize("world,nx,case,exchange", [(2, 10, "indef", "peers"), (3, 10, "indef", "peers"), (2, 12, "pe30", "peers"), (2, 10, "indef", "rowdist"),
                                                    (3, 12, "indef", "rowdist")])
def test_open_method_hands_over_across_ranks(world, nx, case, exchange):
    """FDAPDE_SOLVER_AUTO on the ranks of a multi-GPU job: a symmetric indefinite operator breaks CG -- every rank agrees on it and all of
    them re-prepare and run BiCGStab; an advection-dominated one breaks BiCGStab down -- all ranks restart it from the iterate together.  Both used to end with
    the breakdown reported (one-GPU contexts only took these turns).  Against the single-domain solve of the whole mesh.  (No "pe30" in the row-distributed
    form: at that Peclet number rows have a non-positive diagonal, which that form declines -- FDAPDE_EUNSUPPORTED on every rank -- and the caller takes the
    element form, as the multi-device context does by itself.)"""
    _run_ranks(world, nx, case, "shared", exchange)


@pytest.mark.parametrize("world,nx,case", [(2, 12, "p1"), (3, 10, "p1"), (3, 7, "adr2"), (2, 9, "parab")])
def test_partitioned_device_solve_dense_interface_allreduce(world, nx, case):
    """the round-1 exchange (fdapde_halo_setup: one all-reduce of the whole interface vector) stays available"""
    _run_ranks(world, nx, case, "shared", "dense")


@pytest.mark.parametrize("world,nx,case", [(2, 24, "p1"), (2, 8, "p2"), (2, 10, "adr1"), (2, 9, "parab"), (2, 9, "handle"), (4, 24, "p1"),
                                           (8, 30, "p1"), (8, 9, "adr2")])
def test_partitioned_solve_over_real_rccl(world, nx, case):
    """One rank per GPU, the library's RCCL communicator (ncclAllReduce over xGMI) -- BASELINE config C4's transport.  Solution and
    iteration count against the single-domain run <= 1e-8 (asserted in the worker)."""
    if _n_gpus() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_n_gpus()}")
    _run_ranks(world, nx, case, "rccl")


@pytest.mark.parametrize("world,nx,case", [(2, 16, "p1"), (3, 14, "p1"), (4, 20, "p1"), (2, 40, "sq2"), (2, 8, "p2"), (4, 36, "p1"),
                                           (2, 16, "adr1"), (3, 8, "adr2"), (2, 12, "parab"), (3, 10, "parab"), (3, 12, "handle"), (3, 16, "stall"),
                                           (3, 14, "p1:2level"), (2, 16, "adr1:2level"), (3, 16, "stall:2level"),
                                           (2, 12, "p1"), (3, 10, "adr1"), (2, 6, "p2"),   # small enough for the oracle's direct solve of the whole mesh
                                           (3, 12, "fail0"), (2, 12, "fail1"), (3, 12, "fail2"), (2, 12, "fail3"), (3, 12, "fail4")])
def test_row_distributed_persistent_launches_share_one_gpu(world, nx, case):
    """fdapde_rowdist_setup: every rank assembles the complete rows of the DOFs it owns (ghost layer of cells) and the whole CG runs as ONE
    launch per rank; the launches exchange search-direction entries and dot records through each other's boards (hipIpc-mapped across the
    processes), no collective inside the iteration (dot records in one hop -- the automatic choice up to 1024 workgroups -- or, ":2level",
    as rank records in two).  Against the single-domain solve: same iteration count, solution <= 1e-9, bitwise
    repeatable -- and, for the cases of nx <= 12, directly against the CPU oracle's direct solve of the whole mesh (<= 1e-8 over the owned DOFs).
    "failK": a hard local failure of one rank at stage K of the collective set-up: no rank is left waiting, the failing rank reports its
    error, the others the collective refusal.  All ranks on GPU 0 with an equal share of its CUs each."""
    _run_ranks(world, nx, case, "shared", "rowdist")


@pytest.mark.parametrize("world,nx,case", [(2, 24, "p1"), (2, 8, "p2"), (2, 16, "adr1"), (2, 12, "parab"), (2, 16, "stall"), (4, 30, "p1"), (8, 40, "p1"), (8, 12, "adr2")])
def test_row_distributed_launches_over_xgmi(world, nx, case):
    """the product configuration of the row-distributed form: one rank per GPU, boards mapped across the devices through hipIpc, entries
    and rank records pushed over xGMI, set-up over the library's RCCL communicator.  Needs `world` GPUs (skipped on the 1-GPU boxes of this
    pool, where the same code runs with the ranks sharing one device: test_row_distributed_persistent_launches_share_one_gpu)."""
    if _n_gpus() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_n_gpus()}")
    _run_ranks(world, nx, case, "rccl", "rowdist")


def _check_enlarged_line(rec, world, nx):
    """one SCALE record answers everything: the chosen form with its in-kernel phase stamps (slowest rank), how many
    ranks the communicator really had, the canary's verdict, and north_star's own form -- the RCCL neighbour exchange -- measured for the
    same number of steps next to it"""
    cfg = rec["config"]
    assert cfg["comm_ranks"] == world and cfg["canary"].startswith("passed") and cfg["fallback"] is None
    assert "predicted_us_per_iteration" in cfg   # (filled for the full-size C3 mesh only: DESIGN 7.2's acceptance table, dist.predict_c3)
    ph = cfg["phase_stamps_us_per_iteration"]
    assert ph["operator_slowest_workgroup_slowest_rank"] >= ph["operator_mean_workgroup_mean_of_ranks"] > 0
    assert ph["allgather_slowest_rank"] >= ph["allgather_mean_of_ranks"] > 0
    other = rec["extra"]["rccl_neighbour_exchange"]
    assert "error" not in other, other
    assert other["iterations"] > 0 and other["relres"] <= 1e-10 and other["ms_per_step"] > 0 and other["us_per_iteration"] > 0
    assert other["comm_ranks"] == world and "bytes sent per rank" in other["parallelism"]
    assert other["max_abs_error_vs_analytic"] < 6.0 * (1.0 / nx) ** 2 * 3.15**2
    # both forms run the same Krylov iteration on the same system (single-reduction CG may need a few iterations more)
    assert abs(other["iterations"] - cfg["cg_iterations"]) <= max(3, cfg["cg_iterations"] // 10)


def test_bench_multi_gpu_leg_plumbing_on_one_gpu():
    """bench.py --gpus 2 under torch.distributed.run with FDAPDE_BENCH_BACKEND=gloo: both ranks share GPU 0 and the exchange is
    host-staged, everything else -- partition, neighbour lists, the device-side pack / sum kernels, the JSON line -- is what the
    driver's real launch runs"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FDAPDE_BENCH_BACKEND="gloo", FDAPDE_BENCH_EXCHANGE="peers")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--nx", "24",
           "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["relres"] <= 1e-10
    assert "bytes sent per rank" in rec["config"]["parallelism"]
    assert rec["config"]["max_abs_error_vs_analytic"] < 6.0 * (1.0 / 24) ** 2 * 3.15**2
    # the enlarged line: what the collectives ran over, why this form, and that no second form was measured because this IS the RCCL form
    assert rec["config"]["comm_ranks"] == 2 and rec["config"]["canary"].startswith("not run") and rec["config"]["fallback"] is None
    assert rec["config"]["ranks_per_device"] == 2 // min(2, rec["config"]["devices_visible"])
    assert "IS the RCCL neighbour exchange" in rec["extra"]["rccl_neighbour_exchange"]
    assert not [d for d in os.listdir("/tmp") if d.startswith(f"fdapde_rdzv_{cmd[cmd.index('--master-port') + 1]}_")], "the ranks' rendezvous directory must not outlive the job"


def test_bench_under_torchrun_runs_the_canary_and_takes_the_row_distributed_form():
    """the driver's N > 1 command -- python -m torch.distributed.run ... bench.py --gpus N -- with the form left to the bench: rank 0 starts
    the canary job (processes of its own, before any rank touches the GPU; the launcher's environment must not leak into them), all ranks
    then run the row-distributed solve.  Gloo plumbing mode, ranks share GPU 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FDAPDE_BENCH_BACKEND="gloo")
    env.pop("FDAPDE_BENCH_EXCHANGE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--nx", "24"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["relres"] <= 1e-10
    assert rec["config"]["exchange_form"] == "rowdist" and rec["config"]["persistent_launch"] == 1
    assert rec["config"]["max_abs_error_vs_analytic"] < 6.0 * (1.0 / 24) ** 2 * 3.15**2
    _check_enlarged_line(rec, 2, 24)


def test_bench_falls_back_when_the_row_distributed_solve_declines_the_system():
    """the canary proves the mechanism on a small mesh; if the library then declines the real system (a rank's share does not fit one
    launch), every rank gets the same refusal and the bench re-runs the leg with the RCCL neighbour exchange"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FDAPDE_BENCH_BACKEND="gloo", FDAPDE_BENCH_EXCHANGE="rowdist", FDAPDE_ROWDIST_REFUSE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FDAPDE_BENCH_RDZV"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--nx", "20"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["config"]["exchange_form"] == "peers" and rec["config"]["relres"] <= 1e-10 and "declined" in out.stderr


def test_bench_c5_workload_across_ranks():
    """BASELINE config C5 in its multi-GPU form (3-D P2 advection-diffusion-reaction, Jacobi-BiCGStab, row-distributed: one persistent
    launch per rank) through bench.py's launcher, reduced mesh, ranks sharing GPU 0"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FDAPDE_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FDAPDE_BENCH_RDZV", "FDAPDE_BENCH_EXCHANGE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--workload", "c5", "--steps", "1", "--warmup", "1", "--nx", "10"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 3 and "C5" in rec["config"]["workload"] and rec["config"]["exchange_form"] == "rowdist"
    assert rec["config"]["persistent_launch"] == 1 and rec["config"]["relres"] <= 1e-10
    assert f"{21**3} DOFs" in rec["config"]["workload"]   # every P2 DOF of the 10^3 x 6 mesh is owned exactly once
    assert rec["config"]["max_abs_error_vs_analytic"] < 4e-2   # second-order accurate only (the reference's 5-point rule)


@pytest.mark.parametrize("gpus", [2, 3])
def test_bench_starts_its_own_ranks_without_a_launcher(gpus):
    """`python bench.py --gpus N` with WORLD_SIZE unset -- the shape of the driver's N = 1 command: the script starts N fresh rank
    processes itself (decided before anything touches the GPU) and relays rank 0's JSON line.  Gloo plumbing mode, ranks share GPU 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FDAPDE_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FDAPDE_BENCH_RDZV"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "1", "--nx", "20"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == gpus and rec["scaling"] == "strong" and rec["value"] > 0 and rec["config"]["relres"] <= 1e-10
    assert rec["config"]["max_abs_error_vs_analytic"] < 6.0 * (1.0 / 20) ** 2 * 3.15**2
    # the canary job found the row-distributed launches working between the ranks' devices (here: one shared device): that form ran
    assert rec["config"]["exchange_form"] == "rowdist" and rec["config"]["persistent_launch"] == 1
    assert all(v is None or v <= 1.0 for k, v in rec["roofline"].items() if k in ("frac", "traffic_frac"))
    _check_enlarged_line(rec, gpus, 20)


def test_bench_self_launch_reports_a_failing_rank():
    """a rank that cannot run (more ranks than devices over RCCL) must end the whole job with a non-zero code, not leave the others waiting"""
    if _n_gpus() >= 2:
        pytest.skip("needs a box with ONE GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FDAPDE_BENCH_BACKEND", "FDAPDE_BENCH_RDZV"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--nx", "8"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "HIP devices" in out.stderr


@pytest.mark.parametrize("gpus", [2, 8])
def test_bench_multi_gpu_leg_as_the_driver_launches_it(gpus):
    """python -m torch.distributed.run --nproc-per-node N bench.py --gpus N on a reduced mesh: one JSON line, converged, the
    analytic error at the level of the single-GPU run (the partitioned solve is the same Krylov iteration)."""
    if _n_gpus() < gpus:
        pytest.skip(f"needs {gpus} GPUs, this box has {_n_gpus()}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for launcher in ("torchrun", "self"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
               "--master-port", _free_port()] if launcher == "torchrun" else [sys.executable]
        cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--nx", "48"]
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
        assert out.returncode == 0, (launcher, out.stderr[-3000:])
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        rec = json.loads(line)
        assert rec["n_gpus"] == gpus and rec["scaling"] == "strong" and rec["value"] > 0
        assert rec["config"]["relres"] <= 1e-10 and rec["config"]["max_abs_error_vs_analytic"] < 6.0 * (1.0 / 48) ** 2 * 3.15**2
        assert "/opt/rocm" in rec["config"]["transport"], rec["config"]["transport"]   # ONE stack: the system's RCCL next to the system's HIP
        assert rec["config"]["comm_ranks"] == gpus   # ncclCommCount of the communicator the collectives ran over
        if rec["config"]["exchange_form"] == "rowdist":
            _check_enlarged_line(rec, gpus, 48)
