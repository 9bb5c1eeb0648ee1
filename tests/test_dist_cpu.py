"""world_size-2 (and 3) gloo jobs on CPU covering the N > 1 path's host logic: partition, sub-meshes, interface maps,
ownership (P1 and P2: DOFs matched across ranks by node / edge keys), and the distributed Jacobi-PCG and BiCGStab recurrences (tests/dist_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,case", [(2, "unit_sphere"), (2, "square"), (3, "cube"), (2, "square:2"), (3, "cube:2"),
                                        (2, "cube:1:adr"), (2, "unit_sphere:2:adr")])
def test_partitioned_pcg_matches_single_domain(world, case):
    port = str(_free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), port, case],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out}"
        assert f"rank {r}: ok" in out
