"""Two ranks sharing one GPU: the device-side element-partitioned solve end to end with a host-staged gloo all-reduce in
place of RCCL (which cannot put two ranks on one device).  See tests/dist_gpu_worker.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,nx,case", [(2, 12, "p1"), (3, 10, "p1"), (2, 8, "p2"), (3, 20, "sq2"), (2, 10, "adr1"), (3, 7, "adr2"),
                                           (2, 9, "parab"), (2, 9, "handle")])
def test_partitioned_device_solve_matches_single_domain(world, nx, case):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(r), str(world), port, str(nx), case],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
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
