#!/usr/bin/env python3
"""Randomised multi-rank runs with the ranks sharing ONE GPU (tests/dist_gpu_worker.py: every rank checks its part of the solution against the single-domain
solve, and for small meshes against the CPU oracle): random number of ranks (2-5), mesh size, case (P1 / P2 / 2-D P2 / advection-diffusion-reaction / parabolic /
handle) and exchange form (row-distributed single launches / neighbour exchange / dense interface all-reduce).  Lives in tools/ but runs the TEST worker.
usage: fuzz_dist.py [cases] [seed]"""
import os
import random
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def run(world, nx, case, exchange):
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(r), str(world), port, str(nx), case, "shared", exchange],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    ok = True
    for p in procs:
        try:
            out, _ = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            return False, "timeout"
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        if p.returncode != 0 or f"rank {r}: ok" not in out:
            ok = False
    return ok, "\n".join(o[-1500:] for o in outs) if not ok else ""


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for k in range(n_cases):
    world = rnd.choice([2, 2, 3, 3, 4, 5, 6, 8])
    case = rnd.choice(["p1", "p1", "p2", "sq2", "adr1", "adr2", "parab", "handle"])
    exchange = rnd.choice(["rowdist", "rowdist", "peers", "dense"])
    if case in ("p2", "adr2"):
        nx = rnd.randint(4, 9)
    elif case == "sq2":
        nx = rnd.randint(8, 40)
    else:
        nx = rnd.randint(6, 22)
    if exchange == "dense" and case in ("handle", "sq2"):
        exchange = "peers"
    ok, msg = run(world, nx, case, exchange)
    print(f"case {k}: world {world} nx {nx} {case} {exchange}: {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        fails += 1
        print(msg, flush=True)
print(f"{n_cases} multi-rank cases, failures {fails}")
sys.exit(1 if fails else 0)
