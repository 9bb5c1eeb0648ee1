"""Set-up and step time of the multi-device context (fdapde_ctx_create_multi) at C3's size with the "devices" all GPU 0, next to the single-device
context: what the split costs (<= 0.5 s for partition + rank set-up; dist.py needed 16.3 s).  usage: group_time.py [nx] [n_dev ...]"""
import sys
import time

import numpy as np

import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # (ranks sharing ONE GPU: a hardware queue per launch, see tests/conftest.py)
sys.path.insert(0, ".")
from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 119
devs = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
nodes, cells, bnd = meshgen.unit_cube(nx)
u_exact, f = meshgen.manufactured(3)
n_real = capi.load().fdapde_device_count()
for n in devs:
    dl = [r % n_real for r in range(n)]
    t0 = time.perf_counter()
    c = capi.Context(0) if n == 1 else capi.Context(devices=dl)
    t1 = time.perf_counter()
    c.mesh_upload(nodes, cells, bnd)
    t2 = time.perf_counter()
    nd = c.dofs_build(1)
    if n > 1 and os.environ.get("GROUP_FORM"):   # (measurements: the form pinned; GROUP_DIRECT=0: the element form's exchange staged through host memory)
        c.tune("group_direct", int(os.environ.get("GROUP_DIRECT", "1")))
        c.tune("group_form", int(os.environ["GROUP_FORM"]))
    t3 = time.perf_counter()
    c.set_operator(-capi.laplacian())
    fq = f(c.quadrature_nodes())
    t4 = time.perf_counter()
    c.set_forcing(fq)
    c.set_dirichlet(np.zeros(nd))
    t5 = time.perf_counter()
    c.init()
    info = c.solve(rtol=1e-10)   # (first solve: layouts, boards)
    t6 = time.perf_counter()
    steps = []
    for _ in range(3):
        c.synchronize()
        s0 = time.perf_counter()
        c.init()
        info = c.solve(rtol=1e-10)
        c.synchronize()
        steps.append(time.perf_counter() - s0)
    _, _, coords = c.dofs_get()
    err = float(np.abs(c.solution() - u_exact(coords)).max())
    d = c.devices()
    print(f"nx {nx} devices {dl}: create {1e3 * (t1 - t0):.1f} ms, mesh_upload {1e3 * (t2 - t1):.1f} ms, dofs_build {1e3 * (t3 - t2):.1f} ms "
          f"(partition {d['t_partition_ms']:.1f} + rank set-up {d['t_rank_setup_ms']:.1f}), set_forcing+dirichlet {1e3 * (t5 - t4):.1f} ms, first init+solve {1e3 * (t6 - t5):.1f} ms; "
          f"step {1e3 * min(steps):.2f} ms ({info.iters} iterations, {1e3 * info.t_solve_ms / max(info.iters, 1):.1f} us each, persistent {info.persistent}, form {d['form']}), "
          f"err {err:.2e}", flush=True)
    c.close()
