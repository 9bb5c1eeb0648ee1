"""BASELINE config C5: 3-D P2 advection-diffusion-reaction (non-symmetric -> Jacobi-BiCGStab), 87^3 x 6 = 3 951 018 tetrahedra.
One GPU:   python tools/run_c5.py
N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/run_c5.py
           (element partition, distributed BiCGStab, P2 DOFs matched across ranks by keys; FDAPDE_BENCH_BACKEND=gloo swaps RCCL
           for a host-staged all-reduce so that several ranks can share one GPU in plumbing checks).  NX=<n> changes the size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from fdapde_loader import load_package

load_package()
from fdapde_core_amd import capi, meshgen
from fdapde_core_amd import dist as fdist


def main():
    nx = int(os.environ.get("NX", "87"))
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    backend = os.environ.get("FDAPDE_BENCH_BACKEND", "nccl")
    t = time.time()
    nodes, cells, bnd = meshgen.unit_cube(nx)
    t_gen = time.time() - t
    b, c, pi = np.array([1.0, 0.5, 0.25]), 1.0, np.pi
    u = lambda x: np.prod(np.sin(pi * x), axis=1)

    def f(x):
        s_, c_ = np.sin(pi * x), np.cos(pi * x)
        grad = np.stack([pi * c_[:, 0] * s_[:, 1] * s_[:, 2], pi * s_[:, 0] * c_[:, 1] * s_[:, 2], pi * s_[:, 0] * s_[:, 1] * c_[:, 2]], axis=1)
        return 3 * pi**2 * u(x) + grad @ b + c * u(x)

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, init_method="env://", rank=rank, world_size=world)
    n_cells_total, n_g = cells.shape[0], nodes.shape[0]
    t = time.time()
    ctx = capi.Context(local_rank if backend == "nccl" else 0)
    if world > 1:
        part = fdist.partition_cells(nodes, cells, world)
        info_if = fdist.interface_info(cells, part, n_g, world, 2, bnd)
        sub = fdist.sub_mesh(nodes, cells, bnd, part, rank)
        ctx.mesh_upload(sub["nodes"], sub["cells"], sub["boundary"])
        nd = ctx.dofs_build(2)
        table, _, coords = ctx.dofs_get()
        maps = fdist.interface_maps(sub, table, info_if, rank, n_g, 2)
        nd_total = int(info_if[0].size)
        del table, sub, info_if, part
        ctx.dofs_set_boundary(maps["boundary_dofs"])
        if backend == "nccl":
            uid = [capi.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(world, rank, uid[0])
        else:
            ctx.comm_init_callback(world, rank, lambda arr: dist.all_reduce(torch.from_numpy(arr)))
        ctx.halo_setup(maps["n_if_global"], maps["local_dof"], maps["if_index"], maps["owned"])
    else:
        ctx.mesh_upload(nodes, cells, bnd)
        nd = nd_total = ctx.dofs_build(2)
        _, _, coords = ctx.dofs_get()
    del nodes, cells
    t_setup = time.time() - t
    s = ctx.sizes()
    ctx.set_operator(-capi.laplacian() + capi.advection(b) + capi.reaction(c))
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    ctx.set_dirichlet(np.zeros(nd))
    if not dist:
        ctx.tune("pmg_auto", 0)   # (this tool profiles the Jacobi-BiCGStab stage: k_spmv_blocked and the vector kernels)
    wall = 0.0
    for i in range(2):                                   # second pass is the measured one
        if dist:
            dist.barrier()
        t = time.time()
        ctx.init()
        info = ctx.solve(rtol=1e-10)
        if dist:
            dist.barrier()
        wall = time.time() - t
    err = float(np.abs(ctx.solution() - u(coords)).max())
    if dist:
        import torch

        dev = "cuda" if backend == "nccl" else "cpu"
        v = torch.tensor([err, wall], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        err, wall = float(v[0]), float(v[1])
    if rank == 0:
        print(f"C5 nx={nx} ranks={world}: cells {n_cells_total} dofs {nd_total} (rank 0: {nd}, nnz {s['nnz']}) | meshgen {t_gen:.1f}s "
              f"setup {t_setup:.1f}s | assemble {info.t_assemble_ms:.2f} ms  solve {info.t_solve_ms:.2f} ms  wall {wall * 1e3:.1f} ms  "
              f"iters {info.iters} method {info.method_used} relres {info.relres:.2e} | max err {err:.2e} | "
              f"DOF/s {nd_total / wall:.3e}")
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
