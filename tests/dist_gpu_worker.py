"""Worker of tests/test_gpu_dist.py: one rank of a 2-rank job in which BOTH ranks share GPU 0.  The device-side distributed
path (sub-assembly, interface pack / all-reduce / unpack, owner-masked dots, stop decisions) is the product's; only the
all-reduce transport is swapped for a host-staged torch.distributed/gloo callback, because RCCL refuses two ranks on one
device.  Result is compared with a single-domain solve of the whole mesh on the same GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, nx = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    import torch
    import torch.distributed as dist

    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi, meshgen
    from fdapde_core_amd import dist as fdist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    nodes, cells, bnd = meshgen.unit_cube(nx)
    u_exact, f = meshgen.manufactured(3)
    g_fn = lambda x: 0.3 * x[:, 0] - 0.2 * x[:, 2]
    part = fdist.partition_cells(nodes, cells, world)
    lp = fdist.local_problem(nodes, cells, bnd, part, rank, world)

    def allreduce(arr):
        t = torch.from_numpy(arr)
        dist.all_reduce(t)

    ctx = capi.Context(device=0)
    ctx.mesh_upload(lp["nodes"], lp["cells"], lp["boundary"])
    n_loc = ctx.dofs_build(1)
    ctx.comm_init_callback(world, rank, allreduce)
    ctx.halo_setup(lp["n_if_global"], lp["local_dof"], lp["if_index"], lp["owned"])
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    ctx.set_dirichlet(g_fn(lp["nodes"]))
    ctx.init()
    info = ctx.solve(rtol=1e-11)
    u = ctx.solution()
    # single-domain solve of the whole mesh (every rank does it; same GPU)
    ref = capi.Context(device=0)
    ref.mesh_upload(nodes, cells, bnd)
    ref.dofs_build(1)
    ref.set_operator(-capi.laplacian())
    ref.set_forcing(f(ref.quadrature_nodes()))
    ref.set_dirichlet(g_fn(nodes))
    ref.init()
    rinfo = ref.solve(rtol=1e-11)
    uref = ref.solution()
    err = np.linalg.norm(u - uref[lp["l2g"]]) / np.linalg.norm(uref)
    assert info.converged == 1 and err < 1e-8, (info.converged, err)
    assert abs(info.iters - rinfo.iters) <= 2, (info.iters, rinfo.iters)     # same Krylov iteration up to rounding
    print(f"rank {rank}: ok  local dofs {n_loc}  interface {lp['local_dof'].size}/{lp['n_if_global']}  iters {info.iters} "
          f"(single domain {rinfo.iters})  err {err:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
