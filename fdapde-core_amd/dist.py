"""Element-partitioned multi-GPU driver: one process / one fdapde_ctx per GPU.

Host logic (numpy, identical on every rank, no communication needed because every rank sees the same mesh):
  * partition_cells      Morton chunks of the cells (equal counts) -> part[cell] = rank
  * local_problem        the sub-mesh of a rank (local node numbering) + interface maps:
                         interface DOFs = nodes touched by >= 2 ranks, globally indexed 0..n_if-1;
                         owner of a node = lowest rank touching it (each global DOF is counted once in dot products)
The device side (csrc/capi.hip) sums interface contributions with one RCCL all-reduce per operator application.
P1 only for now (DOF = node); P2 needs the edge numbering of the sub-meshes matched across ranks (next round).
"""
from __future__ import annotations

import numpy as np


def _morton_keys(pts: np.ndarray) -> np.ndarray:
    n, d = pts.shape
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    bits = 21 if d == 3 else 31
    q = np.minimum(((pts - lo) / span * ((1 << bits) - 1)).astype(np.uint64), (1 << bits) - 1)
    key = np.zeros(n, dtype=np.uint64)
    for b in range(bits):
        for k in range(d):
            key |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * d + k)
    return key


def partition_cells(nodes: np.ndarray, cells: np.ndarray, world: int) -> np.ndarray:
    """part[cell] in [0, world): contiguous chunks of the Morton order of the cell barycentres"""
    if world == 1:
        return np.zeros(cells.shape[0], dtype=np.int32)
    bary = nodes[cells].mean(axis=1)
    order = np.argsort(_morton_keys(bary), kind="stable")
    part = np.empty(cells.shape[0], dtype=np.int32)
    bounds = np.linspace(0, cells.shape[0], world + 1).astype(np.int64)
    for r in range(world):
        part[order[bounds[r]:bounds[r + 1]]] = r
    return part


def interface_info(cells: np.ndarray, part: np.ndarray, n_nodes: int, world: int):
    """-> mult[node] (ranks touching it), owner[node] (lowest such rank), ifnodes (sorted global ids with mult >= 2)"""
    mult = np.zeros(n_nodes, dtype=np.int32)
    owner = np.full(n_nodes, world, dtype=np.int32)
    for r in range(world):
        touched = np.unique(cells[part == r])
        mult[touched] += 1
        owner[touched] = np.minimum(owner[touched], r)
    return mult, owner, np.nonzero(mult >= 2)[0]


def local_problem(nodes, cells, boundary, part, rank, world, info=None):
    """Sub-mesh of `rank` and its interface maps (see fdapde_halo_setup in include/fdapde_hip.h)."""
    mult, owner, ifnodes = info if info is not None else interface_info(cells, part, nodes.shape[0], world)
    my_cells = np.nonzero(part == rank)[0]
    l2g = np.unique(cells[my_cells])                       # local node id -> global node id (sorted)
    local_cells = np.searchsorted(l2g, cells[my_cells]).astype(np.int32)
    is_if = mult[l2g] >= 2
    local_dof = np.nonzero(is_if)[0].astype(np.int32)      # P1: DOF = node
    if_index = np.searchsorted(ifnodes, l2g[local_dof]).astype(np.int32)
    return dict(
        nodes=np.ascontiguousarray(nodes[l2g]), cells=np.ascontiguousarray(local_cells),
        boundary=np.ascontiguousarray(boundary[l2g]), l2g=l2g, cell_ids=my_cells,
        n_if_global=int(ifnodes.size), local_dof=local_dof, if_index=if_index,
        owned=(owner[l2g] == rank).astype(np.uint8),
    )


def bench_partitioned(capi, nodes, cells, bnd, f, u_exact, rank, world, local_rank, args, barrier, rtol, backend="nccl"):
    """bench.py's N > 1 leg: the same C3 mesh split over `world` GPUs (strong scaling)."""
    import time

    import torch
    import torch.distributed as dist

    part = partition_cells(nodes, cells, world)
    lp = local_problem(nodes, cells, bnd, part, rank, world)
    dev = "cuda" if backend == "nccl" else "cpu"
    ctx = capi.Context(device=local_rank)
    ctx.mesh_upload(lp["nodes"], lp["cells"], lp["boundary"])
    n_loc = ctx.dofs_build(1)
    if backend == "nccl":   # RCCL communicator of the library, bootstrapped with a broadcast of its unique id
        uid = [capi.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])
    else:                   # plumbing checks only: host-staged all-reduce over gloo
        ctx.comm_init_callback(world, rank, lambda arr: dist.all_reduce(torch.from_numpy(arr)))
    ctx.halo_setup(lp["n_if_global"], lp["local_dof"], lp["if_index"], lp["owned"])
    qn = ctx.quadrature_nodes()
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(f(qn))
    ctx.set_dirichlet(np.zeros(n_loc))
    del qn

    def step(time_spmv=0):
        ctx.init()
        return ctx.solve(rtol=rtol, time_spmv=time_spmv)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    infos = [step(args.time_spmv) for _ in range(args.steps)]
    barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)       # slowest rank defines the step time
    u = ctx.solution()
    err = torch.tensor([float(np.abs(u - u_exact(lp["nodes"])).max())], dtype=torch.float64, device=dev)
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    info = infos[-1]
    stats = torch.tensor([np.mean([i.spmv_avg_ms for i in infos]), np.mean([i.t_assemble_ms for i in infos]),
                          np.mean([i.t_solve_ms for i in infos]), ctx.info().t_setup_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    sizes = ctx.sizes()
    _, alg_bytes = ctx.bench_spmv(reps=1)
    nnz_tot = torch.tensor([float(sizes["nnz"]), alg_bytes], dtype=torch.float64, device=dev)
    dist.all_reduce(nnz_tot, op=dist.ReduceOp.MAX)       # the largest local matrix bounds the SpMV roofline figure
    sizes = dict(sizes, nnz=int(nnz_tot[0].item()))
    parallelism = (f"{world} GPUs, element partition (Morton chunks), {lp['n_if_global']} interface DOFs, "
                   "1 RCCL all-reduce of interface entries + p.Ap and 1 scalar all-reduce per CG iteration; "
                   "roofline figures are the largest rank-local SpMV")
    return (float(elapsed.item()), info, float(stats[0].item()), float(stats[1].item()), float(stats[2].item()),
            float(err.item()), float(stats[3].item()), float(nnz_tot[1].item()), sizes, int(nodes.shape[0]), parallelism)
