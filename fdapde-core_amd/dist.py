"""Element-partitioned multi-GPU driver: one process / one fdapde_ctx per GPU.

Host logic (numpy, identical on every rank, no communication needed because every rank sees the same mesh):
  * partition_cells      Morton chunks of the cells (equal counts) -> part[cell] = rank
  * local_problem        the sub-mesh of a rank (local node numbering) + interface maps:
                         interface DOFs = nodes touched by >= 2 ranks, globally indexed 0..n_if-1;
                         owner of a node = lowest rank touching it (each global DOF is counted once in dot products)
The device side (csrc/capi.hip) sums interface contributions with one RCCL all-reduce per operator application.
DOFs are identified across ranks by numbering-independent keys (node id, or the end-node pair of an edge for P2), so
every rank keeps the reference's enumeration on its own sub-mesh.
"""
from __future__ import annotations

import os
import sys

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


# vertex pair of the edge DOF in local slot (M + 1) + j  (csrc/tables.cpp EDGE2 / EDGE3; 2-D: the reference's m01, m02, m12)
_EDGE_SLOTS = {3: ((0, 1), (0, 2), (1, 2)), 4: ((1, 2), (0, 2), (0, 1), (1, 3), (2, 3), (0, 3))}


def _cell_keys(cells_g: np.ndarray, n_nodes_g: int, order: int) -> np.ndarray:
    """numbering-independent identity of the DOFs of every cell: vertex DOF -> its global node id, edge DOF (order 2) ->
    n + lo * n + hi of its global end nodes; shape n_cells x n_basis in local-slot order"""
    cg = cells_g.astype(np.int64)
    if order == 1:
        return cg
    cols = [cg]
    for a, b in _EDGE_SLOTS[cells_g.shape[1]]:
        lo, hi = np.minimum(cg[:, a], cg[:, b]), np.maximum(cg[:, a], cg[:, b])
        cols.append((n_nodes_g + lo * n_nodes_g + hi)[:, None])
    return np.concatenate(cols, axis=1)


def boundary_flags(cells, boundary_nodes, keys, order):
    """boundary-DOF flag of every key of the WHOLE mesh, by the reference's rules: node markers; 2-D edge = seen by exactly
    one cell (triangulation.h:177,187); 3-D edge = both end nodes on the boundary (triangulation.h:371)"""
    n = boundary_nodes.shape[0]
    flags = np.zeros(keys.size, dtype=np.uint8)
    is_node = keys < n
    flags[is_node] = boundary_nodes[keys[is_node]] != 0
    if order == 2:
        ek = keys[~is_node] - n
        if cells.shape[1] == 3:
            ck = _cell_keys(cells, n, 2)[:, 3:].ravel()
            uk, cnt = np.unique(ck, return_counts=True)
            flags[~is_node] = cnt[np.searchsorted(uk, keys[~is_node])] == 1
        else:
            flags[~is_node] = (boundary_nodes[ek // n] != 0) & (boundary_nodes[ek % n] != 0)
    return flags


def interface_info(cells, part, n_nodes, world, order=1, boundary_nodes=None, key_sets=None):
    """-> (keys, owner, ifkeys, bflags): the sorted DOF keys of the whole mesh, the lowest rank touching each, the sorted
    keys touched by >= 2 ranks (the interface DOFs, globally indexed by their position in ifkeys), and the whole-mesh
    boundary flag per key (None without boundary_nodes)"""
    per_rank = key_sets if key_sets is not None else rank_key_sets(cells, part, n_nodes, world, order)
    allk = np.concatenate(per_rank)
    rank_of = np.repeat(np.arange(world, dtype=np.int32), [k.size for k in per_rank])
    keys, first, counts = np.unique(allk, return_index=True, return_counts=True)
    bflags = boundary_flags(cells, boundary_nodes, keys, order) if boundary_nodes is not None else None
    return keys, rank_of[first], keys[counts >= 2], bflags


def rank_key_sets(cells, part, n_nodes, world, order=1):
    """the sorted DOF keys each rank touches: one pass over the cells of the whole mesh.  interface_info and peer_lists of one
    partition both need it -- compute it once and hand it to both (key_sets=...)"""
    ck = _cell_keys(cells, n_nodes, order)
    return [np.unique(ck[part == r]) for r in range(world)]


def peer_lists(local_keys, key_sets, rank):
    """neighbour-only exchange lists for fdapde_halo_setup_peers: (peer_rank, peer_off, peer_dof).  local_keys: key of every local DOF
    (interface_maps()['keys']); a peer is a rank sharing at least one key; both ranks of a pair list the shared DOFs by ascending key"""
    order = np.argsort(local_keys)
    sorted_keys = local_keys[order]
    ranks, offs, dofs = [], [0], []
    for q, kq in enumerate(key_sets):
        if q == rank:
            continue
        shared = np.intersect1d(key_sets[rank], kq, assume_unique=True)
        if shared.size == 0:
            continue
        ranks.append(q)
        dofs.append(order[np.searchsorted(sorted_keys, shared)].astype(np.int32))
        offs.append(offs[-1] + shared.size)
    return (np.asarray(ranks, dtype=np.int32), np.asarray(offs, dtype=np.int64),
            np.concatenate(dofs).astype(np.int32) if dofs else np.zeros(0, dtype=np.int32))


def sub_mesh(nodes, cells, boundary, part, rank):
    """the cells of `rank` with their nodes renumbered locally (ascending global id)"""
    my_cells = np.nonzero(part == rank)[0]
    l2g = np.unique(cells[my_cells])                       # local node id -> global node id (sorted)
    local_cells = np.searchsorted(l2g, cells[my_cells]).astype(np.int32)
    return dict(nodes=np.ascontiguousarray(nodes[l2g]), cells=np.ascontiguousarray(local_cells),
                boundary=np.ascontiguousarray(boundary[l2g]), l2g=l2g, cell_ids=my_cells)


def dof_keys(cells_g, table, n_nodes_g, order):
    """key of every DOF of a DOF table (n_cells x n_basis, any numbering) whose cells are given in GLOBAL node ids"""
    ck = _cell_keys(cells_g, n_nodes_g, order)
    keys = np.empty(int(table.max()) + 1, dtype=np.int64)
    keys[table.ravel()] = ck.ravel()
    return keys


def interface_maps(sub, table, info, rank, n_nodes_g, order):
    """interface maps of a sub-mesh for fdapde_halo_setup; `table` = the rank's own DOF table (fdapde_dofs_get)"""
    keys_all, owner, ifkeys, bflags = info
    k = dof_keys(sub["l2g"][sub["cells"]], table, n_nodes_g, order)     # key of each local DOF
    pos = np.searchsorted(ifkeys, k)
    is_if = (pos < ifkeys.size) & (ifkeys[np.minimum(pos, max(ifkeys.size - 1, 0))] == k) if ifkeys.size else np.zeros(k.size, bool)
    local_dof = np.nonzero(is_if)[0].astype(np.int32)
    gpos = np.searchsorted(keys_all, k)
    return dict(n_if_global=int(ifkeys.size), local_dof=local_dof, if_index=pos[local_dof].astype(np.int32),
                owned=(owner[gpos] == rank).astype(np.uint8), keys=k,
                boundary_dofs=bflags[gpos] if bflags is not None else None)   # whole-mesh truth (fdapde_dofs_set_boundary)


def local_problem(nodes, cells, boundary, part, rank, world, info=None):
    """P1 convenience (DOF = node, the DOF table is the local cell table): sub-mesh + interface maps in one dict"""
    info = info if info is not None else interface_info(cells, part, nodes.shape[0], world, 1, boundary)
    sub = sub_mesh(nodes, cells, boundary, part, rank)
    sub.update(interface_maps(sub, sub["cells"], info, rank, nodes.shape[0], 1))
    return sub


# ---- row-distributed form (fdapde_rowdist_setup): every DOF is owned by one rank, whose sub-mesh holds every cell touching it ----------
def node_owners(cells, part, n_nodes, nodes=None):
    """owner of a node = one of the ranks whose cells touch it.  With coordinates: the lowest such rank in the "white" boxes of a coarse
    checkerboard (16 boxes per axis of the bounding box), the highest in the "black" ones -- the nodes of an interface are dealt to both
    sides in PATCHES.  (Dealing them node by node balances just as well but makes every row near an interface read a ghost column: the
    single-launch layout keeps at most half of a workgroup's slots for rows that import, and C3 split in two then no longer fits.)
    Without coordinates: the lowest rank."""
    lo = np.full(n_nodes, np.iinfo(np.int32).max, dtype=np.int32)
    p = np.repeat(part.astype(np.int32), cells.shape[1])
    np.minimum.at(lo, cells.ravel(), p)
    if nodes is None:
        return lo
    hi = np.full(n_nodes, -1, dtype=np.int32)
    np.maximum.at(hi, cells.ravel(), p)
    a, b = nodes.min(axis=0), nodes.max(axis=0)
    box = np.floor((nodes - a) / np.where(b > a, b - a, 1.0) * 16.0).astype(np.int64).sum(axis=1)
    return np.where(box % 2 == 0, lo, hi).astype(np.int32)


def rowdist_sub_mesh(nodes, cells, boundary, owner, rank):
    """the cells touching a node `rank` owns (its cells of the partition that do + one layer of its neighbours' cells), nodes renumbered
    locally (ascending global id)"""
    my_cells = np.nonzero((owner[cells] == rank).any(axis=1))[0]
    l2g = np.unique(cells[my_cells])
    local_cells = np.searchsorted(l2g, cells[my_cells]).astype(np.int32)
    return dict(nodes=np.ascontiguousarray(nodes[l2g]), cells=np.ascontiguousarray(local_cells), boundary=np.ascontiguousarray(boundary[l2g]),
                l2g=l2g, cell_ids=my_cells)


def rowdist_keys_owners(sub, table, owner, n_nodes_g, order):
    """-> (key, owning rank) of every DOF of a rank's DOF table: a vertex DOF belongs to its node's owner, an edge DOF to the owner of its
    end node with the LOWER global id (that rank holds every cell touching the node, hence every cell touching the edge)"""
    keys = dof_keys(sub["l2g"][sub["cells"]], table, n_nodes_g, order)
    is_node = keys < n_nodes_g
    own = np.empty(keys.size, dtype=np.int32)
    own[is_node] = owner[keys[is_node]]
    own[~is_node] = owner[(keys[~is_node] - n_nodes_g) // n_nodes_g]
    return keys, own


def rank_problems_rowdist_p1(nodes, cells, boundary, world, with_node_owners=False):
    """the row-distributed problem of every rank as plain arrays (bench.py's rank 0 ships them): sub-mesh with its ghost layer, global
    node id and owner of every local node (= the P1 DOF keys / owners); with_node_owners: also the owner of every node of the WHOLE mesh,
    from which a rank derives keys and owners of its P2 DOFs once its DOF table exists (rowdist_keys_owners)"""
    part = partition_cells(nodes, cells, world)
    owner = node_owners(cells, part, nodes.shape[0], nodes)
    out = []
    for r in range(world):
        sub = rowdist_sub_mesh(nodes, cells, boundary, owner, r)
        d = dict(nodes=sub["nodes"], cells=sub["cells"], boundary=sub["boundary"], l2g=sub["l2g"], key=sub["l2g"].astype(np.int64),
                 owner=owner[sub["l2g"]].astype(np.int32), n_nodes_total=np.int64(nodes.shape[0]), n_cells_total=np.int64(cells.shape[0]))
        if with_node_owners:
            d["node_owner"] = owner
        out.append(d)
    return out


def rank_problems_p1(nodes, cells, boundary, world):
    """the P1 problem of every rank of an element partition of the whole mesh, as plain arrays (what bench.py's rank 0 ships to the
    others): sub-mesh, interface maps of both exchange forms, ownership.  One pass over the whole mesh, on ONE rank."""
    part = partition_cells(nodes, cells, world)
    n = nodes.shape[0]
    key_sets = rank_key_sets(cells, part, n, world, 1)
    info = interface_info(cells, part, n, world, 1, boundary, key_sets=key_sets)
    out = []
    for r in range(world):
        lp = local_problem(nodes, cells, boundary, part, r, world, info)
        pr, po, pd = peer_lists(lp["keys"], key_sets, r)
        out.append(dict(nodes=lp["nodes"], cells=lp["cells"], boundary=lp["boundary"], l2g=lp["l2g"], owned=lp["owned"],
                        local_dof=lp["local_dof"], if_index=lp["if_index"], n_if_global=np.int64(lp["n_if_global"]),
                        peer_rank=pr, peer_off=po, peer_dof=pd, n_nodes_total=np.int64(n), n_cells_total=np.int64(cells.shape[0])))
    return out


def rank_problem_on_device(capi, device, nodes, cells, boundary, world, rank, form, with_node_owners=False):
    """this rank's problem from the library's device-side partitioner (fdapde_partition_build: csrc/dev_partition.hip): every rank process uploads the
    whole mesh to its OWN device, partitions it there and takes its share -- nothing is shipped between the processes.  Same dict as
    rank_problems_rowdist_p1()[rank] (form "rowdist") / rank_problems_p1()[rank] ("peers", "dense"); -> (dict, seconds spent partitioning)"""
    import time

    t0 = time.perf_counter()
    root = capi.Context(device=device)
    root.mesh_upload(nodes, cells, boundary)
    root.partition_build(world, capi.PARTITION_ROWDIST if form == "rowdist" else capi.PARTITION_ELEMENTS)
    lp = root.partition_get(rank)
    d = dict(nodes=lp["nodes"], cells=lp["cells"], boundary=lp["boundary"], l2g=lp["l2g"], n_nodes_total=np.int64(nodes.shape[0]),
             n_cells_total=np.int64(cells.shape[0]))
    if form == "rowdist":
        d.update(key=lp["l2g"].astype(np.int64), owner=lp["owner"])
        if with_node_owners:
            d["node_owner"] = root.partition_whole()[1]
    else:
        pr, po, pd, owned = root.partition_peers(rank)
        _, _, mask = root.partition_whole()
        multi = (mask & (mask - np.uint64(1))) != 0            # nodes in two or more sub-meshes: the interface, indexed by ascending node id
        ifkeys = np.nonzero(multi)[0]
        is_if = multi[lp["l2g"]]
        local_dof = np.nonzero(is_if)[0].astype(np.int32)
        d.update(owned=owned, peer_rank=pr, peer_off=po, peer_dof=pd, n_if_global=np.int64(ifkeys.size), local_dof=local_dof,
                 if_index=np.searchsorted(ifkeys, lp["l2g"][local_dof]).astype(np.int32))
    root.close()
    return d, time.perf_counter() - t0


class _RcclGroup:
    """barrier / reductions of the rank processes through the library's own RCCL communicator (no second GPU library in the process)"""

    def __init__(self, ctx):
        self.ctx = ctx

    def max(self, values):
        return self.ctx.comm_allreduce(values, "max")

    def sum(self, values):
        return self.ctx.comm_allreduce(values, "sum")

    def barrier(self):
        self.ctx.comm_allreduce([0.0], "sum")


class _GlooGroup:
    def __init__(self):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist

    def max(self, values):
        t = self.torch.tensor(np.asarray(values, dtype=float))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.numpy()

    def sum(self, values):
        t = self.torch.tensor(np.asarray(values, dtype=float))
        self.dist.all_reduce(t)
        return t.numpy()

    def barrier(self):
        self.dist.barrier()


def _comm_setup(capi, ctx, rdzv, rank, world, backend, tag=""):
    """joins the ranks' communicator: the library's own RCCL (id through the rendezvous directory) or, for plumbing checks, host-staged
    transports over gloo.  -> (group for barriers / reductions, transport description)"""
    if backend == "rccl":
        if rank == 0:
            rdzv.put(f"rccl_id{tag}", capi.Context.comm_unique_id())
        ctx.comm_init(world, rank, rdzv.get(f"rccl_id{tag}"))
        return _RcclGroup(ctx), f"RCCL ({capi.Context.comm_library()}), one rank per GPU; no other GPU library in the rank processes"
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        dist.init_process_group("gloo")
    ctx.comm_init_callback(world, rank, lambda arr: dist.all_reduce(torch.from_numpy(arr)))

    def exchange(ranks, off, send, recv):
        reqs, parts = [], []
        for q, r in enumerate(ranks):
            a, b = int(off[q]), int(off[q + 1])
            t_out, t_in = torch.from_numpy(send[a:b].copy()), torch.empty(b - a, dtype=torch.float64)
            reqs += [dist.isend(t_out, int(r)), dist.irecv(t_in, int(r))]
            parts.append((a, b, t_in, t_out))
        for rq in reqs:
            rq.wait()
        for a, b, t_in, _ in parts:
            recv[a:b] = t_in.numpy()

    ctx.comm_set_exchange_callback(exchange)
    return _GlooGroup(), "host-staged gloo (plumbing check, ranks may share a device; never used for reported numbers)"


def canary(capi, rdzv, rank, world, device, backend, share):
    """a two-second row-distributed solve on a small mesh: run by bench.py in a job of its OWN rank processes before the real ranks touch
    the GPUs, so that a fabric on which peer-mapped boards do not work (no hipIpc, stale reads over xGMI -> in-kernel timeouts, or a
    fault that takes the process down) costs the bench its fast path, not its result.  Exit code 0 = every rank solved and agreed."""
    from . import meshgen

    lp, _ = rank_problem_on_device(capi, device, *meshgen.unit_cube(24), world, rank, "rowdist")
    u_exact, f = meshgen.manufactured(3)
    ctx = capi.Context(device=device)
    ctx.mesh_upload(lp["nodes"], lp["cells"], lp["boundary"])
    n_loc = ctx.dofs_build(1)
    grp, _ = _comm_setup(capi, ctx, rdzv, rank, world, backend, ".canary")
    if share > 1:
        ctx.tune("rowdist_share", share)
    ctx.rowdist_setup(lp["key"], lp["owner"])
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    ctx.set_dirichlet(np.zeros(n_loc))
    ctx.init()
    ok = 1.0
    try:
        for _ in range(2):
            info = ctx.solve(rtol=1e-10)
            mine = lp["owner"] == rank
            err = float(np.abs(ctx.solution()[mine] - u_exact(lp["nodes"])[mine]).max()) if mine.any() else 0.0
            if not (info.converged == 1 and info.persistent == 1 and err < 6.0 * (1.0 / 24) ** 2 * 3.15**2):
                ok = 0.0
    except capi.FdapdeError as e:   # (collective by construction: every rank gets the same refusal)
        print(f"canary rank {rank}: {e}", file=sys.stderr)
        ok = 0.0
    bad = grp.max([1.0 - ok])
    ctx.close()
    return 0 if bad[0] == 0.0 else 1


class _FormRefused(Exception):
    """the row-distributed solve declined the system on every rank (collective by construction)"""


def bench_partitioned(capi, rdzv, rank, world, device, args, rtol, backend="rccl", form="peers", share=1):
    """bench.py's N > 1 leg in the chosen form (args.workload "c5": BASELINE config C5 -- 3-D P2 advection-diffusion-reaction, Jacobi-BiCGStab --
    in the row-distributed form only); a row-distributed solve that the library declines at the real size (the canary only proved
    the mechanism on a small mesh) falls back to the neighbour exchange on all ranks together"""
    try:
        return _bench_forms(capi, rdzv, rank, world, device, args, rtol, backend, form, share)
    finally:
        if backend != "rccl":
            import torch.distributed as dist

            if dist.is_initialized():
                dist.destroy_process_group()


def _bench_forms(capi, rdzv, rank, world, device, args, rtol, backend, form, share):
    if getattr(args, "workload", "c3") == "c5":
        return _bench_form(capi, rdzv, rank, world, device, args, rtol, backend, "rowdist", share, tag="")
    fallback = None
    try:
        res = _bench_form(capi, rdzv, rank, world, device, args, rtol, backend, form, share, tag="")
    except _FormRefused as e:
        if rank == 0:
            print(f"bench.py: the row-distributed solve declined this system ({e}); using the RCCL neighbour exchange", file=sys.stderr)
        fallback = f"the row-distributed solve declined this system ({e}); the line is the neighbour exchange's"
        res = _bench_form(capi, rdzv, rank, world, device, args, rtol, backend, "peers", share, tag=".2")
        form = "peers"
    # the OTHER form for the same number of steps, so that one record answers both: north_star's own form -- sub-assembled operators,
    # interface contributions exchanged per operator application over RCCL -- next to the row-distributed launches
    other = None
    if form == "rowdist":
        try:
            o = _bench_form(capi, rdzv, rank, world, device, args, rtol, backend, "peers", share, tag=".peers")
            if o is not None:
                oi = o["info"]
                other = {"ms_per_step": 1e3 * o["elapsed"] / args.steps, "value_dof_per_s": o["total_dofs"] * args.steps / o["elapsed"],
                         "us_per_iteration": 1e3 * o["t_sol"] / max(int(oi.iters), 1), "iterations": int(oi.iters), "relres": float(oi.relres),
                         "t_assemble_ms": o["t_asm"], "t_solve_ms": o["t_sol"], "max_abs_error_vs_analytic": o["err"], "comm_ranks": o["comm_ranks"],
                         "spmv_avg_ms": float(oi.spmv_avg_ms), "parallelism": o["parallelism"], "transport": o["transport"]}
        except Exception as e:   # never let the secondary measurement take the line down (every rank fails the same way or not at all)
            if rank == 0:
                other = {"error": f"{type(e).__name__}: {e}"[:300]}
    if res is not None:
        res["other"] = other if form == "rowdist" else "this line IS the RCCL neighbour exchange"
        res["fallback"] = fallback
    return res


def _agree(rdzv, rank, world, key, error):
    """every rank says whether its LOCAL work up to here succeeded (error = None) -- through the rendezvous directory, which needs no communicator -- and
    every rank learns the same verdict: a failure on ONE rank (out of memory, a refused form, a transport error) then makes ALL ranks leave together
    instead of leaving the others inside the collectives that follow"""
    rdzv.put(f"{key}.{rank}", (b"0" + str(error).encode()[:300]) if error is not None else b"1")
    bad = []
    for r in range(world):
        v = rdzv.get(f"{key}.{r}")
        if v[:1] != b"1":
            bad.append((r, v[1:].decode(errors="replace")))
    if bad:
        raise _RankFailed("; ".join(f"rank {r}: {m}" for r, m in bad))


class _RankFailed(RuntimeError):
    pass


def _bench_form(capi, rdzv, rank, world, device, args, rtol, backend, form, share, tag):
    """bench.py's N > 1 leg: the C3 mesh split over `world` GPUs (strong scaling).  Every rank generates the mesh, partitions it on its own device
    with the library's partitioner and takes its share (rank_problem_on_device).
    form "rowdist": the row-distributed solve (fdapde_rowdist_setup: complete rows per rank, the whole CG as one persistent launch per
    rank, launches exchanging through peer-mapped boards); "peers" / "dense": the element-partitioned solve with an RCCL exchange of the
    interface contributions per operator application.  -> dict (rank 0) / None"""
    import time

    from . import meshgen

    c5 = getattr(args, "workload", "c3") == "c5"
    order = 2 if c5 else 1
    u_exact, f = meshgen.manufactured(3)
    if c5:
        from . import workloads

        u_exact, f = workloads.c5_exact, workloads.c5_forcing
    ctx, n_loc, local_error, lp, t_part, t_gen = None, 0, None, None, 0.0, 0.0
    try:   # this rank's own set-up: nothing collective yet
        # every rank generates the (deterministic) mesh itself and partitions it on its own device with the library (fdapde_partition_build): no rank 0
        # bottleneck, nothing shipped through the rendezvous directory (rounds 1-5: numpy on rank 0, 16.3 s at C3's size)
        t_gen = time.perf_counter()
        whole = meshgen.unit_cube(args.nx, seed=int(getattr(args, "mesh_seed", 12345)))
        t_gen = time.perf_counter() - t_gen
        lp, t_part = rank_problem_on_device(capi, device, *whole, world, rank, form, with_node_owners=c5)
        del whole
        ctx = capi.Context(device=device)
        ctx.mesh_upload(lp["nodes"], lp["cells"], lp["boundary"])
        n_loc = ctx.dofs_build(order)
    except Exception as e:
        local_error = f"{type(e).__name__}: {e}"
    _agree(rdzv, rank, world, "local_setup" + tag, local_error)   # (all ranks leave here together if one of them failed)
    grp, transport = _comm_setup(capi, ctx, rdzv, rank, world, backend, tag)
    comm_ranks = ctx.comm_count()
    coords = lp["nodes"]
    local_error, mine, msg_in = None, None, [0.0, 0.0]
    try:   # the rank's exchange lists: local work again
        if form == "rowdist":
            if share > 1:
                ctx.tune("rowdist_share", share)
            key, own = lp["key"], lp["owner"]
            if order == 2:   # keys / owners of the edge DOFs from the rank's own DOF table (3-D: the library's boundary rule for edges -- both end
                             # nodes on the boundary -- already is the whole mesh's)
                table, _, coords = ctx.dofs_get()
                key, own = rowdist_keys_owners(dict(l2g=lp["l2g"], cells=lp["cells"]), table, lp["node_owner"], int(lp["n_nodes_total"]), 2)
            ctx.rowdist_setup(key, own)
            mine = own == rank
            msg_in = [float(lp["nodes"].shape[0] - int(mine.sum())), float(lp["cells"].shape[0])]
        else:
            # neighbour-only exchange: per-peer packed segments (ncclSend / ncclRecv in one group), then the scalar all-reduce
            pr, po = lp["peer_rank"], lp["peer_off"]
            if form == "peers":
                ctx.halo_setup_peers(pr, po, lp["peer_dof"], lp["owned"])
            else:
                ctx.halo_setup(int(lp["n_if_global"]), lp["local_dof"], lp["if_index"], lp["owned"])
            mine = lp["owned"] != 0
            msg_in = [float(8 * int(po[-1])), float(pr.size)]   # bytes sent per exchange, peers
    except Exception as e:
        local_error = f"{type(e).__name__}: {e}"
    _agree(rdzv, rank, world, "exchange_setup" + tag, local_error)
    msg = grp.max(msg_in)
    local_error = None
    try:
        qn = ctx.quadrature_nodes()
        ctx.set_operator(workloads.c5_operator(capi) if c5 else -capi.laplacian())
        ctx.set_forcing(f(qn))
        ctx.set_dirichlet(np.zeros(n_loc))
        del qn
        if form != "rowdist":
            ctx.solver_prepare(True)   # set-up (untimed): solver layout of this rank's sub-mesh
    except Exception as e:
        local_error = f"{type(e).__name__}: {e}"
    _agree(rdzv, rank, world, "problem_setup" + tag, local_error)   # (before the first collective step)

    def step(time_spmv=0):
        ctx.init()
        return ctx.solve(rtol=rtol, time_spmv=time_spmv)

    if form == "rowdist":   # the first solve builds the layout and maps the boards; a refusal (EUNSUPPORTED) is the same on every rank
        try:
            step()
        except capi.FdapdeError as e:
            if e.status != capi.EUNSUPPORTED:
                raise
            ctx.close()
            raise _FormRefused(str(e)) from None
    for _ in range(args.warmup):
        step()
    grp.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    infos = [step(args.time_spmv) for _ in range(args.steps)]
    ctx.synchronize()
    grp.barrier()
    elapsed = time.perf_counter() - t0
    u = ctx.solution()
    err = float(np.abs(u - u_exact(coords))[mine].max()) if mine.any() else 0.0
    n_own = grp.sum([float(mine.sum())])
    info = infos[-1]
    sizes = ctx.sizes()
    _, alg_bytes = ctx.bench_spmv(reps=1)
    streamed = 0.0 if form == "rowdist" else ctx.solver_layout(True)[2]
    red = grp.max([elapsed, err, np.mean([i.t_assemble_ms for i in infos]), np.mean([i.t_solve_ms for i in infos]), ctx.info().t_setup_ms,
                   float(sizes["nnz"]), alg_bytes, streamed, np.mean([i.spmv_avg_ms for i in infos]), t_part, np.mean([i.launch_ms for i in infos]),
                   np.mean([i.gather_avg_ms for i in infos]), np.mean([i.spmv_mean_ms for i in infos]), np.mean([i.update_avg_ms for i in infos]),
                   float(comm_ranks), -float(comm_ranks)])
    ph_sum = grp.sum([np.mean([i.spmv_avg_ms for i in infos]), np.mean([i.gather_avg_ms for i in infos]), np.mean([i.spmv_mean_ms for i in infos])])
    if backend != "rccl":   # (the gloo group of the plumbing mode outlives this form: bench_partitioned may run a second one)
        import torch.distributed as dist

        dist.barrier()
    ctx.close()
    if rank != 0:
        return None
    for i in infos:   # the slowest rank bounds the figures
        i.spmv_avg_ms, i.launch_ms = float(red[8]), float(red[10])
    if form == "rowdist":
        parallelism = (f"{world} GPUs, row-distributed: every rank owns the DOFs of its Morton chunk of the element partition and assembles their rows "
                       f"completely (sub-mesh = its cells + one layer of its neighbours': <= {int(msg[1])} cells, <= {int(msg[0])} ghost DOFs per rank); the whole "
                       "Jacobi-PCG is ONE persistent launch per rank, the launches of all ranks act as one grid: search-direction entries and dot records "
                       "cross through peer-mapped boards (hipIpc over xGMI), no collective call inside the iteration")
    elif form == "dense":
        parallelism = (f"{world} GPUs, element partition (Morton chunks), {int(lp['n_if_global'])} interface DOFs, single-reduction CG: ONE RCCL "
                       f"all-reduce per iteration (interface entries of A r + r.Ar + r.r: {8 * (int(lp['n_if_global']) + 2)} bytes)")
    else:
        parallelism = (f"{world} GPUs, element partition (Morton chunks), {int(lp['n_if_global'])} interface DOFs; single-reduction CG, per "
                       f"iteration one grouped RCCL send / receive with every neighbour -- the interface entries of A r, <= {int(msg[1])} peers, "
                       f"<= {int(msg[0])} bytes sent per rank -- and one 16-byte all-reduce of (r.Ar, r.r)")
    persistent = int(getattr(info, "persistent", 0))
    phases = None
    if persistent:   # (s_memrealtime stamps inside the persistent launches; the multi-launch path has none)
        phases = {"operator_slowest_workgroup_slowest_rank": 1e3 * float(red[8]), "operator_slowest_workgroup_mean_of_ranks": 1e3 * float(ph_sum[0]) / world,
                  "operator_mean_workgroup_slowest_rank": 1e3 * float(red[12]), "operator_mean_workgroup_mean_of_ranks": 1e3 * float(ph_sum[2]) / world,
                  "allgather_slowest_rank": 1e3 * float(red[11]), "allgather_mean_of_ranks": 1e3 * float(ph_sum[1]) / world,
                  "update_slowest_rank": 1e3 * float(red[13])}
    # every rank must report the same communicator size (max == min)
    comm = int(red[14]) if int(red[14]) == -int(red[15]) else f"INCONSISTENT: between {-int(red[15])} and {int(red[14])}"
    return dict(elapsed=float(red[0]), err=float(red[1]), t_asm=float(red[2]), t_sol=float(red[3]), setup_ms=float(red[4]), info=info, infos=infos,
                alg_bytes=float(red[6]), streamed_bytes=float(red[7]), total_dofs=int(round(n_own[0])), n_cells_total=int(lp["n_cells_total"]),
                parallelism=parallelism, transport=transport, t_partition=float(red[9]), t_meshgen=float(t_gen), form=form, comm_ranks=comm, phases=phases)


# ---- what the first record from N real GPUs should show (the acceptance table) ----------------------------------------------------------
# Single-GPU iteration of the single launch against the rows a GPU holds (3-D P1, measured on MI355X: DESIGN 4.0 table and BENCH_r05)
# interior rows -> us per iteration of the single-launch CG, 3-D P1, one MI355X (tools/iter_by_rows.py, final round-5 build)
_ITER_US_BY_ROWS = ((85e3, 6.6), (205e3, 8.0), (250e3, 9.8), (358e3, 12.4), (705e3, 16.6), (970e3, 20.2), (1.643e6, 29.2))
XGMI_HOP_US = 1.5   # ASSUMED one-way latency of a posted 16-byte store into a peer's board over xGMI (no measurement on this pool; the first record replaces it)


def _interp_iter_us(rows):
    pts = _ITER_US_BY_ROWS
    if rows <= pts[0][0]:
        return pts[0][1] * max(rows / pts[0][0], 0.6)   # (below 85 k rows the iteration is the hand-off latencies, not the rows)
    for (r0, t0), (r1, t1) in zip(pts, pts[1:]):
        if rows <= r1:
            return t0 + (t1 - t0) * (rows - r0) / (r1 - r0)
    return pts[-1][1] * rows / pts[-1][0]


def predict_c3(world, form, interior_rows=1643032, iterations=505, n_dofs=1728000, init_ms_one_gpu=0.92):
    """Predicted us per iteration and DOF/s of C3 on `world` MI355X for the two exchange forms, from single-GPU phase measurements + the assumed
    xGMI hop: what the first multi-GPU record is judged against (a measurement beyond `wrong_above_us` means the form, or the canary's choice of
    it, does not work as designed on that fabric)."""
    rows = interior_rows / world
    if form == "rowdist":
        base = _interp_iter_us(rows)   # the rank's own rows as one launch on all 256 CUs (its 256-record dot gather included)
        g = 256 * world
        gather_extra = (0.0058 * (g - 256) + XGMI_HOP_US) if g <= 1024 else (2.5 + XGMI_HOP_US)   # one hop: a sweep of g records; two levels: + one rank-record hop
        imports = 0.5                  # entries from other ranks arrive while the import-free passes run; what is not hidden
        us = base + gather_extra + imports
        per_solve_ms = 0.10            # ghost-scale exchange + one 16-byte all-reduce before the launch
    else:                              # element-partitioned: SpMV + pack / sum launches + one grouped send / recv + one 16-byte all-reduce per iteration
        us = max(21.5, 68.0 * rows / 1.643e6 + 6.0) + 11.0 + 2 * (8.0 + XGMI_HOP_US * (2 if world > 2 else 1)) + 8.0
        per_solve_ms = 0.3
    step_ms = iterations * us * 1e-3 + init_ms_one_gpu / world + 0.1 + per_solve_ms
    return {"form": form, "world": world, "rows_per_rank": int(rows), "us_per_iteration": round(us, 1), "ms_per_step": round(step_ms, 2),
            "dof_per_s": round(n_dofs / (step_ms * 1e-3)), "wrong_above_us": round(2.0 * us, 1),
            "basis": "single-GPU iteration at the rank's row count (measured) + gather over all ranks' workgroups + ASSUMED xGMI hop of %.1f us" % XGMI_HOP_US}


def predict_weak(world, form, nx, iterations=None):
    """The weak-scaling leg (bench.py --scaling weak: (nx + 1)^3 ~ world x 120^3 nodes, ~1.73 M rows per GPU): the same model as predict_c3 at a constant
    row count per rank; Jacobi-PCG iterations grow with the mesh (~ nx: 503 at nx = 119).  A rank of 1.7 M rows + ghost columns is at the edge of what
    the row-distributed single launch takes (half of a workgroup's slots are kept for rows that import): where the library declines it the record is
    the RCCL neighbour exchange's -- both predictions are given."""
    n_dofs = (nx + 1) ** 3
    interior = (nx - 1) ** 3
    its = iterations if iterations is not None else int(round(503 * nx / 119.0))
    p = predict_c3(world, form, interior_rows=interior, iterations=its, n_dofs=n_dofs, init_ms_one_gpu=0.92 * n_dofs / 1728000.0)
    p["scaling"] = "weak"
    return p
