"""Worker of tests/test_gpu_dist.py: one rank of a 2- or 3-rank job in which ALL ranks share GPU 0.  The device-side
distributed path (sub-assembly, interface pack / all-reduce / unpack, owner-masked dots, stop decisions) is the product's;
only the transport (neighbour exchange + scalar all-reduce, or the dense interface all-reduce) is swapped for host-staged
torch.distributed/gloo callbacks, because RCCL refuses two ranks on one device.  Results are compared with a single-domain run of the whole mesh on the same GPU.
cases: p1 | p2 | sq2 (2-D P2: interface edges must not become Dirichlet) | adr1 | adr2 (BiCGStab) | parab | handle
transport (argv[6]): "shared" (default: all ranks on GPU 0, host-staged gloo all-reduce) | "rccl" (rank r on GPU r, the library's
own RCCL communicator over xGMI -- the product configuration; needs >= world GPUs)
exchange (argv[7]): "peers" (default: fdapde_halo_setup_peers, per-peer packed send / receive) | "dense" (fdapde_halo_setup)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


ORACLE_NX = 20   # cases up to this size are compared with the CPU oracle's direct solve of the WHOLE mesh as well (multi-rank parity
                 # pinned directly, not only through the single-domain HIP solve)


def oracle_solution(nodes, cells, bnd, order, op_name, bvec, fq, g):
    """oracle.pde_init_solve on the whole mesh: the reference algorithm restated on the CPU (test infrastructure, tests only)"""
    from oracle import oracle as o

    m = o.Mesh(np.ascontiguousarray(nodes), np.ascontiguousarray(cells, dtype=np.int32), np.ascontiguousarray(bnd, dtype=np.uint8))
    op = {"lap": lambda: -o.laplacian(),
          "lap+r": lambda: -o.laplacian() + o.reaction(0.5) + o.reaction(0.0),
          "adr": lambda: -o.laplacian() + o.advection(np.asarray(bvec, dtype=float)) + o.reaction(1.0),
          "adr+r": lambda: -o.laplacian() + o.reaction(0.5) + o.advection(np.asarray(bvec, dtype=float))}[op_name]()
    return o.pde_init_solve(m, order, op, forcing_q=fq, dirichlet=g).solution


def main():
    rank, world, port, nx = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    case = sys.argv[5] if len(sys.argv) > 5 else "p1"
    two_level = case.endswith(":2level")   # row-distributed form: force the two-level dot gather (the automatic choice for > 1024 workgroups)
    case = case.split(":")[0]
    transport = sys.argv[6] if len(sys.argv) > 6 else "shared"
    exchange_mode = sys.argv[7] if len(sys.argv) > 7 else "peers"   # "peers": neighbour-only exchange | "dense": interface all-reduce
    dev_id = rank if transport == "rccl" else 0
    import torch
    import torch.distributed as dist

    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi, meshgen
    from fdapde_core_amd import dist as fdist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    order = 2 if case in ("p2", "sq2", "adr2") else 1
    nodes, cells, bnd = meshgen.unit_square(nx) if case == "sq2" else meshgen.unit_cube(nx)
    N = nodes.shape[1]
    n_g = nodes.shape[0]
    u_exact, f = meshgen.manufactured(N)
    g_fn = lambda x: 0.3 * x[:, 0] - 0.2 * x[:, -1]
    bvec = [1.0, 0.5, 0.25][:N]
    # "indef": -Lap u - k^2 u with a handful of negative eigenvalues -- symmetric, CG breaks down (p.Ap <= 0), the open method must hand over to
    # BiCGStab on ALL ranks together; "pe30": cell Peclet number 30 -- BiCGStab breaks down and restarts, collectively
    pe_b = (2.0 * 30.0 * nx / np.linalg.norm(bvec)) * np.asarray(bvec)
    mkop = (lambda: -capi.laplacian() + capi.advection(bvec) + capi.reaction(1.0)) if case.startswith("adr") else \
           (lambda: capi.dt() - capi.laplacian()) if case == "parab" else \
           (lambda: -capi.laplacian() - capi.reaction(60.0 if N == 3 else 45.0)) if case == "indef" else \
           (lambda: -capi.laplacian() + capi.advection(pe_b)) if case == "pe30" else (lambda: -capi.laplacian())
    part = fdist.partition_cells(nodes, cells, world)
    if exchange_mode == "rowdist":
        return rowdist_case(rank, world, case, order, nodes, cells, bnd, part, capi, fdist, dist, torch, u_exact, f, g_fn, transport, two_level, nx)
    info_if = fdist.interface_info(cells, part, n_g, world, order, bnd)
    sub = fdist.sub_mesh(nodes, cells, bnd, part, rank)

    def allreduce(arr):
        t = torch.from_numpy(arr)
        dist.all_reduce(t)

    ctx = capi.Context(device=dev_id)
    ctx.mesh_upload(sub["nodes"], sub["cells"], sub["boundary"])
    n_loc = ctx.dofs_build(order)
    table, _, lcoords = ctx.dofs_get()
    maps = fdist.interface_maps(sub, table, info_if, rank, n_g, order)
    ctx.dofs_set_boundary(maps["boundary_dofs"])
    if transport == "rccl":   # the 128-byte RCCL id travels over the gloo group; the data path is RCCL only
        uid = [capi.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])
    else:
        ctx.comm_init_callback(world, rank, allreduce)

    def exchange(ranks, off, send, recv):   # host-staged neighbour exchange over gloo: one isend + one irecv per peer
        reqs, parts = [], []
        for q, r in enumerate(ranks):
            a, b = int(off[q]), int(off[q + 1])
            t_out, t_in = torch.from_numpy(send[a:b].copy()), torch.empty(b - a, dtype=torch.float64)
            reqs.append(dist.isend(t_out, int(r)))
            reqs.append(dist.irecv(t_in, int(r)))
            parts.append((a, b, t_in, t_out))
        for rq in reqs:
            rq.wait()
        for a, b, t_in, _ in parts:
            recv[a:b] = t_in.numpy()

    if exchange_mode == "peers":
        pr, po, pd = fdist.peer_lists(maps["keys"], fdist.rank_key_sets(cells, part, n_g, world, order), rank)
        assert pr.size >= 1 and (world > 2 or pr.size == 1)
        if transport != "rccl":
            ctx.comm_set_exchange_callback(exchange)
        ctx.halo_setup_peers(pr, po, pd, maps["owned"])
    else:
        ctx.halo_setup(maps["n_if_global"], maps["local_dof"], maps["if_index"], maps["owned"])
    # single-domain context of the whole mesh (every rank builds it; same GPU)
    ref = capi.Context(device=dev_id)
    ref.mesh_upload(nodes, cells, bnd)
    ref.dofs_build(order)
    gtable, gbnd, gcoords = ref.dofs_get()
    gk = fdist.dof_keys(cells, gtable, n_g, order)
    l2g = np.argsort(gk)[np.searchsorted(np.sort(gk), maps["keys"])]       # local DOF -> whole-mesh DOF id
    assert np.array_equal(gbnd[l2g], maps["boundary_dofs"]) and np.abs(gcoords[l2g] - lcoords).max() <= 1e-15

    ctx.set_operator(mkop())
    ref.set_operator(mkop())
    if case == "parab":
        times = np.linspace(0.0, 0.5, 6)
        ut = lambda x, t: u_exact(x) * np.exp(-t)
        ft = lambda x, t: (f(x) - u_exact(x)) * np.exp(-t)
        out = []
        for c_, co in ((ctx, lcoords), (ref, gcoords)):
            qn = c_.quadrature_nodes()
            c_.set_forcing(np.stack([ft(qn, t) for t in times], axis=1))
            c_.init()
            G = np.stack([ut(co, t) for t in times], axis=1)
            sol, inf = c_.solve_parabolic(times, G[:, 0], G, rtol=1e-11)
            assert inf.converged == 1
            out.append(sol)
        err = max(np.linalg.norm(out[0][:, j] - out[1][l2g, j]) / np.linalg.norm(out[1][:, j]) for j in range(1, times.size))
        assert err < 1e-8, err
        msg = f"steps {times.size - 1}"
    elif case == "handle":
        for c_ in (ctx, ref):
            c_.set_forcing(f(c_.quadrature_nodes()))
            c_.init()
            c_.lin_compute(capi.MAT_MASS, symmetric=True)
        z = np.cos(3.0 * gcoords[:, 0]) + gcoords[:, 1] * gcoords[:, 2]
        B = np.stack([ctx.spmv(capi.MAT_MASS, z[l2g]), ctx.spmv(capi.MAT_MASS, (z * z)[l2g])], axis=1)   # sub-assembled M z
        X, inf = ctx.lin_solve(B, rtol=1e-12)
        assert inf.converged == 1
        err = max(np.linalg.norm(X[:, 0] - z[l2g]) / np.linalg.norm(z[l2g]), np.linalg.norm(X[:, 1] - (z * z)[l2g]) / np.linalg.norm((z * z)[l2g]))
        assert err < 1e-8, err
        msg = "handle"
    else:
        res = []
        for c_, co in ((ctx, lcoords), (ref, gcoords)):
            c_.set_forcing(f(c_.quadrature_nodes()))
            c_.set_dirichlet(g_fn(co))
            c_.init()
            res.append((c_.solve(rtol=1e-11), c_.solution()))
        (info, u), (rinfo, uref) = res
        err = np.linalg.norm(u - uref[l2g]) / np.linalg.norm(uref)
        assert info.converged == 1 and err < 1e-8, (info.converged, err)
        if case in ("indef", "pe30"):   # the single-domain context took the same turn (CG -> BiCGStab / BiCGStab restarts); paths differ with rounding
            assert info.method_used == capi.SOLVER_BICGSTAB and rinfo.converged == 1, (info.method_used, rinfo.converged)
        elif case.startswith("adr"):
            assert info.method_used == capi.SOLVER_BICGSTAB and info.iters <= 1.3 * rinfo.iters + 5, (info.iters, rinfo.iters)
        else:
            assert abs(info.iters - rinfo.iters) <= max(2, rinfo.iters // 50), (info.iters, rinfo.iters)   # same Krylov iteration up to rounding
        msg = f"iters {info.iters} (single domain {rinfo.iters})"
        if nx <= ORACLE_NX and case not in ("indef", "pe30"):   # ... and directly against the oracle's direct solve of the whole mesh (same numbering: l2g)
            uo = oracle_solution(nodes, cells, bnd, order, "adr" if case.startswith("adr") else "lap", bvec, f(ref.quadrature_nodes()), g_fn(gcoords))
            err_o = np.linalg.norm(u - uo[l2g]) / np.linalg.norm(uo)
            assert err_o < 1e-8, err_o
            msg += f"  vs oracle {err_o:.1e}"
    dist.barrier()   # nobody tears its communicator down while a peer is still inside a collective
    print(f"rank {rank}: ok  case {case}  local dofs {n_loc}  interface {maps['local_dof'].size}/{maps['n_if_global']}  {msg}  err {err:.2e}")
    dist.destroy_process_group()


def rowdist_case(rank, world, case, order, nodes, cells, bnd, part, capi, fdist, dist, torch, u_exact, f, g_fn, transport="shared", two_level=False, nx=0):
    """row-distributed form (fdapde_rowdist_setup): one persistent launch per rank, all ranks' launches acting as one grid through
    peer-mapped boards -- here all on GPU 0, each rank with an equal share of the CUs, boards mapped across the processes by hipIpc"""
    n_g = nodes.shape[0]
    owner = fdist.node_owners(cells, part, n_g, nodes)
    sub = fdist.rowdist_sub_mesh(nodes, cells, bnd, owner, rank)
    dev_id = rank if transport == "rccl" else 0   # "rccl": one rank per GPU, boards mapped across the devices (xGMI) -- the product configuration
    ctx = capi.Context(device=dev_id)
    ctx.mesh_upload(sub["nodes"], sub["cells"], sub["boundary"])
    n_loc = ctx.dofs_build(order)
    table, _, lcoords = ctx.dofs_get()
    keys, own = fdist.rowdist_keys_owners(sub, table, owner, n_g, order)
    if order == 2:   # whole-mesh boundary flags (a 2-D edge on the rim of the sub-mesh is seen by one LOCAL cell only)
        allk = np.unique(fdist._cell_keys(cells, n_g, 2))
        flags = fdist.boundary_flags(cells, bnd, allk, 2)
        ctx.dofs_set_boundary(flags[np.searchsorted(allk, keys)])

    def allreduce(arr):
        dist.all_reduce(torch.from_numpy(arr))

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

    if transport == "rccl":
        uid = [capi.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])
    else:
        ctx.comm_init_callback(world, rank, allreduce)
        ctx.comm_set_exchange_callback(exchange)
        ctx.tune("rowdist_share", world)
    if two_level:
        ctx.tune("rowdist_flat_gather", 0)
    ctx.rowdist_setup(keys, own)
    ref = capi.Context(device=dev_id)
    ref.mesh_upload(nodes, cells, bnd)
    ref.dofs_build(order)
    gtable, gbnd, gcoords = ref.dofs_get()
    gk = fdist.dof_keys(cells, gtable, n_g, order)
    l2g = np.argsort(gk)[np.searchsorted(np.sort(gk), keys)]
    mine = own == rank
    res = []
    N = nodes.shape[1]
    adr = case.startswith("adr")
    if case == "parab":   # implicit Euler through the row-distributed launches: warm starts need the ghost entries of every step's solution
        times = np.linspace(0.0, 0.5, 6)
        ut = lambda x, t: u_exact(x) * np.exp(-t)
        ft = lambda x, t: (f(x) - u_exact(x)) * np.exp(-t)
        out = []
        for c_, co in ((ctx, lcoords), (ref, gcoords)):
            qn = c_.quadrature_nodes()
            c_.set_operator(capi.dt() - capi.laplacian())
            c_.set_forcing(np.stack([ft(qn, t) for t in times], axis=1))
            c_.init()
            G = np.stack([ut(co, t) for t in times], axis=1)
            sol, inf = c_.solve_parabolic(times, G[:, 0], G, rtol=1e-11)
            assert inf.converged == 1
            if c_ is ctx:   # every step of the rank's context ran as the row-distributed SINGLE launch (warm start through the ghost entries), not
                assert inf.persistent == 1, "parabolic step fell back from the row-distributed launch"   # the element-partitioned multi-launch form
            out.append(sol)
        e2 = np.array([sum(np.sum((out[0][mine, j] - out[1][l2g, j][mine]) ** 2) for j in range(1, times.size))])
        allreduce(e2)
        err = float(np.sqrt(e2[0])) / np.linalg.norm(out[1][:, 1:])
        assert err < 1e-9, err
        oracle_msg = ""
        if nx <= 12:   # ... and against the oracle's LU stepping of the whole mesh (fem_linear_parabolic_solver.h:37-72 restated), over the owned DOFs
            from oracle import oracle as o

            m = o.Mesh(np.ascontiguousarray(nodes), np.ascontiguousarray(cells, dtype=np.int32), np.ascontiguousarray(bnd, dtype=np.uint8))
            qn_g = ref.quadrature_nodes()
            G_g = np.stack([ut(gcoords, t) for t in times], axis=1)
            uo, _ = o.pde_parabolic_solve(m, order, o.dt() - o.laplacian(), times, np.stack([ft(qn_g, t) for t in times], axis=1), G_g, G_g[:, 0])
            eo = np.array([sum(np.sum((out[0][mine, j] - uo[l2g, j][mine]) ** 2) for j in range(1, times.size))])
            allreduce(eo)
            err_o = float(np.sqrt(eo[0])) / np.linalg.norm(uo[:, 1:])
            assert err_o < 1e-8, err_o
            oracle_msg = f"  vs oracle LU stepping {err_o:.1e}"
        dist.barrier()
        print(f"rank {rank}: ok  case parab rowdist  local dofs {n_loc}  steps {times.size - 1}  err {err:.2e}{oracle_msg}")
        dist.destroy_process_group()
        return
    if case == "stall":   # one workgroup of ONE rank stops taking part: every rank must give up together (no hang) and say so the same way;
                          # the next solve -- epochs advanced past anything the failed launches can have written -- works again
        for c_, co in ((ctx, lcoords), (ref, gcoords)):
            c_.set_operator(-capi.laplacian() + capi.reaction(0.5))
            c_.set_forcing(f(c_.quadrature_nodes()))
            c_.set_dirichlet(g_fn(co))
            c_.init()
        rinfo, uref = ref.solve(rtol=1e-11), ref.solution()
        ctx.tune("persist_timeout_us", 3000)
        ctx.tune("rowdist_timeout_first_ms", 200)
        if rank == world - 1:
            ctx.tune("persist_debug_stall", 3)
        refused = False
        try:
            ctx.solve(rtol=1e-11)
        except capi.FdapdeError as e:
            refused = e.status == capi.EUNSUPPORTED and "hand-off" in str(e)
        assert refused, "a rank whose peer dropped out must report the collective refusal"
        ctx.tune("persist_debug_stall", 0)
        info = ctx.solve(rtol=1e-11)
        u = ctx.solution()
        e2 = np.array([np.sum((u[mine] - uref[l2g][mine]) ** 2)])
        allreduce(e2)
        err = float(np.sqrt(e2[0])) / np.linalg.norm(uref)
        assert info.converged == 1 and info.persistent == 1 and err < 1e-9, (info.converged, err)
        dist.barrier()
        print(f"rank {rank}: ok  case stall rowdist  local dofs {n_loc}  err {err:.2e}")
        dist.destroy_process_group()
        return
    if case.startswith("fail"):   # a HARD local failure on ONE rank during the collective set-up (stage = digit: 0 host mirrors, 1 / 2 key
                                  # lists, 3 boards, 4 uploads): nobody may be left waiting in an exchange -- every rank's solve returns, the
                                  # failing rank with its own error, the others with the collective refusal; and again on the next solve
        stage = int(case[4:])
        ctx.set_operator(-capi.laplacian() + capi.reaction(0.5))
        ctx.set_forcing(f(ctx.quadrature_nodes()))
        ctx.set_dirichlet(g_fn(lcoords))
        ctx.init()
        os.environ["FDAPDE_ROWDIST_FAIL_RANK"], os.environ["FDAPDE_ROWDIST_FAIL_AT"] = str(world - 1), str(stage)
        statuses = []
        for _ in range(2):
            try:
                ctx.solve(rtol=1e-11)
                statuses.append(capi.OK)
            except capi.FdapdeError as e:
                statuses.append(e.status)
        mine_failed = rank == world - 1 and stage != 3   # (stage 3, boards: a property of the fabric -> the soft refusal everywhere)
        assert statuses[0] == (capi.EHIP if mine_failed else capi.EUNSUPPORTED), statuses
        assert statuses[1] == capi.EUNSUPPORTED, statuses   # the layout stays refused: no second collective attempt, same answer everywhere
        dist.barrier()
        print(f"rank {rank}: ok  case {case} rowdist  statuses {statuses}")
        dist.destroy_process_group()
        return
    if case == "handle":   # factor-once handle on the mass matrix: right-hand sides complete at the owned DOFs
        for c_ in (ctx, ref):
            c_.set_operator(-capi.laplacian())
            c_.set_forcing(f(c_.quadrature_nodes()))
            c_.init()
            c_.lin_compute(capi.MAT_MASS, symmetric=True)
        z = np.cos(3.0 * gcoords[:, 0]) + gcoords[:, 1] * gcoords[:, -1]
        Bref = np.stack([ref.spmv(capi.MAT_MASS, z), ref.spmv(capi.MAT_MASS, z * z)], axis=1)   # M z of the whole mesh
        X, inf = ctx.lin_solve(Bref[l2g], rtol=1e-12)
        assert inf.converged == 1 and inf.persistent == 1
        e2 = np.array([np.sum((X[mine, 0] - z[l2g][mine]) ** 2) + np.sum((X[mine, 1] - (z * z)[l2g][mine]) ** 2)])
        allreduce(e2)
        err = float(np.sqrt(e2[0])) / np.linalg.norm(z)
        assert err < 1e-8, err
        dist.barrier()
        print(f"rank {rank}: ok  case handle rowdist  local dofs {n_loc}  err {err:.2e}")
        dist.destroy_process_group()
        return
    hard = case in ("indef", "pe30")
    bdir = np.asarray([1.0, 0.5, 0.25][:N])
    for c_, co in ((ctx, lcoords), (ref, gcoords)):
        if case == "indef":
            c_.set_operator(-capi.laplacian() - capi.reaction(60.0 if N == 3 else 45.0))
        elif case == "pe30":
            c_.set_operator(-capi.laplacian() + capi.advection((2.0 * 30.0 * nx / np.linalg.norm(bdir)) * bdir))
        else:
            c_.set_operator(-capi.laplacian() + capi.reaction(0.5) + (capi.advection([1.0, 0.5, 0.25][:N]) if adr else capi.reaction(0.0)))
        c_.set_forcing(f(c_.quadrature_nodes()))
        c_.set_dirichlet(g_fn(co))
        c_.init()
        res.append((c_.solve(rtol=1e-11), c_.solution()))
    (info, u), (rinfo, uref) = res
    assert info.converged == 1 and info.persistent == 1, (info.converged, info.persistent)
    assert info.method_used == (capi.SOLVER_BICGSTAB if adr or hard else capi.SOLVER_CG_FUSED)
    err2 = np.array([np.sum((u[mine] - uref[l2g][mine]) ** 2), float(mine.sum())])
    allreduce(err2)
    err = float(np.sqrt(err2[0])) / np.linalg.norm(uref)
    assert int(err2[1]) == gk.size, "every DOF of the whole mesh is owned exactly once"
    assert err < (1e-7 if hard else 1e-8 if adr else 1e-9), err
    if hard:
        assert rinfo.converged == 1
    elif adr:
        assert info.iters <= 1.3 * rinfo.iters + 5, (info.iters, rinfo.iters)
    else:
        assert abs(info.iters - rinfo.iters) <= max(1, rinfo.iters // 100), (info.iters, rinfo.iters)
    oracle_msg = ""
    if 0 < nx <= ORACLE_NX and not hard:   # directly against the oracle's direct solve of the whole mesh, over the DOFs this rank owns
        uo = oracle_solution(nodes, cells, bnd, order, "adr+r" if adr else "lap+r", [1.0, 0.5, 0.25][:N], f(ref.quadrature_nodes()), g_fn(gcoords))
        eo = np.array([np.sum((u[mine] - uo[l2g][mine]) ** 2)])
        allreduce(eo)
        err_o = float(np.sqrt(eo[0])) / np.linalg.norm(uo)
        assert err_o < 1e-8, err_o
        oracle_msg = f"  vs oracle {err_o:.1e}"
    # second solve on the same context: epochs advance, boards are not cleared; identical bits
    info2 = ctx.solve(rtol=1e-11)
    assert info2.converged == 1 and (hard or (info2.iters == info.iters and np.array_equal(ctx.solution()[mine], u[mine])))
    dist.barrier()
    print(f"rank {rank}: ok  case {case} rowdist  local dofs {n_loc} (owned {int(mine.sum())})  iters {info.iters} (single domain {rinfo.iters})  "
          f"err {err:.2e}  launch {info.launch_ms:.3f} ms{oracle_msg}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
