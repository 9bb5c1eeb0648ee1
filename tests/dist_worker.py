"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job (CPU).

Checks the host logic of the element-partitioned path (fdapde-core_amd/dist.py: partition, sub-meshes, interface maps,
ownership) by running the SAME distributed Jacobi-PCG the device code runs (csrc/capi.hip, `dist` branch of fdapde_solve:
one all-reduce of packed interface entries + p.Ap per iteration, one scalar all-reduce) in numpy on top of the oracle's
local operators, with torch.distributed (gloo) as the transport, and comparing with the single-domain oracle solve.
The oracle is used here as the checker and local-operator provider of a TEST; nothing in the product imports it.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    case, order, opk = (case.split(":") + ["1", "lap"])[:3]     # mesh[:order[:lap|adr]]
    order = int(order)
    import torch
    import torch.distributed as dist

    from fdapde_loader import load_package
    from oracle import oracle as o

    load_package()
    from fdapde_core_amd import dist as fdist
    from fdapde_core_amd import meshgen

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    if case == "cube":
        nodes, cells, bnd = meshgen.unit_cube(7)
    elif case == "square":
        nodes, cells, bnd = meshgen.unit_square(24)
    else:
        m = o.load_mesh(os.path.join(ROOT, "tests", "golden", "mesh", case))
        nodes, cells, bnd = m.nodes, m.cells, m.boundary
    N = nodes.shape[1]
    u_exact, f = meshgen.manufactured(N)
    g_fn = lambda x: 0.3 * x[:, 0] - 0.2 * x[:, -1]      # non-trivial Dirichlet data

    part = fdist.partition_cells(nodes, cells, world)
    assert np.array_equal(np.bincount(part, minlength=world) > 0, np.ones(world, bool))
    info = fdist.interface_info(cells, part, nodes.shape[0], world, order, bnd)
    lp = fdist.sub_mesh(nodes, cells, bnd, part, rank)
    lm = o.Mesh(lp["nodes"], lp["cells"], lp["boundary"])
    dofs, lb, nd, _ = o.enumerate_dofs(lm, order)          # the reference's enumeration on the sub-mesh
    lp.update(fdist.interface_maps(lp, dofs, info, rank, nodes.shape[0], order))
    if order == 1:                                           # the P1 convenience wrapper gives the same maps
        lp1 = fdist.local_problem(nodes, cells, bnd, part, rank, world, info)
        assert all(np.array_equal(lp1[k], lp[k]) for k in ("local_dof", "if_index", "owned", "cells", "l2g"))
        assert np.array_equal(lp["keys"], lp["l2g"])
    # ---- invariants of the maps
    keys_all, owner, ifkeys, bflags = info
    gd, gb, _, _ = o.enumerate_dofs(o.Mesh(nodes, cells, bnd), order)
    assert np.array_equal(bflags[np.searchsorted(keys_all, fdist.dof_keys(cells, gd, nodes.shape[0], order))], gb)   # whole-mesh boundary DOFs
    lb = lp["boundary_dofs"]                                # NOT the sub-mesh's own flags (interface edges look like boundary there)
    assert np.all(lp["keys"][lp["local_dof"]] == ifkeys[lp["if_index"]])
    own_count = torch.zeros(keys_all.size, dtype=torch.int64)
    own_count[torch.from_numpy(np.searchsorted(keys_all, lp["keys"][lp["owned"] == 1]))] += 1
    dist.all_reduce(own_count)
    assert int(own_count.min()) == 1 and int(own_count.max()) == 1      # every global DOF owned exactly once
    ncell = torch.tensor([lp["cells"].shape[0]])
    dist.all_reduce(ncell)
    assert int(ncell) == cells.shape[0]

    # ---- neighbour-only exchange lists (fdapde_halo_setup_peers): symmetric per pair, and the rank-ordered sum over real isend / irecv
    #      equals the dense interface all-reduce, with identical bits on every rank sharing a DOF
    pr, po, pd = fdist.peer_lists(lp["keys"], fdist.rank_key_sets(cells, part, nodes.shape[0], world, order), rank)
    assert pr.size >= 1 and np.all(np.diff(pr) > 0) and rank not in pr and po[0] == 0 and po[-1] == pd.size
    assert set(pd.tolist()) == set(lp["local_dof"].tolist())          # every interface DOF is shared with somebody, and nothing else is
    v_loc = np.random.default_rng(100 + rank).standard_normal(lp["keys"].size)
    send, recv = v_loc[pd].copy(), np.empty(pd.size)
    reqs, parts = [], []
    for q, r in enumerate(pr):
        a, b_ = int(po[q]), int(po[q + 1])
        kq = torch.from_numpy(lp["keys"][pd[a:b_]].copy())               # the keys travel too: both sides must list the same DOFs in the same order
        t_out, t_in, k_in = torch.from_numpy(send[a:b_].copy()), torch.empty(b_ - a, dtype=torch.float64), torch.empty(b_ - a, dtype=torch.int64)
        reqs += [dist.isend(t_out, int(r), tag=1), dist.irecv(t_in, int(r), tag=1), dist.isend(kq, int(r), tag=2), dist.irecv(k_in, int(r), tag=2)]
        parts.append((a, b_, t_in, k_in, kq, t_out))
    for rq in reqs:
        rq.wait()
    for a, b_, t_in, k_in, kq, _ in parts:
        assert torch.equal(k_in, kq)
        recv[a:b_] = t_in.numpy()
    summed = {}
    for d in np.unique(pd):                                              # contributions in ascending rank order, own among them
        terms = sorted([(int(pr[np.searchsorted(po, j, side="right") - 1]), float(recv[j])) for j in np.nonzero(pd == d)[0]] + [(rank, float(v_loc[d]))])
        acc = 0.0
        for _, val in terms:
            acc += val
        summed[int(d)] = acc
    dense = torch.zeros(lp["n_if_global"], dtype=torch.float64)
    dense[torch.from_numpy(lp["if_index"].astype(np.int64))] = torch.from_numpy(v_loc[lp["local_dof"]])
    dist.all_reduce(dense)
    mine = np.array([summed[int(d)] for d in lp["local_dof"]])
    assert np.abs(mine - dense.numpy()[lp["if_index"]]).max() <= 1e-14
    hi = torch.full((lp["n_if_global"],), -np.inf, dtype=torch.float64)
    lo = torch.full((lp["n_if_global"],), np.inf, dtype=torch.float64)
    hi[torch.from_numpy(lp["if_index"].astype(np.int64))] = torch.from_numpy(mine)
    lo[torch.from_numpy(lp["if_index"].astype(np.int64))] = torch.from_numpy(mine)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    assert torch.equal(hi, lo), "every rank sharing a DOF must hold the same bits"

    # ---- local sub-assembled operators (oracle on the sub-mesh of this rank's cells)
    bvec = np.array([1.0, 0.5, 0.25])[:N]
    mkop = (lambda: -o.laplacian()) if opk == "lap" else (lambda: -o.laplacian() + o.advection(bvec) + o.reaction(1.0))
    A = o.assemble_operator(lm, order, dofs, nd, mkop())
    b = o.assemble_forcing(lm, order, dofs, nd, f(o.quadrature_nodes(lm, order)))
    gm = o.Mesh(nodes, cells, bnd)
    gdofs, _, gnd, _ = o.enumerate_dofs(gm, order)
    gcoords = o.dofs_coords(gm, order, gdofs, gnd)
    gk = fdist.dof_keys(cells, gdofs, nodes.shape[0], order)
    l2g_dof = np.argsort(gk)[np.searchsorted(np.sort(gk), lp["keys"])]   # local DOF -> DOF id of the whole-mesh enumeration
    g = g_fn(gcoords[l2g_dof])
    n_if, ld, ix, owned = lp["n_if_global"], lp["local_dof"], lp["if_index"], lp["owned"].astype(bool)

    def halo_sum(v, extra=()):
        buf = torch.zeros(n_if + len(extra), dtype=torch.float64)
        buf[torch.from_numpy(ix.astype(np.int64))] = torch.from_numpy(v[ld])
        for k, e in enumerate(extra):
            buf[n_if + k] = e
        dist.all_reduce(buf)
        v[ld] = buf.numpy()[ix]
        return buf.numpy()[n_if:]

    def gsum(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t)

    diag = A.to_scipy().diagonal().copy()
    halo_sum(diag)
    isb = lb.astype(bool)
    s = np.where(isb, 0.0, 1.0 / np.sqrt(np.abs(diag)))
    fc = b.copy()
    halo_sum(fc)
    gt = np.where(isb, g, 0.0)
    y = A.matvec(gt)
    halo_sum(y)
    r = s * (fc - y)
    x = np.zeros(nd)
    p = r.copy()
    At = lambda v: s * A.matvec(s * v)
    rr = gsum(np.sum(r[owned] ** 2))
    rr0 = rr
    it = 0
    if opk != "lap":      # the device's element-partitioned BiCGStab (capi.hip solve_run, dist branch), 4 all-reduces / iteration
        r0 = r.copy()
        rho = alpha = omega = 1.0
        rho_new = rr
        v = np.zeros(nd)
        while rr > (1e-11 ** 2) * rr0 and it < 5000:
            beta = 0.0 if it == 0 else (rho_new / rho) * (alpha / omega)
            p = r.copy() if it == 0 else r + beta * (p - omega * v)
            v = At(p)
            (r0v,) = halo_sum(v, extra=(float(r0 @ v),))       # w.(A x): no weighting
            alpha_new = rho_new / r0v
            sv = r - alpha_new * v
            t = At(sv)
            (ts,) = halo_sum(t, extra=(float(sv @ t),))
            tt = gsum(np.sum(t[owned] ** 2))                    # assembled t: owned rows
            om = ts / tt
            x += alpha_new * p + om * sv
            r = sv - om * t
            rho, alpha, omega = rho_new, alpha_new, om
            rho_new, rr = gsum(np.sum((r0 * r)[owned])), gsum(np.sum(r[owned] ** 2))
            it += 1
    while opk == "lap" and rr > (1e-11 ** 2) * rr0 and it < 5000:
        y = At(p)
        (pAp,) = halo_sum(y, extra=(float(p @ y),))          # p.(A_p p) needs no weighting
        alpha = rr / pAp
        x += alpha * p
        r -= alpha * y
        rr_new = gsum(np.sum(r[owned] ** 2))
        p = r + (rr_new / rr) * p
        rr = rr_new
        it += 1
    u = s * x + gt

    # ---- single-domain oracle solve of the whole mesh
    ref = o.pde_init_solve(gm, order, mkop(), forcing_q=f(o.quadrature_nodes(gm, order)), dirichlet=g_fn(gcoords), direct=True)
    err = np.linalg.norm(u - ref.solution[l2g_dof]) / np.linalg.norm(ref.solution)
    assert err < 1e-8, err
    print(f"rank {rank}: ok  cells {lp['cells'].shape[0]}  nodes {nd}  interface {ld.size}/{n_if}  iters {it}  err {err:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
