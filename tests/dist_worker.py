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
