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
    info = fdist.interface_info(cells, part, nodes.shape[0], world)
    lp = fdist.local_problem(nodes, cells, bnd, part, rank, world, info)
    # ---- invariants of the maps
    mult, owner, ifnodes = info
    assert np.all(lp["l2g"][lp["local_dof"]] == ifnodes[lp["if_index"]])
    own_count = torch.zeros(nodes.shape[0], dtype=torch.int64)
    own_count[torch.from_numpy(lp["l2g"][lp["owned"] == 1])] += 1
    dist.all_reduce(own_count)
    assert int(own_count.min()) == 1 and int(own_count.max()) == 1      # every global DOF owned exactly once
    ncell = torch.tensor([lp["cells"].shape[0]])
    dist.all_reduce(ncell)
    assert int(ncell) == cells.shape[0]

    # ---- local sub-assembled operators (oracle on the sub-mesh of this rank's cells)
    lm = o.Mesh(lp["nodes"], lp["cells"], lp["boundary"])
    dofs, lb, nd, _ = o.enumerate_dofs(lm, 1)
    A = o.assemble_operator(lm, 1, dofs, nd, -o.laplacian())
    b = o.assemble_forcing(lm, 1, dofs, nd, f(o.quadrature_nodes(lm, 1)))
    g = g_fn(lp["nodes"])
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
    while rr > (1e-11 ** 2) * rr0 and it < 5000:
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
    gm = o.Mesh(nodes, cells, bnd)
    ref = o.pde_init_solve(gm, 1, -o.laplacian(), forcing_q=f(o.quadrature_nodes(gm, 1)), dirichlet=g_fn(nodes), direct=True)
    err = np.linalg.norm(u - ref.solution[lp["l2g"]]) / np.linalg.norm(ref.solution)
    assert err < 1e-8, err
    print(f"rank {rank}: ok  cells {lp['cells'].shape[0]}  nodes {nd}  interface {ld.size}/{n_if}  iters {it}  err {err:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
