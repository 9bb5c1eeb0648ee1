"""The all-cores CPU baseline (oracle/fem_oracle_mt.c, BASELINE.md "CPU-best") against the faithful single-threaded port:
same pattern, values to 1e-12 relative (summation order differs: colour classes instead of cell order), same PCG iterates to 1e-9."""
import numpy as np
import pytest


@pytest.mark.parametrize("dim,nx,order", [(2, 24, 1), (2, 12, 2), (3, 6, 1), (3, 4, 2)])
def test_multithreaded_port_matches_faithful_port(oracle, dim, nx, order):
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    m = oracle.Mesh(nodes, cells, bnd)
    dofs, b, nd, _ = oracle.enumerate_dofs(m, order)
    fq = np.sin(oracle.quadrature_nodes(m, order).sum(axis=1))
    colouring = oracle.mt_colour_cells(dofs, nd)
    order_c, cptr = colouring
    assert sorted(order_c.tolist()) == list(range(cells.shape[0]))
    for k in range(len(cptr) - 1):   # no two cells of a colour share a DOF
        d = dofs[order_c[cptr[k]:cptr[k + 1]]].reshape(-1)
        assert np.unique(d).size == d.size
    for op in (-oracle.laplacian(), -oracle.laplacian() + oracle.advection(np.arange(1, dim + 1) * 0.5) + oracle.reaction(0.7)):
        A = oracle.assemble_operator(m, order, dofs, nd, op)
        rhs = oracle.assemble_forcing(m, order, dofs, nd, fq)
        A2, rhs2 = oracle.mt_assemble(m, order, dofs, nd, op, A, colouring, fq)
        scale = np.abs(A.values).max()
        assert np.abs(A2.values - A.values).max() <= 1e-12 * scale
        assert np.abs(rhs2 - rhs).max() <= 1e-12 * max(1.0, np.abs(rhs).max())
    A = oracle.assemble_operator(m, order, dofs, nd, -oracle.laplacian())
    u1, it1, rr1, rc1 = oracle.pcg(A, rhs, b, np.zeros(nd), rtol=1e-11)
    u2, it2, rr2, rc2 = oracle.mt_pcg(A, rhs, b, np.zeros(nd), rtol=1e-11)
    assert rc1 == 0 and rc2 == 0 and abs(it1 - it2) <= 2
    assert np.abs(u1 - u2).max() <= 1e-9 * max(1.0, np.abs(u1).max())
    assert oracle.mt_threads() >= 1
