"""Basis evaluation matrices (PDE__::eval_basis, pde/pde.h:149-158) on the device against the
reference's four golden .mtx files (test/src/lagrangian_basis_test.cpp:200-238) and against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from fdapde_loader import load_package

    m = load_package().capi
    assert m.load().fdapde_device_count() >= 1
    return m


def almost_equal_mat(a, b, eps=1e-7):   # test/src/utils/utils.h:44-48
    d = np.abs(a - b).max()
    return d < eps or d < max(np.abs(a).max(), np.abs(b).max()) * eps


@pytest.mark.parametrize("order", [1, 2])
def test_pointwise_evaluation_golden(capi, oracle, mesh_loader, golden_dir, order):
    """lagrangian_basis_test.cpp:200-208, 222-229"""
    m = mesh_loader("c_shaped")
    locs = oracle.read_csv(os.path.join(golden_dir, "mesh", "c_shaped", "locs.csv"))
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    ctx.dofs_build(order)
    psi, D, cells = ctx.eval_pointwise(locs)
    gold = oracle.read_mtx(os.path.join(golden_dir, "mtx", f"lagrangian_pointwise_eval_order{order}.mtx"))
    assert psi.shape == gold.shape and np.all(cells >= 0) and np.all(D == 1.0)
    assert almost_equal_mat(psi.toarray(), gold)
    assert np.abs(psi.toarray() - gold).max() < 1e-12
    ctx.close()


@pytest.mark.parametrize("order", [1, 2])
def test_areal_evaluation_golden(capi, oracle, mesh_loader, golden_dir, order):
    """lagrangian_basis_test.cpp:211-219, 232-238"""
    m = mesh_loader("quasi_circle")
    inc = oracle.read_csv(os.path.join(golden_dir, "mesh", "quasi_circle", "incidence_matrix.csv"))
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    psi, D = ctx.eval_areal(inc)
    gold = oracle.read_mtx(os.path.join(golden_dir, "mtx", f"lagrangian_areal_eval_order{order}.mtx"))
    assert psi.shape == gold.shape
    assert almost_equal_mat(psi.toarray(), gold)
    od, _, ond, _ = oracle.enumerate_dofs(m, order)
    _, Dref = oracle.areal_psi(m, order, od, ond, inc)
    assert np.abs(D - Dref).max() < 1e-14
    ctx.close()


@pytest.mark.parametrize("mesh_name,order", [("unit_sphere", 1), ("unit_sphere", 2), ("unit_square", 2)])
def test_pointwise_evaluation_random_points(capi, oracle, mesh_loader, mesh_name, order):
    """random interior points (incl. 3-D), points outside the domain, points exactly on vertices: located / not located as
    expected, partition of unity, and the interpolant of a polynomial in the space is exact"""
    m = mesh_loader(mesh_name)
    rng = np.random.default_rng(9)
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    _, _, coords = ctx.dofs_get()
    cells = rng.integers(0, m.n_cells, 500)
    w = rng.dirichlet(np.ones(m.M + 1), 500)
    inside = np.einsum("ij,ijk->ik", w, m.nodes[m.cells[cells]])
    outside = m.nodes.max(axis=0) + rng.uniform(0.5, 1.0, (20, m.N))
    vertices = m.nodes[rng.integers(0, m.n_nodes, 50)]
    locs = np.vstack([inside, outside, vertices])
    psi, _, found = ctx.eval_pointwise(locs)
    assert np.all(found[:500] >= 0) and np.all(found[500:520] == -1) and np.all(found[520:] >= 0)
    rows = np.r_[0:500, 520:570]
    assert np.abs(np.asarray(psi.sum(axis=1)).ravel()[rows] - 1.0).max() < 1e-12     # partition of unity
    assert psi[500:520].nnz == 0
    f = (lambda x: 1.0 + 2 * x[:, 0] - x[:, -1]) if order == 1 else (lambda x: x[:, 0] ** 2 - x[:, 0] * x[:, -1] + x[:, 1])
    assert np.abs((psi @ f(coords))[rows] - f(locs[rows])).max() < 1e-11
    ctx.close()


@pytest.mark.parametrize("dim", [2, 3])
def test_point_location_on_minimal_and_anisotropic_meshes(capi, dim):
    """the point-location grid (built on the device once per mesh): one cell (a single bin), a thin strip (cells far longer than the bins are
    wide: every cell registered in many bins), repeated calls and a second mesh on the same context (the grid of the mesh before must go)"""
    ctx = capi.Context(device=0)
    rng = np.random.default_rng(2)
    # one cell
    nodes = np.vstack([np.zeros(dim), np.eye(dim)])
    cells = np.arange(dim + 1, dtype=np.int32).reshape(1, -1)
    ctx.mesh_upload(nodes, cells, np.ones(dim + 1, dtype=np.uint8))
    ctx.dofs_build(1)
    w = rng.dirichlet(np.ones(dim + 1), 50)
    locs = w @ nodes
    psi, _, found = ctx.eval_pointwise(np.vstack([locs, [[2.0] * dim]]))
    assert np.all(found[:50] == 0) and found[50] == -1
    assert np.abs(psi[:50].toarray() - w).max() < 1e-12     # P1 basis values at a point = its barycentric coordinates
    # a strip of very elongated cells, on the same context
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import meshgen

    n2, c2, b2 = meshgen.unit_square(12) if dim == 2 else meshgen.unit_cube(5)
    n2 = n2.copy()
    n2[:, 0] *= 200.0
    ctx.mesh_upload(n2, c2, b2)
    nd = ctx.dofs_build(2)
    _, _, coords = ctx.dofs_get()
    pick = rng.integers(0, c2.shape[0], 400)
    w = rng.dirichlet(np.ones(dim + 1), 400)
    locs = np.einsum("ij,ijk->ik", w, n2[c2[pick]])
    f = lambda x: 0.5 + x[:, 0] * 1e-2 - x[:, 1] ** 2 + x[:, 0] * x[:, -1] * 1e-2
    for _ in range(2):
        psi, _, found = ctx.eval_pointwise(locs)
        assert np.all(found >= 0)
        assert np.abs(np.asarray(psi.sum(axis=1)).ravel() - 1.0).max() < 1e-11
        assert np.abs(psi @ f(coords) - f(locs)).max() < 1e-9
    ctx.close()
