"""Degenerate but legal inputs of the solve: no interior DOF, zero right-hand side, no Dirichlet DOF, a singular operator.
The reference handles them through its LU (`fem_linear_elliptic_solver.h:38-47`: `success = false` when the factorisation
fails, otherwise the exact answer); the Krylov path must give the same answers without NaNs, hangs or wasted iterations."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from fdapde_loader import load_package

    m = load_package().capi
    assert m.load().fdapde_device_count() >= 1
    return m


def _solve(capi, nodes, cells, bnd, op, f=1.0, g=0.0, order=1, **kw):
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    c.set_operator(op)
    c.set_forcing(np.full(c.quadrature_nodes().shape[0], f))
    c.set_dirichlet(np.full(nd, g))
    c.init()
    info = c.solve(rtol=1e-10, **kw)
    return info, c.solution()


TRI = (np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]]), np.array([[0, 1, 2]], dtype=np.int32), np.ones(3, dtype=np.uint8))


@pytest.mark.parametrize("order", [1, 2])
def test_single_cell_all_dirichlet(capi, order):
    info, u = _solve(capi, *TRI, -capi.laplacian(), g=2.0, order=order)
    assert info.iters == 0 and np.array_equal(u, np.full(u.shape, 2.0))


@pytest.mark.parametrize("advect", [False, True])
def test_every_dof_on_the_boundary(capi, advect):
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(8)
    op = -capi.laplacian() + capi.advection(np.array([1.0, 0.5])) if advect else -capi.laplacian()
    info, u = _solve(capi, nodes, cells, np.ones_like(bnd), op, g=1.5)
    assert info.iters == 0 and np.array_equal(u, np.full(u.shape, 1.5))


@pytest.mark.parametrize("dim,order,advect", [(2, 1, False), (2, 1, True), (3, 2, False)])
def test_zero_right_hand_side(capi, dim, order, advect):
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(8) if dim == 2 else meshgen.unit_cube(3)
    op = -capi.laplacian() + capi.advection(np.array([1.0, 0.5])) if advect else -capi.laplacian()
    info, u = _solve(capi, nodes, cells, bnd, op, f=0.0, order=order)
    assert info.iters == 0 and not u.any()


def test_constant_lift_is_reproduced(capi):
    """f = 0 and g = 3 on the whole boundary: the discrete harmonic extension of a constant is that constant."""
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(8)
    info, u = _solve(capi, nodes, cells, bnd, -capi.laplacian(), f=0.0, g=3.0)
    assert np.abs(u - 3.0).max() < 1e-8


def test_no_dirichlet_dof(capi):
    """-Laplace u + u = 1 with natural boundary conditions everywhere: u = 1."""
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(8)
    info, u = _solve(capi, nodes, cells, np.zeros_like(bnd), -capi.laplacian() + capi.reaction(1.0))
    assert np.abs(u - 1.0).max() < 1e-8


def test_singular_operator_reports_failure(capi):
    """Pure Neumann Laplacian with an incompatible right-hand side: the reference's LU reports failure (`success = false`); here the
    solve must stop with a status, not hang or return NaNs silently."""
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(8)
    with pytest.raises(capi.FdapdeError):
        _solve(capi, nodes, cells, np.zeros_like(bnd), -capi.laplacian(), maxit=200)


def test_space_varying_data_follows_the_operator(capi):
    """The space-varying coefficient data stays on the device between two `init` calls of the same operator; an
    `assemble_operator` call with other data, or a new `set_operator`, must invalidate it."""
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(6)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    qn = c.quadrature_nodes()
    c1, c2 = 1.0 + qn[:, 0], 3.0 + np.sin(qn[:, 1])
    c.set_forcing(np.ones(qn.shape[0]))
    c.set_dirichlet(np.zeros(nd))
    c.set_operator(-capi.laplacian() + capi.reaction_field(c1))
    c.init()
    v1 = c.matrix_values(capi.MAT_STIFF).copy()
    c.init()                                                   # same operator: data reused
    assert np.array_equal(c.matrix_values(capi.MAT_STIFF), v1)
    c.assemble_operator(capi.MAT_STIFF, capi.reaction_field(c2))   # other data through the same device slots
    c.init()
    assert np.array_equal(c.matrix_values(capi.MAT_STIFF), v1)
    c.set_operator(-capi.laplacian() + capi.reaction_field(c2))
    c.init()
    v2 = c.matrix_values(capi.MAT_STIFF)
    assert not np.array_equal(v2, v1)
    c.set_operator(-capi.laplacian() + capi.reaction_field(c1))
    c.init()
    assert np.array_equal(c.matrix_values(capi.MAT_STIFF), v1)
