"""Parabolic stepper: FEMLinearParabolicSolver::solve (fem_linear_parabolic_solver.h:37-72)
through the C ABI against the oracle's LU time stepping and against the reference's own gates (fem_pde_test.cpp:222-368)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from fdapde_loader import load_package

    m = load_package().capi
    assert m.load().fdapde_device_count() >= 1
    return m


def _problem(ctx, n_times):
    pi = np.pi
    times = np.linspace(0.0, 1.0, n_times)
    u = lambda x, t: np.sin(2 * pi * x[:, 0]) * np.sin(2 * pi * x[:, 1]) * np.exp(-t)
    f = lambda x, t: (8 * pi * pi - 1.0) * np.sin(2 * pi * x[:, 0]) * np.sin(2 * pi * x[:, 1]) * np.exp(-t)
    _, _, coords = ctx.dofs_get()
    qn = ctx.quadrature_nodes()
    return times, np.stack([f(qn, t) for t in times], axis=1), np.stack([u(coords, t) for t in times], axis=1)


def _run(capi, mesh, order, n_times):
    ctx = capi.Context(device=0)
    ctx.mesh_upload(mesh.nodes, mesh.cells, mesh.boundary)
    ctx.dofs_build(order)
    times, F, G = _problem(ctx, n_times)
    ctx.set_operator(capi.dt() - capi.laplacian())
    ctx.set_forcing(F)
    ctx.init()
    sol, info = ctx.solve_parabolic(times, G[:, 0], G, rtol=1e-11)
    assert info.converged == 1
    return ctx, times, F, G, sol


@pytest.mark.parametrize("order", [1, 2])
def test_parabolic_matches_oracle(capi, oracle, mesh_loader, order):
    m = mesh_loader("unit_square_32")
    ctx, times, F, G, sol = _run(capi, m, order, 11)
    ref, _ = oracle.pde_parabolic_solve(m, order, oracle.dt() - oracle.laplacian(), times, F, G, G[:, 0])
    assert np.array_equal(sol[:, 0], G[:, 0])
    for j in range(1, times.size):
        assert np.linalg.norm(sol[:, j] - ref[:, j]) / np.linalg.norm(ref[:, j]) <= 1e-8, j
    assert ctx.force(ncols=times.size).shape[0] == sol.shape[0] * times.size
    ctx.close()


def test_parabolic_isotropic_order2(capi, oracle, mesh_loader):
    """fem_pde_test.cpp:222-285 through the device path: 101 time points, P2 on unit_square, < 1e-7"""
    import scipy.sparse as sp

    m = mesh_loader("unit_square")
    ctx, times, F, G, sol = _run(capi, m, 2, 101)
    rp, ci = ctx.pattern_get()
    M = sp.csr_matrix((ctx.matrix_values(capi.MAT_MASS), ci, rp), shape=(sol.shape[0],) * 2)
    errs = [float(np.sum(M @ ((G[:, j] - sol[:, j]) ** 2))) for j in range(times.size)]
    assert max(errs) < 1e-7
    ctx.close()


def test_parabolic_order1_convergence(capi, oracle, mesh_loader):
    """fem_pde_test.cpp:295-368 on the 16/32/64 fixtures: floor(log2(e_h / e_{h/2})) == 2"""
    import scipy.sparse as sp

    errs = []
    for name in ("unit_square_16", "unit_square_32", "unit_square_64"):
        m = mesh_loader(name)
        ctx, times, F, G, sol = _run(capi, m, 1, 31)
        rp, ci = ctx.pattern_get()
        M = sp.csr_matrix((ctx.matrix_values(capi.MAT_MASS), ci, rp), shape=(sol.shape[0],) * 2)
        errs.append(np.sqrt(float(np.sum(M @ ((G[:, -1] - sol[:, -1]) ** 2)))))
        ctx.close()
    for a, b in zip(errs[:-1], errs[1:]):
        assert np.floor(np.log2(a / b)) == 2


def test_parabolic_needs_init_and_forcing_columns(capi, mesh_loader):
    m = mesh_loader("unit_square_16")
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(1)
    times = np.linspace(0, 1, 5)
    with pytest.raises(capi.FdapdeError) as e:
        ctx.solve_parabolic(times, np.zeros(nd))
    assert e.value.status == capi.ENOTINIT
    ctx.set_operator(capi.dt() - capi.laplacian())
    ctx.set_forcing(np.zeros((3 * m.n_cells, 2)))
    ctx.init()
    with pytest.raises(capi.FdapdeError) as e:
        ctx.solve_parabolic(times, np.zeros(nd))
    assert e.value.status == capi.EINVAL
    ctx.close()
