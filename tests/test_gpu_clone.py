"""fdapde_ctx_clone -- the copy operation behind the reference's type-erased PDE handle (make_pde -> heap_storage: new T(obj),
fdaPDE/pde/pde.h:167-169, fdaPDE/utils/type_erasure.h:130-146): the clone answers every getter like the source and can be solved
without another init; what either context does afterwards does not reach the other.  Checked against the oracle like any solve."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from fdapde_loader import load_package

    m = load_package().capi
    assert m.load().fdapde_device_count() >= 1
    return m


def _setup(capi, m, order, op):
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    _, _, coords = ctx.dofs_get()
    qn = ctx.quadrature_nodes()
    ctx.set_operator(op)
    ctx.set_forcing(np.sin(qn.sum(axis=1)))
    return ctx, nd, coords, qn


@pytest.mark.parametrize("mesh_name,order", [("unit_square_32", 1), ("c_shaped", 2), ("unit_sphere", 1), ("unit_sphere", 2)])
def test_clone_after_init_solves_without_a_second_init(capi, oracle, mesh_loader, mesh_name, order):
    m = mesh_loader(mesh_name)
    ctx, nd, coords, qn = _setup(capi, m, order, -capi.laplacian() + capi.reaction(0.5))
    g = coords.sum(axis=1)
    ctx.set_dirichlet(g)
    ctx.init()
    twin = ctx.clone()
    # every getter of the clone returns the source's bits
    for which in (capi.MAT_STIFF, capi.MAT_MASS):
        assert np.array_equal(twin.matrix_values(which), ctx.matrix_values(which))
    assert np.array_equal(twin.force(), ctx.force())
    for a, b in zip(twin.dofs_get(), ctx.dofs_get()):
        assert np.array_equal(a, b)
    for a, b in zip(twin.pattern_get(), ctx.pattern_get()):
        assert np.array_equal(a, b)
    # the clone solves the same problem to the same bits (same deterministic set-up, same kernels) ...
    i1, i2 = ctx.solve(rtol=1e-10), twin.solve(rtol=1e-10)
    assert i1.converged == 1 and i2.converged == 1 and i1.iters == i2.iters
    assert np.array_equal(ctx.solution(), twin.solution())
    # ... and the right one: the oracle's direct solve of the row-zeroed system
    fq = np.sin(qn.sum(axis=1))
    ref = oracle.pde_init_solve(m, order, -oracle.laplacian() + oracle.reaction(0.5), forcing_q=fq, dirichlet=g)
    assert np.linalg.norm(twin.solution() - ref.solution) / np.linalg.norm(ref.solution) <= 1e-8
    # diverge: new data on the clone only
    twin.set_dirichlet(g + 1.0)
    twin.set_forcing(fq + 0.5)
    twin.init()
    twin.solve(rtol=1e-10)
    ref2 = oracle.pde_init_solve(m, order, -oracle.laplacian() + oracle.reaction(0.5), forcing_q=fq + 0.5, dirichlet=g + 1.0)
    assert np.linalg.norm(twin.solution() - ref2.solution) / np.linalg.norm(ref2.solution) <= 1e-8
    before = ctx.solution()
    ctx.solve(rtol=1e-10)
    assert np.array_equal(ctx.solution(), before)   # the source never saw the clone's data
    twin.close(), ctx.close()


def test_clone_after_solve_carries_the_solution_and_the_row_zeroed_export(capi, oracle, mesh_loader):
    m = mesh_loader("unit_square_16")
    ctx, nd, coords, qn = _setup(capi, m, 2, -capi.laplacian() + capi.advection(np.array([0.7, -0.2])))
    ctx.set_dirichlet(np.zeros(nd))
    ctx.init()
    info = ctx.solve(rtol=1e-10)
    assert info.converged == 1
    twin = ctx.clone()
    assert np.array_equal(twin.solution(), ctx.solution())
    assert np.array_equal(twin.matrix_values(capi.MAT_STIFF), ctx.matrix_values(capi.MAT_STIFF))   # boundary rows zeroed, unit diagonal
    assert np.array_equal(twin.force(), ctx.force())
    ctx.close()   # the clone owns everything it needs
    assert twin.solve(rtol=1e-10).converged == 1
    ref = oracle.pde_init_solve(m, 2, -oracle.laplacian() + oracle.advection(np.array([0.7, -0.2])), forcing_q=np.sin(qn.sum(axis=1)),
                                dirichlet=np.zeros(nd))
    assert np.linalg.norm(twin.solution() - ref.solution) / np.linalg.norm(ref.solution) <= 1e-8
    twin.close()


def test_clone_carries_space_varying_coefficients_and_the_handle(capi, oracle, mesh_loader):
    m = mesh_loader("unit_sphere")
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(1)
    qn = ctx.quadrature_nodes()
    cq = 1.0 + qn[:, 0] ** 2
    ctx.set_operator(-capi.laplacian() + capi.reaction_field(cq))
    ctx.set_forcing(np.ones(qn.shape[0]))
    twin0 = ctx.clone()   # before init: operator data and forcing samples travel
    ctx.init(), twin0.init()
    assert np.array_equal(ctx.matrix_values(capi.MAT_STIFF), twin0.matrix_values(capi.MAT_STIFF))
    od, _, _, _ = oracle.enumerate_dofs(m, 1)
    ref = oracle.assemble_operator(m, 1, od, nd, -oracle.laplacian() + oracle.reaction_field(cq))
    assert np.abs(twin0.matrix_values(capi.MAT_STIFF) - ref.values).max() <= 1e-13 * np.abs(ref.values).max()
    # the factor-once handle: its matrix travels, the clone prepares the scaled system again on its first solve
    ctx.lin_compute(capi.MAT_STIFF)
    b = np.cos(np.arange(nd) * 0.01)
    x1, _ = ctx.lin_solve(b)
    twin1 = ctx.clone()
    x2, _ = twin1.lin_solve(b)
    assert np.array_equal(x1, x2)
    for c in (ctx, twin0, twin1):
        c.close()


def test_a_rank_of_a_multi_gpu_job_is_not_cloned(capi, mesh_loader):
    m = mesh_loader("unit_square_16")
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    ctx.dofs_build(1)
    ctx.comm_init_callback(1, 0, lambda buf: None)
    with pytest.raises(capi.FdapdeError) as e:
        ctx.clone()
    assert e.value.status == capi.EUNSUPPORTED
    ctx.close()
