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


@pytest.mark.parametrize("advect", [False, True])
def test_one_interior_dof(capi, advect):
    """27 nodes of a 2 x 2 x 2 cube mesh, 26 of them on the boundary: the interior block is ONE row without an off-diagonal entry (found by
    tools/fuzz_small.py: the fill of the solver layout launched a grid of 0 workgroups).  The answer is f_i / A_ii with the lift."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl

    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_cube(2)
    op = -capi.laplacian() + (capi.advection((1.0, 0.5, 0.25)) + capi.reaction(1.0) if advect else capi.reaction(0.0))
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    _, bdofs, coords = c.dofs_get()
    assert nd == 27 and int((bdofs == 0).sum()) == 1
    c.set_operator(op)
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.set_dirichlet(0.5 + coords[:, 0])
    c.init()
    for _ in range(2):
        info = c.solve(rtol=1e-12)
        rp, ci = c.pattern_get()
        A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
        ref = spl.spsolve(A.tocsc(), c.force())
        assert info.converged == 1 and np.linalg.norm(c.solution() - ref) <= 1e-12 * np.linalg.norm(ref)
    c.close()


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


def _handle_case(capi):
    """mass matrix (SPD) behind the handle + a Dirichlet Laplace problem on the same context"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_cube(6)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    rp, ci = c.pattern_get()
    qn = c.quadrature_nodes()
    c.set_operator(-capi.laplacian())
    c.set_forcing(np.ones(qn.shape[0]))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    M = sp.csr_matrix((c.matrix_values(capi.MAT_MASS), ci, rp), shape=(nd, nd))
    b = np.random.default_rng(5).standard_normal(nd)
    xr = spla.splu(M.tocsc()).solve(b)
    return c, qn, nd, b, xr


def test_handle_survives_init_solve_init(capi):
    """lin_compute(M) -> init + solve -> init -> lin_solve(b).  The second init resets `solved`, but scale / sval still
    hold the stiffness system of the solve: the handle must notice and prepare its own matrix again."""
    c, qn, nd, b, xr = _handle_case(capi)
    c.lin_compute(capi.MAT_MASS, symmetric=True)
    x0, _ = c.lin_solve(b, rtol=1e-12)
    assert np.linalg.norm(x0 - xr) <= 1e-8 * np.linalg.norm(xr)
    c.init()
    c.solve()
    c.init()                                                  # solved = false again, scaled buffers still the stiffness system's
    x1, info = c.lin_solve(b, rtol=1e-12)
    assert info.converged == 1 and np.linalg.norm(x1 - xr) <= 1e-8 * np.linalg.norm(xr)
    c.set_forcing(2.0 * np.ones(qn.shape[0]))                 # every setter that used to reset `solved`
    c.set_dirichlet(np.ones(nd))
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    x2, _ = c.lin_solve(np.stack([b, 2 * b, -b, b, b], axis=1), rtol=1e-12)   # 5 columns: batch of 4 + 1
    assert np.linalg.norm(x2[:, 0] - xr) <= 1e-8 * np.linalg.norm(xr) and np.linalg.norm(x2[:, 1] - 2 * xr) <= 2e-8 * np.linalg.norm(xr)
    with pytest.raises(capi.FdapdeError):                     # c->u holds the handle's solutions now, not PDE::solution()
        c.solution()
    c.close()


def test_handle_survives_parabolic_solve(capi):
    """lin_compute(M) -> fdapde_solve_parabolic (which scales K = M/dt + A into the shared buffers and never sets
    `solved`) -> lin_solve(b)"""
    c, qn, nd, b, xr = _handle_case(capi)
    c.lin_compute(capi.MAT_MASS, symmetric=True)
    times = np.linspace(0.0, 0.2, 4)
    c.set_operator(capi.dt() - capi.laplacian())
    c.set_forcing(np.ones((qn.shape[0], times.size)))
    c.init()
    sol, pinfo = c.solve_parabolic(times, np.zeros(nd), np.zeros((nd, times.size)))
    assert pinfo.converged == 1
    x1, info = c.lin_solve(b, rtol=1e-12)
    assert info.converged == 1 and np.linalg.norm(x1 - xr) <= 1e-8 * np.linalg.norm(xr)
    c.close()


def test_pure_advection_zero_diagonal(capi):
    """b . grad u = f with u = 0 on the boundary: on a structured symmetric patch the advection matrix has an exactly zero diagonal
    at interior nodes (int psi_i d_x psi_i over a symmetric patch).  The Jacobi scale must not become 1/0: the
    reference's SparseLU either solves the system or reports failure; here the answer must be finite and, when the solve reports
    convergence, satisfy the exported system."""
    import scipy.sparse as sp
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(8, jitter=0.0, permute=False)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(capi.advection(np.array([1.0, 0.0])))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    rp, ci = c.pattern_get()
    A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
    interior = bnd == 0
    assert np.any(np.abs(A.diagonal()[interior]) < 1e-14)      # the case is what it claims to be
    info = c.solve(rtol=1e-10, maxit=2000, raise_on_noconv=False)
    u = c.solution()
    assert np.all(np.isfinite(u)) and np.isfinite(info.relres)
    assert info.method_used in (capi.SOLVER_BICGSTAB, capi.SOLVER_GMRES)   # (the open method: BiCGStab, then GMRES with what is left of the budget)
    if info.converged:
        Az = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
        bz = c.force()
        assert np.linalg.norm(Az @ u - bz) <= 1e-7 * np.linalg.norm(bz)
    c.close()


def test_peer_lists_are_validated(capi):
    """fdapde_halo_setup_peers refuses lists that cannot describe a neighbour exchange (and says why) instead of hanging in it"""
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(6)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    owned = np.ones(nd, dtype=np.uint8)
    with pytest.raises(capi.FdapdeError) as e:   # no communicator yet
        c.halo_setup_peers([1], [0, 2], [0, 1], owned)
    assert e.value.status == capi.ENOTINIT
    c.comm_init_callback(3, 1, lambda arr: None)
    for ranks, off, dofs in (([1], [0, 2], [0, 1]),            # this rank among its peers
                             ([2, 0], [0, 1, 2], [0, 1]),      # not ascending
                             ([0, 3], [0, 1, 2], [0, 1]),      # rank outside the communicator
                             ([0], [0, 2], [0, nd]),           # DOF id out of range
                             ([0], [0, 2], [3, 3]),            # a DOF listed twice for one peer
                             ([0], [1, 2], [0, 1])):           # offsets not starting at 0
        with pytest.raises(capi.FdapdeError) as e:
            c.halo_setup_peers(ranks, off, dofs, owned)
        assert e.value.status == capi.EINVAL, (ranks, off, dofs)
    c.halo_setup_peers([0, 2], [0, 2, 3], [0, 1, 1], owned)     # a DOF shared with two peers is fine
    c.halo_setup_peers([], [0], [], owned)                       # and so is a rank without neighbours
    c.close()


def test_device_memory_does_not_drift_over_context_lifetimes(capi):
    """every engine unit touched, context closed, eight times over: the free device memory after close() stays where the second lifetime
    left it (space-varying operator data, column boards, point-location grid, row-distributed / persistent layouts are all released)"""
    import ctypes as C

    from fdapde_core_amd import meshgen

    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    import gc

    gc.collect()   # (contexts earlier tests of the session left to the collector must not be released in the middle of this measurement)
    nodes, cells, bnd = meshgen.unit_cube(12)
    n2, c2, b2 = meshgen.unit_square(40)
    seen = []
    for it in range(8):
        c = capi.Context(0)
        c.mesh_upload(nodes, cells, bnd)
        nd = c.dofs_build(1 + it % 2)
        qn = c.quadrature_nodes()
        c.set_operator(-capi.laplacian() + capi.reaction_field(1.0 + qn[:, 0]))
        c.set_forcing(np.ones(qn.shape[0]))
        c.set_dirichlet(np.zeros(nd))
        c.init()
        c.solve(rtol=1e-9)
        c.set_operator(-capi.laplacian() + capi.advection([1.0, 0.5, 0.25]) + capi.reaction(1.0))
        c.init()
        c.solve(rtol=1e-9)
        c.lin_compute(capi.MAT_STIFF, symmetric=False)
        c.lin_solve(np.ones((nd, 5)))
        c.eval_pointwise(np.random.default_rng(0).uniform(0.1, 0.9, (200, 3)))
        c.mesh_upload(n2, c2, b2)
        nd = c.dofs_build(2)
        c.set_operator(capi.dt() - capi.laplacian())
        qn = c.quadrature_nodes()
        c.set_forcing(np.zeros((qn.shape[0], 4)))
        c.init()
        _, _, co = c.dofs_get()
        c.solve_parabolic(np.linspace(0, 0.1, 4), np.prod(np.sin(np.pi * co), axis=1), dirichlet=np.zeros((nd, 4)))
        c.eval_pointwise(np.random.default_rng(0).uniform(0.1, 0.9, (200, 2)))
        c.close()
        seen.append(free_bytes())
    assert max(seen[1:]) - min(seen[1:]) <= 8 << 20, seen   # (the first lifetime also fills the runtime's own pools)
