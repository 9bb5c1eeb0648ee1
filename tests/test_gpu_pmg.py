"""FDAPDE_SOLVER_PMG (csrc/eng_pmg.hip): BiCGStab on an order-2 space with a two-level preconditioner -- the fine level's Jacobi sweep + a correction from the
P1 space of the same mesh -- against scipy's SuperLU on the reference's own row-zeroed system (fem_solver_base.h:142-155, fem_linear_elliptic_solver.h:38-47)
and against the Jacobi-preconditioned stages of the open method."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi, meshgen, workloads

    assert capi.load().fdapde_device_count() >= 1
    return capi, meshgen, workloads


def _csr(c, capi, nd):
    import scipy.sparse as sp

    rp, ci = c.pattern_get()
    return sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))


def _problem(capi, meshgen, dim, nx, op, dirichlet):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    c.tune("pmg_setup_check", 1)   # (every test problem: the device-built transfer tables against the host loops that built them first -- an error if they differ)
    _, bd, coords = c.dofs_get()
    c.set_operator(op)
    qn = c.quadrature_nodes()
    c.set_forcing(1.0 + np.sin(3.0 * qn[:, 0]) * qn[:, 1])
    if dirichlet == "zero":
        c.set_dirichlet(np.zeros(nd))
    elif dirichlet == "data":
        c.set_dirichlet(0.3 * np.cos(2.0 * coords[:, 0]) + coords[:, -1])
    c.init()
    return c, nd, bd, coords


@pytest.mark.parametrize("dim,nx,kind,dirichlet", [(2, 24, "adr", "data"), (2, 40, "sym", "zero"), (3, 6, "adr", "data"), (3, 10, "adr", "zero"), (3, 8, "sym", "data"),
                                                   (2, 16, "tensor", "data")])
def test_two_level_solver_against_lu(env, dim, nx, kind, dirichlet):
    """2-D / 3-D, with advection (a BiCGStab coarse solve) and without (CG), a diffusion tensor, zero and non-zero Dirichlet data: the LU solution of the
    reference's row-zeroed system to 1e-8, in a number of iterations that does not grow with the mesh"""
    import scipy.sparse.linalg as spl

    capi, meshgen, _ = env
    b = [3.0, -1.5] if dim == 2 else [1.0, 0.5, 0.25]
    K = np.array([[2.0, 0.3], [0.3, 1.0]])
    op = {"adr": -capi.laplacian() + capi.advection(b) + capi.reaction(1.0), "sym": -capi.laplacian() + capi.reaction(2.0),
          "tensor": -capi.diffusion(K) + capi.reaction(0.5)}[kind]
    c, nd, bd, coords = _problem(capi, meshgen, dim, nx, op, dirichlet)
    info = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)
    assert info.converged == 1 and info.method_used == capi.SOLVER_PMG and info.relres <= 1e-10
    assert info.iters <= 40, info.iters
    u = c.solution()
    A = _csr(c, capi, nd)   # (after the solve: the reference's row-zeroed matrix)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(u - ref) <= 1e-8 * np.linalg.norm(ref)
    again = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)   # (the coarse level is kept: the same iterations, the same bits)
    assert again.iters == info.iters and np.array_equal(c.solution(), u)
    krylov = c.solve(rtol=1e-11)   # the open method on a system this small: the Jacobi-preconditioned stages
    assert krylov.method_used != capi.SOLVER_PMG
    assert np.linalg.norm(c.solution() - u) <= 1e-8 * np.linalg.norm(u)
    c.close()


def test_iterations_do_not_grow_with_the_mesh(env):
    capi, meshgen, workloads = env
    its = []
    for nx in (6, 12, 20):
        c, nd, _, _ = _problem(capi, meshgen, 3, nx, workloads.c5_operator(capi), "zero")
        info = c.solve(method=capi.SOLVER_PMG, rtol=1e-10)
        its.append(info.iters)
        c.close()
    assert max(its) <= 40 and max(its) - min(its) <= 6, its   # (flexible GMRES: one operator / preconditioner application per iteration; 34 - 36)


def test_new_matrix_new_coarse_operator_and_custom_boundary(env):
    """fdapde_init again with another operator: the coarse operator follows; a boundary mask set by the caller (Dirichlet data on a part of the boundary)
    reaches the coarse level"""
    import scipy.sparse.linalg as spl

    capi, meshgen, _ = env
    nodes, cells, bnd = meshgen.unit_cube(8)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    _, bd, coords = c.dofs_get()
    part = (bd != 0) & (coords[:, 0] < 1e-12)   # Dirichlet data on the face x = 0 only
    c.dofs_set_boundary(part.astype(np.uint8))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.set_dirichlet(np.where(part, 1.0 + coords[:, 1], 0.0))
    for op in (-capi.laplacian() + capi.reaction(1.0), -capi.laplacian() + capi.advection([2.0, 0.0, -1.0]) + capi.reaction(3.0)):
        c.set_operator(op)
        c.init()
        info = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)
        assert info.converged == 1 and info.iters <= 40
        A = _csr(c, capi, nd)
        ref = spl.spsolve(A.tocsc(), c.force())
        assert np.linalg.norm(c.solution() - ref) <= 1e-8 * np.linalg.norm(ref)
    # the mask changed AFTER the coarse level was built: it is built again with the new one
    part2 = (bd != 0) & (coords[:, 1] > 1.0 - 1e-12)
    c.dofs_set_boundary(part2.astype(np.uint8))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.set_dirichlet(np.where(part2, 2.0, 0.0))
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.init()
    info = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)
    assert info.converged == 1 and info.iters <= 40
    A = _csr(c, capi, nd)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(c.solution() - ref) <= 1e-8 * np.linalg.norm(ref)
    c.close()


def test_what_it_does_not_take_and_when_the_open_method_takes_it(env):
    capi, meshgen, _ = env
    nodes, cells, bnd = meshgen.unit_square(16)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)   # order 1: no coarse level to take
    c.set_operator(-capi.laplacian())
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    with pytest.raises(capi.FdapdeError) as e:
        c.solve(method=capi.SOLVER_PMG)
    assert e.value.status == capi.EUNSUPPORTED
    nd = c.dofs_build(2)
    qn = c.quadrature_nodes()
    c.set_forcing(np.ones(qn.shape[0]))
    c.set_dirichlet(np.zeros(nd))
    # the open method takes the two-level solver from `pmg_auto_rows` DOFs on (default 300 k: where it starts to win)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.init()
    assert c.solve().method_used != capi.SOLVER_PMG
    u_small = c.solution()
    c.tune("pmg_auto_rows", 100)
    c.tune("pmg_auto_first_rows", 100)
    info = c.solve()
    assert info.converged == 1 and info.method_used == capi.SOLVER_PMG
    assert np.abs(c.solution() - u_small).max() <= 1e-8 * np.abs(u_small).max()
    c.tune("pmg_auto", 0)
    assert c.solve().method_used != capi.SOLVER_PMG
    c.close()
    # rent-or-buy: between `pmg_auto_rows` and `pmg_auto_first_rows` a context's FIRST open-method solve keeps the Jacobi stages (the coarse level's set-up costs
    # more than one solve saves), the second builds the level; a parabolic run of more than four steps takes it at once
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    c.tune("pmg_auto_rows", 100)
    first, second = c.solve(), c.solve()
    assert first.method_used != capi.SOLVER_PMG and second.method_used == capi.SOLVER_PMG and second.converged == 1
    c.close()
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    nq = c.quadrature_nodes().shape[0]
    c.tune("pmg_auto_rows", 100)
    c.set_forcing(np.ones((nq, 7)))
    c.init()
    _, info = c.solve_parabolic(0.01 * np.arange(7), np.zeros(nd), dirichlet=np.zeros((nd, 7)))
    assert info.method_used == capi.SOLVER_PMG and info.converged == 1
    c.close()


def test_open_method_falls_through_where_the_coarse_level_does_not_help(env):
    """a strongly indefinite operator (-Lap - 300: negative eigenvalues on both levels): whatever the two-level solver makes of it -- a solution, or giving
    up after its short budget / four coarse solves in a row that did not converge -- the open method hands out the LU solution"""
    import scipy.sparse.linalg as spl

    capi, meshgen, _ = env
    c, nd, bd, coords = _problem(capi, meshgen, 3, 8, -capi.laplacian() + capi.reaction(-300.0), "data")
    c.tune("pmg_auto_rows", 100)
    c.tune("pmg_auto_first_rows", 100)
    info = c.solve(rtol=1e-11, raise_on_noconv=False)
    assert info.converged == 1, (info.method_used, info.iters, info.relres)
    A = _csr(c, capi, nd)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    named = c.solve(method=capi.SOLVER_PMG, rtol=1e-11, raise_on_noconv=False)   # by name: its own outcome, reported as it is
    assert named.method_used == capi.SOLVER_PMG
    if named.converged:
        assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    c.close()


def test_coarse_solves_that_stop_at_their_budget_still_serve(env):
    """a coarse budget far too small for the coarse tolerance (what a large 2-D P1 level does to the default budget): the corrections are rougher, the outer
    iteration takes longer -- and still ends at the LU solution; only coarse solves that get NOWHERE make the solver give up"""
    import scipy.sparse.linalg as spl

    capi, meshgen, _ = env
    c, nd, bd, coords = _problem(capi, meshgen, 2, 60, -capi.laplacian() + capi.advection([2.0, 1.0]) + capi.reaction(1.0), "zero")
    full = c.solve(method=capi.SOLVER_PMG, rtol=1e-10)
    u = c.solution()
    A = _csr(c, capi, nd)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(u - ref) <= 1e-7 * np.linalg.norm(ref)
    for budget in (40, 12):
        c.tune("pmg_inner_maxit", budget)
        short = c.solve(method=capi.SOLVER_PMG, rtol=1e-10, maxit=300, raise_on_noconv=False)
        assert short.method_used == capi.SOLVER_PMG
        if short.converged:   # (rough corrections inside a BiCGStab that is not flexible: slower, and either the solution or an honest "not converged")
            assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
        else:
            assert short.relres > 1e-10
    c.close()


@pytest.mark.parametrize("dim,nx", [(2, 20), (3, 6)])
def test_parabolic_stepper_through_the_two_level_solver(env, dim, nx):
    """fdapde_solve_parabolic with FDAPDE_SOLVER_PMG: K = M / dt + A, the coarse operator the P1 assembly of the same terms + M1 / dt, every step warm-started
    from the previous column -- against the Jacobi-preconditioned stepper and, step by step, against LU on the reference's row-zeroed K
    (fem_linear_parabolic_solver.h:37-72); another dt re-assembles the coarse operator"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl

    capi, meshgen, _ = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    _, bd, coords = c.dofs_get()
    qn = c.quadrature_nodes()
    b = [1.0, -0.5] if dim == 2 else [1.0, 0.5, 0.25]
    c.set_operator(capi.dt() - capi.laplacian() + capi.advection(b))
    c.tune("dense_rows", 0)
    u0 = np.cos(coords[:, 0]) * coords[:, 1]
    for times in (np.linspace(0.0, 0.2, 6), np.linspace(0.0, 0.05, 4)):
        c.set_forcing(np.stack([np.sin(2.0 * qn[:, 0]) * (1.0 + t) for t in times], axis=1))
        c.init()
        G = np.stack([0.1 * np.sin(coords[:, 0] + 3.0 * t) for t in times], axis=1)
        sol, info = c.solve_parabolic(times, u0, G, method=capi.SOLVER_PMG, rtol=1e-11)
        assert info.converged == 1 and info.method_used == capi.SOLVER_PMG and info.iters <= 40 * (times.size - 1)
        ref, kinfo = c.solve_parabolic(times, u0, G, rtol=1e-12)
        assert kinfo.method_used != capi.SOLVER_PMG
        assert np.abs(sol - ref).max() <= 1e-8 * np.abs(ref).max()
        rp, ci = c.pattern_get()
        A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
        M = sp.csr_matrix((c.matrix_values(capi.MAT_MASS), ci, rp), shape=(nd, nd))
        F = c.force(ncols=times.size).reshape(times.size, nd).T
        dt_ = times[1] - times[0]
        K = (M / dt_ + A).tolil()
        bidx = np.nonzero(bd)[0]
        K[bidx, :] = 0.0
        K[bidx, bidx] = 1.0
        lu = spl.splu(sp.csc_matrix(K))
        u = u0.copy()
        assert np.array_equal(sol[:, 0], u0)
        for i in range(times.size - 1):
            rhs = (M / dt_) @ u + F[:, i + 1]
            rhs[bidx] = G[bidx, i + 1]
            u = lu.solve(rhs)
            assert np.linalg.norm(sol[:, i + 1] - u) <= 1e-8 * np.linalg.norm(u), i
    c.close()


@pytest.mark.parametrize("dim,nx", [(2, 24), (3, 7)])
def test_coefficient_fields_reach_the_coarse_level_as_cell_means(env, dim, nx):
    """space-varying reaction, advection and diffusion (sampled at the order-2 rule's quadrature nodes): the coarse operator is assembled from their cell means --
    a preconditioner needs the coarse operator only approximately -- and the solve still ends at the LU solution in two dozen iterations"""
    import scipy.sparse.linalg as spl

    capi, meshgen, _ = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    _, bd, coords = c.dofs_get()
    qn = c.quadrature_nodes()
    nq = qn.shape[0]
    creact = 1.0 + 5.0 * qn[:, 0] * qn[:, 1]
    badv = np.stack([1.0 + qn[:, 1], -0.5 + qn[:, 0]] + ([0.3 * np.ones(nq)] if dim == 3 else []), axis=1)
    K = np.zeros((nq, dim * dim))
    for a in range(dim):
        K[:, a * dim + a] = 1.0 + 0.5 * np.sin(2.0 * qn[:, a]) ** 2
    K[:, 1] = K[:, dim] = 0.2 * qn[:, 0]
    c.set_forcing(np.cos(qn[:, 0]) + 1.0)
    c.set_dirichlet(0.2 * coords[:, 0])
    for op in (-capi.laplacian() + capi.reaction_field(creact), -capi.diffusion_field(K) + capi.advection_field(badv) + capi.reaction_field(creact)):
        c.set_operator(op)
        c.init()
        info = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)
        assert info.converged == 1 and info.method_used == capi.SOLVER_PMG and info.iters <= 45, info.iters
        A = _csr(c, capi, nd)
        ref = spl.spsolve(A.tocsc(), c.force())
        assert np.linalg.norm(c.solution() - ref) <= 1e-8 * np.linalg.norm(ref)
    c.close()


@pytest.mark.parametrize("dim,nx", [(2, 30), (3, 7)])
def test_partial_boundary_masks_keep_the_coarse_space_inside_the_fine_one(env, dim, nx):
    """Dirichlet data on a PART of the boundary nodes.  In 2-D the reference constrains every edge DOF of a geometric boundary edge whatever its end nodes are
    (triangulation.h:150-193), so the coarse level must constrain those end nodes too -- before it did, such masks took 100 - 300 outer iterations
    (tools/fuzz_pmg.py, seed 12); and a mask without Dirichlet data leaves both levels unconstrained.  Both forms of the fine operator (knob pmg_blocked)."""
    import scipy.sparse.linalg as spl

    capi, meshgen, _ = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    rng = np.random.default_rng(4)
    keep = ((rng.uniform(0, 1, bnd.shape[0]) < 0.3) & (bnd != 0)).astype(bnd.dtype)
    assert 0 < keep.sum() < (bnd != 0).sum()
    for with_data in (True, False):
        for blocked in (1, 0):
            c = capi.Context(0)
            c.mesh_upload(nodes, cells, keep)
            nd = c.dofs_build(2)
            _, _, coords = c.dofs_get()
            c.tune("pmg_blocked", blocked)
            c.set_operator(-capi.laplacian() + capi.advection(np.array([2.0, -1.0, 0.5][:dim])) + capi.reaction(0.7))
            c.set_forcing(1.0 + c.quadrature_nodes()[:, 0])
            if with_data:
                c.set_dirichlet(0.3 + coords @ np.array([0.5, -0.2, 0.1][:dim]))
            c.init()
            info = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)
            u = c.solution()
            ref = spl.spsolve(_csr(c, capi, nd).tocsc(), c.force())
            assert info.converged == 1 and info.method_used == capi.SOLVER_PMG
            assert info.iters <= 45, (with_data, blocked, info.iters)
            assert np.linalg.norm(u - ref) <= 1e-8 * np.linalg.norm(ref)
            c.close()


def test_restarted_cycles_and_the_earlier_forms_reach_the_same_solution(env):
    """The flexible GMRES restarted every 5 vectors (knob pmg_restart: the path a hard system takes after 50), the additive preconditioner inside it (pmg_smooth 0)
    and BiCGStab around that (pmg_outer 1: the round's first form): all of them against SuperLU, and two runs of the default form bit for bit."""
    import scipy.sparse.linalg as spl

    capi, meshgen, workloads = env
    c, nd, _, _ = _problem(capi, meshgen, 3, 9, workloads.c5_operator(capi), "data")
    ref = None
    its = {}
    for name, knobs in (("default", {}), ("restart5", {"pmg_restart": 5}), ("additive", {"pmg_smooth": 0}), ("bicgstab", {"pmg_outer": 1, "pmg_inner_tol_exp": 2})):
        for k, v in {"pmg_restart": 50, "pmg_smooth": 1, "pmg_outer": 0, "pmg_inner_tol_exp": 1, **knobs}.items():
            c.tune(k, v)
        info = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)
        u = c.solution()
        if ref is None:
            ref = spl.spsolve(_csr(c, capi, nd).tocsc(), c.force())
            info2 = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)
            assert np.array_equal(u, c.solution()) and info2.iters == info.iters
        assert info.converged == 1 and info.method_used == capi.SOLVER_PMG, name
        assert np.linalg.norm(u - ref) <= 1e-8 * np.linalg.norm(ref), name
        its[name] = info.iters
    assert its["default"] <= 25 and its["default"] <= its["restart5"] <= 80 and its["additive"] > its["default"], its
    c.close()


def test_a_repeated_init_of_the_same_operator_keeps_what_was_derived_from_the_matrix(env):
    """fdapde_init again with the same operator (a new forcing): the row-owner sweep reproduces the stiffness matrix bit for bit, so the coarse operator, the blocked-ELL
    fill and the damping stay (the matrix epoch does not move); a new operator, or fdapde_init with a scatter form of the assembly, moves it -- either way the
    answers are those of sparse LU."""
    import scipy.sparse.linalg as spl

    capi, meshgen, workloads = env
    c, nd, _, _ = _problem(capi, meshgen, 3, 9, workloads.c5_operator(capi), "data")
    qn = c.quadrature_nodes()
    rng = np.random.default_rng(8)
    stiff0 = None
    for step, what in enumerate(("first", "same operator", "same operator", "scatter assembly", "new operator", "same operator")):
        if what == "new operator":
            c.set_operator(-capi.laplacian() + capi.advection(np.array([0.3, -2.0, 1.0])) + capi.reaction(2.5))
        c.set_forcing(rng.standard_normal(qn.shape[0]))
        c.init(assembly=capi.ASSEMBLY_ATOMIC) if what == "scatter assembly" else c.init()
        info = c.solve(method=capi.SOLVER_PMG, rtol=1e-11)
        u = c.solution()
        A = _csr(c, capi, nd)
        if what == "first":
            stiff0 = A.data.copy()
        elif what == "same operator" and step < 3:
            assert np.array_equal(A.data, stiff0)
        ref = spl.spsolve(A.tocsc(), c.force())
        assert info.converged == 1 and info.iters <= 25, (what, info.iters)
        assert np.linalg.norm(u - ref) <= 1e-8 * np.linalg.norm(ref), what
    c.close()
