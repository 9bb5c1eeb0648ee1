"""Symmetric but NOT positive definite operators (Helmholtz-type: -Lap u - k^2 u with k^2 above the first Dirichlet eigenvalue).  The
reference's SparseLU solves them like any other system (fem_linear_elliptic_solver.h:38-47, utils/symbols.h:133-160).  Here a symmetric
operator selects CG, whose breakdown (p.Ap <= 0) is detected inside the launch; with the method left open (FDAPDE_SOLVER_AUTO) the solve is
then repeated with BiCGStab -- elliptic solve, parabolic stepper and the factor-once handle alike -- and must give the LU solution; with the
method pinned to CG the breakdown is reported (success = false), never a wrong answer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi, meshgen

    assert capi.load().fdapde_device_count() >= 1
    return capi, meshgen


def _csr(c, capi, which, nd):
    import scipy.sparse as sp

    rp, ci = c.pattern_get()
    return sp.csr_matrix((c.matrix_values(which), ci, rp), shape=(nd, nd))


def _setup(capi, meshgen, dim, nx, order, creact):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    _, bdofs, coords = c.dofs_get()
    c.set_operator(-capi.laplacian() + capi.reaction(creact))
    c.set_forcing(f(c.quadrature_nodes()))
    return c, nd, bdofs, coords


@pytest.mark.parametrize("dim,nx,order,creact", [(2, 32, 1, -100.0), (2, 24, 2, -40.0), (3, 10, 1, -300.0), (3, 5, 2, -60.0)])
def test_elliptic_solve_of_an_indefinite_symmetric_operator(env, dim, nx, order, creact):
    import scipy.sparse.linalg as spl

    capi, meshgen = env
    c, nd, bdofs, coords = _setup(capi, meshgen, dim, nx, order, creact)
    c.set_dirichlet(0.2 * coords[:, 0])
    c.init()
    info = c.solve(rtol=1e-11, raise_on_noconv=False)
    A = _csr(c, capi, capi.MAT_STIFF, nd)   # (after the solve: Dirichlet rows zeroed, unit diagonal -- the reference's system)
    inter = np.flatnonzero(bdofs == 0)
    ev = np.linalg.eigvalsh(A[inter][:, inter].toarray())
    assert ev.min() < 0.0 < ev.max() and np.abs(ev).min() > 1e-5   # indefinite, not singular: what this test is about
    # CG breaks down, BiCGStab takes over -- and where it has not converged within what an inversion costs (rent or buy: 1 089 DOFs ~ 550 iterations), the
    # direct stage answers
    assert info.converged == 1 and info.method_used in (capi.SOLVER_BICGSTAB, capi.SOLVER_DENSE)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    again = c.solve(rtol=1e-11, raise_on_noconv=False)   # (the context remembers the breakdown until the matrix is assembled again: no second CG attempt)
    assert again.converged == 1 and again.method_used == info.method_used and again.iters == info.iters
    c.tune("dense_rows", 0)   # BiCGStab on its own
    alone = c.solve(rtol=1e-11, raise_on_noconv=False)
    assert alone.converged == 1 and alone.method_used == capi.SOLVER_BICGSTAB
    assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    # the method pinned to CG: the breakdown is reported, no answer is passed off as a solution
    info_cg = c.solve(method=capi.SOLVER_CG_FUSED, rtol=1e-11, raise_on_noconv=False)
    assert info_cg.converged == 0
    c.close()


def test_handle_and_stepper_on_an_indefinite_symmetric_operator(env):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl

    capi, meshgen = env
    c, nd, bdofs, coords = _setup(capi, meshgen, 2, 24, 1, -150.0)
    c.init()
    A = _csr(c, capi, capi.MAT_STIFF, nd)
    ev = np.linalg.eigvalsh(A.toarray())
    assert ev.min() < 0.0 < ev.max() and np.abs(ev).min() > 1e-6
    rng = np.random.default_rng(5)
    B = rng.standard_normal((nd, 3))
    c.lin_compute(capi.MAT_STIFF)   # (a symmetric operator: the handle starts with CG)
    X, info = c.lin_solve(B, rtol=1e-11)
    assert info.converged == 1 and info.method_used == capi.SOLVER_BICGSTAB
    ref = spl.spsolve(A.tocsc(), B)
    assert np.linalg.norm(X - ref) <= 1e-7 * np.linalg.norm(ref)
    x1, info1 = c.lin_solve(B[:, 0], rtol=1e-11)   # the handle stays usable, single columns too
    assert info1.converged == 1 and np.linalg.norm(x1 - ref[:, 0]) <= 1e-7 * np.linalg.norm(ref[:, 0])
    c.lin_compute(capi.MAT_STIFF)
    with pytest.raises(capi.FdapdeError):
        c.lin_solve(B, method=capi.SOLVER_CG_FUSED)
    # implicit Euler with M / dt + A indefinite (dt = 0.02: 50 M - 150 M + K): two steps against scipy
    M = _csr(c, capi, capi.MAT_MASS, nd)
    times = np.array([0.0, 0.02, 0.04])
    u0 = np.sin(np.pi * coords[:, 0]) * np.sin(np.pi * coords[:, 1])
    fq = c.quadrature_nodes()[:, 0]
    c.set_forcing(np.tile(fq[:, None], (1, 3)))
    c.init()
    U, pinfo = c.solve_parabolic(times, u0, rtol=1e-11)
    assert pinfo.converged == 1
    b = c.force(3).reshape(3, nd)   # (one column per time point)
    K = (M / 0.02 + A).tocsc()
    u = u0.copy()
    for k in (1, 2):
        u = spl.spsolve(K, M @ u / 0.02 + b[k])
        assert np.linalg.norm(U[:, k] - u) <= 1e-7 * np.linalg.norm(u)
    c.close()


@pytest.mark.parametrize("ncols", [4, 9])
def test_batched_columns_of_an_indefinite_handle_fall_back_too(env, ncols):
    """four or more columns against a system that does not run its columns in one launch take the batched multi-RHS CG
    (kernels_multirhs.h); its breakdown must lead to the BiCGStab retry like the column-by-column paths -- and an in-place solve (x = b's
    own storage) must survive the retry."""
    import ctypes as C

    import scipy.sparse.linalg as spl

    capi, meshgen = env
    c, nd, bdofs, coords = _setup(capi, meshgen, 2, 24, 1, -150.0)
    c.init()
    A = _csr(c, capi, capi.MAT_STIFF, nd)
    rng = np.random.default_rng(11)
    B = rng.standard_normal((nd, ncols))
    ref = spl.spsolve(A.tocsc(), B)
    c.tune("persist", 0)   # (no single-launch solver: the batched path takes the groups of 8 / 4 columns)
    c.tune("dense_rows", 0)   # (... and no dense inverse: the time the broken-down CG attempt took would already pay for one)
    c.lin_compute(capi.MAT_STIFF)
    X, info = c.lin_solve(B, rtol=1e-11)
    assert info.converged == 1 and info.method_used == capi.SOLVER_BICGSTAB
    assert np.linalg.norm(X - ref) <= 1e-7 * np.linalg.norm(ref)
    # in place through the C ABI: x and b the same buffer
    c.lin_compute(capi.MAT_STIFF)
    buf = np.asfortranarray(B.copy())
    opt = capi.Options(method=capi.SOLVER_AUTO, maxit=0, rtol=1e-11, assembly=0, check_every=0, time_spmv=0)
    out = capi.Info()
    ptr = buf.ctypes.data_as(C.POINTER(C.c_double))   # (column-major n_dofs x ncols, as the ABI wants it)
    rc = capi.load().fdapde_lin_solve(c._ctx, C.byref(opt), ptr, ncols, ptr, C.byref(out))
    assert rc == 0 and out.converged == 1
    assert np.linalg.norm(buf - ref) <= 1e-7 * np.linalg.norm(ref)
    c.close()


@pytest.mark.parametrize("dim,nx,bmag", [(2, 32, 1000.0), (3, 10, 1000.0)])
def test_bicgstab_restarts_after_a_breakdown(env, dim, nx, bmag):
    """advection-dominated operators (cell Peclet 15 - 50: under-resolved, but the reference's LU solves what it is given): BiCGStab's
    recurrences break down (rho, r0.v or omega exactly 0) long before convergence; the solve is started again from the iterate reached
    (new shadow residual, the stop rule still relative to the original right-hand side) until it converges -- knob bicg_restart 0: the first
    breakdown ends it, as before"""
    import scipy.sparse.linalg as spl

    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    _, _, coords = c.dofs_get()
    c.set_operator(-capi.laplacian() + capi.advection(bmag * np.array([1.0, 0.5, 0.25])[:dim]))
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(0.2 * coords[:, 0])
    c.init()
    c.tune("bicg_restart", 0)
    first = c.solve(rtol=1e-10, raise_on_noconv=False)
    c.tune("bicg_restart", 1)
    info = c.solve(rtol=1e-10, raise_on_noconv=False)
    assert info.converged == 1 and info.relres <= 1e-10
    # (whether the run WITHOUT restarts breaks down depends on the last bits of the matrix -- BiCGStab's path on these operators is chaotic;
    #  where it does, the restarts are what finished the solve)
    if first.converged == 0:
        assert info.iters > first.iters
    A = _csr(c, capi, capi.MAT_STIFF, nd)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    c.close()


def _advection_case(capi, meshgen, dim, nx, peclet):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    _, _, coords = c.dofs_get()
    d = np.array([1.0, 0.5, 0.25])[:dim]
    c.set_operator(-capi.laplacian() + capi.advection((2.0 * peclet * nx / np.linalg.norm(d)) * d))   # cell Peclet number |b| h / 2
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(0.2 * coords[:, 0])
    c.init()
    return c, nd


@pytest.mark.parametrize("dim,nx,peclet", [(2, 32, 150.0), (2, 32, 500.0), (2, 64, 1000.0), (2, 128, 150.0), (3, 10, 1000.0)])
def test_auto_ends_in_gmres_where_bicgstab_gives_up(env, dim, nx, peclet):
    """cell Peclet numbers of 150 - 1000: BiCGStab's recurrences stall or blow up, restarts included; the reference's LU solves
    what it is given (fem_linear_elliptic_solver.h:38-47).  With the method left open the solve ends in restarted GMRES(50) on the same scaled
    system and must reach the LU solution; info.method_used names the stage that produced the answer; knob auto_gmres 0 shows what it was before."""
    import scipy.sparse.linalg as spl

    capi, meshgen = env
    c, nd = _advection_case(capi, meshgen, dim, nx, peclet)
    first = c.solve(rtol=1e-10, raise_on_noconv=False)   # (round 6: systems of up to 4096 DOFs get the direct stage in front of GMRES -- tests/test_gpu_dense.py)
    assert first.converged == 1 and first.method_used in (capi.SOLVER_DENSE, capi.SOLVER_GMRES, capi.SOLVER_BICGSTAB)
    c.tune("dense_rows", 0)   # ... here: the GMRES stage itself
    info = c.solve(rtol=1e-10, raise_on_noconv=False)
    assert info.converged == 1 and info.relres <= 1e-10
    assert info.method_used in (capi.SOLVER_GMRES, capi.SOLVER_BICGSTAB)
    A = _csr(c, capi, capi.MAT_STIFF, nd)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    if info.method_used == capi.SOLVER_GMRES:   # (the tail did the work: without it the call reports failure, as before)
        c.tune("auto_gmres", 0)
        before = c.solve(rtol=1e-10, raise_on_noconv=False)
        assert before.converged == 0 and before.method_used == capi.SOLVER_BICGSTAB
        c.tune("auto_gmres", 1)
    # a method named explicitly is never replaced: BiCGStab pinned reports its failure / GMRES pinned solves it by itself
    g = c.solve(method=capi.SOLVER_GMRES, rtol=1e-10, raise_on_noconv=False)
    assert g.converged == 1 and g.method_used == capi.SOLVER_GMRES
    assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    c.close()


def test_gmres_on_a_plain_elliptic_problem(env):
    """GMRES named explicitly on a symmetric positive definite system: the same solution as CG (restart lengths 50 and 20)"""
    capi, meshgen = env
    c, nd, bdofs, coords = _setup(capi, meshgen, 3, 8, 1, 1.0)
    c.set_dirichlet(0.2 * coords[:, 0])
    c.init()
    cg = c.solve(rtol=1e-11)
    u_cg = c.solution().copy()
    for m in (50, 20):
        c.tune("gmres_m", m)
        g = c.solve(method=capi.SOLVER_GMRES, rtol=1e-11)
        assert g.converged == 1 and g.method_used == capi.SOLVER_GMRES and g.relres <= 1e-11
        assert np.linalg.norm(c.solution() - u_cg) <= 1e-9 * np.linalg.norm(u_cg)
    assert cg.method_used == capi.SOLVER_CG_FUSED
    c.close()


@pytest.mark.parametrize("dim,nx,k2", [(2, 32, 5000.0), (2, 64, 5000.0), (3, 10, 3000.0)])
def test_strongly_indefinite_symmetric_operators(env, dim, nx, k2):
    """-Lap u - k^2 u with hundreds of negative eigenvalues (269 of 1 089, 336 of 4 225, 595 of 1 331): the open method must end at the LU solution"""
    import scipy.sparse.linalg as spl

    capi, meshgen = env
    c, nd, bdofs, coords = _setup(capi, meshgen, dim, nx, 1, -k2)
    c.set_dirichlet(0.2 * coords[:, 0])
    c.init()
    info = c.solve(rtol=1e-10, raise_on_noconv=False)
    assert info.converged == 1
    A = _csr(c, capi, capi.MAT_STIFF, nd)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(c.solution() - ref) <= 1e-6 * np.linalg.norm(ref)
    c.close()


def _mid_size_advection(capi, meshgen, nx=300, peclet=3.0, with_dt=False):
    """90 601 DOFs (vec_grid 354 > the 256 stripes of GMRES's dot partials): the multi-workgroup partial path of kernels_gmres.h"""
    nodes, cells, bnd = meshgen.unit_square(nx)
    _, f = meshgen.manufactured(2)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    _, _, coords = c.dofs_get()
    d = np.array([1.0, 0.5])
    op = -capi.laplacian() + capi.advection((2.0 * peclet * nx / np.linalg.norm(d)) * d)
    c.set_operator(capi.dt() + op if with_dt else op)
    return c, nd, coords, f


def test_gmres_above_65536_dofs_through_solve(env):
    """the |w|^2 partials of k_gm_axpy / k_gm_hess are one per workgroup of the vector kernels (up to 1024), not one per stripe:
    a system above 65 536 DOFs wrote past gm_part.  Explicit GMRES and the open method on 90 601 DOFs against LU."""
    import scipy.sparse.linalg as spl

    capi, meshgen = env
    c, nd, coords, f = _mid_size_advection(capi, meshgen)
    assert nd > 65536
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(0.2 * coords[:, 0])
    c.init()
    g = c.solve(method=capi.SOLVER_GMRES, rtol=1e-10, raise_on_noconv=False)
    assert g.converged == 1 and g.method_used == capi.SOLVER_GMRES and g.relres <= 1e-10
    A = _csr(c, capi, capi.MAT_STIFF, nd)
    ref = spl.spsolve(A.tocsc(), c.force())
    assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    a = c.solve(rtol=1e-10, raise_on_noconv=False)   # the open method on the same system, after a GMRES run left its words in ctl
    assert a.converged == 1
    assert np.linalg.norm(c.solution() - ref) <= 1e-7 * np.linalg.norm(ref)
    c.close()


def test_gmres_above_65536_dofs_through_the_handle_and_the_stepper(env):
    """the GMRES stage inside fdapde_lin_solve and fdapde_solve_parabolic (named explicitly, 90 601 DOFs), each against scipy's LU"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl

    capi, meshgen = env
    c, nd, coords, f = _mid_size_advection(capi, meshgen, with_dt=True)
    times = np.linspace(0.0, 0.02, 3)
    fq = f(c.quadrature_nodes())
    c.set_forcing(np.stack([fq] * times.size, axis=1))
    c.init()
    A = _csr(c, capi, capi.MAT_STIFF, nd)
    M = _csr(c, capi, capi.MAT_MASS, nd)
    # handle: two columns against the assembled operator (no Dirichlet reduction in the handle: a mass shift of the size of an implicit Euler step's keeps restarted GMRES away from stagnation)
    vals = c.matrix_values(capi.MAT_STIFF) + 5e5 * c.matrix_values(capi.MAT_MASS)
    c.lin_compute(values=vals, symmetric=False)
    rng = np.random.default_rng(5)
    b = rng.standard_normal((nd, 2))
    x, info = c.lin_solve(b, method=capi.SOLVER_GMRES, rtol=1e-10)
    assert info.converged == 1 and info.method_used == capi.SOLVER_GMRES
    K = (A + 5e5 * M).tocsc()
    lu = spl.splu(K)
    for j in range(2):
        ref = lu.solve(b[:, j])
        assert np.linalg.norm(x[:, j] - ref) <= 1e-7 * np.linalg.norm(ref)
    F = c.force(ncols=times.size).reshape(times.size, nd).T   # (before the stepper writes the Dirichlet rows)
    # stepper: implicit Euler with GMRES named, against LU stepping of the same matrices (fem_linear_parabolic_solver.h:56-68)
    u0 = np.sin(np.pi * coords[:, 0]) * np.sin(np.pi * coords[:, 1])
    gcols = np.zeros((nd, times.size))
    sol, pinfo = c.solve_parabolic(times, u0, dirichlet=gcols, method=capi.SOLVER_GMRES, rtol=1e-11)
    assert pinfo.converged == 1
    dt_ = times[1] - times[0]
    _, bd, _ = c.dofs_get()
    Kp = (M / dt_ + A).tolil()
    bidx = np.nonzero(bd)[0]
    Kp[bidx, :] = 0.0
    Kp[bidx, bidx] = 1.0
    lup = spl.splu(sp.csc_matrix(Kp))
    u = u0.copy()
    for i in range(times.size - 1):
        rhs = (M / dt_) @ u + F[:, i + 1]
        rhs[bidx] = 0.0
        u = lup.solve(rhs)
        assert np.linalg.norm(sol[:, i + 1] - u) <= 1e-6 * np.linalg.norm(u)
    c.close()
