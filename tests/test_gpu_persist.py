"""The single-launch CG of small systems (kernels_persist.h: matrix resident in LDS, granule hand-offs between workgroups) against
the multi-launch path it replaces and against the oracle: same recurrence, so the same iteration counts (the dot products are
summed in another order: +-1 iteration at rtol 1e-10) and the same solution to rounding; every hand-off shape is covered -- one
workgroup (no exchange), a few, one per CU with rows per thread 1 ... 8, 2-D and 3-D, P1 and P2, with and without Dirichlet
data, warm starts (parabolic stepping) and the factor-once handle."""
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


def _problem(capi, meshgen, dim, nx, order, dirichlet=True):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    u_exact, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd if dirichlet else np.zeros_like(bnd))
    nd = c.dofs_build(order)
    _, _, coords = c.dofs_get()
    c.set_operator(-capi.laplacian() + (capi.reaction(0.0) if dirichlet else capi.reaction(1.0)))
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(0.25 * coords[:, 0] if dirichlet else None)
    c.init()
    return c, nd


@pytest.mark.parametrize("dim,nx,order,dirichlet", [
    (2, 20, 1, True),      # 441 DOFs: one workgroup, no exchange
    (2, 60, 1, True),      # 3 721 DOFs: two workgroups
    (2, 60, 2, True),      # 14 641 DOFs, P2 rows
    (2, 150, 1, False),    # 22 801 DOFs, no Dirichlet DOF
    (3, 30, 1, True),      # 29 791 DOFs: 3-D rows (15 entries), larger import lists
    (3, 12, 2, True),      # 15 625 DOFs, 3-D P2 rows (up to 64 entries)
    (2, 400, 1, True),     # 160 801 DOFs: ~80 workgroups
    (3, 64, 1, True),      # 274 625 DOFs: 135 workgroups
    (2, 708, 1, True),     # C2: one workgroup per CU, 4 rows per thread
    (2, 1000, 1, True),    # 1 002 001 DOFs: 8 rows per thread, part of the matrix streams from the caches
])
def test_persistent_path_matches_multi_launch_path(env, dim, nx, order, dirichlet):
    capi, meshgen = env
    c, nd = _problem(capi, meshgen, dim, nx, order, dirichlet)
    c.tune("persist", 0)
    i0 = c.solve(rtol=1e-10)
    u0 = c.solution()
    assert i0.persistent == 0 and i0.converged == 1
    c.tune("persist", 1)
    i1 = c.solve(rtol=1e-10)
    u1 = c.solution()
    assert i1.persistent == 1, "the system qualifies: the single-launch path must have run"
    assert i1.converged == 1 and i1.method_used == capi.SOLVER_CG_FUSED and i1.relres <= 1e-10
    assert abs(i1.iters - i0.iters) <= max(1, i0.iters // 200), (i1.iters, i0.iters)
    assert np.linalg.norm(u1 - u0) <= 1e-9 * np.linalg.norm(u0)
    # a second launch on the same context (boards re-zeroed, tags restart): identical bits
    i2 = c.solve(rtol=1e-10)
    assert i2.iters == i1.iters and np.array_equal(c.solution(), u1)
    # maxit reached: reported, not hung
    i3 = c.solve(rtol=1e-10, maxit=5, raise_on_noconv=False)
    assert i3.persistent == 1 and i3.converged == 0 and i3.iters == 5
    c.close()


@pytest.mark.parametrize("dim,nx,order,dirichlet", [
    (2, 20, 1, True),      # one workgroup: every pair of rows stored once, no exchange
    (2, 60, 2, True),      # P2 rows, a few workgroups
    (2, 150, 1, False),    # no Dirichlet DOF
    (3, 30, 1, True),      # 3-D rows, import lists
    (3, 12, 2, True),      # 3-D P2 rows (up to 64 entries): long rows, many pairs per accumulator slot
    (3, 64, 1, True),      # plain blocks stream, symmetric blocks are resident: the automatic choice takes them
    (2, 1000, 1, True),    # 8 rows per thread (what the automatic choice picks the symmetric storage for)
    (3, 105, 1, True),     # 16 rows per thread, blocks stream
])
def test_symmetric_storage_matches_plain_storage(env, dim, nx, order, dirichlet):
    """tune persist_sym 1: in-block pairs stored once, transposed products through 64-bit fixed-point LDS accumulators
    (kernels_persist.h SYM).  Same recurrence and the same operator to rounding: the same iteration count (+-1) and solution as
    the plain storage; integer accumulation makes the result independent of the order the wavefronts arrive in: bitwise
    reproducible from launch to launch; the streamed bytes shrink."""
    capi, meshgen = env
    c, nd = _problem(capi, meshgen, dim, nx, order, dirichlet)
    c.tune("persist_sym", 0)
    i0 = c.solve(rtol=1e-10)
    u0 = c.solution()
    bytes_plain = c.solver_layout(dirichlet)[2]
    assert i0.persistent == 1 and i0.converged == 1
    c.tune("persist_sym", 1)
    i1 = c.solve(rtol=1e-10)
    u1 = c.solution()
    bytes_sym = c.solver_layout(dirichlet)[2]
    assert i1.persistent == 1 and i1.converged == 1 and i1.relres <= 1e-10
    assert abs(i1.iters - i0.iters) <= max(1, i0.iters // 200), (i1.iters, i0.iters)
    assert np.linalg.norm(u1 - u0) <= 1e-9 * np.linalg.norm(u0)
    assert bytes_sym < 0.8 * bytes_plain, (bytes_sym, bytes_plain)
    for _ in range(2):
        i2 = c.solve(rtol=1e-10)
        assert i2.iters == i1.iters and np.array_equal(c.solution(), u1), "integer accumulation: identical bits on every launch"
    c.tune("persist_sym", 2)   # automatic: symmetric where the plain blocks would stream -- for workgroups of <= 2048 rows only if the
    i3 = c.solve(rtol=1e-10)   # symmetric blocks are then resident (3-D nx 64: 42 MB plain stream against 29 MB in LDS)
    want_sym = (dim, nx) in ((2, 1000), (3, 105), (3, 64))
    assert (c.solver_layout(dirichlet)[2] == bytes_sym) == want_sym
    assert i3.converged == 1
    c.close()


def test_symmetric_storage_with_coefficients_of_very_different_size(env):
    """a reaction coefficient that varies in space by six orders of magnitude: after the Jacobi scaling the rows range from
    stiffness-dominated to mass-dominated, and search directions whose entries differ by orders of magnitude inside one workgroup
    share one accumulator scale (chosen from the block's largest |p|)"""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_cube(40)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    q = c.quadrature_nodes()
    c.set_operator(-capi.laplacian() + capi.reaction_field(10.0 ** (6.0 * q[:, 0] - 2.0)))
    c.set_forcing(np.sin(3.0 * q[:, 1]) + 2.0)
    c.set_dirichlet(np.zeros(nd))
    c.init()
    sols = {}
    for sym in (0, 1):
        c.tune("persist_sym", sym)
        i = c.solve(rtol=1e-11)
        assert i.persistent == 1 and i.converged == 1
        sols[sym] = (c.solution().copy(), i.iters)
    assert abs(sols[1][1] - sols[0][1]) <= max(1, sols[0][1] // 100), (sols[1][1], sols[0][1])
    assert np.linalg.norm(sols[1][0] - sols[0][0]) <= 1e-9 * np.linalg.norm(sols[0][0])
    c.close()


@pytest.mark.parametrize("mesh,order,sym", [("unit_square", 1, 0), ("unit_square", 1, 1), ("unit_square", 2, 0), ("unit_square", 2, 1),
                                            ("unit_sphere", 2, 1), ("unit_square_64", 2, 1)])
def test_persistent_path_against_the_oracle(env, oracle, mesh_loader, mesh, order, sym):
    """both storages of the single launch against the ORACLE's direct solve (not against another HIP path): unit_square P1 = two
    workgroups, P2 = 14 161 DOFs in seven workgroups with imports and exports on every one; sym = 1 forces the symmetric-storage
    instantiation (what C3's headline runs on) where the automatic choice would keep the plain one"""
    capi, _ = env
    m = mesh_loader(mesh)
    c = capi.Context(0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(order)
    _, _, coords = c.dofs_get()
    nq = c.sizes()["n_quadrature"]
    fq = np.ones(nq * m.n_cells)
    g = coords[:, 0] * coords[:, 1]
    c.set_operator(-capi.laplacian())
    c.set_forcing(fq)
    c.set_dirichlet(g)
    c.init()
    c.tune("persist_sym", sym)
    info = c.solve(rtol=1e-11)
    assert info.persistent == 1 and info.converged == 1
    lay = c.solver_layout_kind(True)
    assert lay["sym"] == sym and lay["kind"] in (2, 3)
    if order == 2 and mesh != "unit_sphere":
        assert lay["workgroups"] >= 4, lay
    ref = oracle.pde_init_solve(m, order, -oracle.laplacian(), forcing_q=fq, dirichlet=g)
    assert np.linalg.norm(c.solution() - ref.solution) <= 1e-8 * np.linalg.norm(ref.solution)
    c.close()


@pytest.mark.parametrize("stall_at,sym", [(3, 0), (4, 0), (3, 1), (1, 0)])
def test_hand_off_timeout_mid_solve_falls_back_from_the_same_state(env, stall_at, sym):
    """a workgroup that stops taking part at iteration k (what a peer that is not resident looks like): every other workgroup runs into
    its bounded wait, the launch gives up, and the multi-launch path restarts the solve.  The launch must have left x, r, p, the
    scalars and the iteration counter as it found them -- workgroups that finished their iterations before the others noticed must not
    have stored anything the restart reads (odd and even k: the fall-back's lazy x update goes by the iteration parity)."""
    capi, meshgen = env
    c, nd = _problem(capi, meshgen, 3, 30, 1, True)
    c.tune("persist", 0)
    i0 = c.solve(rtol=1e-10)
    u0 = c.solution()
    c.tune("persist", 1)
    c.tune("persist_sym", sym)
    c.tune("persist_timeout_us", 2000)
    c.tune("persist_debug_stall", stall_at)
    i1 = c.solve(rtol=1e-10)
    assert i1.persistent == 0 and i1.converged == 1, "the stalled launch must have been replaced by the multi-launch path"
    assert i1.iters == i0.iters, (i1.iters, i0.iters)
    # (not the same bits: with the single launch enabled the multi-launch kernels apply the full-pattern scaled matrix, without it the
    #  compact one -- another summation order inside a row)
    assert np.linalg.norm(c.solution() - u0) <= 1e-12 * np.linalg.norm(u0), "the fall-back must start from the state the launch found"
    assert i1.t_solve_ms < 60.0, i1.t_solve_ms   # one bounded wait (2 ms), not a hang
    c.tune("persist_debug_stall", 0)
    c.tune("persist_retry", 1)   # (the context would otherwise stay on the multi-launch path for the next 8 systems)
    i2 = c.solve(rtol=1e-10)
    assert i2.persistent == 1 and i2.converged == 1
    assert np.linalg.norm(c.solution() - u0) <= 1e-9 * np.linalg.norm(u0)
    c.close()


def test_first_solve_of_a_small_system_is_cheap(env, mesh_loader):
    """downstream models solve many small systems: the lazy build of the single-launch layout must not dominate the first solve
    (round 2: 13.5 ms for 587 DOFs -- ~40 device launches and sorts; now built on the host for small systems)"""
    capi, _ = env
    m = mesh_loader("unit_sphere")
    warm = capi.Context(0)   # (the very first context of a process also pays for loading the code objects)
    warm.mesh_upload(m.nodes, m.cells, m.boundary)
    warm.dofs_build(1)
    warm.set_operator(-capi.laplacian())
    warm.set_forcing(np.zeros(4 * m.n_cells))
    warm.set_dirichlet(np.zeros(m.n_nodes))
    warm.init()
    warm.solve(rtol=1e-10)
    c = capi.Context(0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(1)
    _, _, coords = c.dofs_get()
    c.set_operator(-capi.laplacian())
    c.set_forcing(np.zeros(4 * m.n_cells))
    c.set_dirichlet(coords.sum(axis=1))
    c.init()
    i1 = c.solve(rtol=1e-10)
    i2 = c.solve(rtol=1e-10)
    assert i1.persistent == 1 and i2.persistent == 1
    assert i1.t_solve_ms <= 2.0, (i1.t_solve_ms, i2.t_solve_ms)
    assert np.abs(c.solution() - coords.sum(axis=1)).max() < 1e-8
    c.close(), warm.close()


def test_persistent_path_under_parabolic_stepping_and_handle(env):
    """warm-started solves (x0 != 0) and the factor-once handle go through the same solve_run"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(48)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    _, _, coords = c.dofs_get()
    u_exact, f = meshgen.manufactured(2)
    times = np.linspace(0.0, 0.3, 5)
    qn = c.quadrature_nodes()
    c.set_operator(capi.dt() - capi.laplacian())
    c.set_forcing(np.stack([(f(qn) - u_exact(qn)) * np.exp(-t) for t in times], axis=1))
    c.init()
    G = np.stack([u_exact(coords) * np.exp(-t) for t in times], axis=1)
    outs = []
    for knob in (0, 1):
        c.tune("persist", knob)
        sol, info = c.solve_parabolic(times, G[:, 0], G, rtol=1e-11)
        assert info.converged == 1
        outs.append(sol)
    assert np.abs(outs[0] - outs[1]).max() <= 1e-9
    rp, ci = c.pattern_get()
    M = sp.csr_matrix((c.matrix_values(capi.MAT_MASS), ci, rp), shape=(nd, nd))
    b = np.random.default_rng(3).standard_normal(nd)
    c.tune("dense_rows", 0)   # (this file is about the Krylov launches; the dense inverse of small systems: tests/test_gpu_dense.py)
    c.lin_compute(capi.MAT_MASS, symmetric=True)
    x, info = c.lin_solve(b, rtol=1e-12)
    assert info.persistent == 1
    xr = spla.splu(M.tocsc()).solve(b)
    assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr)
    c.close()


@pytest.mark.parametrize("nx", [8, 20])
def test_blocked_ell_spmv_under_every_krylov_method(env, nx):
    """The blocked-ELL SpMV (k_spmv_blocked) inside the multi-launch solves, 3-D P2 rows, NON-homogeneous Dirichlet data: the layout
    leaves the Dirichlet rows out, the vector kernels sweep all n entries -- y must be defined there too (it once kept the lift A g~
    of the same buffer, which broke every method that sums r.r over all rows)."""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(3)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    _, _, coords = c.dofs_get()
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(0.25 * coords[:, 0])
    c.init()
    c.tune("persist", 0)
    ref = {}
    for blocked in (0, 2):
        c.tune("blocked", blocked)
        for m in (capi.SOLVER_CG_FUSED, capi.SOLVER_CG, capi.SOLVER_CG_SR, capi.SOLVER_BICGSTAB):
            info = c.solve(method=m, rtol=1e-11)
            assert info.converged == 1 and info.persistent == 0
            u = c.solution()
            if blocked == 0:
                ref[m] = (info.iters, u)
            else:
                assert abs(info.iters - ref[m][0]) <= max(2, ref[m][0] // 10), (m, info.iters, ref[m][0])
                assert np.linalg.norm(u - ref[m][1]) <= 1e-8 * np.linalg.norm(ref[m][1])
    c.close()


@pytest.mark.parametrize("dim,nx,order", [
    (2, 20, 1),     # one workgroup
    (2, 120, 1),    # several workgroups, blocks resident
    (2, 60, 2),     # P2 rows
    (3, 30, 1),     # 3-D rows, import lists
    (3, 12, 2),     # 3-D P2: rows of 10 ... 60+ entries
    (3, 80, 1),     # 531 441 DOFs: one workgroup per CU, blocks stream
    (2, 1000, 1),   # 1 002 001 DOFs: 8 rows per thread (the most the six register vectors leave room for)
])
def test_single_launch_bicgstab_matches_multi_launch(env, dim, nx, order):
    """non-symmetric operator (advection): the whole Jacobi-BiCGStab as ONE launch (kernels_persist_bicg.h) against the multi-launch
    kernels: same recurrence up to the summation order of the dot products and rho' = r0.s - omega r0.t instead of the explicit r0.r --
    BiCGStab's iteration count moves with rounding, the solutions agree to the solver tolerance"""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    u_exact, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    _, _, coords = c.dofs_get()
    bvec = [1.0, 0.5, 0.25][:dim]
    c.set_operator(-capi.laplacian() + capi.advection(bvec) + capi.reaction(1.0))
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(0.25 * coords[:, 0])
    c.init()
    c.tune("persist_bicg", 0)
    i0 = c.solve(rtol=1e-11)
    u0 = c.solution()
    assert i0.persistent == 0 and i0.converged == 1 and i0.method_used == capi.SOLVER_BICGSTAB
    c.tune("persist_bicg", 1)
    i1 = c.solve(rtol=1e-11)
    u1 = c.solution()
    assert i1.persistent == 1, "the system qualifies: the single-launch BiCGStab must have run"
    assert i1.converged == 1 and i1.method_used == capi.SOLVER_BICGSTAB and i1.relres <= 1e-11
    assert i1.iters <= 1.3 * i0.iters + 5, (i1.iters, i0.iters)
    assert np.linalg.norm(u1 - u0) <= 1e-8 * np.linalg.norm(u0)
    i2 = c.solve(rtol=1e-11)   # deterministic reductions: identical bits from launch to launch
    assert i2.iters == i1.iters and np.array_equal(c.solution(), u1)
    i3 = c.solve(rtol=1e-11, maxit=4, raise_on_noconv=False)
    assert i3.persistent == 1 and i3.converged == 0 and i3.iters == 4
    # a stalled peer: the launch gives up, the multi-launch BiCGStab takes over from the same state
    if c.solver_layout_kind(True)["workgroups"] > 1:
        c.tune("persist_timeout_us", 2000)
        c.tune("persist_debug_stall", 2)
        i4 = c.solve(rtol=1e-11)
        assert i4.persistent == 0 and i4.converged == 1
        assert np.linalg.norm(c.solution() - u0) <= 1e-8 * np.linalg.norm(u0)
    c.close()


@pytest.mark.parametrize("dim,nx,adr", [(3, 80, False), (2, 1000, False), (3, 80, True)])
def test_touching_the_next_entries_during_the_all_gather_changes_no_bit(env, dim, nx, adr):
    """streaming layouts: the idle wavefronts touch the first entry step of the next operator application while the workgroup waits for
    the dot records (knob persist_prefetch).  The touches move data towards the L2 and nothing else: identical iterations and bits."""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    op = -capi.laplacian()
    if adr:
        op = op + capi.advection([1.0, 0.5, 0.25][:dim]) + capi.reaction(1.0)
    c.set_operator(op)
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    out = {}
    for knob in (0, 1):
        c.tune("persist_prefetch", knob)
        i = c.solve(rtol=1e-10)
        assert i.persistent == 1 and i.converged == 1
        assert c.solver_layout_kind(True)["kind"] == 2, "the case is meant to stream its blocks"
        out[knob] = (i.iters, c.solution())
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    c.close()


@pytest.mark.parametrize("dim,nx,order,dirichlet", [(2, 44, 1, False), (3, 12, 1, True), (2, 20, 2, True), (3, 11, 1, False)])
def test_one_workgroup_without_hand_offs_matches_several(env, dim, nx, order, dirichlet):
    """systems of up to 2048 interior rows run as ONE workgroup, which publishes and polls no dot record (knob persist_single_rows; the
    block streams from the L2 where it does not fit the LDS); against the layout of several workgroups the general rule would give"""
    capi, meshgen = env
    c, nd = _problem(capi, meshgen, dim, nx, order, dirichlet)
    c.tune("persist_single_rows", 0)
    i0 = c.solve(rtol=1e-11)
    u0 = c.solution()
    g0 = c.solver_layout_kind(dirichlet)["workgroups"]
    c.tune("persist_single_rows", 2048)
    i1 = c.solve(rtol=1e-11)
    u1 = c.solution()
    assert i0.persistent == 1 and i1.persistent == 1 and i0.converged == 1 and i1.converged == 1
    assert g0 >= 2 and c.solver_layout_kind(dirichlet)["workgroups"] == 1, (g0, c.solver_layout_kind(dirichlet))
    assert abs(i1.iters - i0.iters) <= 2
    assert np.linalg.norm(u1 - u0) <= 1e-9 * np.linalg.norm(u0)
    i2 = c.solve(rtol=1e-11)
    assert i2.iters == i1.iters and np.array_equal(c.solution(), u1)
    c.close()


@pytest.mark.parametrize("dim,nx,order,dirichlet,zero_g", [
    (2, 16, 1, True, True),     # 289 DOFs, homogeneous data: ONE kernel in front of the launch
    (2, 16, 1, True, False),    # non-zero lift: the product A g~ between two launches of that kernel
    (2, 40, 1, True, False),    # 1 681 DOFs: seven workgroups' worth of start-up sums run by one
    (3, 9, 1, True, False),     # 3-D rows
    (2, 14, 2, True, True),     # P2
    (2, 30, 1, False, True),    # no Dirichlet DOF
])
def test_small_front_kernel_gives_the_separate_launches_bits(env, dim, nx, order, dirichlet, zero_g):
    _small_front_case(env, dim, nx, order, dirichlet, zero_g, adr=False)


@pytest.mark.parametrize("dim,nx,order,zero_g", [(2, 16, 1, True), (2, 24, 1, False), (3, 8, 1, False), (2, 12, 2, False)])
def test_small_front_kernel_with_the_single_launch_bicgstab(env, dim, nx, order, zero_g):
    """the same for a non-symmetric operator (advection-diffusion-reaction): k_small_front seeds the shadow residual too, the single-launch BiCGStab
    writes its outcome record and the unscaled solution itself"""
    _small_front_case(env, dim, nx, order, True, zero_g, adr=True)


def _small_front_case(env, dim, nx, order, dirichlet, zero_g, adr):
    """fdapde_solve of a one-workgroup system of at most `small_front_rows` DOFs enqueues flag reset, Jacobi scale, layout fill, lift and the Krylov
    start-up as ONE kernel (k_small_front: the same device functions in the same launch geometry) and the epilogue inside the launch
    (PersistArgs::u_out): same iterations, identical solution bits as the separate launches (knob small_front_rows = 0), from the second solve on
    (the first one builds the layout's column table)"""
    capi, meshgen = env
    c, nd = _problem(capi, meshgen, dim, nx, order, dirichlet)
    if adr:
        b = (1.0, 0.5) if dim == 2 else (1.0, 0.5, 0.25)
        c.set_operator(-capi.laplacian() + capi.advection(b) + capi.reaction(1.0))
        c.init()
    if dirichlet and zero_g:
        c.set_dirichlet(np.zeros(nd))
    c.tune("small_front_rows", 0)
    c.solve(rtol=1e-11)
    i0 = c.solve(rtol=1e-11)
    u0 = c.solution()
    c.tune("small_front_rows", 2048)
    i1 = c.solve(rtol=1e-11)
    u1 = c.solution()
    assert c.solver_layout_kind(dirichlet)["workgroups"] == 1
    assert i0.persistent == 1 and i1.persistent == 1 and i0.converged == 1 and i1.converged == 1
    assert i1.iters == i0.iters and i1.relres == i0.relres, (i0.iters, i1.iters, i0.relres, i1.relres)
    assert np.array_equal(u1, u0), float(np.abs(u1 - u0).max())
    # a changed forcing, then changed Dirichlet data, through the fused front
    c.set_forcing(2.0 * np.ones(c.quadrature_nodes().shape[0]))
    c.init()
    i2 = c.solve(rtol=1e-11)
    u2 = c.solution()
    c.tune("small_front_rows", 0)
    i3 = c.solve(rtol=1e-11)
    assert i3.iters == i2.iters and np.array_equal(c.solution(), u2)
    c.close()


@pytest.mark.parametrize("dim,nx,order,n_rhs", [
    (2, 16, 1, 5),      # one workgroup per column
    (2, 16, 1, 300),    # more columns than one launch takes (256 CUs): several launches
    (2, 60, 1, 7),      # two workgroups per column: boards per column
    (3, 16, 1, 9),      # 3-D rows, import lists
    (2, 30, 2, 4),      # P2
    (2, 256, 1, 9),     # 35 workgroups per column: 7 columns per launch, then 2
])
def test_columns_of_a_handle_solve_side_by_side(env, dim, nx, order, n_rhs):
    """fdapde::SparseLU::solve(B) with several columns (utils/symbols.h:133-160; SMW, linear_algebra/smw.h:38-59): the columns of a system the
    single-launch CG holds in few workgroups run side by side in ONE launch (knob persist_cols) -- per column the arithmetic of a launch of
    its own, so identical bits and iteration counts; a zero column and a column that runs out of iterations included"""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.init()
    c.tune("dense_rows", 0)   # (this test is about the Krylov columns: a handle asked for hundreds of columns would otherwise invert, tests/test_gpu_dense.py)
    c.lin_compute(capi.MAT_STIFF, symmetric=True)
    B = np.random.default_rng(3).standard_normal((nd, n_rhs))
    B[:, 1] = 0.0
    B[:, 2] *= 1e-150   # (tiny but not zero: the relative stop test must not care)
    c.tune("persist_cols", 0)
    X0, i0 = c.lin_solve(B, rtol=1e-11)
    c.tune("persist_cols", 1)
    X1, i1 = c.lin_solve(B, rtol=1e-11)
    assert i0.persistent == 1 and i1.persistent == 1 and i0.converged == 1 and i1.converged == 1
    assert i1.iters == i0.iters and np.array_equal(X0, X1)
    assert not X1[:, 1].any()
    # against the matrix itself
    _, _, _ = c.dofs_get()
    rowptr, colidx = c.pattern_get()
    import scipy.sparse as sp

    A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), colidx, rowptr), shape=(nd, nd))
    R = A @ X1 - B
    assert np.linalg.norm(R[:, 0]) <= 1e-9 * np.linalg.norm(B[:, 0])
    c.close()


@pytest.mark.parametrize("dim,nx,n_rhs", [(2, 16, 6), (2, 60, 70), (3, 16, 5)])
def test_columns_of_a_non_symmetric_handle_side_by_side(env, dim, nx, n_rhs):
    """the same for a non-symmetric matrix (advection): the single-launch BiCGStab with the columns side by side"""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian() + capi.advection([1.0, 0.5, 0.25][:dim]) + capi.reaction(1.0))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.init()
    c.tune("dense_rows", 0)   # (the Krylov columns: see above)
    c.lin_compute(capi.MAT_STIFF, symmetric=False)
    B = np.random.default_rng(5).standard_normal((nd, n_rhs))
    B[:, 1] = 0.0
    c.tune("persist_cols", 0)
    X0, i0 = c.lin_solve(B, rtol=1e-11)
    c.tune("persist_cols", 1)
    X1, i1 = c.lin_solve(B, rtol=1e-11)
    assert i0.method_used == capi.SOLVER_BICGSTAB and i1.method_used == capi.SOLVER_BICGSTAB
    assert i0.persistent == 1 and i1.persistent == 1 and i0.converged == 1 and i1.converged == 1
    assert i1.iters == i0.iters and np.array_equal(X0, X1)
    rowptr, colidx = c.pattern_get()
    import scipy.sparse as sp

    A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), colidx, rowptr), shape=(nd, nd))
    assert np.linalg.norm(A @ X1[:, 0] - B[:, 0]) <= 1e-9 * np.linalg.norm(B[:, 0])
    c.close()


def test_side_by_side_columns_fall_back_when_a_launch_gives_up(env):
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(60)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.init()
    c.tune("dense_rows", 0)   # (this file is about the Krylov launches; the dense inverse of small systems: tests/test_gpu_dense.py)
    c.lin_compute(capi.MAT_STIFF, symmetric=True)
    B = np.random.default_rng(4).standard_normal((nd, 6))
    X0, i0 = c.lin_solve(B, rtol=1e-11)
    assert i0.persistent == 1 and c.solver_layout_kind(False)["workgroups"] >= 2
    c.tune("persist_timeout_us", 2000)
    c.tune("persist_debug_stall", 3)   # the last workgroup of every column leaves at iteration 3: every launch gives up
    X1, i1 = c.lin_solve(B, rtol=1e-11)
    assert i1.converged == 1 and i1.persistent == 0   # (the multi-launch path finished every column)
    assert np.linalg.norm(X1 - X0) <= 1e-9 * np.linalg.norm(X0)
    c.tune("persist_debug_stall", 0)
    c.tune("persist_retry", 1)
    X2, i2 = c.lin_solve(B, rtol=1e-11)
    assert i2.persistent == 1 and np.array_equal(X2, X0)
    c.close()


def test_graph_replay_is_rebuilt_when_the_blocked_layout_changes(env):
    """use_graph = 1 on a system that takes the blocked-ELL SpMV (2-D P2; the 3-D P2 "mass matrix" of the reference's 5-point rule with its
    negative weight is indefinite, no CG applies to it): a solve with Dirichlet data captures the fused-CG chunk on
    layout 1; the handle then solves with the mass matrix on layout 0 (no Dirichlet reduction) -- the captured graph bakes in the other
    layout's arrays and grid and must not be replayed (the graph key once ignored the blocked layout)"""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(48)
    _, f = meshgen.manufactured(2)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(2)
    _, _, coords = c.dofs_get()
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(0.25 * coords[:, 0])
    c.init()
    c.tune("persist", 0)
    c.tune("blocked", 2)
    b = np.cos(3.0 * coords[:, 0]) + coords[:, 1]
    res = {}
    for graph in (0, 1):
        c.tune("use_graph", graph)
        i = c.solve(method=capi.SOLVER_CG_FUSED, rtol=1e-11, check_every=8)
        assert i.converged == 1 and i.persistent == 0
        u = c.solution()
        c.tune("dense_rows", 0)   # (this file is about the Krylov launches; the dense inverse of small systems: tests/test_gpu_dense.py)
        c.lin_compute(capi.MAT_MASS, symmetric=True)
        x, li = c.lin_solve(b, rtol=1e-12, check_every=8)
        assert li.converged == 1
        res[graph] = (u, x, i.iters, li.iters)
    assert res[1][2] == res[0][2] and res[1][3] == res[0][3]
    assert np.linalg.norm(res[1][0] - res[0][0]) <= 1e-12 * np.linalg.norm(res[0][0])
    assert np.linalg.norm(res[1][1] - res[0][1]) <= 1e-12 * np.linalg.norm(res[0][1])
    c.close()


def test_persistent_path_under_contention_from_other_processes():
    """three processes solving C2-size systems on the same GPU at once: a persistent launch needs ALL its workgroups resident, which the
    other processes' launches can prevent; the bounded waits must turn that into a fall-back to the multi-launch kernels (or a late
    start), never into a hang or a wrong answer (tools/persist_contention.py asserts convergence and the analytic error per solve)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "persist_contention.py"), "3", "12"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:]
    assert out.stdout.count("solves ok") == 3, out.stdout[-3000:]


@pytest.mark.parametrize("dim,nx,order", [(2, 16, 1), (2, 40, 1), (3, 8, 1), (2, 12, 2)])
def test_direct_launch_of_a_single_right_hand_side(env, dim, nx, order):
    """fdapde::SparseLU::solve(b) with ONE column against a system of one workgroup: the launch reads b from pinned host
    memory, scales it, and writes the unscaled solution and its outcome record back itself (knob persist_direct).  Same recurrence as the
    general path -- its reference norm is the launch's own first r.r instead of the prologue kernel's sum, so the stop test may fire one
    iteration apart -- identical bits from call to call, the zero right-hand side and an exhausted iteration budget included."""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
    c.init()
    c.tune("dense_rows", 0)   # (this file is about the Krylov launches; the dense inverse of small systems: tests/test_gpu_dense.py)
    c.lin_compute(capi.MAT_STIFF, symmetric=True)
    assert c.solver_layout_kind(False)["workgroups"] == 1
    b = np.random.default_rng(11).standard_normal(nd)
    c.tune("persist_direct", 0)
    x0, i0 = c.lin_solve(b, rtol=1e-11)
    for spin in (0, 2000):
        c.tune("persist_direct", 1)
        c.tune("persist_direct_spin_us", spin)
        x1, i1 = c.lin_solve(b, rtol=1e-11)
        x2, i2 = c.lin_solve(b, rtol=1e-11)
        assert i1.persistent == 1 and i1.converged == 1 and abs(i1.iters - i0.iters) <= 1
        assert np.linalg.norm(x1 - x0) <= 1e-10 * np.linalg.norm(x0)
        assert i2.iters == i1.iters and np.array_equal(x1, x2)
        assert abs(i1.relres - i0.relres) <= 1e-3 * i0.relres + 1e-16 or abs(i1.iters - i0.iters) == 1
    rowptr, colidx = c.pattern_get()
    import scipy.sparse as sp

    A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), colidx, rowptr), shape=(nd, nd))
    assert np.linalg.norm(A @ x1 - b) <= 1e-9 * np.linalg.norm(b)
    xz, iz = c.lin_solve(np.zeros(nd))
    assert iz.iters == 0 and iz.converged == 1 and not xz.any()
    with pytest.raises(capi.FdapdeError) as e:
        c.lin_solve(b, rtol=1e-14, maxit=3)
    assert e.value.status == capi.ENOCONV
    xs, _ = c.lin_solve(1e-150 * b, rtol=1e-11)   # (tiny but not zero: the relative stop test must not care)
    assert np.linalg.norm(xs - 1e-150 * x1) <= 1e-10 * np.linalg.norm(1e-150 * x1)
    c.close()


@pytest.mark.parametrize("dim,nx,order,max_wg", [(3, 44, 1, 8), (2, 200, 1, 4), (2, 100, 2, 4)])
def test_wide_single_launch_matches_the_multi_launch_path(env, dim, nx, order, max_wg):
    """systems of more than 8 192 rows per workgroup (2.1 to 3.1 M rows on 256 CUs) run as ONE launch in the wide form of the plain storage:
    24 rows per thread, r and y in registers, p in its LDS table, x in HBM in slot order.  Here the form is forced on
    moderate systems by limiting the workgroups (knob persist_max_wg): same iteration count as the multi-launch kernels, same solution,
    the true residual on the exported matrix, identical bits from launch to launch, a warm start (parabolic stepper) included."""
    import scipy.sparse as sp

    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    _, _, coords = c.dofs_get()
    g = 0.1 * coords[:, 0]
    c.set_operator(-capi.laplacian() + capi.reaction(0.5))
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(g)
    c.init()
    c.tune("persist", 0)
    i0 = c.solve(rtol=1e-11)
    u0 = c.solution()
    c.tune("persist", 1)
    c.tune("persist_max_wg", max_wg)
    i1 = c.solve(rtol=1e-11)
    u1 = c.solution()
    lay = c.solver_layout_kind(True)
    assert lay["rows_per_thread"] == 24 and lay["sym"] == 0 and lay["workgroups"] <= max_wg, lay
    assert i1.persistent == 1 and i1.converged == 1 and abs(i1.iters - i0.iters) <= max(1, i0.iters // 100)
    assert np.linalg.norm(u1 - u0) <= 1e-10 * np.linalg.norm(u0)
    i2 = c.solve(rtol=1e-11)
    assert i2.iters == i1.iters and np.array_equal(c.solution(), u1)
    rp, ci = c.pattern_get()
    A = sp.csr_matrix((c.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
    b = c.force()
    assert np.linalg.norm(A @ u1 - b) <= 1e-9 * np.linalg.norm(b)
    assert c.solver_layout(True)[2] > 0
    c.close()


# ---- the instantiations of rounds 3 / 4 pinned DIRECTLY to the oracle, not through another HIP path -------------------------
@pytest.mark.parametrize("dim,nx,order,max_wg", [(2, 200, 1, 4), (2, 100, 2, 4)])
def test_wide_form_against_the_oracle(env, oracle, dim, nx, order, max_wg):
    """k_cg_persist<24, ...> (24 rows per thread, x in HBM between the iterations) against the oracle's assembly + direct solve on the same mesh"""
    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_square(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    _, _, coords = c.dofs_get()
    g = 0.1 * coords[:, 0]
    fq = f(c.quadrature_nodes())
    c.set_operator(-capi.laplacian() + capi.reaction(0.5))
    c.set_forcing(fq)
    c.set_dirichlet(g)
    c.init()
    c.tune("persist_max_wg", max_wg)
    info = c.solve(rtol=1e-11)
    lay = c.solver_layout_kind(True)
    assert info.persistent == 1 and info.converged == 1 and lay["rows_per_thread"] == 24, lay
    m = oracle.Mesh(nodes, cells, bnd)
    ref = oracle.pde_init_solve(m, order, -oracle.laplacian() + oracle.reaction(0.5), forcing_q=fq, dirichlet=g)
    assert ref.n_dofs == nd
    assert np.linalg.norm(c.solution() - ref.solution) <= 1e-8 * np.linalg.norm(ref.solution)
    c.close()


@pytest.mark.parametrize("mesh,order", [("unit_square_16", 1), ("unit_square_32", 1), ("unit_square_16", 2)])
def test_direct_launch_against_the_oracle(env, oracle, mesh_loader, mesh, order):
    """fdapde_lin_solve of ONE column as the zero-copy launch (PersistArgs::direct) against a sparse LU of the ORACLE's matrix"""
    import scipy.sparse.linalg as spl

    capi, _ = env
    m = mesh_loader(mesh)
    c = capi.Context(0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(order)
    c.set_operator(-capi.laplacian() + capi.reaction(1.0))
    c.set_forcing(np.zeros(c.sizes()["n_quadrature"] * m.n_cells))
    c.init()
    c.tune("dense_rows", 0)   # (this file is about the Krylov launches; the dense inverse of small systems: tests/test_gpu_dense.py)
    c.lin_compute(capi.MAT_STIFF)
    assert c.solver_layout_kind(False)["workgroups"] == 1
    c.tune("persist_direct", 1)
    b = np.random.default_rng(3).standard_normal(nd)
    x, info = c.lin_solve(b, rtol=1e-12)
    assert info.persistent == 1 and info.converged == 1
    od, _, ond, _ = oracle.enumerate_dofs(m, order)
    A = oracle.assemble_operator(m, order, od, ond, -oracle.laplacian() + oracle.reaction(1.0))
    ref = spl.splu(A.to_scipy().tocsc()).solve(b)
    assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
    c.close()


@pytest.mark.parametrize("mesh,order", [("unit_square", 1), ("unit_square", 2), ("unit_sphere", 2), ("c_shaped", 2)])
def test_single_launch_bicgstab_against_the_oracle(env, oracle, mesh_loader, mesh, order):
    """k_bicg_persist (the whole Jacobi-BiCGStab as one launch) on an advection-diffusion-reaction operator against the oracle's assembly + LU"""
    capi, _ = env
    m = mesh_loader(mesh)
    c = capi.Context(0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(order)
    _, _, coords = c.dofs_get()
    dim = m.nodes.shape[1]
    b = np.array([1.0, 0.5, 0.25])[:dim]
    g = coords[:, 0] * coords[:, 1]
    fq = np.ones(c.sizes()["n_quadrature"] * m.n_cells)
    c.set_operator(-capi.laplacian() + capi.advection(b) + capi.reaction(1.0))
    c.set_forcing(fq)
    c.set_dirichlet(g)
    c.init()
    info = c.solve(rtol=1e-11)
    assert info.converged == 1 and info.method_used == capi.SOLVER_BICGSTAB and info.persistent == 1
    ref = oracle.pde_init_solve(m, order, -oracle.laplacian() + oracle.advection(b) + oracle.reaction(1.0), forcing_q=fq, dirichlet=g)
    assert np.linalg.norm(c.solution() - ref.solution) <= 1e-8 * np.linalg.norm(ref.solution)
    c.close()
