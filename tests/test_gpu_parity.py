"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the reference's fixtures.

Bars (SURVEY.md section 8d / BASELINE.md):
  * DOF tables, boundary sets, CSR pattern ............ bit-exact
  * matrix / vector entries ........................... |d| <= 1e-12 * max(1, ||A||_max)   (summation order differs) AND
                                                        |d| <= 1e-13 * ||A||_max: SURVEY's bound alone is absolute below 1 and would let a
                                                        relative error of 1e-7 through on mass entries of ~1e-5
  * solutions ......................................... ||u_gpu - u_ref||_2 / ||u_ref||_2 <= 1e-8 with rtol 1e-10
  * the reference's own criterion ..................... sum(M * err^2) < 1e-7 (1e-5 for P1 advection-diffusion)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENTRY_TOL = 1e-12
ENTRY_RTOL = 1e-13   # relative to the largest entry of the reference matrix / vector
SOL_TOL = 1e-8


@pytest.fixture(scope="module")
def capi():
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi as m

    assert m.load().fdapde_device_count() >= 1, "no HIP device visible: the GPU tests must not fall back to anything"
    return m


@pytest.fixture(scope="module")
def ctx(capi):
    c = capi.Context(device=0)
    yield c
    c.close()


def _ops(mod, M):
    K = np.array([[2.0, 0.3], [0.3, 1.0]]) if M == 2 else np.array([[2.0, 0.3, 0.1], [0.3, 1.0, 0.2], [0.1, 0.2, 1.5]])
    b = np.array([0.7, -0.2]) if M == 2 else np.array([0.7, -0.2, 0.4])
    return {
        "neg_laplacian": -mod.laplacian(),
        "mass": mod.reaction(1.0),
        "adr": -mod.laplacian() + mod.advection(b) + mod.reaction(1.5),
        "diffusion": mod.diffusion(K) + 0.5 * mod.reaction(2.0),
        "laplacian_minus_dt": mod.laplacian() - mod.dt(),
        # a diffusion tensor that is NOT symmetric (the summed tensor then takes the full 9 + 3 + 1 reference tables, not the compact symmetric ones)
        "diffusion_nonsym": mod.diffusion(K + (np.array([[0.0, 0.4], [-0.2, 0.0]]) if M == 2 else np.array([[0.0, 0.4, 0.0], [-0.2, 0.0, 0.3], [0.1, -0.3, 0.0]])))
        + mod.advection(b) + mod.reaction(0.5),
        # ... and the same tensor WITHOUT advection: the reference takes the expression for symmetric (diffusion.h:42) and integrates only the pairs
        # dof_i >= dof_j, mirrored (fem_assembler.h:94-102, 116-117) -- found by tests/test_gpu_fuzz_assembly.py
        "diffusion_nonsym_mirrored": mod.diffusion(K + (np.array([[0.0, 0.4], [-0.2, 0.0]]) if M == 2 else np.array([[0.0, 0.4, 0.0], [-0.2, 0.0, 0.3], [0.1, -0.3, 0.0]])))
        + mod.reaction(0.5),
    }


def _entry_close(a, b):
    amax = np.abs(b).max()
    err = np.abs(a - b).max()
    return err <= ENTRY_TOL * max(1.0, amax) and err <= ENTRY_RTOL * amax


CASES = [("unit_square_16", 1), ("unit_square_16", 2), ("c_shaped", 1), ("c_shaped", 2), ("unit_square", 1), ("unit_square", 2),
         ("unit_sphere", 1), ("unit_sphere", 2), ("quasi_circle", 2)]


@pytest.mark.parametrize("mesh_name,order", CASES)
def test_space_is_bit_exact(capi, ctx, oracle, mesh_loader, mesh_name, order):
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, ob, ond, _ = oracle.enumerate_dofs(m, order)
    dofs, bnd, coords = ctx.dofs_get()
    assert nd == ond and np.array_equal(dofs, od) and np.array_equal(bnd, ob)
    assert np.array_equal(coords, oracle.dofs_coords(m, order, od, ond))
    A = oracle.assemble_operator(m, order, od, ond, -oracle.laplacian())
    rp, ci = ctx.pattern_get()
    assert np.array_equal(rp, A.rowptr) and np.array_equal(ci, A.colidx)
    qn = ctx.quadrature_nodes()
    assert np.abs(qn - oracle.quadrature_nodes(m, order)).max() < 1e-14


@pytest.mark.parametrize("mesh_name,order", CASES)
@pytest.mark.parametrize("variant", ["rows", "atomic", "coloured", "partitioned", "wave"])
def test_operator_assembly_matches_oracle(capi, ctx, oracle, mesh_loader, mesh_name, order, variant):
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, _, _, _ = oracle.enumerate_dofs(m, order)
    assembly = {"rows": capi.ASSEMBLY_ROWS, "atomic": capi.ASSEMBLY_ATOMIC, "coloured": capi.ASSEMBLY_COLOURED,
                "partitioned": capi.ASSEMBLY_PARTITIONED, "wave": capi.ASSEMBLY_WAVE}[variant]
    for name in _ops(capi, m.M):
        ctx.assemble_operator(capi.MAT_STIFF, _ops(capi, m.M)[name], assembly)
        got = ctx.matrix_values(capi.MAT_STIFF)
        ref = oracle.assemble_operator(m, order, od, nd, _ops(oracle, m.M)[name])
        assert _entry_close(got, ref.values), (name, np.abs(got - ref.values).max())


def test_rows_assembly_is_deterministic_and_symmetric(capi, ctx, oracle, mesh_loader):
    m = mesh_loader("unit_sphere")
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(2)
    op = -capi.laplacian() + capi.reaction(0.5)
    ctx.assemble_operator(capi.MAT_STIFF, op)
    a = ctx.matrix_values(capi.MAT_STIFF)
    ctx.assemble_operator(capi.MAT_STIFF, op)
    b = ctx.matrix_values(capi.MAT_STIFF)
    assert np.array_equal(a, b)   # bitwise reproducible: no atomics in the default path
    import scipy.sparse as sp

    rp, ci = ctx.pattern_get()
    A = sp.csr_matrix((a, ci, rp), shape=(nd, nd))
    assert abs(A - A.T).max() == 0.0   # bitwise symmetric for symmetric forms


@pytest.mark.parametrize("mesh_name,order", [("unit_square_16", 1), ("unit_square_16", 2), ("unit_sphere", 1), ("unit_sphere", 2)])
def test_space_varying_coefficients(capi, ctx, oracle, mesh_loader, mesh_name, order):
    """Discretized*Field::forward(nq*cell + q) row indexing (integrator.h:98-101) with genuinely varying data"""
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, _, _, _ = oracle.enumerate_dofs(m, order)
    rows = ctx.sizes()["n_quadrature"] * m.n_cells
    rng = np.random.default_rng(7)
    N = m.N
    Kq = rng.uniform(0.5, 1.5, (rows, N * N))
    bq = rng.uniform(-1, 1, (rows, N))
    cq = rng.uniform(0, 2, rows)
    mk = lambda mod: mod.diffusion_field(Kq) + mod.advection_field(bq) + mod.reaction_field(cq)
    ctx.assemble_operator(capi.MAT_STIFF, mk(capi))
    got = ctx.matrix_values(capi.MAT_STIFF)
    ref = oracle.assemble_operator(m, order, od, nd, mk(oracle))
    assert _entry_close(got, ref.values)


@pytest.mark.parametrize("mesh_name,order", [("unit_square_16", 2), ("unit_sphere", 1), ("unit_sphere", 2)])
def test_space_varying_symmetric_operator_is_bitwise_symmetric(capi, ctx, oracle, mesh_loader, mesh_name, order):
    """the space-varying integrand (quadrature node outermost, one pulled-back tensor per node) evaluates the bilinear form pairwise and makes
    the pulled-back tensor exactly symmetric where the summed coefficient tensor is: a symmetric operator -- reaction field, symmetric
    diffusion field -- gives A_ij == A_ji bit for bit, like the constant-coefficient forms; and it matches the oracle"""
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, _, _, _ = oracle.enumerate_dofs(m, order)
    rows = ctx.sizes()["n_quadrature"] * m.n_cells
    rng = np.random.default_rng(11)
    N = m.N
    L = rng.uniform(-0.3, 0.3, (rows, N, N))
    K = np.einsum("qij,qkj->qik", L, L) + np.eye(N)[None] * rng.uniform(0.5, 1.5, (rows, 1, 1))   # symmetric positive per node
    K = 0.5 * (K + np.transpose(K, (0, 2, 1)))
    cq = rng.uniform(0.1, 2.0, rows)
    import scipy.sparse as sp

    rp, ci = None, None
    for mk in (lambda mod: -mod.laplacian() + mod.reaction_field(cq),
               lambda mod: -mod.diffusion_field(K.reshape(rows, N * N)) + mod.reaction_field(cq) + 0.5 * mod.reaction(1.0)):
        ctx.assemble_operator(capi.MAT_STIFF, mk(capi))
        got = ctx.matrix_values(capi.MAT_STIFF)
        ref = oracle.assemble_operator(m, order, od, nd, mk(oracle))
        assert _entry_close(got, ref.values)
        if rp is None:
            rp, ci = ctx.pattern_get()
        A = sp.csr_matrix((got, ci, rp), shape=(nd, nd))
        assert abs(A - A.T).max() == 0.0


@pytest.mark.parametrize("mesh_name,order", [("unit_square_16", 1), ("unit_square_16", 2), ("unit_sphere", 1), ("unit_sphere", 2), ("c_shaped", 2)])
def test_varying_advection_reaction_next_to_constant_diffusion(capi, ctx, oracle, mesh_loader, mesh_name, order):
    """operators whose advection / reaction vary while the diffusion part does not take the split integrand (constants through the reference
    tensors, the varying leaves per node without a tensor pull-back; element_row OPK 5): against the oracle, and against the per-node tensor form
    of the fully space-varying case (knob asm_split_varying 0) -- same numbers up to rounding"""
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, _, _, _ = oracle.enumerate_dofs(m, order)
    rows = ctx.sizes()["n_quadrature"] * m.n_cells
    rng = np.random.default_rng(23)
    N = m.N
    bq = rng.uniform(-1, 1, (rows, N))
    cq = rng.uniform(0.2, 2, rows)
    K = np.eye(N) + 0.2 * rng.uniform(-1, 1, (N, N))   # constant, not symmetric
    bc = np.array([0.7, -0.3, 0.45])[:N]
    ops = (lambda mod: -mod.laplacian() + mod.reaction_field(cq),
           lambda mod: -mod.diffusion(K) + mod.advection_field(bq) + mod.reaction_field(cq) + 0.25 * mod.reaction(2.0),
           lambda mod: -mod.laplacian() + mod.advection_field(bq) + mod.advection(bc))
    for mk in ops:
        ctx.tune("asm_split_varying", 1)
        ctx.assemble_operator(capi.MAT_STIFF, mk(capi))
        got = ctx.matrix_values(capi.MAT_STIFF)
        ref = oracle.assemble_operator(m, order, od, nd, mk(oracle))
        assert _entry_close(got, ref.values)
        ctx.tune("asm_split_varying", 0)
        ctx.assemble_operator(capi.MAT_STIFF, mk(capi))
        other = ctx.matrix_values(capi.MAT_STIFF)
        assert _entry_close(got, other)
    ctx.tune("asm_split_varying", 1)


@pytest.mark.parametrize("mesh_name,order", [("unit_square_16", 2), ("unit_sphere", 1), ("unit_sphere", 2)])
def test_init_with_space_varying_coefficients_and_forcing(capi, ctx, oracle, mesh_loader, mesh_name, order):
    """one sweep reads the coefficient rows by CELL and the forcing samples by BLOCK-CELL: the two row indices must not be mixed up"""
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, _, _, _ = oracle.enumerate_dofs(m, order)
    qn = oracle.quadrature_nodes(m, order)
    cq = 1.0 + qn[:, 0] ** 2 + 0.5 * np.cos(2.0 * qn[:, 1])
    fq = np.sin(3.0 * qn[:, 0]) + qn[:, 1] ** 2
    ctx.set_operator(-capi.laplacian() + capi.reaction_field(cq))
    ctx.set_forcing(fq)
    ctx.init()
    assert _entry_close(ctx.force(), oracle.assemble_forcing(m, order, od, nd, fq))
    ref = oracle.assemble_operator(m, order, od, nd, -oracle.laplacian() + oracle.reaction_field(cq))
    assert _entry_close(ctx.matrix_values(capi.MAT_STIFF), ref.values)
    # the mass matrix of the same init (P2: second pass of the same visit-parallel launch, also behind a space-varying integrand)
    mass = oracle.assemble_operator(m, order, od, nd, oracle.reaction(1.0))
    assert _entry_close(ctx.matrix_values(capi.MAT_MASS), mass.values)
    Kq = np.tile(np.eye(m.N).reshape(-1), (qn.shape[0], 1)) * (1.0 + 0.3 * np.sin(qn[:, :1]))
    ctx.set_operator(-capi.diffusion_field(Kq) + capi.reaction_field(cq))
    ctx.init()
    ref = oracle.assemble_operator(m, order, od, nd, -oracle.diffusion_field(Kq) + oracle.reaction_field(cq))
    assert _entry_close(ctx.matrix_values(capi.MAT_STIFF), ref.values)
    assert _entry_close(ctx.matrix_values(capi.MAT_MASS), mass.values)
    assert _entry_close(ctx.force(), oracle.assemble_forcing(m, order, od, nd, fq))


@pytest.mark.parametrize("mesh_name,order", CASES)
def test_init_force_mass_spmv(capi, ctx, oracle, mesh_loader, mesh_name, order):
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, _, _, _ = oracle.enumerate_dofs(m, order)
    qn = oracle.quadrature_nodes(m, order)
    fq = np.sin(3.0 * qn[:, 0]) + qn[:, 1] ** 2
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(fq)
    ctx.init()
    f_default = ctx.force()   # the sweep reads the samples from their block-cell ordered copy (fdapde_set_forcing)
    assert _entry_close(f_default, oracle.assemble_forcing(m, order, od, nd, fq))
    for knob, val in (("asm_fq_bc", 0), ("asm_fq_block", 1)):   # samples gathered by cell id / per-visit load coefficients: the same sums,
        ctx.tune(knob, val)                                      # in the same order -> the same bits
        ctx.set_forcing(fq)
        ctx.init()
        assert np.array_equal(ctx.force(), f_default), knob
    ctx.tune("asm_fq_block", 0), ctx.tune("asm_fq_bc", 1)
    ctx.set_forcing(fq)
    ctx.init()
    assert np.array_equal(ctx.force(), f_default)
    Mo = oracle.assemble_operator(m, order, od, nd, oracle.reaction(1.0))
    assert _entry_close(ctx.matrix_values(capi.MAT_MASS), Mo.values)
    Ao = oracle.assemble_operator(m, order, od, nd, -oracle.laplacian())
    x = np.random.default_rng(3).standard_normal(nd)
    for which, ref in ((capi.MAT_STIFF, Ao), (capi.MAT_MASS, Mo)):
        y = ctx.spmv(which, x)
        yr = ref.matvec(x)
        assert np.abs(y - yr).max() <= 1e-12 * max(1.0, np.abs(yr).max())


@pytest.mark.parametrize("variant", ["stream", "team4", "team8", "team16", "team32", "team64", "pair2", "pair4", "pair8", "pair16", "pair32"])
@pytest.mark.parametrize("mesh_name,order", [("unit_square", 1), ("unit_sphere", 1), ("unit_sphere", 2), ("c_shaped", 2)])
def test_spmv_variants(capi, ctx, oracle, mesh_loader, monkeypatch, variant, mesh_name, order):
    """every SpMV kernel form / team width gives the oracle's y = A x (ragged rows, rows longer than a team, tails)"""
    if variant == "stream":
        monkeypatch.setenv("FDAPDE_SPMV", "stream")
    else:
        monkeypatch.setenv("FDAPDE_SPMV", "team" if variant.startswith("team") else "pair")
        monkeypatch.setenv("FDAPDE_SPMV_TEAM", variant[4:])
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)   # the kernel form is chosen here
    od, _, _, _ = oracle.enumerate_dofs(m, order)
    op = lambda mod: -mod.laplacian() + mod.reaction(0.3)
    ctx.assemble_operator(capi.MAT_STIFF, op(capi))
    ref = oracle.assemble_operator(m, order, od, nd, op(oracle))
    x = np.random.default_rng(11).standard_normal(nd)
    y, yr = ctx.spmv(capi.MAT_STIFF, x), ref.matvec(x)
    assert np.abs(y - yr).max() <= 1e-12 * max(1.0, np.abs(yr).max())
    # and inside a solve (fused dot products)
    ctx.set_operator(op(capi))
    ctx.set_forcing(np.ones(ctx.sizes()["n_quadrature"] * m.n_cells))
    ctx.set_dirichlet(None)
    ctx.init()
    info = ctx.solve(method=capi.SOLVER_CG, rtol=1e-11)
    sol = oracle.pde_init_solve(m, order, op(oracle), forcing_q=np.ones(ctx.sizes()["n_quadrature"] * m.n_cells))
    assert info.converged == 1
    assert np.linalg.norm(ctx.solution() - sol.solution) / np.linalg.norm(sol.solution) <= SOL_TOL


def _l2(oracle_mass, err):
    return float(np.sum(oracle_mass.matvec(err * err)))


def _solve_case(capi, ctx, oracle, m, order, mk_op, forcing_fn, dirichlet_fn, method=None):
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    _, _, coords = ctx.dofs_get()
    qn = ctx.quadrature_nodes()
    fq = np.array([forcing_fn(p) for p in qn]) if forcing_fn else np.zeros(qn.shape[0])
    g = np.array([dirichlet_fn(p) for p in coords])
    ctx.set_operator(mk_op(capi))
    ctx.set_forcing(fq)
    ctx.set_dirichlet(g)
    ctx.init()
    info = ctx.solve(method=capi.SOLVER_AUTO if method is None else method, rtol=1e-10)
    assert info.converged == 1 and info.relres <= 1e-10
    ref = oracle.pde_init_solve(m, order, mk_op(oracle), forcing_q=fq, dirichlet=g, direct=True)
    u = ctx.solution()
    assert np.linalg.norm(u - ref.solution) / np.linalg.norm(ref.solution) <= SOL_TOL
    # after solve(), stiff() is the row-zeroed matrix and force() carries g on the boundary (fem_solver_base.h:148-152)
    assert _entry_close(ctx.matrix_values(capi.MAT_STIFF), ref.stiff.values)
    assert _entry_close(ctx.force(), ref.force)
    return u, coords, ref, info


def test_pde_laplacian_isotropic_order1(capi, ctx, oracle, mesh_loader):
    """fem_pde_test.cpp:43-75"""
    u_ex = lambda x: x[0] + x[1]
    u, coords, ref, _ = _solve_case(capi, ctx, oracle, mesh_loader("unit_square"), 1, lambda mod: -mod.laplacian(), None, u_ex)
    ex = np.array([u_ex(p) for p in coords])
    assert _l2(ref.mass, ex - u) < 1e-7


def test_pde_laplacian_isotropic_order2_callable_force(capi, ctx, oracle, mesh_loader):
    """fem_pde_test.cpp:78-107"""
    u_ex = lambda x: 1.0 - x[0] * x[0] - x[1] * x[1]
    u, coords, ref, _ = _solve_case(capi, ctx, oracle, mesh_loader("unit_square"), 2, lambda mod: -mod.laplacian(), lambda x: 4.0, u_ex)
    ex = np.array([u_ex(p) for p in coords])
    assert _l2(ref.mass, ex - u) < 1e-7


def _advdiff():
    pi = np.pi
    alpha, gamma = 1.0, pi
    l1 = -alpha / 2 - np.sqrt((alpha / 2) ** 2 + pi * pi)
    l2 = -alpha / 2 + np.sqrt((alpha / 2) ** 2 + pi * pi)
    p = (1 - np.exp(l2)) / (np.exp(l1) - np.exp(l2))
    u = lambda x: -gamma / (pi * pi) * (p * np.exp(l1 * x[0]) + (1 - p) * np.exp(l2 * x[0]) - 1.0) * np.sin(pi * x[1])
    f = lambda x: gamma * np.sin(pi * x[1])
    return u, f, np.array([-alpha, 0.0])


@pytest.mark.parametrize("order,gate", [(1, 1e-5), (2, 1e-7)])
def test_pde_advection_diffusion(capi, ctx, oracle, mesh_loader, order, gate):
    """fem_pde_test.cpp:113-166 (P1, 1e-5) and 172-212 (P2, 1e-7): non-symmetric operator -> BiCGStab"""
    u_ex, f, beta = _advdiff()
    u, coords, ref, info = _solve_case(capi, ctx, oracle, mesh_loader("unit_square"), order,
                                       lambda mod: -mod.laplacian() + mod.advection(beta), f, lambda x: 0.0)
    ex = np.array([u_ex(p) for p in coords])
    assert _l2(ref.mass, ex - u) < gate


@pytest.mark.parametrize("order", [1, 2])
def test_pde_3d(capi, ctx, oracle, mesh_loader, order):
    """3-D (unit_sphere, 1395 negatively oriented tets): P1 reproduces x+y+z, P2 reproduces x^2 + yz"""
    if order == 1:
        u_ex, f = (lambda x: x[0] + x[1] + x[2]), None
    else:
        u_ex, f = (lambda x: x[0] * x[0] + x[1] * x[2]), (lambda x: -2.0)
    u, coords, ref, _ = _solve_case(capi, ctx, oracle, mesh_loader("unit_sphere"), order, lambda mod: -mod.laplacian(), f, u_ex)
    ex = np.array([u_ex(p) for p in coords])
    assert np.abs(u - ex).max() < 1e-8


def test_solve_without_dirichlet_and_bicgstab_on_spd(capi, ctx, oracle, mesh_loader):
    """PDE::solve skips set_dirichlet_bc when no boundary data is set (pde.h:103); -Lap + c is SPD on its own.
    Also runs BiCGStab on the same SPD system."""
    m = mesh_loader("unit_square_32")
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(2)
    qn = ctx.quadrature_nodes()
    fq = np.cos(qn[:, 0]) * qn[:, 1]
    mk = lambda mod: -mod.laplacian() + mod.reaction(3.0)
    ctx.set_operator(mk(capi))
    ctx.set_forcing(fq)
    ctx.set_dirichlet(None)
    ctx.init()
    ref = oracle.pde_init_solve(m, 2, mk(oracle), forcing_q=fq, dirichlet=None)
    for method in (capi.SOLVER_CG, capi.SOLVER_BICGSTAB):
        info = ctx.solve(method=method, rtol=1e-11)
        assert info.converged == 1
        u = ctx.solution()
        assert np.linalg.norm(u - ref.solution) / np.linalg.norm(ref.solution) <= SOL_TOL
    assert _entry_close(ctx.matrix_values(capi.MAT_STIFF), ref.stiff.values)


def test_error_behaviour(capi, ctx, mesh_loader):
    """'solver must be initialized first!' (fem_solver_base.h:146, fem_linear_elliptic_solver.h:36) -> ENOTINIT;
    non-convergence -> ENOCONV / success = false"""
    m = mesh_loader("unit_square_16")
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    ctx.dofs_build(1)
    with pytest.raises(capi.FdapdeError) as e:
        ctx.solve()
    assert e.value.status == capi.ENOTINIT
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(np.ones(3 * m.n_cells))
    ctx.set_dirichlet(np.zeros(m.n_nodes))
    ctx.init()
    info = ctx.solve(maxit=2, raise_on_noconv=False)
    assert info.converged == 0 and info.iters == 2
    with pytest.raises(capi.FdapdeError) as e:
        ctx.solve(maxit=2)
    assert e.value.status == capi.ENOCONV
    info = ctx.solve()
    assert info.converged == 1


def test_distributed_code_path_on_one_rank(capi, ctx, oracle, mesh_loader):
    """The multi-GPU branch of fdapde_solve (RCCL communicator, interface pack / all-reduce / unpack, owner-masked dots)
    exercised with a 1-rank communicator and an artificial interface set: the all-reduce is then the identity and the
    result must equal the plain solve.  (Multi-rank correctness of the maps + recurrence: tests/test_dist_cpu.py.)"""
    m = mesh_loader("unit_sphere")
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(1)
    _, _, coords = ctx.dofs_get()
    qn = ctx.quadrature_nodes()
    fq = np.sin(2.0 * qn[:, 0]) + qn[:, 2]
    g = coords[:, 0] - 0.5 * coords[:, 1]
    ctx.set_operator(-capi.laplacian() + capi.reaction(0.7))
    ctx.set_forcing(fq)
    ctx.set_dirichlet(g)
    ctx.init()
    plain = ctx.solve(rtol=1e-11)
    u_plain = ctx.solution()
    ctx.comm_init(1, 0, capi.Context.comm_unique_id())
    local_dof = np.arange(0, nd, 3, dtype=np.int32)
    if_index = np.random.default_rng(5).permutation(local_dof.size).astype(np.int32)
    ctx.halo_setup(local_dof.size, local_dof, if_index, np.ones(nd, dtype=np.uint8))
    info = ctx.solve(rtol=1e-11)
    u_dist = ctx.solution()
    assert info.converged == 1 and info.iters == plain.iters
    sr = ctx.solve(method=capi.SOLVER_CG_SR, rtol=1e-11)   # the variant multi-rank runs use: one all-reduce per iteration
    assert sr.converged == 1 and np.abs(ctx.solution() - u_plain).max() <= 1e-9 * max(1.0, np.abs(u_plain).max())
    # two different CG forms (fused-update with lazy x against the 3-kernel multi-GPU form) stopped at the same residual
    assert np.abs(u_dist - u_plain).max() <= 1e-9 * max(1.0, np.abs(u_plain).max())
    ref = oracle.pde_init_solve(m, 1, -oracle.laplacian() + oracle.reaction(0.7), forcing_q=fq, dirichlet=g)
    assert np.linalg.norm(u_dist - ref.solution) / np.linalg.norm(ref.solution) <= SOL_TOL
    bi = ctx.solve(method=capi.SOLVER_BICGSTAB, rtol=1e-11)   # distributed BiCGStab over the same 1-rank RCCL communicator
    assert bi.converged == 1 and bi.method_used == capi.SOLVER_BICGSTAB
    assert np.abs(ctx.solution() - u_plain).max() <= 1e-9 * max(1.0, np.abs(u_plain).max())


def test_row_distributed_form_and_rccl_reductions_on_one_rank(capi, ctx, oracle, mesh_loader):
    """fdapde_rowdist_setup over the library's own RCCL communicator with ONE rank (every DOF owned here, no ghost): the row-distributed
    instantiation of the single launch (fine-grained rank board, two-level dot gather with one rank record, system-scope accesses) against
    the ORACLE's direct solve; and fdapde_comm_allreduce, what bench.py's ranks use for barriers and the max over ranks, over real RCCL"""
    m = mesh_loader("unit_square")
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(2)
    _, _, coords = ctx.dofs_get()
    nq = ctx.sizes()["n_quadrature"]
    fq = np.cos(2.0 * ctx.quadrature_nodes()[:, 0])
    g = coords[:, 0] - 0.5 * coords[:, 1]
    ctx.comm_init(1, 0, capi.Context.comm_unique_id())
    assert "/opt/rocm" in capi.Context.comm_library(), capi.Context.comm_library()   # the RCCL of the HIP runtime this library is bound to
    assert np.array_equal(ctx.comm_allreduce([1.5, -2.0, 7.0], "max"), [1.5, -2.0, 7.0])
    assert np.array_equal(ctx.comm_allreduce([0.25, 3.0], "sum"), [0.25, 3.0])
    ctx.rowdist_setup(np.arange(nd, dtype=np.int64), np.zeros(nd, dtype=np.int32))
    for op_c, op_o, method in ((-capi.laplacian() + capi.reaction(0.7), -oracle.laplacian() + oracle.reaction(0.7), capi.SOLVER_CG_FUSED),
                               (-capi.laplacian() + capi.advection([0.8, -0.4]), -oracle.laplacian() + oracle.advection([0.8, -0.4]), capi.SOLVER_BICGSTAB)):
        ctx.set_operator(op_c)
        ctx.set_forcing(fq)
        ctx.set_dirichlet(g)
        ctx.init()
        info = ctx.solve(rtol=1e-11)
        assert info.converged == 1 and info.persistent == 1 and info.method_used == method
        ref = oracle.pde_init_solve(m, 2, op_o, forcing_q=fq, dirichlet=g)
        assert np.linalg.norm(ctx.solution() - ref.solution) / np.linalg.norm(ref.solution) <= SOL_TOL


def test_factor_once_solve_many(capi, ctx, oracle, mesh_loader):
    """fdapde::SparseLU usage (utils/symbols.h:133-160; SMW, linear_algebra/smw.h:46-48): compute(A) once, solve(B) for
    several right-hand sides; symmetric (CG) and general (BiCGStab) matrices; values handed over or taken from the context"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    m = mesh_loader("unit_sphere")
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(2)
    rp, ci = ctx.pattern_get()
    ctx.set_operator(-capi.laplacian() + capi.reaction(1.0))
    ctx.set_forcing(None)
    ctx.init()
    B = np.random.default_rng(2).standard_normal((nd, 3))
    A = sp.csr_matrix((ctx.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
    ctx.lin_compute(capi.MAT_STIFF)
    X, info = ctx.lin_solve(B, rtol=1e-12)
    Xr = spla.splu(A.tocsc()).solve(B)
    assert info.converged == 1 and np.linalg.norm(X - Xr) / np.linalg.norm(Xr) <= SOL_TOL
    # a caller-supplied non-symmetric matrix on the same pattern (mass + 0.3 * advection-like perturbation)
    ctx.assemble_operator(capi.MAT_STIFF, capi.reaction(2.0) + capi.advection(np.array([0.4, -0.1, 0.2])) - capi.laplacian())
    vals = ctx.matrix_values(capi.MAT_STIFF)
    A2 = sp.csr_matrix((vals, ci, rp), shape=(nd, nd))
    ctx.lin_compute(capi.MAT_STIFF, values=vals, symmetric=False)
    x, info = ctx.lin_solve(B[:, 0], rtol=1e-12)
    xr = spla.splu(A2.tocsc()).solve(B[:, 0])
    assert info.method_used == capi.SOLVER_BICGSTAB and np.linalg.norm(x - xr) / np.linalg.norm(xr) <= SOL_TOL
    # the handle survives an ordinary PDE solve in between
    ctx.set_forcing(np.ones(ctx.sizes()["n_quadrature"] * m.n_cells))
    ctx.init()
    ctx.solve()
    x2, _ = ctx.lin_solve(B[:, 0], rtol=1e-12)
    assert np.linalg.norm(x2 - xr) / np.linalg.norm(xr) <= SOL_TOL


@pytest.mark.parametrize("mesh_name,order", [("unit_square", 1), ("unit_square", 2), ("unit_sphere", 1), ("unit_sphere", 2)])
def test_single_reduction_cg_matches_cg(capi, ctx, oracle, mesh_loader, mesh_name, order):
    """FDAPDE_SOLVER_CG_SR (Chronopoulos-Gear: both dot products fused into the SpMV) gives the CG solution, in about the same
    number of iterations, with and without Dirichlet data"""
    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    _, _, coords = ctx.dofs_get()
    qn = ctx.quadrature_nodes()
    fq = np.cos(2.0 * qn[:, 0]) * qn[:, 1] + 1.0
    g = np.sin(coords[:, 0]) + coords[:, -1]
    mk = lambda mod: -mod.laplacian() + mod.reaction(0.5)
    ctx.set_operator(mk(capi))
    ctx.set_forcing(fq)
    for dirichlet in (g, None):
        ctx.set_dirichlet(dirichlet)
        ctx.init()
        a = ctx.solve(method=capi.SOLVER_CG, rtol=1e-11)
        ua = ctx.solution()
        b = ctx.solve(method=capi.SOLVER_CG_SR, rtol=1e-11)
        ub = ctx.solution()
        assert a.converged == 1 and b.converged == 1 and b.method_used == capi.SOLVER_CG_SR
        # same Krylov space in exact arithmetic; in floating point the recurrences of the single-reduction form drift a
        # little more near the attainable accuracy (observed: +12 % iterations at rtol 1e-11 on the P2 unit_square system)
        assert a.iters - 3 <= b.iters <= int(1.2 * a.iters) + 3, (a.iters, b.iters)
        ref = oracle.pde_init_solve(m, order, mk(oracle), forcing_q=fq, dirichlet=dirichlet)
        cf = ctx.solve(method=capi.SOLVER_CG_FUSED, rtol=1e-11)    # alpha from the explicit r.r, beta from alpha^2 Ap.Ap - r.r
        uc = ctx.solution()
        print(f"{mesh_name} P{order} dirichlet={dirichlet is not None}: CG {a.iters}  CG_SR {b.iters}  CG_FUSED {cf.iters}")
        assert cf.converged == 1 and cf.method_used == capi.SOLVER_CG_FUSED
        assert a.iters - 3 <= cf.iters <= int(1.05 * a.iters) + 3, (a.iters, cf.iters)
        for u in (ua, ub, uc):
            assert np.linalg.norm(u - ref.solution) / np.linalg.norm(ref.solution) <= SOL_TOL


@pytest.mark.parametrize("order", [1, 2])
def test_minimal_meshes_on_device(capi, ctx, oracle, order):
    """edge sizes: a single triangle / tetrahedron and two tetrahedra (fewer rows than one SpMV tile, one assembly block
    mostly empty): assembly, SpMV and a solve without Dirichlet data (-Lap + c is SPD on its own)"""
    tri = (np.array([[0.0, 0.0], [1.0, 0.2], [0.1, 0.9]]), np.array([[0, 1, 2]], dtype=np.int32))
    tet = (np.array([[0.0, 0, 0], [1.0, 0.1, 0], [0, 1.0, 0.2], [0.1, 0.2, 1.0]]), np.array([[0, 1, 2, 3]], dtype=np.int32))
    two = (np.array([[0.0, 0, 0], [1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [1.0, 1.0, 1.0]]), np.array([[0, 1, 2, 3], [4, 2, 1, 3]], dtype=np.int32))
    for nodes, cells in (tri, tet, two):
        m = oracle.Mesh(nodes, cells, np.zeros(nodes.shape[0], dtype=np.uint8))
        ctx.mesh_upload(m.nodes, m.cells, m.boundary)
        nd = ctx.dofs_build(order)
        od, _, _, _ = oracle.enumerate_dofs(m, order)
        op = lambda mod: -mod.laplacian() + mod.reaction(1.0)
        nq = ctx.sizes()["n_quadrature"]
        fq = np.arange(1.0, nq * m.n_cells + 1.0)
        ctx.set_operator(op(capi))
        ctx.set_forcing(fq)
        ctx.set_dirichlet(None)
        ctx.init()
        ref = oracle.pde_init_solve(m, order, op(oracle), forcing_q=fq)
        assert _entry_close(ctx.matrix_values(capi.MAT_STIFF), ref.stiff.values)
        assert _entry_close(ctx.force(), ref.force)
        x = np.arange(1.0, nd + 1.0)
        assert np.abs(ctx.spmv(capi.MAT_STIFF, x) - ref.stiff.matvec(x)).max() <= 1e-12 * np.abs(ref.stiff.matvec(x)).max()
        info = ctx.solve(rtol=1e-12)
        assert info.converged == 1
        assert np.linalg.norm(ctx.solution() - ref.solution) / np.linalg.norm(ref.solution) <= SOL_TOL


def test_lumped_mass_matches_oracle(capi, ctx, oracle, mesh_loader):
    """lump(mass()) (linear_algebra/lumping.h:30-41): device row sums against the oracle's mass matrix, P1 and P2"""
    m = mesh_loader("unit_sphere")
    for order in (1, 2):
        ctx.mesh_upload(m.nodes, m.cells, m.boundary)
        nd = ctx.dofs_build(order)
        ctx.set_operator(-capi.laplacian())
        ctx.set_forcing(np.zeros(ctx.sizes()["n_quadrature"] * m.n_cells))
        ctx.init()
        dofs, _, ond, _ = oracle.enumerate_dofs(m, order)
        M = oracle.assemble_operator(m, order, dofs, ond, oracle.reaction(1.0))
        ref = np.asarray(M.to_scipy().sum(axis=1)).ravel()
        got = ctx.lump(capi.MAT_MASS)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        assert abs(got.sum() - ref.sum()) <= 1e-12 * abs(ref.sum())


def test_partial_dirichlet_boundary_and_solver_prepare(capi, ctx, oracle, mesh_loader):
    """fdapde_dofs_set_boundary: Dirichlet data on part of the boundary (natural condition elsewhere) against the oracle with the
    same mask; fdapde_solver_prepare builds the compact solver layout for the new mask ahead of the solve and is idempotent"""
    m = mesh_loader("unit_square_32")
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(1)
    _, bnd, coords = ctx.dofs_get()
    mask = (bnd.astype(bool) & (coords[:, 0] < 0.5)).astype(np.uint8)
    ctx.dofs_set_boundary(mask)
    ctx.solver_prepare(True)
    ctx.solver_prepare(True)
    qn = ctx.quadrature_nodes()
    fq = 1.0 + qn[:, 0] * qn[:, 1]
    g = np.sin(3.0 * coords[:, 1])
    op = lambda mod: -mod.laplacian() + mod.reaction(0.3)
    ctx.set_operator(op(capi))
    ctx.set_forcing(fq)
    ctx.set_dirichlet(g)
    ctx.init()
    info = ctx.solve(rtol=1e-12)
    assert info.converged == 1
    # oracle: same operator / forcing, Dirichlet rows only where the mask says so
    dofs, _, ond, _ = oracle.enumerate_dofs(m, 1)
    A = oracle.assemble_operator(m, 1, dofs, ond, op(oracle)).to_scipy().tolil()
    b = oracle.assemble_forcing(m, 1, dofs, ond, fq)
    for i in np.nonzero(mask)[0]:
        A.rows[i], A.data[i] = [int(i)], [1.0]
        b[i] = g[i]
    import scipy.sparse.linalg as spla

    ref = spla.spsolve(A.tocsc(), b)
    assert np.linalg.norm(ctx.solution() - ref) / np.linalg.norm(ref) <= SOL_TOL
    ctx.dofs_set_boundary(bnd)   # module-scoped context: restore the reference's mask


@pytest.mark.parametrize("mesh_name,order,n_rhs", [("unit_square", 1, 15), ("unit_sphere", 2, 6), ("unit_square_32", 2, 4)])
def test_multi_rhs_batches_match_column_by_column(capi, ctx, oracle, mesh_loader, mesh_name, order, n_rhs):
    """fdapde_lin_solve with several columns (SMW's A^-1 U): batches of 8 / 4 columns share the matrix stream
    (kernels_multirhs.h); same solutions as the column-by-column path and as a direct solve"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    m = mesh_loader(mesh_name)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    ctx.set_operator(-capi.laplacian() + capi.reaction(1.5))
    ctx.set_forcing(np.zeros(ctx.sizes()["n_quadrature"] * m.n_cells))
    ctx.init()
    ctx.lin_compute(capi.MAT_STIFF, symmetric=True)
    rng = np.random.default_rng(3)
    B = rng.standard_normal((nd, n_rhs))
    B[:, -1] = 0.0                                   # a zero column converges at once and must not disturb the others
    rp, ci = ctx.pattern_get()
    A = sp.csr_matrix((ctx.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd)).tocsc()
    ref = spla.splu(A).solve(B)
    ctx.tune("multi_rhs", 1)
    X1, i1 = ctx.lin_solve(B, rtol=1e-12)
    ctx.tune("multi_rhs", 0)
    X0, i0 = ctx.lin_solve(B, rtol=1e-12)
    ctx.tune("multi_rhs", 1)
    assert i1.converged == 1 and i0.converged == 1
    scale = np.linalg.norm(ref, axis=0).max()
    assert np.abs(X1 - ref).max() <= 1e-8 * scale and np.abs(X0 - ref).max() <= 1e-8 * scale
    assert np.all(X1[:, -1] == 0.0)


def test_lazy_x_update_matches_eager(capi, ctx, oracle, mesh_loader):
    """The fused-update CG touches x every second launch (pending updates are rebuilt from p and r); whatever the parity of the
    converging iteration, and whether convergence is seen inside a launch, at a host poll or never (maxit), the result must be the
    eager one up to rounding"""
    m = mesh_loader("unit_sphere")
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(2)
    _, _, coords = ctx.dofs_get()
    qn = ctx.quadrature_nodes()
    ctx.set_operator(-capi.laplacian() + capi.reaction(0.2))
    ctx.set_forcing(np.cos(qn[:, 0]) + qn[:, 1] * qn[:, 2])
    ctx.set_dirichlet(coords[:, 0] + 0.1)
    ctx.init()
    seen = set()
    cases = [dict(rtol=10.0 ** -k) for k in range(3, 13)] + [dict(rtol=1e-30, maxit=k) for k in (1, 2, 31, 32, 33, 64)]
    for kw in cases:
        sol = {}
        for lazy in (0, 1):
            ctx.tune("cgf_lazy", lazy)
            try:
                info = ctx.solve(method=capi.SOLVER_CG_FUSED, **kw)
            except capi.FdapdeError as e:
                assert e.status == capi.ENOCONV
                info = ctx.info()
            sol[lazy] = (ctx.solution(), info.iters)
        assert sol[0][1] == sol[1][1]
        seen.add(sol[0][1] & 1)
        assert np.abs(sol[0][0] - sol[1][0]).max() <= 1e-13 * np.abs(sol[0][0]).max(), kw
    assert seen == {0, 1}          # both parities of the last executed update were exercised
    ctx.tune("cgf_lazy", 1)


def test_tuning_knobs_do_not_change_results(capi, oracle, mesh_loader):
    """every fdapde_tune knob selects another measured form of the same computation: same iterations, same
    solution up to rounding"""
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_cube(16)
    c = capi.Context(device=0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    _, _, coords = c.dofs_get()
    qn = c.quadrature_nodes()
    c.set_operator(-capi.laplacian() + capi.reaction(0.4))
    c.set_forcing(np.sin(2 * qn[:, 0]) + qn[:, 2])
    c.set_dirichlet(0.2 * coords[:, 1])
    c.init()
    base = c.solve(rtol=1e-11)
    u0 = c.solution()
    defaults = dict(spmv_deep=0, use_graph=0, spmv_c16=1, cgf_band=1, cgf_nt=7, cgf_lazy=1, cgf_v=8, cgf_split=0, spmv_ntv=-1, spmv_bpx=None)
    for key, value in [("spmv_deep", 1), ("use_graph", 1), ("spmv_c16", 0), ("cgf_band", 0), ("cgf_nt", 0), ("cgf_lazy", 0), ("cgf_v", 2), ("cgf_split", 1),
                       ("spmv_ntv", 1), ("spmv_ablate", 150), ("spmv_ablate", 151), ("spmv_ablate", 152), ("spmv_ablate", 3)]:
        c.tune(key, value)
        info = c.solve(rtol=1e-11)
        assert info.converged == 1 and abs(info.iters - base.iters) <= 1, (key, value, info.iters, base.iters)
        assert np.abs(c.solution() - u0).max() <= 1e-10 * np.abs(u0).max(), (key, value)
        c.tune(key, 0 if key == "spmv_ablate" else defaults[key])
    c.close()


@pytest.mark.parametrize("mesh_name", ["unit_square_16", "c_shaped", "unit_sphere", "cube8", "cube14"])
def test_visit_parallel_assembly_gives_the_row_walking_bits(capi, oracle, mesh_loader, mesh_name):
    """k_assemble_items (P2 spaces: the (row, visit) pairs of a block are the work items of 1024 threads) adds, into every slot, the
    addends of the row-walking sweep in the same order: stiff_, mass_ and force_ must be the same BITS with the knob on and off -- for the
    Laplacian (+ mass as second sweep), a constant-coefficient advection-diffusion-reaction operator, space-varying coefficients, the
    mass matrix alone and forcing-only sweeps (second forcing column) -- and the oracle's numbers within the entry tolerance."""
    if mesh_name.startswith("cube"):
        from fdapde_core_amd import meshgen

        m = oracle.Mesh(*meshgen.unit_cube(int(mesh_name[4:])))   # jittered, permuted Kuhn tetrahedra: 4913 / 24389 P2 DOFs, many blocks
    else:
        m = mesh_loader(mesh_name)
    c = capi.Context(device=0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(2)
    qn = c.quadrature_nodes()
    rng = np.random.default_rng(7)
    fq = np.stack([np.sin(qn.sum(axis=1)), rng.standard_normal(qn.shape[0])], axis=1)
    b = np.array([0.7, -0.2, 0.4][:m.M])
    cq = 1.0 + qn[:, 0] ** 2
    ops = {"lap": lambda mod: -mod.laplacian(), "adr": lambda mod: -mod.laplacian() + mod.advection(b) + mod.reaction(1.5),
           "field": lambda mod: -mod.laplacian() + mod.reaction_field(cq)}
    od, _, _, _ = oracle.enumerate_dofs(m, 2)
    for name, mk in ops.items():
        got = {}
        for items in (0, 1):
            c.tune("asm_items", items)
            c.set_operator(mk(capi))
            c.set_forcing(fq)
            c.init()
            got[items] = (c.matrix_values(capi.MAT_STIFF), c.matrix_values(capi.MAT_MASS), c.force(ncols=2))
            c.assemble_operator(capi.MAT_MASS, capi.reaction(1.0))   # OPK 2 on its own
            got[items] += (c.matrix_values(capi.MAT_MASS),)
        for x, y in zip(got[0], got[1]):
            assert np.array_equal(x, y), name
        ref = oracle.assemble_operator(m, 2, od, nd, mk(oracle))
        assert _entry_close(got[1][0], ref.values), name
        assert _entry_close(got[1][2][:nd], oracle.assemble_forcing(m, 2, od, nd, fq[:, 0])), name
    c.close()
