"""Pins the CPU oracle (oracle/fem_oracle.c) against every golden vector, known-answer test and fixture the
reference's own tests hold for the assemble-and-solve path.  CPU-only.

Each test names the reference test it re-expresses.  Tolerances are the reference's own:
DOUBLE_TOLERANCE = 1e-7 (test/src/utils/constants.h:11) with almost_equal = abs-or-relative
(test/src/utils/utils.h:33-36); the oracle actually agrees far tighter and the tighter bound is asserted too.
"""
import os

import numpy as np
import pytest

DOUBLE_TOLERANCE = 1e-7


def almost_equal(a, b, eps=DOUBLE_TOLERANCE):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    d = np.abs(a - b)
    return bool(np.all((d < eps) | (d < np.maximum(np.abs(a), np.abs(b)) * eps)))


def almost_equal_mat(a, b, eps=DOUBLE_TOLERANCE):
    d = np.abs(a - b).max()
    return d < eps or d < max(np.abs(a).max(), np.abs(b).max()) * eps


# ---------------------------------------------------------------------------------------------- golden numbers
# test/src/fem_operators_test.cpp:83-96 (c_shaped cell 175, P2, L = -laplacian, row-major over (i, j))
GOLDEN_LOCAL_P2 = np.array([
    0.7043890316492852, 0.1653830261033185, 0.0694133177797771, -0.6615321044132733, -0.2776532711191089, 0.0000000000000013,
    0.1653830261033185, 0.7043890316492852, 0.0694133177797769, -0.6615321044132735, 0.0000000000000003, -0.2776532711191076,
    0.0694133177797771, 0.0694133177797769, 0.4164799066786617, 0.0000000000000002, -0.2776532711191083, -0.2776532711191075,
    -0.6615321044132733, -0.6615321044132735, 0.0000000000000002, 2.4336772933029756, -0.5553065422382126, -0.5553065422382162,
    -0.2776532711191089, 0.0000000000000003, -0.2776532711191083, -0.5553065422382126, 2.4336772933029738, -1.3230642088265447,
    0.0000000000000013, -0.2776532711191075, -0.2776532711191076, -0.5553065422382162, -1.3230642088265447, 2.4336772933029751,
]).reshape(6, 6)
# test/src/lagrangian_basis_test.cpp:158-161
GOLDEN_GRAD_P1 = np.array([[-5.2557081783567776, -3.5888585000106943], [6.2494499783110298, -1.2028086954015513],
                           [-0.9937417999542519, 4.7916671954122458]])
# test/src/lagrangian_basis_test.cpp:185-187
GOLDEN_GRAD_P2 = np.array([[2.9830765115928704, 2.0369927574935405], [4.8982811692194259, -0.9427541948981384],
                           [-0.7788888242446018, 3.7556798236502558], [-6.6727629051483941, -6.9218931297696376],
                           [-9.8048064747509027, -4.3298093852388320], [9.3751005233316018, 6.4017841287628112]])


def test_fem_operators_laplacian_order_2(oracle, mesh_loader):
    """fem_operators_test.cpp:41-100"""
    m = mesh_loader("c_shaped")
    A = oracle.local_matrix(m, 2, 175, -oracle.laplacian())
    assert almost_equal(A, GOLDEN_LOCAL_P2)
    assert np.abs(A - GOLDEN_LOCAL_P2).max() < 5e-15


def test_lagrangian_physical_gradients(oracle, mesh_loader):
    """lagrangian_basis_test.cpp:150-197: invJ^T grad(psi_i) at node 0 of the 6-point rule, orders 1 and 2"""
    m = mesh_loader("c_shaped")
    qn, _ = oracle.quadrature(2, nq=6)
    g1 = oracle.physical_gradients_at(m, 1, 175, qn[0])
    g2 = oracle.physical_gradients_at(m, 2, 175, qn[0])
    assert almost_equal(g1, GOLDEN_GRAD_P1) and np.abs(g1 - GOLDEN_GRAD_P1).max() < 1e-13
    assert almost_equal(g2, GOLDEN_GRAD_P2) and np.abs(g2 - GOLDEN_GRAD_P2).max() < 1e-13


def test_lagrangian_reference_gradients(oracle):
    """lagrangian_basis_test.cpp:104-147"""
    c1 = oracle.reference_basis(2, 1)
    exp1 = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    for i in range(3):
        assert almost_equal(oracle.poly_grad(2, 1, c1[i], [0.0, 0.0]), exp1[i])
    c2 = oracle.reference_basis(2, 2)
    p = np.array([0.5, 0.5])
    exp2 = np.array([
        [1 - 4 * (1 - p[0] - p[1]), 1 - 4 * (1 - p[0] - p[1])], [4 * p[0] - 1, 0], [0, 4 * p[1] - 1],
        [4 * (1 - 2 * p[0] - p[1]), -4 * p[0]], [-4 * p[1], 4 * (1 - p[0] - 2 * p[1])], [4 * p[1], 4 * p[0]],
    ])
    for i in range(6):
        assert almost_equal(oracle.poly_grad(2, 2, c2[i], p), exp2[i])


@pytest.mark.parametrize("M,R", [(2, 1), (3, 1), (2, 2), (3, 2)])
def test_lagrangian_reference_element_support(oracle, M, R):
    """lagrangian_basis_test.cpp:80-101: Lagrange property, incl. the 3-D P1/P2 reference elements"""
    coeff = oracle.reference_basis(M, R)
    nodes = oracle.reference_nodes(M, R)
    nb = oracle.n_basis(M, R)
    assert nb == {(2, 1): 3, (3, 1): 4, (2, 2): 6, (3, 2): 10}[(M, R)]
    V = np.array([[oracle.poly_eval(M, R, coeff[b], nodes[i]) for i in range(nb)] for b in range(nb)])
    assert np.abs(V - np.eye(nb)).max() < 1e-13


def test_poly_table_order(oracle):
    """multivariate_polynomial.h:52-79: digit 0 runs fastest"""
    assert oracle.poly_table(2, 2).tolist() == [[0, 0], [1, 0], [2, 0], [0, 1], [1, 1], [0, 2]]
    assert oracle.poly_table(3, 1).tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]


# ---------------------------------------------------------------------------------------------- integer invariants
# SURVEY.md section 8c table (computed from the CSVs with the reference's enumeration rules)
INVARIANTS = {
    # mesh: (nodes, cells, edges, boundary nodes, P1 nnz, P2 n_dofs)
    "unit_square_16": (289, 512, 800, 64, 1889, 1089),
    "unit_square_32": (1089, 2048, 3136, 128, 7361, 4225),
    "unit_square_64": (4225, 8192, 12416, 256, 29057, 16641),
    "unit_square": (3600, 6962, 10561, 236, 24722, 14161),
    "c_shaped": (264, 418, 681, 108, 1626, 945),
    "quasi_circle": (341, 630, 970, 50, 2281, 1311),
    "unit_sphere": (587, 2775, 3606, 247, 7799, 4193),
}


@pytest.mark.parametrize("name", sorted(INVARIANTS))
def test_fixture_integer_invariants(oracle, mesh_loader, name):
    nodes, cells, edges, bnodes, nnz1, ndof2 = INVARIANTS[name]
    m = mesh_loader(name)
    assert (m.n_nodes, m.n_cells, int(m.boundary.sum())) == (nodes, cells, bnodes)
    d1, b1, n1, _ = oracle.enumerate_dofs(m, 1)
    assert n1 == nodes and np.array_equal(d1, m.cells) and np.array_equal(b1, m.boundary)
    d2, b2, n2, ne = oracle.enumerate_dofs(m, 2)
    assert (ne, n2) == (edges, ndof2)
    assert np.array_equal(d2[:, : m.M + 1], m.cells)
    A = oracle.assemble_operator(m, 1, d1, n1, -oracle.laplacian())
    assert A.rowptr[-1] == nnz1
    if name == "unit_square":
        assert int(b2.sum()) == 472  # 236 boundary nodes + 236 boundary edges
        A2 = oracle.assemble_operator(m, 2, d2, n2, -oracle.laplacian())
        assert A2.rowptr[-1] == 161071


def test_p2_edge_dofs_first_seen_order(oracle, mesh_loader):
    """lagrangian_basis.h:105-128 + triangulation.h:150-193: edge ids are first-seen over cells ascending x
    (0,1),(0,2),(1,2); slot = 3 + index of the pair"""
    m = mesh_loader("unit_square_16")
    d2, _, _, _ = oracle.enumerate_dofs(m, 2)
    seen, nxt = {}, m.n_nodes
    for c in range(m.n_cells):
        for j, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
            key = tuple(sorted((int(m.cells[c, a]), int(m.cells[c, b]))))
            if key not in seen:
                seen[key] = nxt
                nxt += 1
            assert d2[c, 3 + j] == seen[key]


# ---------------------------------------------------------------------------------------------- golden Psi matrices
@pytest.mark.parametrize("order", [1, 2])
def test_lagrangian_pointwise_evaluation(oracle, mesh_loader, golden_dir, order):
    """lagrangian_basis_test.cpp:200-208, 222-229: pins the P2 global DOF numbering (945 columns)"""
    m = mesh_loader("c_shaped")
    locs = oracle.read_csv(os.path.join(golden_dir, "mesh", "c_shaped", "locs.csv"))
    dofs, _, nd, _ = oracle.enumerate_dofs(m, order)
    psi = oracle.pointwise_psi(m, order, dofs, nd, locs)
    gold = oracle.read_mtx(os.path.join(golden_dir, "mtx", f"lagrangian_pointwise_eval_order{order}.mtx"))
    assert gold.shape == psi.shape
    assert almost_equal_mat(psi, gold)
    assert np.abs(psi - gold).max() < 1e-12


@pytest.mark.parametrize("order", [1, 2])
def test_lagrangian_areal_evaluation(oracle, mesh_loader, golden_dir, order):
    """lagrangian_basis_test.cpp:211-219, 232-238: pins numbering + quadrature on a second mesh (1311 columns)"""
    m = mesh_loader("quasi_circle")
    inc = oracle.read_csv(os.path.join(golden_dir, "mesh", "quasi_circle", "incidence_matrix.csv"))
    dofs, _, nd, _ = oracle.enumerate_dofs(m, order)
    psi, D = oracle.areal_psi(m, order, dofs, nd, inc)
    gold = oracle.read_mtx(os.path.join(golden_dir, "mtx", f"lagrangian_areal_eval_order{order}.mtx"))
    assert gold.shape == psi.shape
    assert almost_equal_mat(psi, gold)
    assert np.abs(psi - gold).max() < 1e-12


# ---------------------------------------------------------------------------------------------- integration
def test_integration_identities(oracle, mesh_loader):
    """integration_test.cpp:46-80: int 1 = measure; int linear = measure * mean(vertex values); int_{unit_square} 1 = 1"""
    for name in ("unit_square", "unit_sphere"):
        m = mesh_loader(name)
        qn, qw = oracle.quadrature(m.M, 1)
        rng = np.random.default_rng(0)
        total = 0.0
        for c in rng.integers(0, m.n_cells, 20):
            J, invJ, meas = oracle.cell_geometry(m, int(c))
            x0 = m.nodes[m.cells[c, 0]]
            pts = qn @ J.T + x0
            assert almost_equal(meas, np.sum(qw * 1.0) * meas)
            f = lambda x: x[0] + x[1]
            lin = sum(f(p) * w for p, w in zip(pts, qw)) * meas
            h = sum(f(m.nodes[v]) for v in m.cells[c])
            assert almost_equal(meas * h / (m.M + 1), lin)
        if name == "unit_square":
            for c in range(m.n_cells):
                total += oracle.cell_geometry(m, c)[2] * np.sum(qw)
            assert almost_equal(1.0, total)


@pytest.mark.parametrize("M,rules", [(2, [3, 6]), (3, [4, 5, 11])])
def test_quadrature_rules_agree_on_linear(oracle, M, rules):
    """integration_test.cpp:112-126: every table integrates the first P1 basis function to the same value"""
    c = oracle.reference_basis(M, 1)[0]
    res = []
    for nq in rules:
        qn, qw = oracle.quadrature(M, nq=nq)
        res.append(sum(oracle.poly_eval(M, 1, c, p) * w for p, w in zip(qn, qw)) / M)
    for i in range(len(res)):
        for j in range(i + 1, len(res)):
            assert almost_equal(res[i], res[j])


def test_simplex_tetrahedron_measure(oracle):
    """simplex_test.cpp:89-97"""
    nodes = np.array([[0.0, 0.0, 0.0], [0.4, 0.2, 0.0], [0.0, 0.8, 0.6], [0.4, 0.6, 0.8]])
    m = oracle.Mesh(nodes, np.array([[0, 1, 2, 3]], dtype=np.int32), np.zeros(4, dtype=np.uint8))
    J, invJ, meas = oracle.cell_geometry(m, 0)
    assert almost_equal(meas, 0.0266666666666666)
    assert np.abs(J @ invJ - np.eye(3)).max() < 1e-14
    assert almost_equal(J @ np.full(3, 0.25) + nodes[0], [0.2, 0.4, 0.35])  # barycenter


def test_unit_sphere_negative_orientation_is_load_bearing(oracle, mesh_loader):
    """1395 of the 2775 tets have det J < 0 -- std::abs at simplex.h:188 matters"""
    m = mesh_loader("unit_sphere")
    neg = sum(np.linalg.det(oracle.cell_geometry(m, c)[0]) < 0 for c in range(m.n_cells))
    assert neg == 1395
    assert all(oracle.cell_geometry(m, c)[2] > 0 for c in range(0, m.n_cells, 37))


# ---------------------------------------------------------------------------------------------- full path (fem_pde_test)
def _l2_error(sol, exact):
    err = exact - sol.solution
    return float(np.sum(sol.mass.matvec(err * err)))


def test_pde_laplacian_isotropic_order1(oracle, mesh_loader):
    """fem_pde_test.cpp:43-75: u = x + y, f = 0 sampled at quadrature nodes, Dirichlet everywhere; < 1e-7"""
    m = mesh_loader("unit_square")
    u = lambda x: x[0] + x[1]
    sol = oracle.pde_init_solve(m, 1, -oracle.laplacian(), forcing_fn=None, dirichlet=u)
    exact = np.array([u(p) for p in sol.dof_coords])
    assert _l2_error(sol, exact) < DOUBLE_TOLERANCE


def test_pde_laplacian_isotropic_order2_callable_force(oracle, mesh_loader):
    """fem_pde_test.cpp:78-107: u = 1 - x^2 - y^2, f = 4; < 1e-7"""
    m = mesh_loader("unit_square")
    u = lambda x: 1.0 - x[0] * x[0] - x[1] * x[1]
    sol = oracle.pde_init_solve(m, 2, -oracle.laplacian(), forcing_fn=lambda x: 4.0, dirichlet=u)
    exact = np.array([u(p) for p in sol.dof_coords])
    assert sol.n_dofs == 14161
    assert _l2_error(sol, exact) < DOUBLE_TOLERANCE


def _advdiff_exact():
    pi = np.pi
    alpha, gamma = 1.0, pi
    l1 = -alpha / 2 - np.sqrt((alpha / 2) ** 2 + pi * pi)
    l2 = -alpha / 2 + np.sqrt((alpha / 2) ** 2 + pi * pi)
    p = (1 - np.exp(l2)) / (np.exp(l1) - np.exp(l2))
    u = lambda x: -gamma / (pi * pi) * (p * np.exp(l1 * x[0]) + (1 - p) * np.exp(l2 * x[0]) - 1.0) * np.sin(pi * x[1])
    f = lambda x: gamma * np.sin(pi * x[1])
    return u, f, np.array([-alpha, 0.0])


def test_pde_advection_diffusion_order1(oracle, mesh_loader):
    """fem_pde_test.cpp:113-166: non-symmetric -Lap + b.grad, b = (-1, 0); < 1e-5.  The gate discriminates the
    row = test function / column = trial function orientation."""
    m = mesh_loader("unit_square")
    u, f, beta = _advdiff_exact()
    op = -oracle.laplacian() + oracle.advection(beta)
    sol = oracle.pde_init_solve(m, 1, op, forcing_fn=f, dirichlet=lambda x: 0.0)
    exact = np.array([u(p) for p in sol.dof_coords])
    e = _l2_error(sol, exact)
    assert e < 1e-5
    # transposed orientation must fail the same gate
    import scipy.sparse.linalg as spla

    dofs, bnd, nd, _ = oracle.enumerate_dofs(m, 1)
    A = oracle.assemble_operator(m, 1, dofs, nd, op)
    At = A.to_scipy().T.tocsr()
    At.sort_indices()
    T = oracle.CSR(At.indptr.astype(np.int32), At.indices.astype(np.int32), At.data.copy(), nd)
    b = sol.force.copy()
    oracle.set_dirichlet(T, b, bnd, np.zeros(nd))
    ut = spla.splu(T.to_scipy().tocsc()).solve(b)
    err = exact - ut
    assert float(np.sum(sol.mass.matvec(err * err))) > 1e-5


def test_pde_advection_diffusion_order2(oracle, mesh_loader):
    """fem_pde_test.cpp:172-212: same problem, P2, callable forcing; < 1e-7"""
    m = mesh_loader("unit_square")
    u, f, beta = _advdiff_exact()
    op = -oracle.laplacian() + oracle.advection(beta)
    sol = oracle.pde_init_solve(m, 2, op, forcing_fn=f, dirichlet=lambda x: 0.0)
    exact = np.array([u(p) for p in sol.dof_coords])
    assert _l2_error(sol, exact) < DOUBLE_TOLERANCE


def test_pde_3d_p1_linear_exact(oracle, mesh_loader):
    """Build-added 3-D pin (the reference has no 3-D FEM test, F7): u = x + y + z is in the P1 space, so the
    discrete solution reproduces it to rounding on unit_sphere."""
    m = mesh_loader("unit_sphere")
    u = lambda x: x[0] + x[1] + x[2]
    sol = oracle.pde_init_solve(m, 1, -oracle.laplacian(), dirichlet=u)
    exact = np.array([u(p) for p in sol.dof_coords])
    assert np.abs(sol.solution - exact).max() < 1e-12


def test_pde_3d_p2_quadratic_exact(oracle, mesh_loader):
    """Build-defined 3-D P2 numbering (unpinned by the reference): u = x^2 + y z is in the P2 space and the 5-point
    rule integrates the P2 stiffness exactly, so -Lap u = -2 is reproduced to rounding.  A wrong edge-slot map
    breaks conformity and this test."""
    m = mesh_loader("unit_sphere")
    u = lambda x: x[0] * x[0] + x[1] * x[2]
    sol = oracle.pde_init_solve(m, 2, -oracle.laplacian(), forcing_fn=lambda x: -2.0, dirichlet=u)
    exact = np.array([u(p) for p in sol.dof_coords])
    assert sol.n_dofs == 4193
    assert np.abs(sol.solution - exact).max() < 1e-10


# ---------------------------------------------------------------------------------------------- oracle self-consistency
def test_symmetric_assembly_is_exactly_symmetric(oracle, mesh_loader):
    """fem_assembler.h:94-102,116-117: lower triangle computed, mirrored by selfadjointView<Lower>"""
    m = mesh_loader("unit_square_32")
    dofs, _, nd, _ = oracle.enumerate_dofs(m, 2)
    A = oracle.assemble_operator(m, 2, dofs, nd, -oracle.laplacian() + oracle.reaction(2.0)).to_scipy()
    assert (A - A.T).nnz == 0 or abs(A - A.T).max() == 0.0


def test_krylov_matches_direct(oracle, mesh_loader):
    """The iterative solves on the interior block give the reference's row-zeroed LU solution (SURVEY 7 hard parts)"""
    m = mesh_loader("unit_square_32")
    u = lambda x: np.sin(np.pi * x[0]) * np.sin(np.pi * x[1]) + x[0]
    f = lambda x: 2 * np.pi**2 * np.sin(np.pi * x[0]) * np.sin(np.pi * x[1])
    a = oracle.pde_init_solve(m, 1, -oracle.laplacian(), forcing_fn=f, dirichlet=u, direct=True)
    b = oracle.pde_init_solve(m, 1, -oracle.laplacian(), forcing_fn=f, dirichlet=u, direct=False)
    assert np.linalg.norm(a.solution - b.solution) / np.linalg.norm(a.solution) < 1e-10
    _, fadv, beta = _advdiff_exact()
    op = -oracle.laplacian() + oracle.advection(beta) + oracle.reaction(0.5)
    a = oracle.pde_init_solve(m, 2, op, forcing_fn=fadv, dirichlet=u, direct=True)
    b = oracle.pde_init_solve(m, 2, op, forcing_fn=fadv, dirichlet=u, direct=False)
    assert np.linalg.norm(a.solution - b.solution) / np.linalg.norm(a.solution) < 1e-9
    assert np.array_equal(a.stiff.values, b.stiff.values)


def test_space_varying_coefficients_match_constants(oracle, mesh_loader):
    """fields forward(i) row indexing, integrator.h:98-101: row nq*cell + q of row-major coefficient data"""
    m = mesh_loader("unit_square_16")
    for order in (1, 2):
        dofs, _, nd, _ = oracle.enumerate_dofs(m, order)
        nq = oracle.quadrature(2, order)[1].size
        rows = nq * m.n_cells
        K = np.array([[2.0, 0.3], [0.3, 1.0]])
        b = np.array([0.7, -0.2])
        const = oracle.diffusion(K) + oracle.advection(b) + oracle.reaction(1.5)
        var = (oracle.diffusion_field(np.tile(K.reshape(1, 4), (rows, 1))) + oracle.advection_field(np.tile(b, (rows, 1)))
               + oracle.reaction_field(np.full(rows, 1.5)))
        A = oracle.assemble_operator(m, order, dofs, nd, const)
        B = oracle.assemble_operator(m, order, dofs, nd, var)
        assert np.array_equal(A.colidx, B.colidx) and np.array_equal(A.values, B.values)


def _parabolic_problem(o, m, order, n_times):
    pi = np.pi
    times = np.linspace(0.0, 1.0, n_times)
    u = lambda x, t: np.sin(2 * pi * x[:, 0]) * np.sin(2 * pi * x[:, 1]) * np.exp(-t)
    f = lambda x, t: (8 * pi * pi - 1.0) * np.sin(2 * pi * x[:, 0]) * np.sin(2 * pi * x[:, 1]) * np.exp(-t)
    dofs, _, nd, _ = o.enumerate_dofs(m, order)
    coords = o.dofs_coords(m, order, dofs, nd)
    qn = o.quadrature_nodes(m, order)
    F = np.stack([f(qn, t) for t in times], axis=1)
    G = np.stack([u(coords, t) for t in times], axis=1)
    return times, F, G, G[:, 0].copy()


def test_pde_parabolic_isotropic_order2(oracle, mesh_loader):
    """fem_pde_test.cpp:222-285: dt(u) - Lap(u) = f, P2, 101 time points, max_t sum(M err^2) < 1e-7"""
    m = mesh_loader("unit_square")
    times, F, G, u0 = _parabolic_problem(oracle, m, 2, 101)
    sol, Mm = oracle.pde_parabolic_solve(m, 2, oracle.dt() - oracle.laplacian(), times, F, G, u0)
    errs = [float(np.sum(Mm.matvec((G[:, j] - sol[:, j]) ** 2))) for j in range(times.size)]
    assert max(errs) < 1e-7


def test_pde_parabolic_order1_convergence(oracle, mesh_loader):
    """fem_pde_test.cpp:295-368: P1, 31 time points, meshes 16/32/64 (the 128 fixture is 3 MB and not copied): order 2"""
    errs = []
    for name in ("unit_square_16", "unit_square_32", "unit_square_64"):
        m = mesh_loader(name)
        times, F, G, u0 = _parabolic_problem(oracle, m, 1, 31)
        sol, Mm = oracle.pde_parabolic_solve(m, 1, oracle.dt() - oracle.laplacian(), times, F, G, u0)
        errs.append(np.sqrt(float(np.sum(Mm.matvec((G[:, -1] - sol[:, -1]) ** 2)))))
    for a, b in zip(errs[:-1], errs[1:]):
        assert np.floor(np.log2(a / b)) == 2
