"""Randomised operator expressions through fdapde_init (stiff_ + mass_ + force_ in one call) against the CPU oracle: random mesh (generated, jittered, ids
permuted; 2-D / 3-D), order, and a random sum of terms -- Laplacian, diffusion (constant symmetric / constant non-symmetric / field), advection (constant /
field), reaction (constant / field), dt -- with random signs and scales.  What the fixtures' fixed operator list cannot reach: every combination of
"constant through the reference tensors" and "per-node integrand" the operator classifier (OPK) can be handed.  Entries to 1e-12 / 1e-13 relative (the
parity suite's bar)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENTRY_TOL = 1e-12
ENTRY_RTOL = 1e-13


@pytest.fixture(scope="module")
def capi():
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi as m

    assert m.load().fdapde_device_count() >= 1
    return m


def _close(a, b):
    amax = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max()
    return err <= ENTRY_TOL * max(1.0, amax) and err <= ENTRY_RTOL * amax, err / amax


def _random_operator(mod_list, rng, N, rows):
    """the same random expression built for every module in mod_list (capi, oracle)"""
    terms = []   # (kind, scale, data)
    n_terms = int(rng.integers(1, 5))
    kinds = list(rng.choice(["lap", "diffc", "diffn", "difff", "advc", "advf", "reac", "reaf", "dt"], size=n_terms, replace=True))
    if not any(k in ("lap", "diffc", "diffn", "difff") for k in kinds):
        kinds.append("lap")
    for k in kinds:
        scale = float(rng.choice([-1.0, 1.0]) * rng.uniform(0.3, 2.0))
        if k == "diffc":
            L = rng.uniform(-0.4, 0.4, (N, N))
            data = L @ L.T + np.eye(N)
        elif k == "diffn":
            data = np.eye(N) + rng.uniform(-0.4, 0.4, (N, N))
        elif k == "difff":
            L = rng.uniform(-0.3, 0.3, (rows, N, N))
            K = np.einsum("qij,qkj->qik", L, L) + np.eye(N)[None]
            if rng.integers(0, 2):
                K = K + rng.uniform(-0.2, 0.2, (rows, N, N))   # not symmetric
            data = K.reshape(rows, N * N)
        elif k == "advc":
            data = rng.uniform(-1.0, 1.0, N)
        elif k == "advf":
            data = rng.uniform(-1.0, 1.0, (rows, N))
        elif k == "reac":
            data = float(rng.uniform(0.1, 2.0))
        elif k == "reaf":
            data = rng.uniform(0.1, 2.0, rows)
        else:
            data = None
        terms.append((k, scale, data))
    out = []
    for mod in mod_list:
        op = None
        for k, scale, data in terms:
            t = {"lap": lambda: mod.laplacian(), "diffc": lambda: mod.diffusion(data), "diffn": lambda: mod.diffusion(data),
                 "difff": lambda: mod.diffusion_field(data), "advc": lambda: mod.advection(data), "advf": lambda: mod.advection_field(data),
                 "reac": lambda: mod.reaction(data), "reaf": lambda: mod.reaction_field(data), "dt": lambda: mod.dt()}[k]()
            t = scale * t
            op = t if op is None else op + t
        out.append(op)
    return out, "+".join(k for k, _, _ in terms)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FDAPDE_FUZZ_SEEDS", "24"))))   # (FDAPDE_FUZZ_SEEDS=400 for a long run)
def test_random_operator_expression_against_the_oracle(capi, oracle, seed):
    from fdapde_core_amd import meshgen

    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.integers(2, 4))
    order = int(rng.integers(1, 3))
    nx = int(rng.integers(3, 14)) if dim == 2 else int(rng.integers(2, 6))
    nodes, cells, bnd = meshgen.unit_square(nx, seed=seed + 1) if dim == 2 else meshgen.unit_cube(nx, seed=seed + 1)
    m = oracle.Mesh(np.ascontiguousarray(nodes), np.ascontiguousarray(cells), np.ascontiguousarray(bnd))
    c = capi.Context(0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(order)
    od, _, ond, _ = oracle.enumerate_dofs(m, order)
    assert nd == ond
    rows = c.sizes()["n_quadrature"] * m.n_cells
    (op_c, op_o), what = _random_operator([capi, oracle], rng, m.N, rows)
    fq = rng.standard_normal(rows)
    c.set_operator(op_c)
    c.set_forcing(fq)
    c.init()
    ref = oracle.assemble_operator(m, order, od, nd, op_o)
    ok, rel = _close(c.matrix_values(capi.MAT_STIFF), ref.values)
    assert ok, (what, dim, order, nx, rel)
    mass = oracle.assemble_operator(m, order, od, nd, oracle.reaction(1.0))
    ok, rel = _close(c.matrix_values(capi.MAT_MASS), mass.values)
    assert ok, ("mass", what, dim, order, nx, rel)
    ok, rel = _close(c.force(), oracle.assemble_forcing(m, order, od, nd, fq))
    assert ok, ("force", what, dim, order, nx, rel)
    c.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FDAPDE_FUZZ_SEEDS_E2E", "16"))))
def test_random_pde_end_to_end_against_the_oracle(capi, oracle, seed):
    """PDE::init + solve (elliptic, seeds 0-9 of every 16) or the parabolic stepper (seeds 10-15) with a random operator expression, random forcing and random Dirichlet
    data against the oracle's init + direct solve (pde/pde.h:101-105, fem_linear_elliptic_solver.h:34-50 / fem_linear_parabolic_solver.h:37-72): 1e-8"""
    from fdapde_core_amd import meshgen

    rng = np.random.default_rng(5000 + seed)
    dim = int(rng.integers(2, 4))
    order = int(rng.integers(1, 3))
    nx = int(rng.integers(3, 12)) if dim == 2 else int(rng.integers(2, 5))
    nodes, cells, bnd = meshgen.unit_square(nx, seed=seed + 7) if dim == 2 else meshgen.unit_cube(nx, seed=seed + 7)
    m = oracle.Mesh(np.ascontiguousarray(nodes), np.ascontiguousarray(cells), np.ascontiguousarray(bnd))
    c = capi.Context(0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(order)
    _, _, coords = c.dofs_get()
    rows = c.sizes()["n_quadrature"] * m.n_cells
    # a coercive expression: diffusion-type leaves with a positive sign, reaction >= 0, mild advection
    K = np.eye(m.N) + 0.3 * np.diag(rng.uniform(0, 1, m.N))
    if rng.integers(0, 2):
        K = K + rng.uniform(-0.2, 0.2, (m.N, m.N))   # not symmetric: the mirrored lower triangle without advection, the full form with it
    pieces = [lambda mod: -mod.diffusion(K), lambda mod: -mod.laplacian()]
    if rng.integers(0, 2):
        cq = rng.uniform(0.1, 2.0, rows)
        pieces.append(lambda mod: mod.reaction_field(cq))
    else:
        creact = float(rng.uniform(0.0, 2.0))
        pieces.append(lambda mod: mod.reaction(creact))
    if rng.integers(0, 2):
        b = rng.uniform(-0.5, 0.5, m.N)
        pieces.append(lambda mod: mod.advection(b))

    def build(mod, with_dt):
        op = pieces[0](mod)
        for p in pieces[1:]:
            op = op + p(mod)
        return mod.dt() + op if with_dt else op

    if seed % 16 < 10:
        fq = rng.standard_normal(rows)
        g = coords @ rng.uniform(-1, 1, m.N) + 0.2
        c.set_operator(build(capi, False))
        c.set_forcing(fq)
        c.set_dirichlet(g)
        c.init()
        info = c.solve(rtol=1e-12)
        ref = oracle.pde_init_solve(m, order, build(oracle, False), forcing_q=fq, dirichlet=g, direct=True)
        assert info.converged == 1
        assert np.linalg.norm(c.solution() - ref.solution) <= 1e-8 * np.linalg.norm(ref.solution), (dim, order, nx)
        # what the reference leaves behind in stiff_ / force_ after set_dirichlet_bc (fem_solver_base.h:142-155): rows of boundary DOFs zeroed, unit diagonal, g
        ok, rel = _close(c.matrix_values(capi.MAT_STIFF), ref.stiff.values)
        assert ok, ("stiff after the solve", dim, order, nx, rel)
        ok, rel = _close(c.force(), ref.force)
        assert ok, ("force after the solve", dim, order, nx, rel)
        ok, rel = _close(c.matrix_values(capi.MAT_MASS), ref.mass.values)
        assert ok, ("mass", dim, order, nx, rel)
    else:
        mt = int(rng.integers(3, 7))
        times = np.linspace(0.0, 0.04 * (mt - 1), mt)
        F = rng.standard_normal((rows, mt))
        G = np.tile((coords @ rng.uniform(-1, 1, m.N))[:, None], (1, mt)) * np.linspace(0.5, 1.0, mt)[None, :]
        u0 = G[:, 0].copy()
        c.set_operator(build(capi, True))
        c.set_forcing(F)
        c.init()
        sol, info = c.solve_parabolic(times, u0, G, rtol=1e-12)
        ref, _ = oracle.pde_parabolic_solve(m, order, build(oracle, True), times, F, G, u0)
        assert info.converged == 1
        for j in range(1, mt):
            assert np.linalg.norm(sol[:, j] - ref[:, j]) <= 1e-8 * np.linalg.norm(ref[:, j]), (dim, order, nx, j)
    c.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FDAPDE_FUZZ_SEEDS_PSI", "12"))))
def test_random_basis_evaluation_against_the_oracle(capi, oracle, seed):
    """PDE__::eval_basis (pde.h:149-158; lagrangian_basis.h:203-283) on generated meshes: random interior points, vertices, edge midpoints, points outside;
    random subdomain incidence matrices.  Psi rows against the oracle's (a point on a facet may be located in either cell: the basis is continuous there)."""
    from fdapde_core_amd import meshgen

    rng = np.random.default_rng(9000 + seed)
    dim = int(rng.integers(2, 4))
    order = int(rng.integers(1, 3))
    nx = int(rng.integers(2, 12)) if dim == 2 else int(rng.integers(2, 5))
    nodes, cells, bnd = meshgen.unit_square(nx, seed=seed + 3) if dim == 2 else meshgen.unit_cube(nx, seed=seed + 3)
    m = oracle.Mesh(np.ascontiguousarray(nodes), np.ascontiguousarray(cells), np.ascontiguousarray(bnd))
    c = capi.Context(0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(order)
    od, _, ond, _ = oracle.enumerate_dofs(m, order)
    n_in = 200
    cid = rng.integers(0, m.n_cells, n_in)
    w = rng.dirichlet(np.ones(m.M + 1), n_in)
    inside = np.einsum("ij,ijk->ik", w, m.nodes[m.cells[cid]])
    verts = m.nodes[rng.integers(0, m.n_nodes, 30)]
    e = m.cells[rng.integers(0, m.n_cells, 30)]
    mids = 0.5 * (m.nodes[e[:, 0]] + m.nodes[e[:, 1]])
    outside = m.nodes.max(axis=0) + rng.uniform(0.2, 1.0, (10, m.N))
    locs = np.vstack([inside, verts, mids, outside])
    psi, D, found = c.eval_pointwise(locs)
    ref = oracle.pointwise_psi(m, order, od, ond, locs)
    assert np.all(found[: n_in + 60] >= 0) and np.all(found[n_in + 60:] == -1)
    assert np.abs(psi.toarray() - ref).max() < 1e-12
    # areal evaluation: random subdomains (rows of an incidence matrix over the cells)
    inc = (rng.uniform(0, 1, (5, m.n_cells)) < 0.3).astype(float)
    inc[0, :] = 1.0
    psi_a, D_a = c.eval_areal(inc)
    ref_a, Dref = oracle.areal_psi(m, order, od, ond, inc)
    assert np.abs(psi_a.toarray() - ref_a).max() < 1e-12 * max(1.0, np.abs(ref_a).max())
    assert np.abs(D_a - Dref).max() < 1e-13 * max(1.0, np.abs(Dref).max())
    c.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FDAPDE_FUZZ_SEEDS_BND", "12"))))
def test_boundary_dofs_of_a_partial_node_mask(capi, oracle, seed):
    """Dirichlet markers on a random PART of the boundary nodes (the reference takes the node markers as given, boundary.csv): P1 boundary DOFs = the node
    markers; P2 edge DOFs: 2-D -- every edge seen by exactly one cell, whatever its end nodes carry (triangulation.h:177, 187 via lagrangian_basis.h:125);
    3-D -- both end nodes marked (triangulation.h:371).  DOF table, boundary set and coordinates bit-exact against the oracle."""
    from fdapde_core_amd import meshgen

    rng = np.random.default_rng(12000 + seed)
    dim = int(rng.integers(2, 4))
    order = int(rng.integers(1, 3))
    nx = int(rng.integers(2, 10)) if dim == 2 else int(rng.integers(2, 5))
    nodes, cells, bnd = meshgen.unit_square(nx, seed=seed + 11) if dim == 2 else meshgen.unit_cube(nx, seed=seed + 11)
    mask = ((rng.uniform(0, 1, bnd.shape[0]) < rng.uniform(0.1, 0.9)) & (bnd != 0)).astype(np.uint8)
    m = oracle.Mesh(np.ascontiguousarray(nodes), np.ascontiguousarray(cells), np.ascontiguousarray(mask))
    c = capi.Context(0)
    c.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = c.dofs_build(order)
    od, ob, ond, _ = oracle.enumerate_dofs(m, order)
    dofs, b, coords = c.dofs_get()
    assert nd == ond and np.array_equal(dofs, od)
    assert np.array_equal(b, ob), (dim, order, int((b != ob).sum()))
    assert np.array_equal(coords, oracle.dofs_coords(m, order, od, ond))
    c.close()
