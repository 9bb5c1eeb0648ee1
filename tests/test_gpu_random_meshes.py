"""Irregular (Delaunay) meshes with seeded random points: vertex valences, row lengths and cell shapes the structured fixtures
do not have (long P1 rows, P2 rows of very different lengths -> segmented solver pattern, slivers in 3-D).  Device path against
the oracle: numbering bit-exact, matrix entries to 1e-12, solutions to 1e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from fdapde_loader import load_package

    m = load_package().capi
    assert m.load().fdapde_device_count() >= 1
    return m


def _delaunay(dim, n, seed):
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    pts = rng.uniform(0.0, 1.0, (n, dim))
    corners = np.array(np.meshgrid(*[[0.0, 1.0]] * dim)).reshape(dim, -1).T
    pts = np.vstack([corners, pts])
    tri = Delaunay(pts)
    cells = tri.simplices.astype(np.int32)
    # drop degenerate simplices (Delaunay of random points can return near-flat ones on the hull)
    J = pts[cells[:, 1:]] - pts[cells[:, :1]]
    vol = np.abs(np.linalg.det(J))
    cells = cells[vol > 1e-9]
    bnd = np.zeros(pts.shape[0], dtype=np.uint8)
    bnd[np.unique(tri.convex_hull)] = 1
    used = np.zeros(pts.shape[0], dtype=bool)
    used[np.unique(cells)] = True
    assert used.all()
    return pts, np.ascontiguousarray(cells), bnd


@pytest.mark.parametrize("dim,n,order,seed", [(2, 600, 1, 1), (2, 600, 2, 2), (3, 350, 1, 3), (3, 350, 2, 4), (2, 3000, 2, 5)])
def test_delaunay_mesh_parity(capi, oracle, dim, n, order, seed):
    nodes, cells, bnd = _delaunay(dim, n, seed)
    m = oracle.Mesh(nodes, cells, bnd)
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, ob, ond, _ = oracle.enumerate_dofs(m, order)
    dofs, b, coords = ctx.dofs_get()
    assert nd == ond and np.array_equal(dofs, od) and np.array_equal(b, ob)
    bvec = np.array([0.8, -0.4, 0.3])[:dim]
    ops = {"spd": lambda mod: -mod.laplacian() + mod.reaction(1.0), "adr": lambda mod: -mod.laplacian() + mod.advection(bvec) + mod.reaction(0.5)}
    qn = ctx.quadrature_nodes()
    fq = np.sin(3.0 * qn[:, 0]) + qn[:, -1]
    g = coords[:, 0] - 0.5 * coords[:, -1]
    for name, op in ops.items():
        ctx.set_operator(op(capi))
        ctx.set_forcing(fq)
        ctx.set_dirichlet(g)
        ctx.init()
        ref = oracle.pde_init_solve(m, order, op(oracle), forcing_q=fq, dirichlet=g)
        info = ctx.solve(rtol=1e-12, maxit=20000)
        assert info.converged == 1, (name, info.iters, info.relres)
        u = ctx.solution()
        assert np.linalg.norm(u - ref.solution) / np.linalg.norm(ref.solution) <= 1e-8, name
        # entries: assemble alone (no Dirichlet rows) against the oracle's assembly
        ctx.assemble_operator(capi.MAT_STIFF, op(capi))
        got = ctx.matrix_values(capi.MAT_STIFF)
        exp = oracle.assemble_operator(m, order, od, nd, op(oracle)).values
        assert np.abs(got - exp).max() <= 1e-12 * max(1.0, np.abs(exp).max()), name
    ctx.close()


@pytest.mark.parametrize("dim,n,order,seed", [(2, 6000, 1, 11), (2, 4000, 2, 12), (3, 2500, 1, 13), (3, 1200, 2, 14)])
def test_delaunay_mesh_symmetric_storage(capi, dim, n, order, seed):
    """the persistent CG's symmetric storage forced on irregular meshes (rows of very different lengths, rows whose pairs all lie in
    other workgroups or all in their own, several workgroups): the solution and the iteration count of the plain storage, and of the
    multi-launch path, and identical bits on a second launch"""
    nodes, cells, bnd = _delaunay(dim, n, seed)
    ctx = capi.Context(device=0)
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(order)
    _, _, coords = ctx.dofs_get()
    qn = ctx.quadrature_nodes()
    ctx.set_operator(-capi.laplacian() + capi.reaction(0.3))
    ctx.set_forcing(np.cos(2.0 * qn[:, 0]) - qn[:, -1])
    ctx.set_dirichlet(0.5 * coords[:, 0] + coords[:, -1] ** 2)
    ctx.init()
    res = {}
    for name, knobs in (("multi", {"persist": 0}), ("plain", {"persist": 1, "persist_sym": 0}), ("sym", {"persist": 1, "persist_sym": 1})):
        for k, v in knobs.items():
            ctx.tune(k, v)
        i = ctx.solve(rtol=1e-11, maxit=20000)
        assert i.converged == 1 and i.persistent == (0 if name == "multi" else 1), name
        res[name] = (ctx.solution().copy(), i.iters)
    i2 = ctx.solve(rtol=1e-11, maxit=20000)
    assert i2.iters == res["sym"][1] and np.array_equal(ctx.solution(), res["sym"][0])
    for name in ("plain", "sym"):
        assert abs(res[name][1] - res["multi"][1]) <= max(2, res["multi"][1] // 100), (name, res[name][1], res["multi"][1])
        assert np.linalg.norm(res[name][0] - res["multi"][0]) <= 1e-9 * np.linalg.norm(res["multi"][0]), name
    ctx.close()


@pytest.mark.parametrize("n_rim,order", [(40, 1), (700, 1), (700, 2)])
def test_fan_mesh_with_one_very_long_row(capi, oracle, n_rim, order):
    """A fan of n_rim triangles around one vertex: that vertex's row has n_rim + 1 (P1) or 2 n_rim + 1 (P2) entries, far above a team
    pass; with n_rim = 700 it also exceeds what the segmented solver pattern accepts (32 chunks per row), so the plain pattern with
    its tail passes must take over.  Two rings, so that the centre and the first ring are interior DOFs."""
    th = np.linspace(0.0, 2.0 * np.pi, n_rim, endpoint=False)
    ring1 = 0.5 * np.stack([np.cos(th), np.sin(th)], axis=1)
    ring2 = 1.0 * np.stack([np.cos(th + np.pi / n_rim), np.sin(th + np.pi / n_rim)], axis=1)
    nodes = np.vstack([[0.0, 0.0], ring1, ring2])
    c, r1, r2 = 0, 1 + np.arange(n_rim), 1 + n_rim + np.arange(n_rim)
    nxt = np.roll(np.arange(n_rim), -1)
    cells = np.vstack([np.stack([np.full(n_rim, c), r1, r1[nxt]], axis=1),
                       np.stack([r1, r2, r1[nxt]], axis=1),
                       np.stack([r1[nxt], r2, r2[nxt]], axis=1)]).astype(np.int32)
    bnd = np.zeros(nodes.shape[0], dtype=np.uint8)
    bnd[r2] = 1
    m = oracle.Mesh(nodes, cells, bnd)
    ctx = capi.Context(device=0)
    ctx.mesh_upload(m.nodes, m.cells, m.boundary)
    nd = ctx.dofs_build(order)
    od, ob, ond, _ = oracle.enumerate_dofs(m, order)
    dofs, b, coords = ctx.dofs_get()
    assert nd == ond and np.array_equal(dofs, od) and np.array_equal(b, ob)
    rp, _ = ctx.pattern_get()
    assert np.diff(rp).max() >= n_rim + 1
    op = lambda mod: -mod.laplacian() + mod.reaction(0.3)
    qn = ctx.quadrature_nodes()
    fq = 1.0 + qn[:, 0]
    g = coords[:, 1]
    ctx.set_operator(op(capi))
    ctx.set_forcing(fq)
    ctx.set_dirichlet(g)
    ctx.init()
    ref = oracle.pde_init_solve(m, order, op(oracle), forcing_q=fq, dirichlet=g)
    info = ctx.solve(rtol=1e-12)
    assert info.converged == 1
    assert np.linalg.norm(ctx.solution() - ref.solution) / np.linalg.norm(ref.solution) <= 1e-8
    x = np.sin(np.arange(nd, dtype=float))
    ctx.assemble_operator(capi.MAT_STIFF, op(capi))
    A = oracle.assemble_operator(m, order, od, nd, op(oracle))
    assert np.abs(ctx.spmv(capi.MAT_STIFF, x) - A.matvec(x)).max() <= 1e-11 * np.abs(A.matvec(x)).max()
    ctx.close()
