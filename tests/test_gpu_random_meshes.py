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
