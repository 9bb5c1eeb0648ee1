"""BASELINE.json full-size workloads (C2: ~1M triangles, C3: ~10M tetrahedra; plus a P2 3-D case) on the GPU, checked through
size-independent properties -- the oracle does not finish these sizes in seconds:
  * A 1 = 0 for the Laplacian (constants are in the kernel of the stiffness), A = A^T bitwise
  * 1^T M 1 = |Omega| = 1 and sum(force) = |Omega| for f = 1 (partition of unity)
  * assembling twice gives identical bits (no atomics in the default path); atomic and coloured scatter agree to rounding
  * the discrete solution of -Lap u = f, u = sin(pi x)... converges to the analytic one at the O(h^2) level expected of P1
    and satisfies ||A u - b|| <= 1e-8 ||b|| on interior rows (checked with an independent SpMV in scipy on the exported CSR)
"""
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


def _csr(ctx, capi, which):
    import scipy.sparse as sp

    rp, ci = ctx.pattern_get()
    n = rp.size - 1
    return sp.csr_matrix((ctx.matrix_values(which), ci, rp), shape=(n, n))


def test_above_two_million_rows(env):
    """Above 2 M rows the SpMV is launched with the nontemporal hint on its value stream (another instantiation) and the column-code
    windows see larger index distances: 132^3 x 6 tetrahedra, 2.35 M DOFs, checked by the true residual on the exported matrix"""
    import scipy.sparse as sp

    capi, meshgen = env
    nodes, cells, bnd = meshgen.unit_cube(132)
    u_exact, f = meshgen.manufactured(3)
    ctx = capi.Context(device=0)
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(1)
    assert nd > 2_000_000
    _, bdofs, coords = ctx.dofs_get()
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    g = 0.1 * coords[:, 0]                                   # non-homogeneous: the lift A g~ is exercised too
    ctx.set_dirichlet(g)
    ctx.init()
    info = ctx.solve(rtol=1e-10)
    assert info.converged == 1
    # round 4: above 8 192 rows per workgroup the solve still runs as ONE launch -- the wide form of the plain storage (24 rows per thread, x in
    # HBM) -- instead of falling to the multi-launch path
    assert info.persistent == 1 and ctx.solver_layout_kind(True)["rows_per_thread"] == 24
    u = ctx.solution()
    rp, ci = ctx.pattern_get()
    Az = sp.csr_matrix((ctx.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))
    b = ctx.force()
    assert np.linalg.norm(Az @ u - b) <= 1e-8 * np.linalg.norm(b)
    assert np.array_equal(u[bdofs.astype(bool)], g[bdofs.astype(bool)])
    ctx.tune("persist_wide", 0)   # the multi-launch path the system took before: the same answer
    info0 = ctx.solve(rtol=1e-10)
    assert info0.persistent == 0 and abs(info0.iters - info.iters) <= max(2, info.iters // 100)
    assert np.linalg.norm(ctx.solution() - u) <= 1e-9 * np.linalg.norm(u)
    ctx.close()


def test_c5_fullsize(env):
    """BASELINE config C5 at full size on one GPU: 3-D P2 advection-diffusion-reaction, 87^3 x 6 = 3 951 018 tetrahedra,
    5 359 375 DOFs, Jacobi-BiCGStab.  3-D P2 numbering is build-defined (the reference does not compile it): what is checked here
    is numbering-independent -- partition of unity, the true residual on the exported matrix, second-order accuracy against the
    manufactured solution (the 5-point rule has degree of precision 3, integrator_tables.h:274, so O(h^2) is the reference's own
    order for this load vector), bitwise-repeatable assembly."""
    import scipy.sparse as sp

    capi, meshgen = env
    from fdapde_core_amd import workloads

    nx = 87
    nodes, cells, bnd = meshgen.unit_cube(nx)
    assert cells.shape[0] == 3951018
    ctx = capi.Context(device=0)
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(2)
    del nodes, cells
    assert nd == 88**3 + ctx.sizes()["n_edges"] and nd > 5_300_000
    _, bdofs, coords = ctx.dofs_get()
    qn = ctx.quadrature_nodes()
    ctx.set_operator(workloads.c5_operator(capi))
    ctx.set_forcing(np.ones(qn.shape[0]))
    ctx.init()
    rp, ci = ctx.pattern_get()
    mvals = ctx.matrix_values(capi.MAT_MASS)
    assert abs(mvals.sum() - 1.0) < 1e-10                                 # 1^T M 1 = |Omega|
    assert abs(ctx.force().sum() - 1.0) < 1e-10
    a0 = ctx.matrix_values(capi.MAT_STIFF)
    # row sums of the operator: Laplacian and advection annihilate constants, the reaction term leaves c * M 1
    A = sp.csr_matrix((a0, ci, rp), shape=(nd, nd))
    M = sp.csr_matrix((mvals, ci, rp), shape=(nd, nd))
    ones = np.ones(nd)
    assert np.abs(A @ ones - workloads.C5_C * (M @ ones)).max() <= 1e-11 * np.abs(a0).max()
    del M, mvals
    ctx.init()
    assert np.array_equal(ctx.matrix_values(capi.MAT_STIFF), a0)         # bitwise-repeatable assembly
    del A, a0
    ctx.set_forcing(workloads.c5_forcing(qn))
    del qn
    ctx.set_dirichlet(np.zeros(nd))
    ctx.init()
    # the open method at this size: the two-level solver (eng_pmg.hip) -- two dozen iterations, the TRUE residual in info.relres
    info2 = ctx.solve(rtol=1e-10)
    assert info2.converged == 1 and info2.method_used == capi.SOLVER_PMG and info2.relres <= 1e-9 and info2.iters <= 40, (info2.method_used, info2.iters)
    u2 = ctx.solution()
    assert np.abs(u2 - workloads.c5_exact(coords)).max() < 1.0 * (1.0 / nx) ** 2 * np.pi**2
    assert np.all(u2[bdofs.astype(bool)] == 0.0)
    # ... and the Jacobi-preconditioned stage it replaces
    ctx.tune("pmg_auto", 0)
    info = ctx.solve(rtol=1e-10)
    assert info.converged == 1 and info.method_used == capi.SOLVER_BICGSTAB and info.relres <= 1e-10
    assert np.abs(ctx.solution() - u2).max() <= 1e-8
    # The count of this chaotic recurrence is one draw from a distribution (the same problem with its right-hand side scaled by 1 + k 2^-48 takes 659 - 785
    # iterations: tools/c5_iter_spread.py, profiles/r6_c5_iter_spread.txt): the bound catches a degraded iteration, not a rounding change.
    assert info.iters <= 900, info.iters
    u = ctx.solution()
    err = np.abs(u - workloads.c5_exact(coords)).max()
    assert err < 1.0 * (1.0 / nx) ** 2 * np.pi**2                         # O(h^2): 1.3e-3 bound, 3.3e-4 observed
    Az = sp.csr_matrix((ctx.matrix_values(capi.MAT_STIFF), ci, rp), shape=(nd, nd))   # row-zeroed, unit diagonal on the boundary
    b = ctx.force()
    assert np.linalg.norm(Az @ u - b) <= 1e-8 * np.linalg.norm(b)
    assert np.all(u[bdofs.astype(bool)] == 0.0)
    ctx.close()


@pytest.mark.parametrize("config", ["C2", "C3", "P2_3D"])
def test_fullsize_properties(env, config):
    capi, meshgen = env
    if config == "C2":
        nodes, cells, bnd = meshgen.unit_square(708)      # 1 002 528 triangles, 502 681 nodes
        order, expect_cells, h = 1, 1002528, 1.0 / 708
    elif config == "C3":
        nodes, cells, bnd = meshgen.unit_cube(119)        # 10 110 954 tetrahedra, 1 728 000 nodes
        order, expect_cells, h = 1, 10110954, 1.0 / 119
    else:
        nodes, cells, bnd = meshgen.unit_cube(36)         # 279 936 tetrahedra, P2: ~390k DOFs, ~28 nnz / row
        order, expect_cells, h = 2, 279936, 1.0 / 36
    assert cells.shape[0] == expect_cells
    N = nodes.shape[1]
    u_exact, f = meshgen.manufactured(N)
    ctx = capi.Context(device=0)
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(order)
    dofs, bdofs, coords = ctx.dofs_get()
    if order == 1:
        assert nd == nodes.shape[0] and np.array_equal(dofs, cells) and np.array_equal(bdofs, bnd)
    qn = ctx.quadrature_nodes()
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(np.ones(qn.shape[0]))
    ctx.init()
    A, M = _csr(ctx, capi, capi.MAT_STIFF), _csr(ctx, capi, capi.MAT_MASS)
    scale = abs(A).max()
    assert np.abs(A @ np.ones(nd)).max() <= 1e-11 * scale               # constants in the kernel
    assert abs(A - A.T).max() == 0.0 and abs(M - M.T).max() == 0.0       # bitwise symmetric
    assert abs(M.sum() - 1.0) < 1e-10                                    # |Omega| = 1
    assert abs(ctx.force().sum() - 1.0) < 1e-10
    a0 = A.data.copy()
    ctx.assemble_operator(capi.MAT_STIFF, -capi.laplacian(), capi.ASSEMBLY_ROWS)
    assert np.array_equal(ctx.matrix_values(capi.MAT_STIFF), a0)        # reproducible bits
    for variant in (capi.ASSEMBLY_ATOMIC, capi.ASSEMBLY_COLOURED, capi.ASSEMBLY_PARTITIONED) + ((capi.ASSEMBLY_WAVE,) if order == 1 else ()):
        ctx.assemble_operator(capi.MAT_STIFF, -capi.laplacian(), variant)
        assert np.abs(ctx.matrix_values(capi.MAT_STIFF) - a0).max() <= 1e-12 * scale
    # full solve against the analytic solution
    ctx.set_forcing(f(qn))
    ctx.set_dirichlet(np.zeros(nd))
    ctx.init()
    info = ctx.solve(rtol=1e-10)
    assert info.converged == 1 and info.relres <= 1e-10
    u = ctx.solution()
    err = np.abs(u - u_exact(coords)).max()
    assert err < (6.0 if order == 1 else 1.0) * h * h * np.pi**2         # O(h^2) with a generous constant
    b = ctx.force()                                                       # boundary rows carry g = 0 after the solve
    Az = _csr(ctx, capi, capi.MAT_STIFF)                                  # row-zeroed, unit diagonal on the boundary
    res = Az @ u - b
    assert np.linalg.norm(res) <= 1e-8 * np.linalg.norm(b)
    assert np.all(u[bdofs.astype(bool)] == 0.0)
    # SpMV of the device agrees with an independent CSR product on the exported matrix
    x = np.random.default_rng(1).standard_normal(nd)
    y = ctx.spmv(capi.MAT_MASS, x)
    yr = M @ x
    assert np.abs(y - yr).max() <= 1e-12 * max(1.0, np.abs(yr).max())
    ctx.close()


@pytest.mark.parametrize("config", ["C3", "C2"])
def test_entries_against_the_oracle_at_full_size(env, config):
    """the headline matrices entry by entry.  The oracle (oracle/fem_oracle.c: fem_assembler.h:52-121 restated, triplets summed in the
    reference's order) assembles the very meshes the numbers are quoted on -- 10 110 954 tetrahedra in ~17 s, 1 002 528 triangles in ~1 s -- and every one
    of the 25.6 M (3.5 M) entries of stiff_ and mass_ and every entry of force_ is compared at the bar of SURVEY 8(d): |d| <= 1e-12 max(1, |A|max);
    DOF table, boundary DOFs and CSR pattern bit-exact."""
    from oracle import oracle as o

    capi, meshgen = env
    if config == "C3":
        nodes, cells, bnd = meshgen.unit_cube(119)
    else:
        nodes, cells, bnd = meshgen.unit_square(708)
    _, f = meshgen.manufactured(nodes.shape[1])
    ctx = capi.Context(device=0)
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(1)
    qn = ctx.quadrature_nodes()
    fq = f(qn)
    ctx.set_operator(-capi.laplacian())
    ctx.set_forcing(fq)
    ctx.init()
    m = o.Mesh(nodes, cells, bnd)
    od, ob, ond, _ = o.enumerate_dofs(m, 1)
    dofs, bdofs, _ = ctx.dofs_get()
    assert nd == ond and np.array_equal(dofs, od) and np.array_equal(bdofs != 0, ob != 0)
    assert np.abs(qn - o.quadrature_nodes(m, 1)).max() <= 1e-15
    rp, ci = ctx.pattern_get()
    A = o.assemble_operator(m, 1, od, ond, -o.laplacian())
    assert np.array_equal(rp, A.rowptr) and np.array_equal(ci, A.colidx)
    assert np.abs(ctx.matrix_values(capi.MAT_STIFF) - A.values).max() <= 1e-12 * max(1.0, np.abs(A.values).max())
    del A
    Mm = o.assemble_operator(m, 1, od, ond, o.reaction(1.0))
    assert np.abs(ctx.matrix_values(capi.MAT_MASS) - Mm.values).max() <= 1e-12 * max(1.0, np.abs(Mm.values).max())
    del Mm
    rhs = o.assemble_forcing(m, 1, od, ond, fq)
    assert np.abs(ctx.force() - rhs).max() <= 1e-12 * max(1.0, np.abs(rhs).max())
    ctx.close()
