"""The device-side partitioner (fdapde_partition_build, csrc/dev_partition.hip) against its numpy original (fdapde-core_amd/dist.py: partition_cells,
node_owners, rowdist_sub_mesh / sub_mesh, peer_lists) -- array for array, bit for bit -- and the multi-device context (fdapde_ctx_create_multi,
csrc/eng_group.hip) against the single-device context and the CPU oracle.  "Devices" are all GPU 0 here (a device named several times shares
out its CUs); on a node with more GPUs the same tests run across them (test_group_across_real_devices)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi, meshgen
    from fdapde_core_amd import dist as fdist

    assert capi.load().fdapde_device_count() >= 1
    return capi, meshgen, fdist


def _mesh(meshgen, kind, nx):
    return meshgen.unit_square(nx) if kind == "square" else meshgen.unit_cube(nx)


@pytest.mark.parametrize("kind,nx,world", [("cube", 7, 2), ("cube", 7, 3), ("cube", 12, 5), ("square", 24, 2), ("square", 31, 4), ("cube", 20, 8)])
def test_row_distributed_partition_is_dist_py_bit_for_bit(env, kind, nx, world):
    capi, meshgen, fdist = env
    nodes, cells, bnd = _mesh(meshgen, kind, nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    c.partition_build(world, capi.PARTITION_ROWDIST)
    part, owner, mask = c.partition_whole()
    ref_part = fdist.partition_cells(nodes, cells, world)
    assert np.array_equal(part, ref_part)
    ref_owner = fdist.node_owners(cells, ref_part, nodes.shape[0], nodes)
    assert np.array_equal(owner, ref_owner)
    probs = fdist.rank_problems_rowdist_p1(nodes, cells, bnd, world)
    for r in range(world):
        got, ref = c.partition_get(r), probs[r]
        sub = fdist.rowdist_sub_mesh(nodes, cells, bnd, ref_owner, r)
        assert np.array_equal(got["l2g"], ref["l2g"]) and np.array_equal(got["cell_ids"], sub["cell_ids"])
        assert np.array_equal(got["cells"], ref["cells"]) and np.array_equal(got["nodes"], ref["nodes"])
        assert np.array_equal(got["boundary"], ref["boundary"]) and np.array_equal(got["owner"], ref["owner"])
        assert np.array_equal((mask >> np.uint64(r)) & np.uint64(1), np.isin(np.arange(nodes.shape[0]), ref["l2g"]).astype(np.uint64))
    c.close()


@pytest.mark.parametrize("kind,nx,world", [("cube", 7, 2), ("cube", 9, 3), ("square", 24, 5), ("cube", 16, 8)])
def test_element_partition_and_peer_lists_are_dist_py_bit_for_bit(env, kind, nx, world):
    capi, meshgen, fdist = env
    nodes, cells, bnd = _mesh(meshgen, kind, nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    c.partition_build(world, capi.PARTITION_ELEMENTS)
    probs = fdist.rank_problems_p1(nodes, cells, bnd, world)
    part, owner, _ = c.partition_whole()
    assert np.array_equal(part, fdist.partition_cells(nodes, cells, world))
    for r in range(world):
        got, ref = c.partition_get(r), probs[r]
        assert np.array_equal(got["l2g"], ref["l2g"]) and np.array_equal(got["cells"], ref["cells"]) and np.array_equal(got["nodes"], ref["nodes"])
        assert np.array_equal(got["boundary"], ref["boundary"])
        pr, po, pd, owned = c.partition_peers(r)
        assert np.array_equal(pr, ref["peer_rank"]) and np.array_equal(po, ref["peer_off"]) and np.array_equal(pd, ref["peer_dof"])
        assert np.array_equal(owned, ref["owned"])
        assert np.array_equal(got["owner"] == r, ref["owned"] != 0)
    c.close()


def test_partition_needs_a_mesh_and_sane_arguments(env):
    capi, meshgen, _ = env
    c = capi.Context(0)
    with pytest.raises(capi.FdapdeError) as e:
        c.partition_build(2)
    assert e.value.status == capi.ENOTINIT
    c.mesh_upload(*meshgen.unit_cube(4))
    for world in (0, 65):
        with pytest.raises(capi.FdapdeError) as e:
            c.partition_build(world)
        assert e.value.status == capi.EINVAL
    with pytest.raises(capi.FdapdeError):
        c.partition_get(0)   # (a failed build leaves no partition behind)
    c.partition_build(1)
    one = c.partition_get(0)
    assert one["cells"].shape[0] == 6 * 4**3 and np.array_equal(one["l2g"], np.arange(5**3))
    c.close()


# ---- the multi-device context ------------------------------------------------------------------------------------------------------------------
def _problem(capi, meshgen, kind, nx, order, op_name):
    nodes, cells, bnd = _mesh(meshgen, kind, nx)
    N = nodes.shape[1]
    _, f = meshgen.manufactured(N)
    bvec = [1.0, 0.5, 0.25][:N]
    op = {"lap": lambda: -capi.laplacian(), "adr": lambda: -capi.laplacian() + capi.advection(bvec) + capi.reaction(1.0),
          "lap+r": lambda: -capi.laplacian() + capi.reaction(0.5)}[op_name]
    return nodes, cells, bnd, f, op


def _solve(capi, ctx, nodes, cells, bnd, order, f, op, dirichlet=True):
    ctx.mesh_upload(nodes, cells, bnd)
    nd = ctx.dofs_build(order)
    dofs, bdofs, coords = ctx.dofs_get()
    ctx.set_operator(op())
    ctx.set_forcing(f(ctx.quadrature_nodes()))
    ctx.set_dirichlet(0.3 * coords[:, 0] - 0.2 * coords[:, -1] if dirichlet else None)
    ctx.init()
    raw = (ctx.matrix_values(capi.MAT_STIFF), ctx.matrix_values(capi.MAT_MASS), ctx.force())
    info = ctx.solve(rtol=1e-11)
    return nd, info, ctx.solution(), raw, (dofs, bdofs, coords)


CASES = [("cube", 10, 1, "lap", 2), ("cube", 10, 1, "lap", 4), ("cube", 8, 2, "lap", 2), ("cube", 8, 2, "adr", 3), ("square", 24, 2, "lap", 2),
         ("square", 30, 1, "adr", 4), ("cube", 9, 1, "lap+r", 3)]


@pytest.mark.parametrize("kind,nx,order,op_name,n_dev", CASES)
def test_multi_device_context_equals_the_single_device_one(env, kind, nx, order, op_name, n_dev):
    """sharded_laplacian_order1 / _order2 / _adr over 2 - 4 "devices" (all GPU 0): every getter in the reference numbering of the WHOLE mesh, the
    solution equal to the one-device solution <= 1e-9 and to the oracle's direct solve <= 1e-8, the assembled matrices to the bits' last places"""
    from oracle import oracle as o

    capi, meshgen, _ = env
    nodes, cells, bnd, f, op = _problem(capi, meshgen, kind, nx, order, op_name)
    one = capi.Context(0)
    nd1, info1, u1, raw1, sp1 = _solve(capi, one, nodes, cells, bnd, order, f, op)
    grp = capi.Context(devices=[0] * n_dev)
    ndg, infog, ug, rawg, spg = _solve(capi, grp, nodes, cells, bnd, order, f, op)
    d = grp.devices()
    assert d["devices"] == [0] * n_dev and d["form"] in (0, 1)
    assert ndg == nd1 and infog.converged == 1
    for a, b in zip(sp1, spg):
        assert np.array_equal(a, b)
    assert np.array_equal(one.pattern_get()[0], grp.pattern_get()[0]) and np.array_equal(one.pattern_get()[1], grp.pattern_get()[1])
    for a, b in zip(raw1, rawg):
        assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(a).max())
    assert np.linalg.norm(ug - u1) <= 1e-9 * np.linalg.norm(u1)
    # after the Dirichlet solve: the row-zeroed matrix and the force with g on the boundary rows, as on one device
    assert np.abs(grp.matrix_values(capi.MAT_STIFF) - one.matrix_values(capi.MAT_STIFF)).max() <= 1e-13 * np.abs(raw1[0]).max()
    assert np.abs(grp.force() - one.force()).max() <= 1e-13 * max(1.0, np.abs(one.force()).max())
    assert np.abs(grp.lump(capi.MAT_MASS) - one.lump(capi.MAT_MASS)).max() <= 1e-14
    x = np.random.default_rng(3).standard_normal(nd1)
    assert np.abs(grp.spmv(capi.MAT_MASS, x) - one.spmv(capi.MAT_MASS, x)).max() <= 1e-13
    # what the root context serves by itself: quadrature nodes, point location + basis values, cell integrals, topology
    assert np.array_equal(grp.quadrature_nodes(), one.quadrature_nodes())
    locs = np.random.default_rng(4).uniform(0.05, 0.95, (40, nodes.shape[1]))
    (Pg, Dg, cg), (P1, D1, c1) = grp.eval_pointwise(locs), one.eval_pointwise(locs)
    assert abs(Pg - P1).max() == 0.0 and np.array_equal(Dg, D1) and np.array_equal(cg, c1)
    tg, t1 = grp.topology(), one.topology()
    for k in t1:
        assert np.array_equal(tg[k], t1[k]), k
    if nx <= 10 or kind == "square":   # the oracle's direct solve of the whole mesh
        m = o.Mesh(np.ascontiguousarray(nodes), np.ascontiguousarray(cells, dtype=np.int32), np.ascontiguousarray(bnd, dtype=np.uint8))
        oop = {"lap": lambda: -o.laplacian(), "adr": lambda: -o.laplacian() + o.advection(np.array([1.0, 0.5, 0.25][:nodes.shape[1]])) + o.reaction(1.0),
               "lap+r": lambda: -o.laplacian() + o.reaction(0.5)}[op_name]()
        coords = sp1[2]
        ref = o.pde_init_solve(m, order, oop, forcing_q=f(o.quadrature_nodes(m, order)), dirichlet=0.3 * coords[:, 0] - 0.2 * coords[:, -1]).solution
        if order == 1 or nodes.shape[1] == 2:   # (3-D P2 numbering is build-defined: the oracle's may differ)
            assert np.linalg.norm(ug - ref) <= 1e-8 * np.linalg.norm(ref)
    one.close(), grp.close()


@pytest.mark.parametrize("form", [0, 1])
def test_multi_device_context_in_both_forms(env, form):
    """the form pinned (knob group_form): row-distributed persistent launches / element partition with the neighbour exchange -- same answers"""
    capi, meshgen, _ = env
    nodes, cells, bnd, f, op = _problem(capi, meshgen, "cube", 10, 1, "lap")
    one = capi.Context(0)
    _, _, u1, raw1, _ = _solve(capi, one, nodes, cells, bnd, 1, f, op)
    grp = capi.Context(devices=[0, 0, 0])
    grp.mesh_upload(nodes, cells, bnd)
    nd = grp.dofs_build(1)
    grp.tune("group_form", form)
    assert grp.devices()["form"] == form
    _, _, coords = grp.dofs_get()
    grp.set_operator(op())
    grp.set_forcing(f(grp.quadrature_nodes()))
    grp.set_dirichlet(0.3 * coords[:, 0] - 0.2 * coords[:, -1])
    grp.init()
    assert np.abs(grp.matrix_values(capi.MAT_STIFF) - raw1[0]).max() <= 1e-13 * np.abs(raw1[0]).max()
    info = grp.solve(rtol=1e-11)
    assert info.converged == 1 and info.persistent == (1 if form == 0 else 0)
    assert np.linalg.norm(grp.solution() - u1) <= 1e-9 * np.linalg.norm(u1)
    assert np.abs(grp.matrix_values(capi.MAT_STIFF) - one.matrix_values(capi.MAT_STIFF)).max() <= 1e-13 * np.abs(raw1[0]).max()
    assert np.abs(grp.force() - one.force()).max() <= 1e-13 * max(1.0, np.abs(one.force()).max())
    one.close(), grp.close()


def test_multi_device_context_changes_form_when_the_row_distributed_solve_declines(env, monkeypatch):
    """FDAPDE_ROWDIST_REFUSE: every rank's share "does not fit" the single launch -> FDAPDE_EUNSUPPORTED on all ranks -> the context re-partitions
    in the element form, deals the problem data again, assembles again and answers"""
    capi, meshgen, _ = env
    nodes, cells, bnd, f, op = _problem(capi, meshgen, "cube", 8, 1, "lap")
    one = capi.Context(0)
    _, _, u1, _, _ = _solve(capi, one, nodes, cells, bnd, 1, f, op)
    monkeypatch.setenv("FDAPDE_ROWDIST_REFUSE", "1")
    grp = capi.Context(devices=[0, 0])
    _, info, ug, _, _ = _solve(capi, grp, nodes, cells, bnd, 1, f, op)
    assert grp.devices()["form"] == 1 and info.converged == 1 and info.persistent == 0
    assert np.linalg.norm(ug - u1) <= 1e-9 * np.linalg.norm(u1)
    one.close(), grp.close()


def test_multi_device_parabolic_and_handle(env):
    """fdapde_solve_parabolic and the factor-once handle through the multi-device context against the single-device one"""
    capi, meshgen, _ = env
    nodes, cells, bnd = meshgen.unit_square(20)
    times = np.linspace(0.0, 0.1, 6)
    out = []
    for devices in (None, [0, 0]):
        c = capi.Context(0) if devices is None else capi.Context(devices=devices)
        c.mesh_upload(nodes, cells, bnd)
        nd = c.dofs_build(1)
        _, _, coords = c.dofs_get()
        qn = c.quadrature_nodes()
        c.set_operator(capi.dt() - capi.laplacian())
        F = np.stack([np.sin(np.pi * qn[:, 0]) * (1.0 + t) for t in times], axis=1)
        c.set_forcing(F)
        c.init()
        u0 = np.sin(np.pi * coords[:, 0]) * np.sin(np.pi * coords[:, 1])
        G = np.zeros((nd, times.size))
        sol, info = c.solve_parabolic(times, u0, G, rtol=1e-11)
        assert info.converged == 1
        vals = c.matrix_values(capi.MAT_STIFF) + 30.0 * c.matrix_values(capi.MAT_MASS)
        c.lin_compute(values=vals, symmetric=True)
        B = np.random.default_rng(1).standard_normal((nd, 3))
        X, linfo = c.lin_solve(B, rtol=1e-11)
        assert linfo.converged == 1
        out.append((sol, X))
        c.close()
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-9 * np.abs(out[0][0]).max()
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-8 * np.abs(out[0][1]).max()


def test_multi_device_refusals(env):
    capi, meshgen, _ = env
    grp = capi.Context(devices=[0, 0])
    grp.mesh_upload(*meshgen.unit_cube(4))
    with pytest.raises(capi.FdapdeError) as e:
        grp.set_operator(-capi.laplacian())
    assert e.value.status == capi.ENOTINIT
    grp.dofs_build(1)
    with pytest.raises(capi.FdapdeError) as e:
        grp.solve()
    assert e.value.status == capi.ENOTINIT
    with pytest.raises(capi.FdapdeError) as e:
        grp.comm_allreduce([1.0])
    assert e.value.status == capi.EUNSUPPORTED
    grp.close()


def test_multi_device_context_clones(env):
    """fdapde_ctx_clone of a multi-device context (what the copy-on-write owner of the host-side bindings calls): every getter returns the source's
    bits, it solves without another init, and diverges without the source noticing"""
    capi, meshgen, _ = env
    nodes, cells, bnd, f, op = _problem(capi, meshgen, "cube", 8, 1, "lap")
    grp = capi.Context(devices=[0, 0])
    nd, info, u, raw, _ = _solve(capi, grp, nodes, cells, bnd, 1, f, op)
    twin = grp.clone()
    twin.M, twin.N = grp.M, grp.N
    assert twin.devices()["devices"] == [0, 0] and twin.devices()["form"] == grp.devices()["form"]
    assert np.array_equal(twin.solution(), u) and np.array_equal(twin.force(), grp.force())
    assert np.array_equal(twin.matrix_values(capi.MAT_STIFF), grp.matrix_values(capi.MAT_STIFF))
    assert np.array_equal(twin.matrix_values(capi.MAT_MASS), raw[1])
    again = twin.solve(rtol=1e-11)   # (no init on the clone)
    assert again.converged == 1 and np.array_equal(twin.solution(), u)
    _, _, coords = twin.dofs_get()
    twin.set_dirichlet(np.zeros(nd))
    twin.solve(rtol=1e-11)
    assert np.abs(twin.solution() - u).max() > 1e-3 and np.array_equal(grp.solution(), u)
    twin.close(), grp.close()


def test_group_across_real_devices(env):
    """two real GPUs, one context: auto-enabled where the node has them"""
    capi, meshgen, _ = env
    if capi.load().fdapde_device_count() < 2:
        pytest.skip("needs 2 GPUs")
    nodes, cells, bnd, f, op = _problem(capi, meshgen, "cube", 24, 1, "lap")
    one = capi.Context(0)
    _, _, u1, _, _ = _solve(capi, one, nodes, cells, bnd, 1, f, op)
    grp = capi.Context(devices=[0, 1])
    _, info, ug, _, _ = _solve(capi, grp, nodes, cells, bnd, 1, f, op)
    assert info.converged == 1 and np.linalg.norm(ug - u1) <= 1e-9 * np.linalg.norm(u1)
    one.close(), grp.close()


def test_random_multi_device_problems_against_the_single_device_context():
    """tools/fuzz_group.py: 30 random problems (dimension, order, 2 - 5 "devices", form left open or pinned, constant and SPACE-VARYING operator leaves dealt to the
    ranks by cell id, Dirichlet data none / zero / non-zero / on part of the boundary) and a random walk over the entry points (getters, assemble_operator, spmv,
    lump, handle with given values, implicit Euler, new data, clone), every answer against the single-device context's"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_group.py"), "30", "23"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "failures 0" in r.stdout.splitlines()[-1], r.stdout[-500:]
