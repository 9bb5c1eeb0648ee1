"""Triangulation topology tables (fdaPDE/geometry/triangulation.h:143-196, 319-399): neighbours, facets (edges / faces), markers.
  * not gpu: the oracle's literal restatement of the constructor against the data files the reference's own test suite holds
    (test/data/mesh/*/neigh.csv: neighbour across the facet opposite to each local vertex, 1-based, -1 = none; edges.csv: the
    edges of the triangle meshes / the faces of unit_sphere, as a set -- the files are not in first-seen order);
  * gpu: the device builder (dev_topology.hip: stable radix sorts + scans instead of the hash-map walk) against the oracle,
    bit for bit, on every fixture and on generated meshes with permuted ids."""
import os

import numpy as np
import pytest

FIXTURES = ["unit_square_16", "unit_square_32", "unit_square_64", "unit_square", "c_shaped", "quasi_circle", "unit_sphere"]
KEYS2 = ["neighbors", "cell_facets", "facet_nodes", "facet_cells", "facet_boundary"]
KEYS3 = KEYS2 + ["edge_nodes", "edge_boundary", "face_edges"]


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_topology_matches_the_reference_fixtures(oracle, mesh_loader, golden_dir, name):
    m = mesh_loader(name)
    t = oracle.topology(m)
    d = os.path.join(golden_dir, "mesh", name)
    neigh = oracle.read_csv(os.path.join(d, "neigh.csv")).astype(np.int64)
    assert np.array_equal(np.where(neigh > 0, neigh - 1, -1), t["neighbors"])
    facets = np.sort(oracle.read_csv(os.path.join(d, "edges.csv")).astype(np.int64) - 1, axis=1)
    assert len(facets) == len(t["facet_nodes"]) and set(map(tuple, facets)) == set(map(tuple, t["facet_nodes"]))
    # markers: a boundary facet has one cell; its nodes are boundary nodes of the fixture
    bf = t["facet_boundary"].astype(bool)
    assert np.array_equal(bf, t["facet_cells"][:, 1] < 0)
    assert np.all(m.boundary[t["facet_nodes"][bf]] == 1)
    # cell_facets is consistent with facet_nodes
    for c in (0, m.n_cells // 2, m.n_cells - 1):
        for j in range(m.M + 1):
            assert set(t["facet_nodes"][t["cell_facets"][c, j]]) <= set(m.cells[c])
    if m.M == 3:   # SURVEY 8c invariants of unit_sphere
        assert len(t["facet_nodes"]) == 5795 and len(t["edge_nodes"]) == 3606


def _check(capi, oracle, mesh):
    c = capi.Context(0)
    c.mesh_upload(mesh.nodes, mesh.cells, mesh.boundary)
    got, ref = c.topology(), oracle.topology(mesh)
    for k in (KEYS3 if mesh.M == 3 else KEYS2):
        assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), k
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_device_topology_matches_the_oracle_on_fixtures(oracle, mesh_loader, name):
    from fdapde_loader import load_package

    _check(load_package().capi, oracle, mesh_loader(name))


@pytest.mark.gpu
@pytest.mark.parametrize("dim,nx", [(2, 37), (3, 9)])
def test_device_topology_matches_the_oracle_on_generated_meshes(oracle, dim, nx):
    from fdapde_loader import load_package

    pkg = load_package()
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _check(pkg.capi, oracle, oracle.Mesh(nodes, cells, bnd))


@pytest.mark.gpu
def test_device_topology_at_full_size():
    """C3 (10.1 M tetrahedra): counts by Euler's formula for the Kuhn triangulation of a 119^3 box, symmetry of the neighbour
    relation, every interior face shared by two cells"""
    from fdapde_loader import load_package

    pkg = load_package()
    from fdapde_core_amd import meshgen

    nx = 119
    nodes, cells, bnd = meshgen.unit_cube(nx)
    c = pkg.capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    t = c.topology()
    nc = cells.shape[0]
    n_bfaces = 6 * nx * nx * 2
    assert len(t["facet_nodes"]) == (4 * nc + n_bfaces) // 2 and int(t["facet_boundary"].sum()) == n_bfaces
    # Euler: V - E + F - C = 1 for a ball
    assert nodes.shape[0] - len(t["edge_nodes"]) + len(t["facet_nodes"]) - nc == 1
    nb = t["neighbors"]
    has = nb >= 0
    rows = np.repeat(np.arange(nc), 4).reshape(nc, 4)
    assert np.all((nb[nb[has]] == rows[has][:, None]).sum(axis=1) == 1)   # i is a neighbour of each of its neighbours, exactly once
    c.close()
