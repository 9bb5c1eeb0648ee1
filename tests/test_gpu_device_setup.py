"""The set-up on the device (dev_setup.hip, dev_persist.hip, dev_topology.hip: radix sorts + scans + small kernels) against the
multi-threaded host builder it replaces, array for array: FDAPDE_SETUP_CHECK=1 makes fdapde_dofs_build / fdapde_solver_prepare run
both and fail on the first differing element (numbering permutations, adjacency, both CSR patterns, slot maps, sliced-ELL adjacency,
block tables, order-2 DOF table and coordinates, the persistent CG's layout).  Bit-exact by construction: same Morton quantisation,
stable sorts, same tie-breaking."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def checked_env():
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi, meshgen

    os.environ["FDAPDE_SETUP_CHECK"] = "1"
    yield capi, meshgen
    os.environ.pop("FDAPDE_SETUP_CHECK", None)


@pytest.mark.parametrize("dim,nx,order", [(2, 9, 1), (2, 64, 1), (2, 40, 2), (3, 5, 1), (3, 24, 1), (3, 11, 2), (2, 300, 1), (3, 40, 2)])
def test_device_builders_equal_the_host_builders(checked_env, dim, nx, order):
    capi, meshgen = checked_env
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)            # raises FdapdeError(EHIP) on the first mismatch, details on stderr
    c.solver_prepare(True)              # persistent layout (or the blocked-ELL layout for long-row systems)
    c.solver_prepare(False)
    # and the space is usable: a solve through it
    _, f = meshgen.manufactured(dim)
    c.set_operator(-capi.laplacian())
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    assert c.solve(rtol=1e-10).converged == 1
    c.close()


def test_host_builder_still_serves_on_request(checked_env):
    """FDAPDE_SETUP=host: the multi-threaded host builder + uploads (what device-less contexts use for their queries)"""
    capi, meshgen = checked_env
    os.environ.pop("FDAPDE_SETUP_CHECK", None)
    os.environ["FDAPDE_SETUP"] = "host"
    try:
        nodes, cells, bnd = meshgen.unit_cube(10)
        outs = []
        for mode in ("host", "device"):
            if mode == "device":
                os.environ.pop("FDAPDE_SETUP", None)
            c = capi.Context(0)
            c.mesh_upload(nodes, cells, bnd)
            nd = c.dofs_build(2)
            c.set_operator(-capi.laplacian() + capi.reaction(1.0))
            c.set_forcing(np.ones(c.quadrature_nodes().shape[0]))
            c.init()
            outs.append((c.pattern_get(), c.matrix_values(capi.MAT_STIFF), c.dofs_get()))
            c.close()
        (p0, v0, d0), (p1, v1, d1) = outs
        assert all(np.array_equal(a, b) for a, b in zip(p0, p1)) and np.array_equal(v0, v1)       # same bits whichever builder ran
        assert all(np.array_equal(a, b) for a, b in zip(d0, d1))
    finally:
        os.environ.pop("FDAPDE_SETUP", None)
