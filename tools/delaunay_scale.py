#!/usr/bin/env python3
"""C2 / C3-like problems on genuinely unstructured meshes: Delaunay triangulations (scipy / Qhull) of jittered-grid point clouds, vertex
valences and row lengths as Qhull leaves them (2-D: 3 .. 12 neighbours; 3-D: 8 .. 40), ids permuted.  What bench.py's box meshes cannot show:
that set-up on the device, the row-owner assembly and the single-launch CG keep their pace when the connectivity is irregular.
Prints one line per mesh: sizes, set-up / init / solve times, iterations, us per iteration, the layout the solve ran on, error against the
manufactured solution.  NX2 / NX3 set the point grids (defaults: 708 -> ~1.0 M triangles, 64 -> ~1.6 M tetrahedra)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen


def cloud(dim, nx, seed=7, jitter=0.35):
    """(nx + 1)^dim grid points; interior ones moved by up to jitter * h along every axis (boundary points stay on the boundary: the domain is
    exactly [0,1]^dim and the hull facets are not slivers)"""
    rng = np.random.default_rng(seed)
    ax = np.linspace(0.0, 1.0, nx + 1)
    pts = np.stack(np.meshgrid(*([ax] * dim), indexing="ij"), axis=-1).reshape(-1, dim)
    on_bnd = np.any((pts == 0.0) | (pts == 1.0), axis=1)
    h = 1.0 / nx
    pts[~on_bnd] += rng.uniform(-jitter * h, jitter * h, (int((~on_bnd).sum()), dim))
    return pts, on_bnd


def delaunay(dim, nx):
    from scipy.spatial import Delaunay

    pts, on_bnd = cloud(dim, nx)
    t0 = time.perf_counter()
    tri = Delaunay(pts)
    t_q = time.perf_counter() - t0
    cells = tri.simplices.astype(np.int32)
    J = pts[cells[:, 1:]] - pts[cells[:, :1]]
    vol = np.abs(np.linalg.det(J)) / (2.0 if dim == 2 else 6.0)
    keep = vol > 1e-14   # (co-planar boundary points give Qhull exactly flat simplices on the faces of the box)
    cells = np.ascontiguousarray(cells[keep])
    assert abs(vol[keep].sum() - 1.0) < 1e-9, vol[keep].sum()
    perm = np.random.default_rng(11).permutation(pts.shape[0])   # ids carry no locality
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    return np.ascontiguousarray(pts[perm]), inv[cells].astype(np.int32), on_bnd[perm].astype(np.uint8), t_q, float(vol[keep].min())


def run(dim, nx):
    nodes, cells, bnd, t_q, vmin = delaunay(dim, nx)
    u_exact, f = meshgen.manufactured(dim)
    val = np.bincount(cells.ravel(), minlength=nodes.shape[0])
    c = capi.Context(0)
    t0 = time.perf_counter()
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(1)
    c.solver_prepare(True)
    t_setup = time.perf_counter() - t0
    c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd))
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        c.init()
        i = c.solve(rtol=1e-10)
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, i)
    wall, i = best
    _, _, coords = c.dofs_get()
    err = float(np.max(np.abs(c.solution() - u_exact(coords))))
    us = 1e3 * (i.launch_ms if i.persistent else i.t_solve_ms) / max(i.iters, 1)
    print(f"{dim}-D Delaunay, {nx + 1}^{dim} points: {cells.shape[0]} cells, {nd} DOFs (Qhull {t_q:.1f} s on the host, smallest cell {vmin:.2e}), cells per vertex "
          f"{int(val.min())}..{int(val.max())} | set-up {1e3 * t_setup:.0f} ms, init {i.t_assemble_ms:.3f} ms, solve {i.t_solve_ms:.2f} ms, {i.iters} iterations, "
          f"{us:.2f} us per iteration, persistent {i.persistent}, layout {c.solver_layout_kind(True)}, relres {i.relres:.2e}, {nd / wall / 1e6:.1f} M DOF/s, "
          f"max |u - u_exact| {err:.2e}", flush=True)
    c.close()


if __name__ == "__main__":
    run(2, int(os.environ.get("NX2", "708")))
    run(3, int(os.environ.get("NX3", "64")))
