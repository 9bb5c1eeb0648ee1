"""Seeded synthetic workloads of BASELINE.md / SURVEY.md section 8d (no external mesher, no network).

  C2: [0,1]^2, nx x nx quads -> 2 triangles each, diagonal flipped by Bernoulli(0.5), interior nodes jittered by
      U(-0.2h, 0.2h) per axis, node and cell ids randomly permuted.          nx = 708 -> 1 002 528 triangles.
  C3: [0,1]^3, nx^3 cubes x 6 Kuhn tetrahedra (all sharing the cube's main diagonal, conforming by translation),
      same jitter and permutation.                                            nx = 119 -> 10 110 954 tetrahedra.
The reference takes |det J| (geometry/simplex.h:188), so cell orientation is free; the generator asserts that no cell
is inverted by the jitter (sign of det J unchanged), which on these conforming box meshes rules out overlaps.
Seed 12345, numpy MT19937.
"""
from __future__ import annotations

import itertools

import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.MT19937(seed))


def _det(nodes, cells):
    x0 = nodes[cells[:, 0]]
    J = np.stack([nodes[cells[:, k + 1]] - x0 for k in range(cells.shape[1] - 1)], axis=2)
    return np.linalg.det(J)


def unit_square(nx: int, seed: int = 12345, jitter: float = 0.2, permute: bool = True):
    """-> nodes (n,2) float64, cells (m,3) int32, boundary (n,) uint8"""
    rng = _rng(seed)
    n1 = nx + 1
    h = 1.0 / nx
    gx, gy = np.meshgrid(np.arange(n1), np.arange(n1), indexing="ij")
    ideal = np.stack([gx.ravel() * h, gy.ravel() * h], axis=1)
    interior = ((gx > 0) & (gx < nx) & (gy > 0) & (gy < nx)).ravel()
    nodes = ideal.copy()
    nodes[interior] += rng.uniform(-jitter * h, jitter * h, (int(interior.sum()), 2))
    nid = lambda i, j: i * n1 + j
    ci, cj = np.meshgrid(np.arange(nx), np.arange(nx), indexing="ij")
    ci, cj = ci.ravel(), cj.ravel()
    a, b, c, d = nid(ci, cj), nid(ci + 1, cj), nid(ci + 1, cj + 1), nid(ci, cj + 1)
    flip = rng.random(ci.size) < 0.5
    t1 = np.where(flip[:, None], np.stack([a, b, d], 1), np.stack([a, b, c], 1))
    t2 = np.where(flip[:, None], np.stack([b, c, d], 1), np.stack([a, c, d], 1))
    cells = np.concatenate([t1, t2], axis=0).astype(np.int64)
    assert np.all(np.sign(_det(nodes, cells)) == np.sign(_det(ideal, cells))), "jitter inverted a triangle"
    boundary = (~interior).astype(np.uint8)
    return _finish(nodes, cells, boundary, rng, permute)


_KUHN = np.array([[0] + list(np.cumsum([1 << p for p in perm])) for perm in itertools.permutations(range(3))])  # corner bit codes


def unit_cube(nx: int, seed: int = 12345, jitter: float = 0.2, permute: bool = True):
    """-> nodes (n,3) float64, cells (m,4) int32, boundary (n,) uint8"""
    rng = _rng(seed)
    n1 = nx + 1
    h = 1.0 / nx
    g = np.arange(n1)
    gx, gy, gz = np.meshgrid(g, g, g, indexing="ij")
    ideal = np.stack([gx.ravel() * h, gy.ravel() * h, gz.ravel() * h], axis=1)
    interior = ((gx > 0) & (gx < nx) & (gy > 0) & (gy < nx) & (gz > 0) & (gz < nx)).ravel()
    nodes = ideal.copy()
    nodes[interior] += rng.uniform(-jitter * h, jitter * h, (int(interior.sum()), 3))
    c = np.arange(nx)
    ci, cj, ck = (a.ravel() for a in np.meshgrid(c, c, c, indexing="ij"))
    base = (ci * n1 + cj) * n1 + ck
    # corner with bit code b = (bx | by<<1 | bz<<2) sits at offset bx*n1*n1 + by*n1 + bz
    off = np.array([((b & 1) * n1 + ((b >> 1) & 1)) * n1 + ((b >> 2) & 1) for b in range(8)], dtype=np.int64)
    cells = (base[:, None, None] + off[_KUHN][None, :, :]).reshape(-1, 4)
    assert np.all(np.sign(_det(nodes, cells)) == np.sign(_det(ideal, cells))), "jitter inverted a tetrahedron"
    boundary = (~interior).astype(np.uint8)
    return _finish(nodes, cells, boundary, rng, permute)


def _finish(nodes, cells, boundary, rng, permute):
    if permute:
        pn = rng.permutation(nodes.shape[0])          # new id -> old id
        inv = np.empty_like(pn)
        inv[pn] = np.arange(pn.size)
        nodes, boundary = nodes[pn], boundary[pn]
        cells = inv[cells]
        cells = cells[rng.permutation(cells.shape[0])]
    return np.ascontiguousarray(nodes), np.ascontiguousarray(cells.astype(np.int32)), np.ascontiguousarray(boundary)


def manufactured(N: int):
    """u = prod sin(pi x_d), f = -Lap u = N pi^2 u; homogeneous Dirichlet data on the unit box"""
    u = lambda x: np.prod(np.sin(np.pi * x), axis=1)
    f = lambda x: N * np.pi**2 * np.prod(np.sin(np.pi * x), axis=1)
    return u, f
