"""ctypes front-end of the CPU oracle (oracle/fem_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
package (fdapde-core_amd/) never does.  See the header of fem_oracle.c for the parity-pinning status.

Also holds the fixture readers that restate the reference's test-side loaders:
  * read_csv        -- utils/IO/csv_reader.h:75-117 dialect (header row, first column = row index, quotes stripped)
  * load_mesh       -- test/src/utils/mesh_loader.h:62-84 (elements 1-based -> 0-based)
  * read_mtx        -- MatrixMarket coordinate files the reference's tests compare against
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfem_oracle.so")

LAPLACIAN, DIFFUSION, ADVECTION, REACTION, DT = 0, 1, 2, 3, 4


class _Term(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("space_varying", C.c_int32),
        ("coef", C.c_double),
        ("cst", C.c_double * 9),
        ("data", C.POINTER(C.c_double)),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc -O2)."""
    src = os.path.join(_HERE, "fem_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libfem_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.fo_poly_eval.restype = C.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _dp(a):
    return _p(a, C.c_double)


def _ip(a):
    return _p(a, C.c_int32)


def _bp(a):
    return _p(a, C.c_uint8)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with status {rc}")


# ------------------------------------------------------------------------------------------------ fixtures
def read_csv(path: str, dtype=float) -> np.ndarray:
    rows = []
    with open(path) as f:
        f.readline()
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            toks = [t.replace('"', "").strip() for t in line.split(",")][1:]
            rows.append([np.nan if t in ("NA", "NaN", "nan") else float(t) for t in toks])
    a = np.asarray(rows, dtype=float)
    return a.astype(dtype) if dtype is not float else a


def read_mtx(path: str) -> np.ndarray:
    """Dense array from a MatrixMarket 'coordinate real general' file (1-based indices, duplicates summed)."""
    with open(path) as f:
        line = f.readline()
        assert line.startswith("%%MatrixMarket")
        while True:
            line = f.readline()
            if not line.startswith("%"):
                break
        nr, nc, nnz = (int(t) for t in line.split())
        out = np.zeros((nr, nc))
        for _ in range(nnz):
            r, c, v = f.readline().split()
            out[int(r) - 1, int(c) - 1] += float(v)
    return out


@dataclass
class Mesh:
    """nodes: (n_nodes, N) C-order here; the oracle and the C-ABI take column-major (reference layout,
    geometry/triangulation.h:119) via .nodes_colmajor; cells row-major int32 0-based; boundary uint8 per node."""

    nodes: np.ndarray
    cells: np.ndarray
    boundary: np.ndarray

    @property
    def M(self):
        return self.cells.shape[1] - 1

    @property
    def N(self):
        return self.nodes.shape[1]

    @property
    def n_nodes(self):
        return self.nodes.shape[0]

    @property
    def n_cells(self):
        return self.cells.shape[0]

    @property
    def nodes_colmajor(self) -> np.ndarray:
        return np.ascontiguousarray(self.nodes.T).reshape(-1)


def load_mesh(directory: str) -> Mesh:
    pts = read_csv(os.path.join(directory, "points.csv"))
    el = read_csv(os.path.join(directory, "elements.csv")).astype(np.int32) - 1
    bd = read_csv(os.path.join(directory, "boundary.csv")).astype(np.uint8).reshape(-1)
    return Mesh(np.ascontiguousarray(pts), np.ascontiguousarray(el), np.ascontiguousarray(bd))


# ------------------------------------------------------------------------------------------------ topology
def topology(mesh: Mesh) -> dict:
    """The rest of the Triangulation constructor, restated literally (plain Python loops + dicts: fixtures only).
    Triangles  (fdaPDE/geometry/triangulation.h:150-193): cells ascending x combinations<2,3> = (0,1),(0,2),(1,2); an edge never seen
      gets the next id, `edge_to_cells = {i, -1}`, marker true; the second cell on it clears the marker, fills edge_to_cells[1], sets
      neighbors_(k, v) = i and neighbors_(i, v') = k with v = the first local vertex of the cell that is not a node of the edge
      (node_opposite_to_edge, 156-167), and ERASES the map entry (188).
    Tetrahedra (triangulation.h:348-388): the same with faces over combinations<3,4>; a newly seen face registers its three edges
      (combinations<2,3> of its sorted nodes) in first-seen order, marker = both end nodes on the boundary (371).
    Returns 0-based int32 arrays: neighbors, cell_facets, facet_nodes, facet_cells, facet_boundary (+ edge_nodes, edge_boundary,
    face_edges for tetrahedra)."""
    import itertools

    M, nc = mesh.M, mesh.n_cells
    cells = mesh.cells
    pattern = list(itertools.combinations(range(M + 1), M))          # utils/combinatorics.h:37-51: lexicographic
    edge_pattern = list(itertools.combinations(range(3), 2))
    neighbors = -np.ones((nc, M + 1), dtype=np.int32)                # triangulation.h:57
    cell_facets = np.zeros((nc, M + 1), dtype=np.int32)
    facet_nodes, facet_cells, markers = [], [], []
    fmap = {}
    edge_nodes, edge_markers, face_edges, emap = [], [], [], {}

    def opposite(facet_id, cell):                                    # node_opposite_to_edge / node_opposite_to_face
        for j in range(M + 1):
            if cells[cell, j] not in facet_nodes[facet_id]:
                return j
        return M + 1

    for i in range(nc):
        for j, pat in enumerate(pattern):
            facet = tuple(sorted(int(cells[i, k]) for k in pat))
            hit = fmap.get(facet)
            if hit is None:
                fid = len(facet_nodes)
                facet_nodes.append(facet), facet_cells.append([i, -1]), markers.append(1)
                fmap[facet] = (fid, i)
                cell_facets[i, j] = fid
                if M == 3:
                    for (a, b) in edge_pattern:
                        edge = tuple(sorted((facet[a], facet[b])))
                        if edge not in emap:
                            emap[edge] = len(edge_nodes)
                            edge_nodes.append(edge)
                            edge_markers.append(1 if (mesh.boundary[edge[0]] and mesh.boundary[edge[1]]) else 0)
                        face_edges.append(emap[edge])
            else:
                h, k = hit
                neighbors[k, opposite(h, k)] = i
                neighbors[i, opposite(h, i)] = k
                cell_facets[i, j] = h
                markers[h] = 0
                facet_cells[h][1] = i
                del fmap[facet]
    out = dict(neighbors=neighbors, cell_facets=cell_facets, facet_nodes=np.asarray(facet_nodes, dtype=np.int32).reshape(-1, M),
               facet_cells=np.asarray(facet_cells, dtype=np.int32).reshape(-1, 2), facet_boundary=np.asarray(markers, dtype=np.uint8))
    if M == 3:
        out.update(edge_nodes=np.asarray(edge_nodes, dtype=np.int32).reshape(-1, 2), edge_boundary=np.asarray(edge_markers, dtype=np.uint8),
                   face_edges=np.asarray(face_edges, dtype=np.int32).reshape(-1, 3))
    return out


# ------------------------------------------------------------------------------------------------ tables
def n_basis(M, R):
    return lib().fo_n_basis(M, R)


def quadrature(M, R=None, nq=None):
    if nq is None:
        nq = lib().fo_quadrature_rule(M, R)
    nodes = np.zeros((nq, M))
    w = np.zeros(nq)
    _check(lib().fo_quadrature_table(M, nq, _dp(nodes), _dp(w)), "quadrature_table")
    return nodes, w


def reference_nodes(M, R):
    out = np.zeros((n_basis(M, R), M))
    _check(lib().fo_reference_nodes(M, R, _dp(out)), "reference_nodes")
    return out


def reference_basis(M, R):
    nb = n_basis(M, R)
    coeff = np.zeros((nb, nb))
    _check(lib().fo_reference_basis(M, R, _dp(coeff)), "reference_basis")
    return coeff


def poly_table(M, R):
    nb = n_basis(M, R)
    out = np.zeros((nb, M), dtype=np.int32)
    lib().fo_poly_table(M, R, _ip(out))
    return out


def poly_eval(M, R, coeff_i, p):
    c = np.ascontiguousarray(coeff_i, dtype=float)
    p = np.ascontiguousarray(p, dtype=float)
    return lib().fo_poly_eval(M, R, _dp(c), _dp(p))


def poly_grad(M, R, coeff_i, p):
    c = np.ascontiguousarray(coeff_i, dtype=float)
    p = np.ascontiguousarray(p, dtype=float)
    g = np.zeros(M)
    lib().fo_poly_grad(M, R, _dp(c), _dp(p), _dp(g))
    return g


def basis_tables(M, R):
    nb, nq = n_basis(M, R), lib().fo_quadrature_rule(M, R)
    psi = np.zeros((nb, nq))
    dpsi = np.zeros((nb, nq, M))
    _check(lib().fo_basis_tables(M, R, _dp(psi), _dp(dpsi)), "basis_tables")
    return psi, dpsi


# ------------------------------------------------------------------------------------------------ geometry / dofs
def cell_geometry(mesh: Mesh, cell_id: int):
    M = mesh.M
    J = np.zeros((M, M))
    invJ = np.zeros((M, M))
    meas = C.c_double()
    nodes = mesh.nodes_colmajor
    cell = np.ascontiguousarray(mesh.cells[cell_id])
    _check(lib().fo_cell_geometry(M, C.c_int64(mesh.n_nodes), _dp(nodes), _ip(cell), _dp(J), _dp(invJ), C.byref(meas)), "geom")
    return J, invJ, meas.value


def enumerate_dofs(mesh: Mesh, order: int):
    """-> dofs (n_cells, n_basis) int32, boundary_dofs (n_dofs,) uint8, n_dofs, n_edges"""
    nb = n_basis(mesh.M, order)
    dofs = np.zeros((mesh.n_cells, nb), dtype=np.int32)
    bnd = np.zeros(mesh.n_nodes + 6 * mesh.n_cells, dtype=np.uint8)
    nd, ne = C.c_int32(), C.c_int32()
    _check(
        lib().fo_enumerate_dofs(mesh.M, order, C.c_int64(mesh.n_nodes), C.c_int64(mesh.n_cells), _ip(mesh.cells),
                                _bp(mesh.boundary), _ip(dofs), _bp(bnd), C.byref(nd), C.byref(ne)),
        "enumerate_dofs",
    )
    return dofs, bnd[: nd.value].copy(), nd.value, ne.value


def dofs_coords(mesh: Mesh, order: int, dofs: np.ndarray, n_dofs: int) -> np.ndarray:
    out = np.zeros(mesh.M * n_dofs)
    nodes = mesh.nodes_colmajor
    _check(
        lib().fo_dofs_coords(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), C.c_int64(mesh.n_cells), _ip(mesh.cells),
                             _ip(dofs), C.c_int64(n_dofs), _dp(out)),
        "dofs_coords",
    )
    return np.ascontiguousarray(out.reshape(mesh.M, n_dofs).T)


def quadrature_nodes(mesh: Mesh, order: int) -> np.ndarray:
    nq = lib().fo_quadrature_rule(mesh.M, order)
    out = np.zeros(mesh.M * nq * mesh.n_cells)
    nodes = mesh.nodes_colmajor
    _check(
        lib().fo_quadrature_nodes(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), C.c_int64(mesh.n_cells), _ip(mesh.cells), _dp(out)),
        "quadrature_nodes",
    )
    return np.ascontiguousarray(out.reshape(mesh.M, nq * mesh.n_cells).T)


# ------------------------------------------------------------------------------------------------ operators
class Operator:
    """Left-to-right sum of scaled leaves; mirrors the reference's operator algebra
    (pde/differential_expressions.h): `-laplacian()`, `a + b`, `a - b`, `2.0 * a`."""

    def __init__(self, terms=None):
        self.terms = list(terms or [])  # (kind, coef, const ndarray | None, data ndarray | None)

    def __neg__(self):
        return Operator([(k, -c, cst, d) for (k, c, cst, d) in self.terms])

    def __add__(self, o):
        return Operator(self.terms + o.terms)

    def __sub__(self, o):
        return Operator(self.terms + (-o).terms)

    def __rmul__(self, s):
        return Operator([(k, float(s) * c, cst, d) for (k, c, cst, d) in self.terms])

    @property
    def is_symmetric(self):
        return all(k != ADVECTION for (k, _, _, _) in self.terms)

    def c_terms(self):
        arr = (_Term * len(self.terms))()
        keep = []
        for t, (k, c, cst, d) in zip(arr, self.terms):
            t.kind, t.coef = k, c
            t.space_varying = 0 if d is None else 1
            if cst is not None:
                flat = np.asarray(cst, dtype=float).reshape(-1)
                for i, v in enumerate(flat):
                    t.cst[i] = v
            if d is not None:
                dd = np.ascontiguousarray(d, dtype=float)
                keep.append(dd)
                t.data = _dp(dd)
        return arr, keep


def laplacian():
    return Operator([(LAPLACIAN, 1.0, None, None)])


def diffusion(K):
    K = np.asarray(K, dtype=float)
    return Operator([(DIFFUSION, 1.0, K, None)]) if K.ndim == 2 and K.shape[0] == K.shape[1] and K.shape[0] <= 3 else Operator([(DIFFUSION, 1.0, None, K)])


def diffusion_field(Kq):
    return Operator([(DIFFUSION, 1.0, None, np.asarray(Kq, dtype=float))])


def advection(b):
    return Operator([(ADVECTION, 1.0, np.asarray(b, dtype=float), None)])


def advection_field(bq):
    return Operator([(ADVECTION, 1.0, None, np.asarray(bq, dtype=float))])


def reaction(c):
    return Operator([(REACTION, 1.0, np.asarray([c], dtype=float), None)])


def reaction_field(cq):
    return Operator([(REACTION, 1.0, None, np.asarray(cq, dtype=float))])


def dt():
    return Operator([(DT, 1.0, None, None)])


def local_matrix(mesh: Mesh, order: int, cell_id: int, op: Operator) -> np.ndarray:
    nb = n_basis(mesh.M, order)
    out = np.zeros((nb, nb))
    terms, keep = op.c_terms()
    nodes = mesh.nodes_colmajor
    cell = np.ascontiguousarray(mesh.cells[cell_id])
    _check(
        lib().fo_local_matrix(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), _ip(cell), C.c_int64(cell_id), len(op.terms), terms, _dp(out)),
        "local_matrix",
    )
    return out


def physical_gradients_at(mesh: Mesh, order: int, cell_id: int, p) -> np.ndarray:
    nb = n_basis(mesh.M, order)
    out = np.zeros((nb, mesh.M))
    nodes = mesh.nodes_colmajor
    cell = np.ascontiguousarray(mesh.cells[cell_id])
    p = np.ascontiguousarray(p, dtype=float)
    _check(lib().fo_physical_gradients_at(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), _ip(cell), _dp(p), _dp(out)), "phys_grad")
    return out


@dataclass
class CSR:
    rowptr: np.ndarray
    colidx: np.ndarray
    values: np.ndarray
    n: int

    def to_scipy(self):
        import scipy.sparse as sp

        return sp.csr_matrix((self.values, self.colidx, self.rowptr), shape=(self.n, self.n))

    def matvec(self, x):
        y = np.zeros(self.n)
        x = np.ascontiguousarray(x, dtype=float)
        lib().fo_spmv(C.c_int64(self.n), _ip(self.rowptr), _ip(self.colidx), _dp(self.values), _dp(x), _dp(y))
        return y

    def copy(self):
        return CSR(self.rowptr.copy(), self.colidx.copy(), self.values.copy(), self.n)


def assemble_operator(mesh: Mesh, order: int, dofs: np.ndarray, n_dofs: int, op: Operator) -> CSR:
    terms, keep = op.c_terms()
    rp, ci, va = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()
    nnz = C.c_int64()
    nodes = mesh.nodes_colmajor
    dofs = np.ascontiguousarray(dofs, dtype=np.int32)
    _check(
        lib().fo_assemble_operator(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), C.c_int64(mesh.n_cells), _ip(mesh.cells), _ip(dofs),
                                   C.c_int64(n_dofs), len(op.terms), terms, C.byref(rp), C.byref(ci), C.byref(va), C.byref(nnz)),
        "assemble_operator",
    )
    try:
        rowptr = np.ctypeslib.as_array(rp, shape=(n_dofs + 1,)).copy()
        colidx = np.ctypeslib.as_array(ci, shape=(max(nnz.value, 1),))[: nnz.value].copy()
        values = np.ctypeslib.as_array(va, shape=(max(nnz.value, 1),))[: nnz.value].copy()
    finally:
        lib().fo_free(rp), lib().fo_free(ci), lib().fo_free(va)
    return CSR(rowptr, colidx, values, n_dofs)


def assemble_forcing(mesh: Mesh, order: int, dofs: np.ndarray, n_dofs: int, f_q: np.ndarray) -> np.ndarray:
    b = np.zeros(n_dofs)
    nodes = mesh.nodes_colmajor
    f_q = np.ascontiguousarray(f_q, dtype=float).reshape(-1)
    dofs = np.ascontiguousarray(dofs, dtype=np.int32)
    _check(
        lib().fo_assemble_forcing(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), C.c_int64(mesh.n_cells), _ip(mesh.cells), _ip(dofs),
                                  C.c_int64(n_dofs), _dp(f_q), _dp(b)),
        "assemble_forcing",
    )
    return b


def set_dirichlet(A: CSR, force: np.ndarray, boundary_dofs: np.ndarray, g: np.ndarray):
    """In place: FEMSolverBase::set_dirichlet_bc (fem_solver_base.h:142-155)."""
    g = np.ascontiguousarray(g, dtype=float).reshape(-1)
    bd = np.ascontiguousarray(boundary_dofs, dtype=np.uint8)
    _check(lib().fo_set_dirichlet(C.c_int64(A.n), _ip(A.rowptr), _ip(A.colidx), _dp(A.values), _dp(force), _bp(bd), _dp(g)), "set_dirichlet")


def solve_direct(A: CSR, b: np.ndarray) -> np.ndarray:
    """Sparse LU (SuperLU via scipy; Eigen's SparseLU descends from it) standing in for
    fem_linear_elliptic_solver.h:38-47 on test-sized systems."""
    import scipy.sparse.linalg as spla

    return spla.splu(A.to_scipy().tocsc()).solve(np.asarray(b, dtype=float))


def _krylov(fn, A: CSR, force, boundary_dofs, g, rtol, maxit):
    u = np.zeros(A.n)
    it, rr = C.c_int(), C.c_double()
    force = np.ascontiguousarray(force, dtype=float)
    bd = np.ascontiguousarray(boundary_dofs, dtype=np.uint8) if boundary_dofs is not None else None
    gg = np.ascontiguousarray(g, dtype=float).reshape(-1) if g is not None else np.zeros(A.n)
    rc = fn(C.c_int64(A.n), _ip(A.rowptr), _ip(A.colidx), _dp(A.values), _dp(force), _bp(bd) if bd is not None else None, _dp(gg),
            C.c_double(rtol), int(maxit), _dp(u), C.byref(it), C.byref(rr))
    return u, it.value, rr.value, rc


def pcg(A: CSR, force, boundary_dofs=None, g=None, rtol=1e-10, maxit=10000):
    return _krylov(lib().fo_pcg, A, force, boundary_dofs, g, rtol, maxit)


def bicgstab(A: CSR, force, boundary_dofs=None, g=None, rtol=1e-10, maxit=10000):
    return _krylov(lib().fo_bicgstab, A, force, boundary_dofs, g, rtol, maxit)


# ---- "CPU-best" column: the same restatement on all host cores, oracle/fem_oracle_mt.c ---------------------
_lib_mt = None


def _cpu_tag() -> str:
    """Short hash of this host's CPU model + feature flags: the -march=native build is only ever loaded where it was made."""
    import hashlib

    text = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith(("model name", "flags")):
                    text += line
                if line.strip() == "" and text:
                    break
    except OSError:
        pass
    return hashlib.md5(text.encode()).hexdigest()[:10]


def lib_mt() -> C.CDLL:
    """libfem_oracle_mt.<cpu tag>.so (gcc -O3 -march=native -fopenmp), built on demand on the machine that runs it."""
    global _lib_mt
    if _lib_mt is None:
        name = f"libfem_oracle_mt.{_cpu_tag()}.so"
        path = os.path.join(_HERE, name)
        src = [os.path.join(_HERE, f) for f in ("fem_oracle.c", "fem_oracle_mt.c")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in src):
            subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "libfem_oracle_mt.so", f"MT_OUT={name}"])
        _lib_mt = C.CDLL(path)
        _lib_mt.fo_mt_threads.restype = C.c_int
    return _lib_mt


def mt_threads() -> int:
    return int(lib_mt().fo_mt_threads())


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (cpu.max); threads beyond the
    quota are throttled by the scheduler and make an OpenMP run slower, not faster."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def mt_set_threads(n: int) -> None:
    lib_mt().fo_mt_set_threads(int(n))


def mt_colour_cells(dofs: np.ndarray, n_dofs: int):
    """Greedy colouring of the cells (no two cells of a colour share a DOF): (order, colour_ptr).  Serial set-up."""
    dofs = np.ascontiguousarray(dofs, dtype=np.int32)
    order = np.zeros(dofs.shape[0], dtype=np.int32)
    cptr = np.zeros(257, dtype=np.int64)
    ncol = C.c_int32()
    _check(lib_mt().fo_mt_colour_cells(C.c_int64(n_dofs), C.c_int64(dofs.shape[0]), int(dofs.shape[1]), _ip(dofs), _ip(order),
                                       cptr.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(ncol)), "mt_colour_cells")
    return order, cptr[: ncol.value + 1].copy()


def mt_assemble(mesh: Mesh, order: int, dofs: np.ndarray, n_dofs: int, op: Operator, pattern: CSR, colouring, f_q=None):
    """Operator values into `pattern` (+ the forcing vector when f_q is given), colour class by colour class on all cores."""
    terms, keep = op.c_terms()
    cell_order, cptr = colouring
    nodes = mesh.nodes_colmajor
    dofs = np.ascontiguousarray(dofs, dtype=np.int32)
    values = np.empty(pattern.colidx.shape[0])
    b = np.empty(n_dofs) if f_q is not None else None
    fq = np.ascontiguousarray(f_q, dtype=float).reshape(-1) if f_q is not None else None
    _check(
        lib_mt().fo_mt_assemble(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), C.c_int64(mesh.n_cells), _ip(mesh.cells), _ip(dofs),
                                C.c_int64(n_dofs), len(op.terms), terms, _ip(pattern.rowptr), _ip(pattern.colidx), int(len(cptr) - 1),
                                cptr.ctypes.data_as(C.POINTER(C.c_int64)), _ip(cell_order), _dp(values),
                                _dp(fq) if fq is not None else None, _dp(b) if b is not None else None),
        "mt_assemble",
    )
    return CSR(pattern.rowptr, pattern.colidx, values, n_dofs), b


def mt_pcg(A: CSR, force, boundary_dofs=None, g=None, rtol=1e-10, maxit=10000):
    return _krylov(lib_mt().fo_mt_pcg, A, force, boundary_dofs, g, rtol, maxit)


def pointwise_psi(mesh: Mesh, order: int, dofs, n_dofs, locs) -> np.ndarray:
    locs = np.asarray(locs, dtype=float)
    nl = locs.shape[0]
    out = np.zeros((nl, n_dofs))
    nodes = mesh.nodes_colmajor
    lc = np.ascontiguousarray(locs.T).reshape(-1)
    dofs = np.ascontiguousarray(dofs, dtype=np.int32)
    _check(
        lib().fo_pointwise_psi(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), C.c_int64(mesh.n_cells), _ip(mesh.cells), _ip(dofs),
                               C.c_int64(n_dofs), C.c_int64(nl), _dp(lc), _dp(out)),
        "pointwise_psi",
    )
    return out


def areal_psi(mesh: Mesh, order: int, dofs, n_dofs, incidence):
    inc = np.ascontiguousarray(incidence, dtype=float)
    ns = inc.shape[0]
    out = np.zeros((ns, n_dofs))
    D = np.zeros(ns)
    nodes = mesh.nodes_colmajor
    dofs = np.ascontiguousarray(dofs, dtype=np.int32)
    _check(
        lib().fo_areal_psi(mesh.M, order, C.c_int64(mesh.n_nodes), _dp(nodes), C.c_int64(mesh.n_cells), _ip(mesh.cells), _ip(dofs),
                           C.c_int64(n_dofs), C.c_int64(ns), _dp(inc), _dp(out), _dp(D)),
        "areal_psi",
    )
    return out, D


# ------------------------------------------------------------------------------------------------ full path
@dataclass
class Solved:
    stiff: CSR  # after init(): the assembled operator; after solve(): row-zeroed as the reference leaves it
    mass: CSR
    force: np.ndarray
    solution: np.ndarray
    dofs: np.ndarray
    boundary_dofs: np.ndarray
    n_dofs: int
    dof_coords: np.ndarray


def pde_init_solve(mesh: Mesh, order: int, op: Operator, forcing_q=None, forcing_fn=None, dirichlet=None, direct=True,
                   rtol=1e-12) -> Solved:
    """PDE::init + PDE::solve (pde/pde.h:101-105 -> fem_solver_base.h:104-155 -> fem_linear_elliptic_solver.h:34-50)."""
    dofs, bnd, nd, _ = enumerate_dofs(mesh, order)
    coords = dofs_coords(mesh, order, dofs, nd)
    if forcing_q is None:
        qn = quadrature_nodes(mesh, order)
        forcing_q = np.array([forcing_fn(p) for p in qn]) if forcing_fn is not None else np.zeros(qn.shape[0])
    A = assemble_operator(mesh, order, dofs, nd, op)
    b = assemble_forcing(mesh, order, dofs, nd, forcing_q)
    Mm = assemble_operator(mesh, order, dofs, nd, reaction(1.0))  # fem_solver_base.h:136
    if dirichlet is not None:
        g = np.array([dirichlet(p) for p in coords]) if callable(dirichlet) else np.asarray(dirichlet, dtype=float).reshape(-1)
        if direct:
            set_dirichlet(A, b, bnd, g)
            u = solve_direct(A, b)
        else:
            fn = pcg if op.is_symmetric else bicgstab
            u, it, rr, rc = fn(A, b, bnd, g, rtol=rtol)
            assert rc == 0, (rc, it, rr)
            set_dirichlet(A, b, bnd, g)
    else:
        u = solve_direct(A, b)
    return Solved(A, Mm, b, u, dofs, bnd, nd, coords)


def pde_parabolic_solve(mesh: Mesh, order: int, op: Operator, times, forcing_q, dirichlet, initial_condition):
    """PDE::init + FEMLinearParabolicSolver::solve (fem_linear_parabolic_solver.h:37-72): implicit Euler with one LU of
    K = M/dt + A (Dirichlet rows zeroed, unit diagonal).  forcing_q: (nq*n_cells, m); dirichlet: (n_dofs, m) or None.
    -> solution (n_dofs, m), mass CSR"""
    import scipy.sparse.linalg as spla

    times = np.asarray(times, dtype=float).reshape(-1)
    m = times.size
    dofs, bnd, nd, _ = enumerate_dofs(mesh, order)
    A = assemble_operator(mesh, order, dofs, nd, op)
    Mm = assemble_operator(mesh, order, dofs, nd, reaction(1.0))
    force = np.stack([assemble_forcing(mesh, order, dofs, nd, forcing_q[:, j]) for j in range(m)], axis=1)
    dt_ = times[1] - times[0]                                  # line 42
    K = CSR(A.rowptr.copy(), A.colidx.copy(), Mm.values / dt_ + A.values, nd)   # line 49 (same pattern)
    isb = bnd.astype(bool) if dirichlet is not None else np.zeros(nd, bool)
    if dirichlet is not None:
        dummy = np.zeros(nd)
        set_dirichlet(K, dummy, bnd, np.zeros(nd))             # lines 51-54: rows zeroed, unit diagonal
    lu = spla.splu(K.to_scipy().tocsc())
    sol = np.zeros((nd, m))
    sol[:, 0] = np.asarray(initial_condition, dtype=float).reshape(-1)          # line 46
    for i in range(m - 1):
        rhs = Mm.matvec(sol[:, i]) / dt_ + force[:, i + 1]     # line 63
        if dirichlet is not None:
            rhs[isb] = np.asarray(dirichlet)[isb, i + 1]       # lines 65-67
        sol[:, i + 1] = lu.solve(rhs)
    return sol, Mm
