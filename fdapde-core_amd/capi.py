"""ctypes binding of the C ABI in include/fdapde_hip.h (libfdapde_hip.so).

Plumbing for tests/ and bench.py only: the product's host side is the header-only C++20 facade in
include/fdapde_amd/ (the reference is a C++ library).  There is no CPU fallback here -- if the shared library is
missing, import fails loudly; if a context has no device, compute calls raise FdapdeError(ENODEVICE).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDAPDE_HIP_LIB") or os.path.join(_HERE, "lib", "libfdapde_hip.so")   # (override: A/B builds of tools/)

OK, EINVAL, ENOMEM, ENODEVICE, EHIP, ENOTINIT, ENOCONV, EUNSUPPORTED, ERCCL = range(9)
LAPLACIAN, DIFFUSION, ADVECTION, REACTION, DT = range(5)
SOLVER_AUTO, SOLVER_CG, SOLVER_BICGSTAB, SOLVER_CG_SR, SOLVER_CG_FUSED, SOLVER_GMRES, SOLVER_DENSE, SOLVER_PMG = range(8)
ASSEMBLY_ROWS, ASSEMBLY_ATOMIC, ASSEMBLY_COLOURED, ASSEMBLY_PARTITIONED, ASSEMBLY_WAVE = range(5)
MAT_STIFF, MAT_MASS = 0, 1


class Term(C.Structure):
    _fields_ = [("kind", C.c_int32), ("space_varying", C.c_int32), ("coef", C.c_double), ("cst", C.c_double * 9),
                ("data", C.POINTER(C.c_double))]


class Options(C.Structure):
    _fields_ = [("method", C.c_int32), ("maxit", C.c_int32), ("rtol", C.c_double), ("assembly", C.c_int32),
                ("check_every", C.c_int32), ("time_spmv", C.c_int32)]


class Info(C.Structure):
    _fields_ = [("iters", C.c_int32), ("converged", C.c_int32), ("relres", C.c_double), ("t_assemble_ms", C.c_double),
                ("t_solve_ms", C.c_double), ("t_setup_ms", C.c_double), ("spmv_avg_ms", C.c_double), ("spmv_timed", C.c_int32),
                ("method_used", C.c_int32), ("persistent", C.c_int32), ("gather_avg_ms", C.c_double), ("update_avg_ms", C.c_double),
                ("spmv_mean_ms", C.c_double), ("launch_ms", C.c_double)]


# every symbol include/fdapde_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "fdapde_abi_version", "fdapde_device_count", "fdapde_ctx_create", "fdapde_ctx_destroy", "fdapde_last_error",
    "fdapde_status_string", "fdapde_mesh_upload", "fdapde_dofs_build", "fdapde_dofs_get", "fdapde_dofs_set_boundary", "fdapde_sizes",
    "fdapde_pattern_get", "fdapde_quadrature_nodes", "fdapde_set_operator", "fdapde_set_forcing", "fdapde_set_dirichlet",
    "fdapde_init", "fdapde_assemble_operator", "fdapde_solver_prepare", "fdapde_solve", "fdapde_matrix_values", "fdapde_lump", "fdapde_force", "fdapde_solution",
    "fdapde_info_get", "fdapde_spmv", "fdapde_bench_spmv", "fdapde_tune", "fdapde_stream", "fdapde_synchronize",
    "fdapde_comm_unique_id", "fdapde_comm_init", "fdapde_halo_setup", "fdapde_solve_parabolic",
    "fdapde_lin_compute", "fdapde_lin_solve", "fdapde_eval_pointwise", "fdapde_cell_integrals", "fdapde_comm_init_callback", "fdapde_comm_set_exchange_callback", "fdapde_halo_setup_peers",
    "fdapde_solver_layout", "fdapde_topology_build", "fdapde_topology_get", "fdapde_comm_allreduce", "fdapde_comm_library", "fdapde_solver_layout_kind", "fdapde_rowdist_setup", "fdapde_ctx_clone", "fdapde_comm_count",
    "fdapde_ctx_create_multi", "fdapde_ctx_devices", "fdapde_partition_build", "fdapde_partition_sizes", "fdapde_partition_get", "fdapde_partition_whole",
    "fdapde_partition_peers",
]
PARTITION_ROWDIST, PARTITION_ELEMENTS = 0, 1

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() or `make -C fdapde-core_amd/csrc`. "
                              "There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        lib.fdapde_last_error.restype = C.c_char_p
        lib.fdapde_status_string.restype = C.c_char_p
        lib.fdapde_stream.restype = C.c_void_p
        lib.fdapde_comm_library.restype = C.c_char_p
        lib.fdapde_ctx_destroy.restype = None
        _lib = lib
    return _lib


class FdapdeError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"[status {status}] {msg}")
        self.status = status


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _bp(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Operator:
    """Left-to-right sum of scaled leaves, mirroring the reference's operator algebra
    (fdaPDE/pde/differential_expressions.h): -laplacian(), a + b, a - b, 2.0 * a."""

    def __init__(self, terms=None):
        self.terms = list(terms or [])  # (kind, coef, const | None, data | None)

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
        arr = (Term * len(self.terms))()
        keep = []
        for t, (k, c, cst, d) in zip(arr, self.terms):
            t.kind, t.coef, t.space_varying = k, c, 0 if d is None else 1
            if cst is not None:
                for i, v in enumerate(np.asarray(cst, dtype=float).reshape(-1)):
                    t.cst[i] = v
            if d is not None:
                dd = np.ascontiguousarray(d, dtype=float)
                keep.append(dd)
                t.data = _dp(dd)
        return arr, keep


def laplacian():
    return Operator([(LAPLACIAN, 1.0, None, None)])


def diffusion(K):
    return Operator([(DIFFUSION, 1.0, np.asarray(K, dtype=float), None)])


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


class Context:
    """One fdapde_ctx.  device=None -> host-only context (numbering / pattern queries only)."""

    def __init__(self, device=0, devices=None):
        """devices=[d0, d1, ...]: ONE context over several devices (fdapde_ctx_create_multi; a device may be named several times)"""
        self.lib = load()
        self._ctx = C.c_void_p()
        if devices is not None:
            dv = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            rc = self.lib.fdapde_ctx_create_multi(dv, len(devices), C.byref(self._ctx))
        else:
            rc = self.lib.fdapde_ctx_create(-1 if device is None else int(device), C.byref(self._ctx))
        if rc != OK:
            raise FdapdeError(rc, self.lib.fdapde_status_string(rc).decode())
        self.M = self.N = 0
        self.n_cells = self.n_nodes = 0

    def devices(self):
        """-> dict(devices=[...], form=0 row-distributed | 1 element partition | -1 single device, t_partition_ms, t_rank_setup_ms)"""
        n = C.c_int32()
        self._check(self.lib.fdapde_ctx_devices(self._ctx, C.byref(n), None, None, None, None))
        dv = (C.c_int32 * n.value)()
        form, tp, tr = C.c_int32(), C.c_double(), C.c_double()
        self._check(self.lib.fdapde_ctx_devices(self._ctx, C.byref(n), dv, C.byref(form), C.byref(tp), C.byref(tr)))
        return dict(devices=list(dv), form=form.value, t_partition_ms=tp.value, t_rank_setup_ms=tr.value)

    # ---- the device-side partitioner (rank processes: every process partitions the whole mesh on its own device and takes its share)
    def partition_build(self, world, form=PARTITION_ROWDIST):
        self._check(self.lib.fdapde_partition_build(self._ctx, int(world), int(form)))
        self._part_world = int(world)

    def partition_get(self, rank):
        """-> dict(nodes (n, N), cells (m, M + 1) local ids, boundary, l2g (global node ids, ascending), cell_ids, owner (rank owning each local node))"""
        nn, nc = C.c_int64(), C.c_int64()
        self._check(self.lib.fdapde_partition_sizes(self._ctx, int(rank), C.byref(nn), C.byref(nc)))
        nn, nc = nn.value, nc.value
        nodes = np.zeros(nn * self.N)
        cells = np.zeros((nc, self.M + 1), dtype=np.int32)
        bnd = np.zeros(nn, dtype=np.uint8)
        l2g, cids, own = np.zeros(nn, dtype=np.int64), np.zeros(nc, dtype=np.int64), np.zeros(nn, dtype=np.int32)
        self._check(self.lib.fdapde_partition_get(self._ctx, int(rank), _dp(nodes), _ip(cells), _bp(bnd), l2g.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  cids.ctypes.data_as(C.POINTER(C.c_int64)), _ip(own)))
        return dict(nodes=np.ascontiguousarray(nodes.reshape(self.N, nn).T), cells=cells, boundary=bnd, l2g=l2g, cell_ids=cids, owner=own)

    def partition_whole(self):
        """-> (rank of every cell, owner of every node, bit mask of the ranks whose sub-mesh holds a node)"""
        part = np.zeros(self.n_cells, dtype=np.int32)
        own = np.zeros(self.n_nodes, dtype=np.int32)
        mask = np.zeros(self.n_nodes, dtype=np.uint64)
        self._check(self.lib.fdapde_partition_whole(self._ctx, _ip(part), _ip(own), mask.ctypes.data_as(C.POINTER(C.c_uint64))))
        return part, own, mask

    def partition_peers(self, rank):
        """element form, P1: (peer_rank, peer_off, peer_dof, owned) for halo_setup_peers"""
        npeers, nshared = C.c_int32(), C.c_int64()
        self._check(self.lib.fdapde_partition_peers(self._ctx, int(rank), C.byref(npeers), None, None, None, None, C.byref(nshared)))
        nn = C.c_int64()
        self._check(self.lib.fdapde_partition_sizes(self._ctx, int(rank), C.byref(nn), None))
        pr = np.zeros(max(npeers.value, 1), dtype=np.int32)
        po = np.zeros(npeers.value + 1, dtype=np.int64)
        pd = np.zeros(max(nshared.value, 1), dtype=np.int32)
        owned = np.zeros(nn.value, dtype=np.uint8)
        self._check(self.lib.fdapde_partition_peers(self._ctx, int(rank), C.byref(npeers), _ip(pr), po.ctypes.data_as(C.POINTER(C.c_int64)), _ip(pd), _bp(owned),
                                                    C.byref(nshared)))
        return pr[:npeers.value], po, pd[:nshared.value], owned

    def clone(self):
        """fdapde_ctx_clone: an independent context with the same problem, assembled state and solution"""
        other = Context.__new__(Context)
        other.lib, other._ctx = self.lib, C.c_void_p()
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("lib", "_ctx")})
        self._check(self.lib.fdapde_ctx_clone(self._ctx, C.byref(other._ctx)))
        return other

    def close(self):
        if self._ctx:
            self.lib.fdapde_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise FdapdeError(rc, self.lib.fdapde_last_error(self._ctx).decode() or self.lib.fdapde_status_string(rc).decode())

    # ---- domain / space
    def mesh_upload(self, nodes, cells, boundary):
        """nodes (n_nodes, N) any order; cells (n_cells, M+1) int32 0-based; boundary (n_nodes,) 0/1"""
        nodes = np.asarray(nodes, dtype=float)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        boundary = np.ascontiguousarray(boundary, dtype=np.uint8).reshape(-1)
        colmajor = np.ascontiguousarray(nodes.T).reshape(-1)
        self.M, self.N = cells.shape[1] - 1, nodes.shape[1]
        self.n_nodes, self.n_cells = nodes.shape[0], cells.shape[0]
        self._check(self.lib.fdapde_mesh_upload(self._ctx, self.M, self.N, C.c_int64(self.n_nodes), _dp(colmajor),
                                                C.c_int64(self.n_cells), _ip(cells), _bp(boundary)))

    def topology(self):
        """Triangulation topology tables (device-built): dict of neighbors, cell_facets, facet_nodes, facet_cells, facet_boundary
        and, for tetrahedra, edge_nodes, edge_boundary, face_edges"""
        nf, ne = C.c_int64(), C.c_int64()
        self._check(self.lib.fdapde_topology_build(self._ctx, C.byref(nf), C.byref(ne)))
        M, nc, nf, ne = self.M, self.n_cells, nf.value, ne.value
        t = dict(neighbors=np.zeros((nc, M + 1), np.int32), cell_facets=np.zeros((nc, M + 1), np.int32), facet_nodes=np.zeros((nf, M), np.int32),
                 facet_cells=np.zeros((nf, 2), np.int32), facet_boundary=np.zeros(nf, np.uint8))
        if M == 3:
            t.update(edge_nodes=np.zeros((ne, 2), np.int32), edge_boundary=np.zeros(ne, np.uint8), face_edges=np.zeros((nf, 3), np.int32))
        self._check(self.lib.fdapde_topology_get(self._ctx, _ip(t["neighbors"]), _ip(t["cell_facets"]), _ip(t["facet_nodes"]),
                                                 _ip(t["facet_cells"]), _bp(t["facet_boundary"]),
                                                 _ip(t["edge_nodes"]) if M == 3 else None, _bp(t["edge_boundary"]) if M == 3 else None,
                                                 _ip(t["face_edges"]) if M == 3 else None))
        return t

    def dofs_build(self, order):
        nd = C.c_int64()
        self._check(self.lib.fdapde_dofs_build(self._ctx, int(order), C.byref(nd)))
        self.order = order
        return nd.value

    def sizes(self):
        nd, nnz, ne = C.c_int64(), C.c_int64(), C.c_int64()
        nb, nq = C.c_int32(), C.c_int32()
        self._check(self.lib.fdapde_sizes(self._ctx, C.byref(nd), C.byref(nnz), C.byref(nb), C.byref(nq), C.byref(ne)))
        return dict(n_dofs=nd.value, nnz=nnz.value, n_basis=nb.value, n_quadrature=nq.value, n_edges=ne.value)

    def dofs_get(self):
        s = self.sizes()
        dofs = np.zeros((self.n_cells, s["n_basis"]), dtype=np.int32)
        bnd = np.zeros(s["n_dofs"], dtype=np.uint8)
        coords = np.zeros(self.N * s["n_dofs"])
        self._check(self.lib.fdapde_dofs_get(self._ctx, _ip(dofs), _bp(bnd), _dp(coords)))
        return dofs, bnd, np.ascontiguousarray(coords.reshape(self.N, s["n_dofs"]).T)

    def dofs_set_boundary(self, bnd):
        bnd = np.ascontiguousarray(bnd, dtype=np.uint8)
        assert bnd.size == self.sizes()["n_dofs"]
        self._check(self.lib.fdapde_dofs_set_boundary(self._ctx, _bp(bnd)))

    def pattern_get(self):
        s = self.sizes()
        rowptr = np.zeros(s["n_dofs"] + 1, dtype=np.int32)
        colidx = np.zeros(s["nnz"], dtype=np.int32)
        self._check(self.lib.fdapde_pattern_get(self._ctx, _ip(rowptr), _ip(colidx)))
        return rowptr, colidx

    def quadrature_nodes(self):
        s = self.sizes()
        rows = s["n_quadrature"] * self.n_cells
        out = np.zeros(self.N * rows)
        self._check(self.lib.fdapde_quadrature_nodes(self._ctx, _dp(out)))
        return np.ascontiguousarray(out.reshape(self.N, rows).T)

    # ---- problem data
    def set_operator(self, op: Operator):
        terms, keep = op.c_terms()
        self._check(self.lib.fdapde_set_operator(self._ctx, len(op.terms), terms))

    def set_forcing(self, f_q):
        if f_q is None:
            self._check(self.lib.fdapde_set_forcing(self._ctx, None, 0))
            self._force_cols = 1
            return
        f = np.asarray(f_q, dtype=float)
        ncols = 1 if f.ndim == 1 else f.shape[1]
        self._force_cols = ncols
        flat = np.ascontiguousarray(f.reshape(f.shape[0], ncols).T).reshape(-1)  # column-major
        self._check(self.lib.fdapde_set_forcing(self._ctx, _dp(flat), ncols))

    def set_dirichlet(self, g):
        if g is None:
            self._check(self.lib.fdapde_set_dirichlet(self._ctx, None))
        else:
            g = np.ascontiguousarray(g, dtype=float).reshape(-1)
            self._check(self.lib.fdapde_set_dirichlet(self._ctx, _dp(g)))

    # ---- compute
    def init(self, assembly=ASSEMBLY_ROWS):
        opt = Options(method=0, maxit=0, rtol=0.0, assembly=assembly, check_every=0, time_spmv=0)
        self._check(self.lib.fdapde_init(self._ctx, C.byref(opt)))

    def assemble_operator(self, which, op: Operator, assembly=ASSEMBLY_ROWS):
        terms, keep = op.c_terms()
        self._check(self.lib.fdapde_assemble_operator(self._ctx, which, len(op.terms), terms, assembly))

    def solver_prepare(self, with_dirichlet=True):
        self._check(self.lib.fdapde_solver_prepare(self._ctx, 1 if with_dirichlet else 0))

    def solve(self, method=SOLVER_AUTO, rtol=1e-10, maxit=0, check_every=0, raise_on_noconv=True, time_spmv=0):
        opt = Options(method=method, maxit=maxit, rtol=rtol, assembly=0, check_every=check_every, time_spmv=time_spmv)
        info = Info()
        rc = self.lib.fdapde_solve(self._ctx, C.byref(opt), C.byref(info))
        if rc != OK and (raise_on_noconv or rc != ENOCONV):
            self._check(rc)
        return info

    def solve_parabolic(self, times, initial_condition, dirichlet=None, method=SOLVER_AUTO, rtol=1e-10, maxit=0):
        """FEMLinearParabolicSolver::solve; dirichlet (n_dofs, n_times) or None -> solution (n_dofs, n_times), Info"""
        times = np.asarray(times, dtype=float).reshape(-1)
        m, nd = times.size, self.sizes()["n_dofs"]
        u0 = np.ascontiguousarray(initial_condition, dtype=float).reshape(-1)
        g = None if dirichlet is None else np.ascontiguousarray(np.asarray(dirichlet, dtype=float).T).reshape(-1)
        out = np.zeros(nd * m)
        opt = Options(method=method, maxit=maxit, rtol=rtol, assembly=0, check_every=0, time_spmv=0)
        info = Info()
        self._check(self.lib.fdapde_solve_parabolic(self._ctx, C.byref(opt), int(m), C.c_double(times[1] - times[0]), _dp(u0),
                                                    None if g is None else _dp(g), _dp(out), C.byref(info)))
        return np.ascontiguousarray(out.reshape(m, nd).T), info

    def lin_compute(self, which=MAT_STIFF, values=None, symmetric=False):
        """fdapde::SparseLU::compute: 'factor once'"""
        v = None if values is None else np.ascontiguousarray(values, dtype=float)
        self._check(self.lib.fdapde_lin_compute(self._ctx, which, None if v is None else _dp(v), 1 if symmetric else 0))

    def lin_solve(self, b, method=SOLVER_AUTO, rtol=1e-10, check_every=0, maxit=0):
        """fdapde::SparseLU::solve(b); b (n_dofs,) or (n_dofs, n_rhs)"""
        b = np.asarray(b, dtype=float)
        one = b.ndim == 1
        B = b.reshape(b.shape[0], -1)
        flat = np.ascontiguousarray(B.T).reshape(-1)
        out = np.zeros_like(flat)
        opt = Options(method=method, maxit=maxit, rtol=rtol, assembly=0, check_every=check_every, time_spmv=0)
        info = Info()
        self._check(self.lib.fdapde_lin_solve(self._ctx, C.byref(opt), _dp(flat), B.shape[1], _dp(out), C.byref(info)))
        X = np.ascontiguousarray(out.reshape(B.shape[1], B.shape[0]).T)
        return (X[:, 0] if one else X), info

    # ---- basis evaluation (PDE__::eval_basis): Psi as scipy CSR + D
    def eval_pointwise(self, locs):
        """pointwise_evaluation::eval -> (Psi csr n_locs x n_dofs, D = ones, cell ids)"""
        import scipy.sparse as sp

        locs = np.asarray(locs, dtype=float)
        nl, s = locs.shape[0], self.sizes()
        flat = np.ascontiguousarray(locs.T).reshape(-1)
        cells = np.zeros(nl, dtype=np.int32)
        vals = np.zeros((nl, s["n_basis"]))
        self._check(self.lib.fdapde_eval_pointwise(self._ctx, C.c_int64(nl), _dp(flat), _ip(cells), _dp(vals)))
        dofs, _, _ = self.dofs_get()
        ok = cells >= 0
        rows = np.repeat(np.nonzero(ok)[0], s["n_basis"])
        cols = dofs[cells[ok]].reshape(-1)
        psi = sp.csr_matrix((vals[ok].reshape(-1), (rows, cols)), shape=(nl, s["n_dofs"]))
        return psi, np.ones(nl), cells

    def eval_areal(self, incidence):
        """areal_evaluation::eval: incidence (n_sub, n_cells) of 0/1 -> (Psi csr n_sub x n_dofs, D = subdomain measures)"""
        import scipy.sparse as sp

        inc = sp.csr_matrix(np.asarray(incidence) == 1, dtype=float)
        s = self.sizes()
        meas = np.zeros(self.n_cells)
        pint = np.zeros((self.n_cells, s["n_basis"]))
        self._check(self.lib.fdapde_cell_integrals(self._ctx, _dp(meas), _dp(pint)))
        dofs, _, _ = self.dofs_get()
        D = inc @ meas
        nb = s["n_basis"]
        cellmat = sp.csr_matrix((pint.reshape(-1), (np.repeat(np.arange(self.n_cells), nb), dofs.reshape(-1))),
                                shape=(self.n_cells, s["n_dofs"]))
        psi = sp.diags(1.0 / D) @ (inc @ cellmat)
        return psi.tocsr(), D

    def info(self):
        info = Info()
        self._check(self.lib.fdapde_info_get(self._ctx, C.byref(info)))
        return info

    # ---- getters
    def matrix_values(self, which=MAT_STIFF):
        out = np.zeros(self.sizes()["nnz"])
        self._check(self.lib.fdapde_matrix_values(self._ctx, which, _dp(out)))
        return out

    def lump(self, which=MAT_MASS):
        """diagonal of lump(mass()) / lump(stiff()) (row sums)"""
        out = np.zeros(self.sizes()["n_dofs"])
        self._check(self.lib.fdapde_lump(self._ctx, which, _dp(out)))
        return out

    def force(self, ncols=None):
        """force_ (n_dofs x forcing columns, column after column).  ncols defaults to the columns of the last set_forcing of THIS wrapper: fdapde_force
        writes n_dofs * (forcing columns) doubles whatever the caller expects (a shorter buffer is a heap overrun -- tools/fuzz_handle.py found that
        the hard way)"""
        have = getattr(self, "_force_cols", 1)
        if ncols is None:
            ncols = have
        if ncols < have:
            raise ValueError(f"force(): the context holds {have} forcing columns, a buffer for {ncols} was asked for")
        out = np.zeros(self.sizes()["n_dofs"] * ncols)
        self._check(self.lib.fdapde_force(self._ctx, _dp(out)))
        return out

    def solution(self):
        out = np.zeros(self.sizes()["n_dofs"])
        self._check(self.lib.fdapde_solution(self._ctx, _dp(out)))
        return out

    def spmv(self, which, x):
        x = np.ascontiguousarray(x, dtype=float)
        y = np.zeros_like(x)
        self._check(self.lib.fdapde_spmv(self._ctx, which, _dp(x), _dp(y)))
        return y

    def bench_spmv(self, reps=50):
        ms, by = C.c_double(), C.c_double()
        self._check(self.lib.fdapde_bench_spmv(self._ctx, reps, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def solver_layout(self, with_dirichlet=True):
        """(rows, entries) of the interior block as a plain CSR operator + bytes one in-solve SpMV launch streams"""
        ni, nz, by = C.c_int64(), C.c_int64(), C.c_double()
        self._check(self.lib.fdapde_solver_layout(self._ctx, 1 if with_dirichlet else 0, C.byref(ni), C.byref(nz), C.byref(by)))
        return ni.value, nz.value, by.value

    def solver_layout_kind(self, with_dirichlet=True):
        """dict(kind: 0 CSR | 1 blocked ELL | 2 persistent streaming | 3 persistent resident, sym, workgroups, rows_per_thread)"""
        k, sy, g, r = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.fdapde_solver_layout_kind(self._ctx, 1 if with_dirichlet else 0, C.byref(k), C.byref(sy), C.byref(g), C.byref(r)))
        return dict(kind=k.value, sym=sy.value, workgroups=g.value, rows_per_thread=r.value)

    # ---- multi-GPU
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = load().fdapde_comm_unique_id(buf)
        if rc != OK:
            raise FdapdeError(rc, "fdapde_comm_unique_id failed (RCCL not loadable?)")
        return buf.raw

    def comm_init(self, world, rank, unique_id: bytes):
        self._check(self.lib.fdapde_comm_init(self._ctx, int(world), int(rank), C.c_char_p(unique_id)))

    def rowdist_setup(self, dof_key, dof_owner):
        """row-distributed multi-GPU form: global key and owning rank of every local DOF (fdapde_rowdist_setup)"""
        k = np.ascontiguousarray(dof_key, dtype=np.int64)
        o = np.ascontiguousarray(dof_owner, dtype=np.int32)
        self._check(self.lib.fdapde_rowdist_setup(self._ctx, k.ctypes.data_as(C.POINTER(C.c_int64)), _ip(o)))

    def comm_allreduce(self, values, op="sum"):
        """sum / max of a float64 array over the ranks of the communicator (through the library's own RCCL communicator)"""
        a = np.ascontiguousarray(np.asarray(values, dtype=float)).copy()
        self._check(self.lib.fdapde_comm_allreduce(self._ctx, _dp(a), int(a.size), 0 if op == "sum" else 1))
        return a

    def comm_count(self):
        """ranks of the communicator as RCCL reports them (ncclCommCount); the registered world size under the host-staged transport"""
        n = C.c_int32(0)
        self._check(self.lib.fdapde_comm_count(self._ctx, C.byref(n)))
        return int(n.value)

    @staticmethod
    def comm_library():
        return (load().fdapde_comm_library() or b"").decode()

    def comm_init_callback(self, world, rank, allreduce):
        """host-staged transport: allreduce(numpy float64 array) must sum it over all ranks in place"""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)

        def _cb(user, ptr, count):
            try:
                allreduce(np.ctypeslib.as_array(ptr, shape=(count,)))
                return 0
            except Exception:   # noqa: BLE001  (must not propagate through the C frame)
                return 1

        self._ar_cb = proto(_cb)   # keep alive
        self._check(self.lib.fdapde_comm_init_callback(self._ctx, int(world), int(rank), self._ar_cb, None))

    def halo_setup(self, n_if_global, local_dof, if_index, owned):
        local_dof = np.ascontiguousarray(local_dof, dtype=np.int32)
        if_index = np.ascontiguousarray(if_index, dtype=np.int32)
        owned = np.ascontiguousarray(owned, dtype=np.uint8)
        self._check(self.lib.fdapde_halo_setup(self._ctx, C.c_int64(int(n_if_global)), C.c_int64(local_dof.size), _ip(local_dof),
                                               _ip(if_index), _bp(owned)))

    def comm_set_exchange_callback(self, exchange):
        """host-staged neighbour exchange: exchange(peer_rank (int32 array), peer_off (int64 array), send (float64 array), recv (float64
        array, to fill)) must deliver send[peer_off[q]:peer_off[q + 1]] to rank peer_rank[q] and fill the same segment of recv with
        what that rank sent to this one"""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double))

        def _cb(user, n_peers, ranks, offs, send, recv):
            try:
                off = np.ctypeslib.as_array(offs, shape=(n_peers + 1,))
                n = int(off[-1])
                exchange(np.ctypeslib.as_array(ranks, shape=(n_peers,)), off, np.ctypeslib.as_array(send, shape=(n,)),
                         np.ctypeslib.as_array(recv, shape=(n,)))
                return 0
            except Exception:   # noqa: BLE001  (must not propagate through the C frame)
                return 1

        self._xchg_cb = proto(_cb)   # keep alive
        self._check(self.lib.fdapde_comm_set_exchange_callback(self._ctx, self._xchg_cb, None))

    def halo_setup_peers(self, peer_rank, peer_off, peer_dof, owned):
        """neighbour-only exchange (fdapde_halo_setup_peers): peer ranks ascending, their segments in peer_dof (DOF ids shared with the
        peer, in an order both ranks of the pair agree on)"""
        peer_rank = np.ascontiguousarray(peer_rank, dtype=np.int32)
        peer_off = np.ascontiguousarray(peer_off, dtype=np.int64)
        peer_dof = np.ascontiguousarray(peer_dof, dtype=np.int32)
        owned = np.ascontiguousarray(owned, dtype=np.uint8)
        self._check(self.lib.fdapde_halo_setup_peers(self._ctx, C.c_int32(peer_rank.size), _ip(peer_rank),
                                                     peer_off.ctypes.data_as(C.POINTER(C.c_int64)), _ip(peer_dof), _bp(owned)))

    def tune(self, key, value):
        self._check(self.lib.fdapde_tune(self._ctx, key.encode(), int(value)))

    def stream(self):
        return self.lib.fdapde_stream(self._ctx)

    def synchronize(self):
        self._check(self.lib.fdapde_synchronize(self._ctx))
