// capi.hip -- the extern "C" shim of include/fdapde_hip.h: context, device buffers, kernel launches.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include "context.h"
#include "engine.h"
#include "kernels.h"

using namespace fdapde_engine;

namespace fdapde_engine {

// big host-side index arrays of a device-built space, fetched the first time host code needs them (the persistent layout and the
// solver patterns read rowptr_i / colidx_i; the colouring and the partitioned assembly cdofs_i; point location cverts_i / vcoords_i;
// fdapde_pattern_get the reference pattern)
int ensure_host(fdapde_ctx* c, int what) {
    if (!c->dev_built) return FDAPDE_OK;
    HostSpace& hs = c->hs;
    hipStream_t st = c->stream;
    HIPCHK(c, hipSetDevice(c->device));
    if ((what & kHostPattern) && hs.colidx_i.empty()) {
        hs.colidx_i.resize(c->colidx.n);
        HIPCHK(c, hipMemcpyAsync(hs.colidx_i.data(), c->colidx.p, sizeof(int32_t) * c->colidx.n, hipMemcpyDeviceToHost, st));
    }
    if ((what & kHostCells) && hs.cdofs_i.empty()) {
        hs.cdofs_i.resize(c->cdofs.n), hs.cverts_i.resize(c->cverts.n), hs.vcoords_i.resize(c->vcoords.n);
        HIPCHK(c, hipMemcpyAsync(hs.cdofs_i.data(), c->cdofs.p, sizeof(int32_t) * c->cdofs.n, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(hs.cverts_i.data(), c->cverts.p, sizeof(int32_t) * c->cverts.n, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(hs.vcoords_i.data(), c->vcoords.p, sizeof(double) * c->vcoords.n, hipMemcpyDeviceToHost, st));
    }
    if ((what & kHostDofs) && hs.dofs.empty()) {   // DOF table and DOF coordinates in the reference numbering
        hs.dofs.resize((size_t)hs.n_cells * hs.nb), hs.dof_coords.resize((size_t)hs.n_dofs * hs.N);
        if (hs.order == 1) {   // dofs = cells, coordinates = nodes
            std::memcpy(hs.dofs.data(), hs.cells.data(), sizeof(int32_t) * hs.dofs.size());
            std::memcpy(hs.dof_coords.data(), hs.nodes.data(), sizeof(double) * hs.dof_coords.size());
        } else {
            HIPCHK(c, hipMemcpyAsync(hs.dofs.data(), c->dofs_e.p, sizeof(int32_t) * hs.dofs.size(), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(hs.dof_coords.data(), c->coords_e.p, sizeof(double) * hs.dof_coords.size(), hipMemcpyDeviceToHost, st));
        }
    }
    if ((what & kHostRefPattern) && hs.colidx_e.empty()) {
        hs.rowptr_e.resize(c->rowptr_e.n), hs.colidx_e.resize(c->colidx_e.n);
        HIPCHK(c, hipMemcpyAsync(hs.rowptr_e.data(), c->rowptr_e.p, sizeof(int32_t) * c->rowptr_e.n, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(hs.colidx_e.data(), c->colidx_e.p, sizeof(int32_t) * c->colidx_e.n, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return FDAPDE_OK;
}

}   // namespace fdapde_engine

namespace {

// index arrays of the space built on the device (dev_setup.hip): the context's buffers adopt them, nothing is uploaded
int adopt_dev_space(fdapde_ctx* c, DevSpace& s) {
    const HostSpace& hs = c->hs;
    const size_t nd = (size_t)hs.n_dofs, nc = (size_t)hs.n_cells, nnz = (size_t)hs.nnz, nv = (size_t)hs.M + 1, nb = (size_t)hs.nb;
    adopt(c->cverts, s.cverts, nc * nv), adopt(c->cdofs, s.cdofs, nc * nb), adopt(c->vcoords, s.vcoords, (size_t)hs.n_nodes * (hs.N == 2 ? 2 : 4));
    adopt(c->adj, s.adj, (size_t)s.n_adj), adopt(c->slotw, s.slotw, (size_t)s.n_adj * hs.nbw), adopt(c->sl_off, s.sl_off, (size_t)s.n_slices + 1);
    if (s.dealt) adopt(c->lane_row, s.lane_row, (size_t)s.n_blk * kAsmBlock);
    else c->lane_row.release();
    adopt(c->bc_off, s.bc_off, (size_t)s.n_blk + 1), adopt(c->bn_off, s.bn_off, (size_t)s.n_blk + 1);
    adopt(c->bc_cell, s.bc_cell, (size_t)s.n_bc), adopt(c->bn_node, s.bn_node, (size_t)s.n_bn), adopt(c->bc_vert, s.bc_vert, (size_t)s.n_bc * 4);
    adopt(c->rowptr, s.rowptr, nd + 1), adopt(c->colidx, s.colidx, nnz + 2), adopt(c->diag, s.diag, nd), adopt(c->slot_i2e, s.slot_i2e, nnz);
    adopt(c->dof_i2e, s.dof_i2e, nd), adopt(c->dof_e2i, s.dof_e2i, nd), adopt(c->cell_i2e, s.cell_i2e, nc), adopt(c->bnd, s.bnd, nd);
    adopt(c->rowptr_e, s.rowptr_e, nd + 1), adopt(c->colidx_e, s.colidx_e, nnz);
    HIPCHK(c, c->rb_row.upload(hs.rb_row.data(), hs.rb_row.size(), c->stream));
    dev_space_release(&s);   // what nobody adopted (node_i2e)
    c->dev_built = true;
    return FDAPDE_OK;
}

// FDAPDE_SETUP_CHECK: the device-built space against the host builder's, array for array
int check_dev_space(fdapde_ctx* c, const DevSpace& s, int order) {
    HostSpace ref;
    const HostSpace& hs = c->hs;
    ref.M = hs.M, ref.N = hs.N, ref.n_nodes = hs.n_nodes, ref.n_cells = hs.n_cells, ref.nodes = hs.nodes, ref.cells = hs.cells, ref.node_bnd = hs.node_bnd;
    std::string err;
    if (int rc = host_build_space(ref, order, err)) return fail(c, rc, "set-up check: the host builder failed");
    int bad = 0;
    auto cmp = [&](const char* name, const void* dev, const void* host, size_t bytes, size_t elem) {
        std::vector<unsigned char> tmp(bytes ? bytes : 1);
        if (bytes && hipMemcpy(tmp.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            std::fprintf(stderr, "set-up check %-10s: download failed\n", name), ++bad;
            return;
        }
        size_t at = 0;
        while (at < bytes && tmp[at] == static_cast<const unsigned char*>(host)[at]) ++at;
        if (at < bytes) std::fprintf(stderr, "set-up check %-10s: MISMATCH at element %zu of %zu\n", name, at / elem, bytes / elem), ++bad;
        else std::fprintf(stderr, "set-up check %-10s: ok (%zu elements)\n", name, bytes / elem);
    };
    auto scalar = [&](const char* name, int64_t dev, int64_t host) {
        if (dev != host) std::fprintf(stderr, "set-up check %-10s: MISMATCH %lld vs %lld\n", name, (long long)dev, (long long)host), ++bad;
    };
    scalar("nnz", hs.nnz, ref.nnz), scalar("max_row", hs.max_row, ref.max_row), scalar("blk_nnz", hs.max_blk_nnz, ref.max_blk_nnz);
    scalar("blk_cells", hs.max_blk_cells, ref.max_blk_cells), scalar("blk_nodes", hs.max_blk_nodes, ref.max_blk_nodes);
    scalar("n_adj", s.n_adj, (int64_t)ref.adj.size()), scalar("n_bc", s.n_bc, (int64_t)ref.bc_cell.size()), scalar("n_bn", s.n_bn, (int64_t)ref.bn_node.size());
    scalar("dealt", s.dealt, !ref.lane_row.empty());
    scalar("n_edges", hs.n_edges, ref.n_edges), scalar("n_dofs", hs.n_dofs, ref.n_dofs);
    if (bad == 0 && hs.dof_bnd != ref.dof_bnd) std::fprintf(stderr, "set-up check dof_bnd: MISMATCH\n"), ++bad;
    if (bad == 0) {
#define CMP(name, dptr, hvec_) cmp(name, dptr, (hvec_).data(), (hvec_).size() * sizeof((hvec_)[0]), sizeof((hvec_)[0]))
        CMP("dof_i2e", s.dof_i2e, ref.dof_i2e), CMP("dof_e2i", s.dof_e2i, ref.dof_e2i), CMP("cell_i2e", s.cell_i2e, ref.cell_i2e);
        CMP("node_i2e", s.node_i2e, ref.node_i2e), CMP("vcoords", s.vcoords, ref.vcoords_i), CMP("bnd", s.bnd, ref.dof_bnd_i);
        CMP("cverts", s.cverts, ref.cverts_i), CMP("cdofs", s.cdofs, ref.cdofs_i), CMP("rowptr", s.rowptr, ref.rowptr_i);
        CMP("colidx", s.colidx, ref.colidx_i), CMP("diag", s.diag, ref.diag_i), CMP("rowptr_e", s.rowptr_e, ref.rowptr_e);
        CMP("colidx_e", s.colidx_e, ref.colidx_e), CMP("slot_i2e", s.slot_i2e, ref.slot_i2e), CMP("sl_off", s.sl_off, ref.sl_off);
        if (s.dealt) CMP("lane_row", s.lane_row, ref.lane_row);
        if (order == 2) CMP("dofs", c->dofs_e.p, ref.dofs), CMP("dof_coords", c->coords_e.p, ref.dof_coords);
        CMP("bc_off", s.bc_off, ref.bc_off), CMP("bn_off", s.bn_off, ref.bn_off), CMP("bc_cell", s.bc_cell, ref.bc_cell);
        CMP("bn_node", s.bn_node, ref.bn_node), CMP("bc_vert", s.bc_vert, ref.bc_vert), CMP("adj", s.adj, ref.adj), CMP("slotw", s.slotw, ref.slotw);
#undef CMP
        if (hs.rb_row != ref.rb_row) std::fprintf(stderr, "set-up check rb_row: MISMATCH\n"), ++bad;
    }
    if (bad) return fail(c, FDAPDE_EHIP, "FDAPDE_SETUP_CHECK: the device-built space differs from the host builder's (see stderr)");
    return FDAPDE_OK;
}

int upload_space(fdapde_ctx* c) {
    HostSpace& hs = c->hs;
    hipStream_t st = c->stream;
    if (!c->dev_built) {
    HIPCHK(c, c->cverts.upload(hs.cverts_i.data(), hs.cverts_i.size(), st));
    HIPCHK(c, c->cdofs.upload(hs.cdofs_i.data(), hs.cdofs_i.size(), st));
    HIPCHK(c, c->vcoords.upload(hs.vcoords_i.data(), hs.vcoords_i.size(), st));
    HIPCHK(c, c->adj.upload(hs.adj.data(), hs.adj.size(), st));
    HIPCHK(c, c->slotw.upload(hs.slotw.data(), hs.slotw.size(), st));
    HIPCHK(c, c->sl_off.upload(hs.sl_off.data(), hs.sl_off.size(), st));
    if (!hs.lane_row.empty()) HIPCHK(c, c->lane_row.upload(hs.lane_row.data(), hs.lane_row.size(), st));
    else c->lane_row.release();
    HIPCHK(c, c->bc_off.upload(hs.bc_off.data(), hs.bc_off.size(), st));
    HIPCHK(c, c->bn_off.upload(hs.bn_off.data(), hs.bn_off.size(), st));
    HIPCHK(c, c->bc_cell.upload(hs.bc_cell.data(), hs.bc_cell.size(), st));
    HIPCHK(c, c->bn_node.upload(hs.bn_node.data(), hs.bn_node.size(), st));
    HIPCHK(c, c->bc_vert.upload(hs.bc_vert.data(), hs.bc_vert.size(), st));
    HIPCHK(c, c->rowptr.upload(hs.rowptr_i.data(), hs.rowptr_i.size(), st));
    HIPCHK(c, c->colidx.upload(hs.colidx_i.data(), hs.colidx_i.size(), st));   // nnz + 2 padding entries
    HIPCHK(c, c->diag.upload(hs.diag_i.data(), hs.diag_i.size(), st));
    HIPCHK(c, c->slot_i2e.upload(hs.slot_i2e.data(), hs.slot_i2e.size(), st));
    HIPCHK(c, c->dof_i2e.upload(hs.dof_i2e.data(), hs.dof_i2e.size(), st));
    HIPCHK(c, c->dof_e2i.upload(hs.dof_e2i.data(), hs.dof_e2i.size(), st));
    HIPCHK(c, c->cell_i2e.upload(hs.cell_i2e.data(), hs.cell_i2e.size(), st));
    HIPCHK(c, c->rb_row.upload(hs.rb_row.data(), hs.rb_row.size(), st));
    HIPCHK(c, c->bnd.upload(hs.dof_bnd_i.data(), hs.dof_bnd_i.size(), st));
    }
    DevTables dt{};
    std::memcpy(dt.qw, c->tb.qw, sizeof dt.qw);
    std::memcpy(dt.psi, c->tb.psi, sizeof dt.psi);
    std::memcpy(dt.dpsi, c->tb.dpsi, sizeof dt.dpsi);
    std::memcpy(dt.qn, c->tb.qn, sizeof dt.qn);
    dt.wsum = 0;
    for (int q = 0; q < c->tb.nq; ++q) dt.wsum += c->tb.qw[q];
    for (int i = 0; i < c->tb.nb; ++i)
        for (int j = 0; j < c->tb.nb; ++j) {
            double m = 0;
            const int lo = i < j ? i : j, hi = i < j ? j : i;   // same expression for (i,j) and (j,i): bitwise symmetric mass
            for (int q = 0; q < c->tb.nq; ++q) m += c->tb.qw[q] * (c->tb.psi[lo * c->tb.nq + q] * c->tb.psi[hi * c->tb.nq + q]);
            dt.mtab[i * c->tb.nb + j] = m;
        }
    HIPCHK(c, c->tables.upload(&dt, 1, st));
    {   // reference tensors of the constant-coefficient form (element_row OPK 3), same quadrature nodes and weights
        auto rt_own = std::make_unique<DevRefTensors>();   // ~10 KB: off the stack, and per call (contexts of different threads build concurrently)
        DevRefTensors& rt = *rt_own;
        std::memset(&rt, 0, sizeof rt);
        const int nb = c->tb.nb, nq = c->tb.nq, nn = nb * nb;
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < nb; ++i)
                for (int j = 0; j < nb; ++j) {
                    for (int l = 0; l < 3; ++l) {
                        double v = 0;
                        for (int q = 0; q < nq; ++q) v += c->tb.qw[q] * (c->tb.dpsi[(i * nq + q) * 3 + k] * c->tb.dpsi[(j * nq + q) * 3 + l]);
                        rt.ktab[(k * 3 + l) * nn + i * nb + j] = v;
                    }
                    double v = 0;
                    for (int q = 0; q < nq; ++q) v += c->tb.qw[q] * (c->tb.psi[i * nq + q] * c->tb.dpsi[(j * nq + q) * 3 + k]);
                    rt.ctab[k * nn + i * nb + j] = v;
                }
        HIPCHK(c, c->reftab.upload(&rt, 1, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    const size_t n = (size_t)hs.n_dofs, nnz = (size_t)hs.nnz;
    HIPCHK(c, c->vals[0].alloc(nnz + 2));   // + 2: pair loads of the SpMV may touch one entry past a row's end
    HIPCHK(c, c->vals[1].alloc(nnz + 2));
    HIPCHK(c, c->sval.alloc(nnz + 2));
    HIPCHK(c, c->tmp_v.alloc(nnz));
    for (DBuf<double>* b : {&c->scale, &c->gt, &c->x, &c->r, &c->p, &c->y, &c->s, &c->t, &c->r0, &c->u, &c->tmp_e, &c->tmp_i, &c->g})
        HIPCHK(c, b->alloc(n));
    HIPCHK(c, c->force.alloc(n));
    c->n_rb = (int)hs.rb_row.size() - 1;
    c->rb_per_band = (c->n_rb + 7) / 8;
    // workgroups per band: 192 (1536 workgroups = 1.5 rounds of the 1024 resident ones) measured best on C3 with the default cache
    // policy (solve 33.0 ms at 256, 32.6 at 192, 33.5 at 160 / 224); smaller matrices get one workgroup per 4096 nonzeros
    int bpx = c->rb_per_band < 192 ? c->rb_per_band : 192;
    if (bpx < 1) bpx = 1;
    {
        const char* v = std::getenv("FDAPDE_SPMV");
        c->spmv_variant = (v && std::strcmp(v, "stream") == 0) ? 1 : ((v && std::strcmp(v, "team") == 0) ? 0 : 2);
        const double mean_row = (double)hs.nnz / (double)(hs.n_dofs > 0 ? hs.n_dofs : 1);
        int t = 4;
        while (t < 64 && t < mean_row) t *= 2;
        c->spmv_team = t;
        if (c->spmv_variant == 2) c->spmv_team = t / 2 < 2 ? 2 : (t / 2 > 32 ? 32 : t / 2);
        if (const char* e = std::getenv("FDAPDE_SPMV_TEAM")) c->spmv_team = std::atoi(e);
        if (const char* e = std::getenv("FDAPDE_SPMV_ABLATE")) c->spmv_ablate = std::atoi(e);
        if (c->spmv_variant == 2) {
            if (const char* e = std::getenv("FDAPDE_SPMV_UNROLL")) c->spmv_unroll = std::atoi(e);
            const int tt = c->spmv_team, u = tt == 2 ? 1 : (tt == 4 ? 2 : (tt == 8 ? c->spmv_unroll : 4));
            const int wrows = (64 / tt) * u * 4;
            const int64_t tiles = ((hs.n_dofs + 7) / 8 + wrows - 1) / wrows;
            bpx = (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
            if (const char* e = std::getenv("FDAPDE_SPMV_BPX")) bpx = std::atoi(e);
        }
        if (c->spmv_variant == 0) {
            if (const char* e = std::getenv("FDAPDE_SPMV_UNROLL")) c->spmv_unroll = std::atoi(e);
            const int u = c->spmv_team == 64 || c->spmv_team == 4 ? 2 : (c->spmv_team == 16 ? c->spmv_unroll : 4);
            const int wrows = (64 / c->spmv_team) * u * 4;   // rows per workgroup-iteration
            const int64_t tiles = ((hs.n_dofs + 7) / 8 + wrows - 1) / wrows;
            bpx = (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
            if (const char* e = std::getenv("FDAPDE_SPMV_BPX")) bpx = std::atoi(e);
        }
    }
    c->spmv_grid = 8 * bpx;
    int64_t vg = (hs.n_dofs + 255) / 256;
    c->vec_grid = (int)(vg < 1024 ? (vg < 1 ? 1 : vg) : 1024);
    HIPCHK(c, c->part_a.alloc(2 * (size_t)c->spmv_grid));
    {
        const int64_t n2 = hs.n_dofs / 2, per = 256 * kCgV;
        c->cg_grid = (int)((n2 + per - 1) / per);
        if (c->cg_grid < 1) c->cg_grid = 1;
    }
    HIPCHK(c, c->part_b.alloc(8 * (size_t)(c->vec_grid > c->cg_grid ? c->vec_grid : c->cg_grid) + 16));   // two halves at every k_cgf_update width
    HIPCHK(c, c->sc.alloc(24));
    HIPCHK(c, c->ctl.alloc(4));
    HIPCHK(c, hipMemsetAsync(c->ctl.p, 0, 4 * sizeof(int32_t), st));
    HIPCHK(c, hipMemsetAsync(c->force.p, 0, n * sizeof(double), st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->dev_ready = true;
    return FDAPDE_OK;
}

AsmArgs asm_args(fdapde_ctx* c) {
    AsmArgs a{};
    a.n_dofs = c->hs.n_dofs, a.n_cells = c->hs.n_cells;
    a.cverts = c->cverts.p, a.cdofs = c->cdofs.p, a.vcoords = c->vcoords.p;
    a.sl_off = c->sl_off.p, a.adj = c->adj.p, a.slotw = c->slotw.p, a.lane_row = c->lane_row.p;   // nullptr = identity
    a.rowptr = c->rowptr.p, a.colidx = c->colidx.p, a.tables = c->tables.p, a.reftab = c->reftab.p;
    a.bc_off = c->bc_off.p, a.bc_cell = c->bc_cell.p, a.bc_vert = c->bc_vert.p, a.bn_off = c->bn_off.p, a.bn_node = c->bn_node.p;
    a.lds_nodes = c->hs.max_blk_nodes;
    return a;
}

// validate an operator expression and stage its (permuted) coefficient data on the device
// reuse: the coefficient buffers already hold THIS operator's data (set by the previous fdapde_init): no upload
int make_dev_op(fdapde_ctx* c, const std::vector<HostTerm>& terms, DevOp* out, int coef_slot0, bool reuse = false) {
    DevOp op{};
    op.n = (int32_t)terms.size();
    op.needs_psi = 0, op.needs_rows = 0;
    for (size_t k = 0; k < terms.size(); ++k) {
        const fdapde_term& t = terms[k].t;
        DevTerm& d = op.t[k];
        d.kind = t.kind, d.space_varying = t.space_varying, d.coef = t.coef, d.data = nullptr;
        std::memcpy(d.cst, t.cst, sizeof d.cst);
        if (t.kind == FDAPDE_ADVECTION || t.kind == FDAPDE_REACTION) op.needs_psi = 1;
        if (t.space_varying) op.needs_rows = 1;
        if (t.space_varying) {
            DBuf<double>& buf = c->coef[coef_slot0 + k];
            if (!(reuse && buf.p && buf.n >= terms[k].data_i.size()))
                HIPCHK(c, buf.upload(terms[k].data_i.data(), terms[k].data_i.size(), c->stream));
            d.data = buf.p;
        }
    }
    // constant-coefficient summary (element_row OPK 3)
    const int N = c->hs.N;
    bool adv = false;
    for (size_t k = 0; k < terms.size(); ++k) {
        const fdapde_term& t = terms[k].t;
        if (t.kind == FDAPDE_LAPLACIAN)
            for (int r = 0; r < N; ++r) op.kt[r * N + r] += t.coef;
        else if (t.kind == FDAPDE_DIFFUSION)
            for (int e = 0; e < N * N; ++e) op.kt[e] += t.coef * t.cst[e];
        else if (t.kind == FDAPDE_ADVECTION) {
            adv = true;
            for (int e = 0; e < N; ++e) op.bt[e] += t.coef * t.cst[e];
        } else if (t.kind == FDAPDE_REACTION)
            op.ct += t.coef * t.cst[0];
    }
    bool ksym = true;
    for (int r = 0; r < N; ++r)
        for (int q = 0; q < r; ++q) ksym = ksym && op.kt[r * N + q] == op.kt[q * N + r];
    op.tab_sym = (ksym && !adv) ? 1 : 0;
    *out = op;
    return FDAPDE_OK;
}

int check_terms(fdapde_ctx* c, int32_t n_terms, const fdapde_term* terms, std::vector<HostTerm>* out, bool* symmetric) {
    if (n_terms < 1 || n_terms > kMaxTerms || !terms) return fail(c, FDAPDE_EINVAL, "operator needs 1..8 leaves");
    if (!c->space_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build before setting an operator");
    const HostSpace& hs = c->hs;
    out->clear();
    *symmetric = true;
    const int64_t rows = (int64_t)hs.nq * hs.n_cells;
    for (int k = 0; k < n_terms; ++k) {
        HostTerm h;
        h.t = terms[k];
        int width = 0;
        switch (terms[k].kind) {
        case FDAPDE_LAPLACIAN: break;
        case FDAPDE_DT: break;
        case FDAPDE_DIFFUSION: width = hs.N * hs.N; break;
        case FDAPDE_ADVECTION: width = hs.N, *symmetric = false; break;   // advection.h:45 is_symmetric = false
        case FDAPDE_REACTION: width = 1; break;
        default: return fail(c, FDAPDE_EINVAL, "unknown operator kind");
        }
        if (terms[k].space_varying) {
            if (width == 0 || !terms[k].data) return fail(c, FDAPDE_EINVAL, "space-varying leaf without data");
            h.data_i.resize((size_t)rows * width);
            for (int64_t ci = 0; ci < hs.n_cells; ++ci) {
                const int64_t ce = hs.cell_i2e[(size_t)ci];
                std::memcpy(&h.data_i[(size_t)ci * hs.nq * width], &terms[k].data[(size_t)ce * hs.nq * width],
                            sizeof(double) * hs.nq * width);
            }
            h.t.data = nullptr;
        }
        out->push_back(std::move(h));
    }
    return FDAPDE_OK;
}

template <int M, int R>
int launch_assembly_t(fdapde_ctx* c, AsmArgs a, const DevOp& op, int assembly) {
    const HostSpace& hs = c->hs;
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    if (assembly == FDAPDE_ASSEMBLY_ROWS) {
        // specialised integrands (see element_row): the two operators FEMSolverBase::init always assembles, and any other
        // constant-coefficient expression through the reference tensors
        int opk = 0;
        if (!op.needs_rows) opk = 3;
        if (op.n == 1 && op.t[0].kind == FDAPDE_LAPLACIAN) opk = 1;
        if (op.n == 1 && op.t[0].kind == FDAPDE_REACTION && !op.t[0].space_varying) opk = 2;
        if (std::getenv("FDAPDE_ASM_GENERIC")) opk = 0;
        const size_t tab = sizeof(DevTables) + (opk == 3 ? sizeof(DevRefTensors) : 0) +
                           (size_t)hs.max_blk_nodes * (M == 2 ? 2 : 3) * sizeof(double);
        size_t acc = (size_t)hs.max_blk_nnz * sizeof(double);
        if (tab + acc > (size_t)c->lds_limit) acc = tab < (size_t)c->lds_limit ? (size_t)c->lds_limit - tab : 0;
        a.lds_acc_cap = (int32_t)(acc / sizeof(double));
        if (a.fq != nullptr && a.fq == c->fq.p && c->fq_blk_ready) a.fq = c->fq_blk.p, a.fq_block = 1;   // column 0: one load coefficient per visit slot
        else if (a.fq != nullptr && a.fq == c->fq.p && c->fq_bc_ready) a.fq = c->fq_bc.p, a.fq_block = 2;   // column 0: samples in block-cell order
        const int grid = 8 * (int)(((hs.n_dofs + kAsmBlock - 1) / kAsmBlock + 7) / 8);   // 8 XCD bands of blocks (k_assemble_rows)
        size_t lds = tab + acc;
        if (lds > 64 * 1024)
            for (const void* fn : {reinterpret_cast<const void*>(&k_assemble_rows<M, R, 0>),
                                   reinterpret_cast<const void*>(&k_assemble_rows<M, R, 1>),
                                   reinterpret_cast<const void*>(&k_assemble_rows<M, R, 2>),
                                   reinterpret_cast<const void*>(&k_assemble_rows<M, R, 3>)})
                (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (opk == 3)
            hipLaunchKernelGGL((k_assemble_rows<M, R, 3>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
        else if (opk == 1)
            hipLaunchKernelGGL((k_assemble_rows<M, R, 1>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
        else if (opk == 2)
            hipLaunchKernelGGL((k_assemble_rows<M, R, 2>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
        else
            hipLaunchKernelGGL((k_assemble_rows<M, R, 0>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
    } else {
        if (a.vals) HIPCHK(c, hipMemsetAsync(a.vals, 0, sizeof(double) * (size_t)hs.nnz, c->stream));
        if (a.force) HIPCHK(c, hipMemsetAsync(a.force, 0, sizeof(double) * (size_t)hs.n_dofs, c->stream));
        if (assembly == FDAPDE_ASSEMBLY_PARTITIONED) {
            if (!c->part_ready) {   // partitions of 2048 cells, local colours, shared-row flags, slot map (host index work, once)
                if (int rc = ensure_host(c, kHostPattern | kHostCells)) return rc;
                CellPartitions cp;
                int cells = 2048;   // measured on C3: 2048 cells 5.4 ms (58 % of the rows shared -> atomics); see tools/asm_ab.py for larger ones
                if (const char* e = std::getenv("FDAPDE_PART_CELLS")) cells = std::atoi(e);
                if (int rc = host_build_cell_partitions(hs, cells, cp, c->err)) return rc;
                HIPCHK(c, c->part_cells.upload(cp.cell_list.data(), cp.cell_list.size(), c->stream));
                HIPCHK(c, c->part_off.upload(cp.colour_off.data(), cp.colour_off.size(), c->stream));
                HIPCHK(c, c->part_slots.upload(cp.slot_map.data(), cp.slot_map.size(), c->stream));
                HIPCHK(c, c->part_shared.upload(cp.dof_shared.data(), cp.dof_shared.size(), c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                c->part_colours = cp.max_colours, c->n_parts = cp.n_parts, c->part_ready = true;
            }
            int opk = 0;
            if (!op.needs_rows) opk = 3;
            if (op.n == 1 && op.t[0].kind == FDAPDE_LAPLACIAN) opk = 1;
            if (op.n == 1 && op.t[0].kind == FDAPDE_REACTION && !op.t[0].space_varying) opk = 2;
            const size_t lds = sizeof(DevTables) + (opk == 3 ? sizeof(DevRefTensors) : 0);
#define PART_GO(K_)                                                                                                            \
    hipLaunchKernelGGL((k_assemble_part<M, R, K_>), dim3((unsigned)c->n_parts), dim3(256), lds, c->stream, a, op, c->part_cells.p, \
                       c->part_off.p, c->part_colours, c->part_shared.p, c->part_slots.p)
            if (opk == 3) PART_GO(3);
            else if (opk == 1) PART_GO(1);
            else if (opk == 2) PART_GO(2);
            else PART_GO(0);
#undef PART_GO
        } else if (assembly == FDAPDE_ASSEMBLY_WAVE) {
            if constexpr (R != 1) {
                return fail(c, FDAPDE_EUNSUPPORTED, "the wavefront-per-element assembly exists for P1 only");
            } else {
                if (!c->colour_ready) {
                    if (int rc = ensure_host(c, kHostCells)) return rc;
                    int rc = host_build_colouring(c->hs, c->err);
                    if (rc) return rc;
                    HIPCHK(c, c->colour_cells.upload(hs.colour_cells.data(), hs.colour_cells.size(), c->stream));
                    c->colour_ready = true;
                }
                if (!c->wave_ready) {
                    if (int rc = ensure_host(c, kHostPattern | kHostCells)) return rc;
                    std::vector<int32_t> sm;
                    host_build_slot_map(hs, hs.colour_cells.data(), hs.n_cells, sm);
                    HIPCHK(c, c->wave_slots.upload(sm.data(), sm.size(), c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    c->wave_ready = true;
                }
                for (int k = 0; k < hs.n_colours; ++k) {
                    const int64_t o0 = hs.colour_off[(size_t)k], cnt = hs.colour_off[(size_t)k + 1] - o0;
                    if (cnt == 0) continue;
                    hipLaunchKernelGGL((k_assemble_wave<M>), dim3((unsigned)((cnt + 3) / 4)), dim3(256), sizeof(DevTables), c->stream, a, op,
                                       c->colour_cells.p + o0, c->wave_slots.p + (size_t)o0 * NB * NB, cnt);
                }
            }
        } else if (assembly == FDAPDE_ASSEMBLY_ATOMIC) {
            const int64_t work = hs.n_cells * NB;
            hipLaunchKernelGGL((k_assemble_scatter<M, R, true>), dim3((unsigned)((work + 255) / 256)), dim3(256),
                               sizeof(DevTables), c->stream, a, op, (const int32_t*)nullptr, hs.n_cells);
        } else {
            if (!c->colour_ready) {
                if (int rc = ensure_host(c, kHostCells)) return rc;
                int rc = host_build_colouring(c->hs, c->err);
                if (rc) return rc;
                HIPCHK(c, c->colour_cells.upload(hs.colour_cells.data(), hs.colour_cells.size(), c->stream));
                c->colour_ready = true;
            }
            for (int k = 0; k < hs.n_colours; ++k) {
                const int64_t cnt = hs.colour_off[(size_t)k + 1] - hs.colour_off[(size_t)k];
                if (cnt == 0) continue;
                const int64_t work = cnt * NB;
                hipLaunchKernelGGL((k_assemble_scatter<M, R, false>), dim3((unsigned)((work + 255) / 256)), dim3(256),
                                   sizeof(DevTables), c->stream, a, op, c->colour_cells.p + hs.colour_off[(size_t)k], cnt);
            }
        }
    }
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}

int launch_assembly(fdapde_ctx* c, const AsmArgs& a, const DevOp& op, int assembly) {
    const int M = c->hs.M, R = c->hs.order;
    if (assembly < 0 || assembly > 4) return fail(c, FDAPDE_EINVAL, "unknown assembly variant");
    if (M == 2 && R == 1) return launch_assembly_t<2, 1>(c, a, op, assembly);
    if (M == 2 && R == 2) return launch_assembly_t<2, 2>(c, a, op, assembly);
    if (M == 3 && R == 1) return launch_assembly_t<3, 1>(c, a, op, assembly);
    if (M == 3 && R == 2) return launch_assembly_t<3, 2>(c, a, op, assembly);
    return fail(c, FDAPDE_EUNSUPPORTED, "unsupported (M, order)");
}

// the captured CG chunk bakes pointers and sizes in: drop it whenever a layout, buffer or knob may have changed
inline void drop_graph(fdapde_ctx* c) {
    if (c->cg_graph_exec) (void)hipGraphExecDestroy(c->cg_graph_exec);
    c->cg_graph_exec = nullptr;
}

// e0 / e1 (optional): HIP events attached to the dispatch itself (hipExtLaunchKernelGGL), i.e. the kernel's own begin / end
// timestamps on the stream it runs on -- the same interval rocprofv3 --kernel-trace reports, with no extra marker packet
// between the neighbouring kernels.
void launch_spmv(fdapde_ctx* c, const double* vals, const double* x, double* y, const double* w, double* partial,
                 const int32_t* stop, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, int dot2_ww = 0,
                 const uint8_t* owned = nullptr) {
    if (vals == c->sval.p && c->bk_cur >= 0 && owned == nullptr) {   // the solver's scaled matrix in blocked-ELL form (k_spmv_blocked)
        const fdapde_ctx::Blocked& bk = c->bk[c->bk_cur];
        BlockedSpmvArgs a{};
        a.G = bk.meta.G, a.nsl = bk.meta.nsl, a.imp_cap = bk.imp_cap, a.dot2_ww = dot2_ww;
        a.slot_dof = bk.slot_dof.p, a.ell_off = bk.ell_off.p, a.sl_off = bk.sl_off.p, a.ell_code = bk.ell_code.p, a.ell_val = bk.ell_val.p;
        a.imp_off = bk.imp_off.p, a.imp_dof = bk.imp_dof.p, a.drop_dof = bk.drop_dof.p, a.n_drop = (int32_t)bk.meta.n_drop, a.x = x, a.y = y, a.w = partial ? (w ? w : x) : nullptr, a.partial = partial, a.stop = stop;
#define BLOCKED_GO(R_)                                                                                                          \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmv_blocked<R_>), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                  (int)bk.lds_bytes);                                                                           \
        if (e0 || e1) hipExtLaunchKernelGGL((k_spmv_blocked<R_>), dim3(a.G), dim3(kPersistT), bk.lds_bytes, c->stream, e0, e1, 0, a); \
        else hipLaunchKernelGGL((k_spmv_blocked<R_>), dim3(a.G), dim3(kPersistT), bk.lds_bytes, c->stream, a);                  \
    } while (0)
        switch (bk.meta.R) {
        case 2: BLOCKED_GO(2); break;
        case 4: BLOCKED_GO(4); break;
        case 8: BLOCKED_GO(8); break;
        default: BLOCKED_GO(16); break;
        }
#undef BLOCKED_GO
        return;
    }
    SpmvArgs s{};
    s.rowptr = c->rowptr.p, s.colidx = c->colidx.p, s.vals = vals, s.x = x, s.y = y;
    s.rb_row = c->rb_row.p, s.n_rb = c->n_rb, s.rb_per_band = c->rb_per_band, s.nnz = (int32_t)c->hs.nnz;
    s.w = w, s.partial = partial, s.stop = stop, s.dot2_ww = dot2_ww, s.owned = owned, s.unit_diag = 0;
    s.n_cols = (int32_t)c->hs.n_dofs;
    // value-stream policy by size: x and y slices of a row band (16 bytes per row, 8 bands) against the 4 MB L2 of an XCD
    const bool ntv = c->spmv_ntv < 0 ? c->hs.n_dofs > kNtValsRows : c->spmv_ntv != 0;
    int64_t n = c->hs.n_dofs;   // rows of the CSR arrays the kernel walks (virtual rows for a segmented pattern)
    bool vrows = false;
    if (vals == c->sval.p && c->sp_cur >= 0) {   // the solver's scaled matrix lives in the compact pattern
        s.rowptr = c->sp_rowptr[c->sp_cur].p, s.colidx = c->sp_colidx[c->sp_cur].p, s.nnz = (int32_t)c->sp_nnz[c->sp_cur];
        if (c->spmv_c16) s.col16 = c->sp_col16[c->sp_cur].p, s.tbase = c->sp_tbase[c->sp_cur].p;
        if (c->sp_nv[c->sp_cur] > 0) {   // segmented: only the VROWS instantiations understand it (always with column codes)
            vrows = true, n = c->sp_nv[c->sp_cur], s.vrow = c->sp_vrow[c->sp_cur].p;
            s.col16 = c->sp_col16[c->sp_cur].p, s.tbase = c->sp_tbase[c->sp_cur].p;
        }
        s.unit_diag = 1;
        // multi-GPU: the local diagonals s_i^2 (A_p)_ii of an interface DOF sum to 1 over the ranks sharing it; the implicit
        // unit diagonal is therefore contributed by the DOF's owner only (any split of the entries among ranks is valid)
        if ((c->comm != nullptr || c->ar_fn != nullptr) && c->halo_ready) s.owned = c->owned.p;
    }
    // eight row bands (one per XCD); band starts on a multiple of 32 rows so that a wavefront tile lies in one code group
    const int64_t rpb = (((n + 7) / 8) + 31) & ~int64_t(31);
    const dim3 grid(c->spmv_grid), block(256);
    // dispatch-attached events only where a launch is timed; the plain launch can be captured into a hipGraph
#define SPMV_GO(...)                                                                                 \
    do {                                                                                             \
        if (e0 || e1) hipExtLaunchKernelGGL((__VA_ARGS__), grid, block, 0, c->stream, e0, e1, 0, s, n, rpb); \
        else hipLaunchKernelGGL((__VA_ARGS__), grid, block, 0, c->stream, s, n, rpb);                \
    } while (0)
    if (c->spmv_variant == 1) {
        if (e0 || e1) hipExtLaunchKernelGGL(k_spmv, grid, block, 0, c->stream, e0, e1, 0, s);
        else hipLaunchKernelGGL(k_spmv, grid, block, 0, c->stream, s);
        return;
    }
    if (c->spmv_variant == 2) {   // two entries per lane: team = lanes per row, covering 2 * team entries per pass
        // production forms: 16-byte aligned entry pairs (2048), + 16-bit column codes when the pattern has them (4096),
        // + unconditional ownership loads when the implicit diagonal is owner-masked (multi-GPU, 8192)
        const bool c16 = s.col16 != nullptr, dist = s.unit_diag && s.owned != nullptr;
        const bool wx = s.w == nullptr || s.w == s.x;   // dot operand == x (CG: p.Ap): one row load serves both (16384)
#define SPMV_PROD_FEW(T_, U_)                                                    \
    do {                                                                         \
        if (c16 && dist) SPMV_GO(k_spmv_team2<T_, U_, 2048 | 4096 | 8192>);      \
        else if (c16) SPMV_GO(k_spmv_team2<T_, U_, 2048 | 4096>);                \
        else if (dist) SPMV_GO(k_spmv_team2<T_, U_, 2048 | 8192>);               \
        else SPMV_GO(k_spmv_team2<T_, U_, 2048>);                                \
    } while (0)
#define SPMV_PROD(T_, U_)                                                                    \
    do {                                                                                     \
        if (!wx) SPMV_PROD_FEW(T_, U_);                                                      \
        else if (c16 && dist) SPMV_GO(k_spmv_team2<T_, U_, 2048 | 4096 | 8192 | 16384>);     \
        else if (c16) SPMV_GO(k_spmv_team2<T_, U_, 2048 | 4096 | 16384>);                    \
        else if (dist) SPMV_GO(k_spmv_team2<T_, U_, 2048 | 8192 | 16384>);                   \
        else SPMV_GO(k_spmv_team2<T_, U_, 2048 | 16384>);                                    \
    } while (0)
        if (vrows) {   // built for this team size (build_solver_pattern); T = 8 or 16
#define SPMV_VROWS(T_)                                                                                  \
    do {                                                                                                \
        if (dist && wx) SPMV_GO(k_spmv_team2<T_, 4, 2048 | 4096 | 131072 | 8192 | 16384>);              \
        else if (dist) SPMV_GO(k_spmv_team2<T_, 4, 2048 | 4096 | 131072 | 8192>);                       \
        else if (wx) SPMV_GO(k_spmv_team2<T_, 4, 2048 | 4096 | 131072 | 16384>);                        \
        else SPMV_GO(k_spmv_team2<T_, 4, 2048 | 4096 | 131072>);                                        \
    } while (0)
            if (c->sp_team == 8) SPMV_VROWS(8);
            else SPMV_VROWS(16);
#undef SPMV_VROWS
            return;
        }
        switch (c->spmv_team) {
        case 2: SPMV_PROD_FEW(2, 1); break;
        case 4: SPMV_PROD(4, 2); break;
        case 8:
            // the diagnostic forms >= 100 exist for the compact coded matrix only: any other product takes the production path
            switch ((c->spmv_ablate >= 100 && !c16) ? 0 : c->spmv_ablate) {
            case 1: SPMV_GO(k_spmv_team2<8, 4, 1>); break;
            case 2: SPMV_GO(k_spmv_team2<8, 4, 2>); break;
            case 4: SPMV_GO(k_spmv_team2<8, 4, 4>); break;
            case 5: SPMV_GO(k_spmv_team2<8, 4, 5>); break;
            case 8: SPMV_GO(k_spmv_team2<8, 4, 8>); break;     // no y store, no w read
            case 9: SPMV_GO(k_spmv_team2<8, 4, 9>); break;     // + no gather
            case 16: SPMV_GO(k_spmv_team2<8, 4, 16>); break;   // one band (no XCD banding)
            case 32: SPMV_GO(k_spmv_team2<8, 4, 32>); break;   // no y store
            case 64: SPMV_GO(k_spmv_team2<8, 4, 64>); break;   // no w load
            case 3: SPMV_GO(k_spmv_team2<8, 4>); break;        // unaligned entry pairs, 32-bit columns (the form before)
            // diagnostics on the production form (16-bit codes, w == x); meaningful only on the compact solver matrix
            case 101: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 1>); break;    // no x gather
            case 132: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 32>); break;   // no y store
            case 133: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 33>); break;   // neither
            case 140:   // y rows kept in LDS until the wavefront's tile loop ends (needs <= 8 tiles per wavefront)
                if (c16 && (rpb / 32 + (int64_t)(c->spmv_grid / 8) * 4 - 1) / ((int64_t)(c->spmv_grid / 8) * 4) <= 8)
                    SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 32768>);
                break;
            case 150: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 262144>); break;   // window bases as a 16-byte broadcast load (the form before)
            case 151: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 524288>); break;             // nontemporal column codes
            case 152: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 524288 | 1048576>); break;   // + nontemporal values (the form before)
            case 153: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 1048576>); break;            // nontemporal values only
            case 102: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 2>); break;       // gathers inside 16 lines
            case 103: if (c16) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 65536>); break;   // gathers inside 1 line
            case 2048: SPMV_GO(k_spmv_team2<8, 4, 2048>); break;   // aligned pairs, 32-bit columns
            default:
                if (c->spmv_unroll == 2 && c16 && c->spmv_deep)
                    SPMV_GO(k_spmv_c16p<8, 2, 16384>);
                else if (c->spmv_unroll == 2 && c16)
                    SPMV_GO(k_spmv_team2<8, 2, 2048 | 4096 | 16384>);
                else if (c->spmv_unroll == 2)
                    SPMV_GO(k_spmv_team2<8, 2>);
                else if (c->spmv_unroll == 6)
                    SPMV_GO(k_spmv_team2<8, 6>);
                else if (c16 && c->spmv_deep) {   // deep-pipelined form: gathers one tile ahead
                    if (dist && wx) SPMV_GO(k_spmv_c16p<8, 4, 8192 | 16384>);
                    else if (dist) SPMV_GO(k_spmv_c16p<8, 4, 8192>);
                    else if (wx) SPMV_GO(k_spmv_c16p<8, 4, 16384>);
                    else SPMV_GO(k_spmv_c16p<8, 4, 0>);
                } else if (ntv && c16 && wx) {   // large matrix: hinted value stream (see load_pair)
                    if (dist) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 8192 | 16384 | 1048576>);
                    else SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 16384 | 1048576>);
                } else if (ntv && c16) {
                    if (dist) SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 8192 | 1048576>);
                    else SPMV_GO(k_spmv_team2<8, 4, 2048 | 4096 | 1048576>);
                } else
                    SPMV_PROD(8, 4);
                break;
            }
            break;
        case 16: SPMV_PROD(16, 4); break;
        default: SPMV_PROD_FEW(32, 4); break;
        }
#undef SPMV_PROD
#undef SPMV_PROD_FEW
        return;
    }
    switch (c->spmv_team) {
    case 4: SPMV_GO(k_spmv_team<4, 2>); break;
    case 8: SPMV_GO(k_spmv_team<8, 4>); break;
    case 16:
        if (c->spmv_unroll == 8)
            SPMV_GO(k_spmv_team<16, 8>);
        else
            SPMV_GO(k_spmv_team<16, 4>);
        break;
    case 32: SPMV_GO(k_spmv_team<32, 4>); break;
    default: SPMV_GO(k_spmv_team<64, 2>); break;
    }
#undef SPMV_GO
}

inline unsigned g1(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

#define RCCLCHK(ctx, expr)                                                                   \
    do {                                                                                     \
        ncclResult_t r__ = (expr);                                                           \
        if (r__ != ncclSuccess) {                                                            \
            (ctx)->err = std::string(#expr) + ": " + g_rccl.GetErrorString(r__);             \
            return FDAPDE_ERCCL;                                                             \
        }                                                                                    \
    } while (0)

int allreduce_sum(fdapde_ctx* c, double* buf, size_t count) {
    if (c->ar_fn) {   // host-staged: device -> host, caller-provided sum over ranks, host -> device
        c->ar_host.resize(count);
        HIPCHK(c, hipMemcpyAsync(c->ar_host.data(), buf, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->ar_fn(c->ar_user, c->ar_host.data(), (int64_t)count) != 0) return fail(c, FDAPDE_ERCCL, "all-reduce callback failed");
        HIPCHK(c, hipMemcpyAsync(buf, c->ar_host.data(), sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
        return FDAPDE_OK;
    }
    RCCLCHK(c, g_rccl.AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, c->stream));
    return FDAPDE_OK;
}
// v (internal DOF order, sub-assembled) -> interface entries summed over the ranks sharing them; optionally carries the
// two fused dot partials of the SpMV (part_a, stride 2) through the same all-reduce: they land in hbuf[n_if], [n_if + 1]
// neighbour-only form (fdapde_halo_setup_peers): pack the per-peer segments, one grouped RCCL call with a send + a receive per peer, the
// all-reduce of the two scalars, then the contributions of every local interface DOF summed in rank order
int halo_sum_peers(fdapde_ctx* c, double* v, const double* part, int np, bool unpack) {
    hipStream_t st = c->stream;
    const int n_peers = (int)c->peer_rank.size();
    const int64_t n_send = c->peer_off.empty() ? 0 : c->peer_off.back();
    double* scal = c->hbuf.p + c->n_if;
    hipLaunchKernelGGL(k_peer_pack, dim3(g1(n_send > 0 ? n_send : 1)), dim3(256), 0, st, n_send, c->peer_send_dof.p, v, c->peer_sendbuf.p, part, np,
                       scal);
    if (c->ar_fn) {   // host-staged
        if (n_peers > 0) {
            if (!c->xchg_fn) return fail(c, FDAPDE_ENOTINIT, "fdapde_comm_set_exchange_callback not called");
            c->xchg_send_h.resize((size_t)n_send), c->xchg_recv_h.resize((size_t)n_send);
            HIPCHK(c, hipMemcpyAsync(c->xchg_send_h.data(), c->peer_sendbuf.p, sizeof(double) * (size_t)n_send, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (c->xchg_fn(c->xchg_user, n_peers, c->peer_rank.data(), c->peer_off.data(), c->xchg_send_h.data(), c->xchg_recv_h.data()) != 0)
                return fail(c, FDAPDE_ERCCL, "exchange callback failed");
            HIPCHK(c, hipMemcpyAsync(c->peer_recvbuf.p, c->xchg_recv_h.data(), sizeof(double) * (size_t)n_send, hipMemcpyHostToDevice, st));
        }
        if (int rc = allreduce_sum(c, scal, 2)) return rc;
    } else {
        if (n_peers > 0) {   // all sends and receives of the exchange form one group (one fused launch, no ordering between peers)
            RCCLCHK(c, g_rccl.GroupStart());
            for (int q = 0; q < n_peers; ++q) {
                const size_t cnt = (size_t)(c->peer_off[(size_t)q + 1] - c->peer_off[(size_t)q]);
                RCCLCHK(c, g_rccl.Send(c->peer_sendbuf.p + c->peer_off[(size_t)q], cnt, ncclFloat64, c->peer_rank[(size_t)q], c->comm, st));
                RCCLCHK(c, g_rccl.Recv(c->peer_recvbuf.p + c->peer_off[(size_t)q], cnt, ncclFloat64, c->peer_rank[(size_t)q], c->comm, st));
            }
            RCCLCHK(c, g_rccl.GroupEnd());
        }
        RCCLCHK(c, g_rccl.AllReduce(scal, scal, 2, ncclFloat64, ncclSum, c->comm, st));
    }
    if (c->n_loc_if > 0)
        hipLaunchKernelGGL(k_peer_sum, dim3(g1(c->n_loc_if)), dim3(256), 0, st, c->n_loc_if, c->halo_dof.p, c->peer_src_off.p, c->peer_src.p,
                           c->peer_recvbuf.p, v, c->hbuf.p, unpack ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}
int halo_sum(fdapde_ctx* c, double* v, const double* part, int np, bool unpack = true) {
    if (c->peer_mode) return halo_sum_peers(c, v, part, np, unpack);
    hipStream_t st = c->stream;
    const unsigned grid = g1(c->n_loc_if > 0 ? c->n_loc_if : 1);
    hipLaunchKernelGGL(k_halo_pack_all, dim3(g1(c->n_if > 0 ? c->n_if : 1)), dim3(256), 0, st, c->n_if, c->halo_inv.p, v, c->hbuf.p, part,
                       np);   // one launch writes every slot (zeros where this rank has no DOF): no memset
    if (int rc = allreduce_sum(c, c->hbuf.p, (size_t)(c->n_if + 2))) return rc;
    if (c->n_loc_if > 0 && unpack)
        hipLaunchKernelGGL(k_halo_unpack, dim3(grid), dim3(256), 0, st, c->n_loc_if, c->halo_dof.p, c->halo_pos.p, c->hbuf.p, v);
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}

}  // namespace

// =================================================================================================================
extern "C" {

int fdapde_abi_version(void) { return FDAPDE_ABI_VERSION; }

int fdapde_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* fdapde_status_string(int s) {
    switch (s) {
    case FDAPDE_OK: return "ok";
    case FDAPDE_EINVAL: return "invalid argument";
    case FDAPDE_ENOMEM: return "out of memory";
    case FDAPDE_ENODEVICE: return "no HIP device";
    case FDAPDE_EHIP: return "HIP runtime error";
    case FDAPDE_ENOTINIT: return "solver must be initialized first!";
    case FDAPDE_ENOCONV: return "Krylov solve did not converge";
    case FDAPDE_EUNSUPPORTED: return "unsupported configuration";
    case FDAPDE_ERCCL: return "RCCL error";
    }
    return "unknown status";
}

int fdapde_ctx_create(int device, fdapde_ctx** out) {
    if (!out) return FDAPDE_EINVAL;
    *out = nullptr;
    fdapde_ctx* c = new (std::nothrow) fdapde_ctx();
    if (!c) return FDAPDE_ENOMEM;
    if (device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device >= n) {
            delete c;
            return FDAPDE_ENODEVICE;
        }
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
            hipEventCreate(&c->ev_p0) != hipSuccess || hipEventCreate(&c->ev_p1) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&c->h_ctl), 4 * sizeof(int32_t)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&c->h_sc), 16 * sizeof(double)) != hipSuccess) {
            delete c;
            return FDAPDE_EHIP;
        }
        c->device = device, c->has_device = true;
        if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->n_cu = 0;
    }
    *out = c;
    return FDAPDE_OK;
}

void fdapde_ctx_destroy(fdapde_ctx* c) {
    if (!c) return;
    if (c->has_device) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        drop_graph(c);
        for (DBuf<int32_t>* b : {&c->cverts, &c->cdofs, &c->adj, &c->rowptr, &c->colidx, &c->diag, &c->slot_i2e, &c->dof_i2e,
                                 &c->dof_e2i, &c->cell_i2e, &c->rb_row, &c->colour_cells, &c->ctl, &c->rowptr_e, &c->colidx_e, &c->dofs_e})
            b->release();
        for (DBuf<double>* b : {&c->vcoords, &c->vals[0], &c->vals[1], &c->force, &c->fq, &c->g, &c->sval, &c->scale, &c->gt,
                                &c->x, &c->r, &c->p, &c->y, &c->s, &c->t, &c->r0, &c->u, &c->part_a, &c->part_b, &c->sc,
                                &c->tmp_e, &c->tmp_i, &c->tmp_v})
            b->release();
        for (auto& b : c->coef) b.release();
        c->slotw.release(), c->sl_off.release(), c->lane_row.release(), c->fq_blk.release(), c->bnd.release(), c->tables.release(), c->reftab.release(), c->lin_sq.release();
        c->bc_off.release(), c->bn_off.release(), c->bc_cell.release(), c->bn_node.release(), c->bc_vert.release();
        c->halo_dof.release(), c->halo_pos.release(), c->owned.release(), c->hbuf.release(), c->sbuf.release();
        c->halo_inv.release(), c->if_slot.release();
        c->peer_send_dof.release(), c->peer_src_off.release(), c->peer_src.release(), c->peer_sendbuf.release(), c->peer_recvbuf.release();
        release_rowdist(c);
        if (c->comm) (void)g_rccl.CommDestroy(c->comm);
        c->lin_mat.release(), c->ar_dev.release(), c->persist_stats.release(), c->persist_x.release(), c->coords_e.release();
        dev_topology_release(&c->topo);
        c->part_cells.release(), c->part_off.release(), c->part_slots.release(), c->wave_slots.release(), c->part_shared.release();
        for (auto& bk : c->bk)
            bk.slot_dof.release(), bk.sl_off.release(), bk.ell_src.release(), bk.imp_off.release(), bk.imp_dof.release(), bk.drop_dof.release(), bk.ell_off.release(),
              bk.ell_code.release(), bk.ell_val.release();
        for (auto& ps : c->ps)
            ps.slot_dof.release(), ps.sl_off.release(), ps.ell_src.release(), ps.exp_off.release(), ps.imp_off.release(),
              ps.imp_pos.release(), ps.ell_off.release(), ps.ell_code.release(), ps.exp_slot.release(), ps.ell_val.release(), ps.board.release(), ps.amax.release();
        for (int v = 0; v < 2; ++v)
            c->sp_rowptr[v].release(), c->sp_colidx[v].release(), c->sp_map[v].release(), c->sp_tbase[v].release(), c->sp_col16[v].release(),
              c->sp_vrow[v].release();
        if (c->h_ctl) (void)hipHostFree(c->h_ctl);
        if (c->h_sc) (void)hipHostFree(c->h_sc);
        (void)hipEventDestroy(c->ev0), (void)hipEventDestroy(c->ev1), (void)hipEventDestroy(c->ev_p0), (void)hipEventDestroy(c->ev_p1);
        for (hipEvent_t e : c->ev_spmv) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(c->stream);
    }
    delete c->lin_state;
    delete c;
}

const char* fdapde_last_error(const fdapde_ctx* c) { return c ? c->err.c_str() : "null context"; }

int fdapde_mesh_upload(fdapde_ctx* c, int M, int N, int64_t n_nodes, const double* nodes, int64_t n_cells,
                       const int32_t* cells, const uint8_t* bnd) {
    if (!c) return FDAPDE_EINVAL;
    c->space_ready = c->dev_ready = c->colour_ready = c->fq_blk_ready = c->fq_bc_ready = false;
    c->assembled[0] = c->assembled[1] = c->force_ready = c->solved = c->dirichlet_applied = false;
    c->op.clear(), c->coef_of_op = false, c->fq_i.clear(), c->fq_cols = 0, c->g_i.clear(), c->have_g = false;
    if (c->topo_ready) dev_topology_release(&c->topo), c->topo_ready = false;
    return host_set_mesh(c->hs, M, N, n_nodes, nodes, n_cells, cells, bnd, c->err);
}

int fdapde_dofs_build(fdapde_ctx* c, int order, int64_t* n_dofs) {
    if (!c) return FDAPDE_EINVAL;
    auto t0 = std::chrono::steady_clock::now();
    c->space_ready = c->dev_ready = c->colour_ready = c->fq_blk_ready = c->part_ready = c->wave_ready = false;
    c->assembled[0] = c->assembled[1] = c->force_ready = c->solved = c->dirichlet_applied = false;
    c->op.clear(), c->coef_of_op = false, c->fq_i.clear(), c->fq_cols = 0, c->g_i.clear(), c->have_g = false;
    c->halo_ready = false, c->lin_ready = false, c->sp_built[0] = c->sp_built[1] = false, c->sp_cur = -1;
    c->scaled_owner = fdapde_ctx::kScaledNone;
    c->ps[0].tried = c->ps[0].ok = c->ps[1].tried = c->ps[1].ok = false;
    c->bk[0].tried = c->bk[0].ok = c->bk[1].tried = c->bk[1].ok = false, c->bk_cur = -1;
    drop_graph(c);
    // The DOF table (reference numbering) is host index work; everything derived from it -- locality numbering, adjacency, CSR
    // patterns, slot maps, assembly block tables -- is built on the device (dev_setup.hip) when the context has one.
    // FDAPDE_SETUP=host keeps the multi-threaded host builder; FDAPDE_SETUP_CHECK=1 runs both and compares every array.
    const char* mode = std::getenv("FDAPDE_SETUP");
    const bool on_device = c->has_device && !(mode && std::strcmp(mode, "host") == 0);
    c->dev_built = false;
    HostSpace& hs = c->hs;
    hs.colidx_i.clear(), hs.cdofs_i.clear(), hs.cverts_i.clear(), hs.vcoords_i.clear(), hs.colidx_e.clear(), hs.rowptr_e.clear(), hs.adj.clear(),
      hs.slotw.clear(), hs.lane_row.clear();
    hs.dofs.clear(), hs.dof_coords.clear();
    int rc = host_build_space(hs, order, c->err, on_device ? 2 : 0);
    if (rc) return rc;
    rc = build_basis_tables(hs.M, order, &c->tb);
    if (rc) return fail(c, rc, "basis tables");
    if (on_device) {
        HIPCHK(c, hipSetDevice(c->device));
        DBuf<double> d_nodes;
        DBuf<int32_t> d_cells;
        DBuf<uint8_t> d_nbnd, d_bnd;
        c->dofs_e.release(), c->coords_e.release();
        HIPCHK(c, d_nodes.upload(hs.nodes.data(), hs.nodes.size(), c->stream));
        HIPCHK(c, d_cells.upload(hs.cells.data(), hs.cells.size(), c->stream));
        HIPCHK(c, d_nbnd.upload(hs.node_bnd.data(), hs.node_bnd.size(), c->stream));
        if (order == 1) {   // LagrangianBasis<D, 1>: dofs = cells, boundary DOFs = node markers (lagrangian_basis.h:96-99)
            hs.n_edges = 0, hs.n_dofs = hs.n_nodes;
            hs.dof_bnd.assign(hs.node_bnd.begin(), hs.node_bnd.end());
        } else {            // order 2: edge DOFs numbered through the device-built topology (dev_topology.hip)
            int32_t* dd = nullptr;
            uint8_t* db = nullptr;
            double* dc = nullptr;
            int64_t ne = 0;
            rc = dev_build_p2_dofs(hs.M, hs.n_nodes, hs.n_cells, d_nodes.p, d_cells.p, d_nbnd.p, c->tb.refnodes, c->stream, &dd, &db, &dc, &ne, c->err);
            if (rc) return rc;
            hs.n_edges = ne, hs.n_dofs = hs.n_nodes + ne;
            adopt(c->dofs_e, dd, (size_t)hs.n_cells * hs.nb), adopt(c->coords_e, dc, (size_t)hs.n_dofs * hs.N), adopt(d_bnd, db, (size_t)hs.n_dofs);
            hs.dof_bnd.resize((size_t)hs.n_dofs);
            HIPCHK(c, hipMemcpyAsync(hs.dof_bnd.data(), d_bnd.p, (size_t)hs.n_dofs, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        DevSpace ds;
        rc = dev_build_space(hs, d_nodes.p, d_cells.p, order == 1 ? d_cells.p : c->dofs_e.p, order == 1 ? d_nbnd.p : d_bnd.p,
                             order == 1 ? d_nodes.p : c->coords_e.p, c->stream, &ds, c->err);
        d_nodes.release(), d_cells.release(), d_nbnd.release(), d_bnd.release();
        if (rc) return rc;
        if (std::getenv("FDAPDE_SETUP_CHECK")) {
            rc = check_dev_space(c, ds, order);
            if (rc) {
                dev_space_release(&ds);
                return rc;
            }
        }
        rc = adopt_dev_space(c, ds);
        if (rc) return rc;
    }
    c->space_ready = true;
    if (n_dofs) *n_dofs = c->hs.n_dofs;
    if (c->has_device) {
        HIPCHK(c, hipSetDevice(c->device));
        rc = upload_space(c);
        if (rc) return rc;
        // small systems build their single-launch solver layout on the host (build_persist_once): the pattern's host mirror is fetched
        // here, as part of the set-up, not by the first solve (the first larger device-to-host copy of a process costs ~8 ms)
        if (c->hs.n_dofs <= c->persist_host_below)
            if (int rc2 = ensure_host(c, kHostPattern)) return rc2;
    }
    c->info = fdapde_info{};
    c->info.t_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FDAPDE_OK;
}

// Triangulation<M,N>(nodes, cells, boundary) beyond the cell list: edges / faces, neighbours, boundary markers
// (fdaPDE/geometry/triangulation.h:143-196 for triangles, 319-399 for tetrahedra), built on the device (dev_topology.hip)
int fdapde_topology_build(fdapde_ctx* c, int64_t* n_facets, int64_t* n_edges) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    const HostSpace& hs = c->hs;
    if (hs.n_cells < 1) return fail(c, FDAPDE_ENOTINIT, "call fdapde_mesh_upload first");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->topo_ready) {
        DBuf<int32_t> d_cells;
        DBuf<uint8_t> d_bnd;
        HIPCHK(c, d_cells.upload(hs.cells.data(), hs.cells.size(), c->stream));
        HIPCHK(c, d_bnd.upload(hs.node_bnd.data(), hs.node_bnd.size(), c->stream));
        const int rc = dev_build_topology(hs.M, hs.n_nodes, hs.n_cells, d_cells.p, d_bnd.p, c->stream, &c->topo, c->err);
        d_cells.release(), d_bnd.release();
        if (rc) return rc;
        c->topo_ready = true;
    }
    if (n_facets) *n_facets = c->topo.n_facets;
    if (n_edges) *n_edges = c->topo.n_edges;
    return FDAPDE_OK;
}

int fdapde_topology_get(fdapde_ctx* c, int32_t* neighbors, int32_t* cell_facets, int32_t* facet_nodes, int32_t* facet_cells,
                        uint8_t* facet_boundary, int32_t* edge_nodes, uint8_t* edge_boundary, int32_t* face_edges) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->topo_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_topology_build first");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTopology& t = c->topo;
    const int M = t.M;
    auto get = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return (dst && src && bytes) ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
    };
    HIPCHK(c, get(neighbors, t.neighbors, sizeof(int32_t) * (size_t)t.n_cells * (M + 1)));
    HIPCHK(c, get(cell_facets, t.cell_facets, sizeof(int32_t) * (size_t)t.n_cells * (M + 1)));
    HIPCHK(c, get(facet_nodes, t.facet_nodes, sizeof(int32_t) * (size_t)t.n_facets * M));
    HIPCHK(c, get(facet_cells, t.facet_cells, sizeof(int32_t) * (size_t)t.n_facets * 2));
    HIPCHK(c, get(facet_boundary, t.facet_bnd, (size_t)t.n_facets));
    if (M == 3) {
        HIPCHK(c, get(edge_nodes, t.edge_nodes, sizeof(int32_t) * (size_t)t.n_edges * 2));
        HIPCHK(c, get(edge_boundary, t.edge_bnd, (size_t)t.n_edges));
        HIPCHK(c, get(face_edges, t.face_edges, sizeof(int32_t) * (size_t)t.n_facets * 3));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDAPDE_OK;
}

int fdapde_sizes(const fdapde_ctx* c, int64_t* n_dofs, int64_t* nnz, int32_t* n_basis, int32_t* n_quadrature, int64_t* n_edges) {
    if (!c || !c->space_ready) return FDAPDE_ENOTINIT;
    if (n_dofs) *n_dofs = c->hs.n_dofs;
    if (nnz) *nnz = c->hs.nnz;
    if (n_basis) *n_basis = c->hs.nb;
    if (n_quadrature) *n_quadrature = c->hs.nq;
    if (n_edges) *n_edges = c->hs.n_edges;
    return FDAPDE_OK;
}

int fdapde_dofs_set_boundary(fdapde_ctx* c, const uint8_t* bnd) {
    if (!c || !bnd) return FDAPDE_EINVAL;
    HostSpace& hs = c->hs;
    if (hs.n_dofs == 0) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    for (int64_t i = 0; i < hs.n_dofs; ++i) hs.dof_bnd[(size_t)i] = bnd[i] ? 1 : 0;
    for (int64_t i = 0; i < hs.n_dofs; ++i) hs.dof_bnd_i[(size_t)i] = hs.dof_bnd[(size_t)hs.dof_i2e[(size_t)i]];
    c->sp_built[1] = false;   // the compact solver pattern drops Dirichlet rows / columns
    c->ps[1].tried = c->ps[1].ok = false, c->bk[1].tried = c->bk[1].ok = false, c->bk_cur = -1;
    drop_graph(c);
    c->solved = false, c->scaled_owner = fdapde_ctx::kScaledNone;
    if (c->dev_ready) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, c->bnd.upload(hs.dof_bnd_i.data(), hs.dof_bnd_i.size(), c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return FDAPDE_OK;
}

int fdapde_dofs_get(const fdapde_ctx* c, int32_t* dofs, uint8_t* bnd, double* coords) {
    if (!c || !c->space_ready) return FDAPDE_ENOTINIT;
    if (dofs || coords)
        if (int rc = ensure_host(const_cast<fdapde_ctx*>(c), kHostDofs)) return rc;   // a device-built space keeps them on the device until asked
    if (dofs) std::memcpy(dofs, c->hs.dofs.data(), sizeof(int32_t) * c->hs.dofs.size());
    if (bnd) std::memcpy(bnd, c->hs.dof_bnd.data(), c->hs.dof_bnd.size());
    if (coords) std::memcpy(coords, c->hs.dof_coords.data(), sizeof(double) * c->hs.dof_coords.size());
    return FDAPDE_OK;
}

int fdapde_pattern_get(const fdapde_ctx* c, int32_t* rowptr, int32_t* colidx) {
    if (!c || !c->space_ready) return FDAPDE_ENOTINIT;
    if (int rc = ensure_host(const_cast<fdapde_ctx*>(c), kHostRefPattern)) return rc;   // a device-built space keeps it on the device until asked
    if (rowptr) std::memcpy(rowptr, c->hs.rowptr_e.data(), sizeof(int32_t) * c->hs.rowptr_e.size());
    if (colidx) std::memcpy(colidx, c->hs.colidx_e.data(), sizeof(int32_t) * c->hs.colidx_e.size());
    return FDAPDE_OK;
}

int fdapde_quadrature_nodes(fdapde_ctx* c, double* out) {
    if (!c || !out) return FDAPDE_EINVAL;
    if (!c->space_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (int rc = need_device(c)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int64_t rows = hs.n_cells * hs.nq;
    DBuf<double> d;
    HIPCHK(c, d.alloc((size_t)rows * hs.N));
    AsmArgs a = asm_args(c);
    if (hs.M == 2)
        hipLaunchKernelGGL(k_quadrature_nodes<2>, dim3(g1(rows)), dim3(256), 0, c->stream, a, c->cell_i2e.p, hs.nq, d.p);
    else
        hipLaunchKernelGGL(k_quadrature_nodes<3>, dim3(g1(rows)), dim3(256), 0, c->stream, a, c->cell_i2e.p, hs.nq, d.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, d.p, sizeof(double) * (size_t)rows * hs.N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    d.release();
    return FDAPDE_OK;
}

int fdapde_set_operator(fdapde_ctx* c, int32_t n_terms, const fdapde_term* terms) {
    if (!c) return FDAPDE_EINVAL;
    std::vector<HostTerm> t;
    bool sym = true;
    int rc = check_terms(c, n_terms, terms, &t, &sym);
    if (rc) return rc;
    c->op = std::move(t), c->op_symmetric = sym, c->coef_of_op = false;
    c->assembled[0] = false, c->solved = false;
    return FDAPDE_OK;
}

int fdapde_set_forcing(fdapde_ctx* c, const double* f_q, int32_t n_cols) {
    if (!c) return FDAPDE_EINVAL;
    if (!c->space_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    const HostSpace& hs = c->hs;
    if (!f_q || n_cols < 1) {
        c->fq_i.clear(), c->fq_cols = 0, c->fq_blk_ready = false, c->fq_bc_ready = false;
        return FDAPDE_OK;
    }
    const int64_t rows = (int64_t)hs.nq * hs.n_cells;
    c->fq_cols = n_cols;
    c->force_ready = false, c->solved = false;
    if (!c->has_device || !c->dev_ready) {   // device-less context: keep the samples in internal cell order on the host
        c->fq_i.resize((size_t)rows * n_cols);
        for (int col = 0; col < n_cols; ++col)
            for (int64_t ci = 0; ci < hs.n_cells; ++ci) {
                const int64_t ce = hs.cell_i2e[(size_t)ci];
                std::memcpy(&c->fq_i[(size_t)col * rows + (size_t)ci * hs.nq], &f_q[(size_t)col * rows + (size_t)ce * hs.nq],
                            sizeof(double) * hs.nq);
            }
    }
    if (c->has_device) {
        HIPCHK(c, hipSetDevice(c->device));
        if (c->dev_ready) {   // upload as handed over, permute to the internal cell order on the device
            c->fq_i.clear();
            DBuf<double> stage;
            HIPCHK(c, stage.upload(f_q, (size_t)rows * n_cols, c->stream));
            HIPCHK(c, c->fq.alloc((size_t)rows * n_cols));
            for (int col = 0; col < n_cols; ++col)
                hipLaunchKernelGGL(k_gather_row_groups, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, hs.n_cells, hs.nq,
                                   c->cell_i2e.p, stage.p + (size_t)col * rows, c->fq.p + (size_t)col * rows);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(c->stream));   // stage is released at the end of this scope
        } else {
            HIPCHK(c, c->fq.upload(c->fq_i.data(), c->fq_i.size(), c->stream));
        }
        c->fq_blk_ready = false;   // fdapde_init turns column 0 into per-visit load coefficients (that IS the quadrature of
                                   // discretize_forcing, fem_assembler.h:122-136, so it belongs to init's timed region)
        // A second copy of column 0 in BLOCK-CELL order for the row-owner sweep: the nq samples of a cell once per assembly block that
        // visits it (1.65 copies on C3), so that the sweep finds them in the window of its own block instead of gathering 32 bytes per
        // visit from all over a 323 MB array (PMC: 2.3 GB fetched for them).  A re-layout of the caller's data, like the permutation
        // above -- no weight, no basis value, no sum enters it: the quadrature stays in fdapde_init.
        c->fq_bc_ready = false;
        if (c->dev_ready && c->adj.n > 0 && c->bc_cell.n > 0 && c->asm_fq_bc) {
            const int64_t n_bc = (int64_t)c->bc_cell.n;
            HIPCHK(c, c->fq_bc.alloc((size_t)n_bc * hs.nq));
            hipLaunchKernelGGL(k_gather_row_groups, dim3((unsigned)((n_bc * hs.nq + 255) / 256)), dim3(256), 0, c->stream, n_bc, hs.nq, c->bc_cell.p,
                               c->fq.p, c->fq_bc.p);
            HIPCHK(c, hipGetLastError());
            c->fq_bc_ready = true;
        }
        HIPCHK(c, c->force.alloc((size_t)hs.n_dofs * n_cols));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return FDAPDE_OK;
}

int fdapde_set_dirichlet(fdapde_ctx* c, const double* g) {
    if (!c) return FDAPDE_EINVAL;
    if (!c->space_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    const HostSpace& hs = c->hs;
    c->solved = false;
    if (!g) {
        c->have_g = false, c->g_i.clear();
        return FDAPDE_OK;
    }
    c->g_i.resize((size_t)hs.n_dofs);
    bool all_zero = true;
    for (int64_t i = 0; i < hs.n_dofs; ++i) {
        c->g_i[(size_t)i] = g[hs.dof_i2e[(size_t)i]];
        all_zero = all_zero && c->g_i[(size_t)i] == 0.0;
    }
    c->have_g = true, c->g_zero = all_zero;
    if (c->has_device) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, c->g.upload(c->g_i.data(), c->g_i.size(), c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return FDAPDE_OK;
}

int fdapde_assemble_operator(fdapde_ctx* c, int32_t which, int32_t n_terms, const fdapde_term* terms, int32_t assembly) {
    if (!c || which < 0 || which > 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<HostTerm> t;
    bool sym = true;
    int rc = check_terms(c, n_terms, terms, &t, &sym);
    if (rc) return rc;
    DevOp op;
    c->coef_of_op = false;   // the shared coefficient slots now hold this call's data
    rc = make_dev_op(c, t, &op, 0);
    if (rc) return rc;
    AsmArgs a = asm_args(c);
    a.vals = c->vals[which].p;
    rc = launch_assembly(c, a, op, assembly);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->assembled[which] = true;
    if (which == FDAPDE_MAT_STIFF) c->op_symmetric = sym, c->solved = false, c->dirichlet_applied = false;
    return FDAPDE_OK;
}

int fdapde_init(fdapde_ctx* c, const fdapde_options* opt) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (c->op.empty()) return fail(c, FDAPDE_ENOTINIT, "no differential operator set");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int assembly = opt ? opt->assembly : FDAPDE_ASSEMBLY_ROWS;
    DevOp op, mass_op{};
    int rc = make_dev_op(c, c->op, &op, 0, c->coef_of_op);
    c->coef_of_op = rc == FDAPDE_OK;
    if (rc) return rc;
    mass_op.n = 1, mass_op.needs_psi = 1, mass_op.needs_rows = 0;
    mass_op.t[0].kind = FDAPDE_REACTION, mass_op.t[0].space_varying = 0, mass_op.t[0].coef = 1.0, mass_op.t[0].cst[0] = 1.0;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    // forcing quadrature, first half (fem_assembler.h:122-136): column 0 as ONE load coefficient per visit slot of the row-owner
    // sweep, sum_q w_q f_q psi_i(p_q), in the summation order the visit loop would use; the sweep below then streams one
    // coalesced double per visit instead of gathering the cell's samples.  Runs on every init: the samples may have changed.
    if (c->fq_cols > 0 && assembly == FDAPDE_ASSEMBLY_ROWS && c->asm_fq_block && c->adj.n > 0) {
        const int64_t n_slices = (int64_t)hs.sl_off.size() - 1;
        HIPCHK(c, c->fq_blk.alloc(c->adj.n));
        hipLaunchKernelGGL(k_visit_load_coeffs, dim3((unsigned)n_slices), dim3(64, 8), 0, c->stream, n_slices, hs.nq, c->sl_off.p,
                           c->adj.p, c->bc_off.p, c->bc_cell.p, c->fq.p, c->tables.p, c->fq_blk.p);
        HIPCHK(c, hipGetLastError());
        c->fq_blk_ready = true;
    } else
        c->fq_blk_ready = false;
    // stiff_ (+ force_ column 0 in the same sweep): fem_solver_base.h:113, 121/133
    AsmArgs a = asm_args(c);
    a.vals = c->vals[FDAPDE_MAT_STIFF].p;
    const int64_t rows = (int64_t)hs.nq * hs.n_cells;
    if (c->fq_cols > 0) a.fq = c->fq.p, a.force = c->force.p;
    rc = launch_assembly(c, a, op, assembly);
    if (rc) return rc;
    if (c->fq_cols == 0) HIPCHK(c, hipMemsetAsync(c->force.p, 0, sizeof(double) * (size_t)hs.n_dofs, c->stream));
    for (int col = 1; col < c->fq_cols; ++col) {   // remaining time columns (parabolic forcing), fem_solver_base.h:124-128
        AsmArgs f = asm_args(c);
        f.fq = c->fq.p + (size_t)col * rows, f.force = c->force.p + (size_t)col * hs.n_dofs;
        rc = launch_assembly(c, f, op, assembly);
        if (rc) return rc;
    }
    // mass_ = discretize_operator(Reaction(1.0)): fem_solver_base.h:136
    AsmArgs m = asm_args(c);
    m.vals = c->vals[FDAPDE_MAT_MASS].p;
    rc = launch_assembly(c, m, mass_op, assembly);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->info.t_assemble_ms = ms;
    c->assembled[0] = c->assembled[1] = true, c->force_ready = true, c->solved = false, c->dirichlet_applied = false;
    return FDAPDE_OK;
}

}   // extern "C" (the solve is split into helpers shared by the elliptic and the parabolic entry points)

namespace {



// compact solver pattern v (0: no Dirichlet reduction, 1: Dirichlet rows / columns dropped) + its 16-bit column codes; host work
// and uploads, done once per function space and boundary mask (fdapde_solver_prepare, or lazily by the first solve)
int build_solver_pattern(fdapde_ctx* c, int v) {
    if (c->sp_built[v]) return FDAPDE_OK;
    if (int rc = ensure_host(c, kHostPattern)) return rc;
    drop_graph(c);
    hipStream_t st = c->stream;
    std::vector<int32_t> rp, ci, map, vrow;
    // rows longer than a team pass (P2): segmented pattern, one team pass per chunk; else the plain compact pattern
    const int T = c->spmv_team;
    bool seg = false;
    if ((T == 8 || T == 16) && c->hs.max_row - 1 > 2 * T && !std::getenv("FDAPDE_SPMV_NOSEG")) {
        const int rc = host_build_solver_pattern_seg(c->hs, v == 1, 2 * T, (64 / T) * 4, rp, ci, map, vrow);
        if (rc == FDAPDE_OK) seg = true;
        else if (rc != FDAPDE_EUNSUPPORTED) return rc;
    }
    if (!seg)
        if (int rc = host_build_solver_pattern(c->hs, v == 1, rp, ci, map)) return rc;
    const int64_t n_csr = (int64_t)rp.size() - 1;   // rows of the CSR arrays (virtual rows when segmented)
    c->sp_nv[v] = seg ? n_csr : 0, c->sp_team = T;
    if (seg) HIPCHK(c, c->sp_vrow[v].upload(vrow.data(), vrow.size(), st));
    if ((size_t)rp.back() + 2 > c->sval.n) HIPCHK(c, c->sval.alloc((size_t)rp.back() + 2));   // pad entries may exceed nnz
    c->sval_layout = -2;
    HIPCHK(c, c->sp_rowptr[v].upload(rp.data(), rp.size(), st));
    HIPCHK(c, c->sp_colidx[v].upload(ci.data(), ci.size(), st));
    HIPCHK(c, c->sp_map[v].upload(map.data(), map.size(), st));
    {   // 16-bit column codes of the same pattern
        std::vector<uint16_t> code;
        std::vector<int32_t> tb;
        if (int rc = host_build_col16(n_csr, rp, ci, code, tb, &c->sp_wide[v])) return rc;
        HIPCHK(c, c->sp_col16[v].upload(code.data(), code.size(), st));
        HIPCHK(c, c->sp_tbase[v].upload(tb.data(), tb.size(), st));
        if (std::getenv("FDAPDE_DEBUG_SETUP"))
            std::fprintf(stderr, "solver pattern %d: %lld entries in %lld %srows, %lld of %lld row groups wide\n", v, (long long)rp.back(),
                         (long long)n_csr, seg ? "virtual " : "", (long long)c->sp_wide[v], (long long)((n_csr + kCodeRows - 1) / kCodeRows));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    c->sp_nnz[v] = rp.back(), c->sp_built[v] = true;
    return FDAPDE_OK;
}

// blocked-ELL layout of the multi-launch SpMV for boundary variant v (k_spmv_blocked), built on the device from the pattern
int build_blocked(fdapde_ctx* c, int v) {
    fdapde_ctx::Blocked& bk = c->bk[v];
    if (bk.tried) return FDAPDE_OK;
    bk.tried = true, bk.ok = false;
    const char* mode = std::getenv("FDAPDE_SETUP");
    if (mode && std::strcmp(mode, "host") == 0) return FDAPDE_OK;   // (no host builder for this layout: the compact CSR path serves)
    PersistLayout pl;
    DevPersist dp;
    // rows per block, measured on C5 (P2, 28 entries per row; CSR kernel 400 us per SpMV): 1024 -> 375 us, 2048 -> 367, 4096 -> 387, 8192 -> 439
    int rows = 2048;
    if (const char* e = std::getenv("FDAPDE_BLOCKED_ROWS")) rows = std::atoi(e);
    const int rc = dev_build_persist_layout(c->hs.n_dofs, c->hs.max_row, c->rowptr.p, c->colidx.p, c->bnd.p, v == 1, 1 << 19, 0, rows, nullptr, 0, false, c->stream, pl, &dp,
                                            c->err);
    if (rc == FDAPDE_EUNSUPPORTED) return FDAPDE_OK;
    if (rc) return rc;
    const int S = pl.R * kPersistT;
    bk.imp_cap = (pl.max_imp + 63) & ~63;
    bk.lds_bytes = 8 * (size_t)(S + bk.imp_cap) + 64;
    if (bk.lds_bytes > 150 * 1024) {
        dev_persist_release(&dp);
        return FDAPDE_OK;
    }
    const size_t n_alloc = (size_t)pl.n_entries + 256;
    adopt(bk.slot_dof, dp.slot_dof, (size_t)pl.G * S), adopt(bk.ell_off, dp.ell_off, (size_t)pl.G + 1), adopt(bk.sl_off, dp.sl_off, (size_t)pl.G * (pl.nsl + 1));
    adopt(bk.ell_code, dp.ell_code, n_alloc), adopt(bk.ell_src, dp.ell_src, n_alloc), adopt(bk.imp_off, dp.imp_off, (size_t)pl.G + 1);
    adopt(bk.imp_dof, dp.imp_pos, (size_t)(pl.n_imp ? pl.n_imp : 1)), adopt(bk.drop_dof, dp.drop_dof, (size_t)(pl.n_drop ? pl.n_drop : 1));
    dev_persist_release(&dp);
    HIPCHK(c, bk.ell_val.alloc(n_alloc));
    HIPCHK(c, hipMemsetAsync(bk.ell_val.p, 0, sizeof(double) * n_alloc, c->stream));
    if (2 * (size_t)pl.G > c->part_a.n) HIPCHK(c, c->part_a.alloc(2 * (size_t)pl.G));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (std::getenv("FDAPDE_DEBUG_SETUP"))
        std::fprintf(stderr, "blocked-ELL SpMV layout %d: %d workgroups x %d rows/thread, %lld interior rows, %lld entries (%lld stored, %.1f %% padding), "
                     "LDS %zu B, imports <= %d (%lld in all)\n", v, pl.G, pl.R, (long long)pl.n_int, (long long)pl.n_entries, (long long)pl.nnz,
                     100.0 * (double)(pl.n_entries - pl.nnz) / (double)(pl.n_entries > 0 ? pl.n_entries : 1), bk.lds_bytes, pl.max_imp, (long long)pl.n_imp);
    bk.meta = std::move(pl);
    bk.filled = false, bk.ok = true;
    return FDAPDE_OK;
}

// Dirichlet reduction + Jacobi scaling of the system matrix A (internal slots): scale, sval = diag(s) A diag(s).
// Done once per matrix (per solve for the elliptic problem, once for all time steps of the parabolic one).
int solve_prepare(fdapde_ctx* c, const double* A, int use_bnd, SolveState* ss, bool symmetric) {
    const int64_t n = c->hs.n_dofs;
    hipStream_t st = c->stream;
    ss->dist = (c->comm != nullptr || c->ar_fn != nullptr) && c->halo_ready;   // multi-GPU: sub-assembled operator of this rank's cells
    ss->owned = ss->dist ? c->owned.p : nullptr;
    ss->use_bnd = use_bnd;
    ss->rowdist = (c->comm != nullptr || c->ar_fn != nullptr) && c->rd.ready && !ss->dist;
    if (ss->rowdist) ss->owned = c->rd.owned.p;
    HIPCHK(c, hipMemsetAsync(c->ctl.p, 0, 4 * sizeof(int32_t), st));
    if (ss->dist) {   // the diagonal is a sum over the ranks sharing a DOF
        hipLaunchKernelGGL(k_diag_extract, dim3(g1(n)), dim3(256), 0, st, n, c->diag.p, A, c->tmp_i.p);
        if (int rc = halo_sum(c, c->tmp_i.p, nullptr, 0)) return rc;
        hipLaunchKernelGGL(k_jacobi_scale_from_diag, dim3(g1(n)), dim3(256), 0, st, n, c->tmp_i.p, c->bnd.p, use_bnd, c->scale.p,
                           c->ctl.p + 3);
    } else {
        hipLaunchKernelGGL(k_jacobi_scale, dim3(g1(n * 16)), dim3(256), 0, st, n, c->rowptr.p, c->diag.p, A, c->bnd.p, use_bnd, c->scale.p, c->ctl.p + 3);
    }
    HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (ss->dist || ss->rowdist) {   // "positive diagonal" (CG admissible) must be ONE decision for all ranks: sum the per-rank flags
        c->h_sc[8] = (double)c->h_ctl[3];
        HIPCHK(c, hipMemcpyAsync(c->sbuf.p + 2, c->h_sc + 8, sizeof(double), hipMemcpyHostToDevice, st));
        if (int rc = allreduce_sum(c, c->sbuf.p + 2, 1)) return rc;
        HIPCHK(c, hipMemcpyAsync(c->h_sc + 8, c->sbuf.p + 2, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        c->h_ctl[3] = c->h_sc[8] != 0.0 ? 1 : 0;
    }
    ss->diag_positive = c->h_ctl[3] == 0;
    if (ss->rowdist) {
        // row-distributed form: this rank's rows are complete (its sub-mesh holds every cell touching an owned DOF), the columns other
        // ranks own take their Jacobi scale from the owner; the whole CG then runs as one launch per rank on a layout of its own
        c->ps[0].filled = c->ps[1].filled = false, c->bk_cur = -1, c->bk[0].filled = c->bk[1].filled = false;
        if (!ss->diag_positive) return fail(c, FDAPDE_EUNSUPPORTED, "the row-distributed solve needs a positive diagonal (Jacobi scaling)");
        const int v = use_bnd ? 1 : 0;
        c->persist_plain = symmetric ? 0 : 1;
        if (int rc = build_rowdist(c, v)) return rc;
        if (c->rd.lay[v].ok && !symmetric && (c->rd.lay[v].ps.meta.sym || c->rd.lay[v].ps.meta.R > 8))
            return fail(c, FDAPDE_EUNSUPPORTED, "the row-distributed BiCGStab needs plain storage and at most 8 rows per thread (layout built for a symmetric operator? re-create the context)");
        if (!c->rd.lay[v].ok) return fail(c, FDAPDE_EUNSUPPORTED, "the row-distributed solve does not take this system (a rank's share needs more than 8 rows per thread, or its lists do not fit)");
        if (int rc = rowdist_import_ghosts(c, v, c->scale.p)) return rc;
        hipLaunchKernelGGL(k_scale_matrix, dim3(g1(n * 16)), dim3(256), 0, st, n, c->rowptr.p, c->colidx.p, A, c->scale.p, c->sval.p);
        c->sp_cur = -1, c->sval_layout = -2;
        if (int rc = fill_rowdist(c, v)) return rc;
        HIPCHK(c, hipGetLastError());
        return FDAPDE_OK;
    }
    // scaled matrix: compact (no diagonal, no Dirichlet rows / columns: ~12 % fewer entries on C3) when every interior
    // diagonal is positive, so that the scaled diagonal is exactly 1; else the full pattern
    // symmetric positive system on one GPU of at most ~2 M interior rows: the solve will run as ONE persistent launch on its own
    // resident layout (kernels_persist.h); the multi-launch kernels then only serve the lift, warm starts and the fall-back, and
    // take the plain full-pattern scaled matrix (no compact pattern / column codes are built for such a system)
    c->ps[0].filled = c->ps[1].filled = false;
    bool persist = false;
    if (c->persist_broken && --c->persist_retry_in <= 0) c->persist_broken = false;   // the contention that broke it may be over
    // symmetric: the single launch is a CG (kernels_persist.h); non-symmetric: a BiCGStab on the plain storage (kernels_persist_bicg.h: six
    // vectors in registers, so at most 8 rows per thread -- larger systems keep the multi-launch BiCGStab)
    c->persist_plain = symmetric ? 0 : 1;
    if ((symmetric || c->persist_bicg) && ss->diag_positive && !ss->dist && c->persist && !c->persist_broken && c->spmv_variant == 2) {
        if (int rc = build_persist(c, use_bnd ? 1 : 0)) return rc;
        const fdapde_ctx::Persist& ps = c->ps[use_bnd ? 1 : 0];
        persist = ps.ok && (symmetric || (!ps.meta.sym && ps.meta.R <= 8));
    }
    // one GPU, positive diagonal, not taken by the persistent CG (non-symmetric operator, or too many rows): the multi-launch
    // kernels apply the operator from the blocked-ELL layout (k_spmv_blocked); the compact CSR pattern is then not built either
    c->bk_cur = -1, c->bk[0].filled = c->bk[1].filled = false;
    bool blocked = false;
    // ... where it pays: long rows (P2).  On 14-entry rows (C3 with the persistent CG switched off) the CSR kernel's finer-grained,
    // software-pipelined workgroups win (45 us against 48-51 us per SpMV), so short-row systems keep the compact CSR pattern.
    const bool long_rows = (double)c->hs.nnz >= 20.0 * (double)n || c->blocked == 2;
    if (!persist && ss->diag_positive && !ss->dist && c->blocked && long_rows && c->spmv_variant == 2) {
        if (int rc = build_blocked(c, use_bnd ? 1 : 0)) return rc;
        blocked = c->bk[use_bnd ? 1 : 0].ok;
    }
    const bool compact = !persist && !blocked && ss->diag_positive && c->spmv_variant == 2 && !std::getenv("FDAPDE_SPMV_FULL");
    if (compact) {
        const int v = use_bnd ? 1 : 0;
        if (int rc = build_solver_pattern(c, v)) return rc;
        if (c->sval_layout != v) {   // entries no full-pattern entry maps to (padding of a segmented pattern) must read 0
            HIPCHK(c, hipMemsetAsync(c->sval.p, 0, sizeof(double) * c->sval.n, st));
            c->sval_layout = v;
        }
        hipLaunchKernelGGL(k_scale_matrix_compact, dim3(g1(n * 16)), dim3(256), 0, st, n, c->rowptr.p, c->colidx.p, A, c->scale.p,
                           c->sp_map[v].p, c->sval.p);
        c->sp_cur = v;
    } else {
        hipLaunchKernelGGL(k_scale_matrix, dim3(g1(n * 16)), dim3(256), 0, st, n, c->rowptr.p, c->colidx.p, A, c->scale.p, c->sval.p);
        c->sp_cur = -1, c->sval_layout = -2;
    }
    if (blocked) {
        const int v = use_bnd ? 1 : 0;
        hipLaunchKernelGGL(k_persist_fill, dim3(g1(c->bk[v].meta.n_entries)), dim3(256), 0, st, c->bk[v].meta.n_entries, c->bk[v].ell_src.p,
                           c->sval.p, c->bk[v].ell_val.p, (unsigned long long*)nullptr);
        c->bk[v].filled = true, c->bk_cur = v;
    }
    if (persist)
        if (int rc = fill_persist(c, use_bnd ? 1 : 0)) return rc;
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}

// Krylov solve of A u = f with u = g on the Dirichlet DOFs (if ss.use_bnd), on the system prepared by solve_prepare.
//   f_dev : right-hand side, internal order, sub-assembled (summed over ranks here when dist)
//   g_dev : Dirichlet values, internal order (read on boundary DOFs only)
//   u0_dev: initial guess in u-space or nullptr (cold start)
// Result in c->u.  Fills c->info (iters, relres, converged, method_used, spmv timing).
int solve_run(fdapde_ctx* c, const SolveState& ss, const double* A, const double* f_dev, const double* g_dev, const double* u0_dev,
              int method, double rtol, int maxit, int check_every, int n_timed) {
    const int64_t n = c->hs.n_dofs;
    hipStream_t st = c->stream;
    const bool dist = ss.dist;
    const uint8_t* owned = ss.owned;
    // partial pairs the SpMV leaves for the vector kernels: one per workgroup of the kernel that applies the scaled operator
    const int np_spmv = (c->bk_cur >= 0 && !dist) ? c->bk[c->bk_cur].meta.G : c->spmv_grid;
    const double* fvec = f_dev;
    if (ss.rowdist) {
        if (u0_dev) return fail(c, FDAPDE_EUNSUPPORTED, "warm starts are not part of the row-distributed solve");
        const bool want_bicg = method == FDAPDE_SOLVER_BICGSTAB || c->rd.lay[ss.use_bnd ? 1 : 0].ps.built_plain;
        if (method == FDAPDE_SOLVER_CG_SR) return fail(c, FDAPDE_EUNSUPPORTED, "the row-distributed solve runs the fused-update CG or BiCGStab");
        method = want_bicg ? FDAPDE_SOLVER_BICGSTAB : FDAPDE_SOLVER_CG_FUSED;
    }
    if (dist) {   // the forcing vector is a sum over the ranks sharing a DOF
        HIPCHK(c, hipMemcpyAsync(c->tmp_e.p, f_dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
        if (int rc = halo_sum(c, c->tmp_e.p, nullptr, 0)) return rc;
        fvec = c->tmp_e.p;
    }
    hipLaunchKernelGGL(k_lift, dim3(g1(n)), dim3(256), 0, st, n, c->bnd.p, g_dev, ss.use_bnd, c->gt.p);
    // the lift is zero (no Dirichlet data, or homogeneous data on one GPU -- across ranks the data may differ, and every rank
    // must take the same path through the collectives): A g~ = 0
    if (!ss.use_bnd || (!dist && !ss.rowdist && g_dev == c->g.p && c->g_zero)) {
        HIPCHK(c, hipMemsetAsync(c->y.p, 0, sizeof(double) * (size_t)n, st));
    } else {
        launch_spmv(c, A, c->gt.p, c->y.p, nullptr, nullptr, nullptr);   // y = A g~
        if (dist)
            if (int rc = halo_sum(c, c->y.p, nullptr, 0)) return rc;
    }
    if (method == FDAPDE_SOLVER_AUTO)   // symmetric + positive diagonal: CG (fused-update form on one GPU, single-reduction form on several)
        method = (c->op_symmetric && ss.diag_positive) ? (dist ? FDAPDE_SOLVER_CG : FDAPDE_SOLVER_CG_FUSED) : FDAPDE_SOLVER_BICGSTAB;
    if (method == FDAPDE_SOLVER_CG && dist && c->world > 1) method = FDAPDE_SOLVER_CG_SR;   // one all-reduce per iteration
    if (method == FDAPDE_SOLVER_CG_FUSED && dist) method = FDAPDE_SOLVER_CG_SR;   // y.y of the assembled y would need its own all-reduce
    if ((method == FDAPDE_SOLVER_CG || method == FDAPDE_SOLVER_CG_SR || method == FDAPDE_SOLVER_CG_FUSED) && !ss.diag_positive)
        return fail(c, FDAPDE_ENOCONV, "CG needs a positive diagonal (operator not SPD?); use BiCGStab");
    const bool bicg = method == FDAPDE_SOLVER_BICGSTAB, cgsr = method == FDAPDE_SOLVER_CG_SR, cgf = method == FDAPDE_SOLVER_CG_FUSED;
    const double tol2 = rtol * rtol;
    const double* ax = nullptr;
    if (u0_dev) {   // warm start: x = (u0 - g~) / s, r = b~ - At x
        hipLaunchKernelGGL(k_krylov_init, dim3(c->vec_grid), dim3(256), 0, st, n, fvec, c->y.p, c->scale.p, c->x.p, c->r.p, c->p.p,
                           (double*)nullptr, c->part_b.p, owned, u0_dev, c->gt.p, (const double*)nullptr, 1);
        launch_spmv(c, c->sval.p, c->x.p, c->t.p, nullptr, nullptr, nullptr);
        if (dist)
            if (int rc = halo_sum(c, c->t.p, nullptr, 0)) return rc;
        ax = c->t.p;
    }
    hipLaunchKernelGGL(k_krylov_init, dim3(c->vec_grid), dim3(256), 0, st, n, fvec, c->y.p, c->scale.p, c->x.p, c->r.p, c->p.p,
                       bicg ? c->r0.p : (double*)nullptr, c->part_b.p, owned, u0_dev, c->gt.p, ax, 0);
    if (dist || ss.rowdist) {
        hipLaunchKernelGGL(k_reduce_partials2, dim3(1), dim3(256), 0, st, c->part_b.p, c->vec_grid, c->sbuf.p);
        if (int rc = allreduce_sum(c, c->sbuf.p, 2)) return rc;
        hipLaunchKernelGGL(k_krylov_init_fin, dim3(1), dim3(256), 0, st, c->sbuf.p, 1, c->sc.p, c->ctl.p, tol2, (double*)nullptr, 0);
    } else {
        // fused-update CG: its launch 0 reads the explicit r.r from the second half of part_b (seeded here)
        const int V = c->cgf_v;
        const int64_t b2 = c->cgf_band ? (((((n + 7) / 8) + 31) & ~int64_t(31)) >> 1) : 0, span = b2 > 0 ? b2 : (n >> 1);
        const int per = (int)((span + 256 * V - 1) / (256 * V)) > 0 ? (int)((span + 256 * V - 1) / (256 * V)) : 1;
        const int cg = b2 > 0 ? 8 * per : per;
        hipLaunchKernelGGL(k_krylov_init_fin, dim3(1), dim3(256), 0, st, c->part_b.p, c->vec_grid, c->sc.p, c->ctl.p, tol2,
                           cgf ? c->part_b.p + cg : (double*)nullptr, cgf ? cg : 0);
    }
    if (cgsr) {   // p = s = 0 before the first update (beta = 0 there)
        HIPCHK(c, hipMemsetAsync(c->p.p, 0, sizeof(double) * (size_t)n, st));
        HIPCHK(c, hipMemsetAsync(c->s.p, 0, sizeof(double) * (size_t)n, st));
    }
    HIPCHK(c, hipGetLastError());
    n_timed = n_timed < 0 ? 0 : (n_timed > 256 ? 256 : n_timed);
    while ((int)c->ev_spmv.size() < 2 * n_timed) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreate(&e));
        c->ev_spmv.push_back(e);
    }
    int timed = 0, launched = 0;
    bool stop = false;
    bool persisted = false;
    if (ss.rowdist) {   // one launch per rank, the launches of all ranks acting as one grid (kernels_persist.h DIST)
        if (int rc = run_rowdist(c, ss.use_bnd ? 1 : 0, tol2, maxit, &persisted, bicg)) return rc;
        if (!persisted) return fail(c, FDAPDE_EUNSUPPORTED, "row-distributed solve: an in-kernel hand-off between the ranks' launches timed out (boards not visible across the devices, or a rank's launch could not be resident); use the element-partitioned exchange (fdapde_halo_setup_peers) instead");
        stop = true, launched = c->h_ctl[1];
    }
    if (cgf && !dist && !ss.rowdist && c->persist && !c->persist_broken && c->ps[ss.use_bnd ? 1 : 0].ok && c->ps[ss.use_bnd ? 1 : 0].filled) {
        // the whole iteration as ONE launch (kernels_persist.h); it leaves sc / ctl as the loop below would
        DebugClock clk;
        if (int rc = run_persist(c, ss.use_bnd ? 1 : 0, tol2, maxit, &persisted)) return rc;
        clk.mark("solve_run: run_persist");
        if (persisted) stop = true, launched = c->h_ctl[1];
    }
    if (bicg && !dist && !ss.rowdist && c->persist && c->persist_bicg && !c->persist_broken) {   // the whole BiCGStab as one launch
        const fdapde_ctx::Persist& ps = c->ps[ss.use_bnd ? 1 : 0];
        if (ps.ok && ps.filled && !ps.meta.sym && ps.meta.R <= 8) {
            if (int rc = run_persist(c, ss.use_bnd ? 1 : 0, tol2, maxit, &persisted, /*bicg=*/true)) return rc;
            if (persisted) stop = true, launched = c->h_ctl[1];
        }
    }
    // one iteration of the fused-update CG: SpMV (p.y, y.y) + k_cgf_update; arguments depend on the iteration's parity only
    const int cgf_V = c->cgf_v;
    // XCD-aware mapping of the update kernel (knob cgf_band): workgroup b serves the elements of SpMV row band b % 8
    const int64_t cgf_band2 = c->cgf_band ? (((((n + 7) / 8) + 31) & ~int64_t(31)) >> 1) : 0;
    const int64_t cgf_span = cgf_band2 > 0 ? cgf_band2 : (n >> 1);
    const int cgf_per = (int)((cgf_span + 256 * cgf_V - 1) / (256 * cgf_V)) > 0 ? (int)((cgf_span + 256 * cgf_V - 1) / (256 * cgf_V)) : 1;
    const int cgf_grid = cgf_band2 > 0 ? 8 * cgf_per : cgf_per;
    auto enqueue_cgf = [&](int it, hipEvent_t e0, hipEvent_t e1) {
        const int cg = cgf_grid;
        launch_spmv(c, c->sval.p, c->p.p, c->y.p, c->p.p, c->part_a.p, c->ctl.p, e0, e1);   // p.y and y.y
#define CGF_GO(...)                                                                                                        \
    hipLaunchKernelGGL((k_cgf_update<__VA_ARGS__>), dim3(cg), dim3(256), 0, st, n, c->y.p, c->p.p, c->x.p, c->r.p, c->part_a.p, np_spmv, \
                       c->part_b.p + (size_t)((it + 1) & 1) * cg, cg, c->part_b.p + (size_t)(it & 1) * cg, c->sc.p, tol2, c->ctl.p, \
                       cgf_band2, c->cgf_nt, c->cgf_lazy, it & 1)
        if (c->cgf_split && cgf_V == 8) CGF_GO(8, 1);
        else if (c->cgf_split && cgf_V == 4) CGF_GO(4, 1);
        else if (cgf_V == 1) CGF_GO(1);
        else if (cgf_V == 2) CGF_GO(2);
        else if (cgf_V == 8) CGF_GO(8);
        else CGF_GO(4);
#undef CGF_GO
    };
    auto enqueue_cgf_fin = [&](int done) {   // explicit r.r of the last update -> sc[3] / stop flag
        hipLaunchKernelGGL(k_cgf_fin, dim3(1), dim3(256), 0, st, c->part_b.p + (size_t)((done - 1) & 1) * cgf_grid, cgf_grid, c->sc.p, tol2,
                           c->ctl.p);
    };
    const int bi_grid = (int)(((n >> 1) + 256 * kBiV - 1) / (256 * kBiV)) > 0 ? (int)(((n >> 1) + 256 * kBiV - 1) / (256 * kBiV)) : 1;
    while (!stop && launched < maxit && !ss.rowdist) {
        const int chunk = (maxit - launched) < check_every ? (maxit - launched) : check_every;
        // a full chunk of the fused-update CG with no timed launch replays ONE hipGraph (2 * chunk + 1 kernel nodes): the
        // arguments repeat with period 2, so the graph captured for iterations 0 .. chunk-1 serves every even-aligned chunk
        bool graphed = false;
        if (cgf && c->use_graph && chunk == check_every && (chunk & 1) == 0 && (launched & 1) == 0 && timed >= n_timed) {
            GraphKey key;
            std::memset(&key, 0, sizeof key);   // padding bytes take part in the memcmp below
            key.sval = c->sval.p, key.rowptr = c->sp_cur >= 0 ? (const void*)c->sp_rowptr[c->sp_cur].p : (const void*)c->rowptr.p;
            key.n = n, key.tol2 = tol2, key.chunk = chunk, key.v = cgf_V, key.grid = c->spmv_grid, key.team = c->spmv_team;
            key.ablate = c->spmv_ablate, key.c16 = c->spmv_c16, key.deep = c->spmv_deep, key.unroll = c->spmv_unroll, key.sp_cur = c->sp_cur;
            // the blocked-ELL layout the captured SpMV nodes read from (its arrays and grid are baked into the graph)
            key.bk_cur = c->bk_cur, key.bk_G = c->bk_cur >= 0 ? c->bk[c->bk_cur].meta.G : 0;
            key.bk_val = c->bk_cur >= 0 ? (const void*)c->bk[c->bk_cur].ell_val.p : nullptr;
            if (!c->cg_graph_exec || std::memcmp(&key, &c->cg_graph_key, sizeof key) != 0) {
                if (c->cg_graph_exec) (void)hipGraphExecDestroy(c->cg_graph_exec), c->cg_graph_exec = nullptr;
                hipGraph_t g = nullptr;
                if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    for (int it = 0; it < chunk; ++it) enqueue_cgf(it, nullptr, nullptr);
                    enqueue_cgf_fin(chunk);
                    if (hipStreamEndCapture(st, &g) == hipSuccess && g &&
                        hipGraphInstantiate(&c->cg_graph_exec, g, nullptr, nullptr, 0) == hipSuccess)
                        c->cg_graph_key = key;
                    else
                        c->cg_graph_exec = nullptr;
                    if (g) (void)hipGraphDestroy(g);
                }
                (void)hipGetLastError();
            }
            if (c->cg_graph_exec && hipGraphLaunch(c->cg_graph_exec, st) == hipSuccess) launched += chunk, graphed = true;
        }
        for (int it = 0; !graphed && it < chunk; ++it, ++launched) {
            if (cgsr) {
                const int parity = launched & 1;
                const bool tm = timed < n_timed && launched % kTimeStride == kTimePhase;   // every kTimeStride-th iteration is timed
                // w = At r with delta = r.(At r) and gamma = r.r (owned rows) fused; multi-GPU: ONE all-reduce carries the
                // interface entries of w and both partials
                launch_spmv(c, c->sval.p, c->r.p, c->y.p, c->r.p, c->part_a.p, c->ctl.p, tm ? c->ev_spmv[2 * timed] : nullptr,
                            tm ? c->ev_spmv[2 * timed + 1] : nullptr, 1, owned);
                if (tm) ++timed;
                const double* part = c->part_a.p;
                int np = np_spmv;
                if (dist) {
                    // pack -> all-reduce; the update kernel reads the summed interface rows straight from hbuf (no unpack launch)
                    if (int rc = halo_sum(c, c->y.p, c->part_a.p, c->spmv_grid, /*unpack=*/false)) return rc;
                    part = c->hbuf.p + c->n_if, np = 1;
                }
                // XCD-aware mapping like k_cgf_update (kCgV elements per lane, bands of the SpMV's rows)
                const int64_t sr_band2 = c->cgf_band ? (((((n + 7) / 8) + 31) & ~int64_t(31)) >> 1) : 0, sr_span = sr_band2 > 0 ? sr_band2 : (n >> 1);
                const int sr_per = (int)((sr_span + 256 * kCgV - 1) / (256 * kCgV)) > 0 ? (int)((sr_span + 256 * kCgV - 1) / (256 * kCgV)) : 1;
                hipLaunchKernelGGL(k_cgsr_update, dim3(sr_band2 > 0 ? 8 * sr_per : sr_per), dim3(256), 0, st, n, c->r.p, c->y.p, c->p.p,
                                   c->s.p, c->x.p, part, np, c->sc.p, parity, launched == 0 ? 1 : 0, tol2, c->ctl.p,
                                   dist ? c->if_slot.p : (const int32_t*)nullptr, dist ? c->hbuf.p : (const double*)nullptr, sr_band2);
            } else if (cgf) {
                const bool tm = timed < n_timed && launched % kTimeStride == kTimePhase;   // every kTimeStride-th iteration is timed
                enqueue_cgf(launched, tm ? c->ev_spmv[2 * timed] : nullptr, tm ? c->ev_spmv[2 * timed + 1] : nullptr);
                if (tm) ++timed;
            } else if (!bicg) {
                const int parity = launched & 1;
                const bool tm = timed < n_timed && launched % kTimeStride == kTimePhase;   // every kTimeStride-th iteration is timed
                launch_spmv(c, c->sval.p, c->p.p, c->y.p, c->p.p, c->part_a.p, c->ctl.p,
                            tm ? c->ev_spmv[2 * timed] : nullptr, tm ? c->ev_spmv[2 * timed + 1] : nullptr);
                if (tm) ++timed;
                if (!dist) {
                    hipLaunchKernelGGL(k_cg_update_xr, dim3(c->cg_grid), dim3(256), 0, st, n, c->y.p, c->r.p, c->part_a.p,
                                       np_spmv, c->part_b.p, c->sc.p, parity, c->ctl.p, owned);
                    hipLaunchKernelGGL(k_cg_update_p, dim3(c->cg_grid), dim3(256), 0, st, n, c->r.p, c->p.p, c->x.p, c->part_b.p,
                                       c->cg_grid, c->sc.p, parity, tol2, c->ctl.p);
                } else {
                    // one all-reduce carries the interface entries of A_p p and the rank's p.Ap partial; a second one
                    // (a single double) carries r.r.  Every rank takes the same stop decision from the same numbers.
                    if (int rc = halo_sum(c, c->y.p, c->part_a.p, c->spmv_grid)) return rc;
                    hipLaunchKernelGGL(k_cg_update_xr, dim3(c->cg_grid), dim3(256), 0, st, n, c->y.p, c->r.p, c->hbuf.p + c->n_if, 1,
                                       c->part_b.p, c->sc.p, parity, c->ctl.p, owned);
                    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, c->part_b.p, c->cg_grid, c->sbuf.p);
                    if (int rc = allreduce_sum(c, c->sbuf.p, 1)) return rc;
                    hipLaunchKernelGGL(k_cg_update_p, dim3(c->cg_grid), dim3(256), 0, st, n, c->r.p, c->p.p, c->x.p, c->sbuf.p, 1,
                                       c->sc.p, parity, tol2, c->ctl.p);
                }
            } else if (!dist) {
                hipLaunchKernelGGL(k_bicg_p, dim3(bi_grid), dim3(256), 0, st, n, c->r.p, c->y.p, c->p.p, c->part_b.p,
                                   bi_grid, c->sc.p, launched == 0 ? 1 : 0, c->ctl.p);
                const bool tm = timed < n_timed && launched % kTimeStride == kTimePhase;   // every kTimeStride-th iteration is timed
                launch_spmv(c, c->sval.p, c->p.p, c->y.p, c->r0.p, c->part_a.p, c->ctl.p,   // v = At p, r0.v
                            tm ? c->ev_spmv[2 * timed] : nullptr, tm ? c->ev_spmv[2 * timed + 1] : nullptr);
                if (tm) ++timed;
                hipLaunchKernelGGL(k_bicg_s, dim3(bi_grid), dim3(256), 0, st, n, c->r.p, c->y.p, c->s.p, c->part_a.p,
                                   np_spmv, c->sc.p, c->ctl.p);
                launch_spmv(c, c->sval.p, c->s.p, c->t.p, c->s.p, c->part_a.p, c->ctl.p);    // t = At s, t.s, t.t
                hipLaunchKernelGGL(k_bicg_xr, dim3(bi_grid), dim3(256), 0, st, n, c->p.p, c->s.p, c->t.p, c->r0.p, c->x.p,
                                   c->r.p, c->part_a.p, np_spmv, c->part_b.p, c->sc.p, c->ctl.p, (const uint8_t*)nullptr);
                hipLaunchKernelGGL(k_bicg_fin, dim3(1), dim3(256), 0, st, c->part_a.p, np_spmv, c->part_b.p, bi_grid,
                                   c->sc.p, tol2, c->ctl.p);
            } else {
                // element-partitioned BiCGStab: every operator application is followed by the interface sum, which also carries
                // the dot fused into the SpMV (w.(A x) needs no weighting); dots of assembled vectors (t.t, r0.r, r.r) count
                // owned rows and cross in two small all-reduces.  sbuf: [0..1] = (r0.r, r.r), [4..5] = (t.s, t.t).
                hipLaunchKernelGGL(k_bicg_p, dim3(bi_grid), dim3(256), 0, st, n, c->r.p, c->y.p, c->p.p, c->sbuf.p, 1, c->sc.p,
                                   launched == 0 ? 1 : 0, c->ctl.p);
                const bool tm = timed < n_timed && launched % kTimeStride == kTimePhase;   // every kTimeStride-th iteration is timed
                launch_spmv(c, c->sval.p, c->p.p, c->y.p, c->r0.p, c->part_a.p, c->ctl.p,
                            tm ? c->ev_spmv[2 * timed] : nullptr, tm ? c->ev_spmv[2 * timed + 1] : nullptr);
                if (tm) ++timed;
                if (int rc = halo_sum(c, c->y.p, c->part_a.p, c->spmv_grid)) return rc;
                hipLaunchKernelGGL(k_bicg_s, dim3(bi_grid), dim3(256), 0, st, n, c->r.p, c->y.p, c->s.p, c->hbuf.p + c->n_if, 1,
                                   c->sc.p, c->ctl.p);
                launch_spmv(c, c->sval.p, c->s.p, c->t.p, c->s.p, c->part_a.p, c->ctl.p);
                if (int rc = halo_sum(c, c->t.p, c->part_a.p, c->spmv_grid)) return rc;
                hipLaunchKernelGGL(k_sq_owned, dim3(c->vec_grid), dim3(256), 0, st, n, c->t.p, owned, c->part_b.p, c->ctl.p);
                hipLaunchKernelGGL(k_bicg_tt_fin, dim3(1), dim3(256), 0, st, c->part_b.p, c->vec_grid, c->hbuf.p + c->n_if,
                                   c->sbuf.p + 4);
                if (int rc = allreduce_sum(c, c->sbuf.p + 5, 1)) return rc;
                hipLaunchKernelGGL(k_bicg_xr, dim3(bi_grid), dim3(256), 0, st, n, c->p.p, c->s.p, c->t.p, c->r0.p, c->x.p,
                                   c->r.p, c->sbuf.p + 4, 1, c->part_b.p, c->sc.p, c->ctl.p, owned);
                hipLaunchKernelGGL(k_reduce_partials2, dim3(1), dim3(256), 0, st, c->part_b.p, bi_grid, c->sbuf.p);
                if (int rc = allreduce_sum(c, c->sbuf.p, 2)) return rc;
                hipLaunchKernelGGL(k_bicg_fin, dim3(1), dim3(256), 0, st, c->sbuf.p + 4, 1, c->sbuf.p, 1, c->sc.p, tol2, c->ctl.p);
            }
        }
        if (cgf && launched > 0 && !graphed) enqueue_cgf_fin(launched);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(c->h_sc, c->sc.p, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        stop = c->h_ctl[0] != 0;
    }
    if (launched == 0 && !persisted) {   // already converged at the initial guess
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(c->h_sc, c->sc.p, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    if (cgf && c->cgf_lazy && !persisted)   // an update of x may still be pending (convergence seen at a poll, or maxit)
        hipLaunchKernelGGL(k_cgf_flush, dim3(c->vec_grid), dim3(256), 0, st, n, c->p.p, c->r.p, c->x.p, c->sc.p, c->ctl.p);
    hipLaunchKernelGGL(k_unscale, dim3(g1(n)), dim3(256), 0, st, n, c->scale.p, persisted ? c->persist_x.p : c->x.p, c->gt.p, c->u.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    const double bb = c->h_sc[0], rr = c->h_sc[3];
    c->info.iters = c->h_ctl[1];
    c->info.relres = bb > 0 ? sqrt(rr / bb) : 0.0;
    c->info.converged = (rr <= tol2 * bb && c->h_ctl[2] == 0) ? 1 : 0;
    c->info.method_used = method;
    c->info.spmv_avg_ms = 0, c->info.spmv_timed = 0;
    {   // launches after the stop flag return at once; only iterations that really ran are averaged
        int real = 0;   // sample k was iteration k * kTimeStride
        while (real < timed && real * kTimeStride + kTimePhase < c->info.iters) ++real;
        double sum = 0;
        for (int i = 0; i < real; ++i) {
            float t = 0;
            HIPCHK(c, hipEventElapsedTime(&t, c->ev_spmv[2 * i], c->ev_spmv[2 * i + 1]));
            sum += t;
        }
        if (real > 0) c->info.spmv_avg_ms = sum / real, c->info.spmv_timed = real;
    }
    c->info.persistent = persisted ? 1 : 0;
    c->info.launch_ms = persisted ? c->persist_launch_ms : 0.0;
    c->info.gather_avg_ms = c->info.update_avg_ms = c->info.spmv_mean_ms = 0;
    if (persisted && !c->persist_host_stats.empty() && c->persist_host_stats[0] > 0) {
        // phase stamps of every workgroup (s_memrealtime ticks of 10 ns).  The operator application of an iteration is complete when
        // the SLOWEST workgroup has its rows: spmv_avg_ms = max over workgroups of their average operator phase (SpMV + import wait);
        // the mean over workgroups is reported next to it; all-gather (which contains the wait for the slowest) and update: means
        const size_t G = c->persist_host_stats.size() / 4;
        double mx = 0, mean = 0, gat = 0, upd = 0;
        for (size_t g = 0; g < G; ++g) {
            const double* st = &c->persist_host_stats[4 * g];
            const double n_it = st[0] > 0 ? st[0] : 1;
            mx = std::max(mx, st[1] / n_it), mean += st[1] / n_it, gat += st[2] / n_it, upd += st[3] / n_it;
        }
        if (std::getenv("FDAPDE_DEBUG_PERSIST")) {   // per-workgroup operator phases (us), with the workgroup's ELL entries
            std::vector<int64_t> eo(G + 1);
            std::vector<int32_t> io(G + 1), xo(G + 1);
            const int v = ss.use_bnd ? 1 : 0;
            (void)hipMemcpy(eo.data(), c->ps[v].ell_off.p, sizeof(int64_t) * (G + 1), hipMemcpyDeviceToHost);
            (void)hipMemcpy(io.data(), c->ps[v].imp_off.p, sizeof(int32_t) * (G + 1), hipMemcpyDeviceToHost);
            (void)hipMemcpy(xo.data(), c->ps[v].exp_off.p, sizeof(int32_t) * (G + 1), hipMemcpyDeviceToHost);
            for (size_t g = 0; g < G; ++g)
                std::fprintf(stderr, "persist wg %zu: operator %.2f us gather %.2f us entries %lld imports %d exports %d\n", g,
                             c->persist_host_stats[4 * g + 1] / std::max(1.0, c->persist_host_stats[4 * g]) * 1e-2,
                             c->persist_host_stats[4 * g + 2] / std::max(1.0, c->persist_host_stats[4 * g]) * 1e-2, (long long)(eo[g + 1] - eo[g]),
                             io[g + 1] - io[g], xo[g + 1] - xo[g]);
        }
        c->info.spmv_avg_ms = mx * 1e-5, c->info.spmv_timed = (int32_t)c->persist_host_stats[0];
        c->info.spmv_mean_ms = mean / (double)G * 1e-5, c->info.gather_avg_ms = gat / (double)G * 1e-5, c->info.update_avg_ms = upd / (double)G * 1e-5;
    }
    if (!c->info.converged) {
        c->err = c->h_ctl[2] ? "Krylov breakdown (operator not SPD for CG, or BiCGStab rho/omega = 0)" : "maxit reached";
        return FDAPDE_ENOCONV;   // reference: success = false (fem_linear_elliptic_solver.h:42-45)
    }
    return FDAPDE_OK;
}

}   // namespace

extern "C" {

// One-time preparation of the solver's compact matrix layout for the current boundary-DOF mask (part of set-up, like
// fdapde_dofs_build; the first solve does it lazily otherwise).  with_dirichlet: the layout used when Dirichlet data are set.
int fdapde_solver_prepare(fdapde_ctx* c, int32_t with_dirichlet) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->spmv_variant != 2) return FDAPDE_OK;
    const int v = with_dirichlet ? 1 : 0;
    if (c->persist && !c->persist_broken && c->comm == nullptr && c->ar_fn == nullptr && (c->op_symmetric || c->persist_bicg)) {
        // single GPU: the single-launch solver's layout (CG for a symmetric operator, BiCGStab on the plain storage otherwise); when the
        // system qualifies for it, the compact pattern and the column codes of the multi-launch SpMV are not needed (a matrix that turns
        // out not to qualify at solve time falls back and builds them lazily)
        c->persist_plain = c->op_symmetric ? 0 : 1;
        if (int rc = build_persist(c, v)) return rc;
        if (c->ps[v].ok && (c->op_symmetric || (!c->ps[v].meta.sym && c->ps[v].meta.R <= 8))) return FDAPDE_OK;
    }
    if (c->blocked && c->comm == nullptr && c->ar_fn == nullptr &&
        ((double)c->hs.nnz >= 20.0 * (double)c->hs.n_dofs || c->blocked == 2)) {   // single GPU, long rows: the multi-launch kernels use the blocked-ELL layout
        if (int rc = build_blocked(c, v)) return rc;
        if (c->bk[v].ok) return FDAPDE_OK;
    }
    return build_solver_pattern(c, v);
}

// What the in-solve SpMV works on, for the roofline figures of bench.py: the interior block A_II as a plain CSR operator
// (rows / entries; algorithmic bytes = 12 nnz + 4 (n + 1) + 16 n on it) and the bytes one launch of the solver's kernel really
// streams from the compact coded layout (values 8 B + column codes 2 B per stored entry, row pointers, window bases, virtual-row
// table of a segmented pattern, x gathered once and y written once for every row of the full vector).
int fdapde_solver_layout(fdapde_ctx* c, int32_t with_dirichlet, int64_t* n_interior, int64_t* nnz_interior, double* streamed_bytes) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    HIPCHK(c, hipSetDevice(c->device));
    const int v = with_dirichlet ? 1 : 0;
    const bool persist = c->persist && !c->persist_broken && c->ps[v].tried && c->ps[v].ok;
    const bool blocked = !persist && c->bk[v].tried && c->bk[v].ok;
    if (c->spmv_variant == 2 && !persist && !blocked)
        if (int rc = build_solver_pattern(c, v)) return rc;
    if (int rc = ensure_host(c, kHostPattern)) return rc;
    const HostSpace& hs = c->hs;
    int64_t ni = 0, nz = 0;
    for (int64_t i = 0; i < hs.n_dofs; ++i) {
        if (v && hs.dof_bnd_i[(size_t)i]) continue;
        ++ni;
        for (int32_t k = hs.rowptr_i[(size_t)i]; k < hs.rowptr_i[(size_t)i + 1]; ++k)
            if (!(v && hs.dof_bnd_i[(size_t)hs.colidx_i[(size_t)k]])) ++nz;
    }
    if (n_interior) *n_interior = ni;
    if (nnz_interior) *nnz_interior = nz;
    if (streamed_bytes) {
        if (persist) {   // one iteration of the persistent CG: the ELL blocks (8 + 2 bytes per entry, padding included) + the exchanged
                         // entries of p (two 8-byte granules each, written once and read once)
            // (fdapde_solver_layout_kind tells whether the blocks stream at all: the resident form reads them from LDS)
            *streamed_bytes = 10.0 * (double)c->ps[v].meta.n_entries + 32.0 * (double)c->ps[v].meta.n_board;
        } else if (blocked) {   // ELL blocks + x staged once per block (own rows and imports) + y written once
            *streamed_bytes = 10.0 * (double)c->bk[v].meta.n_entries + 8.0 * (double)(c->bk[v].meta.n_int + c->bk[v].meta.n_imp) + 8.0 * (double)c->bk[v].meta.n_int;
        } else if (c->spmv_variant == 2 && c->sp_built[v]) {
            const int64_t n_csr = c->sp_nv[v] > 0 ? c->sp_nv[v] : hs.n_dofs;
            *streamed_bytes = 10.0 * (double)c->sp_nnz[v] + 4.0 * (double)(n_csr + 1) + 16.0 * (double)((n_csr + kCodeRows - 1) / kCodeRows) +
                              (c->sp_nv[v] > 0 ? 8.0 * (double)n_csr : 0.0) + 16.0 * (double)hs.n_dofs +
                              4.0 * (double)c->sp_wide[v] * kCodeRows * ((double)c->sp_nnz[v] / (double)(n_csr > 0 ? n_csr : 1));
        } else
            *streamed_bytes = 12.0 * (double)hs.nnz + 4.0 * (double)(hs.n_dofs + 1) + 16.0 * (double)hs.n_dofs;
    }
    return FDAPDE_OK;
}

// which layout the solver holds for the boundary variant (after fdapde_solver_prepare / a solve): kind 0 compact CSR, 1 blocked ELL
// (multi-launch), 2 persistent launch with streaming blocks, 3 persistent launch with the blocks resident in LDS
int fdapde_solver_layout_kind(fdapde_ctx* c, int32_t with_dirichlet, int32_t* kind, int32_t* symmetric_storage, int32_t* workgroups,
                              int32_t* rows_per_thread) {
    if (!c) return FDAPDE_EINVAL;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    const int v = with_dirichlet ? 1 : 0;
    const bool persist = c->persist && !c->persist_broken && c->ps[v].tried && c->ps[v].ok;
    const bool blocked = !persist && c->bk[v].tried && c->bk[v].ok;
    if (kind) *kind = persist ? (c->ps[v].stream ? 2 : 3) : blocked ? 1 : 0;
    if (symmetric_storage) *symmetric_storage = persist && c->ps[v].meta.sym ? 1 : 0;
    if (workgroups) *workgroups = persist ? c->ps[v].meta.G : blocked ? c->bk[v].meta.G : 0;
    if (rows_per_thread) *rows_per_thread = persist ? c->ps[v].meta.R : blocked ? c->bk[v].meta.R : 0;
    return FDAPDE_OK;
}

int fdapde_solve(fdapde_ctx* c, const fdapde_options* opt, fdapde_info* info) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->assembled[0] || !c->force_ready)
        return fail(c, FDAPDE_ENOTINIT, "solver must be initialized first!");   // fem_linear_elliptic_solver.h:36
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n = c->hs.n_dofs;
    const double rtol = (opt && opt->rtol > 0) ? opt->rtol : 1e-10;
    const int maxit = (opt && opt->maxit > 0) ? opt->maxit : (int)(10 * n < 100000 ? 10 * n : 100000);
    const int check_every = (opt && opt->check_every > 0) ? opt->check_every : 32;
    const double* A = c->vals[FDAPDE_MAT_STIFF].p;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    SolveState ss;
    c->scaled_owner = fdapde_ctx::kScaledSolve;
    DebugClock clk;
    if (int rc = solve_prepare(c, A, c->have_g ? 1 : 0, &ss, c->op_symmetric)) return rc;
    clk.mark("fdapde_solve: solve_prepare");
    const int rc = solve_run(c, ss, A, c->force.p, c->g.p, nullptr, opt ? opt->method : FDAPDE_SOLVER_AUTO, rtol, maxit, check_every,
                             opt ? opt->time_spmv : 0);
    clk.mark("fdapde_solve: solve_run");
    if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->info.t_solve_ms = ms;
    c->solved = true, c->dirichlet_applied = c->have_g;
    if (info) *info = c->info;
    return rc;
}

// FEMLinearParabolicSolver::solve (fdaPDE/finite_elements/solvers/fem_linear_parabolic_solver.h:37-72): implicit Euler,
//   K = M / dt + A ; Dirichlet rows of K ; for i = 0 .. m-2:  rhs = (M / dt) u_i + f_{i+1} ; rhs[boundary] = g(., i+1) ;
//   u_{i+1} = K^{-1} rhs.   The reference factorises K once with SparseLU; here K is scaled once and every step is a
//   Jacobi-PCG (or BiCGStab) solve warm-started from u_i.  Forcing columns come from fdapde_set_forcing (n_times columns).
int fdapde_solve_parabolic(fdapde_ctx* c, const fdapde_options* opt, int32_t n_times, double delta_t, const double* initial_condition,
                           const double* dirichlet, double* solution, fdapde_info* info) {
    if (!c || n_times < 1 || !(delta_t > 0) || !initial_condition || !solution) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->assembled[0] || !c->assembled[1] || !c->force_ready)
        return fail(c, FDAPDE_ENOTINIT, "solver must be initialized first!");   // fem_linear_parabolic_solver.h:39
    if (c->fq_cols < n_times) return fail(c, FDAPDE_EINVAL, "forcing data needs one column per time point");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int64_t n = hs.n_dofs;
    hipStream_t st = c->stream;
    const double rtol = (opt && opt->rtol > 0) ? opt->rtol : 1e-10;
    const int maxit = (opt && opt->maxit > 0) ? opt->maxit : (int)(10 * n < 100000 ? 10 * n : 100000);
    const int check_every = (opt && opt->check_every > 0) ? opt->check_every : 8;
    const double inv_dt = 1.0 / delta_t;
    DBuf<double> kmat, uprev, rhs, gcol;
    HIPCHK(c, kmat.alloc((size_t)hs.nnz + 2));
    HIPCHK(c, uprev.alloc((size_t)n));
    HIPCHK(c, rhs.alloc((size_t)n));
    HIPCHK(c, gcol.alloc((size_t)n));
    std::vector<double> tmp((size_t)n);
    auto to_internal = [&](const double* ext) {
        for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = ext[hs.dof_i2e[(size_t)i]];
    };
    HIPCHK(c, hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(k_matrix_combine, dim3(g1(hs.nnz)), dim3(256), 0, st, hs.nnz, c->vals[FDAPDE_MAT_MASS].p,
                       c->vals[FDAPDE_MAT_STIFF].p, inv_dt, kmat.p);
    SolveState ss;
    c->scaled_owner = fdapde_ctx::kScaledParabolic;
    if (int rc = solve_prepare(c, kmat.p, dirichlet ? 1 : 0, &ss, c->op_symmetric)) return rc;
    to_internal(initial_condition);
    HIPCHK(c, hipMemcpyAsync(uprev.p, tmp.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    std::memcpy(solution, initial_condition, sizeof(double) * (size_t)n);   // solution_.col(0) = initial condition (line 46)
    int total_iters = 0, rc_all = FDAPDE_OK;
    double worst = 0;
    for (int32_t i = 0; i + 1 < n_times; ++i) {
        launch_spmv(c, c->vals[FDAPDE_MAT_MASS].p, uprev.p, c->s.p, nullptr, nullptr, nullptr);   // M u_i
        hipLaunchKernelGGL(k_parabolic_rhs, dim3(g1(n)), dim3(256), 0, st, n, c->s.p, inv_dt, c->force.p + (size_t)(i + 1) * n, rhs.p);
        if (dirichlet) {
            to_internal(dirichlet + (size_t)(i + 1) * n);
            HIPCHK(c, hipMemcpyAsync(gcol.p, tmp.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));
        }
        const int rc = solve_run(c, ss, kmat.p, rhs.p, gcol.p, uprev.p, opt ? opt->method : FDAPDE_SOLVER_AUTO, rtol, maxit,
                                 check_every, 0);
        if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
        if (rc == FDAPDE_ENOCONV) rc_all = rc;
        total_iters += c->info.iters;
        worst = c->info.relres > worst ? c->info.relres : worst;
        HIPCHK(c, hipMemcpyAsync(uprev.p, c->u.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_scatter_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->u.p, c->tmp_e.p);
        HIPCHK(c, hipMemcpyAsync(solution + (size_t)(i + 1) * n, c->tmp_e.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->info.t_solve_ms = ms, c->info.iters = total_iters, c->info.relres = worst, c->info.converged = rc_all == FDAPDE_OK ? 1 : 0;
    if (info) *info = c->info;
    kmat.release(), uprev.release(), rhs.release(), gcol.release();
    return rc_all;
}

}   // extern "C"

namespace {

// Q columns of fdapde_lin_solve at once (kernels_multirhs.h): b_ext / x_ext are Q host columns of n, reference numbering
template <int Q>
int lin_solve_batch(fdapde_ctx* c, const double* b_ext, double* x_ext, double rtol, int maxit, int check_every, int* iters,
                    double* relres, bool* converged) {
    const HostSpace& hs = c->hs;
    const int64_t n = hs.n_dofs;
    hipStream_t st = c->stream;
    const double tol2 = rtol * rtol;
    DBuf<double> B, X, R, P, Y, part_spmm, part_rr, sc;
    const size_t nq = (size_t)n * Q;
    for (DBuf<double>* b : {&B, &X, &R, &P, &Y}) HIPCHK(c, b->alloc(nq));
    const int64_t nh = n * (Q / 2);
    const int grid_v = (int)std::min<int64_t>(std::max<int64_t>(1, (nh + 256 * kQV - 1) / (256 * kQV)), 1024);
    const int grid_m = (int)std::min<int64_t>((n + 31) / 32, 2048);
    HIPCHK(c, part_spmm.alloc((size_t)grid_m * 2 * Q));
    HIPCHK(c, part_rr.alloc((size_t)grid_v * Q));
    HIPCHK(c, sc.alloc(5 * Q));
    HIPCHK(c, hipMemcpyAsync(B.p, b_ext, sizeof(double) * nq, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_q_init<Q>, dim3(grid_v), dim3(256), 0, st, n, B.p, c->dof_i2e.p, c->scale.p, X.p, R.p, P.p, part_rr.p);
    hipLaunchKernelGGL(k_q_init_fin<Q>, dim3(1), dim3(256), 0, st, part_rr.p, grid_v, sc.p, c->ctl.p);
    int launched = 0;
    bool stop = false;
    std::vector<double> h_sc(5 * Q);
    while (!stop && launched < maxit) {
        const int chunk = (maxit - launched) < check_every ? (maxit - launched) : check_every;
        for (int it = 0; it < chunk; ++it, ++launched) {
            hipLaunchKernelGGL(k_spmm_full<Q>, dim3(grid_m), dim3(256), 0, st, n, c->rowptr.p, c->colidx.p, c->lin_sq.p, P.p, Y.p,
                               part_spmm.p, c->ctl.p);
            hipLaunchKernelGGL(k_q_scalars<Q>, dim3(1), dim3(256), 0, st, part_spmm.p, grid_m, part_rr.p, grid_v, sc.p, tol2, c->ctl.p);
            hipLaunchKernelGGL(k_q_update<Q>, dim3(grid_v), dim3(256), 0, st, n, Y.p, P.p, X.p, R.p, sc.p, part_rr.p, c->ctl.p);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        stop = c->h_ctl[0] != 0;
    }
    if (!stop) {   // maxit: one more scalar pass so that sc holds the r.r of the last update
        hipLaunchKernelGGL(k_spmm_full<Q>, dim3(grid_m), dim3(256), 0, st, n, c->rowptr.p, c->colidx.p, c->lin_sq.p, P.p, Y.p, part_spmm.p,
                           c->ctl.p);
        hipLaunchKernelGGL(k_q_scalars<Q>, dim3(1), dim3(256), 0, st, part_spmm.p, grid_m, part_rr.p, grid_v, sc.p, tol2, c->ctl.p);
    }
    hipLaunchKernelGGL(k_q_unscale<Q>, dim3(g1(n)), dim3(256), 0, st, n, X.p, c->scale.p, c->dof_i2e.p, B.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(x_ext, B.p, sizeof(double) * nq, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h_sc.data(), sc.p, sizeof(double) * 5 * Q, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *iters = c->h_ctl[1], *converged = c->h_ctl[2] == 0, *relres = 0;
    for (int q = 0; q < Q; ++q) {
        const double bb = h_sc[(size_t)q], rr = h_sc[(size_t)Q + q];
        const double rel = bb > 0 ? sqrt(rr / bb) : 0.0;
        *relres = rel > *relres ? rel : *relres;
        if (!(rr <= tol2 * bb)) *converged = false;
    }
    for (DBuf<double>* b : {&B, &X, &R, &P, &Y, &part_spmm, &part_rr, &sc}) b->release();
    return FDAPDE_OK;
}

}   // namespace

extern "C" {

// fdapde::SparseLU<SpMatrix<double>>::compute (fdaPDE/utils/symbols.h:142-146): "factorise" once.  Here: copy the matrix,
// Jacobi-scale it once; every later fdapde_lin_solve is a Krylov run on the prepared system.
int fdapde_lin_compute(fdapde_ctx* c, int32_t which, const double* values, int32_t symmetric) {
    if (!c || which < 0 || which > 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (!values && !c->assembled[which]) return fail(c, FDAPDE_ENOTINIT, "matrix not assembled");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    HIPCHK(c, c->lin_mat.alloc((size_t)hs.nnz + 2));
    if (values) {   // reference slot order -> internal slots
        HIPCHK(c, hipMemcpyAsync(c->tmp_v.p, values, sizeof(double) * (size_t)hs.nnz, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_f64, dim3(g1(hs.nnz)), dim3(256), 0, c->stream, hs.nnz, c->slot_i2e.p, c->tmp_v.p, c->lin_mat.p);
        HIPCHK(c, hipGetLastError());
        c->lin_symmetric = symmetric != 0;
    } else {
        HIPCHK(c, hipMemcpyAsync(c->lin_mat.p, c->vals[which].p, sizeof(double) * (size_t)hs.nnz, hipMemcpyDeviceToDevice, c->stream));
        c->lin_symmetric = which == FDAPDE_MAT_MASS ? true : c->op_symmetric;
    }
    if (!c->lin_state) c->lin_state = new SolveStateHolder();
    c->scaled_owner = fdapde_ctx::kScaledNone;
    if (int rc = solve_prepare(c, c->lin_mat.p, 0, &c->lin_state->ss, c->lin_symmetric)) return rc;
    c->scaled_owner = fdapde_ctx::kScaledLin;   // scale / sval now belong to the handle
    c->lin_ready = true, c->lin_sq_ready = false;
    return FDAPDE_OK;
}

// fdapde::SparseLU::solve(b) (fdaPDE/utils/symbols.h:148-155), dense right-hand sides: b, x column-major n_dofs x n_rhs
int fdapde_lin_solve(fdapde_ctx* c, const fdapde_options* opt, const double* b, int32_t n_rhs, double* x, fdapde_info* info) {
    if (!c || !b || !x || n_rhs < 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->lin_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_lin_compute first");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int64_t n = hs.n_dofs;
    hipStream_t st = c->stream;
    const double rtol = (opt && opt->rtol > 0) ? opt->rtol : 1e-10;
    const int maxit = (opt && opt->maxit > 0) ? opt->maxit : (int)(10 * n < 100000 ? 10 * n : 100000);
    const int check_every = (opt && opt->check_every > 0) ? opt->check_every : 32;
    int method = opt ? opt->method : FDAPDE_SOLVER_AUTO;
    if (method == FDAPDE_SOLVER_AUTO)
        method = (c->lin_symmetric && c->lin_state->ss.diag_positive) ? FDAPDE_SOLVER_CG_FUSED : FDAPDE_SOLVER_BICGSTAB;
    if (c->scaled_owner != fdapde_ctx::kScaledLin) {   // an elliptic / parabolic solve in between has overwritten scale and the scaled copy
        c->scaled_owner = fdapde_ctx::kScaledNone;      // (whatever init / set_* calls followed it): prepare again (cheap)
        if (int rc = solve_prepare(c, c->lin_mat.p, 0, &c->lin_state->ss, c->lin_symmetric)) return rc;
        c->scaled_owner = fdapde_ctx::kScaledLin;
    }
    c->solved = false;   // c->u is about to hold the handle's solutions, not PDE::solution()
    DBuf<double> rhs;
    HIPCHK(c, rhs.alloc((size_t)n));
    HIPCHK(c, hipEventRecord(c->ev0, st));
    int total = 0, rc_all = FDAPDE_OK;
    double worst = 0;
    int32_t j0 = 0;
    // several columns against a symmetric positive system on one GPU: batches of 8 / 4 columns share every pass over the
    // matrix (kernels_multirhs.h); what is left goes column by column
    // (a system the persistent CG takes is faster column by column -- one launch each, no vector traffic -- than batched through
    // the multi-launch SpMM: C3-size, 22.5 ms per column against 32 ms per column in a batch of 8)
    const bool persist_cols = c->persist && !c->persist_broken && c->ps[0].ok && c->ps[0].filled;
    const bool batched = c->multi_rhs && n_rhs >= 4 && method == FDAPDE_SOLVER_CG_FUSED && !c->lin_state->ss.dist && !c->lin_state->ss.rowdist && !persist_cols;
    if (batched) {
        if (!c->lin_sq_ready) {   // full-pattern scaled copy (explicit unit diagonal), once per prepared matrix
            HIPCHK(c, c->lin_sq.alloc((size_t)hs.nnz + 2));
            hipLaunchKernelGGL(k_scale_matrix, dim3(g1(n * 16)), dim3(256), 0, st, n, c->rowptr.p, c->colidx.p, c->lin_mat.p, c->scale.p,
                               c->lin_sq.p);
            c->lin_sq_ready = true;
        }
        while (n_rhs - j0 >= 4) {   // pairs are faster column by column (C3-size system: 77 ms against 92 ms batched)
            const int q = n_rhs - j0 >= 8 ? 8 : 4;
            int its = 0, rc = FDAPDE_OK;
            double rel = 0;
            bool ok = true;
            if (q == 8) rc = lin_solve_batch<8>(c, b + (size_t)j0 * n, x + (size_t)j0 * n, rtol, maxit, check_every, &its, &rel, &ok);
            else rc = lin_solve_batch<4>(c, b + (size_t)j0 * n, x + (size_t)j0 * n, rtol, maxit, check_every, &its, &rel, &ok);
            if (rc != FDAPDE_OK) return rc;
            if (!ok) rc_all = FDAPDE_ENOCONV;
            total += its, worst = rel > worst ? rel : worst;
            c->info.method_used = method;
            j0 += q;
        }
    }
    for (int32_t j = j0; j < n_rhs; ++j) {
        HIPCHK(c, hipMemcpyAsync(c->tmp_e.p, b + (size_t)j * n, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_gather_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->tmp_e.p, rhs.p);
        const int rc = solve_run(c, c->lin_state->ss, c->lin_mat.p, rhs.p, c->g.p, nullptr, method, rtol, maxit, check_every, 0);
        if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
        if (rc == FDAPDE_ENOCONV) rc_all = rc;
        total += c->info.iters, worst = c->info.relres > worst ? c->info.relres : worst;
        hipLaunchKernelGGL(k_scatter_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->u.p, c->tmp_e.p);
        HIPCHK(c, hipMemcpyAsync(x + (size_t)j * n, c->tmp_e.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->info.t_solve_ms = ms, c->info.iters = total, c->info.relres = worst, c->info.converged = rc_all == FDAPDE_OK ? 1 : 0;
    if (info) *info = c->info;
    rhs.release();
    return rc_all;
}

// pointwise_evaluation::eval (basis/lagrangian_basis.h:203-235): locate + evaluate.  The bin grid over the cells' bounding
// boxes is built on the host per call (index work, like the reference's KD-tree build at first use, tree_search.h:47-62).
int fdapde_eval_pointwise(fdapde_ctx* c, int64_t n_locs, const double* locs_colmajor, int32_t* cell_ids, double* values) {
    if (!c || n_locs < 1 || !locs_colmajor || !cell_ids || !values) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_host(c, kHostCells)) return rc;
    const HostSpace& hs = c->hs;
    const int M = hs.M, nv = M + 1, NP = M == 2 ? 2 : 4;
    // uniform grid with about one cell per bin on average
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, inv_h[3] = {0, 0, 0};
    int32_t dims[3] = {1, 1, 1};
    for (int d = 0; d < M; ++d) {
        lo[d] = hi[d] = hs.vcoords_i[(size_t)d];
        for (int64_t i = 0; i < hs.n_nodes; ++i) {
            const double v = hs.vcoords_i[(size_t)i * NP + d];
            lo[d] = v < lo[d] ? v : lo[d], hi[d] = v > hi[d] ? v : hi[d];
        }
    }
    const int g = (int)std::max(1.0, std::floor(std::pow((double)hs.n_cells, 1.0 / M)));
    int64_t n_bins = 1;
    for (int d = 0; d < M; ++d) {
        dims[d] = g, n_bins *= g;
        inv_h[d] = hi[d] > lo[d] ? g / (hi[d] - lo[d]) : 0.0;
    }
    auto range = [&](int64_t cell, int d, int& b0, int& b1) {
        double mn = 1e300, mx = -1e300;
        for (int v = 0; v < nv; ++v) {
            const double x = hs.vcoords_i[(size_t)hs.cverts_i[(size_t)cell * nv + v] * NP + d];
            mn = x < mn ? x : mn, mx = x > mx ? x : mx;
        }
        b0 = (int)std::floor((mn - lo[d]) * inv_h[d] - 1e-9), b1 = (int)std::floor((mx - lo[d]) * inv_h[d] + 1e-9);
        b0 = b0 < 0 ? 0 : b0, b1 = b1 >= dims[d] ? dims[d] - 1 : b1;
    };
    std::vector<int32_t> bin_ptr((size_t)n_bins + 1, 0), bin_cells, pos;
    for (int pass = 0; pass < 2; ++pass) {   // pass 0 counts the (cell, bin) overlaps, pass 1 fills the bin lists and runs the kernel
        if (pass == 1) pos.assign(bin_ptr.begin(), bin_ptr.end() - 1);
        if (pass == 1) bin_cells.assign((size_t)bin_ptr[(size_t)n_bins], 0);
        for (int64_t cell = 0; cell < hs.n_cells; ++cell) {
            int b0[3] = {0, 0, 0}, b1[3] = {0, 0, 0};
            for (int d = 0; d < M; ++d) range(cell, d, b0[d], b1[d]);
            for (int z = b0[2]; z <= b1[2]; ++z)
                for (int y = b0[1]; y <= b1[1]; ++y)
                    for (int x = b0[0]; x <= b1[0]; ++x) {
                        const int64_t bin = M == 2 ? (int64_t)y * dims[0] + x : ((int64_t)z * dims[1] + y) * dims[0] + x;
                        if (pass == 0)
                            ++bin_ptr[(size_t)bin + 1];
                        else
                            bin_cells[(size_t)pos[(size_t)bin]++] = (int32_t)cell;
                    }
        }
        if (pass == 0) {
            for (int64_t b = 0; b < n_bins; ++b) bin_ptr[(size_t)b + 1] += bin_ptr[(size_t)b];
        } else {
            DBuf<int32_t> d_ptr, d_cells, d_dims, d_out;
            DBuf<double> d_locs, d_lo, d_invh, d_vals;
            hipStream_t st = c->stream;
            HIPCHK(c, d_ptr.upload(bin_ptr.data(), bin_ptr.size(), st));
            HIPCHK(c, d_cells.upload(bin_cells.data(), bin_cells.size(), st));
            HIPCHK(c, d_dims.upload(dims, 3, st));
            HIPCHK(c, d_lo.upload(lo, 3, st));
            HIPCHK(c, d_invh.upload(inv_h, 3, st));
            HIPCHK(c, d_locs.upload(locs_colmajor, (size_t)n_locs * M, st));
            HIPCHK(c, d_out.alloc((size_t)n_locs));
            HIPCHK(c, d_vals.alloc((size_t)n_locs * hs.nb));
            AsmArgs a = asm_args(c);
            const double tol = 1e-12;
            const dim3 grid(g1(n_locs)), block(256);
#define EVAL_GO(MM, RR)                                                                                                  \
    hipLaunchKernelGGL((k_eval_pointwise<MM, RR>), grid, block, 0, st, a, n_locs, d_locs.p, d_lo.p, d_invh.p, d_dims.p, d_ptr.p, \
                       d_cells.p, c->cell_i2e.p, tol, d_out.p, d_vals.p)
            if (M == 2 && hs.order == 1) EVAL_GO(2, 1);
            else if (M == 2) EVAL_GO(2, 2);
            else if (hs.order == 1) EVAL_GO(3, 1);
            else EVAL_GO(3, 2);
#undef EVAL_GO
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(cell_ids, d_out.p, sizeof(int32_t) * (size_t)n_locs, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(values, d_vals.p, sizeof(double) * (size_t)n_locs * hs.nb, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            d_ptr.release(), d_cells.release(), d_dims.release(), d_out.release(), d_locs.release(), d_lo.release(), d_invh.release(),
              d_vals.release();
        }
    }
    return FDAPDE_OK;
}

// ingredients of areal_evaluation::eval (basis/lagrangian_basis.h:238-283): per-cell measure and integrals of the local basis
int fdapde_cell_integrals(fdapde_ctx* c, double* measure, double* psi_int) {
    if (!c || !measure || !psi_int) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    DBuf<double> d_m, d_p;
    HIPCHK(c, d_m.alloc((size_t)hs.n_cells));
    HIPCHK(c, d_p.alloc((size_t)hs.n_cells * hs.nb));
    AsmArgs a = asm_args(c);
    if (hs.M == 2)
        hipLaunchKernelGGL(k_cell_integrals<2>, dim3(g1(hs.n_cells)), dim3(256), 0, c->stream, a, hs.nb, hs.nq, c->cell_i2e.p, d_m.p, d_p.p);
    else
        hipLaunchKernelGGL(k_cell_integrals<3>, dim3(g1(hs.n_cells)), dim3(256), 0, c->stream, a, hs.nb, hs.nq, c->cell_i2e.p, d_m.p, d_p.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(measure, d_m.p, sizeof(double) * (size_t)hs.n_cells, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(psi_int, d_p.p, sizeof(double) * (size_t)hs.n_cells * hs.nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    d_m.release(), d_p.release();
    return FDAPDE_OK;
}

int fdapde_matrix_values(fdapde_ctx* c, int32_t which, double* values) {
    if (!c || !values || which < 0 || which > 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->assembled[which]) return fail(c, FDAPDE_ENOTINIT, "matrix not assembled");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int zero_rows = (which == FDAPDE_MAT_STIFF && c->dirichlet_applied) ? 1 : 0;
    hipLaunchKernelGGL(k_export_values, dim3(g1(hs.n_dofs * 16)), dim3(256), 0, c->stream, hs.n_dofs, c->rowptr.p, c->colidx.p,
                       c->vals[which].p, c->slot_i2e.p, c->bnd.p, zero_rows, c->tmp_v.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(values, c->tmp_v.p, sizeof(double) * (size_t)hs.nnz, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDAPDE_OK;
}

// lump(stiff() | mass()) (fdaPDE/linear_algebra/lumping.h:30-41): the diagonal of the row-sum lumped matrix, reference numbering
int fdapde_lump(fdapde_ctx* c, int32_t which, double* diag) {
    if (!c || !diag || which < 0 || which > 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->assembled[which]) return fail(c, FDAPDE_ENOTINIT, "matrix not assembled");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    hipLaunchKernelGGL(k_row_sums, dim3(g1(hs.n_dofs * 16)), dim3(256), 0, c->stream, hs.n_dofs, c->rowptr.p, c->vals[which].p, c->tmp_i.p);
    hipLaunchKernelGGL(k_scatter_f64, dim3(g1(hs.n_dofs)), dim3(256), 0, c->stream, hs.n_dofs, c->dof_i2e.p, c->tmp_i.p, c->tmp_e.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(diag, c->tmp_e.p, sizeof(double) * (size_t)hs.n_dofs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDAPDE_OK;
}

int fdapde_force(fdapde_ctx* c, double* force) {
    if (!c || !force) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->force_ready) return fail(c, FDAPDE_ENOTINIT, "force not assembled");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int cols = c->fq_cols > 0 ? c->fq_cols : 1;
    for (int col = 0; col < cols; ++col) {
        HIPCHK(c, hipMemcpyAsync(c->tmp_i.p, c->force.p + (size_t)col * hs.n_dofs, sizeof(double) * (size_t)hs.n_dofs,
                                 hipMemcpyDeviceToDevice, c->stream));
        if (col == 0 && c->dirichlet_applied)
            hipLaunchKernelGGL(k_force_bc, dim3(g1(hs.n_dofs)), dim3(256), 0, c->stream, hs.n_dofs, c->bnd.p, c->g.p, c->tmp_i.p);
        hipLaunchKernelGGL(k_scatter_f64, dim3(g1(hs.n_dofs)), dim3(256), 0, c->stream, hs.n_dofs, c->dof_i2e.p, c->tmp_i.p, c->tmp_e.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(force + (size_t)col * hs.n_dofs, c->tmp_e.p, sizeof(double) * (size_t)hs.n_dofs,
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return FDAPDE_OK;
}

int fdapde_solution(fdapde_ctx* c, double* solution) {
    if (!c || !solution) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->solved) return fail(c, FDAPDE_ENOTINIT, "no solution: call fdapde_solve first");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    hipLaunchKernelGGL(k_scatter_f64, dim3(g1(hs.n_dofs)), dim3(256), 0, c->stream, hs.n_dofs, c->dof_i2e.p, c->u.p, c->tmp_e.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(solution, c->tmp_e.p, sizeof(double) * (size_t)hs.n_dofs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDAPDE_OK;
}

int fdapde_info_get(const fdapde_ctx* c, fdapde_info* info) {
    if (!c || !info) return FDAPDE_EINVAL;
    *info = c->info;
    return FDAPDE_OK;
}

int fdapde_spmv(fdapde_ctx* c, int32_t which, const double* x, double* y) {
    if (!c || !x || !y || which < 0 || which > 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->assembled[which]) return fail(c, FDAPDE_ENOTINIT, "matrix not assembled");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const size_t bytes = sizeof(double) * (size_t)hs.n_dofs;
    HIPCHK(c, hipMemcpyAsync(c->tmp_e.p, x, bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_f64, dim3(g1(hs.n_dofs)), dim3(256), 0, c->stream, hs.n_dofs, c->dof_i2e.p, c->tmp_e.p, c->tmp_i.p);
    launch_spmv(c, c->vals[which].p, c->tmp_i.p, c->t.p, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(k_scatter_f64, dim3(g1(hs.n_dofs)), dim3(256), 0, c->stream, hs.n_dofs, c->dof_i2e.p, c->t.p, c->tmp_e.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(y, c->tmp_e.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDAPDE_OK;
}

int fdapde_bench_spmv(fdapde_ctx* c, int32_t reps, double* avg_ms, double* algorithmic_bytes) {
    if (!c || reps < 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->assembled[0]) return fail(c, FDAPDE_ENOTINIT, "matrix not assembled");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    // the launch timed here is the one inside CG: scaled matrix stream, fused p.Ap partials
    const double* A = (c->solved && c->scaled_owner == fdapde_ctx::kScaledSolve) ? c->sval.p : c->vals[0].p;
    hipLaunchKernelGGL(k_fill_f64, dim3(g1(hs.n_dofs)), dim3(256), 0, c->stream, hs.n_dofs, 1.0, c->tmp_i.p);
    for (int i = 0; i < 3; ++i) launch_spmv(c, A, c->tmp_i.p, c->t.p, c->tmp_i.p, c->part_a.p, nullptr);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; ++i) launch_spmv(c, A, c->tmp_i.p, c->t.p, c->tmp_i.p, c->part_a.p, nullptr);
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (std::getenv("FDAPDE_READ_PROBE")) {   // diagnostic: pure read stream of the matrix arrays, same stream, HIP events
        const int64_t n16 = ((int64_t)hs.nnz * 8) / 16;
        for (int grid : {1024, 2048, 4096, 8192}) {
            hipLaunchKernelGGL(k_read_probe, dim3(grid), dim3(256), 0, c->stream, reinterpret_cast<const double2*>(A), n16, c->tmp_i.p);
            HIPCHK(c, hipEventRecord(c->ev0, c->stream));
            for (int i = 0; i < 20; ++i)
                hipLaunchKernelGGL(k_read_probe, dim3(grid), dim3(256), 0, c->stream, reinterpret_cast<const double2*>(A), n16, c->tmp_i.p);
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            float pm = 0;
            HIPCHK(c, hipEventElapsedTime(&pm, c->ev0, c->ev1));
            std::fprintf(stderr, "read_probe grid=%d: %.1f MB in %.2f us -> %.0f GB/s\n", grid, n16 * 16 / 1e6, pm / 20 * 1e3,
                         n16 * 16 / (pm / 20 * 1e-3) / 1e9);
        }
    }
    if (std::getenv("FDAPDE_STREAM_PROBE")) {   // diagnostic: the matrix arrays streamed once, nothing else
        const int64_t n2 = (int64_t)hs.nnz / 2;
        for (int grid : {2048, 8192}) {
            hipLaunchKernelGGL(k_stream_probe, dim3(grid), dim3(256), 0, c->stream, reinterpret_cast<const double2*>(A),
                               reinterpret_cast<const int2*>(c->colidx.p), n2, c->tmp_i.p);
            HIPCHK(c, hipEventRecord(c->ev0, c->stream));
            for (int i = 0; i < 50; ++i)
                hipLaunchKernelGGL(k_stream_probe, dim3(grid), dim3(256), 0, c->stream, reinterpret_cast<const double2*>(A),
                                   reinterpret_cast<const int2*>(c->colidx.p), n2, c->tmp_i.p);
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            float pm = 0;
            HIPCHK(c, hipEventElapsedTime(&pm, c->ev0, c->ev1));
            std::fprintf(stderr, "stream_probe grid=%d: %.1f MB in %.2f us -> %.0f GB/s\n", grid, n2 * 24 / 1e6, pm / 50 * 1e3,
                         n2 * 24 / (pm / 50 * 1e-3) / 1e9);
            HIPCHK(c, hipEventRecord(c->ev0, c->stream));
            for (int i = 0; i < 50; ++i)
                hipLaunchKernelGGL(k_stream_probe_unaligned, dim3(grid), dim3(256), 0, c->stream, A,
                                   reinterpret_cast<const int2*>(c->colidx.p), n2, c->tmp_i.p);
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            HIPCHK(c, hipEventElapsedTime(&pm, c->ev0, c->ev1));
            std::fprintf(stderr, "stream_probe_unaligned grid=%d: %.2f us -> %.0f GB/s\n", grid, pm / 50 * 1e3,
                         n2 * 24 / (pm / 50 * 1e-3) / 1e9);
            HIPCHK(c, hipEventRecord(c->ev0, c->stream));
            for (int i = 0; i < 50; ++i)
                hipLaunchKernelGGL(k_stream_probe_w, dim3(grid), dim3(256), 0, c->stream, reinterpret_cast<const double2*>(A),
                                   reinterpret_cast<const int2*>(c->colidx.p), n2, c->tmp_v.p);
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            HIPCHK(c, hipEventElapsedTime(&pm, c->ev0, c->ev1));
            std::fprintf(stderr, "stream_probe + %.1f MB of writes grid=%d: %.2f us\n", n2 / 8 * 8 / 1e6, grid, pm / 50 * 1e3);
        }
    }
    if (avg_ms) *avg_ms = (double)ms / reps;
    if (algorithmic_bytes) *algorithmic_bytes = 12.0 * (double)hs.nnz + 4.0 * (double)(hs.n_dofs + 1) + 16.0 * (double)hs.n_dofs;
    return FDAPDE_OK;
}

int fdapde_comm_unique_id(void* out128) {
    if (!out128) return FDAPDE_EINVAL;
    std::string err;
    if (!g_rccl.load(err)) return FDAPDE_ERCCL;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return FDAPDE_ERCCL;
    std::memcpy(out128, &id, sizeof id);
    return FDAPDE_OK;
}

int fdapde_comm_init(fdapde_ctx* c, int32_t world, int32_t rank, const void* unique_id128) {
    if (!c || !unique_id128 || world < 1 || rank < 0 || rank >= world) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!g_rccl.load(c->err)) return FDAPDE_ERCCL;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) (void)g_rccl.CommDestroy(c->comm), c->comm = nullptr;
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof id);
    {   // RCCL reads the runtime's last-error slot while it sets up: an error some earlier, unrelated call of this process left there
        // (HIP keeps it until somebody asks) would be reported as RCCL's own
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess && std::getenv("FDAPDE_DEBUG_SETUP")) std::fprintf(stderr, "comm_init: stale HIP error cleared: %s\n", hipGetErrorString(stale));
    }
    RCCLCHK(c, g_rccl.CommInitRank(&c->comm, world, id, rank));
    c->world = world, c->rank = rank, c->ar_fn = nullptr;
    return FDAPDE_OK;
}

// sum (op 0) or max (op 1) of n host doubles over the ranks of the context's communicator, in place: the barrier / timing reductions of a
// multi-process driver that holds no other collective library (bench.py's ranks load this library and nothing else that touches the GPU)
int fdapde_comm_allreduce(fdapde_ctx* c, double* host_inout, int32_t n, int32_t op) {
    if (!c || !host_inout || n < 1 || (op != 0 && op != 1)) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->comm && !c->ar_fn) return fail(c, FDAPDE_ENOTINIT, "call fdapde_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->ar_fn) {
        if (op != 0 && c->world > 1) return fail(c, FDAPDE_EUNSUPPORTED, "the host-staged transport only sums");
        if (c->world > 1 && c->ar_fn(c->ar_user, host_inout, (int64_t)n) != 0) return fail(c, FDAPDE_ERCCL, "all-reduce callback failed");
        return FDAPDE_OK;
    }
    HIPCHK(c, c->ar_dev.alloc((size_t)n));
    HIPCHK(c, hipMemcpyAsync(c->ar_dev.p, host_inout, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(c, g_rccl.AllReduce(c->ar_dev.p, c->ar_dev.p, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(host_inout, c->ar_dev.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDAPDE_OK;
}

// which RCCL the library bound itself to (diagnostics; empty before the first communicator call)
const char* fdapde_comm_library(void) { return g_rccl.path.c_str(); }

int fdapde_comm_init_callback(fdapde_ctx* c, int32_t world, int32_t rank, fdapde_allreduce_fn fn, void* user) {
    if (!c || !fn || world < 1 || rank < 0 || rank >= world) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm), c->comm = nullptr;
    c->ar_fn = fn, c->ar_user = user, c->world = world, c->rank = rank;
    return FDAPDE_OK;
}

int fdapde_halo_setup(fdapde_ctx* c, int64_t n_if_global, int64_t n_if_local, const int32_t* local_dof, const int32_t* if_index,
                      const uint8_t* owned) {
    if (!c || n_if_global < 0 || n_if_local < 0 || (n_if_local > 0 && (!local_dof || !if_index)) || !owned) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (!c->comm && !c->ar_fn) return fail(c, FDAPDE_ENOTINIT, "call fdapde_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    std::vector<int32_t> dof_i((size_t)n_if_local), pos((size_t)n_if_local);
    for (int64_t k = 0; k < n_if_local; ++k) {
        if (local_dof[k] < 0 || local_dof[k] >= hs.n_dofs || if_index[k] < 0 || if_index[k] >= n_if_global)
            return fail(c, FDAPDE_EINVAL, "interface map entry out of range");
        dof_i[(size_t)k] = hs.dof_e2i[(size_t)local_dof[k]], pos[(size_t)k] = if_index[k];
    }
    std::vector<uint8_t> own_i((size_t)hs.n_dofs);
    for (int64_t i = 0; i < hs.n_dofs; ++i) own_i[(size_t)i] = owned[hs.dof_i2e[(size_t)i]] ? 1 : 0;
    std::vector<int32_t> inv((size_t)(n_if_global > 0 ? n_if_global : 1), -1), slot((size_t)hs.n_dofs + 2, -1);
    for (int64_t k = 0; k < n_if_local; ++k) inv[(size_t)pos[(size_t)k]] = dof_i[(size_t)k], slot[(size_t)dof_i[(size_t)k]] = pos[(size_t)k];
    HIPCHK(c, c->halo_inv.upload(inv.data(), inv.size(), c->stream));
    HIPCHK(c, c->if_slot.upload(slot.data(), slot.size(), c->stream));
    HIPCHK(c, c->halo_dof.upload(dof_i.data(), dof_i.size(), c->stream));
    HIPCHK(c, c->halo_pos.upload(pos.data(), pos.size(), c->stream));
    HIPCHK(c, c->owned.upload(own_i.data(), own_i.size(), c->stream));
    HIPCHK(c, c->hbuf.alloc((size_t)n_if_global + 2));
    HIPCHK(c, c->sbuf.alloc(8));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_if = n_if_global, c->n_loc_if = n_if_local, c->halo_ready = true, c->peer_mode = false;
    return FDAPDE_OK;
}

// Row-distributed multi-GPU form: every DOF of the whole mesh is OWNED by one rank; a rank's sub-mesh holds every cell touching one of its
// DOFs (its own cells + one layer of cells of its neighbours), so that its assembly completes the rows of its DOFs without any exchange.
int fdapde_rowdist_setup(fdapde_ctx* c, const int64_t* dof_key, const int32_t* dof_owner) {
    if (!c || !dof_key || !dof_owner) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (!c->comm && !c->ar_fn) return fail(c, FDAPDE_ENOTINIT, "call fdapde_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    release_rowdist(c);
    const HostSpace& hs = c->hs;
    c->rd.owner_i.resize((size_t)hs.n_dofs), c->rd.key_i.resize((size_t)hs.n_dofs);
    std::vector<uint8_t> own((size_t)hs.n_dofs);
    for (int64_t i = 0; i < hs.n_dofs; ++i) {
        const int32_t e = hs.dof_i2e[(size_t)i];
        if (dof_owner[e] < 0 || dof_owner[e] >= c->world) return fail(c, FDAPDE_EINVAL, "fdapde_rowdist_setup: owner out of range");
        c->rd.owner_i[(size_t)i] = dof_owner[e], c->rd.key_i[(size_t)i] = dof_key[e], own[(size_t)i] = dof_owner[e] == c->rank ? 1 : 0;
    }
    HIPCHK(c, c->rd.owned.upload(own.data(), own.size(), c->stream));
    HIPCHK(c, c->sbuf.alloc(8));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->rd.ready = true, c->halo_ready = false;
    return FDAPDE_OK;
}

int fdapde_comm_set_exchange_callback(fdapde_ctx* c, fdapde_exchange_fn fn, void* user) {
    if (!c || !fn) return FDAPDE_EINVAL;
    c->xchg_fn = fn, c->xchg_user = user;
    return FDAPDE_OK;
}

int fdapde_halo_setup_peers(fdapde_ctx* c, int32_t n_peers, const int32_t* peer_rank, const int64_t* peer_off, const int32_t* peer_dof,
                            const uint8_t* owned) {
    if (!c || n_peers < 0 || !owned || (n_peers > 0 && (!peer_rank || !peer_off || !peer_dof))) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (!c->comm && !c->ar_fn) return fail(c, FDAPDE_ENOTINIT, "call fdapde_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int64_t n_send = n_peers > 0 ? peer_off[n_peers] : 0;
    for (int q = 0; q < n_peers; ++q)
        if (peer_rank[q] < 0 || peer_rank[q] >= c->world || peer_rank[q] == c->rank || (q > 0 && peer_rank[q] <= peer_rank[q - 1]) ||
            peer_off[q + 1] < peer_off[q] || (q == 0 && peer_off[0] != 0))
            return fail(c, FDAPDE_EINVAL, "peer list: ranks must be ascending, without this rank, offsets non-decreasing from 0");
    std::vector<int32_t> send_dof((size_t)(n_send > 0 ? n_send : 1), 0);
    std::vector<int32_t> k_of((size_t)hs.n_dofs, -1), if_dof;   // internal DOF -> local interface index
    for (int64_t j = 0; j < n_send; ++j) {
        if (peer_dof[j] < 0 || peer_dof[j] >= hs.n_dofs) return fail(c, FDAPDE_EINVAL, "peer list: DOF id out of range");
        const int32_t d = hs.dof_e2i[(size_t)peer_dof[j]];
        send_dof[(size_t)j] = d;
        if (k_of[(size_t)d] < 0) k_of[(size_t)d] = (int32_t)if_dof.size(), if_dof.push_back(d);
    }
    const int64_t n_loc = (int64_t)if_dof.size();
    // contributions of every local interface DOF in ascending rank order: the peers are ascending, this rank's own goes where its
    // rank falls among them
    std::vector<int32_t> cnt((size_t)n_loc + 1, 0);
    for (int64_t j = 0; j < n_send; ++j) ++cnt[(size_t)k_of[(size_t)send_dof[(size_t)j]] + 1];
    for (int64_t k = 0; k < n_loc; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k] + 1;   // + 1: the own contribution
    std::vector<int32_t> src_off(cnt), src((size_t)(n_send + n_loc > 0 ? n_send + n_loc : 1), 0), fill_at(cnt.begin(), cnt.end() - 1);
    std::vector<uint8_t> own_in((size_t)n_loc, 0);
    std::vector<int32_t> last_peer((size_t)n_loc, -1);
    for (int q = 0; q < n_peers; ++q) {
        for (int64_t j = peer_off[q]; j < peer_off[q + 1]; ++j) {
            const int32_t k = k_of[(size_t)send_dof[(size_t)j]];
            if (last_peer[(size_t)k] == q) return fail(c, FDAPDE_EINVAL, "peer list: a DOF is listed twice for one peer");
            last_peer[(size_t)k] = q;
            if (peer_rank[q] > c->rank && !own_in[(size_t)k]) src[(size_t)fill_at[(size_t)k]++] = -1, own_in[(size_t)k] = 1;
            src[(size_t)fill_at[(size_t)k]++] = (int32_t)j;
        }
    }
    for (int64_t k = 0; k < n_loc; ++k)
        if (!own_in[(size_t)k]) src[(size_t)fill_at[(size_t)k]++] = -1;
    std::vector<uint8_t> own_i((size_t)hs.n_dofs);
    for (int64_t i = 0; i < hs.n_dofs; ++i) own_i[(size_t)i] = owned[hs.dof_i2e[(size_t)i]] ? 1 : 0;
    std::vector<int32_t> slot((size_t)hs.n_dofs + 2, -1), pos((size_t)(n_loc > 0 ? n_loc : 1), 0);
    for (int64_t k = 0; k < n_loc; ++k) slot[(size_t)if_dof[(size_t)k]] = (int32_t)k, pos[(size_t)k] = (int32_t)k;
    if (if_dof.empty()) if_dof.push_back(0);
    hipStream_t st = c->stream;
    HIPCHK(c, c->peer_send_dof.upload(send_dof.data(), send_dof.size(), st));
    HIPCHK(c, c->peer_src_off.upload(src_off.data(), src_off.size(), st));
    HIPCHK(c, c->peer_src.upload(src.data(), src.size(), st));
    HIPCHK(c, c->peer_sendbuf.alloc((size_t)(n_send > 0 ? n_send : 1)));
    HIPCHK(c, c->peer_recvbuf.alloc((size_t)(n_send > 0 ? n_send : 1)));
    HIPCHK(c, c->if_slot.upload(slot.data(), slot.size(), st));
    HIPCHK(c, c->halo_dof.upload(if_dof.data(), if_dof.size(), st));
    HIPCHK(c, c->halo_pos.upload(pos.data(), pos.size(), st));
    HIPCHK(c, c->owned.upload(own_i.data(), own_i.size(), st));
    HIPCHK(c, c->hbuf.alloc((size_t)n_loc + 2));
    HIPCHK(c, c->sbuf.alloc(8));
    HIPCHK(c, hipStreamSynchronize(st));
    c->peer_rank.assign(peer_rank, peer_rank + n_peers);
    c->peer_off.assign(1, 0);
    if (n_peers > 0) c->peer_off.assign(peer_off, peer_off + n_peers + 1);
    c->n_if = n_loc, c->n_loc_if = n_loc, c->peer_mode = true, c->halo_ready = true;
    return FDAPDE_OK;
}

int fdapde_tune(fdapde_ctx* c, const char* key, int32_t value) {
    if (!c || !key) return FDAPDE_EINVAL;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    const std::string k(key);
    drop_graph(c);
    if (k == "spmv_variant" && value >= 0 && value <= 2) c->spmv_variant = value;
    else if (k == "spmv_team" && (value == 2 || value == 4 || value == 8 || value == 16 || value == 32 || value == 64)) {
        if (value != c->spmv_team) c->sp_built[0] = c->sp_built[1] = false, c->sp_cur = -1, c->solved = false, c->scaled_owner = fdapde_ctx::kScaledNone;   // segmented patterns depend on it
        c->spmv_team = value;
    }
    else if (k == "spmv_unroll" && value >= 1 && value <= 8) c->spmv_unroll = value;
    else if (k == "spmv_ablate") c->spmv_ablate = value;
    else if (k == "spmv_c16" && (value == 0 || value == 1)) c->spmv_c16 = value;
    else if (k == "spmv_deep" && (value == 0 || value == 1)) c->spmv_deep = value;
    else if (k == "cgf_split" && (value == 0 || value == 1)) c->cgf_split = value;
    else if (k == "cgf_v" && (value == 1 || value == 2 || value == 4 || value == 8)) c->cgf_v = value;
    else if (k == "use_graph" && (value == 0 || value == 1)) c->use_graph = value;
    else if (k == "cgf_band" && (value == 0 || value == 1)) c->cgf_band = value;
    else if (k == "cgf_nt" && value >= 0 && value <= 15) c->cgf_nt = value;
    else if (k == "cgf_lazy" && (value == 0 || value == 1)) c->cgf_lazy = value;
    else if (k == "multi_rhs" && (value == 0 || value == 1)) c->multi_rhs = value;
    else if (k == "asm_fq_block" && (value == 0 || value == 1)) c->asm_fq_block = value;
    else if (k == "asm_fq_bc" && (value == 0 || value == 1)) c->asm_fq_bc = value, c->fq_bc_ready = c->fq_bc_ready && value;   // (takes effect fully at the next fdapde_set_forcing)
    else if (k == "persist" && (value == 0 || value == 1)) c->persist = value, c->persist_broken = false;
    else if (k == "persist_time" && (value == 0 || value == 1)) c->persist_time = value;
    else if (k == "persist_coop" && (value == 0 || value == 1)) c->persist_coop = value;
    else if (k == "persist_bicg" && (value == 0 || value == 1)) c->persist_bicg = value;
    else if (k == "rowdist_max_wg" && value >= 0) {   // workgroups of this rank's launch (tests: several ranks share one device)
        c->rd.max_wg = value;
        for (auto& L : c->rd.lay) L.tried = L.ok = false;
    } else if (k == "rowdist_share" && value >= 1) {   // that many ranks share this device: an equal share of its CUs each
        c->rd.max_wg = std::max(1, c->n_cu / value);
        for (auto& L : c->rd.lay) L.tried = L.ok = false;
    } else if (k == "rowdist_timeout_first_ms" && value >= 1) c->rd.timeout_first_ms = value;
    else if (k == "persist_timeout_us" && value >= 100 && value <= 10000000) c->persist_timeout_us = value;
    else if (k == "persist_debug_stall" && value >= 0) c->persist_debug_stall = value;   // (tests: forces the hand-off timeout at that iteration)
    else if (k == "persist_retry" && value == 1) c->persist_broken = false, c->persist_retry_in = 0, c->persist_backoff = 8;   // (tests: forget an earlier timeout)
    else if (k == "blocked" && value >= 0 && value <= 2) c->blocked = value;   // 2: also for short-row systems
    else if (k == "persist_gather_waves" && (value == 1 || value == 4)) c->persist_gather_waves = value;
    else if (k == "persist_poll_sleep" && value >= 0 && value <= 3) c->persist_poll_sleep = value;
    else if (k == "persist_balance" && (value == 0 || value == 1)) {   // workgroup boundaries of the persistent CG at equal cost (1) or equal row counts (0)
        c->persist_balance = value;
        for (auto& ps : c->ps) ps.tried = ps.ok = ps.filled = false;
    } else if (k == "persist_sym" && value >= 0 && value <= 2) {   // 0 plain storage, 1 symmetric, 2 symmetric where the plain blocks would stream
        c->persist_sym = value;
        for (auto& ps : c->ps) ps.tried = ps.ok = ps.filled = false;   // the layouts are rebuilt on the next solve
    }
    else if (k == "spmv_ntv" && value >= -1 && value <= 1) c->spmv_ntv = value;
    else if (k == "spmv_bpx" && value >= 1 && value <= 1024) {
        c->spmv_grid = 8 * value;
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, c->part_a.alloc(2 * (size_t)c->spmv_grid));
    } else return fail(c, FDAPDE_EINVAL, "unknown tuning key or value out of range");
    return FDAPDE_OK;
}

void* fdapde_stream(fdapde_ctx* c) { return c ? (void*)c->stream : nullptr; }

int fdapde_synchronize(fdapde_ctx* c) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipDeviceSynchronize());   // (what torch.cuda.synchronize() would do: nothing of this process is left running on the device)
    return FDAPDE_OK;
}

}  // extern "C"
