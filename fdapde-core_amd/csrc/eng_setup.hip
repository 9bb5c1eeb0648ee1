// eng_setup.hip -- function space set-up behind fdapde_dofs_build / fdapde_topology_* / the numbering getters: adoption of the device-built
// index structures (dev_setup.hip), uploads of a host-built space, lazily fetched host mirrors, FDAPDE_SETUP_CHECK comparisons.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>

#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include "context.h"
#include "engine.h"
#include "kernels.h"

namespace fdapde_engine {

// ---- code objects of the set-up units (dev_setup / dev_persist / dev_topology: 25 MB of device code, mostly the radix sorts' instantiations;
//      HIP loads a unit's code object when one of its kernels is first looked up: ~30 ms inside the first fdapde_dofs_build of a process before).
//      The first context of a process starts the loading on a helper thread; the first build waits for it.
namespace {
std::once_flag g_preload_once;
std::mutex g_preload_mx;
std::condition_variable g_preload_cv;
int g_preload_done = 0;        // units loaded so far, in the order dev_setup (1), dev_persist (2), dev_topology (3); 3 too when nothing was started
bool g_preload_started = false;
std::thread g_preload_thread;
struct PreloadJoiner {
    ~PreloadJoiner() {   // (a process that ends before its first build: the thread is joined, never left running)
        if (g_preload_thread.joinable()) g_preload_thread.join();
    }
} g_preload_joiner;
}   // namespace

void preload_setup_async(int device) {
    if (std::getenv("FDAPDE_NO_PRELOAD")) return;
    std::call_once(g_preload_once, [device] {
        std::lock_guard<std::mutex> lk(g_preload_mx);
        g_preload_started = true;
        g_preload_thread = std::thread([device] {
            const bool dbg = std::getenv("FDAPDE_DEBUG_TIMING") != nullptr;
            auto t0 = std::chrono::steady_clock::now();
            auto done = [&](int unit, const char* name) {
                {
                    std::lock_guard<std::mutex> lk2(g_preload_mx);
                    g_preload_done = unit;
                }
                g_preload_cv.notify_all();
                if (dbg) {
                    const auto t1 = std::chrono::steady_clock::now();
                    std::fprintf(stderr, "[timing] preload thread: %-16s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
                    t0 = t1;
                }
            };
            const bool ok = hipSetDevice(device) == hipSuccess;
            if (ok) dev_setup_preload();
            done(1, "dev_setup");
            if (ok) dev_persist_preload();
            done(2, "dev_persist");
            if (ok) dev_topology_preload();
            done(3, "dev_topology");
        });
    });
}
// waits until the first `units` set-up units are loaded: 1 = dev_setup (fdapde_dofs_build of an order-1 space), 2 = + dev_persist (the single-launch
// solver's layout), 3 = + dev_topology (order-2 spaces, fdapde_topology_build)
void preload_wait(int units) {
    std::unique_lock<std::mutex> lk(g_preload_mx);
    if (!g_preload_started) return;
    g_preload_cv.wait(lk, [units] { return g_preload_done >= units; });
}

// big host-side index arrays of a device-built space, fetched the first time host code needs them (the persistent layout and the
// solver patterns read rowptr_i / colidx_i; the colouring and the partitioned assembly cdofs_i; point location cverts_i / vcoords_i;
// fdapde_pattern_get the reference pattern)
int ensure_host(fdapde_ctx* c, int what) {
    if (!c->dev_built) return FDAPDE_OK;
    HostSpace& hs = c->hs;
    hipStream_t st = c->stream;
    HIPCHK(c, hipSetDevice(c->device));
    if ((what & (kHostPerm | kHostPattern)) && hs.dof_i2e.empty()) {
        const size_t nd = (size_t)hs.n_dofs, nc = (size_t)hs.n_cells;
        hs.dof_i2e.resize(nd), hs.dof_e2i.resize(nd), hs.cell_i2e.resize(nc), hs.dof_bnd_i.resize(nd);
        HIPCHK(c, hipMemcpyAsync(hs.dof_i2e.data(), c->dof_i2e.p, sizeof(int32_t) * nd, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(hs.dof_e2i.data(), c->dof_e2i.p, sizeof(int32_t) * nd, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(hs.cell_i2e.data(), c->cell_i2e.p, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(hs.dof_bnd_i.data(), c->bnd.p, nd, hipMemcpyDeviceToHost, st));
    }
    if ((what & kHostPattern) && hs.rowptr_i.empty()) {
        hs.rowptr_i.resize(c->rowptr.n);
        HIPCHK(c, hipMemcpyAsync(hs.rowptr_i.data(), c->rowptr.p, sizeof(int32_t) * c->rowptr.n, hipMemcpyDeviceToHost, st));
    }
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
        if (hs.rb_row.size() > 2 && hs.rb_row != ref.rb_row) std::fprintf(stderr, "set-up check rb_row: MISMATCH\n"), ++bad;   // (built for FDAPDE_SPMV=stream only)
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
        auto rs_own = std::make_unique<DevRefTensorsSym>();
        DevRefTensorsSym& rs = *rs_own;
        std::memset(&rs, 0, sizeof rs);
        for (int i = 0; i < nn; ++i) {
            for (int k = 0; k < 3; ++k) rs.kdiag[k * nn + i] = rt.ktab[(k * 3 + k) * nn + i], rs.ctab[k * nn + i] = rt.ctab[k * nn + i];
            rs.ksum[0 * nn + i] = rt.ktab[(0 * 3 + 1) * nn + i] + rt.ktab[(1 * 3 + 0) * nn + i];   // pairs (k, l), k < l, at index k + l - 1
            rs.ksum[1 * nn + i] = rt.ktab[(0 * 3 + 2) * nn + i] + rt.ktab[(2 * 3 + 0) * nn + i];
            rs.ksum[2 * nn + i] = rt.ktab[(1 * 3 + 2) * nn + i] + rt.ktab[(2 * 3 + 1) * nn + i];
        }
        HIPCHK(c, c->reftab_sym.upload(&rs, 1, st));
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
    HIPCHK(c, c->ctl.alloc(8));   // [0] stop, [1] iterations, [2] breakdown, [3] positive-diagonal flag / launch gave up, [4] deferred positive-diagonal flag
    HIPCHK(c, hipMemsetAsync(c->ctl.p, 0, 8 * sizeof(int32_t), st));
    HIPCHK(c, hipMemsetAsync(c->force.p, 0, n * sizeof(double), st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->dev_ready = true;
    return FDAPDE_OK;
}

int e_dofs_build(fdapde_ctx* c, int order, int64_t* n_dofs) {
    if (!c) return FDAPDE_EINVAL;
    auto t0 = std::chrono::steady_clock::now();
    c->space_ready = c->dev_ready = c->colour_ready = c->fq_blk_ready = c->part_ready = c->wave_ready = false;
    c->assembled[0] = c->assembled[1] = c->force_ready = c->solved = c->dirichlet_applied = false;
    c->op.clear(), c->coef_of_op = false, c->fq_i.clear(), c->fq_cols = 0, c->g_i.clear(), c->have_g = false;
    pmg_release(c);   // (the coarse level of the space that is being replaced)
    c->matrix_dirty = true;
    c->halo_ready = false, c->lin_ready = false, c->sp_built[0] = c->sp_built[1] = false, c->sp_cur = -1;
    release_rowdist(c);   // (keys / owners / layouts of the row-distributed form belong to the space that is being replaced)
    c->eval_grid.release();   // (the point-location grid of the mesh before)
    c->asm_max_visits = -1;
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
    DebugClock clk;
    int rc = host_build_space(hs, order, c->err, on_device ? 2 : 0);
    if (rc) return rc;
    clk.mark("dofs_build: host_build_space");
    rc = build_basis_tables(hs.M, order, &c->tb);
    if (rc) return fail(c, rc, "basis tables");
    if (on_device) {
        HIPCHK(c, hipSetDevice(c->device));
        preload_wait(order == 1 ? 1 : 3);   // (the set-up units' code objects: loading started with the process's first context)
        clk.mark("dofs_build: preload_wait");
        DBuf<uint8_t> d_bnd;
        c->dofs_e.release(), c->coords_e.release();
        if (!c->mesh_on_dev) {   // (a mesh that came in while the context had no device copy of it)
            HIPCHK(c, c->mesh_nodes.upload(hs.nodes.data(), hs.nodes.size(), c->stream));
            HIPCHK(c, c->mesh_cells.upload(hs.cells.data(), hs.cells.size(), c->stream));
            HIPCHK(c, c->mesh_nbnd.upload(hs.node_bnd.data(), hs.node_bnd.size(), c->stream));
            c->mesh_on_dev = true;
        }
        DBuf<double>& d_nodes = c->mesh_nodes;
        DBuf<int32_t>& d_cells = c->mesh_cells;
        DBuf<uint8_t>& d_nbnd = c->mesh_nbnd;
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
        d_bnd.release();
        if (rc) return rc;
        clk.mark("dofs_build: dev_build_space");
        if (std::getenv("FDAPDE_SETUP_CHECK")) {
            rc = check_dev_space(c, ds, order);
            if (rc) {
                dev_space_release(&ds);
                return rc;
            }
        }
        rc = adopt_dev_space(c, ds);
        if (rc) return rc;
        clk.mark("dofs_build: adopt");
    }
    c->space_ready = true;
    if (n_dofs) *n_dofs = c->hs.n_dofs;
    if (c->has_device) {
        HIPCHK(c, hipSetDevice(c->device));
        rc = upload_space(c);
        if (rc) return rc;
        clk.mark("dofs_build: upload_space");
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
int e_topology_build(fdapde_ctx* c, int64_t* n_facets, int64_t* n_edges) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    const HostSpace& hs = c->hs;
    if (hs.n_cells < 1) return fail(c, FDAPDE_ENOTINIT, "call fdapde_mesh_upload first");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->topo_ready) {
        preload_wait(3);
        if (!c->mesh_on_dev) {
            HIPCHK(c, c->mesh_nodes.upload(hs.nodes.data(), hs.nodes.size(), c->stream));
            HIPCHK(c, c->mesh_cells.upload(hs.cells.data(), hs.cells.size(), c->stream));
            HIPCHK(c, c->mesh_nbnd.upload(hs.node_bnd.data(), hs.node_bnd.size(), c->stream));
            c->mesh_on_dev = true;
        }
        const int rc = dev_build_topology(hs.M, hs.n_nodes, hs.n_cells, c->mesh_cells.p, c->mesh_nbnd.p, c->stream, &c->topo, c->err);
        if (rc) return rc;
        c->topo_ready = true;
    }
    if (n_facets) *n_facets = c->topo.n_facets;
    if (n_edges) *n_edges = c->topo.n_edges;
    return FDAPDE_OK;
}

int e_topology_get(fdapde_ctx* c, int32_t* neighbors, int32_t* cell_facets, int32_t* facet_nodes, int32_t* facet_cells,
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

int e_dofs_set_boundary(fdapde_ctx* c, const uint8_t* bnd) {
    if (!c || !bnd) return FDAPDE_EINVAL;
    HostSpace& hs = c->hs;
    if (hs.n_dofs == 0) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    pmg_release(c);   // (the coarse level of the two-level solver carries the boundary mask it was built with)
    if (c->dev_built) {
        HIPCHK(c, hipSetDevice(c->device));
        if (int rc = ensure_host(c, kHostPerm)) return rc;
    }
    for (int64_t i = 0; i < hs.n_dofs; ++i) hs.dof_bnd[(size_t)i] = bnd[i] ? 1 : 0;
    for (int64_t i = 0; i < hs.n_dofs; ++i) hs.dof_bnd_i[(size_t)i] = hs.dof_bnd[(size_t)hs.dof_i2e[(size_t)i]];
    c->sp_built[1] = false;   // the compact solver pattern drops Dirichlet rows / columns
    c->ps[1].tried = c->ps[1].ok = false, c->bk[1].tried = c->bk[1].ok = false, c->bk_cur = -1;
    c->rd.lay[1].tried = c->rd.lay[1].ok = false;   // (row-distributed form: rebuilt -- collectively -- by the next solve)
    drop_graph(c);
    c->solved = false, c->scaled_owner = fdapde_ctx::kScaledNone;
    if (c->dev_ready) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, c->bnd.upload(hs.dof_bnd_i.data(), hs.dof_bnd_i.size(), c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return FDAPDE_OK;
}

int e_dofs_get(const fdapde_ctx* c, int32_t* dofs, uint8_t* bnd, double* coords) {
    if (!c || !c->space_ready) return FDAPDE_ENOTINIT;
    if (dofs || coords)
        if (int rc = ensure_host(const_cast<fdapde_ctx*>(c), kHostDofs)) return rc;   // a device-built space keeps them on the device until asked
    if (dofs) std::memcpy(dofs, c->hs.dofs.data(), sizeof(int32_t) * c->hs.dofs.size());
    if (bnd) std::memcpy(bnd, c->hs.dof_bnd.data(), c->hs.dof_bnd.size());
    if (coords) std::memcpy(coords, c->hs.dof_coords.data(), sizeof(double) * c->hs.dof_coords.size());
    return FDAPDE_OK;
}

int e_pattern_get(const fdapde_ctx* c, int32_t* rowptr, int32_t* colidx) {
    if (!c || !c->space_ready) return FDAPDE_ENOTINIT;
    if (int rc = ensure_host(const_cast<fdapde_ctx*>(c), kHostRefPattern)) return rc;   // a device-built space keeps it on the device until asked
    if (rowptr) std::memcpy(rowptr, c->hs.rowptr_e.data(), sizeof(int32_t) * c->hs.rowptr_e.size());
    if (colidx) std::memcpy(colidx, c->hs.colidx_e.data(), sizeof(int32_t) * c->hs.colidx_e.size());
    return FDAPDE_OK;
}


}   // namespace fdapde_engine
