// capi.hip -- the extern "C" shim of include/fdapde_hip.h: context lifetime, argument guards, forwarding to the engine units (engine.h),
// status strings, tuning knobs.  No kernel is launched from here.
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

using namespace fdapde_engine;

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
            hipHostMalloc(reinterpret_cast<void**>(&c->h_ctl), 8 * sizeof(int32_t)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&c->h_sc), 16 * sizeof(double)) != hipSuccess) {
            delete c;
            return FDAPDE_EHIP;
        }
        c->device = device, c->has_device = true;
        preload_assembly(), preload_solve(), preload_dist(), preload_persist();   // (first context of a process: the units' code objects)
        preload_setup_async(device);   // ... those of the set-up units on a helper thread, joined by the first fdapde_dofs_build
        if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->n_cu = 0;
    }
    *out = c;
    return FDAPDE_OK;
}

void fdapde_ctx_destroy(fdapde_ctx* c) {
    if (!c) return;
    if (c->group) fdapde_engine::g_destroy(c);   // (the rank contexts and their threads first)
    fdapde_engine::pmg_release(c);               // (... and the coarse level's context)
    if (c->has_device) {
        (void)hipSetDevice(c->device);
        fdapde_engine::partition_free(c);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        drop_graph(c);
        for (DBuf<int32_t>* b : {&c->cverts, &c->cdofs, &c->adj, &c->rowptr, &c->colidx, &c->diag, &c->slot_i2e, &c->dof_i2e,
                                 &c->dof_e2i, &c->cell_i2e, &c->rb_row, &c->colour_cells, &c->ctl, &c->rowptr_e, &c->colidx_e, &c->dofs_e})
            b->release();
        for (DBuf<double>* b : {&c->vcoords, &c->vals[0], &c->vals[1], &c->force, &c->fq, &c->g, &c->sval, &c->scale, &c->gt,
                                &c->x, &c->r, &c->p, &c->y, &c->s, &c->t, &c->r0, &c->u, &c->part_a, &c->part_b, &c->sc,
                                &c->tmp_e, &c->tmp_i, &c->tmp_v, &c->lin_rhs, &c->cols_b, &c->cols_r, &c->cols_x, &c->cols_sc, &c->cols_part})
            b->release();
        c->cols_ctl.release(), c->eval_grid.release(), c->eval_locs.release(), c->eval_vals.release(), c->eval_out.release();
        for (auto& b : c->coef) b.release();
        c->slotw.release(), c->sl_off.release(), c->lane_row.release(), c->fq_blk.release(), c->bnd.release(), c->tables.release(), c->reftab.release(), c->reftab_sym.release(), c->lin_sq.release();
        c->bc_off.release(), c->bn_off.release(), c->bc_cell.release(), c->bn_node.release(), c->bc_vert.release();
        c->halo_dof.release(), c->halo_pos.release(), c->owned.release(), c->hbuf.release(), c->sbuf.release();
        c->halo_inv.release(), c->if_slot.release();
        c->peer_send_dof.release(), c->peer_src_off.release(), c->peer_src.release(), c->peer_sendbuf.release(), c->peer_recvbuf.release();
        c->xd.remote[0].release(), c->xd.remote[1].release(), c->xd.slots.release(), c->xd.slot_ptr.release();
        release_rowdist(c);
        if (c->comm) (void)g_rccl.CommDestroy(c->comm);
        c->mesh_nodes.release(), c->mesh_cells.release(), c->mesh_nbnd.release();
        c->gm_V.release(), c->gm_b.release(), c->gm_s.release(), c->gm_part.release();
        c->lin_dense.X.release(), c->step_dense.X.release(), c->solve_dense.X.release(), c->dn_b.release(), c->dn_x.release(), c->dn_r.release(), c->dn_cnt.release();
        c->lin_mat.release(), c->stiff_stat.release(), c->ar_dev.release(), c->persist_stats.release(), c->persist_x.release(), c->persist_xs.release(), c->coords_e.release();
        dev_topology_release(&c->topo);
        c->part_cells.release(), c->part_off.release(), c->part_slots.release(), c->wave_slots.release(), c->part_shared.release();
        for (auto& bk : c->bk)
            bk.slot_dof.release(), bk.sl_off.release(), bk.ell_src.release(), bk.imp_off.release(), bk.imp_dof.release(), bk.drop_dof.release(), bk.ell_off.release(),
              bk.ell_code.release(), bk.ell_val.release();
        for (auto& ps : c->ps)
            ps.slot_dof.release(), ps.sl_off.release(), ps.ell_src.release(), ps.exp_off.release(), ps.imp_off.release(),
              ps.imp_pos.release(), ps.ell_off.release(), ps.ell_code.release(), ps.exp_slot.release(), ps.ell_val.release(), ps.board.release(), ps.board_cols.release(), ps.amax.release(), ps.wg_late.release(), ps.ell_col.release();
        for (int v = 0; v < 2; ++v)
            c->sp_rowptr[v].release(), c->sp_colidx[v].release(), c->sp_map[v].release(), c->sp_tbase[v].release(), c->sp_col16[v].release(),
              c->sp_vrow[v].release();
        if (c->h_io) (void)hipHostFree(c->h_io);
        if (c->h_ctl) (void)hipHostFree(c->h_ctl);
        if (c->h_sc) (void)hipHostFree(c->h_sc);
        (void)hipEventDestroy(c->ev0), (void)hipEventDestroy(c->ev1), (void)hipEventDestroy(c->ev_p0), (void)hipEventDestroy(c->ev_p1);
        for (hipEvent_t e : c->ev_spmv) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(c->stream);
    }
    delete c->lin_state;
    delete c;
}

int fdapde_ctx_clone(const fdapde_ctx* src, fdapde_ctx** out) {
    if (!src || !out) return FDAPDE_EINVAL;
    *out = nullptr;
    if (src->group) return fdapde_engine::g_clone(src, out);
    fdapde_ctx* c = nullptr;
    if (int rc = fdapde_ctx_create(src->has_device ? src->device : -1, &c)) return rc;
    const int rc = fdapde_engine::e_ctx_clone(src, c);
    if (rc != FDAPDE_OK) {
        const_cast<fdapde_ctx*>(src)->err = "fdapde_ctx_clone: " + c->err;   // (the half-built clone is not handed out: the text goes to the source)
        fdapde_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return FDAPDE_OK;
}

const char* fdapde_last_error(const fdapde_ctx* c) { return c ? c->err.c_str() : "null context"; }

int fdapde_mesh_upload(fdapde_ctx* c, int M, int N, int64_t n_nodes, const double* nodes, int64_t n_cells,
                       const int32_t* cells, const uint8_t* bnd) {
    if (!c) return FDAPDE_EINVAL;
    c->space_ready = c->dev_ready = c->colour_ready = c->fq_blk_ready = c->fq_bc_ready = c->stiff_stat_valid = false;
    c->assembled[0] = c->assembled[1] = c->force_ready = c->solved = c->dirichlet_applied = false;
    c->op.clear(), c->coef_of_op = false, c->fq_i.clear(), c->fq_cols = 0, c->g_i.clear(), c->have_g = false;
    if (c->topo_ready) dev_topology_release(&c->topo), c->topo_ready = false;
    c->mesh_on_dev = false;
    if (int rc = host_set_mesh(c->hs, M, N, n_nodes, nodes, n_cells, cells, bnd, c->err)) return rc;
    if (c->has_device) {   // the mesh is resident on the device from here on: the function space (any order) and the topology tables are built from it
        const HostSpace& hs = c->hs;
        if (hipSetDevice(c->device) != hipSuccess || c->mesh_nodes.upload(hs.nodes.data(), hs.nodes.size(), c->stream) != hipSuccess ||
            c->mesh_cells.upload(hs.cells.data(), hs.cells.size(), c->stream) != hipSuccess ||
            c->mesh_nbnd.upload(hs.node_bnd.data(), hs.node_bnd.size(), c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            c->mesh_nodes.release(), c->mesh_cells.release(), c->mesh_nbnd.release();
            c->err = "fdapde_mesh_upload: copying the mesh to the device failed";
            return FDAPDE_EHIP;
        }
        c->mesh_on_dev = true;
    }
    fdapde_engine::partition_free(c);   // (a partition of the previous mesh)
    if (c->group) fdapde_engine::g_mesh_changed(c);
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

// ---- everything else forwards to the engine units (the guards above the call are the shim's; the units check call order and ranges)
int fdapde_dofs_build(fdapde_ctx* c, int order, int64_t* n_dofs) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_dofs_build(c, order, n_dofs);
    return fdapde_engine::e_dofs_build(c, order, n_dofs);
}
int fdapde_topology_build(fdapde_ctx* c, int64_t* n_facets, int64_t* n_edges) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_topology_build(c, n_facets, n_edges);
}
int fdapde_topology_get(fdapde_ctx* c, int32_t* neighbors, int32_t* cell_facets, int32_t* facet_nodes, int32_t* facet_cells, uint8_t* facet_boundary, int32_t* edge_nodes, uint8_t* edge_boundary, int32_t* face_edges) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_topology_get(c, neighbors, cell_facets, facet_nodes, facet_cells, facet_boundary, edge_nodes, edge_boundary, face_edges);
}
int fdapde_dofs_set_boundary(fdapde_ctx* c, const uint8_t* bnd) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_dofs_set_boundary(c, bnd);
    return fdapde_engine::e_dofs_set_boundary(c, bnd);
}
int fdapde_dofs_get(const fdapde_ctx* c, int32_t* dofs, uint8_t* bnd, double* coords) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_dofs_get(c, dofs, bnd, coords);
}
int fdapde_pattern_get(const fdapde_ctx* c, int32_t* rowptr, int32_t* colidx) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_pattern_get(c, rowptr, colidx);
}
int fdapde_quadrature_nodes(fdapde_ctx* c, double* out) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_quadrature_nodes(c, out);
}
int fdapde_set_operator(fdapde_ctx* c, int32_t n_terms, const fdapde_term* terms) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_set_operator(c, n_terms, terms);
    return fdapde_engine::e_set_operator(c, n_terms, terms);
}
int fdapde_set_forcing(fdapde_ctx* c, const double* f_q, int32_t n_cols) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_set_forcing(c, f_q, n_cols);
    return fdapde_engine::e_set_forcing(c, f_q, n_cols);
}
int fdapde_set_dirichlet(fdapde_ctx* c, const double* g) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_set_dirichlet(c, g);
    return fdapde_engine::e_set_dirichlet(c, g);
}
int fdapde_assemble_operator(fdapde_ctx* c, int32_t which, int32_t n_terms, const fdapde_term* terms, int32_t assembly) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_assemble_operator(c, which, n_terms, terms, assembly);
    return fdapde_engine::e_assemble_operator(c, which, n_terms, terms, assembly);
}
int fdapde_init(fdapde_ctx* c, const fdapde_options* opt) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_init(c, opt);
    return fdapde_engine::e_init(c, opt);
}
int fdapde_eval_pointwise(fdapde_ctx* c, int64_t n_locs, const double* locs_colmajor, int32_t* cell_ids, double* values) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_eval_pointwise(c, n_locs, locs_colmajor, cell_ids, values);
}
int fdapde_cell_integrals(fdapde_ctx* c, double* measure, double* psi_int) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_cell_integrals(c, measure, psi_int);
}
int fdapde_solver_prepare(fdapde_ctx* c, int32_t with_dirichlet) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_solver_prepare(c, with_dirichlet);
    return fdapde_engine::e_solver_prepare(c, with_dirichlet);
}
int fdapde_solver_layout(fdapde_ctx* c, int32_t with_dirichlet, int64_t* n_interior, int64_t* nnz_interior, double* streamed_bytes) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_layout_kind(c, with_dirichlet, nullptr, nullptr, nullptr, nullptr);
    return fdapde_engine::e_solver_layout(c, with_dirichlet, n_interior, nnz_interior, streamed_bytes);
}
int fdapde_solver_layout_kind(fdapde_ctx* c, int32_t with_dirichlet, int32_t* kind, int32_t* symmetric_storage, int32_t* workgroups, int32_t* rows_per_thread) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_layout_kind(c, with_dirichlet, kind, symmetric_storage, workgroups, rows_per_thread);
    return fdapde_engine::e_solver_layout_kind(c, with_dirichlet, kind, symmetric_storage, workgroups, rows_per_thread);
}
int fdapde_solve(fdapde_ctx* c, const fdapde_options* opt, fdapde_info* info) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_solve(c, opt, info);
    return fdapde_engine::e_solve(c, opt, info);
}
int fdapde_solve_parabolic(fdapde_ctx* c, const fdapde_options* opt, int32_t n_times, double delta_t, const double* initial_condition, const double* dirichlet, double* solution, fdapde_info* info) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_solve_parabolic(c, opt, n_times, delta_t, initial_condition, dirichlet, solution, info);
    return fdapde_engine::e_solve_parabolic(c, opt, n_times, delta_t, initial_condition, dirichlet, solution, info);
}
int fdapde_lin_compute(fdapde_ctx* c, int32_t which, const double* values, int32_t symmetric) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_lin_compute(c, which, values, symmetric);
    return fdapde_engine::e_lin_compute(c, which, values, symmetric);
}
int fdapde_lin_solve(fdapde_ctx* c, const fdapde_options* opt, const double* b, int32_t n_rhs, double* x, fdapde_info* info) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_lin_solve(c, opt, b, n_rhs, x, info);
    return fdapde_engine::e_lin_solve(c, opt, b, n_rhs, x, info);
}
int fdapde_matrix_values(fdapde_ctx* c, int32_t which, double* values) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_matrix_values(c, which, values);
    return fdapde_engine::e_matrix_values(c, which, values);
}
int fdapde_lump(fdapde_ctx* c, int32_t which, double* diag) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_lump(c, which, diag);
    return fdapde_engine::e_lump(c, which, diag);
}
int fdapde_force(fdapde_ctx* c, double* force) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_force(c, force);
    return fdapde_engine::e_force(c, force);
}
int fdapde_solution(fdapde_ctx* c, double* solution) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_solution(c, solution);
    return fdapde_engine::e_solution(c, solution);
}
int fdapde_spmv(fdapde_ctx* c, int32_t which, const double* x, double* y) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fdapde_engine::g_spmv(c, which, x, y);
    return fdapde_engine::e_spmv(c, which, x, y);
}
int fdapde_bench_spmv(fdapde_ctx* c, int32_t reps, double* avg_ms, double* algorithmic_bytes) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_bench_spmv: a multi-device context wires its own ranks");
    return fdapde_engine::e_bench_spmv(c, reps, avg_ms, algorithmic_bytes);
}
int fdapde_comm_unique_id(void* out128) {
    return fdapde_engine::e_comm_unique_id(out128);
}
int fdapde_comm_init(fdapde_ctx* c, int32_t world, int32_t rank, const void* unique_id128) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_comm_init: a multi-device context wires its own ranks");
    return fdapde_engine::e_comm_init(c, world, rank, unique_id128);
}
int fdapde_comm_allreduce(fdapde_ctx* c, double* host_inout, int32_t n, int32_t op) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_comm_allreduce: a multi-device context wires its own ranks");
    return fdapde_engine::e_comm_allreduce(c, host_inout, n, op);
}
int fdapde_comm_count(fdapde_ctx* c, int32_t* ranks) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_comm_count(c, ranks);
}
int fdapde_comm_init_callback(fdapde_ctx* c, int32_t world, int32_t rank, fdapde_allreduce_fn fn, void* user) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_comm_init_callback: a multi-device context wires its own ranks");
    return fdapde_engine::e_comm_init_callback(c, world, rank, fn, user);
}
int fdapde_halo_setup(fdapde_ctx* c, int64_t n_if_global, int64_t n_if_local, const int32_t* local_dof, const int32_t* if_index, const uint8_t* owned) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_halo_setup: a multi-device context wires its own ranks");
    return fdapde_engine::e_halo_setup(c, n_if_global, n_if_local, local_dof, if_index, owned);
}
int fdapde_rowdist_setup(fdapde_ctx* c, const int64_t* dof_key, const int32_t* dof_owner) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_rowdist_setup: a multi-device context wires its own ranks");
    return fdapde_engine::e_rowdist_setup(c, dof_key, dof_owner);
}
int fdapde_comm_set_exchange_callback(fdapde_ctx* c, fdapde_exchange_fn fn, void* user) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_comm_set_exchange_callback: a multi-device context wires its own ranks");
    return fdapde_engine::e_comm_set_exchange_callback(c, fn, user);
}
int fdapde_halo_setup_peers(fdapde_ctx* c, int32_t n_peers, const int32_t* peer_rank, const int64_t* peer_off, const int32_t* peer_dof, const uint8_t* owned) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_halo_setup_peers: a multi-device context wires its own ranks");
    return fdapde_engine::e_halo_setup_peers(c, n_peers, peer_rank, peer_off, peer_dof, owned);
}

int fdapde_ctx_create_multi(const int32_t* devices, int32_t n_devices, fdapde_ctx** out) {
    return fdapde_engine::g_create(devices, n_devices, out);
}
int fdapde_ctx_devices(const fdapde_ctx* c, int32_t* n_devices, int32_t* devices, int32_t* form, double* t_partition_ms, double* t_rank_setup_ms) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::g_info(c, n_devices, devices, form, t_partition_ms, t_rank_setup_ms);
}
int fdapde_partition_build(fdapde_ctx* c, int32_t world, int32_t form) {
    if (!c) return FDAPDE_EINVAL;
    if (c->group) return fail(c, FDAPDE_EUNSUPPORTED, "fdapde_partition_build: a multi-device context partitions its mesh itself (fdapde_dofs_build)");
    return fdapde_engine::e_partition_build(c, world, form);
}
int fdapde_partition_sizes(const fdapde_ctx* c, int32_t rank, int64_t* n_nodes, int64_t* n_cells) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_partition_sizes(c, rank, n_nodes, n_cells);
}
int fdapde_partition_get(fdapde_ctx* c, int32_t rank, double* nodes_colmajor, int32_t* cells, uint8_t* boundary, int64_t* node_ids, int64_t* cell_ids, int32_t* node_owner) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_partition_get(c, rank, nodes_colmajor, cells, boundary, node_ids, cell_ids, node_owner);
}
int fdapde_partition_whole(fdapde_ctx* c, int32_t* cell_rank, int32_t* node_owner, uint64_t* node_ranks) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_partition_whole(c, cell_rank, node_owner, node_ranks);
}
int fdapde_partition_peers(fdapde_ctx* c, int32_t rank, int32_t* n_peers, int32_t* peer_rank, int64_t* peer_off, int32_t* peer_node, uint8_t* owned, int64_t* n_shared) {
    if (!c) return FDAPDE_EINVAL;
    return fdapde_engine::e_partition_peers(c, rank, n_peers, peer_rank, peer_off, peer_node, owned, n_shared);
}

int fdapde_info_get(const fdapde_ctx* c, fdapde_info* info) {
    if (!c || !info) return FDAPDE_EINVAL;
    *info = c->info;
    return FDAPDE_OK;
}

// which RCCL the library bound itself to (diagnostics; empty before the first communicator call)
const char* fdapde_comm_library(void) { return g_rccl.path.c_str(); }

int fdapde_tune(fdapde_ctx* c, const char* key, int32_t value) {
    if (!c || !key) return FDAPDE_EINVAL;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (c->group) return fdapde_engine::g_tune(c, key, value);
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
    else if (k == "asm_fuse_mass" && value >= 0 && value <= 3) c->asm_fuse_mass = value;
    else if (k == "asm_items" && (value == 0 || value == 1)) c->asm_items = value;
    else if (k == "asm_items_fuse" && (value == 0 || value == 1)) c->asm_items_fuse = value;
    else if (k == "bicg_restart" && (value == 0 || value == 1)) c->bicg_restart = value;
    else if (k == "bicg_shadow" && value >= 0 && value <= 2) c->bicg_shadow = value;
    else if (k == "gmres_m" && value >= 2 && value <= 200) c->gmres_m = value;
    else if (k == "persist_exp_lds" && (value == 0 || value == 1)) {
        c->persist_exp_lds = value;
        c->ps[0].tried = c->ps[0].ok = c->ps[1].tried = c->ps[1].ok = false, c->scaled_owner = fdapde_ctx::kScaledNone;
        fdapde_engine::drop_graph(c);
    }
    else if (k == "dense_rows" && value >= 0 && value <= 8192) c->dense_rows = value, c->lin_dense.ready = false, c->lin_dense.failed = false;
    else if (k == "dense_after" && value >= 0) c->dense_after = value;
    else if (k == "dense_block" && (value == 0 || value == 1)) c->dense_block = value;
    else if (k == "dense_direct" && (value == 0 || value == 1)) c->dense_direct = value;
    else if (k == "dense_fold" && (value == 0 || value == 1)) c->dense_fold = value;
    else if (k == "dense_multi" && (value == 0 || value == 1)) c->dense_multi = value;
    else if (k == "pmg_inner_tol_exp" && value >= 1 && value <= 12) c->pmg_inner_rtol = std::pow(10.0, -(double)value);
    else if (k == "pmg_inner_maxit" && value >= 1) c->pmg_inner_maxit = value;
    else if (k == "pmg_auto" && (value == 0 || value == 1)) c->pmg_auto = value;
    else if (k == "pmg_blocked" && (value == 0 || value == 1)) c->pmg_blocked = value;
    else if (k == "pmg_smooth" && (value == 0 || value == 1)) c->pmg_smooth = value;
    else if (k == "pmg_setup_check" && (value == 0 || value == 1)) c->pmg_setup_check = value;
    else if (k == "pmg_restart" && value >= 2 && value <= 50) c->pmg_restart = (int)value;
    else if (k == "pmg_outer" && (value == 0 || value == 1)) c->pmg_outer = value;
    else if (k == "pmg_auto_rows" && value >= 0) c->pmg_auto_rows = value;
    else if (k == "pmg_auto_first_rows" && value >= 0) c->pmg_auto_first_rows = value;
    else if (k == "dense_bulk" && (value == 0 || value == 1)) c->dense_bulk = value;
    else if (k == "dense_hostb" && (value == 0 || value == 1)) c->dense_hostb = value;
    else if (k == "small_rows" && value >= 0) c->small_rows = value;
    else if (k == "small_front_rows" && value >= 0) c->small_front_rows = value;
    else if (k == "auto_gmres" && (value == 0 || value == 1)) c->auto_gmres = value;
    else if (k == "asm_split_varying" && (value == 0 || value == 1)) c->asm_split_varying = value;
    else if (k == "asm_row_stat" && (value == 0 || value == 1)) c->asm_row_stat = value, c->stiff_stat_valid = false;
    else if (k == "asm_fq_bc" && (value == 0 || value == 1)) c->asm_fq_bc = value, c->fq_bc_ready = c->fq_bc_ready && value;   // (takes effect fully at the next fdapde_set_forcing)
    else if (k == "persist" && (value == 0 || value == 1)) c->persist = value, c->persist_broken = false;
    else if (k == "persist_time" && (value == 0 || value == 1)) c->persist_time = value;
    else if (k == "persist_coop" && (value == 0 || value == 1)) c->persist_coop = value;
    else if (k == "persist_bicg" && (value == 0 || value == 1)) c->persist_bicg = value;
    else if (k == "persist_fill_fused" && (value == 0 || value == 1)) c->persist_fill_fused = value;
    else if (k == "persist_prefetch" && (value == 0 || value == 1)) c->persist_prefetch = value;
    else if (k == "persist_cols" && (value == 0 || value == 1)) c->persist_cols = value;
    else if (k == "persist_direct" && (value == 0 || value == 1)) c->persist_direct = value;
    else if (k == "persist_wide_gj" && (value == 4 || value == 6 || value == 12)) {
        c->persist_wide_gj = value;
        for (auto& ps : c->ps) ps.attr_set = nullptr;
    }
    else if ((k == "persist_wide" && (value == 0 || value == 1)) || (k == "persist_max_wg" && value >= 0)) {
        if (k == "persist_wide") c->persist_wide = value;
        else c->persist_max_wg = value;
        for (auto& ps : c->ps) ps.tried = ps.ok = ps.filled = false;
    }
    else if (k == "persist_direct_spin_us" && value >= 0 && value <= 1000000) c->persist_direct_spin_us = value;
    else if (k == "persist_single_rows" && value >= 0 && value <= 8192) {
        c->persist_single_rows = value;
        for (auto& ps : c->ps) ps.tried = ps.ok = ps.filled = false;
    }
    else if (k == "persist_late" && (value == 0 || value == 1)) {
        c->persist_late = value;
        for (auto& ps : c->ps) ps.tried = ps.ok = ps.filled = false;
    }
    else if (k == "rowdist_max_wg" && value >= 0) {   // workgroups of this rank's launch (tests: several ranks share one device)
        c->rd.max_wg = value;
        for (auto& L : c->rd.lay) L.tried = L.ok = false;
    } else if (k == "rowdist_share" && value >= 1) {   // that many ranks share this device: an equal share of its CUs each
        c->rd.max_wg = std::max(1, c->n_cu / value);
        for (auto& L : c->rd.lay) L.tried = L.ok = false;
    } else if (k == "rowdist_timeout_first_ms" && value >= 1) c->rd.timeout_first_ms = value;
    else if (k == "rowdist_flat_gather" && value >= -1 && value <= 1) c->rd.flat_gather = value;
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
    if (c->group)
        if (int rc = fdapde_engine::g_synchronize(c)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipDeviceSynchronize());   // (what torch.cuda.synchronize() would do: nothing of this process is left running on the device)
    return FDAPDE_OK;
}

}  // extern "C"
