// engine.h -- what the translation units of libfdapde_hip.so share besides the context: small helpers and the entry points of the
// internal engines.  capi.hip holds the C ABI and the multi-launch solve driver; persist_engine.hip the single-launch solver
// (layouts, launches, the row-distributed multi-GPU form).  Internal to the library.
#ifndef FDAPDE_ENGINE_H
#define FDAPDE_ENGINE_H

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "context.h"

namespace fdapde_engine {

// FDAPDE_DEBUG_TIMING=1: wall-clock marks of the host side of a solve on stderr (where a first solve spends its time)
struct DebugClock {
    bool on = std::getenv("FDAPDE_DEBUG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[timing] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

inline int fail(fdapde_ctx* c, int code, const char* msg) {
    c->err = msg;
    return code;
}
inline int need_device(fdapde_ctx* c) {
    if (!c->has_device) return fail(c, FDAPDE_ENODEVICE, "this context has no HIP device (host-only); there is no CPU fallback");
    return FDAPDE_OK;
}

// a DBuf takes over a device array built elsewhere (dev_setup.hip)
template <typename T> void adopt(DBuf<T>& b, T*& p, size_t n) {
    b.release();
    b.p = p, b.n = n, p = nullptr;
}


inline unsigned grid1(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// big host-side index arrays of a device-built space, fetched the first time host code needs them (capi.hip)
// kHostPerm: the permutations (dof_i2e, dof_e2i, cell_i2e) and the boundary flags in internal order (dof_bnd_i); implied by kHostPattern
enum { kHostPattern = 1, kHostCells = 2, kHostRefPattern = 4, kHostDofs = 8, kHostPerm = 16 };
int ensure_host(fdapde_ctx* c, int what);

inline unsigned g1(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

#define RCCLCHK(ctx, expr)                                                                   \
    do {                                                                                     \
        ncclResult_t r__ = (expr);                                                           \
        if (r__ != ncclSuccess) {                                                            \
            (ctx)->err = std::string(#expr) + ": " + g_rccl.GetErrorString(r__);             \
            return FDAPDE_ERCCL;                                                             \
        }                                                                                    \
    } while (0)

// ---- helpers one engine unit offers the others ---------------------------------------------------------------------------------------
int ensure_sval(fdapde_ctx* c);                                            // eng_solve.hip: the scaled full-pattern copy, if a solve skipped it
void drop_graph(fdapde_ctx* c);                                            // eng_solve.hip: the captured CG chunk bakes pointers and sizes in
int ranks_saying_yes(fdapde_ctx* c, bool mine, int* yes);                // eng_dist.hip: how many ranks of the job answer yes (COLLECTIVE)
int allreduce_sum(fdapde_ctx* c, double* buf, size_t count);               // eng_dist.hip: device buffer summed over the ranks
int halo_sum(fdapde_ctx* c, double* v, const double* part, int np, bool unpack = true);   // eng_dist.hip: interface entries summed over the sharing ranks

// ---- the bodies behind the C ABI (capi.hip forwards to them; each unit's header comment says what it holds) ----------------------------
int e_ctx_clone(const fdapde_ctx* src, fdapde_ctx* dst);   // eng_clone.hip
int clone_state(const fdapde_ctx* src, fdapde_ctx* dst);   // ... its second half: problem data + assembled / solved state onto an identical space
int g_clone(const fdapde_ctx* src_root, fdapde_ctx** out);   // eng_group.hip
int e_dofs_build(fdapde_ctx* c, int order, int64_t* n_dofs);
int e_topology_build(fdapde_ctx* c, int64_t* n_facets, int64_t* n_edges);
int e_topology_get(fdapde_ctx* c, int32_t* neighbors, int32_t* cell_facets, int32_t* facet_nodes, int32_t* facet_cells, uint8_t* facet_boundary, int32_t* edge_nodes, uint8_t* edge_boundary, int32_t* face_edges);
int e_dofs_set_boundary(fdapde_ctx* c, const uint8_t* bnd);
int e_dofs_get(const fdapde_ctx* c, int32_t* dofs, uint8_t* bnd, double* coords);
int e_pattern_get(const fdapde_ctx* c, int32_t* rowptr, int32_t* colidx);
int e_quadrature_nodes(fdapde_ctx* c, double* out);
int e_set_operator(fdapde_ctx* c, int32_t n_terms, const fdapde_term* terms);
int e_set_forcing(fdapde_ctx* c, const double* f_q, int32_t n_cols);
int e_set_dirichlet(fdapde_ctx* c, const double* g);
int e_assemble_operator(fdapde_ctx* c, int32_t which, int32_t n_terms, const fdapde_term* terms, int32_t assembly);
int e_init(fdapde_ctx* c, const fdapde_options* opt);
int e_eval_pointwise(fdapde_ctx* c, int64_t n_locs, const double* locs_colmajor, int32_t* cell_ids, double* values);
int e_cell_integrals(fdapde_ctx* c, double* measure, double* psi_int);
int e_solver_prepare(fdapde_ctx* c, int32_t with_dirichlet);
int e_solver_layout(fdapde_ctx* c, int32_t with_dirichlet, int64_t* n_interior, int64_t* nnz_interior, double* streamed_bytes);
int e_solver_layout_kind(fdapde_ctx* c, int32_t with_dirichlet, int32_t* kind, int32_t* symmetric_storage, int32_t* workgroups, int32_t* rows_per_thread);
int e_solve(fdapde_ctx* c, const fdapde_options* opt, fdapde_info* info);
int e_solve_parabolic(fdapde_ctx* c, const fdapde_options* opt, int32_t n_times, double delta_t, const double* initial_condition, const double* dirichlet, double* solution, fdapde_info* info);
int e_lin_compute(fdapde_ctx* c, int32_t which, const double* values, int32_t symmetric);
int e_lin_solve(fdapde_ctx* c, const fdapde_options* opt, const double* b, int32_t n_rhs, double* x, fdapde_info* info);
int e_matrix_values(fdapde_ctx* c, int32_t which, double* values);
int e_lump(fdapde_ctx* c, int32_t which, double* diag);
int e_force(fdapde_ctx* c, double* force);
int e_solution(fdapde_ctx* c, double* solution);
int e_spmv(fdapde_ctx* c, int32_t which, const double* x, double* y);
int e_bench_spmv(fdapde_ctx* c, int32_t reps, double* avg_ms, double* algorithmic_bytes);
int e_comm_unique_id(void* out128);
int e_comm_init(fdapde_ctx* c, int32_t world, int32_t rank, const void* unique_id128);
int e_comm_allreduce(fdapde_ctx* c, double* host_inout, int32_t n, int32_t op);
int e_comm_count(fdapde_ctx* c, int32_t* ranks);
int e_comm_init_callback(fdapde_ctx* c, int32_t world, int32_t rank, fdapde_allreduce_fn fn, void* user);
int e_halo_setup(fdapde_ctx* c, int64_t n_if_global, int64_t n_if_local, const int32_t* local_dof, const int32_t* if_index, const uint8_t* owned);
int e_rowdist_setup(fdapde_ctx* c, const int64_t* dof_key, const int32_t* dof_owner);
int e_comm_set_exchange_callback(fdapde_ctx* c, fdapde_exchange_fn fn, void* user);
int e_halo_setup_peers(fdapde_ctx* c, int32_t n_peers, const int32_t* peer_rank, const int64_t* peer_off, const int32_t* peer_dof, const uint8_t* owned);

// ---- the device-side partitioner's public face and the multi-device context (eng_group.hip) ------------------------------------------------
int e_partition_build(fdapde_ctx* c, int32_t world, int32_t form);
int e_partition_sizes(const fdapde_ctx* c, int32_t rank, int64_t* n_nodes, int64_t* n_cells);
int e_partition_get(fdapde_ctx* c, int32_t rank, double* nodes_colmajor, int32_t* cells, uint8_t* boundary, int64_t* node_ids, int64_t* cell_ids, int32_t* node_owner);
int e_partition_whole(fdapde_ctx* c, int32_t* cell_rank, int32_t* node_owner, uint64_t* node_ranks);
int e_partition_peers(fdapde_ctx* c, int32_t rank, int32_t* n_peers, int32_t* peer_rank, int64_t* peer_off, int32_t* peer_node, uint8_t* owned, int64_t* n_shared);
void partition_free(fdapde_ctx* c);
int g_create(const int32_t* devices, int32_t n, fdapde_ctx** out);
void g_destroy(fdapde_ctx* root);
int g_info(const fdapde_ctx* root, int32_t* n_devices, int32_t* devices, int32_t* form, double* t_partition_ms, double* t_rank_setup_ms);
void g_mesh_changed(fdapde_ctx* root);
int g_dofs_build(fdapde_ctx* root, int order, int64_t* n_dofs);
int g_dofs_set_boundary(fdapde_ctx* root, const uint8_t* bnd);
int g_set_operator(fdapde_ctx* root, int32_t n_terms, const fdapde_term* terms);
int g_assemble_operator(fdapde_ctx* root, int32_t which, int32_t n_terms, const fdapde_term* terms, int32_t assembly);
int g_set_forcing(fdapde_ctx* root, const double* f_q, int32_t n_cols);
int g_set_dirichlet(fdapde_ctx* root, const double* g);
int g_init(fdapde_ctx* root, const fdapde_options* opt);
int g_solver_prepare(fdapde_ctx* root, int32_t with_dirichlet);
int g_solve(fdapde_ctx* root, const fdapde_options* opt, fdapde_info* info);
int g_solution(fdapde_ctx* root, double* solution);
int g_force(fdapde_ctx* root, double* force);
int g_lump(fdapde_ctx* root, int32_t which, double* diag);
int g_matrix_values(fdapde_ctx* root, int32_t which, double* values);
int g_spmv(fdapde_ctx* root, int32_t which, const double* x, double* y);
int g_solve_parabolic(fdapde_ctx* root, const fdapde_options* opt, int32_t n_times, double delta_t, const double* initial_condition, const double* dirichlet, double* solution, fdapde_info* info);
int g_lin_compute(fdapde_ctx* root, int32_t which, const double* values, int32_t symmetric);
int g_lin_solve(fdapde_ctx* root, const fdapde_options* opt, const double* b, int32_t n_rhs, double* x, fdapde_info* info);
int g_tune(fdapde_ctx* root, const char* key, int32_t value);
int g_synchronize(fdapde_ctx* root);
int g_layout_kind(fdapde_ctx* root, int32_t with_dirichlet, int32_t* kind, int32_t* symmetric_storage, int32_t* workgroups, int32_t* rows_per_thread);

// ---- dense inverse of small systems (eng_dense.hip)
bool dense_eligible(const fdapde_ctx* c);
double dense_build_estimate_ms(int64_t n);
int dense_build(fdapde_ctx* c, const double* A, int use_bnd, fdapde_ctx::Dense& D);
int dense_apply(fdapde_ctx* c, fdapde_ctx::Dense& D, int nc, const double* b, double* x);
int dense_solve_host(fdapde_ctx* c, fdapde_ctx::Dense& D, const double* b_host, int nc, double* x_host);
int dense_direct(fdapde_ctx* c, const double* A, int use_bnd, const double* f_dev, const double* g_dev, bool* solved);
void dense_step_rhs(fdapde_ctx* c, const double* mu, double inv_dt, const double* f, const double* g_ext_dev, double* rhs);
void dense_step_out(fdapde_ctx* c, const double* u, double* uprev, double* sol_ext_dev);
// eng_solve.hip: the blocked-ELL SpMV layout of boundary variant v (c->bk[v]; ok = false where the system does not take it) and its launch on whatever ell_val holds
int build_blocked(fdapde_ctx* c, int v);
void launch_spmv_blocked(fdapde_ctx* c, int v, const double* x, double* y, const double* w, double* partial, const int32_t* stop, hipEvent_t e0, hipEvent_t e1, int dot2_ww);
// eng_solve.hip: the coarse-level solves of the two-level solver -- the solver prepared once per coarse operator, then one run per right-hand side (c->force)
int coarse_prepare(fdapde_ctx* c, SolveState* ss);
int coarse_solve(fdapde_ctx* c, SolveState* ss, double rtol, int maxit, fdapde_info* info);
// eng_pmg.hip
bool pmg_eligible(const fdapde_ctx* c);
void pmg_release(fdapde_ctx* c);
int e_solve_pmg(fdapde_ctx* c, const fdapde_options* opt, fdapde_info* info);
int pmg_run(fdapde_ctx* c, const double* A, const double* f_dev, const double* g_dev, int use_bnd, const double* x0_dev, double extra_reaction, int64_t coarse_key,
            double rtol, int maxit);
int dense_step_loop(fdapde_ctx* c, fdapde_ctx::Dense& D, int32_t n_times, double inv_dt, const double* g_ext_dev, double* u0, double* sol_ext);
void preload_dense();

// code objects of the units loaded up front (fdapde_ctx_create)
void preload_assembly();
void preload_solve();
void preload_dist();
void preload_persist();
// ... and those of the set-up units (13 + 6 + 6 MB of device code: 30 ms to load), on a helper thread started by the first fdapde_ctx_create
// of a process; preload_wait(units) waits for the first `units` of them (fdapde_dofs_build, the solver layout, fdapde_topology_build)
void preload_setup_async(int device);
void preload_wait(int units);

// ---- single-launch solver (persist_engine.hip) -----------------------------------------------------------------------------------
// layout of boundary variant v, built on first use (ps.tried / ps.ok tell the outcome)
int build_persist(fdapde_ctx* c, int v);
// scaled values into the layout's blocks (after the scaled full-pattern matrix c->sval is in place)
int fill_persist(fdapde_ctx* c, int v);
int fill_persist_scaled(fdapde_ctx* c, int v, const double* A);   // ... from the unscaled matrix + c->scale (no scaled full-pattern copy needed)
// the whole fused-update CG as one launch; *ran = false: the launch gave up (hand-off timeout) or can never be resident
int run_persist_cols(fdapde_ctx* c, int v, double tol2, int maxit, int n_cols, const double* r_cols, double* x_cols, double* sc_cols, int32_t* ctl_cols,
                     int32_t* h_ctl, double* h_sc, bool* ran, bool bicg = false);   // several right-hand sides side by side in one launch
int run_persist(fdapde_ctx* c, int v, double tol2, int maxit, bool* ran, bool bicg = false);
int run_persist_direct(fdapde_ctx* c, int v, double tol2, int maxit, const double* b_host, double* x_host, bool* ran);   // one column, one workgroup, zero-copy in and out   // bicg: the BiCGStab kernel (plain layouts, <= 8 rows per thread)

// ---- row-distributed multi-GPU form of the single-launch solver (persist_engine.hip); all COLLECTIVE over the context's ranks
int build_rowdist(fdapde_ctx* c, int v);
int rowdist_import_ghosts(fdapde_ctx* c, int v, double* vec);   // owners' values of a per-DOF vector into this rank's ghost columns
int fill_rowdist(fdapde_ctx* c, int v);
int run_rowdist(fdapde_ctx* c, int v, double tol2, int maxit, bool* ran, bool bicg = false);
void release_rowdist(fdapde_ctx* c);

}   // namespace fdapde_engine

#endif
