// eng_clone.hip -- fdapde_ctx_clone: an independent context holding the same problem as another one.
//
// Why it exists: the reference consumes PDEs through a type-erased handle that DEEP-COPIES the whole PDE -- solver and its matrices
// included -- on construction and on every handle copy (make_pde -> erase<heap_storage, PDE__>: fdaPDE/pde/pde.h:167-169,
// fdaPDE/utils/type_erasure.h:124-146).  A solver object that owns a device context therefore needs a copy operation.  The host-side
// bindings share a context between copies and call this only when one of them is about to change it (copy-on-write,
// include/fdapde_hip.hpp); what comes out must behave exactly like the source: same mesh, space and boundary mask, same operator /
// forcing / Dirichlet data, the same assembled stiff_ / mass_ / force_ and, after a solve, the same solution_.
//
// How: the function space is REBUILT from the source's host copy of the mesh (the same deterministic set-up: every index array comes out
// bit-identical), the problem data and everything fdapde_init / fdapde_solve produced are copied device to device.  Solver layouts,
// scaled copies and work vectors are not copied: the clone's first solve prepares them like any first solve.
#include <cstring>
#include <new>

#include "context.h"
#include "engine.h"

namespace fdapde_engine {

namespace {
template <typename T> int copy_buf(fdapde_ctx* d, DBuf<T>& dst, const DBuf<T>& src, size_t count) {
    if (!src.p || count == 0) return FDAPDE_OK;
    if (count > src.n) count = src.n;
    HIPCHK(d, dst.alloc(count));
    HIPCHK(d, hipMemcpyAsync(dst.p, src.p, sizeof(T) * count, hipMemcpyDeviceToDevice, d->stream));
    return FDAPDE_OK;
}
}   // namespace

int e_ctx_clone(const fdapde_ctx* s, fdapde_ctx* d) {
    if (s->comm != nullptr || s->ar_fn != nullptr || s->halo_ready || s->rd.ready)
        return fail(d, FDAPDE_EUNSUPPORTED, "fdapde_ctx_clone: a context that is a rank of a multi-GPU job is not cloned (communicators are not copyable)");
    const HostSpace& hs = s->hs;
    if (hs.n_cells == 0) return FDAPDE_OK;   // nothing uploaded yet
    if (int rc = host_set_mesh(d->hs, hs.M, hs.N, hs.n_nodes, hs.nodes.data(), hs.n_cells, hs.cells.data(), hs.node_bnd.data(), d->err)) return rc;
    if (s->has_device && d->has_device && s->mesh_on_dev) {   // the device copy of the mesh: device to device
        HIPCHK(d, hipSetDevice(s->device));
        if (int rc = copy_buf(d, d->mesh_nodes, s->mesh_nodes, s->mesh_nodes.n)) return rc;
        if (int rc = copy_buf(d, d->mesh_cells, s->mesh_cells, s->mesh_cells.n)) return rc;
        if (int rc = copy_buf(d, d->mesh_nbnd, s->mesh_nbnd, s->mesh_nbnd.n)) return rc;
        d->mesh_on_dev = true;
    }
    if (!s->space_ready) {
        if (d->mesh_on_dev) HIPCHK(d, hipStreamSynchronize(d->stream));
        return FDAPDE_OK;
    }
    if (s->has_device) {
        HIPCHK(d, hipSetDevice(s->device));
        HIPCHK(d, hipStreamSynchronize(s->stream));   // whatever the source still has in flight
    }
    if (int rc = e_dofs_build(d, hs.order, nullptr)) return rc;
    if (d->hs.n_dofs != hs.n_dofs || d->hs.nnz != hs.nnz) return fail(d, FDAPDE_EHIP, "fdapde_ctx_clone: the rebuilt space differs from the source's");
    if (d->hs.dof_bnd != hs.dof_bnd)   // fdapde_dofs_set_boundary was used on the source
        if (int rc = e_dofs_set_boundary(d, hs.dof_bnd.data())) return rc;
    return clone_state(s, d);
}

// the part of a clone that follows the rebuilt space: problem data and everything fdapde_init / fdapde_solve / fdapde_lin_compute produced, device to
// device.  Also what a multi-device context's clone does rank by rank (eng_group.hip: its ranks' spaces are rebuilt by the same deterministic split).
int clone_state(const fdapde_ctx* s, fdapde_ctx* d) {
    const HostSpace& hs = s->hs;
    if (s->has_device && d->has_device) {
        HIPCHK(d, hipSetDevice(d->device));
        HIPCHK(d, hipStreamSynchronize(s->stream));   // whatever the source still has in flight
    }
    d->info = s->info;
    // ---- problem data: operator (space-varying coefficient rows already in the internal cell order), forcing samples, Dirichlet data
    d->op.clear();
    for (const HostTerm& t : s->op) {
        HostTerm c;
        c.t = t.t, c.data_i = t.data_i, c.field_nonsym = t.field_nonsym;
        if (t.data_dev) {
            c.data_dev = std::make_shared<DBuf<double>>();
            if (int rc = copy_buf(d, *c.data_dev, *t.data_dev, t.data_dev->n)) return rc;
        }
        d->op.push_back(std::move(c));
    }
    d->op_symmetric = s->op_symmetric, d->coef_of_op = false;   // (coefficient slots are uploaded again by the clone's next fdapde_init)
    d->fq_i = s->fq_i, d->fq_cols = s->fq_cols;
    d->g_i = s->g_i, d->have_g = s->have_g, d->g_zero = s->g_zero;
    if (!s->has_device || !d->has_device) return FDAPDE_OK;
    const size_t n = (size_t)hs.n_dofs, nnz = (size_t)hs.nnz, rows = (size_t)hs.nq * (size_t)hs.n_cells;
    const size_t cols = s->fq_cols > 0 ? (size_t)s->fq_cols : 1;
    if (s->fq_cols > 0) {
        if (int rc = copy_buf(d, d->fq, s->fq, rows * cols)) return rc;
        if (s->fq_bc_ready) {
            if (int rc = copy_buf(d, d->fq_bc, s->fq_bc, s->fq_bc.n)) return rc;
            d->fq_bc_ready = true;
        }
        HIPCHK(d, d->force.alloc(n * cols));
    }
    d->asm_fq_bc = s->asm_fq_bc, d->asm_fq_block = s->asm_fq_block, d->asm_fuse_mass = s->asm_fuse_mass, d->asm_row_stat = s->asm_row_stat;
    if (s->have_g)
        if (int rc = copy_buf(d, d->g, s->g, n)) return rc;
    // ---- what fdapde_init left: stiff_, mass_, force_ (+ the row statistics the Jacobi scaling reads)
    for (int w = 0; w < 2; ++w)
        if (s->assembled[w]) {
            if (int rc = copy_buf(d, d->vals[w], s->vals[w], nnz + 2)) return rc;
            d->assembled[w] = true;
        }
    if (s->force_ready) {
        if (int rc = copy_buf(d, d->force, s->force, n * cols)) return rc;
        d->force_ready = true;
    }
    if (s->stiff_stat_valid) {
        if (int rc = copy_buf(d, d->stiff_stat, s->stiff_stat, 2 * n)) return rc;
        d->stiff_stat_valid = true;
    }
    // ---- what fdapde_solve left: solution_, and the flag that makes stiff() / force() export the row-zeroed system
    if (s->solved) {
        if (int rc = copy_buf(d, d->u, s->u, n)) return rc;
        d->solved = true;
    }
    d->dirichlet_applied = s->dirichlet_applied;
    // ---- the factor-once handle: its matrix travels, the scaled system is prepared again by the clone's first fdapde_lin_solve
    if (s->lin_ready && s->lin_state) {
        if (int rc = copy_buf(d, d->lin_mat, s->lin_mat, nnz + 2)) return rc;
        d->lin_symmetric = s->lin_symmetric;
        if (!d->lin_state) d->lin_state = new (std::nothrow) SolveStateHolder();
        if (!d->lin_state) return fail(d, FDAPDE_ENOMEM, "fdapde_ctx_clone: out of memory");
        d->lin_state->ss = s->lin_state->ss;
        d->lin_ready = true, d->lin_sq_ready = false, d->scaled_owner = fdapde_ctx::kScaledNone;
    }
    HIPCHK(d, hipStreamSynchronize(d->stream));
    return FDAPDE_OK;
}

}   // namespace fdapde_engine
