// eng_assembly.hip -- operator / forcing / Dirichlet data hand-over, the assembly launches behind fdapde_init / fdapde_assemble_operator
// (row-owner sweep and the scatter forms), basis evaluation (fdapde_eval_pointwise, fdapde_cell_integrals).
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
#include "dev_setup.h"
#include "engine.h"
#include "kernels.h"

namespace fdapde_engine {

AsmArgs asm_args(fdapde_ctx* c) {
    AsmArgs a{};
    a.n_dofs = c->hs.n_dofs, a.n_cells = c->hs.n_cells;
    a.cverts = c->cverts.p, a.cdofs = c->cdofs.p, a.vcoords = c->vcoords.p;
    a.sl_off = c->sl_off.p, a.adj = c->adj.p, a.slotw = c->slotw.p, a.lane_row = c->lane_row.p;   // nullptr = identity
    a.rowptr = c->rowptr.p, a.colidx = c->colidx.p, a.tables = c->tables.p, a.reftab = c->reftab.p, a.ref_doubles = kRefDoubles;
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
        if (t.space_varying && terms[k].data_dev) {
            d.data = terms[k].data_dev->p;   // (permuted on the device when the operator was handed over)
        } else if (t.space_varying) {
            DBuf<double>& buf = c->coef[coef_slot0 + k];
            if (!(reuse && buf.p && buf.n >= terms[k].data_i.size()))
                HIPCHK(c, buf.upload(terms[k].data_i.data(), terms[k].data_i.size(), c->stream));
            d.data = buf.p;
        }
    }
    // constant-coefficient summary (element_row OPK 3)
    const int N = c->hs.N;
    bool adv = false;
    for (size_t k = 0; k < terms.size(); ++k) {   // (the CONSTANT leaves: a space-varying leaf only leaves its mark in var_kinds -- element_row OPK 5)
        const fdapde_term& t = terms[k].t;
        if (t.kind == FDAPDE_ADVECTION) adv = true;
        if (t.space_varying) {
            op.var_kinds |= t.kind == FDAPDE_DIFFUSION ? 1 : t.kind == FDAPDE_ADVECTION ? 2 : t.kind == FDAPDE_REACTION ? 4 : 0;
            continue;
        }
        if (t.kind == FDAPDE_LAPLACIAN)
            for (int r = 0; r < N; ++r) op.kt[r * N + r] += t.coef;
        else if (t.kind == FDAPDE_DIFFUSION)
            for (int e = 0; e < N * N; ++e) op.kt[e] += t.coef * t.cst[e];
        else if (t.kind == FDAPDE_ADVECTION)
            for (int e = 0; e < N; ++e) op.bt[e] += t.coef * t.cst[e];
        else if (t.kind == FDAPDE_REACTION)
            op.ct += t.coef * t.cst[0];
    }
    bool ksym = true;
    for (int r = 0; r < N; ++r)
        for (int q = 0; q < r; ++q) ksym = ksym && op.kt[r * N + q] == op.kt[q * N + r];
    op.tab_sym = (ksym && !adv) ? 1 : 0;
    op.kt_sym = ksym ? 1 : 0;
    bool any_adv = false, field_nonsym = false;
    for (size_t k = 0; k < terms.size(); ++k) {
        any_adv = any_adv || terms[k].t.kind == FDAPDE_ADVECTION;
        field_nonsym = field_nonsym || (terms[k].t.kind == FDAPDE_DIFFUSION && terms[k].field_nonsym);
    }
    op.mirror = (!any_adv && (!ksym || field_nonsym)) ? 1 : 0;
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
            if (c->has_device && c->dev_ready) {   // as handed over to the device, permuted to the internal cell order there (a serial host
                                                   // loop before: 0.77 s for the three fields of a C5-size operator)
                HIPCHK(c, hipSetDevice(c->device));
                DBuf<double> stage;
                h.data_dev = std::make_shared<DBuf<double>>();
                HIPCHK(c, stage.upload(terms[k].data, (size_t)rows * width, c->stream));
                HIPCHK(c, h.data_dev->alloc((size_t)rows * width));
                const int grp = hs.nq * width;
                hipLaunchKernelGGL(k_gather_row_groups, dim3((unsigned)(((int64_t)rows * width + 255) / 256)), dim3(256), 0, c->stream, hs.n_cells, grp,
                                   c->cell_i2e.p, stage.p, h.data_dev->p);
                HIPCHK(c, hipGetLastError());
                if (terms[k].kind == FDAPDE_DIFFUSION) {   // a row that is not a symmetric tensor?  (make_dev_op: DevOp::mirror)
                    DBuf<int32_t> flag;
                    int32_t h_flag = 0;
                    HIPCHK(c, flag.alloc(1));
                    HIPCHK(c, hipMemsetAsync(flag.p, 0, sizeof(int32_t), c->stream));
                    hipLaunchKernelGGL(k_field_asym, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, rows, hs.N, h.data_dev->p, flag.p);
                    HIPCHK(c, hipMemcpyAsync(&h_flag, flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    h.field_nonsym = h_flag != 0;
                }
                HIPCHK(c, hipStreamSynchronize(c->stream));   // (stage is released at the end of this scope)
            } else {
                h.data_i.resize((size_t)rows * width);
                for (int64_t ci = 0; ci < hs.n_cells; ++ci) {
                    const int64_t ce = hs.cell_i2e[(size_t)ci];
                    std::memcpy(&h.data_i[(size_t)ci * hs.nq * width], &terms[k].data[(size_t)ce * hs.nq * width],
                                sizeof(double) * hs.nq * width);
                }
                if (terms[k].kind == FDAPDE_DIFFUSION)
                    for (int64_t r = 0; r < rows && !h.field_nonsym; ++r)
                        for (int a = 1; a < hs.N; ++a)
                            for (int b = 0; b < a; ++b)
                                if (h.data_i[(size_t)r * width + a * hs.N + b] != h.data_i[(size_t)r * width + b * hs.N + a]) h.field_nonsym = true;
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
        int opk = 4;   // space-varying coefficients: one pulled-back tensor per quadrature node
        // ... unless the diffusion part does not vary: constants through the reference tensors, the varying advection / reaction per node
        // (knob asm_split_varying = 0: the per-node tensor form for every space-varying operator)
        if (op.needs_rows && (op.var_kinds & 1) == 0 && c->asm_split_varying) opk = 5;
        if (!op.needs_rows) opk = 3;
        if (op.n == 1 && op.t[0].kind == FDAPDE_LAPLACIAN) opk = 1;
        if (op.n == 1 && op.t[0].kind == FDAPDE_REACTION && !op.t[0].space_varying) opk = 2;
        if (std::getenv("FDAPDE_ASM_GENERIC")) opk = 0;
        if (op.kt_sym) a.reftab = reinterpret_cast<const DevRefTensors*>(c->reftab_sym.p), a.ref_doubles = kRefSymDoubles;   // (the compact tensors)
        else a.reftab = c->reftab.p, a.ref_doubles = kRefDoubles;
        const size_t tab = sizeof(DevTables) + (opk == 3 || opk == 5 ? sizeof(double) * (size_t)a.ref_doubles : 0) +
                           (size_t)hs.max_blk_nodes * (M == 2 ? 2 : 3) * sizeof(double);
        size_t acc = (size_t)hs.max_blk_nnz * sizeof(double);
        // operator + mass in one sweep (a.vals2): both accumulator ranges of every block must fit the LDS, else two sweeps as before
        // ... and the second range must not cost occupancy: measured (tools/asm_fuse_ab.py) 2-D P1 C2 0.073 -> 0.046 ms (40 KB per workgroup), but
        // C3 1.11 -> 1.10 ms (80 KB: one or two workgroups per CU instead of three -- the sweep is bound by its memory traffic, 5.9 TB/s of
        // 2 FETCH + WRITE, and fewer waves hide less of it), 3-D P2 0.75 -> 0.75: fused up to 64 KB (knob asm_fuse_mass 2: whenever it fits)
        // knob asm_fuse_mass: 0 never; 1 (default) two accumulator ranges up to 64 KB, else a second pass in the same launch; 2 two ranges
        // wherever they fit; 3 always the second pass
        const int fm = c->asm_fuse_mass;
        const bool can = a.vals2 != nullptr && a.vals != nullptr && (opk == 1 || opk == 3) && fm != 0;
        const size_t fuse_cap = fm == 2 ? (size_t)c->lds_limit : (size_t)64 * 1024;
        const bool fuse_mass = can && fm != 3 && tab + 2 * acc <= fuse_cap;
        const bool seq_mass = can && !fuse_mass && fm != 2 && tab + acc <= (size_t)c->lds_limit;   // (every block's range must fit the LDS)
        // (the visit-parallel sweep below runs the mass matrix as its second pass for the space-varying integrands as well)
        const bool items_mass = R == 2 && c->asm_items && c->lane_row.p != nullptr && (opk == 4 || opk == 5);
        if (a.vals2 != nullptr && !fuse_mass && !seq_mass && !items_mass) return FDAPDE_EUNSUPPORTED;   // (e_init then runs the mass sweep of its own)
        if (tab + acc > (size_t)c->lds_limit) acc = tab < (size_t)c->lds_limit ? (size_t)c->lds_limit - tab : 0;
        // visit-parallel form (k_assemble_items) for spaces whose rows were dealt to the lane positions by visit count (P2): the (row, visit)
        // pairs of a block are the work items of 1024 threads (512 for the space-varying integrand: registers), same addends in the same
        // order as the row-walking kernel, hence the same bits.  One block per CU with 16 wavefronts instead of 4: its accumulators may take
        // the whole LDS.  Knob asm_items = 0 keeps the row-walking kernel.
        if constexpr (R == 2) {
            const size_t acc_all = (size_t)hs.max_blk_nnz * sizeof(double), lds_items = tab + acc_all + (size_t)hs.max_blk_cells * 8;
            const bool want_mass2 = a.vals2 != nullptr;
            if (c->asm_items && c->lane_row.p != nullptr && opk != 0 && lds_items + 10 * 1024 <= (size_t)160 * 1024 && (!want_mass2 || opk == 1 || opk == 3 || opk == 4 || opk == 5)) {
                if (c->asm_max_visits < 0) {   // longest visit list of the space (once per space: fdapde_dofs_build resets it)
                    c->asm_max_visits = hs.max_slice_width;
                }
                if (c->asm_max_visits <= kItemsMaxVisits) {
                    a.lane_row = c->lane_row.p, a.lds_acc_cap = hs.max_blk_nnz, a.lds_cells = hs.max_blk_cells;
                    c->asm_all_in_lds = true;
                    if (a.fq != nullptr && a.fq == c->fq.p && c->fq_blk_ready) a.fq = c->fq_blk.p, a.fq_block = 1;
                    else if (a.fq != nullptr && a.fq == c->fq.p && c->fq_bc_ready) a.fq = c->fq_bc.p, a.fq_block = 2;
                    const int grid_i = 8 * (int)(((hs.n_dofs + kAsmBlock - 1) / kAsmBlock + 7) / 8);
                    if (std::getenv("FDAPDE_DEBUG_ASM"))
                        std::fprintf(stderr, "assembly launch <%d,%d> opk %d, visit-parallel: grid %d x %d threads, LDS %zu B dynamic (tables %zu + accumulators %zu), longest visit list %d, %s\n",
                                     M, R, opk, grid_i, opk == 4 ? 512 : 1024, lds_items, tab, acc_all, c->asm_max_visits,
                                     !want_mass2 ? "one matrix" : (c->asm_items_fuse && (opk == 3 || opk == 1) && lds_items + acc_all + 4 * 1024 <= (size_t)160 * 1024) ? "operator and mass in ONE sweep (second accumulator range)" : "mass as second sweep");
#define ITEMS_GO(OPK_, M2_, TH_)                                                                                                          \
    do {                                                                                                                                  \
        const void* fn = reinterpret_cast<const void*>(&k_assemble_items<M, R, OPK_, M2_, TH_>);                                          \
        if (lds_items > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_items);             \
        hipLaunchKernelGGL((k_assemble_items<M, R, OPK_, M2_, TH_>), dim3(grid_i), dim3(TH_), lds_items, c->stream, a, op);               \
    } while (0)
                    const char* th_env = std::getenv("FDAPDE_ASM_ITEMS_THREADS");   // (measurements: 512 instead of 1024 threads per block)
                    // both matrices in ONE sweep where a second accumulator range still fits the CU's LDS next to the first (3-D P2: 2 x 65 KB + 23 KB of
                    // tables + the kernel's 3 KB of static arrays: the block owns the CU anyway): the mass rows come from the same geometry, index words
                    // and accumulation rounds instead of a second sweep over the block's items (knob asm_items_fuse 0: the second sweep)
                    const size_t lds_two = lds_items + acc_all;
                    if (want_mass2 && c->asm_items_fuse && (opk == 3 || opk == 1) && lds_two + 4 * 1024 <= (size_t)160 * 1024 && !(th_env && std::atoi(th_env) == 512)) {
                        const size_t lds_one = lds_items;
                        (void)lds_one;
#define ITEMS_GO2(OPK_)                                                                                                                   \
    do {                                                                                                                                  \
        const void* fn = reinterpret_cast<const void*>(&k_assemble_items<M, R, OPK_, 1, 1024>);                                           \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_two);                                          \
        hipLaunchKernelGGL((k_assemble_items<M, R, OPK_, 1, 1024>), dim3(grid_i), dim3(1024), lds_two, c->stream, a, op);                 \
    } while (0)
                        if (opk == 3) ITEMS_GO2(3);
                        else ITEMS_GO2(1);
#undef ITEMS_GO2
                        HIPCHK(c, hipGetLastError());
                        return FDAPDE_OK;
                    }
                    if (want_mass2 && th_env && std::atoi(th_env) == 512) {
                        if (opk == 3) ITEMS_GO(3, 2, 512);
                        else ITEMS_GO(1, 2, 512);
                    } else if (want_mass2) {
                        if (opk == 3) ITEMS_GO(3, 2, 1024);
                        else if (opk == 5) ITEMS_GO(5, 2, 1024);
                        else if (opk == 4) ITEMS_GO(4, 2, 512);
                        else ITEMS_GO(1, 2, 1024);
                    } else if (opk == 4) {
                        if (th_env && std::atoi(th_env) == 1024) ITEMS_GO(4, 0, 1024);
                        else ITEMS_GO(4, 0, 512);
                    }
                    else if (opk == 5) {   // (node loop rolled: 107 registers, none spilled -- C5-size -Lap + c(x): 1024 threads 5.8 ms, 512: 7.4)
                        if (th_env && std::atoi(th_env) == 512) ITEMS_GO(5, 0, 512);
                        else ITEMS_GO(5, 0, 1024);
                    }
                    else if (opk == 3) ITEMS_GO(3, 0, 1024);
                    else if (opk == 2) ITEMS_GO(2, 0, 1024);
                    else ITEMS_GO(1, 0, 1024);
#undef ITEMS_GO
                    HIPCHK(c, hipGetLastError());
                    return FDAPDE_OK;
                }
            }
        }
        if (a.vals2 != nullptr && !fuse_mass && !seq_mass) return FDAPDE_EUNSUPPORTED;   // (items_mass, but the visit-parallel sweep does not take this space)
        a.lds_acc_cap = (int32_t)(acc / sizeof(double));
        c->asm_all_in_lds = acc == (size_t)hs.max_blk_nnz * sizeof(double);   // every block accumulates in LDS (AsmArgs::row_stat is then complete)
        if (a.fq != nullptr && a.fq == c->fq.p && c->fq_blk_ready) a.fq = c->fq_blk.p, a.fq_block = 1;   // column 0: one load coefficient per visit slot
        else if (a.fq != nullptr && a.fq == c->fq.p && c->fq_bc_ready) a.fq = c->fq_bc.p, a.fq_block = 2;   // column 0: samples in block-cell order
        const int grid = 8 * (int)(((hs.n_dofs + kAsmBlock - 1) / kAsmBlock + 7) / 8);   // 8 XCD bands of blocks (k_assemble_rows)
        size_t lds = tab + (fuse_mass ? 2 : 1) * acc;
        if (std::getenv("FDAPDE_DEBUG_ASM"))
            std::fprintf(stderr, "assembly launch <%d,%d> opk %d: grid %d x %d, LDS %zu B (tables %zu + accumulators %zu%s), max block nnz %d nodes %d cells %d, %s\n", M, R, opk,
                         grid, kAsmBlock, lds, tab, acc, fuse_mass ? " x 2" : "", hs.max_blk_nnz, hs.max_blk_nodes, hs.max_blk_cells,
                         fuse_mass ? "mass fused (two ranges)" : seq_mass ? "mass as second pass" : "operator only");
        if (fuse_mass) {
            if (lds > 64 * 1024) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_rows<M, R, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_rows<M, R, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            }
            if (opk == 3) hipLaunchKernelGGL((k_assemble_rows<M, R, 3, 1>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
            else hipLaunchKernelGGL((k_assemble_rows<M, R, 1, 1>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
            HIPCHK(c, hipGetLastError());
            return FDAPDE_OK;
        }
        if (seq_mass) {
            if (lds > 64 * 1024) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_rows<M, R, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_rows<M, R, 3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            }
            if (opk == 3) hipLaunchKernelGGL((k_assemble_rows<M, R, 3, 2>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
            else hipLaunchKernelGGL((k_assemble_rows<M, R, 1, 2>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
            HIPCHK(c, hipGetLastError());
            return FDAPDE_OK;
        }
        if (lds > 64 * 1024)
            for (const void* fn : {reinterpret_cast<const void*>(&k_assemble_rows<M, R, 0>),
                                   reinterpret_cast<const void*>(&k_assemble_rows<M, R, 1>),
                                   reinterpret_cast<const void*>(&k_assemble_rows<M, R, 2>),
                                   reinterpret_cast<const void*>(&k_assemble_rows<M, R, 3>),
                                   reinterpret_cast<const void*>(&k_assemble_rows<M, R, 4>),
                                   reinterpret_cast<const void*>(&k_assemble_rows<M, R, 5>)})
                (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (opk == 4)
            hipLaunchKernelGGL((k_assemble_rows<M, R, 4>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
        else if (opk == 5)
            hipLaunchKernelGGL((k_assemble_rows<M, R, 5>), dim3(grid), dim3(kAsmBlock), lds, c->stream, a, op);
        else if (opk == 3)
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
            if (op.kt_sym) a.reftab = reinterpret_cast<const DevRefTensors*>(c->reftab_sym.p), a.ref_doubles = kRefSymDoubles;
            else a.reftab = c->reftab.p, a.ref_doubles = kRefDoubles;
            const size_t lds = sizeof(DevTables) + (opk == 3 ? sizeof(double) * (size_t)a.ref_doubles : 0);
#define PART_GO(K_)                                                                                                            \
    hipLaunchKernelGGL((k_assemble_part<M, R, K_>), dim3((unsigned)c->n_parts), dim3(256), lds, c->stream, a, op, c->part_cells.p, \
                       c->part_off.p, c->part_colours, c->part_shared.p, c->part_slots.p)
            if (opk == 3) PART_GO(3);
            else if (opk == 1) PART_GO(1);
            else if (opk == 2) PART_GO(2);
            else PART_GO(0);
#undef PART_GO
        } else if (assembly == FDAPDE_ASSEMBLY_WAVE) {
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
                hipLaunchKernelGGL((k_assemble_wave<M, R>), dim3((unsigned)((cnt + 3) / 4)), dim3(256), sizeof(DevTables), c->stream, a, op,
                                   c->colour_cells.p + o0, c->wave_slots.p + (size_t)o0 * NB * NB, cnt);
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

// DevOp::mirror: the assembled matrix as the reference's assembler would have left it (k_mirror_reference_lower)
int mirror_reference_lower(fdapde_ctx* c, double* vals) {
    if (c->comm != nullptr || c->ar_fn != nullptr || c->halo_ready || c->rd.ready)
        return fail(c, FDAPDE_EUNSUPPORTED, "a non-symmetric diffusion tensor in an expression without advection (which the reference assembles as a symmetric "
                                            "operator: lower triangle by DOF id, mirrored) is supported on single-GPU contexts only: the ranks of a multi-GPU job "
                                            "number their DOFs independently");
    const HostSpace& hs = c->hs;
    DBuf<double> tmp;
    HIPCHK(c, tmp.alloc((size_t)hs.nnz));
    hipLaunchKernelGGL(k_mirror_reference_lower, dim3((unsigned)((hs.n_dofs + 255) / 256)), dim3(256), 0, c->stream, hs.n_dofs, c->rowptr.p, c->colidx.p, c->dof_i2e.p,
                       vals, tmp.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(vals, tmp.p, sizeof(double) * (size_t)hs.nnz, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (tmp goes out of scope)
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

int e_set_operator(fdapde_ctx* c, int32_t n_terms, const fdapde_term* terms) {
    if (!c) return FDAPDE_EINVAL;
    std::vector<HostTerm> t;
    bool sym = true;
    int rc = check_terms(c, n_terms, terms, &t, &sym);
    if (rc) return rc;
    c->op = std::move(t), c->op_symmetric = sym, c->coef_of_op = false, c->matrix_dirty = true;
    c->assembled[0] = false, c->solved = false, c->cg_broke_down = false;
    return FDAPDE_OK;
}

int e_set_forcing(fdapde_ctx* c, const double* f_q, int32_t n_cols) {
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

int e_set_dirichlet(fdapde_ctx* c, const double* g) {
    if (!c) return FDAPDE_EINVAL;
    if (!c->space_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    const HostSpace& hs = c->hs;
    c->solved = false;
    if (!g) {
        c->have_g = false, c->g_i.clear();
        return FDAPDE_OK;
    }
    bool all_zero = true;
    if (c->has_device && c->dev_ready) {   // as handed over to the device, permuted to the internal DOF order there (a serial host gather through
                                           // dof_i2e before: 6 ms at C3's size; the permutation's host mirror is not needed any more)
        for (int64_t i = 0; i < hs.n_dofs && all_zero; ++i) all_zero = g[i] == 0.0;
        c->g_i.clear();
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, c->g.alloc((size_t)hs.n_dofs));
        HIPCHK(c, hipMemcpyAsync(c->tmp_e.p, g, sizeof(double) * (size_t)hs.n_dofs, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_f64, dim3(g1(hs.n_dofs)), dim3(256), 0, c->stream, hs.n_dofs, c->dof_i2e.p, c->tmp_e.p, c->g.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));   // (the caller's buffer is free again when the call returns)
    } else {
        if (int rc = ensure_host(c, kHostPerm)) return rc;
        c->g_i.resize((size_t)hs.n_dofs);
        for (int64_t i = 0; i < hs.n_dofs; ++i) {
            c->g_i[(size_t)i] = g[hs.dof_i2e[(size_t)i]];
            all_zero = all_zero && c->g_i[(size_t)i] == 0.0;
        }
        if (c->has_device) {
            HIPCHK(c, hipSetDevice(c->device));
            HIPCHK(c, c->g.upload(c->g_i.data(), c->g_i.size(), c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    c->have_g = true, c->g_zero = all_zero;
    return FDAPDE_OK;
}

int e_assemble_operator(fdapde_ctx* c, int32_t which, int32_t n_terms, const fdapde_term* terms, int32_t assembly) {
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
    if (which == FDAPDE_MAT_STIFF) c->stiff_stat_valid = false;   // (the row statistics belong to what fdapde_init assembled)
    rc = launch_assembly(c, a, op, assembly);
    if (rc) return rc;
    if (op.mirror)
        if (int rc2 = mirror_reference_lower(c, a.vals)) return rc2;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->assembled[which] = true;
    if (which == FDAPDE_MAT_STIFF) c->op_symmetric = sym, c->solved = false, c->dirichlet_applied = false, c->cg_broke_down = false, ++c->init_count, c->matrix_dirty = true;   // (whatever was derived from the old values is stale)
    return FDAPDE_OK;
}

int e_init(fdapde_ctx* c, const fdapde_options* opt) {
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
        const int64_t n_slices = hs.n_slices;
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
    c->stiff_stat_valid = false;
    if (assembly == FDAPDE_ASSEMBLY_ROWS && c->asm_row_stat) {   // (complete only if every block accumulates in LDS: asm_all_in_lds)
        HIPCHK(c, c->stiff_stat.alloc(2 * (size_t)hs.n_dofs));
        a.diag = c->diag.p, a.row_stat = c->stiff_stat.p;
    }
    // the mass matrix in the same sweep where both accumulator ranges fit the LDS (P1 blocks); otherwise (EUNSUPPORTED) a sweep of its own
    bool mass_done = false;
    if (assembly == FDAPDE_ASSEMBLY_ROWS && c->asm_fuse_mass) {
        a.vals2 = c->vals[FDAPDE_MAT_MASS].p;
        rc = launch_assembly(c, a, op, assembly);
        if (rc == FDAPDE_OK) mass_done = true;
        else if (rc != FDAPDE_EUNSUPPORTED) return rc;
        a.vals2 = nullptr;
    }
    if (!mass_done) {
        rc = launch_assembly(c, a, op, assembly);
        if (rc) return rc;
    }
    bool stat_complete = a.row_stat != nullptr && c->asm_all_in_lds;   // (as the operator's launch found it; later launches overwrite the flag)
    if (op.mirror) {   // (rare: a non-symmetric diffusion tensor without advection)
        if (int rc2 = mirror_reference_lower(c, c->vals[FDAPDE_MAT_STIFF].p)) return rc2;
        stat_complete = false;   // (the row maxima belong to the matrix before the mirroring)
    }
    if (c->fq_cols == 0) HIPCHK(c, hipMemsetAsync(c->force.p, 0, sizeof(double) * (size_t)hs.n_dofs, c->stream));
    for (int col = 1; col < c->fq_cols; ++col) {   // remaining time columns (parabolic forcing), fem_solver_base.h:124-128
        AsmArgs f = asm_args(c);
        f.fq = c->fq.p + (size_t)col * rows, f.force = c->force.p + (size_t)col * hs.n_dofs;
        rc = launch_assembly(c, f, op, assembly);
        if (rc) return rc;
    }
    // mass_ = discretize_operator(Reaction(1.0)): fem_solver_base.h:136
    if (!mass_done) {
        AsmArgs m = asm_args(c);
        m.vals = c->vals[FDAPDE_MAT_MASS].p;
        rc = launch_assembly(c, m, mass_op, assembly);
        if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->info.t_assemble_ms = ms;
    c->stiff_stat_valid = stat_complete;
    c->assembled[0] = c->assembled[1] = true, c->force_ready = true, c->solved = false, c->dirichlet_applied = false, c->cg_broke_down = false;
    if (c->matrix_dirty || !c->last_init_rows || assembly != FDAPDE_ASSEMBLY_ROWS) ++c->init_count;   // (else: the same values again, bit for bit)
    c->matrix_dirty = false, c->last_init_rows = assembly == FDAPDE_ASSEMBLY_ROWS;
    return FDAPDE_OK;
}

// pointwise_evaluation::eval (basis/lagrangian_basis.h:203-235): locate + evaluate.  The bin grid over the cells' bounding
// boxes is built on the host per call (index work, like the reference's KD-tree build at first use, tree_search.h:47-62).
int e_eval_pointwise(fdapde_ctx* c, int64_t n_locs, const double* locs_colmajor, int32_t* cell_ids, double* values) {
    if (!c || n_locs < 1 || !locs_colmajor || !cell_ids || !values) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int M = hs.M;
    hipStream_t st = c->stream;
    // uniform bin grid over the mesh, built on the device once per mesh (dev_setup.hip dev_build_bin_grid; it was a serial host loop over
    // the cells on every call: 80 % of a call with 10^6 locations on a 10^6-cell mesh)
    fdapde_ctx::EvalGrid& eg = c->eval_grid;
    if (!eg.ready) {
        DevBinGrid g;
        if (int rc = dev_build_bin_grid(M, hs.n_nodes, hs.n_cells, c->vcoords.p, c->cverts.p, st, &g, c->err)) return rc;
        adopt(eg.ptr, g.bin_ptr, (size_t)g.n_bins + 1), adopt(eg.cells, g.bin_cells, (size_t)g.n_entries + 1);
        HIPCHK(c, eg.dims.upload(g.dims, 3, st));
        HIPCHK(c, eg.lo.upload(g.lo, 3, st));
        HIPCHK(c, eg.invh.upload(g.inv_h, 3, st));
        HIPCHK(c, hipStreamSynchronize(st));   // (g's small arrays live on this stack frame)
        eg.ready = true;
    }
    HIPCHK(c, c->eval_locs.upload(locs_colmajor, (size_t)n_locs * M, st));
    HIPCHK(c, c->eval_out.alloc((size_t)n_locs));
    HIPCHK(c, c->eval_vals.alloc((size_t)n_locs * hs.nb));
    AsmArgs a = asm_args(c);
    const double tol = 1e-12;
    const dim3 grid(g1(n_locs)), block(256);
#define EVAL_GO(MM, RR)                                                                                                  \
    hipLaunchKernelGGL((k_eval_pointwise<MM, RR>), grid, block, 0, st, a, n_locs, c->eval_locs.p, eg.lo.p, eg.invh.p, eg.dims.p, eg.ptr.p, \
                       eg.cells.p, c->cell_i2e.p, tol, c->eval_out.p, c->eval_vals.p)
    if (M == 2 && hs.order == 1) EVAL_GO(2, 1);
    else if (M == 2) EVAL_GO(2, 2);
    else if (hs.order == 1) EVAL_GO(3, 1);
    else EVAL_GO(3, 2);
#undef EVAL_GO
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(cell_ids, c->eval_out.p, sizeof(int32_t) * (size_t)n_locs, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(values, c->eval_vals.p, sizeof(double) * (size_t)n_locs * hs.nb, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDAPDE_OK;
}

// ingredients of areal_evaluation::eval (basis/lagrangian_basis.h:238-283): per-cell measure and integrals of the local basis
int e_cell_integrals(fdapde_ctx* c, double* measure, double* psi_int) {
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

int e_quadrature_nodes(fdapde_ctx* c, double* out) {
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


// the unit's code object is loaded when one of its kernels is first looked up (HIP defers it): done at context creation, so that the
// first solve of a process does not pay for it (6 ms for the smoke problem after the library was split into units)
void preload_assembly() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_quadrature_nodes<3>));
    (void)hipGetLastError();
}

}   // namespace fdapde_engine
