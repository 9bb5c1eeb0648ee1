// eng_solve.hip -- the multi-launch solve driver: SpMV dispatch, solver layouts (compact CSR pattern, blocked ELL), Jacobi scaling and
// Dirichlet reduction (solve_prepare), the Krylov loops (solve_run: fused-update / single-reduction / textbook CG, BiCGStab; hand-over to
// the single-launch solver of persist_engine.hip), elliptic / parabolic / factor-once entry points, result getters, SpMV benchmarks.
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

namespace fdapde_engine {

constexpr int64_t kBlockedShortRowsAbove = 3200000;   // short-row systems (P1) of more DOFs than this take the blocked-ELL SpMV on the multi-launch path

// the captured CG chunk bakes pointers and sizes in: drop it whenever a layout, buffer or knob may have changed
void drop_graph(fdapde_ctx* c) {
    if (c->cg_graph_exec) (void)hipGraphExecDestroy(c->cg_graph_exec);
    c->cg_graph_exec = nullptr;
}

// e0 / e1 (optional): HIP events attached to the dispatch itself (hipExtLaunchKernelGGL), i.e. the kernel's own begin / end
// timestamps on the stream it runs on -- the same interval rocprofv3 --kernel-trace reports, with no extra marker packet
// between the neighbouring kernels.
// y = (I + offdiag) x from the blocked-ELL layout of boundary variant v with the values its ell_val holds (k_spmv_blocked; y = x on the rows it leaves out)
void launch_spmv_blocked(fdapde_ctx* c, int v, const double* x, double* y, const double* w, double* partial, const int32_t* stop, hipEvent_t e0, hipEvent_t e1,
                         int dot2_ww) {
    {
        const fdapde_ctx::Blocked& bk = c->bk[v];
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
    }
}

void launch_spmv(fdapde_ctx* c, const double* vals, const double* x, double* y, const double* w, double* partial,
                 const int32_t* stop, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, int dot2_ww = 0,
                 const uint8_t* owned = nullptr) {
    if (vals == c->sval.p && c->bk_cur >= 0 && owned == nullptr) {   // the solver's scaled matrix in blocked-ELL form (k_spmv_blocked)
        launch_spmv_blocked(c, c->bk_cur, x, y, w, partial, stop, e0, e1, dot2_ww);
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

// default iteration bound: 10 n capped at 100 000 -- but ONE number for all ranks of a row-distributed solve (the launches stop by it and
// advance their epoch tags by it: n is the rank's own DOF count there)
inline int default_maxit(const fdapde_ctx* c, int64_t n) { return c->rd.ready ? 100000 : (int)(10 * n < 100000 ? 10 * n : 100000); }

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
int solve_prepare(fdapde_ctx* c, const double* A, int use_bnd, SolveState* ss, bool symmetric, bool allow_defer = false) {
    const int64_t n = c->hs.n_dofs;
    hipStream_t st = c->stream;
    ss->dist = (c->comm != nullptr || c->ar_fn != nullptr) && c->halo_ready;   // multi-GPU: sub-assembled operator of this rank's cells
    ss->owned = ss->dist ? c->owned.p : nullptr;
    ss->use_bnd = use_bnd, ss->symmetric = symmetric;
    c->sval_stale = false;
    ss->rowdist = (c->comm != nullptr || c->ar_fn != nullptr) && c->rd.ready && !ss->dist;
    if (ss->rowdist) ss->owned = c->rd.owned.p;
    // small systems on one GPU (the sizes the reference is used at: a 289-DOF solve is a 130 us launch inside 250 us of wall time): no wait for
    // the "every interior diagonal is positive" flag -- it is raised at ctl[4] and read back with the solve's outcome; the solve goes ahead as if
    // the diagonal were positive (what an assembled elliptic operator has) and fdapde_solve repeats it the slow way in the rare other case
    ss->diag_deferred = allow_defer && !ss->dist && !ss->rowdist && c->small_rows > 0 && n <= c->small_rows && c->comm == nullptr && c->ar_fn == nullptr;
    int32_t* diag_flag = c->ctl.p + (ss->diag_deferred ? 4 : 3);
    ss->front_pending = false, ss->front_A = A;
    // ... and the smallest of them (one workgroup): flag reset, scale and fill are left to k_small_front (solve_run), if the layout turns out to be one it takes
    bool front_candidate = ss->diag_deferred && (symmetric || c->persist_bicg) && c->small_front_rows > 0 && n <= c->small_front_rows && A == c->vals[FDAPDE_MAT_STIFF].p &&
                           c->stiff_stat_valid && c->persist && !c->persist_broken && c->spmv_variant == 2;
    if (!front_candidate) HIPCHK(c, hipMemsetAsync(c->ctl.p, 0, 8 * sizeof(int32_t), st));
    if (front_candidate) {
    } else if (ss->dist) {   // the diagonal is a sum over the ranks sharing a DOF
        hipLaunchKernelGGL(k_diag_extract, dim3(g1(n)), dim3(256), 0, st, n, c->diag.p, A, c->tmp_i.p);
        if (int rc = halo_sum(c, c->tmp_i.p, nullptr, 0)) return rc;
        hipLaunchKernelGGL(k_jacobi_scale_from_diag, dim3(g1(n)), dim3(256), 0, st, n, c->tmp_i.p, c->bnd.p, use_bnd, c->scale.p,
                           c->ctl.p + 3);
    } else {
        if (A == c->vals[FDAPDE_MAT_STIFF].p && c->stiff_stat_valid)   // (diagonal, row maximum) left by fdapde_init's sweep: same numbers, 16 B per row
            hipLaunchKernelGGL(k_jacobi_scale_stats, dim3(g1(n)), dim3(256), 0, st, n, c->stiff_stat.p, c->bnd.p, use_bnd, c->scale.p, diag_flag);
        else
            hipLaunchKernelGGL(k_jacobi_scale, dim3(g1(n * 16)), dim3(256), 0, st, n, c->rowptr.p, c->diag.p, A, c->bnd.p, use_bnd, c->scale.p, diag_flag);
    }
    if (ss->diag_deferred) {
        c->h_ctl[3] = 0;
    } else {
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
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
    if (front_candidate) {
        const fdapde_ctx::Persist& ps = c->ps[use_bnd ? 1 : 0];
        if (persist && ps.meta.G == 1 && !ps.meta.sym && ps.ell_col.n >= (size_t)ps.meta.n_entries + 256) {
            ss->front_pending = true;
        } else {   // not this time (e.g. the layout's column table is made by its first fill): the separate launches after all
            front_candidate = false;
            HIPCHK(c, hipMemsetAsync(c->ctl.p, 0, 8 * sizeof(int32_t), st));
            hipLaunchKernelGGL(k_jacobi_scale_stats, dim3(g1(n)), dim3(256), 0, st, n, c->stiff_stat.p, c->bnd.p, use_bnd, c->scale.p, diag_flag);
        }
    }
    // one GPU, positive diagonal, not taken by the persistent CG (non-symmetric operator, or too many rows): the multi-launch
    // kernels apply the operator from the blocked-ELL layout (k_spmv_blocked); the compact CSR pattern is then not built either
    c->bk_cur = -1, c->bk[0].filled = c->bk[1].filled = false;
    bool blocked = false;
    // ... where it pays: long rows (P2).  On 14-entry rows (C3 with the persistent CG switched off) the CSR kernel's finer-grained,
    // software-pipelined workgroups win (45 us against 48-51 us per SpMV), so short-row systems keep the compact CSR pattern.
    // Round 6: ... up to the size where the single launch ends (3.1 M rows on 256 CUs).  Above it every operator application streams from HBM and the
    // blocked form wins on short rows too -- 3-D P1, 8.1 M DOFs: 238 us against 281 us per SpMV, 0.67 against 0.56 of the HBM peak on the layout's own
    // bytes (tools/large_ab.py, profiles/r6_large_ab.txt).
    const bool long_rows = (double)c->hs.nnz >= 20.0 * (double)n || c->blocked == 2 || n > kBlockedShortRowsAbove;
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
    } else if (persist && (c->persist_fill_fused || ss->diag_deferred)) {   // the launch's blocks are filled straight from A (fill_persist_scaled): no scaled copy
                                                                           // now (small systems: one launch less in front of the solve; ensure_sval makes it on demand)
        c->sval_stale = true, c->sval_A = A;
        c->sp_cur = -1, c->sval_layout = -2;
    } else {
        hipLaunchKernelGGL(k_scale_matrix, dim3(g1(n * 16)), dim3(256), 0, st, n, c->rowptr.p, c->colidx.p, A, c->scale.p, c->sval.p);
        c->sp_cur = -1, c->sval_layout = -2;
    }
    if (blocked) {
        const int v = use_bnd ? 1 : 0;
        if (c->bk[v].meta.n_entries > 0)
            hipLaunchKernelGGL(k_persist_fill, dim3((unsigned)((c->bk[v].meta.n_entries + 1023) / 1024)), dim3(256), 0, st, c->bk[v].meta.n_entries, c->bk[v].ell_src.p,
                           c->sval.p, c->bk[v].ell_val.p, (unsigned long long*)nullptr);
        c->bk[v].filled = true, c->bk_cur = v;
    }
    if (persist && ss->front_pending) c->ps[use_bnd ? 1 : 0].filled = true;   // (by k_small_front, in stream order before the launch)
    else if (persist)
        if (int rc = c->sval_stale ? fill_persist_scaled(c, use_bnd ? 1 : 0, A) : fill_persist(c, use_bnd ? 1 : 0)) return rc;
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}

// the scaled full-pattern copy of the current system, if solve_prepare left it out (single-launch solve)
int ensure_sval(fdapde_ctx* c) {
    if (!c->sval_stale) return FDAPDE_OK;
    const int64_t n = c->hs.n_dofs;
    hipLaunchKernelGGL(k_scale_matrix, dim3(g1(n * 16)), dim3(256), 0, c->stream, n, c->rowptr.p, c->colidx.p, c->sval_A, c->scale.p, c->sval.p);
    HIPCHK(c, hipGetLastError());
    c->sval_stale = false;
    return FDAPDE_OK;
}

// "CG broke down" (p.Ap <= 0) can only be said of a CG: the breakdown word of a BiCGStab or GMRES stage that followed means something else
static inline bool is_cg_method(int m) { return m == FDAPDE_SOLVER_CG || m == FDAPDE_SOLVER_CG_SR || m == FDAPDE_SOLVER_CG_FUSED; }

// Restarted GMRES(m) on the scaled system (kernels_gmres.h); x, r, sc, ctl as k_krylov_init / _fin left them, bt = the scaled right-hand side.
// Leaves x (scaled iterate), sc[0] / sc[3] (|b|^2, |b - A x|^2 TRUE) and ctl like the other methods; h_ctl / h_sc hold the last read-back.
int run_gmres(fdapde_ctx* c, double tol2, int maxit) {
    const int64_t n = c->hs.n_dofs;
    const int m = c->gmres_m;
    hipStream_t st = c->stream;
    HIPCHK(c, c->gm_V.alloc((size_t)(m + 1) * (size_t)n));
    HIPCHK(c, c->gm_s.alloc((size_t)gm_state_doubles(m) + (size_t)m + 2));
    const int vg = c->vec_grid;
    // [(m + 1) x kGmStripes stripes of V^T w | one |w|^2 partial per workgroup of the vector kernels]: vec_grid goes up to 1024 workgroups
    // (systems above 65 536 DOFs), the true-residual partials of k_gm_residual need as many from the start of the buffer
    const size_t gm_norm_at = (size_t)(m + 1) * kGmStripes;
    HIPCHK(c, c->gm_part.alloc(gm_norm_at + (size_t)std::max(vg, kGmStripes)));
    double* gm_norm = c->gm_part.p + gm_norm_at;
    double* gs = c->gm_s.p;
    double* h_pass = gs + gm_state_doubles(m);   // coefficients of the Gram-Schmidt pass in flight
    double* hcol = gs + 3 * m + 1;
    double* w = c->y.p;
    HIPCHK(c, hipMemsetAsync(c->ctl.p + 5, 0, sizeof(int32_t), st));   // the stall counter of k_gm_cycle_fin (a word of its own: ctl[3] belongs to the Jacobi scaling / the single launch)
    HIPCHK(c, hipMemcpyAsync(c->sc.p + 21, c->sc.p + 3, sizeof(double), hipMemcpyDeviceToDevice, st));   // |r|^2 at the start of the first cycle
    bool stop = false;
    HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    stop = c->h_ctl[0] != 0;   // (converged at the initial guess)
    while (!stop) {
        hipLaunchKernelGGL(k_gm_cycle_init, dim3(vg), dim3(256), 0, st, n, m, c->r.p, c->sc.p, gs, c->gm_V.p, c->ctl.p);
        for (int j = 0; j < m; ++j) {
            const double* vj = c->gm_V.p + (size_t)j * n;
            launch_spmv(c, c->sval.p, vj, w, nullptr, c->part_a.p, c->ctl.p);   // w = At v_j (leaves at once when the stop flag is up)
            for (int pass = 0; pass < 2; ++pass) {                              // classical Gram-Schmidt, twice
                hipLaunchKernelGGL(k_gm_dots, dim3(kGmStripes, j + 1), dim3(256), 0, st, n, c->gm_V.p, w, c->gm_part.p, c->ctl.p);
                hipLaunchKernelGGL(k_gm_reduce, dim3(j + 1), dim3(256), 0, st, c->gm_part.p, h_pass, hcol, pass, c->ctl.p);
                hipLaunchKernelGGL(k_gm_axpy, dim3(vg), dim3(256), 0, st, n, j + 1, c->gm_V.p, h_pass, w, gm_norm, pass, c->ctl.p);
            }
            hipLaunchKernelGGL(k_gm_hess, dim3(1), dim3(256), 0, st, j, m, gm_norm, vg, gs, c->sc.p, c->ctl.p, tol2, maxit);
            hipLaunchKernelGGL(k_gm_next, dim3(vg), dim3(256), 0, st, n, w, gs, m, c->gm_V.p + (size_t)(j + 1) * n, c->ctl.p);
            if (j % 10 == 9 && j + 1 < m) {   // a look at the stop flag every ten steps: the rest of a cycle whose estimate has converged (or whose
                                              // budget is spent) would be ~8 empty launches per step
                HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                if (c->h_ctl[0] != 0) break;
            }
        }
        // end of the cycle: y, x += V y, the true residual -- which decides whether another cycle follows
        hipLaunchKernelGGL(k_gm_solve_y, dim3(1), dim3(1), 0, st, m, gs);
        hipLaunchKernelGGL(k_gm_update_x, dim3(vg), dim3(256), 0, st, n, m, c->gm_V.p, gs, c->x.p);
        launch_spmv(c, c->sval.p, c->x.p, c->t.p, nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(k_gm_residual, dim3(vg), dim3(256), 0, st, n, c->gm_b.p, c->t.p, c->r.p, c->gm_part.p);
        hipLaunchKernelGGL(k_gm_cycle_fin, dim3(1), dim3(256), 0, st, c->gm_part.p, vg, c->sc.p, c->ctl.p, tol2, maxit);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(c->h_sc, c->sc.p, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        stop = c->h_ctl[0] != 0;
    }
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
    // nothing of an earlier solve may be read as this one's outcome by a caller that sees an early return (solve_run_restarting)
    c->h_ctl[0] = c->h_ctl[1] = c->h_ctl[2] = c->h_ctl[3] = 0, c->h_ctl_seen = 4;
    c->info.iters = 0, c->info.method_used = method, c->info.relres = 0, c->info.converged = 0;
    // partial pairs the SpMV leaves for the vector kernels: one per workgroup of the kernel that applies the scaled operator
    const int np_spmv = (c->bk_cur >= 0 && !dist) ? c->bk[c->bk_cur].meta.G : c->spmv_grid;
    const double* fvec = f_dev;
    if (ss.rowdist) {
        // (warm starts: the caller's initial guess must hold the owners' values at the ghost columns -- the parabolic stepper imports them
        //  after every step, rowdist_import_ghosts)
        const bool want_bicg = method == FDAPDE_SOLVER_BICGSTAB || !ss.symmetric;
        if (method == FDAPDE_SOLVER_CG_SR) return fail(c, FDAPDE_EUNSUPPORTED, "the row-distributed solve runs the fused-update CG or BiCGStab");
        method = want_bicg ? FDAPDE_SOLVER_BICGSTAB : FDAPDE_SOLVER_CG_FUSED;
    }
    if (dist) {   // the forcing vector is a sum over the ranks sharing a DOF
        HIPCHK(c, hipMemcpyAsync(c->tmp_e.p, f_dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
        if (int rc = halo_sum(c, c->tmp_e.p, nullptr, 0)) return rc;
        fvec = c->tmp_e.p;
    }
    // the lift is zero (no Dirichlet data, or homogeneous data on one GPU -- across ranks the data may differ, and every rank
    // must take the same path through the collectives): A g~ = 0, written by the lift kernel itself (no memset launch)
    const bool zero_lift = !ss.use_bnd || (!dist && !ss.rowdist && g_dev == c->g.p && c->g_zero);
    // the smallest systems (solve_prepare: front_pending): everything in front of the single launch as ONE one-workgroup kernel, or two around A g~
    const bool front = ss.front_pending;
    ss.front_pending = false;
    c->front_used = front;
    SmallFrontArgs fa{};
    if (front) {
        const fdapde_ctx::Persist& ps = c->ps[ss.use_bnd ? 1 : 0];
        fa.n = n, fa.use_bnd = ss.use_bnd, fa.zero_y = zero_lift ? 1 : 0, fa.vec_grid = c->vec_grid;
        fa.stat = c->stiff_stat.p, fa.bnd = c->bnd.p, fa.scale = c->scale.p, fa.ctl = c->ctl.p;
        fa.nsl = ps.meta.nsl, fa.ell_off = ps.ell_off.p, fa.sl_off = ps.sl_off.p, fa.slot_dof = ps.slot_dof.p, fa.src = ps.ell_src.p, fa.col = ps.ell_col.p;
        fa.A = ss.front_A, fa.ell_val = ps.ell_val.p;
        fa.g = g_dev, fa.gt = c->gt.p, fa.y = c->y.p, fa.u = c->u.p;
        if (!zero_lift) {
            fa.phases = 1;
            hipLaunchKernelGGL(k_small_front, dim3(1), dim3(256), 0, st, fa);
        }
    } else
        hipLaunchKernelGGL(k_lift, dim3(g1(n)), dim3(256), 0, st, n, c->bnd.p, g_dev, ss.use_bnd, c->gt.p, zero_lift ? c->y.p : (double*)nullptr);
    if (!zero_lift) {
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
    const bool gmres = method == FDAPDE_SOLVER_GMRES;
    if (gmres && (dist || ss.rowdist)) return fail(c, FDAPDE_EUNSUPPORTED, "GMRES runs on one-GPU contexts");
    const double tol2 = rtol * rtol;
    const double* ax = nullptr;
    if (u0_dev) {   // warm start: x = (u0 - g~) / s, r = b~ - At x
        if (int rc = ensure_sval(c)) return rc;
        hipLaunchKernelGGL(k_krylov_init, dim3(c->vec_grid), dim3(256), 0, st, n, fvec, c->y.p, c->scale.p, c->x.p, c->r.p, c->p.p,
                           (double*)nullptr, c->part_b.p, owned, u0_dev, c->gt.p, (const double*)nullptr, 1);
        launch_spmv(c, c->sval.p, c->x.p, c->t.p, nullptr, nullptr, nullptr);
        if (dist)
            if (int rc = halo_sum(c, c->t.p, nullptr, 0)) return rc;
        ax = c->t.p;
    }
    if (!front)
        hipLaunchKernelGGL(k_krylov_init, dim3(c->vec_grid), dim3(256), 0, st, n, fvec, c->y.p, c->scale.p, c->x.p, c->r.p, c->p.p,
                           bicg ? c->r0.p : (double*)nullptr, c->part_b.p, owned, u0_dev, c->gt.p, ax, 0);
    if (front) {   // (one GPU, cold start: what the two launches of the other branch do)
        const int V = c->cgf_v;
        const int64_t b2 = c->cgf_band ? (((((n + 7) / 8) + 31) & ~int64_t(31)) >> 1) : 0, span = b2 > 0 ? b2 : (n >> 1);
        const int per = (int)((span + 256 * V - 1) / (256 * V)) > 0 ? (int)((span + 256 * V - 1) / (256 * V)) : 1;
        const int cg = b2 > 0 ? 8 * per : per;
        fa.phases = zero_lift ? 3 : 2;
        fa.f = fvec, fa.x = c->x.p, fa.r = c->r.p, fa.p = c->p.p, fa.r0 = bicg ? c->r0.p : (double*)nullptr, fa.partial = c->part_b.p, fa.sc = c->sc.p;
        fa.seed = cgf ? c->part_b.p + cg : (double*)nullptr, fa.n_seed = cgf ? cg : 0, fa.tol2 = tol2;
        hipLaunchKernelGGL(k_small_front, dim3(1), dim3(256), 0, st, fa);
    } else if (dist || ss.rowdist) {
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
    if (bicg && c->bicg_shadow && !dist && !ss.rowdist) {   // a shadow residual other than r0 (knob; measurements): r0~ and rho_0 = (r0~, r0)
        hipLaunchKernelGGL(k_bicg_shadow, dim3(c->vec_grid), dim3(256), 0, st, n, c->bicg_shadow, c->r.p, c->r0.p, c->part_b.p);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, c->part_b.p, c->vec_grid, c->sc.p + 9);
    }
    if (gmres) {   // the scaled right-hand side, for the true residual at the end of every restart cycle (c->y still holds A g~)
        HIPCHK(c, c->gm_b.alloc((size_t)n));
        hipLaunchKernelGGL(k_gm_rhs, dim3(g1(n)), dim3(256), 0, st, n, fvec, c->y.p, c->scale.p, c->gm_b.p);
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
    bool unscaled = false;   // c->u already holds the solution (the single launch's epilogue ran behind it)
    if (gmres) {
        if (int rc = ensure_sval(c)) return rc;
        if (int rc = run_gmres(c, tol2, maxit)) return rc;
        stop = true, launched = c->h_ctl[1];
        HIPCHK(c, hipMemcpyAsync(c->h_sc, c->sc.p, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    if (ss.rowdist) {   // one launch per rank, the launches of all ranks acting as one grid (kernels_persist.h DIST)
        if (int rc = run_rowdist(c, ss.use_bnd ? 1 : 0, tol2, maxit, &persisted, bicg)) return rc;
        if (!persisted) return fail(c, FDAPDE_EUNSUPPORTED, "row-distributed solve: an in-kernel hand-off between the ranks' launches timed out (boards not visible across the devices, or a rank's launch could not be resident); use the element-partitioned exchange (fdapde_halo_setup_peers) instead");
        stop = true, launched = c->h_ctl[1];
    }
    if (cgf && !dist && !ss.rowdist && c->persist && !c->persist_broken && c->ps[ss.use_bnd ? 1 : 0].ok && c->ps[ss.use_bnd ? 1 : 0].filled) {
        // the whole iteration as ONE launch (kernels_persist.h); it leaves sc / ctl as the loop below would
        DebugClock clk;
        // the epilogue kernel (and, for fdapde_solve, the end-of-solve event) is enqueued BEHIND the launch before the host waits for it: one wait
        // for launch, outcome and solution together; if the launch gave up, the fall-back below redoes the epilogue from its own iterate
        c->persist_tail = [&]() -> int {
            if (!c->tail_in_launch) hipLaunchKernelGGL(k_unscale, dim3(g1(n)), dim3(256), 0, st, n, c->scale.p, c->persist_x.p, c->gt.p, c->u.p);
            if (c->ev1_at_end) HIPCHK(c, hipEventRecord(c->ev1, st));
            return FDAPDE_OK;
        };
        const int rc_p = run_persist(c, ss.use_bnd ? 1 : 0, tol2, maxit, &persisted);
        c->persist_tail = nullptr;
        if (rc_p) return rc_p;
        clk.mark("solve_run: run_persist");
        if (persisted) stop = true, launched = c->h_ctl[1], unscaled = true;
    }
    if (bicg && !dist && !ss.rowdist && c->persist && c->persist_bicg && !c->persist_broken) {   // the whole BiCGStab as one launch
        const fdapde_ctx::Persist& ps = c->ps[ss.use_bnd ? 1 : 0];
        if (ps.ok && ps.filled && !ps.meta.sym && ps.meta.R <= 8) {
            c->persist_tail = [&]() -> int {   // (as for the CG launch above)
                if (!c->tail_in_launch) hipLaunchKernelGGL(k_unscale, dim3(g1(n)), dim3(256), 0, st, n, c->scale.p, c->persist_x.p, c->gt.p, c->u.p);
                if (c->ev1_at_end) HIPCHK(c, hipEventRecord(c->ev1, st));
                return FDAPDE_OK;
            };
            const int rc_p = run_persist(c, ss.use_bnd ? 1 : 0, tol2, maxit, &persisted, /*bicg=*/true);
            c->persist_tail = nullptr;
            if (rc_p) return rc_p;
            if (persisted) stop = true, launched = c->h_ctl[1], unscaled = true;
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
    if (!stop && launched < maxit && !ss.rowdist)   // the multi-launch kernels read the scaled full-pattern copy
        if (int rc = ensure_sval(c)) return rc;
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
    if (!unscaled) {
        hipLaunchKernelGGL(k_unscale, dim3(g1(n)), dim3(256), 0, st, n, c->scale.p, persisted ? c->persist_x.p : c->x.p, c->gt.p, c->u.p);
        HIPCHK(c, hipGetLastError());
        if (c->ev1_at_end) HIPCHK(c, hipEventRecord(c->ev1, st));
        if (!c->defer_end_sync) HIPCHK(c, hipStreamSynchronize(st));   // (h_ctl / h_sc were read back behind a wait of their own)
    }
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

// solve_run, with a BiCGStab that BROKE DOWN (rho, r0.v or omega exactly 0 -- on advection-dominated operators the recurrences get there --
// or an iterate that stopped being finite) started again from the iterate it had reached: new shadow residual, the stop rule still relative to
// the original right-hand side (solve_run's warm start).  Up to kBicgRestarts times; the iterations add up in info.iters.  One-GPU contexts.
constexpr int kBicgRestarts = 30;
int solve_run_restarting(fdapde_ctx* c, const SolveState& ss, const double* A, const double* f_dev, const double* g_dev, const double* u0_dev,
                         int method, double rtol, int maxit, int check_every, int n_timed, int gmres_budget = -1) {
    // gmres_budget: -1 no GMRES stage (the caller named a method); 0: what BiCGStab left of `maxit` (a budget the caller set is a budget for the whole
    // call); > 0: that many iterations of its own (the default budget, per stage)
    const int64_t n = c->hs.n_dofs;
    // Where the open method has a later stage for this system (the dense direct stage, GMRES) and no budget of the caller's, BiCGStab's own stage is
    // capped at 2 n + 200 iterations instead of the default 10 n: a BiCGStab that has not converged by then has stalled or diverged (cell Peclet numbers
    // of 10^2 - 10^3: relres 1e-3 ... 1e+10 after 10 n iterations, profiles/r5_gmres_probe.txt), and the budget it burnt was most of the call's time
    // (1 089 DOFs, Pe 150: 43 ms of BiCGStab in front of a 10 ms inversion).  CG stages and budgets the caller set are untouched.
    const bool will_bicg = method == FDAPDE_SOLVER_BICGSTAB || (method == FDAPDE_SOLVER_AUTO && !(c->op_symmetric && ss.diag_positive));
    const bool later_stage = gmres_budget > 0 && !ss.dist && !ss.rowdist && (c->auto_gmres || dense_eligible(c));
    const int full_maxit = maxit;
    if (will_bicg && later_stage) {
        int64_t cap = 2 * n + 200;
        // ... and where that later stage is the dense direct solve, rent or buy: BiCGStab may spend what the inversion will cost (its estimate over ~3 us +
        // n / 700 us per iteration of the single launch: 550 iterations at 1 089 DOFs, 2 100 at 4 225), not more -- a system it has not solved by then
        // is solved directly, and the call costs at most twice the inversion (1 089 DOFs, cell Peclet 150 - 1 000: 20 ms -> 5)
        if (dense_eligible(c)) cap = std::min<int64_t>(cap, std::max<int64_t>(200, (int64_t)(1e3 * dense_build_estimate_ms(n) / (3.0 + (double)n / 700.0))));
        maxit = (int)std::min<int64_t>(maxit, cap);
    }
    (void)full_maxit;
    int rc = solve_run(c, ss, A, f_dev, g_dev, u0_dev, method, rtol, maxit, check_every, n_timed);
    int total = c->info.iters;
    const bool may_restart = method == FDAPDE_SOLVER_BICGSTAB || method == FDAPDE_SOLVER_AUTO;   // a method named explicitly is never replaced
    // (ranks of a multi-GPU job restart together: breakdown word, iteration count and residual come out of sums all of them hold bit for bit)
    for (int k = 0; k < kBicgRestarts && may_restart && rc == FDAPDE_ENOCONV && c->h_ctl[2] != 0 && c->info.method_used == FDAPDE_SOLVER_BICGSTAB &&
                    total < maxit && std::isfinite(c->info.relres) && c->info.relres < 1e3 && c->bicg_restart;   // (an iterate 1000 x worse than zero is no start)
         ++k) {
        if (ss.rowdist)   // the iterate at the columns other ranks own: the warm start reads them
            if (int rc2 = rowdist_import_ghosts(c, ss.use_bnd ? 1 : 0, c->u.p)) return rc2;
        HIPCHK(c, c->restart_u.alloc((size_t)n));
        HIPCHK(c, hipMemcpyAsync(c->restart_u.p, c->u.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
        rc = solve_run(c, ss, A, f_dev, g_dev, c->restart_u.p, FDAPDE_SOLVER_BICGSTAB, rtol, maxit - total, check_every, 0);
        total += c->info.iters;
    }
    c->info.iters = total;
    // The last stage where the caller left the method open (FDAPDE_SOLVER_AUTO): BiCGStab gave up -- restarts exhausted, stalled until maxit, or an
    // iterate that stopped being finite -- on a system the reference's LU would have solved (fem_linear_elliptic_solver.h:38-47; typically an
    // advection-dominated operator): restarted GMRES(m) on the same scaled system, from BiCGStab's iterate if that was any closer than zero.
    // ... but first, for a system small enough to invert (kernels_dense.h): a DIRECT solve of the reference's own row-zeroed matrix -- what its
    // SparseLU does, in milliseconds where restarted GMRES needs 10^4 - 10^5 iterations (cell Peclet numbers of 10^3 on a few thousand DOFs)
    // (a budget the caller set -- maxit -- and BiCGStab spent is spent for this stage too: the call reports what its iterations reached)
    if (gmres_budget >= 0 && (gmres_budget > 0 || maxit - total > 0) && rc == FDAPDE_ENOCONV && c->info.method_used == FDAPDE_SOLVER_BICGSTAB && !ss.dist && !ss.rowdist &&
        dense_eligible(c)) {
        bool solved = false;
        if (int rc2 = dense_direct(c, A, ss.use_bnd, f_dev, g_dev, &solved)) return rc2;
        if (solved) {
            c->info.method_used = FDAPDE_SOLVER_DENSE, c->info.converged = 1, c->info.relres = c->solve_dense.check, c->info.persistent = 0;
            c->h_ctl[0] = 1, c->h_ctl[2] = 0;
            c->err.clear();
            if (c->ev1_at_end) HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            return FDAPDE_OK;
        }
    }
    const int gm_maxit = gmres_budget > 0 ? gmres_budget : maxit - total;
    if (gmres_budget >= 0 && gm_maxit > 0 && c->auto_gmres && rc == FDAPDE_ENOCONV && c->info.method_used == FDAPDE_SOLVER_BICGSTAB && !ss.dist && !ss.rowdist) {
        const bool warm = std::isfinite(c->info.relres) && c->info.relres < 1.0;
        if (warm) {
            HIPCHK(c, c->restart_u.alloc((size_t)n));
            HIPCHK(c, hipMemcpyAsync(c->restart_u.p, c->u.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
        }
        rc = solve_run(c, ss, A, f_dev, g_dev, warm ? c->restart_u.p : u0_dev, FDAPDE_SOLVER_GMRES, rtol, gm_maxit, check_every, 0);
        c->info.iters += total;
    }
    return rc;
}

// One-time preparation of the solver's compact matrix layout for the current boundary-DOF mask (part of set-up, like
// fdapde_dofs_build; the first solve does it lazily otherwise).  with_dirichlet: the layout used when Dirichlet data are set.
int e_solver_prepare(fdapde_ctx* c, int32_t with_dirichlet) {
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
        ((double)c->hs.nnz >= 20.0 * (double)c->hs.n_dofs || c->blocked == 2 || c->hs.n_dofs > kBlockedShortRowsAbove)) {   // single GPU, long rows: the multi-launch kernels use the blocked-ELL layout
        if (int rc = build_blocked(c, v)) return rc;
        if (c->bk[v].ok) return FDAPDE_OK;
    }
    return build_solver_pattern(c, v);
}

// What the in-solve SpMV works on, for the roofline figures of bench.py: the interior block A_II as a plain CSR operator
// (rows / entries; algorithmic bytes = 12 nnz + 4 (n + 1) + 16 n on it) and the bytes one launch of the solver's kernel really
// streams from the compact coded layout (values 8 B + column codes 2 B per stored entry, row pointers, window bases, virtual-row
// table of a segmented pattern, x gathered once and y written once for every row of the full vector).
int e_solver_layout(fdapde_ctx* c, int32_t with_dirichlet, int64_t* n_interior, int64_t* nnz_interior, double* streamed_bytes) {
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
            *streamed_bytes = 10.0 * (double)c->ps[v].meta.n_entries + 32.0 * (double)c->ps[v].meta.n_board +
                              (c->ps[v].meta.R > kPersistRmax ? 16.0 * (double)c->ps[v].meta.n_int : 0.0);   // (wide form: x read and written once per row)
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
int e_solver_layout_kind(fdapde_ctx* c, int32_t with_dirichlet, int32_t* kind, int32_t* symmetric_storage, int32_t* workgroups,
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

int e_solve(fdapde_ctx* c, const fdapde_options* opt, fdapde_info* info) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->assembled[0] || !c->force_ready)
        return fail(c, FDAPDE_ENOTINIT, "solver must be initialized first!");   // fem_linear_elliptic_solver.h:36
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n = c->hs.n_dofs;
    const double rtol = (opt && opt->rtol > 0) ? opt->rtol : 1e-10;
    const int maxit = (opt && opt->maxit > 0) ? opt->maxit : default_maxit(c, n);
    const int check_every = (opt && opt->check_every > 0) ? opt->check_every : 32;
    const double* A = c->vals[FDAPDE_MAT_STIFF].p;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (opt && opt->method == FDAPDE_SOLVER_PMG) return e_solve_pmg(c, opt, info);   // the two-level solver of order-2 spaces (eng_pmg.hip), by name
    if (opt && opt->method == FDAPDE_SOLVER_DENSE) {
        // the direct solve asked for by name (what the reference's SparseLU does, fem_linear_elliptic_solver.h:38-47): the reference's own row-zeroed matrix
        // inverted on the device, one product; no Krylov stage in front, no fall-back behind -- a singular matrix is reported (success = false)
        if (!dense_eligible(c))
            return fail(c, FDAPDE_EUNSUPPORTED, "FDAPDE_SOLVER_DENSE takes one-GPU systems of up to `dense_rows` (at most 8192) DOFs");
        bool solved = false;
        c->scaled_owner = fdapde_ctx::kScaledNone;
        if (int rc = dense_direct(c, A, c->have_g ? 1 : 0, c->force.p, c->g.p, &solved)) return rc;
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        HIPCHK(c, hipEventSynchronize(c->ev1));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        const double t_asm = c->info.t_assemble_ms;   // (fdapde_init's figure stays with the record)
        c->info = fdapde_info{};
        c->info.t_assemble_ms = t_asm;
        c->info.method_used = FDAPDE_SOLVER_DENSE, c->info.converged = solved ? 1 : 0, c->info.relres = c->solve_dense.check, c->info.t_solve_ms = ms;
        c->solved = solved, c->dirichlet_applied = c->have_g;
        if (info) *info = c->info;
        if (!solved) {
            c->err = "FDAPDE_SOLVER_DENSE: the matrix is singular to working precision (no usable pivot, or max |I - A X| beyond 1e-6)";
            return FDAPDE_ENOCONV;
        }
        return FDAPDE_OK;
    }
    // The open method on a LARGE order-2 system it qualifies for: the two-level solver first (eng_pmg.hip: ~19 iterations whatever the mesh size, where the
    // Jacobi-preconditioned stages below need O(1 / h) -- C5, 5.36 M DOFs: 64 ms against 617; the two meet near 200 - 300 k DOFs: 3-D 185 k 9.3 against 8.6 ms,
    // 389 k 13 against 21; 2-D, symmetric, 315 k 27 against 25).  Whatever it does not solve falls through to the stages below.
    // (rent-or-buy: below `pmg_auto_first_rows` the context's first open-method solve keeps the Jacobi stages -- the coarse level is built by the second)
    if ((!opt || opt->method == FDAPDE_SOLVER_AUTO) && c->pmg_auto && n >= c->pmg_auto_rows &&
        (n >= c->pmg_auto_first_rows || c->pmg.ready || c->open_solves >= 1) && pmg_eligible(c)) {
        fdapde_options po{};
        po.method = FDAPDE_SOLVER_PMG, po.rtol = rtol, po.maxit = (opt && opt->maxit > 0) ? std::min(opt->maxit, 60) : 60;   // (it converges in two dozen iterations or not at all)
        const int rc = e_solve_pmg(c, &po, info);
        if (rc == FDAPDE_OK) {
            ++c->open_solves;
            return rc;
        }
        if (rc != FDAPDE_ENOCONV && rc != FDAPDE_EUNSUPPORTED) return rc;
        c->err.clear();
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    }
    SolveState ss;
    c->scaled_owner = fdapde_ctx::kScaledSolve;
    DebugClock clk;
    const bool open_method = !opt || opt->method == FDAPDE_SOLVER_AUTO;
    const bool skip_cg = open_method && c->cg_broke_down;   // (this very matrix broke CG before: see below)
    const int gm_budget = !open_method ? -1 : ((opt && opt->maxit > 0) ? 0 : default_maxit(c, n));   // (the GMRES stage of the open method)
    if (int rc = solve_prepare(c, A, c->have_g ? 1 : 0, &ss, c->op_symmetric && !skip_cg, /*allow_defer=*/true)) return rc;
    clk.mark("fdapde_solve: solve_prepare");
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));   // (a solve that ends before its epilogue -- an early refusal -- still leaves a recorded event)
    c->ev1_at_end = true;
    struct Ev1Guard {
        fdapde_ctx* c;
        ~Ev1Guard() { c->ev1_at_end = false; }
    } ev1_guard{c};
    int rc = solve_run_restarting(c, ss, A, c->force.p, c->g.p, nullptr, skip_cg ? FDAPDE_SOLVER_BICGSTAB : (opt ? opt->method : FDAPDE_SOLVER_AUTO), rtol,
                                  maxit, check_every, opt ? opt->time_spmv : 0, gm_budget);
    clk.mark("fdapde_solve: solve_run");
    if (ss.diag_deferred) {   // the flag the solve did not wait for
        if (c->h_ctl_seen < 5) {   // (an outcome read-back that did not carry it: fetch it now)
            HIPCHK(c, hipMemcpyAsync(c->h_ctl + 4, c->ctl.p + 4, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (c->h_ctl[4] != 0) {   // a non-positive interior diagonal after all: everything again with the decision taken up front
            if (int rc2 = solve_prepare(c, A, c->have_g ? 1 : 0, &ss, c->op_symmetric && !skip_cg, false)) return rc2;
            rc = solve_run_restarting(c, ss, A, c->force.p, c->g.p, nullptr, skip_cg ? FDAPDE_SOLVER_BICGSTAB : (opt ? opt->method : FDAPDE_SOLVER_AUTO), rtol,
                                      maxit, check_every, opt ? opt->time_spmv : 0, gm_budget);
        }
    }
    // (the ranks of a multi-GPU job take this turn TOGETHER: the breakdown word comes out of sums every rank holds bit for bit, and the ranks say so to
    //  each other before any of them re-prepares -- a rank that disagreed would leave the others in the collectives of the second solve)
    bool cg_broke = rc == FDAPDE_ENOCONV && c->h_ctl[2] != 0 && (!opt || opt->method == FDAPDE_SOLVER_AUTO) && is_cg_method(c->info.method_used);
    if ((ss.dist || ss.rowdist) && (!opt || opt->method == FDAPDE_SOLVER_AUTO) && (rc == FDAPDE_OK || rc == FDAPDE_ENOCONV)) {
        int yes = 0;
        if (int rc2 = ranks_saying_yes(c, cg_broke, &yes)) return rc2;
        if (yes != 0 && yes != c->world) return fail(c, FDAPDE_ERCCL, "fdapde_solve: the ranks disagree on the outcome of the CG stage");
        cg_broke = yes == c->world;
    }
    if (cg_broke) {
        // CG broke down (p.Ap <= 0): the operator is symmetric but not positive definite -- e.g. 3-D P2 with a large reaction term: the
        // reference's 5-point rule has a negative weight, its mass matrix is indefinite (integrator_tables.h:275-292).  The reference's LU solves
        // such a system all the same (fem_linear_elliptic_solver.h:38-47); so does BiCGStab.  Only where the caller left the method open.
        c->cg_broke_down = true;   // (until the matrix is assembled again)
        if (int rc2 = solve_prepare(c, A, c->have_g ? 1 : 0, &ss, false)) return rc2;
        rc = solve_run_restarting(c, ss, A, c->force.p, c->g.p, nullptr, FDAPDE_SOLVER_BICGSTAB, rtol, maxit, check_every, opt ? opt->time_spmv : 0, gm_budget);
        clk.mark("fdapde_solve: solve_run (BiCGStab after a CG breakdown)");
    }
    {
        if (rc == FDAPDE_OK && open_method && c->cg_broke_down && !ss.dist && !ss.rowdist) {
            // A symmetric operator that broke CG is indefinite -- or SINGULAR (a pure Neumann problem): on a matrix that is singular up to rounding
            // BiCGStab "converges" to a solution with a huge multiple of the null vector in it (-Lap u = 1 without Dirichlet data: |u| ~ 1e14, residual
            // 1e-11 by the recurrence -- a number that cannot be checked: the rounding error of A u alone is eps |A| |u| >> tol |b|).  The reference's LU
            // reports failure on such a matrix (fem_linear_elliptic_solver.h:42-45).  Guard: |x| / |b| of the scaled system (unit diagonal, |A~| ~ 1)
            // beyond 1e12 is reported as success = false, never handed out as a solution.
            const double* xs = c->info.persistent ? c->persist_x.p : c->x.p;
            hipLaunchKernelGGL(k_sq_norm, dim3(c->vec_grid), dim3(256), 0, c->stream, n, xs, c->part_b.p);
            hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, c->stream, c->part_b.p, c->vec_grid, c->sc.p + 20);
            double xx = 0;
            HIPCHK(c, hipMemcpyAsync(&xx, c->sc.p + 20, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const double bb = c->h_sc[0];
            if (!(xx <= 1e24 * bb)) {
                c->info.converged = 0;
                c->err = "operator numerically singular: the iterate grew beyond 1e12 |b| (e.g. no Dirichlet DOF and a right-hand side outside the range)";
                rc = FDAPDE_ENOCONV;
            }
        }
    }
    if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
    HIPCHK(c, hipEventSynchronize(c->ev1));   // (recorded by solve_run behind its epilogue kernel: already reached, its final wait came after it)
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->info.t_solve_ms = ms;
    c->solved = true, c->dirichlet_applied = c->have_g;
    if (open_method) ++c->open_solves;
    if (info) *info = c->info;
    return rc;
}

// The coarse level of the two-level solver (eng_pmg.hip) is solved ~50 times per fine solve, every time for the SAME matrix: what fdapde_solve does in front
// of its Krylov stage (Jacobi scale, scaled copy, the launch's layout filled: ~0.15 ms of kernels and two waits at 681 k rows) is done once per coarse
// operator, as the parabolic stepper does for its steps; a solve is then solve_run on the context's load vector.  Same method choice as fdapde_solve's open
// method (CG for a symmetric operator with a positive diagonal, BiCGStab after a CG breakdown or else), without its direct / GMRES stages.
int coarse_prepare(fdapde_ctx* c, SolveState* ss) {
    HIPCHK(c, hipSetDevice(c->device));
    c->scaled_owner = fdapde_ctx::kScaledSolve;
    return solve_prepare(c, c->vals[FDAPDE_MAT_STIFF].p, c->have_g ? 1 : 0, ss, c->op_symmetric && !c->cg_broke_down, false);
}

int coarse_solve(fdapde_ctx* c, SolveState* ss, double rtol, int maxit, fdapde_info* info) {
    const double* A = c->vals[FDAPDE_MAT_STIFF].p;
    int method = c->cg_broke_down ? FDAPDE_SOLVER_BICGSTAB : FDAPDE_SOLVER_AUTO;
    int rc = solve_run_restarting(c, *ss, A, c->force.p, c->g.p, nullptr, method, rtol, maxit, 32, 0, 0);
    if (rc == FDAPDE_ENOCONV && c->h_ctl[2] != 0 && method == FDAPDE_SOLVER_AUTO && is_cg_method(c->info.method_used)) {   // (see fdapde_solve)
        c->cg_broke_down = true;
        if (int rc2 = solve_prepare(c, A, c->have_g ? 1 : 0, ss, false, false)) return rc2;
        rc = solve_run_restarting(c, *ss, A, c->force.p, c->g.p, nullptr, FDAPDE_SOLVER_BICGSTAB, rtol, maxit, 32, 0, 0);
    }
    if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
    c->solved = true, c->dirichlet_applied = c->have_g;
    if (info) *info = c->info;
    return rc;
}

// FEMLinearParabolicSolver::solve (fdaPDE/finite_elements/solvers/fem_linear_parabolic_solver.h:37-72): implicit Euler,
//   K = M / dt + A ; Dirichlet rows of K ; for i = 0 .. m-2:  rhs = (M / dt) u_i + f_{i+1} ; rhs[boundary] = g(., i+1) ;
//   u_{i+1} = K^{-1} rhs.   The reference factorises K once with SparseLU; here K is scaled once and every step is a
//   Jacobi-PCG (or BiCGStab) solve warm-started from u_i.  Forcing columns come from fdapde_set_forcing (n_times columns).
int e_solve_parabolic(fdapde_ctx* c, const fdapde_options* opt, int32_t n_times, double delta_t, const double* initial_condition,
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
    const int maxit = (opt && opt->maxit > 0) ? opt->maxit : default_maxit(c, n);
    const int check_every = (opt && opt->check_every > 0) ? opt->check_every : 8;
    const double inv_dt = 1.0 / delta_t;
    DBuf<double> kmat, uprev, rhs, gcol;
    HIPCHK(c, kmat.alloc((size_t)hs.nnz + 2));
    HIPCHK(c, uprev.alloc((size_t)n));
    HIPCHK(c, rhs.alloc((size_t)n));
    HIPCHK(c, gcol.alloc((size_t)n));
    std::vector<double> tmp((size_t)n);
    if (int rc = ensure_host(c, kHostPerm)) return rc;
    auto to_internal = [&](const double* ext) {
        for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = ext[hs.dof_i2e[(size_t)i]];
    };
    HIPCHK(c, hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(k_matrix_combine, dim3(g1(hs.nnz)), dim3(256), 0, st, hs.nnz, c->vals[FDAPDE_MAT_MASS].p,
                       c->vals[FDAPDE_MAT_STIFF].p, inv_dt, kmat.p);
    // K is fixed over the steps: the reference factorises it ONCE and back-substitutes per step (fem_linear_parabolic_solver.h:41,56-68).  A small K is
    // inverted once here (kernels_dense.h) and a step is two products -- M u_i and K^-1 rhs -- with nothing returning to the host inside the loop.
    // (a warm-started Krylov step of such a system costs ~0.2 ms: worth it when the steps add up to half an inversion)
    const bool dense_named = opt && opt->method == FDAPDE_SOLVER_DENSE;
    if (dense_named && !dense_eligible(c))
        return fail(c, FDAPDE_EUNSUPPORTED, "FDAPDE_SOLVER_DENSE takes one-GPU systems of up to `dense_rows` (at most 8192) DOFs");
    if ((!opt || opt->method == FDAPDE_SOLVER_AUTO || dense_named) && dense_eligible(c) &&
        (dense_named || c->dense_after == 0 || (n_times - 1 > c->dense_after && 0.2 * (n_times - 1) >= 0.5 * dense_build_estimate_ms(n)))) {
        fdapde_ctx::Dense& D = c->step_dense;
        D.ready = D.failed = false;
        if (int rc = dense_build(c, kmat.p, dirichlet ? 1 : 0, D)) return rc;
        if (D.ready) {
            DBuf<double> d_dir, d_sol;
            const size_t cols = (size_t)n_times;
            HIPCHK(c, d_sol.alloc((size_t)n * cols));
            if (dirichlet) HIPCHK(c, d_dir.upload(dirichlet, (size_t)n * cols, st));
            to_internal(initial_condition);
            HIPCHK(c, hipMemcpyAsync(uprev.p, tmp.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));   // (tmp is pageable)
            if (!D.refine && c->dense_fold) {   // ONE product per step: u_{i+1} = (K^-1 M / dt) u_i + K^-1 (f_{i+1}, g_{i+1})
                if (int rc = dense_step_loop(c, D, n_times, inv_dt, dirichlet ? d_dir.p : nullptr, uprev.p, d_sol.p)) return rc;
            } else
            for (int32_t i = 0; i + 1 < n_times; ++i) {
                launch_spmv(c, c->vals[FDAPDE_MAT_MASS].p, uprev.p, c->s.p, nullptr, nullptr, nullptr);   // M u_i
                dense_step_rhs(c, c->s.p, inv_dt, c->force.p + (size_t)(i + 1) * n, dirichlet ? d_dir.p + (size_t)(i + 1) * n : nullptr, rhs.p);   // (+ rhs[boundary] = g(., i + 1), line 66)
                if (int rc = dense_apply(c, D, 1, rhs.p, c->u.p)) return rc;
                dense_step_out(c, c->u.p, uprev.p, d_sol.p + (size_t)(i + 1) * n);
            }
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(solution + (size_t)n, d_sol.p + (size_t)n, sizeof(double) * (size_t)n * (cols - 1), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(c->ev1, st));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            std::memcpy(solution, initial_condition, sizeof(double) * (size_t)n);   // solution_.col(0) = initial condition (line 46)
            float ms_d = 0;
            HIPCHK(c, hipEventElapsedTime(&ms_d, c->ev0, c->ev1));
            c->info.t_solve_ms = ms_d, c->info.iters = D.refine ? n_times - 1 : 0, c->info.relres = D.check, c->info.converged = 1;
            c->info.method_used = FDAPDE_SOLVER_DENSE, c->info.persistent = 0, c->info.launch_ms = 0;
            c->scaled_owner = fdapde_ctx::kScaledNone;
            D.X.release(), D.ready = false;   // (K belongs to this call)
            if (info) *info = c->info;
            kmat.release(), uprev.release(), rhs.release(), gcol.release();
            return FDAPDE_OK;
        }
        if (dense_named) {
            c->info = fdapde_info{};
            c->info.method_used = FDAPDE_SOLVER_DENSE, c->info.relres = D.check;
            if (info) *info = c->info;
            c->err = "FDAPDE_SOLVER_DENSE: M / dt + A is singular to working precision (no usable pivot, or max |I - A X| beyond 1e-6)";
            return FDAPDE_ENOCONV;
        }
    }
    // Large order-2 systems (or FDAPDE_SOLVER_PMG by name): every step through the two-level solver (eng_pmg.hip) -- the coarse operator is the P1 assembly of
    // the same terms + M1 / dt, built once for the call; a step starts from the previous column.  A first step it does not solve sends the open method to
    // the Jacobi-preconditioned loop below.
    const bool pmg_named = opt && opt->method == FDAPDE_SOLVER_PMG;
    if (pmg_named && !pmg_eligible(c))
        return fail(c, FDAPDE_EUNSUPPORTED, "FDAPDE_SOLVER_PMG takes one-GPU contexts and order-2 spaces");
    if (pmg_named || ((!opt || opt->method == FDAPDE_SOLVER_AUTO) && c->pmg_auto && n >= c->pmg_auto_rows &&
                      (n >= c->pmg_auto_first_rows || c->pmg.ready || c->open_solves >= 1 || n_times > 5) && pmg_eligible(c))) {
        const int pm_maxit = pmg_named ? ((opt && opt->maxit > 0) ? opt->maxit : 400) : ((opt && opt->maxit > 0) ? std::min(opt->maxit, 60) : 60);
        to_internal(initial_condition);
        HIPCHK(c, hipMemcpyAsync(uprev.p, tmp.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));
        int total = 0, rc_pm = FDAPDE_OK;
        double worst_pm = 0;
        bool gave_up = false;
        for (int32_t i = 0; i + 1 < n_times; ++i) {
            launch_spmv(c, c->vals[FDAPDE_MAT_MASS].p, uprev.p, c->s.p, nullptr, nullptr, nullptr);   // M u_i
            hipLaunchKernelGGL(k_parabolic_rhs, dim3(g1(n)), dim3(256), 0, st, n, c->s.p, inv_dt, c->force.p + (size_t)(i + 1) * n, rhs.p);
            if (dirichlet) {
                HIPCHK(c, hipMemcpyAsync(c->tmp_i.p, dirichlet + (size_t)(i + 1) * n, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_gather_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->tmp_i.p, gcol.p);
            }
            const int rc = pmg_run(c, kmat.p, rhs.p, gcol.p, dirichlet ? 1 : 0, uprev.p, inv_dt, 2 * c->init_count + 1, rtol, pm_maxit);
            if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
            if (rc == FDAPDE_ENOCONV && i == 0 && !pmg_named) {   // not a system for it: the loop below, from the start
                gave_up = true;
                c->err.clear();
                break;
            }
            if (rc == FDAPDE_ENOCONV) rc_pm = rc;
            total += c->info.iters, worst_pm = std::max(worst_pm, c->info.relres);
            HIPCHK(c, hipMemcpyAsync(uprev.p, c->u.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(k_scatter_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->u.p, c->tmp_e.p);
            HIPCHK(c, hipMemcpyAsync(solution + (size_t)(i + 1) * n, c->tmp_e.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));   // (tmp_e is reused by the next step)
        }
        if (!gave_up) {
            std::memcpy(solution, initial_condition, sizeof(double) * (size_t)n);   // solution_.col(0) = initial condition (line 46)
            HIPCHK(c, hipEventRecord(c->ev1, st));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            float ms_p = 0;
            HIPCHK(c, hipEventElapsedTime(&ms_p, c->ev0, c->ev1));
            c->info.t_solve_ms = ms_p, c->info.iters = total, c->info.relres = worst_pm, c->info.converged = rc_pm == FDAPDE_OK ? 1 : 0;
            c->info.method_used = FDAPDE_SOLVER_PMG, c->info.persistent = 0;
            c->scaled_owner = fdapde_ctx::kScaledNone;
            if (!pmg_named) c->open_solves += n_times - 1;
            if (info) *info = c->info;
            kmat.release(), uprev.release(), rhs.release(), gcol.release();
            return rc_pm;
        }
    }
    SolveState ss;
    c->scaled_owner = fdapde_ctx::kScaledParabolic;
    if (int rc = solve_prepare(c, kmat.p, dirichlet ? 1 : 0, &ss, c->op_symmetric)) return rc;
    to_internal(initial_condition);
    HIPCHK(c, hipMemcpyAsync(uprev.p, tmp.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    std::memcpy(solution, initial_condition, sizeof(double) * (size_t)n);   // solution_.col(0) = initial condition (line 46)
    int total_iters = 0, rc_all = FDAPDE_OK;
    double worst = 0;
    int step_method = opt ? opt->method : FDAPDE_SOLVER_AUTO;
    for (int32_t i = 0; i + 1 < n_times; ++i) {
        launch_spmv(c, c->vals[FDAPDE_MAT_MASS].p, uprev.p, c->s.p, nullptr, nullptr, nullptr);   // M u_i
        hipLaunchKernelGGL(k_parabolic_rhs, dim3(g1(n)), dim3(256), 0, st, n, c->s.p, inv_dt, c->force.p + (size_t)(i + 1) * n, rhs.p);
        if (dirichlet) {   // column i + 1 as handed over, brought into the internal order on the device (everything on the one stream: no wait here)
            HIPCHK(c, hipMemcpyAsync(c->tmp_i.p, dirichlet + (size_t)(i + 1) * n, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_gather_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->tmp_i.p, gcol.p);
        }
        c->defer_end_sync = true;   // (the step's outcome is read inside solve_run; what follows it is ordered by the stream)
        const int open_step = !(!opt || opt->method == FDAPDE_SOLVER_AUTO) ? -1 : ((opt && opt->maxit > 0) ? 0 : default_maxit(c, n));
        int rc = solve_run_restarting(c, ss, kmat.p, rhs.p, gcol.p, uprev.p, step_method, rtol, maxit, check_every, 0, open_step);
        bool cg_broke = rc == FDAPDE_ENOCONV && c->h_ctl[2] != 0 && step_method == FDAPDE_SOLVER_AUTO && is_cg_method(c->info.method_used);
        if ((ss.dist || ss.rowdist) && step_method == FDAPDE_SOLVER_AUTO && (rc == FDAPDE_OK || rc == FDAPDE_ENOCONV)) {   // (the ranks decide together: see fdapde_solve)
            int yes = 0;
            if (int rc2 = ranks_saying_yes(c, cg_broke, &yes)) return rc2;
            if (yes != 0 && yes != c->world) return fail(c, FDAPDE_ERCCL, "fdapde_solve_parabolic: the ranks disagree on the outcome of the CG stage");
            cg_broke = yes == c->world;
        }
        if (cg_broke) {   // M / dt + A symmetric but not positive definite (see fdapde_solve): this step again and every later one with BiCGStab
            if (int rc2 = solve_prepare(c, kmat.p, dirichlet ? 1 : 0, &ss, false)) return rc2;
            step_method = FDAPDE_SOLVER_BICGSTAB;
            rc = solve_run_restarting(c, ss, kmat.p, rhs.p, gcol.p, uprev.p, step_method, rtol, maxit, check_every, 0, open_step);
        }
        c->defer_end_sync = false;
        if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
        if (rc == FDAPDE_ENOCONV) rc_all = rc;
        total_iters += c->info.iters;
        worst = c->info.relres > worst ? c->info.relres : worst;
        if (ss.rowdist)   // u_{i+1} of the columns other ranks own: read by the next step's M u_i and warm start
            if (int rc2 = rowdist_import_ghosts(c, ss.use_bnd ? 1 : 0, c->u.p)) return rc2;
        HIPCHK(c, hipMemcpyAsync(uprev.p, c->u.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_scatter_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->u.p, c->tmp_e.p);
        HIPCHK(c, hipMemcpyAsync(solution + (size_t)(i + 1) * n, c->tmp_e.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->info.t_solve_ms = ms, c->info.iters = total_iters, c->info.relres = worst, c->info.converged = rc_all == FDAPDE_OK ? 1 : 0;
    if (!opt || opt->method == FDAPDE_SOLVER_AUTO) c->open_solves += n_times - 1;
    if (info) *info = c->info;
    kmat.release(), uprev.release(), rhs.release(), gcol.release();
    return rc_all;
}

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

// fdapde::SparseLU<SpMatrix<double>>::compute (fdaPDE/utils/symbols.h:142-146): "factorise" once.  Here: copy the matrix,
// Jacobi-scale it once; every later fdapde_lin_solve is a Krylov run on the prepared system.
int e_lin_compute(fdapde_ctx* c, int32_t which, const double* values, int32_t symmetric) {
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
    c->lin_dense.ready = c->lin_dense.failed = false, c->lin_cols = 0, c->lin_krylov_ms = 0;   // (a dense inverse belongs to the matrix it was built from)
    return FDAPDE_OK;
}

// fdapde::SparseLU::solve(b) (fdaPDE/utils/symbols.h:148-155), dense right-hand sides: b, x column-major n_dofs x n_rhs
int e_lin_solve(fdapde_ctx* c, const fdapde_options* opt, const double* b, int32_t n_rhs, double* x, fdapde_info* info) {
    if (!c || !b || !x || n_rhs < 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->lin_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_lin_compute first");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    const int64_t n = hs.n_dofs;
    {   // in-place solves (x overlapping b: Eigen's x = lu.solve(x) idiom): finished columns are written to x while later columns and a
        // BiCGStab retry still read b -- work from a private copy of the right-hand sides
        const double *b0 = b, *b1 = b + (size_t)n * n_rhs, *x0 = x, *x1 = x + (size_t)n * n_rhs;
        if (b0 < x1 && x0 < b1) {
            std::vector<double> b_copy(b0, b1);
            return e_lin_solve(c, opt, b_copy.data(), n_rhs, x, info);
        }
    }
    hipStream_t st = c->stream;
    const double rtol = (opt && opt->rtol > 0) ? opt->rtol : 1e-10;
    const int maxit = (opt && opt->maxit > 0) ? opt->maxit : default_maxit(c, n);
    const int check_every = (opt && opt->check_every > 0) ? opt->check_every : 32;
    int method = opt ? opt->method : FDAPDE_SOLVER_AUTO;
    if (method == FDAPDE_SOLVER_AUTO)
        method = (c->lin_symmetric && c->lin_state->ss.diag_positive) ? FDAPDE_SOLVER_CG_FUSED : FDAPDE_SOLVER_BICGSTAB;
    // "Factor once" that pays per solve (kernels_dense.h): a small system that has been asked for more than `dense_after` columns gets its dense
    // inverse -- built once, ~ms -- and every column from then on is ONE matrix-vector product, b and x through pinned memory.  Where the method was
    // left open; a method named explicitly runs as named.
    const bool dense_named = opt && opt->method == FDAPDE_SOLVER_DENSE;   // asked for by name: the inverse is built now, whatever the columns so far have cost
    if (dense_named && !dense_eligible(c))
        return fail(c, FDAPDE_EUNSUPPORTED, "FDAPDE_SOLVER_DENSE takes one-GPU systems of up to `dense_rows` (at most 8192) DOFs");
    if ((!opt || opt->method == FDAPDE_SOLVER_AUTO || dense_named) && dense_eligible(c)) {
        fdapde_ctx::Dense& D = c->lin_dense;
        if (!D.ready && (!D.failed || dense_named) &&
            (dense_named || c->dense_after == 0 || (c->lin_cols + n_rhs > c->dense_after && c->lin_krylov_ms >= 0.5 * dense_build_estimate_ms(n))))
            if (int rc = dense_build(c, c->lin_mat.p, 0, D)) return rc;
        if (dense_named && !D.ready) {
            c->info = fdapde_info{};
            c->info.method_used = FDAPDE_SOLVER_DENSE, c->info.relres = D.check;
            if (info) *info = c->info;
            c->err = "FDAPDE_SOLVER_DENSE: the matrix is singular to working precision (no usable pivot, or max |I - A X| beyond 1e-6)";
            return FDAPDE_ENOCONV;
        }
        if (D.ready) {
            const auto t0 = std::chrono::steady_clock::now();
            if (int rc = dense_solve_host(c, D, b, n_rhs, x)) return rc;
            c->lin_cols += n_rhs;
            c->info.iters = D.refine ? 1 : 0, c->info.converged = 1, c->info.relres = D.check, c->info.method_used = FDAPDE_SOLVER_DENSE, c->info.persistent = 0;
            c->info.launch_ms = 0, c->info.spmv_avg_ms = 0, c->info.spmv_timed = 0;
            c->info.t_solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (info) *info = c->info;
            return FDAPDE_OK;
        }
    }
    c->lin_cols += n_rhs;
    struct KrylovClock {   // host time of this call's Krylov columns, towards the handle's rent-or-buy decision
        fdapde_ctx* c;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~KrylovClock() { c->lin_krylov_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    } krylov_clock{c};
    if (c->scaled_owner != fdapde_ctx::kScaledLin) {   // an elliptic / parabolic solve in between has overwritten scale and the scaled copy
        c->scaled_owner = fdapde_ctx::kScaledNone;      // (whatever init / set_* calls followed it): prepare again (cheap)
        if (int rc = solve_prepare(c, c->lin_mat.p, 0, &c->lin_state->ss, c->lin_symmetric)) return rc;
        c->scaled_owner = fdapde_ctx::kScaledLin;
    }
    c->solved = false;   // c->u is about to hold the handle's solutions, not PDE::solution()
    DBuf<double>& rhs = c->lin_rhs;
    HIPCHK(c, rhs.alloc((size_t)n));
    DebugClock clk;
    HIPCHK(c, hipEventRecord(c->ev0, st));
    int total = 0, rc_all = FDAPDE_OK;
    double worst = 0;
    int32_t j0 = 0;
    bool breakdown = false;   // a CG column stopped on p.Ap <= 0
    // the handle was told "symmetric" and CG broke down: the matrix is not positive definite (3-D P2 mass matrices are not: negative quadrature
    // weight).  The reference's SparseLU does not care (utils/symbols.h:133-160): prepare the plain storage and solve every column with BiCGStab.
    auto retry_as_bicgstab = [&]() -> int {
        c->lin_symmetric = false, c->scaled_owner = fdapde_ctx::kScaledNone;
        if (int rc = solve_prepare(c, c->lin_mat.p, 0, &c->lin_state->ss, false)) return rc;
        c->scaled_owner = fdapde_ctx::kScaledLin, c->lin_sq_ready = false;
        const int spent = total;   // the CG iterations before the breakdown count
        c->lin_cols -= n_rhs;      // (the same columns again: counted once)
        const int rc = e_lin_solve(c, opt, b, n_rhs, x, info);
        c->info.iters += spent;
        if (info) info->iters = c->info.iters;
        return rc;
    };
    const bool may_retry = c->lin_symmetric && (!opt || opt->method == FDAPDE_SOLVER_AUTO) && method == FDAPDE_SOLVER_CG_FUSED &&
                           !c->lin_state->ss.dist && !c->lin_state->ss.rowdist;
    // several columns against a symmetric positive system on one GPU: batches of 8 / 4 columns share every pass over the
    // matrix (kernels_multirhs.h); what is left goes column by column
    // (a system the persistent CG takes is faster column by column -- one launch each, no vector traffic -- than batched through
    // the multi-launch SpMM: C3-size, 22.5 ms per column against 32 ms per column in a batch of 8)
    const bool persist_cols = c->persist && !c->persist_broken && c->ps[0].ok && c->ps[0].filled && c->ps[0].meta.R <= kPersistRmax;   // (the wide form: one column)
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
            if (!ok) rc_all = FDAPDE_ENOCONV, breakdown = breakdown || c->h_ctl[2] != 0;   // (lin_solve_batch leaves h_ctl of its last read-back)
            total += its, worst = rel > worst ? rel : worst;
            c->info.method_used = method;
            j0 += q;
        }
    }
    // several columns against a system the single-launch CG holds in few workgroups (the small systems the reference's users solve by the
    // thousand: one or two of 256 CUs busy per solve): Q columns side by side in ONE launch of G x Q workgroups, each column with boards of
    // its own; the same arithmetic per column as one by one, hence the same bits
    // ONE column against a system of one workgroup (a caller that cannot batch: 0.15 ms of wall time around a 50 us launch on the general
    // path -- upload, two small kernels, the launch, two read-backs, three waits): the launch does it all (run_persist_direct)
    if (n_rhs == 1 && persist_cols && c->persist_cols && c->persist_direct && method == FDAPDE_SOLVER_CG_FUSED && !c->lin_state->ss.dist &&
        !c->lin_state->ss.rowdist && !c->lin_state->ss.use_bnd && c->ps[0].meta.G == 1) {
        bool ran = false;
        const double tol2 = rtol * rtol;
        if (int rc = run_persist_direct(c, 0, tol2, maxit, b, x, &ran)) return rc;
        if (ran) {
            const double bb = c->h_sc[0], rr = c->h_sc[3];
            c->info.iters = c->h_ctl[1], c->info.relres = bb > 0 ? sqrt(rr / bb) : 0.0;
            c->info.converged = (rr <= tol2 * bb && c->h_ctl[2] == 0) ? 1 : 0;
            c->info.method_used = method, c->info.persistent = 1, c->info.launch_ms = 0.0, c->info.t_solve_ms = 0.0;
            c->info.spmv_avg_ms = 0, c->info.spmv_timed = 0;
            if (info) *info = c->info;
            if (!c->info.converged) {
                if (c->h_ctl[2] && may_retry) {
                    total = c->info.iters;
                    return retry_as_bicgstab();
                }
                c->err = c->h_ctl[2] ? "Krylov breakdown (operator not SPD for CG)" : "maxit reached";
                return FDAPDE_ENOCONV;
            }
            return FDAPDE_OK;
        }
    }
    const bool cols_bicg = method == FDAPDE_SOLVER_BICGSTAB && c->persist_bicg && !c->ps[0].meta.sym && c->ps[0].meta.R <= 8;   // (the single-launch BiCGStab's own limits)
    if (persist_cols && c->persist_cols && (method == FDAPDE_SOLVER_CG_FUSED || cols_bicg) && !c->lin_state->ss.dist && !c->lin_state->ss.rowdist &&
        !c->lin_state->ss.use_bnd) {
        fdapde_ctx::Persist& ps = c->ps[0];
        const int G = ps.meta.G;
        const int cap = (ps.per_cu > 0 ? ps.per_cu : 1) * c->n_cu / (G > 0 ? G : 1);   // columns whose workgroups are resident together
        const double tol2 = rtol * rtol;
        const int np = c->vec_grid;
        while (n_rhs - j0 >= 1 && cap >= 1) {   // (a single column too: three small launches around the solve instead of seven)
            const int Q = std::min(std::min<int>(n_rhs - j0, cap), 64);
            const size_t qn = (size_t)Q * (size_t)n;
            HIPCHK(c, c->cols_b.alloc(qn));
            HIPCHK(c, c->cols_r.alloc(qn));
            HIPCHK(c, c->cols_x.alloc(qn));
            HIPCHK(c, c->cols_sc.alloc(4 * (size_t)Q));
            HIPCHK(c, c->cols_ctl.alloc(4 * (size_t)Q));
            HIPCHK(c, c->cols_part.alloc(2 * (size_t)np * Q));
            HIPCHK(c, hipMemcpyAsync(c->cols_b.p, b + (size_t)j0 * n, sizeof(double) * qn, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_cols_init, dim3(np, Q), dim3(256), 0, st, n, c->cols_b.p, c->dof_i2e.p, c->scale.p, c->cols_r.p, c->cols_part.p);
            hipLaunchKernelGGL(k_cols_init_fin, dim3(Q), dim3(256), 0, st, c->cols_part.p, np, c->cols_sc.p, c->cols_ctl.p);
            HIPCHK(c, hipGetLastError());
            std::vector<int32_t> h_ctl(4 * (size_t)Q);
            std::vector<double> h_sc(4 * (size_t)Q);
            bool ran = false;
            if (int rc = run_persist_cols(c, 0, tol2, maxit, Q, c->cols_r.p, c->cols_x.p, c->cols_sc.p, c->cols_ctl.p, h_ctl.data(), h_sc.data(), &ran, cols_bicg))
                return rc;
            if (!ran) break;   // (more workgroups than fit after all, or a launch that gave up: these columns go one by one below)
            if (cols_bicg && c->bicg_restart) {   // a BiCGStab column that broke down: this batch and what follows one by one (solve_run_restarting)
                bool broke = false;
                for (int k = 0; k < Q; ++k) broke = broke || h_ctl[4 * (size_t)k + 2] != 0;
                if (broke) break;
            }
            hipLaunchKernelGGL(k_cols_finish, dim3(g1(n), Q), dim3(256), 0, st, n, c->cols_x.p, c->scale.p, c->dof_i2e.p, c->cols_b.p);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(x + (size_t)j0 * n, c->cols_b.p, sizeof(double) * qn, hipMemcpyDeviceToHost, st));
            for (int k = 0; k < Q; ++k) {
                const double bb = h_sc[4 * (size_t)k], rr = h_sc[4 * (size_t)k + 3];
                const double rel = bb > 0 ? sqrt(rr / bb) : 0.0;
                if (!(rr <= tol2 * bb && h_ctl[4 * (size_t)k + 2] == 0)) rc_all = FDAPDE_ENOCONV;
                breakdown = breakdown || h_ctl[4 * (size_t)k + 2] != 0;
                total += h_ctl[4 * (size_t)k + 1], worst = rel > worst ? rel : worst;
            }
            c->info.method_used = method, c->info.persistent = 1, c->info.launch_ms = c->persist_launch_ms;
            j0 += Q;
        }
    }
    for (int32_t j = j0; j < n_rhs; ++j) {
        HIPCHK(c, hipMemcpyAsync(c->tmp_e.p, b + (size_t)j * n, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_gather_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->tmp_e.p, rhs.p);
        clk.mark("lin_solve: upload + gather issued");
        c->defer_end_sync = true;
        const int rc = solve_run_restarting(c, c->lin_state->ss, c->lin_mat.p, rhs.p, c->g.p, nullptr, method, rtol, maxit, check_every, 0,
                                            !(!opt || opt->method == FDAPDE_SOLVER_AUTO) ? -1 : ((opt && opt->maxit > 0) ? 0 : default_maxit(c, n)));
        c->defer_end_sync = false;
        clk.mark("lin_solve: solve_run");
        if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
        if (rc == FDAPDE_ENOCONV) rc_all = rc, breakdown = breakdown || (c->h_ctl[2] != 0 && is_cg_method(c->info.method_used));
        total += c->info.iters, worst = c->info.relres > worst ? c->info.relres : worst;
        hipLaunchKernelGGL(k_scatter_f64, dim3(g1(n)), dim3(256), 0, st, n, c->dof_i2e.p, c->u.p, c->tmp_e.p);
        HIPCHK(c, hipMemcpyAsync(x + (size_t)j * n, c->tmp_e.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        clk.mark("lin_solve: solution download issued");
    }
    HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    if (rc_all == FDAPDE_ENOCONV && breakdown && may_retry) return retry_as_bicgstab();
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->info.t_solve_ms = ms, c->info.iters = total, c->info.relres = worst, c->info.converged = rc_all == FDAPDE_OK ? 1 : 0;
    if (rc_all == FDAPDE_ENOCONV)
        c->err = breakdown ? "Krylov breakdown in at least one column (matrix not positive definite for CG, or BiCGStab rho / omega = 0)" : "maxit reached in at least one column";
    if (info) *info = c->info;
    return rc_all;
}

int e_matrix_values(fdapde_ctx* c, int32_t which, double* values) {
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
int e_lump(fdapde_ctx* c, int32_t which, double* diag) {
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

int e_force(fdapde_ctx* c, double* force) {
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

int e_solution(fdapde_ctx* c, double* solution) {
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

int e_spmv(fdapde_ctx* c, int32_t which, const double* x, double* y) {
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

int e_bench_spmv(fdapde_ctx* c, int32_t reps, double* avg_ms, double* algorithmic_bytes) {
    if (!c || reps < 1) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready || !c->assembled[0]) return fail(c, FDAPDE_ENOTINIT, "matrix not assembled");
    HIPCHK(c, hipSetDevice(c->device));
    const HostSpace& hs = c->hs;
    // the launch timed here is the one inside CG: scaled matrix stream, fused p.Ap partials
    const double* A = (c->solved && c->scaled_owner == fdapde_ctx::kScaledSolve) ? c->sval.p : c->vals[0].p;
    if (A == c->sval.p)
        if (int rc = ensure_sval(c)) return rc;
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


// the unit's code object is loaded when one of its kernels is first looked up (HIP defers it): done at context creation, so that the
// first solve of a process does not pay for it (6 ms for the smoke problem after the library was split into units)
void preload_solve() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_jacobi_scale));
    (void)hipGetLastError();
}

}   // namespace fdapde_engine
