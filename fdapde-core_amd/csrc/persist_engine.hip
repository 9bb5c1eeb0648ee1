// persist_engine.hip -- host side of the single-launch solver (kernels_persist.h): layout selection and construction for a boundary
// variant, the launch itself (occupancy-checked, epoch-tagged), the fall-back policy after a hand-off timeout.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <utility>

#include <unistd.h>

#include "engine.h"
#include "kernels_persist.h"
#include "kernels_persist_bicg.h"

// the dot records start on a 64-byte boundary behind the p entries (hipMalloc: 256-byte aligned base), 8 granules = 64 bytes apart: a record never
// straddles two 128-byte lines; every board length is a multiple of 8 granules so that the boards of side-by-side columns keep the alignment
static inline size_t dboard_offset(int64_t n_board) { return (2 * (size_t)n_board + 7) & ~(size_t)7; }

namespace fdapde_engine {

// FDAPDE_SETUP_CHECK: the device-built persistent layout against the host builder's
int check_dev_persist(fdapde_ctx* c, int v, const PersistLayout& pl, const DevPersist& dp, const std::vector<int32_t>* block_rows, bool balance, int n_wg) {
    if (int rc = ensure_host(c, kHostPattern)) return rc;
    PersistLayout ref;
    ref.single_rows = pl.single_rows;
    if (host_build_persist_layout(c->hs, v == 1, n_wg, 12000, ref, block_rows ? block_rows->data() : nullptr,
                                  pl.sym ? 1 : 0, balance) != FDAPDE_OK) return fail(c, FDAPDE_EHIP, "set-up check: host persistent layout failed");
    int bad = 0;
    auto scalar = [&](const char* name, int64_t a, int64_t b) {
        if (a != b) std::fprintf(stderr, "persist check %-9s: MISMATCH %lld vs %lld\n", name, (long long)a, (long long)b), ++bad;
    };
    scalar("G", pl.G, ref.G), scalar("R", pl.R, ref.R), scalar("n_int", pl.n_int, ref.n_int), scalar("n_entries", pl.n_entries, ref.n_entries);
    scalar("nnz", pl.nnz, ref.nnz), scalar("n_board", pl.n_board, ref.n_board), scalar("max_imp", pl.max_imp, ref.max_imp), scalar("max_exp", pl.max_exp, ref.max_exp);
    auto cmp = [&](const char* name, const void* dev, const void* host, size_t bytes, size_t elem) {
        std::vector<unsigned char> tmp(bytes ? bytes : 1);
        if (bytes && hipMemcpy(tmp.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            ++bad;
            return;
        }
        size_t at = 0;
        while (at < bytes && tmp[at] == static_cast<const unsigned char*>(host)[at]) ++at;
        if (at < bytes) std::fprintf(stderr, "persist check %-9s: MISMATCH at element %zu of %zu\n", name, at / elem, bytes / elem), ++bad;
        else std::fprintf(stderr, "persist check %-9s: ok (%zu elements)\n", name, bytes / elem);
    };
    if (bad == 0) {
#define CMP(name, dptr, hvec_) cmp(name, dptr, (hvec_).data(), (hvec_).size() * sizeof((hvec_)[0]), sizeof((hvec_)[0]))
        CMP("slot_dof", dp.slot_dof, ref.slot_dof), CMP("ell_off", dp.ell_off, ref.ell_off), CMP("sl_off", dp.sl_off, ref.sl_off);
        CMP("ell_code", dp.ell_code, ref.ell_code), CMP("ell_src", dp.ell_src, ref.ell_src), CMP("exp_off", dp.exp_off, ref.exp_off);
        CMP("exp_slot", dp.exp_slot, ref.exp_slot), CMP("imp_off", dp.imp_off, ref.imp_off), CMP("imp_pos", dp.imp_pos, ref.imp_pos);
#undef CMP
    }
    if (bad) return fail(c, FDAPDE_EHIP, "FDAPDE_SETUP_CHECK: the device-built persistent layout differs from the host builder's (see stderr)");
    return FDAPDE_OK;
}

// resident layout of the persistent CG for boundary variant v (kernels_persist.h): host index work + uploads, once per function
// space and boundary mask.  ok stays false when the system does not qualify (too many rows for one launch of resident
// workgroups, or more matrix than is worth re-reading from the caches every iteration).
int build_persist_once(fdapde_ctx* c, int v, const std::vector<int32_t>* block_rows, bool balance) {
    fdapde_ctx::Persist& ps = c->ps[v];
    ps.ok = false;
    ps.ell_col.release();
    if (c->n_cu < 1) return FDAPDE_OK;
    const int32_t* brows = block_rows ? block_rows->data() : nullptr;
    const int n_wg = block_rows ? (int)block_rows->size() : (c->persist_max_wg > 0 ? std::min(c->persist_max_wg, c->n_cu) : c->n_cu);
    PersistLayout pl;
    DevPersist dp;
    DebugClock clk;
    const char* mode = std::getenv("FDAPDE_SETUP");
    bool on_device = !(mode && std::strcmp(mode, "host") == 0);
    // small systems: the host builder (microseconds of index work + nine small uploads) instead of ~40 device launches, sorts and
    // synchronisations that cost the same whatever the size -- the first solve of a 587-DOF system took 13.5 ms with them
    // (downstream models solve many small systems; the arrays are identical either way)
    if (c->hs.n_dofs <= c->persist_host_below && !(mode && std::strcmp(mode, "device") == 0)) on_device = false;
    double max_mb = 1024.0;   // ELL bytes (10 per entry) of the whole system (the row bound -- G x 8192 -- is reached first for P1 systems)
    if (const char* e = std::getenv("FDAPDE_PERSIST_MAX_MB")) max_mb = std::atof(e);
    const size_t lds_total = 160 * 1024 - 1024;   // static arrays of the kernel + slack
    size_t fixed = 0;
    int64_t need = 0;
    int imp_cap = 0, exp_cap = 0, S = 0;
    // symmetric storage (kernels_persist.h SYM) where the plain blocks would not fit the LDS; the plain form where they do (C2: the
    // iteration is latency-bound there, fewer bytes buy nothing) or where the accumulator table leaves no room for the vectors
    bool late_ok = c->persist_late != 0 && block_rows == nullptr;   // (knob; off by default: see DESIGN 4.0)
    if (late_ok) on_device = false;   // (only the host builder knows late workgroups)
    int sym_mode = c->persist_sym == 2 ? 3 : c->persist_sym;   // 0 never, 1 always, 2 auto: tried wherever the plain blocks would stream (3) ...
    if (c->persist_plain) sym_mode = 0;                        // (a non-symmetric system: BiCGStab on the plain storage)
    for (int attempt = 0; attempt < 2; ++attempt) {
        pl = PersistLayout{};
        pl.single_rows = c->persist_single_rows;
        int rc = FDAPDE_EUNSUPPORTED;
        if (on_device) {   // the layout is built where the pattern lives (dev_persist.hip)
            preload_wait(2);
            rc = dev_build_persist_layout(c->hs.n_dofs, c->hs.max_row, c->rowptr.p, c->colidx.p, c->bnd.p, v == 1, n_wg, 12000, 0, brows, sym_mode,
                                          balance, c->stream, pl, &dp, c->err);
            if (rc == FDAPDE_EUNSUPPORTED && c->hs.max_row > 255) on_device = false;   // rows too long for its sort keys: host builder
        }
        if (!on_device) {
            clk.mark("build_persist: (before ensure_host)");
            if (int rc2 = ensure_host(c, kHostPattern)) return rc2;
            clk.mark("build_persist: ensure_host");
            rc = host_build_persist_layout(c->hs, v == 1, n_wg, 12000, pl, brows, sym_mode, balance, nullptr, late_ok);   // ~12 000 ELL entries (120 KB) next to the vectors of <= 4096 rows
        }
        // a non-symmetric system (BiCGStab: six vectors in registers, at most 8 rows per thread) whose rows WOULD fit 8 per thread but whose
        // importing rows overflow the second half of a workgroup's slots (3-D: ~50 % of a block's rows read a neighbouring block) got 16: the
        // host builder once more, with such workgroups marked late (imports before their first pass) instead of doubled rows per thread
        // (3-D P2 advection-diffusion-reaction, 913 k DOFs: multi-launch 142 us per iteration, single launch ~105)
        if (c->persist_plain && !late_ok && block_rows == nullptr && (rc == FDAPDE_EUNSUPPORTED || (rc == FDAPDE_OK && pl.R > 8)) &&
            (rc == FDAPDE_EUNSUPPORTED ? (c->hs.n_dofs + n_wg - 1) / n_wg <= 8 * kPersistT : (pl.n_int + pl.G - 1) / pl.G <= 8 * kPersistT) && attempt == 0) {
            dev_persist_release(&dp);
            late_ok = true, on_device = false;
            --attempt;
            continue;
        }
        if (rc == FDAPDE_EUNSUPPORTED && sym_mode != 0 && attempt == 0) {   // (e.g. more rows per workgroup than the symmetric form takes: the plain one)
            dev_persist_release(&dp);
            sym_mode = 0;
            continue;
        }
        if (rc == FDAPDE_EUNSUPPORTED) return FDAPDE_OK;
        if (rc) return rc;
        if (10.0 * (double)pl.n_entries > max_mb * 1e6) {
            dev_persist_release(&dp);
            return FDAPDE_OK;
        }
        S = pl.R * kPersistT;
        imp_cap = (pl.max_imp + 63) & ~63, exp_cap = (pl.max_exp + 63) & ~63;
        fixed = pl.sym ? 8 * (size_t)(S + imp_cap) + 8 * (size_t)S + 64 : 8 * (size_t)(S + imp_cap) + 4 * (size_t)imp_cap + 2 * (size_t)exp_cap + 64;
        if (pl.R > kPersistRmax) fixed = 8 * (size_t)(S + imp_cap) + 64;   // wide form: the p table and the imports (lists stay in global memory)
        need = pl.max_block;   // largest workgroup block
        for (int g = 0; g < pl.G && !on_device; ++g) need = std::max<int64_t>(need, pl.ell_off[(size_t)g + 1] - pl.ell_off[(size_t)g]);
        need += 128;        // one pair row of zeros behind the block: slices narrower than their pass's widest re-read it (clamped loads)
        if (pl.sym && fixed > lds_total && attempt == 0) {   // no room for the accumulator table: the plain form
            dev_persist_release(&dp);
            sym_mode = 0;
            continue;
        }
        // ... and kept only where it pays (tools/persist_sym_ab.py): not if the PLAIN blocks of this partition would be resident (C2-size
        // systems: 6.2 against 8.3 us per iteration), and for workgroups of at most 2048 rows only if the symmetric blocks are resident
        // (3-D 314 k rows: 13.7 -> 11.7 us) -- streamed, the plain form is faster at that size (439 k rows: 14.7 against 15.7)
        if (c->persist_sym == 2 && pl.sym && attempt == 0 && pl.G == 1) {   // one workgroup streams from the L2: the plain form (2-D 2 304 rows: 4.5 -> 4.3 us)
            dev_persist_release(&dp);
            sym_mode = 0;
            continue;
        }
        if (c->persist_sym == 2 && pl.sym && attempt == 0) {
            const size_t fixed_plain = 8 * (size_t)(S + imp_cap) + 4 * (size_t)imp_cap + 2 * (size_t)exp_cap + 64;
            const size_t block_plain = (size_t)(1.03 * (double)pl.nnz_full / (double)pl.G) + 256;   // (boundaries at equal cost: blocks of equal size)
            const bool plain_resident = pl.R < 16 && fixed_plain + 10 * block_plain <= lds_total;
            const bool sym_resident = fixed + 10 * (size_t)need <= lds_total;
            if (plain_resident || ((pl.n_int + pl.G - 1) / pl.G <= 2048 && !sym_resident)) {
                dev_persist_release(&dp);
                sym_mode = 0;
                continue;
            }
        }
        break;
    }
    clk.mark("build_persist: layout");
    // resident form when every block fits its workgroup's LDS next to the vectors; else the blocks stream every iteration
    ps.stream = fixed + 10 * (size_t)need > lds_total;
    if (fixed > lds_total || (pl.R >= 16 && !ps.stream) || (pl.R > kPersistRmax && (c->persist_plain || !c->persist_wide))) {   // (no resident instantiation for
                                                         // 8192 rows and more: they never fit; the wide form is the CG's -- BiCGStab keeps six vectors in registers)
        dev_persist_release(&dp);
        return FDAPDE_OK;
    }
    ps.lds_cap = ps.stream ? 0 : (int32_t)need, ps.imp_cap = imp_cap;
    ps.lds_bytes = fixed + (ps.stream ? 0 : 10 * (size_t)need);
    // symmetric streaming form: the export list too, where the workgroup has room left (the kernel's static arrays take ~0.6 KB of the 160)
    ps.exp_lds = pl.sym && ps.stream && pl.R <= kPersistRmax && ps.lds_bytes + 2 * (size_t)exp_cap + 1024 <= 160 * 1024 && c->persist_exp_lds;
    if (ps.exp_lds) ps.lds_bytes += 2 * (size_t)exp_cap;
    hipStream_t st = c->stream;
    if (on_device) {
        if (std::getenv("FDAPDE_SETUP_CHECK")) {
            if (int rc2 = check_dev_persist(c, v, pl, dp, block_rows, balance, n_wg)) {
                dev_persist_release(&dp);
                return rc2;
            }
        }
        const size_t GS = (size_t)pl.G * S, n_alloc = (size_t)pl.n_entries + 256;
        adopt(ps.slot_dof, dp.slot_dof, GS), adopt(ps.ell_off, dp.ell_off, (size_t)pl.G + 1), adopt(ps.sl_off, dp.sl_off, (size_t)pl.G * (pl.nsl + 1));
        adopt(ps.ell_code, dp.ell_code, n_alloc), adopt(ps.ell_src, dp.ell_src, n_alloc), adopt(ps.exp_off, dp.exp_off, (size_t)pl.G + 1);
        adopt(ps.exp_slot, dp.exp_slot, (size_t)(pl.n_board ? pl.n_board : 1)), adopt(ps.imp_off, dp.imp_off, (size_t)pl.G + 1);
        adopt(ps.imp_pos, dp.imp_pos, (size_t)(pl.n_imp ? pl.n_imp : 1));
    } else {
        HIPCHK(c, ps.slot_dof.upload(pl.slot_dof.data(), pl.slot_dof.size(), st));
        HIPCHK(c, ps.ell_off.upload(pl.ell_off.data(), pl.ell_off.size(), st));
        HIPCHK(c, ps.sl_off.upload(pl.sl_off.data(), pl.sl_off.size(), st));
        HIPCHK(c, ps.ell_code.alloc(pl.ell_code.size() + 256));   // + slack: clamped loads of the last slices may run past the last block
        HIPCHK(c, hipMemsetAsync(ps.ell_code.p, 0, sizeof(uint16_t) * (pl.ell_code.size() + 256), st));
        HIPCHK(c, hipMemcpyAsync(ps.ell_code.p, pl.ell_code.data(), sizeof(uint16_t) * pl.ell_code.size(), hipMemcpyHostToDevice, st));
        HIPCHK(c, ps.ell_src.upload(pl.ell_src.data(), pl.ell_src.size(), st));
        HIPCHK(c, ps.exp_off.upload(pl.exp_off.data(), pl.exp_off.size(), st));
        HIPCHK(c, ps.exp_slot.upload(pl.exp_slot.data(), pl.exp_slot.size(), st));
        HIPCHK(c, ps.imp_off.upload(pl.imp_off.data(), pl.imp_off.size(), st));
        HIPCHK(c, ps.imp_pos.upload(pl.imp_pos.data(), pl.imp_pos.size(), st));
    }
    if (late_ok && !pl.wg_late.empty()) HIPCHK(c, ps.wg_late.upload(pl.wg_late.data(), pl.wg_late.size(), st));
    else ps.wg_late.release();
    HIPCHK(c, ps.ell_val.alloc((size_t)pl.n_entries + 256));
    HIPCHK(c, hipMemsetAsync(ps.ell_val.p, 0, sizeof(double) * ((size_t)pl.n_entries + 256), st));
    HIPCHK(c, ps.board.alloc(dboard_offset(pl.n_board) + 2 * (size_t)pl.G * 8 + 8));   // p entries | dot records x 2 buffers (CG: 3 doubles wide, BiCGStab: 4)
    HIPCHK(c, hipMemsetAsync(ps.board.p, 0, sizeof(unsigned long long) * ps.board.n, st));   // every tag 0: no launch uses epoch 0
    ps.epoch_next = 0, ps.attr_set = nullptr;
    ps.board_cols.release();   // (boards of multi-column launches: tags of the layout before must not meet the new epochs)
    HIPCHK(c, ps.amax.alloc(1));
    HIPCHK(c, c->persist_stats.alloc(4 * 1024));
    HIPCHK(c, hipStreamSynchronize(st));
    clk.mark("build_persist: uploads + allocs");
    if (std::getenv("FDAPDE_DEBUG_SETUP"))
        std::fprintf(stderr, "persistent CG layout %d (%s-built): %d workgroups x %d rows/thread, %lld interior rows, %lld entries (%lld stored, %.1f %% padding), "
                     "LDS %zu B (%s%s, largest block %lld), imports <= %d, exports <= %d, board %lld\n", v, on_device ? "device" : "host", pl.G, pl.R,
                     (long long)pl.n_int, (long long)pl.n_entries, (long long)pl.nnz,
                     100.0 * (double)(pl.n_entries - pl.nnz) / (double)(pl.n_entries > 0 ? pl.n_entries : 1), ps.lds_bytes,
                     ps.stream ? "blocks stream" : "blocks resident", pl.sym ? ", symmetric storage" : "", (long long)need, pl.max_imp, pl.max_exp, (long long)pl.n_board);
    // keep the sizes, drop the big host arrays
    pl.slot_dof = {}, pl.ell_code = {}, pl.ell_src = {}, pl.exp_slot = {}, pl.imp_pos = {}, pl.sl_off = {}, pl.ell_off = {};
    ps.meta = std::move(pl);
    ps.filled = false, ps.ok = true;
    return FDAPDE_OK;
}

// launch of k_cg_persist on the layout ps.  The boards are NOT cleared: every launch tags its granules with epochs of its own
// (ps.epoch_next + iteration + 1, strictly increasing from launch to launch), so what an earlier launch left behind never matches.
int launch_persist(fdapde_ctx* c, fdapde_ctx::Persist& ps, PersistArgs& a, bool dist = false, bool bicg = false) {
    hipStream_t st = c->stream;
    a.G = ps.meta.G, a.nsl = ps.meta.nsl, a.imp_cap = ps.imp_cap, a.lds_cap = ps.lds_cap;
    a.gather_waves = c->persist_gather_waves, a.poll_sleep = c->persist_poll_sleep;
    a.slot_dof = ps.slot_dof.p, a.ell_off = ps.ell_off.p, a.sl_off = ps.sl_off.p, a.ell_code = ps.ell_code.p;
    a.ell_val = ps.ell_val.p, a.exp_off = ps.exp_off.p, a.exp_slot = ps.exp_slot.p, a.imp_off = ps.imp_off.p, a.imp_pos = ps.imp_pos.p;
    if (!dist && a.n_cols <= 1) a.pboard = ps.board.p, a.dboard = ps.board.p + dboard_offset(ps.meta.n_board);   // (row-distributed, several columns: set by the caller)
    if (!dist) a.wg_late = ps.wg_late.p;
    a.exp_lds = (!dist && !bicg && ps.exp_lds) ? 1 : 0;
    a.amax_bits = ps.amax.p, a.max_len = c->hs.max_row, a.stats = c->persist_stats.p;
    a.timeout_ticks = c->persist_timeout_us * 100, a.debug_stall_it = c->persist_debug_stall, a.pf_steps = c->persist_prefetch;
    if (!dist && ps.epoch_next > 0xC0000000u - 2u * (uint32_t)a.maxit) {   // (the tags are 32 bits wide: start over on clean boards)
        HIPCHK(c, hipMemsetAsync(ps.board.p, 0, sizeof(unsigned long long) * ps.board.n, st));
        if (ps.board_cols.p) HIPCHK(c, hipMemsetAsync(ps.board_cols.p, 0, sizeof(unsigned long long) * ps.board_cols.n, st));
        ps.epoch_next = 0;
    }
    a.epoch0 = ps.epoch_next;
    if (a.time_phases) HIPCHK(c, hipMemsetAsync(c->persist_stats.p, 0, 4 * (size_t)a.G * sizeof(double), st));
    HIPCHK(c, hipEventRecord(c->ev_p0, st));
    void* kargs[] = {&a};
    // co-residency of the G workgroups is what the in-kernel hand-offs rely on: G <= (workgroups of this instantiation the runtime says a
    // CU holds) x CUs, checked below through the occupancy API; knob persist_coop makes the launch cooperative on top (the runtime then
    // refuses a grid that cannot be resident instead of letting it spin -- at 10.5 ms for the first such launch of a process)
#define PERSIST_GO(R_, ST_, SY_) PERSIST_GO_D(R_, ST_, SY_, false)
#define PERSIST_GO_D(R_, ST_, SY_, DI_)                                                                                         \
    do {                                                                                                                        \
        const void* fn = reinterpret_cast<const void*>(&k_cg_persist<R_, ST_, SY_, DI_>);                                       \
        if (ps.attr_set != fn) {                                                                                                \
            DebugClock clk2;                                                                                                    \
            HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ps.lds_bytes));                  \
            clk2.mark("launch_persist: hipFuncSetAttribute");                                                                   \
            int per_cu = 0;                                                                                                     \
            HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kPersistT, ps.lds_bytes));                      \
            clk2.mark("launch_persist: occupancy query");                                                                       \
            if ((int64_t)per_cu * c->n_cu < (int64_t)a.G) return FDAPDE_EUNSUPPORTED;   /* the grid cannot be resident at once */ \
            ps.attr_set = fn, ps.per_cu = per_cu;                                                                               \
        }                                                                                                                       \
        const unsigned ncol = a.n_cols > 1 ? (unsigned)a.n_cols : 1u;                                                           \
        if ((int64_t)ps.per_cu * c->n_cu < (int64_t)a.G * ncol) return FDAPDE_EUNSUPPORTED;                                     \
        if (c->persist_coop) HIPCHK(c, hipLaunchCooperativeKernel(fn, dim3(a.G, ncol), dim3(kPersistT), kargs, (unsigned)ps.lds_bytes, st)); \
        else hipLaunchKernelGGL((k_cg_persist<R_, ST_, SY_, DI_>), dim3(a.G, ncol), dim3(kPersistT), ps.lds_bytes, st, a);      \
    } while (0)
#define PERSIST_GO2(R_, ST_)                                                                                                    \
    do {                                                                                                                        \
        if (ps.meta.sym) PERSIST_GO(R_, ST_, true);                                                                             \
        else PERSIST_GO(R_, ST_, false);                                                                                        \
    } while (0)
#define PERSIST_GOD(R_, ST_)                                                                                                    \
    do {                                                                                                                        \
        if (ps.meta.sym) PERSIST_GO_D(R_, ST_, true, true);                                                                     \
        else PERSIST_GO_D(R_, ST_, false, true);                                                                                \
    } while (0)
#define BICG_GO(R_, ST_, DI_)                                                                                                   \
    do {                                                                                                                        \
        const void* fn = reinterpret_cast<const void*>(&k_bicg_persist<R_, ST_, DI_>);                                          \
        if (ps.attr_set != fn) {                                                                                                \
            HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ps.lds_bytes));                  \
            int per_cu = 0;                                                                                                     \
            HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kPersistT, ps.lds_bytes));                      \
            if ((int64_t)per_cu * c->n_cu < (int64_t)a.G) return FDAPDE_EUNSUPPORTED;                                           \
            ps.attr_set = fn, ps.per_cu = per_cu;                                                                               \
        }                                                                                                                       \
        const unsigned ncol = a.n_cols > 1 ? (unsigned)a.n_cols : 1u;                                                           \
        if ((int64_t)ps.per_cu * c->n_cu < (int64_t)a.G * ncol) return FDAPDE_EUNSUPPORTED;                                     \
        hipLaunchKernelGGL((k_bicg_persist<R_, ST_, DI_>), dim3(a.G, ncol), dim3(kPersistT), ps.lds_bytes, st, a);              \
    } while (0)
#define BICG_GO2(R_, ST_)                                                                                                       \
    do {                                                                                                                        \
        if (dist) BICG_GO(R_, ST_, true);                                                                                       \
        else BICG_GO(R_, ST_, false);                                                                                           \
    } while (0)
    if (bicg) {   // BiCGStab: plain storage, six vectors in registers: at most 8 rows per thread
        if (ps.meta.sym) return FDAPDE_EUNSUPPORTED;
        if (ps.stream) switch (ps.meta.R) {
            case 2: BICG_GO2(2, true); break;
            case 4: BICG_GO2(4, true); break;
            case 8: BICG_GO2(8, true); break;
            default: return FDAPDE_EUNSUPPORTED;
            }
        else switch (ps.meta.R) {
            case 2: BICG_GO2(2, false); break;
            case 4: BICG_GO2(4, false); break;
            case 8: BICG_GO2(8, false); break;
            default: return FDAPDE_EUNSUPPORTED;
            }
    } else if (dist) {   // row-distributed form
        if (ps.stream) switch (ps.meta.R) {
            case 2: PERSIST_GOD(2, true); break;
            case 4: PERSIST_GOD(4, true); break;
            case 8: PERSIST_GOD(8, true); break;
            case 16: PERSIST_GOD(16, true); break;
            default: return FDAPDE_EUNSUPPORTED;
            }
        else switch (ps.meta.R) {
            case 2: PERSIST_GOD(2, false); break;
            case 4: PERSIST_GOD(4, false); break;
            case 8: PERSIST_GOD(8, false); break;
            default: return FDAPDE_EUNSUPPORTED;
            }
    } else if (ps.stream) switch (ps.meta.R) {
        case 2: PERSIST_GO2(2, true); break;
        case 4: PERSIST_GO2(4, true); break;
        case 8: PERSIST_GO2(8, true); break;
        case 16: PERSIST_GO2(16, true); break;
        case kPersistRwide:   // wide form: plain storage, x in HBM in slot order, one column
            if (ps.meta.sym || a.n_cols > 1 || a.direct) return FDAPDE_EUNSUPPORTED;
            if (c->persist_xs.n < (size_t)ps.meta.G * kPersistRwide * kPersistT) HIPCHK(c, c->persist_xs.alloc((size_t)ps.meta.G * kPersistRwide * kPersistT));
            a.x_slots = c->persist_xs.p;
            {   // (measurement knob persist_wide_gj: passes of a phase whose loads go out together)
                const void* fn = c->persist_wide_gj == 12 ? reinterpret_cast<const void*>(&k_cg_persist<kPersistRwide, true, false, false, 12>)
                               : c->persist_wide_gj == 4  ? reinterpret_cast<const void*>(&k_cg_persist<kPersistRwide, true, false, false, 4>)
                                                          : reinterpret_cast<const void*>(&k_cg_persist<kPersistRwide, true, false, false, 6>);
                if (ps.attr_set != fn) {
                    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ps.lds_bytes));
                    int per_cu = 0;
                    HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kPersistT, ps.lds_bytes));
                    if ((int64_t)per_cu * c->n_cu < (int64_t)a.G) return FDAPDE_EUNSUPPORTED;
                    ps.attr_set = fn, ps.per_cu = per_cu;
                }
                if (c->persist_wide_gj == 12) hipLaunchKernelGGL((k_cg_persist<kPersistRwide, true, false, false, 12>), dim3(a.G), dim3(kPersistT), ps.lds_bytes, st, a);
                else if (c->persist_wide_gj == 4) hipLaunchKernelGGL((k_cg_persist<kPersistRwide, true, false, false, 4>), dim3(a.G), dim3(kPersistT), ps.lds_bytes, st, a);
                else hipLaunchKernelGGL((k_cg_persist<kPersistRwide, true, false, false, 6>), dim3(a.G), dim3(kPersistT), ps.lds_bytes, st, a);
            }
            break;
        default: return FDAPDE_EUNSUPPORTED;
        }
    else switch (ps.meta.R) {
        case 2: PERSIST_GO2(2, false); break;
        case 4: PERSIST_GO2(4, false); break;
        default: PERSIST_GO2(8, false); break;
        }
#undef BICG_GO2
#undef BICG_GO
#undef PERSIST_GOD
#undef PERSIST_GO2
#undef PERSIST_GO_D
#undef PERSIST_GO
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_p1, st));
    return FDAPDE_OK;
}

int build_persist(fdapde_ctx* c, int v) {
    fdapde_ctx::Persist& ps = c->ps[v];
    if (ps.tried && ps.built_plain != (c->persist_plain != 0) && (ps.built_plain || (ps.ok && ps.meta.sym) || !ps.ok)) ps.tried = false;   // symmetric <-> plain storage
    if (ps.tried) return FDAPDE_OK;
    ps.built_plain = c->persist_plain != 0;
    ps.tried = true;
    if (int rc = build_persist_once(c, v, nullptr, c->persist_balance != 0)) return rc;
    // boundaries at equal cost can leave one workgroup with more rows that import than its import-free passes have room for
    // where equal row counts would not: the system must not lose the single-launch path over that
    if (!ps.ok && c->persist_balance) return build_persist_once(c, v, nullptr, false);
    return FDAPDE_OK;
}

// the whole fused-update CG as one launch; returns FDAPDE_OK with *ran = false when the launch gave up (hand-off timeout): a launch
// that gives up leaves x (it writes the solution to persist_x), r, p, sc and ctl[0..2] as it found them, so the multi-launch
// path restarts the same solve from the same state
int run_persist(fdapde_ctx* c, int v, double tol2, int maxit, bool* ran, bool bicg) {
    fdapde_ctx::Persist& ps = c->ps[v];
    hipStream_t st = c->stream;
    const size_t n = (size_t)c->hs.n_dofs;
    if (c->persist_x.n < n) {   // rows the layout leaves out (Dirichlet DOFs) are never written: they must read as finite numbers
        HIPCHK(c, c->persist_x.alloc(n));
        HIPCHK(c, hipMemsetAsync(c->persist_x.p, 0, sizeof(double) * n, st));
    }
    PersistArgs a{};
    a.maxit = maxit, a.time_phases = c->persist_time, a.tol2 = tol2;
    a.r_in = c->r.p, a.x = c->x.p, a.x_out = c->persist_x.p, a.sc = c->sc.p, a.ctl = c->ctl.p;
    // small systems in one workgroup (what the reference's users mostly solve): no phase stamps (a memset and a copy), and the outcome comes back
    // through a record in pinned host memory the launch writes itself instead of through three device-to-host copies (13 us of blit kernels
    // behind a 130 us launch)
    const bool host_rec = ps.meta.G == 1 && c->small_rows > 0 && (int64_t)n <= c->small_rows;
    if (host_rec) a.time_phases = 0, a.hrec = c->h_sc + 8;
    // ... and behind k_small_front (the Dirichlet entries of u are in place) the epilogue is the launch's own write-out
    c->tail_in_launch = host_rec && c->front_used && c->persist_tail != nullptr;
    if (c->tail_in_launch) a.u_out = c->u.p, a.scale = c->scale.p;
    DebugClock clk;
    const int rc_launch = launch_persist(c, ps, a, false, bicg);
    clk.mark("run_persist: launch call");
    if (int rc = rc_launch) {
        if (rc != FDAPDE_EUNSUPPORTED) return rc;
        ps.ok = false, *ran = false;   // the occupancy the runtime reports does not hold the grid: this layout never launches
        return FDAPDE_OK;
    }
    if (!host_rec) {
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st));   // ([4]: the deferred positive-diagonal flag)
        HIPCHK(c, hipMemcpyAsync(c->h_sc, c->sc.p, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    c->h_ctl_seen = 8;
    if (a.time_phases) {
        c->persist_host_stats.resize(4 * (size_t)a.G);
        HIPCHK(c, hipMemcpyAsync(c->persist_host_stats.data(), c->persist_stats.p, 4 * (size_t)a.G * sizeof(double), hipMemcpyDeviceToHost, st));
    } else
        c->persist_host_stats.clear();
    if (c->persist_tail)
        if (int rc = c->persist_tail()) return rc;
    HIPCHK(c, hipStreamSynchronize(st));
    if (host_rec) {   // (written by the launch; visible after the wait)
        const volatile double* hr = c->h_sc + 8;
        for (int k = 0; k < 5; ++k) c->h_ctl[k] = (int32_t)hr[k];
        c->h_sc[0] = hr[5], c->h_sc[3] = hr[6];
    }
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_p0, c->ev_p1));
    c->persist_launch_ms = ms;
    *ran = c->h_ctl[3] == 0;
    if (!*ran) {   // a peer workgroup was not resident (other work on the device?).  The context stays on the multi-launch path for a
                   // while and tries again later, twice as much later after every failure (8, 16, ... 1024 solves)
        c->persist_broken = true;
        c->persist_retry_in = c->persist_backoff;
        c->persist_backoff = std::min(1024, 2 * c->persist_backoff);
        HIPCHK(c, hipMemsetAsync(c->ctl.p + 3, 0, sizeof(int32_t), st));
        HIPCHK(c, hipMemsetAsync(ps.board.p, 0, sizeof(unsigned long long) * ps.board.n, st));   // (how far its epochs got is unknown)
        if (ps.board_cols.p) HIPCHK(c, hipMemsetAsync(ps.board_cols.p, 0, sizeof(unsigned long long) * ps.board_cols.n, st));   // (the tags start over for them too)
        ps.epoch_next = 0;
    } else {
        c->persist_backoff = 8;
        ps.epoch_next += (bicg ? 2u : 1u) * ((uint32_t)c->h_ctl[1] + 2u);   // (BiCGStab: two tagged hand-offs per iteration)
    }
    return FDAPDE_OK;
}


// ONE right-hand side against a system the single-launch CG holds in ONE workgroup, for callers that cannot batch columns: the launch reads b
// (reference DOF order) from pinned host memory, scales it, solves, and writes the unscaled solution (reference order) and its outcome record
// back into pinned host memory itself (kernels_persist.h PersistArgs::direct) -- no prologue / epilogue kernels, no device-to-host copies, ONE
// wait.  The host spins on the record's status word (written last, system scope) for a while before it falls back to waiting for the stream.
// *ran = false: not applicable / the launch gave up (the caller takes the general path).
int run_persist_direct(fdapde_ctx* c, int v, double tol2, int maxit, const double* b_host, double* x_host, bool* ran) {
    fdapde_ctx::Persist& ps = c->ps[v];
    *ran = false;
    if (!ps.ok || !ps.filled || ps.meta.G != 1 || ps.meta.n_drop != 0) return FDAPDE_OK;
    hipStream_t st = c->stream;
    const size_t n = (size_t)c->hs.n_dofs;
    if (c->h_io_cap < 2 * n + 8) {
        if (c->h_io) (void)hipHostFree(c->h_io);
        c->h_io = nullptr, c->h_io_cap = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_io), sizeof(double) * (2 * n + 8)));   // (pinned, mapped, coherent: the device reads and writes it over PCIe)
        c->h_io_cap = 2 * n + 8;
    }
    double* hb = c->h_io;
    double* hx = c->h_io + n;
    volatile double* rec = c->h_io + 2 * n;
    std::memcpy(hb, b_host, sizeof(double) * n);
    rec[0] = rec[1] = rec[2] = 0.0;
    *reinterpret_cast<volatile long long*>(rec + 3) = 0;
    PersistArgs a{};
    a.maxit = maxit, a.time_phases = 0, a.tol2 = tol2;
    a.direct = 1, a.b_ext = hb, a.x_ext = hx, a.rec = c->h_io + 2 * n, a.i2e = c->dof_i2e.p, a.scale = c->scale.p;
    a.sc = c->sc.p, a.ctl = c->ctl.p;   // (not touched by a direct launch)
    const int rc_launch = launch_persist(c, ps, a, false, false);
    if (rc_launch == FDAPDE_EUNSUPPORTED) return FDAPDE_OK;
    if (rc_launch) return rc_launch;
    // the outcome: the status word first (spin, bounded), the stream's end as the fall-back and as the fence for the event pair
    long long status1 = 0;
    if (c->persist_direct_spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        while ((status1 = *reinterpret_cast<volatile long long*>(rec + 3)) == 0) {
            if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > (double)c->persist_direct_spin_us) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (status1 == 0) {
        HIPCHK(c, hipStreamSynchronize(st));
        status1 = *reinterpret_cast<volatile long long*>(rec + 3);
    }
    c->persist_launch_ms = 0;   // (the event pair is only read when somebody waits for the stream anyway: fdapde_info_get of a direct solve reports 0)
    if (status1 == 0 || status1 == 4) {   // the launch gave up (tests: persist_debug_stall) or never ran
        HIPCHK(c, hipStreamSynchronize(st));
        return FDAPDE_OK;
    }
    std::memcpy(x_host, hx, sizeof(double) * n);
    c->h_ctl[0] = status1 == 2 ? 1 : 0, c->h_ctl[1] = (int32_t)rec[0], c->h_ctl[2] = status1 == 3 ? 1 : 0, c->h_ctl[3] = 0;
    c->h_sc[0] = rec[2], c->h_sc[3] = rec[1];
    ps.epoch_next += (uint32_t)c->h_ctl[1] + 2u;
    *ran = true;
    return FDAPDE_OK;
}

// ... straight from the unscaled matrix A and the Jacobi scale (k_persist_fill_scaled): the scaled full-pattern copy is then not needed
int fill_persist_scaled(fdapde_ctx* c, int v, const double* A) {
    fdapde_ctx::Persist& ps = c->ps[v];
    hipStream_t st = c->stream;
    if (ps.meta.sym) HIPCHK(c, hipMemsetAsync(ps.amax.p, 0, sizeof(unsigned long long), st));
    const size_t n_alloc = (size_t)ps.meta.n_entries + 256;
    if (ps.ell_col.n < n_alloc) {   // once per layout
        HIPCHK(c, ps.ell_col.alloc(n_alloc));
        HIPCHK(c, hipMemsetAsync(ps.ell_col.p, 0, sizeof(int32_t) * n_alloc, st));
        if (ps.meta.n_entries > 0)   // (a block without off-diagonal entries -- ONE interior row -- has nothing to fill: a grid of 0 workgroups is a launch error)
            hipLaunchKernelGGL(k_persist_ell_col, dim3(grid1(ps.meta.n_entries)), dim3(256), 0, st, ps.meta.n_entries, ps.ell_src.p, c->colidx.p, ps.ell_col.p);
    }
    const int per = (ps.meta.nsl + 3) / 4;
    hipLaunchKernelGGL(k_persist_fill_scaled, dim3((unsigned)(ps.meta.G * per)), dim3(256), 0, st, ps.meta.G, ps.meta.nsl, ps.ell_off.p, ps.sl_off.p, ps.slot_dof.p,
                       ps.ell_src.p, ps.ell_col.p, A, c->scale.p, ps.ell_val.p, ps.meta.sym ? ps.amax.p : (unsigned long long*)nullptr);
    HIPCHK(c, hipGetLastError());
    ps.filled = true;
    return FDAPDE_OK;
}

// n_cols right-hand sides against the matrix of layout v as ONE launch of G x n_cols workgroups (kernels_persist.h n_cols): column k reads its
// scaled right-hand side r_cols + k n and ||.||^2 from sc_cols[4 k], starts from 0, writes x_cols + k n, sc_cols[4 k + 3], ctl_cols[4 k ..].
// The columns do not interact: each has boards of its own.  *ran = false when any column's launch gave up (nothing is lost: the caller
// solves the columns one by one).  h_ctl / h_sc: 4 n_cols values each, read back here.
int run_persist_cols(fdapde_ctx* c, int v, double tol2, int maxit, int n_cols, const double* r_cols, double* x_cols, double* sc_cols, int32_t* ctl_cols,
                     int32_t* h_ctl, double* h_sc, bool* ran, bool bicg) {
    fdapde_ctx::Persist& ps = c->ps[v];
    hipStream_t st = c->stream;
    const size_t n = (size_t)c->hs.n_dofs, blen = ps.board.n;
    if (ps.board_cols.n < blen * (size_t)n_cols) {
        HIPCHK(c, ps.board_cols.alloc(blen * (size_t)n_cols));
        HIPCHK(c, hipMemsetAsync(ps.board_cols.p, 0, sizeof(unsigned long long) * ps.board_cols.n, st));   // every tag 0: no launch uses epoch 0
    }
    PersistArgs a{};
    a.maxit = maxit, a.time_phases = 0, a.tol2 = tol2;
    a.r_in = r_cols, a.x = nullptr, a.x_out = x_cols, a.sc = sc_cols, a.ctl = ctl_cols;
    a.n_cols = n_cols, a.col_stride = (int64_t)n, a.board_stride = (int64_t)blen;
    a.pboard = ps.board_cols.p, a.dboard = ps.board_cols.p + dboard_offset(ps.meta.n_board);
    const int rc_launch = launch_persist(c, ps, a, false, bicg);
    if (rc_launch == FDAPDE_EUNSUPPORTED) {
        *ran = false;
        return FDAPDE_OK;
    }
    if (rc_launch) return rc_launch;
    HIPCHK(c, hipMemcpyAsync(h_ctl, ctl_cols, 4 * (size_t)n_cols * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h_sc, sc_cols, 4 * (size_t)n_cols * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_p0, c->ev_p1));
    c->persist_launch_ms = ms;
    bool gave_up = false;
    uint32_t most = 0;
    for (int k = 0; k < n_cols; ++k) gave_up = gave_up || h_ctl[4 * k + 3] != 0, most = std::max<uint32_t>(most, (uint32_t)h_ctl[4 * k + 1]);
    ps.epoch_next += (bicg ? 2u : 1u) * ((gave_up ? (uint32_t)maxit : most) + 2u);   // past every tag any column can have written
    *ran = !gave_up;
    return FDAPDE_OK;
}

int fill_persist(fdapde_ctx* c, int v) {
    fdapde_ctx::Persist& ps = c->ps[v];
    hipStream_t st = c->stream;
    if (ps.meta.sym) HIPCHK(c, hipMemsetAsync(ps.amax.p, 0, sizeof(unsigned long long), st));
    if (ps.meta.n_entries > 0)
        hipLaunchKernelGGL(k_persist_fill, dim3((unsigned)((ps.meta.n_entries + 1023) / 1024)), dim3(256), 0, st, ps.meta.n_entries, ps.ell_src.p, c->sval.p, ps.ell_val.p,
                       ps.meta.sym ? ps.amax.p : (unsigned long long*)nullptr);
    HIPCHK(c, hipGetLastError());
    ps.filled = true;
    return FDAPDE_OK;
}

// =====================================================================================================================================
// Row-distributed multi-GPU form (fdapde_rowdist_setup): one launch per rank, all of them acting as ONE grid (kernels_persist.h DIST).
// Set-up talks through the context's transports: the RCCL communicator (device buffers) or the host-staged callbacks of the tests.
// =====================================================================================================================================
namespace {

#define RCCLCHK_E(ctx, expr)                                                                 \
    do {                                                                                     \
        ncclResult_t r__ = (expr);                                                           \
        if (r__ != ncclSuccess) {                                                            \
            (ctx)->err = std::string(#expr) + ": " + g_rccl.GetErrorString(r__);             \
            return FDAPDE_ERCCL;                                                             \
        }                                                                                    \
    } while (0)

// sum of a small host vector over the ranks, in place
int host_allreduce(fdapde_ctx* c, std::vector<double>& v) {
    if (c->world <= 1) return FDAPDE_OK;
    if (c->ar_fn) return c->ar_fn(c->ar_user, v.data(), (int64_t)v.size()) == 0 ? FDAPDE_OK : fail(c, FDAPDE_ERCCL, "all-reduce callback failed");
    HIPCHK(c, c->ar_dev.alloc(v.size()));
    HIPCHK(c, hipMemcpyAsync(c->ar_dev.p, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice, c->stream));
    RCCLCHK_E(c, g_rccl.AllReduce(c->ar_dev.p, c->ar_dev.p, v.size(), ncclFloat64, ncclSum, c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(v.data(), c->ar_dev.p, sizeof(double) * v.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDAPDE_OK;
}

// doubles between pairs of ranks, segments of different size in the two directions: send[soff[q] .. soff[q + 1]) goes to ranks[q], what
// that rank sends arrives in recv[roff[q] .. roff[q + 1]).  Device buffers.  RCCL: one grouped call.  Host-staged transport: its callback
// moves segments of ONE size per pair, so both directions are padded to the larger one.
int pair_exchange(fdapde_ctx* c, const std::vector<int32_t>& ranks, const std::vector<int64_t>& soff, const std::vector<int64_t>& roff,
                  const double* send_dev, double* recv_dev) {
    const int np = (int)ranks.size();
    if (np == 0) return FDAPDE_OK;
    hipStream_t st = c->stream;
    if (!c->ar_fn) {
        RCCLCHK_E(c, g_rccl.GroupStart());
        for (int q = 0; q < np; ++q) {
            const size_t ns = (size_t)(soff[(size_t)q + 1] - soff[(size_t)q]), nr = (size_t)(roff[(size_t)q + 1] - roff[(size_t)q]);
            if (ns) RCCLCHK_E(c, g_rccl.Send(send_dev + soff[(size_t)q], ns, ncclFloat64, ranks[(size_t)q], c->comm, st));
            if (nr) RCCLCHK_E(c, g_rccl.Recv(recv_dev + roff[(size_t)q], nr, ncclFloat64, ranks[(size_t)q], c->comm, st));
        }
        RCCLCHK_E(c, g_rccl.GroupEnd());
        return FDAPDE_OK;
    }
    if (!c->xchg_fn) return fail(c, FDAPDE_ENOTINIT, "fdapde_comm_set_exchange_callback not called");
    std::vector<int64_t> poff((size_t)np + 1, 0);
    for (int q = 0; q < np; ++q)
        poff[(size_t)q + 1] = poff[(size_t)q] + std::max(soff[(size_t)q + 1] - soff[(size_t)q], roff[(size_t)q + 1] - roff[(size_t)q]);
    std::vector<double> hs_((size_t)soff[(size_t)np] + 1), hr_((size_t)roff[(size_t)np] + 1), ps_((size_t)poff[(size_t)np] + 1, 0.0), pr_((size_t)poff[(size_t)np] + 1, 0.0);
    HIPCHK(c, hipMemcpyAsync(hs_.data(), send_dev, sizeof(double) * (size_t)soff[(size_t)np], hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    for (int q = 0; q < np; ++q)
        std::copy(hs_.begin() + soff[(size_t)q], hs_.begin() + soff[(size_t)q + 1], ps_.begin() + poff[(size_t)q]);
    if (c->xchg_fn(c->xchg_user, np, ranks.data(), poff.data(), ps_.data(), pr_.data()) != 0) return fail(c, FDAPDE_ERCCL, "exchange callback failed");
    for (int q = 0; q < np; ++q)
        std::copy(pr_.begin() + poff[(size_t)q], pr_.begin() + poff[(size_t)q] + (roff[(size_t)q + 1] - roff[(size_t)q]), hr_.begin() + roff[(size_t)q]);
    HIPCHK(c, hipMemcpyAsync(recv_dev, hr_.data(), sizeof(double) * (size_t)roff[(size_t)np], hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDAPDE_OK;
}

__global__ __launch_bounds__(256) void k_rd_gather(int64_t n, const int32_t* dof, const double* v, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v[dof[i]];
}
__global__ __launch_bounds__(256) void k_rd_scatter(int64_t n, const int32_t* dof, const double* in, double* v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[dof[i]] = in[i];
}

struct BoardBlob {   // what a rank tells the others about its board
    int64_t pid;
    uint64_t ptr;
    int32_t device, pad;
    hipIpcMemHandle_t handle;
};

}   // namespace

// layout of boundary variant v + everything the ranks agree on for it (positions in each other's boards, board mappings).  COLLECTIVE:
// every rank calls it for the same v at the same point.  lay.ok = false on ALL ranks when any rank's share does not qualify.
int build_rowdist(fdapde_ctx* c, int v) {
    fdapde_ctx::RowDist& rd = c->rd;
    fdapde_ctx::RowDist::Layout& L = rd.lay[v];
    if (L.tried) return FDAPDE_OK;
    L.tried = true, L.ok = false;
    for (void* m : L.ipc_opened) (void)hipIpcCloseMemHandle(m);   // (a rebuild after the boundary flags changed: the old mappings go)
    L.ipc_opened.clear();
    const int W = c->world, me = c->rank;
    const HostSpace& hs = c->hs;
    const int64_t nd = hs.n_dofs;
    hipStream_t st = c->stream;
    // A local failure that is an ERROR (not "this rank's share does not fit") must not make this rank leave while the others wait in the next
    // exchange: it is remembered here, travels as "not ok" / as an error flag to the next agreement point, where ALL ranks leave together
    // (lay.ok stays false everywhere), and only then is it returned to this rank's caller.
    int hard_rc = FDAPDE_OK;
    auto hard = [&](int rc) {
        if (rc != FDAPDE_OK && hard_rc == FDAPDE_OK) hard_rc = rc;
        return rc != FDAPDE_OK;
    };
    int local_ok = std::getenv("FDAPDE_ROWDIST_REFUSE") ? 0 : 1;   // (tests: this rank's share "does not fit")
    if (const char* e = std::getenv("FDAPDE_ROWDIST_FAIL_RANK"))    // (tests: a hard local failure on one rank, at the stage FDAPDE_ROWDIST_FAIL_AT names)
        if (std::atoi(e) == me && std::getenv("FDAPDE_ROWDIST_FAIL_AT") && std::atoi(std::getenv("FDAPDE_ROWDIST_FAIL_AT")) == 0) hard(fail(c, FDAPDE_EHIP, "injected failure (stage 0)"));
    if (hard(ensure_host(c, kHostPattern))) local_ok = 0;
    if (hard_rc) local_ok = 0;
    // ghosts in (owner, key) order
    std::vector<int32_t> gh;
    for (int64_t d = 0; d < nd; ++d)
        if (rd.owner_i[(size_t)d] != me) gh.push_back((int32_t)d);
    std::sort(gh.begin(), gh.end(), [&](int32_t a, int32_t b) {
        if (rd.owner_i[(size_t)a] != rd.owner_i[(size_t)b]) return rd.owner_i[(size_t)a] < rd.owner_i[(size_t)b];
        return rd.key_i[(size_t)a] < rd.key_i[(size_t)b];
    });
    std::vector<int32_t> ghost_order((size_t)nd, -1);
    for (size_t k = 0; k < gh.size(); ++k) ghost_order[(size_t)gh[k]] = (int32_t)k;
    PersistLayout pl;
    const int n_wg = rd.max_wg > 0 ? std::min(rd.max_wg, c->n_cu) : c->n_cu;
    int sym_mode = c->persist_plain ? 0 : (c->persist_sym == 2 ? 3 : c->persist_sym);
    const size_t lds_total = 160 * 1024 - 1024;
    size_t fixed = 0;
    int64_t need = 0;
    int imp_cap = 0, exp_cap = 0, S = 0;
    for (int attempt = 0; attempt < 2 && local_ok; ++attempt) {
        pl = PersistLayout{};
        const int rc = host_build_persist_layout(hs, v == 1, n_wg, 12000, pl, nullptr, sym_mode, c->persist_balance != 0, ghost_order.data(), /*allow_late=*/true);
        if (rc == FDAPDE_EUNSUPPORTED) {
            if (std::getenv("FDAPDE_DEBUG_SETUP")) std::fprintf(stderr, "rank %d: row-distributed layout %d: the host builder refuses this rank's share (sym_mode %d, %d workgroups)\n", me, v, sym_mode, n_wg);
            if (sym_mode != 0 && attempt == 0) {   // (symmetric storage leaves half the slots to rows that import: try the plain form)
                sym_mode = 0;
                continue;
            }
            local_ok = 0;
            break;
        }
        if (hard(rc)) {
            local_ok = 0;
            break;
        }
        if (pl.R > kPersistRmax) {   // (the wide form exists on one GPU only)
            local_ok = 0;
            break;
        }
        S = pl.R * kPersistT;
        imp_cap = (pl.max_imp + 63) & ~63, exp_cap = (pl.max_exp + 63) & ~63;
        fixed = pl.sym ? 8 * (size_t)(S + imp_cap) + 8 * (size_t)S + 64 : 8 * (size_t)(S + imp_cap) + 4 * (size_t)imp_cap + 2 * (size_t)exp_cap + 64;
        need = 128;
        for (int g = 0; g < pl.G; ++g) need = std::max<int64_t>(need, pl.ell_off[(size_t)g + 1] - pl.ell_off[(size_t)g] + 128);
        const bool sym_resident = fixed + 10 * (size_t)need <= lds_total;
        // symmetric storage only where it turns a streaming block into a resident one or the rows per thread are many (as build_persist_once)
        if (pl.sym && attempt == 0 && c->persist_sym == 2 && (fixed > lds_total || ((pl.n_int + pl.G - 1) / pl.G <= 2048 && !sym_resident))) {
            sym_mode = 0;
            continue;
        }
        if (pl.sym && fixed > lds_total && attempt == 0) {
            sym_mode = 0;
            continue;
        }
        break;
    }
    if (local_ok && (fixed > lds_total || (pl.R == 16 && fixed + 10 * (size_t)need <= lds_total))) {   // (no resident form for 16 rows per thread)
        if (std::getenv("FDAPDE_DEBUG_SETUP")) std::fprintf(stderr, "rank %d: row-distributed layout %d: LDS %zu B of %zu (R %d, imports <= %d)\n", me, v, fixed, lds_total, pl.R, pl.max_imp);
        local_ok = 0;
    }
    // ---- what the ranks tell each other: [needs from every rank | exported entries (local section of the board) | workgroups | ok]
    const int32_t n_ghost = local_ok ? (int32_t)pl.ghost_needed.size() : 0;
    std::vector<int64_t> need_from((size_t)W + 1, 0);   // prefix over owner ranks of the ghost list
    if (local_ok)
        for (int32_t d : pl.ghost_needed) ++need_from[(size_t)rd.owner_i[(size_t)d] + 1];
    for (int q = 0; q < W; ++q) need_from[(size_t)q + 1] += need_from[(size_t)q];
    const int RW = W + 3;
    std::vector<double> mat((size_t)W * RW, 0.0);
    for (int q = 0; q < W; ++q) mat[(size_t)me * RW + q] = (double)(need_from[(size_t)q + 1] - need_from[(size_t)q]);
    mat[(size_t)me * RW + W] = local_ok ? (double)pl.n_board : 0.0, mat[(size_t)me * RW + W + 1] = local_ok ? (double)pl.G : 0.0, mat[(size_t)me * RW + W + 2] = (double)local_ok;
    if (int rc = host_allreduce(c, mat)) return rc;
    auto M = [&](int p, int col) { return (int64_t)mat[(size_t)p * RW + col]; };
    for (int p = 0; p < W; ++p)
        if (M(p, W + 2) == 0) return hard_rc;   // some rank's share does not qualify (or failed): nobody takes this path (ok stays false everywhere)
    L.G_tot = 0, L.g_base = 0;
    for (int p = 0; p < W; ++p) {
        if (p < me) L.g_base += (int32_t)M(p, W + 1);
        L.G_tot += (int32_t)M(p, W + 1);
    }
    if (W > 64) return FDAPDE_OK;   // (one lane per rank record in the second level of the dot gather)
    // ---- key lists: to every owner the keys needed from it; from every rank the keys it needs from this one
    std::vector<int32_t> xr;
    std::vector<int64_t> soff(1, 0), roff(1, 0);
    for (int q = 0; q < W; ++q) {
        if (q == me) continue;
        const int64_t ns = M(me, q), nr = M(q, me);
        if (ns == 0 && nr == 0) continue;
        xr.push_back(q), soff.push_back(soff.back() + ns), roff.push_back(roff.back() + nr);
    }
    std::vector<double> keys_out((size_t)soff.back() + 1), keys_in((size_t)roff.back() + 1);
    {
        size_t at = 0;
        for (int32_t d : pl.ghost_needed) keys_out[at++] = (double)rd.key_i[(size_t)d];   // (owner order = peer order: both ascending)
    }
    // (a rank that cannot allocate, upload, export or map must not leave the others waiting in the next exchange: local failures are
    //  carried as flags to the agreement points, where every rank takes the same decision)
    int local_err = 0;
    std::string local_msg;
    auto soft = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && !local_err) local_err = 1, local_msg = std::string(what) + ": " + hipGetErrorString(e);
        if (e != hipSuccess) (void)hipGetLastError();
        return e == hipSuccess;
    };
    auto inject = [&](int stage) {   // (tests: FDAPDE_ROWDIST_FAIL_RANK / _AT)
        const char* r = std::getenv("FDAPDE_ROWDIST_FAIL_RANK");
        const char* a = std::getenv("FDAPDE_ROWDIST_FAIL_AT");
        if (r && a && std::atoi(r) == me && std::atoi(a) == stage && !local_err) local_err = 1, local_msg = "injected failure (stage " + std::to_string(stage) + ")";
    };
    auto agree = [&](bool* any) -> int {   // one decision for all ranks: did anybody fail locally so far?
        std::vector<double> flag(1, (double)local_err);
        if (int rc = host_allreduce(c, flag)) return rc;
        *any = flag[0] != 0.0;
        return FDAPDE_OK;
    };
    auto leave = [&]() -> int {   // every rank is leaving (ok stays false); the one that failed says why
        for (void* m : L.ipc_opened) (void)hipIpcCloseMemHandle(m);
        L.ipc_opened.clear();
        L.ps.board.release(), L.rboard.release();
        if (local_err) {
            std::fprintf(stderr, "fdapde rank %d: row-distributed set-up failed locally (%s)\n", me, local_msg.c_str());
            return fail(c, FDAPDE_EHIP, ("row-distributed set-up: " + local_msg).c_str());
        }
        return FDAPDE_OK;
    };
    DBuf<double> d_out, d_in;
    soft(d_out.upload(keys_out.data(), keys_out.size(), st), "key list upload") && soft(d_in.alloc(keys_in.size()), "key list allocation");
    inject(1);
    {
        bool any = false;
        if (int rc = agree(&any)) return rc;
        if (any) return leave();
    }
    if (int rc = pair_exchange(c, xr, soff, roff, d_out.p, d_in.p)) return rc;   // (collective itself: a transport failure is every rank's)
    soft(hipMemcpyAsync(keys_in.data(), d_in.p, sizeof(double) * keys_in.size(), hipMemcpyDeviceToHost, st), "key list download") &&
      soft(hipStreamSynchronize(st), "key list download");
    inject(2);
    // own DOFs by key
    std::vector<std::pair<int64_t, int32_t>> mine;
    for (int64_t d = 0; d < nd; ++d)
        if (rd.owner_i[(size_t)d] == me) mine.push_back({rd.key_i[(size_t)d], (int32_t)d});
    std::sort(mine.begin(), mine.end());
    struct Rexp { int32_t wg, slot, peer, pos; };
    std::vector<Rexp> rex;
    std::vector<int32_t> send_dof((size_t)roff.back());   // per-DOF values this rank sends: the DOFs the peers asked for, in their order
    int bad = 0;
    for (size_t q = 0; q < xr.size(); ++q) {
        const int p = xr[q];
        int64_t base = 0;   // p's remote section: the sections of the owners below this rank come first
        for (int r = 0; r < me; ++r) base += M(p, r);
        for (int64_t k = roff[q]; k < roff[q + 1]; ++k) {
            const int64_t key = (int64_t)keys_in[(size_t)k];
            auto it = std::lower_bound(mine.begin(), mine.end(), std::make_pair(key, (int32_t)-1));
            if (it == mine.end() || it->first != key || pl.wg_of[(size_t)it->second] < 0) {
                ++bad;   // not this rank's, or a DOF without a row here (Dirichlet): the caller's ownership / boundary data disagree across ranks
                send_dof[(size_t)k] = 0;
                continue;
            }
            const int32_t d = it->second;
            send_dof[(size_t)k] = d;
            rex.push_back({pl.wg_of[(size_t)d], pl.slot_of[(size_t)d], p, (int32_t)(base + (k - roff[q]))});
        }
    }
    {
        std::vector<double> flag{(double)(local_err ? 0 : bad), (double)local_err};   // (a rank whose download failed looked keys up in garbage: its count means nothing)
        if (int rc = host_allreduce(c, flag)) return rc;
        if (flag[1] != 0.0) return leave();
        if (flag[0] != 0.0) return fail(c, FDAPDE_EINVAL, "fdapde_rowdist_setup: a rank asked for a DOF its owner has no row for (ownership or boundary flags differ between ranks)");
    }
    std::stable_sort(rex.begin(), rex.end(), [](const Rexp& a, const Rexp& b) { return a.wg < b.wg; });
    std::vector<int32_t> rexp_off((size_t)pl.G + 1, 0), rexp_peer(rex.size() + 1), rexp_pos(rex.size() + 1);
    std::vector<uint16_t> rexp_slot(rex.size() + 1);
    for (size_t i = 0; i < rex.size(); ++i) ++rexp_off[(size_t)rex[i].wg + 1], rexp_slot[i] = (uint16_t)rex[i].slot, rexp_peer[i] = rex[i].peer, rexp_pos[i] = rex[i].pos;
    for (int g = 0; g < pl.G; ++g) rexp_off[(size_t)g + 1] += rexp_off[(size_t)g];
    // ---- boards.  Inside the rank: [local exports | dot records of the rank's workgroups x 2], ordinary device memory, exactly as on one
    //      GPU.  Across ranks: [entries imported from other ranks | one dot record per rank x 2], fine-grained, mapped by every rank
    fdapde_ctx::Persist& ps = L.ps;
    const size_t n_p = (size_t)n_ghost;
    inject(3);
    if (!local_err && soft(ps.board.alloc(dboard_offset(pl.n_board) + 2 * (size_t)pl.G * 8 + 8), "board allocation") &&
        soft(L.rboard.alloc_fine(2 * n_p + 2 * (size_t)W * 8 + 2 * (size_t)L.G_tot * 8 + 2), "board allocation (fine-grained)"))
        if (soft(hipMemsetAsync(ps.board.p, 0, sizeof(unsigned long long) * ps.board.n, st), "board clear") &&
            soft(hipMemsetAsync(L.rboard.p, 0, sizeof(unsigned long long) * L.rboard.n, st), "board clear"))
            soft(hipStreamSynchronize(st), "board clear");
    ps.epoch_next = 0, ps.attr_set = nullptr;
    const int BW = (int)sizeof(BoardBlob) + 1;   // + this rank's error flag
    std::vector<double> blobs((size_t)W * BW, 0.0);
    {
        BoardBlob b{};
        b.pid = (int64_t)getpid(), b.ptr = (uint64_t)(uintptr_t)L.rboard.p, b.device = c->device;
        if (!local_err) soft(hipIpcGetMemHandle(&b.handle, L.rboard.p), "hipIpcGetMemHandle");
        const unsigned char* raw = reinterpret_cast<const unsigned char*>(&b);
        for (int i = 0; i < BW - 1; ++i) blobs[(size_t)me * BW + i] = (double)raw[i];   // (bytes as small integers: exact through a floating-point sum)
        blobs[(size_t)me * BW + BW - 1] = (double)local_err;
    }
    if (int rc = host_allreduce(c, blobs)) return rc;
    bool any_err = false;
    for (int p = 0; p < W; ++p) any_err = any_err || blobs[(size_t)p * BW + BW - 1] != 0.0;
    std::vector<unsigned long long*> pp((size_t)W), pd((size_t)W);
    for (int p = 0; p < W && !any_err; ++p) {
        BoardBlob b;
        unsigned char* raw = reinterpret_cast<unsigned char*>(&b);
        for (int i = 0; i < BW - 1; ++i) raw[i] = (unsigned char)blobs[(size_t)p * BW + i];
        unsigned long long* base = nullptr;
        if (p == me) base = L.rboard.p;
        else if (b.pid == (int64_t)getpid()) {   // another context of this process: the pointer itself (peer access if it lives on another device)
            base = reinterpret_cast<unsigned long long*>((uintptr_t)b.ptr);
            if (b.device != c->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                if (e != hipErrorPeerAccessAlreadyEnabled) soft(e, "hipDeviceEnablePeerAccess");
                (void)hipGetLastError();
            }
        } else {
            void* mapped = nullptr;
            if (soft(hipIpcOpenMemHandle(&mapped, b.handle, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) L.ipc_opened.push_back(mapped);
            base = static_cast<unsigned long long*>(mapped);
        }
        int64_t np_p = 0;   // entries rank p imports from other ranks
        for (int r = 0; r < W; ++r) np_p += M(p, r);
        pp[(size_t)p] = base, pd[(size_t)p] = base + 2 * (size_t)np_p;
    }
    {
        std::vector<double> flag(1, (double)(local_err || any_err));
        if (int rc = host_allreduce(c, flag)) return rc;
        if (flag[0] != 0.0) {   // every rank leaves here together; ok stays false.  Boards that cannot be allocated, exported or mapped are a
                                // property of the fabric, not an error of the call: the caller falls back to the RCCL exchange
            if (local_err) std::fprintf(stderr, "fdapde rank %d: row-distributed boards unavailable (%s)\n", me, local_msg.c_str());
            local_err = 0;
            return leave();
        }
    }
    // ---- uploads (a failure here is local too: agreed on before anybody marks the layout usable)
    auto uploads = [&]() -> int {
    HIPCHK(c, ps.slot_dof.upload(pl.slot_dof.data(), pl.slot_dof.size(), st));
    HIPCHK(c, ps.ell_off.upload(pl.ell_off.data(), pl.ell_off.size(), st));
    HIPCHK(c, ps.sl_off.upload(pl.sl_off.data(), pl.sl_off.size(), st));
    HIPCHK(c, ps.ell_code.alloc(pl.ell_code.size() + 256));
    HIPCHK(c, hipMemsetAsync(ps.ell_code.p, 0, sizeof(uint16_t) * (pl.ell_code.size() + 256), st));
    HIPCHK(c, hipMemcpyAsync(ps.ell_code.p, pl.ell_code.data(), sizeof(uint16_t) * pl.ell_code.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, ps.ell_src.upload(pl.ell_src.data(), pl.ell_src.size(), st));
    HIPCHK(c, ps.exp_off.upload(pl.exp_off.data(), pl.exp_off.size(), st));
    HIPCHK(c, ps.exp_slot.upload(pl.exp_slot.data(), pl.exp_slot.size(), st));
    HIPCHK(c, ps.imp_off.upload(pl.imp_off.data(), pl.imp_off.size(), st));
    HIPCHK(c, ps.imp_pos.upload(pl.imp_pos.data(), pl.imp_pos.size(), st));
    HIPCHK(c, ps.ell_val.alloc((size_t)pl.n_entries + 256));
    HIPCHK(c, hipMemsetAsync(ps.ell_val.p, 0, sizeof(double) * ((size_t)pl.n_entries + 256), st));
    HIPCHK(c, ps.amax.alloc(1));
    HIPCHK(c, c->persist_stats.alloc(4 * 1024));
    HIPCHK(c, L.wg_late.upload(pl.wg_late.data(), pl.wg_late.size(), st));
    HIPCHK(c, L.rexp_off.upload(rexp_off.data(), rexp_off.size(), st));
    HIPCHK(c, L.rexp_slot.upload(rexp_slot.data(), rexp_slot.size(), st));
    HIPCHK(c, L.rexp_peer.upload(rexp_peer.data(), rexp_peer.size(), st));
    HIPCHK(c, L.rexp_pos.upload(rexp_pos.data(), rexp_pos.size(), st));
    HIPCHK(c, L.peer_pboard.upload(pp.data(), pp.size(), st));
    HIPCHK(c, L.peer_dboard.upload(pd.data(), pd.size(), st));
    // per-DOF value exchange of the ghost columns (Jacobi scale): receive in board order = ghost list order
    L.x_rank = xr, L.x_soff = roff, L.x_roff = soff;   // (what was RECEIVED as requests is what gets SENT as values, and the other way round)
    HIPCHK(c, L.x_send_dof.upload(send_dof.data(), send_dof.size() ? send_dof.size() : 0, st));
    HIPCHK(c, L.x_recv_dof.upload(pl.ghost_needed.data(), pl.ghost_needed.size(), st));
    HIPCHK(c, L.x_sendbuf.alloc(send_dof.size() + 1));
    HIPCHK(c, L.x_recvbuf.alloc(pl.ghost_needed.size() + 1));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDAPDE_OK;
    };
    {
        const int rc_up = uploads();
        if (rc_up != FDAPDE_OK) local_err = 1, local_msg = c->err;
        inject(4);
        bool any = false;
        if (int rc = agree(&any)) return rc;
        if (any) {
            const int rc = leave();
            return rc_up != FDAPDE_OK ? rc_up : rc;
        }
    }
    ps.stream = fixed + 10 * (size_t)need > lds_total;
    ps.lds_cap = ps.stream ? 0 : (int32_t)need, ps.imp_cap = imp_cap;
    ps.lds_bytes = fixed + (ps.stream ? 0 : 10 * (size_t)need);
    L.n_ghost = n_ghost;
    if (std::getenv("FDAPDE_DEBUG_SETUP"))
        std::fprintf(stderr, "rank %d: row-distributed CG layout %d: %d of %d workgroups (first %d) x %d rows/thread, %lld rows, %lld entries, %s%s, "
                     "local exports %lld, ghosts %d, pushed to other ranks %zu, peers %zu\n", me, v, pl.G, L.G_tot, L.g_base, pl.R, (long long)pl.n_int,
                     (long long)pl.n_entries, ps.stream ? "blocks stream" : "blocks resident", pl.sym ? ", symmetric storage" : "", (long long)pl.n_board,
                     n_ghost, rex.size(), xr.size());
    pl.slot_dof = {}, pl.ell_code = {}, pl.ell_src = {}, pl.exp_slot = {}, pl.imp_pos = {}, pl.sl_off = {}, pl.ell_off = {}, pl.wg_of = {}, pl.slot_of = {};
    ps.meta = std::move(pl);
    ps.filled = false, ps.ok = true, ps.tried = true, ps.built_plain = c->persist_plain != 0;
    L.ok = true;
    return FDAPDE_OK;
}

// the owners' value of a per-DOF vector (internal order) into the entries of the ghost columns this rank reads
int rowdist_import_ghosts(fdapde_ctx* c, int v, double* vec) {
    fdapde_ctx::RowDist::Layout& L = c->rd.lay[v];
    hipStream_t st = c->stream;
    const int64_t ns = L.x_soff.empty() ? 0 : L.x_soff.back(), nr = L.x_roff.empty() ? 0 : L.x_roff.back();
    if (ns > 0) hipLaunchKernelGGL(k_rd_gather, dim3(grid1(ns)), dim3(256), 0, st, ns, L.x_send_dof.p, vec, L.x_sendbuf.p);
    if (int rc = pair_exchange(c, L.x_rank, L.x_soff, L.x_roff, L.x_sendbuf.p, L.x_recvbuf.p)) return rc;
    if (nr > 0) hipLaunchKernelGGL(k_rd_scatter, dim3(grid1(nr)), dim3(256), 0, st, nr, L.x_recv_dof.p, L.x_recvbuf.p, vec);
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}

int fill_rowdist(fdapde_ctx* c, int v) {
    fdapde_ctx::Persist& ps = c->rd.lay[v].ps;
    hipStream_t st = c->stream;
    if (ps.meta.sym) HIPCHK(c, hipMemsetAsync(ps.amax.p, 0, sizeof(unsigned long long), st));
    if (ps.meta.n_entries > 0)
        hipLaunchKernelGGL(k_persist_fill, dim3((unsigned)((ps.meta.n_entries + 1023) / 1024)), dim3(256), 0, st, ps.meta.n_entries, ps.ell_src.p, c->sval.p, ps.ell_val.p,
                       ps.meta.sym ? ps.amax.p : (unsigned long long*)nullptr);
    HIPCHK(c, hipGetLastError());
    ps.filled = true;
    return FDAPDE_OK;
}

// the whole CG as one launch PER RANK.  COLLECTIVE.  *ran = false on every rank when any rank's launch gave up.
int run_rowdist(fdapde_ctx* c, int v, double tol2, int maxit, bool* ran, bool bicg) {
    fdapde_ctx::RowDist::Layout& L = c->rd.lay[v];
    fdapde_ctx::Persist& ps = L.ps;
    hipStream_t st = c->stream;
    const size_t n = (size_t)c->hs.n_dofs;
    if (c->persist_x.n < n) {
        HIPCHK(c, c->persist_x.alloc(n));
        HIPCHK(c, hipMemsetAsync(c->persist_x.p, 0, sizeof(double) * n, st));
    }
    if (ps.epoch_next > 0xC0000000u - 2u * (uint32_t)maxit) return fail(c, FDAPDE_EUNSUPPORTED, "row-distributed CG: epoch tags exhausted (re-create the context)");
    PersistArgs a{};
    a.maxit = maxit, a.time_phases = c->persist_time, a.tol2 = tol2;
    a.r_in = c->r.p, a.x = c->x.p, a.x_out = c->persist_x.p, a.sc = c->sc.p, a.ctl = c->ctl.p;
    a.world = c->world, a.rank = c->rank, a.n_board_local = (int32_t)ps.meta.n_board, a.timeout_first_ticks = c->rd.timeout_first_ms * 100000;
    // dot products in one hop (every workgroup's record to every rank) while the records are few, in two (rank records) beyond: a flat
    // gather over 8 x 256 workgroups is 25 MB of uncached reads per GPU and iteration
    a.g_base = L.g_base, a.G_tot = L.G_tot, a.flat_gather = c->rd.flat_gather < 0 ? (L.G_tot <= 1024 ? 1 : 0) : c->rd.flat_gather;
    a.wg_late = L.wg_late.p, a.rexp_off = L.rexp_off.p, a.rexp_slot = L.rexp_slot.p, a.rexp_peer = L.rexp_peer.p, a.rexp_pos = L.rexp_pos.p;
    a.peer_pboard = L.peer_pboard.p, a.peer_dboard = L.peer_dboard.p;
    a.pboard = ps.board.p, a.dboard = ps.board.p + dboard_offset(ps.meta.n_board), a.rboard = L.rboard.p;
    // a launch or read-back ERROR of this rank counts as "failed" in the ranks' decision first (they would wait for ever in the all-reduce
    // below, or for this rank's granules until their time-outs), and is returned to this rank's caller after it
    const int rc_launch = launch_persist(c, ps, a, /*dist=*/true, bicg);
    int hard_rc = (rc_launch != FDAPDE_OK && rc_launch != FDAPDE_EUNSUPPORTED) ? rc_launch : FDAPDE_OK;
    int failed = rc_launch != FDAPDE_OK ? 1 : 0;
    if (!failed) {
        auto read_back = [&]() -> int {
            HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(c->h_sc, c->sc.p, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
            if (a.time_phases) {
                c->persist_host_stats.resize(4 * (size_t)a.G);
                HIPCHK(c, hipMemcpyAsync(c->persist_host_stats.data(), c->persist_stats.p, 4 * (size_t)a.G * sizeof(double), hipMemcpyDeviceToHost, st));
            } else
                c->persist_host_stats.clear();
            HIPCHK(c, hipStreamSynchronize(st));
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev_p0, c->ev_p1));
            c->persist_launch_ms = ms;
            return FDAPDE_OK;
        };
        hard_rc = read_back();
        failed = hard_rc != FDAPDE_OK || c->h_ctl[3] != 0;
    }
    std::vector<double> flag(1, (double)failed);   // one decision for all ranks
    if (int rc = host_allreduce(c, flag)) return rc;
    *ran = flag[0] == 0.0;
    if (hard_rc != FDAPDE_OK) {
        *ran = false;
        return hard_rc;
    }
    if (!*ran) {
        HIPCHK(c, hipMemsetAsync(c->ctl.p + 3, 0, sizeof(int32_t), st));
        ps.epoch_next += (bicg ? 2u : 1u) * ((uint32_t)maxit + 2u);   // how far the other ranks got is unknown: past anything they can have written
    } else
        ps.epoch_next += (bicg ? 2u : 1u) * ((uint32_t)c->h_ctl[1] + 2u);
    return FDAPDE_OK;
}

void release_rowdist(fdapde_ctx* c) {
    for (auto& L : c->rd.lay) {
        for (void* m : L.ipc_opened) (void)hipIpcCloseMemHandle(m);
        L.ipc_opened.clear();
        fdapde_ctx::Persist& ps = L.ps;
        ps.slot_dof.release(), ps.sl_off.release(), ps.ell_src.release(), ps.exp_off.release(), ps.imp_off.release(), ps.imp_pos.release(), ps.ell_off.release(),
          ps.ell_code.release(), ps.exp_slot.release(), ps.ell_val.release(), ps.board.release(), ps.amax.release();
        L.rexp_off.release(), L.rexp_peer.release(), L.rexp_pos.release(), L.rexp_slot.release(), L.peer_pboard.release(), L.peer_dboard.release(), L.rboard.release(), L.wg_late.release();
        L.x_send_dof.release(), L.x_recv_dof.release(), L.x_sendbuf.release(), L.x_recvbuf.release();
        L.tried = L.ok = false;
    }
    c->rd.owned.release();
    c->rd.ready = false;
}


// the unit's code object is loaded when one of its kernels is first looked up (HIP defers it): done at context creation, so that the
// first solve of a process does not pay for it (6 ms for the smoke problem after the library was split into units)
void preload_persist() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_persist_fill));
    (void)hipGetLastError();
}

}   // namespace fdapde_engine
