// host_persist.cpp -- index work for the persistent small-problem CG (kernels_persist.h): cuts the interior block of the system
// into one contiguous row range per workgroup and lays each range out as sliced ELL in the order the workgroup's threads own the
// rows; builds the export / import lists of the vector entries neighbouring workgroups exchange every iteration.
// Host code, done once per function space and boundary mask (like host_build_solver_pattern).
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <thread>
#include <vector>

#include "internal.h"

namespace fdapde_hip {

namespace {

template <typename F> void for_each_wg(int G, F&& fn) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if ((unsigned)G < nt) nt = (unsigned)G;
    if (nt <= 1) {
        for (int g = 0; g < G; ++g) fn(g);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (int g = (int)t; g < G; g += (int)nt) fn(g);
        });
    for (auto& x : th) x.join();
}

}  // namespace

int host_build_persist_layout(const HostSpace& hs, bool use_bnd, int n_wg, int lds_entries, PersistLayout& pl, const int32_t* block_rows, int sym_mode, bool balance,
                              const int32_t* ghost_order, bool allow_late) {
    constexpr int T = kPersistT;
    const int64_t nd = hs.n_dofs;
    if (n_wg < 1 || nd < 1) return FDAPDE_EUNSUPPORTED;
    if (n_wg > T) n_wg = T;   // one thread per workgroup gathers that workgroup's dot-product partials
    auto dropped = [&](int64_t d) { return use_bnd && hs.dof_bnd_i[(size_t)d] != 0; };
    // interior rows in internal (locality) order
    std::vector<int32_t> irow_dof;
    irow_dof.reserve((size_t)nd);
    auto ghost = [&](int64_t d) { return ghost_order != nullptr && ghost_order[(size_t)d] >= 0; };
    for (int64_t d = 0; d < nd; ++d)
        if (!dropped(d) && !ghost(d)) irow_dof.push_back((int32_t)d);
    const int64_t n_int = (int64_t)irow_dof.size();
    if (n_int < 1) return FDAPDE_EUNSUPPORTED;
    auto kept = [&](int64_t row, int32_t col) { return col != row && !dropped(col); };
    int64_t nnz_kept = 0;
    for (int64_t i = 0; i < n_int; ++i) {
        const int32_t d = irow_dof[(size_t)i];
        for (int32_t k = hs.rowptr_i[(size_t)d]; k < hs.rowptr_i[(size_t)d + 1]; ++k) nnz_kept += kept(d, hs.colidx_i[(size_t)k]);
    }
    // workgroups: ~2048 rows each, more (fewer rows each) when that makes every block of the matrix fit its workgroup's LDS
    const int64_t want = persist_want_workgroups(n_int, nnz_kept, lds_entries, ghost_order != nullptr ? 0 : pl.single_rows);
    int G = (int)std::min<int64_t>(n_wg, want);
    if (G < 1) G = 1;
    int64_t rpw = (n_int + G - 1) / G;   // rows of the largest workgroup
    if (block_rows == nullptr && rpw > (int64_t)kPersistRwide * T) return FDAPDE_EUNSUPPORTED;   // too many rows for one launch of resident workgroups
    const bool sym = persist_want_sym(sym_mode, nnz_kept, G, rpw) && rpw <= (int64_t)kPersistRmax * T;   // (the wide form is plain: kPersistRwide)
    std::vector<int64_t> wgs;             // interior-row boundaries of the workgroups
    if (block_rows != nullptr) {
        G = n_wg;                         // caller-given block sizes; they add up to n_int
        wgs.assign((size_t)G + 1, 0);
        rpw = 0;
        for (int g = 0; g < G; ++g) wgs[(size_t)g + 1] = wgs[(size_t)g] + block_rows[g], rpw = std::max<int64_t>(rpw, block_rows[g]);
        if (wgs[(size_t)G] != n_int) return FDAPDE_EINVAL;
    } else {
        G = (int)((n_int + rpw - 1) / rpw);   // trailing workgroups that would stay empty are not launched
        bool uniform = true;
        if (balance && G >= 2) {   // equal cost (entries + 2 per row) instead of equal row counts: workgroup g starts at the first
                                   // interior row whose exclusive cost prefix reaches g * total / G (dev_persist.hip k_balance_bounds)
            const int64_t total = nnz_kept + 2 * n_int;
            wgs.assign((size_t)G + 1, n_int);
            int64_t cost = 0;
            int g = 0;
            for (int64_t i = 0; i < n_int; ++i) {
                while (g < G && cost >= (int64_t)g * total / G) wgs[(size_t)g++] = i;
                const int32_t d = irow_dof[(size_t)i];
                int64_t len = 0;
                for (int32_t k = hs.rowptr_i[(size_t)d]; k < hs.rowptr_i[(size_t)d + 1]; ++k) len += kept(d, hs.colidx_i[(size_t)k]);
                cost += len + 2;
            }
            uniform = false;
            int64_t mx = 0;
            for (int q = 0; q < G; ++q) {
                if (wgs[(size_t)q + 1] <= wgs[(size_t)q]) uniform = true;   // an empty workgroup (tiny systems): equal row counts
                mx = std::max(mx, wgs[(size_t)q + 1] - wgs[(size_t)q]);
            }
            const int64_t cap_rows = (int64_t)(rpw <= (int64_t)kPersistRmax * T ? kPersistRmax : kPersistRwide) * T;
            if (mx > cap_rows && rpw <= cap_rows) uniform = true;   // equal counts fit a workgroup (of that form), equal cost would not
            if (!uniform) rpw = mx;
        }
        if (uniform) {
            wgs.assign((size_t)G + 1, 0);
            for (int g = 0; g <= G; ++g) wgs[(size_t)g] = std::min<int64_t>(n_int, (int64_t)g * rpw);
        }
    }
    std::vector<int32_t> wg_of((size_t)nd, -1), slot_of((size_t)nd, -1);
    for (int g = 0; g < G; ++g)
        for (int64_t i = wgs[(size_t)g]; i < wgs[(size_t)g + 1]; ++i) wg_of[(size_t)irow_dof[(size_t)i]] = g;

    // ---- symmetric storage: who stores an in-block pair.  Start from the hash rule (persist_sym_owner), then make the stored row lengths
    //      EVEN where possible: the ELL keeps entries in lane pairs, so an odd row pays for one entry of padding (half of the rows: 5 % of
    //      the stream on 3-D P1 systems, 13 % in 2-D).  Rows of a workgroup in ascending order: a row whose stored length is odd hands the
    //      pair with its smallest in-block neighbour of HIGHER index over to (or takes it from) that neighbour -- its own length becomes
    //      even, the neighbour's parity flips and is settled when its turn comes.  Sequential inside a workgroup, independent across them
    //      (dev_persist.hip k_sym_parity runs the same walk with one wavefront per workgroup).
    std::vector<uint8_t> own;
    if (sym) {
        own.assign((size_t)hs.nnz, 0);
        for (int64_t i = 0; i < n_int; ++i) {
            const int32_t d = irow_dof[(size_t)i];
            for (int32_t k = hs.rowptr_i[(size_t)d]; k < hs.rowptr_i[(size_t)d + 1]; ++k) own[(size_t)k] = persist_sym_owner(d, hs.colidx_i[(size_t)k]) ? 1 : 0;
        }
        for_each_wg(G, [&](int g) {
            for (int64_t i = wgs[(size_t)g]; i < wgs[(size_t)g + 1]; ++i) {
                const int32_t d = irow_dof[(size_t)i];
                int32_t len = 0, k_up = -1;
                for (int32_t k = hs.rowptr_i[(size_t)d]; k < hs.rowptr_i[(size_t)d + 1]; ++k) {
                    const int32_t c = hs.colidx_i[(size_t)k];
                    if (!kept(d, c)) continue;
                    const bool in_block = wg_of[(size_t)c] == g;
                    len += !in_block || own[(size_t)k];
                    if (in_block && c > d && k_up < 0) k_up = k;   // columns are sorted: the first one is the smallest
                }
                if ((len & 1) == 0 || k_up < 0) continue;
                const int32_t c = hs.colidx_i[(size_t)k_up];
                const int32_t* lo = &hs.colidx_i[(size_t)hs.rowptr_i[(size_t)c]];
                const int32_t* hi = &hs.colidx_i[(size_t)hs.rowptr_i[(size_t)c + 1]];
                const int32_t k_mirror = (int32_t)(std::lower_bound(lo, hi, d) - &hs.colidx_i[0]);   // entry (c, d): the pattern is symmetric
                own[(size_t)k_up] ^= 1, own[(size_t)k_mirror] ^= 1;
            }
        });
    }

    // ---- rows of a workgroup: (references another workgroup, length, DOF) + its import list
    struct Key { int32_t halo, len, dof; };
    std::vector<std::vector<Key>> wg_rows((size_t)G);
    std::vector<std::vector<int32_t>> imports((size_t)G);   // DOFs of other workgroups a workgroup reads, unique
    std::vector<int32_t> row_len((size_t)nd, 0), n_halo((size_t)G, 0);
    for_each_wg(G, [&](int g) {
        const int64_t i0 = wgs[(size_t)g], i1 = wgs[(size_t)g + 1];
        std::vector<Key>& rows = wg_rows[(size_t)g];
        rows.reserve((size_t)(i1 - i0));
        std::vector<int32_t>& imp = imports[(size_t)g];
        for (int64_t i = i0; i < i1; ++i) {
            const int32_t d = irow_dof[(size_t)i];
            int32_t len = 0, halo = 0;
            for (int32_t k = hs.rowptr_i[(size_t)d]; k < hs.rowptr_i[(size_t)d + 1]; ++k) {
                const int32_t c = hs.colidx_i[(size_t)k];
                if (!kept(d, c)) continue;
                if (sym && wg_of[(size_t)c] == g && !own[(size_t)k]) continue;   // stored in row c
                ++len;
                if (wg_of[(size_t)c] != g) halo = 1, imp.push_back(c);
            }
            row_len[(size_t)d] = len;
            rows.push_back({halo, len, d});
            n_halo[(size_t)g] += halo;
        }
        std::sort(imp.begin(), imp.end());
        imp.erase(std::unique(imp.begin(), imp.end()), imp.end());
    });
    // rows per thread: the first half of a thread's passes holds rows without imports (multiplied while the neighbours' entries
    // travel), the second half everything else -- so a workgroup needs T R / 2 slots for its importing rows
    const int32_t max_halo = *std::max_element(n_halo.begin(), n_halo.end());
    // allow_late: a workgroup with more importing rows than the second half of its slots holds does not force twice the rows per thread
    // on everybody (or the refusal of the system): it is marked LATE -- the kernel fetches its imports before its first pass -- and its
    // rows fill the slots in one run
    const int R = persist_rows_per_thread(rpw, max_halo, allow_late, sym);
    if (R == 0) return FDAPDE_EUNSUPPORTED;
    const int S = R * T, nsl = S / 64, SA = (R / 2) * T;
    pl.G = G, pl.R = R, pl.nsl = nsl, pl.n_int = n_int, pl.sym = sym, pl.nnz_full = nnz_kept;
    pl.wg_late.assign((size_t)G, 0);
    if (allow_late)
        for (int g = 0; g < G; ++g) pl.wg_late[(size_t)g] = n_halo[(size_t)g] > SA ? 1 : 0;

    // ---- slots: [0, SA) rows without imports, longest first (as many as fit); [SA, S) all other rows, longest first
    pl.slot_dof.assign((size_t)G * S, -1);
    for_each_wg(G, [&](int g) {
        std::vector<Key>& rows = wg_rows[(size_t)g];
        std::sort(rows.begin(), rows.end(), [](const Key& a, const Key& b) {
            if (a.halo != b.halo) return a.halo < b.halo;
            if (a.len != b.len) return a.len > b.len;
            return a.dof < b.dof;
        });
        const bool late = allow_late && n_halo[(size_t)g] > SA;
        const size_t n_noimp = rows.size() - (size_t)n_halo[(size_t)g], n_a = late ? rows.size() : std::min<size_t>(n_noimp, (size_t)SA);
        if (late) std::stable_sort(rows.begin(), rows.end(), [](const Key& a, const Key& b) { return a.len > b.len; });
        else std::stable_sort(rows.begin() + (std::ptrdiff_t)n_a, rows.end(), [](const Key& a, const Key& b) { return a.len > b.len; });
        for (size_t i = 0; i < rows.size(); ++i) {
            const size_t s = i < n_a ? i : (size_t)SA + (i - n_a);
            pl.slot_dof[(size_t)g * S + s] = rows[i].dof;
            slot_of[(size_t)rows[i].dof] = (int32_t)s;
        }
        rows = {};
    });

    // ---- exports: a workgroup publishes the entries other workgroups import; board position = exp_off[owner] + index, in slot order
    std::vector<uint8_t> is_exp((size_t)nd, 0);
    for (int g = 0; g < G; ++g)
        for (int32_t d : imports[(size_t)g]) is_exp[(size_t)d] = 1;
    pl.exp_off.assign((size_t)G + 1, 0);
    std::vector<int32_t> board_of((size_t)nd, -1);
    pl.exp_slot.clear();
    pl.max_exp = 0;
    for (int g = 0; g < G; ++g) {
        int32_t cnt = 0;
        for (int s = 0; s < S; ++s) {
            const int32_t d = pl.slot_dof[(size_t)g * S + s];
            if (d < 0 || !is_exp[(size_t)d]) continue;
            board_of[(size_t)d] = pl.exp_off[(size_t)g] + cnt;
            pl.exp_slot.push_back((uint16_t)s);
            ++cnt;
        }
        pl.exp_off[(size_t)g + 1] = pl.exp_off[(size_t)g] + cnt;
        pl.max_exp = std::max(pl.max_exp, cnt);
    }
    pl.n_board = pl.exp_off[(size_t)G];
    // row-distributed form: the entries other ranks own sit behind the local exports, in the caller's order
    pl.ghost_needed.clear();
    if (ghost_order != nullptr) {
        for (int64_t d = 0; d < nd; ++d)
            if (ghost(d) && is_exp[(size_t)d]) pl.ghost_needed.push_back((int32_t)d);
        std::sort(pl.ghost_needed.begin(), pl.ghost_needed.end(), [&](int32_t a, int32_t b) { return ghost_order[(size_t)a] < ghost_order[(size_t)b]; });
        for (size_t k = 0; k < pl.ghost_needed.size(); ++k) board_of[(size_t)pl.ghost_needed[k]] = (int32_t)(pl.n_board + (int64_t)k);
    }
    // ---- imports in board order (neighbouring entries of one exporter are read together)
    pl.imp_off.assign((size_t)G + 1, 0);
    pl.max_imp = 0;
    for (int g = 0; g < G; ++g) {
        std::vector<int32_t>& imp = imports[(size_t)g];
        std::sort(imp.begin(), imp.end(), [&](int32_t a, int32_t b) { return board_of[(size_t)a] < board_of[(size_t)b]; });
        pl.imp_off[(size_t)g + 1] = pl.imp_off[(size_t)g] + (int32_t)imp.size();
        pl.max_imp = std::max(pl.max_imp, (int32_t)imp.size());
    }
    if (S + pl.max_imp > 65535) return FDAPDE_EUNSUPPORTED;   // 16-bit column codes
    pl.imp_pos.resize((size_t)pl.imp_off[(size_t)G]);
    pl.n_imp = pl.imp_off[(size_t)G];
    for (int g = 0; g < G; ++g)
        for (size_t h = 0; h < imports[(size_t)g].size(); ++h)
            pl.imp_pos[(size_t)pl.imp_off[(size_t)g] + h] = board_of[(size_t)imports[(size_t)g][h]];

    // ---- sliced ELL in lane pairs: slice q of a workgroup = slots [64 q, 64 q + 64), width = its longest row rounded up to even;
    //      a lane's entries (2 k, 2 k + 1) are adjacent (one 16-byte value load + one 4-byte code load): pair row k of the slice
    //      = 128 consecutive entries, lane l at 2 l
    pl.sl_off.assign((size_t)G * (nsl + 1), 0);
    pl.ell_off.assign((size_t)G + 1, 0);
    for (int g = 0; g < G; ++g) {
        int32_t off = 0;
        for (int q = 0; q < nsl; ++q) {
            int32_t w = 0;
            for (int l = 0; l < 64; ++l) {
                const int32_t d = pl.slot_dof[(size_t)g * S + (size_t)q * 64 + l];
                if (d >= 0) w = std::max(w, row_len[(size_t)d]);
            }
            pl.sl_off[(size_t)g * (nsl + 1) + q] = off;
            off += (w + 1) / 2;
        }
        pl.sl_off[(size_t)g * (nsl + 1) + nsl] = off;
        pl.ell_off[(size_t)g + 1] = pl.ell_off[(size_t)g] + (int64_t)off * 128;
    }
    pl.n_entries = pl.ell_off[(size_t)G];
    pl.ell_code.assign((size_t)pl.n_entries, 0);
    pl.ell_src.assign((size_t)pl.n_entries, -1);
    std::vector<int64_t> nnz_wg((size_t)G, 0);
    for_each_wg(G, [&](int g) {
        const std::vector<int32_t>& imp = imports[(size_t)g];   // sorted by board position: look columns up through an index
        std::vector<int32_t> by_dof(imp.size());
        std::iota(by_dof.begin(), by_dof.end(), 0);
        std::sort(by_dof.begin(), by_dof.end(), [&](int32_t a, int32_t b) { return imp[(size_t)a] < imp[(size_t)b]; });
        auto import_index = [&](int32_t dof) {
            size_t lo = 0, hi = by_dof.size();
            while (lo < hi) {
                const size_t mid = (lo + hi) / 2;
                if (imp[(size_t)by_dof[mid]] < dof) lo = mid + 1;
                else hi = mid;
            }
            return by_dof[lo];
        };
        int64_t nz = 0;
        for (int s = 0; s < S; ++s) {
            const int32_t d = pl.slot_dof[(size_t)g * S + s];
            if (d < 0 && !sym) continue;
            const int q = s / 64, l = s % 64;
            const int64_t base = pl.ell_off[(size_t)g] + (int64_t)pl.sl_off[(size_t)g * (nsl + 1) + q] * 128 + 2 * l;
            int32_t e = 0;
            for (int32_t k = d < 0 ? 0 : hs.rowptr_i[(size_t)d]; k < (d < 0 ? 0 : hs.rowptr_i[(size_t)d + 1]); ++k) {
                const int32_t c = hs.colidx_i[(size_t)k];
                if (!kept(d, c)) continue;
                if (sym && wg_of[(size_t)c] == g && !own[(size_t)k]) continue;
                const int64_t at = base + (int64_t)(e / 2) * 128 + (e & 1);
                pl.ell_src[(size_t)at] = k;
                pl.ell_code[(size_t)at] = (uint16_t)(wg_of[(size_t)c] == g ? slot_of[(size_t)c] : S + import_index(c));
                ++e, ++nz;
            }
            if (sym) {   // padding points at the lane's own slot: its (zero) transposed product meets no other lane's in the accumulator table
                const int32_t e1 = 2 * (pl.sl_off[(size_t)g * (nsl + 1) + q + 1] - pl.sl_off[(size_t)g * (nsl + 1) + q]);
                for (; e < e1; ++e) pl.ell_code[(size_t)(base + (int64_t)(e / 2) * 128 + (e & 1))] = (uint16_t)s;
            }
        }
        nnz_wg[(size_t)g] = nz;
    });
    pl.nnz = std::accumulate(nnz_wg.begin(), nnz_wg.end(), (int64_t)0);
    if (ghost_order != nullptr) pl.wg_of = std::move(wg_of), pl.slot_of = std::move(slot_of);
    return FDAPDE_OK;
}

}  // namespace fdapde_hip
