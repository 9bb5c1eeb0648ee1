// dev_persist.hip -- the resident layout of the persistent CG (kernels_persist.h) built ON THE DEVICE: the same arrays as
// host_build_persist_layout (host_persist.cpp), produced by radix sorts, scans and small kernels instead of per-workgroup host loops.
//   rows of a workgroup      one 64-bit key per interior row -- (workgroup, imports?, 255 - length, DOF) -- sorted once gives the host's
//                            (halo, length descending, DOF) order; a second key moves the rows that do not fit the import-free half
//                            behind it in the host's stable (length descending) order; slot = position in its class
//   import / export lists    (workgroup, DOF) pairs of the entries that leave a workgroup, sorted and deduplicated; exported rows ranked
//                            in slot order = board positions; imports re-sorted by board position
//   sliced ELL               slice widths by a max over 64 slots, offsets by one scan, entries written by the thread that owns the row
// FDAPDE_SETUP_CHECK=1 compares every array with the host builder's.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <string>
#include <vector>

#include "dev_persist.h"

namespace fdapde_hip {

namespace {

#define DP_CHK(expr)                                                          \
    do {                                                                      \
        hipError_t e__ = (expr);                                              \
        if (e__ != hipSuccess) {                                              \
            err = std::string(#expr) + ": " + hipGetErrorString(e__);         \
            return FDAPDE_EHIP;                                               \
        }                                                                     \
    } while (0)

// Temporaries of one build come out of an ARENA: a few large slabs, bump-allocated, released together when the build ends -- instead of ~60
// hipMalloc / hipFree pairs of 40 - 650 MB each (a hipFree waits for the device, a cold hipMalloc maps fresh pages: together a double-digit
// share of a cold fdapde_dofs_build).  A Tmp taken from the arena is not given back before the end of the build (peak: the sum of the build's
// temporaries, ~7 GB at C3's size, ~15 GB at C5's: small change on a 288 GB device); without an arena in scope a Tmp owns its allocation.
struct Arena {
    std::vector<void*> slabs;
    char* cur = nullptr;
    size_t left = 0;
    static constexpr size_t kSlab = size_t(768) << 20;
    hipError_t take(size_t bytes, void** out) {
        bytes = (bytes + 255) & ~size_t(255);
        if (bytes > left) {
            const size_t sz = bytes > kSlab ? bytes : kSlab;
            void* p = nullptr;
            const hipError_t e = hipMalloc(&p, sz);
            if (e != hipSuccess) return e;
            slabs.push_back(p), cur = static_cast<char*>(p), left = sz;
        }
        *out = cur, cur += bytes, left -= bytes;
        return hipSuccess;
    }
    ~Arena() {
        for (void* p : slabs) (void)hipFree(p);
    }
};
thread_local Arena* t_arena = nullptr;
struct ArenaScope {
    Arena* prev;
    explicit ArenaScope(Arena* a) : prev(t_arena) { t_arena = a; }
    ~ArenaScope() { t_arena = prev; }
};
template <typename T> struct Tmp {   // scratch buffer released on scope exit (or with the build's arena)
    T* p = nullptr;
    size_t n = 0;
    bool own = false;
    hipError_t alloc(size_t count) {
        reset();
        n = count;
        if (t_arena) return t_arena->take(sizeof(T) * (count ? count : 1), reinterpret_cast<void**>(&p));
        own = true;
        return hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (count ? count : 1));
    }
    void reset() {
        if (p && own) (void)hipFree(p);
        p = nullptr, n = 0, own = false;
    }
    ~Tmp() { reset(); }
};
struct Scratch {
    void* p = nullptr;
    size_t n = 0;
    hipError_t need(size_t bytes) {
        if (bytes <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        n = bytes + bytes / 4;
        return hipMalloc(&p, n);
    }
    ~Scratch() {
        if (p) (void)hipFree(p);
    }
};
inline unsigned grid_of(int64_t n) { return (unsigned)((n + 255) / 256); }

// workgroup of an interior row index: wgs[g] <= i < wgs[g + 1]
__device__ __forceinline__ int32_t wg_of_irow(const int32_t* wgs, int G, int64_t i) {
    int lo = 0, hi = G;   // invariant: wgs[lo] <= i < wgs[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (wgs[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}
// cost-balanced boundaries: workgroup g starts at the first interior row whose exclusive cost prefix (entries + 2 per row) reaches
// g * total / G; wgs[g] = number of interior rows before it
__global__ void k_balance_bounds(int G, int64_t nd, int64_t total, const int32_t* len_scan, const int32_t* irow_scan, int32_t* wgs) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > G) return;
    if (g == G) {
        wgs[G] = irow_scan[nd];
        return;
    }
    const int64_t target = (int64_t)g * total / G;
    int64_t lo = 0, hi = nd;   // first d in [0, nd] with cost(d) >= target (cost(nd) = total >= target)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)len_scan[mid] + 2 * (int64_t)irow_scan[mid] >= target) hi = mid; else lo = mid + 1;
    }
    wgs[g] = irow_scan[lo];
}
__global__ void k_wg_of(int64_t nd, const uint8_t* keep, const int32_t* irow, const int32_t* wgs, int G, int32_t* wg) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < nd) wg[d] = keep[d] ? wg_of_irow(wgs, G, irow[d]) : -1;
}
__device__ __forceinline__ bool kept_entry(const uint8_t* keep, int32_t row, int32_t col) { return col != row && keep[col]; }

__global__ void k_keep_flags(int64_t nd, const uint8_t* bnd, int use_bnd, uint8_t* keep, int32_t* keep32) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    const uint8_t k = (use_bnd && bnd[d]) ? 0 : 1;
    keep[d] = k, keep32[d] = k;
}
__global__ void k_row_lengths(int64_t nd, const int32_t* rowptr, const int32_t* colidx, const uint8_t* keep, int32_t* len) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    int32_t n = 0;
    if (keep[d])
        for (int32_t k = rowptr[d]; k < rowptr[d + 1]; ++k) n += kept_entry(keep, (int32_t)d, colidx[k]);
    len[d] = n;
}
// symmetric storage, who stores an in-block pair (host_persist.cpp has the walk in plain words).  own[k] for entry k = (row, col):
// 1 = the row stores it.  Start: the hash rule ...
__global__ void k_sym_own_init(int64_t nd, const int32_t* rowptr, const int32_t* colidx, const uint8_t* keep, int32_t* own, const int32_t* irow,
                               int32_t* kept_dof) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd || !keep[d]) return;
    if (kept_dof) kept_dof[irow[d]] = (int32_t)d;
    for (int32_t k = rowptr[d]; k < rowptr[d + 1]; ++k) own[k] = persist_sym_owner((int32_t)d, colidx[k]) ? 1 : 0;
}
// ... then the parity walk (host_persist.cpp has it in plain words): rows of a workgroup in ascending order, a row whose stored length is odd
// flips the ownership of the pair with its smallest in-block neighbour of higher index.  Which pair that is does not depend on any flip, and a
// flip changes nothing but the PARITY of the two rows involved -- so the walk splits into
//   k_sym_rows  (parallel, one thread per row): the parity of the row's stored length under the hash rule, its up-pair entry k_up, the mirror
//               entry (c, d) of that pair, and the up neighbour's position among the workgroup's rows;
//   k_sym_walk  (one workgroup per block of rows): the parities in LDS, ONE lane walks them in order -- a row that is odd when its turn comes
//               is marked and toggles its up neighbour's parity -- then all lanes apply the marked flips to own[].
// An entry is flipped at most once (as k_up by its own row only, as mirror by the one row whose up pair it is), so applying the flips after the
// walk gives the flags of the entry-by-entry walk (16 ms of dependent global loads and fences at C3's size; FDAPDE_SETUP_CHECK compares).
__global__ void k_sym_rows(int64_t nd, const int32_t* rowptr, const int32_t* colidx, const uint8_t* keep, const int32_t* wg, const int32_t* irow,
                           const int32_t* wgs, const int32_t* own, uint8_t* par, int32_t* up_k, int32_t* up_m, int32_t* up_loc) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd || !keep[d]) return;
    const int32_t g = wg[d], i = irow[d];
    int32_t len = 0, k_up = -1;
    for (int32_t k = rowptr[d]; k < rowptr[d + 1]; ++k) {
        const int32_t c = colidx[k];
        if (!kept_entry(keep, (int32_t)d, c)) continue;
        const bool inb = wg[c] == g;
        len += (!inb || own[k] != 0) ? 1 : 0;
        if (inb && c > (int32_t)d && k_up < 0) k_up = k;   // columns are sorted: the first one is the smallest
    }
    par[i] = (uint8_t)(len & 1), up_k[i] = k_up;
    int32_t km = -1, loc = -1;
    if (k_up >= 0) {
        const int32_t c = colidx[k_up];
        int32_t lo = rowptr[c], hi = rowptr[c + 1];
        while (lo < hi) {   // entry (c, d): the pattern is symmetric
            const int32_t mid = (lo + hi) >> 1;
            if (colidx[mid] < (int32_t)d) lo = mid + 1; else hi = mid;
        }
        km = lo, loc = irow[c] - wgs[g];
    }
    up_m[i] = km, up_loc[i] = loc;
}
__global__ __launch_bounds__(256) void k_sym_walk(const int32_t* wgs, const uint8_t* par_g, const int32_t* up_k, const int32_t* up_m, const int32_t* up_loc,
                                                  int32_t* own) {
    extern __shared__ int32_t sw_lds[];   // [rows] up neighbour (local row index or -1), then [rows] bytes: parity, bit 1 = marked
    const int g = blockIdx.x;
    const int32_t i0 = wgs[g], rows = wgs[g + 1] - i0;
    int32_t* up = sw_lds;
    uint8_t* par = reinterpret_cast<uint8_t*>(sw_lds + rows);
    for (int32_t i = threadIdx.x; i < rows; i += blockDim.x) up[i] = up_loc[i0 + i], par[i] = par_g[i0 + i];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int32_t i = 0; i < rows; ++i) {
            const int32_t u = up[i];
            if ((par[i] & 1) && u >= 0) par[i] = 2, par[u] ^= 1;
        }
    __syncthreads();
    for (int32_t i = threadIdx.x; i < rows; i += blockDim.x)
        if (par[i] & 2) {
            const int32_t ku = up_k[i0 + i], km = up_m[i0 + i];
            own[ku] ^= 1, own[km] ^= 1;
        }
}
// symmetric storage: a row keeps the in-block pairs it owns (own[]) and every entry of another workgroup's column
__global__ void k_row_lengths_sym(int64_t nd, const int32_t* rowptr, const int32_t* colidx, const uint8_t* keep, const int32_t* wg, const int32_t* own,
                                  int32_t* len) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    int32_t n = 0;
    if (keep[d]) {
        const int32_t g = wg[d];
        for (int32_t k = rowptr[d]; k < rowptr[d + 1]; ++k) {
            const int32_t c = colidx[k];
            n += kept_entry(keep, (int32_t)d, c) && (wg[c] != g || own[k] != 0);
        }
    }
    len[d] = n;
}
// cnt[key] += number of lanes of the wave with flag set, ONE atomic per distinct key of the wave (the keys of neighbouring rows / sorted
// entries are nearly all equal: an atomic per lane piled 256 workgroup counters with 1.6 M adds -- 1.5 ms a kernel)
__device__ __forceinline__ void wave_count_by_key(bool flag, int32_t key, int32_t* cnt) {
    unsigned long long todo = __ballot(flag);
    const int lane = (int)(threadIdx.x & 63);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int32_t k0 = __shfl(key, leader);
        const unsigned long long same = __ballot(flag && key == k0);
        if (lane == leader) atomicAdd(&cnt[k0], (int32_t)__popcll(same));
        todo &= ~same;
    }
}
// per interior row: does it read another workgroup's rows, and how many such entries; per workgroup: rows that do
__global__ void k_row_halo(int64_t nd, const int32_t* rowptr, const int32_t* colidx, const uint8_t* keep, const int32_t* wg,
                           uint8_t* halo, int32_t* n_out, int32_t* wg_halo) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t n = 0, g = -1;
    if (d < nd && keep[d]) {
        g = wg[d];
        for (int32_t k = rowptr[d]; k < rowptr[d + 1]; ++k) {
            const int32_t c = colidx[k];
            n += kept_entry(keep, (int32_t)d, c) && wg[c] != g;
        }
    }
    wave_count_by_key(n != 0, g, wg_halo);
    if (d < nd) halo[d] = n ? 1 : 0, n_out[d] = n;
}
__global__ void k_import_pairs(int64_t nd, const int32_t* rowptr, const int32_t* colidx, const uint8_t* keep, const int32_t* wg,
                               const int32_t* at, uint64_t* key) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd || !keep[d]) return;
    const int32_t g = wg[d];
    int32_t o = at[d];
    for (int32_t k = rowptr[d]; k < rowptr[d + 1]; ++k) {
        const int32_t c = colidx[k];
        if (kept_entry(keep, (int32_t)d, c) && wg[c] != g) key[o++] = ((uint64_t)(uint32_t)g << 32) | (uint32_t)c;
    }
}
__global__ void k_unique_flags(int64_t n, const uint64_t* key, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}
__global__ void k_compact_keys(int64_t n, const uint64_t* key, const int32_t* flag, const int32_t* pos, uint64_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[pos[i]] = key[i];
}
// first ordering of the rows of a workgroup: (imports?, length descending, DOF)
__global__ void k_row_keys1(int64_t nd, const uint8_t* keep, const int32_t* irow, const int32_t* wg, const uint8_t* halo, const int32_t* len,
                            uint64_t* key, int32_t* val) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd || !keep[d]) return;
    const int32_t i = irow[d];
    key[i] = ((uint64_t)(uint32_t)wg[d] << 41) | ((uint64_t)halo[d] << 40) | ((uint64_t)(255 - len[d]) << 32) | (uint32_t)d;
    val[i] = (int32_t)d;
}
// second ordering: the first `sa` import-free rows keep their rank; the others follow in (length descending, imports?, DOF) order
__global__ void k_row_keys2(int64_t n_int, const int32_t* dof_sorted, const int32_t* wgs, int G, int32_t sa, const uint8_t* halo, const int32_t* len,
                            uint64_t* key) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_int) return;
    const int32_t d = dof_sorted[p];
    const int64_t g = wg_of_irow(wgs, G, p), rank = p - wgs[g];   // the rows of workgroup g are positions [wgs[g], wgs[g + 1]) of the first ordering
    const bool first = !halo[d] && rank < sa;
    const uint64_t payload = first ? (uint64_t)rank : (((uint64_t)(255 - len[d]) << 33) | ((uint64_t)halo[d] << 32) | (uint32_t)d);
    key[p] = ((uint64_t)g << 42) | ((uint64_t)(first ? 0 : 1) << 41) | payload;
}
__global__ void k_assign_slots(int64_t n_int, const int32_t* dof_sorted, const int32_t* wgs, int G, int32_t sa, int32_t S, const int32_t* wg_noimp, int32_t* slot_of,
                               int32_t* slot_dof) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_int) return;
    const int32_t d = dof_sorted[p];
    const int64_t g = wg_of_irow(wgs, G, p), i = p - wgs[g];
    const int32_t n_a = min(wg_noimp[g], sa);
    const int32_t slot = i < n_a ? (int32_t)i : sa + (int32_t)(i - n_a);
    slot_of[d] = slot;
    slot_dof[g * S + slot] = d;
}
__global__ void k_mark_exports(int64_t n_imp, const uint64_t* imp_key, uint8_t* is_exp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_imp) is_exp[imp_key[i] & 0xffffffffu] = 1;
}
__global__ void k_export_keys(int64_t nd, const uint8_t* is_exp, const int32_t* wg, const int32_t* slot_of, uint64_t* key, int32_t* val,
                              int32_t* n_out, int32_t* wg_exp) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool e = d < nd && is_exp[d];
    const int32_t g = e ? wg[d] : -1;
    const unsigned long long m = __ballot(e);   // the wave's exports take consecutive places behind ONE add (the list is sorted afterwards)
    if (m == 0) return;
    const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(n_out, (int32_t)__popcll(m));
    base = __shfl(base, leader);
    if (e) {
        const int32_t o = base + (int32_t)__popcll(m & ((1ull << lane) - 1ull));
        key[o] = ((uint64_t)(uint32_t)g << 32) | (uint32_t)slot_of[d];
        val[o] = (int32_t)d;
    }
    wave_count_by_key(e, g, wg_exp);
}
__global__ void k_board_tables(int64_t n_board, const uint64_t* key_sorted, const int32_t* dof_sorted, uint16_t* exp_slot, int32_t* board_of) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_board) return;
    exp_slot[i] = (uint16_t)(key_sorted[i] & 0xffffu);
    board_of[dof_sorted[i]] = (int32_t)i;
}
__global__ void k_import_board_keys(int64_t n_imp, const uint64_t* imp_key, const int32_t* board_of, uint64_t* key, int32_t* wg_imp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n_imp;
    const uint64_t g = in ? imp_key[i] >> 32 : 0;
    if (in) key[i] = (g << 32) | (uint32_t)board_of[imp_key[i] & 0xffffffffu];
    wave_count_by_key(in, (int32_t)g, wg_imp);
}
__global__ void k_count_groups(int64_t n, const uint64_t* key, int32_t* count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    wave_count_by_key(i < n, i < n ? (int32_t)(key[i] >> 32) : 0, count);
}
__global__ void k_low32(int64_t n, const uint64_t* key, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)(key[i] & 0xffffffffu);
}
__global__ void k_slice_pairs(int64_t n_sl, const int32_t* slot_dof, const int32_t* len, int32_t* pairs) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // global slice index g * nsl + q
    if (q >= n_sl) return;
    int32_t w = 0;
    for (int l = 0; l < 64; ++l) {
        const int32_t d = slot_dof[q * 64 + l];
        if (d >= 0) w = max(w, len[d]);
    }
    pairs[q] = (w + 1) / 2;
}
__global__ void k_slice_offsets(int G, int nsl, const int32_t* scan, int32_t* sl_off, int64_t* ell_off, int32_t* max_block) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)G * (nsl + 1)) return;
    const int64_t g = t / (nsl + 1), q = t - g * (nsl + 1);
    const int32_t base = scan[g * nsl];
    sl_off[t] = scan[g * nsl + q] - base;   // scan holds G * nsl + 1 entries: q == nsl reads the next workgroup's base / the total
    if (q == nsl) atomicMax(max_block, scan[g * nsl + nsl] - base);
    if (q == 0) ell_off[g] = (int64_t)base * 128;
    if (t == 0) ell_off[G] = (int64_t)scan[(int64_t)G * nsl] * 128;
}
__global__ void k_fill_ell(int64_t n_slots, int32_t S, int nsl, const int32_t* slot_dof, const int32_t* rowptr, const int32_t* colidx, const uint8_t* keep,
                           const int32_t* wg, const int32_t* slot_of, const int32_t* board_of, const int32_t* imp_off,
                           const int32_t* imp_pos, const int32_t* sl_off, const int64_t* ell_off, int sym, const int32_t* own, uint16_t* code, int32_t* src) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // g * S + slot
    if (t >= n_slots) return;
    const int32_t d = slot_dof[t];
    if (d < 0 && !sym) return;
    const int64_t g = t / S;
    const int32_t s = (int32_t)(t - g * S), q = s / 64, l = s % 64;
    const int64_t base = ell_off[g] + (int64_t)sl_off[g * (nsl + 1) + q] * 128 + 2 * l;
    int32_t e = 0;
    for (int32_t k = d < 0 ? 0 : rowptr[d]; k < (d < 0 ? 0 : rowptr[d + 1]); ++k) {
        const int32_t c = colidx[k];
        if (!kept_entry(keep, d, c)) continue;
        if (sym && wg[c] == (int32_t)g && own[k] == 0) continue;   // stored in row c
        const int64_t at = base + (int64_t)(e / 2) * 128 + (e & 1);
        src[at] = k;
        if (wg[c] == (int32_t)g) {
            code[at] = (uint16_t)slot_of[c];
        } else {   // position of the column's board entry in this workgroup's import list (sorted by board position)
            int32_t lo = imp_off[g], hi = imp_off[g + 1];
            const int32_t b = board_of ? board_of[c] : c;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (imp_pos[mid] < b) lo = mid + 1; else hi = mid;
            }
            code[at] = (uint16_t)(S + (lo - imp_off[g]));
        }
        ++e;
    }
    if (sym) {   // padding points at the lane's own slot: its (zero) transposed product then meets no other lane's in the accumulator table
        const int32_t e1 = 2 * (sl_off[g * (nsl + 1) + q + 1] - sl_off[g * (nsl + 1) + q]);
        for (; e < e1; ++e) code[base + (int64_t)(e / 2) * 128 + (e & 1)] = (uint16_t)s;
    }
}
__global__ void k_drop_list(int64_t nd, const uint8_t* keep, const int32_t* irow_scan, int32_t* drop) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < nd && !keep[d]) drop[d - irow_scan[d]] = (int32_t)d;   // rank among the dropped rows = d - (kept rows before d)
}
__global__ void k_scatter_irow(int64_t nd, const uint8_t* keep, const int32_t* irow_scan, int32_t* irow) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < nd) irow[d] = keep[d] ? irow_scan[d] : -1 - irow_scan[d];   // dropped rows: negative, never compared as a workgroup
}

template <typename K, typename V>
int sort_pairs(Scratch& sc, K* k_in, K* k_out, V* v_in, V* v_out, int64_t n, int end_bit, hipStream_t st, std::string& err) {
    size_t need = 0;
    DP_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st));
    DP_CHK(sc.need(need));
    DP_CHK(hipcub::DeviceRadixSort::SortPairs(sc.p, need, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st));
    return FDAPDE_OK;
}
template <typename K> int sort_keys(Scratch& sc, K* k_in, K* k_out, int64_t n, int end_bit, hipStream_t st, std::string& err) {
    size_t need = 0;
    DP_CHK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, k_in, k_out, (int)n, 0, end_bit, st));
    DP_CHK(sc.need(need));
    DP_CHK(hipcub::DeviceRadixSort::SortKeys(sc.p, need, k_in, k_out, (int)n, 0, end_bit, st));
    return FDAPDE_OK;
}
template <typename In, typename Out> int exclusive_sum(Scratch& sc, In* in, Out* out, int64_t n, hipStream_t st, std::string& err) {
    size_t need = 0;
    DP_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, (int)n, st));
    DP_CHK(sc.need(need));
    DP_CHK(hipcub::DeviceScan::ExclusiveSum(sc.p, need, in, out, (int)n, st));
    return FDAPDE_OK;
}

}  // namespace

void dev_persist_release(DevPersist* p) {
    if (!p) return;
    for (void* q : {(void*)p->slot_dof, (void*)p->sl_off, (void*)p->ell_src, (void*)p->exp_off, (void*)p->imp_off, (void*)p->imp_pos, (void*)p->ell_off,
                    (void*)p->ell_code, (void*)p->exp_slot, (void*)p->drop_dof})
        if (q) (void)hipFree(q);
    *p = DevPersist{};
}

int dev_build_persist_layout(int64_t nd, int32_t max_row, const int32_t* d_rowptr, const int32_t* d_colidx, const uint8_t* d_bnd, bool use_bnd,
                             int n_wg, int lds_entries, int blocked_rows, const int32_t* block_rows, int sym_mode, bool balance, void* stream, PersistLayout& pl, DevPersist* out,
                             std::string& err) {
    constexpr int T = kPersistT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_wg < 1 || nd < 1 || max_row > 255) return FDAPDE_EUNSUPPORTED;   // the row keys carry 255 - length in 8 bits
    const bool blocked = blocked_rows > 0;   // layout of the blocked-ELL SpMV (kernels_persist.h, k_spmv_blocked): any number of workgroups of
                                             // ~blocked_rows rows, imports addressed by DOF id in the global vector, no board
    if (!blocked && n_wg > T) n_wg = T;
    Arena arena;   // (declared before everything that allocates from it: released last)
    ArenaScope arena_scope(&arena);
    Scratch sc;
    DevPersist o;
    struct Guard {
        DevPersist* p;
        bool armed = true;
        ~Guard() {
            if (armed) dev_persist_release(p);
        }
    } guard{&o};
    const bool dbg = std::getenv("FDAPDE_DEBUG_SETUP") != nullptr;
    auto t_ph = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {   // FDAPDE_DEBUG_SETUP: wall time of every stage (the stream is drained at each mark)
        if (!dbg) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "persist layout %-26s %8.2f ms\n", name, std::chrono::duration<double, std::milli>(now - t_ph).count());
        t_ph = now;
    };
    Tmp<uint8_t> keep, halo, is_exp;
    Tmp<int32_t> keep32, irow_scan, irow, len, n_imp_row, imp_at, wg_cnt;   // wg_cnt: [0..G) halo rows, [G..2G) exports, [2G..3G) imports
    DP_CHK(keep.alloc((size_t)nd));
    DP_CHK(keep32.alloc((size_t)nd + 1));
    DP_CHK(irow_scan.alloc((size_t)nd + 1));
    DP_CHK(irow.alloc((size_t)nd));
    DP_CHK(len.alloc((size_t)nd + 1));
    DP_CHK(hipMemsetAsync(keep32.p + nd, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_keep_flags, dim3(grid_of(nd)), dim3(256), 0, st, nd, d_bnd, use_bnd ? 1 : 0, keep.p, keep32.p);
    if (int rc = exclusive_sum(sc, keep32.p, irow_scan.p, nd + 1, st, err)) return rc;
    hipLaunchKernelGGL(k_scatter_irow, dim3(grid_of(nd)), dim3(256), 0, st, nd, keep.p, irow_scan.p, irow.p);
    DP_CHK(hipMemsetAsync(len.p + nd, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_row_lengths, dim3(grid_of(nd)), dim3(256), 0, st, nd, d_rowptr, d_colidx, keep.p, len.p);
    Tmp<int32_t> len_scan;
    DP_CHK(len_scan.alloc((size_t)nd + 1));
    if (int rc = exclusive_sum(sc, len.p, len_scan.p, nd + 1, st, err)) return rc;
    int32_t h_nint = 0, h_nnz = 0;
    DP_CHK(hipMemcpyAsync(&h_nint, irow_scan.p + nd, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_CHK(hipMemcpyAsync(&h_nnz, len_scan.p + nd, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_CHK(hipStreamSynchronize(st));
    const int64_t n_int = h_nint, nnz_kept = h_nnz;
    if (n_int < 1) return FDAPDE_EUNSUPPORTED;
    // workgroups: ~2048 rows each, more (fewer rows each) when that makes every block of the matrix fit its workgroup's LDS
    const int64_t want = persist_want_workgroups(n_int, nnz_kept, lds_entries, blocked ? 0 : pl.single_rows);
    int G = (int)std::min<int64_t>(n_wg, want);
    if (blocked) G = (int)((n_int + blocked_rows - 1) / blocked_rows);
    if (G < 1) G = 1;
    if (G >= (1 << 20)) return FDAPDE_EUNSUPPORTED;   // 20 bits of the row keys
    int64_t rpw = (n_int + G - 1) / G;   // rows of the largest workgroup
    if (!blocked && block_rows == nullptr && rpw > (int64_t)kPersistRwide * T) return FDAPDE_EUNSUPPORTED;   // too many rows for one launch of resident workgroups
    const bool sym = !blocked && persist_want_sym(sym_mode, nnz_kept, G, rpw) && rpw <= (int64_t)kPersistRmax * T;   // (the wide form is plain: kPersistRwide)
    std::vector<int32_t> h_wgs;           // interior-row boundaries of the workgroups
    Tmp<int32_t> wgs, wg;
    bool uniform = true;
    if (block_rows != nullptr && !blocked) {
        G = n_wg;   // caller-given block sizes; they add up to n_int
        h_wgs.assign((size_t)G + 1, 0);
        rpw = 0;
        for (int g = 0; g < G; ++g) h_wgs[(size_t)g + 1] = h_wgs[(size_t)g] + block_rows[g], rpw = std::max<int64_t>(rpw, block_rows[g]);
        if (h_wgs[(size_t)G] != n_int) return FDAPDE_EINVAL;
        uniform = false;
    } else {
        G = (int)((n_int + rpw - 1) / rpw);   // trailing workgroups that would stay empty are not launched
        DP_CHK(wgs.alloc((size_t)G + 1));
        if (balance && !blocked && G >= 2) {   // equal cost (entries + 2 per row) instead of equal row counts
            hipLaunchKernelGGL(k_balance_bounds, dim3((G + 256) / 256), dim3(256), 0, st, G, nd, nnz_kept + 2 * n_int, len_scan.p, irow_scan.p, wgs.p);
            h_wgs.assign((size_t)G + 1, 0);
            DP_CHK(hipMemcpyAsync(h_wgs.data(), wgs.p, sizeof(int32_t) * ((size_t)G + 1), hipMemcpyDeviceToHost, st));
            DP_CHK(hipStreamSynchronize(st));
            uniform = false;
            int64_t mx = 0;
            for (int g = 0; g < G; ++g) {
                if (h_wgs[(size_t)g + 1] <= h_wgs[(size_t)g]) uniform = true;   // an empty workgroup (tiny systems): equal row counts
                mx = std::max<int64_t>(mx, h_wgs[(size_t)g + 1] - h_wgs[(size_t)g]);
            }
            const int64_t cap_rows = (int64_t)(rpw <= (int64_t)kPersistRmax * T ? kPersistRmax : kPersistRwide) * T;
            if (mx > cap_rows && rpw <= cap_rows) uniform = true;   // equal counts fit a workgroup (of that form), equal cost would not
            if (!uniform) rpw = mx;
        }
        if (uniform) {
            h_wgs.assign((size_t)G + 1, 0);
            for (int g = 0; g <= G; ++g) h_wgs[(size_t)g] = (int32_t)std::min<int64_t>(n_int, (int64_t)g * rpw);
        }
    }
    if (block_rows != nullptr && !blocked) DP_CHK(wgs.alloc((size_t)G + 1));
    DP_CHK(wg.alloc((size_t)nd));
    if (uniform || block_rows != nullptr) DP_CHK(hipMemcpyAsync(wgs.p, h_wgs.data(), sizeof(int32_t) * ((size_t)G + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_wg_of, dim3(grid_of(nd)), dim3(256), 0, st, nd, keep.p, irow.p, wgs.p, G, wg.p);
    phase("rows, workgroups");
    int64_t nnz_stored = nnz_kept;
    Tmp<int32_t> own;   // symmetric storage: per entry of the pattern, does its row store the pair
    if (sym) {   // ownership of the in-block pairs (hash rule, then rows made even), then the rows' stored lengths
        int32_t h_nnz_full = 0;
        DP_CHK(hipMemcpyAsync(&h_nnz_full, d_rowptr + nd, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        DP_CHK(hipStreamSynchronize(st));
        DP_CHK(own.alloc((size_t)(h_nnz_full > 0 ? h_nnz_full : 1)));
        hipLaunchKernelGGL(k_sym_own_init, dim3(grid_of(nd)), dim3(256), 0, st, nd, d_rowptr, d_colidx, keep.p, own.p, irow.p, (int32_t*)nullptr);
        {
            Tmp<uint8_t> par;
            Tmp<int32_t> up_k, up_m, up_loc;
            DP_CHK(par.alloc((size_t)n_int));
            DP_CHK(up_k.alloc((size_t)n_int));
            DP_CHK(up_m.alloc((size_t)n_int));
            DP_CHK(up_loc.alloc((size_t)n_int));
            hipLaunchKernelGGL(k_sym_rows, dim3(grid_of(nd)), dim3(256), 0, st, nd, d_rowptr, d_colidx, keep.p, wg.p, irow.p, wgs.p, own.p, par.p, up_k.p, up_m.p,
                               up_loc.p);
            const size_t lds = (size_t)rpw * 5 + 16;   // rpw = rows of the largest workgroup (<= kPersistRmax * T: 40 KB)
            hipLaunchKernelGGL(k_sym_walk, dim3(G), dim3(256), lds, st, wgs.p, par.p, up_k.p, up_m.p, up_loc.p, own.p);
            DP_CHK(hipGetLastError());
            if (!t_arena) DP_CHK(hipStreamSynchronize(st));   // (temporaries that own their memory are released on leaving this scope)
        }
        hipLaunchKernelGGL(k_row_lengths_sym, dim3(grid_of(nd)), dim3(256), 0, st, nd, d_rowptr, d_colidx, keep.p, wg.p, own.p, len.p);
        if (int rc = exclusive_sum(sc, len.p, len_scan.p, nd + 1, st, err)) return rc;
        DP_CHK(hipMemcpyAsync(&h_nnz, len_scan.p + nd, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        DP_CHK(hipStreamSynchronize(st));
        nnz_stored = h_nnz;
    }
    phase("pair ownership (sym)");
    pl.n_drop = nd - n_int, pl.sym = sym;
    if (blocked) {
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.drop_dof), sizeof(int32_t) * (size_t)(pl.n_drop ? pl.n_drop : 1)));
        hipLaunchKernelGGL(k_drop_list, dim3(grid_of(nd)), dim3(256), 0, st, nd, keep.p, irow_scan.p, o.drop_dof);
    }
    // ---- rows that import, import pairs
    DP_CHK(halo.alloc((size_t)nd));
    DP_CHK(n_imp_row.alloc((size_t)nd + 1));
    DP_CHK(imp_at.alloc((size_t)nd + 1));
    DP_CHK(wg_cnt.alloc((size_t)3 * G));
    DP_CHK(hipMemsetAsync(wg_cnt.p, 0, sizeof(int32_t) * 3 * (size_t)G, st));
    DP_CHK(hipMemsetAsync(n_imp_row.p + nd, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_row_halo, dim3(grid_of(nd)), dim3(256), 0, st, nd, d_rowptr, d_colidx, keep.p, wg.p, halo.p, n_imp_row.p, wg_cnt.p);
    if (int rc = exclusive_sum(sc, n_imp_row.p, imp_at.p, nd + 1, st, err)) return rc;
    int32_t h_pairs = 0;
    std::vector<int32_t> h_cnt(3 * (size_t)G);
    DP_CHK(hipMemcpyAsync(&h_pairs, imp_at.p + nd, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_CHK(hipMemcpyAsync(h_cnt.data(), wg_cnt.p, sizeof(int32_t) * (size_t)G, hipMemcpyDeviceToHost, st));
    DP_CHK(hipStreamSynchronize(st));
    int32_t max_halo = 0;
    for (int g = 0; g < G; ++g) max_halo = std::max(max_halo, h_cnt[(size_t)g]);
    const int R = persist_rows_per_thread(rpw, max_halo, blocked, sym);   // (the blocked SpMV has no import-free phase)
    if (R == 0) return FDAPDE_EUNSUPPORTED;
    const int S = R * T, nsl = S / 64, SA = blocked ? S : (R / 2) * T;   // blocked: one class, plain (imports?, length, DOF) order
    pl.G = G, pl.R = R, pl.nsl = nsl, pl.n_int = n_int, pl.nnz = nnz_stored, pl.nnz_full = nnz_kept;
    Tmp<uint64_t> imp_key;   // unique (workgroup, DOF) imports, DOF ascending inside a workgroup
    int64_t n_imp = 0;
    {
        Tmp<uint64_t> k_a, k_s;
        Tmp<int32_t> flag, pos;
        DP_CHK(k_a.alloc((size_t)h_pairs));
        DP_CHK(k_s.alloc((size_t)h_pairs));
        DP_CHK(flag.alloc((size_t)h_pairs + 1));
        DP_CHK(pos.alloc((size_t)h_pairs + 1));
        hipLaunchKernelGGL(k_import_pairs, dim3(grid_of(nd)), dim3(256), 0, st, nd, d_rowptr, d_colidx, keep.p, wg.p, imp_at.p, k_a.p);
        if (h_pairs > 0) {
            if (int rc = sort_keys(sc, k_a.p, k_s.p, h_pairs, 53, st, err)) return rc;
            DP_CHK(hipMemsetAsync(flag.p + h_pairs, 0, sizeof(int32_t), st));
            hipLaunchKernelGGL(k_unique_flags, dim3(grid_of(h_pairs)), dim3(256), 0, st, (int64_t)h_pairs, k_s.p, flag.p);
            if (int rc = exclusive_sum(sc, flag.p, pos.p, (int64_t)h_pairs + 1, st, err)) return rc;
            int32_t h_n = 0;
            DP_CHK(hipMemcpyAsync(&h_n, pos.p + h_pairs, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            DP_CHK(hipStreamSynchronize(st));
            n_imp = h_n;
            DP_CHK(imp_key.alloc((size_t)n_imp));
            hipLaunchKernelGGL(k_compact_keys, dim3(grid_of(h_pairs)), dim3(256), 0, st, (int64_t)h_pairs, k_s.p, flag.p, pos.p, imp_key.p);
            DP_CHK(hipStreamSynchronize(st));
        }
    }
    phase("import pairs");
    // ---- slots
    Tmp<int32_t> slot_of, dof1, wg_noimp;
    DP_CHK(slot_of.alloc((size_t)nd));
    DP_CHK(dof1.alloc((size_t)n_int));
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.slot_dof), sizeof(int32_t) * (size_t)G * S));
    DP_CHK(hipMemsetAsync(o.slot_dof, 0xff, sizeof(int32_t) * (size_t)G * S, st));
    {
        Tmp<uint64_t> k_a, k_s;
        Tmp<int32_t> v_a, v_s;
        DP_CHK(k_a.alloc((size_t)n_int));
        DP_CHK(k_s.alloc((size_t)n_int));
        DP_CHK(v_a.alloc((size_t)n_int));
        DP_CHK(v_s.alloc((size_t)n_int));
        hipLaunchKernelGGL(k_row_keys1, dim3(grid_of(nd)), dim3(256), 0, st, nd, keep.p, irow.p, wg.p, halo.p, len.p, k_a.p, v_a.p);
        if (int rc = sort_pairs(sc, k_a.p, k_s.p, v_a.p, v_s.p, n_int, 62, st, err)) return rc;
        hipLaunchKernelGGL(k_row_keys2, dim3(grid_of(n_int)), dim3(256), 0, st, n_int, v_s.p, wgs.p, G, (int32_t)SA, halo.p, len.p, k_a.p);
        if (int rc = sort_pairs(sc, k_a.p, k_s.p, v_s.p, dof1.p, n_int, 63, st, err)) return rc;
        std::vector<int32_t> h_noimp((size_t)G);
        for (int g = 0; g < G; ++g)   // blocked: one class -- every row counts as "fits the first class", so that slot = position
            h_noimp[(size_t)g] = (h_wgs[(size_t)g + 1] - h_wgs[(size_t)g]) - (blocked ? 0 : h_cnt[(size_t)g]);
        DP_CHK(wg_noimp.alloc((size_t)G));
        DP_CHK(hipMemcpyAsync(wg_noimp.p, h_noimp.data(), sizeof(int32_t) * (size_t)G, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_assign_slots, dim3(grid_of(n_int)), dim3(256), 0, st, n_int, dof1.p, wgs.p, G, (int32_t)SA, (int32_t)S, wg_noimp.p, slot_of.p,
                           o.slot_dof);
        DP_CHK(hipStreamSynchronize(st));
    }
    phase("slots");
    // ---- exports (board) and imports in board order
    Tmp<int32_t> board_of;
    DP_CHK(board_of.alloc((size_t)nd));
    DP_CHK(is_exp.alloc((size_t)nd));
    DP_CHK(hipMemsetAsync(is_exp.p, 0, (size_t)nd, st));
    if (n_imp > 0 && !blocked) hipLaunchKernelGGL(k_mark_exports, dim3(grid_of(n_imp)), dim3(256), 0, st, n_imp, imp_key.p, is_exp.p);
    int64_t n_board = 0;
    if (!blocked) {
        Tmp<uint64_t> k_a, k_s;
        Tmp<int32_t> v_a, v_s, cnt;
        const int64_t cap = n_imp > 0 ? n_imp : 1;   // exported DOFs <= imported (workgroup, DOF) pairs
        DP_CHK(k_a.alloc((size_t)cap));
        DP_CHK(k_s.alloc((size_t)cap));
        DP_CHK(v_a.alloc((size_t)cap));
        DP_CHK(v_s.alloc((size_t)cap));
        DP_CHK(cnt.alloc(1));
        DP_CHK(hipMemsetAsync(cnt.p, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_export_keys, dim3(grid_of(nd)), dim3(256), 0, st, nd, is_exp.p, wg.p, slot_of.p, k_a.p, v_a.p, cnt.p, wg_cnt.p + G);
        int32_t h_nb = 0;
        DP_CHK(hipMemcpyAsync(&h_nb, cnt.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        DP_CHK(hipStreamSynchronize(st));
        n_board = h_nb;
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.exp_slot), sizeof(uint16_t) * (size_t)(n_board ? n_board : 1)));
        if (n_board > 0) {
            if (int rc = sort_pairs(sc, k_a.p, k_s.p, v_a.p, v_s.p, n_board, 42, st, err)) return rc;
            hipLaunchKernelGGL(k_board_tables, dim3(grid_of(n_board)), dim3(256), 0, st, n_board, k_s.p, v_s.p, o.exp_slot, board_of.p);
        }
        DP_CHK(hipStreamSynchronize(st));
    }
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.imp_pos), sizeof(int32_t) * (size_t)(n_imp ? n_imp : 1)));
    if (blocked) {   // imports = DOF ids, ascending inside a workgroup (the order of the unique import keys)
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.exp_slot), sizeof(uint16_t)));
        if (n_imp > 0) {
            hipLaunchKernelGGL(k_low32, dim3(grid_of(n_imp)), dim3(256), 0, st, n_imp, imp_key.p, o.imp_pos);
            hipLaunchKernelGGL(k_count_groups, dim3(grid_of(n_imp)), dim3(256), 0, st, n_imp, imp_key.p, wg_cnt.p + 2 * G);
        }
    } else if (n_imp > 0) {
        Tmp<uint64_t> k_a, k_s;
        DP_CHK(k_a.alloc((size_t)n_imp));
        DP_CHK(k_s.alloc((size_t)n_imp));
        hipLaunchKernelGGL(k_import_board_keys, dim3(grid_of(n_imp)), dim3(256), 0, st, n_imp, imp_key.p, board_of.p, k_a.p, wg_cnt.p + 2 * G);
        if (int rc = sort_keys(sc, k_a.p, k_s.p, n_imp, 42, st, err)) return rc;
        hipLaunchKernelGGL(k_low32, dim3(grid_of(n_imp)), dim3(256), 0, st, n_imp, k_s.p, o.imp_pos);
        DP_CHK(hipStreamSynchronize(st));
    }
    DP_CHK(hipMemcpyAsync(h_cnt.data() + G, wg_cnt.p + G, sizeof(int32_t) * 2 * (size_t)G, hipMemcpyDeviceToHost, st));
    DP_CHK(hipStreamSynchronize(st));
    std::vector<int32_t> h_exp_off((size_t)G + 1, 0), h_imp_off((size_t)G + 1, 0);
    pl.max_exp = pl.max_imp = 0;
    for (int g = 0; g < G; ++g) {
        h_exp_off[(size_t)g + 1] = h_exp_off[(size_t)g] + h_cnt[(size_t)G + g], pl.max_exp = std::max(pl.max_exp, h_cnt[(size_t)G + g]);
        h_imp_off[(size_t)g + 1] = h_imp_off[(size_t)g] + h_cnt[(size_t)2 * G + g], pl.max_imp = std::max(pl.max_imp, h_cnt[(size_t)2 * G + g]);
    }
    pl.n_board = n_board, pl.n_imp = n_imp;
    if (S + pl.max_imp > 65535) return FDAPDE_EUNSUPPORTED;
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.exp_off), sizeof(int32_t) * ((size_t)G + 1)));
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.imp_off), sizeof(int32_t) * ((size_t)G + 1)));
    DP_CHK(hipMemcpyAsync(o.exp_off, h_exp_off.data(), sizeof(int32_t) * ((size_t)G + 1), hipMemcpyHostToDevice, st));
    DP_CHK(hipMemcpyAsync(o.imp_off, h_imp_off.data(), sizeof(int32_t) * ((size_t)G + 1), hipMemcpyHostToDevice, st));
    phase("boards");
    // ---- sliced ELL in lane pairs
    const int64_t n_sl = (int64_t)G * nsl;
    Tmp<int32_t> pairs, pscan, max_block;
    DP_CHK(pairs.alloc((size_t)n_sl + 1));
    DP_CHK(pscan.alloc((size_t)n_sl + 1));
    DP_CHK(max_block.alloc(1));
    DP_CHK(hipMemsetAsync(pairs.p + n_sl, 0, sizeof(int32_t), st));
    DP_CHK(hipMemsetAsync(max_block.p, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_slice_pairs, dim3(grid_of(n_sl)), dim3(256), 0, st, n_sl, o.slot_dof, len.p, pairs.p);
    if (int rc = exclusive_sum(sc, pairs.p, pscan.p, n_sl + 1, st, err)) return rc;
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.sl_off), sizeof(int32_t) * (size_t)G * (nsl + 1)));
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.ell_off), sizeof(int64_t) * ((size_t)G + 1)));
    hipLaunchKernelGGL(k_slice_offsets, dim3(grid_of((int64_t)G * (nsl + 1))), dim3(256), 0, st, G, nsl, pscan.p, o.sl_off, o.ell_off, max_block.p);
    int32_t h_total = 0, h_maxb = 0;
    DP_CHK(hipMemcpyAsync(&h_total, pscan.p + n_sl, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_CHK(hipMemcpyAsync(&h_maxb, max_block.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_CHK(hipStreamSynchronize(st));
    pl.n_entries = (int64_t)h_total * 128;
    pl.max_block = (int64_t)h_maxb * 128;
    const size_t n_alloc = (size_t)pl.n_entries + 256;   // + slack: clamped loads of the last slices may run past the last block
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.ell_code), sizeof(uint16_t) * n_alloc));
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&o.ell_src), sizeof(int32_t) * n_alloc));
    DP_CHK(hipMemsetAsync(o.ell_code, 0, sizeof(uint16_t) * n_alloc, st));
    DP_CHK(hipMemsetAsync(o.ell_src, 0xff, sizeof(int32_t) * n_alloc, st));
    hipLaunchKernelGGL(k_fill_ell, dim3(grid_of((int64_t)G * S)), dim3(256), 0, st, (int64_t)G * S, (int32_t)S, nsl, o.slot_dof, d_rowptr, d_colidx, keep.p, wg.p,
                       slot_of.p, blocked ? (const int32_t*)nullptr : board_of.p, o.imp_off, o.imp_pos, o.sl_off, o.ell_off, sym ? 1 : 0, sym ? own.p : (const int32_t*)nullptr, o.ell_code, o.ell_src);
    DP_CHK(hipGetLastError());
    DP_CHK(hipStreamSynchronize(st));
    phase("sliced ELL");
    guard.armed = false;
    *out = o;
    return FDAPDE_OK;
}

void dev_persist_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_keep_flags));
    (void)hipGetLastError();
}

}  // namespace fdapde_hip
