// dev_setup.hip -- see dev_setup.h.  Every stage mirrors a stage of host_build_space (host_setup.cpp) and produces the same array:
//   locality numbering   Morton keys of node / DOF / cell positions (same quantisation, same bit interleave) + stable radix sort
//   row-owner adjacency  (DOF, cell * 16 + local index) pairs, stable sort by DOF = the host's per-row sorted visit lists
//   CSR pattern          every (row, column) pair a visit contributes, sorted as 64-bit keys, duplicates dropped
//   reference pattern    the same entries keyed by the reference's DOF ids; the sort positions are the slot map
//   sliced-ELL adjacency slice widths by a max over 64 lane positions, rows of a block dealt by visit count when padding is heavy
//   block tables         (block, cell) and (block, node) pairs sorted and deduplicated; positions by binary search
// Sorts, scans and reductions are hipCUB device primitives (set-up, not the hot path); the rest is small hand-written kernels.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dev_setup.h"

namespace fdapde_hip {

namespace {

#define DS_CHK(expr)                                                          \
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

inline unsigned grid_of(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }
inline int bits_of(int64_t n) {   // bits needed for values < n
    int b = 1;
    while ((int64_t(1) << b) < n) ++b;
    return b;
}

__device__ __forceinline__ uint64_t d_spread3(uint64_t x) {   // host_setup.cpp spread3
    x &= 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ uint64_t d_spread2(uint64_t x) {   // host_setup.cpp spread2
    x &= 0x7fffffff;
    x = (x | x << 16) & 0x0000ffff0000ffffull;
    x = (x | x << 8) & 0x00ff00ff00ff00ffull;
    x = (x | x << 4) & 0x0f0f0f0f0f0f0f0full;
    x = (x | x << 2) & 0x3333333333333333ull;
    x = (x | x << 1) & 0x5555555555555555ull;
    return x;
}

// ---- locality numbering ----------------------------------------------------------------------------------------------
__global__ void k_morton_keys(int N, int64_t n, const double* pts, const double* bb /* lo[3], hi[3] */, double span, uint64_t* key,
                              int32_t* idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t q[3] = {0, 0, 0};
    for (int d = 0; d < N; ++d) {
        const double lo = bb[d], hi = bb[3 + d];
        const double w = hi > lo ? (pts[(int64_t)d * n + i] - lo) / (hi - lo) : 0.0;
        q[d] = (uint64_t)llround(fmin(1.0, fmax(0.0, w)) * span);
    }
    key[i] = N == 3 ? (d_spread3(q[0]) | d_spread3(q[1]) << 1 | d_spread3(q[2]) << 2) : (d_spread2(q[0]) | d_spread2(q[1]) << 1);
    idx[i] = (int32_t)i;
}
__global__ void k_invert(int64_t n, const int32_t* p, int32_t* inv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[p[i]] = (int32_t)i;
}
// the nodes' coordinates side by side (NP = 2 or 4 doubles per node, reference order): one 16- / 32-byte gather per vertex afterwards instead
// of N 8-byte ones from the column-major array
__global__ void k_pack_nodes(int64_t nn, int N, int NP, const double* nodes, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    for (int d = 0; d < NP; ++d) out[i * NP + d] = d < N ? nodes[(int64_t)d * nn + i] : 0.0;
}
__global__ void k_barycentres(int64_t nc, int N, int NP, int nv, const double* npack, const int32_t* cells, double* bary) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    double s[3] = {0, 0, 0};
    for (int v = 0; v < nv; ++v) {   // (per coordinate the same sum in the same order as the host builder's)
        const double* x = npack + (int64_t)cells[c * nv + v] * NP;
        for (int d = 0; d < N; ++d) s[d] += x[d];
    }
    for (int d = 0; d < N; ++d) bary[(int64_t)d * nc + c] = s[d] / nv;
}
__global__ void k_vcoords(int64_t nn, int NP, const double* npack, const int32_t* node_i2e, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    const int64_t e = node_i2e[i];
    for (int d = 0; d < NP; ++d) out[i * NP + d] = npack[e * NP + d];
}
__global__ void k_gather_u8(int64_t n, const uint8_t* src, const int32_t* idx, uint8_t* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_cell_tables(int64_t nc, int nv, int nb, const int32_t* cells, const int32_t* dofs, const int32_t* cell_i2e,
                              const int32_t* node_e2i, const int32_t* dof_e2i, int32_t* cverts, int32_t* cdofs) {
    const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= nc) return;
    const int64_t ce = cell_i2e[ci];
    if (dofs == cells && dof_e2i == node_e2i) {   // order 1: the DOF table is the cell list, one numbering
        for (int v = 0; v < nv; ++v) {
            const int32_t x = node_e2i[cells[ce * nv + v]];
            cverts[ci * nv + v] = x, cdofs[ci * nv + v] = x;
        }
        return;
    }
    for (int v = 0; v < nv; ++v) cverts[ci * nv + v] = node_e2i[cells[ce * nv + v]];
    for (int j = 0; j < nb; ++j) cdofs[ci * nb + j] = dof_e2i[dofs[ce * nb + j]];
}

// ---- row-owner adjacency ---------------------------------------------------------------------------------------------
__global__ void k_visit_pairs(int64_t n_vis, int nb, const int32_t* cdofs, int32_t* key, int32_t* val) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_vis) return;
    const int64_t c = k / nb;
    const int j = (int)(k - c * nb);
    key[k] = cdofs[k];
    val[k] = (int32_t)(c * 16 + j);
}
// start of every group of equal keys in a sorted list: start[key] = first position (groups that do not occur keep what start[] held)
__global__ void k_group_starts(int64_t n, const int32_t* key, int32_t* start) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (i == 0 || key[i] != key[i - 1])) start[key[i]] = (int32_t)i;
}

// ---- CSR pattern -----------------------------------------------------------------------------------------------------
// the columns a visit contributes to its row, visits in (row, cell) order: the candidates of row r are positions [vptr[r] nb, vptr[r + 1] nb)
__global__ void k_pattern_cols(int64_t n_vis, int nb, const int32_t* vis, const int32_t* cdofs, int32_t* col) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_vis) return;
    const int32_t* cd = cdofs + (int64_t)(vis[k] >> 4) * nb;
    for (int j = 0; j < nb; ++j) col[k * nb + j] = cd[j];
}
__global__ void k_scaled_offsets(int64_t n, const int32_t* off, int mul, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = off[i] * mul;
}
// candidates sorted inside every row's segment: an entry is kept where it differs from its predecessor or opens its row's segment
__global__ void k_unique_in_rows(int64_t n, int nb, const int32_t* col_sorted, const int32_t* row_of_visit, const int32_t* vptr, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = row_of_visit[i / nb];
    flag[i] = (i == (int64_t)vptr[r] * nb || col_sorted[i] != col_sorted[i - 1]) ? 1 : 0;
}
__global__ void k_compact_rows(int64_t n, int nb, const int32_t* col_sorted, const int32_t* row_of_visit, const int32_t* vptr, const int32_t* flag,
                               const int32_t* pos, int32_t* colidx, int32_t* rowptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    colidx[pos[i]] = col_sorted[i];
    const int32_t r = row_of_visit[i / nb];
    if (i == (int64_t)vptr[r] * nb) rowptr[r] = pos[i];
}
__global__ void k_unique_flags(int64_t n, const uint64_t* key, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}
// unique keys (hi = group id, lo = member id) -> members list + group offsets (every group non-empty)
template <typename OffT>
__global__ void k_compact_groups(int64_t n, const uint64_t* key, const int32_t* flag, const int32_t* pos, int32_t* member, OffT* group_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    member[pos[i]] = (int32_t)(key[i] & 0xffffffffu);
    if (i == 0 || (key[i] >> 32) != (key[i - 1] >> 32)) group_off[key[i] >> 32] = (OffT)pos[i];
}
__global__ void k_diag(int64_t nd, const int32_t* rowptr, const int32_t* colidx, int32_t* diag) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nd) return;
    int32_t lo = rowptr[r], hi = rowptr[r + 1];
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (colidx[mid] < (int32_t)r) lo = mid + 1; else hi = mid;
    }
    diag[r] = lo;
}

// ---- reference-numbering pattern ---------------------------------------------------------------------------------------
// the entries of internal row r, as (reference column, internal slot) pairs at the place of reference row dof_i2e[r] (sorted per row afterwards)
__global__ void k_ref_cols(int64_t nd, const int32_t* rowptr, const int32_t* colidx, const int32_t* dof_i2e, const int32_t* rowptr_e, int32_t* key,
                           int32_t* val) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nd) return;
    int32_t o = rowptr_e[dof_i2e[r]];
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k, ++o) key[o] = dof_i2e[colidx[k]], val[o] = k;
}
__global__ void k_ref_lengths(int64_t nd, const int32_t* rowptr, const int32_t* dof_e2i, int32_t* len_e) {
    const int64_t re = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (re >= nd) return;
    const int64_t ri = dof_e2i[re];
    len_e[re] = rowptr[ri + 1] - rowptr[ri];
}
__global__ void k_ref_slots(int64_t nnz, const int32_t* val_sorted, int32_t* slot_i2e) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnz) slot_i2e[val_sorted[i]] = (int32_t)i;
}

// ---- sliced-ELL adjacency --------------------------------------------------------------------------------------------
__global__ void k_slice_widths(int64_t n_slices, int64_t nd, const int32_t* vptr, const int32_t* lane_row, int64_t* width) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slices) return;
    int32_t w = 0;
    for (int64_t q = s * kSlice; q < (s + 1) * kSlice; ++q) {
        const int64_t r = lane_row ? (int64_t)lane_row[q] : q;
        if (r >= 0 && r < nd) w = max(w, vptr[r + 1] - vptr[r]);
    }
    width[s] = w;
}
// rows of a block to its lane positions in descending order of visit count, stable (host: std::stable_sort per block)
__global__ __launch_bounds__(kAsmBlock) void k_deal_rows(int64_t nd, const int32_t* vptr, int32_t* lane_row, int32_t* row_pos) {
    __shared__ int32_t cnt[kAsmBlock];
    const int64_t r0 = (int64_t)blockIdx.x * kAsmBlock, r = r0 + threadIdx.x;
    const int32_t mine = r < nd ? vptr[r + 1] - vptr[r] : -1;
    cnt[threadIdx.x] = mine;
    lane_row[r0 + threadIdx.x] = -1;
    __syncthreads();
    if (r >= nd) return;
    int rank = 0;
    for (int u = 0; u < kAsmBlock; ++u) {
        const int32_t o = cnt[u];
        rank += (o > mine) || (o == mine && u < (int)threadIdx.x);
    }
    __syncthreads();
    lane_row[r0 + rank] = (int32_t)r;
    row_pos[r] = (int32_t)(r0 + rank);
}
__device__ __forceinline__ int32_t lower_bound_i32(const int32_t* a, int32_t lo, int32_t hi, int32_t x) {
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int64_t lower_bound_i64(const int32_t* a, int64_t lo, int64_t hi, int32_t x) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// one workgroup per slice: lane = threadIdx.x, visits v = threadIdx.y, + blockDim.y, ...  The column lists of the slice's 64 rows and the cell list
// of its assembly block are staged in LDS once: a visit looks its cell up in the block's list and each of its nb DOFs in its row's columns -- 10 + nb 6
// dependent steps per visit, from LDS instead of as chains of global loads (C5: 17 ms of a 67 ms dofs_build before); the slot codes of a visit leave as
// whole 32-bit words.
__device__ __forceinline__ int32_t lower_bound_lds(const int32_t* a, int32_t n, int32_t x) {
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ void k_fill_adjacency(int64_t nd, int nb, int nbw, int max_row, const int64_t* sl_off, const int32_t* lane_row, const int32_t* vptr,
                                 const int32_t* vis, const int32_t* cdofs, const int32_t* rowptr, const int32_t* colidx, const int64_t* bc_off,
                                 const int32_t* bc_cell, int32_t* adj, uint32_t* slotw) {
    extern __shared__ int32_t fa_lds[];   // [kSlice][max_row] columns of the slice's rows, then the block's cell list
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t off = sl_off[s], width = sl_off[s + 1] - off;
    const int64_t q = s * kSlice + lane;
    const int64_t rr = lane_row ? (int64_t)lane_row[q] : q;
    const int64_t r = (rr < 0 || rr >= nd) ? nd : rr;
    const int32_t len = r < nd ? vptr[r + 1] - vptr[r] : 0;
    const int32_t k0 = r < nd ? rowptr[r] : 0, n_cols = r < nd ? rowptr[r + 1] - k0 : 0;
    const int64_t b = (s * kSlice) / kAsmBlock;   // the slice's assembly block (its rows are dealt inside the block)
    int32_t* cols = fa_lds + lane * max_row;
    for (int32_t k = threadIdx.y; k < n_cols; k += blockDim.y) cols[k] = colidx[k0 + k];
    int32_t* cells_l = fa_lds + kSlice * max_row;
    const int64_t c0 = bc_off[b];
    const int32_t n_bcl = (int32_t)(bc_off[b + 1] - c0);
    for (int32_t i = threadIdx.y * kSlice + lane; i < n_bcl; i += kSlice * blockDim.y) cells_l[i] = bc_cell[c0 + i];
    __syncthreads();
    for (int64_t v = threadIdx.y; v < width; v += blockDim.y) {
        const int64_t at = (off + v) * kSlice + lane;
        uint32_t* sw = slotw + at * nbw;
        if (v < len) {
            const int32_t visit = vis[vptr[r] + v];
            const int32_t cell = visit >> 4;
            adj[at] = lower_bound_lds(cells_l, n_bcl, cell) * 16 + (visit & 15);
            const int32_t* cd = cdofs + (int64_t)cell * nb;
            for (int w = 0; w < nbw; ++w) {
                const int j0 = 2 * w, j1 = 2 * w + 1;
                const uint32_t lo = j0 < nb ? (uint32_t)lower_bound_lds(cols, n_cols, cd[j0]) & 0xffffu : 0u;
                const uint32_t hi = j1 < nb ? (uint32_t)lower_bound_lds(cols, n_cols, cd[j1]) & 0xffffu : 0u;
                sw[w] = lo | (hi << 16);
            }
        } else {
            adj[at] = -1;
            for (int w = 0; w < nbw; ++w) sw[w] = 0;
        }
    }
}

// the same table with the searches in global memory: rows too long (or blocks too large) for the LDS staging -- a fan of 700 cells around one vertex
__global__ void k_fill_adjacency_global(int64_t nd, int nb, int nbw, const int64_t* sl_off, const int32_t* lane_row, const int32_t* vptr,
                                        const int32_t* vis, const int32_t* cdofs, const int32_t* rowptr, const int32_t* colidx, const int64_t* bc_off,
                                        const int32_t* bc_cell, int32_t* adj, uint32_t* slotw) {
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t off = sl_off[s], width = sl_off[s + 1] - off;
    const int64_t q = s * kSlice + lane;
    const int64_t rr = lane_row ? (int64_t)lane_row[q] : q;
    const int64_t r = (rr < 0 || rr >= nd) ? nd : rr;
    const int32_t len = r < nd ? vptr[r + 1] - vptr[r] : 0;
    const int32_t k0 = r < nd ? rowptr[r] : 0, k1 = r < nd ? rowptr[r + 1] : 0;
    const int64_t b = (s * kSlice) / kAsmBlock;
    for (int64_t v = threadIdx.y; v < width; v += blockDim.y) {
        const int64_t at = (off + v) * kSlice + lane;
        uint16_t* sw = reinterpret_cast<uint16_t*>(slotw + at * nbw);
        if (v < len) {
            const int32_t visit = vis[vptr[r] + v];
            const int32_t cell = visit >> 4;
            adj[at] = (int32_t)(lower_bound_i64(bc_cell, bc_off[b], bc_off[b + 1], cell) - bc_off[b]) * 16 + (visit & 15);
            const int32_t* cd = cdofs + (int64_t)cell * nb;
            for (int j = 0; j < nb; ++j) sw[j] = (uint16_t)(lower_bound_i32(colidx, k0, k1, cd[j]) - k0);
            for (int j = nb; j < 2 * nbw; ++j) sw[j] = 0;
        } else {
            adj[at] = -1;
            for (int j = 0; j < 2 * nbw; ++j) sw[j] = 0;
        }
    }
}

// ---- block tables ----------------------------------------------------------------------------------------------------
// visits are in row order, so the visits of assembly block b are positions [vptr[b kAsmBlock], vptr[(b + 1) kAsmBlock)): segment offsets
__global__ void k_block_visit_offsets(int64_t n_blk, int64_t nd, const int32_t* vptr, int32_t* off) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= n_blk) off[b] = vptr[min(nd, b * kAsmBlock)];
}
__global__ void k_visit_cells(int64_t n_vis, const int32_t* vis, int32_t* cell) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_vis) cell[k] = vis[k] >> 4;
}
// ids sorted inside the segments of the blocks: kept where an id differs from its predecessor or opens its block's segment.
// blk_of_group[i / group] = block of position i (group = 1: per visit, by its row; group = nv: per block-cell)
__global__ void k_unique_in_blocks(int64_t n, int group, int div, const int32_t* id_sorted, const int32_t* blk_src, const int32_t* seg_off, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t b = blk_src[i / group] / div;
    flag[i] = (i == (int64_t)seg_off[b] || id_sorted[i] != id_sorted[i - 1]) ? 1 : 0;
}
__global__ void k_compact_blocks(int64_t n, int group, int div, const int32_t* id_sorted, const int32_t* blk_src, const int32_t* seg_off, const int32_t* flag,
                                 const int32_t* pos, int32_t* member, int32_t* member_blk, int64_t* group_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const int32_t b = blk_src[i / group] / div;
    member[pos[i]] = id_sorted[i];
    if (member_blk) member_blk[pos[i]] = b;
    if (i == (int64_t)seg_off[b]) group_off[b] = (int64_t)pos[i];
}
__global__ void k_block_cell_nodes(int64_t n_bc, int nv, const int32_t* bc_cell, const int32_t* cverts, int32_t* node) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bc) return;
    const int64_t cell = bc_cell[i];
    for (int v = 0; v < nv; ++v) node[i * nv + v] = cverts[cell * nv + v];
}
__global__ void k_scaled_offsets64(int64_t n, const int64_t* off, int mul, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)(off[i] * mul);
}
__global__ void k_block_verts(int64_t n_bc, int nv, const int32_t* bc_blk, const int32_t* bc_cell, const int32_t* cverts, const int64_t* bn_off,
                              const int32_t* bn_node, uint16_t* bc_vert) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bc) return;
    const int64_t b = bc_blk[i], cell = bc_cell[i];
    for (int v = 0; v < 4; ++v)
        bc_vert[i * 4 + v] = v < nv ? (uint16_t)(lower_bound_i64(bn_node, bn_off[b], bn_off[b + 1], cverts[cell * nv + v]) - bn_off[b]) : (uint16_t)0;
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
template <typename T> __global__ void k_adjacent_diff_max(int64_t n, const T* off, int32_t* out_max) {   // (grid-stride; one atomic per wave)
    int32_t v = INT32_MIN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v = max(v, (int32_t)(off[i + 1] - off[i]));
    v = wave_max_i32(v);
    if ((threadIdx.x & 63) == 0 && v != INT32_MIN) atomicMax(out_max, v);
}
__global__ void k_block_nnz_max(int64_t n_blk, int64_t nd, const int32_t* rowptr, int32_t* out_max) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    const int64_t r1 = min(nd, (b + 1) * kAsmBlock);
    atomicMax(out_max, rowptr[r1] - rowptr[b * kAsmBlock]);
}
__global__ void k_min_i32(int64_t n, const int32_t* v, int32_t* out_min) {   // (grid-stride; one atomic per wave)
    int32_t m = INT32_MIN + 1;   // (of the negated values)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = max(m, -v[i]);
    m = -wave_max_i32(m);
    if ((threadIdx.x & 63) == 0) atomicMin(out_min, m);
}

struct Scratch {   // one growing scratch allocation for the device primitives
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

template <typename K, typename V>
int sort_pairs(Scratch& sc, K* k_in, K* k_out, V* v_in, V* v_out, int64_t n, int end_bit, hipStream_t st, std::string& err) {
    size_t need = 0;
    DS_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st));
    DS_CHK(sc.need(need));
    DS_CHK(hipcub::DeviceRadixSort::SortPairs(sc.p, need, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st));
    return FDAPDE_OK;
}
template <typename K> int sort_keys(Scratch& sc, K* k_in, K* k_out, int64_t n, int end_bit, hipStream_t st, std::string& err) {
    size_t need = 0;
    DS_CHK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, k_in, k_out, (int)n, 0, end_bit, st));
    DS_CHK(sc.need(need));
    DS_CHK(hipcub::DeviceRadixSort::SortKeys(sc.p, need, k_in, k_out, (int)n, 0, end_bit, st));
    return FDAPDE_OK;
}
// keys (with values) sorted inside the segments [off[s], off[s + 1]) -- rows of a matrix, blocks of rows: 32-bit keys and one pass over
// short segments where the same order by a global sort of (segment, key) pairs takes 64-bit keys and 7 passes
template <typename K> int seg_sort_keys(Scratch& sc, const K* k_in, K* k_out, int64_t n, int64_t n_seg, const int32_t* off, int end_bit, hipStream_t st,
                                        std::string& err) {
    size_t need = 0;
    DS_CHK(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, need, k_in, k_out, (int)n, (int)n_seg, off, off + 1, 0, end_bit, st));
    DS_CHK(sc.need(need));
    DS_CHK(hipcub::DeviceSegmentedRadixSort::SortKeys(sc.p, need, k_in, k_out, (int)n, (int)n_seg, off, off + 1, 0, end_bit, st));
    return FDAPDE_OK;
}
template <typename K, typename V>
int seg_sort_pairs(Scratch& sc, const K* k_in, K* k_out, const V* v_in, V* v_out, int64_t n, int64_t n_seg, const int32_t* off, int end_bit,
                   hipStream_t st, std::string& err) {
    size_t need = 0;
    DS_CHK(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, need, k_in, k_out, v_in, v_out, (int)n, (int)n_seg, off, off + 1, 0, end_bit, st));
    DS_CHK(sc.need(need));
    DS_CHK(hipcub::DeviceSegmentedRadixSort::SortPairs(sc.p, need, k_in, k_out, v_in, v_out, (int)n, (int)n_seg, off, off + 1, 0, end_bit, st));
    return FDAPDE_OK;
}
template <typename In, typename Out> int exclusive_sum(Scratch& sc, In* in, Out* out, int64_t n, hipStream_t st, std::string& err) {
    size_t need = 0;
    DS_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, (int)n, st));
    DS_CHK(sc.need(need));
    DS_CHK(hipcub::DeviceScan::ExclusiveSum(sc.p, need, in, out, (int)n, st));
    return FDAPDE_OK;
}

// ---- bin grid for point location (dev_build_bin_grid) ----------------------------------------------------------------------------------
template <int M> struct BinGeo {
    double lo[M], inv_h[M];
    int32_t dims[M];
};
// bins [b0, b1] a cell's bounding box overlaps along axis d (the expressions of the former host loop, operation for operation)
template <int M>
__device__ __forceinline__ void bin_range(const BinGeo<M>& G, const double* vcoords, const int32_t* cv, int d, int& b0, int& b1) {
    constexpr int NP = M == 2 ? 2 : 4;
    double mn = 1e300, mx = -1e300;
#pragma unroll
    for (int v = 0; v <= M; ++v) {
        const double x = vcoords[(size_t)cv[v] * NP + d];
        mn = x < mn ? x : mn, mx = x > mx ? x : mx;
    }
    b0 = (int)floor(__dadd_rn(__dmul_rn(mn - G.lo[d], G.inv_h[d]), -1e-9)), b1 = (int)floor(__dadd_rn(__dmul_rn(mx - G.lo[d], G.inv_h[d]), 1e-9));
    b0 = b0 < 0 ? 0 : b0, b1 = b1 >= G.dims[d] ? G.dims[d] - 1 : b1;
}
// FILL = false: cnt[bin] += 1 per (cell, bin) overlap; FILL = true: the cell into the bin's list at the next free position
template <int M, bool FILL>
__global__ __launch_bounds__(256) void k_bin_cells(BinGeo<M> G, int64_t n_cells, const double* vcoords, const int32_t* cverts, int32_t* cnt_or_cursor,
                                                   int32_t* bin_cells) {
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    const int32_t* cv = cverts + cell * (M + 1);
    int b0[3] = {0, 0, 0}, b1[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < M; ++d) bin_range<M>(G, vcoords, cv, d, b0[d], b1[d]);
    for (int z = b0[2]; z <= b1[2]; ++z)
        for (int y = b0[1]; y <= b1[1]; ++y)
            for (int x = b0[0]; x <= b1[0]; ++x) {
                const int64_t bin = M == 2 ? (int64_t)y * G.dims[0] + x : ((int64_t)z * G.dims[1] + y) * G.dims[0] + x;
                const int32_t at = atomicAdd(cnt_or_cursor + bin, 1);
                if constexpr (FILL) bin_cells[at] = (int32_t)cell;
            }
}
// the cells of every bin in ascending id (the fill pass leaves them in arrival order); lists are a handful of cells long
__global__ __launch_bounds__(256) void k_bin_sort(int64_t n_bins, const int32_t* bin_ptr, int32_t* bin_cells) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    const int32_t lo = bin_ptr[b], hi = bin_ptr[b + 1];
    for (int32_t i = lo + 1; i < hi; ++i) {
        const int32_t v = bin_cells[i];
        int32_t j = i - 1;
        while (j >= lo && bin_cells[j] > v) bin_cells[j + 1] = bin_cells[j], --j;
        bin_cells[j + 1] = v;
    }
}
// per workgroup: min / max of every coordinate over a strided share of the nodes -> out[6 b + d], out[6 b + 3 + d]
__global__ __launch_bounds__(256) void k_bbox_partials(int NP, int M, int64_t n_nodes, const double* vcoords, double* out) {
    __shared__ double red[2][3][4];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += (int64_t)gridDim.x * blockDim.x)
        for (int d = 0; d < M; ++d) {
            const double v = vcoords[(size_t)i * NP + d];
            mn[d] = v < mn[d] ? v : mn[d], mx[d] = v > mx[d] ? v : mx[d];
        }
    for (int d = 0; d < 3; ++d)
        for (int o = 32; o > 0; o >>= 1) mn[d] = fmin(mn[d], __shfl_xor(mn[d], o)), mx[d] = fmax(mx[d], __shfl_xor(mx[d], o));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int d = 0; d < 3; ++d) red[0][d][wave] = mn[d], red[1][d][wave] = mx[d];
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        double a = red[0][d][0], b = red[1][d][0];
        for (int w = 1; w < 4; ++w) a = fmin(a, red[0][d][w]), b = fmax(b, red[1][d][w]);
        out[6 * blockIdx.x + d] = a, out[6 * blockIdx.x + 3 + d] = b;
    }
}

// Morton order of n points (column-major n x N on the device): i2e (new -> old), stable for equal keys
int morton_order(Scratch& sc, int N, int64_t n, const double* d_pts, int bits, hipStream_t st, int32_t* d_i2e, std::string& err) {
    Tmp<uint64_t> key_a, key_b;
    Tmp<int32_t> idx_a;
    Tmp<double> bb;
    DS_CHK(key_a.alloc((size_t)n));
    DS_CHK(key_b.alloc((size_t)n));
    DS_CHK(idx_a.alloc((size_t)n));
    DS_CHK(bb.alloc(6));
    for (int d = 0; d < N; ++d) {
        size_t need = 0;
        DS_CHK(hipcub::DeviceReduce::Min(nullptr, need, d_pts + (int64_t)d * n, bb.p + d, (int)n, st));
        DS_CHK(sc.need(need));
        DS_CHK(hipcub::DeviceReduce::Min(sc.p, need, d_pts + (int64_t)d * n, bb.p + d, (int)n, st));
        DS_CHK(hipcub::DeviceReduce::Max(sc.p, need, d_pts + (int64_t)d * n, bb.p + 3 + d, (int)n, st));
    }
    const int max_bits = N == 3 ? 21 : 31;
    const int b = bits > 0 && bits < max_bits ? bits : max_bits;
    const double span = (double)((uint64_t(1) << b) - 1);
    hipLaunchKernelGGL(k_morton_keys, dim3(grid_of(n)), dim3(256), 0, st, N, n, d_pts, bb.p, span, key_a.p, idx_a.p);
    if (int rc = sort_pairs(sc, key_a.p, key_b.p, idx_a.p, d_i2e, n, N * b, st, err)) return rc;
    if (!t_arena) DS_CHK(hipStreamSynchronize(st));   // temporaries that own their memory are freed on return (an arena's live to the end of the build)
    return FDAPDE_OK;
}

}  // namespace

int dev_build_bin_grid(int M, int64_t n_nodes, int64_t n_cells, const double* d_vcoords, const int32_t* d_cverts, void* stream, DevBinGrid* out,
                       std::string& err) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!out || (M != 2 && M != 3) || n_nodes < 1 || n_cells < 1) return FDAPDE_EINVAL;
    const int NP = M == 2 ? 2 : 4;
    Scratch sc;
    // bounding box
    const int nblk = (int)std::min<int64_t>(256, (n_nodes + 255) / 256);
    Tmp<double> d_part;
    DS_CHK(d_part.alloc(6 * (size_t)nblk));
    hipLaunchKernelGGL(k_bbox_partials, dim3(nblk), dim3(256), 0, st, NP, M, n_nodes, d_vcoords, d_part.p);
    std::vector<double> part(6 * (size_t)nblk);
    DS_CHK(hipMemcpyAsync(part.data(), d_part.p, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
    DS_CHK(hipStreamSynchronize(st));
    DevBinGrid g;
    double hi[3] = {0, 0, 0};
    for (int d = 0; d < M; ++d) {
        g.lo[d] = part[(size_t)d], hi[d] = part[3 + (size_t)d];
        for (int b = 1; b < nblk; ++b) g.lo[d] = std::min(g.lo[d], part[6 * (size_t)b + d]), hi[d] = std::max(hi[d], part[6 * (size_t)b + 3 + d]);
    }
    // about one cell per bin on average
    const int gd = (int)std::max(1.0, std::floor(std::pow((double)n_cells, 1.0 / M)));
    g.n_bins = 1;
    for (int d = 0; d < M; ++d) {
        g.dims[d] = gd, g.n_bins *= gd;
        g.inv_h[d] = hi[d] > g.lo[d] ? gd / (hi[d] - g.lo[d]) : 0.0;
    }
    if (g.n_bins >= (int64_t(1) << 31) - 2) return FDAPDE_EUNSUPPORTED;
    Tmp<int32_t> cnt;
    DS_CHK(cnt.alloc((size_t)g.n_bins + 1));
    DS_CHK(hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * ((size_t)g.n_bins + 1), st));
    DS_CHK(hipMalloc(reinterpret_cast<void**>(&g.bin_ptr), sizeof(int32_t) * ((size_t)g.n_bins + 1)));
    auto fail_free = [&](int rc) {
        if (g.bin_ptr) (void)hipFree(g.bin_ptr);
        if (g.bin_cells) (void)hipFree(g.bin_cells);
        return rc;
    };
#define BIN_GO(MM, FILL_, CNT_, CELLS_)                                                                                        \
    do {                                                                                                                       \
        BinGeo<MM> G;                                                                                                          \
        for (int d = 0; d < MM; ++d) G.lo[d] = g.lo[d], G.inv_h[d] = g.inv_h[d], G.dims[d] = g.dims[d];                        \
        hipLaunchKernelGGL((k_bin_cells<MM, FILL_>), dim3(grid_of(n_cells)), dim3(256), 0, st, G, n_cells, d_vcoords, d_cverts, CNT_, CELLS_); \
    } while (0)
    if (M == 2) BIN_GO(2, false, cnt.p, (int32_t*)nullptr);
    else BIN_GO(3, false, cnt.p, (int32_t*)nullptr);
    if (int rc = exclusive_sum(sc, cnt.p, g.bin_ptr, g.n_bins + 1, st, err)) return fail_free(rc);
    int32_t total = 0;
    if (hipMemcpyAsync(&total, g.bin_ptr + g.n_bins, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        err = "bin grid: read-back of the overlap count failed";
        return fail_free(FDAPDE_EHIP);
    }
    g.n_entries = total;
    if (hipMalloc(reinterpret_cast<void**>(&g.bin_cells), sizeof(int32_t) * ((size_t)total + 1)) != hipSuccess) {
        err = "bin grid: allocation of the bin lists failed";
        return fail_free(FDAPDE_EHIP);
    }
    // cursors = the offsets; the fill pass advances them
    if (hipMemcpyAsync(cnt.p, g.bin_ptr, sizeof(int32_t) * ((size_t)g.n_bins + 1), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail_free(FDAPDE_EHIP);
    if (M == 2) BIN_GO(2, true, cnt.p, g.bin_cells);
    else BIN_GO(3, true, cnt.p, g.bin_cells);
#undef BIN_GO
    hipLaunchKernelGGL(k_bin_sort, dim3(grid_of(g.n_bins)), dim3(256), 0, st, g.n_bins, g.bin_ptr, g.bin_cells);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {   // (the temporaries of this scope are freed on return)
        err = "bin grid: kernels failed";
        return fail_free(FDAPDE_EHIP);
    }
    *out = g;
    return FDAPDE_OK;
}

void dev_space_release(DevSpace* s) {
    if (!s) return;
    for (void* p : {(void*)s->cverts, (void*)s->cdofs, (void*)s->adj, (void*)s->rowptr, (void*)s->colidx, (void*)s->diag, (void*)s->slot_i2e,
                    (void*)s->dof_i2e, (void*)s->dof_e2i, (void*)s->cell_i2e, (void*)s->node_i2e, (void*)s->bc_cell, (void*)s->bn_node,
                    (void*)s->lane_row, (void*)s->rowptr_e, (void*)s->colidx_e, (void*)s->slotw, (void*)s->bc_vert, (void*)s->sl_off,
                    (void*)s->bc_off, (void*)s->bn_off, (void*)s->vcoords, (void*)s->bnd})
        if (p) (void)hipFree(p);
    *s = DevSpace{};
}

int dev_build_space(HostSpace& hs, const double* d_nodes, const int32_t* d_cells, const int32_t* d_dofs, const uint8_t* d_dof_bnd,
                    const double* d_dof_coords, void* stream, DevSpace* out, std::string& err) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int M = hs.M, N = hs.N, nv = M + 1, nb = hs.nb, order = hs.order;
    const int64_t nc = hs.n_cells, nn = hs.n_nodes, nd = hs.n_dofs;
    const int64_t n_vis = nc * nb;
    if (n_vis * nb > INT32_MAX) {
        err = "too many (row, column) contributions for the device set-up (cells x nb^2 exceeds int32)";
        return FDAPDE_EUNSUPPORTED;
    }
    const bool dbg = std::getenv("FDAPDE_DEBUG_SETUP") != nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (dbg) (void)hipEventCreate(&ev0), (void)hipEventCreate(&ev1), (void)hipEventRecord(ev0, st);
    auto phase = [&](const char* name) {
        if (!dbg) return;
        (void)hipEventRecord(ev1, st), (void)hipEventSynchronize(ev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev0, ev1);
        std::fprintf(stderr, "device setup %-28s %8.2f ms\n", name, ms);
        (void)hipEventRecord(ev0, st);
    };
    Arena arena;   // (declared before everything that allocates from it: released last)
    ArenaScope arena_scope(&arena);
    Scratch sc;
    DevSpace s;
    struct Guard {   // frees whatever has been allocated if the build fails half-way
        DevSpace* s;
        bool armed = true;
        ~Guard() {
            if (armed) dev_space_release(s);
        }
    } guard{&s};
#define DS_ALLOC(ptr, T, count) DS_CHK(hipMalloc(reinterpret_cast<void**>(&(ptr)), sizeof(T) * (size_t)((count) > 0 ? (count) : 1)))

    // ---- locality numbering (host_setup.cpp: "locality numbering")
    Tmp<int32_t> node_e2i;
    DS_ALLOC(s.node_i2e, int32_t, nn);
    DS_CHK(node_e2i.alloc((size_t)nn));
    if (int rc = morton_order(sc, N, nn, d_nodes, 0, st, s.node_i2e, err)) return rc;
    hipLaunchKernelGGL(k_invert, dim3(grid_of(nn)), dim3(256), 0, st, nn, s.node_i2e, node_e2i.p);
    DS_ALLOC(s.dof_i2e, int32_t, nd);
    DS_ALLOC(s.dof_e2i, int32_t, nd);
    if (order == 1) {
        DS_CHK(hipMemcpyAsync(s.dof_i2e, s.node_i2e, sizeof(int32_t) * (size_t)nd, hipMemcpyDeviceToDevice, st));
        DS_CHK(hipMemcpyAsync(s.dof_e2i, node_e2i.p, sizeof(int32_t) * (size_t)nd, hipMemcpyDeviceToDevice, st));
    } else {
        if (int rc = morton_order(sc, N, nd, d_dof_coords, 0, st, s.dof_i2e, err)) return rc;
        hipLaunchKernelGGL(k_invert, dim3(grid_of(nd)), dim3(256), 0, st, nd, s.dof_i2e, s.dof_e2i);
    }
    DS_ALLOC(s.cell_i2e, int32_t, nc);
    const int NP = N == 2 ? 2 : 4;
    Tmp<double> npack;
    DS_CHK(npack.alloc((size_t)nn * NP));
    hipLaunchKernelGGL(k_pack_nodes, dim3(grid_of(nn)), dim3(256), 0, st, nn, N, NP, d_nodes, npack.p);
    {
        Tmp<double> bary;
        DS_CHK(bary.alloc((size_t)nc * N));
        hipLaunchKernelGGL(k_barycentres, dim3(grid_of(nc)), dim3(256), 0, st, nc, N, NP, nv, npack.p, d_cells, bary.p);
        if (int rc = morton_order(sc, N, nc, bary.p, order == 1 ? (N == 3 ? 11 : 16) : 0, st, s.cell_i2e, err)) return rc;
    }
    DS_ALLOC(s.vcoords, double, nn * NP);
    DS_ALLOC(s.bnd, uint8_t, nd);
    DS_ALLOC(s.cverts, int32_t, nc * nv);
    DS_ALLOC(s.cdofs, int32_t, nc * nb);
    hipLaunchKernelGGL(k_vcoords, dim3(grid_of(nn)), dim3(256), 0, st, nn, NP, npack.p, s.node_i2e, s.vcoords);
    hipLaunchKernelGGL(k_gather_u8, dim3(grid_of(nd)), dim3(256), 0, st, nd, d_dof_bnd, s.dof_i2e, s.bnd);
    hipLaunchKernelGGL(k_cell_tables, dim3(grid_of(nc)), dim3(256), 0, st, nc, nv, nb, d_cells, d_dofs, s.cell_i2e, node_e2i.p,
                       order == 1 ? (const int32_t*)node_e2i.p : (const int32_t*)s.dof_e2i, s.cverts, s.cdofs);
    phase("locality numbering");

    // ---- row-owner adjacency: visits of a DOF = (cell * 16 + local index), cells ascending (stable sort by DOF)
    Tmp<int32_t> vkey_a, vkey, vval_a, vis, vptr;
    DS_CHK(vkey_a.alloc((size_t)n_vis));
    DS_CHK(vkey.alloc((size_t)n_vis));
    DS_CHK(vval_a.alloc((size_t)n_vis));
    DS_CHK(vis.alloc((size_t)n_vis));
    DS_CHK(vptr.alloc((size_t)nd + 1));
    DS_CHK(hipMemsetAsync(vptr.p, 0xff, sizeof(int32_t) * (size_t)nd, st));   // (-1: a row without a visit stays so)
    const int32_t h_nvis = (int32_t)n_vis;
    DS_CHK(hipMemcpyAsync(vptr.p + nd, &h_nvis, sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_visit_pairs, dim3(grid_of(n_vis)), dim3(256), 0, st, n_vis, nb, s.cdofs, vkey_a.p, vval_a.p);
    if (int rc = sort_pairs(sc, vkey_a.p, vkey.p, vval_a.p, vis.p, n_vis, bits_of(nd), st, err)) return rc;
    hipLaunchKernelGGL(k_group_starts, dim3(grid_of(n_vis)), dim3(256), 0, st, n_vis, vkey.p, vptr.p);   // the rows' visit lists start where the sorted keys change
    {   // a node no cell references has an empty matrix row: the reference's LU fails on such a mesh
        Tmp<int32_t> mn;
        DS_CHK(mn.alloc(1));
        const int32_t big = INT32_MAX;
        DS_CHK(hipMemcpyAsync(mn.p, &big, sizeof big, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_min_i32, dim3(std::min(grid_of(nd), 512u)), dim3(256), 0, st, nd, vptr.p, mn.p);
        int32_t h = 0;
        DS_CHK(hipMemcpyAsync(&h, mn.p, sizeof h, hipMemcpyDeviceToHost, st));
        DS_CHK(hipStreamSynchronize(st));
        if (h < 0) {
            err = "a node is not referenced by any cell: its DOF has an empty matrix row (the reference's LU fails on such a mesh)";
            return FDAPDE_EINVAL;
        }
    }
    vkey_a.reset(), vval_a.reset();
    phase("row-owner adjacency");

    // ---- internal CSR pattern: sorted union of the DOFs of the visiting cells
    int64_t nnz = 0;
    {
        // the candidates of a row (the DOFs of its visiting cells) lie together -- the visits are in row order: sorted inside every row's
        // segment (32-bit keys, short segments), duplicates dropped, compacted
        const int64_t n_pairs = n_vis * nb;
        Tmp<int32_t> pk_a, pk, seg, flag, pos;
        DS_CHK(pk_a.alloc((size_t)n_pairs));
        DS_CHK(pk.alloc((size_t)n_pairs));
        DS_CHK(seg.alloc((size_t)nd + 1));
        hipLaunchKernelGGL(k_pattern_cols, dim3(grid_of(n_vis)), dim3(256), 0, st, n_vis, nb, vis.p, s.cdofs, pk_a.p);
        hipLaunchKernelGGL(k_scaled_offsets, dim3(grid_of(nd + 1)), dim3(256), 0, st, nd + 1, vptr.p, nb, seg.p);
        if (int rc = seg_sort_keys(sc, pk_a.p, pk.p, n_pairs, nd, seg.p, bits_of(nd), st, err)) return rc;
        if (!t_arena) DS_CHK(hipStreamSynchronize(st));
        pk_a.reset();
        DS_CHK(flag.alloc((size_t)n_pairs + 1));
        DS_CHK(pos.alloc((size_t)n_pairs + 1));
        DS_CHK(hipMemsetAsync(flag.p + n_pairs, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_unique_in_rows, dim3(grid_of(n_pairs)), dim3(256), 0, st, n_pairs, nb, pk.p, vkey.p, vptr.p, flag.p);
        if (int rc = exclusive_sum(sc, flag.p, pos.p, n_pairs + 1, st, err)) return rc;
        int32_t h_nnz = 0;
        DS_CHK(hipMemcpyAsync(&h_nnz, pos.p + n_pairs, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        DS_CHK(hipStreamSynchronize(st));
        nnz = h_nnz;
        DS_ALLOC(s.rowptr, int32_t, nd + 1);
        DS_ALLOC(s.colidx, int32_t, nnz + 2);   // + 2 zeros: the SpMV's pair loads may touch one entry past a row's end
        DS_CHK(hipMemsetAsync(s.colidx + nnz, 0, 2 * sizeof(int32_t), st));
        hipLaunchKernelGGL(k_compact_rows, dim3(grid_of(n_pairs)), dim3(256), 0, st, n_pairs, nb, pk.p, vkey.p, vptr.p, flag.p, pos.p, s.colidx, s.rowptr);
        DS_CHK(hipMemcpyAsync(s.rowptr + nd, &h_nnz, sizeof(int32_t), hipMemcpyHostToDevice, st));
        DS_CHK(hipStreamSynchronize(st));
    }
    DS_ALLOC(s.diag, int32_t, nd);
    hipLaunchKernelGGL(k_diag, dim3(grid_of(nd)), dim3(256), 0, st, nd, s.rowptr, s.colidx, s.diag);
    Tmp<int32_t> maxes;   // [0] max row, [1] max block nnz, [2] max block cells, [3] max block nodes, [4] widest adjacency slice
    DS_CHK(maxes.alloc(5));
    DS_CHK(hipMemsetAsync(maxes.p, 0, 5 * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_adjacent_diff_max<int32_t>, dim3(std::min(grid_of(nd), 512u)), dim3(256), 0, st, nd, s.rowptr, maxes.p);
    phase("internal CSR pattern");

    // ---- reference-numbering pattern + internal slot -> reference slot
    DS_ALLOC(s.rowptr_e, int32_t, nd + 1);
    DS_ALLOC(s.colidx_e, int32_t, nnz);
    DS_ALLOC(s.slot_i2e, int32_t, nnz);
    {
        // reference row re = internal row dof_e2i[re]: its entries are written at the row's place in the reference pattern with their reference
        // columns and sorted there (segments = rows); the sorted positions of the internal slots are the slot map
        Tmp<int32_t> rk_a, rv_a, rv, len_e;
        DS_CHK(rk_a.alloc((size_t)nnz));
        DS_CHK(rv_a.alloc((size_t)nnz));
        DS_CHK(rv.alloc((size_t)nnz));
        DS_CHK(len_e.alloc((size_t)nd + 1));
        DS_CHK(hipMemsetAsync(len_e.p + nd, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_ref_lengths, dim3(grid_of(nd)), dim3(256), 0, st, nd, s.rowptr, s.dof_e2i, len_e.p);
        if (int rc = exclusive_sum(sc, len_e.p, s.rowptr_e, nd + 1, st, err)) return rc;
        hipLaunchKernelGGL(k_ref_cols, dim3(grid_of(nd)), dim3(256), 0, st, nd, s.rowptr, s.colidx, s.dof_i2e, s.rowptr_e, rk_a.p, rv_a.p);
        if (int rc = seg_sort_pairs(sc, rk_a.p, s.colidx_e, rv_a.p, rv.p, nnz, nd, s.rowptr_e, bits_of(nd), st, err)) return rc;
        hipLaunchKernelGGL(k_ref_slots, dim3(grid_of(nnz)), dim3(256), 0, st, nnz, rv.p, s.slot_i2e);
        DS_CHK(hipStreamSynchronize(st));
    }
    phase("reference pattern + slot map");

    // ---- block tables: cells visited by the rows of an assembly block, and their vertices (both ascending)
    const int64_t n_blk = (nd + kAsmBlock - 1) / kAsmBlock, n_slices = (nd + kSlice - 1) / kSlice;
    s.n_blk = n_blk, s.n_slices = n_slices;
    Tmp<int32_t> bc_blk;   // block of every block-cell
    {
        Tmp<int32_t> k_a, k_s, seg, flag, pos;
        DS_CHK(k_a.alloc((size_t)n_vis));
        DS_CHK(k_s.alloc((size_t)n_vis));
        DS_CHK(seg.alloc((size_t)n_blk + 1));
        DS_CHK(flag.alloc((size_t)n_vis + 1));
        DS_CHK(pos.alloc((size_t)n_vis + 1));
        hipLaunchKernelGGL(k_visit_cells, dim3(grid_of(n_vis)), dim3(256), 0, st, n_vis, vis.p, k_a.p);
        hipLaunchKernelGGL(k_block_visit_offsets, dim3(grid_of(n_blk + 1)), dim3(256), 0, st, n_blk, nd, vptr.p, seg.p);
        if (int rc = seg_sort_keys(sc, k_a.p, k_s.p, n_vis, n_blk, seg.p, bits_of(nc), st, err)) return rc;
        DS_CHK(hipMemsetAsync(flag.p + n_vis, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_unique_in_blocks, dim3(grid_of(n_vis)), dim3(256), 0, st, n_vis, 1, kAsmBlock, k_s.p, vkey.p, seg.p, flag.p);
        if (int rc = exclusive_sum(sc, flag.p, pos.p, n_vis + 1, st, err)) return rc;
        int32_t h_n = 0;
        DS_CHK(hipMemcpyAsync(&h_n, pos.p + n_vis, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        DS_CHK(hipStreamSynchronize(st));
        s.n_bc = h_n;
        DS_ALLOC(s.bc_cell, int32_t, s.n_bc);
        DS_ALLOC(s.bc_off, int64_t, n_blk + 1);
        DS_CHK(bc_blk.alloc((size_t)s.n_bc));
        hipLaunchKernelGGL(k_compact_blocks, dim3(grid_of(n_vis)), dim3(256), 0, st, n_vis, 1, kAsmBlock, k_s.p, vkey.p, seg.p, flag.p, pos.p, s.bc_cell, bc_blk.p,
                           s.bc_off);
        const int64_t h_nbc = s.n_bc;
        DS_CHK(hipMemcpyAsync(s.bc_off + n_blk, &h_nbc, sizeof(int64_t), hipMemcpyHostToDevice, st));
        DS_CHK(hipStreamSynchronize(st));
    }
    {
        const int64_t n2 = s.n_bc * nv;
        if (n2 > INT32_MAX) {
            err = "too many block-cell vertices for the device set-up";
            return FDAPDE_EUNSUPPORTED;
        }
        Tmp<int32_t> k_a, k_s, seg, flag, pos;
        DS_CHK(k_a.alloc((size_t)n2));
        DS_CHK(k_s.alloc((size_t)n2));
        DS_CHK(seg.alloc((size_t)n_blk + 1));
        DS_CHK(flag.alloc((size_t)n2 + 1));
        DS_CHK(pos.alloc((size_t)n2 + 1));
        hipLaunchKernelGGL(k_block_cell_nodes, dim3(grid_of(s.n_bc)), dim3(256), 0, st, s.n_bc, nv, s.bc_cell, s.cverts, k_a.p);
        hipLaunchKernelGGL(k_scaled_offsets64, dim3(grid_of(n_blk + 1)), dim3(256), 0, st, n_blk + 1, s.bc_off, nv, seg.p);
        if (int rc = seg_sort_keys(sc, k_a.p, k_s.p, n2, n_blk, seg.p, bits_of(nn), st, err)) return rc;
        DS_CHK(hipMemsetAsync(flag.p + n2, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_unique_in_blocks, dim3(grid_of(n2)), dim3(256), 0, st, n2, nv, 1, k_s.p, bc_blk.p, seg.p, flag.p);
        if (int rc = exclusive_sum(sc, flag.p, pos.p, n2 + 1, st, err)) return rc;
        int32_t h_n = 0;
        DS_CHK(hipMemcpyAsync(&h_n, pos.p + n2, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        DS_CHK(hipStreamSynchronize(st));
        s.n_bn = h_n;
        DS_ALLOC(s.bn_node, int32_t, s.n_bn);
        DS_ALLOC(s.bn_off, int64_t, n_blk + 1);
        hipLaunchKernelGGL(k_compact_blocks, dim3(grid_of(n2)), dim3(256), 0, st, n2, nv, 1, k_s.p, bc_blk.p, seg.p, flag.p, pos.p, s.bn_node, (int32_t*)nullptr,
                           s.bn_off);
        const int64_t h_nbn = s.n_bn;
        DS_CHK(hipMemcpyAsync(s.bn_off + n_blk, &h_nbn, sizeof(int64_t), hipMemcpyHostToDevice, st));
        DS_ALLOC(s.bc_vert, uint16_t, s.n_bc * 4);
        hipLaunchKernelGGL(k_block_verts, dim3(grid_of(s.n_bc)), dim3(256), 0, st, s.n_bc, nv, bc_blk.p, s.bc_cell, s.cverts, s.bn_off, s.bn_node, s.bc_vert);
        hipLaunchKernelGGL(k_block_nnz_max, dim3(grid_of(n_blk)), dim3(256), 0, st, n_blk, nd, s.rowptr, maxes.p + 1);
        hipLaunchKernelGGL(k_adjacent_diff_max<int64_t>, dim3(std::min(grid_of(n_blk), 512u)), dim3(256), 0, st, n_blk, s.bc_off, maxes.p + 2);
        hipLaunchKernelGGL(k_adjacent_diff_max<int64_t>, dim3(std::min(grid_of(n_blk), 512u)), dim3(256), 0, st, n_blk, s.bn_off, maxes.p + 3);
        DS_CHK(hipStreamSynchronize(st));
    }
    phase("block tables");

    // ---- sliced-ELL adjacency + per-visit column slots
    Tmp<int64_t> width;
    DS_CHK(width.alloc((size_t)n_slices + 1));
    DS_ALLOC(s.sl_off, int64_t, n_slices + 1);
    auto slice_offsets = [&](const int32_t* lane_row, int64_t* total) -> int {
        DS_CHK(hipMemsetAsync(width.p + n_slices, 0, sizeof(int64_t), st));
        hipLaunchKernelGGL(k_slice_widths, dim3(grid_of(n_slices)), dim3(256), 0, st, n_slices, nd, vptr.p, lane_row, width.p);
        if (int rc = exclusive_sum(sc, width.p, s.sl_off, n_slices + 1, st, err)) return rc;
        DS_CHK(hipMemcpyAsync(total, s.sl_off + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        DS_CHK(hipStreamSynchronize(st));
        return FDAPDE_OK;
    };
    int64_t sl_total = 0;
    if (int rc = slice_offsets(nullptr, &sl_total)) return rc;
    Tmp<int32_t> row_pos;
    if ((double)sl_total * kSlice > 1.25 * (double)n_vis) {   // mostly padding: deal the rows of a block to its lanes by visit count
        DS_ALLOC(s.lane_row, int32_t, n_blk * kAsmBlock);
        DS_CHK(row_pos.alloc((size_t)nd));
        hipLaunchKernelGGL(k_deal_rows, dim3((unsigned)n_blk), dim3(kAsmBlock), 0, st, nd, vptr.p, s.lane_row, row_pos.p);
        if (int rc = slice_offsets(s.lane_row, &sl_total)) return rc;
        s.dealt = true;
    }
    const int nbw = (nb * 2 + 3) / 4;
    s.n_adj = sl_total * kSlice;
    if (dbg)
        std::fprintf(stderr, "sliced-ELL adjacency: %lld visit slots for %lld visits (%.2fx), slot words %.2f GB\n", (long long)s.n_adj,
                     (long long)n_vis, (double)s.n_adj / (double)n_vis, (double)s.n_adj * nbw * 4.0 / 1e9);
    DS_ALLOC(s.adj, int32_t, s.n_adj);
    DS_ALLOC(s.slotw, uint32_t, s.n_adj * nbw);
    {
        // (the maxima of the row length and of the blocks' cell lists are needed on the host now: they size the launch's LDS)
        int32_t h_m[3] = {0, 0, 0};
        DS_CHK(hipMemcpyAsync(h_m, maxes.p, sizeof h_m, hipMemcpyDeviceToHost, st));
        DS_CHK(hipStreamSynchronize(st));
        const int max_row = h_m[0] > 0 ? h_m[0] : 1, max_cells = h_m[2];
        const size_t lds = sizeof(int32_t) * ((size_t)kSlice * max_row + (size_t)max_cells);
        if (lds > 150 * 1024) {   // (a row of hundreds of entries: the searches stay in global memory)
            hipLaunchKernelGGL(k_fill_adjacency_global, dim3((unsigned)n_slices), dim3(kSlice, 4), 0, st, nd, nb, nbw, s.sl_off, s.lane_row, vptr.p, vis.p,
                               s.cdofs, s.rowptr, s.colidx, s.bc_off, s.bc_cell, s.adj, s.slotw);
        } else {
            if (lds > 48 * 1024) DS_CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fill_adjacency), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_fill_adjacency, dim3((unsigned)n_slices), dim3(kSlice, 4), lds, st, nd, nb, nbw, max_row, s.sl_off, s.lane_row, vptr.p, vis.p,
                               s.cdofs, s.rowptr, s.colidx, s.bc_off, s.bc_cell, s.adj, s.slotw);
        }
    }
    DS_CHK(hipGetLastError());
    phase("sliced-ELL adjacency + slots");

    // ---- what the host side of the library keeps: permutations, boundary flags, row pointers, sizes
    int32_t h_max[5] = {0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_adjacent_diff_max<int64_t>, dim3(std::min(grid_of(n_slices), 512u)), dim3(256), 0, st, n_slices, s.sl_off, maxes.p + 4);
    DS_CHK(hipMemcpyAsync(h_max, maxes.p, sizeof h_max, hipMemcpyDeviceToHost, st));
    // (the permutations and the boundary flags in internal order stay on the device until host code asks: ensure_host, kHostPerm)
    hs.dof_i2e.clear(), hs.dof_e2i.clear(), hs.cell_i2e.clear(), hs.dof_bnd_i.clear();
    // the row-block list of the CSR-stream SpMV (FDAPDE_SPMV=stream, a diagnostic variant) is host index work on the row pointers: only then
    // are they fetched here; otherwise they come with the pattern's host mirror (ensure_host, kHostPattern)
    const char* spmv_env = std::getenv("FDAPDE_SPMV");
    const bool want_rb = spmv_env && std::string(spmv_env) == "stream";
    hs.rowptr_i.clear(), hs.sl_off.clear();   // (sl_off: n_slices and the widest slice are all host code reads of it)
    hs.n_slices = n_slices;
    if (want_rb) {
        hs.rowptr_i.resize((size_t)nd + 1);
        DS_CHK(hipMemcpyAsync(hs.rowptr_i.data(), s.rowptr, sizeof(int32_t) * ((size_t)nd + 1), hipMemcpyDeviceToHost, st));
    }
    DS_CHK(hipStreamSynchronize(st));
    hs.max_slice_width = h_max[4];
    hs.nnz = nnz, hs.max_row = h_max[0], hs.max_blk_nnz = h_max[1], hs.max_blk_cells = h_max[2], hs.max_blk_nodes = h_max[3], hs.nbw = nbw;
    if (hs.max_row > 65535 || hs.max_row > kSpmvNnz) {
        err = "row too long for the uint16 slot map / SpMV row block";
        return FDAPDE_EUNSUPPORTED;
    }
    if (hs.max_blk_nodes > 65535) {
        err = "assembly block touches more than 65535 nodes";
        return FDAPDE_EUNSUPPORTED;
    }
    hs.rb_row.clear();
    hs.rb_row.push_back(0);
    if (!want_rb) hs.rb_row.push_back((int32_t)nd);   // (never read by the default SpMV)
    for (int64_t r = 0; want_rb && r < nd;) {   // SpMV row blocks: consecutive rows with at most kSpmvNnz nonzeros
        int64_t e = r;
        const int32_t base = hs.rowptr_i[(size_t)r];
        while (e < nd && hs.rowptr_i[(size_t)e + 1] - base <= kSpmvNnz && e - r < 1024) ++e;
        hs.rb_row.push_back((int32_t)e);
        r = e;
    }
    hs.n_colours = 0, hs.colour_off.clear(), hs.colour_cells.clear();
    phase("host mirrors");
    if (dbg) (void)hipEventDestroy(ev0), (void)hipEventDestroy(ev1);
    guard.armed = false;
    *out = s;
    return FDAPDE_OK;
}

void dev_setup_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_invert));
    (void)hipGetLastError();
}

}  // namespace fdapde_hip
