// internal.h -- context layout shared by the host-side setup (host_setup.cpp) and the HIP side (capi.hip).
#ifndef FDAPDE_INTERNAL_H
#define FDAPDE_INTERNAL_H

#include <cstdint>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fdapde_hip.h"

namespace fdapde_hip {

// std::vector whose resize(n) / vector(n) leave trivially-constructible elements UNINITIALISED: the multi-hundred-MB index arrays
// of the set-up get their first touch (page faults included) in the parallel loops that fill them, not in a serial
// value-initialisation.  resize(n, v) / assign(n, v) still write v.
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) noexcept {}
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using hvec = std::vector<T, NoInitAlloc<T>>;

constexpr int kMaxBasis = 10;   // 3-D P2
constexpr int kMaxQuad = 6;     // 2-D P2 rule
constexpr int kMaxTerms = 8;    // leaves per operator expression
constexpr int kSlice = 64;      // rows per adjacency slice = one wavefront
constexpr int kAsmBlock = 256;  // rows per assembly workgroup

// quadrature + reference basis values, built once per (M, R) on the host (tables.cpp)
struct BasisTables {
    int M = 0, R = 0, nb = 0, nq = 0;
    double qn[kMaxQuad * 3] = {};             // quadrature nodes, row-major nq x M
    double qw[kMaxQuad] = {};                 // weights (sum 1)
    double psi[kMaxBasis * kMaxQuad] = {};    // psi_i(p_q), [i*nq + q]
    double dpsi[kMaxBasis * kMaxQuad * 3] = {};   // d psi_i / d xi_k (p_q), [(i*nq + q)*3 + k]
    double refnodes[kMaxBasis * 3] = {};      // reference coordinates of the local DOFs, row-major nb x M
};
int build_basis_tables(int M, int R, BasisTables* t);
int n_basis_of(int M, int R);
int n_quadrature_of(int M, int R);

// host-side description of the discrete space; everything below is integer/index work
struct HostSpace {
    // ---- mesh as handed over (reference numbering, "ext")
    int M = 0, N = 0;
    int64_t n_nodes = 0, n_cells = 0;
    hvec<double> nodes;      // column-major n_nodes x N
    hvec<int32_t> cells;     // row-major n_cells x (M+1)
    std::vector<uint8_t> node_bnd;  // per node
    // ---- DOFs in the reference's numbering
    int order = 0, nb = 0, nq = 0;
    int64_t n_dofs = 0, n_edges = 0;
    hvec<int32_t> dofs;        // row-major n_cells x nb
    std::vector<uint8_t> dof_bnd;     // per DOF
    hvec<double> dof_coords;   // column-major n_dofs x N
    // ---- internal (locality) numbering: Morton order of DOF / node / cell positions
    hvec<int32_t> dof_e2i, dof_i2e, node_e2i, node_i2e, cell_e2i, cell_i2e;
    // ---- CSR pattern, reference numbering (what stiff()/mass() expose)
    hvec<int32_t> rowptr_e, colidx_e;
    // ---- CSR pattern, internal numbering (what the kernels use) + map internal slot -> reference slot
    hvec<int32_t> rowptr_i, colidx_i, slot_i2e, diag_i;   // colidx_i holds nnz + 2 entries (two trailing zeros)
    int64_t nnz = 0;
    int32_t max_row = 0;
    // ---- internal cell data: vertex node ids (internal node numbering) and DOF ids (internal DOF numbering)
    hvec<int32_t> cverts_i;   // n_cells x (M+1)
    hvec<int32_t> cdofs_i;    // n_cells x nb
    hvec<double> vcoords_i;   // internal node id -> NP doubles (NP = 2 for N=2, 4 for N=3)
    std::vector<uint8_t> dof_bnd_i;
    // ---- row-owner adjacency in sliced-ELL layout: slice s covers rows [64 s, 64 s + 64)
    //      entry (s, v, lane) at (sl_off[s] + v) * 64 + lane  holds  (index of the cell in its assembly block's table) * 16
    //      + local_index, or -1 (padding)
    std::vector<int64_t> sl_off;      // n_slices + 1, in units of 64-lane rows (host-built spaces; a device-built space keeps it on the device)
    int64_t n_slices = 0;             // ... its size - 1, and the widest slice (= the longest visit list), whoever built the space
    int32_t max_slice_width = 0;
    hvec<int32_t> adj;         // sl_off.back() * 64
    // lane position -> row (empty = identity).  When the rows of a 256-row assembly block differ a lot in visit count (P2: vertex
    // DOFs ~24 cells, edge DOFs ~5) they are dealt to the block's lanes sorted by visit count, so that a 64-lane slice holds rows
    // of similar length: n_blk * 256 entries, -1 = no row
    hvec<int32_t> lane_row;
    int nbw = 0;                      // 32-bit words of slot data per visit: ceil(nb * 2 / 4)
    hvec<uint32_t> slotw;      // adj.size() * nbw; packed uint16 row-relative slots of the nb local columns
    std::vector<int32_t> blk_nnz_cap; // per assembly block: nnz of its 256 rows
    int32_t max_blk_nnz = 0;
    // ---- per assembly block (256 rows): the cells its rows visit and the vertex nodes of those cells.  `adj` addresses
    //      cells by their index in the block's table; the block stages its nodes' coordinates in LDS once.
    std::vector<int64_t> bc_off;      // n_blk + 1: offsets into bc_cell / bc_vert (in cells)
    hvec<int32_t> bc_cell;     // internal cell id of each block-cell (forcing / coefficient rows)
    hvec<uint16_t> bc_vert;    // 4 per block-cell: block-local node index of each vertex (M+1 used)
    std::vector<int64_t> bn_off;      // n_blk + 1: offsets into bn_node
    hvec<int32_t> bn_node;     // internal node ids staged by the block
    int32_t max_blk_nodes = 0, max_blk_cells = 0;
    // ---- element colouring (cells of one colour share no DOF), for the colour-partitioned scatter
    int n_colours = 0;
    std::vector<int32_t> colour_off;  // n_colours + 1
    std::vector<int32_t> colour_cells;// internal cell ids grouped by colour
    // ---- SpMV row blocks (CSR-stream): rows [rb_row[b], rb_row[b+1]) hold <= kSpmvNnz nonzeros
    std::vector<int32_t> rb_row;
    double setup_ms = 0;
};

constexpr int kSpmvNnz = 4096;  // products staged in LDS per row block (32 KiB)

int host_set_mesh(HostSpace& hs, int M, int N, int64_t n_nodes, const double* nodes, int64_t n_cells,
                  const int32_t* cells, const uint8_t* bnd, std::string& err);
// stop_after: 0 = everything; 1 = stop after the DOF table, boundary DOFs and DOF coordinates (reference numbering): the index
// structures are then built on the device (dev_setup.hip); 2 = sizes only (order, n_basis, n_quadrature): the DOF table too is built on
// the device (dev_topology.hip)
int host_build_space(HostSpace& hs, int order, std::string& err, int stop_after = 0);
int host_build_colouring(HostSpace& hs, std::string& err);
// Solver pattern: the internal CSR pattern without the diagonal and (use_bnd) without rows / columns of Dirichlet DOFs.
// full2s[k] = slot of full entry k in the compact arrays, or -1 when the entry is dropped.
int host_build_solver_pattern_seg(const HostSpace& hs, bool use_bnd, int seg, int wrows, std::vector<int32_t>& rowptr_v,
                                  std::vector<int32_t>& colidx_s, std::vector<int32_t>& full2s, std::vector<int32_t>& vrow);
// 16-bit column codes (k_spmv_team2): groups of kCodeRows rows, four windows of kCodeWindow columns each
constexpr int kCodeRows = 32, kCodeWindow = 1 << 14;
int host_build_col16(int64_t n, const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, std::vector<uint16_t>& code,
                     std::vector<int32_t>& tbase, int64_t* n_wide);
int host_build_solver_pattern(const HostSpace& hs, bool use_bnd, std::vector<int32_t>& rowptr_s, std::vector<int32_t>& colidx_s,
                              std::vector<int32_t>& full2s);


// ---- element-wise scatter forms of the assembly (host_partition.cpp)
struct CellPartitions {
    int32_t cells_per_part = 0, max_colours = 0;
    int64_t n_parts = 0;
    std::vector<int32_t> cell_list;    // n_cells: internal cell ids ordered by (partition, colour inside the partition)
    std::vector<int32_t> colour_off;   // n_parts * (max_colours + 1): positions in cell_list where a partition's colours start
    std::vector<uint8_t> dof_shared;   // n_dofs: 1 = cells of two or more partitions touch this DOF (its row needs atomics)
    std::vector<int32_t> slot_map;     // n_cells * nb * nb: CSR slot of entry (i, j) of the cell at each list position
};
int host_build_cell_partitions(const HostSpace& hs, int cells_per_part, CellPartitions& cp, std::string& err);
void host_build_slot_map(const HostSpace& hs, const int32_t* list, int64_t n, std::vector<int32_t>& out);

// ---- resident layout of the persistent small-problem CG (kernels_persist.h): the interior block of the scaled system cut into
//      one contiguous row range per workgroup (one workgroup per CU), each range as sliced ELL in the order the workgroup's
//      threads own the rows, plus the lists of vector entries workgroups exchange every iteration.
constexpr int kPersistT = 512;        // threads per workgroup
constexpr int kPersistRmax = 16;      // rows per thread at most -- with x, r (and p) of a thread's rows in registers
constexpr int kPersistRwide = 24;     // ... and in the WIDE form of the plain streaming storage (kernels_persist.h, R > 16): x in HBM (one coalesced read + write
                                      // per row and iteration, in slot order), p in its LDS table only, r and y in registers -- 12 288 rows per workgroup,
                                      // i.e. single launches up to 3.1 M rows on 256 CUs.  The symmetric storage cannot follow (its accumulator table
                                      // doubles the LDS per row): beyond 8 192 rows per workgroup the layout is plain
// rows per thread of a layout whose largest workgroup holds rpw rows, max_halo of them importing (halo_free: the importing rows need no
// half of the slots to themselves -- late workgroups / the blocked SpMV); 0 = the system does not fit one launch.  Shared by the host and
// the device builder (the layouts must come out identical).
inline int persist_rows_per_thread(int64_t rpw, int64_t max_halo, bool halo_free, bool sym) {
    constexpr int64_t T = kPersistT;
    int R = 2;
    while (R <= kPersistRmax && ((int64_t)R * T < rpw || (!halo_free && (int64_t)(R / 2) * T < max_halo))) R *= 2;
    if (R <= kPersistRmax) return R;
    if (!sym && (int64_t)kPersistRwide * T >= rpw && (halo_free || (int64_t)(kPersistRwide / 2) * T >= max_halo)) return kPersistRwide;
    return 0;
}
// Symmetric storage (PersistLayout::sym): an off-diagonal pair (i, j) whose rows both lie in one workgroup's block is stored ONCE, in
// the row this rule names (a hash bit, so that every row keeps about half of its in-block entries); the kernel applies it to both
// rows.  Entries whose column belongs to another workgroup stay in both rows.
#ifdef __HIPCC__
__host__ __device__
#endif
inline bool persist_sym_owner(int32_t row, int32_t col) {
    const uint32_t lo = (uint32_t)(row < col ? row : col), hi = (uint32_t)(row < col ? col : row);
    const bool low_owns = ((((lo * 2654435761u) ^ (hi * 2246822519u)) >> 16) & 1u) != 0;
    return (low_owns ? lo : hi) == (uint32_t)row;
}
// sym_mode of the layout builders: 0 plain, 1 symmetric, 2 symmetric where it pays: the plain blocks would not fit the LDS (estimate)
// AND a workgroup owns more than 2048 rows (8 or 16 rows per thread); 3 symmetric wherever the plain blocks would not fit (the caller
// keeps such a layout for smaller workgroups only if its blocks turn out RESIDENT: capi.hip build_persist_once).  Measured on MI355X (tools/persist_sym_ab.py, us per iteration,
// plain -> symmetric): 3-D P1 439 k rows 14.7 -> 15.6, 754 k 20.4 -> 18.6, 1.19 M 31.2 -> 26.2, 1.73 M 39.4 -> 31.0; 2-D P1 1.0 M
// 14.3 -> 12.1, 1.96 M 22.9 -> 18.3; systems whose plain blocks are resident lose 25-30 % (an entry costs ~25 instructions and an LDS atomic instead of a
// multiply-add: it only pays against bytes that would otherwise stream).
// workgroups of a layout: ~2048 rows each, more (fewer rows each) when that makes every block of the matrix fit its workgroup's LDS; ONE
// for a system of up to single_rows rows even if its block then streams (from the L2): a single workgroup needs no hand-off at all.
// MI355X, us per iteration (tools/persist_knob_ab.py, knob persist_single_rows): 3-D P1 1 331 interior rows, two workgroups resident 4.76 ->
// one workgroup streaming 4.23; beyond 2048 rows one workgroup loses: 2-D P1 2 600 rows 4.39 -> 4.82, 3 480 rows 4.87 -> 5.40, 3-D 2 197
// rows (three workgroups) 4.99 -> 6.39
inline int64_t persist_want_workgroups(int64_t n_int, int64_t nnz_kept, int lds_entries, int32_t single_rows) {
    if (n_int <= single_rows) return 1;
    int64_t want = (n_int + 2047) / 2048;
    if (lds_entries > 0) {
        const int64_t fit = (nnz_kept + nnz_kept / 16 + lds_entries - 1) / lds_entries;
        if (fit > want) want = fit;
    }
    return want;
}
inline bool persist_want_sym(int sym_mode, int64_t nnz_kept, int G, int64_t rows_per_wg) {
    const bool plain_streams = 10.6 * (double)nnz_kept / (double)G + 16.0 * (double)rows_per_wg > 150e3;
    return sym_mode == 1 || (sym_mode == 2 && rows_per_wg > 2048 && plain_streams) || (sym_mode == 3 && plain_streams);
}
struct PersistLayout {
    int32_t single_rows = 2048;       // IN (set before the builder is called): systems of up to that many interior rows get ONE workgroup -- its
                                      // iteration needs no hand-off at all (kernels_persist.h, a.G == 1); 0: the general rule only
    bool sym = false;                 // in-block pairs stored once (persist_sym_owner)
    int G = 0, R = 0, nsl = 0;        // workgroups; rows per thread (2, 4, 8, 16); slices of 64 slots per workgroup = R * T / 64.
                                      // Slots [0, T R / 2): rows that import nothing; [T R / 2, T R): the others
    int64_t n_int = 0;                // interior (non-Dirichlet) rows
    int64_t n_entries = 0;            // ELL entries over all workgroups, padding included
    int64_t nnz = 0;                  // stored off-diagonal entries (no padding)
    int64_t nnz_full = 0;             // off-diagonal entries of the interior block (what the plain storage would store)
    int64_t n_board = 0;              // exported vector entries over all workgroups
    int64_t n_imp = 0;                // imported vector entries over all workgroups
    int64_t n_drop = 0;               // rows left out (Dirichlet DOFs)
    int32_t max_imp = 0, max_exp = 0;
    int64_t max_block = 0;            // ELL entries of the largest workgroup block (0: compute from ell_off)
    std::vector<int32_t> slot_dof;    // G * S (S = R * T): internal DOF id of the row a slot holds, -1 = empty slot
    std::vector<int64_t> ell_off;     // G + 1: first ELL entry of a workgroup (multiple of 128)
    std::vector<int32_t> sl_off;      // G * (nsl + 1): slice offsets inside the workgroup's block, in pair rows (128 entries: 64 lanes x 2)
    std::vector<uint16_t> ell_code;   // n_entries: < S: slot of the column's row in this workgroup; >= S: S + index in its import list
    std::vector<int32_t> ell_src;     // n_entries: entry of the full internal pattern holding the value, -1 = padding
    std::vector<int32_t> exp_off;     // G + 1: export list offsets (= board positions)
    std::vector<uint16_t> exp_slot;   // n_board: slot whose vector entry is published at that board position
    std::vector<int32_t> imp_off;     // G + 1
    std::vector<int32_t> imp_pos;     // board position each imported entry is read from
    // row-distributed form (ghost_order given to the host builder): DOFs other ranks own are columns only; a workgroup imports them from
    // board positions n_board + (index in ghost_needed) -- entries the owning ranks push there
    std::vector<int32_t> ghost_needed;   // ghost DOFs some row of this rank reads, in board order (ascending ghost_order)
    std::vector<int32_t> wg_of, slot_of; // per DOF: workgroup / slot of its row (-1: no row here)
    std::vector<uint8_t> wg_late;        // allow_late: 1 = the workgroup's importing rows overflow the second half of its slots: imports before the first pass
};
// n_wg: workgroups available (CUs of the device); lds_entries: ELL entries a workgroup can keep in LDS -- the rows are spread over
// enough workgroups for every block to be resident where the device has that many (3-D rows: 14 entries each, so ~850 rows per
// workgroup instead of 2048).  Returns FDAPDE_EUNSUPPORTED when the system does not fit the layout (more than n_wg * T * Rmax interior
// rows, rows or lists too long for the 16-bit codes).
// ghost_order (row-distributed form): per DOF -1 = this rank's own, else a unique non-negative sort key (the caller's order of the DOFs
// owned by other ranks: by owner, then by global key): such a DOF has no row here and is imported from the board's remote section.
int host_build_persist_layout(const HostSpace& hs, bool use_bnd, int n_wg, int lds_entries, PersistLayout& pl, const int32_t* block_rows = nullptr,
                              int sym_mode = 0, bool balance = false, const int32_t* ghost_order = nullptr, bool allow_late = false);

}  // namespace fdapde_hip
#endif
