// dev_topology.hip -- the topology tables of Triangulation<2,2> / Triangulation<3,3> built ON THE DEVICE
// (fdaPDE/geometry/triangulation.h:143-196 and 319-399): facets (edges of triangles / faces of tetrahedra) in the reference's
// first-seen numbering, facet -> cells, cell -> facets, neighbours, boundary markers; for tetrahedra also the edges (numbered as
// the reference does, through the newly seen faces) and face -> edges.
//
// The reference walks the cells in ascending order with a hash map per facet.  The same numbering falls out of sorts:
//   every cell emits its M + 1 facets in the order of combinations<M, M+1> (utils/combinatorics.h:37-51) as (sorted node tuple,
//   emission index e = cell * (M + 1) + j); a STABLE radix sort by the tuple brings the two occurrences of an interior facet
//   together with the earlier one first; a facet's id is the rank of its first occurrence among all first occurrences in emission
//   order (an exclusive scan of a 0/1 flag indexed by e) -- exactly "edge_id++ when never seen before".  The neighbour across
//   facet j of a cell sits in the column of the vertex that is not on it, which for combinations order is M - j.
// Sorts and scans are hipCUB device primitives (set-up, not the hot path); everything else is small hand-written kernels.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <string>

#include "dev_topology.h"

namespace fdapde_hip {

namespace {

#define TOPO_CHK(expr)                                                        \
    do {                                                                      \
        hipError_t e__ = (expr);                                              \
        if (e__ != hipSuccess) {                                              \
            err = std::string(#expr) + ": " + hipGetErrorString(e__);         \
            return FDAPDE_EHIP;                                               \
        }                                                                     \
    } while (0)

template <typename T> struct Tmp {   // scratch buffer released on scope exit
    T* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (n ? n : 1)); }
    ~Tmp() {
        if (p) (void)hipFree(p);
    }
};

// sorted node tuple of facet j (combinations order) of a cell; t[M-1] is the largest node
template <int M> __device__ __forceinline__ void facet_tuple(const int32_t* cells, int64_t e, int32_t* t) {
    const int64_t c = e / (M + 1);
    const int j = (int)(e - c * (M + 1));
    const int32_t* cv = cells + c * (M + 1);
    // combinations<M, M+1> in lexicographic order: facet j leaves out local vertex M - j
    int k = 0;
#pragma unroll
    for (int v = 0; v <= M; ++v)
        if (v != M - j) t[k++] = cv[v];
    if constexpr (M == 2) {
        if (t[0] > t[1]) { const int32_t s = t[0]; t[0] = t[1], t[1] = s; }
    } else {
        if (t[0] > t[1]) { const int32_t s = t[0]; t[0] = t[1], t[1] = s; }
        if (t[1] > t[2]) { const int32_t s = t[1]; t[1] = t[2], t[2] = s; }
        if (t[0] > t[1]) { const int32_t s = t[0]; t[0] = t[1], t[1] = s; }
    }
}

template <int M> __global__ void k_emit_facets(int64_t n_em, const int32_t* cells, uint64_t* key_lo, uint32_t* key_hi, int32_t* val) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_em) return;
    int32_t t[M];
    facet_tuple<M>(cells, e, t);
    key_lo[e] = ((uint64_t)(uint32_t)t[0] << 32) | (uint32_t)t[1];
    if constexpr (M == 3) key_hi[e] = (uint32_t)t[2];
    val[e] = (int32_t)e;
}
// after the first (third-node) pass of the 3-D sort: the 64-bit key of each value in its current order
__global__ void k_gather_keys(int64_t n, const uint64_t* key_by_e, const int32_t* val, uint64_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = key_by_e[val[i]];
}
// head flags of the runs of equal tuples in sorted order; first_flag[e] = 1 iff emission e is the first occurrence of its facet
template <int M> __global__ void k_facet_heads(int64_t n_em, const int32_t* cells, const int32_t* val, uint8_t* head, int32_t* first_flag,
                                               int32_t* bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_em) return;
    int32_t t[M], u[M];
    facet_tuple<M>(cells, val[i], t);
    bool h = i == 0;
    if (!h) {
        facet_tuple<M>(cells, val[i - 1], u);
        bool same = true;
#pragma unroll
        for (int k = 0; k < M; ++k) same = same && t[k] == u[k];
        h = !same;
        if (same && i >= 2) {   // a third cell on the same facet: not a manifold mesh (the reference would renumber it)
            facet_tuple<M>(cells, val[i - 2], u);
            bool same2 = true;
#pragma unroll
            for (int k = 0; k < M; ++k) same2 = same2 && t[k] == u[k];
            if (same2) atomicOr(bad, 1);
        }
    }
    head[i] = h ? 1 : 0;
    first_flag[val[i]] = h ? 1 : 0;
}
template <int M>
__global__ void k_facet_tables(int64_t n_em, const int32_t* cells, const int32_t* val, const uint8_t* head, const int32_t* rank,
                               int32_t* facet_nodes, int32_t* facet_cells, uint8_t* facet_bnd, int32_t* cell_facets, int32_t* neighbors) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_em || !head[i]) return;
    const int32_t e = val[i];
    const int32_t p = (i + 1 < n_em && !head[i + 1]) ? val[i + 1] : -1;   // the second cell's occurrence, if any
    const int32_t id = rank[e];
    int32_t t[M];
    facet_tuple<M>(cells, e, t);
#pragma unroll
    for (int k = 0; k < M; ++k) facet_nodes[(int64_t)id * M + k] = t[k];
    const int32_t c0 = e / (M + 1), j0 = e - c0 * (M + 1);
    cell_facets[e] = id;
    if (p >= 0) {
        const int32_t c1 = p / (M + 1), j1 = p - c1 * (M + 1);
        facet_cells[2 * (int64_t)id] = c0, facet_cells[2 * (int64_t)id + 1] = c1;
        facet_bnd[id] = 0;
        cell_facets[p] = id;
        neighbors[(int64_t)c0 * (M + 1) + (M - j0)] = c1;   // column = the vertex that is not on the shared facet
        neighbors[(int64_t)c1 * (M + 1) + (M - j1)] = c0;
    } else {
        facet_cells[2 * (int64_t)id] = c0, facet_cells[2 * (int64_t)id + 1] = -1;
        facet_bnd[id] = 1;   // seen by exactly one cell (triangulation.h:177,187 / 364,385)
        neighbors[(int64_t)c0 * (M + 1) + (M - j0)] = -1;
    }
}

// ---- edges of tetrahedra: face f (first-seen order) emits the pairs (0,1), (0,2), (1,2) of its sorted nodes, e2 = 3 f + k
__global__ void k_emit_face_edges(int64_t n_em, const int32_t* face_nodes, uint64_t* key, int32_t* val) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_em) return;
    const int64_t f = e / 3;
    const int k = (int)(e - f * 3);
    const int32_t a = face_nodes[f * 3 + (k == 2 ? 1 : 0)], b = face_nodes[f * 3 + (k == 0 ? 1 : 2)];   // already ascending
    key[e] = ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
    val[e] = (int32_t)e;
}
__global__ void k_edge_heads(int64_t n, const uint64_t* key_sorted, const int32_t* val, int32_t* head_pos, int32_t* first_flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool h = i == 0 || key_sorted[i] != key_sorted[i - 1];
    head_pos[i] = h ? (int32_t)i : 0;   // running maximum -> position of the run's head
    first_flag[val[i]] = h ? 1 : 0;
}
__global__ void k_edge_tables(int64_t n, const uint64_t* key_sorted, const int32_t* val, const int32_t* head_pos, const int32_t* rank,
                              const uint8_t* node_bnd, int32_t* edge_nodes, uint8_t* edge_bnd, int32_t* face_edges, int32_t* edge_face) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t hp = head_pos[i];
    const int32_t id = rank[val[hp]];
    face_edges[val[i]] = id;
    if (hp == i) {
        const int32_t a = (int32_t)(key_sorted[i] >> 32), b = (int32_t)(key_sorted[i] & 0xffffffffu);
        edge_nodes[2 * (int64_t)id] = a, edge_nodes[2 * (int64_t)id + 1] = b;
        edge_bnd[id] = (node_bnd[a] && node_bnd[b]) ? 1 : 0;   // triangulation.h:371
        edge_face[id] = val[i] / 3;                             // the head of the run is the edge's first emission: face = e2 / 3
    }
}

// ---- order-2 DOF table from the topology ---------------------------------------------------------------------------------
__global__ void k_p2_dofs_2d(int64_t nc, int32_t nn, const int32_t* cells, const int32_t* cell_facets, int32_t* dofs) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    for (int v = 0; v < 3; ++v) dofs[c * 6 + v] = cells[c * 3 + v];
    for (int j = 0; j < 3; ++j) dofs[c * 6 + 3 + j] = nn + cell_facets[c * 3 + j];   // pairs (0,1), (0,2), (1,2) = local nodes 3, 4, 5
}
__global__ void k_p2_dofs_3d(int64_t nc, int32_t nn, const int32_t* cells, const int32_t* cell_facets, const int32_t* facet_nodes,
                             const int32_t* face_edges, int32_t* dofs) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const int32_t* cv = cells + c * 4;
    for (int v = 0; v < 4; ++v) dofs[c * 10 + v] = cv[v];
    // local pair (a, b) -> slot of its midpoint in ReferenceElement<3,2>::nodes (host_setup.cpp edge_slot)
    const int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3}, SL[6] = {6, 5, 9, 4, 7, 8};
    for (int p = 0; p < 6; ++p) {
        const int a = PA[p], b = PB[p];
        int j = 0;   // a face of the cell that contains both: face j (combinations order) leaves out local vertex 3 - j
        while (3 - j == a || 3 - j == b) ++j;
        const int64_t f = cell_facets[c * 4 + j];
        const int32_t na = cv[a], nb_ = cv[b], lo = na < nb_ ? na : nb_, hi = na < nb_ ? nb_ : na;
        const int32_t* t = facet_nodes + f * 3;
        const int k = (lo == t[0] && hi == t[1]) ? 0 : ((lo == t[0] && hi == t[2]) ? 1 : 2);
        dofs[c * 10 + SL[p]] = nn + face_edges[f * 3 + k];
    }
}
__global__ void k_p2_bnd(int64_t nn, int64_t ne, const uint8_t* node_bnd, const uint8_t* edge_bnd, uint8_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nn) out[i] = node_bnd[i];
    else if (i < nn + ne) out[i] = edge_bnd[i - nn];
}
struct RefNodes {
    double v[10 * 3];
};
// the edge DOF's coordinates from the first cell that visits it: acc = sum_k (x_{k+1} - x_0) ref_k (in this order, no contraction), + x_0
template <int M>
__global__ void k_p2_coords(int64_t nn, int64_t ne, int N, int nb, const double* nodes, const int32_t* cells, const int32_t* dofs,
                            const int32_t* first_cell_of /* 2-D: facet_cells (stride 2); 3-D: edge_face */, const int32_t* facet_cells,
                            RefNodes ref, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nd = nn + ne;
    if (i < nn) {
        for (int d = 0; d < N; ++d) out[(int64_t)d * nd + i] = nodes[(int64_t)d * nn + i];
        return;
    }
    if (i >= nd) return;
    const int64_t e = i - nn;
    const int64_t c = M == 2 ? first_cell_of[2 * e] : facet_cells[2 * (int64_t)first_cell_of[e]];
    const int32_t dof = (int32_t)i;
    int j = M + 1;
    while (j < nb - 1 && dofs[c * nb + j] != dof) ++j;
    const int32_t v0 = cells[c * (M + 1)];
    for (int d = 0; d < N; ++d) {
        const double x0 = nodes[(int64_t)d * nn + v0];
        double acc = 0;
        for (int k = 0; k < M; ++k)
            acc = __dadd_rn(acc, __dmul_rn(__dsub_rn(nodes[(int64_t)d * nn + cells[c * (M + 1) + k + 1]], x0), ref.v[j * M + k]));
        out[(int64_t)d * nd + i] = __dadd_rn(acc, x0);
    }
}

struct MaxOp {
    __device__ __forceinline__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; }
};

inline unsigned grid_of(int64_t n) { return (unsigned)((n + 255) / 256); }

template <int M>
int build_t(int64_t n_nodes, int64_t n_cells, const int32_t* d_cells, const uint8_t* d_node_bnd, hipStream_t st, DevTopology* out,
            std::string& err) {
    const int64_t n_em = n_cells * (M + 1);
    if (n_em > INT32_MAX) {
        err = "too many cells for the device topology builder (cells x facets per cell exceeds int32)";
        return FDAPDE_EUNSUPPORTED;
    }
    (void)n_nodes;
    Tmp<uint64_t> key_a, key_b, key_e;
    Tmp<uint32_t> hi_a, hi_b;
    Tmp<int32_t> val_a, val_b, flag, rank, bad;
    Tmp<uint8_t> head, scratch;
    TOPO_CHK(key_a.alloc((size_t)n_em));
    TOPO_CHK(key_b.alloc((size_t)n_em));
    TOPO_CHK(val_a.alloc((size_t)n_em));
    TOPO_CHK(val_b.alloc((size_t)n_em));
    TOPO_CHK(flag.alloc((size_t)n_em + 1));
    TOPO_CHK(rank.alloc((size_t)n_em + 1));
    TOPO_CHK(head.alloc((size_t)n_em));
    TOPO_CHK(bad.alloc(1));
    if (M == 3) {
        TOPO_CHK(hi_a.alloc((size_t)n_em));
        TOPO_CHK(hi_b.alloc((size_t)n_em));
        TOPO_CHK(key_e.alloc((size_t)n_em));
    }
    TOPO_CHK(hipMemsetAsync(bad.p, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_emit_facets<M>, dim3(grid_of(n_em)), dim3(256), 0, st, n_em, d_cells, M == 3 ? key_e.p : key_a.p, hi_a.p, val_a.p);
    // scratch for the device primitives: sized for the largest request below
    size_t sb = 0, need = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, need, key_a.p, key_b.p, val_a.p, val_b.p, (int)n_em, 0, 64, st);
    sb = need;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, need, hi_a.p, hi_b.p, val_a.p, val_b.p, (int)n_em, 0, 32, st);
    sb = need > sb ? need : sb;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, need, flag.p, rank.p, (int)n_em + 1, st);
    sb = need > sb ? need : sb;
    TOPO_CHK(scratch.alloc(sb));
    int32_t* sorted_val = nullptr;
    if (M == 2) {
        TOPO_CHK(hipcub::DeviceRadixSort::SortPairs(scratch.p, sb, key_a.p, key_b.p, val_a.p, val_b.p, (int)n_em, 0, 64, st));
        sorted_val = val_b.p;
    } else {   // least significant key first: the third node, then (stable) the first two
        TOPO_CHK(hipcub::DeviceRadixSort::SortPairs(scratch.p, sb, hi_a.p, hi_b.p, val_a.p, val_b.p, (int)n_em, 0, 32, st));
        hipLaunchKernelGGL(k_gather_keys, dim3(grid_of(n_em)), dim3(256), 0, st, n_em, key_e.p, val_b.p, key_a.p);
        TOPO_CHK(hipcub::DeviceRadixSort::SortPairs(scratch.p, sb, key_a.p, key_b.p, val_b.p, val_a.p, (int)n_em, 0, 64, st));
        sorted_val = val_a.p;
    }
    TOPO_CHK(hipMemsetAsync(flag.p + n_em, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_facet_heads<M>, dim3(grid_of(n_em)), dim3(256), 0, st, n_em, d_cells, sorted_val, head.p, flag.p, bad.p);
    TOPO_CHK(hipcub::DeviceScan::ExclusiveSum(scratch.p, sb, flag.p, rank.p, (int)n_em + 1, st));
    int32_t h_nf = 0, h_bad = 0;
    TOPO_CHK(hipMemcpyAsync(&h_nf, rank.p + n_em, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TOPO_CHK(hipMemcpyAsync(&h_bad, bad.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TOPO_CHK(hipStreamSynchronize(st));
    if (h_bad) {
        err = "a facet is shared by more than two cells (not a manifold mesh)";
        return FDAPDE_EUNSUPPORTED;
    }
    const int64_t nf = h_nf;
    out->M = M, out->n_cells = n_cells, out->n_facets = nf;
    TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->facet_nodes), sizeof(int32_t) * (size_t)(nf ? nf : 1) * M));
    TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->facet_cells), sizeof(int32_t) * (size_t)(nf ? nf : 1) * 2));
    TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->facet_bnd), (size_t)(nf ? nf : 1)));
    TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->cell_facets), sizeof(int32_t) * (size_t)(n_em ? n_em : 1)));
    TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->neighbors), sizeof(int32_t) * (size_t)(n_em ? n_em : 1)));
    hipLaunchKernelGGL(k_facet_tables<M>, dim3(grid_of(n_em)), dim3(256), 0, st, n_em, d_cells, sorted_val, head.p, rank.p, out->facet_nodes,
                       out->facet_cells, out->facet_bnd, out->cell_facets, out->neighbors);
    if (M == 3) {
        const int64_t n2 = nf * 3;
        if (n2 > INT32_MAX) {
            err = "too many faces for the device topology builder";
            return FDAPDE_EUNSUPPORTED;
        }
        Tmp<uint64_t> ek_a, ek_b;
        Tmp<int32_t> ev_a, ev_b, eflag, erank, head_pos, head_max;
        Tmp<uint8_t> escratch;
        TOPO_CHK(ek_a.alloc((size_t)n2));
        TOPO_CHK(ek_b.alloc((size_t)n2));
        TOPO_CHK(ev_a.alloc((size_t)n2));
        TOPO_CHK(ev_b.alloc((size_t)n2));
        TOPO_CHK(eflag.alloc((size_t)n2 + 1));
        TOPO_CHK(erank.alloc((size_t)n2 + 1));
        TOPO_CHK(head_pos.alloc((size_t)n2));
        TOPO_CHK(head_max.alloc((size_t)n2));
        size_t eb = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, need, ek_a.p, ek_b.p, ev_a.p, ev_b.p, (int)n2, 0, 64, st);
        eb = need;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, need, eflag.p, erank.p, (int)n2 + 1, st);
        eb = need > eb ? need : eb;
        (void)hipcub::DeviceScan::InclusiveScan(nullptr, need, head_pos.p, head_max.p, MaxOp(), (int)n2, st);
        eb = need > eb ? need : eb;
        TOPO_CHK(escratch.alloc(eb));
        hipLaunchKernelGGL(k_emit_face_edges, dim3(grid_of(n2)), dim3(256), 0, st, n2, out->facet_nodes, ek_a.p, ev_a.p);
        TOPO_CHK(hipcub::DeviceRadixSort::SortPairs(escratch.p, eb, ek_a.p, ek_b.p, ev_a.p, ev_b.p, (int)n2, 0, 64, st));
        TOPO_CHK(hipMemsetAsync(eflag.p + n2, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_edge_heads, dim3(grid_of(n2)), dim3(256), 0, st, n2, ek_b.p, ev_b.p, head_pos.p, eflag.p);
        TOPO_CHK(hipcub::DeviceScan::ExclusiveSum(escratch.p, eb, eflag.p, erank.p, (int)n2 + 1, st));
        TOPO_CHK(hipcub::DeviceScan::InclusiveScan(escratch.p, eb, head_pos.p, head_max.p, MaxOp(), (int)n2, st));
        int32_t h_ne = 0;
        TOPO_CHK(hipMemcpyAsync(&h_ne, erank.p + n2, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        TOPO_CHK(hipStreamSynchronize(st));
        out->n_edges = h_ne;
        TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->edge_nodes), sizeof(int32_t) * (size_t)(h_ne ? h_ne : 1) * 2));
        TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->edge_bnd), (size_t)(h_ne ? h_ne : 1)));
        TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->face_edges), sizeof(int32_t) * (size_t)(n2 ? n2 : 1)));
        TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&out->edge_face), sizeof(int32_t) * (size_t)(h_ne ? h_ne : 1)));
        hipLaunchKernelGGL(k_edge_tables, dim3(grid_of(n2)), dim3(256), 0, st, n2, ek_b.p, ev_b.p, head_max.p, erank.p, d_node_bnd,
                           out->edge_nodes, out->edge_bnd, out->face_edges, out->edge_face);
        TOPO_CHK(hipGetLastError());
        TOPO_CHK(hipStreamSynchronize(st));   // the scratch buffers of this scope are freed on exit
    } else {
        out->n_edges = nf;
    }
    TOPO_CHK(hipGetLastError());
    TOPO_CHK(hipStreamSynchronize(st));
    return FDAPDE_OK;
}

}  // namespace

void dev_topology_release(DevTopology* t) {
    if (!t) return;
    for (void* p : {(void*)t->facet_nodes, (void*)t->facet_cells, (void*)t->facet_bnd, (void*)t->cell_facets, (void*)t->neighbors,
                    (void*)t->edge_nodes, (void*)t->edge_bnd, (void*)t->face_edges, (void*)t->edge_face})
        if (p) (void)hipFree(p);
    *t = DevTopology{};
}

int dev_build_topology(int M, int64_t n_nodes, int64_t n_cells, const int32_t* d_cells, const uint8_t* d_node_bnd, void* stream,
                       DevTopology* out, std::string& err) {
    dev_topology_release(out);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = FDAPDE_EUNSUPPORTED;
    if (M == 2) rc = build_t<2>(n_nodes, n_cells, d_cells, d_node_bnd, st, out, err);
    else if (M == 3) rc = build_t<3>(n_nodes, n_cells, d_cells, d_node_bnd, st, out, err);
    else err = "topology tables exist for triangles and tetrahedra";
    if (rc != FDAPDE_OK) dev_topology_release(out);
    return rc;
}

int dev_build_p2_dofs(int M, int64_t n_nodes, int64_t n_cells, const double* d_nodes, const int32_t* d_cells, const uint8_t* d_node_bnd,
                      const double* refnodes, void* stream, int32_t** d_dofs, uint8_t** d_dof_bnd, double** d_dof_coords, int64_t* n_edges,
                      std::string& err) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (M != 2 && M != 3) return FDAPDE_EUNSUPPORTED;
    DevTopology t;
    if (int rc = dev_build_topology(M, n_nodes, n_cells, d_cells, d_node_bnd, stream, &t, err)) return rc;
    struct Guard {
        DevTopology* t;
        ~Guard() { dev_topology_release(t); }
    } guard{&t};
    const int nb = M == 2 ? 6 : 10, N = M;
    const int64_t ne = t.n_edges, nd = n_nodes + ne;
    if (nd > INT32_MAX) {
        err = "too many DOFs";
        return FDAPDE_EUNSUPPORTED;
    }
    int32_t* dofs = nullptr;
    uint8_t* bnd = nullptr;
    double* coords = nullptr;
    TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&dofs), sizeof(int32_t) * (size_t)n_cells * nb));
    TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&bnd), (size_t)nd));
    TOPO_CHK(hipMalloc(reinterpret_cast<void**>(&coords), sizeof(double) * (size_t)nd * N));
    RefNodes ref{};
    for (int i = 0; i < nb * M; ++i) ref.v[i] = refnodes[i];
    if (M == 2) {
        hipLaunchKernelGGL(k_p2_dofs_2d, dim3(grid_of(n_cells)), dim3(256), 0, st, n_cells, (int32_t)n_nodes, d_cells, t.cell_facets, dofs);
        hipLaunchKernelGGL(k_p2_bnd, dim3(grid_of(nd)), dim3(256), 0, st, n_nodes, ne, d_node_bnd, t.facet_bnd, bnd);
        hipLaunchKernelGGL(k_p2_coords<2>, dim3(grid_of(nd)), dim3(256), 0, st, n_nodes, ne, N, nb, d_nodes, d_cells, dofs, t.facet_cells, t.facet_cells, ref,
                           coords);
    } else {
        hipLaunchKernelGGL(k_p2_dofs_3d, dim3(grid_of(n_cells)), dim3(256), 0, st, n_cells, (int32_t)n_nodes, d_cells, t.cell_facets, t.facet_nodes,
                           t.face_edges, dofs);
        hipLaunchKernelGGL(k_p2_bnd, dim3(grid_of(nd)), dim3(256), 0, st, n_nodes, ne, d_node_bnd, t.edge_bnd, bnd);
        hipLaunchKernelGGL(k_p2_coords<3>, dim3(grid_of(nd)), dim3(256), 0, st, n_nodes, ne, N, nb, d_nodes, d_cells, dofs, t.edge_face, t.facet_cells, ref,
                           coords);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipFree(dofs), (void)hipFree(bnd), (void)hipFree(coords);
        err = "order-2 DOF kernels failed";
        return FDAPDE_EHIP;
    }
    *d_dofs = dofs, *d_dof_bnd = bnd, *d_dof_coords = coords, *n_edges = ne;
    return FDAPDE_OK;
}

void dev_topology_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_gather_keys));
    (void)hipGetLastError();
}

}  // namespace fdapde_hip
