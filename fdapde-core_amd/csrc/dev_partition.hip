// dev_partition.hip -- see dev_partition.h.  Stage for stage the arithmetic of fdapde-core_amd/dist.py (partition_cells, node_owners,
// rowdist_sub_mesh / sub_mesh), so that the arrays come out identical to the numpy ones (tests/test_gpu_partition.py compares them):
//   cell barycentres  ((x0 + x1) + x2 [+ x3]) / (M + 1), their bounding box (device reductions: exact whatever the order)
//   Morton keys       q_k = trunc((p_k - lo_k) / span_k * (2^bits - 1)), bits 21 (3-D) / 31 (2-D), bit b of axis k at position b d + k
//   chunks            stable radix sort of (key, cell); cell at sorted position i goes to the rank whose [bounds[r], bounds[r + 1]) holds i,
//                     bounds[r] = trunc(r * (n_cells / world))   (numpy.linspace(0, n_cells, world + 1).astype(int64))
//   node owners       lowest / highest rank among the cells touching a node (atomic min / max); row-distributed form: the lowest in the "white"
//                     boxes of a 16-per-axis checkerboard over the nodes' bounding box, the highest in the "black" ones (interfaces dealt to both
//                     sides in patches); element form: the lowest
//   sub-meshes        per rank: flag its cells (row-distributed: any vertex owned; element form: part == rank), exclusive scan = local cell
//                     position (ascending global id), flag + scan the nodes of those cells = local node id (ascending global id), gather.
// Sorts, scans and reductions are hipCUB device primitives (set-up, not the hot path); the rest is small hand-written kernels.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/fdapde_hip.h"
#include "dev_partition.h"

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

template <typename T> struct Scratch {   // device scratch released on scope exit
    T* p = nullptr;
    hipError_t alloc(size_t n) {
        release();
        return hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (n ? n : 1));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    ~Scratch() { release(); }
};

inline unsigned grid_of(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

__device__ __forceinline__ uint64_t p_spread3(uint64_t x) {
    x &= 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ uint64_t p_spread2(uint64_t x) {
    x &= 0x7fffffff;
    x = (x | x << 16) & 0x0000ffff0000ffffull;
    x = (x | x << 8) & 0x00ff00ff00ff00ffull;
    x = (x | x << 4) & 0x0f0f0f0f0f0f0f0full;
    x = (x | x << 2) & 0x3333333333333333ull;
    x = (x | x << 1) & 0x5555555555555555ull;
    return x;
}

// bary column-major n_cells x N: the vertices summed in local order, then ONE division (numpy: nodes[cells].mean(axis=1))
__global__ void k_part_bary(int64_t nc, int N, int nv, int64_t nn, const double* nodes, const int32_t* cells, double* bary) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    for (int k = 0; k < N; ++k) {
        double s = nodes[(int64_t)k * nn + cells[c * nv]];
        for (int v = 1; v < nv; ++v) s = s + nodes[(int64_t)k * nn + cells[c * nv + v]];
        bary[(int64_t)k * nc + c] = s / (double)nv;
    }
}
// bb: lo[3] | hi[3]
__global__ void k_part_keys(int64_t n, int N, const double* pts, const double* bb, uint64_t* key, int32_t* idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int bits = N == 3 ? 21 : 31;
    const uint64_t top = (uint64_t(1) << bits) - 1;
    uint64_t q[3] = {0, 0, 0};
    for (int k = 0; k < N; ++k) {
        const double lo = bb[k], hi = bb[3 + k];
        const double span = hi > lo ? hi - lo : 1.0;
        const double t = ((pts[(int64_t)k * n + i] - lo) / span) * (double)top;
        const uint64_t v = (uint64_t)t;
        q[k] = v < top ? v : top;
    }
    key[i] = N == 3 ? (p_spread3(q[0]) | p_spread3(q[1]) << 1 | p_spread3(q[2]) << 2) : (p_spread2(q[0]) | p_spread2(q[1]) << 1);
    idx[i] = (int32_t)i;
}
// bounds: world + 1 entries (device); sorted position i -> rank
__global__ void k_part_assign(int64_t nc, int world, const int64_t* bounds, const int32_t* order, int32_t* part) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    int r = 0;
    while (r + 1 < world && i >= bounds[r + 1]) ++r;
    part[order[i]] = r;
}
__global__ void k_part_fill_i32(int64_t n, int32_t v, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
__global__ void k_part_node_minmax(int64_t nc, int nv, const int32_t* cells, const int32_t* part, int32_t* lo, int32_t* hi) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const int32_t p = part[c];
    for (int v = 0; v < nv; ++v) {
        const int32_t n = cells[c * nv + v];
        atomicMin(&lo[n], p);
        atomicMax(&hi[n], p);
    }
}
// bbn: lo[3] | hi[3] of the NODES
__global__ void k_part_owner(int64_t nn, int N, int form, const double* nodes, const double* bbn, const int32_t* lo, const int32_t* hi, int32_t* owner) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    if (form != kPartitionRowdist) {
        owner[i] = lo[i];
        return;
    }
    int64_t box = 0;
    for (int k = 0; k < N; ++k) {
        const double a = bbn[k], b = bbn[3 + k];
        const double span = b > a ? b - a : 1.0;
        box += (int64_t)floor(((nodes[(int64_t)k * nn + i] - a) / span) * 16.0);
    }
    owner[i] = (box % 2 == 0) ? lo[i] : hi[i];
}
__global__ void k_part_flag_cells(int64_t nc, int nv, int form, int rank, const int32_t* cells, const int32_t* part, const int32_t* owner, int32_t* flag) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    int f = 0;
    if (form == kPartitionRowdist) {
        for (int v = 0; v < nv; ++v) f |= owner[cells[c * nv + v]] == rank ? 1 : 0;
    } else
        f = part[c] == rank ? 1 : 0;
    flag[c] = f;
}
__global__ void k_part_flag_nodes(int64_t nc, int nv, const int32_t* cells, const int32_t* cflag, int32_t* nflag) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc || !cflag[c]) return;
    for (int v = 0; v < nv; ++v) nflag[cells[c * nv + v]] = 1;
}
__global__ void k_part_compact_cells(int64_t nc, int nv, const int32_t* cells, const int32_t* cflag, const int32_t* cpos, const int32_t* npos, int32_t* cell_ids,
                                     int32_t* lcells) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc || !cflag[c]) return;
    const int64_t at = cpos[c];
    cell_ids[at] = (int32_t)c;
    for (int v = 0; v < nv; ++v) lcells[at * nv + v] = npos[cells[c * nv + v]];
}
__global__ void k_part_compact_nodes(int64_t nn, int N, int64_t n_loc, uint64_t bit, const double* nodes, const uint8_t* bnd, const int32_t* owner, const int32_t* nflag,
                                     const int32_t* npos, int32_t* l2g, double* lnodes, uint8_t* lbnd, int32_t* lowner, uint64_t* mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn || !nflag[i]) return;
    const int64_t at = npos[i];
    l2g[at] = (int32_t)i;
    for (int k = 0; k < N; ++k) lnodes[(int64_t)k * n_loc + at] = nodes[(int64_t)k * nn + i];
    lbnd[at] = bnd[i], lowner[at] = owner[i];
    mask[i] |= bit;   // (one launch per rank on one stream: no two threads touch the same word)
}

struct MinOp {
    __host__ __device__ double operator()(double a, double b) const { return a < b ? a : b; }
};
struct MaxOp {
    __host__ __device__ double operator()(double a, double b) const { return a > b ? a : b; }
};

// lo[3] | hi[3] of the N columns of a column-major n x N array -> d_bb (device, 6 doubles)
int column_bounds(const double* d_pts, int64_t n, int N, hipStream_t st, double* d_bb, std::string& err) {
    size_t tmp_bytes = 0;
    DP_CHK(hipcub::DeviceReduce::Reduce(nullptr, tmp_bytes, d_pts, d_bb, (int)n, MinOp{}, 1e300, st));
    Scratch<char> tmp;
    DP_CHK(tmp.alloc(tmp_bytes + 16));
    DP_CHK(hipMemsetAsync(d_bb, 0, 6 * sizeof(double), st));
    for (int k = 0; k < N; ++k) {
        DP_CHK(hipcub::DeviceReduce::Reduce(tmp.p, tmp_bytes, d_pts + (int64_t)k * n, d_bb + k, (int)n, MinOp{}, 1e300, st));
        DP_CHK(hipcub::DeviceReduce::Reduce(tmp.p, tmp_bytes, d_pts + (int64_t)k * n, d_bb + 3 + k, (int)n, MaxOp{}, -1e300, st));
    }
    DP_CHK(hipStreamSynchronize(st));   // (tmp dies here)
    return FDAPDE_OK;
}

}   // namespace

void dev_partition_release(DevPartition* p) {
    if (!p) return;
    if (p->device >= 0) (void)hipSetDevice(p->device);
    for (RankMeshDev& r : p->ranks) {
        for (void* q : {(void*)r.l2g, (void*)r.cell_ids, (void*)r.cells, (void*)r.nodes, (void*)r.bnd, (void*)r.node_owner})
            if (q) (void)hipFree(q);
        r = RankMeshDev{};
    }
    p->ranks.clear();
    for (void* q : {(void*)p->part, (void*)p->node_owner, (void*)p->node_mask})
        if (q) (void)hipFree(q);
    p->part = nullptr, p->node_owner = nullptr, p->node_mask = nullptr, p->world = 0;
}

int dev_partition_build(int M, int N, int64_t nn, int64_t nc, const double* d_nodes, const int32_t* d_cells, const uint8_t* d_bnd, int world, int form, void* stream,
                        DevPartition* out, std::string& err) {
    if (world < 1 || world > 64 || (form != kPartitionRowdist && form != kPartitionElements) || nn < 1 || nc < 1 || nn >= (int64_t(1) << 31) || nc >= (int64_t(1) << 31)) {
        err = "fdapde_partition_build: 1 <= world <= 64, form 0 (row-distributed) or 1 (element partition), a non-empty mesh of fewer than 2^31 nodes / cells";
        return FDAPDE_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nv = M + 1;
    dev_partition_release(out);
    out->world = world, out->form = form, out->M = M, out->N = N, out->n_nodes = nn, out->n_cells = nc;
    (void)hipGetDevice(&out->device);
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&out->part), sizeof(int32_t) * (size_t)nc));
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&out->node_owner), sizeof(int32_t) * (size_t)nn));
    DP_CHK(hipMalloc(reinterpret_cast<void**>(&out->node_mask), sizeof(uint64_t) * (size_t)nn));
    DP_CHK(hipMemsetAsync(out->node_mask, 0, sizeof(uint64_t) * (size_t)nn, st));
    Scratch<double> bb;
    DP_CHK(bb.alloc(12));
    // ---- element partition: Morton chunks of the barycentres
    if (world == 1) {
        DP_CHK(hipMemsetAsync(out->part, 0, sizeof(int32_t) * (size_t)nc, st));
    } else {
        Scratch<double> bary;
        Scratch<uint64_t> key_a, key_b;
        Scratch<int32_t> idx_a, idx_b;
        Scratch<int64_t> d_bounds;
        DP_CHK(bary.alloc((size_t)nc * N));
        DP_CHK(key_a.alloc((size_t)nc));
        DP_CHK(key_b.alloc((size_t)nc));
        DP_CHK(idx_a.alloc((size_t)nc));
        DP_CHK(idx_b.alloc((size_t)nc));
        DP_CHK(d_bounds.alloc((size_t)world + 1));
        hipLaunchKernelGGL(k_part_bary, dim3(grid_of(nc)), dim3(256), 0, st, nc, N, nv, nn, d_nodes, d_cells, bary.p);
        if (int rc = column_bounds(bary.p, nc, N, st, bb.p, err)) return rc;
        hipLaunchKernelGGL(k_part_keys, dim3(grid_of(nc)), dim3(256), 0, st, nc, N, bary.p, bb.p, key_a.p, idx_a.p);
        size_t tmp_bytes = 0;
        const int end_bit = N == 3 ? 63 : 62;
        DP_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key_a.p, key_b.p, idx_a.p, idx_b.p, (int)nc, 0, end_bit, st));
        Scratch<char> tmp;
        DP_CHK(tmp.alloc(tmp_bytes + 16));
        DP_CHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key_a.p, key_b.p, idx_a.p, idx_b.p, (int)nc, 0, end_bit, st));
        std::vector<int64_t> bounds((size_t)world + 1);
        const double step = (double)nc / (double)world;   // numpy.linspace(0, nc, world + 1): arange * step, the last one set to the stop
        for (int r = 0; r < world; ++r) bounds[(size_t)r] = (int64_t)((double)r * step);
        bounds[(size_t)world] = nc;
        DP_CHK(hipMemcpyAsync(d_bounds.p, bounds.data(), sizeof(int64_t) * bounds.size(), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_part_assign, dim3(grid_of(nc)), dim3(256), 0, st, nc, world, d_bounds.p, idx_b.p, out->part);
        DP_CHK(hipGetLastError());
        DP_CHK(hipStreamSynchronize(st));   // (the scratch buffers and `bounds` die here)
    }
    // ---- node owners
    Scratch<int32_t> lo, hi;
    DP_CHK(lo.alloc((size_t)nn));
    DP_CHK(hi.alloc((size_t)nn));
    hipLaunchKernelGGL(k_part_fill_i32, dim3(grid_of(nn)), dim3(256), 0, st, nn, (int32_t)0x7fffffff, lo.p);
    hipLaunchKernelGGL(k_part_fill_i32, dim3(grid_of(nn)), dim3(256), 0, st, nn, (int32_t)-1, hi.p);
    hipLaunchKernelGGL(k_part_node_minmax, dim3(grid_of(nc)), dim3(256), 0, st, nc, nv, d_cells, out->part, lo.p, hi.p);
    if (int rc = column_bounds(d_nodes, nn, N, st, bb.p + 6, err)) return rc;
    hipLaunchKernelGGL(k_part_owner, dim3(grid_of(nn)), dim3(256), 0, st, nn, N, form, d_nodes, bb.p + 6, lo.p, hi.p, out->node_owner);
    DP_CHK(hipGetLastError());
    // ---- sub-meshes, rank after rank on the one stream
    Scratch<int32_t> cflag, cpos, nflag, npos;
    DP_CHK(cflag.alloc((size_t)nc + 1));
    DP_CHK(cpos.alloc((size_t)nc + 1));
    DP_CHK(nflag.alloc((size_t)nn + 1));
    DP_CHK(npos.alloc((size_t)nn + 1));
    size_t scan_bytes_c = 0, scan_bytes_n = 0;
    DP_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes_c, cflag.p, cpos.p, (int)nc + 1, st));
    DP_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes_n, nflag.p, npos.p, (int)nn + 1, st));
    Scratch<char> scan_tmp;
    const size_t scan_bytes = std::max(scan_bytes_c, scan_bytes_n) + 16;
    DP_CHK(scan_tmp.alloc(scan_bytes));
    out->ranks.assign((size_t)world, RankMeshDev{});
    for (int r = 0; r < world; ++r) {
        RankMeshDev& R = out->ranks[(size_t)r];
        // (the entry one past the end is zero: the exclusive scan leaves the total there)
        DP_CHK(hipMemsetAsync(cflag.p + nc, 0, sizeof(int32_t), st));
        DP_CHK(hipMemsetAsync(nflag.p, 0, sizeof(int32_t) * ((size_t)nn + 1), st));
        hipLaunchKernelGGL(k_part_flag_cells, dim3(grid_of(nc)), dim3(256), 0, st, nc, nv, form, r, d_cells, out->part, out->node_owner, cflag.p);
        hipLaunchKernelGGL(k_part_flag_nodes, dim3(grid_of(nc)), dim3(256), 0, st, nc, nv, d_cells, cflag.p, nflag.p);
        size_t sb = scan_bytes;
        DP_CHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, sb, cflag.p, cpos.p, (int)nc + 1, st));
        sb = scan_bytes;
        DP_CHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, sb, nflag.p, npos.p, (int)nn + 1, st));
        int32_t totals[2] = {0, 0};
        DP_CHK(hipMemcpyAsync(&totals[0], cpos.p + nc, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        DP_CHK(hipMemcpyAsync(&totals[1], npos.p + nn, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        DP_CHK(hipStreamSynchronize(st));
        R.n_cells = totals[0], R.n_nodes = totals[1];
        const size_t lc = (size_t)std::max<int64_t>(R.n_cells, 1), ln = (size_t)std::max<int64_t>(R.n_nodes, 1);
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&R.cell_ids), sizeof(int32_t) * lc));
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&R.cells), sizeof(int32_t) * lc * (size_t)nv));
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&R.l2g), sizeof(int32_t) * ln));
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&R.nodes), sizeof(double) * ln * (size_t)N));
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&R.bnd), sizeof(uint8_t) * ln));
        DP_CHK(hipMalloc(reinterpret_cast<void**>(&R.node_owner), sizeof(int32_t) * ln));
        hipLaunchKernelGGL(k_part_compact_cells, dim3(grid_of(nc)), dim3(256), 0, st, nc, nv, d_cells, cflag.p, cpos.p, npos.p, R.cell_ids, R.cells);
        hipLaunchKernelGGL(k_part_compact_nodes, dim3(grid_of(nn)), dim3(256), 0, st, nn, N, R.n_nodes, uint64_t(1) << r, d_nodes, d_bnd, out->node_owner, nflag.p, npos.p,
                           R.l2g, R.nodes, R.bnd, R.node_owner, out->node_mask);
        DP_CHK(hipGetLastError());
    }
    DP_CHK(hipStreamSynchronize(st));
    return FDAPDE_OK;
}

void dev_partition_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_part_bary));
    (void)hipGetLastError();
}

}   // namespace fdapde_hip
