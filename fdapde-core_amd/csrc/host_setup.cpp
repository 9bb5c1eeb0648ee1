// host_setup.cpp -- integer / index work done once per discrete space, on the host, off the timed path:
//   * DOF enumeration in the reference's numbering         (LagrangianBasis::enumerate_dofs, basis/lagrangian_basis.h:94-136;
//                                                            edge ids of Triangulation<2,N> / <3,3>, geometry/triangulation.h:150-193, 348-377)
//   * DOF coordinates                                       (LagrangianBasis::dofs_coords, basis/lagrangian_basis.h:159-183)
//   * CSR sparsity pattern of the assembled operators       (what setFromTriplets + makeCompressed produce, fem_assembler.h:112-113)
//   * the device-side layout: locality (Morton) renumbering of nodes / DOFs / cells, row-owner adjacency in
//     sliced-ELL form with per-visit column slots, SpMV row blocks, element colouring.
// Nothing here touches floating-point results except copying coordinates; all of it is checked bit-exactly against the
// oracle by tests/test_host_logic.py on a host-only context.
//
// The edge numbering is reproduced with sorts instead of the reference's hash maps: an edge's id is the rank of its
// first occurrence in the reference's traversal order (cells ascending x local pattern), which a sort by
// (edge key, occurrence index) yields directly and in parallel-friendly form.
#include <algorithm>
#include <atomic>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <thread>

#include "internal.h"

namespace fdapde_hip {

namespace {

unsigned hw_threads() {
    static const unsigned cached = [] {
        if (const char* e = std::getenv("FDAPDE_THREADS")) {   // set-up threads (default: hardware concurrency, at most 64)
            const int v = std::atoi(e);
            if (v >= 1) return (unsigned)std::min(v, 256);
        }
        const unsigned n = std::thread::hardware_concurrency();
        return n == 0 ? 4u : std::min(n, 64u);
    }();
    return cached;
}

// run fn(begin, end, tid) over [0, n) in contiguous chunks
template <typename F> void parallel_for(int64_t n, F&& fn, int64_t grain = 4096) {
    unsigned nt = hw_threads();
    if (n < grain * 2 || nt == 1) {
        fn(int64_t(0), n, 0u);
        return;
    }
    nt = (unsigned)std::min<int64_t>(nt, (n + grain - 1) / grain);
    std::vector<std::thread> th;
    const int64_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        int64_t b = t * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        th.emplace_back([=, &fn] { fn(b, e, t); });
    }
    for (auto& x : th) x.join();
}
unsigned n_chunks(int64_t n, int64_t grain = 4096) {
    unsigned nt = hw_threads();
    if (n < grain * 2 || nt == 1) return 1;
    return (unsigned)std::min<int64_t>(nt, (n + grain - 1) / grain);
}

// LSD radix sort of (key, value) pairs by 64-bit key, 11-bit digits; stable; every pass is parallel (per-thread digit
// histograms over contiguous chunks, offsets in digit-major / thread-minor order, in-order scatter)
void radix_sort_pairs(hvec<uint64_t>& key, hvec<int32_t>& val) {
    const size_t n = key.size();
    if (n < 2) return;
    constexpr int B = 11;
    constexpr size_t R = size_t(1) << B;
    const unsigned nt = std::min(64u, n_chunks((int64_t)n, 1 << 16));
    const size_t chunk = (n + nt - 1) / nt;
    std::vector<uint64_t> ors(nt, 0);
    parallel_for((int64_t)nt, [&](int64_t tb, int64_t te, unsigned) {
        for (int64_t t = tb; t < te; ++t) {
            uint64_t o = 0;
            for (size_t i = (size_t)t * chunk, e = std::min(n, ((size_t)t + 1) * chunk); i < e; ++i) o |= key[i];
            ors[(size_t)t] = o;
        }
    }, 1);
    uint64_t all_or = 0;
    for (uint64_t o : ors) all_or |= o;
    hvec<uint64_t> k2(n);   // uninitialised: first touched by the scatter threads
    hvec<int32_t> v2(n);
    std::vector<size_t> cnt((size_t)nt * R);
    for (int shift = 0; shift < 64 && (all_or >> shift) != 0; shift += B) {
        parallel_for((int64_t)nt, [&](int64_t tb, int64_t te, unsigned) {
            for (int64_t t = tb; t < te; ++t) {
                size_t* c = &cnt[(size_t)t * R];
                std::fill(c, c + R, 0);
                for (size_t i = (size_t)t * chunk, e = std::min(n, ((size_t)t + 1) * chunk); i < e; ++i) ++c[(key[i] >> shift) & (R - 1)];
            }
        }, 1);
        size_t run = 0;
        for (size_t d = 0; d < R; ++d)
            for (unsigned t = 0; t < nt; ++t) {
                const size_t c = cnt[(size_t)t * R + d];
                cnt[(size_t)t * R + d] = run, run += c;
            }
        parallel_for((int64_t)nt, [&](int64_t tb, int64_t te, unsigned) {
            for (int64_t t = tb; t < te; ++t) {
                size_t* c = &cnt[(size_t)t * R];
                for (size_t i = (size_t)t * chunk, e = std::min(n, ((size_t)t + 1) * chunk); i < e; ++i) {
                    const size_t p = c[(key[i] >> shift) & (R - 1)]++;
                    k2[p] = key[i], v2[p] = val[i];
                }
            }
        }, 1);
        key.swap(k2), val.swap(v2);
    }
}

// Parallel comparison sort (sample sort): splitters from a sorted sample, elements bucketed by binary search, every bucket
// sorted by its own thread.  `less` must be a strict total order for the result to equal std::sort's.
template <typename T, typename Less> void parallel_sort(std::vector<T>& v, Less less) {
    const size_t n = v.size();
    const unsigned nt = std::min(64u, n_chunks((int64_t)n, 1 << 16));
    if (nt < 2) {
        std::sort(v.begin(), v.end(), less);
        return;
    }
    const size_t n_sample = (size_t)nt * 64;
    std::vector<T> sample(n_sample);
    for (size_t i = 0; i < n_sample; ++i) sample[i] = v[(size_t)((double)i * (double)n / (double)n_sample)];
    std::sort(sample.begin(), sample.end(), less);
    std::vector<T> split(nt - 1);
    for (unsigned b = 1; b < nt; ++b) split[b - 1] = sample[(size_t)b * 64];
    const size_t chunk = (n + nt - 1) / nt;
    std::vector<size_t> cnt((size_t)nt * nt, 0);   // [thread][bucket]
    std::vector<uint8_t> bucket(n);
    parallel_for((int64_t)nt, [&](int64_t tb, int64_t te, unsigned) {
        for (int64_t t = tb; t < te; ++t)
            for (size_t i = (size_t)t * chunk, e = std::min(n, ((size_t)t + 1) * chunk); i < e; ++i) {
                const unsigned b = (unsigned)(std::upper_bound(split.begin(), split.end(), v[i], less) - split.begin());
                bucket[i] = (uint8_t)b, ++cnt[(size_t)t * nt + b];
            }
    }, 1);
    std::vector<size_t> start(nt + 1, 0);
    size_t run = 0;
    for (unsigned b = 0; b < nt; ++b) {
        start[b] = run;
        for (unsigned t = 0; t < nt; ++t) {
            const size_t c = cnt[(size_t)t * nt + b];
            cnt[(size_t)t * nt + b] = run, run += c;
        }
    }
    start[nt] = run;
    std::vector<T> out(n);
    parallel_for((int64_t)nt, [&](int64_t tb, int64_t te, unsigned) {
        for (int64_t t = tb; t < te; ++t)
            for (size_t i = (size_t)t * chunk, e = std::min(n, ((size_t)t + 1) * chunk); i < e; ++i) out[cnt[(size_t)t * nt + bucket[i]]++] = v[i];
    }, 1);
    parallel_for((int64_t)nt, [&](int64_t bb, int64_t be, unsigned) {
        for (int64_t b = bb; b < be; ++b) std::sort(out.begin() + (int64_t)start[(size_t)b], out.begin() + (int64_t)start[(size_t)b + 1], less);
    }, 1);
    v.swap(out);
}

inline uint64_t spread3(uint64_t x) {  // 21 bits -> every third bit
    x &= 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
inline uint64_t spread2(uint64_t x) {  // 31 bits -> every second bit
    x &= 0x7fffffff;
    x = (x | x << 16) & 0x0000ffff0000ffffull;
    x = (x | x << 8) & 0x00ff00ff00ff00ffull;
    x = (x | x << 4) & 0x0f0f0f0f0f0f0f0full;
    x = (x | x << 2) & 0x3333333333333333ull;
    x = (x | x << 1) & 0x5555555555555555ull;
    return x;
}

// small id set / id -> value map for the per-row and per-block unions of the set-up
struct Probe {   // id -> value, power-of-two open addressing; clear() resets exactly the slots in use
    hvec<int32_t> key, val;
    std::vector<uint32_t> used;
    uint32_t mask = 0;
    void reserve(size_t n_ids) {
        size_t cap = 64;
        while (cap < 2 * n_ids) cap <<= 1;
        if (cap > key.size()) key.assign(cap, -1), val.resize(cap);
        mask = (uint32_t)key.size() - 1;
    }
    uint32_t slot(int32_t id) const {
        uint32_t h = ((uint32_t)id * 2654435761u) & mask;
        while (key[h] != -1 && key[h] != id) h = (h + 1) & mask;
        return h;
    }
    bool insert(int32_t id) {
        const uint32_t h = slot(id);
        if (key[h] == id) return false;
        key[h] = id, used.push_back(h);
        return true;
    }
    void clear() {
        for (uint32_t h : used) key[h] = -1;
        used.clear();
    }
};

// permutation that sorts points (column-major n x N) along the Morton curve; returns i2e (new -> old).  bits = resolution per
// axis (0: the maximum, 21 in 3-D / 31 in 2-D); points with equal keys keep their input order.  The radix sort runs one pass per
// 11 key bits, so a coarse key (cells: only locality matters, not the numbering) halves its cost.
hvec<int32_t> morton_order(int N, int64_t n, const double* pts_colmajor, int bits = 0) {
    hvec<int32_t> idx((size_t)n);
    parallel_for(n, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t i = b; i < e; ++i) idx[(size_t)i] = (int32_t)i;
    }, 1 << 16);
    if (n < 2) return idx;
    double lo[3], hi[3];
    {
        const unsigned nt = n_chunks(n, 1 << 16);
        std::vector<double> plo((size_t)nt * 3), phi((size_t)nt * 3);
        const int64_t chunk = (n + nt - 1) / nt;
        parallel_for((int64_t)nt, [&](int64_t tb, int64_t te, unsigned) {
            for (int64_t t = tb; t < te; ++t)
                for (int d = 0; d < N; ++d) {
                    const int64_t b = t * chunk, e = std::min(n, b + chunk);
                    double l = pts_colmajor[(int64_t)d * n + std::min(b, n - 1)], h = l;
                    for (int64_t i = b; i < e; ++i) {
                        const double v = pts_colmajor[(int64_t)d * n + i];
                        l = std::min(l, v), h = std::max(h, v);
                    }
                    plo[(size_t)t * 3 + d] = l, phi[(size_t)t * 3 + d] = h;
                }
        }, 1);
        for (int d = 0; d < N; ++d) {
            lo[d] = plo[(size_t)d], hi[d] = phi[(size_t)d];
            for (unsigned t = 1; t < nt; ++t) lo[d] = std::min(lo[d], plo[(size_t)t * 3 + d]), hi[d] = std::max(hi[d], phi[(size_t)t * 3 + d]);
        }
    }
    const int max_bits = N == 3 ? 21 : 31;
    const double span = (double)((uint64_t(1) << (bits > 0 && bits < max_bits ? bits : max_bits)) - 1);
    hvec<uint64_t> key((size_t)n);
    parallel_for(n, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t i = b; i < e; ++i) {
            uint64_t q[3] = {0, 0, 0};
            for (int d = 0; d < N; ++d) {
                double w = hi[d] > lo[d] ? (pts_colmajor[(int64_t)d * n + i] - lo[d]) / (hi[d] - lo[d]) : 0.0;
                q[d] = (uint64_t)std::llround(std::min(1.0, std::max(0.0, w)) * span);
            }
            key[(size_t)i] = N == 3 ? (spread3(q[0]) | spread3(q[1]) << 1 | spread3(q[2]) << 2)
                                    : (spread2(q[0]) | spread2(q[1]) << 1);
        }
    });
    radix_sort_pairs(key, idx);
    return idx;
}

hvec<int32_t> invert(const hvec<int32_t>& p) {
    hvec<int32_t> inv(p.size());
    parallel_for((int64_t)p.size(), [&](int64_t b, int64_t e, unsigned) {
        for (int64_t i = b; i < e; ++i) inv[(size_t)p[(size_t)i]] = (int32_t)i;
    }, 1 << 16);
    return inv;
}

// local vertex pairs in the reference's enumeration orders
constexpr int COMB23[3][2] = {{0, 1}, {0, 2}, {1, 2}};             // combinations<2,3>, utils/combinatorics.h:37-51
constexpr int COMB34[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};   // combinations<3,4>
// local slot of the edge joining local vertices (a,b): position of its midpoint in ReferenceElement<M,2>::nodes
int edge_slot(int M, int a, int b) {
    if (a > b) std::swap(a, b);
    if (M == 2) return 3 + (a == 0 ? (b == 1 ? 0 : 1) : 2);
    constexpr int S[4][4] = {{-1, 6, 5, 9}, {6, -1, 4, 7}, {5, 4, -1, 8}, {9, 7, 8, -1}};
    return S[a][b];
}

// ---------------------------------------------------------------------------------------------------------------
// edge ids in first-seen order.
// 2-D occurrence index: cell*3 + j, j over combinations<2,3>.
// 3-D occurrence index: cell*12 + face*3 + k, faces over combinations<3,4> with sorted nodes, k over combinations<2,3>
//      of the sorted face (triangulation.h:348-377).  The first occurrence of an edge is always inside a newly seen
//      face (were the face seen before, the edge would have an earlier occurrence), so "min occurrence" == "first seen".
// ---------------------------------------------------------------------------------------------------------------
// Edges are bucketed by their smaller node (counted and scattered by all threads), every bucket is sorted by (larger node,
// cell * epc + local pair), so that the head of a run of equal edges is its occurrence in the lowest cell; the unique edges are
// then ranked by their first occurrence index with one radix sort.  first_cell[id] = the lowest cell containing edge id.
int enumerate_edges(HostSpace& hs, hvec<int32_t>& cell_edge /* n_cells x (3|6), by local pair order */,
                    std::vector<uint8_t>& edge_bnd, hvec<int32_t>& first_cell, std::string& err) {
    const int M = hs.M, nv = M + 1, epc = M == 2 ? 3 : 6;
    const int64_t nc = hs.n_cells, nn = hs.n_nodes, n_ent = nc * epc;
    constexpr int P3[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    if (n_ent > INT32_MAX) {
        err = "too many cells for the order-2 edge enumeration (cells x edges per cell exceeds int32)";
        return FDAPDE_EUNSUPPORTED;
    }
    auto pair_of = [&](const int32_t* cv, int p, uint32_t& a, uint32_t& b) {
        a = (uint32_t)cv[M == 2 ? COMB23[p][0] : P3[p][0]], b = (uint32_t)cv[M == 2 ? COMB23[p][1] : P3[p][1]];
        if (a > b) std::swap(a, b);
    };
    // occurrence index of local pair p inside its cell: 2-D p itself; 3-D the first (face, pair of the sorted face) that is this edge
    auto occ_in_cell = [&](const int32_t* cv, int p) -> int {
        if (M == 2) return p;
        uint32_t a, b;
        pair_of(cv, p, a, b);
        for (int f = 0; f < 4; ++f) {
            std::array<int32_t, 3> face = {cv[COMB34[f][0]], cv[COMB34[f][1]], cv[COMB34[f][2]]};
            std::sort(face.begin(), face.end());
            for (int k = 0; k < 3; ++k)
                if ((uint32_t)face[COMB23[k][0]] == a && (uint32_t)face[COMB23[k][1]] == b) return f * 3 + k;
        }
        return 11;   // not reached: every pair of a tetrahedron lies in two of its faces
    };
    std::unique_ptr<std::atomic<int32_t>[]> cnt(new std::atomic<int32_t>[(size_t)nn]);
    auto zero_cnt = [&] {
        parallel_for(nn, [&](int64_t b0, int64_t b1, unsigned) {
            for (int64_t i = b0; i < b1; ++i) cnt[(size_t)i].store(0, std::memory_order_relaxed);
        }, 1 << 16);
    };
    zero_cnt();
    parallel_for(nc, [&](int64_t b0, int64_t b1, unsigned) {
        for (int64_t c = b0; c < b1; ++c)
            for (int p = 0; p < epc; ++p) {
                uint32_t a, b;
                pair_of(&hs.cells[(size_t)c * nv], p, a, b);
                cnt[a].fetch_add(1, std::memory_order_relaxed);
            }
    }, 1 << 14);
    std::vector<int64_t> ptr((size_t)nn + 1, 0);
    for (int64_t i = 0; i < nn; ++i) ptr[(size_t)i + 1] = ptr[(size_t)i] + cnt[(size_t)i].load(std::memory_order_relaxed);
    zero_cnt();
    hvec<uint64_t> ent((size_t)n_ent);   // (larger node << 32) | (cell * epc + local pair)
    parallel_for(nc, [&](int64_t b0, int64_t b1, unsigned) {
        for (int64_t c = b0; c < b1; ++c)
            for (int p = 0; p < epc; ++p) {
                uint32_t a, b;
                pair_of(&hs.cells[(size_t)c * nv], p, a, b);
                ent[(size_t)(ptr[a] + cnt[a].fetch_add(1, std::memory_order_relaxed))] = (uint64_t)b << 32 | (uint32_t)(c * epc + p);
            }
    }, 1 << 14);
    std::vector<int64_t> uptr((size_t)nn + 1, 0);
    parallel_for(nn, [&](int64_t b0, int64_t b1, unsigned) {
        for (int64_t a = b0; a < b1; ++a) {
            std::sort(ent.begin() + ptr[(size_t)a], ent.begin() + ptr[(size_t)a + 1]);
            int64_t u = 0;
            for (int64_t i = ptr[(size_t)a]; i < ptr[(size_t)a + 1]; ++i)
                u += i == ptr[(size_t)a] || (ent[(size_t)i] >> 32) != (ent[(size_t)i - 1] >> 32);
            uptr[(size_t)a + 1] = u;
        }
    }, 1 << 12);
    for (int64_t a = 0; a < nn; ++a) uptr[(size_t)a + 1] += uptr[(size_t)a];
    const int64_t ne = uptr[(size_t)nn];
    if (nn + ne > INT32_MAX) {
        err = "DOF count exceeds int32";
        return FDAPDE_EUNSUPPORTED;
    }
    hs.n_edges = ne;
    // unique edge u (in (smaller, larger) node order): position of its run head, run length, first occurrence index
    hvec<int64_t> head((size_t)ne);
    hvec<int32_t> mult((size_t)ne), order((size_t)ne);
    hvec<uint64_t> first((size_t)ne);
    parallel_for(nn, [&](int64_t b0, int64_t b1, unsigned) {
        for (int64_t a = b0; a < b1; ++a) {
            int64_t u = uptr[(size_t)a] - 1;
            for (int64_t i = ptr[(size_t)a]; i < ptr[(size_t)a + 1]; ++i) {
                if (i == ptr[(size_t)a] || (ent[(size_t)i] >> 32) != (ent[(size_t)i - 1] >> 32)) {
                    ++u;
                    const int64_t val = (int64_t)(uint32_t)ent[(size_t)i], c = val / epc;
                    head[(size_t)u] = i, mult[(size_t)u] = 0, order[(size_t)u] = (int32_t)u;
                    first[(size_t)u] = (uint64_t)(c * (M == 2 ? 3 : 12) + occ_in_cell(&hs.cells[(size_t)c * nv], (int)(val % epc)));
                }
                ++mult[(size_t)u];
            }
        }
    }, 1 << 12);
    radix_sort_pairs(first, order);   // order[id] = u: edge ids in first-seen order
    edge_bnd.resize((size_t)ne);
    first_cell.resize((size_t)ne);
    cell_edge.resize((size_t)n_ent);
    parallel_for(ne, [&](int64_t b0, int64_t b1, unsigned) {
        for (int64_t id = b0; id < b1; ++id) {
            const int64_t u = order[(size_t)id], h = head[(size_t)u];
            first_cell[(size_t)id] = (int32_t)(first[(size_t)id] / (M == 2 ? 3 : 12));
            if (M == 2) {
                edge_bnd[(size_t)id] = mult[(size_t)u] == 1;   // seen by exactly one cell (triangulation.h:177,187)
            } else {
                const int64_t val = (int64_t)(uint32_t)ent[(size_t)h], c = val / epc;
                uint32_t a, b;
                pair_of(&hs.cells[(size_t)c * nv], (int)(val % epc), a, b);
                edge_bnd[(size_t)id] = hs.node_bnd[a] && hs.node_bnd[b];   // triangulation.h:371
            }
            for (int64_t i = h; i < h + mult[(size_t)u]; ++i) cell_edge[(size_t)(uint32_t)ent[(size_t)i]] = (int32_t)id;
        }
    }, 1 << 12);
    return FDAPDE_OK;
}

}  // namespace

int host_set_mesh(HostSpace& hs, int M, int N, int64_t n_nodes, const double* nodes, int64_t n_cells,
                  const int32_t* cells, const uint8_t* bnd, std::string& err) {
    if (!((M == 2 && N == 2) || (M == 3 && N == 3))) {
        err = "only Triangulation<2,2> and Triangulation<3,3> are on the accelerated path";
        return FDAPDE_EUNSUPPORTED;
    }
    if (n_nodes <= 0 || n_cells <= 0 || !nodes || !cells || !bnd) {
        err = "empty mesh or null pointer";
        return FDAPDE_EINVAL;
    }
    if (n_nodes > INT32_MAX || n_cells >= (int64_t(1) << 27)) {
        err = "mesh too large for int32 indices (n_nodes <= 2^31-1, n_cells < 2^27)";
        return FDAPDE_EUNSUPPORTED;
    }
    for (int64_t i = 0; i < n_cells * (M + 1); ++i)
        if (cells[i] < 0 || cells[i] >= n_nodes) {
            err = "cell references a node id out of range";
            return FDAPDE_EINVAL;
        }
    hs = HostSpace{};
    hs.M = M, hs.N = N, hs.n_nodes = n_nodes, hs.n_cells = n_cells;
    hs.nodes.resize((size_t)(n_nodes * N)), hs.cells.resize((size_t)(n_cells * (M + 1)));
    parallel_for(n_nodes * N, [&](int64_t b, int64_t e, unsigned) { std::memcpy(&hs.nodes[(size_t)b], nodes + b, sizeof(double) * (size_t)(e - b)); }, 1 << 16);
    parallel_for(n_cells * (M + 1), [&](int64_t b, int64_t e, unsigned) { std::memcpy(&hs.cells[(size_t)b], cells + b, sizeof(int32_t) * (size_t)(e - b)); }, 1 << 16);
    hs.node_bnd.resize((size_t)n_nodes);
    for (int64_t i = 0; i < n_nodes; ++i) hs.node_bnd[(size_t)i] = bnd[i] ? 1 : 0;
    return FDAPDE_OK;
}

int host_build_space(HostSpace& hs, int order, std::string& err, int stop_after) {
    auto t0 = std::chrono::steady_clock::now();
    auto t_phase = t0;
    const bool dbg_time = std::getenv("FDAPDE_DEBUG_SETUP") != nullptr;
    auto phase = [&](const char* name) {   // FDAPDE_DEBUG_SETUP: wall time of every set-up phase
        if (!dbg_time) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "setup %-28s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(now - t_phase).count());
        t_phase = now;
    };
    if (hs.n_cells == 0) {
        err = "mesh not uploaded";
        return FDAPDE_ENOTINIT;
    }
    if (order != 1 && order != 2) {
        err = "fem_order must be 1 or 2 (LagrangianBasis::enumerate_dofs requires Order <= 2)";
        return FDAPDE_EUNSUPPORTED;
    }
    if (hs.n_cells >= (int64_t(1) << 27)) {   // visits are packed as cell * 16 + local index in 32 bits
        err = "more than 2^27 cells per context: partition the mesh (fdapde_halo_setup)";
        return FDAPDE_EUNSUPPORTED;
    }
    const int M = hs.M, N = hs.N, nv = M + 1;
    const int nb = n_basis_of(M, order);
    hs.order = order, hs.nb = nb, hs.nq = n_quadrature_of(M, order);
    const int64_t nc = hs.n_cells, nn = hs.n_nodes;
    if (stop_after == 2) return FDAPDE_OK;   // sizes only: the DOF table too is built on the device (dev_topology.hip)

    // ---- DOF table, boundary DOFs (reference numbering) ------------------------------------------------------
    hs.dofs.resize((size_t)nc * nb);   // vertex slots here, edge slots (order 2) below: every slot is written
    parallel_for(nc, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t c = b; c < e; ++c)
            for (int v = 0; v < nv; ++v) hs.dofs[(size_t)c * nb + v] = hs.cells[(size_t)c * nv + v];
    }, 1 << 14);
    hs.dof_bnd.assign(hs.node_bnd.begin(), hs.node_bnd.end());
    hs.n_edges = 0;
    hvec<int32_t> edge_first_cell;   // order 2: lowest cell containing each edge
    if (order == 2) {
        hvec<int32_t> cell_edge;
        std::vector<uint8_t> edge_bnd;
        int rc = enumerate_edges(hs, cell_edge, edge_bnd, edge_first_cell, err);
        if (rc) return rc;
        const int epc = M == 2 ? 3 : 6;
        constexpr int P3[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
        parallel_for(nc, [&](int64_t c0, int64_t c1, unsigned) {
            for (int64_t c = c0; c < c1; ++c)
                for (int k = 0; k < epc; ++k) {
                    int a = M == 2 ? COMB23[k][0] : P3[k][0], b = M == 2 ? COMB23[k][1] : P3[k][1];
                    hs.dofs[(size_t)c * nb + edge_slot(M, a, b)] = (int32_t)nn + cell_edge[(size_t)c * epc + k];
                }
        }, 1 << 14);
        hs.dof_bnd.insert(hs.dof_bnd.end(), edge_bnd.begin(), edge_bnd.end());
    }
    hs.n_dofs = nn + hs.n_edges;
    const int64_t nd = hs.n_dofs;

    phase("dof table");
    // ---- DOF coordinates: vertices, then J * ref + x0 from the first visiting cell (lagrangian_basis.h:159-183)
    BasisTables tb;
    build_basis_tables(M, order, &tb);
    hs.dof_coords.resize((size_t)nd * N);   // vertices here, edge DOFs (order 2) below
    for (int d = 0; d < N; ++d)
        parallel_for(nn, [&](int64_t b, int64_t e, unsigned) {
            std::memcpy(&hs.dof_coords[(size_t)d * nd + b], &hs.nodes[(size_t)d * nn + b], sizeof(double) * (size_t)(e - b));
        }, 1 << 16);
    if (order == 2) {   // an edge DOF takes its coordinates from the FIRST cell that visits it (the reference's loop order)
        parallel_for(hs.n_edges, [&](int64_t e0, int64_t e1, unsigned) {
            for (int64_t e = e0; e < e1; ++e) {
                const int64_t c = edge_first_cell[(size_t)e];
                const int32_t dof = (int32_t)(nn + e);
                int j = nv;
                while (j < nb - 1 && hs.dofs[(size_t)c * nb + j] != dof) ++j;
                const int32_t v0 = hs.cells[(size_t)c * nv];
                for (int d = 0; d < N; ++d) {
                    double x0 = hs.nodes[(size_t)d * nn + v0], acc = 0;
                    for (int k = 0; k < M; ++k)
                        acc += (hs.nodes[(size_t)d * nn + hs.cells[(size_t)c * nv + k + 1]] - x0) * tb.refnodes[j * M + k];
                    hs.dof_coords[(size_t)d * nd + dof] = acc + x0;
                }
            }
        }, 1 << 12);
    }

    phase("dof coordinates");
    if (stop_after == 1) return FDAPDE_OK;   // the device builder (dev_setup.hip) takes it from here
    // ---- locality numbering --------------------------------------------------------------------------------
    hs.node_i2e = morton_order(N, nn, hs.nodes.data());
    hs.node_e2i = invert(hs.node_i2e);
    phase("  numbering: nodes");
    if (order == 1) {
        hs.dof_i2e = hs.node_i2e, hs.dof_e2i = hs.node_e2i;
    } else {
        hs.dof_i2e = morton_order(N, nd, hs.dof_coords.data());
        hs.dof_e2i = invert(hs.dof_i2e);
    }
    {
        hvec<double> bary((size_t)nc * N);
        parallel_for(nc, [&](int64_t b, int64_t e, unsigned) {
            for (int64_t c = b; c < e; ++c)
                for (int d = 0; d < N; ++d) {
                    double s = 0;
                    for (int v = 0; v < nv; ++v) s += hs.nodes[(size_t)d * nn + hs.cells[(size_t)c * nv + v]];
                    bary[(size_t)d * nc + c] = s / nv;
                }
        });
        phase("  numbering: dofs + barycentres");
        // order 1: a coarse key (cells only need locality).  Order 2 keeps the full key: its sort is short anyway, and the visit
        // order decides the summation order of the assembled entries -- harmless for CG, but BiCGStab's iteration count on C5
        // moves between 712 and 797 with such last-bit differences, and the committed profile was taken with this order
        hs.cell_i2e = morton_order(N, nc, bary.data(), order == 1 ? (N == 3 ? 11 : 16) : 0);
        hs.cell_e2i = invert(hs.cell_i2e);
    }
    phase("  numbering: cells");
    const int NP = N == 2 ? 2 : 4;
    hs.vcoords_i.resize((size_t)nn * NP);
    parallel_for(nn, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t i = b; i < e; ++i)
            for (int d = 0; d < NP; ++d) hs.vcoords_i[(size_t)i * NP + d] = d < N ? hs.nodes[(size_t)d * nn + hs.node_i2e[(size_t)i]] : 0.0;
    }, 1 << 16);
    hs.dof_bnd_i.resize((size_t)nd);
    parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t i = b; i < e; ++i) hs.dof_bnd_i[(size_t)i] = hs.dof_bnd[(size_t)hs.dof_i2e[(size_t)i]];
    }, 1 << 16);
    hs.cverts_i.resize((size_t)nc * nv), hs.cdofs_i.resize((size_t)nc * nb);
    parallel_for(nc, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t ci = b; ci < e; ++ci) {
            int64_t ce = hs.cell_i2e[(size_t)ci];
            for (int v = 0; v < nv; ++v) hs.cverts_i[(size_t)ci * nv + v] = hs.node_e2i[(size_t)hs.cells[(size_t)ce * nv + v]];
            for (int j = 0; j < nb; ++j) hs.cdofs_i[(size_t)ci * nb + j] = hs.dof_e2i[(size_t)hs.dofs[(size_t)ce * nb + j]];
        }
    });

    phase("locality numbering");
    // ---- row-owner adjacency: DOF -> (cell, local index), cells ascending -----------------------------------
    // counted and scattered by all threads with relaxed atomic increments (rows are short and contention is low); the scatter
    // leaves a row's visits in arrival order, so every row is sorted afterwards: the result is the serial counting sort's
    std::vector<int64_t> vptr((size_t)nd + 1, 0);
    hvec<int32_t> vis((size_t)(nc * nb));
    {
        std::unique_ptr<std::atomic<int32_t>[]> cnt(new std::atomic<int32_t>[(size_t)nd]);
        parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
            for (int64_t i = b; i < e; ++i) cnt[(size_t)i].store(0, std::memory_order_relaxed);
        }, 1 << 16);
        parallel_for(nc * nb, [&](int64_t b, int64_t e, unsigned) {
            for (int64_t k = b; k < e; ++k) cnt[(size_t)hs.cdofs_i[(size_t)k]].fetch_add(1, std::memory_order_relaxed);
        }, 1 << 16);
        for (int64_t i = 0; i < nd; ++i) vptr[(size_t)i + 1] = vptr[(size_t)i] + cnt[(size_t)i].load(std::memory_order_relaxed);
        parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
            for (int64_t i = b; i < e; ++i) cnt[(size_t)i].store(0, std::memory_order_relaxed);
        }, 1 << 16);
        parallel_for(nc, [&](int64_t b, int64_t e, unsigned) {
            for (int64_t c = b; c < e; ++c)
                for (int j = 0; j < nb; ++j) {
                    const int32_t dof = hs.cdofs_i[(size_t)c * nb + j];
                    vis[(size_t)(vptr[(size_t)dof] + cnt[(size_t)dof].fetch_add(1, std::memory_order_relaxed))] = (int32_t)(c * 16 + j);
                }
        }, 1 << 14);
        parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
            for (int64_t i = b; i < e; ++i) std::sort(vis.begin() + vptr[(size_t)i], vis.begin() + vptr[(size_t)i + 1]);
        }, 1 << 12);
    }

    phase("row-owner adjacency");
    // ---- internal CSR pattern: row = sorted union of the DOFs of the visiting cells ---------------------------
    {
        const unsigned nt = n_chunks(nd, 2048);
        std::vector<hvec<int32_t>> tcols(nt);
        std::vector<int32_t> rowlen((size_t)nd);
        std::vector<int64_t> tbegin(nt, 0);
        parallel_for(nd, [&](int64_t b, int64_t e, unsigned t) {
            tbegin[t] = b;
            std::vector<int32_t> cand;
            Probe seen;   // the DOFs of the visiting cells repeat 4-10 times: only the distinct ones are sorted
            auto& out = tcols[t];
            out.reserve((size_t)(vptr[(size_t)e] - vptr[(size_t)b]) * (size_t)nb);   // upper bound; untouched pages cost nothing
            for (int64_t r = b; r < e; ++r) {
                cand.clear();
                seen.reserve((size_t)(vptr[(size_t)r + 1] - vptr[(size_t)r]) * (size_t)nb);
                for (int64_t k = vptr[(size_t)r]; k < vptr[(size_t)r + 1]; ++k) {
                    const int32_t* cd = &hs.cdofs_i[(size_t)(vis[(size_t)k] >> 4) * nb];
                    for (int j = 0; j < nb; ++j)
                        if (seen.insert(cd[j])) cand.push_back(cd[j]);
                }
                seen.clear();
                std::sort(cand.begin(), cand.end());
                rowlen[(size_t)r] = (int32_t)cand.size();
                out.insert(out.end(), cand.begin(), cand.end());
            }
        }, 2048);
        int64_t total = 0;
        for (auto& v : tcols) total += (int64_t)v.size();
        if (total > INT32_MAX) {
            err = "nnz exceeds int32";
            return FDAPDE_EUNSUPPORTED;
        }
        hs.nnz = total;
        hs.rowptr_i.assign((size_t)nd + 1, 0);
        hs.max_row = 0;
        for (int64_t r = 0; r < nd; ++r) {
            hs.rowptr_i[(size_t)r + 1] = hs.rowptr_i[(size_t)r] + rowlen[(size_t)r];
            hs.max_row = std::max(hs.max_row, rowlen[(size_t)r]);
        }
        hs.colidx_i.resize((size_t)total + 2);   // + 2 zeros: the SpMV's pair loads may touch one entry past a row's end
        hs.colidx_i[(size_t)total] = hs.colidx_i[(size_t)total + 1] = 0;
        parallel_for((int64_t)nt, [&](int64_t tb, int64_t te, unsigned) {
            for (int64_t t = tb; t < te; ++t)
                if (!tcols[(size_t)t].empty())
                    std::memcpy(&hs.colidx_i[(size_t)hs.rowptr_i[(size_t)tbegin[(size_t)t]]], tcols[(size_t)t].data(),
                                sizeof(int32_t) * tcols[(size_t)t].size());
        }, 1);
    }
    for (int64_t r = 0; r < nd; ++r)
        if (hs.rowptr_i[(size_t)r + 1] == hs.rowptr_i[(size_t)r]) {
            err = "a node is not referenced by any cell: its DOF has an empty matrix row (the reference's LU fails on such a mesh)";
            return FDAPDE_EINVAL;
        }
    if (const char* dump = std::getenv("FDAPDE_DEBUG_DUMP")) {   // internal CSR pattern + boundary flags for offline layout studies
        if (FILE* fp = std::fopen(dump, "wb")) {
            const int64_t hdr[2] = {nd, hs.nnz};
            std::fwrite(hdr, sizeof(int64_t), 2, fp);
            std::fwrite(hs.rowptr_i.data(), sizeof(int32_t), (size_t)nd + 1, fp);
            std::fwrite(hs.colidx_i.data(), sizeof(int32_t), (size_t)hs.nnz, fp);
            std::fwrite(hs.dof_bnd_i.data(), 1, (size_t)nd, fp);
            std::fclose(fp);
        }
    }
    if (hs.max_row > 65535 || hs.max_row > kSpmvNnz) {
        err = "row too long for the uint16 slot map / SpMV row block";
        return FDAPDE_EUNSUPPORTED;
    }
    hs.diag_i.resize((size_t)nd);
    parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t r = b; r < e; ++r) {
            const int32_t* beg = &hs.colidx_i[(size_t)hs.rowptr_i[(size_t)r]];
            const int32_t* end = &hs.colidx_i[(size_t)hs.rowptr_i[(size_t)r + 1]];
            hs.diag_i[(size_t)r] = (int32_t)(std::lower_bound(beg, end, (int32_t)r) - hs.colidx_i.data());
        }
    });

    phase("internal CSR pattern");
    // ---- reference-numbering CSR pattern + internal slot -> reference slot ------------------------------------
    hs.rowptr_e.assign((size_t)nd + 1, 0);
    for (int64_t re = 0; re < nd; ++re) {
        int64_t ri = hs.dof_e2i[(size_t)re];
        hs.rowptr_e[(size_t)re + 1] = hs.rowptr_e[(size_t)re] + (hs.rowptr_i[(size_t)ri + 1] - hs.rowptr_i[(size_t)ri]);
    }
    hs.colidx_e.resize((size_t)hs.nnz), hs.slot_i2e.resize((size_t)hs.nnz);
    parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
        std::vector<std::pair<int32_t, int32_t>> tmp;
        for (int64_t ri = b; ri < e; ++ri) {
            const int32_t k0 = hs.rowptr_i[(size_t)ri], k1 = hs.rowptr_i[(size_t)ri + 1];
            tmp.clear();
            for (int32_t k = k0; k < k1; ++k) tmp.emplace_back(hs.dof_i2e[(size_t)hs.colidx_i[(size_t)k]], k);
            std::sort(tmp.begin(), tmp.end());
            const int32_t base = hs.rowptr_e[(size_t)hs.dof_i2e[(size_t)ri]];
            for (size_t t = 0; t < tmp.size(); ++t) {
                hs.colidx_e[(size_t)base + t] = tmp[t].first;
                hs.slot_i2e[(size_t)tmp[t].second] = base + (int32_t)t;
            }
        }
    }, 2048);

    phase("reference pattern + slot map");
    // ---- sliced-ELL adjacency + per-visit column slots -------------------------------------------------------
    const int64_t n_slices = (nd + kSlice - 1) / kSlice;
    const int64_t n_blk = (nd + kAsmBlock - 1) / kAsmBlock;
    auto visits_of = [&](int64_t r) -> int64_t { return r >= 0 && r < nd ? vptr[(size_t)r + 1] - vptr[(size_t)r] : 0; };
    auto slice_widths = [&] {   // width of a slice = the longest visit list among the rows its 64 lane positions hold
        hs.sl_off.assign((size_t)n_slices + 1, 0);
        for (int64_t s = 0; s < n_slices; ++s) {
            int64_t w = 0;
            for (int64_t q = s * kSlice; q < (s + 1) * kSlice; ++q)
                w = std::max(w, visits_of(hs.lane_row.empty() ? q : (int64_t)hs.lane_row[(size_t)q]));
            hs.sl_off[(size_t)s + 1] = hs.sl_off[(size_t)s] + w;
        }
    };
    hs.lane_row.clear();
    slice_widths();
    hvec<int32_t> row_pos;   // row -> lane position (only when the rows are dealt by visit count)
    if ((double)hs.sl_off[(size_t)n_slices] * kSlice > 1.25 * (double)(nc * nb)) {
        // the slices are mostly padding (P2: vertex rows next to edge rows): deal the rows of every assembly block to its lanes
        // in descending order of their visit count (stable), so that a slice holds rows of similar length
        hs.lane_row.resize((size_t)n_blk * kAsmBlock);
        row_pos.resize((size_t)nd);
        parallel_for(n_blk, [&](int64_t b0, int64_t b1, unsigned) {
            for (int64_t b = b0; b < b1; ++b) {
                const int64_t r0 = b * kAsmBlock, r1 = std::min(nd, r0 + kAsmBlock);
                int32_t* lr = &hs.lane_row[(size_t)r0];
                for (int64_t k = 0; k < kAsmBlock; ++k) lr[k] = r0 + k < r1 ? (int32_t)(r0 + k) : -1;
                std::stable_sort(lr, lr + (r1 - r0), [&](int32_t x, int32_t y) { return visits_of(x) > visits_of(y); });
                for (int64_t k = 0; k < r1 - r0; ++k) row_pos[(size_t)lr[k]] = (int32_t)(r0 + k);
            }
        }, 64);
        slice_widths();
    }
    hs.nbw = (nb * 2 + 3) / 4;
    hs.n_slices = n_slices, hs.max_slice_width = 0;
    for (int64_t sl = 0; sl < n_slices; ++sl) hs.max_slice_width = std::max<int32_t>(hs.max_slice_width, (int32_t)(hs.sl_off[(size_t)sl + 1] - hs.sl_off[(size_t)sl]));
    const int64_t padded = hs.sl_off[(size_t)n_slices] * kSlice;
    hs.adj.resize((size_t)padded), hs.slotw.resize((size_t)padded * hs.nbw);
    if (dbg_time)
        std::fprintf(stderr, "sliced-ELL adjacency: %lld visit slots for %lld visits (%.2fx), slot words %.2f GB\n", (long long)padded,
                     (long long)(nc * nb), (double)padded / (double)(nc * nb), (double)padded * hs.nbw * 4.0 / 1e9);
    parallel_for(n_slices, [&](int64_t s0, int64_t s1, unsigned) {   // one pass: real visits and padding (-1 / 0) alike
        Probe slot_of;   // column -> position in the row, built once per row and asked nb times per visit
        for (int64_t sl = s0; sl < s1; ++sl) {
            const int64_t width = hs.sl_off[(size_t)sl + 1] - hs.sl_off[(size_t)sl];
            for (int64_t lane = 0; lane < kSlice; ++lane) {
                const int64_t q = sl * kSlice + lane;   // lane position
                const int64_t rr = hs.lane_row.empty() ? q : (int64_t)hs.lane_row[(size_t)q];
                const int64_t r = rr < 0 ? nd : rr;
                const int64_t len = r < nd ? vptr[(size_t)r + 1] - vptr[(size_t)r] : 0;
                if (r < nd) {
                    const int32_t k0 = hs.rowptr_i[(size_t)r], k1 = hs.rowptr_i[(size_t)r + 1];
                    slot_of.reserve((size_t)(k1 - k0));
                    for (int32_t k = k0; k < k1; ++k) slot_of.insert(hs.colidx_i[(size_t)k]), slot_of.val[slot_of.slot(hs.colidx_i[(size_t)k])] = k - k0;
                }
                for (int64_t v = 0; v < width; ++v) {
                    const int64_t at = (hs.sl_off[(size_t)sl] + v) * kSlice + lane;
                    uint16_t* sw = reinterpret_cast<uint16_t*>(&hs.slotw[(size_t)at * hs.nbw]);
                    if (v < len) {
                        const int32_t visit = vis[(size_t)(vptr[(size_t)r] + v)];
                        hs.adj[(size_t)at] = visit;
                        const int32_t* cd = &hs.cdofs_i[(size_t)(visit >> 4) * nb];
                        for (int j = 0; j < nb; ++j) sw[j] = (uint16_t)slot_of.val[slot_of.slot(cd[j])];
                        for (int j = nb; j < 2 * hs.nbw; ++j) sw[j] = 0;
                    } else {
                        hs.adj[(size_t)at] = -1;
                        for (int j = 0; j < 2 * hs.nbw; ++j) sw[j] = 0;
                    }
                }
                slot_of.clear();
            }
        }
    }, 32);
    hs.blk_nnz_cap.resize((size_t)n_blk);
    hs.max_blk_nnz = 0;
    for (int64_t b = 0; b < n_blk; ++b) {
        int32_t v = hs.rowptr_i[(size_t)std::min(nd, (b + 1) * kAsmBlock)] - hs.rowptr_i[(size_t)b * kAsmBlock];
        hs.blk_nnz_cap[(size_t)b] = v, hs.max_blk_nnz = std::max(hs.max_blk_nnz, v);
    }
    phase("sliced-ELL adjacency + slots");
    // ---- block-local cell / node tables; adj re-encoded as (index in the block's cell table) * 16 + local index -------
    // Per block: the set of visited cells and the set of their vertices, both ascending.  The sets are collected through small
    // open-addressing tables (only the unique ids are sorted; the same tables then answer "position of this id in the sorted
    // set").  A thread serves consecutive blocks and appends their tables to its own buffers, which are copied to their final
    // offsets once the block sizes are known.
    {
        const unsigned nt = n_chunks(n_blk, 8);
        struct ThreadOut {
            int64_t b0 = 0, b1 = 0;
            hvec<int32_t> cell, node;
            hvec<uint16_t> vert;
        };
        std::vector<ThreadOut> tout(nt);
        std::vector<int32_t> n_cells_of((size_t)n_blk), n_nodes_of((size_t)n_blk);
        parallel_for(n_blk, [&](int64_t b0, int64_t b1, unsigned t) {
            ThreadOut& out = tout[t];
            out.b0 = b0, out.b1 = b1;
            Probe pc, pn;
            std::vector<int32_t> cs, ns;
            for (int64_t b = b0; b < b1; ++b) {
                cs.clear(), ns.clear();
                const int64_t r0 = b * kAsmBlock, r1 = std::min(nd, r0 + kAsmBlock);
                pc.reserve((size_t)(vptr[(size_t)r1] - vptr[(size_t)r0]));
                for (int64_t k = vptr[(size_t)r0]; k < vptr[(size_t)r1]; ++k)
                    if (pc.insert(vis[(size_t)k] >> 4)) cs.push_back(vis[(size_t)k] >> 4);
                std::sort(cs.begin(), cs.end());
                for (size_t i = 0; i < cs.size(); ++i) pc.val[pc.slot(cs[i])] = (int32_t)i;
                pn.reserve(cs.size() * (size_t)nv);
                for (int32_t c : cs)
                    for (int v = 0; v < nv; ++v)
                        if (pn.insert(hs.cverts_i[(size_t)c * nv + v])) ns.push_back(hs.cverts_i[(size_t)c * nv + v]);
                std::sort(ns.begin(), ns.end());
                for (size_t i = 0; i < ns.size(); ++i) pn.val[pn.slot(ns[i])] = (int32_t)i;
                n_cells_of[(size_t)b] = (int32_t)cs.size(), n_nodes_of[(size_t)b] = (int32_t)ns.size();
                out.cell.insert(out.cell.end(), cs.begin(), cs.end());
                out.node.insert(out.node.end(), ns.begin(), ns.end());
                for (int32_t c : cs)
                    for (int v = 0; v < 4; ++v)
                        out.vert.push_back(v < nv ? (uint16_t)pn.val[pn.slot(hs.cverts_i[(size_t)c * nv + v])] : (uint16_t)0);
                for (int64_t r = r0; r < r1; ++r) {
                    const int64_t q = row_pos.empty() ? r : (int64_t)row_pos[(size_t)r];   // lane position of the row
                    const int64_t sidx = q / kSlice, lane = q % kSlice;
                    for (int64_t k = vptr[(size_t)r]; k < vptr[(size_t)r + 1]; ++k) {
                        const int64_t at = (hs.sl_off[(size_t)sidx] + (k - vptr[(size_t)r])) * kSlice + lane;
                        hs.adj[(size_t)at] = pc.val[pc.slot(vis[(size_t)k] >> 4)] * 16 + (vis[(size_t)k] & 15);
                    }
                }
                pc.clear(), pn.clear();
            }
        }, 8);
        hs.bc_off.assign((size_t)n_blk + 1, 0), hs.bn_off.assign((size_t)n_blk + 1, 0);
        hs.max_blk_cells = hs.max_blk_nodes = 0;
        for (int64_t b = 0; b < n_blk; ++b) {
            hs.bc_off[(size_t)b + 1] = hs.bc_off[(size_t)b] + n_cells_of[(size_t)b];
            hs.bn_off[(size_t)b + 1] = hs.bn_off[(size_t)b] + n_nodes_of[(size_t)b];
            hs.max_blk_cells = std::max(hs.max_blk_cells, n_cells_of[(size_t)b]);
            hs.max_blk_nodes = std::max(hs.max_blk_nodes, n_nodes_of[(size_t)b]);
        }
        if (std::getenv("FDAPDE_DEBUG_SETUP"))
            std::fprintf(stderr, "assembly blocks: %lld, cells/block avg %.0f max %d, nodes/block avg %.0f max %d, nnz/block max %d\n",
                         (long long)n_blk, (double)hs.bc_off[(size_t)n_blk] / n_blk, hs.max_blk_cells,
                         (double)hs.bn_off[(size_t)n_blk] / n_blk, hs.max_blk_nodes, hs.max_blk_nnz);
        if (hs.max_blk_nodes > 65535) {
            err = "assembly block touches more than 65535 nodes";
            return FDAPDE_EUNSUPPORTED;
        }
        hs.bc_cell.resize((size_t)hs.bc_off[(size_t)n_blk]);
        hs.bc_vert.resize((size_t)hs.bc_off[(size_t)n_blk] * 4);
        hs.bn_node.resize((size_t)hs.bn_off[(size_t)n_blk]);
        parallel_for((int64_t)nt, [&](int64_t tb, int64_t te, unsigned) {
            for (int64_t t = tb; t < te; ++t) {
                const ThreadOut& out = tout[(size_t)t];
                if (out.b1 <= out.b0) continue;
                std::copy(out.cell.begin(), out.cell.end(), hs.bc_cell.begin() + hs.bc_off[(size_t)out.b0]);
                std::copy(out.vert.begin(), out.vert.end(), hs.bc_vert.begin() + hs.bc_off[(size_t)out.b0] * 4);
                std::copy(out.node.begin(), out.node.end(), hs.bn_node.begin() + hs.bn_off[(size_t)out.b0]);
            }
        }, 1);
    }

    phase("block tables");
    // ---- SpMV row blocks: consecutive rows with at most kSpmvNnz nonzeros -----------------------------------
    hs.rb_row.clear();
    hs.rb_row.push_back(0);
    for (int64_t r = 0; r < nd;) {
        int64_t e = r;
        const int32_t base = hs.rowptr_i[(size_t)r];
        while (e < nd && hs.rowptr_i[(size_t)e + 1] - base <= kSpmvNnz && e - r < 1024) ++e;
        hs.rb_row.push_back((int32_t)e);
        r = e;
    }
    hs.n_colours = 0, hs.colour_off.clear(), hs.colour_cells.clear();
    phase("spmv row blocks");
    hs.setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FDAPDE_OK;
}

int host_build_solver_pattern(const HostSpace& hs, bool use_bnd, std::vector<int32_t>& rowptr_s, std::vector<int32_t>& colidx_s,
                              std::vector<int32_t>& full2s) {
    const int64_t nd = hs.n_dofs;
    rowptr_s.assign((size_t)nd + 1, 0);
    full2s.assign((size_t)hs.nnz, -1);
    auto keep = [&](int64_t r, int32_t col) {
        if (col == r) return false;                                               // unit diagonal after Jacobi scaling
        if (use_bnd && (hs.dof_bnd_i[(size_t)r] || hs.dof_bnd_i[(size_t)col])) return false;   // scaled to exact zeros
        return true;
    };
    parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t r = b; r < e; ++r) {
            int32_t cnt = 0;
            for (int32_t k = hs.rowptr_i[(size_t)r]; k < hs.rowptr_i[(size_t)r + 1]; ++k) cnt += keep(r, hs.colidx_i[(size_t)k]);
            rowptr_s[(size_t)r + 1] = cnt;
        }
    });
    for (int64_t r = 0; r < nd; ++r) rowptr_s[(size_t)r + 1] += rowptr_s[(size_t)r];
    colidx_s.assign((size_t)rowptr_s[(size_t)nd] + 2, 0);   // + 2: pair loads may touch one entry past a row's end
    // Entry order inside a row: k_spmv_team2 gives lane l the ALIGNED pair of positions (2 l, 2 l + 1) counted from the even
    // index at or below the row start, and gathers x for all even positions with one instruction and for all odd positions with
    // another.  The row's columns (ascending) are therefore dealt so that the even positions hold the lower half and the odd
    // positions the upper half: each gather instruction then touches about half as many distinct cache lines of x.
    const bool split = !std::getenv("FDAPDE_SPMV_NOSPLIT");
    parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t r = b; r < e; ++r) {
            const int32_t base = rowptr_s[(size_t)r], m = rowptr_s[(size_t)r + 1] - base;
            const int32_t first_even = base & 1;            // offset of the first even position in the row
            const int32_t n_even = (m - first_even + 1) / 2;  // positions base + first_even, + 2, ...
            int32_t j = 0;
            for (int32_t k = hs.rowptr_i[(size_t)r]; k < hs.rowptr_i[(size_t)r + 1]; ++k) {
                if (!keep(r, hs.colidx_i[(size_t)k])) continue;
                int32_t at;
                if (!split)
                    at = base + j;
                else if (j < n_even)
                    at = base + first_even + 2 * j;
                else
                    at = base + (1 - first_even) + 2 * (j - n_even);
                colidx_s[(size_t)at] = hs.colidx_i[(size_t)k], full2s[(size_t)k] = at;
                ++j;
            }
        }
    });
    return FDAPDE_OK;
}

// Segmented form of the compact solver pattern for matrices with rows longer than one team pass (P2: 18 / 26 / 64 entries per
// row on a Kuhn mesh).  Every row is cut into chunks of at most `seg` entries; a chunk is a VIRTUAL row of the CSR arrays that
// k_spmv_team2 streams (one team pass each, no tail loop), padded to an even number of entries so that every virtual row
// starts on an aligned pair.  The chunks of a row are adjacent and never straddle a tile of `wrows` virtual rows (the tile is
// filled up with empty chunks of its last row instead), so the kernel can add them up in its per-tile LDS transpose.
//   vrow[2 v] = row of virtual row v,  vrow[2 v + 1] = chunk index | number of chunks << 8
// Returns FDAPDE_EUNSUPPORTED when a row needs more than `wrows` chunks (the caller then keeps the plain pattern).
int host_build_solver_pattern_seg(const HostSpace& hs, bool use_bnd, int seg, int wrows, std::vector<int32_t>& rowptr_v,
                                  std::vector<int32_t>& colidx_s, std::vector<int32_t>& full2s, std::vector<int32_t>& vrow) {
    const int64_t nd = hs.n_dofs;
    auto keep = [&](int64_t r, int32_t col) {
        if (col == r) return false;
        if (use_bnd && (hs.dof_bnd_i[(size_t)r] || hs.dof_bnd_i[(size_t)col])) return false;
        return true;
    };
    std::vector<int32_t> cnt((size_t)nd);
    parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t r = b; r < e; ++r) {
            int32_t c = 0;
            for (int32_t k = hs.rowptr_i[(size_t)r]; k < hs.rowptr_i[(size_t)r + 1]; ++k) c += keep(r, hs.colidx_i[(size_t)k]);
            cnt[(size_t)r] = c;
        }
    });
    // layout: first virtual row of every row; tiles are closed with empty chunks of their last row
    std::vector<int64_t> vfirst((size_t)nd + 1);
    std::vector<int32_t> nchunk((size_t)nd);
    int64_t nv = 0;
    for (int64_t r = 0; r < nd; ++r) {
        const int32_t nc = cnt[(size_t)r] > 0 ? (cnt[(size_t)r] + seg - 1) / seg : 1;
        if (nc > wrows || nc > 255) return FDAPDE_EUNSUPPORTED;
        const int64_t slot = nv % wrows;
        if (slot + nc > wrows) {   // does not fit: the previous row (same tile) takes the free slots as empty chunks
            const int32_t pad = (int32_t)(wrows - slot);
            if (nchunk[(size_t)r - 1] + pad > 255) return FDAPDE_EUNSUPPORTED;
            nchunk[(size_t)r - 1] += pad, nv += pad;
        }
        vfirst[(size_t)r] = nv, nchunk[(size_t)r] = nc, nv += nc;
    }
    vfirst[(size_t)nd] = nv;
    if (nv > (int64_t)1 << 30) return FDAPDE_EUNSUPPORTED;
    rowptr_v.assign((size_t)nv + 1, 0);
    vrow.assign((size_t)nv * 2 + 2, 0);
    parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
        for (int64_t r = b; r < e; ++r) {
            const int32_t m = cnt[(size_t)r], nc = nchunk[(size_t)r];
            for (int32_t k = 0; k < nc; ++k) {
                const int64_t v = vfirst[(size_t)r] + k;
                const int32_t len = std::max(0, std::min(seg, m - k * seg));
                rowptr_v[(size_t)v + 1] = (len + 1) & ~1;   // even length
                vrow[(size_t)v * 2] = (int32_t)r, vrow[(size_t)v * 2 + 1] = k | (nc << 8);
            }
        }
    });
    for (int64_t v = 0; v < nv; ++v) rowptr_v[(size_t)v + 1] += rowptr_v[(size_t)v];
    if ((int64_t)rowptr_v[(size_t)nv] < 0) return FDAPDE_EUNSUPPORTED;
    colidx_s.assign((size_t)rowptr_v[(size_t)nv] + 2, 0);
    full2s.assign((size_t)hs.nnz, -1);
    parallel_for(nd, [&](int64_t b, int64_t e, unsigned) {
        std::vector<int32_t> ks;
        for (int64_t r = b; r < e; ++r) {
            ks.clear();
            for (int32_t k = hs.rowptr_i[(size_t)r]; k < hs.rowptr_i[(size_t)r + 1]; ++k)
                if (keep(r, hs.colidx_i[(size_t)k])) ks.push_back(k);
            const int32_t m = (int32_t)ks.size();
            for (int32_t c = 0; c * seg < m || c == 0; ++c) {
                const int64_t v = vfirst[(size_t)r] + c;
                const int32_t base = rowptr_v[(size_t)v], len = std::max(0, std::min(seg, m - c * seg));
                const int32_t n_even = (len + 1) / 2;   // even positions hold the lower half of the chunk's columns (see above)
                for (int32_t j = 0; j < len; ++j) {
                    const int32_t at = j < n_even ? base + 2 * j : base + 1 + 2 * (j - n_even);
                    const int32_t k = ks[(size_t)(c * seg + j)];
                    colidx_s[(size_t)at] = hs.colidx_i[(size_t)k], full2s[(size_t)k] = at;
                }
                if (len & 1) colidx_s[(size_t)(base + len)] = hs.colidx_i[(size_t)ks[(size_t)(c * seg + len - 1)]];   // pad entry, value stays 0
                if (len == 0) break;
            }
        }
    });
    return FDAPDE_OK;
}

// 16-bit column codes of a CSR pattern for k_spmv_team2: the kCodeRows consecutive rows of a group share up
// to four windows of 2^14 columns; an entry is stored as (window << 14) | (column - window base).  Windows are placed greedily
// over the group's sorted distinct columns.  A group that needs more than four windows (rows at a corner of the coarse blocks
// of the locality numbering: ~1 % on C3) is marked wide (tbase[4 g] = -1) and read from the 32-bit column array instead.
int host_build_col16(int64_t n, const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, std::vector<uint16_t>& code,
                     std::vector<int32_t>& tbase, int64_t* n_wide) {
    const int64_t ng = (n + kCodeRows - 1) / kCodeRows;
    code.assign(colidx.size(), 0);
    tbase.assign((size_t)ng * 4, 0);
    std::atomic<int64_t> wide{0};
    parallel_for(ng, [&](int64_t g0, int64_t g1, unsigned) {
        std::vector<int32_t> cols;
        for (int64_t g = g0; g < g1; ++g) {
            const int32_t kb = rowptr[(size_t)(g * kCodeRows)], ke = rowptr[(size_t)std::min<int64_t>(n, (g + 1) * kCodeRows)];
            cols.assign(colidx.begin() + kb, colidx.begin() + ke);
            std::sort(cols.begin(), cols.end());
            cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            int32_t base[4] = {0, 0, 0, 0};
            int nw = 0;
            bool fits = true;
            for (size_t i = 0; i < cols.size();) {
                if (nw == 4) {
                    fits = false;
                    break;
                }
                base[nw] = cols[i];
                const int64_t lim = (int64_t)cols[i] + kCodeWindow;
                while (i < cols.size() && cols[i] < lim) ++i;
                ++nw;
            }
            if (!fits) {
                tbase[(size_t)g * 4] = -1;
                wide.fetch_add(1, std::memory_order_relaxed);
                continue;
            }
            for (int w = nw; w < 4; ++w) base[w] = base[nw > 0 ? nw - 1 : 0];
            for (int w = 0; w < 4; ++w) tbase[(size_t)g * 4 + w] = base[w];
            for (int32_t k = kb; k < ke; ++k) {
                const int32_t c = colidx[(size_t)k];
                int w = 0;
                while (w + 1 < nw && c >= base[w + 1]) ++w;
                code[(size_t)k] = (uint16_t)((w << 14) | (c - base[w]));
            }
        }
    }, 64);
    if (n_wide) *n_wide = wide.load();
    return FDAPDE_OK;
}

// Greedy element colouring over the internal cell order: cells of one colour share no DOF, so their scatter into the
// global matrix needs no atomics.  Colour-contiguous cell lists keep the index reads of each pass coalesced.
int host_build_colouring(HostSpace& hs, std::string& err) {
    if (hs.n_colours > 0) return FDAPDE_OK;
    constexpr int W = 4;   // 256 colours
    const int nb = hs.nb;
    std::vector<uint64_t> mask((size_t)hs.n_dofs * W, 0);
    std::vector<int32_t> colour((size_t)hs.n_cells);
    int ncol = 0;
    for (int64_t c = 0; c < hs.n_cells; ++c) {
        uint64_t used[W] = {0, 0, 0, 0};
        for (int j = 0; j < nb; ++j)
            for (int w = 0; w < W; ++w) used[w] |= mask[(size_t)hs.cdofs_i[(size_t)c * nb + j] * W + w];
        int col = -1;
        for (int w = 0; w < W && col < 0; ++w)
            if (~used[w]) col = w * 64 + __builtin_ctzll(~used[w]);
        if (col < 0) {
            err = "more than 256 colours needed";
            return FDAPDE_EUNSUPPORTED;
        }
        colour[(size_t)c] = col, ncol = std::max(ncol, col + 1);
        for (int j = 0; j < nb; ++j) mask[(size_t)hs.cdofs_i[(size_t)c * nb + j] * W + col / 64] |= uint64_t(1) << (col % 64);
    }
    hs.n_colours = ncol;
    hs.colour_off.assign((size_t)ncol + 1, 0);
    for (int64_t c = 0; c < hs.n_cells; ++c) ++hs.colour_off[(size_t)colour[(size_t)c] + 1];
    for (int k = 0; k < ncol; ++k) hs.colour_off[(size_t)k + 1] += hs.colour_off[(size_t)k];
    hs.colour_cells.resize((size_t)hs.n_cells);
    std::vector<int32_t> pos(hs.colour_off.begin(), hs.colour_off.end() - 1);
    for (int64_t c = 0; c < hs.n_cells; ++c) hs.colour_cells[(size_t)pos[(size_t)colour[(size_t)c]]++] = (int32_t)c;
    return FDAPDE_OK;
}

}  // namespace fdapde_hip
