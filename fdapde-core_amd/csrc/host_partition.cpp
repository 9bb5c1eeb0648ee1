// host_partition.cpp -- index work for the element-wise scatter forms of the assembly (kernels_assembly.h, k_assemble_part /
// k_assemble_wave): the north-star form of BASELINE.json, kept as a measured alternative to the row-owner sweep.
//   * cell partitions: contiguous chunks of the internal (Morton) cell order, one per workgroup; inside a partition the cells are
//     coloured greedily (cells of a colour share no DOF) and listed colour by colour, so that a workgroup walks its colours with a
//     barrier in between and needs atomics only for rows that cells of ANOTHER partition also touch;
//   * the slot map: for every listed cell the CSR slot of each of its nb x nb element-matrix entries, streamed by the kernels
//     instead of being searched for.
// Built lazily, only when one of these assembly variants is asked for.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

namespace fdapde_hip {

namespace {

template <typename F> void chunked(int64_t n, F&& fn) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if (n < 4096 || nt == 1) {
        fn(int64_t(0), n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t b = (int64_t)t * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        th.emplace_back([=, &fn] { fn(b, e); });
    }
    for (auto& x : th) x.join();
}

}  // namespace

void host_build_slot_map(const HostSpace& hs, const int32_t* list, int64_t n, std::vector<int32_t>& out) {
    const int nb = hs.nb;
    out.resize((size_t)n * nb * nb);
    chunked(n, [&](int64_t b, int64_t e) {
        for (int64_t li = b; li < e; ++li) {
            const int32_t* cd = &hs.cdofs_i[(size_t)list[li] * nb];
            for (int i = 0; i < nb; ++i) {
                const int32_t k0 = hs.rowptr_i[(size_t)cd[i]], k1 = hs.rowptr_i[(size_t)cd[i] + 1];
                for (int j = 0; j < nb; ++j) {
                    const int32_t* at = std::lower_bound(&hs.colidx_i[(size_t)k0], &hs.colidx_i[(size_t)k1], cd[j]);
                    out[((size_t)li * nb + i) * nb + j] = (int32_t)(at - hs.colidx_i.data());
                }
            }
        }
    });
}

int host_build_cell_partitions(const HostSpace& hs, int cells_per_part, CellPartitions& cp, std::string& err) {
    const int nb = hs.nb;
    const int64_t nc = hs.n_cells;
    if (cells_per_part < 64) cells_per_part = 64;
    const int64_t n_parts = (nc + cells_per_part - 1) / cells_per_part;
    cp.cells_per_part = cells_per_part, cp.n_parts = n_parts;
    // DOFs touched by cells of two or more partitions
    std::vector<std::atomic<int32_t>> pmin((size_t)hs.n_dofs), pmax((size_t)hs.n_dofs);
    for (int64_t d = 0; d < hs.n_dofs; ++d) pmin[(size_t)d].store(INT32_MAX, std::memory_order_relaxed), pmax[(size_t)d].store(-1, std::memory_order_relaxed);
    chunked(nc, [&](int64_t b, int64_t e) {
        for (int64_t c = b; c < e; ++c) {
            const int32_t p = (int32_t)(c / cells_per_part);
            for (int j = 0; j < nb; ++j) {
                const size_t d = (size_t)hs.cdofs_i[(size_t)c * nb + j];
                int32_t cur = pmin[d].load(std::memory_order_relaxed);
                while (p < cur && !pmin[d].compare_exchange_weak(cur, p, std::memory_order_relaxed)) {}
                cur = pmax[d].load(std::memory_order_relaxed);
                while (p > cur && !pmax[d].compare_exchange_weak(cur, p, std::memory_order_relaxed)) {}
            }
        }
    });
    cp.dof_shared.resize((size_t)hs.n_dofs);
    for (int64_t d = 0; d < hs.n_dofs; ++d)
        cp.dof_shared[(size_t)d] = pmin[(size_t)d].load(std::memory_order_relaxed) != pmax[(size_t)d].load(std::memory_order_relaxed) ? 1 : 0;
    // greedy colouring inside every partition (128 colours at most), cells listed colour by colour
    constexpr int W = 2;
    cp.cell_list.resize((size_t)nc);
    std::vector<int32_t> ncol((size_t)n_parts, 0);
    std::vector<uint8_t> colour((size_t)nc);
    std::atomic<int> bad{0};
    chunked(n_parts, [&](int64_t pb, int64_t pe) {
        // DOF -> colours in use, for the DOFs of the current partition: open addressing table reset per partition
        size_t cap = (size_t)1 << 12;   // open addressing at load <= 1/4: 4 x (cells x nb) slots
        while (cap < (size_t)cells_per_part * nb * 4) cap <<= 1;
        std::vector<int32_t> key(cap, -1);
        std::vector<uint64_t> val(cap * W, 0);
        std::vector<size_t> touched;
        for (int64_t p = pb; p < pe; ++p) {
            const int64_t c0 = p * cells_per_part, c1 = std::min(nc, c0 + cells_per_part);
            if ((size_t)(c1 - c0) * nb * 4 > cap) {
                bad.store(1);
                return;
            }
            int nc_p = 0;
            for (int64_t c = c0; c < c1; ++c) {
                uint64_t used[W] = {0, 0};
                size_t at[kMaxBasis];
                for (int j = 0; j < nb; ++j) {
                    const int32_t d = hs.cdofs_i[(size_t)c * nb + j];
                    size_t h = ((size_t)d * 0x9E3779B97F4A7C15ull) >> 47 & (cap - 1);
                    while (key[h] != -1 && key[h] != d) h = (h + 1) & (cap - 1);
                    if (key[h] == -1) key[h] = d, touched.push_back(h);
                    at[j] = h;
                    for (int w = 0; w < W; ++w) used[w] |= val[h * W + w];
                }
                int col = -1;
                for (int w = 0; w < W && col < 0; ++w)
                    if (~used[w]) col = w * 64 + __builtin_ctzll(~used[w]);
                if (col < 0) {
                    bad.store(2);
                    return;
                }
                colour[(size_t)c] = (uint8_t)col, nc_p = std::max(nc_p, col + 1);
                for (int j = 0; j < nb; ++j) val[at[j] * W + col / 64] |= uint64_t(1) << (col % 64);
            }
            ncol[(size_t)p] = nc_p;
            for (size_t h : touched) {
                key[h] = -1;
                for (int w = 0; w < W; ++w) val[h * W + w] = 0;
            }
            touched.clear();
        }
    });
    if (bad.load() == 1) {
        err = "cell partitions too large for the colouring table";
        return FDAPDE_EUNSUPPORTED;
    }
    if (bad.load() == 2) {
        err = "more than 128 colours needed inside a cell partition";
        return FDAPDE_EUNSUPPORTED;
    }
    cp.max_colours = *std::max_element(ncol.begin(), ncol.end());
    const int MC = cp.max_colours;
    cp.colour_off.assign((size_t)n_parts * (MC + 1), 0);
    chunked(n_parts, [&](int64_t pb, int64_t pe) {
        for (int64_t p = pb; p < pe; ++p) {
            const int64_t c0 = p * cells_per_part, c1 = std::min(nc, c0 + cells_per_part);
            int32_t* off = &cp.colour_off[(size_t)p * (MC + 1)];
            std::vector<int32_t> cnt((size_t)MC + 1, 0);
            for (int64_t c = c0; c < c1; ++c) ++cnt[(size_t)colour[(size_t)c] + 1];
            off[0] = (int32_t)c0;
            for (int k = 0; k < MC; ++k) off[k + 1] = off[k] + cnt[(size_t)k + 1];
            std::vector<int32_t> pos(off, off + MC);
            for (int64_t c = c0; c < c1; ++c) cp.cell_list[(size_t)pos[(size_t)colour[(size_t)c]]++] = (int32_t)c;
        }
    });
    host_build_slot_map(hs, cp.cell_list.data(), nc, cp.slot_map);
    return FDAPDE_OK;
}

}  // namespace fdapde_hip
