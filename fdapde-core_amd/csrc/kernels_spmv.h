// kernels_spmv.h -- CSR SpMV fused with the Krylov dot products (stream, team, pair / production forms); see kernels.h
#ifndef FDAPDE_KERNELS_SPMV_H
#define FDAPDE_KERNELS_SPMV_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "internal.h"
#include "kernels_reduce.h"

namespace fdapde_hip {

// ---------------------------------------------------------------------------------------------------------------
// CSR SpMV, "stream" form: a workgroup takes a row block (consecutive rows, <= kSpmvNnz nonzeros), streams its
// contiguous val/colidx range with unit stride, multiplies by the gathered x[col] into LDS, then one lane per row
// adds up that row's products (ascending column order, like the scalar oracle).  Fused: y = A x and the partial of
// dot(w, y) with w = x (CG's p.Ap) or w = a second vector (BiCGStab's r0.v, t.s) and of dot(y, y).
// Grid = 8 * BPX workgroups; workgroup b serves the row blocks of band (b % 8): workgroups that share an XCD (and
// its 4 MiB L2) work on one contiguous eighth of the rows, so the x entries they gather stay in that L2.
// Algorithmic HBM bytes per launch: 12 nnz + 4 (n+1) + 16 n   (BASELINE.md).
// ---------------------------------------------------------------------------------------------------------------
struct SpmvArgs {
    const int32_t* rowptr;
    const int32_t* colidx;
    const double* vals;
    const double* x;
    double* y;
    const int32_t* rb_row;
    int32_t n_rb, rb_per_band, nnz;
    const double* w;       // second vector of the fused dot products; nullptr: no dots
    double* partial;       // [2 * gridDim.x]: workgroup b writes (w.y, y.y) at 2b, 2b+1; nullptr: no dots
    const int32_t* stop;   // device flag: nonzero -> converged, kernel returns immediately (may be nullptr)
    int32_t unit_diag;     // compact solver matrix: the (dropped) diagonal is 1, y_i = x_i + sum of the stored entries
    int32_t dot2_ww;       // second fused dot: 0 -> y.y (BiCGStab's t.t), 1 -> w.w over owned rows (single-reduction CG's r.r)
    const uint8_t* owned;  // multi-GPU: rows this rank counts in w.w (nullptr = all)
    const uint16_t* col16; // 16-bit column codes (window << 14 | offset) of the pattern, or nullptr   (host_build_col16)
    const int32_t* tbase;  // four window bases per group of 32 rows; tbase[4 g] < 0: wide group, read colidx instead
    const int32_t* vrow;   // segmented pattern (host_build_solver_pattern_seg): (row, chunk | n_chunks << 8) per virtual row
    int32_t n_cols;        // number of columns = length of x (the row count the kernels get may be the virtual one)
};
__device__ __forceinline__ double spmv_dot2(const SpmvArgs& s, int64_t row, double wv, double out) {
    if (!s.dot2_ww) return out * out;
    return (s.owned && !s.owned[row]) ? 0.0 : wv * wv;
}

static __global__ __launch_bounds__(256) void k_spmv(SpmvArgs s) {
    __shared__ double prod[kSpmvNnz];
    __shared__ double red[8];
    if (s.stop && __syncthreads_or(*s.stop != 0)) return;   // wave- and workgroup-uniform exit
    const int band = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int rb_end = min(s.n_rb, (band + 1) * s.rb_per_band);
    double d_wy = 0, d_yy = 0;
    for (int rb = band * s.rb_per_band + lb; rb < rb_end; rb += bpx) {
        const int r0 = s.rb_row[rb], r1 = s.rb_row[rb + 1];
        const int k0 = s.rowptr[r0], k1 = s.rowptr[r1];
        int k = k0 + threadIdx.x;
        for (; k + 3 * 256 < k1; k += 4 * 256) {   // 4 independent streams per lane in flight
            const double v0 = s.vals[k], v1 = s.vals[k + 256], v2 = s.vals[k + 512], v3 = s.vals[k + 768];
            const int c0 = s.colidx[k], c1 = s.colidx[k + 256], c2 = s.colidx[k + 512], c3 = s.colidx[k + 768];
            const double x0 = s.x[c0], x1 = s.x[c1], x2 = s.x[c2], x3 = s.x[c3];
            prod[k - k0] = v0 * x0, prod[k - k0 + 256] = v1 * x1;
            prod[k - k0 + 512] = v2 * x2, prod[k - k0 + 768] = v3 * x3;
        }
        for (; k < k1; k += 256) prod[k - k0] = s.vals[k] * s.x[s.colidx[k]];
        __syncthreads();
        for (int r = r0 + threadIdx.x; r < r1; r += 256) {
            const int a = s.rowptr[r] - k0, b = s.rowptr[r + 1] - k0;
            double acc = 0;
            for (int i = a; i < b; ++i) acc += prod[i];
            s.y[r] = acc;
            if (s.w) d_wy += s.w[r] * acc, d_yy += spmv_dot2(s, r, s.w[r], acc);
        }
        __syncthreads();
    }
    if (s.partial) {
        const double a = block_sum(d_wy, red);
        const double b = block_sum(d_yy, red);
        if (threadIdx.x == 0) s.partial[2 * blockIdx.x] = a, s.partial[2 * blockIdx.x + 1] = b;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// CSR SpMV, "team" form (default): T lanes per row, T = the power of two covering the mean row length (16 for 3-D P1,
// 15 nonzeros per interior row), U rows per team in flight.  Consecutive teams take consecutive rows, so every
// val / colidx load instruction of a wavefront covers one contiguous range of the CSR arrays (64/T rows); there is no
// LDS staging and no barrier, each lane keeps U independent load -> gather chains in flight and all 32 wave slots of a
// CU are usable (the stream form is capped at 20 by its LDS tile and stalls at two barriers per tile: measured 3.5 TB/s
// with 81 % of wave cycles waiting, profiles/r1_c3_summary.txt).  The U x 64/T row sums of a wave-iteration are
// shuffled to adjacent lanes so that y (and the fused dot operands) move as one contiguous segment.
// Same XCD banding, same fused partial dots, same algorithmic bytes as the stream form.  The in-team tree sum is a
// fixed order: results are bitwise reproducible run to run.
// ---------------------------------------------------------------------------------------------------------------
template <int T, int U>
static __global__ __launch_bounds__(256) void k_spmv_team(SpmvArgs s, int64_t n, int64_t rows_per_band) {
    constexpr int TEAMS = 64 / T;
    constexpr int WROWS = TEAMS * U;   // rows per wave-iteration (tile); WROWS + 1 <= 64
    static_assert(WROWS < 64, "one rowptr load per tile");
    __shared__ double red[8];
    if (s.stop && __syncthreads_or(*s.stop != 0)) return;
    const int band = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, team = lane / T, l = lane % T;
    const int64_t band_begin = band * rows_per_band;
    const int64_t band_end = min(n, band_begin + rows_per_band);
    const int64_t stride = (int64_t)bpx * 4 * WROWS;
    double d_wy = 0, d_yy = 0;
    // Software pipeline over tiles, three stages in flight per wavefront:
    //   tile i+2: its WROWS+1 row pointers (one coalesced load; rows past the band clamp to an empty range)
    //   tile i+1: its val / colidx loads (issued AFTER tile i's gathers, so the wait on the gathers leaves them in flight)
    //   tile i  : x gathers, products, in-team sums, store
    // Every load below is UNCONDITIONAL (indices are clamped, idle lanes re-read a neighbour's entry and discard it): a
    // load under an exec-masked branch makes hipcc's s_waitcnt insertion assume it may not have been issued and fall
    // back to vmcnt(0), which would drain the next tile's loads at the gather wait and undo the pipeline.
    const int last = s.rowptr[n] - 1;   // nnz - 1 (>= 0)
    auto load_rp = [&](int64_t base) -> int {
        const int64_t r = base + lane;
        return s.rowptr[r < band_end ? r : band_end];
    };
    int64_t base = band_begin + (int64_t)(lb * 4 + wave) * WROWS;
    if (base < band_end) {
        int rp0 = load_rp(base);
        int rp1 = load_rp(base + stride);
        int rs[U], re[U], c[U];
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rs[u] = __shfl(rp0, u * TEAMS + team, 64), re[u] = __shfl(rp0, u * TEAMS + team + 1, 64);
            const int k = rs[u] + l;
            const int kc = k < last ? k : last;
            const double vv = s.vals[kc];
            c[u] = s.colidx[kc];
            v[u] = k < re[u] ? vv : 0.0;
        }
        for (; base < band_end; base += stride) {
            double xg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xg[u] = s.x[c[u]];
            // stage the next tile before consuming the gathers
            int rsn[U], ren[U], cn[U];
            double vn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rsn[u] = __shfl(rp1, u * TEAMS + team, 64), ren[u] = __shfl(rp1, u * TEAMS + team + 1, 64);
                const int k = rsn[u] + l;
                const int kc = k < last ? k : last;
                const double vv = s.vals[kc];
                cn[u] = s.colidx[kc];
                vn[u] = k < ren[u] ? vv : 0.0;
            }
            rp1 = load_rp(base + 2 * stride);
            double acc[U];
            bool long_row = false;
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = v[u] * xg[u], long_row |= re[u] - rs[u] > T;
            if (__any(long_row)) {   // rows longer than a team (rare when T covers the mean row)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    for (int k = rs[u] + l + T; k < re[u]; k += T) acc[u] += s.vals[k] * s.x[s.colidx[k]];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int o = T / 2; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o, T);
            }
            // row (u, team) -> lane u*TEAMS + team: lanes 0..WROWS-1 hold WROWS consecutive rows
            double out = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double t = __shfl(acc[u], (lane % TEAMS) * T, 64);
                if (lane / TEAMS == u) out = t;
            }
            const int64_t row = base + lane;
            if (lane < WROWS && row < band_end) {
                s.y[row] = out;
                if (s.w) d_wy += s.w[row] * out, d_yy += spmv_dot2(s, row, s.w[row], out);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) rs[u] = rsn[u], re[u] = ren[u], c[u] = cn[u], v[u] = vn[u];
        }
    }
    if (s.partial) {
        const double a = block_sum(d_wy, red);
        const double b = block_sum(d_yy, red);
        if (threadIdx.x == 0) s.partial[2 * blockIdx.x] = a, s.partial[2 * blockIdx.x + 1] = b;
    }
}

// read-bandwidth probe: streams `bytes` (multiple of 16) with 16 B per lane, persistent grid; calibrates what the chip
// delivers for a pure read stream next to the SpMV numbers
static __global__ __launch_bounds__(256) void k_read_probe(const double2* src, int64_t n16, double* sink) {
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e-300) sink[0] = acc;   // never true; keeps the loads alive
}

// matrix-stream probe: reads vals (16 B / lane) and colidx (8 B / lane) exactly once, in order, nothing else: the time a
// CSR SpMV of this matrix cannot beat on this chip
typedef double v2f64_t __attribute__((ext_vector_type(2)));
typedef int v2i32_t __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) F64x2u { double x, y; };
// same stream with the 16-byte loads based at an address that is only 8-byte aligned (what an odd row start gives)
static __global__ __launch_bounds__(256) void k_stream_probe_unaligned(const double* vals, const int2* col2, int64_t n2, double* sink) {
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2 - 1; i += (int64_t)gridDim.x * blockDim.x) {
        const F64x2u v = *reinterpret_cast<const F64x2u*>(vals + 2 * i + 1);
        const int2 c = col2[i];
        acc += v.x * c.x + v.y * c.y;
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}
static __global__ __launch_bounds__(256) void k_stream_probe(const double2* vals2, const int2* col2, int64_t n2, double* sink) {
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const v2f64_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f64_t*>(vals2) + i);
        const v2i32_t c = __builtin_nontemporal_load(reinterpret_cast<const v2i32_t*>(col2) + i);
        acc += v.x * c.x + v.y * c.y;
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}

// matrix-stream probe + a small write stream: every lane writes one double per 8 pairs it reads (about the y / matrix byte
// ratio of the SpMV), contiguous across the wavefront
static __global__ __launch_bounds__(256) void k_stream_probe_w(const double2* vals2, const int2* col2, int64_t n2, double* out) {
    const int64_t nth = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = tid, o = tid;
    while (i < n2) {
        double acc = 0;
        for (int k = 0; k < 8 && i < n2; ++k, i += nth) {
            const v2f64_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f64_t*>(vals2) + i);
            const v2i32_t c = __builtin_nontemporal_load(reinterpret_cast<const v2i32_t*>(col2) + i);
            acc += v.x * c.x + v.y * c.y;
        }
        out[o] = acc;
        o += nth;
    }
}

// Team form with two consecutive entries per lane: every val load instruction is 16 B per lane (1 KiB per wavefront, the
// widest global access), every colidx load 8 B per lane.  T lanes cover 2 T entries of a row per pass.  The CSR value /
// index arrays carry two padding entries so that the pair load of a row's last odd entry stays in bounds; the pair base
// is 8-byte aligned only (row starts are arbitrary), which global_load_dwordx4 accepts.
typedef __attribute__((address_space(3))) volatile double lds_vf64_t;
struct __attribute__((packed, aligned(8))) F64x2 { double x, y; };
struct __attribute__((packed, aligned(4))) I32x2 { int x, y; };

// ABL: bit set of layout / instantiation flags (2048 aligned pairs, 4096 16-bit column codes, 8192 multi-GPU ownership loads,
// 16384 dot operand == x, 131072 segmented rows) and of diagnostic switches selected by FDAPDE_SPMV_ABLATE / fdapde_tune (results
// of 1, 2, 8, 32, 64, 65536 are wrong on purpose):
//   1: no x gather   2 / 65536: gathers confined to 16 lines / 1 line   8, 32, 64: no y store / no w load   16: no XCD banding
//   4: the oldest (unaligned) pair form with default-policy loads   32768: y kept in LDS until the tile loop ends
//   262144: window bases as a 16-byte broadcast load   524288 / 1048576: flip the cache policy of the codes / values
// Cache policy of the matrix stream: see load_pair (default for teams of <= 8 lanes, nontemporal for wider teams; the oldest
// unaligned form still carries the hint it was tuned with: 69.4 us against 70.7 us default at the time).
template <int T, int U, int ABL = 0, int OCC = 4>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, 8))) void k_spmv_team2(SpmvArgs s, int64_t n,
                                                                                              int64_t rows_per_band) {
    constexpr int TEAMS = 64 / T;
    constexpr int WROWS = TEAMS * U;
    static_assert(WROWS < 64, "one rowptr load per tile");
    __shared__ double red[8];
    __shared__ double ystage[4][WROWS];
    // DEFER (diagnostic / tuning): the y rows of a wavefront stay in LDS until its tile loop ends and leave in one burst
    constexpr bool DEFER = (ABL & 32768) != 0;
    constexpr int kDeferTiles = 8;
    __shared__ double ydef[DEFER ? 4 * kDeferTiles * WROWS : 1];
    if (s.stop && __syncthreads_or(*s.stop != 0)) return;
    const int band = (ABL & 16) ? 0 : (blockIdx.x & 7), lb = (ABL & 16) ? blockIdx.x : (blockIdx.x >> 3),
              bpx = (ABL & 16) ? gridDim.x : (gridDim.x >> 3);
    if (ABL & 16) rows_per_band = n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, team = lane / T, l = lane % T;
    // Tiles are dealt round-robin to the wavefronts of a band, so that at any moment the wavefronts of an XCD read one
    // advancing window of the CSR arrays.  (Giving every wavefront its own contiguous share of rows balances the tail
    // better but measured 9 % slower at C3 size, 81.7 vs 74.8 us: thousands of independent address streams.)
    const int64_t band_begin = band * rows_per_band;
    const int64_t band_end = min(n, band_begin + rows_per_band);
    const int64_t stride = (int64_t)bpx * 4 * WROWS;
    double d_wy = 0, d_yy = 0;
    int n_def = 0;
    const int last = s.nnz - 1;
    // row pointers: ONE coalesced load of the tile's WROWS + 1 pointers, shuffled to the teams (8 ds_bpermute per tile).
    // Loading them per (tile, u) with team-uniform 8-byte loads instead was measured slower (79.5 vs 66.0 us): every
    // extra vector-memory instruction costs address-processing time whatever its footprint.
    auto load_rp = [&](int64_t base) -> int {
        const int64_t r = base + lane;
        return s.rowptr[r < band_end ? r : band_end];
    };
    // ALIGNED: a row's lane pairs start at the even index rs & ~1, so that every pair is one 16-byte-aligned val load and
    // one 8-byte-aligned colidx load (the entry below rs, if any, belongs to the previous row and is masked)
    constexpr bool ALIGNED = (ABL & 2048) != 0;
    // C16: the columns come as 16-bit codes, two per 4-byte load (needs the aligned pairs), decoded with the four window
    // bases of the 32-row group when the gathers are issued: 2 instead of 4 index bytes per entry
    constexpr bool C16 = (ABL & 4096) != 0;
    static_assert(!C16 || (ALIGNED && 32 % WROWS == 0), "16-bit column codes need aligned pairs and tiles inside a 32-row group");
    // VROWS: the CSR rows are the chunks ("virtual rows") of a segmented pattern; the chunks of a row sit in one tile and are
    // added up after the LDS transpose, every chunk lane storing the row's total to the row's y (same value, same address)
    constexpr bool VROWS = (ABL & 131072) != 0;
    static_assert(!VROWS || ALIGNED, "segmented patterns start every virtual row on an aligned pair");
    auto load_pair = [&](int rs, int re, F64x2& v, I32x2& c) {
        const int k = (ALIGNED ? (rs & ~1) : rs) + 2 * l;
        const int kc = ALIGNED ? (k < last ? k : (last & ~1)) : (k < last ? k : last);
        F64x2 vv;
        I32x2 cc;
        // Cache policy of the matrix stream: DEFAULT.  A wavefront's val / code load covers 8 rows, i.e. pieces of 128-byte lines
        // that the next load of the same wavefront (the next 8 rows) completes; with the nontemporal hint the lines are not kept
        // and get fetched again (C3, same box: both nontemporal 58.2 us, codes default 52.0, values default 45.2, both 45.3 us
        // back-to-back; inside CG 40.6 / 37.3 / 33.9 / 32.7 ms per solve).  NT_V / NT_C re-enable the hint for measurements.
        // Teams of 16+ lanes (rows of 32+ entries, P2) read whole lines per row and keep the hint (C5-size matrix, same box: both
        // nontemporal 361.6 us, both default 371.5 us).
        constexpr bool NT_V = ((ABL & 1048576) != 0) != (T >= 16), NT_C = ((ABL & 524288) != 0) != (T >= 16);
        if constexpr (C16) {
            // Large matrices -- the x / y slices of a row band no longer fit the XCD's L2 next to a default-policy value stream --
            // are launched with the hint on the values only (flag 1048576: 2.5 M rows 88.5 -> 76.9 us; C3 45.3 -> 52.0 us).  A
            // run-time switch between the two loads does not survive the optimiser (the loads are merged and the hint dropped).
            v2f64_t a;
            if constexpr (NT_V)
                a = __builtin_nontemporal_load(reinterpret_cast<const v2f64_t*>(s.vals + kc));
            else
                a = *reinterpret_cast<const v2f64_t*>(s.vals + kc);
            vv.x = a.x, vv.y = a.y;
            if constexpr (NT_C)
                cc.x = (int)__builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(s.col16 + kc)), cc.y = 0;
            else
                cc.x = (int)*reinterpret_cast<const unsigned int*>(s.col16 + kc), cc.y = 0;
        } else if constexpr (ALIGNED) {
            const v2f64_t a = *reinterpret_cast<const v2f64_t*>(s.vals + kc);
            const v2i32_t b = *reinterpret_cast<const v2i32_t*>(s.colidx + kc);
            vv.x = a.x, vv.y = a.y, cc.x = b.x, cc.y = b.y;
        } else if constexpr (!(ABL & 4)) {
            vv.x = __builtin_nontemporal_load(s.vals + kc), vv.y = __builtin_nontemporal_load(s.vals + kc + 1);
            cc.x = __builtin_nontemporal_load(s.colidx + kc), cc.y = __builtin_nontemporal_load(s.colidx + kc + 1);
        } else {
            vv = *reinterpret_cast<const F64x2*>(s.vals + kc);
            cc = *reinterpret_cast<const I32x2*>(s.colidx + kc);
        }
        const bool ok0 = k < re && (!ALIGNED || k >= rs), ok1 = k + 1 < re;
        v.x = ok0 ? vv.x : 0.0, v.y = ok1 ? vv.y : 0.0;
        if constexpr (C16)
            c = cc;   // raw code pair; masked entries have a zero value and their decoded column is clamped into range
        else
            c.x = ok0 ? cc.x : 0, c.y = ok1 ? cc.y : 0;
    };
    // window bases of the 32-row group of a tile (wave-uniform address)
    typedef int v4i32_t __attribute__((ext_vector_type(4)));
    // the four window bases travel as ONE dword per lane (lane & 3) and are broadcast with v_readlane when the tile is decoded,
    // instead of a 16-byte load that returns the same 16 bytes to all 64 lanes (61.7 -> 61.4 us; ABL & 262144 keeps the old form)
    constexpr bool TBLANE = (ABL & 262144) == 0;
    auto load_tb = [&](int64_t b) -> v4i32_t {
        if constexpr (C16) {
            const int64_t bc = b < band_end ? b : band_begin;
            const int g = __builtin_amdgcn_readfirstlane((int)(bc >> 5));
            if constexpr (TBLANE)
                return v4i32_t{s.tbase[4 * (int64_t)g + (lane & 3)], 0, 0, 0};
            else
                return *reinterpret_cast<const v4i32_t*>(s.tbase + 4 * (int64_t)g);
        } else
            return v4i32_t{0, 0, 0, 0};
    };
    auto expand_tb = [&](const v4i32_t& t) -> v4i32_t {
        if constexpr (C16 && TBLANE)
            return v4i32_t{__builtin_amdgcn_readlane(t.x, 0), __builtin_amdgcn_readlane(t.x, 1), __builtin_amdgcn_readlane(t.x, 2),
                           __builtin_amdgcn_readlane(t.x, 3)};
        else
            return t;
    };
    const int ncol1 = s.n_cols - 1;
    auto decode = [&](unsigned int code, const v4i32_t& tb) -> int {
        const int b01 = (code & 0x4000u) ? tb.y : tb.x, b23 = (code & 0x4000u) ? tb.w : tb.z;
        const int col = ((code & 0x8000u) ? b23 : b01) + (int)(code & 0x3fffu);
        return col < ncol1 ? col : ncol1;
    };
    auto load_vi = [&](int64_t b) -> v2i32_t {   // (row, chunk info) of this lane's virtual row in the tile at b (clamped)
        if constexpr (VROWS) {
            const int64_t v = b + (lane % WROWS);
            return *reinterpret_cast<const v2i32_t*>(s.vrow + 2 * (v < band_end ? v : band_end - 1));
        } else
            return v2i32_t{0, 0};
    };
    int64_t base = band_begin + (int64_t)(lb * 4 + wave) * WROWS;
    if (base < band_end) {
        int rp0 = load_rp(base);
        int rp1 = load_rp(base + stride);
        int rs[U], re[U];
        F64x2 v[U];
        I32x2 c[U];
        v4i32_t tb = load_tb(base);
        v2i32_t vi = load_vi(base);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rs[u] = __shfl(rp0, u * TEAMS + team, 64), re[u] = __shfl(rp0, u * TEAMS + team + 1, 64);
            load_pair(rs[u], re[u], v[u], c[u]);
        }
        // explicit LDS address space: a volatile GENERIC pointer compiles to flat_load / flat_store, which count on vmcnt and
        // made every tile drain all of its prefetched loads (s_waitcnt vmcnt(0))
        lds_vf64_t* ys = (lds_vf64_t*)&ystage[wave][0];
        const double* wp = s.w ? s.w : s.x;   // always dereferenceable; the dots are discarded when s.w is null
        const double dots = s.w ? 1.0 : 0.0;
        // One tile.  FULL tiles (all WROWS rows inside the band) run branch-free: the w operand of the fused dot is loaded
        // WITH the gathers (a load issued after the reduction would expose a full memory latency per tile), and the y
        // store is unconditional -- lanes l >= U repeat lane l % U (same value, same address), because a store under an
        // exec-masked branch makes the next iteration's wait for the val/colidx loads a vmcnt(0) that also drains the store.
        // Both together: 66.9 -> 57.7 us in the ablation.  The band's last, partial tile takes the masked path once.
        auto tile = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            double xa[U], xb[U];
            const v4i32_t tb_lane = tb;
            tb = expand_tb(tb_lane);
            if constexpr (C16) {
                if (tb.x < 0) {   // wide group (wave-uniform, rare): its columns do not fit four windows
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = (rs[u] & ~1) + 2 * l;
                        const v2i32_t b = *reinterpret_cast<const v2i32_t*>(s.colidx + (k < last ? k : (last & ~1)));
                        c[u].x = b.x, c[u].y = b.y;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const unsigned int code = (unsigned int)c[u].x;
                        c[u].x = decode(code & 0xffffu, tb), c[u].y = decode(code >> 16, tb);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (ABL & 1)
                    xa[u] = (double)(c[u].x & 1), xb[u] = (double)(c[u].y & 1);
                else if constexpr (ABL & 65536)   // diagnostic: every lane of a gather hits ONE cache line
                    xa[u] = s.x[c[u].x & 15], xb[u] = s.x[c[u].y & 15];
                else if constexpr (ABL & 2)
                    xa[u] = s.x[c[u].x & 255], xb[u] = s.x[c[u].y & 255];
                else
                    xa[u] = s.x[c[u].x], xb[u] = s.x[c[u].y];
            }
            // lane j < WROWS reports row base + j (rows are transposed into lane order through ystage below)
            const int64_t vr = base + (lane % WROWS);   // CSR (virtual) row of this lane
            const bool row_ok = FULL || vr < band_end;
            // y / x / w row of this lane: the virtual row itself, or the row it is a chunk of
            const int64_t rowc = VROWS ? (int64_t)vi.x : (row_ok ? vr : band_end - 1);
            const int64_t row = rowc;
            // implicit unit diagonal of the compact solver matrix (multi-GPU: added by the owner of the DOF only)
            double wv, xd;
            if constexpr (ABL & 16384) {   // the dot operand IS x (CG: p.Ap): one row load serves the dot and the diagonal
                const double xv = s.x[rowc];
                wv = xv;
                if constexpr (ABL & 8192)
                    xd = (s.unit_diag && s.owned[rowc]) ? xv : 0.0;
                else
                    xd = s.unit_diag ? xv : 0.0;
            } else {
                wv = (ABL & (8 | 64)) ? 1.0 : wp[rowc];
                if constexpr (ABL & 8192) {   // multi-GPU instantiation: ownership byte and x loaded unconditionally with the gathers
                    const uint8_t mine = s.owned[rowc];
                    const double xv = s.x[rowc];
                    xd = (s.unit_diag && mine) ? xv : 0.0;
                } else
                    xd = (s.unit_diag && !(s.owned && !s.owned[rowc])) ? s.x[rowc] : 0.0;
            }
            int rsn[U], ren[U];
            F64x2 vn[U];
            I32x2 cn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rsn[u] = __shfl(rp1, u * TEAMS + team, 64), ren[u] = __shfl(rp1, u * TEAMS + team + 1, 64);
                load_pair(rsn[u], ren[u], vn[u], cn[u]);
            }
            const v4i32_t tbn = load_tb(base + stride);
            const v2i32_t vin = load_vi(base + stride);
            rp1 = load_rp(base + 2 * stride);
            double acc[U];
            bool long_row = false;
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc[u] = v[u].x * xa[u] + v[u].y * xb[u], long_row |= re[u] - (ALIGNED ? (rs[u] & ~1) : rs[u]) > 2 * T;
            if (__any(long_row)) {   // rows longer than a team pass: further passes of 2 T entries, all U rows at once
                if constexpr (ALIGNED) {
                    int maxlen = 0;
#pragma unroll
                    for (int u = 0; u < U; ++u) maxlen = max(maxlen, re[u] - (rs[u] & ~1));
                    for (int off = 2 * T; __any(off < maxlen); off += 2 * T) {
                        // only lanes that still have entries issue loads (a clamped, unmasked load would fetch the next rows' data)
                        F64x2 tv[U];
                        I32x2 tc[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int k = (rs[u] & ~1) + off + 2 * l;
                            tv[u].x = tv[u].y = 0.0, tc[u].x = tc[u].y = 0;
                            if (k < re[u]) {
                                const v2f64_t a = *reinterpret_cast<const v2f64_t*>(s.vals + k);
                                tv[u].x = a.x, tv[u].y = k + 1 < re[u] ? a.y : 0.0;
                                bool coded = false;
                                if constexpr (C16) coded = tb.x >= 0;
                                if (coded) {
                                    const unsigned int code = *reinterpret_cast<const unsigned int*>(s.col16 + k);
                                    tc[u].x = decode(code & 0xffffu, tb), tc[u].y = decode(code >> 16, tb);
                                } else {
                                    const v2i32_t b = *reinterpret_cast<const v2i32_t*>(s.colidx + k);
                                    tc[u].x = b.x, tc[u].y = k + 1 < re[u] ? b.y : 0;
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int k = (rs[u] & ~1) + off + 2 * l;
                            if (k < re[u]) acc[u] += tv[u].x * s.x[tc[u].x] + tv[u].y * s.x[tc[u].y];
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        for (int k = rs[u] + l + 2 * T; k < re[u]; k += T) acc[u] += s.vals[k] * s.x[s.colidx[k]];
                }
            }
            // every lane of a team gets the team's U row sums; lane l < U keeps row (u = l, team) = tile row l*TEAMS + team.
            // Stored from there, consecutive lanes would write rows TEAMS apart: 32 separate 8-byte partial writes per
            // instruction (measured: as expensive as all the x gathers).  The sums are transposed into lane order through
            // a 256-byte per-wavefront LDS buffer (one ds_write_b64 + one ds_read_b64, wave-synchronous, no barrier), so
            // that lanes 0..WROWS-1 store WROWS consecutive rows = whole cache lines; lanes above repeat them.
            double pick = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double t = team_sum<T>(acc[u]);
                if (l == u) pick = t;
            }
            if (l < U) ys[l * TEAMS + team] = pick;
            __builtin_amdgcn_wave_barrier();
            double out;
            if constexpr (VROWS) {   // total of the row this lane's chunk belongs to (its chunks are adjacent in the tile)
                const int ck = vi.y & 255, cn = vi.y >> 8, j0 = (lane % WROWS) - ck;
                double t = 0;
                for (int d = 0; d < cn; ++d) t += ys[j0 + d];
                out = t + xd;
            } else
                out = ys[lane % WROWS] + xd;
            if constexpr (DEFER) {
                if (lane < WROWS) ydef[(wave * kDeferTiles + n_def) * WROWS + lane] = out;
                ++n_def;
            } else if constexpr (!(ABL & (8 | 32))) {
                if constexpr (FULL) {
                    // store flavours measured and dropped (all within noise of the plain store): nontemporal, write-through
                    // (sc1), 16 bytes per lane, stores confined to 32 KiB
                    s.y[row] = out;
                } else {
                    if (row_ok) s.y[row] = out;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const double once = (lane < WROWS && row_ok && (!VROWS || (vi.y & 255) == 0)) ? dots : 0.0;   // each row counted by one lane
            d_wy += once * (wv * out), d_yy += once * spmv_dot2(s, rowc, wv, out);
#pragma unroll
            for (int u = 0; u < U; ++u) rs[u] = rsn[u], re[u] = ren[u], c[u] = cn[u], v[u] = vn[u];
            tb = tbn, vi = vin;
        };
        const int64_t base0 = base;
        for (; base + WROWS <= band_end; base += stride) tile(std::true_type {});
        if (base < band_end) tile(std::false_type {});
        if constexpr (DEFER) {
            __builtin_amdgcn_wave_barrier();
            for (int t = 0; t < n_def; ++t) {
                const int64_t row = base0 + t * stride + lane;
                if (lane < WROWS && row < band_end) s.y[row] = ydef[(wave * kDeferTiles + t) * WROWS + lane];
            }
        }
    }
    if (s.partial) {
        const double a = block_sum(d_wy, red);
        const double b = block_sum(d_yy, red);
        if (threadIdx.x == 0) s.partial[2 * blockIdx.x] = a, s.partial[2 * blockIdx.x + 1] = b;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_spmv_c16p: the production SpMV of the Krylov solvers on the compact solver matrix (implicit unit diagonal, 16-bit column
// codes, aligned entry pairs).  Same tiling, team sums and transposed y store as k_spmv_team2, but a software pipeline that is
// one stage deeper: the x gathers of a tile are issued ONE TILE AHEAD of their use.  In k_spmv_team2 every tile pays the
// gather round trip serially (codes arrive -> decode -> gather -> wait -> FMA); diagnostic builds show that this wait, not
// the gathered bytes or lines, is what the gathers cost (all lanes of a gather forced into ONE cache line: 56.5 us, real
// gathers 59.3 us, no gathers 48.9 us on C3).  Per iteration i of the tile loop, in issue order (loads return in order):
//     a. column codes, window bases of tile i+2 and row pointers of tile i+3
//     c. decode the codes of tile i+1 (loaded during iteration i-1), issue its x gathers and its row operands
//     d. matrix values of tile i+1
//     e. wait for the gathers and values of tile i (issued during iteration i-1), FMA, team sums, y store, dots
// FLAGS: 8192 = multi-GPU (implicit diagonal and w.w counted by the owner of the row), 16384 = the dot operand w is x.
// ---------------------------------------------------------------------------------------------------------------
template <int T, int U, int FLAGS>
static __global__ __launch_bounds__(256) void k_spmv_c16p(SpmvArgs s, int64_t n, int64_t rows_per_band) {
    constexpr int TEAMS = 64 / T;
    constexpr int WROWS = TEAMS * U;
    constexpr bool DIST = (FLAGS & 8192) != 0, WX = (FLAGS & 16384) != 0;
    static_assert(WROWS < 64 && 32 % WROWS == 0, "one rowptr load per tile, tiles inside a 32-row code group");
    typedef int v4i32_t __attribute__((ext_vector_type(4)));
    __shared__ double red[8];
    __shared__ double ystage[4][WROWS];
    if (s.stop && __syncthreads_or(*s.stop != 0)) return;
    const int band = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, team = lane / T, l = lane % T;
    const int64_t band_begin = band * rows_per_band;
    const int64_t band_end = min(n, band_begin + rows_per_band);
    const int64_t stride = (int64_t)bpx * 4 * WROWS;
    const int last = s.nnz - 1, ncol1 = s.n_cols - 1;
    double d_wy = 0, d_yy = 0;
    auto load_rp = [&](int64_t b) -> int {
        const int64_t r = b + lane;
        return s.rowptr[r < band_end ? r : band_end];
    };
    auto load_tb = [&](int64_t b) -> v4i32_t {
        const int64_t bc = b < band_end ? b : band_begin;
        const int g = __builtin_amdgcn_readfirstlane((int)(bc >> 5));
        return *reinterpret_cast<const v4i32_t*>(s.tbase + 4 * (int64_t)g);
    };
    auto pair_at = [&](int rs) -> int {   // aligned pair of lane l in a row starting at rs, clamped into the arrays
        const int k = (rs & ~1) + 2 * l;
        return k < last ? k : (last & ~1);
    };
    auto load_codes = [&](int rs) -> unsigned int {
        return *reinterpret_cast<const unsigned int*>(s.col16 + pair_at(rs));
    };
    auto load_vals = [&](int rs, int re, F64x2& v) {
        const int k = (rs & ~1) + 2 * l;
        const v2f64_t a = *reinterpret_cast<const v2f64_t*>(s.vals + pair_at(rs));
        v.x = (k >= rs && k < re) ? a.x : 0.0, v.y = (k + 1 < re) ? a.y : 0.0;
    };
    auto decode = [&](unsigned int code, const v4i32_t& tb) -> int {
        const int b01 = (code & 0x4000u) ? tb.y : tb.x, b23 = (code & 0x4000u) ? tb.w : tb.z;
        const int col = ((code & 0x8000u) ? b23 : b01) + (int)(code & 0x3fffu);
        return col < ncol1 ? col : ncol1;   // entries of neighbouring rows (masked, value 0) may decode out of range
    };
    // columns of a tile -> its x gathers (wide groups, wave-uniform and rare, re-read the 32-bit columns)
    auto gather = [&](const unsigned int (&code)[U], const int (&rs)[U], const v4i32_t& tb, double (&xa)[U], double (&xb)[U]) {
        int ca[U], cb[U];
        if (tb.x < 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const v2i32_t b = *reinterpret_cast<const v2i32_t*>(s.colidx + pair_at(rs[u]));
                ca[u] = b.x, cb[u] = b.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) ca[u] = decode(code[u] & 0xffffu, tb), cb[u] = decode(code[u] >> 16, tb);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xa[u] = s.x[ca[u]], xb[u] = s.x[cb[u]];
    };
    // row operands of lane j < WROWS (row base + j): x for the implicit diagonal, w for the dot, ownership (multi-GPU)
    const double* wp = s.w ? s.w : s.x;
    const double dots = s.w ? 1.0 : 0.0;
    auto row_ops = [&](int64_t b, double& xr, double& wr, int& mine) {
        const int64_t row = b + (lane % WROWS);
        const int64_t rowc = row < band_end ? row : band_end - 1;
        xr = s.x[rowc];
        if constexpr (WX) wr = xr; else wr = wp[rowc];
        if constexpr (DIST) mine = s.owned[rowc]; else mine = 1;
    };
    int64_t base = band_begin + (int64_t)(lb * 4 + wave) * WROWS;
    if (base < band_end) {
        // ---- prologue: tile 0 fully loaded and gathered, codes of tile 1, row pointers of tile 2
        int rs[U], re[U], rsn[U], ren[U];
        unsigned int cn[U];
        F64x2 v[U];
        double xa[U], xb[U], xr, wr;
        int mine;
        int rp2;
        v4i32_t tbn;
        {
            const int rp0 = load_rp(base), rp1 = load_rp(base + stride);
            rp2 = load_rp(base + 2 * stride);
            const v4i32_t tb0 = load_tb(base);
            tbn = load_tb(base + stride);
            unsigned int c0[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rs[u] = __shfl(rp0, u * TEAMS + team, 64), re[u] = __shfl(rp0, u * TEAMS + team + 1, 64);
                c0[u] = load_codes(rs[u]);
                load_vals(rs[u], re[u], v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rsn[u] = __shfl(rp1, u * TEAMS + team, 64), ren[u] = __shfl(rp1, u * TEAMS + team + 1, 64);
                cn[u] = load_codes(rsn[u]);
            }
            gather(c0, rs, tb0, xa, xb);
            row_ops(base, xr, wr, mine);
        }
        // explicit LDS address space: a volatile GENERIC pointer compiles to flat_load / flat_store, which count on vmcnt and
        // made every tile drain all of its prefetched loads (s_waitcnt vmcnt(0))
        lds_vf64_t* ys = (lds_vf64_t*)&ystage[wave][0];
        auto tile = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            // a. tile i+2: row pointers -> codes, window bases; row pointers of tile i+3
            int rs2[U], re2[U];
            unsigned int c2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rs2[u] = __shfl(rp2, u * TEAMS + team, 64), re2[u] = __shfl(rp2, u * TEAMS + team + 1, 64);
                c2[u] = load_codes(rs2[u]);
            }
            const v4i32_t tb2 = load_tb(base + 2 * stride);
            rp2 = load_rp(base + 3 * stride);
            // c. tile i+1: gathers and row operands
            double xan[U], xbn[U], xrn, wrn;
            int minen;
            gather(cn, rsn, tbn, xan, xbn);
            row_ops(base + stride, xrn, wrn, minen);
            // d. tile i+1: values
            F64x2 vn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) load_vals(rsn[u], ren[u], vn[u]);
            // e. tile i
            const int64_t row = base + (lane % WROWS);
            const bool row_ok = FULL || row < band_end;
            double acc[U];
            bool long_row = false;
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = v[u].x * xa[u] + v[u].y * xb[u], long_row |= re[u] - (rs[u] & ~1) > 2 * T;
            if (__any(long_row)) {   // rows longer than a team pass (rare when 2 T covers the mean row)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    for (int k = (rs[u] & ~1) + l + 2 * T; k < re[u]; k += T) acc[u] += s.vals[k] * s.x[s.colidx[k]];
            }
            double pick = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double t = team_sum<T>(acc[u]);
                if (l == u) pick = t;
            }
            if (l < U) ys[l * TEAMS + team] = pick;
            __builtin_amdgcn_wave_barrier();
            const double out = ys[lane % WROWS] + (mine ? xr : 0.0);   // + implicit unit diagonal (owner only on several GPUs)
            if constexpr (FULL)
                s.y[row] = out;   // lanes >= WROWS repeat lanes < WROWS: unconditional store, no exec-masked branch
            else if (row_ok)
                s.y[row] = out;
            __builtin_amdgcn_wave_barrier();
            const double once = (lane < WROWS && row_ok) ? dots : 0.0;   // each row counted by one lane
            d_wy += once * (wr * out);
            d_yy += once * (s.dot2_ww ? (mine ? wr * wr : 0.0) : out * out);
            // rotate the pipeline registers
#pragma unroll
            for (int u = 0; u < U; ++u)
                rs[u] = rsn[u], re[u] = ren[u], v[u] = vn[u], xa[u] = xan[u], xb[u] = xbn[u], rsn[u] = rs2[u], ren[u] = re2[u],
                cn[u] = c2[u];
            xr = xrn, wr = wrn, mine = minen, tbn = tb2;
        };
        for (; base + WROWS <= band_end; base += stride) tile(std::true_type {});
        if (base < band_end) tile(std::false_type {});
    }
    if (s.partial) {
        const double a = block_sum(d_wy, red);
        const double b = block_sum(d_yy, red);
        if (threadIdx.x == 0) s.partial[2 * blockIdx.x] = a, s.partial[2 * blockIdx.x + 1] = b;
    }
}

}  // namespace fdapde_hip
#endif
