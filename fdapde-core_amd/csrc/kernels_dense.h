// kernels_dense.h -- "factor once" that pays PER SOLVE on small systems: the dense inverse of an FEM matrix, and its application.
//
// The reference's users hold fdapde::SparseLU handles (fdaPDE/utils/symbols.h:133-160), factorise once and then solve column after column --
// the parabolic loop (fem_linear_parabolic_solver.h:56-68: one compute, a triangular solve per step), SMW (linear_algebra/smw.h:38-59),
// every downstream model.  At the reference's own sizes (289 .. a few thousand DOFs) a back-substitution costs 10 - 50 us on a CPU; a Krylov
// run per column costs 150 - 400 us on the GPU whatever the kernel quality (tens of iterations x hand-off latency).  For systems of up to a
// few thousand rows the device can do better than either: invert once, then ONE dense matrix-vector product per column.
//
//   k_dense_fill      the matrix of the sparse pattern (optionally with the Dirichlet rows zeroed and a unit diagonal: the reference's
//                     set_dirichlet_bc matrix, fem_solver_base.h:142-155) as a dense row-major n x n array
//   k_dense_invert    in-place Gauss-Jordan with partial pivoting as ONE launch of <= one workgroup per CU: row i lives with workgroup i mod G;
//                     per elimination step ONE grid barrier (agent-scope release / acquire, cdna_hip_programming.md guideline 16): every
//                     workgroup copies the pivot row to LDS, updates its rows, and -- in the same sweep -- finds its candidate for the NEXT
//                     column's pivot.  Rows are never swapped (a permutation is recorded), the pivot row's own scaling is deferred to its
//                     owner's next visit, so nobody writes a row somebody else may still be reading.
//   k_dense_invert_blocked  the default: the same elimination 16 pivots at a time -- a panel workgroup that factorises in registers one panel ahead, the update
//                     of everything else as 16 x 16 tiles on the f64 matrix cores with the panel's own final columns as the operand (described at the kernel)
//   k_dense_unpermute the inverse in natural row / column order from the in-place result and the pivot sequence
//   k_dense_check     max |I - A X| through the sparse rows of A (decides whether a solve adds one step of iterative refinement)
//   k_dense_stage / k_dense_gemv / k_dense_residual / k_dense_out   a solve: right-hand sides from pinned host memory into internal order,
//                     x = X b (one wavefront per row, the columns in tiles of NC), optionally r = b - A x and x += X r, the result back in the
//                     reference numbering into pinned host memory and a completion word the host spins on
//   k_dense_xm / k_dense_step_cols / k_dense_step   the implicit Euler stepper folded into one product per step (B = X D M / dt once, then u' = B u + c)
// The products are HBM / cache-bound streaming of X (n^2 doubles per column tile): no MFMA -- a GEMV has no reuse to feed one; the inversion's update is a
// 16-deep product per tile and runs on v_mfma_f64_16x16x4.
#ifndef FDAPDE_KERNELS_DENSE_H
#define FDAPDE_KERNELS_DENSE_H

#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels_reduce.h"

namespace fdapde_hip {

constexpr int kDenseT = 512;        // threads of an inversion workgroup (8 wavefronts: 8 rows in flight)
constexpr int kDenseMaxRows = 8192; // pivot row in LDS: 64 KB

typedef __attribute__((address_space(1))) unsigned int dn_u32;
typedef __attribute__((address_space(1))) unsigned long long dn_u64;

static __global__ void k_dense_fill(int64_t n, int64_t ld, const int32_t* rowptr, const int32_t* colidx, const double* vals, const uint8_t* bnd, int use_bnd, double* D) {
    const int64_t i = blockIdx.x;
    double* row = D + i * ld;
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) row[j] = 0.0;
    __syncthreads();
    if (use_bnd && bnd[i]) {
        if (threadIdx.x == 0) row[i] = 1.0;
        return;
    }
    for (int32_t k = rowptr[i] + (int32_t)threadIdx.x; k < rowptr[i + 1]; k += (int32_t)blockDim.x) row[colidx[k]] = vals[k];
}

struct DenseInvArgs {
    int32_t n, G, ld;
    double* S;                     // n rows of stride ld (a multiple of 16 doubles: no cache line holds entries of two rows), inverted in place (up to the permutation)
    int32_t* perm;                 // [n] pivot row of step k
    unsigned long long* cand;      // [2][G] one tagged granule per workgroup and step parity (zeroed before the launch)
    int32_t* status;               // [0] 1 = singular (no usable pivot), 2 = a sweep timed out; zeroed before the launch
    long long timeout_ticks;       // bound of a barrier wait (s_memrealtime ticks, 100 MHz)
};

// Everything one workgroup writes and another reads in this launch -- the rows, the candidates -- is stored write-through (sc1: relaxed agent-scope
// atomic stores of 8 bytes) and read past the CU's L1 (sc1 loads); a row is only ever read by its owner (plain loads: same wavefront, same CU as the
// stores) and, ONCE, as the pivot row of its step by everybody (sc1 loads).  The step's barrier IS the candidate exchange (cdna_hip_programming.md
// guideline 16, form R2 "the data is the flag"): every workgroup publishes ONE 8-byte granule per step into a slot of its own -- [step + 1 : 18 bits |
// row : 14 bits | float magnitude of the candidate : 32 bits] -- after all its waves have drained their row stores, and one wavefront per workgroup
// sweeps the G granules until every tag says "this step".  No counter that G workgroups hammer with atomics and polls (that form: 16 us per step at
// 137 workgroups), no cache write-back / invalidate.  Slots alternate between two arrays by step parity: a workgroup can be at most one step ahead
// of the slowest one.  A float magnitude is enough to choose a pivot (any of the near-largest entries will do); every workgroup reads the same
// granules, so all of them choose the same row.
__device__ __forceinline__ unsigned long long dense_granule(int step, int row, double mag) {
    const float f = (float)mag;   // (monotone; a double beyond float's range becomes inf / 0 -- still ordered)
    return ((unsigned long long)(unsigned)(step + 1) << 46) | ((unsigned long long)(unsigned)(row & 0x3fff) << 32) | (unsigned long long)__float_as_uint(f);
}
__device__ __forceinline__ void dense_store(double* p, double v) {
    __hip_atomic_store((dn_u64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
}
__device__ __forceinline__ double dense_load_shared(const double* p) {   // a value another workgroup wrote in this launch
    return __longlong_as_double((long long)__hip_atomic_load((const dn_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // sc1: past the L1
}

// In-place Gauss-Jordan, pivot (p, k) of step k, no row swaps:  f_i = S[i][k] / d,  S[i][j] -= f_i S[p][j] (j != k),  S[i][k] = -f_i  for i != p;
// S[p][j] /= d (j != k), S[p][k] = 1 / d.  After n steps column k of the storage holds column p_k of (E = the product of the row operations), and
// A^-1 = P^T E:  A^-1[k][p_j] = S[p_k][j]  (k_dense_unpermute).
static __global__ __launch_bounds__(kDenseT) void k_dense_invert(DenseInvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dn_smem[];
    double* prow = reinterpret_cast<double*>(dn_smem);                         // [n] the pivot row of the step, unscaled
    constexpr int W = kDenseT / 64;
    __shared__ unsigned long long red_val[W];
    __shared__ int red_row[W];
    __shared__ int piv_s;
    // the pivot row this workgroup owns whose scaling is still owed: applied when its owner next visits it -- the very next step
    __shared__ int pend_row_s, pend_k_s;
    __shared__ double pend_d_s;
    __shared__ unsigned long long used_bits[(kDenseMaxRows + 63) / 64];   // rows of this workgroup that have been pivots (bit per local row)
    const int n = a.n, G = a.G, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int my_rows = g < n ? (n - g + G - 1) / G : 0;
    for (int w = tid; w < (my_rows + 63) / 64; w += kDenseT) used_bits[w] = 0ull;
    if (tid == 0) pend_row_s = -1, pend_k_s = -1, pend_d_s = 1.0;
    __syncthreads();
    auto better = [](double v, int row, double bv, int brow) { return v > bv || (v == bv && row < brow); };
    // this workgroup's candidate for the pivot of column `col`: the best of its waves, published as the granule of step `col`
    auto publish = [&](double best, int best_row, int col) {
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o);
            const int orow = __shfl_xor(best_row, o);
            if (better(ov, orow, best, best_row)) best = ov, best_row = orow;
        }
        if (lane == 0) red_val[wave] = (unsigned long long)__double_as_longlong(best), red_row[wave] = best_row;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's row stores are out before the granule says so
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < W; ++w) {
                const double wv = __longlong_as_double((long long)red_val[w]);
                if (better(wv, red_row[w], best, best_row)) best = wv, best_row = red_row[w];
            }
            __hip_atomic_store((dn_u64*)(a.cand + (size_t)(col & 1) * G + g), dense_granule(col, best_row == 0x7fffffff ? 0x3fff : best_row, best), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    {   // candidates for column 0
        double best = -1.0;
        int best_row = 0x7fffffff;
        for (int r = tid; r < my_rows; r += kDenseT) {
            const int i = g + G * r;
            const double v = fabs(a.S[(int64_t)i * a.ld]);
            if (better(v, i, best, best_row)) best = v, best_row = i;
        }
        publish(best, best_row, 0);
    }
    for (int k = 0; k < n; ++k) {
        // ---- the barrier and the pivot of column k in one sweep: wait until all G granules carry this step's tag, take the best of them
        if (wave == 0) {
            const unsigned long long* cv = a.cand + (size_t)(k & 1) * G;
            const unsigned long long want = (unsigned long long)(unsigned)(k + 1);
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            float best = -1.0f;
            int best_row = 0x7fffffff;
            int ok = 1;
            for (int q = lane; q < G; q += 64) {
                unsigned long long x;
                while (((x = __hip_atomic_load((const dn_u64*)(cv + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 46) != want) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                        ok = 0;
                        break;
                    }
                }
                if (!ok) break;
                const int row = (int)((x >> 32) & 0x3fff);
                const float v = __uint_as_float((unsigned)x);
                if (row != 0x3fff && (v > best || (v == best && row < best_row))) best = v, best_row = row;
            }
            ok = __all(ok);
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o);
                const int orow = __shfl_xor(best_row, o);
                if (ov > best || (ov == best && orow < best_row)) best = ov, best_row = orow;
            }
            if (lane == 0) {
                if (!ok) a.status[0] = 2;
                piv_s = !ok ? -2 : (best_row == 0x7fffffff || !(best > 0.0f) || !isfinite(best)) ? -1 : best_row;
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the compiler from moving loads above the sweep)
        const int p = piv_s;
        if (p < 0) {   // no usable pivot: the matrix is singular to working precision (every workgroup takes this exit together); or a sweep timed out
            if (p == -1 && g == 0 && tid == 0) a.status[0] = 1;
            return;
        }
        for (int j = tid; j < n; j += kDenseT) prow[j] = dense_load_shared(a.S + (int64_t)p * a.ld + j);   // (row p is nobody's to write in this step: its scaling is deferred)
        const int pend_row = pend_row_s, pend_k = pend_k_s;
        const double pend_sc = 1.0 / pend_d_s;
        __syncthreads();
        const double d = prow[k];
        const double inv_d = 1.0 / d;
        // ---- the rows of this workgroup, a wavefront per row; the lane that holds column k + 1 proposes the next pivot
        double best = -1.0;
        int best_row = 0x7fffffff;
        for (int r = wave; r < my_rows; r += W) {
            const int i = g + G * r;
            if (i == p) continue;
            double* row = a.S + (int64_t)i * a.ld;
            const bool owed = i == pend_row;
            double sik = row[k];
            if (owed) sik = sik * pend_sc;   // (k != pend_k: that column was replaced in its own step)
            const double f = sik * inv_d;
            const bool was_pivot = (used_bits[r >> 6] >> (r & 63)) & 1ull;
            for (int j = lane; j < n; j += 64) {
                double v = row[j];
                if (owed) v = (j == pend_k ? 1.0 : v) * pend_sc;
                const double nv = j == k ? -f : v - f * prow[j];
                dense_store(row + j, nv);
                if (j == k + 1 && !was_pivot && better(fabs(nv), i, best, best_row)) best = fabs(nv), best_row = i;
            }
        }
        __syncthreads();
        if (tid == 0) {
            const bool mine = (p % G) == g;
            if (mine) {
                a.perm[k] = p;
                const int r = (p - g) / G;
                used_bits[r >> 6] |= 1ull << (r & 63);
                pend_row_s = p, pend_k_s = k, pend_d_s = d;
            } else
                pend_row_s = -1;
        }
        if (k + 1 < n) publish(best, best_row, k + 1);   // (its __syncthreads also orders the update above before the next step's reads)
        else __syncthreads();
    }
    // the last pivot row's scaling is still owed
    if (pend_row_s >= 0) {
        double* row = a.S + (int64_t)pend_row_s * a.ld;
        const double sc = 1.0 / pend_d_s;
        const int pk = pend_k_s;
        for (int j = tid; j < n; j += kDenseT) row[j] = (j == pk ? 1.0 : row[j]) * sc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same inversion BLOCKED: nb pivots per grid-wide exchange instead of one.  k_dense_invert above pays a chain of dependent trips through the
// fabric per pivot (sweep, pivot row, write-through, granule: ~9 us + 5 ns per workgroup -- 15 ms for 1 089 rows).  Here a PANEL of nb <= 16
// columns lives in the registers of ONE workgroup (the last of the grid), which factorises it with workgroup barriers only -- nb pivot searches and
// rank-1 updates -- and every other workgroup applies the nb steps at once to its block of the matrix.  What makes that cheap: in-place Gauss-Jordan
// leaves in the panel's columns the row operations themselves.  After the panel's steps, E[i][u] = S'[i][k0 + u] is the coefficient of the ORIGINAL
// pivot row p_u in the new row i (one step: S'[i][j] = S[i][j] + S'[i][k] S[p][j] with S'[i][k] = -f_i, and S'[p][j] = S'[p][k] S[p][j] with
// S'[p][k] = 1 / d; by induction over the steps).  So
//     S'[i][j] = base_i[j] + sum_u E[i][u] S[p_u][j],     base_i = S[i][.] for ordinary rows, 0 for the panel's pivot rows
// -- a 16-deep matrix product per 16 x 16 tile, on the matrix cores (v_mfma_f64_16x16x4); nothing to publish but the panel's final columns (which the
// result needs anyway) and the pivots' indices.
//   (above 2 048 rows there are several panel workgroups, 1 536 rows of the panel each, and a pivot step's choice is agreed over the fabric: MULTI)
//   the panel workgroup  factorises panel P while the others still apply panel P - 1: the columns of panel P as panel P - 1 leaves them it computes
//                        ITSELF (the same product: it holds E of panel P - 1 and reads 16 x 16 entries of the pivot rows), so the factorisations
//                        follow each other without a gap; it waits for "update P - 1 done everywhere" only before it publishes panel P
//   the workers          an R x C grid of blocks, one workgroup each: wait for "panel P published", read their rows of E (one trip) and the pivots,
//                        16 x 16 tiles (a column tile per wavefront, the pivot-row fragment once, four row tiles' loads in flight), skip the columns of
//                        panels P and P + 1 (the panel workgroup's), say "done".  Blocks, not whole rows: every load comes over the fabric (sc1: another
//                        XCD wrote the line), and with whole rows every workgroup reads the 16 pivot rows in full -- 69 of a panel's 108 MB at 2 116 rows.
// A panel reads one buffer and writes the other (nobody writes what somebody may still read).  Payload through sc1 buffer stores / loads, flags as
// in k_dense_invert.  History of this kernel (1 089 rows): LDS-resident panel 15 ms; register panel + scalar update with published multipliers 6.8 ms
// (3.7 us per pivot step: spills whose reloads waited for the step's write-through stores; 38 us per panel in LDS broadcasts of the update);
// this form 2.3 ms (0.4 ms at 289 rows, 18 ms at 4 225, 0.11 s at 8 100).
// ---------------------------------------------------------------------------------------------------------------------------------------
// a workgroup barrier for hand-offs through LDS only: waits for this wave's LDS traffic, not for its global stores in flight
__device__ __forceinline__ void dense_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and between the lanes of ONE wavefront (their LDS operations complete in order)
__device__ __forceinline__ void dense_lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kDenseNB = 16;    // pivots per panel at most
constexpr int kDenseTB = 512;   // threads of a workgroup of the blocked inversion: 8 wavefronts, 2 per SIMD -- 256 vector registers each: the panel (up to 64 doubles per
                                // thread) AND a pivot step's working set stay in registers.  (1 024 threads: 128 registers, spills whose reloads' s_waitcnt vmcnt(0) also
                                // waited for write-through stores in flight -- 3.7 us per pivot step; 256 threads: five panel rows per thread at 1 089 rows.)

struct DenseBlkArgs {
    int32_t n, G, ld, nb;          // G worker workgroups; the grid is G + 1, the last one the panel workgroup
    int32_t C, RB, CB;             // the update's grid of blocks: G = R x C workgroups, RB rows x CB columns each (multiples of 16)
    int32_t KP;                    // panel workgroups (the grid is G + KP): 1, or -- above 4 096 rows -- one per 1 536 rows of the panel, a pivot step's choice agreed through xch
    double *S0, *S1;               // panel P reads S(P & 1), writes the other
    int32_t* perm;                 // [n] pivot row of column k
    int32_t* piv_row;              // [nb]
    unsigned long long* done;      // [G] granule per worker: panels whose update it has finished
    unsigned long long* ready;     // [1] (panels published << 1) | failed
    unsigned long long* xch;       // [2][KP][40] KP > 1: a pivot step's candidates, tagged granules {step : 32 | half a double : 32}: the key, then the candidate row's nb entries
    unsigned long long* pdone;     // [KP] KP > 1: (panels whose columns panel workgroup k has written out << 1) | failed
    int32_t* status;               // [0] 1 = singular, 2 = a wait timed out
    long long timeout_ticks;
    long long* stamps;             // diagnostic (FDAPDE_DENSE_STAMPS): s_memrealtime at the phase boundaries of panel 8, [0..7] the panel workgroup, [8..15] worker 0,
                                   // [16..23] inside pivot step 5; or null
};

// payload loads / stores of the blocked inversion: BUFFER instructions with the sc1 bit (aux 16).  Against the 8-byte atomics above they (a) address
// through a resource in scalar registers + a 32-bit offset (the atomics' 64-bit addresses, hoisted out of the unrolled pivot steps, filled 64 vector
// registers and spilled) and (b) are ordinary memory operations to the compiler: it issues a batch of loads back to back instead of one round trip
// per load.
typedef unsigned int dn_v2u __attribute__((ext_vector_type(2)));
typedef double dn_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double dense_bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const dn_v2u x = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 16);
    return __longlong_as_double((long long)(((unsigned long long)x[1] << 32) | x[0]));
}
__device__ __forceinline__ void dense_bstore(__amdgpu_buffer_rsrc_t rs, int voff, int soff, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    dn_v2u x;
    x[0] = (unsigned)b, x[1] = (unsigned)(b >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(x, rs, voff, soff, 16);
}
// maximum over the wavefront in every lane, on the VALU (permlane swaps + DPP: kernels_reduce.h) -- a __shfl_xor butterfly of (value, row) is 18
// ds_bpermute per pivot step
__device__ __forceinline__ double dense_wave_max(double v) {
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        v = fmax(__longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])), __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1])));
    }
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        v = fmax(__longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])), __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1])));
    }
    v = fmax(v, reduce_dpp_f64<0x128>(v));
    v = fmax(v, reduce_dpp_f64<0x124>(v));
    v = fmax(v, reduce_dpp_f64<0x4E>(v));
    v = fmax(v, reduce_dpp_f64<0xB1>(v));
    return v;
}
// A pivot candidate as ONE double that orders like (magnitude, then smaller row): the magnitude's low 13 bits replaced by 8 191 - row.  Partial pivoting
// wants a near-largest entry, not the largest to the last bit (2^-39 relative here), and the tie rule stays "the smallest row"; one reduction per
// choice instead of a maximum followed by a minimum over the rows that have it.  No candidate: -1.  (NaN entries never win: v_max_f64 drops them.)
__device__ __forceinline__ double dense_cand_key(double mag, int row) {
    return __longlong_as_double((long long)(((unsigned long long)__double_as_longlong(mag) & ~0x1fffull) | (unsigned long long)(0x1fff - row)));
}
__device__ __forceinline__ int dense_cand_row(double key) { return 0x1fff - (int)((unsigned long long)__double_as_longlong(key) & 0x1fffull); }
__device__ __forceinline__ double dense_cand_mag(double key) { return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(key) & ~0x1fffull)); }

template <int RPT, int NBT, bool MULTI> static __global__ __launch_bounds__(kDenseTB) void k_dense_invert_blocked(DenseBlkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dn_smem3[];
    double* buf = reinterpret_cast<double*>(dn_smem3);
    constexpr int T = kDenseTB, W = kDenseTB / 64;
    static_assert(W <= 64, "the pivot scan below reads a candidate per lane");
    static_assert(kDenseMaxRows <= 0x2000, "a candidate key holds 13 bits of row");
    __shared__ int piv_row_s[kDenseNB], wait_s;
    __shared__ __attribute__((aligned(16))) double cand_key[2][W], cand_rows[2][W][kDenseNB];
    __shared__ unsigned char isp_s[kDenseTB];
    __shared__ unsigned xh_s[6 * 40];   // KP > 1: the halves collected from xch
    const int n = a.n, G = a.G, NB = a.nb, ld = a.ld, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(a.S0, 0, n * ld * 8, 0x00020000);   // (<= 8 192 x 8 192 doubles: 2^29 bytes)
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(a.S1, 0, n * ld * 8, 0x00020000);
    const int n_panels = (n + NB - 1) / NB;
    const int q = lane >> 4, jl = lane & 15;   // MFMA operand layouts (measured): A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[4 v + q][j] in register v of lane 16 q + j
    auto sstamp = [&](int P, int t, int slot) {   // inside pivot step 5 of panel 8
        if (a.stamps && P == 8 && t == 5 && tid == 0 && g == G) a.stamps[16 + slot] = (long long)__builtin_amdgcn_s_memrealtime();
    };
    auto stamp = [&](int P, int who, int slot) {
        if (a.stamps && P == 8 && tid == 0 && (g == 0 || g == G)) a.stamps[who * 8 + slot] = (long long)__builtin_amdgcn_s_memrealtime();
    };

    if (g >= G) {
        // =================================================================== the panel workgroup(s)
        const int pk = MULTI ? g - G : 0, KP = MULTI ? a.KP : 1, row0 = pk * RPT * T;   // this one holds rows row0 .. row0 + RPT T - 1 of the panel
        // The panel lives in REGISTERS: thread tid holds rows tid, tid + T, ... (RPT of them) x NBT columns.  A step is a register scan for the pivot
        // (+ one block-wide arg-max), the pivot row's NBT values through LDS, and NBT fused multiply-adds per row and thread -- an LDS-resident panel
        // cost 5 - 10 us per step (hundreds of LDS read-modify-writes per thread).  Rows move between the register layout (a row per thread) and
        // memory / the MFMA layouts through an LDS slab (T x NBT doubles, row stride NBT + 1) in which every wavefront only touches ITS 64 rows: no
        // workgroup barrier outside the pivot steps.
        double* slab = buf + (size_t)wave * 64 * (NBT + 1);   // this wavefront's 64 rows
        constexpr int SL = NBT + 1;
        double pr[RPT][NBT];
        unsigned used_mine = 0u;   // bit r = this thread's row tid + r T has been a pivot
        // panel 0: the matrix's first columns as they are
        {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                dense_lds_wave_sync();
#pragma unroll
                for (int u = 0; u < NBT; ++u) {   // (16 lanes per row: whole 128-byte lines; NBT loads in flight)
                    const int rl = (64 / NBT) * u + lane / NBT, cl = lane % NBT, i = row0 + r * T + wave * 64 + rl;
                    const double v = dense_bload(rs0, ((i < n ? i : 0) * ld + cl) * 8, 0);
                    slab[rl * SL + cl] = (i < n && cl < (n < NB ? n : NB)) ? v : 0.0;
                }
                dense_lds_wave_sync();
#pragma unroll
                for (int t = 0; t < NBT; ++t) pr[r][t] = slab[lane * SL + t];
            }
        }
        for (int P = 0; P < n_panels; ++P) {
            const int k0 = P * NB, nbp = n - k0 < NB ? n - k0 : NB;
            const __amdgpu_buffer_rsrc_t rs_cur = (P & 1) ? rs1 : rs0, rs_next = (P & 1) ? rs0 : rs1;
            stamp(P, 0, 0);   // panel in registers
            // (an opaque copy of the thread index: everything derived from tid below is loop-invariant, and hoisted out of the panel loop it fills a
            // hundred registers that are then spilled and reloaded inside every pivot step)
            int ft = tid;
            asm volatile("" : "+v"(ft));
            int failed = 0;
            unsigned piv_now = 0u;   // bit r: the row is a pivot of THIS panel
            // ONE barrier per pivot step: every wavefront reduces its candidates (VALU), the lane that owns the wavefront's best row publishes the key AND
            // the row's NBT panel entries (arrays double-buffered by step parity); after the barrier every wavefront reduces the W keys, knows the pivot
            // and reads the pivot row -- no second round for "who won" and "hand me the row".  The largest magnitude wins, the smallest row among equals.
            // The steps run in GROUPS of four: inside a group they are unrolled (step s works on register position s, static indices into the panel), the
            // loop over the groups is not, and after a group the panel's columns ROTATE by four positions so that the next group finds its columns at
            // positions 0 .. 3 again.  (All sixteen steps unrolled were 60 - 90 KB of code -- with the exchange of several panel workgroups in them, past
            // the instruction cache: 9 - 13 us per step instead of 1; one step per iteration with a rotation by one cost 85 register moves per step.)
            // After the loop position c holds column (c + rot) mod NBT: whoever copies the panel out undoes that.
            constexpr int GS = MULTI ? 4 : NBT;   // (one panel workgroup: the whole panel unrolled -- its steps fit the instruction cache -- and no rotation)
            int rot = 0;
#pragma nounroll
            for (int t0 = 0; t0 < nbp; t0 += GS) {
#pragma unroll
              for (int s = 0; s < GS; ++s) {
                const int t = t0 + s;
                if (t < nbp && !failed) {
                    double best = -1.0;
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        const int i = row0 + ft + r * T;
                        const double kd = (i < n && !((used_mine >> r) & 1u)) ? dense_cand_key(fabs(pr[r][s]), i) : -1.0;
                        best = fmax(best, kd);
                    }
                    const int par = s & 1;
                    sstamp(P, t, 0);
                    const double wmax = dense_wave_max(best);
                    if (!(wmax >= 0.0)) {
                        if (lane == 0) cand_key[par][wave] = -1.0;
                    } else if (best == wmax) {   // (rows are unique to a lane, so are keys)
                        cand_key[par][wave] = wmax;
                        const int rp = (dense_cand_row(wmax) - row0) / T;
#pragma unroll
                        for (int r = 0; r < RPT; ++r)
                            if (r == rp) {
#pragma unroll
                                for (int tt = 0; tt < NBT; ++tt) cand_rows[par][wave][tt] = pr[r][tt];
                            }
                    }
                    sstamp(P, t, 1);
                    dense_lds_barrier();
                    sstamp(P, t, 2);
                    const double ck = lane < W ? cand_key[par][lane] : -1.0;
                    double kb = dense_wave_max(ck);
                    const int ws = __ffsll((unsigned long long)__ballot(lane < W && ck == kb)) - 1;
                    double pw[NBT];   // the pivot row's panel entries (LDS, the same address in every lane: broadcast reads), once per step
                    if (MULTI && KP > 1) {
                        // several panel workgroups: every one publishes its best -- the key and the row's entries as tagged 8-byte granules, "the data is
                        // the flag" -- and collects everybody's; the largest key wins everywhere.  One trip over the fabric and a second barrier per step.
                        constexpr int S = 2 * (NBT + 1);
                        const unsigned epoch = (unsigned)(P * NB + t + 1);
                        const int xpar = (int)(epoch & 1u);
                        if (wave == 0) {
                            if (lane < S) {
                                const int d = lane >> 1;
                                const double val = d == 0 ? kb : (kb >= 0.0 ? cand_rows[par][ws][d > 0 ? d - 1 : 0] : 0.0);
                                const unsigned long long bits = (unsigned long long)__double_as_longlong(val);
                                const unsigned half = (lane & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
                                __hip_atomic_store((dn_u64*)(a.xch + ((size_t)xpar * KP + pk) * 40 + lane), ((unsigned long long)epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            int ok = 1;
                            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                            for (int idx = lane; idx < KP * S && ok; idx += 64) {
                                const int k = idx / S, gi = idx - k * S;
                                unsigned long long x;
                                while (((x = __hip_atomic_load((const dn_u64*)(a.xch + ((size_t)xpar * KP + k) * 40 + gi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) {
                                    __builtin_amdgcn_s_sleep(1);
                                    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                                        ok = 0;
                                        break;
                                    }
                                }
                                xh_s[k * 40 + gi] = (unsigned)x;
                            }
                            ok = __all(ok);
                            if (lane == 0) wait_s = ok;
                        }
                        dense_lds_barrier();
                        if (!wait_s) failed = 2;
                        auto val_of = [&](int k, int d) { return __longlong_as_double((long long)(((unsigned long long)xh_s[k * 40 + 2 * d + 1] << 32) | xh_s[k * 40 + 2 * d])); };
                        kb = -1.0;
                        int kw = 0;
                        for (int k = 0; k < KP; ++k) {
                            const double kk = val_of(k, 0);
                            if (kk > kb) kb = kk, kw = k;
                        }
#pragma unroll
                        for (int tt = 0; tt < NBT; ++tt) pw[tt] = val_of(kw, 1 + tt);
                    } else {
#pragma unroll
                        for (int tt = 0; tt < NBT; ++tt) pw[tt] = cand_rows[par][ws][tt];
                    }
                    const double mag = dense_cand_mag(kb);
                    if (failed) {
                    } else if (!(kb >= 0.0) || !(mag > 0.0) || !isfinite(mag)) {
                        failed = 1;
                    } else {
                        const int p = dense_cand_row(kb);
                        sstamp(P, t, 3);
                        const double inv_d = 1.0 / pw[s];
                        sstamp(P, t, 4);
                        // every row as an ordinary row, without a branch (rows beyond n hold zeros and keep them; the pivot row itself comes out as
                        // zeros and is set below by the one thread that owns it)
#pragma unroll
                        for (int r = 0; r < RPT; ++r) {
                            const double f = pr[r][s] * inv_d;
#pragma unroll
                            for (int c = 0; c < NBT; ++c)
                                if (c != s) pr[r][c] -= f * pw[c];
                            pr[r][s] = -f;
                        }
                        if (ft == 0) piv_row_s[t] = p;
                        const int pl = p - row0;
                        if (pl >= 0 && pl < RPT * T && pl % T == ft) {
                            const int rp = pl / T;
                            used_mine |= 1u << rp, piv_now |= 1u << rp;
#pragma unroll
                            for (int r = 0; r < RPT; ++r)
                                if (r == rp) {
#pragma unroll
                                    for (int c = 0; c < NBT; ++c) pr[r][c] = c == s ? inv_d : pw[c] * inv_d;
                                }
                        }
                        sstamp(P, t, 5);
                    }
                }
              }
              if (GS < NBT) {   // the next group's columns to positions 0 .. GS - 1
#pragma unroll
                  for (int r = 0; r < RPT; ++r) {
                      double head[GS];
#pragma unroll
                      for (int k = 0; k < GS; ++k) head[k] = pr[r][k];
#pragma unroll
                      for (int c = 0; c + GS < NBT; ++c) pr[r][c] = pr[r][c + GS];
#pragma unroll
                      for (int k = 0; k < GS; ++k) pr[r][NBT - GS + k] = head[k];
                  }
                  rot += GS;
              }
            }
            stamp(P, 0, 1);   // panel factorised
            dense_lds_barrier();   // (piv_row_s of the last step)
            const bool more = P + 1 < n_panels;
            const int k1 = k0 + NB, nbp1 = n - k1 < NB ? n - k1 : NB;
            // (up to 1 024 rows the wait for "update P - 1 done" comes FIRST -- in that regime it is long past -- and the first loads of the next panel's
            // columns are issued before the panel's own columns go out: one trip over the fabric less on this workgroup's turn, 23.7 -> 21.1 us per panel at
            // 289 rows; at 1 089 rows the mixed loads and stores cost what the trip saves, and beyond that the workers' update is the longer side and the
            // panel's columns go out before the wait)
            constexpr bool EARLY = RPT <= 2;
            double bf[NBT / 4];
            dn_v4d acc[4];
            auto load_base = [&](int r, dn_v4d (&dst)[4]) {
                int q = lane >> 4, jl = lane & 15, wv = wave;   // (opaque: offsets derived from them are loop-invariant, and hoisted they are RPT x 16 registers)
                asm volatile("" : "+v"(q), "+v"(jl), "+v"(wv));
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = row0 + r * T + wv * 64 + 16 * rt + 4 * v + q;
                        const bool ok = i < n && jl < nbp1;
                        dst[rt][v] = dense_bload(rs_cur, ok ? (i * ld + k1 + jl) * 8 : 0, 0);
                    }
                }
            };
            auto load_first = [&]() {
#pragma unroll
                for (int c = 0; c < NBT / 4; ++c) {
                    const int s_ = 4 * c + q;
                    const bool ok = s_ < nbp && jl < nbp1;
                    const double x = dense_bload(rs_cur, ok ? (piv_row_s[s_ < nbp ? s_ : 0] * ld + k1 + jl) * 8 : 0, 0);
                    bf[c] = ok ? x : 0.0;
                }
                load_base(0, acc);
            };
            if (EARLY) {
                // ---- every worker has finished the update of panel P - 1: buffer `cur` is complete
                if (P > 0 && !failed) {
                    if (wave == 0) {
                        int ok = 1;
                        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                        for (int w = lane; w < G && ok; w += 64)
                            while (__hip_atomic_load((const dn_u64*)(a.done + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)P) {
                                __builtin_amdgcn_s_sleep(1);
                                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                                    ok = 0;
                                    break;
                                }
                            }
                        ok = __all(ok);
                        if (lane == 0) wait_s = ok;
                    }
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (!wait_s) failed = 2;
                }
                if (more && !failed) load_first();
            }
            // ---- the panel's final columns E into `next` (nobody reads these columns of that buffer: the workers still applying panel P - 1 skip them)
            if (!failed) {
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    if (row0 + r * T < n) {
                        int ln = lane, wv = wave;   // (opaque copies: offsets derived from them are loop-invariant, and hoisted they are RPT x 16 registers)
                        asm volatile("" : "+v"(ln), "+v"(wv));
                        dense_lds_wave_sync();
#pragma unroll
                        for (int t = 0; t < NBT; ++t) slab[ln * SL + ((t + rot) & (NBT - 1))] = pr[r][t];   // (the rotation of the steps undone)
                        dense_lds_wave_sync();
#pragma unroll
                        for (int u = 0; u < NBT; ++u) {   // (16 lanes per row: whole 128-byte lines)
                            const int rl = (64 / NBT) * u + ln / NBT, cl = ln % NBT, i = row0 + r * T + wv * 64 + rl;
                            if (i < n && cl < nbp) dense_bstore(rs_next, (i * ld + k0 + cl) * 8, 0, slab[rl * SL + cl]);
                        }
                    }
                }
            }
            if (!EARLY) {
                // ---- every worker has finished the update of panel P - 1: buffer `cur` is complete
                if (P > 0 && !failed) {
                    if (wave == 0) {
                        int ok = 1;
                        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                        for (int w = lane; w < G && ok; w += 64)
                            while (__hip_atomic_load((const dn_u64*)(a.done + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)P) {
                                __builtin_amdgcn_s_sleep(1);
                                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                                    ok = 0;
                                    break;
                                }
                            }
                        ok = __all(ok);
                        if (lane == 0) wait_s = ok;
                    }
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (!wait_s) failed = 2;
                }
            }
            stamp(P, 0, 2);   // update P - 1 done everywhere
            if (MULTI && pk > 0) {   // another panel workgroup: "my rows of the panel's columns are out"; the first one publishes
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    if (failed) a.status[0] = failed;
                    __hip_atomic_store((dn_u64*)(a.pdone + pk), ((unsigned long long)(P + 1) << 1) | (failed ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (MULTI && KP > 1 && !failed) {   // ... once every other panel workgroup has said so
                    if (wave == 0) {
                        int ok = 1;
                        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                        if (lane >= 1 && lane < KP) {
                            unsigned long long x;
                            while (((x = __hip_atomic_load((const dn_u64*)(a.pdone + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != (unsigned long long)(P + 1)) {
                                __builtin_amdgcn_s_sleep(1);
                                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                                    ok = 0;
                                    break;
                                }
                            }
                            if (ok && (x & 1ull)) ok = 0;
                        }
                        ok = __all(ok);
                        if (lane == 0) wait_s = ok;
                    }
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (!wait_s) failed = 2;
                }
                if (!failed && ft < nbp) {   // (the workers read piv_row of panel P - 1 until they are done with it)
                    a.perm[k0 + ft] = piv_row_s[ft];
                    __hip_atomic_store((dn_u32*)(a.piv_row + ft), (unsigned)piv_row_s[ft], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    if (failed) a.status[0] = failed;
                    __hip_atomic_store((dn_u64*)a.ready, ((unsigned long long)(P + 1) << 1) | (failed ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            stamp(P, 0, 3);   // panel published
            if (failed || !more) return;
            // ---- the columns of panel P + 1 after the update of panel P, computed HERE (the workers skip them): new = base + E S[pivot rows][columns of
            // P + 1], rows in tiles of 16 (this wavefront's 64 rows of every row block, E through the slab into the A layout), the pivot rows' 16 x 16
            // entries once per wavefront; the next block's base rows are loaded while this block is computed
            {
                if (!EARLY) load_first();
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    asm volatile("" ::: "memory");   // (the loads of block r + 1 stay in block r)
                    if (row0 + r * T >= n) {   // (a row block beyond the matrix: zeros)
#pragma unroll
                        for (int t = 0; t < NBT; ++t) pr[r][t] = 0.0;
                        continue;
                    }
                    int q = lane >> 4, jl = lane & 15, ln = lane, wv = wave;   // (opaque copies, as above)
                    asm volatile("" : "+v"(q), "+v"(jl), "+v"(ln), "+v"(wv));
                    dense_lds_wave_sync();
#pragma unroll
                    for (int t = 0; t < NBT; ++t) slab[ln * SL + ((t + rot) & (NBT - 1))] = pr[r][t];   // (the rotation of the steps undone)
                    isp_s[wv * 64 + ln] = (unsigned char)((piv_now >> r) & 1u);
                    dense_lds_wave_sync();
                    dn_v4d cur[4];
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) cur[rt] = acc[rt];
                    if (r + 1 < RPT && row0 + (r + 1) * T < n) load_base(r + 1, acc);
                    asm volatile("" ::: "memory");
                    double af[4][NBT / 4];
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
                        for (int c = 0; c < NBT / 4; ++c) af[rt][c] = slab[(16 * rt + jl) * SL + 4 * c + q];
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int rl = 16 * rt + 4 * v + q, i = row0 + r * T + wv * 64 + rl;
                            if (!(i < n && jl < nbp1) || isp_s[wv * 64 + rl]) cur[rt][v] = 0.0;   // (base = 0: the panel's pivot rows; nothing beyond the matrix)
                        }
                    }
                    dense_lds_wave_sync();   // (the fragments are out of the slab before the new values go in)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
                        for (int c = 0; c < NBT / 4; ++c) cur[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[rt][c], bf[c], cur[rt], 0, 0, 0);
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (jl < NBT) slab[(16 * rt + 4 * v + q) * SL + jl] = cur[rt][v];
                    }
                    dense_lds_wave_sync();
#pragma unroll
                    for (int t = 0; t < NBT; ++t) pr[r][t] = (t < nbp1 && row0 + r * T + wv * 64 + ln < n) ? slab[ln * SL + t] : 0.0;
                }
            }
            stamp(P, 0, 4);   // the next panel's columns in registers
        }
        return;
    }

    // ======================================================================= the workers
    // this workgroup's block of the update: rows [i0, i0 + nr) x columns [j0, j0 + nc)
    const int i0 = (g / a.C) * a.RB, j0 = (g % a.C) * a.CB;
    const int nr = i0 < n && j0 < n ? (n - i0 < a.RB ? n - i0 : a.RB) : 0, nc = i0 < n && j0 < n ? (n - j0 < a.CB ? n - j0 : a.CB) : 0;
    const int rows_pad = (nr + 15) & ~15;
    double* En = buf;                                                    // [rows_pad][16] the panel's final columns, this block's rows
    unsigned char* is_piv = reinterpret_cast<unsigned char*>(En + (size_t)rows_pad * 16);    // [rows_pad] 1: the row is one of the panel's pivots
    for (int P = 0; P < n_panels; ++P) {
        const int k0 = P * NB, nbp = n - k0 < NB ? n - k0 : NB;
        const int skip_end = k0 + 2 * NB;   // the columns of panels P and P + 1 are the panel workgroup's
        const __amdgpu_buffer_rsrc_t rs_cur = (P & 1) ? rs1 : rs0, rs_next = (P & 1) ? rs0 : rs1;
        if (tid == 0) {
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            unsigned long long x;
            int ok = 1;
            while (((x = __hip_atomic_load((const dn_u64*)a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != (unsigned long long)(P + 1)) {
                __builtin_amdgcn_s_sleep(2);
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                    ok = 0;
                    break;
                }
            }
            if (!ok) a.status[0] = 2;
            wait_s = ok && !(x & 1ull) ? 1 : 0;
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!wait_s) return;
        if (g == 0) stamp(P, 1, 0);   // panel seen
        if (tid < nbp) piv_row_s[tid] = (int)__hip_atomic_load((const dn_u32*)(a.piv_row + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int idx = tid; idx < rows_pad * 16; idx += T) {   // (in flight with the pivots above; 16 lanes per row: whole 128-byte lines)
            const int rr = idx >> 4, t = idx & 15;
            const double e = dense_bload(rs_next, ((i0 + (rr < nr ? rr : 0)) * ld + k0 + (t < nbp ? t : 0)) * 8, 0);
            En[idx] = (rr < nr && t < nbp) ? e : 0.0;
        }
        __syncthreads();
        for (int r = tid; r < rows_pad; r += T) {
            unsigned char f = 0;
            for (int t = 0; t < nbp; ++t)
                if (r < nr && piv_row_s[t] == i0 + r) f = 1;
            is_piv[r] = f;
        }
        __syncthreads();
        if (g == 0) stamp(P, 1, 1);   // the panel's rows in LDS
        {
            constexpr int TU = 4;   // row tiles whose loads are in flight together
            const int n_ct = (nc + 15) >> 4, n_rb = rows_pad >> 4;
            int piv_off[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) piv_off[c] = 4 * c + q < nbp ? piv_row_s[4 * c + q] * ld * 8 : -1;
            for (int ct = wave; ct < n_ct; ct += W) {   // a column tile per wavefront: its pivot-row fragment once, then the row tiles
                const int j = j0 + ct * 16 + jl, j8 = j * 8;   // (j < ld: inside the row; columns n .. ld - 1 are computed and not stored)
                const bool col_ok = j < n && (j < k0 || j >= skip_end);
                if (!__any(col_ok)) continue;   // (a tile entirely inside the two panels)
                double bf[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double x = dense_bload(rs_cur, piv_off[c] >= 0 ? piv_off[c] + j8 : 0, 0);
                    bf[c] = piv_off[c] >= 0 ? x : 0.0;
                }
                for (int rb0 = 0; rb0 < n_rb; rb0 += TU) {
                    dn_v4d acc[TU];
                    double af[TU][4];
#pragma unroll
                    for (int u = 0; u < TU; ++u) {
                        const int rb = rb0 + u;
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int r = rb * 16 + 4 * v + q;
                            const bool base = rb < n_rb && r < nr && !is_piv[r < rows_pad ? r : 0];
                            const double x = dense_bload(rs_cur, base ? (i0 + r) * ld * 8 + j8 : 0, 0);
                            acc[u][v] = base ? x : 0.0;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) af[u][c] = rb < n_rb ? En[(rb * 16 + jl) * 16 + 4 * c + q] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < TU; ++u) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][c], bf[c], acc[u], 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < TU; ++u) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int r = (rb0 + u) * 16 + 4 * v + q;
                            if (col_ok && r < nr) dense_bstore(rs_next, (i0 + r) * ld * 8 + j8, 0, acc[u][v]);
                        }
                    }
                }
            }
        }
        if (g == 0) stamp(P, 1, 2);   // update issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (g == 0) stamp(P, 1, 3);   // ... and drained
        if (tid == 0) __hip_atomic_store((dn_u64*)(a.done + g), (unsigned long long)(P + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// X[k][perm[j]] = S[perm[k]][j]
static __global__ void k_dense_unpermute(int64_t n, int64_t ld, const double* S, const int32_t* perm, double* X) {
    const int64_t k = blockIdx.x;
    const double* src = S + (int64_t)perm[k] * ld;
    double* dst = X + k * n;
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) dst[perm[j]] = src[j];
}

// out[0] = bits of max |delta_ic - sum_t A[i][t] X[t][c]| (A: the same rows k_dense_fill wrote)
static __global__ void k_dense_check(int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals, const uint8_t* bnd, int use_bnd, const double* X,
                                     unsigned long long* out) {
    const int64_t i = blockIdx.x;
    const bool unit = use_bnd && bnd[i];
    double worst = 0.0;
    for (int64_t c = threadIdx.x; c < n; c += blockDim.x) {
        double s = 0.0;
        if (unit) s = X[i * n + c];
        else
            for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += vals[k] * X[(int64_t)colidx[k] * n + c];
        const double e = fabs((c == i ? 1.0 : 0.0) - s);
        worst = e > worst || !(e == e) ? (e == e ? e : 1e300) : worst;
    }
    for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(worst));
}

// right-hand sides handed over in the reference numbering (column-major n x nc, pinned host or device memory) -> internal order
static __global__ void k_dense_stage(int64_t n, int nc, const int32_t* i2e, const double* b_ext, double* b_int, unsigned int* count) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && count) *count = 0u;   // (the arrival counter of the launch that signals the host: ordered by the stream)
    if (t >= n * nc) return;
    const int64_t c = t / n, i = t - c * n;
    b_int[t] = b_ext[c * n + i2e[i]];
}

// y (+)= X v for nc columns (column-major n x nc), one wavefront per row, the columns in tiles of NC.  The row is walked four 64-entry chunks at a time,
// all loads of a round issued before the first multiply (a plain loop waited for every chunk's trip to L2 in turn: 17 trips, 7 us at 1 089 rows).
template <int NC> static __global__ __launch_bounds__(256) void k_dense_gemv(int64_t n, int nc, const double* X, const double* v, double* y, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* row = X + i * n;
    constexpr int U = NC <= 2 ? 4 : NC <= 4 ? 2 : 1;   // chunks per round
    for (int c0 = 0; c0 < nc; c0 += NC) {
        double acc[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[q] = 0.0;
        for (int64_t j0 = lane; j0 < n; j0 += 64 * U) {
            double x[U], w[U][NC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = j0 + 64 * u;
                x[u] = j < n ? row[j] : 0.0;
#pragma unroll
                for (int q = 0; q < NC; ++q) w[u][q] = (j < n && c0 + q < nc) ? v[(int64_t)(c0 + q) * n + j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int q = 0; q < NC; ++q) acc[q] += x[u] * w[u][q];
            }
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const double s = wave_sum(acc[q]);
            if (lane == 0 && c0 + q < nc) y[(int64_t)(c0 + q) * n + i] = accumulate ? y[(int64_t)(c0 + q) * n + i] + s : s;
        }
    }
}

// x = X b for ONE column of a system of up to 512 rows, b read straight from pinned HOST memory (internal order: the host permutes it while copying it there) once
// per workgroup into LDS -- 73 x 2.3 KB over PCIe in whole lines at 289 rows: a PCIe round trip instead of the launch seam of k_dense_stage.
static __global__ __launch_bounds__(256) void k_dense_gemv_hostb(int n, const double* X, const double* b_host, double* y, unsigned int* count) {
    __shared__ double b_s[512];
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = 0u;   // (the arrival counter of k_dense_out behind this launch)
    for (int j = threadIdx.x; j < n; j += blockDim.x) b_s[j] = b_host[j];
    __syncthreads();
    const int lane = threadIdx.x & 63, i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* row = X + (int64_t)i * n;
    double s = 0.0;
    for (int j0 = lane; j0 < n; j0 += 256) {
        double x[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + 64 * k;
            x[k] = j < n ? row[j] : 0.0, w[k] = j < n ? b_s[j] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s += x[k] * w[k];
    }
    s = wave_sum(s);
    if (lane == 0) y[i] = s;
}

// ONE column, product and hand-over in ONE launch: x = X b with b in the internal order, the result straight into pinned host memory in the reference numbering,
// the last workgroup to arrive signals the host.  HOSTB: b sits in pinned HOST memory (the host has permuted it while copying it there) and every workgroup reads it
// once into LDS -- n x 8 bytes per workgroup over PCIe in whole lines, affordable while n and the workgroup count are small (<= 512 rows: 73 x 2.3 KB at 289); the whole
// solve is then this one launch.  Otherwise b is a device vector (k_dense_stage in front: two launches instead of stage -> product -> out).
// (The first form of this kernel gathered b_ext[i2e[j]] over PCIe -- 21 000 lone 8-byte reads at 289 rows, 35 us -- and was left off.)
template <bool HOSTB> static __global__ __launch_bounds__(256) void k_dense_gemv_direct(int n, const double* X, const int32_t* i2e, const double* b_int, double* x_ext,
                                                                                    long long* done, unsigned int* count) {
    __shared__ double b_s[HOSTB ? 512 : 1];
    if (HOSTB) {
        for (int j = threadIdx.x; j < n; j += blockDim.x) b_s[j] = b_int[j];
        __syncthreads();
    }
    const double* bv = HOSTB ? b_s : b_int;
    const int lane = threadIdx.x & 63, i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i < n) {
        const double* row = X + (int64_t)i * n;
        double s = 0.0;
        for (int j0 = lane; j0 < n; j0 += 256) {   // (four chunks' loads in flight: see k_dense_gemv)
            double x[4], w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 64 * k;
                x[k] = j < n ? row[j] : 0.0, w[k] = j < n ? bv[j] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) s += x[k] * w[k];
        }
        s = wave_sum(s);
        // (write-through to the host, drained below: NO system-scope fence here -- __threadfence_system in every wavefront writes back and invalidates the
        // L2, X with it: 33 instead of 21 us at 289 rows, 159 instead of 57 at 4 225)
        if (lane == 0) __hip_atomic_store((dn_u64*)(x_ext + i2e[i]), (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned int last;
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add((dn_u32*)count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __hip_atomic_store((dn_u32*)count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ready for the next launch)
        __hip_atomic_store((dn_u64*)done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (everybody's stores were drained before their count)
    }
}

// The implicit Euler stepper with K^-1 = X in hand (fem_linear_parabolic_solver.h:56-68: rhs = M u_i / dt + f_{i+1}, Dirichlet rows = g, u_{i+1} = K^-1 rhs) folded into
// ONE product per step:  u_{i+1} = B u_i + c_{i+1},   B = X D M / dt (D: 1 on the rows that are not Dirichlet rows),   c_{i+1} = X (D f_{i+1} + (1 - D) g_{i+1}).
// k_dense_xm builds B (M is symmetric: column j of M is its row j), k_dense_step_cols the right-hand sides of all steps (C = X F with k_dense_gemv), k_dense_step one step.
static __global__ void k_dense_xm(int64_t n, const double* X, const int32_t* rowptr, const int32_t* colidx, const double* mass, const uint8_t* bnd, int use_bnd, double inv_dt,
                                  double* B) {
    const int64_t i = blockIdx.y;
    const double* xr = X + i * n;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int32_t k = rowptr[j]; k < rowptr[j + 1]; ++k) {
            const int32_t r = colidx[k];
            if (!(use_bnd && bnd[r])) s += xr[r] * mass[k];
        }
        B[i * n + j] = s * inv_dt;
    }
}
// F[., t] = f(., t + 1) with g(., t + 1) on the Dirichlet rows (g in the reference numbering), t = 0 .. nt - 1, internal order
static __global__ void k_dense_step_cols(int64_t n, int64_t nt, const double* force, const double* g_ext, const int32_t* i2e, const uint8_t* bnd, double* F) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nt) return;
    const int64_t c = t / n, i = t - c * n;
    F[t] = (g_ext && bnd[i]) ? g_ext[(c + 1) * n + i2e[i]] : force[(c + 1) * n + i];
}
// u_next = B u + c; also into the solution column (reference numbering); one wavefront per row
static __global__ __launch_bounds__(256) void k_dense_step(int64_t n, const double* B, const double* u, const double* cvec, const int32_t* i2e, double* u_next, double* sol_ext) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* row = B + i * n;
    double s = 0.0;
    for (int64_t j0 = lane; j0 < n; j0 += 256) {   // (four chunks' loads in flight: see k_dense_gemv)
        double x[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t j = j0 + 64 * k;
            x[k] = j < n ? row[j] : 0.0, w[k] = j < n ? u[j] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s += x[k] * w[k];
    }
    s = wave_sum(s);
    if (lane == 0) {
        const double v = s + cvec[i];
        u_next[i] = v, sol_ext[i2e[i]] = v;
    }
}

// r = b - A x (the rows of k_dense_fill), nc columns
static __global__ void k_dense_residual(int64_t n, int nc, const int32_t* rowptr, const int32_t* colidx, const double* vals, const uint8_t* bnd, int use_bnd,
                                        const double* b, const double* x, double* r) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nc) return;
    const int64_t c = t / n, i = t - c * n;
    double s = 0.0;
    if (use_bnd && bnd[i]) s = x[t];
    else
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += vals[k] * x[c * n + colidx[k]];
    r[t] = b[t] - s;
}

// the result in the reference numbering into x_ext (pinned host memory of a direct solve, or a device buffer) + the completion word
static __global__ void k_dense_out(int64_t n, int nc, const int32_t* e2i, const double* x_int, double* x_ext, long long* done, unsigned int* count) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * nc) {
        const int64_t c = t / n, e = t - c * n;
        x_ext[t] = x_int[c * n + e2i[e]];
    }
    if (!done) return;
    // the last workgroup to finish signals the host: its own stores and everybody else's are out (system scope: the word lives in host memory)
    __threadfence_system();
    __syncthreads();
    __shared__ unsigned int last;
    if (threadIdx.x == 0) last = atomicAdd(count, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence_system();
        __atomic_store_n(reinterpret_cast<volatile long long*>(done), 1ll, __ATOMIC_RELEASE);
    }
}

}   // namespace fdapde_hip
#endif
