// kernels_krylov.h -- Jacobi scaling, CG / BiCGStab vector kernels, interface exchange, numbering changes; see kernels.h
#ifndef FDAPDE_KERNELS_KRYLOV_H
#define FDAPDE_KERNELS_KRYLOV_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "internal.h"
#include "kernels_reduce.h"

namespace fdapde_hip {

// ---------------------------------------------------------------------------------------------------------------
// solve set-up kernels
// ---------------------------------------------------------------------------------------------------------------
// scale[i] = 0 on Dirichlet rows, 1/sqrt(|A_ii|) elsewhere; flag[0] |= 1 if some interior diagonal is <= 0.
// A zero (or non-finite) interior diagonal -- pure advection on a symmetric patch, an arbitrary matrix handed to
// fdapde_lin_compute -- leaves its row unscaled (scale 1) instead of dividing by zero: the flag then selects the full
// pattern, whose stored diagonal is streamed as it is, and BiCGStab.
__device__ __forceinline__ double jacobi_scale_of(double d) {
    const double a = fabs(d);
    return (a > 0.0 && a < 1.7976931348623157e308) ? 1.0 / sqrt(a) : 1.0;
}
// Single GPU: a diagonal that is tiny against its row (|d| <= 1e-8 max_j |a_ij|: rounding-level diagonals of pure advection,
// where int psi_i b.grad psi_i vanishes over an interior patch) is treated like a zero one, and the row is scaled by its largest
// entry instead, so that the scaled system stays O(1).
static __global__ __launch_bounds__(256) void k_jacobi_scale(int64_t n, const int32_t* rowptr, const int32_t* diag, const double* vals,
                                                      const uint8_t* bnd, int use_bnd, double* scale, int32_t* flag) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;   // 16 lanes per row: the row's entries are read coalesced
    const int l = threadIdx.x & 15;
    const bool live = i < n;
    double rmax = 0.0;
    if (live)
        for (int32_t k = rowptr[i] + l; k < rowptr[i + 1]; k += 16) rmax = fmax(rmax, fabs(vals[k]));
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, o, 16));
    if (!live || l != 0) return;
    if (use_bnd && bnd[i]) {
        scale[i] = 0.0;
        return;
    }
    const double d = vals[diag[i]];
    const bool tiny = !(fabs(d) > 1e-8 * rmax);
    if (!(d > 0.0) || tiny) atomicOr(flag, 1);
    scale[i] = jacobi_scale_of(tiny ? rmax : d);
}
// the same from (diagonal, row maximum) pairs the assembly left behind (AsmArgs::row_stat): 16 bytes per row instead of the whole matrix
__device__ __forceinline__ void jacobi_scale_stats_row(int64_t i, const double* stat, const uint8_t* bnd, int use_bnd, double* scale, int32_t* flag) {
    if (use_bnd && bnd[i]) {
        scale[i] = 0.0;
        return;
    }
    const double d = stat[2 * i], rmax = stat[2 * i + 1];
    const bool tiny = !(fabs(d) > 1e-8 * rmax);
    if (!(d > 0.0) || tiny) atomicOr(flag, 1);
    scale[i] = jacobi_scale_of(tiny ? rmax : d);
}
static __global__ __launch_bounds__(256) void k_jacobi_scale_stats(int64_t n, const double* stat, const uint8_t* bnd, int use_bnd, double* scale,
                                                                   int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) jacobi_scale_stats_row(i, stat, bnd, use_bnd, scale, flag);
}
// At = diag(scale) A diag(scale): symmetric Jacobi scaling == Jacobi preconditioning folded into the matrix stream.
// Rows and columns of Dirichlet DOFs vanish (scale = 0), which restricts the Krylov iteration to the interior block.
static __global__ __launch_bounds__(256) void k_scale_matrix(int64_t n, const int32_t* rowptr, const int32_t* colidx,
                                                      const double* vals, const double* scale, double* out) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;   // 16 lanes per row
    const int l = threadIdx.x & 15;
    if (row >= n) return;
    const double si = scale[row];
    for (int k = rowptr[row] + l; k < rowptr[row + 1]; k += 16) out[k] = si * vals[k] * scale[colidx[k]];
}
// the same into the compact solver matrix: entries with map[k] < 0 are dropped (the diagonal, which scales to exactly 1,
// and every entry in a row or column of a Dirichlet DOF, which scales to exactly 0)
static __global__ __launch_bounds__(256) void k_scale_matrix_compact(int64_t n, const int32_t* rowptr, const int32_t* colidx,
                                                              const double* vals, const double* scale, const int32_t* map,
                                                              double* out) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (row >= n) return;
    const double si = scale[row];
    for (int k = rowptr[row] + l; k < rowptr[row + 1]; k += 16) {
        const int32_t m = map[k];
        if (m >= 0) out[m] = si * vals[k] * scale[colidx[k]];
    }
}
// gt = g on Dirichlet DOFs, 0 elsewhere (or all zero without Dirichlet data)
static __global__ void k_lift(int64_t n, const uint8_t* bnd, const double* g, int use_bnd, double* gt, double* zero_y = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        gt[i] = (use_bnd && bnd[i]) ? g[i] : 0.0;
        if (zero_y) zero_y[i] = 0.0;   // (A g~ of a zero lift)
    }
}
// bt = scale * (f - A gt)  (y holds A gt): right-hand side of the scaled interior system.
// Cold start (u0 == nullptr): x = 0, r = bt.  Warm start: x = (u0 - gt) / scale on interior DOFs, r = bt - ax where ax holds
// At x (one extra SpMV by the caller between the two launches: first launch with ax == nullptr only fills x).
// partial[2 b] = sum r^2, partial[2 b + 1] = sum bt^2 (the stopping rule is relative to ||bt||, not to the warm residual).
// (the body of workgroup `bid` of `nblocks`, 256 threads: k_small_front runs the workgroups of a small system one after the other -- same sums)
__device__ __forceinline__ void krylov_init_block(int bid, int nblocks, int64_t n, const double* f, const double* y, const double* scale,
                                                  double* x, double* r, double* p, double* r0, double* partial,
                                                  const uint8_t* owned, const double* u0, const double* gt,
                                                  const double* ax, int fill_x_only, double* red) {
    double acc = 0, accb = 0;
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < n; i += (int64_t)nblocks * blockDim.x) {
        if (fill_x_only) {
            x[i] = scale[i] > 0.0 ? (u0[i] - gt[i]) / scale[i] : 0.0;
            continue;
        }
        const double bt = scale[i] * (f[i] - y[i]);
        const double ri = ax ? bt - ax[i] : bt;
        if (!u0) x[i] = 0.0;
        r[i] = ri, p[i] = ri;
        if (r0) r0[i] = ri;
        if (!owned || owned[i]) acc += ri * ri, accb += bt * bt;
    }
    if (fill_x_only) return;
    const double s = block_sum(acc, red);
    const double sb = block_sum(accb, red);
    if (threadIdx.x == 0) partial[2 * bid] = s, partial[2 * bid + 1] = sb;
}
static __global__ __launch_bounds__(256) void k_krylov_init(int64_t n, const double* f, const double* y, const double* scale,
                                                      double* x, double* r, double* p, double* r0, double* partial,
                                                      const uint8_t* owned, const double* u0, const double* gt,
                                                      const double* ax, int fill_x_only) {
    __shared__ double red[8];
    krylov_init_block((int)blockIdx.x, (int)gridDim.x, n, f, y, scale, x, r, p, r0, partial, owned, u0, gt, ax, fill_x_only, red);
}
// scalars layout (device doubles): [0] reference norm^2 (||bt||^2), [1] rr_even, [2] rr_odd, [3] last rr, [4..] method specific
// ctl layout (device int32): [0] stop flag, [1] iterations done, [2] breakdown flag
// seed (optional): the fused-update CG reads its explicit r.r from n_seed per-workgroup partials; they are seeded with
// (rr, 0, 0, ...) so that its first launch needs no special case (and the launch sequence can be replayed as a graph)
__device__ __forceinline__ void krylov_init_fin_body(const double* partial, int np, double* sc, int32_t* ctl, double tol2, double* seed, int n_seed,
                                                     double* red) {
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) a += partial[2 * i], b += partial[2 * i + 1];
    const double rr = block_sum(a, red);
    const double bb = block_sum(b, red);
    __syncthreads();   // all reads of `partial` are done: seed may alias it
    for (int i = threadIdx.x; i < n_seed; i += blockDim.x) seed[i] = i == 0 ? rr : 0.0;
    if (threadIdx.x == 0) {
        sc[0] = bb, sc[1] = rr, sc[2] = rr, sc[3] = rr;
        sc[4] = 1.0, sc[5] = 1.0, sc[6] = 1.0;   // bicgstab: rho, alpha, omega
        sc[9] = rr;                               // bicgstab: (r0, r0) of the first iteration
        sc[16] = 0.0, sc[17] = 0.0, sc[18] = 0.0; // fused-update CG: no update of x pending
        ctl[0] = rr <= tol2 * bb ? 1 : 0, ctl[1] = 0, ctl[2] = 0;
    }
}
static __global__ __launch_bounds__(256) void k_krylov_init_fin(const double* partial, int np, double* sc, int32_t* ctl, double tol2,
                                                         double* seed, int n_seed) {
    __shared__ double red[8];
    krylov_init_fin_body(partial, np, sc, ctl, tol2, seed, n_seed, red);
}
// Several right-hand sides of fdapde_lin_solve at once (one persistent launch solves them side by side, kernels_persist.h n_cols): what
// k_gather_f64 + k_krylov_init / _fin do for one column, for column blockIdx.y -- the same loops and the same reduction order, hence the
// same bits: r = bt = scale * b[ext order -> internal], partial sums of bt^2 per workgroup, then sc[4 col] = ||bt||^2
static __global__ __launch_bounds__(256) void k_cols_init(int64_t n, const double* b_ext, const int32_t* i2e, const double* scale, double* r, double* partial) {
    __shared__ double red[8];
    const size_t col = blockIdx.y;
    b_ext += col * (size_t)n, r += col * (size_t)n, partial += col * 2 * (size_t)gridDim.x;
    double acc = 0, accb = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double bt = scale[i] * (b_ext[i2e[i]] - 0.0);
        r[i] = bt;
        acc += bt * bt, accb += bt * bt;
    }
    const double s = block_sum(acc, red);
    const double sb = block_sum(accb, red);
    if (threadIdx.x == 0) partial[2 * blockIdx.x] = s, partial[2 * blockIdx.x + 1] = sb;
}
static __global__ __launch_bounds__(256) void k_cols_init_fin(const double* partial, int np, double* sc, int32_t* ctl) {
    __shared__ double red[8];
    const size_t col = blockIdx.x;
    partial += col * 2 * (size_t)np;
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) a += partial[2 * i], b += partial[2 * i + 1];
    const double rr = block_sum(a, red);
    const double bb = block_sum(b, red);
    if (threadIdx.x == 0) {
        sc[4 * col] = bb, sc[4 * col + 1] = rr, sc[4 * col + 2] = rr, sc[4 * col + 3] = rr;
        ctl[4 * col] = 0, ctl[4 * col + 1] = 0, ctl[4 * col + 2] = 0, ctl[4 * col + 3] = 0;
    }
}
// u = scale * x (k_unscale without lift) of column blockIdx.y, written in the reference's DOF order (k_scatter_f64)
static __global__ __launch_bounds__(256) void k_cols_finish(int64_t n, const double* x, const double* scale, const int32_t* i2e, double* out_ext) {
    const size_t col = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out_ext[col * (size_t)n + i2e[i]] = scale[i] * x[col * (size_t)n + i] + 0.0;
}
// out[0], out[1] = sums of the stride-2 partial pairs, fixed order; single workgroup
static __global__ __launch_bounds__(256) void k_reduce_partials2(const double* part, int np, double* out) {
    __shared__ double red[8];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) a += part[2 * i], b += part[2 * i + 1];
    const double sa = block_sum(a, red);
    const double sb = block_sum(b, red);
    if (threadIdx.x == 0) out[0] = sa, out[1] = sb;
}
// K = M / dt + A  (FEMLinearParabolicSolver::solve, fem_linear_parabolic_solver.h:49), same pattern, elementwise
static __global__ void k_matrix_combine(int64_t nnz, const double* mass, const double* stiff, double inv_dt, double* out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) out[k] = mass[k] * inv_dt + stiff[k];
}
// rhs = mu * inv_dt + f   (mu = M u_i)
static __global__ void k_parabolic_rhs(int64_t n, const double* mu, double inv_dt, const double* f, double* rhs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rhs[i] = mu[i] * inv_dt + f[i];
}

// ---------------------------------------------------------------------------------------------------------------
// CG (on the symmetrically scaled system, i.e. Jacobi-PCG on the original one)
//   k_spmv            : y = At p, partials of p.y
//   k_cg_update_xr    : alpha = rr / p.y ; x += alpha p ; r -= alpha y ; partials of r.r
//   k_cg_update_p     : x += alpha p ; beta = rr_new / rr ; p = r + beta p ; bookkeeping + stopping test
//   (x is updated where p is streamed anyway: 3 + 5 vector passes per iteration instead of 6 + 3)
// ---------------------------------------------------------------------------------------------------------------
// Both update kernels are single-shot: workgroup b owns kCgV * 256 consecutive double2 elements, every lane issues all of
// its 16-byte loads FIRST, and only then re-reduces the producer's partials (an L2 round trip plus two barriers) -- the
// reduction hides under the loads instead of delaying them (measured per-kernel saving ~2 us of 17 / 9 us).
constexpr int kCgV = 4;
// owned (multi-GPU): 1 for DOFs this rank counts in global dot products, nullptr = all (single GPU)
static __global__ __launch_bounds__(256) void k_cg_update_xr(int64_t n, const double* y, double* r, const double* part_in, int np_in,
                                                       double* part_out, double* sc, int parity, int32_t* ctl,
                                                       const uint8_t* owned) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kCgV) + threadIdx.x;
    const double2* y2 = reinterpret_cast<const double2*>(y);
    double2* r2 = reinterpret_cast<double2*>(r);
    double2 yv[kCgV], rv[kCgV];
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        yv[k] = y2[ic], rv[k] = r2[ic];
    }
    double v = 0;
    for (int i = threadIdx.x; i < np_in; i += blockDim.x) v += part_in[2 * i];
    const double pAp = block_sum(v, red);
    const double rr = sc[1 + parity];
    const double alpha = pAp > 0.0 ? rr / pAp : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[14] = alpha;                  // x += alpha p is done by k_cg_update_p, which streams p anyway
        if (!(pAp > 0.0)) ctl[2] = 1;    // not SPD / breakdown
    }
    double acc = 0;
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) {
            rv[k].x -= alpha * yv[k].x, rv[k].y -= alpha * yv[k].y;
            r2[i] = rv[k];
            if (owned)
                acc += (owned[2 * i] ? rv[k].x * rv[k].x : 0.0) + (owned[2 * i + 1] ? rv[k].y * rv[k].y : 0.0);
            else
                acc += rv[k].x * rv[k].x + rv[k].y * rv[k].y;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double ri = r[i] - alpha * y[i];
        r[i] = ri;
        if (!owned || owned[i]) acc += ri * ri;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part_out[blockIdx.x] = s;
}
static __global__ __launch_bounds__(256) void k_cg_update_p(int64_t n, const double* r, double* p, double* x, const double* part_in,
                                                      int np_in, double* sc, int parity, double tol2, int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kCgV) + threadIdx.x;
    const double2* r2 = reinterpret_cast<const double2*>(r);
    double2* p2 = reinterpret_cast<double2*>(p);
    double2* x2 = reinterpret_cast<double2*>(x);
    double2 rv[kCgV], pv[kCgV], xv[kCgV];
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        rv[k] = r2[ic], pv[k] = p2[ic], xv[k] = x2[ic];
    }
    const double rr_new = sum_partials(part_in, np_in, red);
    const double rr = sc[1 + parity], alpha = sc[14];
    const double beta = rr > 0.0 ? rr_new / rr : 0.0;
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) {
            xv[k].x += alpha * pv[k].x, xv[k].y += alpha * pv[k].y;
            pv[k].x = rv[k].x + beta * pv[k].x, pv[k].y = rv[k].y + beta * pv[k].y;
            x2[i] = xv[k], p2[i] = pv[k];
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        x[n - 1] += alpha * p[n - 1];
        p[n - 1] = r[n - 1] + beta * p[n - 1];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        sc[1 + (parity ^ 1)] = rr_new, sc[3] = rr_new;
        ctl[1] += 1;
        // the stop flag is read by this launch's other workgroups only at their start; writing it here is seen by the
        // next kernel (kernel boundary = device-scope release/acquire)
        if (rr_new <= tol2 * sc[0] || ctl[2]) ctl[0] = 1;
    }
}

// The producer's stride-2 partials are requested BEFORE a kernel's vector loads and summed after them: loads return in order, so
// partials issued after the vectors would arrive only when the whole vector phase has (k_cgf_update: 32.8 -> 31.3 ms per C3 solve).
#define PRELOAD_PAIRS(name, ptr, np)                                                                             \
    double2 name[8];                                                                                             \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                                           \
        const int i_ = threadIdx.x + j_ * 256;                                                                   \
        name[j_] = i_ < (np) ? *reinterpret_cast<const double2*>((ptr) + 2 * i_) : make_double2(0.0, 0.0);       \
    }
#define SUM_PAIRS(name, ptr, np, a, b)                                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) a += name[j_].x, b += name[j_].y;                           \
    for (int i_ = threadIdx.x + 8 * 256; i_ < (np); i_ += blockDim.x) a += (ptr)[2 * i_], b += (ptr)[2 * i_ + 1];

// Single-reduction CG (Chronopoulos & Gear): the SpMV acts on r, both dot products of an iteration -- gamma = r.r and
// delta = r.(At r) -- are fused into it, and ONE kernel then updates all vectors:
//     beta = gamma / gamma_old ; alpha = gamma / (delta - beta gamma / alpha_old)
//     p = r + beta p ; s = w + beta s (= At p) ; x += alpha p ; r -= alpha s
// Two launches and (multi-GPU) one all-reduce per iteration instead of three and two.  Same Krylov iterates as CG in
// exact arithmetic.  part_in: stride-2 pairs (delta, gamma); scalars: sc[10 + parity] gamma_old, sc[12 + parity] alpha_old.
// Every workgroup takes the stop decision from the same reduced numbers, so no workgroup updates past convergence.
// if_slot / hb (multi-GPU, else nullptr): rows with if_slot[row] >= 0 take w from the all-reduced interface buffer hb, which
// saves the separate unpack launch (w itself is not read again)
static __global__ __launch_bounds__(256) void k_cgsr_update(int64_t n, double* r, const double* w, double* p, double* s, double* x,
                                                      const double* part_in, int np_in, double* sc, int parity, int first,
                                                      double tol2, int32_t* ctl, const int32_t* if_slot, const double* hb,
                                                      int64_t band2) {
    // band2 > 0: XCD-aware mapping (workgroup b serves the elements of SpMV row band b % 8, see k_cgf_update); w, p, s, x, which
    // the next SpMV does not read, then move with the nontemporal hint and leave the L2 to r
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;
    const int64_t lo = band2 > 0 ? (int64_t)(blockIdx.x & 7) * band2 : 0;
    const int64_t n2 = band2 > 0 ? min(n >> 1, lo + band2) : (n >> 1);
    const int64_t i0 = lo + (int64_t)(band2 > 0 ? (blockIdx.x >> 3) : blockIdx.x) * (256 * kCgV) + threadIdx.x;
    typedef double v2f64s_t __attribute__((ext_vector_type(2)));
    auto ld = [&](const double2* ptr, int64_t i) -> double2 {
        if (band2 > 0) {
            const v2f64s_t t = __builtin_nontemporal_load(reinterpret_cast<const v2f64s_t*>(ptr + i));
            return make_double2(t.x, t.y);
        }
        return ptr[i];
    };
    auto st = [&](double2* ptr, int64_t i, double2 v) {
        if (band2 > 0)
            __builtin_nontemporal_store(v2f64s_t{v.x, v.y}, reinterpret_cast<v2f64s_t*>(ptr + i));
        else
            ptr[i] = v;
    };
    double2* r2 = reinterpret_cast<double2*>(r);
    const double2* w2 = reinterpret_cast<const double2*>(w);
    double2* p2 = reinterpret_cast<double2*>(p);
    double2* s2 = reinterpret_cast<double2*>(s);
    double2* x2 = reinterpret_cast<double2*>(x);
    double2 rv[kCgV], wv[kCgV], pv[kCgV], sv[kCgV], xv[kCgV];
    PRELOAD_PAIRS(pp, part_in, np_in)
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        rv[k] = r2[ic], wv[k] = ld(w2, ic), pv[k] = ld(p2, ic), sv[k] = ld(s2, ic), xv[k] = ld(x2, ic);
        if (if_slot) {
            const int s0 = if_slot[2 * ic], s1 = if_slot[2 * ic + 1];
            if (s0 >= 0) wv[k].x = hb[s0];
            if (s1 >= 0) wv[k].y = hb[s1];
        }
    }
    double a = 0, b = 0;
    SUM_PAIRS(pp, part_in, np_in, a, b)
    const double delta = block_sum(a, red);
    const double gamma = block_sum(b, red);
    const bool last = blockIdx.x == gridDim.x - 1 && threadIdx.x == 0;
    if (gamma <= tol2 * sc[0]) {   // converged at the residual the SpMV has just measured: x, r stay as they are
        if (last) sc[3] = gamma, ctl[0] = 1;
        return;
    }
    const double gamma_old = sc[10 + parity], alpha_old = sc[12 + parity];
    const double beta = first ? 0.0 : gamma / gamma_old;
    const double denom = first ? delta : delta - beta * gamma / alpha_old;
    const double alpha = denom > 0.0 ? gamma / denom : 0.0;
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) {
            pv[k].x = rv[k].x + beta * pv[k].x, pv[k].y = rv[k].y + beta * pv[k].y;
            sv[k].x = wv[k].x + beta * sv[k].x, sv[k].y = wv[k].y + beta * sv[k].y;
            xv[k].x += alpha * pv[k].x, xv[k].y += alpha * pv[k].y;
            rv[k].x -= alpha * sv[k].x, rv[k].y -= alpha * sv[k].y;
            st(p2, i, pv[k]), st(s2, i, sv[k]), st(x2, i, xv[k]), r2[i] = rv[k];
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double wi = (if_slot && if_slot[i] >= 0) ? hb[if_slot[i]] : w[i];
        p[i] = r[i] + beta * p[i], s[i] = wi + beta * s[i];
        x[i] += alpha * p[i], r[i] -= alpha * s[i];
    }
    if (last) {
        sc[10 + (parity ^ 1)] = gamma, sc[12 + (parity ^ 1)] = alpha, sc[3] = gamma;
        ctl[1] += 1;
        if (!(denom > 0.0)) ctl[2] = 1, ctl[0] = 1;   // not SPD / breakdown
    }
}

// Fused-update CG (single GPU): the SpMV y = At p carries p.y and y.y; ONE kernel then does
//     alpha = rr / p.y ; x += alpha p ; r -= alpha y ; beta = (alpha^2 y.y - rr) / rr ; p = r + beta p
// rr is the EXPLICIT r.r (partials written by the previous launch of this kernel); alpha^2 y.y - rr equals r_new.r_new in
// exact arithmetic (r.y = p.y by A-conjugacy) and is used for beta only, so that p needs no second pass after a global
// reduction: 7 vector passes and 2 launches per iteration instead of 8 and 3.  The stop test is taken at the start of the
// next launch (or by k_cgf_fin at a host poll) from the explicit r.r, uniformly by every workgroup.
// band2 > 0 (XCD-aware form): workgroup b serves the double2 elements of SpMV row band b % 8 ([band * band2, (band + 1) * band2)),
// so that the p it writes sits in the L2 of the XCD whose SpMV workgroups gather it next; y, x, r (not read by the SpMV) move
// with nontemporal loads / stores.  band2 == 0: plain contiguous mapping.
// lazy (knob cgf_lazy): x is touched every second launch only.  A launch that finds no pending update and computes beta >=
// kLazyBeta does not read or write x; it leaves (alpha, beta) in sc[14], sc[15] and raises the pending flag sc[16 + next parity].
// The next launch reconstructs the previous direction from what it streams anyway, p_prev = (p - r) / beta_prev (p = r + beta_prev
// p_prev), and applies both updates: x += alpha_prev p_prev + alpha p.  6 instead of 7 vector passes on average.  Whoever
// detects convergence with an update pending applies it first (here, or k_cgf_flush after the loop).
constexpr double kLazyBeta = 0.25;
// kSplit (knob cgf_split): the second half of the lane's elements is requested only after the scalars are known, so that its
// loads are in flight while the first half is stored (read and write phases of the launch overlap).
template <int kCgV, int kSplit = 0>
static __global__ __launch_bounds__(256) void k_cgf_update(int64_t n, const double* y, double* p, double* x, double* r,
                                                     const double* part_spmv, int np_spmv, const double* part_rr_in, int np_rr,
                                                     double* part_rr_out, double* sc, double tol2, int32_t* ctl, int64_t band2,
                                                     int nt, int lazy, int parity) {
    __shared__ double red[8];
    // the stop flag is REQUESTED first and tested after the vector loads have been issued: its round trip overlaps with theirs
    // instead of preceding them (a stopped launch wastes its loads, which costs nothing that matters)
    const int32_t stop_flag = __builtin_nontemporal_load(ctl);
    const int64_t n2_all = n >> 1;
    const int64_t lo = band2 > 0 ? (int64_t)(blockIdx.x & 7) * band2 : 0;
    const int64_t n2 = band2 > 0 ? min(n2_all, lo + band2) : n2_all;   // end of this workgroup's element range
    const int64_t i0 = lo + (int64_t)(band2 > 0 ? (blockIdx.x >> 3) : blockIdx.x) * (256 * kCgV) + threadIdx.x;
    const double2* y2 = reinterpret_cast<const double2*>(y);
    double2* p2 = reinterpret_cast<double2*>(p);
    double2* x2 = reinterpret_cast<double2*>(x);
    double2* r2 = reinterpret_cast<double2*>(r);
    double2 yv[kCgV], pv[kCgV], xv[kCgV], rv[kCgV];
    typedef double v2f64k_t __attribute__((ext_vector_type(2)));
    // nt: bit 0 y, bit 1 x, bit 2 r, bit 3 p (loads; x and r also their stores) move with the nontemporal hint
    auto ld = [&](const double2* ptr, int64_t i, bool hint) -> double2 {
        if (hint) {
            const v2f64k_t t = __builtin_nontemporal_load(reinterpret_cast<const v2f64k_t*>(ptr + i));
            return make_double2(t.x, t.y);
        }
        return ptr[i];
    };
    auto st_x = [&](int64_t i, double2 v) {
        if (nt & 2)
            __builtin_nontemporal_store(v2f64k_t{v.x, v.y}, reinterpret_cast<v2f64k_t*>(x2 + i));
        else
            x2[i] = v;
    };
    const bool pend = lazy && sc[16 + parity] != 0.0;   // an update of x is pending from the previous launch
    const bool need_x = !lazy || pend;
    const double a_prev = pend ? sc[14] : 0.0, ib_prev = pend ? 1.0 / sc[15] : 0.0;
    // the producer's partials are requested BEFORE the vectors (loads return in order: issued after them, the partials would
    // arrive only when the whole vector phase has, and the reduction would start late); they are summed after the vector
    // loads have been issued
    constexpr int kPart = 8;   // partial pairs per lane held in registers (2048 pairs per workgroup); more are summed directly
    double2 pp[kPart];
#pragma unroll
    for (int j = 0; j < kPart; ++j) {
        const int i = threadIdx.x + j * 256;
        pp[j] = i < np_spmv ? *reinterpret_cast<const double2*>(part_spmv + 2 * i) : make_double2(0.0, 0.0);
    }
    constexpr int kPartR = 4;   // r.r partials per lane held in registers (1024 per workgroup)
    double qq[kPartR];
#pragma unroll
    for (int j = 0; j < kPartR; ++j) {
        const int i = threadIdx.x + j * 256;
        qq[j] = i < np_rr ? part_rr_in[i] : 0.0;
    }
    constexpr int kFirst = (kSplit && kCgV > 1) ? kCgV / 2 : kCgV;
#pragma unroll
    for (int k = 0; k < kFirst; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        pv[k] = ld(p2, ic, nt & 8), yv[k] = ld(y2, ic, nt & 1), rv[k] = ld(r2, ic, nt & 4);
        if (need_x) xv[k] = ld(x2, ic, nt & 2);
    }
    if (__syncthreads_or(stop_flag != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    double a = 0, b = 0;
#pragma unroll
    for (int j = 0; j < kPart; ++j) a += pp[j].x, b += pp[j].y;
    for (int i = threadIdx.x + kPart * 256; i < np_spmv; i += blockDim.x) a += part_spmv[2 * i], b += part_spmv[2 * i + 1];
    const double pAp = block_sum(a, red);
    const double yy = block_sum(b, red);
    double c_ = 0;
#pragma unroll
    for (int j = 0; j < kPartR; ++j) c_ += qq[j];
    for (int i = threadIdx.x + kPartR * 256; i < np_rr; i += blockDim.x) c_ += part_rr_in[i];
    const double rr = block_sum(c_, red);   // launch 0: seeded by k_krylov_init_fin
#pragma unroll
    for (int k = kFirst; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        pv[k] = ld(p2, ic, nt & 8), yv[k] = ld(y2, ic, nt & 1), rv[k] = ld(r2, ic, nt & 4);
        if (need_x) xv[k] = ld(x2, ic, nt & 2);
    }
    const bool last = blockIdx.x == gridDim.x - 1 && threadIdx.x == 0;
    const bool tail = (n & 1) && blockIdx.x == 0 && threadIdx.x == 0;
    if (rr <= tol2 * sc[0]) {   // converged by the previous update: r stays as it is; a pending update of x is applied
        if (pend) {
#pragma unroll
            for (int k = 0; k < kCgV; ++k) {
                const int64_t i = i0 + k * 256;
                if (i < n2) {
                    xv[k].x += a_prev * ((pv[k].x - rv[k].x) * ib_prev), xv[k].y += a_prev * ((pv[k].y - rv[k].y) * ib_prev);
                    st_x(i, xv[k]);
                }
            }
            if (tail) x[n - 1] += a_prev * ((p[n - 1] - r[n - 1]) * ib_prev);
        }
        // sc[18]: the pending update has been applied here (its flag cannot be cleared while other workgroups still read it)
        if (last) sc[3] = rr, ctl[0] = 1, sc[16 + (parity ^ 1)] = 0.0, sc[18] = 1.0;
        return;
    }
    const double alpha = pAp > 0.0 ? rr / pAp : 0.0;
    const double est = alpha * alpha * yy - rr;
    const double beta = (est > 0.0 && rr > 0.0) ? est / rr : 0.0;
    const bool defer = lazy && !pend && beta >= kLazyBeta;   // uniform: every workgroup computes the same beta
    if (!need_x && !defer) {   // rare: no update pending, but this one cannot be reconstructed later -> apply it now
#pragma unroll
        for (int k = 0; k < kCgV; ++k) {
            const int64_t i = i0 + k * 256;
            xv[k] = ld(x2, i < n2 ? i : 0, nt & 2);
        }
    }
    double acc = 0;
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) {
            if (!defer) {
                if (pend) xv[k].x += a_prev * ((pv[k].x - rv[k].x) * ib_prev), xv[k].y += a_prev * ((pv[k].y - rv[k].y) * ib_prev);
                xv[k].x += alpha * pv[k].x, xv[k].y += alpha * pv[k].y;
                st_x(i, xv[k]);
            }
            rv[k].x -= alpha * yv[k].x, rv[k].y -= alpha * yv[k].y;
            pv[k].x = rv[k].x + beta * pv[k].x, pv[k].y = rv[k].y + beta * pv[k].y;
            if (nt & 4)
                __builtin_nontemporal_store(v2f64k_t{rv[k].x, rv[k].y}, reinterpret_cast<v2f64k_t*>(r2 + i));
            else
                r2[i] = rv[k];
            p2[i] = pv[k];
            acc += rv[k].x * rv[k].x + rv[k].y * rv[k].y;
        }
    }
    if (tail) {
        const int64_t i = n - 1;
        if (!defer) {
            if (pend) x[i] += a_prev * ((p[i] - r[i]) * ib_prev);
            x[i] += alpha * p[i];
        }
        const double ri = r[i] - alpha * y[i];
        r[i] = ri, p[i] = ri + beta * p[i];
        acc += ri * ri;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part_rr_out[blockIdx.x] = s;
    if (last) {
        sc[3] = rr;
        ctl[1] += 1;
        if (defer) sc[14] = alpha, sc[15] = beta;
        sc[16 + (parity ^ 1)] = defer ? 1.0 : 0.0;
        if (!(pAp > 0.0)) ctl[2] = 1, ctl[0] = 1;   // not SPD / breakdown
    }
}
// after the iteration loop: an update of x may still be pending (convergence seen by k_cgf_fin, or maxit)
static __global__ __launch_bounds__(256) void k_cgf_flush(int64_t n, const double* p, const double* r, double* x, double* sc, const int32_t* ctl) {
    const int parity = ctl[1] & 1;   // the launch after the last executed update would have had this parity
    if (sc[16 + parity] == 0.0 || sc[18] != 0.0) return;   // nothing pending, or already applied by the launch that saw convergence
    const double a_prev = sc[14], ib_prev = 1.0 / sc[15];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] += a_prev * ((p[i] - r[i]) * ib_prev);
}
// host poll of the fused-update CG: explicit r.r of the last update -> sc[3], stop flag
static __global__ __launch_bounds__(256) void k_cgf_fin(const double* part_rr, int np, double* sc, double tol2, int32_t* ctl) {
    __shared__ double red[8];
    if (ctl[0] != 0) return;
    const double rr = sum_partials(part_rr, np, red);
    if (threadIdx.x == 0) {
        sc[3] = rr;
        if (rr <= tol2 * sc[0]) ctl[0] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// BiCGStab on the scaled system (non-symmetric operators: advection)
//   k_bicg_p   : rho = r0.r ; beta = (rho/rho_old)(alpha/omega) ; p = r + beta (p - omega v)
//   k_spmv     : v = At p, partial of r0.v
//   k_bicg_s   : alpha = rho / r0.v ; s = r - alpha v
//   k_spmv     : t = At s, partials of t.s (w = s) and t.t
//   k_bicg_xr  : omega = t.s / t.t ; x += alpha p + omega s ; r = s - omega t ; partials r0.r and r.r
// ---------------------------------------------------------------------------------------------------------------
// The three vector kernels are single-shot like the CG ones: workgroup b owns kBiV * 256 consecutive double2 elements, every
// lane issues all of its 16-byte loads first and only then re-reduces the producer's partials (grid = bicg_grid(n)).
constexpr int kBiV = 4;
// dead streams of the BiCGStab vector kernels (everything but the p and s the next SpMV reads) move with the nontemporal hint
// (C5, same box: solve 666.3 -> 647.6 ms); -DFDAPDE_BICG_NT=0 restores the default policy
#ifndef FDAPDE_BICG_NT
#define FDAPDE_BICG_NT 1
#endif
typedef double v2f64b_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 bi_ld(const double2* p, int64_t i) {
#if FDAPDE_BICG_NT
    const v2f64b_t t = __builtin_nontemporal_load(reinterpret_cast<const v2f64b_t*>(p + i));
    return make_double2(t.x, t.y);
#else
    return p[i];
#endif
}
__device__ __forceinline__ void bi_st(double2* p, int64_t i, double2 v) {
#if FDAPDE_BICG_NT
    __builtin_nontemporal_store(v2f64b_t{v.x, v.y}, reinterpret_cast<v2f64b_t*>(p + i));
#else
    p[i] = v;
#endif
}
#define BI_LD(ptr, i) bi_ld(ptr, i)
#define BI_ST(ptr, i, v) bi_st(ptr, i, v)
static __global__ __launch_bounds__(256) void k_bicg_p(int64_t n, const double* r, const double* v, double* p,
                                                 const double* part_in /* (r0.r, r.r) pairs */, int np_in, double* sc,
                                                 int first, int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kBiV) + threadIdx.x;
    const double2* r2 = reinterpret_cast<const double2*>(r);
    const double2* v2 = reinterpret_cast<const double2*>(v);
    double2* p2 = reinterpret_cast<double2*>(p);
    double2 rv[kBiV], vv[kBiV], pv[kBiV];
    PRELOAD_PAIRS(pp, part_in, first ? 0 : np_in)
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        rv[k] = BI_LD(r2, ic);
        if (!first) vv[k] = BI_LD(v2, ic), pv[k] = BI_LD(p2, ic);
    }
    double a = 0, a_unused = 0;
    SUM_PAIRS(pp, part_in, first ? 0 : np_in, a, a_unused)
    (void)a_unused;
    const double rho_new = first ? sc[9] : block_sum(a, red);
    const double rho = sc[4], alpha = sc[5], omega = sc[6];
    const double beta = first ? 0.0 : (rho_new / rho) * (alpha / omega);
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) {
            double2 o = rv[k];
            if (!first) o.x += beta * (pv[k].x - omega * vv[k].x), o.y += beta * (pv[k].y - omega * vv[k].y);
            p2[i] = o;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = first ? r[n - 1] : r[n - 1] + beta * (p[n - 1] - omega * v[n - 1]);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        sc[7] = rho_new;
        if (rho_new == 0.0) ctl[2] = 1;
    }
}
static __global__ __launch_bounds__(256) void k_bicg_s(int64_t n, const double* r, const double* v, double* s,
                                                 const double* part_in /* (r0.v, .) */, int np_in, double* sc,
                                                 int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kBiV) + threadIdx.x;
    const double2* r2 = reinterpret_cast<const double2*>(r);
    const double2* v2 = reinterpret_cast<const double2*>(v);
    double2* s2 = reinterpret_cast<double2*>(s);
    double2 rv[kBiV], vv[kBiV];
    PRELOAD_PAIRS(pp, part_in, np_in)
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        rv[k] = BI_LD(r2, ic), vv[k] = BI_LD(v2, ic);
    }
    double a = 0, a_unused = 0;
    SUM_PAIRS(pp, part_in, np_in, a, a_unused)
    (void)a_unused;
    const double r0v = block_sum(a, red);
    const double alpha = r0v != 0.0 ? sc[7] / r0v : 0.0;
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) s2[i] = make_double2(rv[k].x - alpha * vv[k].x, rv[k].y - alpha * vv[k].y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) s[n - 1] = r[n - 1] - alpha * v[n - 1];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        sc[8] = alpha;
        if (r0v == 0.0) ctl[2] = 1;
    }
}
static __global__ __launch_bounds__(256) void k_bicg_xr(int64_t n, const double* p, const double* s, const double* t,
                                                  const double* r0, double* x, double* r,
                                                  const double* part_in /* (t.s, t.t) */, int np_in, double* part_out,
                                                  const double* sc, int32_t* ctl, const uint8_t* owned) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kBiV) + threadIdx.x;
    const double2* p2 = reinterpret_cast<const double2*>(p);
    const double2* s2 = reinterpret_cast<const double2*>(s);
    const double2* t2 = reinterpret_cast<const double2*>(t);
    const double2* q2 = reinterpret_cast<const double2*>(r0);
    double2* x2 = reinterpret_cast<double2*>(x);
    double2* r2 = reinterpret_cast<double2*>(r);
    double2 pv[kBiV], sv[kBiV], tv[kBiV], qv[kBiV], xv[kBiV];
    PRELOAD_PAIRS(pp, part_in, np_in)
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        pv[k] = BI_LD(p2, ic), sv[k] = BI_LD(s2, ic), tv[k] = BI_LD(t2, ic), qv[k] = BI_LD(q2, ic), xv[k] = BI_LD(x2, ic);
    }
    double a = 0, b = 0;
    SUM_PAIRS(pp, part_in, np_in, a, b)
    const double ts = block_sum(a, red);
    const double tt = block_sum(b, red);
    const double omega = tt > 0.0 ? ts / tt : 0.0;
    const double alpha = sc[8];
    double d0 = 0, d1 = 0;
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) {
            BI_ST(x2, i, make_double2(xv[k].x + alpha * pv[k].x + omega * sv[k].x, xv[k].y + alpha * pv[k].y + omega * sv[k].y));
            const double2 ri = make_double2(sv[k].x - omega * tv[k].x, sv[k].y - omega * tv[k].y);
            BI_ST(r2, i, ri);
            const bool o0 = !owned || owned[2 * i], o1 = !owned || owned[2 * i + 1];
            d0 += (o0 ? qv[k].x * ri.x : 0.0) + (o1 ? qv[k].y * ri.y : 0.0);
            d1 += (o0 ? ri.x * ri.x : 0.0) + (o1 ? ri.y * ri.y : 0.0);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        x[i] += alpha * p[i] + omega * s[i];
        const double ri = s[i] - omega * t[i];
        r[i] = ri;
        if (!owned || owned[i]) d0 += r0[i] * ri, d1 += ri * ri;
    }
    const double s0 = block_sum(d0, red);
    const double s1 = block_sum(d1, red);
    if (threadIdx.x == 0) part_out[2 * blockIdx.x] = s0, part_out[2 * blockIdx.x + 1] = s1;
}
#undef BI_LD
#undef BI_ST
// Shadow residual of BiCGStab other than r0 itself (knob bicg_shadow; tools/c5_iter_spread.py): mode 1 = pseudo-random entries in (-1, 1) from a hash of
// the DOF's position (the same on every run), mode 2 = r0 with every entry scaled by a pseudo-random factor in (0.5, 1.5).  part[b] = partial of (shadow, r).
static __global__ __launch_bounds__(256) void k_bicg_shadow(int64_t n, int mode, const double* r, double* r0, double* part) {
    __shared__ double red[8];
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long z = (unsigned long long)i + 0x9e3779b97f4a7c15ull;   // splitmix64
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);   // [0, 1)
        const double v = mode == 1 ? 2.0 * u - 1.0 : r[i] * (0.5 + u);
        r0[i] = v;
        a += v * r[i];
    }
    const double s = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// multi-GPU BiCGStab: t.t over the owned rows of the ASSEMBLED t (the SpMV's fused y.y only sees this rank's sub-assembled
// part); per-workgroup partials, then out = (t.s already summed over ranks, local t.t) for the scalar all-reduce of out[1]
static __global__ __launch_bounds__(256) void k_sq_owned(int64_t n, const double* t, const uint8_t* owned, double* part, const int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (owned[i]) a += t[i] * t[i];
    const double s = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// sum of squares of a vector (entries that are not finite count as huge): the size of a claimed solution, for the singularity guard of fdapde_solve
static __global__ __launch_bounds__(256) void k_sq_norm(int64_t n, const double* t, double* part) {
    __shared__ double red[8];
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = t[i];
        a += isfinite(v) ? v * v : 1e300;
    }
    const double s = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
static __global__ __launch_bounds__(256) void k_bicg_tt_fin(const double* part, int np, const double* ts_src, double* out) {
    __shared__ double red[8];
    const double s = sum_partials(part, np, red);
    if (threadIdx.x == 0) out[0] = ts_src[0], out[1] = s;
}
// closes a BiCGStab iteration: rho <- rho_new, alpha, omega = t.s/t.t recomputed from the same partials, stop test
static __global__ __launch_bounds__(256) void k_bicg_fin(const double* part_ts, int np_ts, const double* part_rr, int np_rr,
                                                   double* sc, double tol2, int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np_ts; i += blockDim.x) a += part_ts[2 * i], b += part_ts[2 * i + 1];
    const double ts = block_sum(a, red);
    const double tt = block_sum(b, red);
    double c = 0;
    for (int i = threadIdx.x; i < np_rr; i += blockDim.x) c += part_rr[2 * i + 1];
    const double rr = block_sum(c, red);
    if (threadIdx.x == 0) {
        const double omega = tt > 0.0 ? ts / tt : 0.0;
        sc[4] = sc[7], sc[5] = sc[8], sc[6] = omega, sc[3] = rr;
        ctl[1] += 1;
        if (omega == 0.0) ctl[2] = 1;
        if (!(rr <= 1e16 * sc[0])) ctl[2] = 1;   // the residual has grown by 1e8 (or is no number): diverged -- reported like a breakdown, no point in going on
        if (rr <= tol2 * sc[0] || ctl[2]) ctl[0] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// multi-GPU: interface ("halo") exchange.  Every rank holds the sub-assembled operator of its own cells; an operator
// application is y_p = A_p x_p followed by the sum of the interface entries over the ranks that share them.  The
// interface entries are packed into one globally indexed buffer (zero elsewhere), summed by ONE ncclAllReduce together
// with the rank's partial of the fused dot product (slot n_if), and unpacked.  dot(x, A x) = sum_p x_p . (A_p x_p) needs
// no weighting; dots of assembled vectors count every DOF once through the `owned` mask.
// ---------------------------------------------------------------------------------------------------------------
// One lane per GLOBAL interface slot: inv[j] = this rank's DOF of slot j or -1 (zero written where the rank has no DOF, so no
// memset is needed); workgroup 0 also folds the local dot partials (stride 2) into buf[n_if], buf[n_if + 1].
static __global__ __launch_bounds__(256) void k_halo_pack_all(int64_t n_if, const int32_t* inv, const double* v, double* buf,
                                                        const double* part, int np) {
    __shared__ double red[8];
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_if) {
        const int32_t d = inv[j];
        buf[j] = d >= 0 ? v[d] : 0.0;
    }
    if (blockIdx.x == 0) {
        double a = 0, b = 0;
        if (part != nullptr)
            for (int k = threadIdx.x; k < np; k += blockDim.x) a += part[2 * k], b += part[2 * k + 1];
        const double sa = block_sum(a, red);
        const double sb = block_sum(b, red);
        if (threadIdx.x == 0) buf[n_if] = sa, buf[n_if + 1] = sb;
    }
}
// neighbour-only exchange: send buffer = this rank's sub-assembled values at the DOFs it shares, one segment per peer; block 0 also
// folds the SpMV's fused dot partials (as k_halo_pack_all)
static __global__ __launch_bounds__(256) void k_peer_pack(int64_t n_send, const int32_t* send_dof, const double* v, double* sendbuf, const double* part,
                                                    int np, double* scal) {
    __shared__ double red[8];
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_send) sendbuf[j] = v[send_dof[j]];
    if (blockIdx.x == 0) {
        double a = 0, b = 0;
        if (part != nullptr)
            for (int k = threadIdx.x; k < np; k += blockDim.x) a += part[2 * k], b += part[2 * k + 1];
        const double sa = block_sum(a, red);
        const double sb = block_sum(b, red);
        if (threadIdx.x == 0) scal[0] = sa, scal[1] = sb;
    }
}
// in-process direct transport: the receive buffer straight from the peers' send buffers (device pointers, one per entry)
static __global__ void k_peer_fetch(int64_t n, const double* const* src, double* dst) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = *src[j];
}
// ... and a small vector summed over the ranks' slots in ascending rank order (every rank computes the same bits)
static __global__ void k_slot_sum(int world, int count, const double* const* slot, int64_t offset, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int r = 0; r < world; ++r) s += slot[r][offset + i];
    out[i] = s;
}
// summed value of every local interface DOF: the contributions of the ranks sharing it added in ASCENDING RANK ORDER (this rank's own
// among them), so that every sharer computes the same bits; buf[k] for the kernels that read interface rows from there, v on request
static __global__ void k_peer_sum(int64_t n_loc_if, const int32_t* dof, const int32_t* src_off, const int32_t* src, const double* recvbuf, double* v,
                           double* buf, int write_v) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_loc_if) return;
    const int32_t d = dof[k];
    double s = 0.0;
    for (int32_t q = src_off[k]; q < src_off[k + 1]; ++q) s += src[q] < 0 ? v[d] : recvbuf[src[q]];
    buf[k] = s;
    if (write_v) v[d] = s;
}
static __global__ void k_halo_unpack(int64_t n_loc_if, const int32_t* dof, const int32_t* pos, const double* buf, double* v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_loc_if) v[dof[i]] = buf[pos[i]];
}
// out[0] = sum(part[0..np)) in the fixed order; single workgroup
static __global__ __launch_bounds__(256) void k_reduce_partials(const double* part, int np, double* out) {
    __shared__ double red[8];
    const double s = sum_partials(part, np, red);
    if (threadIdx.x == 0) out[0] = s;
}
static __global__ void k_diag_extract(int64_t n, const int32_t* diag, const double* vals, double* d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = vals[diag[i]];
}
// Jacobi scale from an already summed diagonal (multi-GPU)
static __global__ void k_jacobi_scale_from_diag(int64_t n, const double* d, const uint8_t* bnd, int use_bnd, double* scale, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool b = use_bnd && bnd[i];
    if (!b && !(d[i] > 0.0)) atomicOr(flag, 1);
    scale[i] = b ? 0.0 : jacobi_scale_of(d[i]);
}

// u = scale * x + gt   (back to the unscaled unknowns, Dirichlet values restored)
static __global__ void k_unscale(int64_t n, const double* scale, const double* x, const double* gt, double* u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) u[i] = scale[i] * x[i] + gt[i];
}

// ---------------------------------------------------------------------------------------------------------------
// numbering changes at the boundary (reference numbering <-> internal numbering)
// ---------------------------------------------------------------------------------------------------------------
static __global__ void k_gather_f64(int64_t n, const int32_t* idx, const double* src, double* dst) {   // dst[i] = src[idx[i]]
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
static __global__ void k_scatter_f64(int64_t n, const int32_t* idx, const double* src, double* dst) {  // dst[idx[i]] = src[i]
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}
// export of stiff() after a Dirichlet solve: FEMSolverBase::set_dirichlet_bc (fem_solver_base.h:148-149) zeroes the
// boundary rows and puts 1 on their diagonal; 16 lanes per row, output in reference slots
static __global__ __launch_bounds__(256) void k_export_values(int64_t n, const int32_t* rowptr, const int32_t* colidx,
                                                       const double* vals, const int32_t* slot_i2e, const uint8_t* bnd,
                                                       int zero_bnd_rows, double* out) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (row >= n) return;
    const bool z = zero_bnd_rows && bnd[row];
    for (int k = rowptr[row] + l; k < rowptr[row + 1]; k += 16)
        out[slot_i2e[k]] = z ? (colidx[k] == row ? 1.0 : 0.0) : vals[k];
}
// row-sum lumping (fdaPDE/linear_algebra/lumping.h:30-41): out[row] = sum of the row's entries; 16 lanes per row, fixed order
static __global__ __launch_bounds__(256) void k_row_sums(int64_t n, const int32_t* rowptr, const double* vals, double* out) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const bool ok = row < n;
    double a = 0;
    if (ok)
        for (int k = rowptr[row] + l; k < rowptr[row + 1]; k += 16) a += vals[k];
    a = team_sum<16>(a);
    if (ok && l == 0) out[row] = a;
}
static __global__ void k_fill_f64(int64_t n, double v, double* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}
// force export after a Dirichlet solve: force_[i] = g[i] on boundary DOFs (fem_solver_base.h:152)
static __global__ void k_force_bc(int64_t n, const uint8_t* bnd, const double* g, double* f) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && bnd[i]) f[i] = g[i];
}

#undef PRELOAD_PAIRS
#undef SUM_PAIRS

}  // namespace fdapde_hip
#endif
