// kernels_multirhs.h -- Q right-hand sides against ONE prepared system (the factor-once handle with several columns:
// SMW's A^-1 U, fdaPDE/linear_algebra/smw.h:46-48): Q independent fused-update CG iterations share every pass over the matrix.
// Vectors are interleaved (row i holds its Q values contiguously), so a gathered column index fetches Q operands with one or
// two 16/32-byte loads and the matrix (12 bytes per entry, full pattern, Jacobi-scaled) is streamed once per iteration for all
// Q systems instead of once per system.  Per iteration: k_spmm_full (Y = At P, partials of p.y and y.y per system),
// k_q_scalars (one workgroup: alpha, beta, explicit r.r, convergence per system), k_q_update (x, r, p).  Converged systems
// freeze (alpha = beta = 0) until all have converged.  Same recurrences as k_cgf_update (kernels_krylov.h).
#ifndef FDAPDE_KERNELS_MULTIRHS_H
#define FDAPDE_KERNELS_MULTIRHS_H

#include <hip/hip_runtime.h>

#include "internal.h"
#include "kernels_reduce.h"

namespace fdapde_hip {

// scalars of the Q systems: sc[0..Q) b.b, [Q..2Q) r.r, [2Q..3Q) alpha, [3Q..4Q) beta, [4Q..5Q) 1 if converged
template <int Q>
static __global__ __launch_bounds__(256) void k_q_init(int64_t n, const double* b_ext /* Q columns of n, reference numbering */,
                                                const int32_t* dof_i2e, const double* scale, double* X, double* R, double* P,
                                                double* part_rr /* [grid][Q] */) {
    __shared__ double red[8];
    double acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double s = scale[i];
        const int64_t e = dof_i2e[i];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const double r = s * b_ext[(int64_t)q * n + e];
            X[i * Q + q] = 0.0, R[i * Q + q] = r, P[i * Q + q] = r;
            acc[q] += r * r;
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const double t = block_sum(acc[q], red);
        if (threadIdx.x == 0) part_rr[(int64_t)blockIdx.x * Q + q] = t;
    }
}
template <int Q> __global__ __launch_bounds__(256) void k_q_init_fin(const double* part_rr, int np, double* sc, int32_t* ctl) {
    __shared__ double red[8];
    for (int q = 0; q < Q; ++q) {
        double a = 0;
        for (int i = threadIdx.x; i < np; i += blockDim.x) a += part_rr[(int64_t)i * Q + q];
        const double rr = block_sum(a, red);
        if (threadIdx.x == 0) sc[q] = rr, sc[Q + q] = rr, sc[2 * Q + q] = 0, sc[3 * Q + q] = 0, sc[4 * Q + q] = 0;
    }
    if (threadIdx.x == 0) ctl[0] = 0, ctl[1] = 0, ctl[2] = 0;
}

// sum over the threads of a workgroup that own the same pair of systems (threadIdx.x % H): sh = 2 * 256 doubles of LDS;
// thread q < 2 H returns the total of system q, the others 0
template <int H> __device__ __forceinline__ double pair_block_sum(double2 v, double* sh) {
    __syncthreads();
    sh[threadIdx.x] = v.x, sh[256 + threadIdx.x] = v.y;
    __syncthreads();
    double s = 0;
    if (threadIdx.x < 2 * H) {
        const int hq = threadIdx.x >> 1, comp = threadIdx.x & 1;
        for (int i = hq; i < 256; i += H) s += sh[comp * 256 + i];
    }
    return s;
}

// Y = At P for the Q interleaved vectors; 8 lanes per row, one matrix entry per lane and step, the Q operands of an entry
// fetched by that lane as Q/2 double2 loads (a row of the interleaved vector is contiguous).  Measured on C3, Q = 8: 262 us;
// the variant in which Q/2 lanes share an entry (coalesced 64-byte operand rows, but 4 x fewer entries per instruction): 365 us.
//   partial[(b * 2 + 0) * Q + q] = p.y, [(b * 2 + 1) * Q + q] = y.y
template <int Q>
static __global__ __launch_bounds__(256) void k_spmm_full(int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals,
                                                   const double* X, double* Y, double* partial, const int32_t* stop) {
    static_assert(Q % 2 == 0, "rows are moved as double2");
    constexpr int T = 8;
    __shared__ double red[8];
    if (__syncthreads_or(*stop != 0)) return;
    const int team = threadIdx.x / T, l = threadIdx.x % T;
    double d_py[Q], d_yy[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) d_py[q] = 0, d_yy[q] = 0;
    for (int64_t row0 = (int64_t)blockIdx.x * (256 / T); row0 < n; row0 += (int64_t)gridDim.x * (256 / T)) {
        const int64_t row = row0 + team;
        const bool ok = row < n;
        const int rs = ok ? rowptr[row] : 0, re = ok ? rowptr[row + 1] : 0;
        double acc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = 0;
        for (int k = rs + l; k < re; k += T) {
            const double v = vals[k];
            const double2* xp = reinterpret_cast<const double2*>(X + (int64_t)colidx[k] * Q);
#pragma unroll
            for (int h = 0; h < Q / 2; ++h) {
                const double2 t = xp[h];
                acc[2 * h] += v * t.x, acc[2 * h + 1] += v * t.y;
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = team_sum<T>(acc[q]);   // every lane of the team holds the row's Q results
        if (ok && l == 0) {
            double2* yp = reinterpret_cast<double2*>(Y + row * Q);
            const double2* xr = reinterpret_cast<const double2*>(X + row * Q);
#pragma unroll
            for (int h = 0; h < Q / 2; ++h) {
                const double2 t = xr[h];
                yp[h] = make_double2(acc[2 * h], acc[2 * h + 1]);
                d_py[2 * h] += t.x * acc[2 * h], d_py[2 * h + 1] += t.y * acc[2 * h + 1];
                d_yy[2 * h] += acc[2 * h] * acc[2 * h], d_yy[2 * h + 1] += acc[2 * h + 1] * acc[2 * h + 1];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const double a = block_sum(d_py[q], red);
        const double b = block_sum(d_yy[q], red);
        if (threadIdx.x == 0) partial[((int64_t)blockIdx.x * 2 + 0) * Q + q] = a, partial[((int64_t)blockIdx.x * 2 + 1) * Q + q] = b;
    }
}

// one workgroup: per system alpha = rr / p.y, beta = (alpha^2 y.y - rr) / rr with the EXPLICIT rr of the previous update
template <int Q>
static __global__ __launch_bounds__(256) void k_q_scalars(const double* part_spmm, int np, const double* part_rr, int np_rr, double* sc,
                                                   double tol2, int32_t* ctl) {
    __shared__ double sa[256], sb[256], sc_[256];
    __shared__ int flags[2];
    if (ctl[0] != 0) return;
    const int q = threadIdx.x % Q, i0 = threadIdx.x / Q;
    double a = 0, b = 0, c = 0;
    for (int i = i0; i < np; i += 256 / Q) a += part_spmm[((int64_t)i * 2 + 0) * Q + q], b += part_spmm[((int64_t)i * 2 + 1) * Q + q];
    for (int i = i0; i < np_rr; i += 256 / Q) c += part_rr[(int64_t)i * Q + q];
    sa[threadIdx.x] = a, sb[threadIdx.x] = b, sc_[threadIdx.x] = c;
    if (threadIdx.x == 0) flags[0] = 1, flags[1] = 0;   // all done, breakdown
    __syncthreads();
    if (threadIdx.x < Q) {
        double py = 0, yy = 0, rr = 0;
        for (int i = threadIdx.x; i < 256; i += Q) py += sa[i], yy += sb[i], rr += sc_[i];
        const bool done = rr <= tol2 * sc[q];
        double alpha = 0, beta = 0;
        if (!done) {
            if (py > 0.0) {
                alpha = rr / py;
                const double est = alpha * alpha * yy - rr;
                beta = (est > 0.0 && rr > 0.0) ? est / rr : 0.0;
            } else
                atomicOr(&flags[1], 1);   // not SPD
            atomicAnd(&flags[0], 0);
        }
        sc[Q + q] = rr, sc[2 * Q + q] = alpha, sc[3 * Q + q] = beta, sc[4 * Q + q] = done ? 1.0 : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (flags[1]) ctl[2] = 1;
        if (flags[0] || flags[1]) ctl[0] = 1;
        else ctl[1] += 1;
    }
}

// x += alpha p ; r -= alpha y ; p = r + beta p for the Q systems: the interleaved vectors are walked as flat double2 arrays;
// a thread always meets the same pair of systems (its index mod Q/2); kQV elements per lane in flight
constexpr int kQV = 4;
template <int Q>
static __global__ __launch_bounds__(256) void k_q_update(int64_t n, const double* Y, double* P, double* X, double* R, const double* sc,
                                                  double* part_rr, const int32_t* ctl) {
    static_assert(Q == 2 || Q == 4 || Q == 8, "a lane moves one double2 = one pair of systems");
    constexpr int H = Q / 2;
    __shared__ double sh[512];
    if (__syncthreads_or(ctl[0] != 0)) return;
    const int hq = threadIdx.x % H;
    const double a0 = sc[2 * Q + 2 * hq], a1 = sc[2 * Q + 2 * hq + 1], b0 = sc[3 * Q + 2 * hq], b1 = sc[3 * Q + 2 * hq + 1];
    const int64_t nh = n * H;
    const double2* y2 = reinterpret_cast<const double2*>(Y);
    double2* p2 = reinterpret_cast<double2*>(P);
    double2* x2 = reinterpret_cast<double2*>(X);
    double2* r2 = reinterpret_cast<double2*>(R);
    double2 acc = make_double2(0, 0);
    // a bounded grid (the consumer re-reduces gridDim.x partials per system) striding over chunks of 256 * kQV elements
    for (int64_t i0 = (int64_t)blockIdx.x * (256 * kQV) + threadIdx.x; i0 < nh; i0 += (int64_t)gridDim.x * (256 * kQV)) {
        double2 yv[kQV], pv[kQV], xv[kQV], rv[kQV];
#pragma unroll
        for (int k = 0; k < kQV; ++k) {
            const int64_t e = i0 + k * 256, ec = e < nh ? e : 0;
            pv[k] = p2[ec];
            {   // y, x, r are not read by the next SpMM: nontemporal, the L2s are left to P
                typedef double v2f64q_t __attribute__((ext_vector_type(2)));
                const v2f64q_t a_ = __builtin_nontemporal_load(reinterpret_cast<const v2f64q_t*>(y2 + ec));
                const v2f64q_t b_ = __builtin_nontemporal_load(reinterpret_cast<const v2f64q_t*>(x2 + ec));
                const v2f64q_t c_ = __builtin_nontemporal_load(reinterpret_cast<const v2f64q_t*>(r2 + ec));
                yv[k] = make_double2(a_.x, a_.y), xv[k] = make_double2(b_.x, b_.y), rv[k] = make_double2(c_.x, c_.y);
            }
        }
#pragma unroll
        for (int k = 0; k < kQV; ++k) {
            const int64_t e = i0 + k * 256;
            if (e < nh) {
                xv[k].x += a0 * pv[k].x, xv[k].y += a1 * pv[k].y;
                rv[k].x -= a0 * yv[k].x, rv[k].y -= a1 * yv[k].y;
                pv[k].x = rv[k].x + b0 * pv[k].x, pv[k].y = rv[k].y + b1 * pv[k].y;
                {
                    typedef double v2f64q_t __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(v2f64q_t{xv[k].x, xv[k].y}, reinterpret_cast<v2f64q_t*>(x2 + e));
                    __builtin_nontemporal_store(v2f64q_t{rv[k].x, rv[k].y}, reinterpret_cast<v2f64q_t*>(r2 + e));
                }
                p2[e] = pv[k];
                acc.x += rv[k].x * rv[k].x, acc.y += rv[k].y * rv[k].y;
            }
        }
    }
    const double t = pair_block_sum<H>(acc, sh);
    if (threadIdx.x < Q) part_rr[(int64_t)blockIdx.x * Q + threadIdx.x] = t;
}

// u_q = scale * x_q, written in the reference numbering: out[q * n + dof_i2e[i]]
template <int Q>
static __global__ void k_q_unscale(int64_t n, const double* X, const double* scale, const int32_t* dof_i2e, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = scale[i];
    const int64_t e = dof_i2e[i];
#pragma unroll
    for (int q = 0; q < Q; ++q) out[(int64_t)q * n + e] = s * X[i * Q + q];
}

}  // namespace fdapde_hip
#endif
