// kernels_gmres.h -- restarted GMRES(m) on the Jacobi-scaled system, the LAST stage of FDAPDE_SOLVER_AUTO (CG -> BiCGStab with restarts -> GMRES):
// the reference factorises whatever it is given (fem_linear_elliptic_solver.h:38-47); BiCGStab's recurrences stall or break down on
// advection-dominated operators (cell Peclet numbers of 10^2 - 10^3) where the minimal-residual iteration still converges.
//
// One Arnoldi step = the solver's SpMV (eng_solve.hip launch_spmv, whatever layout the scaled matrix lives in) + classical Gram-Schmidt applied
// TWICE (CGS2: as stable as modified Gram-Schmidt, but every pass is ONE batched reduction over all basis vectors instead of j dependent dots):
//     h = V^T w (k_gm_dots: grid (stripes, j + 1), then k_gm_reduce)     w -= V h (k_gm_axpy; the second pass also leaves |w|^2 partials)
// + a one-workgroup kernel that rotates the new Hessenberg column (Givens), updates the least-squares right-hand side and raises the stop flag
// when its last entry -- the residual norm of the iterate this basis would give -- is below tol |b| (k_gm_hess).  Nothing returns to the host
// inside a restart cycle: every kernel leaves at once when the stop flag is up, the host reads flag and scalars once per cycle, after the
// cycle's end (triangular solve, x += V y, TRUE residual b - A x, which is what the next cycle starts from and what the final relres reports).
// Memory: (m + 1) n doubles for the basis (m = 50, C5's 5.36 M rows: 2.2 GB).
#ifndef FDAPDE_KERNELS_GMRES_H
#define FDAPDE_KERNELS_GMRES_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace fdapde_hip {

constexpr int kGmStripes = 256;   // partial sums per basis vector

// small state of a cycle (device doubles): layout of `gs`
//   [0 .. m]            g: right-hand side of the least-squares problem (g[j + 1] = residual estimate, signed)
//   [m+1 .. 2m]         cs, [2m+1 .. 3m] sn: Givens rotations
//   [3m+1 .. 4m+2]      hcol: the current column, h[0 .. j + 1]
//   [4m+3 .. 5m+2]      y
//   [5m+3]              1 / h[j + 1][j] of the current step;  [5m+4] steps done in this cycle;  [5m+5] spare
//   [5m+8 ...]          R: column j at [5m+8 + j (m + 1) + i], i <= j (upper triangular after the rotations)
__host__ __device__ inline int gm_state_doubles(int m) { return 5 * m + 8 + m * (m + 1); }

// right-hand side of the scaled interior system, kept for the true residuals: bt = scale (f - A gt)   (y holds A gt)
static __global__ void k_gm_rhs(int64_t n, const double* f, const double* y, const double* scale, double* bt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bt[i] = scale[i] * (f[i] - y[i]);
}
// start of a cycle: v0 = r / |r| (|r|^2 = sc[3], left by the initialisation or by the last cycle's true residual), g = (|r|, 0, ...)
static __global__ void k_gm_cycle_init(int64_t n, int m, const double* r, const double* sc, double* gs, double* V0, const int32_t* ctl) {
    if (ctl[0] != 0) return;
    const double beta = sqrt(sc[3]);
    const double inv = beta > 0.0 ? 1.0 / beta : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) V0[i] = r[i] * inv;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        gs[0] = beta;
        for (int k = 1; k <= m; ++k) gs[k] = 0.0;
        gs[5 * m + 4] = 0.0;
    }
}
__device__ __forceinline__ double gm_block_sum(double v, double* red) {   // 256 threads
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
// part[i * kGmStripes + s] = V_i . w over stripe s            grid (kGmStripes, j + 1)
static __global__ __launch_bounds__(256) void k_gm_dots(int64_t n, const double* V, const double* w, double* part, const int32_t* ctl) {
    __shared__ double red[4];
    if (ctl[0] != 0) return;
    const double* v = V + (int64_t)blockIdx.y * n;
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a += v[i] * w[i];
    const double s = gm_block_sum(a, red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * kGmStripes + blockIdx.x] = s;
}
// h_pass[i] = sum of the stripes; hcol[i] = h_pass[i] (first pass) or += (second pass)            grid (j + 1)
static __global__ __launch_bounds__(256) void k_gm_reduce(const double* part, double* h_pass, double* hcol, int second, const int32_t* ctl) {
    __shared__ double red[4];
    if (ctl[0] != 0) return;
    const double s = gm_block_sum(threadIdx.x < kGmStripes ? part[(int64_t)blockIdx.x * kGmStripes + threadIdx.x] : 0.0, red);
    if (threadIdx.x == 0) h_pass[blockIdx.x] = s, hcol[blockIdx.x] = second ? hcol[blockIdx.x] + s : s;
}
// w -= sum_i h_pass[i] V_i ; with_norm: part2[b] = |w|^2 over the block's share
static __global__ __launch_bounds__(256) void k_gm_axpy(int64_t n, int nv, const double* V, const double* h_pass, double* w, double* part2, int with_norm,
                                                        const int32_t* ctl) {
    __shared__ double red[4];
    if (ctl[0] != 0) return;
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double x = w[i];
        for (int k = 0; k < nv; ++k) x -= h_pass[k] * V[(int64_t)k * n + i];
        w[i] = x, a += x * x;
    }
    if (with_norm) {
        const double s = gm_block_sum(a, red);
        if (threadIdx.x == 0) part2[blockIdx.x] = s;
    }
}
// the new column of the Hessenberg matrix: h[j + 1] = |w| (from np partials), earlier rotations applied, the new one computed, g updated; stop flag
// when the residual estimate meets the tolerance (relative to |b|: sc[0]), when the step budget is used up, or on a (happy) breakdown |w| = 0
static __global__ __launch_bounds__(256) void k_gm_hess(int j, int m, const double* part2, int np, double* gs, double* sc, int32_t* ctl, double tol2, int maxit) {
    __shared__ double red[4];
    if (ctl[0] != 0) return;
    double a = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) a += part2[i];
    const double ww = gm_block_sum(a, red);
    if (threadIdx.x != 0) return;
    double* g = gs;
    double* cs = gs + m + 1;
    double* sn = gs + 2 * m + 1;
    double* h = gs + 3 * m + 1;
    double* R = gs + 5 * m + 8 + (int64_t)j * (m + 1);
    const double hn = sqrt(ww);
    h[j + 1] = hn;
    for (int i = 0; i < j; ++i) {
        const double t = cs[i] * h[i] + sn[i] * h[i + 1];
        h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
        h[i] = t;
    }
    const double d = sqrt(h[j] * h[j] + hn * hn);
    const double c = d > 0.0 ? h[j] / d : 1.0, s = d > 0.0 ? hn / d : 0.0;
    cs[j] = c, sn[j] = s;
    h[j] = d;
    g[j + 1] = -s * g[j];
    g[j] = c * g[j];
    for (int i = 0; i <= j; ++i) R[i] = h[i];
    gs[5 * m + 3] = hn > 0.0 ? 1.0 / hn : 0.0;
    gs[5 * m + 4] = (double)(j + 1);
    const double res2 = g[j + 1] * g[j + 1];
    sc[3] = res2;   // (the estimate; the cycle's end overwrites it with the true residual)
    const int it = ctl[1] + 1;
    ctl[1] = it;
    if (!(d > 0.0) || !isfinite(res2)) ctl[2] = 1, ctl[0] = 1;                                  // a singular Hessenberg column: breakdown
    else if (res2 <= tol2 * sc[0] || hn == 0.0 || it >= maxit) ctl[0] = 1;
}
static __global__ void k_gm_next(int64_t n, const double* w, const double* gs, int m, double* Vn, const int32_t* ctl) {
    if (ctl[0] != 0) return;
    const double inv = gs[5 * m + 3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) Vn[i] = w[i] * inv;
}
// end of a cycle (runs whatever the stop flag says): R y = g by back substitution over the k steps the cycle made
static __global__ void k_gm_solve_y(int m, double* gs) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int k = (int)gs[5 * m + 4];
    const double* g = gs;
    double* y = gs + 4 * m + 3;
    const double* R0 = gs + 5 * m + 8;
    for (int i = k - 1; i >= 0; --i) {
        double s = g[i];
        for (int l = i + 1; l < k; ++l) s -= R0[(int64_t)l * (m + 1) + i] * y[l];
        const double d = R0[(int64_t)i * (m + 1) + i];
        y[i] = d != 0.0 ? s / d : 0.0;
    }
}
// x += sum_i y_i V_i
static __global__ void k_gm_update_x(int64_t n, int m, const double* V, const double* gs, double* x) {
    const int k = (int)gs[5 * m + 4];
    const double* y = gs + 4 * m + 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = x[i];
        for (int l = 0; l < k; ++l) s += y[l] * V[(int64_t)l * n + i];
        x[i] = s;
    }
}
// true residual r = bt - t (t = At x), |r|^2 partials
static __global__ __launch_bounds__(256) void k_gm_residual(int64_t n, const double* bt, const double* t, double* r, double* part2) {
    __shared__ double red[4];
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = bt[i] - t[i];
        r[i] = v, a += isfinite(v) ? v * v : 1e300;
    }
    const double s = gm_block_sum(a, red);
    if (threadIdx.x == 0) part2[blockIdx.x] = s;
}
// sc[3] = |r|^2; the stop flag now says whether the TRUE residual meets the tolerance (or the budget is spent / the cycle broke down / stagnated:
// three cycles in a row that did not bring the residual below 0.999 of the one before)
static __global__ __launch_bounds__(256) void k_gm_cycle_fin(const double* part2, int np, double* sc, int32_t* ctl, double tol2, int maxit) {
    __shared__ double red[4];
    double a = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) a += part2[i];
    const double rr = gm_block_sum(a, red);
    if (threadIdx.x != 0) return;
    const double before = sc[21];   // |r|^2 at the start of the cycle that just ended
    int32_t stalled = ctl[5];   // (a word of its own, zeroed by run_gmres: ctl[3] is the Jacobi scaling's / the single launch's)
    stalled = (rr > 0.998 * before) ? stalled + 1 : 0;
    ctl[5] = stalled;
    sc[3] = rr, sc[21] = rr;
    const bool conv = rr <= tol2 * sc[0];
    if (!conv && (stalled >= 3 || !isfinite(rr))) ctl[2] = 1;
    ctl[0] = (conv || ctl[1] >= maxit || ctl[2] != 0) ? 1 : 0;
}

}  // namespace fdapde_hip
#endif
