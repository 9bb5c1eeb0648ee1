// eng_dense.hip -- the dense inverse of a small FEM system (kernels_dense.h): built once per matrix, applied as ONE matrix-vector product per
// right-hand side.  Who uses it (one-GPU contexts, systems of up to `dense_rows` DOFs, the method left open):
//   fdapde_lin_solve        once a handle has been asked for more than `dense_after` columns ("factor once, solve many": fdaPDE/utils/symbols.h:133-160,
//                           linear_algebra/smw.h:38-59) -- b and x through pinned host memory, the host spinning on a completion word
//   fdapde_solve_parabolic  K = M / dt + A is fixed over the steps (fem_linear_parabolic_solver.h:41,56-68: one compute, one solve per step): every step
//                           is one product with M, one with K^-1, nothing returns to the host inside the loop
//   fdapde_solve            the stage of FDAPDE_SOLVER_AUTO behind BiCGStab and in front of GMRES: a direct solve of the reference's own row-zeroed matrix
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>

#include "context.h"
#include "engine.h"
#include "kernels_dense.h"

namespace fdapde_engine {

namespace {
__global__ void k_dense_rhs(int64_t n, const double* f, const double* g, const uint8_t* bnd, int use_bnd, double* rhs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rhs[i] = (use_bnd && bnd[i]) ? g[i] : f[i];
}
// one implicit Euler step's right-hand side: rhs = (M u_i) / dt + f_{i+1}, and g(., i + 1) on the Dirichlet DOFs (g handed over in the reference numbering;
// fem_linear_parabolic_solver.h:60-66)
__global__ void k_dense_step_rhs(int64_t n, const double* mu, double inv_dt, const double* f, const double* g_ext, const int32_t* i2e, const uint8_t* bnd, double* rhs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rhs[i] = (g_ext && bnd[i]) ? g_ext[i2e[i]] : mu[i] * inv_dt + f[i];
}
// ... and what follows the product: u_{i+1} becomes the next step's u_i and column i + 1 of the solution (reference numbering)
__global__ void k_dense_step_out(int64_t n, const double* u, const int32_t* i2e, double* uprev, double* sol_ext) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double v = u[i];
        uprev[i] = v, sol_ext[i2e[i]] = v;
    }
}
}   // namespace

bool dense_eligible(const fdapde_ctx* c) {
    return c->has_device && c->dense_rows > 0 && c->hs.n_dofs <= c->dense_rows && c->hs.n_dofs <= kDenseMaxRows && c->comm == nullptr && c->ar_fn == nullptr &&
           !c->halo_ready && !c->rd.ready;
}

// What an inversion costs, from the measured shape of k_dense_invert_blocked on MI355X: n / nb panels, each the longer of the panel workgroup's turn
// (~1.2 us per pivot step + ~4.5 us per 512-row block for the panel's columns out and the next panel's in + hand-offs) and the workers' update (a
// read-modify-write sweep of the n x n array over the fabric at ~4.5 TB/s): 289 rows 0.5 ms, 1 089 rows 2.6 ms, 2 116 rows 8.5 ms, 4 225 rows 19 ms, 8 100 rows 112 ms.
// Callers build an inverse when the Krylov time it replaces is of that order ("rent or buy": the handle after it has spent half of this on Krylov
// columns, the stepper when its steps will).
double dense_build_estimate_ms(int64_t n) {
    const int64_t rpt = (n + kDenseTB - 1) / kDenseTB;
    const bool multi = rpt > 4;   // several panel workgroups, a panel of 16 (dense_build): ~3.5 us per pivot step
    const int64_t nb = multi ? 16 : rpt <= 4 ? 16 : rpt <= 8 ? 8 : 4;
    const double panel_us = multi ? 3.5 * 16.0 + 12.0 : 1.2 * (double)nb + 4.0 + 4.5 * (double)rpt, update_us = 8.0 + 16.0 * (double)n * (double)n / 4.5e6;
    return 1e-3 * (double)((n + nb - 1) / nb) * std::max(panel_us, update_us) + 0.2;
}

// D.X = (the matrix A of the pattern, its Dirichlet rows replaced by unit rows if use_bnd)^-1, internal DOF order.  D.ready, or D.failed where the
// matrix is singular to working precision / the launch could not run: the caller keeps its Krylov path.
int dense_build(fdapde_ctx* c, const double* A, int use_bnd, fdapde_ctx::Dense& D) {
    const int64_t n = c->hs.n_dofs;
    hipStream_t st = c->stream;
    D.ready = false, D.failed = true, D.n = n, D.use_bnd = use_bnd, D.A = A;
    const auto t0 = std::chrono::steady_clock::now();
    DBuf<double> S;
    DBuf<int32_t> perm, status;
    DBuf<unsigned long long> cand, worst;
    int G = (int)std::max<int64_t>(1, std::min<int64_t>(c->n_cu > 0 ? c->n_cu : 64, (n + 7) / 8));   // (pivot by pivot: row i lives with workgroup i mod G)
    if (const char* e = std::getenv("FDAPDE_DENSE_ROWS_PER_WG")) G = (int)std::max<int64_t>(1, std::min<int64_t>(c->n_cu > 0 ? c->n_cu : 64, (n + std::atoi(e) - 1) / std::max(1, std::atoi(e))));   // (measurements)
    const int64_t ld = (n + 15) & ~int64_t(15);
    // pivots per panel of the blocked inversion: the panel lives in the registers of ONE workgroup (512 threads x RPT rows x nb columns, at most 64 doubles
    // per thread): 16 columns up to 2 048 rows, 8 up to 4 096, 4 up to 8 192
    // ... and above 2 048 rows SEVERAL panel workgroups of 1 536 rows x 16 columns each (a pivot step's choice agreed over the fabric: ~5 us per step instead
    // of 1.1, but a panel of 16: the update sweeps the matrix a quarter as often -- 4 225 rows 66 -> 19 ms, 5 929 rows 188 -> 48 ms, 8 100 rows 471 -> 112 ms)
    const int rpt = (int)((n + kDenseTB - 1) / kDenseTB);
    int multi_above = 4;   // row blocks of 512: above 2 048 rows (measured: 1 681 rows 4.9 ms with one panel workgroup against 7.1, 2 116 rows 8.4 - 9.3 either way, 2 601 rows 11.7 against 10.0, 3 025 rows 14.3 against 11.6, 4 096 rows 25.1 against 18.9)
    if (const char* e = std::getenv("FDAPDE_DENSE_MULTI_RPT")) multi_above = std::max(1, std::atoi(e));   // (measurements)
    const int KP = (c->dense_multi && rpt > multi_above) ? (int)((n + 3 * kDenseTB - 1) / (3 * kDenseTB)) : 1;
    int nb = KP > 1 ? 16 : rpt <= 4 ? 16 : rpt <= 8 ? 8 : 4;
    if (const char* e = std::getenv("FDAPDE_DENSE_NB")) nb = std::max(1, std::min(nb, std::atoi(e)));   // (measurements)
    const bool blocked = c->dense_block && nb >= 2;
    DBuf<double> S1;
    DBuf<int32_t> piv_row;
    DBuf<unsigned long long> flags, xch;
    DBuf<long long> stamps;
    HIPCHK(c, S.alloc((size_t)n * (size_t)ld));
    HIPCHK(c, D.X.alloc((size_t)n * n));
    HIPCHK(c, perm.alloc((size_t)n));
    HIPCHK(c, cand.alloc(2 * (size_t)G));
    HIPCHK(c, status.alloc(4));
    HIPCHK(c, worst.alloc(2));
    HIPCHK(c, hipMemsetAsync(cand.p, 0, 2 * (size_t)G * sizeof(unsigned long long), st));
    HIPCHK(c, hipMemsetAsync(status.p, 0, 4 * sizeof(int32_t), st));
    HIPCHK(c, hipMemsetAsync(worst.p, 0, 2 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_dense_fill, dim3((unsigned)n), dim3(256), 0, st, n, ld, c->rowptr.p, c->colidx.p, A, c->bnd.p, use_bnd, S.p);
    const double* result = S.p;
    if (blocked) {
        HIPCHK(c, S1.alloc((size_t)n * (size_t)ld));
        HIPCHK(c, piv_row.alloc(kDenseNB));
        // the update's grid: R x C blocks of the matrix, one workgroup each (all co-resident: <= one per CU) -- about 8 column tiles of 16 per block (one
        // per wavefront) and 4 row tiles (one batch of loads in flight)
        const int n_tiles = (int)((n + 15) / 16), cus = c->n_cu > 0 ? c->n_cu : 64;
        int C = std::max(1, std::min(17, (n_tiles + 7) / 8));
        int R = std::max(1, std::min((cus - KP) / C, (n_tiles + 3) / 4));   // (+ the panel workgroups)
        if (const char* e = std::getenv("FDAPDE_DENSE_GRID")) {   // "R,C" (measurements)
            int r_ = 0, c_ = 0;
            if (std::sscanf(e, "%d,%d", &r_, &c_) == 2 && r_ >= 1 && c_ >= 1 && r_ * c_ + KP <= cus) R = r_, C = c_;
        }
        const int RB = 16 * ((n_tiles + R - 1) / R), CB = 16 * ((n_tiles + C - 1) / C);
        R = (int)((n + RB - 1) / RB), C = (int)((n + CB - 1) / CB);   // (the blocks in use)
        G = R * C;
        HIPCHK(c, flags.alloc((size_t)G + 2));
        HIPCHK(c, hipMemsetAsync(flags.p, 0, ((size_t)G + 2) * sizeof(unsigned long long), st));
        DenseBlkArgs b{};
        b.n = (int32_t)n, b.G = G, b.ld = (int32_t)ld, b.nb = nb, b.C = C, b.RB = RB, b.CB = CB, b.S0 = S.p, b.S1 = S1.p, b.perm = perm.p, b.piv_row = piv_row.p;
        b.done = flags.p, b.ready = flags.p + G, b.status = status.p, b.timeout_ticks = 200000000ll;
        HIPCHK(c, xch.alloc(2 * (size_t)KP * 40 + (size_t)KP));
        HIPCHK(c, hipMemsetAsync(xch.p, 0, (2 * (size_t)KP * 40 + (size_t)KP) * sizeof(unsigned long long), st));
        b.KP = KP, b.xch = xch.p, b.pdone = xch.p + 2 * (size_t)KP * 40;
        if (std::getenv("FDAPDE_DENSE_STAMPS")) {   // (measurements)
            HIPCHK(c, stamps.alloc(32));
            HIPCHK(c, hipMemsetAsync(stamps.p, 0, 32 * sizeof(long long), st));
            b.stamps = stamps.p;
        }
        const size_t lds = std::max(sizeof(double) * (size_t)RB * 16 + (size_t)RB, sizeof(double) * (size_t)kDenseTB * (kDenseNB + 1)) + 64;
        const void* fn = KP > 1     ? reinterpret_cast<const void*>(&k_dense_invert_blocked<3, 16, true>)
                         : rpt <= 2 ? reinterpret_cast<const void*>(&k_dense_invert_blocked<2, 16, false>)
                         : rpt <= 3 ? reinterpret_cast<const void*>(&k_dense_invert_blocked<3, 16, false>)
                         : rpt <= 4 ? reinterpret_cast<const void*>(&k_dense_invert_blocked<4, 16, false>)
                         : rpt <= 8 ? reinterpret_cast<const void*>(&k_dense_invert_blocked<8, 8, false>)
                                    : reinterpret_cast<const void*>(&k_dense_invert_blocked<16, 4, false>);
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        void* kargs[] = {&b};
        HIPCHK(c, hipLaunchKernel(fn, dim3((unsigned)(G + KP)), dim3(kDenseTB), kargs, lds, st));
        const int n_panels = (int)((n + nb - 1) / nb);
        result = (n_panels & 1) ? S1.p : S.p;
    } else {
        DenseInvArgs a{};
        a.n = (int32_t)n, a.G = G, a.ld = (int32_t)ld, a.S = S.p, a.perm = perm.p, a.cand = cand.p, a.status = status.p;
        a.timeout_ticks = 200000000ll;   // 2 s at 100 MHz: a workgroup that is not resident (another process's kernels on the device) ends the attempt
        const size_t lds = sizeof(double) * (size_t)n;
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_invert), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_dense_invert, dim3((unsigned)G), dim3(kDenseT), lds, st, a);
    }
    hipLaunchKernelGGL(k_dense_unpermute, dim3((unsigned)n), dim3(256), 0, st, n, ld, result, perm.p, D.X.p);
    hipLaunchKernelGGL(k_dense_check, dim3((unsigned)n), dim3(256), 0, st, n, c->rowptr.p, c->colidx.p, A, c->bnd.p, use_bnd, D.X.p, worst.p);
    HIPCHK(c, hipGetLastError());
    int32_t h_status[4] = {0, 0, 0, 0};
    unsigned long long h_worst = 0;
    HIPCHK(c, hipMemcpyAsync(h_status, status.p, sizeof h_status, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&h_worst, worst.p, sizeof h_worst, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    double chk;
    std::memcpy(&chk, &h_worst, sizeof chk);
    D.check = chk;
    D.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stamps.p) {
        long long h[32];
        HIPCHK(c, hipMemcpy(h, stamps.p, sizeof h, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "dense inverse, panel 8 (us from 'panel in registers'): panel workgroup: factorised %.1f previous update done %.1f published %.1f next panel ready %.1f | worker 0: seen %.1f rows of E in LDS %.1f update issued %.1f drained %.1f\n",
                     0.01 * (h[1] - h[0]), 0.01 * (h[2] - h[0]), 0.01 * (h[3] - h[0]), 0.01 * (h[4] - h[0]), 0.01 * (h[8] - h[0]), 0.01 * (h[9] - h[0]), 0.01 * (h[10] - h[0]), 0.01 * (h[11] - h[0]));
        std::fprintf(stderr, "  pivot step 5 (us from its start): own candidate %.2f barrier %.2f pivot known %.2f pivot row read %.2f rows updated %.2f\n", 0.01 * (h[17] - h[16]), 0.01 * (h[18] - h[16]), 0.01 * (h[19] - h[16]), 0.01 * (h[20] - h[16]), 0.01 * (h[21] - h[16]));
    }
    if (std::getenv("FDAPDE_DEBUG_SETUP"))
        std::fprintf(stderr, "dense inverse: %lld rows, %d workgroups, %s (nb %d), status %d, max |I - A X| = %.3e, %.2f ms\n", (long long)n, G, blocked ? "blocked" : "pivot by pivot", nb, h_status[0], chk, D.build_ms);
    if (h_status[0] != 0 || !(chk < 1e-6)) {   // singular / timed out / an inverse too poor for one refinement step to repair
        D.X.release();
        return FDAPDE_OK;
    }
    D.refine = !(chk < 1e-13);
    D.ready = true, D.failed = false;
    return FDAPDE_OK;
}

static int ensure_dense_work(fdapde_ctx* c, size_t count) {
    if (c->dn_b.n < count) HIPCHK(c, c->dn_b.alloc(count));
    if (c->dn_x.n < count) HIPCHK(c, c->dn_x.alloc(count));
    if (c->dn_r.n < count) HIPCHK(c, c->dn_r.alloc(count));
    if (c->dn_cnt.n < 4) {
        HIPCHK(c, c->dn_cnt.alloc(4));
        HIPCHK(c, hipMemsetAsync(c->dn_cnt.p, 0, 4 * sizeof(unsigned int), c->stream));
    }
    return FDAPDE_OK;
}

static void launch_gemv(fdapde_ctx* c, const fdapde_ctx::Dense& D, int nc, const double* v, double* y, int accumulate) {
    const int64_t n = D.n;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    if (nc == 1) hipLaunchKernelGGL(k_dense_gemv<1>, grid, block, 0, c->stream, n, nc, D.X.p, v, y, accumulate);
    else if (nc <= 2) hipLaunchKernelGGL(k_dense_gemv<2>, grid, block, 0, c->stream, n, nc, D.X.p, v, y, accumulate);
    else if (nc <= 4) hipLaunchKernelGGL(k_dense_gemv<4>, grid, block, 0, c->stream, n, nc, D.X.p, v, y, accumulate);
    else if (nc <= 8) hipLaunchKernelGGL(k_dense_gemv<8>, grid, block, 0, c->stream, n, nc, D.X.p, v, y, accumulate);
    else hipLaunchKernelGGL(k_dense_gemv<16>, grid, block, 0, c->stream, n, nc, D.X.p, v, y, accumulate);   // (64 columns: the rows cross four times, not sixteen)
}

// x = X b for nc columns, device to device, internal order (x must not alias b); one step of iterative refinement where the inverse asked for it
int dense_apply(fdapde_ctx* c, fdapde_ctx::Dense& D, int nc, const double* b, double* x) {
    const int64_t n = D.n;
    launch_gemv(c, D, nc, b, x, 0);
    if (D.refine) {
        if (int rc = ensure_dense_work(c, (size_t)n * nc)) return rc;
        hipLaunchKernelGGL(k_dense_residual, dim3(g1(n * nc)), dim3(256), 0, c->stream, n, nc, c->rowptr.p, c->colidx.p, D.A, c->bnd.p, D.use_bnd, b, x, c->dn_r.p);
        launch_gemv(c, D, nc, c->dn_r.p, x, 1);
    }
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}

// nc right-hand sides from host memory (reference numbering, column-major n x nc) -> solutions in host memory: staged through the context's pinned
// block, the device reads and writes it itself, the host spins on the completion word (then falls back to waiting for the stream)
int dense_solve_host(fdapde_ctx* c, fdapde_ctx::Dense& D, const double* b_host, int nc, double* x_host) {
    const size_t n = (size_t)D.n, cnt = n * (size_t)nc;
    hipStream_t st = c->stream;
    if (c->h_io_cap < 2 * cnt + 8) {
        if (c->h_io) {
            HIPCHK(c, hipStreamSynchronize(st));
            (void)hipHostFree(c->h_io);
        }
        c->h_io = nullptr, c->h_io_cap = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_io), sizeof(double) * (2 * cnt + 8)));
        c->h_io_cap = 2 * cnt + 8;
    }
    if (int rc = ensure_dense_work(c, cnt)) return rc;
    double* hb = c->h_io;
    double* hx = c->h_io + cnt;
    volatile long long* done = reinterpret_cast<volatile long long*>(c->h_io + 2 * cnt);
    const bool one_launch = nc == 1 && !D.refine && c->dense_direct;
    const bool host_b = nc == 1 && !D.refine && n <= 512 && (c->dense_direct || c->dense_hostb);   // b permuted by the host on its way into the pinned block
    if (host_b) {
        if (int rc = ensure_host(c, kHostPerm)) return rc;
        const int32_t* i2e = c->hs.dof_i2e.data();
        for (size_t i = 0; i < n; ++i) hb[i] = b_host[i2e[i]];
    } else
        std::memcpy(hb, b_host, sizeof(double) * cnt);
    *done = 0;
    std::atomic_thread_fence(std::memory_order_release);
    // stage, product(s), hand-over: three short launches behind each other.  Knob dense_direct (off: measured slower, context.h): one column's product hands the
    // result over itself (k_dense_gemv_direct) -- up to 512 rows as the only launch (every workgroup reads b from the pinned block), above that behind k_dense_stage.
    // (Also measured and dropped: the whole solve of a 289-row system as a ONE-WORKGROUP launch -- 35 us against 23: sixteen wavefronts walk eighteen rows each,
    // latency-bound.)
    const dim3 dgrid((unsigned)((n + 3) / 4));
    // many columns (from 256 KB on): the pinned block crosses PCIe by DMA in both directions -- a kernel reading or writing host memory itself moves 6 - 10 GB/s
    // (k_dense_stage: 375 us for 64 columns of 4 225 rows), the copy engine 25+ -- and the host waits for the stream instead of spinning on a word
    const bool bulk = sizeof(double) * cnt >= (size_t(256) << 10) && !host_b && !one_launch && c->dense_bulk;
    if (bulk) {
        if (c->dn_e.n < 2 * cnt) HIPCHK(c, c->dn_e.alloc(2 * cnt));
        HIPCHK(c, hipMemcpyAsync(c->dn_e.p, hb, sizeof(double) * cnt, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_dense_stage, dim3(g1((int64_t)cnt)), dim3(256), 0, st, (int64_t)n, nc, c->dof_i2e.p, c->dn_e.p, c->dn_b.p, c->dn_cnt.p);
        if (int rc = dense_apply(c, D, nc, c->dn_b.p, c->dn_x.p)) return rc;
        hipLaunchKernelGGL(k_dense_out, dim3(g1((int64_t)cnt)), dim3(256), 0, st, (int64_t)n, nc, c->dof_e2i.p, c->dn_x.p, c->dn_e.p + cnt, nullptr, c->dn_cnt.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hx, c->dn_e.p + cnt, sizeof(double) * cnt, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        std::memcpy(x_host, hx, sizeof(double) * cnt);
        return FDAPDE_OK;
    }
    if (host_b && one_launch) {
        hipLaunchKernelGGL(k_dense_gemv_direct<true>, dgrid, dim3(256), 0, st, (int)n, D.X.p, c->dof_i2e.p, hb, hx, const_cast<long long*>(done), c->dn_cnt.p + 1);
    } else if (host_b) {   // two launches: the product reads b from the pinned block itself, then the hand-over
        hipLaunchKernelGGL(k_dense_gemv_hostb, dgrid, dim3(256), 0, st, (int)n, D.X.p, hb, c->dn_x.p, c->dn_cnt.p);
        hipLaunchKernelGGL(k_dense_out, dim3(g1((int64_t)cnt)), dim3(256), 0, st, (int64_t)n, nc, c->dof_e2i.p, c->dn_x.p, hx, const_cast<long long*>(done), c->dn_cnt.p);
    } else if (one_launch) {
        hipLaunchKernelGGL(k_dense_stage, dim3(g1((int64_t)cnt)), dim3(256), 0, st, (int64_t)n, nc, c->dof_i2e.p, hb, c->dn_b.p, c->dn_cnt.p);
        hipLaunchKernelGGL(k_dense_gemv_direct<false>, dgrid, dim3(256), 0, st, (int)n, D.X.p, c->dof_i2e.p, c->dn_b.p, hx, const_cast<long long*>(done), c->dn_cnt.p + 1);
    } else {
        hipLaunchKernelGGL(k_dense_stage, dim3(g1((int64_t)cnt)), dim3(256), 0, st, (int64_t)n, nc, c->dof_i2e.p, hb, c->dn_b.p, c->dn_cnt.p);
        if (int rc = dense_apply(c, D, nc, c->dn_b.p, c->dn_x.p)) return rc;
        hipLaunchKernelGGL(k_dense_out, dim3(g1((int64_t)cnt)), dim3(256), 0, st, (int64_t)n, nc, c->dof_e2i.p, c->dn_x.p, hx, const_cast<long long*>(done), c->dn_cnt.p);
    }
    HIPCHK(c, hipGetLastError());
    long long seen = 0;
    if (c->persist_direct_spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        while ((seen = *done) == 0) {
            if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > (double)c->persist_direct_spin_us) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (seen == 0) HIPCHK(c, hipStreamSynchronize(st));
    std::memcpy(x_host, hx, sizeof(double) * cnt);
    return FDAPDE_OK;
}

// the direct stage of the open method: u = K_z^-1 rhs_z with K_z the reference's row-zeroed matrix (fem_solver_base.h:142-155) and rhs_z = f with g on
// the Dirichlet rows (internal order, device).  *solved = false: no inverse (singular, too large, not a one-GPU context) -- the caller goes on.
int dense_direct(fdapde_ctx* c, const double* A, int use_bnd, const double* f_dev, const double* g_dev, bool* solved) {
    *solved = false;
    if (!dense_eligible(c)) return FDAPDE_OK;
    fdapde_ctx::Dense& D = c->solve_dense;
    if (int rc = dense_build(c, A, use_bnd, D)) return rc;
    if (!D.ready) return FDAPDE_OK;
    const int64_t n = D.n;
    if (int rc = ensure_dense_work(c, (size_t)n)) return rc;
    hipLaunchKernelGGL(k_dense_rhs, dim3(g1(n)), dim3(256), 0, c->stream, n, f_dev, g_dev, c->bnd.p, use_bnd, c->dn_b.p);
    if (int rc = dense_apply(c, D, 1, c->dn_b.p, c->u.p)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    D.X.release(), D.ready = false;   // (one solve: the next fdapde_init may change the matrix)
    *solved = true;
    return FDAPDE_OK;
}

// the two fused kernels around the product of one step of the parabolic stepper's dense loop
void dense_step_rhs(fdapde_ctx* c, const double* mu, double inv_dt, const double* f, const double* g_ext_dev, double* rhs) {
    const int64_t n = c->hs.n_dofs;
    hipLaunchKernelGGL(k_dense_step_rhs, dim3(g1(n)), dim3(256), 0, c->stream, n, mu, inv_dt, f, g_ext_dev, c->dof_i2e.p, c->bnd.p, rhs);
}
void dense_step_out(fdapde_ctx* c, const double* u, double* uprev, double* sol_ext_dev) {
    const int64_t n = c->hs.n_dofs;
    hipLaunchKernelGGL(k_dense_step_out, dim3(g1(n)), dim3(256), 0, c->stream, n, u, c->dof_i2e.p, uprev, sol_ext_dev);
}

// the parabolic stepper's loop with K^-1 = D.X in hand, ONE product per step (kernels_dense.h, k_dense_step): u0 = the initial condition (internal order, device),
// sol_ext = n x n_times columns in the reference numbering (device; column 0 is the caller's).  Needs an inverse that asked for no refinement.
int dense_step_loop(fdapde_ctx* c, fdapde_ctx::Dense& D, int32_t n_times, double inv_dt, const double* g_ext_dev, double* u0, double* sol_ext) {
    const int64_t n = D.n, nt = n_times - 1;
    hipStream_t st = c->stream;
    DBuf<double> B, F, Cm, u1;
    HIPCHK(c, B.alloc((size_t)n * n));
    HIPCHK(c, F.alloc((size_t)n * nt));
    HIPCHK(c, Cm.alloc((size_t)n * nt));
    HIPCHK(c, u1.alloc((size_t)n));
    hipLaunchKernelGGL(k_dense_xm, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, st, n, D.X.p, c->rowptr.p, c->colidx.p, c->vals[FDAPDE_MAT_MASS].p, c->bnd.p, D.use_bnd, inv_dt, B.p);
    hipLaunchKernelGGL(k_dense_step_cols, dim3(g1(n * nt)), dim3(256), 0, st, n, nt, c->force.p, D.use_bnd ? g_ext_dev : nullptr, c->dof_i2e.p, c->bnd.p, F.p);
    launch_gemv(c, D, (int)nt, F.p, Cm.p, 0);
    double* cur = u0;
    double* nxt = u1.p;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    for (int64_t i = 0; i < nt; ++i) {
        hipLaunchKernelGGL(k_dense_step, grid, block, 0, st, n, B.p, cur, Cm.p + (size_t)i * n, c->dof_i2e.p, nxt, sol_ext + (size_t)(i + 1) * n);
        std::swap(cur, nxt);
    }
    HIPCHK(c, hipGetLastError());
    if (cur != c->u.p) HIPCHK(c, hipMemcpyAsync(c->u.p, cur, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));   // (the last step's solution, where fdapde_solution looks)
    HIPCHK(c, hipStreamSynchronize(st));   // (the buffers above go out of scope)
    return FDAPDE_OK;
}

void preload_dense() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_dense_invert_blocked<2, 16, false>));
    (void)hipGetLastError();
}

}   // namespace fdapde_engine
