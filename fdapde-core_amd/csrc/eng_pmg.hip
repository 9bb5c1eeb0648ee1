// eng_pmg.hip -- FDAPDE_SOLVER_PMG: a TWO-LEVEL solver for order-2 spaces (flexible GMRES around a V(1,1) cycle).
//
// Why: Jacobi-preconditioned Krylov on a P2 system needs O(1 / h) iterations (C5: 5.36 M DOFs, 660 - 790 BiCGStab iterations, 1 400 operator
// applications at 0.35 ms each), where the reference's SparseLU (fem_linear_elliptic_solver.h:38-47) does not care about conditioning at all.  The P1
// space on the SAME mesh is a coarse level that comes for free: its DOFs are the mesh nodes, a P2 vertex DOF takes the vertex value, a P2 edge DOF
// the mean of its edge's two vertices (the P2 interpolant of a P1 function), and the library can assemble the same operator on it.
//
// What: right-preconditioned FLEXIBLE GMRES; the preconditioner of an iteration is a cycle
//     z = w S v ;  z += P A1^-1 P^T (v - A z) ;  z += w S (v - A z)          (S = D^-1, w = 1.5 / lambda_max(D^-1 A): damped Jacobi)
// with the coarse system solved to 1e-1 by the library's own Krylov solver (a different operator every time: hence the flexible method).  17 - 20 iterations from
// 16 k to 5.4 M DOFs in 2-D and 3-D, the same count whatever the data; C5: 64 ms against 608 ms of the Jacobi-BiCGStab stage.  The forms that came first -- BiCGStab
// around the additive M^-1 = D^-1 + P A1^-1 P^T (23 - 28 iterations of two applications, a count that moved with the last bits of the data), flexible GMRES around
// the same -- remain behind knobs (pmg_outer, pmg_smooth) for the A/B figures of DESIGN.md 4.7.
//
// How: the coarse problem lives in a CONTEXT OF ITS OWN (c->pmg.coarse: same mesh, fdapde_dofs_build(1), the same operator terms -- coefficient
// fields, sampled at the order-2 rule's quadrature nodes, as their cell means --, homogeneous Dirichlet data on the same boundary); a coarse
// solve is that context's solver, prepared once per coarse operator, run on the restricted residual.  The transfer tables are built on the device (pmg_setup).  The
// outer iteration is driven from the host (twenty iterations of ~ms: launch and read-back latency do not matter); the fine operator on the Krylov vectors is the
// blocked-ELL SpMV on A D^-1 (every vector of the cycle kept D-scaled), on the iterate itself the CSR kernel on the raw matrix with the Dirichlet rows put back as
// unit rows -- the reference's own row-zeroed system (fem_solver_base.h:142-155).  One-GPU contexts.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "context.h"
#include "engine.h"
#include "kernels_reduce.h"

namespace fdapde_engine {

// (eng_solve.hip)
void launch_spmv(fdapde_ctx* c, const double* vals, const double* x, double* y, const double* w, double* partial, const int32_t* stop, hipEvent_t e0, hipEvent_t e1,
                 int dot2_ww, const uint8_t* owned);

namespace {
using namespace fdapde_hip;

// local edge slot -> its two local vertices (csrc/tables.cpp EDGE2 / EDGE3; reference_element.h:60-62, 93-96)
const int kEdge2[3][2] = {{0, 1}, {0, 2}, {1, 2}};
const int kEdge3[6][2] = {{1, 2}, {0, 2}, {0, 1}, {1, 3}, {2, 3}, {0, 3}};

__global__ void k_pmg_diag_inv(int64_t n, const int32_t* diag, const double* A, const uint8_t* bnd, int use_bnd, double* dinv, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = 1.0;
    if (!(use_bnd && bnd[i])) d = A[diag[i]];
    if (!(d != 0.0) || !isfinite(1.0 / d)) atomicOr(flag, 1), d = 1.0;   // (no unit-diagonal form of such a matrix: the CSR kernel serves)
    dinv[i] = 1.0 / d;
}
// the blocked-ELL layout's values for the fine operator in the form A D^-1 = I + offdiag(A) D^-1 (k_spmv_blocked keeps the unit diagonal implicit): entry (i, j)
// of the full pattern times 1 / d_j; src < 0: padding
__global__ void k_pmg_fill_cols(int64_t n, const int32_t* src, const int32_t* colidx, const double* A, const double* dinv, double* out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int32_t k = src[e];
    out[e] = k >= 0 ? A[k] * dinv[colidx[k]] : 0.0;
}
// y = A x has been computed on the raw matrix: the Dirichlet rows of the reference's system are unit rows
__global__ void k_pmg_unit_rows(int64_t n, const uint8_t* bnd, const double* x, double* y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && bnd[i]) y[i] = x[i];
}
// x0 = g on the Dirichlet rows, 0 elsewhere
__global__ void k_pmg_start(int64_t n, const uint8_t* bnd, int use_bnd, const double* g, double* x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = (use_bnd && bnd[i]) ? g[i] : 0.0;
}
// r = rhs - K x with rhs = f on the free rows and g on the Dirichlet rows (y = K x given)
__global__ void k_pmg_residual(int64_t n, const uint8_t* bnd, int use_bnd, const double* f, const double* g, const double* y, double* r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = ((use_bnd && bnd[i]) ? g[i] : f[i]) - y[i];
}
// coarse load = P^T v (rows of P^T: the vertex DOF itself + half of every edge DOF at the vertex), 0 on the coarse Dirichlet rows.  Sixteen lanes per coarse
// row (a vertex has ~15 entries in 3-D): the index / weight reads are contiguous, the sum is a DPP reduction -- a thread per row walked its ~15 scattered
// gathers one after the other: 342 us at C5's 681 k coarse rows
__global__ __launch_bounds__(256) void k_pmg_restrict(int64_t n1, const int32_t* ptr, const int32_t* idx, const double* w, const uint8_t* bnd1, const double* v, double* out) {
    const int lane16 = threadIdx.x & 15;
    const int64_t a = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    double s = 0.0;
    if (a < n1 && !(bnd1 && bnd1[a]))
        for (int32_t k = ptr[a] + lane16; k < ptr[a + 1]; k += 16) s += w[k] * v[idx[k]];
    s = team_sum<16>(s);
    if (a < n1 && lane16 == 0) out[a] = s;
}
// out = D^-1 v + P e on the free rows, v on the Dirichlet rows (where v is 0 throughout the iteration)
__global__ void k_pmg_apply(int64_t n2, const int32_t* pa, const int32_t* pb, const uint8_t* bnd2, int use_bnd, const double* dinv, const double* v, const double* e,
                            int by_d, double wv, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    if (use_bnd && bnd2[i]) {
        out[i] = wv * v[i];
        return;
    }
    const int32_t a = pa[i], b = pb[i];
    const double corr = b < 0 ? e[a] : 0.5 * (e[a] + e[b]);
    out[i] = by_d ? wv * v[i] + corr / dinv[i] : wv * dinv[i] * v[i] + corr;   // by_d: D M^-1 v, what the A D^-1 form of the fine operator takes (the x update divides again)
}
// the smoothed cycle's vector steps (all in D-scaled variables: z' = D z)
__global__ void k_pmg_post(int64_t n, const double* v, const double* t, double om, double* z, double* r) {   // r = v - t (t = A D^-1 z'), z' += om r
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ri = v[i] - t[i];
    r[i] = ri, z[i] += om * ri;
}
__global__ void k_pmg_wfin(int64_t n, const double* v, const double* r, const double* t, double om, double* w) {   // w = A z = (v - r) + om t (t = A D^-1 r)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = v[i] - r[i] + om * t[i];
}
__global__ void k_pmg_mulv(int64_t n, const double* a, const double* b, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}
// a fixed pseudo-random start vector for the power iteration (0 on the Dirichlet rows)
__global__ void k_pmg_hashvec(int64_t n, const uint8_t* bnd, int use_bnd, double* x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull, z = (z ^ (z >> 27)) * 0x94D049BB133111EBull, z ^= z >> 31;
    x[i] = (use_bnd && bnd[i]) ? 0.0 : (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
// up to three dot products in one pass, per-workgroup partials in a fixed order (summed by k_pmg_reduce: the same bits every run)
__global__ __launch_bounds__(256) void k_pmg_dots(int64_t n, const double* a0, const double* b0, const double* a1, const double* b1, const double* a2, const double* b2,
                                                  double* part) {
    __shared__ double red[3][4];
    double s0 = 0, s1 = 0, s2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        s0 += a0[i] * b0[i];
        if (a1) s1 += a1[i] * b1[i];
        if (a2) s2 += a2[i] * b2[i];
    }
    s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[0][w] = s0, red[1][w] = s1, red[2][w] = s2;
    __syncthreads();
    if (threadIdx.x < 3) part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
__global__ __launch_bounds__(256) void k_pmg_reduce(const double* part, int np, double* out) {
    __shared__ double red[5];
    for (int k = 0; k < 3; ++k) {
        double s = 0;
        for (int i = threadIdx.x; i < np; i += 256) s += part[(size_t)k * np + i];
        s = block_sum(s, red);
        if (threadIdx.x == 0) out[k] = s;
        __syncthreads();
    }
}
__global__ void k_pmg_p(int64_t n, const double* r, const double* v, double beta, double omega, double* p) {   // p = r + beta (p - omega v)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = r[i] + beta * (p[i] - omega * v[i]);
}
__global__ void k_pmg_lin(int64_t n, const double* a, double alpha, const double* b, double* out) {   // out = a - alpha b
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - alpha * b[i];
}
__global__ void k_pmg_x(int64_t n, double alpha, const double* ph, double omega, const double* sh, const double* dinv, double* x) {   // x += alpha ph + omega sh
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (dinv: ph, sh are D M^-1 p, D M^-1 s)
    if (i < n) x[i] += (alpha * ph[i] + (sh ? omega * sh[i] : 0.0)) * (dinv ? dinv[i] : 1.0);
}
// ---- flexible GMRES (the outer method): Gram-Schmidt against the basis V_0 .. V_{k-1} (vectors `stride` apart), twice per new vector ----
// h_q = V_q . w for q < k, and w . w behind them: per-workgroup partials part[q * np + block] (fixed order: the same bits every run); eight basis vectors per
// sweep over the workgroup's elements (w is read again per sweep: one extra vector per eight)
__global__ __launch_bounds__(256) void k_pmg_mdot(int64_t n, const double* V, int64_t stride, int k, const double* w, double* part) {
    __shared__ double red[9][4];
    const int wv = threadIdx.x >> 6;
    for (int c0 = 0; c0 < k || c0 == 0; c0 += 8) {
        double sq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sw = 0;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const double wi = w[i];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (c0 + q < k) sq[q] += V[(int64_t)(c0 + q) * stride + i] * wi;
            if (c0 == 0) sw += wi * wi;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double r = wave_sum(sq[q]);
            if ((threadIdx.x & 63) == 0) red[q][wv] = r;
        }
        sw = wave_sum(sw);
        if ((threadIdx.x & 63) == 0) red[8][wv] = sw;
        __syncthreads();
        if (threadIdx.x < 8 && c0 + (int)threadIdx.x < k)
            part[(size_t)(c0 + threadIdx.x) * gridDim.x + blockIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (threadIdx.x == 8 && c0 == 0) part[(size_t)k * gridDim.x + blockIdx.x] = red[8][0] + red[8][1] + red[8][2] + red[8][3];
        __syncthreads();
    }
}
// out[q] = sum of part[q * np ..] for q < cnt: one workgroup per value
__global__ __launch_bounds__(256) void k_pmg_mreduce(const double* part, int np, double* out) {
    __shared__ double red[5];
    double s = 0;
    for (int i = threadIdx.x; i < np; i += 256) s += part[(size_t)blockIdx.x * np + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
// w -= sum_q h_q V_q, and the partials of what is left: |w|^2 to part[block]
__global__ __launch_bounds__(256) void k_pmg_msub(int64_t n, const double* V, int64_t stride, int k, const double* h, double* w, double* part) {
    __shared__ double hs[64];
    __shared__ double red[4];
    if ((int)threadIdx.x < k) hs[threadIdx.x] = h[threadIdx.x];
    __syncthreads();
    double sw = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double wi = w[i];
        for (int q = 0; q < k; ++q) wi -= hs[q] * V[(int64_t)q * stride + i];
        w[i] = wi, sw += wi * wi;
    }
    sw = wave_sum(sw);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sw;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_pmg_scale(int64_t n, const double* a, double f, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f * a[i];
}
// x += (sum_q y_q Z_q) [* dinv: the Z_q are D M^-1 v_q]
__global__ __launch_bounds__(256) void k_pmg_comb(int64_t n, const double* Z, int64_t stride, int k, const double* y, const double* dinv, double* x) {
    __shared__ double ys[64];
    if ((int)threadIdx.x < k) ys[threadIdx.x] = y[threadIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0;
    for (int q = 0; q < k; ++q) s += ys[q] * Z[(int64_t)q * stride + i];
    x[i] += dinv ? s * dinv[i] : s;
}
// a coefficient field of the fine level -- nq2 samples per cell at the order-2 rule's nodes, `width` values each -- for the coarse level: every cell's
// weighted mean at each of the nq1 nodes of the P1 rule (a preconditioner needs the coarse OPERATOR only approximately)
__global__ void k_pmg_cell_mean(int64_t n_cells, int nq2, int nq1, int width, const double* qw2, const int32_t* fine_cell, const double* in, double* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_cells * width) return;
    const int64_t cell = t / width, fc = fine_cell[cell];   // (the two contexts number the cells of the mesh their own way)
    const int k = (int)(t - cell * width);
    double s = 0.0;
    for (int q = 0; q < nq2; ++q) s += qw2[q] * in[(fc * nq2 + q) * width + k];
    for (int q = 0; q < nq1; ++q) out[(cell * nq1 + q) * width + k] = s;
}
inline unsigned g1n(int64_t n) { return (unsigned)((n + 255) / 256); }
}   // namespace

void pmg_release(fdapde_ctx* c) {
    fdapde_ctx::Pmg& m = c->pmg;
    if (m.coarse) fdapde_ctx_destroy(m.coarse);
    m.coarse = nullptr, m.ready = false, m.init_seen = -1, m.omega = 0.0, m.omega_A = nullptr, m.fine_A = nullptr;
    m.fine_cell.release(), m.pa.release(), m.pb.release(), m.rt_ptr.release(), m.rt_idx.release(), m.rt_w.release(), m.dinv.release(), m.vec.release(), m.part.release(), m.dots.release(), m.basis.release();
}

bool pmg_eligible(const fdapde_ctx* c) {
    if (!c->has_device || !c->dev_ready || c->hs.order != 2 || c->comm != nullptr || c->ar_fn != nullptr || c->halo_ready || c->rd.ready || c->group) return false;
    for (const HostTerm& t : c->op)
        if (t.t.space_varying && !t.data_dev) return false;   // (a coefficient field that is not on the device)
    return !c->op.empty();
}

// ---- the transfer tables, built on the device (the host loops they replace took 0.67 s of a 0.9 s first call at C5's size: 40 M scattered accesses) ----
struct PmgEdges {
    int a[6], b[6];
};
__global__ void k_pmg_inv_perm(int64_t n, const int32_t* i2e, int32_t* e2i) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) e2i[i2e[i]] = (int32_t)i;
}
__global__ void k_pmg_fine_cell(int64_t n, const int32_t* i2e1, const int32_t* e2i2, int32_t* fine_cell) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fine_cell[i] = e2i2[i2e1[i]];
}
// one thread per coarse cell: local DOFs 0 .. M of a P2 cell are its vertices in the cell's vertex order -- the P1 cell's local DOFs --, local DOF M + 1 + k sits
// on the edge of the local vertices (ed.a[k], ed.b[k]).  pass 0: pa / pb and the vertices' boundary flags (every cell of a DOF writes the same values); pass 1: a
// constrained edge DOF constrains both of its end nodes on the coarse level
__global__ void k_pmg_transfer(int64_t n_cells, int nv, int nb2, PmgEdges ed, const int32_t* fine_cell, const int32_t* cd1, const int32_t* cd2, const uint8_t* bnd2,
                               int32_t* pa, int32_t* pb, uint8_t* bnd1, int pass) {
    const int64_t c1 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c1 >= n_cells) return;
    const int32_t* d1 = cd1 + c1 * nv;
    const int32_t* d2 = cd2 + (int64_t)fine_cell[c1] * nb2;
    if (pass == 0) {
        for (int k = 0; k < nv; ++k) pa[d2[k]] = d1[k], bnd1[d1[k]] = bnd2[d2[k]];
        for (int k = nv; k < nb2; ++k) {   // (the cells sharing an edge see it in either direction: the pair goes in as (smaller, larger) -- the same words from all)
            const int32_t x = d1[ed.a[k - nv]], y = d1[ed.b[k - nv]];
            pa[d2[k]] = x < y ? x : y, pb[d2[k]] = x < y ? y : x;
        }
    } else {
        for (int k = nv; k < nb2; ++k)
            if (bnd2[d2[k]]) bnd1[d1[ed.a[k - nv]]] = 1, bnd1[d1[ed.b[k - nv]]] = 1;
    }
}
// (coarse DOF << 32 | fine DOF) of every entry of P, the slot of a vertex DOF's missing second entry sorts behind everything; entries per coarse DOF counted
__global__ void k_pmg_keys(int64_t n2, const int32_t* pa, const int32_t* pb, unsigned long long* keys, int32_t* count, int32_t* bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const int32_t a = pa[i], b = pb[i];
    if (a < 0) {
        atomicOr(bad, 1);
        keys[2 * i] = keys[2 * i + 1] = ~0ull;
        return;
    }
    keys[2 * i] = ((unsigned long long)(uint32_t)a << 32) | (uint32_t)i, atomicAdd(&count[a], 1);
    if (b >= 0) keys[2 * i + 1] = ((unsigned long long)(uint32_t)b << 32) | (uint32_t)i, atomicAdd(&count[b], 1);
    else keys[2 * i + 1] = ~0ull;
}
__global__ void k_pmg_rt_fill(int64_t n, const unsigned long long* keys, const int32_t* pb, int32_t* idx, double* w) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int32_t i = (int32_t)(keys[e] & 0xFFFFFFFFull);
    idx[e] = i, w[e] = pb[i] >= 0 ? 0.5 : 1.0;
}
__global__ void k_pmg_to_reference(int64_t n, const int32_t* i2e, const uint8_t* in, uint8_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i2e[i]] = in[i];
}

// the same tables by the host loops of the first version (knob pmg_setup_check: both are built and compared)
static int pmg_tables_host(fdapde_ctx* c, fdapde_ctx* cc, std::vector<int32_t>& fine_cell, std::vector<int32_t>& pa, std::vector<int32_t>& pb, std::vector<uint8_t>& bnd1,
                           std::vector<int32_t>& ptr, std::vector<int32_t>& idx, std::vector<double>& w) {
    if (int rc = ensure_host(c, kHostDofs | kHostPerm)) return rc;
    if (int rc = ensure_host(cc, kHostDofs | kHostPerm)) return rc;
    const HostSpace &h2 = c->hs, &h1 = cc->hs;
    const int nv = h2.M + 1, nb2 = h2.nb;
    const int64_t n2 = h2.n_dofs, n1 = h1.n_dofs;
    fine_cell.assign((size_t)h1.n_cells, 0);
    std::vector<int32_t> e2i2((size_t)h2.n_cells);
    for (int64_t ci = 0; ci < h2.n_cells; ++ci) e2i2[(size_t)h2.cell_i2e[(size_t)ci]] = (int32_t)ci;
    for (int64_t ci = 0; ci < h1.n_cells; ++ci) fine_cell[(size_t)ci] = e2i2[(size_t)h1.cell_i2e[(size_t)ci]];
    pa.assign((size_t)n2, -1), pb.assign((size_t)n2, -1), bnd1.assign((size_t)n1, 0);
    for (int pass = 0; pass < 2; ++pass)
        for (int64_t e = 0; e < h2.n_cells; ++e) {
            const int32_t* d2 = &h2.dofs[(size_t)e * nb2];
            const int32_t* d1 = &h1.dofs[(size_t)e * nv];
            if (pass == 0)
                for (int k = 0; k < nv; ++k) pa[(size_t)h2.dof_e2i[(size_t)d2[k]]] = h1.dof_e2i[(size_t)d1[k]], bnd1[(size_t)d1[k]] = h2.dof_bnd[(size_t)d2[k]];
            for (int k = nv; k < nb2; ++k) {
                const int* ed = h2.M == 2 ? kEdge2[k - nv] : kEdge3[k - nv];
                if (pass == 0) {
                    const size_t fi = (size_t)h2.dof_e2i[(size_t)d2[k]];
                    const int32_t x = h1.dof_e2i[(size_t)d1[ed[0]]], y = h1.dof_e2i[(size_t)d1[ed[1]]];
                    pa[fi] = std::min(x, y), pb[fi] = std::max(x, y);
                } else if (h2.dof_bnd[(size_t)d2[k]])
                    bnd1[(size_t)d1[ed[0]]] = 1, bnd1[(size_t)d1[ed[1]]] = 1;
            }
        }
    ptr.assign((size_t)n1 + 1, 0);
    for (int64_t i = 0; i < n2; ++i) {
        if (pa[(size_t)i] < 0) return fail(c, FDAPDE_EHIP, "a P2 DOF that no cell's table names");
        ++ptr[(size_t)pa[(size_t)i] + 1];
        if (pb[(size_t)i] >= 0) ++ptr[(size_t)pb[(size_t)i] + 1];
    }
    for (int64_t a = 0; a < n1; ++a) ptr[(size_t)a + 1] += ptr[(size_t)a];
    idx.assign((size_t)ptr[(size_t)n1], 0), w.assign((size_t)ptr[(size_t)n1], 0.0);
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < n2; ++i) {
        const bool edge = pb[(size_t)i] >= 0;
        int32_t& fa = fill[(size_t)pa[(size_t)i]];
        idx[(size_t)fa] = (int32_t)i, w[(size_t)fa] = edge ? 0.5 : 1.0, ++fa;
        if (edge) {
            int32_t& fb = fill[(size_t)pb[(size_t)i]];
            idx[(size_t)fb] = (int32_t)i, w[(size_t)fb] = 0.5, ++fb;
        }
    }
    return FDAPDE_OK;
}

// the coarse context and the transfer operators, once per function space
static int pmg_setup(fdapde_ctx* c) {
    fdapde_ctx::Pmg& m = c->pmg;
    if (m.ready) return FDAPDE_OK;
    pmg_release(c);
    const auto t0 = std::chrono::steady_clock::now();
    DebugClock clk;
    const HostSpace& h2 = c->hs;
    fdapde_ctx* cc = nullptr;
    if (int rc = fdapde_ctx_create(c->device, &cc)) return fail(c, rc, "FDAPDE_SOLVER_PMG: the coarse context could not be created");
    m.coarse = cc;
    auto bail = [&](int rc) {
        c->err = "FDAPDE_SOLVER_PMG (coarse level): " + cc->err;
        pmg_release(c);
        return rc;
    };
    if (int rc = host_set_mesh(cc->hs, h2.M, h2.N, h2.n_nodes, h2.nodes.data(), h2.n_cells, h2.cells.data(), h2.node_bnd.data(), cc->err)) return bail(rc);
    clk.mark("pmg_setup: coarse context + mesh");
    if (int rc = e_dofs_build(cc, 1, nullptr)) return bail(rc);
    clk.mark("pmg_setup: coarse dofs_build");
    const HostSpace& h1 = cc->hs;
    const int nv = h2.M + 1, nb2 = h2.nb;
    const int64_t n2 = h2.n_dofs, n1 = h1.n_dofs, ncell = h2.n_cells;
    if (h1.nb != nv || h1.n_cells != h2.n_cells) return bail(fail(cc, FDAPDE_EHIP, "the P1 space of the mesh does not match the P2 space's cells"));
    if (c->cdofs.n < (size_t)ncell * nb2 || cc->cdofs.n < (size_t)ncell * nv || c->cell_i2e.n < (size_t)ncell || cc->cell_i2e.n < (size_t)ncell || cc->dof_i2e.n < (size_t)n1)
        return bail(fail(cc, FDAPDE_EHIP, "the DOF tables of the two spaces are not on the device"));
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIPCHK(c, hipStreamSynchronize(cc->stream));   // (the coarse space's tables were built on its own stream)
    const dim3 bv(256);
    // coarse internal cell -> fine internal cell (the two contexts number the cells of the mesh their own way)
    DBuf<int32_t> e2i2, count;
    DBuf<uint8_t> bnd1, bnd1_e;
    DBuf<unsigned long long> keys, keys_sorted;
    DBuf<int32_t> bad;
    DBuf<char> tmp;
    HIPCHK(c, e2i2.alloc((size_t)ncell));
    HIPCHK(c, m.fine_cell.alloc((size_t)ncell));
    hipLaunchKernelGGL(k_pmg_inv_perm, dim3(g1n(ncell)), bv, 0, st, ncell, c->cell_i2e.p, e2i2.p);
    hipLaunchKernelGGL(k_pmg_fine_cell, dim3(g1n(ncell)), bv, 0, st, ncell, cc->cell_i2e.p, e2i2.p, m.fine_cell.p);
    // fine DOF -> its one (vertex DOF) or two (edge DOF) coarse DOFs, the coarse boundary mask: the fine one, as set or as built, decides -- and the coarse space
    // must stay INSIDE the fine one: a constrained edge DOF constrains both of its end nodes on the coarse level.  With the mask the reference builds from a full
    // set of boundary nodes that is already so; with a partial node mask in 2-D it is not -- there every edge DOF of a geometric boundary edge is constrained
    // whatever its end nodes are (triangulation.h:150-193 / fe_space DOF marking), and a coarse function that does not vanish at those nodes prolongs to a zig-zag
    // the coarse operator takes for a smooth mode (100 - 300 outer iterations on such masks; tools/fuzz_pmg.py)
    PmgEdges ed{};
    for (int k = 0; k < nb2 - nv; ++k) ed.a[k] = h2.M == 2 ? kEdge2[k][0] : kEdge3[k][0], ed.b[k] = h2.M == 2 ? kEdge2[k][1] : kEdge3[k][1];
    HIPCHK(c, m.pa.alloc((size_t)n2));
    HIPCHK(c, m.pb.alloc((size_t)n2));
    HIPCHK(c, bnd1.alloc((size_t)n1));
    HIPCHK(c, bnd1_e.alloc((size_t)n1));
    HIPCHK(c, hipMemsetAsync(m.pa.p, 0xFF, sizeof(int32_t) * (size_t)n2, st));
    HIPCHK(c, hipMemsetAsync(m.pb.p, 0xFF, sizeof(int32_t) * (size_t)n2, st));
    HIPCHK(c, hipMemsetAsync(bnd1.p, 0, (size_t)n1, st));
    for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(k_pmg_transfer, dim3(g1n(ncell)), bv, 0, st, ncell, nv, nb2, ed, m.fine_cell.p, cc->cdofs.p, c->cdofs.p, c->bnd.p, m.pa.p, m.pb.p, bnd1.p, pass);
    hipLaunchKernelGGL(k_pmg_to_reference, dim3(g1n(n1)), bv, 0, st, n1, cc->dof_i2e.p, bnd1.p, bnd1_e.p);
    std::vector<uint8_t> bnd1_host((size_t)n1);
    HIPCHK(c, hipMemcpyAsync(bnd1_host.data(), bnd1_e.p, (size_t)n1, hipMemcpyDeviceToHost, st));
    // P^T as CSR over the coarse DOFs, a row's entries by ascending fine DOF (a fixed order: the restriction sums the same way every run): sort the entries'
    // (coarse, fine) keys
    HIPCHK(c, keys.alloc(2 * (size_t)n2));
    HIPCHK(c, keys_sorted.alloc(2 * (size_t)n2));
    HIPCHK(c, count.alloc((size_t)n1 + 1));
    HIPCHK(c, bad.alloc(1));
    HIPCHK(c, m.rt_ptr.alloc((size_t)n1 + 1));
    HIPCHK(c, hipMemsetAsync(count.p, 0, sizeof(int32_t) * ((size_t)n1 + 1), st));
    HIPCHK(c, hipMemsetAsync(bad.p, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_pmg_keys, dim3(g1n(n2)), bv, 0, st, n2, m.pa.p, m.pb.p, keys.p, count.p, bad.p);
    size_t need_sort = 0, need_scan = 0;
    const int end_bit = 64;   // (all of them: the sentinels must end up last)
    HIPCHK(c, hipcub::DeviceRadixSort::SortKeys(nullptr, need_sort, keys.p, keys_sorted.p, (int)(2 * n2), 0, end_bit, st));
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, need_scan, count.p, m.rt_ptr.p, (int)(n1 + 1), st));
    HIPCHK(c, tmp.alloc(std::max(need_sort, need_scan)));
    HIPCHK(c, hipcub::DeviceRadixSort::SortKeys(tmp.p, need_sort, keys.p, keys_sorted.p, (int)(2 * n2), 0, end_bit, st));
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, need_scan, count.p, m.rt_ptr.p, (int)(n1 + 1), st));
    int32_t total = 0, bad_h = 0;
    HIPCHK(c, hipMemcpyAsync(&total, m.rt_ptr.p + n1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&bad_h, bad.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (bad_h) return bail(fail(cc, FDAPDE_EHIP, "a P2 DOF that no cell's table names"));
    HIPCHK(c, m.rt_idx.alloc((size_t)total));
    HIPCHK(c, m.rt_w.alloc((size_t)total));
    hipLaunchKernelGGL(k_pmg_rt_fill, dim3(g1n(total)), bv, 0, st, (int64_t)total, keys_sorted.p, m.pb.p, m.rt_idx.p, m.rt_w.p);
    HIPCHK(c, hipGetLastError());
    clk.mark("pmg_setup: transfer tables (device)");
    if (int rc = e_dofs_set_boundary(cc, bnd1_host.data())) return bail(rc);
    clk.mark("pmg_setup: coarse boundary mask");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, m.dinv.alloc((size_t)n2));
    HIPCHK(c, m.vec.alloc(9 * (size_t)n2));
    m.np = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n2 + 4095) / 4096));
    HIPCHK(c, m.part.alloc(3 * (size_t)m.np));
    HIPCHK(c, m.dots.alloc(4));
    HIPCHK(c, hipStreamSynchronize(st));
    if (c->pmg_setup_check) {   // the host loops of the first version build the same tables: compared entry by entry
        std::vector<int32_t> fc, pa, pb, ptr, idx;
        std::vector<uint8_t> b1;
        std::vector<double> w;
        if (int rc = pmg_tables_host(c, cc, fc, pa, pb, b1, ptr, idx, w)) return bail(rc);
        auto same_i = [&](const DBuf<int32_t>& d, const std::vector<int32_t>& hv) {
            std::vector<int32_t> g(hv.size());
            if (hipMemcpy(g.data(), d.p, sizeof(int32_t) * hv.size(), hipMemcpyDeviceToHost) != hipSuccess) return false;
            return g == hv;
        };
        std::vector<double> gw(w.size());
        std::string which;
        if ((size_t)total != idx.size()) which += " entries";
        if (!same_i(m.fine_cell, fc)) which += " fine_cell";
        if (!same_i(m.pa, pa)) which += " pa";
        if (!same_i(m.pb, pb)) which += " pb";
        if (!same_i(m.rt_ptr, ptr)) which += " rt_ptr";
        if ((size_t)total == idx.size() && !same_i(m.rt_idx, idx)) which += " rt_idx";
        if ((size_t)total == idx.size() && !(hipMemcpy(gw.data(), m.rt_w.p, sizeof(double) * w.size(), hipMemcpyDeviceToHost) == hipSuccess && gw == w)) which += " rt_w";
        if (b1 != bnd1_host) which += " boundary";
        if (!which.empty()) {
            cc->err = "pmg_setup_check: the device-built transfer tables differ from the host-built ones:" + which;
            return bail(FDAPDE_EHIP);
        }
    }
    std::vector<double> zeros((size_t)n1, 0.0);
    if (int rc = e_set_dirichlet(cc, zeros.data())) return bail(rc);
    clk.mark("pmg_setup: allocations + coarse Dirichlet data");
    m.ready = true, m.init_seen = -1;
    m.setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("FDAPDE_DEBUG_SETUP"))
        std::fprintf(stderr, "pmg: coarse level %lld DOFs under %lld, set-up %.1f ms\n", (long long)n1, (long long)n2, m.setup_ms);
    return FDAPDE_OK;
}

// The two-level solve of K u = rhs (K: `A` with the Dirichlet rows as unit rows if use_bnd; rhs = f_dev on the free rows, g_dev on the Dirichlet rows),
// started from x0_dev (or from g on the Dirichlet rows and 0 elsewhere); the coarse operator is the context's operator terms on the P1 space plus
// `extra_reaction` times the mass matrix (the stepper's M / dt), assembled again when `coarse_key` differs from the one it was assembled for.  Result in c->u,
// outcome in c->info (method_used, iters, converged, relres = the TRUE relative residual).  FDAPDE_OK / FDAPDE_ENOCONV / an error.
int pmg_run(fdapde_ctx* c, const double* A, const double* f_dev, const double* g_dev, int use_bnd, const double* x0_dev, double extra_reaction, int64_t coarse_key,
            double rtol, int maxit) {
    if (int rc = pmg_setup(c)) return rc;
    fdapde_ctx::Pmg& m = c->pmg;
    fdapde_ctx* cc = m.coarse;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int64_t n2 = c->hs.n_dofs, n1 = cc->hs.n_dofs;
    const auto t_begin = std::chrono::steady_clock::now();
    // the coarse level is constrained where the fine one is -- and only if it is (a context with a boundary mask but no Dirichlet data solves the natural problem)
    if ((cc->have_g ? 1 : 0) != (use_bnd ? 1 : 0)) {
        std::vector<double> zeros((size_t)n1, 0.0);
        if (int rc = e_set_dirichlet(cc, use_bnd ? zeros.data() : nullptr)) {
            c->err = "FDAPDE_SOLVER_PMG (coarse level): " + cc->err;
            return rc;
        }
        m.init_seen = -1;
    }
    // the coarse operator: the same terms on the P1 space, assembled again whenever the fine one has been
    if (m.init_seen != coarse_key || m.extra_seen != extra_reaction) {
        cc->op.clear(), cc->op_symmetric = c->op_symmetric, cc->coef_of_op = false;
        for (const HostTerm& ft : c->op) {
            HostTerm ct;
            ct.t = ft.t, ct.field_nonsym = ft.field_nonsym;
            if (ft.t.space_varying) {   // a coefficient field: its cell means at the P1 rule's nodes
                const int width = ft.t.kind == FDAPDE_DIFFUSION ? c->hs.N * c->hs.N : ft.t.kind == FDAPDE_ADVECTION ? c->hs.N : 1;
                BasisTables bt;
                if (int rc = build_basis_tables(c->hs.M, 2, &bt)) return rc;
                DBuf<double> qw;
                HIPCHK(c, qw.upload(bt.qw, (size_t)bt.nq, st));
                ct.data_dev = std::make_shared<DBuf<double>>();
                HIPCHK(c, ct.data_dev->alloc((size_t)cc->hs.nq * (size_t)cc->hs.n_cells * width));
                hipLaunchKernelGGL(k_pmg_cell_mean, dim3(g1n(c->hs.n_cells * width)), dim3(256), 0, st, c->hs.n_cells, c->hs.nq, cc->hs.nq, width, qw.p, m.fine_cell.p,
                                   ft.data_dev->p, ct.data_dev->p);
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipStreamSynchronize(st));   // (qw goes out of scope; the coarse context reads the field on its own stream)
            }
            cc->op.push_back(std::move(ct));
        }
        if (extra_reaction != 0.0) {
            HostTerm rt{};
            rt.t.kind = FDAPDE_REACTION, rt.t.space_varying = 0, rt.t.coef = 1.0, rt.t.cst[0] = extra_reaction;
            cc->op.push_back(rt);
        }
        if (int rc = e_init(cc, nullptr)) {
            c->err = "FDAPDE_SOLVER_PMG (coarse level): " + cc->err;
            return rc;
        }
        if (int rc = coarse_prepare(cc, &m.coarse_ss)) {
            c->err = "FDAPDE_SOLVER_PMG (coarse level): " + cc->err;
            return rc;
        }
        m.init_seen = coarse_key, m.extra_seen = extra_reaction;
    }
    fdapde_options inner{};
    inner.method = FDAPDE_SOLVER_AUTO, inner.rtol = c->pmg_inner_rtol, inner.maxit = c->pmg_inner_maxit, inner.assembly = FDAPDE_ASSEMBLY_ROWS;
    double *x = m.vec.p, *r = x + n2, *r0 = r + n2, *p = r0 + n2, *v = p + n2, *s = v + n2, *t = s + n2, *ph = t + n2, *sh = ph + n2;
    const dim3 gv(g1n(n2)), bv(256);
    HIPCHK(c, m.flag.alloc(1));
    HIPCHK(c, hipMemsetAsync(m.flag.p, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_pmg_diag_inv, gv, bv, 0, st, n2, c->diag.p, A, c->bnd.p, use_bnd, m.dinv.p, m.flag.p);
    // The fine operator on DIRECTION vectors (p^, s^: their Dirichlet entries are exactly 0) through the blocked-ELL SpMV the multi-launch Krylov stages use for
    // long rows (k_spmv_blocked: x staged in LDS once per block, the matrix streamed at the HBM rate -- C5: 328 us against 592 us of the CSR kernel on the raw
    // matrix).  That kernel keeps a unit diagonal implicit, so the layout is filled with A D^-1: right preconditioning is A M^-1 = (A D^-1)(D M^-1), and
    // D M^-1 v = v + D P A1^-1 P^T v comes out of k_pmg_apply for free -- the residuals stay those of the unscaled system.  The three applications to the
    // iterate itself (start, warm start, true residual: Dirichlet columns count there) keep the CSR kernel.
    const int fv = use_bnd ? 1 : 0;
    bool fb = false;
    if (c->pmg_blocked && c->blocked && c->spmv_variant == 2) {
        if (int rc = build_blocked(c, fv)) return rc;
        if (c->bk[fv].ok) {
            int32_t zero_diag = 0;
            HIPCHK(c, hipMemcpyAsync(&zero_diag, m.flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            fb = zero_diag == 0;
        }
        if (fb && !(c->scaled_owner == fdapde_ctx::kScaledPmg && m.fine_A == A && m.fine_key == coarse_key && m.fine_bnd == fv)) {
            const fdapde_ctx::Blocked& bk = c->bk[fv];
            if (bk.meta.n_entries > 0)
                hipLaunchKernelGGL(k_pmg_fill_cols, dim3(g1n(bk.meta.n_entries)), bv, 0, st, bk.meta.n_entries, bk.ell_src.p, c->colidx.p, A, m.dinv.p, bk.ell_val.p);
            HIPCHK(c, hipGetLastError());
            c->bk[0].filled = c->bk[1].filled = false, c->bk_cur = -1;   // (not the scaled matrix of the Krylov stages: they fill it again)
            c->scaled_owner = fdapde_ctx::kScaledPmg, m.fine_A = A, m.fine_key = coarse_key, m.fine_bnd = fv;
        }
    }
    const double* x_dinv = fb ? m.dinv.p : nullptr;
    if (x0_dev) {   // a warm start (the stepper's previous column): its Dirichlet rows take this system's data
        HIPCHK(c, hipMemcpyAsync(x, x0_dev, sizeof(double) * (size_t)n2, hipMemcpyDeviceToDevice, st));
        if (use_bnd) hipLaunchKernelGGL(k_pmg_unit_rows, gv, bv, 0, st, n2, c->bnd.p, g_dev, x);
    } else
        hipLaunchKernelGGL(k_pmg_start, gv, bv, 0, st, n2, c->bnd.p, use_bnd, g_dev, x);
    auto apply_K = [&](const double* in, double* out) {
        launch_spmv(c, A, in, out, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
        if (use_bnd) hipLaunchKernelGGL(k_pmg_unit_rows, gv, bv, 0, st, n2, c->bnd.p, in, out);
    };
    auto apply_K_dir = [&](const double* in, double* out) {   // in: D M^-1 of a direction vector (fb) or M^-1 of it
        if (fb) launch_spmv_blocked(c, fv, in, out, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
        else apply_K(in, out);
    };
    int coarse_iters = 0, coarse_calls = 0, coarse_multi = 0, coarse_fail = 0;   // (coarse_fail: coarse solves in a row that got nowhere)
    // out = wv * (D^-1) vin + P A1^-1 P^T in (by_d: the same times D)
    auto coarse_and_apply = [&](const double* in, const double* vin, double wv, int by_d, double* out) -> int {
        // (the fine stream first: the coarse context has a stream of its own)
        hipLaunchKernelGGL(k_pmg_restrict, dim3(g1n(16 * n1)), bv, 0, st, n1, m.rt_ptr.p, m.rt_idx.p, m.rt_w.p, use_bnd ? cc->bnd.p : (const uint8_t*)nullptr, in, cc->force.p);
        HIPCHK(c, hipStreamSynchronize(st));
        fdapde_info ii{};
        const int rc = coarse_solve(cc, &m.coarse_ss, inner.rtol, inner.maxit, &ii);
        if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) {   // (an inner solve that stopped at its budget still is a correction)
            c->err = "FDAPDE_SOLVER_PMG (coarse level): " + cc->err;
            return rc;
        }
        coarse_iters += ii.iters, ++coarse_calls, coarse_multi += ii.persistent ? 0 : 1;
        // (a solve that stopped at its budget but got somewhere is a correction; how far it has to get depends on who uses it: BiCGStab assumes ONE preconditioner,
        //  the flexible GMRES takes whatever reduces the residual at all)
        coarse_fail = (ii.converged || (std::isfinite(ii.relres) && ii.relres < (c->pmg_outer == 0 ? 0.95 : 0.5))) ? 0 : coarse_fail + 1;
        HIPCHK(c, hipStreamSynchronize(cc->stream));
        hipLaunchKernelGGL(k_pmg_apply, gv, bv, 0, st, n2, m.pa.p, m.pb.p, c->bnd.p, use_bnd, m.dinv.p, vin, cc->u.p, by_d, wv, out);
        return FDAPDE_OK;
    };
    auto apply_Minv = [&](const double* in, double* out) -> int { return coarse_and_apply(in, in, 1.0, fb ? 1 : 0, out); };
    double h[3] = {0, 0, 0};
    auto dots = [&](const double* a0, const double* b0, const double* a1, const double* b1, const double* a2, const double* b2) -> int {
        hipLaunchKernelGGL(k_pmg_dots, dim3((unsigned)m.np), bv, 0, st, n2, a0, b0, a1, b1, a2, b2, m.part.p);
        hipLaunchKernelGGL(k_pmg_reduce, dim3(1), bv, 0, st, m.part.p, m.np, m.dots.p);
        HIPCHK(c, hipMemcpyAsync(h, m.dots.p, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        return FDAPDE_OK;
    };
    // r = rhs - K x0 (the lift of the Dirichlet data), shadow residual r0 = r
    apply_K(x, v);
    hipLaunchKernelGGL(k_pmg_residual, gv, bv, 0, st, n2, c->bnd.p, use_bnd, f_dev, g_dev, v, r);
    HIPCHK(c, hipMemcpyAsync(r0, r, sizeof(double) * (size_t)n2, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipMemsetAsync(p, 0, sizeof(double) * (size_t)n2, st));
    HIPCHK(c, hipMemsetAsync(v, 0, sizeof(double) * (size_t)n2, st));
    if (int rc = dots(r, r, nullptr, nullptr, nullptr, nullptr)) return rc;
    double bb = h[0];
    const double rr_start = h[0];
    if (x0_dev) {   // the stop rule stays relative to the right-hand side (lifted), not to what a good warm start leaves of it
        hipLaunchKernelGGL(k_pmg_start, gv, bv, 0, st, n2, c->bnd.p, use_bnd, g_dev, s);
        apply_K(s, t);
        hipLaunchKernelGGL(k_pmg_residual, gv, bv, 0, st, n2, c->bnd.p, use_bnd, f_dev, g_dev, t, s);
        if (int rc = dots(s, s, nullptr, nullptr, nullptr, nullptr)) return rc;
        bb = h[0];
        HIPCHK(c, hipMemsetAsync(v, 0, sizeof(double) * (size_t)n2, st));
    }
    double rr = rr_start, rho = 1.0, alpha = 1.0, omega = 1.0;
    int it = 0, fine_apps = 1;
    bool converged = rr_start <= rtol * rtol * bb, broke = false;
    // The outer method: FLEXIBLE GMRES (right-preconditioned: A Z_k = V_{k+1} H_k with Z_j = M^-1 v_j kept, so M^-1 may be a different operator every time
    // it is applied -- and it is: the coarse systems are solved to a loose tolerance by a Krylov method).  One M^-1 and one operator application per
    // iteration, the residual norm from the Givens recurrence; 36 - 37 iterations on C5's operator whatever the mesh and whether the coarse solves stop at
    // 1e-1, 1e-2 or are exact (tools/c5_fgmres_proto.py), where BiCGStab -- which assumes ONE preconditioner -- takes 50 - 58 applications at 1e-2, 64 - 90
    // at 1e-1, and its count moves by +-5 with the last bits of the data (tools/pmg_spread_probe.py).  Gram-Schmidt twice per vector (the Krylov basis of a
    // solve to 1e-10 is ill-conditioned by then); restart after `mk` vectors (memory: 2 mk + 1 vectors).
    const bool fgmres = c->pmg_outer == 0;
    int mk = 0;
    if (fgmres) {
        mk = (int)std::min<int64_t>(std::max(2, std::min(c->pmg_restart, 50)), std::max<int64_t>(5, (int64_t)(16e9 / (16.0 * (double)n2))));   // (at most ~16 GB of basis)
        mk = std::min(mk, std::max(maxit, 1));
        if (m.basis.n < (size_t)(2 * mk + 1) * (size_t)n2) {
            m.basis.release();
            if (m.basis.alloc((size_t)(2 * mk + 1) * (size_t)n2) != hipSuccess) {
                (void)hipGetLastError();
                m.basis.release();
                mk = 0;   // (no room for a basis: BiCGStab below)
            }
        }
        if (mk > 0 && (m.part.n < (size_t)(mk + 2) * (size_t)m.np || m.dots.n < (size_t)(mk + 4))) {
            HIPCHK(c, m.part.alloc((size_t)(mk + 2) * (size_t)m.np));
            HIPCHK(c, m.dots.alloc((size_t)(mk + 4)));
        }
    }
    if (fgmres && mk > 0) {
        // The preconditioner of an iteration: a V(1,1) CYCLE -- damped Jacobi, coarse correction, damped Jacobi (17 - 18 iterations on C5's operator where the
        // additive form D^-1 + P A1^-1 P^T takes 36; tools/c5_fgmres_variants_proto.py) -- at three fine operator applications instead of one, which is the
        // cheap part: half the coarse solves, a quarter of the Gram-Schmidt traffic.  Damping 1.5 / lambda_max(D^-1 A), lambda_max by 15 power iterations
        // once per matrix (undamped Jacobi is no smoother on an order-2 space: lambda_max > 2).  Everything in D-scaled variables z' = D z, so that every
        // operator application is the blocked-ELL kernel's A D^-1 (without that layout: D^-1 as a pass of its own in front of the CSR kernel).
        const bool smooth = c->pmg_smooth != 0;
        const bool primed = fb || smooth;
        double* tb = p;    // (BiCGStab's vectors are free here)
        double* rb = s;
        auto KD = [&](const double* in, double* out) {   // out = K D^-1 in
            if (fb) launch_spmv_blocked(c, fv, in, out, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
            else {
                hipLaunchKernelGGL(k_pmg_mulv, gv, bv, 0, st, n2, m.dinv.p, in, ph);
                apply_K(ph, out);
            }
        };
        double om = 0.0;
        if (smooth) {
            if (!(m.omega > 0.0 && m.omega_A == A && m.omega_key == coarse_key && m.omega_bnd == fv && m.omega_extra == extra_reaction)) {
                hipLaunchKernelGGL(k_pmg_hashvec, gv, bv, 0, st, n2, c->bnd.p, use_bnd, tb);
                double lam = 0.0;
                for (int pi = 0; pi < 15; ++pi) {
                    KD(tb, rb);
                    if (int rc = dots(tb, tb, rb, rb, nullptr, nullptr)) return rc;
                    if (!(h[0] > 0.0) || !std::isfinite(h[1])) break;
                    lam = std::sqrt(h[1] / h[0]);
                    if (std::getenv("FDAPDE_DEBUG_PMG_POWER")) std::fprintf(stderr, "pmg: power iteration %d: %.4f\n", pi, lam);
                    if (!(h[1] > 0.0)) break;
                    hipLaunchKernelGGL(k_pmg_scale, gv, bv, 0, st, n2, rb, 1.0 / std::sqrt(h[1]), tb);
                }
                m.omega = lam > 0.0 && std::isfinite(lam) ? 1.5 / lam : 0.0;
                m.omega_A = A, m.omega_key = coarse_key, m.omega_bnd = fv, m.omega_extra = extra_reaction;
                if (std::getenv("FDAPDE_DEBUG_SETUP")) std::fprintf(stderr, "pmg: lambda_max(D^-1 A) ~ %.3f, damping %.3f\n", lam, m.omega);
            }
            om = m.omega;
        }
        const bool cycle = smooth && om > 0.0;
        const double* comb_dinv = (fb || cycle) ? m.dinv.p : nullptr;
        (void)primed;
        double* V = m.basis.p;
        double* Z = V + (size_t)(mk + 1) * (size_t)n2;
        std::vector<double> H((size_t)(mk + 1) * mk, 0.0), cs((size_t)mk), sn((size_t)mk), gg((size_t)mk + 1), hj((size_t)mk + 4), yy((size_t)mk);
        auto Hat = [&](int i, int j) -> double& { return H[(size_t)j * (mk + 1) + i]; };
        const unsigned gnp = (unsigned)m.np;
        while (!converged && !broke && it < maxit) {
            // (r holds the residual of x, rr its square)
            const double beta = std::sqrt(rr);
            if (!(beta > 0.0) || !std::isfinite(beta)) {
                broke = !std::isfinite(beta);
                converged = !broke;
                break;
            }
            hipLaunchKernelGGL(k_pmg_scale, gv, bv, 0, st, n2, r, 1.0 / beta, V);
            std::fill(gg.begin(), gg.end(), 0.0);
            gg[0] = beta;
            int k = 0;   // columns of this cycle
            bool cycle_done = false;
            while (!cycle_done && k < mk && it < maxit) {
                const int j = k;
                double* zj = Z + (size_t)j * (size_t)n2;
                double* w = V + (size_t)(j + 1) * (size_t)n2;
                const double* vj = V + (size_t)j * (size_t)n2;
                if (cycle) {
                    KD(vj, tb);                                                                     // A z1, z1' = om v
                    hipLaunchKernelGGL(k_pmg_lin, gv, bv, 0, st, n2, vj, om, tb, rb);               // r1 = v - om A D^-1 v
                    if (int rc = coarse_and_apply(rb, vj, om, 1, zj)) return rc;                    // z2' = om v + D P A1^-1 P^T r1
                } else if (int rc = apply_Minv(vj, zj))
                    return rc;
                if (coarse_fail >= 4) {
                    broke = true;
                    break;
                }
                if (cycle) {
                    KD(zj, tb);
                    hipLaunchKernelGGL(k_pmg_post, gv, bv, 0, st, n2, vj, tb, om, zj, rb);          // r2 = v - A z2, z3' = z2' + om r2
                    KD(rb, tb);
                    hipLaunchKernelGGL(k_pmg_wfin, gv, bv, 0, st, n2, vj, rb, tb, om, w);           // w = A z3 = (v - r2) + om A D^-1 r2
                    fine_apps += 3;
                } else {
                    apply_K_dir(zj, w);
                    ++fine_apps;
                }
                // Gram-Schmidt, twice: h = V^T w, w -= V h, then the same on what is left (its coefficients add to h)
                double wnorm2 = 0;
                for (int pass = 0; pass < 2; ++pass) {
                    hipLaunchKernelGGL(k_pmg_mdot, dim3(gnp), bv, 0, st, n2, V, n2, j + 1, w, m.part.p);
                    hipLaunchKernelGGL(k_pmg_mreduce, dim3((unsigned)(j + 2)), bv, 0, st, m.part.p, m.np, m.dots.p);
                    hipLaunchKernelGGL(k_pmg_msub, dim3(gnp), bv, 0, st, n2, V, n2, j + 1, m.dots.p, w, m.part.p);
                    hipLaunchKernelGGL(k_pmg_mreduce, dim3(1), bv, 0, st, m.part.p, m.np, m.dots.p + (j + 2));
                    HIPCHK(c, hipMemcpyAsync(hj.data(), m.dots.p, sizeof(double) * (size_t)(j + 3), hipMemcpyDeviceToHost, st));
                    HIPCHK(c, hipStreamSynchronize(st));
                    for (int i = 0; i <= j; ++i) Hat(i, j) = pass == 0 ? hj[(size_t)i] : Hat(i, j) + hj[(size_t)i];
                    wnorm2 = hj[(size_t)j + 2];
                }
                const double hn = std::sqrt(wnorm2);
                if (!std::isfinite(hn)) {
                    broke = true;
                    break;
                }
                Hat(j + 1, j) = hn;
                for (int i = 0; i < j; ++i) {
                    const double a0 = Hat(i, j), a1 = Hat(i + 1, j);
                    Hat(i, j) = cs[(size_t)i] * a0 + sn[(size_t)i] * a1, Hat(i + 1, j) = -sn[(size_t)i] * a0 + cs[(size_t)i] * a1;
                }
                const double d = std::hypot(Hat(j, j), Hat(j + 1, j));
                if (!(d > 0.0)) {   // (a zero column: M^-1 v_j = 0 -- nothing this basis can do)
                    broke = true;
                    break;
                }
                cs[(size_t)j] = Hat(j, j) / d, sn[(size_t)j] = Hat(j + 1, j) / d;
                Hat(j, j) = d, Hat(j + 1, j) = 0.0;
                gg[(size_t)j + 1] = -sn[(size_t)j] * gg[(size_t)j], gg[(size_t)j] = cs[(size_t)j] * gg[(size_t)j];
                ++k, ++it;
                rr = gg[(size_t)k] * gg[(size_t)k];
                if (rr <= rtol * rtol * bb || !(hn > 1e-300)) cycle_done = true;   // (hn = 0: the exact solution lies in this basis)
                else hipLaunchKernelGGL(k_pmg_scale, gv, bv, 0, st, n2, w, 1.0 / hn, w);
            }
            if (k > 0) {   // x += Z y, H y = g (upper triangular after the rotations)
                for (int i = k - 1; i >= 0; --i) {
                    double sacc = gg[(size_t)i];
                    for (int q = i + 1; q < k; ++q) sacc -= Hat(i, q) * yy[(size_t)q];
                    yy[(size_t)i] = sacc / Hat(i, i);
                }
                HIPCHK(c, hipMemcpyAsync(m.dots.p, yy.data(), sizeof(double) * (size_t)k, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_pmg_comb, gv, bv, 0, st, n2, Z, n2, k, m.dots.p, comb_dinv, x);
                HIPCHK(c, hipStreamSynchronize(st));   // (yy is the host's)
            }
            if (broke) break;
            // the residual of the new iterate, computed (it starts the next cycle, and the recurrence's word is not taken for convergence)
            apply_K(x, v);
            hipLaunchKernelGGL(k_pmg_residual, gv, bv, 0, st, n2, c->bnd.p, use_bnd, f_dev, g_dev, v, r);
            if (int rc = dots(r, r, nullptr, nullptr, nullptr, nullptr)) return rc;
            rr = h[0];
            if (!std::isfinite(rr)) {
                broke = true;
                break;
            }
            converged = rr <= rtol * rtol * bb;
            if (!converged && k == 0) {
                broke = true;
                break;
            }
        }
    } else
    while (!converged && it < maxit) {
        if (int rc = dots(r0, r, nullptr, nullptr, nullptr, nullptr)) return rc;
        const double rho_new = h[0];
        if (rho_new == 0.0 || !std::isfinite(rho_new)) {
            broke = true;
            break;
        }
        const double beta = it == 0 ? 0.0 : (rho_new / rho) * (alpha / omega);
        hipLaunchKernelGGL(k_pmg_p, gv, bv, 0, st, n2, r, v, beta, omega, p);
        if (int rc = apply_Minv(p, ph)) return rc;
        if (coarse_fail >= 4) {   // a coarse operator its own solver cannot handle (strongly indefinite, singular): no preconditioner -- the caller's other stages
            broke = true;
            break;
        }
        apply_K_dir(ph, v);
        ++fine_apps;
        if (int rc = dots(r0, v, nullptr, nullptr, nullptr, nullptr)) return rc;
        if (h[0] == 0.0 || !std::isfinite(h[0])) {
            broke = true;
            break;
        }
        alpha = rho_new / h[0];
        hipLaunchKernelGGL(k_pmg_lin, gv, bv, 0, st, n2, r, alpha, v, s);
        if (int rc = dots(s, s, nullptr, nullptr, nullptr, nullptr)) return rc;
        if (h[0] <= rtol * rtol * bb) {   // (half a step is enough)
            hipLaunchKernelGGL(k_pmg_x, gv, bv, 0, st, n2, alpha, ph, 0.0, (const double*)nullptr, x_dinv, x);
            rr = h[0], ++it, converged = true;
            break;
        }
        if (int rc = apply_Minv(s, sh)) return rc;
        apply_K_dir(sh, t);
        ++fine_apps;
        if (int rc = dots(t, s, t, t, nullptr, nullptr)) return rc;
        if (h[1] == 0.0 || !std::isfinite(h[0]) || !std::isfinite(h[1])) {
            broke = true;
            break;
        }
        omega = h[0] / h[1];
        hipLaunchKernelGGL(k_pmg_x, gv, bv, 0, st, n2, alpha, ph, omega, sh, x_dinv, x);
        hipLaunchKernelGGL(k_pmg_lin, gv, bv, 0, st, n2, s, omega, t, r);
        if (int rc = dots(r, r, nullptr, nullptr, nullptr, nullptr)) return rc;
        rr = h[0], rho = rho_new, ++it;
        if (!std::isfinite(rr) || omega == 0.0) {
            broke = true;
            break;
        }
        converged = rr <= rtol * rtol * bb;
    }
    // the TRUE residual of what is handed out
    apply_K(x, v);
    hipLaunchKernelGGL(k_pmg_residual, gv, bv, 0, st, n2, c->bnd.p, use_bnd, f_dev, g_dev, v, t);
    if (int rc = dots(t, t, nullptr, nullptr, nullptr, nullptr)) return rc;
    const double true_rel = bb > 0 ? std::sqrt(h[0] / bb) : 0.0;
    if (converged && !(true_rel <= 10.0 * rtol)) converged = false;   // (a recurrence that drifted from the truth is not a solution)
    HIPCHK(c, c->u.alloc((size_t)n2));
    HIPCHK(c, hipMemcpyAsync(c->u.p, x, sizeof(double) * (size_t)n2, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const double t_asm = c->info.t_assemble_ms;   // (fdapde_init's figure stays with the record)
    c->info = fdapde_info{};
    c->info.t_assemble_ms = t_asm;
    c->info.method_used = FDAPDE_SOLVER_PMG, c->info.iters = it, c->info.converged = converged ? 1 : 0, c->info.relres = true_rel;
    c->info.t_solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    c->info.persistent = 0;
    if (std::getenv("FDAPDE_DEBUG_SETUP"))
        std::fprintf(stderr, "pmg: %d outer iterations, %d fine applications, %d coarse solves with %d iterations (%d of them not as one launch), true relres %.2e, %.2f ms\n", it,
                     fine_apps, coarse_calls, coarse_iters, coarse_multi, true_rel, c->info.t_solve_ms);
    c->pmg.last_coarse_iters = coarse_iters, c->pmg.last_coarse_calls = coarse_calls;
    if (!converged) {
        c->err = coarse_fail >= 4 ? "FDAPDE_SOLVER_PMG: the coarse level's solves do not converge" : broke ? "FDAPDE_SOLVER_PMG: BiCGStab broke down" : "maxit reached";
        return FDAPDE_ENOCONV;
    }
    return FDAPDE_OK;
}

// fdapde_solve with FDAPDE_SOLVER_PMG (fem_linear_elliptic_solver.h:38-47: the system is the reference's, the way to its solution is not)
int e_solve_pmg(fdapde_ctx* c, const fdapde_options* opt, fdapde_info* info) {
    if (!pmg_eligible(c))
        return fail(c, FDAPDE_EUNSUPPORTED, "FDAPDE_SOLVER_PMG takes one-GPU contexts and order-2 spaces");
    const double rtol = (opt && opt->rtol > 0) ? opt->rtol : 1e-10;
    const int maxit = (opt && opt->maxit > 0) ? opt->maxit : 400;
    const int rc = pmg_run(c, c->vals[FDAPDE_MAT_STIFF].p, c->force.p, c->g.p, c->have_g ? 1 : 0, nullptr, 0.0, 2 * c->init_count, rtol, maxit);
    if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
    c->solved = true, c->dirichlet_applied = c->have_g;   // (scaled_owner: kScaledPmg while the blocked layout holds this matrix, pmg_run)
    if (info) *info = c->info;
    return rc;
}

}   // namespace fdapde_engine
