// kernels.h -- hand-written HIP kernels for gfx950 (MI355X): FEM operator / forcing assembly and the Krylov solve.
//
// Everything here is HBM/L2-bandwidth-bound fp64 + int32 index work; there is no dense contraction, so no MFMA.
// What matters (cdna_hip_programming.md guidelines 2, 3, 11-13; MI355X_MICROARCH.md "Global float atomics"):
//   * index streams (adjacency, slots, colidx) and value streams are read with unit stride across the 64 lanes of a
//     wavefront; gathers (vertex coordinates, x[col]) go through the XCD's L2, which the locality numbering built in
//     host_setup.cpp keeps hot;
//   * the default assembly is atomic-free: one lane owns one matrix row, visits the cells around its DOF, accumulates
//     in LDS and the workgroup writes its rows once with coalesced stores (float atomics run at ~1.3 TB/s of added
//     bytes chip-wide and ~17x slower when the 64 lanes hit 64 different rows -- an element-wise scatter is exactly
//     that shape);
//   * reductions are deterministic: per-workgroup partials in a fixed slot, re-reduced in a fixed order by the
//     consumer kernel -- no float atomics in the solve.
#ifndef FDAPDE_KERNELS_H
#define FDAPDE_KERNELS_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "internal.h"

namespace fdapde_hip {

// ---------------------------------------------------------------------------------------------------------------
// device-side operator description (kernel argument, lives in SGPRs / scalar cache)
// ---------------------------------------------------------------------------------------------------------------
struct DevTerm {
    int32_t kind, space_varying;
    double coef;
    double cst[9];
    const double* data;   // device pointer, rows in INTERNAL cell order: (nq*cell_i + q) x (N*N | N | 1)
};
struct DevOp {
    int32_t n;
    int32_t needs_psi;    // any advection / reaction leaf
    int32_t needs_rows;   // any space-varying leaf (needs the global cell id for its data row)
    DevTerm t[kMaxTerms];
    // constant-coefficient operators (OPK 3): the leaves summed once on the host
    //   form = -(g_i . Kt g_j) + psi_i (g_j . bt) + ct psi_i psi_j,  Kt = sum coef K (Laplacian: coef I), bt = sum coef b, ct = sum coef c
    double kt[9], bt[3], ct;
    int32_t tab_sym;      // Kt symmetric and no advection: evaluate in the bitwise-symmetric order
};

// quadrature + basis tables as they sit in device memory (copied to LDS by every workgroup that integrates)
struct DevTables {
    double qw[kMaxQuad];
    double psi[kMaxBasis * kMaxQuad];        // [i*nq + q]
    double dpsi[kMaxBasis * kMaxQuad * 3];   // [(i*nq + q)*3 + k]
    double qn[kMaxQuad * 3];                 // [q*M + k]
    double mtab[kMaxBasis * kMaxBasis];      // sum_q w_q psi_i(p_q) psi_j(p_q), [i*nb + j]: reference mass integrals
    double wsum;                             // sum_q w_q (0.999999999999999 for the 3-point rule: part of the contract)
    double pad_;
};
constexpr int kTablesDoubles = sizeof(DevTables) / sizeof(double);
// reference tensors of the constant-coefficient form (OPK 3), staged in LDS behind DevTables by that instantiation only:
//   ktab[(k*3 + l)*NB*NB + i*NB + j] = sum_q w_q d_k psi_i(p_q) d_l psi_j(p_q)      ctab[l*NB*NB + i*NB + j] = sum_q w_q psi_i d_l psi_j
struct DevRefTensors {
    double ktab[9 * kMaxBasis * kMaxBasis];
    double ctab[3 * kMaxBasis * kMaxBasis];
};
constexpr int kRefDoubles = sizeof(DevRefTensors) / sizeof(double);

struct AsmArgs {
    int64_t n_dofs, n_cells;
    const int32_t* cverts;     // n_cells x (M+1), internal node ids
    const int32_t* cdofs;      // n_cells x nb, internal DOF ids
    const double* vcoords;     // internal node id -> NP doubles
    const int64_t* sl_off;     // adjacency slices
    const int32_t* adj;
    const uint32_t* slotw;
    const int32_t* rowptr;
    const int32_t* colidx;
    const DevTables* tables;
    const DevRefTensors* reftab;   // OPK 3 only
    double* vals;              // CSR values (internal slots) or nullptr
    const double* fq;          // forcing at quadrature nodes, internal cell order, or nullptr
    double* force;             // forcing vector (internal DOF order) or nullptr
    int32_t lds_acc_cap;       // doubles available for the row accumulators
    // block-local tables of the row-owner kernel (host_setup.cpp): cells visited by the block's rows, their vertex nodes
    const int64_t* bc_off;
    const int32_t* bc_cell;
    const uint16_t* bc_vert;   // 4 per block-cell
    const int64_t* bn_off;
    const int32_t* bn_node;
    int32_t lds_nodes;         // coordinate slots reserved in LDS (max nodes of any block)
};

template <int M> struct Geo {
    double invJ[M][M];   // J^{-1}
    double measure;      // |det J| / M!
};

// Simplex::initialize (fdaPDE/geometry/simplex.h:184-195): J col j = x_{j+1} - x_0, invJ, measure = |det J| / M!
// p0..p3: vertex coordinates (global memory or the workgroup's LDS copy)
template <int M>
__device__ __forceinline__ void geo_from_vertices(const double* p0, const double* p1, const double* p2, const double* p3, Geo<M>& g) {
    if constexpr (M == 2) {
        const double j00 = p1[0] - p0[0], j01 = p2[0] - p0[0], j10 = p1[1] - p0[1], j11 = p2[1] - p0[1];
        const double det = j00 * j11 - j01 * j10;
        const double id = 1.0 / det;
        g.invJ[0][0] = j11 * id, g.invJ[0][1] = -j01 * id;
        g.invJ[1][0] = -j10 * id, g.invJ[1][1] = j00 * id;
        g.measure = fabs(det) * 0.5;
    } else {
        const double a00 = p1[0] - p0[0], a01 = p2[0] - p0[0], a02 = p3[0] - p0[0];
        const double a10 = p1[1] - p0[1], a11 = p2[1] - p0[1], a12 = p3[1] - p0[1];
        const double a20 = p1[2] - p0[2], a21 = p2[2] - p0[2], a22 = p3[2] - p0[2];
        const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
        const double det = a00 * c00 + a01 * c01 + a02 * c02;
        const double id = 1.0 / det;
        g.invJ[0][0] = c00 * id, g.invJ[1][0] = c01 * id, g.invJ[2][0] = c02 * id;
        g.invJ[0][1] = (a02 * a21 - a01 * a22) * id;
        g.invJ[1][1] = (a00 * a22 - a02 * a20) * id;
        g.invJ[2][1] = (a01 * a20 - a00 * a21) * id;
        g.invJ[0][2] = (a01 * a12 - a02 * a11) * id;
        g.invJ[1][2] = (a02 * a10 - a00 * a12) * id;
        g.invJ[2][2] = (a00 * a11 - a01 * a10) * id;
        g.measure = fabs(det) * (1.0 / 6.0);
    }
}
template <int M> __device__ __forceinline__ void cell_geometry(const AsmArgs& a, int cell, Geo<M>& g) {
    if constexpr (M == 2) {
        const int32_t* cv = a.cverts + (int64_t)cell * 3;
        const double2 x0 = *reinterpret_cast<const double2*>(a.vcoords + (int64_t)cv[0] * 2);
        const double2 x1 = *reinterpret_cast<const double2*>(a.vcoords + (int64_t)cv[1] * 2);
        const double2 x2 = *reinterpret_cast<const double2*>(a.vcoords + (int64_t)cv[2] * 2);
        geo_from_vertices<2>(&x0.x, &x1.x, &x2.x, nullptr, g);
    } else {
        const int4 cv = *reinterpret_cast<const int4*>(a.cverts + (int64_t)cell * 4);
        const double4 x0 = *reinterpret_cast<const double4*>(a.vcoords + (int64_t)cv.x * 4);
        const double4 x1 = *reinterpret_cast<const double4*>(a.vcoords + (int64_t)cv.y * 4);
        const double4 x2 = *reinterpret_cast<const double4*>(a.vcoords + (int64_t)cv.z * 4);
        const double4 x3 = *reinterpret_cast<const double4*>(a.vcoords + (int64_t)cv.w * 4);
        geo_from_vertices<3>(&x0.x, &x1.x, &x2.x, &x3.x, g);
    }
}

// physical gradient J^{-T} grad_ref: out[r] = sum_k invJ[k][r] * d[k]   (buff_invJ = invJ^T, fem_assembler.h:81)
template <int M> __device__ __forceinline__ void phys_grad(const Geo<M>& g, const double* d, double* out) {
#pragma unroll
    for (int r = 0; r < M; ++r) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) v += g.invJ[k][r] * d[k];
        out[r] = v;
    }
}

// integrand of the weak form at one quadrature node: left-to-right sum of scaled leaves
//   laplacian.h:43  -(g_i . g_j)      diffusion.h:54  -(g_i . K g_j)
//   advection.h:55  psi_i (g_j . b)   reaction.h:52   c psi_i psi_j      dt.h:34-36  0
template <int M>
__device__ __forceinline__ double weak_form(const DevOp& op, int64_t qrow, double psi_i, double psi_j, const double* gi,
                                            const double* gj) {
    double total = 0;
    for (int t = 0; t < op.n; ++t) {
        const DevTerm& T = op.t[t];
        double v = 0;
        if (T.kind == FDAPDE_LAPLACIAN) {
            double d = 0;
#pragma unroll
            for (int k = 0; k < M; ++k) d += gi[k] * gj[k];
            v = -d;
        } else if (T.kind == FDAPDE_DIFFUSION) {
            double K[M * M];
#pragma unroll
            for (int k = 0; k < M * M; ++k) K[k] = T.space_varying ? T.data[qrow * (M * M) + k] : T.cst[k];
            double d = 0;
#pragma unroll
            for (int r = 0; r < M; ++r) {
                double kg = 0;
#pragma unroll
                for (int c = 0; c < M; ++c) kg += K[r * M + c] * gj[c];
                d += gi[r] * kg;
            }
            v = -d;
        } else if (T.kind == FDAPDE_ADVECTION) {
            double d = 0;
#pragma unroll
            for (int k = 0; k < M; ++k) d += gj[k] * (T.space_varying ? T.data[qrow * M + k] : T.cst[k]);
            v = psi_i * d;
        } else if (T.kind == FDAPDE_REACTION) {
            const double c = T.space_varying ? T.data[qrow] : T.cst[0];
            v = c * psi_i * psi_j;
        }
        total = t == 0 ? T.coef * v : total + T.coef * v;
    }
    return total;
}

// One row of one element matrix: for local test function `il` of `cell`, emit(j, value) for every local trial
// function j, value = measure * sum_q w_q * form(psi_il, psi_j)(p_q)   (integrator.h:92-106), and return the forcing
// contribution measure * sum_q f_q psi_il(p_q) w_q (integrator.h:73-90) when fq is given.
// `tb` points at the LDS copy of the tables.
// OPK selects a specialised integrand (same numbers up to rounding, far fewer instructions -- the assembly kernels are
// instruction-issue bound, not bandwidth bound):
//   0  generic: any sum of leaves, evaluated per quadrature node as the reference does
//   1  a single Laplacian leaf: the term loop and its branches fold away; for P1 the gradients are constant over the cell
//      and come straight from J^{-1} (grad lambda_0 = -sum_k row_k, grad lambda_k = row_k), no table reads
//   2  a single constant reaction leaf (mass matrix): value = c * measure * sum_q w_q psi_i psi_j, the reference integrals
//      sum_q w_q psi_i psi_j do not depend on the cell and are tabulated (DevTables::mtab)
//   3  any sum of CONSTANT-coefficient leaves: on an affine cell the element matrix is a contraction of cell constants with
//      reference tensors that do not depend on the cell (DevRefTensors), summed over the same quadrature nodes as the reference:
//        A_ij = |e| ( -sum_kl Gp[k][l] ktab[k][l][i][j] + sum_l beta[l] ctab[l][i][j] + ct mtab[i][j] ),
//        Gp = J^-1 Kt J^-T,  beta = J^-1 bt.     13 multiply-adds per entry in 3-D instead of a loop over the quadrature nodes
//      (C5, 3-D P2 advection-diffusion-reaction: 98 ms -> see DESIGN.md).  Symmetric operators are evaluated in an order that
//      gives bitwise A_ij == A_ji.
template <int M, int R, int OPK, typename Emit>
__device__ __forceinline__ double element_row(const AsmArgs& a, const DevOp& op, const DevTables* tb, const Geo<M>& g, int cell,
                                              int il, bool want_matrix, Emit&& emit, const DevRefTensors* rt = nullptr) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    constexpr int NQ = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 5);
    const int64_t qrow0 = (int64_t)NQ * cell;
    double fsum = 0;
    if (a.fq != nullptr) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) fsum += (a.fq[qrow0 + q] * tb->psi[il * NQ + q]) * tb->qw[q];
        fsum *= g.measure;
    }
    if (!want_matrix) return fsum;
    if constexpr (OPK == 2) {
        const double cm = op.t[0].coef * op.t[0].cst[0] * g.measure;
#pragma unroll
        for (int j = 0; j < NB; ++j) emit(j, cm * tb->mtab[il * NB + j]);
        return fsum;
    } else if constexpr (OPK == 3) {
        constexpr int NN = NB * NB;
        double Gp[M][M], beta[M];
        const bool sym = op.tab_sym != 0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            double kr[M];   // row k of J^-1 Kt
#pragma unroll
            for (int c = 0; c < M; ++c) {
                double v = 0;
#pragma unroll
                for (int r = 0; r < M; ++r) v += g.invJ[k][r] * op.kt[r * M + c];
                kr[c] = v;
            }
#pragma unroll
            for (int l = 0; l < M; ++l) {
                double v = 0;
#pragma unroll
                for (int c = 0; c < M; ++c) v += kr[c] * g.invJ[l][c];
                Gp[k][l] = v;
            }
            double bv = 0;
#pragma unroll
            for (int r = 0; r < M; ++r) bv += g.invJ[k][r] * op.bt[r];
            beta[k] = bv;
        }
        if (sym) {
#pragma unroll
            for (int k = 0; k < M; ++k)
#pragma unroll
                for (int l = 0; l < k; ++l) Gp[k][l] = Gp[l][k];
        }
        const double* kt = rt->ktab + il * NB;
        const double* ct = rt->ctab + il * NB;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double d = 0;
            if (sym) {
#pragma unroll
                for (int k = 0; k < M; ++k) d += Gp[k][k] * kt[(k * 3 + k) * NN + j];
#pragma unroll
                for (int k = 0; k < M; ++k)
#pragma unroll
                    for (int l = k + 1; l < M; ++l) d += Gp[k][l] * (kt[(k * 3 + l) * NN + j] + kt[(l * 3 + k) * NN + j]);
            } else {
#pragma unroll
                for (int k = 0; k < M; ++k)
#pragma unroll
                    for (int l = 0; l < M; ++l) d += Gp[k][l] * kt[(k * 3 + l) * NN + j];
            }
            double adv = 0;
            if (!sym) {
#pragma unroll
                for (int l = 0; l < M; ++l) adv += beta[l] * ct[l * NN + j];
            }
            emit(j, g.measure * ((adv - d) + op.ct * tb->mtab[il * NB + j]));
        }
        return fsum;
    } else if constexpr (OPK == 1 && R == 1) {
        double G[M + 1][M];   // physical gradients of the M+1 barycentric coordinates
#pragma unroll
        for (int r = 0; r < M; ++r) {
            double s0 = 0;
#pragma unroll
            for (int k = 0; k < M; ++k) G[k + 1][r] = g.invJ[k][r], s0 -= g.invJ[k][r];
            G[0][r] = s0;
        }
        double gi[M];
#pragma unroll
        for (int r = 0; r < M; ++r) {
            double v = G[0][r];
#pragma unroll
            for (int k = 1; k <= M; ++k) v = il == k ? G[k][r] : v;
            gi[r] = v;
        }
        const double cm = op.t[0].coef * tb->wsum * g.measure;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double d = 0;
#pragma unroll
            for (int r = 0; r < M; ++r) d += gi[r] * G[j][r];
            emit(j, cm * (-d));
        }
        return fsum;
    } else {
        // gradients of the owned test function at every quadrature node (P1: constant over the cell)
        constexpr int NGQ = R == 1 ? 1 : NQ;
        double gi[NGQ][M];
#pragma unroll
        for (int q = 0; q < NGQ; ++q) phys_grad<M>(g, &tb->dpsi[(il * NQ + q) * 3], gi[q]);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double gj[NGQ][M];
#pragma unroll
            for (int q = 0; q < NGQ; ++q) phys_grad<M>(g, &tb->dpsi[(j * NQ + q) * 3], gj[q]);
            double value = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if constexpr (OPK == 1) {
                    double d = 0;
#pragma unroll
                    for (int r = 0; r < M; ++r) d += gi[q][r] * gj[q][r];
                    value += (op.t[0].coef * (-d)) * tb->qw[q];
                } else {
                    const double pi = op.needs_psi ? tb->psi[il * NQ + q] : 0.0;
                    const double pj = op.needs_psi ? tb->psi[j * NQ + q] : 0.0;
                    value += weak_form<M>(op, qrow0 + q, pi, pj, gi[R == 1 ? 0 : q], gj[R == 1 ? 0 : q]) * tb->qw[q];
                }
            }
            emit(j, value * g.measure);
        }
        return fsum;
    }
}

__device__ __forceinline__ const DevTables* stage_tables(const DevTables* gsrc, double* lds) {
    const double* src = reinterpret_cast<const double*>(gsrc);
    for (int i = threadIdx.x; i < kTablesDoubles; i += blockDim.x) lds[i] = src[i];
    return reinterpret_cast<const DevTables*>(lds);
}

// ---------------------------------------------------------------------------------------------------------------
// Row-owner assembly (default).  Workgroup = 256 consecutive matrix rows = 4 wavefronts = 4 adjacency slices.
// Lane `l` of wavefront `w` owns row 256*block + 64*w + l, walks the sliced-ELL adjacency of its slice (unit-stride
// int32 + packed-uint16 slot words across the wavefront), integrates its row of each visited element matrix and adds
// it into LDS at (rowptr[row] - rowptr[row0]) + slot.  The workgroup then streams its contiguous value range to HBM
// once.  No atomics, no colouring, bitwise reproducible, and for symmetric forms bitwise symmetric (both (i,j) and
// (j,i) sum the same products over the same cells in the same order).
// Replaces Assembler::discretize_operator + discretize_forcing (fdaPDE/finite_elements/fem_assembler.h:52-136).
// ---------------------------------------------------------------------------------------------------------------
template <int M, int R, int OPK>
__global__ __launch_bounds__(kAsmBlock) void k_assemble_rows(AsmArgs a, DevOp op) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    constexpr int NBW = (NB * 2 + 3) / 4;
    constexpr int NP = M == 2 ? 2 : 4;   // doubles per staged vertex (xyz padded to 32 B)
    extern __shared__ double lds[];
    const DevTables* tb = stage_tables(a.tables, lds);
    const DevRefTensors* rt = nullptr;
    double* xyz = lds + kTablesDoubles;                       // the block's vertex coordinates
    if constexpr (OPK == 3) {   // reference tensors of the constant-coefficient form behind the basis tables
        const double* src = reinterpret_cast<const double*>(a.reftab);
        for (int i = threadIdx.x; i < kRefDoubles; i += blockDim.x) xyz[i] = src[i];
        rt = reinterpret_cast<const DevRefTensors*>(xyz);
        xyz += kRefDoubles;
    }
    double* acc = xyz + (int64_t)a.lds_nodes * NP;            // the block's CSR value range

    const int64_t row0 = (int64_t)blockIdx.x * kAsmBlock;
    const int64_t row = row0 + threadIdx.x;
    const int64_t row_end = min(a.n_dofs, row0 + kAsmBlock);
    const bool want_matrix = a.vals != nullptr;
    const int32_t base = a.rowptr[row0];
    const int32_t blk_nnz = a.rowptr[row_end] - base;
    const bool in_lds = blk_nnz <= a.lds_acc_cap;
    const int32_t my0 = row < a.n_dofs ? a.rowptr[row] : 0;
    const int32_t my1 = row < a.n_dofs ? a.rowptr[row + 1] : 0;
    // stage the vertex coordinates of every cell this block visits: each node is fetched from HBM/L2 once per block
    // instead of once per (row, visit) -- the gathers of the visit loop below then hit LDS
    const int64_t bn0 = a.bn_off[blockIdx.x], nbn = a.bn_off[blockIdx.x + 1] - bn0;
    for (int i = threadIdx.x; i < nbn; i += kAsmBlock) {
        const int64_t node = a.bn_node[bn0 + i];
        if constexpr (M == 2) {
            *reinterpret_cast<double2*>(xyz + i * 2) = *reinterpret_cast<const double2*>(a.vcoords + node * 2);
        } else {
            const double4 v = *reinterpret_cast<const double4*>(a.vcoords + node * 4);
            *reinterpret_cast<double2*>(xyz + i * 4) = make_double2(v.x, v.y);
            *reinterpret_cast<double2*>(xyz + i * 4 + 2) = make_double2(v.z, 0.0);
        }
    }
    if (want_matrix) {
        if (in_lds) {
            for (int k = threadIdx.x; k < blk_nnz; k += kAsmBlock) acc[k] = 0.0;
        } else {
            for (int k = my0; k < my1; ++k) a.vals[k] = 0.0;
        }
    }
    __syncthreads();

    const int64_t slice = row >> 6;
    const int lane = threadIdx.x & 63;
    const int64_t bc0 = a.bc_off[blockIdx.x];
    double fsum = 0;
    if (row0 + (threadIdx.x & ~63) < a.n_dofs) {   // wave-uniform: slice exists
        const int64_t off = a.sl_off[slice], width = a.sl_off[slice + 1] - off;
        for (int64_t v = 0; v < width; ++v) {
            const int64_t at = (off + v) * kSlice + lane;
            const int32_t code = a.adj[at];
            if (code < 0) continue;
            uint32_t sw[NBW];
#pragma unroll
            for (int w = 0; w < NBW; ++w) sw[w] = a.slotw[at * NBW + w];
            const int64_t bc = bc0 + (code >> 4);
            const ushort4 lv = *reinterpret_cast<const ushort4*>(a.bc_vert + bc * 4);   // block-local vertex indices
            Geo<M> g;
            geo_from_vertices<M>(xyz + lv.x * NP, xyz + lv.y * NP, xyz + lv.z * NP, xyz + lv.w * NP, g);
            const int cell = (a.fq != nullptr || op.needs_rows) ? a.bc_cell[bc] : 0;   // only forcing / varying coefficients need it
            fsum += element_row<M, R, OPK>(a, op, tb, g, cell, code & 15, want_matrix, [&](int j, double value) {
                const uint32_t slot = (sw[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                if (in_lds)
                    acc[my0 - base + (int32_t)slot] += value;
                else
                    a.vals[my0 + (int32_t)slot] += value;
            }, rt);
        }
    }
    if (a.force != nullptr && row < a.n_dofs) a.force[row] = fsum;
    if (want_matrix && in_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < blk_nnz; k += kAsmBlock) a.vals[base + k] = acc[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Element-wise scatter variants (kept for cross-checking and for the measurements in DESIGN.md):
//   ATOMIC = true : one lane per (cell, local row), fp64 global atomics into the CSR slot found by binary search.
//   ATOMIC = false: the same kernel launched once per colour over colour-contiguous cell lists; cells of a colour
//                   share no DOF, so plain read-modify-write is race-free ("colour-partitioned passes").
// vals must be zeroed before the first launch.
// ---------------------------------------------------------------------------------------------------------------
template <int M, int R, bool ATOMIC>
__global__ __launch_bounds__(256) void k_assemble_scatter(AsmArgs a, DevOp op, const int32_t* cell_list, int64_t n_list) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    extern __shared__ double lds[];
    const DevTables* tb = stage_tables(a.tables, lds);
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_list * NB) return;
    const int64_t li = idx / NB;
    const int il = (int)(idx - li * NB);
    const int cell = cell_list ? cell_list[li] : (int)li;
    const int32_t* cd = a.cdofs + (int64_t)cell * NB;
    const int32_t row = cd[il];
    const int32_t k0 = a.rowptr[row], k1 = a.rowptr[row + 1];
    Geo<M> g;
    cell_geometry<M>(a, cell, g);
    const double f = element_row<M, R, 0>(a, op, tb, g, cell, il, a.vals != nullptr, [&](int j, double value) {
        const int32_t col = cd[j];
        int32_t lo = k0, hi = k1;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (a.colidx[mid] < col) lo = mid + 1; else hi = mid;
        }
        if (ATOMIC)
            unsafeAtomicAdd(&a.vals[lo], value);
        else
            a.vals[lo] += value;
    });
    if (a.force != nullptr) {
        if (ATOMIC) unsafeAtomicAdd(&a.force[row], f); else a.force[row] += f;
    }
}

// Integrator::quadrature_nodes (integrator.h:109-121): out row nq*cell_ext + q = J p_q + x0, column-major rows x N
template <int M>
__global__ void k_quadrature_nodes(AsmArgs a, const int32_t* cell_i2e, int nq, double* out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_cells * nq) return;
    const int64_t ci = idx / nq;
    const int q = (int)(idx - ci * nq);
    constexpr int NP = M == 2 ? 2 : 4;
    const int32_t* cv = a.cverts + ci * (M + 1);
    const double* x0 = a.vcoords + (int64_t)cv[0] * NP;
    const int64_t rows = a.n_cells * nq;
    const int64_t orow = (int64_t)cell_i2e[ci] * nq + q;
    for (int d = 0; d < M; ++d) {
        double v = 0;
        for (int k = 0; k < M; ++k) v += (a.vcoords[(int64_t)cv[k + 1] * NP + d] - x0[d]) * a.tables->qn[q * M + k];
        out[(int64_t)d * rows + orow] = v + x0[d];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Basis evaluation: the Psi matrices every downstream model asks for (PDE__::eval_basis,
// fdaPDE/pde/pde.h:149-158).
// ---------------------------------------------------------------------------------------------------------------
// Lagrange basis of order R at reference point xi, the reference's local node order (closed forms of tables.cpp)
template <int M, int R> __device__ __forceinline__ void eval_ref_basis(const double* xi, double* out) {
    double lam[M + 1];
    lam[0] = 1.0;
#pragma unroll
    for (int k = 0; k < M; ++k) lam[0] -= xi[k], lam[k + 1] = xi[k];
    if constexpr (R == 1) {
#pragma unroll
        for (int i = 0; i <= M; ++i) out[i] = lam[i];
    } else {
#pragma unroll
        for (int i = 0; i <= M; ++i) out[i] = lam[i] * (2.0 * lam[i] - 1.0);
        if constexpr (M == 2) {
            out[3] = 4.0 * lam[0] * lam[1], out[4] = 4.0 * lam[0] * lam[2], out[5] = 4.0 * lam[1] * lam[2];
        } else {   // ReferenceElement<3,2> nodes 4..9 = m12, m02, m01, m13, m23, m03
            out[4] = 4.0 * lam[1] * lam[2], out[5] = 4.0 * lam[0] * lam[2], out[6] = 4.0 * lam[0] * lam[1];
            out[7] = 4.0 * lam[1] * lam[3], out[8] = 4.0 * lam[2] * lam[3], out[9] = 4.0 * lam[0] * lam[3];
        }
    }
}
// pointwise_evaluation::eval (basis/lagrangian_basis.h:203-235) with the point location of TreeSearch::locate
// (geometry/tree_search.h:73-90) done through a uniform bin grid: one lane per location scans the cells registered in its
// bin and takes the first one whose barycentric coordinates are all >= -tol (Simplex::contains, geometry/simplex.h:118-131).
// cell_out: reference cell id or -1; values: n_basis basis values psi_h(invJ (p - x0)) per location.
template <int M, int R>
__global__ void k_eval_pointwise(AsmArgs a, int64_t n_locs, const double* locs /*col-major n_locs x M*/, const double* lo,
                                 const double* inv_h, const int32_t* dims, const int32_t* bin_ptr, const int32_t* bin_cells,
                                 const int32_t* cell_i2e, double tol, int32_t* cell_out, double* values) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    constexpr int NP = M == 2 ? 2 : 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_locs) return;
    double p[M];
    int64_t bin = 0;
    bool inside_box = true;
#pragma unroll
    for (int d = M - 1; d >= 0; --d) {
        p[d] = locs[(int64_t)d * n_locs + i];
        const double t = (p[d] - lo[d]) * inv_h[d];
        int b = (int)floor(t);
        if (b == dims[d] && t <= dims[d] + 1e-9) b = dims[d] - 1;   // points on the upper face of the bounding box
        inside_box &= b >= 0 && b < dims[d];
        bin = bin * dims[d] + (b < 0 ? 0 : (b >= dims[d] ? dims[d] - 1 : b));
    }
    int found = -1;
    double xi[M];
    if (inside_box) {
        for (int32_t k = bin_ptr[bin]; k < bin_ptr[bin + 1] && found < 0; ++k) {
            const int32_t cell = bin_cells[k];
            const int32_t* cv = a.cverts + (int64_t)cell * (M + 1);
            const double* x0 = a.vcoords + (int64_t)cv[0] * NP;
            Geo<M> g;
            if constexpr (M == 2)
                geo_from_vertices<2>(x0, a.vcoords + (int64_t)cv[1] * NP, a.vcoords + (int64_t)cv[2] * NP, nullptr, g);
            else
                geo_from_vertices<3>(x0, a.vcoords + (int64_t)cv[1] * NP, a.vcoords + (int64_t)cv[2] * NP,
                                     a.vcoords + (int64_t)cv[3] * NP, g);
            double z0 = 1.0;
            bool in = true;
#pragma unroll
            for (int r = 0; r < M; ++r) {
                double v = 0;
#pragma unroll
                for (int c = 0; c < M; ++c) v += g.invJ[r][c] * (p[c] - x0[c]);
                xi[r] = v, z0 -= v, in &= v >= -tol;
            }
            if (in && z0 >= -tol) found = cell;
        }
    }
    cell_out[i] = found >= 0 ? cell_i2e[found] : -1;
    double val[NB];
    if (found >= 0) eval_ref_basis<M, R>(xi, val);
#pragma unroll
    for (int h = 0; h < NB; ++h) values[i * NB + h] = found >= 0 ? val[h] : 0.0;
}
// per cell (reference numbering): measure and the integrals of the local basis functions,
//   int_e psi_h = measure * sum_q w_q psi_h(p_q)   (Integrator::integrate_cell, utils/integration/integrator.h:47-63)
// -- the ingredients of areal_evaluation::eval (basis/lagrangian_basis.h:238-283)
template <int M>
__global__ void k_cell_integrals(AsmArgs a, int nb, int nq, const int32_t* cell_i2e, double* measure, double* psi_int) {
    const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= a.n_cells) return;
    Geo<M> g;
    cell_geometry<M>(a, (int)ci, g);
    const int64_t ce = cell_i2e[ci];
    measure[ce] = g.measure;
    for (int h = 0; h < nb; ++h) {
        double v = 0;
        for (int q = 0; q < nq; ++q) v += a.tables->psi[h * nq + q] * a.tables->qw[q];
        psi_int[ce * nb + h] = v * g.measure;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// sum over the workgroup; result valid in every thread.  red must hold blockDim/64 + 1 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < nw; ++i) s += red[i];
        red[nw] = s;
    }
    __syncthreads();
    return red[nw];
}
// every workgroup re-reduces the producer's per-workgroup partials in the same fixed order: deterministic, no atomics
__device__ __forceinline__ double sum_partials(const double* partial, int n, double* red) {
    double v = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += partial[i];
    return block_sum(v, red);
}

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

__global__ __launch_bounds__(256) void k_spmv(SpmvArgs s) {
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
__global__ __launch_bounds__(256) void k_spmv_team(SpmvArgs s, int64_t n, int64_t rows_per_band) {
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
__global__ __launch_bounds__(256) void k_read_probe(const double2* src, int64_t n16, double* sink) {
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
__global__ __launch_bounds__(256) void k_stream_probe_unaligned(const double* vals, const int2* col2, int64_t n2, double* sink) {
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2 - 1; i += (int64_t)gridDim.x * blockDim.x) {
        const F64x2u v = *reinterpret_cast<const F64x2u*>(vals + 2 * i + 1);
        const int2 c = col2[i];
        acc += v.x * c.x + v.y * c.y;
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream_probe(const double2* vals2, const int2* col2, int64_t n2, double* sink) {
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
__global__ __launch_bounds__(256) void k_stream_probe_w(const double2* vals2, const int2* col2, int64_t n2, double* out) {
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
// Sum over aligned groups of T lanes with DPP row operations (VALU data path; the ds_bpermute the compiler emits for
// __shfl_xor goes through the LDS crossbar, shared by the 4 SIMDs of the CU: 40 of them per 32-row tile were on the
// critical path of the team kernels).  Every lane of the group ends up with the group's total.  T <= 16 stays inside a
// DPP row (16 lanes); wider groups finish with __shfl_xor.
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int T> __device__ __forceinline__ double team_sum(double v) {
    if constexpr (T >= 32) {
#pragma unroll
        for (int o = T / 2; o >= 16; o >>= 1) v += __shfl_xor(v, o, T);
    }
    if constexpr (T >= 16) v += dpp_mov_f64<0x140>(v);   // row_mirror:       lane i <-> 15 - i
    if constexpr (T >= 8) v += dpp_mov_f64<0x141>(v);    // row_half_mirror:  lane i <-> 7 - i
    if constexpr (T >= 4) v += dpp_mov_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    if constexpr (T >= 2) v += dpp_mov_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    return v;
}

typedef __attribute__((address_space(3))) volatile double lds_vf64_t;
struct __attribute__((packed, aligned(8))) F64x2 { double x, y; };
struct __attribute__((packed, aligned(4))) I32x2 { int x, y; };

// ABL (diagnostic builds only, selected by FDAPDE_SPMV_ABLATE; results are wrong on purpose):
//   1: no x gather (colidx still loaded and consumed)   2: gather confined to a 2 KiB window of x
//   4: plain (default cache policy) val / colidx loads instead of nontemporal ones (results stay correct)
// The matrix arrays are read exactly once per launch and are larger than the 256 MiB Infinity Cache, so they are
// loaded nontemporal (interleaved A/B in one process: 69.4 us vs 70.7 us with default-policy loads on C3).
template <int T, int U, int ABL = 0, int OCC = 4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, 8))) void k_spmv_team2(SpmvArgs s, int64_t n,
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
        if constexpr (C16) {
            const v2f64_t a = __builtin_nontemporal_load(reinterpret_cast<const v2f64_t*>(s.vals + kc));
            vv.x = a.x, vv.y = a.y;
            cc.x = (int)__builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(s.col16 + kc)), cc.y = 0;
        } else if constexpr (ALIGNED) {
            const v2f64_t a = __builtin_nontemporal_load(reinterpret_cast<const v2f64_t*>(s.vals + kc));
            const v2i32_t b = __builtin_nontemporal_load(reinterpret_cast<const v2i32_t*>(s.colidx + kc));
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
    auto load_tb = [&](int64_t b) -> v4i32_t {
        if constexpr (C16) {
            const int64_t bc = b < band_end ? b : band_begin;
            const int g = __builtin_amdgcn_readfirstlane((int)(bc >> 5));
            return *reinterpret_cast<const v4i32_t*>(s.tbase + 4 * (int64_t)g);
        } else
            return v4i32_t{0, 0, 0, 0};
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
                    if constexpr (ABL & 128)
                        __builtin_nontemporal_store(out, s.y + row);
                    else if constexpr (ABL & 256)
                        s.y[row & 4095] = out;   // diagnostic: same store instruction stream, 32 KiB footprint
                    else if constexpr (ABL & 512) {   // 16 B per lane: lanes 0..WROWS/2-1 store row pairs
                        const int j = lane % (WROWS / 2);
                        const double2 o2 = make_double2(ys[2 * j], ys[2 * j + 1]);
                        *reinterpret_cast<double2*>(s.y + base + 2 * j) = o2;
                    } else if constexpr (ABL & 1024) {
                        // agent-scope relaxed store = global_store ... sc1: written through, the line is not kept in this
                        // XCD's L2 (MI355X_MICROARCH.md, stores of each flavour), leaving the L2 to the gathered x
                        __hip_atomic_store(s.y + row, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else
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
__global__ __launch_bounds__(256) void k_spmv_c16p(SpmvArgs s, int64_t n, int64_t rows_per_band) {
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
        return __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(s.col16 + pair_at(rs)));
    };
    auto load_vals = [&](int rs, int re, F64x2& v) {
        const int k = (rs & ~1) + 2 * l;
        const v2f64_t a = __builtin_nontemporal_load(reinterpret_cast<const v2f64_t*>(s.vals + pair_at(rs)));
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

// ---------------------------------------------------------------------------------------------------------------
// solve set-up kernels
// ---------------------------------------------------------------------------------------------------------------
// scale[i] = 0 on Dirichlet rows, 1/sqrt(|A_ii|) elsewhere; flag[0] |= 1 if some interior diagonal is <= 0
__global__ void k_jacobi_scale(int64_t n, const int32_t* diag, const double* vals, const uint8_t* bnd, int use_bnd,
                               double* scale, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = vals[diag[i]];
    const bool b = use_bnd && bnd[i];
    if (!b && !(d > 0.0)) atomicOr(flag, 1);
    scale[i] = b ? 0.0 : 1.0 / sqrt(fabs(d));
}
// At = diag(scale) A diag(scale): symmetric Jacobi scaling == Jacobi preconditioning folded into the matrix stream.
// Rows and columns of Dirichlet DOFs vanish (scale = 0), which restricts the Krylov iteration to the interior block.
__global__ __launch_bounds__(256) void k_scale_matrix(int64_t n, const int32_t* rowptr, const int32_t* colidx,
                                                      const double* vals, const double* scale, double* out) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;   // 16 lanes per row
    const int l = threadIdx.x & 15;
    if (row >= n) return;
    const double si = scale[row];
    for (int k = rowptr[row] + l; k < rowptr[row + 1]; k += 16) out[k] = si * vals[k] * scale[colidx[k]];
}
// the same into the compact solver matrix: entries with map[k] < 0 are dropped (the diagonal, which scales to exactly 1,
// and every entry in a row or column of a Dirichlet DOF, which scales to exactly 0)
__global__ __launch_bounds__(256) void k_scale_matrix_compact(int64_t n, const int32_t* rowptr, const int32_t* colidx,
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
__global__ void k_lift(int64_t n, const uint8_t* bnd, const double* g, int use_bnd, double* gt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gt[i] = (use_bnd && bnd[i]) ? g[i] : 0.0;
}
// bt = scale * (f - A gt)  (y holds A gt): right-hand side of the scaled interior system.
// Cold start (u0 == nullptr): x = 0, r = bt.  Warm start: x = (u0 - gt) / scale on interior DOFs, r = bt - ax where ax holds
// At x (one extra SpMV by the caller between the two launches: first launch with ax == nullptr only fills x).
// partial[2 b] = sum r^2, partial[2 b + 1] = sum bt^2 (the stopping rule is relative to ||bt||, not to the warm residual).
__global__ __launch_bounds__(256) void k_krylov_init(int64_t n, const double* f, const double* y, const double* scale,
                                                      double* x, double* r, double* p, double* r0, double* partial,
                                                      const uint8_t* owned, const double* u0, const double* gt,
                                                      const double* ax, int fill_x_only) {
    __shared__ double red[8];
    double acc = 0, accb = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
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
    if (threadIdx.x == 0) partial[2 * blockIdx.x] = s, partial[2 * blockIdx.x + 1] = sb;
}
// scalars layout (device doubles): [0] reference norm^2 (||bt||^2), [1] rr_even, [2] rr_odd, [3] last rr, [4..] method specific
// ctl layout (device int32): [0] stop flag, [1] iterations done, [2] breakdown flag
__global__ __launch_bounds__(256) void k_krylov_init_fin(const double* partial, int np, double* sc, int32_t* ctl, double tol2) {
    __shared__ double red[8];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) a += partial[2 * i], b += partial[2 * i + 1];
    const double rr = block_sum(a, red);
    const double bb = block_sum(b, red);
    if (threadIdx.x == 0) {
        sc[0] = bb, sc[1] = rr, sc[2] = rr, sc[3] = rr;
        sc[4] = 1.0, sc[5] = 1.0, sc[6] = 1.0;   // bicgstab: rho, alpha, omega
        sc[9] = rr;                               // bicgstab: (r0, r0) of the first iteration
        ctl[0] = rr <= tol2 * bb ? 1 : 0, ctl[1] = 0, ctl[2] = 0;
    }
}
// out[0], out[1] = sums of the stride-2 partial pairs, fixed order; single workgroup
__global__ __launch_bounds__(256) void k_reduce_partials2(const double* part, int np, double* out) {
    __shared__ double red[8];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) a += part[2 * i], b += part[2 * i + 1];
    const double sa = block_sum(a, red);
    const double sb = block_sum(b, red);
    if (threadIdx.x == 0) out[0] = sa, out[1] = sb;
}
// K = M / dt + A  (FEMLinearParabolicSolver::solve, fem_linear_parabolic_solver.h:49), same pattern, elementwise
__global__ void k_matrix_combine(int64_t nnz, const double* mass, const double* stiff, double inv_dt, double* out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) out[k] = mass[k] * inv_dt + stiff[k];
}
// rhs = mu * inv_dt + f   (mu = M u_i)
__global__ void k_parabolic_rhs(int64_t n, const double* mu, double inv_dt, const double* f, double* rhs) {
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
__global__ __launch_bounds__(256) void k_cg_update_xr(int64_t n, const double* y, double* r, const double* part_in, int np_in,
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
__global__ __launch_bounds__(256) void k_cg_update_p(int64_t n, const double* r, double* p, double* x, const double* part_in,
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

// Single-reduction CG (Chronopoulos & Gear): the SpMV acts on r, both dot products of an iteration -- gamma = r.r and
// delta = r.(At r) -- are fused into it, and ONE kernel then updates all vectors:
//     beta = gamma / gamma_old ; alpha = gamma / (delta - beta gamma / alpha_old)
//     p = r + beta p ; s = w + beta s (= At p) ; x += alpha p ; r -= alpha s
// Two launches and (multi-GPU) one all-reduce per iteration instead of three and two.  Same Krylov iterates as CG in
// exact arithmetic.  part_in: stride-2 pairs (delta, gamma); scalars: sc[10 + parity] gamma_old, sc[12 + parity] alpha_old.
// Every workgroup takes the stop decision from the same reduced numbers, so no workgroup updates past convergence.
// if_slot / hb (multi-GPU, else nullptr): rows with if_slot[row] >= 0 take w from the all-reduced interface buffer hb, which
// saves the separate unpack launch (w itself is not read again)
__global__ __launch_bounds__(256) void k_cgsr_update(int64_t n, double* r, const double* w, double* p, double* s, double* x,
                                                      const double* part_in, int np_in, double* sc, int parity, int first,
                                                      double tol2, int32_t* ctl, const int32_t* if_slot, const double* hb) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kCgV) + threadIdx.x;
    double2* r2 = reinterpret_cast<double2*>(r);
    const double2* w2 = reinterpret_cast<const double2*>(w);
    double2* p2 = reinterpret_cast<double2*>(p);
    double2* s2 = reinterpret_cast<double2*>(s);
    double2* x2 = reinterpret_cast<double2*>(x);
    double2 rv[kCgV], wv[kCgV], pv[kCgV], sv[kCgV], xv[kCgV];
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        rv[k] = r2[ic], wv[k] = w2[ic], pv[k] = p2[ic], sv[k] = s2[ic], xv[k] = x2[ic];
        if (if_slot) {
            const int s0 = if_slot[2 * ic], s1 = if_slot[2 * ic + 1];
            if (s0 >= 0) wv[k].x = hb[s0];
            if (s1 >= 0) wv[k].y = hb[s1];
        }
    }
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np_in; i += blockDim.x) a += part_in[2 * i], b += part_in[2 * i + 1];
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
            p2[i] = pv[k], s2[i] = sv[k], x2[i] = xv[k], r2[i] = rv[k];
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
template <int kCgV>
__global__ __launch_bounds__(256) void k_cgf_update(int64_t n, const double* y, double* p, double* x, double* r,
                                                     const double* part_spmv, int np_spmv, const double* part_rr_in, int np_rr,
                                                     double* part_rr_out, double* sc, int first, double tol2, int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kCgV) + threadIdx.x;
    const double2* y2 = reinterpret_cast<const double2*>(y);
    double2* p2 = reinterpret_cast<double2*>(p);
    double2* x2 = reinterpret_cast<double2*>(x);
    double2* r2 = reinterpret_cast<double2*>(r);
    double2 yv[kCgV], pv[kCgV], xv[kCgV], rv[kCgV];
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        yv[k] = y2[ic], pv[k] = p2[ic], xv[k] = x2[ic], rv[k] = r2[ic];
    }
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np_spmv; i += blockDim.x) a += part_spmv[2 * i], b += part_spmv[2 * i + 1];
    const double pAp = block_sum(a, red);
    const double yy = block_sum(b, red);
    const double rr = first ? sc[1] : sum_partials(part_rr_in, np_rr, red);
    const bool last = blockIdx.x == gridDim.x - 1 && threadIdx.x == 0;
    if (!first && rr <= tol2 * sc[0]) {   // converged by the previous update: x, r stay as they are
        if (last) sc[3] = rr, ctl[0] = 1;
        return;
    }
    const double alpha = pAp > 0.0 ? rr / pAp : 0.0;
    const double est = alpha * alpha * yy - rr;
    const double beta = (est > 0.0 && rr > 0.0) ? est / rr : 0.0;
    double acc = 0;
#pragma unroll
    for (int k = 0; k < kCgV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) {
            xv[k].x += alpha * pv[k].x, xv[k].y += alpha * pv[k].y;
            rv[k].x -= alpha * yv[k].x, rv[k].y -= alpha * yv[k].y;
            pv[k].x = rv[k].x + beta * pv[k].x, pv[k].y = rv[k].y + beta * pv[k].y;
            x2[i] = xv[k], r2[i] = rv[k], p2[i] = pv[k];
            acc += rv[k].x * rv[k].x + rv[k].y * rv[k].y;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * y[i];
        r[i] = ri, p[i] = ri + beta * p[i];
        acc += ri * ri;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part_rr_out[blockIdx.x] = s;
    if (last) {
        sc[3] = rr;
        ctl[1] += 1;
        if (!(pAp > 0.0)) ctl[2] = 1, ctl[0] = 1;   // not SPD / breakdown
    }
}
// host poll of the fused-update CG: explicit r.r of the last update -> sc[3], stop flag
__global__ __launch_bounds__(256) void k_cgf_fin(const double* part_rr, int np, double* sc, double tol2, int32_t* ctl) {
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
__global__ __launch_bounds__(256) void k_bicg_p(int64_t n, const double* r, const double* v, double* p,
                                                 const double* part_in /* (r0.r, r.r) pairs */, int np_in, double* sc,
                                                 int first, int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kBiV) + threadIdx.x;
    const double2* r2 = reinterpret_cast<const double2*>(r);
    const double2* v2 = reinterpret_cast<const double2*>(v);
    double2* p2 = reinterpret_cast<double2*>(p);
    double2 rv[kBiV], vv[kBiV], pv[kBiV];
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        rv[k] = r2[ic];
        if (!first) vv[k] = v2[ic], pv[k] = p2[ic];
    }
    double a = 0;
    for (int i = threadIdx.x; i < np_in; i += blockDim.x) a += part_in[2 * i];
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
__global__ __launch_bounds__(256) void k_bicg_s(int64_t n, const double* r, const double* v, double* s,
                                                 const double* part_in /* (r0.v, .) */, int np_in, double* sc,
                                                 int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;   // uniform even if another workgroup raises the flag meanwhile
    const int64_t n2 = n >> 1, i0 = (int64_t)blockIdx.x * (256 * kBiV) + threadIdx.x;
    const double2* r2 = reinterpret_cast<const double2*>(r);
    const double2* v2 = reinterpret_cast<const double2*>(v);
    double2* s2 = reinterpret_cast<double2*>(s);
    double2 rv[kBiV], vv[kBiV];
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        rv[k] = r2[ic], vv[k] = v2[ic];
    }
    double a = 0;
    for (int i = threadIdx.x; i < np_in; i += blockDim.x) a += part_in[2 * i];
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
__global__ __launch_bounds__(256) void k_bicg_xr(int64_t n, const double* p, const double* s, const double* t,
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
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256, ic = i < n2 ? i : 0;
        pv[k] = p2[ic], sv[k] = s2[ic], tv[k] = t2[ic], qv[k] = q2[ic], xv[k] = x2[ic];
    }
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < np_in; i += blockDim.x) a += part_in[2 * i], b += part_in[2 * i + 1];
    const double ts = block_sum(a, red);
    const double tt = block_sum(b, red);
    const double omega = tt > 0.0 ? ts / tt : 0.0;
    const double alpha = sc[8];
    double d0 = 0, d1 = 0;
#pragma unroll
    for (int k = 0; k < kBiV; ++k) {
        const int64_t i = i0 + k * 256;
        if (i < n2) {
            x2[i] = make_double2(xv[k].x + alpha * pv[k].x + omega * sv[k].x, xv[k].y + alpha * pv[k].y + omega * sv[k].y);
            const double2 ri = make_double2(sv[k].x - omega * tv[k].x, sv[k].y - omega * tv[k].y);
            r2[i] = ri;
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
// multi-GPU BiCGStab: t.t over the owned rows of the ASSEMBLED t (the SpMV's fused y.y only sees this rank's sub-assembled
// part); per-workgroup partials, then out = (t.s already summed over ranks, local t.t) for the scalar all-reduce of out[1]
__global__ __launch_bounds__(256) void k_sq_owned(int64_t n, const double* t, const uint8_t* owned, double* part, const int32_t* ctl) {
    __shared__ double red[8];
    if (__syncthreads_or(ctl[0] != 0)) return;
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (owned[i]) a += t[i] * t[i];
    const double s = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_bicg_tt_fin(const double* part, int np, const double* ts_src, double* out) {
    __shared__ double red[8];
    const double s = sum_partials(part, np, red);
    if (threadIdx.x == 0) out[0] = ts_src[0], out[1] = s;
}
// closes a BiCGStab iteration: rho <- rho_new, alpha, omega = t.s/t.t recomputed from the same partials, stop test
__global__ __launch_bounds__(256) void k_bicg_fin(const double* part_ts, int np_ts, const double* part_rr, int np_rr,
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
// buf must be zero on entry.  Workgroup 0 also folds the local dot partials (stride 2) into buf[n_if] (+ second component
// into buf[n_if + 1]).
__global__ __launch_bounds__(256) void k_halo_pack(int64_t n_loc_if, const int32_t* dof, const int32_t* pos, const double* v,
                                                    double* buf, int64_t n_if, const double* part, int np) {
    __shared__ double red[8];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_loc_if) buf[pos[i]] = v[dof[i]];
    if (blockIdx.x == 0 && part != nullptr) {
        double a = 0, b = 0;
        for (int k = threadIdx.x; k < np; k += blockDim.x) a += part[2 * k], b += part[2 * k + 1];
        const double sa = block_sum(a, red);
        const double sb = block_sum(b, red);
        if (threadIdx.x == 0) buf[n_if] = sa, buf[n_if + 1] = sb;
    }
}
// the same without a prior memset: one lane per GLOBAL interface slot; inv[j] = this rank's DOF of slot j or -1
__global__ __launch_bounds__(256) void k_halo_pack_all(int64_t n_if, const int32_t* inv, const double* v, double* buf,
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
__global__ void k_halo_unpack(int64_t n_loc_if, const int32_t* dof, const int32_t* pos, const double* buf, double* v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_loc_if) v[dof[i]] = buf[pos[i]];
}
// out[0] = sum(part[0..np)) in the fixed order; single workgroup
__global__ __launch_bounds__(256) void k_reduce_partials(const double* part, int np, double* out) {
    __shared__ double red[8];
    const double s = sum_partials(part, np, red);
    if (threadIdx.x == 0) out[0] = s;
}
__global__ void k_diag_extract(int64_t n, const int32_t* diag, const double* vals, double* d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = vals[diag[i]];
}
// Jacobi scale from an already summed diagonal (multi-GPU)
__global__ void k_jacobi_scale_from_diag(int64_t n, const double* d, const uint8_t* bnd, int use_bnd, double* scale, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool b = use_bnd && bnd[i];
    if (!b && !(d[i] > 0.0)) atomicOr(flag, 1);
    scale[i] = b ? 0.0 : 1.0 / sqrt(fabs(d[i]));
}

// u = scale * x + gt   (back to the unscaled unknowns, Dirichlet values restored)
__global__ void k_unscale(int64_t n, const double* scale, const double* x, const double* gt, double* u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) u[i] = scale[i] * x[i] + gt[i];
}

// ---------------------------------------------------------------------------------------------------------------
// numbering changes at the boundary (reference numbering <-> internal numbering)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_gather_f64(int64_t n, const int32_t* idx, const double* src, double* dst) {   // dst[i] = src[idx[i]]
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_scatter_f64(int64_t n, const int32_t* idx, const double* src, double* dst) {  // dst[idx[i]] = src[i]
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}
// export of stiff() after a Dirichlet solve: FEMSolverBase::set_dirichlet_bc (fem_solver_base.h:148-149) zeroes the
// boundary rows and puts 1 on their diagonal; 16 lanes per row, output in reference slots
__global__ __launch_bounds__(256) void k_export_values(int64_t n, const int32_t* rowptr, const int32_t* colidx,
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
__global__ __launch_bounds__(256) void k_row_sums(int64_t n, const int32_t* rowptr, const double* vals, double* out) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const bool ok = row < n;
    double a = 0;
    if (ok)
        for (int k = rowptr[row] + l; k < rowptr[row + 1]; k += 16) a += vals[k];
    a = team_sum<16>(a);
    if (ok && l == 0) out[row] = a;
}
__global__ void k_fill_f64(int64_t n, double v, double* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}
// force export after a Dirichlet solve: force_[i] = g[i] on boundary DOFs (fem_solver_base.h:152)
__global__ void k_force_bc(int64_t n, const uint8_t* bnd, const double* g, double* f) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && bnd[i]) f[i] = g[i];
}

}  // namespace fdapde_hip
#endif
