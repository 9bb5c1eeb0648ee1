// kernels_assembly.h -- operator / forcing assembly (row-owner default, scatter cross-checks) and basis evaluation; see kernels.h
#ifndef FDAPDE_KERNELS_ASSEMBLY_H
#define FDAPDE_KERNELS_ASSEMBLY_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "internal.h"

#ifndef FDAPDE_OPK5_QUNROLL
#define FDAPDE_OPK5_QUNROLL 1
#endif

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
    int32_t kt_sym;       // Kt symmetric (with or without advection): AsmArgs::reftab holds the compact tensors (DevRefTensorsSym)
    int32_t var_kinds;    // which kinds have a space-varying leaf: 1 diffusion, 2 advection, 4 reaction (kt / bt / ct sum the CONSTANT leaves only)
    int32_t mirror;       // the expression is one the reference treats as SYMMETRIC (no advection leaf: every leaf's is_symmetric is true, diffusion.h:42)
                          // but its diffusion tensor is not: the reference then integrates only the pairs dof_i >= dof_j and mirrors them
                          // (fem_assembler.h:94-102, 116-117) -- k_mirror_reference_lower does the same to the assembled matrix (host-side flag)
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
// ... and where the summed diffusion tensor Kt is symmetric (every Laplacian, every symmetric K: advection or not), the COMPACT form of the same tensors:
//   kdiag[k*NN + i*NB + j] = ktab[(k,k)]   ksum[p*NN + ...] = ktab[(k,l)] + ktab[(l,k)], p = (0,1), (0,2), (1,2) -- the sum the symmetric evaluation
//   formed per entry at run time, made once on the host (the same fp64 add: the same bits)   ctab as above
// J^-1 Kt J^-T is then evaluated as a symmetric tensor: 6 + 3 + 1 table reads and multiply-adds per entry instead of 9 + 3 + 1, and 2.4 KB less of LDS.
struct DevRefTensorsSym {
    double kdiag[3 * kMaxBasis * kMaxBasis];
    double ksum[3 * kMaxBasis * kMaxBasis];
    double ctab[3 * kMaxBasis * kMaxBasis];
};
constexpr int kRefSymDoubles = sizeof(DevRefTensorsSym) / sizeof(double);

struct AsmArgs {
    int64_t n_dofs, n_cells;
    const int32_t* cverts;     // n_cells x (M+1), internal node ids
    const int32_t* cdofs;      // n_cells x nb, internal DOF ids
    const double* vcoords;     // internal node id -> NP doubles
    const int64_t* sl_off;     // adjacency slices
    const int32_t* lane_row;   // lane position -> row (-1: none), or nullptr = identity
    const int32_t* adj;
    const uint32_t* slotw;
    const int32_t* rowptr;
    const int32_t* colidx;
    const DevTables* tables;
    const DevRefTensors* reftab;   // OPK 3 / 5 only; DevRefTensorsSym where DevOp::kt_sym
    int32_t ref_doubles;           // ... and how many doubles of it the kernels stage in LDS
    double* vals;              // CSR values (internal slots) or nullptr
    double* vals2;             // k_assemble_rows<..., MASS2>: the mass matrix, assembled by the same launch
    const int32_t* diag;       // CSR slot of every row's diagonal (row_stat only)
    double* row_stat;          // nullptr, or [2 n_dofs]: (diagonal value, max |entry|) of every assembled row of `vals` -- what the Jacobi scaling of
                               // the solve reads instead of the whole matrix (k_jacobi_scale_stats); written by the row's owner from its LDS range
    const double* fq;          // forcing at quadrature nodes, internal cell order, or nullptr
    int fq_block;              // k_assemble_rows only.  1: fq holds one load coefficient per visit slot (k_visit_load_coeffs);
                               // 2: fq holds the samples in block-cell order (row group = block-cell index)
    double* force;             // forcing vector (internal DOF order) or nullptr
    int32_t lds_acc_cap;       // doubles available for the row accumulators
    // block-local tables of the row-owner kernel (host_setup.cpp): cells visited by the block's rows, their vertex nodes
    const int64_t* bc_off;
    const int32_t* bc_cell;
    const uint16_t* bc_vert;   // 4 per block-cell
    const int64_t* bn_off;
    const int32_t* bn_node;
    int32_t lds_nodes;         // coordinate slots reserved in LDS (max nodes of any block)
    int32_t lds_cells;         // k_assemble_items: block-cell slots reserved in LDS for the vertex-slot words (max cells of any block)
};

// A finished element-matrix entry (or load-vector entry) as a VALUE: the compiler may not merge the multiplication that produced it with
// the addition that accumulates it into one fused multiply-add.  The reference rounds every integral before it is summed
// (fem_assembler.h:97-107: `value` goes into a triplet, the triplets are added later), and the two row-owner kernels -- which accumulate
// the same addends in the same order through differently shaped code -- must not depend on where the optimiser finds a contraction.
// Applied where two kernels must agree bit for bit -- the P2 instantiations (R == 2: k_assemble_rows and k_assemble_items); the P1 sweep
// has one form only and keeps the contraction (C3 init 0.93 against 0.99 ms with the barrier: FDAPDE_ASM_ROUND_P1 decides).
#ifndef FDAPDE_ASM_ROUND_P1
#define FDAPDE_ASM_ROUND_P1 0
#endif
template <int R> __device__ __forceinline__ double rounded(double v) {
    if constexpr (R == 2 || FDAPDE_ASM_ROUND_P1)
        asm("" : "+v"(v));   // (not volatile: the statement may move with the code around it, it only hides the product from the contraction)
    return v;
}

// A finished value into an accumulator in LDS: `ds_add_f64` (the LDS adds it -- IEEE round-to-nearest on the rounded product, exactly what
// read + v_add_f64 + write computes) instead of a read-modify-write through registers.  No returned value, hence no LDS round trip for the wave to
// wait on, half the LDS instructions, and the add leaves the VALU.  A wavefront's LDS operations complete in program order and the lanes of one
// instruction hit distinct accumulators (distinct rows, or the distinct columns of one element row), so every accumulator still receives its
// addends in visit order: bitwise reproducible, bitwise symmetric for symmetric forms.  Every row-owner kernel (P1 and P2, both sweeps) adds this
// way, so kernels of different shape agree bit for bit; the product handed in is a finished double (no contraction reaches into the LDS).
__device__ __forceinline__ void lds_add(double* slot, double v) { unsafeAtomicAdd(slot, v); }

// One element row's NB values into the row's accumulators in LDS (the NB slots of a visit are distinct columns of the row).
template <int NB, int NBW> __device__ __forceinline__ void add_row_lds(double* acc_row, const uint32_t (&sw)[NBW], const double (&val)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) lds_add(acc_row + ((sw[j >> 1] >> ((j & 1) * 16)) & 0xffffu), val[j]);
}

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

// g_kl (a_k b_l) + g_lk (a_l b_k) with every product rounded on its own: swapping a and b swaps the two summands and nothing else, so
// with g_kl == g_lk the result is the same bits (a fused multiply-add would round one of the pair differently from the other;
// __dmul_rn / __dadd_rn are plain operators on this platform and get fused like them)
__device__ __forceinline__ double sym_pair(double g_kl, double g_lk, double a_k, double b_l, double a_l, double b_k) {
#pragma clang fp contract(off)
    const double p = a_k * b_l, q = a_l * b_k;
    const double x = g_kl * p, y = g_lk * q;
    return x + y;
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
//   4  space-varying coefficients of any kind: one pulled-back tensor / vector / scalar per quadrature node (below)
//   5  space-varying advection and / or reaction next to a diffusion part that does NOT vary (`-Lap + c(x)`, `-div K grad + b(x).grad`):
//      the constant leaves through the reference tensors as OPK 3, the varying ones per node without any tensor pull-back --
//        A_ij = |e| ( [OPK 3 sum of the constant leaves]_ij + sum_q w_q ( psi_i (J^-1 b_q . dpsi_j) + c_q psi_i psi_j ) )
//      (C5-size mesh, -Lap + c(x): 23 ms with OPK 4 -> see DESIGN.md 4.3)
template <int M, int R, int OPK, typename Emit>
__device__ __forceinline__ double element_row(const AsmArgs& a, const DevOp& op, const DevTables* tb, const Geo<M>& g, int cell,
                                              int il, bool want_matrix, Emit&& emit, const DevRefTensors* rt = nullptr,
                                              int64_t fcell = -1 /* >= 0: a.fq holds load coefficients per visit ... */,
                                              double fcoef = 0.0 /* ... and this is the visit's, loaded ahead by the caller */,
                                              int64_t frow = -1 /* >= 0: row group of the cell's samples in a.fq (block-cell order) */) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    constexpr int NQ = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 5);
    const int64_t qrow0 = (int64_t)NQ * cell;                                 // rows of space-varying coefficient data
    const int64_t frow0 = frow >= 0 ? (int64_t)NQ * frow : qrow0;             // rows of the forcing samples
    double fsum = 0;
    if (a.fq != nullptr) {
        if (fcell >= 0) {   // the quadrature sum was taken once per visit slot (k_visit_load_coeffs), in this very order
            fsum = fcoef;
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) fsum += (a.fq[frow0 + q] * tb->psi[il * NQ + q]) * tb->qw[q];
        }
        fsum *= g.measure;
    }
    if (!want_matrix) return fsum;
    if constexpr (OPK == 2) {
        const double cm = op.t[0].coef * op.t[0].cst[0] * g.measure;
#pragma unroll
        for (int j = 0; j < NB; ++j) emit(j, cm * tb->mtab[il * NB + j]);
        return fsum;
    } else if constexpr (OPK == 3 || OPK == 5) {
        constexpr int NN = NB * NB;
        double Gp[M][M], beta[M];
        const bool sym = op.tab_sym != 0;
        [[maybe_unused]] double accj[OPK == 5 ? NB : 1];
        if constexpr (OPK == 5) {   // the varying advection / reaction leaves, node by node (rows nq cell + q of their data)
            const bool vb = (op.var_kinds & 2) != 0, vc = (op.var_kinds & 4) != 0;
#pragma unroll
            for (int j = 0; j < NB; ++j) accj[j] = 0.0;
            constexpr int QU5 = R == 1 ? NQ : FDAPDE_OPK5_QUNROLL;
#pragma unroll QU5
            for (int q = 0; q < NQ; ++q) {
                const int64_t qrow = qrow0 + q;
                double bq[M], ctq = 0.0;
#pragma unroll
                for (int e = 0; e < M; ++e) bq[e] = 0.0;
                for (int t = 0; t < op.n; ++t) {
                    const DevTerm& T = op.t[t];
                    if (!T.space_varying) continue;
                    if (T.kind == FDAPDE_ADVECTION) {
#pragma unroll
                        for (int e = 0; e < M; ++e) bq[e] += T.coef * T.data[qrow * M + e];
                    } else if (T.kind == FDAPDE_REACTION) {
                        ctq += T.coef * T.data[qrow];
                    }
                }
                double betaq[M];
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    double bv = 0;
#pragma unroll
                    for (int r = 0; r < M; ++r) bv += g.invJ[k][r] * bq[r];
                    betaq[k] = bv;
                }
                const double pi = tb->psi[il * NQ + q], wq = tb->qw[q];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double v = 0;
                    if (vb) {
                        const double* gj = &tb->dpsi[(j * NQ + q) * 3];
                        double adv = 0;
#pragma unroll
                        for (int l = 0; l < M; ++l) adv += betaq[l] * gj[l];
                        v = adv * pi;
                    }
                    if (vc) v += ctq * (pi * tb->psi[j * NQ + q]);
                    accj[j] += v * wq;
                }
            }
        }
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
        const bool ksym = op.kt_sym != 0;   // (sym implies ksym)
        if (ksym) {
#pragma unroll
            for (int k = 0; k < M; ++k)
#pragma unroll
                for (int l = 0; l < k; ++l) Gp[k][l] = Gp[l][k];
        }
        const DevRefTensorsSym* rs = reinterpret_cast<const DevRefTensorsSym*>(rt);
        const double* kt = rt->ktab + il * NB;
        const double* ct = ksym ? rs->ctab + il * NB : rt->ctab + il * NB;
        const double* kd = rs->kdiag + il * NB;
        const double* ks = rs->ksum + il * NB;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double d = 0;
            if (ksym) {   // the symmetric tensor: diagonal terms, then the pairs k < l in the order (0,1), (0,2), (1,2) against the pre-summed table
#pragma unroll
                for (int k = 0; k < M; ++k) d += Gp[k][k] * kd[k * NN + j];
                int pr = 0;
#pragma unroll
                for (int k = 0; k < M; ++k)
#pragma unroll
                    for (int l = k + 1; l < M; ++l) {
                        // (pair index in the 3-D enumeration: (0,1) -> 0, (0,2) -> 1, (1,2) -> 2; in 2-D only (0,1))
                        d += Gp[k][l] * ks[(k + l - 1) * NN + j];
                        ++pr;
                    }
                (void)pr;
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
            if constexpr (OPK == 5) emit(j, g.measure * (((adv - d) + op.ct * tb->mtab[il * NB + j]) + accj[j]));
            else emit(j, g.measure * ((adv - d) + op.ct * tb->mtab[il * NB + j]));
        }
        return fsum;
    } else if constexpr (OPK == 4) {
        // SPACE-VARYING coefficients (Discretized*Field::forward(nq cell + q), fields/*_expressions.h), quadrature node OUTERMOST: the
        // coefficient rows of node q are read once per visit (the generic form reads them once per trial function), the leaves are summed
        // into ONE tensor / vector / scalar per node -- Kt_q, bt_q, ct_q, constants included -- and pulled back to the reference cell,
        //   Gp = J^-1 Kt_q J^-T,  beta = J^-1 bt_q,   value_j += w_q ( psi_i (beta . dpsi_j) - dpsi_i^T Gp dpsi_j + ct_q (psi_i psi_j) ),
        // so that the trial-function loop works on reference gradients from the tables (no physical gradient per (j, q)).  The bilinear
        // form is evaluated pairwise -- Gp[k][l] (gi_k gj_l) + Gp[l][k] (gi_l gj_k) -- and Gp is made exactly symmetric where Kt_q is:
        // a symmetric operator then gives A_ij == A_ji bit for bit, as the constant-coefficient forms do.
        // (C5-size mesh, diffusion + advection + reaction fields: init 163 -> see DESIGN.md 4.3)
        double accj[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) accj[j] = 0.0;
        bool any_adv = false;
        for (int t = 0; t < op.n; ++t) any_adv = any_adv || op.t[t].kind == FDAPDE_ADVECTION;
        // P2: the node loop stays a loop -- unrolled, the five (3-D) node bodies with their 13 coefficient loads each want 256 registers
        // + 311 spilled (k_assemble_items<3,2,4>: 1 248 B of scratch per lane); rolled 114 and none: C5-size, three fields, init 29.7 -> 10.6 ms.
        // P1 (4 trial functions, constant gradients) is faster unrolled (1.30 M tetrahedra: 1.49 against 1.81 ms).
        constexpr int QU = R == 1 ? NQ : 1;
#pragma unroll QU
        for (int q = 0; q < NQ; ++q) {
            const int64_t qrow = qrow0 + q;
            double Kt[M * M], bt[M], ctq = 0.0;
#pragma unroll
            for (int e = 0; e < M * M; ++e) Kt[e] = 0.0;
#pragma unroll
            for (int e = 0; e < M; ++e) bt[e] = 0.0;
            for (int t = 0; t < op.n; ++t) {
                const DevTerm& T = op.t[t];
                if (T.kind == FDAPDE_LAPLACIAN) {
#pragma unroll
                    for (int r = 0; r < M; ++r) Kt[r * M + r] += T.coef;
                } else if (T.kind == FDAPDE_DIFFUSION) {
#pragma unroll
                    for (int e = 0; e < M * M; ++e) Kt[e] += T.coef * (T.space_varying ? T.data[qrow * (M * M) + e] : T.cst[e]);
                } else if (T.kind == FDAPDE_ADVECTION) {
#pragma unroll
                    for (int e = 0; e < M; ++e) bt[e] += T.coef * (T.space_varying ? T.data[qrow * M + e] : T.cst[e]);
                } else if (T.kind == FDAPDE_REACTION) {
                    ctq += T.coef * (T.space_varying ? T.data[qrow] : T.cst[0]);
                }
            }
            bool ksym = true;
#pragma unroll
            for (int r = 0; r < M; ++r)
#pragma unroll
                for (int c2 = 0; c2 < r; ++c2) ksym = ksym && Kt[r * M + c2] == Kt[c2 * M + r];
            double Gp[M][M], beta[M];
#pragma unroll
            for (int k = 0; k < M; ++k) {
                double kr[M];   // row k of J^-1 Kt
#pragma unroll
                for (int c2 = 0; c2 < M; ++c2) {
                    double v = 0;
#pragma unroll
                    for (int r = 0; r < M; ++r) v += g.invJ[k][r] * Kt[r * M + c2];
                    kr[c2] = v;
                }
#pragma unroll
                for (int l = 0; l < M; ++l) {
                    double v = 0;
#pragma unroll
                    for (int c2 = 0; c2 < M; ++c2) v += kr[c2] * g.invJ[l][c2];
                    Gp[k][l] = v;
                }
                double bv = 0;
#pragma unroll
                for (int r = 0; r < M; ++r) bv += g.invJ[k][r] * bt[r];
                beta[k] = bv;
            }
            if (ksym) {
#pragma unroll
                for (int k = 0; k < M; ++k)
#pragma unroll
                    for (int l = 0; l < k; ++l) Gp[k][l] = Gp[l][k];
            }
            const double* gi = &tb->dpsi[(il * NQ + q) * 3];
            const double pi = tb->psi[il * NQ + q];
            const double wq = tb->qw[q];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const double* gj = &tb->dpsi[(j * NQ + q) * 3];
                double d = 0;
#pragma unroll
                for (int k = 0; k < M; ++k) d += Gp[k][k] * (gi[k] * gj[k]);
#pragma unroll
                for (int k = 0; k < M; ++k)
#pragma unroll
                    for (int l = k + 1; l < M; ++l) d += sym_pair(Gp[k][l], Gp[l][k], gi[k], gj[l], gi[l], gj[k]);
                double adv = 0;
                if (any_adv) {
#pragma unroll
                    for (int l = 0; l < M; ++l) adv += beta[l] * gj[l];
                    adv *= pi;
                }
                accj[j] += ((adv - d) + ctq * (pi * tb->psi[j * NQ + q])) * wq;
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) emit(j, accj[j] * g.measure);
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
// MASS2: FEMSolverBase::init assembles the operator, the forcing AND the mass matrix (fem_solver_base.h:113-136); the three share every
// index stream of the visit loop and the cell geometry, so the mass rows are accumulated in the same sweep (second accumulator range in
// LDS; a visit's mass row is |e| times a tabulated reference row: NB multiply-adds).  Same visits in the same order as the separate mass
// sweep: the same bits.
// MASS2 == 2: the mass rows in a SECOND pass over the block's visits inside the same launch, through the same accumulator range (no extra
// LDS, hence no occupancy lost): the block's index streams, vertex slots and staged coordinates are re-read from L2 / LDS right after the
// first pass instead of from HBM by a launch of its own.
template <int M, int R, int OPK, int MASS2 = 0>
static __global__ __launch_bounds__(kAsmBlock) void k_assemble_rows(AsmArgs a, DevOp op) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    constexpr int NBW = (NB * 2 + 3) / 4;
    constexpr int NP = M == 2 ? 2 : 3;   // doubles per staged vertex in LDS (unpadded: C3 blocks then fit three to a CU, not two)
    extern __shared__ double lds[];
    // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (own L2 each), so workgroup b serves block
    // (b % 8) * band + b / 8 -- an XCD walks a contiguous range of blocks, and what neighbouring blocks share (vertex coordinates
    // on their common boundary, the forcing samples and vertex lists of the cells both visit) is found in its L2
    const int64_t n_blk = (a.n_dofs + kAsmBlock - 1) / kAsmBlock, band = (n_blk + 7) / 8;
    const int64_t blk = (int64_t)(blockIdx.x & 7) * band + (blockIdx.x >> 3);
    if (blk >= n_blk || (int64_t)(blockIdx.x >> 3) >= band) return;   // uniform for the workgroup
    const DevTables* tb = stage_tables(a.tables, lds);
    const DevRefTensors* rt = nullptr;
    double* xyz = lds + kTablesDoubles;                       // the block's vertex coordinates
    if constexpr (OPK == 3 || OPK == 5) {   // reference tensors of the constant-coefficient form behind the basis tables
        const double* src = reinterpret_cast<const double*>(a.reftab);
        for (int i = threadIdx.x; i < a.ref_doubles; i += blockDim.x) xyz[i] = src[i];
        rt = reinterpret_cast<const DevRefTensors*>(xyz);
        xyz += a.ref_doubles;
    }
    double* acc = xyz + (int64_t)a.lds_nodes * NP;            // the block's CSR value range
    double* acc2 = acc + a.lds_acc_cap;                       // ... of the mass matrix (MASS2; the host made sure both ranges fit)

    const int64_t row0 = blk * kAsmBlock;
    const int64_t pos = row0 + threadIdx.x;   // lane position; the adjacency slices are laid out by position
    int64_t row = pos;
    if (a.lane_row) {   // rows of the block dealt to its lanes by visit count (host_setup.cpp)
        const int32_t lr = a.lane_row[pos];
        row = lr < 0 ? a.n_dofs : (int64_t)lr;
    }
    const int64_t row_end = min(a.n_dofs, row0 + kAsmBlock);
    const bool want_matrix = a.vals != nullptr;
    const int32_t base = a.rowptr[row0];
    const int32_t blk_nnz = a.rowptr[row_end] - base;
    const bool in_lds = blk_nnz <= a.lds_acc_cap;
    const int32_t my0 = row < a.n_dofs ? a.rowptr[row] : 0;
    const int32_t my1 = row < a.n_dofs ? a.rowptr[row + 1] : 0;
    // stage the vertex coordinates of every cell this block visits: each node is fetched from HBM/L2 once per block
    // instead of once per (row, visit) -- the gathers of the visit loop below then hit LDS
    const int64_t bn0 = a.bn_off[blk], nbn = a.bn_off[blk + 1] - bn0;
    for (int i = threadIdx.x; i < nbn; i += kAsmBlock) {
        const int64_t node = a.bn_node[bn0 + i];
        if constexpr (M == 2) {
            *reinterpret_cast<double2*>(xyz + i * 2) = *reinterpret_cast<const double2*>(a.vcoords + node * 2);
        } else {
            const double4 v = *reinterpret_cast<const double4*>(a.vcoords + node * 4);   // global copy stays padded to 32 B
            xyz[i * 3] = v.x, xyz[i * 3 + 1] = v.y, xyz[i * 3 + 2] = v.z;
        }
    }
    if (want_matrix) {
        if (in_lds) {
            for (int k = threadIdx.x; k < blk_nnz; k += kAsmBlock) acc[k] = 0.0;
            if constexpr (MASS2 == 1)
                for (int k = threadIdx.x; k < blk_nnz; k += kAsmBlock) acc2[k] = 0.0;
        } else {
            for (int k = my0; k < my1; ++k) a.vals[k] = 0.0;
        }
    }
    __syncthreads();

    const int64_t slice = pos >> 6;
    const int lane = threadIdx.x & 63;
    const int64_t bc0 = a.bc_off[blk];
    double fsum = 0;
    if (row0 + (threadIdx.x & ~63) < a.n_dofs) {   // wave-uniform: slice exists
        const int64_t off = a.sl_off[slice], width = a.sl_off[slice + 1] - off;
        // two-stage software pipeline over the visits: the index words of visit v + 2 and the vertex indices of visit v + 1 are
        // in flight while visit v integrates (the loop body is a dependent chain adj -> bc_vert -> LDS coordinates otherwise)
        auto load_code = [&](int64_t v) -> int32_t { return v < width ? a.adj[(off + v) * kSlice + lane] : -1; };
        auto load_lv = [&](int32_t code) -> ushort4 {
            return *reinterpret_cast<const ushort4*>(a.bc_vert + (bc0 + ((code < 0 ? 0 : code) >> 4)) * 4);
        };
        auto load_sw = [&](int64_t v, uint32_t (&w)[NBW]) {
            const int64_t at = (off + (v < width ? v : 0)) * kSlice + lane;
#pragma unroll
            for (int k = 0; k < NBW; ++k) w[k] = a.slotw[at * NBW + k];
        };
        // load coefficient of the forcing for a visit: one coalesced double per visit slot, streamed next to the adjacency word and
        // requested ahead like it (gathered by cell id inside the visit, the forcing cost 0.36 ms of a C3 init)
        const bool fblk = a.fq != nullptr && a.fq_block == 1;
        const bool fbc = a.fq != nullptr && a.fq_block == 2;   // samples in block-cell order: no cell id, no gather outside the block's window
        auto load_fc = [&](int64_t v) -> double { return (fblk && v < width) ? a.fq[(off + v) * kSlice + lane] : 0.0; };
        int32_t code_n = load_code(0), code_nn = load_code(1);
        ushort4 lv_n = load_lv(code_n);
        double fc_n = load_fc(0);
        uint32_t sw_n[NBW];
        load_sw(0, sw_n);
        for (int64_t v = 0; v < width; ++v) {
            const int32_t code = code_n;
            const ushort4 lv = lv_n;                  // block-local vertex indices of this visit
            uint32_t sw[NBW];
#pragma unroll
            for (int k = 0; k < NBW; ++k) sw[k] = sw_n[k];
            const double fc = fc_n;
            code_n = code_nn, code_nn = load_code(v + 2);
            lv_n = load_lv(code_n);
            fc_n = load_fc(v + 1);
            load_sw(v + 1, sw_n);
            if (code < 0) continue;
            const int64_t bc = bc0 + (code >> 4);
            Geo<M> g;
            geo_from_vertices<M>(xyz + lv.x * NP, xyz + lv.y * NP, xyz + lv.z * NP, xyz + lv.w * NP, g);
            // the global cell id is needed by varying coefficients and by forcing samples kept in cell order only
            const int cell = ((a.fq != nullptr && a.fq_block == 0) || op.needs_rows) ? a.bc_cell[bc] : 0;
            if constexpr (R == 2) {   // the NB values of the visit's row first, then ONE read-modify-write round trip for all of them (add_row_lds)
                double val[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) val[j] = 0.0;
                fsum += rounded<R>(element_row<M, R, OPK>(a, op, tb, g, cell, code & 15, want_matrix, [&](int j, double value) { val[j] = rounded<R>(value); },
                                                          rt, fblk ? bc : (int64_t)-1, fc, fbc ? bc : (int64_t)-1));
                if (want_matrix) {
                    if (in_lds) {
                        add_row_lds<NB, NBW>(acc + (my0 - base), sw, val);
                    } else {
#pragma unroll
                        for (int j = 0; j < NB; ++j) a.vals[my0 + (int32_t)((sw[j >> 1] >> ((j & 1) * 16)) & 0xffffu)] += val[j];
                    }
                }
            } else {
                fsum += rounded<R>(element_row<M, R, OPK>(a, op, tb, g, cell, code & 15, want_matrix, [&](int j, double value) {
                    const uint32_t slot = (sw[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                    if (in_lds)
                        lds_add(&acc[my0 - base + (int32_t)slot], value);
                    else
                        a.vals[my0 + (int32_t)slot] += rounded<2>(value);   // (a finished product here too: the same bits as through the LDS)
                }, rt, fblk ? bc : (int64_t)-1, fc, fbc ? bc : (int64_t)-1));
            }
            if constexpr (MASS2 == 1) {   // (OPK 2's own formula with coefficient 1: cm = 1.0 * 1.0 * |e|)
                const double cm = 1.0 * 1.0 * g.measure;
                const int il = code & 15;
                if constexpr (R == 2) {
                    double mv[NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) mv[j] = rounded<R>(cm * tb->mtab[il * NB + j]);
                    add_row_lds<NB, NBW>(acc2 + (my0 - base), sw, mv);
                } else {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const uint32_t slot = (sw[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                        lds_add(&acc2[my0 - base + (int32_t)slot], cm * tb->mtab[il * NB + j]);
                    }
                }
            }
        }
    }
    if (a.force != nullptr && row < a.n_dofs) a.force[row] = fsum;
    if (want_matrix && in_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < blk_nnz; k += kAsmBlock) a.vals[base + k] = acc[k];
        if (a.row_stat != nullptr && row < a.n_dofs) {
            double rmax = 0.0;
            for (int k = my0; k < my1; ++k) rmax = fmax(rmax, fabs(acc[k - base]));
            a.row_stat[2 * row] = acc[a.diag[row] - base], a.row_stat[2 * row + 1] = rmax;
        }
        if constexpr (MASS2 == 1)
            for (int k = threadIdx.x; k < blk_nnz; k += kAsmBlock) a.vals2[base + k] = acc2[k];
    }
    if constexpr (MASS2 == 2) {   // second pass: the mass rows through the same accumulators (the host launches this form only where in_lds holds)
        __syncthreads();
        for (int k = threadIdx.x; k < blk_nnz; k += kAsmBlock) acc[k] = 0.0;
        __syncthreads();
        if (row0 + (threadIdx.x & ~63) < a.n_dofs) {
            const int64_t off = a.sl_off[slice], width = a.sl_off[slice + 1] - off;
            auto load_code = [&](int64_t v) -> int32_t { return v < width ? a.adj[(off + v) * kSlice + lane] : -1; };
            auto load_lv = [&](int32_t code) -> ushort4 { return *reinterpret_cast<const ushort4*>(a.bc_vert + (bc0 + ((code < 0 ? 0 : code) >> 4)) * 4); };
            auto load_sw = [&](int64_t v, uint32_t (&w)[NBW]) {
                const int64_t at = (off + (v < width ? v : 0)) * kSlice + lane;
#pragma unroll
                for (int k = 0; k < NBW; ++k) w[k] = a.slotw[at * NBW + k];
            };
            int32_t code_n = load_code(0), code_nn = load_code(1);
            ushort4 lv_n = load_lv(code_n);
            uint32_t sw_n[NBW];
            load_sw(0, sw_n);
            for (int64_t v = 0; v < width; ++v) {
                const int32_t code = code_n;
                const ushort4 lv = lv_n;
                uint32_t sw[NBW];
#pragma unroll
                for (int k = 0; k < NBW; ++k) sw[k] = sw_n[k];
                code_n = code_nn, code_nn = load_code(v + 2);
                lv_n = load_lv(code_n);
                load_sw(v + 1, sw_n);
                if (code < 0) continue;
                Geo<M> g;
                geo_from_vertices<M>(xyz + lv.x * NP, xyz + lv.y * NP, xyz + lv.z * NP, xyz + lv.w * NP, g);
                const double cm = 1.0 * 1.0 * g.measure;   // (OPK 2's own formula with coefficient 1)
                const int il = code & 15;
                if constexpr (R == 2) {
                    double mv[NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) mv[j] = rounded<R>(cm * tb->mtab[il * NB + j]);
                    add_row_lds<NB, NBW>(acc + (my0 - base), sw, mv);
                } else {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const uint32_t slot = (sw[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                        lds_add(&acc[my0 - base + (int32_t)slot], cm * tb->mtab[il * NB + j]);
                    }
                }
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < blk_nnz; k += kAsmBlock) a.vals2[base + k] = acc[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Row-owner assembly, VISIT-PARALLEL form -- for spaces whose rows differ a lot in visit count (P2: a vertex row is visited by
// ~24 tetrahedra, an edge row by ~6; host_setup.cpp then deals the rows of a block to its lane positions in DESCENDING order of
// their visit count).  k_assemble_rows walks a row's visits one after the other in one lane: a block then takes as long as its
// longest row (24 visits) while three of its four wavefronts are done after 6-8, and with ~83 KB of LDS per block there is one
// block, i.e. ONE wavefront per SIMD, to hide a visit's dependent chain behind (r4 counters on C5: SQ_WAIT_ANY 83 % of the wave
// cycles, 35 cycles per instruction).  Here the (row, visit) pairs of a block are the work items:
//   * item t <-> (visit index v, lane position q): because the positions are sorted by visit count, the rows that HAVE a visit v
//     are the prefix [0, n_v) of the positions, so the items of a block, v-major, are addressed by a prefix sum over v -- no
//     extra index array, the adjacency slices are read where they lie;
//   * the block has 8 / 16 wavefronts (512 / 1024 threads: 2 / 4 per SIMD for the same LDS).  While more than 64 rows still have
//     visits, wavefront w takes 16 contiguous positions x 4 visit indices per step (phase A: every wavefront is busy, the index
//     words of a step are contiguous runs); the tail -- the few long rows -- is spread as 4 positions x 16 visit indices
//     (phase B).  Every lane integrates a full element row per step, whatever the length of the row it belongs to;
//   * an item's values go into the block's accumulators IN VISIT ORDER: after every step, one accumulation round per visit index
//     of the step (the items of one visit index belong to different rows, hence to different slots).  Inside a phase all items
//     of a row live in ONE wavefront, whose LDS operations complete in program order, so the rounds need no barrier; one
//     workgroup barrier separates the phases.  (Measured on C5, init = operator + forcing + mass: row-walking 8.9 ms; items dealt
//     v-major to all 1024 threads with a workgroup barrier per round 4.6 ms; wavefronts owning strided positions, no barrier but
//     scattered index words, 5.7 ms.)  Every slot receives exactly the addends of k_assemble_rows in exactly its order: the same
//     bits, no atomics, bitwise symmetric for symmetric forms.
// The forcing sum of a row is accumulated the same way (one more accumulator per position).  MASS2 == 2: the mass rows in a
// second sweep over the items through the same accumulators, as in k_assemble_rows.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kItemsMaxVisits = 64;   // visit lists longer than this keep the row-walking kernel (the host checks)
template <int M, int R, int OPK, int MASS2, int THREADS>
static __global__ __launch_bounds__(THREADS) void k_assemble_items(AsmArgs a, DevOp op) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    constexpr int NBW = (NB * 2 + 3) / 4;
    constexpr int NP = M == 2 ? 2 : 3;
    constexpr int NW = THREADS / 64, RW = kAsmBlock / NW;   // wavefronts; rows (lane positions) a wavefront owns: w, w + NW, w + 2 NW, ...
    static_assert(RW <= 64 && RW * NW == kAsmBlock, "a wavefront owns at most 64 rows");
    extern __shared__ double lds[];
    __shared__ int32_t width_s[kAsmBlock / kSlice];            // visit rows of the block's adjacency slices (= visits of the longest row of each)
    __shared__ int32_t rbase_s[kAsmBlock];                     // first accumulator of the position's row
    __shared__ double facc_s[kAsmBlock];                       // forcing sums by position
    __shared__ int64_t sloff_s[kAsmBlock / kSlice];            // first visit row of the block's adjacency slices
    const int64_t n_blk = (a.n_dofs + kAsmBlock - 1) / kAsmBlock, band = (n_blk + 7) / 8;
    const int64_t blk = (int64_t)(blockIdx.x & 7) * band + (blockIdx.x >> 3);   // XCD-aware block order, as k_assemble_rows
    if (blk >= n_blk || (int64_t)(blockIdx.x >> 3) >= band) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DevTables* tb = nullptr;
    {
        const double* src = reinterpret_cast<const double*>(a.tables);
        for (int i = tid; i < kTablesDoubles; i += THREADS) lds[i] = src[i];
        tb = reinterpret_cast<const DevTables*>(lds);
    }
    const DevRefTensors* rt = nullptr;
    double* xyz = lds + kTablesDoubles;
    if constexpr (OPK == 3 || OPK == 5) {
        const double* src = reinterpret_cast<const double*>(a.reftab);
        for (int i = tid; i < a.ref_doubles; i += THREADS) xyz[i] = src[i];
        rt = reinterpret_cast<const DevRefTensors*>(xyz);
        xyz += a.ref_doubles;
    }
    uint2* lvs = reinterpret_cast<uint2*>(xyz + (int64_t)a.lds_nodes * NP);   // the block-cells' vertex slots (4 x 16 bit each)
    double* acc = reinterpret_cast<double*>(lvs + a.lds_cells);
    [[maybe_unused]] double* acc2 = acc + a.lds_acc_cap;   // MASS2 == 1: the mass matrix's accumulators, filled by the SAME sweep (the host made sure both fit)
    const int64_t row0 = blk * kAsmBlock, row_end = min(a.n_dofs, row0 + kAsmBlock);
    const bool want_matrix = a.vals != nullptr;
    const int32_t base = a.rowptr[row0];
    const int32_t blk_nnz = a.rowptr[row_end] - base;
    const int64_t bn0 = a.bn_off[blk], nbn = a.bn_off[blk + 1] - bn0;
    for (int i = tid; i < nbn; i += THREADS) {
        const int64_t node = a.bn_node[bn0 + i];
        if constexpr (M == 2) {
            *reinterpret_cast<double2*>(xyz + i * 2) = *reinterpret_cast<const double2*>(a.vcoords + node * 2);
        } else {
            const double4 v = *reinterpret_cast<const double4*>(a.vcoords + node * 4);
            xyz[i * 3] = v.x, xyz[i * 3 + 1] = v.y, xyz[i * 3 + 2] = v.z;
        }
    }
    const int64_t bc0 = a.bc_off[blk], nbc = a.bc_off[blk + 1] - bc0;
    for (int i = tid; i < nbc; i += THREADS) lvs[i] = reinterpret_cast<const uint2*>(a.bc_vert)[bc0 + i];
    if (want_matrix)
        for (int k = tid; k < blk_nnz; k += THREADS) acc[k] = 0.0;
    if constexpr (MASS2 == 1)
        for (int k = tid; k < blk_nnz; k += THREADS) acc2[k] = 0.0;
    const int64_t slice0 = row0 >> 6;   // the block's four adjacency slices
    const int64_t n_slices_all = (a.n_dofs + kSlice - 1) / kSlice;
    if (tid < kAsmBlock) {
        const int64_t s = slice0 + (tid >> 6);
        int32_t rb = 0;
        if (s < n_slices_all) {
            if ((tid & 63) == 0) sloff_s[tid >> 6] = a.sl_off[s], width_s[tid >> 6] = (int32_t)(a.sl_off[s + 1] - a.sl_off[s]);
            const int32_t lr = a.lane_row[row0 + tid];
            rb = lr >= 0 ? a.rowptr[lr] - base : 0;
        } else if ((tid & 63) == 0)
            sloff_s[tid >> 6] = 0, width_s[tid >> 6] = 0;
        rbase_s[tid] = rb, facc_s[tid] = 0.0;
    }
    __syncthreads();
    // ---- the item (visit v, position q) exists if the row at q has more than v visits: v inside its slice's width and a valid code
    //      there (padding: -1).  Positions are sorted by count, so the rows that have a visit v are a prefix of the positions; the longest
    //      row of the block is the first of slice 0, and the rows at positions >= 64 have at most width(slice 1) visits -- no counting.
    //      Two phases, rows pinned to wavefronts inside each:
    //      A  while more than 64 rows still have visits: wavefront w takes the positions [QA w, QA (w + 1)), VA visit indices per step
    //         (QA = 256 / NW contiguous positions: the index words of a step are VA contiguous runs);
    //      B  the tail, at most 64 long rows: wavefront w takes the positions [QB w, QB (w + 1)), VB visit indices per step.
    //      One workgroup barrier between the phases (a row changes hands there), none inside them.
    constexpr int QA = kAsmBlock / NW, VA = 64 / QA, QB = 64 / NW, VB = 64 / QB;
    const int maxv = width_s[0];
    const int tail_from = width_s[1];   // rows at positions >= 64 have at most this many visits
    const bool fblk = a.fq != nullptr && a.fq_block == 1, fbc = a.fq != nullptr && a.fq_block == 2;
    auto sweep = [&](auto mass_pass) {
        constexpr bool MASS = decltype(mass_pass)::value;
        bool tail = false;
        for (int v0 = 0; v0 < maxv;) {   // (every quantity that steers the loop is uniform for the workgroup)
            if (!tail && tail_from <= v0) {
                tail = true;
                __syncthreads();   // rows change hands: everything phase A added is in place
            }
            // phase B: the tail's 64 positions x VB visit indices of a step as tiles of QA positions x VA visit indices, like phase A's: wavefront
            // w takes position group w % GQ and visit range w / GQ.  A row's items of one step then sit in GR wavefronts, which accumulate one
            // after the other (a workgroup barrier between the ranges) -- VA accumulation rounds per wavefront instead of VB: the rounds (NB LDS
            // read-modify-writes under a mask that leaves QB lanes active) were what a phase-B step spent most of its instructions on
            constexpr int GQ = 64 / QA, GR = NW / GQ;
            static_assert(GQ * QA == 64 && GR * GQ == NW && GR * VA == VB, "phase B tiles");
            const int gq = wave % GQ, gr = wave / GQ;
            const int V = tail ? VB : VA;
            const int v = tail ? v0 + VA * gr + lane / QA : v0 + lane / QA;
            const int q = tail ? QA * gq + lane % QA : QA * wave + lane % QA;
            const int64_t at = (sloff_s[q >> 6] + v) * kSlice + (q & 63);
            const int32_t code = v < width_s[q >> 6] ? a.adj[at] : -1;
            const bool on = code >= 0;
            double val[NB];
            [[maybe_unused]] double mval[NB];   // MASS2 == 1: the visit's row of the mass matrix, accumulated in the same rounds
            double fval = 0;
            uint32_t sw[NBW];
            if (on) {
#pragma unroll
                for (int k = 0; k < NBW; ++k) sw[k] = a.slotw[at * NBW + k];
                const int32_t bcl = code >> 4;
                const uint2 lw = lvs[bcl];
                const unsigned l0 = lw.x & 0xffffu, l1 = lw.x >> 16, l2 = lw.y & 0xffffu, l3 = lw.y >> 16;
                Geo<M> g;
                geo_from_vertices<M>(xyz + l0 * NP, xyz + l1 * NP, xyz + l2 * NP, xyz + l3 * NP, g);
                if constexpr (MASS) {
                    const double cm = 1.0 * 1.0 * g.measure;   // (OPK 2's own formula with coefficient 1)
                    const int il = code & 15;
#pragma unroll
                    for (int j = 0; j < NB; ++j) val[j] = rounded<R>(cm * tb->mtab[il * NB + j]);
                } else {
                    const int64_t bc = bc0 + bcl;
                    const int cell = ((a.fq != nullptr && a.fq_block == 0) || op.needs_rows) ? a.bc_cell[bc] : 0;
                    const double fc = fblk ? a.fq[at] : 0.0;
                    fval = rounded<R>(element_row<M, R, OPK>(a, op, tb, g, cell, code & 15, want_matrix, [&](int j, double value) { val[j] = rounded<R>(value); }, rt,
                                                             fblk ? bc : (int64_t)-1, fc, fbc ? bc : (int64_t)-1));
                    if constexpr (MASS2 == 1) {   // the mass row from the same geometry: what the second sweep would compute for this item, bit for bit
                        const double cm = 1.0 * 1.0 * g.measure;
                        const int il = code & 15;
#pragma unroll
                        for (int j = 0; j < NB; ++j) mval[j] = rounded<R>(cm * tb->mtab[il * NB + j]);
                    }
                }
            }
            // accumulation rounds, one per visit index of the step, ascending.  All items of a row live in THIS wavefront, whose LDS
            // operations complete in program order: a round's read-modify-writes are behind those of the round before, no barrier
            const int32_t rb = rbase_s[q];
            auto rounds = [&](int u0, int u1) {
                for (int u = u0; u < u1 && u < maxv; ++u) {
                    if (on && v == u) {
                        if (MASS || want_matrix) add_row_lds<NB, NBW>(acc + rb, sw, val);
                        if constexpr (!MASS && MASS2 == 1) add_row_lds<NB, NBW>(acc2 + rb, sw, mval);
                        if constexpr (!MASS) facc_s[q] += fval;
                    }
                }
            };
            if (!tail) {
                rounds(v0, v0 + VA);
            } else {
                for (int r = 0; r < GR; ++r) {   // (uniform for the workgroup)
                    if (gr == r) rounds(v0 + VA * r, v0 + VA * (r + 1));
                    __syncthreads();
                }
            }
            v0 += V;
        }
    };
    sweep(std::integral_constant<bool, false>{});
    __syncthreads();
    if (a.force != nullptr && tid < kAsmBlock) {
        const int32_t lr = a.lane_row[row0 + tid];
        if (lr >= 0) a.force[lr] = facc_s[tid];
    }
    if (want_matrix) {
        for (int k = tid; k < blk_nnz; k += THREADS) a.vals[base + k] = acc[k];
        if (a.row_stat != nullptr && tid < kAsmBlock) {
            const int32_t lr = a.lane_row[row0 + tid];
            if (lr >= 0) {
                const int32_t my0 = a.rowptr[lr], my1 = a.rowptr[lr + 1];
                double rmax = 0.0;
                for (int k = my0; k < my1; ++k) rmax = fmax(rmax, fabs(acc[k - base]));
                a.row_stat[2 * (int64_t)lr] = acc[a.diag[lr] - base], a.row_stat[2 * (int64_t)lr + 1] = rmax;
            }
        }
    }
    if constexpr (MASS2 == 1)
        for (int k = tid; k < blk_nnz; k += THREADS) a.vals2[base + k] = acc2[k];
    if constexpr (MASS2 == 2) {
        __syncthreads();
        for (int k = tid; k < blk_nnz; k += THREADS) acc[k] = 0.0;
        __syncthreads();
        sweep(std::integral_constant<bool, true>{});
        __syncthreads();
        for (int k = tid; k < blk_nnz; k += THREADS) a.vals2[base + k] = acc[k];
    }
}

// dst row group b (nq doubles) = src row group idx[b]: forcing samples from the caller's cell order to the internal one
// any row of a diffusion field that is not a symmetric tensor?  (rows x N x N, row-major per quadrature node)
static __global__ __launch_bounds__(256) void k_field_asym(int64_t rows, int N, const double* k, int32_t* flag) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const double* t = k + r * N * N;
    bool asym = false;
    for (int a = 1; a < N; ++a)
        for (int b = 0; b < a; ++b) asym = asym || t[a * N + b] != t[b * N + a];
    if (asym) atomicOr(flag, 1);
}
// What the reference's assembler makes of an expression it takes for symmetric (fem_assembler.h:94-102: only the pairs dof_i >= dof_j -- REFERENCE dof ids --
// are integrated; 116-117: selfadjointView<Lower> mirrors them) when its diffusion tensor is not: entry (i, j) with dof_i < dof_j is the integral of (j, i).
// The sweep has integrated every pair; this pass keeps the reference's half and mirrors it (bitwise symmetric result).  One thread per row; the columns of a
// row are ascending (binary search for the transposed entry).
static __global__ __launch_bounds__(256) void k_mirror_reference_lower(int64_t n, const int32_t* rowptr, const int32_t* colidx, const int32_t* i2e, const double* in,
                                                                       double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t ei = i2e[i];
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int32_t j = colidx[k];
        double v = in[k];
        if (ei < i2e[j]) {
            int32_t lo = rowptr[j], hi = rowptr[j + 1] - 1;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (colidx[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
            }
            if (colidx[lo] == (int32_t)i) v = in[lo];
        }
        out[k] = v;
    }
}
static __global__ __launch_bounds__(256) void k_gather_row_groups(int64_t n, int nq, const int32_t* idx, const double* src, double* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nq) return;
    const int64_t b = i / nq, q = i - b * nq;
    dst[i] = src[(int64_t)idx[b] * nq + q];
}

// Forcing reduced to what a visit needs, in VISIT order (the layout of the adjacency slices): for the visit at slot (s, v, lane)
//   dst[(sl_off[s] + v) * 64 + lane] = sum_q (f[cell * nq + q] * psi_il(p_q)) * w_q        (0 in padding slots)
// -- the (cell, il) entry of the cell's load vector without |e| (integrator.h:73-90), summed in the order element_row uses.  The
// row-owner kernel then streams one coalesced double per visit next to its adjacency word instead of gathering samples by cell id.
// Built once per fdapde_set_forcing; one workgroup of 8 wavefronts per slice, wavefront y takes the visits y, y + 8, ...
static __global__ __launch_bounds__(512) void k_visit_load_coeffs(int64_t n_slices, int nq, const int64_t* sl_off, const int32_t* adj,
                                                          const int64_t* bc_off, const int32_t* bc_cell, const double* src,
                                                          const DevTables* tab, double* dst) {
    const int64_t s = blockIdx.x;
    if (s >= n_slices) return;
    const int lane = threadIdx.x;
    const int64_t off = sl_off[s], width = sl_off[s + 1] - off;
    const int64_t bc0 = bc_off[s / (kAsmBlock / kSlice)];   // block-cell table of the slice's assembly block
    for (int64_t v = threadIdx.y; v < width; v += blockDim.y) {
        const int64_t at = (off + v) * kSlice + lane;
        const int32_t code = adj[at];
        double val = 0;
        if (code >= 0) {
            const double* f = src + (int64_t)bc_cell[bc0 + (code >> 4)] * nq;
            const int il = code & 15;
            for (int q = 0; q < nq; ++q) val += (f[q] * tab->psi[il * nq + q]) * tab->qw[q];
        }
        dst[at] = val;
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
static __global__ __launch_bounds__(256) void k_assemble_scatter(AsmArgs a, DevOp op, const int32_t* cell_list, int64_t n_list) {
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

// ---------------------------------------------------------------------------------------------------------------
// The north-star scatter forms (BASELINE.json: "one wavefront per element ... scatter into a CSR global matrix via
// colour-partitioned passes (atomics only at colour boundaries)"), written out properly and TIMED against the row-owner sweep
// (tools/asm_ab.py -> profiles/r2_asm_ab.json).  Both stream the element -> CSR slot map instead of searching for slots.
//
// k_assemble_part: one workgroup per cell partition (a contiguous chunk of the Morton cell order).  The partition's cells are
//   listed colour by colour (cells of a colour share no DOF); the workgroup walks its colours with a barrier in between, lane =
//   (cell, local row), and adds its row of the element matrix with plain read-modify-writes -- except in rows that cells of
//   another partition touch as well (dof_shared), where it uses fp64 atomics: atomics only on partition boundaries.  The rule
//   is per ROW, so every contribution to one matrix entry takes the same path (an atomic executes at the memory side and would
//   not be seen by a neighbour's cached read-modify-write).  One launch; vals / force zeroed by the caller.
// ---------------------------------------------------------------------------------------------------------------
template <int M, int R, int OPK>
static __global__ __launch_bounds__(256) void k_assemble_part(AsmArgs a, DevOp op, const int32_t* cell_list, const int32_t* colour_off,
                                                        int max_colours, const uint8_t* dof_shared, const int32_t* slot_map) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    extern __shared__ double lds[];
    const DevTables* tb = stage_tables(a.tables, lds);
    const DevRefTensors* rt = nullptr;
    if constexpr (OPK == 3 || OPK == 5) {
        const double* src = reinterpret_cast<const double*>(a.reftab);
        for (int i = threadIdx.x; i < a.ref_doubles; i += blockDim.x) lds[kTablesDoubles + i] = src[i];
        rt = reinterpret_cast<const DevRefTensors*>(lds + kTablesDoubles);
    }
    __syncthreads();
    const int32_t* off = colour_off + (int64_t)blockIdx.x * (max_colours + 1);
    for (int c = 0; c < max_colours; ++c) {
        const int32_t o0 = off[c], o1 = off[c + 1];
        for (int idx = threadIdx.x; idx < (o1 - o0) * NB; idx += 256) {
            const int64_t li = o0 + idx / NB;
            const int il = idx % NB;
            const int cell = cell_list[li];
            const int32_t row = a.cdofs[(int64_t)cell * NB + il];
            const bool shared = dof_shared[row] != 0;
            const int32_t* sm = slot_map + (li * NB + il) * NB;
            Geo<M> g;
            cell_geometry<M>(a, cell, g);
            const double f = element_row<M, R, OPK>(a, op, tb, g, cell, il, a.vals != nullptr, [&](int j, double value) {
                double* dst = a.vals + sm[j];
                if (shared) unsafeAtomicAdd(dst, value);
                else *dst += value;
            }, rt);
            if (a.force != nullptr) {
                if (shared) unsafeAtomicAdd(&a.force[row], f);
                else a.force[row] += f;
            }
        }
        __syncthreads();   // the next colour may touch the rows this one has just written
    }
}

// k_assemble_wave: the literal form -- ONE WAVEFRONT PER ELEMENT, lane = (i, j, q): the 64 (test, trial, quadrature node) triples
//   of a 3-D P1 element (27 of a 2-D one) in one pass; a P2 element (216 triples in 2-D, 500 in 3-D) in passes of 8 (i, j) pairs x 8
//   node lanes.  Every lane evaluates the weak form at its quadrature node, the nodes are summed across lanes, the lanes with q = 0
//   add their entry through the streamed slot map.  Launched once per colour over colour-contiguous cell lists (cells of a colour
//   share no DOF: plain read-modify-write, no atomics).
template <int M, int R>
static __global__ __launch_bounds__(256) void k_assemble_wave(AsmArgs a, DevOp op, const int32_t* cell_list, const int32_t* slot_map,
                                                        int64_t n_list) {
    constexpr int NB = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 10);
    constexpr int NQ = (M == 2) ? (R == 1 ? 3 : 6) : (R == 1 ? 4 : 5);
    constexpr int NQP = NQ <= 4 ? 4 : 8, PP = 64 / NQP;   // node lanes per pair (a power of two), pairs per pass
    extern __shared__ double lds[];
    const DevTables* tb = stage_tables(a.tables, lds);
    __syncthreads();
    const int64_t li = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wavefront per element
    if (li >= n_list) return;
    const int lane = threadIdx.x & 63, q = lane % NQP;
    const int cell = cell_list[li];
    Geo<M> g;
    cell_geometry<M>(a, cell, g);   // the same addresses in every lane: one broadcast fetch per vertex
    for (int p0 = 0; p0 < NB * NB; p0 += PP) {   // (wave-uniform trip count: 1 pass for P1, 5 / 13 for P2 in 2-D / 3-D)
        const int ij = p0 + lane / NQP, i = ij / NB, j = ij - i * NB;
        const bool live = ij < NB * NB && q < NQ;
        double v = 0, fv = 0;
        if (live) {
            double gi[M], gj[M];
            phys_grad<M>(g, &tb->dpsi[(i * NQ + q) * 3], gi);
            phys_grad<M>(g, &tb->dpsi[(j * NQ + q) * 3], gj);
            const int64_t qrow = (int64_t)NQ * cell + q;
            v = weak_form<M>(op, qrow, tb->psi[i * NQ + q], tb->psi[j * NQ + q], gi, gj) * tb->qw[q];
            if (a.fq != nullptr && j == 0) fv = (a.fq[qrow] * tb->psi[i * NQ + q]) * tb->qw[q];
        }
#pragma unroll
        for (int o = 1; o < NQP; o <<= 1) v += __shfl_xor(v, o), fv += __shfl_xor(fv, o);   // sum over the quadrature nodes of a pair
        if (live && q == 0) {
            if (a.vals != nullptr) a.vals[slot_map[(li * NB + i) * NB + j]] += v * g.measure;
            if (a.force != nullptr && j == 0) a.force[a.cdofs[(int64_t)cell * NB + i]] += fv * g.measure;
        }
    }
}

// Integrator::quadrature_nodes (integrator.h:109-121): out row nq*cell_ext + q = J p_q + x0, column-major rows x N
template <int M>
static __global__ void k_quadrature_nodes(AsmArgs a, const int32_t* cell_i2e, int nq, double* out) {
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
static __global__ void k_eval_pointwise(AsmArgs a, int64_t n_locs, const double* locs /*col-major n_locs x M*/, const double* lo,
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
static __global__ void k_cell_integrals(AsmArgs a, int nb, int nq, const int32_t* cell_i2e, double* measure, double* psi_int) {
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

}  // namespace fdapde_hip
#endif
