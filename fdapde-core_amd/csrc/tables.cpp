// tables.cpp -- quadrature rules and reference Lagrange basis values at the quadrature nodes.
//
// Replaces, for the device path, what the reference evaluates again for every (cell, i, j, q):
//   IntegratorTable<M,K>          fdaPDE/utils/integration/integrator_tables.h:146-183, 256-292
//   standard_fem_quadrature_rule  fdaPDE/utils/integration/integrator_tables.h:23-58
//   LagrangianElement<M,R>        fdaPDE/finite_elements/basis/lagrangian_basis.h:39-92
//   ReferenceElement<M,R>::nodes  fdaPDE/finite_elements/basis/reference_element.h:50-66, 83-97
// The reference obtains the basis by a Vandermonde solve; here the same polynomials are written in closed form in
// barycentric coordinates (lambda_0 = 1 - sum xi, lambda_k = xi_k):  P1: lambda_k;  P2 vertex: lambda (2 lambda - 1),
// P2 edge (a,b): 4 lambda_a lambda_b, in the reference's local node order.  The quadrature constants are data and
// are replicated digit for digit (weights sum to 0.999999999999999 in the 3-point rule; that is part of the
// contract, integrator_tables.h:158-160).
#include <cstring>

#include "internal.h"

namespace fdapde_hip {

int n_basis_of(int M, int R) {
    if (M == 2) return R == 1 ? 3 : 6;
    if (M == 3) return R == 1 ? 4 : 10;
    return 0;
}
int n_quadrature_of(int M, int R) {
    if (M == 2) return R == 1 ? 3 : 6;
    if (M == 3) return R == 1 ? 4 : 5;
    return 0;
}

namespace {
const double Q23n[] = {0.166666666666667, 0.166666666666667, 0.666666666666667,
                       0.166666666666667, 0.166666666666667, 0.666666666666667};
const double Q23w[] = {0.333333333333333, 0.333333333333333, 0.333333333333333};
const double Q26n[] = {0.445948490915965, 0.445948490915965, 0.445948490915965, 0.108103018168070,
                       0.108103018168070, 0.445948490915965, 0.091576213509771, 0.091576213509771,
                       0.091576213509771, 0.816847572980459, 0.816847572980459, 0.091576213509771};
const double Q26w[] = {0.223381589678011, 0.223381589678011, 0.223381589678011,
                       0.109951743655322, 0.109951743655322, 0.109951743655322};
const double Q34n[] = {0.585410196624969, 0.138196601125011, 0.138196601125011, 0.138196601125011,
                       0.138196601125011, 0.138196601125011, 0.138196601125011, 0.138196601125011,
                       0.585410196624969, 0.138196601125011, 0.585410196624969, 0.138196601125011};
const double Q34w[] = {0.250000000000000, 0.250000000000000, 0.250000000000000, 0.250000000000000};
const double Q35n[] = {0.250000000000000, 0.250000000000000, 0.250000000000000, 0.500000000000000, 0.166666666666667,
                       0.166666666666667, 0.166666666666667, 0.500000000000000, 0.166666666666667, 0.166666666666667,
                       0.166666666666667, 0.500000000000000, 0.166666666666667, 0.166666666666667, 0.166666666666667};
const double Q35w[] = {-0.80000000000000, 0.450000000000000, 0.450000000000000, 0.450000000000000, 0.450000000000000};

// local edge slot -> its two local vertices, in the reference element's node order
// 2-D (reference_element.h:60-62): nodes 3,4,5 = (.5,0), (0,.5), (.5,.5)            -> (0,1), (0,2), (1,2)
// 3-D (reference_element.h:93-96): nodes 4..9 = m12, m02, m01, m13, m23, m03
const int EDGE2[3][2] = {{0, 1}, {0, 2}, {1, 2}};
const int EDGE3[6][2] = {{1, 2}, {0, 2}, {0, 1}, {1, 3}, {2, 3}, {0, 3}};
}  // namespace

int build_basis_tables(int M, int R, BasisTables* t) {
    *t = BasisTables{};
    if ((M != 2 && M != 3) || (R != 1 && R != 2)) return FDAPDE_EUNSUPPORTED;
    t->M = M, t->R = R, t->nb = n_basis_of(M, R), t->nq = n_quadrature_of(M, R);
    const double *qn, *qw;
    if (M == 2)
        qn = R == 1 ? Q23n : Q26n, qw = R == 1 ? Q23w : Q26w;
    else
        qn = R == 1 ? Q34n : Q35n, qw = R == 1 ? Q34w : Q35w;
    std::memcpy(t->qn, qn, sizeof(double) * t->nq * M);
    std::memcpy(t->qw, qw, sizeof(double) * t->nq);
    const int nv = M + 1;
    // reference coordinates of the local DOFs
    for (int v = 0; v < nv; ++v)
        for (int k = 0; k < M; ++k) t->refnodes[v * M + k] = (v == k + 1) ? 1.0 : 0.0;
    for (int s = nv; s < t->nb; ++s) {
        const int* e = M == 2 ? EDGE2[s - nv] : EDGE3[s - nv];
        for (int k = 0; k < M; ++k) t->refnodes[s * M + k] = 0.5 * (t->refnodes[e[0] * M + k] + t->refnodes[e[1] * M + k]);
    }
    for (int q = 0; q < t->nq; ++q) {
        double lam[4], dlam[4][3];
        lam[0] = 1.0;
        for (int k = 0; k < M; ++k) lam[0] -= qn[q * M + k], lam[k + 1] = qn[q * M + k];
        for (int v = 0; v < nv; ++v)
            for (int k = 0; k < M; ++k) dlam[v][k] = v == 0 ? -1.0 : (v == k + 1 ? 1.0 : 0.0);
        for (int i = 0; i < t->nb; ++i) {
            double val, d[3] = {0, 0, 0};
            if (R == 1) {
                val = lam[i];
                for (int k = 0; k < M; ++k) d[k] = dlam[i][k];
            } else if (i < nv) {
                val = lam[i] * (2.0 * lam[i] - 1.0);
                for (int k = 0; k < M; ++k) d[k] = (4.0 * lam[i] - 1.0) * dlam[i][k];
            } else {
                const int* e = M == 2 ? EDGE2[i - nv] : EDGE3[i - nv];
                val = 4.0 * lam[e[0]] * lam[e[1]];
                for (int k = 0; k < M; ++k) d[k] = 4.0 * (lam[e[0]] * dlam[e[1]][k] + lam[e[1]] * dlam[e[0]][k]);
            }
            t->psi[i * t->nq + q] = val;
            for (int k = 0; k < 3; ++k) t->dpsi[(i * t->nq + q) * 3 + k] = d[k];
        }
    }
    return FDAPDE_OK;
}

}  // namespace fdapde_hip
