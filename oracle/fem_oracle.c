/*
 * fem_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A dependency-free, single-threaded C restatement of the fdaPDE-core finite-element
 * assemble-and-solve path (Assembler<FEM,...> + FEMSolverBase + elliptic solve).  It exists so that
 * the HIP path can be checked against the reference's algorithm on a machine where the reference
 * itself cannot be built (it needs Eigen 3.4, which is not in this image).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (fdapde-core_amd/, include/) never links, imports or calls it.
 *
 * Parity status: PINNED by the reference's own golden vectors (tests/test_oracle_golden.py):
 *   - 6x6 local P2 stiffness of c_shaped cell 175       (test/src/fem_operators_test.cpp:83-96)
 *   - physical P1/P2 gradients on that cell             (test/src/lagrangian_basis_test.cpp:158-161,185-187)
 *   - reference P1/P2 gradients                         (test/src/lagrangian_basis_test.cpp:114,133-140)
 *   - 4 golden Psi matrices (.mtx) -> P2 DOF numbering  (test/src/lagrangian_basis_test.cpp:200-238)
 *   - analytic-solution PDE tests                       (test/src/fem_pde_test.cpp:43-212)
 *   - quadrature identities, tetrahedron measure        (test/src/integration_test.cpp:46-126, simplex_test.cpp:89-97)
 * UNPINNED: 3-D P2 global edge-DOF numbering (the reference does not compile for <3,3> order 2;
 *   lagrangian_basis.h:111-123 uses members Triangulation<3,3> lacks).  The rule used here extends the
 *   reference's own 3-D edge ids (geometry/triangulation.h:348-377) and is documented in DESIGN.md.
 *
 * Every function cites the reference file:line whose behaviour it restates (paths relative to
 * /root/reference/fdaPDE unless they start with test/).  No reference source text is copied; quadrature
 * constants are data replicated digit-for-digit because they are the numerical contract.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FO_MAXB 10 /* max local dofs: 3-D P2 */
#define FO_MAXQ 6  /* max quadrature nodes used by standard_fem_quadrature_rule for R<=2 */

#define FO_OK 0
#define FO_EINVAL 1
#define FO_ENOMEM 2
#define FO_ESINGULAR 3
#define FO_ENOCONV 4

/* ------------------------------------------------------------------------------------------------
 * operator description: a left-to-right sum of scaled leaf operators.  The reference builds an
 * expression tree (pde/differential_expressions.h:49,95-96,114-118); unary minus, scalar* and binary
 * +/- of leaves collapse to  sum_k coef_k * leaf_k  evaluated left to right.
 * ---------------------------------------------------------------------------------------------- */
enum { FO_LAPLACIAN = 0, FO_DIFFUSION = 1, FO_ADVECTION = 2, FO_REACTION = 3, FO_DT = 4 };

typedef struct {
    int32_t kind;          /* FO_* */
    int32_t space_varying; /* 1: coefficient comes from data[] (fields/{scalar,vector,matrix}_expressions.h forward(i)) */
    double coef;           /* scalar multiplier */
    double cst[9];         /* constant coefficient: K row-major NxN | b[N] | c */
    const double *data;    /* space-varying: row-major (nq*n_cells) x (N*N | N | 1), row = nq*cell + q */
} fo_term;

/* ------------------------------------------------------------------------------------------------
 * quadrature: utils/integration/integrator_tables.h
 * ---------------------------------------------------------------------------------------------- */
/* standard_fem_quadrature_rule<M,R>::K (integrator_tables.h:23-58) */
int fo_quadrature_rule(int M, int R) {
    switch (M) {
    case 1: return R == 1 ? 2 : 3;
    case 2: return R == 1 ? 3 : (R == 2 ? 6 : 12);
    case 3: return R == 1 ? 4 : 5;
    }
    return 0;
}

/* IntegratorTable<M,nq> nodes (first M barycentric coords) and weights (sum 1).
 * <2,3>: integrator_tables.h:146-161; <2,6>: 164-183; <3,4>: 256-272; <3,5>: 275-292; <3,11>: 295-320 */
int fo_quadrature_table(int M, int nq, double *nodes, double *weights) {
    if (M == 2 && nq == 3) {
        static const double n[] = {0.166666666666667, 0.166666666666667, 0.666666666666667,
                                   0.166666666666667, 0.166666666666667, 0.666666666666667};
        static const double w[] = {0.333333333333333, 0.333333333333333, 0.333333333333333};
        memcpy(nodes, n, sizeof n), memcpy(weights, w, sizeof w);
        return FO_OK;
    }
    if (M == 2 && nq == 6) {
        static const double n[] = {0.445948490915965, 0.445948490915965, 0.445948490915965, 0.108103018168070,
                                   0.108103018168070, 0.445948490915965, 0.091576213509771, 0.091576213509771,
                                   0.091576213509771, 0.816847572980459, 0.816847572980459, 0.091576213509771};
        static const double w[] = {0.223381589678011, 0.223381589678011, 0.223381589678011,
                                   0.109951743655322, 0.109951743655322, 0.109951743655322};
        memcpy(nodes, n, sizeof n), memcpy(weights, w, sizeof w);
        return FO_OK;
    }
    if (M == 3 && nq == 4) {
        static const double n[] = {0.585410196624969, 0.138196601125011, 0.138196601125011, 0.138196601125011,
                                   0.138196601125011, 0.138196601125011, 0.138196601125011, 0.138196601125011,
                                   0.585410196624969, 0.138196601125011, 0.585410196624969, 0.138196601125011};
        static const double w[] = {0.250000000000000, 0.250000000000000, 0.250000000000000, 0.250000000000000};
        memcpy(nodes, n, sizeof n), memcpy(weights, w, sizeof w);
        return FO_OK;
    }
    if (M == 3 && nq == 5) {
        static const double n[] = {0.250000000000000, 0.250000000000000, 0.250000000000000, 0.500000000000000,
                                   0.166666666666667, 0.166666666666667, 0.166666666666667, 0.500000000000000,
                                   0.166666666666667, 0.166666666666667, 0.166666666666667, 0.500000000000000,
                                   0.166666666666667, 0.166666666666667, 0.166666666666667};
        static const double w[] = {-0.80000000000000, 0.450000000000000, 0.450000000000000, 0.450000000000000,
                                   0.450000000000000};
        memcpy(nodes, n, sizeof n), memcpy(weights, w, sizeof w);
        return FO_OK;
    }
    if (M == 3 && nq == 11) {
        static const double n[] = {
          0.2500000000000000, 0.2500000000000000, 0.2500000000000000, 0.7857142857142857, 0.0714285714285714,
          0.0714285714285714, 0.0714285714285714, 0.0714285714285714, 0.0714285714285714, 0.0714285714285714,
          0.0714285714285714, 0.7857142857142857, 0.0714285714285714, 0.7857142857142857, 0.0714285714285714,
          0.1005964238332008, 0.3994035761667992, 0.3994035761667992, 0.3994035761667992, 0.1005964238332008,
          0.3994035761667992, 0.3994035761667992, 0.3994035761667992, 0.1005964238332008, 0.3994035761667992,
          0.1005964238332008, 0.1005964238332008, 0.1005964238332008, 0.3994035761667992, 0.1005964238332008,
          0.1005964238332008, 0.1005964238332008, 0.3994035761667992};
        static const double w[] = {-0.0789333333333333, 0.0457333333333333, 0.0457333333333333, 0.0457333333333333,
                                   0.0457333333333333,  0.1493333333333333, 0.1493333333333333, 0.1493333333333333,
                                   0.1493333333333333,  0.1493333333333333, 0.1493333333333333};
        memcpy(nodes, n, sizeof n), memcpy(weights, w, sizeof w);
        return FO_OK;
    }
    return FO_EINVAL;
}

/* ------------------------------------------------------------------------------------------------
 * reference element + Lagrangian basis
 * ---------------------------------------------------------------------------------------------- */
static int binom(int n, int k) {
    int r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}
/* ct_nnodes / n_basis = C(M+R, R) (finite_elements/basis/lagrangian_basis.h:47) */
int fo_n_basis(int M, int R) { return binom(M + R, R); }

/* ReferenceElement<M,R>::nodes (finite_elements/basis/reference_element.h:50-66,83-97) */
int fo_reference_nodes(int M, int R, double *nodes) {
    static const double n21[] = {0, 0, 1, 0, 0, 1};
    static const double n22[] = {0, 0, 1, 0, 0, 1, 0.5, 0, 0, 0.5, 0.5, 0.5};
    static const double n31[] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    static const double n32[] = {0, 0,   0, 1,   0, 0, 0,   1, 0,   0,   0,   1,   0.5, 0.5, 0,
                                 0, 0.5, 0, 0.5, 0, 0, 0.5, 0, 0.5, 0,   0.5, 0.5, 0,   0,   0.5};
    const double *src = 0;
    if (M == 2 && R == 1) src = n21;
    if (M == 2 && R == 2) src = n22;
    if (M == 3 && R == 1) src = n31;
    if (M == 3 && R == 2) src = n32;
    if (!src) return FO_EINVAL;
    memcpy(nodes, src, sizeof(double) * (size_t)fo_n_basis(M, R) * M);
    return FO_OK;
}

/* ct_poly_exp<N,R>() monomial exponent table (basis/multivariate_polynomial.h:52-79): a mixed-radix
 * counter whose digit 0 runs fastest, rows with total degree > R skipped. */
static int poly_table(int N, int R, int table[][3]) {
    int tmp[4] = {0, 0, 0, 0}, j = 0;
    const int n_mono = binom(N + R, R);
    while (j < n_mono) {
        int i = 0, found = 0;
        while (i < N && !found) {
            int sum = 0;
            for (int k = 0; k < N; ++k) sum += tmp[k];
            if (tmp[i] <= R && sum <= R) {
                found = 1;
                for (int k = 0; k < 3; ++k) table[j][k] = k < N ? tmp[k] : 0;
                ++tmp[0], ++j;
            } else { /* propagate the carry to the next digit */
                tmp[i] = 0;
                ++tmp[++i];
            }
        }
    }
    return n_mono;
}
int fo_poly_table(int N, int R, int32_t *out) {
    int t[FO_MAXB][3];
    int n = poly_table(N, R, t);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < N; ++k) out[j * N + k] = t[j][k];
    return n;
}

/* MonomialProduct::unfold (multivariate_polynomial.h:125-134): prod_k p[k]^e[k] with pow() */
static double monomial(int N, const double *p, const int *e) {
    double v = e[0] == 0 ? 1.0 : pow(p[0], e[0]);
    for (int k = 1; k < N; ++k)
        if (e[k] != 0) v = pow(p[k], e[k]) * v;
    return v;
}

/* LagrangianElement::compute_coefficients_ (lagrangian_basis.h:65-91): Vandermonde V(i,j) = monomial_j(node_i),
 * solve V a = e_i with partial-pivot LU.  coeff[i*n + m] = coefficient of monomial m in basis function i. */
int fo_reference_basis(int M, int R, double *coeff) {
    const int n = fo_n_basis(M, R);
    double nodes[FO_MAXB * 3], V[FO_MAXB][FO_MAXB];
    int tab[FO_MAXB][3], piv[FO_MAXB];
    if (fo_reference_nodes(M, R, nodes) != FO_OK) return FO_EINVAL;
    poly_table(M, R, tab);
    for (int i = 0; i < n; ++i) {
        V[i][0] = 1.0;
        for (int j = 1; j < n; ++j) V[i][j] = monomial(M, &nodes[i * M], tab[j]);
    }
    for (int k = 0; k < n; ++k) { /* in-place LU with row pivoting */
        int p = k;
        for (int i = k + 1; i < n; ++i)
            if (fabs(V[i][k]) > fabs(V[p][k])) p = i;
        piv[k] = p;
        if (V[p][k] == 0.0) return FO_ESINGULAR;
        if (p != k)
            for (int j = 0; j < n; ++j) {
                double t = V[k][j];
                V[k][j] = V[p][j], V[p][j] = t;
            }
        for (int i = k + 1; i < n; ++i) {
            V[i][k] /= V[k][k];
            for (int j = k + 1; j < n; ++j) V[i][j] -= V[i][k] * V[k][j];
        }
    }
    for (int b = 0; b < n; ++b) {
        double x[FO_MAXB];
        for (int i = 0; i < n; ++i) x[i] = (i == b) ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k)
            if (piv[k] != k) {
                double t = x[k];
                x[k] = x[piv[k]], x[piv[k]] = t;
            }
        for (int i = 1; i < n; ++i)
            for (int j = 0; j < i; ++j) x[i] -= V[i][j] * x[j];
        for (int i = n - 1; i >= 0; --i) {
            for (int j = i + 1; j < n; ++j) x[i] -= V[i][j] * x[j];
            x[i] /= V[i][i];
        }
        for (int m = 0; m < n; ++m) coeff[b * n + m] = x[m];
    }
    return FO_OK;
}

/* MultivariatePolynomial::operator() (multivariate_polynomial.h:209-213, MonomialSum 139-155):
 * c0*m0 summed first, higher monomials added on the left */
double fo_poly_eval(int M, int R, const double *c, const double *p) {
    int tab[FO_MAXB][3];
    const int n = poly_table(M, R, tab);
    double v = c[0] * monomial(M, p, tab[0]);
    for (int m = 1; m < n; ++m) v = c[m] * monomial(M, p, tab[m]) + v;
    return v;
}
/* PolynomialDerivative::operator() (multivariate_polynomial.h:172-182) for every direction */
void fo_poly_grad(int M, int R, const double *c, const double *p, double *g) {
    int tab[FO_MAXB][3];
    const int n = poly_table(M, R, tab);
    for (int d = 0; d < M; ++d) {
        double v = 0;
        for (int m = 0; m < n; ++m) {
            if (tab[m][d] != 0) {
                int e[3] = {tab[m][0], tab[m][1], tab[m][2]};
                e[d] -= 1;
                v += c[m] * tab[m][d] * monomial(M, p, e);
            }
        }
        g[d] = v;
    }
}

/* psi_i(p_q), grad psi_i(p_q) for the standard rule of (M,R): what the reference re-evaluates for every
 * (cell,i,j,q) (fem_assembler.h:88-93, integrator.h:95-102); the numbers do not depend on the cell. */
typedef struct {
    int M, R, nb, nq;
    double qn[FO_MAXQ * 3], qw[FO_MAXQ];
    double psi[FO_MAXB][FO_MAXQ];
    double dpsi[FO_MAXB][FO_MAXQ][3];
    double refnodes[FO_MAXB * 3];
} fo_tables;

static int build_tables(int M, int R, fo_tables *t) {
    double coeff[FO_MAXB * FO_MAXB];
    memset(t, 0, sizeof *t);
    t->M = M, t->R = R, t->nb = fo_n_basis(M, R), t->nq = fo_quadrature_rule(M, R);
    if (t->nb > FO_MAXB || t->nq > FO_MAXQ || M < 2 || M > 3 || R < 1 || R > 2) return FO_EINVAL;
    if (fo_quadrature_table(M, t->nq, t->qn, t->qw) != FO_OK) return FO_EINVAL;
    if (fo_reference_basis(M, R, coeff) != FO_OK) return FO_ESINGULAR;
    fo_reference_nodes(M, R, t->refnodes);
    for (int i = 0; i < t->nb; ++i)
        for (int q = 0; q < t->nq; ++q) {
            t->psi[i][q] = fo_poly_eval(M, R, &coeff[i * t->nb], &t->qn[q * M]);
            fo_poly_grad(M, R, &coeff[i * t->nb], &t->qn[q * M], t->dpsi[i][q]);
        }
    return FO_OK;
}
/* export for tests: psi (nb x nq) and dpsi (nb x nq x M) */
int fo_basis_tables(int M, int R, double *psi, double *dpsi) {
    fo_tables t;
    int rc = build_tables(M, R, &t);
    if (rc) return rc;
    for (int i = 0; i < t.nb; ++i)
        for (int q = 0; q < t.nq; ++q) {
            psi[i * t.nq + q] = t.psi[i][q];
            for (int d = 0; d < M; ++d) dpsi[(i * t.nq + q) * M + d] = t.dpsi[i][q][d];
        }
    return FO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * per-cell geometry: Simplex::initialize (geometry/simplex.h:184-195), local_dim == embed_dim only
 *   J col j = x_{j+1} - x_0 ; invJ = J^{-1} ; measure = |det J| / M!
 * nodes are column-major n_nodes x N (geometry/triangulation.h:119), cells row-major (120)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    double x0[3], J[3][3], invJ[3][3], measure;
} fo_geom;

static void cell_geometry(int M, int64_t n_nodes, const double *nodes, const int32_t *cell, fo_geom *g) {
    for (int d = 0; d < M; ++d) g->x0[d] = nodes[(int64_t)d * n_nodes + cell[0]];
    for (int j = 0; j < M; ++j)
        for (int d = 0; d < M; ++d) g->J[d][j] = nodes[(int64_t)d * n_nodes + cell[j + 1]] - g->x0[d];
    if (M == 2) {
        double det = g->J[0][0] * g->J[1][1] - g->J[0][1] * g->J[1][0];
        double id = 1.0 / det;
        g->invJ[0][0] = g->J[1][1] * id, g->invJ[0][1] = -g->J[0][1] * id;
        g->invJ[1][0] = -g->J[1][0] * id, g->invJ[1][1] = g->J[0][0] * id;
        g->measure = fabs(det) / 2.0;
    } else {
        const double(*a)[3] = g->J;
        double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
        double c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2];
        double c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
        double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
        double id = 1.0 / det;
        g->invJ[0][0] = c00 * id;
        g->invJ[1][0] = c01 * id;
        g->invJ[2][0] = c02 * id;
        g->invJ[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
        g->invJ[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id;
        g->invJ[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
        g->invJ[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id;
        g->invJ[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
        g->invJ[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
        g->measure = fabs(det) / 6.0;
    }
}
/* export: measures of all cells, and J/invJ of one cell (row-major MxM) */
int fo_cell_geometry(int M, int64_t n_nodes, const double *nodes, const int32_t *cell, double *J, double *invJ,
                     double *measure) {
    fo_geom g;
    if (M < 2 || M > 3) return FO_EINVAL;
    cell_geometry(M, n_nodes, nodes, cell, &g);
    for (int r = 0; r < M; ++r)
        for (int c = 0; c < M; ++c) J[r * M + c] = g.J[r][c], invJ[r * M + c] = g.invJ[r][c];
    *measure = g.measure;
    return FO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * small open-addressing hash map keyed by up to three ints (stands in for the reference's
 * std::unordered_map<std::array<int,K>, ...>; only find/insert/erase semantics matter)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t k[3];
    int32_t v0, v1;
    int8_t state; /* 0 empty, 1 used, 2 erased */
} hm_slot;
typedef struct {
    hm_slot *s;
    uint64_t cap, used;
} hmap;
static uint64_t hm_hash(const int32_t *k) {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 3; ++i) h = (h ^ (uint32_t)k[i]) * 1099511628211ull, h ^= h >> 29;
    return h;
}
static int hm_init(hmap *m, uint64_t expect) {
    m->cap = 16;
    while (m->cap < 2 * expect + 16) m->cap <<= 1;
    m->used = 0;
    m->s = (hm_slot *)calloc(m->cap, sizeof(hm_slot));
    return m->s ? FO_OK : FO_ENOMEM;
}
static hm_slot *hm_find(hmap *m, const int32_t *k) {
    uint64_t i = hm_hash(k) & (m->cap - 1);
    for (;;) {
        hm_slot *s = &m->s[i];
        if (s->state == 0) return 0;
        if (s->state == 1 && s->k[0] == k[0] && s->k[1] == k[1] && s->k[2] == k[2]) return s;
        i = (i + 1) & (m->cap - 1);
    }
}
static hm_slot *hm_insert(hmap *m, const int32_t *k, int32_t v0, int32_t v1) {
    uint64_t i = hm_hash(k) & (m->cap - 1);
    while (m->s[i].state == 1) i = (i + 1) & (m->cap - 1);
    hm_slot *s = &m->s[i];
    s->k[0] = k[0], s->k[1] = k[1], s->k[2] = k[2], s->v0 = v0, s->v1 = v1, s->state = 1;
    ++m->used;
    return s;
}
static void sort_small(int32_t *a, int n) {
    for (int i = 1; i < n; ++i) {
        int32_t x = a[i];
        int j = i - 1;
        for (; j >= 0 && a[j] > x; --j) a[j + 1] = a[j];
        a[j + 1] = x;
    }
}

/* combinations<K,N>() row order (utils/combinatorics.h:37-51, std::prev_permutation on a K-ones bitmask) */
static const int COMB23[3][2] = {{0, 1}, {0, 2}, {1, 2}};
static const int COMB34[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};

/* Triangulation<2,N> constructor edge discovery (geometry/triangulation.h:150-193): cells ascending x
 * combinations<2,3>, first-seen numbering; an edge met twice is interior (187) and leaves the map (190).
 * edges: n_edges x 2 sorted node ids; edge_boundary: 1 iff seen once; cell_to_edges n_cells x 3. */
static int edges_2d(int64_t n_cells, const int32_t *cells, int32_t **edges_out, uint8_t **bnd_out,
                    int32_t *cell_to_edges, int32_t *n_edges_out) {
    hmap m;
    int64_t cap = 3 * n_cells;
    int32_t *edges = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(cap ? cap : 1));
    uint8_t *bnd = (uint8_t *)malloc((size_t)(cap ? cap : 1));
    if (!edges || !bnd || hm_init(&m, (uint64_t)cap)) return FO_ENOMEM;
    int32_t ne = 0;
    for (int64_t i = 0; i < n_cells; ++i)
        for (int j = 0; j < 3; ++j) {
            int32_t e[3] = {cells[3 * i + COMB23[j][0]], cells[3 * i + COMB23[j][1]], 0};
            sort_small(e, 2);
            hm_slot *s = hm_find(&m, e);
            if (!s) {
                edges[2 * ne] = e[0], edges[2 * ne + 1] = e[1], bnd[ne] = 1;
                hm_insert(&m, e, ne, (int32_t)i);
                cell_to_edges[3 * i + j] = ne++;
            } else {
                cell_to_edges[3 * i + j] = s->v0, bnd[s->v0] = 0;
                s->state = 2;
            }
        }
    free(m.s);
    *edges_out = edges, *bnd_out = bnd, *n_edges_out = ne;
    return FO_OK;
}

/* Triangulation<3,3> constructor (geometry/triangulation.h:348-388): cells ascending x combinations<3,4> faces
 * (sorted nodes); for every NEWLY seen face, its combinations<2,3> edges (of the sorted face) get first-seen ids;
 * edge boundary marker = both end nodes are boundary nodes (371).  cell_to_edges n_cells x 6 in the order of
 * (a,b) local vertex pairs (0,1),(0,2),(0,3),(1,2),(1,3),(2,3). */
static int edges_3d(int64_t n_cells, const int32_t *cells, const uint8_t *node_bnd, int32_t **edges_out,
                    uint8_t **bnd_out, int32_t *cell_to_edges, int32_t *n_edges_out, int32_t *n_faces_out) {
    hmap fm, em;
    int64_t ecap = 6 * n_cells;
    int32_t *edges = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(ecap ? ecap : 1));
    uint8_t *bnd = (uint8_t *)malloc((size_t)(ecap ? ecap : 1));
    if (!edges || !bnd || hm_init(&fm, (uint64_t)(4 * n_cells)) || hm_init(&em, (uint64_t)ecap)) return FO_ENOMEM;
    int32_t ne = 0, nf = 0;
    for (int64_t i = 0; i < n_cells; ++i)
        for (int j = 0; j < 4; ++j) {
            int32_t f[3] = {cells[4 * i + COMB34[j][0]], cells[4 * i + COMB34[j][1]], cells[4 * i + COMB34[j][2]]};
            sort_small(f, 3);
            hm_slot *s = hm_find(&fm, f);
            if (!s) {
                hm_insert(&fm, f, nf++, (int32_t)i);
                for (int k = 0; k < 3; ++k) {
                    int32_t e[3] = {f[COMB23[k][0]], f[COMB23[k][1]], 0};
                    sort_small(e, 2);
                    if (!hm_find(&em, e)) {
                        edges[2 * ne] = e[0], edges[2 * ne + 1] = e[1];
                        bnd[ne] = (uint8_t)(node_bnd[e[0]] && node_bnd[e[1]]);
                        hm_insert(&em, e, ne++, 0);
                    }
                }
            } else {
                s->state = 2; /* a face is shared by at most two cells (faces_map.erase, 386) */
            }
        }
    static const int P[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    for (int64_t i = 0; i < n_cells; ++i)
        for (int k = 0; k < 6; ++k) {
            int32_t e[3] = {cells[4 * i + P[k][0]], cells[4 * i + P[k][1]], 0};
            sort_small(e, 2);
            cell_to_edges[6 * i + k] = hm_find(&em, e)->v0;
        }
    free(fm.s), free(em.s);
    *edges_out = edges, *bnd_out = bnd, *n_edges_out = ne, *n_faces_out = nf;
    return FO_OK;
}

/* local slot (>= M+1) of the edge joining local vertices (a,b), a<b.
 * 2-D: slot 3 + index in combinations<2,3> (lagrangian_basis.h:105-128): m01,m02,m12 = ReferenceElement<2,2> nodes 3,4,5.
 * 3-D (build-defined, F7): the ReferenceElement<3,2> node (reference_element.h:93-96) at the midpoint of (a,b):
 *   nodes 4..9 = m12, m02, m01, m13, m23, m03. */
static int edge_slot(int M, int a, int b) {
    if (M == 2) return 3 + (a == 0 ? (b == 1 ? 0 : 1) : 2);
    static const int S[4][4] = {{-1, 6, 5, 9}, {6, -1, 4, 7}, {5, 4, -1, 8}, {9, 7, 8, -1}};
    return S[a][b];
}

/* LagrangianBasis::enumerate_dofs (lagrangian_basis.h:94-136).
 * order 1: dofs = cells, boundary = node markers.  order 2: dof = n_nodes + edge_id in slot edge_slot().
 * dofs row-major n_cells x n_basis; boundary_dofs has room for n_nodes + 6*n_cells flags. */
int fo_enumerate_dofs(int M, int order, int64_t n_nodes, int64_t n_cells, const int32_t *cells,
                      const uint8_t *node_bnd, int32_t *dofs, uint8_t *boundary_dofs, int32_t *n_dofs_out,
                      int32_t *n_edges_out) {
    const int nv = M + 1, nb = fo_n_basis(M, order);
    if (M < 2 || M > 3 || order < 1 || order > 2) return FO_EINVAL;
    for (int64_t i = 0; i < n_cells; ++i)
        for (int v = 0; v < nv; ++v) dofs[i * nb + v] = cells[i * nv + v];
    for (int64_t i = 0; i < n_nodes; ++i) boundary_dofs[i] = node_bnd[i] ? 1 : 0;
    if (order == 1) {
        *n_dofs_out = (int32_t)n_nodes, *n_edges_out = 0;
        return FO_OK;
    }
    const int epc = M == 2 ? 3 : 6;
    int32_t *c2e = (int32_t *)malloc(sizeof(int32_t) * (size_t)epc * (size_t)(n_cells ? n_cells : 1));
    int32_t *edges = 0, ne = 0, nf = 0;
    uint8_t *ebnd = 0;
    if (!c2e) return FO_ENOMEM;
    int rc = M == 2 ? edges_2d(n_cells, cells, &edges, &ebnd, c2e, &ne)
                    : edges_3d(n_cells, cells, node_bnd, &edges, &ebnd, c2e, &ne, &nf);
    if (rc) return rc;
    static const int P3[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    for (int64_t i = 0; i < n_cells; ++i)
        for (int k = 0; k < epc; ++k) {
            int a = M == 2 ? COMB23[k][0] : P3[k][0], b = M == 2 ? COMB23[k][1] : P3[k][1];
            dofs[i * nb + edge_slot(M, a, b)] = (int32_t)n_nodes + c2e[i * epc + k];
        }
    for (int32_t e = 0; e < ne; ++e) boundary_dofs[n_nodes + e] = ebnd[e];
    *n_dofs_out = (int32_t)n_nodes + ne, *n_edges_out = ne;
    free(c2e), free(edges), free(ebnd);
    return FO_OK;
}

/* LagrangianBasis::dofs_coords (lagrangian_basis.h:159-183): vertices, then the first visiting cell maps the
 * reference node of each extra slot: J * ref + x0.  coords column-major n_dofs x M. */
int fo_dofs_coords(int M, int order, int64_t n_nodes, const double *nodes, int64_t n_cells, const int32_t *cells,
                   const int32_t *dofs, int64_t n_dofs, double *coords) {
    const int nb = fo_n_basis(M, order), nv = M + 1;
    double ref[FO_MAXB * 3];
    if (fo_reference_nodes(M, order, ref)) return FO_EINVAL;
    for (int d = 0; d < M; ++d)
        for (int64_t i = 0; i < n_nodes; ++i) coords[d * n_dofs + i] = nodes[d * n_nodes + i];
    if (order == 1) return FO_OK;
    uint8_t *seen = (uint8_t *)calloc((size_t)n_dofs, 1);
    if (!seen) return FO_ENOMEM;
    for (int64_t c = 0; c < n_cells; ++c) {
        fo_geom g;
        int need = 0;
        for (int j = nv; j < nb; ++j) need |= !seen[dofs[c * nb + j]];
        if (!need) continue;
        cell_geometry(M, n_nodes, nodes, &cells[c * nv], &g);
        for (int j = nv; j < nb; ++j) {
            int32_t dof = dofs[c * nb + j];
            if (seen[dof]) continue;
            seen[dof] = 1;
            for (int d = 0; d < M; ++d) {
                double v = 0;
                for (int k = 0; k < M; ++k) v += g.J[d][k] * ref[j * M + k];
                coords[d * n_dofs + dof] = v + g.x0[d];
            }
        }
    }
    free(seen);
    return FO_OK;
}

/* Integrator::quadrature_nodes (utils/integration/integrator.h:109-121): row nq*cell + q = J p_q + x0,
 * output column-major (nq*n_cells) x M */
int fo_quadrature_nodes(int M, int R, int64_t n_nodes, const double *nodes, int64_t n_cells, const int32_t *cells,
                        double *out) {
    fo_tables t;
    int rc = build_tables(M, R, &t);
    if (rc) return rc;
    const int64_t rows = (int64_t)t.nq * n_cells;
    for (int64_t c = 0; c < n_cells; ++c) {
        fo_geom g;
        cell_geometry(M, n_nodes, nodes, &cells[c * (M + 1)], &g);
        for (int q = 0; q < t.nq; ++q)
            for (int d = 0; d < M; ++d) {
                double v = 0;
                for (int k = 0; k < M; ++k) v += g.J[d][k] * t.qn[q * M + k];
                out[d * rows + t.nq * c + q] = v + g.x0[d];
            }
    }
    return FO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * weak forms at one quadrature node (finite_elements/operators/{laplacian,diffusion,advection,reaction,dt}.h).  gi/gj = J^{-T} grad psi (physical
 * gradients): buff_invJ = invJ^T (fem_assembler.h:81) applied to the reference gradient.
 * ---------------------------------------------------------------------------------------------- */
static inline double weak_form_at(int M, int n_terms, const fo_term *terms, int64_t qrow, double psi_i, double psi_j,
                                  const double *gi, const double *gj) {
    double total = 0;
    for (int t = 0; t < n_terms; ++t) {
        const fo_term *T = &terms[t];
        double v = 0;
        switch (T->kind) {
        case FO_LAPLACIAN: { /* laplacian.h:42-43: -(g_i . g_j) */
            double d = 0;
            for (int k = 0; k < M; ++k) d += gi[k] * gj[k];
            v = -d;
        } break;
        case FO_DIFFUSION: { /* diffusion.h:53-54: -(g_i . K g_j) */
            const double *K = T->space_varying ? &T->data[qrow * M * M] : T->cst;
            double d = 0;
            for (int r = 0; r < M; ++r) {
                double kg = 0;
                for (int c = 0; c < M; ++c) kg += K[r * M + c] * gj[c];
                d += gi[r] * kg;
            }
            v = -d;
        } break;
        case FO_ADVECTION: { /* advection.h:54-55: psi_i * (g_j . b) */
            const double *b = T->space_varying ? &T->data[qrow * M] : T->cst;
            double d = 0;
            for (int k = 0; k < M; ++k) d += gj[k] * b[k];
            v = psi_i * d;
        } break;
        case FO_REACTION: { /* reaction.h:51-52: c * psi_i * psi_j */
            double c = T->space_varying ? T->data[qrow] : T->cst[0];
            v = c * psi_i * psi_j;
        } break;
        default: v = 0; /* dt.h:34-36: zero field */
        }
        total = t == 0 ? T->coef * v : total + T->coef * v;
    }
    return total;
}

static int op_is_symmetric(int n_terms, const fo_term *terms) {
    for (int t = 0; t < n_terms; ++t)
        if (terms[t].kind == FO_ADVECTION) return 0; /* advection.h:45; all other leaves are symmetric */
    return 1;
}

/* physical gradients of all basis functions at all quadrature nodes of a cell */
static void physical_gradients(const fo_tables *t, const fo_geom *g, double grad[FO_MAXB][FO_MAXQ][3]) {
    const int M = t->M;
    for (int i = 0; i < t->nb; ++i)
        for (int q = 0; q < t->nq; ++q)
            for (int r = 0; r < M; ++r) { /* (invJ^T)(r,k) = invJ(k,r) */
                double v = 0;
                for (int k = 0; k < M; ++k) v += g->invJ[k][r] * t->dpsi[i][q][k];
                grad[i][q][r] = v;
            }
}

/* Integrator::integrate_weak_form (integrator.h:92-106): sum_q w_q f(p_q), then * measure */
static inline double integrate_pair(const fo_tables *t, const fo_geom *g, int n_terms, const fo_term *terms,
                                    int64_t cell, int i, int j, double grad[FO_MAXB][FO_MAXQ][3]) {
    double value = 0;
    for (int q = 0; q < t->nq; ++q)
        value += weak_form_at(t->M, n_terms, terms, (int64_t)t->nq * cell + q, t->psi[i][q], t->psi[j][q],
                              grad[i][q], grad[j][q]) *
                 t->qw[q];
    return value * g->measure;
}

/* full local matrix of one cell, row-major nb x nb, all (i,j) pairs (what fem_operators_test.cpp:66-79 loops) */
int fo_local_matrix(int M, int R, int64_t n_nodes, const double *nodes, const int32_t *cell, int64_t cell_id,
                    int n_terms, const fo_term *terms, double *out) {
    fo_tables t;
    fo_geom g;
    double grad[FO_MAXB][FO_MAXQ][3];
    int rc = build_tables(M, R, &t);
    if (rc) return rc;
    cell_geometry(M, n_nodes, nodes, cell, &g);
    physical_gradients(&t, &g, grad);
    for (int i = 0; i < t.nb; ++i)
        for (int j = 0; j < t.nb; ++j) out[i * t.nb + j] = integrate_pair(&t, &g, n_terms, terms, cell_id, i, j, grad);
    return FO_OK;
}
/* physical gradients at quadrature node q of a given rule (lagrangian_basis_test.cpp:150-197 uses the 6-pt rule
 * node 0 for both orders): out nb x M */
int fo_physical_gradients_at(int M, int R, int64_t n_nodes, const double *nodes, const int32_t *cell, const double *p,
                             double *out) {
    double coeff[FO_MAXB * FO_MAXB], gr[3];
    fo_geom g;
    const int nb = fo_n_basis(M, R);
    if (fo_reference_basis(M, R, coeff)) return FO_EINVAL;
    cell_geometry(M, n_nodes, nodes, cell, &g);
    for (int i = 0; i < nb; ++i) {
        fo_poly_grad(M, R, &coeff[i * nb], p, gr);
        for (int r = 0; r < M; ++r) {
            double v = 0;
            for (int k = 0; k < M; ++k) v += g.invJ[k][r] * gr[k];
            out[i * M + r] = v;
        }
    }
    return FO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Assembler::discretize_operator (finite_elements/fem_assembler.h:52-121)
 *   triplet list over cells x i x j (symmetric operators: only dof_i >= dof_j, 96), Eigen setFromTriplets
 *   (duplicates summed in insertion order) + makeCompressed (112-113), selfadjointView<Lower> for symmetric
 *   operators (116-117).  Output: CSR with sorted columns (== the reference's CSC arrays for these
 *   structurally symmetric patterns; compare as matrices otherwise).  Arrays are malloc'ed: fo_free().
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t r, c;
    double v;
} triplet;

static int triplets_to_csr(int64_t n, triplet *T, int64_t nt, int32_t **rowptr_out, int32_t **col_out,
                           double **val_out, int64_t *nnz_out) {
    /* stable counting sort by row, then per row a stable insertion/merge by column with duplicates summed in
     * insertion order (Eigen's collapseDuplicates adds later duplicates onto the first occurrence) */
    int64_t *cnt = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    triplet *S = (triplet *)malloc(sizeof(triplet) * (size_t)(nt ? nt : 1));
    int32_t *rowptr = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    if (!cnt || !S || !rowptr) return FO_ENOMEM;
    for (int64_t k = 0; k < nt; ++k) ++cnt[T[k].r + 1];
    for (int64_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    {
        int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
        if (!pos) return FO_ENOMEM;
        memcpy(pos, cnt, sizeof(int64_t) * ((size_t)n + 1));
        for (int64_t k = 0; k < nt; ++k) S[pos[T[k].r]++] = T[k];
        free(pos);
    }
    /* per row: stable sort by column (rows are short), then fold duplicates */
    int64_t nnz = 0;
    for (int64_t i = 0; i < n; ++i) {
        triplet *a = &S[cnt[i]];
        int64_t len = cnt[i + 1] - cnt[i];
        for (int64_t x = 1; x < len; ++x) { /* insertion sort is stable */
            triplet t = a[x];
            int64_t y = x - 1;
            for (; y >= 0 && a[y].c > t.c; --y) a[y + 1] = a[y];
            a[y + 1] = t;
        }
        rowptr[i] = (int32_t)nnz;
        for (int64_t x = 0; x < len;) {
            double s = a[x].v;
            int64_t y = x + 1;
            for (; y < len && a[y].c == a[x].c; ++y) s += a[y].v;
            S[nnz].c = a[x].c, S[nnz].v = s, ++nnz;
            x = y;
        }
    }
    rowptr[n] = (int32_t)nnz;
    int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    double *val = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    if (!col || !val) return FO_ENOMEM;
    for (int64_t k = 0; k < nnz; ++k) col[k] = S[k].c, val[k] = S[k].v;
    free(S), free(cnt);
    *rowptr_out = rowptr, *col_out = col, *val_out = val, *nnz_out = nnz;
    return FO_OK;
}

int fo_assemble_operator(int M, int R, int64_t n_nodes, const double *nodes, int64_t n_cells, const int32_t *cells,
                         const int32_t *dofs, int64_t n_dofs, int n_terms, const fo_term *terms, int32_t **rowptr,
                         int32_t **colidx, double **values, int64_t *nnz) {
    fo_tables t;
    int rc = build_tables(M, R, &t);
    if (rc) return rc;
    const int nb = t.nb, sym = op_is_symmetric(n_terms, terms);
    int64_t cap = (int64_t)nb * nb * n_cells, nt = 0;
    triplet *T = (triplet *)malloc(sizeof(triplet) * (size_t)(cap ? cap : 1));
    if (!T) return FO_ENOMEM;
    for (int64_t c = 0; c < n_cells; ++c) {
        fo_geom g;
        double grad[FO_MAXB][FO_MAXQ][3];
        const int32_t *d = &dofs[c * nb];
        cell_geometry(M, n_nodes, nodes, &cells[c * (M + 1)], &g);
        physical_gradients(&t, &g, grad);
        for (int i = 0; i < nb; ++i)
            for (int j = 0; j < nb; ++j) {
                if (sym && d[i] < d[j]) continue; /* fem_assembler.h:96 */
                T[nt].r = d[i], T[nt].c = d[j];
                T[nt].v = integrate_pair(&t, &g, n_terms, terms, c, i, j, grad);
                ++nt;
            }
    }
    if (sym) { /* selfadjointView<Lower>: mirror the strictly-lower part (fem_assembler.h:116-117) */
        int64_t lower = nt;
        for (int64_t k = 0; k < lower; ++k)
            if (T[k].r != T[k].c) T[nt].r = T[k].c, T[nt].c = T[k].r, T[nt].v = T[k].v, ++nt;
    }
    rc = triplets_to_csr(n_dofs, T, nt, rowptr, colidx, values, nnz);
    free(T);
    return rc;
}
void fo_free(void *p) { free(p); }

/* Assembler::discretize_forcing (fem_assembler.h:122-136) + Integrator::integrate(e,f,Phi) (integrator.h:73-90)
 * with f sampled at quadrature nodes: b[dof(e,i)] += measure * sum_q f[nq*e+q] * psi_i(p_q) * w_q */
int fo_assemble_forcing(int M, int R, int64_t n_nodes, const double *nodes, int64_t n_cells, const int32_t *cells,
                        const int32_t *dofs, int64_t n_dofs, const double *f_q, double *b) {
    fo_tables t;
    int rc = build_tables(M, R, &t);
    if (rc) return rc;
    memset(b, 0, sizeof(double) * (size_t)n_dofs);
    for (int64_t c = 0; c < n_cells; ++c) {
        fo_geom g;
        cell_geometry(M, n_nodes, nodes, &cells[c * (M + 1)], &g);
        for (int i = 0; i < t.nb; ++i) {
            double value = 0;
            for (int q = 0; q < t.nq; ++q) value += (f_q[(int64_t)t.nq * c + q] * t.psi[i][q]) * t.qw[q];
            b[dofs[c * t.nb + i]] += value * g.measure;
        }
    }
    return FO_OK;
}

/* FEMSolverBase::set_dirichlet_bc (finite_elements/solvers/fem_solver_base.h:142-155): zero the row (pattern
 * keeps explicit zeros), unit diagonal, rhs = g */
int fo_set_dirichlet(int64_t n_dofs, const int32_t *rowptr, const int32_t *colidx, double *values, double *force,
                     const uint8_t *boundary_dofs, const double *g) {
    for (int64_t i = 0; i < n_dofs; ++i) {
        if (!boundary_dofs[i]) continue;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) values[k] = colidx[k] == i ? 1.0 : 0.0;
        force[i] = g[i];
    }
    return FO_OK;
}

/* y = A x, CSR */
void fo_spmv(int64_t n, const int32_t *rowptr, const int32_t *colidx, const double *values, const double *x,
             double *y) {
    for (int64_t i = 0; i < n; ++i) {
        double s = 0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += values[k] * x[colidx[k]];
        y[i] = s;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Solves.  The reference's solve is Eigen::SparseLU (fem_linear_elliptic_solver.h:38-47), third-party and not
 * restated; tests/ uses scipy's SuperLU for small systems.  For sizes where LU fill is infeasible the CPU
 * baseline is the same Krylov method the HIP path uses, on the row-zeroed Dirichlet system reduced to its
 * interior block (identical solution; see DESIGN.md "Dirichlet handling").
 *
 * fo_pcg: Jacobi-preconditioned CG on A_II u_I = f_I - A_IB g with u_B = g.  A is the UNMODIFIED operator.
 * fo_bicgstab: Jacobi (left) preconditioned BiCGStab, same reduction, for non-symmetric operators.
 * Stop when ||r||_{D^-1} <= rtol * ||r0||_{D^-1}  (pcg)  /  ||D^-1 r||_2 <= rtol * ||D^-1 r0||_2 (bicgstab).
 * ---------------------------------------------------------------------------------------------- */
int fo_pcg(int64_t n, const int32_t *rowptr, const int32_t *colidx, const double *values, const double *force,
           const uint8_t *bnd, const double *g, double rtol, int maxit, double *u, int *iters, double *relres) {
    double *r = (double *)malloc(sizeof(double) * 4 * (size_t)n), *z, *p, *Ap, *dinv;
    dinv = (double *)malloc(sizeof(double) * (size_t)n);
    if (!r || !dinv) return FO_ENOMEM;
    z = r + n, p = z + n, Ap = p + n;
    for (int64_t i = 0; i < n; ++i) {
        double d = 0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (colidx[k] == i) d = values[k];
        dinv[i] = (bnd && bnd[i]) ? 0.0 : 1.0 / d;
        u[i] = (bnd && bnd[i]) ? g[i] : 0.0;
    }
    fo_spmv(n, rowptr, colidx, values, u, Ap); /* A g~ */
    double rz = 0;
    for (int64_t i = 0; i < n; ++i) {
        r[i] = (bnd && bnd[i]) ? 0.0 : force[i] - Ap[i];
        z[i] = dinv[i] * r[i], p[i] = z[i], rz += r[i] * z[i];
    }
    const double rz0 = rz;
    int it = 0, rc = FO_ENOCONV;
    if (rz0 == 0.0) rc = FO_OK;
    while (rc != FO_OK && it < maxit) {
        fo_spmv(n, rowptr, colidx, values, p, Ap);
        double pAp = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (bnd && bnd[i]) Ap[i] = 0.0;
            pAp += p[i] * Ap[i];
        }
        const double alpha = rz / pAp;
        double rz_new = 0;
        for (int64_t i = 0; i < n; ++i) {
            u[i] += alpha * p[i], r[i] -= alpha * Ap[i];
            z[i] = dinv[i] * r[i], rz_new += r[i] * z[i];
        }
        ++it;
        if (sqrt(rz_new) <= rtol * sqrt(rz0)) {
            rz = rz_new, rc = FO_OK;
            break;
        }
        const double beta = rz_new / rz;
        for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
        rz = rz_new;
    }
    *iters = it, *relres = rz0 > 0 ? sqrt(rz / rz0) : 0.0;
    free(r), free(dinv);
    return rc;
}

int fo_bicgstab(int64_t n, const int32_t *rowptr, const int32_t *colidx, const double *values, const double *force,
                const uint8_t *bnd, const double *g, double rtol, int maxit, double *u, int *iters, double *relres) {
    double *w = (double *)malloc(sizeof(double) * 8 * (size_t)n);
    if (!w) return FO_ENOMEM;
    double *r = w, *r0 = w + n, *p = w + 2 * n, *v = w + 3 * n, *s = w + 4 * n, *t = w + 5 * n, *dinv = w + 6 * n,
           *tmp = w + 7 * n;
    for (int64_t i = 0; i < n; ++i) {
        double d = 0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (colidx[k] == i) d = values[k];
        dinv[i] = (bnd && bnd[i]) ? 0.0 : 1.0 / d;
        u[i] = (bnd && bnd[i]) ? g[i] : 0.0;
    }
    /* preconditioned system  D^-1 A_II u_I = D^-1 (f_I - A_IB g); boundary rows/cols masked by dinv = 0 */
    fo_spmv(n, rowptr, colidx, values, u, tmp);
    double rr0 = 0;
    for (int64_t i = 0; i < n; ++i) {
        r[i] = dinv[i] * (force[i] - tmp[i]);
        r0[i] = r[i], p[i] = 0, v[i] = 0, rr0 += r[i] * r[i];
    }
    double rho = 1, alpha = 1, omega = 1, rr = rr0;
    int it = 0, rc = rr0 == 0.0 ? FO_OK : FO_ENOCONV;
    while (rc != FO_OK && it < maxit) {
        double rho_new = 0;
        for (int64_t i = 0; i < n; ++i) rho_new += r0[i] * r[i];
        if (rho_new == 0.0) break;
        const double beta = (rho_new / rho) * (alpha / omega);
        for (int64_t i = 0; i < n; ++i) p[i] = r[i] + beta * (p[i] - omega * v[i]);
        fo_spmv(n, rowptr, colidx, values, p, tmp);
        double r0v = 0;
        for (int64_t i = 0; i < n; ++i) v[i] = dinv[i] * tmp[i], r0v += r0[i] * v[i];
        alpha = rho_new / r0v;
        double ss = 0;
        for (int64_t i = 0; i < n; ++i) s[i] = r[i] - alpha * v[i], ss += s[i] * s[i];
        ++it;
        if (sqrt(ss) <= rtol * sqrt(rr0)) {
            for (int64_t i = 0; i < n; ++i) u[i] += alpha * p[i];
            rr = ss, rc = FO_OK;
            break;
        }
        fo_spmv(n, rowptr, colidx, values, s, tmp);
        double ts = 0, tt = 0;
        for (int64_t i = 0; i < n; ++i) t[i] = dinv[i] * tmp[i], ts += t[i] * s[i], tt += t[i] * t[i];
        omega = ts / tt;
        rr = 0;
        for (int64_t i = 0; i < n; ++i) {
            u[i] += alpha * p[i] + omega * s[i];
            r[i] = s[i] - omega * t[i], rr += r[i] * r[i];
        }
        rho = rho_new;
        if (sqrt(rr) <= rtol * sqrt(rr0)) {
            rc = FO_OK;
            break;
        }
    }
    *iters = it, *relres = rr0 > 0 ? sqrt(rr / rr0) : 0.0;
    free(w);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Basis evaluation matrices (pin the DOF numbering against the reference's golden .mtx files)
 * ---------------------------------------------------------------------------------------------- */
/* pointwise_evaluation::eval (lagrangian_basis.h:203-235): Psi(i, dofs(e,h)) = psi_h(invJ (p_i - x0)), e = cell
 * containing p_i.  Point location here is a brute-force scan (first cell with all barycentric coords >= -tol);
 * the reference uses a tree search (geometry/tree_search.h) -- on shared vertices/edges the choice of cell does
 * not change the matrix values.  Output dense row-major n_locs x n_dofs (test-sized only). */
int fo_pointwise_psi(int M, int R, int64_t n_nodes, const double *nodes, int64_t n_cells, const int32_t *cells,
                     const int32_t *dofs, int64_t n_dofs, int64_t n_locs, const double *locs /*col-major*/,
                     double *psi_dense) {
    double coeff[FO_MAXB * FO_MAXB];
    const int nb = fo_n_basis(M, R);
    if (fo_reference_basis(M, R, coeff)) return FO_EINVAL;
    memset(psi_dense, 0, sizeof(double) * (size_t)(n_locs * n_dofs));
    for (int64_t l = 0; l < n_locs; ++l) {
        for (int64_t c = 0; c < n_cells; ++c) {
            fo_geom g;
            double z[3], z0 = 1.0;
            cell_geometry(M, n_nodes, nodes, &cells[c * (M + 1)], &g);
            for (int r = 0; r < M; ++r) {
                double v = 0;
                for (int k = 0; k < M; ++k) v += g.invJ[r][k] * (locs[k * n_locs + l] - g.x0[k]);
                z[r] = v, z0 -= v;
            }
            int inside = z0 >= -1e-12;
            for (int r = 0; r < M; ++r) inside &= z[r] >= -1e-12;
            if (!inside) continue;
            for (int h = 0; h < nb; ++h) psi_dense[l * n_dofs + dofs[c * nb + h]] = fo_poly_eval(M, R, &coeff[h * nb], z);
            break;
        }
    }
    return FO_OK;
}
/* areal_evaluation::eval (lagrangian_basis.h:238-283): Psi(k, dofs(e,h)) += int_e psi_h / |D_k| for cells flagged in
 * incidence row k; the integral uses Integrator::integrate_cell (integrator.h:47-63): sum_q psi_h(p_q) w_q * measure */
int fo_areal_psi(int M, int R, int64_t n_nodes, const double *nodes, int64_t n_cells, const int32_t *cells,
                 const int32_t *dofs, int64_t n_dofs, int64_t n_sub, const double *incidence /*row-major n_sub x n_cells*/,
                 double *psi_dense, double *D) {
    fo_tables t;
    int rc = build_tables(M, R, &t);
    if (rc) return rc;
    memset(psi_dense, 0, sizeof(double) * (size_t)(n_sub * n_dofs));
    for (int64_t k = 0; k < n_sub; ++k) {
        double Di = 0;
        for (int64_t c = 0; c < n_cells; ++c)
            if (incidence[k * n_cells + c] == 1) {
                fo_geom g;
                cell_geometry(M, n_nodes, nodes, &cells[c * (M + 1)], &g);
                Di += g.measure;
            }
        for (int64_t c = 0; c < n_cells; ++c) {
            if (incidence[k * n_cells + c] != 1) continue;
            fo_geom g;
            cell_geometry(M, n_nodes, nodes, &cells[c * (M + 1)], &g);
            for (int h = 0; h < t.nb; ++h) {
                double v = 0;
                for (int q = 0; q < t.nq; ++q) v += t.psi[h][q] * t.qw[q];
                psi_dense[k * n_dofs + dofs[c * t.nb + h]] += (v * g.measure) / Di;
            }
        }
        D[k] = Di;
    }
    return FO_OK;
}
