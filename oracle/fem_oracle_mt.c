/* fem_oracle_mt.c -- "CPU-best" column of BASELINE.md section 2: the CPU restatement of fem_oracle.c run on ALL host cores.
 *
 * TEST / BENCHMARK INFRASTRUCTURE ONLY, like fem_oracle.c (which this file includes so that it shares the per-cell
 * routines: reference tables, cell geometry, weak forms).  Nothing under fdapde-core_amd/ may link or load it; only
 * tests/ and bench.py's cpu_baseline leg do.
 *
 * What changes against the faithful single-threaded port:
 *   - assembly writes into a PREBUILT CSR pattern (the one fo_assemble_operator produced) instead of sorting triplets,
 *     cell by cell inside colour classes: cells of one colour share no DOF, so a class is an OpenMP parallel loop
 *     without atomics (the reference loop fem_assembler.h:61-111 is serial; colouring is this build's addition);
 *   - the Jacobi-PCG of fo_pcg with row-parallel SpMV and OpenMP reductions (same recurrences, same stopping rule).
 * Built with -O3 -march=native -fopenmp (oracle/Makefile: libfem_oracle_mt.so).  Results agree with the faithful port up
 * to summation order; tests/test_oracle_mt.py states the tolerances. */
#include "fem_oracle.c"

#include <omp.h>

int fo_mt_threads(void) { return omp_get_max_threads(); }
void fo_mt_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* Greedy cell colouring through the DOF -> cells adjacency (serial set-up, not timed): colour[c] = smallest colour not
 * used by an already coloured cell sharing a DOF with c.  Outputs the cells grouped by colour. */
int fo_mt_colour_cells(int64_t n_dofs, int64_t n_cells, int nb, const int32_t *dofs, int32_t *order,
                       int64_t *colour_ptr /* >= 257 */, int32_t *n_colours) {
    int64_t *ptr = (int64_t *)calloc((size_t)n_dofs + 1, sizeof(int64_t));
    int32_t *adj = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_cells * nb > 0 ? n_cells * nb : 1));
    int16_t *colour = (int16_t *)malloc(sizeof(int16_t) * (size_t)(n_cells ? n_cells : 1));
    if (!ptr || !adj || !colour) return FO_ENOMEM;
    for (int64_t k = 0; k < n_cells * nb; ++k) ++ptr[dofs[k] + 1];
    for (int64_t i = 0; i < n_dofs; ++i) ptr[i + 1] += ptr[i];
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_dofs ? n_dofs : 1));
    if (!pos) return FO_ENOMEM;
    memcpy(pos, ptr, sizeof(int64_t) * (size_t)n_dofs);
    for (int64_t c = 0; c < n_cells; ++c)
        for (int j = 0; j < nb; ++j) adj[pos[dofs[c * nb + j]]++] = (int32_t)c;
    int ncol = 0;
    for (int64_t c = 0; c < n_cells; ++c) {
        uint64_t used[4] = {0, 0, 0, 0}; /* up to 256 colours */
        for (int j = 0; j < nb; ++j) {
            const int32_t d = dofs[c * nb + j];
            for (int64_t k = ptr[d]; k < ptr[d + 1] && adj[k] < c; ++k) used[colour[adj[k]] >> 6] |= 1ull << (colour[adj[k]] & 63);
        }
        int col = 0;
        while (col < 256 && (used[col >> 6] >> (col & 63) & 1)) ++col;
        if (col == 256) return FO_EINVAL;
        colour[c] = (int16_t)col;
        if (col + 1 > ncol) ncol = col + 1;
    }
    for (int k = 0; k <= ncol; ++k) colour_ptr[k] = 0;
    for (int64_t c = 0; c < n_cells; ++c) ++colour_ptr[colour[c] + 1];
    for (int k = 0; k < ncol; ++k) colour_ptr[k + 1] += colour_ptr[k];
    int64_t fill[256];
    for (int k = 0; k < ncol; ++k) fill[k] = colour_ptr[k];
    for (int64_t c = 0; c < n_cells; ++c) order[fill[colour[c]]++] = (int32_t)c;
    *n_colours = ncol;
    free(ptr), free(adj), free(colour), free(pos);
    return FO_OK;
}

static inline int32_t find_slot(const int32_t *colidx, int32_t lo, int32_t hi, int32_t col) {
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (colidx[mid] < col) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

/* values of the operator into the given CSR pattern + (optionally) the forcing vector, colour class by colour class */
int fo_mt_assemble(int M, int R, int64_t n_nodes, const double *nodes, int64_t n_cells, const int32_t *cells,
                   const int32_t *dofs, int64_t n_dofs, int n_terms, const fo_term *terms, const int32_t *rowptr,
                   const int32_t *colidx, int32_t n_colours, const int64_t *colour_ptr, const int32_t *order,
                   double *values, const double *f_q, double *b) {
    fo_tables t;
    int rc = build_tables(M, R, &t);
    if (rc) return rc;
    const int nb = t.nb, sym = op_is_symmetric(n_terms, terms);
    const int64_t nnz = rowptr[n_dofs];
    (void)n_cells; /* the colour classes enumerate the cells */
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nnz; ++k) values[k] = 0.0;
    if (b) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n_dofs; ++i) b[i] = 0.0;
    }
    for (int32_t col = 0; col < n_colours; ++col) {
#pragma omp parallel for schedule(static)
        for (int64_t k = colour_ptr[col]; k < colour_ptr[col + 1]; ++k) {
            const int64_t c = order[k];
            fo_geom g;
            double grad[FO_MAXB][FO_MAXQ][3];
            const int32_t *d = &dofs[c * nb];
            cell_geometry(M, n_nodes, nodes, &cells[c * (M + 1)], &g);
            physical_gradients(&t, &g, grad);
            for (int i = 0; i < nb; ++i) {
                for (int j = 0; j < nb; ++j) {
                    if (sym && j > i) continue; /* one integral per unordered pair, mirrored below */
                    const double v = integrate_pair(&t, &g, n_terms, terms, c, i, j, grad);
                    values[find_slot(colidx, rowptr[d[i]], rowptr[d[i] + 1], d[j])] += v;
                    if (sym && j != i) values[find_slot(colidx, rowptr[d[j]], rowptr[d[j] + 1], d[i])] += v;
                }
                if (b) {
                    double value = 0;
                    for (int q = 0; q < t.nq; ++q) value += (f_q[(int64_t)t.nq * c + q] * t.psi[i][q]) * t.qw[q];
                    b[d[i]] += value * g.measure;
                }
            }
        }
    }
    return FO_OK;
}

static void spmv_mt(int64_t n, const int32_t *rowptr, const int32_t *colidx, const double *values, const double *x, double *y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double s = 0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += values[k] * x[colidx[k]];
        y[i] = s;
    }
}

/* fo_pcg with every loop on all cores */
int fo_mt_pcg(int64_t n, const int32_t *rowptr, const int32_t *colidx, const double *values, const double *force,
              const uint8_t *bnd, const double *g, double rtol, int maxit, double *u, int *iters, double *relres) {
    double *r = (double *)malloc(sizeof(double) * 4 * (size_t)n), *z, *p, *Ap, *dinv;
    dinv = (double *)malloc(sizeof(double) * (size_t)n);
    if (!r || !dinv) return FO_ENOMEM;
    z = r + n, p = z + n, Ap = p + n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double d = 0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (colidx[k] == i) d = values[k];
        dinv[i] = (bnd && bnd[i]) ? 0.0 : 1.0 / d;
        u[i] = (bnd && bnd[i]) ? g[i] : 0.0;
    }
    spmv_mt(n, rowptr, colidx, values, u, Ap);
    double rz = 0;
#pragma omp parallel for schedule(static) reduction(+ : rz)
    for (int64_t i = 0; i < n; ++i) {
        r[i] = (bnd && bnd[i]) ? 0.0 : force[i] - Ap[i];
        z[i] = dinv[i] * r[i], p[i] = z[i], rz += r[i] * z[i];
    }
    const double rz0 = rz;
    int it = 0, rc = FO_ENOCONV;
    if (rz0 == 0.0) rc = FO_OK;
    while (rc != FO_OK && it < maxit) {
        spmv_mt(n, rowptr, colidx, values, p, Ap);
        double pAp = 0;
#pragma omp parallel for schedule(static) reduction(+ : pAp)
        for (int64_t i = 0; i < n; ++i) {
            if (bnd && bnd[i]) Ap[i] = 0.0;
            pAp += p[i] * Ap[i];
        }
        const double alpha = rz / pAp;
        double rz_new = 0;
#pragma omp parallel for schedule(static) reduction(+ : rz_new)
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
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
        rz = rz_new;
    }
    *iters = it, *relres = rz0 > 0 ? sqrt(rz / rz0) : 0.0;
    free(r), free(dinv);
    return rc;
}
