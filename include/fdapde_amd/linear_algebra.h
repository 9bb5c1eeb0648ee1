// Dense / sparse helpers of the statistical models that consume the PDE solver, written on the
// facade's own containers so that they compile where Eigen is absent:
//   lump(A)              row-sum lumping                    fdaPDE/linear_algebra/lumping.h:30-51
//   PartialPivLU         small dense LU (q x q)             stands in for Eigen::PartialPivLU<DMatrix<double>>
//   SMW<SparseSolver>    Sherman-Morrison-Woodbury solve    fdaPDE/linear_algebra/smw.h:38-59
// The heavy step of SMW, A^{-1} [b | U] (1 + q right-hand sides against one prepared sparse system), runs on the device as ONE
// batched multi-column solve through the factor-once handle (PDE::SparseSolver -> fdapde_lin_compute / fdapde_lin_solve);
// everything dense is q x q or n x q host work.
#ifndef FDAPDE_AMD_LINEAR_ALGEBRA_H
#define FDAPDE_AMD_LINEAR_ALGEBRA_H

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <utility>
#include <vector>

#include "pde.h"

namespace fdapde {
namespace amd {

// ---- dense products / sums on column-major DMatrix ---------------------------------------------------------------------
inline DMatrix<double> operator*(const DMatrix<double>& a, const DMatrix<double>& b) {
    if (a.cols() != b.rows()) throw std::runtime_error("DMatrix product: inner dimensions differ");
    DMatrix<double> c(a.rows(), b.cols(), 0.0);
    for (int64_t j = 0; j < b.cols(); ++j)
        for (int64_t k = 0; k < a.cols(); ++k) {
            const double bkj = b(k, j);
            for (int64_t i = 0; i < a.rows(); ++i) c(i, j) += a(i, k) * bkj;
        }
    return c;
}
inline DMatrix<double> operator+(const DMatrix<double>& a, const DMatrix<double>& b) {
    if (a.rows() != b.rows() || a.cols() != b.cols()) throw std::runtime_error("DMatrix sum: shapes differ");
    DMatrix<double> c(a.rows(), a.cols());
    for (int64_t i = 0; i < a.size(); ++i) c.data()[i] = a.data()[i] + b.data()[i];
    return c;
}
inline DMatrix<double> operator-(const DMatrix<double>& a, const DMatrix<double>& b) {
    if (a.rows() != b.rows() || a.cols() != b.cols()) throw std::runtime_error("DMatrix difference: shapes differ");
    DMatrix<double> c(a.rows(), a.cols());
    for (int64_t i = 0; i < a.size(); ++i) c.data()[i] = a.data()[i] - b.data()[i];
    return c;
}
inline DMatrix<double> transpose(const DMatrix<double>& a) {
    DMatrix<double> t(a.cols(), a.rows());
    for (int64_t j = 0; j < a.cols(); ++j)
        for (int64_t i = 0; i < a.rows(); ++i) t(j, i) = a(i, j);
    return t;
}

// ---- lumping (lumping.h:30-41): diagonal sparse matrix of the row sums -------------------------------------------------
template <typename T> SpMatrix<T> lump(const SpMatrix<T>& a) {
    if (a.rows() != a.cols()) throw std::runtime_error("lump: matrix must be square");   // fdapde_assert, lumping.h:31
    SpMatrix<T> d;
    d.n_rows = d.n_cols = a.rows();
    d.rowptr.resize((size_t)a.rows() + 1);
    d.colidx.resize((size_t)a.rows());
    d.values.resize((size_t)a.rows());
    for (int64_t i = 0; i < a.rows(); ++i) {
        T s = 0;
        for (int32_t k = a.rowptr[(size_t)i]; k < a.rowptr[(size_t)i + 1]; ++k) s += a.values[(size_t)k];
        d.rowptr[(size_t)i] = (int32_t)i, d.colidx[(size_t)i] = (int32_t)i, d.values[(size_t)i] = s;
    }
    d.rowptr[(size_t)a.rows()] = (int32_t)a.rows();
    return d;
}

// ---- small dense LU with partial pivoting (the DenseSolver of SMW; q x q) ----------------------------------------------
class PartialPivLU {
   public:
    void compute(const DMatrix<double>& a) {
        if (a.rows() != a.cols()) throw std::runtime_error("PartialPivLU: matrix must be square");
        n_ = a.rows(), lu_ = a, piv_.resize((size_t)n_);
        for (int64_t k = 0; k < n_; ++k) {
            int64_t p = k;
            for (int64_t i = k + 1; i < n_; ++i)
                if (std::fabs(lu_(i, k)) > std::fabs(lu_(p, k))) p = i;
            piv_[(size_t)k] = p;
            if (p != k)
                for (int64_t j = 0; j < n_; ++j) std::swap(lu_(k, j), lu_(p, j));
            if (lu_(k, k) == 0.0) throw std::runtime_error("PartialPivLU: singular matrix");
            for (int64_t i = k + 1; i < n_; ++i) {
                lu_(i, k) /= lu_(k, k);
                for (int64_t j = k + 1; j < n_; ++j) lu_(i, j) -= lu_(i, k) * lu_(k, j);
            }
        }
    }
    DMatrix<double> solve(const DMatrix<double>& b) const {
        if (b.rows() != n_) throw std::runtime_error("PartialPivLU: right-hand side has the wrong number of rows");
        DMatrix<double> x = b;
        for (int64_t c = 0; c < x.cols(); ++c) {
            for (int64_t k = 0; k < n_; ++k) std::swap(x(k, c), x(piv_[(size_t)k], c));
            for (int64_t i = 1; i < n_; ++i)
                for (int64_t j = 0; j < i; ++j) x(i, c) -= lu_(i, j) * x(j, c);
            for (int64_t i = n_ - 1; i >= 0; --i) {
                for (int64_t j = i + 1; j < n_; ++j) x(i, c) -= lu_(i, j) * x(j, c);
                x(i, c) /= lu_(i, i);
            }
        }
        return x;
    }
   private:
    int64_t n_ = 0;
    DMatrix<double> lu_;
    std::vector<int64_t> piv_;
};

// ---- Sherman-Morrison-Woodbury (same call signature as fdaPDE/linear_algebra/smw.h:38-59) -----------------------------
//   (A + U C V) x = b,  invC = C^{-1} supplied:   x = W_b - W_U (C^{-1} + V W_U)^{-1} (V W_b),   [W_b | W_U] = A^{-1} [b | U]
// ONE pass through the sparse solver: the m columns of b and the q columns of U go to the device as one (m + q)-column
// right-hand side: the columns of a small system run side by side in one persistent launch, those of a large one share every sweep over
// the matrix (fdapde_lin_solve batches 8 / 4 columns per multi-RHS CG, kernels_multirhs.h).  The correction A^{-1} U t of the formula is W_U t -- a dense n x q by q x m product on data already
// there, not another sparse solve (the reference performs three separate solves: b, U, and U t).
template <typename SparseSolver, typename DenseSolver = PartialPivLU> struct SMW {
    SMW() = default;
    DMatrix<double> solve(const SparseSolver& invA, const DMatrix<double>& U, const DMatrix<double>& invC, const DMatrix<double>& V,
                          const DMatrix<double>& b) {
        const int64_t n = b.rows(), m = b.cols(), q = U.cols();
        if (U.rows() != n || V.cols() != n || V.rows() != q || invC.rows() != q || invC.cols() != q)
            throw std::runtime_error("SMW: shapes of U, C^{-1}, V, b do not fit");
        DMatrix<double> rhs(n, m + q);
        std::copy(b.data(), b.data() + n * m, rhs.data());
        std::copy(U.data(), U.data() + n * q, rhs.data() + n * m);
        const DMatrix<double> W = invA.solve(rhs);                 // the only sparse solve: n x (m + q)
        // small dense part: S = C^{-1} + V W_U (q x q),  z = V W_b (q x m)
        DMatrix<double> S = invC, z(q, m, 0.0);
        for (int64_t c = 0; c < q + m; ++c) {
            const double* w = W.data() + n * c;                    // column c of W: first the m columns of b, then U's
            for (int64_t r = 0; r < q; ++r) {
                double acc = 0;
                for (int64_t i = 0; i < n; ++i) acc += V(r, i) * w[i];
                if (c < m) z(r, c) = acc;
                else S(r, c - m) += acc;
            }
        }
        DenseSolver dense;
        dense.compute(S);
        const DMatrix<double> t = dense.solve(z);                  // q x m
        DMatrix<double> x(n, m);
        for (int64_t c = 0; c < m; ++c) {
            double* xc = x.data() + n * c;
            std::copy(W.data() + n * c, W.data() + n * (c + 1), xc);
            for (int64_t k = 0; k < q; ++k) {
                const double tk = t(k, c);
                const double* wu = W.data() + n * (m + k);
                for (int64_t i = 0; i < n; ++i) xc[i] -= wu[i] * tk;
            }
        }
        return x;
    }
};

}   // namespace amd
}   // namespace fdapde

#endif
