// fdapde_amd/pde.h -- header-only C++20 facade of the MI355X assemble-and-solve path.
//
// Mirrors, name for name, the part of fdaPDE-core's public interface that sits on the hot path, on top of the C ABI of
// include/fdapde_hip.h (no Eigen required; an adapter is provided when <Eigen/Sparse> is available):
//
//   reference (fdaPDE/...)                                           here (namespace fdapde::amd)
//   ---------------------------------------------------------------  -------------------------------------------------
//   Triangulation<M,N>(nodes, cells, boundary)  geometry/triangulation.h:49   Triangulation<M,N>
//   laplacian<FEM>() diffusion<FEM>(K) advection<FEM>(b) reaction<FEM>(c) dt<FEM>()
//       pde/differential_operators.h:27-52, finite_elements/operators/*.h       same names, tag FEM_HIP
//   operator algebra  -L, L1 + L2, L1 - L2, c * L  pde/differential_expressions.h:49,95-118   same
//   fem_order<R>      finite_elements/fem_symbols.h:24-29                         fem_order<R>
//   PDE<D,E,F,S,Ts...>  pde/pde.h:40-114: ctors 58-72, set_forcing / set_differential_operator / set_dirichlet_bc 74-77,
//       domain() forcing_data() boundary_data() n_dofs() solution() force() stiff() mass() dof_coords() dofs()
//       quadrature_nodes() 79-100, init() 101, solve() 102-105                     PDE<D,E,F,FEM_HIP,fem_order<R>>
//   solver flags is_init / success  finite_elements/solvers/fem_solver_base.h:61-62     PDE::is_init(), PDE::success()
//
// Semantics kept: the mesh is held by reference and must outlive the PDE (pde.h:107); operator / forcing / boundary data are
// copied; getters return const references to solver-owned storage; "solver must be initialized first!" is a
// std::runtime_error (fem_solver_base.h:146, fem_linear_elliptic_solver.h:36); numerical failure of the solve sets
// success() = false without throwing (fem_linear_elliptic_solver.h:42-45); after solve() with Dirichlet data stiff() is the
// row-zeroed matrix and force() carries the boundary values (fem_solver_base.h:148-152).
// Differences: the forcing type F is DMatrix<double> (values at quadrature nodes, pde.h:49) or ScalarField<N> (a callable,
// evaluated on the host at the quadrature nodes the device computes); sparse matrices are CSR (SpMatrix below), the solve
// is Jacobi-PCG / BiCGStab instead of SparseLU (options in PDE::solver_options()).
#ifndef FDAPDE_AMD_PDE_H
#define FDAPDE_AMD_PDE_H

#include <array>
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <algorithm>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include <typeinfo>

#include "../fdapde_hip.h"
#include "../fdapde_hip.hpp"   // copy-on-write owner of the device context

namespace fdapde {
namespace amd {

// ---- containers (column-major dense, CSR sparse): the layouts of the reference's DMatrix / SpMatrix getters -----------
template <typename T> class DMatrix {
   public:
    DMatrix() = default;
    DMatrix(int64_t rows, int64_t cols, T v = T()) : rows_(rows), cols_(cols), data_((size_t)(rows * cols), v) { }
    static DMatrix Zero(int64_t rows, int64_t cols) { return DMatrix(rows, cols, T(0)); }
    void resize(int64_t rows, int64_t cols) { rows_ = rows, cols_ = cols, data_.assign((size_t)(rows * cols), T()); }
    int64_t rows() const { return rows_; }
    int64_t cols() const { return cols_; }
    int64_t size() const { return rows_ * cols_; }
    T& operator()(int64_t i, int64_t j = 0) { return data_[(size_t)(j * rows_ + i)]; }
    const T& operator()(int64_t i, int64_t j = 0) const { return data_[(size_t)(j * rows_ + i)]; }
    T* data() { return data_.data(); }
    const T* data() const { return data_.data(); }
    std::array<T, 3> row3(int64_t i) const {   // row i padded to 3 entries (points in R^2 / R^3)
        std::array<T, 3> r {};
        for (int64_t j = 0; j < cols_ && j < 3; ++j) r[(size_t)j] = (*this)(i, j);
        return r;
    }
   private:
    int64_t rows_ = 0, cols_ = 0;
    std::vector<T> data_;
};
template <typename T> using DVector = DMatrix<T>;
template <typename T> bool is_empty(const DMatrix<T>& m) { return m.size() == 0; }   // utils/symbols.h:177

template <typename T> struct SpMatrix {   // CSR, sorted columns, int32 indices
    int64_t n_rows = 0, n_cols = 0;
    std::vector<int32_t> rowptr, colidx;
    std::vector<T> values;
    int64_t rows() const { return n_rows; }
    int64_t cols() const { return n_cols; }
    int64_t nonZeros() const { return (int64_t)values.size(); }
    T coeff(int64_t i, int64_t j) const {
        for (int32_t k = rowptr[(size_t)i]; k < rowptr[(size_t)i + 1]; ++k)
            if (colidx[(size_t)k] == j) return values[(size_t)k];
        return T(0);
    }
    DMatrix<T> operator*(const DMatrix<T>& x) const {   // host-side product for checks such as sum(M * err^2)
        DMatrix<T> y(n_rows, x.cols(), T(0));
        for (int64_t c = 0; c < x.cols(); ++c)
            for (int64_t i = 0; i < n_rows; ++i) {
                T s = 0;
                for (int32_t k = rowptr[(size_t)i]; k < rowptr[(size_t)i + 1]; ++k) s += values[(size_t)k] * x(colidx[(size_t)k], c);
                y(i, c) = s;
            }
        return y;
    }
};

// ---- tags ------------------------------------------------------------------------------------------------------------
struct FEM_HIP { };   // strategy tag: finite elements assembled and solved on MI355X
template <int R> struct fem_order { static constexpr int value = R; };

// ---- domain ----------------------------------------------------------------------------------------------------------
template <int M, int N> class Triangulation {
   public:
    static constexpr int local_dim = M, embed_dim = N, n_nodes_per_cell = M + 1;
    Triangulation() = default;
    // nodes: n_nodes x N; cells: n_cells x (M+1), 0-based node ids; boundary: n_nodes x 1 (0/1)
    Triangulation(const DMatrix<double>& nodes, const DMatrix<int>& cells, const DMatrix<int>& boundary) :
        nodes_(nodes), cells_(cells), boundary_(boundary) {
        static_assert((M == 2 && N == 2) || (M == 3 && N == 3), "only Triangulation<2,2> and <3,3> are on the accelerated path");
        if (nodes.cols() != N || cells.cols() != M + 1 || boundary.rows() != nodes.rows())
            throw std::runtime_error("Triangulation: inconsistent matrix shapes");
    }
    const DMatrix<double>& nodes() const { return nodes_; }
    const DMatrix<int>& cells() const { return cells_; }
    const DMatrix<int>& boundary_nodes() const { return boundary_; }
    int64_t n_nodes() const { return nodes_.rows(); }
    int64_t n_cells() const { return cells_.rows(); }
    // ---- the rest of the reference constructor (geometry/triangulation.h:143-196, 319-399), built on the device at first use
    //      (fdapde_topology_build: facets = edges of triangles / faces of tetrahedra in first-seen numbering)
    const DMatrix<int>& neighbors() const { return topo().neighbors; }       // n_cells x (M+1), -1 = none (triangulation.h:65, 402)
    const DMatrix<int>& facets() const { return topo().facet_nodes; }        // edges() for triangles, faces() for tetrahedra
    const DMatrix<int>& facet_to_cells() const { return topo().facet_cells; }
    const DMatrix<int>& cell_to_facets() const { return topo().cell_facets; }
    bool is_facet_on_boundary(int64_t id) const { return topo().facet_bnd[(size_t)id] != 0; }
    int64_t n_facets() const { return topo().facet_nodes.rows(); }
    int64_t n_edges() const { return M == 2 ? n_facets() : topo().edge_nodes.rows(); }
    const DMatrix<int>& edges() const { return M == 2 ? topo().facet_nodes : topo().edge_nodes; }
    bool is_edge_on_boundary(int64_t id) const { return (M == 2 ? topo().facet_bnd : topo().edge_bnd)[(size_t)id] != 0; }
    const DMatrix<int>& face_to_edges() const { return topo().face_edges; }  // tetrahedra only
   private:
    struct Topology {
        DMatrix<int> neighbors, cell_facets, facet_nodes, facet_cells, edge_nodes, face_edges;
        std::vector<uint8_t> facet_bnd, edge_bnd;
    };
    const Topology& topo() const {
        if (topo_) return *topo_;
        fdapde_ctx* ctx = nullptr;
        if (fdapde_ctx_create(0, &ctx) != FDAPDE_OK) throw std::runtime_error("Triangulation: no HIP device for the topology tables (there is no CPU fallback)");
        auto fail = [&](const char* what) {
            const std::string msg = std::string(what) + ": " + fdapde_last_error(ctx);
            fdapde_ctx_destroy(ctx);
            throw std::runtime_error(msg);
        };
        const int64_t nn = n_nodes(), nc = n_cells();
        std::vector<int32_t> cells((size_t)(nc * (M + 1)));
        std::vector<uint8_t> bnd((size_t)nn);
        for (int64_t i = 0; i < nc; ++i)
            for (int j = 0; j <= M; ++j) cells[(size_t)(i * (M + 1) + j)] = cells_(i, j);
        for (int64_t i = 0; i < nn; ++i) bnd[(size_t)i] = boundary_(i) != 0;
        if (fdapde_mesh_upload(ctx, M, N, nn, nodes_.data(), nc, cells.data(), bnd.data()) != FDAPDE_OK) fail("mesh upload");
        int64_t nf = 0, ne = 0;
        if (fdapde_topology_build(ctx, &nf, &ne) != FDAPDE_OK) fail("topology");
        std::vector<int32_t> nb((size_t)(nc * (M + 1))), cf((size_t)(nc * (M + 1))), fn((size_t)(nf * M)), fc((size_t)(nf * 2)), en((size_t)(ne * 2)),
          fe((size_t)(nf * 3));
        auto t = std::make_shared<Topology>();
        t->facet_bnd.resize((size_t)nf), t->edge_bnd.resize((size_t)ne);
        if (fdapde_topology_get(ctx, nb.data(), cf.data(), fn.data(), fc.data(), t->facet_bnd.data(), M == 3 ? en.data() : nullptr,
                                M == 3 ? t->edge_bnd.data() : nullptr, M == 3 ? fe.data() : nullptr) != FDAPDE_OK)
            fail("topology download");
        fdapde_ctx_destroy(ctx);
        auto rowmajor = [](const std::vector<int32_t>& v, int64_t rows, int64_t cols) {
            DMatrix<int> m(rows, cols);
            for (int64_t i = 0; i < rows; ++i)
                for (int64_t j = 0; j < cols; ++j) m(i, j) = v[(size_t)(i * cols + j)];
            return m;
        };
        t->neighbors = rowmajor(nb, nc, M + 1), t->cell_facets = rowmajor(cf, nc, M + 1), t->facet_nodes = rowmajor(fn, nf, M);
        t->facet_cells = rowmajor(fc, nf, 2);
        if (M == 3) t->edge_nodes = rowmajor(en, ne, 2), t->face_edges = rowmajor(fe, nf, 3);
        topo_ = t;
        return *topo_;
    }
    DMatrix<double> nodes_;
    DMatrix<int> cells_, boundary_;
    mutable std::shared_ptr<const Topology> topo_;   // lazily built, like the reference's location policy (triangulation.h:267)
};

// ---- forcing as a callable (reference: ScalarField<N, F>, fields/scalar_field.h) -----------------------------------------
template <int N> struct ScalarField {
    std::function<double(const std::array<double, N>&)> f;
    ScalarField() = default;
    template <typename Fn> ScalarField(Fn fn) : f(fn) { }
    double operator()(const std::array<double, N>& x) const { return f(x); }
};

// ---- operator expressions ---------------------------------------------------------------------------------------------
// The reference's expression tree collapses to a left-to-right sum of scaled leaves (see include/fdapde_hip.h).
class DifferentialExpr {
   public:
    struct Leaf {
        fdapde_term term {};
        std::vector<double> data;   // space-varying coefficient, row-major (nq*n_cells) x width
    };
    DifferentialExpr() = default;
    explicit DifferentialExpr(Leaf l) { leaves_.push_back(std::move(l)); }
    DifferentialExpr operator-() const {
        DifferentialExpr r(*this);
        for (auto& l : r.leaves_) l.term.coef = -l.term.coef;
        return r;
    }
    friend DifferentialExpr operator+(const DifferentialExpr& a, const DifferentialExpr& b) {
        DifferentialExpr r(a);
        r.leaves_.insert(r.leaves_.end(), b.leaves_.begin(), b.leaves_.end());
        return r;
    }
    friend DifferentialExpr operator-(const DifferentialExpr& a, const DifferentialExpr& b) { return a + (-b); }
    friend DifferentialExpr operator*(double c, const DifferentialExpr& a) {
        DifferentialExpr r(a);
        for (auto& l : r.leaves_) l.term.coef *= c;
        return r;
    }
    // is_symmetric = AND over leaves, advection is the only non-symmetric one (differential_expressions.h:70-73)
    bool is_symmetric() const {
        for (const auto& l : leaves_)
            if (l.term.kind == FDAPDE_ADVECTION) return false;
        return true;
    }
    // is_parabolic: the expression contains dT() (pde/differential_operators.h:47-52)
    bool is_parabolic() const {
        for (const auto& l : leaves_)
            if (l.term.kind == FDAPDE_DT) return true;
        return false;
    }
    bool is_space_varying() const {
        for (const auto& l : leaves_)
            if (l.term.space_varying) return true;
        return false;
    }
    const std::vector<Leaf>& leaves() const { return leaves_; }
    std::vector<fdapde_term> c_terms() const {   // pointers into this object's storage
        std::vector<fdapde_term> t;
        for (const auto& l : leaves_) {
            fdapde_term x = l.term;
            x.data = l.term.space_varying ? l.data.data() : nullptr;
            t.push_back(x);
        }
        return t;
    }
   private:
    std::vector<Leaf> leaves_;
};
namespace detail {
inline DifferentialExpr leaf(int kind, const double* cst, int n_cst, const DMatrix<double>* data) {
    DifferentialExpr::Leaf l;
    l.term.kind = kind, l.term.coef = 1.0, l.term.space_varying = data ? 1 : 0;
    for (int i = 0; i < n_cst && i < 9; ++i) l.term.cst[i] = cst[i];
    if (data) {   // DMatrix is column-major; the ABI wants row-major rows = nq*cell + q
        l.data.resize((size_t)data->size());
        for (int64_t r = 0; r < data->rows(); ++r)
            for (int64_t c = 0; c < data->cols(); ++c) l.data[(size_t)(r * data->cols() + c)] = (*data)(r, c);
    }
    return DifferentialExpr(std::move(l));
}
}   // namespace detail
template <typename Tag = FEM_HIP> DifferentialExpr laplacian() { return detail::leaf(FDAPDE_LAPLACIAN, nullptr, 0, nullptr); }
template <typename Tag = FEM_HIP> DifferentialExpr dt() { return detail::leaf(FDAPDE_DT, nullptr, 0, nullptr); }
template <typename Tag = FEM_HIP> DifferentialExpr reaction(double c) { return detail::leaf(FDAPDE_REACTION, &c, 1, nullptr); }
template <typename Tag = FEM_HIP, size_t N> DifferentialExpr advection(const std::array<double, N>& b) {
    return detail::leaf(FDAPDE_ADVECTION, b.data(), (int)N, nullptr);
}
// K row-major N x N
template <typename Tag = FEM_HIP, size_t NN> DifferentialExpr diffusion(const std::array<double, NN>& K) {
    return detail::leaf(FDAPDE_DIFFUSION, K.data(), (int)NN, nullptr);
}
// space-varying coefficients: one row per quadrature node, row nq*cell + q (Discretized*Field::forward, integrator.h:98-101)
template <typename Tag = FEM_HIP> DifferentialExpr reaction(const DMatrix<double>& c_q) { return detail::leaf(FDAPDE_REACTION, nullptr, 0, &c_q); }
template <typename Tag = FEM_HIP> DifferentialExpr advection(const DMatrix<double>& b_q) { return detail::leaf(FDAPDE_ADVECTION, nullptr, 0, &b_q); }
template <typename Tag = FEM_HIP> DifferentialExpr diffusion(const DMatrix<double>& K_q) { return detail::leaf(FDAPDE_DIFFUSION, nullptr, 0, &K_q); }

// Where a PDE computes: one HIP device (an int converts) or several -- PDE(domain, op, forcing, {0, 1, 2, 3}) shards the mesh over them behind
// the same interface (the reference's user holds one object in one thread, pde.h:58-105; include/fdapde_hip.h fdapde_ctx_create_multi)
struct device_list {
    std::vector<int> ids;
    device_list(int device) : ids {device} { }
    device_list(std::initializer_list<int> devices) : ids(devices) { }
    device_list(std::vector<int> devices) : ids(std::move(devices)) { }
};

// ---- PDE ---------------------------------------------------------------------------------------------------------------
// what PDE__::eval_basis returns (pde/pde.h:148)
struct EvalReturnType {
    SpMatrix<double> Psi;
    DVector<double> D;
};

template <typename D, typename E, typename F, typename S, typename... Ts> class PDE;

template <typename D, typename F, int R> class PDE<D, DifferentialExpr, F, FEM_HIP, fem_order<R>> {
   public:
    using SpaceDomainType = D;
    using OperatorType = DifferentialExpr;
    using ForcingType = F;
    static constexpr int M = D::local_dim, N = D::embed_dim;
    static_assert(std::is_same_v<F, DMatrix<double>> || std::is_same_v<F, ScalarField<N>>,
                  "forcing is DMatrix<double> (values at quadrature nodes) or ScalarField<N> (pde.h:49-51)");
    static_assert(R == 1 || R == 2, "LagrangianBasis::enumerate_dofs requires Order <= 2");

    explicit PDE(const D& domain, device_list device = 0) : domain_(domain) { open(device); }
    PDE(const D& domain, OperatorType diff_op, device_list device = 0) : domain_(domain), diff_op_(std::move(diff_op)) { open(device); }
    PDE(const D& domain, OperatorType diff_op, const ForcingType& forcing, device_list device = 0) :
        domain_(domain), diff_op_(std::move(diff_op)), forcing_data_(forcing) {
        open(device);
    }
    // space-time constructors (pde.h:66-72): times = the time grid [t_0 ... t_{m-1}], uniform step
    PDE(const D& domain, const DVector<double>& t, device_list device = 0) : domain_(domain) {
        time_domain_ = t;
        open(device);
    }
    PDE(const D& domain, const DVector<double>& t, OperatorType diff_op, device_list device = 0) : domain_(domain), diff_op_(std::move(diff_op)) {
        time_domain_ = t;
        open(device);
    }
    PDE(const D& domain, const DVector<double>& t, OperatorType diff_op, const ForcingType& forcing, device_list device = 0) :
        domain_(domain), diff_op_(std::move(diff_op)), forcing_data_(forcing) {
        time_domain_ = t;
        open(device);
    }
    // Copies are what the reference's type-erased handle lives on (make_pde copies the PDE to the heap, every handle copy copies it
    // again: pde.h:167-169, type_erasure.h:130-146).  A copy takes the host-side state (operator, forcing, boundary data, the matrices and
    // vectors the getters return) and SHARES the device context; whichever object next changes the context's problem state (init, solve,
    // a solver handle's compute) first leaves with a clone of it (fdapde_ctx_clone: include/fdapde_hip.hpp), so every object keeps
    // computing on the state it was copied with.  Assignment: not available, as in the reference (const members, pde.h:107-108).
    PDE(const PDE&) = default;
    PDE(PDE&&) = default;
    PDE& operator=(const PDE&) = delete;
    ~PDE() = default;

    // setters (pde.h:74-77)
    void set_forcing(const ForcingType& forcing_data) { forcing_data_ = forcing_data; }
    void set_differential_operator(OperatorType diff_op) { diff_op_ = std::move(diff_op); }
    void set_dirichlet_bc(const DMatrix<double>& data) { boundary_data_ = data; }
    void set_initial_condition(const DVector<double>& data) { initial_condition_ = data; }
    const DVector<double>& time_domain() const { return time_domain_; }
    const DVector<double>& initial_condition() const { return initial_condition_; }
    // getters (pde.h:79-100)
    const D& domain() const { return domain_; }
    OperatorType differential_operator() const { return diff_op_; }
    const ForcingType& forcing_data() const { return forcing_data_; }
    const DMatrix<double>& boundary_data() const { return boundary_data_; }
    int n_dofs() const { return (int)n_dofs_; }
    const DMatrix<double>& solution() const { return solution_; }
    const DMatrix<double>& force() const { return force_; }
    const SpMatrix<double>& stiff() const { return stiff_; }
    const SpMatrix<double>& mass() const { return mass_; }
    const DMatrix<int>& dofs() const { return dofs_; }
    const DMatrix<int>& boundary_dofs() const { return boundary_dofs_; }
    DMatrix<double> dof_coords() const { return dof_coords_; }
    DMatrix<double> quadrature_nodes() const {
        DMatrix<double> q((int64_t)nq_ * domain_.n_cells(), N);
        check(fdapde_quadrature_nodes(ctx_.get(), q.data()));
        return q;
    }
    bool is_init() const { return is_init_; }
    bool success() const { return success_; }
    fdapde_options& solver_options() { return opt_; }
    const fdapde_info& info() const { return info_; }

    // FEMSolverBase::init (fem_solver_base.h:104-139)
    void init() {
        auto terms = diff_op_.c_terms();
        if (terms.empty()) throw std::runtime_error("PDE::init: no differential operator set");
        fdapde_ctx* const ctx = ctx_.unique();   // (a copy that still shares the context leaves with its own clone here)
        check(fdapde_set_operator(ctx, (int32_t)terms.size(), terms.data()));
        const int64_t rows = (int64_t)nq_ * domain_.n_cells();
        if constexpr (std::is_same_v<F, DMatrix<double>>) {
            if (is_empty(forcing_data_)) {
                check(fdapde_set_forcing(ctx, nullptr, 0));
            } else {
                if (forcing_data_.rows() != rows) throw std::runtime_error("forcing data must have nq * n_cells rows");
                check(fdapde_set_forcing(ctx, forcing_data_.data(), (int32_t)forcing_data_.cols()));
            }
        } else {   // callable: evaluate at the mapped quadrature nodes (integrator.h:77-81)
            DMatrix<double> q = quadrature_nodes(), f(rows, 1);
            for (int64_t i = 0; i < rows; ++i) {
                std::array<double, N> x;
                for (int d = 0; d < N; ++d) x[(size_t)d] = q(i, d);
                f(i) = forcing_data_(x);
            }
            check(fdapde_set_forcing(ctx, f.data(), 1));
        }
        check(fdapde_init(ctx, &opt_));
        fetch_matrix(FDAPDE_MAT_STIFF, stiff_);
        fetch_matrix(FDAPDE_MAT_MASS, mass_);
        const int cols = std::is_same_v<F, DMatrix<double>> && !is_empty_forcing() ? (int)forcing_cols() : 1;
        force_.resize(n_dofs_ * cols, 1);
        check(fdapde_force(ctx, force_.data()));
        is_init_ = true, success_ = false;
    }
    // PDE::solve (pde.h:102-105): set_dirichlet_bc if boundary data is set, then the linear solve
    void solve() {
        if (!is_init_) throw std::runtime_error("solver must be initialized first!");
        if (diff_op_.is_parabolic() && !is_empty(time_domain_)) {
            solve_parabolic();
            return;
        }
        fdapde_ctx* const ctx = ctx_.unique();   // (the clone carries stiff_ / force_ / mass_: no second init)
        if (!is_empty(boundary_data_)) {
            if (boundary_data_.rows() != n_dofs_) throw std::runtime_error("dirichlet data must have n_dofs rows");
            check(fdapde_set_dirichlet(ctx, boundary_data_.data()));
        } else {
            check(fdapde_set_dirichlet(ctx, nullptr));
        }
        const int rc = fdapde_solve(ctx, &opt_, &info_);
        if (rc == FDAPDE_ENOCONV) {   // reference: success = false, no throw
            success_ = false;
            return;
        }
        check(rc);
        solution_.resize(n_dofs_, 1);
        check(fdapde_solution(ctx, solution_.data()));
        check(fdapde_matrix_values(ctx, FDAPDE_MAT_STIFF, stiff_.values.data()));   // row-zeroed if Dirichlet data was applied
        check(fdapde_force(ctx, force_.data()));
        success_ = true;
    }

    // "factor once, solve many" handle on one of this PDE's matrices: fdapde::SparseLU (utils/symbols.h:133-160)
    class SparseSolver {
       public:
        // compute(matrix): any matrix on the FEM pattern (e.g. pde.mass(), or a combination of stiff and mass values)
        void compute(const SpMatrix<double>& m, bool symmetric = false) {
            computed_ = fdapde_lin_compute(ctx_.get(), FDAPDE_MAT_STIFF, m.values.data(), symmetric ? 1 : 0) == FDAPDE_OK;
        }
        DMatrix<double> solve(const DMatrix<double>& b) const {
            if (!computed_) throw std::runtime_error("SparseSolver: compute() first");
            DMatrix<double> x(b.rows(), b.cols());
            fdapde_info info;
            const int rc = fdapde_lin_solve(ctx_.get(), nullptr, b.data(), (int32_t)b.cols(), x.data(), &info);
            if (rc != FDAPDE_OK) throw std::runtime_error(fdapde_last_error(ctx_.get()));
            return x;
        }
        explicit operator bool() const { return computed_; }
       private:
        friend class PDE;
        explicit SparseSolver(fdapde::hip::context_handle ctx) : ctx_(std::move(ctx)) { }
        fdapde::hip::context_handle ctx_;   // keeps the PDE's context alive; acts on it (the handle's matrix is no part of the PDE's state)
        bool computed_ = false;
    };
    // The handle works on the context this PDE holds NOW.  Copy the PDE afterwards and then change the original (init / solve / a setter): the
    // original leaves with a clone and the handle stays with the context the copy kept (fdapde_hip.hpp, observer()'s lifetime rule) -- take the
    // handle after the last copy, or from the object that is kept; attached_to(pde) tells.
    SparseSolver make_solver() {
        ctx_.unique();   // the handle works on THIS object's context, not on one still shared with a copy
        return SparseSolver(ctx_.observer());
    }
    bool owns_context_of(const SparseSolver& s) const { return s.ctx_.shared_with(ctx_); }
    fdapde_ctx* context() const { return ctx_.get(); }   // the C-ABI context (for entry points the facade does not wrap)

    // PDE__::eval_basis (pde/pde.h:149-158): 0 = Sampling::pointwise (locs: n_locs x N coordinates), 1 = Sampling::areal (locs:
    // n_subdomains x n_cells incidence matrix of 0 / 1).  Psi rows have sorted columns; duplicates are summed like setFromTriplets.
    using EvalReturnType = fdapde::amd::EvalReturnType;
    std::optional<EvalReturnType> eval_basis(int eval_type, const DMatrix<double>& locs) const {
        if (eval_type != 0 && eval_type != 1) return std::nullopt;
        const int64_t nc = domain_.cells().rows(), nb = dofs_.cols();
        std::vector<std::vector<std::pair<int32_t, double>>> rows;
        EvalReturnType out;
        if (eval_type == 0) {   // pointwise_evaluation::eval (lagrangian_basis.h:203-235)
            const int64_t nl = locs.rows();
            std::vector<int32_t> cell((size_t)nl);
            std::vector<double> val((size_t)(nl * nb));
            check(fdapde_eval_pointwise(ctx_.get(), nl, locs.data(), cell.data(), val.data()));
            rows.resize((size_t)nl);
            out.D = DVector<double>(nl, 1, 1.0);
            for (int64_t i = 0; i < nl; ++i) {
                if (cell[(size_t)i] < 0) continue;   // outside the domain: empty row
                for (int64_t h = 0; h < nb; ++h) rows[(size_t)i].push_back({dofs_(cell[(size_t)i], h), val[(size_t)(i * nb + h)]});
            }
        } else {   // areal_evaluation::eval (lagrangian_basis.h:238-283)
            if (locs.cols() != nc) throw std::runtime_error("eval_basis: the incidence matrix needs one column per cell");
            const int64_t ns = locs.rows();
            std::vector<double> meas((size_t)nc), pint((size_t)(nc * nb));
            check(fdapde_cell_integrals(ctx_.get(), meas.data(), pint.data()));
            rows.resize((size_t)ns);
            out.D = DVector<double>(ns, 1, 0.0);
            for (int64_t k = 0; k < ns; ++k) {
                for (int64_t e = 0; e < nc; ++e)
                    if (locs(k, e) == 1.0) {
                        out.D(k) += meas[(size_t)e];
                        for (int64_t h = 0; h < nb; ++h) rows[(size_t)k].push_back({dofs_(e, h), pint[(size_t)(e * nb + h)]});
                    }
                for (auto& cv : rows[(size_t)k]) cv.second /= out.D(k);
            }
        }
        SpMatrix<double>& P = out.Psi;
        P.n_rows = (int64_t)rows.size(), P.n_cols = n_dofs_;
        P.rowptr.assign(rows.size() + 1, 0);
        for (size_t i = 0; i < rows.size(); ++i) {
            auto& r = rows[i];
            std::stable_sort(r.begin(), r.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            for (size_t k = 0; k < r.size(); ++k) {
                if (k > 0 && r[k].first == r[k - 1].first)
                    P.values.back() += r[k].second;   // same DOF reached through several cells of a subdomain
                else
                    P.colidx.push_back(r[k].first), P.values.push_back(r[k].second);
            }
            P.rowptr[i + 1] = (int32_t)P.colidx.size();
        }
        return out;
    }

   private:
    // FEMLinearParabolicSolver::solve (fem_linear_parabolic_solver.h:37-72)
    void solve_parabolic() {
        const int64_t m = time_domain_.rows();
        if (m < 2) throw std::runtime_error("time domain needs at least two points");
        if (initial_condition_.rows() != n_dofs_) throw std::runtime_error("initial condition must have n_dofs rows");
        if (!is_empty(boundary_data_) && (boundary_data_.rows() != n_dofs_ || boundary_data_.cols() < m))
            throw std::runtime_error("dirichlet data must be n_dofs x n_times");
        solution_.resize(n_dofs_, m);
        fdapde_ctx* const ctx = ctx_.unique();
        const double dt_ = time_domain_(1) - time_domain_(0);
        const int rc = fdapde_solve_parabolic(ctx, &opt_, (int32_t)m, dt_, initial_condition_.data(),
                                              is_empty(boundary_data_) ? nullptr : boundary_data_.data(), solution_.data(), &info_);
        if (rc == FDAPDE_ENOCONV) {
            success_ = false;
            return;
        }
        check(rc);
        success_ = true;
    }
    bool is_empty_forcing() const {
        if constexpr (std::is_same_v<F, DMatrix<double>>) return is_empty(forcing_data_);
        return false;
    }
    int64_t forcing_cols() const {
        if constexpr (std::is_same_v<F, DMatrix<double>>) return forcing_data_.cols();
        return 1;
    }
    void check(int rc) const {
        if (rc == FDAPDE_OK) return;
        const std::string msg = ctx_ ? fdapde_last_error(ctx_.get()) : "";
        throw std::runtime_error(msg.empty() ? fdapde_status_string(rc) : msg);
    }
    void open(const device_list& device) {
        ctx_ = fdapde::hip::context_handle(device.ids);   // one id: a single-device context; several: the mesh sharded over them (fdapde_ctx_create_multi)
        fdapde_ctx* const ctx = ctx_.get();
        // hand the mesh over in the ABI's layouts (nodes column-major, cells row-major, boundary bytes)
        const int64_t nn = domain_.n_nodes(), nc = domain_.n_cells();
        std::vector<int32_t> cells((size_t)(nc * (M + 1)));
        for (int64_t c = 0; c < nc; ++c)
            for (int v = 0; v <= M; ++v) cells[(size_t)(c * (M + 1) + v)] = domain_.cells()(c, v);
        std::vector<uint8_t> bnd((size_t)nn);
        for (int64_t i = 0; i < nn; ++i) bnd[(size_t)i] = domain_.boundary_nodes()(i, 0) ? 1 : 0;
        check(fdapde_mesh_upload(ctx, M, N, nn, domain_.nodes().data(), nc, cells.data(), bnd.data()));
        check(fdapde_dofs_build(ctx, R, &n_dofs_));
        int64_t nnz = 0, n_edges = 0;
        check(fdapde_sizes(ctx, &n_dofs_, &nnz, &nb_, &nq_, &n_edges));
        std::vector<int32_t> dofs((size_t)(nc * nb_));
        std::vector<uint8_t> bd((size_t)n_dofs_);
        dof_coords_.resize(n_dofs_, N);
        check(fdapde_dofs_get(ctx, dofs.data(), bd.data(), dof_coords_.data()));
        dofs_.resize(nc, nb_), boundary_dofs_.resize(n_dofs_, 1);
        for (int64_t c = 0; c < nc; ++c)
            for (int j = 0; j < nb_; ++j) dofs_(c, j) = dofs[(size_t)(c * nb_ + j)];
        for (int64_t i = 0; i < n_dofs_; ++i) boundary_dofs_(i) = bd[(size_t)i];
        for (SpMatrix<double>* m : {&stiff_, &mass_}) {
            m->n_rows = m->n_cols = n_dofs_;
            m->rowptr.resize((size_t)n_dofs_ + 1), m->colidx.resize((size_t)nnz), m->values.assign((size_t)nnz, 0.0);
            check(fdapde_pattern_get(ctx, m->rowptr.data(), m->colidx.data()));
        }
    }
    void fetch_matrix(int which, SpMatrix<double>& m) { check(fdapde_matrix_values(ctx_.get(), which, m.values.data())); }

    const D& domain_;                 // must outlive the PDE (pde.h:107)
    OperatorType diff_op_;
    ForcingType forcing_data_ {};
    DVector<double> time_domain_ {};         // [t_0 ... t_{m-1}] for space-time problems (pde.h:108)
    DVector<double> initial_condition_ {};   // pde.h:111
    DMatrix<double> boundary_data_;
    fdapde::hip::context_handle ctx_;        // shared between copies until one of them changes it (include/fdapde_hip.hpp)
    fdapde_options opt_ {FDAPDE_SOLVER_AUTO, 0, 1e-10, FDAPDE_ASSEMBLY_ROWS, 0, 0};
    fdapde_info info_ {};
    int64_t n_dofs_ = 0;
    int32_t nb_ = 0, nq_ = 0;
    bool is_init_ = false, success_ = false;
    DMatrix<double> solution_, force_, dof_coords_;
    DMatrix<int> dofs_, boundary_dofs_;
    SpMatrix<double> stiff_, mass_;
};

// ---- the type-erased runtime face ---------------------------------------------------------------------------------------
// How downstream code holds a PDE without knowing its template arguments: fdapde::erase<fdapde::heap_storage, core::PDE__>, made by
// make_pde (pde/pde.h:117-169 on utils/type_erasure.h:124-160, 209-259).  Same spelling here -- erase<heap_storage, PDE__> -- with the
// 18 slots of pde.h:120-163 under the same names and signatures.  The reference builds the dispatch table by hand (an array of
// function pointers indexed by slot number, arguments re-cast at the call site); here the slots are virtual members of a private
// concept / model pair, which gives the same observable behaviour with checked argument types:
//   * the handle OWNS a heap copy of the PDE (heap_storage(const T&): new T(obj), type_erasure.h:130) -- the PDE handed to the
//     constructor is copied, later changes to the original are not seen;
//   * copying the handle deep-copies the PDE (type_erasure.h:136-146), moving it transfers ownership (148-160); an empty handle
//     converts to false (vtable_handler::operator bool, type_erasure.h:205);
//   * set_forcing / set_differential_operator are templates on the handle (pde.h:160-163): the argument must be of the PDE's own
//     ForcingType / OperatorType.  The reference re-casts the function pointer and has undefined behaviour otherwise; here a mismatch
//     throws std::runtime_error.  forcing_data() is slot 9's `const DMatrix<double>&` (pde.h:145): for a PDE whose forcing is a
//     callable it throws.
// Copies of a PDE share the device context until one of them computes (PDE's copy constructor above), so make_pde's temporary and
// every handle copy cost no device work by themselves.
struct heap_storage { };   // storage policy tag (type_erasure.h:124)
struct PDE__ { };          // interface tag (pde.h:118)
template <typename StorageType, typename... I> class erase;

template <> class erase<heap_storage, PDE__> {
   public:
    using EvalReturnType = fdapde::amd::EvalReturnType;   // PDE__::EvalReturnType (pde.h:148)
    erase() = default;
    template <typename T>
        requires(!std::is_same_v<std::decay_t<T>, erase> && std::is_copy_constructible_v<std::decay_t<T>>)
    erase(const T& pde) : p_(std::make_unique<model<std::decay_t<T>>>(pde)) { }
    erase(const erase& other) : p_(other.p_ ? other.p_->clone() : nullptr) { }
    erase(erase&& other) noexcept = default;
    erase& operator=(const erase& other) {
        if (this != &other) p_ = other.p_ ? other.p_->clone() : nullptr;
        return *this;
    }
    erase& operator=(erase&& other) noexcept = default;
    template <typename T>
        requires(!std::is_same_v<std::decay_t<T>, erase> && std::is_copy_constructible_v<std::decay_t<T>>)
    erase& operator=(const T& pde) {
        p_ = std::make_unique<model<std::decay_t<T>>>(pde);
        return *this;
    }
    operator bool() const { return p_ != nullptr; }
    // slots 0-1
    void init() { get().init(); }
    void solve() { get().solve(); }
    // slots 2-11: getters
    const DMatrix<double>& solution() const { return get().solution(); }
    const DMatrix<double>& force() const { return get().force(); }
    const SpMatrix<double>& stiff() const { return get().stiff(); }
    const SpMatrix<double>& mass() const { return get().mass(); }
    DMatrix<double> quadrature_nodes() const { return get().quadrature_nodes(); }
    int n_dofs() const { return get().n_dofs(); }
    DMatrix<double> dof_coords() const { return get().dof_coords(); }
    const DMatrix<double>& forcing_data() const { return get().forcing_data(); }
    const DVector<double>& time_domain() const { return get().time_domain(); }
    const DVector<double>& initial_condition() const { return get().initial_condition(); }
    // slots 12-13: eval_type 0 = Sampling::pointwise, 1 = Sampling::areal, anything else std::nullopt (pde.h:149-158)
    std::optional<EvalReturnType> eval_basis(int eval_type, const DMatrix<double>& locs) const { return get().eval_basis(eval_type, locs); }
    // slots 14-17: setters
    template <typename ForcingType> void set_forcing(const ForcingType& data) { get().set_forcing(&data, typeid(ForcingType)); }
    void set_dirichlet_bc(const DMatrix<double>& data) { get().set_dirichlet_bc(data); }
    void set_initial_condition(const DVector<double>& data) { get().set_initial_condition(data); }
    template <typename E> void set_differential_operator(E diff_op) { get().set_differential_operator(&diff_op, typeid(E)); }
    // beyond the reference's slots: the solver flags of the held PDE (fem_solver_base.h:61-62; the reference's handle cannot reach them)
    bool success() const { return get().success(); }

   private:
    struct concept_t {
        virtual ~concept_t() = default;
        virtual std::unique_ptr<concept_t> clone() const = 0;
        virtual void init() = 0;
        virtual void solve() = 0;
        virtual const DMatrix<double>& solution() const = 0;
        virtual const DMatrix<double>& force() const = 0;
        virtual const SpMatrix<double>& stiff() const = 0;
        virtual const SpMatrix<double>& mass() const = 0;
        virtual DMatrix<double> quadrature_nodes() const = 0;
        virtual int n_dofs() const = 0;
        virtual DMatrix<double> dof_coords() const = 0;
        virtual const DMatrix<double>& forcing_data() const = 0;
        virtual const DVector<double>& time_domain() const = 0;
        virtual const DVector<double>& initial_condition() const = 0;
        virtual std::optional<EvalReturnType> eval_basis(int eval_type, const DMatrix<double>& locs) const = 0;
        virtual void set_forcing(const void* data, const std::type_info& type) = 0;
        virtual void set_dirichlet_bc(const DMatrix<double>& data) = 0;
        virtual void set_initial_condition(const DVector<double>& data) = 0;
        virtual void set_differential_operator(const void* diff_op, const std::type_info& type) = 0;
        virtual bool success() const = 0;
    };
    template <typename T> struct model final : concept_t {
        T pde;
        explicit model(const T& obj) : pde(obj) { }   // new T(obj): the deep copy of heap_storage (type_erasure.h:130)
        std::unique_ptr<concept_t> clone() const override { return std::make_unique<model>(pde); }
        void init() override { pde.init(); }
        void solve() override { pde.solve(); }
        const DMatrix<double>& solution() const override { return pde.solution(); }
        const DMatrix<double>& force() const override { return pde.force(); }
        const SpMatrix<double>& stiff() const override { return pde.stiff(); }
        const SpMatrix<double>& mass() const override { return pde.mass(); }
        DMatrix<double> quadrature_nodes() const override { return pde.quadrature_nodes(); }
        int n_dofs() const override { return pde.n_dofs(); }
        DMatrix<double> dof_coords() const override { return pde.dof_coords(); }
        const DMatrix<double>& forcing_data() const override {
            if constexpr (std::is_same_v<typename T::ForcingType, DMatrix<double>>) return pde.forcing_data();
            else throw std::runtime_error("PDE__::forcing_data: the forcing of this PDE is a callable, not a matrix");
        }
        const DVector<double>& time_domain() const override { return pde.time_domain(); }
        const DVector<double>& initial_condition() const override { return pde.initial_condition(); }
        std::optional<EvalReturnType> eval_basis(int eval_type, const DMatrix<double>& locs) const override { return pde.eval_basis(eval_type, locs); }
        void set_forcing(const void* data, const std::type_info& type) override {
            if (type != typeid(typename T::ForcingType)) throw std::runtime_error("PDE__::set_forcing: argument is not of the PDE's ForcingType");
            pde.set_forcing(*static_cast<const typename T::ForcingType*>(data));
        }
        void set_dirichlet_bc(const DMatrix<double>& data) override { pde.set_dirichlet_bc(data); }
        void set_initial_condition(const DVector<double>& data) override { pde.set_initial_condition(data); }
        void set_differential_operator(const void* diff_op, const std::type_info& type) override {
            if (type != typeid(typename T::OperatorType)) throw std::runtime_error("PDE__::set_differential_operator: argument is not of the PDE's OperatorType");
            pde.set_differential_operator(*static_cast<const typename T::OperatorType*>(diff_op));
        }
        bool success() const override { return pde.success(); }
    };
    concept_t& get() {
        if (!p_) throw std::runtime_error("empty PDE handle");
        return *p_;
    }
    const concept_t& get() const {
        if (!p_) throw std::runtime_error("empty PDE handle");
        return *p_;
    }
    std::unique_ptr<concept_t> p_;
};

// factory (pde.h:167-169): make_pde<D, E, F, S, Ts...>(args...) constructs PDE<D, E, F, S, Ts...>(args...) and hands back the handle
template <typename... Args_, typename... Args> erase<heap_storage, PDE__> make_pde(Args&&... args) {
    return erase<heap_storage, PDE__>(PDE<Args_...>(std::forward<Args>(args)...));
}

// ---- Integrator<FEM, M, R>::integrate(mesh, f) (utils/integration/integrator.h:61-69) ----------------------------------------------
// Integral of a callable over the whole triangulation with the quadrature rule the order-R assembly uses.  On the device it is the load
// sweep: f sampled at the mapped quadrature nodes of every cell, b_i = sum_e |e| sum_q w_q f(x_q) psi_i(x_q); the Lagrangian basis sums
// to one at every point, so sum_i b_i is the quadrature of f itself.  (The per-element form, integrate_cell, is what the assembly does
// inside that sweep and has no entry of its own.)
template <typename S, int M, int R> class Integrator;
template <int M, int R> class Integrator<FEM_HIP, M, R> {
   public:
    explicit Integrator(int device = 0) : device_(device) { }
    template <int N, typename ExprType> double integrate(const Triangulation<M, N>& m, const ExprType& f) const {
        using Field = ScalarField<N>;
        PDE<Triangulation<M, N>, DifferentialExpr, Field, FEM_HIP, fem_order<R>> sweep(
          m, laplacian<FEM_HIP>(), Field([&f](const std::array<double, N>& x) { return (double)f(x); }), device_);
        sweep.init();
        const DMatrix<double>& b = sweep.force();
        double value = 0;
        for (int64_t i = 0; i < b.size(); ++i) value += b(i);
        return value;
    }
   private:
    int device_;
};

}   // namespace amd
}   // namespace fdapde

#if __has_include(<Eigen/Sparse>)
#include <Eigen/Sparse>
namespace fdapde {
namespace amd {
// the reference's SpMatrix<double> (column-major Eigen::SparseMatrix, utils/symbols.h:33) from the CSR the device exports
inline Eigen::SparseMatrix<double> to_eigen(const SpMatrix<double>& m) {
    Eigen::Map<const Eigen::SparseMatrix<double, Eigen::RowMajor, int32_t>> view(
      m.n_rows, m.n_cols, m.nonZeros(), m.rowptr.data(), m.colidx.data(), m.values.data());
    return Eigen::SparseMatrix<double>(view);
}
}   // namespace amd
}   // namespace fdapde
#endif
#endif   // FDAPDE_AMD_PDE_H
