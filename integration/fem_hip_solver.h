// fem_hip_solver.h -- REFERENCE-SIDE binding of the MI355X path: drop this file into fdaPDE/finite_elements/solvers/ of
// fdaPDE-core (reference @ 2024-10-16), include it from fdaPDE/finite_elements.h, and link libfdapde_hip.so.  User code then
// changes one template argument:   PDE<decltype(mesh), decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde(mesh, L);
//
// The operator expression stays one built from the reference's FEM leaves (laplacian<FEM>(), advection<FEM>(b) ...: their weak forms are
// what this file reads); FEM_HIP is the STRATEGY argument only -- it selects the solver type, exactly what pde_solver_selector is for.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE: it needs the reference tree and Eigen 3.4, which the image lacks (Oracle).
// The same binding written on Eigen-free containers -- include/fdapde_amd/pde.h -- is what tests/cpp/fem_pde_test.cpp compiles and
// runs; this file is that binding expressed in the reference's own types, desk-checked member by member against the headers it
// includes (INTEGRATION.md section 2c lists every use with the file:line it was checked against).
// Copies: the reference's type-erased PDE handle deep-copies the PDE with its solver (make_pde, pde/pde.h:167-169 ->
// heap_storage(const T&): new T(obj), utils/type_erasure.h:130; handle copies: 136-146), so the solver types below are COPYABLE: a copy
// takes FEMSolverBase's members (basis, matrices, flags) as any copy does and SHARES the device context; the first object that is about
// to change the context's problem state leaves with a clone of it (fdapde::hip::context_handle, include/fdapde_hip.hpp ->
// fdapde_ctx_clone).  make_pde<D, E, F, FEM_HIP, fem_order<R>>(...) therefore instantiates and costs no device work by itself.
// Reference members it plugs into / replaces:
//   pde_solver_selector<S, ...>            fdaPDE/pde/symbols.h:36, finite_elements/solvers/fem_solver_selector.h:29-33
//   FEMSolverBase::init                    finite_elements/solvers/fem_solver_base.h:104-139
//   FEMSolverBase::set_dirichlet_bc        fem_solver_base.h:142-155
//   FEMLinearEllipticSolver::solve         finite_elements/solvers/fem_linear_elliptic_solver.h:34-50
//   FEMLinearParabolicSolver::solve        finite_elements/solvers/fem_linear_parabolic_solver.h:37-72 (+ set_deltaT, line 34)
// The selector switches on is_parabolic<E> exactly as the reference's own (fem_solver_selector.h:29-33): an operator with a dT()
// leaf under tag FEM_HIP gets the time-stepping solver, every other one the elliptic solver.
#ifndef __FEM_HIP_SOLVER_H__
#define __FEM_HIP_SOLVER_H__

#include <array>
#include <stdexcept>
#include <vector>

#include <fdapde_hip.h>     // this repository's C ABI (include/)
#include <fdapde_hip.hpp>   // copy-on-write owner of a fdapde_ctx (header-only, no Eigen)

#include "../../pde/differential_operators.h"   // is_parabolic<E>
#include "../../pde/symbols.h"
#include "../../utils/traits.h"                  // switch_type
#include "../../utils/symbols.h"
#include "fem_solver_base.h"

namespace fdapde {
namespace core {

struct FEM_HIP { };   // strategy tag, next to FEM and SPLINE

namespace hip_detail {

// ---- operator expression -> fdapde_term[] -------------------------------------------------------------------------------------
// The nodes of the reference's expression tree keep their operands private (DifferentialBinOp::op1_, op2_, f_;
// DifferentialNegateOp::op_; Diffusion::K_ ..., pde/differential_expressions.h:57-135), so the tree is not walked.  It does not have to be:
// E.integrate(mem_buffer) is, for every expression the algebra can build, the integrand
//     -(J^-T grad psi_i) . K (J^-T grad psi_j)  +  psi_i (J^-T grad psi_j) . b  +  c psi_i psi_j
// with K, b, c the SUMS of the scaled leaves (laplacian.h:43, diffusion.h:54, advection.h:55, reaction.h:52, dt.h:34-36 -> 0).  The
// binding evaluates that integrand on probe functions through the very buffers the assembler uses (fem_assembler.h:62-72):
// invJ = I, point p = 0, psi in {1, x_1 .. x_M}:
//     psi_i = x_k, psi_j = x_l  ->  -K[k][l]          psi_i = 1, psi_j = x_l  ->  b[l]          psi_i = psi_j = 1  ->  c
// For space-varying coefficients the probes are repeated after forward(row) for every quadrature row (integrator.h:98-101).
template <int M, int R> MultivariatePolynomial<M, R> probe_polynomial(int axis /* -1: the constant 1, k: the coordinate x_k */) {
    using Poly = MultivariatePolynomial<M, R>;
    std::array<double, ct_binomial_coefficient(R + M, R)> coeff {};
    for (std::size_t m = 0; m < coeff.size(); ++m) {   // find the monomial in the public exponent table
        int degree = 0, at = -1;
        for (int d = 0; d < M; ++d) {
            degree += Poly::poly_table[m][d];
            if (Poly::poly_table[m][d] == 1) at = d;
        }
        if ((axis < 0 && degree == 0) || (axis >= 0 && degree == 1 && at == axis)) coeff[m] = 1.0;
    }
    return Poly(coeff);
}

struct CollapsedOperator {
    std::vector<fdapde_term> terms;          // at most: one diffusion (Laplacian folded in), one advection, one reaction leaf
    std::vector<double> K, b, c;             // per-quadrature-row data the terms point into (space-varying operators only)
};

template <int M, int N, int R, typename E> CollapsedOperator to_terms(const E& op, std::int64_t n_quadrature_rows) {
    static_assert(M == N, "manifold domains are not on the accelerated path");
    using Poly = MultivariatePolynomial<M, R>;
    using Nabla = decltype(std::declval<Poly>().derive());
    Poly psi_i, psi_j;
    Nabla nabla_i, nabla_j;
    Matrix<M, N, M> invJ(SMatrix<N, M>::Identity());   // Matrix<N_, M_, K_> wraps an SMatrix<M_, K_> (fields/matrix_expressions.h:191-203); M == N here
    DVector<double> f;
    auto mem_buffer = std::make_tuple(ScalarPtr(&psi_i), ScalarPtr(&psi_j), VectorPtr(&nabla_i), VectorPtr(&nabla_j), MatrixPtr(&invJ), &f);
    auto weak_form = op.integrate(mem_buffer);
    const SVector<M> origin = SVector<M>::Zero();
    std::array<Poly, M + 1> probe;           // probe[0] = 1, probe[1 + k] = x_k
    for (int k = -1; k < M; ++k) probe[k + 1] = probe_polynomial<M, R>(k);
    auto eval = [&](int i, int j) {
        psi_i = probe[i], psi_j = probe[j], nabla_i = psi_i.derive(), nabla_j = psi_j.derive();
        return weak_form(origin);
    };
    constexpr bool varying = E::is_space_varying;
    const std::int64_t rows = varying ? n_quadrature_rows : 1;
    CollapsedOperator out;
    out.K.resize(rows * M * M), out.b.resize(rows * M), out.c.resize(rows);
    bool has_K = false, has_b = false, has_c = false;
    for (std::int64_t r = 0; r < rows; ++r) {
        if constexpr (varying) weak_form.forward(r);
        for (int k = 0; k < M; ++k)
            for (int l = 0; l < M; ++l) has_K |= (out.K[(r * M + k) * M + l] = -eval(1 + k, 1 + l)) != 0.0;
        for (int l = 0; l < M; ++l) has_b |= (out.b[r * M + l] = eval(0, 1 + l)) != 0.0;
        has_c |= (out.c[r] = eval(0, 0)) != 0.0;
    }
    auto leaf = [&](int kind, const std::vector<double>& data, int width) {
        fdapde_term t {};
        t.kind = kind, t.coef = 1.0, t.space_varying = varying ? 1 : 0;
        if (varying) t.data = data.data();
        else for (int e = 0; e < width; ++e) t.cst[e] = data[e];
        out.terms.push_back(t);
    };
    if (has_K) leaf(FDAPDE_DIFFUSION, out.K, M * M);
    if (has_b) leaf(FDAPDE_ADVECTION, out.b, M);
    if (has_c) leaf(FDAPDE_REACTION, out.c, 1);
    if (out.terms.empty()) leaf(FDAPDE_DT, out.c, 0);   // e.g. dT alone: integrates to zero (dt.h:34-36)
    return out;
}

// stiff() / mass(): CSR (pattern_get + matrix_values, the reference's DOF numbering) -> the reference's column-major SpMatrix
inline void fetch(fdapde_ctx* ctx, int which, std::int64_t n_dofs, SpMatrix<double>& dst) {
    std::int64_t nnz = 0;
    fdapde_sizes(ctx, nullptr, &nnz, nullptr, nullptr, nullptr);
    std::vector<int32_t> rowptr(n_dofs + 1), colidx(nnz);
    std::vector<double> values(nnz);
    fdapde_pattern_get(ctx, rowptr.data(), colidx.data());
    if (fdapde_matrix_values(ctx, which, values.data()) != FDAPDE_OK) throw std::runtime_error(fdapde_last_error(ctx));
    dst = Eigen::Map<const Eigen::SparseMatrix<double, Eigen::RowMajor, int>>(n_dofs, n_dofs, nnz, rowptr.data(), colidx.data(), values.data());
}

}   // namespace hip_detail

// what both solvers share: the device context, init (assembly) and the Dirichlet data hand-over
template <typename D, typename E, typename F, typename... Ts>
struct FEMHipSolverBase : public FEMSolverBase<D, E, F, Ts...> {   // keeps basis_, integrator_, the getters and the flags
    using Base = FEMSolverBase<D, E, F, Ts...>;
    static constexpr int M = D::local_dim, N = D::embed_dim, R = Base::fem_order;
    fdapde::hip::context_handle ctx_;   // shared between copies of the solver until one of them changes it (copy-on-write)

    FEMHipSolverBase() = default;                                             // (FEMSolverBase is default constructible too, fem_solver_base.h:48)
    FEMHipSolverBase(const D& domain) : Base(domain), ctx_(fdapde::hip::default_devices()) {   // one device, or the mesh sharded over several
        // (fdapde::hip::set_default_devices / FDAPDE_HIP_DEVICES: pde.h:58 hands the solver nothing but the domain); throws without a HIP device: no CPU fallback
        DMatrix<int, Eigen::RowMajor> cells = domain.cells();                 // row-major int32 0-based (triangulation.h:64, 120)
        std::vector<uint8_t> bnd(domain.n_nodes());
        for (int i = 0; i < domain.n_nodes(); ++i) bnd[i] = domain.is_node_on_boundary(i);   // triangulation.h:62
        check(fdapde_mesh_upload(ctx_.get(), M, N, domain.n_nodes(), domain.nodes().data() /* column-major, triangulation.h:63, 119 */,
                                 domain.n_cells(), cells.data(), bnd.data()));
        std::int64_t n = 0;
        check(fdapde_dofs_build(ctx_.get(), R, &n));                          // same numbering as basis_.dofs(), bit for bit
    }
    // copy / move: the members' own (heap_storage copies the PDE, the PDE copies its solver_: pde.h:112, type_erasure.h:130); the device
    // context is shared until the copy or the original computes
    FEMHipSolverBase(const FEMHipSolverBase&) = default;
    FEMHipSolverBase(FEMHipSolverBase&&) = default;
    FEMHipSolverBase& operator=(const FEMHipSolverBase&) = default;
    FEMHipSolverBase& operator=(FEMHipSolverBase&&) = default;
    ~FEMHipSolverBase() = default;

    template <typename PDE> void init(const PDE& pde) {                       // replaces fem_solver_base.h:104-139
        static_assert(is_pde<PDE>::value, "pde is not a valid PDE object");
        this->n_dofs_ = this->basis_.size(), this->dofs_ = this->basis_.dofs(), this->boundary_dofs_ = this->basis_.boundary_dofs();
        const std::int64_t rows = (std::int64_t)this->integrator_.num_nodes() * pde.domain().n_cells();
        auto op = hip_detail::to_terms<M, N, R>(pde.differential_operator(), rows);
        fdapde_ctx* const ctx = ctx_.unique();                                // (a copy that still shares the context leaves with a clone here)
        check(fdapde_set_operator(ctx, (int32_t)op.terms.size(), op.terms.data()));
        // forcing at the quadrature nodes, row nq * cell + q (integrator.h:85); a callable forcing is sampled at quadrature_nodes()
        // first, exactly what Integrator::integrate does with it (integrator.h:80-81).  Columns: the reference discretises column 0
        // always and the others only for a parabolic operator (fem_solver_base.h:118-128) -- the same count is handed over here
        DMatrix<double> fq;
        if constexpr (std::is_base_of<ScalarBase, F>::value) {
            DMatrix<double> qn = this->integrator_.quadrature_nodes(pde.domain());
            fq.resize(qn.rows(), 1);
            for (int i = 0; i < qn.rows(); ++i) fq(i, 0) = pde.forcing_data()(SVector<N>(qn.row(i)));
        } else
            fq = pde.forcing_data();
        const int32_t cols = is_parabolic<E>::value ? (int32_t)fq.cols() : 1;
        check(fdapde_set_forcing(ctx, fq.data(), cols));                    // DMatrix is column-major: the leading `cols` columns
        check(fdapde_init(ctx, nullptr));
        hip_detail::fetch(ctx, FDAPDE_MAT_STIFF, this->n_dofs_, this->stiff_);
        hip_detail::fetch(ctx, FDAPDE_MAT_MASS, this->n_dofs_, this->mass_);
        this->force_.resize(this->n_dofs_ * fq.cols(), 1);                    // (the reference sizes it n * m whatever the operator, line 119)
        this->force_.setZero();
        check(fdapde_force(ctx, this->force_.data()));                      // fills the first n * cols entries
        this->is_init = true;
    }
    template <typename PDE> void set_dirichlet_bc(const PDE& pde) {           // replaces fem_solver_base.h:142-155
        if (!this->is_init) throw std::runtime_error("solver must be initialized first!");
        if constexpr (!is_parabolic<E>::value) {
            if (pde.boundary_data().rows() != this->n_dofs_) throw std::runtime_error("FEM_HIP: boundary data must have one row per DOF");
            check(fdapde_set_dirichlet(ctx_.unique(), pde.boundary_data().data()));   // column 0, indexed by DOF id (fem_solver_base.h:152)
        }
        // (parabolic: the reference's set_dirichlet_bc touches column 0 of the data and the steady matrix only, while the time stepper
        //  imposes column i + 1 at step i itself, fem_linear_parabolic_solver.h:51-55,64-67 -- the data travel with solve() below)
    }
   protected:
    void check(int rc) const {
        if (rc != FDAPDE_OK) throw std::runtime_error(fdapde_last_error(ctx_.get()));
    }
};

template <typename D, typename E, typename F, typename... Ts>
struct FEMHipEllipticSolver : public FEMHipSolverBase<D, E, F, Ts...> {
    using Base = FEMHipSolverBase<D, E, F, Ts...>;
    FEMHipEllipticSolver(const D& domain) : Base(domain) { }
    template <typename PDE> void solve(const PDE&) {                          // replaces fem_linear_elliptic_solver.h:34-50
        if (!this->is_init) throw std::runtime_error("solver must be initialized first!");
        fdapde_info info;
        fdapde_ctx* const ctx = this->ctx_.unique();                          // (the clone of a shared context carries stiff_ / force_: no second init)
        const int rc = fdapde_solve(ctx, nullptr, &info);
        if (rc == FDAPDE_ENOCONV) {                                           // <-> the reference's success = false (lines 42-45)
            this->success = false;
            return;
        }
        this->check(rc);
        this->solution_.resize(this->n_dofs_, 1);
        this->check(fdapde_solution(ctx, this->solution_.data()));
        hip_detail::fetch(ctx, FDAPDE_MAT_STIFF, this->n_dofs_, this->stiff_);   // the row-zeroed matrix the reference leaves behind
        this->check(fdapde_force(ctx, this->force_.data()));                 // boundary rows = g
        this->success = true;
    }
};

// FEMLinearParabolicSolver's members mirrored: deltaT_ / set_deltaT (lines 30-34), solve (37-72).  The reference factorises
// K = M / dt + A once with SparseLU and back-substitutes per step; fdapde_solve_parabolic scales K once and runs a warm-started
// Jacobi-PCG / BiCGStab per step on the device, with the same time loop and the same Dirichlet rule.
template <typename D, typename E, typename F, typename... Ts>
struct FEMHipParabolicSolver : public FEMHipSolverBase<D, E, F, Ts...> {
   private:
    double deltaT_ = 1e-2;
   public:
    using Base = FEMHipSolverBase<D, E, F, Ts...>;
    FEMHipParabolicSolver(const D& domain) : Base(domain) { }
    void set_deltaT(double deltaT) { deltaT_ = deltaT; }

    template <typename PDE> void solve(const PDE& pde) {
        static_assert(is_pde<PDE>::value, "pde is not a valid PDE object");
        if (!this->is_init) throw std::runtime_error("solver must be initialized first!");
        this->set_deltaT(pde.time_domain()[1] - pde.time_domain()[0]);       // line 42
        const std::size_t n = this->n_dofs();                                 // degrees of freedom in space
        const std::size_t m = pde.forcing_data().cols();                      // time points (line 44)
        this->solution_.resize(n, m);
        const DVector<double> u0 = pde.initial_condition();                   // column 0 of the solution (line 46)
        // The reference's loop ALWAYS zeroes the boundary rows of K and imposes boundary_data()(dof, i + 1) at step i (lines 51-55, 64-67),
        // whether or not set_dirichlet_bc ran: so the data are imposed whenever they are there -- n_dofs x (at least m); data of another
        // shape are an error here (the reference would index out of range).  Only EMPTY boundary data -- which the reference cannot run
        // on a mesh with boundary DOFs at all -- gives natural conditions.
        const DMatrix<double>& g = pde.boundary_data();
        const bool with_bc = !is_empty(g);
        if (with_bc && (g.rows() != (Eigen::Index)n || g.cols() < (Eigen::Index)m))
            throw std::runtime_error("FEM_HIP: boundary data of a parabolic problem must be n_dofs x n_times");
        fdapde_info info;
        const int rc = fdapde_solve_parabolic(this->ctx_.unique(), nullptr, (int32_t)m, deltaT_, u0.data(), with_bc ? g.data() : nullptr,
                                              this->solution_.data() /* column-major n x m */, &info);
        if (rc == FDAPDE_ENOCONV) {                                           // <-> solver.info() != Eigen::Success (lines 57-60)
            this->success = false;
            return;
        }
        this->check(rc);
        this->success = true;
    }
};

// selects the solver type from the operator, as fem_solver_selector.h:29-33 does for tag FEM
template <typename D, typename E, typename F, typename... Ts> struct pde_solver_selector<FEM_HIP, D, E, F, Ts...> {
    using type = typename switch_type<
      switch_type_case<!is_parabolic<E>::value, FEMHipEllipticSolver <D, E, F, Ts...>>,
      switch_type_case< is_parabolic<E>::value, FEMHipParabolicSolver<D, E, F, Ts...>> >::type;
};

}   // namespace core
}   // namespace fdapde

#endif   // __FEM_HIP_SOLVER_H__
