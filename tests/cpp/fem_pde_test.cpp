// fem_pde_test.cpp -- the reference's elliptic fem_pde_test cases (test/src/fem_pde_test.cpp:43-212) re-expressed against the
// header-only facade include/fdapde_amd/pde.h, plus the golden local-matrix check of test/src/fem_operators_test.cpp:41-100
// read back through stiff().  Same meshes, same exact solutions, same gates.  Runs on a real MI355X (pytest -m gpu).
//
// usage: fem_pde_test <path to tests/golden/mesh>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

#include <unistd.h>

#include "fdapde_amd/pde.h"
#include "fdapde_amd/io.h"
#include "fdapde_amd/linear_algebra.h"

using namespace fdapde::amd;

static int failures = 0, checks = 0;
#define EXPECT_TRUE(cond)                                                                 \
    do {                                                                                  \
        ++checks;                                                                         \
        if (!(cond)) { ++failures; std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)
#define TEST(suite, name) static void suite##_##name()
#define RUN(suite, name)                                  \
    do {                                                  \
        std::printf("[ RUN      ] %s.%s\n", #suite, #name); \
        int before = failures;                            \
        suite##_##name();                                 \
        std::printf("[ %s ] %s.%s\n", failures == before ? "      OK" : " FAILED ", #suite, #name); \
    } while (0)

constexpr double DOUBLE_TOLERANCE = 1e-7;   // test/src/utils/constants.h:11
static std::string MESH_PATH;

static bool almost_equal(double a, double b, double eps = DOUBLE_TOLERANCE) {   // test/src/utils/utils.h:33-36
    return std::fabs(a - b) < eps || std::fabs(a - b) < std::fmax(std::fabs(a), std::fabs(b)) * eps;
}

// the product's reader and loader (include/fdapde_amd/io.h), with the reference test suite's calling conventions
template <typename T> DMatrix<T> read_csv(const std::string& file) { return fdapde::amd::CSVReader<T>().parse_file(file); }
template <int M, int N> struct FixtureMesh : fdapde::amd::MeshLoader<M, N> {
    explicit FixtureMesh(const std::string& id) : fdapde::amd::MeshLoader<M, N>(MESH_PATH, id) { }
};

template <typename PDE_, typename Fn> static double l2_error(PDE_& pde, Fn solution_expr) {
    DMatrix<double> nodes = pde.dof_coords();
    DMatrix<double> e2(nodes.rows(), 1);
    for (int64_t i = 0; i < nodes.rows(); ++i) {
        const double err = solution_expr(nodes.row3(i)) - pde.solution()(i);
        e2(i) = err * err;
    }
    DMatrix<double> Me = pde.mass() * e2;   // (mass * err.cwiseProduct(err)).sum(), fem_pde_test.cpp:73
    double s = 0;
    for (int64_t i = 0; i < Me.rows(); ++i) s += Me(i);
    return s;
}
template <typename PDE_, typename Fn> static DMatrix<double> eval_at_dofs(PDE_& pde, Fn f) {
    DMatrix<double> nodes = pde.dof_coords(), out(nodes.rows(), 1);
    for (int64_t i = 0; i < nodes.rows(); ++i) out(i) = f(nodes.row3(i));
    return out;
}

// fem_pde_test.cpp:43-75
TEST(fem_pde_test, laplacian_isotropic_order1) {
    auto solution_expr = [](std::array<double, 3> x) -> double { return x[0] + x[1]; };
    FixtureMesh<2, 2> unit_square("unit_square");
    auto L = -laplacian<FEM_HIP>();
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(unit_square.mesh, L);
    pde_.set_dirichlet_bc(eval_at_dofs(pde_, solution_expr));
    DMatrix<double> quadrature_nodes = pde_.quadrature_nodes();
    pde_.set_forcing(DMatrix<double>::Zero(quadrature_nodes.rows(), 1));
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    EXPECT_TRUE(pde_.n_dofs() == 3600);
    EXPECT_TRUE(l2_error(pde_, solution_expr) < DOUBLE_TOLERANCE);
}
// fem_pde_test.cpp:78-107
TEST(fem_pde_test, laplacian_isotropic_order2_callable_force) {
    auto solution_expr = [](std::array<double, 3> x) -> double { return 1. - x[0] * x[0] - x[1] * x[1]; };
    ScalarField<2> forcing([](const std::array<double, 2>&) -> double { return 4.0; });
    FixtureMesh<2, 2> unit_square("unit_square");
    auto L = -laplacian<FEM_HIP>();
    PDE<Triangulation<2, 2>, decltype(L), ScalarField<2>, FEM_HIP, fem_order<2>> pde_(unit_square.mesh, L, forcing);
    pde_.set_dirichlet_bc(eval_at_dofs(pde_, solution_expr));
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    EXPECT_TRUE(pde_.n_dofs() == 14161);
    EXPECT_TRUE(l2_error(pde_, solution_expr) < DOUBLE_TOLERANCE);
}
struct AdvDiff {
    static constexpr double pi = 3.14159265358979323846;
    double alpha_ = 1.0, gamma_ = pi, lambda1, lambda2, p_;
    AdvDiff() {
        lambda1 = -alpha_ / 2 - std::sqrt((alpha_ / 2) * (alpha_ / 2) + pi * pi);
        lambda2 = -alpha_ / 2 + std::sqrt((alpha_ / 2) * (alpha_ / 2) + pi * pi);
        p_ = (1 - std::exp(lambda2)) / (std::exp(lambda1) - std::exp(lambda2));
    }
    double solution(std::array<double, 3> x) const {
        return -gamma_ / (pi * pi) * (p_ * std::exp(lambda1 * x[0]) + (1 - p_) * std::exp(lambda2 * x[0]) - 1.) * std::sin(pi * x[1]);
    }
    double forcing(double y) const { return gamma_ * std::sin(pi * y); }
};
// fem_pde_test.cpp:113-166
TEST(fem_pde_test, advection_diffusion_isotropic_order1) {
    AdvDiff ad;
    std::array<double, 2> beta_ {-ad.alpha_, 0.};
    auto L = -laplacian<FEM_HIP>() + advection<FEM_HIP>(beta_);
    FixtureMesh<2, 2> unit_square("unit_square");
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(unit_square.mesh);
    pde_.set_differential_operator(L);
    pde_.set_dirichlet_bc(DMatrix<double>::Zero(pde_.n_dofs(), 1));
    DMatrix<double> quadrature_nodes = pde_.quadrature_nodes();
    DMatrix<double> f(quadrature_nodes.rows(), 1);
    for (int64_t i = 0; i < quadrature_nodes.rows(); ++i) f(i) = ad.forcing(quadrature_nodes(i, 1));
    pde_.set_forcing(f);
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    EXPECT_TRUE(pde_.info().method_used == FDAPDE_SOLVER_BICGSTAB);
    EXPECT_TRUE(l2_error(pde_, [&](std::array<double, 3> x) { return ad.solution(x); }) < 1e-5);
}
// the same problem with the DIRECT solve asked for by name (what the reference's SparseLU does, fem_linear_elliptic_solver.h:38-47): no Krylov stage,
// info.method_used = FDAPDE_SOLVER_DENSE, the same solution
TEST(fem_pde_test, advection_diffusion_direct_solve_by_name) {
    AdvDiff ad;
    std::array<double, 2> beta_ {-ad.alpha_, 0.};
    auto L = -laplacian<FEM_HIP>() + advection<FEM_HIP>(beta_);
    FixtureMesh<2, 2> unit_square("unit_square");
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(unit_square.mesh);
    pde_.set_differential_operator(L);
    pde_.set_dirichlet_bc(DMatrix<double>::Zero(pde_.n_dofs(), 1));
    DMatrix<double> quadrature_nodes = pde_.quadrature_nodes();
    DMatrix<double> f(quadrature_nodes.rows(), 1);
    for (int64_t i = 0; i < quadrature_nodes.rows(); ++i) f(i) = ad.forcing(quadrature_nodes(i, 1));
    pde_.set_forcing(f);
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    const DMatrix<double> krylov = pde_.solution();
    pde_.solver_options().method = FDAPDE_SOLVER_DENSE;
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    EXPECT_TRUE(pde_.info().method_used == FDAPDE_SOLVER_DENSE);
    EXPECT_TRUE(pde_.info().iters == 0);
    double worst = 0, scale = 0;
    for (int64_t i = 0; i < krylov.rows(); ++i) {
        worst = std::max(worst, std::abs(krylov(i) - pde_.solution()(i)));
        scale = std::max(scale, std::abs(krylov(i)));
    }
    EXPECT_TRUE(worst <= 1e-8 * scale);
    EXPECT_TRUE(l2_error(pde_, [&](std::array<double, 3> x) { return ad.solution(x); }) < 1e-5);
}
// fem_pde_test.cpp:172-212
TEST(fem_pde_test, advection_diffusion_isotropic_order2) {
    AdvDiff ad;
    ScalarField<2> forcing([ad](const std::array<double, 2>& x) -> double { return ad.forcing(x[1]); });
    std::array<double, 2> beta_ {-ad.alpha_, 0.};
    auto L = -laplacian<FEM_HIP>() + advection<FEM_HIP>(beta_);
    FixtureMesh<2, 2> unit_square("unit_square");
    PDE<Triangulation<2, 2>, decltype(L), ScalarField<2>, FEM_HIP, fem_order<2>> pde_(unit_square.mesh, L, forcing);
    pde_.set_dirichlet_bc(DMatrix<double>::Zero(pde_.n_dofs(), 1));
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    EXPECT_TRUE(l2_error(pde_, [&](std::array<double, 3> x) { return ad.solution(x); }) < DOUBLE_TOLERANCE);
}
// the same case through FDAPDE_SOLVER_PMG by name (csrc/eng_pmg.hip: the P1 space of the mesh as the coarse level of a V(1,1) cycle inside a flexible GMRES -- what
// the open method takes for order-2 systems from 300 k DOFs on): the reference's gate (fem_pde_test.cpp:212) and the open method's solution
TEST(fem_pde_test, advection_diffusion_order2_two_level_solver_by_name) {
    AdvDiff ad;
    ScalarField<2> forcing([ad](const std::array<double, 2>& x) -> double { return ad.forcing(x[1]); });
    std::array<double, 2> beta_ {-ad.alpha_, 0.};
    auto L = -laplacian<FEM_HIP>() + advection<FEM_HIP>(beta_);
    FixtureMesh<2, 2> unit_square("unit_square");
    PDE<Triangulation<2, 2>, decltype(L), ScalarField<2>, FEM_HIP, fem_order<2>> pde_(unit_square.mesh, L, forcing);
    pde_.set_dirichlet_bc(DMatrix<double>::Zero(pde_.n_dofs(), 1));
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    const DMatrix<double> open_method = pde_.solution();
    pde_.solver_options().method = FDAPDE_SOLVER_PMG;
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    EXPECT_TRUE(pde_.info().method_used == FDAPDE_SOLVER_PMG);
    EXPECT_TRUE(pde_.info().iters > 0 && pde_.info().iters <= 30);
    double worst = 0, scale = 0;
    for (int64_t i = 0; i < open_method.rows(); ++i) {
        worst = std::max(worst, std::abs(open_method(i) - pde_.solution()(i)));
        scale = std::max(scale, std::abs(open_method(i)));
    }
    EXPECT_TRUE(worst <= 1e-8 * scale);
    EXPECT_TRUE(l2_error(pde_, [&](std::array<double, 3> x) { return ad.solution(x); }) < DOUBLE_TOLERANCE);
}

// ---- the mesh sharded over several devices BEHIND the same interface (include/fdapde_hip.h fdapde_ctx_create_multi): the reference's cases again
//      with a device list -- "devices" all GPU 0 here, its CUs shared out --, against the analytic solutions at the reference's gates and against
//      the one-device solve (<= 1e-9)
template <typename P1, typename P2> static double rel_diff(const P1& a, const P2& b) {
    double num = 0, den = 0;
    for (int64_t i = 0; i < a.solution().rows(); ++i) {
        const double d = a.solution()(i) - b.solution()(i);
        num += d * d, den += b.solution()(i) * b.solution()(i);
    }
    return std::sqrt(num / den);
}
TEST(sharded_test, laplacian_order1) {
    auto solution_expr = [](std::array<double, 3> x) -> double { return x[0] + x[1]; };
    FixtureMesh<2, 2> unit_square("unit_square");
    auto L = -laplacian<FEM_HIP>();
    using PDE_t = PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>>;
    PDE_t one(unit_square.mesh, L);
    one.set_dirichlet_bc(eval_at_dofs(one, solution_expr));
    one.set_forcing(DMatrix<double>::Zero(one.quadrature_nodes().rows(), 1));
    one.solver_options().rtol = 1e-12;
    one.init();
    one.solve();
    for (device_list devices : {device_list {0, 0}, device_list {0, 0, 0, 0}}) {
        PDE_t pde_(unit_square.mesh, L, devices);
        pde_.set_dirichlet_bc(eval_at_dofs(pde_, solution_expr));
        pde_.set_forcing(DMatrix<double>::Zero(pde_.quadrature_nodes().rows(), 1));
        pde_.solver_options().rtol = 1e-12;
        pde_.init();
        pde_.solve();
        EXPECT_TRUE(pde_.success());
        EXPECT_TRUE(pde_.n_dofs() == 3600);
        EXPECT_TRUE(l2_error(pde_, solution_expr) < DOUBLE_TOLERANCE);
        EXPECT_TRUE(rel_diff(pde_, one) < 1e-9);
        int32_t n_dev = 0;
        fdapde_ctx_devices(pde_.context(), &n_dev, nullptr, nullptr, nullptr, nullptr);
        EXPECT_TRUE(n_dev == (int32_t)devices.ids.size());
        double worst = 0;   // stiff() in the whole mesh's numbering: the one-device matrix to the last places
        for (size_t k = 0; k < one.stiff().values.size(); ++k) worst = std::fmax(worst, std::fabs(one.stiff().values[k] - pde_.stiff().values[k]));
        EXPECT_TRUE(worst < 1e-13);
    }
}
TEST(sharded_test, laplacian_order2_callable_force) {
    auto solution_expr = [](std::array<double, 3> x) -> double { return 1. - x[0] * x[0] - x[1] * x[1]; };
    ScalarField<2> forcing([](const std::array<double, 2>&) -> double { return 4.0; });
    FixtureMesh<2, 2> unit_square("unit_square");
    auto L = -laplacian<FEM_HIP>();
    PDE<Triangulation<2, 2>, decltype(L), ScalarField<2>, FEM_HIP, fem_order<2>> pde_(unit_square.mesh, L, forcing, device_list {0, 0});
    pde_.set_dirichlet_bc(eval_at_dofs(pde_, solution_expr));
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    EXPECT_TRUE(pde_.n_dofs() == 14161);
    EXPECT_TRUE(l2_error(pde_, solution_expr) < DOUBLE_TOLERANCE);
}
TEST(sharded_test, advection_diffusion) {
    AdvDiff ad;
    std::array<double, 2> beta_ {-ad.alpha_, 0.};
    auto L = -laplacian<FEM_HIP>() + advection<FEM_HIP>(beta_);
    FixtureMesh<2, 2> unit_square("unit_square");
    for (int order : {1, 2}) {
        auto run = [&](auto& pde_) {
            pde_.set_dirichlet_bc(DMatrix<double>::Zero(pde_.n_dofs(), 1));
            DMatrix<double> quadrature_nodes = pde_.quadrature_nodes();
            DMatrix<double> f(quadrature_nodes.rows(), 1);
            for (int64_t i = 0; i < quadrature_nodes.rows(); ++i) f(i) = ad.forcing(quadrature_nodes(i, 1));
            pde_.set_forcing(f);
            pde_.init();
            pde_.solve();
            EXPECT_TRUE(pde_.success());
            EXPECT_TRUE(pde_.info().method_used == FDAPDE_SOLVER_BICGSTAB);
            EXPECT_TRUE(l2_error(pde_, [&](std::array<double, 3> x) { return ad.solution(x); }) < (order == 1 ? 1e-5 : DOUBLE_TOLERANCE));
        };
        if (order == 1) {
            PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(unit_square.mesh, L, std::vector<int> {0, 0, 0});
            run(pde_);
        } else {
            PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<2>> pde_(unit_square.mesh, L, std::vector<int> {0, 0});
            run(pde_);
        }
    }
}
// the type-erased handle over a sharded PDE: make_pde copies the PDE, the copies share the multi-device context until one of them computes something
// different -- then it leaves with a clone on the same devices (fdapde_ctx_clone of a multi-device context)
TEST(sharded_test, make_pde_copy_and_diverge) {
    auto solution_expr = [](std::array<double, 3> x) -> double { return x[0] + x[1]; };
    FixtureMesh<2, 2> mesh("unit_square_32");
    auto L = -laplacian<FEM_HIP>();
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> typed(mesh.mesh, L, device_list {0, 0});
    typed.set_dirichlet_bc(eval_at_dofs(typed, solution_expr));
    typed.set_forcing(DMatrix<double>::Zero(typed.quadrature_nodes().rows(), 1));
    typed.init();
    typed.solve();
    EXPECT_TRUE(typed.success());
    auto copy = typed;   // shares the context
    copy.set_dirichlet_bc(DMatrix<double>::Zero(copy.n_dofs(), 1));
    copy.solve();        // ... and leaves with a clone here (no second init: the clone carries stiff_ / force_)
    EXPECT_TRUE(copy.success());
    double worst_copy = 0, worst_typed = 0;
    for (int64_t i = 0; i < typed.solution().rows(); ++i) {
        worst_copy = std::fmax(worst_copy, std::fabs(copy.solution()(i)));
        worst_typed = std::fmax(worst_typed, std::fabs(typed.solution()(i) - solution_expr(typed.dof_coords().row3(i))));
    }
    EXPECT_TRUE(worst_copy < 1e-12);    // zero data, zero forcing
    EXPECT_TRUE(worst_typed < 1e-6);    // the original kept its answer
    int32_t n_dev = 0;
    fdapde_ctx_devices(copy.context(), &n_dev, nullptr, nullptr, nullptr, nullptr);
    EXPECT_TRUE(n_dev == 2 && copy.context() != typed.context());
}

// fem_operators_test.cpp:41-100: golden P2 stiffness of c_shaped cell 175.  The facade exposes the assembled matrix, not
// element matrices; entries of pairs of DOFs that only cell 175 contains equal the local integrals (the edge-midpoint
// pairs on an edge-shared pair are sums over two cells), so compare those through stiff().
TEST(fem_operators_test, laplacian_order_2_through_stiff) {
    FixtureMesh<2, 2> CShaped("c_shaped");
    auto L = -laplacian<FEM_HIP>();
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<2>> pde_(CShaped.mesh, L);
    pde_.init();
    EXPECT_TRUE(pde_.n_dofs() == 945);   // columns of lagrangian_pointwise_eval_order2.mtx
    const double expected[6][6] = {
      { 0.7043890316492852,  0.1653830261033185,  0.0694133177797771, -0.6615321044132733, -0.2776532711191089,  0.0000000000000013},
      { 0.1653830261033185,  0.7043890316492852,  0.0694133177797769, -0.6615321044132735,  0.0000000000000003, -0.2776532711191076},
      { 0.0694133177797771,  0.0694133177797769,  0.4164799066786617,  0.0000000000000002, -0.2776532711191083, -0.2776532711191075},
      {-0.6615321044132733, -0.6615321044132735,  0.0000000000000002,  2.4336772933029756, -0.5553065422382126, -0.5553065422382162},
      {-0.2776532711191089,  0.0000000000000003, -0.2776532711191083, -0.5553065422382126,  2.4336772933029738, -1.3230642088265447},
      { 0.0000000000000013, -0.2776532711191075, -0.2776532711191076, -0.5553065422382162, -1.3230642088265447,  2.4336772933029751}};
    // pairs (vertex i, midpoint of the edge opposite to i) and (midpoint, midpoint) couple only inside cell 175
    const int pairs[6][2] = {{0, 5}, {1, 4}, {2, 3}, {3, 4}, {3, 5}, {4, 5}};
    for (auto& pr : pairs) {
        const int di = pde_.dofs()(175, pr[0]), dj = pde_.dofs()(175, pr[1]);
        EXPECT_TRUE(almost_equal(pde_.stiff().coeff(di, dj), expected[pr[0]][pr[1]]));
        EXPECT_TRUE(almost_equal(pde_.stiff().coeff(dj, di), expected[pr[1]][pr[0]]));
    }
}
// error behaviour: fem_solver_base.h:146 / fem_linear_elliptic_solver.h:36 throw; non-convergence -> success = false
TEST(fem_pde_test, error_behaviour) {
    FixtureMesh<2, 2> m("unit_square_16");
    auto L = -laplacian<FEM_HIP>();
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(m.mesh, L);
    bool thrown = false;
    try { pde_.solve(); } catch (const std::runtime_error& e) { thrown = std::string(e.what()) == "solver must be initialized first!"; }
    EXPECT_TRUE(thrown);
    pde_.set_forcing(DMatrix<double>(3 * m.mesh.n_cells(), 1, 1.0));
    pde_.set_dirichlet_bc(DMatrix<double>::Zero(pde_.n_dofs(), 1));
    pde_.init();
    pde_.solver_options().maxit = 2;
    pde_.solve();
    EXPECT_TRUE(!pde_.success());
    pde_.solver_options().maxit = 0;
    pde_.solve();
    EXPECT_TRUE(pde_.success());
}
// 3-D: unit_sphere (the reference's only 3-D mesh; 1395 negatively oriented tetrahedra), P1 reproduces x + y + z
TEST(fem_pde_test, laplacian_3d_order1) {
    auto solution_expr = [](std::array<double, 3> x) -> double { return x[0] + x[1] + x[2]; };
    FixtureMesh<3, 3> sphere("unit_sphere");
    auto L = -laplacian<FEM_HIP>();
    PDE<Triangulation<3, 3>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(sphere.mesh, L);
    pde_.set_dirichlet_bc(eval_at_dofs(pde_, solution_expr));
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    DMatrix<double> ex = eval_at_dofs(pde_, solution_expr);
    double worst = 0;
    for (int64_t i = 0; i < ex.rows(); ++i) worst = std::fmax(worst, std::fabs(ex(i) - pde_.solution()(i)));
    EXPECT_TRUE(worst < 1e-8);
}

// integration_test.cpp:73-80: a field integrated over the whole triangulation (Integrator<FEM, M, R>::integrate, integrator.h:61-69)
TEST(integration_test, integrate_over_triangulation) {
    FixtureMesh<2, 2> unit_square("unit_square");
    fdapde::amd::Integrator<FEM_HIP, 2, 1> integrator {};
    EXPECT_TRUE(almost_equal(1.0, integrator.integrate(unit_square.mesh, [](std::array<double, 2>) -> double { return 1; })));
    // a linear field is integrated exactly by both rules: int_[0,1]^2 (x + y) = 1
    EXPECT_TRUE(almost_equal(1.0, integrator.integrate(unit_square.mesh, [](std::array<double, 2> x) -> double { return x[0] + x[1]; }), 1e-12));
    fdapde::amd::Integrator<FEM_HIP, 2, 2> integrator2 {};
    EXPECT_TRUE(almost_equal(2.0 / 3.0, integrator2.integrate(unit_square.mesh, [](std::array<double, 2> x) -> double { return x[0] * x[0] + x[1] * x[1]; }), 1e-12));
    // 3-D: the volume of the fixture's polyhedral ball = the sum of its cells' measures (1395 of them negatively oriented)
    FixtureMesh<3, 3> sphere("unit_sphere");
    const auto& nd = sphere.mesh.nodes();
    const auto& cl = sphere.mesh.cells();
    double volume = 0;
    for (int64_t c = 0; c < cl.rows(); ++c) {
        double e[3][3];
        for (int k = 0; k < 3; ++k)
            for (int d = 0; d < 3; ++d) e[k][d] = nd(cl(c, k + 1), d) - nd(cl(c, 0), d);
        const double det = e[0][0] * (e[1][1] * e[2][2] - e[1][2] * e[2][1]) - e[0][1] * (e[1][0] * e[2][2] - e[1][2] * e[2][0]) +
                           e[0][2] * (e[1][0] * e[2][1] - e[1][1] * e[2][0]);
        volume += std::fabs(det) / 6.0;
    }
    fdapde::amd::Integrator<FEM_HIP, 3, 1> integrator3 {};
    EXPECT_TRUE(almost_equal(volume, integrator3.integrate(sphere.mesh, [](std::array<double, 3>) -> double { return 1; }), 1e-12));
}

// fem_pde_test.cpp:222-285: parabolic, P2, 101 time points
TEST(fem_pde_test, parabolic_isotropic_order2) {
    constexpr double pi = 3.14159265358979323846;
    const int M = 101;
    DMatrix<double> times(M, 1);
    for (int j = 0; j < M; ++j) times(j) = 1.0 / (M - 1) * j;
    auto solution_expr = [](std::array<double, 3> x, double t) { return std::sin(2 * pi * x[0]) * std::sin(2 * pi * x[1]) * std::exp(-t); };
    auto forcing_expr = [](double x0, double x1, double t) {
        return (8 * pi * pi - 1.) * std::sin(2 * pi * x0) * std::sin(2 * pi * x1) * std::exp(-t);
    };
    FixtureMesh<2, 2> unit_square("unit_square");
    auto L = dt<FEM_HIP>() - laplacian<FEM_HIP>();
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<2>> pde_(unit_square.mesh, times);
    pde_.set_differential_operator(L);
    DMatrix<double> nodes_ = pde_.dof_coords();
    DMatrix<double> dirichlet_bc(nodes_.rows(), M), solution_ex(nodes_.rows(), M), initial_condition(nodes_.rows(), 1);
    for (int64_t i = 0; i < nodes_.rows(); ++i)
        for (int j = 0; j < M; ++j) dirichlet_bc(i, j) = solution_ex(i, j) = solution_expr(nodes_.row3(i), times(j));
    for (int64_t i = 0; i < nodes_.rows(); ++i) initial_condition(i) = solution_expr(nodes_.row3(i), times(0));
    pde_.set_dirichlet_bc(dirichlet_bc);
    pde_.set_initial_condition(initial_condition);
    DMatrix<double> quadrature_nodes = pde_.quadrature_nodes();
    DMatrix<double> f(quadrature_nodes.rows(), M);
    for (int64_t i = 0; i < quadrature_nodes.rows(); ++i)
        for (int j = 0; j < M; ++j) f(i, j) = forcing_expr(quadrature_nodes(i, 0), quadrature_nodes(i, 1), times(j));
    pde_.set_forcing(f);
    pde_.init();
    pde_.solve();
    EXPECT_TRUE(pde_.success());
    double worst = 0;
    for (int j = 0; j < M; ++j) {
        DMatrix<double> e2(nodes_.rows(), 1);
        for (int64_t i = 0; i < nodes_.rows(); ++i) {
            const double e = solution_ex(i, j) - pde_.solution()(i, j);
            e2(i) = e * e;
        }
        DMatrix<double> Me = pde_.mass() * e2;
        double s = 0;
        for (int64_t i = 0; i < Me.rows(); ++i) s += Me(i);
        worst = std::fmax(worst, s);
    }
    EXPECT_TRUE(worst < 1e-7);
}
// fem_pde_test.cpp:295-368: convergence order of the parabolic solve, P1, fixed time step, on the structured fixtures.  The reference
// refines 16 / 32 / 64 / 128; unit_square_128 (3 MB of CSV) is not among the committed fixtures, so the order is checked on the
// first two halvings -- the same criterion, floor(log2(e_h / e_{h/2})) == 2.
TEST(fem_pde_test, parabolic_isotropic_order1_convergence) {
    constexpr double pi = 3.14159265358979323846;
    const int M = 31;
    DMatrix<double> times(M, 1);
    const double time_max = 1.;
    for (int j = 0; j < M; ++j) times(j) = time_max / (M - 1) * j;
    const int num_refinements = 3;
    const int N[num_refinements] = {16, 32, 64};
    DMatrix<double> error_L2(M, num_refinements, 0.0);
    auto solution_expr = [](std::array<double, 3> x, double t) -> double { return std::sin(2 * pi * x[0]) * std::sin(2 * pi * x[1]) * std::exp(-t); };
    auto forcing_expr = [](double x0, double x1, double t) -> double {
        return (8 * pi * pi - 1.) * std::sin(2 * pi * x0) * std::sin(2 * pi * x1) * std::exp(-t);
    };
    for (int n = 0; n < num_refinements; ++n) {
        FixtureMesh<2, 2> unit_square("unit_square_" + std::to_string(N[n]));
        auto L = dt<FEM_HIP>() - laplacian<FEM_HIP>();
        PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(unit_square.mesh, times);
        pde_.set_differential_operator(L);
        DMatrix<double> nodes_ = pde_.dof_coords();
        DMatrix<double> dirichlet_bc(nodes_.rows(), M), solution_ex(nodes_.rows(), M), initial_condition(nodes_.rows(), 1);
        for (int64_t i = 0; i < nodes_.rows(); ++i)
            for (int j = 0; j < M; ++j) dirichlet_bc(i, j) = solution_ex(i, j) = solution_expr(nodes_.row3(i), times(j));
        for (int64_t i = 0; i < nodes_.rows(); ++i) initial_condition(i) = solution_expr(nodes_.row3(i), times(0));
        pde_.set_dirichlet_bc(dirichlet_bc);
        pde_.set_initial_condition(initial_condition);
        DMatrix<double> quadrature_nodes = pde_.quadrature_nodes();
        DMatrix<double> f(quadrature_nodes.rows(), M);
        for (int64_t i = 0; i < quadrature_nodes.rows(); ++i)
            for (int j = 0; j < M; ++j) f(i, j) = forcing_expr(quadrature_nodes(i, 0), quadrature_nodes(i, 1), times(j));
        pde_.set_forcing(f);
        pde_.init();
        pde_.solve();
        EXPECT_TRUE(pde_.success());
        for (int j = 0; j < M; ++j) {
            DMatrix<double> e2(nodes_.rows(), 1);
            for (int64_t i = 0; i < nodes_.rows(); ++i) {
                const double e = solution_ex(i, j) - pde_.solution()(i, j);
                e2(i) = e * e;
            }
            DMatrix<double> Me = pde_.mass() * e2;
            double s = 0;
            for (int64_t i = 0; i < Me.rows(); ++i) s += Me(i);
            error_L2(j, n) = std::sqrt(s);
        }
    }
    for (int n = 1; n < num_refinements; ++n) {
        const double order = std::log2(error_L2(M - 1, n - 1) / error_L2(M - 1, n));
        EXPECT_TRUE(std::floor(order) == 2);
    }
}

// fdapde::SparseLU usage pattern (utils/symbols.h:133-160, linear_algebra/smw.h:46-48): factor once, solve many columns
TEST(sparse_solver_test, factor_once_solve_many) {
    FixtureMesh<2, 2> m("unit_square_32");
    auto L = -laplacian<FEM_HIP>() + reaction<FEM_HIP>(2.0);
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(m.mesh, L);
    pde_.init();
    auto invA = pde_.make_solver();
    invA.compute(pde_.stiff(), /*symmetric=*/true);
    EXPECT_TRUE(bool(invA));
    DMatrix<double> X(pde_.n_dofs(), 3);
    for (int64_t i = 0; i < X.rows(); ++i)
        for (int j = 0; j < 3; ++j) X(i, j) = std::sin(0.01 * i * (j + 1)) + j;
    DMatrix<double> B = pde_.stiff() * X;
    DMatrix<double> Y = invA.solve(B);
    double worst = 0;
    for (int64_t i = 0; i < X.rows(); ++i)
        for (int j = 0; j < 3; ++j) worst = std::fmax(worst, std::fabs(X(i, j) - Y(i, j)));
    EXPECT_TRUE(worst < 1e-7);
}

// Triangulation beyond the cell list (geometry/triangulation.h:143-196, 319-399), device-built, against the files the reference's
// MeshLoader reads next to the mesh (test/src/utils/mesh_loader.h:62-84): neigh.csv, edges.csv
template <int M, int N> static void check_topology(const std::string& id) {
    FixtureMesh<M, N> m(id);
    const DMatrix<int>& nb = m.mesh.neighbors();
    EXPECT_TRUE(nb.rows() == m.neighbors_.rows() && nb.cols() == m.neighbors_.cols());
    int64_t bad = 0;
    for (int64_t i = 0; i < nb.rows(); ++i)
        for (int64_t j = 0; j < nb.cols(); ++j) bad += nb(i, j) != m.neighbors_(i, j);
    EXPECT_TRUE(bad == 0);
    EXPECT_TRUE(m.mesh.n_facets() == m.edges_.rows());   // the file lists the same facets (in another order)
    int64_t n_bnd = 0, wrong = 0;
    for (int64_t f = 0; f < m.mesh.n_facets(); ++f) {
        const bool b = m.mesh.is_facet_on_boundary(f);
        n_bnd += b;
        wrong += b != (m.mesh.facet_to_cells()(f, 1) < 0);
    }
    EXPECT_TRUE(wrong == 0 && n_bnd > 0);
}
TEST(mesh_test, neighbours_and_facets_match_the_fixture_files) {
    check_topology<2, 2>("unit_square");
    check_topology<2, 2>("c_shaped");
    check_topology<3, 3>("unit_sphere");
}

// SMW (linear_algebra/smw.h:38-59) on the factor-once handle, and row-sum lumping (linear_algebra/lumping.h:30-41)
TEST(linear_algebra_test, smw_and_lumping) {
    FixtureMesh<2, 2> m("unit_square_32");
    auto L = -laplacian<FEM_HIP>() + reaction<FEM_HIP>(2.0);
    PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>> pde_(m.mesh, L);
    pde_.init();
    const int64_t n = pde_.n_dofs(), q = 3;
    auto invA = pde_.make_solver();
    invA.compute(pde_.stiff(), /*symmetric=*/true);
    DMatrix<double> U(n, q), invC(q, q, 0.0), b(n, 1);
    for (int64_t i = 0; i < n; ++i) {
        b(i) = std::cos(0.02 * i) + 0.3;
        for (int64_t j = 0; j < q; ++j) U(i, j) = 0.05 * std::sin(0.013 * i * (j + 1)) + 0.01 * j;
    }
    for (int64_t j = 0; j < q; ++j) invC(j, j) = 1.0 + 0.5 * j;   // C = diag(1, 2/3, 1/2)
    invC(0, 1) = 0.1;
    DMatrix<double> V = transpose(U);
    SMW<decltype(invA)> smw;
    DMatrix<double> x = smw.solve(invA, U, invC, V, b);
    // residual of (A + U C V) x = b with C = invC^{-1}
    PartialPivLU luC;
    luC.compute(invC);
    DMatrix<double> r = pde_.stiff() * x + U * luC.solve(V * x) - b;
    double rn = 0, bn = 0;
    for (int64_t i = 0; i < n; ++i) rn += r(i) * r(i), bn += b(i) * b(i);
    EXPECT_TRUE(std::sqrt(rn / bn) < 1e-8);
    // lump(mass): row sums = integrals of the basis functions; they add up to the area of the unit square
    SpMatrix<double> R0 = lump(pde_.mass());
    double area = 0;
    for (int64_t i = 0; i < n; ++i) {
        EXPECT_TRUE(R0.rowptr[(size_t)i + 1] - R0.rowptr[(size_t)i] == 1 && R0.colidx[(size_t)i] == i);
        area += R0.values[(size_t)i];
    }
    EXPECT_TRUE(almost_equal(area, 1.0, 1e-12));
    std::vector<double> dev((size_t)n);
    EXPECT_TRUE(fdapde_lump(pde_.context(), FDAPDE_MAT_MASS, dev.data()) == FDAPDE_OK);   // the same on the device
    double worst = 0;
    for (int64_t i = 0; i < n; ++i) worst = std::fmax(worst, std::fabs(dev[(size_t)i] - R0.values[(size_t)i]));
    EXPECT_TRUE(worst < 1e-15);
}

// PDE__::eval_basis (pde/pde.h:149-158) against the reference's golden matrices (test/src/lagrangian_basis_test.cpp:190-260):
// pointwise evaluation at c_shaped/locs.csv, areal evaluation with quasi_circle/incidence_matrix.csv, orders 1 and 2
static SpMatrix<double> read_mtx(const std::string& file) {
    std::ifstream in(file);
    if (!in) throw std::runtime_error("cannot open " + file);
    std::string line;
    do { std::getline(in, line); } while (!line.empty() && line[0] == '%');
    std::istringstream hdr(line);
    int64_t nr, nc, nz;
    hdr >> nr >> nc >> nz;
    std::vector<std::vector<std::pair<int32_t, double>>> rows((size_t)nr);
    for (int64_t k = 0; k < nz; ++k) {
        int64_t i, j;
        double v;
        in >> i >> j >> v;
        rows[(size_t)i - 1].push_back({(int32_t)(j - 1), v});
    }
    SpMatrix<double> m;
    m.n_rows = nr, m.n_cols = nc, m.rowptr.assign((size_t)nr + 1, 0);
    for (int64_t i = 0; i < nr; ++i) {
        std::sort(rows[(size_t)i].begin(), rows[(size_t)i].end());
        for (auto& cv : rows[(size_t)i]) m.colidx.push_back(cv.first), m.values.push_back(cv.second);
        m.rowptr[(size_t)i + 1] = (int32_t)m.colidx.size();
    }
    return m;
}
static double max_abs_diff(const SpMatrix<double>& a, const SpMatrix<double>& b) {   // as dense matrices (explicit zeros may differ)
    if (a.rows() != b.rows() || a.cols() != b.cols()) return 1e300;
    double worst = 0;
    for (int64_t i = 0; i < a.rows(); ++i) {
        for (int32_t k = a.rowptr[(size_t)i]; k < a.rowptr[(size_t)i + 1]; ++k)
            worst = std::fmax(worst, std::fabs(a.values[(size_t)k] - b.coeff(i, a.colidx[(size_t)k])));
        for (int32_t k = b.rowptr[(size_t)i]; k < b.rowptr[(size_t)i + 1]; ++k)
            worst = std::fmax(worst, std::fabs(b.values[(size_t)k] - a.coeff(i, b.colidx[(size_t)k])));
    }
    return worst;
}
template <int R> static void eval_basis_case() {
    const std::string mtx = MESH_PATH + "/../mtx/";
    {
        FixtureMesh<2, 2> m("c_shaped");
        auto L = -laplacian<FEM_HIP>();
        PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<R>> pde_(m.mesh, L);
        DMatrix<double> locs = read_csv<double>(MESH_PATH + "/c_shaped/locs.csv");
        auto res = pde_.eval_basis(0, locs);
        EXPECT_TRUE(res.has_value());
        SpMatrix<double> gold = read_mtx(mtx + "lagrangian_pointwise_eval_order" + std::to_string(R) + ".mtx");
        EXPECT_TRUE(max_abs_diff(res->Psi, gold) < 1e-12);
        EXPECT_TRUE(res->D.rows() == locs.rows() && res->D(0) == 1.0);
        EXPECT_TRUE(!pde_.eval_basis(7, locs).has_value());
    }
    {
        FixtureMesh<2, 2> m("quasi_circle");
        auto L = -laplacian<FEM_HIP>();
        PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<R>> pde_(m.mesh, L);
        DMatrix<int> inc_i = read_csv<int>(MESH_PATH + "/quasi_circle/incidence_matrix.csv");
        DMatrix<double> inc(inc_i.rows(), inc_i.cols());
        for (int64_t i = 0; i < inc.rows(); ++i)
            for (int64_t j = 0; j < inc.cols(); ++j) inc(i, j) = inc_i(i, j);
        auto res = pde_.eval_basis(1, inc);
        EXPECT_TRUE(res.has_value());
        SpMatrix<double> gold = read_mtx(mtx + "lagrangian_areal_eval_order" + std::to_string(R) + ".mtx");
        EXPECT_TRUE(max_abs_diff(res->Psi, gold) < 1e-12);
        double area = 0;
        for (int64_t k = 0; k < res->D.rows(); ++k) area += res->D(k);
        EXPECT_TRUE(area > 0);
    }
}
TEST(lagrangian_basis_test, eval_basis_golden) {
    eval_basis_case<1>();
    eval_basis_case<2>();
}

// the 3-column sparse dialect (csv_reader.h:119-166): the reference keeps no sparse fixture, so the dense incidence matrix of
// quasi_circle is written out entry by entry -- ids 1-based, one entry split in two (occurrences are summed), one NA -- and read back
TEST(csv_reader_test, sparse_three_column_format) {
    const std::string dense_file = MESH_PATH + "/quasi_circle/incidence_matrix.csv";
    DMatrix<double> dense = read_csv<double>(dense_file);
    EXPECT_TRUE(dense.rows() > 0 && dense.cols() == 630);
    const std::string tmp = "/tmp/fdapde_sparse_" + std::to_string((long)getpid()) + ".csv";
    {
        std::ofstream out(tmp);
        out << "\"\",\"i\",\"j\",\"x\"\n";
        long id = 0;
        bool split_done = false;
        for (int64_t j = dense.cols() - 1; j >= 0; --j)        // (not in row order: the reader must sort)
            for (int64_t i = 0; i < dense.rows(); ++i) {
                if (dense(i, j) == 0.0) continue;
                if (!split_done) {
                    out << "\"" << ++id << "\", " << i + 1 << ", " << j + 1 << ", 0.25\n";
                    out << "\"" << ++id << "\"," << i + 1 << "," << j + 1 << "," << dense(i, j) - 0.25 << "\n";
                    split_done = true;
                } else
                    out << "\"" << ++id << "\"," << i + 1 << "," << j + 1 << "," << dense(i, j) << "\n";
            }
    }
    SpMatrix<double> sp = fdapde::amd::CSVReader<double>().parse_sparse_file(tmp);
    // the matrix is (largest row id) x (largest column id) (csv_reader.h:141-144,160): trailing empty rows / columns are not represented
    int64_t last_row = 0, last_col = 0, nnz = 0;
    for (int64_t i = 0; i < dense.rows(); ++i)
        for (int64_t j = 0; j < dense.cols(); ++j)
            if (dense(i, j) != 0.0) last_row = std::max(last_row, i + 1), last_col = std::max(last_col, j + 1), ++nnz;
    EXPECT_TRUE(sp.rows() == last_row && sp.cols() == last_col && sp.nonZeros() == nnz);
    bool same = true, sorted = true;
    for (int64_t i = 0; i < sp.rows(); ++i) {
        for (int32_t k = sp.rowptr[(size_t)i]; k + 1 < sp.rowptr[(size_t)i + 1]; ++k) sorted &= sp.colidx[(size_t)k] < sp.colidx[(size_t)k + 1];
        for (int64_t j = 0; j < sp.cols(); ++j) same &= almost_equal(sp.coeff(i, j), dense(i, j), 1e-15);
    }
    EXPECT_TRUE(same && sorted);
    {   // a dense file is refused with the reference's message; NA reads as NaN
        bool threw = false;
        try { fdapde::amd::CSVReader<double>().parse_sparse_file(dense_file); } catch (const std::runtime_error& e) { threw = std::string(e.what()).find("sparse 3-column") != std::string::npos; }
        EXPECT_TRUE(threw);
        std::ofstream out(tmp);
        out << "\"\",\"i\",\"j\",\"x\"\n\"1\",2,3,NA\n";
        out.close();
        SpMatrix<double> na = fdapde::amd::CSVReader<double>().parse_sparse_file(tmp);
        EXPECT_TRUE(na.rows() == 2 && na.cols() == 3 && na.nonZeros() == 1 && std::isnan(na.coeff(1, 2)));
    }
    std::remove(tmp.c_str());
}

// ---- the type-erased face: make_pde / erase<heap_storage, PDE__> (pde/pde.h:117-169, utils/type_erasure.h:124-160) -----------------
// What downstream models do with a PDE: build it through make_pde, keep the handle by value, copy it around.  Every copy is a deep
// copy of the PDE in the reference; here copies share the device context until one of them computes (include/fdapde_hip.hpp).
using ErasedPDE = erase<heap_storage, PDE__>;
static double handle_l2_error(const ErasedPDE& pde, double (*u)(std::array<double, 3>)) {
    DMatrix<double> nodes = pde.dof_coords(), e2(nodes.rows(), 1);
    for (int64_t i = 0; i < nodes.rows(); ++i) {
        const double err = u(nodes.row3(i)) - pde.solution()(i);
        e2(i) = err * err;
    }
    DMatrix<double> Me = pde.mass() * e2;
    double s = 0;
    for (int64_t i = 0; i < Me.rows(); ++i) s += Me(i);
    return s;
}
static double u_linear(std::array<double, 3> x) { return x[0] + x[1]; }
static double u_bowl(std::array<double, 3> x) { return 1. - x[0] * x[0] - x[1] * x[1]; }
TEST(type_erasure_test, make_pde_copy_the_handle_solve_both) {
    FixtureMesh<2, 2> unit_square("unit_square");
    auto L = -laplacian<FEM_HIP>();
    using PDE_ = PDE<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<2>>;
    ErasedPDE a = make_pde<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<2>>(unit_square.mesh, L);
    EXPECT_TRUE(bool(a) && !bool(ErasedPDE()));
    EXPECT_TRUE(a.n_dofs() == 14161);
    DMatrix<double> nodes = a.dof_coords(), quadrature_nodes = a.quadrature_nodes();
    DMatrix<double> g_linear(nodes.rows(), 1), g_bowl(nodes.rows(), 1);
    for (int64_t i = 0; i < nodes.rows(); ++i) g_linear(i) = u_linear(nodes.row3(i)), g_bowl(i) = u_bowl(nodes.row3(i));
    // problem A: -lap u = 0, u = x + y on the boundary (fem_pde_test.cpp:43-75)
    a.set_dirichlet_bc(g_linear);
    a.set_forcing(DMatrix<double>::Zero(quadrature_nodes.rows(), 1));
    // the copy gets problem B: -lap u = 4, u = 1 - x^2 - y^2 (fem_pde_test.cpp:78-107).  Neither has touched the device yet.
    ErasedPDE b = a;
    b.set_forcing(DMatrix<double>(quadrature_nodes.rows(), 1, 4.0));
    b.set_dirichlet_bc(g_bowl);
    EXPECT_TRUE(a.forcing_data()(0) == 0.0 && b.forcing_data()(0) == 4.0);   // two PDEs, not two views of one
    a.init(), b.init();   // (the second init leaves the shared context with a clone of its own)
    a.solve(), b.solve();
    EXPECT_TRUE(a.success() && b.success());
    EXPECT_TRUE(handle_l2_error(a, u_linear) < DOUBLE_TOLERANCE);
    EXPECT_TRUE(handle_l2_error(b, u_bowl) < DOUBLE_TOLERANCE);
    // a copy taken AFTER init carries stiff_ / mass_ / force_ with it: new boundary data and solve(), no second init -- and the
    // original is untouched by what the copy does (reference: the solver's matrices are part of the deep copy)
    ErasedPDE c = b;
    DMatrix<double> g_shift = g_bowl;
    for (int64_t i = 0; i < g_shift.rows(); ++i) g_shift(i) += 1.0;   // u + 1 solves the same equation with the shifted data
    c.set_dirichlet_bc(g_shift);
    c.solve();
    EXPECT_TRUE(c.success());
    double worst_c = 0, worst_b = 0;
    for (int64_t i = 0; i < nodes.rows(); ++i) worst_c = std::fmax(worst_c, std::fabs(c.solution()(i) - b.solution()(i) - 1.0));
    EXPECT_TRUE(worst_c < 1e-7);
    DMatrix<double> b_before = b.solution();
    b.solve();   // the context b computes on still holds b's problem
    for (int64_t i = 0; i < nodes.rows(); ++i) worst_b = std::fmax(worst_b, std::fabs(b.solution()(i) - b_before(i)));
    EXPECT_TRUE(worst_b == 0.0);
    EXPECT_TRUE(handle_l2_error(b, u_bowl) < DOUBLE_TOLERANCE);
    // moving transfers ownership (type_erasure.h:148-160); assignment from a handle deep-copies (136-146)
    ErasedPDE d = std::move(c);
    EXPECT_TRUE(bool(d) && !bool(c));
    c = a;
    EXPECT_TRUE(bool(c) && c.n_dofs() == a.n_dofs() && handle_l2_error(c, u_linear) < DOUBLE_TOLERANCE);
    // typed setters of the erased interface (pde.h:160-163): the PDE's own types pass, others are refused
    bool refused = false;
    try { d.set_forcing(ScalarField<2>([](const std::array<double, 2>&) { return 1.0; })); } catch (const std::runtime_error&) { refused = true; }
    EXPECT_TRUE(refused);
    d.set_differential_operator(-laplacian<FEM_HIP>() + reaction<FEM_HIP>(1.0));
    d.init();
    EXPECT_TRUE(d.stiff().nonZeros() == a.stiff().nonZeros() && d.stiff().values != a.stiff().values);
    // the PDE object itself is copyable the same way (PDE(const PDE&)): what heap_storage's `new T(obj)` needs
    PDE_ direct(unit_square.mesh, L);
    PDE_ twin(direct);
    twin.set_forcing(DMatrix<double>::Zero(quadrature_nodes.rows(), 1));
    twin.set_dirichlet_bc(g_linear);
    twin.init();
    twin.solve();
    EXPECT_TRUE(twin.success() && !direct.is_init());
    EXPECT_TRUE(l2_error(twin, u_linear) < DOUBLE_TOLERANCE);
}
// the other slots through the handle: eval_basis (12-13) against the golden matrix, space-time slots (10, 11, 16) on a parabolic PDE
TEST(type_erasure_test, eval_basis_and_space_time_slots) {
    {
        FixtureMesh<2, 2> m("c_shaped");
        auto L = -laplacian<FEM_HIP>();
        ErasedPDE pde = make_pde<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>>(m.mesh, L);
        ErasedPDE copy = pde;   // evaluation reads the context only: both handles use the one they share
        DMatrix<double> locs = read_csv<double>(MESH_PATH + "/c_shaped/locs.csv");
        auto res = copy.eval_basis(0, locs);
        EXPECT_TRUE(res.has_value());
        SpMatrix<double> gold = read_mtx(MESH_PATH + "/../mtx/lagrangian_pointwise_eval_order1.mtx");
        EXPECT_TRUE(max_abs_diff(res->Psi, gold) < 1e-12);
        EXPECT_TRUE(!pde.eval_basis(2, locs).has_value());
    }
    {
        constexpr double pi = 3.14159265358979323846;
        const int M = 11;
        DMatrix<double> times(M, 1);
        for (int j = 0; j < M; ++j) times(j) = 0.1 * j;
        FixtureMesh<2, 2> m("unit_square_16");
        auto L = dt<FEM_HIP>() - laplacian<FEM_HIP>();
        ErasedPDE pde = make_pde<Triangulation<2, 2>, decltype(L), DMatrix<double>, FEM_HIP, fem_order<1>>(m.mesh, times, L);
        EXPECT_TRUE(pde.time_domain().rows() == M && pde.time_domain()(M - 1) == times(M - 1));
        DMatrix<double> nodes = pde.dof_coords(), qn = pde.quadrature_nodes();
        auto u = [](std::array<double, 3> x, double t) { return std::sin(2 * pi * x[0]) * std::sin(2 * pi * x[1]) * std::exp(-t); };
        DMatrix<double> g(nodes.rows(), M), u0(nodes.rows(), 1), f(qn.rows(), M);
        for (int64_t i = 0; i < nodes.rows(); ++i) {
            u0(i) = u(nodes.row3(i), 0.0);
            for (int j = 0; j < M; ++j) g(i, j) = u(nodes.row3(i), times(j));
        }
        for (int64_t i = 0; i < qn.rows(); ++i)
            for (int j = 0; j < M; ++j) f(i, j) = (8 * pi * pi - 1.) * std::sin(2 * pi * qn(i, 0)) * std::sin(2 * pi * qn(i, 1)) * std::exp(-times(j));
        pde.set_dirichlet_bc(g), pde.set_initial_condition(u0), pde.set_forcing(f);
        EXPECT_TRUE(pde.initial_condition().rows() == nodes.rows());
        pde.init();
        ErasedPDE later = pde;   // copied between init and solve
        pde.solve(), later.solve();
        EXPECT_TRUE(pde.success() && later.success() && pde.solution().cols() == M);
        double worst = 0;
        for (int64_t i = 0; i < nodes.rows(); ++i) worst = std::fmax(worst, std::fabs(pde.solution()(i, M - 1) - later.solution()(i, M - 1)));
        EXPECT_TRUE(worst == 0.0);   // the clone holds the same bits and runs the same deterministic solve
    }
}

// include/fdapde_hip.hpp on a host-only context (no device needed): copies share, the first one that asks for unique() leaves with a
// clone that rebuilt the same space (fdapde_ctx_clone), observers never count
TEST(context_handle_test, copy_on_write_host_only) {
    FixtureMesh<2, 2> m("unit_square_16");
    fdapde::hip::context_handle a(-1);
    const int64_t nn = m.mesh.n_nodes(), nc = m.mesh.n_cells();
    std::vector<int32_t> cells((size_t)nc * 3);
    std::vector<uint8_t> bnd((size_t)nn);
    for (int64_t c = 0; c < nc; ++c)
        for (int v = 0; v < 3; ++v) cells[(size_t)(c * 3 + v)] = m.mesh.cells()(c, v);
    for (int64_t i = 0; i < nn; ++i) bnd[(size_t)i] = m.mesh.boundary_nodes()(i) != 0;
    EXPECT_TRUE(fdapde_mesh_upload(a.get(), 2, 2, nn, m.mesh.nodes().data(), nc, cells.data(), bnd.data()) == FDAPDE_OK);
    int64_t n_dofs = 0;
    EXPECT_TRUE(fdapde_dofs_build(a.get(), 2, &n_dofs) == FDAPDE_OK && n_dofs == 1089);
    std::vector<uint8_t> mask((size_t)n_dofs, 0);
    mask[5] = 1;   // a boundary mask of the caller's own (fdapde_dofs_set_boundary) must travel with the clone
    EXPECT_TRUE(fdapde_dofs_set_boundary(a.get(), mask.data()) == FDAPDE_OK);
    EXPECT_TRUE(!a.shared() && a.unique() == a.get());
    fdapde::hip::context_handle b = a, watcher = a.observer();
    EXPECT_TRUE(a.shared() && b.get() == a.get() && watcher.get() == a.get());
    fdapde_ctx* const before = a.get();
    fdapde_ctx* const mine = b.unique();   // b leaves; a keeps the original and is its only owner again
    EXPECT_TRUE(mine != before && a.get() == before && !a.shared() && !b.shared() && a.unique() == before);
    int64_t nd2 = 0, nnz1 = 0, nnz2 = 0;
    EXPECT_TRUE(fdapde_sizes(mine, &nd2, &nnz2, nullptr, nullptr, nullptr) == FDAPDE_OK && fdapde_sizes(before, nullptr, &nnz1, nullptr, nullptr, nullptr) == FDAPDE_OK);
    EXPECT_TRUE(nd2 == n_dofs && nnz1 == nnz2);
    std::vector<int32_t> d1((size_t)nc * 6), d2((size_t)nc * 6), rp1((size_t)n_dofs + 1), rp2((size_t)n_dofs + 1), ci1((size_t)nnz1), ci2((size_t)nnz2);
    std::vector<uint8_t> m1((size_t)n_dofs), m2((size_t)n_dofs);
    EXPECT_TRUE(fdapde_dofs_get(before, d1.data(), m1.data(), nullptr) == FDAPDE_OK && fdapde_dofs_get(mine, d2.data(), m2.data(), nullptr) == FDAPDE_OK);
    EXPECT_TRUE(fdapde_pattern_get(before, rp1.data(), ci1.data()) == FDAPDE_OK && fdapde_pattern_get(mine, rp2.data(), ci2.data()) == FDAPDE_OK);
    EXPECT_TRUE(d1 == d2 && m1 == m2 && m1 == mask && rp1 == rp2 && ci1 == ci2);
    {
        fdapde::hip::context_handle moved = std::move(b);
        EXPECT_TRUE(!bool(b) && moved.get() == mine);
        a = moved;   // assignment: a drops its context (the watcher keeps it alive) and shares moved's
        EXPECT_TRUE(a.get() == mine && a.shared() && watcher.get() == before);
    }
    EXPECT_TRUE(!a.shared());   // `moved` is gone
    int64_t still = 0;
    EXPECT_TRUE(fdapde_sizes(watcher.get(), &still, nullptr, nullptr, nullptr, nullptr) == FDAPDE_OK && still == n_dofs);
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "16", 0);   // the sharded tests name ONE GPU several times: a hardware queue per rank launch (tests/conftest.py)
    if (argc < 2) { std::printf("usage: %s <tests/golden/mesh> [--io-only]\n", argv[0]); return 2; }
    MESH_PATH = argv[1];
    if (argc > 2 && std::string(argv[2]) == "--io-only") {   // host-side pieces of the facade: no device needed
        RUN(csv_reader_test, sparse_three_column_format);
        RUN(context_handle_test, copy_on_write_host_only);
        std::printf("%d checks, %d failures\n", checks, failures);
        return failures == 0 ? 0 : 1;
    }
    if (fdapde_device_count() < 1) { std::printf("no HIP device: these tests have no CPU fallback\n"); return 3; }
    RUN(fem_pde_test, laplacian_isotropic_order1);
    RUN(fem_pde_test, laplacian_isotropic_order2_callable_force);
    RUN(fem_pde_test, advection_diffusion_isotropic_order1);
    RUN(fem_pde_test, advection_diffusion_direct_solve_by_name);
    RUN(fem_pde_test, advection_diffusion_isotropic_order2);
    RUN(fem_pde_test, advection_diffusion_order2_two_level_solver_by_name);
    RUN(fem_operators_test, laplacian_order_2_through_stiff);
    RUN(sharded_test, laplacian_order1);
    RUN(sharded_test, laplacian_order2_callable_force);
    RUN(sharded_test, advection_diffusion);
    RUN(sharded_test, make_pde_copy_and_diverge);
    RUN(fem_pde_test, error_behaviour);
    RUN(fem_pde_test, laplacian_3d_order1);
    RUN(integration_test, integrate_over_triangulation);
    RUN(fem_pde_test, parabolic_isotropic_order2);
    RUN(fem_pde_test, parabolic_isotropic_order1_convergence);
    RUN(sparse_solver_test, factor_once_solve_many);
    RUN(linear_algebra_test, smw_and_lumping);
    RUN(mesh_test, neighbours_and_facets_match_the_fixture_files);
    RUN(lagrangian_basis_test, eval_basis_golden);
    RUN(type_erasure_test, make_pde_copy_the_handle_solve_both);
    RUN(type_erasure_test, eval_basis_and_space_time_slots);
    RUN(csv_reader_test, sparse_three_column_format);
    RUN(context_handle_test, copy_on_write_host_only);
    std::printf("%d checks, %d failures\n", checks, failures);
    return failures == 0 ? 0 : 1;
}
