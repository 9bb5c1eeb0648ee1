/*
 * fdapde_hip.h -- C ABI of the MI355X (gfx950) finite-element assemble-and-solve path.
 *
 * This is the drop-in boundary for fdaPDE-core's FEM hot path.  The reference has no FFI: its boundary is the
 * compile-time strategy interface  PDE<D,E,F,S,Ts...> -> pde_solver_selector<S,...>::type  (fdaPDE/pde/pde.h:52,
 * fdaPDE/pde/symbols.h:36, fdaPDE/finite_elements/solvers/fem_solver_selector.h:29-33).  A solver type plugged in
 * there must provide init(pde) / set_dirichlet_bc(pde) / solve(pde) and the getters solution() force() stiff() mass()
 * n_dofs() dofs() dofs_coords() (fdaPDE/pde/pde.h:85-105, fdaPDE/finite_elements/solvers/fem_solver_base.h:50-65).
 * The header-only C++20 facade in include/fdapde_amd/ implements exactly that interface on top of the entry points
 * below; each entry point names the reference member it replaces.  See INTEGRATION.md for the binding.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer in this header is a caller-owned HOST buffer unless its name ends
 *     in _dev.  Layouts are the reference's: nodes / dof coordinates / quadrature nodes column-major
 *     (Eigen DMatrix default, fdaPDE/geometry/triangulation.h:119), cells and DOF tables row-major int32 0-based
 *     (triangulation.h:120), sparse matrices as CSR with sorted columns, int32 indices, fp64 values (for the
 *     structurally symmetric FEM patterns these are the index arrays of the reference's CSC SpMatrix<double>).
 *   - every function returns an int status: FDAPDE_OK or an FDAPDE_E* code; fdapde_last_error() gives the text.
 *     Precondition violations that the reference reports by throwing std::runtime_error
 *     (fem_solver_base.h:146, fem_linear_elliptic_solver.h:36) map to FDAPDE_ENOTINIT; numerical failure of the
 *     solve (reference: success = false, fem_linear_elliptic_solver.h:42-45) maps to FDAPDE_ENOCONV.
 *   - one context per host thread; contexts are independent (the reference has no global state either).
 *   - there is NO CPU fallback: compute entry points on a context without a device fail with FDAPDE_ENODEVICE.
 */
#ifndef FDAPDE_HIP_H
#define FDAPDE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDAPDE_ABI_VERSION 5

enum {
    FDAPDE_OK = 0,
    FDAPDE_EINVAL = 1,    /* bad argument */
    FDAPDE_ENOMEM = 2,    /* host or device allocation failed */
    FDAPDE_ENODEVICE = 3, /* no HIP device / host-only context used for compute */
    FDAPDE_EHIP = 4,      /* a HIP runtime call failed */
    FDAPDE_ENOTINIT = 5,  /* call order violated ("solver must be initialized first!") */
    FDAPDE_ENOCONV = 6,   /* Krylov solve did not reach rtol within maxit (reference: success = false) */
    FDAPDE_EUNSUPPORTED = 7,
    FDAPDE_ERCCL = 8
};

/* leaf differential operators (fdaPDE/finite_elements/operators/{laplacian,diffusion,advection,reaction,dt}.h) */
enum { FDAPDE_LAPLACIAN = 0, FDAPDE_DIFFUSION = 1, FDAPDE_ADVECTION = 2, FDAPDE_REACTION = 3, FDAPDE_DT = 4 };

/* One scaled leaf of the operator expression.  The reference's expression algebra (unary minus, scalar *, binary
 * + and -, fdaPDE/pde/differential_expressions.h:49,95-96,114-118) collapses to a left-to-right sum of these. */
typedef struct {
    int32_t kind;          /* FDAPDE_LAPLACIAN ... FDAPDE_DT */
    int32_t space_varying; /* 0: cst[] holds the coefficient; 1: data points to per-quadrature-node values */
    double coef;           /* accumulated scalar factor (e.g. -1 for `-laplacian<FEM>()`) */
    double cst[9];         /* diffusion tensor K row-major NxN | advection vector b[N] | reaction c */
    const double *data;    /* row-major (nq*n_cells) x (N*N | N | 1); row nq*cell + q
                              (Discretized*Field::forward, fdaPDE/utils/integration/integrator.h:98-101) */
} fdapde_term;

/* CG_SR: single-reduction (Chronopoulos-Gear) CG: same iterates, both dot products fused into the SpMV, two launches and
 * (multi-GPU) one all-reduce per iteration */
/* CG_FUSED: the SpMV carries p.Ap and Ap.Ap, ONE kernel then updates x, r, p: alpha from the explicit r.r, beta from the
 * estimate alpha^2 Ap.Ap - r.r (only the search direction sees the estimate); two launches per iteration, single GPU */
/* FDAPDE_SOLVER_AUTO: Jacobi-PCG for a symmetric operator with a positive diagonal, Jacobi-BiCGStab otherwise; if CG breaks down (p.Ap <= 0: the
 * operator is symmetric but not positive definite, e.g. -Lap u - k^2 u) the solve is repeated with BiCGStab -- the reference's LU solves such systems too (one-GPU contexts; a rank of a multi-GPU job reports the breakdown).
 * Where BiCGStab gives up as well -- breakdowns beyond its restarts, a stalled or exploding recurrence: advection-dominated operators from cell Peclet
 * numbers of ~10^2 on -- the solve ends in restarted GMRES(50) on the same Jacobi-scaled system (kernels_gmres.h; one-GPU contexts), started from
 * BiCGStab's iterate when that is closer than zero; info.method_used names the stage that produced the answer.  A symmetric operator that is singular
 * up to rounding (pure Neumann data, right-hand side outside the range) is reported as a failure, not answered with a multiple of the null vector.
 * A method named explicitly is never replaced: its failure is reported (FDAPDE_ENOCONV, success = false). */
enum { FDAPDE_SOLVER_AUTO = 0, FDAPDE_SOLVER_CG = 1, FDAPDE_SOLVER_BICGSTAB = 2, FDAPDE_SOLVER_CG_SR = 3, FDAPDE_SOLVER_CG_FUSED = 4,
       FDAPDE_SOLVER_GMRES = 5 /* restarted GMRES(50), Jacobi-scaled; one-GPU contexts */,
       FDAPDE_SOLVER_DENSE = 6 /* the direct solve: the system is small enough (knob dense_rows, default 8192 DOFs; one-GPU contexts) for its
                                  dense inverse -- built ONCE on the device, ~ms, Gauss-Jordan with partial pivoting as one launch -- and the answer is one
                                  matrix-vector product (plus one step of iterative refinement where max |I - A X| says so).  Taken with the method left
                                  open by: fdapde_lin_solve once a handle has been asked for more than `dense_after` (2) columns ("factor once, solve many");
                                  fdapde_solve_parabolic with more than `dense_after` steps (K = M / dt + A is fixed: one inversion, then ONE product per step, u' = (K^-1 M / dt) u + K^-1 (f, g));
                                  fdapde_solve as the stage of FDAPDE_SOLVER_AUTO behind BiCGStab (which, with this stage behind it, may spend what the inversion
                                  will cost and no more) and in front of GMRES.  info.relres = max |I - A X|.  Both rent-or-buy rules go by dense_build_estimate_ms:
                                  0.5 ms at 289 rows, 2.6 ms at 1 089, 19 ms at 4 225, 0.11 s at 8 100.
                                  Asked for BY NAME it runs at once -- fdapde_solve: no Krylov stage in front; fdapde_lin_solve: the inverse is built by this call;
                                  fdapde_solve_parabolic: K inverted whatever the number of steps -- and nothing stands behind it: a matrix singular to working
                                  precision is FDAPDE_ENOCONV (success = false, like the reference's LU), a system it does not take FDAPDE_EUNSUPPORTED. */,
       FDAPDE_SOLVER_PMG = 7 /* order-2 spaces, one-GPU contexts: flexible GMRES with a TWO-LEVEL preconditioner -- a V(1,1) cycle:
                                damped Jacobi on the fine level, a correction from the P1 space of the same mesh (its own context inside this one: the same operator
                                terms, coefficient fields as their cell means; its systems solved to 1e-1 by the open method), damped Jacobi again -- 17 - 20
                                iterations whatever the mesh size where Jacobi-BiCGStab needs O(1 / h).  info.iters counts its iterations (three operator
                                applications and one coarse solve each), info.relres is the true relative residual of what is handed out.  fdapde_solve and
                                fdapde_solve_parabolic (the factor-once handle keeps the Jacobi-preconditioned stages); the open method takes it from
                                `pmg_auto_rows` (300 k) DOFs on -- below `pmg_auto_first_rows` (1 M) only from a context's second open-method solve on (the coarse level's
                                set-up is rent-or-buy); a parabolic run of more than four steps at once.  Knobs: pmg_outer 1 = BiCGStab around the additive form D^-1 + P A1^-1 P^T (the first
                                form), pmg_smooth 0 = flexible GMRES around that additive form, pmg_blocked 0 = the fine operator through the CSR kernel. */ };
/* ROWS: row-owner sweep (default; no atomics, bitwise reproducible).  The others are element-wise scatter forms kept as measured
 * alternatives and cross-checks: ATOMIC / COLOURED = lane per (cell, row) with a slot search, fp64 atomics / one launch per colour;
 * PARTITIONED = one workgroup per cell partition, colours walked inside the workgroup, atomics only on rows shared between
 * partitions, slot map streamed; WAVE = one wavefront per element, lane = (i, j, quadrature node) -- a P2 element in passes of 8 (i, j)
 * pairs --, one launch per colour, slot map streamed. */
enum { FDAPDE_ASSEMBLY_ROWS = 0, FDAPDE_ASSEMBLY_ATOMIC = 1, FDAPDE_ASSEMBLY_COLOURED = 2, FDAPDE_ASSEMBLY_PARTITIONED = 3,
       FDAPDE_ASSEMBLY_WAVE = 4 };
enum { FDAPDE_MAT_STIFF = 0, FDAPDE_MAT_MASS = 1 };

typedef struct {
    int32_t method;   /* FDAPDE_SOLVER_*; AUTO = CG for symmetric operators (CG_FUSED on one GPU, CG_SR on several), BiCGStab otherwise */
    int32_t maxit;    /* <= 0: 10 * n_dofs capped at 100000 */
    double rtol;      /* <= 0: 1e-10.  Stop when ||r||_{D^-1} <= rtol * ||r0||_{D^-1} (D = diag A) */
    int32_t assembly; /* FDAPDE_ASSEMBLY_*; used by fdapde_init */
    int32_t check_every; /* iterations between host convergence polls; <= 0: 32 */
    int32_t time_spmv;   /* > 0: time the SpMV launch of `time_spmv` Krylov iterations (every 8th one, max 256 samples) with HIP
                            events attached to those dispatches on the context's stream; average in fdapde_info.spmv_avg_ms */
} fdapde_options;

typedef struct {
    int32_t iters;     /* Krylov iterations performed */
    int32_t converged; /* 1 iff the stopping rule was met */
    double relres;     /* final ||r||_{D^-1} / ||r0||_{D^-1} */
    double t_assemble_ms; /* device time of the last fdapde_init (stiff + force + mass), HIP events */
    double t_solve_ms;    /* device time of the last fdapde_solve (Dirichlet reduction + Krylov), HIP events */
    double t_setup_ms;    /* host wall time of the last fdapde_dofs_build (numbering, pattern, upload) */
    double spmv_avg_ms;   /* average duration of the SpMV launches timed inside the last solve (0 if none) */
    int32_t spmv_timed;   /* how many launches that average covers */
    int32_t method_used;  /* the FDAPDE_SOLVER_* that ran */
    int32_t persistent;   /* 1: the solve ran as ONE persistent launch (x, r, p in registers, in-kernel hand-offs; matrix resident in LDS or
                             streamed once per iteration).  spmv_avg_ms is then the operator-application phase (SpMV + neighbour
                             import) stamped inside the kernel: the average, over the iterations, of the SLOWEST workgroup's phase */
    double gather_avg_ms; /* persistent path: all-gather phase of the dot products per iteration (mean over workgroups; contains
                             their wait for the slowest workgroup's operator phase) */
    double update_avg_ms; /* persistent path: vector update phase per iteration (mean over workgroups) */
    double spmv_mean_ms;  /* persistent path: operator-application phase, mean over workgroups */
    double launch_ms;     /* persistent path: duration of the ONE launch that ran the whole Krylov iteration (HIP events recorded on the
                             context's stream right before and after the dispatch); 0 on the multi-launch path */
} fdapde_info;

typedef struct fdapde_ctx fdapde_ctx;

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
int fdapde_abi_version(void);
int fdapde_device_count(void);
/* device >= 0: bind the context to that HIP device.  device < 0: host-only context (numbering / pattern queries
 * work, every compute call fails with FDAPDE_ENODEVICE). */
int fdapde_ctx_create(int device, fdapde_ctx **ctx);
void fdapde_ctx_destroy(fdapde_ctx *ctx);
/* An independent context (same device) holding the same problem as `src`: mesh, function space and boundary mask, operator / forcing /
 * Dirichlet data, the assembled stiff_ / mass_ / force_, the solution and the factor-once handle's matrix -- every getter returns what
 * it returns on `src`, and fdapde_solve / fdapde_lin_solve may be called on it without another fdapde_init.  This is the copy operation
 * behind the reference's type-erased PDE handle, which deep-copies the PDE with its solver on construction and on every handle copy
 * (make_pde -> erase<heap_storage, PDE__>: fdaPDE/pde/pde.h:167-169, fdaPDE/utils/type_erasure.h:124-146); the host-side bindings share
 * one context between copies and clone only when one of them is about to change it (include/fdapde_hip.hpp).  The function space is
 * rebuilt (same deterministic set-up, bit-identical index arrays), the data are copied device to device; tuning knobs and solver
 * layouts are not carried over.  A context that is a rank of a multi-GPU job is refused (FDAPDE_EUNSUPPORTED). */
int fdapde_ctx_clone(const fdapde_ctx *src, fdapde_ctx **out);
/* ONE context over SEVERAL devices of a node -- the mesh sharded behind the reference's one-object, one-thread interface (fdaPDE/pde/pde.h:58-105
 * holds one PDE; north_star shards its mesh over the GPUs).  The context handed out is used exactly like a single-device one: every entry point of
 * this header below takes it, every array keeps the reference's numbering of the WHOLE mesh.  Inside, fdapde_dofs_build splits the resident mesh on
 * the device (Morton chunks of the cell barycentres, node owners dealt in patches: csrc/dev_partition.hip), builds one rank context per device --
 * each driven by a worker thread of the library -- and wires them through an in-process transport; fdapde_init / fdapde_solve /
 * fdapde_solve_parabolic / fdapde_lin_solve run on all devices at once: the row-distributed form (complete rows per owner, the whole Krylov
 * iteration as one persistent launch per device, launches exchanging through peer-mapped boards) where the library takes the system, the
 * element-partitioned neighbour exchange otherwise (the context changes form by itself).  A device may be named several times (its CUs are then
 * shared out: how the tests run 2 - 8 "devices" on one GPU -- the ranks' persistent launches then need a hardware queue each: set GPU_MAX_HW_QUEUES >= n + 2
 * in the environment before the process first touches the GPU, or the row-distributed launches time out and the context takes the element form; devices of
 * their own have queues of their own).  devices[0] also keeps the whole mesh and function space for the index getters.
 * fdapde_ctx_clone gives another multi-device context on the same devices (the split is deterministic; the assembled and solved state travels
 * device to device, rank by rank).  Not available on such a context: fdapde_comm_*, fdapde_halo_setup*, fdapde_rowdist_setup, fdapde_partition_build,
 * fdapde_bench_spmv, fdapde_solver_layout* (FDAPDE_EUNSUPPORTED). */
int fdapde_ctx_create_multi(const int32_t *devices, int32_t n_devices, fdapde_ctx **ctx);
/* devices of a context (1 and its own device for a single-device one), the form its ranks are in (0 row-distributed, 1 element partition,
 * -1 single device), and what the last fdapde_dofs_build spent on the split: partitioning on the device / the ranks' set-up.  Any pointer may be NULL. */
int fdapde_ctx_devices(const fdapde_ctx *ctx, int32_t *n_devices, int32_t *devices, int32_t *form, double *t_partition_ms,
                       double *t_rank_setup_ms);
const char *fdapde_last_error(const fdapde_ctx *ctx);
const char *fdapde_status_string(int status);

/* ---- domain: Triangulation<M,N>(nodes, cells, boundary)  (fdaPDE/geometry/triangulation.h:49-60,143,319) ------- */
int fdapde_mesh_upload(fdapde_ctx *ctx, int M, int N, int64_t n_nodes, const double *nodes_colmajor, int64_t n_cells,
                       const int32_t *cells_rowmajor, const uint8_t *boundary_nodes);

/* The rest of the Triangulation constructor (geometry/triangulation.h:143-196 triangles, 319-399 tetrahedra), built on the device:
 * facets = edges of triangles / faces of tetrahedra in the reference's first-seen numbering (cells ascending x combinations<M,M+1>),
 *   facet_nodes  n_facets x M ascending node ids (edges_ / faces_), facet_cells n_facets x 2 with -1 on the boundary (edge_to_cells_
 *   / face_to_cells_), facet_boundary (edges_markers_ / faces_markers_: seen by exactly one cell), cell_facets n_cells x (M+1)
 *   (cell_to_edges_ / cell_to_faces_), neighbors n_cells x (M+1) with -1 for none, column = local vertex opposite to the shared
 *   facet (neighbors_, triangulation.h:56-57,184-185,381-382);
 * tetrahedra also: edge_nodes n_edges x 2 (edges_, numbered through the newly seen faces, 365-377), edge_boundary (both end nodes on
 *   the boundary, 371), face_edges n_facets x 3 (face_to_edges_).  For triangles n_edges == n_facets and those three stay untouched.
 * All row-major int32, 0-based; any output pointer may be NULL. */
int fdapde_topology_build(fdapde_ctx *ctx, int64_t *n_facets, int64_t *n_edges);
int fdapde_topology_get(fdapde_ctx *ctx, int32_t *neighbors, int32_t *cell_facets, int32_t *facet_nodes, int32_t *facet_cells,
                        uint8_t *facet_boundary, int32_t *edge_nodes, uint8_t *edge_boundary, int32_t *face_edges);

/* ---- function space: LagrangianBasis<D,R>(domain) -> enumerate_dofs  (basis/lagrangian_basis.h:94-136,147) ----- */
/* Builds the DOF table in the reference's numbering, the CSR pattern, and the device-side layout. */
int fdapde_dofs_build(fdapde_ctx *ctx, int order, int64_t *n_dofs);
/* FEMSolverBase::n_dofs(), dofs(), boundary dofs, dofs_coords() (fem_solver_base.h:57-59, lagrangian_basis.h:155-183).
 * Any output pointer may be NULL.  dofs row-major n_cells x n_basis; coords column-major n_dofs x N. */
int fdapde_dofs_get(const fdapde_ctx *ctx, int32_t *dofs_rowmajor, uint8_t *boundary_dofs, double *dof_coords_colmajor);
/* Overrides the boundary-DOF mask (n_dofs flags, reference numbering) that fdapde_set_dirichlet / fdapde_solve use.  The
 * reference always takes basis_.boundary_dofs() (fem_solver_base.h:143-153); an element-partitioned rank needs the override
 * because the 2-D rule "edge seen by one cell" (triangulation.h:177,187) would mark its interface edges on the sub-mesh. */
int fdapde_dofs_set_boundary(fdapde_ctx *ctx, const uint8_t *boundary_dofs);
int fdapde_sizes(const fdapde_ctx *ctx, int64_t *n_dofs, int64_t *nnz, int32_t *n_basis, int32_t *n_quadrature,
                 int64_t *n_edges);
/* sparsity pattern of stiff()/mass() in the reference's numbering: rowptr[n_dofs+1], colidx[nnz] sorted per row */
int fdapde_pattern_get(const fdapde_ctx *ctx, int32_t *rowptr, int32_t *colidx);
/* Integrator::quadrature_nodes (utils/integration/integrator.h:109-121): column-major (nq*n_cells) x N */
int fdapde_quadrature_nodes(fdapde_ctx *ctx, double *out_colmajor);

/* ---- problem data: PDE::set_differential_operator / set_forcing / set_dirichlet_bc (pde/pde.h:74-77) ----------- */
int fdapde_set_operator(fdapde_ctx *ctx, int32_t n_terms, const fdapde_term *terms);
/* forcing sampled at quadrature nodes: column-major (nq*n_cells) x n_cols (n_cols > 1 only for parabolic problems) */
int fdapde_set_forcing(fdapde_ctx *ctx, const double *f_q, int32_t n_cols);
/* Dirichlet data indexed by DOF id (fem_solver_base.h:152); NULL clears (PDE::solve then skips set_dirichlet_bc) */
int fdapde_set_dirichlet(fdapde_ctx *ctx, const double *g);

/* ---- FEMSolverBase::init (fem_solver_base.h:104-139): stiff_, force_, mass_ ------------------------------------ */
int fdapde_init(fdapde_ctx *ctx, const fdapde_options *opt);
/* Assembler::discretize_operator for an arbitrary operator into slot `which` (fem_assembler.h:52-121) */
int fdapde_assemble_operator(fdapde_ctx *ctx, int32_t which, int32_t n_terms, const fdapde_term *terms, int32_t assembly);

/* ---- PDE::solve (pde/pde.h:102-105): set_dirichlet_bc (fem_solver_base.h:142-155) + elliptic solve
 *      (fem_linear_elliptic_solver.h:34-50; Eigen::SparseLU replaced by Jacobi-PCG / BiCGStab on the interior block) */
/* One-time preparation (host index work + uploads) of the solver's compact matrix layout for the current boundary-DOF mask;
 * set-up like fdapde_dofs_build.  Optional: the first fdapde_solve / fdapde_lin_solve does it lazily. */
int fdapde_solver_prepare(fdapde_ctx *ctx, int32_t with_dirichlet);
int fdapde_solve(fdapde_ctx *ctx, const fdapde_options *opt, fdapde_info *info);

/* ---- FEMLinearParabolicSolver::solve (finite_elements/solvers/fem_linear_parabolic_solver.h:37-72): implicit Euler ---------
 * Needs fdapde_init with one forcing column per time point (fem_solver_base.h:118-128).  delta_t = times[1] - times[0] (line
 * 42); initial_condition[n_dofs] (pde.h:77); dirichlet column-major n_dofs x n_times or NULL (column i+1 is imposed at step
 * i, line 66); solution column-major n_dofs x n_times, column 0 = initial condition (line 46).
 * info.iters = Krylov iterations over all steps, info.relres = worst step. */
int fdapde_solve_parabolic(fdapde_ctx *ctx, const fdapde_options *opt, int32_t n_times, double delta_t,
                           const double *initial_condition, const double *dirichlet, double *solution, fdapde_info *info);

/* ---- "factor once, solve many": fdapde::SparseLU<SpMatrix<double>> (fdaPDE/utils/symbols.h:133-160), the handle SMW
 *      (fdaPDE/linear_algebra/smw.h:38-59) and the downstream models solve against ---------------------------------------
 * fdapde_lin_compute = ::compute(matrix): `values` = nnz entries aligned with fdapde_pattern_get (any matrix on the FEM
 * pattern; symmetric != 0 allows CG), or NULL to take the assembled matrix `which`.  No Dirichlet reduction is applied.
 * fdapde_lin_solve = ::solve(b): dense right-hand sides, b and x column-major n_dofs x n_rhs.  Several columns are worth handing over
 * together: the columns of a small system run side by side in one launch (6 us instead of 300 us per column for 289 DOFs x 64).
 * x may overlap b (an in-place solve): the call then works from a private copy of the right-hand sides. */
int fdapde_lin_compute(fdapde_ctx *ctx, int32_t which, const double *values, int32_t symmetric);
int fdapde_lin_solve(fdapde_ctx *ctx, const fdapde_options *opt, const double *b, int32_t n_rhs, double *x, fdapde_info *info);

/* ---- basis evaluation (PDE__::eval_basis, pde/pde.h:149-158; policies basis/lagrangian_basis.h:203-283) -------------------
 * fdapde_eval_pointwise: for each location (column-major n_locs x N) the containing cell (reference cell id, -1 if outside;
 * TreeSearch::locate, geometry/tree_search.h:73-90) and the n_basis values psi_h(invJ (p - x0)), row-major n_locs x n_basis.
 * Psi(i, dofs(cell_i, h)) = values[i*n_basis + h], D = ones (pointwise_evaluation::eval).
 * fdapde_cell_integrals: measure[n_cells] and psi_int[n_cells x n_basis] = int_e psi_h, reference cell order; with an
 * incidence matrix they give Psi(k, dofs(e,h)) += psi_int[e][h] / |D_k|, D_k = sum measure (areal_evaluation::eval). */
int fdapde_eval_pointwise(fdapde_ctx *ctx, int64_t n_locs, const double *locs_colmajor, int32_t *cell_ids, double *values);
int fdapde_cell_integrals(fdapde_ctx *ctx, double *measure, double *psi_int);

/* ---- getters (fem_solver_base.h:50-53) ------------------------------------------------------------------------- */
/* values[nnz] aligned with fdapde_pattern_get.  After a solve with Dirichlet data, FDAPDE_MAT_STIFF is the
 * row-zeroed matrix the reference leaves in stiff_ (rows of boundary DOFs zero, unit diagonal). */
int fdapde_matrix_values(fdapde_ctx *ctx, int32_t which, double *values);
/* lump(stiff() | mass()) (fdaPDE/linear_algebra/lumping.h:30-41): diagonal of the row-sum lumped matrix, diag[n_dofs] */
int fdapde_lump(fdapde_ctx *ctx, int32_t which, double *diag);
int fdapde_force(fdapde_ctx *ctx, double *force);       /* n_dofs * n_cols; boundary rows = g after a Dirichlet solve */
int fdapde_solution(fdapde_ctx *ctx, double *solution); /* n_dofs */
int fdapde_info_get(const fdapde_ctx *ctx, fdapde_info *info);

/* ---- building blocks exposed for parity tests and the roofline benchmark --------------------------------------- */
/* y = A x in the reference's DOF numbering, host buffers */
int fdapde_spmv(fdapde_ctx *ctx, int32_t which, const double *x, double *y);
/* times `reps` launches of the solver's SpMV kernel (the one inside CG) with HIP events on the context's stream;
 * returns the average kernel duration in ms and the algorithmic bytes per launch
 * (12*nnz + 4*(n+1) + 16*n) */
int fdapde_bench_spmv(fdapde_ctx *ctx, int32_t reps, double *avg_ms, double *algorithmic_bytes);
/* the operator the in-solve SpMV applies, for the roofline figures: rows and entries of the interior block A_II (the reference's
 * set_dirichlet_bc leaves those rows active, fem_solver_base.h:142-155) seen as a plain CSR matrix, and the bytes one launch of
 * the solver's kernel streams from its compact coded layout.  Builds the layout if fdapde_solver_prepare has not. */
int fdapde_solver_layout(fdapde_ctx *ctx, int32_t with_dirichlet, int64_t *n_interior, int64_t *nnz_interior,
                         double *streamed_bytes);
/* Which layout that is: kind 0 compact CSR (k_spmv_team2), 1 blocked ELL (k_spmv_blocked), 2 single persistent launch with the blocks
 * streaming from memory every iteration, 3 single persistent launch with the blocks resident in LDS (then streamed_bytes above is what
 * the launch stages once, and an iteration moves the exchanged granules only); symmetric storage, workgroups, rows per thread. */
int fdapde_solver_layout_kind(fdapde_ctx *ctx, int32_t with_dirichlet, int32_t *kind, int32_t *symmetric_storage,
                              int32_t *workgroups, int32_t *rows_per_thread);
/* ---- multi-GPU: element-partitioned meshes, one context (= one rank) per GPU --------------------------------------------
 * No reference counterpart (the reference is single-threaded, single address space).  Each rank uploads the sub-mesh of
 * its own cells (local node numbering), assembles its sub-assembled operator with the calls above, and the solve sums the
 * interface ("halo") DOF contributions over the ranks sharing them with one RCCL all-reduce per operator application
 * (fused with the p.Ap partial) plus one scalar all-reduce per iteration.  The partitioner: fdapde_partition_build below (fdapde-core_amd/dist.py: the numpy original it is tested against).
 *   fdapde_comm_unique_id : rank 0 creates the 128-byte RCCL id; the caller broadcasts it (e.g. torch.distributed)
 *   fdapde_comm_init      : every rank joins the communicator on its context's device and stream
 *   fdapde_halo_setup     : interface maps.  n_if_global = number of interface DOFs of the whole mesh; local_dof[k] (this
 *                           rank's DOF id, reference numbering of the LOCAL space) <-> if_index[k] (slot in the global
 *                           interface vector), k < n_if_local; owned[d] = 1 iff this rank counts local DOF d in global
 *                           dot products (every global DOF is owned by exactly one rank).
 * After fdapde_halo_setup every solve of the context is element-partitioned: fdapde_solve (single-reduction CG or BiCGStab),
 * fdapde_solve_parabolic, fdapde_lin_solve (right-hand sides sub-assembled, i.e. summed over the ranks sharing a DOF).
 * (One PROCESS driving several devices needs none of this: fdapde_ctx_create_multi above.) */
/* host-staged transport instead of RCCL: fn(user, host_buf, count) must replace host_buf by its sum over all ranks and
 * return 0.  Lets the distributed path run over any fabric (tests drive it with torch.distributed/gloo, two ranks on one GPU). */
typedef int (*fdapde_allreduce_fn)(void *user, double *host_buf, int64_t count);
int fdapde_comm_init_callback(fdapde_ctx *ctx, int32_t world, int32_t rank, fdapde_allreduce_fn fn, void *user);
int fdapde_comm_unique_id(void *out128);
int fdapde_comm_init(fdapde_ctx *ctx, int32_t world, int32_t rank, const void *unique_id128);
/* sum (op 0) or max (op 1) of n host doubles over the ranks of the context's communicator, in place -- the barrier / timing
 * reductions of a multi-process driver that loads no other GPU library (bench.py's ranks: ONE HIP / RCCL stack per process).
 * RCCL is dlopen'ed from the installation the HIP runtime this library is bound to belongs to; fdapde_comm_library() names it. */
int fdapde_comm_allreduce(fdapde_ctx *ctx, double *host_inout, int32_t n, int32_t op);
/* the number of ranks of the context's communicator as RCCL itself reports it (ncclCommCount) -- so that a multi-GPU result can state what
 * its collectives really ran over; the registered world size under the host-staged transport; 1 without a communicator */
int fdapde_comm_count(fdapde_ctx *ctx, int32_t *ranks);
const char *fdapde_comm_library(void);
int fdapde_halo_setup(fdapde_ctx *ctx, int64_t n_if_global, int64_t n_if_local, const int32_t *local_dof,
                      const int32_t *if_index, const uint8_t *owned);
/* Neighbour-only exchange instead of the dense interface vector of fdapde_halo_setup: per operator application a rank sends each
 * PEER (a rank it shares DOFs with) its sub-assembled values at the shared DOFs and receives the peer's, in one grouped RCCL call
 * (ncclSend / ncclRecv per peer), followed by an all-reduce of the two fused dot partials; message size = shared DOFs x 8 bytes per peer, not
 * the whole interface.  peer_rank: n_peers ranks, ascending, without this one; peer_off [n_peers + 1]: segment of each peer in
 * peer_dof; peer_dof: DOF ids (reference numbering of this rank's sub-mesh) shared with the peer -- BOTH ranks of a pair must list their
 * shared DOFs in the same order (e.g. ascending global key).  A DOF shared by k ranks appears in the lists of all k - 1 peers; its
 * contributions are added in ascending rank order on every sharer, so all of them hold the same bits.  owned as in fdapde_halo_setup.
 * Over the host-staged transport (fdapde_comm_init_callback) the exchange itself goes through fdapde_comm_set_exchange_callback's fn:
 * fn(user, n_peers, peer_rank, peer_off, send, recv) must deliver send[peer_off[q] .. peer_off[q + 1]) to rank peer_rank[q] and fill the
 * same segment of recv with what that rank sent to this one; return 0. */
typedef int (*fdapde_exchange_fn)(void *user, int32_t n_peers, const int32_t *peer_rank, const int64_t *peer_off, const double *send,
                                  double *recv);
int fdapde_comm_set_exchange_callback(fdapde_ctx *ctx, fdapde_exchange_fn fn, void *user);
int fdapde_halo_setup_peers(fdapde_ctx *ctx, int32_t n_peers, const int32_t *peer_rank, const int64_t *peer_off,
                            const int32_t *peer_dof, const uint8_t *owned);

/* Row-distributed form -- the multi-GPU solve that keeps the single-GPU solver's structure: every DOF of the whole mesh is OWNED by
 * exactly one rank, and a rank's sub-mesh holds every cell that touches one of its DOFs (its cells of the element partition + one layer
 * of its neighbours' cells), so fdapde_init completes the rows of the owned DOFs with no exchange at all (the halo contributions are
 * summed by the assembly itself).  fdapde_solve then runs the WHOLE Jacobi-PCG as one persistent launch per rank; the workgroups of all
 * ranks act as one grid: entries of the search direction another rank needs are pushed into that rank's board through peer-mapped
 * pointers (hipIpc over xGMI), the dot records of all workgroups are all-gathered the same way, every workgroup on every GPU sums them
 * in the same order (bitwise identical scalars, identical stop decision) -- no RCCL call inside the iteration.  RCCL (or the host-staged
 * transport) carries set-up data only: the layout agreement, the board handles, and per solve the Jacobi scale of the ghost columns
 * and three scalars.
 *   dof_key[d]   : global identity of local DOF d (reference numbering of this rank's space): node id, or n + lo * n + hi for an edge DOF
 *   dof_owner[d] : the rank that owns it; every cell touching a DOF must be in its owner's sub-mesh.  Boundary flags must be the whole
 *                  mesh's (fdapde_dofs_set_boundary where the sub-mesh's own rule would differ).
 * fdapde_solve (CG for symmetric operators, BiCGStab otherwise: <= 16 / <= 8 rows per thread and workgroup), fdapde_solve_parabolic
 * (initial condition and Dirichlet data given for ALL local DOFs; the stepper imports the ghost entries of every step's solution) and
 * fdapde_lin_solve (right-hand sides complete at the owned DOFs) take this path; a system that does not fit returns FDAPDE_EUNSUPPORTED
 * on every rank and the caller uses the element-partitioned exchange above.  Solution entries of DOFs owned by other ranks: the
 * Dirichlet lift where there is one, otherwise not computed (the parabolic stepper returns the imported values). */
int fdapde_rowdist_setup(fdapde_ctx *ctx, const int64_t *dof_key, const int32_t *dof_owner);

/* The partitioner by itself, for rank PROCESSES (one context per process, as above): every process uploads the whole mesh to its own device,
 * calls fdapde_partition_build and takes its rank's share -- nothing is shipped between processes.  On the mesh resident on the device since
 * fdapde_mesh_upload; form 0: row-distributed (a rank's sub-mesh = every cell touching a node it owns; owners = the lowest / highest rank
 * touching a node by checkerboard box), form 1: element partition (sub-mesh = the rank's Morton chunk of the cells; owner = the lowest rank
 * touching a node).  world <= 64.  Array for array what fdapde-core_amd/dist.py computes with numpy (tests/test_gpu_partition.py).
 *   fdapde_partition_get   : a rank's sub-mesh in the layouts of fdapde_mesh_upload, the global ids of its nodes and cells (ascending: local
 *                            numbering = ascending global id) and the owning rank of every local node; any pointer may be NULL
 *   fdapde_partition_whole : per cell of the whole mesh its rank, per node its owner and the bit mask of the ranks whose sub-mesh holds it
 *   fdapde_partition_peers : form 1, P1 (DOF = node): the lists fdapde_halo_setup_peers takes -- call with NULL arrays for the sizes first */
int fdapde_partition_build(fdapde_ctx *ctx, int32_t world, int32_t form);
int fdapde_partition_sizes(const fdapde_ctx *ctx, int32_t rank, int64_t *n_nodes, int64_t *n_cells);
int fdapde_partition_get(fdapde_ctx *ctx, int32_t rank, double *nodes_colmajor, int32_t *cells_rowmajor, uint8_t *boundary_nodes,
                         int64_t *node_ids, int64_t *cell_ids, int32_t *node_owner);
int fdapde_partition_whole(fdapde_ctx *ctx, int32_t *cell_rank, int32_t *node_owner, uint64_t *node_ranks);
int fdapde_partition_peers(fdapde_ctx *ctx, int32_t rank, int32_t *n_peers, int32_t *peer_rank, int64_t *peer_off, int32_t *peer_node,
                           uint8_t *owned, int64_t *n_shared);

/* tuning / diagnostic knobs (A/B measurements inside one process; defaults are the measured best):
 *   SpMV launch   "spmv_variant" (2 pair form, 0 team form, 1 stream form), "spmv_team", "spmv_unroll", "spmv_bpx" (workgroups
 *                 per XCD band), "spmv_ablate" (diagnostic instantiations), "spmv_c16" (16-bit column codes), "spmv_deep"
 *                 (gathers one tile ahead), "spmv_ntv" (-1 auto / 0 / 1: nontemporal value stream)
 *   fused CG      "cgf_v" (double2 per lane), "cgf_band" (XCD-aware mapping), "cgf_nt" (bit set: y, x, r, p nontemporal),
 *                 "cgf_lazy" (x touched every second launch), "cgf_split" (second half of the loads after the scalars),
 *                 "use_graph" (hipGraph replay of a chunk of iterations)
 *   handle        "multi_rhs" (batched multi-column solves)
 *   assembly      "asm_fq_block" (forcing as per-visit load coefficients computed by a kernel of their own inside init),
 *                 "asm_fq_bc" (0: the sweep gathers the forcing samples by cell id instead of reading their block-cell ordered copy),
 *                 "bicg_restart" (0: a BiCGStab breakdown -- rho, r0.v or omega exactly 0 -- ends the solve with FDAPDE_ENOCONV instead of restarting it from
 *                 the iterate reached, up to 30 times),
 *                 "asm_split_varying" (0: operators whose advection / reaction vary but whose diffusion part does not take the per-node tensor
 *                 integrand of the fully space-varying case instead of constants-through-reference-tensors + per-node vector / scalar),
 *                 "asm_items" (0: spaces whose rows are dealt by visit count -- P2 -- keep the row-walking sweep instead of the visit-parallel one),
 *                 "asm_fuse_mass" (0: the mass matrix always in a sweep of its own; 1: in the operator's sweep where both accumulator
 *                 ranges fit 64 KB of LDS; 2: wherever they fit at all)
 *   single-launch CG  "persist" (0: never run the solve as one persistent launch), "persist_time" (phase stamps), "persist_sym"
 *                 (0 plain storage, 1 symmetric storage, 2 symmetric where the plain blocks would stream), "persist_balance"
 *                 (workgroup boundaries at equal cost / equal row counts), "blocked" (blocked-ELL SpMV of the multi-launch solves),
 *                 "persist_coop" (0: plain instead of cooperative launch), "persist_timeout_us" (bound of every in-kernel wait),
 *                 "persist_debug_stall" / "persist_retry" (tests: force a hand-off timeout at an iteration / forget one),
 *                 "persist_cols" (0: the columns of fdapde_lin_solve always one launch each, never side by side in one),
 *                 "persist_wide" (0: systems of more than 8 192 rows per workgroup -- 2.1 to 3.1 M rows on 256 CUs -- keep the multi-launch path instead of
 *                 the single launch with x in HBM and 24 rows per thread), "persist_max_wg" (tests: fewer workgroups than CUs for the single launch),
 *                 "persist_direct" (0: a single right-hand side of a one-workgroup system takes the general path instead of the launch that
 *                 reads b and writes x and its outcome through pinned host memory itself), "persist_direct_spin_us" (host spin on that outcome),
 *                 "persist_single_rows" (systems of up to that many interior rows run as one workgroup, without hand-offs),
 *                 "persist_prefetch" (0: the streaming forms do not touch the next operator application's first lines during the dot all-gather),
 *                 "persist_exp_lds" (0: the symmetric streaming form re-reads its export list from global memory every iteration)
 *   solve         "dense_rows" (systems of up to that many DOFs may take the dense inverse; 0: never), "dense_after" (columns / steps before it is built),
 *                 "dense_block" (0: the pivot-by-pivot inversion), "dense_multi" (0: above 2 048 rows ONE panel workgroup with a panel of 8 / 4 columns instead of several with 16),
 *                 "dense_fold" (0: the stepper's dense loop as four launches per step instead of one product),
 *                 "auto_gmres" (0: the open method ends with BiCGStab), "gmres_m" (restart length, default 50),
 *                 "small_rows" (systems of up to that many DOFs: no wait for the positive-diagonal flag, outcome through a pinned record; 0: off),
 *                 "small_front_rows" (one-workgroup systems of up to that many DOFs: ONE kernel in front of the single launch -- k_small_front --
 *                 and the epilogue inside the launch; 0: the separate launches), "asm_items_fuse" (0: the P2 mass matrix in a sweep of its own)
 *   measurement   knobs kept for the A/B figures of DESIGN.md: "asm_row_stat" (0: the solve's Jacobi scaling reads the whole matrix instead of the (diagonal, row maximum)
 *                 pairs fdapde_init leaves), "bicg_shadow" (shadow residual of the multi-launch BiCGStab: 0 = r0, 1 = pseudo-random, 2 = r0 with randomly scaled entries),
 *                 "persist_bicg" (0: non-symmetric systems always take the multi-launch BiCGStab), "persist_fill_fused" (1: the single launch's blocks filled straight from
 *                 the unscaled matrix), "persist_late" (CG layouts with late-import workgroups instead of doubled rows per thread), "persist_gather_waves" /
 *                 "persist_poll_sleep" (the dot all-gather of the single launch), "persist_wide_gj" (passes of a phase of the wide form that load together: 4 / 6 / 12),
 *                 "dense_bulk" (0: many columns staged by the kernels themselves instead of by DMA), "dense_hostb" / "dense_direct" (1: a single column's product reads b
 *                 from / hands x over to the pinned block itself -- both measured slower), "rowdist_share" (that many ranks share this device: an equal share of its CUs
 *                 each), "rowdist_max_wg" (workgroups of this rank's launch), "rowdist_flat_gather" (-1 auto / 0 / 1: the dot gather across ranks in one hop),
 *                 "rowdist_timeout_first_ms" (bound of the waits of iteration 0: launch skew between the ranks)
 *   two-level     "pmg_auto" (0: the open method never takes FDAPDE_SOLVER_PMG), "pmg_auto_rows" / "pmg_auto_first_rows" (order-2 systems of at least that many DOFs
 *                 take it: from a context's second open-method solve on / at once; 300 000 / 1 000 000), "pmg_inner_tol_exp" (the coarse solves stop at 10^-exp; 1),
 *                 "pmg_inner_maxit" (their budget; 200), "pmg_restart" (vectors per cycle of the flexible GMRES, 2 .. 50), "pmg_outer" (1: BiCGStab around the
 *                 additive preconditioner, the round's first form), "pmg_smooth" (0: flexible GMRES around the additive preconditioner instead of the V(1,1)
 *                 cycle), "pmg_blocked" (0: the fine operator through the CSR kernel instead of the blocked-ELL SpMV), "pmg_setup_check" (1: the transfer tables
 *                 are also built by host loops and compared; an error if they differ) */
int fdapde_tune(fdapde_ctx *ctx, const char *key, int32_t value);
/* the context's HIP stream (hipStream_t) so that callers can bracket work with their own events */
void *fdapde_stream(fdapde_ctx *ctx);
int fdapde_synchronize(fdapde_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* FDAPDE_HIP_H */
