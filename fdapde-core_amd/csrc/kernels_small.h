// Small systems: what fdapde_solve enqueues in front of the single launch, as one kernel.
#ifndef FDAPDE_HIP_KERNELS_SMALL_H
#define FDAPDE_HIP_KERNELS_SMALL_H
#include "kernels_krylov.h"
#include "kernels_persist.h"

namespace fdapde_hip {

// ---------------------------------------------------------------------------------------------------------------------
// k_small_front: everything fdapde_solve enqueues in FRONT of the single launch of a small system (one workgroup, plain storage), as ONE
// workgroup's work -- at 289 DOFs the six operations it replaces (flag reset, Jacobi scale, layout fill, lift, Krylov start, its reduction)
// cost ~4 us each, next to a 130 us launch.  The bodies are the device functions of the kernels it replaces, in their launch geometry
// (the start-up sums run workgroup by workgroup, `vec_grid` of them), so every number has the bits of the separate launches.
//   phases & 1: reset ctl, scale (from the row statistics fdapde_init left), fill, lift, u = g on Dirichlet DOFs (the launch's own epilogue
//               writes the interior entries of u: PersistArgs::u_out)
//   phases & 2: x = 0, r = p = b~, the start-up sums and scalars (with a non-zero lift the product A g~ runs between two launches of this kernel)
// ---------------------------------------------------------------------------------------------------------------------
struct SmallFrontArgs {
    int64_t n;
    int32_t phases, use_bnd, zero_y, vec_grid;
    // scale
    const double* stat;
    const uint8_t* bnd;
    double* scale;
    int32_t* ctl;
    // fill (workgroup 0 of the layout)
    int32_t nsl;
    const int64_t* ell_off;
    const int32_t* sl_off;
    const int32_t* slot_dof;
    const int32_t* src;
    const int32_t* col;
    const double* A;
    double* ell_val;
    // lift
    const double* g;
    double* gt;
    double* y;
    double* u;
    // start of the Krylov iteration
    const double* f;
    double *x, *r, *p, *r0, *partial, *sc, *seed;
    int32_t n_seed;
    double tol2;
};
static __global__ __launch_bounds__(256) void k_small_front(SmallFrontArgs a) {
    __shared__ double red[8];
    const int tid = threadIdx.x;
    if (a.phases & 1) {
        if (tid < 8) a.ctl[tid] = 0;
        __syncthreads();   // (the flag is raised below)
        for (int64_t i = tid; i < a.n; i += 256) {
            jacobi_scale_stats_row(i, a.stat, a.bnd, a.use_bnd, a.scale, a.ctl + 4);
            const bool b = a.use_bnd && a.bnd[i];
            const double gi = b ? a.g[i] : 0.0;
            a.gt[i] = gi;
            if (a.zero_y) a.y[i] = 0.0;
            if (b) a.u[i] = 0.0 + gi;   // (k_unscale on a Dirichlet DOF: scale = 0)
        }
        __syncthreads();   // scale complete (global memory, this workgroup's own stores)
        for (int q = tid >> 6; q < a.nsl; q += 4)
            (void)persist_fill_scaled_slice(0, q, tid & 63, a.nsl, a.ell_off, a.sl_off, a.slot_dof, a.src, a.col, a.A, a.scale, a.ell_val);
    }
    if (a.phases & 2) {
        __syncthreads();
        for (int b = 0; b < a.vec_grid; ++b)
            krylov_init_block(b, a.vec_grid, a.n, a.f, a.y, a.scale, a.x, a.r, a.p, a.r0, a.partial, nullptr, nullptr, a.gt, nullptr, 0, red);
        __syncthreads();   // the partials (thread 0's stores)
        krylov_init_fin_body(a.partial, a.vec_grid, a.sc, a.ctl, a.tol2, a.seed, a.n_seed, red);
    }
}

}  // namespace fdapde_hip
#endif
