// kernels.h -- hand-written HIP kernels for gfx950 (MI355X): FEM operator / forcing assembly and the Krylov solve.
//
// Everything here is HBM/L2-bandwidth-bound fp64 + int32 index work; there is no dense contraction, so no MFMA.
// What matters (cdna_hip_programming.md guidelines 2, 3, 11-13; MI355X_MICROARCH.md "Global float atomics"):
//   * index streams (adjacency, slots, colidx) and value streams are read with unit stride across the 64 lanes of a
//     wavefront; gathers (vertex coordinates, x[col]) go through the XCD's L2, which the locality numbering built in
//     host_setup.cpp keeps hot;
//   * the default assembly is atomic-free: one lane owns one matrix row, visits the cells around its DOF, accumulates
//     in LDS and the workgroup writes its rows once with coalesced stores (float atomics run at ~1.3 TB/s of added
//     bytes chip-wide and ~17x slower when the 64 lanes hit 64 different rows -- an element-wise scatter is exactly
//     that shape);
//   * reductions are deterministic: per-workgroup partials in a fixed slot, re-reduced in a fixed order by the
//     consumer kernel -- no float atomics in the solve.
// Split by subsystem:
//   kernels_assembly.h  row-owner assembly (+ scatter cross-checks), basis evaluation
//   kernels_reduce.h    deterministic reductions (wavefront, workgroup, DPP team sums)
//   kernels_spmv.h      CSR SpMV forms fused with the Krylov dot products
//   kernels_krylov.h    Jacobi scaling, CG / BiCGStab vector kernels, multi-GPU interface exchange, numbering changes
//   kernels_multirhs.h  Q right-hand sides at once against one prepared system (SpMM + batched fused-update CG)
//   kernels_persist.h   the whole CG solve of a small system as one launch (matrix resident in LDS, granule hand-offs)
#ifndef FDAPDE_KERNELS_H
#define FDAPDE_KERNELS_H

#include "kernels_assembly.h"
#include "kernels_reduce.h"
#include "kernels_spmv.h"
#include "kernels_krylov.h"
#include "kernels_gmres.h"
#include "kernels_multirhs.h"
#include "kernels_persist.h"
#include "kernels_small.h"

#endif
