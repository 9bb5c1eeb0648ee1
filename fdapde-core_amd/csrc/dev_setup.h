// dev_setup.h -- the discrete space's index structures built ON THE DEVICE (dev_setup.hip): what host_build_space (host_setup.cpp)
// computes with host threads -- locality numbering, row-owner adjacency, CSR patterns, slot maps, sliced-ELL adjacency, assembly
// block tables -- as radix sorts / scans (hipCUB) and small kernels, array for array identical to the host builder's output
// (FDAPDE_SETUP_CHECK=1 runs both and compares).  Plain C++ interface used by capi.hip.
#ifndef FDAPDE_DEV_SETUP_H
#define FDAPDE_DEV_SETUP_H

#include <cstdint>
#include <string>

#include "internal.h"

namespace fdapde_hip {

// device arrays of the space (hipMalloc'd here; ownership passes to the caller, who frees them with hipFree)
struct DevSpace {
    int32_t *cverts = nullptr, *cdofs = nullptr, *adj = nullptr, *rowptr = nullptr, *colidx = nullptr, *diag = nullptr, *slot_i2e = nullptr;
    int32_t *dof_i2e = nullptr, *dof_e2i = nullptr, *cell_i2e = nullptr, *node_i2e = nullptr, *bc_cell = nullptr, *bn_node = nullptr;
    int32_t *lane_row = nullptr, *rowptr_e = nullptr, *colidx_e = nullptr;
    uint32_t* slotw = nullptr;
    uint16_t* bc_vert = nullptr;
    int64_t *sl_off = nullptr, *bc_off = nullptr, *bn_off = nullptr;
    double* vcoords = nullptr;
    uint8_t* bnd = nullptr;
    int64_t n_adj = 0, n_slices = 0, n_blk = 0, n_bc = 0, n_bn = 0;   // padded visit slots; adjacency slices; assembly blocks; block-cells; block-nodes
    bool dealt = false;                                               // lane_row in use
};
void dev_space_release(DevSpace* s);

// Inputs on the device: nodes (column-major n_nodes x N), cells (row-major n_cells x (M+1)), and the DOF table in the reference's
// numbering (dofs n_cells x nb, dof_bnd, dof_coords column-major n_dofs x N).  Fills `out` and the small host-side members of `hs`
// the rest of the library reads (permutations, boundary flags in internal order, rowptr_i, sizes, rb_row); the big host arrays
// (colidx_i, cdofs_i, cverts_i, vcoords_i, colidx_e, rowptr_e) are left empty and fetched on demand by the caller.
int dev_build_space(HostSpace& hs, const double* d_nodes, const int32_t* d_cells, const int32_t* d_dofs, const uint8_t* d_dof_bnd,
                    const double* d_dof_coords, void* stream, DevSpace* out, std::string& err);

}  // namespace fdapde_hip
#endif
