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

// Uniform bin grid over the mesh for point location (fdapde_eval_pointwise: stands in for the reference's TreeSearch, geometry/tree_search.h):
// about one cell per bin on average, every cell registered in the bins its bounding box overlaps, the cells of a bin in ascending
// (internal) id -- the location kernel takes the FIRST cell of the bin that contains the point.  Built on the device from the arrays the
// assembly uses (vcoords: NP doubles per node, cverts: M + 1 internal node ids per cell): a count pass, a scan, a fill pass, a sort of
// every bin's short list.  Arrays hipMalloc'd here, ownership passes to the caller.
struct DevBinGrid {
    int32_t *bin_ptr = nullptr, *bin_cells = nullptr;   // n_bins + 1 offsets; cells
    int64_t n_bins = 0, n_entries = 0;
    double lo[3] = {0, 0, 0}, inv_h[3] = {0, 0, 0};
    int32_t dims[3] = {1, 1, 1};
};
int dev_build_bin_grid(int M, int64_t n_nodes, int64_t n_cells, const double* d_vcoords, const int32_t* d_cverts, void* stream, DevBinGrid* out,
                       std::string& err);

void dev_setup_preload();      // loads this unit's code object (hipFuncGetAttributes of one of its kernels)
}  // namespace fdapde_hip
#endif
