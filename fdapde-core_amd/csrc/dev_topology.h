// dev_topology.h -- device-built topology tables of a simplicial mesh (dev_topology.hip); plain C++ interface used by capi.hip.
#ifndef FDAPDE_DEV_TOPOLOGY_H
#define FDAPDE_DEV_TOPOLOGY_H

#include <cstdint>
#include <string>

#include "../../include/fdapde_hip.h"

namespace fdapde_hip {

// every pointer is device memory owned by this struct (dev_topology_release); numbering = the reference's first-seen order
struct DevTopology {
    int M = 0;
    int64_t n_cells = 0, n_facets = 0, n_edges = 0;   // facets: edges of triangles / faces of tetrahedra; 2-D: n_edges == n_facets
    int32_t* facet_nodes = nullptr;   // n_facets x M, ascending node ids        (edges_ / faces_)
    int32_t* facet_cells = nullptr;   // n_facets x 2, second = -1 on the boundary (edge_to_cells_ / face_to_cells_)
    uint8_t* facet_bnd = nullptr;     // n_facets                                 (edges_markers_ / faces_markers_)
    int32_t* cell_facets = nullptr;   // n_cells x (M+1), combinations order      (cell_to_edges_ / cell_to_faces_)
    int32_t* neighbors = nullptr;     // n_cells x (M+1), column = opposite vertex (neighbors_)
    int32_t* edge_nodes = nullptr;    // 3-D: n_edges x 2                          (edges_)
    uint8_t* edge_bnd = nullptr;      // 3-D: n_edges                              (edges_markers_)
    int32_t* face_edges = nullptr;    // 3-D: n_facets x 3                         (face_to_edges_)
    int32_t* edge_face = nullptr;     // 3-D: n_edges: the (newly seen) face that introduced the edge
};
int dev_build_topology(int M, int64_t n_nodes, int64_t n_cells, const int32_t* d_cells, const uint8_t* d_node_bnd, void* stream,
                       DevTopology* out, std::string& err);
void dev_topology_release(DevTopology* t);

// LagrangianBasis<D, 2>::enumerate_dofs on the device (basis/lagrangian_basis.h:105-133; 3-D: the natural extension):
// dofs n_cells x nb (vertex slots = the cell's nodes; edge slot of local pair (a, b) = n_nodes + edge id, slot = position of the
// edge midpoint in ReferenceElement<M,2>::nodes), dof_bnd = [node markers | edge markers], dof_coords column-major n_dofs x N with
// an edge DOF placed by the FIRST cell that visits it (J * reference node + x0, lagrangian_basis.h:159-183).  refnodes: nb x M
// reference coordinates of the local DOFs (BasisTables::refnodes).  Outputs are hipMalloc'd; the caller frees them.
int dev_build_p2_dofs(int M, int64_t n_nodes, int64_t n_cells, const double* d_nodes, const int32_t* d_cells, const uint8_t* d_node_bnd,
                      const double* refnodes, void* stream, int32_t** d_dofs, uint8_t** d_dof_bnd, double** d_dof_coords, int64_t* n_edges,
                      std::string& err);

void dev_topology_preload();   // loads this unit's code object
}  // namespace fdapde_hip
#endif
