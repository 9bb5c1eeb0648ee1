// dev_partition.h -- the multi-GPU split of a mesh, computed ON THE DEVICE from the mesh fdapde_mesh_upload left there (dev_partition.hip):
// Morton chunks of the cell barycentres, node owners dealt in checkerboard patches, and every rank's sub-mesh (its cells + -- row-distributed
// form -- one layer of its neighbours' cells) with nodes renumbered locally in ascending global id.  No reference counterpart: fdaPDE-core is
// single-threaded and holds one mesh in one address space (fdaPDE/pde/pde.h:58-105); SURVEY 8(e) names the partition, the arithmetic is that of
// fdapde-core_amd/dist.py (rounds 1-5: numpy on one rank, 16 s at C3's size), array for array.
#ifndef FDAPDE_DEV_PARTITION_H
#define FDAPDE_DEV_PARTITION_H

#include <cstdint>
#include <string>
#include <vector>

namespace fdapde_hip {

struct RankMeshDev {   // one rank's sub-mesh, device arrays owned by the DevPartition
    int64_t n_nodes = 0, n_cells = 0;
    int32_t* l2g = nullptr;          // [n_nodes] global node id of every local node, ascending
    int32_t* cell_ids = nullptr;     // [n_cells] global cell id of every local cell, ascending
    int32_t* cells = nullptr;        // [n_cells x (M + 1)] local node ids, row-major
    double* nodes = nullptr;         // column-major n_nodes x N
    uint8_t* bnd = nullptr;          // [n_nodes] node markers of the whole mesh
    int32_t* node_owner = nullptr;   // [n_nodes] rank that owns the node
};

struct DevPartition {
    int world = 0, form = 0, M = 0, N = 0, device = -1;
    int64_t n_nodes = 0, n_cells = 0;
    int32_t* part = nullptr;          // [n_cells] rank of the element partition (Morton chunk)
    int32_t* node_owner = nullptr;    // [n_nodes] row-distributed form: lowest / highest touching rank by checkerboard box; element form: lowest
    uint64_t* node_mask = nullptr;    // [n_nodes] bit r: the node is in rank r's sub-mesh (world <= 64)
    std::vector<RankMeshDev> ranks;
};

enum { kPartitionRowdist = 0, kPartitionElements = 1 };

// d_nodes column-major n_nodes x N, d_cells row-major n_cells x (M + 1), d_bnd [n_nodes]; all on the current device.  Synchronises `stream`.
int dev_partition_build(int M, int N, int64_t n_nodes, int64_t n_cells, const double* d_nodes, const int32_t* d_cells, const uint8_t* d_bnd, int world, int form,
                        void* stream, DevPartition* out, std::string& err);
void dev_partition_release(DevPartition* p);
void dev_partition_preload();

}   // namespace fdapde_hip
#endif
