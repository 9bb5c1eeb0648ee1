// context.h -- what a fdapde_ctx holds: device buffers, launch configuration and tuning knobs, the compact solver patterns, the
// multi-GPU state (RCCL communicator or host-staged transport, interface maps), the factor-once handle.  Internal to
// libfdapde_hip.so (the C ABI in include/fdapde_hip.h only sees an opaque pointer); shared by the library's translation units (engine.h).
#ifndef FDAPDE_CONTEXT_H
#define FDAPDE_CONTEXT_H

#include <cstdint>
#include <memory>
#include <functional>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/fdapde_hip.h"
#include "dev_persist.h"
#include "dev_setup.h"
#include "dev_topology.h"
#include "internal.h"

namespace fdapde_engine {
struct Group;            // eng_group.hip: the ranks of a multi-device context
}
namespace fdapde_hip {
struct DevPartition;     // dev_partition.h
struct DevTables;        // kernels_assembly.h (the context only holds device buffers of them)
struct DevRefTensors;
struct DevRefTensorsSym;
}

using namespace fdapde_hip;

#define HIPCHK(ctx, expr)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                           \
            return FDAPDE_EHIP;                                                                        \
        }                                                                                              \
    } while (0)

namespace fdapde_detail {

template <typename T> struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        if (p && n >= count && count > 0) return hipSuccess;
        release();
        n = count;
        return hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (count ? count : 1));
    }
    // fine-grained device memory (coherent with accesses of other GPUs / the host while kernels run: the boards of the row-distributed
    // persistent launches); released like any other allocation
    hipError_t alloc_fine(size_t count) {
        release();
        n = count;
        return hipExtMallocWithFlags(reinterpret_cast<void**>(&p), sizeof(T) * (count ? count : 1), hipDeviceMallocFinegrained);
    }
    hipError_t upload(const T* src, size_t count, hipStream_t st) {
        hipError_t e = alloc(count);
        if (e != hipSuccess || count == 0) return e;
        return hipMemcpyAsync(p, src, sizeof(T) * count, hipMemcpyHostToDevice, st);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr, n = 0;
    }
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { release(); }   // locals on an error return; context members are released explicitly before the context dies
};

// RCCL is loaded on first use (dlopen) so that single-GPU users and CPU-only boxes never need it
struct RcclApi {
    void* handle = nullptr;
    std::string path;   // what was loaded (diagnostics)
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // optional (diagnostics: fdapde_comm_count)
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (handle) return true;
        // ONE HIP / RCCL stack per process: RCCL is taken from the installation the HIP runtime THIS library is bound to comes from
        // (the directory of the libamdhip64 that provides hipMalloc here: /opt/rocm/lib normally; the copy bundled with a PyTorch
        // wheel when that was loaded into the process first and the loader resolved the shared soname to it), never by bare soname
        // first -- a bare "librccl.so.1" returns whichever RCCL some other module of the process happened to load
        std::vector<std::string> names;
        Dl_info di;
        if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t cut = dir.rfind('/');
            if (cut != std::string::npos) {
                dir.resize(cut + 1);
                names.push_back(dir + "librccl.so.1"), names.push_back(dir + "librccl.so");
            }
        }
        for (const char* nme : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) names.push_back(nme);
        for (const std::string& name : names) {
            handle = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (handle) {
                path = name;
                break;
            }
        }
        if (!handle) {
            err = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(handle, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(handle, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(handle, "ncclCommCount"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(handle, "ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(handle, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(handle, "ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString || !Send || !Recv || !GroupStart || !GroupEnd) {
            err = "librccl lacks a required symbol";
            return false;
        }
        return true;
    }
};
inline RcclApi g_rccl;

struct HostTerm {
    fdapde_term t;
    std::vector<double> data_i;              // space-varying data permuted to internal cell order (contexts whose space is not on a device yet)
    std::shared_ptr<DBuf<double>> data_dev;  // ... or already on the device, permuted there (check_terms)
    bool field_nonsym = false;               // a diffusion FIELD with a row that is not a symmetric tensor (check_terms)
};

}  // namespace fdapde_detail
using namespace fdapde_detail;

// fdapde_options.time_spmv samples every kTimeStride-th Krylov iteration (the first iterations after init run on cold caches
// and are not representative of the solve: 45.7 us against a kernel-trace average of 43.9 us on C3)
constexpr int64_t kNtValsRows = 2000000;   // above this many rows the SpMV of teams <= 8 hints its value stream (see load_pair)
constexpr int kTimeStride = 8, kTimePhase = 3;   // phase 3: never the first launch after a host poll (the GPU has just idled)

// everything the captured launch sequence of a CG chunk depends on (the graph is rebuilt when any of it changes)
struct GraphKey {
    const void* sval;
    const void* rowptr;
    int64_t n;
    double tol2;
    int chunk, v, grid, team, ablate, c16, deep, unroll, sp_cur, bk_cur, bk_G;
    const void* bk_val;
};

// what solve_prepare decided for a system matrix (shared by the elliptic, parabolic and handle solves)
struct SolveState {
    bool dist = false, diag_positive = true;
    bool symmetric = true;  // what solve_prepare was told about the matrix (row-distributed form: CG or BiCGStab launch)
    bool rowdist = false;   // row-distributed multi-GPU form: complete rows of the owned DOFs, one persistent launch per rank
    const uint8_t* owned = nullptr;
    int use_bnd = 0;
    bool diag_deferred = false;   // small one-GPU systems: "every interior diagonal is positive" was ASSUMED (flag at ctl[4], read back with the
                                  // solve's outcome instead of behind a wait of its own); the caller repeats the solve the slow way if it was wrong
    mutable bool front_pending = false;   // ... and of at most `small_front_rows` rows in ONE workgroup: flag reset, scale, fill, lift and the Krylov start-up
    const double* front_A = nullptr;      // have NOT been enqueued by solve_prepare -- the first solve_run enqueues them as one launch (k_small_front)
};
struct SolveStateHolder {
    SolveState ss;
};

struct fdapde_ctx {
    int device = -1;
    bool has_device = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_p0 = nullptr, ev_p1 = nullptr;   // (ev_p*: around the persistent launch)
    std::vector<hipEvent_t> ev_spmv;   // 2 per timed SpMV launch
    std::string err;
    HostSpace hs;
    BasisTables tb;
    bool space_ready = false, dev_ready = false, colour_ready = false;
    bool assembled[2] = {false, false};
    bool force_ready = false, solved = false, dirichlet_applied = false;
    // problem data (host copies)
    std::vector<HostTerm> op;
    bool op_symmetric = true;
    std::vector<double> fq_i;   // internal cell order, column-major rows x ncols
    int fq_cols = 0;
    std::vector<double> g_i;    // internal DOF order
    bool have_g = false, g_zero = false;   // Dirichlet data set; all of it zero (homogeneous: the lift A g~ vanishes)
    fdapde_info info{};
    // device buffers
    DBuf<int32_t> cverts, cdofs, adj, rowptr, colidx, diag, slot_i2e, dof_i2e, dof_e2i, cell_i2e, rb_row, colour_cells;
    DBuf<int32_t> rowptr_e, colidx_e;   // reference-numbering pattern (device-built spaces: fetched by fdapde_pattern_get on demand)
    DBuf<int32_t> dofs_e;               // order 2, device-built: the DOF table in the reference numbering (host mirror on demand)
    DBuf<double> coords_e;              // ... and the DOF coordinates
    // the mesh as handed over, resident on the device from fdapde_mesh_upload on (nodes column-major, cells row-major, node markers 0 / 1):
    // fdapde_dofs_build and fdapde_topology_build start from these; 41 + 162 MB at C3's size
    DBuf<double> mesh_nodes;
    DBuf<int32_t> mesh_cells;
    DBuf<uint8_t> mesh_nbnd;
    bool mesh_on_dev = false;
    DBuf<double> gm_V, gm_b, gm_s, gm_part;   // restarted GMRES (kernels_gmres.h): basis (m + 1) x n, scaled right-hand side, small state, partial sums
    int persist_exp_lds = 1;                  // knob: the symmetric streaming launch keeps its export list in LDS where there is room
    int gmres_m = 50;                         // restart length (knob gmres_m)
    int auto_gmres = 1;                       // FDAPDE_SOLVER_AUTO ends in GMRES after BiCGStab gave up (knob auto_gmres)
    bool dev_built = false;             // the index structures were built on the device (dev_setup.hip); big host mirrors are lazy
    DBuf<uint32_t> slotw;
    DBuf<int64_t> bc_off, bn_off;
    DBuf<int32_t> bc_cell, bn_node;
    DBuf<uint16_t> bc_vert;
    DBuf<int64_t> sl_off;
    DBuf<int32_t> lane_row;   // assembly lane position -> row (unallocated = identity)
    DBuf<double> stiff_stat;   // (diagonal, row maximum) of every row of vals[STIFF], written by fdapde_init's sweep; valid while stiff_stat_valid
    bool stiff_stat_valid = false, asm_all_in_lds = false;
    int asm_row_stat = 1;      // knob: 0 = the solve's Jacobi scaling always reads the whole matrix
    DBuf<double> fq_blk;      // column 0 of the forcing as one load coefficient per visit slot (k_visit_load_coeffs); valid while fq_blk_ready
    bool fq_blk_ready = false;
    DBuf<double> fq_bc;       // column 0 of the forcing samples in BLOCK-CELL order (a cell's nq samples repeated in every assembly block that
                              // visits it): the row-owner sweep reads them where it reads the block's cells; built by fdapde_set_forcing
    bool fq_bc_ready = false;
    bool cg_broke_down = false;  // the assembled stiff_ is symmetric but CG broke down on it (fdapde_solve with the method left open): the next solves go to BiCGStab at once
    int bicg_shadow = 0;         // knob: shadow residual of the multi-launch BiCGStab: 0 = r0 (BiCGStab as published), 1 = pseudo-random, 2 = r0 with randomly scaled entries
    int bicg_restart = 1;        // knob: 0 = a BiCGStab breakdown ends the solve (FDAPDE_ENOCONV) instead of restarting it from the iterate reached
    int asm_split_varying = 1;   // knob: 0 = every space-varying operator takes the per-node tensor integrand (element_row OPK 4), also where only advection / reaction vary (OPK 5)
    int asm_items = 1;        // knob: 0 = spaces with dealt rows (P2) keep the row-walking sweep instead of the visit-parallel one (k_assemble_items)
    int32_t asm_max_visits = -1;   // longest visit list of the space (computed on first use; -1 = not yet)
    int asm_fuse_mass = 1;    // knob: fdapde_init accumulates the mass matrix in the operator's sweep where both accumulator ranges fit the LDS
    int asm_items_fuse = 1;   // knob: the visit-parallel sweep (P2) accumulates operator AND mass matrix in one sweep where both accumulator ranges fit the CU
    int asm_fq_bc = 1;        // knob: keep the block-cell ordered copy (1) or let the sweep gather from the cell-ordered samples (0)
    int asm_fq_block = 0;     // tuning knob: 1 = fdapde_init first reduces the forcing samples to one load coefficient per visit slot (k_visit_load_coeffs);
                              // measured on C3 with that kernel inside init's timed region: init 1.49-1.53 ms against 1.23-1.26 ms for the sweep
                              // gathering the cell's samples itself (a cell's nq samples share one 32-byte sector with the coefficient they would become)
    DBuf<double> vcoords, vals[2], force, fq, g, sval, scale, gt, x, r, p, y, s, t, r0, u, part_a, part_b, sc, tmp_e,
      tmp_i, tmp_v, restart_u, lin_rhs;   // (restart_u: iterate a broken-down BiCGStab is started again from; lin_rhs: right-hand side of fdapde_lin_solve in internal order; kept between calls -- hipMalloc + hipFree per
                               //  call were a third of a small solve)
    DBuf<uint8_t> bnd;
    DBuf<DevTables> tables;
    DBuf<DevRefTensors> reftab;
    DBuf<DevRefTensorsSym> reftab_sym;   // the compact tensors for operators with a symmetric Kt
    DBuf<int32_t> ctl;
    DBuf<double> coef[kMaxTerms];
    bool coef_of_op = false;   // coef[] hold the space-varying data of the operator set by fdapde_set_operator (uploaded by fdapde_init)
    double* h_io = nullptr;     // pinned, device-mapped: [b | x | outcome record] of a direct small solve (run_persist_direct)
    size_t h_io_cap = 0;
    int persist_direct = 1;            // knob: 0 = single right-hand sides of one-workgroup systems take the general path
    int persist_direct_spin_us = 2000; // how long the host spins on the record's status word before it waits for the stream (0: never spins)
    int32_t* h_ctl = nullptr;   // pinned: ctl[3]
    double* h_sc = nullptr;     // pinned: sc[0..3]
    int spmv_grid = 0, rb_per_band = 0, vec_grid = 0, n_rb = 0, cg_grid = 0;
    int spmv_variant = 2;   // 2: team form, 2 entries per lane (default); 0: team form, 1 entry per lane
                            // (FDAPDE_SPMV=team); 1: stream form (FDAPDE_SPMV=stream) -- kept for A/B measurements
    int spmv_team = 16, spmv_unroll = 4, spmv_ablate = 0;
    int lds_limit = 96 * 1024;   // per assembly workgroup: tables + staged vertices + row accumulators
    // compact solver pattern (no diagonal; [1]: also no Dirichlet rows / columns), built on first use
    DBuf<int32_t> sp_rowptr[2], sp_colidx[2], sp_map[2], sp_tbase[2], sp_vrow[2];
    int64_t sp_nv[2] = {0, 0};               // > 0: the compact pattern is segmented into this many virtual rows (sp_vrow)
    int sval_layout = -2;                    // layout sval was last zero-filled for (pad entries of a segmented pattern stay 0)
    DBuf<uint16_t> sp_col16[2];              // 16-bit column codes of the compact pattern (host_build_col16)
    int64_t sp_wide[2] = {0, 0};             // groups of 32 rows that fall back to the 32-bit columns
    int spmv_ntv = -1;                       // tuning knob: -1 auto (by size), 0 / 1 force the value-stream policy of teams <= 8
    int sp_team = 0;                         // team size the segmented patterns were built for
    int spmv_c16 = 1;                        // tuning knob: 0 = always stream the 32-bit columns
    int use_graph = 0;                       // tuning knob: replay full chunks of the fused-update CG as one hipGraph
    hipGraphExec_t cg_graph_exec = nullptr;
    GraphKey cg_graph_key{};
    int cgf_lazy = 1;                        // tuning knob: x updated every second launch of k_cgf_update (C3 solve 33.3 -> 32.5 ms, same iterations)
    int cgf_nt = 7;                          // tuning knob, bit set: nontemporal y (1), x (2), r (4), p load (8) in k_cgf_update
    int cgf_band = 1;                        // tuning knob: XCD-aware mapping + nontemporal x / r / y in k_cgf_update (C3 solve 41.80 -> 41.10 ms)
    int cgf_split = 0;                       // k_cgf_update requests the second half of its elements after the scalars (diagnostic)
    int cgf_v = 8;                           // double2 elements per lane of k_cgf_update (1, 2, 4, 8); C3 solve: 47.2 / 41.8 / 41.3 / 40.9 ms
    int spmv_deep = 0;                       // tuning knob: 1 = k_spmv_c16p (gathers one tile ahead; measured slower: 3 waves / SIMD)
    int64_t sp_nnz[2] = {0, 0};
    bool sp_built[2] = {false, false};
    int sp_cur = -1;   // which compact pattern c->sval currently holds (-1: full pattern)
    bool sval_stale = false;        // the single-launch solver filled its blocks straight from the unscaled matrix: c->sval was NOT written for the
    const double* sval_A = nullptr; // current system; whoever needs it (warm start, multi-launch fall-back, SpMV benchmark) calls ensure_sval first
    int persist_fill_fused = 0;     // knob: 1 = fill the launch's blocks straight from the unscaled matrix (k_persist_fill_scaled) and skip the scaled
                                    // full-pattern copy.  Measured on C3: solve 15.63 -> 15.82 ms with the column looked up through the pattern
                                    // (three dependent gathers), 15.56 -> 15.53 ms with a precomputed column per entry and four pair rows per step:
                                    // no gain worth 58 MB more of layout: off
    // whose system the shared scale / sval / sp_cur buffers hold (every solve_prepare caller records itself; the factor-once
    // handle prepares again whenever anybody else has been there in between)
    enum { kScaledNone = 0, kScaledSolve, kScaledParabolic, kScaledLin, kScaledPmg };
    int scaled_owner = kScaledNone;
    // multi-GPU (element partition): RCCL communicator + interface maps
    ncclComm_t comm = nullptr;
    fdapde_allreduce_fn ar_fn = nullptr;     // host-staged transport (tests / non-RCCL fabrics) instead of the RCCL communicator
    void* ar_user = nullptr;
    std::vector<double> ar_host;
    DBuf<double> ar_dev;                     // staging of fdapde_comm_allreduce
    int world = 1, rank = 0;
    bool halo_ready = false;
    int64_t n_if = 0, n_loc_if = 0;          // global / local interface DOF counts
    DBuf<int32_t> halo_dof, halo_pos;        // local interface DOF (internal id) -> slot in the global interface vector
    DBuf<int32_t> halo_inv, if_slot;         // global slot -> local DOF or -1 [n_if]; local DOF -> global slot or -1 [n_dofs]
    DBuf<uint8_t> owned;                     // internal DOF order: 1 = this rank counts the DOF in global dot products
    DBuf<double> hbuf, sbuf;                 // [n_if + 2] packed interface values + fused dot partials; [4] scalars
    // neighbour-only exchange (fdapde_halo_setup_peers): in this mode n_if = n_loc_if, hbuf[k] = summed value of local interface DOF k
    bool peer_mode = false;
    fdapde_exchange_fn xchg_fn = nullptr;    // host-staged transport of the neighbour exchange (tests / non-RCCL fabrics)
    void* xchg_user = nullptr;
    std::vector<int32_t> peer_rank;          // ranks this one shares DOFs with, ascending
    std::vector<int64_t> peer_off;           // [n_peers + 1] segment of each peer in the send / receive buffers
    DBuf<int32_t> peer_send_dof;             // [n_send] internal DOF id of every send-buffer entry
    DBuf<int32_t> peer_src_off, peer_src;    // per local interface DOF: its contributions in ascending rank order (-1 = this rank's own,
                                             // else index into the receive buffer)
    DBuf<double> peer_sendbuf, peer_recvbuf;
    std::vector<double> xchg_send_h, xchg_recv_h;
    // IN-PROCESS direct transport (the ranks of a multi-device context, eng_group.hip): every rank's buffers are addressable from every other rank's
    // kernels (one process: the device pointers themselves, peer access between devices), so an exchange is "pack, drain the stream, all ranks arrive,
    // fetch from the peers' buffers" -- no staging through host memory.  Send buffers and slots are double-buffered by call parity: a rank overwrites a
    // buffer two calls after its peers read it, and they finished that read before their own next arrival.
    struct PeerDirect {
        bool on = false;
        int (*arrive)(void*) = nullptr;   // every rank of the process arrives; 0 = all there
        void* user = nullptr;
        int parity = 0, ar_parity = 0;
        int64_t n_send = 0;
        DBuf<const double*> remote[2];    // per receive-buffer entry: where the peer's value lies (parity buffer of ITS send buffer)
        DBuf<double> slots;               // this rank's published small vectors: [parity][kind: 0 halo scalars, 1 all-reduce payload][kSlotDoubles]
        DBuf<const double*> slot_ptr;     // [world] the slots of every rank
    } xd;
    static constexpr int kSlotDoubles = 64;
    // "factor once, solve many" handle (fdapde::SparseLU wrapper, utils/symbols.h:133-160)
    DBuf<double> lin_mat;                    // the matrix handed to fdapde_lin_compute, internal slots
    bool lin_ready = false, lin_symmetric = false;
    DBuf<double> lin_sq;                     // full-pattern Jacobi-scaled copy of the handle's matrix (multi-RHS SpMM)
    bool lin_sq_ready = false;
    int multi_rhs = 1;                       // tuning knob: 0 = always solve the columns of fdapde_lin_solve one by one
    SolveStateHolder* lin_state = nullptr;
    // persistent small-problem CG (kernels_persist.h): resident layouts for the two boundary variants, built on first use
    int n_cu = 0;
    int persist = 1;                         // tuning knob: 0 = never take the single-launch path
    int persist_time = 1;                    // workgroup 0 stamps the phases of every iteration (a handful of s_memrealtime per iteration)
    int persist_balance = 1;                                // workgroup boundaries of the persistent CG at equal cost (entries + 2 per row)
    int persist_sym = 2;                                    // symmetric storage of the persistent CG: 0 never, 1 always, 2 where the plain blocks would stream
    int persist_gather_waves = 4, persist_poll_sleep = 2;   // tuning knobs of the dot all-gather (kernels_persist.h)
    bool persist_broken = false;             // a hand-off timed out (workgroups not co-resident): on the multi-launch path for persist_retry_in more systems
    int persist_retry_in = 0, persist_backoff = 8;
    struct Persist {
        bool tried = false, ok = false;
        PersistLayout meta;                  // sizes only (the big arrays are released after the upload)
        int32_t lds_cap = 0, imp_cap = 0;
        size_t lds_bytes = 0;
        DBuf<int32_t> slot_dof, sl_off, ell_src, exp_off, imp_off, imp_pos;
        DBuf<int32_t> ell_col;               // column DOF of every entry (fill_persist_scaled; built on first use)
        bool stream = false;                 // the blocks do not fit the LDS: streaming instantiation
        bool exp_lds = false;                // symmetric streaming form: the export list rides in LDS (lds_bytes includes it)
        DBuf<int64_t> ell_off;
        DBuf<uint16_t> ell_code, exp_slot;
        DBuf<double> ell_val;
        DBuf<unsigned long long> amax;       // symmetric storage: bit pattern of max |ell_val| (k_persist_fill)
        DBuf<uint8_t> wg_late;               // BiCGStab layouts built with late-import workgroups (host_build_persist_layout allow_late), else empty
        DBuf<unsigned long long> board;      // [2 n_board granules of p | 2 x G x 8 granules of dot partials]; never cleared between launches (epoch tags)
        DBuf<unsigned long long> board_cols; // one such board per column of a launch that solves several right-hand sides (run_persist_cols)
        int per_cu = 0;                      // workgroups of the instantiation attr_set a CU holds (occupancy API)
        bool filled = false;                 // ell_val holds the currently scaled system
        uint32_t epoch_next = 0;             // the next launch tags its granules epoch_next + iteration + 1
        const void* attr_set = nullptr;      // kernel instantiation whose dynamic-LDS attribute is in place
        bool built_plain = false;            // built for a non-symmetric system (plain storage whatever persist_sym says)
    } ps[2];
    // row-distributed multi-GPU form of the persistent CG (fdapde_rowdist_setup; persist_engine.hip): every rank owns a set of DOFs and
    // assembles their rows completely (its sub-mesh holds every cell touching an owned DOF); the workgroups of all ranks' launches act
    // as one grid, exchanging through peer-mapped boards
    struct RowDist {
        bool ready = false;
        std::vector<int32_t> owner_i;        // internal DOF order: owning rank
        std::vector<int64_t> key_i;          // ... global key
        DBuf<uint8_t> owned;                 // 1 = this rank's DOF
        int max_wg = 0;                      // workgroups this rank's launch may use (0: one per CU; tests put several ranks on one device)
        int flat_gather = -1;                // dot gather across ranks: -1 auto (one hop while G_tot <= 1024, else two levels), 0 / 1 forced
        int timeout_first_ms = 2000;         // bound of the waits of iteration 0 (launch skew between the ranks)
        struct Layout {
            bool tried = false, ok = false;
            Persist ps;                      // the local part: blocks, local exports / imports, board (fine-grained), epochs
            int32_t n_ghost = 0, G_tot = 0, g_base = 0;
            DBuf<int32_t> rexp_off, rexp_peer, rexp_pos;
            DBuf<uint16_t> rexp_slot;
            DBuf<uint8_t> wg_late;           // workgroups that fetch their imports before their first pass
            DBuf<unsigned long long> rboard;  // what crosses ranks: [entries imported from other ranks | one dot record per rank x 2], fine-grained, IPC-mapped
            DBuf<unsigned long long*> peer_pboard, peer_dboard;
            std::vector<void*> ipc_opened;   // peer boards mapped through hipIpcOpenMemHandle (closed with the context)
            // exchange of per-DOF values of the ghost columns (the Jacobi scale, once per prepared system): per peer, what goes out / comes in
            std::vector<int32_t> x_rank;     // peers, ascending
            std::vector<int64_t> x_soff, x_roff;
            DBuf<int32_t> x_send_dof, x_recv_dof;
            DBuf<double> x_sendbuf, x_recvbuf;
        } lay[2];
    } rd;
    DBuf<double> persist_stats;
    DBuf<double> persist_x;                  // the persistent launch writes its solution here (x stays the initial guess)
    DBuf<double> persist_xs;                 // wide form (24 rows per thread): x of every slot, in slot order, between the iterations of a launch
    int persist_wide_gj = 12;                // knob (measurements): 4 / 6 / 12 passes of a phase of the wide form load together (2.35 M rows: 73.0 / 73.4 / 71.8 us per iteration)
    int persist_wide = 1;                    // knob: 0 = systems of more than 8 192 rows per workgroup keep the multi-launch path
    int persist_max_wg = 0;                  // knob (tests): the single-launch layouts use at most this many workgroups (0: one per CU)
    double persist_launch_ms = 0;            // duration of the last persistent launch (HIP events on the stream)
    int persist_host_below = 32768;          // systems of at most this many DOFs build the persistent layout on the host (first-solve latency)
    int persist_cols = 1;                    // knob: several columns of fdapde_lin_solve as ONE persistent launch where G x columns workgroups are resident
    DBuf<double> cols_b, cols_r, cols_x, cols_sc, cols_part;   // their staging: right-hand sides as handed over, scaled, solutions, scalars, partial sums
    DBuf<int32_t> cols_ctl;
    struct EvalGrid {   // bin grid for point location (fdapde_eval_pointwise), built on the device once per mesh
        bool ready = false;
        DBuf<int32_t> ptr, cells, dims;
        DBuf<double> lo, invh;
        void release() { ptr.release(), cells.release(), dims.release(), lo.release(), invh.release(), ready = false; }
    } eval_grid;
    DBuf<double> eval_locs, eval_vals;   // locations / basis values of a call (kept between calls)
    DBuf<int32_t> eval_out;
    std::function<int()> persist_tail;      // what run_persist enqueues behind the launch and its read-backs, before it waits (solve_run's epilogue)
    int h_ctl_seen = 4;                      // how many words of ctl the last outcome read-back fetched into h_ctl
    bool ev1_at_end = false;                 // fdapde_solve: solve_run records ev1 right before its final wait (no event wait of the caller's own)
    int64_t small_rows = 8192;               // systems of up to that many DOFs take the wait-free tail (knob small_rows; 0: off)
    int64_t small_front_rows = 2048;         // knob: fdapde_solve of a one-workgroup system of at most this many DOFs enqueues ONE kernel in front of the
                                             // single launch (k_small_front) and none behind it (PersistArgs::u_out); 0 = the separate launches
    bool front_used = false;                 // (this solve's front ran as k_small_front: the Dirichlet entries of u are written)
    bool tail_in_launch = false;             // (run_persist -> persist_tail: the launch wrote u itself)
    bool defer_end_sync = false;             // set by callers that loop over solves (parabolic steps, handle columns): solve_run does not wait for its
                                             // last kernel (the outcome is read behind a wait of its own; the rest is ordered by the stream)
    int persist_single_rows = 2048;          // knob: systems of up to that many interior rows run as ONE workgroup (no hand-off in the iteration)
    int persist_prefetch = 1;                // knob: entry steps of the next operator application touched during the dot all-gather (streaming forms)
    int persist_late = 0;                    // knob: CG layouts with late-import workgroups (host builder) instead of doubled rows per thread
    int persist_plain = 0;                   // the system being prepared is non-symmetric: plain storage, BiCGStab kernel
    int persist_bicg = 1;                    // tuning knob: 0 = non-symmetric systems always take the multi-launch BiCGStab
    int persist_coop = 0;                    // 1: cooperative launch (the runtime refuses a grid that cannot be resident).  Off by default: the grid is
                                             // sized from hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs already, the cooperative queue costs 10.5 ms
                                             // the first time a process uses it, and launches of several processes on one device would serialise
    int persist_timeout_us = 5000;           // bound of every in-kernel wait
    int persist_debug_stall = 0;             // tests: iteration at which the last workgroup plays a non-resident peer
    // blocked-ELL SpMV (k_spmv_blocked) of the multi-launch Krylov kernels on one GPU: layouts for the two boundary variants
    struct Blocked {
        bool tried = false, ok = false, filled = false;
        PersistLayout meta;
        int32_t imp_cap = 0;
        size_t lds_bytes = 0;
        DBuf<int32_t> slot_dof, sl_off, ell_src, imp_off, imp_dof, drop_dof;
        DBuf<int64_t> ell_off;
        DBuf<uint16_t> ell_code;
        DBuf<double> ell_val;
    } bk[2];
    int bk_cur = -1;      // the scaled system of the current solve lives in bk[bk_cur] (launch_spmv routes products with c->sval there)
    int blocked = 1;      // tuning knob: 0 = the CSR kernel (k_spmv_team2) on the compact pattern instead
    DevTopology topo;   // Triangulation topology tables, built on the device by fdapde_topology_build
    bool topo_ready = false;
    // element-wise scatter forms of the assembly (built on first use)
    bool part_ready = false, wave_ready = false;
    int32_t part_colours = 0;
    int64_t n_parts = 0;
    DBuf<int32_t> part_cells, part_off, part_slots, wave_slots;
    DBuf<uint8_t> part_shared;
    std::vector<double> persist_host_stats;
    // the two-level solver of order-2 spaces (eng_pmg.hip): the P1 space of the same mesh as a context of its own + the transfer operators
    struct Pmg {
        fdapde_ctx* coarse = nullptr;
        bool ready = false;
        int64_t init_seen = -1;          // the fine context's init_count the coarse operator was assembled for ...
        double extra_seen = 0.0;         // ... and the multiple of the mass matrix added to it (the stepper's 1 / dt)
        DBuf<int32_t> pa, pb;            // fine DOF -> its coarse DOF(s) (internal numberings; pb = -1: a vertex DOF)
        DBuf<int32_t> fine_cell;         // coarse internal cell -> fine internal cell
        DBuf<int32_t> rt_ptr, rt_idx;    // P^T as CSR over the coarse DOFs
        DBuf<double> rt_w, dinv, vec, part, dots, basis;   // (basis: V_0 .. V_mk, Z_0 .. Z_{mk-1} of the flexible GMRES)
        int np = 1;
        double setup_ms = 0;
        int last_coarse_iters = 0, last_coarse_calls = 0;
        DBuf<int32_t> flag;              // raised by k_pmg_diag_inv: an interior diagonal entry is 0 (no unit-diagonal form A D^-1)
        const double* fine_A = nullptr;  // what the blocked-ELL layout's values were filled from (A D^-1; valid while scaled_owner == kScaledPmg) ...
        int64_t fine_key = -1;           // ... the fine context's init_count key of that fill ...
        int fine_bnd = -1;               // ... and its boundary variant
        double omega = 0.0;              // the cycle's Jacobi damping, 1.5 / lambda_max(D^-1 A) of the matrix named by the four below
        const double* omega_A = nullptr;
        int64_t omega_key = -1;
        int omega_bnd = -1;
        double omega_extra = 0.0;
        SolveState coarse_ss;            // the coarse context's solver, prepared ONCE per coarse operator (coarse_prepare / coarse_solve, eng_solve.hip)
    } pmg;
    int64_t init_count = 0;       // the assembled stiffness matrix's EPOCH: fdapde_init calls that may have changed its values (who caches something derived from them compares).
                                  // A repeated fdapde_init of the SAME operator with the row-owner sweep reproduces the matrix bit for bit (only the load vector is new): same epoch
    bool matrix_dirty = true;     // the operator (or the space) changed since the last fdapde_init, or fdapde_assemble rewrote the stiffness matrix
    bool last_init_rows = false;  // ... and that fdapde_init used the bitwise-reproducible row-owner sweep
    double pmg_inner_rtol = 1e-1; // knob pmg_inner_tol_exp: the coarse solves stop at 10^-exp (flexible GMRES outside: 1e-1 costs no outer iteration over 1e-2)
    int pmg_inner_maxit = 200;    // knob: ... or after that many iterations.  3-D never gets there (C5: 35 per solve); a 2-D P1 level of 490 k DOFs would take 230 - 650 per solve to
                                  // 1e-1, and the flexible GMRES is better served by more, weaker corrections: 2 M DOFs, -Lap 76 -> 41 ms, -Lap + b.grad + 1 67 -> 53 ms (budget 100: 56 / 46)
    int pmg_auto = 1;             // knob: 1 = the open method takes the two-level solver for large order-2 systems it is eligible for ...
    int pmg_outer = 0;            // knob: the outer method of the two-level solver: 0 = flexible GMRES, 1 = BiCGStab (round 6's first form)
    int pmg_setup_check = 0;      // knob: 1 = the coarse level's transfer tables are also built by the host loops of the first version and compared
    int pmg_restart = 50;         // knob: vectors per cycle of the flexible GMRES (2 .. 50; the basis is also held under ~16 GB)
    int pmg_smooth = 1;           // knob: 1 = the preconditioner of the flexible GMRES is a V(1,1) cycle (damped Jacobi around the coarse correction), 0 = the additive form
    int64_t pmg_auto_first_rows = 1000000;   // knob: ... and at once -- on the context's FIRST open-method solve -- only from that many DOFs on: the coarse level's set-up (the first
                                             // P1 space of a process: ~100 ms at 389 k DOFs, 0.3 s at 5.4 M) is rent-or-buy like the dense inverse's: a caller that solves once a
                                             // system of 389 k DOFs is served faster by the Jacobi stages (24 against 120 ms), the second solve builds the level (12 against 21 ms from then on)
    int64_t open_solves = 0;      // open-method solves (fdapde_solve / steps of fdapde_solve_parabolic) this context has finished
    int pmg_blocked = 1;          // knob: 1 = the fine operator of the two-level solver through the blocked-ELL SpMV (0: the CSR kernel on the raw matrix)
    int64_t pmg_auto_rows = 300000;    // knob: ... of at least that many DOFs (where it starts to win: 3-D 185 k 9.3 against 8.6 ms, 389 k 13 against 21; 2-D symmetric 315 k 27 against 25, 642 k 38 against 59)
    // the dense inverse of a small system (kernels_dense.h / eng_dense.hip): the factor-once handle's, the parabolic stepper's, the open method's direct stage
    struct Dense {
        DBuf<double> X;             // n x n, internal DOF order
        int64_t n = 0;
        int use_bnd = 0;            // the Dirichlet rows were replaced by unit rows (fem_solver_base.h:142-155)
        const double* A = nullptr;  // the matrix values it inverts (device, internal slots): one step of refinement reads them
        bool ready = false, refine = false, failed = false;
        double check = 0, build_ms = 0;   // max |I - A X|; what the build cost (host wall clock)
    } lin_dense, step_dense, solve_dense;
    int dense_bulk = 1;           // knob: 0 = many columns staged by the kernels reading / writing the pinned block themselves (as single columns are) instead of by DMA
    int dense_fold = 1;           // knob: 1 = the parabolic stepper's dense loop as ONE product per step (u' = B u + c, B = K^-1 M / dt); 0 = M u, rhs, K^-1 rhs, hand-over (four launches)
    int dense_hostb = 0;          // knob: 1 = one column of a system of up to 512 rows: the product reads b from the pinned block itself (k_dense_gemv_hostb), no k_dense_stage in front.
                                  // Measured: 33 us per column against 22 at 289 and 484 rows -- a workgroup's read of host memory costs ~10 us, twice the launch it saves: off
    int dense_direct = 0;         // knob: 1 = a single column's product hands the result over itself (k_dense_gemv_direct): up to 512 rows the whole solve is that ONE launch
                                  // (b permuted by the host into the pinned block, read by every workgroup), above that two launches; 0 = stage -> product -> out.
                                  // Measured (us per column, 289 / 1 089 / 4 225 rows): 30 / 43 - 140 / 83 - 159 against 21 / 27 / 57 -- every wavefront's hand-over to the
                                  // host (write-through stores drained, or a system-scope fence that also empties the L2 of X) costs more than the launches it saves: off
    int dense_multi = 1;          // knob: 0 = above 4 096 rows ONE panel workgroup with a panel of 4 columns instead of several with 16
    int dense_block = 1;          // knob: 0 = the inversion pivot by pivot (k_dense_invert) instead of in panels (k_dense_invert_blocked)
    int dense_rows = 8192;        // knob: systems of up to that many DOFs may take the dense path (0: never)
    int dense_after = 2;          // knob: ... once a handle's matrix has been asked for more than that many columns / a stepper for that many steps
                                  // AND the Krylov time spent (handle) / to be expected (stepper) reaches half of what the inversion costs; 0: at once
    int64_t lin_cols = 0;         // columns solved against the handle's current matrix
    double lin_krylov_ms = 0;     // ... and the host time the Krylov columns among them took
    DBuf<double> dn_b, dn_x, dn_r, dn_e;   // (dn_e: many columns in the reference numbering on the device, in and out)
    DBuf<unsigned int> dn_cnt;
    // multi-device context (fdapde_ctx_create_multi): this context is the ROOT -- whole mesh, whole function space, every index getter -- of a group
    // of rank contexts, one per device (eng_group.hip)
    fdapde_engine::Group* group = nullptr;
    // the split of the resident mesh computed by fdapde_partition_build (dev_partition.hip)
    fdapde_hip::DevPartition* partition = nullptr;
    double partition_ms = 0;
    std::vector<uint64_t> partition_mask_h;
};

#endif
