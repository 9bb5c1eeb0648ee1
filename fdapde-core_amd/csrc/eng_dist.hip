// eng_dist.hip -- multi-GPU transports and set-up: the RCCL communicator (or host-staged callbacks), device-buffer all-reduce, interface
// ("halo") sums of the element-partitioned solve in both exchange forms, ownership / key hand-over of the row-distributed form.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include "context.h"
#include "engine.h"
#include "kernels.h"

namespace fdapde_engine {

int allreduce_sum(fdapde_ctx* c, double* buf, size_t count) {
    if (c->xd.on && count <= (size_t)fdapde_ctx::kSlotDoubles) {   // in-process: publish in this rank's slot, all arrive, sum the slots in rank order
        const int par = c->xd.ar_parity;
        c->xd.ar_parity ^= 1;
        const int64_t off = ((int64_t)par * 2 + 1) * fdapde_ctx::kSlotDoubles;
        HIPCHK(c, hipMemcpyAsync(c->xd.slots.p + off, buf, sizeof(double) * count, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->xd.arrive(c->xd.user) != 0) return fail(c, FDAPDE_ERCCL, "in-process transport: a rank did not arrive");
        hipLaunchKernelGGL(k_slot_sum, dim3(1), dim3(64), 0, c->stream, c->world, (int)count, c->xd.slot_ptr.p, off, buf);
        HIPCHK(c, hipGetLastError());
        return FDAPDE_OK;
    }
    if (c->ar_fn) {   // host-staged: device -> host, caller-provided sum over ranks, host -> device
        c->ar_host.resize(count);
        HIPCHK(c, hipMemcpyAsync(c->ar_host.data(), buf, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->ar_fn(c->ar_user, c->ar_host.data(), (int64_t)count) != 0) return fail(c, FDAPDE_ERCCL, "all-reduce callback failed");
        HIPCHK(c, hipMemcpyAsync(buf, c->ar_host.data(), sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
        return FDAPDE_OK;
    }
    RCCLCHK(c, g_rccl.AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, c->stream));
    return FDAPDE_OK;
}

// v (internal DOF order, sub-assembled) -> interface entries summed over the ranks sharing them; optionally carries the
// two fused dot partials of the SpMV (part_a, stride 2) through the same all-reduce: they land in hbuf[n_if], [n_if + 1]
// neighbour-only form (fdapde_halo_setup_peers): pack the per-peer segments, one grouped RCCL call with a send + a receive per peer, the
// all-reduce of the two scalars, then the contributions of every local interface DOF summed in rank order
int halo_sum_peers(fdapde_ctx* c, double* v, const double* part, int np, bool unpack) {
    hipStream_t st = c->stream;
    const int n_peers = (int)c->peer_rank.size();
    const int64_t n_send = c->peer_off.empty() ? 0 : c->peer_off.back();
    double* scal = c->hbuf.p + c->n_if;
    if (c->xd.on) {   // in-process direct transport: pack into this call's buffer, drain, all ranks arrive, fetch from the peers' buffers + sum the scalar slots
        const int par = c->xd.parity;
        c->xd.parity ^= 1;
        const int64_t off = ((int64_t)par * 2 + 0) * fdapde_ctx::kSlotDoubles;
        hipLaunchKernelGGL(k_peer_pack, dim3(g1(n_send > 0 ? n_send : 1)), dim3(256), 0, st, n_send, c->peer_send_dof.p, v, c->peer_sendbuf.p + (size_t)par * (size_t)c->xd.n_send,
                           part, np, c->xd.slots.p + off);
        HIPCHK(c, hipStreamSynchronize(st));
        if (c->xd.arrive(c->xd.user) != 0) return fail(c, FDAPDE_ERCCL, "in-process transport: a rank did not arrive");
        if (n_send > 0) hipLaunchKernelGGL(k_peer_fetch, dim3(g1(n_send)), dim3(256), 0, st, n_send, c->xd.remote[par].p, c->peer_recvbuf.p);
        hipLaunchKernelGGL(k_slot_sum, dim3(1), dim3(64), 0, st, c->world, 2, c->xd.slot_ptr.p, off, scal);
        if (c->n_loc_if > 0)
            hipLaunchKernelGGL(k_peer_sum, dim3(g1(c->n_loc_if)), dim3(256), 0, st, c->n_loc_if, c->halo_dof.p, c->peer_src_off.p, c->peer_src.p,
                               c->peer_recvbuf.p, v, c->hbuf.p, unpack ? 1 : 0);
        HIPCHK(c, hipGetLastError());
        return FDAPDE_OK;
    }
    hipLaunchKernelGGL(k_peer_pack, dim3(g1(n_send > 0 ? n_send : 1)), dim3(256), 0, st, n_send, c->peer_send_dof.p, v, c->peer_sendbuf.p, part, np,
                       scal);
    if (c->ar_fn) {   // host-staged
        if (n_peers > 0) {
            if (!c->xchg_fn) return fail(c, FDAPDE_ENOTINIT, "fdapde_comm_set_exchange_callback not called");
            c->xchg_send_h.resize((size_t)n_send), c->xchg_recv_h.resize((size_t)n_send);
            HIPCHK(c, hipMemcpyAsync(c->xchg_send_h.data(), c->peer_sendbuf.p, sizeof(double) * (size_t)n_send, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (c->xchg_fn(c->xchg_user, n_peers, c->peer_rank.data(), c->peer_off.data(), c->xchg_send_h.data(), c->xchg_recv_h.data()) != 0)
                return fail(c, FDAPDE_ERCCL, "exchange callback failed");
            HIPCHK(c, hipMemcpyAsync(c->peer_recvbuf.p, c->xchg_recv_h.data(), sizeof(double) * (size_t)n_send, hipMemcpyHostToDevice, st));
        }
        if (int rc = allreduce_sum(c, scal, 2)) return rc;
    } else {
        if (n_peers > 0) {   // all sends and receives of the exchange form one group (one fused launch, no ordering between peers)
            RCCLCHK(c, g_rccl.GroupStart());
            for (int q = 0; q < n_peers; ++q) {
                const size_t cnt = (size_t)(c->peer_off[(size_t)q + 1] - c->peer_off[(size_t)q]);
                RCCLCHK(c, g_rccl.Send(c->peer_sendbuf.p + c->peer_off[(size_t)q], cnt, ncclFloat64, c->peer_rank[(size_t)q], c->comm, st));
                RCCLCHK(c, g_rccl.Recv(c->peer_recvbuf.p + c->peer_off[(size_t)q], cnt, ncclFloat64, c->peer_rank[(size_t)q], c->comm, st));
            }
            RCCLCHK(c, g_rccl.GroupEnd());
        }
        RCCLCHK(c, g_rccl.AllReduce(scal, scal, 2, ncclFloat64, ncclSum, c->comm, st));
    }
    if (c->n_loc_if > 0)
        hipLaunchKernelGGL(k_peer_sum, dim3(g1(c->n_loc_if)), dim3(256), 0, st, c->n_loc_if, c->halo_dof.p, c->peer_src_off.p, c->peer_src.p,
                           c->peer_recvbuf.p, v, c->hbuf.p, unpack ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}

int halo_sum(fdapde_ctx* c, double* v, const double* part, int np, bool unpack) {
    if (c->peer_mode) return halo_sum_peers(c, v, part, np, unpack);
    hipStream_t st = c->stream;
    const unsigned grid = g1(c->n_loc_if > 0 ? c->n_loc_if : 1);
    hipLaunchKernelGGL(k_halo_pack_all, dim3(g1(c->n_if > 0 ? c->n_if : 1)), dim3(256), 0, st, c->n_if, c->halo_inv.p, v, c->hbuf.p, part,
                       np);   // one launch writes every slot (zeros where this rank has no DOF): no memset
    if (int rc = allreduce_sum(c, c->hbuf.p, (size_t)(c->n_if + 2))) return rc;
    if (c->n_loc_if > 0 && unpack)
        hipLaunchKernelGGL(k_halo_unpack, dim3(grid), dim3(256), 0, st, c->n_loc_if, c->halo_dof.p, c->halo_pos.p, c->hbuf.p, v);
    HIPCHK(c, hipGetLastError());
    return FDAPDE_OK;
}

int e_comm_unique_id(void* out128) {
    if (!out128) return FDAPDE_EINVAL;
    std::string err;
    if (!g_rccl.load(err)) return FDAPDE_ERCCL;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return FDAPDE_ERCCL;
    std::memcpy(out128, &id, sizeof id);
    return FDAPDE_OK;
}

int e_comm_init(fdapde_ctx* c, int32_t world, int32_t rank, const void* unique_id128) {
    if (!c || !unique_id128 || world < 1 || rank < 0 || rank >= world) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!g_rccl.load(c->err)) return FDAPDE_ERCCL;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) (void)g_rccl.CommDestroy(c->comm), c->comm = nullptr;
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof id);
    {   // RCCL reads the runtime's last-error slot while it sets up: an error some earlier, unrelated call of this process left there
        // (HIP keeps it until somebody asks) would be reported as RCCL's own
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess && std::getenv("FDAPDE_DEBUG_SETUP")) std::fprintf(stderr, "comm_init: stale HIP error cleared: %s\n", hipGetErrorString(stale));
    }
    RCCLCHK(c, g_rccl.CommInitRank(&c->comm, world, id, rank));
    c->world = world, c->rank = rank, c->ar_fn = nullptr;
    return FDAPDE_OK;
}

// how many ranks the context's communicator REALLY has -- asked of RCCL itself (ncclCommCount), not echoed from what the caller passed to
// fdapde_comm_init: a bench line that says "8 GPUs" should be able to prove that its collective ran over 8 ranks.  Host-staged transport: the
// world size the callback was registered with.  No communicator: 1.
int e_comm_count(fdapde_ctx* c, int32_t* ranks) {
    if (!c || !ranks) return FDAPDE_EINVAL;
    *ranks = 1;
    if (c->comm) {
        if (!g_rccl.CommCount) return fail(c, FDAPDE_EUNSUPPORTED, "this librccl has no ncclCommCount");
        int n = 0;
        RCCLCHK(c, g_rccl.CommCount(c->comm, &n));
        *ranks = n;
    } else if (c->ar_fn)
        *ranks = c->world;
    return FDAPDE_OK;
}

// sum (op 0) or max (op 1) of n host doubles over the ranks of the context's communicator, in place: the barrier / timing reductions of a
// multi-process driver that holds no other collective library (bench.py's ranks load this library and nothing else that touches the GPU)
int e_comm_allreduce(fdapde_ctx* c, double* host_inout, int32_t n, int32_t op) {
    if (!c || !host_inout || n < 1 || (op != 0 && op != 1)) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->comm && !c->ar_fn) return fail(c, FDAPDE_ENOTINIT, "call fdapde_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->ar_fn) {
        if (op != 0 && c->world > 1) return fail(c, FDAPDE_EUNSUPPORTED, "the host-staged transport only sums");
        if (c->world > 1 && c->ar_fn(c->ar_user, host_inout, (int64_t)n) != 0) return fail(c, FDAPDE_ERCCL, "all-reduce callback failed");
        return FDAPDE_OK;
    }
    HIPCHK(c, c->ar_dev.alloc((size_t)n));
    HIPCHK(c, hipMemcpyAsync(c->ar_dev.p, host_inout, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(c, g_rccl.AllReduce(c->ar_dev.p, c->ar_dev.p, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(host_inout, c->ar_dev.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDAPDE_OK;
}

// one decision for all ranks of a multi-GPU job: how many of them say yes (a single-GPU context: its own answer).  COLLECTIVE.
int ranks_saying_yes(fdapde_ctx* c, bool mine, int* yes) {
    *yes = mine ? 1 : 0;
    if (c->world <= 1 || (!c->comm && !c->ar_fn)) return FDAPDE_OK;
    double v = mine ? 1.0 : 0.0;
    if (int rc = e_comm_allreduce(c, &v, 1, 0)) return rc;
    *yes = (int)(v + 0.5);
    return FDAPDE_OK;
}

int e_comm_init_callback(fdapde_ctx* c, int32_t world, int32_t rank, fdapde_allreduce_fn fn, void* user) {
    if (!c || !fn || world < 1 || rank < 0 || rank >= world) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm), c->comm = nullptr;
    c->ar_fn = fn, c->ar_user = user, c->world = world, c->rank = rank;
    return FDAPDE_OK;
}

int e_halo_setup(fdapde_ctx* c, int64_t n_if_global, int64_t n_if_local, const int32_t* local_dof, const int32_t* if_index,
                      const uint8_t* owned) {
    if (!c || n_if_global < 0 || n_if_local < 0 || (n_if_local > 0 && (!local_dof || !if_index)) || !owned) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (!c->comm && !c->ar_fn) return fail(c, FDAPDE_ENOTINIT, "call fdapde_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_host(c, kHostPerm)) return rc;
    const HostSpace& hs = c->hs;
    std::vector<int32_t> dof_i((size_t)n_if_local), pos((size_t)n_if_local);
    for (int64_t k = 0; k < n_if_local; ++k) {
        if (local_dof[k] < 0 || local_dof[k] >= hs.n_dofs || if_index[k] < 0 || if_index[k] >= n_if_global)
            return fail(c, FDAPDE_EINVAL, "interface map entry out of range");
        dof_i[(size_t)k] = hs.dof_e2i[(size_t)local_dof[k]], pos[(size_t)k] = if_index[k];
    }
    std::vector<uint8_t> own_i((size_t)hs.n_dofs);
    for (int64_t i = 0; i < hs.n_dofs; ++i) own_i[(size_t)i] = owned[hs.dof_i2e[(size_t)i]] ? 1 : 0;
    std::vector<int32_t> inv((size_t)(n_if_global > 0 ? n_if_global : 1), -1), slot((size_t)hs.n_dofs + 2, -1);
    for (int64_t k = 0; k < n_if_local; ++k) inv[(size_t)pos[(size_t)k]] = dof_i[(size_t)k], slot[(size_t)dof_i[(size_t)k]] = pos[(size_t)k];
    HIPCHK(c, c->halo_inv.upload(inv.data(), inv.size(), c->stream));
    HIPCHK(c, c->if_slot.upload(slot.data(), slot.size(), c->stream));
    HIPCHK(c, c->halo_dof.upload(dof_i.data(), dof_i.size(), c->stream));
    HIPCHK(c, c->halo_pos.upload(pos.data(), pos.size(), c->stream));
    HIPCHK(c, c->owned.upload(own_i.data(), own_i.size(), c->stream));
    HIPCHK(c, c->hbuf.alloc((size_t)n_if_global + 2));
    HIPCHK(c, c->sbuf.alloc(8));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_if = n_if_global, c->n_loc_if = n_if_local, c->halo_ready = true, c->peer_mode = false;
    return FDAPDE_OK;
}

// Row-distributed multi-GPU form: every DOF of the whole mesh is OWNED by one rank; a rank's sub-mesh holds every cell touching one of its
// DOFs (its own cells + one layer of cells of its neighbours), so that its assembly completes the rows of its DOFs without any exchange.
int e_rowdist_setup(fdapde_ctx* c, const int64_t* dof_key, const int32_t* dof_owner) {
    if (!c || !dof_key || !dof_owner) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (!c->comm && !c->ar_fn) return fail(c, FDAPDE_ENOTINIT, "call fdapde_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    release_rowdist(c);
    if (int rc = ensure_host(c, kHostPerm)) return rc;
    const HostSpace& hs = c->hs;
    c->rd.owner_i.resize((size_t)hs.n_dofs), c->rd.key_i.resize((size_t)hs.n_dofs);
    std::vector<uint8_t> own((size_t)hs.n_dofs);
    for (int64_t i = 0; i < hs.n_dofs; ++i) {
        const int32_t e = hs.dof_i2e[(size_t)i];
        if (dof_owner[e] < 0 || dof_owner[e] >= c->world) return fail(c, FDAPDE_EINVAL, "fdapde_rowdist_setup: owner out of range");
        c->rd.owner_i[(size_t)i] = dof_owner[e], c->rd.key_i[(size_t)i] = dof_key[e], own[(size_t)i] = dof_owner[e] == c->rank ? 1 : 0;
    }
    HIPCHK(c, c->rd.owned.upload(own.data(), own.size(), c->stream));
    HIPCHK(c, c->sbuf.alloc(8));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->rd.ready = true, c->halo_ready = false;
    return FDAPDE_OK;
}

int e_comm_set_exchange_callback(fdapde_ctx* c, fdapde_exchange_fn fn, void* user) {
    if (!c || !fn) return FDAPDE_EINVAL;
    c->xchg_fn = fn, c->xchg_user = user;
    return FDAPDE_OK;
}

int e_halo_setup_peers(fdapde_ctx* c, int32_t n_peers, const int32_t* peer_rank, const int64_t* peer_off, const int32_t* peer_dof,
                            const uint8_t* owned) {
    if (!c || n_peers < 0 || !owned || (n_peers > 0 && (!peer_rank || !peer_off || !peer_dof))) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (!c->dev_ready) return fail(c, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    if (!c->comm && !c->ar_fn) return fail(c, FDAPDE_ENOTINIT, "call fdapde_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_host(c, kHostPerm)) return rc;
    const HostSpace& hs = c->hs;
    const int64_t n_send = n_peers > 0 ? peer_off[n_peers] : 0;
    for (int q = 0; q < n_peers; ++q)
        if (peer_rank[q] < 0 || peer_rank[q] >= c->world || peer_rank[q] == c->rank || (q > 0 && peer_rank[q] <= peer_rank[q - 1]) ||
            peer_off[q + 1] < peer_off[q] || (q == 0 && peer_off[0] != 0))
            return fail(c, FDAPDE_EINVAL, "peer list: ranks must be ascending, without this rank, offsets non-decreasing from 0");
    std::vector<int32_t> send_dof((size_t)(n_send > 0 ? n_send : 1), 0);
    std::vector<int32_t> k_of((size_t)hs.n_dofs, -1), if_dof;   // internal DOF -> local interface index
    for (int64_t j = 0; j < n_send; ++j) {
        if (peer_dof[j] < 0 || peer_dof[j] >= hs.n_dofs) return fail(c, FDAPDE_EINVAL, "peer list: DOF id out of range");
        const int32_t d = hs.dof_e2i[(size_t)peer_dof[j]];
        send_dof[(size_t)j] = d;
        if (k_of[(size_t)d] < 0) k_of[(size_t)d] = (int32_t)if_dof.size(), if_dof.push_back(d);
    }
    const int64_t n_loc = (int64_t)if_dof.size();
    // contributions of every local interface DOF in ascending rank order: the peers are ascending, this rank's own goes where its
    // rank falls among them
    std::vector<int32_t> cnt((size_t)n_loc + 1, 0);
    for (int64_t j = 0; j < n_send; ++j) ++cnt[(size_t)k_of[(size_t)send_dof[(size_t)j]] + 1];
    for (int64_t k = 0; k < n_loc; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k] + 1;   // + 1: the own contribution
    std::vector<int32_t> src_off(cnt), src((size_t)(n_send + n_loc > 0 ? n_send + n_loc : 1), 0), fill_at(cnt.begin(), cnt.end() - 1);
    std::vector<uint8_t> own_in((size_t)n_loc, 0);
    std::vector<int32_t> last_peer((size_t)n_loc, -1);
    for (int q = 0; q < n_peers; ++q) {
        for (int64_t j = peer_off[q]; j < peer_off[q + 1]; ++j) {
            const int32_t k = k_of[(size_t)send_dof[(size_t)j]];
            if (last_peer[(size_t)k] == q) return fail(c, FDAPDE_EINVAL, "peer list: a DOF is listed twice for one peer");
            last_peer[(size_t)k] = q;
            if (peer_rank[q] > c->rank && !own_in[(size_t)k]) src[(size_t)fill_at[(size_t)k]++] = -1, own_in[(size_t)k] = 1;
            src[(size_t)fill_at[(size_t)k]++] = (int32_t)j;
        }
    }
    for (int64_t k = 0; k < n_loc; ++k)
        if (!own_in[(size_t)k]) src[(size_t)fill_at[(size_t)k]++] = -1;
    std::vector<uint8_t> own_i((size_t)hs.n_dofs);
    for (int64_t i = 0; i < hs.n_dofs; ++i) own_i[(size_t)i] = owned[hs.dof_i2e[(size_t)i]] ? 1 : 0;
    std::vector<int32_t> slot((size_t)hs.n_dofs + 2, -1), pos((size_t)(n_loc > 0 ? n_loc : 1), 0);
    for (int64_t k = 0; k < n_loc; ++k) slot[(size_t)if_dof[(size_t)k]] = (int32_t)k, pos[(size_t)k] = (int32_t)k;
    if (if_dof.empty()) if_dof.push_back(0);
    hipStream_t st = c->stream;
    HIPCHK(c, c->peer_send_dof.upload(send_dof.data(), send_dof.size(), st));
    HIPCHK(c, c->peer_src_off.upload(src_off.data(), src_off.size(), st));
    HIPCHK(c, c->peer_src.upload(src.data(), src.size(), st));
    HIPCHK(c, c->peer_sendbuf.alloc(2 * (size_t)(n_send > 0 ? n_send : 1)));   // (two parity buffers: the in-process direct transport)
    c->xd.on = false, c->xd.n_send = n_send > 0 ? n_send : 1;
    HIPCHK(c, c->peer_recvbuf.alloc((size_t)(n_send > 0 ? n_send : 1)));
    HIPCHK(c, c->if_slot.upload(slot.data(), slot.size(), st));
    HIPCHK(c, c->halo_dof.upload(if_dof.data(), if_dof.size(), st));
    HIPCHK(c, c->halo_pos.upload(pos.data(), pos.size(), st));
    HIPCHK(c, c->owned.upload(own_i.data(), own_i.size(), st));
    HIPCHK(c, c->hbuf.alloc((size_t)n_loc + 2));
    HIPCHK(c, c->sbuf.alloc(8));
    HIPCHK(c, hipStreamSynchronize(st));
    c->peer_rank.assign(peer_rank, peer_rank + n_peers);
    c->peer_off.assign(1, 0);
    if (n_peers > 0) c->peer_off.assign(peer_off, peer_off + n_peers + 1);
    c->n_if = n_loc, c->n_loc_if = n_loc, c->peer_mode = true, c->halo_ready = true;
    return FDAPDE_OK;
}


// the unit's code object is loaded when one of its kernels is first looked up (HIP defers it): done at context creation, so that the
// first solve of a process does not pay for it (6 ms for the smoke problem after the library was split into units)
void preload_dist() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_peer_pack));
    (void)hipGetLastError();
}

}   // namespace fdapde_engine
