// eng_group.hip -- ONE context over SEVERAL devices (fdapde_ctx_create_multi), and the public face of the device-side partitioner.
//
// The reference's user holds one PDE object in one thread (fdaPDE/pde/pde.h:58-105); north_star wants the mesh sharded over the GPUs of a node
// BEHIND that interface.  A multi-device context is a ROOT context (the one handed to the caller: it keeps the whole mesh and the whole function
// space on its first device, so every index getter -- dofs(), the CSR pattern, quadrature nodes, point location -- is the single-device code
// path, reference numbering included) plus one RANK context per device, each driven by a worker thread of its own ("one context per host
// thread", include/fdapde_hip.h).  fdapde_dofs_build partitions the resident mesh on the device (dev_partition.hip), hands every rank its
// sub-mesh, builds the ranks' spaces side by side and wires them with an IN-PROCESS transport (barrier + rank-ordered sums in shared memory:
// the host-staged callbacks of eng_dist.hip, natively).  Problem data are dealt to the ranks by cell id / DOF id, fdapde_init / fdapde_solve run
// on all ranks at once, results are gathered from the DOFs each rank owns.
//
// Two forms, as for rank processes (include/fdapde_hip.h): the ROW-DISTRIBUTED one first -- complete rows per owner, the whole Krylov iteration
// as one persistent launch per device, the launches exchanging through peer-mapped boards (same process: the boards' device pointers themselves,
// hipDeviceEnablePeerAccess; persist_engine.hip build_rowdist) -- and, when the library declines a system in that form (FDAPDE_EUNSUPPORTED on
// every rank), the ELEMENT partition with the neighbour exchange of interface sums: the group re-partitions, re-deals the problem data it keeps
// host copies of, re-assembles and solves again.  Inside the row-distributed iteration nothing crosses the host; the in-process transport carries
// set-up data only.  (The element form exchanges through host staging here; rank PROCESSES use RCCL for it -- fdapde_comm_init.)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "context.h"
#include "dev_partition.h"
#include "engine.h"

namespace fdapde_engine {

namespace {

// every rank arrives, all leave; a rank that fails outside a collective breaks the rendezvous so that nobody waits for it for ever
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    int n = 1, waiting = 0;
    uint64_t gen = 0;
    bool broken = false;
    int timeout_s = 300;
    bool arrive() {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t g = gen;
        if (++waiting == n) {
            waiting = 0, ++gen;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return gen != g || broken; })) broken = true, cv.notify_all();
        return gen != g && !broken;
    }
    void reset() {
        std::lock_guard<std::mutex> lk(mu);
        waiting = 0, broken = false;
    }
    void fail() {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};

struct StoredTerm {
    fdapde_term t{};
    std::vector<double> data;   // row-major (nq * n_cells) x width, whole mesh
    int width = 0;
};

}   // namespace

struct Group;
struct GroupRank {
    fdapde_ctx* ctx = nullptr;
    void* cb_user = nullptr;         // (the callbacks' cookie: lives as long as the rank's communicator)
    int device = -1;
    std::thread th;
    std::vector<int32_t> cell_ids;   // local cell -> cell of the whole mesh
    std::vector<int32_t> l2g;        // local node -> node of the whole mesh
    std::vector<int32_t> gdof;       // local DOF (reference numbering of the rank's space) -> DOF of the whole mesh (reference numbering)
    std::vector<uint8_t> owned;      // local DOF: this rank's to report
    std::vector<int64_t> gslot;      // local pattern entry -> entry of the whole mesh's pattern (built on first use)
    // scratch of the transports
    const double* x_send = nullptr;
    double* x_recv = nullptr;
    const int32_t* x_rank = nullptr;
    const int64_t* x_off = nullptr;
    int32_t x_np = 0;
    double* ar_buf = nullptr;
    int64_t ar_cnt = 0;
    std::vector<double> ar_tmp;
    std::vector<double> h0, h1;      // host staging of scatter / gather
    fdapde_info info{};
    int rc = FDAPDE_OK;
};

struct Group {
    fdapde_ctx* root = nullptr;
    int n = 0, form = kPartitionRowdist, order = 0, share = 1;
    bool ranks_built = false, initialised = false, form_locked = false;
    bool direct = true;   // element form: exchanges fetch from the peers' buffers (knob group_direct 0: staged through host memory)
    std::vector<GroupRank> rk;
    Rendezvous rdv;
    // worker threads: one job at a time, every rank runs it
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t job_gen = 0;
    int pending = 0;
    bool quit = false;
    std::function<int(int)> job;
    // what the caller handed over, kept for a change of form
    std::vector<StoredTerm> op;
    bool have_op = false;
    std::vector<double> fq;
    int fq_cols = 0;
    std::vector<double> gdir;
    bool have_g = false;
    std::vector<uint8_t> bnd_override;   // fdapde_dofs_set_boundary on the group (whole-mesh DOF ids), empty: the root's own flags
    fdapde_options init_opt{};
    bool have_init_opt = false;
    std::vector<std::pair<std::string, int32_t>> knobs;   // fdapde_tune calls, replayed on ranks that are rebuilt
    fdapde_info info{};
    std::vector<uint8_t> nnz_rank;       // element form: the rank that carries a whole-mesh pattern entry of a matrix handed to fdapde_lin_compute
    double t_partition_ms = 0, t_rank_setup_ms = 0;
};

namespace {

void worker_main(Group* g, int r) {
    (void)hipSetDevice(g->rk[(size_t)r].device);
    uint64_t seen = 0;
    for (;;) {
        std::function<int(int)> job;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->quit || g->job_gen != seen; });
            if (g->quit) return;
            seen = g->job_gen, job = g->job;
        }
        int rc = FDAPDE_OK;
        try {
            rc = job(r);
        } catch (const std::bad_alloc&) {
            rc = FDAPDE_ENOMEM;
        } catch (...) {
            rc = FDAPDE_EINVAL;
        }
        g->rk[(size_t)r].rc = rc;
        if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV && rc != FDAPDE_EUNSUPPORTED) g->rdv.fail();   // (the others may be waiting for this rank in a collective)
        {
            std::lock_guard<std::mutex> lk(g->mu);
            if (--g->pending == 0) g->cv_done.notify_all();
        }
    }
}

// runs fn(rank) on every rank's thread; -> FDAPDE_OK, or the first rank's status that is not (its text goes to the root's error slot)
int run_all(Group* g, const std::function<int(int)>& fn) {
    g->rdv.reset();
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->job = fn, g->pending = g->n, ++g->job_gen;
    }
    g->cv_job.notify_all();
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv_done.wait(lk, [&] { return g->pending == 0; });
    }
    // a hard error outranks "not converged" / "declined": the ranks that were left alone in a collective report the transport, not the cause
    int worst = FDAPDE_OK, at = -1;
    auto weight = [](int rc) { return rc == FDAPDE_OK ? 0 : rc == FDAPDE_ENOCONV ? 1 : rc == FDAPDE_EUNSUPPORTED ? 2 : rc == FDAPDE_ERCCL ? 3 : 4; };
    for (int r = 0; r < g->n; ++r)
        if (weight(g->rk[(size_t)r].rc) > weight(worst)) worst = g->rk[(size_t)r].rc, at = r;
    if (worst != FDAPDE_OK) g->root->err = "rank " + std::to_string(at) + " (device " + std::to_string(g->rk[(size_t)at].device) + "): " + g->rk[(size_t)at].ctx->err;
    return worst;
}

// ---- in-process transport: the host-staged callbacks of eng_dist.hip ------------------------------------------------------------------------
struct RankCookie {
    Group* g;
    int rank;
};
int cb_allreduce(void* user, double* buf, int64_t count) {
    RankCookie* u = static_cast<RankCookie*>(user);
    Group* g = u->g;
    GroupRank& me = g->rk[(size_t)u->rank];
    me.ar_buf = buf, me.ar_cnt = count;
    if (!g->rdv.arrive()) return 1;
    me.ar_tmp.assign((size_t)count, 0.0);
    for (int p = 0; p < g->n; ++p) {   // ascending rank order on every rank: identical bits everywhere (the ranks take decisions on these sums)
        const GroupRank& o = g->rk[(size_t)p];
        if (o.ar_cnt != count) {   // (ranks in different collectives: nobody may wait for the second arrival)
            g->rdv.fail();
            return 1;
        }
        for (int64_t i = 0; i < count; ++i) me.ar_tmp[(size_t)i] += o.ar_buf[i];
    }
    if (!g->rdv.arrive()) return 1;   // everybody has read everybody's buffer
    std::memcpy(buf, me.ar_tmp.data(), sizeof(double) * (size_t)count);
    return 0;
}
int cb_exchange(void* user, int32_t n_peers, const int32_t* peer_rank, const int64_t* peer_off, const double* send, double* recv) {
    RankCookie* u = static_cast<RankCookie*>(user);
    Group* g = u->g;
    GroupRank& me = g->rk[(size_t)u->rank];
    me.x_np = n_peers, me.x_rank = peer_rank, me.x_off = peer_off, me.x_send = send, me.x_recv = recv;
    if (!g->rdv.arrive()) return 1;
    int bad = 0;
    for (int q = 0; q < n_peers; ++q) {
        const GroupRank& o = g->rk[(size_t)peer_rank[q]];
        int j = -1;
        for (int k = 0; k < o.x_np; ++k)
            if (o.x_rank[k] == u->rank) j = k;
        const int64_t cnt = peer_off[q + 1] - peer_off[q];
        if (j < 0 || o.x_off[j + 1] - o.x_off[j] != cnt) {
            bad = 1;
            continue;
        }
        std::memcpy(recv + peer_off[q], o.x_send + o.x_off[j], sizeof(double) * (size_t)cnt);
    }
    if (!g->rdv.arrive()) return 1;
    return bad;
}

int cb_arrive(void* user) {
    RankCookie* u = static_cast<RankCookie*>(user);
    return u->g->rdv.arrive() ? 0 : 1;
}

inline int width_of(int kind, int N) { return kind == FDAPDE_DIFFUSION ? N * N : kind == FDAPDE_ADVECTION ? N : 1; }

// vertex pair of the edge DOF in local slot (M + 1) + j (tables.cpp EDGE2 / EDGE3)
const int kEdge2[3][2] = {{0, 1}, {0, 2}, {1, 2}};
const int kEdge3[6][2] = {{1, 2}, {0, 2}, {0, 1}, {1, 3}, {2, 3}, {0, 3}};

int copy_d2h(fdapde_ctx* root, void* dst, const void* src, size_t bytes) {
    HIPCHK(root, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));   // (blocking, legacy stream: callable from any thread with the root's device current)
    return FDAPDE_OK;
}

// ---- ranks from the partition: sub-meshes, spaces, transports, DOF maps, exchange lists.  Called with the root's space built. ---------------
int build_ranks(Group* g) {
    fdapde_ctx* root = g->root;
    const HostSpace& H = root->hs;
    const int n = g->n, nv = H.M + 1;
    HIPCHK(root, hipSetDevice(root->device));
    const auto t0 = std::chrono::steady_clock::now();
    DevPartition part;
    if (int rc = dev_partition_build(H.M, H.N, H.n_nodes, H.n_cells, root->mesh_nodes.p, root->mesh_cells.p, root->mesh_nbnd.p, n, g->form, root->stream, &part, root->err)) {
        dev_partition_release(&part);
        return rc;
    }
    for (int r = 0; r < n; ++r)
        if (part.ranks[(size_t)r].n_cells == 0) {
            dev_partition_release(&part);
            return fail(root, FDAPDE_EINVAL, "multi-device context: a rank's share of the mesh is empty (fewer cells than devices?)");
        }
    std::vector<int32_t> node_owner((size_t)H.n_nodes);
    std::vector<uint64_t> node_mask;
    if (int rc = copy_d2h(root, node_owner.data(), part.node_owner, sizeof(int32_t) * node_owner.size())) {
        dev_partition_release(&part);
        return rc;
    }
    const auto t1 = std::chrono::steady_clock::now();
    g->t_partition_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    const int nb = H.nb, order = H.order;
    if (order != 1)   // (P2: the ranks' DOFs are matched to the whole mesh's through the DOF tables)
        if (int rc = ensure_host(root, kHostDofs)) {
            dev_partition_release(&part);
            return rc;
        }
    std::vector<uint8_t> bnd_g = g->bnd_override.empty() ? std::vector<uint8_t>(H.dof_bnd.begin(), H.dof_bnd.end()) : g->bnd_override;
    int rc_all = run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        fdapde_ctx* c = R.ctx;
        const RankMeshDev& P = part.ranks[(size_t)r];
        const bool dbg = std::getenv("FDAPDE_DEBUG_SETUP") != nullptr;
        auto tm0 = std::chrono::steady_clock::now();
        auto mark = [&](const char* what) {
            if (!dbg) return;
            const auto t = std::chrono::steady_clock::now();
            std::fprintf(stderr, "rank %d set-up: %-28s %8.2f ms\n", r, what, std::chrono::duration<double, std::milli>(t - tm0).count());
            tm0 = t;
        };
        // ---- the sub-mesh: device (root's) -> host -> the rank's context
        (void)hipSetDevice(root->device);
        R.cell_ids.resize((size_t)P.n_cells), R.l2g.resize((size_t)P.n_nodes);
        std::vector<double> nodes((size_t)P.n_nodes * H.N);
        std::vector<int32_t> cells((size_t)P.n_cells * nv), owner_l((size_t)P.n_nodes);
        std::vector<uint8_t> nb_l((size_t)P.n_nodes);
        auto d2h = [&](void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
        if (!d2h(R.cell_ids.data(), P.cell_ids, sizeof(int32_t) * R.cell_ids.size()) || !d2h(R.l2g.data(), P.l2g, sizeof(int32_t) * R.l2g.size()) ||
            !d2h(nodes.data(), P.nodes, sizeof(double) * nodes.size()) || !d2h(cells.data(), P.cells, sizeof(int32_t) * cells.size()) ||
            !d2h(nb_l.data(), P.bnd, nb_l.size()) || !d2h(owner_l.data(), P.node_owner, sizeof(int32_t) * owner_l.size()))
            return fail(c, FDAPDE_EHIP, "multi-device context: fetching a rank's sub-mesh from the partition failed");
        (void)hipSetDevice(R.device);
        mark("sub-mesh device -> host");
        if (int rc = fdapde_mesh_upload(c, H.M, H.N, P.n_nodes, nodes.data(), P.n_cells, cells.data(), nb_l.data())) return rc;
        mark("fdapde_mesh_upload");
        int64_t nd = 0;
        if (int rc = e_dofs_build(c, order, &nd)) return rc;
        mark("fdapde_dofs_build");
        for (const auto& kv : g->knobs) (void)fdapde_tune(c, kv.first.c_str(), kv.second);
        // ---- transport
        if (!R.cb_user) R.cb_user = new RankCookie{g, r};
        void* u = R.cb_user;
        if (int rc = e_comm_init_callback(c, n, r, &cb_allreduce, u)) return rc;
        if (int rc = e_comm_set_exchange_callback(c, &cb_exchange, u)) return rc;
        // ---- local DOF -> DOF of the whole mesh.  P1: a DOF is a node (lagrangian_basis.h:96-99): the node map itself.  P2: through the two DOF tables
        //      (the rank's own enumeration of its sub-mesh against the root's)
        const HostSpace& L = c->hs;
        if (order == 1) {
            R.gdof.assign(R.l2g.begin(), R.l2g.end());
        } else {
            if (int rc = ensure_host(c, kHostDofs)) return rc;
            R.gdof.assign((size_t)L.n_dofs, -1);
            for (int64_t cl = 0; cl < L.n_cells; ++cl) {
                const int32_t* tl = L.dofs.data() + cl * nb;
                const int32_t* tg = H.dofs.data() + (int64_t)R.cell_ids[(size_t)cl] * nb;
                for (int j = 0; j < nb; ++j) R.gdof[(size_t)tl[j]] = tg[j];
            }
            for (int64_t d = 0; d < L.n_dofs; ++d)
                if (R.gdof[(size_t)d] < 0) return fail(c, FDAPDE_EINVAL, "multi-device context: a DOF of a sub-mesh is touched by none of its cells");
        }
        if ((int64_t)R.gdof.size() != L.n_dofs) return fail(c, FDAPDE_EINVAL, "multi-device context: a rank's DOF count differs from its node count");
        R.gslot.clear();
        mark("DOF maps");
        // the boundary flags are the WHOLE mesh's (the 2-D rule "edge seen by one cell", triangulation.h:177,187, would mark interface edges)
        std::vector<uint8_t> bl((size_t)L.n_dofs);
        bool differs = false;
        for (int64_t d = 0; d < L.n_dofs; ++d) bl[(size_t)d] = bnd_g[(size_t)R.gdof[(size_t)d]], differs = differs || bl[(size_t)d] != L.dof_bnd[(size_t)d];
        if (differs)
            if (int rc = e_dofs_set_boundary(c, bl.data())) return rc;
        // ---- who owns a DOF
        std::vector<int32_t> own((size_t)L.n_dofs, -1);
        if (g->form == kPartitionRowdist) {
            // a vertex DOF belongs to its node's owner, an edge DOF to the owner of its end node with the LOWER global id: that rank's sub-mesh
            // holds every cell touching the node, hence every cell touching the edge
            if (order == 1) {
                for (int64_t d = 0; d < L.n_dofs; ++d) own[(size_t)d] = owner_l[(size_t)d];
            } else
                for (int64_t cl = 0; cl < L.n_cells; ++cl) {
                    const int32_t* tl = L.dofs.data() + cl * nb;
                    const int32_t* vl = L.cells.data() + cl * nv;
                    for (int v = 0; v < nv; ++v) own[(size_t)tl[v]] = owner_l[(size_t)vl[v]];
                    for (int j = nv; j < nb; ++j) {
                        const int* e = H.M == 2 ? kEdge2[j - nv] : kEdge3[j - nv];
                        const int32_t ga = R.l2g[(size_t)vl[e[0]]], gb = R.l2g[(size_t)vl[e[1]]];
                        own[(size_t)tl[j]] = node_owner[(size_t)(ga < gb ? ga : gb)];
                    }
                }
            R.owned.resize((size_t)L.n_dofs);
            for (int64_t d = 0; d < L.n_dofs; ++d) R.owned[(size_t)d] = own[(size_t)d] == r ? 1 : 0;
            if (g->share > 1) (void)fdapde_tune(c, "rowdist_share", g->share);
            std::vector<int64_t> key((size_t)L.n_dofs);
            for (int64_t d = 0; d < L.n_dofs; ++d) key[(size_t)d] = R.gdof[(size_t)d];
            if (int rc = e_rowdist_setup(c, key.data(), own.data())) return rc;
        }
        mark("owners + exchange set-up");
        return FDAPDE_OK;
    });
    dev_partition_release(&part);
    if (rc_all != FDAPDE_OK) return rc_all;
    if (g->form == kPartitionElements) {
        // ---- element form: ranks touching a DOF (bit mask over the whole mesh's DOFs), owner = the lowest, neighbour lists in ascending DOF id
        std::vector<uint64_t> mask((size_t)H.n_dofs, 0);
        for (int r = 0; r < n; ++r)
            for (int32_t gd : g->rk[(size_t)r].gdof) mask[(size_t)gd] |= uint64_t(1) << r;
        rc_all = run_all(g, [&](int r) -> int {
            GroupRank& R = g->rk[(size_t)r];
            fdapde_ctx* c = R.ctx;
            const int64_t nl = (int64_t)R.gdof.size();
            R.owned.resize((size_t)nl);
            std::vector<std::pair<int32_t, int32_t>> shared;   // (whole-mesh DOF, local DOF) of the interface DOFs
            for (int64_t d = 0; d < nl; ++d) {
                const uint64_t m = mask[(size_t)R.gdof[(size_t)d]];
                R.owned[(size_t)d] = (m & (~m + 1)) == (uint64_t(1) << r) ? 1 : 0;
                if (m & (m - 1)) shared.push_back({R.gdof[(size_t)d], (int32_t)d});
            }
            std::sort(shared.begin(), shared.end());
            std::vector<int32_t> pr, pd;
            std::vector<int64_t> po(1, 0);
            for (int q = 0; q < n; ++q) {
                if (q == r) continue;
                const size_t before = pd.size();
                for (const auto& s : shared)
                    if (mask[(size_t)s.first] >> q & 1) pd.push_back(s.second);
                if (pd.size() > before) pr.push_back(q), po.push_back((int64_t)pd.size());
            }
            release_rowdist(c);   // (a change of form: this rank is no longer an owner of complete rows)
            return e_halo_setup_peers(c, (int32_t)pr.size(), pr.data(), po.data(), pd.data(), R.owned.data());
        });
        if (rc_all != FDAPDE_OK) return rc_all;
        // ---- the in-process DIRECT transport of the element form (eng_dist.hip PeerDirect): every rank learns where, in its peers' send buffers, the values
        //      destined for it lie, and where every rank publishes its small vectors -- an exchange is then pack, drain, arrive, fetch; nothing is staged
        //      through host memory (the host-staged callbacks stay registered for the set-up collectives and as the fall-back: knob group_direct 0)
        if (g->direct) {
            rc_all = run_all(g, [&](int r) -> int {
                fdapde_ctx* c = g->rk[(size_t)r].ctx;
                HIPCHK(c, hipSetDevice(c->device));
                HIPCHK(c, c->xd.slots.alloc(4 * (size_t)fdapde_ctx::kSlotDoubles));
                HIPCHK(c, hipMemsetAsync(c->xd.slots.p, 0, sizeof(double) * 4 * (size_t)fdapde_ctx::kSlotDoubles, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                return FDAPDE_OK;
            });
            if (rc_all != FDAPDE_OK) return rc_all;
            rc_all = run_all(g, [&](int r) -> int {
                GroupRank& R = g->rk[(size_t)r];
                fdapde_ctx* c = R.ctx;
                HIPCHK(c, hipSetDevice(c->device));
                for (int q = 0; q < n; ++q)
                    if (g->rk[(size_t)q].device != R.device) {
                        const hipError_t e = hipDeviceEnablePeerAccess(g->rk[(size_t)q].device, 0);
                        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(c, FDAPDE_EHIP, "hipDeviceEnablePeerAccess failed");
                        (void)hipGetLastError();
                    }
                const int64_t n_send = c->peer_off.empty() ? 0 : c->peer_off.back();
                for (int par = 0; par < 2; ++par) {
                    std::vector<const double*> rem((size_t)std::max<int64_t>(n_send, 1), nullptr);
                    for (size_t qi = 0; qi < c->peer_rank.size(); ++qi) {
                        const fdapde_ctx* qc = g->rk[(size_t)c->peer_rank[qi]].ctx;
                        int j = -1;
                        for (size_t k = 0; k < qc->peer_rank.size(); ++k)
                            if (qc->peer_rank[k] == r) j = (int)k;
                        const int64_t cnt = c->peer_off[qi + 1] - c->peer_off[qi];
                        if (j < 0 || qc->peer_off[(size_t)j + 1] - qc->peer_off[(size_t)j] != cnt) return fail(c, FDAPDE_EINVAL, "multi-device context: the peers' exchange lists do not match");
                        const double* base = qc->peer_sendbuf.p + (size_t)par * (size_t)qc->xd.n_send + qc->peer_off[(size_t)j];
                        for (int64_t k = 0; k < cnt; ++k) rem[(size_t)(c->peer_off[qi] + k)] = base + k;
                    }
                    HIPCHK(c, c->xd.remote[par].upload(rem.data(), rem.size(), c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));   // (rem is a local)
                }
                std::vector<const double*> slots((size_t)n);
                for (int q = 0; q < n; ++q) slots[(size_t)q] = g->rk[(size_t)q].ctx->xd.slots.p;
                HIPCHK(c, c->xd.slot_ptr.upload(slots.data(), slots.size(), c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                c->xd.arrive = &cb_arrive, c->xd.user = R.cb_user, c->xd.parity = c->xd.ar_parity = 0, c->xd.on = true;
                return FDAPDE_OK;
            });
            if (rc_all != FDAPDE_OK) return rc_all;
        }
    }
    g->t_rank_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    g->ranks_built = true;
    g->nnz_rank.clear();
    return FDAPDE_OK;
}

// a rank's share of per-quadrature-node rows (row nq * cell + q of the whole mesh), `width` doubles per row, row-major
void deal_rows(const GroupRank& R, int nq, int width, const double* whole, std::vector<double>& out) {
    const int64_t ncl = (int64_t)R.cell_ids.size();
    out.resize((size_t)(ncl * nq * width));
    const size_t blk = (size_t)nq * width;
    for (int64_t cl = 0; cl < ncl; ++cl) std::memcpy(out.data() + (size_t)cl * blk, whole + (size_t)R.cell_ids[(size_t)cl] * blk, sizeof(double) * blk);
}

int push_operator(Group* g, int which /* -1: set_operator; 0 / 1: assemble_operator into that slot */, int assembly) {
    const HostSpace& H = g->root->hs;
    return run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        std::vector<fdapde_term> terms(g->op.size());
        std::vector<std::vector<double>> data(g->op.size());
        for (size_t k = 0; k < g->op.size(); ++k) {
            terms[k] = g->op[k].t;
            terms[k].data = nullptr;
            if (g->op[k].t.space_varying) {
                deal_rows(R, H.nq, g->op[k].width, g->op[k].data.data(), data[k]);
                terms[k].data = data[k].data();
            }
        }
        if (which < 0) return e_set_operator(R.ctx, (int32_t)terms.size(), terms.data());
        return e_assemble_operator(R.ctx, which, (int32_t)terms.size(), terms.data(), assembly);
    });
}

int push_forcing(Group* g) {
    const HostSpace& H = g->root->hs;
    return run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        if (g->fq_cols == 0) return e_set_forcing(R.ctx, nullptr, 0);
        const int64_t rows_g = (int64_t)H.nq * H.n_cells, rows_l = (int64_t)H.nq * (int64_t)R.cell_ids.size();
        std::vector<double> f((size_t)(rows_l * g->fq_cols)), one;
        for (int col = 0; col < g->fq_cols; ++col) {
            deal_rows(R, H.nq, 1, g->fq.data() + (size_t)col * rows_g, one);
            std::memcpy(f.data() + (size_t)col * rows_l, one.data(), sizeof(double) * (size_t)rows_l);
        }
        return e_set_forcing(R.ctx, f.data(), g->fq_cols);
    });
}

int push_dirichlet(Group* g) {
    return run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        if (!g->have_g) return e_set_dirichlet(R.ctx, nullptr);
        std::vector<double> gl(R.gdof.size());
        for (size_t d = 0; d < gl.size(); ++d) gl[d] = g->gdir[(size_t)R.gdof[d]];
        return e_set_dirichlet(R.ctx, gl.data());
    });
}

// the library declined the row-distributed form for this system: the element partition with the neighbour exchange instead -- ranks rebuilt,
// problem data dealt again, matrices assembled again
int change_form(Group* g, int form) {
    if (std::getenv("FDAPDE_DEBUG_SETUP")) std::fprintf(stderr, "multi-device context: changing to the %s form\n", form == kPartitionRowdist ? "row-distributed" : "element-partitioned");
    g->form = form, g->ranks_built = false;
    if (int rc = build_ranks(g)) return rc;
    if (g->have_op)
        if (int rc = push_operator(g, -1, 0)) return rc;
    if (int rc = push_forcing(g)) return rc;
    if (int rc = push_dirichlet(g)) return rc;
    if (g->initialised)
        if (int rc = run_all(g, [&](int r) { return e_init(g->rk[(size_t)r].ctx, g->have_init_opt ? &g->init_opt : nullptr); })) return rc;
    return FDAPDE_OK;
}

void merge_info(Group* g) {
    g->info = g->rk[0].info;
    for (int r = 1; r < g->n; ++r) {
        const fdapde_info& i = g->rk[(size_t)r].info;
        g->info.t_solve_ms = std::max(g->info.t_solve_ms, i.t_solve_ms), g->info.launch_ms = std::max(g->info.launch_ms, i.launch_ms);
        g->info.spmv_avg_ms = std::max(g->info.spmv_avg_ms, i.spmv_avg_ms), g->info.gather_avg_ms = std::max(g->info.gather_avg_ms, i.gather_avg_ms);
        g->info.relres = std::max(g->info.relres, i.relres), g->info.converged = g->info.converged && i.converged;
    }
    g->info.t_setup_ms = g->root->info.t_setup_ms + g->t_partition_ms + g->t_rank_setup_ms;
    g->root->info = g->info;
}

// local pattern entry -> entry of the whole mesh's pattern (both in the reference numbering of their spaces), every local row
int ensure_gslot(Group* g) {
    fdapde_ctx* root = g->root;
    bool need = false;
    for (const GroupRank& R : g->rk) need = need || R.gslot.empty();
    if (!need) return FDAPDE_OK;
    HIPCHK(root, hipSetDevice(root->device));
    if (int rc = ensure_host(root, kHostRefPattern)) return rc;
    const HostSpace& H = root->hs;
    return run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        fdapde_ctx* c = R.ctx;
        if (!R.gslot.empty()) return FDAPDE_OK;
        if (int rc = ensure_host(c, kHostRefPattern)) return rc;
        const HostSpace& L = c->hs;
        R.gslot.resize((size_t)L.nnz);
        for (int64_t i = 0; i < L.n_dofs; ++i) {
            const int32_t gi = R.gdof[(size_t)i];
            const int32_t* gb = H.colidx_e.data() + H.rowptr_e[(size_t)gi];
            const int32_t* ge = H.colidx_e.data() + H.rowptr_e[(size_t)gi + 1];
            for (int32_t k = L.rowptr_e[(size_t)i]; k < L.rowptr_e[(size_t)i + 1]; ++k) {
                const int32_t gj = R.gdof[(size_t)L.colidx_e[(size_t)k]];
                const int32_t* it = std::lower_bound(gb, ge, gj);
                if (it == ge || *it != gj) return fail(c, FDAPDE_EINVAL, "multi-device context: an entry of a rank's pattern is missing from the whole mesh's");
                R.gslot[(size_t)k] = (int64_t)(it - H.colidx_e.data());
            }
        }
        return FDAPDE_OK;
    });
}

// per-DOF results of the ranks into a whole-mesh vector (n_cols columns): owned entries (row-distributed form; solutions in either form), or the
// sum over the ranks (sub-assembled vectors of the element form), rank after rank
int gather_dofs(Group* g, int n_cols, bool sum, const std::function<int(int, double*)>& fetch, double* out) {
    const int64_t nd = g->root->hs.n_dofs;
    const int rc = run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        R.h0.resize(R.gdof.size() * (size_t)n_cols);
        if (int rc2 = fetch(r, R.h0.data())) return rc2;
        if (!sum) {
            const size_t nl = R.gdof.size();
            for (int col = 0; col < n_cols; ++col)
                for (size_t d = 0; d < nl; ++d)
                    if (R.owned[d]) out[(size_t)col * nd + (size_t)R.gdof[d]] = R.h0[(size_t)col * nl + d];   // (every DOF has one owner: disjoint writes)
        }
        return FDAPDE_OK;
    });
    if (rc != FDAPDE_OK) return rc;
    if (sum) {
        std::fill(out, out + (size_t)nd * n_cols, 0.0);
        for (int r = 0; r < g->n; ++r) {
            const GroupRank& R = g->rk[(size_t)r];
            const size_t nl = R.gdof.size();
            for (int col = 0; col < n_cols; ++col)
                for (size_t d = 0; d < nl; ++d) out[(size_t)col * nd + (size_t)R.gdof[d]] += R.h0[(size_t)col * nl + d];
        }
    }
    return FDAPDE_OK;
}

inline Group* group_of(const fdapde_ctx* c) { return c ? c->group : nullptr; }

int need_ranks(Group* g) {
    if (!g->ranks_built) return fail(g->root, FDAPDE_ENOTINIT, "call fdapde_dofs_build first");
    return FDAPDE_OK;
}

}   // namespace

// =====================================================================================================================================
// what capi.hip forwards to for a multi-device context
// =====================================================================================================================================
int g_create(const int32_t* devices, int32_t n, fdapde_ctx** out) {
    if (!devices || n < 1 || n > 64 || !out) return FDAPDE_EINVAL;
    *out = nullptr;
    fdapde_ctx* root = nullptr;
    if (int rc = fdapde_ctx_create(devices[0], &root)) return rc;
    Group* g = new (std::nothrow) Group();
    if (!g) {
        fdapde_ctx_destroy(root);
        return FDAPDE_ENOMEM;
    }
    g->root = root, g->n = n, g->rdv.n = n;
    if (const char* e = std::getenv("FDAPDE_GROUP_FORM")) g->form = std::atoi(e) == 1 ? kPartitionElements : kPartitionRowdist, g->form_locked = true;
    if (const char* e = std::getenv("FDAPDE_GROUP_TIMEOUT_S")) g->rdv.timeout_s = std::max(1, std::atoi(e));
    g->rk.resize((size_t)n);
    std::vector<int> per_dev(64 * 1024, 0);
    for (int r = 0; r < n; ++r) {
        GroupRank& R = g->rk[(size_t)r];
        R.device = devices[r];
        if (int rc = fdapde_ctx_create(devices[r], &R.ctx)) {
            for (int q = 0; q < r; ++q) fdapde_ctx_destroy(g->rk[(size_t)q].ctx);
            delete g;
            fdapde_ctx_destroy(root);
            return rc;
        }
        if (devices[r] >= 0 && devices[r] < (int)per_dev.size()) g->share = std::max(g->share, ++per_dev[(size_t)devices[r]]);
    }
    if (g->share > 1) {   // several ranks on one device: their persistent launches must overlap, i.e. sit on different hardware queues
        const char* q = std::getenv("GPU_MAX_HW_QUEUES");
        if ((!q || std::atoi(q) < g->share + 2) && std::getenv("FDAPDE_DEBUG_SETUP"))
            std::fprintf(stderr, "multi-device context: %d ranks share a device and GPU_MAX_HW_QUEUES is %s: launches that share a hardware queue cannot overlap -- the "
                         "row-distributed form will time out and the context will take the element form (set GPU_MAX_HW_QUEUES >= %d before the process touches the GPU)\n",
                         g->share, q ? q : "unset (4)", g->share + 2);
    }
    for (int r = 0; r < n; ++r) g->rk[(size_t)r].th = std::thread(worker_main, g, r);
    root->group = g;
    *out = root;
    return FDAPDE_OK;
}

void g_destroy(fdapde_ctx* root) {
    Group* g = group_of(root);
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
    }
    g->cv_job.notify_all();
    for (GroupRank& R : g->rk)
        if (R.th.joinable()) R.th.join();
    for (GroupRank& R : g->rk) {
        fdapde_ctx_destroy(R.ctx);
        delete static_cast<RankCookie*>(R.cb_user);
    }
    root->group = nullptr;
    delete g;
}

int g_info(const fdapde_ctx* root, int32_t* n_devices, int32_t* devices, int32_t* form, double* t_partition_ms, double* t_rank_setup_ms) {
    const Group* g = group_of(root);
    if (n_devices) *n_devices = g ? g->n : 1;
    if (devices) {
        if (g)
            for (int r = 0; r < g->n; ++r) devices[r] = g->rk[(size_t)r].device;
        else
            devices[0] = root->device;
    }
    if (form) *form = g ? g->form : -1;
    if (t_partition_ms) *t_partition_ms = g ? g->t_partition_ms : 0.0;
    if (t_rank_setup_ms) *t_rank_setup_ms = g ? g->t_rank_setup_ms : 0.0;
    return FDAPDE_OK;
}

// A multi-device context holding the same problem as `src` (fdapde_ctx_clone): the same devices, the mesh split again -- the partitioner is
// deterministic: the ranks' sub-meshes and spaces come out identical --, the group's host copies of the problem data, and rank by rank the
// assembled / solved state device to device (eng_clone.hip clone_state).
int g_clone(const fdapde_ctx* src_root, fdapde_ctx** out) {
    const Group* sg = group_of(src_root);
    fdapde_ctx* sroot = const_cast<fdapde_ctx*>(src_root);
    std::vector<int32_t> devices((size_t)sg->n);
    for (int r = 0; r < sg->n; ++r) devices[(size_t)r] = sg->rk[(size_t)r].device;
    fdapde_ctx* root = nullptr;
    if (int rc = g_create(devices.data(), sg->n, &root)) return fail(sroot, rc, "fdapde_ctx_clone: creating the multi-device context failed");
    Group* g = root->group;
    auto bail = [&](int rc) {
        sroot->err = "fdapde_ctx_clone: " + root->err;
        fdapde_ctx_destroy(root);
        return rc;
    };
    const HostSpace& hs = src_root->hs;
    if (hs.n_cells == 0) {
        *out = root;
        return FDAPDE_OK;
    }
    if (int rc = fdapde_mesh_upload(root, hs.M, hs.N, hs.n_nodes, hs.nodes.data(), hs.n_cells, hs.cells.data(), hs.node_bnd.data())) return bail(rc);
    if (!src_root->space_ready || !sg->ranks_built) {
        *out = root;
        return FDAPDE_OK;
    }
    g->form = sg->form, g->form_locked = sg->form_locked, g->knobs = sg->knobs;
    {
        const bool keep = g->form_locked;
        g->form_locked = true;   // (g_dofs_build starts from the row-distributed form unless the form is pinned: the clone takes the source's)
        const int rc = g_dofs_build(root, hs.order, nullptr);
        g->form_locked = keep;
        if (rc) return bail(rc);
    }
    if (!sg->bnd_override.empty())
        if (int rc = g_dofs_set_boundary(root, sg->bnd_override.data())) return bail(rc);
    for (int r = 0; r < g->n; ++r)
        if (g->rk[(size_t)r].gdof != sg->rk[(size_t)r].gdof) return bail(fail(root, FDAPDE_EHIP, "the rebuilt split differs from the source's"));
    g->op = sg->op, g->have_op = sg->have_op, g->fq = sg->fq, g->fq_cols = sg->fq_cols, g->gdir = sg->gdir, g->have_g = sg->have_g;
    g->init_opt = sg->init_opt, g->have_init_opt = sg->have_init_opt, g->initialised = sg->initialised, g->info = sg->info, root->info = src_root->info;
    const int rc = run_all(g, [&](int r) { return clone_state(sg->rk[(size_t)r].ctx, g->rk[(size_t)r].ctx); });
    if (rc) return bail(rc);
    *out = root;
    return FDAPDE_OK;
}

void g_mesh_changed(fdapde_ctx* root) {
    Group* g = group_of(root);
    g->ranks_built = false, g->initialised = false, g->have_op = false, g->op.clear(), g->fq.clear(), g->fq_cols = 0, g->have_g = false, g->gdir.clear();
    g->bnd_override.clear();
    if (!g->form_locked) g->form = kPartitionRowdist;
}

int g_dofs_build(fdapde_ctx* root, int order, int64_t* n_dofs) {
    Group* g = group_of(root);
    g->ranks_built = false, g->initialised = false;
    g->bnd_override.clear();
    if (int rc = e_dofs_build(root, order, n_dofs)) return rc;
    g->order = order;
    if (!g->form_locked) g->form = kPartitionRowdist;
    return build_ranks(g);
}

int g_dofs_set_boundary(fdapde_ctx* root, const uint8_t* bnd) {
    Group* g = group_of(root);
    if (int rc = e_dofs_set_boundary(root, bnd)) return rc;
    if (int rc = need_ranks(g)) return rc;
    g->bnd_override.assign(bnd, bnd + root->hs.n_dofs);
    return run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        std::vector<uint8_t> bl(R.gdof.size());
        for (size_t d = 0; d < bl.size(); ++d) bl[d] = bnd[(size_t)R.gdof[d]] ? 1 : 0;
        return e_dofs_set_boundary(R.ctx, bl.data());
    });
}

int g_set_operator(fdapde_ctx* root, int32_t n_terms, const fdapde_term* terms) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (n_terms < 1 || n_terms > kMaxTerms || !terms) return fail(root, FDAPDE_EINVAL, "fdapde_set_operator: 1 .. 8 terms");
    const HostSpace& H = root->hs;
    g->op.assign((size_t)n_terms, StoredTerm{});
    for (int k = 0; k < n_terms; ++k) {
        StoredTerm& s = g->op[(size_t)k];
        s.t = terms[k], s.width = width_of(terms[k].kind, H.N);
        if (terms[k].space_varying) {
            if (!terms[k].data) return fail(root, FDAPDE_EINVAL, "fdapde_set_operator: a space-varying term without data");
            s.data.assign(terms[k].data, terms[k].data + (size_t)H.nq * (size_t)H.n_cells * (size_t)s.width);
        }
        s.t.data = nullptr;
    }
    g->have_op = true, g->initialised = false;
    return push_operator(g, -1, 0);
}

int g_assemble_operator(fdapde_ctx* root, int32_t which, int32_t n_terms, const fdapde_term* terms, int32_t assembly) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (n_terms < 1 || n_terms > kMaxTerms || !terms) return fail(root, FDAPDE_EINVAL, "fdapde_assemble_operator: 1 .. 8 terms");
    const HostSpace& H = root->hs;
    std::vector<StoredTerm> keep;
    keep.swap(g->op);   // (the operator of fdapde_set_operator stays what a change of form deals again)
    g->op.assign((size_t)n_terms, StoredTerm{});
    for (int k = 0; k < n_terms; ++k) {
        StoredTerm& s = g->op[(size_t)k];
        s.t = terms[k], s.width = width_of(terms[k].kind, H.N);
        if (terms[k].space_varying && terms[k].data) s.data.assign(terms[k].data, terms[k].data + (size_t)H.nq * (size_t)H.n_cells * (size_t)s.width);
    }
    const int rc = push_operator(g, which, assembly);
    g->op.swap(keep);
    return rc;
}

int g_set_forcing(fdapde_ctx* root, const double* f_q, int32_t n_cols) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    const HostSpace& H = root->hs;
    if (!f_q || n_cols <= 0) g->fq.clear(), g->fq_cols = 0;
    else g->fq.assign(f_q, f_q + (size_t)H.nq * (size_t)H.n_cells * (size_t)n_cols), g->fq_cols = n_cols;
    return push_forcing(g);
}

int g_set_dirichlet(fdapde_ctx* root, const double* gvals) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    g->have_g = gvals != nullptr;
    if (gvals) g->gdir.assign(gvals, gvals + root->hs.n_dofs);
    else g->gdir.clear();
    return push_dirichlet(g);
}

int g_init(fdapde_ctx* root, const fdapde_options* opt) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    g->have_init_opt = opt != nullptr;
    if (opt) g->init_opt = *opt;
    const int rc = run_all(g, [&](int r) { return e_init(g->rk[(size_t)r].ctx, opt); });
    if (rc != FDAPDE_OK) return rc;
    g->initialised = true;
    double t = 0;
    for (const GroupRank& R : g->rk) t = std::max(t, R.ctx->info.t_assemble_ms);
    root->info.t_assemble_ms = t, g->info.t_assemble_ms = t;
    return FDAPDE_OK;
}

int g_solver_prepare(fdapde_ctx* root, int32_t with_dirichlet) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    return run_all(g, [&](int r) { return e_solver_prepare(g->rk[(size_t)r].ctx, with_dirichlet); });
}

// runs a collective solve on all ranks; if the row-distributed form is declined, once more in the element form
static int solve_with_fallback(Group* g, const std::function<int(int)>& fn) {
    int rc = run_all(g, fn);
    if (rc == FDAPDE_EUNSUPPORTED && g->form == kPartitionRowdist && !g->form_locked) {
        const std::string why = g->root->err;
        if (int rc2 = change_form(g, kPartitionElements)) return rc2;
        rc = run_all(g, fn);
        if (rc == FDAPDE_OK || rc == FDAPDE_ENOCONV) g->root->err = "";
        (void)why;
    }
    return rc;
}

int g_solve(fdapde_ctx* root, const fdapde_options* opt, fdapde_info* info) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!g->initialised) return fail(root, FDAPDE_ENOTINIT, "solver must be initialized first!");
    const int rc = solve_with_fallback(g, [&](int r) { return e_solve(g->rk[(size_t)r].ctx, opt, &g->rk[(size_t)r].info); });
    if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
    merge_info(g);
    if (info) *info = g->info;
    return rc;
}

int g_solution(fdapde_ctx* root, double* solution) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!solution) return FDAPDE_EINVAL;
    return gather_dofs(g, 1, false, [&](int r, double* buf) { return e_solution(g->rk[(size_t)r].ctx, buf); }, solution);
}

int g_force(fdapde_ctx* root, double* force) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!force) return FDAPDE_EINVAL;
    const int ncols = std::max(1, g->fq_cols);
    const bool sum = g->form == kPartitionElements;
    if (int rc = gather_dofs(g, ncols, sum, [&](int r, double* buf) { return e_force(g->rk[(size_t)r].ctx, buf); }, force)) return rc;
    if (sum && g->rk[0].ctx->dirichlet_applied) {   // boundary rows carry g on every rank that holds the DOF: the owner's value, not the sum
        const int64_t nd = root->hs.n_dofs;
        for (int r = 0; r < g->n; ++r) {
            const GroupRank& R = g->rk[(size_t)r];
            const HostSpace& L = R.ctx->hs;
            const size_t nl = R.gdof.size();
            for (int col = 0; col < ncols; ++col)
                for (size_t d = 0; d < nl; ++d)
                    if (R.owned[d] && L.dof_bnd[d]) force[(size_t)col * nd + (size_t)R.gdof[d]] = R.h0[(size_t)col * nl + d];
        }
    }
    return FDAPDE_OK;
}

int g_lump(fdapde_ctx* root, int32_t which, double* diag) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!diag) return FDAPDE_EINVAL;
    return gather_dofs(g, 1, g->form == kPartitionElements, [&](int r, double* buf) { return e_lump(g->rk[(size_t)r].ctx, which, buf); }, diag);
}

int g_matrix_values(fdapde_ctx* root, int32_t which, double* values) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!values) return FDAPDE_EINVAL;
    if (int rc = ensure_gslot(g)) return rc;
    const bool sum = g->form == kPartitionElements;
    const HostSpace& H = root->hs;
    const int rc = run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        const HostSpace& L = R.ctx->hs;
        R.h1.resize((size_t)L.nnz);
        // (the element form sums sub-assembled rows: the row-zeroing of a Dirichlet solve is applied to the sum below, not rank by rank)
        const bool keep = R.ctx->dirichlet_applied;
        if (sum) R.ctx->dirichlet_applied = false;
        const int rc2 = e_matrix_values(R.ctx, which, R.h1.data());
        R.ctx->dirichlet_applied = keep;
        if (rc2) return rc2;
        if (!sum)
            for (int64_t i = 0; i < L.n_dofs; ++i)
                if (R.owned[(size_t)i])
                    for (int32_t k = L.rowptr_e[(size_t)i]; k < L.rowptr_e[(size_t)i + 1]; ++k) values[(size_t)R.gslot[(size_t)k]] = R.h1[(size_t)k];
        return FDAPDE_OK;
    });
    if (rc != FDAPDE_OK) return rc;
    if (sum) {
        std::fill(values, values + (size_t)H.nnz, 0.0);
        for (int r = 0; r < g->n; ++r) {
            const GroupRank& R = g->rk[(size_t)r];
            for (size_t k = 0; k < R.gslot.size(); ++k) values[(size_t)R.gslot[k]] += R.h1[k];
        }
        if (which == FDAPDE_MAT_STIFF && g->rk[0].ctx->dirichlet_applied) {   // set_dirichlet_bc (fem_solver_base.h:142-155): row zeroed, unit diagonal
            const std::vector<uint8_t>& bnd = g->bnd_override.empty() ? H.dof_bnd : g->bnd_override;
            for (int64_t i = 0; i < H.n_dofs; ++i)
                if (bnd[(size_t)i])
                    for (int32_t k = H.rowptr_e[(size_t)i]; k < H.rowptr_e[(size_t)i + 1]; ++k) values[(size_t)k] = H.colidx_e[(size_t)k] == i ? 1.0 : 0.0;
        }
    }
    return FDAPDE_OK;
}

int g_spmv(fdapde_ctx* root, int32_t which, const double* x, double* y) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!x || !y) return FDAPDE_EINVAL;
    return gather_dofs(g, 1, g->form == kPartitionElements, [&](int r, double* buf) -> int {
        GroupRank& R = g->rk[(size_t)r];
        R.h1.resize(R.gdof.size());
        for (size_t d = 0; d < R.gdof.size(); ++d) R.h1[d] = x[(size_t)R.gdof[d]];
        return e_spmv(R.ctx, which, R.h1.data(), buf);
    }, y);
}

int g_solve_parabolic(fdapde_ctx* root, const fdapde_options* opt, int32_t n_times, double delta_t, const double* initial_condition, const double* dirichlet,
                      double* solution, fdapde_info* info) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!g->initialised) return fail(root, FDAPDE_ENOTINIT, "solver must be initialized first!");
    if (n_times < 2 || !initial_condition || !solution) return FDAPDE_EINVAL;
    const int64_t nd = root->hs.n_dofs;
    std::vector<std::vector<double>> sol((size_t)g->n);
    const int rc = solve_with_fallback(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        const size_t nl = R.gdof.size();
        std::vector<double> ic(nl), dir;
        for (size_t d = 0; d < nl; ++d) ic[d] = initial_condition[(size_t)R.gdof[d]];
        if (dirichlet) {
            dir.resize(nl * (size_t)n_times);
            for (int t = 0; t < n_times; ++t)
                for (size_t d = 0; d < nl; ++d) dir[(size_t)t * nl + d] = dirichlet[(size_t)t * nd + (size_t)R.gdof[d]];
        }
        sol[(size_t)r].assign(nl * (size_t)n_times, 0.0);
        return e_solve_parabolic(R.ctx, opt, n_times, delta_t, ic.data(), dirichlet ? dir.data() : nullptr, sol[(size_t)r].data(), &R.info);
    });
    if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
    for (int r = 0; r < g->n; ++r) {
        const GroupRank& R = g->rk[(size_t)r];
        const size_t nl = R.gdof.size();
        for (int t = 0; t < n_times; ++t)
            for (size_t d = 0; d < nl; ++d)
                if (R.owned[d]) solution[(size_t)t * nd + (size_t)R.gdof[d]] = sol[(size_t)r][(size_t)t * nl + d];
    }
    merge_info(g);
    if (info) *info = g->info;
    return rc;
}

int g_lin_compute(fdapde_ctx* root, int32_t which, const double* values, int32_t symmetric) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!values) return run_all(g, [&](int r) { return e_lin_compute(g->rk[(size_t)r].ctx, which, nullptr, symmetric); });
    if (int rc = ensure_gslot(g)) return rc;
    const bool elem = g->form == kPartitionElements;
    if (elem && g->nnz_rank.empty()) {   // a sub-assembled split of the matrix: every entry on the lowest rank whose pattern has it
        g->nnz_rank.assign((size_t)root->hs.nnz, 255);
        for (int r = g->n - 1; r >= 0; --r)
            for (int64_t k : g->rk[(size_t)r].gslot) g->nnz_rank[(size_t)k] = (uint8_t)r;
    }
    return run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        R.h1.resize(R.gslot.size());
        for (size_t k = 0; k < R.gslot.size(); ++k)
            R.h1[k] = (!elem || g->nnz_rank[(size_t)R.gslot[k]] == r) ? values[(size_t)R.gslot[k]] : 0.0;
        return e_lin_compute(R.ctx, which, R.h1.data(), symmetric);
    });
}

int g_lin_solve(fdapde_ctx* root, const fdapde_options* opt, const double* b, int32_t n_rhs, double* x, fdapde_info* info) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    if (!b || !x || n_rhs < 1) return FDAPDE_EINVAL;
    const int64_t nd = root->hs.n_dofs;
    const bool elem = g->form == kPartitionElements;
    std::vector<double> b_copy;
    if (b < x + (size_t)nd * n_rhs && x < b + (size_t)nd * n_rhs) b_copy.assign(b, b + (size_t)nd * n_rhs), b = b_copy.data();   // (in-place solve)
    std::vector<std::vector<double>> xs((size_t)g->n);
    const int rc = run_all(g, [&](int r) -> int {
        GroupRank& R = g->rk[(size_t)r];
        const size_t nl = R.gdof.size();
        std::vector<double> bl(nl * (size_t)n_rhs);
        // row-distributed form: complete at the owned DOFs (the others are not read); element form: sub-assembled -- the owner carries the entry
        for (int j = 0; j < n_rhs; ++j)
            for (size_t d = 0; d < nl; ++d) bl[(size_t)j * nl + d] = (!elem || R.owned[d]) ? b[(size_t)j * nd + (size_t)R.gdof[d]] : 0.0;
        xs[(size_t)r].assign(nl * (size_t)n_rhs, 0.0);
        return e_lin_solve(R.ctx, opt, bl.data(), n_rhs, xs[(size_t)r].data(), &R.info);
    });
    if (rc != FDAPDE_OK && rc != FDAPDE_ENOCONV) return rc;
    for (int r = 0; r < g->n; ++r) {
        const GroupRank& R = g->rk[(size_t)r];
        const size_t nl = R.gdof.size();
        for (int j = 0; j < n_rhs; ++j)
            for (size_t d = 0; d < nl; ++d)
                if (R.owned[d]) x[(size_t)j * nd + (size_t)R.gdof[d]] = xs[(size_t)r][(size_t)j * nl + d];
    }
    merge_info(g);
    if (info) *info = g->info;
    return rc;
}

int g_tune(fdapde_ctx* root, const char* key, int32_t value) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    const std::string k(key);
    if (k == "group_direct" && (value == 0 || value == 1)) {   // (measurements: the element form's exchange direct or host-staged)
        g->direct = value != 0;
        for (GroupRank& R : g->rk) R.ctx->xd.on = g->direct && g->form == kPartitionElements && R.ctx->xd.slot_ptr.p != nullptr;
        return FDAPDE_OK;
    }
    if (k == "group_form" && (value == 0 || value == 1)) {   // tests / measurements: the form, fixed
        g->form_locked = true;
        if (value != g->form) return change_form(g, value);
        return FDAPDE_OK;
    }
    for (auto& kv : g->knobs)
        if (kv.first == k) kv.second = value;
    if (std::find_if(g->knobs.begin(), g->knobs.end(), [&](const auto& kv) { return kv.first == k; }) == g->knobs.end()) g->knobs.push_back({k, value});
    return run_all(g, [&](int r) { return fdapde_tune(g->rk[(size_t)r].ctx, key, value); });
}

int g_synchronize(fdapde_ctx* root) {
    Group* g = group_of(root);
    if (!g->ranks_built) return FDAPDE_OK;
    return run_all(g, [&](int r) { return fdapde_synchronize(g->rk[(size_t)r].ctx); });
}

int g_layout_kind(fdapde_ctx* root, int32_t with_dirichlet, int32_t* kind, int32_t* symmetric_storage, int32_t* workgroups, int32_t* rows_per_thread) {
    Group* g = group_of(root);
    if (int rc = need_ranks(g)) return rc;
    return fail(root, FDAPDE_EUNSUPPORTED, "fdapde_solver_layout*: per-device layouts of a multi-device context are not exposed (fdapde_info: persistent, launch_ms)");
    (void)with_dirichlet, (void)kind, (void)symmetric_storage, (void)workgroups, (void)rows_per_thread;
}

// =====================================================================================================================================
// the partitioner's public face (single-device contexts: rank processes partition the mesh they all generate, each on its own device)
// =====================================================================================================================================
int e_partition_build(fdapde_ctx* c, int32_t world, int32_t form) {
    if (!c) return FDAPDE_EINVAL;
    if (int rc = need_device(c)) return rc;
    if (c->hs.n_cells == 0) return fail(c, FDAPDE_ENOTINIT, "call fdapde_mesh_upload first");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->mesh_on_dev) {
        const HostSpace& hs = c->hs;
        HIPCHK(c, c->mesh_nodes.upload(hs.nodes.data(), hs.nodes.size(), c->stream));
        HIPCHK(c, c->mesh_cells.upload(hs.cells.data(), hs.cells.size(), c->stream));
        HIPCHK(c, c->mesh_nbnd.upload(hs.node_bnd.data(), hs.node_bnd.size(), c->stream));
        c->mesh_on_dev = true;
    }
    if (!c->partition) c->partition = new (std::nothrow) DevPartition();
    if (!c->partition) return FDAPDE_ENOMEM;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = dev_partition_build(c->hs.M, c->hs.N, c->hs.n_nodes, c->hs.n_cells, c->mesh_nodes.p, c->mesh_cells.p, c->mesh_nbnd.p, world, form, c->stream, c->partition, c->err);
    if (rc != FDAPDE_OK) dev_partition_release(c->partition);
    c->partition_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

static int partition_rank(const fdapde_ctx* c, int32_t rank, const RankMeshDev** out) {
    if (!c || !c->partition || c->partition->world == 0) return FDAPDE_ENOTINIT;
    if (rank < 0 || rank >= c->partition->world) return FDAPDE_EINVAL;
    *out = &c->partition->ranks[(size_t)rank];
    return FDAPDE_OK;
}

int e_partition_sizes(const fdapde_ctx* c, int32_t rank, int64_t* n_nodes, int64_t* n_cells) {
    const RankMeshDev* R = nullptr;
    if (int rc = partition_rank(c, rank, &R)) return rc;
    if (n_nodes) *n_nodes = R->n_nodes;
    if (n_cells) *n_cells = R->n_cells;
    return FDAPDE_OK;
}

int e_partition_get(fdapde_ctx* c, int32_t rank, double* nodes_colmajor, int32_t* cells, uint8_t* boundary, int64_t* node_ids, int64_t* cell_ids, int32_t* node_owner) {
    const RankMeshDev* R = nullptr;
    if (int rc = partition_rank(c, rank, &R)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const int nv = c->partition->M + 1, N = c->partition->N;
    if (nodes_colmajor) HIPCHK(c, hipMemcpy(nodes_colmajor, R->nodes, sizeof(double) * (size_t)R->n_nodes * N, hipMemcpyDeviceToHost));
    if (cells) HIPCHK(c, hipMemcpy(cells, R->cells, sizeof(int32_t) * (size_t)R->n_cells * nv, hipMemcpyDeviceToHost));
    if (boundary) HIPCHK(c, hipMemcpy(boundary, R->bnd, (size_t)R->n_nodes, hipMemcpyDeviceToHost));
    if (node_owner) HIPCHK(c, hipMemcpy(node_owner, R->node_owner, sizeof(int32_t) * (size_t)R->n_nodes, hipMemcpyDeviceToHost));
    std::vector<int32_t> tmp;
    if (node_ids) {
        tmp.resize((size_t)R->n_nodes);
        HIPCHK(c, hipMemcpy(tmp.data(), R->l2g, sizeof(int32_t) * tmp.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) node_ids[i] = tmp[i];
    }
    if (cell_ids) {
        tmp.resize((size_t)R->n_cells);
        HIPCHK(c, hipMemcpy(tmp.data(), R->cell_ids, sizeof(int32_t) * tmp.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) cell_ids[i] = tmp[i];
    }
    return FDAPDE_OK;
}

int e_partition_whole(fdapde_ctx* c, int32_t* cell_rank, int32_t* node_owner, uint64_t* node_ranks) {
    if (!c || !c->partition || c->partition->world == 0) return FDAPDE_ENOTINIT;
    HIPCHK(c, hipSetDevice(c->device));
    const DevPartition& P = *c->partition;
    if (cell_rank) HIPCHK(c, hipMemcpy(cell_rank, P.part, sizeof(int32_t) * (size_t)P.n_cells, hipMemcpyDeviceToHost));
    if (node_owner) HIPCHK(c, hipMemcpy(node_owner, P.node_owner, sizeof(int32_t) * (size_t)P.n_nodes, hipMemcpyDeviceToHost));
    if (node_ranks) HIPCHK(c, hipMemcpy(node_ranks, P.node_mask, sizeof(uint64_t) * (size_t)P.n_nodes, hipMemcpyDeviceToHost));
    return FDAPDE_OK;
}

// element form, P1 (DOF = node): the neighbour lists of fdapde_halo_setup_peers from the node masks -- peers ascending, the nodes shared with a
// peer in ascending global id on both sides, owned = this rank is the lowest one holding the node
int e_partition_peers(fdapde_ctx* c, int32_t rank, int32_t* n_peers, int32_t* peer_rank, int64_t* peer_off, int32_t* peer_node, uint8_t* owned, int64_t* n_shared) {
    const RankMeshDev* R = nullptr;
    if (int rc = partition_rank(c, rank, &R)) return rc;
    if (c->partition->form != kPartitionElements) return fail(c, FDAPDE_EINVAL, "fdapde_partition_peers: the partition was built in the row-distributed form");
    HIPCHK(c, hipSetDevice(c->device));
    const DevPartition& P = *c->partition;
    if (c->partition_mask_h.size() != (size_t)P.n_nodes) {
        c->partition_mask_h.resize((size_t)P.n_nodes);
        HIPCHK(c, hipMemcpy(c->partition_mask_h.data(), P.node_mask, sizeof(uint64_t) * (size_t)P.n_nodes, hipMemcpyDeviceToHost));
    }
    std::vector<int32_t> l2g((size_t)R->n_nodes);
    HIPCHK(c, hipMemcpy(l2g.data(), R->l2g, sizeof(int32_t) * l2g.size(), hipMemcpyDeviceToHost));
    const std::vector<uint64_t>& mask = c->partition_mask_h;
    int32_t np = 0;
    int64_t at = 0;
    if (peer_off) peer_off[0] = 0;
    for (int q = 0; q < P.world; ++q) {
        if (q == rank) continue;
        const int64_t before = at;
        for (size_t i = 0; i < l2g.size(); ++i)   // (local ids ascend with the global ones)
            if (mask[(size_t)l2g[i]] >> q & 1) {
                if (peer_node) peer_node[at] = (int32_t)i;
                ++at;
            }
        if (at > before) {
            if (peer_rank) peer_rank[np] = q;
            if (peer_off) peer_off[np + 1] = at;
            ++np;
        }
    }
    if (owned)
        for (size_t i = 0; i < l2g.size(); ++i) {
            const uint64_t m = mask[(size_t)l2g[i]];
            owned[i] = (m & (~m + 1)) == (uint64_t(1) << rank) ? 1 : 0;
        }
    if (n_peers) *n_peers = np;
    if (n_shared) *n_shared = at;
    return FDAPDE_OK;
}

void partition_free(fdapde_ctx* c) {
    if (!c->partition) return;
    dev_partition_release(c->partition);
    delete c->partition;
    c->partition = nullptr;
    c->partition_mask_h.clear();
}

}   // namespace fdapde_engine
