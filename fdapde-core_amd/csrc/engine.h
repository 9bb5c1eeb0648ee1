// engine.h -- what the translation units of libfdapde_hip.so share besides the context: small helpers and the entry points of the
// internal engines.  capi.hip holds the C ABI and the multi-launch solve driver; persist_engine.hip the single-launch solver
// (layouts, launches, the row-distributed multi-GPU form).  Internal to the library.
#ifndef FDAPDE_ENGINE_H
#define FDAPDE_ENGINE_H

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "context.h"

namespace fdapde_engine {

// FDAPDE_DEBUG_TIMING=1: wall-clock marks of the host side of a solve on stderr (where a first solve spends its time)
struct DebugClock {
    bool on = std::getenv("FDAPDE_DEBUG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[timing] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

inline int fail(fdapde_ctx* c, int code, const char* msg) {
    c->err = msg;
    return code;
}
inline int need_device(fdapde_ctx* c) {
    if (!c->has_device) return fail(c, FDAPDE_ENODEVICE, "this context has no HIP device (host-only); there is no CPU fallback");
    return FDAPDE_OK;
}

// a DBuf takes over a device array built elsewhere (dev_setup.hip)
template <typename T> void adopt(DBuf<T>& b, T*& p, size_t n) {
    b.release();
    b.p = p, b.n = n, p = nullptr;
}


inline unsigned grid1(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// big host-side index arrays of a device-built space, fetched the first time host code needs them (capi.hip)
enum { kHostPattern = 1, kHostCells = 2, kHostRefPattern = 4, kHostDofs = 8 };
int ensure_host(fdapde_ctx* c, int what);

// ---- single-launch solver (persist_engine.hip) -----------------------------------------------------------------------------------
// layout of boundary variant v, built on first use (ps.tried / ps.ok tell the outcome)
int build_persist(fdapde_ctx* c, int v);
// scaled values into the layout's blocks (after the scaled full-pattern matrix c->sval is in place)
int fill_persist(fdapde_ctx* c, int v);
// the whole fused-update CG as one launch; *ran = false: the launch gave up (hand-off timeout) or can never be resident
int run_persist(fdapde_ctx* c, int v, double tol2, int maxit, bool* ran, bool bicg = false);   // bicg: the BiCGStab kernel (plain layouts, <= 8 rows per thread)

// ---- row-distributed multi-GPU form of the single-launch solver (persist_engine.hip); all COLLECTIVE over the context's ranks
int build_rowdist(fdapde_ctx* c, int v);
int rowdist_import_ghosts(fdapde_ctx* c, int v, double* vec);   // owners' values of a per-DOF vector into this rank's ghost columns
int fill_rowdist(fdapde_ctx* c, int v);
int run_rowdist(fdapde_ctx* c, int v, double tol2, int maxit, bool* ran, bool bicg = false);
void release_rowdist(fdapde_ctx* c);

}   // namespace fdapde_engine

#endif
