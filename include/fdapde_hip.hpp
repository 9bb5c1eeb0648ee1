// fdapde_hip.hpp -- the one piece of C++ both host-side bindings of the C ABI share: a copyable owner of a fdapde_ctx.
//
// The reference hands PDEs around BY VALUE: make_pde copies the PDE -- solver, matrices and all -- into an
// erase<heap_storage, PDE__> handle, and every copy of that handle copies it again (fdaPDE/pde/pde.h:167-169,
// fdaPDE/utils/type_erasure.h:124-146: `new T(obj)` on construction, copy(ptr) on handle copy); fdapde::SparseLU makes its Eigen solver
// copyable the same way, behind a shared_ptr (fdaPDE/utils/symbols.h:133-160).  A solver object that owns a device context has to be
// copyable too, and a copy must not cost a set-up unless it is used to compute something different.  Hence copy-on-write:
//
//   * copies SHARE the context (a copy is a reference count);
//   * get()    -- for calls that read (getters, basis evaluation, quadrature nodes): whatever context the object currently shares;
//   * unique() -- before any call that changes the context's problem state (set_* / init / solve / lin_compute): if another object
//                 still shares the context, this one leaves with a clone of it (fdapde_ctx_clone: same mesh, space, problem data,
//                 assembled matrices and solution), so the other sharers keep seeing exactly the state they were copied with.
//
// Invariant: a context's state is only ever changed by an object that is its sole owner, so it is consistent with every object
// sharing it.  make_pde's temporary -> heap copy -> temporary destroyed therefore costs no device work at all.
// Like the reference's objects, none of this is thread-safe; distinct PDE objects on distinct contexts are independent.
#ifndef FDAPDE_HIP_HPP
#define FDAPDE_HIP_HPP

#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <utility>

#include "fdapde_hip.h"

namespace fdapde {
namespace hip {

// The devices a solver plugged into the reference's selector computes on.  The reference constructs its solver as SolverType(domain)
// (fdaPDE/pde/pde.h:58): there is no argument through which a caller could name devices, so the binding takes them from here -- set once per process
// (set_default_devices({0, 1, 2, 3}): the mesh of every PDE made afterwards is sharded over these GPUs behind the unchanged PDE<> interface) or from
// the environment (FDAPDE_HIP_DEVICES=0,1,2,3); default: device 0.
inline std::vector<int>& default_devices() {
    static std::vector<int> devices = [] {
        std::vector<int> d;
        if (const char* e = std::getenv("FDAPDE_HIP_DEVICES")) {
            int v = 0;
            bool have = false;
            for (const char* p = e;; ++p) {
                if (*p >= '0' && *p <= '9') v = 10 * v + (*p - '0'), have = true;
                else {
                    if (have) d.push_back(v);
                    v = 0, have = false;
                    if (!*p) break;
                }
            }
        }
        if (d.empty()) d.push_back(0);
        return d;
    }();
    return devices;
}
inline void set_default_devices(std::vector<int> devices) {
    if (!devices.empty()) default_devices() = std::move(devices);
}

class context_handle {
   public:
    context_handle() = default;
    // a fresh context on `device` (throws: the product has no CPU fallback)
    explicit context_handle(int device) : s_(std::make_shared<state>()) {
        const int rc = fdapde_ctx_create(device, &s_->ctx);
        if (rc != FDAPDE_OK) throw std::runtime_error(std::string("fdapde_ctx_create: ") + fdapde_status_string(rc));
        s_->owners = 1;
    }
    // ONE context over several devices (fdapde_ctx_create_multi): used exactly like a single-device one; a copy that diverges is cloned onto the same
    // devices
    explicit context_handle(const std::vector<int>& devices) : s_(std::make_shared<state>()) {
        std::vector<int32_t> d(devices.begin(), devices.end());
        const int rc = d.size() == 1 ? fdapde_ctx_create(d[0], &s_->ctx) : fdapde_ctx_create_multi(d.data(), (int32_t)d.size(), &s_->ctx);
        if (rc != FDAPDE_OK) throw std::runtime_error(std::string("fdapde_ctx_create_multi: ") + fdapde_status_string(rc));
        s_->owners = 1;
    }
    context_handle(const context_handle& other) : s_(other.s_), counted_(other.counted_) {
        if (s_ && counted_) ++s_->owners;
    }
    context_handle(context_handle&& other) noexcept : s_(std::move(other.s_)), counted_(other.counted_) { other.s_.reset(); }
    context_handle& operator=(const context_handle& other) {
        if (this != &other) {
            context_handle tmp(other);
            swap(tmp);
        }
        return *this;
    }
    context_handle& operator=(context_handle&& other) noexcept {
        if (this != &other) {
            drop();
            s_ = std::move(other.s_), counted_ = other.counted_;
            other.s_.reset();
        }
        return *this;
    }
    ~context_handle() { drop(); }
    void swap(context_handle& other) noexcept { std::swap(s_, other.s_), std::swap(counted_, other.counted_); }

    // the context as it is: reads only
    fdapde_ctx* get() const { return s_ ? s_->ctx : nullptr; }
    // the context for a call that changes its problem state: cloned first if another owner still shares it
    fdapde_ctx* unique() {
        if (!s_) throw std::runtime_error("fdapde::hip::context_handle: no context");
        if (counted_ && s_->owners > 1) {
            auto mine = std::make_shared<state>();
            const int rc = fdapde_ctx_clone(s_->ctx, &mine->ctx);
            if (rc != FDAPDE_OK) throw std::runtime_error(std::string("fdapde_ctx_clone: ") + fdapde_last_error(s_->ctx));
            mine->owners = 1;
            --s_->owners;
            s_ = std::move(mine);
        }
        return s_->ctx;
    }
    bool shared() const { return s_ && s_->owners > 1; }
    // a handle that keeps the context alive and follows none of the copy-on-write rules: for helper objects that act ON the owner's
    // context (the factor-once solver handle a PDE gives out).
    // LIFETIME RULE: an observer is bound to the context its owner held WHEN IT WAS TAKEN.  If the owner is copied afterwards and
    // then changes its problem state, the owner leaves with a clone (copy-on-write) and the observer keeps acting on the context that now
    // belongs to the copy alone.  Take the observer after the last copy of the owner, or from the object that is kept; an observer never
    // outlives its usefulness silently -- shared_with(owner) tells whether the two still mean the same context.
    context_handle observer() const {
        context_handle h;
        h.s_ = s_, h.counted_ = false;
        return h;
    }
    bool shared_with(const context_handle& other) const { return s_ && s_ == other.s_; }
    explicit operator bool() const { return s_ && s_->ctx; }

   private:
    struct state {
        fdapde_ctx* ctx = nullptr;
        int owners = 0;   // objects whose problem state this context holds (observers are not counted)
        state() = default;
        state(const state&) = delete;
        state& operator=(const state&) = delete;
        ~state() { fdapde_ctx_destroy(ctx); }
    };
    void drop() {
        if (s_ && counted_) --s_->owners;
        s_.reset();
    }
    std::shared_ptr<state> s_;
    bool counted_ = true;
};

}   // namespace hip
}   // namespace fdapde

#endif   // FDAPDE_HIP_HPP
