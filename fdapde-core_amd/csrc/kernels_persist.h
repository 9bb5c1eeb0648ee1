// kernels_persist.h -- the whole Jacobi-PCG solve of a SMALL system as ONE launch (fdapde_solve / fdapde_lin_solve /
// the parabolic stepper, single GPU, symmetric positive operator): the in-solve SpMV of matrices of a few tens of MB is bound by
// launch and hand-off latency, not by bandwidth (C2: 17 us per SpMV launch + 6 us update + two dependent launches = 27-29 us per
// iteration for 40 MB of matrix), so the iteration is restructured around what the chip can keep ON the CUs:
//   * one workgroup per CU, each owning a contiguous range of interior rows (locality numbering => few neighbours);
//   * its slice of the scaled matrix RESIDENT IN LDS as sliced ELL (8-byte value + 16-bit column code per entry; 256 CUs x 160 KB
//     = 40 MB), x / r / p of its rows in registers; slices that do not fit stream from global memory (they stay in the L2s);
//   * neighbours exchange the entries of p they need through 8-byte {epoch, payload} granules (the data is the flag: no barrier,
//     no fence, MI355X_MICROARCH.md "handoff-1to1"), rows that need no import are multiplied while the granules travel;
//   * the three dot products of an iteration (p.Ap, Ap.Ap, r.r) cross in ONE all-gather of tagged granules that every workgroup
//     sums in the same fixed order (deterministic, bitwise identical scalars everywhere; "allgather" row of the same price list);
//   * the recurrence is the fused-update CG of k_cgf_update (alpha from the explicit r.r, beta from alpha^2 Ap.Ap - r.r), so the
//     iteration counts equal the multi-launch path's; the stop test runs on the device, the host never polls.
// Every spin is bounded: a workgroup that waits too long (a peer not resident, e.g. other work on the device) raises ctl[3] and
// the host re-runs the solve through the multi-launch path.
#ifndef FDAPDE_KERNELS_PERSIST_H
#define FDAPDE_KERNELS_PERSIST_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "internal.h"

namespace fdapde_hip {

struct PersistArgs {
    int32_t G, nsl, maxit, imp_cap;   // workgroups; slices per workgroup; iteration bound; import slots reserved in LDS
    int32_t lds_cap;                  // ELL entries staged in LDS by the resident form (multiple of 128; >= the largest block)
    int32_t time_phases;              // != 0: every workgroup accumulates its phase durations into stats
    int32_t gather_waves, poll_sleep; // all-gather of the dot records: polling wavefronts (4 | 1) and the pause between polls (0 none .. 3 long)
    double tol2;
    const int32_t* slot_dof;
    const int64_t* ell_off;
    const int32_t* sl_off;
    const uint16_t* ell_code;
    const double* ell_val;
    const int32_t* exp_off;
    const uint16_t* exp_slot;
    const int32_t* imp_off;
    const int32_t* imp_pos;
    unsigned long long* pboard;   // 2 granules per exported entry
    unsigned long long* dboard;   // [parity][workgroup][3 values][2 granules]
    const double* r_in;           // initial residual (= initial direction), internal DOF order
    double* x;                    // in: initial guess, out: solution (scaled unknowns), internal DOF order
    double* sc;                   // sc[0] = reference norm^2 (in); sc[3] = final r.r (out)
    int32_t* ctl;                 // out: [0] converged, [1] iterations, [2] breakdown, [3] hand-off timeout
    double* stats;                // per workgroup: [0] iterations timed, [1] operator phase (SpMV + imports), [2] all-gather phase (incl. the
                                  // wait for the slowest workgroup), [3] update phase -- sums of 10 ns ticks
};

typedef __attribute__((address_space(1))) unsigned long long pg_u64;

__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned epoch, unsigned payload) {
    __hip_atomic_store((pg_u64*)p, ((unsigned long long)epoch << 32) | payload, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);   // one aligned 8-byte write-through (sc1) store
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load((const pg_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: bypasses this CU's L1
}
// both granules of a published double with ONE 16-byte sc1 load (each aligned 8-byte half is a granule of its own; a half is never torn)
typedef unsigned long long pg_v2u64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pg_v2u64 granule_load2(const unsigned long long* p) {
    pg_v2u64 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void granule_load6(const unsigned long long* p, pg_v2u64& a, pg_v2u64& b, pg_v2u64& c) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p) : "memory");
}
__device__ __forceinline__ bool granule_pair_ok(pg_v2u64 v, unsigned epoch) {
    return (unsigned)(v.x >> 32) == epoch && (unsigned)(v.y >> 32) == epoch;
}
__device__ __forceinline__ double granule_pair_f64(pg_v2u64 v) {
    return __longlong_as_double((long long)((v.y << 32) | (v.x & 0xffffffffull)));
}
__device__ __forceinline__ void publish_f64(unsigned long long* g2, unsigned epoch, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    granule_store(g2, epoch, (unsigned)b);
    granule_store(g2 + 1, epoch, (unsigned)(b >> 32));
}
__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr long long kPersistTimeoutTicks = 5000000;   // 50 ms of s_memrealtime (100 MHz): a legitimate wait is an iteration's skew, tens of us

// R rows per thread (2, 4, 8, 16); passes [0, R / 2) hold rows that import nothing, passes [R / 2, R) the others (host_persist.cpp).
// STREAM = false: the workgroup's whole block of the matrix is staged in LDS once and re-read from there every iteration.
// STREAM = true : the block streams from global memory every iteration (it is larger than the LDS: systems of a few hundred MB).
//                 x, r, p still never leave the registers, so an iteration moves the matrix and the exchanged entries of p and
//                 nothing else -- the multi-launch path moves 7 vector passes on top.  All R / 2 value loads of an entry step are
//                 issued before the first is used (unconditional loads, clamped to the slice: narrower slices re-read their last
//                 pair row, which multiplies by zero).
template <int R, bool STREAM>
__global__ __launch_bounds__(kPersistT) void k_cg_persist(PersistArgs a) {
    constexpr int T = kPersistT, W = T / 64, S = R * T, RI = R / 2;
    extern __shared__ double lds[];
    __shared__ double red[W][3];
    __shared__ double tot[3];
    __shared__ int32_t fail_flag;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsl = a.nsl;
    const int H = a.imp_off[g + 1] - a.imp_off[g], E = a.exp_off[g + 1] - a.exp_off[g];
    double* p_tab = lds;                                                                  // [S + imp_cap]
    double2* ev = reinterpret_cast<double2*>(p_tab + (S + a.imp_cap));                    // [lds_cap / 2] entry pairs (resident form)
    uint32_t* ec = reinterpret_cast<uint32_t*>(ev + a.lds_cap / 2);                       // [lds_cap / 2] code pairs
    int32_t* impl = reinterpret_cast<int32_t*>(ec + a.lds_cap / 2);                       // [imp_cap]
    uint16_t* expl = reinterpret_cast<uint16_t*>(impl + a.imp_cap);                       // [E]
    const int64_t e0 = a.ell_off[g];
    const double2* gv = reinterpret_cast<const double2*>(a.ell_val + e0);
    const uint32_t* gc = reinterpret_cast<const uint32_t*>(a.ell_code + e0);
    // ---- stage the workgroup's tables (and, resident form, its block of the matrix)
    for (int i = tid; i < H; i += T) impl[i] = a.imp_pos[a.imp_off[g] + i];
    for (int i = tid; i < E; i += T) expl[i] = a.exp_slot[a.exp_off[g] + i];
    if (tid == 0) fail_flag = 0;
    const int32_t* slo = a.sl_off + (size_t)g * (nsl + 1);
    if constexpr (!STREAM) {
        const int n_pairs = slo[nsl] * 64;
        for (int i = tid; i < n_pairs; i += T) ev[i] = gv[i], ec[i] = gc[i];
    }
    // slices of this wavefront: pass j -> slice j * W + wave, pair rows [o0[j], o0[j] + w[j]) (wave-uniform: scalar registers)
    int o0[R], w[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        o0[j] = __builtin_amdgcn_readfirstlane(slo[j * W + wave]);
        w[j] = __builtin_amdgcn_readfirstlane(slo[j * W + wave + 1]) - o0[j];
    }
    // ---- the rows of this thread: slot j * T + tid, j < R
    double xv[R], rv[R], pv[R];
    int32_t dof[R];
    double rr_part = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        dof[j] = a.slot_dof[(size_t)g * S + j * T + tid];
        const bool on = dof[j] >= 0;
        rv[j] = on ? a.r_in[dof[j]] : 0.0;
        xv[j] = on ? a.x[dof[j]] : 0.0;
        pv[j] = rv[j];
        rr_part += rv[j] * rv[j];
    }
    const double bb = a.sc[0];
    const bool stamper = a.time_phases && tid == 0;   // every workgroup stamps its own phases (a few s_memrealtime per iteration)
    long long t_spmv = 0, t_gather = 0, t_update = 0, n_stamped = 0;
    int it = 0, status = 0;   // status: 1 converged, 2 breakdown, 3 hand-off timeout
    double rr = 0;
    __syncthreads();
    for (;;) {
        const unsigned epoch = (unsigned)it + 1u;
        long long c0 = 0, c1 = 0, c2 = 0;
        if (stamper) c0 = wall_clock64();
        // ---- p of the own rows into the LDS table; exported entries onto the board
#pragma unroll
        for (int j = 0; j < R; ++j) p_tab[j * T + tid] = pv[j];
        __syncthreads();
        for (int i = tid; i < E; i += T) publish_f64(a.pboard + 2 * (size_t)(a.exp_off[g] + i), epoch, p_tab[expl[i]]);
        // ---- y = (I + At_offdiag) p: the passes without imports first, the others once the neighbours' entries have arrived
        double yv[R];
#pragma unroll
        for (int j = 0; j < R; ++j) yv[j] = pv[j];   // unit diagonal of the scaled system
        auto product = [&](auto phase) {
            constexpr int J0 = decltype(phase)::value ? RI : 0, J1 = decltype(phase)::value ? R : RI;
            int mw = 0;
#pragma unroll
            for (int j = J0; j < J1; ++j) mw = max(mw, w[j]);
            for (int e = 0; e < mw; ++e) {
                double2 v[J1 - J0];
                uint32_t c[J1 - J0];
#pragma unroll
                for (int j = J0; j < J1; ++j) {
                    const int ee = min(e, max(w[j] - 1, 0));
                    const int idx = (o0[j] + ee) * 64 + lane;
                    if constexpr (STREAM) v[j - J0] = gv[idx], c[j - J0] = gc[idx];
                    else v[j - J0] = ev[idx], c[j - J0] = ec[idx];
                }
#pragma unroll
                for (int j = J0; j < J1; ++j) {
                    const double t = v[j - J0].x * p_tab[c[j - J0] & 0xffffu] + v[j - J0].y * p_tab[c[j - J0] >> 16];
                    yv[j] += e < w[j] ? t : 0.0;
                }
            }
        };
        product(std::integral_constant<int, 0>{});
        {   // imports: a lane re-reads its granule pair until both halves carry this iteration's tag; lanes that have theirs stop loading
            bool fail = false;
            for (int h0 = wave * 64; h0 < H; h0 += T) {   // wave-uniform trip count
                const int h = h0 + lane;
                const unsigned long long* gp = a.pboard + 2 * (size_t)(h < H ? impl[h] : impl[0]);
                pg_v2u64 v = {0, 0};
                bool done = h >= H;
                long long t_wait = 0;
                for (unsigned spins = 0;; ++spins) {
                    if (!done) v = granule_load2(gp), done = granule_pair_ok(v, epoch);
                    if (__all(done)) break;
                    if ((spins & 63u) == 63u) {   // bounded by time, checked now and then
                        const long long now = wall_clock64();
                        if (t_wait == 0) t_wait = now;
                        else if (now - t_wait > kPersistTimeoutTicks) {
                            fail = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (h < H) p_tab[S + h] = granule_pair_f64(v);
                if (fail) break;
            }
            if (fail && lane == 0) fail_flag = 1;
        }
        __syncthreads();
        if (fail_flag) {
            status = 3;
            break;
        }
        product(std::integral_constant<int, 1>{});
        if (stamper) c1 = wall_clock64();
        // ---- partials of p.y, y.y and of the explicit r.r; one all-gather over the workgroups
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) s0 += pv[j] * yv[j], s1 += yv[j] * yv[j];
        s0 = wave_sum64(s0), s1 = wave_sum64(s1);
        const double s2 = wave_sum64(rr_part);
        if (lane == 0) red[wave][0] = s0, red[wave][1] = s1, red[wave][2] = s2;
        __syncthreads();
        unsigned long long* dslot = a.dboard + (size_t)(it & 1) * a.G * 6;
        if (tid < 3) {
            double v = 0;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
            publish_f64(dslot + (size_t)g * 6 + 2 * tid, epoch, v);
        }
        {   // thread t collects workgroup t's three sums (a lane re-reads its record until all six tags match, then stops loading); every
            // workgroup adds the G records in the same order.  A two-level form (groups of 8 / 16 / 32 workgroups handled by one wavefront,
            // then the group sums) was measured and dropped: a granule hop costs ~4 us under this load, two of them 7.9 / 9.8 / 12.0 us
            // against 4.6 us for the flat sweep on C2 (246 workgroups)
            double v0 = 0, v1 = 0, v2 = 0;
            bool fail = false;
            // gather_waves: how many wavefronts poll (4: thread t takes workgroup t's record; 1: wavefront 0 takes them all, 4 per lane)
            const int per_lane = a.gather_waves == 1 ? (a.G + 63) / 64 : 1;
            if (a.gather_waves == 1 ? wave == 0 : wave * 64 < a.G) {   // wave-uniform
                for (int rsel = 0; rsel < per_lane; ++rsel) {
                    const int w = a.gather_waves == 1 ? rsel * 64 + lane : tid;
                    const unsigned long long* gp = dslot + (size_t)(w < a.G ? w : 0) * 6;
                    pg_v2u64 q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0};
                    bool done = w >= a.G;
                    long long t_wait = 0;
                    for (unsigned spins = 0;; ++spins) {
                        if (!done) {
                            granule_load6(gp, q0, q1, q2);
                            done = granule_pair_ok(q0, epoch) && granule_pair_ok(q1, epoch) && granule_pair_ok(q2, epoch);
                        }
                        if (__all(done)) break;
                        if ((spins & 63u) == 63u) {
                            const long long now = wall_clock64();
                            if (t_wait == 0) t_wait = now;
                            else if (now - t_wait > kPersistTimeoutTicks) {
                                fail = true;
                                break;
                            }
                        }
                        if (a.poll_sleep == 1) __builtin_amdgcn_s_sleep(1);
                        else if (a.poll_sleep == 2) __builtin_amdgcn_s_sleep(2);
                        else if (a.poll_sleep >= 3) __builtin_amdgcn_s_sleep(8);
                    }
                    if (w < a.G) v0 += granule_pair_f64(q0), v1 += granule_pair_f64(q1), v2 += granule_pair_f64(q2);
                    if (fail) break;
                }
            }
            if (fail && lane == 0) fail_flag = 1;
            v0 = wave_sum64(v0), v1 = wave_sum64(v1), v2 = wave_sum64(v2);
            __syncthreads();   // the partials in red have been consumed
            if (lane == 0) red[wave][0] = v0, red[wave][1] = v1, red[wave][2] = v2;
            __syncthreads();
            if (tid < 3) {
                double v = 0;
#pragma unroll
                for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
                tot[tid] = v;
            }
        }
        __syncthreads();
        if (fail_flag) {
            status = 3;
            break;
        }
        const double pAp = tot[0], yy = tot[1];
        rr = tot[2];
        if (stamper) c2 = wall_clock64();
        // ---- the recurrence of k_cgf_update (kernels_krylov.h): stop test on the explicit r.r, then x, r, p
        if (rr <= a.tol2 * bb) {
            status = 1;
            break;
        }
        if (!(pAp > 0.0)) {
            status = 2;
            break;
        }
        if (it >= a.maxit) break;
        const double alpha = rr / pAp;
        const double est = alpha * alpha * yy - rr;
        const double beta = (est > 0.0 && rr > 0.0) ? est / rr : 0.0;
        rr_part = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            xv[j] += alpha * pv[j];
            rv[j] -= alpha * yv[j];
            pv[j] = rv[j] + beta * pv[j];
            rr_part += rv[j] * rv[j];
        }
        ++it;
        if (stamper) {
            const long long c3 = wall_clock64();
            t_spmv += c1 - c0, t_gather += c2 - c1, t_update += c3 - c2, ++n_stamped;
        }
    }
    if (status != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (dof[j] >= 0) a.x[dof[j]] = xv[j];
    }
    if (g == 0 && tid == 0) {
        a.sc[3] = rr;
        a.ctl[0] = status == 1 ? 1 : 0, a.ctl[1] = it, a.ctl[2] = status == 2 ? 1 : 0;
    }
    if (stamper) {
        double* st = a.stats + 4 * (size_t)g;
        st[0] = (double)n_stamped, st[1] = (double)t_spmv, st[2] = (double)t_gather, st[3] = (double)t_update;
    }
    if (status == 3 && tid == 0) atomicExch(a.ctl + 3, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_spmv_blocked: y = (I + At_offdiag) x for systems the persistent CG does not take (non-symmetric operators: BiCGStab; more
// than 2 M rows), on the same blocked sliced-ELL layout.  One workgroup per block of ~4096 rows: it stages the entries of x its
// rows read -- its own rows, then the imports from neighbouring blocks, by DOF id -- in LDS ONCE, then streams its block of the
// matrix with the gathers served by LDS.  The multi-launch CSR kernel (k_spmv_team2) gathers x from L2, which stops paying
// when x outgrows the L2s (C5: 43 MB of x, 62-67 % of 8 TB/s on the algorithmic bytes); here the stream runs at the HBM rate.
// Fused dots as k_spmv_team2: workgroup b writes (w.y, y.y | w.w) to partial[2 b], partial[2 b + 1].
// ---------------------------------------------------------------------------------------------------------------------
struct BlockedSpmvArgs {
    int32_t G, nsl, imp_cap, dot2_ww;
    const int32_t* slot_dof;
    const int64_t* ell_off;
    const int32_t* sl_off;
    const uint16_t* ell_code;
    const double* ell_val;
    const int32_t* imp_off;
    const int32_t* imp_dof;
    const int32_t* drop_dof;   // rows the layout leaves out (Dirichlet DOFs): unit diagonal only, y = x there
    int32_t n_drop;
    const double* x;
    double* y;
    const double* w;         // second vector of the fused dots (may equal x), nullptr: no dots
    double* partial;
    const int32_t* stop;
};
template <int R>
__global__ __launch_bounds__(kPersistT) void k_spmv_blocked(BlockedSpmvArgs a) {
    constexpr int T = kPersistT, W = T / 64, S = R * T;
    extern __shared__ double lds[];
    __shared__ double red[W][2];
    if (a.stop && __syncthreads_or(*a.stop != 0)) return;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsl = a.nsl;
    double* p_tab = lds;
    const int H = a.imp_off[g + 1] - a.imp_off[g];
    const int64_t e0 = a.ell_off[g];
    const double2* gv = reinterpret_cast<const double2*>(a.ell_val + e0);
    const uint32_t* gc = reinterpret_cast<const uint32_t*>(a.ell_code + e0);
    const int32_t* slo = a.sl_off + (size_t)g * (nsl + 1);
    int o0[R], w[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        o0[j] = __builtin_amdgcn_readfirstlane(slo[j * W + wave]);
        w[j] = __builtin_amdgcn_readfirstlane(slo[j * W + wave + 1]) - o0[j];
    }
    double yv[R];
    int32_t dof[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        dof[j] = a.slot_dof[(size_t)g * S + j * T + tid];
        yv[j] = dof[j] >= 0 ? a.x[dof[j]] : 0.0;   // unit diagonal of the scaled system
        p_tab[j * T + tid] = yv[j];
    }
    for (int h = tid; h < H; h += T) p_tab[S + h] = a.x[a.imp_dof[a.imp_off[g] + h]];
    __syncthreads();
    double wy = 0, second = 0;
    if (a.w != nullptr && a.dot2_ww) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const double wv = (a.w == a.x) ? yv[j] : (dof[j] >= 0 ? a.w[dof[j]] : 0.0);
            second += wv * wv;
        }
    }
    auto product = [&](auto half) {
        constexpr int J0 = decltype(half)::value ? R / 2 : 0, J1 = decltype(half)::value ? R : R / 2;
        int mw = 0;
#pragma unroll
        for (int j = J0; j < J1; ++j) mw = max(mw, w[j]);
        for (int e = 0; e < mw; ++e) {
            double2 v[J1 - J0];
            uint32_t c[J1 - J0];
#pragma unroll
            for (int j = J0; j < J1; ++j) {
                const int idx = (o0[j] + min(e, max(w[j] - 1, 0))) * 64 + lane;
                v[j - J0] = gv[idx], c[j - J0] = gc[idx];
            }
#pragma unroll
            for (int j = J0; j < J1; ++j) {
                const double t = v[j - J0].x * p_tab[c[j - J0] & 0xffffu] + v[j - J0].y * p_tab[c[j - J0] >> 16];
                yv[j] += e < w[j] ? t : 0.0;
            }
        }
    };
    product(std::integral_constant<int, 0>{});
    product(std::integral_constant<int, 1>{});
    {   // the left-out rows (the vector kernels sweep all n entries: y must be defined there; x is zero on them inside a solve)
        const int chunk = (a.n_drop + a.G - 1) / a.G;
        for (int i = g * chunk + tid; i < min(a.n_drop, (g + 1) * chunk); i += T) a.y[a.drop_dof[i]] = a.x[a.drop_dof[i]];
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (dof[j] < 0) continue;
        a.y[dof[j]] = yv[j];
        if (a.w != nullptr) {
            const double wv = (a.w == a.x) ? p_tab[j * T + tid] : a.w[dof[j]];
            wy += wv * yv[j];
            if (!a.dot2_ww) second += yv[j] * yv[j];
        }
    }
    if (a.partial != nullptr) {
        wy = wave_sum64(wy), second = wave_sum64(second);
        if (lane == 0) red[wave][0] = wy, red[wave][1] = second;
        __syncthreads();
        if (tid < 2) {
            double v = 0;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
            a.partial[2 * g + tid] = v;
        }
    }
}

// ell_val[e] = scaled full-pattern value the entry maps to, 0 in padding
__global__ __launch_bounds__(256) void k_persist_fill(int64_t n, const int32_t* src, const double* scaled_full, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[i] >= 0 ? scaled_full[src[i]] : 0.0;
}

}  // namespace fdapde_hip
#endif
