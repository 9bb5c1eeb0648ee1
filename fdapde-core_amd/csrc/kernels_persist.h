// kernels_persist.h -- the whole Jacobi-PCG solve of a system of up to ~2 M rows as ONE launch (fdapde_solve / fdapde_lin_solve /
// the parabolic stepper, single GPU, symmetric positive operator): the in-solve SpMV of matrices of a few tens of MB is bound by
// launch and hand-off latency, not by bandwidth (C2: 17 us per SpMV launch + 6 us update + two dependent launches = 27-29 us per
// iteration for 40 MB of matrix), so the iteration is restructured around what the chip can keep ON the CUs:
//   * one workgroup per CU, each owning a contiguous range of interior rows (locality numbering => few neighbours);
//   * its slice of the scaled matrix RESIDENT IN LDS as sliced ELL (8-byte value + 16-bit column code per entry; 256 CUs x 160 KB
//     = 40 MB), x / r / p of its rows in registers; a system whose blocks do not fit streams them from memory every iteration
//     (STREAM) -- then in symmetric storage where that pays (SYM): half the bytes, the vectors still never leave the CU;
//   * neighbours exchange the entries of p they need through 8-byte {epoch, payload} granules (the data is the flag: no barrier,
//     no fence, MI355X_MICROARCH.md "handoff-1to1"; a double = two granules written by ONE 16-byte store and read by one 16-byte
//     load, four imports per lane in flight together), rows that need no import are multiplied while the granules travel;
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

// A/B switches of the symmetric streaming form at 16 rows per thread (C3; tools/c3_stream_ab.sh -> profiles/r5_c3_stream_ab.txt):
//   FDAPDE_SYM_U8    pair rows of every pass loaded together per entry step where a phase has 8 passes (1: one, the default; 2: two)
//   FDAPDE_SYM_NT_V  cache policy of the value loads (buffer_load aux: 0 default, 2 = nt);  FDAPDE_SYM_NT_C  the same for the column-code loads
#ifndef FDAPDE_SYM_U8
#define FDAPDE_SYM_U8 1
#endif
#ifndef FDAPDE_SYM_NT_V
#define FDAPDE_SYM_NT_V 0
#endif
#ifndef FDAPDE_SYM_NT_C
#define FDAPDE_SYM_NT_C 0
#endif
#ifndef FDAPDE_SYM_U
#define FDAPDE_SYM_U 1
#endif

namespace fdapde_hip {

#ifdef FDAPDE_DOT_STRIDE6   // (A/B builds of tools/: the dot records 48 bytes apart -- every third one straddles two lines)
constexpr int kDotStride = 6;
#else
constexpr int kDotStride = 8;   // granules between the dot records of two workgroups: 64 bytes, on a 64-byte boundary (persist_engine.hip dboard_offset)
#endif
#ifdef FDAPDE_GATHER_3BAR   // (A/B builds of tools/: the record sums through `tot` and three barriers)
constexpr bool kGather3 = true;
#else
constexpr bool kGather3 = false;
#endif

struct PersistArgs {
    int32_t G, nsl, maxit, imp_cap;   // workgroups; slices per workgroup; iteration bound; import slots reserved in LDS
    int32_t lds_cap;                  // ELL entries staged in LDS by the resident form (multiple of 128; >= the largest block)
    int32_t time_phases;              // != 0: every workgroup accumulates its phase durations into stats
    int32_t gather_waves, poll_sleep; // all-gather of the dot records: polling wavefronts (4 | 1) and the pause between polls (0 none .. 3 long)
    uint32_t epoch0;                  // tags of this launch are epoch0 + iteration + 1: boards are never cleared between launches
    int32_t timeout_ticks;            // bound of every wait, in 10 ns ticks of s_memrealtime
    int32_t debug_stall_it;           // > 0 (tests): the last workgroup leaves at this iteration without publishing, as a peer that is not resident would
    int32_t n_cols;                   // > 1 (single GPU, CG): gridDim.y independent systems with the same matrix -- column c reads r_in + c col_stride and
                                      // sc[4 c], writes x_out + c col_stride, sc[4 c + 3] and ctl[4 c ..], and talks over boards of its own
                                      // (pboard / dboard + c board_stride); x == nullptr: every column starts from 0
    int64_t col_stride, board_stride;
    int32_t direct;                   // ONE workgroup, one column, started from 0 (a small handle solve that cannot be batched): the launch does the
                                      // caller's prologue and epilogue itself -- reads the right-hand side as handed over (b_ext, reference DOF order,
                                      // pinned host memory read over PCIe), scales it, takes ||b~||^2 from its own first r.r, and writes the UNSCALED
                                      // solution in the reference order (x_ext) and the outcome record (rec) straight into pinned host memory:
                                      // no prologue / epilogue kernels, no copies, one wait
    const double* b_ext;              // direct: right-hand side, reference DOF order
    double* x_ext;                    // direct: solution, reference DOF order
    double* rec;                      // direct: [0] iterations, [1] final r.r, [2] ||b~||^2, [3] status + 1 (written LAST: 1 maxit, 2 converged, 3 breakdown)
    int32_t exp_lds;                  // symmetric streaming form: the workgroup's export list (slot codes) is staged in LDS once (the host found room for it)
                                      // instead of being re-read from global memory in front of every iteration's export stores
    double* hrec;                     // one-workgroup launches of fdapde_solve (G == 1, not direct): the outcome ALSO into pinned host memory, so that the
                                      // host reads it after its one wait without device-to-host copies: [0] stop flag, [1] iterations, [2] breakdown,
                                      // [3] gave up, [4] ctl[4] (the deferred positive-diagonal flag), [5] ||b~||^2, [6] final r.r
    double* u_out;                    // one-workgroup launches of fdapde_solve behind k_small_front: the epilogue too -- u = scale * x on the rows of the
                                      // layout (k_unscale's expression: the lift is zero there; the Dirichlet entries of u were written by k_small_front)
    const int32_t* i2e;               // direct: internal DOF -> reference DOF
    const double* scale;              // direct: Jacobi scale, internal order
    int32_t pf_steps;                 // streaming forms, != 0: the first entry step of the next operator application is touched (pulled into the L2)
                                      // while the workgroup waits for the dot records
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
    const unsigned long long* amax_bits;   // symmetric storage: bit pattern of max |stored value| (k_persist_fill)
    int32_t max_len;                       // symmetric storage: a row receives at most this many transposed products
    const double* r_in;           // initial residual (= initial direction), internal DOF order
    const double* x;              // initial guess (scaled unknowns), internal DOF order
    double* x_slots;              // WIDE form (R > 16): x of every slot, [G][R T] in slot order -- read and written once per iteration, coalesced
    double* x_out;                // solution; a buffer of its own: a launch that gives up (ctl[3]) must leave the guess as it found it,
                                  // whatever the workgroups that did finish have stored (the host restarts from x, r, p)
    double* sc;                   // sc[0] = reference norm^2 (in); sc[3] = final r.r (out)
    int32_t* ctl;                 // out: [0] converged, [1] iterations, [2] breakdown -- written only by a launch that did not give up;
                                  // [3] hand-off timeout (then ctl[0..2] and sc[3] are as the host left them)
    double* stats;                // per workgroup: [0] iterations timed, [1] operator phase (SpMV + imports), [2] all-gather phase (incl. the
                                  // wait for the slowest workgroup), [3] update phase -- sums of 10 ns ticks
    // ---- row-distributed form (DIST): the workgroups of `world` launches -- one per rank, each on its own GPU (tests: sharing one) -- act
    //      as ONE grid.  Hand-offs inside a rank work exactly as on one GPU (pboard / dboard: ordinary device memory, agent scope).  What
    //      crosses ranks goes through a second, small board per rank in FINE-GRAINED memory that every rank maps (hipIpc):
    //      [entries imported from other ranks | one dot record per RANK x 2 buffers].  Exporters PUSH an entry another rank needs into that
    //      rank's board (posted writes over xGMI through the peer-mapped pointer), so all polling is local; the dot products are gathered in
    //      two levels: inside the rank as ever, then workgroup 0 pushes the rank's sums to every rank and all workgroups add the `world` rank
    //      records in rank order (the same bits everywhere).  Accesses of that board are system-scope (sc0 sc1).
    int32_t world, rank, n_board_local;   // ranks; this rank; import positions >= n_board_local lie in the remote section of rboard
    int32_t flat_gather, g_base, G_tot;   // flat_gather: ONE hop instead of two for the dot products -- every workgroup pushes its record to every
                                          // rank (flat section behind the rank records: [buffer][G_tot][4 values][2 granules]) and gathers all G_tot
                                          // records of all ranks' workgroups (global index g_base + g); chosen for few ranks (G_tot <= 1024)
    int32_t timeout_first_ticks;      // bound of the waits of iteration 0 (the launches of the ranks start at different times)
    const int32_t* rexp_off;          // [G + 1] remote exports of a workgroup
    const uint16_t* rexp_slot;        // slot whose entry goes out
    const int32_t* rexp_peer;         // to which rank
    const int32_t* rexp_pos;          // at which position of that rank's remote section
    const uint8_t* wg_late;                   // [G] or nullptr: 1 = this workgroup fetches its imports before its first pass (its importing rows overflow the
                                              // second half of its slots: host_build_persist_layout allow_late)
    unsigned long long* rboard;               // this rank's fine-grained board: remote section
    unsigned long long* const* peer_pboard;   // [world] remote sections of all ranks' fine-grained boards
    unsigned long long* const* peer_dboard;   // [world] rank-record sections: [buffer][world][4 values][2 granules]
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
typedef unsigned int pg_u32x4 __attribute__((ext_vector_type(4)));
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
// a double published with ONE 16-byte write-through store: {low word, epoch, high word, epoch} -- the same two granules as publish_f64
// (each aligned 8-byte half carries its own tag and is validated on its own by the reader), half the fabric writes
__device__ __forceinline__ void publish_f64_x4(unsigned long long* g2, unsigned epoch, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    pg_u32x4 q;
    q.x = (unsigned)b, q.y = epoch, q.z = (unsigned)(b >> 32), q.w = epoch;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g2), "v"(q) : "memory");
}
// four published doubles fetched with the loads in flight together (one round trip instead of four)
__device__ __forceinline__ void granule_load2x4(const unsigned long long* p0, const unsigned long long* p1, const unsigned long long* p2,
                                                const unsigned long long* p3, pg_v2u64& a, pg_v2u64& b, pg_v2u64& c, pg_v2u64& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
// system-scope forms (row-distributed launches: the other end of a hand-off may be another GPU; fine-grained memory)
__device__ __forceinline__ void publish_f64_x4_sys(unsigned long long* g2, unsigned epoch, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    pg_u32x4 q;
    q.x = (unsigned)b, q.y = epoch, q.z = (unsigned)(b >> 32), q.w = epoch;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(g2), "v"(q) : "memory");
}
__device__ __forceinline__ pg_v2u64 granule_load2_sys(const unsigned long long* p) {
    pg_v2u64 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void granule_load6_sys(const unsigned long long* p, pg_v2u64& a, pg_v2u64& b, pg_v2u64& c) {
    asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p) : "memory");
}
__device__ __forceinline__ void granule_load2x4_sys(const unsigned long long* p0, const unsigned long long* p1, const unsigned long long* p2,
                                                    const unsigned long long* p3, pg_v2u64& a, pg_v2u64& b, pg_v2u64& c, pg_v2u64& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
// one 128-byte line of the matrix stream pulled towards the L2.  The loaded word goes to a dump in LDS (global_load_lds: no register to
// keep reserved while the load is in flight, and the compiler, which does not see the load, waits for it nowhere); m0 = LDS base of the dump
__device__ __forceinline__ void touch_line(const void* p, unsigned lds_dump) {
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_saved) : "v"(p), "s"(lds_dump) : "memory");
}
// the first entry step of the passes without imports ([0, RI)) of a workgroup's block, touched for the whole workgroup by the wavefronts
// W / 2 .. W - 1 (those that do not poll the dot records): one lane per line, 8 lines of values + 2 of codes per pair row
template <int RI, int W>
__device__ __forceinline__ void touch_first_step(const int32_t* slo, const void* gv, const void* gc, int wave, int lane, unsigned lds_dump) {
    constexpr int per_wave = RI * 10, total = 2 * per_wave;
    if (wave < W / 2) return;
#pragma unroll
    for (int t = 0; t * 64 < total; ++t) {
        int idx = min(t * 64 + lane, total - 1);
        asm volatile("" : "+v"(idx));   // (the addresses do not change between iterations: recomputed here all the same, not kept in registers)
        const int half = idx / per_wave, rem = idx - half * per_wave;
        const int j = rem / 10, k = rem - j * 10;
        const int tw = (wave - W / 2) + half * (W / 2);   // the wavefront whose slices these are
        const int o = slo[j * W + tw], wj = slo[j * W + tw + 1] - o;
        const char* line = k < 8 ? static_cast<const char*>(gv) + (size_t)o * 1024 + k * 128 : static_cast<const char*>(gc) + (size_t)o * 256 + (k - 8) * 128;
        if (t * 64 + lane < total && wj > 0) touch_line(line, lds_dump);
    }
}
// Sum over the wavefront, every lane gets the total: the butterfly v += v[lane ^ o], o = 32 .. 1.  The four steps inside a row of 16 lanes are DPP
// moves (VALU path) instead of the ds_bpermute pairs __shfl_xor compiles to (LDS crossbar, ~100 cycles each, twelve of them chained per sum): a rotation
// by 8 / 4 inside the row pairs every lane with a lane of the class lane ^ 8 / lane ^ 4 of the butterfly (the partial sums depend on the lane's class
// only), so the operands -- and the bits -- are those of the xor butterfly.
template <int CTRL> __device__ __forceinline__ double persist_dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int O> __device__ __forceinline__ double persist_swap_sum(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned l = (unsigned)(b & 0xffffffffll), h = (unsigned)(b >> 32);
    const auto lo = O == 32 ? __builtin_amdgcn_permlane32_swap(l, l, false, false) : __builtin_amdgcn_permlane16_swap(l, l, false, false);
    const auto hi = O == 32 ? __builtin_amdgcn_permlane32_swap(h, h, false, false) : __builtin_amdgcn_permlane16_swap(h, h, false, false);
    return __longlong_as_double(((long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
}
// max over the wavefront, in every lane (exact whatever the order): the same VALU-only butterfly
__device__ __forceinline__ double wave_max64(double v) {
#ifdef FDAPDE_WAVE_SUM_SHFL
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
#endif
    const long long b = __double_as_longlong(v);
    const unsigned l = (unsigned)(b & 0xffffffffll), h = (unsigned)(b >> 32);
    const auto lo = __builtin_amdgcn_permlane32_swap(l, l, false, false), hi = __builtin_amdgcn_permlane32_swap(h, h, false, false);
    v = fmax(__longlong_as_double(((long long)hi[0] << 32) | lo[0]), __longlong_as_double(((long long)hi[1] << 32) | lo[1]));
    const long long b2 = __double_as_longlong(v);
    const unsigned l2 = (unsigned)(b2 & 0xffffffffll), h2 = (unsigned)(b2 >> 32);
    const auto lo2 = __builtin_amdgcn_permlane16_swap(l2, l2, false, false), hi2 = __builtin_amdgcn_permlane16_swap(h2, h2, false, false);
    v = fmax(__longlong_as_double(((long long)hi2[0] << 32) | lo2[0]), __longlong_as_double(((long long)hi2[1] << 32) | lo2[1]));
    v = fmax(v, persist_dpp_f64<0x128>(v));
    v = fmax(v, persist_dpp_f64<0x124>(v));
    v = fmax(v, persist_dpp_f64<0x4E>(v));
    v = fmax(v, persist_dpp_f64<0xB1>(v));
    return v;
}
__device__ __forceinline__ double wave_sum64(double v) {
#ifdef FDAPDE_WAVE_SUM_SHFL   // (A/B builds of tools/)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
#endif
    // lane ^ 32, lane ^ 16: gfx950's swaps (v_permlane32_swap / v_permlane16_swap exchange the upper half / the odd rows of one register with the lower
    // half / the even rows of another; given the same value twice they leave {low, low} and {high, high}: their sum is the butterfly's in every lane)
    v = persist_swap_sum<32>(v);
    v = persist_swap_sum<16>(v);
    v += persist_dpp_f64<0x128>(v);   // row_ror:8
    v += persist_dpp_f64<0x124>(v);   // row_ror:4
    v += persist_dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += persist_dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    return v;
}

// waits are bounded by PersistArgs::timeout_ticks of s_memrealtime (100 MHz): a legitimate wait is an iteration's skew, tens of us

// R rows per thread (2, 4, 8, 16); passes [0, R / 2) hold rows that import nothing, passes [R / 2, R) the others (host_persist.cpp).
// STREAM = false: the workgroup's whole block of the matrix is staged in LDS once and re-read from there every iteration.
// STREAM = true : the block streams from global memory every iteration (it is larger than the LDS: systems of a few hundred MB).
//                 x, r, p still never leave the registers, so an iteration moves the matrix and the exchanged entries of p and
//                 nothing else -- the multi-launch path moves 7 vector passes on top.  All R / 2 value loads of an entry step are
//                 issued before the first is used (raw buffer loads; a pass that has run out of entries is skipped).
// SYM = true   : symmetric storage (internal.h persist_sym_owner): a pair of rows of this block is stored once and applied to both
//                 rows.  The row that stores it adds a p[col] to its own sum (registers, as ever) and hands a p[row] to row col through
//                 a table of 64-bit FIXED-POINT accumulators in LDS (ds_add_u64): integer addition is associative, so the sums -- and
//                 everything downstream -- are bitwise reproducible whatever order the wavefronts arrive in, which fp64 atomics would
//                 not give.  The scale is a power of two chosen per iteration from the block's own max |p|, max |a| and the longest
//                 row, so that no partial sum can reach 2^62: the transposed sums carry an ABSOLUTE error below
//                 (products) x 2^-56 x max_len x max|a| x max_block |p| -- finer than fp64's own rounding of the largest products.
//                 The import / export lists are read from global memory (L2) instead of LDS: the table takes their room.
// DIST = true   : row-distributed form, see PersistArgs.
// R > 16 (WIDE) : plain streaming storage for workgroups of up to 24 x 512 rows -- systems of 2.1 to 3.1 M rows on 256 CUs, which had to take the
//                 multi-launch path (seven vector passes and two launches per iteration: 98 us per iteration at 2.35 M rows against ~57 here).
//                 r and y of a thread's rows stay in registers; x lives in HBM in SLOT order (one coalesced read + write per row and iteration:
//                 + 16 B per row on a stream of ~130 B per row), p in its LDS table only (as in the symmetric form), import / export lists are
//                 read from global memory.  LDS: 96 KB of table + imports.
template <int R, bool STREAM, bool SYM, bool DIST = false, int WIDE_GJ = 12>
static __global__ __launch_bounds__(kPersistT) void k_cg_persist(PersistArgs a) {
    constexpr int T = kPersistT, W = T / 64, S = R * T, RI = R / 2;
    constexpr bool WIDE = R > kPersistRmax;   // x in HBM, p in LDS only
    constexpr bool PLDS = SYM || WIDE;        // p of the own rows lives in its LDS table only; DOF ids re-read at the end; lists in global memory
    static_assert(!WIDE || (STREAM && !SYM && !DIST), "the wide form is the plain streaming storage on one GPU");
    extern __shared__ __attribute__((aligned(128))) double lds[];   // (16-byte LDS reads of the resident blocks: a base the static arrays left 8-byte aligned halves their rate)
    __shared__ double red[W][3], red2[W][3];
    __shared__ double tot[3];
    __shared__ double pmax_w[W];
    __shared__ int32_t fail_flag;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (!DIST) {
        if (a.n_cols > 1) {   // several right-hand sides at once: this workgroup belongs to column blockIdx.y (uniform)
            const size_t col = blockIdx.y;
            a.r_in += col * a.col_stride, a.x_out += col * a.col_stride, a.sc += 4 * col, a.ctl += 4 * col;
            if (a.x != nullptr) a.x += col * a.col_stride;
            a.pboard += col * a.board_stride, a.dboard += col * a.board_stride;
        }
    }
    const int nsl = a.nsl;
    const int H = a.imp_off[g + 1] - a.imp_off[g], E = a.exp_off[g + 1] - a.exp_off[g];
    double* p_tab = lds;                                                                  // [S + imp_cap]
    long long* y_tab = reinterpret_cast<long long*>(p_tab + (S + a.imp_cap));             // [S] transposed sums, fixed point (SYM)
    double2* ev = reinterpret_cast<double2*>(y_tab + (SYM ? S : 0));                      // [lds_cap / 2] entry pairs (resident form)
    uint32_t* ec = reinterpret_cast<uint32_t*>(ev + a.lds_cap / 2);                       // [lds_cap / 2] code pairs
    int32_t* impl_l = reinterpret_cast<int32_t*>(ec + a.lds_cap / 2);                     // [imp_cap]   (not SYM)
    uint16_t* expl_l = reinterpret_cast<uint16_t*>(impl_l + a.imp_cap);                   // [E]         (not SYM)
    const int64_t e0 = a.ell_off[g];
    const double2* gv = reinterpret_cast<const double2*>(a.ell_val + e0);
    const uint32_t* gc = reinterpret_cast<const uint32_t*>(a.ell_code + e0);
    // streaming form: the block is read through raw buffer loads (descriptor in scalar registers: base = this workgroup's block)
    // (records end with the arrays -- all blocks + 256 entries of slack: a touch past the last block reads 0 instead of faulting)
    const int64_t e_left = a.ell_off[a.G] + 256 - e0;
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(gv), 0, (int)min(e_left * 8, (int64_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(gc), 0, (int)min(e_left * 2, (int64_t)0x7fffffff), 0x00020000);
    const int lane16 = lane * 16, lane4 = lane * 4;
    // ---- stage the workgroup's tables (and, resident form, its block of the matrix)
    const int32_t* impl = PLDS ? a.imp_pos + a.imp_off[g] : impl_l;
    const uint16_t* expl = PLDS ? a.exp_slot + a.exp_off[g] : expl_l;
    if constexpr (SYM && STREAM && !DIST) {
        // the export list behind the accumulator table where the host found room (C3: 5.4 KB of the 6 KB the workgroup had left): the global re-read
        // of the codes put a round trip in front of the export stores of every iteration, and those sit in front of the matrix stream (vmcnt is in
        // order) -- a fit of the operator-phase stamps priced an export at 3.8 stored entries
        if (a.exp_lds) {
            uint16_t* expl_s = reinterpret_cast<uint16_t*>(y_tab + S);
            for (int i = tid; i < E; i += T) expl_s[i] = a.exp_slot[a.exp_off[g] + i];
            expl = expl_s;   // (visible to all threads after the barrier that follows the staging of the tables)
        }
    }
    if constexpr (!PLDS) {
        for (int i = tid; i < H; i += T) impl_l[i] = a.imp_pos[a.imp_off[g] + i];
        for (int i = tid; i < E; i += T) expl_l[i] = a.exp_slot[a.exp_off[g] + i];
    } else if constexpr (SYM) {
#pragma unroll
        for (int j = 0; j < R; ++j) y_tab[j * T + tid] = 0;
    }
    // symmetric storage: exponent of the bound on (longest row) x max |a| (the values are in place when the solve is launched)
    int bexp = 0;
    if constexpr (SYM) {
        const double amax = __longlong_as_double((long long)*a.amax_bits);
        bexp = amax > 0.0 && amax < 1.7e308 ? ilogb(amax * (double)a.max_len) + 1 : 0;
    }
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
    // ---- the rows of this thread: slot j * T + tid, j < R.  SYM: p of the own rows lives in its LDS table only and the DOF ids are
    //      re-read at the end (the registers go to the transposed products)
    double xv[WIDE ? 1 : R], rv[R], pv[PLDS ? 1 : R];
    int32_t dof[PLDS ? 1 : R];
    double* const xs = WIDE ? a.x_slots + (size_t)g * S + tid : nullptr;   // this thread's x entries: xs[j T]
    double rr_part = 0;
    auto P = [&](int j) -> double {
        if constexpr (PLDS) return p_tab[j * T + tid];
        else return pv[j];
    };
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int32_t d = a.slot_dof[(size_t)g * S + j * T + tid];
        const bool on = d >= 0;
        if (!DIST && a.direct) rv[j] = on ? a.scale[d] * (a.b_ext[a.i2e[d]] - 0.0) : 0.0;   // (k_cols_init's expression)
        else rv[j] = on ? a.r_in[d] : 0.0;
        const double x0 = on && a.x != nullptr ? a.x[d] : 0.0;
        if constexpr (WIDE) xs[j * T] = x0;
        else xv[j] = x0;
        if constexpr (PLDS) p_tab[j * T + tid] = rv[j];
        else pv[j] = rv[j], dof[j] = d;
        rr_part += rv[j] * rv[j];
    }
    double bb = (!DIST && a.direct) ? 0.0 : a.sc[0];   // (direct: the launch's own first r.r, below)
    const bool stamper = a.time_phases && tid == 0;   // every workgroup stamps its own phases (a few s_memrealtime per iteration)
    long long t_spmv = 0, t_gather = 0, t_update = 0, n_stamped = 0;
    int it = 0, status = 0;   // status: 1 converged, 2 breakdown, 3 hand-off timeout
    double rr = 0;
    // streaming forms: while the workgroups wait for the dot records the memory system idles (they all wait at the same time, ~3 us of ~30
    // on C3), and the matrix does not depend on p: the wavefronts that do not poll (4 .. 7) touch the lines of the first entry step(s) of the
    // workgroup's next operator application -- one lane per 128-byte line (8 of values + 2 of codes per pair row), the loaded words are
    // dropped -- so that the stream starts from the L2.  Nobody waits for the touches (touch_line).
    // C3 30.27 -> 29.71 us per iteration, 1.03 M rows 21.33 -> 20.91.  Measured and dropped: every wavefront touching its own lines before it
    // polls (loads return in order, the poll sits behind the touches: 31.2 -> 31.9); two entry steps instead of one (160 KB per workgroup: more
    // than an XCD's L2 holds for its 32 workgroups, 30.3 -> 31.6); the first step of the passes WITH imports touched by each wavefront before
    // it collects its imports (the imports have arrived by then and now wait behind the touches: 30.3 -> 31.2); touching ahead INSIDE the
    // operator phase (same reason).
    __shared__ unsigned pf_dump[64];
    [[maybe_unused]] auto prefetch_next = [&]() {
        if constexpr (STREAM)
            if (a.pf_steps != 0) touch_first_step<RI, W>(slo, gv, gc, wave, lane, (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)pf_dump);
    };
    __syncthreads();
    for (;;) {
        const unsigned epoch = a.epoch0 + (unsigned)it + 1u;
        const long long tmo = (DIST && it == 0) ? (long long)a.timeout_first_ticks : (long long)a.timeout_ticks;
        if (a.debug_stall_it > 0 && it == a.debug_stall_it && g == a.G - 1) {   // (wave-uniform)
            status = 3;
            break;
        }
        long long c0 = 0, c1 = 0, c2 = 0;
        if (stamper) c0 = wall_clock64();
        // ---- p of the own rows into the LDS table; exported entries onto the board
        if constexpr (!PLDS) {
#pragma unroll
            for (int j = 0; j < R; ++j) p_tab[j * T + tid] = pv[j];
        }
        double tscale = 0, tinv = 0;   // fixed-point scale of the transposed sums and its inverse (powers of two)
        if constexpr (SYM) {
            double m = 0;
#pragma unroll
            for (int j = 0; j < R; ++j) m = fmax(m, fabs(P(j)));
            m = wave_max64(m);
            if (lane == 0) pmax_w[wave] = m;
        }
        __syncthreads();
        if constexpr (SYM) {
            double m = 0;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) m = fmax(m, pmax_w[ww]);
            // |sum| <= max_len max|a| max|p| < 2^(bexp + ilogb(m) + 1): scaled below 2^62
            // (clamped: a block whose largest |p| is subnormal-small would ask for 2^ex beyond the exponent range -- inf scale, 0 inverse;
            // such a p contributes nothing at fp64 whatever the scale)
            const int ex = m > 0.0 && m < 1.7e308 ? min(max(61 - bexp - ilogb(m), -1000), 1000) : 0;
            tscale = ldexp(1.0, ex), tinv = ldexp(1.0, -ex);
        }
        // 8 (or fewer) rows per thread: the scaled p of the own rows stays in registers for the operator phase (with 16 there is no room:
        // it is re-read from the LDS table at every entry step)
        constexpr bool PS_REG = SYM && R <= 8;
        double psj[PS_REG ? R : 1];
        if constexpr (PS_REG) {
#pragma unroll
            for (int j = 0; j < R; ++j) psj[j] = P(j) * tscale;
        }
        for (int i0 = tid; i0 < E; i0 += 4 * T) {   // four per thread at a time: slot codes, then table reads, then the stores
            unsigned code[4];
            double pe[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) code[k] = expl[min(i0 + k * T, E - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) pe[k] = p_tab[code[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k * T < E) publish_f64_x4(a.pboard + 2 * (size_t)(a.exp_off[g] + i0 + k * T), epoch, pe[k]);
        }
        if constexpr (DIST) {   // entries other ranks import: pushed into their boards
            const int re0 = a.rexp_off[g], RE = a.rexp_off[g + 1] - re0;
            for (int i = tid; i < RE; i += T) {
                const unsigned slot = a.rexp_slot[re0 + i];
                publish_f64_x4_sys(a.peer_pboard[a.rexp_peer[re0 + i]] + 2 * (size_t)a.rexp_pos[re0 + i], epoch, p_tab[slot]);
            }
        }
        // ---- y = (I + At_offdiag) p: the passes without imports first, the others once the neighbours' entries have arrived
        double yv[R];
#pragma unroll
        for (int j = 0; j < R; ++j) yv[j] = P(j);   // unit diagonal of the scaled system
        auto product = [&](auto phase) {
            constexpr int J0 = decltype(phase)::value ? RI : 0, J1 = decltype(phase)::value ? R : RI, NJ = J1 - J0;
            int mw = 0;   // widest slice of the phase
#pragma unroll
            for (int j = J0; j < J1; ++j) mw = max(mw, w[j]);
            if constexpr (!SYM && STREAM) {
                // plain storage, streaming: a pass that has run out of entries is skipped (wave-uniform) instead of re-reading its last
                // pair row -- the rows are sorted by length, and on P2 systems (10 to 60+ entries per row) the passes of a phase differ
                // by a factor in width; raw buffer loads (scalar row offset + constant lane offset)
                // U pair rows of every pass per step where a phase has only 1 or 2 passes (2 / 4 rows per thread): the loads in flight per
                // wavefront, not the memory, bound the stream otherwise (kernels_persist_bicg.h)
                constexpr int U = NJ >= 4 ? 1 : 8 / NJ;
                // (the wide form has 12 passes per phase; WIDE_GJ of them load together.  All 12: 60 registers in flight next to r and y of 24 rows,
                //  a few scalars spill into lanes -- still the fastest: 2.35 M rows 71.8 us per iteration against 73.4 with 6 and 73.0 with 4)
                constexpr int GJ = NJ > 8 ? WIDE_GJ : NJ;
                static_assert(NJ % GJ == 0, "the passes of a phase in equal groups");
                for (int e = 0; e < mw; e += U) {
#pragma unroll
                    for (int jg = J0; jg < J1; jg += GJ) {
                        pg_u32x4 v[U][GJ];
                        uint32_t c[U][GJ];
#pragma unroll
                        for (int u = 0; u < U; ++u)
#pragma unroll
                            for (int j = jg; j < jg + GJ; ++j) {
                                if (e + u < w[j]) {
                                    const int row = o0[j] + e + u;
                                    v[u][j - jg] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, lane16, row * 1024, 0);
                                    c[u][j - jg] = __builtin_amdgcn_raw_buffer_load_b32(rs_c, lane4, row * 256, 0);
                                }
                            }
#pragma unroll
                        for (int u = 0; u < U; ++u)
#pragma unroll
                            for (int j = jg; j < jg + GJ; ++j) {
                                if (e + u < w[j]) {
                                    const pg_u32x4 q = v[u][j - jg];
                                    const double vx = __hiloint2double((int)q.y, (int)q.x), vy = __hiloint2double((int)q.w, (int)q.z);
                                    yv[j] += vx * p_tab[c[u][j - jg] & 0xffffu] + vy * p_tab[c[u][j - jg] >> 16];
                                }
                            }
                    }
                }
            } else if constexpr (!SYM) {
                // resident blocks: an entry step is two dependent LDS round trips (value + codes, then the table reads the codes address); the
                // pair rows of step e + 1 are requested behind the table reads of step e (LDS returns in order: the table reads are waited for with
                // the next step's loads still in flight), so a step costs one round trip -- what a workgroup of 8 wavefronts alone on its CU cannot
                // hide otherwise.  Same sums in the same order.
                double2 vn[NJ];
                uint32_t cn[NJ];
                auto fetch = [&](int e) {
#pragma unroll
                    for (int j = J0; j < J1; ++j) {
                        const int ee = min(e, max(w[j] - 1, 0));
                        const int idx = (o0[j] + ee) * 64 + lane;
                        if constexpr (STREAM) vn[j - J0] = gv[idx], cn[j - J0] = gc[idx];
                        else vn[j - J0] = ev[idx], cn[j - J0] = ec[idx];
                    }
                };
                if (mw > 0) fetch(0);
                for (int e = 0; e < mw; ++e) {
                    double2 v[NJ];
                    double pa[NJ], pb[NJ];
#pragma unroll
                    for (int j = J0; j < J1; ++j) {
                        v[j - J0] = vn[j - J0];
                        pa[j - J0] = p_tab[cn[j - J0] & 0xffffu], pb[j - J0] = p_tab[cn[j - J0] >> 16];
                    }
                    if (e + 1 < mw) fetch(e + 1);
#pragma unroll
                    for (int j = J0; j < J1; ++j) {
                        const double t = v[j - J0].x * pa[j - J0] + v[j - J0].y * pb[j - J0];
                        yv[j] += e < w[j] ? t : 0.0;
                    }
                }
            } else {
                // pair row e of the passes [jg, jg + GS).  The rows of a workgroup are sorted by length, so the passes of a phase differ in
                // width: a pass that has run out of entries is SKIPPED (wave-uniform branch) -- unlike the plain form, where re-reading its
                // last pair row and leaving it out of the sum costs a multiply-add, here an entry is ~25 instructions and an LDS atomic
                auto load = [&](auto& v, auto& c, int jg, int e) {
                    constexpr int GS = sizeof(c) / sizeof(c[0]);
#pragma unroll
                    for (int j = jg; j < jg + GS; ++j) {
                        if (e < w[j]) {
                            const int row = o0[j] + e;
                            if constexpr (STREAM) {
                                v[j - jg] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, lane16, row * 1024, FDAPDE_SYM_NT_V);
                                c[j - jg] = __builtin_amdgcn_raw_buffer_load_b32(rs_c, lane4, row * 256, FDAPDE_SYM_NT_C);
                            } else v[j - jg] = reinterpret_cast<const pg_u32x4*>(ev)[row * 64 + lane], c[j - jg] = ec[row * 64 + lane];
                        }
                    }
                };
                // the entries of `load`, applied to their own rows (registers) and to the rows of their columns (accumulator table)
                auto compute = [&](const auto& v, const auto& c, int jg, int e) {
                    constexpr int GS = sizeof(c) / sizeof(c[0]);
#pragma unroll
                    for (int j = jg; j < jg + GS; ++j) {
                        if (e < w[j]) {
                            const unsigned c_lo = c[j - jg] & 0xffffu, c_hi = c[j - jg] >> 16;
                            const double vx = __hiloint2double((int)v[j - jg].y, (int)v[j - jg].x), vy = __hiloint2double((int)v[j - jg].w, (int)v[j - jg].z);
                            yv[j] += vx * p_tab[c_lo] + vy * p_tab[c_hi];
                            double ps;
                            if constexpr (PS_REG) ps = psj[j];
                            else ps = P(j) * tscale;
                            const long long q_lo = (long long)(vx * ps), q_hi = (long long)(vy * ps);
                            // branch-free: what has no target here (a column of another workgroup) adds 0 to the lane's own slot, where it meets
                            // no other lane's (padding: value 0, column = own slot).  Measured against lanes skipping their add: 36.9 vs 38.0 us
                            const bool t_lo = c_lo < (unsigned)S, t_hi = c_hi < (unsigned)S;
                            __hip_atomic_fetch_add(&y_tab[t_lo ? c_lo : (unsigned)(j * T + tid)], t_lo ? q_lo : 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(&y_tab[t_hi ? c_hi : (unsigned)(j * T + tid)], t_hi ? q_hi : 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                };
                // all passes of the phase loaded, then multiplied.  Measured and dropped (C3, us per iteration; this form 36.9): two groups of
                // passes software-pipelined by hand 38.5; the same with a third buffer and x moved to global fp64 atomics to make room 67;
                // one lane per 128-byte line touching the lines of the group after next 58 (loads return in order: a touch that misses holds
                // up the real loads behind it); re-reading the last pair row of a finished pass and multiplying by 0, as the plain form does,
                // instead of skipping the pass 43; the magic-number conversion below instead of the compiler's fptosi 37.5
                // (two pair rows of every pass per step, -DFDAPDE_SYM_U=2, measured for 8 rows per thread: 3-D 754 k / 1.03 M DOFs 17.8 / 21.3 us per
                //  iteration either way, 2-D 1.0 M 12.07 -> 11.59: this form is bound by its LDS traffic, not by loads in flight -- left at 1)
                constexpr int U = !STREAM ? 1 : (NJ <= 4 ? FDAPDE_SYM_U : (NJ == 8 ? FDAPDE_SYM_U8 : 1));
                if constexpr (U == 1) {
                    for (int e = 0; e < mw; ++e) {
                        pg_u32x4 v[NJ];
                        uint32_t c[NJ];
                        load(v, c, J0, e);
                        compute(v, c, J0, e);
                    }
                } else {
                    for (int e = 0; e < mw; e += U) {
                        pg_u32x4 v[U][NJ];
                        uint32_t c[U][NJ];
#pragma unroll
                        for (int u = 0; u < U; ++u) load(v[u], c[u], J0, e + u);
#pragma unroll
                        for (int u = 0; u < U; ++u) compute(v[u], c[u], J0, e + u);
                    }
                }
            }
        };
        const bool late = a.wg_late != nullptr && a.wg_late[g] != 0;   // (uniform for the workgroup)
        if (!late) product(std::integral_constant<int, 0>{});
        {   // imports, four per lane at a time with their loads in flight together (a workgroup in the middle of a 3-D mesh imports ~3 000
            // entries: one after the other that was six dependent round trips per wavefront, measured as 1.5 us per 1 000 imports); a lane
            // re-reads what does not carry this iteration's tag in both halves yet, lanes that have theirs stop loading
            bool fail = false;
            for (int hb = wave * 64; hb < H; hb += 4 * T) {   // wave-uniform trip count
                const unsigned long long* gp[4];
                pg_v2u64 v[4];
                bool done[4];
                [[maybe_unused]] bool far[4];   // DIST: the entry comes from another rank (remote section of the fine-grained board)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int h = hb + k * T + lane;
                    done[k] = h >= H;
                    const int32_t pos = impl[done[k] ? 0 : h];
                    if constexpr (DIST) {
                        far[k] = pos >= a.n_board_local;
                        gp[k] = far[k] ? a.rboard + 2 * (size_t)(pos - a.n_board_local) : a.pboard + 2 * (size_t)pos;
                    } else
                        gp[k] = a.pboard + 2 * (size_t)pos;
                }
                if constexpr (DIST) {
                    // (most lanes read local entries: one batched agent-scope round for everybody, then the far lanes re-read at system scope)
                    granule_load2x4(gp[0], gp[1], gp[2], gp[3], v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (far[k]) v[k] = granule_load2_sys(gp[k]);
                } else
                    granule_load2x4(gp[0], gp[1], gp[2], gp[3], v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) done[k] = done[k] || granule_pair_ok(v[k], epoch);
                long long t_wait = 0;
                for (unsigned spins = 0; !__all(done[0] && done[1] && done[2] && done[3]); ++spins) {
                    if ((spins & 63u) == 63u) {   // bounded by time, checked now and then
                        const long long now = wall_clock64();
                        if (t_wait == 0) t_wait = now;
                        else if (now - t_wait > tmo) {
                            fail = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(2);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (!done[k]) {
                            if constexpr (DIST) v[k] = far[k] ? granule_load2_sys(gp[k]) : granule_load2(gp[k]);
                            else v[k] = granule_load2(gp[k]);
                            done[k] = granule_pair_ok(v[k], epoch);
                        }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int h = hb + k * T + lane;
                    if (h < H) p_tab[S + h] = granule_pair_f64(v[k]);
                }
                if (fail) break;
            }
            if (fail && lane == 0) fail_flag = 1;
        }
        const bool solo = !DIST && a.G == 1;   // ONE workgroup: no imports, no records -- two barriers per iteration instead of four
        if (!solo) {
            __syncthreads();
            if (fail_flag) {
                status = 3;
                break;
            }
        }
        if (late) product(std::integral_constant<int, 0>{});
        product(std::integral_constant<int, 1>{});
        if constexpr (SYM) {   // collect the transposed sums of the own rows; the table is zero again for the next iteration
            __syncthreads();
#pragma unroll
            for (int j = 0; j < R; ++j) {
                yv[j] += (double)y_tab[j * T + tid] * tinv;
                y_tab[j * T + tid] = 0;
            }
        }
        if (stamper) c1 = wall_clock64();
        // ---- partials of p.y, y.y and of the explicit r.r; one all-gather over the workgroups
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) s0 += P(j) * yv[j], s1 += yv[j] * yv[j];
        s0 = wave_sum64(s0), s1 = wave_sum64(s1);
        const double s2 = wave_sum64(rr_part);
        if (lane == 0) red[wave][0] = s0, red[wave][1] = s1, red[wave][2] = s2;
        __syncthreads();
        double solo_tot[3] = {0.0, 0.0, 0.0};
        const bool flat = DIST && a.flat_gather != 0;   // (uniform for the launch)
        if (flat) {
            // ONE hop: the record goes into the flat section of EVERY rank's board (thread (k, q) pushes value k to rank q), then thread t
            // collects the records t, t + 512, ... of all ranks' workgroups from the LOCAL board; the same order everywhere
            if (tid < 3) {
                double v = 0;
#pragma unroll
                for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
                tot[tid] = v;
            }
            __syncthreads();
            const size_t fbuf = (size_t)2 * a.world * 8 + (size_t)(it & 1) * a.G_tot * 8;
            if (tid < 3 * a.world) {
                const int k = tid % 3, q = tid / 3;
                publish_f64_x4_sys(a.peer_dboard[q] + fbuf + (size_t)(a.g_base + g) * 8 + 2 * k, epoch, tot[k]);
            }
            if (a.G_tot <= (W / 2) * 64) prefetch_next();   // (beyond that the touching wavefronts poll records themselves: their polls would queue behind the touches)
            double v0 = 0, v1 = 0, v2 = 0;
            bool fail = false;
            const int per_lane = (a.G_tot + T - 1) / T;
            if (wave * 64 < a.G_tot) {   // wave-uniform
                for (int rsel = 0; rsel < per_lane; ++rsel) {
                    const int w = rsel * T + tid;
                    const unsigned long long* gp = a.peer_dboard[a.rank] + fbuf + (size_t)(w < a.G_tot ? w : 0) * 8;
                    pg_v2u64 q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0};
                    bool done = w >= a.G_tot;
                    long long t_wait = 0;
                    for (unsigned spins = 0;; ++spins) {
                        if (!done) {
                            granule_load6_sys(gp, q0, q1, q2);
                            done = granule_pair_ok(q0, epoch) && granule_pair_ok(q1, epoch) && granule_pair_ok(q2, epoch);
                        }
                        if (__all(done)) break;
                        if ((spins & 63u) == 63u) {
                            const long long now = wall_clock64();
                            if (t_wait == 0) t_wait = now;
                            else if (now - t_wait > tmo) {
                                fail = true;
                                break;
                            }
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (w < a.G_tot && !fail) v0 += granule_pair_f64(q0), v1 += granule_pair_f64(q1), v2 += granule_pair_f64(q2);
                    if (fail) break;
                }
            }
            if (fail && lane == 0) fail_flag = 1;
            v0 = wave_sum64(v0), v1 = wave_sum64(v1), v2 = wave_sum64(v2);
            __syncthreads();   // (the values in tot / red have been consumed)
            if (lane == 0) red[wave][0] = v0, red[wave][1] = v1, red[wave][2] = v2;
            __syncthreads();
            if (tid < 3) {
                double v = 0;
#pragma unroll
                for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
                tot[tid] = v;
            }
        } else if (!DIST && a.G == 1) {
            // ONE workgroup (systems of a few thousand rows -- most of what the reference's own users solve): its sums are the totals, no record
            // is published or polled (a granule round trip is ~2 us of a ~3.5 us iteration); the same bits as through the board
            // (every thread adds the W partials itself, in the order one thread per sum did when the totals went through `tot` and a second
            //  barrier: the same bits; `red` is rewritten behind the next iteration's first barrier)
#pragma unroll
            for (int ww = 0; ww < W; ++ww) solo_tot[0] += red[ww][0], solo_tot[1] += red[ww][1], solo_tot[2] += red[ww][2];
            prefetch_next();
        } else {
        unsigned long long* dslot = a.dboard + (size_t)(it & 1) * a.G * kDotStride;
        if (tid < 3) {
            double v = 0;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
            publish_f64_x4(dslot + (size_t)g * kDotStride + 2 * tid, epoch, v);
        }
        prefetch_next();
        {   // thread t collects workgroup t's three sums (a lane re-reads its record until all six tags match, then stops loading); every
            // workgroup adds the G records in the same order.  A two-level form (groups of 8 / 16 / 32 workgroups handled by one wavefront,
            // then the group sums) was measured and dropped: a granule hop costs ~4 us under this load, two of them 7.9 / 9.8 / 12.0 us
            // against 4.6 us for the flat sweep on C2 (246 workgroups).  The records as [sum][workgroup] (a polling wavefront's three loads cover whole
            // lines: a third of the line requests) were measured too (round 5): C2 5.9 -> 6.5 us per iteration, C3 29.8 -> 30.4 -- a record's three sums
            // leave as three stores to three lines instead of one, and the reader needs all of them.
            double v0 = 0, v1 = 0, v2 = 0;
            bool fail = false;
            // gather_waves: how many wavefronts poll (4: thread t takes workgroup t's record; 1: wavefront 0 takes them all, 4 per lane)
            const int per_lane = a.gather_waves == 1 ? (a.G + 63) / 64 : 1;
            if (a.gather_waves == 1 ? wave == 0 : wave * 64 < a.G) {   // wave-uniform
                for (int rsel = 0; rsel < per_lane; ++rsel) {
                    const int w = a.gather_waves == 1 ? rsel * 64 + lane : tid;
                    const unsigned long long* gp = dslot + (size_t)(w < a.G ? w : 0) * kDotStride;
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
                            else if (now - t_wait > tmo) {
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
            if constexpr (DIST || kGather3) {
                __syncthreads();   // the partials in red have been consumed
                if (lane == 0) red[wave][0] = v0, red[wave][1] = v1, red[wave][2] = v2;
                __syncthreads();
                if (tid < 3) {
                    double v = 0;
#pragma unroll
                    for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
                    tot[tid] = v;
                }
            } else if (lane == 0)   // (a table of its own: no barrier in front; added up by every thread behind the ONE barrier below)
                red2[wave][0] = v0, red2[wave][1] = v1, red2[wave][2] = v2;
        }
        if constexpr (DIST) {
            // second level: the rank's sums (identical in all of its workgroups) go to every rank -- pushed by workgroup 0 -- and every
            // workgroup of every rank adds the `world` rank records in one fixed order (lane q holds rank q; the same butterfly everywhere)
            __syncthreads();
            const bool lfail = fail_flag != 0;   // (uniform; a rank that failed locally publishes nothing: the others time out and give up too)
            const size_t rbuf = (size_t)(it & 1) * a.world * 8;
            if (g == 0 && !lfail && tid < 3 * a.world) {
                const int k = tid % 3, q = tid / 3;
                publish_f64_x4_sys(a.peer_dboard[q] + rbuf + (size_t)a.rank * 8 + 2 * k, epoch, tot[k]);
            }
            __syncthreads();   // (tot is rewritten below)
            if (wave == 0 && !lfail) {
                const unsigned long long* gp = a.peer_dboard[a.rank] + rbuf + (size_t)(lane < a.world ? lane : 0) * 8;
                pg_v2u64 q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0};
                bool done = lane >= a.world, fail = false;
                long long t_wait = 0;
                for (unsigned spins = 0;; ++spins) {
                    if (!done) {
                        granule_load6_sys(gp, q0, q1, q2);
                        done = granule_pair_ok(q0, epoch) && granule_pair_ok(q1, epoch) && granule_pair_ok(q2, epoch);
                    }
                    if (__all(done)) break;
                    if ((spins & 63u) == 63u) {
                        const long long now = wall_clock64();
                        if (t_wait == 0) t_wait = now;
                        else if (now - t_wait > tmo) {
                            fail = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const bool on = lane < a.world && !fail;
                const double w0 = wave_sum64(on ? granule_pair_f64(q0) : 0.0), w1 = wave_sum64(on ? granule_pair_f64(q1) : 0.0),
                             w2 = wave_sum64(on ? granule_pair_f64(q2) : 0.0);
                if (lane == 0) {
                    tot[0] = w0, tot[1] = w1, tot[2] = w2;
                    if (fail) fail_flag = 1;
                }
            }
        }
        }
        if (!solo) {
            __syncthreads();
            if (fail_flag) {
                status = 3;
                break;
            }
            if constexpr (!DIST && !kGather3) {   // the wavefronts' sums of the records, in the order the one summing thread per total used (same bits; one barrier, not three)
#pragma unroll
                for (int ww = 0; ww < W; ++ww) solo_tot[0] += red2[ww][0], solo_tot[1] += red2[ww][1], solo_tot[2] += red2[ww][2];
            }
        }
        const bool own_tot = !DIST && (!kGather3 || solo);
        const double pAp = own_tot ? solo_tot[0] : tot[0], yy = own_tot ? solo_tot[1] : tot[1];
        rr = own_tot ? solo_tot[2] : tot[2];
        if (!DIST && a.direct && it == 0) bb = rr;   // x0 = 0: r0 = b~
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
        if constexpr (WIDE) {   // x in HBM: the loads of all rows first (yv is free from here on: its registers take them), then the updates
#pragma unroll
            for (int j = 0; j < R; ++j) rv[j] -= alpha * yv[j], rr_part += rv[j] * rv[j];
#pragma unroll
            for (int j = 0; j < R; ++j) yv[j] = xs[j * T];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const double pj = P(j);
                xs[j * T] = yv[j] + alpha * pj;
                p_tab[j * T + tid] = rv[j] + beta * pj;   // (every read of this iteration's table lies behind a barrier)
            }
        } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const double pj = P(j);
            xv[j] += alpha * pj;
            rv[j] -= alpha * yv[j];
            if constexpr (PLDS) p_tab[j * T + tid] = rv[j] + beta * pj;   // (every read of this iteration's table lies behind a barrier)
            else pv[j] = rv[j] + beta * pj;
            rr_part += rv[j] * rv[j];
        }
        }
        ++it;
        if (stamper) {
            const long long c3 = wall_clock64();
            t_spmv += c1 - c0, t_gather += c2 - c1, t_update += c3 - c2, ++n_stamped;
        }
    }
    if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (touches of an iteration that did not come)
    if (status != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            int32_t d;
            if constexpr (PLDS) d = a.slot_dof[(size_t)g * S + j * T + tid];
            else d = dof[j];
            double xj;
            if constexpr (WIDE) xj = xs[j * T];
            else xj = xv[j];
            if (!DIST && a.direct) {
                if (d >= 0) a.x_ext[a.i2e[d]] = a.scale[d] * xj + 0.0;   // (k_cols_finish's expression)
            } else if (d >= 0) {
                a.x_out[d] = xj;
                if (!DIST && a.u_out != nullptr) a.u_out[d] = a.scale[d] * xj + 0.0;
            }
        }
    }
    if (!DIST && a.direct) {
        // the record goes out behind the solution: every wavefront's stores are drained, then ONE thread publishes the status at system scope
        // (the host may be spinning on it instead of waiting for the stream)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            a.rec[0] = (double)it, a.rec[1] = rr, a.rec[2] = bb;
            __atomic_store_n(reinterpret_cast<volatile long long*>(a.rec + 3), (long long)(status + 1), __ATOMIC_RELEASE);
        }
    } else if (g == 0 && tid == 0 && status != 3) {
        a.sc[3] = rr;
        a.ctl[0] = status == 1 ? 1 : 0, a.ctl[1] = it, a.ctl[2] = status == 2 ? 1 : 0;
    }
    if constexpr (!DIST) {
        if (a.hrec != nullptr && g == 0 && tid == 0) {   // (G == 1: this workgroup's status is the launch's)
            a.hrec[0] = status == 1 ? 1.0 : 0.0, a.hrec[1] = (double)it, a.hrec[2] = status == 2 ? 1.0 : 0.0, a.hrec[3] = status == 3 ? 1.0 : 0.0;
            a.hrec[4] = (double)a.ctl[4], a.hrec[5] = bb, a.hrec[6] = rr;
        }
    }
    if (stamper) {
        double* st = a.stats + 4 * (size_t)g;
        st[0] = (double)n_stamped, st[1] = (double)t_spmv, st[2] = (double)t_gather, st[3] = (double)t_update;
    }
    if (status == 3 && tid == 0 && !(!DIST && a.direct)) atomicExch(a.ctl + 3, 1);
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
static __global__ __launch_bounds__(kPersistT) void k_spmv_blocked(BlockedSpmvArgs a) {
    constexpr int T = kPersistT, W = T / 64, S = R * T;
    extern __shared__ __attribute__((aligned(128))) double lds[];   // (16-byte LDS reads of the resident blocks: a base the static arrays left 8-byte aligned halves their rate)
    __shared__ double red[W][2];
    if (a.stop && __syncthreads_or(*a.stop != 0)) return;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsl = a.nsl;
    double* p_tab = lds;
    const int H = a.imp_off[g + 1] - a.imp_off[g];
    const int64_t e0 = a.ell_off[g];
    const double2* gv = reinterpret_cast<const double2*>(a.ell_val + e0);
    const uint32_t* gc = reinterpret_cast<const uint32_t*>(a.ell_code + e0);
    // streaming form: the block is read through raw buffer loads (descriptor in scalar registers: base = this workgroup's block)
    // (records end with the arrays -- all blocks + 256 entries of slack: a touch past the last block reads 0 instead of faulting)
    const int64_t e_left = a.ell_off[a.G] + 256 - e0;
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(gv), 0, (int)min(e_left * 8, (int64_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(gc), 0, (int)min(e_left * 2, (int64_t)0x7fffffff), 0x00020000);
    const int lane16 = lane * 16, lane4 = lane * 4;
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
    auto product = [&](auto) {   // all R passes at once: nothing to overlap with here, and the loads in flight per wavefront are what
                                 // bounds the stream (C5, 2 workgroups per CU: R / 2 loads at a time 348 us per SpMV, R loads 336)
        constexpr int J0 = 0, J1 = R;
        int mw = 0;
#pragma unroll
        for (int j = J0; j < J1; ++j) mw = max(mw, w[j]);
        // a pass that has run out of entries is skipped (wave-uniform): the rows are sorted by length and P2 rows range from 10 to 60+
        // entries, so the passes of a wavefront differ widely in width -- re-reading the last pair row of the narrow ones, as the
        // persistent CG's plain form does, would issue up to twice the loads; raw buffer loads: scalar row offset + constant lane offset
        constexpr int U = 2;   // pair rows of every pass per step: U R loads in flight per wavefront (the kernel needs ~60 registers; what
                               // bounds it is the bytes in flight per CU).  C5, us per SpMV: U = 1 336, 2 327, 3 345, 4 342
        for (int e = 0; e < mw; e += U) {
            pg_u32x4 v[U][J1 - J0];
            uint32_t c[U][J1 - J0];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = J0; j < J1; ++j) {
                    if (e + u < w[j]) {
                        const int row = o0[j] + e + u;
                        v[u][j - J0] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, lane16, row * 1024, 0);
                        c[u][j - J0] = __builtin_amdgcn_raw_buffer_load_b32(rs_c, lane4, row * 256, 0);
                    }
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = J0; j < J1; ++j) {
                    if (e + u < w[j]) {
                        const pg_u32x4 q = v[u][j - J0];
                        const double vx = __hiloint2double((int)q.y, (int)q.x), vy = __hiloint2double((int)q.w, (int)q.z);
                        yv[j] += vx * p_tab[c[u][j - J0] & 0xffffu] + vy * p_tab[c[u][j - J0] >> 16];
                    }
                }
        }
    };
    product(std::integral_constant<int, 0>{});
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
// amax_bits != nullptr (zeroed by the caller): also max |value| as a bit pattern (non-negative doubles order like their bits)
// (four entries per thread, a workgroup covering 1024 consecutive ones: the four index loads, then the four gathers, in flight together)
static __global__ __launch_bounds__(256) void k_persist_fill(int64_t n, const int32_t* src, const double* scaled_full, double* out,
                                                      unsigned long long* amax_bits) {
    const int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    int32_t s4[4];
    double v4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s4[k] = i0 + 256 * k < n ? src[i0 + 256 * k] : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) v4[k] = s4[k] >= 0 ? scaled_full[s4[k]] : 0.0;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + 256 * k < n) out[i0 + 256 * k] = v4[k];
        v = fmax(v, fabs(v4[k]));
    }
    if (amax_bits != nullptr) {
        double m = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        // most wavefronts find a maximum at least as large already in place and skip the atomic
        if ((threadIdx.x & 63) == 0 && bits > __hip_atomic_load(amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_bits, bits);
    }
}

// the same straight from the UNSCALED matrix: ell_val[e] = scale[row] * A[src] * scale[col] (the expression, and therefore the bits, of
// k_scale_matrix), so that a solve that runs as one persistent launch never writes or reads the scaled full-pattern copy (205 MB written +
// gathered on C3).  One wavefront per slice of 64 slots: lane l owns slot 64 q + l, hence knows its row; its entries are the lane pairs
// (2 l, 2 l + 1) of the slice's pair rows.
// (slice q of workgroup g, one wavefront; returns the lane's max |value|)
__device__ __forceinline__ double persist_fill_scaled_slice(int g, int q, int lane, int32_t nsl, const int64_t* ell_off, const int32_t* sl_off, const int32_t* slot_dof,
                                                            const int32_t* src, const int32_t* col, const double* A, const double* scale, double* out) {
    const int S = nsl * 64;
    const int32_t d = slot_dof[(size_t)g * S + q * 64 + lane];
    const double si = d >= 0 ? scale[d] : 0.0;
    const int32_t* slo = sl_off + (size_t)g * (nsl + 1);
    const int o0 = slo[q], w = slo[q + 1] - o0;
    const int64_t base = ell_off[g] + (int64_t)o0 * 128 + 2 * lane;
    double m = 0.0;
    constexpr int U = 4;   // pair rows per step: 8 independent value gathers + 8 scale gathers in flight per lane
    for (int pr = 0; pr < w; pr += U) {
        int2 s2[U], c2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = base + (int64_t)min(pr + u, w - 1) * 128;
            s2[u] = *reinterpret_cast<const int2*>(src + e), c2[u] = *reinterpret_cast<const int2*>(col + e);
        }
        double a0[U], a1[U], t0[U], t1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a0[u] = s2[u].x >= 0 ? A[s2[u].x] : 0.0, a1[u] = s2[u].y >= 0 ? A[s2[u].y] : 0.0;
            t0[u] = scale[c2[u].x], t1[u] = scale[c2[u].y];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pr + u < w) {
                const double v0 = s2[u].x >= 0 ? si * a0[u] * t0[u] : 0.0, v1 = s2[u].y >= 0 ? si * a1[u] * t1[u] : 0.0;
                *reinterpret_cast<double2*>(out + base + (int64_t)(pr + u) * 128) = make_double2(v0, v1);
                m = fmax(m, fmax(fabs(v0), fabs(v1)));
            }
        }
    }
    return m;
}
static __global__ __launch_bounds__(256) void k_persist_fill_scaled(int32_t G, int32_t nsl, const int64_t* ell_off, const int32_t* sl_off, const int32_t* slot_dof,
                                                                    const int32_t* src, const int32_t* col, const double* A, const double* scale, double* out,
                                                                    unsigned long long* amax_bits) {
    const int per = (nsl + 3) / 4;
    const int g = blockIdx.x / per, q = (blockIdx.x % per) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G || q >= nsl) return;   // (wave-uniform)
    double m = persist_fill_scaled_slice(g, q, lane, nsl, ell_off, sl_off, slot_dof, src, col, A, scale, out);
    if (amax_bits != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        if (lane == 0 && bits > __hip_atomic_load(amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_bits, bits);
    }
}

// column DOF of every ELL entry (0 in padding), once per layout: lets the fill gather A[src] and scale[col] independently
static __global__ __launch_bounds__(256) void k_persist_ell_col(int64_t n, const int32_t* src, const int32_t* colidx, int32_t* col) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) col[i] = src[i] >= 0 ? colidx[src[i]] : 0;
}

}  // namespace fdapde_hip
#endif
