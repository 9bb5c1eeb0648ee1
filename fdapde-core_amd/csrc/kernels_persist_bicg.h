// kernels_persist_bicg.h -- the whole Jacobi-BiCGStab solve of a non-symmetric system (advection) as ONE launch, on the layout and the
// hand-off machinery of kernels_persist.h (plain storage): x, r, r0, p, v, t of a workgroup's rows live in registers, the matrix block is
// resident in LDS or streams once per operator application, neighbours exchange entries through granule boards.  An iteration is TWO
// operator applications and TWO all-gathers:
//     p = r + beta (p - omega v)                       (registers)
//     v = A p            gather 1: (r0.v | r.r of the previous iteration, summed explicitly)      -> alpha, stop test
//     s = r - alpha v                                  (kept in r's registers)
//     t = A s            gather 2: (t.s, t.t, r0.s, r0.t)                                         -> omega, rho' = r0.s - omega r0.t
//     x += alpha p + omega s ; r = s - omega t
// The multi-launch kernels (kernels_krylov.h k_bicg_*) sum r0.r explicitly after the update; here rho' comes from the two dots gathered
// with omega -- the same number in exact arithmetic, one all-gather less -- and the explicit r.r rides in the next iteration's first
// gather, so the stop test runs half an iteration late on the explicitly summed value (x is already final when it fires).
// Epochs: application / gather k of iteration it carries tag epoch0 + 2 it + k (k = 1, 2); dot records are 4 doubles wide, double-buffered
// by k.  DIST: the row-distributed multi-GPU form (PersistArgs).
#ifndef FDAPDE_KERNELS_PERSIST_BICG_H
#define FDAPDE_KERNELS_PERSIST_BICG_H

#include "kernels_persist.h"

namespace fdapde_hip {

__device__ __forceinline__ void granule_load8(const unsigned long long* p, pg_v2u64& a, pg_v2u64& b, pg_v2u64& c, pg_v2u64& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void granule_load8_sys(const unsigned long long* p, pg_v2u64& a, pg_v2u64& b, pg_v2u64& c, pg_v2u64& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}

// R rows per thread (2, 4, 8).  ctl / sc on return as the multi-launch loop leaves them: ctl[0] converged, [1] iterations, [2] breakdown,
// sc[3] = final r.r; [3] hand-off timeout (then nothing else was written).
template <int R, bool STREAM, bool DIST>
static __global__ __launch_bounds__(kPersistT) void k_bicg_persist(PersistArgs a) {
    constexpr int T = kPersistT, W = T / 64, S = R * T, RI = R / 2;
    extern __shared__ __attribute__((aligned(128))) double lds[];   // (16-byte LDS reads of the resident blocks: a base the static arrays left 8-byte aligned halves their rate)
    __shared__ double red[W][4];
    __shared__ double tot[4];
    __shared__ int32_t fail_flag;
    __shared__ unsigned pf_dump[64];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (!DIST) {
        if (a.n_cols > 1) {   // several right-hand sides at once: this workgroup belongs to column blockIdx.y (kernels_persist.h)
            const size_t col = blockIdx.y;
            a.r_in += col * a.col_stride, a.x_out += col * a.col_stride, a.sc += 4 * col, a.ctl += 4 * col;
            if (a.x != nullptr) a.x += col * a.col_stride;
            a.pboard += col * a.board_stride, a.dboard += col * a.board_stride;
        }
    }
    const int nsl = a.nsl;
    const int H = a.imp_off[g + 1] - a.imp_off[g], E = a.exp_off[g + 1] - a.exp_off[g];
    double* p_tab = lds;                                                      // [S + imp_cap]
    double2* ev = reinterpret_cast<double2*>(p_tab + (S + a.imp_cap));        // [lds_cap / 2] entry pairs (resident form)
    uint32_t* ec = reinterpret_cast<uint32_t*>(ev + a.lds_cap / 2);           // [lds_cap / 2] code pairs
    int32_t* impl = reinterpret_cast<int32_t*>(ec + a.lds_cap / 2);           // [imp_cap]
    uint16_t* expl = reinterpret_cast<uint16_t*>(impl + a.imp_cap);           // [E]
    const int64_t e0 = a.ell_off[g];
    const double2* gv = reinterpret_cast<const double2*>(a.ell_val + e0);
    const uint32_t* gc = reinterpret_cast<const uint32_t*>(a.ell_code + e0);
    const int64_t e_left = a.ell_off[a.G] + 256 - e0;
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(gv), 0, (int)min(e_left * 8, (int64_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(gc), 0, (int)min(e_left * 2, (int64_t)0x7fffffff), 0x00020000);
    const int lane16 = lane * 16, lane4 = lane * 4;
    for (int i = tid; i < H; i += T) impl[i] = a.imp_pos[a.imp_off[g] + i];
    for (int i = tid; i < E; i += T) expl[i] = a.exp_slot[a.exp_off[g] + i];
    if (tid == 0) fail_flag = 0;
    const int32_t* slo = a.sl_off + (size_t)g * (nsl + 1);
    if constexpr (!STREAM) {
        const int n_pairs = slo[nsl] * 64;
        for (int i = tid; i < n_pairs; i += T) ev[i] = gv[i], ec[i] = gc[i];
    }
    int o0[R], w[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        o0[j] = __builtin_amdgcn_readfirstlane(slo[j * W + wave]);
        w[j] = __builtin_amdgcn_readfirstlane(slo[j * W + wave + 1]) - o0[j];
    }
    double xv[R], rv[R], qv[R], pv[R], vv[R], tv[R];   // x, r (s between the two applications), r0, p, v = A p, t = A s
    int32_t dof[R];
    double rr_part = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int32_t d = a.slot_dof[(size_t)g * S + j * T + tid];
        const bool on = d >= 0;
        dof[j] = d;
        rv[j] = on ? a.r_in[d] : 0.0;
        xv[j] = on && a.x != nullptr ? a.x[d] : 0.0;
        qv[j] = rv[j], pv[j] = rv[j], vv[j] = 0.0, tv[j] = 0.0;
        rr_part += rv[j] * rv[j];
    }
    const double bb = a.sc[0];
    int it = 0, status = 0;   // status: 1 converged, 2 breakdown, 3 hand-off timeout
    bool rr_pending = false;  // a breakdown right after an update: r.r of the returned x still to be summed over the workgroups
    double rr = 0, rho = 0, rho_old = 1.0, alpha = 1.0, omega = 1.0;
    long long tmo = DIST ? (long long)a.timeout_first_ticks : (long long)a.timeout_ticks;
    __syncthreads();

    // ---- y = (I + At_offdiag) src: src of the own rows into the LDS table, exported entries onto the board(s), the passes without imports,
    //      the imports, the other passes.  Returns false when a wait timed out (fail_flag raised; the caller leaves the loop).
    auto apply = [&](const double (&src)[R], double (&y)[R], unsigned epoch) -> bool {
#pragma unroll
        for (int j = 0; j < R; ++j) p_tab[j * T + tid] = src[j];
        __syncthreads();
        for (int i0 = tid; i0 < E; i0 += 4 * T) {
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
        if constexpr (DIST) {
            const int re0 = a.rexp_off[g], RE = a.rexp_off[g + 1] - re0;
            for (int i = tid; i < RE; i += T)
                publish_f64_x4_sys(a.peer_pboard[a.rexp_peer[re0 + i]] + 2 * (size_t)a.rexp_pos[re0 + i], epoch, p_tab[a.rexp_slot[re0 + i]]);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) y[j] = src[j];   // unit diagonal of the scaled system
        auto product = [&](auto phase) {
            constexpr int J0 = decltype(phase)::value ? RI : 0, J1 = decltype(phase)::value ? R : RI, NJ = J1 - J0;
            int mw = 0;
#pragma unroll
            for (int j = J0; j < J1; ++j) mw = max(mw, w[j]);
            // a pass that has run out of entries is skipped (wave-uniform; P2 rows differ widely in length).  U pair rows of every pass per
            // step: with 2 or 4 rows per thread a phase has 1 or 2 passes, i.e. 1 or 2 loads in flight per wavefront -- the stream then runs at
            // the rate of the memory latency, not of the memory (3-D P2 227 k DOFs, 4 rows per thread: 62.5 -> 41.0 us per iteration with U = 4; 2-D P2 361 k 30.4 -> 23.7).
            // With 8 rows per thread (4 passes per phase) U = 2 gains on long rows (3-D P2 754 k: 91 -> 83) and loses on short ones (P1 1.0 M: 53.1 ->
            // 54.7, 2-D 31.0 -> 32.9): left at 1
            constexpr int U = STREAM ? (NJ >= 4 ? 1 : 8 / NJ) : 1;
            for (int e = 0; e < mw; e += U) {
                pg_u32x4 v[U][NJ];
                uint32_t c[U][NJ];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = J0; j < J1; ++j) {
                        if (e + u < w[j]) {
                            const int row = o0[j] + e + u;
                            if constexpr (STREAM) {
                                v[u][j - J0] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, lane16, row * 1024, 0);
                                c[u][j - J0] = __builtin_amdgcn_raw_buffer_load_b32(rs_c, lane4, row * 256, 0);
                            } else
                                v[u][j - J0] = reinterpret_cast<const pg_u32x4*>(ev)[row * 64 + lane], c[u][j - J0] = ec[row * 64 + lane];
                        }
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = J0; j < J1; ++j) {
                        if (e + u < w[j]) {
                            const pg_u32x4 q = v[u][j - J0];
                            const double vx = __hiloint2double((int)q.y, (int)q.x), vy = __hiloint2double((int)q.w, (int)q.z);
                            y[j] += vx * p_tab[c[u][j - J0] & 0xffffu] + vy * p_tab[c[u][j - J0] >> 16];
                        }
                    }
            }
        };
        const bool late = a.wg_late != nullptr && a.wg_late[g] != 0;   // (uniform for the workgroup)
        if (!late) product(std::integral_constant<int, 0>{});
        {
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
                granule_load2x4(gp[0], gp[1], gp[2], gp[3], v[0], v[1], v[2], v[3]);
                if constexpr (DIST) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (far[k]) v[k] = granule_load2_sys(gp[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) done[k] = done[k] || granule_pair_ok(v[k], epoch);
                long long t_wait = 0;
                for (unsigned spins = 0; !__all(done[0] && done[1] && done[2] && done[3]); ++spins) {
                    if ((spins & 63u) == 63u) {
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
        __syncthreads();
        if (fail_flag) return false;
        if (late) product(std::integral_constant<int, 0>{});
        product(std::integral_constant<int, 1>{});
        return true;
    };
    // ---- all-gather of up to four sums over all workgroups (of all ranks): every workgroup adds the records in the same order
    // (streaming forms: the first entry step of the next application touched while the workgroup waits for the records, kernels_persist.h)
    [[maybe_unused]] auto prefetch_next = [&]() {
        if constexpr (STREAM)
            if (a.pf_steps != 0) touch_first_step<RI, W>(slo, gv, gc, wave, lane, (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)pf_dump);
    };
    double gt[4] = {0.0, 0.0, 0.0, 0.0};   // the totals of the last gather, in every thread
    int gsel = 0;
    __shared__ double red_solo[2][W][4];
    auto gather = [&](double s0, double s1, double s2, double s3, int phase, unsigned epoch) -> bool {
        s0 = wave_sum64(s0), s1 = wave_sum64(s1), s2 = wave_sum64(s2), s3 = wave_sum64(s3);
        if (!DIST && a.G == 1) {
            // ONE workgroup: its sums are the totals.  One barrier per gather instead of three: the wavefronts' partials go into the table of this
            // gather's parity (rewritten two gathers on, behind the barrier of the gather in between), and every thread adds them itself in the order the
            // one summing thread per total uses elsewhere -- the same bits.  The barrier also ends the table reads of the application before.
            const int b = gsel & 1;
            ++gsel;
            if (lane == 0) red_solo[b][wave][0] = s0, red_solo[b][wave][1] = s1, red_solo[b][wave][2] = s2, red_solo[b][wave][3] = s3;
            __syncthreads();
            gt[0] = gt[1] = gt[2] = gt[3] = 0.0;
#pragma unroll
            for (int ww = 0; ww < W; ++ww)
                gt[0] += red_solo[b][ww][0], gt[1] += red_solo[b][ww][1], gt[2] += red_solo[b][ww][2], gt[3] += red_solo[b][ww][3];
            prefetch_next();
            return true;
        }
        auto fin = [&]() -> bool {   // (behind the barrier that follows the totals)
            gt[0] = tot[0], gt[1] = tot[1], gt[2] = tot[2], gt[3] = tot[3];
            return fail_flag == 0;
        };
        __syncthreads();   // (the table reads of the application before, and the totals of the gather before, are done with)
        if (lane == 0) red[wave][0] = s0, red[wave][1] = s1, red[wave][2] = s2, red[wave][3] = s3;
        __syncthreads();
        if (DIST && a.flat_gather != 0) {   // ONE hop (kernels_persist.h): every workgroup's record to every rank, all G_tot records gathered locally
            if (tid < 4) {
                double v = 0;
#pragma unroll
                for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
                tot[tid] = v;
            }
            __syncthreads();
            const size_t fbuf = (size_t)2 * a.world * 8 + (size_t)phase * a.G_tot * 8;
            if (tid < 4 * a.world) {
                const int k = tid & 3, q = tid >> 2;
                publish_f64_x4_sys(a.peer_dboard[q] + fbuf + (size_t)(a.g_base + g) * 8 + 2 * k, epoch, tot[k]);
            }
            if (a.G_tot <= (W / 2) * 64) prefetch_next();   // (beyond that the touching wavefronts poll records themselves: their polls would queue behind the touches)
            double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            bool fail = false;
            const int per_lane = (a.G_tot + T - 1) / T;
            if (wave * 64 < a.G_tot) {
                for (int rsel = 0; rsel < per_lane; ++rsel) {
                    const int wq = rsel * T + tid;
                    const unsigned long long* gp = a.peer_dboard[a.rank] + fbuf + (size_t)(wq < a.G_tot ? wq : 0) * 8;
                    pg_v2u64 q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0}, q3 = {0, 0};
                    bool done = wq >= a.G_tot;
                    long long t_wait = 0;
                    for (unsigned spins = 0;; ++spins) {
                        if (!done) {
                            granule_load8_sys(gp, q0, q1, q2, q3);
                            done = granule_pair_ok(q0, epoch) && granule_pair_ok(q1, epoch) && granule_pair_ok(q2, epoch) && granule_pair_ok(q3, epoch);
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
                    if (wq < a.G_tot && !fail) v0 += granule_pair_f64(q0), v1 += granule_pair_f64(q1), v2 += granule_pair_f64(q2), v3 += granule_pair_f64(q3);
                    if (fail) break;
                }
            }
            if (fail && lane == 0) fail_flag = 1;
            v0 = wave_sum64(v0), v1 = wave_sum64(v1), v2 = wave_sum64(v2), v3 = wave_sum64(v3);
            __syncthreads();
            if (lane == 0) red[wave][0] = v0, red[wave][1] = v1, red[wave][2] = v2, red[wave][3] = v3;
            __syncthreads();
            if (tid < 4) {
                double v = 0;
#pragma unroll
                for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
                tot[tid] = v;
            }
            __syncthreads();
            return fin();
        }
        unsigned long long* dslot = a.dboard + (size_t)phase * a.G * 8;
        if (tid < 4) {
            double v = 0;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
            publish_f64_x4(dslot + (size_t)g * 8 + 2 * tid, epoch, v);
        }
        prefetch_next();
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        bool fail = false;
        if (wave * 64 < a.G) {   // wave-uniform: thread t takes workgroup t's record
            const int wq = tid;
            const unsigned long long* gp = dslot + (size_t)(wq < a.G ? wq : 0) * 8;
            pg_v2u64 q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0}, q3 = {0, 0};
            bool done = wq >= a.G;
            long long t_wait = 0;
            for (unsigned spins = 0;; ++spins) {
                if (!done) {
                    granule_load8(gp, q0, q1, q2, q3);
                    done = granule_pair_ok(q0, epoch) && granule_pair_ok(q1, epoch) && granule_pair_ok(q2, epoch) && granule_pair_ok(q3, epoch);
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
            if (wq < a.G && !fail) v0 = granule_pair_f64(q0), v1 = granule_pair_f64(q1), v2 = granule_pair_f64(q2), v3 = granule_pair_f64(q3);
        }
        if (fail && lane == 0) fail_flag = 1;
        v0 = wave_sum64(v0), v1 = wave_sum64(v1), v2 = wave_sum64(v2), v3 = wave_sum64(v3);
        __syncthreads();
        if (lane == 0) red[wave][0] = v0, red[wave][1] = v1, red[wave][2] = v2, red[wave][3] = v3;
        __syncthreads();
        if (tid < 4) {
            double v = 0;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) v += red[ww][tid];
            tot[tid] = v;
        }
        __syncthreads();
        if constexpr (DIST) {   // second level (kernels_persist.h): the rank's sums to every rank by workgroup 0, all add the rank records in rank order
            const bool lfail = fail_flag != 0;
            const size_t rbuf = (size_t)phase * a.world * 8;
            if (g == 0 && !lfail && tid < 4 * a.world) {
                const int k = tid & 3, q = tid >> 2;
                publish_f64_x4_sys(a.peer_dboard[q] + rbuf + (size_t)a.rank * 8 + 2 * k, epoch, tot[k]);
            }
            __syncthreads();
            if (wave == 0 && !lfail) {
                const unsigned long long* gp = a.peer_dboard[a.rank] + rbuf + (size_t)(lane < a.world ? lane : 0) * 8;
                pg_v2u64 q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0}, q3 = {0, 0};
                bool done = lane >= a.world, rfail = false;
                long long t_wait = 0;
                for (unsigned spins = 0;; ++spins) {
                    if (!done) {
                        granule_load8_sys(gp, q0, q1, q2, q3);
                        done = granule_pair_ok(q0, epoch) && granule_pair_ok(q1, epoch) && granule_pair_ok(q2, epoch) && granule_pair_ok(q3, epoch);
                    }
                    if (__all(done)) break;
                    if ((spins & 63u) == 63u) {
                        const long long now = wall_clock64();
                        if (t_wait == 0) t_wait = now;
                        else if (now - t_wait > tmo) {
                            rfail = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const bool on = lane < a.world && !rfail;
                const double w0 = wave_sum64(on ? granule_pair_f64(q0) : 0.0), w1 = wave_sum64(on ? granule_pair_f64(q1) : 0.0),
                             w2 = wave_sum64(on ? granule_pair_f64(q2) : 0.0), w3 = wave_sum64(on ? granule_pair_f64(q3) : 0.0);
                if (lane == 0) {
                    tot[0] = w0, tot[1] = w1, tot[2] = w2, tot[3] = w3;
                    if (rfail) fail_flag = 1;
                }
            }
            __syncthreads();
        }
        return fin();
    };

    for (;;) {
        const unsigned ep1 = a.epoch0 + 2u * (unsigned)it + 1u, ep2 = ep1 + 1u;
        if (a.debug_stall_it > 0 && it == a.debug_stall_it && g == a.G - 1) {   // (wave-uniform; tests)
            status = 3;
            break;
        }
        // ---- v = A p ; gather 1: r0.v and the explicit r.r of the residual this iteration starts from
        if (!apply(pv, vv, ep1)) {
            status = 3;
            break;
        }
        double d0 = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) d0 += qv[j] * vv[j];
        if (!gather(d0, rr_part, 0.0, 0.0, 0, ep1)) {
            status = 3;
            break;
        }
        const double r0v = gt[0];
        rr = gt[1];
        if (it == 0) rho = rr;   // r0 = r: rho = r0.r0
        if (rr <= a.tol2 * bb) {
            status = 1;
            break;
        }
        if (it >= a.maxit) break;
        if (rho == 0.0 || r0v == 0.0) {
            status = 2;
            break;
        }
        alpha = rho / r0v;
#pragma unroll
        for (int j = 0; j < R; ++j) rv[j] -= alpha * vv[j];   // s
        // ---- t = A s ; gather 2: t.s, t.t, r0.s, r0.t
        if (!apply(rv, tv, ep2)) {
            status = 3;
            break;
        }
        double e0s = 0, e1s = 0, e2s = 0, e3s = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) e0s += tv[j] * rv[j], e1s += tv[j] * tv[j], e2s += qv[j] * rv[j], e3s += qv[j] * tv[j];
        if (!gather(e0s, e1s, e2s, e3s, 1, ep2)) {
            status = 3;
            break;
        }
        const double ts = gt[0], tt = gt[1], r0s = gt[2], r0t = gt[3];
        omega = tt > 0.0 ? ts / tt : 0.0;
        rr_part = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            xv[j] += alpha * pv[j] + omega * rv[j];
            rv[j] -= omega * tv[j];
            rr_part += rv[j] * rv[j];
        }
        ++it;
        tmo = (long long)a.timeout_ticks;
        if (omega == 0.0) {
            status = 2, rr_pending = true;   // (x and r were just updated: their r.r has not been summed over the workgroups yet)
            break;
        }
        rho_old = rho, rho = r0s - omega * r0t;   // = r0.r of the new residual
        const double beta = (rho / rho_old) * (alpha / omega);
#pragma unroll
        for (int j = 0; j < R; ++j) pv[j] = rv[j] + beta * (pv[j] - omega * vv[j]);
    }
    if (rr_pending) {   // breakdown noticed right AFTER an update (omega = 0): the residual norm of the returned x, summed explicitly.
        // Every workgroup takes this branch together (the scalars are identical everywhere).  Its tag is the first gather's of the iteration
        // that did not start -- nobody has published under it.  A breakdown noticed BEFORE the update (rho = 0 or r0.v = 0) needs no gather:
        // rr already is the global sum gather 1 of that iteration delivered, and a second record under the SAME tag could be picked up half
        // old, half new by a poller.
        if (gather(0.0, rr_part, 0.0, 0.0, 0, a.epoch0 + 2u * (unsigned)it + 1u)) rr = gt[1];
        else status = 3;
    }
    if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (touches of an application that did not come)
    if (status != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (dof[j] >= 0) {
                a.x_out[dof[j]] = xv[j];
                if (!DIST && a.u_out != nullptr) a.u_out[dof[j]] = a.scale[dof[j]] * xv[j] + 0.0;   // (the epilogue of a small solve: kernels_persist.h)
            }
    }
    if (g == 0 && tid == 0 && status != 3) {
        a.sc[3] = rr;
        a.ctl[0] = status == 1 ? 1 : 0, a.ctl[1] = it, a.ctl[2] = status == 2 ? 1 : 0;
    }
    if constexpr (!DIST) {
        if (a.hrec != nullptr && g == 0 && tid == 0) {   // (one-workgroup launches of fdapde_solve: the outcome into pinned host memory, kernels_persist.h)
            a.hrec[0] = status == 1 ? 1.0 : 0.0, a.hrec[1] = (double)it, a.hrec[2] = status == 2 ? 1.0 : 0.0, a.hrec[3] = status == 3 ? 1.0 : 0.0;
            a.hrec[4] = (double)a.ctl[4], a.hrec[5] = bb, a.hrec[6] = rr;
        }
    }
    if (status == 3 && tid == 0) atomicExch(a.ctl + 3, 1);
}

}  // namespace fdapde_hip
#endif
