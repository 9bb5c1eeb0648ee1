// kernels_reduce.h -- wavefront / workgroup / team reductions shared by the SpMV and Krylov kernels; see kernels.h
#ifndef FDAPDE_KERNELS_REDUCE_H
#define FDAPDE_KERNELS_REDUCE_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "internal.h"

namespace fdapde_hip {

// ---------------------------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------------------------
// Sum over the wavefront (valid in lane 0 -- in every lane, in fact): the butterfly v += v[lane ^ o], o = 32 .. 1, on the VALU -- gfx950's
// v_permlane32_swap / v_permlane16_swap for the two widest steps, DPP moves inside the rows of 16 -- instead of the six ds_bpermute pairs a shuffle
// loop compiles to (LDS crossbar, ~100 cycles each).  Lane 0's operands, hence its bits, are those of the __shfl_down tree this replaces
// (kernels_persist.h wave_sum64 has the argument).
template <int CTRL> __device__ __forceinline__ double reduce_dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int O> __device__ __forceinline__ double reduce_swap_sum(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned l = (unsigned)(b & 0xffffffffll), h = (unsigned)(b >> 32);
    const auto lo = O == 32 ? __builtin_amdgcn_permlane32_swap(l, l, false, false) : __builtin_amdgcn_permlane16_swap(l, l, false, false);
    const auto hi = O == 32 ? __builtin_amdgcn_permlane32_swap(h, h, false, false) : __builtin_amdgcn_permlane16_swap(h, h, false, false);
    return __longlong_as_double(((long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = reduce_swap_sum<32>(v);
    v = reduce_swap_sum<16>(v);
    v += reduce_dpp_f64<0x128>(v);   // row_ror:8
    v += reduce_dpp_f64<0x124>(v);   // row_ror:4
    v += reduce_dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += reduce_dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    return v;
}
// sum over the workgroup; result valid in every thread.  red must hold blockDim/64 + 1 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < nw; ++i) s += red[i];
        red[nw] = s;
    }
    __syncthreads();
    return red[nw];
}
// every workgroup re-reduces the producer's per-workgroup partials in the same fixed order: deterministic, no atomics
__device__ __forceinline__ double sum_partials(const double* partial, int n, double* red) {
    double v = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += partial[i];
    return block_sum(v, red);
}

// Sum over aligned groups of T lanes with DPP row operations (VALU data path; the ds_bpermute the compiler emits for
// __shfl_xor goes through the LDS crossbar, shared by the 4 SIMDs of the CU: 40 of them per 32-row tile were on the
// critical path of the team kernels).  Every lane of the group ends up with the group's total.  T <= 16 stays inside a
// DPP row (16 lanes); wider groups finish with __shfl_xor.
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int T> __device__ __forceinline__ double team_sum(double v) {
    if constexpr (T >= 32) {
#pragma unroll
        for (int o = T / 2; o >= 16; o >>= 1) v += __shfl_xor(v, o, T);
    }
    if constexpr (T >= 16) v += dpp_mov_f64<0x140>(v);   // row_mirror:       lane i <-> 15 - i
    if constexpr (T >= 8) v += dpp_mov_f64<0x141>(v);    // row_half_mirror:  lane i <-> 7 - i
    if constexpr (T >= 4) v += dpp_mov_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    if constexpr (T >= 2) v += dpp_mov_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    return v;
}

}  // namespace fdapde_hip
#endif
