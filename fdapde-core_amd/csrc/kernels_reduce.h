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
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
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
