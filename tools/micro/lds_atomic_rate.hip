// LDS atomic throughput on gfx950: one 512-thread workgroup per CU, every lane adds to pseudo-random slots of a 64 KB table.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip ; prints ns per wavefront instruction and lanes per clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned* idx, int n, long long* ticks, double* sink) {
    __shared__ unsigned long long tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) tab[i] = 0;
    unsigned a[16];
    for (int j = 0; j < 16; ++j) a[j] = idx[(blockIdx.x * 16 + j) * 512 + threadIdx.x];
    __syncthreads();
    const long long t0 = wall_clock64();
    double acc = 0;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned s = (a[j] + it * 37) & 8191u;
            if constexpr (MODE == 0) __hip_atomic_fetch_add(&tab[s], (unsigned long long)(it + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(tab) + s, (unsigned)(it + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<double*>(tab) + s, (double)(it + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 3) acc += reinterpret_cast<double*>(tab)[s];
            else if constexpr (MODE == 4) reinterpret_cast<double*>(tab)[s] = (double)(it + j);
            else if constexpr (MODE == 5) __hip_atomic_fetch_add(reinterpret_cast<float*>(tab) + s, (float)(it + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 6) {   // two 32-bit adds per 64-bit value (low, high): carry-free accumulation needs a different encoding
                __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(tab) + 2 * s, (unsigned)(it + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(tab) + 2 * s + 1, (unsigned)(it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (acc == 12345.678) sink[0] = acc + (double)tab[threadIdx.x];
    if (n < 0) sink[1] = (double)tab[threadIdx.x];
}
int main() {
    const int G = 256, n = 200;
    std::vector<unsigned> h((size_t)G * 16 * 512);
    unsigned x = 12345;
    for (auto& v : h) x = x * 1664525u + 1013904223u, v = (x >> 8) & 8191u;
    unsigned* d;
    long long* t;
    double* sink;
    hipMalloc(&d, h.size() * 4), hipMalloc(&t, G * 8), hipMalloc(&sink, 64);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[] = {"ds_add_u64", "ds_add_u32", "ds_add_f64", "ds_read_b64", "ds_write_b64", "ds_add_f32", "2 x ds_add_u32"};
    for (int mode = 0; mode < 7; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(G), dim3(512), 0, 0, d, n, t, sink); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(G), dim3(512), 0, 0, d, n, t, sink); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(G), dim3(512), 0, 0, d, n, t, sink); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(G), dim3(512), 0, 0, d, n, t, sink); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(G), dim3(512), 0, 0, d, n, t, sink); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(G), dim3(512), 0, 0, d, n, t, sink); break;
                default: hipLaunchKernelGGL(k<6>, dim3(G), dim3(512), 0, 0, d, n, t, sink); break;
            }
            hipDeviceSynchronize();
        }
        std::vector<long long> ht(G);
        hipMemcpy(ht.data(), t, G * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : ht) mean += (double)v;
        mean /= G;
        const double ns = mean * 10.0, ops = (double)n * 16 * 512;   // lane operations per workgroup
        std::printf("%-14s: %8.1f us per workgroup, %.3f ns per lane-op, %.2f lane-ops per clock at 2.4 GHz\n", names[mode], ns * 1e-3, ns / ops, ops / (ns * 2.4));
    }
    return 0;
}
