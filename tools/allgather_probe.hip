// What the dot all-gather of the single-launch solvers costs BY ITSELF, as a function of the number of workgroups taking part and of what else the
// memory system is doing: G resident workgroups of 512 threads run `rounds` rounds of
//     [optional: every wavefront streams `stream_kb` KB of a private buffer (the operator phase's loads)]
//     three lanes publish the workgroup's record (3 x 16-byte sc1 stores into one 64-byte slot of the round's parity)
//     thread t polls record t (3 x 16-byte sc1 loads, re-read until all six tags carry the round) -- k_cg_persist's protocol
//     one barrier
// and workgroup 0 reports the average round.  Nothing is computed: the round time is publish + visibility + poll + barrier (+ stream).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/allgather_probe tools/allgather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u64 v2u64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_sc1(u64* p, u32x4 q) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory"); }
__device__ __forceinline__ void ld6_sc1(const u64* p, v2u64& a, v2u64& b, v2u64& c) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p) : "memory");
}
__device__ __forceinline__ bool ok(v2u64 v, unsigned r) { return (unsigned)(v.x >> 32) == r && (unsigned)(v.y >> 32) == r; }
// mode 0: every workgroup polls all G records (the product's flat sweep); mode 1: only wavefront 0 polls, 4+ records per lane (gather_waves = 1)
// mode 2: no gather at all (stream + barrier): the base line of the rows with a stream;  stride: granules (8 bytes) between two records
__global__ __launch_bounds__(512) void allgather(u64* board, int G, int rounds, int mode, int sleep, int stride, const double* stream, int stream_kb, double* sink,
                                                long long* ticks, int* fail) {
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int bad;
    if (tid == 0) bad = 0;
    __syncthreads();
    double acc = 0;
    long long t0 = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (r == 3 && g == 0 && tid == 0) t0 = wall_clock64();   // (two rounds to get everybody resident)
        if (stream_kb > 0) {   // 16 bytes per lane per load: stream_kb KB per wavefront and round, a different window every round
            const double2* sp = reinterpret_cast<const double2*>(stream) + ((size_t)(g * 8 + wave) * rounds + r) % 4096 * (size_t)stream_kb * 64;
            for (int k = 0; k < stream_kb; ++k) {
                const double2 v = sp[k * 64 + lane];
                acc += v.x + v.y;
            }
        }
        u64* slot = board + (size_t)(r & 1) * G * stride;
        if (mode != 2 && tid < 3) {
            u32x4 q = {(unsigned)g, (unsigned)r, (unsigned)tid, (unsigned)r};
            st_sc1(slot + (size_t)g * stride + 2 * tid, q);
        }
        const int per_lane = mode == 1 ? (G + 63) / 64 : 1;
        if (mode == 2) {
        } else if (mode == 1 ? wave == 0 : wave * 64 < G) {
            for (int s = 0; s < per_lane; ++s) {
                const int w = mode == 1 ? s * 64 + lane : tid;
                const u64* gp = slot + (size_t)(w < G ? w : 0) * stride;
                v2u64 a = {0, 0}, b = {0, 0}, c = {0, 0};
                bool done = w >= G;
                for (unsigned spins = 0;; ++spins) {
                    if (!done) {
                        ld6_sc1(gp, a, b, c);
                        done = ok(a, (unsigned)r) && ok(b, (unsigned)r) && ok(c, (unsigned)r);
                    }
                    if (__all(done)) break;
                    if (spins > 4000000u) {
                        bad = 1;
                        break;
                    }
                    if (sleep == 1) __builtin_amdgcn_s_sleep(1);
                    else if (sleep == 2) __builtin_amdgcn_s_sleep(2);
                    else if (sleep >= 3) __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        __syncthreads();
        if (bad) break;
    }
    if (g == 0 && tid == 0) *ticks = wall_clock64() - t0;
    if (bad && tid == 0) *fail = 1;
    if (acc == 12345.678) sink[g] = acc;
}
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    u64* board;
    long long* ticks;
    int* fail;
    double *stream, *sink;
    const size_t stream_doubles = (size_t)4096 * 64 * 64 * 2 + (1 << 20);   // up to 64 KB per wavefront and window
    hipMalloc(&board, 2 * 256 * 32 * sizeof(u64) + 4096);
    hipMalloc(&ticks, 8);
    hipMalloc(&fail, 4);
    hipMalloc(&stream, stream_doubles * 8);
    hipMalloc(&sink, 256 * 8);
    hipMemset(stream, 0, stream_doubles * 8);
    printf("rounds %d; round time in us (workgroup 0's clock, 100 MHz), average over the rounds after the second\n", rounds);
    printf("%-6s %-8s %-6s %-8s %-10s %s\n", "G", "pollers", "sleep", "stride B", "stream KB", "us per round");
    const int Gs[] = {2, 8, 32, 64, 128, 256};
    for (int stream_kb : {0, 16, 64})
        for (int mode : {0, 1, 2})
            for (int stride : {6, 8, 16, 32})
                for (int sleep : {0, 1, 2})
                    for (int G : Gs) {
                        if (stream_kb > 0 && (G != 256 && G != 64)) continue;
                        if (sleep != 1 && (G != 256 || stride != 8)) continue;
                        if (mode == 2 && (stream_kb == 0 || stride != 8 || sleep != 1)) continue;
                        if (mode == 1 && stride != 8) continue;
                        if (stride != 8 && stream_kb == 64) continue;
                        hipMemset(board, 0, 2 * 256 * 32 * sizeof(u64));
                        hipMemset(fail, 0, 4);
                        hipLaunchKernelGGL(allgather, dim3(G), dim3(512), 0, 0, board, G, rounds, mode, sleep, stride, stream, stream_kb, sink, ticks, fail);
                        if (hipDeviceSynchronize() != hipSuccess) {
                            printf("launch failed\n");
                            return 1;
                        }
                        long long t;
                        int f;
                        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
                        hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
                        printf("%-6d %-8s %-6d %-8d %-10d %s%.3f\n", G, mode == 0 ? "4 waves" : (mode == 1 ? "1 wave" : "none"), sleep, stride * 8, stream_kb,
                               f ? "TIMEOUT " : "", t * 0.01 / (rounds - 2));
                        fflush(stdout);
                    }
    return 0;
}
