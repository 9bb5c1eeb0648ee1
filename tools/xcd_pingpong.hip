// Latency of a hand-off between two workgroups through memory: across XCDs (16-byte store / load with sc1, what k_cg_persist's boards use) against
// both workgroups on ONE XCD with L2-scope accesses (plain store, sc0 load).  Blocks are dealt to the XCDs round-robin (block i -> XCD i % 8): the
// pair (0, 8) shares XCD 0, the pair (0, 1) does not; every block reports the XCC_ID it ran on.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/xcd_pingpong tools/xcd_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u64 v2u64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_sc1(u64* p, u32x4 q) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory"); }
__device__ __forceinline__ void st_l2(u64* p, u32x4 q) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(q) : "memory"); }
__device__ __forceinline__ v2u64 ld_sc1(const u64* p) { v2u64 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ v2u64 ld_sc0(const u64* p) { v2u64 v; asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
// blocks a and b play; everybody else leaves.  mode 0: sc1 / sc1, mode 1: plain store / sc0 load
__global__ void pingpong(u64* board, int a, int b, int rounds, int mode, long long* ticks, unsigned* xcc, int* fail) {
    const int me = blockIdx.x;
    if (threadIdx.x == 0) xcc[me] = xcc_id();
    if (me != a && me != b) return;
    if (threadIdx.x != 0) return;
    u64* mine = board + (me == a ? 0 : 16);   // (separate 128-byte lines)
    const u64* theirs = board + (me == a ? 16 : 0);
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (me == a) {
            u32x4 q = {(unsigned)r, (unsigned)r, (unsigned)r, (unsigned)r};
            if (mode == 0) st_sc1(mine, q); else st_l2(mine, q);
        }
        long long spins = 0;
        for (;;) {
            const v2u64 v = mode == 0 ? ld_sc1(theirs) : ld_sc0(theirs);
            if ((unsigned)(v.x >> 32) == (unsigned)r && (unsigned)(v.y >> 32) == (unsigned)r) break;
            if (++spins > 20000000) { *fail = 1; return; }
        }
        if (me == b) {
            u32x4 q = {(unsigned)r, (unsigned)r, (unsigned)r, (unsigned)r};
            if (mode == 0) st_sc1(mine, q); else st_l2(mine, q);
        }
    }
    if (me == a) *ticks = wall_clock64() - t0;
}
int main() {
    u64* board; long long* ticks; unsigned* xcc; int* fail;
    hipMalloc(&board, 4096); hipMalloc(&ticks, 8); hipMalloc(&xcc, 64 * 4); hipMalloc(&fail, 4);
    const int rounds = 2000;
    const int pairs[4][2] = {{0, 1}, {0, 8}, {0, 16}, {3, 11}};
    for (int mode = 0; mode < 2; ++mode)
        for (auto& pr : pairs) {
            hipMemset(board, 0, 4096); hipMemset(fail, 0, 4);
            hipLaunchKernelGGL(pingpong, dim3(32), dim3(64), 0, 0, board, pr[0], pr[1], rounds, mode, ticks, xcc, fail);
            hipDeviceSynchronize();
            long long t; unsigned x[64]; int f;
            hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
            std::printf("mode %s, blocks (%d, %d) on XCC (%u, %u): %s%.2f us per round trip (= two hand-offs)\n", mode == 0 ? "sc1 store / sc1 load" : "plain store / sc0 load",
                        pr[0], pr[1], x[pr[0]], x[pr[1]], f ? "TIMED OUT -- " : "", f ? 0.0 : (double)t / 100.0 / rounds);
        }
    return 0;
}
