// Instruction-rate probes for gfx950: how many cycles does a wave64 VALU op of each kind cost?
// Each kernel runs a long unrolled stream of INDEPENDENT instructions (8 accumulators),
// 8 waves per SIMD resident, so the result is the issue rate, not the latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;

#define REP8(X) X X X X X X X X
#define BODY(OPS) \
    u32 a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    const u32 s = k[blockIdx.x & 7];                                                                                     \
    for (int i = 0; i < iters; ++i) { REP8(OPS) }                                                                       \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;

#define OP8(INS) asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n" \
                              INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
#define BCNT8 asm volatile("v_bcnt_u32_b32 %0, %0, %8\nv_bcnt_u32_b32 %1, %1, %8\nv_bcnt_u32_b32 %2, %2, %8\nv_bcnt_u32_b32 %3, %3, %8\n" \
                           "v_bcnt_u32_b32 %4, %4, %8\nv_bcnt_u32_b32 %5, %5, %8\nv_bcnt_u32_b32 %6, %6, %8\nv_bcnt_u32_b32 %7, %7, %8\n" \
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));
#define CMP8 asm volatile("v_cmp_le_u32 vcc, %0, %8\nv_cmp_le_u32 vcc, %1, %8\nv_cmp_le_u32 vcc, %2, %8\nv_cmp_le_u32 vcc, %3, %8\n" \
                          "v_cmp_le_u32 vcc, %4, %8\nv_cmp_le_u32 vcc, %5, %8\nv_cmp_le_u32 vcc, %6, %8\nv_cmp_le_u32 vcc, %7, %8\n" \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a1) : "vcc");
#define MIX8 asm volatile("v_xor_b32 %0, %8, %1\nv_xor_b32 %2, %8, %3\nv_bcnt_u32_b32 %0, %0, 0\nv_bcnt_u32_b32 %0, %2, %0\n" \
                          "v_xor_b32 %4, %8, %5\nv_xor_b32 %6, %8, %7\nv_bcnt_u32_b32 %4, %4, 0\nv_bcnt_u32_b32 %4, %6, %4\n" \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));

__global__ __launch_bounds__(256) void k_xor(u32* out, const u32* __restrict__ k, int iters) { BODY(OP8("v_xor_b32")) }
__global__ __launch_bounds__(256) void k_add(u32* out, const u32* __restrict__ k, int iters) { BODY(OP8("v_add_u32")) }
__global__ __launch_bounds__(256) void k_bcnt(u32* out, const u32* __restrict__ k, int iters) { BODY(BCNT8) }
__global__ __launch_bounds__(256) void k_cmp(u32* out, const u32* __restrict__ k, int iters) { BODY(CMP8) }
__global__ __launch_bounds__(256) void k_mix(u32* out, const u32* __restrict__ k, int iters) { BODY(MIX8) }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class K> int run(const char* name, K kern, u32* out, u32* k) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * 8, iters = 2000;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, k, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    double winstr = (double)blocks * 4 * iters * 64;          // wave-instructions
    double per_simd = winstr / 1024.0;                        // per SIMD
    printf("%-8s %.3f ms  %.2f T lane-ops/s   %.2f cycles/wave-instr/SIMD @2.4GHz (%.2f @2.1GHz)\n", name, ms,
           winstr * 64 / ms / 1e9, ms * 1e-3 * 2.4e9 / per_simd, ms * 1e-3 * 2.1e9 / per_simd);
    return 0;
}
int main() {
    u32 *out, *k; CK(hipMalloc(&out, 2048 * 256 * 4)); CK(hipMalloc(&k, 64)); CK(hipMemset(k, 0x5a, 64));
    run("v_xor", k_xor, out, k); run("v_add", k_add, out, k); run("v_bcnt", k_bcnt, out, k);
    run("v_cmp", k_cmp, out, k); run("mix2x2b", k_mix, out, k);
    return 0;
}
