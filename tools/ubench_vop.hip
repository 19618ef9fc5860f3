// VALU issue rate on gfx950 by encoding / operand kind and by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
#define I4(INS, OPK) asm volatile(INS " %0, " OPK ", %0\n" INS " %1, " OPK ", %1\n" INS " %2, " OPK ", %2\n" INS " %3, " OPK ", %3\n" \
                                  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(kv), "s"(ks));
template <int OP>
__global__ __launch_bounds__(256) void k_rate(u32* out, const u32* __restrict__ src, int iters) {
    u32 x0 = threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
    const u32 kv = src[threadIdx.x & 63], ks = src[blockIdx.x & 7];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) { I4("v_xor_b32", "%4") }
            if (OP == 1) { I4("v_xor_b32", "%5") }
            if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %4, 31\nv_alignbit_b32 %1, %1, %4, 31\nv_alignbit_b32 %2, %2, %4, 31\nv_alignbit_b32 %3, %3, %4, 31\n"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(kv));
            if (OP == 3) asm volatile("v_bcnt_u32_b32 %0, %4, %0\nv_bcnt_u32_b32 %1, %4, %1\nv_bcnt_u32_b32 %2, %4, %2\nv_bcnt_u32_b32 %3, %4, %3\n"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(kv));
            if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %4, %0\nv_bcnt_u32_b32 %1, %4, %1\nv_bcnt_u32_b32 %2, %4, %2\nv_bcnt_u32_b32 %3, %4, %3\n"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(ks));
            if (OP == 5) asm volatile("v_lshl_or_b32 %0, %0, 1, %4\nv_lshl_or_b32 %1, %1, 1, %4\nv_lshl_or_b32 %2, %2, 1, %4\nv_lshl_or_b32 %3, %3, 1, %4\n"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(kv));
            if (OP == 6) { I4("v_add_u32", "%4") }
            if (OP == 7) asm volatile("v_xor_b32 %0, %0, %0\n" : "+v"(x0));   // dependent chain of 1
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class K> int run(const char* name, K kern, u32* out, u32* src, int wps, int per_iter) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, out, src, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("%-30s waves/SIMD=%d  %.3f ms  %.2f ns/instr/SIMD (%.2f cycles @2.2GHz)\n", name, wps, ms,
           ms * 1e6 / ((double)wps * iters * per_iter), ms * 1e6 / ((double)wps * iters * per_iter) * 2.2);
    return 0;
}
int main() {
    u32 *out, *src; CK(hipMalloc(&out, 2048 * 256 * 4)); CK(hipMalloc(&src, 4096)); CK(hipMemset(src, 0x5a, 4096));
    for (int wps : {1, 2, 4, 8}) {
        run("v_xor_b32 v,v,v", k_rate<0>, out, src, wps, 32);
        run("v_xor_b32 v,s,v", k_rate<1>, out, src, wps, 32);
        run("v_alignbit_b32 v,v,v,31", k_rate<2>, out, src, wps, 32);
        run("v_bcnt_u32_b32 v,v,v", k_rate<3>, out, src, wps, 32);
        run("v_bcnt_u32_b32 v,s,v", k_rate<4>, out, src, wps, 32);
        run("v_lshl_or_b32 v,v,1,v", k_rate<5>, out, src, wps, 32);
        run("v_add_u32 v,v,v", k_rate<6>, out, src, wps, 32);
        run("v_xor dependent chain", k_rate<7>, out, src, wps, 8);
    }
    return 0;
}
