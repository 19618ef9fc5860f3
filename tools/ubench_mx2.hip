// Can the matrix pipe and the vector ALU of a gfx950 SIMD work at the same time?
// Blocks of 8 waves = 2 per SIMD.  role 0: all waves run an MFMA stream; role 1: all run a VALU stream;
// role 2: of each SIMD's two waves one runs MFMAs and the other VALU ops (same per-wave work as in 0 / 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int VOP>
__global__ __launch_bounds__(512) void k_roles(u32* out, const int* __restrict__ src, int iters, int role) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // waves w and w+4 share a SIMD (round-robin placement)
    const bool do_mfma = role == 0 || (role == 2 && wave < 4);
    const bool do_valu = role == 1 || (role == 2 && wave >= 4);
    i32x4 a0 = *(const i32x4*)(src + lane * 4), b0 = *(const i32x4*)(src + 256 + lane * 4);
    i32x16 c0, c1, c2, c3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = src[r]; c1[r] = src[r + 1]; c2[r] = src[r + 2]; c3[r] = src[r + 3]; }
    u32 x0 = lane, x1 = lane * 3, x2 = lane * 5, x3 = lane * 7, k = src[5];
    if (do_mfma) {
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c3, 0, 0, 0);
        }
    }
    if (do_valu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (VOP == 0) asm volatile("v_xor_b32 %0, %4, %0\nv_xor_b32 %1, %4, %1\nv_xor_b32 %2, %4, %2\nv_xor_b32 %3, %4, %3\n"
                                           : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(k));
                else asm volatile("v_alignbit_b32 %0, %0, %4, 31\nv_alignbit_b32 %1, %1, %4, 31\nv_alignbit_b32 %2, %2, %4, 31\nv_alignbit_b32 %3, %3, %4, 31\n"
                                  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(k));
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ (u32)c0[0] ^ (u32)c1[1] ^ (u32)c2[2] ^ (u32)c3[3];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class K> int run(const char* name, K kern, u32* out, int* src, int role) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256, iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, src, iters, role);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("%-46s %.3f ms\n", name, ms);
    return 0;
}
int main() {
    u32* out; int* src; CK(hipMalloc(&out, 512 * 512 * 4)); CK(hipMalloc(&src, 65536)); CK(hipMemset(src, 0x01, 65536));
    run("xor: all 8 waves MFMA (4 per iter)", k_roles<0>, out, src, 0);
    run("xor: all 8 waves VALU (32 xor per iter)", k_roles<0>, out, src, 1);
    run("xor: 4 waves MFMA + 4 waves VALU", k_roles<0>, out, src, 2);
    run("alignbit: all 8 waves VALU (32 per iter)", k_roles<1>, out, src, 1);
    run("alignbit: 4 waves MFMA + 4 waves VALU", k_roles<1>, out, src, 2);
    return 0;
}
