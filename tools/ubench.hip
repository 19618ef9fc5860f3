// Micro-benchmarks that bound the pair passes on gfx950 (run on the GPU box):
//   valu     : v_xor_b32 + v_bcnt_u32_b32 chain, SGPR operands, no memory
//   ds_add   : conflict-free ds_add_u32 (lane-private column), no VALU beyond the address
//   hist_mix : 4 VALU + address + ds_add per pair, the k_hist inner loop without loads
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;

__global__ __launch_bounds__(256) void valu(u32* out, const u32* __restrict__ k, int iters) {
    u32 a = threadIdx.x * 2654435761u, b = blockIdx.x * 40503u + 1, acc = 0;
    const u32 s0 = k[0], s1 = k[1], s2 = k[2], s3 = k[3];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc += __builtin_popcount(a ^ (s0 + j)) + __builtin_popcount(b ^ (s1 + j));
            acc += __builtin_popcount(a ^ (s2 + j)) + __builtin_popcount(b ^ (s3 + j));
        }
        a += acc; 
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void ds_add(u32* out, int iters, int nb) {
    extern __shared__ u32 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32* h = lds + wave * nb * 64;
    for (int d = 0; d < nb; ++d) h[d * 64 + lane] = 0;
    u32 d = lane % nb;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { atomicAdd(&h[d * 64 + lane], 1u); d = d + 1 < (u32)nb ? d + 1 : 0; }
    }
    u32 s = 0;
    for (int dd = 0; dd < nb; ++dd) s += h[dd * 64 + lane];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void hist_mix(u32* out, const u32* __restrict__ k, int iters, int nb) {
    extern __shared__ u32 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32* h = lds + wave * nb * 64;
    for (int d = 0; d < nb; ++d) h[d * 64 + lane] = 0;
    u32 a = threadIdx.x * 2654435761u, b = blockIdx.x * 40503u + 1;
    const u32* p = k;
    for (int i = 0; i < iters; ++i) {
        u32 c[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) c[j] = p[(i & 63) * 32 + j];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u32 d = __builtin_popcount(a ^ c[2 * j]) + __builtin_popcount(b ^ c[2 * j + 1]);
            atomicAdd(&h[d * 64 + lane], 1u);
        }
    }
    u32 s = 0;
    for (int dd = 0; dd < nb; ++dd) s += h[dd * 64 + lane] * dd;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    u32 *out, *k;
    CK(hipMalloc(&out, 4096 * 256 * 4));
    CK(hipMalloc(&k, 64 * 32 * 4));
    u32 hk[64 * 32];
    for (int i = 0; i < 64 * 32; ++i) hk[i] = i * 2654435761u;
    CK(hipMemcpy(k, hk, sizeof hk, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    const int blocks = 2048;
    for (int rep = 0; rep < 2; ++rep) {
        int iters = 4000;
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(valu, dim3(blocks), dim3(256), 0, 0, out, k, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        double ops = (double)blocks * 256 * iters * 8 * 8;   // 4 xor + 4 bcnt per j
        if (rep) printf("valu     : %.3f ms  %.2f T xor|bcnt lane-ops/s (peak 78.6)\n", ms, ops / ms / 1e9);
    }
    for (int nb : {33, 65, 129}) {
        size_t lds = 4 * nb * 64 * 4;
        CK(hipFuncSetAttribute((const void*)ds_add, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void*)hist_mix, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int rep = 0; rep < 2; ++rep) {
            int iters = 2000;
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(ds_add, dim3(blocks), dim3(256), lds, 0, out, iters, nb);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            double ops = (double)blocks * 256 * iters * 16;
            if (rep) printf("ds_add   nb=%3d: %.3f ms  %.2f T lane-atomics/s  (%.2f lanes/clk/CU @2.4GHz)\n", nb, ms, ops / ms / 1e9, ops / ms / 1e9 * 1e12 / 256 / 2.4e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            int iters = 2000;
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(hist_mix, dim3(blocks), dim3(256), lds, 0, out, k, iters, nb);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            double pairs = (double)blocks * 256 * iters * 16;
            if (rep) printf("hist_mix nb=%3d: %.3f ms  %.2f T pairs/s  (1e10 pairs -> %.3f ms)\n", nb, ms, pairs / ms / 1e9, 1e10 / (pairs / ms));
        }
    }
    CK(hipGetLastError());
    return 0;
}
