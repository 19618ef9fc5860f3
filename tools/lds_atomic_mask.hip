// Does an LDS atomic add cost less when most lanes are masked off?  Each thread adds to its own counter column ([row][thread], the
// histogram kernels' layout: the bank is the lane) at a pseudo-random row; `keep` of 256 lanes take part.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lam tools/lds_atomic_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned keep) {
    __shared__ unsigned cnt[64 * 256];
    for (int i = threadIdx.x; i < 64 * 256; i += 256) cnt[i] = 0;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
    const bool on = ((threadIdx.x * 37u) & 255u) < keep;       // which lanes take part: spread over the wavefront
    unsigned* ad[16];                                          // sixteen rows of the thread's column, fixed: no address arithmetic in the loop
    for (int u = 0; u < 16; ++u) { x = x * 1664525u + 1013904223u; ad[u] = &cnt[((x >> 20) & 63u) * 256 + threadIdx.x]; }
    if (on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) atomicAdd(ad[u], 1u);
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();
    unsigned s = 0;
    for (int i = threadIdx.x; i < 64 * 256; i += 256) s += cnt[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + x;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (unsigned keep : {256u, 128u, 64u, 32u, 8u, 0u}) {
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 2000, keep);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 2000, keep);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("lanes taking part %3u of 256: %.3f ms\n", keep, ms);
    }
    return 0;
}
