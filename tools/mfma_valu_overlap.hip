// Does a SIMD run vector-ALU instructions while one of its MFMAs is in flight?  Three loops per wavefront, N wavefronts per SIMD:
//   A: independent MFMAs only;  B: independent v_alignbit only;  C: both, interleaved 1 MFMA : V VALU (independent of each other).
// time(C) ~ max(A, B): they overlap;  ~ A + B: the vector pipe issues one or the other.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/mfma_valu_overlap.hip ; run: /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[2];
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 1.f; }
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE != 1) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 1], 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int v = 0; v < V; ++v) x[(u + v) & 7] = __builtin_amdgcn_alignbit(x[(u + v) & 7], x[(u + v + 3) & 7], 7);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    unsigned y = 0;
    for (int i = 0; i < 8; ++i) y ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)y;
}
template <int MODE, int V> float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, V>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, V>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {           // wavefronts per SIMD: blocks of 4 wavefronts, one per SIMD
        const int blocks = 256 * wps;
        printf("wavefronts/SIMD %d:  MFMA only %.3f ms | VALU only V=8 %.3f, V=4 %.3f | both V=8 %.3f, V=4 %.3f\n", wps,
               run<0, 8>(d, blocks, iters), run<1, 8>(d, blocks, iters), run<1, 4>(d, blocks, iters), run<2, 8>(d, blocks, iters), run<2, 4>(d, blocks, iters));
    }
    return 0;
}
