// One-off check on the GPU: for integers 1 <= cnt <= k, is  q' = fma(fma(-q, k, cnt), y, q)  with  y = RN(1 / k),  q = RN(cnt * y)
// the correctly rounded cnt / k (what k_ap needs bit for bit)?  Exhaustive for k <= KMAX, random pairs beyond.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ap_div_check tools/ap_div_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__device__ __forceinline__ double fast_div(double a, double b, double y) {
    const double q = a * y;
    const double r = __builtin_fma(-q, b, a);
    return __builtin_fma(r, y, q);
}
__global__ void k_exhaustive(int kmax, u64* bad, u64* tested) {
    const int k = blockIdx.x + 1;
    if (k > kmax) return;
    const double b = (double)k, y = 1.0 / b;
    u64 nb = 0, nt = 0;
    for (int c = threadIdx.x + 1; c <= k; c += blockDim.x) {
        const double a = (double)c;
        const double ref = a / b, got = fast_div(a, b, y);
        nb += __double_as_longlong(ref) != __double_as_longlong(got);
        ++nt;
    }
    if (nb) atomicAdd(bad, nb);
    atomicAdd(tested, nt);
}
__device__ __forceinline__ u64 mix(u64 x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void k_random(u64 seed, int per, unsigned kmaxbits, u64* bad, u64* tested) {
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 nb = 0;
    for (int i = 0; i < per; ++i) {
        const u64 h = mix(seed + tid * per + i), h2 = mix(h);
        const u64 k = (h >> (64 - kmaxbits)) + 1;
        const u64 c = h2 % k + 1;
        const double a = (double)c, b = (double)k, y = 1.0 / b;
        nb += __double_as_longlong(a / b) != __double_as_longlong(fast_div(a, b, y));
    }
    if (nb) atomicAdd(bad, nb);
    atomicAdd(tested, (u64)per);
}
int main(int argc, char** argv) {
    const int kmax = argc > 1 ? atoi(argv[1]) : 131072;
    u64 *d, h[2] = {0, 0};
    hipMalloc(&d, 16); hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k_exhaustive, dim3(kmax), dim3(256), 0, 0, kmax, d, d + 1);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("exhaustive 1 <= cnt <= k <= %d: %llu pairs, %llu differ from the division\n", kmax, h[1], h[0]);
    for (unsigned bits : {20u, 24u, 28u, 31u}) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(k_random, dim3(65536), dim3(256), 0, 0, 12345ull + bits, 256, bits, d, d + 1);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("random k < 2^%u: %llu pairs, %llu differ\n", bits, h[1], h[0]);
    }
    return h[0] != 0;
}
