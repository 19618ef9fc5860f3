// What the two-kernel design of BASELINE.json's north star would cost at C2 (Q=10k, N=1M, b=64):
// kernel A writes the Q x N uint8 Hamming-distance matrix (10 GB), kernel B reads it back once
// (the minimum any top-R kernel must do).  Both are pure HBM streams -- this measures the floor
// of that design on this GPU, to compare with the fused path's whole step.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_materialize tools/ubench_materialize.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;

// lanes <-> 16 consecutive database rows each (one 16-byte store), query uniform per block row
__global__ __launch_bounds__(256) void dist_matrix(const u64* __restrict__ q, const u64* __restrict__ db,
                                                   uint4* __restrict__ out, int Q, long N) {
    const long chunk = (long)blockIdx.x * 256 + threadIdx.x;          // 16 rows per thread
    const long n0 = chunk * 16;
    if (n0 >= N) return;
    u64 rows[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) rows[j] = db[n0 + j];
    for (int qi = blockIdx.y; qi < Q; qi += gridDim.y) {
        const u64 qc = q[qi];
        u32 w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc |= (u32)__popcll(qc ^ rows[k * 4 + j]) << (8 * j);
            w[k] = acc;
        }
        out[((long)qi * N + n0) / 16] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__global__ __launch_bounds__(256) void read_matrix(const uint4* __restrict__ in, u32* __restrict__ out, long n16) {
    u32 acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;      // keep the loads alive
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const int Q = 10000; const long N = 1000000;
    u64 *q, *db; uint4* mat; u32* sink;
    CK(hipMalloc(&q, Q * 8)); CK(hipMalloc(&db, N * 8)); CK(hipMalloc(&mat, (size_t)Q * N)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(q, 0x5a, Q * 8)); CK(hipMemset(db, 0x33, N * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(dist_matrix, dim3((N / 16 + 255) / 256, 64), dim3(256), 0, 0, q, db, mat, Q, N);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("write Q x N u8 distance matrix (%.1f GB): %.3f ms  %.2f TB/s\n", (double)Q * N / 1e9, ms, (double)Q * N / ms / 1e9);
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(read_matrix, dim3(256 * 8), dim3(256), 0, 0, mat, sink, (long)Q * N / 16);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("read it back once                        : %.3f ms  %.2f TB/s\n", ms, (double)Q * N / ms / 1e9);
    }
    CK(hipGetLastError());
    return 0;
}
