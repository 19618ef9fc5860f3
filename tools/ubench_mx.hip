// Rate probes for the k_select_mx inner step on gfx950: NW=2 v_mfma_i32_32x32x32_i8 + 16 v_alignbit per
// (row tile, query tile) step.  Reports ns per step per SIMD for different mixes and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only   1: alignbit only (one dependent chain of 16)   2: both, pipelined (MFMAs of step i+1 before chain i)
// MODE 3: both, two independent chains of 8   4: alignbit only, two chains   5: both, not pipelined
template <int MODE, int QT>
__global__ __launch_bounds__(256) void k_step(u32* out, const int* __restrict__ src, int iters) {
    const int lane = threadIdx.x & 63;
    i32x4 a0 = *(const i32x4*)(src + lane * 4), a1 = *(const i32x4*)(src + 256 + lane * 4);
    i32x4 b[QT][2];
    i32x16 bias[QT];
    u32 m[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        b[t][0] = *(const i32x4*)(src + 512 + t * 512 + lane * 4);
        b[t][1] = *(const i32x4*)(src + 768 + t * 512 + lane * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[t][r] = src[t] - 40;
        m[t] = 0;
    }
    auto issue = [&](int t) -> i32x16 {
        i32x16 acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b[t][0], bias[t], 0, 0, 0);
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b[t][1], acc, 0, 0, 0);
    };
    i32x16 accn = issue(0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            i32x16 acc = accn;
            if (MODE == 0) {
                accn = issue((t + 1) % QT);
                asm volatile("" : "+v"(accn));
                if (i == iters - 1) m[t] ^= (u32)acc[3];
            } else if (MODE == 1 || MODE == 4) {
                u32 mm = m[t], m2 = m[t] + 1;
                if (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mm = __builtin_amdgcn_alignbit(mm, (u32)acc[r], 31);
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { mm = __builtin_amdgcn_alignbit(mm, (u32)acc[r], 31); m2 = __builtin_amdgcn_alignbit(m2, (u32)acc[r + 8], 31); }
                    mm = (mm << 8) | (m2 & 255);
                }
                asm volatile("" : "+v"(mm));
                m[t] = mm;
            } else {
                if (MODE != 5) accn = issue((t + 1) % QT);
                u32 mm = m[t], m2 = 0;
                if (MODE == 3) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { mm = __builtin_amdgcn_alignbit(mm, (u32)acc[r], 31); m2 = __builtin_amdgcn_alignbit(m2, (u32)acc[r + 8], 31); }
                    mm = (mm << 8) | (m2 & 255);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mm = __builtin_amdgcn_alignbit(mm, (u32)acc[r], 31);
                }
                asm volatile("" : "+v"(mm));
                m[t] = mm;
                if (MODE == 5) accn = issue((t + 1) % QT);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    u32 x = (u32)accn[0];
#pragma unroll
    for (int t = 0; t < QT; ++t) x ^= m[t];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

typedef int i32x4v __attribute__((ext_vector_type(4)));
// MODE 0: two INDEPENDENT 32x32x32 MFMAs per step + 16 alignbits on the previous step's first result
// MODE 1: four independent 16x16x64 MFMAs per step + 16 alignbits on the previous step's results
// MODE 2: as 1, MFMA only      MODE 3: as 0, MFMA only
template <int MODE, int QT, bool ZC = false>
__global__ __launch_bounds__(256) void k_indep(u32* out, const int* __restrict__ src, int iters) {
    const int lane = threadIdx.x & 63;
    i32x4 a0 = *(const i32x4*)(src + lane * 4), a1 = *(const i32x4*)(src + 256 + lane * 4);
    i32x4 b[QT][2];
    i32x16 bias[QT];
    u32 m[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        b[t][0] = *(const i32x4*)(src + 512 + t * 512 + lane * 4);
        b[t][1] = *(const i32x4*)(src + 768 + t * 512 + lane * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[t][r] = src[t] - 40;
        m[t] = 0;
    }
    i32x16 p0 = bias[0], p1 = bias[0];
    i32x4v q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (MODE == 0 || MODE == 3) {
                const i32x16 c0 = p0, c1 = p1;
                if (ZC) {
                    const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    p0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b[t][0], z, 0, 0, 0);
                    p1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b[t][1], z, 0, 0, 0);
                } else {
                p0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b[t][0], bias[t], 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b[t][1], bias[t], 0, 0, 0);
                }
                u32 mm = m[t];
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { mm = __builtin_amdgcn_alignbit(mm, (u32)c0[r], 31); mm = __builtin_amdgcn_alignbit(mm, (u32)c1[r + 8], 31); }
                } else mm ^= (u32)c0[1] ^ (u32)c1[2];
                asm volatile("" : "+v"(mm));
                m[t] = mm;
            } else {
                const i32x4v c0 = q0, c1 = q1, c2 = q2, c3 = q3;
                const i32x4v z = {bias[t][0], bias[t][1], bias[t][2], bias[t][3]};
                q0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b[t][0], z, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b[t][0], z, 0, 0, 0);
                q2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b[t][1], z, 0, 0, 0);
                q3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b[t][1], z, 0, 0, 0);
                u32 mm = m[t];
                if (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) mm = __builtin_amdgcn_alignbit(mm, (u32)c0[r], 31);
#pragma unroll
                    for (int r = 0; r < 4; ++r) mm = __builtin_amdgcn_alignbit(mm, (u32)c1[r], 31);
#pragma unroll
                    for (int r = 0; r < 4; ++r) mm = __builtin_amdgcn_alignbit(mm, (u32)c2[r], 31);
#pragma unroll
                    for (int r = 0; r < 4; ++r) mm = __builtin_amdgcn_alignbit(mm, (u32)c3[r], 31);
                } else mm ^= (u32)c0[1] ^ (u32)c1[2] ^ (u32)c2[0] ^ (u32)c3[3];
                asm volatile("" : "+v"(mm));
                m[t] = mm;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    u32 x = (u32)p0[0] ^ (u32)p1[1] ^ (u32)q0[0] ^ (u32)q1[0] ^ (u32)q2[0] ^ (u32)q3[0];
#pragma unroll
    for (int t = 0; t < QT; ++t) x ^= m[t];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class K> int run(const char* name, K kern, u32* out, int* src, int wps, int qt) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * wps, iters = 2000;   // wps blocks x 4 waves per CU = wps waves per SIMD
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, src, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    const double steps_per_simd = (double)wps * iters * qt;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.1f ns/step/SIMD = %.0f cycles @2.2GHz\n", name, wps, ms, ms * 1e6 / steps_per_simd,
           ms * 1e6 / steps_per_simd * 2.2);
    return 0;
}
int main() {
    u32* out; int* src; CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&src, 65536)); CK(hipMemset(src, 0x01, 65536));
    for (int wps = 1; wps <= 2; ++wps) {
        run("mfma only (2 per step)", k_step<0, 4>, out, src, wps, 4);
        run("alignbit only, 1 chain", k_step<1, 4>, out, src, wps, 4);
        run("alignbit only, 2 chains", k_step<4, 4>, out, src, wps, 4);
        run("both pipelined, 1 chain", k_step<2, 4>, out, src, wps, 4);
        run("both pipelined, 2 chains", k_step<3, 4>, out, src, wps, 4);
        run("both, not pipelined", k_step<5, 4>, out, src, wps, 4);
    }
    for (int wps = 1; wps <= 4; ++wps) {
        run("indep 2x 32x32x32 + 16 alignbit", k_indep<0, 4>, out, src, wps, 4);
        run("indep 2x 32x32x32 only", k_indep<3, 4>, out, src, wps, 4);
        run("indep 2x 32x32x32 C=0 + 16 alignbit", k_indep<0, 4, true>, out, src, wps, 4);
        run("indep 2x 32x32x32 C=0 only", k_indep<3, 4, true>, out, src, wps, 4);
        run("indep 4x 16x16x64 + 16 alignbit", k_indep<1, 4>, out, src, wps, 4);
        run("indep 4x 16x16x64 only", k_indep<2, 4>, out, src, wps, 4);
    }
    for (int wps = 3; wps <= 4; ++wps) {
        run("both pipelined, 1 chain QT=2", k_step<2, 2>, out, src, wps, 2);
        run("both pipelined, 2 chains QT=2", k_step<3, 2>, out, src, wps, 2);
    }
    return 0;
}
