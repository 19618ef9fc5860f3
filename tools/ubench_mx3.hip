// Three distances per f32 accumulator on the fp4 matrix instruction (gfx950): exactness check + issue-rate probe.
//
// A 32x32x64 fp4 MFMA with C = 2^23 + 2^s (T - pop(q) + 64)(1 + 2^7 + 2^14) accumulates THREE 16-row tiles into the
// same 16 registers, tile f at A-scale 2^(7 f + s): every accumulator is the integer
//     2^23 + 2^s * sum_f 2^(7 f) [T - dist(row_f) + 64]
// whose 7-bit fields lie in [0, 127] (codes of <= 64 bits, 0 <= T <= 63), so bit 6 + 7 f + s says dist(row_f) <= T.
// s = register-dependent (the MX scale is per lane = per A row): three registers share one mask word through
// v_and_or_b32 with K_s = bits {6, 13, 20} + s, two more shift-merges fill seven sub-positions per field:
// 20 vector ops harvest 48 rows per lane instead of 48 v_alignbit.
//
//   part 1: every lane checks its 3 mask words of one supertile against xor + popcount (random codes, random T)
//   part 2: time per supertile (2 query tiles: 6 MFMAs + 40 harvest ops), A fragments from LDS, against the
//           one-tile-per-accumulator form (6 MFMAs + 96 v_alignbit)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ u32 spread8(u32 y) {
    y = (y | (y << 12)) & 0x000F000Fu;
    y = (y | (y << 6)) & 0x03030303u;
    y = (y | (y << 3)) & 0x11111111u;
    return y;
}
__device__ __forceinline__ i32x4 expand_db(u32 x) {          // bit -> 0.0 / 1.0
    i32x4 o;
    o.x = spread8(x & 255u) << 1; o.y = spread8((x >> 8) & 255u) << 1; o.z = spread8((x >> 16) & 255u) << 1; o.w = spread8(x >> 24) << 1;
    return o;
}
__device__ __forceinline__ i32x4 expand_q(u32 x) {           // bit 1 -> +1.0 (0x2), bit 0 -> -1.0 (0xA)
    const i32x4 e = expand_db(x);
    i32x4 o;
    o.x = 0xAAAAAAAA ^ (e.x << 2); o.y = 0xAAAAAAAA ^ (e.y << 2); o.z = 0xAAAAAAAA ^ (e.z << 2); o.w = 0xAAAAAAAA ^ (e.w << 2);
    return o;
}

// register r (0..15) of tile f (0..2) -> row of the 48-row supertile; scale shift of the register
__host__ __device__ inline int m3_row(int f, int r) { return r < 7 ? 7 * f + r : r < 14 ? 21 + 7 * f + (r - 7) : 42 + 2 * f + (r - 14); }
__host__ __device__ inline int m3_shift(int r) { return r < 7 ? r % 3 : r < 14 ? (r - 7) % 3 : r - 14; }

__device__ __forceinline__ void harvest(const f32x16& acc, u32& A, u32& B, u32& C, const u32 K0 = 0x102040u, const u32 K1 = 0x204080u, const u32 K2 = 0x408100u) {
#define U(r) __float_as_uint(acc[r])
    u32 a0 = U(0) & K0; a0 = (U(1) & K1) | a0; a0 = (U(2) & K2) | a0;
    u32 a1 = U(3) & K0; a1 = (U(4) & K1) | a1; a1 = (U(5) & K2) | a1;
    u32 a2 = U(6) & K0;
    A = (a2 << 6) | ((a1 << 3) | a0);
    u32 b0 = U(7) & K0; b0 = (U(8) & K1) | b0; b0 = (U(9) & K2) | b0;
    u32 b1 = U(10) & K0; b1 = (U(11) & K1) | b1; b1 = (U(12) & K2) | b1;
    u32 b2 = U(13) & K0;
    B = (b2 << 6) | ((b1 << 3) | b0);
    C = (U(15) & K1) | (U(14) & K0);
#undef U
}

// codes: [2 halves][48 rows] u64 per block; queries: [32] u64; T: [32]
__global__ __launch_bounds__(64) void k_check(const u64* __restrict__ rows, const u64* __restrict__ qs, const int* __restrict__ Ts,
                                             u32* __restrict__ bad, u32* __restrict__ hits) {
    const int lane = threadIdx.x, h = lane >> 5, j = lane & 31;
    const u64* myrows = rows + (size_t)blockIdx.x * 96;
    const u64 q = qs[blockIdx.x * 32 + j];
    const int T = Ts[blockIdx.x * 32 + j];
    // A operand: lane = (A row i = lane & 31, k-half = lane >> 5).  A row i feeds lane-half (i >> 2) & 1, register (i & 3) + 4 (i >> 3).
    const int i = j, ah = (i >> 2) & 1, ar = (i & 3) + 4 * (i >> 3);
    const int s = m3_shift(ar);
    const int scale_a = (127 + s) | ((134 + s) << 8) | ((141 + s) << 16);
    const int scale_b = 0x7F7F7F7F;
    const i32x4 bq = expand_q((u32)(h ? q >> 32 : q));           // B: column j, k-half h
    f32x16 acc;
    const float base = (float)(T - __builtin_popcountll(q) + 64) * 16513.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 8388608.0f + base * (float)(1 << m3_shift(r));
#define STEP(f)                                                                                              \
    {                                                                                                        \
        const u64 x = myrows[ah * 48 + m3_row(f, ar)];                                                        \
        const i32x4 af = expand_db((u32)(h ? x >> 32 : x));                                                   \
        const i32x8 Av = {af.x, af.y, af.z, af.w, 0, 0, 0, 0};                                                \
        const i32x8 Bv = {bq.x, bq.y, bq.z, bq.w, 0, 0, 0, 0};                                                \
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Av, Bv, acc, 4, 4, f, scale_a, 0, scale_b);     \
    }
    STEP(0) STEP(1) STEP(2)
#undef STEP
    u32 A, B, C;
    harvest(acc, A, B, C);
    // flat 48-bit mask: bit P <-> row P (C's six rows sit at 42 + {0,1,7,8,14,15})
    u32 nbad = 0, nhit = 0;
    for (int row = 0; row < 48; ++row) {
        const int d = __builtin_popcountll(myrows[h * 48 + row] ^ q);
        const bool want = d <= T;
        bool got;
        if (row < 21) got = (A >> (6 + row)) & 1u;
        else if (row < 42) got = (B >> (6 + row - 21)) & 1u;
        else { const int c = row - 42; got = (C >> (6 + 7 * (c >> 1) + (c & 1))) & 1u; }
        nbad += want != got;
        nhit += want;
    }
    // stray bits outside the indicator positions
    if (A & ~(0x1FFFFFu << 6)) ++nbad;
    if (B & ~(0x1FFFFFu << 6)) ++nbad;
    if (C & ~((3u << 6) | (3u << 13) | (3u << 20))) ++nbad;
    atomicAdd(bad, nbad);
    atomicAdd(hits, nhit);
}

// issue-rate probe: 4 waves per block, A fragments from LDS (3 KiB per supertile), QT = 2
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rate(const int* __restrict__ src, u32* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(1024))) int lds[6 * 256];       // two supertiles of A fragments
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 6 * 256; e += 256) lds[e] = src[e];
    __syncthreads();
    const i32x4 b0 = *(const i32x4*)(src + 2048 + lane * 4), b1 = *(const i32x4*)(src + 4096 + lane * 4);
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 8388608.0f + 16513.0f * (float)((20 + (lane & 7)) << m3_shift(r)); c1[r] = c0[r] + 16513.0f; }
    const int s = m3_shift((lane & 3) + 4 * ((lane & 31) >> 3));
    const int scale_a = (127 + s) | ((134 + s) << 8) | ((141 + s) << 16);
    const int scale_b = 0x7F7F7F7F;
    u32 x = 0;
    u32 k0 = 0x102040u, k1 = 0x204080u, k2 = 0x408100u;
    asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const i32x4 a0 = *(const i32x4*)(lds + (st * 3 + 0) * 256 + lane * 4);
            const i32x4 a1 = *(const i32x4*)(lds + (st * 3 + 1) * 256 + lane * 4);
            const i32x4 a2 = *(const i32x4*)(lds + (st * 3 + 2) * 256 + lane * 4);
            const i32x8 A0 = {a0.x, a0.y, a0.z, a0.w, 0, 0, 0, 0}, A1 = {a1.x, a1.y, a1.z, a1.w, 0, 0, 0, 0}, A2 = {a2.x, a2.y, a2.z, a2.w, 0, 0, 0, 0};
            const i32x8 B0 = {b0.x, b0.y, b0.z, b0.w, 0, 0, 0, 0}, B1 = {b1.x, b1.y, b1.z, b1.w, 0, 0, 0, 0};
            if (MODE == 0 || MODE == 2) {          // three tiles per accumulator (2: the K masks in VGPRs -> v_and_b32 v, v, v)
                f32x16 p = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, B0, c0, 4, 4, 0, scale_a, 0, scale_b);
                f32x16 q = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, B1, c1, 4, 4, 0, scale_a, 0, scale_b);
                p = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, B0, p, 4, 4, 1, scale_a, 0, scale_b);
                q = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, B1, q, 4, 4, 1, scale_a, 0, scale_b);
                p = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A2, B0, p, 4, 4, 2, scale_a, 0, scale_b);
                q = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A2, B1, q, 4, 4, 2, scale_a, 0, scale_b);
                u32 A, B, C;
                if (MODE == 2) {
                    harvest(p, A, B, C, k0, k1, k2); x ^= A + B + C; asm volatile("" : "+v"(x));
                    harvest(q, A, B, C, k0, k1, k2); x ^= A + B + C; asm volatile("" : "+v"(x));
                } else {
                    harvest(p, A, B, C); x ^= A + B + C; asm volatile("" : "+v"(x));
                    harvest(q, A, B, C); x ^= A + B + C; asm volatile("" : "+v"(x));
                }
            } else {                  // one tile per accumulator: 16 v_alignbit per MFMA
                const i32x8* As[3] = {&A0, &A1, &A2};
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    f32x16 p = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(*As[f], B0, c0, 4, 4, 0, scale_b, 0, scale_b);
                    f32x16 q = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(*As[f], B1, c1, 4, 4, 0, scale_b, 0, scale_b);
                    u32 m = x, n = x;
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = __builtin_amdgcn_alignbit(m, __float_as_uint(p[r]), 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) n = __builtin_amdgcn_alignbit(n, __float_as_uint(q[r]), 31);
                    x = m ^ n; asm volatile("" : "+v"(x));
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd() { u64 z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main() {
    const int NB = 4096;
    std::vector<u64> rows((size_t)NB * 96), qs((size_t)NB * 32);
    std::vector<int> Ts((size_t)NB * 32);
    for (int b = 0; b < NB; ++b) {
        // a mix: i.i.d. rows; rows near the queries (small distances); extreme queries (all ones / zeros); every T in 0..63
        for (int j = 0; j < 32; ++j) {
            u64 q = rnd();
            if (b % 7 == 1) q = 0; else if (b % 7 == 2) q = ~0ull; else if (b % 7 == 3) q &= rnd() & rnd();
            qs[(size_t)b * 32 + j] = q;
            Ts[(size_t)b * 32 + j] = (b % 5 == 0) ? (int)(rnd() % 64) : 18 + (int)(rnd() % 12);
        }
        for (int r = 0; r < 96; ++r) {
            u64 x = rnd();
            if (b % 3 == 1) { x = qs[(size_t)b * 32 + (r % 32)]; for (int k = (int)(rnd() % 30); k > 0; --k) x ^= 1ull << (rnd() % 64); }
            if (b % 11 == 4) x = (r & 1) ? ~0ull : 0ull;
            rows[(size_t)b * 96 + r] = x;
        }
    }
    u64 *drows, *dqs; int* dT; u32 *dbad, *dhits;
    CK(hipMalloc(&drows, rows.size() * 8)); CK(hipMalloc(&dqs, qs.size() * 8)); CK(hipMalloc(&dT, Ts.size() * 4));
    CK(hipMalloc(&dbad, 4)); CK(hipMalloc(&dhits, 4));
    CK(hipMemcpy(drows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dqs, qs.data(), qs.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, Ts.data(), Ts.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dbad, 0, 4)); CK(hipMemset(dhits, 0, 4));
    hipLaunchKernelGGL(k_check, dim3(NB), dim3(64), 0, 0, drows, dqs, dT, dbad, dhits);
    CK(hipDeviceSynchronize());
    u32 bad = 0, hits = 0;
    CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hits, dhits, 4, hipMemcpyDeviceToHost));
    printf("check: %d supertiles x 64 lanes x 48 rows = %lld pairs, %u hits, %u mismatches\n", NB, (long long)NB * 64 * 48, hits, bad);

    int* src; u32* out;
    CK(hipMalloc(&src, 65536)); CK(hipMemset(src, 0x22, 65536)); CK(hipMalloc(&out, 4096 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 1024, iters = 2000;                      // 4 blocks per CU, 4 waves per SIMD
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
            else if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double pairs = (double)blocks * 4 * iters * 2 * 6 * 1024;      // 6 MFMAs of 1024 pairs per supertile and wave
        printf("%-40s %.3f ms   %.2f Tpairs/s   -> 1e10 pairs in %.3f ms\n", mode == 0 ? "3 tiles per accumulator (literal masks)" : mode == 1 ? "1 tile per accumulator (16 alignbit)" : "3 tiles per accumulator (masks in VGPRs)",
               ms, pairs / ms * 1e-9, 1e10 / (pairs / ms));
    }
    return 0;
}
