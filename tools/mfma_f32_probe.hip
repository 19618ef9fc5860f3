// Which float32 arithmetic does v_mfma_f32_32x32x2_f32 perform?  D = C + a0*b0 + a1*b1 per output element, chained over
// K/2 instructions.  Candidates: (A) fma(a1,b1, fma(a0,b0,c)), (B) fma(a0,b0, fma(a1,b1,c)), (C) c + (a0*b0 + a1*b1) with
// rounded products.  Prints the number of mismatching outputs per candidate over random inputs (tanh-like values).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 64;
__global__ void k(const float* A, const float* B, float* D) {   // A [32][K], B [K][32] (B[k][j]), D [32][32]
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int m = 0; m < K / 2; ++m) {
        const float a = A[i * K + 2 * m + h];          // A operand: row i, k = 2m + h
        const float b = B[(2 * m + h) * 32 + i];        // B operand: k = 2m + h, column i
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        D[row * 32 + i] = acc[r];
    }
}
int main() {
    std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
    srand(7);
    for (auto& x : A) x = tanhf((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    for (auto& x : B) x = tanhf((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int badA = 0, badB = 0, badC = 0, badD = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float ca = 0.f, cb = 0.f, cc = 0.f; double cd = 0.0;
            for (int m = 0; m < K / 2; ++m) {
                const float a0 = A[i * K + 2 * m], a1 = A[i * K + 2 * m + 1], b0 = B[(2 * m) * 32 + j], b1 = B[(2 * m + 1) * 32 + j];
                ca = fmaf(a1, b1, fmaf(a0, b0, ca));
                cb = fmaf(a0, b0, fmaf(a1, b1, cb));
                cc = cc + (a0 * b0 + a1 * b1);
                cd += (double)a0 * b0 + (double)a1 * b1;
            }
            const float d = D[i * 32 + j];
            badA += d != ca; badB += d != cb; badC += d != cc; badD += d != (float)cd;
        }
    printf("mismatches of 1024: fma chain k0 then k1: %d | k1 then k0: %d | c + (p0 + p1) rounded: %d | float(double sum): %d\n", badA, badB, badC, badD);
    printf("sample D[0][0]=%.9g\n", D[0]);
    return 0;
}
