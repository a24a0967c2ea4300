// Does v_mfma_f32_32x32x2_f32 accumulate like a chain of fp32 FMAs in ascending k?  D = A(32xK) B(Kx32) + 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_mfma(const float *A, const float *B, float *D, int K) {  // A row-major 32xK, B row-major Kx32
    const int lane = threadIdx.x;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) {
        const float a = A[(lane % 32) * K + k + lane / 32];   // A[i = lane%32][k + lane/32]
        const float b = B[(k + lane / 32) * 32 + lane % 32];  // B[k + lane/32][j = lane%32]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r / 4) * 8 + (lane / 32) * 4 + (r % 4), col = lane % 32;
        D[row * 32 + col] = acc[r];
    }
}
int main() {
    const int K = 64;
    std::vector<float> A(32 * K), B(K * 32), D(32 * 32), R1(32 * 32), R2(32 * 32), R3(32*32);
    srand(1);
    for (auto &v : A) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto &v : B) v = (float)rand() / RAND_MAX * 2 - 1;
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dD, K);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0, bad3 = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float c1 = 0.f, c2 = 0.f, c3 = 0.f;
            for (int k = 0; k < K; ++k) c1 = fmaf(A[i * K + k], B[k * 32 + j], c1);       // sequential FMA chain
            for (int k = 0; k < K; k += 2) {                                                  // pairwise: (a0b0 + a1b1) + c
                double p = (double)A[i * K + k] * B[k * 32 + j] + (double)A[i * K + k + 1] * B[(k + 1) * 32 + j];
                c2 = (float)((double)c2 + p);
            }
            for (int k = 0; k < K; ++k) c3 = c3 + A[i * K + k] * B[k * 32 + j];            // mul then add, separately rounded
            bad1 += memcmp(&c1, &D[i * 32 + j], 4) != 0;
            bad2 += memcmp(&c2, &D[i * 32 + j], 4) != 0;
            bad3 += memcmp(&c3, &D[i * 32 + j], 4) != 0;
        }
    printf("mismatches vs sequential-FMA chain: %d / 1024; vs exact-pair-sum: %d; vs mul+add: %d\n", bad1, bad2, bad3);
    return 0;
}
