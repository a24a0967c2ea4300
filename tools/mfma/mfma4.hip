// layout + exactness probe of v_mfma_f32_4x4x1f32 (16 blocks of 4x4, K = 1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, const float *c0, float *out, int steps) {
    const int l = threadIdx.x;
    f32x4 acc = {c0[l], c0[64 + l], c0[128 + l], c0[192 + l]};
    for (int s = 0; s < steps; ++s) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s * 64 + l], b[s * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = acc[r];
}
int main() {
    const int S = 40;
    float ha[S * 64], hb[S * 64], hc[256], ho[256];
    srand(3);
    for (int i = 0; i < S * 64; ++i) { ha[i] = (float)rand() / RAND_MAX * 2 - 1; hb[i] = ((float)rand() / RAND_MAX) * 0.1f; }
    for (int i = 0; i < 256; ++i) hc[i] = (float)rand() / RAND_MAX;
    float *da, *db, *dc, *dout;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(hc)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc, dout, S);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    // hypothesis: D[r] of lane l = chain over s of fma(A[s][4*(l/4) + r], B[s][l], .)
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            float c = hc[r * 64 + l];
            for (int s = 0; s < S; ++s) c = fmaf(ha[s * 64 + 4 * (l / 4) + r], hb[s * 64 + l], c);
            bad += memcmp(&c, &ho[r * 64 + l], 4) != 0;
        }
    printf("4x4x1: mismatches vs hypothesis (A row r from lane 4*(l/4)+r, B from own lane, fma chain): %d / 256\n", bad);
    return 0;
}
