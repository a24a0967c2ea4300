// gp_kernels.h — HIP kernels (gfx950, wave64) for the GPOctoMap per-scan path.
//
// Reference (CPU, Eigen):
//   GPRegressor::train      include/gpoctomap/gpregressor.h:42-51   K = Matern(x,x) + noise I, LLT, alpha = K^-1 y
//   GPRegressor::predict    include/gpoctomap/gpregressor.h:80-92   m = Ks^T alpha, v = L^-1 Ks, var = sf2 - diag(v^T v)
//   covMaterniso3           include/gpoctomap/gpregressor.h:114-117 (1 + a) exp(-a) sf2, a = |(1.73205/ell)(x - x')|
//   BCM Occupancy::update   src/gpoctomap/gpoctree_node.cpp:31-49
//   update loop             src/gpoctomap/gpoctomap.cpp:306-319 (unconditional, ExtendedBlock order)
//
// Numerics: every inner product is an fp32 FMA chain in ascending index order — exactly the order
// the oracle uses and, as measured on MI355X (scratch/mfma/mfma_order.hip), the order
// v_mfma_f32_32x32x2_f32 accumulates in — so the results are bit-identical to the CPU restatement
// whether a chain runs on the VALU or on the matrix cores; exp() is the correctly rounded single-precision value
// (f64 library exp rounded once).  Eigen's own LLT/GEMV/packet-exp orders are unpinned.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgk_kernels.h"

namespace la3dm_dev {

struct GpArgs {
    const float4 *pts;          // training points scaled by (float)(1.73205/ell), label in w
    const uint32_t *train_off;
    const uint2 *nbr_range;     // [n_test_blk * 7] {first point, count}
    const int32_t *nbr;         // [n_test_blk * 7] training block index or -1
    const unsigned long long *l_off;  // [n_train_blk] offset (floats) of each block's N x N factor
    float *Lmat;                // Cholesky factors, row-major
    float *alpha_k;             // [n_train_pts] K^-1 y
    const float *blk_center;
    const uint32_t *leaf_off;
    const uint32_t *leaf_key;
    float *m_ivar;
    float *ivar;
    uint8_t *state;
    const float4 *lut;
    float *vscratch;            // [n_tasks][vmax][64] when a block exceeds the LDS capacity, else null
    uint32_t n_test_blk, tpb_shift, n_tasks, n_train_blk;
    uint32_t vmax;              // rows of v per task in vscratch
    float scale;                // (float)(1.73205 / ell)
    float sf2, noise, l, min_ivar, max_ivar, min_known_ivar, free_thresh, occupied_thresh;
};

constexpr int kGpTrainLdsMaxN = 128;  // blocks up to this size are factored by one wave in LDS

__device__ __forceinline__ float cr_expf_dev(float x) { return (float)exp((double)x); }

__device__ __forceinline__ float matern3_dev(float ax, float ay, float az, float bx, float by, float bz, float sf2) {
    const float dx = bx - ax, dy = by - ay, dz = bz - az;
    const float d = sqrtf(dx * dx + (dy * dy + dz * dz));
    return ((1 + d) * cr_expf_dev(-d)) * sf2;
}

__device__ __forceinline__ float matern3_fast(float ax, float ay, float az, float bx, float by, float bz, float sf2);

// scale the training points (x * s, gpregressor.h:115) and resolve neighbour ranges
__global__ void gp_prepare(const float4 *__restrict__ in, float4 *__restrict__ out, uint32_t n, float s,
                           const int32_t *__restrict__ nbr, const uint32_t *__restrict__ train_off,
                           uint2 *__restrict__ nbr_range, uint32_t n_nbr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float4 p = in[i];
        out[i] = make_float4(s * p.x, s * p.y, s * p.z, p.w);
    }
    if (i < n_nbr) {
        const int tb = nbr[i];
        uint2 r = make_uint2(0u, 0u);
        if (tb >= 0) {
            r.x = train_off[tb];
            r.y = train_off[tb + 1] - r.x;
        }
        nbr_range[i] = r;
    }
}

// exclusive scan of N_b^2 (one workgroup; n_train_blk is a few 10^4) + max N_b
__global__ void gp_factor_offsets(const uint32_t *__restrict__ train_off, uint32_t n_blk,
                                  unsigned long long *__restrict__ l_off, unsigned long long *__restrict__ totals) {
    __shared__ unsigned long long s_sum[256];
    __shared__ uint32_t s_max[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_blk + 255u) / 256u;
    const uint32_t b0 = tid * per, b1 = min(n_blk, b0 + per);
    unsigned long long sum = 0;
    uint32_t mx = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        const unsigned long long n = train_off[b + 1] - train_off[b];
        sum += n * n;
        mx = max(mx, (uint32_t)n);
    }
    s_sum[tid] = sum;
    s_max[tid] = mx;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        uint32_t m = 0;
        for (int i = 0; i < 256; ++i) {
            const unsigned long long t = s_sum[i];
            s_sum[i] = run;
            run += t;
            m = max(m, s_max[i]);
        }
        totals[0] = run;
        totals[1] = m;
    }
    __syncthreads();
    unsigned long long run = s_sum[tid];
    for (uint32_t b = b0; b < b1; ++b) {
        const unsigned long long n = train_off[b + 1] - train_off[b];
        l_off[b] = run;
        run += n * n;
    }
}

// GPRegressor::train for one training block per workgroup (256 threads, rows strided over threads).
// K is built in place in the block's N x N slot, factored column by column (FMA chains over k
// ascending), then alpha = L^-T (L^-1 y).
__global__ __launch_bounds__(256) void gp_train_kernel(GpArgs a) {
    const uint32_t b = blockIdx.x;
    const uint32_t p0 = a.train_off[b];
    const int N = (int)(a.train_off[b + 1] - p0);
    if (N == 0 || N <= kGpTrainLdsMaxN) return;  // small blocks: gp_train_wave_kernel
    float *L = a.Lmat + a.l_off[b];
    const float4 *x = a.pts + p0;
    const int tid = threadIdx.x;
    __shared__ float s_d;
    __shared__ float s_z;
    // K(i, j), i >= j; + noise on the diagonal (gpregressor.h:44-46)
    for (int e = tid; e < N * N; e += 256) {
        const int i = e / N, j = e - i * N;
        if (j > i) continue;
        const float4 xi = x[i], xj = x[j];
        float k = matern3_fast(xi.x, xi.y, xi.z, xj.x, xj.y, xj.z, a.sf2);  // dist(x, z)(i, j) = |z_j - x_i|
        if (i == j) k = k + a.noise;
        L[(size_t)i * N + j] = k;
    }
    __syncthreads();
    // LLT (gpregressor.h:47)
    for (int j = 0; j < N; ++j) {
        if (tid == 0) {
            float acc = L[(size_t)j * N + j];
            for (int k = 0; k < j; ++k) acc = __builtin_fmaf(-L[(size_t)j * N + k], L[(size_t)j * N + k], acc);
            s_d = sqrtf(acc);
        }
        __syncthreads();
        const float d = s_d;
        for (int i = j + 1 + tid; i < N; i += 256) {
            float acc = L[(size_t)i * N + j];
            for (int k = 0; k < j; ++k) acc = __builtin_fmaf(-L[(size_t)i * N + k], L[(size_t)j * N + k], acc);
            L[(size_t)i * N + j] = acc / d;
        }
        if (tid == 0) L[(size_t)j * N + j] = d;
        __syncthreads();
    }
    // alpha = llt.solve(y) (gpregressor.h:48): z = L^-1 y (chains over k ascending), alpha = L^-T z
    // (chains over k descending).  Right-looking: rows owned by threads, accumulators in alpha_k.
    float *al = a.alpha_k + p0;
    for (int i = tid; i < N; i += 256) al[i] = x[i].w;
    __syncthreads();
    for (int j = 0; j < N; ++j) {
        if (tid == 0) {
            const float z = al[j] / L[(size_t)j * N + j];
            al[j] = z;
            s_z = z;
        }
        __syncthreads();
        const float z = s_z;
        for (int i = j + 1 + tid; i < N; i += 256) al[i] = __builtin_fmaf(-L[(size_t)i * N + j], z, al[i]);
        __syncthreads();
    }
    for (int j = N - 1; j >= 0; --j) {
        if (tid == 0) {
            const float v = al[j] / L[(size_t)j * N + j];
            al[j] = v;
            s_z = v;
        }
        __syncthreads();
        const float v = s_z;
        for (int i = tid; i < j; i += 256) al[i] = __builtin_fmaf(-L[(size_t)j * N + i], v, al[i]);
        __syncthreads();
    }
}

// Same arithmetic as gp_train_kernel for blocks with N <= 128: one wave64 per training block, the lower
// triangle packed in LDS (row i at i(i+1)/2), lane = row (two rows per lane above 64), no workgroup barriers.
// The factor is written to its global slot once, for the predict kernel.
__global__ __launch_bounds__(kWave) void gp_train_wave_kernel(GpArgs a) {
    extern __shared__ float s_l[];  // [N(N+1)/2 + N]
    const uint32_t b = blockIdx.x;
    const uint32_t p0 = a.train_off[b];
    const int N = (int)(a.train_off[b + 1] - p0);
    if (N == 0 || N > kGpTrainLdsMaxN) return;  // large blocks: gp_train_kernel
    const int lane = threadIdx.x;
    const float4 *x = a.pts + p0;
    float *Lg = a.Lmat + a.l_off[b];
    float *zs = s_l + (N * (N + 1)) / 2;  // right-hand side / solution
    auto tri = [](int i, int j) { return (i * (i + 1)) / 2 + j; };
    // K(i, j), i >= j, + noise on the diagonal
    for (int i = lane; i < N; i += kWave) {
        const float4 xi = x[i];
        for (int j = 0; j <= i; ++j) {
            const float4 xj = x[j];
            float kv = matern3_fast(xi.x, xi.y, xi.z, xj.x, xj.y, xj.z, a.sf2);
            if (i == j) kv = kv + a.noise;
            s_l[tri(i, j)] = kv;
        }
        zs[i] = xi.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < N; ++j) {
        // diagonal (every lane computes the same chain; wave-uniform result)
        float dacc = s_l[tri(j, j)];
        for (int k = 0; k < j; ++k) {
            const float ljk = s_l[tri(j, k)];
            dacc = __builtin_fmaf(-ljk, ljk, dacc);
        }
        const float d = sqrtf(dacc);
        for (int i = j + 1 + lane; i < N; i += kWave) {
            float acc = s_l[tri(i, j)];
            for (int k = 0; k < j; ++k) acc = __builtin_fmaf(-s_l[tri(i, k)], s_l[tri(j, k)], acc);
            s_l[tri(i, j)] = acc / d;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) s_l[tri(j, j)] = d;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // forward then backward substitution, right-looking (chains over k ascending / descending)
    for (int j = 0; j < N; ++j) {
        const float z = zs[j] / s_l[tri(j, j)];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) zs[j] = z;
        for (int i = j + 1 + lane; i < N; i += kWave) zs[i] = __builtin_fmaf(-s_l[tri(i, j)], z, zs[i]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int j = N - 1; j >= 0; --j) {
        const float v = zs[j] / s_l[tri(j, j)];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) zs[j] = v;
        for (int i = lane; i < j; i += kWave) zs[i] = __builtin_fmaf(-s_l[tri(j, i)], v, zs[i]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = lane; i < N; i += kWave) {
        a.alpha_k[p0 + i] = zs[i];
        for (int j = 0; j <= i; ++j) Lg[(size_t)i * N + j] = s_l[tri(i, j)];
    }
}

// GP node update, src/gpoctomap/gpoctree_node.cpp:36-49 (double expression for ivar, double exp)
__device__ __forceinline__ void gp_node_update_dev(const GpArgs &a, float &m_ivar, float &ivar, uint8_t &state, float new_m,
                                                   float new_var) {
    ivar = (float)((double)ivar + (1.0 / (double)new_var - (double)a.sf2));
    m_ivar += new_m / new_var;
    if (ivar < a.min_known_ivar) {
        state = 2;
    } else {
        ivar = ivar > a.max_ivar ? a.max_ivar : ivar;
        const float p = 1.0f / (1.0f + (float)exp((double)(-a.l * m_ivar / a.max_ivar)));
        state = p > a.occupied_thresh ? 1 : (p < a.free_thresh ? 0 : 2);
    }
}

// exp(x) for x in [-60, 0], correctly rounded to f32 in all but ~1e-7 of the cases (the same contract
// as the oracle's (float)exp((double)x)): x = n ln2 + r, |r| <= ln2/2, degree-13 Taylor/Horner in
// f64 (truncation < 2e-17 relative), scaled by 2^n through the exponent field.
__device__ __forceinline__ float exp_cr_dev(float xf) {
    const double x = (double)xf;
    const double n = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);   // ln2 high part
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);          // ln2 low part
    double p = 1.6059043836821613e-10;                             // 1/13!
    p = __builtin_fma(p, r, 2.08767569878680990e-09);              // 1/12!
    p = __builtin_fma(p, r, 2.50521083854417188e-08);              // 1/11!
    p = __builtin_fma(p, r, 2.75573192239858907e-07);              // 1/10!
    p = __builtin_fma(p, r, 2.75573192239858907e-06);              // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873016e-05);              // 1/8!
    p = __builtin_fma(p, r, 1.98412698412698413e-04);              // 1/7!
    p = __builtin_fma(p, r, 1.38888888888888889e-03);              // 1/6!
    p = __builtin_fma(p, r, 8.33333333333333333e-03);              // 1/5!
    p = __builtin_fma(p, r, 4.16666666666666667e-02);              // 1/4!
    p = __builtin_fma(p, r, 1.66666666666666667e-01);              // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const long long bits = (long long)__double_as_longlong(p) + ((long long)(int)n << 52);  // p in [0.7, 1.5], n >= -87
    return (float)__longlong_as_double(bits);
}

__device__ __forceinline__ float matern3_fast(float ax, float ay, float az, float bx, float by, float bz, float sf2) {
    const float dx = bx - ax, dy = by - ay, dz = bz - az;
    const float d = sqrtf(dx * dx + (dy * dy + dz * dz));
    return ((1 + d) * exp_cr_dev(-d)) * sf2;
}

// ---------------------------------------------------------------------------
// v = L^-1 Ks for training blocks with N > 64 on the matrix cores, bit-identical to the FMA chains.
//
// v_mfma_f32_32x32x2_f32 accumulates D = C + A B as a chain of fp32 FMAs over k ascending (measured: 0 mismatches
// against fmaf chains, scratch/mfma/mfma_order.hip), so a blocked forward substitution whose off-diagonal updates
//      C[K] = Ks[K] - sum_{J<K} L[K][J] V[J]            (J ascending, k ascending inside a block)
// run on MFMA reproduces  acc = fma(-L_ki, v_i, acc), i = 0..k-1  of the oracle exactly; the 32 x 32 diagonal
// blocks continue each chain on the VALU.  Layout: one wave handles the tile's 64 leaves as two 32-column
// accumulators; lane (c = lane % 32, h = lane / 32) owns rows 8g + 4h + j of a block for leaves c and 32 + c, so the
// serial row order alternates between the half-waves every four rows: chains (m, sum v^2) and the freshly solved
// v are handed over with a lane ^ 32 exchange.  V lives in the task's global scratch [row][64] (L2 resident),
// which is also the B operand of the MFMAs; L is read in place (A operand, negated on load).
// Returns m = Ks^T alpha and ss = sum v_k^2 for leaf = lane.
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }

__device__ __forceinline__ void gp_solve_mfma(const GpArgs &a, const float *__restrict__ L, const float4 *__restrict__ x,
                                              const float *__restrict__ al, const int N, const float tx, const float ty,
                                              const float tz, float *__restrict__ vg, const int lane, float &mj_out,
                                              float &ss_out) {
    const int c = lane & 31, h = lane >> 5;
    const float t0x = __shfl(tx, c), t0y = __shfl(ty, c), t0z = __shfl(tz, c);
    const float t1x = __shfl(tx, 32 + c), t1y = __shfl(ty, 32 + c), t1z = __shfl(tz, 32 + c);
    float mj0 = 0.0f, mj1 = 0.0f, ss0 = 0.0f, ss1 = 0.0f;  // the live copies sit in half 0 at every block start
    const int nblk = (N + 31) >> 5;
    for (int K = 0; K < nblk; ++K) {
        const int R0 = 32 * K;
        f32x16 C0, C1;
        float alv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // Ks(k, j) = k(x_k, xs_j) in accumulator layout
            const int row = R0 + 8 * (r >> 2) + 4 * h + (r & 3);
            const bool valid = row < N;
            const float4 xr = valid ? x[row] : make_float4(0.f, 0.f, 0.f, 0.f);
            C0[r] = valid ? matern3_fast(xr.x, xr.y, xr.z, t0x, t0y, t0z, a.sf2) : 0.0f;
            C1[r] = valid ? matern3_fast(xr.x, xr.y, xr.z, t1x, t1y, t1z, a.sf2) : 0.0f;
            alv[r] = valid ? al[row] : 0.0f;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // m = Ks^T alpha, rows in order: 4 rows in half 0, 4 rows in half 1, ...
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (h == hh) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        mj0 = __builtin_fmaf(C0[4 * g + j], alv[4 * g + j], mj0);
                        mj1 = __builtin_fmaf(C1[4 * g + j], alv[4 * g + j], mj1);
                    }
                }
                mj0 = xhalf(mj0);
                mj1 = xhalf(mj1);
            }
        }
        // off-diagonal blocks on the matrix cores: C -= L[K][J] V[J]
        {
            const int arow = R0 + c;
            const bool avalid = arow < N;
            const float *Lrow = L + (size_t)(avalid ? arow : 0) * N;
            for (int J = 0; J < K; ++J) {
#pragma unroll 4
                for (int m2 = 0; m2 < 16; ++m2) {
                    const int kcol = 32 * J + 2 * m2 + h;
                    const float av = avalid ? -Lrow[kcol] : 0.0f;
                    const float b0 = vg[(size_t)kcol * kWave + c], b1 = vg[(size_t)kcol * kWave + 32 + c];
                    C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, C0, 0, 0, 0);
                    C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, C1, 0, 0, 0);
                }
            }
        }
        // diagonal block: rows in order, four at a time in alternating half-waves
        float vb0[32], vb1[32];
#pragma unroll
        for (int G = 0; G < 8; ++G) {
            const int hh = G & 1, base = 4 * (G >> 1);
            float n0[4], n1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * G + j;
                const int row = R0 + r;
                const bool valid = row < N;  // wave-uniform
                const float *Ld = L + (size_t)(valid ? row : 0) * N + R0;
                float acc0 = C0[base + j], acc1 = C1[base + j];
#pragma unroll
                for (int w = 0; w < r; ++w) {
                    const float lw = valid ? Ld[w] : 0.0f;
                    acc0 = __builtin_fmaf(-lw, vb0[w], acc0);
                    acc1 = __builtin_fmaf(-lw, vb1[w], acc1);
                }
                const float d = valid ? Ld[r] : 1.0f;
                const float v0 = acc0 / d, v1 = acc1 / d;
                vb0[r] = v0;  // right in the owning half; the other half is repaired after the group
                vb1[r] = v1;
                n0[j] = v0;
                n1[j] = v1;
                if (h == hh) {
                    ss0 = __builtin_fmaf(v0, v0, ss0);
                    ss1 = __builtin_fmaf(v1, v1, ss1);
                    if (valid) {
                        vg[(size_t)row * kWave + c] = v0;
                        vg[(size_t)row * kWave + 32 + c] = v1;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float o0 = xhalf(n0[j]), o1 = xhalf(n1[j]);
                vb0[4 * G + j] = h == hh ? n0[j] : o0;
                vb1[4 * G + j] = h == hh ? n1[j] : o1;
            }
            ss0 = xhalf(ss0);
            ss1 = xhalf(ss1);
        }
        // V[K] is read back as MFMA B operands by the other lanes of this wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const float m1x = xhalf(mj1), s1x = xhalf(ss1);
    mj_out = h == 0 ? mj0 : m1x;
    ss_out = h == 0 ? ss0 : s1x;
}

// GPRegressor::predict + BCM fusion: one wave64 per leaf tile, lane = leaf (test point).
// v = L^-1 Ks per lane: v_k = (Ks_k - sum_{i<k} L_ki v_i) / L_kk.  Four rows at a time: their FMA chains
// over the already solved i are independent (latency hidden, one LDS read of v_i feeds four rows), each
// chain still runs over i ascending — the oracle's order.  Rows of L are loaded with coalesced vector
// loads (lane = column) and broadcast with v_readlane; v lives in LDS [row][lane].  Blocks with more than 64
// rows go through gp_solve_mfma (above).
constexpr int kGpLdsRows = 64;  // blocks up to one wave of rows are solved in LDS; larger ones on the matrix cores

__global__ __launch_bounds__(kWave) void gp_predict_fuse_kernel(GpArgs a) {
    extern __shared__ float s_vraw[];  // [min(max N, kGpLdsRows)][64]: sized per launch, LDS is the occupancy limiter
    float (*s_v)[kWave] = reinterpret_cast<float (*)[kWave]>(s_vraw);
    const int lane = threadIdx.x;
    const uint32_t task = blockIdx.x;
    if (task >= a.n_tasks) return;
    const uint32_t blk = task >> a.tpb_shift;
    const uint32_t tile = task & ((1u << a.tpb_shift) - 1u);
    const uint32_t l0 = a.leaf_off[blk] + tile * kWave;
    const uint32_t l1 = a.leaf_off[blk + 1];
    if (l0 >= l1) return;
    const uint32_t nl = min(l1 - l0, (uint32_t)kWave);
    const bool active = (uint32_t)lane < nl;
    const uint32_t li = l0 + (active ? lane : 0);
    const uint32_t key = a.leaf_key[li];
    const float4 off4 = a.lut[lut_layer_base(key >> 16) + (key & 0xFFFFu)];
    // 1.73205/ell * xs (gpregressor.h:115): position first (LUT + centre), then the scale
    const float tx = a.scale * (off4.x + a.blk_center[3 * blk + 0]);
    const float ty = a.scale * (off4.y + a.blk_center[3 * blk + 1]);
    const float tz = a.scale * (off4.z + a.blk_center[3 * blk + 2]);
    float m_ivar = a.m_ivar[li], ivar = a.ivar[li];
    uint8_t state = 2;
    bool updated = false;
    float *vg = a.vscratch ? a.vscratch + (size_t)task * a.vmax * kWave : nullptr;

    for (int nb = 0; nb < 7; ++nb) {
        const uint2 r = a.nbr_range[7 * blk + nb];
        const int N = (int)r.y;
        if (N == 0) continue;
        const int tb = a.nbr[7 * blk + nb];
        const float *L = a.Lmat + a.l_off[tb];
        const float4 *x = a.pts + r.x;
        const float *al = a.alpha_k + r.x;
        float mj = 0.0f, ss = 0.0f;
        if (N <= kWave) {
            // fast path: a row of L fits one register (lane = column)
            for (int k0 = 0; k0 < N; k0 += 4) {
                float Lr[4], acc[4], ks[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = min(k0 + u, N - 1);
                    Lr[u] = lane <= k ? L[(size_t)k * N + lane] : 0.0f;
                    const float4 xk = x[k];
                    ks[u] = matern3_fast(xk.x, xk.y, xk.z, tx, ty, tz, a.sf2);  // Ks(k, j) = k(x_k, xs_j)
                    acc[u] = ks[u];
                }
                for (int i = 0; i < k0; ++i) {  // four independent chains over the solved rows
                    const float vi = s_v[i][lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[u] = __builtin_fmaf(-__int_as_float(__builtin_amdgcn_readlane(__float_as_int(Lr[u]), i)), vi, acc[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {  // the 4x4 triangle
                    const int k = k0 + u;
                    if (k < N) {
#pragma unroll
                        for (int w = 0; w < u; ++w)
                            acc[u] = __builtin_fmaf(-__int_as_float(__builtin_amdgcn_readlane(__float_as_int(Lr[u]), k0 + w)),
                                                    s_v[k0 + w][lane], acc[u]);
                        const float vk = acc[u] / __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Lr[u]), k));
                        s_v[k][lane] = vk;
                        mj = __builtin_fmaf(ks[u], al[k], mj);
                        ss = __builtin_fmaf(vk, vk, ss);
                    }
                }
            }
        } else {
            gp_solve_mfma(a, L, x, al, N, tx, ty, tz, vg, lane, mj, ss);
        }
        const float var = a.sf2 - ss;
        gp_node_update_dev(a, m_ivar, ivar, state, mj, var);
        updated = true;
    }
    if (active) {
        if (updated) {
            a.m_ivar[li] = m_ivar;
            a.ivar[li] = ivar;
            a.state[li] = (uint8_t)(state | 0x80u);
        } else {
            a.state[li] = 0;
        }
    }
}

}  // namespace la3dm_dev
