// gp_eigen_kernels.h — GPOctoMap, option "gp_mode" 1 (round 6): the regressor in the order of operations an x86-64 / SSE2
// (ROS Noetic) build of Eigen 3.3.7 most plausibly runs, on the VALU — no FMA anywhere, 4-lane packet inner products,
// llt_inplace's unblocked / blocked factorisation, triangular solves in panels of 8 with reciprocal diagonals, exp() as the
// SSE packet pexp.  It is the device counterpart of the restatement's orc_set_gp_mode(1) (oracle/la3dm_oracle.cpp
// "GP mode 1": gp_train_eigen / gp_predict_eigen, which document where each rule comes from in Eigen) and BIT-IDENTICAL to
// it (tests/test_gp_gpu.py) for training blocks of up to kGpEigenMaxN points — configs[2] at the YAML's block_depth 3
// (N <= 79).  Larger blocks (block_depth 4: N up to 531) are refused in this mode: they belong to the matrix-core path,
// whose accumulation order is the FMA chain of mode 0.
//
// Reference: GPRegressor::train / predict  include/gpoctomap/gpregressor.h:42-51, 80-92; covMaterniso3 :114-117.
// This translation unit is built with -ffp-contract=off: `a * b + c` below is a rounded product and a rounded sum.
#pragma once
#include "gp_kernels.h"

namespace la3dm_dev {

constexpr int kGpEigenMaxN = 128;   // one wave per block, the packed lower triangle in LDS (33 KB at 128)

// pexp<Packet4f> of Eigen 3.3.7 (arch/SSE/MathFunctions.h, the Cephes expf), one lane, SSE2 path — oracle: orc_eigen337::pexp
__device__ __forceinline__ float pexp_eigen_dev(float x0) {
    float x = 88.3762626647950f < x0 ? 88.3762626647950f : x0;     // std::min(x0, hi)
    x = x < -88.3762626647949f ? -88.3762626647949f : x;           // std::max(x, lo)
    float fx = x * 1.44269504088896341f + 0.5f;
    float tmp = (float)(int32_t)fx;
    if (tmp > fx) tmp = tmp - 1.0f;
    fx = tmp;
    tmp = fx * 0.693359375f;
    float z = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - z;
    z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x + 1.3981999507E-3f;
    y = y * x + 8.3334519073E-3f;
    y = y * x + 4.1665795894E-2f;
    y = y * x + 1.6666665459E-1f;
    y = y * x + 5.0000001201E-1f;
    y = y * z + x;
    y = y + 1.0f;
    const int32_t e = ((int32_t)fx + 0x7f) << 23;
    const float r = y * __int_as_float(e);
    return r < x0 ? x0 : r;                                         // std::max(r, x0)
}
// matern3_eigen(a, b): d = |b - a| with the fixed-size-3 reduction's association, ((1 + d) pexp(-d)) sf2
__device__ __forceinline__ float matern3_eigen_dev(float ax, float ay, float az, float bx, float by, float bz, float sf2) {
    const float dx = bx - ax, dy = by - ay, dz = bz - az;
    const float d = sqrtf(dx * dx + (dy * dy + dz * dz));
    return ((1 + d) * pexp_eigen_dev(-d)) * sf2;
}
// dot_sse of the restatement: two 4-lane accumulators over the aligned part, p0 + p1, (s0 + s2) + (s1 + s3), the tail one
// by one.  A(i), B(i) fetch the operands.
template <class FA, class FB>
__device__ __forceinline__ float dot_sse_dev(FA &&A, FB &&B, int n) {
    const int n4 = n & ~3, n8 = n & ~7;
    float res = 0.0f;
    if (n4) {
        float p0[4], p1[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) p0[l] = A(l) * B(l);
        if (n4 > 4) {
#pragma unroll
            for (int l = 0; l < 4; ++l) p1[l] = A(4 + l) * B(4 + l);
            for (int i = 8; i < n8; i += 8) {
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    p0[l] = p0[l] + A(i + l) * B(i + l);
                    p1[l] = p1[l] + A(i + 4 + l) * B(i + 4 + l);
                }
            }
#pragma unroll
            for (int l = 0; l < 4; ++l) p0[l] = p0[l] + p1[l];
            if (n4 > n8) {
#pragma unroll
                for (int l = 0; l < 4; ++l) p0[l] = p0[l] + A(n8 + l) * B(n8 + l);
            }
        }
        res = (p0[0] + p0[2]) + (p0[1] + p0[3]);
    }
    for (int i = n4; i < n; ++i) res = res + A(i) * B(i);
    return res;
}

__device__ __forceinline__ void gp_eigen_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// GPRegressor::train in Eigen's order: one wave64 per training block, the packed lower triangle (row i at i (i + 1) / 2) and the
// right-hand side in LDS.  Launched in the same size classes as gp_train_wave_kernel (n_lo < N <= n_hi, LDS sized for n_hi).
__global__ __launch_bounds__(kWave) void gp_train_eigen_kernel(GpArgs a, int n_lo, int n_hi) {
    extern __shared__ __attribute__((aligned(16))) float s_l[];  // [N (N + 1) / 2 + N] + the block's points [N] float4 (gp_train_wave_lds)
    if (blockIdx.x >= (uint32_t)a.totals[3]) return;
    const uint32_t b = a.order[(uint32_t)a.totals[2] + blockIdx.x];
    const uint32_t p0g = a.train_off[b];
    const int N = (int)(a.train_off[b + 1] - p0g);
    if (N <= n_lo || N > n_hi) return;
    const int lane = threadIdx.x;
    const float4 *x = a.pts + p0g;
    float *Lg = a.Lmat + a.l_off[b];
    const int T = (N * (N + 1)) / 2;
    float *xs = s_l + T;  // right-hand side / solution
    float4 *s_x = reinterpret_cast<float4 *>(s_l + ((T + N + 3) & ~3));
    auto tri = [](int i, int j) { return (i * (i + 1)) / 2 + j; };
    for (int i = lane; i < N; i += kWave) {
        const float4 xi = x[i];
        s_x[i] = xi;
        xs[i] = xi.w;
    }
    gp_eigen_sync();
    // K(i, j) = matern3_eigen(x_i, x_j), i >= j, + noise on the diagonal (gp_train_eigen)
    for (int e = lane; e < T; e += kWave) {
        int i = (int)((__builtin_sqrtf((float)(8 * e + 1)) - 1.0f) * 0.5f);
        if ((i * (i + 1)) / 2 > e) --i;
        if (((i + 1) * (i + 2)) / 2 <= e) ++i;
        const int j = e - (i * (i + 1)) / 2;
        const float4 xi = s_x[i], xj = s_x[j];
        float kv = matern3_eigen_dev(xi.x, xi.y, xi.z, xj.x, xj.y, xj.z, a.sf2);
        if (i == j) kv = kv + a.noise;
        s_l[e] = kv;
    }
    gp_eigen_sync();
    // llt_inplace<float, Lower>::unblocked on the n x n block at (o, o)
    auto unblocked = [&](int o, int n) {
        for (int k = 0; k < n; ++k) {
            const float *rk = s_l + tri(o + k, o);
            float xd = rk[k];
            if (k > 0) xd = xd - dot_sse_dev([&](int q) { return rk[q]; }, [&](int q) { return rk[q]; }, k);
            xd = sqrtf(xd);
            float vnew[2] = {0.f, 0.f};   // (rows k + 1 + lane, k + 1 + lane + 64: a block has at most 128 rows)
            for (int r = 0; r < 2; ++r) {
                const int i = k + 1 + lane + r * kWave;
                if (i < n) {
                    const float *ri = s_l + tri(o + i, o);
                    float v = ri[k];
                    if (k > 0) v = v - dot_sse_dev([&](int q) { return ri[q]; }, [&](int q) { return rk[q]; }, k);
                    vnew[r] = v / xd;
                }
            }
            gp_eigen_sync();   // (every lane has read row k's diagonal entry before it is replaced)
            if (lane == 0) s_l[tri(o + k, o + k)] = xd;
            for (int r = 0; r < 2; ++r) {
                const int i = k + 1 + lane + r * kWave;
                if (i < n) s_l[tri(o + i, o + k)] = vnew[r];
            }
            gp_eigen_sync();
        }
    };
    if (N < 32) {
        unblocked(0, N);
    } else {
        int bs = N / 8;
        bs = (bs / 16) * 16;
        bs = min(max(bs, 8), 128);
        for (int k0 = 0; k0 < N; k0 += bs) {
            const int nb = min(bs, N - k0), rs = N - k0 - nb;
            unblocked(k0, nb);
            if (rs > 0) {
                // trsm_right_eigen: rows [k0 + nb, N) of A21 <- A21 A11^-T, lane = row, panels of 8 columns
                for (int r = 0; r < 2; ++r) {
                    const int i = k0 + nb + lane + r * kWave;
                    if (i < N) {
                        float *ri = s_l + tri(i, k0);
                        for (int pp = 0; pp < nb; pp += 8) {
                            const int pw = min(8, nb - pp);
                            for (int k = 0; k < pw; ++k) {
                                const float *lk = s_l + tri(k0 + pp + k, k0);
                                float bb = 0.0f;
                                for (int q = 0; q < k; ++q) bb = bb + lk[pp + q] * ri[pp + q];
                                ri[pp + k] = (ri[pp + k] - bb) * (1.0f / lk[pp + k]);
                            }
                            for (int c = pp + pw; c < nb; ++c) {
                                const float *lc = s_l + tri(k0 + c, k0);
                                float acc = 0.0f;
                                for (int q = 0; q < pw; ++q) acc = acc + ri[pp + q] * lc[pp + q];
                                ri[c] = ri[c] - acc;
                            }
                        }
                    }
                }
                gp_eigen_sync();
                // A22 -= A21 A21^T (lower part), every entry one from-zero accumulation over the panel's columns
                const int m = rs, np = (m * (m + 1)) / 2;
                for (int e = lane; e < np; e += kWave) {
                    int ii = (int)((__builtin_sqrtf((float)(8 * e + 1)) - 1.0f) * 0.5f);
                    if ((ii * (ii + 1)) / 2 > e) --ii;
                    if (((ii + 1) * (ii + 2)) / 2 <= e) ++ii;
                    const int jj = e - (ii * (ii + 1)) / 2;
                    const int i = k0 + nb + ii, j = k0 + nb + jj;
                    const float *ai = s_l + tri(i, k0), *aj = s_l + tri(j, k0);
                    float acc = 0.0f;
                    for (int q = 0; q < nb; ++q) acc = acc + ai[q] * aj[q];
                    s_l[tri(i, j)] = s_l[tri(i, j)] - acc;
                }
                gp_eigen_sync();
            }
        }
    }
    // alpha = L^-T (L^-1 y): trsv_lower_eigen, trsv_upper_eigen — the panel's own rows in sequence (every lane the same values),
    // the rows outside it lane by lane
    for (int pp = 0; pp < N; pp += 8) {
        const int pw = min(8, N - pp);
        float xp[8];
        for (int k = 0; k < pw; ++k) {
            const float *lk = s_l + tri(pp + k, 0);
            float bb = 0.0f;
            for (int q = 0; q < k; ++q) bb = bb + lk[pp + q] * xp[q];
            xp[k] = (xs[pp + k] - bb) * (1.0f / lk[pp + k]);
        }
        float upd[2] = {0.f, 0.f};
        for (int r = 0; r < 2; ++r) {
            const int i = pp + pw + lane + r * kWave;
            if (i < N) {
                const float *li = s_l + tri(i, 0);
                float acc = 0.0f;
                for (int q = 0; q < pw; ++q) acc = acc + li[pp + q] * xp[q];
                upd[r] = xs[i] - acc;
            }
        }
        gp_eigen_sync();
        if (lane < pw) xs[pp + lane] = xp[lane];
        for (int r = 0; r < 2; ++r) {
            const int i = pp + pw + lane + r * kWave;
            if (i < N) xs[i] = upd[r];
        }
        gp_eigen_sync();
    }
    for (int p1 = N; p1 > 0; p1 -= 8) {
        const int pw = min(8, p1), pp = p1 - pw;
        float xp[8];
        for (int k = pw - 1; k >= 0; --k) {
            float bb = 0.0f;
            for (int q = pw - 1; q > k; --q) bb = bb + s_l[tri(pp + q, pp + k)] * xp[q];
            xp[k] = (xs[pp + k] - bb) * (1.0f / s_l[tri(pp + k, pp + k)]);
        }
        float upd[2] = {0.f, 0.f};
        for (int r = 0; r < 2; ++r) {
            const int i = lane + r * kWave;
            if (i < pp) {
                float acc = 0.0f;
                for (int q = 0; q < pw; ++q) acc = acc + s_l[tri(pp + q, i)] * xp[q];
                upd[r] = xs[i] - acc;
            }
        }
        gp_eigen_sync();
        if (lane < pw) xs[pp + lane] = xp[lane];
        for (int r = 0; r < 2; ++r) {
            const int i = lane + r * kWave;
            if (i < pp) xs[i] = upd[r];
        }
        gp_eigen_sync();
    }
    for (int i = lane; i < N; i += kWave) a.alpha_k[p0g + i] = xs[i];
    for (int e = lane; e < T; e += kWave) {   // the factor to its global slot (row-major N x N, lower triangle)
        int i = (int)((__builtin_sqrtf((float)(8 * e + 1)) - 1.0f) * 0.5f);
        if ((i * (i + 1)) / 2 > e) --i;
        if (((i + 1) * (i + 2)) / 2 <= e) ++i;
        Lg[(size_t)i * N + (e - (i * (i + 1)) / 2)] = s_l[e];
    }
}

// GPRegressor::predict + the BCM update in Eigen's order (gp_predict_eigen): one wave per 64-leaf tile, lane = leaf; a leaf's
// v = Ks column lives in LDS column `lane` ([N][64]); the factor's entries are the same for every lane (uniform loads).
__global__ __launch_bounds__(kWave) void gp_predict_fuse_eigen_kernel(GpArgs a) {
    extern __shared__ float s_vraw[];  // [max N][64]
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
    const float tx = a.scale * (off4.x + a.blk_center[3 * blk + 0]);
    const float ty = a.scale * (off4.y + a.blk_center[3 * blk + 1]);
    const float tz = a.scale * (off4.z + a.blk_center[3 * blk + 2]);
    float m_ivar = a.m_ivar[li], ivar = a.ivar[li];
    bool updated = false, unknown = true;
    for (int nb = 0; nb < 7; ++nb) {
        const uint2 r = a.nbr_range[7 * blk + nb];
        const int N = __builtin_amdgcn_readfirstlane((int)r.y);
        if (N == 0) continue;
        const int tb = a.nbr[7 * blk + nb];
        const float *L = a.Lmat + a.l_off[tb];
        const float4 *x = a.pts + r.x;
        const float *al = a.alpha_k + r.x;   // (L, x, alpha: the same address in every lane.  As SCALAR loads — the pointers made uniform with
                                            //  v_readfirstlane — the kernel was slower, 5.0 against 4.4 ms: s_load and ds_read share lgkmcnt, and
                                            //  every wait for an LDS column then also waits for the scalar loads in flight)
        for (int k = 0; k < N; ++k) {
            const float4 xk = x[k];
            s_v[k][lane] = matern3_eigen_dev(xk.x, xk.y, xk.z, tx, ty, tz, a.sf2);   // matern3_eigen(&xn[3 k], t)
        }
        const float mj = dot_sse_dev([&](int q) { return s_v[q][lane]; }, [&](int q) { return al[q]; }, N);   // (Ks^T alpha)(j)
        // trsv_lower_eigen on the lane's own column
        for (int pp = 0; pp < N; pp += 8) {
            const int pw = min(8, N - pp);
            float xp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < pw) {
                    const float *lk = L + (size_t)(pp + k) * N + pp;
                    float bb = 0.0f;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q < k) bb = bb + lk[q] * xp[q];
                    xp[k] = (s_v[pp + k][lane] - bb) * (1.0f / lk[k]);
                    s_v[pp + k][lane] = xp[k];
                }
            }
            if (pw == 8) {
                for (int i = pp + 8; i < N; ++i) {
                    const float *lr = L + (size_t)i * N + pp;
                    float acc = 0.0f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = acc + lr[q] * xp[q];
                    s_v[i][lane] = s_v[i][lane] - acc;
                }
            }
        }
        const float ss = dot_sse_dev([&](int q) { return s_v[q][lane]; }, [&](int q) { return s_v[q][lane]; }, N);
        const float var = a.sf2 - ss;
        gp_node_accumulate_dev(a, m_ivar, ivar, unknown, mj, var);
        updated = true;
    }
    if (active) {
        if (updated) {
            a.m_ivar[li] = m_ivar;
            a.ivar[li] = ivar;
            a.state[li] = (uint8_t)(gp_node_state_dev(a, m_ivar, unknown) | 0x80u);
        } else {
            a.state[li] = 0;
        }
    }
}

}  // namespace la3dm_dev
