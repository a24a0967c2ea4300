// bgk_kernels.h — HIP kernels (gfx950 / CDNA4, wave64) for the BGK predict + fuse path.
//
// What the kernels compute (reference, CPU):
//   BGKInference::predict        include/bgkoctomap/bgkinference.h:73-79
//   dist / covSparse             include/bgkoctomap/bgkinference.h:88-93, 113-126
//   7-neighbour update loop      src/bgkoctomap/bgkoctomap.cpp:314-335
//   Occupancy::update / get_var  src/bgkoctomap/bgkoctree_node.cpp:31-44, bgkoctree_node.h:60
//   Block::get_loc               include/bgkoctomap/bgkblock.h:64-66
//
// Numerics contract: strict fp32, the reference's operation order,
// no FMA contraction in the reference-visible expressions (this translation unit is
// built with -ffp-contract=off; every fused multiply-add below is an explicit
// __builtin_fmaf in code that has no reference counterpart), IEEE division and
// square root (hipcc's default correctly rounded f32 div/sqrt), truncated pi
// 3.1415926f.
//
// Work decomposition: one wave64 = one "leaf tile" = up to 64 consecutive leaves of one
// test block in LeafIterator order (for an un-pruned block, 64 consecutive leaves are
// one 4x4x4 voxel cube).  Lane = leaf.  The wave walks the <=7 neighbour training
// blocks in ExtendedBlock order; for each it streams the block-contiguous, pre-scaled
// training points with coalesced 16-byte loads (lane = point), culls them against the
// tile's bounding box (a point farther than ell from the box cannot reach any leaf),
// compacts the survivors into the wave's LDS slot by ballot/mbcnt, and then every
// lane evaluates its leaf against the staged points (LDS broadcast reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace la3dm_dev {

constexpr int kWave = 64;
constexpr int kWavesPerWG = 4;

struct BgkArgs {
    const float4 *pts;          // pre-scaled training points (x/ell, y/ell, z/ell, label)
    const uint32_t *train_off;
    const int32_t *nbr;
    const float *blk_center;
    const uint32_t *leaf_off;
    const uint32_t *leaf_key;
    float *alpha;
    float *beta;
    uint8_t *state;
    const float4 *lut;          // voxel LUT, depth-major, w unused
    uint32_t n_test_blk;
    uint32_t tpb_shift;         // log2(tiles per test block)
    uint32_t n_tasks;           // n_test_blk << tpb_shift
    uint32_t flags;
    float sf2, ell, free_thresh, occupied_thresh, var_thresh;
};

// (8^d - 1) / 7 : octal 0o111...1 with d digits
__device__ __forceinline__ uint32_t lut_layer_base(uint32_t depth) {
    return 0x249249u & ((1u << (3u * depth)) - 1u);
}

// Bijective XCD-aware remap (workgroup w runs on XCD w % 8): give each XCD a
// contiguous range of logical workgroups so neighbouring test blocks (which share
// training blocks) hit the same L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t w, uint32_t G) {
    uint32_t q = G >> 3, r = G & 7u, xcd = w & 7u, slot = w >> 3;
    uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + slot;
}

// ---------------------------------------------------------------------------
// sin/cos of t in [0, 2*pi] — shared Cody-Waite reduction by pi/2 and the classic
// single-precision minimax kernels on [-pi/4, pi/4]; <= 1 ulp (tests sweep every
// fp32 t against fp64).  No reference counterpart (the reference calls Eigen's
// cos()/sin()), so FMAs are allowed here.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void sincos_0_2pi(float t, float &s, float &c) {
    const float two_over_pi = 0.636619772f;
    float kf = __builtin_rintf(t * two_over_pi);
    // pi/2 = P1 + P2 + P3, P1 has 8 significant bits so kf*P1 is exact for kf <= 4
    const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.54978995489188216e-8f;
    float y = __builtin_fmaf(-kf, P1, t);
    y = __builtin_fmaf(-kf, P2, y);
    y = __builtin_fmaf(-kf, P3, y);
    float z = y * y;
    // sin(y) = y + y*z*(S1 + z*(S2 + z*(S3 + z*S4)))
    float ps = __builtin_fmaf(z, 2.718311493989822e-6f, -1.9839334836096632e-4f);
    ps = __builtin_fmaf(z, ps, 8.333329385889463e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666666641626524e-1f);
    float sy = __builtin_fmaf(y * z, ps, y);
    // cos(y) = 1 - z/2 + z*z*(C1 + z*(C2 + z*(C3 + z*C4)))
    float pc = __builtin_fmaf(z, -2.6051615464872668e-7f, 2.4760495088926859e-5f);
    pc = __builtin_fmaf(z, pc, -1.3888377661039897e-3f);
    pc = __builtin_fmaf(z, pc, 4.1666638865338612e-2f);
    float hz = 0.5f * z;
    float w = 1.0f - hz;
    float cy = w + (((1.0f - w) - hz) + z * z * pc);
    int k = (int)kf;
    float ss = (k & 1) ? cy : sy;
    float cc = (k & 1) ? sy : cy;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
}

// ---------------------------------------------------------------------------
// Correctly rounded sinf/cosf for t in [0, 2*pi]: reduce by pi/2 and evaluate the
// classic double-precision minimax kernels (error < 2^-57 on [-pi/4, pi/4]) in f64,
// then round once to f32.  The f32 result differs from the exactly rounded one only
// when the true value lies within ~1e-15 relative of a rounding tie (~1e-7 of inputs).
// This is what the oracle's cr_sinf/cr_cosf compute (double libm rounded to float).
// f64 FMA runs at half the f32 rate on gfx950 (78.6 TF); ~20 f64 ops per pair.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void sincos_cr(float t, float &s, float &c) {
    const float kf = __builtin_rintf(t * 0.636619772f);
    const double k = (double)kf;
    const double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI
    double y = __builtin_fma(-k, PIO2_HI, (double)t);   // exact
    y = __builtin_fma(-k, PIO2_LO, y);
    const double z = y * y;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sy = __builtin_fma(z * y, ps, y);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cy = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const float sf = (float)sy, cf = (float)cy;
    const int q = (int)kf;
    const float ss = (q & 1) ? cf : sf;
    const float cc = (q & 1) ? sf : cf;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// correctly rounded x / d for a compile-time constant d (|x| far from the subnormal range):
// q = RN(x * (1/d)); one Newton correction with the exact residual.
__device__ __forceinline__ float div_const(float x, float d, float inv_d) {
    const float q = x * inv_d;
    const float rem = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(rem, inv_d, q);
}

// trig flavours: 0 = correctly rounded (default, parity), 1 = f32 polynomial (<= 1.5 ulp),
// 2 = OCML sinf/cosf
// covSparse elementwise, bgkinference.h:115-125.  r = distance of ell-scaled coords.
template <bool kClamp, int kTrig>
__device__ __forceinline__ float cov_sparse(float r, float sf2) {
    float t = (r * 2.0f) * 3.1415926f;
    float s, c;
    if (kTrig == 0) {
        sincos_cr(t, s, c);
    } else if (kTrig == 1) {
        sincos_0_2pi(t, s, c);
    } else {
        s = sinf(t);
        c = cosf(t);
    }
    float a = ((2.0f + c) * (1.0f - r)) / 3.0f;
    float b = s / (2.0f * 3.1415926f);
    float k = (a + b) * sf2;
    if (kClamp && k < 0.0f) k = 0.0f;
    return k;
}

// x / ell per coordinate (bgkinference.h:114); label rides in w.
__global__ void bgk_prescale_points(const float4 *__restrict__ in, float4 *__restrict__ out, uint32_t n, float ell) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = in[i];
    out[i] = make_float4(p.x / ell, p.y / ell, p.z / ell, p.w);
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Occupancy::update tail: variance / probability / state (bgkoctree_node.cpp:36-43)
__device__ __forceinline__ uint8_t classify(float A, float B, const BgkArgs &a) {
    float s = A + B;
    float var = (A * B) / ((s * s) * (s + 1.0f));
    if (var > a.var_thresh) return 2;
    float p = A / s;
    return p > a.occupied_thresh ? 1 : (p < a.free_thresh ? 0 : 2);
}

// ---------------------------------------------------------------------------
// Variant 1: lane = leaf, wave-uniform skip of points no lane can see.
// ---------------------------------------------------------------------------
template <int kTrig>
__global__ __launch_bounds__(kWavesPerWG *kWave) void bgk_predict_fuse_v1(BgkArgs a) {
    __shared__ float4 s_pts[kWavesPerWG][kWave];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wg = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t task = __builtin_amdgcn_readfirstlane(wg * kWavesPerWG + wv);
    if (task >= a.n_tasks) return;
    const uint32_t blk = task >> a.tpb_shift;
    const uint32_t tile = task & ((1u << a.tpb_shift) - 1u);
    const uint32_t l0 = a.leaf_off[blk] + tile * kWave;
    const uint32_t l1 = a.leaf_off[blk + 1];
    if (l0 >= l1) return;
    const uint32_t nl = min(l1 - l0, (uint32_t)kWave);
    const bool active = (uint32_t)lane < nl;
    const uint32_t li = l0 + (active ? lane : 0);

    // leaf position: LUT[key] + centre (f32), then / ell
    const uint32_t key = a.leaf_key[li];
    const float4 off = a.lut[lut_layer_base(key >> 16) + (key & 0xFFFFu)];
    const float cx = a.blk_center[3 * blk + 0], cy = a.blk_center[3 * blk + 1], cz = a.blk_center[3 * blk + 2];
    const float xs = (off.x + cx) / a.ell, ys = (off.y + cy) / a.ell, zs = (off.z + cz) / a.ell;
    float A = a.alpha[li], B = a.beta[li];

    // tile bounding box in scaled coordinates (inactive lanes replicate lane 0's leaf)
    const float lox = wave_min(xs), loy = wave_min(ys), loz = wave_min(zs);
    const float hix = wave_max(xs), hiy = wave_max(ys), hiz = wave_max(zs);

    bool updated = false;
    const bool ungated = (a.flags & 1u) != 0;
    float4 *sp = s_pts[wv];

    for (int nb = 0; nb < 7; ++nb) {
        const int tb = a.nbr[7 * blk + nb];
        if (tb < 0) continue;
        const uint32_t p0 = a.train_off[tb], p1 = a.train_off[tb + 1];
        float ybar = 0.0f, kbar = 0.0f;
        for (uint32_t base = p0; base < p1; base += kWave) {
            // stage: lane = point
            const uint32_t pi = base + lane;
            bool keep = false;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pi < p1) {
                q = a.pts[pi];
                float ex = fmaxf(fmaxf(lox - q.x, q.x - hix), 0.0f);
                float ey = fmaxf(fmaxf(loy - q.y, q.y - hiy), 0.0f);
                float ez = fmaxf(fmaxf(loz - q.z, q.z - hiz), 0.0f);
                keep = (ex * ex + ey * ey + ez * ez) < 1.00001f;
            }
            const unsigned long long m = __ballot(keep);
            const int n = __popcll(m);
            const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            __builtin_amdgcn_wave_barrier();
            if (keep) sp[slot] = q;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // evaluate: lane = leaf
            for (int j = 0; j < n; ++j) {
                const float4 t = sp[j];
                const float dx = t.x - xs, dy = t.y - ys, dz = t.z - zs;
                const float d2 = dx * dx + (dy * dy + dz * dz);
                if (d2 < 1.0f) {  // k(r) <= 0 for every fp32 r >= 1 (tests/test_oracle.py)
                    const float r = sqrtf(d2);
                    const float k = cov_sparse<true, kTrig>(r, a.sf2);
                    ybar += k * t.w;
                    kbar += k;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (kbar > 0.0f || ungated) {
            A += ybar;
            B += kbar - ybar;
            updated = true;
        }
    }
    if (active) {
        if (updated) {
            a.alpha[li] = A;
            a.beta[li] = B;
            a.state[li] = (uint8_t)(classify(A, B, a) | 0x80u);
        } else {
            a.state[li] = 0;
        }
    }
}

// diagnostics for the parity tests
__global__ void diag_eval_kernel(int op, const float *in, float *out, uint32_t n, float sf2, float ell) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = in[i], y;
    float s, c;
    switch (op) {
    case 0: y = sqrtf(x); break;
    case 1: y = sinf(x); break;
    case 2: y = cosf(x); break;
    case 3: y = cov_sparse<true, 0>(x, sf2); break;
    case 4: y = x / ell; break;
    case 5: y = cov_sparse<false, 0>(x, sf2); break;
    case 6: sincos_0_2pi(x, s, c); y = s; break;
    case 7: sincos_0_2pi(x, s, c); y = c; break;
    case 8: y = cov_sparse<true, 1>(x, sf2); break;
    case 9: sincos_cr(x, s, c); y = s; break;
    case 10: sincos_cr(x, s, c); y = c; break;
    case 11: y = cov_sparse<true, 2>(x, sf2); break;
    default: y = 0.0f;
    }
    out[i] = y;
}

}  // namespace la3dm_dev
