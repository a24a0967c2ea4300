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
#include <stddef.h>
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
    const uint2 *nbr_range;     // [n_test_blk * 7] {first point, count} of each neighbour model (resolved by the prescale launch)
    const uint32_t *blk_desc;   // [(n_test_blk << desc_shift) * 16] flat view of the 7 neighbour ranges for bgk_predict_fuse_r / _t (see bgk_prepare)
    const uint32_t *label_seq;  // == seq when this scan has a label other than 0 / 1 (written by bgk_prepare)
    uint32_t seq;               // number of this scan
    uint32_t n_test_blk;
    uint32_t tpb_shift;         // log2(tiles per test block)
    uint32_t desc_shift;        // 0: one descriptor per test block in blk_desc; tpb_shift: one per TILE (block_depth >= 4, see bgk_prepare)
    uint32_t n_tasks;           // n_test_blk << tpb_shift
    uint32_t flags;
    uint32_t remap;             // 0 contiguous range per XCD, 1 identity, 2 chunks of 8
    uint32_t depth;             // block_depth (leaves of an un-pruned block sit at depth - 1)
    float sf2, ell, free_thresh, occupied_thresh, var_thresh;
    float inv_ell;              // RN(1 / ell); 0 when x / ell has to stay an IEEE division (see div_by_ell)
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
// Correctly rounded sinf/cosf for t in [0, 2*pi]: table + short f64 polynomials (sincos_cr_core below), rounded once
// to f32.  The f32 result differs from the exactly rounded one only when the true value lies within ~1e-15 relative
// of a rounding tie; swept exhaustively against what the oracle's cr_sinf/cr_cosf compute (double libm rounded to
// float).  An f64 FMA costs the same 4 cycles per wave as an f32 one on gfx950.
// ---------------------------------------------------------------------------
// v_fma_f64 with the addend (a constant) in an SGPR pair: the compiler's own choice for a Horner chain on constants is
// v_fmac_f64 + a v_mov_b64 of the constant into the accumulator per step (10 extra 4-cycle instructions per sin/cos
// pair), and 14 constants in VGPR pairs do not fit the 64-VGPR budget of 8 waves per SIMD.
__device__ __forceinline__ double fma64_sc(double a, double b, double c_const) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_const));
    return r;
}
__device__ __forceinline__ double fma64_sb(double a, double b_const, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_const), "v"(c));
    return r;
}
// {sin(q h), cos(q h)} for q = 0 .. 64, h = RN_f32(pi / 32), correctly rounded doubles (tools/gen/sincos_table.py)
static __device__ const unsigned long long kSinCosTab[128][2] = {
#include "sincos_table.inc"
};
typedef double la3dm_v2d __attribute__((ext_vector_type(2)));

// Round 4 form: t = q h + y with q = rint(t / h) in 0 .. 64 and |y| <= h / 2 (+ the drift of q h against q pi / 32,
// < 2e-6); the fp32 FMA t - q h is EXACT (t and q h are multiples of 2^-28 once q >= 1, |y| < 2^-4), so the table angle
// q h needs no low word; sin(t) = sa + (ca sin y + sa (cos y - 1)), cos(t) = ca + (ca (cos y - 1) - sa sin y) with
// the table's {sa, ca} and two short f64 polynomials (|error| < 2e-19 / 1e-21 on the interval).  Against the round-3
// form (reduction by pi / 2, two 5-term chains, swap / sign logic on the quadrant's bits): 20 instead of 33 VALU per
// pair of results, all of the quadrant logic gone.  tools/check/sincos_sweep.c: 0 mismatches against
// (float)sin((double)t), (float)cos((double)t) over every fp32 t in [0, 2 pi] on the CPU (IEEE fma), la3dm_diag_sweep(3)
// the same on the device.  fetch(bits of u, sa, ca) delivers the table entry of q = bits & 127.
// first half: the argument reduction; returns the bits of u (q in the low 7 bits) and y (exact)
__device__ __forceinline__ uint32_t sincos_cr_reduce(float t, float &y32) {
    const float u = __builtin_fmaf(t, 10.1859164f /* 32 / pi */, 12582912.0f);  // low mantissa bits: q
    const float kf = u - 12582912.0f;
    y32 = __builtin_fmaf(-kf, 0x1.921fb6p-4f, t);  // exact
    return __float_as_uint(u);
}
// second half: sin / cos of q h + y from the table entry {sa, ca} of q
__device__ __forceinline__ void sincos_cr_eval(float y32, double sa, double ca, float &s, float &c) {
    const double y = (double)y32;
    const double z = y * y;
    double ps = fma64_sc(z, __builtin_bit_cast(double, 0xbf2a014a5f1de813ull), __builtin_bit_cast(double, 0x3f81111110c194d4ull));
    double pc = fma64_sc(z, __builtin_bit_cast(double, 0x3efa015b846c26deull), __builtin_bit_cast(double, 0xbf56c16c1681d56aull));
    const double zy = z * y;
    ps = fma64_sc(z, ps, __builtin_bit_cast(double, 0xbfc555555555552aull));
    pc = fma64_sc(z, pc, __builtin_bit_cast(double, 0x3fa5555555555544ull));
    const double sy = __builtin_fma(zy, ps, y);      // sin y
    pc = __builtin_fma(z, pc, -0.5);
    const double cm1 = z * pc;                       // cos y - 1
    const double sd = __builtin_fma(ca, sy, __builtin_fma(sa, cm1, sa));
    const double cd = __builtin_fma(-sa, sy, __builtin_fma(ca, cm1, ca));
    s = (float)sd;
    c = (float)cd;
}
template <class Fetch>
__device__ __forceinline__ void sincos_cr_core(float t, float &s, float &c, Fetch fetch) {
    float y32;
    const uint32_t ub = sincos_cr_reduce(t, y32);
    double sa, ca;
    fetch(ub, sa, ca);
    sincos_cr_eval(y32, sa, ca, s, c);
}
__device__ __forceinline__ la3dm_v2d sincos_cr_entry(uint32_t ub) {  // the table entry of q = ub & 127 (a 16-byte gather)
    return *reinterpret_cast<const la3dm_v2d *>(reinterpret_cast<const char *>(kSinCosTab) + ((ub & 127u) << 4));
}
// the table read from memory (a 16-byte gather that stays in the vector L1): any t in [0, 2 pi], NaN in -> NaN out
__device__ __forceinline__ void sincos_cr(float t, float &s, float &c) {
    sincos_cr_core(t, s, c, [](uint32_t ub, double &sa, double &ca) {
        const la3dm_v2d e = sincos_cr_entry(ub);
        sa = e.x;
        ca = e.y;
    });
}
// (Rounds 4 and 5 read the entry of a ring entry's t without the mask — `(ub << 4)` against a table pointer biased by 0xB4000000,
//  one instruction less per batch — which is only in bounds while t is finite: true by construction of the ring, but a wild read
//  the day that invariant breaks.  Round 6: the masked form everywhere; the 128-entry table cannot be left.)
// (the table across the wave's lanes by ds_bpermute_b32 was measured slower: 24.5 cycles per SIMD per crossbar
// instruction, four per batch, against one L1 gather)

// ---------------------------------------------------------------------------
// fast_trig 3 — the LIKELY REFERENCE trig: Eigen 3.3.7's psin / pcos<Packet4f> (arch/SSE/MathFunctions.h,
// the Cephes sinf / cosf: reduction by pi / 4 in three Cody-Waite steps, two degree-3 / degree-2-in-z polynomials), one
// lane, with every multiply-add as a rounded multiply followed by a rounded add (an x86-64 baseline build has no FMA) —
// what la3dm's `cos(...)` / `sin(...)` array expressions (bgkinference.h:115-116) most plausibly execute in a ROS Noetic
// build.  The same operations in the same order as the restatement's orc_eigen337::psin / pcos (oracle/la3dm_oracle.cpp,
// oracle.set_modes(1, 0)): bit-identical results (tests/test_bgk_gpu.py).  Outside the parity contract of the default
// (fast_trig 0 = correctly rounded): it exists so that a user who holds the real reference can compare against it.  The
// hit threshold kHitTBits stays valid: with this trig the kernel is exactly 0 for every d2 >= 0x3f779dec < kHitTBits (swept
// over every fp32 d2 of [0.9, 1.5) with the restatement's arithmetic).  This translation unit is built with
// -ffp-contract=off: `a * b + c` below is two instructions.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void sincos_eigen337(float t, float &s, float &c) {
    const uint32_t sign_in = __float_as_uint(t) & 0x80000000u;
    float x = __builtin_fabsf(t);
    float y = x * 1.27323954473516f;          // cephes_FOPI = 4 / pi
    int32_t j = (int32_t)y;                   // _mm_cvttps_epi32
    j = (j + 1) & ~1;
    y = (float)j;
    x = x + y * -0.78515625f;
    x = x + y * -2.4187564849853515625e-4f;
    x = x + y * -3.77489497744594108e-8f;
    const float z = x * x;
    float pc = 2.443315711809948E-005f;        // the cosine polynomial
    pc = pc * z + -1.388731625493765E-003f;
    pc = pc * z + 4.166664568298827E-002f;
    pc = pc * z;
    pc = pc * z;
    pc = pc - z * 0.5f;
    pc = pc + 1.0f;
    float ps = -1.9515295891E-4f;              // the sine polynomial
    ps = ps * z + 8.3321608736E-3f;
    ps = ps * z + -1.6666654611E-1f;
    ps = ps * z;
    ps = ps * x;
    ps = ps + x;
    const bool use_sin = (j & 2) == 0;         // psin: the sine polynomial; pcos (j - 2): the other one
    const uint32_t sign_s = sign_in ^ (((uint32_t)(j & 4)) << 29);
    const uint32_t sign_c = ((uint32_t)(~(j - 2) & 4)) << 29;
    s = __uint_as_float(__float_as_uint(use_sin ? ps : pc) ^ sign_s);
    c = __uint_as_float(__float_as_uint(use_sin ? pc : ps) ^ sign_c);
}

// correctly rounded x / d for a compile-time constant d (|x| far from the subnormal range):
// q = RN(x * (1/d)); one Newton correction with the exact residual.
__device__ __forceinline__ float div_const(float x, float d, float inv_d) {
    const float q = x * inv_d;
    const float rem = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(rem, inv_d, q);
}
// two correction steps (the refinement core of the hardware division sequence with the
// reciprocal folded into a constant)
__device__ __forceinline__ float div_const2(float x, float d, float inv_d) {
    const float q0 = x * inv_d;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, x), inv_d, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, x), inv_d, q1);
}

// x / ell for the per-map constant ell (bgkinference.h:114 divides every coordinate): with y = RN(1 / ell),
// q = RN(x * y), r = x - q * ell (exact, one FMA), RN(q + r * y) is the correctly rounded quotient whenever the
// significand of ell is not all ones (Markstein's theorem) and nothing under- or overflows — the host only passes
// inv_ell != 0 for such an ell, map coordinates are far from both ends of the exponent range, and
// la3dm_diag_sweep(what = 7) checks every fp32 x of the coordinate range against the IEEE division on the device.
__device__ __forceinline__ float div_by_ell(float x, float ell, float inv_ell) {
    return inv_ell != 0.0f ? div_const(x, ell, inv_ell) : x / ell;
}

// trig flavours: 0 = correctly rounded (default, parity), 1 = f32 polynomial (<= 1.5 ulp),
// 2 = OCML sinf/cosf, 3 = Eigen 3.3.7's psin / pcos without FMA (the likely reference build, sincos_eigen337)
// covSparse elementwise, bgkinference.h:115-125.  r = distance of ell-scaled coords.
template <bool kClamp, int kTrig>
__device__ __forceinline__ float cov_sparse(float r, float sf2) {
    float t = (r * 2.0f) * 3.1415926f;
    float s, c;
    if (kTrig == 0) {
        sincos_cr(t, s, c);
    } else if (kTrig == 1) {
        sincos_0_2pi(t, s, c);
    } else if (kTrig == 3) {
        sincos_eigen337(t, s, c);
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

// ---------------------------------------------------------------------------
struct BgkDescArgs {   // per-tile descriptors (shift != 0): what bgk_prepare needs beside the block-wide inputs
    uint32_t shift = 0;            // 0: one descriptor per block; log2(tiles per block): one per tile
    uint32_t depth = 0;            // block_depth
    const uint32_t *leaf_off = nullptr;
};
// Same launch shape, second job: resolve nbr[t][b] -> {train_off[nb], count} once per scan, so that
// the predict kernel's prologue needs one dependent memory round trip less per tile.
// Third job (blk_desc != nullptr): the same seven ranges of a test block as ONE flat index space [0, M) — the
// concatenation of the neighbours' points in ExtendedBlock order — for the kernel that stages 64 points per trip whatever
// neighbour they belong to: words 0-6 = first point of neighbour b minus the flat index where b starts (flat index +
// this = point index, modulo 2^32), words 8-14 = flat index where neighbour b ends (word 14 = M), word 7 = bit b set
// when neighbour b has a trained model, word 15 = 0.
__global__ void bgk_prepare(const float4 *__restrict__ in, float4 *__restrict__ out, uint32_t n, float ell,
                            const int32_t *__restrict__ nbr, const uint32_t *__restrict__ train_off,
                            uint2 *__restrict__ nbr_range, uint32_t n_nbr, uint32_t *__restrict__ blk_desc,
                            uint32_t *__restrict__ label_seq, uint32_t seq, BgkDescArgs dsc = BgkDescArgs()) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk_desc && i < ((n_nbr / 7u) << dsc.shift)) {
        // dsc.shift != 0 (block_depth >= 4, round 6): one descriptor per TILE.  A tile of a FULL block is an aligned 4 x 4 x 4 cube of
        // voxels at a known place in its block; a face neighbour the cube does not touch is at least four voxel edges away from
        // the cube's box, 4.5 from its nearest leaf centre — beyond the kernel's support when ell <= 4 * resolution (the host checks
        // that) — so its points are left out of the tile's flat range: at depth 4 every tile keeps its own block and three of the six
        // face neighbours, where the block-wide descriptor made all eight tiles stage and cull all seven (the points a tile drops
        // this way can reach none of its leaves: same pairs, same sums).  Tiles of pruned blocks keep all seven ranges.
        const uint32_t blk = i >> dsc.shift, tile = i & ((1u << dsc.shift) - 1u);
        uint32_t keep = 0x7Fu;
        if (dsc.shift) {
            const uint32_t n_fine = 1u << (3u * (dsc.depth - 1u));
            if (dsc.leaf_off[blk + 1] - dsc.leaf_off[blk] == n_fine) {
                const uint32_t q = (n_fine >> 6) - 1u - tile;   // the cube's index on the (depth - 3)-level octree of cubes: 3 bits per level, child = 4 x + 2 y + z
                uint32_t px = 0, py = 0, pz = 0;
                const uint32_t levels = dsc.depth - 3u;
                for (uint32_t l = 0; l < levels; ++l) {
                    const uint32_t c = (q >> (3u * l)) & 7u;
                    px |= ((c >> 2) & 1u) << l;
                    py |= ((c >> 1) & 1u) << l;
                    pz |= (c & 1u) << l;
                }
                const uint32_t last = (1u << levels) - 1u;
                // ExtendedBlock order: self, +x, -x, +y, -y, +z, -z
                keep = 1u | (px == last ? 2u : 0u) | (px == 0u ? 4u : 0u) | (py == last ? 8u : 0u) | (py == 0u ? 16u : 0u) |
                       (pz == last ? 32u : 0u) | (pz == 0u ? 64u : 0u);
            }
        }
        uint32_t d[16];
        uint32_t pre = 0, trained = 0;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const int tb = nbr[7 * blk + b];
            uint32_t first = 0, cnt = 0;
            if (tb >= 0) {
                first = train_off[tb];
                cnt = train_off[tb + 1] - first;
            }
            trained |= (cnt ? 1u : 0u) << b;
            if (!((keep >> b) & 1u)) cnt = 0;
            d[b] = first - pre;
            pre += cnt;
            d[8 + b] = pre;
        }
        d[7] = trained;
        d[15] = 0;
        uint4 *o = (uint4 *)(blk_desc + 16 * (size_t)i);
        o[0] = make_uint4(d[0], d[1], d[2], d[3]);
        o[1] = make_uint4(d[4], d[5], d[6], d[7]);
        o[2] = make_uint4(d[8], d[9], d[10], d[11]);
        o[3] = make_uint4(d[12], d[13], d[14], d[15]);
    }
    if (i < n) {
        const float4 p = in[i];
        out[i] = make_float4(p.x / ell, p.y / ell, p.z / ell, p.w);
        if (label_seq && !(p.w == 0.0f || p.w == 1.0f)) *label_seq = seq;  // (same value from every writer)
    }
    if (nbr_range && i < n_nbr) {   // (the ordered kernel's view of the neighbour ranges: nullptr when another kernel runs)
        const int tb = nbr[i];
        uint2 r = make_uint2(0u, 0u);
        if (tb >= 0) {
            r.x = train_off[tb];
            r.y = train_off[tb + 1] - r.x;
        }
        nbr_range[i] = r;
    }
}

__device__ __forceinline__ uint8_t classify(float A, float B, const BgkArgs &a) {
    float s = A + B;
    float var = (A * B) / ((s * s) * (s + 1.0f));
    if (var > a.var_thresh) return 2;
    float p = A / s;
    return p > a.occupied_thresh ? 1 : (p < a.free_thresh ? 0 : 2);
}

// The same state from approximate quotients (v_rcp_f32 + one product: within 2 ulp of the IEEE quotients) whenever no
// lane's quotient lies within 2^-18 (relative) of a threshold it is compared with — there the comparison of the
// approximation decides like the exact one; otherwise (rare) the whole wave takes classify().  Two IEEE divisions
// cost 22 VALU instructions per leaf tile, the approximations 4.  NaN / inf operands behave alike on both paths
// (every comparison false -> 2; a product over an infinite denominator is 0 on both).
__device__ __forceinline__ uint8_t classify_fast(float A, float B, const BgkArgs &a) {
    const float s = A + B, den = (s * s) * (s + 1.0f);
    const float v = (A * B) * __builtin_amdgcn_rcpf(den);
    const float p = A * __builtin_amdgcn_rcpf(s);
    // (v_rcp_f32 treats a denormal operand as zero: denominators that small — priors below 1e-19 — take the divisions too)
    const bool near = __builtin_fabsf(v - a.var_thresh) <= __builtin_fabsf(a.var_thresh) * 0x1p-18f ||
                      __builtin_fabsf(p - a.occupied_thresh) <= __builtin_fabsf(a.occupied_thresh) * 0x1p-18f ||
                      __builtin_fabsf(p - a.free_thresh) <= __builtin_fabsf(a.free_thresh) * 0x1p-18f || !(den > 0x1p-100f);
    if (__ballot(near) != 0ull) return classify(A, B, a);
    if (v > a.var_thresh) return 2;
    return p > a.occupied_thresh ? 1 : (p < a.free_thresh ? 0 : 2);
}

// ---------------------------------------------------------------------------
// Wave64 reductions on the DPP network, lean square root, kernel with constant reciprocals
// ---------------------------------------------------------------------------
template <int kCtrl, int kRowMask = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), kCtrl, kRowMask, 0xF, false));
}
// min / max over the 64 lanes, result wave-uniform
__device__ __forceinline__ float wave_min_dpp(float v) {
    v = fminf(v, dpp_f<0xB1>(v));         // quad_perm [1,0,3,2]
    v = fminf(v, dpp_f<0x4E>(v));         // quad_perm [2,3,0,1]
    v = fminf(v, dpp_f<0x141>(v));        // row_half_mirror
    v = fminf(v, dpp_f<0x140>(v));        // row_mirror
    v = fminf(v, dpp_f<0x142, 0xA>(v));   // row_bcast15 into rows 1,3
    v = fminf(v, dpp_f<0x143, 0xC>(v));   // row_bcast31 into rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    v = fmaxf(v, dpp_f<0x142, 0xA>(v));
    v = fmaxf(v, dpp_f<0x143, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// correctly rounded sqrt for x in {0} U [2^-100, 2^100]: hardware estimate (<= 1 ulp) plus the
// standard one-ulp residual fix-up (no denormal pre-scaling: d2 is 0 or >= ~1e-15 here).
__device__ __forceinline__ float sqrt_cr(float x) {
    // s = hardware estimate (within 1 ulp); sm / sp its neighbours.  RN(sqrt x) is sm when sm * s >= x, sp when
    // sp * s < x, else s: with em = sm * s - x and ep = sp * s - x (one FMA each; an exact zero comes out as +0) that is
    // bits(sm) + [em < 0] + [ep < 0], the two flags being the sign bits — no compare / select pairs (each needs a wait
    // state on gfx940).  x = 0: s = 0, sm is a NaN pattern with the sign set (em = NaN keeps it), ep = +0: result 0.
    const float s = __builtin_amdgcn_sqrtf(x);
    const uint32_t smb = __float_as_uint(s) - 1u;
    const float sm = __uint_as_float(smb), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float em = __builtin_fmaf(sm, s, -x), ep = __builtin_fmaf(sp, s, -x);
    return __uint_as_float(smb + (__float_as_uint(em) >> 31) + (__float_as_uint(ep) >> 31));
}

// covSparse elementwise (bgkinference.h:115-125) with the two constant divisions done by
// div_const; bit-identical to cov_sparse<true, kTrig> (tests sweep the divisions exhaustively).
// the kernel formula from r and sin / cos of 2 pi r (bgkinference.h:115-125), the two constant divisions by div_const
template <bool kClamp = true>
__device__ __forceinline__ float cov_sparse_formula(float r, float s, float c, float sf2) {
    const float a = div_const((2.0f + c) * (1.0f - r), 3.0f, 0.333333343f);
    const float b = div_const(s, 2.0f * 3.1415926f, 0.159154952f);
    float k = (a + b) * sf2;
    if (kClamp) k = fmaxf(k, 0.0f);  // k is never NaN here; (-0 -> +0 adds the same to every sum)
    return k;
}
template <int kTrig, bool kClamp = true, bool kFinite = false>
__device__ __forceinline__ float cov_sparse_fast(float r, float sf2) {
    const float t = (r * 2.0f) * 3.1415926f;
    float s, c;
    if (kTrig == 0) sincos_cr(t, s, c);   // (kFinite: kept in the signature; since round 6 every gather of the table is masked)
    else if (kTrig == 1) sincos_0_2pi(t, s, c);
    else if (kTrig == 3) sincos_eigen337(t, s, c);
    else { s = sinf(t); c = cosf(t); }
    return cov_sparse_formula<kClamp>(r, s, c, sf2);
}

// ---------------------------------------------------------------------------
// bgk_predict_fuse_v5 (the kernel; variants 1-4 and 6 of the measurement history in DESIGN.md are gone):
// three tight phases per candidate round.
//   B  test + push, branch free: four candidates per trip (hit <=> d2 < kHitT); hits of candidate j are written
//      to ring[tail + rank] (rank = mbcnt of the hit mask), misses to a per-lane scratch
//      slot; every lane shifts its own hit bit into a 64-candidate history.
//   C  dense evaluation: lane i evaluates ring[p + i] and writes {k, k*y} back in place.
//   D  ordered fuse: candidates are replayed in order; the hit mask is rebuilt from the
//      history by a ballot, the leaf lane reads its own {k, k*y} at ring[off + rank]
//      (conflict-free, contiguous) and adds them to its private (ybar, kbar) — the
//      reference's summation order, so results stay bit-identical to the CPU restatement.
//      A new neighbour slot is a precomputed bit per candidate: the flush into
//      (alpha, beta) is a uniform branch in ExtendedBlock order.
// LDS per wave: 64 candidates (1 KB) + 448-pair ring and scratch (4 KB).
// ---------------------------------------------------------------------------
constexpr int kCand5 = 64;
constexpr int kRing5 = 432;  // with the 64 scratch slots and the zero entry the wave's LDS stays within 5 120 B (8 waves per SIMD)
// 0.96778184f: the smallest fp32 T such that k(sqrt(d2)) == 0 (after the < 0 clamp) for EVERY fp32 d2 in [T, 1) —
// and k <= 0 for every r >= 1 —, so a pair with d2 >= T adds +0 to both sums and can be dropped at the distance
// test (4.8 % of the pairs inside the unit ball).  The sign of (a + b) does not depend on sf2 > 0.  Checked against the
// oracle's kernel over all 1.68 M fp32 values of [0.9, 1) (tests/test_oracle.py) and against the device arithmetic by
// la3dm_diag_sweep(what = 8) (tests/test_bgk_gpu.py).  Exact for the correctly rounded trig (fast_trig = 0, the parity
// configuration) only: with fast_trig 1 / 2 (f32 polynomial / OCML, <= 1.5 ulp, outside the parity contract) a pair with d2
// just below 1 may have a kernel value of a few 1e-8 that this threshold drops.
constexpr uint32_t kHitTBits = 0x3f77c08du;

// B, four candidates per trip, PACKED fp32: a VALU instruction occupies the SIMD for ~4 cycles whatever it is
// (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.0 in this kernel), so v_pk_add / v_pk_mul_f32 — two candidates per
// instruction — halve the cost of the distance: 8 packed instructions per PAIR of candidates
//   (dx, dy, dz)(A,B) = (X, Y, Z)(A,B) - (xs, ys, zs);  d2 = dx*dx + (dy*dy + dz*dz)   (bgkinference.h:88-93, same association)
// against 8 per candidate.  The leaf's coordinates are broadcast to both halves by op_sel; the two pairs of a trip
// alternate so that no packed instruction reads the result of the one before it (one wait state).  Registers are fixed
// (v40-v63): the halves of a packed result feed v_cmp and ds_write separately.
//   v[52:55] X of the four candidates, v[56:59] Y, v[60:63] Z;  pair 0 -> d2 in v40, v41;  pair 1 -> v46, v47
// Push of one candidate (5 VALU + 4 SALU + 1 LDS): hit mask in vcc; the scalar popcount and the next entry word sit
// between the v_cmp and the first VALU reader of vcc (gfx940: two wait states); hit lanes write {d2, entry word} at
// ring[tail + rank] under exec = hit mask.
#define LA3DM_RP_LOAD                                                                     \
    "ds_read_b128 v[52:55], %[ca]\n"                                                     \
    "ds_read_b128 v[56:59], %[ca] offset:272\n"                                          \
    "ds_read_b128 v[60:63], %[ca] offset:544\n"                                          \
    "s_waitcnt lgkmcnt(0)\n"
#define LA3DM_RP_SUB(DX, DY, DZ, X, Y, Z)                                                 \
    "v_pk_add_f32 " DX ", " X ", %[xy] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"      \
    "v_pk_add_f32 " DY ", " Y ", %[xy] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n" \
    "v_pk_add_f32 " DZ ", " Z ", %[zz] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
#define LA3DM_RP_SQ(DX, DY, DZ)                                                           \
    "v_pk_mul_f32 " DX ", " DX ", " DX "\n"                                              \
    "v_pk_mul_f32 " DY ", " DY ", " DY "\n"                                              \
    "v_pk_mul_f32 " DZ ", " DZ ", " DZ "\n"
#define LA3DM_RP_PUSH(D, I, IN)                             \
    "v_cmp_gt_f32 vcc, %[T], " D "\n"                       \
    "s_bcnt1_i32_b64 %[st], vcc\n"                          \
    "v_add_u32 " IN ", 4, " I "\n"                          \
    "v_mbcnt_lo_u32_b32 %[tr], vcc_lo, 0\n"                 \
    "v_mbcnt_hi_u32_b32 %[tr], vcc_hi, %[tr]\n"             \
    "v_lshl_add_u32 %[tr], %[tr], 3, %[tail]\n"             \
    "s_mov_b64 exec, vcc\n"                                 \
    "ds_write2_b32 %[tr], " D ", " I " offset1:1\n"         \
    "s_mov_b64 exec, -1\n"                                  \
    "s_lshl3_add_u32 %[tail], %[st], %[tail]\n"

// the ordered kernel's push: the same, + the lane's hit bit shifted into its history word (h = 2 h + hit: v_addc with the
// mask as carry, after the write — it overwrites vcc) and a plain candidate index as the entry word
#define LA3DM_RP_PUSH_H(D, I, IN)                           \
    "v_cmp_gt_f32 vcc, %[T], " D "\n"                       \
    "s_bcnt1_i32_b64 %[st], vcc\n"                          \
    "v_add_u32 " IN ", 1, " I "\n"                          \
    "v_mbcnt_lo_u32_b32 %[tr], vcc_lo, 0\n"                 \
    "v_mbcnt_hi_u32_b32 %[tr], vcc_hi, %[tr]\n"             \
    "v_lshl_add_u32 %[tr], %[tr], 3, %[tail]\n"             \
    "s_mov_b64 exec, vcc\n"                                 \
    "ds_write2_b32 %[tr], " D ", " I " offset1:1\n"         \
    "s_mov_b64 exec, -1\n"                                  \
    "v_addc_co_u32 %[h], vcc, %[h], %[h], vcc\n"            \
    "s_lshl3_add_u32 %[tail], %[st], %[tail]\n"
typedef float la3dm_v2f __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(16))) WaveLds5 {
    float cx[kCand5 + 4], cy[kCand5 + 4], cz[kCand5 + 4], cw[kCand5 + 4];   // x/ell, y/ell, z/ell, label: one array per component
    uint2 ring[kRing5 + kWave + 1];  // {d2, candidate} -> {k, k*y}; (64 spare slots) and one {0, 0} entry
};

// The two loops that run once per staged candidate are written in gfx950 assembly: the compiler's versions carried
// ~9 scalar and ~16 vector instructions per candidate in B (exec-mask juggling around the compaction, 64-bit history
// shifts) and 6 + 13 in D; scalar instructions cost an issue slot like vector ones on this chip
// (profiles/r02/valu_issue.txt), so the instruction count is the time.  Here B is 15 VALU + 2 SALU + 1 LDS and D's
// gather 6 VALU + 2 SALU + 1 LDS per candidate (the ordered adds and the per-neighbour flush stay in C++).  No DPP / lane-select / transcendental / packed instruction is used, so none
// of the gfx9 data hazards applies except gfx940's "VALU writes an SGPR / VCC -> a VALU reads it as an operand, carry or
// mask: 2 wait states" — every v_cmp below is followed by two instructions that do not read its mask.  LDS returns
// in order, so lgkmcnt(n) counts this block's own reads from the back of the queue.
//
// B, one candidate {X, Y, Z} (training point / ell), lane = leaf at (xs, ys, zs):
//   d2 = dx*dx + (dy*dy + dz*dz) in the reference's association (bgkinference.h:88-93); hit <=> d2 < T;
//   hit lanes write {d2, candidate index} to ring[tail + rank] (rank = mbcnt of the hit mask), the others to their
//   scratch slot; the lane's hit bit is shifted into its history word (h = 2 h + hit: v_addc with the mask as carry).
#ifndef LA3DM_ASM_B
#define LA3DM_ASM_B 1
#endif
#ifndef LA3DM_ASM_D
#define LA3DM_ASM_D 1
#endif
#define LA3DM_B_HEAD(A, X, Y)                               \
    "v_sub_f32 " A ", " X ", %[xs]\n"                       \
    "v_sub_f32 %[tb], " Y ", %[ys]\n"
#define LA3DM_B_TAIL(A, Z)                                  \
    "v_sub_f32 %[tc], " Z ", %[zs]\n"                       \
    "v_mul_f32 " A ", " A ", " A "\n"                       \
    "v_mul_f32 %[tb], %[tb], %[tb]\n"                       \
    "v_mul_f32 %[tc], %[tc], %[tc]\n"                       \
    "v_add_f32 %[tb], %[tb], %[tc]\n"                       \
    "v_add_f32 " A ", " A ", %[tb]\n"                       \
    "v_cmp_gt_f32 vcc, %[T], " A "\n"
#define LA3DM_B_PUSH(A)                                     \
    "v_mbcnt_lo_u32_b32 %[tr], vcc_lo, 0\n"                 \
    "v_mbcnt_hi_u32_b32 %[tr], vcc_hi, %[tr]\n"             \
    "v_lshl_add_u32 %[tr], %[tr], 3, %[tail]\n"             \
    "v_cndmask_b32 %[tr], %[scr], %[tr], vcc\n"             \
    "ds_write2_b32 %[tr], " A ", %[idx] offset1:1\n"        \
    "s_bcnt1_i32_b64 %[st], vcc\n"                          \
    "v_add_u32 %[idx], 1, %[idx]\n"                         \
    "v_addc_co_u32 %[h], vcc, %[h], %[h], vcc\n"            \
    "s_lshl3_add_u32 %[tail], %[st], %[tail]\n"
// D, one candidate: the lane's hit bit leaves the history word at the top (oldest first); hit lanes read their
// {k, k*y} at ring[roff + rank], the others the zero entry; E = the 64-bit register pair that receives it.
#define LA3DM_D_CAND(E)                                     \
    "v_cmp_gt_i32 vcc, 0, %[h]\n"                           \
    "v_add_u32 %[h], %[h], %[h]\n"                          \
    "s_nop 0\n"                                             \
    "v_mbcnt_lo_u32_b32 %[tr], vcc_lo, 0\n"                 \
    "v_mbcnt_hi_u32_b32 %[tr], vcc_hi, %[tr]\n"             \
    "v_lshl_add_u32 %[tr], %[tr], 3, %[roff]\n"             \
    "v_cndmask_b32 %[tr], %[zad], %[tr], vcc\n"             \
    "ds_read_b64 " E ", %[tr]\n"                            \
    "s_bcnt1_i32_b64 %[st], vcc\n"                          \
    "s_lshl3_add_u32 %[roff], %[st], %[roff]\n"

template <int kTrig, int kWaves>
__global__ __launch_bounds__(kWaves *kWave) __attribute__((amdgpu_waves_per_eu(kWaves == 1 ? 8 : 4, 8))) void bgk_predict_fuse_v5(BgkArgs a) {
    __shared__ WaveLds5 s_lds[kWaves];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    uint32_t wg = blockIdx.x;
    if (a.remap == 0) wg = xcd_remap(wg, gridDim.x);
    else if (a.remap == 2) {
        const uint32_t G8 = gridDim.x & ~63u;
        if (wg < G8) wg = (wg & ~63u) | ((wg & 7u) << 3) | ((wg >> 3) & 7u);
    }
    const uint32_t task = __builtin_amdgcn_readfirstlane(wg * kWaves + wv);
    if (task >= a.n_tasks) return;
    const uint32_t blk = task >> a.tpb_shift;
    const uint32_t tile = task & ((1u << a.tpb_shift) - 1u);
    const uint32_t l0 = a.leaf_off[blk] + tile * kWave;
    const uint32_t l1 = a.leaf_off[blk + 1];
    if (l0 >= l1) return;
    WaveLds5 &L = s_lds[wv];
    const uint32_t nl = min(l1 - l0, (uint32_t)kWave);
    const bool active = (uint32_t)lane < nl;
    const uint32_t li = l0 + (active ? lane : 0);

    int tb[7];
    uint32_t p0[7], cnt[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        const uint2 r = a.nbr_range[7 * blk + b];
        p0[b] = r.x;
        cnt[b] = r.y;
        tb[b] = r.y ? 0 : -1;  // only "has a trained model" matters below (a model has >= 1 point)
    }
    float4 q[7];
#pragma unroll
    for (int b = 0; b < 7; ++b)
        q[b] = ((uint32_t)lane < cnt[b]) ? a.pts[p0[b] + lane] : make_float4(0.f, 0.f, 0.f, 0.f);

    const uint32_t key = a.leaf_key[li];
    const float4 off4 = a.lut[lut_layer_base(key >> 16) + (key & 0xFFFFu)];
    const float cx = a.blk_center[3 * blk + 0], cy = a.blk_center[3 * blk + 1], cz = a.blk_center[3 * blk + 2];
    const float xs0 = div_by_ell(off4.x + cx, a.ell, a.inv_ell), ys0 = div_by_ell(off4.y + cy, a.ell, a.inv_ell),
                zs0 = div_by_ell(off4.z + cz, a.ell, a.inv_ell);
    float A = a.alpha[li], B = a.beta[li];

    // bounding box of the tile's leaf centres (only culls: any box that contains them is valid).  64 finest-depth
    // leaves with the indices 64 c + 63 ... 64 c (LeafIterator order is descending, bgkoctree.h:101-135) are one aligned
    // 4x4x4 cube: lane 0 holds its (+x,+y,+z) corner and lane 63 the (-x,-y,-z) one.  Anything else — a short tile,
    // coarse leaves, or 64 fine leaves that straddle two cubes of a partly pruned block — takes the reduction.
    float lox, loy, loz, hix, hiy, hiz;
    const uint32_t key_first = __builtin_amdgcn_readlane(key, 0), key_last = __builtin_amdgcn_readlane(key, 63);
    if (nl == (uint32_t)kWave && (key_last & 63u) == 0u && key_first == key_last + 63u &&
        __ballot((key >> 16) + 1u != a.depth) == 0ull) {
        lox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs0), 63));
        loy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ys0), 63));
        loz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zs0), 63));
        hix = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs0), 0));
        hiy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ys0), 0));
        hiz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zs0), 0));
    } else {
        lox = wave_min_dpp(xs0), loy = wave_min_dpp(ys0), loz = wave_min_dpp(zs0);
        hix = wave_max_dpp(xs0), hiy = wave_max_dpp(ys0), hiz = wave_max_dpp(zs0);
    }
    const float xs = active ? xs0 : __builtin_nanf(""), ys = ys0, zs = zs0;  // NaN never matches
    la3dm_v2f xy, zz;   // the leaf's coordinates as packed operands: (xs, ys) and (zs, -)
    xy.x = xs, xy.y = ys, zz.x = zs, zz.y = 0.0f;
    (void)xs; (void)ys; (void)zs;

    const bool ungated = (a.flags & 1u) != 0;
    bool updated = false;
    float kbar = 0.0f, ybar = 0.0f;
    uint32_t ncand = 0;
    unsigned long long nbstart = 0;  // bit s: candidate slot s is the first of a new neighbour
    int last_nb = -1;

    auto flush_nb = [&]() {  // Occupancy::update, bgkoctree_node.cpp:31-35
        if (kbar > 0.0f || ungated) {
            A += ybar;
            B += kbar - ybar;
            updated = true;
        }
        kbar = 0.0f;
        ybar = 0.0f;
    };

    auto stage = [&](const float4 &p, bool valid, int b) {
        bool keep = false;
        if (valid) {
            const float ex = fmaxf(fmaxf(lox - p.x, p.x - hix), 0.0f);
            const float ey = fmaxf(fmaxf(loy - p.y, p.y - hiy), 0.0f);
            const float ez = fmaxf(fmaxf(loz - p.z, p.z - hiz), 0.0f);
            // the distance to the box of the leaf centres bounds every leaf's distance from below: at or beyond the hit
            // threshold (plus slack for the different rounding of this sum) no leaf can hit
            keep = (ex * ex + ey * ey + ez * ez) < 0.96780f;
        }
        const unsigned long long m = __ballot(keep);
        if (b != last_nb && m != 0ull) {
            nbstart |= 1ull << ncand;
            last_nb = b;
        }
        const uint32_t slot = ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (keep) {
            L.cx[slot] = p.x;
            L.cy[slot] = p.y;
            L.cz[slot] = p.z;
            L.cw[slot] = p.w;
        }
        ncand += (uint32_t)__popcll(m);
    };

    // A: first chunks in ExtendedBlock order while they fit and no neighbour needs a 2nd chunk
    uint32_t it_b = 7, it_base = 0;
    {
        bool open = true;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            if (!open || cnt[b] == 0) continue;
            if (ncand + min(cnt[b], (uint32_t)kWave) <= (uint32_t)kCand5) {
                stage(q[b], (uint32_t)lane < cnt[b], b);
                if (cnt[b] > (uint32_t)kWave) {
                    open = false;
                    it_b = b;
                    it_base = kWave;
                }
            } else {
                open = false;
                it_b = b;
                it_base = 0;
            }
        }
    }

    const float hit_t = __uint_as_float(kHitTBits);
    // LDS byte addresses (the low 32 bits of a generic pointer into LDS are the LDS offset)
    const uint32_t ring_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&L.ring[0]);
    const uint32_t zero_addr = ring_base + 8u * (uint32_t)(kRing5 + kWave);
    if (lane == 0) L.ring[kRing5 + kWave] = make_uint2(0u, 0u);
    bool more = true;
    while (more) {
        // pad the list to a multiple of four with points no leaf can reach
        if (lane < 4) {
            L.cx[ncand + lane] = 3.0e18f;
            L.cy[ncand + lane] = 3.0e18f;
            L.cz[ncand + lane] = 3.0e18f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t ngroup = (a.flags & 0x200u) ? 0u : (ncand + 3u) >> 2;  // 0x200: profiling ablation
        uint32_t g = 0;
        while (g < ngroup) {
            // ---- B: test + push (assembly, four candidates per trip) ----
            const uint32_t g0 = g;
            uint32_t tailb = ring_base;          // LDS byte address of ring[tail]
            uint32_t hA = 0, hB = 0;             // hit history: candidates 0-31 of this round in hA, 32-63 in hB
            uint32_t idx = 4 * g;                // candidate index, uniform, in a VGPR for the ring entry
            const uint32_t cand_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&L.cx[0]);
            auto b_trip = [&](uint32_t &hw) {
#if LA3DM_ASM_B
                // packed fp32, two candidates per instruction (LA3DM_RP_* above): 4 + 6 VALU per candidate instead of 8 + 7
                uint32_t i1, st;
                float tr;
                const uint32_t ca = cand_base + 16u * g;
                asm volatile(LA3DM_RP_LOAD
                             LA3DM_RP_SUB("v[40:41]", "v[42:43]", "v[44:45]", "v[52:53]", "v[56:57]", "v[60:61]")
                             LA3DM_RP_SUB("v[46:47]", "v[48:49]", "v[50:51]", "v[54:55]", "v[58:59]", "v[62:63]")
                             LA3DM_RP_SQ("v[40:41]", "v[42:43]", "v[44:45]")
                             LA3DM_RP_SQ("v[46:47]", "v[48:49]", "v[50:51]")
                             "v_pk_add_f32 v[42:43], v[42:43], v[44:45]\n"
                             "v_pk_add_f32 v[48:49], v[48:49], v[50:51]\n"
                             "s_nop 0\n"
                             "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n"
                             "v_pk_add_f32 v[46:47], v[46:47], v[48:49]\n"
                             "s_nop 0\n"
                             LA3DM_RP_PUSH_H("v40", "%[i0]", "%[i1]") LA3DM_RP_PUSH_H("v41", "%[i1]", "%[i0]")
                             LA3DM_RP_PUSH_H("v46", "%[i0]", "%[i1]") LA3DM_RP_PUSH_H("v47", "%[i1]", "%[i0]")
                             : [h] "+v"(hw), [i0] "+v"(idx), [i1] "=&v"(i1), [tail] "+s"(tailb), [tr] "=&v"(tr), [st] "=&s"(st)
                             : [xy] "v"(xy), [zz] "v"(zz), [T] "s"(hit_t), [ca] "v"(ca)
                             : "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
                               "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
#else
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = L.cx[4 * g + u] - xs, dy = L.cy[4 * g + u] - ys, dz = L.cz[4 * g + u] - zs;
                    const float d2 = dx * dx + (dy * dy + dz * dz);
                    const bool hit = d2 < hit_t;
                    const unsigned long long m = __ballot(hit);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (hit) L.ring[((tailb - ring_base) >> 3) + rank] = make_uint2(__float_as_uint(d2), idx);
                    tailb += 8u * (uint32_t)__popcll(m);
                    idx += 1;
                    hw = (hw << 1) | (hit ? 1u : 0u);
                }
#endif
            };
            // one counter and one data-dependent exit per trip (the ring's head room) keep the scalar loop control short
            const uint32_t tail_cap = ring_base + 8u * (uint32_t)(kRing5 - 4 * kWave);
            for (uint32_t left = min(ngroup - g, 8u); left != 0u; --left) {
                b_trip(hA);
                ++g;
                if (tailb > tail_cap) break;
            }
            if (g - g0 == 8u && tailb <= tail_cap)
                for (uint32_t left = ngroup - g; left != 0u; --left) {
                    b_trip(hB);
                    ++g;
                    if (tailb > tail_cap) break;
                }
            const uint32_t nstep = 4 * (g - g0);
            const uint32_t tail = (tailb - ring_base) >> 3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- C: dense evaluation, {d2, candidate} -> {k, k*y} in place ----
            for (uint32_t p = 0; p < ((a.flags & 0x100u) ? 0u : tail); p += kWave) {  // 0x100: profiling ablation
                const uint32_t i = p + lane;
                if (i < tail) {
                    const uint2 e = L.ring[i];
                    const float y = L.cw[e.y];
                    const float kv = cov_sparse_fast<kTrig>(sqrt_cr(__uint_as_float(e.x)), a.sf2);
                    L.ring[i] = make_uint2(__float_as_uint(kv), __float_as_uint(kv * y));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- D: ordered fuse, four candidates per trip; a trip with a neighbour start takes the C++ path ----
            uint32_t roffb = ring_base;
            unsigned long long starts = nbstart >> (4 * g0);
            // oldest candidate first: the first word (hA when the round has more than 32 candidates) is full; the last one
            // holds nlast bits at its bottom
            const uint32_t nlast = nstep > 32u ? nstep - 32u : nstep;
            const uint32_t hlast = nlast ? (nstep > 32u ? hB : hA) << (32u - nlast) : 0u;
            auto d_trips = [&](uint32_t hc, uint32_t ntrip) {
                for (; ntrip != 0u; --ntrip) {
                    const uint32_t sb = (uint32_t)starts & 0xFu;
                    starts >>= 4;
                {
                    // gather (assembly): the four entries land in fixed register pairs, all four reads are back at the end
#if LA3DM_ASM_D
                    register float e0k asm("v56"), e0y asm("v57"), e1k asm("v58"), e1y asm("v59"), e2k asm("v60"), e2y asm("v61"),
                        e3k asm("v62"), e3y asm("v63");
                    float tr;
                    uint32_t st;
#else
                    float e0k, e0y, e1k, e1y, e2k, e2y, e3k, e3y;
#endif
#if LA3DM_ASM_D
                    asm volatile(LA3DM_D_CAND("v[56:57]") LA3DM_D_CAND("v[58:59]") LA3DM_D_CAND("v[60:61]") LA3DM_D_CAND("v[62:63]")
                                 "s_waitcnt lgkmcnt(0)\n"
                                 : [h] "+v"(hc), [roff] "+s"(roffb), [tr] "=&v"(tr), [st] "=&s"(st), "=&v"(e0k), "=&v"(e0y), "=&v"(e1k),
                                   "=&v"(e1y), "=&v"(e2k), "=&v"(e2y), "=&v"(e3k), "=&v"(e3y)  // early clobber: written before zad / h are dead
                                 : [zad] "v"(zero_addr)
                                 : "vcc", "scc", "memory");  // s_bcnt1 / s_lshl3_add write SCC
#else
                    {
                        uint32_t roff = (roffb - ring_base) >> 3;
                        float *ek[4] = {&e0k, &e1k, &e2k, &e3k}, *ey[4] = {&e0y, &e1y, &e2y, &e3y};
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool mine = (hc & 0x80000000u) != 0u;
                            hc <<= 1;
                            const unsigned long long m = __ballot(mine);
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                            const uint2 e = L.ring[mine ? roff + rank : (uint32_t)(kRing5 + kWave)];
                            *ek[u] = __uint_as_float(e.x);
                            *ey[u] = __uint_as_float(e.y);
                            roff += (uint32_t)__popcll(m);
                        }
                        roffb = ring_base + 8u * roff;
                    }
#endif
                    // ordered sums; Occupancy::update where a neighbour starts (uniform).  Adding the zero entry's +0.0
                    // leaves a non-negative-zero accumulator unchanged.
                    if (sb & 1u) flush_nb();
                    ybar += e0y;
                    kbar += e0k;
                    if (sb & 2u) flush_nb();
                    ybar += e1y;
                    kbar += e1k;
                    if (sb & 4u) flush_nb();
                    ybar += e2y;
                    kbar += e2k;
                    if (sb & 8u) flush_nb();
                    ybar += e3y;
                    kbar += e3k;
                }
                }
            };
            if (!(a.flags & 0x400u)) {  // 0x400: profiling ablation
                if (nstep > 32u) d_trips(hA, 8u);
                d_trips(hlast, nlast >> 2);
            }
            __builtin_amdgcn_wave_barrier();
        }
        ncand = 0;
        nbstart = 0;
        // refill in order (rare: > 64 points in a block, or a crowded 7-neighbourhood)
        more = false;
        while (it_b < 7) {
            const uint2 rr = a.nbr_range[7 * blk + it_b];
            const uint32_t pp0 = rr.x, pc = rr.y;
            if (it_base >= pc) {
                ++it_b;
                it_base = 0;
                continue;
            }
            if (ncand != 0u) break;  // one 64-point chunk per refill round
            const bool valid = it_base + lane < pc;
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) p = a.pts[pp0 + it_base + lane];
            stage(p, valid, (int)it_b);
            it_base += kWave;
            more = true;
        }
    }
    flush_nb();
    if (ungated) {  // insert_training_data: update() runs for every trained neighbour
#pragma unroll
        for (int b = 0; b < 7; ++b) updated |= tb[b] >= 0;
    }

    if (active) {
        // the store addresses are formed here, from the leaf index, rather than carried through the kernel from the
        // alpha / beta loads (two 64-bit pointers: the four VGPRs that would otherwise spill at 8 waves per SIMD)
        uint32_t lw = li;
        asm volatile("" : "+v"(lw));
        if (updated) {
            a.alpha[lw] = A;
            a.beta[lw] = B;
            a.state[lw] = (uint8_t)(classify_fast(A, B, a) | 0x80u);
        } else {
            a.state[lw] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// bgk_predict_fuse_r (round 3) — the same pairs and the same kernel values as bgk_predict_fuse_v5, but the two sums
// per leaf are formed WITHOUT the reference's summation order: every evaluated pair adds {k, k*y} to the leaf's pair of
// DOUBLE accumulators in LDS (ds_add_f64 — measured 32 cycles per wave instruction on gfx950, against 770 for
// ds_add_f32: the f32 LDS float atomic is not a hardware path here, the f64 one is), and the epilogue rounds
// alpha + sum(k*y) and beta + (sum(k) - sum(k*y)) to fp32 once.  A double sum of <= a few thousand fp32 terms is exact to
// 2^-53 relative whatever the order, so the result does not depend on the order of the pairs (it is the correctly
// rounded value of what the reference's fp32 chain approximates) and differs from the ordered kernel / the CPU
// restatement by a few fp32 ulps of alpha, beta: |dp| <= ~2e-7, against the 1e-5 the north star asks for.  The 7
// neighbours are not kept apart: Occupancy::update (bgkoctree_node.cpp:31-35) adds ybar and kbar - ybar per neighbour
// with kbar > 0; k >= 0 everywhere, so "some neighbour had kbar > 0" is "sum(k) > 0", and a neighbour with kbar = 0
// adds +0.
//
// What that removes from v5: phase D (the ordered replay: ~400 VALU + 180 SALU + 43 LDS per tile), the hit history,
// the per-neighbour bookkeeping of the staging, the write-back of {k, k*y} to the ring.  What else is new here:
//   * staging walks ONE flat index space over the 7 neighbours' points (bgk_prepare's blk_desc): 64 points per trip
//     whatever neighbour they belong to (v5: one trip per neighbour at ~20 % lane utilisation), neighbour selection
//     by a compare/select chain against the descriptor's prefix sums;
//   * the ring is a queue: a C round evaluates full 64-entry batches only and moves the remainder (< 64 entries)
//     to the front, so every batch but the tile's last runs at full lane utilisation;
//   * hit lanes write under exec = hit mask (no scratch slots, 24 instead of 32 LDS cycles per write).
// LDS per wave: 68 candidates (1 088 B) + 2 x 64 double accumulators (1 024 B) + 376-entry ring (3 008 B) = 5 120 B.
// ---------------------------------------------------------------------------
constexpr int kRingR = 376;
constexpr int kCandPad = kCand5 + 4;
struct __attribute__((aligned(16))) WaveLdsR {
    float cx[kCandPad], cy[kCandPad], cz[kCandPad], cw[kCandPad];  // candidates, one array per component (B reads four at a time)
    double acc0[kWave];  // binary labels: sum(k) over the label-0 pairs;  any labels: sum(k)
    double acc1[kWave];  // binary labels: sum(k) over the label-1 pairs;  any labels: sum(k * y)
    uint2 ring[kRingR];  // {d2, (lane << 13) | (candidate << 2)}
};

// workgroup -> tile: chunks of 8 logical workgroups stay on one XCD (remap 2, the default), see xcd_remap for mode 0
__device__ __forceinline__ uint32_t bgk_task_of_workgroup(const BgkArgs &a) {
    uint32_t wg = blockIdx.x;
    if (a.remap == 0) wg = xcd_remap(wg, gridDim.x);
    else if (a.remap == 2) {
        const uint32_t G8 = gridDim.x & ~63u;
        if (wg < G8) wg = (wg & ~63u) | ((wg & 7u) << 3) | ((wg >> 3) & 7u);
    }
    return __builtin_amdgcn_readfirstlane(wg);
}

// one tile through the general path (any leaf layout, any labels); L: the wave's 5 120 B of LDS
template <int kTrig>
__device__ __forceinline__ void bgk_tile_r(const BgkArgs &a, WaveLdsR &L, const uint32_t task) {
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = task >> a.tpb_shift;
    const uint32_t tile = task & ((1u << a.tpb_shift) - 1u);
    const uint32_t l0 = a.leaf_off[blk] + tile * kWave;
    const uint32_t l1 = a.leaf_off[blk + 1];
    if (l0 >= l1) return;
    const uint32_t nl = min(l1 - l0, (uint32_t)kWave);
    const bool active = lane < nl;
    const uint32_t li = l0 + (active ? lane : 0);
    // every label of this scan is 0 or 1 (bgk_prepare stamps label_seq with the scan's number when it meets another
    // value): a pair then adds k to ONE accumulator, chosen by its label
    const bool binary = a.label_seq[0] != a.seq;

    // flat view of the 7 neighbour ranges
    const uint32_t *dsc = a.blk_desc + 16 * (size_t)(a.desc_shift ? task : blk);
    uint32_t adj[7], pend[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        adj[b] = dsc[b];
        pend[b] = dsc[8 + b];
    }
    const uint32_t M = pend[6];
    // neighbour of flat index f by six compares into six SGPR pairs and a select chain over the offsets (held in VGPRs: a
    // v_cndmask may read one scalar operand, and the mask is one).  In assembly: the compiler turns the C++ select
    // chain into an index chain plus a lookup in a scratch-memory copy of adj[], and its vcc-based pairs need a
    // wait state each (gfx940: VALU writes vcc -> VALU reads it as a mask).
    uint32_t adjv[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        adjv[b] = adj[b];
        asm volatile("" : "+v"(adjv[b]));
    }
    auto load_chunk = [&](uint32_t cb) {
        const uint32_t f = cb + lane;
        uint32_t ad;
        unsigned long long m1, m2, m3, m4, m5, m6;
        asm("v_cmp_le_u32 %[m1], %[e0], %[f]\n"
            "v_cmp_le_u32 %[m2], %[e1], %[f]\n"
            "v_cmp_le_u32 %[m3], %[e2], %[f]\n"
            "v_cmp_le_u32 %[m4], %[e3], %[f]\n"
            "v_cmp_le_u32 %[m5], %[e4], %[f]\n"
            "v_cmp_le_u32 %[m6], %[e5], %[f]\n"
            "v_cndmask_b32 %[ad], %[a0], %[a1], %[m1]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a2], %[m2]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a3], %[m3]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a4], %[m4]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a5], %[m5]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a6], %[m6]\n"
            : [ad] "=&v"(ad), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5), [m6] "=&s"(m6)
            : [f] "v"(f), [e0] "s"(pend[0]), [e1] "s"(pend[1]), [e2] "s"(pend[2]), [e3] "s"(pend[3]), [e4] "s"(pend[4]),
              [e5] "s"(pend[5]), [a0] "v"(adjv[0]), [a1] "v"(adjv[1]), [a2] "v"(adjv[2]), [a3] "v"(adjv[3]), [a4] "v"(adjv[4]),
              [a5] "v"(adjv[5]), [a6] "v"(adjv[6]));
        // lanes past the end hold a point no leaf can reach
        return f < M ? a.pts[f + ad] : make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.0f);
    };
    const float4 q0 = load_chunk(0);
    float4 q1 = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.0f);
    if (M > (uint32_t)kWave) q1 = load_chunk(kWave);

    const uint32_t key = a.leaf_key[li];
    const float4 off4 = a.lut[lut_layer_base(key >> 16) + (key & 0xFFFFu)];
    const float cx = a.blk_center[3 * blk + 0], cy = a.blk_center[3 * blk + 1], cz = a.blk_center[3 * blk + 2];
    const float xs0 = div_by_ell(off4.x + cx, a.ell, a.inv_ell), ys0 = div_by_ell(off4.y + cy, a.ell, a.inv_ell),
                zs0 = div_by_ell(off4.z + cz, a.ell, a.inv_ell);
    float A = a.alpha[li], B = a.beta[li];
    L.acc0[lane] = 0.0;
    L.acc1[lane] = 0.0;

    // bounding box of the tile's leaf centres: as in bgk_predict_fuse_v5
    float lox, loy, loz, hix, hiy, hiz;
    const uint32_t key_first = __builtin_amdgcn_readlane(key, 0), key_last = __builtin_amdgcn_readlane(key, 63);
    if (nl == (uint32_t)kWave && (key_last & 63u) == 0u && key_first == key_last + 63u &&
        __ballot((key >> 16) + 1u != a.depth) == 0ull) {
        lox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs0), 63));
        loy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ys0), 63));
        loz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zs0), 63));
        hix = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs0), 0));
        hiy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ys0), 0));
        hiz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zs0), 0));
    } else {
        lox = wave_min_dpp(xs0), loy = wave_min_dpp(ys0), loz = wave_min_dpp(zs0);  // inactive lanes hold leaf 0's position
        hix = wave_max_dpp(xs0), hiy = wave_max_dpp(ys0), hiz = wave_max_dpp(zs0);
    }
    // the leaf's coordinates as packed operands: (xs, ys) and (zs, -); NaN never matches (inactive lanes)
    la3dm_v2f xy, zz;
    xy.x = active ? xs0 : __builtin_nanf("");
    xy.y = ys0;
    zz.x = zs0;
    zz.y = 0.0f;

    uint32_t ncand = 0;
    auto stage = [&](const float4 &p) {
        const float ex = fmaxf(fmaxf(lox - p.x, p.x - hix), 0.0f);
        const float ey = fmaxf(fmaxf(loy - p.y, p.y - hiy), 0.0f);
        const float ez = fmaxf(fmaxf(loz - p.z, p.z - hiz), 0.0f);
        // as in v5: at or beyond the hit threshold (plus slack for the rounding of this sum) from the box of the leaf
        // centres no leaf can hit; the filler points of lanes past the end fail it too
        const bool keep = (ex * ex + ey * ey + ez * ez) < 0.96780f;
        const unsigned long long m = __ballot(keep);
        const uint32_t slot = ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (keep) {
            L.cx[slot] = p.x;
            L.cy[slot] = p.y;
            L.cz[slot] = p.z;
            L.cw[slot] = p.w;
        }
        ncand += (uint32_t)__popcll(m);
    };

    const float hit_t = __uint_as_float(kHitTBits);
    const uint32_t ring_base = (uint32_t)(uintptr_t)&L.ring[0];
    const uint32_t acc_base = (uint32_t)(uintptr_t)&L.acc0[0];
    const uint32_t cw_base = (uint32_t)(uintptr_t)&L.cw[0];
    const uint32_t cx_base = (uint32_t)(uintptr_t)&L.cx[0];
    const uint32_t tail_cap = ring_base + 8u * (uint32_t)(kRingR - 4 * kWave);
    uint32_t tailb = ring_base;  // LDS byte address of the ring's first free entry

    // C: lane evaluates ring entry i and adds k (and k * y) to the leaf's accumulators
    auto c_eval = [&](uint32_t i) {
        const uint2 e = L.ring[i];
        float y;
        asm volatile("ds_read_b32 %0, %1\n" : "=v"(y) : "v"(cw_base + (e.y & 0xFCu)) : "memory");
        const float kv = cov_sparse_fast<kTrig, true, true>(sqrt_cr(__uint_as_float(e.x)), a.sf2);
        const double kd = (double)kv;
        uint32_t ad = acc_base + (e.y >> 10);
        if (a.flags & 0x800u) ad = acc_base + 8u * lane;  // profiling ablation: conflict-free accumulate (results invalid)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (binary) {
            ad += (__float_as_uint(y) >> 20) & 0x200u;  // 1.0f -> acc1 (512 bytes up), 0.0f -> acc0
            asm volatile("ds_add_f64 %0, %1\n" : : "v"(ad), "v"(kd) : "memory");
        } else {
            const double yd = (double)(kv * y);
            asm volatile("ds_add_f64 %0, %1\n"
                         "ds_add_f64 %0, %2 offset:512\n"
                         :
                         : "v"(ad), "v"(kd), "v"(yd)
                         : "memory");
        }
    };

    uint32_t cb = 0;  // flat index of the next chunk of training points
    for (;;) {
        // A: stage chunks while the worst case still fits the 64-entry candidate list
        while (cb < M) {
            const uint32_t n = min(M - cb, (uint32_t)kWave);
            if (ncand + n > (uint32_t)kCand5) break;
            if (cb == 0u) stage(q0);
            else if (cb == (uint32_t)kWave) stage(q1);
            else stage(load_chunk(cb));
            cb += kWave;
        }
        // pad the list to a multiple of four with points no leaf can reach
        if (lane < 4u) {
            float big;
            asm volatile("v_mov_b32 %0, 0x5e268890" : "=v"(big));
            L.cx[ncand + lane] = big;
            L.cy[ncand + lane] = big;
            L.cz[ncand + lane] = big;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t ngroup = (a.flags & 0x200u) ? 0u : (ncand + 3u) >> 2;  // 0x200: profiling ablation
        uint32_t g = 0;
        while (g < ngroup) {
            // ---- B: test + push ----
            uint32_t i0 = (lane << 13) | (g << 4);  // entry word of candidate 4 g: candidate * 4 in bits 2-7
            uint32_t ca = cx_base + 16u * g;        // LDS byte address of cx[4 g]
            for (uint32_t left = ngroup - g; left != 0u; --left) {
                uint32_t i1, st;
                float tr;
                asm volatile(LA3DM_RP_LOAD
                             LA3DM_RP_SUB("v[40:41]", "v[42:43]", "v[44:45]", "v[52:53]", "v[56:57]", "v[60:61]")
                             LA3DM_RP_SUB("v[46:47]", "v[48:49]", "v[50:51]", "v[54:55]", "v[58:59]", "v[62:63]")
                             LA3DM_RP_SQ("v[40:41]", "v[42:43]", "v[44:45]")
                             LA3DM_RP_SQ("v[46:47]", "v[48:49]", "v[50:51]")
                             "v_pk_add_f32 v[42:43], v[42:43], v[44:45]\n"
                             "v_pk_add_f32 v[48:49], v[48:49], v[50:51]\n"
                             "s_nop 0\n"
                             "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n"
                             "v_pk_add_f32 v[46:47], v[46:47], v[48:49]\n"
                             "s_nop 0\n"
                             LA3DM_RP_PUSH("v40", "%[i0]", "%[i1]") LA3DM_RP_PUSH("v41", "%[i1]", "%[i0]")
                             LA3DM_RP_PUSH("v46", "%[i0]", "%[i1]") LA3DM_RP_PUSH("v47", "%[i1]", "%[i0]")
                             : [i0] "+v"(i0), [i1] "=&v"(i1), [tail] "+s"(tailb), [tr] "=&v"(tr), [st] "=&s"(st)
                             : [xy] "v"(xy), [zz] "v"(zz), [T] "s"(hit_t), [ca] "v"(ca)
                             : "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
                               "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
                ++g;
                ca += 16u;
                if (tailb > tail_cap) break;
            }
            if (tailb > tail_cap) {
                // ---- C: the full batches; the remainder moves to the front of the ring ----
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t tail = (tailb - ring_base) >> 3;
                const uint32_t nfull = tail & ~63u, rem = tail & 63u;
                // (the batches come from the ring's end, entries [rem, tail): the remainder stays at the front, nothing moves)
                if (!(a.flags & 0x100u))  // 0x100: profiling ablation
                    for (uint32_t p = rem; p < tail; p += kWave) c_eval(p + lane);
                tailb = ring_base + 8u * rem;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // the ring's entries name their candidate by its slot in the list (the label is read from there): everything
        // is evaluated before the next round of candidates overwrites the list — the tile's last batch, or the last
        // batch of a round of a tile with more than 64 candidates, is the only one that may be partly filled
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            const uint32_t tail = (tailb - ring_base) >> 3;
            if (!(a.flags & 0x100u))
                for (uint32_t p = 0; p < tail; p += kWave)
                    if (p + lane < tail) c_eval(p + lane);
            tailb = ring_base;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ncand = 0;
        if (cb >= M) break;
    }

    if (active) {
        const double s0 = L.acc0[lane], s1 = L.acc1[lane];
        const double K = binary ? s0 + s1 : s0, Y = s1;
        const bool updated = K > 0.0 || (a.flags & 1u) != 0u;  // flag 1: insert_training_data, update() runs unconditionally
        uint32_t lw = li;
        asm volatile("" : "+v"(lw));
        if (updated) {
            A = (float)((double)A + Y);
            B = (float)((double)B + (K - Y));
            a.alpha[lw] = A;
            a.beta[lw] = B;
            a.state[lw] = (uint8_t)(classify_fast(A, B, a) | 0x80u);
        } else {
            a.state[lw] = 0;
        }
    }
}

template <int kTrig>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(8, 8))) void bgk_predict_fuse_r(BgkArgs a) {
    __shared__ WaveLdsR L;
    const uint32_t task = bgk_task_of_workgroup(a);
    if (task >= a.n_tasks) return;
    bgk_tile_r<kTrig>(a, L, task);
}

// ---------------------------------------------------------------------------
// bgk_predict_fuse_t (round 4, the default of bgk_sum = 1) — per-axis distance TABLES for aligned 4x4x4 tiles.
//
// Same pairs, same fp32 kernel values and the same double accumulators as bgk_predict_fuse_r; what changes is how a
// (candidate, leaf) distance is formed.  The 64 leaves of an un-pruned tile are the product grid {X0..X3} x {Y0..Y3} x
// {Z0..Z3} (LUT offsets depend on an axis' own index bits only, bgkblock.cpp:7-32), so the reference's
//     d2 = dx*dx + (dy*dy + dz*dz)                                             (bgkinference.h:88-93)
// of candidate c and leaf (i, j, k) is  X2[i][c] + (Y2[j][c] + Z2[k][c])  with 12 squares per candidate instead of
// 64 x 3: the staging lanes (lane = training point) compute them — (p.x - X_i)^2 etc., the same two roundings as the
// per-leaf form — and write them TRANSPOSED into LDS (row = axis value, column = candidate slot); the test lanes
// (lane = leaf) read their three rows for four candidates at a time (3 ds_read_b128) and need two packed adds per PAIR
// of candidates: 1 VALU per candidate for the distance instead of 4.  Bit-identical d2 (same operations, same order).
//   * the box cull is exact and free of slack: fp32 + and * are monotone, so min over the leaves of d2 is
//     minX2 + (minY2 + minZ2); a point is staged iff that is below the hit threshold, i.e. iff SOME leaf hits;
//   * the label travels in the SIGN of the table entries (label 1: all twelve squares negated; the sums of negated
//     terms are the negated sums, bit for bit), the hit test compares |d2|, the ring entry is {+-d2, address of the
//     leaf's accumulator 0}: no per-candidate entry word (one VALU per candidate less in B) and no label read in C;
//   * ring entries no longer name a candidate slot, so the ring is only drained at the end of the tile: a tile has ONE
//     partly filled batch whatever its candidate count, and the tables can be small (32 slots: 1 536 B).
// The table path takes the tiles of FULL blocks (8^(depth-1) leaves, i.e. nothing pruned: decided by the leaf count alone,
// no key is read); the tiles of the other blocks go through bgk_tile_r in the same launch.  The kernel needs every label
// to be 0 or 1 (LA3DM_SCAN_LABELS_01); without that guarantee the host launches bgk_predict_fuse_r.  kGeneral = false is
// the same kernel without the general path, for scans whose caller vouches that no test block is pruned
// (LA3DM_SCAN_FULL_BLOCKS: the first scan into an empty map, a packed scan whose leaf count says so): the general
// path's presence costs the table path ~2 % (register allocation of the shared prologue).
// LDS per wave: 2 x 12 x 16 table (1 536 B) + 2 x 64 double accumulators (1 024 B) + 320-entry ring (2 560 B) = 5 120 B.
// B per candidate: 1 (packed adds) + v_cmp + 2 v_mbcnt + v_lshl_add = 5 VALU (r: 9).
// ---------------------------------------------------------------------------
#ifndef LA3DM_T_EARLY_AB
#define LA3DM_T_EARLY_AB 1
#endif
constexpr int kTabSlots = 32;
constexpr int kRingT = 320;
struct __attribute__((aligned(16))) WaveLdsT {
    // rows 0-3 (p.x - X_i)^2, 4-7 y, 8-11 z; all twelve negated for a label-1 point.  Two halves of 16 slots: a row is 64
    // bytes, so the four rows of an axis that the lanes of one ds_read_b128 group read (same column group, different
    // rows) start 16 banks apart and cover the 64 banks exactly — with 32-slot rows (128 B) rows i and i + 2 met on
    // the same banks: 136 LDS conflict cycles per tile, a fifth of the kernel's LDS-array time
    uint2 ring[kRingT];        // {+-d2, LDS address of acc0[leaf]}
    float tab[kTabSlots / 16][12][16];
    double acc0[kWave];        // sum(k) over the label-0 pairs: at byte 4 096 — bit 9 of its addresses is clear, and
    double acc1[kWave];        // sum(k) over the label-1 pairs: 512 bytes up, so the label bit can be OR-ed into the address
};
static_assert(offsetof(WaveLdsT, acc0) == 4096 && offsetof(WaveLdsT, tab) % 64 == 0, "c_eval ORs the label into bit 9 of acc0's address; the table rows are 64-byte aligned");
static_assert(sizeof(WaveLdsT) == 5120 && sizeof(WaveLdsR) == 5120, "8 waves per SIMD need <= 5 120 B of LDS per wave");

// B, four candidates (one table column group): the four compares first (four SGPR pairs), then the four pushes.  Push of
// one candidate: 3 VALU + 4 SALU + 1 LDS, all under exec = hit mask (only the hit lanes need a rank); hit lanes write
// {d2, w} at ring[tail + rank].
#define LA3DM_TP_LOAD(OFF)                                                                \
    "ds_read_b128 v[52:55], %[ax] offset:" OFF "\n"                                       \
    "ds_read_b128 v[56:59], %[ay] offset:" OFF "\n"                                       \
    "ds_read_b128 v[60:63], %[az] offset:" OFF "\n"                                       \
    "s_waitcnt lgkmcnt(0)\n"                                                              \
    "v_pk_add_f32 v[56:57], v[56:57], v[60:61]\n"                                         \
    "v_pk_add_f32 v[58:59], v[58:59], v[62:63]\n"                                         \
    "s_nop 0\n"                                                                           \
    "v_pk_add_f32 v[52:53], v[52:53], v[56:57]\n"                                         \
    "v_pk_add_f32 v[54:55], v[54:55], v[58:59]\n"                                         \
    "s_nop 0\n"                                                                           \
    "v_cmp_gt_f32_e64 %[m0], %[T], |v52|\n"                                               \
    "v_cmp_gt_f32_e64 %[m1], %[T], |v53|\n"                                               \
    "v_cmp_gt_f32_e64 %[m2], %[T], |v54|\n"                                               \
    "v_cmp_gt_f32_e64 %[m3], %[T], |v55|\n"
// the rank is only needed on the hit lanes: exec = mask first, the two v_mbcnt read exec as their mask operand
#define LA3DM_TP_PUSH(D, M)                                 \
    "s_mov_b64 exec, " M "\n"                               \
    "s_bcnt1_i32_b64 %[st], " M "\n"                        \
    "v_mbcnt_lo_u32_b32 %[tr], exec_lo, 0\n"                \
    "v_mbcnt_hi_u32_b32 %[tr], exec_hi, %[tr]\n"            \
    "v_lshl_add_u32 %[tr], %[tr], 3, %[tail]\n"             \
    "ds_write2_b32 %[tr], " D ", %[w] offset1:1\n"          \
    "s_lshl3_add_u32 %[tail], %[st], %[tail]\n"
#define LA3DM_TP_TRIP(OFF)                                                                          \
    asm volatile(LA3DM_TP_LOAD(OFF)                                                                 \
                 LA3DM_TP_PUSH("v52", "%[m0]") LA3DM_TP_PUSH("v53", "%[m1]") LA3DM_TP_PUSH("v54", "%[m2]") LA3DM_TP_PUSH("v55", "%[m3]") \
                 "s_mov_b64 exec, -1\n"                                                             \
                 : [tail] "+s"(tailb), [tr] "=&v"(tr), [st] "=&s"(st), [m0] "=&s"(hm0), [m1] "=&s"(hm1), [m2] "=&s"(hm2), [m3] "=&s"(hm3) \
                 : [ax] "v"(aX), [ay] "v"(aY), [az] "v"(aZ), [T] "s"(hit_t), [w] "v"(w0)            \
                 : "scc", "memory", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63")

template <int kTrig, bool kGeneral = true>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(8, 8))) void bgk_predict_fuse_t(BgkArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_lds[5120];
    const uint32_t task = bgk_task_of_workgroup(a);
    if (task >= a.n_tasks) return;
    const uint32_t blk = task >> a.tpb_shift;
    // the table path takes the tiles of FULL blocks (8^(depth-1) leaves: no node above the finest level is a leaf, so
    // every 64 consecutive leaves in LeafIterator order are one aligned 4x4x4 cube); the tiles of the other blocks go
    // through the general path — decided here, before anything else is live, so that each path keeps its own
    // register allocation (behind a later branch the general path spilled to scratch memory)
    // the block's scalar inputs — leaf range, the descriptor of its 7 neighbour ranges, centre — in ONE round trip (the
    // compiler's own schedule chained them: leaf range, then the descriptor's count, then its other words, the centre
    // as three vector loads)
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t *dsc = a.blk_desc + 16 * (size_t)(a.desc_shift ? task : blk);
    u32x8 dlo, dhi;
    u32x2 lbr;
    float cx, cy, cz;
    asm("s_load_dwordx2 %[lb], %[lop], 0x0\n"  // (not volatile: behind a volatile asm the general path's scalar reads turn into vector loads)
                 "s_load_dwordx8 %[dlo], %[dsc], 0x0\n"
                 "s_load_dwordx8 %[dhi], %[dsc], 0x20\n"
                 "s_load_dword %[cx], %[ctr], 0x0\n"
                 "s_load_dword %[cy], %[ctr], 0x4\n"
                 "s_load_dword %[cz], %[ctr], 0x8\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 : [lb] "=&s"(lbr), [dlo] "=&s"(dlo), [dhi] "=&s"(dhi), [cx] "=&s"(cx), [cy] "=&s"(cy), [cz] "=&s"(cz)
                 : [lop] "s"(a.leaf_off + blk), [dsc] "s"(dsc), [ctr] "s"(a.blk_center + 3 * (size_t)blk));
    const uint32_t lb0 = lbr[0];
    if (lbr[1] - lb0 != 1u << (3u * (a.depth - 1u))) {
        if (kGeneral) bgk_tile_r<kTrig>(a, *reinterpret_cast<WaveLdsR *>(s_lds), task);
        return;  // (kGeneral false: the caller vouched for full blocks, LA3DM_SCAN_FULL_BLOCKS — a tile that is not is left untouched)
    }
    WaveLdsT &L = *reinterpret_cast<WaveLdsT *>(s_lds);
    const uint32_t lane = threadIdx.x;
    const uint32_t tile = task & ((1u << a.tpb_shift) - 1u);
    const uint32_t li = lb0 + tile * kWave + lane;
    // LeafIterator order is descending (bgkoctree.h:101-135): the leaf at list position j of a full block has the
    // finest-level index 8^(depth-1) - 1 - j, so the key needs no load
    // (LUT entry: the finest layer's base + that index — wave-uniform base, one subtraction per lane)
    const uint32_t n_fine = 1u << (3u * (a.depth - 1u));
    const uint32_t lut_idx = lut_layer_base(a.depth - 1u) + (n_fine - 1u - tile * kWave) - lane;

    // flat view of the 7 neighbour ranges (bgk_prepare's blk_desc, as in bgk_tile_r).  The first two chunks use the
    // descriptor read above; a tile with more than 128 points reads it again per further chunk (rare) — its 13 words
    // would otherwise sit in registers for the whole tile.
    // the points of flat indices cb + lane; a lane past the end reads the range's last point (the caller masks it)
    auto gather = [&](const uint32_t (&adjv)[7], const uint32_t (&pend)[6], uint32_t cb, uint32_t M) {
        const uint32_t f = min(cb + lane, M - 1u);
        uint32_t ad;
        unsigned long long m1, m2, m3, m4, m5, m6;
        asm("v_cmp_le_u32 %[m1], %[e0], %[f]\n"
            "v_cmp_le_u32 %[m2], %[e1], %[f]\n"
            "v_cmp_le_u32 %[m3], %[e2], %[f]\n"
            "v_cmp_le_u32 %[m4], %[e3], %[f]\n"
            "v_cmp_le_u32 %[m5], %[e4], %[f]\n"
            "v_cmp_le_u32 %[m6], %[e5], %[f]\n"
            "v_cndmask_b32 %[ad], %[a0], %[a1], %[m1]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a2], %[m2]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a3], %[m3]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a4], %[m4]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a5], %[m5]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a6], %[m6]\n"
            : [ad] "=&v"(ad), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5), [m6] "=&s"(m6)
            : [f] "v"(f), [e0] "s"(pend[0]), [e1] "s"(pend[1]), [e2] "s"(pend[2]), [e3] "s"(pend[3]), [e4] "s"(pend[4]),
              [e5] "s"(pend[5]), [a0] "v"(adjv[0]), [a1] "v"(adjv[1]), [a2] "v"(adjv[2]), [a3] "v"(adjv[3]), [a4] "v"(adjv[4]),
              [a5] "v"(adjv[5]), [a6] "v"(adjv[6]));
        return a.pts[f + ad];
    };
    const uint32_t M = dhi[6];
#if LA3DM_T_EARLY_AB
    const float A0 = a.alpha[li], B0 = a.beta[li];  // needed by the epilogue only: loaded here, a round trip off the tile's tail
#endif
    if (M == 0u) {  // no training point in the 7 blocks: nothing reaches the tile
        if (!(a.flags & 1u)) a.state[li] = 0;
        else {  // insert_training_data: update() runs with (0, 0)
            const float A = a.alpha[li], B = a.beta[li];
            a.state[li] = (uint8_t)(classify(A, B, a) | 0x80u);
        }
        return;
    }
    float4 pc, pn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    {
        uint32_t adj[7], pend[6];
#pragma unroll
        for (int b = 0; b < 7; ++b) adj[b] = dlo[b];
#pragma unroll
        for (int b = 0; b < 6; ++b) pend[b] = dhi[b];
        // the offsets in VGPRs (a v_cndmask reads one scalar operand, and its mask is one)
        uint32_t adjv[7];
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            adjv[b] = adj[b];
            asm volatile("" : "+v"(adjv[b]));
        }
        pc = gather(adjv, pend, 0, M);
        if (M > (uint32_t)kWave) pn = gather(adjv, pend, kWave, M);
    }
    auto gather_cold = [&](uint32_t cb) {  // an explicit scalar read (behind the memory-clobbering asm blocks the compiler makes it vector loads)
        u32x8 lo, hi;
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n"
                     "s_load_dwordx8 %1, %2, 0x20\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&s"(lo), "=&s"(hi)
                     : "s"(dsc));
        uint32_t adjv[7] = {lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6]};
#pragma unroll
        for (int b = 0; b < 7; ++b) asm volatile("" : "+v"(adjv[b]));
        const uint32_t pend[6] = {hi[0], hi[1], hi[2], hi[3], hi[4], hi[5]};
        return gather(adjv, pend, cb, M);
    };

    const float4 off4 = a.lut[lut_idx];
    const float xs0 = div_by_ell(off4.x + cx, a.ell, a.inv_ell), ys0 = div_by_ell(off4.y + cy, a.ell, a.inv_ell),
                zs0 = div_by_ell(off4.z + cz, a.ell, a.inv_ell);
    L.acc0[lane] = 0.0;
    L.acc1[lane] = 0.0;

    // the four coordinates per axis.  Lane l holds leaf index c = 63 - l of the cube; c = (i1 j1 k1 i0 j0 k0) in binary:
    // child number i*4 + j*2 + k at the parent level (bits 5-3) and at the leaf level (bits 2-0), bgkblock.cpp:14-27.
    // Axis value r = 2 * (high bit) + (low bit); the lanes read below have the other two axes' bits clear.
    auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    const la3dm_v2f X01 = {rl(xs0, 63), rl(xs0, 59)}, X23 = {rl(xs0, 31), rl(xs0, 27)};
    const la3dm_v2f Y01 = {rl(ys0, 63), rl(ys0, 61)}, Y23 = {rl(ys0, 47), rl(ys0, 45)};
    const la3dm_v2f Z01 = {rl(zs0, 63), rl(zs0, 62)}, Z23 = {rl(zs0, 55), rl(zs0, 54)};
    const uint32_t c6 = lane ^ 63u;
    const uint32_t ix = ((c6 >> 4) & 2u) | ((c6 >> 2) & 1u), iy = ((c6 >> 3) & 2u) | ((c6 >> 1) & 1u), iz = ((c6 >> 2) & 2u) | (c6 & 1u);
    const uint32_t tab_base = (uint32_t)(uintptr_t)&L.tab[0][0][0];
    constexpr uint32_t kRowB = 4u * 16u, kHalfB = 12u * kRowB;
    static_assert(kTabSlots == 32, "the B loop below walks two halves of 16 slots");
    const uint32_t ax0 = tab_base + ix * kRowB, ay0 = tab_base + (4u + iy) * kRowB, az0 = tab_base + (8u + iz) * kRowB;
    const uint32_t ring_base = (uint32_t)(uintptr_t)&L.ring[0];
    const uint32_t w0 = (uint32_t)(uintptr_t)&L.acc0[0] + 8u * lane;
    if (w0 & 0x200u) __builtin_trap();   // (s_lds is the kernel's only static LDS object: it starts at LDS address 0)
    static_assert(offsetof(WaveLdsT, acc1) - offsetof(WaveLdsT, acc0) == 512, "c_eval adds 512 to the address of acc0[leaf] for a label-1 pair");
    const float hit_t = __uint_as_float(kHitTBits);
    const uint32_t tail_cap = ring_base + 8u * (uint32_t)(kRingT - 4 * kWave);
    uint32_t tailb = ring_base;  // LDS byte address of the ring's first free entry

    // C: lane evaluates ring entry i and adds k to the leaf's accumulator 0 or 1 (the sign of d2 is the label).
    // (Measured and dropped, round 4: draining the ring only when a trip's hits no longer fit — three or four batches
    // per round instead of mostly one — with the batches software-pipelined (ring read, square root, reduction and table
    // gather of batch b + 1 ahead of the polynomials of batch b): 72.9 against 72.1 us.)
    auto c_eval = [&](uint32_t i) {
        const uint2 e = L.ring[i];
        const float kv = cov_sparse_fast<kTrig, true, true>(sqrt_cr(__builtin_fabsf(__uint_as_float(e.x))), a.sf2);
        const double kd = (double)kv;
        const uint32_t ad = e.y | ((e.x >> 22) & 0x200u);  // label 1 (negative d2): acc1[leaf], 512 bytes up (v_lshrrev + v_and_or)
        asm volatile("ds_add_f64 %0, %1\n" : : "v"(ad), "v"(kd) : "memory");
    };
    // C round: the full 64-entry batches; the remainder (< 64 entries) moves to the front of the ring
    auto c_flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t tail = (tailb - ring_base) >> 3;
        const uint32_t nfull = tail & ~63u, rem = tail & 63u;
        // the full batches are taken from the ring's END — entries [rem, tail) — so that the remainder stays where it is,
        // at the front (the sums are order-free; moving the remainder cost a read, a write and two barriers per round)
        if (!(a.flags & 0x100u))  // 0x100: profiling ablation
            for (uint32_t p = rem; p < tail; p += kWave) c_eval(p + lane);
        tailb = ring_base + 8u * rem;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // A + B for one chunk of 64 training points (lane = point)
    auto chunk = [&](const float4 &p, const uint32_t cb) {
        const la3dm_v2f bx = {p.x, p.x}, by = {p.y, p.y}, bz = {p.z, p.z};
        la3dm_v2f x01 = bx - X01, x23 = bx - X23, y01 = by - Y01, y23 = by - Y23, z01 = bz - Z01, z23 = bz - Z23;
        x01 *= x01, x23 *= x23, y01 *= y01, y23 *= y23, z01 *= z01, z23 *= z23;
        const float mx = fminf(fminf(x01.x, x01.y), fminf(x23.x, x23.y));
        const float my = fminf(fminf(y01.x, y01.y), fminf(y23.x, y23.y));
        const float mz = fminf(fminf(z01.x, z01.y), fminf(z23.x, z23.y));
        // min over the 64 leaves of d2 (+ and * are monotone): staged iff some leaf hits (lanes past the end hold a
        // copy of the last point: masked)
        const bool keep = mx + (my + mz) < hit_t && cb + lane < M;
        const unsigned long long m = __ballot(keep);
        if (m == 0ull) return;
        const float sg = 1.0f - (p.w + p.w);  // label 0 -> +1, label 1 -> -1 (exact)
        const la3dm_v2f s2 = {sg, sg};
        x01 *= s2, x23 *= s2, y01 *= s2, y23 *= s2, z01 *= s2, z23 *= s2;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        const uint32_t n = (uint32_t)__popcll(m);
        for (uint32_t base = 0; base < n; base += kTabSlots) {
            const uint32_t nr = min(n - base, (uint32_t)kTabSlots), ngroup = (nr + 3u) >> 2;
            // pad the last column group with entries no leaf can reach (an X row suffices: the sum stays huge or NaN)
            if (lane < 4u && nr + lane < 4u * ngroup) {
                float big;
                asm volatile("v_mov_b32 %0, 0x5e268890" : "=v"(big));
#pragma unroll
                for (int r = 0; r < 4; ++r) L.tab[(nr + lane) >> 4][r][(nr + lane) & 15u] = big;
            }
            const uint32_t slot = rank - base;
            if (keep && slot < (uint32_t)kTabSlots) {
                float(&T)[12][16] = L.tab[slot >> 4];
                const uint32_t c = slot & 15u;
                T[0][c] = x01.x, T[1][c] = x01.y, T[2][c] = x23.x, T[3][c] = x23.y;
                T[4][c] = y01.x, T[5][c] = y01.y, T[6][c] = y23.x, T[7][c] = y23.y;
                T[8][c] = z01.x, T[9][c] = z01.y, T[10][c] = z23.x, T[11][c] = z23.y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (a.flags & 0x200u) continue;  // 0x200: profiling ablation
            uint32_t aX = ax0, aY = ay0, aZ = az0;
            for (uint32_t g = 0; g < ngroup; g += 2u) {  // two column groups per trip: the three row addresses move once
                uint32_t st;
                float tr;
                unsigned long long hm0, hm1, hm2, hm3;
                LA3DM_TP_TRIP("0");
                if (tailb > tail_cap) c_flush();
                if (g + 1u < ngroup) {
                    LA3DM_TP_TRIP("16");
                    if (tailb > tail_cap) c_flush();
                }
                const uint32_t step = g == 2u ? kHalfB - 32u : 32u;   // column groups 0-3 sit in the first half, 4-7 in the second
                aX += step, aY += step, aZ += step;
            }
            // (the next sub-round overwrites the table: LDS operations of a wave complete in order)
        }
    };

    for (uint32_t cb = 0;;) {
        chunk(pc, cb);
        cb += kWave;
        if (cb >= M) break;
        pc = pn;
        if (cb + kWave < M) pn = gather_cold(cb + kWave);
    }

    // the tile's last, partly filled batch(es)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const uint32_t tail = (tailb - ring_base) >> 3;
        if (!(a.flags & 0x100u))
            for (uint32_t p = 0; p < tail; p += kWave)
                if (p + lane < tail) c_eval(p + lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    {
        const double s0 = L.acc0[lane], s1 = L.acc1[lane];
        const double K = s0 + s1, Y = s1;
        uint32_t lw = li;
        asm volatile("" : "+v"(lw));
        if (K > 0.0 || (a.flags & 1u) != 0u) {  // flag 1: insert_training_data, update() runs unconditionally
#if LA3DM_T_EARLY_AB
            const float A = (float)((double)A0 + Y);
            const float B = (float)((double)B0 + (K - Y));
#else
            const float A = (float)((double)a.alpha[lw] + Y);
            const float B = (float)((double)a.beta[lw] + (K - Y));
#endif
            a.alpha[lw] = A;
            a.beta[lw] = B;
            a.state[lw] = (uint8_t)(classify_fast(A, B, a) | 0x80u);
        } else {
            a.state[lw] = 0;
        }
    }
}

}  // namespace la3dm_dev
namespace la3dm_dev {

// exhaustive sweeps of the kernel's shortcuts against the IEEE operations:
// counts fp32 inputs in [lo_bits, hi_bits] (as unsigned bit patterns) where they differ.
__global__ void sweep_check_kernel(int what, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mismatch, float ell,
                                   float inv_ell, float sf2) {
    const uint64_t n = (uint64_t)hi_bits - lo_bits + 1;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        bool ok;
        if (what == 0) ok = div_const(x, 3.0f, 0.333333343f) == x / 3.0f;
        else if (what == 1) ok = div_const2(x, 2.0f * 3.1415926f, 0.159154952f) == x / (2.0f * 3.1415926f);
        else if (what == 2) ok = sqrt_cr(x) == sqrtf(x);
        else if (what == 4) ok = sqrtf(x) == (float)sqrt((double)x);
        else if (what == 5) ok = div_const(x, 2.0f * 3.1415926f, 0.159154952f) == x / (2.0f * 3.1415926f);
        else if (what == 7) ok = div_by_ell(x, ell, inv_ell) == x / ell && div_by_ell(-x, ell, inv_ell) == -x / ell;
        else if (what == 8) ok = !(cov_sparse_fast<0>(sqrt_cr(x), sf2) > 0.0f);  // no support left at this d2
        else if (what == 9) {  // fp32 quotient == the double quotient narrowed (lv_kernels.h lv_seg_point): x against 8 hashed divisors
            ok = true;
            uint32_t h = (lo_bits + (uint32_t)i) * 2654435761u + 0x9E3779B9u;
            for (int rep = 0; rep < 8; ++rep) {
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
                // a divisor in [x, 4097 x): the projection parameter c1 / c2 lies in (0, 1)
                const float y = x * (1.0f + (float)(h & 0xFFFFFFu) * 0x1p-12f);
                const float q = x / y;
                if (q >= 0x1p-100f) ok = ok && q == (float)((double)x / (double)y);
                const float y2 = __uint_as_float(0x30000000u + (h & 0x1FFFFFFFu));  // any magnitude in [2^-31, 2^32)
                const float q2 = x / y2;
                if (q2 >= 0x1p-100f && q2 < 0x1p100f) ok = ok && q2 == (float)((double)x / (double)y2);
            }
        }
        else {  // sincos_cr vs the double-precision library functions rounded to float
            float s, c;
            sincos_cr(x, s, c);
            ok = s == (float)sin((double)x) && c == (float)cos((double)x);
        }
        bad += ok ? 0 : 1;
    }
    if (bad) atomicAdd(mismatch, bad);
}

// diagnostics for the parity tests
__global__ void diag_eval_kernel(int op, const float *in, float *out, uint32_t n, float sf2, float ell, float free_thresh = 0.0f,
                                 float occupied_thresh = 0.0f, float var_thresh = 0.0f) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (op == 12 || op == 13) {  // node state of (alpha, beta) = (in[2 j], in[2 j + 1]): 12 classify_fast, 13 classify; out[2 j] = state
        // (whole waves stay active: classify_fast votes over the wave)
        const uint32_t j = i < n / 2u ? i : 0u;
        BgkArgs a;
        a.free_thresh = free_thresh;
        a.occupied_thresh = occupied_thresh;
        a.var_thresh = var_thresh;
        const float A = in[2 * j], B = in[2 * j + 1];
        const uint8_t st = op == 12 ? classify_fast(A, B, a) : classify(A, B, a);
        if (i < n / 2u) {
            out[2 * i] = (float)st;
            out[2 * i + 1] = 0.0f;
        }
        return;
    }
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
    case 14: sincos_eigen337(x, s, c); y = s; break;
    case 15: sincos_eigen337(x, s, c); y = c; break;
    case 16: y = cov_sparse<true, 3>(x, sf2); break;
    case 17: y = cov_sparse_fast<3, true>(x, sf2); break;
    default: y = 0.0f;
    }
    out[i] = y;
}

}  // namespace la3dm_dev
