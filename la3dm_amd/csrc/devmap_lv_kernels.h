// devmap_lv_kernels.h — BGKLVOctoMap on the device-resident pool: the stages of insert_pointcloud
// (src/bgklvoctomap/bgklvoctomap.cpp:89-285) around the per-voxel kernel of lv_kernels.h.
//
//   front end      ray shortening, the downward-ray filter, free segments and their samples      :303-423, :439-462
//   partition      bounding box -> candidate blocks (all of them are created)                    :105-135
//                  gather grid: samples bucketed on a grid of edge g aligned with the blocks     (replaces the R-tree, :137-160)
//                  blocks with a sample within reach of one of their voxels are packed            (:170-176 "has information")
//   per pass       bgklv_voxel_kernel in place on the pool, then dm_lv_finish                    :155-255
//   afterwards     prune of the blocks that had information                                       :262-273
//
// Everything is the arithmetic of la3dm_amd/csrc/host/bgklvoctomap.cpp (which is bit-identical to the oracle) moved
// onto the GPU expression by expression — the reference mixes float and double freely here and every promotion is kept.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "devmap_kernels.h"

namespace la3dm_dev {

struct LvBeamArgs {
    float ox, oy, oz;
    float max_range, free_res;
    double offset;     // ell * sqrt(2)   (bgklvoctomap.cpp:313)
    double influence;  // ell
};

// point3f::norm(): double sqrt of a float sum of float squares
__device__ __forceinline__ double lv_norm(float x, float y, float z) { return sqrt((double)(x * x + y * y + z * z)); }

// distance of every hit from the sensor (used by every beam for every hit)
__global__ __launch_bounds__(256) void dm_lv_ranges(const float *__restrict__ hits, uint32_t nh, LvBeamArgs a, double *__restrict__ rng) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nh) return;
    rng[q] = lv_norm(a.ox - hits[3 * (size_t)q], a.oy - hits[3 * (size_t)q + 1], a.oz - hits[3 * (size_t)q + 2]);
}

constexpr uint32_t kLvBeamHit = 1u, kLvBeamSkip = 2u;
constexpr uint32_t kErrLvExtent = 4u;   // counters[kCntError]: coordinates beyond a grid's index range

// ---- the beam computation in two parallel-friendly steps (the dense form: small scans, nh < 8192) ----
// One lane per beam walking all nh hits (the reference's loop, bgklvoctomap.cpp:313-423: the shortening is order dependent — each
// accepted hit changes the length the next ones are tested against) is nh / 64 waves of nh iterations with three f64 square roots
// each — 0.7 ms for a 3 000-hit scan on a chip that then sits 95 % idle.  The membership test of the "nearby" gather depends
// only on the beam's INITIAL length and end point, so it runs for all (beam, hit) pairs at once (one wave per beam and
// 64 hits, the ballot is the row's mask word); the order-dependent shortening then only visits the set bits.
struct LvBeam {
    double l0;
    float ex, ey, ez;   // initial free_endpt
    float nx, ny, nz;
    float pz;
    uint32_t fl;
};
__global__ __launch_bounds__(256) void dm_lv_beam_init(const float *__restrict__ hits, uint32_t nh, LvBeamArgs a, LvBeam *__restrict__ beams) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nh) return;
    const float px = hits[3 * (size_t)h], py = hits[3 * (size_t)h + 1], pz = hits[3 * (size_t)h + 2];
    const float ox = a.ox, oy = a.oy, oz = a.oz;
    double l = lv_norm(px - ox, py - oy, pz - oz);
    LvBeam b;
    b.nx = (float)((px - ox) / l);
    b.ny = (float)((py - oy) / l);
    b.nz = (float)((pz - oz) / l);
    b.fl = 0;
    if (a.max_range > 0) {
        if (l < a.max_range) {
            l = (float)sqrt((double)((px - ox) * (px - ox) + (py - oy) * (py - oy) + (pz - oz) * (pz - oz)));
            l = l - a.offset;
            b.fl |= kLvBeamHit;
        } else {
            l = a.max_range - a.offset;
        }
    }
    b.l0 = l;
    b.ex = (float)(ox + b.nx * l);
    b.ey = (float)(oy + b.ny * l);
    b.ez = (float)(oz + b.nz * l);
    b.pz = pz;
    beams[h] = b;
}
// grid (ceil(nh / 64), ceil(nh / kLvNearTile)), 64 threads: mask[h * nw + w] bit j <=> hit 64 w + j is "nearby" for beam h AND
// within `influence` of the beam's line.  A wave keeps its 64 hits in registers and walks a tile of kLvNearTile beams (their
// records are wave-uniform scalar loads): one wave per (beam, 64 hits) re-read the hit list once per beam — 24 GB through
// the L2 at 45 k hits, 6.2 ms.
constexpr uint32_t kLvNearTile = 256;
template <bool kStage>
__global__ __launch_bounds__(64) void dm_lv_nearby(const float *__restrict__ hits, uint32_t nh, LvBeamArgs a, const double *__restrict__ rng,
                                                  const LvBeam *__restrict__ beams, unsigned long long *__restrict__ mask, uint32_t tile) {
    const uint32_t q = blockIdx.x * 64u + threadIdx.x;
    const bool have = q < nh;
    const float qx = have ? hits[3 * (size_t)q] : 0.f, qy = have ? hits[3 * (size_t)q + 1] : 0.f, qz = have ? hits[3 * (size_t)q + 2] : 0.f;
    const double dist2 = have ? rng[q] : 0.0;
    const bool in_range = have && !(a.max_range > 0 && dist2 > a.max_range);
    const bool low = (double)qz < (double)a.oz + a.influence;
    // kStage (small scans): the tile's beams go to LDS first.  Read one by one from memory (a wave-uniform load per beam, each
    // waited for) the walk of a tile was a chain of memory round trips — 120 us for the 3 500-beam scans of configs[3], where a
    // SIMD holds one wave.  Large scans keep the direct loads: thousands of waves hide them, and there the staged form
    // measured slower (2.9 against 2.5 ms at 29 000 beams).
    __shared__ LvBeam s_beam[kStage ? kLvNearTile : 1];
    const uint32_t h0 = blockIdx.y * tile, h1 = min(nh, h0 + tile);
    if (kStage) {
        for (uint32_t h = h0 + threadIdx.x; h < h1; h += 64u) s_beam[h - h0] = beams[h];
        __syncthreads();
    }
#pragma unroll 4
    for (uint32_t h = h0; h < h1; ++h) {
        const LvBeam b = kStage ? s_beam[h - h0] : beams[h];
        bool near = false;
        const bool high = (double)b.pz > (a.offset + (double)a.oz);
        if (in_range && !(high && low)) {
            const double dist1 = lv_norm(b.ex - qx, b.ey - qy, b.ez - qz);
            near = dist1 < a.influence || (dist1 < b.l0 && dist2 < b.l0);
            if (near) {
                // the distance from the hit to the beam's line does not depend on the running length either: only hits
                // within `influence` of the line can shorten the beam, the ordered walk just gates them on b <= l^2
                const float ox = a.ox, oy = a.oy, oz = a.oz;
                const float lvx = b.ex - ox, lvy = b.ey - oy, lvz = b.ez - oz;
                const double lvn = lv_norm(lvx, lvy, lvz);
                const float vx = qx - ox, vy = qy - oy, vz = qz - oz;
                const double bb = (double)(vx * lvx + vy * lvy + vz * lvz);
                const float t = (float)(bb / (lvn * lvn));
                const float mx = ox + lvx * t, my = oy + lvy * t, mz = oz + lvz * t;
                near = lv_norm(qx - mx, qy - my, qz - mz) < a.influence;
            }
        }
        const unsigned long long m = __ballot(near);
        if (threadIdx.x == 0) mask[(size_t)h * gridDim.x + blockIdx.x] = m;
    }
}
__global__ __launch_bounds__(256) void dm_lv_beams_walk(const float *__restrict__ hits, uint32_t nh, LvBeamArgs a,
                                                      const LvBeam *__restrict__ beams, const unsigned long long *__restrict__ mask,
                                                      uint32_t nw, uint8_t *__restrict__ flags, float *__restrict__ seg,
                                                      uint32_t *__restrict__ nsamp, uint32_t *__restrict__ nray, uint32_t *counters,
                                                      int lds_hits) {
    // the walk reads one hit per set bit, each read depending on the previous bit: with the hit list staged in LDS
    // (lds_hits != 0: it fits) that is an LDS latency per step instead of a trip to L2
    extern __shared__ __attribute__((aligned(16))) float lv_walk_hits[];
    if (lds_hits) {   // 16-byte loads, all in flight (one 4-byte load per trip, each waited for, was most of this kernel at 3 500 beams)
        // (hits is the caller's own cloud when ds_resolution < 0 — any 4-byte aligned device pointer: the 16-byte form
        //  only for a 16-byte aligned one)
        const uint32_t n4 = ((uintptr_t)hits & 15u) == 0u ? (3u * nh) / 4u : 0u;
        const float4 *h4 = reinterpret_cast<const float4 *>(hits);
        float4 *l4 = reinterpret_cast<float4 *>(lv_walk_hits);
#pragma unroll 4
        for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) l4[i] = h4[i];
        for (uint32_t i = 4u * n4 + threadIdx.x; i < 3u * nh; i += blockDim.x) lv_walk_hits[i] = hits[i];
        __syncthreads();
    }
    const float *hp = lds_hits ? lv_walk_hits : hits;
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0;
    bool is_hit = false;
    if (h < nh) {
        const LvBeam bm = beams[h];
        const float ox = a.ox, oy = a.oy, oz = a.oz;
        const float nx = bm.nx, ny = bm.ny, nz = bm.nz;
        double l = bm.l0;
        uint32_t fl = bm.fl;
        float npz = bm.pz;
        const float lvx = bm.ex - ox, lvy = bm.ey - oy, lvz = bm.ez - oz;
        const double lvn = lv_norm(lvx, lvy, lvz);
        for (uint32_t w = 0; w < nw; ++w) {
            unsigned long long m = mask[(size_t)h * nw + w];
            while (m) {
                const uint32_t q = 64u * w + (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                const float qx = hp[3 * (size_t)q], qy = hp[3 * (size_t)q + 1], qz = hp[3 * (size_t)q + 2];
                const float vx = qx - ox, vy = qy - oy, vz = qz - oz;
                const double b = (double)(vx * lvx + vy * lvy + vz * lvz);
                if (b > l * l) continue;   // (the distance-to-the-line test is already in the mask)
                npz = qz;
                l = b / lvn;
            }
        }
        if (l < a.max_range / 5.0 && l / (a.offset - (double)npz) > 0) {  // downward rays close to the sensor
            fl |= kLvBeamSkip;
        } else {
            const float fex = (float)(ox + nx * l), fey = (float)(oy + ny * l), fez = (float)(oz + nz * l);
            float fox = fex, foy = fey, foz = fez;
            if (l > a.influence * 1.0) {
                fox = (float)(ox + nx * a.influence * 1.0);
                foy = (float)(oy + ny * a.influence * 1.0);
                foz = (float)(oz + nz * a.influence * 1.0);
            }
            float *s = seg + 6 * (size_t)h;
            s[0] = fox; s[1] = foy; s[2] = foz; s[3] = fex; s[4] = fey; s[5] = fez;
            const float len = (float)sqrt((double)((fex - fox) * (fex - fox) + (fey - foy) * (fey - foy) + (fez - foz) * (fez - foz)));
            c = 1;
            for (float d = len; d > 0.0 && c < kBeamCap; d -= a.free_res) ++c;
            if (c >= kBeamCap) {
                atomicOr(&counters[kCntError], kErrBeam);
                c = 1;
            }
        }
        flags[h] = (uint8_t)fl;
        nray[h] = (fl & kLvBeamSkip) ? 0u : 1u;
        c += (fl & kLvBeamHit) ? 1u : 0u;
        nsamp[h] = c;
        is_hit = (fl & kLvBeamHit) != 0u;
    }
    beam_total_add(c, counters);
    const unsigned long long hm = __ballot(is_hit);
    if ((threadIdx.x & 63) == 0 && hm) atomicAdd(&counters[kCntTrained], (uint32_t)__popcll(hm));
}

// ---- ray shortening in O(N k) (round 6): the "nearby" sets from a uniform grid over the hits ------------------------
// dm_lv_nearby above tests every (beam, hit) pair and keeps an nh x nh bit matrix: O(N^2) time AND memory, as the reference's
// own loop (bgklvoctomap.cpp:313-423) is in time — 2.5 of the 3.85 ms of a 50 k-ray insert, 5 GB of mask at 200 k hits.
// A hit can only be "nearby" AND shorten a beam if it lies within `influence` of the beam's line and (within `influence` of the
// free end point, or nearer than the initial length l0 to both the sensor and that end point) — i.e. inside a capsule of radius
// `influence` around the segment sensor -> end point.  So: the in-range hits go into a uniform grid (cell >= influence), every
// beam (one WAVE) visits the cells its capsule can touch — slab by slab along the beam's dominant axis, a handful of cells per
// slab — and applies EXACTLY dm_lv_nearby's tests to the hits it finds there (lv_near below: the same expressions).  The
// candidates of a beam come out in grid order; the shortening is order dependent (every accepted hit changes the length the
// next ones are gated on), so the wave sorts them by hit index (in LDS) and walks them exactly as dm_lv_beams_walk walks the
// set bits of its mask row: the same hits in the same order, bit-identical segments (tests/test_lv_gpu.py compares the two
// paths and the restatement).  Measured on the way (50 k rays, 43 k hits, per pass): one lane per beam 1.05 ms (~1 000 dependent
// cell look-ups in a row, 672 waves waiting for memory); one wave per beam with lane = slab for the tests too 0.48 ms (the hits
// sit in the few slabs at the beam's end: two to four lanes did all the f64 work); cells queued in LDS and tested lane = hit
// behind an fp32 pre-test 0.25 ms — those three as a count pass + a fill pass + two device-wide radix sorts of the (beam, hit)
// pairs + a walk kernel; everything in one kernel (below) 0.31 ms in all.
struct LvHitGrid {
    double cell;      // edge: influence * 2^k (k > 0 only when the scan's extent needs more than 2^24 cells)
    double g0[3];     // lower corner: cell (i, j, k) = floor((p - g0) / cell)
    int32_t dim[3];
};
// the membership test of one (beam, hit) pair — the expressions of dm_lv_nearby, in its order
__device__ __forceinline__ bool lv_near(const LvBeam &b, const LvBeamArgs &a, float qx, float qy, float qz, double dist2) {
    if (a.max_range > 0 && dist2 > a.max_range) return false;
    const bool low = (double)qz < (double)a.oz + a.influence;
    const bool high = (double)b.pz > (a.offset + (double)a.oz);
    if (high && low) return false;
    const double dist1 = lv_norm(b.ex - qx, b.ey - qy, b.ez - qz);
    if (!(dist1 < a.influence || (dist1 < b.l0 && dist2 < b.l0))) return false;
    const float ox = a.ox, oy = a.oy, oz = a.oz;
    const float lvx = b.ex - ox, lvy = b.ey - oy, lvz = b.ez - oz;
    const double lvn = lv_norm(lvx, lvy, lvz);
    const float vx = qx - ox, vy = qy - oy, vz = qz - oz;
    const double bb = (double)(vx * lvx + vy * lvy + vz * lvz);
    const float t = (float)(bb / (lvn * lvn));
    const float mx = ox + lvx * t, my = oy + lvy * t, mz = oz + lvz * t;
    return lv_norm(qx - mx, qy - my, qz - mz) < a.influence;
}
// cell coordinates (on the finest grid: edge = influence, origin = the sensor) of the hits that can be nearby at all:
// finite and inside max_range; their bounds by one set of atomics per workgroup
__global__ __launch_bounds__(256) void dm_lv_hit_bounds(const float *__restrict__ hits, uint32_t nh, LvBeamArgs a, const double *__restrict__ rng,
                                                       uint32_t *counters) {
    __shared__ int32_t s_lo[4][3], s_hi[4][3];
    __shared__ uint32_t s_n[4];
    int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    uint32_t n_ok = 0;
    const double org[3] = {(double)a.ox, (double)a.oy, (double)a.oz};
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nh; q += gridDim.x * blockDim.x) {
        const float p[3] = {hits[3 * (size_t)q], hits[3 * (size_t)q + 1], hits[3 * (size_t)q + 2]};
        if (!finite3(p[0], p[1], p[2]) || (a.max_range > 0 && rng[q] > a.max_range)) continue;
        bool ok = true;
        int32_t c[3];
        for (int k = 0; k < 3; ++k) {
            const double v = floor(((double)p[k] - org[k]) / a.influence);
            ok &= v >= -1073741824.0 && v <= 1073741824.0;
            c[k] = (int32_t)v;
        }
        if (!ok) {
            atomicOr(&counters[kCntError], kErrLvExtent);
            continue;
        }
        for (int k = 0; k < 3; ++k) {
            lo[k] = min(lo[k], c[k]);
            hi[k] = max(hi[k], c[k]);
        }
        ++n_ok;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = min(lo[k], __shfl_xor(lo[k], o));
            hi[k] = max(hi[k], __shfl_xor(hi[k], o));
        }
        n_ok += __shfl_xor(n_ok, o);
    }
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; ++k) {
            s_lo[wv][k] = lo[k];
            s_hi[wv][k] = hi[k];
        }
        s_n[wv] = n_ok;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (int w = 0; w < 4; ++w) n += s_n[w];
        if (n) {
            int32_t *mm = (int32_t *)(counters + kCntLvHmm);
            for (int k = 0; k < 3; ++k) {
                int32_t l = s_lo[0][k], h = s_hi[0][k];
                for (int w = 1; w < 4; ++w) {
                    l = min(l, s_lo[w][k]);
                    h = max(h, s_hi[w][k]);
                }
                atomicMin(&mm[k], l);
                atomicMax(&mm[3 + k], h);
            }
            atomicAdd(&counters[kCntLvHmm + 6], n);
        }
    }
}
__device__ __forceinline__ uint32_t lv_hit_cell(const LvHitGrid &G, const LvBeamArgs &a, float x, float y, float z, double dist2) {
    if (!finite3(x, y, z) || (a.max_range > 0 && dist2 > a.max_range)) return 0xFFFFFFFFu;
    const long long i = (long long)floor(((double)x - G.g0[0]) / G.cell), j = (long long)floor(((double)y - G.g0[1]) / G.cell),
                    k = (long long)floor(((double)z - G.g0[2]) / G.cell);
    if (i < 0 || j < 0 || k < 0 || i >= G.dim[0] || j >= G.dim[1] || k >= G.dim[2]) return 0xFFFFFFFFu;   // (cannot happen: the grid spans their bounds)
    return (uint32_t)((k * G.dim[1] + j) * G.dim[0] + i);
}
// cell of every hit + the cells' populations (cnt is zero on entry)
__global__ __launch_bounds__(256) void dm_lv_hgrid_count(const float *__restrict__ hits, uint32_t nh, LvBeamArgs a, const double *__restrict__ rng,
                                                        LvHitGrid G, uint32_t *__restrict__ hcell, uint32_t *cnt) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nh) return;
    const uint32_t c = lv_hit_cell(G, a, hits[3 * (size_t)q], hits[3 * (size_t)q + 1], hits[3 * (size_t)q + 2], rng[q]);
    hcell[q] = c;
    if (c != 0xFFFFFFFFu) atomicAdd(&cnt[c], 1u);
}
// the cells' hit lists (any order inside a cell: the pairs are sorted afterwards); fill is zero on entry
__global__ __launch_bounds__(256) void dm_lv_hgrid_fill(const uint32_t *__restrict__ hcell, uint32_t nh, const uint32_t *__restrict__ off,
                                                       uint32_t *fill, uint32_t *__restrict__ list) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nh) return;
    const uint32_t c = hcell[q];
    if (c != 0xFFFFFFFFu) list[off[c] + atomicAdd(&fill[c], 1u)] = q;
}
// The cells a beam's capsule can touch, slab by slab along the dominant axis of the (extended) segment.
struct LvTube {
    double A[3], D[3], w[3], inv;   // extended segment A + t D, t in [0, 1]; extent of the cylinder's cross-section per axis; 1 / D[ax]
    int ax, u, v, i0, i1;           // dominant axis, the other two, slab range (empty: i0 > i1)
};
__device__ __forceinline__ LvTube lv_tube_setup(const LvBeam &b, const LvBeamArgs &a, const LvHitGrid &G) {
    LvTube T;
    T.ax = 0; T.u = 1; T.v = 2; T.i0 = 0; T.i1 = -1; T.inv = 0.0;
    const double R = a.influence * 1.001 + 1e-4;   // (margin: the tests' own roundings, the float end point)
    const double P0[3] = {(double)a.ox, (double)a.oy, (double)a.oz}, P1[3] = {(double)b.ex, (double)b.ey, (double)b.ez};
    double d[3], len2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        d[k] = P1[k] - P0[k];
        len2 += d[k] * d[k];
    }
    for (int k = 0; k < 3; ++k) T.A[k] = T.D[k] = T.w[k] = 0.0;
    if (!(len2 < 1e30)) return T;   // a NaN / infinite beam is near nothing (every comparison of lv_near fails)
    const double len = sqrt(len2);
    double n[3] = {1.0, 0.0, 0.0};
    if (len > 1e-9)
        for (int k = 0; k < 3; ++k) n[k] = d[k] / len;
    // the capsule lies inside the cylinder of radius R around [A, A + D], the segment extended by R at both ends
    for (int k = 0; k < 3; ++k) {
        T.A[k] = P0[k] - n[k] * R;
        T.D[k] = d[k] + 2.0 * n[k] * R;
        T.w[k] = R * sqrt(fmax(0.0, 1.0 - n[k] * n[k]));
    }
    int ax = fabs(T.D[1]) > fabs(T.D[0]) ? 1 : 0;
    if (fabs(T.D[2]) > fabs(T.D[ax])) ax = 2;
    T.ax = ax;
    T.u = ax == 0 ? 1 : 0;
    T.v = ax == 2 ? 1 : 2;
    const double lo_a = fmin(T.A[ax], T.A[ax] + T.D[ax]) - T.w[ax], hi_a = fmax(T.A[ax], T.A[ax] + T.D[ax]) + T.w[ax];
    const double fi0 = floor((lo_a - G.g0[ax]) / G.cell), fi1 = floor((hi_a - G.g0[ax]) / G.cell);
    if (fi1 < 0.0 || fi0 >= (double)G.dim[ax]) return T;
    T.i0 = (int)fmax(fi0, 0.0);
    T.i1 = (int)fmin(fi1, (double)(G.dim[ax] - 1));
    T.inv = 1.0 / T.D[ax];
    return T;
}
// the cells of slab i (index along the dominant axis): visit(cell) once per cell
template <class F>
__device__ __forceinline__ void lv_tube_slab(const LvTube &T, const LvHitGrid &G, int i, F &&visit) {
    const int ax = T.ax, u = T.u, v = T.v;
    const double slo = G.g0[ax] + (double)i * G.cell - T.w[ax], shi = G.g0[ax] + (double)(i + 1) * G.cell + T.w[ax];
    double t0 = (slo - T.A[ax]) * T.inv, t1 = (shi - T.A[ax]) * T.inv;
    if (t0 > t1) {
        const double tt = t0;
        t0 = t1;
        t1 = tt;
    }
    t0 = fmax(t0, 0.0);
    t1 = fmin(t1, 1.0);
    if (t0 > t1) return;
    const double ua = T.A[u] + t0 * T.D[u], ub = T.A[u] + t1 * T.D[u], va = T.A[v] + t0 * T.D[v], vb = T.A[v] + t1 * T.D[v];
    const double fj0 = floor((fmin(ua, ub) - T.w[u] - G.g0[u]) / G.cell), fj1 = floor((fmax(ua, ub) + T.w[u] - G.g0[u]) / G.cell);
    const double fk0 = floor((fmin(va, vb) - T.w[v] - G.g0[v]) / G.cell), fk1 = floor((fmax(va, vb) + T.w[v] - G.g0[v]) / G.cell);
    if (fj1 < 0.0 || fj0 >= (double)G.dim[u] || fk1 < 0.0 || fk0 >= (double)G.dim[v]) return;
    const int j0 = (int)fmax(fj0, 0.0), j1 = (int)fmin(fj1, (double)(G.dim[u] - 1));
    const int k0 = (int)fmax(fk0, 0.0), k1 = (int)fmin(fk1, (double)(G.dim[v] - 1));
    const int stride[3] = {1, G.dim[0], G.dim[0] * G.dim[1]};
    for (int k = k0; k <= k1; ++k)
        for (int j = j0; j <= j1; ++j) visit((uint32_t)(i * stride[ax] + j * stride[u] + k * stride[v]));
}
constexpr uint32_t kLvCellQ = 192;   // queued cells per beam and round of 64 slabs; cells beyond that are tested by the lane that found them
// ---- the whole beam in ONE kernel (round 6, third form): capsule walk -> the beam's nearby hits collected in LDS -> sorted by
// hit index in LDS -> the ordered shortening -> segment + sample count.  No pair arrays, no device-wide sorts, no host read-back of
// a pair count.  One wave per beam (its LDS is its own: only wave-level barriers).  A beam whose nearby set does not fit the LDS
// buffer is walked in windows of the hit index (the shortening only needs the hits in ascending order: window after window
// is the same sequence); cells beyond the queue's capacity are tested by the lane that found them.
constexpr uint32_t kLvMatchCap = 1024;   // nearby hits per beam and window held in LDS
__device__ __forceinline__ void lv_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr uint32_t kLvGridWaves = 2;   // beams per workgroup: 2 x (cell queue 1.5 KB + match buffer 4 KB)
__global__ __launch_bounds__(64 * kLvGridWaves) void dm_lv_beams_grid(const float *__restrict__ hits, uint32_t nh, LvBeamArgs a, const double *__restrict__ rng,
                                                                     const LvBeam *__restrict__ beams, LvHitGrid G, const uint32_t *__restrict__ cell_off,
                                                                     const uint32_t *__restrict__ cell_list, uint8_t *__restrict__ flags, float *__restrict__ seg,
                                                                     uint32_t *__restrict__ nsamp, uint32_t *__restrict__ nray, uint32_t *counters) {
    __shared__ uint32_t s_n[kLvGridWaves], s_nc[kLvGridWaves];
    __shared__ uint2 s_cells[kLvGridWaves][kLvCellQ];
    __shared__ uint32_t s_m[kLvGridWaves][kLvMatchCap];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t h = blockIdx.x * kLvGridWaves + wv;
    if (h >= nh) return;   // (no workgroup barrier below)
    const LvBeam b = beams[h];
    const LvTube T = lv_tube_setup(b, a, G);
    float ux = 0.f, uy = 0.f, uz = 0.f;
    {
        const double dl = sqrt(T.D[0] * T.D[0] + T.D[1] * T.D[1] + T.D[2] * T.D[2]);
        if (dl > 0.0) {
            ux = (float)(T.D[0] / dl);
            uy = (float)(T.D[1] / dl);
            uz = (float)(T.D[2] / dl);
        }
    }
    const float r_pre = (float)(a.influence * 1.01 + 1e-3);
    const float r_pre2 = r_pre * r_pre;
    const float vv_max = (r_pre2 - (float)(a.influence * a.influence)) * 1e6f;   // (see dm_lv_near_grid)
    const float ox = a.ox, oy = a.oy, oz = a.oz;
    const float lvx = b.ex - ox, lvy = b.ey - oy, lvz = b.ez - oz;
    const double lvn = lv_norm(lvx, lvy, lvz);
    double l = b.l0;
    float npz = b.pz;
    uint32_t qlo = 0, win = nh;
    while (qlo < nh) {
        const uint32_t qhi = (nh - qlo <= win) ? nh : qlo + win;
        if (lane == 0) s_n[wv] = 0u;
        lv_wave_sync();
        auto test = [&](uint32_t q) {
            if (q < qlo || q >= qhi) return;
            const float qx = hits[3 * (size_t)q], qy = hits[3 * (size_t)q + 1], qz = hits[3 * (size_t)q + 2];
            const float vx = qx - ox, vy = qy - oy, vz = qz - oz;
            const float along = vx * ux + vy * uy + vz * uz, vv = vx * vx + vy * vy + vz * vz;
            if (vv < vv_max && vv - along * along > r_pre2) return;
            if (lv_near(b, a, qx, qy, qz, rng[q])) {
                const uint32_t pos = atomicAdd(&s_n[wv], 1u);
                if (pos < kLvMatchCap) s_m[wv][pos] = q;
            }
        };
        for (int r0 = T.i0; r0 <= T.i1; r0 += 64) {
            if (lane == 0) s_nc[wv] = 0u;
            lv_wave_sync();
            const int i = r0 + (int)lane;
            if (i <= T.i1)
                lv_tube_slab(T, G, i, [&](uint32_t c) {
                    const uint32_t p0 = cell_off[c], p1 = cell_off[c + 1];
                    if (p1 == p0) return;
                    const uint32_t slot = atomicAdd(&s_nc[wv], 1u);
                    if (slot < kLvCellQ) s_cells[wv][slot] = make_uint2(p0, p1);
                    else
                        for (uint32_t p = p0; p < p1; ++p) test(cell_list[p]);
                });
            lv_wave_sync();
            const uint32_t nc = min(s_nc[wv], kLvCellQ);
            for (uint32_t ci = 0; ci < nc; ++ci) {
                const uint2 r = s_cells[wv][ci];
                for (uint32_t p = r.x + lane; p < r.y; p += 64u) test(cell_list[p]);
            }
            lv_wave_sync();
        }
        const uint32_t n = s_n[wv];
        if (n > kLvMatchCap) {   // too many for the buffer: the same hits, a narrower window of the hit index
            win = max(1u, (qhi - qlo) >> 1);
            continue;
        }
        if (n > 1u) {   // bitonic sort of s_m[0, n) (ascending), padded to a power of two with the largest key
            uint32_t M = 64u;
            while (M < n) M <<= 1;
            for (uint32_t idx = n + lane; idx < M; idx += 64u) s_m[wv][idx] = 0xFFFFFFFFu;
            lv_wave_sync();
            for (uint32_t k = 2u; k <= M; k <<= 1)
                for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
                    for (uint32_t t = lane; t < (M >> 1); t += 64u) {
                        const uint32_t i0 = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), i1 = i0 + j;
                        const uint32_t x0 = s_m[wv][i0], x1 = s_m[wv][i1];
                        const bool up = (i0 & k) == 0u;
                        if ((x0 > x1) == up) {
                            s_m[wv][i0] = x1;
                            s_m[wv][i1] = x0;
                        }
                    }
                    lv_wave_sync();
                }
        }
        // the ordered walk: a hit's projection term does not depend on the running length, so 64 of them are formed at once
        for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
            const uint32_t m = min(64u, n - c0);
            double bq = 0.0;
            float qzl = 0.f;
            if (lane < m) {
                const uint32_t q = s_m[wv][c0 + lane];
                const float qx = hits[3 * (size_t)q], qy = hits[3 * (size_t)q + 1], qz = hits[3 * (size_t)q + 2];
                const float vx = qx - ox, vy = qy - oy, vz = qz - oz;
                bq = (double)(vx * lvx + vy * lvy + vz * lvz);
                qzl = qz;
            }
            for (uint32_t i = 0; i < m; ++i) {
                const double bi = __shfl(bq, (int)i);
                const float qzi = __shfl(qzl, (int)i);
                if (bi > l * l) continue;   // (the distance-to-the-line test is already in the list)
                npz = qzi;
                l = bi / lvn;
            }
        }
        lv_wave_sync();
        qlo = qhi;
    }
    if (lane != 0) return;
    uint32_t fl = b.fl, c = 0;
    const float nx = b.nx, ny = b.ny, nz = b.nz;
    if (l < a.max_range / 5.0 && l / (a.offset - (double)npz) > 0) {  // downward rays close to the sensor
        fl |= kLvBeamSkip;
    } else {
        const float fex = (float)(ox + nx * l), fey = (float)(oy + ny * l), fez = (float)(oz + nz * l);
        float fox = fex, foy = fey, foz = fez;
        if (l > a.influence * 1.0) {
            fox = (float)(ox + nx * a.influence * 1.0);
            foy = (float)(oy + ny * a.influence * 1.0);
            foz = (float)(oz + nz * a.influence * 1.0);
        }
        float *s = seg + 6 * (size_t)h;
        s[0] = fox; s[1] = foy; s[2] = foz; s[3] = fex; s[4] = fey; s[5] = fez;
        const float len = (float)sqrt((double)((fex - fox) * (fex - fox) + (fey - foy) * (fey - foy) + (fez - foz) * (fez - foz)));
        c = 1;
        for (float d = len; d > 0.0 && c < kBeamCap; d -= a.free_res) ++c;
        if (c >= kBeamCap) {
            atomicOr(&counters[kCntError], kErrBeam);
            c = 1;
        }
    }
    flags[h] = (uint8_t)fl;
    nray[h] = (fl & kLvBeamSkip) ? 0u : 1u;
    c += (fl & kLvBeamHit) ? 1u : 0u;
    nsamp[h] = c;
}
// the counters the walk kernels keep (64-bit sample total, hit samples), for dm_lv_beams_grid — one set of atomics per workgroup
__global__ __launch_bounds__(256) void dm_lv_beam_totals(const uint32_t *__restrict__ nsamp, const uint8_t *__restrict__ flags, uint32_t nh, uint32_t *counters) {
    __shared__ unsigned long long s_t[4];
    __shared__ uint32_t s_h[4];
    unsigned long long t = 0;
    uint32_t hc = 0;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nh; h += gridDim.x * blockDim.x) {
        t += nsamp[h];
        hc += (flags[h] & kLvBeamHit) ? 1u : 0u;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        t += __shfl_xor(t, o);
        hc += __shfl_xor(hc, o);
    }
    if ((threadIdx.x & 63) == 0) {
        s_t[threadIdx.x >> 6] = t;
        s_h[threadIdx.x >> 6] = hc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long all = 0;
        uint32_t hh = 0;
        for (int w = 0; w < 4; ++w) {
            all += s_t[w];
            hh += s_h[w];
        }
        if (all) atomicAdd(reinterpret_cast<unsigned long long *>(counters + kCntBeamTotal), all);
        if (hh) atomicAdd(&counters[kCntTrained], hh);
    }
}

// samples {x, y, z, ray as float (-1 = hit)} and segments {start, first sample index bits | end, 0} in beam order
__global__ __launch_bounds__(256) void dm_lv_emit(const float *__restrict__ hits, uint32_t nh, LvBeamArgs a,
                                                 const uint8_t *__restrict__ flags, const float *__restrict__ seg,
                                                 const uint32_t *__restrict__ samp_off, const uint32_t *__restrict__ ray_off,
                                                 float4 *__restrict__ samples, float4 *__restrict__ rays) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nh) return;
    const uint32_t fl = flags[h];
    size_t w = samp_off[h];
    if (fl & kLvBeamHit) samples[w++] = make_float4(hits[3 * (size_t)h], hits[3 * (size_t)h + 1], hits[3 * (size_t)h + 2], -1.0f);
    if (fl & kLvBeamSkip) return;
    const float *s = seg + 6 * (size_t)h;
    const float x0 = s[0], y0 = s[1], z0 = s[2], x = s[3], y = s[4], z = s[5];
    const uint32_t ray = ray_off[h];
    const float rf = (float)ray;
    const uint32_t first = (uint32_t)w;
    samples[w++] = make_float4(x0, y0, z0, rf);
    const float len = (float)sqrt((double)((x - x0) * (x - x0) + (y - y0) * (y - y0) + (z - z0) * (z - z0)));
    const float ux = (x - x0) / len, uy = (y - y0) / len, uz = (z - z0) / len;
    uint32_t c = 1;
    for (float d = len; d > 0.0 && c < kBeamCap; d -= a.free_res, ++c) samples[w++] = make_float4(x0 + ux * d, y0 + uy * d, z0 + uz * d, rf);
    rays[2 * (size_t)ray] = make_float4(x0, y0, z0, __uint_as_float(first));
    rays[2 * (size_t)ray + 1] = make_float4(x, y, z, 0.0f);
}

// ---- gather grid ---------------------------------------------------------------------------------------------
struct LvGridArgs {
    double g, half;          // bucket edge, block_size / 2
    int32_t cmin[3], cdim[3];
};
__device__ __forceinline__ long long lv_cidx(float v, double half, double g) { return (long long)floor(((double)v + half) / g); }

// mm[0..2] = min, mm[3..5] = max bucket coordinate over the finite samples (int32 range; error bit 4 otherwise), mm[6] = their count
__global__ __launch_bounds__(256) void dm_lv_cell_bounds(const float4 *__restrict__ samples, uint32_t ns, double half, double g,
                                                        int32_t *mm, uint32_t *counters) {
    __shared__ int32_t s_lo[4][3], s_hi[4][3];
    __shared__ uint32_t s_n[4];
    int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    uint32_t n_ok = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const float4 s = samples[i];
        if (isfinite(s.x) && isfinite(s.y) && isfinite(s.z)) {
            const long long c[3] = {lv_cidx(s.x, half, g), lv_cidx(s.y, half, g), lv_cidx(s.z, half, g)};
            bool ok = true;
            for (int a = 0; a < 3; ++a) ok &= c[a] >= -(1ll << 30) && c[a] <= (1ll << 30);
            if (!ok) atomicOr(&counters[kCntError], kErrLvExtent);
            else {
                for (int a = 0; a < 3; ++a) {
                    lo[a] = min(lo[a], (int32_t)c[a]);
                    hi[a] = max(hi[a], (int32_t)c[a]);
                }
                ++n_ok;
            }
        }
    }
    // one set of atomics per workgroup, at most 128 workgroups (an atomic on one address costs ~25 ns per caller, serialised)
    for (int o = 32; o >= 1; o >>= 1) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], o));
            hi[a] = max(hi[a], __shfl_xor(hi[a], o));
        }
        n_ok += __shfl_xor(n_ok, o);
    }
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; ++a) {
            s_lo[wv][a] = lo[a];
            s_hi[wv][a] = hi[a];
        }
        s_n[wv] = n_ok;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (int w = 0; w < 4; ++w) n += s_n[w];
        if (n) {
            for (int a = 0; a < 3; ++a) {
                int32_t l = s_lo[0][a], h = s_hi[0][a];
                for (int w = 1; w < 4; ++w) {
                    l = min(l, s_lo[w][a]);
                    h = max(h, s_hi[w][a]);
                }
                atomicMin(&mm[a], l);
                atomicMax(&mm[3 + a], h);
            }
            atomicAdd((uint32_t *)&mm[6], n);
        }
    }
}
// key = linear bucket index (x fastest), ncell for a sample that is not binned (non-finite: it keeps its place in
// `samples` — the rays refer to sample indices — but lies in no voxel's box)
__global__ __launch_bounds__(256) void dm_lv_cell_keys(const float4 *__restrict__ samples, uint32_t ns, LvGridArgs ga, uint32_t ncell,
                                                      uint32_t *__restrict__ key, uint32_t *__restrict__ val) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const float4 s = samples[i];
    uint32_t k = ncell;
    if (isfinite(s.x) && isfinite(s.y) && isfinite(s.z)) {
        const long long cx = lv_cidx(s.x, ga.half, ga.g) - ga.cmin[0], cy = lv_cidx(s.y, ga.half, ga.g) - ga.cmin[1],
                        cz = lv_cidx(s.z, ga.half, ga.g) - ga.cmin[2];
        k = (uint32_t)((cz * ga.cdim[1] + cy) * ga.cdim[0] + cx);
    }
    key[i] = k;
    val[i] = i;
}
__global__ __launch_bounds__(256) void dm_lv_sorted(const float4 *__restrict__ samples, const uint32_t *__restrict__ order, uint32_t ns,
                                                   float4 *__restrict__ sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const uint32_t o = order[i];
    const float4 s = samples[o];
    sorted[i] = make_float4(s.x, s.y, s.z, __uint_as_float(o));
}
// cell_off[c] = first sorted position with key >= c, for c = 0 .. ncell (CSR over the dense grid)
__global__ __launch_bounds__(256) void dm_lv_cell_off(const uint32_t *__restrict__ sorted_key, uint32_t ns, uint32_t ncell,
                                                     uint32_t *__restrict__ cell_off) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncell) return;
    uint32_t lo = 0, hi = ns;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sorted_key[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    cell_off[c] = lo;
}

// ---- candidate blocks ------------------------------------------------------------------------------------------
// The float-stepped triple loop (:105-117) produces the product of three per-axis index sequences; duplicates of an
// index multiply the key's multiplicity.  ax holds, per axis, the distinct indices ascending (so the keys come out in
// ascending order, like the host's std::sort) and how often each occurs.
struct LvCandArgs {
    const int32_t *idx[3];
    const uint32_t *mult[3];
    uint32_t n[3];
    float block_size;
    double g;
    int32_t reach, bpb;      // buckets per block edge
    int32_t cmin[3], cdim[3];
};
__global__ __launch_bounds__(256) void dm_lv_candidates(LvCandArgs a, uint32_t n_cand, const uint32_t *__restrict__ cell_off,
                                                       long long *__restrict__ keys, uint32_t *__restrict__ mult,
                                                       uint32_t *__restrict__ flag, uint32_t *counters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_cand) return;
    if (t == 0) counters[kCntTest] = n_cand;   // dm_table_insert takes its key count from there
    const uint32_t iz = t % a.n[2], iy = (t / a.n[2]) % a.n[1], ix = t / (a.n[2] * a.n[1]);
    const long long kx = a.idx[0][ix], ky = a.idx[1][iy], kz = a.idx[2][iz];
    keys[t] = (kx << 40) | (ky << 20) | kz;
    mult[t] = a.mult[0][ix] * a.mult[1][iy] * a.mult[2][iz];
    // centre (hash_key_to_block: int64 -> float, float multiply) and the block's lowest bucket
    const float c[3] = {(float)(kx - 524288) * a.block_size, (float)(ky - 524288) * a.block_size, (float)(kz - 524288) * a.block_size};
    long long lo[3], hi[3];
    bool any = true;
    for (int d = 0; d < 3; ++d) {
        const long long b0 = llround((double)c[d] / a.g);
        lo[d] = max(b0 - a.reach, (long long)a.cmin[d]);
        hi[d] = min(b0 + a.bpb - 1 + a.reach, (long long)a.cmin[d] + a.cdim[d] - 1);
        any &= lo[d] <= hi[d];
    }
    bool found = false;
    if (any)
        for (long long z = lo[2]; z <= hi[2] && !found; ++z)
            for (long long y = lo[1]; y <= hi[1]; ++y) {
                const size_t row = (size_t)(((z - a.cmin[2]) * a.cdim[1] + (y - a.cmin[1])) * a.cdim[0]);
                if (cell_off[row + (size_t)(hi[0] - a.cmin[0]) + 1] != cell_off[row + (size_t)(lo[0] - a.cmin[0])]) {
                    found = true;
                    break;
                }
            }
    flag[t] = found ? 1u : 0u;
}
// packed blocks in key order: centre, lowest bucket, pool slot, multiplicity
__global__ __launch_bounds__(256) void dm_lv_pack(const long long *__restrict__ keys, const uint32_t *__restrict__ mult,
                                                 const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                                 const uint32_t *__restrict__ slot, uint32_t n_cand, float block_size, double g,
                                                 float *__restrict__ center, int32_t *__restrict__ cell0, uint32_t *__restrict__ p_slot,
                                                 uint32_t *__restrict__ p_mult, uint32_t *counters, uint32_t *__restrict__ info) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_cand) return;
    info[t] = 0u;   // "had information" flags of the packed blocks (at most n_cand of them)
    if (t == n_cand - 1) counters[kCntTest] = pos[t] + flag[t];
    if (!flag[t]) return;
    const uint32_t w = pos[t];
    const long long k = keys[t];
    const long long kk[3] = {k >> 40, (k >> 20) & 0xFFFFF, k & 0xFFFFF};
    for (int d = 0; d < 3; ++d) {
        const float c = (float)(kk[d] - 524288) * block_size;
        center[3 * (size_t)w + d] = c;
        cell0[3 * (size_t)w + d] = (int32_t)llround((double)c / g);
    }
    p_slot[w] = slot[t];
    p_mult[w] = mult[t];
}

// after a pass: a packed block "had information" if one of its finest-layer voxels saw a sample (bit 6, set by the
// voxel kernel); the bit is cleared again.  One wave per packed block.
__global__ __launch_bounds__(256) void dm_lv_finish(const uint32_t *__restrict__ p_slot, const uint32_t *__restrict__ p_mult,
                                                   uint32_t n_packed, uint32_t pass, uint8_t *S, uint32_t npb, uint32_t layer_off,
                                                   uint32_t layer_n, uint32_t *__restrict__ info) {
    const uint32_t b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= n_packed || p_mult[b] <= pass) return;
    uint8_t *s = S + (size_t)p_slot[b] * npb + layer_off;
    bool any = false;
    for (uint32_t i = lane; i < layer_n; i += 64) {
        const uint8_t v = s[i];
        if (v & 0x40u) {
            any = true;
            s[i] = (uint8_t)(v & ~0x40u);
        }
    }
    if (__any(any) && lane == 0) info[b] = 1u;
}
// slot list for dm_prune: the blocks with information, 0xFFFFFFFF (skipped there) for the others
__global__ __launch_bounds__(256) void dm_lv_prune_list(const uint32_t *__restrict__ p_slot, const uint32_t *__restrict__ info,
                                                       uint32_t n_packed, uint32_t *__restrict__ list, uint32_t *counters) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_packed) return;
    const bool i = info[b] != 0u;
    list[b] = i ? p_slot[b] : 0xFFFFFFFFu;
    const unsigned long long m = __ballot(i);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counters[kCntGeo], (uint32_t)__popcll(m));
}

}  // namespace la3dm_dev
