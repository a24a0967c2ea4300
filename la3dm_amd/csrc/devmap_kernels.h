// devmap_kernels.h — HIP kernels (gfx950, wave64) of the device-resident map: the stages of
// BGKOctoMap::insert_pointcloud on either side of the predict/fuse kernel.
//
// Reference behaviour followed (file:line relative to RobustFieldAutonomyLab/la3dm); every stage is
// bit-identical to the host implementation in host/bgkoctomap.cpp, which the parity tests pin to the oracle:
//   f1  get_training_data / beam_sample / downsample   src/bgkoctomap/bgkoctomap.cpp:383-458
//       (pcl::VoxelGrid semantics: cell = floor(p * inv_leaf) - floor(min * inv_leaf), cells in ascending
//        linear index, centroid = fp32 sum in cloud order / count)
//   f2  bbox / get_blocks_in_bbox / closed-box gather   src/bgkoctomap/bgkoctomap.cpp:234-284, 464-552,
//       include/common/rtree.h:1519-1532; hashing src/bgkoctomap/bgkblock.cpp:73-130
//   f3  leaf enumeration (LeafIterator order)            include/bgkoctomap/bgkoctree.h:62-147
//       OcTree::prune                                    src/bgkoctomap/bgkoctree.cpp:101-148
//       node write-back (Occupancy::update results)      src/bgkoctomap/bgkoctree_node.cpp:31-44
//
// These are HBM/latency-bound integer and gather kernels (no MFMA work here): coalesced streams where the
// data allows it, ballot/mbcnt compaction inside a wave, device-wide sorts and scans from rocPRIM/hipCUB.
// Floating-point expressions follow the host code operation for operation (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace la3dm_dev {

constexpr uint32_t kInvalidCell = 0xFFFFFFFFu;
constexpr uint8_t kStatePruned = 3;   // State::PRUNED
constexpr uint8_t kStateUnknown = 2;  // State::UNKNOWN
constexpr uint8_t kClassifiedBit = 0x80;

// (8^d - 1) / 7: first node of layer d in the depth-major node order of a block
__device__ __forceinline__ uint32_t dm_layer_base(uint32_t depth) { return 0x249249u & ((1u << (3u * depth)) - 1u); }

// counters[] slots (device u32 array mirrored into pinned host memory)
enum DevCounter {
    kCntGridSegs = 0,    // occupied voxel-grid cells of the current filter call
    kCntGridValid = 1,   // finite points of the current filter call
    kCntKept = 2,        // hits that passed the range gate
    kCntFreeRaw = 3,     // beam samples before the second filter
    kCntMembers = 4,     // (block, point) membership pairs
    kCntGeo = 5,         // blocks that geometrically hold points
    kCntTest = 6,        // test blocks of the current pass
    kCntLeaves = 7,      // leaves of the current pass (U)
    kCntBlocks = 8,      // blocks in the pool
    kCntError = 9,       // sticky error bits
    kCntTrainReads = 10, // low 32 bits of sum of neighbourhood sizes
    kCntTrainReadsHi = 11,
    kCntBig = 12,        // voxel-grid cells with more than kBigCell points
    kCntTrained = 13,    // blocks with training points that are in the candidate list
    kCntPairEvals = 14,  // 64-bit (words 14, 15): sum of neighbourhood points x leaves; kCntTrainReads likewise (10, 11)
    kCntBeamTotal = 16,  // 64-bit (words 16, 17): beam samples of the scan, summed without the 32-bit wrap of the offsets
    kCntLvPlan = 44,     // BGK-LV work plan (4 words): workgroups of split cubes, scratch rows, split cubes, other workgroups
    kCntGrid = 21,       // GridParams of the last voxel-filter call (10 words), for the host
    kCntBbox = 31,       // training-set box (6 floats as bits), for the host
    kCntLvmm = 37,       // BGK-LV: bucket bounds of the finite samples (3 min, 3 max as int32) + their count
    kCntLvHmm = 48,      // BGK-LV: cell bounds of the in-range hits on the ray-shortening grid (3 min, 3 max as int32) + their count
    kCntWords = 56       // (< 64: one lane per word in dm_publish_wave, word kCntWords of the mailbox is the sequence number)
};

struct GridParams {  // pcl::VoxelGrid bookkeeping of one filter call
    int lo[3];
    int span[3];
    int m1, m2;
    int passthrough;  // 1: index space overflows int32 -> PCL hands the input back
    int empty;        // 1: no finite point
};

// order-preserving float <-> uint32 map for atomicMin/atomicMax
__device__ __forceinline__ uint32_t enc_f32(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}

__device__ __forceinline__ bool finite3(float x, float y, float z) {
    return isfinite(x) && isfinite(y) && isfinite(z);
}

// ---- min / max of the finite points; mm[0..2] = enc(min), mm[3..5] = enc(max) --------------------------
// What the LAST workgroup of a min/max launch does with the result (it then resets mm and the arrival counter, so that
// neither a reset launch before nor a one-thread launch after the reduction is needed):
//   mode 1: the voxel filter's GridParams (-> gp for the kernels, counters[kCntGrid..] for the host)
//   mode 2: the training set's box (-> counters[kCntBbox..]).  The reference (bbox(), bgkoctomap.cpp:464-484) reduces
//           with `<`, which a NaN never wins: NaN coordinates are ignored — except in the FIRST training point, which
//           seeds the reduction and then never loses (restated as a min / max chain from xy[0]): that axis' limits stay
//           NaN, get_blocks_in_bbox makes no step and the scan is a no-op.  `first` = xy[0] (x, y, z, label).
struct MinmaxFin {
    int mode;
    float inv;
    GridParams *gp;
    const float *first;
    uint32_t *counters;
    uint32_t *done;      // zero between launches
    uint32_t *mm_extra;  // mode 2: a second encoded box to unite with (and reset), or nullptr
    volatile uint32_t *mailbox;   // mode 2: the finishing thread also publishes the counter block (dm_publish_lane), or nullptr
    uint32_t mailbox_seq;
};
__device__ void grid_params_from(const uint32_t *mmv, float inv, GridParams &g);
__device__ __forceinline__ void box_add(float mn[3], float mx[3], float x, float y, float z) {   // dm_minmax's rule: finite points only
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) return;
    mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
    mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
}

// The counter block to the host's mailbox from the first wave of a workgroup that has just written the last counters the host is
// waiting for (instead of a dm_publish_counters launch behind it: a dependent dispatch costs ~4.7 us whatever it does).
// What earlier kernels wrote is visible at kernel start, what this workgroup wrote before its last barrier too; counters that
// other threads of the same launch are still writing (error bits of a look-back loop, a zeroed slot) are not waited for
// — the host reads those after a later, real publish.
// The mailbox is pinned host memory (uncached on the device side): its words must be on their way before the sequence
// number, nothing else has to be — a system-scope release fence here would also write back every dirty line the kernel
// left in the L2 (measured in dm_append_frees, right behind tens of MB of samples: 13 -> 36 us).  Stores of a wave
// retire through vmcnt on gfx9, and writes to the host leave through one ordered path.
__device__ __forceinline__ void dm_mailbox_order() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void dm_publish_wave(const uint32_t *counters, volatile uint32_t *mailbox, uint32_t seq) {
    if (threadIdx.x >= 64u) return;   // the first wave of the workgroup: one word per lane (one coalesced write to the host;
                                      // word by word from one thread the 48 writes took 25 us)
    if (threadIdx.x < (uint32_t)kCntWords)
        mailbox[threadIdx.x] = __hip_atomic_load(&counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dm_mailbox_order();
    if (threadIdx.x == 0) mailbox[kCntWords] = seq;
}

// Workgroup part of a min/max reduction (256 threads, every thread of the workgroup calls it): per-thread min/max ->
// one atomic per value per workgroup on mm -> (fin.mode != 0) the last workgroup of the launch finishes the result.
// red: [4][6] floats of LDS, s_last: one word of LDS.
__device__ __forceinline__ void minmax_wg(float mn[3], float mx[3], uint32_t *mm, const MinmaxFin &fin, float (*red)[6],
                                          uint32_t *s_last) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o >= 1; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    }
    const int wv = threadIdx.x >> 6;
    __syncthreads();   // (red may still be read from an earlier call)
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[wv][a] = mn[a];
            red[wv][3 + a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {  // one atomic per value per workgroup; an all-rejected workgroup contributes +-inf = no-op
        const int a = threadIdx.x;
        float v = red[0][a];
        for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, red[w][a]) : fmaxf(v, red[w][a]);
        // (returning atomics: the arrival count below must not overtake them; no fence — a release fence writes the
        // whole L2 back, which doubled this kernel's time after a producer of tens of MB)
        uint32_t prev = 0;
        if (a < 3) {
            if (v != INFINITY) prev = atomicMin(&mm[a], enc_f32(v));
        } else {
            if (v != -INFINITY) prev = atomicMax(&mm[a], enc_f32(v));
        }
        asm volatile("" ::"v"(prev));
    }
    if (fin.mode == 0) return;
    __syncthreads();
    if (threadIdx.x == 0) *s_last = atomicAdd(fin.done, 1u) + 1u == gridDim.x ? 1u : 0u;
    __syncthreads();
    if (!*s_last) return;
    if (threadIdx.x == 0) {
    uint32_t mmv[6];
    for (int a = 0; a < 6; ++a) {
        mmv[a] = __hip_atomic_load(&mm[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mm[a] = a < 3 ? 0xFFFFFFFFu : 0u;
    }
    if (fin.mm_extra) {   // a second box, complete since an earlier launch: the union of the two
        for (int a = 0; a < 6; ++a) {
            const uint32_t e = fin.mm_extra[a];
            mmv[a] = a < 3 ? min(mmv[a], e) : max(mmv[a], e);
            fin.mm_extra[a] = a < 3 ? 0xFFFFFFFFu : 0u;
        }
    }
    *fin.done = 0u;
    if (fin.mode == 1) {
        GridParams g;
        grid_params_from(mmv, fin.inv, g);
        *fin.gp = g;
        const int *gi = (const int *)&g;
        for (int k = 0; k < (int)(sizeof(GridParams) / 4); ++k) fin.counters[kCntGrid + k] = (uint32_t)gi[k];
    } else {
        for (int a = 0; a < 6; ++a) {
            const float f = fin.first[a % 3];
            fin.counters[kCntBbox + a] = __float_as_uint(f != f ? f : dec_f32(mmv[a]));
        }
    }
    }
    if (fin.mode == 2 && fin.mailbox) {   // (uniform over the workgroup) the counter block to the host, one word per lane
        __syncthreads();
        dm_publish_wave(fin.counters, fin.mailbox, fin.mailbox_seq);
    }
}

template <int kStride>
__global__ __launch_bounds__(256) void dm_minmax(const float *__restrict__ p, uint32_t n, uint32_t *mm, MinmaxFin fin) {
    __shared__ float red[4][6];
    __shared__ uint32_t s_last;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    // eight points per thread and trip, their loads in flight together: a trip per point is a round trip to memory per point
    // (BGK-L's 2.9 M samples on 128 workgroups: 88 dependent trips, 37 us)
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 8u * stride) {
        float q[8][3];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t ij = i + j * stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) q[j][c] = ij < n ? p[(size_t)kStride * ij + c] : NAN;   // (box_add skips a NaN point)
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) box_add(mn, mx, q[j][0], q[j][1], q[j][2]);
    }
    minmax_wg(mn, mx, mm, fin, red, &s_last);
}

__device__ void grid_params_from(const uint32_t *mm, float inv, GridParams &g) {
    g.passthrough = 0;
    g.empty = mm[0] == 0xFFFFFFFFu && mm[3] == 0u;
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) {
        mn[a] = dec_f32(mm[a]);
        mx[a] = dec_f32(mm[3 + a]);
    }
    if (g.empty) {
        for (int a = 0; a < 3; ++a) g.lo[a] = 0, g.span[a] = 1;
        g.m1 = g.m2 = 1;
        return;
    }
    const long long ex = (long long)((mx[0] - mn[0]) * inv) + 1, ey = (long long)((mx[1] - mn[1]) * inv) + 1,
                    ez = (long long)((mx[2] - mn[2]) * inv) + 1;
    if (ex * ey * ez > 2147483647ll) g.passthrough = 1;
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = (int)floorf(mn[a] * inv);
        g.span[a] = (int)floorf(mx[a] * inv) - g.lo[a] + 1;
    }
    g.m1 = g.span[0];
    g.m2 = g.span[0] * g.span[1];
}

// Sort key of a test block: heaviest first only balances the predict launch (the blocks are independent, the leaves are
// committed by node index), so the order need not be exact — one radix pass on a weight class in the top byte (16 points
// per class, everything above 4 080 points in class 255) instead of four on the weight; the weight rides in the low bits.
__host__ __device__ __forceinline__ uint32_t test_key(uint32_t weight) {
    const uint32_t cls = weight >> 4 > 255u ? 255u : weight >> 4;
    return ((255u - cls) << 24) | (weight > 0xFFFFFFu ? 0xFFFFFFu : weight);
}
__host__ __device__ __forceinline__ uint32_t test_key_weight(uint32_t key) { return key & 0xFFFFFFu; }

// Stable ascending sort of n <= 4096 (key, value) pairs in ONE workgroup: a bitonic network over (key << 32 | position)
// in LDS (the library sort is 7-9 dependent launches whatever n is; a pass's test-block list has a few hundred to a
// few thousand entries).  N = n rounded up to a power of two.
constexpr uint32_t kSortSmallMax = 4096;
__global__ __launch_bounds__(1024) void dm_sort_small(const uint32_t *__restrict__ k_in, const uint32_t *__restrict__ v_in, uint32_t n,
                                                     uint32_t N, uint32_t *__restrict__ k_out, uint32_t *__restrict__ v_out) {
    __shared__ unsigned long long e[kSortSmallMax];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < N; i += 1024) e[i] = i < n ? ((unsigned long long)k_in[i] << 32) | i : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < N; i += 1024) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long x = e[i], y = e[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) {
                        e[i] = y;
                        e[l] = x;
                    }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = tid; i < n; i += 1024) {
        const unsigned long long x = e[i];
        k_out[i] = (uint32_t)(x >> 32);
        v_out[i] = v_in[(uint32_t)x];
    }
}

// start of an insert: the counter block is zero except the pool's block count, the min/max words are at their identities
__device__ __forceinline__ uint32_t counter_begin_value(uint32_t i, uint32_t n_blocks) {
    uint32_t v = i == (uint32_t)kCntBlocks ? n_blocks : 0u;
    if (i >= (uint32_t)kCntLvmm && i < (uint32_t)kCntLvmm + 3u) v = 0x7FFFFFFFu;        // INT32_MAX
    else if (i >= (uint32_t)kCntLvmm + 3u && i < (uint32_t)kCntLvmm + 6u) v = 0x80000000u;   // INT32_MIN
    else if (i >= (uint32_t)kCntLvHmm && i < (uint32_t)kCntLvHmm + 3u) v = 0x7FFFFFFFu;
    else if (i >= (uint32_t)kCntLvHmm + 3u && i < (uint32_t)kCntLvHmm + 6u) v = 0x80000000u;
    return v;
}
__global__ void dm_begin(uint32_t *counters, uint32_t n_blocks, uint32_t *mm, uint32_t *done) {
    const uint32_t i = threadIdx.x;
    if (i < (uint32_t)kCntWords) counters[i] = counter_begin_value(i, n_blocks);
    if (i < 3) mm[i] = mm[8 + i] = 0xFFFFFFFFu;   // mm[8..13]: the second box (kept hits) of the fused front end
    else if (i < 6) mm[i] = mm[8 + i] = 0u;
    if (i == 6) *done = 0u;
    for (uint32_t k = i; k < 1u + 64u; k += blockDim.x) mm[32u + 32u * k] = 0u;   // dm_commit_prune's arrival words (kArriveBase, kArriveStride)
}

// cell index of every point (kInvalidCell for non-finite points), value = cloud index
__global__ __launch_bounds__(256) void dm_grid_cells(const float *__restrict__ p, uint32_t n, float inv,
                                                    const GridParams *__restrict__ gp, uint32_t *keys, uint32_t *vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = p[3 * (size_t)i], y = p[3 * (size_t)i + 1], z = p[3 * (size_t)i + 2];
    uint32_t cell = kInvalidCell;
    if (finite3(x, y, z)) {
        const int c0 = (int)(floorf(x * inv) - (float)gp->lo[0]);
        const int c1 = (int)(floorf(y * inv) - (float)gp->lo[1]);
        const int c2 = (int)(floorf(z * inv) - (float)gp->lo[2]);
        cell = (uint32_t)(c0 + c1 * gp->m1 + c2 * gp->m2);
    }
    keys[i] = cell;
    vals[i] = i;
}

// dm_grid_cells as the key source of the sort's histogram launch (devmap_sort.h dm_radix_hist_src): one launch less per filter
struct GridCellsSrc {
    const float *p;
    float inv;
    const GridParams *gp;
    uint32_t *keys, *vals;
    __device__ __forceinline__ void begin(uint32_t, uint32_t) const {}
    template <class Add>
    __device__ __forceinline__ void operator()(uint32_t i, Add &add) const {
        const float x = p[3 * (size_t)i], y = p[3 * (size_t)i + 1], z = p[3 * (size_t)i + 2];
        uint32_t cell = kInvalidCell;
        if (finite3(x, y, z)) {
            const int c0 = (int)(floorf(x * inv) - (float)gp->lo[0]);
            const int c1 = (int)(floorf(y * inv) - (float)gp->lo[1]);
            const int c2 = (int)(floorf(z * inv) - (float)gp->lo[2]);
            cell = (uint32_t)(c0 + c1 * gp->m1 + c2 * gp->m2);
        }
        keys[i] = cell;
        vals[i] = i;
        add(cell);
    }
};

// Centroid of one voxel-grid cell: fp32 sums in cloud order (the sort is stable, values ascend inside a
// segment).  One thread per cell for the ordinary cells; cells with more than kBigCell points (the voxels
// next to the sensor collect one sample per beam — tens of thousands of points) go to dm_grid_centroids_big.
constexpr uint32_t kBigCell = 64;

__device__ __forceinline__ void grid_centroids_thread(const uint32_t seg, const float *__restrict__ p, const uint32_t *__restrict__ vals,
                                                      const uint32_t *__restrict__ seg_start, uint32_t *counters,
                                                      int seg_slot, int big_slot, uint4 *big, float *out, uint32_t big_cell) {
    if (seg >= counters[seg_slot]) return;
    const uint32_t s0 = seg_start[seg], s1 = seg_start[seg + 1];
    if (s1 - s0 > big_cell) {
        big[atomicAdd(&counters[big_slot], 1u)] = make_uint4(seg, s0, s1, 0u);   // (the bounds ride along: one dependent load less there)
        return;
    }
    // eight points per trip: the index loads, then the coordinate loads, are issued together (a cell is a chain of
    // dependent gathers otherwise); the sums still run in cloud order
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (uint32_t j = s0; j < s1; j += 8) {
        uint32_t v[8];
        float x[8], y[8], z[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = vals[min(j + u, s1 - 1u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = p[3 * (size_t)v[u]];
            y[u] = p[3 * (size_t)v[u] + 1];
            z[u] = p[3 * (size_t)v[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j + u < s1) {
                sx += x[u];
                sy += y[u];
                sz += z[u];
            }
    }
    const float cnt = (float)(s1 - s0);
    out[3 * (size_t)seg] = sx / cnt;
    out[3 * (size_t)seg + 1] = sy / cnt;
    out[3 * (size_t)seg + 2] = sz / cnt;
}

// s after m sequential fp32 additions of the same x:  for (k < m) s = s + x;  — bit-exact, in O(binades
// crossed) instead of O(m).  (Every beam contributes the sensor origin as its first free-space sample, so the
// voxel that holds the origin sums tens of thousands of identical values.)  While s stays inside one binade
// [2^e, 2^(e+1)) its ulp u is constant and round-to-nearest-even of s + x adds a constant number d of ulps:
// with x = (q + r) u, 0 <= r < 1:  d = q (r < 1/2), q + 1 (r > 1/2); a tie r = 1/2 resolves to the even
// neighbour, which after at most one step (s/u odd) makes every further step add q + (q & 1).  Steps that
// could leave the binade, and everything irregular (zero / opposite sign / |s| < |x| / non-normal values), are
// single real additions.
__host__ __device__ inline float add_repeat_f32(float s, float x, uint32_t m) {
    union FU { float f; uint32_t u; };
    FU fx; fx.f = x;
    const uint32_t xb = fx.u, xe = (xb >> 23) & 0xFFu;
    if (m == 0 || (xb << 1) == 0u) return s;  // +-0 never changes a sum that is not -0 (and s is never -0 here)
    const uint32_t mx = (xb & 0x7FFFFFu) | 0x800000u;
    while (m > 0) {
        FU fs; fs.f = s;
        const uint32_t sb = fs.u, se = (sb >> 23) & 0xFFu;
        const bool regular = se != 0u && se != 0xFFu && xe != 0u && xe != 0xFFu && ((sb ^ xb) >> 31) == 0u && se >= xe;
        if (regular) {
            const uint32_t shift = se - xe;
            if (shift > 24u) return s;  // x < u / 2: s + x rounds back to s for good
            const uint32_t S = (sb & 0x7FFFFFu) | 0x800000u;
            const uint32_t q = mx >> shift, r = mx & ((1u << shift) - 1u), half = shift ? (1u << (shift - 1u)) : 0u;
            uint32_t d;
            bool jump = true;
            if (shift == 0u || r < half) d = q;
            else if (r > half) d = q + 1u;
            else {  // tie
                d = q + (q & 1u);
                jump = (S & 1u) == 0u;
            }
            if (jump) {
                if (d == 0u) return s;
                uint32_t k = (0xFFFFFFu - S) / d;  // steps that provably stay inside the binade
                if (k > m) k = m;
                if (k > 0u) {
                    fs.u = (sb & 0xFF800000u) | ((S + k * d) & 0x7FFFFFu);
                    s = fs.f;
                    m -= k;
                    if (m == 0) break;
                }
            }
        }
        s = s + x;  // one real addition (binade crossing or irregular operands)
        --m;
    }
    return s;
}

// Chunk descriptors for the large cells: chunk q = sorted positions [512 q, 512 q + 512).  If the whole chunk lies in
// one cell, desc[q] = {bit c set when all 512 values of coordinate c are identical, x bits, y bits, z bits}; else
// {0, ...}.  Fully parallel (one wave per chunk) — it lets the serial kernel below skip whole chunks of the run of
// identical samples (the sensor origin, once per beam) without touching their data.
constexpr uint32_t kChunk = 512;

__device__ __forceinline__ void big_chunks_wave(const uint32_t q, const int lane, const float *__restrict__ p, const uint32_t *__restrict__ vals,
                                                const uint32_t *__restrict__ flag, const uint32_t *__restrict__ scan,
                                                const uint32_t *__restrict__ counters, int valid_slot, uint4 *desc) {
    const uint32_t i0 = q * kChunk;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (i0 + kChunk <= counters[valid_slot]) {
        const uint32_t sa = scan[i0] + flag[i0], sb = scan[i0 + kChunk - 1] + flag[i0 + kChunk - 1];
        if (sa == sb) {
            uint32_t first[3];
            bool same[3] = {true, true, true};
#pragma unroll
            for (int u = 0; u < (int)(kChunk / 64); ++u) {
                const uint32_t v = vals[i0 + 64u * u + lane];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t xb = __float_as_uint(p[3 * (size_t)v + c]);
                    if (u == 0) first[c] = __builtin_amdgcn_readfirstlane(xb);
                    same[c] = same[c] && xb == first[c];
                }
            }
            uint32_t bits = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) bits |= (__ballot(same[c]) == ~0ull ? 1u : 0u) << c;
            d = make_uint4(bits, first[0], first[1], first[2]);
        }
    }
    if (lane == 0) desc[q] = d;
}

// The ordinary cells' centroids and the chunk descriptors in ONE launch (round 5: both only need the segment scan; two launches
// before): the last main_wgs workgroups take one cell per thread, the workgroups before them one chunk per wave.
__global__ __launch_bounds__(256) void dm_grid_centroids(const float *__restrict__ p, const uint32_t *__restrict__ vals,
                                                        const uint32_t *__restrict__ seg_start, uint32_t *counters,
                                                        int seg_slot, int big_slot, uint4 *big, float *out, uint32_t main_wgs,
                                                        const uint32_t *__restrict__ flag, const uint32_t *__restrict__ scan,
                                                        int valid_slot, uint32_t nchunk, uint4 *desc, uint32_t big_cell) {
    // (the chunk waves first: their short gather chains then run beside the cell threads instead of forming the launch's tail)
    const uint32_t chunk_wgs = gridDim.x - main_wgs;
    if (blockIdx.x >= chunk_wgs) {
        grid_centroids_thread((blockIdx.x - chunk_wgs) * blockDim.x + threadIdx.x, p, vals, seg_start, counters, seg_slot, big_slot, big, out, big_cell);
        return;
    }
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q < nchunk) big_chunks_wave(q, (int)(threadIdx.x & 63u), p, vals, flag, scan, counters, valid_slot, desc);
}

// Large cells: one wave per cell.  The sums must stay serial fp32 chains in cloud order, so the wave loads 512 points at a time
// (coalesced index loads one trip ahead, then 24 gathers), parks their coordinates in LDS, and lanes 0..2 each walk one
// coordinate's values, 64 at a time, with dependent adds — sum_k = sum_{k-1} + x_k exactly as the sequential loop, three chains
// in the same instructions.  Runs of identical values are summed in closed form (add_repeat_f32, per coordinate); chunks that
// dm_big_chunks found uniform in all three coordinates extend the pending runs without loading anything.
// Round 5, what the earlier form (rounds 2-4) taught: it ran the chain across the lanes with a full-wave DPP shift, one wave per
// (cell, coordinate), every 64-step chain and all 8 sub-batches of a trip unrolled — 3 000 VALU lines of straight-line code that a wave
// executes once or twice.  The launch took 41 us at 1 300 large cells whatever the chain or the loads cost (LDS chain: 54 us;
// unpredicated loads: the same): a wave that runs cold straight-line code is bound by instruction fetch, not by the instructions.
// This form keeps the sub-batch and chain loops rolled (the whole kernel is a few KB).
constexpr uint32_t kBigTrip = 512;
__global__ __launch_bounds__(64) void dm_grid_centroids_big(const float *__restrict__ p, const uint32_t *__restrict__ vals,
                                                           const uint32_t *__restrict__ seg_start,
                                                           const uint32_t *__restrict__ counters, int big_slot,
                                                           const uint4 *__restrict__ big, const uint4 *__restrict__ desc,
                                                           float *out) {
    __shared__ __attribute__((aligned(16))) float buf[3][kBigTrip];
    const int lane = threadIdx.x;
    const int c = lane < 3 ? lane : 0;   // the coordinate this lane sums (lanes >= 3 shadow lane 0, their results are dropped)
    const uint32_t nbig = counters[big_slot];
    for (uint32_t i = blockIdx.x; i < nbig; i += gridDim.x) {
        const uint4 item = big[i];
        const uint32_t seg = item.x, s0 = item.y, s1 = item.z;
        float s = 0.f;
        uint32_t run_x = 0u, run_len = 0u;  // pending run of identical values of coordinate c (bits, count)

        // generic path over sorted positions [lo, hi), 512 per trip.  Positions behind the stretch are clamped to its last one
        // instead of predicated (a predicated load is a branch, and branches between the loads serialise their round trips).
        auto process = [&](const uint32_t lo, const uint32_t hi) {
            if (lo >= hi) return;
            const uint32_t last = hi - 1u;
            uint32_t vi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vi[u] = vals[min(lo + 64u * u + lane, last)];
            for (uint32_t b = lo; b < hi; b += kBigTrip) {
                float xc[8], yc[8], zc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    xc[u] = p[3 * (size_t)vi[u]];
                    yc[u] = p[3 * (size_t)vi[u] + 1];
                    zc[u] = p[3 * (size_t)vi[u] + 2];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) vi[u] = vals[min(b + kBigTrip + 64u * u + lane, last)];   // the next trip's indices
                __syncthreads();   // (one wave per workgroup: the buffer's previous readers are done)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    buf[0][64 * u + lane] = xc[u];
                    buf[1][64 * u + lane] = yc[u];
                    buf[2][64 * u + lane] = zc[u];
                }
                __syncthreads();
                const uint32_t nsub = (min(kBigTrip, hi - b) + 63u) / 64u;
#pragma unroll 1
                for (uint32_t u = 0; u < nsub; ++u) {
                    const uint32_t nb = min(64u, hi - (b + 64u * u));
                    const bool in = (uint32_t)lane < nb;
                    const uint32_t xb = __float_as_uint(buf[0][64u * u + lane]), yb = __float_as_uint(buf[1][64u * u + lane]),
                                   zb = __float_as_uint(buf[2][64u * u + lane]);
                    // per coordinate: a sub-batch of identical values extends the pending run (summed in closed form when it ends)
                    const uint32_t fx = __builtin_amdgcn_readfirstlane(xb), fy = __builtin_amdgcn_readfirstlane(yb),
                                   fz = __builtin_amdgcn_readfirstlane(zb);
                    const bool ux = __ballot(!in || xb == fx) == ~0ull, uy = __ballot(!in || yb == fy) == ~0ull,
                               uz = __ballot(!in || zb == fz) == ~0ull;
                    const bool uniform = c == 0 ? ux : (c == 1 ? uy : uz);
                    const uint32_t first = c == 0 ? fx : (c == 1 ? fy : fz);
                    bool chain = false;
                    if (uniform && (run_len == 0u || first == run_x)) {
                        run_x = first;
                        run_len += nb;
                    } else {
                        if (run_len) {
                            s = add_repeat_f32(s, __uint_as_float(run_x), run_len);
                            run_len = 0;
                        }
                        if (uniform) {
                            run_x = first;
                            run_len = nb;
                        } else {
                            chain = true;
                        }
                    }
                    if (chain) {   // (the values behind the stretch's end are the clamped last point's: only nb of the 64 are added)
                        const float *row = &buf[c][64u * u];
                        uint32_t j = 0;
#pragma unroll 2
                        for (; j + 4u <= nb; j += 4u) {
                            const float4 q = *reinterpret_cast<const float4 *>(row + j);
                            s = s + q.x;
                            s = s + q.y;
                            s = s + q.z;
                            s = s + q.w;
                        }
#pragma unroll 1
                        for (; j < nb; ++j) s = s + row[j];
                    }
                }
            }
        };

        // Head up to the first chunk boundary, whole chunks by descriptor, tail — but a stretch is only handed to process()
        // when a uniform chunk interrupts it (every process() call starts with two dependent memory round trips — indices,
        // then coordinates).  Descriptors: 64 per load, the next load in flight.
        const uint32_t a = min(s1, (s0 + kChunk - 1u) & ~(kChunk - 1u));
        const uint32_t z = max(a, s1 & ~(kChunk - 1u));
        uint32_t cur_lo = s0, cur_hi = a;   // pending stretch (sorted positions)
        const uint32_t qa = a / kChunk, qz = z / kChunk;
        uint4 dl_next = make_uint4(0u, 0u, 0u, 0u);
        if (qa + (uint32_t)lane < qz) dl_next = desc[qa + lane];
        for (uint32_t q0 = qa; q0 < qz; q0 += 64u) {
            const uint32_t nq = min(64u, qz - q0);
            const uint4 dl = dl_next;
            dl_next = make_uint4(0u, 0u, 0u, 0u);
            if (q0 + 64u + (uint32_t)lane < qz) dl_next = desc[q0 + 64u + lane];
            // (scalar view of the 64 flags: a chunk counts when all three coordinates are uniform in it)
            const unsigned long long fmask = __ballot((dl.x & 7u) == 7u);
            // The sensor's own voxel: every chunk of the round uniform, and with the same values — the round extends the runs in
            // one step.  (Chunk by chunk, the loop below is ~60 instructions of ONE wave per chunk: 0.25 us each, 46 us for the
            // 185 chunks of a 200 000-ray scan — measured in round 5; it was the whole launch.)
            {
                const bool inq = (uint32_t)lane < nq;
                const uint32_t ax = __builtin_amdgcn_readfirstlane(dl.y), ay = __builtin_amdgcn_readfirstlane(dl.z),
                               az = __builtin_amdgcn_readfirstlane(dl.w);
                const bool all_same = __ballot(!inq || ((dl.x & 7u) == 7u && dl.y == ax && dl.z == ay && dl.w == az)) == ~0ull;
                if (all_same) {   // (uniform over the wave)
                    process(cur_lo, cur_hi);   // (no-op when empty)
                    const uint32_t xb = c == 0 ? ax : (c == 1 ? ay : az);
                    if (run_len && xb != run_x) {
                        s = add_repeat_f32(s, __uint_as_float(run_x), run_len);
                        run_len = 0;
                    }
                    run_x = xb;
                    run_len += nq * kChunk;
                    cur_lo = cur_hi = (q0 + nq) * kChunk;
                    continue;
                }
            }
#pragma unroll 1
            for (uint32_t k = 0; k < nq; ++k) {
                if ((fmask >> k) & 1ull) {
                    process(cur_lo, cur_hi);   // (no-op when empty)
                    const uint32_t xb = c == 0 ? __builtin_amdgcn_readlane(dl.y, (int)k)
                                               : (c == 1 ? __builtin_amdgcn_readlane(dl.z, (int)k) : __builtin_amdgcn_readlane(dl.w, (int)k));
                    if (run_len && xb != run_x) {
                        s = add_repeat_f32(s, __uint_as_float(run_x), run_len);
                        run_len = 0;
                    }
                    run_x = xb;
                    run_len += kChunk;
                    cur_lo = cur_hi = (q0 + k + 1u) * kChunk;
                } else {
                    cur_hi = (q0 + k + 1u) * kChunk;
                }
            }
        }
        process(cur_lo, s1);
        if (run_len) s = add_repeat_f32(s, __uint_as_float(run_x), run_len);
        if (lane < 3) out[3 * (size_t)seg + lane] = s / (float)(s1 - s0);
    }
}

// test hook: out_fast[i] = add_repeat_f32(s[i], x[i], m[i]), out_loop[i] = the plain loop
__global__ void dm_diag_add_repeat(const float *s, const float *x, const uint32_t *m, uint32_t n, float *out_fast, float *out_loop) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_fast[i] = add_repeat_f32(s[i], x[i], m[i]);
    float acc = s[i];
    const float xi = x[i];
    for (uint32_t k = 0; k < m[i]; ++k) acc = acc + xi;
    out_loop[i] = acc;
}

// ---- beam sampling (bgkoctomap.cpp:383-417, 433-458) ------------------------------------------------
struct BeamArgs {
    float ox, oy, oz;
    float free_res;
    float max_range;
};

__device__ __forceinline__ float f32_sqrt_cr(float x) {  // correctly rounded (== (float)sqrt((double)x)):
    return sqrtf(x);                                      // hipcc's default f32 sqrt is IEEE
}

// number of free-space samples of one beam: the origin, every free_res, one sample free_res short of the hit
// A beam whose walk would not end is cut at kBeamCap samples and flagged (error bit 2): an infinite range, or a
// free_resolution so small against the range that d += fr stops advancing in fp32 — the reference's host loop
// (bgkoctomap.cpp:445-457) would spin or exhaust memory there; on the GPU it would hang the device.
constexpr uint32_t kBeamCap = 1u << 22;
constexpr uint32_t kErrBeam = 2u;
__device__ __forceinline__ uint32_t beam_count(float l, float fr, uint32_t *counters) {
    uint32_t c = 1;
    for (float d = fr; d < l && c < kBeamCap; d += fr) ++c;
    if (c >= kBeamCap) {
        atomicOr(&counters[kCntError], kErrBeam);
        return 1;
    }
    if (l > fr) ++c;
    return c;
}
// wave-reduced 64-bit total of the per-beam sample counts (the 32-bit offsets wrap silently above 2^32 samples)
// (one atomic per workgroup of up to 256 threads, all of which call this: every atomic on this one address costs ~25 ns,
// serialised)
__device__ __forceinline__ void beam_total_add(unsigned long long c, uint32_t *counters) {
    __shared__ unsigned long long s_t[4];
    unsigned long long t = c;
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63) == 0) s_t[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long all = 0;
        for (uint32_t w = 0; w < (blockDim.x + 63u) >> 6 && w < 4u; ++w) all += s_t[w];
        if (all) atomicAdd(reinterpret_cast<unsigned long long *>(counters + kCntBeamTotal), all);
    }
}

constexpr uint32_t kBeamCountWgs = 256;
__global__ __launch_bounds__(256) void dm_beam_count(const float *__restrict__ hits, uint32_t n, BeamArgs a, uint32_t *keep,
                                                    uint32_t *nfree, uint32_t *counters) {
    // grid-stride over at most kBeamCountWgs workgroups: the 64-bit total is ONE atomic per workgroup on one address (~25 ns per
    // caller, serialised) — with a workgroup per 256 beams that chain WAS this kernel at configs[4]'s size (1 630 atomics, 24 us)
    unsigned long long total = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = hits[3 * (size_t)i], y = hits[3 * (size_t)i + 1], z = hits[3 * (size_t)i + 2];
        const float dx = x - a.ox, dy = y - a.oy, dz = z - a.oz;
        const float s = dx * dx + dy * dy + dz * dz;
        bool k = true;
        // point3f::norm() > max_range in f64: sqrt((double)s) > R  <=>  s > R*R (R*R is exact in f64 and the
        // gap between a float s and R*R is far above half an ulp of the f64 root)
        if (a.max_range > 0.0f) k = !((double)s > (double)a.max_range * (double)a.max_range);
        keep[i] = k ? 1u : 0u;
        const uint32_t c = k ? beam_count(f32_sqrt_cr(s), a.free_res, counters) : 0u;
        nfree[i] = c;
        total += c;
    }
    beam_total_add(total, counters);
}

// hits that pass the gate -> xy (label 1) in order; their beam samples -> frees (xyz) in order
// Also reduces, on the way, the box of the free samples it writes (the second voxel filter's grid: its min/max launch
// and one-thread parameter launch are gone) and the box of the kept hits (half of the training set's box).
// kOwn (sharded sample filter, below): free_off = offsets of the OWN samples, only samples whose filter layer
// (int)floorf(z * inv) lies in [lo, hi) are written; the box is still the box of all samples.
template <bool kOwn>
__global__ __launch_bounds__(256) void dm_beam_write(const float *__restrict__ hits, uint32_t n, BeamArgs a,
                                                    const uint32_t *__restrict__ keep,
                                                    const uint32_t *__restrict__ keep_off,
                                                    const uint32_t *__restrict__ free_off, float4 *xy, float *frees,
                                                    uint32_t *mm_frees, MinmaxFin fin_frees, uint32_t *mm_hits, float inv, int lo,
                                                    int hi) {
    auto own = [&](float sz) {
        if (!kOwn) return true;
        const int L = (int)floorf(sz * inv);
        return L >= lo && L < hi;
    };
    __shared__ float red[4][6];
    __shared__ uint32_t s_last;
    float fmn[3] = {INFINITY, INFINITY, INFINITY}, fmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float hmn[3] = {INFINITY, INFINITY, INFINITY}, hmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!keep[i]) continue;
        const float x = hits[3 * (size_t)i], y = hits[3 * (size_t)i + 1], z = hits[3 * (size_t)i + 2];
        xy[keep_off[i]] = make_float4(x, y, z, 1.0f);
        box_add(hmn, hmx, x, y, z);
        const float dx = x - a.ox, dy = y - a.oy, dz = z - a.oz;
        const float l = f32_sqrt_cr(dx * dx + dy * dy + dz * dz);
        const float nx = dx / l, ny = dy / l, nz = dz / l;
        float *f = frees + 3 * (size_t)free_off[i];
        if (own(a.oz)) {
            f[0] = a.ox; f[1] = a.oy; f[2] = a.oz;
            f += 3;
        }
        box_add(fmn, fmx, a.ox, a.oy, a.oz);
        for (float d = a.free_res; d < l; d += a.free_res) {
            const float sx = a.ox + nx * d, sy = a.oy + ny * d, sz = a.oz + nz * d;
            if (own(sz)) {
                f[0] = sx; f[1] = sy; f[2] = sz;
                f += 3;
            }
            box_add(fmn, fmx, sx, sy, sz);
        }
        if (l > a.free_res) {
            const float d = l - a.free_res;
            const float sx = a.ox + nx * d, sy = a.oy + ny * d, sz = a.oz + nz * d;
            if (own(sz)) {
                f[0] = sx; f[1] = sy; f[2] = sz;
            }
            box_add(fmn, fmx, sx, sy, sz);
        }
    }
    if (!mm_frees) return;   // (uniform) the unfused form
    MinmaxFin none = fin_frees;
    none.mode = 0;
    minmax_wg(hmn, hmx, mm_hits, none, red, &s_last);
    minmax_wg(fmn, fmx, mm_frees, fin_frees, red, &s_last);
}

// ---- sharded sample filter (block-sharded insert, la3dm_devmap_set_shard): the second voxel filter — the largest stage of
// the front end — is divided over the ranks by ABSOLUTE z-LAYER of its grid: layer(s) = (int)floorf(s.z * inv), the same
// expression dm_grid_cells uses, so a cell's samples all carry one layer.  The filter's linear cell index has z slowest
// (bgkoctomap.cpp:419-431 -> pcl::VoxelGrid: idx = i0 + i1 d0 + i2 d0 d1), so a contiguous range of layers is a
// contiguous range of cell indices: every cell lives on exactly one rank with ALL its samples in cloud order (the fp32
// centroid chain is unchanged), and the ranks' filtered outputs, concatenated in rank order, ARE the single-GPU output.
// Three walks over the beams (each is the float-stepped loop of beam_sample, bgkoctomap.cpp:433-458, positions included):
//   dm_beam_hist        samples per layer (<= kShardLayers bins, LDS-private per workgroup) -> the host cuts the layers
//   dm_beam_count_own   samples of beam i inside this rank's layer range
//   dm_beam_write<true> kept hits -> xy as usual; only the own samples are written; the box of ALL samples is reduced
//                       (the filter grid's parameters must be the global ones)
constexpr uint32_t kShardLayers = 1024;
template <class F>
__device__ __forceinline__ void beam_walk_z(float x, float y, float z, const BeamArgs &a, F &&f) {   // f(sz) per sample, in order
    const float dx = x - a.ox, dy = y - a.oy, dz = z - a.oz;
    const float l = f32_sqrt_cr(dx * dx + dy * dy + dz * dz);
    const float nz = dz / l;
    f(a.oz);
    uint32_t c = 1;
    for (float d = a.free_res; d < l && c < kBeamCap; d += a.free_res, ++c) f(a.oz + nz * d);
    if (l > a.free_res) f(a.oz + nz * (l - a.free_res));
}
__global__ __launch_bounds__(256) void dm_beam_hist(const float *__restrict__ hits, uint32_t n, BeamArgs a,
                                                   const uint32_t *__restrict__ keep, float inv, int zbase, uint32_t nlayer,
                                                   uint32_t *__restrict__ hist, uint32_t *counters) {
    __shared__ uint32_t s_h[kShardLayers];
    for (uint32_t j = threadIdx.x; j < nlayer; j += blockDim.x) s_h[j] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!keep[i]) continue;
        beam_walk_z(hits[3 * (size_t)i], hits[3 * (size_t)i + 1], hits[3 * (size_t)i + 2], a, [&](float sz) {
            const int L = (int)floorf(sz * inv) - zbase;
            if ((unsigned)L < nlayer) atomicAdd(&s_h[L], 1u);
            else atomicOr(&counters[kCntError], 1u);   // outside the bound the host derived from the cloud's box: cannot happen
        });
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nlayer; j += blockDim.x)
        if (s_h[j]) atomicAdd(&hist[j], s_h[j]);
}
__global__ __launch_bounds__(256) void dm_beam_count_own(const float *__restrict__ hits, uint32_t n, BeamArgs a,
                                                        const uint32_t *__restrict__ keep, float inv, int lo, int hi,
                                                        uint32_t *__restrict__ nown) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = 0;
    if (keep[i])
        beam_walk_z(hits[3 * (size_t)i], hits[3 * (size_t)i + 1], hits[3 * (size_t)i + 2], a, [&](float sz) {
            const int L = (int)floorf(sz * inv);
            c += (L >= lo && L < hi) ? 1u : 0u;
        });
    nown[i] = c;
}
// (rank q's filtered points into their place of the common list: a plain copy, after the count exchange)
__global__ void dm_copy_f3(const float *__restrict__ src, uint32_t n3, float *__restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) dst[i] = src[i];
}

// ---- BGKLOctoMap front end (src/bgkloctomap/bgkloctomap.cpp:300-343, beam_sample :359-381): a hit is re-projected
// as origin + n * l, its free samples step DOWN from l - free_resolution while d > 0 (no second voxel filter), every
// sample of a beam — the origin sample first — remembers its beam, and the beam itself is the segment
// origin -> origin + n * (l - free_resolution).  Per kept hit, in order: {hit, origin sample, free samples}.
struct LBeam {
    float ex, ey, ez;  // re-projected end point
    float nx, ny, nz;  // direction
    float l;           // range
    float mx, my, mz;  // direction of the re-projected end point
    float l2;          // its range
};
__device__ __forceinline__ LBeam l_beam(float x, float y, float z, const BeamArgs &a) {
    LBeam b;
    const float dx = x - a.ox, dy = y - a.oy, dz = z - a.oz;
    b.l = f32_sqrt_cr(dx * dx + dy * dy + dz * dz);
    b.nx = dx / b.l;
    b.ny = dy / b.l;
    b.nz = dz / b.l;
    b.ex = a.ox + b.nx * b.l;
    b.ey = a.oy + b.ny * b.l;
    b.ez = a.oz + b.nz * b.l;
    const float fx = b.ex - a.ox, fy = b.ey - a.oy, fz = b.ez - a.oz;
    b.l2 = f32_sqrt_cr(fx * fx + fy * fy + fz * fz);
    b.mx = fx / b.l2;
    b.my = fy / b.l2;
    b.mz = fz / b.l2;
    return b;
}

__global__ __launch_bounds__(256) void dm_l_beam_count(const float *__restrict__ hits, uint32_t n, BeamArgs a, uint32_t *keep,
                                                      uint32_t *nsamp, uint32_t *counters) {
    unsigned long long total = 0;   // (grid-stride, at most kBeamCountWgs workgroups: see dm_beam_count)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = hits[3 * (size_t)i], y = hits[3 * (size_t)i + 1], z = hits[3 * (size_t)i + 2];
        const float dx = x - a.ox, dy = y - a.oy, dz = z - a.oz;
        const float s = dx * dx + dy * dy + dz * dz;
        bool k = true;
        if (a.max_range > 0.0f) k = !((double)s > (double)a.max_range * (double)a.max_range);  // see dm_beam_count
        uint32_t c = 0;
        if (k) {
            const LBeam b = l_beam(x, y, z, a);
            c = 2;  // the re-projected hit and the origin sample
            for (float d = b.l2 - a.free_res; d > 0.0f && c < kBeamCap; d -= a.free_res) ++c;
            if (c >= kBeamCap) {  // see beam_count
                atomicOr(&counters[kCntError], kErrBeam);
                c = 2;
            }
        }
        keep[i] = k ? 1u : 0u;
        nsamp[i] = c;
        total += c;
    }
    beam_total_add(total, counters);
}

__global__ __launch_bounds__(256) void dm_l_beam_write(const float *__restrict__ hits, uint32_t n, BeamArgs a,
                                                      const uint32_t *__restrict__ keep, const uint32_t *__restrict__ beam_of,
                                                      const uint32_t *__restrict__ samp_off, float4 *xy, int32_t *ray_idx,
                                                      float *rays) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const LBeam b = l_beam(hits[3 * (size_t)i], hits[3 * (size_t)i + 1], hits[3 * (size_t)i + 2], a);
    const int32_t id = (int32_t)beam_of[i];
    size_t w = samp_off[i];
    xy[w] = make_float4(b.ex, b.ey, b.ez, 1.0f);
    ray_idx[w++] = -1;
    xy[w] = make_float4(a.ox, a.oy, a.oz, 0.0f);
    ray_idx[w++] = id;
    for (float d = b.l2 - a.free_res; d > 0.0f; d -= a.free_res) {
        xy[w] = make_float4(a.ox + b.mx * d, a.oy + b.my * d, a.oz + b.mz * d, 0.0f);
        ray_idx[w++] = id;
    }
    const float l = b.l - a.free_res;
    float *r = rays + 6 * (size_t)id;
    r[0] = a.ox; r[1] = a.oy; r[2] = a.oz;
    r[3] = a.ox + b.nx * l; r[4] = a.oy + b.ny * l; r[5] = a.oz + b.nz * l;
}

// Training rows of every block (bgkloctomap.cpp:141-170): a hit becomes a degenerate segment with label 1, a beam
// contributes its segment once per block.  The members of a block are in ascending sample order and the samples of
// one beam are consecutive, so "once per block" = "differs from the previous member's beam".
__global__ __launch_bounds__(256) void dm_l_row_flags(const uint32_t *__restrict__ member_pt, const uint32_t *__restrict__ head,
                                                     uint32_t n_mem, const int32_t *__restrict__ ray_idx, uint32_t *rflag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_mem) return;
    const int32_t r = ray_idx[member_pt[k]];
    rflag[k] = (r < 0 || head[k] || ray_idx[member_pt[k - 1]] != r) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void dm_l_rows_write(const uint32_t *__restrict__ member_pt, uint32_t n_mem,
                                                      const uint32_t *__restrict__ rflag, const uint32_t *__restrict__ rscan,
                                                      const float4 *__restrict__ xy, const int32_t *__restrict__ ray_idx,
                                                      const float *__restrict__ rays, float4 *rows) {
    // The rows go out in the inference kernels' own 12-float form (bgkl_kernels.h bgkl_rows_prepare: the segment's direction,
    // squared length and "shorter than 0.1 mm" decision beside the end points — the same fp32 expressions, so the same bits;
    // LA3DM_SCAN_ROWS_PREPARED): the 8-float rows and the launch that widened them were 57 us and 0.17 GB of a 200 k-ray insert.
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_mem || !rflag[k]) return;
    const uint32_t src = member_pt[k];
    const int32_t r = ray_idx[src];
    float4 *o = rows + 3 * (size_t)rscan[k];
    float4 p0, p1;
    if (r < 0) {
        const float4 p = xy[src];
        p0 = make_float4(p.x, p.y, p.z, p.x);
        p1 = make_float4(p.y, p.z, 1.0f, 0.0f);
    } else {
        const float *q = rays + 6 * (size_t)r;
        p0 = make_float4(q[0], q[1], q[2], q[3]);
        p1 = make_float4(q[4], q[5], 0.0f, 0.0f);
    }
    const float lx = p0.w - p0.x, ly = p1.x - p0.y, lz = p1.y - p0.z;
    const float c2 = lx * lx + ly * ly + lz * lz;
    const bool degenerate = sqrtf(c2) < 0.0001f;
    o[0] = p0;
    o[1] = make_float4(p1.x, p1.y, p1.z, degenerate ? 1.0f : 0.0f);
    o[2] = make_float4(lx, ly, lz, c2);
}

// CSR of the rows over the training blocks: rows_off[b] = rows before the block's first member
__global__ __launch_bounds__(256) void dm_l_rows_off(const uint32_t *__restrict__ train_off, const uint32_t *__restrict__ counters,
                                                    const uint32_t *__restrict__ rflag, const uint32_t *__restrict__ rscan,
                                                    uint32_t n_mem, uint32_t *rows_off) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_geo = counters[kCntGeo];
    if (b > n_geo) return;
    rows_off[b] = b < n_geo ? rscan[train_off[b]] : rscan[n_mem - 1] + rflag[n_mem - 1];
}

// work counters of a pass in rows (what the oracle counts for this variant)
__global__ __launch_bounds__(256) void dm_l_test_stats(const int32_t *__restrict__ nbr, const uint32_t *__restrict__ rows_off,
                                                      const uint32_t *__restrict__ nleaf, uint32_t n_test, uint32_t *counters) {
    __shared__ unsigned long long s_w[4], s_pw[4];
    unsigned long long *acc_reads = reinterpret_cast<unsigned long long *>(counters + kCntTrainReads);
    unsigned long long *acc_pairs = reinterpret_cast<unsigned long long *>(counters + kCntPairEvals);
    unsigned long long w = 0, pw = 0;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_test; t += gridDim.x * blockDim.x) {
        unsigned long long wt = 0;
        for (int q = 0; q < 7; ++q) {
            const int32_t tb = nbr[7 * (size_t)t + q];
            if (tb >= 0) wt += rows_off[tb + 1] - rows_off[tb];
        }
        w += wt;
        pw += wt * nleaf[t];
    }
    for (int o = 32; o >= 1; o >>= 1) {
        w += __shfl_xor(w, o);
        pw += __shfl_xor(pw, o);
    }
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = w;
        s_pw[threadIdx.x >> 6] = pw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        w = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        pw = s_pw[0] + s_pw[1] + s_pw[2] + s_pw[3];
        if (w | pw) {
            atomicAdd(acc_reads, w);
            atomicAdd(acc_pairs, pw);
        }
    }
}

// read_counters without a stream synchronisation: the counter block goes straight to pinned host memory, followed by a
// sequence number the host spins on (a D2H copy + hipStreamSynchronize costs ~20 us of driver latency per read-back,
// seven times per scan; this is a few us).  The kernel runs after everything queued before it on the stream, so the host
// seeing the sequence number is as good as a synchronisation for that work.
__global__ void dm_publish_counters(const uint32_t *__restrict__ counters, volatile uint32_t *mailbox, uint32_t seq) {
    const uint32_t i = threadIdx.x;
    if (i < (uint32_t)kCntWords) mailbox[i] = counters[i];
    __threadfence_system();
    __syncthreads();
    if (i == 0) {
        mailbox[kCntWords] = seq;
        __threadfence_system();
    }
}

__global__ __launch_bounds__(256) void dm_append_frees(const float *__restrict__ pts, uint32_t n, uint32_t base, float label,
                                                      float4 *xy, uint32_t *mm, MinmaxFin fin) {
    __shared__ float red[4][6];
    __shared__ uint32_t s_last;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 4u * stride) {   // (four points per trip, loads in flight together: see dm_minmax)
        float q[4][3];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t ij = i + j * stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) q[j][c] = ij < n ? pts[3 * (size_t)ij + c] : NAN;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t ij = i + j * stride;
            if (ij < n) xy[base + ij] = make_float4(q[j][0], q[j][1], q[j][2], label);
            box_add(mn, mx, q[j][0], q[j][1], q[j][2]);
        }
    }
    if (!mm) return;   // (uniform) the unfused form
    minmax_wg(mn, mx, mm, fin, red, &s_last);   // mode 2 with mm_extra = the kept hits' box: the training set's box
}

// ---- partition (stage B..D) -----------------------------------------------------------------------------
struct PartArgs {
    float bs, half;        // block size, size / 2
    int g0[3];             // biased block index of grid cell 0 on each axis
    int gn[3];             // grid extent
    const uint8_t *mult[3];  // multiplicity of each biased index in the float-stepped candidate sequence,
                             // indexed by (idx - g0[a]); 0 = not a candidate index
};

__device__ __forceinline__ long long axis_index(float v, float bs) {  // bgkblock.cpp:73-77, one axis
    return (long long)((double)v / (double)bs + 524288.5);
}
__device__ __forceinline__ float axis_center(long long i, float bs) {  // bgkblock.cpp:79-83
    return (float)(i - 524288) * bs;
}

struct AxisCandD {
    int first;   // the first of the n consecutive biased indices (a closed box can only be shared with an adjacent block)
    int n;
};
// biased block indices whose CLOSED fp32 box [c - h, c + h] holds v (what an R-tree box query sees).  No index array: filled
// at a run-time position it lived in scratch memory (every store and load a memory round trip: dm_members_count took 32 us)
__device__ __forceinline__ AxisCandD axis_candidates_dev(float v, float bs, float h) {
    AxisCandD r;
    r.first = 0;
    r.n = 0;
    const long long i0 = axis_index(v, bs);
#pragma unroll
    for (int o = -1; o <= 1; ++o) {
        const float c = axis_center(i0 + o, bs);
        if (c - h <= v && v <= c + h) {
            if (r.n == 0) r.first = (int)(i0 + o);
            ++r.n;
        }
    }
    return r;
}

__device__ __forceinline__ bool grid_cid(const PartArgs &a, int ix, int iy, int iz, uint32_t &cid) {
    const int x = ix - a.g0[0], y = iy - a.g0[1], z = iz - a.g0[2];
    if ((unsigned)x >= (unsigned)a.gn[0] || (unsigned)y >= (unsigned)a.gn[1] || (unsigned)z >= (unsigned)a.gn[2]) return false;
    cid = ((uint32_t)x * (uint32_t)a.gn[1] + (uint32_t)y) * (uint32_t)a.gn[2] + (uint32_t)z;
    return true;
}

// the same cell in the COUNT array of the x-slab partition: x fastest.  The training set is sorted by voxel-grid cell, x fastest, so
// the lanes of a wave hold consecutive x blocks of one (y, z): in the x-major cell index those are ~12 KB apart (one L2 line per
// lane and atomic), x fastest they share a line
__device__ __forceinline__ bool grid_cnt_idx(const PartArgs &a, int ix, int iy, int iz, uint32_t &idx) {
    const int x = ix - a.g0[0], y = iy - a.g0[1], z = iz - a.g0[2];
    if ((unsigned)x >= (unsigned)a.gn[0] || (unsigned)y >= (unsigned)a.gn[1] || (unsigned)z >= (unsigned)a.gn[2]) return false;
    idx = ((uint32_t)z * (uint32_t)a.gn[1] + (uint32_t)y) * (uint32_t)a.gn[0] + (uint32_t)x;
    return true;
}

// per point: its closed-box candidates per axis, packed {first index x, y, z, nx | ny << 2 | nz << 4} (the f64 index
// arithmetic runs once; the write pass replays the code), and the number of (block, point) pairs
__global__ __launch_bounds__(256) void dm_members_count(const float4 *__restrict__ xy, uint32_t n, PartArgs a, uint32_t *cnt,
                                                       int4 *code) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = xy[i];
    const AxisCandD ax = axis_candidates_dev(p.x, a.bs, a.half), ay = axis_candidates_dev(p.y, a.bs, a.half),
                    az = axis_candidates_dev(p.z, a.bs, a.half);
    cnt[i] = (uint32_t)(ax.n * ay.n * az.n);
    // the candidates of one axis are consecutive indices (a closed box can only be shared with an adjacent block)
    code[i] = make_int4(ax.first, ay.first, az.first, ax.n | (ay.n << 2) | (az.n << 4));
}

__global__ __launch_bounds__(256) void dm_members_write(const int4 *__restrict__ code, uint32_t n, PartArgs a,
                                                       const uint32_t *__restrict__ off, uint32_t *keys, uint32_t *vals,
                                                       uint32_t *counters, int32_t *grid, uint32_t ncid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // (also: grid[cid] = -1 for dm_gather_geo, which runs two launches later — a memset launch less)
    for (uint32_t c = i; c < ncid; c += gridDim.x * blockDim.x) grid[c] = -1;
    if (i >= n) return;
    const int4 cd = code[i];
    const int nx = cd.w & 3, ny = (cd.w >> 2) & 3, nz = (cd.w >> 4) & 3;
    uint32_t o = off[i];
    for (int u = 0; u < nx; ++u)
        for (int v = 0; v < ny; ++v)
            for (int w = 0; w < nz; ++w) {
                uint32_t cid = 0;
                if (!grid_cid(a, cd.x + u, cd.y + v, cd.z + w, cid)) {
                    atomicOr(&counters[kCntError], 1u);  // a point outside the index grid: cannot happen
                    cid = 0;
                }
                keys[o] = cid;
                vals[o] = i;
                ++o;
            }
}

// dm_members_write as the key source of the membership sort's histogram launch (one launch less)
struct MembersSrc {
    const int4 *code;
    PartArgs a;
    const uint32_t *off;
    uint32_t *keys, *vals, *counters;
    int32_t *grid;
    uint32_t ncid;
    __device__ __forceinline__ void begin(uint32_t gtid, uint32_t gsize) const {
        for (uint32_t c = gtid; c < ncid; c += gsize) grid[c] = -1;   // (for dm_gather_geo, two launches later)
    }
    template <class Add>
    __device__ __forceinline__ void operator()(uint32_t i, Add &add) const {
        const int4 cd = code[i];
        const int nx = cd.w & 3, ny = (cd.w >> 2) & 3, nz = (cd.w >> 4) & 3;
        uint32_t o = off[i];
        for (int u = 0; u < nx; ++u)
            for (int v = 0; v < ny; ++v)
                for (int w = 0; w < nz; ++w) {
                    uint32_t cid = 0;
                    if (!grid_cid(a, cd.x + u, cd.y + v, cd.z + w, cid)) {
                        atomicOr(&counters[kCntError], 1u);  // a point outside the index grid: cannot happen
                        cid = 0;
                    }
                    keys[o] = cid;
                    vals[o] = i;
                    add(cid);
                    ++o;
                }
    }
};

// Two jobs of one index space in one launch (round 5; dm_gather_train + dm_geo_fill before):
//   i < n:                  train[i] = xy[vals[i]]   (the training rows in CSR order; train == nullptr: BGK-L gathers rows elsewhere)
//   i < counters[kCntGeo]:  grid[cid] = segment (training block) index — -1 elsewhere, written by dm_members_write
__global__ __launch_bounds__(256) void dm_gather_geo(const float4 *__restrict__ xy, const uint32_t *__restrict__ vals, uint32_t n,
                                                    float4 *train, const uint32_t *__restrict__ seg_key, uint32_t *counters,
                                                    PartArgs a, int32_t *grid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (train && i < n) train[i] = xy[vals[i]];
    if (i < counters[kCntGeo]) grid[seg_key[i]] = (int32_t)i;
}
// counters[kCntTrained] += training blocks (CSR segments) that are in the candidate list.  Its own launch of at most 64 workgroups,
// one atomic each: inside dm_gather_geo it was one atomic per wave on ONE address — ~25 ns per caller, serialised: 65 of that
// kernel's 94 us at configs[4]'s 200 k training blocks (found while measuring the x-slab form, round 6).
__global__ __launch_bounds__(256) void dm_count_trained(const uint32_t *__restrict__ seg_key, PartArgs a, uint32_t *counters) {
    __shared__ uint32_t s_n[4];
    const uint32_t n_geo = counters[kCntGeo];
    uint32_t c = 0;
#pragma unroll 4
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_geo; i += gridDim.x * blockDim.x) {
        const uint32_t cid = seg_key[i];
        const uint32_t z = cid % (uint32_t)a.gn[2], y = (cid / (uint32_t)a.gn[2]) % (uint32_t)a.gn[1],
                       x = cid / ((uint32_t)a.gn[2] * (uint32_t)a.gn[1]);
        c += (a.mult[0][x] && a.mult[1][y] && a.mult[2][z]) ? 1u : 0u;
    }
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63u) == 0u) s_n[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_n[0] + s_n[1] + s_n[2] + s_n[3];
        if (t) atomicAdd(&counters[kCntTrained], t);
    }
}

// ---- x-slab partition of the block-sharded insert (round 6, VERDICT r05 #1d) ---------------------------------------------
// Unsharded, every (block, point) membership pair is sorted into the CSR of ALL training blocks (the largest stage of the
// partition).  A rank of a sharded insert only evaluates a contiguous range of the test list — candidate order is x-major, so
// that is an x-slab of blocks — and needs the training blocks of that range's 7-neighbourhoods only.  What has to stay global is
// small: the per-block point COUNTS (which candidates are test blocks, their weights for the cut, the scan's statistics) — a
// histogram over the dense block grid.  So: counts for all cells (dm_members_hist), candidates / weights / the cut from the
// counts (dm_candidates<true>), then pairs, sort, CSR, rows and neighbour tables for the cells of the own slab only.
// cell_cnt[cid] += 1 for every (block cell, point) pair (zero on entry).  The training set is two voxel-filter outputs, i.e. sorted
// by grid cell: the lanes of a wave mostly share a block, and the sensor's own blocks hold thousands of points — one atomic per
// lane on the x-major cell index: 104 us at configs[4]'s size; matching equal cells by a ballot loop: 55 us (a wave spans ~16 blocks
// along x); the head of every RUN of equal cells in lane order adds the run's length — one ballot, no loop: 62 us, i.e. not the
// number of atomics but WHERE they go: a wave's cells are consecutive in x, ~12 KB apart in the x-major index.  With the count
// array indexed x fastest (grid_cnt_idx) they share cache lines: 20 us.
__global__ __launch_bounds__(256) void dm_members_hist(const int4 *__restrict__ code, uint32_t n, PartArgs a, uint32_t *cell_cnt,
                                                      uint32_t *counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int4 cd = make_int4(0, 0, 0, 0);
    if (i < n) cd = code[i];
    const int nx = cd.w & 3, ny = (cd.w >> 2) & 3, nz = (cd.w >> 4) & 3;
    const int mine = i < n ? nx * ny * nz : 0;
    const uint32_t lane = threadIdx.x & 63u;
    for (int sl = 0; sl < 8; ++sl) {
        uint32_t c = 0xFFFFFFFFu;
        if (sl < mine) {
            const int w = sl % nz, v = (sl / nz) % ny, u = sl / (nz * ny);
            if (!grid_cnt_idx(a, cd.x + u, cd.y + v, cd.z + w, c)) {
                atomicOr(&counters[kCntError], 1u);  // a point outside the index grid: cannot happen
                c = 0xFFFFFFFFu;
            }
        }
        // runs of equal cells in lane order: the head of a run adds its length (the same cell in two runs of a wave: two atomics)
        const uint32_t prev = (uint32_t)__shfl_up((int)c, 1);
        const unsigned long long heads = __ballot(lane == 0u || c != prev);
        if (c != 0xFFFFFFFFu && ((heads >> lane) & 1ull)) {
            const unsigned long long later = lane == 63u ? 0ull : heads >> (lane + 1u);
            const uint32_t len = later ? (uint32_t)__builtin_ctzll(later) + 1u : 64u - lane;
            atomicAdd(&cell_cnt[c], len);
        }
    }
}
// blocks with training points that are in the candidate list (what dm_gather_geo counts from the CSR's segments); at most 64
// workgroups, one atomic each (an atomic on one address costs ~25 ns per caller, serialised)
__global__ __launch_bounds__(256) void dm_cell_trained(const uint32_t *__restrict__ cell_cnt, uint32_t ncid, PartArgs a, uint32_t *counters) {
    __shared__ uint32_t s_n[4];
    uint32_t c = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x + threadIdx.x; base < ncid; base += 8u * stride) {
        uint32_t v[8];   // eight loads in flight per thread (most cells are empty; behind a `continue` the compiler issued them one by one: 20 us)
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) v[u] = base + u * stride < ncid ? cell_cnt[base + u * stride] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            if (!v[u]) continue;
            const uint32_t cid = base + u * stride;   // (cell_cnt is indexed x fastest: grid_cnt_idx)
            const uint32_t x = cid % (uint32_t)a.gn[0], y = (cid / (uint32_t)a.gn[0]) % (uint32_t)a.gn[1],
                           z = cid / ((uint32_t)a.gn[0] * (uint32_t)a.gn[1]);
            c += (a.mult[0][x] && a.mult[1][y] && a.mult[2][z]) ? 1u : 0u;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63u) == 0u) s_n[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_n[0] + s_n[1] + s_n[2] + s_n[3];
        if (t) atomicAdd(&counters[kCntTrained], t);
    }
}

// ---- block-sharded insert: every rank builds the same test list, predicts a contiguous range of it ----
// weight of a test block for the balance = its neighbourhood size (what the kernel streams) + a constant per tile
__global__ void dm_shard_weight(const uint32_t *__restrict__ t_key, uint32_t n_test, uint32_t cap, uint32_t *__restrict__ w) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_test) w[t] = min(test_key_weight(t_key[t]) + 16u, cap);
}
// bounds[q] = first test block of rank q (q = 0..world): the list is cut where the running weight crosses q/world of
// the total — contiguous ranges of the candidate order (x-major block index order: spatially coherent), equal work
__global__ void dm_shard_bounds(const uint32_t *__restrict__ cumw /* exclusive */, const uint32_t *__restrict__ w, uint32_t n_test,
                                uint32_t world, uint32_t *__restrict__ bounds) {
    const uint32_t q = threadIdx.x;
    if (q > world) return;
    const unsigned long long total = (unsigned long long)cumw[n_test - 1] + w[n_test - 1];
    const unsigned long long target = total * q / world;
    uint32_t lo = 0, hi = n_test;  // first t with cumw[t] >= target
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((unsigned long long)cumw[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    bounds[q] = q == world ? n_test : lo;
}
__global__ void dm_shard_leaf_bounds(const uint32_t *__restrict__ bounds, const uint32_t *__restrict__ leaf_off, uint32_t world,
                                     uint32_t *__restrict__ leaf_bounds) {
    const uint32_t q = threadIdx.x;
    if (q <= world) leaf_bounds[q] = leaf_off[bounds[q]];
}
struct CandArgs {
    PartArgs part;
    const int *seq[3];    // float-stepped candidate sequence per axis (biased indices, repeats kept)
    const uint8_t *rank[3];  // occurrence number of each sequence entry among equal indices
    int nseq[3];
    uint32_t pass;        // occurrence number handled by this pass
};

// key -> the 7 ExtendedBlock members (self, +x, -x, +y, -y, +z, -z) by the reference's float path
// (centre = (idx - 524288) * size, neighbour = re-hash of centre +- size), bgkblock.cpp:85-101
__device__ __forceinline__ void extended_indices(int ix, int iy, int iz, float bs, int e[7][3]) {
    const float cx = axis_center(ix, bs), cy = axis_center(iy, bs), cz = axis_center(iz, bs);
    e[0][0] = ix; e[0][1] = iy; e[0][2] = iz;
    const int x0 = (int)axis_index(0 + cx, bs), y0 = (int)axis_index(0 + cy, bs), z0 = (int)axis_index(0 + cz, bs);
    e[1][0] = (int)axis_index(bs + cx, bs);  e[1][1] = y0; e[1][2] = z0;
    e[2][0] = (int)axis_index(-bs + cx, bs); e[2][1] = y0; e[2][2] = z0;
    e[3][0] = x0; e[3][1] = (int)axis_index(bs + cy, bs);  e[3][2] = z0;
    e[4][0] = x0; e[4][1] = (int)axis_index(-bs + cy, bs); e[4][2] = z0;
    e[5][0] = x0; e[5][1] = y0; e[5][2] = (int)axis_index(bs + cz, bs);
    e[6][0] = x0; e[6][1] = y0; e[6][2] = (int)axis_index(-bs + cz, bs);
}

// one thread per entry of the candidate list (x-major, then y, then z — get_blocks_in_bbox order):
// flag = "this entry is a test block of this pass", weight = training points in its 7-neighbourhood
// kCounts (x-slab partition of a sharded insert): `grid` holds the cells' point counts instead of the CSR's segment indices
template <bool kCounts>
__global__ __launch_bounds__(256) void dm_candidates(CandArgs a, const int32_t *__restrict__ grid,
                                                    const uint32_t *__restrict__ train_off, uint32_t n_entries,
                                                    uint32_t *flag, uint32_t *weight) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t kc = e % (uint32_t)a.nseq[2], kb = (e / (uint32_t)a.nseq[2]) % (uint32_t)a.nseq[1],
                   ka = e / ((uint32_t)a.nseq[2] * (uint32_t)a.nseq[1]);
    const int ix = a.seq[0][ka], iy = a.seq[1][kb], iz = a.seq[2][kc];
    const PartArgs &p = a.part;
    const uint32_t my = p.mult[1][iy - p.g0[1]], mz = p.mult[2][iz - p.g0[2]];
    const uint32_t occ = ((uint32_t)a.rank[0][ka] * my + (uint32_t)a.rank[1][kb]) * mz + (uint32_t)a.rank[2][kc];
    int eb[7][3];
    extended_indices(ix, iy, iz, p.bs, eb);
    bool any = false;
    uint32_t w = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        uint32_t cid;
        if (kCounts ? !grid_cnt_idx(p, eb[q][0], eb[q][1], eb[q][2], cid) : !grid_cid(p, eb[q][0], eb[q][1], eb[q][2], cid)) continue;
        const int32_t s = grid[cid];
        if (kCounts ? s == 0 : s < 0) continue;
        any = true;  // the block geometrically holds points (R-tree hit)
        const bool cand = p.mult[0][eb[q][0] - p.g0[0]] && p.mult[1][eb[q][1] - p.g0[1]] && p.mult[2][eb[q][2] - p.g0[2]];
        if (cand) w += kCounts ? (uint32_t)s : train_off[s + 1] - train_off[s];  // ... and was trained (it is in the candidate list)
    }
    flag[e] = (any && occ == a.pass) ? 1u : 0u;
    weight[e] = w;
}

// compacted, ordered list of test entries; sort key = heaviest first (stable)
__global__ __launch_bounds__(256) void dm_test_compact(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ scan,
                                                      const uint32_t *__restrict__ weight, uint32_t n_entries,
                                                      uint32_t *t_key, uint32_t *t_entry, uint32_t *counters) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    if (flag[e]) {
        t_key[scan[e]] = test_key(weight[e]);
        t_entry[scan[e]] = e;
    }
    if (e + 1 == n_entries) counters[kCntTest] = scan[e] + flag[e];
}

// dm_test_compact as the key source of the test list's sort (its histogram launch; the pass behind it reads the list's length
// from counters[kCntTest], so neither waits for the host to learn it)
struct TestCompactSrc {
    const uint32_t *flag, *scan, *weight;
    uint32_t n_entries;
    uint32_t *t_key, *t_entry, *counters;
    __device__ __forceinline__ void begin(uint32_t, uint32_t) const {}
    template <class Add>
    __device__ __forceinline__ void operator()(uint32_t e, Add &add) const {
        if (flag[e]) {
            const uint32_t k = test_key(weight[e]);
            t_key[scan[e]] = k;
            t_entry[scan[e]] = e;
            add(k);
        }
        if (e + 1 == n_entries) counters[kCntTest] = scan[e] + flag[e];
    }
};

// ---- block table (open addressing, linear probing) and pool ------------------------------------------------
constexpr long long kEmptyKey = -1;

__device__ __forceinline__ uint32_t hash_key64(long long k, uint32_t mask) {
    unsigned long long x = (unsigned long long)k;
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    return (uint32_t)x & mask;
}

// find-or-create; new blocks take consecutive pool slots [old count, new count)
struct TableArgs {
    long long *tab_key;
    uint32_t *tab_val;
    uint32_t mask;
    uint32_t *n_blocks;
    long long *blk_key;
};
__device__ __forceinline__ uint32_t table_find_or_create(const TableArgs &tb, long long k) {
    uint32_t h = hash_key64(k, tb.mask);
    for (;;) {
        const long long cur = tb.tab_key[h];
        // the creator may not have published the slot yet only if another thread of THIS launch holds the
        // same key — keys of one pass are distinct, so the value is from an earlier launch
        if (cur == k) return tb.tab_val[h];
        if (cur == kEmptyKey) {
            const long long prev = (long long)atomicCAS((unsigned long long *)&tb.tab_key[h], (unsigned long long)kEmptyKey, (unsigned long long)k);
            if (prev == kEmptyKey) {
                const uint32_t s = atomicAdd(tb.n_blocks, 1u);
                tb.tab_val[h] = s;
                tb.blk_key[s] = k;
                return s;
            }
            if (prev == k) return tb.tab_val[h];
        }
        h = (h + 1) & tb.mask;
    }
}

__global__ __launch_bounds__(256) void dm_table_insert(const long long *__restrict__ keys, const uint32_t *__restrict__ counters,
                                                      long long *tab_key, uint32_t *tab_val, uint32_t mask,
                                                      uint32_t *n_blocks, long long *blk_key, uint32_t *slot) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= counters[kCntTest]) return;
    const TableArgs tb = {tab_key, tab_val, mask, n_blocks, blk_key};
    slot[t] = table_find_or_create(tb, keys[t]);
}

// per test block (heaviest first): key, centre, neighbour table (training-block index or -1), and the block's pool slot
// (find or create, bgkoctomap.cpp:298-305: the same thread-per-block launch did it as a kernel of its own before round 5)
__global__ __launch_bounds__(256) void dm_test_build(CandArgs a, const int32_t *__restrict__ grid,
                                                    const uint32_t *__restrict__ t_entry, const uint32_t *__restrict__ counters,
                                                    long long *t_blockkey, float *center, int32_t *nbr, TableArgs tb, uint32_t *slot) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= counters[kCntTest]) return;
    const uint32_t e = t_entry[t];
    const uint32_t kc = e % (uint32_t)a.nseq[2], kb = (e / (uint32_t)a.nseq[2]) % (uint32_t)a.nseq[1],
                   ka = e / ((uint32_t)a.nseq[2] * (uint32_t)a.nseq[1]);
    const int ix = a.seq[0][ka], iy = a.seq[1][kb], iz = a.seq[2][kc];
    const PartArgs &p = a.part;
    const long long key = ((long long)ix << 40) | ((long long)iy << 20) | (long long)iz;
    t_blockkey[t] = key;
    slot[t] = table_find_or_create(tb, key);
    center[3 * (size_t)t] = axis_center(ix, p.bs);
    center[3 * (size_t)t + 1] = axis_center(iy, p.bs);
    center[3 * (size_t)t + 2] = axis_center(iz, p.bs);
    if (!grid) return;   // (x-slab partition: the CSR does not exist yet — dm_test_nbr writes the own range's tables later)
    int eb[7][3];
    extended_indices(ix, iy, iz, p.bs, eb);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        int32_t s = -1;
        uint32_t cid;
        if (grid_cid(p, eb[q][0], eb[q][1], eb[q][2], cid)) {
            const bool cand = p.mult[0][eb[q][0] - p.g0[0]] && p.mult[1][eb[q][1] - p.g0[1]] && p.mult[2][eb[q][2] - p.g0[2]];
            if (cand) s = grid[cid];
        }
        nbr[7 * (size_t)t + q] = s;
    }
}

// ---- x-slab partition, second half: the own range [t0, t1) of the test list -----------------------------------------------
__device__ __forceinline__ void test_entry_indices(const CandArgs &a, uint32_t e, int &ix, int &iy, int &iz) {
    const uint32_t kc = e % (uint32_t)a.nseq[2], kb = (e / (uint32_t)a.nseq[2]) % (uint32_t)a.nseq[1],
                   ka = e / ((uint32_t)a.nseq[2] * (uint32_t)a.nseq[1]);
    ix = a.seq[0][ka];
    iy = a.seq[1][kb];
    iz = a.seq[2][kc];
}
// range[0] = first, range[1] = last grid cell of the own slab: the test list is in candidate order, x-major with non-decreasing x
// indices (the float-stepped sequence of get_blocks_in_bbox), and the grid's cell index is x-major too — so the own blocks
// [t0, t1) and their face neighbours lie in the y-z planes x_first - 1 .. x_last + 1.  One thread.  (First form: min / max over the
// extended blocks of every own test block — 23 us of f64 index arithmetic for a range that is two look-ups.)
__global__ void dm_slab_range(CandArgs a, const uint32_t *__restrict__ t_entry, uint32_t t0, uint32_t t1, uint32_t *range, uint32_t whole_ncid) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || t1 <= t0) return;   // (empty range: {0xFFFFFFFF, 0} stays — no cell)
    if (whole_ncid) {   // (LA3DM_FORCE_SLAB on an unsharded map: its test list is sorted heaviest first, and its slab is everything)
        range[0] = 0u;
        range[1] = whole_ncid - 1u;
        return;
    }
    int ix0, ix1, iy, iz;
    test_entry_indices(a, t_entry[t0], ix0, iy, iz);
    test_entry_indices(a, t_entry[t1 - 1], ix1, iy, iz);
    const PartArgs &p = a.part;
    const int plane = p.gn[1] * p.gn[2];
    const int xlo = max(min(ix0, ix1) - 1 - p.g0[0], 0), xhi = min(max(ix0, ix1) + 1 - p.g0[0], p.gn[0] - 1);
    range[0] = (uint32_t)xlo * (uint32_t)plane;
    range[1] = (uint32_t)(xhi + 1) * (uint32_t)plane - 1u;
}
// (block, point) pairs of a point whose cell lies in the slab
__global__ __launch_bounds__(256) void dm_members_count_slab(const int4 *__restrict__ code, uint32_t n, PartArgs a,
                                                            const uint32_t *__restrict__ range, uint32_t *cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t lo = range[0], hi = range[1];
    const int4 cd = code[i];
    const int nx = cd.w & 3, ny = (cd.w >> 2) & 3, nz = (cd.w >> 4) & 3;
    uint32_t c = 0;
    for (int u = 0; u < nx; ++u)
        for (int v = 0; v < ny; ++v)
            for (int w = 0; w < nz; ++w) {
                uint32_t cid = 0;
                if (grid_cid(a, cd.x + u, cd.y + v, cd.z + w, cid) && cid >= lo && cid <= hi) ++c;
            }
    cnt[i] = c;
}
// MembersSrc for the slab: the pairs of the cells in [range[0], range[1]] only
struct MembersSlabSrc {
    const int4 *code;
    PartArgs a;
    const uint32_t *off;
    const uint32_t *range;
    uint32_t *keys, *vals;
    int32_t *grid;
    uint32_t ncid;
    __device__ __forceinline__ void begin(uint32_t gtid, uint32_t gsize) const {
        for (uint32_t c = gtid; c < ncid; c += gsize) grid[c] = -1;   // (for dm_gather_geo, two launches later)
    }
    template <class Add>
    __device__ __forceinline__ void operator()(uint32_t i, Add &add) const {
        const uint32_t lo = range[0], hi = range[1];
        const int4 cd = code[i];
        const int nx = cd.w & 3, ny = (cd.w >> 2) & 3, nz = (cd.w >> 4) & 3;
        uint32_t o = off[i];
        for (int u = 0; u < nx; ++u)
            for (int v = 0; v < ny; ++v)
                for (int w = 0; w < nz; ++w) {
                    uint32_t cid = 0;
                    if (!grid_cid(a, cd.x + u, cd.y + v, cd.z + w, cid) || cid < lo || cid > hi) continue;
                    keys[o] = cid;
                    vals[o] = i;
                    add(cid);
                    ++o;
                }
    }
};
// a key source run for its writes alone (the library-sort fallback has no histogram launch to ride on)
template <class Src>
__global__ __launch_bounds__(256) void dm_run_src(Src src, uint32_t n_items) {
    src.begin(blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    auto add = [](uint32_t) {};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) src(i, add);
}
// neighbour tables (training-block index of the slab's CSR, or -1) of the test blocks [t0, t1)
__global__ __launch_bounds__(256) void dm_test_nbr(CandArgs a, const int32_t *__restrict__ grid, const uint32_t *__restrict__ t_entry,
                                                  uint32_t t0, uint32_t t1, int32_t *nbr) {
    const uint32_t t = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= t1) return;
    int ix, iy, iz;
    test_entry_indices(a, t_entry[t], ix, iy, iz);
    const PartArgs &p = a.part;
    int eb[7][3];
    extended_indices(ix, iy, iz, p.bs, eb);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        int32_t s = -1;
        uint32_t cid;
        if (grid_cid(p, eb[q][0], eb[q][1], eb[q][2], cid)) {
            const bool cand = p.mult[0][eb[q][0] - p.g0[0]] && p.mult[1][eb[q][1] - p.g0[1]] && p.mult[2][eb[q][2] - p.g0[2]];
            if (cand) s = grid[cid];
        }
        nbr[7 * (size_t)t + q] = s;
    }
}

// ---- block table: rebuild, search, pool ---------------------------------------------------------------
__global__ __launch_bounds__(256) void dm_table_rebuild(const long long *__restrict__ blk_key, uint32_t n, long long *tab_key,
                                                       uint32_t *tab_val, uint32_t mask) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const long long k = blk_key[s];
    uint32_t h = hash_key64(k, mask);
    for (;;) {
        const long long prev = (long long)atomicCAS((unsigned long long *)&tab_key[h], (unsigned long long)kEmptyKey, (unsigned long long)k);
        if (prev == kEmptyKey) {
            tab_val[h] = s;
            return;
        }
        h = (h + 1) & mask;
    }
}

// BGKOctoMap::search(point) for a batch of query points, straight from the device pool (bgkoctomap.cpp:554-567 ->
// Block::search, bgkblock.cpp:152-160): block by hash of the point, finest-layer voxel by the clamped cell index
// (child bit 4 = +x, 2 = +y, 1 = +z per level); a missing block answers with a default node.
__global__ __launch_bounds__(256) void dm_search(const float *__restrict__ q, uint32_t n, const long long *__restrict__ tab_key,
                                                const uint32_t *__restrict__ tab_val, uint32_t mask, const float *__restrict__ A,
                                                const float *__restrict__ B, const uint8_t *__restrict__ S, uint32_t npb,
                                                uint32_t block_depth, float bs, float resolution, float a0, float b0,
                                                uint8_t *exists, float *oA, float *oB, uint8_t *ostate) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = q[3 * (size_t)i], y = q[3 * (size_t)i + 1], z = q[3 * (size_t)i + 2];
    const long long ix = axis_index(x, bs), iy = axis_index(y, bs), iz = axis_index(z, bs);
    const long long key = (ix << 40) | (iy << 20) | iz;
    uint32_t h = hash_key64(key, mask);
    long long cur;
    while ((cur = tab_key[h]) != key && cur != kEmptyKey) h = (h + 1) & mask;
    if (cur != key) {
        exists[i] = 0;
        oA[i] = a0;
        oB[i] = b0;
        ostate[i] = kStateUnknown;
        return;
    }
    const int cells = 1 << (block_depth - 1);
    const float c[3] = {axis_center(ix, bs), axis_center(iy, bs), axis_center(iz, bs)}, v[3] = {x, y, z};
    int ci[3];
    for (int a = 0; a < 3; ++a) {
        const int t = (int)floorf((v[a] - c[a]) / resolution + cells / 2.0f);
        ci[a] = max(0, min(t, cells - 1));
    }
    uint32_t index = 0;
    for (int level = (int)block_depth - 2; level >= 0; --level)
        index = index * 8u + (uint32_t)((((ci[0] >> level) & 1) << 2) | (((ci[1] >> level) & 1) << 1) | ((ci[2] >> level) & 1));
    const size_t node = (size_t)tab_val[h] * npb + dm_layer_base(block_depth - 1) + index;
    exists[i] = 1;
    oA[i] = A[node];
    oB[i] = B[node];
    ostate[i] = S[node] & 7u;
}

// default nodes of the blocks created by the last dm_table_insert — slots [old_blocks, *new_blocks):
// Occupancy() = {prior A, prior B, UNKNOWN, not classified}
__global__ __launch_bounds__(256) void dm_pool_init(float *A, float *B, uint8_t *S, uint32_t old_blocks,
                                                   const uint32_t *__restrict__ new_blocks, uint32_t npb, float a0, float b0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t count = (size_t)(*new_blocks - old_blocks) * npb;
    if (i >= count) return;
    const size_t first = (size_t)old_blocks * npb;
    A[first + i] = a0;
    B[first + i] = b0;
    S[first + i] = kStateUnknown;
}

// Leaf of the finest-layer cell c of one block: climb while the node is PRUNED (a collapsed sibling group
// lives on in its parent).  Returns the node offset inside the block and the leaf's OcTreeHashKey.
__device__ __forceinline__ void covering_leaf(const uint8_t *__restrict__ Sb, uint32_t depth_last, uint32_t c, uint32_t &d,
                                              uint32_t &i) {
    d = depth_last;
    i = c;
    while (d > 0 && (Sb[dm_layer_base(d) + i] & 7u) == kStatePruned) {
        --d;
        i >>= 3;
    }
}

// Leaves of the test blocks in LeafIterator order (descending DFS = descending finest-cell interval).
// One wave per block, 64 finest cells per trip from the top; kEmit = false counts, true writes.
// Round 5 (launch chain): the counting launch also gives the blocks the pass created — slots >= old_blocks, each exactly
// once in a pass's list — their default nodes (Occupancy() = {prior A, prior B, UNKNOWN, not classified}; dm_pool_init was a
// launch of its own), and such a block's leaves are its finest cells; the emitting launch carries the pass's work counters
// in `stat_wgs` extra workgroups behind the block waves (dm_test_stats was a launch of its own): sum over test blocks of
// their neighbourhood size and of neighbourhood size x leaf count, 64-bit words inside the counter block.
struct LeafExtra {
    uint32_t old_blocks;    // count launch: blocks with slot >= old_blocks are new (0xFFFFFFFF: none are initialised here)
    float a0, b0;
    float *A_w, *B_w;       // (writable views of the pool for the count launch)
    uint8_t *S_w;
    const uint32_t *t_key;  // emit launch: keys of the test blocks (weights), or nullptr: no work counters
    uint32_t *counters_w;
    uint32_t main_wgs;      // emit launch: workgroups that hold block waves; the rest accumulate the counters
    uint32_t t_begin, t_end;  // emit launch: the test blocks [t_begin, min(t_end, count)) only (block-sharded insert: the rank's own
                              // range — the other ranks' leaves arrive with their keys); the count launch takes every block
};

__device__ __forceinline__ void test_stats_wg(const uint32_t *__restrict__ t_key, const uint32_t *__restrict__ nleaf, uint32_t n_test,
                                              uint32_t *counters, uint32_t wg, uint32_t n_wg) {
    // grid-stride partial sums, one pair of 64-bit atomics per workgroup (a few dozen in all: the per-wave version
    // serialised ~1300 atomics on two addresses)
    __shared__ unsigned long long s_w[4], s_pw[4];
    unsigned long long *acc_reads = reinterpret_cast<unsigned long long *>(counters + kCntTrainReads);
    unsigned long long *acc_pairs = reinterpret_cast<unsigned long long *>(counters + kCntPairEvals);
    unsigned long long w = 0, pw = 0;
    for (uint32_t t = wg * blockDim.x + threadIdx.x; t < n_test; t += n_wg * blockDim.x) {
        const unsigned long long wt = test_key_weight(t_key[t]);
        w += wt;
        pw += wt * nleaf[t];
    }
    for (int o = 32; o >= 1; o >>= 1) {
        w += __shfl_xor(w, o);
        pw += __shfl_xor(pw, o);
    }
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = w;
        s_pw[threadIdx.x >> 6] = pw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        w = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        pw = s_pw[0] + s_pw[1] + s_pw[2] + s_pw[3];
        if (w | pw) {
            atomicAdd(acc_reads, w);
            atomicAdd(acc_pairs, pw);
        }
    }
}

template <bool kEmit>
__global__ __launch_bounds__(256) void dm_leaves(const uint32_t *__restrict__ slot, const uint32_t *__restrict__ counters,
                                                const uint8_t *__restrict__ S, const float *__restrict__ A,
                                                const float *__restrict__ B, uint32_t npb, uint32_t block_depth,
                                                uint32_t *nleaf, const uint32_t *__restrict__ leaf_off, uint32_t *leaf_key,
                                                float *alpha, float *beta, uint32_t *leaf_node, LeafExtra x) {
    if (kEmit && x.t_key && blockIdx.x >= x.main_wgs) {   // (uniform over the workgroup)
        test_stats_wg(x.t_key, nleaf, counters[kCntTest], x.counters_w, blockIdx.x - x.main_wgs, gridDim.x - x.main_wgs);
        return;
    }
    const uint32_t t = (kEmit ? x.t_begin : 0u) + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (!kEmit && blockIdx.x == 0 && threadIdx.x == 0) nleaf[counters[kCntTest]] = 0;   // the scan runs over n_test + 1 counts
    if (t >= counters[kCntTest] || (kEmit && t >= x.t_end)) return;
    const size_t base = (size_t)slot[t] * npb;
    const uint8_t *Sb = S + base;
    const uint32_t dl = block_depth - 1, ncell = 1u << (3 * dl);
    if (!kEmit && slot[t] >= x.old_blocks) {   // created by this pass: default nodes, every finest cell a leaf
        for (uint32_t i = lane; i < npb; i += 64) {
            x.A_w[base + i] = x.a0;
            x.B_w[base + i] = x.b0;
            x.S_w[base + i] = kStateUnknown;
        }
        if (lane == 0) nleaf[t] = ncell;
        return;
    }
    uint32_t out = kEmit ? leaf_off[t] : 0u;
    for (uint32_t top = ncell; top > 0; top -= min(top, 64u)) {
        const bool in = (uint32_t)lane < top;
        const uint32_t c = in ? top - 1u - lane : 0u;  // lane order = descending cell index
        uint32_t d, i;
        covering_leaf(Sb, dl, c, d, i);
        const uint32_t span = 3u * (dl - d);
        const bool head = in && c == (((i + 1u) << span) - 1u);  // highest cell of the leaf's interval
        const unsigned long long m = __ballot(head);
        if (kEmit && head) {
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            const uint32_t node = dm_layer_base(d) + i;
            leaf_key[out + r] = (d << 16) + i;
            alpha[out + r] = A[base + node];
            beta[out + r] = B[base + node];
            leaf_node[out + r] = (uint32_t)(base + node);
        }
        out += (uint32_t)__popcll(m);
    }
    if (!kEmit && lane == 0) nleaf[t] = out;
}

// write-back of the leaves Occupancy::update ran for (state bit 7)
__global__ __launch_bounds__(256) void dm_commit(const uint32_t *__restrict__ counters, const uint32_t *__restrict__ leaf_node,
                                                const float *__restrict__ alpha, const float *__restrict__ beta,
                                                const uint8_t *__restrict__ state, float *A, float *B, uint8_t *S) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= counters[kCntLeaves]) return;
    const uint8_t st = state[l];
    if (!(st & 0x80u)) return;
    const uint32_t node = leaf_node[l];
    A[node] = alpha[l];
    B[node] = beta[l];
    S[node] = (uint8_t)((st & 3u) | kClassifiedBit);
}

// OcTree::prune for the test blocks (bgkoctree.cpp:101-148): bottom-up, a sibling group whose eight members
// share one non-UNKNOWN state collapses into its parent (alpha, beta, state of child 0 — `classified` is not
// carried by the node copy), the children become PRUNED.  One wave per block; the states are staged in LDS
// (the layers depend on each other), and a collapsed chain parent <- child 0 <- ... is resolved to its
// bottom node first so that alpha/beta are copied once from nodes this launch never writes.
// Dynamic LDS: 4 waves x prune_lds_stride(npb) bytes.
__host__ __device__ inline uint32_t prune_lds_stride(uint32_t npb) { return (((npb + 1u) & ~1u) + 2u * npb + 15u) & ~15u; }

__global__ __launch_bounds__(256) void dm_prune(const uint32_t *__restrict__ slot, uint32_t n_test, float *A, float *B,
                                               uint8_t *S, uint32_t npb, uint32_t block_depth) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dm_prune_smem[];
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * (blockDim.x >> 6) + wv;
    const int lane = threadIdx.x & 63;
    if (t >= n_test || slot[t] == 0xFFFFFFFFu) return;  // (0xFFFFFFFF: an entry the caller masked out)
    uint8_t *sS = dm_prune_smem + wv * prune_lds_stride(npb);
    uint16_t *src = (uint16_t *)(sS + ((npb + 1u) & ~1u));
    const size_t base = (size_t)slot[t] * npb;
    for (uint32_t i = lane; i < npb; i += 64) {
        sS[i] = S[base + i];
        src[i] = (uint16_t)i;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool any = false;
    for (int depth = (int)block_depth - 1; depth > 0; --depth) {
        const uint32_t lb = dm_layer_base(depth), pb = dm_layer_base(depth - 1), ngroup = 1u << (3 * (depth - 1));
        for (uint32_t g = lane; g < ngroup; g += 64) {
            const uint32_t c0 = lb + 8u * g;
            const uint8_t st0 = sS[c0] & 7u;
            if (st0 == kStatePruned || st0 == kStateUnknown) continue;
            bool same = true;
#pragma unroll
            for (int c = 1; c < 8; ++c) same &= (sS[c0 + c] & 7u) == st0;
            if (!same) continue;
            const uint32_t par = pb + g;
            sS[par] = (uint8_t)((sS[par] & kClassifiedBit) | st0);
            src[par] = src[c0];
#pragma unroll
            for (int c = 0; c < 8; ++c) sS[c0 + c] = (uint8_t)((sS[c0 + c] & ~7u) | kStatePruned);
            any = true;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (!__any(any)) return;
    for (uint32_t i = lane; i < npb; i += 64) {
        S[base + i] = sS[i];
        const uint32_t sr = src[i];
        if (sr != i) {
            A[base + i] = A[base + sr];
            B[base + i] = B[base + sr];
        }
    }
}

// dm_commit_prune's arrival words: top level at d_mm[kArriveBase], bucket k at d_mm[kArriveBase + kArriveStride (k + 1)] — a 128-byte
// line each (atomics on one line serialise like atomics on one word)
constexpr uint32_t kArriveBuckets = 64, kArriveStride = 32, kArriveBase = 32, kMmWords = kArriveBase + kArriveStride * (1 + kArriveBuckets);
constexpr uint32_t kCommitPruneWgs = 2048;   // workgroups of the launch at most (each one arrival): grid-stride over the test blocks

// Write-back + prune of a pass in ONE launch (round 5; single-pass scans: a pass's test blocks are distinct, and a block's
// leaves live in that block only, so the wave that owns block t commits the leaves [leaf_off[t], leaf_off[t + 1]) and prunes
// the block behind them).  The committed states go to the pool AND to the wave's LDS copy of the block, the sibling test runs on
// that copy; alpha / beta of a collapsed chain are read back from nodes this wave may just have written — after its stores have
// retired (vmcnt) and past the vector L1 (device-scope loads).  Bit for bit what dm_commit + dm_prune leave.  `done` (1 + kArriveBuckets words, zero
// between launches): the last workgroup to finish also publishes the counter block (mailbox != nullptr) instead of a
// dm_publish_counters launch behind this one.
__global__ __launch_bounds__(256) void dm_commit_prune(const uint32_t *__restrict__ slot, uint32_t n_test,
                                                      const uint32_t *__restrict__ leaf_off, const uint32_t *__restrict__ leaf_node,
                                                      const uint32_t *__restrict__ leaf_key, const float *__restrict__ alpha, const float *__restrict__ beta,
                                                      const uint8_t *__restrict__ state, float *A, float *B, uint8_t *S, uint32_t npb,
                                                      uint32_t block_depth, uint32_t *counters, uint32_t *done,
                                                      volatile uint32_t *mailbox, uint32_t mailbox_seq) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dm_prune_smem[];
    __shared__ uint32_t s_last;
    const uint32_t wv = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    for (uint32_t t = blockIdx.x * (blockDim.x >> 6) + wv; t < n_test; t += gridDim.x * (blockDim.x >> 6)) {
        uint8_t *sS = dm_prune_smem + wv * prune_lds_stride(npb);
        uint16_t *src = (uint16_t *)(sS + ((npb + 1u) & ~1u));
        const size_t base = (size_t)slot[t] * npb;
        for (uint32_t i = lane; i < npb; i += 64) {
            sS[i] = S[base + i];
            src[i] = (uint16_t)i;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t l0 = leaf_off[t], l1 = leaf_off[t + 1];
        for (uint32_t l = l0 + lane; l < l1; l += 64) {
            const uint8_t st = state[l];
            if (!(st & 0x80u)) continue;
            // (leaf_key != nullptr — block-sharded insert: a foreign leaf arrives with its key {depth << 16 | index}, pool slots are
            // numbered per replica — the node is found from this replica's slot of the block)
            uint32_t node;
            if (leaf_key) {
                const uint32_t key = leaf_key[l];
                node = (uint32_t)base + dm_layer_base(key >> 16) + (key & 0xFFFFu);
            } else {
                node = leaf_node[l];
            }
            const uint8_t ns = (uint8_t)((st & 3u) | kClassifiedBit);
            A[node] = alpha[l];
            B[node] = beta[l];
            S[node] = ns;
            sS[node - (uint32_t)base] = ns;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool any = false;
        for (int depth = (int)block_depth - 1; depth > 0; --depth) {
            const uint32_t lb = dm_layer_base(depth), pb = dm_layer_base(depth - 1), ngroup = 1u << (3 * (depth - 1));
            for (uint32_t g = lane; g < ngroup; g += 64) {
                const uint32_t c0 = lb + 8u * g;
                const uint8_t st0 = sS[c0] & 7u;
                if (st0 == kStatePruned || st0 == kStateUnknown) continue;
                bool same = true;
#pragma unroll
                for (int c = 1; c < 8; ++c) same &= (sS[c0 + c] & 7u) == st0;
                if (!same) continue;
                const uint32_t par = pb + g;
                sS[par] = (uint8_t)((sS[par] & kClassifiedBit) | st0);
                src[par] = src[c0];
#pragma unroll
                for (int c = 0; c < 8; ++c) sS[c0 + c] = (uint8_t)((sS[c0 + c] & ~7u) | kStatePruned);
                any = true;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (__any(any)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's committed alpha / beta have left for the L2
            for (uint32_t i = lane; i < npb; i += 64) {
                S[base + i] = sS[i];
                const uint32_t sr = src[i];
                if (sr != i) {
                    const uint32_t av = __hip_atomic_load((const uint32_t *)&A[base + sr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t bv = __hip_atomic_load((const uint32_t *)&B[base + sr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    A[base + i] = __uint_as_float(av);
                    B[base + i] = __uint_as_float(bv);
                }
            }
        }
    }
    if (!mailbox) return;   // (uniform)
    // arrival in two levels — kArriveBuckets words, then one: ten thousand returning atomics on ONE address serialise at ~25 ns each
    // (and so do atomics on one 128-byte line).  done[0] = top level, done[kArriveStride (1 + k)] = bucket k; all zero
    // between launches (the last arrival of a level resets its word).
    __syncthreads();
    if (threadIdx.x == 0) {
        // (no release fence: on this multi-L2 part it writes the whole L2 back — 10 000 of them made this launch 270 us.  The mailbox only
        // tells the host that every workgroup has got here; whatever reads the pool afterwards is stream-ordered behind the launch.)
        const uint32_t k = blockIdx.x % kArriveBuckets;
        const uint32_t expect_k = (gridDim.x - k + kArriveBuckets - 1u) / kArriveBuckets;
        uint32_t last = 0u;
        uint32_t *bucket = done + kArriveStride * (1u + k);
        if (atomicAdd(bucket, 1u) + 1u == expect_k) {
            *bucket = 0u;
            const uint32_t expect_top = min(gridDim.x, kArriveBuckets);
            if (atomicAdd(&done[0], 1u) + 1u == expect_top) {
                done[0] = 0u;
                last = 1u;
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last) {
        // ... and leaves the counter block as dm_begin would (the pool's block count stays): the next insert of a map whose last
        // insert ended here starts without that launch.  (The min / max words and the arrival words reset themselves.)
        uint32_t v = 0;
        if (threadIdx.x < (uint32_t)kCntWords) v = __hip_atomic_load(&counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dm_publish_wave(counters, mailbox, mailbox_seq);
        if (threadIdx.x < (uint32_t)kCntWords) counters[threadIdx.x] = counter_begin_value(threadIdx.x, v);
    }
}


// ---------------------------------------------------------------------------------------------------------
// Leaf export (row f3): the publish loop of the static node, src/bgkoctomap/bgkoctomap_static_node.cpp:101-136,
// with the cube-list bookkeeping of MarkerArrayPub::insert_point3d and heightMapColor
// (include/common/markerarray_pub.h:21-147), minus ROS: every leaf in `want_state` becomes a cell
// {x, y, z, size}, a colour and a marker level; with original_size == 0 a collapsed leaf is expanded into the
// base-resolution cells of get_pruned_locs (bgkoctomap.h:269-287: float-stepped loops, kept verbatim).
// One wave per pool block, cells in LeafIterator order; kEmit = false counts, true writes at blk_off[slot].
// ---------------------------------------------------------------------------------------------------------
struct ExportArgs {
    const long long *blk_key;
    const uint8_t *S;
    const float *A;
    const float *B;
    const float4 *lut;
    uint32_t *blk_cnt;         // [n_blocks + 1] (count pass; the last entry stays 0)
    const uint32_t *blk_off;   // exclusive scan of blk_cnt
    float4 *cells;
    float4 *rgba;
    int32_t *level;
    uint32_t n_blocks, npb, depth;
    int want_state, original, variant, coloured;
    float block_size, resolution, min_z, max_z, gp_l, gp_max_ivar;
    float size_of_depth[8];    // Block::get_size per layer (host expression)
    int level_of_depth[8];     // (int) log2(size / resolution)
};

__device__ inline void height_map_color(double h, float4 &c) {  // markerarray_pub.h:21-76 (s = v = 1)
    h -= floor(h);
    h *= 6;
    const int i = (int)floor(h);
    double f = h - i;
    if (!(i & 1)) f = 1 - f;
    const double v = 1.0, m = 0.0, n = 1.0 - f;
    double r, g, b;
    switch (i) {
    case 6:
    case 0: r = v; g = n; b = m; break;
    case 1: r = n; g = v; b = m; break;
    case 2: r = m; g = v; b = n; break;
    case 3: r = m; g = n; b = v; break;
    case 4: r = n; g = m; b = v; break;
    case 5: r = v; g = m; b = n; break;
    default: r = 1; g = 0.5; b = 0.5; break;
    }
    c = make_float4((float)r, (float)g, (float)b, 1.0f);
}

__device__ inline float4 export_colour(const ExportArgs &a, float z, float A, float B) {
    if (a.want_state == 1) {  // insert_point3d(x, y, z, min_z, max_z, size): coloured by height when min_z < max_z
        if (!a.coloured) return make_float4(0.0f, 0.0f, 1.0f, 1.0f);  // the marker's default colour
        const double h = (1.0 - (double)fminf(fmaxf((z - a.min_z) / (a.max_z - a.min_z), 0.0f), 1.0f)) * 0.8;
        float4 c;
        height_map_color(h, c);
        return c;
    }
    float prob;  // insert_point3d(..., prob)
    if (a.variant == 1) prob = 1.0f / (1.0f + (float)exp((double)(-a.gp_l * A / a.gp_max_ivar)));
    else prob = A / (A + B);
    if (prob < 0.5f) return make_float4(0.8f, 0.8f, 0.8f, 1.0f);
    float4 c;
    height_map_color(fmin(2.0 - 2.0 * (double)prob, 0.6), c);
    return c;
}

template <bool kEmit>
__global__ __launch_bounds__(256) void dm_export(ExportArgs a) {
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slot >= a.n_blocks) return;
    const size_t base = (size_t)slot * a.npb;
    const uint8_t *Sb = a.S + base;
    const long long key = a.blk_key[slot];
    const float cx = axis_center(key >> 40, a.block_size), cy = axis_center((key >> 20) & 0xFFFFF, a.block_size),
                cz = axis_center(key & 0xFFFFF, a.block_size);
    const uint32_t dl = a.depth - 1, ncell = 1u << (3 * dl);
    uint32_t out = kEmit ? a.blk_off[slot] : 0u;
    for (uint32_t top = ncell; top > 0; top -= min(top, 64u)) {
        const bool in = (uint32_t)lane < top;
        const uint32_t c = in ? top - 1u - lane : 0u;  // lane order = descending cell index = LeafIterator order
        uint32_t d, i;
        covering_leaf(Sb, dl, c, d, i);
        const uint32_t span = 3u * (dl - d);
        const uint32_t node = dm_layer_base(d) + i;
        const bool head = in && c == (((i + 1u) << span) - 1u) && (int)(Sb[node] & 7u) == a.want_state;
        // cells of this leaf
        float lx = 0.f, ly = 0.f, lz = 0.f, x0 = 0.f, y0 = 0.f, z0 = 0.f, x1 = 0.f, y1 = 0.f, z1 = 0.f;
        uint32_t nx = 1, ny = 1, nz = 1;
        const float size = a.size_of_depth[d];
        if (head) {
            const float4 o = a.lut[node];
            lx = o.x + cx;
            ly = o.y + cy;
            lz = o.z + cz;
            if (!a.original) {
                x0 = (float)((double)lx - (double)size * 0.5 + (double)a.resolution * 0.5);
                y0 = (float)((double)ly - (double)size * 0.5 + (double)a.resolution * 0.5);
                z0 = (float)((double)lz - (double)size * 0.5 + (double)a.resolution * 0.5);
                x1 = (float)((double)lx + (double)size * 0.5);
                y1 = (float)((double)ly + (double)size * 0.5);
                z1 = (float)((double)lz + (double)size * 0.5);
                nx = ny = nz = 0;
                for (float x = x0; x < x1; x += a.resolution) ++nx;
                for (float y = y0; y < y1; y += a.resolution) ++ny;
                for (float z = z0; z < z1; z += a.resolution) ++nz;
            }
        }
        const uint32_t mine = head ? nx * ny * nz : 0u;
        uint32_t incl = mine;  // inclusive prefix over the lanes
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t up = __shfl_up(incl, sft, 64);
            if (lane >= sft) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        if (kEmit && mine) {
            uint32_t w = out + incl - mine;
            const float A = a.A[base + node], B = a.B[base + node];
            if (a.original) {
                a.cells[w] = make_float4(lx, ly, lz, size);
                a.rgba[w] = export_colour(a, lz, A, B);
                a.level[w] = a.level_of_depth[d];
            } else {
                for (float x = x0; x < x1; x += a.resolution)
                    for (float y = y0; y < y1; y += a.resolution)
                        for (float z = z0; z < z1; z += a.resolution) {
                            a.cells[w] = make_float4(x, y, z, a.resolution);
                            a.rgba[w] = export_colour(a, z, A, B);
                            a.level[w] = 0;  // (int) log2(resolution / resolution)
                            ++w;
                        }
            }
        }
        out += total;
    }
    if (!kEmit && lane == 0) a.blk_cnt[slot] = out;
}

// index box of the pool's blocks (get_bbox without a download): mm[0..2] = min, mm[3..5] = max of the 20-bit indices
__global__ void dm_key_bounds(const long long *__restrict__ blk_key, uint32_t n, uint32_t *mm) {
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const long long k = blk_key[i];
        const uint32_t v[3] = {(uint32_t)(k >> 40) & 0xFFFFFu, (uint32_t)(k >> 20) & 0xFFFFFu, (uint32_t)k & 0xFFFFFu};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], v[a]);
            hi[a] = max(hi[a], v[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            lo[a] = min(lo[a], (uint32_t)__shfl_down(lo[a], sft, 64));
            hi[a] = max(hi[a], (uint32_t)__shfl_down(hi[a], sft, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&mm[a], lo[a]);
            atomicMax(&mm[3 + a], hi[a]);
        }
    }
}

}  // namespace la3dm_dev
