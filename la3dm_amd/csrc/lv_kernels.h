// lv_kernels.h — HIP kernel (gfx950, wave64) for the BGKLVOctoMap per-voxel path.
//
// Reference (CPU):
//   leaf loop of insert_pointcloud   src/bgklvoctomap/bgklvoctomap.cpp:155-244
//   point_to_line_dist               include/bgklvoctomap/bgklvinference.h:100-134 (dots and b = c1/c2 in double)
//   covSparseLine                    include/bgklvoctomap/bgklvinference.h:143-156 (r = min(d/ell, 1), no < 0 clamp)
//   LV Occupancy                     src/bgklvoctomap/bgklvoctree_node.cpp:29-77
//
// One workgroup = one 4x4x4 cube of base-resolution voxels = 64 consecutive nodes of the finest layer =
// one bucket of the gather grid.  Lane = voxel.  The workgroup walks the (2r+1)^3 buckets around its own
// z-major; a bucket's samples are loaded 64 per wave (lane = sample, coalesced 16 B), culled against
// the cube's +-ell box, ballot-compacted in order into LDS together with what the de-duplication needs
// (previous sample of the same ray, segment end points), and then every lane tests each staged sample
// against its own closed +-ell box.  A hit adds one row; a ray adds one row at its lowest-index sample
// inside the box: because each coordinate of a ray's samples is monotone along the ray (also in fp32),
// the samples inside a box are one contiguous run, so "lowest index" <=> neither the ray's first sample
// nor the previous sample is inside.  Rows are summed in gather order = the CPU restatement's order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgk_kernels.h"

namespace la3dm_dev {

struct LvArgs {
    const float4 *samples;   // original order: x, y, z, ray (float)
    const float4 *sorted;    // bucket order: x, y, z, original index (int bits)
    const float4 *rays;      // 2 x float4 per segment: {start, first sample index bits}, {end, 0}
    const uint32_t *cell_off;
    const float *blk_center;
    const int32_t *blk_cell0;
    float *alpha;
    float *beta;
    uint8_t *state;
    const float4 *lut;       // finest layer at lut_base
    int32_t cell_min[3];
    int32_t cell_dim[3];
    uint32_t lut_base;
    uint32_t nodes_per_blk;  // 8^(depth-1)
    uint32_t cubes_shift;    // log2(cubes per block)
    uint32_t cubes_bits;     // bits per axis of the cube coordinate inside a block
    uint32_t n_tasks;
    int32_t reach;           // r = ceil(ell / g)
    float sf2, ell, free_thresh, occupied_thresh, var_thresh, min_W;
    // pool mode (device-resident map, devmap.hip lv_insert): the node arrays are the block pool itself — block b lives at
    // blk_slot[b] * npb + layer_off, the state byte carries the HOST enum (PRUNED 3, UNCERTAIN 4) under the classified bit,
    // a voxel that saw samples gets bit 6 until dm_lv_finish clears it, blocks whose candidate key repeats fewer than
    // pass + 1 times sit the pass out, and updates are counted into *upd_counter.
    const uint32_t *blk_slot;   // nullptr = packed mode (the la3dm_lv_scan arrays)
    const uint32_t *blk_mult;
    uint32_t *upd_counter;
    uint32_t npb, layer_off, pass;
    float inv_ell;              // RN(1 / ell) or 0 (bgk_kernels.h div_by_ell)
    int trig;                   // "fast_trig": 0 correctly rounded (default), 3 = Eigen 3.3.7 psin / pcos (sincos_eigen337)
    struct LvCand *cand;        // [n_samples] in bucket order: everything the voxel kernel needs of a sample (bgklv_cand_kernel)
    uint32_t n_samples;
    // work plan (bgklv_plan_kernel): workgroup -> cube, and the row scratch of the cubes that are split over workgroups
    uint32_t *sub_task;         // workgroup -> cube: [0, max_subs) the sub-tasks of split cubes, [max_subs, 2 max_subs) the other cubes
    uint32_t max_subs;
    uint32_t *task_first, *task_nsub, *task_row0, *split_list;
    uint32_t *plan_totals;      // [0] workgroups of split cubes, [1] scratch rows, [2] split cubes, [3] workgroups of the other cubes
    const uint32_t *n_blk_dev;  // plan kernel: number of packed blocks when it is still on the device (else n_tasks counts)
    float *rows;                // [row slot][64 voxels] k; only rows with a non-zero k are written
    unsigned long long *sub_nz; // per workgroup: kLvChunk bits, row slot written
    unsigned long long *sub_y;  // per workgroup: kLvChunk bits, the row's sample is a hit (y = 1)
    unsigned long long *sub_info;   // per workgroup: voxels that saw a sample in their box
    double2 *sub_part;          // order-free mode: per workgroup of a split cube, {sum k y, sum k} of its part of the stream per voxel
};

// LV state code <-> the host enum stored in the pool (State::PRUNED = 3, State::UNCERTAIN = 4; la3dm_lv_scan uses the
// reference's numbering UNCERTAIN 3, PRUNED 4: src/bgklvoctomap/bgklvoctree_node.h:11-13)
__device__ __forceinline__ uint8_t lv_from_pool(uint8_t s) { s &= 7u; return s == 3u ? 4u : (s == 4u ? 3u : s); }
__device__ __forceinline__ uint8_t lv_to_pool(uint8_t s) { return s == 3u ? 4u : (s == 4u ? 3u : s); }

// point3f::norm(): double sqrt of a float sum, narrowed where the reference stores it in a float matrix
__device__ __forceinline__ float norm3f(float x, float y, float z) { return (float)sqrt((double)(x * x + y * y + z * z)); }

// include/bgklvoctomap/bgklvinference.h:104-131, as the reference writes it (double sqrt / double division).  Only the
// numerics test uses this form; the kernel evaluates lv_seg_point + lv_kernel_at below, which return the same bits.
__device__ __forceinline__ float seg_dist_dev(float px, float py, float pz, float ax, float ay, float az, float bx, float by,
                                              float bz) {
    const float lx = bx - ax, ly = by - ay, lz = bz - az;
    const float line_len = norm3f(lx, ly, lz);
    const float vx = px - ax, vy = py - ay, vz = pz - az;
    if (line_len < 0.0001f) return norm3f(vx, vy, vz);
    const double c1 = (double)(vx * lx + vy * ly + vz * lz);
    const double c2 = (double)(lx * lx + ly * ly + lz * lz);
    if (c1 <= 0) return norm3f(vx, vy, vz);
    if (c2 <= c1) return norm3f(px - bx, py - by, pz - bz);
    const float b = (float)(c1 / c2);
    const float nx = ax + lx * b, ny = ay + ly * b, nz = az + lz * b;
    return norm3f(px - nx, py - ny, pz - nz);
}

__device__ __forceinline__ float cov_sparse_line_dev(float d, float ell, float sf2) {
    float r = d / ell;
    if (r > 1.0f) r = 1.0f;
    return cov_sparse<false, 0>(r, sf2);
}

// The same function in fp32 only.  (float)sqrt((double)x) == sqrtf(x) for every fp32 x and (float)((double)a / (double)b)
// == a / b for fp32 a, b with a normal quotient: rounding an exact square root / quotient to 53 bits and then to 24 is
// the same as rounding it to 24 bits at once whenever the wide format has at least 2 * 24 + 2 bits (Figueroa, "When is
// double rounding innocuous?", 1995); la3dm_diag_sweep(what = 4) checks the square root for all 2^31 non-negative fp32
// inputs on the device, what = 9 a few billion quotients, and a quotient below 2^-100 takes the double division.
// Returns the point of the segment a + t (b - a), t in [0, 1], the distance is measured to; l = b - a.
__device__ __forceinline__ void lv_seg_point(float px, float py, float pz, float ax, float ay, float az, float bx, float by, float bz,
                                             float lx, float ly, float lz, float &qx, float &qy, float &qz) {
    const float vx = px - ax, vy = py - ay, vz = pz - az;
    const float c1 = vx * lx + vy * ly + vz * lz;
    const float c2 = lx * lx + ly * ly + lz * lz;
    qx = ax; qy = ay; qz = az;
    if (c1 <= 0.0f) return;
    if (c2 <= c1) {
        qx = bx; qy = by; qz = bz;
        return;
    }
    float b = c1 / c2;
    if (b < 0x1p-100f) b = (float)((double)c1 / (double)c2);
    qx = ax + lx * b; qy = ay + ly * b; qz = az + lz * b;
}
// point_to_line_dist as one fp32 function (the segment's direction recomputed per call): the same bits as seg_dist_dev
__device__ __forceinline__ float seg_dist_f32(float px, float py, float pz, float ax, float ay, float az, float bx, float by, float bz) {
    const float lx = bx - ax, ly = by - ay, lz = bz - az;
    float qx = ax, qy = ay, qz = az;
    if (!(sqrtf(lx * lx + ly * ly + lz * lz) < 0.0001f)) lv_seg_point(px, py, pz, ax, ay, az, bx, by, bz, lx, ly, lz, qx, qy, qz);
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

// covSparseLine at distance |p - q| (bgklvinference.h:143-156: r = min(d / ell, 1), no clamp of negative values)
__device__ __forceinline__ float lv_kernel_at(float px, float py, float pz, float qx, float qy, float qz, float ell, float inv_ell,
                                              float sf2, int trig = 0) {
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    float r = div_by_ell(d, ell, inv_ell);
    if (r > 1.0f) r = 1.0f;
    return trig == 3 ? cov_sparse_fast<3, false>(r, sf2) : cov_sparse_fast<0, false>(r, sf2);   // (wave-uniform)
}

// src/bgklvoctomap/bgklvoctree_node.cpp:29-63
__device__ __forceinline__ float lv_prob_dev(float A, float B, float min_W) {
    const float W = (A + B < min_W) ? min_W : A + B;
    if (A > B) return (float)((double)(A / (W - B)) + (double)(W - A - B) * 0.5 / (double)(W - B));
    return (float)(0.5 * (double)(W - B - A) / (double)(W - A));
}
__device__ __forceinline__ float lv_var_dev(float A, float B, float min_W, float prob) {
    const float W = (A + B < min_W) ? min_W : A + B;
    const double a = (double)(1 - prob), b = 0.5 - (double)prob, c = (double)prob;
    return (float)((double)(A / W) * (a * a) + (double)((W - A - B) / W) * (b * b) + (double)(B / W) * (c * c));
}

// A staged sample.  type (p.w): 0 = hit, 1 = first sample of its ray, 2 = later sample of a ray, + 4 = the ray's segment
// is shorter than 0.1 mm (point_to_line_dist then measures to the segment start).  The segment direction l = end - start
// rides in the w components.
struct __attribute__((aligned(16))) LvCand {
    float4 p;     // sample position, type
    float4 prev;  // previous sample of the same ray (type 2), l.x
    float4 r0;    // segment start (= the ray's first sample), l.y
    float4 r1;    // segment end, l.z
};

// One workgroup of kLvWaves waves per cube, or per sub-task of a cube.  The samples the cube has to look at are the
// concatenation, z-major, of the buckets around its own (the "stream"); the workgroup walks its part of the stream 512
// positions at a time: every wave stages its own 64 samples (lane = sample: locate the bucket, load, cull against the
// cube's +-ell box, ordered compaction across the waves through LDS), then the staged candidates are evaluated kLvWaves
// at a time — wave w takes candidates w, w + kLvWaves, ... of a 64-candidate round, lane = voxel, and writes k (or +0)
// into a dense [candidate][voxel] tile — and wave 0 adds the tile row by row: the evaluations spread over the CU's four
// SIMDs while the two running sums keep the gather order (adding +0 leaves a sum that started at +0 unchanged: it can
// never be -0; k * y with y in {0, 1} is k or a zero).
//
// A cube next to the sensor sees every beam: one workgroup for it would run ten times longer than the rest of the grid
// together.  bgklv_plan_kernel therefore cuts a cube whose stream is longer than kLvChunk into sub-tasks of kLvChunk
// stream positions; the sub-tasks of such a cube write their tile rows to a scratch array instead of adding them, and
// bgklv_split_add_kernel adds the rows of all sub-tasks in stream order — the same additions in the same order.
constexpr int kLvWaves = 8;
constexpr uint32_t kLvChunk = 512;
constexpr uint32_t kLvGroup = 64;   // buckets whose ranges are resident in LDS at a time

// kF64 (order-free accumulate mode, round 5): no [candidate][voxel] tile, no row maps — the E phase adds straight into the
// voxels' double accumulators
template <bool kF64>
struct LvLdsT {
    LvCand cand[kLvWaves * kWave];
    float k[kF64 ? 1 : kWave][kWave];
    unsigned long long y_round[2];               // rows of the current round that are hits with a counted voxel (by round parity)
    unsigned long long cmask[kWave];             // per candidate of the round: the voxels that count it
    unsigned long long nz_map[kF64 ? 1 : kLvChunk / kWave];  // split cube: the same per staged candidate of the sub-task
    unsigned long long y_map[kF64 ? 1 : kLvChunk / kWave];
    uint32_t cnt[kLvWaves];
    unsigned long long info[kLvWaves];
    float ybar[kWave];
    uint32_t g_c0[kLvGroup];
    uint32_t g_incl[kLvGroup];
    double acc_k[kF64 ? kWave : 1];              // order-free mode: sum k per voxel ...
    double acc_y[kF64 ? kWave : 1];              // ... and sum k y (the hit rows: y = 1)
};
typedef LvLdsT<false> LvLds;

struct LvTask {
    uint32_t blk, cube, node;
    bool in_range, pool, active;
    size_t ni;
    uint8_t st_raw;
    float cx, cy, cz;
};

__device__ __forceinline__ LvTask lv_task(const LvArgs &a, uint32_t task, int lane) {
    LvTask t;
    t.blk = task >> a.cubes_shift;
    t.cube = task & ((1u << a.cubes_shift) - 1u);
    t.node = t.cube * kWave + lane;
    t.in_range = t.node < a.nodes_per_blk;
    t.pool = a.blk_slot != nullptr;
    t.ni = t.pool ? (size_t)a.blk_slot[t.blk] * a.npb + a.layer_off + (t.in_range ? t.node : 0)
                  : (size_t)t.blk * a.nodes_per_blk + (t.in_range ? t.node : 0);
    t.st_raw = t.in_range ? a.state[t.ni] : (uint8_t)4;
    const uint8_t st_in = !t.in_range ? (uint8_t)4 : (t.pool ? lv_from_pool(t.st_raw) : t.st_raw);
    t.active = st_in != 4;
    return t;
}

// bucket coordinates (relative to the gather grid) of a cube: the octree index interleaves (x, y, z) bits, coarsest first
__device__ __forceinline__ void lv_cube_cell(const LvArgs &a, uint32_t blk, uint32_t cube, int &gx, int &gy, int &gz) {
    int bxc = 0, byc = 0, bzc = 0;
    for (uint32_t lvl = 0; lvl < a.cubes_bits; ++lvl) {
        const uint32_t sh = 3u * (a.cubes_bits - 1u - lvl);
        const uint32_t oct = (cube >> sh) & 7u;
        bxc = (bxc << 1) | (int)((oct >> 2) & 1u);
        byc = (byc << 1) | (int)((oct >> 1) & 1u);
        bzc = (bzc << 1) | (int)(oct & 1u);
    }
    gx = a.blk_cell0[3 * blk + 0] + bxc - a.cell_min[0];
    gy = a.blk_cell0[3 * blk + 1] + byc - a.cell_min[1];
    gz = a.blk_cell0[3 * blk + 2] + bzc - a.cell_min[2];
}

// sample range of bucket number j (z-major in the (2 reach + 1)^3 neighbourhood of (gx, gy, gz)); empty outside the grid
__device__ __forceinline__ void lv_bucket_range(const LvArgs &a, int gx, int gy, int gz, uint32_t j, uint32_t nb, uint32_t &c0,
                                                uint32_t &c1) {
    c0 = c1 = 0;
    if (j >= nb) return;
    const uint32_t w = 2u * (uint32_t)a.reach + 1u;
    const int dx = (int)(j % w) - a.reach, dy = (int)((j / w) % w) - a.reach, dz = (int)(j / (w * w)) - a.reach;
    const int x = gx + dx, y = gy + dy, z = gz + dz;
    if (x < 0 || y < 0 || z < 0 || x >= a.cell_dim[0] || y >= a.cell_dim[1] || z >= a.cell_dim[2]) return;
    const uint32_t cell = ((uint32_t)z * a.cell_dim[1] + (uint32_t)y) * a.cell_dim[0] + (uint32_t)x;
    c0 = a.cell_off[cell];
    c1 = a.cell_off[cell + 1];
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, kWave);
        if (lane >= d) v += o;
    }
    return v;
}

// The rows of a [row][voxel] tile whose bit is set in m, added to acc in row order: the row numbers come out of the mask
// on the scalar unit, eight LDS reads are in flight at a time (one read per dependent add would cost the LDS latency per
// row).  Skipped rows are rows of zeros, or rows that add +0 to this sum: x + 0 = x for every sum that started at +0.
__device__ __forceinline__ float lv_add_rows(const float (*k)[kWave], unsigned long long m, int lane, float acc) {
    while (m) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = m ? __builtin_ctzll(m) : 0;
            v[i] = m ? k[j][lane] : 0.0f;
            m &= m - 1ull;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    }
    return acc;
}

// rows [0, n16) of a tile added to acc in row order, n16 a multiple of 16 (the rows past the last real one hold zeros):
// constant LDS offsets, 16 reads in flight — a fifth of the masked walk's cost per row, so the sum over all rows (most
// of them non-zero) goes this way and only the sparse hit rows use the mask.
__device__ __forceinline__ float lv_add_dense(const float (*k)[kWave], uint32_t n16, int lane, float acc) {
    for (uint32_t j = 0; j < n16; j += 16) {
        float v[16];
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) v[i] = k[j + i][lane];
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) acc += v[i];
    }
    return acc;
}

// The voxel kernel's view of the samples, in bucket order, one 64-byte record each (built once per scan): position and
// type, the previous sample of the same ray, the ray's segment and its direction.
__global__ __launch_bounds__(256) void bgklv_cand_kernel(LvArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_samples) return;
    const float4 s = a.sorted[i];
    const uint32_t orig = __float_as_uint(s.w);
    LvCand c;
    c.p = make_float4(s.x, s.y, s.z, 0.0f);
    c.prev = c.r0 = c.r1 = c.p;
    const int ray = orig < a.n_samples ? (int)a.samples[orig].w : -1;
    if (ray >= 0) {
        const float4 ra = a.rays[2 * ray], rb = a.rays[2 * ray + 1];
        const uint32_t first = __float_as_uint(ra.w);
        const float lx = rb.x - ra.x, ly = rb.y - ra.y, lz = rb.z - ra.z;
        float type = orig == first ? 1.0f : 2.0f;
        if (sqrtf(lx * lx + ly * ly + lz * lz) < 0.0001f) type += 4.0f;
        float4 pv = c.p;
        if (orig != first) pv = a.samples[orig - 1];
        c.p.w = type;
        c.prev = make_float4(pv.x, pv.y, pv.z, lx);
        c.r0 = make_float4(ra.x, ra.y, ra.z, ly);
        c.r1 = make_float4(rb.x, rb.y, rb.z, lz);
    }
    a.cand[i] = c;
}

// position of the r-th set bit (r = 0: the lowest) of a 64-bit mask that has more than r bits set
__device__ __forceinline__ uint32_t lv_nth_bit(unsigned long long m, uint32_t r) {
    uint32_t x = (uint32_t)m, pos = 0;
    const uint32_t c0 = (uint32_t)__popc(x);
    if (r >= c0) {
        r -= c0;
        x = (uint32_t)(m >> 32);
        pos = 32;
    }
#pragma unroll
    for (uint32_t w = 16; w >= 1; w >>= 1) {
        const uint32_t c = (uint32_t)__popc(x & ((1u << w) - 1u));
        if (r >= c) {
            r -= c;
            x >>= w;
            pos += w;
        }
    }
    return pos;
}

// Work plan: cubes without a base-resolution leaf or (pool mode) without a sample in reach get no workgroup; the others
// ceil(stream / kLvChunk).  A wave looks at 16 cubes one after the other (lane = voxel for the status bytes, lane = bucket
// for the stream length) and then hands out workgroup numbers, scratch rows and list places for all of them with one
// atomic each — their order is irrelevant, every cube's result is a function of its own rows alone.
// totals: [0] workgroups of split cubes, [1] scratch rows, [2] split cubes, [3] workgroups of the other cubes.
constexpr uint32_t kLvPlanPerWave = 16;
__global__ __launch_bounds__(256) void bgklv_plan_kernel(LvArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t n_tasks = a.n_blk_dev ? *a.n_blk_dev << a.cubes_shift : a.n_tasks;
    const uint32_t task0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * kLvPlanPerWave;   // (may lie beyond n_tasks: the wave idles)
    const uint32_t w = 2u * (uint32_t)a.reach + 1u, nb = w * w * w;
    uint32_t my_nsub = 0, my_stream = 0;   // lane i: cube task0 + i
    for (uint32_t i = 0; i < kLvPlanPerWave && task0 + i < n_tasks; ++i) {
        const LvTask t = lv_task(a, task0 + i, lane);
        uint32_t nsub = 0, stream = 0;
        if (__any(t.active)) {
            int gx, gy, gz;
            lv_cube_cell(a, t.blk, t.cube, gx, gy, gz);
            for (uint32_t j = lane; j < nb; j += kWave) {
                uint32_t c0, c1;
                lv_bucket_range(a, gx, gy, gz, j, nb, c0, c1);
                stream += c1 - c0;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) stream += (uint32_t)__shfl_xor((int)stream, d, kWave);
            nsub = (stream + kLvChunk - 1) / kLvChunk;
        }
        if (!t.pool && nsub == 0) nsub = 1;   // packed mode: the workgroup still clears the cube's status bytes
        if (lane == (int)i) {
            my_nsub = nsub;
            my_stream = nsub > 1 ? stream : 0u;
        }
    }
    const bool is_split = my_nsub > 1;
    const uint32_t sub_incl = wave_incl_scan_u32(my_nsub, lane), row_incl = wave_incl_scan_u32(my_stream, lane);
    const unsigned long long sm = __ballot(is_split);
    // one atomic per total and WORKGROUP (an atomic on one address costs ~25 ns per caller, serialised: per wave they
    // were two thirds of this kernel's time): the four waves' totals meet in LDS, thread 0 draws, the waves take their parts
    __shared__ uint32_t s_tot[4][4], s_base[4];
    const uint32_t wave = threadIdx.x >> 6;
    // the sub-tasks of split cubes are numbered first: a launch hands its workgroups out in order, and a split cube's
    // workgroup runs several times longer than the average one — started last, it was the launch's tail
    const uint32_t heavy_incl = wave_incl_scan_u32(is_split ? my_nsub : 0u, lane);
    const uint32_t light_incl = wave_incl_scan_u32(my_nsub == 1u ? 1u : 0u, lane);
    if (lane == kWave - 1) {
        s_tot[wave][0] = heavy_incl;
        s_tot[wave][1] = row_incl;
        s_tot[wave][2] = (uint32_t)__popcll(sm);
        s_tot[wave][3] = light_incl;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t t = 0;
        for (uint32_t v = 0; v < 4; ++v) t += s_tot[v][threadIdx.x];
        s_base[threadIdx.x] = t ? atomicAdd(a.plan_totals + threadIdx.x, t) : 0u;
    }
    __syncthreads();
    uint32_t base_heavy = s_base[0], base_row = s_base[1], base_split = s_base[2], base_light = s_base[3];
    for (uint32_t v = 0; v < wave; ++v) {
        base_heavy += s_tot[v][0];
        base_row += s_tot[v][1];
        base_split += s_tot[v][2];
        base_light += s_tot[v][3];
    }
    (void)sub_incl;
    if (lane >= (int)kLvPlanPerWave || task0 + lane >= n_tasks) return;
    const uint32_t task = task0 + lane;
    a.task_nsub[task] = my_nsub;
    if (my_nsub == 0) return;
    if (is_split) {
        const uint32_t first = base_heavy + heavy_incl - my_nsub;
        a.task_first[task] = first;
        for (uint32_t s2 = 0; s2 < my_nsub; ++s2) a.sub_task[first + s2] = task;
        a.task_row0[task] = base_row + row_incl - my_stream;
        a.split_list[base_split + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull))] = task;
    } else {
        a.sub_task[a.max_subs + base_light + light_incl - 1u] = task;
    }
}

// fold the voxel's sums into its node (bgklvoctomap.cpp:236-238 + LV Occupancy); wave 0 of the workgroup that owns the sums
__device__ __forceinline__ void lv_commit(const LvArgs &a, const LvTask &t, float ybar, float kbar, bool info) {
    if (!t.in_range) return;
    uint8_t out = info ? 0x40u : 0u;
    bool updated = false;
    if (t.active && info && kbar > 0.001f) {
        updated = true;
        float A = a.alpha[t.ni], B = a.beta[t.ni];
        A += ybar;
        B += kbar - ybar;
        const float prob = lv_prob_dev(A, B, a.min_W);
        const float var = lv_var_dev(A, B, a.min_W, prob);
        uint8_t st;
        if (var > a.var_thresh) st = 3;
        else st = prob > a.occupied_thresh ? 1 : (prob < a.free_thresh ? 0 : 2);
        a.alpha[t.ni] = A;
        a.beta[t.ni] = B;
        out |= (uint8_t)(0x80u | (t.pool ? lv_to_pool(st) : st));
    }
    if (!t.pool) {
        a.state[t.ni] = out;
        return;
    }
    // pool: an untouched node keeps its byte (state + classified), plus the transient "saw samples" bit
    a.state[t.ni] = updated ? out : (uint8_t)(t.st_raw | (out & 0x40u));
    const unsigned long long um = __ballot(updated);
    if ((threadIdx.x & 63) == 0 && um) atomicAdd(a.upd_counter, (uint32_t)__popcll(um));
}

// kF64 = the order-free accumulate mode ("bgk_sum" 1, the default since round 5): the reference's two fp32 running sums in
// gather order (bgklvinference.h:80-83) become double sums of the same fp32 kv and kv * y, rounded to fp32 once per voxel
// (the restatement's set_sum_mode(1)); the gate kbar > 0.001f and the LV node update are unchanged.  Nothing is ordered any
// more, so the E phase's lanes add their k straight into the voxel's accumulators (ds_add_f64, a native LDS atomic), the
// [candidate][voxel] tile, its zeroing and the A phase disappear, and a split cube's workgroups hand 1 KB of partial sums
// to bgklv_split_apply64 instead of their rows to bgklv_split_add_kernel (3.2x the algorithmic bytes in round 4).
template <bool kF64>
__device__ __forceinline__ void bgklv_voxel_body(const LvArgs &a, LvLdsT<kF64> &L);
template <bool kF64>
__global__ __launch_bounds__(kLvWaves *kWave) __attribute__((amdgpu_waves_per_eu(6, 6))) void bgklv_voxel_kernel(LvArgs a) {
    __shared__ LvLdsT<kF64> L;
    bgklv_voxel_body<kF64>(a, L);
}
// the order-free form without the 16 KB tile: four workgroups fit a CU's LDS, at 8 waves per SIMD (64 VGPRs)
__global__ __launch_bounds__(kLvWaves *kWave) __attribute__((amdgpu_waves_per_eu(8, 8))) void bgklv_voxel_kernel_w8(LvArgs a) {
    __shared__ LvLdsT<true> L;
    bgklv_voxel_body<true>(a, L);
}
template <bool kF64>
__device__ __forceinline__ void bgklv_voxel_body(const LvArgs &a, LvLdsT<kF64> &L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t task = blockIdx.x, sub = 0, nsub = 1;
    if (a.sub_task) {
        const uint32_t n_heavy = a.plan_totals[0];
        if (blockIdx.x < n_heavy) {
            task = a.sub_task[blockIdx.x];
            sub = blockIdx.x - a.task_first[task];
            nsub = a.task_nsub[task];
        } else {
            task = a.sub_task[a.max_subs + (blockIdx.x - n_heavy)];
        }
    }
    if (task >= a.n_tasks) return;
    const LvTask t = lv_task(a, task, lane);
    if (t.pool && a.blk_mult[t.blk] <= a.pass) return;  // (uniform) this block's key repeats fewer times than the pass number
    const bool active = t.active;
    const float4 off4 = a.lut[a.lut_base + (t.in_range ? t.node : 0)];
    const float cx = off4.x + a.blk_center[3 * t.blk + 0], cy = off4.y + a.blk_center[3 * t.blk + 1],
                cz = off4.z + a.blk_center[3 * t.blk + 2];
    // the voxel's closed query box (bgklvoctomap.cpp:162-166)
    const float lox = cx - a.ell, loy = cy - a.ell, loz = cz - a.ell;
    const float hix = cx + a.ell, hiy = cy + a.ell, hiz = cz + a.ell;
    const float inf = __builtin_inff();
    const float tlx = wave_min_dpp(active ? lox : inf), tly = wave_min_dpp(active ? loy : inf), tlz = wave_min_dpp(active ? loz : inf);
    const float thx = wave_max_dpp(active ? hix : -inf), thy = wave_max_dpp(active ? hiy : -inf), thz = wave_max_dpp(active ? hiz : -inf);
    // the intersection of the active voxels' boxes: a later sample of a ray whose previous sample, or whose ray's first
    // sample, lies in there is nobody's lowest-index sample in the box — it adds no row; and the voxels' "saw a sample"
    // flags do not need it either: that previous / first sample is itself in this cube's stream and in every box
    const float clx = wave_max_dpp(active ? lox : -inf), cly = wave_max_dpp(active ? loy : -inf), clz = wave_max_dpp(active ? loz : -inf);
    const float chx = wave_min_dpp(active ? hix : inf), chy = wave_min_dpp(active ? hiy : inf), chz = wave_min_dpp(active ? hiz : inf);
    if (threadIdx.x < 2) L.y_round[threadIdx.x] = 0ull;
    if constexpr (kF64) {
        if (threadIdx.x < kWave) L.acc_k[threadIdx.x] = L.acc_y[threadIdx.x] = 0.0;
    } else {
        if (threadIdx.x < kLvChunk / kWave) L.nz_map[threadIdx.x] = L.y_map[threadIdx.x] = 0ull;
    }
    __syncthreads();  // every wave has read the cube's states before wave 0 may rewrite them
    if (!(tlx <= thx)) {  // no base-resolution leaf in this cube (uniform over the workgroup)
        if (wave == 0 && t.in_range && !t.pool) a.state[t.ni] = 0;
        return;
    }
    int gx, gy, gz;
    lv_cube_cell(a, t.blk, t.cube, gx, gy, gz);
    const bool split = nsub > 1;
    const uint32_t lo = split ? sub * kLvChunk : 0u, hi = split ? lo + kLvChunk : 0xFFFFFFFFu;
    const size_t row0 = split ? (size_t)a.task_row0[task] + lo : 0;
    const uint32_t wdt = 2u * (uint32_t)a.reach + 1u, nb = wdt * wdt * wdt;

    float ybar = 0.0f, kbar = 0.0f;   // kbar lives in wave 0, ybar in wave 1
    bool info = false;
    uint32_t n_rows = 0;   // candidates staged so far (split: the next row slot)
    uint32_t rnd = 0;
    uint32_t pos0 = 0;     // stream position of the bucket group's first sample
    for (uint32_t g0 = 0; g0 < nb && pos0 < hi; g0 += kLvGroup) {
        // ranges of the group's buckets (every wave computes them, wave 0 publishes them)
        uint32_t c0, c1;
        lv_bucket_range(a, gx, gy, gz, g0 + lane, nb, c0, c1);
        const uint32_t incl = wave_incl_scan_u32(c1 - c0, lane);
        const uint32_t gtotal = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (wave == 0) {
            L.g_c0[lane] = c0 - (incl - (c1 - c0));   // sample index = g_c0[j] + position inside the group
            L.g_incl[lane] = incl;
        }
        __syncthreads();
        const uint32_t s0 = lo > pos0 ? lo - pos0 : 0u, s1 = min(hi - pos0, gtotal);   // my part of the group, group-relative
        for (uint32_t base = s0; base < s1; base += kLvWaves * kWave) {
            // stage: lane = stream position, wave w takes [base + 64 w, base + 64 w + 64)
            const uint32_t q = base + (uint32_t)wave * kWave + lane;
            bool keep = false;
            LvCand c;
            c.p = c.prev = c.r0 = c.r1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < s1) {
                uint32_t j = 0;   // first bucket whose inclusive count exceeds q
#pragma unroll
                for (uint32_t step = kLvGroup / 2; step > 0; step >>= 1)
                    if (L.g_incl[j + step - 1] <= q) j += step;
                const uint32_t si = L.g_c0[j] + q;
                // one 64-byte record (was: sorted -> samples -> rays -> previous sample, four dependent loads), as four 16-byte
                // loads (copied as a struct, part of it went through scratch memory: a store, a wait, a load and a wait per sample)
                const float4 *cp = reinterpret_cast<const float4 *>(a.cand + si);
                c.p = cp[0];
                c.prev = cp[1];
                c.r0 = cp[2];
                c.r1 = cp[3];
                keep = !(tlx > c.p.x || c.p.x > thx || tly > c.p.y || c.p.y > thy || tlz > c.p.z || c.p.z > thz);
                if (keep && ((int)c.p.w & 3) == 2) {
                    const bool prev_core = !(clx > c.prev.x || c.prev.x > chx || cly > c.prev.y || c.prev.y > chy || clz > c.prev.z || c.prev.z > chz);
                    const bool first_core = !(clx > c.r0.x || c.r0.x > chx || cly > c.r0.y || c.r0.y > chy || clz > c.r0.z || c.r0.z > chz);
                    if (prev_core || first_core) keep = false;
                }
            }
            const unsigned long long m = __ballot(keep);
            const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (lane == 0) L.cnt[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int v = 0; v < kLvWaves; ++v) {
                const uint32_t cv = L.cnt[v];
                before += v < wave ? cv : 0u;
                total += cv;
            }
            if (keep) {
                float4 *lp = reinterpret_cast<float4 *>(&L.cand[before + slot]);
                lp[0] = c.p;
                lp[1] = c.prev;
                lp[2] = c.r0;
                lp[3] = c.r1;
            }
            __syncthreads();
            // Rounds of 64 staged candidates.  (T) every wave tests eight candidates against the 64 voxels (lane = voxel):
            // which voxels count it — as a hit in the box, or as its ray's lowest-index sample in the box.  (E) the counting
            // (candidate, voxel) pairs — one in eight or so — are enumerated densely, lane = pair, and only they pay for the
            // segment projection and the kernel; the values land in a [candidate][voxel] tile of zeros.  (A) wave 0 adds
            // the tile's rows in order into the k sums, wave 1 the hit rows into the k y sums — or, for a split cube, the
            // rows with a counting voxel go out to the scratch array.
            for (uint32_t r0 = 0; r0 < total; r0 += kWave) {
                const uint32_t nr = min(total - r0, (uint32_t)kWave);
                const uint32_t nr16 = split ? nr : (nr + 15u) & ~15u;
                // ---- T
                for (uint32_t j = wave; j < kWave; j += kLvWaves) {
                    unsigned long long cm = 0ull;
                    int type = 0;
                    if (j < nr) {
                        const LvCand &cd = L.cand[r0 + j];
                        const float4 p = cd.p;
                        type = __builtin_amdgcn_readfirstlane((int)p.w);
                        const bool inb = active && !(lox > p.x || p.x > hix || loy > p.y || p.y > hiy || loz > p.z || p.z > hiz);
                        bool count = inb;
                        if ((type & 3) == 2 && __any(inb)) {  // a later sample of a ray: is it the ray's lowest-index sample in my box?
                            const float4 q0 = cd.r0, pv = cd.prev;
                            const bool first_in = !(lox > q0.x || q0.x > hix || loy > q0.y || q0.y > hiy || loz > q0.z || q0.z > hiz);
                            const bool prev_in = !(lox > pv.x || pv.x > hix || loy > pv.y || pv.y > hiy || loz > pv.z || pv.z > hiz);
                            count = inb && !first_in && !prev_in;
                        }
                        info |= inb;
                        cm = __ballot(count);
                    }
                    if constexpr (!kF64) {
                        if (j < nr16) L.k[j][lane] = 0.0f;
                    }
                    if (lane == 0) {
                        L.cmask[j] = cm;
                        if (!kF64 && cm != 0ull && type == 0 && !split) atomicOr(&L.y_round[rnd & 1u], 1ull << j);
                    }
                }
                __syncthreads();
                // ---- E
                {
                    const unsigned long long mine = L.cmask[lane];                     // lane = candidate
                    const uint32_t cj = (uint32_t)__popcll(mine);
                    const uint32_t incl = wave_incl_scan_u32(cj, lane);
                    const uint32_t excl = incl - cj;
                    const uint32_t n_pairs = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    for (uint32_t e = (uint32_t)wave * kWave + lane; e - lane < n_pairs; e += kLvWaves * kWave) {
                        const bool valid = e < n_pairs;
                        // the candidate of pair e: the last one whose exclusive count is <= e
                        uint32_t jc = 0;
#pragma unroll
                        for (uint32_t step = 32; step >= 1; step >>= 1) {
                            const uint32_t t2 = (uint32_t)__shfl((int)excl, (int)(jc + step), kWave);
                            if (t2 <= e) jc += step;
                        }
                        const unsigned long long cmj = valid ? L.cmask[jc] : 1ull;
                        const uint32_t ex_j = (uint32_t)__shfl((int)excl, (int)jc, kWave);
                        const uint32_t vox = lv_nth_bit(cmj, valid ? e - ex_j : 0u);
                        const float vx = __shfl(cx, (int)vox, kWave), vy = __shfl(cy, (int)vox, kWave), vz = __shfl(cz, (int)vox, kWave);
                        if (valid) {
                            const LvCand &cd = L.cand[r0 + jc];
                            const float4 p = cd.p;
                            const int ty = (int)p.w;
                            float qx = p.x, qy = p.y, qz = p.z;   // a hit: the distance to the sample itself
                            if (ty != 0) {
                                const float4 q0 = cd.r0, q1 = cd.r1;
                                const float lx = cd.prev.w;
                                qx = q0.x; qy = q0.y; qz = q0.z;
                                if (!(ty & 4)) lv_seg_point(vx, vy, vz, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, lx, q0.w, q1.w, qx, qy, qz);
                            }
                            const float kv = lv_kernel_at(vx, vy, vz, qx, qy, qz, a.ell, a.inv_ell, a.sf2, a.trig);
                            if constexpr (kF64) {
                                const uint32_t ak = (uint32_t)(uintptr_t)&L.acc_k[vox], ay = (uint32_t)(uintptr_t)&L.acc_y[vox];
                                asm volatile("ds_add_f64 %0, %1\n" : : "v"(ak), "v"((double)kv) : "memory");
                                // kv * y: y = 1 for a hit, 0 for a ray's row (adds +-0: nothing)
                                if (ty == 0) asm volatile("ds_add_f64 %0, %1\n" : : "v"(ay), "v"((double)kv) : "memory");
                            } else {
                                L.k[jc][vox] = kv;
                            }
                        }
                    }
                }
                __syncthreads();
                if constexpr (kF64) continue;   // (the barrier above also separates this round's reads of cmask / cand from the next T)
                // ---- A
                if (!split) {
                    if (wave == 0) kbar = lv_add_dense(L.k, nr16, lane, kbar);
                    if (wave == 1) ybar = lv_add_rows(L.k, L.y_round[rnd & 1u], lane, ybar);
                    // the other parity's masks: last read in the previous round's add phase, next set after this round's
                    // last barrier
                    if (threadIdx.x == 2 * kWave) L.y_round[(rnd + 1u) & 1u] = 0ull;
                    ++rnd;
                } else {
                    for (uint32_t j = wave; j < nr; j += kLvWaves) {
                        if (L.cmask[j] == 0ull) continue;   // a row of zeros adds nothing to either sum: not written, not read
                        const uint32_t slot = n_rows + r0 + j;
                        a.rows[(row0 + slot) * kWave + lane] = L.k[j][lane];
                        if (lane == 0) {
                            atomicOr(&L.nz_map[slot >> 6], 1ull << (slot & 63u));
                            if ((int)L.cand[r0 + j].p.w == 0) atomicOr(&L.y_map[slot >> 6], 1ull << (slot & 63u));
                        }
                    }
                }
                __syncthreads();
            }
            n_rows += total;
        }
        pos0 += gtotal;
        __syncthreads();   // the group's ranges are read until here
    }
    const unsigned long long im = __ballot(info);
    if (lane == 0) L.info[wave] = im;
    if (wave == 1) L.ybar[lane] = ybar;
    __syncthreads();
    if (wave != 0) return;
    unsigned long long all = 0;
#pragma unroll
    for (int v = 0; v < kLvWaves; ++v) all |= L.info[v];
    if constexpr (kF64) {
        const double ys = L.acc_y[lane], ks = L.acc_k[lane];
        if (split) {
            a.sub_part[(size_t)blockIdx.x * kWave + lane] = make_double2(ys, ks);
            if (lane == 0) a.sub_info[blockIdx.x] = all;
            return;
        }
        lv_commit(a, t, (float)ys, (float)ks, (all >> lane) & 1ull);
        return;
    }
    if (split) {
        constexpr uint32_t kWords = kLvChunk / kWave;
        if (lane < (int)kWords) {
            a.sub_nz[(size_t)blockIdx.x * kWords + lane] = L.nz_map[lane];
            a.sub_y[(size_t)blockIdx.x * kWords + lane] = L.y_map[lane];
        }
        if (lane == 0) a.sub_info[blockIdx.x] = all;
        return;
    }
    lv_commit(a, t, L.ybar[lane], kbar, (all >> lane) & 1ull);
}

// order-free mode: the partial sums of a split cube's workgroups, added in sub-task order, rounded once, committed
__global__ __launch_bounds__(kWave) void bgklv_split_apply64(LvArgs a) {
    const int lane = threadIdx.x;
    const uint32_t task = a.split_list[blockIdx.x];
    const LvTask t = lv_task(a, task, lane);
    if (t.pool && a.blk_mult[t.blk] <= a.pass) return;
    const uint32_t first = a.task_first[task], nsub = a.task_nsub[task];
    double ys = 0.0, ks = 0.0;
    unsigned long long info = 0;
    for (uint32_t s = 0; s < nsub; ++s) {
        info |= a.sub_info[first + s];
        const double2 v = a.sub_part[(size_t)(first + s) * kWave + lane];
        ys += v.x;
        ks += v.y;
    }
    lv_commit(a, t, (float)ys, (float)ks, (info >> lane) & 1ull);
}

// The rows of a split cube, added in stream order: sub-task after sub-task, row after row (rows of zeros were not
// written and are not read).  Per sub-task the set bits of its row map become a list of row slots; the eight waves fetch
// the listed rows 128 at a time, 16 B per lane, three tiles ahead of the one wave 0 adds into the k sum and wave 1 (hit
// rows only) into the k y sum — the rows come from other CUs' stores, ~2 us away, and one tile in flight would leave the
// adder waiting for memory most of the time.
constexpr uint32_t kLvAddRows = 128;
constexpr int kLvAddDepth = 3;
struct LvAddLds {
    float k[kLvAddRows][kWave];
    uint16_t slot[kLvChunk];
    uint8_t hit[kLvChunk];
    uint32_t word_off[kLvChunk / kWave + 1];
    float ybar[kWave];
};

__global__ __launch_bounds__(kLvWaves *kWave) void bgklv_split_add_kernel(LvArgs a) {
    __shared__ LvAddLds L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t task = a.split_list[blockIdx.x];
    const LvTask t = lv_task(a, task, lane);
    if (t.pool && a.blk_mult[t.blk] <= a.pass) return;
    const uint32_t first = a.task_first[task], nsub = a.task_nsub[task];
    const size_t row0 = a.task_row0[task];
    constexpr uint32_t kWords = kLvChunk / kWave;
    constexpr uint32_t kRowsPerLoad = kLvWaves * kWave / 16;   // 32 rows per load instruction of the workgroup
    constexpr uint32_t kLoads = kLvAddRows / kRowsPerLoad;     // 4
    const uint32_t my_row = threadIdx.x >> 4, my_quad = threadIdx.x & 15u;
    float ybar = 0.0f, kbar = 0.0f;                     // kbar lives in wave 0, ybar in wave 1
    unsigned long long info = 0;
    float4 reg[kLvAddDepth][kLoads];
    for (uint32_t s = 0; s < nsub; ++s) {
        info |= a.sub_info[first + s];
        const unsigned long long *nzw = a.sub_nz + (size_t)(first + s) * kWords, *yw = a.sub_y + (size_t)(first + s) * kWords;
        // row list of the sub-task: slot numbers of the set bits, ascending
        if (wave == 0) {
            const uint32_t c = lane < (int)kWords ? (uint32_t)__popcll(nzw[lane]) : 0u;
            const uint32_t incl = wave_incl_scan_u32(c, lane);
            if (lane < (int)kWords) L.word_off[lane + 1] = incl;
            if (lane == 0) L.word_off[0] = 0;
        }
        __syncthreads();
        const uint32_t n = L.word_off[kWords];
        for (uint32_t b = threadIdx.x; b < kLvChunk; b += kLvWaves * kWave) {
            const unsigned long long wd = nzw[b >> 6];
            if ((wd >> (b & 63u)) & 1ull) {
                const uint32_t pos = L.word_off[b >> 6] + (uint32_t)__popcll(wd & ((1ull << (b & 63u)) - 1ull));
                L.slot[pos] = (uint16_t)b;
                L.hit[pos] = (uint8_t)((yw[b >> 6] >> (b & 63u)) & 1ull);
            }
        }
        __syncthreads();
        const float4 *rows4 = (const float4 *)(a.rows + (row0 + (size_t)s * kLvChunk) * kWave);
        auto fetch = [&](float4 (&r)[kLoads], uint32_t r0) {
            uint32_t sl[kLoads];
#pragma unroll
            for (uint32_t i = 0; i < kLoads; ++i) {
                const uint32_t row = r0 + my_row + i * kRowsPerLoad;
                sl[i] = row < n ? (uint32_t)L.slot[row] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (uint32_t i = 0; i < kLoads; ++i)
                r[i] = sl[i] != 0xFFFFFFFFu ? rows4[(size_t)sl[i] * 16 + my_quad] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        };
        auto tile = [&](float4 (&r)[kLoads], uint32_t r0) {   // r holds tile r0; afterwards the tile kLvAddDepth tiles on
            const uint32_t nr = min(n - r0, kLvAddRows);
#pragma unroll
            for (uint32_t i = 0; i < kLoads; ++i) *(float4 *)&L.k[my_row + i * kRowsPerLoad][4 * my_quad] = r[i];
            __syncthreads();
            if (r0 + kLvAddDepth * kLvAddRows < n) fetch(r, r0 + kLvAddDepth * kLvAddRows);
            if (wave == 0) {
                kbar = lv_add_dense(L.k, (nr + 15u) & ~15u, lane, kbar);   // fetch() zero-fills the tile past the last row
            } else if (wave == 1) {
                const unsigned long long m0 = __ballot((uint32_t)lane < nr && L.hit[r0 + lane]);
                const unsigned long long m1 = __ballot(64u + (uint32_t)lane < nr && L.hit[r0 + 64 + lane]);
                ybar = lv_add_rows(L.k, m0, lane, ybar);
                ybar = lv_add_rows(L.k + kWave, m1, lane, ybar);
            }
            __syncthreads();
        };
#pragma unroll
        for (int d = 0; d < kLvAddDepth; ++d)
            if ((uint32_t)d * kLvAddRows < n) fetch(reg[d], (uint32_t)d * kLvAddRows);
        for (uint32_t r0 = 0; r0 < n; r0 += kLvAddDepth * kLvAddRows) {
#pragma unroll
            for (int d = 0; d < kLvAddDepth; ++d)
                if (r0 + (uint32_t)d * kLvAddRows < n) tile(reg[d], r0 + (uint32_t)d * kLvAddRows);
        }
    }
    if (wave == 1) L.ybar[lane] = ybar;
    __syncthreads();
    if (wave != 0) return;
    lv_commit(a, t, L.ybar[lane], kbar, (info >> lane) & 1ull);
}

}  // namespace la3dm_dev
