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
};

// LV state code <-> the host enum stored in the pool (State::PRUNED = 3, State::UNCERTAIN = 4; la3dm_lv_scan uses the
// reference's numbering UNCERTAIN 3, PRUNED 4: src/bgklvoctomap/bgklvoctree_node.h:11-13)
__device__ __forceinline__ uint8_t lv_from_pool(uint8_t s) { s &= 7u; return s == 3u ? 4u : (s == 4u ? 3u : s); }
__device__ __forceinline__ uint8_t lv_to_pool(uint8_t s) { return s == 3u ? 4u : (s == 4u ? 3u : s); }

// point3f::norm(): double sqrt of a float sum, narrowed where the reference stores it in a float matrix
__device__ __forceinline__ float norm3f(float x, float y, float z) { return (float)sqrt((double)(x * x + y * y + z * z)); }

// include/bgklvoctomap/bgklvinference.h:104-131
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

struct __attribute__((aligned(16))) LvCand {
    float4 p;     // sample position, w: 0 = hit, 1 = first sample of its ray, 2 = later sample
    float4 prev;  // previous sample of the same ray (type 2)
    float4 r0;    // segment start (= the ray's first sample)
    float4 r1;    // segment end
};

// One workgroup of kLvWaves waves per cube.  Every wave stages its own 64 samples of a bucket (ordered compaction
// across the waves through LDS), then the staged candidates are evaluated kLvWaves at a time — wave w takes
// candidates w, w + kLvWaves, ... of a 64-candidate round, lane = voxel, and writes {k or +0, k * y or +0} into a
// dense [candidate][voxel] tile — and wave 0 adds the tile row by row: the cube next to the sensor, which sees every
// beam, spreads its distance / kernel evaluations over the CU's four SIMDs while the two running sums keep the
// gather order (adding +0 leaves a sum that started at +0 unchanged: it can never be -0).
constexpr int kLvWaves = 8;

struct LvLds {
    LvCand cand[kLvWaves * kWave];
    float k[kWave][kWave];
    float ky[kWave][kWave];
    uint32_t cnt[kLvWaves];
    uint32_t info[kLvWaves][kWave];
};

__global__ __launch_bounds__(kLvWaves *kWave) void bgklv_voxel_kernel(LvArgs a) {
    __shared__ LvLds L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t task = blockIdx.x;
    if (task >= a.n_tasks) return;
    const uint32_t blk = task >> a.cubes_shift;
    const uint32_t cube = task & ((1u << a.cubes_shift) - 1u);
    const uint32_t node = cube * kWave + lane;
    const bool in_range = node < a.nodes_per_blk;
    const bool pool = a.blk_slot != nullptr;
    if (pool && a.blk_mult[blk] <= a.pass) return;  // (uniform) this block's key repeats fewer times than the pass number
    const size_t ni = pool ? (size_t)a.blk_slot[blk] * a.npb + a.layer_off + (in_range ? node : 0)
                           : (size_t)blk * a.nodes_per_blk + (in_range ? node : 0);
    const uint8_t st_raw = in_range ? a.state[ni] : (uint8_t)4;
    const uint8_t st_in = !in_range ? (uint8_t)4 : (pool ? lv_from_pool(st_raw) : st_raw);
    const bool active = st_in != 4;
    const float4 off4 = a.lut[a.lut_base + (in_range ? node : 0)];
    const float cx = off4.x + a.blk_center[3 * blk + 0], cy = off4.y + a.blk_center[3 * blk + 1],
                cz = off4.z + a.blk_center[3 * blk + 2];
    // the voxel's closed query box (bgklvoctomap.cpp:162-166)
    const float lox = cx - a.ell, loy = cy - a.ell, loz = cz - a.ell;
    const float hix = cx + a.ell, hiy = cy + a.ell, hiz = cz + a.ell;
    const float inf = __builtin_inff();
    const float tlx = wave_min_dpp(active ? lox : inf), tly = wave_min_dpp(active ? loy : inf), tlz = wave_min_dpp(active ? loz : inf);
    const float thx = wave_max_dpp(active ? hix : -inf), thy = wave_max_dpp(active ? hiy : -inf), thz = wave_max_dpp(active ? hiz : -inf);
    __syncthreads();  // every wave has read the cube's states before wave 0 may rewrite them
    if (!(tlx <= thx)) {  // no base-resolution leaf in this cube (uniform over the workgroup)
        if (wave == 0 && in_range && !pool) a.state[ni] = 0;
        return;
    }
    // bucket of this cube: the octree index interleaves (x, y, z) bits, coarsest first
    int bxc = 0, byc = 0, bzc = 0;
    for (uint32_t lvl = 0; lvl < a.cubes_bits; ++lvl) {
        const uint32_t sh = 3u * (a.cubes_bits - 1u - lvl);
        const uint32_t oct = (cube >> sh) & 7u;
        bxc = (bxc << 1) | (int)((oct >> 2) & 1u);
        byc = (byc << 1) | (int)((oct >> 1) & 1u);
        bzc = (bzc << 1) | (int)(oct & 1u);
    }
    const int gx = a.blk_cell0[3 * blk + 0] + bxc - a.cell_min[0], gy = a.blk_cell0[3 * blk + 1] + byc - a.cell_min[1],
              gz = a.blk_cell0[3 * blk + 2] + bzc - a.cell_min[2];

    float ybar = 0.0f, kbar = 0.0f;
    bool info = false;
    for (int dz = -a.reach; dz <= a.reach; ++dz)
        for (int dy = -a.reach; dy <= a.reach; ++dy)
            for (int dx = -a.reach; dx <= a.reach; ++dx) {
                const int x = gx + dx, y = gy + dy, z = gz + dz;
                if (x < 0 || y < 0 || z < 0 || x >= a.cell_dim[0] || y >= a.cell_dim[1] || z >= a.cell_dim[2]) continue;
                const uint32_t cell = ((uint32_t)z * a.cell_dim[1] + (uint32_t)y) * a.cell_dim[0] + (uint32_t)x;
                const uint32_t c0 = a.cell_off[cell], c1 = a.cell_off[cell + 1];
                for (uint32_t base = c0; base < c1; base += kLvWaves * kWave) {
                    // stage: lane = sample, wave w takes samples [base + 64 w, base + 64 w + 64)
                    const uint32_t si = base + (uint32_t)wave * kWave + lane;
                    bool keep = false;
                    LvCand c;
                    if (si < c1) {
                        const float4 s = a.sorted[si];
                        keep = !(tlx > s.x || s.x > thx || tly > s.y || s.y > thy || tlz > s.z || s.z > thz);
                        if (keep) {
                            const uint32_t orig = __float_as_uint(s.w);
                            const float4 so = a.samples[orig];
                            const int ray = (int)so.w;
                            c.p = make_float4(s.x, s.y, s.z, 0.0f);
                            c.prev = c.r0 = c.r1 = c.p;
                            if (ray >= 0) {
                                const float4 ra = a.rays[2 * ray], rb = a.rays[2 * ray + 1];
                                const uint32_t first = __float_as_uint(ra.w);
                                c.r0 = ra;
                                c.r1 = rb;
                                c.p.w = orig == first ? 1.0f : 2.0f;
                                if (orig != first) c.prev = a.samples[orig - 1];
                            }
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
                    if (keep) L.cand[before + slot] = c;
                    __syncthreads();
                    // rounds of 64 staged candidates: evaluate (all waves), then add in order (wave 0)
                    for (uint32_t r0 = 0; r0 < total; r0 += kWave) {
                        const uint32_t nr = min(total - r0, (uint32_t)kWave);
                        for (uint32_t j = wave; j < nr; j += kLvWaves) {
                            const LvCand &cd = L.cand[r0 + j];
                            const float4 p = cd.p;
                            const bool inb = active && !(lox > p.x || p.x > hix || loy > p.y || p.y > hiy || loz > p.z || p.z > hiz);
                            float kv = 0.0f, kyv = 0.0f;
                            if (__any(inb)) {
                                bool count = inb;
                                float ax = p.x, ay = p.y, az = p.z, bx = p.x, by = p.y, bz = p.z, yv = 1.0f;
                                if (p.w != 0.0f) {  // a ray sample (uniform): is it this ray's lowest-index sample in my box?
                                    const float4 q0 = cd.r0, q1 = cd.r1, pv = cd.prev;
                                    ax = q0.x; ay = q0.y; az = q0.z; bx = q1.x; by = q1.y; bz = q1.z;
                                    yv = 0.0f;
                                    if (p.w == 2.0f) {
                                        const bool first_in = !(lox > q0.x || q0.x > hix || loy > q0.y || q0.y > hiy || loz > q0.z || q0.z > hiz);
                                        const bool prev_in = !(lox > pv.x || pv.x > hix || loy > pv.y || pv.y > hiy || loz > pv.z || pv.z > hiz);
                                        count = inb && !first_in && !prev_in;
                                    }
                                }
                                info |= inb;
                                if (count) {
                                    const float d = seg_dist_dev(cx, cy, cz, ax, ay, az, bx, by, bz);
                                    kv = cov_sparse_line_dev(d, a.ell, a.sf2);
                                    kyv = kv * yv;
                                }
                            }
                            L.k[j][lane] = kv;
                            L.ky[j][lane] = kyv;
                        }
                        __syncthreads();
                        if (wave == 0) {
                            for (uint32_t j = 0; j < nr; ++j) {
                                ybar += L.ky[j][lane];
                                kbar += L.k[j][lane];
                            }
                        }
                        __syncthreads();
                    }
                }
            }
    L.info[wave][lane] = info ? 1u : 0u;
    __syncthreads();
    if (wave != 0 || !in_range) return;
#pragma unroll
    for (int v = 1; v < kLvWaves; ++v) info |= L.info[v][lane] != 0u;
    uint8_t out = info ? 0x40u : 0u;
    bool updated = false;
    if (active && info && kbar > 0.001f) {  // bgklvoctomap.cpp:236-238
        updated = true;
        float A = a.alpha[ni], B = a.beta[ni];
        A += ybar;
        B += kbar - ybar;
        const float prob = lv_prob_dev(A, B, a.min_W);
        const float var = lv_var_dev(A, B, a.min_W, prob);
        uint8_t st;
        if (var > a.var_thresh) st = 3;
        else st = prob > a.occupied_thresh ? 1 : (prob < a.free_thresh ? 0 : 2);
        a.alpha[ni] = A;
        a.beta[ni] = B;
        out |= (uint8_t)(0x80u | (pool ? lv_to_pool(st) : st));
    }
    if (!pool) {
        a.state[ni] = out;
        return;
    }
    // pool: an untouched node keeps its byte (state + classified), plus the transient "saw samples" bit
    a.state[ni] = updated ? out : (uint8_t)(st_raw | (out & 0x40u));
    const unsigned long long um = __ballot(updated);
    if (lane == 0 && um) atomicAdd(a.upd_counter, (uint32_t)__popcll(um));
}

}  // namespace la3dm_dev
