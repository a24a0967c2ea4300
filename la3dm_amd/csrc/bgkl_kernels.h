// bgkl_kernels.h — BGKLOctoMap (block-level BGK with free-space line segments), SURVEY.md §8 row f4.
//
// Reference (CPU):
//   BGKLInference::predict        include/bgkloctomap/bgklinference.h:80-88
//   point_to_line_dist            include/bgkloctomap/bgklinference.h:104-140   (== seg_dist_dev, lv_kernels.h; evaluated by
//                                 its fp32 twin seg_dist_f32, same bits: see there)
//   covSparseLine                 include/bgkloctomap/bgklinference.h:186-200   (d / ell, formula, `< 0 -> 0`)
//   7-neighbour update loop       src/bgkloctomap/bgkloctomap.cpp:206-231       (gate: kbar > 0.001)
//   Occupancy::update             src/bgkloctomap/bgkloctree_node.cpp:31-44     (same node as BGKOctoMap)
//
// Training rows are 8 floats {x0, y0, z0, x1, y1, z1, label, 0}: hits are degenerate segments with label 1,
// each beam that has a sample inside the block contributes its segment once with label 0.  bgkl_rows_prepare adds what
// the line distance needs from the segment alone (direction, squared length, the "shorter than 0.1 mm" decision).  One
// wave (or eight, on small scans) = one leaf tile, lane = leaf: the rows of the seven neighbours in order, squared
// point-to-segment distance on every lane, hit <=> d^2 below the exact threshold of `d < ell`, k(sqrt(d^2) / ell) for the
// hits, the two running sums in row order — bit-identical to the CPU restatement.  Tiles with very many rows (the
// blocks around the sensor on large scans) take the split path further down.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace la3dm_dev {

struct BgklArgs {
    const float *rows;           // 8 floats per row, grouped by training block (the ABI's layout: input of bgkl_rows_prepare)
    const float4 *rowx;          // 3 float4 per row {x0 y0 z0 x1 | y1 z1 label degenerate | lx ly lz |l|^2} (bgkl_rows_prepare)
    const uint32_t *row_off;     // CSR over training blocks
    const int32_t *nbr;          // [n_test_blk * 7] training block or -1, ExtendedBlock order
    const float *blk_center;
    const uint32_t *leaf_off;
    const uint32_t *leaf_key;
    float *alpha;
    float *beta;
    uint8_t *state;
    const float4 *lut;
    uint32_t n_test_blk;
    uint32_t tpb_shift;
    uint32_t n_tasks;
    float sf2, ell, free_thresh, occupied_thresh, var_thresh;
    float inv_ell;   // RN(1 / ell) or 0 (bgk_kernels.h div_by_ell)
    float hit_d2;    // the smallest fp32 t with sqrtf(t) >= ell: d >= ell <=> d^2 >= hit_d2 for d = sqrtf(d^2) (host, exact)
    int trig;        // "fast_trig": 0 correctly rounded (default), 3 = Eigen 3.3.7 psin / pcos (bgk_kernels.h sincos_eigen337)
};

// Split path for the tiles around the sensor.  Every beam crosses the sensor's block, so a 200 k-ray scan hands a
// handful of tiles ~10^5 rows each and the row-serial kernel below becomes one long straggler wave.  Only the two
// fp32 running sums have to be serial; the distance test and the kernel evaluation do not.  A tile whose seven
// neighbours hold more than `threshold` rows is cut into items of kLItemRows rows:
//   bgkl_split_mark    per tile: row total, item count, first item, index in the list of split tiles (atomic
//                      bumps; the order of items and tiles is free, every tile is independent)
//   bgkl_split_items   item descriptors {tile, neighbour, row range}, first item of each (tile, neighbour)
//   bgkl_split_eval    one wave per item: distance test, per row {hit mask, label, running hit count}; the hit lanes'
//                      squared distances in (row, lane) order in the item's own value slots
//   bgkl_split_bdesc   per 64-row batch: where its values live
//   bgkl_split_kernelize  squared distances -> k(d / ell), in place, dense
//   bgkl_split_expand  (default form of the replay) compact values -> dense {k of 4 rows per leaf} tiles, all items at once
//   bgkl_split_add     one workgroup per (split tile, neighbour) — the seven (ybar, kbar) pairs of a tile are
//                      independent chains.  Fourteen copy waves bring the next 64-row batches of the dense tiles into
//                      LDS (four batches in flight) while one consumer wave adds the current batch's k row by row and a
//                      second one the rows with label 1 — the same sums as the row-serial kernel, bit for bit
//   bgkl_split_fuse    (option bgkl_dense_add = 0) the same replay with producers that expand the compact values
//                      themselves: 64 KB less scratch per item, the sensor block's chain 1.7x slower
//   bgkl_split_apply   per split tile: the gated update of (alpha, beta) in ExtendedBlock order + state
constexpr int kLItemRows = 256;
constexpr int kLBatch = 64;
constexpr int kLBatches = kLItemRows / kLBatch;
constexpr uint32_t kLItemVals = kLItemRows * kWave;   // value slots of an item (every row x every leaf)
constexpr int kLProducers = 14;   // + two consumer waves = a 1024-thread workgroup
constexpr int kLRowsPerProducer = (kLBatch + kLProducers - 1) / kLProducers;  // 5

struct BgklSplit {
    uint32_t *task_item;          // [2 * n_tasks] {first item or 0xFFFFFFFF, index in split_list}
    uint32_t *counters;           // [0] items, [1] split tiles
    uint32_t *split_list;         // [split tiles] tile
    uint32_t *nb_first;           // [split tiles * 8] first item of neighbour slot b; [7] = end
    uint4 *item_desc;             // {tile, neighbour slot, row begin, row end}
    uint4 *rowrec;                // [items * kLItemRows] {mask lo, mask hi, label, hits before the row inside its batch}
    uint32_t *batch_off;          // [items * kLBatches] hits before the batch inside its item
    uint32_t *item_hits;          // [items]
    uint4 *bdesc;                 // [items * kLBatches] {value index lo, hi, values, rows | slot << 16}
    float *vals;
    float2 *part;                 // [split tiles * 7 * 64] (ybar, kbar) of one neighbour
    float4 *dense;                // [items * kLItemRows / 4 * 64] k of rows 4g .. 4g + 3 for lane = leaf, +0 where out of reach (bgkl_split_expand)
    unsigned long long *labmask;  // [items * kLBatches] bit r: row r of the batch has label 1
    double2 *part64;              // [items * 64] {sum k * label, sum k} of one item in double (order-free mode, bgkl_split_sum)
    uint32_t threshold;
};

// Does a row at distance d enter a leaf's sums?  d < ell: the kernel is positive.  d >= ell: r >= 1, the kernel is <= 0
// and cleaned to 0 — skipped.  d NaN or inf (a beam built from a NaN point of an unfiltered cloud): the reference's
// dense formula yields NaN, the `< 0 -> 0` clean-up lets it through and it poisons ybar / kbar of every leaf that
// meets the row (the kbar > 0.001 gate then rejects the update) — kept, with k = NaN.
__device__ __forceinline__ bool bgkl_row_counts(float d2, float hit_d2) { return !(d2 >= hit_d2) || d2 == __builtin_inff(); }
__device__ __forceinline__ float bgkl_row_kernel(float d, float ell, float inv_ell, float sf2, int trig = 0) {
    if (!(d - d == 0.0f)) return __builtin_nanf("");
    const float r = div_by_ell(d, ell, inv_ell);
    return trig == 3 ? cov_sparse_fast<3, true>(r, sf2) : cov_sparse_fast<0, true>(r, sf2);   // (wave-uniform)
}

// What point_to_line_dist computes from the segment alone — its direction l = b - a, |l|^2 and the "shorter than 0.1 mm"
// decision — once per row instead of once per (row, leaf): the same fp32 expressions as seg_dist_f32 (lv_kernels.h), so the
// same bits.  A third of that function's instructions were these wave-uniform terms (there is no scalar float unit to
// hoist them to).
__global__ __launch_bounds__(256) void bgkl_rows_prepare(const float *__restrict__ rows, uint32_t n_rows, float4 *__restrict__ rowx) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rows) return;
    const float4 p0 = *reinterpret_cast<const float4 *>(rows + 8 * (size_t)j);       // x0 y0 z0 x1
    const float4 p1 = *reinterpret_cast<const float4 *>(rows + 8 * (size_t)j + 4);   // y1 z1 label -
    const float lx = p0.w - p0.x, ly = p1.x - p0.y, lz = p1.y - p0.z;
    const float c2 = lx * lx + ly * ly + lz * lz;
    const bool degenerate = sqrtf(c2) < 0.0001f;
    rowx[3 * (size_t)j] = p0;
    rowx[3 * (size_t)j + 1] = make_float4(p1.x, p1.y, p1.z, degenerate ? 1.0f : 0.0f);
    rowx[3 * (size_t)j + 2] = make_float4(lx, ly, lz, c2);
}

// Squared distance from p to the row's segment: seg_dist_f32 without its final square root (the callers compare d^2 with
// hit_d2 and take the root of the hits only) and with the row's own terms read instead of recomputed.  q1.w != 0 (wave-
// uniform): the distance is measured to the segment's start.
__device__ __forceinline__ float bgkl_seg_d2(float px, float py, float pz, const float4 q0, const float4 q1, const float4 q2) {
    float qx = q0.x, qy = q0.y, qz = q0.z;
    if (q1.w == 0.0f) {   // lv_seg_point (lv_kernels.h), c2 = q2.w
        const float vx = px - q0.x, vy = py - q0.y, vz = pz - q0.z;
        const float c1 = vx * q2.x + vy * q2.y + vz * q2.z;
        if (!(c1 <= 0.0f)) {
            if (q2.w <= c1) {
                qx = q0.w; qy = q1.x; qz = q1.y;
            } else {
                float b = c1 / q2.w;
                if (b < 0x1p-100f) b = (float)((double)c1 / (double)q2.w);
                qx = q0.x + q2.x * b; qy = q0.y + q2.y * b; qz = q0.z + q2.z * b;
            }
        }
    }
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    return dx * dx + dy * dy + dz * dz;
}

// leaf of this lane: false when the tile holds no leaves
__device__ __forceinline__ bool bgkl_leaf(const BgklArgs &a, uint32_t task, int lane, uint32_t &blk, uint32_t &li, bool &active,
                                          float &px, float &py, float &pz) {
    blk = task >> a.tpb_shift;
    const uint32_t tile = task & ((1u << a.tpb_shift) - 1u);
    const uint32_t l0 = a.leaf_off[blk] + tile * kWave;
    const uint32_t l1 = a.leaf_off[blk + 1];
    if (l0 >= l1) return false;
    const uint32_t nl = min(l1 - l0, (uint32_t)kWave);
    active = (uint32_t)lane < nl;
    li = l0 + (active ? lane : 0);
    const uint32_t key = a.leaf_key[li];
    const float4 off4 = a.lut[lut_layer_base(key >> 16) + (key & 0xFFFFu)];
    px = off4.x + a.blk_center[3 * blk + 0];  // Block::get_loc
    py = off4.y + a.blk_center[3 * blk + 1];
    pz = off4.z + a.blk_center[3 * blk + 2];
    return true;
}

__device__ __forceinline__ void bgkl_store(const BgklArgs &a, uint32_t li, bool active, bool updated, float A, float B) {
    if (!active) return;
    if (updated) {
        a.alpha[li] = A;
        a.beta[li] = B;
        BgkArgs c;  // thresholds for classify()
        c.free_thresh = a.free_thresh;
        c.occupied_thresh = a.occupied_thresh;
        c.var_thresh = a.var_thresh;
        a.state[li] = (uint8_t)(classify(A, B, c) | 0x80u);
    } else {
        a.state[li] = 0;
    }
}

// One workgroup of kW waves per tile: the rows of a neighbour are taken 64 at a time, wave w evaluates rows
// w, w + kW, ... (lane = leaf) into a dense [row][leaf] LDS tile ({k or +0}, {k * label or +0}), wave 0 adds the
// tile row by row — the distance tests and kernel evaluations of a tile spread over the CU's four SIMDs, the two
// running sums keep the row order (a sum that starts at +0 can never be -0, so adding +0 changes nothing).
// kW = 8 when the scan has few tiles (each is a serial chain and the GPU is mostly empty: latency counts), kW = 1
// (no LDS, rows straight into the sums) when there are many (the launch is throughput-bound).
constexpr int kLWaves = 8;
constexpr uint32_t kLWideTiles = 4096;

template <int kW>
__global__ __launch_bounds__(kW *kWave) void bgkl_predict_fuse_kernel(BgklArgs a, const uint32_t *__restrict__ task_item) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t task = blockIdx.x;
    if (task >= a.n_tasks) return;
    if (task_item && task_item[2 * task] != 0xFFFFFFFFu) return;  // split tile
    uint32_t blk, li;
    bool active;
    float px, py, pz;
    if (!bgkl_leaf(a, task, lane, blk, li, active, px, py, pz)) return;
    float A = a.alpha[li], B = a.beta[li];
    bool updated = false;

    for (int b = 0; b < 7; ++b) {
        const int32_t tb = a.nbr[7 * blk + b];
        if (tb < 0) continue;
        const uint32_t r0 = __builtin_amdgcn_readfirstlane(a.row_off[tb]), r1 = __builtin_amdgcn_readfirstlane(a.row_off[tb + 1]);
        float ybar = 0.0f, kbar = 0.0f;
        if constexpr (kW == 1) {  // one wave: rows in order, straight into the sums
            for (uint32_t j = r0; j < r1; ++j) {
                const float4 q0 = a.rowx[3 * (size_t)j], q1 = a.rowx[3 * (size_t)j + 1], q2 = a.rowx[3 * (size_t)j + 2];
                const float d2 = bgkl_seg_d2(px, py, pz, q0, q1, q2);
                const bool hit = active && bgkl_row_counts(d2, a.hit_d2);
                if (__ballot(hit) == 0ull) continue;
                if (hit) {
                    const float kv = bgkl_row_kernel(sqrtf(d2), a.ell, a.inv_ell, a.sf2, a.trig);
                    ybar += kv * q1.z;
                    kbar += kv;
                }
            }
        } else {
            __shared__ float s_k[kWave][kWave];
            __shared__ float s_ky[kWave][kWave];
            for (uint32_t q = r0; q < r1; q += kWave) {
                const uint32_t nr = min(r1 - q, (uint32_t)kWave);
                for (uint32_t j = wave; j < nr; j += kW) {
                    const size_t row = (size_t)q + j;
                    const float4 q0 = a.rowx[3 * row], q1 = a.rowx[3 * row + 1], q2 = a.rowx[3 * row + 2];
                    const float d2 = bgkl_seg_d2(px, py, pz, q0, q1, q2);
                    float kv = 0.0f, kyv = 0.0f;
                    if (active && bgkl_row_counts(d2, a.hit_d2)) {
                        kv = bgkl_row_kernel(sqrtf(d2), a.ell, a.inv_ell, a.sf2, a.trig);
                        kyv = kv * q1.z;
                    }
                    s_k[j][lane] = kv;
                    s_ky[j][lane] = kyv;
                }
                __syncthreads();
                if (wave == 0) {
                    for (uint32_t j = 0; j < nr; ++j) {
                        ybar += s_ky[j][lane];
                        kbar += s_k[j][lane];
                    }
                }
                __syncthreads();
            }
        }
        if (kbar > 0.001f) {  // bgkloctomap.cpp:226-227 (only wave 0 holds the sums)
            A += ybar;
            B += kbar - ybar;
            updated = true;
        }
    }
    if (wave == 0) bgkl_store(a, li, active, updated, A, B);
}

__global__ void bgkl_split_mark(BgklArgs a, BgklSplit s) {
    const uint32_t task = blockIdx.x * blockDim.x + threadIdx.x;
    if (task >= a.n_tasks) return;
    const uint32_t blk = task >> a.tpb_shift, tile = task & ((1u << a.tpb_shift) - 1u);
    uint32_t first = 0xFFFFFFFFu, idx = 0;
    if (a.leaf_off[blk] + tile * kWave < a.leaf_off[blk + 1]) {
        unsigned long long total = 0;
        uint32_t items = 0;
        for (int b = 0; b < 7; ++b) {
            const int32_t tb = a.nbr[7 * blk + b];
            if (tb < 0) continue;
            const uint32_t c = a.row_off[tb + 1] - a.row_off[tb];
            total += c;
            items += (c + kLItemRows - 1) / kLItemRows;
        }
        if (total > (unsigned long long)s.threshold) {
            first = atomicAdd(&s.counters[0], items);
            idx = atomicAdd(&s.counters[1], 1u);
            s.split_list[idx] = task;
        }
    }
    s.task_item[2 * task] = first;
    s.task_item[2 * task + 1] = idx;
}

__global__ void bgkl_split_items(BgklArgs a, BgklSplit s) {
    const uint32_t task = blockIdx.x * blockDim.x + threadIdx.x;
    if (task >= a.n_tasks) return;
    uint32_t it = s.task_item[2 * task];
    if (it == 0xFFFFFFFFu) return;
    const uint32_t h = s.task_item[2 * task + 1];
    const uint32_t blk = task >> a.tpb_shift;
    for (int b = 0; b < 7; ++b) {
        s.nb_first[8 * h + b] = it;
        const int32_t tb = a.nbr[7 * blk + b];
        if (tb < 0) continue;
        const uint32_t r0 = a.row_off[tb], r1 = a.row_off[tb + 1];
        for (uint32_t r = r0; r < r1; r += kLItemRows) s.item_desc[it++] = make_uint4(task, (uint32_t)b, r, min(r + (uint32_t)kLItemRows, r1));
    }
    s.nb_first[8 * h + 7] = it;
}

// the item descriptors of the split tiles, one workgroup of seven waves per split tile (wave = neighbour slot, lanes stride
// over the neighbour's items): bgkl_split_items above walks every tile with one thread — the sensor block's thread writes
// thousands of descriptors one after the other (82 us of a 200 k-ray insert)
__global__ __launch_bounds__(7 * kWave) void bgkl_split_items_wide(BgklArgs a, BgklSplit s) {
    const uint32_t h = blockIdx.x, b = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t task = s.split_list[h];
    const uint32_t blk = task >> a.tpb_shift;
    uint32_t it = s.task_item[2 * task];
    uint32_t r0 = 0, r1 = 0;
    for (uint32_t q = 0; q <= b; ++q) {   // items of the slots before mine
        const int32_t tb = a.nbr[7 * blk + q];
        const uint32_t c = tb < 0 ? 0u : a.row_off[tb + 1] - a.row_off[tb];
        if (q == b) {
            if (tb >= 0) {
                r0 = a.row_off[tb];
                r1 = a.row_off[tb + 1];
            }
        } else {
            it += (c + kLItemRows - 1) / kLItemRows;
        }
    }
    const uint32_t n = (r1 - r0 + kLItemRows - 1) / kLItemRows;
    if (lane == 0) {
        s.nb_first[8 * h + b] = it;
        if (b == 6u) s.nb_first[8 * h + 7] = it + n;
    }
    for (uint32_t i = lane; i < n; i += kWave) {
        const uint32_t r = r0 + i * kLItemRows;
        s.item_desc[it + i] = make_uint4(task, b, r, min(r + (uint32_t)kLItemRows, r1));
    }
}

// One wave per item (tile x neighbour x <= kLItemRows rows), lane = leaf: the distance test of every row, the hit masks
// and labels as row records, and the hit lanes' squared distances in (row, lane) order in the item's own value slots.
__global__ __launch_bounds__(kWave) void bgkl_split_eval(BgklArgs a, BgklSplit s) {
    const int lane = threadIdx.x;
    const uint32_t it = blockIdx.x;
    const uint4 dsc = s.item_desc[it];
    uint32_t blk, li;
    bool active;
    float px, py, pz;
    if (!bgkl_leaf(a, dsc.x, lane, blk, li, active, px, py, pz)) return;  // (split tiles always hold leaves)
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(dsc.z), nrows = __builtin_amdgcn_readfirstlane(dsc.w) - r0;
    uint4 *rec = s.rowrec + (size_t)it * kLItemRows;
    float *vals = s.vals + (size_t)it * kLItemVals;
    uint32_t off = 0, boff = 0;
    uint4 mine = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t j = 0; j < nrows; ++j) {
        const size_t row = (size_t)r0 + j;
        const float4 q0 = a.rowx[3 * row], q1 = a.rowx[3 * row + 1], q2 = a.rowx[3 * row + 2];
        const float4 p1 = q1;
        const float d2 = bgkl_seg_d2(px, py, pz, q0, q1, q2);
        const bool hit = active && bgkl_row_counts(d2, a.hit_d2);
        const unsigned long long m = __ballot(hit);
        // bgkl_split_kernelize turns the squared distances into kernel values in place, at full lane utilisation (a second
        // distance pass over all rows that wrote the values was 1.0 ms of a 200 k-ray insert)
        if (hit) vals[off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = d2;
        if ((j & 63u) == 0u) {
            boff = off;
            if (lane == 0) s.batch_off[it * kLBatches + (j >> 6)] = off;
        }
        if ((uint32_t)lane == (j & 63u)) mine = make_uint4((uint32_t)m, (uint32_t)(m >> 32), __float_as_uint(p1.z), off - boff);
        off += (uint32_t)__popcll(m);
        if ((j & 63u) == 63u || j + 1 == nrows) {
            if ((uint32_t)lane <= (j & 63u)) rec[(j & ~63u) + lane] = mine;
        }
    }
    if (lane == 0) s.item_hits[it] = off;
}

// squared distances -> kernel values, in place, dense
__global__ __launch_bounds__(256) void bgkl_split_kernelize(BgklArgs a, BgklSplit s) {
    const uint32_t it = blockIdx.x;
    const uint32_t n = s.item_hits[it];
    float *v = s.vals + (size_t)it * kLItemVals;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) v[i] = bgkl_row_kernel(sqrtf(v[i]), a.ell, a.inv_ell, a.sf2, a.trig);
}

__global__ void bgkl_split_bdesc(BgklSplit s, uint32_t n_items) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_items * kLBatches) return;
    const uint32_t it = q / kLBatches, k = q % kLBatches;
    const uint4 dsc = s.item_desc[it];
    const uint32_t nrows = dsc.w - dsc.z;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (k * kLBatch < nrows) {
        const uint32_t nr = min(nrows - k * kLBatch, (uint32_t)kLBatch);
        const uint32_t o0 = s.batch_off[q];
        const uint32_t o1 = ((k + 1) * kLBatch < nrows) ? s.batch_off[q + 1] : s.item_hits[it];
        const unsigned long long v = (unsigned long long)it * kLItemVals + o0;   // the item's own value slots
        d = make_uint4((uint32_t)v, (uint32_t)(v >> 32), o1 - o0, nr | (dsc.y << 16));
    }
    s.bdesc[q] = d;
}

__global__ __launch_bounds__(kWave *(2 + kLProducers)) void bgkl_split_fuse(BgklArgs a, BgklSplit s) {
    __shared__ float s_k[2][kLBatch][kWave];   // k of (row, leaf), +0 where the leaf is out of reach
    __shared__ float s_lab[2][kLBatch];        // label of the row (1 = hit, 0 = beam): k * label is k or a zero
    __shared__ float s_y[kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t h = blockIdx.x / 7u, b = blockIdx.x % 7u;
    const uint32_t it0 = s.nb_first[8 * h + b], it1 = s.nb_first[8 * h + b + 1];
    if (it0 == it1) return;  // no trained model in this slot (uniform over the workgroup)
    const uint32_t q0 = it0 * kLBatches;
    const int nb = (int)((it1 - it0) * kLBatches);  // batches of this chain (the last item may end with empty ones)
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    const int pw = wave - 2;   // wave 0 adds the k rows, wave 1 the k * label rows (hit rows only), waves 2.. produce

    // producer state: row records of the batch two trips ahead (lane k holds row pw + 7 k; asked for on one trip,
    // used on the next), values of the batch one trip ahead (asked for on one trip, written to LDS on the next)
    uint4 rec = zero4;
    float val[kLRowsPerProducer], lab[kLRowsPerProducer];
#pragma unroll
    for (int k = 0; k < kLRowsPerProducer; ++k) val[k] = lab[k] = 0.0f;
    float ybar = 0.0f, kbar = 0.0f;

    for (int i = -3; i < nb; ++i) {
        if (wave == 0) {
            // ---- consumer of k: batch i, in row order (constant LDS offsets, 16 reads in flight) ----
            if (i >= 0) {
                const int buf = i & 1;
#pragma unroll 16
                for (int r = 0; r < kLBatch; ++r) kbar += s_k[buf][r][lane];
            }
        } else if (wave == 1) {
            // ---- consumer of k * label: only the hit rows add something (x + 0 = x for a sum that started at +0) ----
            if (i >= 0) {
                const int buf = i & 1;
                unsigned long long m = __ballot(s_lab[buf][lane] != 0.0f);
                while (m) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = m ? __builtin_ctzll(m) : 0;
                        v[u] = m ? s_k[buf][r][lane] : 0.0f;
                        m &= m - 1ull;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) ybar += v[u];
                }
            }
        } else {
            // ---- producers ----
            // (w) batch i + 1: the values asked for on the previous trip -> LDS (+0 where the leaf is out of reach
            //     and in the rows past the end of the batch)
            if (i + 1 >= 0 && i + 1 < nb) {
                const int buf = (i + 1) & 1;
#pragma unroll
                for (int k = 0; k < kLRowsPerProducer; ++k) {
                    const int r = pw + kLProducers * k;
                    if (r < kLBatch) {
                        s_k[buf][r][lane] = val[k];
                        if (lane == 0) s_lab[buf][r] = lab[k];
                    }
                }
            }
            // (v) batch i + 2: its row records arrived -> ask for the hit lanes' values
            uint4 rec_next = zero4;
            if (i + 2 >= 0 && i + 2 < nb) {
                const uint4 D = s.bdesc[q0 + (uint32_t)(i + 2)];
                const uint32_t nr = __builtin_amdgcn_readfirstlane(D.w & 0xFFFFu);
                const unsigned long long v0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(D.y) << 32) |
                                              (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(D.x);   // (readfirstlane returns int)
#pragma unroll
                for (int k = 0; k < kLRowsPerProducer; ++k) {
                    const uint32_t r = (uint32_t)(pw + kLProducers * k);
                    float v = 0.0f, l = 0.0f;
                    if (r < nr) {  // uniform
                        const uint32_t mlo = __builtin_amdgcn_readlane(rec.x, k), mhi = __builtin_amdgcn_readlane(rec.y, k);
                        const uint32_t woff = __builtin_amdgcn_readlane(rec.w, k);
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0));
                        l = __uint_as_float(__builtin_amdgcn_readlane(rec.z, k));   // the row's label (uniform)
                        if ((((lane < 32 ? mlo : mhi) >> (lane & 31)) & 1u) != 0u) v = s.vals[v0 + woff + rank];
                    }
                    val[k] = v;
                    lab[k] = l;
                }
            } else {
#pragma unroll
                for (int k = 0; k < kLRowsPerProducer; ++k) val[k] = lab[k] = 0.0f;
            }
            // (r) batch i + 3: ask for its row records, one per lane
            if (i + 3 < nb) {
                const uint4 D = s.bdesc[q0 + (uint32_t)(i + 3)];
                const uint32_t nr = D.w & 0xFFFFu;
                const uint32_t r = (uint32_t)(pw + kLProducers * lane);
                if (lane < kLRowsPerProducer && r < nr) rec_next = s.rowrec[(size_t)(q0 + (uint32_t)(i + 3)) * kLBatch + r];
            }
            rec = rec_next;
        }
        __syncthreads();
    }
    if (wave == 1) s_y[lane] = ybar;
    __syncthreads();
    if (wave == 0) s.part[((size_t)h * 7u + b) * kWave + lane] = make_float2(s_y[lane], kbar);
}

constexpr int kLBatchVec = kLBatch / 4 * kWave;    // float4s of a batch: 16 groups x 64 lanes

// The replay with the expansion taken out of it.  bgkl_split_fuse above expands the compact values of a (tile,
// neighbour) chain into dense LDS tiles inside the one workgroup that also adds them, so the sensor block's chain (10^5
// rows) is bounded by that workgroup's producers (~900 cycles per 64 rows; 1.9 ms of a 200 k-ray insert).  The expansion
// has no order to keep: bgkl_split_expand does it for all items at once, into global memory — per 64-row batch 16 groups
// of 4 rows, a float4 per lane {k of rows 4g .. 4g + 3 for this leaf, +0 where the leaf is out of reach or the row does
// not exist} — and bgkl_split_add keeps the shape of the replay (producers -> LDS -> two consumer waves, one barrier per
// batch) with producers that only copy: 16 KB per batch, fetched kLAddDepth batches ahead (one wave alone cannot keep
// enough loads in flight to stream a chain at memory latency).
// grid kLBatches * items, 256 threads = one 64-row batch: the row records first (one per lane), then every wave asks for
// the values of its 16 rows at once, the tile is assembled in LDS and leaves as 16 KB of coalesced float4 stores
__global__ __launch_bounds__(256) void bgkl_split_expand(BgklSplit s) {
    __shared__ uint4 s_rec[kLBatch];
    __shared__ float s_t[kLBatch / 4][kWave][4];
    const int lane = threadIdx.x & 63, u = threadIdx.x >> 6;
    const uint32_t it = blockIdx.x / (uint32_t)kLBatches, bt = blockIdx.x % (uint32_t)kLBatches;   // (one grid dimension: items can exceed 65535)
    const uint4 dsc = s.item_desc[it];
    const uint32_t nrows = dsc.w - dsc.z;
    if (bt * (uint32_t)kLBatch >= nrows) return;   // (uniform) a batch without rows is never read
    if (u == 0) {
        const uint32_t j = bt * (uint32_t)kLBatch + (uint32_t)lane;
        const uint4 R = j < nrows ? s.rowrec[(size_t)it * kLItemRows + j] : make_uint4(0u, 0u, 0u, 0u);
        s_rec[lane] = R;
        const unsigned long long m = __ballot(j < nrows && __uint_as_float(R.z) != 0.0f);   // rows with label 1
        if (lane == 0) s.labmask[(size_t)it * kLBatches + bt] = m;
    }
    __syncthreads();
    const float *v = s.vals + (size_t)it * kLItemVals + s.batch_off[it * kLBatches + bt];
    float val[kLBatch / 4];
#pragma unroll
    for (int k = 0; k < kLBatch / 4; ++k) {   // row 4k + u of the batch (a row past the end has an empty mask)
        const uint4 R = s_rec[4 * k + u];
        const uint32_t mlo = __builtin_amdgcn_readfirstlane(R.x), mhi = __builtin_amdgcn_readfirstlane(R.y);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0));
        val[k] = 0.0f;
        if ((((lane < 32 ? mlo : mhi) >> (lane & 31)) & 1u) != 0u) val[k] = v[(uint32_t)__builtin_amdgcn_readfirstlane(R.w) + rank];
    }
#pragma unroll
    for (int k = 0; k < kLBatch / 4; ++k) s_t[k][lane][u] = val[k];
    __syncthreads();
    float4 *out = s.dense + ((size_t)it * kLBatches + bt) * kLBatchVec;
    const float4 *t = reinterpret_cast<const float4 *>(&s_t[0][0][0]);
#pragma unroll
    for (int q = 0; q < kLBatchVec / 256; ++q) out[threadIdx.x + 256 * q] = t[threadIdx.x + 256 * q];
}

constexpr int kLAddDepth = 4;   // batches in flight per producer lane (measured: fewer copy waves or one wave that adds straight from
                                // global memory cannot keep enough loads in flight to stream a chain at memory latency)

__global__ __launch_bounds__(kWave *(2 + kLProducers)) void bgkl_split_add(BgklArgs a, BgklSplit s) {
    __shared__ float4 s_k[2][kLBatchVec];   // {k of 4 rows} of (group, leaf), +0 where the leaf is out of reach
    __shared__ float s_y[kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t h = blockIdx.x / 7u, b = blockIdx.x % 7u;
    const uint32_t it0 = s.nb_first[8 * h + b], it1 = s.nb_first[8 * h + b + 1];
    if (it0 == it1) return;  // no trained model in this slot (uniform over the workgroup)
    const uint4 last = s.item_desc[it1 - 1];
    // batches of this chain that hold rows: the items are consecutive in the dense array, 4 batches each, all full but the last
    const int nb = (int)((it1 - it0 - 1) * kLBatches + (last.w - last.z + (uint32_t)kLBatch - 1u) / (uint32_t)kLBatch);
    const float4 *in = s.dense + (size_t)it0 * (kLItemRows / 4) * kWave;
    const unsigned long long *lm = s.labmask + (size_t)it0 * kLBatches;
    const int pw = wave - 2;   // wave 0 adds the k rows, wave 1 the rows with label 1, waves 2.. copy
    const bool two = pw >= 0 && pw < 16 - kLProducers;   // 16 chunks of 64 float4 per batch over 14 copy waves
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    // The ring of batches in flight: four slots as named variables (as arrays, even with compile-time indices, they were
    // placed in scratch memory and every access waited for it).  Slot u holds the batches = u (mod 4); a depth of 8 measured no faster (the
    // consumer's 64 dependent adds per batch and the barrier set the pace: ~870 cycles per batch).
    float4 ra0 = zero, rb0 = zero, ra1 = zero, rb1 = zero, ra2 = zero, rb2 = zero, ra3 = zero, rb3 = zero;
    unsigned long long mk0 = 0ull, mk1 = 0ull, mk2 = 0ull, mk3 = 0ull;
#define LA3DM_L_FETCH(bt_, RA, RB, MK)                                                                      \
    do {                                                                                                    \
        const int fb_ = (bt_);                                                                              \
        if (wave == 1) {                                                                                    \
            MK = fb_ < nb ? lm[fb_] : 0ull;                                                                 \
        } else if (wave >= 2) {                                                                             \
            RA = zero;                                                                                      \
            RB = zero;                                                                                      \
            if (fb_ < nb) {                                                                                 \
                RA = in[(size_t)fb_ * kLBatchVec + pw * kWave + lane];                                      \
                if (two) RB = in[(size_t)fb_ * kLBatchVec + (kLProducers + pw) * kWave + lane];             \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
    // one trip: consumers batch bt (its mask in MKP, the slot of bt), producers batch bt + 1 (slot RA / RB) into LDS and
    // batch bt + 1 + depth asked for
#define LA3DM_L_TRIP(u_, RA, RB, MK, MKP)                                                                   \
    do {                                                                                                    \
        const int bt = i + (u_), nxt = bt + 1;                                                              \
        if (wave == 0) {                                                                                    \
            if (bt >= 0 && bt < nb) {                                                                       \
                const float4 *t = s_k[bt & 1] + lane;                                                       \
                _Pragma("unroll") for (int g = 0; g < kLBatch / 4; ++g) {                                   \
                    const float4 q = t[g * kWave];                                                          \
                    acc += q.x;                                                                             \
                    acc += q.y;                                                                             \
                    acc += q.z;                                                                             \
                    acc += q.w;                                                                             \
                }                                                                                           \
            }                                                                                               \
        } else if (wave == 1) {                                                                             \
            if (bt >= 0 && bt < nb) {                                                                       \
                const unsigned long long m = MKP; /* uniform */                                             \
                if (m != 0ull) {                                                                            \
                    const float4 *t = s_k[bt & 1] + lane;                                                   \
                    for (int g = 0; g < kLBatch / 4; ++g) {                                                 \
                        const uint32_t bits = (uint32_t)(m >> (4 * g)) & 15u; /* uniform */                 \
                        if (bits == 0u) continue;                                                           \
                        const float4 q = t[g * kWave];                                                      \
                        if (bits & 1u) acc += q.x;                                                          \
                        if (bits & 2u) acc += q.y;                                                          \
                        if (bits & 4u) acc += q.z;                                                          \
                        if (bits & 8u) acc += q.w;                                                          \
                    }                                                                                       \
                }                                                                                           \
            }                                                                                               \
            if (bt >= 0) LA3DM_L_FETCH(bt + kLAddDepth, RA, RB, MKP);                                       \
        } else {                                                                                            \
            if (nxt < nb) {                                                                                 \
                float4 *t = s_k[nxt & 1];                                                                   \
                t[pw * kWave + lane] = RA;                                                                  \
                if (two) t[(kLProducers + pw) * kWave + lane] = RB;                                         \
            }                                                                                               \
            LA3DM_L_FETCH(nxt + kLAddDepth, RA, RB, MK);                                                    \
        }                                                                                                   \
        __syncthreads();                                                                                    \
    } while (0)
    LA3DM_L_FETCH(0, ra0, rb0, mk0);
    LA3DM_L_FETCH(1, ra1, rb1, mk1);
    LA3DM_L_FETCH(2, ra2, rb2, mk2);
    LA3DM_L_FETCH(3, ra3, rb3, mk3);
    float acc = 0.0f;
    for (int i = -1; i < nb; i += kLAddDepth) {   // bt = i + u is = u - 1 (mod 4): its mask sits in slot (u + 3) % 4
        LA3DM_L_TRIP(0, ra0, rb0, mk0, mk3);
        LA3DM_L_TRIP(1, ra1, rb1, mk1, mk0);
        LA3DM_L_TRIP(2, ra2, rb2, mk2, mk1);
        LA3DM_L_TRIP(3, ra3, rb3, mk3, mk2);
    }
#undef LA3DM_L_TRIP
#undef LA3DM_L_FETCH
    if (wave == 1) s_y[lane] = acc;
    __syncthreads();
    if (wave == 0) s.part[((size_t)h * 7u + b) * kWave + lane] = make_float2(s_y[lane], acc);
}

__global__ __launch_bounds__(kWave) void bgkl_split_apply(BgklArgs a, BgklSplit s) {
    const int lane = threadIdx.x;
    const uint32_t h = blockIdx.x;
    const uint32_t task = s.split_list[h];
    uint32_t blk, li;
    bool active;
    float px, py, pz;
    if (!bgkl_leaf(a, task, lane, blk, li, active, px, py, pz)) return;
    float A = a.alpha[li], B = a.beta[li];
    bool updated = false;
    for (uint32_t b = 0; b < 7; ++b) {
        if (a.nbr[7 * blk + b] < 0) continue;
        const float2 yk = s.part[((size_t)h * 7u + b) * kWave + lane];
        if (yk.y > 0.001f) {  // bgkloctomap.cpp:226-227
            A += yk.x;
            B += yk.y - yk.x;
            updated = true;
        }
    }
    bgkl_store(a, li, active, updated, A, B);
}

// ---------------------------------------------------------------------------
// Order-free accumulate mode (round 5; la3dm_set_option "bgk_sum" 1, the default — what round 3 did for BGKOctoMap).
// The reference's ybar = Ks * y and kbar = Ks.rowwise().sum() (bgklinference.h:86-87) are fp32 chains in row order, and the
// only reason for the split path's scratch replay above (bgkl_split_eval -> kernelize -> expand -> add: 11x the algorithmic
// bytes, profiles/r04/side_pmc_l.txt) is to reproduce that order across the waves that share a tile's rows.  Here every
// (row, leaf) pair adds the SAME fp32 k and k * label to DOUBLE sums; a neighbour's two sums are rounded to fp32 once — the
// correctly rounded values of what the fp32 chains approximate, whatever the order (a double sum of 10^5 fp32 terms is
// exact to ~2^-36 of an fp32 ulp) — and the per-neighbour gate kbar > 0.001f and the fp32 update in ExtendedBlock order
// (bgkloctomap.cpp:226-231) stay exactly as they are.  Checked against the restatement's own double-sum mode
// (oracle.set_sum_mode(1)): <= 1 ulp on alpha / beta, tests/test_bgkl_sum_gpu.py.
//   bgkl_predict_fuse_f64<kW>  the row-serial kernel with double sums in registers (lane = leaf: no atomics); kW = 8: every
//                              wave sums its own rows, the eight partial sums are added in wave order at the neighbour's end
//   bgkl_split_sum             one wave per 256-row item of a split tile: its two double sums per leaf -> part64
//   bgkl_split_apply64         one workgroup per split tile, wave = neighbour: the items' partial sums added in item order
//                              (deterministic), rounded, then the gated update in ExtendedBlock order
// The order-free split path needs 1 KB of scratch per item (the ordered one: 64 KB + 64 KB + 4 KB) and three launches.
// ---------------------------------------------------------------------------
// The throughput form (many tiles; the items of split tiles): one wave per tile or item, lane = leaf.  Rows are staged 64
// at a time through LDS (one coalesced read per lane instead of three wave-uniform loads per row, each waited for), the
// distance test runs on every lane, and — the sums being order-free — the hit (row, leaf) pairs are COMPACTED into a ring
// {d2, label, leaf} and evaluated densely, lane = pair: sqrt, d / ell, sin / cos and the formula at full lane utilisation
// instead of once per row that has any hit (68 % of the rows of a light tile, ~10 lanes each); k and k * label go to the
// leaf's double accumulators by ds_add_f64 (bgk_kernels.h bgk_predict_fuse_r: a native LDS atomic, 32 cycles).  Neighbours
// are taken one after the other — the gate is per neighbour —, the ring is drained at a neighbour's end.
// Bounding sphere of a tile's active leaves (centre of their box, radius to the farthest one), for the row cull of bgkl_rows_sum.
__device__ __forceinline__ void bgkl_tile_sphere(const bool active, const float px, const float py, const float pz, float &cx,
                                                 float &cy, float &cz, float &rad) {
    float mn[3] = {active ? px : __builtin_inff(), active ? py : __builtin_inff(), active ? pz : __builtin_inff()};
    float mx[3] = {active ? px : -__builtin_inff(), active ? py : -__builtin_inff(), active ? pz : -__builtin_inff()};
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], o));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o));
        }
    cx = 0.5f * (mn[0] + mx[0]);
    cy = 0.5f * (mn[1] + mx[1]);
    cz = 0.5f * (mn[2] + mx[2]);
    const float dx = px - cx, dy = py - cy, dz = pz - cz;
    float r = active ? sqrtf(dx * dx + dy * dy + dz * dz) : 0.0f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) r = fmaxf(r, __shfl_xor(r, o));
    rad = r;   // (NaN / inf positions give a NaN / inf radius: nothing is culled then)
}

constexpr int kLRing = 192;
struct __attribute__((aligned(16))) WaveLdsL {
    float4 rows[3][kWave];      // staged rows, one array per float4 of the row
    double acc_k[kWave];        // sum k of the current neighbour
    double acc_y[kWave];        // sum k * label
    uint32_t ring[kLRing][3];   // {d2, label, leaf}
};

// the rows [r0, r1) into the leaf's two double sums (left in L.acc_k / L.acc_y of lane = leaf; zeroed here)
// Round 5, row cull: a training row is a whole beam (or a hit point), registered in every block it has samples in; most beams of a
// neighbouring block pass the tile at more than ell.  While the 64 rows of a trip are staged (lane = row) each gets ONE distance —
// to the centre of the tile's bounding sphere, by the same routine — and a row whose distance exceeds radius + ell by more than a
// slack that covers the fp32 error of both evaluations (1e-5 of the coordinates' magnitude, two orders above it) cannot reach
// any leaf: d(p, row) >= d(c, row) - |p - c|.  Such a row would have added exactly nothing (its k is 0 at every leaf: the test
// below is d2 >= hit_d2 there), so skipping it is exact; non-finite distances are never culled.
__device__ __forceinline__ void bgkl_rows_sum(const BgklArgs &a, WaveLdsL &L, const int lane, const bool active, const float px,
                                              const float py, const float pz, const uint32_t r0, const uint32_t r1, const float cx,
                                              const float cy, const float cz, const float reach) {
    L.acc_k[lane] = 0.0;
    L.acc_y[lane] = 0.0;
    uint32_t tail = 0;
    auto c_eval = [&](uint32_t i) {
        const float d2 = __uint_as_float(L.ring[i][0]), lab = __uint_as_float(L.ring[i][1]);
        const uint32_t leaf = L.ring[i][2];
        const float kv = bgkl_row_kernel(sqrtf(d2), a.ell, a.inv_ell, a.sf2, a.trig);
        const float ky = kv * lab;
        const uint32_t ak = (uint32_t)(uintptr_t)&L.acc_k[leaf], ay = (uint32_t)(uintptr_t)&L.acc_y[leaf];
        asm volatile("ds_add_f64 %0, %1\n" : : "v"(ak), "v"((double)kv) : "memory");
        if (ky != 0.0f) asm volatile("ds_add_f64 %0, %1\n" : : "v"(ay), "v"((double)ky) : "memory");   // (+-0 adds nothing; NaN != 0)
    };
    for (uint32_t q = r0; q < r1; q += kWave) {
        const uint32_t nr = min(r1 - q, (uint32_t)kWave);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool keep = false;
        if ((uint32_t)lane < nr) {
            const size_t row = (size_t)q + lane;
            const float4 w0 = a.rowx[3 * row], w1 = a.rowx[3 * row + 1], w2 = a.rowx[3 * row + 2];
            L.rows[0][lane] = w0;
            L.rows[1][lane] = w1;
            L.rows[2][lane] = w2;
            const float dc2 = bgkl_seg_d2(cx, cy, cz, w0, w1, w2);
            const float mag = fmaxf(fmaxf(fmaxf(fabsf(w0.x), fabsf(w0.y)), fmaxf(fabsf(w0.z), fabsf(w0.w))),
                                    fmaxf(fmaxf(fabsf(w1.x), fabsf(w1.y)), fmaxf(fmaxf(fabsf(cx), fabsf(cy)), fabsf(cz))));
            const float lim = reach + 1e-5f * mag + 1e-4f;
            keep = !(dc2 > lim * lim && dc2 < 1e30f);   // (NaN anywhere: kept)
        }
        unsigned long long todo = __ballot(keep);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (todo) {
            const uint32_t j = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const float4 q0 = L.rows[0][j], q1 = L.rows[1][j], q2 = L.rows[2][j];
            const float d2 = bgkl_seg_d2(px, py, pz, q0, q1, q2);
            const bool hit = active && bgkl_row_counts(d2, a.hit_d2);
            const unsigned long long m = __ballot(hit);
            if (m == 0ull) continue;
            if (hit) {
                const uint32_t slot = tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                L.ring[slot][0] = __float_as_uint(d2);
                L.ring[slot][1] = __float_as_uint(q1.z);
                L.ring[slot][2] = (uint32_t)lane;
            }
            tail += (uint32_t)__popcll(m);
            if (tail > (uint32_t)(kLRing - kWave)) {
                // the full batches from the ring's end (the sums are order-free): the remainder stays at the front
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t rem = tail & 63u;
                for (uint32_t p = rem; p < tail; p += kWave) c_eval(p + lane);
                tail = rem;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t p = 0; p < tail; p += kWave)
        if (p + lane < tail) c_eval(p + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int kW>
__global__ __launch_bounds__(kW *kWave) void bgkl_predict_fuse_f64(BgklArgs a, const uint32_t *__restrict__ task_item) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t task = blockIdx.x;
    if (task >= a.n_tasks) return;
    if (task_item && task_item[2 * task] != 0xFFFFFFFFu) return;  // split tile
    uint32_t blk, li;
    bool active;
    float px, py, pz;
    if (!bgkl_leaf(a, task, lane, blk, li, active, px, py, pz)) return;
    float A = a.alpha[li], B = a.beta[li];
    bool updated = false;
    constexpr size_t kPartB = sizeof(double) * (size_t)kW * 2 * kWave;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kW > 1 ? kPartB : sizeof(WaveLdsL)];
    float tcx = 0.f, tcy = 0.f, tcz = 0.f, trad = 0.f;
    if constexpr (kW == 1) bgkl_tile_sphere(active, px, py, pz, tcx, tcy, tcz, trad);
    const float reach = trad * 1.000001f + sqrtf(a.hit_d2);
    double(*s_part)[2][kWave] = reinterpret_cast<double(*)[2][kWave]>(s_raw);   // kW > 1
    WaveLdsL &s_L = *reinterpret_cast<WaveLdsL *>(s_raw);                        // kW == 1

    for (int b = 0; b < 7; ++b) {
        const int32_t tb = a.nbr[7 * blk + b];
        if (tb < 0) continue;
        const uint32_t r0 = __builtin_amdgcn_readfirstlane(a.row_off[tb]), r1 = __builtin_amdgcn_readfirstlane(a.row_off[tb + 1]);
        double ysum = 0.0, ksum = 0.0;
        if constexpr (kW == 1) {
            if (r0 == r1) continue;   // (an empty model adds nothing and fails the gate)
            bgkl_rows_sum(a, s_L, lane, active, px, py, pz, r0, r1, tcx, tcy, tcz, reach);
            ysum = s_L.acc_y[lane];
            ksum = s_L.acc_k[lane];
        } else {
            // the latency form (few tiles, the GPU mostly empty): wave w takes rows w, w + kW, ... straight into its own
            // sums; the partial sums are added in wave order
            for (uint32_t j = r0 + (uint32_t)wave; j < r1; j += (uint32_t)kW) {
                const float4 q0 = a.rowx[3 * (size_t)j], q1 = a.rowx[3 * (size_t)j + 1], q2 = a.rowx[3 * (size_t)j + 2];
                const float d2 = bgkl_seg_d2(px, py, pz, q0, q1, q2);
                const bool hit = active && bgkl_row_counts(d2, a.hit_d2);
                if (__ballot(hit) == 0ull) continue;
                if (hit) {
                    const float kv = bgkl_row_kernel(sqrtf(d2), a.ell, a.inv_ell, a.sf2, a.trig);
                    ysum += (double)(kv * q1.z);
                    ksum += (double)kv;
                }
            }
            s_part[wave][0][lane] = ysum;
            s_part[wave][1][lane] = ksum;
            __syncthreads();
            if (wave == 0) {
                ysum = s_part[0][0][lane];
                ksum = s_part[0][1][lane];
#pragma unroll
                for (int w = 1; w < kW; ++w) {
                    ysum += s_part[w][0][lane];
                    ksum += s_part[w][1][lane];
                }
            }
            __syncthreads();
        }
        const float ybar = (float)ysum, kbar = (float)ksum;
        if (kbar > 0.001f) {  // bgkloctomap.cpp:226-227 (only wave 0 holds the sums)
            A += ybar;
            B += kbar - ybar;
            updated = true;
        }
    }
    if (wave == 0) bgkl_store(a, li, active, updated, A, B);
}

// one wave per item of a split tile (tile x neighbour x <= kLItemRows rows), lane = leaf
__global__ __launch_bounds__(kWave) void bgkl_split_sum(BgklArgs a, BgklSplit s) {
    __shared__ WaveLdsL L;
    const int lane = threadIdx.x;
    const uint32_t it = blockIdx.x;
    const uint4 dsc = s.item_desc[it];
    uint32_t blk, li;
    bool active;
    float px, py, pz;
    if (!bgkl_leaf(a, dsc.x, lane, blk, li, active, px, py, pz)) return;  // (split tiles always hold leaves)
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(dsc.z), r1 = __builtin_amdgcn_readfirstlane(dsc.w);
    float tcx, tcy, tcz, trad;
    bgkl_tile_sphere(active, px, py, pz, tcx, tcy, tcz, trad);
    bgkl_rows_sum(a, L, lane, active, px, py, pz, r0, r1, tcx, tcy, tcz, trad * 1.000001f + sqrtf(a.hit_d2));
    s.part64[(size_t)it * kWave + lane] = make_double2(L.acc_y[lane], L.acc_k[lane]);
}

__global__ __launch_bounds__(7 * kWave) void bgkl_split_apply64(BgklArgs a, BgklSplit s) {
    __shared__ float2 s_yk[7][kWave];
    const int lane = threadIdx.x & 63;
    const uint32_t b = threadIdx.x >> 6;
    const uint32_t h = blockIdx.x;
    const uint32_t task = s.split_list[h];
    {
        const uint32_t it0 = s.nb_first[8 * h + b], it1 = s.nb_first[8 * h + b + 1];
        double ysum = 0.0, ksum = 0.0;
        const double2 *p = s.part64 + (size_t)it0 * kWave + lane;
        uint32_t it = it0;
        for (; it + 8u <= it1; it += 8u) {   // eight 512-byte rows in flight
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)u * kWave];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ysum += v[u].x;
                ksum += v[u].y;
            }
            p += 8 * kWave;
        }
        for (; it < it1; ++it) {
            const double2 v = *p;
            ysum += v.x;
            ksum += v.y;
            p += kWave;
        }
        s_yk[b][lane] = make_float2((float)ysum, (float)ksum);
    }
    __syncthreads();
    if (b != 0u) return;
    uint32_t blk, li;
    bool active;
    float px, py, pz;
    if (!bgkl_leaf(a, task, lane, blk, li, active, px, py, pz)) return;
    float A = a.alpha[li], B = a.beta[li];
    bool updated = false;
    for (uint32_t q = 0; q < 7; ++q) {
        if (a.nbr[7 * blk + q] < 0) continue;
        const float2 yk = s_yk[q][lane];
        if (yk.y > 0.001f) {  // bgkloctomap.cpp:226-227
            A += yk.x;
            B += yk.y - yk.x;
            updated = true;
        }
    }
    bgkl_store(a, li, active, updated, A, B);
}

}  // namespace la3dm_dev
