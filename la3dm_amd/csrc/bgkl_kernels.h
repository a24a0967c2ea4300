// bgkl_kernels.h — BGKLOctoMap (block-level BGK with free-space line segments), SURVEY.md §8 row f4.
//
// Reference (CPU):
//   BGKLInference::predict        include/bgkloctomap/bgklinference.h:80-88
//   point_to_line_dist            include/bgkloctomap/bgklinference.h:104-140   (== seg_dist_dev, lv_kernels.h)
//   covSparseLine                 include/bgkloctomap/bgklinference.h:186-200   (d / ell, formula, `< 0 -> 0`)
//   7-neighbour update loop       src/bgkloctomap/bgkloctomap.cpp:206-231       (gate: kbar > 0.001)
//   Occupancy::update             src/bgkloctomap/bgkloctree_node.cpp:31-44     (same node as BGKOctoMap)
//
// Training rows are 8 floats {x0, y0, z0, x1, y1, z1, label, 0}: hits are degenerate segments with label 1,
// each beam that has a sample inside the block contributes its segment once with label 0.  One wave64 = one
// leaf tile, lane = leaf; rows are wave-uniform (scalar loads); the distance runs on every lane, the kernel
// evaluation only when some lane of the tile lies within ell of the segment.  Sums run in row order, so the
// results are bit-identical to the CPU restatement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace la3dm_dev {

struct BgklArgs {
    const float *rows;           // 8 floats per row, grouped by training block
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
};

__global__ __launch_bounds__(kWave) void bgkl_predict_fuse_kernel(BgklArgs a) {
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
    const float px = off4.x + a.blk_center[3 * blk + 0], py = off4.y + a.blk_center[3 * blk + 1],
                pz = off4.z + a.blk_center[3 * blk + 2];  // Block::get_loc
    float A = a.alpha[li], B = a.beta[li];
    bool updated = false;

    for (int b = 0; b < 7; ++b) {
        const int32_t tb = a.nbr[7 * blk + b];
        if (tb < 0) continue;
        const uint32_t r0 = __builtin_amdgcn_readfirstlane(a.row_off[tb]), r1 = __builtin_amdgcn_readfirstlane(a.row_off[tb + 1]);
        float ybar = 0.0f, kbar = 0.0f;
        for (uint32_t j = r0; j < r1; ++j) {
            const float4 p0 = *reinterpret_cast<const float4 *>(a.rows + 8 * (size_t)j);       // x0 y0 z0 x1
            const float4 p1 = *reinterpret_cast<const float4 *>(a.rows + 8 * (size_t)j + 4);   // y1 z1 label -
            const float d = seg_dist_dev(px, py, pz, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y);
            const bool hit = active && d < a.ell;  // d >= ell  =>  d / ell >= 1  =>  the kernel is <= 0 and cleaned to 0
            if (__ballot(hit) == 0ull) continue;
            if (hit) {
                const float kv = cov_sparse<true, 0>(d / a.ell, a.sf2);
                ybar += kv * p1.z;
                kbar += kv;
            }
        }
        if (kbar > 0.001f) {  // bgkloctomap.cpp:226-227
            A += ybar;
            B += kbar - ybar;
            updated = true;
        }
    }
    if (active) {
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
}

}  // namespace la3dm_dev
