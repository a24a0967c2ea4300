// devmap_grid_keys.h — dm_grid_centroids (devmap_kernels.h) with the chunk descriptors' "does this chunk of 512 sorted positions lie
// in ONE cell?" answered from the SORTED CELL KEYS — keys[i0] == keys[i0 + 511] — instead of from the head-flag and prefix arrays of
// the segment scan.  Those two arrays (8 bytes per point) had no other reader in the voxel filter: the scan now leaves them
// unwritten (devmap.hip voxel_grid: 56 MB less per sample filter at configs[4]'s 7 M samples).  Same descriptors, same centroids
// (pcl::VoxelGrid's per-cell mean, call site src/bgkoctomap/bgkoctomap.cpp:419-431).
#pragma once
#include "devmap_kernels.h"

namespace la3dm_dev {

__device__ __forceinline__ void big_chunks_wave_keys(const uint32_t q, const int lane, const float *__restrict__ p,
                                                     const uint32_t *__restrict__ vals, const uint32_t *__restrict__ keys,
                                                     const uint32_t *__restrict__ counters, int valid_slot, uint4 *desc) {
    const uint32_t i0 = q * kChunk;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (i0 + kChunk <= counters[valid_slot]) {        // (the keys are sorted ascending, the invalid key behind every cell)
        if (keys[i0] == keys[i0 + kChunk - 1]) {
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

// the ordinary cells' centroids and the chunk descriptors in one launch, as dm_grid_centroids: the last main_wgs workgroups take one
// cell per thread, the workgroups before them one chunk per wave
__global__ __launch_bounds__(256) void dm_grid_centroids_keys(const float *__restrict__ p, const uint32_t *__restrict__ vals,
                                                             const uint32_t *__restrict__ seg_start, uint32_t *counters, int seg_slot,
                                                             int big_slot, uint4 *big, float *out, uint32_t main_wgs,
                                                             const uint32_t *__restrict__ keys, int valid_slot, uint32_t nchunk,
                                                             uint4 *desc, uint32_t big_cell) {
    const uint32_t chunk_wgs = gridDim.x - main_wgs;
    if (blockIdx.x >= chunk_wgs) {
        grid_centroids_thread((blockIdx.x - chunk_wgs) * blockDim.x + threadIdx.x, p, vals, seg_start, counters, seg_slot, big_slot, big, out,
                              big_cell);
        return;
    }
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q < nchunk) big_chunks_wave_keys(q, (int)(threadIdx.x & 63u), p, vals, keys, counters, valid_slot, desc);
}

}  // namespace la3dm_dev
