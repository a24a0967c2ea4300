// devmap_depth3.h — the leaf-list and write-back + prune launches of the device-resident insert for block_depth 3
// (the YAML's depth: a block is 73 nodes, its finest layer is ONE trip of 64 cells), several test blocks per wave.
//
// Reference: leaf enumeration include/bgkoctomap/bgkoctree.h:62-147 (LeafIterator order), node write-back
// src/bgkoctomap/bgkoctomap.cpp:314-335, OcTree::prune src/bgkoctomap/bgkoctree.cpp:101-148 — the same results, node for node,
// as dm_leaves<> / dm_commit_prune of devmap_kernels.h (which stay the form of every other depth and the A/B form,
// LA3DM_DEPTH3=0; tests/test_devmap_gpu.py::test_fallback_switches_give_the_same_map).
//
// Why: those launches are chains of dependent loads per test block — slot -> states (-> parent states) -> alpha / beta -> stores —
// with ONE block per wave at a time.  At configs[4]'s 266 k test blocks that is 32 rounds of 8 192 resident waves, each round one
// chain's latency: dm_leaves<false> 60 us, dm_leaves<true> 117 us, dm_commit_prune 129 us for 19 / 155 / 190 MB of traffic
// (profiles/r06/devmap_timeline_1M.txt).  Here a wave takes kBatch (4 or 8) consecutive test blocks and issues every level of the chain
// for all of them together (positions behind the list's end are clamped, not predicated: no branch between the loads); the
// states of a leaf's two ancestors are read up front instead of by a climb that waits for each level (they share the block's
// cache line), and the prune's sibling tests of all the batch's blocks run side by side on the wave's lanes.
#pragma once
#include "devmap_kernels.h"

namespace la3dm_dev {

constexpr uint32_t kD3Npb = 73;     // nodes of a block_depth-3 block: 1 + 8 + 64
// kBatch = test blocks per wave and trip: 4 or 8 (8 sibling groups each: at most 8 fit the prune's lane map)

// LDS written by some lanes of a wave and read by others: the LDS queue of a wave is in order, so only the compiler has to
// keep the order (a release fence, even at wavefront scope, also waits for every global load in flight)
__device__ __forceinline__ void d3_lds_order() { asm volatile("" ::: "memory"); }

// dm_leaves<kEmit> for block_depth 3.  Grid: cdiv(blocks, 4 * kBatch) workgroups of 256 (+ the emit launch's work-counter
// workgroups behind them, as in dm_leaves).
template <bool kEmit, uint32_t kD3Batch>
__global__ __launch_bounds__(256) void dm_leaves_d3(const uint32_t *__restrict__ slot, const uint32_t *__restrict__ counters,
                                                   const uint8_t *__restrict__ S, const float *__restrict__ A,
                                                   const float *__restrict__ B, uint32_t *nleaf,
                                                   const uint32_t *__restrict__ leaf_off, uint32_t *leaf_key, float *alpha,
                                                   float *beta, uint32_t *leaf_node, LeafExtra x) {
    if (kEmit && x.t_key && blockIdx.x >= x.main_wgs) {   // (uniform over the workgroup)
        test_stats_wg(x.t_key, nleaf, counters[kCntTest], x.counters_w, blockIdx.x - x.main_wgs, gridDim.x - x.main_wgs);
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_test = counters[kCntTest];
    if (!kEmit && blockIdx.x == 0 && threadIdx.x == 0) nleaf[n_test] = 0;   // the scan runs over n_test + 1 counts
    const uint32_t end = kEmit ? min(x.t_end, n_test) : n_test;
    const uint32_t t0 = (kEmit ? x.t_begin : 0u) + (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kD3Batch;
    if (t0 >= end) return;
    const uint32_t nb = min(kD3Batch, end - t0);
    // level 1: slots (and the leaf offsets) of the batch
    uint32_t sl[kD3Batch], out[kD3Batch];
#pragma unroll
    for (uint32_t b = 0; b < kD3Batch; ++b) {
        const uint32_t tb = t0 + min(b, nb - 1u);
        sl[b] = slot[tb];
        out[b] = kEmit ? leaf_off[tb] : 0u;
    }
    // level 2: the state of the lane's finest cell (lane order = descending cell index) and of its parent
    const uint32_t c = 63u - lane;
    uint32_t s2[kD3Batch], s1[kD3Batch];
#pragma unroll
    for (uint32_t b = 0; b < kD3Batch; ++b) {
        const uint8_t *Sb = S + (size_t)sl[b] * kD3Npb;
        s2[b] = Sb[9u + c];
        s1[b] = Sb[1u + (c >> 3)];
    }
    // covering leaf (covering_leaf of devmap_kernels.h, unrolled: climb while the node is PRUNED; the root is never asked)
    uint32_t node[kD3Batch], key[kD3Batch];
    unsigned long long m[kD3Batch];
    bool head[kD3Batch];
#pragma unroll
    for (uint32_t b = 0; b < kD3Batch; ++b) {
        uint32_t d = 2u, i = c;
        if ((s2[b] & 7u) == kStatePruned) {
            d = 1u;
            i = c >> 3;
            if ((s1[b] & 7u) == kStatePruned) {
                d = 0u;
                i = 0u;
            }
        }
        const uint32_t span = 3u * (2u - d);
        head[b] = b < nb && c == (((i + 1u) << span) - 1u);   // highest cell of the leaf's interval
        m[b] = __ballot(head[b]);
        node[b] = sl[b] * kD3Npb + dm_layer_base(d) + i;
        key[b] = (d << 16) + i;
    }
    if (!kEmit) {
#pragma unroll
        for (uint32_t b = 0; b < kD3Batch; ++b) {
            if (b >= nb) break;
            if (sl[b] >= x.old_blocks) {   // created by this pass: default nodes, every finest cell a leaf (what was read above is not used)
                const size_t base = (size_t)sl[b] * kD3Npb;
                for (uint32_t i = lane; i < kD3Npb; i += 64u) {
                    x.A_w[base + i] = x.a0;
                    x.B_w[base + i] = x.b0;
                    x.S_w[base + i] = kStateUnknown;
                }
                if (lane == 0) nleaf[t0 + b] = 64u;
            } else if (lane == 0) {
                nleaf[t0 + b] = (uint32_t)__popcll(m[b]);
            }
        }
        return;
    }
    // level 3 (emit): alpha / beta of every lane's covering leaf (the lanes of one leaf read one address), then the stores
    float av[kD3Batch], bv[kD3Batch];
#pragma unroll
    for (uint32_t b = 0; b < kD3Batch; ++b) {
        av[b] = A[node[b]];
        bv[b] = B[node[b]];
    }
#pragma unroll
    for (uint32_t b = 0; b < kD3Batch; ++b) {
        if (head[b]) {
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[b] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[b], 0));
            const uint32_t o = out[b] + r;
            leaf_key[o] = key[b];
            alpha[o] = av[b];
            beta[o] = bv[b];
            leaf_node[o] = node[b];
        }
    }
}

// dm_commit_prune for block_depth 3 (a block's leaves are one trip of at most 64).  Grid: at most kCommitPruneWgs workgroups of
// 256, grid-stride over batches of kBatch test blocks; dynamic LDS: 4 waves x kBatch x prune_lds_stride(kD3Npb) bytes.
// The arrival / mailbox tail is dm_commit_prune's.
template <uint32_t kD3Batch>
__global__ __launch_bounds__(256) void dm_commit_prune_d3(const uint32_t *__restrict__ slot, uint32_t n_test,
                                                         const uint32_t *__restrict__ leaf_off, const uint32_t *__restrict__ leaf_node,
                                                         const uint32_t *__restrict__ leaf_key, const float *__restrict__ alpha,
                                                         const float *__restrict__ beta, const uint8_t *__restrict__ state, float *A,
                                                         float *B, uint8_t *S, uint32_t *counters, uint32_t *done,
                                                         volatile uint32_t *mailbox, uint32_t mailbox_seq) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dm_prune_smem[];
    __shared__ uint32_t s_last;
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t lstr = prune_lds_stride(kD3Npb);
    uint8_t *sW = dm_prune_smem + wv * kD3Batch * lstr;
    const uint32_t src_off = (kD3Npb + 1u) & ~1u;
    const uint32_t *__restrict__ nkp = leaf_key ? leaf_key : leaf_node;   // (block-sharded insert: a foreign leaf arrives with its key)
    const uint32_t nbat = (n_test + kD3Batch - 1u) / kD3Batch;
    for (uint32_t q = blockIdx.x * (blockDim.x >> 6) + wv; q < nbat; q += gridDim.x * (blockDim.x >> 6)) {
        const uint32_t t0 = q * kD3Batch, nb = min(kD3Batch, n_test - t0);
        // level 1: slots and leaf ranges
        uint32_t base[kD3Batch], l0[kD3Batch], nl[kD3Batch];
#pragma unroll
        for (uint32_t b = 0; b < kD3Batch; ++b) {
            const uint32_t tb = t0 + min(b, nb - 1u);
            base[b] = slot[tb] * kD3Npb;
            l0[b] = leaf_off[tb];
            nl[b] = leaf_off[tb + 1u] - l0[b];
        }
        // level 2: the blocks' states and the leaves' results
        uint32_t sa[kD3Batch], sb[kD3Batch], st[kD3Batch], nk[kD3Batch];
        float al[kD3Batch], be[kD3Batch];
#pragma unroll
        for (uint32_t b = 0; b < kD3Batch; ++b) {
            sa[b] = S[base[b] + lane];
            sb[b] = S[base[b] + 64u + min(lane, 8u)];
            const uint32_t l = l0[b] + min(lane, max(nl[b], 1u) - 1u);
            st[b] = state[l];
            nk[b] = nkp[l];
            al[b] = alpha[l];
            be[b] = beta[l];
        }
        // the blocks' states into LDS (the layers of the sibling test depend on each other), src = the node a collapsed chain ends in
#pragma unroll
        for (uint32_t b = 0; b < kD3Batch; ++b) {
            uint8_t *sS = sW + b * lstr;
            uint16_t *src = (uint16_t *)(sS + src_off);
            sS[lane] = (uint8_t)sa[b];
            src[lane] = (uint16_t)lane;
            if (lane < 9u) {
                sS[64u + lane] = (uint8_t)sb[b];
                src[64u + lane] = (uint16_t)(64u + lane);
            }
        }
        d3_lds_order();
        __builtin_amdgcn_wave_barrier();
        // write-back of the leaves Occupancy::update ran for (state bit 7), to the pool and to the LDS copy
#pragma unroll
        for (uint32_t b = 0; b < kD3Batch; ++b) {
            if (b < nb && lane < nl[b] && (st[b] & 0x80u)) {
                uint32_t node = nk[b];
                if (leaf_key) node = base[b] + dm_layer_base(nk[b] >> 16) + (nk[b] & 0xFFFFu);
                const uint8_t ns = (uint8_t)((st[b] & 3u) | kClassifiedBit);
                A[node] = al[b];
                B[node] = be[b];
                S[node] = ns;
                (sW + b * lstr)[node - base[b]] = ns;
            }
        }
        d3_lds_order();
        __builtin_amdgcn_wave_barrier();
        // OcTree::prune, bottom-up: layer 2 -> 1 (lane = (block, sibling group)), then layer 1 -> 0 (lane = block)
        unsigned long long any2, any1;
        {
            const uint32_t b = lane >> 3, g = lane & 7u;
            bool did = false;
            if (b < nb) {
                uint8_t *sS = sW + b * lstr;
                uint16_t *src = (uint16_t *)(sS + src_off);
                const uint32_t c0 = 9u + 8u * g;
                const uint8_t st0 = sS[c0] & 7u;
                if (st0 != kStatePruned && st0 != kStateUnknown) {
                    bool same = true;
#pragma unroll
                    for (uint32_t k = 1; k < 8; ++k) same &= (sS[c0 + k] & 7u) == st0;
                    if (same) {
                        const uint32_t par = 1u + g;
                        sS[par] = (uint8_t)((sS[par] & kClassifiedBit) | st0);
                        src[par] = src[c0];
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) sS[c0 + k] = (uint8_t)((sS[c0 + k] & ~7u) | kStatePruned);
                        did = true;
                    }
                }
            }
            any2 = __ballot(did);
        }
        d3_lds_order();
        __builtin_amdgcn_wave_barrier();
        {
            bool did = false;
            if (lane < nb) {
                uint8_t *sS = sW + lane * lstr;
                uint16_t *src = (uint16_t *)(sS + src_off);
                const uint8_t st0 = sS[1] & 7u;
                if (st0 != kStatePruned && st0 != kStateUnknown) {
                    bool same = true;
#pragma unroll
                    for (uint32_t k = 1; k < 8; ++k) same &= (sS[1u + k] & 7u) == st0;
                    if (same) {
                        sS[0] = (uint8_t)((sS[0] & kClassifiedBit) | st0);
                        src[0] = src[1];
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) sS[1u + k] = (uint8_t)((sS[1u + k] & ~7u) | kStatePruned);
                        did = true;
                    }
                }
            }
            any1 = __ballot(did);
        }
        d3_lds_order();
        __builtin_amdgcn_wave_barrier();
        if (any2 | any1) {   // (uniform)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's committed alpha / beta have left for the L2
#pragma unroll
            for (uint32_t b = 0; b < kD3Batch; ++b) {
                if (!(((any2 >> (8u * b)) & 0xFFull) | ((any1 >> b) & 1ull))) continue;   // (uniform)
                const uint8_t *sS = sW + b * lstr;
                const uint16_t *src = (const uint16_t *)(sS + src_off);
                for (uint32_t i = lane; i < kD3Npb; i += 64u) {
                    S[base[b] + i] = sS[i];
                    const uint32_t sr = src[i];
                    if (sr != i) {
                        const uint32_t a32 = __hip_atomic_load((const uint32_t *)&A[base[b] + sr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t b32 = __hip_atomic_load((const uint32_t *)&B[base[b] + sr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        A[base[b] + i] = __uint_as_float(a32);
                        B[base[b] + i] = __uint_as_float(b32);
                    }
                }
            }
        }
        d3_lds_order();
        __builtin_amdgcn_wave_barrier();   // (the next trip's staging overwrites the LDS copies)
    }
    if (!mailbox) return;   // (uniform)
    // arrival in two levels and the counter block to the host — dm_commit_prune's tail, word for word
    __syncthreads();
    if (threadIdx.x == 0) {
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
        uint32_t v = 0;
        if (threadIdx.x < (uint32_t)kCntWords) v = __hip_atomic_load(&counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dm_publish_wave(counters, mailbox, mailbox_seq);
        if (threadIdx.x < (uint32_t)kCntWords) counters[threadIdx.x] = counter_begin_value(threadIdx.x, v);
    }
}

}  // namespace la3dm_dev
