// devmap_sort.h — stable LSD radix sort of (key, value) pairs, 8 bits per pass, ONE launch per pass plus one histogram
// launch per sort, for the device-resident map's front end (devmap.hip).
//
// Why not the library sort: rocPRIM's Onesweep costs three dependent launches per pass (two state resets + the pass)
// and its merge-sort fallback 7-9 for the sizes that occur here; each dependent launch is ≈ 4.7 us on this GPU whatever it
// does, and a scan runs three sorts.  Same algorithm family (Onesweep: chained per-digit prefix with decoupled
// look-back), but the state cleans itself: the histogram launch clears status array 0 and the next sort's histogram,
// every pass clears the rows of the status array the NEXT pass uses.  Everything is zero between sorts.
//
// Stability (the voxel filter's fp32 centroid sums depend on it): a tile is 4 waves x 16 rows x 64 lanes of consecutive
// items, ranked row by row inside a wave (ballot matching), wave after wave, tile after tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace la3dm_dev {

constexpr uint32_t kRsThreads = 256, kRsWaves = 4, kRsRows = 16, kRsTile = kRsThreads * kRsRows;   // 4096 items per tile
constexpr uint32_t kRsErrStuck = 16u;
constexpr int kRsLook = 16;
constexpr uint32_t kRsHistCopies = 8;   // predecessor tiles whose status words are in flight at a time

struct RadixState {
    uint32_t *hist;       // [kRsHistCopies][4][256] digit counts of the sort in flight (zero before its histogram launch):
                          // workgroup b of the histogram launch adds into copy b % kRsHistCopies (an atomic on one
                          // address costs ~25 ns per workgroup, serialised), the passes add the copies up
    uint32_t *hist_next;  // the histogram of the NEXT sort: the histogram launch clears it
    uint32_t *status[2];  // [tiles][256] per array: bits 31..30 = 0 empty / 1 tile count / 2 inclusive prefix, low 30 bits = value
    uint32_t *ticket;     // [4] one tile ticket per pass (cleared by the histogram launch); used when a pass has more tiles
                          // than workgroups (devmap_scan.h explains why index-assigned tiles are only safe for one round)
};

struct RadixArgs {
    const uint32_t *k_in, *v_in;
    uint32_t *k_out, *v_out;
    uint32_t n, n_pass, pass;
    const uint32_t *n_dev;   // nullptr, or where the item count lives on the device (then n is only its bound: the launch was sized before
                             // the host knew the count — the test-block list, devmap.hip run_pass)
    uint32_t begin_bit;   // pass p sorts on key bits [begin_bit + 8 p, + 8)
    uint32_t *counters;
    int err_slot;
    uint32_t use_ticket;  // tiles > workgroups of the launch
};

// A pass has at most kRsResident workgroups (what the chip holds at once: 256 CUs x 3 workgroups at 159 VGPRs); workgroup
// b takes the tiles b, b + G, b + 2 G, ... in this order, and workgroups are handed out in order, so a tile only ever
// waits for tiles of workgroups that started before its own.  No ticket and no arrival counter: every device-scope atomic with a returned value is
// a ~2.5 us round trip, and on ONE address they serialise at ~25 ns per workgroup.
constexpr uint32_t kRsResident = 768;

__device__ __forceinline__ uint32_t rs_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rs_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exclusive prefix over the 256 threads' values (one per thread)
__device__ __forceinline__ uint32_t rs_scan256(uint32_t v, uint32_t tid, uint32_t *s_part) {
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)(tid & 63u) >= d) incl += o;
    }
    __syncthreads();
    if ((tid & 63u) == 63u) s_part[tid >> 6] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) off += s_part[w];
    return off + incl - v;
}

// digit counts of all passes in one sweep over the keys.  Also clears status array 0 for the first pass and the next
// sort's histogram.  Round 5 (launch chain): the keys come from a source functor — `src.begin(gtid, gsize)` once per thread,
// then `src(item, add)` for every item of a grid-stride loop over n_items, calling add(key) for each key the item holds — so that
// the kernel that PRODUCES the keys (voxel-grid cells, membership pairs) is this launch instead of one before it; n = keys in all.
struct RsPlainKeys {
    const uint32_t *keys;
    __device__ __forceinline__ void begin(uint32_t, uint32_t) const {}
    template <class Add>
    __device__ __forceinline__ void operator()(uint32_t i, Add &add) const { add(keys[i]); }
};

template <class Src>
__global__ __launch_bounds__(kRsThreads) void dm_radix_hist_src(Src src, uint32_t n, uint32_t n_items, uint32_t n_pass, uint32_t begin_bit,
                                                               RadixState st) {
    __shared__ uint32_t h[4][256];
    const uint32_t tid = threadIdx.x;
    for (uint32_t p = 0; p < 4; ++p) h[p][tid] = 0;
    const uint32_t n_tiles = (n + kRsTile - 1) / kRsTile;
    for (uint32_t i = blockIdx.x * kRsThreads + tid; i < n_tiles * 256u; i += gridDim.x * kRsThreads) st.status[0][i] = 0u;
    if (blockIdx.x == 0 && tid < 4u) st.ticket[tid] = 0u;
    if (blockIdx.x < kRsHistCopies)
        for (uint32_t p = 0; p < 4; ++p) st.hist_next[(blockIdx.x * 4u + p) * 256u + tid] = 0u;
    src.begin(blockIdx.x * kRsThreads + tid, gridDim.x * kRsThreads);
    __syncthreads();
    auto add = [&](const uint32_t k) {
        const unsigned long long act = __ballot(true);
        for (uint32_t p = 0; p < n_pass; ++p) {
            // the high digits of grid-cell keys are the same for whole waves: one add instead of 64 colliding LDS atomics
            const uint32_t d = (k >> (begin_bit + 8u * p)) & 255u, d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            if (__ballot(d == d0) == act) {
                if ((tid & 63u) == (uint32_t)__builtin_ctzll(act)) atomicAdd(&h[p][d0], (uint32_t)__popcll(act));
            } else {
                atomicAdd(&h[p][d], 1u);
            }
        }
    };
    for (uint32_t i = blockIdx.x * kRsThreads + tid; i < n_items; i += gridDim.x * kRsThreads) src(i, add);
    __syncthreads();
    for (uint32_t p = 0; p < n_pass; ++p) {
        const uint32_t c = h[p][tid];
        if (c) atomicAdd(&st.hist[((blockIdx.x % kRsHistCopies) * 4u + p) * 256u + tid], c);
    }
}

__global__ __launch_bounds__(kRsThreads) void dm_radix_pass(RadixArgs a, RadixState st) {
    __shared__ uint32_t s_key[kRsTile], s_val[kRsTile];
    __shared__ uint32_t s_wcnt[kRsWaves][256];
    __shared__ uint32_t s_start[256], s_gbase[256], s_part[kRsThreads / 64], s_tile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (a.n_dev) a.n = min(a.n, *a.n_dev);
    const uint32_t n_tiles = (a.n + kRsTile - 1) / kRsTile, shift = a.begin_bit + 8u * a.pass;
    uint32_t *status = st.status[a.pass & 1u], *other = st.status[(a.pass + 1u) & 1u];
    uint32_t total_d = 0;   // keys with digit tid in this pass (written by the histogram launch)
#pragma unroll
    for (uint32_t cpy = 0; cpy < kRsHistCopies; ++cpy) total_d += st.hist[(cpy * 4u + a.pass) * 256u + tid];
    // every key has the same digit in this pass (the top byte of grid-cell keys, mostly): the pass is a copy
    const bool copy_pass = __syncthreads_or(total_d == a.n) != 0;
    const uint32_t digit_base = rs_scan256(total_d, tid, s_part);   // first output position of digit tid
    auto next_tile = [&](uint32_t prev) -> uint32_t {
        if (!a.use_ticket) return prev + gridDim.x;
        __syncthreads();
        if (tid == 0) s_tile = atomicAdd(&st.ticket[a.pass], 1u);
        __syncthreads();
        return s_tile;
    };
    for (uint32_t tile = a.use_ticket ? next_tile(0u) : blockIdx.x; tile < n_tiles; tile = next_tile(tile)) {
    other[tile * 256u + tid] = 0u;   // my row of the array the next pass (or the next sort's second pass) uses
    if (copy_pass) {
        const uint32_t in_tile = min(kRsTile, a.n - tile * kRsTile);
        for (uint32_t j = tid; j < in_tile; j += kRsThreads) {
            a.k_out[tile * kRsTile + j] = a.k_in[tile * kRsTile + j];
            a.v_out[tile * kRsTile + j] = a.v_in[tile * kRsTile + j];
        }
        continue;
    }
#pragma unroll
    for (uint32_t w = 0; w < kRsWaves; ++w) s_wcnt[w][tid] = 0u;
    __syncthreads();
    // ---- load: wave w holds items [tile * 4096 + w * 1024, + 1024), row r = 64 consecutive items
    const uint32_t i0 = tile * kRsTile + wave * (kRsRows * 64u) + lane;
    uint32_t key[kRsRows], val[kRsRows], rank[kRsRows];
#pragma unroll
    for (uint32_t r = 0; r < kRsRows; ++r) {
        const uint32_t i = i0 + r * 64u;
        key[r] = i < a.n ? a.k_in[i] : 0xFFFFFFFFu;
        val[r] = i < a.n ? a.v_in[i] : 0u;
    }
    // ---- rank inside the wave, row after row: peers = lanes of the row with my digit
    const unsigned long long lt = (1ull << lane) - 1ull;
    volatile uint32_t *wc = s_wcnt[wave];
#pragma unroll
    for (uint32_t r = 0; r < kRsRows; ++r) {
        const bool valid = i0 + r * 64u < a.n;
        const uint32_t d = (key[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t prev = wc[d];
        rank[r] = prev + (uint32_t)__popcll(peers & lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0ull) wc[d] = prev + (uint32_t)__popcll(peers);   // the lowest peer
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- per digit (thread = digit): count in the tile, wave offsets, tile-local start
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t w = 0; w < kRsWaves; ++w) {
        const uint32_t c = s_wcnt[w][tid];
        s_wcnt[w][tid] = cnt;   // exclusive over the waves
        cnt += c;
    }
    // ---- chained prefix over the tiles, one digit per thread, kRsLook predecessors in flight (all tiles of a front-end
    // sort are resident at once and publish their counts together: a tile walks back over most of its predecessors)
    uint32_t excl = 0;
    if (tile == 0) {
        rs_st(&status[tid], (2u << 30) | cnt);
    } else {
        rs_st(&status[tile * 256u + tid], (1u << 30) | cnt);
    }
    s_start[tid] = rs_scan256(cnt, tid, s_part);
    if (tile != 0) {
        bool done = false;
        for (int idx = (int)tile - 1; !done; idx -= kRsLook) {
            uint32_t s[kRsLook];
#pragma unroll
            for (int u = 0; u < kRsLook; ++u) s[u] = idx - u >= 0 ? rs_ld(&status[(uint32_t)(idx - u) * 256u + tid]) : (2u << 30);
#pragma unroll
            for (int u = 0; u < kRsLook; ++u) {
                if (done) break;
                for (uint32_t spins = 0; (s[u] >> 30) == 0u; ++spins) {
                    s[u] = rs_ld(&status[(uint32_t)(idx - u) * 256u + tid]);
                    if (spins > (1u << 22)) {   // seconds: the state was not clean when the sort began — give up, flag it
                        atomicOr(&a.counters[a.err_slot], kRsErrStuck);
                        s[u] = 2u << 30;
                    }
                }
                excl += s[u] & 0x3FFFFFFFu;
                done = (s[u] >> 30) == 2u;
            }
        }
        rs_st(&status[tile * 256u + tid], (2u << 30) | (excl + cnt));
    }
    s_gbase[tid] = digit_base + excl;
    __syncthreads();
    // ---- reorder through LDS, then write runs of equal digits
#pragma unroll
    for (uint32_t r = 0; r < kRsRows; ++r) {
        if (i0 + r * 64u < a.n) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t pos = s_start[d] + s_wcnt[wave][d] + rank[r];
            s_key[pos] = key[r];
            s_val[pos] = val[r];
        }
    }
    __syncthreads();
    const uint32_t in_tile = min(kRsTile, a.n - tile * kRsTile);
    for (uint32_t j = tid; j < in_tile; j += kRsThreads) {
        const uint32_t k = s_key[j], d = (k >> shift) & 255u;
        const uint32_t g = s_gbase[d] + (j - s_start[d]);
        a.k_out[g] = k;
        a.v_out[g] = s_val[j];
    }
    __syncthreads();   // the LDS arrays are reused by the next tile
    }
}

}  // namespace la3dm_dev
