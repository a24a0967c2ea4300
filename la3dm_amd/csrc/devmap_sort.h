// devmap_sort.h — stable LSD radix sort of (key, value) pairs, 8 bits per pass, ONE launch per pass plus one histogram
// launch per sort, for the device-resident map's front end (devmap.hip).
//
// Why not the library sort: rocPRIM's Onesweep costs three dependent launches per pass (two state resets + the pass)
// and its merge-sort fallback 7-9 for the sizes that occur here; each dependent launch is ≈ 4.7 us on this GPU whatever it
// does, and a scan runs three sorts.  Same algorithm family (Onesweep: per-digit prefix over the tiles through status words in
// memory), but the state cleans itself: the histogram launch clears status / group array 0 and the next sort's histogram,
// every pass clears the rows of the arrays the NEXT pass uses.  Everything a sort reads was cleared by the launch before.
//
// Stability (the voxel filter's fp32 centroid sums depend on it): a tile is 4096 (or 8192) consecutive items, a wave holds kRows rows
// of 64 consecutive items, ranked row by row inside the wave (ballot matching), wave after wave, tile after tile.
//
// Round 6: where a pass's time goes (wall_clock64 stamps per tile, -DLA3DM_RS_TRACE, tools/check/sort_trace.py; 534 k pairs =
// 131 tiles, all resident): every dependent trip to memory is ~1 us on this GPU — histogram, keys, the neighbours' status
// words — and the 15.5 us pass was  1.1 (histogram) + 1.3 (keys) + 4.0 (ranking) + 0.4 + 0.2 … 7 (tile prefix) + 1.0 + 1.5.
//   * ranking: the wave's LDS counters went through a `volatile` pointer = flat, system-coherent accesses behind
//     s_waitcnt vmcnt(0); now ds_read / ds_write (wavefront-scope atomics), 9 -> 5 VALU per key bit; sixteen waves rank
//     four rows each where the whole sort fits on the chip that way (<= 256 tiles): 4.0 -> 1.5 us;
//   * tile prefix: all tiles publish together, so decoupled look-back walked over EVERY predecessor (131 tiles: 8.6 MB of
//     agent-scope loads, the same 6 us with 16 or 48 words in flight); now two levels — the tiles before mine in my group of
//     16, and one entry per earlier group, published by the group's last tile: 7 -> 4.5 us for the last tile;
//   * the first tile's keys are requested before the histogram is read; __syncthreads_or (three barriers around an LDS
//     atomic per thread) replaced by one barrier.
// 15.5 -> 9.7 us in the kernel; the insert of configs[1] 0.576 -> 0.507 ms (ten passes per insert; with the scan's share).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace la3dm_dev {

constexpr uint32_t kRsThreads = 256;   // histogram launch (thread = digit)
constexpr uint32_t kRsTile = 4096;     // items per tile (the status arrays are sized for this; the long sorts' shape takes 8192)
constexpr uint32_t kRsErrStuck = 16u;
constexpr int kRsLook = 16;          // status words per thread in flight in a round of the tile prefix
constexpr uint32_t kRsGroup = 16;   // tiles per group of the two-level prefix (<= kRsLook: one thread reads its group's predecessors in one go)
constexpr uint32_t kRsHistCopies = 8;   // copies of the histogram the histogram launch's workgroups add into (see RadixState.hist)

struct RadixState {
    uint32_t *hist;       // [kRsHistCopies][4][256] digit counts of the sort in flight (zero before its histogram launch):
                          // workgroup b of the histogram launch adds into copy b % kRsHistCopies (an atomic on one
                          // address costs ~25 ns per workgroup, serialised), the passes add the copies up
    uint32_t *hist_next;  // the histogram of the NEXT sort: the histogram launch clears it
    uint32_t *status[2];  // [tiles][256] per array: bits 31..30 = 0 empty / 1 tile count, low 30 bits = value
    uint32_t *agg[2];     // [tiles / kRsGroup][256] per array, one row per group of kRsGroup tiles, written by the group's last tile:
                          // bits 31..30 = 0 empty / 1 the group's count / 2 inclusive prefix through the group
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

// A pass has at most one workgroup per CU (devmap.hip checks with the occupancy calculator, when a map is created, that the chip
// holds that many at once); workgroup b takes the tiles b, b + G, b + 2 G, ... in this order, and workgroups are handed out in
// order, so a tile only ever waits for tiles of workgroups that started before its own.  No ticket and no arrival counter while
// every tile has its own workgroup: every device-scope atomic with a returned value is a ~2.5 us round trip, and on ONE address
// they serialise at ~25 ns per workgroup.

__device__ __forceinline__ uint32_t rs_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rs_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exclusive prefix over the first 256 threads' values (one per thread; the other threads of the workgroup pass 0 and take part in the barriers)
__device__ __forceinline__ uint32_t rs_scan_digits(uint32_t v, uint32_t tid, uint32_t *s_part) {
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

// "does any digit's thread (tid < 256) say yes", told to the whole workgroup with one barrier (__syncthreads_or is three barriers
// around an LDS atomic per thread).  A second call may only follow a barrier behind the first one's return.
__device__ __forceinline__ bool rs_any_digit(bool pred, uint32_t tid, uint32_t *s_flag) {
    const bool w = __ballot(pred) != 0ull;
    if (tid < 256u && (tid & 63u) == 0u) s_flag[tid >> 6] = w ? 1u : 0u;
    __syncthreads();
    return (s_flag[0] | s_flag[1] | s_flag[2] | s_flag[3]) != 0u;
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
    for (uint32_t i = blockIdx.x * kRsThreads + tid; i < (n_tiles / kRsGroup) * 256u; i += gridDim.x * kRsThreads) st.agg[0][i] = 0u;
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

// -DLA3DM_RS_TRACE: wall_clock64 stamps (10 ns) of thread 0 at the phases of every tile's life, read back and printed by
// la3dm_devmap_diag_sort (tools/check/sort_trace.py); not part of a product build
#ifdef LA3DM_RS_TRACE
__device__ unsigned long long g_rs_trace[4 * 1024 * 8];
#define RS_STAMP(k)                                                                                       \
    do {                                                                                                  \
        if (tid == 0 && tile < 1024u) g_rs_trace[(a.pass * 1024u + tile) * 8u + (k)] = wall_clock64();   \
    } while (0)
#else
#define RS_STAMP(k) do {} while (0)
#endif

// Two shapes of the tile, sixteen waves and one workgroup per CU both.  <1024, 4>: 4096 items, four rows per wave — the shortest
// tile life (ranking is VALU work); for sorts whose tiles all fit on the chip that way (<= 1 M items: the cloud's filter, the
// membership pairs, the test list).  <1024, 8>: 8192 items, everything longer — up to 2 M items (the free samples' filter at
// configs[1]) every tile is still on the chip at once, half as many tiles to add up (17.6 us per pass; round 5's four waves x 16
// rows with this round's prefix: 19.6); beyond that the tiles go out through the ticket (10 M items: 56 us per pass against 66).
template <uint32_t kThreads, uint32_t kRows>
__global__ __launch_bounds__(kThreads, 4) void dm_radix_pass(RadixArgs a, RadixState st) {
    constexpr uint32_t kWaves = kThreads / 64u;
    constexpr uint32_t kReaders = kThreads / 256u - 1u;   // thread groups of 256 beside the digits' own: they read the group entries of the tile prefix
    constexpr uint32_t kTile = kThreads * kRows;   // items per tile: 4096, or 8192 in the shape for the long sorts
    static_assert(kTile % kRsTile == 0u && kThreads % 256u == 0u && kThreads >= 512u, "tile shape");
    __shared__ uint32_t s_key[kTile], s_val[kTile];
    __shared__ uint32_t s_wcnt[kWaves][256];
    __shared__ uint32_t s_start[256], s_gbase[256], s_part[kWaves], s_tile;
    __shared__ uint32_t s_lb[kReaders][256];      // look-back: the reader groups' partial sums, bit 31 = the group met an inclusive prefix
    __shared__ uint32_t s_in[256];                // look-back: sum over the tiles before mine in my group
    __shared__ uint32_t s_done[256];              // look-back: digit finished (its own thread tells the groups)
    __shared__ uint32_t s_flag[4];                // rs_any_digit
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t dg = tid & 255u, grp = tid >> 8;   // look-back: digit and group (which predecessors of a round) of this thread
    const bool digit_thread = tid < 256u;             // thread = digit wherever a step is per digit
    if (a.n_dev) a.n = min(a.n, *a.n_dev);
    const uint32_t n_tiles = (a.n + kTile - 1) / kTile, shift = a.begin_bit + 8u * a.pass;
    uint32_t *status = st.status[a.pass & 1u], *other = st.status[(a.pass + 1u) & 1u];
    uint32_t *agg = st.agg[a.pass & 1u], *other_agg = st.agg[(a.pass + 1u) & 1u];
#ifdef LA3DM_RS_TRACE
    const unsigned long long t_entry = wall_clock64();
#endif
    auto next_tile = [&](uint32_t prev) -> uint32_t {
        if (!a.use_ticket) return prev + gridDim.x;
        __syncthreads();
        if (tid == 0) s_tile = atomicAdd(&st.ticket[a.pass], 1u);
        __syncthreads();
        return s_tile;
    };
    // ---- a tile's items: wave w holds items [tile * 4096 + w * kRows * 64, + kRows * 64), row r = 64 consecutive items
    uint32_t key[kRows], val[kRows], rank[kRows];
    auto load_tile = [&](uint32_t t) {
        const uint32_t j0 = t * kTile + wave * (kRows * 64u) + lane;
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            const uint32_t i = j0 + r * 64u;
            key[r] = i < a.n ? a.k_in[i] : 0xFFFFFFFFu;
            val[r] = i < a.n ? a.v_in[i] : 0u;
        }
    };
    uint32_t tile = a.use_ticket ? next_tile(0u) : blockIdx.x;
    bool loaded = tile < n_tiles;
    if (loaded) load_tile(tile);   // the first tile's items are on their way while the histogram is read (a round trip each, ~1 us)
    uint32_t total_d = 0;   // keys with digit tid in this pass (written by the histogram launch)
    if (digit_thread) {
#pragma unroll
        for (uint32_t cpy = 0; cpy < kRsHistCopies; ++cpy) total_d += st.hist[(cpy * 4u + a.pass) * 256u + tid];
    }
    // every key has the same digit in this pass (the top byte of grid-cell keys, mostly): the pass is a copy
    const bool copy_pass = rs_any_digit(digit_thread && total_d == a.n, tid, s_flag);
    const uint32_t digit_base = rs_scan_digits(total_d, tid, s_part);   // first output position of digit tid
    for (; tile < n_tiles; tile = next_tile(tile), loaded = false) {
#ifdef LA3DM_RS_TRACE
    if (tid == 0 && tile < 1024u) g_rs_trace[(a.pass * 1024u + tile) * 8u + 0] = t_entry;
#endif
    RS_STAMP(1);
    if (digit_thread) {   // my rows of the arrays the next pass (or the next sort's second pass) uses
        other[tile * 256u + tid] = 0u;
        if (tile % kRsGroup == kRsGroup - 1u) other_agg[(tile / kRsGroup) * 256u + tid] = 0u;
    }
    if (copy_pass) {
        const uint32_t in_tile = min(kTile, a.n - tile * kTile);
        for (uint32_t j = tid; j < in_tile; j += kThreads) {
            a.k_out[tile * kTile + j] = a.k_in[tile * kTile + j];
            a.v_out[tile * kTile + j] = a.v_in[tile * kTile + j];
        }
        continue;
    }
    if (digit_thread) {
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) s_wcnt[w][tid] = 0u;
    }
    __syncthreads();
    const uint32_t i0 = tile * kTile + wave * (kRows * 64u) + lane;
    if (!loaded) load_tile(tile);
#ifdef LA3DM_RS_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    RS_STAMP(2);
    // ---- rank inside the wave, row after row: peers = lanes of the row with my digit
    // (the wave's counters are read and written through wavefront-scope atomics: a `volatile` pointer made every access a flat,
    // system-coherent load / store behind s_waitcnt vmcnt(0) — behind the tile's key loads —, most of the 4 us the ranking took)
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t *wc = s_wcnt[wave];
#pragma unroll
    for (uint32_t r = 0; r < kRows; ++r) {
        const bool valid = i0 + r * 64u < a.n;
        const uint32_t d = (key[r] >> shift) & 255u;
        uint32_t mlo = 0, mhi = 0;   // lanes whose digit differs from mine in some bit
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const int32_t m = __builtin_amdgcn_sbfe((int32_t)key[r], shift + b, 1u);   // my bit b, as 0 / -1
            const unsigned long long bal = __ballot(m != 0);
            mlo |= (uint32_t)bal ^ (uint32_t)m;
            mhi |= (uint32_t)(bal >> 32) ^ (uint32_t)m;
        }
        const unsigned long long peers = __ballot(valid) & ~(((unsigned long long)mhi << 32) | mlo);
        const uint32_t prev = __hip_atomic_load(&wc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        rank[r] = prev + (uint32_t)__popcll(peers & lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0ull) __hip_atomic_store(&wc[d], prev + (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // the lowest peer
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    RS_STAMP(3);
    // ---- per digit (thread = digit): count in the tile, wave offsets, tile-local start
    uint32_t cnt = 0;
    if (digit_thread) {
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) {
            const uint32_t c = s_wcnt[w][tid];
            s_wcnt[w][tid] = cnt;   // exclusive over the waves
            cnt += c;
        }
        rs_st(&status[tile * 256u + tid], (1u << 30) | cnt);   // published: my count of digit tid
    }
    const uint32_t start = rs_scan_digits(cnt, tid, s_part);
    if (digit_thread) s_start[tid] = start;
    RS_STAMP(4);
    // ---- exclusive prefix over the tiles before this one, two levels.  All tiles of a front-end sort are resident at once and
    // publish their counts together, so a plain decoupled look-back walks over every predecessor: 131 tiles read 8.6 MB of
    // status words through agent-scope loads, 6 us of a 15 us pass, however many were in flight per round
    // (tools/check/sort_trace.py).  Now tiles come in groups of kRsGroup: a tile adds up the counts of the tiles before it in
    // its own group (<= 15 words per digit, read by the digit's own thread) and the group entries before its group, which
    // each group's last tile publishes: first the group's sum (state 1: needs nothing but its own group's counts, so no
    // group waits for an earlier one), then, once it knows its own prefix, the inclusive prefix through its group (state 2: cuts the
    // walk over the groups short in the long sorts, where tiles run in rounds).  The thread groups beside the digits' own read 16 group
    // entries each per round.
    const uint32_t q = tile % kRsGroup, grp_idx = tile / kRsGroup;
    const bool leader = q == kRsGroup - 1u;
    const bool reader = grp >= 1u;
    const uint32_t ri = grp - 1u;
    constexpr int kAgg = kRsLook;   // group entries per reader and round
    uint32_t sa[kAgg];
    auto agg_issue = [&](uint32_t base) {
        const int g0 = (int)grp_idx - 1 - (int)(base + ri * (uint32_t)kAgg);
#pragma unroll
        for (int u = 0; u < kAgg; ++u) sa[u] = g0 - u >= 0 ? rs_ld(&agg[(uint32_t)(g0 - u) * 256u + dg]) : (2u << 30);
    };
    uint32_t excl = 0;
    bool done = false;
    if (tile != 0u) {   // (uniform; tile 0 has nothing before it — a one-tile sort, every pass of a small scan's filters, skips the whole exchange)
        if (digit_thread) {   // the tiles before mine in my group
            uint32_t s[kRsLook], part = 0;
#pragma unroll
            for (uint32_t u = 0; u < kRsLook; ++u) s[u] = u < q ? rs_ld(&status[(tile - 1u - u) * 256u + tid]) : (1u << 30);
#pragma unroll
            for (uint32_t u = 0; u < kRsLook; ++u) {
                for (uint32_t spins = 0; (s[u] >> 30) == 0u; ++spins) {
                    s[u] = rs_ld(&status[(tile - 1u - u) * 256u + tid]);
                    if (spins > (1u << 22)) {   // seconds: the state was not clean when the sort began — give up, flag it
                        atomicOr(&a.counters[a.err_slot], kRsErrStuck);
                        s[u] = 1u << 30;
                    }
                }
                part += s[u] & 0x3FFFFFFFu;
            }
            if (leader) rs_st(&agg[grp_idx * 256u + tid], (1u << 30) | (part + cnt));
            s_in[tid] = part;
            s_done[tid] = 0u;
        }
        for (uint32_t base = 0;; base += kReaders * (uint32_t)kAgg) {
            uint32_t part = 0;
            bool found = false;
            if (reader && !(base != 0u && s_done[dg])) {
                agg_issue(base);
                const int g0 = (int)grp_idx - 1 - (int)(base + ri * (uint32_t)kAgg);
#pragma unroll
                for (int u = 0; u < kAgg; ++u) {
                    if (found) break;
                    for (uint32_t spins = 0; (sa[u] >> 30) == 0u; ++spins) {
                        sa[u] = rs_ld(&agg[(uint32_t)(g0 - u) * 256u + dg]);
                        if (spins > (1u << 22)) {
                            atomicOr(&a.counters[a.err_slot], kRsErrStuck);
                            sa[u] = 2u << 30;
                        }
                    }
                    part += sa[u] & 0x3FFFFFFFu;
                    found = (sa[u] >> 30) == 2u;
                }
            }
            if (reader) s_lb[ri][dg] = part | (found ? 0x80000000u : 0u);
            __syncthreads();
            if (digit_thread && !done) {
                if (base == 0u) excl = s_in[tid];
#pragma unroll
                for (uint32_t g = 0; g < kReaders; ++g) {
                    if (done) break;
                    const uint32_t v = s_lb[g][tid];
                    excl += v & 0x7FFFFFFFu;
                    done = (v >> 31) != 0u;
                }
                s_done[tid] = done ? 1u : 0u;
            }
            if (!rs_any_digit(digit_thread && !done, tid, s_flag)) break;
        }
    }
    if (digit_thread) {
        if (leader) rs_st(&agg[grp_idx * 256u + tid], (2u << 30) | (excl + cnt));
        s_gbase[tid] = digit_base + excl;
    }
    __syncthreads();
    RS_STAMP(5);
    // ---- reorder through LDS, then write runs of equal digits
#pragma unroll
    for (uint32_t r = 0; r < kRows; ++r) {
        if (i0 + r * 64u < a.n) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t pos = s_start[d] + s_wcnt[wave][d] + rank[r];
            s_key[pos] = key[r];
            s_val[pos] = val[r];
        }
    }
    __syncthreads();
    RS_STAMP(6);
    const uint32_t in_tile = min(kTile, a.n - tile * kTile);
    for (uint32_t j = tid; j < in_tile; j += kThreads) {
        const uint32_t k = s_key[j], d = (k >> shift) & 255u;
        const uint32_t g = s_gbase[d] + (j - s_start[d]);
        a.k_out[g] = k;
        a.v_out[g] = s_val[j];
    }
    RS_STAMP(7);
    __syncthreads();   // the LDS arrays are reused by the next tile
    }
}

}  // namespace la3dm_dev
