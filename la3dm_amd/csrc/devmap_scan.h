// devmap_scan.h — exclusive prefix sum in ONE launch (chained scan with decoupled look-back), for the device-resident
// map's front end (devmap.hip).
//
// Why not the library scan: a front-end stage is a chain of short dependent kernels, every one of which costs the
// ≈ 4.7 us a dependent dispatch takes on this GPU whatever it does; hipcub's scan is two launches (state reset + scan)
// and every use here needed one or two more for what surrounds it (head flags before, segment starts / the total
// after).  This kernel takes its input through a small "mode" switch and does the surrounding work itself:
//   mode plain : out[i] = sum of in[0..i), optionally counters[total_slot] = sum of all
//   mode heads : in = keys sorted ascending with the invalid key (0xFFFFFFFF) last; flag[i] = valid key that differs from
//                its predecessor; seg_start[k] = position of the k-th flagged element (+ seg_key[k] = its key),
//                seg_start[n_seg] = number of valid keys, counters[seg_slot] = n_seg, counters[valid_slot] = valid keys
// Two status arrays alternate between launches; a launch clears what the launch before the previous one left in the array
// the next launch will use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "devmap_kernels.h"   // the counter block (dm_publish_lane)

namespace la3dm_dev {

constexpr uint32_t kScanThreads = 512, kScanItems = 8, kScanTile = kScanThreads * kScanItems;
constexpr uint32_t kScanInvalid = 0xFFFFFFFFu;
constexpr int kScanLook = 2;   // status words per lane and look-back round

struct ScanState {
    unsigned long long *status;  // per tile: bits 63..62 = 0 empty / 1 tile aggregate / 2 inclusive prefix; low 32 bits = value
    unsigned long long *other;   // the status array of the NEXT launch: this launch clears its other_n used entries
    uint32_t other_n;
    uint32_t *ticket;            // nullptr: workgroup b takes the tiles b, b + G, ... (every tile of the launch has its own
                                 // workgroup); else the tiles are handed out through this counter (zero at launch)
    uint32_t *ticket_other;      // the ticket of the next launch: cleared here
};
// A launch has at most kScanResident workgroups.  With one workgroup per tile (n <= resident x 4096 elements — every
// front-end scan up to ~4 M elements) tile b belongs to workgroup b and only ever waits for tiles of workgroups with
// a lower index: the dispatcher hands workgroups out in index order per XCD, so the lowest unfinished tile always
// belongs to a workgroup that is running or can be dispatched as soon as a slot frees, whatever else occupies the GPU —
// no ticket, no arrival counter (a device-scope atomic on ONE address costs ~25 ns per workgroup, serialised: 390
// workgroups x (ticket + arrival count) were 20 us of a 23 us scan).  A launch with MORE tiles than workgroups cannot
// assign tiles by index: a workgroup in its second round would wait for a first-round tile whose workgroup may never
// be dispatched while the resident ones spin (the occupancy bound only holds on an exclusively owned, unmasked GPU —
// ADVICE r02).  Such launches hand their tiles out through an atomic ticket: a tile's predecessors then belong to
// workgroups that have already started.  The status arrays (and tickets) alternate between launches and are cleaned by
// the launch in between.
constexpr uint32_t kScanResident = 1024;

struct ScanArgs {
    const uint32_t *in;
    uint32_t *out;        // exclusive prefix per element (nullptr: not wanted)
    uint32_t n;
    uint32_t *counters;
    int total_slot;       // plain mode: counters[total_slot] = grand total (< 0: not wanted)
    // heads mode
    uint32_t *flag;       // head flag per element (nullptr: not wanted)
    uint32_t *seg_start;  // [n_seg + 1]
    uint32_t *seg_key;    // [n_seg] (nullptr: not wanted)
    int seg_slot, valid_slot;
    int zero_slot;        // counters[zero_slot] = 0 as a side effect (< 0: none)
    int err_slot;         // counters[err_slot] |= kScanErrStuck if a predecessor tile never reports (dirty state)
    volatile uint32_t *mailbox;   // the thread that writes the total / closes the segment list also publishes the counter block
    uint32_t mailbox_seq;         //   to the host (dm_publish_lane in devmap_kernels.h), or nullptr
};
constexpr uint32_t kScanErrStuck = 8u;

// A status word carries everything its reader needs (flag and value in one 64-bit access), so the accesses are relaxed
// device-scope atomics: acquire / release would add an L2 write-back before every store and an invalidate after every
// load of the spin loop (measured: a 1.6 M-element scan 90 us instead of 15).
__device__ __forceinline__ unsigned long long scan_ld(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void scan_st(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool kHeads>
__global__ __launch_bounds__(kScanThreads) void dm_scan_lb(ScanArgs a, ScanState st) {
    __shared__ uint32_t s_wave[kScanThreads / 64], s_excl, s_pub, s_tile;
    if (threadIdx.x == 0) s_pub = 0u;   // (two barriers lie between this and the first writer)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_tiles = (a.n + kScanTile - 1) / kScanTile;
    for (uint32_t e = blockIdx.x * kScanThreads + tid; e < st.other_n; e += gridDim.x * kScanThreads) st.other[e] = 0ull;
    if (blockIdx.x == 0 && tid == 0 && st.ticket_other) *st.ticket_other = 0u;
    auto next_tile = [&](uint32_t prev) -> uint32_t {
        if (!st.ticket) return prev + gridDim.x;
        __syncthreads();
        if (tid == 0) s_tile = atomicAdd(st.ticket, 1u);
        __syncthreads();
        return s_tile;
    };
    for (uint32_t tile = st.ticket ? next_tile(0u) : blockIdx.x; tile < n_tiles; tile = next_tile(tile)) {
    const uint32_t i0 = tile * kScanTile + tid * kScanItems;
    // ---- items
    uint32_t raw[kScanItems], v[kScanItems];
    if (i0 + kScanItems <= a.n && ((uintptr_t)a.in & 15u) == 0) {
        const uint4 q0 = *(const uint4 *)(a.in + i0), q1 = *(const uint4 *)(a.in + i0 + 4);
        raw[0] = q0.x; raw[1] = q0.y; raw[2] = q0.z; raw[3] = q0.w;
        raw[4] = q1.x; raw[5] = q1.y; raw[6] = q1.z; raw[7] = q1.w;
    } else {
#pragma unroll
        for (uint32_t u = 0; u < kScanItems; ++u) raw[u] = i0 + u < a.n ? a.in[i0 + u] : (kHeads ? kScanInvalid : 0u);
    }
    if (kHeads) {
        uint32_t prev = i0 > 0 && i0 <= a.n ? a.in[i0 - 1] : kScanInvalid;   // (element 0 is a head whenever it is valid)
#pragma unroll
        for (uint32_t u = 0; u < kScanItems; ++u) {
            v[u] = (raw[u] != kScanInvalid && (i0 + u == 0 || raw[u] != prev)) ? 1u : 0u;
            prev = raw[u];
        }
    } else {
#pragma unroll
        for (uint32_t u = 0; u < kScanItems; ++u) v[u] = raw[u];
    }
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < kScanItems; ++u) sum += v[u];
    // ---- workgroup scan of the thread sums
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, agg = 0;
#pragma unroll
    for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
        wave_off += w < wave ? s_wave[w] : 0u;
        agg += s_wave[w];
    }
    // ---- look-back (wave 0)
    if (wave == 0) {
        uint32_t excl = 0;
        if (tile == 0) {
            if (lane == 0) scan_st(&st.status[0], (2ull << 62) | agg);
        } else {
            if (lane == 0) scan_st(&st.status[tile], (1ull << 62) | agg);
            // kScanLook x 64 predecessors per round: a round is one round trip to the status words (~1 us), and the tiles of a
            // front-end scan all publish together, so tile t walks over all t of them (a 2 M-element scan has 489)
            int look = (int)tile - 1;
            for (;;) {
                unsigned long long s[kScanLook];
#pragma unroll
                for (int u = 0; u < kScanLook; ++u) {
                    const int idx = look - 64 * u - (int)lane;
                    s[u] = idx >= 0 ? scan_ld(&st.status[idx]) : (2ull << 62);
                }
                uint32_t part = 0;
                bool fin = false;
#pragma unroll
                for (int u = 0; u < kScanLook; ++u) {
                    if (fin) break;
                    const int idx = look - 64 * u - (int)lane;
                    for (uint32_t spins = 0; __any((s[u] >> 62) == 0ull); ++spins) {
                        if ((s[u] >> 62) == 0ull) s[u] = scan_ld(&st.status[idx]);
                        if (spins > (1u << 22)) {   // seconds: the state was not clean when the launch began — give up, flag it
                            if ((s[u] >> 62) == 0ull) {
                                atomicOr(&a.counters[a.err_slot], kScanErrStuck);
                                s[u] = 2ull << 62;
                            }
                        }
                    }
                    const unsigned long long pm = __ballot((s[u] >> 62) == 2ull);
                    const int first = pm ? __builtin_ctzll(pm) : 64;
                    part += (int)lane <= first ? (uint32_t)s[u] : 0u;
                    fin = pm != 0ull;
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
                excl += part;
                if (fin) break;
                look -= 64 * kScanLook;
            }
            if (lane == 0) scan_st(&st.status[tile], (2ull << 62) | (unsigned long long)(excl + agg));
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    uint32_t run = s_excl + wave_off + (incl - sum);   // exclusive prefix of this thread's first item
    const uint32_t after = kHeads && i0 + kScanItems < a.n ? a.in[i0 + kScanItems] : kScanInvalid;   // key behind my last item
    // ---- results.  A thread's eight prefixes (and head flags) leave as two 16-byte stores where the tile is full: eight
    // 4-byte stores 32 bytes apart across the lanes touched 64 sectors per instruction (a 2 M-element scan: 19 us)
    const bool vec = i0 + kScanItems <= a.n;
    const bool vec_out = vec && a.out && ((uintptr_t)a.out & 15u) == 0, vec_flag = kHeads && vec && a.flag && ((uintptr_t)a.flag & 15u) == 0;
    if (vec_out) {
        uint32_t pre[kScanItems], r = run;
#pragma unroll
        for (uint32_t u = 0; u < kScanItems; ++u) {
            pre[u] = r;
            r += v[u];
        }
        *(uint4 *)(a.out + i0) = make_uint4(pre[0], pre[1], pre[2], pre[3]);
        *(uint4 *)(a.out + i0 + 4) = make_uint4(pre[4], pre[5], pre[6], pre[7]);
    }
    if (vec_flag) {
        *(uint4 *)(a.flag + i0) = make_uint4(v[0], v[1], v[2], v[3]);
        *(uint4 *)(a.flag + i0 + 4) = make_uint4(v[4], v[5], v[6], v[7]);
    }
#pragma unroll
    for (uint32_t u = 0; u < kScanItems; ++u) {
        const uint32_t i = i0 + u;
        if (i < a.n) {
            if (a.out && !vec_out) a.out[i] = run;
            if (kHeads) {
                if (a.flag && !vec_flag) a.flag[i] = v[u];
                if (v[u]) {
                    a.seg_start[run] = i;
                    if (a.seg_key) a.seg_key[run] = raw[u];
                }
                // the last valid element closes the list (invalid keys sort last); no valid element at all: element 0 does
                const uint32_t next = u + 1 < kScanItems ? raw[(u + 1) % kScanItems] : after;
                if (raw[u] != kScanInvalid && (i + 1 == a.n || next == kScanInvalid)) {
                    const uint32_t n_seg = run + v[u];
                    a.counters[a.valid_slot] = i + 1;
                    a.counters[a.seg_slot] = n_seg;
                    a.seg_start[n_seg] = i + 1;
                    if (a.mailbox) s_pub = 1u;
                } else if (i == 0 && raw[u] == kScanInvalid) {
                    a.counters[a.valid_slot] = 0;
                    a.counters[a.seg_slot] = 0;
                    a.seg_start[0] = 0;
                    if (a.mailbox) s_pub = 1u;
                }
            } else if (i + 1 == a.n && a.total_slot >= 0) {
                a.counters[a.total_slot] = run + v[u];
                if (a.mailbox) s_pub = 1u;
            }
        }
        run += v[u];
    }
    if (tile == 0 && tid == 0 && a.zero_slot >= 0) a.counters[a.zero_slot] = 0;
    __syncthreads();   // s_wave / s_excl are reused by the next tile
    // The workgroup whose thread wrote the total / closed the segment list sends the counter block to the host's mailbox
    // (instead of a dm_publish_counters launch behind this one: a dependent dispatch costs ~4.7 us whatever it does).  One
    // wave, one word per lane — in the writing thread alone the 48 words cost 25 VGPRs, and this kernel must keep its
    // occupancy (the launch relies on all its workgroups being resident).  What earlier kernels wrote is visible since
    // the launch began, what this workgroup wrote since the barrier; counters other workgroups of this launch are still
    // writing (the stuck bit, a zeroed slot) are read by the host after a later, real publish.
    if (a.mailbox && s_pub) {
        dm_publish_wave(a.counters, a.mailbox, a.mailbox_seq);
        if (tid == 0) s_pub = 0u;
    }
    }
}

}  // namespace la3dm_dev
