// devmap.hip — device-resident map: BGKOctoMap::insert_pointcloud start to finish on the GPU
// (C ABI: the la3dm_devmap_* entry points of include/la3dm_hip.h; kernels: devmap_kernels.h).
// The host only sizes launches (a handful of scalar read-backs per scan) and walks the float-stepped
// candidate loops of get_blocks_in_bbox (a few dozen iterations per axis).  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>   // (before rocPRIM: one of its iterator headers calls memset unqualified)

#include <rocprim/rocprim.hpp>

#include <string>
#include <vector>

#include "la3dm_ctx.h"
#include "devmap_kernels.h"
#include "devmap_scan.h"
#include "devmap_sort.h"
#include "devmap_lv_kernels.h"
#include "devmap_depth3.h"
#include "devmap_grid_keys.h"

using namespace la3dm_dev;

namespace {
double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
inline uint32_t npb_of(int depth) {
    uint32_t n = 0;
    for (int d = 0; d < depth; ++d) n += 1u << (3 * d);
    return n;
}
}  // namespace

struct la3dm_devmap {
    la3dm_ctx *ctx = nullptr;
    uint32_t depth = 0, npb = 0, ncell = 0;
    float block_size = 0.f, init_A = 0.f, init_B = 0.f;
    // pool + table
    size_t cap_blocks = 0;
    uint32_t n_blocks = 0;
    bool insert_into_empty = false;  // this insert started on a map without blocks: every test block of its first pass is new, i.e. un-pruned
    float *A = nullptr, *B = nullptr;
    uint8_t *S = nullptr;
    long long *blk_key = nullptr;
    uint32_t tab_cap = 0;
    long long *tab_key = nullptr;
    uint32_t *tab_val = nullptr;
    // small fixed buffers
    uint32_t *d_cnt = nullptr, *h_cnt = nullptr;  // counters (device / pinned host)
    uint32_t *d_mm = nullptr;   // [0..5] encoded min / max, [6] arrival counter of the min/max launches
    float *d_bbox = nullptr, *h_bbox = nullptr;
    GridParams *d_gp = nullptr, *h_gp = nullptr;
    // arenas (grow only)
    Arena cloud, hits, keep, nfree, keep_off, free_off, frees_raw, frees_ds, xy;
    Arena k0, k1, v0, v1, flag, scan, seg_start, seg_key, cub_tmp, big, chunk_desc;
    Arena scan_status;            // devmap_scan.h: two arrays of per-tile status words, zero between launches
    size_t scan_tiles = 0, scan_dirty[2] = {0, 0};   // status entries in use per array
    uint32_t scan_seq = 0;
    Arena radix_state, radix_tmp; // devmap_sort.h: histograms + two status arrays (zero between sorts); ping-pong buffers
    size_t radix_tiles = 0;
    uint32_t radix_seq = 0;       // sorts so far: which of the two histograms is current
    std::vector<uint8_t> lv_axis_host;   // BGK-LV: staging of the per-axis candidate tables (kept until the next insert)
    bool test_sort = true;        // LA3DM_TEST_SORT=0: the test blocks stay in candidate order (no heaviest-first sort)
    bool own_sort = true;         // LA3DM_OWN_SORT=0: rocPRIM's radix sort instead (A/B)
    bool counters_clean = false;  // the last insert's final launch left the counter block as dm_begin would (dm_commit_prune)
    int spec_bits = 32;           // key digits (x 8 bits) the cloud's own voxel filter needed last time (voxel_grid)
    Arena train, grid, axis_tab, m_code, q_out;
    Arena c_flag, c_weight, c_scan, t_key0, t_key1, t_ent0, t_ent1, t_blockkey, t_center, t_nbr, t_slot, t_slot0;
    Arena nleaf, leaf_off, leaf_key, leaf_alpha, leaf_beta, leaf_state, leaf_node;
    Arena l_ray_idx, l_rays, l_rows, l_rows_off, l_rflag, l_rscan;  // BGKLOctoMap: beam of every sample, beam segments, training rows
    // BGKLVOctoMap (variant 2): beams, samples, segments, gather grid, packed blocks
    Arena lv_rng, lv_flags, lv_seg, lv_nsamp, lv_nray, lv_samp_off, lv_ray_off, lv_samples, lv_rays, lv_sorted, lv_cell_off;
    Arena lv_beam, lv_mask;
    Arena lv_hcell, lv_hcnt, lv_hoff, lv_hlist;
    Arena cell_cnt, slab_range;   // x-slab partition of a sharded insert: per-cell point counts; {first, last} grid cell of the own slab
    int shard_slab = getenv("LA3DM_SHARD_SLAB") ? atoi(getenv("LA3DM_SHARD_SLAB")) : 1;   // x-slab partition of a sharded insert (default); 0 = every rank builds the CSR of all training blocks (A/B)
    int depth3_batch = getenv("LA3DM_DEPTH3") ? atoi(getenv("LA3DM_DEPTH3")) : -1;   // block_depth 3: leaf lists and write-back + prune with 4 / 8 test blocks per wave (devmap_depth3.h); -1 = by the size of the test list, 0 = the general kernels (A/B)
    bool force_slab = getenv("LA3DM_FORCE_SLAB") && atoi(getenv("LA3DM_FORCE_SLAB")) == 1;   // (test / profiling hook: the x-slab form on an UNSHARDED map, its slab = the whole test list)   // ray shortening on the hit grid (devmap_lv_kernels.h, round 6)
    Arena lv_axis, lv_keys, lv_mult, lv_flag, lv_pos, lv_slot, lv_center, lv_cell0, lv_pslot, lv_pmult, lv_info, lv_prune;
    int32_t *d_lvmm = nullptr, *h_lvmm = nullptr;   // bucket bounds of the finite samples (+ their count)
    uint32_t lv_n_samples = 0, lv_n_rays = 0;
    bool lv_original_size = true;
    la3dm_devmap_lv_stats lv_stats;
    // block-sharded insert (la3dm_devmap_set_shard)
    int dbg_fail_rank = -1;       // LA3DM_INJECT_FRONT_END_FAILURE at la3dm_devmap_create (a test hook), else -1
    int dbg_fail_slab_rank = -1;  // LA3DM_INJECT_SLAB_FAILURE: that rank of a sharded map fails in build_slab_csr, i.e. AFTER the cut (test hook)
    int dbg_stuck_at = 0;         // LA3DM_INJECT_SCAN_STUCK = n at la3dm_devmap_create (a test hook): the n-th counter read-back of this
                                  // map reports the look-back loops' "stuck" bit as if a scan launch had found its state dirty
    uint32_t shard_rank = 0, shard_world = 1;
    la3dm_allgatherv_fn shard_fn = nullptr;
    void *shard_user = nullptr;
    Arena shard_w, shard_cumw, shard_bounds;
    uint32_t *h_shard = nullptr;  // pinned: bounds[world + 1] | leaf_bounds[world + 1]
    std::vector<uint64_t> shard_off[4], shard_bytes[4];   // the all-gather-v's segments (alpha, beta, state, leaf key), per rank
    Arena shard_hist, shard_nown, shard_own_off, shard_cnt, shard_frees;   // sharded sample filter (front_end)
    uint32_t shard_status_failed = 0xFFFFFFFFu;   // (the source of the slot's "failed" preset: lives as long as the map)
    std::vector<uint32_t> shard_hist_host, shard_cnt_host;
    uint32_t n_xy = 0;
    bool mailbox = true;      // read_counters through pinned host memory + a sequence number (LA3DM_MAILBOX=0: copy + sync)
    uint32_t mailbox_seq = 0;
    uint32_t scan_resident = kScanResident, radix_resident_wide = 256, radix_resident_big = 256;   // workgroups the chip holds at once (create)
    uint32_t mailbox_pending = 0;   // sequence number a queued kernel will publish itself (0: none — read_counters launches the publisher)
    bool poisoned = false;  // a failed insert whose block table could not be reconciled with the host's block count
    bool stage_timing = false;  // LA3DM_TIMING=1 at creation: extra synchronisations that split t_pack / t_kernel / t_commit
    la3dm_devmap_stats stats;
};

#define DM_TRY(expr) HIP_TRY(dm->ctx, expr)
#define DM_RESERVE(arena, bytes)                                                    \
    do {                                                                            \
        int rc_ = arena_reserve(dm->ctx, (arena), (bytes) ? (size_t)(bytes) : 16);  \
        if (rc_ != LA3DM_OK) return rc_;                                            \
    } while (0)

static int dm_fail(la3dm_devmap *dm, int code, const std::string &msg) {
    dm->ctx->err = msg;
    return code;
}

// Workgroups of a launch that ends in atomics on a handful of addresses (min/max words, arrival counters): a device-scope
// atomic on one address costs ~25 ns per workgroup, serialised — 512 workgroups spent 12 us on them, whatever the data.
constexpr uint32_t kMinmaxWgs = 128;
// dm_beam_write's workgroups: a thread walks n / (256 x workgroups) beams one after the other, each a chain of two dependent loads before its
// samples can be written (kept? -> the hit and its offsets), and every workgroup ends in twelve atomics of the two box reductions (~25 ns each,
// serialised per address).  Kernel trace (profiles/r06/wgs.txt), 128 / 256 / 512 / 1024 workgroups: configs[1]'s 92 k kept hits 25.9 / 30.2 / 35.0 /
// 33.9 us, configs[4]'s 417 k 75.8 / 62.6 / 66.4 / 84.6 us — 256 from 2^18 kept hits up, 128 below (LA3DM_BEAM_WGS: A/B)
static const uint32_t kBeamWgsEnv = getenv("LA3DM_BEAM_WGS") ? (uint32_t)std::max(1, atoi(getenv("LA3DM_BEAM_WGS"))) : 0u;
static inline uint32_t beam_wgs(uint32_t n_h) { return kBeamWgsEnv ? kBeamWgsEnv : (n_h >= (1u << 18) ? 2u * kMinmaxWgs : kMinmaxWgs); }

// ---- library plumbing: device-wide sort / scan (rocPRIM through hipCUB) ---------------------------------
template <size_t kMergeLimit>
static int sort_pairs_cfg(la3dm_devmap *dm, const uint32_t *k_in, uint32_t *k_out, const uint32_t *v_in, uint32_t *v_out,
                          uint32_t n, int end_bit, int begin_bit = 0) {
    hipStream_t st = dm->ctx->stream;
    size_t tmp = 0;
    using sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, kMergeLimit>;
    DM_TRY(rocprim::radix_sort_pairs<sort_config>(nullptr, tmp, k_in, k_out, v_in, v_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, st));
    DM_RESERVE(dm->cub_tmp, tmp);
    DM_TRY(rocprim::radix_sort_pairs<sort_config>(dm->cub_tmp.ptr, tmp, k_in, k_out, v_in, v_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, st));
    return LA3DM_OK;
}

// Stable sort of (key, value) pairs on the low `end_bit` key bits: devmap_sort.h (one histogram launch + one launch per
// 8 key bits; the input arrays are left as they are).  rocPRIM's sort (LA3DM_OWN_SORT=0) is kept for comparison: above
// 2^18 items its Onesweep costs three launches per pass, below that its merge sort log2(n / 1024) launch pairs.
// (sort_fusable: the own sort takes this key range, so a caller may hand its key producer to the histogram launch — sort_pairs_src)
static bool sort_fusable(la3dm_devmap *dm, int end_bit, int begin_bit = 0) {
    return dm->own_sort && end_bit <= 32 && end_bit > begin_bit && begin_bit >= 0;
}
struct SortJob {   // a sort between its histogram launch and its passes
    RadixState rs;
    uint32_t n_pass = 0, n_bound = 0;
    int begin_bit = 0;
};
// histogram launch of a sort of at most n_bound keys (the keys come from `src`: n_items items, devmap_sort.h)
template <class Src>
static int sort_begin_src(la3dm_devmap *dm, const Src &src, uint32_t n_items, uint32_t n_bound, int end_bit, int begin_bit, SortJob &job) {
    if (!sort_fusable(dm, end_bit, begin_bit)) return dm_fail(dm, LA3DM_ERR_ARG, "devmap: internal error: fused histogram on a key range the own sort does not take");
    hipStream_t st = dm->ctx->stream;
    const uint32_t n_pass = ((uint32_t)(end_bit - begin_bit) + 7u) / 8u, tiles = cdiv(n_bound, kRsTile);
    if (tiles > dm->radix_tiles) {
        const size_t want = std::max<size_t>(2 * (size_t)tiles, 1024);
        DM_TRY(hipStreamSynchronize(st));
        const size_t bytes = 2 * 4096 * kRsHistCopies + 64 + 2 * want * 1024 + 2 * (want / kRsGroup + 1) * 1024;
        DM_RESERVE(dm->radix_state, bytes);
        DM_TRY(hipMemsetAsync(dm->radix_state.ptr, 0, bytes, st));   // on the sorts' own stream
        dm->radix_tiles = want;
    }
    RadixState &rs = job.rs;   // layout: two histograms (this sort's, the next sort's), 16 spare words, two status arrays, two group arrays
    uint32_t *base = (uint32_t *)dm->radix_state.ptr;
    rs.hist = base + 1024 * kRsHistCopies * (dm->radix_seq & 1u);
    rs.hist_next = base + 1024 * kRsHistCopies * ((dm->radix_seq + 1u) & 1u);
    ++dm->radix_seq;
    rs.ticket = base + 2048 * kRsHistCopies;   // (the 16 spare words)
    rs.status[0] = base + 2048 * kRsHistCopies + 16;
    rs.status[1] = rs.status[0] + dm->radix_tiles * 256;
    rs.agg[0] = rs.status[1] + dm->radix_tiles * 256;
    rs.agg[1] = rs.agg[0] + (dm->radix_tiles / kRsGroup + 1) * 256;
    job.n_pass = n_pass;
    job.n_bound = n_bound;
    job.begin_bit = begin_bit;
    hipLaunchKernelGGL((dm_radix_hist_src<Src>), dim3(std::max<uint32_t>(std::min<uint32_t>(cdiv(n_items, 4 * kRsThreads), 512u), kRsHistCopies)), dim3(kRsThreads), 0, st, src, n_bound, n_items, n_pass, (uint32_t)begin_bit, rs);
    DM_TRY(hipGetLastError());
    return LA3DM_OK;
}
// the passes; n_dev != nullptr: the key count is read there by the launches (n_bound only sizes them)
static int sort_passes(la3dm_devmap *dm, const SortJob &job, const uint32_t *k_in, uint32_t *k_out, const uint32_t *v_in, uint32_t *v_out,
                       const uint32_t *n_dev = nullptr) {
    hipStream_t st = dm->ctx->stream;
    const uint32_t n = job.n_bound, n_pass = job.n_pass, tiles = cdiv(n, kRsTile);
    uint32_t *tk = nullptr, *tv = nullptr;
    if (n_pass > 1) {
        DM_RESERVE(dm->radix_tmp, 8ull * n);
        tk = (uint32_t *)dm->radix_tmp.ptr;
        tv = tk + n;
    }
    const uint32_t *sk = k_in, *sv = v_in;
    for (uint32_t p = 0; p < n_pass; ++p) {
        const bool to_out = ((n_pass - 1u - p) & 1u) == 0u;   // the last pass lands in the output arrays
        RadixArgs a;
        a.k_in = sk;
        a.v_in = sv;
        a.k_out = to_out ? k_out : tk;
        a.v_out = to_out ? v_out : tv;
        a.n = n;
        a.n_dev = n_dev;
        a.n_pass = n_pass;
        a.pass = p;
        a.begin_bit = (uint32_t)job.begin_bit;
        a.counters = dm->d_cnt;
        a.err_slot = (int)kCntError;
        if (tiles <= dm->radix_resident_wide) {   // every tile on the chip at once, sixteen waves each (devmap_sort.h)
            a.use_ticket = 0u;
            hipLaunchKernelGGL((dm_radix_pass<1024, 4>), dim3(tiles), dim3(1024), 0, st, a, job.rs);
        } else {   // 8192-item tiles, one workgroup per CU: every tile on the chip up to ~2 M items (the free samples' filter), through the ticket beyond
            const uint32_t t2 = cdiv(n, 2 * kRsTile);
            a.use_ticket = t2 > dm->radix_resident_big ? 1u : 0u;
            hipLaunchKernelGGL((dm_radix_pass<1024, 8>), dim3(std::min(t2, dm->radix_resident_big)), dim3(1024), 0, st, a, job.rs);
        }
        sk = a.k_out;
        sv = a.v_out;
    }
    DM_TRY(hipGetLastError());
    return LA3DM_OK;
}
template <class Src>
static int sort_pairs_src(la3dm_devmap *dm, const Src &src, uint32_t n_items, const uint32_t *k_in, uint32_t *k_out, const uint32_t *v_in,
                          uint32_t *v_out, uint32_t n, int end_bit, int begin_bit = 0) {
    if (n == 0) return LA3DM_OK;
    SortJob job;
    int rc = sort_begin_src(dm, src, n_items, n, end_bit, begin_bit, job);
    if (rc != LA3DM_OK) return rc;
    return sort_passes(dm, job, k_in, k_out, v_in, v_out);
}

static int sort_pairs(la3dm_devmap *dm, const uint32_t *k_in, uint32_t *k_out, const uint32_t *v_in, uint32_t *v_out,
                      uint32_t n, int end_bit, int begin_bit = 0) {
    if (n == 0) return LA3DM_OK;
    if (!sort_fusable(dm, end_bit, begin_bit))
        return sort_pairs_cfg<(1u << 18)>(dm, k_in, k_out, v_in, v_out, n, end_bit, begin_bit);
    return sort_pairs_src(dm, RsPlainKeys{k_in}, n, k_in, k_out, v_in, v_out, n, end_bit, begin_bit);
}

// The kernel about to be launched publishes the counter block itself (dm_publish_lane) under a fresh sequence number.
static void publish_with(la3dm_devmap *dm, volatile uint32_t *&mailbox, uint32_t &seq) {
    if (!dm->mailbox) return;   // (copy + synchronise mode: read_counters does it all)
    static const bool off = getenv("LA3DM_PUBLISH_IN_KERNEL") && atoi(getenv("LA3DM_PUBLISH_IN_KERNEL")) == 0;
    if (off) return;
    seq = dm->mailbox_pending = ++dm->mailbox_seq;
    mailbox = (volatile uint32_t *)dm->h_cnt;
}

// state of the single-launch scan (devmap_scan.h) for up to n elements
static int scan_state(la3dm_devmap *dm, uint32_t n, ScanState &ss) {
    const size_t tiles = ((size_t)n + kScanTile - 1) / kScanTile;
    if (tiles > dm->scan_tiles) {
        const size_t want = std::max<size_t>(2 * tiles, 4096);
        DM_TRY(hipStreamSynchronize(dm->ctx->stream));
        DM_RESERVE(dm->scan_status, 16 * want + 16);
        DM_TRY(hipMemsetAsync(dm->scan_status.ptr, 0, 16 * want + 16, dm->ctx->stream));   // on the scans' own stream
        dm->scan_tiles = want;
        dm->scan_dirty[0] = dm->scan_dirty[1] = 0;
    }
    // layout: 16 spare bytes, status array 0, status array 1.  The arrays alternate from launch to launch; a launch clears the
    // used part of the other array (written two launches ago), so the array a launch starts on is all zero.
    unsigned long long *base = (unsigned long long *)((uint8_t *)dm->scan_status.ptr + 16);
    const uint32_t cur = dm->scan_seq & 1u;
    ss.status = base + (size_t)cur * dm->scan_tiles;
    ss.other = base + (size_t)(cur ^ 1u) * dm->scan_tiles;
    ss.other_n = (uint32_t)dm->scan_dirty[cur ^ 1u];
    uint32_t *tickets = (uint32_t *)dm->scan_status.ptr;   // (the 16 spare bytes: one ticket per status array)
    ss.ticket = tiles > dm->scan_resident ? tickets + cur : nullptr;
    ss.ticket_other = tickets + (cur ^ 1u);
    dm->scan_dirty[cur ^ 1u] = 0;
    dm->scan_dirty[cur] = tiles;
    ++dm->scan_seq;
    return LA3DM_OK;
}

// out[i] = in[0] + ... + in[i-1]; total_slot >= 0: the sum of all n also goes to d_cnt[total_slot]
// publish: the launch also sends the counter block to the host's mailbox (the next read_counters only waits for it)
static int exclusive_scan(la3dm_devmap *dm, const uint32_t *in, uint32_t *out, uint32_t n, int total_slot = -1, bool publish = false) {
    if (n == 0) {
        if (total_slot >= 0) DM_TRY(hipMemsetAsync(dm->d_cnt + total_slot, 0, 4, dm->ctx->stream));
        return LA3DM_OK;
    }
    ScanState ss;
    int rc = scan_state(dm, n, ss);
    if (rc != LA3DM_OK) return rc;
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in;
    a.out = out;
    a.n = n;
    a.counters = dm->d_cnt;
    a.total_slot = total_slot;
    a.zero_slot = -1;
    a.err_slot = (int)kCntError;
    if (publish && total_slot >= 0) publish_with(dm, a.mailbox, a.mailbox_seq);
    hipLaunchKernelGGL(dm_scan_lb<false>, dim3(std::min<uint32_t>(cdiv(n, kScanTile), dm->scan_resident)), dim3(kScanThreads), 0, dm->ctx->stream, a, ss);
    DM_TRY(hipGetLastError());
    return LA3DM_OK;
}

// Segments of a sorted key array (invalid keys last), one launch: head flags, their exclusive scan, the segment starts
// (+ keys), d_cnt[seg_slot] = segments, d_cnt[valid_slot] = valid keys, seg_start[segments] = valid keys;
// zero_slot >= 0: d_cnt[zero_slot] = 0 on the way.
static int scan_heads(la3dm_devmap *dm, const uint32_t *keys, uint32_t n, uint32_t *flag, uint32_t *scan, uint32_t *seg_start,
                      uint32_t *seg_key, int seg_slot, int valid_slot, int zero_slot = -1, bool publish = false) {
    ScanState ss;
    int rc = scan_state(dm, n, ss);
    if (rc != LA3DM_OK) return rc;
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.in = keys;
    a.out = scan;
    a.n = n;
    a.counters = dm->d_cnt;
    a.total_slot = -1;
    a.flag = flag;
    a.seg_start = seg_start;
    a.seg_key = seg_key;
    a.seg_slot = seg_slot;
    a.valid_slot = valid_slot;
    a.zero_slot = zero_slot;
    a.err_slot = (int)kCntError;
    if (publish) publish_with(dm, a.mailbox, a.mailbox_seq);
    hipLaunchKernelGGL(dm_scan_lb<true>, dim3(std::min<uint32_t>(cdiv(n, kScanTile), dm->scan_resident)), dim3(kScanThreads), 0, dm->ctx->stream, a, ss);
    DM_TRY(hipGetLastError());
    return LA3DM_OK;
}

static int read_counters(la3dm_devmap *dm) {
    hipStream_t st = dm->ctx->stream;
    if (!dm->mailbox) {
        DM_TRY(hipMemcpyAsync(dm->h_cnt, dm->d_cnt, sizeof(uint32_t) * kCntWords, hipMemcpyDeviceToHost, st));
        DM_TRY(hipStreamSynchronize(st));
        return LA3DM_OK;
    }
    uint32_t seq = dm->mailbox_pending;
    dm->mailbox_pending = 0;
    if (!seq) {
        seq = ++dm->mailbox_seq;
        hipLaunchKernelGGL(dm_publish_counters, dim3(1), dim3(64), 0, st, (const uint32_t *)dm->d_cnt, (volatile uint32_t *)dm->h_cnt, seq);
        DM_TRY(hipGetLastError());
    }
    volatile uint32_t *flag = dm->h_cnt + kCntWords;
    const double t0 = wall();
    for (uint32_t spin = 0; *flag != seq; ++spin) {
        if ((spin & 0x3FFu) == 0x3FFu && wall() - t0 > 2.0) {  // something is wrong (a fault upstream): let HIP report it
            DM_TRY(hipStreamSynchronize(st));
            if (*flag != seq) return dm_fail(dm, LA3DM_ERR_HIP, "devmap: the counter mailbox was not written");
            break;
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);   // the counter words are read (non-volatile) after the flag
    if (dm->dbg_stuck_at > 0 && --dm->dbg_stuck_at == 0) dm->h_cnt[kCntError] |= kScanErrStuck;   // (test hook, tests/test_devmap_gpu.py)
    if (dm->h_cnt[kCntError] & (kScanErrStuck | kRsErrStuck)) {
        dm->poisoned = true;
        if (getenv("LA3DM_DEBUG_CNT")) {
            for (int w = 0; w < (int)kCntWords; ++w) fprintf(stderr, "cnt[%d]=%u ", w, dm->h_cnt[w]);
            fprintf(stderr, "\n");
        }
        return dm_fail(dm, LA3DM_ERR_HIP, "devmap: a prefix-sum launch found its state in use (internal error; flags " + std::to_string(dm->h_cnt[kCntError]) + ")");
    }
    return LA3DM_OK;
}

// after the beam-count kernels: a beam that would not terminate, or more samples than the 32-bit offsets can index
static int check_beam_counters(la3dm_devmap *dm) {
    if (dm->h_cnt[kCntError] & kErrBeam)
        return dm_fail(dm, LA3DM_ERR_ARG, "insert_pointcloud: a beam does not terminate (infinite range, or free_resolution too small "
                                          "against the range for fp32 stepping; set max_range / a larger free_resolution)");
    const uint64_t total = (uint64_t)dm->h_cnt[kCntBeamTotal] | ((uint64_t)dm->h_cnt[kCntBeamTotal + 1] << 32);
    if (total > 0x7FFFFFFFull)
        return dm_fail(dm, LA3DM_ERR_ARG, "insert_pointcloud: more than 2^31 beam samples in one scan (raise free_resolution or set max_range)");
    return LA3DM_OK;
}

// pcl::VoxelGrid centroid filter (bgkoctomap.cpp:419-431) on the device.  d_in: n packed xyz; the result is
// written to `out` (reserved here) and its point count returned.  One read-back (cell count).
// key_bits: number of low key bits that can differ between cells (32 = unknown); must satisfy
// cell count <= 2^key_bits - 1 so that the all-ones key of a non-finite point still sorts behind every cell.
// Round 5: key_bits = 0 — the caller does not know (the cloud's own filter: its grid is reduced on the device by this call): the
// sort runs on as many 8-bit digits as the LAST such call needed (dm->spec_bits; all four the first time), and the
// GridParams that come back with the cell count say whether that was enough; if not (the scene grew past 2^16 / 2^24 cells)
// the filter is simply run again on 32 bits — a constant top digit used to cost a copy pass (10 us) on every insert.
static const uint32_t kBigCellEnv = getenv("LA3DM_BIG_CELL") ? (uint32_t)std::max(8, atoi(getenv("LA3DM_BIG_CELL"))) : 0u;   // (A/B) cells above this many points: one wave each
static const uint32_t kBigCellWgs = getenv("LA3DM_BIG_WGS") ? (uint32_t)atoi(getenv("LA3DM_BIG_WGS")) : 2048u;   // waves (one per cell at a time) of dm_grid_centroids_big
static int voxel_grid(la3dm_devmap *dm, const float *d_in, uint32_t n, float leaf, Arena &out, uint32_t *n_out, int key_bits = 0,
                      bool params_ready = false) {
    hipStream_t st = dm->ctx->stream;
    *n_out = 0;
    if (n == 0) return LA3DM_OK;
    const bool speculate = key_bits == 0;
    if (speculate) key_bits = dm->spec_bits;
    const float inv = 1.0f / leaf;
    DM_RESERVE(dm->k0, 4ull * n);
    DM_RESERVE(dm->k1, 4ull * n);
    DM_RESERVE(dm->v0, 4ull * n);
    DM_RESERVE(dm->v1, 4ull * n);
    DM_RESERVE(dm->flag, 4ull * n);
    DM_RESERVE(dm->scan, 4ull * n);
    DM_RESERVE(dm->seg_start, 4ull * (n + 1));
    uint32_t *k0 = (uint32_t *)dm->k0.ptr, *k1 = (uint32_t *)dm->k1.ptr, *v0 = (uint32_t *)dm->v0.ptr, *v1 = (uint32_t *)dm->v1.ptr;
    uint32_t *flag = (uint32_t *)dm->flag.ptr, *scan = (uint32_t *)dm->scan.ptr, *seg_start = (uint32_t *)dm->seg_start.ptr;
    if (!params_ready) {   // (the producer of d_in may have reduced its box and left the GridParams already)
        MinmaxFin fin = {1, inv, dm->d_gp, nullptr, dm->d_cnt, dm->d_mm + 6, nullptr};
        hipLaunchKernelGGL(dm_minmax<3>, dim3(std::min<uint32_t>(cdiv(n, 1024), kMinmaxWgs)), dim3(256), 0, st, d_in, n, dm->d_mm, fin);
    }
    int rc;
    if (sort_fusable(dm, key_bits)) {   // the cell keys are written by the sort's histogram launch
        const GridCellsSrc src = {d_in, inv, dm->d_gp, k0, v0};
        rc = sort_pairs_src(dm, src, n, k0, k1, v0, v1, n, key_bits);
    } else {
        hipLaunchKernelGGL(dm_grid_cells, dim3(cdiv(n, 256)), dim3(256), 0, st, d_in, n, inv, dm->d_gp, k0, v0);
        rc = sort_pairs(dm, k0, k1, v0, v1, n, key_bits);
    }
    if (rc != LA3DM_OK) return rc;
    if (dm->ctx->opt_grid_order == 1) {
        // Verification mode (la3dm_set_option "grid_order" 1): the order of the points INSIDE a cell — the order of the fp32
        // centroid sums — as pcl::VoxelGrid leaves it: std::sort over {cell index, cloud index} comparing the cell index alone
        // (PCL 1.10 voxel_grid.hpp, call site src/bgkoctomap/bgkoctomap.cpp:419-431), i.e. whatever libstdc++'s introsort does with equal
        // keys.  That is a sequential algorithm: the keys go to the host, the host's own std::sort runs on them exactly as PCL's
        // does, the permutation comes back.  Slow by design (two synchronous copies and a host sort per filter call), never the
        // default; with "fast_trig" 3 the BGK family is then bit-identical to the restatement's set_modes(1, 1) — the arithmetic a
        // ROS Noetic build of the reference most plausibly runs (tests/test_likely_trig_gpu.py).
        std::vector<uint32_t> hk(n);
        DM_TRY(hipMemcpyAsync(hk.data(), k0, 4ull * n, hipMemcpyDeviceToHost, st));
        DM_TRY(hipStreamSynchronize(st));
        std::vector<std::pair<unsigned, unsigned>> iv;
        iv.reserve(n);
        for (uint32_t i = 0; i < n; ++i)
            if (hk[i] != kInvalidCell) iv.emplace_back((unsigned)hk[i], (unsigned)i);
        struct PclLess {
            bool operator()(const std::pair<unsigned, unsigned> &a, const std::pair<unsigned, unsigned> &b) const { return a.first < b.first; }
        };
        std::sort(iv.begin(), iv.end(), PclLess());
        for (size_t j = 0; j < iv.size(); ++j) hk[j] = iv[j].second;
        if (!iv.empty()) DM_TRY(hipMemcpyAsync(v1, hk.data(), 4ull * iv.size(), hipMemcpyHostToDevice, st));
        DM_TRY(hipStreamSynchronize(st));   // (hk is a local)
    }
    // (no head-flag / prefix arrays: their one reader, the chunk descriptors, asks the sorted keys instead — devmap_grid_keys.h;
    //  LA3DM_GRID_KEYS=0 keeps the round-5 form for A/B)
    static const bool kGridKeys = !(getenv("LA3DM_GRID_KEYS") && atoi(getenv("LA3DM_GRID_KEYS")) == 0);
    rc = scan_heads(dm, k1, n, kGridKeys ? nullptr : flag, kGridKeys ? nullptr : scan, seg_start, nullptr, (int)kCntGridSegs, (int)kCntGridValid,
                    (int)kCntBig, true);
    if (rc != LA3DM_OK) return rc;
    // The centroid kernels take the cell count from the counter block and are sized for the worst case (every point its
    // own cell), so they are queued before the host waits for the counters: the wait and the next launches' latency overlap
    // with them.  (A pass-through grid lets them write junk that the copy below replaces.)
    DM_RESERVE(out, 12ull * n);
    // cells with more than kBigCell points hold at least kBigCell + 1 of the n points each
    // (a filter of a few hundred thousand points is bound by its longest one-thread cells, a larger one by its gathers: measured)
    const uint32_t kBigCellMin = kBigCellEnv ? kBigCellEnv : (n <= (1u << 19) ? kBigCell / 2u : kBigCell);
    DM_RESERVE(dm->big, 16ull * (n / kBigCellMin + 1));   // {cell, first, end, -} per entry
    const uint32_t nchunk = n / kChunk;
    DM_RESERVE(dm->chunk_desc, 16ull * (nchunk + 1));
    if (kGridKeys)
        hipLaunchKernelGGL(dm_grid_centroids_keys, dim3(cdiv(n, 256) + cdiv(nchunk, 4)), dim3(256), 0, st, d_in, v1, seg_start, dm->d_cnt,
                           (int)kCntGridSegs, (int)kCntBig, (uint4 *)dm->big.ptr, (float *)out.ptr, cdiv(n, 256), (const uint32_t *)k1,
                           (int)kCntGridValid, nchunk, (uint4 *)dm->chunk_desc.ptr, kBigCellMin);
    else
        hipLaunchKernelGGL(dm_grid_centroids, dim3(cdiv(n, 256) + cdiv(nchunk, 4)), dim3(256), 0, st, d_in, v1, seg_start, dm->d_cnt,
                       (int)kCntGridSegs, (int)kCntBig, (uint4 *)dm->big.ptr, (float *)out.ptr, cdiv(n, 256), (const uint32_t *)flag,
                       (const uint32_t *)scan, (int)kCntGridValid, nchunk, (uint4 *)dm->chunk_desc.ptr, kBigCellMin);
    hipLaunchKernelGGL(dm_grid_centroids_big, dim3(kBigCellWgs), dim3(64), 0, st, d_in, v1, seg_start, dm->d_cnt, (int)kCntBig,
                       (const uint4 *)dm->big.ptr, (const uint4 *)dm->chunk_desc.ptr, (float *)out.ptr);
    rc = read_counters(dm);
    if (rc != LA3DM_OK) return rc;
    memcpy(dm->h_gp, dm->h_cnt + kCntGrid, sizeof(GridParams));
    if (dm->h_gp->passthrough) {  // index space overflows int32: PCL returns the cloud unfiltered
        DM_TRY(hipMemcpyAsync(out.ptr, d_in, 12ull * n, hipMemcpyDeviceToDevice, st));
        *n_out = n;
        return LA3DM_OK;
    }
    if (speculate && !dm->h_gp->empty) {
        // cells <= 2^bits - 1 keeps the all-ones key of a non-finite point behind every cell on `bits` key bits
        const uint64_t cells = (uint64_t)dm->h_gp->m2 * (uint64_t)dm->h_gp->span[2];
        int need = 1;
        while (need < 32 && ((1ull << need) - 1ull) < cells) ++need;
        const int need_digits = std::min(32, 8 * ((need + 7) / 8));
        dm->spec_bits = need_digits;
        if (need_digits > key_bits)   // too few digits were sorted: again, with the GridParams in place
            return voxel_grid(dm, d_in, n, leaf, out, n_out, 32, true);
    }
    *n_out = dm->h_cnt[kCntGridSegs];
    return LA3DM_OK;
}

static int grow_pool(la3dm_devmap *dm, size_t want_blocks) {
    if (want_blocks <= dm->cap_blocks) return LA3DM_OK;
    hipStream_t st = dm->ctx->stream;
    // a regrow copies the pool and synchronises: the first allocation leaves room for a few scans' worth of new blocks
    // (9 bytes per node: 164 k blocks of 73 nodes are 108 MB), later ones double
    // (node indices travel as 32-bit words — leaf_node, the write-back's slot * nodes-per-block: a pool ends at 2^32 nodes)
    const size_t cap_max = 0xFFFFFFFFull / dm->npb;
    if (want_blocks > cap_max) return dm_fail(dm, LA3DM_ERR_OOM, "devmap: the block pool would exceed 2^32 nodes");
    size_t cap = std::min(std::max<size_t>((dm->cap_blocks ? 2 : 4) * want_blocks, 4096), cap_max);
    float *A = nullptr, *B = nullptr;
    uint8_t *S = nullptr;
    long long *bk = nullptr;
    const size_t nn = cap * dm->npb;
    if (hipMalloc((void **)&A, 4 * nn) != hipSuccess || hipMalloc((void **)&B, 4 * nn) != hipSuccess ||
        hipMalloc((void **)&S, nn) != hipSuccess || hipMalloc((void **)&bk, 8 * cap) != hipSuccess) {
        if (A) (void)hipFree(A);
        if (B) (void)hipFree(B);
        if (S) (void)hipFree(S);
        if (bk) (void)hipFree(bk);
        return dm_fail(dm, LA3DM_ERR_OOM, "devmap: block pool allocation failed");
    }
    if (dm->n_blocks) {
        const size_t on = (size_t)dm->n_blocks * dm->npb;
        DM_TRY(hipMemcpyAsync(A, dm->A, 4 * on, hipMemcpyDeviceToDevice, st));
        DM_TRY(hipMemcpyAsync(B, dm->B, 4 * on, hipMemcpyDeviceToDevice, st));
        DM_TRY(hipMemcpyAsync(S, dm->S, on, hipMemcpyDeviceToDevice, st));
        DM_TRY(hipMemcpyAsync(bk, dm->blk_key, 8ull * dm->n_blocks, hipMemcpyDeviceToDevice, st));
        DM_TRY(hipStreamSynchronize(st));
    }
    if (dm->A) (void)hipFree(dm->A);
    if (dm->B) (void)hipFree(dm->B);
    if (dm->S) (void)hipFree(dm->S);
    if (dm->blk_key) (void)hipFree(dm->blk_key);
    dm->A = A;
    dm->B = B;
    dm->S = S;
    dm->blk_key = bk;
    dm->cap_blocks = cap;
    return LA3DM_OK;
}

static int grow_table(la3dm_devmap *dm, size_t want_blocks) {
    if (dm->tab_cap >= 2 * want_blocks && dm->tab_cap) return LA3DM_OK;
    hipStream_t st = dm->ctx->stream;
    uint32_t cap = 1u << 16;
    while (cap < 4 * want_blocks) cap <<= 1;
    long long *tk = nullptr;
    uint32_t *tv = nullptr;
    if (hipMalloc((void **)&tk, 8ull * cap) != hipSuccess || hipMalloc((void **)&tv, 4ull * cap) != hipSuccess) {
        if (tk) (void)hipFree(tk);
        return dm_fail(dm, LA3DM_ERR_OOM, "devmap: block table allocation failed");
    }
    DM_TRY(hipMemsetAsync(tk, 0xFF, 8ull * cap, st));
    if (dm->n_blocks)
        hipLaunchKernelGGL(dm_table_rebuild, dim3(cdiv(dm->n_blocks, 256)), dim3(256), 0, st, dm->blk_key, dm->n_blocks, tk, tv,
                           cap - 1);
    DM_TRY(hipStreamSynchronize(st));
    if (dm->tab_key) (void)hipFree(dm->tab_key);
    if (dm->tab_val) (void)hipFree(dm->tab_val);
    dm->tab_key = tk;
    dm->tab_val = tv;
    dm->tab_cap = cap;
    return LA3DM_OK;
}

// float-stepped candidate indices of one axis (get_blocks_in_bbox, bgkoctomap.cpp:486-495): the three nested
// loops restart the inner sequences identically, so the candidate list is the product of three sequences.
static std::vector<int> axis_sequence(float lo, float hi, float bs) {
    std::vector<int> seq;
    for (float v = lo - bs; v <= hi + 2 * bs; v += bs) {
        seq.push_back((int)(int64_t)((double)v / (double)bs + 524288.5));
        if (seq.size() > (1u << 20)) break;
    }
    return seq;
}

extern "C" {

int la3dm_devmap_create(la3dm_ctx *ctx, la3dm_devmap **out) {
    if (!ctx || !out) return LA3DM_ERR_ARG;
    *out = nullptr;
    if (ctx->p.variant < 0 || ctx->p.variant > 3) {
        ctx->err = "la3dm_devmap_create: the device-resident map supports variant 0 (BGK), 1 (GP), 2 (BGK-LV) and 3 (BGK-L)";
        return LA3DM_ERR_ARG;
    }
    if (ctx->p.block_depth > 5) {  // dm_prune stages 3 bytes per node of a block in LDS
        ctx->err = "la3dm_devmap_create: the device-resident map supports block_depth <= 5";
        return LA3DM_ERR_ARG;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    la3dm_devmap *dm = new la3dm_devmap;
    dm->ctx = ctx;
    if (const char *ev = getenv("LA3DM_INJECT_FRONT_END_FAILURE")) dm->dbg_fail_rank = atoi(ev);   // test hooks, read once
    if (const char *ev = getenv("LA3DM_INJECT_SLAB_FAILURE")) dm->dbg_fail_slab_rank = atoi(ev);
    if (const char *ev = getenv("LA3DM_INJECT_SCAN_STUCK")) dm->dbg_stuck_at = atoi(ev);
    ctx->n_devmaps++;   // (la3dm_devmap_destroy, also the failure paths' clean-up, counts it down)
    dm->depth = (uint32_t)ctx->p.block_depth;
    dm->npb = npb_of(ctx->p.block_depth);
    dm->ncell = 1u << (3 * (dm->depth - 1));
    dm->block_size = (float)pow(2, ctx->p.block_depth - 1) * ctx->p.resolution;  // bgkoctomap.cpp:36
    if (ctx->p.variant == 1) {  // GP nodes start at (0, min_ivar), gpoctree_node.cpp:15-17
        dm->init_A = 0.0f;
        dm->init_B = ctx->p.min_ivar;
    } else {
        dm->init_A = ctx->p.prior_A;
        dm->init_B = ctx->p.prior_B;
    }
    memset(&dm->stats, 0, sizeof(dm->stats));
    dm->stage_timing = getenv("LA3DM_TIMING") != nullptr;
    {
        const char *mb = getenv("LA3DM_MAILBOX");
        dm->mailbox = !(mb && mb[0] == '0');
        const char *os = getenv("LA3DM_OWN_SORT");
        dm->own_sort = !(os && os[0] == '0');
        const char *ts = getenv("LA3DM_TEST_SORT");
        dm->test_sort = !(ts && ts[0] == '0');
    }
    bool ok = hipMalloc((void **)&dm->d_cnt, sizeof(uint32_t) * kCntWords) == hipSuccess &&
              // the mailbox: coherent (the in-kernel publish must be visible while the kernel still runs, whatever
              // HIP_HOST_COHERENT says), mapped, and cleared — the first wait is for sequence number 1
              hipHostMalloc((void **)&dm->h_cnt, sizeof(uint32_t) * (kCntWords + 2), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess &&
              (memset(dm->h_cnt, 0, sizeof(uint32_t) * (kCntWords + 2)), true) &&
              hipMalloc((void **)&dm->d_mm, sizeof(uint32_t) * kMmWords) == hipSuccess &&
              hipMalloc((void **)&dm->d_bbox, sizeof(float) * 8) == hipSuccess &&
              hipHostMalloc((void **)&dm->h_bbox, sizeof(float) * 8) == hipSuccess &&
              hipMalloc((void **)&dm->d_gp, sizeof(GridParams)) == hipSuccess &&
              hipHostMalloc((void **)&dm->h_gp, sizeof(GridParams)) == hipSuccess &&
              hipMemset(dm->d_cnt, 0, sizeof(uint32_t) * kCntWords) == hipSuccess &&
              (ctx->p.variant != 2 || (hipMalloc((void **)&dm->d_lvmm, 32) == hipSuccess && hipHostMalloc((void **)&dm->h_lvmm, 32) == hipSuccess));
    if (!ok) {
        ctx->err = "la3dm_devmap_create: allocation failed";
        la3dm_devmap_destroy(dm);
        return LA3DM_ERR_OOM;
    }
    {
        // The single-launch scan and the radix passes hand tiles to at most as many workgroups as the chip holds at once
        // (a workgroup waits for tiles of workgroups that started before it): the bound comes from the occupancy of the
        // kernels as compiled, not from a constant that a change of their register count would silently falsify.
        int cus = 0, b0 = 0, b1 = 0, b3 = 0, b4 = 0;   // (hipGetDeviceProperties would cost tens of ms here)
        ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0 &&
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&b0, dm_scan_lb<false>, (int)kScanThreads, 0) == hipSuccess &&
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&b1, dm_scan_lb<true>, (int)kScanThreads, 0) == hipSuccess &&
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&b3, dm_radix_pass<1024, 4>, 1024, 0) == hipSuccess &&
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4, dm_radix_pass<1024, 8>, 1024, 0) == hipSuccess && b0 > 0 && b1 > 0 && b3 > 0 && b4 > 0;
        if (!ok) {
            ctx->err = "la3dm_devmap_create: occupancy query failed";
            la3dm_devmap_destroy(dm);
            return LA3DM_ERR_HIP;
        }
        dm->scan_resident = std::min<uint32_t>(kScanResident, (uint32_t)std::min(b0, b1) * (uint32_t)cus);
        dm->radix_resident_big = (uint32_t)cus;
        dm->radix_resident_wide = (uint32_t)cus;   // (one sixteen-wave workgroup per CU, whatever the occupancy allows: the shape is for tile latency)
        // test hook: a handful of workgroups per launch forces the multi-round (ticket) form of the scan / sort on small inputs
        if (const char *ev = getenv("LA3DM_SCAN_RESIDENT")) dm->scan_resident = std::max(1, std::min<int>(atoi(ev), (int)dm->scan_resident));
        if (const char *ev = getenv("LA3DM_RADIX_RESIDENT")) {
            const uint32_t cap = (uint32_t)std::max(1, atoi(ev));
            dm->radix_resident_wide = std::min(dm->radix_resident_wide, cap);
            dm->radix_resident_big = std::min(dm->radix_resident_big, cap);
        }
        if (const char *ev = getenv("LA3DM_RADIX_WIDE")) dm->radix_resident_wide = (uint32_t)std::max(0, atoi(ev));   // (A/B: 0 = 8192-item tiles only)
    }
    *out = dm;
    return LA3DM_OK;
}

void la3dm_devmap_destroy(la3dm_devmap *dm) {
    if (!dm) return;
    dm->ctx->n_devmaps--;
    (void)hipSetDevice(dm->ctx->device);
    Arena *all[] = {&dm->cloud, &dm->hits, &dm->keep, &dm->nfree, &dm->keep_off, &dm->free_off, &dm->frees_raw, &dm->frees_ds,
                    &dm->xy, &dm->k0, &dm->k1, &dm->v0, &dm->v1, &dm->flag, &dm->scan, &dm->seg_start, &dm->seg_key,
                    &dm->cub_tmp, &dm->scan_status, &dm->radix_state, &dm->radix_tmp, &dm->big, &dm->chunk_desc, &dm->train, &dm->grid, &dm->axis_tab, &dm->m_code, &dm->q_out, &dm->c_flag, &dm->c_weight, &dm->c_scan, &dm->t_key0,
                    &dm->t_key1, &dm->t_ent0, &dm->t_ent1, &dm->t_blockkey, &dm->t_center, &dm->t_nbr, &dm->t_slot, &dm->t_slot0, &dm->nleaf,
                    &dm->leaf_off, &dm->leaf_key, &dm->leaf_alpha, &dm->leaf_beta, &dm->leaf_state, &dm->leaf_node,
                    &dm->l_ray_idx, &dm->l_rays, &dm->l_rows, &dm->l_rows_off, &dm->l_rflag, &dm->l_rscan,
                    &dm->shard_w, &dm->shard_cumw, &dm->shard_bounds, &dm->shard_hist, &dm->shard_nown, &dm->shard_own_off, &dm->shard_cnt, &dm->shard_frees,
                    &dm->lv_rng, &dm->lv_flags, &dm->lv_seg, &dm->lv_nsamp, &dm->lv_nray, &dm->lv_samp_off, &dm->lv_ray_off, &dm->lv_samples,
                    &dm->lv_rays, &dm->lv_sorted, &dm->lv_cell_off, &dm->lv_axis, &dm->lv_keys, &dm->lv_mult, &dm->lv_flag, &dm->lv_pos,
                    &dm->lv_slot, &dm->lv_center, &dm->lv_cell0, &dm->lv_pslot, &dm->lv_pmult, &dm->lv_info, &dm->lv_prune, &dm->lv_beam, &dm->lv_mask};
    for (Arena *a : all)
        if (a->ptr) (void)hipFree(a->ptr);
    if (dm->h_shard) (void)hipHostFree(dm->h_shard);
    if (dm->d_lvmm) (void)hipFree(dm->d_lvmm);
    if (dm->h_lvmm) (void)hipHostFree(dm->h_lvmm);
    void *dev[] = {dm->A, dm->B, dm->S, dm->blk_key, dm->tab_key, dm->tab_val, dm->d_cnt, dm->d_mm, dm->d_bbox, dm->d_gp};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    if (dm->h_cnt) (void)hipHostFree(dm->h_cnt);
    if (dm->h_bbox) (void)hipHostFree(dm->h_bbox);
    if (dm->h_gp) (void)hipHostFree(dm->h_gp);
    delete dm;
}

// bbox of the training set in dm->xy (bbox(), bgkoctomap.cpp:464-484) -> dm->h_bbox; get_blocks_in_bbox walks it on
// the host.  One synchronisation.
static int training_bbox(la3dm_devmap *dm, bool reduced = false) {
    hipStream_t st = dm->ctx->stream;
    const uint32_t npts = dm->n_xy;
    if (!reduced) {   // (the front end's own kernels may have reduced the box while writing the set)
        MinmaxFin fin = {2, 0.0f, nullptr, (const float *)dm->xy.ptr, dm->d_cnt, dm->d_mm + 6, nullptr};
        publish_with(dm, fin.mailbox, fin.mailbox_seq);
        hipLaunchKernelGGL(dm_minmax<4>, dim3(std::min<uint32_t>(cdiv(npts, 1024), kMinmaxWgs)), dim3(256), 0, st, (const float *)dm->xy.ptr,
                           npts, dm->d_mm, fin);
    }
    int rc = read_counters(dm);   // the box comes back with the counter block
    if (rc != LA3DM_OK) return rc;
    memcpy(dm->h_bbox, dm->h_cnt + kCntBbox, sizeof(float) * 6);
    return LA3DM_OK;
}

// Sharded sample filter: cut[q] = first layer of rank q (q = 0..world), contiguous ranges whose sample counts cross
// q / world of the total (mirrored by la3dm_amd/sharding.py layer_cuts for the CPU test).
static void shard_layer_cuts(const uint32_t *hist, uint32_t nlayer, uint32_t world, std::vector<uint32_t> &cut) {
    uint64_t total = 0;
    for (uint32_t j = 0; j < nlayer; ++j) total += hist[j];
    cut.assign(world + 1, nlayer);
    cut[0] = 0;
    uint64_t run = 0;
    uint32_t q = 1;
    for (uint32_t j = 0; j < nlayer && q < world; ++j) {
        // layer j starts at running count `run`: it opens rank q's range when the count has reached q / world of the total
        while (q < world && run * world >= total * q) cut[q++] = j;
        run += hist[j];
    }
    while (q < world) cut[q++] = nlayer;
}

// What the rank-local part of the front end leaves for its tail (and, in a sharded map, for the exchange in between).
struct FrontState {
    enum { kDone, kFilteredLocally, kShardedFilter } path = kDone;  // kDone: finished (no training set, or BGK-L's own path)
    uint32_t n_kept = 0, n_f = 0, n_f_own = 0;
    const float *d_frees = nullptr;
    float4 *xy = nullptr;
    uint32_t *mm_hits = nullptr;
};
// status word of a rank in the sharded front end's count exchange (values above any sample count)
constexpr uint32_t kShardStatusFailed = 0xFFFFFFFFu;    // a rank-local error: every rank gives the insert up
constexpr uint32_t kShardStatusNoFilter = 0xFFFFFFFEu;  // this insert does not shard its sample filter (decided alike everywhere)

// f1 (bgkoctomap.cpp:383-458): voxel grid over the hits, range gate + beam samples, voxel grid over the free samples;
// leaves the labelled training set in dm->xy (hits first), its size in dm->n_xy, and the scan's bbox in dm->h_bbox.
// front_end_local: everything a rank does on its own, up to (sharded map) its share of the filtered samples.
static int front_end_local(la3dm_devmap *dm, const float *d_xyz, uint32_t n, const float origin[3], float ds_resolution,
                           float free_resolution, float max_range, FrontState &F) {
    la3dm_ctx *ctx = dm->ctx;
    hipStream_t st = ctx->stream;
    la3dm_devmap_stats &S = dm->stats;
    int rc;
    // test hook (tests/test_sharded_insert_gpu.py): rank dbg_fail_rank of a sharded map fails here.  The environment variable is
    // read ONCE, when the map is created (not on every insert of the production path)
    if (dm->dbg_fail_rank >= 0 && dm->shard_world > 1 && dm->dbg_fail_rank == (int)dm->shard_rank)
        return dm_fail(dm, LA3DM_ERR_OOM, "devmap: injected rank-local front-end failure (LA3DM_INJECT_FRONT_END_FAILURE)");
    const float *d_hits = d_xyz;
    uint32_t n_h = n;
    int free_key_bits = 32;
    if (!(ds_resolution < 0)) {
        if ((rc = voxel_grid(dm, d_xyz, n, ds_resolution, dm->hits, &n_h)) != LA3DM_OK) return rc;
        d_hits = (const float *)dm->hits.ptr;
        // The free samples lie between the origin and a (downsampled) hit, i.e. inside the box of the raw hits and
        // the origin, give or take an ulp: their grid has at most (that box + one cell on every side) cells, which
        // bounds the significant bits of their sort keys (one Onesweep pass per 8 bits).
        const GridParams &g = *dm->h_gp;
        if (!g.passthrough && !g.empty) {
            const float inv = 1.0f / ds_resolution;
            double cells = 1.0;
            for (int a = 0; a < 3; ++a) {
                const double oc = floor((double)origin[a] * (double)inv);
                const double lo = std::min((double)g.lo[a], oc) - 1.0, hi = std::max((double)g.lo[a] + g.span[a] - 1.0, oc) + 1.0;
                cells *= hi - lo + 1.0;
            }
            int b = 1;
            while (b < 32 && (double)((1ull << b) - 1ull) < cells) ++b;
            free_key_bits = b;
        }
    }
    if (n_h == 0) return LA3DM_OK;
    DM_RESERVE(dm->keep, 4ull * n_h);
    DM_RESERVE(dm->nfree, 4ull * n_h);
    DM_RESERVE(dm->keep_off, 4ull * n_h);
    DM_RESERVE(dm->free_off, 4ull * n_h);
    BeamArgs ba = {origin[0], origin[1], origin[2], free_resolution, max_range};
    uint32_t *keep = (uint32_t *)dm->keep.ptr, *nfree = (uint32_t *)dm->nfree.ptr, *keep_off = (uint32_t *)dm->keep_off.ptr,
             *free_off = (uint32_t *)dm->free_off.ptr;
    if (ctx->p.variant == 3) {  // BGKLOctoMap: samples keep their beam, no second voxel filter
        hipLaunchKernelGGL(dm_l_beam_count, dim3(std::min<uint32_t>(cdiv(n_h, 256), kBeamCountWgs)), dim3(256), 0, st, d_hits, n_h, ba, keep, nfree, dm->d_cnt);
        if ((rc = exclusive_scan(dm, keep, keep_off, n_h, (int)kCntKept)) != LA3DM_OK) return rc;
        if ((rc = exclusive_scan(dm, nfree, free_off, n_h, (int)kCntFreeRaw, true)) != LA3DM_OK) return rc;
        if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
        if ((rc = check_beam_counters(dm)) != LA3DM_OK) return rc;
        const uint32_t n_beams = dm->h_cnt[kCntKept], n_samples = dm->h_cnt[kCntFreeRaw];
        if (n_beams == 0) return LA3DM_OK;
        DM_RESERVE(dm->xy, 16ull * n_samples);
        DM_RESERVE(dm->l_ray_idx, 4ull * n_samples);
        DM_RESERVE(dm->l_rays, 24ull * n_beams);
        hipLaunchKernelGGL(dm_l_beam_write, dim3(cdiv(n_h, 256)), dim3(256), 0, st, d_hits, n_h, ba, keep, keep_off, free_off,
                           (float4 *)dm->xy.ptr, (int32_t *)dm->l_ray_idx.ptr, (float *)dm->l_rays.ptr);
        dm->n_xy = n_samples;
        S.n_hits = n_beams;
        S.n_frees = n_samples - n_beams;
        return training_bbox(dm);
    }
    hipLaunchKernelGGL(dm_beam_count, dim3(std::min<uint32_t>(cdiv(n_h, 256), kBeamCountWgs)), dim3(256), 0, st, d_hits, n_h, ba, keep, nfree, dm->d_cnt);
    if ((rc = exclusive_scan(dm, keep, keep_off, n_h, (int)kCntKept)) != LA3DM_OK) return rc;
    if ((rc = exclusive_scan(dm, nfree, free_off, n_h, (int)kCntFreeRaw, true)) != LA3DM_OK) return rc;
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    if ((rc = check_beam_counters(dm)) != LA3DM_OK) return rc;
    const uint32_t n_kept = dm->h_cnt[kCntKept], n_free_raw = dm->h_cnt[kCntFreeRaw];
    if (n_kept == 0) return LA3DM_OK;
    DM_RESERVE(dm->xy, 16ull * ((size_t)n_kept + n_free_raw));
    float4 *xy = (float4 *)dm->xy.ptr;
    // the beam kernel also reduces the box of the free samples (-> GridParams of the second voxel filter, finished by its
    // last workgroup) and the box of the kept hits (d_mm[8..13]); dm_append_frees then completes the training set's box
    uint32_t *mm_hits = dm->d_mm + 8;
    const float *d_frees = nullptr;
    uint32_t n_f = 0;
    // Block-sharded insert: the samples' voxel filter is divided over the ranks by z-layer of its grid (devmap_kernels.h,
    // "sharded sample filter"); every rank ends up with the same filtered list, in the single-GPU order.
    const GridParams &g1 = *dm->h_gp;   // (grid of the cloud's own filter: the box of the raw hits)
    if (dm->shard_world > 1 && ctx->opt_grid_order == 1 && !(ds_resolution < 0))
        return dm_fail(dm, LA3DM_ERR_ARG, "devmap: option grid_order 1 (pcl::VoxelGrid's own sort, a verification mode) runs on one GPU only");
    bool sharded_filter = dm->shard_world > 1 && !(ds_resolution < 0) && !g1.passthrough && !g1.empty;
    int zbase = 0;
    uint32_t nlayer = 0;
    if (sharded_filter) {
        // every sample lies between the origin and a filtered hit (a centroid of raw hits): its layer is inside
        // [min(origin, box) - 1, max(origin, box) + 1]
        const float inv = 1.0f / ds_resolution;
        const double oz = floor((double)origin[2] * (double)inv);
        const double lo = std::min((double)g1.lo[2], oz) - 1.0, hi = std::max((double)g1.lo[2] + g1.span[2] - 1.0, oz) + 1.0;
        if (hi - lo + 1.0 > (double)kShardLayers || fabs(lo) > 1e9 || fabs(hi) > 1e9) sharded_filter = false;   // (a very tall scan: unsharded filter)
        else {
            zbase = (int)lo;
            nlayer = (uint32_t)(hi - lo + 1.0);
        }
    }
    if (!sharded_filter) {
        DM_RESERVE(dm->frees_raw, 12ull * n_free_raw);
        {
            MinmaxFin fin = {1, ds_resolution < 0 ? 1.0f : 1.0f / ds_resolution, dm->d_gp, nullptr, dm->d_cnt, dm->d_mm + 6, nullptr};
            hipLaunchKernelGGL(dm_beam_write<false>, dim3(std::min<uint32_t>(cdiv(n_h, 256), beam_wgs(n_h))), dim3(256), 0, st, d_hits, n_h, ba, keep,
                               keep_off, free_off, xy, (float *)dm->frees_raw.ptr, dm->d_mm, fin, mm_hits, 0.0f, 0, 0);
        }
        d_frees = (const float *)dm->frees_raw.ptr;
        n_f = n_free_raw;
        if (!(ds_resolution < 0)) {
            if ((rc = voxel_grid(dm, d_frees, n_free_raw, ds_resolution, dm->frees_ds, &n_f, free_key_bits, true)) != LA3DM_OK) return rc;
            d_frees = (const float *)dm->frees_ds.ptr;
        }
    } else {
        const float inv = 1.0f / ds_resolution;
        const uint32_t world = dm->shard_world, rank = dm->shard_rank;
        // 1. samples per layer -> contiguous layer ranges of equal sample count
        DM_RESERVE(dm->shard_hist, 4ull * kShardLayers);
        DM_TRY(hipMemsetAsync(dm->shard_hist.ptr, 0, 4ull * nlayer, st));
        hipLaunchKernelGGL(dm_beam_hist, dim3(std::min<uint32_t>(cdiv(n_h, 256), kMinmaxWgs)), dim3(256), 0, st, d_hits, n_h, ba, keep, inv, zbase,
                           nlayer, (uint32_t *)dm->shard_hist.ptr, dm->d_cnt);
        dm->shard_hist_host.resize(nlayer);
        DM_TRY(hipMemcpyAsync(dm->shard_hist_host.data(), dm->shard_hist.ptr, 4ull * nlayer, hipMemcpyDeviceToHost, st));
        DM_TRY(hipStreamSynchronize(st));
        std::vector<uint32_t> cut;
        shard_layer_cuts(dm->shard_hist_host.data(), nlayer, world, cut);   // [world + 1] layer indices relative to zbase
        const int lo = zbase + (int)cut[rank], hi = zbase + (int)cut[rank + 1];
        if (getenv("LA3DM_DEBUG_SHARD"))
            fprintf(stderr, "la3dm sharded sample filter: rank %u of %u takes layers [%d, %d) of [%d, %d), %u raw samples in all\n", rank, world,
                    lo, hi, zbase, zbase + (int)nlayer, n_free_raw);
        // 2. own samples per beam -> offsets, total
        DM_RESERVE(dm->shard_nown, 4ull * n_h);
        DM_RESERVE(dm->shard_own_off, 4ull * n_h);
        uint32_t *nown = (uint32_t *)dm->shard_nown.ptr, *own_off = (uint32_t *)dm->shard_own_off.ptr;
        hipLaunchKernelGGL(dm_beam_count_own, dim3(cdiv(n_h, 256)), dim3(256), 0, st, d_hits, n_h, ba, keep, inv, lo, hi, nown);
        if ((rc = exclusive_scan(dm, nown, own_off, n_h, (int)kCntFreeRaw, true)) != LA3DM_OK) return rc;
        if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
        const uint32_t n_own = dm->h_cnt[kCntFreeRaw];
        // 3. kept hits -> xy, own samples -> frees_raw, the box of ALL samples -> the filter grid's (global) parameters
        DM_RESERVE(dm->frees_raw, 12ull * (n_own ? n_own : 1));
        DM_RESERVE(dm->shard_frees, 12ull * (n_free_raw ? n_free_raw : 1));  // the gathered list (<= every raw sample): sized HERE, so that a
                                                                             // rank that cannot hold it fails before the exchange, not inside it
        {
            MinmaxFin fin = {1, inv, dm->d_gp, nullptr, dm->d_cnt, dm->d_mm + 6, nullptr};
            hipLaunchKernelGGL(dm_beam_write<true>, dim3(std::min<uint32_t>(cdiv(n_h, 256), beam_wgs(n_h))), dim3(256), 0, st, d_hits, n_h, ba, keep,
                               keep_off, own_off, xy, (float *)dm->frees_raw.ptr, dm->d_mm, fin, mm_hits, inv, lo, hi);
        }
        // 4. this rank's cells.  (The grid's parameters come back with the filter's counters; a rank without samples fetches
        //    them on its own: whether the grid overflows int32 — PCL then returns its input unfiltered, which the layer
        //    split cannot reproduce — must be decided identically everywhere BEFORE the first collective.)
        uint32_t n_f_own = 0;
        if (n_own) {
            if ((rc = voxel_grid(dm, (const float *)dm->frees_raw.ptr, n_own, ds_resolution, dm->frees_ds, &n_f_own, free_key_bits, true)) != LA3DM_OK)
                return rc;
        } else {
            if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
            memcpy(dm->h_gp, dm->h_cnt + kCntGrid, sizeof(GridParams));
        }
        if (dm->h_gp->passthrough)
            return dm_fail(dm, LA3DM_ERR_ARG, "devmap: sharded insert: the samples' filter grid overflows int32 (PCL passes the cloud through); use one GPU");
        F.path = FrontState::kShardedFilter;
        F.n_kept = n_kept;
        F.n_f_own = n_f_own;
        F.xy = xy;
        F.mm_hits = mm_hits;
        return LA3DM_OK;
    }
    F.path = FrontState::kFilteredLocally;
    F.n_kept = n_kept;
    F.n_f = n_f;
    F.d_frees = d_frees;
    F.xy = xy;
    F.mm_hits = mm_hits;
    return LA3DM_OK;
}

static int front_end(la3dm_devmap *dm, const float *d_xyz, uint32_t n, const float origin[3], float ds_resolution,
                     float free_resolution, float max_range) {
    la3dm_ctx *ctx = dm->ctx;
    hipStream_t st = ctx->stream;
    la3dm_devmap_stats &S = dm->stats;
    FrontState F;
    // sharded map: this rank's slot of the status exchange reads "failed" from here until the local part has succeeded
    if (dm->shard_world > 1 && dm->shard_fn)
        (void)hipMemcpyAsync((uint32_t *)dm->shard_cnt.ptr + dm->shard_rank, &dm->shard_status_failed, 4, hipMemcpyHostToDevice, st);
    const int lrc = front_end_local(dm, d_xyz, n, origin, ds_resolution, free_resolution, max_range, F);
    if (dm->shard_world > 1 && dm->shard_fn) {
        // Sharded map: ONE status / count exchange per insert, entered by every rank whatever happened to it — a rank
        // that failed on its own (out of memory, a scan that tripped, ...) posts kShardStatusFailed and all ranks give
        // the insert up together instead of its peers waiting in a collective it never enters; an insert
        // that does not shard its filter (no voxel grid, a grid that overflows, a very tall scan, nothing to train on)
        // posts kShardStatusNoFilter.  4 bytes per rank.
        const uint32_t world = dm->shard_world, rank = dm->shard_rank;
        uint32_t status = lrc != LA3DM_OK ? kShardStatusFailed : F.path == FrontState::kShardedFilter ? F.n_f_own : kShardStatusNoFilter;
        const std::string local_err = ctx->err;
        // the exchange buffer exists since la3dm_devmap_set_shard, and this rank's slot reads "failed" (preset above, before the
        // local part) until the copy below has replaced it: a rank that cannot even stage its word is seen as failed by its peers instead of handing them a stale
        // count of the insert before
        int xrc = LA3DM_OK;
        if (hipMemcpyAsync((uint32_t *)dm->shard_cnt.ptr + rank, &status, 4, hipMemcpyHostToDevice, st) != hipSuccess) xrc = LA3DM_ERR_HIP;
        dm->shard_off[0].resize(world);
        dm->shard_bytes[0].resize(world);
        for (uint32_t q = 0; q < world; ++q) {
            dm->shard_off[0][q] = 4ull * q;
            dm->shard_bytes[0][q] = 4;
        }
        la3dm_gather_seg seg = {dm->shard_cnt.ptr, dm->shard_off[0].data(), dm->shard_bytes[0].data()};
        const int cbrc = dm->shard_fn(dm->shard_user, &seg, 1, world, rank, (void *)st);
        if (cbrc != 0 && lrc == LA3DM_OK)   // (a rank that had failed on its own keeps its own error)
            return dm_fail(dm, LA3DM_ERR_ARG, "devmap: the all-gather callback of the sharded insert failed (sample counts)");
        if (lrc != LA3DM_OK) {
            ctx->err = local_err;
            return lrc;
        }
        if (xrc != LA3DM_OK) return dm_fail(dm, xrc, "devmap: sharded insert: could not stage the status word");
        dm->shard_cnt_host.resize(world);
        DM_TRY(hipMemcpyAsync(dm->shard_cnt_host.data(), dm->shard_cnt.ptr, 4ull * world, hipMemcpyDeviceToHost, st));
        DM_TRY(hipStreamSynchronize(st));
        for (uint32_t q = 0; q < world; ++q)
            if (dm->shard_cnt_host[q] == kShardStatusFailed)
                return dm_fail(dm, LA3DM_ERR_PEER, "devmap: sharded insert: rank " + std::to_string(q) + " failed in its front end; the insert is given up on every rank");
        for (uint32_t q = 0; q < world; ++q)
            if ((dm->shard_cnt_host[q] == kShardStatusNoFilter) != (status == kShardStatusNoFilter))
                return dm_fail(dm, LA3DM_ERR_HIP, "devmap: sharded insert: the ranks disagree on whether this insert shards its sample filter");
    } else if (lrc != LA3DM_OK) {
        return lrc;
    }
    if (F.path == FrontState::kDone) return LA3DM_OK;
    const uint32_t n_kept = F.n_kept;
    float4 *xy = F.xy;
    uint32_t *mm_hits = F.mm_hits;
    const float *d_frees = F.d_frees;
    uint32_t n_f = F.n_f;
    if (F.path == FrontState::kShardedFilter) {
        // all-gather-v of the filtered points behind the count exchange above: every rank gets the whole list, rank
        // order = ascending cell index = the single-GPU order
        const uint32_t world = dm->shard_world, rank = dm->shard_rank, n_f_own = F.n_f_own;
        uint64_t total = 0, mine_at = 0;
        for (uint32_t q = 0; q < world; ++q) {
            if (q == rank) mine_at = total;
            dm->shard_off[0][q] = 12ull * total;
            dm->shard_bytes[0][q] = 12ull * dm->shard_cnt_host[q];
            total += dm->shard_cnt_host[q];
        }
        if (dm->shard_cnt_host[rank] != n_f_own || total > 0xFFFFFFF0ull)
            return dm_fail(dm, LA3DM_ERR_HIP, "devmap: sharded sample filter: inconsistent counts after the exchange");
        n_f = (uint32_t)total;
        if (n_f) {
            // (dm->shard_frees was sized for every raw sample by the local part: a rank that cannot hold the list has
            //  already said so in its status word)
            if (n_f_own)
                hipLaunchKernelGGL(dm_copy_f3, dim3(cdiv(3 * n_f_own, 256)), dim3(256), 0, st, (const float *)dm->frees_ds.ptr, 3 * n_f_own,
                                   (float *)dm->shard_frees.ptr + 3 * mine_at);
            la3dm_gather_seg seg2 = {dm->shard_frees.ptr, dm->shard_off[0].data(), dm->shard_bytes[0].data()};
            if (dm->shard_fn(dm->shard_user, &seg2, 1, world, rank, (void *)st) != 0)
                return dm_fail(dm, LA3DM_ERR_ARG, "devmap: the all-gather callback of the sharded insert failed (filtered samples)");
        }
        d_frees = (const float *)dm->shard_frees.ptr;
    }
    const float free_label = ctx->p.variant == 1 ? -1.0f : 0.0f;  // bgkoctomap.cpp:415 / gpoctomap.cpp:399
    bool box_reduced = false;
    if (n_f) {
        MinmaxFin fin = {2, 0.0f, nullptr, (const float *)xy, dm->d_cnt, dm->d_mm + 6, mm_hits};
        publish_with(dm, fin.mailbox, fin.mailbox_seq);   // training_bbox only waits
        hipLaunchKernelGGL(dm_append_frees, dim3(std::min<uint32_t>(cdiv(n_f, 256), kMinmaxWgs)), dim3(256), 0, st, d_frees, n_f, n_kept, free_label,
                           xy, dm->d_mm, fin);
        box_reduced = true;
    }
    const uint32_t npts = n_kept + n_f;
    dm->n_xy = npts;
    S.n_hits = n_kept;
    S.n_frees = n_f;

    return training_bbox(dm, box_reduced);
}

// Host-side facts of the scan in flight, shared by the partition and the passes.
struct ScanPlan {
    PartArgs pa;
    CandArgs ca;
    uint64_t ncid = 0;       // cells of the dense block-index grid
    uint32_t n_entries = 0;  // candidate list (product of the three axis sequences)
    uint32_t max_occ = 1;    // passes: how often the most repeated candidate key occurs
    uint32_t n_mem = 0;      // (block, point) membership pairs = gathered training rows
    uint32_t n_geo = 0;      // training blocks
    uint32_t *train_off = nullptr;
    uint32_t *rows_off = nullptr;  // BGKLOctoMap: CSR of the training rows over the training blocks
    uint32_t flags = 0;      // la3dm_bgk_scan.flags of the passes (LA3DM_SCAN_UPDATE_UNGATED for insert_training_data)
    bool slab = false;       // sharded, single pass: the CSR is built per rank, for its own x-slab, inside the pass (build_slab_csr)
    int cell_bits = 0;       // key bits of a grid cell index
};

// f2 (bgkoctomap.cpp:234-284, 486-552): candidate sequences of get_blocks_in_bbox, closed-box membership of every
// training point, CSR by block, dense block-index grid.
static int partition(la3dm_devmap *dm, ScanPlan &P) {
    la3dm_ctx *ctx = dm->ctx;
    hipStream_t st = ctx->stream;
    la3dm_devmap_stats &S = dm->stats;
    const uint32_t npts = dm->n_xy;
    const float4 *xy = (const float4 *)dm->xy.ptr;
    int rc;
    const float bs = dm->block_size, half = bs / 2.0f;
    std::vector<int> seq[3];
    int smin[3], smax[3];
    for (int a = 0; a < 3; ++a) {
        seq[a] = axis_sequence(dm->h_bbox[a], dm->h_bbox[3 + a], bs);
        if (seq[a].empty() || seq[a].size() > (1u << 20)) return dm_fail(dm, LA3DM_ERR_ARG, "devmap: degenerate training-set extent");
        smin[a] = smax[a] = seq[a][0];
        for (int v : seq[a]) {
            smin[a] = std::min(smin[a], v);
            smax[a] = std::max(smax[a], v);
        }
    }
    PartArgs &pa = P.pa;
    pa.bs = bs;
    pa.half = half;
    uint64_t ncid = 1;
    for (int a = 0; a < 3; ++a) {
        pa.g0[a] = smin[a] - 2;
        pa.gn[a] = smax[a] - smin[a] + 5;
        ncid *= (uint64_t)pa.gn[a];
    }
    if (ncid > (1ull << 27))
        return dm_fail(dm, LA3DM_ERR_ARG, "devmap: scan extent / block size needs more than 2^27 block cells (set max_range)");
    // axis tables: seq (int) | rank (u8) | mult (u8), three axes back to back
    std::vector<uint8_t> tab;
    size_t off_seq[3], off_rank[3], off_mult[3];
    uint32_t max_occ = 1;
    {
        size_t o = 0;
        for (int a = 0; a < 3; ++a) {
            off_seq[a] = o;
            o += 4 * seq[a].size();
        }
        for (int a = 0; a < 3; ++a) {
            off_rank[a] = o;
            o += seq[a].size();
        }
        for (int a = 0; a < 3; ++a) {
            off_mult[a] = o;
            o += (size_t)pa.gn[a];
        }
        tab.assign((o + 15) & ~(size_t)15, 0);
        for (int a = 0; a < 3; ++a) {
            memcpy(&tab[off_seq[a]], seq[a].data(), 4 * seq[a].size());
            uint8_t *mult = &tab[off_mult[a]], *rank = &tab[off_rank[a]];
            uint32_t mx = 0;
            for (size_t k = 0; k < seq[a].size(); ++k) {
                uint8_t &m = mult[seq[a][k] - pa.g0[a]];
                if (m == 255) return dm_fail(dm, LA3DM_ERR_ARG, "devmap: a candidate index repeats more than 255 times");
                rank[k] = m++;
                mx = std::max<uint32_t>(mx, m);
            }
            max_occ *= mx;
        }
    }
    DM_RESERVE(dm->axis_tab, tab.size());
    DM_TRY(hipMemcpyAsync(dm->axis_tab.ptr, tab.data(), tab.size(), hipMemcpyHostToDevice, st));
    CandArgs &ca = P.ca;
    for (int a = 0; a < 3; ++a) {
        const uint8_t *base = (const uint8_t *)dm->axis_tab.ptr;
        pa.mult[a] = base + off_mult[a];
        ca.seq[a] = (const int *)(base + off_seq[a]);
        ca.rank[a] = base + off_rank[a];
        ca.nseq[a] = (int)seq[a].size();
    }
    ca.part = pa;
    const uint64_t n_entries64 = (uint64_t)seq[0].size() * seq[1].size() * seq[2].size();
    if (n_entries64 > (1ull << 30)) return dm_fail(dm, LA3DM_ERR_ARG, "devmap: candidate list too long");
    const uint32_t n_entries = (uint32_t)n_entries64;
    S.n_bbox_blocks = n_entries;

    // membership pairs (block, point), grouped by block with ascending point index inside a block
    DM_RESERVE(dm->flag, 4ull * npts);
    DM_RESERVE(dm->scan, 4ull * npts);
    uint32_t *m_cnt = (uint32_t *)dm->flag.ptr, *m_off = (uint32_t *)dm->scan.ptr;
    DM_RESERVE(dm->m_code, 16ull * npts);
    hipLaunchKernelGGL(dm_members_count, dim3(cdiv(npts, 256)), dim3(256), 0, st, xy, npts, pa, m_cnt,
                       (int4 *)dm->m_code.ptr);
    int bits = 1;
    while ((1ull << bits) < ncid) ++bits;
    P.cell_bits = bits;
    const bool want_slab = dm->shard_slab != 0;
    if (((dm->shard_world > 1 && dm->shard_fn && want_slab) || dm->force_slab) && max_occ == 1 && ctx->p.variant != 3) {
        // x-slab partition (devmap_kernels.h): only the per-cell point counts are formed for all blocks here; pairs, sort, CSR, rows
        // and neighbour tables follow inside the pass, once the cut is known, for the cells of this rank's slab (build_slab_csr)
        DM_RESERVE(dm->cell_cnt, 4ull * ncid);
        DM_RESERVE(dm->grid, 4ull * ncid);
        DM_TRY(hipMemsetAsync(dm->cell_cnt.ptr, 0, 4ull * ncid, st));
        hipLaunchKernelGGL(dm_members_hist, dim3(cdiv(npts, 256)), dim3(256), 0, st, (const int4 *)dm->m_code.ptr, npts, pa, (uint32_t *)dm->cell_cnt.ptr,
                           dm->d_cnt);
        hipLaunchKernelGGL(dm_cell_trained, dim3(std::min<uint32_t>(cdiv((uint32_t)ncid, 256), 64u)), dim3(256), 0, st, (const uint32_t *)dm->cell_cnt.ptr, (uint32_t)ncid, pa,
                           dm->d_cnt);
        DM_RESERVE(dm->c_flag, 4ull * n_entries);
        DM_RESERVE(dm->c_scan, 4ull * n_entries);
        P.slab = true;
        P.ncid = ncid;
        P.n_entries = n_entries;
        P.max_occ = max_occ;
        P.n_mem = 0;
        P.n_geo = 0;
        P.train_off = nullptr;
        return LA3DM_OK;
    }
    if ((rc = exclusive_scan(dm, m_cnt, m_off, npts, (int)kCntMembers, true)) != LA3DM_OK) return rc;
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    const uint32_t n_mem = dm->h_cnt[kCntMembers];
    DM_RESERVE(dm->k0, 4ull * n_mem);
    DM_RESERVE(dm->k1, 4ull * n_mem);
    DM_RESERVE(dm->v0, 4ull * n_mem);
    DM_RESERVE(dm->v1, 4ull * n_mem);
    uint32_t *k0 = (uint32_t *)dm->k0.ptr, *k1 = (uint32_t *)dm->k1.ptr, *v0 = (uint32_t *)dm->v0.ptr, *v1 = (uint32_t *)dm->v1.ptr;
    DM_RESERVE(dm->grid, 4ull * ncid);
    if (n_mem && sort_fusable(dm, bits)) {   // the pairs are written by the sort's histogram launch
        const MembersSrc src = {(const int4 *)dm->m_code.ptr, pa, m_off, k0, v0, dm->d_cnt, (int32_t *)dm->grid.ptr, (uint32_t)ncid};
        if ((rc = sort_pairs_src(dm, src, npts, k0, k1, v0, v1, n_mem, bits)) != LA3DM_OK) return rc;
    } else {
        hipLaunchKernelGGL(dm_members_write, dim3(cdiv(npts, 256)), dim3(256), 0, st, (const int4 *)dm->m_code.ptr, npts, pa, m_off,
                           k0, v0, dm->d_cnt, (int32_t *)dm->grid.ptr, (uint32_t)ncid);
        if ((rc = sort_pairs(dm, k0, k1, v0, v1, n_mem, bits)) != LA3DM_OK) return rc;
    }
    DM_RESERVE(dm->c_flag, 4ull * std::max(n_mem, n_entries));
    DM_RESERVE(dm->c_scan, 4ull * std::max(n_mem, n_entries));
    DM_RESERVE(dm->seg_start, 4ull * (n_mem + 1));
    DM_RESERVE(dm->seg_key, 4ull * (n_mem + 1));
    uint32_t *sflag = (uint32_t *)dm->c_flag.ptr;
    uint32_t *train_off = (uint32_t *)dm->seg_start.ptr, *seg_key = (uint32_t *)dm->seg_key.ptr;
    // (the head flags have one reader — BGK-L's row flags; the prefix array none: left unwritten)
    if ((rc = scan_heads(dm, k1, n_mem, ctx->p.variant == 3 ? sflag : nullptr, nullptr, train_off, seg_key, (int)kCntGeo, (int)kCntGridValid)) != LA3DM_OK) return rc;
    if (ctx->p.variant == 3) {  // training rows: hits as degenerate segments, every beam once per block
        DM_RESERVE(dm->l_rflag, 4ull * n_mem);
        DM_RESERVE(dm->l_rscan, 4ull * n_mem);
        DM_RESERVE(dm->l_rows, 48ull * n_mem);   // 12 floats per row (LA3DM_SCAN_ROWS_PREPARED)
        DM_RESERVE(dm->l_rows_off, 4ull * ((size_t)n_mem + 2));
        uint32_t *rflag = (uint32_t *)dm->l_rflag.ptr, *rscan = (uint32_t *)dm->l_rscan.ptr;
        hipLaunchKernelGGL(dm_l_row_flags, dim3(cdiv(n_mem, 256)), dim3(256), 0, st, v1, sflag, n_mem,
                           (const int32_t *)dm->l_ray_idx.ptr, rflag);
        if ((rc = exclusive_scan(dm, rflag, rscan, n_mem)) != LA3DM_OK) return rc;
        hipLaunchKernelGGL(dm_l_rows_write, dim3(cdiv(n_mem, 256)), dim3(256), 0, st, v1, n_mem, rflag, rscan, xy,
                           (const int32_t *)dm->l_ray_idx.ptr, (const float *)dm->l_rays.ptr, (float4 *)dm->l_rows.ptr);
        hipLaunchKernelGGL(dm_l_rows_off, dim3(cdiv((size_t)n_mem + 1, 256)), dim3(256), 0, st, train_off, dm->d_cnt, rflag, rscan,
                           n_mem, (uint32_t *)dm->l_rows_off.ptr);
        P.rows_off = (uint32_t *)dm->l_rows_off.ptr;
    } else {
        DM_RESERVE(dm->train, 16ull * n_mem);
    }
    hipLaunchKernelGGL(dm_gather_geo, dim3(cdiv(n_mem, 256)), dim3(256), 0, st, xy, (const uint32_t *)v1, n_mem,
                       ctx->p.variant == 3 ? (float4 *)nullptr : (float4 *)dm->train.ptr, (const uint32_t *)seg_key, dm->d_cnt, pa,
                       (int32_t *)dm->grid.ptr);
    hipLaunchKernelGGL(dm_count_trained, dim3(std::min<uint32_t>(cdiv(n_mem, 256 * 4), 64u)), dim3(256), 0, st, (const uint32_t *)seg_key, pa, dm->d_cnt);
    // the segment count is only needed on the host by the GP launches; the BGK path reads it (and the error flag)
    // together with the test-block count of the first pass
    uint32_t n_geo = n_mem;
    if (ctx->p.variant == 1) {
        if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
        n_geo = dm->h_cnt[kCntGeo];
    }
    P.ncid = ncid;
    P.n_entries = n_entries;
    P.max_occ = max_occ;
    P.n_mem = n_mem;
    P.n_geo = n_geo;
    P.train_off = train_off;
    return LA3DM_OK;
}

// x-slab partition, second half (sharded single-pass inserts; devmap_kernels.h): the (block, point) pairs, their sort, the CSR, the
// training rows and the dense cell -> training-block index — for the grid cells of the own range [t0, t1) of the test list and its
// halo (the cells of those test blocks' extended blocks) only.  Training-block indices are local to the rank.
static int build_slab_csr(la3dm_devmap *dm, ScanPlan &P, const uint32_t *t_ent, uint32_t t0, uint32_t t1) {
    la3dm_ctx *ctx = dm->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t npts = dm->n_xy;
    const float4 *xy = (const float4 *)dm->xy.ptr;
    const PartArgs &pa = P.pa;
    const uint32_t ncid = (uint32_t)P.ncid;
    int rc;
    if (dm->dbg_fail_slab_rank >= 0 && dm->shard_world > 1 && dm->dbg_fail_slab_rank == (int)dm->shard_rank)
        return dm_fail(dm, LA3DM_ERR_OOM, "devmap: injected rank-local failure in the x-slab partition (LA3DM_INJECT_SLAB_FAILURE)");
    DM_RESERVE(dm->slab_range, 8);
    uint32_t *range = (uint32_t *)dm->slab_range.ptr;
    DM_TRY(hipMemsetAsync(range, 0xFF, 4, st));
    DM_TRY(hipMemsetAsync(range + 1, 0, 4, st));
    if (t1 > t0) hipLaunchKernelGGL(dm_slab_range, dim3(1), dim3(64), 0, st, P.ca, t_ent, t0, t1, range, dm->shard_world > 1 ? 0u : ncid);
    uint32_t *m_cnt = (uint32_t *)dm->flag.ptr, *m_off = (uint32_t *)dm->scan.ptr;   // (reserved by partition: 4 npts each)
    hipLaunchKernelGGL(dm_members_count_slab, dim3(cdiv(npts, 256)), dim3(256), 0, st, (const int4 *)dm->m_code.ptr, npts, pa, (const uint32_t *)range, m_cnt);
    if ((rc = exclusive_scan(dm, m_cnt, m_off, npts, (int)kCntMembers, true)) != LA3DM_OK) return rc;
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    const uint32_t n_mem = dm->h_cnt[kCntMembers];
    const size_t pb = 4ull * std::max<uint32_t>(n_mem, 4u);
    DM_RESERVE(dm->k0, pb);
    DM_RESERVE(dm->k1, pb);
    DM_RESERVE(dm->v0, pb);
    DM_RESERVE(dm->v1, pb);
    uint32_t *k0 = (uint32_t *)dm->k0.ptr, *k1 = (uint32_t *)dm->k1.ptr, *v0 = (uint32_t *)dm->v0.ptr, *v1 = (uint32_t *)dm->v1.ptr;
    DM_RESERVE(dm->c_flag, pb);
    DM_RESERVE(dm->c_scan, pb);
    DM_RESERVE(dm->seg_start, 4ull * ((size_t)n_mem + 1));
    DM_RESERVE(dm->seg_key, 4ull * ((size_t)n_mem + 1));
    DM_RESERVE(dm->train, 16ull * std::max<uint32_t>(n_mem, 1u));
    uint32_t *train_off = (uint32_t *)dm->seg_start.ptr, *seg_key = (uint32_t *)dm->seg_key.ptr;
    if (n_mem) {
        const MembersSlabSrc src = {(const int4 *)dm->m_code.ptr, pa, m_off, range, k0, v0, (int32_t *)dm->grid.ptr, ncid};
        if (sort_fusable(dm, P.cell_bits)) {   // the pairs are written by the sort's histogram launch
            if ((rc = sort_pairs_src(dm, src, npts, k0, k1, v0, v1, n_mem, P.cell_bits)) != LA3DM_OK) return rc;
        } else {
            hipLaunchKernelGGL((dm_run_src<MembersSlabSrc>), dim3(cdiv(npts, 256)), dim3(256), 0, st, src, npts);
            if ((rc = sort_pairs(dm, k0, k1, v0, v1, n_mem, P.cell_bits)) != LA3DM_OK) return rc;
        }
        if ((rc = scan_heads(dm, k1, n_mem, nullptr, nullptr, train_off, seg_key, (int)kCntGeo, (int)kCntGridValid)) != LA3DM_OK)
            return rc;
        hipLaunchKernelGGL(dm_gather_geo, dim3(cdiv(n_mem, 256)), dim3(256), 0, st, xy, (const uint32_t *)v1, n_mem, (float4 *)dm->train.ptr,
                           (const uint32_t *)seg_key, dm->d_cnt, pa, (int32_t *)dm->grid.ptr);
    } else {   // a rank whose slab holds no training point: no block has a model
        DM_TRY(hipMemsetAsync(dm->grid.ptr, 0xFF, 4ull * ncid, st));
        DM_TRY(hipMemsetAsync(train_off, 0, 8, st));
    }
    P.n_mem = n_mem;
    P.n_geo = n_mem;
    P.train_off = train_off;
    if (ctx->p.variant == 1) {   // the GP launches are sized by the number of training blocks
        if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
        P.n_geo = n_mem ? dm->h_cnt[kCntGeo] : 0u;
    }
    if (getenv("LA3DM_DEBUG_SHARD"))
        fprintf(stderr, "la3dm x-slab partition: rank %u of %u builds the CSR of %u (block, point) pairs for test blocks [%u, %u)\n", dm->shard_rank,
                dm->shard_world, n_mem, t0, t1);
    return LA3DM_OK;
}

// block_depth 3: the leaf-list and write-back + prune launches with `batch` (4 or 8) test blocks per wave (devmap_depth3.h)
static void launch_leaves_d3(bool emit, uint32_t batch, uint32_t grid, hipStream_t st, const uint32_t *slot, const uint32_t *counters,
                             const uint8_t *S, const float *A, const float *B, uint32_t *nleaf, const uint32_t *leaf_off, uint32_t *leaf_key,
                             float *alpha, float *beta, uint32_t *leaf_node, const LeafExtra &lx) {
    if (emit && batch == 8)
        hipLaunchKernelGGL((dm_leaves_d3<true, 8>), dim3(grid), dim3(256), 0, st, slot, counters, S, A, B, nleaf, leaf_off, leaf_key, alpha, beta, leaf_node, lx);
    else if (emit)
        hipLaunchKernelGGL((dm_leaves_d3<true, 4>), dim3(grid), dim3(256), 0, st, slot, counters, S, A, B, nleaf, leaf_off, leaf_key, alpha, beta, leaf_node, lx);
    else if (batch == 8)
        hipLaunchKernelGGL((dm_leaves_d3<false, 8>), dim3(grid), dim3(256), 0, st, slot, counters, S, A, B, nleaf, leaf_off, leaf_key, alpha, beta, leaf_node, lx);
    else
        hipLaunchKernelGGL((dm_leaves_d3<false, 4>), dim3(grid), dim3(256), 0, st, slot, counters, S, A, B, nleaf, leaf_off, leaf_key, alpha, beta, leaf_node, lx);
}
static void launch_commit_prune_d3(uint32_t batch, hipStream_t st, const uint32_t *slot, uint32_t n_test, const uint32_t *leaf_off,
                                   const uint32_t *leaf_node, const uint32_t *leaf_key, const float *alpha, const float *beta,
                                   const uint8_t *state, float *A, float *B, uint8_t *S, uint32_t *counters, uint32_t *done,
                                   volatile uint32_t *mailbox, uint32_t mseq) {
    const dim3 grid(std::min(cdiv(cdiv(n_test, batch), 4), kCommitPruneWgs));
    const size_t lds = 4 * (size_t)batch * prune_lds_stride(kD3Npb);
    if (batch == 8)
        hipLaunchKernelGGL((dm_commit_prune_d3<8>), grid, dim3(256), lds, st, slot, n_test, leaf_off, leaf_node, leaf_key, alpha, beta, state, A, B, S,
                           counters, done, mailbox, mseq);
    else
        hipLaunchKernelGGL((dm_commit_prune_d3<4>), grid, dim3(256), lds, st, slot, n_test, leaf_off, leaf_node, leaf_key, alpha, beta, state, A, B, S,
                           counters, done, mailbox, mseq);
}

// One pass over the candidate list (bgkoctomap.cpp:286-353): test-block decision, find-or-create, leaves in
// LeafIterator order, predict + fuse, write-back, prune.  A pass holds every candidate key once; keys the float
// stepping of get_blocks_in_bbox repeats come back in later passes, as in the serial reference.
static int run_pass(la3dm_devmap *dm, ScanPlan &P, uint32_t pass, uint32_t *n_test0) {
    la3dm_ctx *ctx = dm->ctx;
    hipStream_t st = ctx->stream;
    la3dm_devmap_stats &S = dm->stats;
    CandArgs &ca = P.ca;
    const uint32_t n_entries = P.n_entries, max_occ = P.max_occ, ncell = dm->ncell;
    uint32_t n_mem = P.n_mem;
    uint32_t *train_off = P.train_off;   // (both set by build_slab_csr, further down, in the x-slab form of a sharded insert)
    uint32_t *c_flag = (uint32_t *)dm->c_flag.ptr, *c_weight = (uint32_t *)dm->c_weight.ptr, *c_scan = (uint32_t *)dm->c_scan.ptr;
    uint32_t &n_geo = P.n_geo;
    int rc;
    const double tp0 = wall();
    ca.pass = pass;
    if (P.slab)
        hipLaunchKernelGGL((dm_candidates<true>), dim3(cdiv(n_entries, 256)), dim3(256), 0, st, ca, (const int32_t *)dm->cell_cnt.ptr,
                           (const uint32_t *)nullptr, n_entries, c_flag, c_weight);
    else
        hipLaunchKernelGGL((dm_candidates<false>), dim3(cdiv(n_entries, 256)), dim3(256), 0, st, ca, (const int32_t *)dm->grid.ptr,
                           (const uint32_t *)train_off, n_entries, c_flag, c_weight);
    DM_RESERVE(dm->t_key0, 4ull * n_entries);
    DM_RESERVE(dm->t_ent0, 4ull * n_entries);
    // (the scan's total is the test-block count: it publishes the counters, the compaction runs while the host waits)
    if ((rc = exclusive_scan(dm, c_flag, c_scan, n_entries, (int)kCntTest, true)) != LA3DM_OK) return rc;
    // Heaviest test blocks first (the blocks are independent: order only balances the launch).  The compaction is the histogram
    // launch of that one-digit sort, and the pass behind it takes the list's length from the counter block: both are queued
    // before the host has read the length (round 5; compaction, read-back, histogram, pass were four steps in a row).
    const bool sharded = dm->shard_world > 1;
    const uint32_t world = dm->shard_world;
    const bool sorted_on_device = !sharded && dm->test_sort && sort_fusable(dm, 32, 24);
    if (sorted_on_device) {
        DM_RESERVE(dm->t_key1, 4ull * n_entries);
        DM_RESERVE(dm->t_ent1, 4ull * n_entries);
        const TestCompactSrc src = {c_flag, c_scan, c_weight, n_entries, (uint32_t *)dm->t_key0.ptr, (uint32_t *)dm->t_ent0.ptr, dm->d_cnt};
        SortJob job;
        if ((rc = sort_begin_src(dm, src, n_entries, n_entries, 32, 24, job)) != LA3DM_OK) return rc;   // the weight class
        if ((rc = sort_passes(dm, job, (const uint32_t *)dm->t_key0.ptr, (uint32_t *)dm->t_key1.ptr, (const uint32_t *)dm->t_ent0.ptr,
                              (uint32_t *)dm->t_ent1.ptr, dm->d_cnt + kCntTest)) != LA3DM_OK)
            return rc;
    } else {
        hipLaunchKernelGGL(dm_test_compact, dim3(cdiv(n_entries, 256)), dim3(256), 0, st, c_flag, c_scan, c_weight, n_entries,
                           (uint32_t *)dm->t_key0.ptr, (uint32_t *)dm->t_ent0.ptr, dm->d_cnt);
    }
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    if (dm->h_cnt[kCntError])
        return dm_fail(dm, LA3DM_ERR_ARG, "devmap: internal error: training point outside the block index grid");
    if (ctx->p.variant != 1 && !P.slab) n_geo = dm->h_cnt[kCntGeo];
    S.n_train_blocks = dm->h_cnt[kCntTrained];
    const uint32_t n_test = dm->h_cnt[kCntTest];
    if (n_test == 0) return LA3DM_OK;
    S.n_test_blocks += n_test;
    S.n_passes = pass + 1;
    // the list the rest of the pass works on: sorted copies, or (sharded / LA3DM_TEST_SORT=0) the compacted list in candidate order
    const uint32_t *t_key = (const uint32_t *)dm->t_key1.ptr, *t_ent = (const uint32_t *)dm->t_ent1.ptr;
    if (sorted_on_device) {
        // (queued above)
    } else if (!sharded && dm->test_sort) {   // the library sort (LA3DM_OWN_SORT=0)
        DM_RESERVE(dm->t_key1, 4ull * n_test);
        DM_RESERVE(dm->t_ent1, 4ull * n_test);
        t_key = (const uint32_t *)dm->t_key1.ptr;
        t_ent = (const uint32_t *)dm->t_ent1.ptr;
        if (n_test <= kSortSmallMax) {
            uint32_t N = 2;
            while (N < n_test) N <<= 1;
            hipLaunchKernelGGL(dm_sort_small, dim3(1), dim3(1024), 0, st, (const uint32_t *)dm->t_key0.ptr, (const uint32_t *)dm->t_ent0.ptr,
                               n_test, N, (uint32_t *)dm->t_key1.ptr, (uint32_t *)dm->t_ent1.ptr);
        } else if ((rc = sort_pairs(dm, (uint32_t *)dm->t_key0.ptr, (uint32_t *)dm->t_key1.ptr, (uint32_t *)dm->t_ent0.ptr,
                                    (uint32_t *)dm->t_ent1.ptr, n_test, 32, 24)) != LA3DM_OK)   // the weight class
            return rc;
    } else {
        t_key = (const uint32_t *)dm->t_key0.ptr;
        t_ent = (const uint32_t *)dm->t_ent0.ptr;
    }
    if (sharded) {
        // block-sharded: the list stays in candidate order (block indices x-major: neighbouring test blocks, which share
        // training blocks, stay on one GPU's L2) and is cut into `world` contiguous ranges of equal weight
        DM_RESERVE(dm->shard_w, 4ull * n_test);
        DM_RESERVE(dm->shard_cumw, 4ull * n_test);
        DM_RESERVE(dm->shard_bounds, 8ull * (world + 1));
        // (a block's weight is capped so that the 32-bit running sum cannot wrap: n_test * cap < 2^31 — ADVICE r02)
        const uint32_t w_cap = std::max<uint32_t>(32u, (uint32_t)((1ull << 31) / n_test));
        hipLaunchKernelGGL(dm_shard_weight, dim3(cdiv(n_test, 256)), dim3(256), 0, st, t_key, n_test,
                           w_cap, (uint32_t *)dm->shard_w.ptr);
        if ((rc = exclusive_scan(dm, (const uint32_t *)dm->shard_w.ptr, (uint32_t *)dm->shard_cumw.ptr, n_test)) != LA3DM_OK) return rc;
        hipLaunchKernelGGL(dm_shard_bounds, dim3(1), dim3(1024), 0, st, (const uint32_t *)dm->shard_cumw.ptr,
                           (const uint32_t *)dm->shard_w.ptr, n_test, world, (uint32_t *)dm->shard_bounds.ptr);
    }
    DM_RESERVE(dm->t_blockkey, 8ull * n_test);
    DM_RESERVE(dm->t_center, 12ull * n_test);
    DM_RESERVE(dm->t_nbr, 28ull * n_test);
    DM_RESERVE(dm->t_slot, 4ull * n_test);
    // Every fallible reservation of the pack stage comes BEFORE the launch that creates blocks: dm_test_build
    // publishes new keys and bumps the device block count, and the new blocks' default nodes are only written by the
    // leaf-count launch behind it — nothing that can fail may sit between the two.
    DM_RESERVE(dm->nleaf, 4ull * (n_test + 1));
    DM_RESERVE(dm->leaf_off, 4ull * (n_test + 1));
    const size_t max_leaves = (size_t)n_test * ncell;
    DM_RESERVE(dm->leaf_key, 4 * max_leaves);
    DM_RESERVE(dm->leaf_alpha, 4 * max_leaves);
    DM_RESERVE(dm->leaf_beta, 4 * max_leaves);
    DM_RESERVE(dm->leaf_node, 4 * max_leaves);
    DM_RESERVE(dm->leaf_state, max_leaves);
    // blocks: find or create (bgkoctomap.cpp:298-305), in the launch that builds the test blocks' keys and neighbour tables
    if ((rc = grow_pool(dm, (size_t)dm->n_blocks + n_test)) != LA3DM_OK) return rc;
    if ((rc = grow_table(dm, (size_t)dm->n_blocks + n_test)) != LA3DM_OK) return rc;
    // (d_cnt[kCntBlocks] holds the pool's block count since dm_begin; the passes keep it current)
    {
        const TableArgs tb = {dm->tab_key, dm->tab_val, dm->tab_cap - 1, dm->d_cnt + kCntBlocks, dm->blk_key};
        hipLaunchKernelGGL(dm_test_build, dim3(cdiv(n_test, 256)), dim3(256), 0, st, ca, P.slab ? (const int32_t *)nullptr : (const int32_t *)dm->grid.ptr,
                           t_ent, dm->d_cnt, (long long *)dm->t_blockkey.ptr, (float *)dm->t_center.ptr,
                           (int32_t *)dm->t_nbr.ptr, tb, (uint32_t *)dm->t_slot.ptr);
    }
    // default nodes for the blocks this launch created — slots [old count, new count); the new count stays on the device (read
    // back with the pass's other counters) — are written by the leaf-count launch below (LeafExtra)
    // pack: leaves in LeafIterator order
    uint32_t *nleaf = (uint32_t *)dm->nleaf.ptr, *leaf_off = (uint32_t *)dm->leaf_off.ptr;
    LeafExtra lx;
    memset(&lx, 0, sizeof(lx));
    lx.old_blocks = dm->n_blocks;
    lx.a0 = dm->init_A;
    lx.b0 = dm->init_B;
    lx.A_w = dm->A;
    lx.B_w = dm->B;
    lx.S_w = dm->S;
    // test blocks per wave (devmap_depth3.h; measured at configs[4]'s 266 k test blocks: 8 beats 4 by 27 us, at configs[1]'s 42 k they are level)
    const int d3_env = dm->depth3_batch < 0 ? (n_test >= 65536u ? 8 : 4) : dm->depth3_batch;
    const uint32_t d3 = dm->depth == 3 && dm->npb == kD3Npb ? (d3_env >= 8 ? 8u : d3_env > 0 ? 4u : 0u) : 0u;
    if (d3)
        launch_leaves_d3(false, d3, cdiv(n_test, 4 * d3), st, (const uint32_t *)dm->t_slot.ptr, dm->d_cnt, (const uint8_t *)dm->S, (const float *)dm->A,
                         (const float *)dm->B, nleaf, nullptr, nullptr, nullptr, nullptr, nullptr, lx);
    else
        hipLaunchKernelGGL((dm_leaves<false>), dim3(cdiv(n_test, 4)), dim3(256), 0, st, (const uint32_t *)dm->t_slot.ptr, dm->d_cnt,
                           (const uint8_t *)dm->S, (const float *)dm->A, (const float *)dm->B, dm->npb, dm->depth, nleaf,
                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (float *)nullptr, (float *)nullptr, (uint32_t *)nullptr, lx);
    if (ctx->p.variant == 3)
        hipLaunchKernelGGL(dm_l_test_stats, dim3(std::min(cdiv(n_test, 256), 32u)), dim3(256), 0, st, (const int32_t *)dm->t_nbr.ptr,
                           (const uint32_t *)P.rows_off, (const uint32_t *)nleaf, n_test, dm->d_cnt);
    if ((rc = exclusive_scan(dm, nleaf, leaf_off, n_test + 1, (int)kCntLeaves)) != LA3DM_OK) return rc;
    // Block-sharded, single pass (round 5, VERDICT r04 #4b): a rank lists the leaves of its OWN range of test blocks only; the
    // other ranks' leaves arrive through the all-gather-v with their keys (13 B per leaf instead of 9), and the write-back finds
    // their nodes from this replica's slot of the block.  The range cut needs the leaf offsets, so the one host wait of the
    // sharded path moves in front of the emitting launch.
    const bool own_emit = sharded && max_occ == 1;
    uint32_t t0s = 0, t1s = n_test;
    if (sharded) {
        uint32_t *lb = (uint32_t *)dm->shard_bounds.ptr + (world + 1);
        hipLaunchKernelGGL(dm_shard_leaf_bounds, dim3(1), dim3(1024), 0, st, (const uint32_t *)dm->shard_bounds.ptr,
                           (const uint32_t *)leaf_off, world, lb);
        DM_TRY(hipMemcpyAsync(dm->h_shard, dm->shard_bounds.ptr, 8ull * (world + 1), hipMemcpyDeviceToHost, st));
        DM_TRY(hipStreamSynchronize(st));   // (the one host wait of the sharded path: launch sizes depend on the cut)
        const uint32_t *hb = dm->h_shard, *hl = dm->h_shard + (world + 1);
        for (uint32_t q = 0; q < world; ++q)
            if (hb[q] > hb[q + 1] || hb[q + 1] > n_test || hl[q] > hl[q + 1])
                return dm_fail(dm, LA3DM_ERR_HIP, "devmap: internal error: the range cut of the sharded insert is not monotone");
        t0s = hb[dm->shard_rank];
        t1s = hb[dm->shard_rank + 1];
    }
    int rc_slab = LA3DM_OK;
    if (P.slab) {
        // the cut is known: pairs, sort, CSR and rows of the own slab, then the own range's neighbour tables.  A rank of a sharded
        // insert that fails here (out of memory for its pairs, ...) must still enter the leaf exchange its peers are about to wait
        // in: it skips its kernel, hands its range over as "no leaf updated" and reports afterwards (see the exchange below)
        rc_slab = build_slab_csr(dm, P, t_ent, t0s, t1s);
        if (rc_slab != LA3DM_OK && !sharded) return rc_slab;
        n_mem = P.n_mem;
        train_off = P.train_off;
        if (rc_slab == LA3DM_OK && t1s > t0s)
            hipLaunchKernelGGL(dm_test_nbr, dim3(cdiv(t1s - t0s, 256)), dim3(256), 0, st, ca, (const int32_t *)dm->grid.ptr, t_ent, t0s, t1s,
                               (int32_t *)dm->t_nbr.ptr);
    }
    {
        // the emitting launch carries the pass's work counters (train_reads, pair_evals) in a few workgroups of its own
        const uint32_t e0 = own_emit ? t0s : 0u, e1 = own_emit ? t1s : n_test;
        const uint32_t main_wgs = cdiv(e1 - e0, d3 ? 4 * d3 : 4), stat_wgs = ctx->p.variant == 3 ? 0u : std::min(cdiv(n_test, 256), 32u);
        lx.t_key = stat_wgs ? t_key : nullptr;
        lx.counters_w = dm->d_cnt;
        lx.main_wgs = main_wgs;
        lx.t_begin = e0;
        lx.t_end = e1;
        if ((main_wgs + stat_wgs) && d3)
            launch_leaves_d3(true, d3, main_wgs + stat_wgs, st, (const uint32_t *)dm->t_slot.ptr, dm->d_cnt, (const uint8_t *)dm->S, (const float *)dm->A,
                             (const float *)dm->B, nleaf, (const uint32_t *)leaf_off, (uint32_t *)dm->leaf_key.ptr, (float *)dm->leaf_alpha.ptr,
                             (float *)dm->leaf_beta.ptr, (uint32_t *)dm->leaf_node.ptr, lx);
        else if (main_wgs + stat_wgs)
            hipLaunchKernelGGL((dm_leaves<true>), dim3(main_wgs + stat_wgs), dim3(256), 0, st, (const uint32_t *)dm->t_slot.ptr, dm->d_cnt,
                               (const uint8_t *)dm->S, (const float *)dm->A, (const float *)dm->B, dm->npb, dm->depth, nleaf,
                               (const uint32_t *)leaf_off, (uint32_t *)dm->leaf_key.ptr, (float *)dm->leaf_alpha.ptr,
                               (float *)dm->leaf_beta.ptr, (uint32_t *)dm->leaf_node.ptr, lx);
    }
    double tp1 = tp0;
    if (dm->stage_timing) {
        DM_TRY(hipStreamSynchronize(st));
        tp1 = wall();
        S.t_pack += tp1 - tp0;
    }
    // E: predict + fuse
    la3dm_bgk_scan s;
    memset(&s, 0, sizeof(s));
    s.train_xyzy = (const float *)dm->train.ptr;
    s.train_off = train_off;
    s.n_train_pts = n_mem;
    if (ctx->p.variant == 3) {  // rows of 12 floats (prepared form); n_mem bounds their number
        s.train_xyzy = (const float *)dm->l_rows.ptr;
        s.train_off = P.rows_off;
    }
    s.n_train_blk = n_geo;
    s.nbr = (const int32_t *)dm->t_nbr.ptr;
    s.blk_center = (const float *)dm->t_center.ptr;
    s.leaf_off = leaf_off;
    s.n_test_blk = n_test;
    s.n_leaf = (uint32_t)std::min<size_t>(max_leaves, 0xFFFFFFFFu);
    s.leaf_key = (const uint32_t *)dm->leaf_key.ptr;
    s.alpha = (float *)dm->leaf_alpha.ptr;
    s.beta = (float *)dm->leaf_beta.ptr;
    s.state = (uint8_t *)dm->leaf_state.ptr;
    s.flags = P.flags;
    if (ctx->p.variant == 3) s.flags |= LA3DM_SCAN_ROWS_PREPARED;   // dm_l_rows_write wrote the 12-float form
    if (pass == 0 && dm->insert_into_empty) s.flags |= LA3DM_SCAN_FULL_BLOCKS;
    if (sharded) {
        // this rank's contiguous range of test blocks; leaf_off holds absolute leaf indices, so offsetting the per-block
        // arrays is all the kernel needs
        s.nbr += 7ull * t0s;
        s.blk_center += 3ull * t0s;
        s.leaf_off += t0s;
        s.n_test_blk = t1s - t0s;
    }
    rc = rc_slab;
    const std::string slab_err = rc_slab != LA3DM_OK ? ctx->err : std::string();
    if (s.n_test_blk && rc_slab == LA3DM_OK)
        rc = ctx->p.variant == 1   ? la3dm_gp_scan_device(ctx, &s, st, nullptr)
             : ctx->p.variant == 3 ? la3dm_bgkl_scan_device(ctx, &s, st, nullptr)
                                   : la3dm_bgk_scan_device(ctx, &s, st, nullptr);
    if (rc != LA3DM_OK && !sharded) return rc;
    double tg0 = tp1;
    if (sharded) {
        // ONE all-gather-v, in place on the leaf arrays (a rank's leaves are a contiguous index range of alpha, beta and
        // state: 9 B per leaf, no pack / unpack, no padding), queued on this stream by the callback; commit and prune
        // follow on the same stream on identical data everywhere.  A rank whose launch failed still enters the
        // collective (its peers would wait for ever otherwise) and reports afterwards.
        const int rc_kernel = rc;
        const uint32_t *hl = dm->h_shard + (world + 1);
        if (rc_kernel != LA3DM_OK && hl[dm->shard_rank + 1] > hl[dm->shard_rank]) {
            // this rank's range goes out as "update() ran on no leaf" (state 0: the write-back skips such leaves), so that every
            // replica — this one included — stays what it was for these blocks instead of committing whatever the arrays hold
            (void)hipMemsetAsync((uint8_t *)dm->leaf_state.ptr + hl[dm->shard_rank], 0, hl[dm->shard_rank + 1] - hl[dm->shard_rank], st);
        }
        for (int g = 0; g < 4; ++g) {
            dm->shard_off[g].resize(world);
            dm->shard_bytes[g].resize(world);
        }
        for (uint32_t q = 0; q < world; ++q) {
            const uint64_t f = hl[q], n_q = hl[q + 1] - hl[q];
            dm->shard_off[0][q] = dm->shard_off[1][q] = dm->shard_off[3][q] = 4 * f;
            dm->shard_bytes[0][q] = dm->shard_bytes[1][q] = dm->shard_bytes[3][q] = 4 * n_q;
            dm->shard_off[2][q] = f;
            dm->shard_bytes[2][q] = n_q;
        }
        la3dm_gather_seg segs[4] = {{dm->leaf_alpha.ptr, dm->shard_off[0].data(), dm->shard_bytes[0].data()},
                                    {dm->leaf_beta.ptr, dm->shard_off[1].data(), dm->shard_bytes[1].data()},
                                    {dm->leaf_state.ptr, dm->shard_off[2].data(), dm->shard_bytes[2].data()},
                                    {dm->leaf_key.ptr, dm->shard_off[3].data(), dm->shard_bytes[3].data()}};
        const uint32_t n_segs = own_emit ? 4u : 3u;   // (the keys travel only when the ranks did not list each other's leaves)
        if (dm->stage_timing) {
            DM_TRY(hipStreamSynchronize(st));
            tg0 = wall();
        }
        const int xrc = hl[world] ? dm->shard_fn(dm->shard_user, segs, n_segs, world, dm->shard_rank, (void *)st) : 0;
        if (rc_kernel != LA3DM_OK) {
            if (!slab_err.empty()) ctx->err = slab_err;
            return rc_kernel;
        }
        if (xrc != 0) return dm_fail(dm, LA3DM_ERR_ARG, "devmap: the all-gather callback of the sharded insert failed");
        if (dm->stage_timing) {
            DM_TRY(hipStreamSynchronize(st));
            S.t_gather += wall() - tg0;
        }
    }
    double tp2 = tp1;
    if (dm->stage_timing) {
        DM_TRY(hipStreamSynchronize(st));
        tp2 = wall();
        S.t_kernel += (sharded ? tg0 : tp2) - tp1;
    }
    // f3: write-back + prune
    // The reference prunes after ALL test blocks have been predicted (bgkoctomap.cpp:344-353): with repeated keys
    // the later passes must still see the un-pruned leaves, so the prune of pass 0 (which holds every distinct
    // test block) is deferred to the end of the pass loop.  A single-pass scan (the usual case) commits and prunes in ONE
    // launch, whose last workgroup also sends the counter block to the host (dm_commit_prune).
    bool reset_queued = false;   // dm_commit_prune leaves the counter block ready for the next insert
    if (max_occ == 1) {
        volatile uint32_t *mailbox = nullptr;
        uint32_t mseq = 0;
        publish_with(dm, mailbox, mseq);
        if (d3)
            launch_commit_prune_d3(d3, st, (const uint32_t *)dm->t_slot.ptr, n_test, (const uint32_t *)leaf_off, (const uint32_t *)dm->leaf_node.ptr,
                                   own_emit ? (const uint32_t *)dm->leaf_key.ptr : (const uint32_t *)nullptr, (const float *)dm->leaf_alpha.ptr,
                                   (const float *)dm->leaf_beta.ptr, (const uint8_t *)dm->leaf_state.ptr, dm->A, dm->B, dm->S, dm->d_cnt,
                                   dm->d_mm + kArriveBase, mailbox, mseq);
        else
            hipLaunchKernelGGL(dm_commit_prune, dim3(std::min(cdiv(n_test, 4), kCommitPruneWgs)), dim3(256), 4 * prune_lds_stride(dm->npb), st,
                           (const uint32_t *)dm->t_slot.ptr, n_test, (const uint32_t *)leaf_off, (const uint32_t *)dm->leaf_node.ptr,
                           own_emit ? (const uint32_t *)dm->leaf_key.ptr : (const uint32_t *)nullptr, (const float *)dm->leaf_alpha.ptr, (const float *)dm->leaf_beta.ptr, (const uint8_t *)dm->leaf_state.ptr,
                           dm->A, dm->B, dm->S, dm->npb, dm->depth, dm->d_cnt, dm->d_mm + kArriveBase, mailbox, mseq);
        reset_queued = mailbox != nullptr;
    } else {
        hipLaunchKernelGGL(dm_commit, dim3(cdiv(max_leaves, 256)), dim3(256), 0, st, dm->d_cnt, (const uint32_t *)dm->leaf_node.ptr,
                           (const float *)dm->leaf_alpha.ptr, (const float *)dm->leaf_beta.ptr,
                           (const uint8_t *)dm->leaf_state.ptr, dm->A, dm->B, dm->S);
        if (pass == 0) {
            DM_RESERVE(dm->t_slot0, 4ull * n_test);
            DM_TRY(hipMemcpyAsync(dm->t_slot0.ptr, dm->t_slot.ptr, 4ull * n_test, hipMemcpyDeviceToDevice, st));
            *n_test0 = n_test;
        }
    }
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    dm->counters_clean = reset_queued;
    dm->n_blocks = dm->h_cnt[kCntBlocks];
    S.voxel_updates += dm->h_cnt[kCntLeaves];
    S.train_reads = (uint64_t)dm->h_cnt[kCntTrainReads] | ((uint64_t)dm->h_cnt[kCntTrainReads + 1] << 32);
    S.pair_evals = (uint64_t)dm->h_cnt[kCntPairEvals] | ((uint64_t)dm->h_cnt[kCntPairEvals + 1] << 32);
    if (dm->stage_timing) S.t_commit += wall() - tp2;
    return LA3DM_OK;
}

// Stages B..G on the training set in dm->xy (partition, passes, deferred prune); t0 = start of the call.
static int scan_training_set(la3dm_devmap *dm, uint32_t flags, double t0, la3dm_devmap_stats *stats_out) {
    hipStream_t st = dm->ctx->stream;
    la3dm_devmap_stats &S = dm->stats;
    int rc;
    bool no_bbox = false;  // a NaN coordinate in the first training point: the reference's bbox is NaN, no block is visited
    for (int a = 0; a < 6 && dm->n_xy; ++a) no_bbox |= dm->h_bbox[a] != dm->h_bbox[a];
    if (dm->n_xy == 0 || no_bbox) {  // empty cloud, every hit beyond max_range, or no candidate blocks
        S.t_total = wall() - t0;
        if (stats_out) *stats_out = S;
        return LA3DM_OK;
    }
    const double t1 = wall();
    S.t_frontend = t1 - t0;
    ScanPlan P;
    P.flags = flags;
    if ((rc = partition(dm, P)) != LA3DM_OK) return rc;
    S.t_partition = wall() - t1;
    DM_RESERVE(dm->c_weight, 4ull * P.n_entries);
    const uint32_t max_occ = P.max_occ;
    uint32_t n_test0 = 0;
    for (uint32_t pass = 0; pass < max_occ; ++pass)
        if ((rc = run_pass(dm, P, pass, &n_test0)) != LA3DM_OK) {
            // run_pass is not failure-atomic: dm_table_insert may already have published new keys and bumped the device
            // block counter when a later step (arena growth, a sort, the scan) fails.  Adopt the device counter so that the
            // table, the pool (the new blocks get their default nodes here if the pass had not written them) and the host agree again — the map stays usable, the
            // scan is lost, as if the reference had thrown after creating its blocks.  If even that fails: poison.
            const std::string why = dm->ctx->err;
            uint32_t dev_blocks = 0;
            if (hipStreamSynchronize(st) == hipSuccess &&
                hipMemcpy(&dev_blocks, dm->d_cnt + kCntBlocks, 4, hipMemcpyDeviceToHost) == hipSuccess && dev_blocks <= dm->cap_blocks) {
                if (dev_blocks > dm->n_blocks) {
                    // the slots [old, new) may not have been initialised yet (their default nodes are written by the leaf-count
                    // launch, which the failed pass may never have reached): write them now, then adopt the count
                    hipLaunchKernelGGL(dm_pool_init, dim3(cdiv((size_t)(dev_blocks - dm->n_blocks) * dm->npb, 256)), dim3(256), 0, st, dm->A,
                                       dm->B, dm->S, dm->n_blocks, (const uint32_t *)(dm->d_cnt + kCntBlocks), dm->npb, dm->init_A, dm->init_B);
                    if (hipStreamSynchronize(st) == hipSuccess)
                        dm->n_blocks = dev_blocks;
                    else
                        dm->poisoned = true;
                }
            } else {
                dm->poisoned = true;
            }
            dm->ctx->err = why;
            return rc;
        }
    if (max_occ > 1 && n_test0)
        hipLaunchKernelGGL(dm_prune, dim3(cdiv(n_test0, 4)), dim3(256), 4 * prune_lds_stride(dm->npb), st,
                           (const uint32_t *)dm->t_slot0.ptr, n_test0, dm->A, dm->B, dm->S, dm->npb, dm->depth);
    DM_TRY(hipGetLastError());
    if (max_occ > 1) DM_TRY(hipStreamSynchronize(st));  // (single pass: the pass's own read-back was the last sync)
    S.n_blocks = dm->n_blocks;
    S.t_total = wall() - t0;
    if (!dm->stage_timing) S.t_pack = S.t_total - S.t_frontend - S.t_partition;  // pack + kernel + commit, unsplit
    if (stats_out) *stats_out = S;
    return LA3DM_OK;
}

// Start of an insert: the counter block at its initial values — by dm_begin, unless the insert before left it so (round 5).
static void begin_insert(la3dm_devmap *dm) {
    if (dm->counters_clean && dm->mailbox) {
        dm->counters_clean = false;
        return;
    }
    dm->counters_clean = false;
    hipLaunchKernelGGL(dm_begin, dim3(1), dim3(64), 0, dm->ctx->stream, dm->d_cnt, dm->n_blocks, dm->d_mm, dm->d_mm + 6);
}

// BGKOctoMap::insert_training_data (bgkoctomap.cpp:82-212) on the pool: n labelled points {x, y, z, label} (host
// pointer) instead of a scan; every leaf of every test block is updated for every neighbour model (no kbar gate).
int la3dm_devmap_insert_training_data_host(la3dm_devmap *dm, const float *xyzy, uint32_t n, la3dm_devmap_stats *stats_out) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (!dm || (n && !xyzy)) return LA3DM_ERR_ARG;
    la3dm_ctx *ctx = dm->ctx;
    if (ctx->p.variant == 3) return dm_fail(dm, LA3DM_ERR_ARG, "la3dm_devmap_insert_training_data: a BGK-L map needs beams, not labelled points");
    DM_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    la3dm_devmap_stats &S = dm->stats;
    memset(&S, 0, sizeof(S));
    S.n_blocks = dm->n_blocks;
    dm->insert_into_empty = dm->n_blocks == 0;
    dm->n_xy = 0;
    const double t0 = wall();
    begin_insert(dm);
    if (n == 0) {  // bgkoctomap.cpp:83-84
        if (stats_out) *stats_out = S;
        return LA3DM_OK;
    }
    DM_RESERVE(dm->xy, 16ull * n);
    DM_TRY(hipMemcpyAsync(dm->xy.ptr, xyzy, 16ull * n, hipMemcpyHostToDevice, st));
    dm->n_xy = n;
    for (uint32_t i = 0; i < n; ++i) (xyzy[4 * (size_t)i + 3] > 0.5f ? S.n_hits : S.n_frees)++;
    int rc = training_bbox(dm);
    if (rc != LA3DM_OK) return rc;
    return scan_training_set(dm, LA3DM_SCAN_UPDATE_UNGATED, t0, stats_out);
}

// ---- BGKLVOctoMap::insert_pointcloud on the pool (src/bgklvoctomap/bgklvoctomap.cpp:89-285; kernels and the map of
// the stages in devmap_lv_kernels.h).  The pool stores the host's State enum (PRUNED 3, UNCERTAIN 4), so download,
// search and prune are the BGK ones.  Read-backs: hit-grid cell count, beam totals, sample bbox, bucket bounds,
// packed-block count, counters at the end.
static int lv_insert(la3dm_devmap *dm, const float *d_xyz, uint32_t n, const float origin[3], float ds_resolution,
                     float free_resolution, float max_range, double t0) {
    la3dm_ctx *ctx = dm->ctx;
    hipStream_t st = ctx->stream;
    la3dm_devmap_lv_stats &L = dm->lv_stats;
    memset(&L, 0, sizeof(L));
    dm->lv_n_samples = dm->lv_n_rays = 0;
    int rc;
    if (ds_resolution > ctx->p.resolution) ds_resolution = ctx->p.resolution;  // bgklvoctomap.cpp:95-96
    // hits: voxel-grid filter (or the cloud itself)
    const float *d_hits = d_xyz;
    uint32_t nh = n;
    if (!(ds_resolution < 0)) {
        if ((rc = voxel_grid(dm, d_xyz, n, ds_resolution, dm->hits, &nh)) != LA3DM_OK) return rc;
        d_hits = (const float *)dm->hits.ptr;
    }
    if (nh == 0) return LA3DM_OK;
    LvBeamArgs ba;
    ba.ox = origin[0]; ba.oy = origin[1]; ba.oz = origin[2];
    ba.max_range = max_range;
    ba.free_res = free_resolution;
    ba.offset = (double)ctx->p.ell * pow(2, 0.5);
    ba.influence = (double)ctx->p.ell;
    DM_RESERVE(dm->lv_rng, 8ull * nh);
    DM_RESERVE(dm->lv_flags, nh);
    DM_RESERVE(dm->lv_seg, 24ull * nh);
    DM_RESERVE(dm->lv_nsamp, 4ull * nh);
    DM_RESERVE(dm->lv_nray, 4ull * nh);
    DM_RESERVE(dm->lv_samp_off, 4ull * nh);
    DM_RESERVE(dm->lv_ray_off, 4ull * nh);
    uint32_t *nsamp = (uint32_t *)dm->lv_nsamp.ptr, *nray = (uint32_t *)dm->lv_nray.ptr;
    uint32_t *samp_off = (uint32_t *)dm->lv_samp_off.ptr, *ray_off = (uint32_t *)dm->lv_ray_off.ptr;
    hipLaunchKernelGGL(dm_lv_ranges, dim3(cdiv(nh, 256)), dim3(256), 0, st, d_hits, nh, ba, (double *)dm->lv_rng.ptr);
    // ray shortening (bgklvoctomap.cpp:313-423).  Small scans (configs[3]'s 3 500-point clouds): the dense form — membership of
    // the "nearby" gather for all (beam, hit) pairs at once as an nh x nh bit matrix, then the ordered walk over the set bits,
    // hit list in LDS.  Larger scans: the same sets from a uniform grid over the hits, O(N k) in time and memory (round 6).
    // LA3DM_LV_NEAR=dense | grid forces one of them (the tests compare the two).
    const char *near_env = getenv("LA3DM_LV_NEAR");
    const bool force_dense = near_env && !strcmp(near_env, "dense"), force_grid = near_env && !strcmp(near_env, "grid");
    if (force_dense && nh > 98304u)
        return dm_fail(dm, LA3DM_ERR_ARG, "devmap (BGK-LV): LA3DM_LV_NEAR=dense needs nh^2 / 8 bytes of mask: at most 98 304 hits");
    DM_RESERVE(dm->lv_beam, sizeof(LvBeam) * (size_t)nh);
    hipLaunchKernelGGL(dm_lv_beam_init, dim3(cdiv(nh, 256)), dim3(256), 0, st, d_hits, nh, ba, (LvBeam *)dm->lv_beam.ptr);
    static const uint32_t kGridMin = getenv("LA3DM_LV_GRID_MIN") ? (uint32_t)std::max(1, atoi(getenv("LA3DM_LV_GRID_MIN"))) : 8192u;   // (A/B: hits from which the grid form runs)
    if (force_dense || (!force_grid && nh < kGridMin)) {
        const uint32_t nw = cdiv(nh, 64);
        DM_RESERVE(dm->lv_mask, 8ull * nh * nw);
        // (beams per workgroup: small scans need the parallelism — a tile is walked in sequence —, large ones the reuse of the wave's 64 hits)
        const bool small_scan = nh < 16384u;
        const uint32_t near_tile = small_scan ? 32u : kLvNearTile;
        if (small_scan)
            hipLaunchKernelGGL(dm_lv_nearby<true>, dim3(nw, cdiv(nh, near_tile)), dim3(64), 0, st, d_hits, nh, ba, (const double *)dm->lv_rng.ptr,
                               (const LvBeam *)dm->lv_beam.ptr, (unsigned long long *)dm->lv_mask.ptr, near_tile);
        else
            hipLaunchKernelGGL(dm_lv_nearby<false>, dim3(nw, cdiv(nh, near_tile)), dim3(64), 0, st, d_hits, nh, ba, (const double *)dm->lv_rng.ptr,
                               (const LvBeam *)dm->lv_beam.ptr, (unsigned long long *)dm->lv_mask.ptr, near_tile);
        const int lds_hits = 12ull * nh <= 60000 ? 1 : 0;   // the hit list in LDS (2 workgroups per CU still fit)
        const uint32_t walk_threads = lds_hits ? 256u : 64u;   // (the staged hit list is shared by the workgroup's four waves)
        hipLaunchKernelGGL(dm_lv_beams_walk, dim3(cdiv(nh, walk_threads)), dim3(walk_threads), lds_hits ? 12 * nh : 0, st, d_hits, nh, ba,
                           (const LvBeam *)dm->lv_beam.ptr, (const unsigned long long *)dm->lv_mask.ptr, nw, (uint8_t *)dm->lv_flags.ptr,
                           (float *)dm->lv_seg.ptr, nsamp, nray, dm->d_cnt, lds_hits);
    } else {
        const double *rng = (const double *)dm->lv_rng.ptr;
        const LvBeam *beams = (const LvBeam *)dm->lv_beam.ptr;
        // 1. the grid: cells of edge `influence` around the sensor.  With a range gate every hit that can be nearby at all lies within
        //    max_range of the sensor, so the grid's extent is known here; without one, the bounds of the hits come from a reduction and
        //    one counter read-back
        int32_t hmm[7] = {0, 0, 0, 0, 0, 0, 1};
        const double reach = max_range > 0 ? ceil((double)max_range / ba.influence) + 1.0 : 0.0;
        if (max_range > 0 && reach < 1.0e6) {
            hmm[0] = hmm[1] = hmm[2] = -(int32_t)reach;
            hmm[3] = hmm[4] = hmm[5] = (int32_t)reach;
        } else {
            hipLaunchKernelGGL(dm_lv_hit_bounds, dim3(std::min<uint32_t>(cdiv(nh, 256), kMinmaxWgs)), dim3(256), 0, st, d_hits, nh, ba, rng, dm->d_cnt);
            if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
            if (dm->h_cnt[kCntError] & kErrLvExtent) return dm_fail(dm, LA3DM_ERR_ARG, "devmap (BGK-LV): hit coordinates beyond the ray-shortening grid's index range");
            memcpy(hmm, dm->h_cnt + kCntLvHmm, 28);
            if (hmm[6] == 0) hmm[0] = hmm[1] = hmm[2] = hmm[3] = hmm[4] = hmm[5] = 0;   // no hit in range: one empty cell
        }
        LvHitGrid G;
        uint64_t ncell = 0;
        for (int m = 1;; m *= 2) {   // coarser cells while the dense grid would need more than 2^24 of them
            ncell = 1;
            for (int a = 0; a < 3; ++a) {
                G.dim[a] = (int32_t)(((int64_t)hmm[3 + a] - hmm[a]) / m + 1);
                ncell *= (uint64_t)G.dim[a];
            }
            G.cell = ba.influence * m;
            if (ncell <= (1ull << 24)) break;
        }
        G.g0[0] = (double)ba.ox + (double)hmm[0] * ba.influence;
        G.g0[1] = (double)ba.oy + (double)hmm[1] * ba.influence;
        G.g0[2] = (double)ba.oz + (double)hmm[2] * ba.influence;
        DM_RESERVE(dm->lv_hcell, 4ull * nh);
        DM_RESERVE(dm->lv_hlist, 4ull * nh);
        DM_RESERVE(dm->lv_hcnt, 4ull * (ncell + 1));
        DM_RESERVE(dm->lv_hoff, 4ull * (ncell + 1));
        uint32_t *hcnt = (uint32_t *)dm->lv_hcnt.ptr, *hoff = (uint32_t *)dm->lv_hoff.ptr;
        DM_TRY(hipMemsetAsync(hcnt, 0, 4ull * (ncell + 1), st));
        hipLaunchKernelGGL(dm_lv_hgrid_count, dim3(cdiv(nh, 256)), dim3(256), 0, st, d_hits, nh, ba, rng, G, (uint32_t *)dm->lv_hcell.ptr, hcnt);
        if ((rc = exclusive_scan(dm, hcnt, hoff, (uint32_t)ncell + 1)) != LA3DM_OK) return rc;
        DM_TRY(hipMemsetAsync(hcnt, 0, 4ull * (ncell + 1), st));
        hipLaunchKernelGGL(dm_lv_hgrid_fill, dim3(cdiv(nh, 256)), dim3(256), 0, st, (const uint32_t *)dm->lv_hcell.ptr, nh, (const uint32_t *)hoff, hcnt,
                           (uint32_t *)dm->lv_hlist.ptr);
        // 2. every beam in one wave: capsule walk, nearby hits collected and sorted in LDS, the ordered shortening (dm_lv_beams_grid)
        hipLaunchKernelGGL(dm_lv_beams_grid, dim3(cdiv(nh, kLvGridWaves)), dim3(64 * kLvGridWaves), 0, st, d_hits, nh, ba, rng, beams, G,
                           (const uint32_t *)hoff, (const uint32_t *)dm->lv_hlist.ptr, (uint8_t *)dm->lv_flags.ptr, (float *)dm->lv_seg.ptr, nsamp, nray,
                           dm->d_cnt);
        hipLaunchKernelGGL(dm_lv_beam_totals, dim3(std::min<uint32_t>(cdiv(nh, 256), kMinmaxWgs)), dim3(256), 0, st, (const uint32_t *)nsamp,
                           (const uint8_t *)dm->lv_flags.ptr, nh, dm->d_cnt);
        if (getenv("LA3DM_DEBUG_LV")) fprintf(stderr, "la3dm BGK-LV ray shortening on the hit grid: %u hits, %d x %d x %d cells of %.3f m\n", nh, G.dim[0], G.dim[1], G.dim[2], G.cell);
    }
    if ((rc = exclusive_scan(dm, nsamp, samp_off, nh, (int)kCntFreeRaw)) != LA3DM_OK) return rc;
    if ((rc = exclusive_scan(dm, nray, ray_off, nh, (int)kCntKept, true)) != LA3DM_OK) return rc;
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    if ((rc = check_beam_counters(dm)) != LA3DM_OK) return rc;
    const uint32_t ns = dm->h_cnt[kCntFreeRaw], n_rays = dm->h_cnt[kCntKept];
    L.n_rays = n_rays;
    L.n_samples = ns;
    if (ns == 0) return LA3DM_OK;
    DM_RESERVE(dm->lv_samples, 16ull * ns);
    DM_RESERVE(dm->lv_rays, 32ull * (n_rays ? n_rays : 1));
    float4 *samples = (float4 *)dm->lv_samples.ptr;
    hipLaunchKernelGGL(dm_lv_emit, dim3(cdiv(nh, 256)), dim3(256), 0, st, d_hits, nh, ba, (const uint8_t *)dm->lv_flags.ptr,
                       (const float *)dm->lv_seg.ptr, (const uint32_t *)samp_off, (const uint32_t *)ray_off, samples,
                       (float4 *)dm->lv_rays.ptr);
    dm->lv_n_samples = ns;
    dm->lv_n_rays = n_rays;
    L.n_hits = dm->h_cnt[kCntTrained];
    // bounding box of ALL samples (std::min / std::max chains from samples[0]: bgklvoctomap.cpp:105-112)
    {
        MinmaxFin fin = {2, 0.0f, nullptr, (const float *)samples, dm->d_cnt, dm->d_mm + 6, nullptr};
        hipLaunchKernelGGL(dm_minmax<4>, dim3(std::min<uint32_t>(cdiv(ns, 1024), kMinmaxWgs)), dim3(256), 0, st, (const float *)samples, ns, dm->d_mm, fin);
    }
    // bucket bounds of the finite samples
    const int depth = ctx->p.block_depth;
    const float bs = dm->block_size;
    const double g = depth >= 3 ? 4.0 * (double)ctx->p.resolution : (double)bs;
    const double half = 0.5 * (double)bs;
    // (the bounds live in the counter block — initialised by dm_begin, read back with it)
    hipLaunchKernelGGL(dm_lv_cell_bounds, dim3(std::min<uint32_t>(cdiv(ns, 256), kMinmaxWgs)), dim3(256), 0, st, (const float4 *)samples, ns, half, g,
                       (int32_t *)(dm->d_cnt + kCntLvmm), dm->d_cnt);
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    memcpy(dm->h_bbox, dm->h_cnt + kCntBbox, sizeof(float) * 6);
    memcpy(dm->h_lvmm, dm->h_cnt + kCntLvmm, 28);
    if (dm->h_cnt[kCntError] & kErrLvExtent) return dm_fail(dm, LA3DM_ERR_ARG, "devmap (BGK-LV): sample coordinates beyond the gather grid's index range");
    for (int a = 0; a < 6; ++a)
        if (dm->h_bbox[a] != dm->h_bbox[a]) return LA3DM_OK;  // NaN box (first sample not finite): no candidate block, as on the host
    const uint32_t n_binned = (uint32_t)dm->h_lvmm[6];
    if (n_binned == 0) return LA3DM_OK;
    LvGridArgs ga;
    ga.g = g;
    ga.half = half;
    size_t ncell = 1;
    for (int a = 0; a < 3; ++a) {
        ga.cmin[a] = dm->h_lvmm[a];
        ga.cdim[a] = dm->h_lvmm[3 + a] - dm->h_lvmm[a] + 1;
        ncell *= (size_t)ga.cdim[a];
    }
    if (ncell > ((size_t)1 << 28)) return dm_fail(dm, LA3DM_ERR_ARG, "devmap (BGK-LV): scan extent too large for the dense gather grid");
    // candidate blocks: per-axis float-stepped sequences (host, a few dozen values), distinct indices + multiplicities
    std::vector<int32_t> ax_idx[3];
    std::vector<uint32_t> ax_mult[3];
    uint32_t max_mult = 1;
    size_t n_cand = 1, n_bbox = 1;
    for (int a = 0; a < 3; ++a) {
        std::vector<int> seq = axis_sequence(dm->h_bbox[a], dm->h_bbox[3 + a], bs);
        if (seq.empty() || seq.size() > (1u << 20)) return dm_fail(dm, LA3DM_ERR_ARG, "devmap: degenerate training-set extent");
        n_bbox *= seq.size();
        std::sort(seq.begin(), seq.end());
        uint32_t mm = 1;
        for (size_t i = 0; i < seq.size();) {
            size_t j = i;
            while (j < seq.size() && seq[j] == seq[i]) ++j;
            ax_idx[a].push_back(seq[i]);
            ax_mult[a].push_back((uint32_t)(j - i));
            mm = std::max(mm, (uint32_t)(j - i));
            i = j;
        }
        max_mult *= mm;
        n_cand *= ax_idx[a].size();
    }
    if (n_cand > (1u << 26)) return dm_fail(dm, LA3DM_ERR_ARG, "devmap (BGK-LV): more than 2^26 candidate blocks (set max_range)");
    L.n_bbox_blocks = n_bbox;
    // gather grid: stable sort of the samples by bucket, CSR over the dense grid
    DM_RESERVE(dm->k0, 4ull * ns);
    DM_RESERVE(dm->k1, 4ull * ns);
    DM_RESERVE(dm->v0, 4ull * ns);
    DM_RESERVE(dm->v1, 4ull * ns);
    DM_RESERVE(dm->lv_sorted, 16ull * ns);
    DM_RESERVE(dm->lv_cell_off, 4ull * (ncell + 1));
    uint32_t *k0 = (uint32_t *)dm->k0.ptr, *k1 = (uint32_t *)dm->k1.ptr, *v0 = (uint32_t *)dm->v0.ptr, *v1 = (uint32_t *)dm->v1.ptr;
    hipLaunchKernelGGL(dm_lv_cell_keys, dim3(cdiv(ns, 256)), dim3(256), 0, st, (const float4 *)samples, ns, ga, (uint32_t)ncell, k0, v0);
    int key_bits = 1;
    while (((size_t)1 << key_bits) <= ncell) ++key_bits;
    if ((rc = sort_pairs(dm, k0, k1, v0, v1, ns, key_bits)) != LA3DM_OK) return rc;
    hipLaunchKernelGGL(dm_lv_sorted, dim3(cdiv(ns, 256)), dim3(256), 0, st, (const float4 *)samples, (const uint32_t *)v1, ns, (float4 *)dm->lv_sorted.ptr);
    hipLaunchKernelGGL(dm_lv_cell_off, dim3(cdiv((uint32_t)ncell + 1, 256)), dim3(256), 0, st, (const uint32_t *)k1, ns, (uint32_t)ncell,
                       (uint32_t *)dm->lv_cell_off.ptr);
    // candidates: keys (ascending), multiplicities, "has a sample within reach"; every candidate block is created
    {
        size_t bytes = 0;
        for (int a = 0; a < 3; ++a) bytes += 8 * ax_idx[a].size();
        DM_RESERVE(dm->lv_axis, bytes);
    }
    LvCandArgs ca;
    {
        // one upload from a staging vector that lives as long as the map (no synchronisation for the host buffers' sake)
        size_t bytes = 0;
        for (int a = 0; a < 3; ++a) bytes += 8 * ax_idx[a].size();
        dm->lv_axis_host.resize(bytes);
        uint8_t *base = (uint8_t *)dm->lv_axis.ptr, *hb = dm->lv_axis_host.data();
        size_t o = 0;
        for (int a = 0; a < 3; ++a) {
            const size_t m = ax_idx[a].size();
            memcpy(hb + o, ax_idx[a].data(), 4 * m);
            ca.idx[a] = (const int32_t *)(base + o);
            o += 4 * m;
            memcpy(hb + o, ax_mult[a].data(), 4 * m);
            ca.mult[a] = (const uint32_t *)(base + o);
            o += 4 * m;
            ca.n[a] = (uint32_t)m;
            ca.cmin[a] = ga.cmin[a];
            ca.cdim[a] = ga.cdim[a];
        }
        DM_TRY(hipMemcpyAsync(base, hb, bytes, hipMemcpyHostToDevice, st));
    }
    ca.block_size = bs;
    ca.g = g;
    ca.reach = (int32_t)std::ceil((double)ctx->p.ell / g);
    ca.bpb = depth >= 3 ? (1 << (depth - 3)) : 1;
    const uint32_t nc = (uint32_t)n_cand;
    DM_RESERVE(dm->lv_keys, 8ull * nc);
    DM_RESERVE(dm->lv_mult, 4ull * nc);
    DM_RESERVE(dm->lv_flag, 4ull * nc);
    DM_RESERVE(dm->lv_pos, 4ull * nc);
    DM_RESERVE(dm->lv_slot, 4ull * nc);
    hipLaunchKernelGGL(dm_lv_candidates, dim3(cdiv(nc, 256)), dim3(256), 0, st, ca, nc, (const uint32_t *)dm->lv_cell_off.ptr,
                       (long long *)dm->lv_keys.ptr, (uint32_t *)dm->lv_mult.ptr, (uint32_t *)dm->lv_flag.ptr, dm->d_cnt);
    if ((rc = grow_pool(dm, (size_t)dm->n_blocks + nc)) != LA3DM_OK) return rc;
    if ((rc = grow_table(dm, (size_t)dm->n_blocks + nc)) != LA3DM_OK) return rc;
    hipLaunchKernelGGL(dm_table_insert, dim3(cdiv(nc, 256)), dim3(256), 0, st, (const long long *)dm->lv_keys.ptr, dm->d_cnt, dm->tab_key,
                       dm->tab_val, dm->tab_cap - 1, dm->d_cnt + kCntBlocks, dm->blk_key, (uint32_t *)dm->lv_slot.ptr);
    hipLaunchKernelGGL(dm_pool_init, dim3(cdiv((size_t)nc * dm->npb, 256)), dim3(256), 0, st, dm->A, dm->B, dm->S, dm->n_blocks,
                       (const uint32_t *)(dm->d_cnt + kCntBlocks), dm->npb, dm->init_A, dm->init_B);
    if ((rc = exclusive_scan(dm, (const uint32_t *)dm->lv_flag.ptr, (uint32_t *)dm->lv_pos.ptr, nc)) != LA3DM_OK) return rc;
    DM_RESERVE(dm->lv_center, 12ull * nc);
    DM_RESERVE(dm->lv_cell0, 12ull * nc);
    DM_RESERVE(dm->lv_pslot, 4ull * nc);
    DM_RESERVE(dm->lv_pmult, 4ull * nc);
    DM_RESERVE(dm->lv_info, 4ull * nc);
    DM_RESERVE(dm->lv_prune, 4ull * nc);
    hipLaunchKernelGGL(dm_lv_pack, dim3(cdiv(nc, 256)), dim3(256), 0, st, (const long long *)dm->lv_keys.ptr, (const uint32_t *)dm->lv_mult.ptr,
                       (const uint32_t *)dm->lv_flag.ptr, (const uint32_t *)dm->lv_pos.ptr, (const uint32_t *)dm->lv_slot.ptr, nc, bs, g,
                       (float *)dm->lv_center.ptr, (int32_t *)dm->lv_cell0.ptr, (uint32_t *)dm->lv_pslot.ptr, (uint32_t *)dm->lv_pmult.ptr,
                       dm->d_cnt, (uint32_t *)dm->lv_info.ptr);
    // (d_cnt[kCntLeaves] counts the voxel updates of all passes, d_cnt[kCntGeo] the blocks with information: zero since
    // dm_begin, nothing before this point of a BGK-LV insert touches them)
    // work plan of the voxel kernel for at most nc packed blocks (the kernel itself stops at the packed count, which is
    // still on the device): its totals come back with the counters below
    la3dm_lv_pool_scan ps;
    memset(&ps, 0, sizeof(ps));
    ps.n_samples = ns;
    ps.samples = (const float *)samples;
    ps.sorted = (const float *)dm->lv_sorted.ptr;
    ps.rays = (const float *)dm->lv_rays.ptr;
    ps.cell_off = (const uint32_t *)dm->lv_cell_off.ptr;
    for (int a = 0; a < 3; ++a) {
        ps.cell_min[a] = ga.cmin[a];
        ps.cell_dim[a] = ga.cdim[a];
    }
    ps.blk_center = (const float *)dm->lv_center.ptr;
    ps.blk_cell0 = (const int32_t *)dm->lv_cell0.ptr;
    ps.blk_slot = (const uint32_t *)dm->lv_pslot.ptr;
    ps.blk_mult = (const uint32_t *)dm->lv_pmult.ptr;
    ps.A = dm->A;
    ps.B = dm->B;
    ps.S = dm->S;
    ps.npb = dm->npb;
    ps.upd_counter = dm->d_cnt + kCntLeaves;
    ps.n_blk = ps.plan_n_blk = nc;
    ps.n_blk_dev = dm->d_cnt + kCntTest;
    if ((rc = la3dm_bgklv_pool_plan_device(ctx, &ps, dm->d_cnt + kCntLvPlan, st)) != LA3DM_OK) return rc;
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    dm->n_blocks = dm->h_cnt[kCntBlocks];
    const uint32_t n_packed = dm->h_cnt[kCntTest];
    L.n_packed_blocks = n_packed;
    L.voxels = (uint64_t)n_packed << (3 * (depth - 1));
    const double t1 = wall();
    if (n_packed) {
        ps.n_blk = n_packed;
        for (int a = 0; a < 4; ++a) ps.plan_totals[a] = dm->h_cnt[kCntLvPlan + a];
        ps.plan_totals_dev = dm->d_cnt + kCntLvPlan;
        const uint32_t layer_n = 1u << (3 * (depth - 1)), layer_off = dm->npb - layer_n;
        for (uint32_t pass = 0; pass < max_mult; ++pass) {  // a key the float-stepped loop repeats is visited again, serially
            ps.pass = pass;
            if ((rc = la3dm_bgklv_pool_scan_device(ctx, &ps, st)) != LA3DM_OK) return rc;
            hipLaunchKernelGGL(dm_lv_finish, dim3(cdiv(n_packed, 4)), dim3(256), 0, st, (const uint32_t *)dm->lv_pslot.ptr,
                               (const uint32_t *)dm->lv_pmult.ptr, n_packed, pass, dm->S, dm->npb, layer_off, layer_n,
                               (uint32_t *)dm->lv_info.ptr);
        }
        hipLaunchKernelGGL(dm_lv_prune_list, dim3(cdiv(n_packed, 256)), dim3(256), 0, st, (const uint32_t *)dm->lv_pslot.ptr,
                           (const uint32_t *)dm->lv_info.ptr, n_packed, (uint32_t *)dm->lv_prune.ptr, dm->d_cnt);
        if (dm->lv_original_size)
            hipLaunchKernelGGL(dm_prune, dim3(cdiv(n_packed, 4)), dim3(256), 4 * prune_lds_stride(dm->npb), st,
                               (const uint32_t *)dm->lv_prune.ptr, n_packed, dm->A, dm->B, dm->S, dm->npb, dm->depth);
        DM_TRY(hipGetLastError());
        if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
        L.voxel_updates = dm->h_cnt[kCntLeaves];
        L.n_info_blocks = dm->h_cnt[kCntGeo];
    }
    L.n_blocks = dm->n_blocks;
    L.t_frontend = t1 - t0;
    L.t_total = wall() - t0;
    return LA3DM_OK;
}

// Argument checks shared by the insert entry points.  The beam sampler walks `for (d = fr; d < l; d += fr)` on the GPU
// (bgkoctomap.cpp:445-457): a free_resolution that is not a positive finite number would never terminate there (the
// reference would run out of memory on the host instead), so it is rejected here; so are NaN resolutions / ranges and a
// non-finite sensor origin.  Non-finite points of the cloud itself are dropped by the front end's range gate.
static int check_scan_args(la3dm_devmap *dm, const float origin[3], float ds_resolution, float free_resolution, float max_range) {
    if (dm->poisoned)
        return dm_fail(dm, LA3DM_ERR_HIP, "devmap: an earlier insert failed in a way that left the block table unusable; destroy the map");
    if (!(free_resolution > 0.0f) || !std::isfinite(free_resolution))
        return dm_fail(dm, LA3DM_ERR_ARG, "insert_pointcloud: free_resolution must be a positive finite number");
    if (ds_resolution != ds_resolution || std::isinf(ds_resolution) || ds_resolution == 0.0f)
        return dm_fail(dm, LA3DM_ERR_ARG, "insert_pointcloud: ds_resolution must be finite and non-zero (negative = no voxel filter)");
    if (max_range != max_range)
        return dm_fail(dm, LA3DM_ERR_ARG, "insert_pointcloud: max_range is NaN");
    for (int i = 0; i < 3; ++i)
        if (!std::isfinite(origin[i])) return dm_fail(dm, LA3DM_ERR_ARG, "insert_pointcloud: the sensor origin is not finite");
    return LA3DM_OK;
}

// after a failed BGK-LV insert: the candidate blocks may already be in the table (see scan_training_set)
static void lv_recover(la3dm_devmap *dm) {
    const std::string why = dm->ctx->err;
    uint32_t dev_blocks = 0;
    if (hipStreamSynchronize(dm->ctx->stream) == hipSuccess &&
        hipMemcpy(&dev_blocks, dm->d_cnt + kCntBlocks, 4, hipMemcpyDeviceToHost) == hipSuccess && dev_blocks <= dm->cap_blocks) {
        if (dev_blocks > dm->n_blocks) dm->n_blocks = dev_blocks;
    } else {
        dm->poisoned = true;
    }
    dm->ctx->err = why;
}

int la3dm_devmap_lv_stats_get(la3dm_devmap *dm, la3dm_devmap_lv_stats *out) {
    if (!dm || !out) return LA3DM_ERR_ARG;
    *out = dm->lv_stats;
    return LA3DM_OK;
}

int la3dm_devmap_lv_set_original_size(la3dm_devmap *dm, int original_size) {
    if (!dm) return LA3DM_ERR_ARG;
    dm->lv_original_size = original_size != 0;
    return LA3DM_OK;
}

int la3dm_devmap_lv_training(la3dm_devmap *dm, float *samples4, uint32_t cap_samples, float *rays6, uint32_t cap_rays,
                             uint32_t *n_samples, uint32_t *n_rays) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (!dm) return LA3DM_ERR_ARG;
    if (n_samples) *n_samples = dm->lv_n_samples;
    if (n_rays) *n_rays = dm->lv_n_rays;
    if (!samples4 && !rays6) return LA3DM_OK;
    if ((samples4 && cap_samples < dm->lv_n_samples) || (rays6 && cap_rays < dm->lv_n_rays)) return dm_fail(dm, LA3DM_ERR_ARG, "la3dm_devmap_lv_training: buffer too small");
    DM_TRY(hipSetDevice(dm->ctx->device));
    hipStream_t st = dm->ctx->stream;
    if (samples4 && dm->lv_n_samples)
        DM_TRY(hipMemcpyAsync(samples4, dm->lv_samples.ptr, 16ull * dm->lv_n_samples, hipMemcpyDeviceToHost, st));
    std::vector<float> r8;
    if (rays6 && dm->lv_n_rays) {
        r8.resize(8ull * dm->lv_n_rays);
        DM_TRY(hipMemcpyAsync(r8.data(), dm->lv_rays.ptr, 32ull * dm->lv_n_rays, hipMemcpyDeviceToHost, st));
    }
    DM_TRY(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < (rays6 ? dm->lv_n_rays : 0u); ++i) {
        rays6[6 * i] = r8[8 * i]; rays6[6 * i + 1] = r8[8 * i + 1]; rays6[6 * i + 2] = r8[8 * i + 2];
        rays6[6 * i + 3] = r8[8 * i + 4]; rays6[6 * i + 4] = r8[8 * i + 5]; rays6[6 * i + 5] = r8[8 * i + 6];
    }
    return LA3DM_OK;
}

int la3dm_devmap_wait_event(la3dm_devmap *dm, void *event) {
    if (!dm || !event) return LA3DM_ERR_ARG;
    DM_TRY(hipSetDevice(dm->ctx->device));
    DM_TRY(hipStreamWaitEvent(dm->ctx->stream, (hipEvent_t)event, 0));
    return LA3DM_OK;
}

int la3dm_devmap_set_shard(la3dm_devmap *dm, uint32_t rank, uint32_t world, la3dm_allgatherv_fn fn, void *user) {
    if (!dm) return LA3DM_ERR_ARG;
    if (world == 0 || world > 1023 || rank >= world || (world > 1 && !fn))
        return dm_fail(dm, LA3DM_ERR_ARG, "la3dm_devmap_set_shard: need rank < world <= 1023 and a callback when world > 1");
    if (world > 1 && dm->ctx->p.variant != 0 && dm->ctx->p.variant != 1)
        return dm_fail(dm, LA3DM_ERR_ARG, "la3dm_devmap_set_shard: block sharding is built for BGKOctoMap and GPOctoMap contexts");
    DM_TRY(hipSetDevice(dm->ctx->device));
    if (dm->h_shard) {
        (void)hipHostFree(dm->h_shard);
        dm->h_shard = nullptr;
    }
    if (world > 1) {
        DM_TRY(hipHostMalloc((void **)&dm->h_shard, 8ull * (world + 1)));
        // the status / count exchange buffer of every insert, every slot preset to "failed"
        int rc = arena_reserve(dm->ctx, dm->shard_cnt, 4ull * world);
        if (rc != LA3DM_OK) return dm_fail(dm, rc, "la3dm_devmap_set_shard: " + dm->ctx->err);
        DM_TRY(hipMemset(dm->shard_cnt.ptr, 0xFF, 4ull * world));
    }
    dm->shard_rank = rank;
    dm->shard_world = world;
    dm->shard_fn = world > 1 ? fn : nullptr;
    dm->shard_user = user;
    return LA3DM_OK;
}

int la3dm_devmap_insert_pointcloud_device(la3dm_devmap *dm, const float *d_xyz, uint32_t n, const float origin[3],
                                          float ds_resolution, float free_resolution, float max_range,
                                          la3dm_devmap_stats *stats_out) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (!dm || !origin || (n && !d_xyz)) return LA3DM_ERR_ARG;
    la3dm_ctx *ctx = dm->ctx;
    int rc;
    if ((rc = check_scan_args(dm, origin, ds_resolution, free_resolution, max_range)) != LA3DM_OK) return rc;
    DM_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    la3dm_devmap_stats &S = dm->stats;
    memset(&S, 0, sizeof(S));
    S.n_blocks = dm->n_blocks;
    dm->insert_into_empty = dm->n_blocks == 0;
    dm->n_xy = 0;
    const double t0 = wall();
    begin_insert(dm);
    if (ctx->p.variant == 2) {
        rc = lv_insert(dm, d_xyz, n, origin, ds_resolution, free_resolution, max_range, t0);
        if (rc != LA3DM_OK) lv_recover(dm);
        S.n_blocks = dm->n_blocks;
        if (stats_out) *stats_out = S;
        return rc;
    }

    if ((rc = front_end(dm, d_xyz, n, origin, ds_resolution, free_resolution, max_range)) != LA3DM_OK) return rc;
    return scan_training_set(dm, LA3DM_SCAN_LABELS_01, t0, stats_out);  // the front end labels hits 1.0f and free samples 0.0f
}

int la3dm_devmap_insert_pointcloud_host(la3dm_devmap *dm, const float *xyz, uint32_t n, uint32_t stride, const float origin[3],
                                        float ds_resolution, float free_resolution, float max_range,
                                        la3dm_devmap_stats *stats_out) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (!dm || (n && !xyz) || stride < 3) return LA3DM_ERR_ARG;
    DM_TRY(hipSetDevice(dm->ctx->device));
    DM_RESERVE(dm->cloud, 12ull * n);
    if (n) {
        if (stride == 3) {
            DM_TRY(hipMemcpyAsync(dm->cloud.ptr, xyz, 12ull * n, hipMemcpyHostToDevice, dm->ctx->stream));
        } else {
            DM_TRY(hipMemcpy2DAsync(dm->cloud.ptr, 12, xyz, 4ull * stride, 12, n, hipMemcpyHostToDevice, dm->ctx->stream));
        }
    }
    return la3dm_devmap_insert_pointcloud_device(dm, (const float *)dm->cloud.ptr, n, origin, ds_resolution, free_resolution,
                                                 max_range, stats_out);
}

int la3dm_devmap_block_count(la3dm_devmap *dm, uint32_t *n_blocks, uint32_t *nodes_per_block) {
    if (!dm) return LA3DM_ERR_ARG;
    if (n_blocks) *n_blocks = dm->n_blocks;
    if (nodes_per_block) *nodes_per_block = dm->npb;
    return LA3DM_OK;
}

int la3dm_devmap_download(la3dm_devmap *dm, int64_t *keys, float *A, float *B, uint8_t *S) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (!dm) return LA3DM_ERR_ARG;
    if (dm->n_blocks == 0) return LA3DM_OK;
    if (!keys || !A || !B || !S) return LA3DM_ERR_ARG;
    DM_TRY(hipSetDevice(dm->ctx->device));
    hipStream_t st = dm->ctx->stream;
    const size_t nn = (size_t)dm->n_blocks * dm->npb;
    DM_TRY(hipMemcpyAsync(keys, dm->blk_key, 8ull * dm->n_blocks, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(A, dm->A, 4 * nn, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(B, dm->B, 4 * nn, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(S, dm->S, nn, hipMemcpyDeviceToHost, st));
    DM_TRY(hipStreamSynchronize(st));
    return LA3DM_OK;
}

int la3dm_devmap_search_host(la3dm_devmap *dm, const float *xyz, uint32_t n, uint8_t *exists, float *A, float *B,
                             uint8_t *state) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (!dm || (n && (!xyz || !exists || !A || !B || !state))) return LA3DM_ERR_ARG;
    if (n == 0) return LA3DM_OK;
    DM_TRY(hipSetDevice(dm->ctx->device));
    hipStream_t st = dm->ctx->stream;
    if (dm->n_blocks == 0) {  // empty map: every query misses
        for (uint32_t i = 0; i < n; ++i) {
            exists[i] = 0;
            A[i] = dm->init_A;
            B[i] = dm->init_B;
            state[i] = kStateUnknown;
        }
        return LA3DM_OK;
    }
    DM_RESERVE(dm->cloud, 12ull * n);
    DM_RESERVE(dm->q_out, 10ull * n + 16);
    float *qA = (float *)dm->q_out.ptr, *qB = qA + n;
    uint8_t *qe = (uint8_t *)(qB + n), *qs = qe + n;
    DM_TRY(hipMemcpyAsync(dm->cloud.ptr, xyz, 12ull * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dm_search, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float *)dm->cloud.ptr, n, dm->tab_key, dm->tab_val,
                       dm->tab_cap - 1, dm->A, dm->B, dm->S, dm->npb, dm->depth, dm->block_size, dm->ctx->p.resolution,
                       dm->init_A, dm->init_B, qe, qA, qB, qs);
    DM_TRY(hipMemcpyAsync(A, qA, 4ull * n, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(B, qB, 4ull * n, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(exists, qe, n, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(state, qs, n, hipMemcpyDeviceToHost, st));
    DM_TRY(hipStreamSynchronize(st));
    return LA3DM_OK;
}

int la3dm_devmap_key_bounds(la3dm_devmap *dm, int32_t lo[3], int32_t hi[3]) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (!dm || !lo || !hi) return LA3DM_ERR_ARG;
    if (dm->n_blocks == 0) return dm_fail(dm, LA3DM_ERR_ARG, "la3dm_devmap_key_bounds: the map holds no blocks");
    DM_TRY(hipSetDevice(dm->ctx->device));
    hipStream_t st = dm->ctx->stream;
    DM_RESERVE(dm->q_out, 64);
    uint32_t *mm = (uint32_t *)dm->q_out.ptr;
    DM_TRY(hipMemsetAsync(mm, 0xFF, 12, st));
    DM_TRY(hipMemsetAsync(mm + 3, 0, 12, st));
    hipLaunchKernelGGL(dm_key_bounds, dim3(std::min(cdiv(dm->n_blocks, 256), 64u)), dim3(256), 0, st, dm->blk_key, dm->n_blocks, mm);
    uint32_t h[6];
    DM_TRY(hipMemcpyAsync(h, mm, 24, hipMemcpyDeviceToHost, st));
    DM_TRY(hipStreamSynchronize(st));
    for (int a = 0; a < 3; ++a) {
        lo[a] = (int32_t)h[a];
        hi[a] = (int32_t)h[3 + a];
    }
    return LA3DM_OK;
}

int la3dm_devmap_export_cells(la3dm_devmap *dm, int state, int original_size, float min_z, float max_z, float *cells,
                              float *rgba, int32_t *level, uint64_t cap, uint64_t *count) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (dm) dm->counters_clean = false;   // (this entry point's scans / sorts write counter slots; only an insert's own last launch leaves the block clean)
    if (!dm || !count || (state != 0 && state != 1)) return LA3DM_ERR_ARG;
    *count = 0;
    if (dm->n_blocks == 0) return LA3DM_OK;
    DM_TRY(hipSetDevice(dm->ctx->device));
    hipStream_t st = dm->ctx->stream;
    const la3dm_params &p = dm->ctx->p;
    ExportArgs a;
    memset(&a, 0, sizeof(a));
    a.blk_key = dm->blk_key;
    a.S = dm->S;
    a.A = dm->A;
    a.B = dm->B;
    a.lut = dm->ctx->d_lut;
    a.n_blocks = dm->n_blocks;
    a.npb = dm->npb;
    a.depth = dm->depth;
    a.want_state = state;
    a.original = original_size ? 1 : 0;
    a.variant = p.variant;
    a.coloured = min_z < max_z ? 1 : 0;
    a.block_size = dm->block_size;
    a.resolution = p.resolution;
    a.min_z = min_z;
    a.max_z = max_z;
    a.gp_l = p.l;
    a.gp_max_ivar = p.max_ivar;
    for (uint32_t d = 0; d < dm->depth && d < 8; ++d) {
        a.size_of_depth[d] = float(dm->block_size / pow(2, d));                       // Block::get_size, bgkblock.h:69-73
        a.level_of_depth[d] = (int)log2(a.size_of_depth[d] / p.resolution);            // markerarray_pub.h:112-114
    }
    DM_RESERVE(dm->c_scan, 8ull * (dm->n_blocks + 1));
    uint32_t *cnt = (uint32_t *)dm->c_scan.ptr, *off = cnt + dm->n_blocks + 1;
    a.blk_cnt = cnt;
    a.blk_off = off;
    DM_TRY(hipMemsetAsync(cnt + dm->n_blocks, 0, 4, st));
    hipLaunchKernelGGL(dm_export<false>, dim3(cdiv(dm->n_blocks, 4)), dim3(256), 0, st, a);
    int rc = exclusive_scan(dm, cnt, off, dm->n_blocks + 1);
    if (rc != LA3DM_OK) return rc;
    uint32_t total = 0;
    DM_TRY(hipMemcpyAsync(&total, off + dm->n_blocks, 4, hipMemcpyDeviceToHost, st));
    DM_TRY(hipStreamSynchronize(st));
    *count = total;
    if (!cells && !rgba && !level) return LA3DM_OK;  // size query
    if (!cells || !rgba || !level || cap < total) return dm_fail(dm, LA3DM_ERR_ARG, "la3dm_devmap_export_cells: buffers too small");
    if (total == 0) return LA3DM_OK;
    DM_RESERVE(dm->q_out, 36ull * total + 64);
    a.cells = (float4 *)dm->q_out.ptr;
    a.rgba = a.cells + total;
    a.level = (int32_t *)(a.rgba + total);
    hipLaunchKernelGGL(dm_export<true>, dim3(cdiv(dm->n_blocks, 4)), dim3(256), 0, st, a);
    DM_TRY(hipGetLastError());
    DM_TRY(hipMemcpyAsync(cells, a.cells, 16ull * total, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(rgba, a.rgba, 16ull * total, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(level, a.level, 4ull * total, hipMemcpyDeviceToHost, st));
    DM_TRY(hipStreamSynchronize(st));
    return LA3DM_OK;
}

int la3dm_devmap_diag_add_repeat(la3dm_ctx *ctx, const float *s, const float *x, const uint32_t *m, uint32_t n,
                                 float *out_fast, float *out_loop) {
    if (!ctx || !s || !x || !m || !out_fast || !out_loop) return LA3DM_ERR_ARG;
    if (n == 0) return LA3DM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, 20ull * n));
    float *ds = (float *)d, *dx = ds + n, *df = dx + n, *dl = df + n;
    uint32_t *dmm = (uint32_t *)(dl + n);
    hipError_t e = hipMemcpy(ds, s, 4ull * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dx, x, 4ull * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dmm, m, 4ull * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dm_diag_add_repeat, dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, ds, dx, dmm, n, df, dl);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out_fast, df, 4ull * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_loop, dl, 4ull * n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) {
        ctx->err = std::string("la3dm_devmap_diag_add_repeat: ") + hipGetErrorString(e);
        return LA3DM_ERR_HIP;
    }
    return LA3DM_OK;
}

// Test hooks for the front end's own primitives (devmap_scan.h, devmap_sort.h), on the map's stream and state.
// mode 0: out[i] = sum of in[0..i), aux[0] = total.  mode 1: in = sorted keys (0xFFFFFFFF = invalid, last): out = exclusive
// scan of the head flags, aux = {segments, valid keys, seg_start[0..segments]} (aux holds n + 3 words).
int la3dm_devmap_diag_scan(la3dm_devmap *dm, int mode, const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *aux) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (dm) dm->counters_clean = false;   // (this entry point's scans / sorts write counter slots; only an insert's own last launch leaves the block clean)
    if (!dm || !in || !out || !aux || n == 0 || (mode != 0 && mode != 1)) return LA3DM_ERR_ARG;
    DM_TRY(hipSetDevice(dm->ctx->device));
    hipStream_t st = dm->ctx->stream;
    DM_RESERVE(dm->k0, 4ull * n);
    DM_RESERVE(dm->k1, 4ull * n);
    DM_RESERVE(dm->seg_start, 4ull * (n + 1));
    DM_TRY(hipMemcpyAsync(dm->k0.ptr, in, 4ull * n, hipMemcpyHostToDevice, st));
    int rc;
    // (publishing launches, as in insert_pointcloud: the mailbox may arrive before the copy below has finished — hence the
    // synchronisation after read_counters)
    if (mode == 0) rc = exclusive_scan(dm, (const uint32_t *)dm->k0.ptr, (uint32_t *)dm->k1.ptr, n, (int)kCntMembers, true);
    else rc = scan_heads(dm, (const uint32_t *)dm->k0.ptr, n, nullptr, (uint32_t *)dm->k1.ptr, (uint32_t *)dm->seg_start.ptr, nullptr,
                         (int)kCntGridSegs, (int)kCntGridValid, -1, true);
    if (rc != LA3DM_OK) return rc;
    DM_TRY(hipMemcpyAsync(out, dm->k1.ptr, 4ull * n, hipMemcpyDeviceToHost, st));
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;
    DM_TRY(hipStreamSynchronize(st));
    if (mode == 0) {
        aux[0] = dm->h_cnt[kCntMembers];
    } else {
        aux[0] = dm->h_cnt[kCntGridSegs];
        aux[1] = dm->h_cnt[kCntGridValid];
        DM_TRY(hipMemcpy(aux + 2, dm->seg_start.ptr, 4ull * (aux[0] + 1), hipMemcpyDeviceToHost));
    }
    return LA3DM_OK;
}

// stable sort of (keys, vals) on the low `bits` key bits through sort_pairs (the in-house radix sort unless LA3DM_OWN_SORT=0)
int la3dm_devmap_diag_sort(la3dm_devmap *dm, const uint32_t *keys, const uint32_t *vals, uint32_t n, int bits, uint32_t *keys_out,
                           uint32_t *vals_out) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (dm) dm->counters_clean = false;   // (this entry point's scans / sorts write counter slots; only an insert's own last launch leaves the block clean)
    if (!dm || !keys || !vals || !keys_out || !vals_out || n == 0 || bits < 1 || bits > 32) return LA3DM_ERR_ARG;
    DM_TRY(hipSetDevice(dm->ctx->device));
    hipStream_t st = dm->ctx->stream;
    DM_RESERVE(dm->k0, 4ull * n);
    DM_RESERVE(dm->k1, 4ull * n);
    DM_RESERVE(dm->v0, 4ull * n);
    DM_RESERVE(dm->v1, 4ull * n);
    DM_TRY(hipMemcpyAsync(dm->k0.ptr, keys, 4ull * n, hipMemcpyHostToDevice, st));
    DM_TRY(hipMemcpyAsync(dm->v0.ptr, vals, 4ull * n, hipMemcpyHostToDevice, st));
    int rc = sort_pairs(dm, (const uint32_t *)dm->k0.ptr, (uint32_t *)dm->k1.ptr, (const uint32_t *)dm->v0.ptr, (uint32_t *)dm->v1.ptr, n, bits);
    if (rc != LA3DM_OK) return rc;
    DM_TRY(hipMemcpyAsync(keys_out, dm->k1.ptr, 4ull * n, hipMemcpyDeviceToHost, st));
    DM_TRY(hipMemcpyAsync(vals_out, dm->v1.ptr, 4ull * n, hipMemcpyDeviceToHost, st));
    if ((rc = read_counters(dm)) != LA3DM_OK) return rc;   // (carries the "stuck" error bit of the look-back loops)
    DM_TRY(hipStreamSynchronize(st));
#ifdef LA3DM_RS_TRACE
    {   // per pass: the phases of the first, a middle and the last tile, relative to the pass's earliest entry stamp (us)
        std::vector<unsigned long long> tr(4 * 1024 * 8);
        DM_TRY(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(la3dm_dev::g_rs_trace), 8ull * tr.size()));
        const uint32_t tiles = std::min(1024u, (n + la3dm_dev::kRsTile - 1) / la3dm_dev::kRsTile), passes = (uint32_t)(bits + 7) / 8;
        for (uint32_t p = 0; p < passes; ++p) {
            unsigned long long t0 = ~0ull, t1 = 0;
            for (uint32_t t = 0; t < tiles; ++t) t0 = std::min(t0, tr[(p * 1024 + t) * 8]), t1 = std::max(t1, tr[(p * 1024 + t) * 8 + 7]);
            fprintf(stderr, "rs_trace pass %u: %u tiles, first entry -> last end %.2f us\n", p, tiles, (t1 - t0) * 0.01);
            for (uint32_t t : {0u, tiles / 4, tiles / 2, 3 * tiles / 4, tiles - 1}) {
                fprintf(stderr, "  tile %4u:", t);
                for (int k = 0; k < 8; ++k) fprintf(stderr, " %6.2f", (double)(long long)(tr[(p * 1024 + t) * 8 + k] - t0) * 0.01);
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    return LA3DM_OK;
}

int la3dm_devmap_training_data(la3dm_devmap *dm, float *xyzy, uint32_t cap, uint32_t *n) {
    if (dm) dm->mailbox_pending = 0;   // (left behind by a call that failed between a publishing launch and its read_counters)
    if (!dm || !n) return LA3DM_ERR_ARG;
    *n = dm->n_xy;
    if (!xyzy || cap < dm->n_xy) return dm->n_xy ? LA3DM_ERR_ARG : LA3DM_OK;
    if (dm->n_xy == 0) return LA3DM_OK;
    DM_TRY(hipSetDevice(dm->ctx->device));
    DM_TRY(hipMemcpyAsync(xyzy, dm->xy.ptr, 16ull * dm->n_xy, hipMemcpyDeviceToHost, dm->ctx->stream));
    DM_TRY(hipStreamSynchronize(dm->ctx->stream));
    return LA3DM_OK;
}

}  // extern "C"
