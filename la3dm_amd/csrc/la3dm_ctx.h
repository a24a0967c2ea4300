// la3dm_ctx.h — internal: the device context shared by the translation units of libla3dm_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/la3dm_hip.h"

struct Arena {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct la3dm_ctx {
    la3dm_params p;
    int device = 0;
    hipStream_t stream = nullptr;  // used by the host-pointer entry points
    float4 *d_lut = nullptr;
    uint32_t lut_count = 0;
    std::string err;
    int n_devmaps = 0;     // live la3dm_devmap objects that point at this context (la3dm_destroy refuses while > 0)
    int opt_bgk_sum = 1;   // BGK accumulate mode: 1 (default) = order-free double accumulators (bgk_predict_fuse_r: the correctly rounded
                           // sums, |dp| <= ~4e-7 from the reference's fp32 chains), 0 = the reference's fp32 summation order
                           // (bgk_predict_fuse_v5, bit-identical to the CPU restatement); env LA3DM_BGK_SUM sets the default
    int opt_bgk_tables = 1;  // bgk_sum = 1 only: 1 (default) = bgk_predict_fuse_t (per-axis distance tables for aligned 4x4x4 tiles, the
                             // other tiles through the general path in the same launch), 0 = bgk_predict_fuse_r for every tile
    float inv_ell = 0.0f;   // RN(1 / ell), or 0 when x / ell must stay an IEEE division (bgk_kernels.h div_by_ell)
    int opt_fast_trig = 0;  // 0 correctly rounded (f64 kernels), 1 f32 polynomial, 2 OCML, 3 Eigen 3.3.7 psin / pcos without FMA (the likely reference build)
    int opt_gp_mode = 0;    // GPOctoMap: 0 = FMA chains in ascending order (VALU and matrix cores alike: the parity configuration), 1 = the order of an
                            // x86-64 / SSE2 build of Eigen 3.3.7 on the VALU (gp_eigen_kernels.h; blocks of up to 128 points); env LA3DM_GP_MODE
    int opt_bgk_tile_desc = 1;   // block_depth >= 4, bgk_sum 1: per-tile neighbour descriptors without the face neighbours out of the tile's reach (bgk_prepare); 0 = block-wide descriptors (A/B)
    int opt_grid_order = 0;  // device-resident map's voxel-grid filters: 0 = ascending cloud index inside a cell (the parity configuration), 1 = the order
                             // pcl::VoxelGrid's unstable std::sort leaves (host sort of the downloaded keys: verification mode, slow by design)
    int opt_time_kernel = 0;
    int opt_waves = 1;  // waves per workgroup (variant 3)
    int opt_remap = 2;
    int opt_l_dense_add = 1;      // BGK-L split tiles: 1 = expansion for all items at once + two-wave ordered add, 0 = producer / consumer workgroup
    int opt_l_split_rows = 1024;  // BGK-L: tiles with more rows than this are split over waves (< 0: never); measured on the 200 k-ray scan, round 5 (order-free sums, row cull): 8192 1.67 ms, 4096 1.63, 2048 1.30, 1024 1.27, 512 1.27 (round 2, ordered: 2048 was the optimum)
    int opt_lds_pad = 0;  // profiling only: extra dynamic LDS bytes on the BGK predict launch (lowers the waves a CU holds)
    int opt_ablate = 0;  // profiling only: 1 skip kernel evaluation, 2 skip the candidate tests
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;  // events around the dominant kernel
    size_t ev_used = 0;
    // scratch (device-pointer path)
    Arena pts_scaled, nbr_range, blk_desc, label_seq;
    uint32_t scan_seq = 0;  // la3dm_bgk_scan_device calls so far (BgkArgs::seq)
    Arena gp_loff, gp_totals, gp_order, gp_L, gp_alpha, gp_v;
    Arena l_task_item, l_split_list, l_nb_first, l_part, l_counters, l_item_desc, l_rowrec, l_batch_off, l_item_hits, l_bdesc, l_vals, l_rowx, l_dense, l_labmask, l_part64;
    Arena lv_samples, lv_sorted, lv_rays, lv_cell, lv_center, lv_cell0, lv_alpha, lv_beta, lv_state;
    Arena lvp_sub_task, lvp_task, lvp_totals, lvp_rows, lvp_sub_out, lvp_cand;  // BGK-LV work plan + row scratch of split cubes
    // staging (host-pointer path)
    Arena h_train, h_train_off, h_nbr, h_center, h_leaf_off, h_leaf_key, h_alpha, h_beta, h_state, h_diag_in,
        h_diag_out;
};

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
            return LA3DM_ERR_HIP;                                                                   \
        }                                                                                           \
    } while (0)

static inline int arena_reserve(la3dm_ctx *ctx, Arena &a, size_t bytes) {
    if (bytes <= a.cap) return LA3DM_OK;
    if (a.ptr) {
        HIP_TRY(ctx, hipFree(a.ptr));
        a.ptr = nullptr;
        a.cap = 0;
    }
    // 50 % head room: a regrow is a device-wide free + malloc (hundreds of us); above 1 GB an eighth (the BGK-L split
    // scratch is 64 KB per item: tens of GB on a large scan — ADVICE r02)
    size_t want = bytes + (bytes > (1ull << 30) ? bytes / 8 : bytes / 2) + 256;
    hipError_t e = hipMalloc(&a.ptr, want);
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
        a.ptr = nullptr;
        return LA3DM_ERR_OOM;
    }
    a.cap = want;
    return LA3DM_OK;
}


// ---- internal (devmap.hip -> la3dm_hip.hip): the BGKLV voxel kernel run in place on the device-resident block pool ----
struct la3dm_lv_pool_scan {
    const float *samples, *sorted, *rays;  // as la3dm_lv_scan (device pointers)
    const uint32_t *cell_off;
    int32_t cell_min[3], cell_dim[3];
    uint32_t n_blk;
    const float *blk_center;
    const int32_t *blk_cell0;
    const uint32_t *blk_slot, *blk_mult;   // pool slot and candidate-key multiplicity of every packed block
    float *A, *B;                          // the pool
    uint8_t *S;
    uint32_t npb, pass;
    uint32_t *upd_counter;                 // += nodes updated
    uint32_t n_samples;
    uint32_t plan_n_blk;                   // block count the plan arrays were laid out for (>= n_blk)
    const uint32_t *n_blk_dev;             // plan only: the packed block count, still on the device (nullptr: n_blk)
    uint32_t plan_totals[4];               // read back from the plan: workgroups of split cubes, scratch rows, split cubes,
                                           // workgroups of the other cubes
    uint32_t *plan_totals_dev;             // where the plan left them on the device (the voxel kernel reads [0])
};
// plan once per scan (totals_dev: 3 words the caller reads back into plan_totals), then one scan per pass
int la3dm_bgklv_pool_plan_device(la3dm_ctx *ctx, const la3dm_lv_pool_scan *s, uint32_t *totals_dev, hipStream_t stream);
int la3dm_bgklv_pool_scan_device(la3dm_ctx *ctx, const la3dm_lv_pool_scan *s, hipStream_t stream);
