// la3dm_hip.hip — implementation of the C ABI in include/la3dm_hip.h.
// Owns the device context: voxel LUT, grow-only scratch arenas, launch glue.
// There is no CPU fallback anywhere in this file.
#include "../../include/la3dm_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bgk_kernels.h"
#include "gp_kernels.h"
#include "gp_eigen_kernels.h"
#include "lv_kernels.h"
#include "bgkl_kernels.h"

using namespace la3dm_dev;

#include "la3dm_ctx.h"

static thread_local std::string g_create_error;

// the part of LvArgs that only depends on the map's parameters and the number of packed blocks
static void lv_fill_args(la3dm_ctx *ctx, LvArgs &a, uint32_t n_blk) {
    const int d = ctx->p.block_depth;
    memset(&a, 0, sizeof(a));
    a.lut = ctx->d_lut;
    a.lut_base = 0;
    for (int k = 0; k + 1 < d; ++k) a.lut_base += 1u << (3 * k);
    a.nodes_per_blk = 1u << (3 * (d - 1));
    const uint32_t cubes = (a.nodes_per_blk + kWave - 1) / kWave;  // 8^(d-3) for d >= 3, else 1
    a.cubes_shift = 0;
    while ((1u << a.cubes_shift) < cubes) ++a.cubes_shift;
    a.cubes_bits = a.cubes_shift / 3;
    a.n_tasks = n_blk << a.cubes_shift;
    const double g = d >= 3 ? 4.0 * (double)ctx->p.resolution : (double)((float)pow(2, d - 1) * ctx->p.resolution);
    a.reach = (int)ceil((double)ctx->p.ell / g);
    a.sf2 = ctx->p.sf2;
    a.ell = ctx->p.ell;
    a.inv_ell = ctx->inv_ell;
    a.free_thresh = ctx->p.free_thresh;
    a.occupied_thresh = ctx->p.occupied_thresh;
    a.var_thresh = ctx->p.var_thresh;
    a.min_W = ctx->p.min_W;
    a.trig = ctx->opt_fast_trig == 3 ? 3 : 0;
}

// ---- BGK-LV work plan (lv_kernels.h bgklv_plan_kernel): which cube gets how many workgroups ----
// upper bound of the workgroups a plan can hand out: sum over the cubes of ceil(stream / chunk)
static uint64_t lv_max_subs(const LvArgs &a, uint32_t n_tasks, uint32_t n_samples) {
    const uint64_t w = 2ull * (uint64_t)a.reach + 1ull, nb = w * w * w;
    return (uint64_t)n_tasks + nb * (uint64_t)n_samples / kLvChunk + 1;
}
static void lv_plan_layout(la3dm_ctx *ctx, LvArgs &a, uint32_t n_tasks, uint32_t n_samples) {
    a.max_subs = (uint32_t)lv_max_subs(a, n_tasks, n_samples);
    a.sub_task = (uint32_t *)ctx->lvp_sub_task.ptr;
    a.task_first = (uint32_t *)ctx->lvp_task.ptr;
    a.task_nsub = a.task_first + n_tasks;
    a.task_row0 = a.task_nsub + n_tasks;
    a.split_list = a.task_row0 + n_tasks;
}
// Reserves the plan arrays, points `a` at them and launches the plan kernel; the three totals (workgroups, scratch rows,
// split cubes) land in totals_dev (zeroed here unless the caller vouches for them), for the caller to read back.
static int lv_plan_launch(la3dm_ctx *ctx, LvArgs &a, uint32_t n_samples, uint32_t *totals_dev, hipStream_t stream, bool zero_totals = true) {
    const uint64_t max_subs = lv_max_subs(a, a.n_tasks, n_samples);
    if (max_subs > 0x3FFFFFFFull) {
        ctx->err = "lv scan: too many cubes x samples for the work plan";
        return LA3DM_ERR_ARG;
    }
    int rc;
    if ((rc = arena_reserve(ctx, ctx->lvp_sub_task, 8 * max_subs)) != LA3DM_OK) return rc;
    if ((rc = arena_reserve(ctx, ctx->lvp_task, 16ull * a.n_tasks)) != LA3DM_OK) return rc;
    if ((rc = arena_reserve(ctx, ctx->lvp_cand, sizeof(LvCand) * (size_t)(n_samples ? n_samples : 1))) != LA3DM_OK) return rc;
    a.cand = (LvCand *)ctx->lvp_cand.ptr;
    a.n_samples = n_samples;
    if (n_samples) hipLaunchKernelGGL(bgklv_cand_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, stream, a);
    lv_plan_layout(ctx, a, a.n_tasks, n_samples);
    a.plan_totals = totals_dev;
    if (zero_totals) HIP_TRY(ctx, hipMemsetAsync(totals_dev, 0, 16, stream));
    hipLaunchKernelGGL(bgklv_plan_kernel, dim3((a.n_tasks + 4 * kLvPlanPerWave - 1) / (4 * kLvPlanPerWave)), dim3(256), 0, stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return LA3DM_OK;
}

// the voxel kernel over the planned workgroups + the ordered adds of the split cubes (totals: as read back from the plan)
static int lv_run_planned(la3dm_ctx *ctx, LvArgs &a, const uint32_t totals[4], hipStream_t stream) {
    const uint32_t n_heavy = totals[0], n_rows = totals[1], n_split = totals[2], n_subs = totals[0] + totals[3];
    if (n_subs == 0) return LA3DM_OK;
    int rc;
    // scratch of the split cubes: ordered mode = one 256-byte row per staged candidate; order-free mode = 1 KB per workgroup
    const bool lv_f64 = ctx->opt_bgk_sum == 1;
    if ((rc = arena_reserve(ctx, ctx->lvp_rows, lv_f64 ? sizeof(double2) * kWave * (size_t)(n_heavy ? n_heavy : 1) : 256ull * (n_rows ? n_rows : 1))) != LA3DM_OK) return rc;
    constexpr size_t kWords = kLvChunk / kWave;
    if ((rc = arena_reserve(ctx, ctx->lvp_sub_out, 8ull * (2 * kWords + 1) * n_subs)) != LA3DM_OK) return rc;
    a.rows = (float *)ctx->lvp_rows.ptr;
    a.sub_part = (double2 *)ctx->lvp_rows.ptr;
    a.sub_info = (unsigned long long *)ctx->lvp_sub_out.ptr;
    a.sub_nz = a.sub_info + n_subs;
    a.sub_y = a.sub_nz + kWords * n_subs;
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (ctx->opt_time_kernel) {
        if (ctx->ev_used == ctx->ev_pool.size()) {
            std::pair<hipEvent_t, hipEvent_t> p;
            HIP_TRY(ctx, hipEventCreate(&p.first));
            HIP_TRY(ctx, hipEventCreate(&p.second));
            ctx->ev_pool.push_back(p);
        }
        ev = &ctx->ev_pool[ctx->ev_used++];
        HIP_TRY(ctx, hipEventRecord(ev->first, stream));
    }
    if (ctx->opt_bgk_sum == 1) {
        // order-free accumulate mode (the default): double sums per voxel, a split cube's workgroups leave 1 KB of partial sums
        // (without the 16 KB tile four workgroups fit a CU: the 8-waves-per-SIMD build, 0.755 -> 0.661 ms on the 50 k-ray scan)
        hipLaunchKernelGGL(bgklv_voxel_kernel_w8, dim3(n_subs), dim3(kLvWaves * kWave), 0, stream, a);
        if (n_split) hipLaunchKernelGGL(bgklv_split_apply64, dim3(n_split), dim3(kWave), 0, stream, a);
    } else {
        hipLaunchKernelGGL(bgklv_voxel_kernel<false>, dim3(n_subs), dim3(kLvWaves * kWave), 0, stream, a);
        if (n_split) hipLaunchKernelGGL(bgklv_split_add_kernel, dim3(n_split), dim3(kLvWaves * kWave), 0, stream, a);
    }
    if (ev) HIP_TRY(ctx, hipEventRecord(ev->second, stream));
    HIP_TRY(ctx, hipGetLastError());
    return LA3DM_OK;
}

static int lv_pool_args(la3dm_ctx *ctx, const la3dm_lv_pool_scan *s, LvArgs &a) {
    if (!ctx || !s) return LA3DM_ERR_ARG;
    if (ctx->p.variant != 2) {
        ctx->err = "la3dm_bgklv_pool_scan: the context was not created with variant = 2 (BGKLVOctoMap)";
        return LA3DM_ERR_ARG;
    }
    lv_fill_args(ctx, a, s->n_blk);
    a.samples = (const float4 *)s->samples;
    a.sorted = (const float4 *)s->sorted;
    a.rays = (const float4 *)s->rays;
    a.cell_off = s->cell_off;
    a.blk_center = s->blk_center;
    a.blk_cell0 = s->blk_cell0;
    a.alpha = s->A;
    a.beta = s->B;
    a.state = s->S;
    for (int i = 0; i < 3; ++i) {
        a.cell_min[i] = s->cell_min[i];
        a.cell_dim[i] = s->cell_dim[i];
    }
    a.blk_slot = s->blk_slot;
    a.blk_mult = s->blk_mult;
    a.upd_counter = s->upd_counter;
    a.npb = s->npb;
    a.layer_off = a.lut_base;  // the pool is depth-major like the LUT: the finest layer starts at (8^(d-1) - 1) / 7
    a.pass = s->pass;
    a.n_blk_dev = s->n_blk_dev;
    return LA3DM_OK;
}

int la3dm_bgklv_pool_plan_device(la3dm_ctx *ctx, const la3dm_lv_pool_scan *s, uint32_t *totals_dev, hipStream_t stream) {
    if (!s || s->n_blk == 0) return LA3DM_OK;
    LvArgs a;
    int rc = lv_pool_args(ctx, s, a);
    if (rc != LA3DM_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lv_plan_launch(ctx, a, s->n_samples, totals_dev, stream, false);   // the caller's counter block is zero at this point
}

int la3dm_bgklv_pool_scan_device(la3dm_ctx *ctx, const la3dm_lv_pool_scan *s, hipStream_t stream) {
    if (!s || s->n_blk == 0) return LA3DM_OK;
    LvArgs a;
    int rc = lv_pool_args(ctx, s, a);
    if (rc != LA3DM_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the plan arrays are where la3dm_bgklv_pool_plan_device left them
    lv_plan_layout(ctx, a, (s->plan_n_blk ? s->plan_n_blk : s->n_blk) << a.cubes_shift, s->n_samples);
    a.plan_totals = s->plan_totals_dev;
    a.cand = (LvCand *)ctx->lvp_cand.ptr;   // built by the plan step
    a.n_samples = s->n_samples;
    return lv_run_planned(ctx, a, s->plan_totals, stream);
}

extern "C" {

int la3dm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *la3dm_version(void) { return "la3dm_hip 0.1 (gfx950)"; }

const char *la3dm_last_error(const la3dm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int la3dm_create(const la3dm_params *params, la3dm_ctx **out) {
    if (!params || !out) {
        g_create_error = "la3dm_create: null argument";
        return LA3DM_ERR_ARG;
    }
    *out = nullptr;
    if (params->block_depth < 1 || params->block_depth > 6 || !(params->ell > 0.0f) || !params->lut_xyz) {
        g_create_error = "la3dm_create: bad params (block_depth must be 1..6, ell > 0, lut_xyz non-null)";
        return LA3DM_ERR_ARG;
    }
    uint32_t want = 0;
    for (int d = 0; d < params->block_depth; ++d) want += 1u << (3 * d);
    if (params->lut_count != want) {
        g_create_error = "la3dm_create: lut_count must be sum_{d<block_depth} 8^d";
        return LA3DM_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "la3dm_create: no HIP device visible (this library has no CPU fallback)";
        return LA3DM_ERR_NODEVICE;
    }
    if (params->device < 0 || params->device >= ndev) {
        g_create_error = "la3dm_create: device ordinal out of range";
        return LA3DM_ERR_ARG;
    }
    la3dm_ctx *ctx = new la3dm_ctx;
    ctx->p = *params;
    ctx->p.lut_xyz = nullptr;
    ctx->device = params->device;
    {   // x / ell by reciprocal + one exact correction is the IEEE quotient unless ell's significand is all ones
        uint32_t eb;
        memcpy(&eb, &params->ell, 4);
        ctx->inv_ell = (eb & 0x7FFFFFu) != 0x7FFFFFu ? 1.0f / params->ell : 0.0f;
    }
    ctx->lut_count = params->lut_count;
    if (const char *ev = getenv("LA3DM_BGK_SUM")) {  // default accumulate mode of new contexts (la3dm_set_option "bgk_sum" overrides)
        if (ev[0] == '0' || ev[0] == '1') ctx->opt_bgk_sum = ev[0] - '0';
    }
    if (const char *ev = getenv("LA3DM_BGK_TABLES")) {  // bgk_sum = 1: 0 = bgk_predict_fuse_r for every tile
        if (ev[0] == '0' || ev[0] == '1') ctx->opt_bgk_tables = ev[0] - '0';
    }
    if (const char *ev = getenv("LA3DM_BGKL_SPLIT_ROWS")) {   // default of "bgkl_split_rows": an integer (< 0: never split); anything else is ignored
        char *end = nullptr;
        const long v = strtol(ev, &end, 10);
        if (end != ev && *end == 0 && v >= -1 && v <= (1 << 24)) ctx->opt_l_split_rows = (int)v;
    }
    if (const char *ev = getenv("LA3DM_GP_MODE")) {  // default of "gp_mode"
        if (ev[0] == '0' || ev[0] == '1') ctx->opt_gp_mode = ev[0] - '0';
    }
    auto fail = [&](const char *what, hipError_t e) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(e);
        delete ctx;
        return LA3DM_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(ctx->device)) != hipSuccess) return fail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
        return fail("hipStreamCreate", e);
    std::vector<float4> lut4(ctx->lut_count);
    for (uint32_t i = 0; i < ctx->lut_count; ++i)
        lut4[i] = make_float4(params->lut_xyz[3 * i], params->lut_xyz[3 * i + 1], params->lut_xyz[3 * i + 2], 0.0f);
    if ((e = hipMalloc((void **)&ctx->d_lut, sizeof(float4) * ctx->lut_count)) != hipSuccess)
        return fail("hipMalloc(lut)", e);
    if ((e = hipMemcpy(ctx->d_lut, lut4.data(), sizeof(float4) * ctx->lut_count, hipMemcpyHostToDevice)) != hipSuccess)
        return fail("hipMemcpy(lut)", e);
    *out = ctx;
    return LA3DM_OK;
}

void la3dm_destroy(la3dm_ctx *ctx) {
    if (!ctx) return;
    if (ctx->n_devmaps > 0) {  // a device-resident map still points at this context: freeing it now would leave it dangling
        ctx->err = "la3dm_destroy: destroy the context's la3dm_devmap objects first (context kept alive)";
        fprintf(stderr, "%s\n", ctx->err.c_str());
        return;
    }
    (void)hipSetDevice(ctx->device);
    Arena *all[] = {&ctx->l_task_item, &ctx->l_split_list, &ctx->l_nb_first, &ctx->l_part, &ctx->l_counters, &ctx->l_item_desc, &ctx->l_rowrec, &ctx->l_batch_off, &ctx->l_item_hits, &ctx->l_bdesc, &ctx->l_vals, &ctx->l_rowx, &ctx->l_dense, &ctx->l_labmask, &ctx->l_part64,
                    &ctx->pts_scaled, &ctx->nbr_range, &ctx->blk_desc, &ctx->label_seq, &ctx->gp_loff, &ctx->gp_totals, &ctx->gp_order, &ctx->gp_L, &ctx->gp_alpha, &ctx->gp_v, &ctx->lv_samples, &ctx->lv_sorted, &ctx->lv_rays, &ctx->lv_cell, &ctx->lv_center,
                    &ctx->lv_cell0, &ctx->lv_alpha, &ctx->lv_beta, &ctx->lv_state, &ctx->lvp_sub_task, &ctx->lvp_task, &ctx->lvp_cand, &ctx->lvp_totals, &ctx->lvp_rows, &ctx->lvp_sub_out, &ctx->h_train, &ctx->h_train_off, &ctx->h_nbr, &ctx->h_center, &ctx->h_leaf_off,
                    &ctx->h_leaf_key, &ctx->h_alpha, &ctx->h_beta, &ctx->h_state, &ctx->h_diag_in, &ctx->h_diag_out};
    for (Arena *a : all)
        if (a->ptr) (void)hipFree(a->ptr);
    for (auto &p : ctx->ev_pool) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    if (ctx->d_lut) (void)hipFree(ctx->d_lut);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int la3dm_set_option(la3dm_ctx *ctx, const char *name, int value) {
    if (!ctx || !name) return LA3DM_ERR_ARG;
    auto bad_value = [&](const char *allowed) {
        ctx->err = std::string("la3dm_set_option: ") + name + " must be " + allowed;
        return LA3DM_ERR_ARG;
    };
    if (!strcmp(name, "bgk_sum")) {  // 0 = the reference's fp32 summation order, 1 = order-free double accumulators
        if (value < 0 || value > 1) return bad_value("0 or 1");
        ctx->opt_bgk_sum = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "bgk_tables")) {  // bgk_sum = 1: 1 = distance tables for aligned tiles (bgk_predict_fuse_t), 0 = bgk_predict_fuse_r
        if (value < 0 || value > 1) return bad_value("0 or 1");
        ctx->opt_bgk_tables = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "fast_trig")) {  // 0 correctly rounded (default, the parity configuration), 1 f32 polynomial, 2 OCML, 3 = Eigen 3.3.7's psin / pcos
        // without FMA, the likely reference build (BGK, BGK-L and BGK-LV kernels; 1 and 2: the BGK kernels only)
        if (value < 0 || value > 3) return bad_value("0, 1, 2 or 3");
        ctx->opt_fast_trig = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "waves_per_wg")) {  // launch bounds and LDS sizing exist for these three only
        if (value != 1 && value != 2 && value != 4) return bad_value("1, 2 or 4");
        ctx->opt_waves = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "ablate")) {
        if (value < 0 || value > 31) return bad_value("0..31");
        ctx->opt_ablate = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "lds_pad")) {
        if (value < 0 || value > 32768) return bad_value("0..32768");
        ctx->opt_lds_pad = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "remap")) {
        if (value < 0 || value > 2) return bad_value("0, 1 or 2");
        ctx->opt_remap = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "bgk_tile_desc")) {
        if (value < 0 || value > 1) return bad_value("0 or 1");
        ctx->opt_bgk_tile_desc = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "grid_order")) {
        if (value < 0 || value > 1) return bad_value("0 (ascending cloud index inside a voxel-grid cell) or 1 (what libstdc++'s std::sort on the cell index alone leaves, as pcl::VoxelGrid: verification mode)");
        ctx->opt_grid_order = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "gp_mode")) {
        if (value < 0 || value > 1) return bad_value("0 (FMA chains in ascending order: the parity configuration) or 1 (Eigen 3.3.7 / SSE2 order, no FMA, pexp)");
        ctx->opt_gp_mode = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "bgkl_split_rows")) {
        if (value < -1 || value > (1 << 24)) return bad_value("-1 (never split) .. 2^24");
        ctx->opt_l_split_rows = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "bgkl_dense_add")) {
        if (value < 0 || value > 1) return bad_value("0 or 1");
        ctx->opt_l_dense_add = value;
        return LA3DM_OK;
    }
    if (!strcmp(name, "time_kernel")) {
        ctx->opt_time_kernel = value;
        ctx->ev_used = 0;
        return LA3DM_OK;
    }
    ctx->err = std::string("la3dm_set_option: unknown option ") + name;
    return LA3DM_ERR_ARG;
}

int la3dm_get_option(const la3dm_ctx *ctx, const char *name, int *value) {
    if (!ctx || !name || !value) return LA3DM_ERR_ARG;
    if (!strcmp(name, "bgk_sum")) *value = ctx->opt_bgk_sum;
    else if (!strcmp(name, "bgk_tables")) *value = ctx->opt_bgk_tables;
    else if (!strcmp(name, "fast_trig")) *value = ctx->opt_fast_trig;
    else if (!strcmp(name, "gp_mode")) *value = ctx->opt_gp_mode;
    else if (!strcmp(name, "grid_order")) *value = ctx->opt_grid_order;
    else if (!strcmp(name, "bgk_tile_desc")) *value = ctx->opt_bgk_tile_desc;
    else if (!strcmp(name, "waves_per_wg")) *value = ctx->opt_waves;
    else if (!strcmp(name, "remap")) *value = ctx->opt_remap;
    else return LA3DM_ERR_ARG;
    return LA3DM_OK;
}

static int check_scan(la3dm_ctx *ctx, const la3dm_bgk_scan *s) {
    if (!ctx) return LA3DM_ERR_ARG;
    if (!s) {
        ctx->err = "scan: null";
        return LA3DM_ERR_ARG;
    }
    if (s->n_test_blk == 0) return LA3DM_OK;
    if (!s->nbr || !s->blk_center || !s->leaf_off || !s->leaf_key || !s->alpha || !s->beta || !s->state ||
        !s->train_off || (s->n_train_pts && !s->train_xyzy)) {
        ctx->err = "scan: null array pointer";
        return LA3DM_ERR_ARG;
    }
    return LA3DM_OK;
}

int la3dm_bgk_scan_device(la3dm_ctx *ctx, const la3dm_bgk_scan *s, void *stream_, la3dm_bgk_counters *out) {
    int rc = check_scan(ctx, s);
    if (rc != LA3DM_OK) return rc;
    if (out) memset(out, 0, sizeof(*out));
    if (s->n_test_blk == 0) return LA3DM_OK;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // 1. x / ell once per training point
    rc = arena_reserve(ctx, ctx->pts_scaled, sizeof(float4) * (size_t)(s->n_train_pts ? s->n_train_pts : 1));
    if (rc != LA3DM_OK) return rc;
    rc = arena_reserve(ctx, ctx->nbr_range, sizeof(uint2) * 7 * (size_t)s->n_test_blk);
    if (rc != LA3DM_OK) return rc;
    const bool sum_f64 = ctx->opt_bgk_sum == 1;
    uint32_t max_leaves = 1u << (3 * (ctx->p.block_depth - 1));
    uint32_t tpb = (max_leaves + kWave - 1) / kWave;  // power of two
    uint32_t tpb_shift = 0;
    while ((1u << tpb_shift) < tpb) ++tpb_shift;
    // block_depth >= 4: one neighbour descriptor per TILE, without the face neighbours the tile's voxel cube cannot reach
    // (bgk_prepare) — valid while the kernel's support, ell, is at most four voxel edges (the cube's edge), and for gated scans
    // only (insert_training_data updates a leaf for every neighbour that has a model, reachable or not); option "bgk_tile_desc" 0 = off
    const bool tile_desc = sum_f64 && tpb_shift >= 3 && ctx->opt_bgk_tile_desc && !(s->flags & LA3DM_SCAN_UPDATE_UNGATED) &&
                           ctx->p.ell <= 4.0f * ctx->p.resolution;
    const uint32_t desc_shift = tile_desc ? tpb_shift : 0u;
    if (sum_f64) {
        rc = arena_reserve(ctx, ctx->blk_desc, sizeof(uint32_t) * 16 * ((size_t)s->n_test_blk << desc_shift));
        if (rc != LA3DM_OK) return rc;
        if (!ctx->label_seq.ptr) {
            rc = arena_reserve(ctx, ctx->label_seq, sizeof(uint32_t));
            if (rc != LA3DM_OK) return rc;
            HIP_TRY(ctx, hipMemsetAsync(ctx->label_seq.ptr, 0, sizeof(uint32_t), stream));
            ctx->scan_seq = 0;
        }
        if (++ctx->scan_seq == 0u) ctx->scan_seq = 1u;  // 0 is the cleared state
    }
    // the table kernels: tiles of full (un-pruned) blocks through the distance tables, the others through the general
    // path of the same launch; they need every label to be 0 or 1
    const bool use_tables = sum_f64 && ctx->opt_bgk_tables && ctx->p.block_depth >= 3 && (s->flags & LA3DM_SCAN_LABELS_01) != 0u;
    // the instance without the general path only when the caller's LA3DM_SCAN_FULL_BLOCKS can be verified here: a block
    // holds at most 8^(depth-1) leaves, so the total says whether every block is full (a tile of a block that is
    // not would be skipped silently)
    const bool full_blocks = (s->flags & LA3DM_SCAN_FULL_BLOCKS) != 0u && (uint64_t)s->n_leaf == ((uint64_t)s->n_test_blk << (3 * (ctx->p.block_depth - 1)));
    {
        const uint32_t n_nbr = 7u * s->n_test_blk;
        uint32_t n_thr = s->n_train_pts > n_nbr ? s->n_train_pts : n_nbr;
        BgkDescArgs da;
        da.shift = desc_shift;
        da.depth = (uint32_t)ctx->p.block_depth;
        da.leaf_off = s->leaf_off;
        if (desc_shift && n_thr < (s->n_test_blk << desc_shift)) n_thr = s->n_test_blk << desc_shift;
        dim3 g((n_thr + 255) / 256), b(256);
        // what the launch writes beside the scaled points depends on the kernel that follows: the ordered kernel reads the
        // per-neighbour ranges, the others (bgk_predict_fuse_t / _r) the 64-byte descriptors — each of them a dependent chain
        // nbr -> train_off per block, so only the one that will be read is produced
        const bool want_desc = sum_f64;
        hipLaunchKernelGGL(bgk_prepare, g, b, 0, stream, (const float4 *)s->train_xyzy, (float4 *)ctx->pts_scaled.ptr,
                           s->n_train_pts, ctx->p.ell, s->nbr, s->train_off, sum_f64 ? (uint2 *)nullptr : (uint2 *)ctx->nbr_range.ptr, n_nbr,
                           want_desc ? (uint32_t *)ctx->blk_desc.ptr : (uint32_t *)nullptr,
                           sum_f64 ? (uint32_t *)ctx->label_seq.ptr : (uint32_t *)nullptr, ctx->scan_seq, da);
    }

    // 2. predict + fuse
    BgkArgs a;
    a.pts = (const float4 *)ctx->pts_scaled.ptr;
    a.train_off = s->train_off;
    a.nbr = s->nbr;
    a.blk_center = s->blk_center;
    a.leaf_off = s->leaf_off;
    a.leaf_key = s->leaf_key;
    a.alpha = s->alpha;
    a.beta = s->beta;
    a.state = s->state;
    a.lut = ctx->d_lut;
    a.nbr_range = (const uint2 *)ctx->nbr_range.ptr;
    a.blk_desc = (const uint32_t *)ctx->blk_desc.ptr;
    a.label_seq = (const uint32_t *)ctx->label_seq.ptr;
    a.seq = ctx->scan_seq;
    a.n_test_blk = s->n_test_blk;
    a.tpb_shift = tpb_shift;
    a.desc_shift = desc_shift;
    a.n_tasks = s->n_test_blk << tpb_shift;
    a.flags = s->flags | ((uint32_t)ctx->opt_ablate << 8);
    a.remap = (uint32_t)ctx->opt_remap;
    a.depth = (uint32_t)ctx->p.block_depth;
    a.inv_ell = ctx->inv_ell;
    a.sf2 = ctx->p.sf2;
    a.ell = ctx->p.ell;
    a.free_thresh = ctx->p.free_thresh;
    a.occupied_thresh = ctx->p.occupied_thresh;
    a.var_thresh = ctx->p.var_thresh;
    dim3 grid((a.n_tasks + kWavesPerWG - 1) / kWavesPerWG), block(kWavesPerWG * kWave);
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (ctx->opt_time_kernel) {
        if (ctx->ev_used == ctx->ev_pool.size()) {
            std::pair<hipEvent_t, hipEvent_t> p;
            HIP_TRY(ctx, hipEventCreate(&p.first));
            HIP_TRY(ctx, hipEventCreate(&p.second));
            ctx->ev_pool.push_back(p);
        }
        ev = &ctx->ev_pool[ctx->ev_used++];
        HIP_TRY(ctx, hipEventRecord(ev->first, stream));
    }
#define LAUNCH_BGK(KERNEL, ...)                                                                 \
    switch (ctx->opt_fast_trig) {                                                              \
    case 1: hipLaunchKernelGGL((KERNEL<1 __VA_ARGS__>), grid, block, (size_t)ctx->opt_lds_pad, stream, a); break;     \
    case 2: hipLaunchKernelGGL((KERNEL<2 __VA_ARGS__>), grid, block, (size_t)ctx->opt_lds_pad, stream, a); break;     \
    case 3: hipLaunchKernelGGL((KERNEL<3 __VA_ARGS__>), grid, block, (size_t)ctx->opt_lds_pad, stream, a); break;     \
    default: hipLaunchKernelGGL((KERNEL<0 __VA_ARGS__>), grid, block, (size_t)ctx->opt_lds_pad, stream, a); break;    \
    }
    if (sum_f64) {
        grid = dim3(a.n_tasks);
        block = dim3(kWave);
        if (use_tables) {
            if (full_blocks) {  // no pruned block in this scan: the kernel without the general path
                LAUNCH_BGK(bgk_predict_fuse_t, , false)
            } else {
                LAUNCH_BGK(bgk_predict_fuse_t)
            }
        } else {
            LAUNCH_BGK(bgk_predict_fuse_r)
        }
    } else {
        const int w = ctx->opt_waves;
        grid = dim3((a.n_tasks + w - 1) / w);
        block = dim3(w * kWave);
        if (w == 4) {
            LAUNCH_BGK(bgk_predict_fuse_v5, , 4)
        } else if (w == 2) {
            LAUNCH_BGK(bgk_predict_fuse_v5, , 2)
        } else {
            LAUNCH_BGK(bgk_predict_fuse_v5, , 1)
        }
    }
#undef LAUNCH_BGK
    if (ev) HIP_TRY(ctx, hipEventRecord(ev->second, stream));
    HIP_TRY(ctx, hipGetLastError());
    if (out) {
        out->n_tiles = a.n_tasks;
        out->scratch_bytes = sizeof(float4) * (size_t)s->n_train_pts;
    }
    return LA3DM_OK;
}

typedef int (*scan_device_fn)(la3dm_ctx *, const la3dm_bgk_scan *, void *, la3dm_bgk_counters *);

static int scan_host_common(la3dm_ctx *ctx, const la3dm_bgk_scan *s, la3dm_bgk_counters *out, scan_device_fn run,
                            size_t row_floats = 4) {
    int rc = check_scan(ctx, s);
    if (rc != LA3DM_OK) return rc;
    if (out) memset(out, 0, sizeof(*out));
    if (s->flags & LA3DM_SCAN_ROWS_PREPARED) {   // (the host form uploads 8 floats per row; the prepared form is 12 and device-only)
        ctx->err = "la3dm_*_scan_host: LA3DM_SCAN_ROWS_PREPARED is a flag of la3dm_bgkl_scan_device only (rows of 12 floats already in HBM)";
        return LA3DM_ERR_ARG;
    }
    if (s->n_test_blk == 0) return LA3DM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    struct Up {
        Arena *a;
        const void *src;
        size_t bytes;
    } ups[] = {
        {&ctx->h_train, s->train_xyzy, sizeof(float) * row_floats * (size_t)s->n_train_pts},
        {&ctx->h_train_off, s->train_off, sizeof(uint32_t) * ((size_t)s->n_train_blk + 1)},
        {&ctx->h_nbr, s->nbr, sizeof(int32_t) * 7 * (size_t)s->n_test_blk},
        {&ctx->h_center, s->blk_center, sizeof(float) * 3 * (size_t)s->n_test_blk},
        {&ctx->h_leaf_off, s->leaf_off, sizeof(uint32_t) * ((size_t)s->n_test_blk + 1)},
        {&ctx->h_leaf_key, s->leaf_key, sizeof(uint32_t) * (size_t)s->n_leaf},
        {&ctx->h_alpha, s->alpha, sizeof(float) * (size_t)s->n_leaf},
        {&ctx->h_beta, s->beta, sizeof(float) * (size_t)s->n_leaf},
    };
    for (auto &u : ups) {
        rc = arena_reserve(ctx, *u.a, u.bytes ? u.bytes : 16);
        if (rc != LA3DM_OK) return rc;
        if (u.bytes) HIP_TRY(ctx, hipMemcpyAsync(u.a->ptr, u.src, u.bytes, hipMemcpyHostToDevice, st));
    }
    rc = arena_reserve(ctx, ctx->h_state, s->n_leaf ? s->n_leaf : 16);
    if (rc != LA3DM_OK) return rc;
    la3dm_bgk_scan d = *s;
    d.train_xyzy = (const float *)ctx->h_train.ptr;
    d.train_off = (const uint32_t *)ctx->h_train_off.ptr;
    d.nbr = (const int32_t *)ctx->h_nbr.ptr;
    d.blk_center = (const float *)ctx->h_center.ptr;
    d.leaf_off = (const uint32_t *)ctx->h_leaf_off.ptr;
    d.leaf_key = (const uint32_t *)ctx->h_leaf_key.ptr;
    d.alpha = (float *)ctx->h_alpha.ptr;
    d.beta = (float *)ctx->h_beta.ptr;
    d.state = (uint8_t *)ctx->h_state.ptr;
    rc = run(ctx, &d, st, out);
    if (rc != LA3DM_OK) return rc;
    if (s->n_leaf) {
        HIP_TRY(ctx, hipMemcpyAsync(s->alpha, d.alpha, sizeof(float) * (size_t)s->n_leaf, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(s->beta, d.beta, sizeof(float) * (size_t)s->n_leaf, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(s->state, d.state, (size_t)s->n_leaf, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return LA3DM_OK;
}

int la3dm_bgk_scan_host(la3dm_ctx *ctx, const la3dm_bgk_scan *s, la3dm_bgk_counters *out) {
    return scan_host_common(ctx, s, out, la3dm_bgk_scan_device);
}

int la3dm_gp_scan_device(la3dm_ctx *ctx, const la3dm_bgk_scan *s, void *stream_, la3dm_bgk_counters *out) {
    int rc = check_scan(ctx, s);
    if (rc != LA3DM_OK) return rc;
    if (out) memset(out, 0, sizeof(*out));
    if (s->n_test_blk == 0) return LA3DM_OK;
    if (ctx->p.variant != 1) {
        ctx->err = "la3dm_gp_scan: the context was not created with variant = 1 (GPOctoMap)";
        return LA3DM_ERR_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npts = s->n_train_pts ? s->n_train_pts : 1, nblk = s->n_train_blk ? s->n_train_blk : 1;
    if ((rc = arena_reserve(ctx, ctx->pts_scaled, sizeof(float4) * npts)) != LA3DM_OK) return rc;
    if ((rc = arena_reserve(ctx, ctx->nbr_range, sizeof(uint2) * 7 * (size_t)s->n_test_blk)) != LA3DM_OK) return rc;
    if ((rc = arena_reserve(ctx, ctx->gp_loff, sizeof(unsigned long long) * nblk)) != LA3DM_OK) return rc;
    if ((rc = arena_reserve(ctx, ctx->gp_totals, 32)) != LA3DM_OK) return rc;
    if ((rc = arena_reserve(ctx, ctx->gp_order, sizeof(uint32_t) * nblk)) != LA3DM_OK) return rc;
    if ((rc = arena_reserve(ctx, ctx->gp_alpha, sizeof(float) * npts)) != LA3DM_OK) return rc;
    hipLaunchKernelGGL(gp_factor_offsets, dim3(1), dim3(kGpOffThreads), 0, stream, s->train_off, s->n_train_blk,
                       (unsigned long long *)ctx->gp_loff.ptr, (unsigned long long *)ctx->gp_totals.ptr, (uint32_t *)ctx->gp_order.ptr);
    unsigned long long sum_n2 = s->train_sum_n2;
    uint32_t max_n = s->train_max_n;
    if (sum_n2 == 0 || max_n == 0) {  // no hints: one synchronisation to size the factor arena
        unsigned long long tot[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(tot, ctx->gp_totals.ptr, 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        sum_n2 = tot[0];
        max_n = (uint32_t)tot[1];
    }
    if ((rc = arena_reserve(ctx, ctx->gp_L, sizeof(float) * (size_t)(sum_n2 ? sum_n2 : 1))) != LA3DM_OK) return rc;

    uint32_t max_leaves = 1u << (3 * (ctx->p.block_depth - 1));
    uint32_t tpb = (max_leaves + kWave - 1) / kWave, tpb_shift = 0;
    while ((1u << tpb_shift) < tpb) ++tpb_shift;
    GpArgs a;
    memset(&a, 0, sizeof(a));
    a.n_test_blk = s->n_test_blk;
    a.tpb_shift = tpb_shift;
    a.n_tasks = s->n_test_blk << tpb_shift;
    a.n_train_blk = s->n_train_blk;
    a.vmax = 0;
    a.vscratch = nullptr;
    if (max_n >= (uint32_t)kGpMfmaMinN && ctx->opt_gp_mode != 1) {
        a.vmax = (max_n + 31u) & ~31u;   // whole 32-row blocks: the solve stores its padded rows too
        if ((rc = arena_reserve(ctx, ctx->gp_v, sizeof(float) * (size_t)a.n_tasks * a.vmax * kWave)) != LA3DM_OK) return rc;
        a.vscratch = (float *)ctx->gp_v.ptr;
    }
    a.pts = (const float4 *)ctx->pts_scaled.ptr;
    a.train_off = s->train_off;
    a.nbr_range = (const uint2 *)ctx->nbr_range.ptr;
    a.nbr = s->nbr;
    a.l_off = (const unsigned long long *)ctx->gp_loff.ptr;
    a.order = (const uint32_t *)ctx->gp_order.ptr;
    a.totals = (const unsigned long long *)ctx->gp_totals.ptr;
    a.Lmat = (float *)ctx->gp_L.ptr;
    a.alpha_k = (float *)ctx->gp_alpha.ptr;
    a.blk_center = s->blk_center;
    a.leaf_off = s->leaf_off;
    a.leaf_key = s->leaf_key;
    a.m_ivar = s->alpha;
    a.ivar = s->beta;
    a.state = s->state;
    a.lut = ctx->d_lut;
    a.scale = (float)(1.73205 / (double)ctx->p.ell);
    a.sf2 = ctx->p.sf2;
    a.noise = ctx->p.noise;
    a.l = ctx->p.l;
    a.min_ivar = ctx->p.min_ivar;
    a.max_ivar = ctx->p.max_ivar;
    a.min_known_ivar = ctx->p.min_known_ivar;
    a.free_thresh = ctx->p.free_thresh;
    a.occupied_thresh = ctx->p.occupied_thresh;
    {
        const uint32_t n_nbr = 7u * s->n_test_blk;
        const uint32_t n_thr = s->n_train_pts > n_nbr ? s->n_train_pts : n_nbr;
        hipLaunchKernelGGL(gp_prepare, dim3((n_thr + 255) / 256), dim3(256), 0, stream, (const float4 *)s->train_xyzy,
                           (float4 *)ctx->pts_scaled.ptr, s->n_train_pts, a.scale, s->nbr, s->train_off,
                           (uint2 *)ctx->nbr_range.ptr, n_nbr);
    }
    const bool eigen_order = ctx->opt_gp_mode == 1;
    if (eigen_order && max_n > (uint32_t)kGpEigenMaxN) {
        ctx->err = "la3dm_gp_scan: gp_mode 1 (Eigen order on the VALU) takes training blocks of up to " + std::to_string(kGpEigenMaxN) +
                   " points; this scan holds one of " + std::to_string(max_n) + " (block_depth 4 and up belong to the matrix-core path: gp_mode 0)";
        return LA3DM_ERR_ARG;
    }
    if (s->n_train_blk && eigen_order) {
        const uint32_t nn = max_n ? max_n : 1u, n_tiny = nn < (uint32_t)kGpTrainTinyN ? nn : (uint32_t)kGpTrainTinyN;
        hipLaunchKernelGGL(gp_train_eigen_kernel, dim3(s->n_train_blk), dim3(kWave), gp_train_wave_lds(n_tiny), stream, a, 0, (int)n_tiny);
        if (nn > n_tiny)
            hipLaunchKernelGGL(gp_train_eigen_kernel, dim3(s->n_train_blk), dim3(kWave), gp_train_wave_lds(nn), stream, a, (int)n_tiny, (int)nn);
    } else if (s->n_train_blk) {
        const uint32_t nn = max_n < (uint32_t)kGpTrainLdsMaxN ? (max_n ? max_n : 1u) : (uint32_t)kGpTrainLdsMaxN;
        const uint32_t n_tiny = nn < (uint32_t)kGpTrainTinyN ? nn : (uint32_t)kGpTrainTinyN;
        hipLaunchKernelGGL(gp_train_wave_kernel, dim3(s->n_train_blk), dim3(kWave), gp_train_wave_lds(n_tiny), stream, a, 0, (int)n_tiny);
        if (nn > n_tiny)
            hipLaunchKernelGGL(gp_train_wave_kernel, dim3(s->n_train_blk), dim3(kWave), gp_train_wave_lds(nn), stream, a, (int)n_tiny, (int)nn);
        if (max_n > (uint32_t)kGpTrainLdsMaxN)
            hipLaunchKernelGGL(gp_train_kernel, dim3(s->n_train_blk), dim3(kWave), 0, stream, a);
    }
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (ctx->opt_time_kernel) {
        if (ctx->ev_used == ctx->ev_pool.size()) {
            std::pair<hipEvent_t, hipEvent_t> p;
            HIP_TRY(ctx, hipEventCreate(&p.first));
            HIP_TRY(ctx, hipEventCreate(&p.second));
            ctx->ev_pool.push_back(p);
        }
        ev = &ctx->ev_pool[ctx->ev_used++];
        HIP_TRY(ctx, hipEventRecord(ev->first, stream));
    }
    if (eigen_order) {
        hipLaunchKernelGGL(gp_predict_fuse_eigen_kernel, dim3(a.n_tasks), dim3(kWave), (size_t)(max_n ? max_n : 1u) * kWave * sizeof(float), stream, a);
    } else {
        const uint32_t rows = max_n < (uint32_t)kGpLdsRows ? (max_n ? max_n : 1u) : (uint32_t)kGpLdsRows;
        const size_t lds = rows * kWave * sizeof(float);
        // tiles without a large neighbour; then (if there is any large block) the tiles with one — every tile once
        // (the small launch in classes by the tile's largest neighbour block: LDS, and with it the waves per CU, per class)
        int lo = -1;
        for (uint32_t hi : {16u, 32u, (uint32_t)kGpLdsRows}) {
            const uint32_t h = hi < rows ? hi : rows;
            hipLaunchKernelGGL(gp_predict_fuse_small_kernel, dim3(a.n_tasks), dim3(kWave), h * kWave * sizeof(float), stream, a, lo, (int)h);
            lo = (int)h;
            if (h == rows) break;
        }
        if (max_n >= (uint32_t)kGpMfmaMinN)
            hipLaunchKernelGGL(gp_predict_fuse_kernel, dim3(a.n_tasks), dim3(kWave),
                               std::max<size_t>(lds, 3 * 32 * 36 * sizeof(float)) /* gp_solve_mfma's three tile buffers */, stream, a);
    }
    if (ev) HIP_TRY(ctx, hipEventRecord(ev->second, stream));
    HIP_TRY(ctx, hipGetLastError());
    if (out) {
        out->n_tiles = a.n_tasks;
        out->scratch_bytes = sizeof(float) * (size_t)sum_n2 + sizeof(float4) * (size_t)s->n_train_pts;
    }
    return LA3DM_OK;
}

int la3dm_gp_scan_host(la3dm_ctx *ctx, const la3dm_bgk_scan *s, la3dm_bgk_counters *out) {
    return scan_host_common(ctx, s, out, la3dm_gp_scan_device);
}

int la3dm_diag_mfma_chain(la3dm_ctx *ctx, const float *A, const float *B, int K, uint32_t *mismatches) {
    if (!ctx || !A || !B || !mismatches || K <= 0 || (K & 1)) return LA3DM_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d, sizeof(float) * (64 * (size_t)K + 1)));
    unsigned int *dm = (unsigned int *)(d + 64 * (size_t)K);
    hipError_t e = hipMemcpy(d, A, sizeof(float) * 32 * K, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 32 * (size_t)K, B, sizeof(float) * 32 * K, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dm, 0, 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(gp_diag_mfma_chain, dim3(1), dim3(kWave), 0, ctx->stream, d, d + 32 * (size_t)K, K, dm);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(mismatches, dm, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) {
        ctx->err = std::string("la3dm_diag_mfma_chain: ") + hipGetErrorString(e);
        return LA3DM_ERR_HIP;
    }
    return LA3DM_OK;
}

int la3dm_bgklv_scan_device(la3dm_ctx *ctx, const la3dm_lv_scan *s, void *stream_, la3dm_bgk_counters *out) {
    if (!ctx) return LA3DM_ERR_ARG;
    if (!s) {
        ctx->err = "lv scan: null";
        return LA3DM_ERR_ARG;
    }
    if (out) memset(out, 0, sizeof(*out));
    if (s->n_blk == 0) return LA3DM_OK;
    if (ctx->p.variant != 2) {
        ctx->err = "la3dm_bgklv_scan: the context was not created with variant = 2 (BGKLVOctoMap)";
        return LA3DM_ERR_ARG;
    }
    if (!s->sorted || !s->samples || !s->cell_off || !s->blk_center || !s->blk_cell0 || !s->alpha || !s->beta ||
        !s->state || (s->n_rays && !s->rays)) {
        ctx->err = "lv scan: null array pointer";
        return LA3DM_ERR_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LvArgs a;
    lv_fill_args(ctx, a, s->n_blk);
    a.samples = (const float4 *)s->samples;
    a.sorted = (const float4 *)s->sorted;
    a.rays = (const float4 *)s->rays;
    a.cell_off = s->cell_off;
    a.blk_center = s->blk_center;
    a.blk_cell0 = s->blk_cell0;
    a.alpha = s->alpha;
    a.beta = s->beta;
    a.state = s->state;
    for (int i = 0; i < 3; ++i) {
        a.cell_min[i] = s->cell_min[i];
        a.cell_dim[i] = s->cell_dim[i];
    }
    int rc = arena_reserve(ctx, ctx->lvp_totals, 16);
    if (rc != LA3DM_OK) return rc;
    if ((rc = lv_plan_launch(ctx, a, s->n_samples, (uint32_t *)ctx->lvp_totals.ptr, stream)) != LA3DM_OK) return rc;
    uint32_t totals[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(totals, ctx->lvp_totals.ptr, 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    if ((rc = lv_run_planned(ctx, a, totals, stream)) != LA3DM_OK) return rc;
    if (out) out->n_tiles = totals[0] + totals[3];
    return LA3DM_OK;
}

// ---- BGKLOctoMap (row f4): rows of 8 floats in s->train_xyzy, CSR over training blocks in s->train_off ----
int la3dm_bgkl_scan_device(la3dm_ctx *ctx, const la3dm_bgk_scan *s, void *stream_, la3dm_bgk_counters *out) {
    int rc = check_scan(ctx, s);
    if (rc != LA3DM_OK) return rc;
    if (out) memset(out, 0, sizeof(*out));
    if (s->n_test_blk == 0) return LA3DM_OK;
    if (ctx->p.variant != 3) {
        ctx->err = "la3dm_bgkl_scan: the context was not created with variant = 3 (BGKLOctoMap)";
        return LA3DM_ERR_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t max_leaves = 1u << (3 * (ctx->p.block_depth - 1));
    const uint32_t tpb = (max_leaves + kWave - 1) / kWave;
    uint32_t tpb_shift = 0;
    while ((1u << tpb_shift) < tpb) ++tpb_shift;
    BgklArgs a;
    a.rows = s->train_xyzy;
    a.row_off = s->train_off;
    a.nbr = s->nbr;
    a.blk_center = s->blk_center;
    a.leaf_off = s->leaf_off;
    a.leaf_key = s->leaf_key;
    a.alpha = s->alpha;
    a.beta = s->beta;
    a.state = s->state;
    a.lut = ctx->d_lut;
    a.n_test_blk = s->n_test_blk;
    a.tpb_shift = tpb_shift;
    a.n_tasks = s->n_test_blk << tpb_shift;
    a.sf2 = ctx->p.sf2;
    a.ell = ctx->p.ell;
    a.inv_ell = ctx->inv_ell;
    a.trig = ctx->opt_fast_trig == 3 ? 3 : 0;
    a.free_thresh = ctx->p.free_thresh;
    a.occupied_thresh = ctx->p.occupied_thresh;
    a.var_thresh = ctx->p.var_thresh;
    {
        // d >= ell for d = sqrtf(d2)  <=>  d2 >= hit_d2, the smallest fp32 whose (correctly rounded, monotone) root reaches ell
        const float ell = ctx->p.ell;
        float t = ell * ell;
        while (t > 0.0f && sqrtf(t) >= ell) t = nextafterf(t, 0.0f);
        while (!(sqrtf(t) >= ell) && t < INFINITY) t = nextafterf(t, INFINITY);
        a.hit_d2 = t;
    }
    // the rows' own terms of the line distance, once per row
    if (s->flags & LA3DM_SCAN_ROWS_PREPARED) {   // the caller's rows already are in the 12-float form (the device-resident map's)
        a.rowx = (const float4 *)s->train_xyzy;
    } else {
        if ((rc = arena_reserve(ctx, ctx->l_rowx, sizeof(float4) * 3 * (size_t)s->n_train_pts)) != LA3DM_OK) return rc;
        a.rowx = (const float4 *)ctx->l_rowx.ptr;
        if (s->n_train_pts)
            hipLaunchKernelGGL(bgkl_rows_prepare, dim3((s->n_train_pts + 255) / 256), dim3(256), 0, stream, a.rows, s->n_train_pts, (float4 *)ctx->l_rowx.ptr);
    }
    // tiles with more than `bgkl_split_rows` rows take the split path (bgkl_kernels.h); the others run while
    // the host waits for the item count
    BgklSplit sp;
    memset(&sp, 0, sizeof(sp));
    sp.threshold = ctx->opt_l_split_rows < 0 ? 0xFFFFFFFFu : (uint32_t)ctx->opt_l_split_rows;
    const bool split = ctx->opt_l_split_rows >= 0;
    if (split) {
        if ((rc = arena_reserve(ctx, ctx->l_task_item, sizeof(uint32_t) * 2 * (size_t)a.n_tasks)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_split_list, sizeof(uint32_t) * (size_t)a.n_tasks)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_counters, 16)) != LA3DM_OK) return rc;
        sp.task_item = (uint32_t *)ctx->l_task_item.ptr;
        sp.split_list = (uint32_t *)ctx->l_split_list.ptr;
        sp.counters = (uint32_t *)ctx->l_counters.ptr;
        HIP_TRY(ctx, hipMemsetAsync(sp.counters, 0, 16, stream));
        hipLaunchKernelGGL(bgkl_split_mark, dim3((a.n_tasks + 255) / 256), dim3(256), 0, stream, a, sp);
    }
    // few tiles (a scan of a few thousand points): latency counts, eight waves per tile; many tiles: the launch is
    // throughput-bound and one wave per tile is the cheaper form (measured: 764 tiles 2.5 -> 0.8 ms with eight
    // waves, 41 694 tiles 2.9 -> 5.6 ms)
    // accumulate mode ("bgk_sum", as for BGKOctoMap): 1 (default) = double sums, each neighbour's two sums rounded once —
    // no order to keep, so the split tiles need no scratch replay; 0 = the reference's fp32 running sums in row order
    const bool sum_f64 = ctx->opt_bgk_sum == 1;
    if (sum_f64) {
        if (a.n_tasks <= kLWideTiles)
            hipLaunchKernelGGL(bgkl_predict_fuse_f64<kLWaves>, dim3(a.n_tasks), dim3(kLWaves * kWave), 0, stream, a,
                               (const uint32_t *)sp.task_item);
        else
            hipLaunchKernelGGL(bgkl_predict_fuse_f64<1>, dim3(a.n_tasks), dim3(kWave), 0, stream, a, (const uint32_t *)sp.task_item);
    } else if (a.n_tasks <= kLWideTiles)
        hipLaunchKernelGGL(bgkl_predict_fuse_kernel<kLWaves>, dim3(a.n_tasks), dim3(kLWaves * kWave), 0, stream, a,
                           (const uint32_t *)sp.task_item);
    else
        hipLaunchKernelGGL(bgkl_predict_fuse_kernel<1>, dim3(a.n_tasks), dim3(kWave), 0, stream, a, (const uint32_t *)sp.task_item);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t head[2] = {0, 0};  // items, split tiles
    if (split) {
        HIP_TRY(ctx, hipMemcpyAsync(head, sp.counters, 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
    }
    const uint32_t n_items = head[0], n_split = head[1];
    size_t split_scratch = 0;
    if (n_items && sum_f64) {
        // order-free: per item two double sums per leaf (1 KB), added per (tile, neighbour) in item order
        if ((rc = arena_reserve(ctx, ctx->l_item_desc, sizeof(uint4) * (size_t)n_items)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_nb_first, sizeof(uint32_t) * 8 * (size_t)n_split)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_part64, sizeof(double2) * kWave * (size_t)n_items)) != LA3DM_OK) return rc;
        sp.item_desc = (uint4 *)ctx->l_item_desc.ptr;
        sp.nb_first = (uint32_t *)ctx->l_nb_first.ptr;
        sp.part64 = (double2 *)ctx->l_part64.ptr;
        hipLaunchKernelGGL(bgkl_split_items_wide, dim3(n_split), dim3(7 * kWave), 0, stream, a, sp);
        hipLaunchKernelGGL(bgkl_split_sum, dim3(n_items), dim3(kWave), 0, stream, a, sp);
        hipLaunchKernelGGL(bgkl_split_apply64, dim3(n_split), dim3(7 * kWave), 0, stream, a, sp);
        HIP_TRY(ctx, hipGetLastError());
        split_scratch = sizeof(double2) * kWave * (size_t)n_items;
    } else if (n_items) {
        split_scratch = sizeof(uint4) * (size_t)n_items * kLItemRows + sizeof(float) * (size_t)n_items * kLItemVals;
        if ((rc = arena_reserve(ctx, ctx->l_item_desc, sizeof(uint4) * (size_t)n_items)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_rowrec, sizeof(uint4) * (size_t)n_items * kLItemRows)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_batch_off, sizeof(uint32_t) * (size_t)n_items * kLBatches)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_item_hits, sizeof(uint32_t) * (size_t)n_items)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_bdesc, sizeof(uint4) * (size_t)n_items * kLBatches)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_nb_first, sizeof(uint32_t) * 8 * (size_t)n_split)) != LA3DM_OK) return rc;
        if ((rc = arena_reserve(ctx, ctx->l_part, sizeof(float2) * 7 * kWave * (size_t)n_split)) != LA3DM_OK) return rc;
        sp.item_desc = (uint4 *)ctx->l_item_desc.ptr;
        sp.rowrec = (uint4 *)ctx->l_rowrec.ptr;
        sp.batch_off = (uint32_t *)ctx->l_batch_off.ptr;
        sp.item_hits = (uint32_t *)ctx->l_item_hits.ptr;
        sp.bdesc = (uint4 *)ctx->l_bdesc.ptr;
        sp.nb_first = (uint32_t *)ctx->l_nb_first.ptr;
        sp.part = (float2 *)ctx->l_part.ptr;
        // every item owns kLItemVals value slots (64 KB): no read-back of the hit total, no second distance pass.  The
        // scratch is sized from a device counter: bound it (a low bgkl_split_rows on a large scan asks for 10^5 items).
        const size_t vals_bytes = sizeof(float) * (size_t)n_items * kLItemVals;
        if (vals_bytes > (64ull << 30)) {
            ctx->err = "la3dm_bgkl_scan: the split path would need " + std::to_string(vals_bytes >> 30) +
                       " GB of scratch (" + std::to_string(n_items) + " items of 256 rows); raise bgkl_split_rows";
            return LA3DM_ERR_OOM;
        }
        if ((rc = arena_reserve(ctx, ctx->l_vals, vals_bytes)) != LA3DM_OK) return rc;
        sp.vals = (float *)ctx->l_vals.ptr;
        hipLaunchKernelGGL(bgkl_split_items_wide, dim3(n_split), dim3(7 * kWave), 0, stream, a, sp);
        hipLaunchKernelGGL(bgkl_split_eval, dim3(n_items), dim3(kWave), 0, stream, a, sp);
        hipLaunchKernelGGL(bgkl_split_bdesc, dim3((n_items * kLBatches + 255) / 256), dim3(256), 0, stream, sp, n_items);
        hipLaunchKernelGGL(bgkl_split_kernelize, dim3(n_items), dim3(256), 0, stream, a, sp);
        if (ctx->opt_l_dense_add && vals_bytes <= (16ull << 30)) {   // (above 16 GB: the replay expands itself, no second 64 KB per item)
            // the replay's expansion done for all items at once, the ordered part left to two waves per chain
            if ((rc = arena_reserve(ctx, ctx->l_dense, sizeof(float) * (size_t)n_items * kLItemVals)) != LA3DM_OK) return rc;
            if ((rc = arena_reserve(ctx, ctx->l_labmask, sizeof(unsigned long long) * (size_t)n_items * kLBatches)) != LA3DM_OK) return rc;
            sp.dense = (float4 *)ctx->l_dense.ptr;
            sp.labmask = (unsigned long long *)ctx->l_labmask.ptr;
            hipLaunchKernelGGL(bgkl_split_expand, dim3(n_items * kLBatches), dim3(256), 0, stream, sp);
            hipLaunchKernelGGL(bgkl_split_add, dim3(n_split * 7), dim3(kWave * (2 + kLProducers)), 0, stream, a, sp);
        } else {
            hipLaunchKernelGGL(bgkl_split_fuse, dim3(n_split * 7), dim3(kWave * (2 + kLProducers)), 0, stream, a, sp);
        }
        hipLaunchKernelGGL(bgkl_split_apply, dim3(n_split), dim3(kWave), 0, stream, a, sp);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (out) {
        out->n_tiles = a.n_tasks;
        out->scratch_bytes = split_scratch + sizeof(float4) * 3 * (size_t)s->n_train_pts;
    }
    return LA3DM_OK;
}

int la3dm_bgkl_scan_host(la3dm_ctx *ctx, const la3dm_bgk_scan *s, la3dm_bgk_counters *out) {
    return scan_host_common(ctx, s, out, la3dm_bgkl_scan_device, 8);
}

int la3dm_bgklv_scan_host(la3dm_ctx *ctx, const la3dm_lv_scan *s, la3dm_bgk_counters *out) {
    if (!ctx) return LA3DM_ERR_ARG;
    if (!s) {
        ctx->err = "lv scan: null";
        return LA3DM_ERR_ARG;
    }
    if (out) memset(out, 0, sizeof(*out));
    if (s->n_blk == 0) return LA3DM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t ncell = (size_t)s->cell_dim[0] * s->cell_dim[1] * s->cell_dim[2];
    const size_t nnode = (size_t)s->n_blk << (3 * (ctx->p.block_depth - 1));
    struct Up {
        Arena *a;
        const void *src;
        size_t bytes;
    } ups[] = {
        {&ctx->lv_samples, s->samples, sizeof(float) * 4 * (size_t)s->n_samples},
        {&ctx->lv_sorted, s->sorted, sizeof(float) * 4 * (size_t)s->n_samples},
        {&ctx->lv_rays, s->rays, sizeof(float) * 8 * (size_t)s->n_rays},
        {&ctx->lv_cell, s->cell_off, sizeof(uint32_t) * (ncell + 1)},
        {&ctx->lv_center, s->blk_center, sizeof(float) * 3 * (size_t)s->n_blk},
        {&ctx->lv_cell0, s->blk_cell0, sizeof(int32_t) * 3 * (size_t)s->n_blk},
        {&ctx->lv_alpha, s->alpha, sizeof(float) * nnode},
        {&ctx->lv_beta, s->beta, sizeof(float) * nnode},
        {&ctx->lv_state, s->state, nnode},
    };
    int rc;
    for (auto &u : ups) {
        rc = arena_reserve(ctx, *u.a, u.bytes ? u.bytes : 16);
        if (rc != LA3DM_OK) return rc;
        if (u.bytes) HIP_TRY(ctx, hipMemcpyAsync(u.a->ptr, u.src, u.bytes, hipMemcpyHostToDevice, st));
    }
    la3dm_lv_scan d = *s;
    d.samples = (const float *)ctx->lv_samples.ptr;
    d.sorted = (const float *)ctx->lv_sorted.ptr;
    d.rays = (const float *)ctx->lv_rays.ptr;
    d.cell_off = (const uint32_t *)ctx->lv_cell.ptr;
    d.blk_center = (const float *)ctx->lv_center.ptr;
    d.blk_cell0 = (const int32_t *)ctx->lv_cell0.ptr;
    d.alpha = (float *)ctx->lv_alpha.ptr;
    d.beta = (float *)ctx->lv_beta.ptr;
    d.state = (uint8_t *)ctx->lv_state.ptr;
    rc = la3dm_bgklv_scan_device(ctx, &d, st, out);
    if (rc != LA3DM_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(s->alpha, d.alpha, sizeof(float) * nnode, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(s->beta, d.beta, sizeof(float) * nnode, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(s->state, d.state, nnode, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return LA3DM_OK;
}

int la3dm_kernel_times(la3dm_ctx *ctx, float *ms, uint32_t cap, uint32_t *n_out) {
    if (!ctx || !n_out) return LA3DM_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t n = (uint32_t)ctx->ev_used;
    for (uint32_t i = 0; i < n; ++i) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_pool[i].second));
        float t = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev_pool[i].first, ctx->ev_pool[i].second));
        if (ms && i < cap) ms[i] = t;
    }
    *n_out = n;
    ctx->ev_used = 0;
    return LA3DM_OK;
}

int la3dm_diag_sweep(la3dm_ctx *ctx, int what, uint32_t lo_bits, uint32_t hi_bits, uint64_t *mismatches) {
    if (!ctx || !mismatches || hi_bits < lo_bits) return LA3DM_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = arena_reserve(ctx, ctx->h_diag_out, 16);
    if (rc != LA3DM_OK) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync(ctx->h_diag_out.ptr, 0, 8, st));
    if (what == 10)
        hipLaunchKernelGGL(gp_exp_sweep_kernel, dim3(4096), dim3(256), 0, st, lo_bits, hi_bits, (unsigned long long *)ctx->h_diag_out.ptr);
    else
        hipLaunchKernelGGL(sweep_check_kernel, dim3(4096), dim3(256), 0, st, what, lo_bits, hi_bits,
                           (unsigned long long *)ctx->h_diag_out.ptr, ctx->p.ell, ctx->inv_ell, ctx->p.sf2);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&v, ctx->h_diag_out.ptr, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *mismatches = v;
    return LA3DM_OK;
}

int la3dm_diag_eval(la3dm_ctx *ctx, int op, const float *in, uint32_t n, float *out) {
    if (!ctx || !in || !out) return LA3DM_ERR_ARG;
    if (n == 0) return LA3DM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = arena_reserve(ctx, ctx->h_diag_in, sizeof(float) * (size_t)n);
    if (rc != LA3DM_OK) return rc;
    rc = arena_reserve(ctx, ctx->h_diag_out, sizeof(float) * (size_t)n);
    if (rc != LA3DM_OK) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_diag_in.ptr, in, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(diag_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, st, op, (const float *)ctx->h_diag_in.ptr,
                       (float *)ctx->h_diag_out.ptr, n, ctx->p.sf2, ctx->p.ell, ctx->p.free_thresh, ctx->p.occupied_thresh, ctx->p.var_thresh);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->h_diag_out.ptr, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return LA3DM_OK;
}

}  // extern "C"
