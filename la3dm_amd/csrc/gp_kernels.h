// gp_kernels.h — HIP kernels (gfx950, wave64) for the GPOctoMap per-scan path.
//
// Reference (CPU, Eigen):
//   GPRegressor::train      include/gpoctomap/gpregressor.h:42-51   K = Matern(x,x) + noise I, LLT, alpha = K^-1 y
//   GPRegressor::predict    include/gpoctomap/gpregressor.h:80-92   m = Ks^T alpha, v = L^-1 Ks, var = sf2 - diag(v^T v)
//   covMaterniso3           include/gpoctomap/gpregressor.h:114-117 (1 + a) exp(-a) sf2, a = |(1.73205/ell)(x - x')|
//   BCM Occupancy::update   src/gpoctomap/gpoctree_node.cpp:31-49
//   update loop             src/gpoctomap/gpoctomap.cpp:306-319 (unconditional, ExtendedBlock order)
//
// Numerics: every inner product is an fp32 FMA chain in ascending index order — exactly the order
// the oracle uses and, as measured on MI355X (tools/mfma/mfma_order.hip), the order
// v_mfma_f32_32x32x2_f32 accumulates in — so the results are bit-identical to the CPU restatement
// whether a chain runs on the VALU or on the matrix cores; exp() is the correctly rounded single-precision value
// (f64 library exp rounded once).  Eigen's own LLT/GEMV/packet-exp orders are unpinned.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgk_kernels.h"

namespace la3dm_dev {

struct GpArgs {
    const float4 *pts;          // training points scaled by (float)(1.73205/ell), label in w
    const uint32_t *train_off;
    const uint2 *nbr_range;     // [n_test_blk * 7] {first point, count}
    const int32_t *nbr;         // [n_test_blk * 7] training block index or -1
    const unsigned long long *l_off;  // [n_train_blk] offset (floats) of each block's N x N factor
    float *Lmat;                // Cholesky factors, row-major
    float *alpha_k;             // [n_train_pts] K^-1 y
    const float *blk_center;
    const uint32_t *leaf_off;
    const uint32_t *leaf_key;
    float *m_ivar;
    float *ivar;
    uint8_t *state;
    const float4 *lut;
    float *vscratch;            // [n_tasks][vmax][64] when a block exceeds the LDS capacity, else null
    const uint32_t *order;      // large training blocks, largest first (gp_factor_offsets)
    const unsigned long long *totals;   // [0] sum N^2, [1] max N, [2] large blocks, [3] small non-empty blocks (listed behind the large ones)
    uint32_t n_test_blk, tpb_shift, n_tasks, n_train_blk;
    uint32_t vmax;              // rows of v per task in vscratch
    float scale;                // (float)(1.73205 / ell)
    float sf2, noise, l, min_ivar, max_ivar, min_known_ivar, free_thresh, occupied_thresh;
};

constexpr int kGpTrainLdsMaxN = 128;  // blocks up to this size are factored by one wave in LDS
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float cr_expf_dev(float x) { return (float)exp((double)x); }

__device__ __forceinline__ float matern3_dev(float ax, float ay, float az, float bx, float by, float bz, float sf2) {
    const float dx = bx - ax, dy = by - ay, dz = bz - az;
    const float d = sqrtf(dx * dx + (dy * dy + dz * dz));
    return ((1 + d) * cr_expf_dev(-d)) * sf2;
}

__device__ __forceinline__ float matern3_fast(float ax, float ay, float az, float bx, float by, float bz, float sf2);

// scale the training points (x * s, gpregressor.h:115) and resolve neighbour ranges
__global__ void gp_prepare(const float4 *__restrict__ in, float4 *__restrict__ out, uint32_t n, float s,
                           const int32_t *__restrict__ nbr, const uint32_t *__restrict__ train_off,
                           uint2 *__restrict__ nbr_range, uint32_t n_nbr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float4 p = in[i];
        out[i] = make_float4(s * p.x, s * p.y, s * p.z, p.w);
    }
    if (i < n_nbr) {
        const int tb = nbr[i];
        uint2 r = make_uint2(0u, 0u);
        if (tb >= 0) {
            r.x = train_off[tb];
            r.y = train_off[tb + 1] - r.x;
        }
        nbr_range[i] = r;
    }
}

// exclusive scan of N_b^2 (one workgroup; n_train_blk is a few 10^4) + max N_b
// + the large blocks (gp_train_kernel's: one wave each, run time ~ N^3) listed largest first: the launch takes them in this
// order, so the longest factorisations start at once instead of wherever their index puts them (totals[2] = their number);
// the small non-empty blocks (gp_train_wave_kernel's) follow in the same list, largest first too (totals[3])
constexpr uint32_t kGpOffThreads = 1024;   // one workgroup: its threads walk n_blk / 1024 blocks each (256 threads: 86 us at configs[2]'s 31 k
                                            // blocks, a chain of dependent loads per thread)
__global__ __launch_bounds__(kGpOffThreads) void gp_factor_offsets(const uint32_t *__restrict__ train_off, uint32_t n_blk,
                                  unsigned long long *__restrict__ l_off, unsigned long long *__restrict__ totals,
                                  uint32_t *__restrict__ order) {
    __shared__ unsigned long long s_sum[kGpOffThreads];
    __shared__ uint32_t s_max[kGpOffThreads];
    __shared__ uint32_t s_cls[64];   // large blocks per size class (32-row blocks, capped)
    __shared__ uint32_t s_cls2[33];  // small non-empty blocks per size class (4 points)
    if (threadIdx.x < 64) s_cls[threadIdx.x] = 0;
    if (threadIdx.x < 33) s_cls2[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_blk + kGpOffThreads - 1u) / kGpOffThreads;
    const uint32_t b0 = tid * per, b1 = min(n_blk, b0 + per);
    unsigned long long sum = 0;
    uint32_t mx = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        const unsigned long long n = train_off[b + 1] - train_off[b];
        sum += n * n;
        mx = max(mx, (uint32_t)n);
        if (n > (unsigned long long)kGpTrainLdsMaxN) atomicAdd(&s_cls[min(63u, (uint32_t)((n + 31) >> 5))], 1u);
        else if (n) atomicAdd(&s_cls2[(uint32_t)((n + 3) >> 2)], 1u);
    }
    s_sum[tid] = sum;
    s_max[tid] = mx;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        uint32_t m = 0;
        for (int i = 0; i < (int)kGpOffThreads; ++i) {
            const unsigned long long t = s_sum[i];
            s_sum[i] = run;
            run += t;
            m = max(m, s_max[i]);
        }
        totals[0] = run;
        totals[1] = m;
        uint32_t first = 0;   // class c starts behind all larger classes
        for (int c = 63; c >= 0; --c) {
            const uint32_t k = s_cls[c];
            s_cls[c] = first;
            first += k;
        }
        totals[2] = first;
        for (int c = 32; c >= 0; --c) {   // the small blocks behind them, largest first as well
            const uint32_t k = s_cls2[c];
            s_cls2[c] = first;
            first += k;
        }
        totals[3] = first - (uint32_t)totals[2];
    }
    __syncthreads();
    unsigned long long run = s_sum[tid];
    for (uint32_t b = b0; b < b1; ++b) {
        const unsigned long long n = train_off[b + 1] - train_off[b];
        l_off[b] = run;
        run += n * n;
        if (n > (unsigned long long)kGpTrainLdsMaxN) order[atomicAdd(&s_cls[min(63u, (uint32_t)((n + 31) >> 5))], 1u)] = b;
        else if (n) order[atomicAdd(&s_cls2[(uint32_t)((n + 3) >> 2)], 1u)] = b;
    }
}

// GPRegressor::train for training blocks with N > 128: one wave64 per block, blocked 32 x 32, the factor built in
// place in the block's N x N slot of Lmat.  Every entry is the oracle's chain
//     L_ij = (K_ij - sum_{k<j} L_ik L_jk) / d_j,   d_j = sqrt(K_jj - sum_{k<j} L_jk^2),   k ascending:
// the part of each chain that runs over earlier block columns is a 32x32x2 MFMA sequence (A = -L[I][T],
// B = L[J][T]^T, T then k ascending — v_mfma_f32_32x32x2_f32 accumulates as a sequential fp32 FMA chain, see
// gp_solve_mfma), the part inside block column J continues on the VALU after a transpose through LDS to
// lane = row (diagonal block: unblocked Cholesky, row j broadcast with v_readlane; blocks below it: a forward
// substitution against the diagonal block, two row blocks per pass in the two half-waves).
// alpha = L^-T (L^-1 y) follows with lane = row inside a 32-row block and the solved part as wave-uniform scalars.
constexpr int kTrT = 33;  // padded tile row (floats)

__device__ __forceinline__ float rl(float v, int lane_idx) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_idx));
}

// Round 3.  The launch of configs[2] at depth 4 (3 364 large blocks, 1 255 of them with N > 352, N <= 528) took 4.7 ms.
// Timestamps inside the kernel (wall_clock64 per phase) for the N = 518 block: 2.1 ms in the alpha solve (lane = row reading
// its own row of the row-major factor: 32 cache lines per load, and a dependent global load of the pivot per step) -> now
// 32 x 32 tiles through LDS, the solved part read with v_readlane (0.5 ms); results of the substitutions stored one row per
// lane -> now 16-byte stores along the rows after a pass through LDS.  3.3 ms.  What remains is memory: every block re-reads
// its factor ~9 times from HBM (the row panels of each 32 x 32 block, ~4.6 MB per N = 518 block; 2 048 blocks in flight
// are 1.4 GB, nothing stays in the 4 MB L2s), VALU issue is at 30 %, the matrix cores at 12 % (profiles/r03/gp_train_pmc.txt).
// Tried and dropped (all bit-identical, all slower): four waves per block with row panel J in LDS and the pairs of a column
// dealt to the waves (one 1.1 ms block instead of 2.2 ms, but 4.3 wave-ms per block instead of 2.2: the diagonal block is
// redundant per wave, the alpha solve serial, 129 KB of LDS leave one block per CU -> 1 024 blocks in 3.45 ms); the same
// for the N > 352 blocks only (3.45 + 1.55 ms, the launches serialise).
constexpr int kTileF = 32 * 36;   // floats of one operand tile in LDS
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void gp_train_kernel(GpArgs a) {
    // MFMA operand tiles of L (rows x 32 columns of an earlier block column), fetched as four coalesced 16-byte loads per
    // lane and passed through LDS: read straight from the row-major factor, lane c = row c touches 32 cache lines per load
    // and the wave waits a memory round trip per k-pair (this kernel was latency bound: 9.6 ms for 4 726 blocks).
    // The accumulator -> [row][col] transposes (s_tile) happen after the operand loop and share the space.
    __shared__ __attribute__((aligned(16))) float s_lds[4 * kTileF];   // three operand / transpose tiles + the diagonal block (s_dg)
    float (*s_op)[32][36] = reinterpret_cast<float (*)[32][36]>(s_lds);
    float (*s_tile)[32][kTrT] = reinterpret_cast<float (*)[32][kTrT]>(s_lds);
    float (*s_dg)[36] = reinterpret_cast<float (*)[36]>(s_lds + 3 * kTileF);   // L[J][J], row-major, for the substitutions
    if (blockIdx.x >= (uint32_t)a.totals[2]) return;   // (the launch has one workgroup per training block; the large ones are listed)
    const uint32_t b = a.order[blockIdx.x];            // largest first
    const uint32_t p0 = a.train_off[b];
    const int N = (int)(a.train_off[b + 1] - p0);
    if (N <= kGpTrainLdsMaxN) return;  // small blocks: gp_train_wave_kernel
    float *L = a.Lmat + a.l_off[b];
    const float4 *x = a.pts + p0;
    const int lane = threadIdx.x;
    const int c = lane & 31, h = lane >> 5;
    const int nblk = (N + 31) >> 5;

    // K(i, j) for the 32 x 32 block (RI, RJ) in accumulator layout: rows 8g + 4h + j, column c; the padding
    // beyond N is the identity so that padded rows factor to d = 1 and contribute nothing
    auto kblock = [&](int RI, int RJ, f32x16 &C) {
        // every point is fetched before anything is evaluated (indices clamped, no branch per row: with one the 16 rows
        // were 16 dependent memory round trips)
        const int col = RJ + c;
        const float4 xc = x[min(col, N - 1)];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 xr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = 8 * q + u;
                xr[u] = x[min(RI + 8 * (r >> 2) + 4 * h + (r & 3), N - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = 8 * q + u;
                const int row = RI + 8 * (r >> 2) + 4 * h + (r & 3);
                float kv = matern3_fast(xr[u].x, xr[u].y, xr[u].z, xc.x, xc.y, xc.z, a.sf2);
                const bool in = row < N && col < N;
                if (row == col) kv = in ? kv + a.noise : 1.0f;
                else kv = in ? kv : 0.0f;
                C[r] = kv;
            }
        }
    };
    // accumulator layout -> tile[row][col]
    auto to_tile = [&](const f32x16 &C, int t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_tile[t][8 * (r >> 2) + 4 * h + (r & 3)][c] = C[r];
    };

    const int trow = lane >> 3, tcol = 4 * (lane & 7);
    auto load_tile = [&](float4 (&g)[4], int R, int T, bool on) {   // rows R .. R + 31, columns 32 T .. 32 T + 31 (all < N)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = R + 8 * i + trow;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on && row < N) __builtin_memcpy(&q, L + (size_t)row * N + 32 * T + tcol, 16);   // 4-byte aligned
            g[i] = q;
        }
    };
    // the diagonal tile of the last block may reach beyond column N - 1: element-wise there
    auto load_tile_edge = [&](float4 (&g)[4], int R, int T) {
        if (32 * T + 32 <= N) {
            load_tile(g, R, T, true);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = R + 8 * i + trow;
            float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (row < N && 32 * T + tcol + e < N) q[e] = L[(size_t)row * N + 32 * T + tcol + e];
            g[i] = make_float4(q[0], q[1], q[2], q[3]);
        }
    };
    auto store_to = [&](float (*tile)[36], const float4 (&g)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(&tile[8 * i + trow][tcol]) = g[i];
    };
    auto store_tile = [&](int t, const float4 (&g)[4]) { store_to(s_op[t], g); };
    auto read_from = [&](const float (*tile)[36], float (&o)[16]) {   // operand layout: lane (c, h) holds tile[c][2 m2 + h]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 q = *reinterpret_cast<const float4 *>(&tile[c][4 * i]);
            o[2 * i] = h ? q.y : q.x;
            o[2 * i + 1] = h ? q.w : q.z;
        }
    };
    auto read_ops = [&](int t, float (&o)[16]) { read_from(s_op[t], o); };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto group_sync = [&]() {   // the wave's own global accesses are ordered (lanes read what other lanes wrote)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (int J = 0; J < nblk; ++J) {
        const int RJ = 32 * J;
        // ---------------- diagonal block ----------------
        {
            float dg[32];  // lane = row (both half-waves hold the same rows): row (lane % 32) of L[J][J]
            f32x16 C;
            kblock(RJ, RJ, C);
            const int arow = RJ + c;
            {
                float4 g[4];
                if (J > 0) load_tile(g, RJ, 0, true);
                for (int T = 0; T < J; ++T) {
                    float lv[16];
                    store_tile(0, g);
                    wave_sync();
                    read_ops(0, lv);
                    if (T + 1 < J) load_tile(g, RJ, T + 1, true);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int m2 = 0; m2 < 16; ++m2) C = __builtin_amdgcn_mfma_f32_32x32x2f32(-lv[m2], lv[m2], C, 0, 0, 0);
                }
            }
            to_tile(C, 0);
            wave_sync();
#pragma unroll
            for (int w = 0; w < 32; ++w) dg[w] = s_tile[0][c][w];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float acc = dg[j];
#pragma unroll
                for (int w = 0; w < j; ++w) acc = __builtin_fmaf(-dg[w], rl(dg[w], j), acc);
                const float d = sqrtf(rl(acc, j));
                dg[j] = c == j ? d : acc / d;  // rows above the diagonal hold junk that is never read
            }
            if (h == 0 && arow < N) {
                float *dst = L + (size_t)arow * N + RJ;
#pragma unroll
                for (int w = 0; w < 32; ++w)
                    if (w <= c && RJ + w < N) dst[w] = dg[w];
            }
            // the block also goes to LDS: the substitutions below read L[J][J] rows as broadcast 16-byte reads (four entries
            // per LDS instruction instead of a v_readlane per entry), and the 32 registers are free during the operand loop
            // (where the kernel spilled: 35 scratch accesses per step)
            if (h == 0) {
#pragma unroll
                for (int w4 = 0; w4 < 8; ++w4)
                    *reinterpret_cast<float4 *>(&s_dg[c][4 * w4]) = make_float4(dg[4 * w4], dg[4 * w4 + 1], dg[4 * w4 + 2], dg[4 * w4 + 3]);
            }
            wave_sync();
        }
        // ---------------- blocks below the diagonal, two per pass ----------------
        for (int I = J + 1; I < nblk; I += 2) {
            const int RI0 = 32 * I, RI1 = 32 * (I + 1);
            const bool two = I + 1 < nblk;
            f32x16 C0, C1;
            kblock(RI0, RJ, C0);
            if (two) kblock(RI1, RJ, C1);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) C1[r] = 0.0f;
            }
            float4 g0[4], g1[4], gb[4];
            if (J > 0) {
                load_tile(g0, RI0, 0, true);
                load_tile(g1, RI1, 0, two);
                load_tile(gb, RJ, 0, true);
            }
            for (int T = 0; T < J; ++T) {
                float a0[16], a1[16], bv[16];
                store_tile(0, g0);
                store_tile(1, g1);
                store_tile(2, gb);
                wave_sync();
                read_ops(0, a0);
                read_ops(1, a1);
                read_ops(2, bv);
                if (T + 1 < J) {   // the next tiles are in flight while this block's 32 MFMAs run
                    load_tile(g0, RI0, T + 1, true);
                    load_tile(g1, RI1, T + 1, two);
                    load_tile(gb, RJ, T + 1, true);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m2 = 0; m2 < 16; ++m2) {
                    C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(-a0[m2], bv[m2], C0, 0, 0, 0);
                    C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(-a1[m2], bv[m2], C1, 0, 0, 0);
                }
            }
            to_tile(C0, 0);
            to_tile(C1, 1);
            wave_sync();
            float rw[32];  // lane = row: half 0 -> block I, half 1 -> block I + 1
#pragma unroll
            for (int w = 0; w < 32; ++w) rw[w] = s_tile[h][c][w];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float lj[32];   // row j of L[J][J] up to the diagonal (uniform addresses: broadcast reads)
#pragma unroll
                for (int w4 = 0; w4 <= j / 4; ++w4) {
                    const float4 q = *reinterpret_cast<const float4 *>(&s_dg[j][4 * w4]);
                    lj[4 * w4] = q.x, lj[4 * w4 + 1] = q.y, lj[4 * w4 + 2] = q.z, lj[4 * w4 + 3] = q.w;
                }
                float acc = rw[j];
#pragma unroll
                for (int w = 0; w < j; ++w) acc = __builtin_fmaf(-rw[w], lj[w], acc);
                rw[j] = acc / lj[j];
            }
            // back through LDS: 16-byte stores along the rows instead of one row per lane
#pragma unroll
            for (int w = 0; w < 32; ++w) s_tile[h][c][w] = rw[w];
            wave_sync();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !two) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int orow = (t ? RI1 : RI0) + 8 * i + trow;
                    const float4 q = make_float4(s_tile[t][8 * i + trow][tcol], s_tile[t][8 * i + trow][tcol + 1],
                                                 s_tile[t][8 * i + trow][tcol + 2], s_tile[t][8 * i + trow][tcol + 3]);
                    if (orow < N) __builtin_memcpy(L + (size_t)orow * N + RJ + tcol, &q, 16);   // (columns RJ .. RJ + 31 < RI0 <= N)
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the block column is read back (as MFMA operands)
        group_sync();
    }

    // ---------------- alpha = llt.solve(y) (gpregressor.h:48) ----------------
    // forward: z_j = (y_j - sum_{k<j} L_jk z_k) / L_jj, k ascending.  lane = row inside a 32-row block (both half-waves
    // alike).  Round 3: the factor comes in 32 x 32 tiles through LDS (coalesced 16-byte loads) and the solved part as one
    // register per tile read with v_readlane: with lane = row reading its own row from the row-major factor every load
    // touched 32 cache lines, and the 32 steps inside a block each waited for a global load of the pivot (2.1 of the kernel's
    // 4.7 ms for N = 528).  Same chains, same order.
    float *al = a.alpha_k + p0;
    auto fwd_tile = [&](const float (*tile)[36], float zc, float &acc) {
        float lt[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 q = *reinterpret_cast<const float4 *>(&tile[c][4 * i]);
            lt[4 * i] = q.x, lt[4 * i + 1] = q.y, lt[4 * i + 2] = q.z, lt[4 * i + 3] = q.w;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(-lt[k], rl(zc, k), acc);
    };
    auto fwd_diag = [&](const float (*tile)[36], int K, float &acc) {
        float lt[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 q = *reinterpret_cast<const float4 *>(&tile[c][4 * i]);
            lt[4 * i] = q.x, lt[4 * i + 1] = q.y, lt[4 * i + 2] = q.z, lt[4 * i + 3] = q.w;
        }
#pragma unroll
        for (int w = 0; w < 32; ++w) {
            const float dd = 32 * K + w < N ? rl(lt[w], w) : 1.0f;
            const float z = rl(acc, w) / dd;
            if (c == w) acc = z;
            else if (c > w) acc = __builtin_fmaf(-lt[w], z, acc);   // (rows beyond N hold zeros)
        }
    };
    auto bwd_tile = [&](const float (*tile)[36], int T, float zc, float &acc) {
        float lt[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) lt[k] = tile[k][c];
#pragma unroll
        for (int k = 31; k >= 0; --k)
            if (32 * T + k < N) acc = __builtin_fmaf(-lt[k], rl(zc, k), acc);   // wave-uniform
    };
    auto bwd_diag = [&](const float (*tile)[36], int K, float &acc) {
        float lt[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) lt[k] = tile[k][c];
#pragma unroll
        for (int w = 31; w >= 0; --w) {
            if (32 * K + w >= N) continue;  // wave-uniform: padded rows take no part
            const float dd = rl(lt[w], w);
            const float v = rl(acc, w) / dd;
            if (c == w) acc = v;
            else if (c < w) acc = __builtin_fmaf(-lt[w], v, acc);
        }
    };
    for (int K = 0; K < nblk; ++K) {
        const int row = 32 * K + c;
        const bool valid = row < N;
        float acc = valid ? x[row].w : 0.0f;
        float4 g[4];
        float zn = 0.0f;
        if (K > 0) {
            load_tile(g, 32 * K, 0, true);
            zn = al[c];
        } else load_tile_edge(g, 0, 0);
        for (int T = 0; T <= K; ++T) {
            store_tile(0, g);
            wave_sync();
            const float zc = zn;
            if (T + 1 < K) {   // the next tile is in flight while this one is used
                load_tile(g, 32 * K, T + 1, true);
                zn = al[32 * (T + 1) + c];
            } else if (T + 1 == K) load_tile_edge(g, 32 * K, K);
            if (T < K) fwd_tile(s_op[0], zc, acc);
            else fwd_diag(s_op[0], K, acc);
            __builtin_amdgcn_wave_barrier();   // (the tile is in registers by now: the next one may be stored over it)
        }
        if (h == 0 && valid) al[row] = acc;
        group_sync();
    }
    // backward: alpha_j = (z_j - sum_{k>j} L_kj alpha_k) / L_jj, k descending: tile (T, K) holds rows k of block T and this
    // block's columns; lane = column reads it by rows (conflict-free)
    for (int K = nblk - 1; K >= 0; --K) {
        const int row = 32 * K + c;
        const bool valid = row < N;
        float acc = valid ? al[row] : 0.0f;
        float4 g[4];
        float zn = 0.0f;
        if (K < nblk - 1) {
            load_tile(g, 32 * (nblk - 1), K, true);
            zn = 32 * (nblk - 1) + c < N ? al[32 * (nblk - 1) + c] : 0.0f;
        } else load_tile_edge(g, 32 * K, K);
        for (int T = nblk - 1; T >= K; --T) {
            store_tile(0, g);
            wave_sync();
            const float zc = zn;
            if (T - 1 > K) {
                load_tile(g, 32 * (T - 1), K, true);
                zn = al[32 * (T - 1) + c];
            } else if (T - 1 == K) load_tile_edge(g, 32 * K, K);
            if (T > K) bwd_tile(s_op[0], T, zc, acc);
            else bwd_diag(s_op[0], K, acc);
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_wave_barrier();
        if (h == 0 && valid) al[row] = acc;
        group_sync();
    }
}

// Same arithmetic as gp_train_kernel for blocks with N <= 128: one wave64 per training block, the lower
// triangle packed in LDS (row i at i(i+1)/2), lane = row (two rows per lane above 64), no workgroup barriers.
// The factor is written to its global slot once, for the predict kernel.
// The launch takes the blocks with n_lo < N <= n_hi (its LDS is sized for n_hi): the waves are latency bound, what the
// launch delivers is how many of them a CU holds, and sized for the largest small block (14 KB at N = 79) that was 11.
constexpr int kGpTrainTinyN = 32;   // blocks up to this size get a launch of their own (2.8 KB of LDS each)
__global__ __launch_bounds__(kWave) void gp_train_wave_kernel(GpArgs a, int n_lo, int n_hi) {
    extern __shared__ __attribute__((aligned(16))) float s_l[];  // [N(N+1)/2 + N] + the block's points [N] float4 (gp_train_wave_lds)
    if (blockIdx.x >= (uint32_t)a.totals[3]) return;   // (one workgroup per training block; the small non-empty ones are listed)
    const uint32_t b = a.order[(uint32_t)a.totals[2] + blockIdx.x];   // largest first: the launch ends with the short ones
    const uint32_t p0 = a.train_off[b];
    const int N = (int)(a.train_off[b + 1] - p0);
    if (N <= n_lo || N > n_hi) return;  // (n_hi <= kGpTrainLdsMaxN; larger blocks: gp_train_kernel)
    const int lane = threadIdx.x;
    const float4 *x = a.pts + p0;
    float *Lg = a.Lmat + a.l_off[b];
    const int T = (N * (N + 1)) / 2;
    float *zs = s_l + T;  // right-hand side / solution
    float4 *s_x = reinterpret_cast<float4 *>(s_l + ((T + N + 3) & ~3));
    auto tri = [](int i, int j) { return (i * (i + 1)) / 2 + j; };
    // Round 3 (configs[2] at the YAML's depth 3 is 31 k blocks of 9 points on average, and this kernel took 0.86 ms): the
    // points go to LDS first (one coalesced load; before, every kernel value waited for its own global load of x_j) and
    // the N(N+1)/2 kernel values are dealt to the lanes by their packed index e = i(i+1)/2 + j (before, lane i computed
    // row i: N values in sequence where N(N+1)/128 do).
    for (int i = lane; i < N; i += kWave) {
        const float4 xi = x[i];
        s_x[i] = xi;
        zs[i] = xi.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // K(i, j), i >= j, + noise on the diagonal
    for (int e = lane; e < T; e += kWave) {
        int i = (int)((__builtin_sqrtf((float)(8 * e + 1)) - 1.0f) * 0.5f);   // row of packed index e, then exact
        if ((i * (i + 1)) / 2 > e) --i;
        if (((i + 1) * (i + 2)) / 2 <= e) ++i;
        const int j = e - (i * (i + 1)) / 2;
        const float4 xi = s_x[i], xj = s_x[j];
        float kv = matern3_fast(xi.x, xi.y, xi.z, xj.x, xj.y, xj.z, a.sf2);
        if (i == j) kv = kv + a.noise;
        s_l[e] = kv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < N; ++j) {
        // rows j, j + 1, ... of column j, lane by lane; lane 0's row is the diagonal entry: its chain
        // K_jj - sum_k L_jk L_jk is the same FMA sequence as the other rows' with i = j (before, every lane ran it as a
        // separate loop ahead of its own row: twice the LDS reads and FMAs per column)
        float d = 0.0f;
        for (int i0 = j; i0 < N; i0 += kWave) {
            const int i = i0 + lane;
            float acc = 0.0f;
            if (i < N) {
                // eight steps of the chain at a time, their 16 LDS reads issued together (one round trip per step before:
                // 120 cycles per step measured, 100 us of the N = 64 block's 140)
                const float *ri = s_l + tri(i, 0), *rj = s_l + tri(j, 0);
                acc = ri[j];
                int k = 0;
                for (; k + 8 <= j; k += 8) {
                    float li[8], lj[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        li[u] = ri[k + u];
                        lj[u] = rj[k + u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(-li[u], lj[u], acc);
                }
                for (; k < j; ++k) acc = __builtin_fmaf(-ri[k], rj[k], acc);
            }
            if (i0 == j) d = sqrtf(rl(acc, 0));
            if (i < N && i > j) s_l[tri(i, j)] = acc / d;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) s_l[tri(j, j)] = d;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // forward then backward substitution, right-looking (chains over k ascending / descending), lane = row: rows lane and
    // lane + 64 of the right-hand side live in two registers, the solved entry is broadcast with v_readlane, the only LDS
    // access per step and row is the factor's entry (before: right-hand side in LDS, two barriers and ~450 cycles per step)
    const int r0 = lane, r1 = lane + kWave;
    float acc0 = r0 < N ? zs[r0] : 0.0f, acc1 = r1 < N ? zs[r1] : 0.0f;
    const float *row0 = s_l + tri(r0, 0), *row1 = s_l + tri(r1, 0);
    for (int j = 0; j < N; ++j) {
        const float z = (j < kWave ? rl(acc0, j) : rl(acc1, j - kWave)) / s_l[tri(j, j)];
        if (r0 == j) acc0 = z;
        else if (r0 > j && r0 < N) acc0 = __builtin_fmaf(-row0[j], z, acc0);
        if (r1 == j) acc1 = z;
        else if (r1 > j && r1 < N) acc1 = __builtin_fmaf(-row1[j], z, acc1);
    }
    for (int j = N - 1; j >= 0; --j) {
        const float v = (j < kWave ? rl(acc0, j) : rl(acc1, j - kWave)) / s_l[tri(j, j)];
        const float *rowj = s_l + tri(j, 0);
        if (r0 == j) acc0 = v;
        else if (r0 < j) acc0 = __builtin_fmaf(-rowj[r0], v, acc0);
        if (r1 == j) acc1 = v;
        else if (r1 < j) acc1 = __builtin_fmaf(-rowj[r1], v, acc1);
    }
    if (r0 < N) a.alpha_k[p0 + r0] = acc0;
    if (r1 < N) a.alpha_k[p0 + r1] = acc1;
    for (int e = lane; e < T; e += kWave) {   // the factor to its global slot (row-major N x N, lower triangle)
        int i = (int)((__builtin_sqrtf((float)(8 * e + 1)) - 1.0f) * 0.5f);
        if ((i * (i + 1)) / 2 > e) --i;
        if (((i + 1) * (i + 2)) / 2 <= e) ++i;
        Lg[(size_t)i * N + (e - (i * (i + 1)) / 2)] = s_l[e];
    }
}
// dynamic LDS of gp_train_wave_kernel for blocks of up to nn points
__host__ __device__ constexpr size_t gp_train_wave_lds(uint32_t nn) {
    return sizeof(float) * (((size_t)nn * (nn + 1) / 2 + nn + 3) / 4 * 4 + 4 * (size_t)nn);
}

// GP node update, src/gpoctomap/gpoctree_node.cpp:36-49 (double expression for ivar, double exp), in two parts: the
// reference classifies after every update, and each classification overwrites the one before — only the last one of a
// leaf's (up to seven) updates is ever seen.  So the updates carry (m_ivar, ivar) and whether the LAST one left the node
// unknown, and the logistic with its f64 exp (~100 instructions) runs once per leaf instead of once per neighbour.
__device__ __forceinline__ void gp_node_accumulate_dev(const GpArgs &a, float &m_ivar, float &ivar, bool &unknown, float new_m,
                                                       float new_var) {
    ivar = (float)((double)ivar + (1.0 / (double)new_var - (double)a.sf2));
    m_ivar += new_m / new_var;
    unknown = ivar < a.min_known_ivar;
    if (!unknown) ivar = ivar > a.max_ivar ? a.max_ivar : ivar;
}
__device__ __forceinline__ uint8_t gp_node_state_dev(const GpArgs &a, float m_ivar, bool unknown) {
    if (unknown) return 2;
    const float p = 1.0f / (1.0f + (float)exp((double)(-a.l * m_ivar / a.max_ivar)));
    return p > a.occupied_thresh ? 1 : (p < a.free_thresh ? 0 : 2);
}

// exp(x) for x in [-60, 0], correctly rounded to f32 in all but ~1e-7 of the cases (the same contract
// as the oracle's (float)exp((double)x)): x = n ln2 + r, |r| <= ln2/2, degree-13 Taylor/Horner in
// f64 (truncation < 2e-17 relative), scaled by 2^n through the exponent field.
__device__ __forceinline__ float exp_cr_dev(float xf) {
    const double x = (double)xf;
    const double n = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);   // ln2 high part
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);          // ln2 low part
    double p = 1.6059043836821613e-10;                             // 1/13!
    p = __builtin_fma(p, r, 2.08767569878680990e-09);              // 1/12!
    p = __builtin_fma(p, r, 2.50521083854417188e-08);              // 1/11!
    p = __builtin_fma(p, r, 2.75573192239858907e-07);              // 1/10!
    p = __builtin_fma(p, r, 2.75573192239858907e-06);              // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873016e-05);              // 1/8!
    p = __builtin_fma(p, r, 1.98412698412698413e-04);              // 1/7!
    p = __builtin_fma(p, r, 1.38888888888888889e-03);              // 1/6!
    p = __builtin_fma(p, r, 8.33333333333333333e-03);              // 1/5!
    p = __builtin_fma(p, r, 4.16666666666666667e-02);              // 1/4!
    p = __builtin_fma(p, r, 1.66666666666666667e-01);              // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const long long bits = (long long)__double_as_longlong(p) + ((long long)(int)n << 52);  // p in [0.7, 1.5], n >= -87
    return (float)__longlong_as_double(bits);
}

// exhaustive check of exp_cr_dev against the double-precision library function rounded to float (what the restatement's
// cr_expf computes), over the fp32 bit patterns [lo_bits, hi_bits] (la3dm_diag_sweep, what = 10)
__global__ void gp_exp_sweep_kernel(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mismatch) {
    const uint64_t n = (uint64_t)hi_bits - lo_bits + 1;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        bad += exp_cr_dev(x) == (float)exp((double)x) ? 0 : 1;
    }
    if (bad) atomicAdd(mismatch, bad);
}

// The distance's square root is sqrt_cr (bgk_kernels.h: hardware estimate + one residual fix-up, 6 instructions, equal to
// sqrtf on {0} U [2^-100, 2^100] — swept) instead of the compiler's IEEE sequence with its denormal pre-scaling (~12, and a
// branch around it measured +8 % on the depth-3 kernel).  Outside that range it may be off: d^2 > 2^100 does not occur, and a
// non-zero d^2 < 2^-100 means d < 1e-15, where (1 + d) exp(-d) is 1.0f whatever small value comes back.
__device__ __forceinline__ float matern3_fast(float ax, float ay, float az, float bx, float by, float bz, float sf2) {
    const float dx = bx - ax, dy = by - ay, dz = bz - az;
    const float d = sqrt_cr(dx * dx + (dy * dy + dz * dz));
    return ((1 + d) * exp_cr_dev(-d)) * sf2;
}

// ---------------------------------------------------------------------------
// v = L^-1 Ks for training blocks with N > 64 on the matrix cores, bit-identical to the FMA chains.
//
// v_mfma_f32_32x32x2_f32 accumulates D = C + A B as a chain of fp32 FMAs over k ascending (measured: 0 mismatches
// against fmaf chains, tools/mfma/mfma_order.hip), so a blocked forward substitution whose off-diagonal updates
//      C[K] = Ks[K] - sum_{J<K} L[K][J] V[J]            (J ascending, k ascending inside a block)
// run on MFMA reproduces  acc = fma(-L_ki, v_i, acc), i = 0..k-1  of the oracle exactly; the 32 x 32 diagonal
// blocks continue each chain on the VALU.  Layout: one wave handles the tile's 64 leaves as two 32-column
// accumulators; lane (c = lane % 32, h = lane / 32) owns rows 8g + 4h + j of a block for leaves c and 32 + c, so the
// serial row order alternates between the half-waves every four rows: chains (m, sum v^2) and the freshly solved
// v are handed over with v_permlane32_swap_b32 (the owner half's value broadcast to both halves).  V lives in the task's global scratch [row][64] (L2 resident),
// which is also the B operand of the MFMAs; L is read in place (A operand, negated on load).
// Returns m = Ks^T alpha and ss = sum v_k^2 for leaf = lane.
// ---------------------------------------------------------------------------
// value of half-wave `half` (lanes 32*half .. 32*half+31, by lane % 32) in BOTH half-waves: one
// v_permlane32_swap_b32 (gfx950; measured semantics: result[0] = {a.lo, b.lo}, result[1] = {a.hi, b.hi})
__device__ __forceinline__ float bcast_half(float v, int half) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(half ? r[1] : r[0]);
}

// LDS written by some lanes of a wave and read by others: the LDS queue of a wave is in order, so only the compiler has to
// keep the order.  (A release fence, even at wavefront scope, also waits for every global load in flight — in the
// tile loop that is the prefetch of the tile after next: the loop ran at memory latency.)
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }
// Timing experiments (WRONG results, never built by default): -DLA3DM_GP_EXP_V=1 reads every B operand from V's first slab
// (what the kernel costs when V is cache-hot), 3 also drops the diagonal solve's FMA chains, 4 also replaces the kernel
// evaluation by one FMA.  They located the time of DESIGN.md 3.3 (tile loop vs evaluation vs solve).
#ifndef LA3DM_GP_EXP_V
#define LA3DM_GP_EXP_V 0
#endif
__device__ __forceinline__ void gp_solve_mfma(const GpArgs &a, const float *__restrict__ L, const float4 *__restrict__ x,
                                              const float *__restrict__ al, const int N, const float tx, const float ty,
                                              const float tz, float *__restrict__ vg, float *lds, const int lane,
                                              float &mj_out, float &ss_out) {
    const int c = lane & 31, h = lane >> 5;
    float mj = 0.0f, ss = 0.0f;   // lane = leaf: the two chains of this lane's column
    const int nblk = (N + 31) >> 5;
    for (int K = 0; K < nblk; ++K) {
        const int R0 = 32 * K;
        f32x16 C0, C1;
        // the diagonal tile L[K][K] (used after the MFMAs): fetched like the A tiles, in flight during the kernel evaluations
        float (*s_d)[36] = reinterpret_cast<float (*)[36]>(lds + 2 * 32 * 36);
        float4 gd[4];
        {
            const int trow_ = lane >> 3, tcol_ = 4 * (lane & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = R0 + 8 * i + trow_, col = R0 + tcol_;
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < N) {
                    if (col + 3 < N) {
                        __builtin_memcpy(&q, L + (size_t)row * N + col, 16);
                    } else {   // the last block row: the tile sticks out of the matrix
                        const float *p = L + (size_t)row * N;
                        q.x = col < N ? p[col] : 0.f;
                        q.y = col + 1 < N ? p[col + 1] : 0.f;
                        q.z = col + 2 < N ? p[col + 2] : 0.f;
                    }
                } else {   // padded row: identity
                    const int rr = 8 * i + trow_;
                    q.x = rr == tcol_ ? 1.f : 0.f;
                    q.y = rr == tcol_ + 1 ? 1.f : 0.f;
                    q.z = rr == tcol_ + 2 ? 1.f : 0.f;
                    q.w = rr == tcol_ + 3 ? 1.f : 0.f;
                }
                gd[i] = q;
            }
        }
        {
            // Ks(k, j) = k(x_k, xs_j) with lane = leaf j, the 32 rows of the block in registers (training points and alpha
            // are wave-uniform: scalar loads), m = Ks^T alpha continued row by row; then into accumulator layout
            float ks[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                // no branch per row (the 32 evaluations interleave): a padded row evaluates the last point and is zeroed
                const int row = R0 + r, rowc = min(row, N - 1);
                const bool valid = row < N;   // wave-uniform
                const float4 xr = x[rowc];
                const float kv = LA3DM_GP_EXP_V == 4 ? xr.x * tx + xr.y : matern3_fast(xr.x, xr.y, xr.z, tx, ty, tz, a.sf2);
                const float av_ = al[rowc];
                ks[r] = valid ? kv : 0.0f;
                mj = __builtin_fmaf(ks[r], valid ? av_ : 0.0f, mj);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ks[8 * g + j]), __float_as_uint(ks[8 * g + 4 + j]),
                                                                     false, false);
                    C0[4 * g + j] = __uint_as_float(sw[0]);   // rows 8g + 4h + j of leaf c
                    C1[4 * g + j] = __uint_as_float(sw[1]);   // ... of leaf 32 + c
                }
            }
        }
        {
            const int trow_ = lane >> 3, tcol_ = 4 * (lane & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(&s_d[8 * i + trow_][tcol_]) = gd[i];
            lds_order();
            __builtin_amdgcn_wave_barrier();
        }
        // off-diagonal blocks on the matrix cores: C -= L[K][J] V[J].  The A operand wants lane c = row R0 + c, i.e. 32
        // different rows (cache lines) per load if it is read straight from the row-major factor — the address unit then
        // spends ~32 cycles per load and bounds the kernel.  The 32 x 32 tile is fetched as four coalesced 16-byte loads
        // per lane instead (8 rows x 128 B per instruction), negated, passed through LDS ([row][36], two buffers) and
        // read back in operand layout.  Tile J + 2 is in flight and tile J + 1 in LDS while the 32 MFMAs of block J run.
        {
            float (*s_t)[32][36] = reinterpret_cast<float (*)[32][36]>(lds);
            const int trow = lane >> 3, tcol = 4 * (lane & 7);
            float4 g[4];
            // per-lane element offsets of the four tile rows (rows beyond N re-read the last row: those accumulator rows are
            // zeroed before the diagonal solve) and of the B operand inside a 32-row slab of V; the bases stay scalar
            uint32_t loff[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) loff[i] = (uint32_t)min(R0 + 8 * i + trow, N - 1) * (uint32_t)N + (uint32_t)tcol;
            const uint32_t boff = (uint32_t)h * kWave + 2u * (uint32_t)c;
            auto load_tile = [&](int J) {
                const float *Lb = L + 32 * J;
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_memcpy(&g[i], Lb + loff[i], 16);   // 4-byte aligned
            };
            // LDS copy of a tile: row r holds its even columns at [r][0..15] and its odd columns at [r][16..31], so that lane
            // (c, h) finds its 16 operand values (columns 2 m2 + h) contiguous; the sign of the update sits in V (stored
            // negated), the tile goes through unchanged
            auto store_tile = [&](int buf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float *r_ = &s_t[buf][8 * i + trow][tcol >> 1];   // (scalar stores: ds_write2_b32 takes x and z where they are)
                    r_[0] = g[i].x; r_[1] = g[i].z;
                    r_[16] = g[i].y; r_[17] = g[i].w;
                }
            };
            // operands of half a tile (8 MFMA steps): the A values from the LDS copy of the tile, the B values from V
            auto read_half = [&](int J, int buf, int s, float (&A_)[8], float2 (&B_)[8]) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 q = *reinterpret_cast<const float4 *>(&s_t[buf][c][16 * h + 8 * s + 4 * i]);
                    A_[4 * i] = q.x; A_[4 * i + 1] = q.y; A_[4 * i + 2] = q.z; A_[4 * i + 3] = q.w;
                }
                const float *Vb = vg + (size_t)(LA3DM_GP_EXP_V ? 0 : 32 * J) * kWave;
#pragma unroll
                for (int m = 0; m < 8; ++m) B_[m] = *reinterpret_cast<const float2 *>(Vb + (boff + 2u * (8 * s + m) * kWave));
            };
            if (K > 0) {
                // One register set per half tile and no second copy: the operands of tile J + 1's first half are fetched
                // into the registers of tile J's first half as soon as its 16 MFMAs are issued, and arrive while the second
                // half's 16 run (with a full second operand set the kernel sat at 256 VGPRs and the allocator started
                // copying prefetched registers around, each copy a wait for the memory round trip).
                float a0[8], a1[8];
                float2 b0[8], b1[8];
                // The tile registers live inside one iteration only (fetch at the top, LDS store at the bottom, two tiles
                // ahead of the one whose MFMAs run): carried around the loop edge they were split and copied by the
                // allocator, each copy a wait for the fetch just issued.
                load_tile(0);
                store_tile(0);
                if (K > 1) {
                    load_tile(1);
                    store_tile(1);
                }
                lds_order();
                __builtin_amdgcn_wave_barrier();
                read_half(0, 0, 0, a0, b0);
                read_half(0, 0, 1, a1, b1);
                // steady state without branches (so that the waits count exactly the loads in flight): the last tile's
                // MFMAs are peeled off, the tile fetch is clamped (it re-reads tile K - 1 at the end, into a free buffer)
                for (int J = 0; J + 1 < K; ++J) {
                    load_tile(min(J + 2, K - 1));
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[m].x, C0, 0, 0, 0);
                        C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[m].y, C1, 0, 0, 0);
                    }
                    read_half(J + 1, (J + 1) & 1, 0, a0, b0);
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[m].x, C0, 0, 0, 0);
                        C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[m].y, C1, 0, 0, 0);
                    }
                    read_half(J + 1, (J + 1) & 1, 1, a1, b1);
                    store_tile(J & 1);   // buffer J & 1 held tile J: its operands were read during iteration J - 1
                    // the order of the above, spelled out for the scheduler (it otherwise issues all 32 MFMAs first and
                    // fetches the next operands when they are already needed)
                    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);    // tile J + 2: 4 VMEM reads
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // 16 MFMA (first half of tile J)
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // A of tile J + 1, first half: 2 DS reads
                    __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);    // B of tile J + 1, first half: 8 VMEM reads
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // 16 MFMA (second half)
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 8, 0);    // tile J + 2 into LDS: 8 DS writes
                    lds_order();
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[m].x, C0, 0, 0, 0);
                    C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[m].y, C1, 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[m].x, C0, 0, 0, 0);
                    C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[m].y, C1, 0, 0, 0);
                }
            }
        }
        // diagonal block: back to lane = leaf (the same swap), rows in order
        float cc[32], vb[32];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(C0[4 * g + j]), __float_as_uint(C1[4 * g + j]), false, false);
                cc[8 * g + j] = __uint_as_float(sw[0]);
                cc[8 * g + 4 + j] = __uint_as_float(sw[1]);
            }
        }
        if (R0 + 32 > N) {   // the padded rows of the last block solve to 0 (their A rows were not zero)
#pragma unroll
            for (int r = 1; r < 32; ++r) cc[r] = R0 + r < N ? cc[r] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int row = R0 + r;
            // row r of the diagonal tile from LDS, four entries per (broadcast) read; a padded row is a row of the identity
            float lrow[32];
#pragma unroll
            for (int w4 = 0; w4 <= r / 4; ++w4) {
                const float4 q = *reinterpret_cast<const float4 *>(&s_d[r][4 * w4]);
                lrow[4 * w4] = q.x; lrow[4 * w4 + 1] = q.y; lrow[4 * w4 + 2] = q.z; lrow[4 * w4 + 3] = q.w;
            }
            float acc = cc[r];
#pragma unroll
            for (int w = 0; w < (LA3DM_GP_EXP_V == 3 ? 0 : r); ++w) acc = __builtin_fmaf(-lrow[w], vb[w], acc);
            const float v = acc / lrow[r];
            vb[r] = v;
            ss = __builtin_fmaf(v, v, ss);
            vg[(size_t)row * kWave + 2 * c + h] = -v;   // B operand layout: (leaf c, leaf 32 + c) adjacent, negated (C -= L V as C += L (-V): the same FMA bits); vmax is whole blocks
        }
        // V[K] is read back as MFMA B operands by the other lanes of this wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    mj_out = mj;
    ss_out = ss;
}

// GPRegressor::predict + BCM fusion: one wave64 per leaf tile, lane = leaf (test point).
// v = L^-1 Ks per lane: v_k = (Ks_k - sum_{i<k} L_ki v_i) / L_kk.  Four rows at a time: their FMA chains
// over the already solved i are independent (latency hidden, one LDS read of v_i feeds four rows), each
// chain still runs over i ascending — the oracle's order.  Rows of L are loaded with coalesced vector
// loads (lane = column) and broadcast with v_readlane; v lives in LDS [row][lane].  Blocks with more than 64
// rows go through gp_solve_mfma (above).
constexpr int kGpLdsRows = 64;  // rows of v the LDS path can hold (one wave of columns of L)
#ifndef LA3DM_GP_MFMA_MIN_N
#define LA3DM_GP_MFMA_MIN_N 65
#endif
constexpr int kGpMfmaMinN = LA3DM_GP_MFMA_MIN_N;  // blocks with at least this many points are solved on the matrix cores
static_assert(kGpMfmaMinN <= kGpLdsRows + 1, "the small path keeps v in s_v[64][64] (lane = column of L): every block with "
                                             "more than kGpLdsRows points must take the matrix-core path");

// Two launches share this body: the tiles whose seven neighbours all hold fewer than kGpMfmaMinN points (kMixed = false: no
// matrix-core code in the kernel, so no 256-VGPR budget and no spills in the four-row loop — with both paths in one kernel
// the small path of configs[2] ran 7 % slower after the large path had grown) and the tiles with at least one large
// neighbour (kMixed = true).  Every tile is taken by exactly one of them.
// The small launch comes in classes by the tile's largest neighbour block (n_lo < max N <= n_hi, LDS sized for n_hi rows of
// v): the kernel is VALU bound on Ks but a wave issues only a third of its life, and with v sized for 64 rows (16 KB) a SIMD
// held 2.1 waves — 66 % VALU busy (profiles/r03/gp_d3_pmc.txt).
template <bool kMixed>
__device__ __forceinline__ void gp_predict_fuse_body(const GpArgs &a, float *s_vraw, int n_lo = -1, int n_hi = 0x7fffffff) {
    float (*s_v)[kWave] = reinterpret_cast<float (*)[kWave]>(s_vraw);
    __shared__ float4 s_pan[kWave];   // the small path's 4-row panel of L
    const int lane = threadIdx.x;
    // Workgroup w runs on XCD w % 8.  All tiles of a test block go to one XCD (they read the same seven factors: through one
    // L2 instead of eight), the blocks are dealt to the XCDs round-robin (they are sorted heaviest first: contiguous ranges
    // per XCD measured 3x slower).  One tile per block (depth 3): the identity.
    uint32_t task = blockIdx.x;
    {
        const uint32_t nb8 = a.n_test_blk & ~7u;
        if (task < (nb8 << a.tpb_shift)) {
            const uint32_t xcd = task & 7u, slot = task >> 3;
            task = ((((slot >> a.tpb_shift) << 3) | xcd) << a.tpb_shift) | (slot & ((1u << a.tpb_shift) - 1u));
        }
    }
    if (task >= a.n_tasks) return;
    const uint32_t blk = task >> a.tpb_shift;
    {
        int mx = 0;
        for (int nb = 0; nb < 7; ++nb) mx = max(mx, (int)a.nbr_range[7 * blk + nb].y);
        const bool large = mx >= kGpMfmaMinN;
        if (large != kMixed) return;   // (uniform)
        if (!kMixed && (mx <= n_lo || mx > n_hi)) return;
    }
    const uint32_t tile = task & ((1u << a.tpb_shift) - 1u);
    const uint32_t l0 = a.leaf_off[blk] + tile * kWave;
    const uint32_t l1 = a.leaf_off[blk + 1];
    if (l0 >= l1) return;
    const uint32_t nl = min(l1 - l0, (uint32_t)kWave);
    const bool active = (uint32_t)lane < nl;
    const uint32_t li = l0 + (active ? lane : 0);
    const uint32_t key = a.leaf_key[li];
    const float4 off4 = a.lut[lut_layer_base(key >> 16) + (key & 0xFFFFu)];
    // 1.73205/ell * xs (gpregressor.h:115): position first (LUT + centre), then the scale
    const float tx = a.scale * (off4.x + a.blk_center[3 * blk + 0]);
    const float ty = a.scale * (off4.y + a.blk_center[3 * blk + 1]);
    const float tz = a.scale * (off4.z + a.blk_center[3 * blk + 2]);
    float m_ivar = a.m_ivar[li], ivar = a.ivar[li];
    bool updated = false, unknown = true;
    float *vg = a.vscratch ? a.vscratch + (size_t)task * a.vmax * kWave : nullptr;

    for (int nb = 0; nb < 7; ++nb) {
        const uint2 r = a.nbr_range[7 * blk + nb];
        const int N = (int)r.y;
        if (N == 0) continue;
        const int tb = a.nbr[7 * blk + nb];
        const float *L = a.Lmat + a.l_off[tb];
        const float4 *x = a.pts + r.x;
        const float *al = a.alpha_k + r.x;
        float mj = 0.0f, ss = 0.0f;
        if (!kMixed || N < kGpMfmaMinN) {
            // fast path: a row of L fits one register (lane = column)
            for (int k0 = 0; k0 < N; k0 += 4) {
                float Lr[4], acc[4], ks[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = min(k0 + u, N - 1);
                    Lr[u] = lane <= k ? L[(size_t)k * N + lane] : 0.0f;
                    const float4 xk = x[k];
                    ks[u] = matern3_fast(xk.x, xk.y, xk.z, tx, ty, tz, a.sf2);  // Ks(k, j) = k(x_k, xs_j)
                    acc[u] = ks[u];
                }
                // the four rows' entries of a column, side by side in LDS (lane = column wrote them): one broadcast
                // 16-byte read per solved row instead of four v_readlane — the substitution is half of this kernel's VALU
                // at N = 50, and a v_readlane costs an issue slot like the FMA it feeds
                lds_order();
                s_pan[lane] = make_float4(Lr[0], Lr[1], Lr[2], Lr[3]);
                lds_order();
                __builtin_amdgcn_wave_barrier();
#pragma unroll 8
                for (int i = 0; i < k0; ++i) {  // four independent chains over the solved rows (k0 is a multiple of 4)
                    const float vi = s_v[i][lane];
                    const float4 l4 = s_pan[i];
                    acc[0] = __builtin_fmaf(-l4.x, vi, acc[0]);
                    acc[1] = __builtin_fmaf(-l4.y, vi, acc[1]);
                    acc[2] = __builtin_fmaf(-l4.z, vi, acc[2]);
                    acc[3] = __builtin_fmaf(-l4.w, vi, acc[3]);
                }
                float vloc[4];   // the group's own solved rows stay in registers (read back from LDS they were a write -> read round
                                 // trip on the serial path of every row)
#pragma unroll
                for (int u = 0; u < 4; ++u) {  // the 4x4 triangle
                    const int k = k0 + u;
                    if (k < N) {
#pragma unroll
                        for (int w = 0; w < u; ++w)
                            acc[u] = __builtin_fmaf(-__int_as_float(__builtin_amdgcn_readlane(__float_as_int(Lr[u]), k0 + w)),
                                                    vloc[w], acc[u]);
                        const float vk = acc[u] / __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Lr[u]), k));
                        vloc[u] = vk;
                        s_v[k][lane] = vk;
                        mj = __builtin_fmaf(ks[u], al[k], mj);
                        ss = __builtin_fmaf(vk, vk, ss);
                    }
                }
            }
        } else if constexpr (kMixed) {
            gp_solve_mfma(a, L, x, al, N, tx, ty, tz, vg, s_vraw, lane, mj, ss);   // (its tile staging reuses the LDS of the small path)
        }
        const float var = a.sf2 - ss;
        gp_node_accumulate_dev(a, m_ivar, ivar, unknown, mj, var);
        updated = true;
    }
    if (active) {
        if (updated) {
            a.m_ivar[li] = m_ivar;
            a.ivar[li] = ivar;
            a.state[li] = (uint8_t)(gp_node_state_dev(a, m_ivar, unknown) | 0x80u);
        } else {
            a.state[li] = 0;
        }
    }
}

__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void gp_predict_fuse_kernel(GpArgs a) {
    extern __shared__ float s_vraw[];  // [min(max N, kGpLdsRows)][64] or the tile staging: sized per launch
    gp_predict_fuse_body<true>(a, s_vraw);
}
__global__ __launch_bounds__(kWave) void gp_predict_fuse_small_kernel(GpArgs a, int n_lo, int n_hi) {
    extern __shared__ float s_vraw[];  // [n_hi][64]
    gp_predict_fuse_body<false>(a, s_vraw, n_lo, n_hi);
}

// Test hook for the property gp_solve_mfma / gp_train_kernel rely on: D = A B through v_mfma_f32_32x32x2_f32
// (k ascending, two per instruction) against per-element fmaf chains over k ascending.  One wave; counts the
// outputs whose bits differ.  A is 32 x K, B is K x 32 (row-major), K even.
__global__ __launch_bounds__(kWave) void gp_diag_mfma_chain(const float *__restrict__ A, const float *__restrict__ B, int K,
                                                           unsigned int *mismatches) {
    const int lane = threadIdx.x;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(lane & 31) * K + k + (lane >> 5)], B[(k + (lane >> 5)) * 32 + (lane & 31)],
                                                   acc, 0, 0, 0);
    unsigned int bad = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), col = lane & 31;
        float c = 0.0f;
        for (int k = 0; k < K; ++k) c = __builtin_fmaf(A[row * K + k], B[k * 32 + col], c);
        bad += __float_as_uint(c) != __float_as_uint(acc[r]);
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace la3dm_dev
