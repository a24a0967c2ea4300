// bgk_predict_p.h — bgk_predict_fuse_p (round 5; option "bgk_p" 1, NOT the default): the table kernel of bgk_kernels.h
// (bgk_predict_fuse_t) with a one-read prologue and the sin / cos table in LDS.  Included at the end of bgk_kernels.h (same
// namespace, same macros).
//
// Same pairs, same fp32 kernel values, same double accumulators as bgk_predict_fuse_t / _r (bit-identical alpha, beta, state:
// tests/test_bgk_sum_gpu.py, tools/check/p_quick.py).  Reference: include/bgkoctomap/bgkinference.h:73-126 (predict, dist,
// covSparse), src/bgkoctomap/bgkoctomap.cpp:314-335 (7-neighbour update loop), src/bgkoctomap/bgkoctree_node.cpp:31-44.
//
// What changes against bgk_predict_fuse_t:
//   * one 128-byte record per tile (bgk_prepare: bgk_write_tile_rec) replaces the leaf range + descriptor + centre reads, the
//     LUT gather, nine divisions by ell and twelve v_readlane per tile: the prologue is ONE scalar read and the point gathers
//     (prologue + A stage + epilogue alone: 23.9 -> 21.2 us at configs[1]);
//   * the sin / cos table in LDS (64 entries x 16 B; q <= 63 for every d2 below the hit threshold) instead of a 16-byte
//     gather from the vector L1 per batch — an LDS address cannot fault, and the C phase no longer touches the
//     vector-memory counter.  The 1 KB come out of the ring: 192 entries instead of 320, B pushes two candidates between
//     overflow checks instead of four (63 + 128 < 192).  configs[4]'s 1 M-ray scan (C is 80 % of a tile there): 0.771 ->
//     0.744 ms; configs[1]: unchanged (67.0 - 68.4 against 67.6 - 68.2 us).
// The records cost bgk_prepare 2.5 us per step at configs[1] (a second nbr -> train_off chain per block, 5.3 MB written), more
// than the kernel returns there: the step is 82.6 us against 80.0 us with bgk_predict_fuse_t, which therefore stays the default.
// (The same kernel with bgk_predict_fuse_t's prologue instead of the record — LDS table only — measured 70.1 us: dropped.)
// LDS per wave: ring 1 536 B + tables 1 536 B + sin / cos 1 024 B + 2 x 64 double accumulators 1 024 B = 5 120 B (8 waves / SIMD).
//
// Measured in round 5 and NOT kept (VERDICT r04 #1 asked for persistent waves with the next tile prefetched; DESIGN.md 3.2
// has the numbers): (a) persistent waves drawing tiles from one atomic ticket counter per XCD — ~29 ns per atomic on one
// address, 3 100 atomics per counter: 90 us for the launch with B and C switched off; (b) K adjacent tiles of the
// heaviest-first list per workgroup with the next tile's record, points and coordinates in flight under the current one —
// 72.5 (K = 2) / 78 (3) / 85 (4) / 125 us (8) against 67: the round granularity of 41.7 k tiles on 8 192 wave slots costs
// more than the prefetch returns, and the pipeline's scalar state does not fit the 72 SGPRs a wave has at 8 waves per SIMD
// (800 / 8, less the trap handler's 16, in granules of 16, less vcc / flat_scratch / xnack): the compiler spills to VGPR
// lanes, v_writelane / v_readlane are VALU instructions, and VALU issue is what bounds the kernel.
#pragma once

namespace la3dm_dev {

constexpr int kRingP = 192;
struct __attribute__((aligned(16))) WaveLdsP {
    uint2 ring[kRingP];              // {+-d2, LDS address of acc0[leaf]}
    float tab[kTabSlots / 16][12][16];
    la3dm_v2d sc[64];                // {sin, cos}(q h), q = 0 .. 63 (kMode bit 1)
    double acc0[kWave];              // at byte 4 096: bit 9 of its addresses is clear (the label bit is OR-ed in)
    double acc1[kWave];
};
static_assert(offsetof(WaveLdsP, acc0) == 4096 && offsetof(WaveLdsP, acc1) == 4608 && offsetof(WaveLdsP, tab) % 256 == 0 &&
                  offsetof(WaveLdsP, sc) % 1024 == 0 && sizeof(WaveLdsP) == 5120,
              "bgk_predict_fuse_p: LDS layout");

// B, four candidates read and compared at once (as LA3DM_TP_LOAD), pushed two at a time: the first asm block pushes
// candidates 0 and 1 and hands the d2 of 2 and 3 (v54, v55) and their hit masks to the second one, so that the ring's
// overflow check — and a C round — can sit between them: the ring needs 63 + 2 x 64 entries of head room, not 63 + 4 x 64.
#define LA3DM_PP_TRIP_A(OFF)                                                                        \
    asm volatile(LA3DM_TP_LOAD(OFF)                                                                 \
                 LA3DM_TP_PUSH("v52", "%[m0]") LA3DM_TP_PUSH("v53", "%[m1]")                        \
                 "s_mov_b64 exec, -1\n"                                                             \
                 : [tail] "+s"(tailb), [tr] "=&v"(tr), [st] "=&s"(st), [m0] "=&s"(hm0), [m1] "=&s"(hm1), [m2] "=&s"(hm2), [m3] "=&s"(hm3), \
                   "={v54}"(dC), "={v55}"(dD)                                                       \
                 : [ax] "v"(aX), [ay] "v"(aY), [az] "v"(aZ), [T] "s"(hit_t), [w] "v"(w0)            \
                 : "scc", "memory", "v52", "v53", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63")
#define LA3DM_PP_TRIP_B()                                                                           \
    asm volatile(LA3DM_TP_PUSH("%[dc]", "%[m2]") LA3DM_TP_PUSH("%[dd]", "%[m3]")                    \
                 "s_mov_b64 exec, -1\n"                                                             \
                 : [tail] "+s"(tailb), [tr] "=&v"(tr), [st] "=&s"(st)                               \
                 : [dc] "v"(dC), [dd] "v"(dD), [m2] "s"(hm2), [m3] "s"(hm3), [w] "v"(w0)            \
                 : "scc", "memory")

template <int kTrig, bool kGeneral>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(8, 8))) void bgk_predict_fuse_p(BgkArgs a) {
    constexpr bool kLdsTab = true;
    __shared__ __attribute__((aligned(16))) unsigned char s_lds[5120];
    WaveLdsP &L = *reinterpret_cast<WaveLdsP *>(s_lds);
    const uint32_t lane = threadIdx.x;
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    // (constant address space: the records were written by the launch before this one; a scalar read whatever asm block
    // or store sits between it and its use)
    typedef const __attribute__((address_space(4))) u32x16 *RecPtr;
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) u32x8 *Rec8Ptr;
    typedef const __attribute__((address_space(4))) u32x4 *Rec4Ptr;
    typedef __attribute__((address_space(3))) la3dm_v2d lds_v2d;
    auto rec_ptr = [&](uint32_t task) { return (RecPtr)(a.tile_rec + 32 * (size_t)task); };
    const uint32_t n_fine = 1u << (3u * (a.depth - 1u));

    // ---- per-wave constants ----
    // every label of this scan is 0 or 1 (the caller's LA3DM_SCAN_LABELS_01, cross-checked against what bgk_prepare saw)
    const bool binary = *(const __attribute__((address_space(4))) uint32_t *)a.label_seq != a.seq;
    if (!binary && !kGeneral) __builtin_trap();  // the caller's promise is broken and this instance has no general path: fail loudly
    // lane l holds leaf index c = 63 - l of the cube; c = (i1 j1 k1 i0 j0 k0): child number i*4 + j*2 + k at the parent level
    // (bits 5-3) and at the leaf level (bits 2-0), bgkblock.cpp:14-27; axis value r = 2 * (high bit) + (low bit)
    const uint32_t c6 = lane ^ 63u;
    const uint32_t ix = ((c6 >> 4) & 2u) | ((c6 >> 2) & 1u), iy = ((c6 >> 3) & 2u) | ((c6 >> 1) & 1u), iz = ((c6 >> 2) & 2u) | (c6 & 1u);
    const uint32_t tab_base = (uint32_t)(uintptr_t)&L.tab[0][0][0];
    constexpr uint32_t kRowB = 4u * 16u, kHalfB = 12u * kRowB;
    static_assert(kTabSlots == 32, "the B loop below walks two halves of 16 slots");
    const uint32_t ax0 = tab_base + ix * kRowB, ay0 = tab_base + (4u + iy) * kRowB, az0 = tab_base + (8u + iz) * kRowB;
    const uint32_t ring_base = (uint32_t)(uintptr_t)&L.ring[0];
    const uint32_t w0 = (uint32_t)(uintptr_t)&L.acc0[0] + 8u * lane;
    if (w0 & 0x200u) __builtin_trap();   // (s_lds is the kernel's only static LDS object: it starts at LDS address 0)
    const float hit_t = __uint_as_float(kHitTBits);
    const uint32_t tail_cap = ring_base + 8u * (uint32_t)(kRingP - 2 * kWave);
    // LDS address of the sin / cos entry of q from the bits of u = 0x4B400000 + q: (u << 4) + (table - 0xB4000000), one
    // v_lshl_add (q <= 63 for a finite t below 2 pi sqrt(T); an LDS address cannot fault)
    const uint32_t sc_bias = (uint32_t)(uintptr_t)&L.sc[0] - 0xB4000000u;
    auto load_sc = [&]() {
        if (kLdsTab) L.sc[lane] = *reinterpret_cast<const la3dm_v2d *>(&kSinCosTab[lane][0]);
    };
    load_sc();

    // the points of flat indices cb + lane of a tile's 7 neighbour blocks (record words 0-13); a lane past the end reads the
    // range's last point (the caller masks it)
    auto gather = [&](const u32x16 &r, uint32_t cb) {
        const uint32_t M = r[7];
        const uint32_t f = min(cb + lane, M - 1u);
        uint32_t adjv[7];
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            adjv[b] = r[b];
            asm volatile("" : "+v"(adjv[b]));   // (a v_cndmask reads one scalar operand, and its mask is one)
        }
        uint32_t ad;
        unsigned long long m1, m2, m3, m4, m5, m6;
        asm("v_cmp_le_u32 %[m1], %[e0], %[f]\n"
            "v_cmp_le_u32 %[m2], %[e1], %[f]\n"
            "v_cmp_le_u32 %[m3], %[e2], %[f]\n"
            "v_cmp_le_u32 %[m4], %[e3], %[f]\n"
            "v_cmp_le_u32 %[m5], %[e4], %[f]\n"
            "v_cmp_le_u32 %[m6], %[e5], %[f]\n"
            "v_cndmask_b32 %[ad], %[a0], %[a1], %[m1]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a2], %[m2]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a3], %[m3]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a4], %[m4]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a5], %[m5]\n"
            "v_cndmask_b32 %[ad], %[ad], %[a6], %[m6]\n"
            : [ad] "=&v"(ad), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5), [m6] "=&s"(m6)
            : [f] "v"(f), [e0] "s"(r[8]), [e1] "s"(r[9]), [e2] "s"(r[10]), [e3] "s"(r[11]), [e4] "s"(r[12]),
              [e5] "s"(r[13]), [a0] "v"(adjv[0]), [a1] "v"(adjv[1]), [a2] "v"(adjv[2]), [a3] "v"(adjv[3]), [a4] "v"(adjv[4]),
              [a5] "v"(adjv[5]), [a6] "v"(adjv[6]));
        return a.pts[f + ad];
    };

    // C: lane evaluates ring entry i and adds k to the leaf's accumulator 0 or 1 (the sign of d2 is the label)
    auto c_eval = [&](uint32_t i) {
        const uint2 e = L.ring[i];
        const float r = sqrt_cr(__builtin_fabsf(__uint_as_float(e.x)));
        const float t = (r * 2.0f) * 3.1415926f;
        float s, c;
        if (kTrig == 0 && kLdsTab) {
            sincos_cr_core(t, s, c, [&](uint32_t ub, double &sa, double &ca) {
                const la3dm_v2d en = *(const lds_v2d *)(uintptr_t)((ub << 4) + sc_bias);
                sa = en.x;
                ca = en.y;
            });
        } else if (kTrig == 0) {
            sincos_cr_finite(t, s, c);
        } else if (kTrig == 1) {
            sincos_0_2pi(t, s, c);
        } else if (kTrig == 3) {
            sincos_eigen337(t, s, c);
        } else {
            s = sinf(t);
            c = cosf(t);
        }
        const double kd = (double)cov_sparse_formula<true>(r, s, c, a.sf2);
        const uint32_t ad = e.y | ((e.x >> 22) & 0x200u);  // label 1 (negative d2): acc1[leaf], 512 bytes up
        asm volatile("ds_add_f64 %0, %1\n" : : "v"(ad), "v"(kd) : "memory");
    };
    uint32_t tailb = ring_base;  // LDS byte address of the ring's first free entry
    // C round: the full 64-entry batches, taken from the ring's END (entries [rem, tail)): the remainder stays at the front
    auto c_flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t tail = (tailb - ring_base) >> 3;
        const uint32_t rem = tail & 63u;
        if (!(a.flags & 0x100u))  // 0x100: profiling ablation
            for (uint32_t p = rem; p < tail; p += kWave) c_eval(p + lane);
        tailb = ring_base + 8u * rem;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- the tile ----
    const uint32_t task = bgk_task_of_workgroup(a);
    if (task >= a.n_tasks) return;
    const u32x16 rA = rec_ptr(task)[0];
    const u32x8 cXY = *(Rec8Ptr)(a.tile_rec + 32 * (size_t)task + 16);
    const u32x4 cZ = *(Rec4Ptr)(a.tile_rec + 32 * (size_t)task + 24);
    const uint32_t M = rA[7], li = rA[14] + lane;
    if (rA[15] != n_fine || !binary) {
        // a pruned block (or labels other than 0 / 1): the general path, any leaf layout
        if (kGeneral) bgk_tile_r<kTrig>(a, *reinterpret_cast<WaveLdsR *>(s_lds), task);
        return;  // (kGeneral false: the caller vouched for full blocks, LA3DM_SCAN_FULL_BLOCKS, and the host checked the leaf count)
    }
    if (M == 0u) {  // no training point in the 7 blocks: nothing reaches the tile
        if (!(a.flags & 1u)) a.state[li] = 0;
        else {  // insert_training_data: update() runs with (0, 0)
            const float A = a.alpha[li], B = a.beta[li];
            a.state[li] = (uint8_t)(classify(A, B, a) | 0x80u);
        }
        return;
    }
    float4 pc = gather(rA, 0), pn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (M > (uint32_t)kWave) pn = gather(rA, kWave);
    const float A0 = a.alpha[li], B0 = a.beta[li];  // needed by the epilogue only: loaded here, a round trip off the tile's tail
    load_sc();
    L.acc0[lane] = 0.0;
    L.acc1[lane] = 0.0;
    const la3dm_v2f X01 = {__uint_as_float(cXY[0]), __uint_as_float(cXY[1])}, X23 = {__uint_as_float(cXY[2]), __uint_as_float(cXY[3])};
    const la3dm_v2f Y01 = {__uint_as_float(cXY[4]), __uint_as_float(cXY[5])}, Y23 = {__uint_as_float(cXY[6]), __uint_as_float(cXY[7])};
    const la3dm_v2f Z01 = {__uint_as_float(cZ[0]), __uint_as_float(cZ[1])}, Z23 = {__uint_as_float(cZ[2]), __uint_as_float(cZ[3])};

    // A + B for one chunk of 64 training points (lane = point)
    auto chunk = [&](const float4 &p, const uint32_t cb) {
        const la3dm_v2f bx = {p.x, p.x}, by = {p.y, p.y}, bz = {p.z, p.z};
        la3dm_v2f x01 = bx - X01, x23 = bx - X23, y01 = by - Y01, y23 = by - Y23, z01 = bz - Z01, z23 = bz - Z23;
        x01 *= x01, x23 *= x23, y01 *= y01, y23 *= y23, z01 *= z01, z23 *= z23;
        const float mx = fminf(fminf(x01.x, x01.y), fminf(x23.x, x23.y));
        const float my = fminf(fminf(y01.x, y01.y), fminf(y23.x, y23.y));
        const float mz = fminf(fminf(z01.x, z01.y), fminf(z23.x, z23.y));
        // min over the 64 leaves of d2 (+ and * are monotone): staged iff some leaf hits (lanes past the end hold a
        // copy of the last point: masked)
        const bool keep = mx + (my + mz) < hit_t && lane < M - cb;
        const unsigned long long m = __ballot(keep);
        if (m == 0ull) return;
        const float sg = 1.0f - (p.w + p.w);  // label 0 -> +1, label 1 -> -1 (exact)
        const la3dm_v2f s2 = {sg, sg};
        x01 *= s2, x23 *= s2, y01 *= s2, y23 *= s2, z01 *= s2, z23 *= s2;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        const uint32_t n = (uint32_t)__popcll(m);
        for (uint32_t base = 0; base < n; base += kTabSlots) {
            const uint32_t nr = min(n - base, (uint32_t)kTabSlots), ngroup = (nr + 3u) >> 2;
            // pad the last column group with entries no leaf can reach (an X row suffices: the sum stays huge or NaN)
            if (lane < 4u && nr + lane < 4u * ngroup) {
                float big;
                asm volatile("v_mov_b32 %0, 0x5e268890" : "=v"(big));
#pragma unroll
                for (int r = 0; r < 4; ++r) L.tab[(nr + lane) >> 4][r][(nr + lane) & 15u] = big;
            }
            const uint32_t slot = rank - base;
            if (keep && slot < (uint32_t)kTabSlots) {
                float(&T)[12][16] = L.tab[slot >> 4];
                const uint32_t c = slot & 15u;
                T[0][c] = x01.x, T[1][c] = x01.y, T[2][c] = x23.x, T[3][c] = x23.y;
                T[4][c] = y01.x, T[5][c] = y01.y, T[6][c] = y23.x, T[7][c] = y23.y;
                T[8][c] = z01.x, T[9][c] = z01.y, T[10][c] = z23.x, T[11][c] = z23.y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (a.flags & 0x200u) continue;  // 0x200: profiling ablation
            uint32_t aX = ax0, aY = ay0, aZ = az0;
            for (uint32_t g = 0; g < ngroup; g += 2u) {  // two column groups per trip: the three row addresses move once
                uint32_t st;
                float tr, dC, dD;
                unsigned long long hm0, hm1, hm2, hm3;
                LA3DM_PP_TRIP_A("0");
                if (tailb > tail_cap) c_flush();
                LA3DM_PP_TRIP_B();
                if (tailb > tail_cap) c_flush();
                if (g + 1u < ngroup) {
                    LA3DM_PP_TRIP_A("16");
                    if (tailb > tail_cap) c_flush();
                    LA3DM_PP_TRIP_B();
                    if (tailb > tail_cap) c_flush();
                }
                const uint32_t step = g == 2u ? kHalfB - 32u : 32u;   // column groups 0-3 sit in the first half, 4-7 in the second
                aX += step, aY += step, aZ += step;
            }
            // (the next sub-round overwrites the table: LDS operations of a wave complete in order)
        }
    };

    for (uint32_t cb = 0;;) {
        chunk(pc, cb);
        cb += kWave;
        if (cb >= M) break;
        pc = pn;
        if (cb + kWave < M) {   // (a fourth, fifth .. chunk: the record again — its words would otherwise sit in SGPRs for the whole tile)
            const u32x16 rr = rec_ptr(task)[0];
            pn = gather(rr, cb + kWave);
        }
    }

    // the tile's last, partly filled batch(es)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const uint32_t tail = (tailb - ring_base) >> 3;
        if (!(a.flags & 0x100u))
            for (uint32_t p = 0; p < tail; p += kWave)
                if (p + lane < tail) c_eval(p + lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    {
        const double s0 = L.acc0[lane], s1 = L.acc1[lane];
        const double K = s0 + s1, Y = s1;
        uint32_t lw = li;
        asm volatile("" : "+v"(lw));
        if (K > 0.0 || (a.flags & 1u) != 0u) {  // flag 1: insert_training_data, update() runs unconditionally
            const float A = (float)((double)A0 + Y);
            const float B = (float)((double)B0 + (K - Y));
            a.alpha[lw] = A;
            a.beta[lw] = B;
            a.state[lw] = (uint8_t)(classify_fast(A, B, a) | 0x80u);
        } else {
            a.state[lw] = 0;
        }
    }
}

}  // namespace la3dm_dev
