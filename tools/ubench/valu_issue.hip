// valu_issue.hip — measured issue rate of wave64 instructions on gfx950 (MI355X).
// the BGK kernel's "VALU-issue roofline" needs a measured peak.
//
// Every test is one kernel: W waves per SIMD (256 CUs x 4 SIMDs x W waves resident), each wave runs
// kIters trips of a 32-instruction straight-line body of INDEPENDENT instructions of one kind (8 or 16
// separate dependency chains, so latency never binds at >= 2 waves per SIMD).  Reported:
//   cyc = issue cycles per wave-instruction per SIMD = t * f_clk * 1024 / (waves * instructions per wave)
// with f_clk measured in the same launch from s_memtime (shader clock ticks) against the HIP-event wall time.
//
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_issue.hip -o tools/_out/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int kIters = 2000;

// 32 instructions per trip; %0..%15 are 16 live VGPRs (8 pairs for 64-bit forms)
#define R4(a) a a a a
#define BODY_OPEN(NAME)                                                                               \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk, float seed) {    \
        float v0 = seed + threadIdx.x, v1 = v0 * 1.1f, v2 = v0 * 1.2f, v3 = v0 * 1.3f, v4 = v0 * 1.4f, v5 = v0 * 1.5f,   \
              v6 = v0 * 1.6f, v7 = v0 * 1.7f, v8 = v0 * 1.8f, v9 = v0 * 1.9f, v10 = v0 * 2.1f, v11 = v0 * 2.2f,         \
              v12 = v0 * 2.3f, v13 = v0 * 2.4f, v14 = v0 * 2.5f, v15 = v0 * 2.6f;                                       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int it = 0; it < kIters; ++it) {
#define BODY_CLOSE                                                                                    \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] =                                                  \
            v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + v8 + v9 + v10 + v11 + v12 + v13 + v14 + v15;      \
    }
#define REGS16                                                                                                          \
    "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9), "+v"(v10), "+v"(v11), \
        "+v"(v12), "+v"(v13), "+v"(v14), "+v"(v15)

// 16 independent two-operand-in-place instructions, twice
#define T16(OP)                                                                                          \
    OP " %0, %0, %1\n" OP " %2, %2, %3\n" OP " %4, %4, %5\n" OP " %6, %6, %7\n" OP " %8, %8, %9\n"      \
       OP " %10, %10, %11\n" OP " %12, %12, %13\n" OP " %14, %14, %15\n" OP " %1, %1, %0\n"             \
       OP " %3, %3, %2\n" OP " %5, %5, %4\n" OP " %7, %7, %6\n" OP " %9, %9, %8\n" OP " %11, %11, %10\n" \
       OP " %13, %13, %12\n" OP " %15, %15, %14\n"
#define T16_3(OP)                                                                                                      \
    OP " %0, %0, %1, %0\n" OP " %2, %2, %3, %2\n" OP " %4, %4, %5, %4\n" OP " %6, %6, %7, %6\n" OP " %8, %8, %9, %8\n" \
       OP " %10, %10, %11, %10\n" OP " %12, %12, %13, %12\n" OP " %14, %14, %15, %14\n" OP " %1, %1, %0, %1\n"         \
       OP " %3, %3, %2, %3\n" OP " %5, %5, %4, %5\n" OP " %7, %7, %6, %7\n" OP " %9, %9, %8, %9\n"                     \
       OP " %11, %11, %10, %11\n" OP " %13, %13, %12, %13\n" OP " %15, %15, %14, %15\n"

#define KERNEL_2OP(NAME, OP) \
    BODY_OPEN(NAME) asm volatile(T16(OP) T16(OP) : REGS16); BODY_CLOSE
#define KERNEL_3OP(NAME, OP) \
    BODY_OPEN(NAME) asm volatile(T16_3(OP) T16_3(OP) : REGS16); BODY_CLOSE

KERNEL_2OP(k_add_f32, "v_add_f32")
KERNEL_2OP(k_sub_f32, "v_sub_f32")
KERNEL_2OP(k_mul_f32, "v_mul_f32")
KERNEL_2OP(k_max_f32, "v_max_f32")
KERNEL_3OP(k_fma_f32, "v_fma_f32")
KERNEL_2OP(k_and_b32, "v_and_b32")
KERNEL_2OP(k_add_u32, "v_add_u32")
KERNEL_2OP(k_lshl_b32, "v_lshlrev_b32")
KERNEL_3OP(k_med3_f32, "v_med3_f32")
KERNEL_3OP(k_mad_u32_u24, "v_mad_u32_u24")
KERNEL_3OP(k_bfe_u32, "v_bfe_u32")
KERNEL_3OP(k_lshl_or, "v_lshl_or_b32")

// one-operand forms
#define U16(OP)                                                                                                    \
    OP " %0, %1\n" OP " %2, %3\n" OP " %4, %5\n" OP " %6, %7\n" OP " %8, %9\n" OP " %10, %11\n" OP " %12, %13\n"   \
       OP " %14, %15\n" OP " %1, %0\n" OP " %3, %2\n" OP " %5, %4\n" OP " %7, %6\n" OP " %9, %8\n" OP " %11, %10\n" \
       OP " %13, %12\n" OP " %15, %14\n"
#define KERNEL_1OP(NAME, OP) BODY_OPEN(NAME) asm volatile(U16(OP) U16(OP) : REGS16); BODY_CLOSE
KERNEL_1OP(k_mov_b32, "v_mov_b32")
KERNEL_1OP(k_sqrt_f32, "v_sqrt_f32")
KERNEL_1OP(k_rcp_f32, "v_rcp_f32")
KERNEL_1OP(k_rndne_f32, "v_rndne_f32")
KERNEL_1OP(k_cvt_i32_f32, "v_cvt_i32_f32")

KERNEL_2OP(k_or_b32, "v_or_b32")
KERNEL_2OP(k_xor_b32, "v_xor_b32")
KERNEL_2OP(k_sub_u32, "v_sub_u32")
KERNEL_2OP(k_lshr_b32, "v_lshrrev_b32")
KERNEL_2OP(k_ashr_i32, "v_ashrrev_i32")
KERNEL_2OP(k_min_f32, "v_min_f32")
KERNEL_2OP(k_fmac_f32, "v_fmac_f32")
KERNEL_2OP(k_mul_u32_u24, "v_mul_u32_u24")
KERNEL_2OP(k_mul_lo_u32, "v_mul_lo_u32")
KERNEL_2OP(k_max_u32, "v_max_u32")
KERNEL_2OP(k_min_u32, "v_min_u32")
KERNEL_2OP(k_ldexp_f32, "v_ldexp_f32")
KERNEL_2OP(k_add_f16, "v_add_f16")
KERNEL_2OP(k_mul_legacy, "v_mul_legacy_f32")
KERNEL_3OP(k_bfi_b32, "v_bfi_b32")
KERNEL_3OP(k_perm_b32, "v_perm_b32")
KERNEL_3OP(k_add3_u32, "v_add3_u32")
KERNEL_3OP(k_and_or, "v_and_or_b32")
KERNEL_3OP(k_max3_f32, "v_max3_f32")
KERNEL_3OP(k_alignbit, "v_alignbit_b32")
KERNEL_3OP(k_add_lshl, "v_add_lshl_u32")
KERNEL_3OP(k_lshl_add, "v_lshl_add_u32")
KERNEL_3OP(k_xad, "v_xad_u32")
KERNEL_1OP(k_cvt_f32_u32, "v_cvt_f32_u32")
KERNEL_1OP(k_cvt_f32_i32, "v_cvt_f32_i32")
KERNEL_1OP(k_fract_f32, "v_fract_f32")
KERNEL_1OP(k_exp_f32, "v_exp_f32")
KERNEL_1OP(k_sin_f32, "v_sin_f32")
KERNEL_1OP(k_rsq_f32, "v_rsq_f32")
KERNEL_1OP(k_not_b32, "v_not_b32")
KERNEL_1OP(k_bfrev, "v_bfrev_b32")
KERNEL_1OP(k_floor_f32, "v_floor_f32")
KERNEL_1OP(k_trunc_f32, "v_trunc_f32")
KERNEL_1OP(k_ffbh, "v_ffbh_u32")
KERNEL_1OP(k_bcnt_like, "v_cvt_u32_f32")

// v_cmp alone (vcc), v_cndmask alone (vcc fixed), carry ops, SGPR operands, literal constants
#define CV(a, b) "v_cmp_lt_f32 vcc, %" #a ", %" #b "\n"
BODY_OPEN(k_cmp_vcc)
asm volatile(R4(CV(0, 1) CV(2, 3) CV(4, 5) CV(6, 7) CV(8, 9) CV(10, 11) CV(12, 13) CV(14, 15)) : REGS16 : : "vcc");
BODY_CLOSE
#define CE(a, b) "v_cmp_eq_u32 vcc, %" #a ", %" #b "\n"
BODY_OPEN(k_cmp_eq_u32)
asm volatile(R4(CE(0, 1) CE(2, 3) CE(4, 5) CE(6, 7) CE(8, 9) CE(10, 11) CE(12, 13) CE(14, 15)) : REGS16 : : "vcc");
BODY_CLOSE
#define CN(a, b) "v_cndmask_b32 %" #a ", %" #a ", %" #b ", vcc\n"
BODY_OPEN(k_cndmask)
asm volatile("s_mov_b64 vcc, 0x5555\n" R4(CN(0, 1) CN(2, 3) CN(4, 5) CN(6, 7) CN(8, 9) CN(10, 11) CN(12, 13) CN(14, 15)) : REGS16 : : "vcc");
BODY_CLOSE
#define CNS(a, b) "v_cndmask_b32 %" #a ", %" #a ", %" #b ", s[20:21]\n"
BODY_OPEN(k_cndmask_sgpr)
asm volatile("s_mov_b64 s[20:21], 0x5555\n" R4(CNS(0, 1) CNS(2, 3) CNS(4, 5) CNS(6, 7) CNS(8, 9) CNS(10, 11) CNS(12, 13) CNS(14, 15)) : REGS16 : : "s20", "s21");
BODY_CLOSE
#define AC(a, b) "v_addc_co_u32 %" #a ", vcc, %" #a ", %" #b ", vcc\n"
BODY_OPEN(k_addc_co)
asm volatile("s_mov_b64 vcc, 0x5555\n" R4(AC(0, 1) AC(2, 3) AC(4, 5) AC(6, 7) AC(8, 9) AC(10, 11) AC(12, 13) AC(14, 15)) : REGS16 : : "vcc");
BODY_CLOSE
#define ACO(a, b) "v_add_co_u32 %" #a ", vcc, %" #a ", %" #b "\n"
BODY_OPEN(k_add_co)
asm volatile(R4(ACO(0, 1) ACO(2, 3) ACO(4, 5) ACO(6, 7) ACO(8, 9) ACO(10, 11) ACO(12, 13) ACO(14, 15)) : REGS16 : : "vcc");
BODY_CLOSE
#define SS(a) "v_sub_f32 %" #a ", s20, %" #a "\n"
BODY_OPEN(k_sub_f32_sgpr)
asm volatile("s_mov_b32 s20, 0x3f800000\n" R4(SS(0) SS(1) SS(2) SS(3) SS(4) SS(5) SS(6) SS(7)) : REGS16 : : "s20");
BODY_CLOSE
#define SL(a) "v_add_f32 %" #a ", 0x3fc00000, %" #a "\n"
BODY_OPEN(k_add_f32_literal)
asm volatile(R4(SL(0) SL(1) SL(2) SL(3) SL(4) SL(5) SL(6) SL(7)) : REGS16);
BODY_CLOSE
#define FK(a, b) "v_fma_f32 %" #a ", %" #a ", %" #b ", 1.0\n"
BODY_OPEN(k_fma_f32_inline)
asm volatile(R4(FK(0, 1) FK(2, 3) FK(4, 5) FK(6, 7) FK(8, 9) FK(10, 11) FK(12, 13) FK(14, 15)) : REGS16);
BODY_CLOSE
#define FS(a, b) "v_fma_f32 %" #a ", %" #a ", s20, %" #b "\n"
BODY_OPEN(k_fma_f32_sgpr)
asm volatile("s_mov_b32 s20, 0x3f800000\n" R4(FS(0, 1) FS(2, 3) FS(4, 5) FS(6, 7) FS(8, 9) FS(10, 11) FS(12, 13) FS(14, 15)) : REGS16 : : "s20");
BODY_CLOSE
#define FN(a, b) "v_fma_f32 %" #a ", -%" #a ", %" #b ", |%" #b "|\n"
BODY_OPEN(k_fma_f32_mods)
asm volatile(R4(FN(0, 1) FN(2, 3) FN(4, 5) FN(6, 7) FN(8, 9) FN(10, 11) FN(12, 13) FN(14, 15)) : REGS16);
BODY_CLOSE
#define AE(a, b) "v_add_f32_e64 %" #a ", %" #a ", -%" #b "\n"
BODY_OPEN(k_add_f32_e64)
asm volatile(R4(AE(0, 1) AE(2, 3) AE(4, 5) AE(6, 7) AE(8, 9) AE(10, 11) AE(12, 13) AE(14, 15)) : REGS16);
BODY_CLOSE
#define ME(a, b) "v_mul_f32_e64 %" #a ", %" #a ", %" #b " clamp\n"
BODY_OPEN(k_mul_f32_e64_clamp)
asm volatile(R4(ME(0, 1) ME(2, 3) ME(4, 5) ME(6, 7) ME(8, 9) ME(10, 11) ME(12, 13) ME(14, 15)) : REGS16);
BODY_CLOSE
// fast + slow interleaved: 16 v_add_f32 + 16 v_cmp (SGPR): 2 + 4 = 6 per pair if they serialise
#define FSL(a, b) "v_add_f32 %" #a ", %" #a ", %" #b "\n v_cmp_lt_f32 s[20:21], %" #b ", %" #a "\n"
BODY_OPEN(k_mix_fast_slow)
asm volatile(R4(R4(FSL(0, 1) FSL(2, 3))) : REGS16 : : "s20", "s21");
BODY_CLOSE
// slow VALU + SALU
#define SSL(a, b) "v_max_f32 %" #a ", %" #a ", %" #b "\n s_add_u32 s20, s20, 1\n"
BODY_OPEN(k_mix_slow_salu)
asm volatile(R4(R4(SSL(0, 1) SSL(2, 3))) : REGS16 : : "s20", "scc");
BODY_CLOSE
// 1 VALU : 3 SALU
BODY_OPEN(k_mix_valu_3salu)
asm volatile(R4(R4("v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n s_lshl_b32 s21, s20, 1\n s_and_b32 s22, s21, s20\n v_add_f32 %2, %2, %3\n s_add_u32 s20, s20, 1\n s_lshl_b32 s21, s20, 1\n s_and_b32 s22, s21, s20\n")) : REGS16 : : "s20", "s21", "s22", "scc");
BODY_CLOSE
// readfirstlane, permlane-free cross-lane: ds_bpermute, ds_swizzle
#define RF(a) "v_readfirstlane_b32 s20, %" #a "\n"
BODY_OPEN(k_readfirstlane)
asm volatile(R4(RF(0) RF(1) RF(2) RF(3) RF(4) RF(5) RF(6) RF(7)) : REGS16 : : "s20");
BODY_CLOSE
#define WL(a) "v_writelane_b32 %" #a ", s20, 3\n"
BODY_OPEN(k_writelane)
asm volatile("s_mov_b32 s20, 5\n" R4(WL(0) WL(1) WL(2) WL(3) WL(4) WL(5) WL(6) WL(7)) : REGS16 : : "s20");
BODY_CLOSE

// v_cmp -> vcc, v_cndmask consumes it (pairs): 16 cmp + 16 cndmask
#define CC(a, b) "v_cmp_lt_f32 vcc, %" #a ", %" #b "\n v_cndmask_b32 %" #a ", %" #a ", %" #b ", vcc\n"
BODY_OPEN(k_cmp_cndmask)
asm volatile(CC(0, 1) CC(2, 3) CC(4, 5) CC(6, 7) CC(8, 9) CC(10, 11) CC(12, 13) CC(14, 15) CC(1, 0) CC(3, 2) CC(5, 4) CC(7, 6)
                 CC(9, 8) CC(11, 10) CC(13, 12) CC(15, 14)
             : REGS16
             :
             : "vcc");
BODY_CLOSE

// v_cmp into an SGPR pair (what __ballot compiles to): 32 per trip
#define CS(a, b) "v_cmp_lt_f32 s[20:21], %" #a ", %" #b "\n"
BODY_OPEN(k_cmp_sgpr)
asm volatile(CS(0, 1) CS(2, 3) CS(4, 5) CS(6, 7) CS(8, 9) CS(10, 11) CS(12, 13) CS(14, 15) CS(1, 0) CS(3, 2) CS(5, 4) CS(7, 6)
                 CS(9, 8) CS(11, 10) CS(13, 12) CS(15, 14) CS(0, 1) CS(2, 3) CS(4, 5) CS(6, 7) CS(8, 9) CS(10, 11) CS(12, 13)
                     CS(14, 15) CS(1, 0) CS(3, 2) CS(5, 4) CS(7, 6) CS(9, 8) CS(11, 10) CS(13, 12) CS(15, 14)
             : REGS16
             :
             : "s20", "s21");
BODY_CLOSE

// mbcnt pair on a mask in SGPRs: 16 lo + 16 hi
#define MB(a) "v_mbcnt_lo_u32_b32 %" #a ", s20, 0\n v_mbcnt_hi_u32_b32 %" #a ", s21, %" #a "\n"
BODY_OPEN(k_mbcnt)
asm volatile("s_mov_b32 s20, 0x0f0f0f0f\n s_mov_b32 s21, 0x33333333\n" MB(0) MB(1) MB(2) MB(3) MB(4) MB(5) MB(6) MB(7) MB(8) MB(9)
                 MB(10) MB(11) MB(12) MB(13) MB(14) MB(15)
             : REGS16
             :
             : "s20", "s21");
BODY_CLOSE

// readlane / readfirstlane (VALU issue, SGPR result)
#define RL(a) "v_readlane_b32 s20, %" #a ", 5\n"
BODY_OPEN(k_readlane)
asm volatile(R4(RL(0) RL(1) RL(2) RL(3) RL(4) RL(5) RL(6) RL(7)) : REGS16 : : "s20");
BODY_CLOSE

// DPP move (row_shr:1)
#define DP(a, b) "v_mov_b32_dpp %" #a ", %" #b " row_shr:1 row_mask:0xf bank_mask:0xf\n"
BODY_OPEN(k_dpp_mov)
asm volatile(R4(DP(0, 1) DP(2, 3) DP(4, 5) DP(6, 7) DP(8, 9) DP(10, 11) DP(12, 13) DP(14, 15)) : REGS16);
BODY_CLOSE
#define DA(a, b) "v_add_f32_dpp %" #a ", %" #b ", %" #a " row_shr:1 row_mask:0xf bank_mask:0xf\n"
BODY_OPEN(k_dpp_add)
asm volatile(R4(DA(0, 1) DA(2, 3) DA(4, 5) DA(6, 7) DA(8, 9) DA(10, 11) DA(12, 13) DA(14, 15)) : REGS16);
BODY_CLOSE

// ---- 64-bit operands: 8 register pairs ----
#define BODY64_OPEN(NAME)                                                                             \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk, float seed) {    \
        double d0 = seed + threadIdx.x, d1 = d0 * 1.1, d2 = d0 * 1.2, d3 = d0 * 1.3, d4 = d0 * 1.4, d5 = d0 * 1.5,      \
               d6 = d0 * 1.6, d7 = d0 * 1.7;                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int it = 0; it < kIters; ++it) {
#define BODY64_CLOSE                                                                                  \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);  \
    }
#define REGS8 "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
#define P8_3(OP)                                                                                                       \
    OP " %0, %0, %1, %0\n" OP " %2, %2, %3, %2\n" OP " %4, %4, %5, %4\n" OP " %6, %6, %7, %6\n" OP " %1, %1, %0, %1\n" \
       OP " %3, %3, %2, %3\n" OP " %5, %5, %4, %5\n" OP " %7, %7, %6, %7\n"
#define P8_2(OP)                                                                                              \
    OP " %0, %0, %1\n" OP " %2, %2, %3\n" OP " %4, %4, %5\n" OP " %6, %6, %7\n" OP " %1, %1, %0\n" OP " %3, %3, %2\n" \
       OP " %5, %5, %4\n" OP " %7, %7, %6\n"
#define KERNEL64_3OP(NAME, OP) BODY64_OPEN(NAME) asm volatile(R4(P8_3(OP)) : REGS8); BODY64_CLOSE
#define KERNEL64_2OP(NAME, OP) BODY64_OPEN(NAME) asm volatile(R4(P8_2(OP)) : REGS8); BODY64_CLOSE
KERNEL64_3OP(k_fma_f64, "v_fma_f64")
KERNEL64_2OP(k_mul_f64, "v_mul_f64")
KERNEL64_2OP(k_add_f64, "v_add_f64")
KERNEL64_3OP(k_pk_fma_f32, "v_pk_fma_f32")
KERNEL64_2OP(k_pk_mul_f32, "v_pk_mul_f32")
KERNEL64_2OP(k_pk_add_f32, "v_pk_add_f32")
BODY64_OPEN(k_lshl_b64)
asm volatile(R4("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n"
                "v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5\n v_lshlrev_b64 %6, 1, %6\n v_lshlrev_b64 %7, 1, %7\n")
             : REGS8);
BODY64_CLOSE

// f32 <-> f64 conversions: 16 up + 16 down per trip
BODY64_OPEN(k_cvt_f64_f32)
float f0, f1, f2, f3;
asm volatile(R4("v_cvt_f32_f64 %8, %0\n v_cvt_f32_f64 %9, %1\n v_cvt_f32_f64 %10, %2\n v_cvt_f32_f64 %11, %3\n"
                "v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %10\n v_cvt_f64_f32 %7, %11\n")
             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "=&v"(f0), "=&v"(f1), "=&v"(f2),
               "=&v"(f3));
BODY64_CLOSE

// ---- mixes ----
// 16 VALU + 16 SALU: do the scalar instructions take VALU issue slots?
#define VS(a, b) "v_add_f32 %" #a ", %" #a ", %" #b "\n s_add_u32 s20, s20, 1\n"
BODY_OPEN(k_mix_valu_salu)
asm volatile(VS(0, 1) VS(2, 3) VS(4, 5) VS(6, 7) VS(8, 9) VS(10, 11) VS(12, 13) VS(14, 15) VS(1, 0) VS(3, 2) VS(5, 4) VS(7, 6)
                 VS(9, 8) VS(11, 10) VS(13, 12) VS(15, 14)
             : REGS16
             :
             : "s20", "scc");
BODY_CLOSE
// 16 f32 VALU + 16 f64 FMA interleaved
BODY64_OPEN(k_mix_f32_f64)
float a0 = seed, a1 = seed * 2, a2 = seed * 3, a3 = seed * 4;
asm volatile(R4("v_fma_f64 %0, %0, %1, %0\n v_add_f32 %8, %8, %9\n v_fma_f64 %2, %2, %3, %2\n v_add_f32 %10, %10, %11\n"
                "v_fma_f64 %4, %4, %5, %4\n v_add_f32 %9, %9, %8\n v_fma_f64 %6, %6, %7, %6\n v_add_f32 %11, %11, %10\n")
             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(a0), "+v"(a1), "+v"(a2),
               "+v"(a3));
d0 += a0 + a1 + a2 + a3;
BODY64_CLOSE

// ---- LDS: broadcast b128 read, b64 read, b64 write, b32 write (per-lane addresses) ----
#define BODYL_OPEN(NAME)                                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk, float seed) {    \
        __shared__ float4 lds[1024];                                                                  \
        lds[threadIdx.x] = make_float4(seed, seed, seed, seed);                                       \
        lds[threadIdx.x + 256] = lds[threadIdx.x];                                                    \
        lds[threadIdx.x + 512] = lds[threadIdx.x];                                                    \
        lds[threadIdx.x + 768] = lds[threadIdx.x];                                                    \
        __syncthreads();                                                                              \
        float4 q0 = lds[0], q1 = lds[1], q2 = lds[2], q3 = lds[3];                                    \
        double e0 = seed, e1 = seed * 2.0;                                                            \
        const unsigned addr_u = (unsigned)(threadIdx.x >> 6) * 4096u;                                 \
        const unsigned addr_l = addr_u + (threadIdx.x & 63) * 8u;                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int it = 0; it < kIters; ++it) {
#define BODYL_CLOSE                                                                                   \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = q0.x + q1.y + q2.z + q3.w + q0.w + q1.x + (float)(e0 + e1);         \
    }
BODYL_OPEN(k_ds_read_b128_bcast)
asm volatile(R4(R4("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n") "s_waitcnt lgkmcnt(0)\n")
             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
             : "v"(addr_u)
             : "memory");
BODYL_CLOSE
BODYL_OPEN(k_ds_read_b64_lane)
asm volatile(R4(R4("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n") "s_waitcnt lgkmcnt(0)\n")
             : "+v"(e0), "+v"(e1)
             : "v"(addr_l)
             : "memory");
BODYL_CLOSE
BODYL_OPEN(k_ds_write_b64_lane)
asm volatile(R4(R4("ds_write_b64 %2, %0\n ds_write_b64 %2, %1 offset:512\n") "s_waitcnt lgkmcnt(0)\n")
             :
             : "v"(e0), "v"(e1), "v"(addr_l)
             : "memory");
BODYL_CLOSE
BODYL_OPEN(k_ds_write_b32_lane)
asm volatile(R4(R4("ds_write_b32 %2, %0\n ds_write_b32 %2, %1 offset:512\n") "s_waitcnt lgkmcnt(0)\n")
             :
             : "v"(q0.x), "v"(q1.x), "v"(addr_l)
             : "memory");
BODYL_CLOSE
// 16 VALU + 16 LDS broadcast reads: do LDS instructions take VALU issue slots?
BODYL_OPEN(k_mix_valu_ds)
float a0 = seed, a1 = seed * 2, a2 = seed * 3, a3 = seed * 4;
asm volatile(R4("ds_read_b128 %0, %8\n v_add_f32 %4, %4, %5\n v_add_f32 %6, %6, %7\n ds_read_b128 %1, %8 offset:16\n"
                "v_add_f32 %5, %5, %4\n v_add_f32 %7, %7, %6\n ds_read_b128 %2, %8 offset:32\n v_add_f32 %4, %4, %5\n"
                "v_add_f32 %6, %6, %7\n ds_read_b128 %3, %8 offset:48\n v_add_f32 %5, %5, %4\n v_add_f32 %7, %7, %6\n"
                "s_waitcnt lgkmcnt(0)\n")
             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
             : "v"(addr_u)
             : "memory");
q0.x += a0 + a1 + a2 + a3;
BODYL_CLOSE

struct Test {
    const char *name;
    void (*fn)(float *, unsigned long long *, float);
    int insts_per_trip;  // wave-instructions of the kind under test (per trip)
    const char *note;
};

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    float *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    CHECK(hipMalloc(&clk, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const Test tests[] = {
        {"v_add_f32", k_add_f32, 32, ""},
        {"v_sub_f32", k_sub_f32, 32, ""},
        {"v_mul_f32", k_mul_f32, 32, ""},
        {"v_max_f32", k_max_f32, 32, ""},
        {"v_fma_f32", k_fma_f32, 32, ""},
        {"v_and_b32", k_and_b32, 32, ""},
        {"v_add_u32", k_add_u32, 32, ""},
        {"v_lshlrev_b32", k_lshl_b32, 32, ""},
        {"v_med3_f32", k_med3_f32, 32, "VOP3"},
        {"v_mad_u32_u24", k_mad_u32_u24, 32, "VOP3"},
        {"v_bfe_u32", k_bfe_u32, 32, "VOP3"},
        {"v_lshl_or_b32", k_lshl_or, 32, "VOP3"},
        {"v_mov_b32", k_mov_b32, 32, ""},
        {"v_rndne_f32", k_rndne_f32, 32, ""},
        {"v_cvt_i32_f32", k_cvt_i32_f32, 32, ""},
        {"v_sqrt_f32", k_sqrt_f32, 32, "transcendental"},
        {"v_rcp_f32", k_rcp_f32, 32, "transcendental"},
        {"v_or_b32", k_or_b32, 32, ""},
        {"v_xor_b32", k_xor_b32, 32, ""},
        {"v_sub_u32", k_sub_u32, 32, ""},
        {"v_lshrrev_b32", k_lshr_b32, 32, ""},
        {"v_ashrrev_i32", k_ashr_i32, 32, ""},
        {"v_min_f32", k_min_f32, 32, ""},
        {"v_fmac_f32", k_fmac_f32, 32, ""},
        {"v_mul_u32_u24", k_mul_u32_u24, 32, ""},
        {"v_mul_lo_u32", k_mul_lo_u32, 32, ""},
        {"v_max_u32", k_max_u32, 32, ""},
        {"v_min_u32", k_min_u32, 32, ""},
        {"v_ldexp_f32", k_ldexp_f32, 32, ""},
        {"v_add_f16", k_add_f16, 32, ""},
        {"v_mul_legacy_f32", k_mul_legacy, 32, ""},
        {"v_bfi_b32", k_bfi_b32, 32, "VOP3"},
        {"v_perm_b32", k_perm_b32, 32, "VOP3"},
        {"v_add3_u32", k_add3_u32, 32, "VOP3"},
        {"v_and_or_b32", k_and_or, 32, "VOP3"},
        {"v_max3_f32", k_max3_f32, 32, "VOP3"},
        {"v_alignbit_b32", k_alignbit, 32, "VOP3"},
        {"v_add_lshl_u32", k_add_lshl, 32, "VOP3"},
        {"v_lshl_add_u32", k_lshl_add, 32, "VOP3"},
        {"v_xad_u32", k_xad, 32, "VOP3"},
        {"v_cvt_f32_u32", k_cvt_f32_u32, 32, ""},
        {"v_cvt_f32_i32", k_cvt_f32_i32, 32, ""},
        {"v_fract_f32", k_fract_f32, 32, ""},
        {"v_exp_f32", k_exp_f32, 32, ""},
        {"v_sin_f32", k_sin_f32, 32, ""},
        {"v_rsq_f32", k_rsq_f32, 32, ""},
        {"v_not_b32", k_not_b32, 32, ""},
        {"v_bfrev_b32", k_bfrev, 32, ""},
        {"v_floor_f32", k_floor_f32, 32, ""},
        {"v_trunc_f32", k_trunc_f32, 32, ""},
        {"v_ffbh_u32", k_ffbh, 32, ""},
        {"v_cvt_u32_f32", k_bcnt_like, 32, ""},
        {"v_cmp_lt_f32 vcc", k_cmp_vcc, 32, ""},
        {"v_cmp_eq_u32 vcc", k_cmp_eq_u32, 32, ""},
        {"v_cndmask_b32 (vcc)", k_cndmask, 32, ""},
        {"v_cndmask_b32 (sgpr pair)", k_cndmask_sgpr, 32, "VOP3"},
        {"v_addc_co_u32", k_addc_co, 32, ""},
        {"v_add_co_u32", k_add_co, 32, ""},
        {"v_sub_f32 s, v", k_sub_f32_sgpr, 32, "SGPR src0"},
        {"v_add_f32 literal, v", k_add_f32_literal, 32, "32-bit literal"},
        {"v_fma_f32 v, v, 1.0", k_fma_f32_inline, 32, "inline constant"},
        {"v_fma_f32 v, s, v", k_fma_f32_sgpr, 32, "SGPR operand"},
        {"v_fma_f32 -v, v, |v|", k_fma_f32_mods, 32, "neg/abs modifiers"},
        {"v_add_f32_e64 v, v, -v", k_add_f32_e64, 32, "VOP3 encoding of a VOP2 op"},
        {"v_mul_f32_e64 clamp", k_mul_f32_e64_clamp, 32, "VOP3 + clamp"},
        {"mix 16 v_add_f32 + 16 v_cmp(sgpr)", k_mix_fast_slow, 32, "cyc per VALU instruction (3 = serial 2 + 4)"},
        {"mix 16 v_max_f32 + 16 s_add_u32", k_mix_slow_salu, 16, "cyc per VALU instruction"},
        {"mix 8 v_add_f32 + 24 SALU", k_mix_valu_3salu, 8, "cyc per VALU instruction"},
        {"v_readfirstlane_b32", k_readfirstlane, 32, ""},
        {"v_writelane_b32", k_writelane, 32, ""},
        {"v_cmp_lt_f32(vcc)+v_cndmask", k_cmp_cndmask, 32, "16 + 16"},
        {"v_cmp_lt_f32 -> SGPR pair", k_cmp_sgpr, 32, "ballot"},
        {"v_mbcnt_lo + v_mbcnt_hi", k_mbcnt, 32, "16 + 16"},
        {"v_readlane_b32", k_readlane, 32, ""},
        {"v_mov_b32_dpp row_shr:1", k_dpp_mov, 32, ""},
        {"v_add_f32_dpp row_shr:1", k_dpp_add, 32, ""},
        {"v_fma_f64", k_fma_f64, 32, ""},
        {"v_mul_f64", k_mul_f64, 32, ""},
        {"v_add_f64", k_add_f64, 32, ""},
        {"v_pk_fma_f32", k_pk_fma_f32, 32, "2 fp32 results per lane"},
        {"v_pk_mul_f32", k_pk_mul_f32, 32, "2 fp32 results per lane"},
        {"v_pk_add_f32", k_pk_add_f32, 32, "2 fp32 results per lane"},
        {"v_lshlrev_b64", k_lshl_b64, 32, ""},
        {"v_cvt_f32_f64 + v_cvt_f64_f32", k_cvt_f64_f32, 32, "16 + 16"},
        {"mix 16 v_add_f32 + 16 s_add_u32", k_mix_valu_salu, 16, "cyc per VALU instruction; = v_add_f32 alone if SALU issues beside it"},
        {"mix 16 v_fma_f64 + 16 v_add_f32", k_mix_f32_f64, 32, ""},
        {"ds_read_b128 (broadcast)", k_ds_read_b128_bcast, 32, "LDS"},
        {"ds_read_b64 (lane-contiguous)", k_ds_read_b64_lane, 32, "LDS"},
        {"ds_write_b64 (lane-contiguous)", k_ds_write_b64_lane, 32, "LDS"},
        {"ds_write_b32 (lane-contiguous)", k_ds_write_b32_lane, 32, "LDS"},
        {"mix 32 v_add_f32 + 16 ds_read_b128", k_mix_valu_ds, 32, "cyc per VALU instruction; = v_add_f32 alone if LDS issues beside it"},
    };
    const int waves_per_simd[] = {1, 2, 4, 8};
    printf("%-40s %8s %8s %8s %8s   (issue cycles per wave64 instruction per SIMD at 1/2/4/8 waves per SIMD; GHz measured)\n", "instruction",
           "w=1", "w=2", "w=4", "w=8");
    for (const Test &t : tests) {
        printf("%-40s", t.name);
        double ghz_last = 0;
        for (int w : waves_per_simd) {
            // w waves per SIMD = 4w waves per CU = w workgroups of 256 threads per CU
            const int grid = cus * w;
            hipLaunchKernelGGL(t.fn, dim3(grid), dim3(256), 0, 0, out, clk, 1.0f);  // warm-up
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            unsigned long long ticks = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(t.fn, dim3(grid), dim3(256), 0, 0, out, clk, 1.0f);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) {
                    best = ms;
                    CHECK(hipMemcpy(&ticks, clk, 8, hipMemcpyDeviceToHost));
                }
            }
            // s_memtime counts at a fixed 100 MHz on gfx9-family parts; the shader clock is taken from the 1-wave
            // v_add_f32 row below when available.  Report against the 2.4 GHz nominal clock as the guide does.
            const double t_s = best * 1e-3;
            const double insts = (double)kIters * t.insts_per_trip;  // per wave
            const double cyc = t_s * 2.4e9 / (insts * w);           // per SIMD: w waves share it
            ghz_last = (double)ticks / t_s / 1e9;
            printf(" %8.3f", cyc);
        }
        printf("   memtime %.3f GHz  %s\n", ghz_last, t.note);
    }
    return 0;
}
