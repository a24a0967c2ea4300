// lds_issue.hip — LDS instruction costs on gfx950 that decide the shape of the BGK kernel's compaction / accumulate
// phases (round 3): masked writes, many lanes to one address, ds_add_f32 with same-address conflicts, strided b128
// reads.  Same conventions as valu_issue.hip: W waves per SIMD resident, cycles per wave-instruction per SIMD at the
// nominal 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_issue.hip -o tools/_out/lds_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int kIters = 1000;
constexpr int kPer = 16;  // LDS instructions per trip

// mode: how the per-lane byte address is formed (within the wave's private 4 KB window)
//  0 lane * 8            (contiguous 8-byte slots)
//  1 lane * 4            (contiguous 4-byte slots)
//  2 hit lanes (every 7th) contiguous, the others all on ONE address
//  3 (lane % 9) * 4      (7-way same-address groups)
//  4 0                   (all lanes one address)
//  5 (lane >> 4) * 16    (4 distinct 16-byte rows, for b128 reads)
//  6 (lane & 3) * 16 + (lane >> 2 & 3) * 64 ... per-lane 16-byte rows, 16 distinct
__device__ __forceinline__ uint32_t lane_addr(int mode, uint32_t lane) {
    switch (mode) {
    case 0: return lane * 8;
    case 1: return lane * 4;
    case 2: return (lane % 7 == 0) ? (lane / 7) * 8 : 1024;
    case 3: return (lane % 9) * 4;
    case 4: return 0;
    case 5: return (lane >> 4) * 16;
    case 6: return (lane & 15) * 16;
    default: return lane * 16;
    }
}

#define R16(a) a a a a a a a a a a a a a a a a

template <int kOp>
__global__ __launch_bounds__(256) void k_lds(float *out, int mode, unsigned long long execmask) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 4096];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 4096 / 4; i += 256) ((float *)lds)[i] = 0.0f;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)&lds[wv * 4096];
    uint32_t addr = base + lane_addr(mode, lane);
    float v0 = lane * 0.5f, v1 = 1.0f;
    float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if (kOp == 16 || kOp == 17) {  // exec-masked ds_write_b64 / ds_write_b32
        for (int it = 0; it < kIters; ++it) {
            if (kOp == 16)
                asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %2\n" R16("ds_write_b64 %0, %1\n")
                             "s_mov_b64 exec, s[20:21]\n s_waitcnt lgkmcnt(0)\n"
                             :
                             : "v"(addr), "v"(make_float2(v0, v1)), "s"(execmask)
                             : "memory", "s20", "s21");
            else
                asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %2\n" R16("ds_write_b32 %0, %1\n")
                             "s_mov_b64 exec, s[20:21]\n s_waitcnt lgkmcnt(0)\n"
                             :
                             : "v"(addr), "v"(v0), "s"(execmask)
                             : "memory", "s20", "s21");
        }
    } else if (kOp == 18) {  // ds_bpermute_b32 (lane * 4 in the address register)
        uint32_t t, ad = ((lane * 7u) & 63u) << 2;
        for (int it = 0; it < kIters; ++it) {
            asm volatile(R16("ds_bpermute_b32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(t) : "v"(ad), "v"(lane) : "memory");
            r0 += (float)t;
        }
    } else if (kOp == 4) {  // exec-masked write2: only the lanes in execmask write
        for (int it = 0; it < kIters; ++it)
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %3\n" R16("ds_write2_b32 %0, %1, %2 offset1:1\n")
                         "s_mov_b64 exec, s[20:21]\n s_waitcnt lgkmcnt(0)\n"
                         :
                         : "v"(addr), "v"(v0), "v"(v1), "s"(execmask)
                         : "memory", "s20", "s21");
    } else {
        for (int it = 0; it < kIters; ++it) {
            if (kOp == 0) asm volatile(R16("ds_write2_b32 %0, %1, %2 offset1:1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v0), "v"(v1) : "memory");
            if (kOp == 1) asm volatile(R16("ds_add_f32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v1) : "memory");
            if (kOp == 2) {
                float4 t;
                asm volatile(R16("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(t) : "v"(addr) : "memory");
                r0 += t.x;
            }
            if (kOp == 3) asm volatile(R16("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v0) : "memory");
            if (kOp == 5) {
                float2 t;
                asm volatile(R16("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(t) : "v"(addr) : "memory");
                r0 += t.x;
            }
            if (kOp == 6) {
                float t;
                asm volatile(R16("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(t) : "v"(addr) : "memory");
                r0 += t;
            }
            if (kOp == 7) asm volatile(R16("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(make_float2(v0, v1)) : "memory");
            if (kOp == 8) asm volatile(R16("ds_add_u32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(lane) : "memory");
            if (kOp == 9) asm volatile(R16("ds_add_u64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(make_uint2(lane, 0u)) : "memory");
            if (kOp == 10) {
                uint32_t t;
                asm volatile(R16("ds_add_rtn_u32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(t) : "v"(addr), "v"(lane) : "memory");
                r0 += t;
            }
            if (kOp == 11) asm volatile(R16("ds_max_f32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v1) : "memory");
            if (kOp == 12) asm volatile(R16("ds_add_f64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"((double)v1) : "memory");
            if (kOp == 13) asm volatile(R16("ds_pk_add_bf16 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(lane) : "memory");
            if (kOp == 14) asm volatile(R16("ds_max_u32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(lane) : "memory");
            if (kOp == 15) asm volatile(R16("ds_or_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(lane) : "memory");
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + ((float *)lds)[threadIdx.x];
}

struct Test {
    const char *name;
    void (*fn)(float *, int, unsigned long long);
    int mode;
    unsigned long long execmask;
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned long long nine = 0x0102040810204081ull;  // every 7th lane: 10 lanes
    const Test tests[] = {
        {"ds_write2_b32 64 lanes contiguous", k_lds<0>, 0, 0},
        {"ds_write2_b32 10 hit lanes + 54 lanes on ONE address", k_lds<0>, 2, 0},
        {"ds_write2_b32 exec = 10 lanes (+2 s_mov exec per 16)", k_lds<4>, 0, nine},
        {"ds_write2_b32 exec = all lanes (+2 s_mov exec per 16)", k_lds<4>, 0, ~0ull},
        {"ds_write_b32 64 lanes contiguous", k_lds<3>, 1, 0},
        {"ds_write_b32 10 hit + 54 on ONE address", k_lds<3>, 2, 0},
        {"ds_write_b64 64 lanes contiguous", k_lds<7>, 0, 0},
        {"ds_write_b64 exec = 10 lanes", k_lds<16>, 0, nine},
        {"ds_write_b64 exec = all lanes", k_lds<16>, 0, ~0ull},
        {"ds_write_b32 exec = 10 lanes", k_lds<17>, 1, nine},
        {"ds_bpermute_b32", k_lds<18>, 0, 0},
        {"ds_add_f32 64 distinct", k_lds<1>, 1, 0},
        {"ds_add_f32 9 addresses x 7 lanes", k_lds<1>, 3, 0},
        {"ds_add_f32 one address x 64 lanes", k_lds<1>, 4, 0},
        {"ds_add_u32 64 distinct", k_lds<8>, 1, 0},
        {"ds_add_u32 9 addresses x 7 lanes", k_lds<8>, 3, 0},
        {"ds_add_u32 one address x 64", k_lds<8>, 4, 0},
        {"ds_add_u64 64 distinct (8 B stride)", k_lds<9>, 0, 0},
        {"ds_add_u64 one address x 64", k_lds<9>, 4, 0},
        {"ds_add_rtn_u32 64 distinct", k_lds<10>, 1, 0},
        {"ds_add_rtn_u32 9 x 7", k_lds<10>, 3, 0},
        {"ds_max_f32 64 distinct", k_lds<11>, 1, 0},
        {"ds_add_f64 64 distinct (8 B stride)", k_lds<12>, 0, 0},
        {"ds_pk_add_bf16 64 distinct", k_lds<13>, 1, 0},
        {"ds_max_u32 64 distinct", k_lds<14>, 1, 0},
        {"ds_or_b32 9 x 7", k_lds<15>, 3, 0},
        {"ds_read_b128 broadcast (1 address)", k_lds<2>, 4, 0},
        {"ds_read_b128 4 rows x 16 lanes", k_lds<2>, 5, 0},
        {"ds_read_b128 16 rows x 4 lanes", k_lds<2>, 6, 0},
        {"ds_read_b128 64 distinct rows", k_lds<2>, 7, 0},
        {"ds_read_b64 64 contiguous", k_lds<5>, 0, 0},
        {"ds_read_b64 9 addresses x 7", k_lds<5>, 3, 0},
        {"ds_read_b32 64 contiguous", k_lds<6>, 1, 0},
        {"ds_read_b32 one address", k_lds<6>, 4, 0},
    };
    printf("%-58s %8s %8s %8s\n", "LDS instruction (cycles per wave-instruction per SIMD)", "w=2", "w=4", "w=8");
    for (const Test &t : tests) {
        printf("%-58s", t.name);
        for (int w : {2, 4, 8}) {
            const int grid = cus * w;
            hipLaunchKernelGGL(t.fn, dim3(grid), dim3(256), 0, 0, out, t.mode, t.execmask);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(t.fn, dim3(grid), dim3(256), 0, 0, out, t.mode, t.execmask);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double cyc = best * 1e-3 * 2.4e9 / ((double)kIters * kPer * w);
            printf(" %8.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
