// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the raymarch
// loop uses, at 1/2/4/8 waves per SIMD.  Calibration only -- not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define DEFINE_KERNEL(NAME, ASM8)                                                             \
__global__ __launch_bounds__(512) void NAME(float *out, int iters) {                          \
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;             \
    float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                         \
    float c = 1.0001f, d = 0.0003f;                                                           \
    for (int i = 0; i < iters; i++) {                                                         \
        asm volatile(ASM8                                                                     \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(c), "v"(d));                                                       \
    }                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
}

#define OP1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"

DEFINE_KERNEL(k_fma, OP3("v_fma_f32"))
DEFINE_KERNEL(k_add, OP2("v_add_f32"))
DEFINE_KERNEL(k_mul, OP2("v_mul_f32"))
DEFINE_KERNEL(k_cos, OP1("v_cos_f32"))
DEFINE_KERNEL(k_exp, OP1("v_exp_f32"))
DEFINE_KERNEL(k_rcp, OP1("v_rcp_f32"))
DEFINE_KERNEL(k_floor, OP1("v_floor_f32"))
DEFINE_KERNEL(k_fract, OP1("v_fract_f32"))
DEFINE_KERNEL(k_cvt_i32, OP1("v_cvt_i32_f32"))
DEFINE_KERNEL(k_cvt_flr, OP1("v_cvt_flr_i32_f32"))
DEFINE_KERNEL(k_cvt_ub0, OP1("v_cvt_f32_ubyte0"))
DEFINE_KERNEL(k_cvt_ub3, OP1("v_cvt_f32_ubyte3"))
DEFINE_KERNEL(k_cvt_f32_i32, OP1("v_cvt_f32_i32"))
DEFINE_KERNEL(k_min_i32, OP2("v_min_i32"))
DEFINE_KERNEL(k_add_u32, OP2("v_add_u32"))
DEFINE_KERNEL(k_lshl, OP2("v_lshlrev_b32"))
DEFINE_KERNEL(k_and_or, OP3("v_and_or_b32"))
DEFINE_KERNEL(k_mad_u32_u24, OP3("v_mad_u32_u24"))
DEFINE_KERNEL(k_mul_lo_u32, OP2("v_mul_lo_u32"))
DEFINE_KERNEL(k_med3_i32, OP3("v_med3_i32"))
DEFINE_KERNEL(k_sdwa_sub, "v_sub_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
                          "v_sub_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
                          "v_sub_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
                          "v_sub_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
                          "v_sub_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
                          "v_sub_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
                          "v_sub_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
                          "v_sub_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n")

// packed ops need register pairs
__global__ __launch_bounds__(512) void k_pk_fma(float *out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {threadIdx.x * 0.001f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 c = {1.0001f, 0.9999f}, d = {0.0003f, 0.0001f};
    for (int i = 0; i < iters; i++) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ __launch_bounds__(512) void k_mad_u64_u32(float *out, int iters) {
    unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned m = 3 + threadIdx.x, n = 5;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(n) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ __launch_bounds__(512) void k_lshl_add_u64(float *out, int iters) {
    unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long m = 3 + threadIdx.x;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_lshl_add_u64 %0, %0, 0, %8\nv_lshl_add_u64 %1, %1, 0, %8\nv_lshl_add_u64 %2, %2, 0, %8\nv_lshl_add_u64 %3, %3, 0, %8\n"
                     "v_lshl_add_u64 %4, %4, 0, %8\nv_lshl_add_u64 %5, %5, 0, %8\nv_lshl_add_u64 %6, %6, 0, %8\nv_lshl_add_u64 %7, %7, 0, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

typedef void (*kern_t)(float *, int);
struct Entry { const char *name; kern_t k; };

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    float *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 512 * sizeof(float)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<Entry> ks = {{"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_pk_fma_f32", k_pk_fma}, {"v_cos_f32", k_cos},
        {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp}, {"v_floor_f32", k_floor}, {"v_fract_f32", k_fract}, {"v_cvt_i32_f32", k_cvt_i32},
        {"v_cvt_flr_i32_f32", k_cvt_flr}, {"v_cvt_f32_ubyte0", k_cvt_ub0}, {"v_cvt_f32_ubyte3", k_cvt_ub3}, {"v_cvt_f32_i32", k_cvt_f32_i32},
        {"v_min_i32", k_min_i32}, {"v_add_u32", k_add_u32}, {"v_lshlrev_b32", k_lshl}, {"v_and_or_b32", k_and_or}, {"v_mad_u32_u24", k_mad_u32_u24},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_med3_i32", k_med3_i32}, {"v_sub_u32_sdwa", k_sdwa_sub}, {"v_mad_u64_u32", k_mad_u64_u32},
        {"v_lshl_add_u64", k_lshl_add_u64}};
    const int iters = 20000;
    printf("%-22s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD at the measured clock; w = waves/SIMD)\n", "op", "w=1", "w=2", "w=4", "w=8");
    // clock estimate: assume 2.4 GHz nominal; also print ns
    for (auto &en : ks) {
        printf("%-22s", en.name);
        for (int w : {1, 2, 4, 8}) {
            // one block per CU-slot: block = 256*w threads -> 4*w waves per CU -> w waves per SIMD (1 block per CU)
            int threads = 64 * 4 * w;
            dim3 grid(cus), block(threads > 512 ? 512 : threads);
            int blocks_per_cu = threads > 512 ? threads / 512 : 1;
            grid.x = cus * blocks_per_cu;
            hipLaunchKernelGGL(en.k, grid, block, 0, 0, out, 100);  // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(en.k, grid, block, 0, 0, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double instr_per_simd = (double)iters * 8 * w;
            double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
            printf(" %8.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
