// Micro-benchmark: issue cost (shader cycles per wave-instruction per SIMD) of the VALU ops the raymarch loops use, at
// 1/2/4/8 waves per SIMD.  Calibration only -- not part of the product.
//
// Round 3: clock-independent.  Cycles come from s_memtime stamped by every wave around its loop
// (tick = shader cycle, MI355X_MICROARCH.md "s_memtime tick"), the clock actually held from s_memrealtime (100 MHz)
// over the same interval; every measurement runs >= 20 ms after a 60 ms pre-roll of the densest loop, so the DVFS
// state is the sustained one.  The round-1 version assumed 2.4 GHz over 0.5-1 ms kernels after a cold start.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long cyc, rt, rt0, rt1; unsigned hwid, xcc; };

#define PROLOGUE                                                                                  \
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;                 \
    float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                             \
    float c = 1.0001f, d = 0.0003f;                                                               \
    asm volatile("" : "+v"(c), "+v"(d));                                                          \
    for (int i = 0; i < 64; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d)); \
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

#define EPILOGUE                                                                                  \
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;           \
    if ((threadIdx.x & 63) == 0) { Stamp s; s.cyc = m1 - m0; s.rt = r1 - r0; s.rt0 = r0; s.rt1 = r1; s.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4); \
        s.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s; }

// 8 independent chains, 16 instructions per loop iteration (the loop's s_add / s_cmp / s_cbranch are 3 scalar per 16 vector)
#define DEFINE_KERNEL(NAME, ASM8) DEFINE_KERNEL_C(NAME, ASM8, "memory")
#define DEFINE_KERNEL_C(NAME, ASM8, ...)                                                          \
__global__ __launch_bounds__(512) void NAME(float *out, Stamp *st, int iters) {                   \
    PROLOGUE                                                                                      \
    for (int i = 0; i < iters; i++) {                                                             \
        asm volatile(ASM8 ASM8                                                                    \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(c), "v"(d) : __VA_ARGS__);                                             \
    }                                                                                             \
    EPILOGUE                                                                                      \
}

#define OP1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
// one dependent chain: every instruction reads the result of the one before it
#define DEP3(op) op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n"
// two chains alternating (dependent distance 2)
#define DEP3x2(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n"

DEFINE_KERNEL(k_fma, OP3("v_fma_f32"))
DEFINE_KERNEL(k_fma_dep, DEP3("v_fma_f32"))
DEFINE_KERNEL(k_fma_dep2, DEP3x2("v_fma_f32"))
DEFINE_KERNEL(k_add, OP2("v_add_f32"))
DEFINE_KERNEL(k_mul, OP2("v_mul_f32"))
DEFINE_KERNEL(k_min, OP2("v_min_f32"))
DEFINE_KERNEL(k_cos, OP1("v_cos_f32"))
DEFINE_KERNEL(k_exp, OP1("v_exp_f32"))
DEFINE_KERNEL(k_rcp, OP1("v_rcp_f32"))
DEFINE_KERNEL(k_fract, OP1("v_fract_f32"))
DEFINE_KERNEL(k_cvt_flr, OP1("v_cvt_flr_i32_f32"))
DEFINE_KERNEL(k_cvt_f32_i32, OP1("v_cvt_f32_i32"))
DEFINE_KERNEL(k_cvt_ub0, OP1("v_cvt_f32_ubyte0"))
DEFINE_KERNEL(k_min_i32, OP2("v_min_i32"))
DEFINE_KERNEL(k_add_u32, OP2("v_add_u32"))
DEFINE_KERNEL(k_lshl, OP2("v_lshlrev_b32"))
DEFINE_KERNEL(k_lshl_add, OP3("v_lshl_add_u32"))
DEFINE_KERNEL(k_add3, OP3("v_add3_u32"))
DEFINE_KERNEL(k_mad_i24, OP3("v_mad_i32_i24"))
DEFINE_KERNEL(k_med3_i32, OP3("v_med3_i32"))
DEFINE_KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[0,1,1]\n"
                         "v_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[0,1,1]\n"
                         "v_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[0,1,1]\n")
// a VALU stream with scalar work beside it: 8 v_fma + 8 s_add per half (the staged step loop carries ~12 scalar per ~47 vector)
DEFINE_KERNEL_C(k_fma_salu, "v_fma_f32 %0, %0, %8, %9\ns_add_u32 s20, s20, 1\nv_fma_f32 %1, %1, %8, %9\ns_add_u32 s21, s21, 1\nv_fma_f32 %2, %2, %8, %9\ns_add_u32 s20, s20, 1\n"
                          "v_fma_f32 %3, %3, %8, %9\ns_add_u32 s21, s21, 1\nv_fma_f32 %4, %4, %8, %9\ns_add_u32 s20, s20, 1\nv_fma_f32 %5, %5, %8, %9\ns_add_u32 s21, s21, 1\n"
                          "v_fma_f32 %6, %6, %8, %9\ns_add_u32 s20, s20, 1\nv_fma_f32 %7, %7, %8, %9\ns_add_u32 s21, s21, 1\n", "s20", "s21", "scc")
// v_cmp + s_and_saveexec pairs among fmas (divergent loop control)
DEFINE_KERNEL_C(k_fma_cmp, "v_fma_f32 %0, %0, %8, %9\nv_cmp_lt_f32 vcc, %8, %0\nv_fma_f32 %1, %1, %8, %9\nv_cmp_lt_f32 vcc, %8, %1\nv_fma_f32 %2, %2, %8, %9\nv_cmp_lt_f32 vcc, %8, %2\n"
                         "v_fma_f32 %3, %3, %8, %9\nv_cmp_lt_f32 vcc, %8, %3\n", "vcc")

DEFINE_KERNEL(k_fmac, OP2("v_fmac_f32"))
DEFINE_KERNEL(k_add_e64, OP2("v_add_f32_e64"))
DEFINE_KERNEL(k_max, OP2("v_max_f32"))
DEFINE_KERNEL(k_mov, OP1("v_mov_b32"))
DEFINE_KERNEL(k_and, OP2("v_and_b32"))
DEFINE_KERNEL(k_cvt_f32_f16, OP1("v_cvt_f32_f16"))
DEFINE_KERNEL(k_perm, OP3("v_perm_b32"))
DEFINE_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n")
// v_cndmask variants: mask in VCC written once by a VALU compare before the stream; compare + select pairs; mask in an SGPR pair
DEFINE_KERNEL_C(k_cndmask_vcc_set, "v_cmp_gt_f32 vcc, %8, %9\ns_nop 4\n"
                                   "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                                   "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n", "vcc")
DEFINE_KERNEL_C(k_cmp_cndmask, "v_cmp_gt_f32 vcc, %8, %0\nv_cndmask_b32 %0, %0, %9, vcc\nv_cmp_gt_f32 vcc, %8, %1\nv_cndmask_b32 %1, %1, %9, vcc\n"
                               "v_cmp_gt_f32 vcc, %8, %2\nv_cndmask_b32 %2, %2, %9, vcc\nv_cmp_gt_f32 vcc, %8, %3\nv_cndmask_b32 %3, %3, %9, vcc\n", "vcc")
DEFINE_KERNEL_C(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\nv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                                "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\nv_cndmask_b32_e64 %5, %5, %8, s[20:21]\nv_cndmask_b32_e64 %6, %6, %8, s[20:21]\nv_cndmask_b32_e64 %7, %7, %8, s[20:21]\n", "s20", "s21")
DEFINE_KERNEL(k_bfi, OP3("v_bfi_b32"))
DEFINE_KERNEL(k_xor, OP2("v_xor_b32"))
DEFINE_KERNEL(k_floor, OP1("v_floor_f32"))
DEFINE_KERNEL(k_rndne, OP1("v_rndne_f32"))
// additivity: do costs add when a cheap op and a dear one alternate?  (per PAIR)
DEFINE_KERNEL(k_add_cos, "v_add_f32 %0, %0, %8\nv_cos_f32 %1, %1\nv_add_f32 %2, %2, %8\nv_cos_f32 %3, %3\nv_add_f32 %4, %4, %8\nv_cos_f32 %5, %5\nv_add_f32 %6, %6, %8\nv_cos_f32 %7, %7\n")
DEFINE_KERNEL(k_add_mix, "v_add_f32 %0, %0, %8\nv_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[0,1,1]\nv_add_f32 %2, %2, %8\nv_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[0,1,1]\n"
                         "v_add_f32 %4, %4, %8\nv_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[0,1,1]\nv_add_f32 %6, %6, %8\nv_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[0,1,1]\n")
DEFINE_KERNEL(k_add_fract, "v_add_f32 %0, %0, %8\nv_fract_f32 %1, %1\nv_add_f32 %2, %2, %8\nv_fract_f32 %3, %3\nv_add_f32 %4, %4, %8\nv_fract_f32 %5, %5\nv_add_f32 %6, %6, %8\nv_fract_f32 %7, %7\n")
// s_nop 0 among fmas (the compiler's hazard pads in the staged step loop): per v_fma
DEFINE_KERNEL(k_fma_nop, "v_fma_f32 %0, %0, %8, %9\ns_nop 0\nv_fma_f32 %1, %1, %8, %9\ns_nop 0\nv_fma_f32 %2, %2, %8, %9\ns_nop 0\nv_fma_f32 %3, %3, %8, %9\ns_nop 0\n"
                         "v_fma_f32 %4, %4, %8, %9\ns_nop 0\nv_fma_f32 %5, %5, %8, %9\ns_nop 0\nv_fma_f32 %6, %6, %8, %9\ns_nop 0\nv_fma_f32 %7, %7, %8, %9\ns_nop 0\n")
// LDS element reads among adds (8 ds_read_u8 of one address register + wait, as the staged step does), per group of 8 reads + 8 adds
__global__ __launch_bounds__(512) void k_lds_u8(float *out, Stamp *st, int iters) {
    __shared__ unsigned char win[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) win[i] = (unsigned char)i;
    __syncthreads();
    PROLOGUE
    unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)win + (threadIdx.x * 37u) % 4000u;
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
        unsigned l0, l1, l2, l3, l4, l5, l6, l7;
        asm volatile("ds_read_u8 %0, %8\n\tds_read_u8 %1, %8 offset:1\n\tds_read_u8 %2, %8 offset:64\n\tds_read_u8 %3, %8 offset:65\n\t"
                     "ds_read_u8 %4, %8 offset:512\n\tds_read_u8 %5, %8 offset:513\n\tds_read_u8 %6, %8 offset:576\n\tds_read_u8 %7, %8 offset:577\n\t"
                     "v_add_f32 %9, %9, %10\nv_add_f32 %9, %9, %10\nv_add_f32 %9, %9, %10\nv_add_f32 %9, %9, %10\nv_add_f32 %9, %9, %10\nv_add_f32 %9, %9, %10\nv_add_f32 %9, %9, %10\nv_add_f32 %9, %9, %10\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(l4), "=&v"(l5), "=&v"(l6), "=&v"(l7) : "v"(addr), "v"(a1), "v"(c));
        acc += l0 + l1 + l2 + l3 + l4 + l5 + l6 + l7;
    }
    a2 = (float)acc;
    EPILOGUE
}

// packed ops need register pairs
__global__ __launch_bounds__(512) void k_pk_fma(float *out, Stamp *st, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    PROLOGUE
    f2 b0 = {a0, 1.f}, b1 = b0 + 1.f, b2 = b0 + 2.f, b3 = b0 + 3.f, b4 = b0 + 4.f, b5 = b0 + 5.f, b6 = b0 + 6.f, b7 = b0 + 7.f;
    f2 cc = {c, 0.9999f}, dd = {d, 0.0001f};
    for (int i = 0; i < iters; i++) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
                     "v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(cc), "v"(dd));
    }
    f2 s = b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
    a1 = s.x; a2 = s.y;
    EPILOGUE
}
__global__ __launch_bounds__(512) void k_fma_f64(float *out, Stamp *st, int iters) {
    PROLOGUE
    double b0 = a0, b1 = b0 + 1., b2 = b0 + 2., b3 = b0 + 3., b4 = b0 + 4., b5 = b0 + 5., b6 = b0 + 6., b7 = b0 + 7.;
    double cc = 1.0000001, dd = 0.0003;
    asm volatile("" : "+v"(cc), "+v"(dd));
    for (int i = 0; i < iters; i++) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %1, %1, %8, %9\nv_fma_f64 %2, %2, %8, %9\nv_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\nv_fma_f64 %5, %5, %8, %9\nv_fma_f64 %6, %6, %8, %9\nv_fma_f64 %7, %7, %8, %9\n"
                     "v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %1, %1, %8, %9\nv_fma_f64 %2, %2, %8, %9\nv_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\nv_fma_f64 %5, %5, %8, %9\nv_fma_f64 %6, %6, %8, %9\nv_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(cc), "v"(dd));
    }
    a1 = (float)(b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7);
    EPILOGUE
}

// The dense sample of the cell march as the compiler schedules it, with the cell in registers instead of a load
// (vk_kernels.hpp march_stream's trip without its buffer_load / ds_read): the instruction MIX and DEPENDENCE structure of
// the real loop, no memory.  Reported as cycles per trip; divide by the trip's VALU count (printed by --disasm users;
// 43 in round 2) for cycles per instruction.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k_sample_body(float *out, Stamp *st, int iters) {
    PROLOGUE
    float px = a0 * 0.01f, py = a1 * 0.01f, pz = a2 * 0.01f, A = 0.f, Gr = 0.f, Gg = 0.f, Gb = 0.f, t = 0.f;
    const float sx = 1e-4f * c, sy = 2e-4f * c, sz = 3e-4f * c, dt = 1e-3f * c, fn = 256.f * c;
    unsigned acc = 0, lutb = threadIdx.x;
    union { unsigned u[4]; half2_t h[4]; } cell;
    cell.u[0] = 0x2c003c00u + threadIdx.x; cell.u[1] = 0x2c004000u; cell.u[2] = 0x2c004200u; cell.u[3] = 0x2c004400u;
    for (int i = 0; i < iters; i++) {
        asm volatile("" : "+v"(cell.u[0]), "+v"(cell.u[1]), "+v"(cell.u[2]), "+v"(cell.u[3]));  // "loaded" afresh every trip
        const float ux = fmaf(px, fn, -0.5f), uy = fmaf(py, fn, -0.5f), uz = fmaf(pz, fn, -0.5f);
        int ix, iy, iz;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(ux));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(uy));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iz) : "v"(uz));
        const float fx = __builtin_amdgcn_fractf(ux), fy = __builtin_amdgcn_fractf(uy), fz = __builtin_amdgcn_fractf(uz);
        // stands for the three table look-ups (v_lshl_add + ds_read_b32 each) + v_add3 + the buffer_load's address
        unsigned ox, oy, oz;
        asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ox) : "v"(ix), "v"(lutb));
        asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(oy) : "v"(iy), "v"(lutb));
        asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(oz) : "v"(iz), "v"(lutb));
        asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(acc) : "v"(ox), "v"(oy), "v"(oz));
        const float c00 = fmaf(fx, (float)cell.h[0].y, (float)cell.h[0].x), c10 = fmaf(fx, (float)cell.h[1].y, (float)cell.h[1].x);
        const float c01 = fmaf(fx, (float)cell.h[2].y, (float)cell.h[2].x), c11 = fmaf(fx, (float)cell.h[3].y, (float)cell.h[3].x);
        const float l0 = fmaf(fy, c10 - c00, c00), l1 = fmaf(fy, c11 - c01, c01);
        const float v = fmaf(fz, l1 - l0, l0);
        float s = fmaf(fminf(v, 229.5f), (float)(1.0 / (255.0 * 1.1)), (float)(-0.1 / 1.1));
        s = fminf(fmaxf(s, 0.0f), 1.0f);
        const float a = (s * s) * fmaf(-2.0f, s, 3.0f);
        const float cr = __builtin_amdgcn_cosf(a * 0.999997f), cg = __builtin_amdgcn_cosf(fmaf(a, 1.699995f, 0.15f)), cb = __builtin_amdgcn_cosf(fmaf(a, 0.4f, 0.2f));
        const float w = (1.0f - A) * a;
        Gr = fmaf(w, cr, Gr); Gg = fmaf(w, cg, Gg); Gb = fmaf(w, cb, Gb);
        A = A + w * 1e-6f;
        px = px + sx; py = py + sy; pz = pz + sz; t = t + dt;
    }
    a3 = px + py + pz + t + A + Gr + Gg + Gb + (float)acc;
    EPILOGUE
}

typedef void (*kern_t)(float *, Stamp *, int);
struct Entry { const char *name; kern_t k; int per_iter; };

int main(int argc, char **argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 512 * sizeof(float)));
    Stamp *st; CHECK(hipMalloc(&st, (size_t)cus * 8 * 8 * sizeof(Stamp)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<Entry> ks = {{"v_fma_f32", k_fma, 16}, {"v_fma_f32 dependent", k_fma_dep, 16}, {"v_fma_f32 2 chains", k_fma_dep2, 16}, {"v_add_f32", k_add, 16}, {"v_mul_f32", k_mul, 16}, {"v_min_f32", k_min, 16},
        {"v_pk_fma_f32", k_pk_fma, 16}, {"v_fma_f64", k_fma_f64, 16}, {"v_fma_mix_f32", k_fma_mix, 16}, {"v_cos_f32", k_cos, 16}, {"v_exp_f32", k_exp, 16}, {"v_rcp_f32", k_rcp, 16},
        {"v_fract_f32", k_fract, 16}, {"v_cvt_flr_i32_f32", k_cvt_flr, 16}, {"v_cvt_f32_i32", k_cvt_f32_i32, 16}, {"v_cvt_f32_ubyte0", k_cvt_ub0, 16},
        {"v_min_i32", k_min_i32, 16}, {"v_add_u32", k_add_u32, 16}, {"v_lshlrev_b32", k_lshl, 16}, {"v_lshl_add_u32", k_lshl_add, 16}, {"v_add3_u32", k_add3, 16}, {"v_mad_i32_i24", k_mad_i24, 16},
        {"v_med3_i32", k_med3_i32, 16}, {"v_fmac_f32 (VOP2)", k_fmac, 16}, {"v_add_f32_e64 (VOP3)", k_add_e64, 16}, {"v_max_f32", k_max, 16}, {"v_mov_b32", k_mov, 16}, {"v_and_b32", k_and, 16},
        {"v_cvt_f32_f16", k_cvt_f32_f16, 16}, {"v_perm_b32", k_perm, 16}, {"v_cndmask_b32 (vcc never written)", k_cndmask, 16},
        {"v_cndmask_b32 (vcc from v_cmp, per 8+1)", k_cndmask_vcc_set, 2}, {"v_cmp + v_cndmask (per pair)", k_cmp_cndmask, 8}, {"v_cndmask_b32_e64 (SGPR pair)", k_cndmask_sgpr, 16},
        {"v_bfi_b32", k_bfi, 16}, {"v_xor_b32", k_xor, 16}, {"v_floor_f32", k_floor, 16}, {"v_rndne_f32", k_rndne, 16},
        {"v_add + v_cos 1:1 (per pair)", k_add_cos, 8}, {"v_add + v_fma_mix 1:1 (per pair)", k_add_mix, 8}, {"v_add + v_fract 1:1 (per pair)", k_add_fract, 8},
        {"v_fma + s_nop 0 1:1 (per v_fma)", k_fma_nop, 16}, {"8 ds_read_u8 + 8 v_add + wait (per group)", k_lds_u8, 1},
        {"v_fma + s_add 1:1 (per v_fma)", k_fma_salu, 16}, {"v_fma + v_cmp 1:1 (per pair)", k_fma_cmp, 8},
        {"cell sample body (per trip)", k_sample_body, 1}};
    const double target_ms = argc > 1 ? atof(argv[1]) : 25.0;
    printf("device: %s, %d CUs\n", prop.name, cus);
    printf("cell = cycles per wave-instruction on ONE SIMD, median over the SIMDs: (first start .. last end of the waves that ran on it, s_memrealtime x the wave's own clock) / (their wave-instructions);\n"
           "       `Nw` = waves that actually ran on the median SIMD (HW_ID); (clock GHz = s_memtime / s_memrealtime x 100 MHz; chip-level T lane-ops/s from the launch's wall time)\n");
    printf("every cell: >= %.0f ms kernel after a 60 ms pre-roll of the v_fma loop at 8 waves per SIMD; cell = cycles (clock GHz, chip-level T lane-ops/s from the launch's wall time)\n", target_ms);
    printf("%-32s | %24s | %24s | %24s | %24s\n", "op", "4 waves/CU", "8 waves/CU", "16 waves/CU", "32 waves/CU");
    double simd_per = 0, waves_per_simd = 0, span_over_wall = 0;  // filled by run(): per-SIMD accounting by where the waves actually ran
    int per_iter_now = 16;
    auto run = [&](kern_t k, int w, int iters, float *ms_out, double *cyc_med, double *ghz_med) {
        const int threads = 64 * 4 * w;
        dim3 block(threads > 512 ? 512 : threads);
        const int blocks_per_cu = threads > 512 ? threads / 512 : 1;
        dim3 grid(cus * blocks_per_cu);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, grid, block, 0, 0, out, st, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(ms_out, e0, e1));
        const int n_waves = cus * 4 * w;
        std::vector<Stamp> h(n_waves);
        CHECK(hipMemcpy(h.data(), st, n_waves * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> cy, gz;
        for (auto &s : h) { cy.push_back((double)s.cyc); gz.push_back(s.rt ? (double)s.cyc / (double)s.rt * 0.1 : 0.0); }
        std::sort(cy.begin(), cy.end()); std::sort(gz.begin(), gz.end());
        *cyc_med = cy[cy.size() / 2]; *ghz_med = gz[gz.size() / 2];
        // Where did the waves run?  Key = (XCC, SE, SH, CU, SIMD) from HW_ID / XCC_ID.  A SIMD's cost per instruction = the shader cycles
        // between its first wave's start and its last wave's end / the wave-instructions its waves issued -- whatever the dispatcher's
        // placement was (it is NOT "w waves on every SIMD": a launch of exactly the chip's capacity does not land evenly).
        struct Acc { unsigned long long t0 = ~0ull, t1 = 0; int n = 0; };
        std::map<unsigned long long, Acc> by;
        unsigned long long g0 = ~0ull, g1 = 0;
        for (auto &s : h) {
            const unsigned long long key = ((unsigned long long)(s.xcc & 0xf) << 32) | (s.hwid & 0xfff0u & ~0xc0u);  // drop wave_id[3:0] and pipe_id[7:6]
            Acc &a = by[key];
            a.t0 = std::min(a.t0, s.rt0); a.t1 = std::max(a.t1, s.rt1); a.n++;
            g0 = std::min(g0, s.rt0); g1 = std::max(g1, s.rt1);
        }
        std::vector<double> per, nw;
        for (auto &kv : by) {
            const double span_cyc = (double)(kv.second.t1 - kv.second.t0) * 10.0 * (*ghz_med);  // realtime ticks are 10 ns
            per.push_back(span_cyc / ((double)kv.second.n * iters * per_iter_now));
            nw.push_back(kv.second.n);
        }
        std::sort(per.begin(), per.end()); std::sort(nw.begin(), nw.end());
        simd_per = per[per.size() / 2]; waves_per_simd = nw[nw.size() / 2];
        span_over_wall = (double)(g1 - g0) * 1e-5 / *ms_out;
        if (getenv("UB_VERBOSE")) printf("\n   [w=%d: %zu SIMDs used, waves per SIMD min/med/max %g/%g/%g, first start to last end %.2f ms of %.2f ms wall]", w, by.size(), nw.front(), waves_per_simd, nw.back(), (double)(g1 - g0) * 1e-5, *ms_out);
    };
    for (auto &en : ks) {
        if (argc > 2) {  // only the rows whose name contains one of the comma-separated patterns
            bool hit = false;
            std::string pats(argv[2]);
            for (size_t a = 0; a <= pats.size();) { size_t b = pats.find(',', a); if (b == std::string::npos) b = pats.size(); if (b > a && std::string(en.name).find(pats.substr(a, b - a)) != std::string::npos) hit = true; a = b + 1; }
            if (!hit) continue;
        }
        printf("%-32s", en.name);
        for (int w : {1, 2, 4, 8}) {
            float ms; double cyc, ghz;
            run(k_fma, 8, 400000, &ms, &cyc, &ghz);  // pre-roll (~60 ms)
            per_iter_now = en.per_iter;
            run(en.k, w, 2000, &ms, &cyc, &ghz);     // calibrate
            int iters = (int)std::min(4.0e7, std::max(2000.0, 2000.0 * target_ms / std::max(ms, 1e-3f)));
            run(en.k, w, iters, &ms, &cyc, &ghz);
            const double per = cyc / ((double)iters * en.per_iter * w);
            // cross-check that does not use s_memtime at all: wall time of the launch (HIP events) x the clock from s_memrealtime's
            // interval would be circular, so print the chip-level rate instead: wave-instructions x 64 lanes / wall time, in T lane-ops/s
            const double tops = (double)cus * 4 * w * (double)iters * en.per_iter * 64.0 / (ms * 1e-3) / 1e12;
            (void)per;
            printf(" | %5.2f %3.0fw (%4.2f, %5.1fT)", simd_per, waves_per_simd, ghz, tops);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
