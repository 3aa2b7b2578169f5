// Micro-benchmark: does a wave64 VALU instruction cost less when half of the wave is masked off?  gfx950 executes a wave64
// instruction as passes over the SIMD's lanes (v_add_f32: 2 cycles per instruction at 8 waves per SIMD,
// profiles/r03_ubench_valu_issue_rate.txt); if a pass whose 32 lanes are all inactive is skipped, a divergent phase of the
// march that keeps its active lanes in one half of the wave runs at twice the rate, and the lane <-> pixel map matters.
// Calibration only -- not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o exec_half tools/ubench/exec_half.hip && ./exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int OP>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned long long mask, unsigned long long *clk) {
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float c = 1.0001f, d = 0.0003f;
    asm volatile("" : "+v"(c), "+v"(d));
    if ((mask >> (threadIdx.x & 63u)) & 1ull) {
        for (int i = 0; i < iters; i++) {
            if (OP == 0)
                asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                             "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            else if (OP == 1)
                asm volatile("v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[0,1,1]\n"
                             "v_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[0,1,1]\n"
                             "v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[0,1,1]\n"
                             "v_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[0,1,1]\nv_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[0,1,1]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            else
                asm volatile("v_cos_f32 %0, %0\nv_cos_f32 %1, %1\nv_cos_f32 %2, %2\nv_cos_f32 %3, %3\nv_cos_f32 %4, %4\nv_cos_f32 %5, %5\nv_cos_f32 %6, %6\nv_cos_f32 %7, %7\n"
                             "v_cos_f32 %0, %0\nv_cos_f32 %1, %1\nv_cos_f32 %2, %2\nv_cos_f32 %3, %3\nv_cos_f32 %4, %4\nv_cos_f32 %5, %5\nv_cos_f32 %6, %6\nv_cos_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - m0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }  // shader cycles, 100 MHz ticks
}

static unsigned long long *g_clk;
static double g_ghz;
template <int OP>
static float run(float *out, int iters, unsigned long long mask) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 4;  // 512 threads = 8 waves per block, 4 blocks per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, iters, ~0ull, g_clk);  // pre-roll at full width
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, iters, mask, g_clk);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; g_ghz = (double)g_clk[0] / ((double)g_clk[1] * 10.0); }
    }
    return best;
}

int main() {
    float *out;
    CHECK(hipMalloc(&out, 256 * 4 * 512 * sizeof(float)));
    CHECK(hipHostMalloc(&g_clk, 16));
    struct Case { char name[40]; unsigned long long mask; };
    Case cases[64];
    int nc = 0;
    auto add = [&](const char *n, unsigned long long m) { snprintf(cases[nc].name, 40, "%s", n); cases[nc].mask = m; nc++; };
    add("all 64 lanes", ~0ull);
    const int counts[] = {48, 40, 36, 33, 32, 31, 28, 24, 20, 17, 16, 15, 12, 8, 4, 2, 1};
    for (int c : counts) { char n[40]; snprintf(n, 40, "lanes 0-%d", c - 1); add(n, c == 64 ? ~0ull : ((1ull << c) - 1ull)); }
    add("lanes 32-63", 0xffffffff00000000ull);
    add("even lanes (32)", 0x5555555555555555ull);
    add("every 4th lane (16)", 0x1111111111111111ull);
    add("every 8th lane (8)", 0x0101010101010101ull);
    add("lanes 0-7 + 32-39 (16)", 0x000000ff000000ffull);
    add("lanes 0-15 + 32-47 (32)", 0x0000ffff0000ffffull);
    add("lanes 0-3 of each 16 (16)", 0x000f000f000f000full);
    add("lane 63", 1ull << 63);
    const int iters = 20000;
    printf("# wall time of a kernel of 8 waves per SIMD x %d x 16 instructions, by the lanes left active (ms; ratio to all lanes)\n", iters);
    printf("%-26s | %-18s | %-18s | %-18s\n", "active lanes", "v_fma_f32", "v_fma_mix_f32", "v_cos_f32");
    float base[3] = {0, 0, 0};
    for (int ci = 0; ci < nc; ci++) { Case &c = cases[ci];
        float t0 = run<0>(out, iters, c.mask); double g0 = g_ghz;
        float t1 = run<1>(out, iters, c.mask); double g1 = g_ghz;
        float t2 = run<2>(out, iters / 4, c.mask); double g2 = g_ghz;
        if (!base[0]) { base[0] = t0; base[1] = t1; base[2] = t2; }
        printf("%-26s | %8.3f (%4.2f) %4.2f GHz | %8.3f (%4.2f) %4.2f GHz | %8.3f (%4.2f) %4.2f GHz\n", c.name, t0, t0 / base[0], g0, t1, t1 / base[1], g1, t2, t2 / base[2], g2);
    }
    return 0;
}
