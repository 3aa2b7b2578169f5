// Checks the exact semantics of v_fract_f32 / v_cvt_flr_i32_f32 on gfx950 against
// f = u - floor(u) (clamped below 1) and i = (int)floor(u), over special and random inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include <cstdint>
__global__ void k(const float *in, float *fr, int *fl, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float u = in[i], f; int q;
    asm volatile("v_fract_f32 %0, %1" : "=v"(f) : "v"(u));
    asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(q) : "v"(u));
    fr[i] = f; fl[i] = q;
}
int main() {
    std::vector<float> h = {-1e-9f, -0.0f, 0.0f, -1.0f, -0.5f, 0.99999994f, 1.0f, 255.5f, -0.25f, 127.99999f, 2047.9999f, -5.9604645e-8f, -2.9802322e-8f, 1e-40f, -1e-40f, 3.0f, -3.0f, 16777216.0f, -1.4e-45f};
    uint32_t s = 12345;
    for (int i = 0; i < 4000000; i++) { s = s * 1664525u + 1013904223u; float v = ((int)(s >> 8) - 8388608) * (1.0f / 4096.0f); h.push_back(v); s = s * 1664525u + 1013904223u; h.push_back(((int)(s >> 8) - 8388608) * 1e-10f); }
    int n = h.size();
    float *din, *dfr; int *dfl;
    hipMalloc(&din, n * 4); hipMalloc(&dfr, n * 4); hipMalloc(&dfl, n * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, din, dfr, dfl, n);
    std::vector<float> fr(n); std::vector<int> fl(n);
    hipMemcpy(fr.data(), dfr, n * 4, hipMemcpyDeviceToHost); hipMemcpy(fl.data(), dfl, n * 4, hipMemcpyDeviceToHost);
    long bad_clamped = 0, bad_plain = 0, bad_flr = 0;
    for (int i = 0; i < n; i++) {
        float u = h[i], fo = floorf(u), plain = u - fo, cl = plain >= 1.0f ? 0x1.fffffep-1f : plain;
        if (fr[i] != cl) bad_clamped++;
        if (fr[i] != plain) bad_plain++;
        if (fl[i] != (int)fo) bad_flr++;
        if (i < 19) printf("u=%.9g  v_fract=%.9g (plain %.9g)  v_cvt_flr=%d (floor %d)\n", u, fr[i], plain, fl[i], (int)fo);
    }
    printf("n=%d mismatches: vs clamped-model %ld, vs plain u-floor(u) %ld, cvt_flr %ld\n", n, bad_clamped, bad_plain, bad_flr);
    return 0;
}
