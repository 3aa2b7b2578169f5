// vk_trips.hpp -- the trip count of `for (var t = t0; t < t1; t = t + dt)` (raycast_naive.wgsl:101), exactly, without running it.
//
// The reference accumulates t by repeated binary32 addition, so the number of iterations is NOT ceil((t1 - t0) / dt): every
// addition rounds, and the rounding changes with t's binade.  Inside one binade [B, 2B) every t is a multiple of the binade's
// ulp u, and t + dt rounds to t + inc with ONE increment, inc = dt rounded to a multiple of u (round-to-nearest-even; taken here
// by adding and subtracting 1.5 B, whose ulp is u) -- with one exception: when dt sits exactly half-way between two multiples, the
// FIRST addition in the binade depends on the parity of t / u; its sum is even, and from then on the increment is the even one,
// which is what the rounding of dt gives.  So, per binade: one real addition (it settles the parity), then q more values
// t + j inc -- exact in binary32: multiples of u below 2B -- with q chosen so that they all stay below min(2B, t1) whatever the
// reciprocal rounded to, then real additions up to the binade's end or t1; the addition that crosses into the next binade is a
// real one too.  A ray of C2 crosses two or three binades of t: ~60 instructions per ray replace one addition and one compare
// per ITERATION, and the march carries an integer `left` instead of (t, t1, dt) (vk_march.hpp: RayState).
// tests/test_trips_cpu.py fuzzes this function (host build) against the loop itself; on the GPU the per-pixel trip counts of
// every parity test go through it.
//
// Preconditions (check_render, vk_render.hip): t0 >= 0, dt > 0 and dt >= 8 ulp of the largest t the march visits, so every
// addition advances t.  A dt that does not (the reference would never terminate) returns the saturated count.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define VK_HD __host__ __device__ __forceinline__
#else
#define VK_HD inline
#endif

namespace vk {

constexpr uint32_t kTripsForever = 0xffffffffu;

VK_HD uint32_t trips_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
#endif
}
VK_HD float trips_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

VK_HD float trips_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);  // 1 ulp
#else
    return 1.0f / x;
#endif
}

// #{ i >= 0 : t_i < t1 },  t_0 = t0,  t_{i+1} = fl(t_i + dt)
VK_HD uint32_t count_trips(const float t0, const float t1, const float dt) {
    uint32_t n = 0;
    float t = t0;  // the next value of the sequence, not yet tested against t1
    while (t < t1) {
        const uint32_t e = trips_bits(t) & 0x7f800000u;
        const float B = trips_float(e), B2 = B + B;
        float tn = t + dt;  // t passes the test: one iteration, and its addition is a real one
        n++;
        if (tn == t) return kTripsForever;
        // the closed form inside t's binade: a normal t, dt < B / 2 (so that dt + 1.5 B stays in [B, 2B)), the sum still inside and below t1
        const float lim = B2 < t1 ? B2 : t1;
        if ((int)(e - 0x00800000u < 0x7f000000u) & (int)(dt + dt < B) & (int)(tn < lim)) {  // (one branch, not three)
            const float C = B + 0.5f * B;
            const float inc = (dt + C) - C;  // dt to the nearest multiple of ulp(B), ties to even; > 0 (dt >= 8 ulp)
            // q values tn + j inc, j = 0 .. q - 1, pass the test, and tn + q inc -- the next to be looked at -- is still in the binade, when
            // q inc < lim - tn.  lim - tn is exact (the two are within a factor two); the reciprocal and the product carry < 2^-21
            // relative error, the factor 1 - 2^-20 puts x below the true quotient, and so is floor(x).  (x <= 2^23: the conversion and
            // q inc are exact, the sum is a multiple of ulp(B) below 2B: the fma does not round.)
            const float x = (lim - tn) * (trips_rcp(inc) * (1.0f - 0x1p-20f));
            const float qf = __builtin_floorf(x);
            if (inc > 0.0f && qf >= 1.0f) {
                n += (uint32_t)qf;
                tn = __builtin_fmaf(qf, inc, tn);
            }
        }
        t = tn;
    }
    return n;
}

}  // namespace vk
