// TEST INFRASTRUCTURE: host build of vk::count_trips (vokselis_amd/csrc/vk_trips.hpp) against the loop it counts,
// `for (t = t0; t < t1; t = t + dt)` (raycast_naive.wgsl:101), on random and adversarial (t0, t1, dt).
// usage: trips_fuzz <cases> <seed>; prints "bad <n> of <cases>" and exits non-zero on any mismatch.
#include "vk_trips.hpp"

#include <stdio.h>
#include <stdlib.h>

static uint32_t loop_trips(float t0, float t1, float dt) {
    uint32_t n = 0;
    for (volatile float t = t0; t < t1; t = t + dt) n++;
    return n;
}
static uint64_t state;
static uint64_t rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; }
static float unit() { return (float)((rnd() >> 40) / 16777216.0); }

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 1000000;
    state = argc > 2 ? strtoull(argv[2], nullptr, 0) : 88172645463325252ull;
    long bad = 0;
    for (long c = 0; c < cases; c++) {
        const int kind = (int)(c % 10);
        // step lengths of 128^3 .. 2048^3 volumes at dt_scale 0.25 .. 2
        static const float scales[5] = {1.0f / 128, 1.0f / 512, 1.0f / 1024, 1.0f / 4096, 1.0f / 8192};
        float dt = scales[rnd() % 5] * (0.5f + unit());
        if (kind == 3) {  // few mantissa bits: sums that land exactly half-way between two floats (round-to-nearest-even parity)
            const uint32_t b = vk::trips_bits(dt) & ~((1u << (rnd() % 22)) - 1u);
            dt = vk::trips_float(b);
        }
        float t0 = kind == 4 ? 0.0f : unit() * 2.5f;  // (eye inside the box: t0 = max(t0, 0) = 0, t climbs through every binade)
        if (kind == 5) t0 = unit() * 0.01f;
        if (kind == 8) t0 = 1.0f - dt * (float)(rnd() % 5);  // starts just below a power of two
        float t1 = t0 + unit() * 1.8f;
        if (kind == 6) t1 = t0 + dt * (float)(rnd() % 6);  // 0 .. 5 steps, ends on or next to a sum
        if (kind == 7) t1 = (rnd() & 1) ? t0 : t0 - 0.1f;   // empty range
        if (kind == 9) { const float k = (float)(rnd() % 700); t1 = t0; for (int i = 0; i < (int)k; i++) t1 = t1 + dt; }  // t1 IS a value of the sequence
        const uint32_t got = vk::count_trips(t0, t1, dt), want = loop_trips(t0, t1, dt);
        if (got != want) {
            if (bad++ < 10) printf("t0=%a t1=%a dt=%a: count_trips %u, the loop %u\n", t0, t1, dt, got, want);
        }
    }
    printf("bad %ld of %ld\n", bad, cases);
    return bad != 0;
}
