// Micro-benchmark: what does the staged march's window fill reach on its own?  global_load_lds_dwordx4 gathers of
// 16-byte pieces (8 consecutive rows = one 128-byte line) from a table far larger than the caches, in the access pattern of
// vk_staged.hpp's fill -- a wave fetches a box of Es slices x Em rows x Ef pieces from 8^3-brick storage, waits for all of it
// (s_waitcnt vmcnt(0)), "marches" for a given number of cycles, and moves one slab further along its ray bundle -- at the occupancies
// the march runs at (waves per SIMD set by the LDS each wave holds).  No arithmetic on the data: this is the ceiling of the fill.
// Calibration only -- not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o lds_gather tools/ubench/lds_gather.hip && ./lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

struct Params {
    const unsigned char *base;
    unsigned long long layerB;  // bytes of one layer of bricks along S
    unsigned npf, nbm, nbs;     // pieces along F, bricks along M, bricks along S
    unsigned Es, Em, Ef;        // window: slices x rows x pieces
    unsigned rounds;            // windows per wave
    unsigned advance;           // slices the window moves per round (T): Es - advance slices are fetched again, as in the march
    unsigned think;             // s_sleep units between a fill and the next (the steps' time), 0 = none
    unsigned coherent;          // 1: the 16 waves of a "tile" walk neighbouring boxes (shared lines); 0: every wave its own random place
};

__global__ __launch_bounds__(64) void fill_kernel(Params P, unsigned long long *sink) {
    extern __shared__ unsigned char win[];
    const unsigned lane = threadIdx.x;
    const unsigned win_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)win;
    // where this wave's bundle enters the volume: a pseudo-random (M, F) place; waves of one "tile" of 16 sit side by side
    unsigned h = (P.coherent ? blockIdx.x / 16u : blockIdx.x) * 2654435761u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned rowsTot = P.nbm * 8u, pcsTot = P.npf;
    unsigned m0 = (h % (rowsTot - 4u * P.Em - 8u)), f0 = ((h >> 12) % (pcsTot - 4u * P.Ef - 2u));
    if (P.coherent) { const unsigned w = blockIdx.x & 15u; m0 += (w >> 2) * (P.Em - 2u); f0 += (w & 3u) * (P.Ef - 1u); }
    unsigned s0 = (h >> 20) % 8u;
    const unsigned slicePieces = P.Em * P.Ef;
    unsigned long long acc = 0;
    for (unsigned r = 0; r < P.rounds; r++) {
        for (unsigned j = 0; j < slicePieces; j += 64u) {
            const unsigned q = j + lane;
            if (q < slicePieces) {
                const unsigned m = q / P.Ef, f = q - m * P.Ef;
                const unsigned mm = min(m0 + m + (r >> 2), P.nbm * 8u - 1u);  // the bundle drifts one row every four slabs
                const unsigned voff = ((min(f0 + f, P.npf - 1u) + P.npf * (mm >> 3)) << 10) | ((mm & 7u) << 4);
                unsigned sv = s0 + r * P.advance;
                const unsigned char *sbase = P.base + P.layerB * (unsigned long long)(sv >> 3) + ((sv & 7u) << 7);
                unsigned lds_dst = win_lds + j * 16u;
                unsigned keep_m0;
                asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
                for (unsigned si = 0; si < P.Es;) {
                    const unsigned run = min(8u - (sv & 7u), P.Es - si);
                    for (unsigned k = 0; k < run; k++) {
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
                        sbase += 128;
                        lds_dst += slicePieces * 16u;
                    }
                    si += run; sv += run;
                    sbase += P.layerB - 1024u;
                }
                asm volatile("s_mov_b32 m0, %0" : : "s"(keep_m0) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<unsigned *>(win + ((lane * 16u) % (slicePieces * 16u)));
        for (unsigned t = 0; t < P.think; t++) __builtin_amdgcn_s_sleep(8);  // 8 x 64 cycles
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

// Fill + march, two ways.  MODE 0: as the staged march does today -- LDS-DMA fill, wait for all of it, then the slab's steps (here: a
// loop of `think` dependent-free v_fma_f32 pairs standing in for them).  MODE 1: the NEXT slab's pieces are requested into registers
// (global_load_dwordx4, up to 8 per lane) before this slab's steps, and written to the window (ds_write_b128) after them: same LDS,
// same occupancy, the fetch under the steps.
template <int MODE>
__global__ __launch_bounds__(64) void fill_march_kernel(Params P, unsigned long long *sink) {
    extern __shared__ unsigned char win[];
    const unsigned lane = threadIdx.x;
    const unsigned win_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)win;
    unsigned h = (P.coherent ? blockIdx.x / 16u : blockIdx.x) * 2654435761u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned rowsTot = P.nbm * 8u, pcsTot = P.npf;
    unsigned m0 = (h % (rowsTot - 4u * P.Em - 8u)), f0 = ((h >> 12) % (pcsTot - 4u * P.Ef - 2u));
    if (P.coherent) { const unsigned w = blockIdx.x & 15u; m0 += (w >> 2) * (P.Em - 2u); f0 += (w & 3u) * (P.Ef - 1u); }
    const unsigned s0 = (h >> 20) % 8u;
    const unsigned slicePieces = P.Em * P.Ef;  // <= 64: one piece per lane and slice
    const unsigned m = lane / P.Ef, f = lane - m * P.Ef;
    float a0 = lane * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, c = 1.0001f, d = 0.0003f;
    asm volatile("" : "+v"(c), "+v"(d));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    auto voff_of = [&](unsigned r) {
        const unsigned mm = min(m0 + m + (r >> 2), rowsTot - 1u);
        return ((min(f0 + f, P.npf - 1u) + P.npf * (mm >> 3)) << 10) | ((mm & 7u) << 4);
    };
    auto src_of = [&](unsigned r, unsigned si) {
        const unsigned sv = s0 + r * P.advance + si;
        return P.base + P.layerB * (unsigned long long)(sv >> 3) + ((sv & 7u) << 7);
    };
    auto request = [&](unsigned r) {  // slab r -> registers (Es <= 8)
        const unsigned voff = voff_of(r);
        if (lane < slicePieces) {
#define REQ(i, R) if (P.Es > i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(R) : "v"(voff), "s"(src_of(r, i)) : "memory");
            REQ(0, r0) REQ(1, r1) REQ(2, r2) REQ(3, r3) REQ(4, r4) REQ(5, r5) REQ(6, r6) REQ(7, r7)
#undef REQ
        }
    };
    if (MODE == 1) request(0);
    unsigned long long acc = 0;
    if (MODE == 0 && (P.coherent & 2u)) {
        // stagger: waves that start together and do the same work fill at the same time and march at the same time; offset them by a fraction of a round's steps
        const unsigned phase = (blockIdx.x * 2654435761u >> 28) & 7u;  // 0 .. 7 eighths of a round
        for (unsigned t = 0; t < P.think * phase / 8u; t++)
            asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                         "v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
    }
    if (MODE == 2) {
        // a rolling ring of slices: one LDS-DMA per cell of progress, K = P.Es - 1 slices requested ahead of the one being marched, the wait is for
        // the OLDEST slice only (vmcnt counts loads in order); the steps of one cell (think / T instructions) run under the rest
        const unsigned voff = voff_of(0);
        const unsigned K = P.Es - 1u, ringB = P.Es * slicePieces * 16u;
        unsigned keep_m0;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
        unsigned issued = 0;
        auto issue = [&]() {
            if (lane < slicePieces) {
                const unsigned dst = win_lds + (issued % P.Es) * slicePieces * 16u;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src_of(0, issued)), "s"(dst) : "memory");
            }
            issued++;
        };
        for (unsigned k = 0; k < K; k++) issue();
        const unsigned cells = P.rounds * P.advance, per_cell = P.think / P.advance;
        for (unsigned c = 0; c < cells; c++) {
            issue();
            // K loads may stay in flight: the slice of cell c (issued K + 1 loads ago) has landed
            if (K >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (K == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (K == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (K == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (K == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            acc += *reinterpret_cast<unsigned *>(win + ((lane * 16u) % ringB));
            for (unsigned t = 0; t < per_cell; t++)
                asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_mov_b32 m0, %0" : : "s"(keep_m0) : "memory");
        if (acc + (unsigned long long)(a0 + a1 + a2 + a3) == 0x1234567ull) sink[0] = acc;
        return;
    }
    for (unsigned r = 0; r < P.rounds; r++) {
        if (MODE == 0) {
            const unsigned voff = voff_of(r);
            if (lane < slicePieces) {
                unsigned lds_dst = win_lds, keep_m0;
                asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
                for (unsigned si = 0; si < P.Es; si++) {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src_of(r, si)), "s"(lds_dst) : "memory");
                    lds_dst += slicePieces * 16u;
                }
                asm volatile("s_mov_b32 m0, %0" : : "s"(keep_m0) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : : "memory");
            if (lane < slicePieces) {
                unsigned dst = win_lds + lane * 16u;
#define WR(i, R) if (P.Es > i) { asm volatile("ds_write_b128 %0, %1" : : "v"(dst), "v"(R) : "memory"); dst += slicePieces * 16u; }
                WR(0, r0) WR(1, r1) WR(2, r2) WR(3, r3) WR(4, r4) WR(5, r5) WR(6, r6) WR(7, r7)
#undef WR
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (r + 1 < P.rounds) request(r + 1);
        }
        acc += *reinterpret_cast<unsigned *>(win + ((lane * 16u) % (slicePieces * 16u)));
        for (unsigned t = 0; t < P.think; t++)
            asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                         "v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
    }
    if (acc + (unsigned long long)(a0 + a1 + a2 + a3) == 0x1234567ull) sink[0] = acc;
}

// the same fill at dword granularity: a lane moves 4 bytes (2 f16 voxels), a row is Ed dwords
__global__ __launch_bounds__(64) void fill_kernel_b32(Params P, unsigned long long *sink) {
    extern __shared__ unsigned char win[];
    const unsigned lane = threadIdx.x;
    const unsigned win_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)win;
    unsigned h = (P.coherent ? blockIdx.x / 16u : blockIdx.x) * 2654435761u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned rowsTot = P.nbm * 8u, dwTot = P.npf * 4u, Ed = P.Ef;  // (Ef holds the dwords per row here)
    unsigned m0 = (h % (rowsTot - 4u * P.Em - 8u)), d0 = ((h >> 12) % (dwTot - 4u * Ed - 8u));
    if (P.coherent) { const unsigned w = blockIdx.x & 15u; m0 += (w >> 2) * (P.Em - 2u); d0 += (w & 3u) * (Ed - 1u); }
    unsigned s0 = (h >> 20) % 8u;
    const unsigned sliceDw = P.Em * Ed;
    unsigned long long acc = 0;
    for (unsigned r = 0; r < P.rounds; r++) {
        for (unsigned j = 0; j < sliceDw; j += 64u) {
            const unsigned q = j + lane;
            if (q < sliceDw) {
                const unsigned m = q / Ed, d = min(d0 + (q - m * Ed), dwTot - 1u);
                const unsigned mm = min(m0 + m + (r >> 2), rowsTot - 1u);
                const unsigned voff = (((d >> 2) + P.npf * (mm >> 3)) << 10) | ((mm & 7u) << 4) | ((d & 3u) << 2);
                unsigned sv = s0 + r * P.advance;
                const unsigned char *sbase = P.base + P.layerB * (unsigned long long)(sv >> 3) + ((sv & 7u) << 7);
                unsigned lds_dst = win_lds + j * 4u;
                unsigned keep_m0;
                asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
                for (unsigned si = 0; si < P.Es;) {
                    const unsigned run = min(8u - (sv & 7u), P.Es - si);
                    for (unsigned k = 0; k < run; k++) {
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
                        sbase += 128;
                        lds_dst += sliceDw * 4u;
                    }
                    si += run; sv += run;
                    sbase += P.layerB - 1024u;
                }
                asm volatile("s_mov_b32 m0, %0" : : "s"(keep_m0) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<unsigned *>(win + ((lane * 4u) % (sliceDw * 4u)));
        for (unsigned t = 0; t < P.think; t++) __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

int main(int argc, char **argv) {
    // the C4 table: 1024^3 f16 padded to 1040^3 in 8^3 bricks of 1 KiB: 130^3 bricks = 2.2 GB per copy; three copies side by side = 6.7 GB
    const unsigned nb = 130, copies = 3;
    const unsigned long long layerB = (unsigned long long)nb * nb * 1024ull, bytes = layerB * nb * copies;
    unsigned char *tab;
    CHECK(hipMalloc(&tab, bytes));
    CHECK(hipMemset(tab, 1, bytes));
    unsigned long long *sink;
    CHECK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("# window fill by global_load_lds_dwordx4 from a %.1f GB table (8^3 bricks of 16-byte pieces), wait for all, next slab\n", bytes / 1e9);
    printf("# box Es x Em x Ef, slab advance T, LDS per wave -> waves per SIMD; GB/s of bytes written to LDS (every piece counts, also the\n"
           "# Es - T slices fetched again) and of the T NEW slices only; coherent = 16 waves of a tile walk neighbouring boxes\n");
    printf("%-20s %4s %7s %6s %9s %6s | %10s %10s %9s\n", "box (Es x Em x Ef)", "T", "LDS B", "w/SIMD", "coherent", "think", "GB/s LDS", "GB/s new", "ms");
    struct Case { unsigned Es, Em, Ef, T, lds, coherent, think; };
    std::vector<Case> cases;
    for (unsigned coherent : {0u, 1u})
        for (unsigned lds : {6400u, 8192u, 10240u, 13632u, 20480u}) {
            // f16 C4: ~14 rows x 3 pieces of 8 voxels; the slab as thick as the window holds
            const unsigned Em = 14, Ef = 3, Es = lds / (Em * Ef * 16u);
            if (Es < 3) continue;
            cases.push_back({Es, Em, Ef, Es - 1u, lds, coherent, 0u});
        }
    cases.push_back({12, 14, 3, 11, 10240, 1, 4});
    cases.push_back({12, 14, 3, 11, 10240, 1, 16});
    cases.push_back({6, 26, 5, 5, 12800, 1, 0});   // a group-sized box per wave, for the piece / line efficiency
    cases.push_back({12, 8, 8, 11, 12288, 1, 0});           // whole lines: 8 rows x 8 pieces
    for (auto &c : cases) {
        Params P{tab, layerB, nb * (8u * 2u / 16u), nb, nb * copies, c.Es, c.Em, c.Ef, 0, c.T, c.think, c.coherent};
        P.npf = nb;  // one 16-byte piece per brick along F (8 f16 voxels)
        if (c.Es * c.Em * c.Ef * 16u > c.lds || c.T < 1 || c.T > c.Es) { printf("bad case\n"); return 1; }
        const unsigned waves_per_cu = std::min(32u, 163840u / c.lds);
        const unsigned blocks = 256u * waves_per_cu * 4u;  // four rounds of resident waves
        P.rounds = std::min(64u, (nb * copies * 8u - 16u - c.Es) / c.T);
        if ((unsigned long long)(8u + P.rounds * c.T + c.Es + 8u) / 8ull * layerB > bytes) { printf("bad range\n"); return 1; }
        hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(64), c.lds, 0, P, sink);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(64), c.lds, 0, P, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double all = (double)blocks * P.rounds * c.Es * c.Em * c.Ef * 16.0, fresh = all * c.T / c.Es;
        char box[32]; snprintf(box, 32, "%u x %u x %u", c.Es, c.Em, c.Ef);
        printf("%-20s %4u %7u %6.1f %9u %6u | %10.0f %10.0f %9.3f   (slab-cells / us: %.0f)\n", box, c.T, c.lds, waves_per_cu / 4.0, c.coherent, c.think, all / ms / 1e6, fresh / ms / 1e6, ms,
               (double)blocks * P.rounds * c.T / ms / 1e3);
    }
    // the same boxes IN VOXELS (14 rows x 14 f16 voxels + the upper tap = 15 -> 8 dwords, against 3 pieces = 24 voxels), dword by dword
    printf("# dword granularity (global_load_lds_dword): box Es x Em rows x Ed dwords; windows per microsecond is what compares with the x4 boxes above\n");
    printf("%-20s %4s %7s %6s %9s | %10s %12s %9s\n", "box (Es x Em x Ed)", "T", "LDS B", "w/SIMD", "coherent", "GB/s LDS", "boxes / us", "ms");
    struct CaseD { unsigned Es, Em, Ed, T, lds, coherent; };
    std::vector<CaseD> dcases;
    for (unsigned lds : {6400u, 8192u, 10240u, 13632u}) {
        const unsigned Es16 = lds / (14u * 3u * 16u);  // the slab the x4 window holds in this budget
        dcases.push_back({Es16, 14, 8, Es16 - 1u, lds, 1});                       // same slab, same occupancy, fewer bytes
        const unsigned Es4 = lds / (14u * 8u * 4u);                              // or: the budget holds a thicker slab
        dcases.push_back({Es4, 14, 8, Es4 - 1u, lds, 1});
    }
    for (auto &c : dcases) {
        Params P{tab, layerB, nb, nb, nb * copies, c.Es, c.Em, c.Ed, 0, c.T, 0u, c.coherent};
        if (c.Es * c.Em * c.Ed * 4u > c.lds || c.T < 1 || c.T > c.Es) { printf("bad case\n"); return 1; }
        const unsigned waves_per_cu = std::min(32u, 163840u / c.lds);
        const unsigned blocks = 256u * waves_per_cu * 4u;
        P.rounds = std::min(64u, (nb * copies * 8u - 16u - c.Es) / c.T);
        if ((unsigned long long)(8u + P.rounds * c.T + c.Es + 8u) / 8ull * layerB > bytes) { printf("bad range\n"); return 1; }
        hipLaunchKernelGGL(fill_kernel_b32, dim3(blocks), dim3(64), c.lds, 0, P, sink);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(fill_kernel_b32, dim3(blocks), dim3(64), c.lds, 0, P, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double all = (double)blocks * P.rounds * c.Es * c.Em * c.Ed * 4.0;
        char box[32]; snprintf(box, 32, "%u x %u x %u", c.Es, c.Em, c.Ed);
        printf("%-20s %4u %7u %6.1f %9u | %10.0f %12.1f %9.3f   (slab-cells / us: %.0f)\n", box, c.T, c.lds, waves_per_cu / 4.0, c.coherent, all / ms / 1e6, (double)blocks * P.rounds / ms / 1e3, ms,
               (double)blocks * P.rounds * c.T / ms / 1e3);
    }
    // fill + march: today's order against the next slab requested into registers under this slab's steps
    printf("# fill + march: box 7 x 16 x 3 (T = 6), `think` x 8 v_fma_f32 per round stand in for the slab's steps (C4: ~13 steps x 50 instructions = 650)\n");
    printf("%-10s %7s %6s %6s | %12s %12s\n", "mode", "LDS B", "w/SIMD", "think", "us per round", "slab-cells/us");
    for (unsigned lds : {10240u, 12288u, 8192u})
        for (unsigned think : {0u, 40u, 80u, 160u})
            for (int mode = 0; mode < 4; mode++) {
                // (modes 2, 3: the rolling ring -- the same LDS, 6 / 3 slices requested ahead, the remaining slots of the window marched from)
                Params P{tab, layerB, nb, nb, nb * copies, mode == 3 ? 4u : 7u, 16, 3, 0, 6, think, 1};
                const unsigned waves_per_cu = std::min(32u, 163840u / lds);
                const unsigned blocks = 256u * waves_per_cu * 4u;
                P.rounds = 64;
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(fill_march_kernel<0>, dim3(blocks), dim3(64), lds, 0, P, sink);
                    else if (mode == 1) hipLaunchKernelGGL(fill_march_kernel<1>, dim3(blocks), dim3(64), lds, 0, P, sink);
                    else hipLaunchKernelGGL(fill_march_kernel<2>, dim3(blocks), dim3(64), lds, 0, P, sink);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                }
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                // a SIMD's time per round of one of its waves: launch time / (rounds x waves the SIMD ran)
                printf("%-10s %7u %6.1f %6u | %12.3f %12.0f\n", mode == 0 ? "lds-dma" : (mode == 1 ? "registers" : (mode == 2 ? "ring K=6" : "ring K=3")), lds, waves_per_cu / 4.0, think * 8u, ms * 1e3 / (P.rounds * 4.0 * waves_per_cu / 4.0),
                       (double)blocks * P.rounds * P.advance / ms / 1e3);
            }
    // the convoy: blocking fills with and without a staggered start
    printf("# blocking fills, box 7 x 16 x 3 (T = 6): all waves start in phase, against a start staggered by 0 .. 7 eighths of a round's steps\n");
    printf("%-10s %7s %6s %6s | %12s\n", "start", "LDS B", "w/SIMD", "think", "slab-cells/us");
    for (unsigned lds : {10240u, 12288u})
        for (unsigned think : {80u, 160u})
            for (unsigned stag : {0u, 1u}) {
                Params P{tab, layerB, nb, nb, nb * copies, 7, 16, 3, 256, 6, think, 1u | (stag << 1)};
                const unsigned waves_per_cu = std::min(32u, 163840u / lds);
                const unsigned blocks = 256u * waves_per_cu * 4u;
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(fill_march_kernel<0>, dim3(blocks), dim3(64), lds, 0, P, sink);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                }
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("%-10s %7u %6.1f %6u | %12.0f\n", stag ? "staggered" : "in phase", lds, waves_per_cu / 4.0, think * 8u, (double)blocks * P.rounds * P.advance / ms / 1e3);
            }
    // the latency of ONE fill: a single wave per CU (256 workgroups; then one per SIMD, two per SIMD), blocking fills back to back, nothing else on the machine
    printf("# latency of a blocking fill (box 7 x 16 x 3 = 5.4 KB in 7 LDS-DMA instructions) against the waves that fill at the same time\n");
    printf("%-24s | %14s\n", "waves on the machine", "us per fill");
    for (unsigned waves : {256u, 1024u, 2048u, 4096u, 8192u, 16384u}) {
        Params P{tab, layerB, nb, nb, nb * copies, 7, 16, 3, 64, 6, 0, 1};
        const unsigned lds = waves <= 4096u ? 40960u : (waves == 8192u ? 20480u : 10240u);  // (caps the waves per CU at 4 / 8 / 16)
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(fill_march_kernel<0>, dim3(waves), dim3(64), lds, 0, P, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-24u | %14.3f\n", waves, ms * 1e3 / P.rounds);
    }
    return 0;
}
