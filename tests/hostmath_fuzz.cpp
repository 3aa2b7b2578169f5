// TEST INFRASTRUCTURE: vokselis_amd/csrc/vk_hostmath.hpp -- the index arithmetic that decides which pixel lands where on one GPU or eight --
// built by plain g++ with -fsanitize=address,undefined and fuzzed on the CPU (this code had only ever run on a one-GPU
// box through happy paths).  Every buffer below is a std::vector of EXACTLY the size the library allocates, so an index that strays is an
// AddressSanitizer report; every conversion and shift is under UndefinedBehaviorSanitizer.
//
//   blocks:    logical_block and group_logical_block (the XCD-aware relabelling of workgroups, per wave and per 256-thread group) are bijections on
//              whole runs; a group's four waves are the 2 x 2 neighbouring blocks of one tile.
//   deal:      deal_pos / deal_owner / deal_rounds are inverse to each other and dense, for 1..8 ranks, root_skip 0, 2..16, 0..4000 positions.
//   order:     tile_order is a permutation with the active tiles in front, inactive ones in index order, order_pos its inverse; NO pixel whose
//              ray hits the box lies in an inactive tile (rays cast in double, 4 per tile + the tile corners), for cameras around, inside and
//              grazing the cube, frames that are not multiples of the tile, tile sizes 8..1024; vk_tiles_active's decision is the order's.
//   pipeline:  a whole batch replayed with the kernels' own maps: every rank's launch (batch_block_split, deal_pos, tile_pixel,
//              compact_pixel_index) writes pixel ids into its compact buffer [slot_capacity][frames][ts][ts]; the gather concatenates the
//              active prefixes; the un-tile (untile_item, deal_owner, gathered_pixel_index, frame_pixel_index) writes the frames.  Every
//              pixel of every frame must hold its own id (active tile) or the clear value (inactive), nothing is written twice, nothing
//              outside.  1..8 ranks, weighted deals, 1..64 frames with their own cameras, frame_runs on and off.
//
// usage: hostmath_fuzz <cases> <seed>; prints "hostmath_fuzz: OK ..." and exits 0, or the first failures and exits 1.
#include "vk_hostmath.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

using namespace vk;

static uint64_t state;
static uint64_t rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; }
static double unit() { return (double)(rnd() >> 11) / 9007199254740992.0; }
static uint32_t pick(uint32_t lo, uint32_t hi) { return lo + (uint32_t)(rnd() % (uint64_t)(hi - lo + 1)); }

static long g_bad = 0;
#define CHECK(cond, ...)                                                  \
    do {                                                                  \
        if (!(cond)) {                                                    \
            if (g_bad++ < 20) { printf("FAIL %s:%d %s -- ", __FILE__, __LINE__, #cond); printf(__VA_ARGS__); printf("\n"); } \
        }                                                                 \
    } while (0)

// ---- a camera blob like src/camera.rs builds it (float32 storage, double arithmetic is good enough for a fuzz) -----------------------------
struct Cam { float blob[36]; };
static void mat_mul(const double a[16], const double b[16], double o[16]) {  // column-major
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) { double s = 0; for (int k = 0; k < 4; k++) s += a[k * 4 + r] * b[c * 4 + k]; o[c * 4 + r] = s; }
}
static bool mat_inv(const double m[16], double inv[16]) {
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    if (!(fabs(det) > 1e-30)) return false;
    for (int i = 0; i < 16; i++) inv[i] = a[i] / det;
    return true;
}
// Camera::new(zoom, pitch, yaw, target, aspect): eye = target - zoom (sin yaw cos pitch, sin pitch, cos yaw cos pitch); look_at_rh, perspective_rh(pi/2, aspect, .1, 100)
static bool make_camera(double zoom, double pitch, double yaw, const double tgt[3], double aspect, Cam &out) {
    const double pc = cos(pitch);
    const double eye[3] = {tgt[0] - zoom * sin(yaw) * pc, tgt[1] - zoom * sin(pitch), tgt[2] - zoom * cos(yaw) * pc};
    double f[3] = {tgt[0] - eye[0], tgt[1] - eye[1], tgt[2] - eye[2]};
    double fl = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    if (!(fl > 1e-9)) return false;
    for (double &v : f) v /= fl;
    double s[3] = {f[1] * 0 - f[2] * 1, f[2] * 0 - f[0] * 0, f[0] * 1 - f[1] * 0};  // cross(f, up), up = (0, 1, 0)
    double sl = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    if (!(sl > 1e-9)) return false;
    for (double &v : s) v /= sl;
    const double u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    const double V[16] = {s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0,
                          -(s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2]), -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]), f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2], 1};
    const double h = 1.0, w = h / aspect, zn = 0.1, zf = 100.0, r = zf / (zn - zf);
    const double P[16] = {w, 0, 0, 0, 0, h, 0, 0, 0, 0, r, -1, 0, 0, r * zn, 0};
    double PV[16], inv[16];
    mat_mul(P, V, PV);
    if (!mat_inv(PV, inv)) return false;
    out.blob[0] = (float)eye[0]; out.blob[1] = (float)eye[1]; out.blob[2] = (float)eye[2]; out.blob[3] = 1.0f;
    for (int i = 0; i < 16; i++) { out.blob[4 + i] = (float)PV[i]; out.blob[20 + i] = (float)inv[i]; }
    for (float v : out.blob) if (!std::isfinite(v)) return false;
    return true;
}
static bool random_camera(double aspect, int mode, Cam &c) {
    const int kind = (int)(rnd() % 8);
    double tgt[3] = {0.5, 0.5, 0.5};
    if (mode != kModeNaive) tgt[0] = tgt[1] = tgt[2] = 0.0;
    double zoom = 0.3 + unit() * 3.0, pitch = (unit() - 0.5) * 3.0, yaw = unit() * 6.2831853;
    if (kind == 0) { zoom = 1.0; pitch = 0.5; yaw = 1.0; }                       // the bonsai example's own camera
    if (kind == 1) zoom = 0.3 + unit() * 0.3;                                    // eye inside the cube: corners behind the eye plane
    if (kind == 2) { zoom = 20.0 + unit() * 30.0; }                              // far away: the cube is a few pixels
    if (kind == 3) { pitch = 0.0; yaw = (double)(rnd() % 4) * 1.5707963; }       // axis-aligned: an edge-on hull
    if (kind == 4) for (double &t : tgt) t += (unit() - 0.5) * 3.0;              // looking past the cube: partly or wholly off screen
    if (kind == 5) { zoom = 0.87; pitch = 0.6154797; yaw = 0.7853982; }          // along the diagonal: eye near a corner
    return make_camera(zoom, pitch, yaw, tgt, aspect, c);
}

// does the ray through pixel centre (px, py) hit the box?  (NAIVE: [0,1]^3 from the eye; else [-1,1]^3 from the near plane) -- in double
static bool ray_hits(const Cam &c, int mode, uint32_t W, uint32_t H, double px, double py) {
    const float *m = c.blob + 20;
    double e[3], d[3], lo, hi;
    if (mode == kModeNaive) {
        const double X = 2.0 * px / W - 1.0, Y = 1.0 - 2.0 * py / H;
        const double qw = m[3] * X + m[7] * Y + m[11] + m[15];
        for (int k = 0; k < 3; k++) { e[k] = c.blob[k]; d[k] = (m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) / qw - e[k]; }
        lo = 0.0; hi = 1.0;
    } else {
        const double X = 2.0 * px / W - 1.0, Y = (2.0 * py / H - 1.0) * -((double)H / W);
        const double aw = m[3] * X + m[7] * Y + m[15], bw = m[3] * X + m[7] * Y + m[11] + m[15];
        for (int k = 0; k < 3; k++) { e[k] = (m[k] * X + m[4 + k] * Y + m[12 + k]) / aw; d[k] = (m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) / bw - e[k]; }
        lo = -1.0; hi = 1.0;
    }
    double t0 = -1e300, t1 = 1e300;
    for (int k = 0; k < 3; k++) {
        const double inv = 1.0 / d[k], ta = (lo - e[k]) * inv, tb = (hi - e[k]) * inv;
        if (ta != ta || tb != tb) return false;  // (0 * inf: the ray lies in a face plane; not a case to decide here)
        t0 = std::max(t0, std::min(ta, tb)); t1 = std::min(t1, std::max(ta, tb));
    }
    // strictly inside by a margin: a grazing ray may go either way in the kernel's float arithmetic, and either way its pixel is clear-coloured
    return t1 > std::max(t0, 0.0) + 1e-6;
}

// ---- deal -------------------------------------------------------------------------------------------------------------------------------------
static void fuzz_deal() {
    for (uint32_t N = 1; N <= 8; N++)
        for (uint32_t k : {0u, 2u, 3u, 4u, 5u, 8u, 16u}) {
            const uint32_t kk = N > 1 ? k : 0u;
            const uint32_t tiles = pick(0, 4000);
            const uint32_t rounds = deal_rounds(tiles, N, kk);
            std::vector<int> seen((size_t)N * (rounds + 1), 0);
            uint32_t per_rank[8] = {0};
            for (uint32_t pos = 0; pos < tiles; pos++) {
                uint32_t r, s;
                deal_owner(pos, N, kk, r, s);
                CHECK(r < N && s < rounds, "deal_owner(%u, N=%u, k=%u) -> rank %u slot %u of %u rounds", pos, N, kk, r, s, rounds);
                if (r >= N || s > rounds) continue;
                CHECK(deal_pos(r, s, N, kk) == pos, "deal_pos(deal_owner(%u)) = %u (N=%u k=%u)", pos, deal_pos(r, s, N, kk), N, kk);
                CHECK(seen[(size_t)r * (rounds + 1) + s]++ == 0, "two positions in rank %u slot %u (N=%u k=%u)", r, s, N, kk);
                per_rank[r]++;
            }
            // dense: a rank's slots are 0 .. count-1; the root's share is the lightest under a weighted deal; nobody exceeds the rounds
            for (uint32_t r = 0; r < N; r++) {
                for (uint32_t s = 0; s < per_rank[r]; s++) CHECK(seen[(size_t)r * (rounds + 1) + s] == 1, "hole in rank %u at slot %u (N=%u k=%u tiles=%u)", r, s, N, kk, tiles);
                CHECK(per_rank[r] <= rounds, "rank %u holds %u > %u rounds", r, per_rank[r], rounds);
                // a weighted deal: the root sits out every kk-th round (the tiles may run out in the middle of a round: hence the + 1)
                if (kk >= 2 && r > 0) CHECK(per_rank[0] + rounds / kk <= per_rank[r] + 2, "the root's share %u against rank %u's %u over %u rounds (k=%u)", per_rank[0], r, per_rank[r], rounds, kk);
            }
            // deal_rounds is the least number of rounds whose positions (a light round holds one fewer) cover the tiles
            auto capacity = [&](uint32_t r) { return kk >= 2u ? r * N - r / kk : r * N; };
            CHECK(capacity(rounds) >= tiles && (rounds == 0u || capacity(rounds - 1u) < tiles), "deal_rounds(%u, %u, %u) = %u", tiles, N, kk, rounds);
            // positions past the dealt ones never alias a dealt slot of the same rank
            for (uint32_t r = 0; r < N; r++) { const uint32_t p = deal_pos(r, per_rank[r], N, kk); CHECK(p >= tiles, "rank %u's next slot %u maps to dealt position %u", r, per_rank[r], p); }
        }
}

// ---- workgroup -> logical block: bijections on whole runs -----------------------------------------------------------------------------------
static void fuzz_block_maps() {
    const uint32_t runs = pick(1, 6);
    {
        const uint32_t n = runs * 512u;
        std::vector<char> hit(n, 0);
        for (uint32_t b = 0; b < n; b++) {
            const uint32_t lb = logical_block(b);
            CHECK(lb < n && !hit[lb], "logical_block(%u) = %u (of %u)", b, lb, n);
            if (lb < n) hit[lb] = 1;
            // XCD b % 8 receives 64 consecutive logical blocks of the run of 512: one 64 x 64 tile
            CHECK((lb & 511u) >> 6 == (b & 7u), "physical block %u (XCD %u) -> logical %u", b, b & 7u, lb);
        }
    }
    static const uint32_t tss[] = {16, 32, 48, 64, 96, 128};
    const uint32_t ts = tss[rnd() % 6], per_tile = (ts / 8) * (ts / 8);
    // whole runs of 128 groups (512 blocks) that are also whole tiles
    uint32_t n = 512u;
    while (n % per_tile) n += 512u;
    n *= runs;
    std::vector<char> hit(n, 0);
    for (uint32_t G = 0; G < n / 4u; G++) {
        uint32_t lbs[4];
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t lb = lbs[w] = group_logical_block(G, w, ts);
            CHECK(lb < n && !hit[lb], "group_logical_block(%u, %u, ts %u) = %u (of %u)", G, w, ts, lb, n);
            if (lb < n) hit[lb] = 1;
        }
        // the four waves of a group: 2 x 2 neighbouring 8x8 blocks of one tile
        const uint32_t sps = ts / 8, t0 = lbs[0] / per_tile, s0 = lbs[0] % per_tile, x0 = s0 % sps, y0 = s0 / sps;
        CHECK((x0 & 1u) == 0 && (y0 & 1u) == 0, "group %u does not start on an even block (%u, %u)", G, x0, y0);
        for (uint32_t w = 1; w < 4; w++) {
            const uint32_t sw = lbs[w] % per_tile;
            CHECK(lbs[w] / per_tile == t0 && sw % sps == x0 + (w & 1u) && sw / sps == y0 + (w >> 1), "group %u wave %u is not the 2 x 2 neighbour", G, w);
        }
    }
    for (uint32_t i = 0; i < n; i++) CHECK(hit[i], "block %u of %u is reached by no (group, wave) (ts %u)", i, n, ts);
}

// ---- order ------------------------------------------------------------------------------------------------------------------------------------
struct Shape { uint32_t W, H, ts; };
static Shape random_shape(bool small) {
    static const uint32_t tss[] = {8, 16, 24, 32, 40, 64, 128, 256, 512, 1024};
    Shape s;
    s.ts = tss[rnd() % (small ? 6 : 10)];
    s.W = small ? pick(1, 300) : pick(1, 2000);
    s.H = small ? pick(1, 200) : pick(1, 1200);
    if (rnd() % 4 == 0) { s.W = (s.W / s.ts + 1) * s.ts; s.H = (s.H / s.ts + 1) * s.ts; }  // exact multiples too
    return s;
}
static void fuzz_order() {
    const Shape sh = random_shape(false);
    const int mode = (rnd() % 3 == 0) ? 1 : kModeNaive;
    Cam cam;
    if (!random_camera((double)sh.W / sh.H, mode, cam)) return;
    const uint32_t dims[3] = {pick(1, 2048), pick(1, 2048), pick(1, 2048)};
    const uint32_t tx = (sh.W + sh.ts - 1) / sh.ts, ty = (sh.H + sh.ts - 1) / sh.ts;
    const size_t n = (size_t)tx * ty;
    std::vector<uint32_t> order(n, 0xffffffffu), pos(n, 0xffffffffu);
    uint32_t n_active = 0;
    tile_order(sh.W, sh.H, dims, cam.blob, mode, 0, 0, sh.W, sh.H, sh.ts, order.data(), pos.data(), n_active, (int)pick(1, 3));
    CHECK(n_active <= n, "n_active %u of %zu", n_active, n);
    std::vector<char> seen(n, 0);
    for (size_t q = 0; q < n; q++) {
        CHECK(order[q] < n, "order[%zu] = %u of %zu tiles", q, order[q], n);
        if (order[q] >= n) return;
        CHECK(!seen[order[q]], "tile %u twice in the order", order[q]);
        seen[order[q]] = 1;
        CHECK(pos[order[q]] == q, "order_pos is not the inverse at %zu", q);
    }
    for (size_t q = n_active; q + 1 < n; q++) CHECK(order[q] < order[q + 1], "inactive tiles out of index order at %zu", q);
    if (mode != kModeNaive) CHECK(n_active == n, "every tile is active outside NAIVE mode (%u of %zu)", n_active, n);
    // the decision itself, as vk_tiles_active takes it
    int32_t cr[4];
    cull_rect_wh(sh.W, sh.H, cam.blob, mode, cr);
    CHECK(cr[0] >= 0 && cr[1] >= 0 && cr[2] <= (int32_t)sh.W && cr[3] <= (int32_t)sh.H, "cull rectangle outside the frame");
    CullHull hull;
    cull_hull_wh(sh.W, sh.H, cam.blob, mode, hull);
    CHECK(hull.n == 0 || (hull.n >= 3 && hull.n <= 8), "hull of %d points", hull.n);
    for (uint32_t j = 0; j < ty; j++)
        for (uint32_t i = 0; i < tx; i++) {
            const uint32_t tile = j * tx + i;
            const bool inactive = tile_is_inactive(cr, hull, (int64_t)i * sh.ts, (int64_t)j * sh.ts, sh.ts);
            CHECK(inactive == (pos[tile] >= n_active), "tile %u: tile_is_inactive %d, position %u of %u active", tile, (int)inactive, pos[tile], n_active);
            if (!inactive) continue;
            // no ray of an inactive tile may hit the box: its corners' pixels, its centre, and random pixels of it
            for (int s = 0; s < 9; s++) {
                uint32_t lx = s < 4 ? ((s & 1) ? sh.ts - 1 : 0) : (s == 4 ? sh.ts / 2 : pick(0, sh.ts - 1));
                uint32_t ly = s < 4 ? ((s & 2) ? sh.ts - 1 : 0) : (s == 4 ? sh.ts / 2 : pick(0, sh.ts - 1));
                const uint32_t x = i * sh.ts + lx, y = j * sh.ts + ly;
                if (x >= sh.W || y >= sh.H) continue;
                const double off = mode == kModeNaive ? 0.5 : 0.0;  // (the compute mode casts its rays through pixel corners: raycast_compute.wgsl:102)
                CHECK(!ray_hits(cam, mode, sh.W, sh.H, x + off, y + off), "pixel (%u, %u) of INACTIVE tile %u hits the box (%ux%u ts %u)", x, y, tile, sh.W, sh.H, sh.ts);
            }
        }
}

// ---- the whole batch: launch -> gather -> un-tile, with the kernels' own maps ---------------------------------------------------------------
static void fuzz_pipeline() {
    const Shape sh = random_shape(true);
    const uint32_t W = sh.W, H = sh.H, ts = sh.ts;
    const uint32_t nranks = pick(1, 8), n_frames = (rnd() % 3 == 0) ? pick(1, 64) : pick(1, 9);
    static const uint32_t ks[] = {0, 0, 2, 3, 5, 16};
    const uint32_t root_skip = nranks > 1 ? ks[rnd() % 6] : 0u;
    const bool frame_runs = rnd() & 1;
    const uint32_t tx = (W + ts - 1) / ts, ty = (H + ts - 1) / ts, n_tiles = tx * ty;
    const uint32_t dims[3] = {256, 256, 256};
    // the batch's tables, laid out as vk_render_batch lays them out (frame descriptors are 112 bytes: vk_common.hpp)
    const size_t fdesc = 112;
    std::vector<unsigned char> table(batch_table_bytes(n_frames, n_tiles, fdesc), 0);
    uint32_t *order = reinterpret_cast<uint32_t *>(table.data() + batch_order_offset(n_frames, fdesc));
    uint32_t *posn = order + (size_t)n_frames * n_tiles;
    CHECK((unsigned char *)(posn + (size_t)n_frames * n_tiles) == table.data() + table.size(), "table layout does not add up");
    std::vector<uint32_t> n_active(n_frames, 0);
    uint32_t max_active = 0;
    for (uint32_t f = 0; f < n_frames; f++) {
        Cam cam;
        while (!random_camera((double)W / H, kModeNaive, cam)) {}
        tile_order(W, H, dims, cam.blob, kModeNaive, 0, 0, W, H, ts, order + (size_t)f * n_tiles, posn + (size_t)f * n_tiles, n_active[f], 1);
        max_active = std::max(max_active, n_active[f]);
    }
    const uint32_t slots_active = deal_rounds(max_active, nranks, root_skip);
    const uint32_t slot_capacity = slots_active + (uint32_t)(rnd() % 3);  // (the caller's buffer may be larger than the active prefix)
    const uint32_t per_tile = (ts / 8) * (ts / 8);
    const uint64_t n_blocks = (uint64_t)slots_active * n_frames * per_tile;
    if (n_blocks * 64 > 6000000) return;  // (keep a case under a few milliseconds)
    const size_t tt = (size_t)ts * ts;
    // every rank's launch: compact [slot][frame][ts][ts]; a pixel's id encodes (frame, x, y)
    std::vector<std::vector<uint32_t>> compact(nranks, std::vector<uint32_t>((size_t)slot_capacity * n_frames * tt, 0u));
    auto pixel_id = [&](uint32_t f, uint32_t x, uint32_t y) { return 1u + (f * H + y) * W + x; };
    for (uint32_t rank = 0; rank < nranks; rank++)
        for (uint32_t lb = 0; lb < (uint32_t)n_blocks; lb++) {
            const BlockSplit bs = batch_block_split(lb, per_tile, n_frames, frame_runs);
            CHECK(bs.frame < n_frames && bs.slot < slots_active && bs.sub < per_tile, "block %u -> slot %u frame %u sub %u", lb, bs.slot, bs.frame, bs.sub);
            if (bs.frame >= n_frames) continue;
            const uint32_t p = deal_pos(rank, bs.slot, nranks, root_skip);
            const uint32_t tile = p < n_active[bs.frame] ? order[(size_t)bs.frame * n_tiles + p] : n_tiles;  // (map_pixel: past the launch's positions)
            for (uint32_t lane = 0; lane < 64; lane++) {
                const TilePixel tp = tile_pixel(ts, tx, tile, bs.sub, lane);
                const bool valid = tile < n_tiles && tp.rx < W && tp.ry < H;
                if (!valid) continue;
                const size_t idx = compact_pixel_index(compact_record(bs.slot, n_frames, bs.frame), ts, tp.lx, tp.ly);
                CHECK(compact[rank][idx] == 0u, "rank %u writes compact pixel %zu twice", rank, idx);
                compact[rank][idx] = pixel_id(bs.frame, tp.rx, tp.ry);  // (std::vector::operator[] is unchecked: ASan is the bounds check)
            }
        }
    // frame_runs is a relabelling: every (slot, frame, sub) occurs exactly once per rank -- counted through a second pass
    {
        std::vector<char> hit((size_t)slots_active * n_frames * per_tile, 0);
        for (uint32_t lb = 0; lb < (uint32_t)n_blocks; lb++) {
            const BlockSplit bs = batch_block_split(lb, per_tile, n_frames, frame_runs);
            if (bs.frame >= n_frames || bs.slot >= slots_active) continue;
            char &h = hit[((size_t)bs.slot * n_frames + bs.frame) * per_tile + bs.sub];
            CHECK(!h, "(slot %u, frame %u, sub %u) twice in the launch", bs.slot, bs.frame, bs.sub);
            h = 1;
        }
        for (char h : hit) CHECK(h, "a (slot, frame, block) of the launch is never marched");
    }
    // the gather: every rank's active prefix, side by side: [nranks][slots_active][n_frames][ts][ts]
    std::vector<uint32_t> gathered((size_t)nranks * slots_active * n_frames * tt);
    for (uint32_t r = 0; r < nranks; r++) memcpy(gathered.data() + (size_t)r * slots_active * n_frames * tt, compact[r].data(), (size_t)slots_active * n_frames * tt * 4);
    // the un-tile, work item by work item as untile_batch_kernel walks them
    const uint32_t kClear = 0xC1EA5u;
    std::vector<uint32_t> frames((size_t)n_frames * H * W, 0u);
    const uint64_t ublocks = untile_blocks(ts, n_tiles, n_frames);
    for (uint64_t b = 0; b < ublocks; b++)
        for (uint32_t thread = 0; thread < 256; thread++) {
            const UntileItem it = untile_item((uint32_t)b, thread, ts, n_tiles, n_frames);
            if (!it.in_range) continue;
            const uint32_t ly = it.l / ts, lx = it.l - ly * ts;
            const uint32_t tyi = it.tile / tx, txi = it.tile - tyi * tx;
            const uint32_t x = txi * ts + lx, y = tyi * ts + ly;
            if (x >= W || y >= H) continue;
            const bool two = x + 1u < W;
            const size_t dst = frame_pixel_index(it.frame, W, H, x, y);
            const uint32_t p = posn[(size_t)it.frame * n_tiles + it.tile];
            if (p >= n_active[it.frame]) {
                CHECK(frames[dst] == 0u, "pixel written twice (clear)");
                frames[dst] = kClear;
                if (two) { CHECK(frames[dst + 1] == 0u, "pixel written twice (clear)"); frames[dst + 1] = kClear; }
                continue;
            }
            uint32_t rank, slot;
            deal_owner(p, nranks, root_skip, rank, slot);
            CHECK(rank < nranks && slot < slots_active, "position %u -> rank %u slot %u (%u ranks, %u slots)", p, rank, slot, nranks, slots_active);
            if (rank >= nranks || slot >= slots_active) continue;
            const size_t src = gathered_pixel_index(rank, slots_active, slot, n_frames, it.frame, ts, lx, ly);
            CHECK(gathered_record(rank, slots_active, slot, n_frames, it.frame) * tt + (size_t)ly * ts + lx == src, "record and pixel index disagree");
            CHECK(frames[dst] == 0u, "pixel written twice");
            frames[dst] = gathered[src];
            if (two) { CHECK(frames[dst + 1] == 0u, "pixel written twice"); frames[dst + 1] = gathered[src + 1]; }
        }
    for (uint32_t f = 0; f < n_frames; f++)
        for (uint32_t y = 0; y < H; y++)
            for (uint32_t x = 0; x < W; x++) {
                const uint32_t tile = (y / ts) * tx + x / ts;
                const bool active = posn[(size_t)f * n_tiles + tile] < n_active[f];
                const uint32_t got = frames[frame_pixel_index(f, W, H, x, y)], want = active ? pixel_id(f, x, y) : kClear;
                CHECK(got == want, "frame %u pixel (%u, %u): %u, expected %u (%ux%u ts %u, %u ranks, root_skip %u, %u frames, runs %d)", f, x, y, got, want, W, H, ts, nranks,
                      root_skip, n_frames, (int)frame_runs);
            }
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 300;
    state = argc > 2 ? strtoull(argv[2], nullptr, 0) : 88172645463325252ull;
    if (!state) state = 1;
    for (long c = 0; c < cases; c++) {
        fuzz_deal();
        fuzz_block_maps();
        fuzz_order();
        fuzz_order();
        fuzz_pipeline();
        if (g_bad > 20) break;
    }
    if (g_bad) { printf("hostmath_fuzz: %ld FAILURES\n", g_bad); return 1; }
    printf("hostmath_fuzz: OK (%ld cases)\n", cases);
    return 0;
}
