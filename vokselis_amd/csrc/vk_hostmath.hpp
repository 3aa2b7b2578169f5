// vk_hostmath.hpp -- the index arithmetic that decides which pixel lands where: the deal of tile positions over ranks, the block -> pixel
// map of a launch, the compact (slot, frame) records of a partition, the root's un-tile map, and the host-side geometry of a frame (cull
// rectangle, silhouette hull, heaviest-first tile order; the reference's tile loop is examples/xor/main.rs:77-95,235-254).
//
// No HIP in here: the kernels (vk_common.hpp, vk_post.hpp) and the host (vk_order.hip, vk_batch.hip) include it, and so does
// tests/hostmath_fuzz.cpp, a plain g++ -fsanitize=address,undefined program that replays a whole partition -> gather -> un-tile on the CPU
// with these very functions (tile sizes 8..1024, frame shapes that are not multiples of the tile, 1..8 ranks, weighted deals, 1..64 frames).
#pragma once

#include <stdint.h>
#include <stddef.h>

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#if defined(__HIPCC__)
#define VK_HD __host__ __device__ __forceinline__
#else
#define VK_HD inline
#endif

namespace vk {

// ---- dealing positions of the heaviest-first order to ranks ------------------------------------
// Round j gives one position to every rank, in rank order.  With root_skip = k >= 2 the root (rank 0) sits out every
// k-th round (rounds k-1, 2k-1, ...): it also un-tiles every frame, and a lighter share of the march keeps it from
// being the rank everybody waits for.  k < 2: plain round robin, position q -> rank q % N, slot q / N.
VK_HD uint32_t deal_pos(uint32_t rank, uint32_t slot, uint32_t N, uint32_t k) {
    if (k < 2u) return rank + slot * N;
    const uint32_t round = rank ? slot : slot + slot / (k - 1u);  // the root's slot j is its j-th full round
    const uint32_t start = round * N - round / k;                 // one position less for every light round before
    return start + rank - ((round % k) == k - 1u ? 1u : 0u);      // (the root never sees a light round)
}
VK_HD void deal_owner(uint32_t pos, uint32_t N, uint32_t k, uint32_t &rank, uint32_t &slot) {
    if (k < 2u) { rank = pos % N; slot = pos / N; return; }
    const uint32_t G = k * N - 1u, g = pos / G, o = pos - g * G;  // a group: k - 1 full rounds and a light one
    uint32_t rj;
    if (o < (k - 1u) * N) { rj = o / N; rank = o - rj * N; }
    else { rj = k - 1u; rank = o - (k - 1u) * N + 1u; }
    slot = rank ? g * k + rj : g * (k - 1u) + rj;
}
// rounds needed to deal `tiles` positions = slots of a non-root rank
VK_HD uint32_t deal_rounds(uint32_t tiles, uint32_t N, uint32_t k) {
    if (k < 2u) return (tiles + N - 1u) / N;
    uint32_t r = tiles / N;
    while (r * N - r / k < tiles) r++;
    return r;
}

// ---- physical workgroup -> logical 8x8 block ---------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD's L2).  A group of 512 consecutive physical
// blocks is mapped so that each XCD receives 64 consecutive logical blocks = the 64 waves of one 64x64 pixel tile: neighbouring
// rays share an L2, while successive tiles still spread over all XCDs.  A bijection on every multiple of 512 (grids are padded to one).
VK_HD uint32_t logical_block(uint32_t b) {
    const uint32_t group = b >> 9, r = b & 511u;
    return (group << 9) + ((r & 7u) << 6) + (r >> 3);
}
// The staged march's 256-thread groups: workgroup G (round-robin over the XCDs: a run of 128 consecutive groups gives every XCD the 16
// groups of one 64 x 64 tile) and its wave -> the logical 8x8 block; the four waves of a group are the 2 x 2 neighbouring blocks of a tile
// (tile edge a multiple of 16).  A bijection between (G, wave) and the blocks of whole runs of 128 groups.
VK_HD uint32_t group_logical_block(uint32_t G, uint32_t wave, uint32_t ts) {
    const uint32_t chunk = G >> 7, rr = G & 127u;
    const uint32_t LG = (chunk << 7) + ((rr & 7u) << 4) + (rr >> 3);
    const uint32_t sps = ts >> 3, per_tile = sps * sps, hq = sps >> 1;  // 8x8 blocks per tile edge (even), quads per tile edge
    const uint32_t l4 = LG * 4u + wave, u = l4 % per_tile, q = u >> 2, w = u & 3u;
    const uint32_t qy = q / hq, qx = q - qy * hq;
    return (l4 - u) + (2u * qy + (w >> 1)) * sps + 2u * qx + (w & 1u);
}

// ---- a launch's logical block -> (slot, frame, 8x8 block inside the tile) -------------------------
// Batched launches are position-major: (slot, frame) pairs with the frame running fastest, per_tile blocks each.  With
// frame_runs the frame index is relabelled so that the 8 consecutive pairs the launch hands to the 8 XCDs are runs of
// CONSECUTIVE frames of one tile position instead of frames x, x + 8, ... (vk_common.hpp: frame_view).
struct BlockSplit { uint32_t slot, frame, sub; };
VK_HD BlockSplit batch_block_split(uint32_t lb, uint32_t per_tile, uint32_t n_frames, bool frame_runs) {
    BlockSplit s;
    const uint32_t g = lb / per_tile;
    s.sub = lb - g * per_tile;
    s.slot = g / n_frames;
    uint32_t fr = g - s.slot * n_frames;
    if (frame_runs) {
        const uint32_t x = fr & 7u, j = fr >> 3, q = n_frames >> 3, rem = n_frames & 7u;
        fr = x * q + (x < rem ? x : rem) + j;
    }
    s.frame = fr;
    return s;
}

// The pixel a lane of an 8x8 block holds: (lx, ly) inside its tile, (rx, ry) inside the launch's region.
struct TilePixel { uint32_t lx, ly, rx, ry; };
VK_HD TilePixel tile_pixel(uint32_t ts, uint32_t tiles_x, uint32_t tile, uint32_t sub, uint32_t lane) {
    TilePixel p;
    const uint32_t sps = ts >> 3;  // 8x8 blocks per tile edge
    const uint32_t tty = tile / tiles_x, ttx = tile - tty * tiles_x;
    const uint32_t sy = sub / sps, sx = sub - sy * sps;
    p.lx = sx * 8u + (lane & 7u); p.ly = sy * 8u + (lane >> 3);
    p.rx = ttx * ts + p.lx; p.ry = tty * ts + p.ly;
    return p;
}
// compact output of a partition: the (slot, frame) record and the pixel inside it; whole frames: [frame][H][W]
VK_HD uint32_t compact_record(uint32_t slot, uint32_t n_frames, uint32_t frame) { return slot * n_frames + frame; }
VK_HD size_t compact_pixel_index(uint32_t rec, uint32_t ts, uint32_t lx, uint32_t ly) { return (size_t)rec * ((size_t)ts * ts) + (size_t)ly * ts + lx; }
VK_HD size_t frame_pixel_index(uint32_t frame, uint32_t W, uint32_t H, uint32_t x, uint32_t y) { return ((size_t)frame * H + y) * W + x; }
// root side: where pixel (lx, ly) of the tile a rank marched in `slot` of `frame` sits in the gathered [nranks][n_slots][n_frames][ts][ts]
VK_HD size_t gathered_pixel_index(uint32_t rank, uint32_t n_slots, uint32_t slot, uint32_t n_frames, uint32_t frame, uint32_t ts, uint32_t lx, uint32_t ly) {
    return ((((size_t)rank * n_slots + slot) * n_frames + frame) * ts + ly) * ts + lx;
}
// ... and the (slot, frame) record's number in it (VK_WIRE_RGB records are addressed by record, not by pixel)
VK_HD size_t gathered_record(uint32_t rank, uint32_t n_slots, uint32_t slot, uint32_t n_frames, uint32_t frame) { return ((size_t)rank * n_slots + slot) * n_frames + frame; }
// the un-tile kernel's work item: block b of 256 threads moves 512 pixels of one tile of one frame, two adjacent pixels per thread
struct UntileItem { uint32_t frame, tile, l; bool in_range; };
VK_HD UntileItem untile_item(uint32_t b, uint32_t thread, uint32_t ts, uint32_t n_tiles, uint32_t n_frames) {
    UntileItem u;
    const uint32_t chunks = (ts * ts + 511u) / 512u;  // 512-pixel chunks per tile
    const uint32_t chunk = b % chunks; b /= chunks;
    u.tile = b % n_tiles;
    u.frame = b / n_tiles;
    u.l = chunk * 512u + thread * 2u;  // first of this thread's two pixels inside the tile (ts is even)
    u.in_range = u.frame < n_frames && u.l < ts * ts;
    return u;
}
VK_HD uint64_t untile_blocks(uint32_t ts, uint32_t n_tiles, uint32_t n_frames) { return (uint64_t)n_frames * n_tiles * ((ts * ts + 511u) / 512u); }

// what a batch's table block holds, in bytes: FrameDesc[n_frames] | order[n_frames][n_tiles] | pos[n_frames][n_tiles]
VK_HD size_t batch_table_bytes(uint32_t n_frames, size_t n_tiles, size_t frame_desc_bytes) { return (size_t)n_frames * (frame_desc_bytes + 2u * n_tiles * sizeof(uint32_t)); }
VK_HD size_t batch_order_offset(uint32_t n_frames, size_t frame_desc_bytes) { return (size_t)n_frames * frame_desc_bytes; }  // bytes; pos follows at + n_frames * n_tiles words

// ---- host-side geometry of a frame (plain host functions) ---------------------------------------------------------------
constexpr int kModeNaive = 0;  // == VK_MODE_NAIVE_TRILINEAR (include/vokselis_hip.h); the other modes march the [-1, 1]^3 box from the near plane

// Screen-space bounding rectangle of the unit cube (NAIVE mode): the 8 corners projected with
// proj_view in double; any corner at or behind the eye plane disables the cull.  Padded by 2 px.
// Pixels outside [x0,x1) x [y0,y1) cannot hit the box.
inline void cull_rect_wh(uint32_t W, uint32_t H, const float *cam, int mode, int32_t r[4]) {
    r[0] = 0; r[1] = 0; r[2] = (int32_t)W; r[3] = (int32_t)H;
    if (mode != kModeNaive) return;
    const float *pv = cam + 4;
    double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;
    for (int c = 0; c < 8; c++) {
        const double X = c & 1, Y = (c >> 1) & 1, Z = (c >> 2) & 1;
        const double cx = pv[0] * X + pv[4] * Y + pv[8] * Z + pv[12], cy = pv[1] * X + pv[5] * Y + pv[9] * Z + pv[13];
        const double cw = pv[3] * X + pv[7] * Y + pv[11] * Z + pv[15];
        if (!(cw > 1e-6)) return;
        const double sx = (cx / cw * 0.5 + 0.5) * W, sy = (0.5 - cy / cw * 0.5) * H;
        x0 = std::min(x0, sx); x1 = std::max(x1, sx); y0 = std::min(y0, sy); y1 = std::max(y1, sy);
    }
    if (!(std::isfinite(x0) && std::isfinite(x1) && std::isfinite(y0) && std::isfinite(y1))) return;
    // (clamped in double BEFORE the conversion: a corner projected far off screen is a huge double, and int32 of that is undefined)
    r[0] = (int32_t)std::min((double)W, std::max(0.0, std::floor(x0) - 2.0));
    r[1] = (int32_t)std::min((double)H, std::max(0.0, std::floor(y0) - 2.0));
    r[2] = (int32_t)std::max(0.0, std::min((double)W, std::ceil(x1) + 2.0));
    r[3] = (int32_t)std::max(0.0, std::min((double)H, std::ceil(y1) + 2.0));
}

// The cube's silhouette on the screen: the convex hull of its 8 projected corners (counter-clockwise in screen
// coordinates, y down), in double.  A pixel's ray hits the box only if the pixel centre lies inside it, so a tile that a
// hull edge separates from it by more than 2 px holds only clear-colour pixels.  The bounding rectangle alone keeps
// 288 of C2's 510 tiles; the hull keeps the ones a ray can actually hit.  n = 0: no hull (a corner behind the eye
// plane, or another mode) -- the rectangle decides alone.
struct CullHull { int n = 0; double x[16], y[16]; };
inline void cull_hull_wh(uint32_t W, uint32_t H, const float *cam, int mode, CullHull &h) {
    h.n = 0;
    if (mode != kModeNaive) return;
    const float *pv = cam + 4;
    std::pair<double, double> p[8];
    for (int c = 0; c < 8; c++) {
        const double X = c & 1, Y = (c >> 1) & 1, Z = (c >> 2) & 1;
        const double cx = pv[0] * X + pv[4] * Y + pv[8] * Z + pv[12], cy = pv[1] * X + pv[5] * Y + pv[9] * Z + pv[13];
        const double cw = pv[3] * X + pv[7] * Y + pv[11] * Z + pv[15];
        if (!(cw > 1e-6)) return;
        p[c] = {(cx / cw * 0.5 + 0.5) * W, (0.5 - cy / cw * 0.5) * H};
        if (!(std::isfinite(p[c].first) && std::isfinite(p[c].second))) return;
    }
    std::sort(p, p + 8);
    auto cross = [](const std::pair<double, double> &o, const std::pair<double, double> &a, const std::pair<double, double> &b) {
        return (a.first - o.first) * (b.second - o.second) - (a.second - o.second) * (b.first - o.first);
    };
    std::pair<double, double> hull[16];
    int k = 0;
    for (int i = 0; i < 8; i++) { while (k >= 2 && cross(hull[k - 2], hull[k - 1], p[i]) <= 0) k--; hull[k++] = p[i]; }
    for (int i = 6, t = k + 1; i >= 0; i--) { while (k >= t && cross(hull[k - 2], hull[k - 1], p[i]) <= 0) k--; hull[k++] = p[i]; }
    k--;  // (the last point repeats the first)
    if (k < 3) return;  // degenerate (edge-on): the rectangle decides
    h.n = k;
    for (int i = 0; i < k; i++) { h.x[i] = hull[i].first; h.y[i] = hull[i].second; }
}

// true when some hull edge has the whole rectangle [x0,x1] x [y0,y1] more than `pad` pixels on its outer side
inline bool hull_separates(const CullHull &h, double x0, double y0, double x1, double y1, double pad) {
    for (int i = 0; i < h.n; i++) {
        const int j = i + 1 == h.n ? 0 : i + 1;
        const double ex = h.x[j] - h.x[i], ey = h.y[j] - h.y[i];
        const double len = std::sqrt(ex * ex + ey * ey);
        if (!(len > 0)) continue;
        // monotone chain with this cross-product sign walks the hull with its interior on the left: d < 0 is outside
        const double nx = -ey, ny = ex;  // left normal
        const double d0 = nx * (x0 - h.x[i]) + ny * (y0 - h.y[i]), d1 = nx * (x1 - h.x[i]) + ny * (y0 - h.y[i]);
        const double d2 = nx * (x0 - h.x[i]) + ny * (y1 - h.y[i]), d3 = nx * (x1 - h.x[i]) + ny * (y1 - h.y[i]);
        if (std::max(std::max(d0, d1), std::max(d2, d3)) < -pad * len) return true;
    }
    return false;
}

// the tile-level decision, shared by the tile order and vk_tiles_active
inline bool tile_is_inactive(const int32_t cr[4], const CullHull &hull, int64_t x0, int64_t y0, uint32_t ts) {
    return x0 + ts <= cr[0] || x0 >= cr[2] || y0 + ts <= cr[1] || y0 >= cr[3] ||
           (hull.n && hull_separates(hull, (double)x0, (double)y0, (double)(x0 + ts), (double)(y0 + ts), 2.0));
}

// Tiles are dealt to the launch (and, at N > 1, to the ranks) heaviest first.  The frame is ~70 %
// empty and a dense ray ends after 2 steps while a grazing one takes 513, so with ~10 working waves
// per SIMD the kernel's tail is set by whichever heavy tiles start last; starting them first (and
// round-robining them over ranks) shortens it.  The cost estimate is the nominal step count of a
// G x G grid of rays per tile, from the same camera maths as the kernel, in double precision on the
// host.  It is only a launch order: every tile is rendered by the same kernel whatever its rank.
// (W, H): the frame; (ox, oy, rw, rh): the region the launch covers; dims: the volume's.
inline void tile_order(uint32_t W_, uint32_t H_, const uint32_t dims_[3], const float *cam, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                       uint32_t *order, uint32_t *order_pos, uint32_t &order_active, int G) {
    const uint32_t tx = (rw + ts - 1) / ts, ty = (rh + ts - 1) / ts;
    const size_t n = (size_t)tx * ty;
    const double W = W_, H = H_;
    // tiles that do not touch the cube's screen rectangle hold only clear-colour pixels: they sort last (in index
    // order) and are "inactive" -- never marched, never gathered (the root clears them in vk_untile); no rays for them
    int32_t cr[4];
    cull_rect_wh(W_, H_, cam, mode, cr);
    CullHull hull;
    cull_hull_wh(W_, H_, cam, mode, hull);
    struct Key { double cost; uint32_t tile; };
    std::vector<Key> act;
    act.reserve(n);
    const float *m = cam + 20;
    const double dims[3] = {(double)std::max(dims_[0], 1u), (double)std::max(dims_[1], 1u), (double)std::max(dims_[2], 1u)};
    uint32_t n_inactive = 0;
    for (uint32_t j = 0; j < ty; j++)
        for (uint32_t i = 0; i < tx; i++) {
            const int64_t x0 = (int64_t)ox + (int64_t)i * ts, y0 = (int64_t)oy + (int64_t)j * ts;
            const uint32_t tile = j * tx + i;
            if (tile_is_inactive(cr, hull, x0, y0, ts)) { order[n - 1 - n_inactive++] = tile; continue; }  // (reversed below)
            double c = 0.0;
            for (int sy = 0; sy < G; sy++)
                for (int sx = 0; sx < G; sx++) {
                    const double px = (double)x0 + (2 * sx + 1) * ts / (2.0 * G), py = (double)y0 + (2 * sy + 1) * ts / (2.0 * G);
                    if (px < 0 || py < 0 || px >= W || py >= H) continue;
                    double e[3], d[3], lo, hi;
                    if (mode == kModeNaive) {
                        const double X = 2.0 * px / W - 1.0, Y = 1.0 - 2.0 * py / H;
                        const double qw = 1.0 / (m[3] * X + m[7] * Y + m[11] + m[15]);
                        for (int k = 0; k < 3; k++) { e[k] = cam[k]; d[k] = (m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) * qw - e[k]; }
                        lo = 0.0; hi = 1.0;
                    } else {
                        const double X = 2.0 * px / W - 1.0, Y = (2.0 * py / H - 1.0) * -(H / W);
                        const double aw = 1.0 / (m[3] * X + m[7] * Y + m[15]), bw = 1.0 / (m[3] * X + m[7] * Y + m[11] + m[15]);
                        for (int k = 0; k < 3; k++) {
                            e[k] = (m[k] * X + m[4 + k] * Y + m[12 + k]) * aw;
                            d[k] = (m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) * bw - e[k];
                        }
                        lo = -1.0; hi = 1.0;
                    }
                    const double len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                    if (!(len2 > 0)) continue;
                    // steps = (t1 - t0) / dt with t in units of |d| (the normalisation cancels): dt = min_k 1 / (dims_k |d_k|)
                    double t0 = -1e300, t1 = 1e300, inv_dt = 0.0;
                    for (int k = 0; k < 3; k++) {
                        const double inv = 1.0 / d[k], ta = (lo - e[k]) * inv, tb = (hi - e[k]) * inv;
                        t0 = std::max(t0, std::min(ta, tb));
                        t1 = std::min(t1, std::max(ta, tb));
                        inv_dt = std::max(inv_dt, dims[k] * std::fabs(d[k]));
                    }
                    t0 = std::max(t0, 0.0);
                    if (t1 > t0 && inv_dt > 0) c += (t1 - t0) * inv_dt;
                }
            act.push_back({c, tile});
        }
    const uint32_t n_active = (uint32_t)act.size();
    std::stable_sort(act.begin(), act.end(), [](const Key &a, const Key &b) { return a.cost > b.cost; });
    for (uint32_t q = 0; q < n_active; q++) order[q] = act[q].tile;
    std::reverse(order + n_active, order + n);  // inactive tiles in index order
    // Position q goes to XCD q % 8 (rank q % N first, when the frame is partitioned): dealt straight, bin 0
    // would receive the heaviest tile of every round of 8.  Reverse every other round (snake) so the bins'
    // sums even out; the active tiles stay in front.
    for (size_t g = 8; g + 8 <= n_active; g += 16) std::reverse(order + g, order + g + 8);
    order_active = n_active;
    for (size_t q = 0; q < n; q++) order_pos[order[q]] = (uint32_t)q;
}
}  // namespace vk
