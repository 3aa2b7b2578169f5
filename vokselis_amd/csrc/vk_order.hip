// vk_order.hip -- host-side geometry of a frame: the cube's screen rectangle and silhouette, the heaviest-first tile order
// (launch order at N = 1, the deal over ranks at N > 1; the reference's tile loop is examples/xor/main.rs:77-95,235-254) and the
// device ring the order tables travel through.  vk_partition_* / vk_tiles_active.
#include "vk_ctx.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace vk;

// Screen-space bounding rectangle of the unit cube (NAIVE mode): the 8 corners projected with
// proj_view in double; any corner at or behind the eye plane disables the cull.  Padded by 2 px.
// Pixels outside [x0,x1) x [y0,y1) cannot hit the box.
void cull_rect_wh(uint32_t W, uint32_t H, const float *cam, int mode, int32_t r[4]) {
    r[0] = 0; r[1] = 0; r[2] = (int32_t)W; r[3] = (int32_t)H;
    if (mode != VK_MODE_NAIVE_TRILINEAR) return;
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
    r[0] = (int32_t)std::max(0.0, std::floor(x0) - 2.0);
    r[1] = (int32_t)std::max(0.0, std::floor(y0) - 2.0);
    r[2] = (int32_t)std::min((double)W, std::ceil(x1) + 2.0);
    r[3] = (int32_t)std::min((double)H, std::ceil(y1) + 2.0);
}
void cull_rect_cam(const vk_ctx *ctx, const float *cam, int mode, int32_t r[4]) { cull_rect_wh(ctx->width, ctx->height, cam, mode, r); }


// The cube's silhouette on the screen: the convex hull of its 8 projected corners (counter-clockwise in screen
// coordinates, y down), in double.  A pixel's ray hits the box only if the pixel centre lies inside it, so a tile that a
// hull edge separates from it by more than 2 px holds only clear-colour pixels.  The bounding rectangle alone keeps
// 288 of C2's 510 tiles; the hull keeps the ones a ray can actually hit.  n = 0: no hull (a corner behind the eye
// plane, or another mode) -- the rectangle decides alone.
struct CullHull { int n = 0; double x[16], y[16]; };
static void cull_hull_wh(uint32_t W, uint32_t H, const float *cam, int mode, CullHull &h) {
    h.n = 0;
    if (mode != VK_MODE_NAIVE_TRILINEAR) return;
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
// (forward) the tile-level decision, shared by the tile order and vk_tiles_active
static bool tile_is_inactive(const int32_t cr[4], const CullHull &hull, int64_t x0, int64_t y0, uint32_t ts);

// true when some hull edge has the whole rectangle [x0,x1] x [y0,y1] more than `pad` pixels on its outer side
static bool hull_separates(const CullHull &h, double x0, double y0, double x1, double y1, double pad) {
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

static bool tile_is_inactive(const int32_t cr[4], const CullHull &hull, int64_t x0, int64_t y0, uint32_t ts) {
    return x0 + ts <= cr[0] || x0 >= cr[2] || y0 + ts <= cr[1] || y0 >= cr[3] ||
           (hull.n && hull_separates(hull, (double)x0, (double)y0, (double)(x0 + ts), (double)(y0 + ts), 2.0));
}

// Tiles are dealt to the launch (and, at N > 1, to the ranks) heaviest first.  The frame is ~70 %
// empty and a dense ray ends after 2 steps while a grazing one takes 513, so with ~10 working waves
// per SIMD the kernel's tail is set by whichever heavy tiles start last; starting them first (and
// round-robining them over ranks) shortens it.  The cost estimate is the nominal step count of a
// 3x3 grid of rays per tile, from the same camera maths as the kernel, in double precision on the
// host.  It is only a launch order: every tile is rendered by the same kernel whatever its rank.
void compute_tile_order_raw(const vk_ctx *ctx, const float *cam, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                                   uint32_t *order, uint32_t *order_pos, uint32_t &order_active, int G) {
    const uint32_t tx = (rw + ts - 1) / ts, ty = (rh + ts - 1) / ts;
    const size_t n = (size_t)tx * ty;
    const double W = ctx->width, H = ctx->height;
    // tiles that do not touch the cube's screen rectangle hold only clear-colour pixels: they sort last (in index
    // order) and are "inactive" -- never marched, never gathered (the root clears them in vk_untile); no rays for them
    int32_t cr[4];
    cull_rect_cam(ctx, cam, mode, cr);
    CullHull hull;
    cull_hull_wh(ctx->width, ctx->height, cam, mode, hull);
    struct Key { double cost; uint32_t tile; };
    std::vector<Key> act;
    act.reserve(n);
    const float *m = cam + 20;
    const double dims[3] = {(double)std::max(ctx->nx, 1u), (double)std::max(ctx->ny, 1u), (double)std::max(ctx->nz, 1u)};
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
                    if (mode == VK_MODE_NAIVE_TRILINEAR) {
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

static void compute_tile_order(const vk_ctx *ctx, const float *cam, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                               std::vector<uint32_t> &order, std::vector<uint32_t> &order_pos, uint32_t &order_active) {
    const size_t n = (size_t)((rw + ts - 1) / ts) * ((rh + ts - 1) / ts);
    order.resize(n); order_pos.resize(n);
    compute_tile_order_raw(ctx, cam, mode, ox, oy, rw, rh, ts, order.data(), order_pos.data(), order_active, (int)ctx->order_rays);
}

// Every caller of tile_order_update goes on to launch a reader of the current slot on ctx->stream: who that is, for the day
// the ring comes round to the slot again.
static void order_note_use(vk_ctx *ctx) {
    if (ctx->ring_slot < 0) return;
    ctx->ring_use_stream[ctx->ring_slot] = ctx->stream;
    ctx->ring_use_frame[ctx->ring_slot] = ctx->fif_open ? ctx->fif[ctx->fif_cur].id : 0;
}

int tile_order_update(vk_ctx *ctx, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts) {
    const uint32_t tx = (rw + ts - 1) / ts, ty = (rh + ts - 1) / ts;
    const size_t n = (size_t)tx * ty;
    std::vector<unsigned char> key(144 + 40);
    std::memcpy(key.data(), ctx->camera, 144);
    const uint32_t kk[10] = {(uint32_t)mode, (uint32_t)ox, (uint32_t)oy, rw, rh, ts, ctx->width, ctx->height, ctx->nx ^ (ctx->ny << 10) ^ (ctx->nz << 20), 0};
    std::memcpy(key.data() + 144, kk, 40);
    if (key == ctx->order_key && ctx->order.size() == n) { order_note_use(ctx); return VK_OK; }
    compute_tile_order(ctx, ctx->camera, mode, ox, oy, rw, rh, ts, ctx->order, ctx->order_pos, ctx->order_active);
    const uint32_t n_active = ctx->order_active;
    constexpr int kOrderRing = 16;
    if (ctx->d_order_cap < n) {
        HIP_TRY(ctx, hipDeviceSynchronize());  // a larger frame shape: rebuild the ring (rare)
        if (ctx->d_ring) (void)hipFree(ctx->d_ring);
        if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
        ctx->d_ring = ctx->h_ring = nullptr; ctx->d_order = ctx->d_order_pos = nullptr;
        ctx->d_order_cap = 0; ctx->ring_slot = -1;
        HIP_TRY(ctx, hipMalloc(&ctx->d_ring, (size_t)kOrderRing * 2 * n * sizeof(uint32_t)));
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_ring, (size_t)kOrderRing * 2 * n * sizeof(uint32_t)));
        ctx->d_order_cap = n;
    }
    const size_t cap = ctx->d_order_cap;
    const int slot = (int)(++ctx->order_seq % (uint32_t)kOrderRing);
    // Waiting on the slot's previous upload (kOrderRing cameras ago) makes the pinned staging safe to rewrite.  The
    // device slot itself is safe to overwrite: on ONE stream the kernels that read the slot kOrderRing cameras ago were
    // enqueued before this copy (vk_ctx_set_stream drains the old stream first), so the copy is stream-ordered after them;
    // with frames in flight (several streams) the slot's last reader may belong to a frame on ANOTHER stream -- a frame that
    // draws its tiles one launch at a time (the xor example's 3 x 6 tile loop) goes round the ring inside every frame: this
    // stream then waits for that frame's end (nothing to wait for when vk_frame_begin has since retaken its surface: it
    // waited on the host then).
    if (ctx->ring_ev[slot]) HIP_TRY(ctx, hipEventSynchronize(ctx->ring_ev[slot]));
    else HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ring_ev[slot], hipEventDisableTiming));
    if (ctx->fif_k > 1 && ctx->ring_use_stream[slot] && ctx->ring_use_stream[slot] != ctx->stream) {
        bool ordered = false;
        const uint64_t f = ctx->ring_use_frame[slot];
        if (f != 0) {
            const vk_ctx::FrameSlot *fs = nullptr;
            for (uint32_t i = 0; i < ctx->fif_k; i++) if (ctx->fif[i].id == f) fs = &ctx->fif[i];
            if (!fs) ordered = true;
            else if (fs->ended) {
                if (hipEventQuery(fs->done) != hipSuccess) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, fs->done, 0));
                ordered = true;
            }
        }
        if (!ordered) {  // a reader outside any frame: order this stream after everything its stream has been given so far
            if (!ctx->ring_guard_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ring_guard_ev, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(ctx->ring_guard_ev, ctx->ring_use_stream[slot]));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_guard_ev, 0));
        }
    }
    uint32_t *hs = ctx->h_ring + (size_t)slot * 2 * cap, *ds = ctx->d_ring + (size_t)slot * 2 * cap;
    std::memcpy(hs, ctx->order.data(), n * sizeof(uint32_t));
    std::memcpy(hs + cap, ctx->order_pos.data(), n * sizeof(uint32_t));
    HIP_TRY(ctx, hipMemcpyAsync(ds, hs, (cap + n) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ring_ev[slot], ctx->stream));
    ctx->ring_stream[slot] = ctx->stream;
    ctx->ring_done[slot] = false;
    ctx->ring_active[slot] = n_active;
    ctx->ring_slot = slot;
    ctx->d_order = ds;
    ctx->d_order_pos = ds + cap;
    ctx->order_key = key;
    order_note_use(ctx);
    return VK_OK;
}

// A launch on a stream other than the one that uploaded the current order slot waits for that upload.
int order_wait(vk_ctx *ctx) {
    const int s = ctx->ring_slot;
    if (s < 0 || ctx->ring_done[s] || ctx->stream == ctx->ring_stream[s]) return VK_OK;
    if (hipEventQuery(ctx->ring_ev[s]) == hipSuccess) { ctx->ring_done[s] = true; return VK_OK; }
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_ev[s], 0));
    return VK_OK;
}

extern "C" {

int vk_partition_wire(vk_ctx *ctx, int wire) {
    if (!ctx) return VK_ERR_INVALID;
    if (wire != VK_WIRE_RGBA && wire != VK_WIRE_RGB) return fail(ctx, VK_ERR_INVALID, "vk_partition_wire: VK_WIRE_RGBA or VK_WIRE_RGB");
    ctx->wire = wire;
    return VK_OK;
}

int vk_wire_pixel_bytes(vk_ctx *ctx, uint32_t *bytes) {
    if (!ctx || !bytes) return fail(ctx, VK_ERR_INVALID, "vk_wire_pixel_bytes: NULL argument");
    *bytes = (uint32_t)wire_px_bytes(ctx->out_format, ctx->wire);
    return VK_OK;
}

int vk_partition_slots(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, uint32_t *n_slots) {
    return vk_partition_slots_weighted(width, height, tile_size, nranks, 0, n_slots);
}

int vk_partition_slots_weighted(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, uint32_t root_skip, uint32_t *n_slots) {
    if (!n_slots || tile_size == 0 || (tile_size & 7u) || nranks == 0 || width == 0 || height == 0 || root_skip == 1) return VK_ERR_INVALID;
    uint64_t tiles = (uint64_t)((width + tile_size - 1) / tile_size) * ((height + tile_size - 1) / tile_size);
    *n_slots = deal_rounds((uint32_t)tiles, nranks, nranks > 1 ? root_skip : 0u);
    return VK_OK;
}

// Which tiles of a width x height frame can hold a pixel whose ray hits the volume's box under this camera: the decision
// every partition makes (inactive tiles are never marched nor gathered; the root clears them).  Pure host arithmetic, no
// context: active[tile] (row-major, tiles_x * tiles_y bytes) is 1 or 0.
int vk_tiles_active(const void *camera144, int mode, uint32_t width, uint32_t height, uint32_t tile_size, unsigned char *active, uint32_t *n_active) {
    if (!camera144 || !active || tile_size == 0 || (tile_size & 7u) || width == 0 || height == 0) return VK_ERR_INVALID;
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return VK_ERR_INVALID;
    float cam[36];
    std::memcpy(cam, camera144, 144);
    for (float v : cam) if (!std::isfinite(v)) return VK_ERR_INVALID;
    const int geo_mode = mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode;
    int32_t cr[4];
    cull_rect_wh(width, height, cam, geo_mode, cr);
    CullHull hull;
    cull_hull_wh(width, height, cam, geo_mode, hull);
    const uint32_t tx = (width + tile_size - 1) / tile_size, ty = (height + tile_size - 1) / tile_size;
    uint32_t n = 0;
    for (uint32_t j = 0; j < ty; j++)
        for (uint32_t i = 0; i < tx; i++) {
            const bool on = !tile_is_inactive(cr, hull, (int64_t)i * tile_size, (int64_t)j * tile_size, tile_size);
            active[(size_t)j * tx + i] = on ? 1 : 0;
            n += on;
        }
    if (n_active) *n_active = n;
    return VK_OK;
}

int vk_partition_root_skip(vk_ctx *ctx, uint32_t root_skip) {
    if (!ctx) return VK_ERR_INVALID;
    if (root_skip == 1) return fail(ctx, VK_ERR_INVALID, "vk_partition_root_skip: 0 (never) or >= 2 (rank 0 sits out every k-th round)");
    ctx->root_skip = root_skip;
    return VK_OK;
}

int vk_partition_active(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t nranks, uint32_t *n_active_tiles, uint32_t *n_active_slots) {
    if (!ctx) return VK_ERR_INVALID;
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return fail(ctx, VK_ERR_INVALID, "unknown mode");
    if (!ctx->backbuffer || !ctx->have_camera || (ctx->format < 0 && mode != VK_MODE_PROCEDURAL))
        return fail(ctx, VK_ERR_INVALID, "vk_partition_active: needs camera and backbuffer (and a volume, except PROCEDURAL)");
    if (tile_size == 0 || (tile_size & 7u) || nranks == 0) return fail(ctx, VK_ERR_INVALID, "bad tile size / nranks");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the same geometry key as the render calls use (PROCEDURAL marches the compute twin's rays)
    int orc = tile_order_update(ctx, mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode, 0, 0, ctx->width, ctx->height, tile_size);
    if (orc) return orc;
    if (n_active_tiles) *n_active_tiles = ctx->order_active;
    if (n_active_slots) *n_active_slots = deal_rounds(ctx->order_active, nranks, nranks > 1 ? ctx->root_skip : 0u);
    return VK_OK;
}

int vk_debug_set_tile_order(vk_ctx *ctx, const uint32_t *order, uint32_t n) {
    // experiment hook: replace the current (already computed) order table; stays until the key changes
    if (!ctx || !order) return VK_ERR_INVALID;
    if (n != ctx->order.size() || !ctx->d_order) return fail(ctx, VK_ERR_INVALID, "vk_debug_set_tile_order: no order of that size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipDeviceSynchronize());
    for (uint32_t q = 0; q < n; q++) { ctx->order[q] = order[q]; ctx->order_pos[order[q]] = q; }
    HIP_TRY(ctx, hipMemcpy(ctx->d_order, ctx->order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_order_pos, ctx->order_pos.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    return VK_OK;
}

int vk_partition_order(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t *order_out, uint32_t n_tiles) {
    if (!ctx || !order_out) return fail(ctx, VK_ERR_INVALID, "vk_partition_order: NULL argument");
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return fail(ctx, VK_ERR_INVALID, "unknown mode");
    if (!ctx->backbuffer || !ctx->have_camera || (ctx->format < 0 && mode != VK_MODE_PROCEDURAL))
        return fail(ctx, VK_ERR_INVALID, "vk_partition_order: needs camera and backbuffer (and a volume, except PROCEDURAL)");
    if (tile_size == 0 || (tile_size & 7u)) return fail(ctx, VK_ERR_INVALID, "tile size must be a multiple of 8");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int orc = tile_order_update(ctx, mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode, 0, 0, ctx->width, ctx->height, tile_size);
    if (orc) return orc;
    if (n_tiles != ctx->order.size()) return fail(ctx, VK_ERR_INVALID, "vk_partition_order: n_tiles does not match the partition");
    std::memcpy(order_out, ctx->order.data(), n_tiles * sizeof(uint32_t));
    return VK_OK;
}

}  // extern "C"
