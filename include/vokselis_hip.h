/*
 * vokselis_hip.h -- C-ABI of the MI355X-native vokselis raycast path.
 *
 * The reference (pudnax/vokselis) has no FFI: its hot path sits behind the wgpu bind-group
 * contract of shaders/raycast_naive.wgsl and shaders/raycast_compute.wgsl, driven from
 * `Demo::render(&mut self, &Context)` (src/lib.rs:42,178-181).  Each entry point below replaces
 * one wgpu-side step of that contract; the reference lines it replaces are cited per function.
 * INTEGRATION.md shows the `extern "C"` block a Rust maintainer would add to bind it.
 *
 * Conventions: 0 = VK_OK, negative = error, nothing aborts or throws across the boundary.
 * The caller owns every host pointer; the library owns all device memory it allocates.  A
 * vk_ctx is bound to one GPU and one HIP stream (one per frame slot with frames in flight) and is
 * not thread-safe (the reference drives its queue from the single winit thread, src/lib.rs:71).
 */
#ifndef VOKSELIS_HIP_H
#define VOKSELIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_ABI_VERSION 5

typedef struct vk_ctx vk_ctx;

enum vk_status {
    VK_OK = 0,
    VK_ERR_INVALID = -1,   /* bad argument / wrong state */
    VK_ERR_HIP = -2,       /* a HIP runtime call failed; see vk_last_error */
    VK_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    VK_ERR_OOM = -4,
    VK_ERR_UNSUPPORTED = -5
};

/* src/context/volume_texture.rs:39-47 (R8Unorm), SURVEY C4 (R16Float),
 * examples/xor/xor_compute.rs:94-118 (two rgba16float storage textures). */
enum vk_volume_format { VK_FMT_R8_UNORM = 0, VK_FMT_R16_FLOAT = 1, VK_FMT_RGBA16F_PAIR = 2 };

/* raycast_naive.wgsl fs_main vs raycast_compute.wgsl render()/get_col2 */
enum vk_mode {
    VK_MODE_NAIVE_TRILINEAR = 0,
    VK_MODE_COMPUTE_NEAREST = 1,
    /* SURVEY 8(d) C3, "procedural, no volume texture": the reference has no such example (examples/trig draws one
     * triangle); this build defines it as the compute twin's ray and march (raycast_compute.wgsl:62-131) with the
     * texel loads replaced by the xor example's density function noise_volume(p / 2) (shaders/xor.wgsl:55-61,
     * un.time = the uploaded Uniform's time), colour = density.rgb / 2, no normals.  Needs no volume. */
    VK_MODE_PROCEDURAL = 2
};

/* src/context/hdr_backbuffer.rs:10 is rgba16float; RGBA32F is the parity surface. */
enum vk_out_format { VK_OUT_RGBA32F = 0, VK_OUT_RGBA16F = 1 };

/* How the library lays the scalar volume out in HBM (DESIGN.md "Data layout"). */
enum vk_layout {
    VK_LAYOUT_AUTO = 0,
    VK_LAYOUT_LINEAR = 1, /* x-fastest as uploaded; 8 scalar taps per sample (validation kernel) */
    VK_LAYOUT_PACKED = 2, /* 4^3-bricked cells, each holding its 8 trilinear taps (+ per-octant skip maps) */
    VK_LAYOUT_PACKED_PAIRS = 3, /* u8 volumes: cells hold 4 (tap, x-delta) f16 pairs, 16 B */
    VK_LAYOUT_BRICKED = 4, /* dense 8^3 bricks + 1-voxel apron (9^3): 1.42x the dense bytes, 8 taps from one
                              brick as four x-pair loads; the most compact layout (no skip map) */
    VK_LAYOUT_QUADS = 5, /* every element holds a voxel's 2x2 (y,z) neighbourhood: 8 taps = two consecutive elements
                           = ONE 8/16-byte load; 4.5x the dense bytes; pays when the image has more rays than the volume has
                           voxel columns (no skip map; never AUTO's choice) */
    VK_LAYOUT_STAGED = 6 /* dense 8^3 bricks (no apron; one copy per major ray axis) that every wave stages through LDS with
                            coalesced 16-byte LDS-DMA loads, taps read with ds_read: the layout for volumes far larger than
                            the caches (AUTO's choice above 4 GiB of cells; no skip map) */
};

enum vk_render_flags {
    VK_RENDER_NO_SKIP = 1,  /* disable exact empty-space skipping (every step fetches taps) */
    VK_RENDER_COUNT = 2,    /* also accumulate step counters / per-pixel step counts */
    VK_RENDER_SAFE = 4,     /* force the clamped / 64-bit-offset kernel variant */
    VK_RENDER_FORCE_SKIP = 8, /* use the skip kernel (with adaptive probing) whatever the census of transparent cells (default policy:
                                 dense kernel below 45 % of exactly transparent cells, adaptive probing from 45 %, probing on every trip from 55 %) */
    VK_RENDER_DEBUG_TRIPS = 16, /* with COUNT: vk_readback_steps returns march-loop trips (lookups) per pixel */
    VK_RENDER_DEBUG_FALLBACK = 32, /* with COUNT, VK_LAYOUT_STAGED: vk_readback_steps returns the steps whose taps came from global memory */
    VK_RENDER_PROBE_ALWAYS = 64, /* skip kernels: look the distance map up on every trip (no adaptive dense stretches): S_sampled is then
                                   exactly the number of steps that can contribute; the frame is the same either way */
    VK_RENDER_FAST_WALK = 128, /* TOLERANCE MODE of the skip kernels: a run of exactly transparent steps advances the ray in closed form
                                 (one fma per accumulator) instead of by the reference loop's own sequence of rounded additions
                                 (raycast_naive.wgsl:101,118).  t stays exact; the sample positions part from the default, bit-exact mode's
                                 by ~2e-4 cell where a coordinate crosses a power of two inside a skipped run: 99.99 % of C2's pixels
                                 within 1.04e-4, a handful per frame (0.004 %) whose alpha >= 0.95 early-out flips up to 2e-2 off.
                                 19 % faster on C2.  Ignored by kernels that do not skip. */
    /* ABI 5 -- the present pass fused into the raycast pass's epilogue (vk_render only).  When the window has the backbuffer's size,
     * Context::render's present (src/context.rs:251-297, shaders/present.wgsl:111-119) reads each texel once, at its centre: the
     * lane that holds the finished pixel applies ACESFilm + linear_to_srgb itself and writes the context's Rgba8Unorm target (what
     * vk_capture_frame reads) -- no second launch, and the 8-16 bytes per pixel the present pass would read back are never
     * fetched.  A tile-by-tile frame (examples/xor/main.rs:235-254) presents every tile it renders.
     *   VK_RENDER_PRESENT       also write the presented pixel (the target is sized to the backbuffer, as by vk_present(w, h))
     *   VK_RENDER_PRESENT_BGRA  ... and the Bgra8Unorm surface copy (vk_present's also_bgra); implies VK_RENDER_PRESENT
     *   VK_RENDER_PRESENT_ONLY  ... and do NOT store the HDR pixel: the backbuffer keeps what it held; implies VK_RENDER_PRESENT
     * Equal to vk_render followed by vk_present(width, height) on every pixel that pass samples at a texel centre; its f32 pixel
     * coordinates put a few per cent of the samples (52 columns and 43 rows of 1920 x 1080) ~1e-7 of a texel off centre, and there the
     * two may differ by one 8-bit step (a hardware sampler's fixed-point weights read the centre there too). */
    VK_RENDER_PRESENT = 256,
    VK_RENDER_PRESENT_BGRA = 512,
    VK_RENDER_PRESENT_ONLY = 1024,
    /* ABI 5 -- VK_MODE_PROCEDURAL only, a TOLERANCE MODE: hash()'s sine (shaders/xor.wgsl:18-20, `fract(sin(h) * 43758.5453123)`) is the
     * hardware's v_sin_f32 behind a multiply by 1 / 2 pi and a fract -- what a GPU running the shader as written computes -- instead of the
     * specified one (f64 Cody-Waite, shared bit for bit with the oracle).  At the hash's arguments (up to ~8e5) that sine keeps a few correct
     * bits and the hash multiplies the rest by 43758: the frame shows a different noise field with the same statistics (C3 at 1080p: mean
     * colour within 0.3 %, 8 x 8-blurred correlation 0.986, mean |d| 0.017, max 0.17), 2.3x faster (1.51 -> 0.65 ms).  Not comparable pixel
     * by pixel with anything; ignored by the other modes; refused with COUNT. */
    VK_RENDER_DEVICE_SINE = 2048
};

/* ---- context: replaces Context::new device/queue setup, src/context.rs:71-181 ---------- */
int vk_ctx_create(int device_ordinal, vk_ctx **out);
int vk_ctx_destroy(vk_ctx *ctx);
/* Run on a caller-owned HIP stream (e.g. torch's current stream); NULL = the context's own.
 * Stream discipline: every call is asynchronous on the context's stream unless it returns host data.  The context's own
 * stream is non-blocking -- it does NOT synchronise with the legacy default stream -- so device buffers the caller fills
 * (or reads) on another stream need an event / synchronisation of the caller's making, or a shared stream set here. */
int vk_ctx_set_stream(vk_ctx *ctx, void *hip_stream);
int vk_ctx_sync(vk_ctx *ctx);
/* Context::get_info, src/context.rs:183-203 */
int vk_device_info(vk_ctx *ctx, char *name, size_t name_cap, int *compute_units, int *arch_is_gfx950,
                   size_t *total_mem_bytes);
/* Message for the last failing call on this context (ctx == NULL: last vk_ctx_create failure). */
const char *vk_last_error(vk_ctx *ctx);
int vk_abi_version(void);

/* ---- inputs ------------------------------------------------------------------------------ */
/* VolumeTexture::new: create_texture + queue.write_texture, src/context/volume_texture.rs:32-59.
 * `host` is the dense x-fastest array (index x + nx*(y + ny*z)); `host2` only for RGBA16F_PAIR
 * (normals).  The library re-lays it out on the device (layout) and builds the skip maps.
 * RGBA16F_PAIR accepts VK_LAYOUT_LINEAR (two dense arrays) or VK_LAYOUT_PACKED (16-byte density+normals
 * records in 4^3 bricks, <= 1.25 GiB; the AUTO choice up to that size). */
int vk_volume_upload(vk_ctx *ctx, const void *host, const void *host2, uint32_t nx, uint32_t ny, uint32_t nz,
                     int format, int layout);
/* Same, but the dense source already lives in device memory of this GPU (large synthetic volumes). */
int vk_volume_upload_device(vk_ctx *ctx, const void *dev, const void *dev2, uint32_t nx, uint32_t ny,
                            uint32_t nz, int format, int layout);
/* Deterministic synthetic volume generated on the device, integer arithmetic only (bit-identical
 * to the oracle's vo_volume_* and to vokselis_amd/volumes.py).  The reference embeds its volume
 * with include_bytes! (volume_texture.rs:33) and that file is absent from the checkout, so the
 * bench and the tests use these.  FOG: u8 in [lo, lo+span) / f16 bit patterns 0x2D1F + h % 656.
 * BONSAI_STANDIN: pot / trunk / canopy / speckled air, u8.
 * FOG_DENSE_CORE: the fog with a dense ball at the centre (radius: a quarter of the smallest dimension; u8 232 + h % 24,
 * f16 0x3B9A + h % 64), the "dense-core variant" of C4 / C5: rays that reach it leave the loop by the
 * reference's opacity early-out (raycast_naive.wgsl:115-117). */
enum vk_generator { VK_GEN_FOG = 0, VK_GEN_BONSAI_STANDIN = 1, VK_GEN_FOG_DENSE_CORE = 2 };
int vk_volume_generate(vk_ctx *ctx, int kind, uint32_t nx, uint32_t ny, uint32_t nz, int format, uint32_t seed,
                       uint32_t lo, uint32_t span, int layout);
/* XorCompute (examples/xor/xor_compute.rs:93-200): one dispatch of shaders/xor.wgsl `cs_main` at
 * un.time = time (0 in the reference: the pass runs before the first Context::update),
 * filling the two rgba16float storage textures (density, normals) the compute raycast reads. */
int vk_volume_generate_xor(vk_ctx *ctx, uint32_t nx, uint32_t ny, uint32_t nz, float time);
/* Share of the cells that are exactly transparent under the transfer function (packed layouts). */
int vk_volume_empty_fraction(vk_ctx *ctx, double *fraction);
int vk_volume_info(vk_ctx *ctx, uint32_t dims[3], int *format, int *layout, size_t *device_bytes);

/* GlobalUniformBinding::update, src/context/global_ubo.rs:47-49 (48-byte Uniform, :52-65). */
int vk_set_uniform(vk_ctx *ctx, const void *blob48);
/* CameraBinding::update, src/camera.rs:62-71 (144-byte CameraUniform, :5-11).  Always uploads
 * (the reference's `updated` gate leaves the first frame with an identity camera). */
int vk_set_camera(vk_ctx *ctx, const void *blob144);

/* ---- output surface: HdrBackBuffer::new, src/context/hdr_backbuffer.rs:41-87 ------------- */
int vk_backbuffer_resize(vk_ctx *ctx, uint32_t width, uint32_t height, int out_format);
int vk_backbuffer_info(vk_ctx *ctx, uint32_t *width, uint32_t *height, int *out_format, void **device_ptr);
/* Clear to the render pass's LoadOp::Clear(BLACK) = (0,0,0,1), examples/bonsai/main.rs:41. */
int vk_backbuffer_clear(vk_ctx *ctx);

/* ---- frames in flight (ABI 5) -------------------------------------------------------------- */
/* The reference records ONE pass per frame and lets its queue run ahead of the GPU: RedrawRequested calls demo.render and
 * Context::render without waiting for the previous frame (src/lib.rs:178-194); Surface::get_current_texture blocks only
 * when every swapchain image is still in use (src/context.rs:252, PresentMode::Fifo at :118).  A frame that is 70 % empty
 * cannot fill the machine on its own, and a caller that learns its camera one frame at a time cannot use vk_render_batch.
 * With k > 1 the context owns a ring of k frame slots -- backbuffer, present targets and step image of their own, each on
 * its own HIP stream -- and the calls between vk_frame_begin and vk_frame_end go to the slot of that frame, so the first
 * waves of frame i + 1 fill the SIMDs that the tail of frame i leaves idle.  k = 1 (the default) is one slot on the
 * context's stream: the behaviour of ABI 4.  Every frame is bitwise the frame the same calls produce with k = 1.
 *
 * Of a ring of three or four, k - 1 frames execute at once and one waits, recorded, in its queue (its stream waits on the GPU for
 * the frame k - 1 before it): the next frame starts the moment one ends, without the host in between.
 *
 * vk_ctx_frames_in_flight: 1 <= k <= VK_MAX_FRAMES_IN_FLIGHT; drains the context, sizes the ring (new slots take the
 *   backbuffer's current shape, cleared) and forgets earlier frame ids.  Refused while a frame is open, and for k > 1 on a
 *   context that runs on a caller's stream (vk_ctx_set_stream).
 * vk_frame_begin (get_current_texture): takes the next slot of the ring, BLOCKING the host until the frame that used it
 *   k frames ago has completed -- at most k frames are ever in flight; with k = 1 the frames follow one another on one
 *   stream and nothing blocks -- and makes it the context's current surface:
 *   vk_render, vk_backbuffer_clear, vk_present, vk_readback, vk_capture_frame, vk_backbuffer_info and the timers address
 *   it.  *frame_id (>= 1, increasing) names the frame.
 * vk_frame_end (queue.submit + frame.present, src/context.rs:294-296): marks the end of the frame's work.
 * vk_frame_wait: blocks until that frame's work has completed (a frame whose slot has been taken again completed long ago).
 * vk_frame_readback / vk_frame_capture: vk_readback / vk_capture_frame of THAT frame, while its slot still holds it
 *   (VK_ERR_INVALID once a later vk_frame_begin has taken the slot); they wait for the frame only, not for the frames
 *   submitted after it.
 * vk_frame_info: the frame's device buffers (backbuffer; Rgba8 present target or NULL) for consumers on the device -- order
 *   them after vk_frame_wait, and finish with them before the vk_frame_begin that reuses the slot k frames later.
 * Outside begin / end the context behaves as before on its current slot.  vk_ctx_sync drains every slot. */
#define VK_MAX_FRAMES_IN_FLIGHT 4
int vk_ctx_frames_in_flight(vk_ctx *ctx, uint32_t k);
int vk_frame_begin(vk_ctx *ctx, uint64_t *frame_id);
int vk_frame_end(vk_ctx *ctx);
int vk_frame_wait(vk_ctx *ctx, uint64_t frame_id);
int vk_frame_readback(vk_ctx *ctx, uint64_t frame_id, void *dst, size_t row_pitch_bytes);
int vk_frame_capture(vk_ctx *ctx, uint64_t frame_id, void *dst, size_t dst_bytes, uint32_t *out_width, uint32_t *out_height,
                     uint32_t *out_padded_bytes_per_row);
int vk_frame_info(vk_ctx *ctx, uint64_t frame_id, void **backbuffer, void **rgba8, int *complete);

/* ---- the hot path ------------------------------------------------------------------------ */
/* One raycast pass over a tile of the backbuffer, asynchronous on the context's stream.
 * NAIVE: RaycastPipeline::record + queue.submit (examples/bonsai/raycast.rs:118-135,
 * examples/bonsai/main.rs:27-57).  COMPUTE: the `single`/`tile` dispatches with their pixel
 * offset (examples/xor/main.rs:223-254, raycast_compute.wgsl:133-144); pixels of the tile that
 * fall outside the backbuffer are dropped like out-of-range textureStores. */
int vk_render(vk_ctx *ctx, int mode, int32_t tile_x, int32_t tile_y, uint32_t tile_w, uint32_t tile_h,
              float dt_scale, uint32_t flags);

/* Multi-GPU partition of one frame: the backbuffer-sized frame is cut into tile_size^2 tiles
 * (row-major; the reference's TILE_SIZE scheme, examples/xor/main.rs:12,77-95); this call
 * renders the tiles at positions q = rank + j*nranks of the heaviest-first order (vk_partition_order)
 * into `compact_out` (device memory, [n_slots][tile_size][tile_size] pixels of the backbuffer's
 * format, slot j <-> position rank + j*nranks; with vk_partition_root_skip the deal is the weighted one).
 * vk_partition_slots gives n_slots (identical on every rank, for a fixed-size gather). */
int vk_partition_slots(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, uint32_t *n_slots);
/* A lighter share for the root.  Positions are dealt in rounds, one per rank and round; with root_skip = k >= 2 rank 0
 * sits out every k-th round (its share is (k-1)/k of a peer's): rank 0 also receives and un-tiles every frame, and
 * without this the other ranks wait for it.  0 (default): plain round robin.  Set the same value on every rank's
 * context before partitioning; it applies to vk_render_partition, vk_render_batch, vk_untile(_batch) and
 * vk_partition_active.  vk_partition_slots_weighted: the slot count (of a non-root rank) under that deal. */
int vk_partition_root_skip(vk_ctx *ctx, uint32_t root_skip);
int vk_partition_slots_weighted(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, uint32_t root_skip,
                                uint32_t *n_slots);
/* What a pixel of a partition's compact tiles holds.  VK_WIRE_RGBA (default): the backbuffer's pixel.  VK_WIRE_RGB: its
 * three colour channels -- every pixel this path writes has alpha 1 (raycast_naive.wgsl:124, raycast_compute.wgsl:143),
 * and the tiles exist to be moved over xGMI: 6 bytes instead of 8 per rgba16f pixel, a quarter less on the one link
 * each peer has to the root (which is what bounds an 8-GPU node on the 1080p configuration: DESIGN.md 6).  A (slot,
 * frame) record is then ts*ts (r, g) pairs followed by ts*ts b values; the un-tile writes alpha = 1.  Set the same
 * value on every rank before partitioning; it applies to the compact output of vk_render_partition / vk_render_batch,
 * to vk_gather_tiles (n_pixels counts pixels of this size) and to vk_untile / vk_untile_batch.  Frames are bitwise
 * what VK_WIRE_RGBA delivers.  vk_wire_pixel_bytes: bytes per pixel of the context's compact tiles. */
enum { VK_WIRE_RGBA = 0, VK_WIRE_RGB = 1 };
int vk_partition_wire(vk_ctx *ctx, int wire);
int vk_wire_pixel_bytes(vk_ctx *ctx, uint32_t *bytes);
/* The tiles a partition marches and moves ("active"): those the box's projected silhouette -- the convex hull of its 8
 * corners under this camera, 2 px of margin -- can reach; every other tile holds only the clear colour
 * (examples/bonsai/main.rs:41) and is cleared by the root.  Pure host arithmetic, no context: active[] receives
 * tiles_x * tiles_y bytes (row-major, 1 = active).  vk_partition_active reports the same count for a context's camera. */
int vk_tiles_active(const void *camera144, int mode, uint32_t width, uint32_t height, uint32_t tile_size,
                    unsigned char *active, uint32_t *n_active);
int vk_render_partition(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t rank, uint32_t nranks,
                        float dt_scale, uint32_t flags, void *compact_out);
/* The partition deals tiles heaviest-first (a launch/balance heuristic derived from the camera):
 * position q of the order belongs to rank q % nranks, slot q / nranks.  order_out[q] = row-major
 * tile id; identical on every rank for identical camera, volume dims and backbuffer size. */
int vk_partition_order(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t *order_out, uint32_t n_tiles);
/* Only the leading n_active_tiles positions of the order can hold non-clear pixels (tiles touching the
 * cube's screen rectangle); vk_render_partition marches just those, so a rank's compact buffer is
 * meaningful in its first n_active_slots = ceil(n_active_tiles / nranks) slots and only those need
 * to be gathered.  Camera-dependent; identical on every rank. */
int vk_partition_active(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t nranks, uint32_t *n_active_tiles,
                        uint32_t *n_active_slots);
/* Root side: scatter the gathered [nranks][slot_stride][ts][ts] pixels into the backbuffer; tiles beyond
 * the active ones are cleared to (0,0,0,1).  Uses the order of the LAST partition call on this context, so
 * frames marched before a camera change are un-tiled as they were dealt: deliver them before the next
 * partition call under the new camera.  (A stream of frames with changing cameras: vk_render_batch.) */
int vk_untile(vk_ctx *ctx, const void *gathered, uint32_t tile_size, uint32_t nranks, uint32_t slot_stride);
/* Several frames in ONE launch.  The reference keeps frames in flight through its queue (src/lib.rs:178-194 submits a
 * frame per RedrawRequested without waiting); here the grid itself spans n_frames frames, each with its own camera
 * (`cameras`: n_frames x 144-byte CameraUniform blobs, host memory) and its own heaviest-first tile order, dealt
 * position-major so the heaviest tiles of every frame start first.  A frame that is 70 % empty cannot fill 8192 wave
 * slots on its own; a batch can, and a rank holding 1/N of every frame still has n_frames/N frames of work per launch.
 *   compact == 0: `out` receives whole frames (nranks == 1; a rank's share of them with nranks > 1, see below), [n_frames][height][width] pixels of the
 *     backbuffer's format.
 *   compact != 0: `out` receives this rank's tiles, [slot][frame][ts][ts] (slot j <-> position rank + j*nranks of
 *     that frame's order), slot < *n_active_slots <= slot_capacity: a contiguous prefix, ready for one gather.
 * n_frames <= VK_MAX_BATCH_FRAMES.  *batch_id names the batch's tables for vk_untile_batch (the library holds the last 4; a
 * vk_backbuffer_resize or a new volume drops them all).  VK_RENDER_COUNT is refused.
 * Every frame is bitwise equal to the one vk_render produces for the same camera. */
#define VK_MAX_BATCH_FRAMES 1024
/* compact == 0 with nranks > 1 (ABI 4): this rank's tiles written at their place in whole frames [n_frames][height][width] that `out` addresses --
 * typically another GPU's memory, mapped by hipDeviceEnablePeerAccess (vk_group_peer_direct) or HIP IPC: peer-direct tiles, no gather and no
 * un-tile.  Rank 0 also clears the tiles the silhouette cannot reach.  The caller orders the ranks' launches against the consumer. */
int vk_render_batch(vk_ctx *ctx, int mode, uint32_t n_frames, const void *cameras, uint32_t tile_size, uint32_t rank,
                    uint32_t nranks, float dt_scale, uint32_t flags, void *out, int compact, uint32_t slot_capacity,
                    uint32_t *batch_id, uint32_t *n_active_slots);
/* Root side of a batch: gathered [nranks][n_slots][frame][ts][ts] -> out_frames [n_frames][height][width]. */
int vk_untile_batch(vk_ctx *ctx, uint32_t batch_id, const void *gathered, uint32_t n_slots, void *out_frames);
/* The same when out_frames still holds, untouched, what un-tiling `prev_batch_id` (one of the last 4 batches, same frame count
 * and shape) wrote there -- a driver that alternates two frame buffers passes the batch before last: tiles that were inactive
 * then and are inactive now keep their clear colour and are not written again (most of a frame under a still or slowly moving
 * camera: 10.5 of the 22.8 MB an un-tile moves per C2 frame).  prev_batch_id = 0, or a batch that no longer fits: a full
 * un-tile. */
int vk_untile_batch_over(vk_ctx *ctx, uint32_t batch_id, const void *gathered, uint32_t n_slots, void *out_frames, uint32_t prev_batch_id);

/* ---- multi-GPU: the framebuffer's tiles over the node's GPUs, RCCL over xGMI ------------- */
/* Generalises the reference's tile loop (examples/xor/main.rs:235-254: one dispatch per 256x256 tile) to one GPU per
 * share of the tiles.  The volume is replicated (upload it on every context); there is no reduction, only the
 * gather of finished tiles to the root.  RCCL is loaded on first use.
 *
 * (a) one process per GPU (MPI / torchrun style): rank 0 makes an id, the host ships its 128 bytes to the peers,
 *     every rank joins; then per batch vk_render_batch(compact) -> vk_gather_tiles -> (root) vk_untile_batch. */
#define VK_COMM_ID_BYTES 128
/* Can this process load and bind RCCL (librccl.so.1, or what VK_RCCL_LIB names)?  VK_OK, or VK_ERR_UNSUPPORTED with the reason in
 * vk_last_error(NULL).  No side effect beyond loading the library: the probe every rank can make before any of them enters the
 * blocking vk_comm_init_rank.  (vk_comm_unique_id opens a bootstrap listener for the ranks to come: call it on ONE rank.) */
int vk_comm_available(void);
int vk_comm_unique_id(void *id128);
int vk_comm_init_rank(vk_ctx *ctx, const void *id128, int rank, int nranks);
int vk_comm_destroy(vk_ctx *ctx);
/* ABI 5.  Tear the communicator down WITHOUT waiting for what is in flight (ncclCommAbort): for a rank that has found a peer gone -- a
 * gather that does not complete within the caller's time limit (vk_comm_destroy would wait on the stream the dead transfer sits on).
 * Transfers in flight are cancelled, the buffers they were writing are undefined; the context stays usable and may join a new
 * communicator.  No-op without a communicator. */
int vk_comm_abort(vk_ctx *ctx);
int vk_comm_info(vk_ctx *ctx, int *rank, int *nranks);
/* Every rank contributes n_pixels pixels (of the partition's wire format: vk_partition_wire) from `send`; the root receives [nranks][n_pixels] in
 * `recv` (ignored elsewhere).  One grouped send/recv, asynchronous, on `hip_stream` (NULL: the context's stream; a
 * separate stream lets the gather of one batch overlap the march of the next -- the caller orders the two). */
int vk_gather_tiles(vk_ctx *ctx, const void *send, void *recv, size_t n_pixels, int root, void *hip_stream);
/* (b) one process for the node: a group owns one context per GPU (ncclCommInitAll).  Upload the volume and size the
 *     backbuffer on every member (vk_group_ctx), then vk_group_render marches, gathers and un-tiles n_frames frames
 *     into out_frames ([n_frames][height][width], device memory of GPU ordinals[0]).  Asynchronous; vk_group_sync
 *     waits for every member. */
typedef struct vk_group vk_group;
int vk_group_create(int n, const int *ordinals, vk_group **out);
int vk_group_destroy(vk_group *g);
int vk_group_size(vk_group *g);
vk_ctx *vk_group_ctx(vk_group *g, int i);
int vk_group_render(vk_group *g, int mode, uint32_t n_frames, const void *cameras, uint32_t tile_size, float dt_scale,
                    uint32_t flags, void *out_frames);
/* Peer-direct tiles (enable != 0): every member's march stores its pixels straight into out_frames on GPU ordinals[0] over xGMI
 * (hipDeviceEnablePeerAccess) -- no staging buffer, no gather, no un-tile pass on the root, whose fixed 5 us per 1080p frame is what
 * bounds the gathered path at 8 GPUs.  The price: the stores cross the link as they are issued, 8 bytes at a time per lane, instead of
 * in one bulk transfer.  Frames are bitwise the same.  Fails (and leaves the gathered path in place) when a member cannot access
 * the root's memory.  The members write out_frames from their OWN streams (the root's stream waits for them, not they for it): a second
 * vk_group_render into the same out_frames must not be issued before whatever reads the first one's frames on the root has finished
 * (vk_group_sync, or alternate two buffers). */
int vk_group_peer_direct(vk_group *g, int enable);
int vk_group_sync(vk_group *g);
const char *vk_group_last_error(vk_group *g);

/* Device memory for hosts that have no other allocator (frame batches, gather buffers).  vk_device_free and
 * vk_device_download synchronise the context's stream. */
int vk_device_alloc(vk_ctx *ctx, size_t bytes, void **ptr);
int vk_device_free(vk_ctx *ctx, void *ptr);
int vk_device_download(vk_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);

/* ---- present + screenshot ---------------------------------------- */
/* Context::render's present pass (src/context.rs:251-297, shaders/present.wgsl:23-35,111-119): bilinear
 * resample of the backbuffer to width x height (the window size), ACESFilm, linear_to_srgb, into the
 * context-owned Rgba8Unorm target (and a Bgra8Unorm "surface" copy when also_bgra != 0).  Asynchronous. */
int vk_present(vk_ctx *ctx, uint32_t width, uint32_t height, int also_bgra);
/* Context::capture_frame (src/context.rs:299-302, src/context/screenshot.rs:37-77): the presented Rgba8
 * image with the reference's ImageDimentions: even-rounded size, rows padded to 256 B.  dst == NULL only
 * queries the three sizes.  Blocking. */
int vk_capture_frame(vk_ctx *ctx, void *dst, size_t dst_bytes, uint32_t *out_width, uint32_t *out_height,
                     uint32_t *out_padded_bytes_per_row);

/* ---- results ----------------------------------------------------------------------------- */
/* ScreenshotCtx::capture_frame's copy_texture_to_buffer + map, src/context/screenshot.rs:37-77.
 * Blocking.  row_pitch_bytes >= width * bytes_per_pixel. */
int vk_readback(vk_ctx *ctx, void *dst, size_t row_pitch_bytes);
/* Counters of the launches since the last reset (flag VK_RENDER_COUNT): loop iterations the
 * reference would execute (S_ref) and iterations in which taps were fetched (S_sampled). */
int vk_step_counts(vk_ctx *ctx, uint64_t *s_ref, uint64_t *s_sampled);
int vk_step_counts_reset(vk_ctx *ctx);
/* SIMT execution census of the VK_RENDER_COUNT launches since the last reset (NAIVE mode):
 * out[0] wave-level march-loop iterations, out[1] wave-level skipped-step iterations,
 * out[2] wave-level sample executions, out[3] per-lane march-loop iterations (lookups). */
int vk_simt_census(vk_ctx *ctx, uint64_t out[4]);
/* Debug: override the tile order table of the last launch's partition (experiments on launch order): `order` is a permutation of
 * the n tiles (position -> row-major tile id) that keeps the active tiles -- the leading vk_partition_active positions of the
 * current order, the only ones a launch marches -- in front; anything else is VK_ERR_INVALID.  It stays until the partition's
 * key (camera, volume, tile size, rectangle) changes.  The frame does not depend on the order. */
int vk_debug_set_tile_order(vk_ctx *ctx, const uint32_t *order, uint32_t n);
/* Debug: per-8x8-block {first start, last end, HW_ID | XCC_ID << 32, work} of the next VK_RENDER_COUNT NAIVE
 * launch (s_memrealtime stamps).  enable != 0 arms it; out != NULL copies n_blocks quadruples back. */
int vk_debug_wave_trace(vk_ctx *ctx, int enable, uint64_t *out, size_t n_blocks);
/* Debug / tuning knobs of the staged march: "stage_cap_bytes" (LDS window per wave; 0 = default: 8192 for u8, 10240 for f16),
 * "stage_slab_cells" (a round is a slab of at most that many cells along the wave's major axis, default 8), "stage_copies_mask" (bit k: build the brick copy
 * whose slow axis is k; applies to the next upload, default 7). */
int vk_debug_set_param(vk_ctx *ctx, const char *name, double value);
/* Per-pixel executed loop iterations of the last VK_RENDER_COUNT launch ([height][width] u32). */
int vk_readback_steps(vk_ctx *ctx, uint32_t *dst);

/* hipEvent bracket on the context's stream -- the analogue of the reference's timestamp
 * queries around the raycast pass (examples/xor/main.rs:217,258-259,164-187). */
int vk_timer_begin(vk_ctx *ctx);
int vk_timer_end(vk_ctx *ctx);
int vk_timer_elapsed_ms(vk_ctx *ctx, float *ms); /* blocks until the end event has passed */

/* dispatch_optimal, src/utils/mod.rs:15-18 */
uint32_t vk_dispatch_optimal(uint32_t len, uint32_t subgroup_size);

#ifdef __cplusplus
}
#endif
#endif /* VOKSELIS_HIP_H */
