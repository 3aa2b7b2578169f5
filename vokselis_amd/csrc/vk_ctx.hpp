// vk_ctx.hpp -- the context behind the C-ABI (include/vokselis_hip.h) and what the library's translation units share.
// Internal: nothing here is part of the boundary.
//
//   vk_context.hip         context lifetime, uniforms, backbuffer, read-back, counters, timers, tuning knobs
//   vk_volume.hip          vk_volume_*: re-layout of the uploaded / generated volume (VolBuild, transactional commit)
//   vk_order.hip           screen-space cull, heaviest-first tile order, vk_partition_* / vk_tiles_active
//   vk_render.hip          vk_render / vk_render_partition: argument checks, LaunchDesc, kernel choice
//   vk_batch.hip           vk_render_batch: many frames, one launch
//   vk_launch_cells.hip    instantiates the cell-layout march kernels      (vk_march.hpp)
//   vk_launch_staged.hip   instantiates the LDS-staged march kernels       (vk_staged.hpp)
//   vk_launch_compute.hip  instantiates the compute twin and C3            (vk_compute.hpp)
//   vk_post.hip            clear, un-tile, present, capture                (vk_post.hpp)
//   vk_comm.hip            RCCL: vk_comm_*, vk_gather_tiles, vk_group_*
#pragma once

#include "../../include/vokselis_hip.h"
#include "vk_common.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is loaded on first use (vk_comm.hip)

#include <string>
#include <utility>
#include <vector>

struct vk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop{};

    // volume
    void *vol = nullptr, *vol2 = nullptr;  // cells / dense voxels / pair
    void *scopy[3] = {nullptr, nullptr, nullptr};  // VK_LAYOUT_STAGED: one brick copy per slow axis
    vk::StagedDesc sdesc{};
    uint32_t stage_cap_bytes = 0, stage_slab_cells = 0, stage_copies_mask = 7;  // tunables (vk_debug_set_param)
    uint32_t stage_group = 2;    // staged march: one LDS window per 256-thread group of four waves (vk_staged.hpp: raymarch_staged_group_kernel): 0 never, 1 always, 2 where it pays (launch_staged)
    uint32_t frame_runs = 1;     // batched launches: every XCD marches a run of consecutive frames of a tile position (0: frames x, x + 8, ... as in round 2)
    uint32_t stage_grow_every = 0;  // slab search: try one cell above the last fit every n-th round (0: 4 for u8, 1 for f16; docs/archive/tools/staged_grow.py)
    uint32_t stage_row_pad = 2;  // odd LDS row pitch of the staged window: 0 never, 1 always, 2 (default) with group windows on u8 volumes (launch_staged)
    uint8_t *dist = nullptr;
    uint32_t *lut = nullptr;  // per-axis cell-index tables (cell units | byte offsets), cell layouts only
    size_t vol_bytes = 0;
    uint32_t nx = 0, ny = 0, nz = 0, nbx = 0, nby = 0, nbz = 0;
    int format = -1, layout = 0;
    int vol_kind = -1;  // vk::VolKind
    double empty_fraction = 0.0;  // share of cells that are exactly transparent (packed layouts)
    vk::VolumeDesc vdesc{};

    // uniforms (host copies; passed to kernels by value)
    unsigned char uniform[48] = {0};
    float camera[36] = {0};  // 144-byte CameraUniform
    bool have_camera = false;

    // output
    void *backbuffer = nullptr;
    uint32_t width = 0, height = 0;
    int out_format = VK_OUT_RGBA32F;
    uint32_t *steps = nullptr;
    unsigned long long *counters = nullptr;

    // heaviest-first tile order (launch-order heuristic; see tile_order_update)
    std::vector<uint32_t> order, order_pos;
    uint32_t order_active = 0;  // leading positions of `order` whose tiles can contain non-clear pixels
    bool order_on_device = false;  // the current slot of the device ring holds `order` / `order_pos` (single-frame launches of few tiles carry the order in their kernel arguments)
    std::vector<unsigned char> order_key;
    unsigned long long *trace = nullptr;
    size_t trace_blocks = 0;
    int wire = VK_WIRE_RGBA;     // compact tiles of a partition: whole pixels, or colour only (vk_partition_wire)
    bool want_trace = false;
    uint32_t trip_log_cap = 0;   // > 0: the trace buffer holds per-trip logs of that many u32 entries per wave instead of stamps (docs/archive/tools/repack_census.py)
    // The device copies of (order, order_pos) live in a ring of kOrderRing slots fed from pinned staging: a new
    // camera takes the next slot with one stream-ordered copy -- no host or device synchronisation -- while
    // launches still in flight (other streams: frames in flight) keep reading the slots they were given.
    uint32_t *d_order = nullptr, *d_order_pos = nullptr;  // the current slot
    uint32_t *d_ring = nullptr, *h_ring = nullptr;
    size_t d_order_cap = 0;      // entries per table in every slot
    int ring_slot = -1;
    hipEvent_t ring_ev[16] = {};     // slot uploaded
    hipStream_t ring_stream[16] = {};
    bool ring_done[16] = {};
    uint32_t ring_active[16] = {};   // order_active of the slot
    uint32_t order_seq = 0;          // order changes so far; change e lives in slot e % 16
    hipStream_t ring_use_stream[16] = {};  // frames in flight: the stream of the slot's last reader ...
    uint64_t ring_use_frame[16] = {};      // ... and the frame it belonged to (0: launched outside vk_frame_begin / vk_frame_end)
    hipEvent_t ring_guard_ev = nullptr;

    // batched launches (vk_render_batch): per-batch tables {FrameDesc[B], order[B][n_tiles], pos[B][n_tiles]} in a small
    // ring of device slots fed from pinned staging; a slot is rewritten only after the last kernel that read it
    struct BatchSlot {
        unsigned char *d = nullptr, *h = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        uint32_t id = 0, n_frames = 0, n_tiles = 0, ts = 0, nranks = 0, max_active = 0, root_skip = 0;
        uint32_t width = 0, height = 0;  // the frame shape and pixel format the batch was dealt for: an un-tile under another
        int out_format = -1;             // shape (vk_backbuffer_resize in between) would scatter tiles out of bounds
        int wire = 0;                    // ... and the wire format its compact tiles were written in
    } batch[4];
    uint32_t batch_seq = 0;
    // table blocks a growing batch has outgrown: hipFree / hipHostFree synchronise the device, so they wait here for a
    // call that synchronises anyway (vk_backbuffer_resize, vk_ctx_destroy) instead of stalling four launches in flight
    std::vector<std::pair<void *, void *>> batch_retired;
    // the order of the last camera a batch computed one for (a still camera costs no host work from batch to batch)
    std::vector<unsigned char> batch_key;
    std::vector<uint32_t> batch_order, batch_pos;
    uint32_t batch_n_active = 0;
    // Skip kernels: steps a walk may take in a trip in which other lanes sample / in which every lane walks (0: no cap).
    // docs/archive/tools/walk_cap_sweep.py: 8 / 12 -- C2 0.0806 -> 0.0728 ms per frame in batches, 0.1625 -> 0.1555 single; a fog with
    // 80 % of its 16^3 blocks knocked out 0.161 -> 0.135.  Multiples of the walk loop's four steps do best.
    uint32_t walk_cap = 8, walk_cap_all = 12;
    uint32_t render_tile = 0;    // vk_render: tile size of the launch order (0: 64, 32 for the staged march)
    uint32_t pair_ring = 0;      // compute twin: request buffers in the SHADE ring (4, 6; 42: 4 buffers, two revolutions per loop iteration; 0: by launch shape)
    uint32_t pair_walk_min = 4;  // compute twin: shortest run of empty records worth a walk (a walk restarts the request ring; docs/archive/tools/compute_mode.py)
    uint32_t probe_ahead = 2;    // skip kernels: request the next position's distance byte under the sample (0 never, 1 always, 2 single-frame launches)
    uint32_t order_rays = 3;     // estimate rays per tile edge of the heaviest-first order (single-frame launches)
    uint32_t order_rays_batch = 1;  // ... of launches spanning >= 4 frames
    uint32_t order_never_inline = 0;  // debug / A-B: single-frame launches read the tile order from the device table as before round 6
    uint32_t wave_prio = 1;      // issue priority by ray length (set_wave_priority); 0 for A/B measurements
    uint32_t naive_lds_pad = 0;  // debug: extra dynamic LDS per workgroup of the cell kernels (caps the waves per SIMD)
    uint32_t root_skip = 0;  // dealing: rank 0 sits out every root_skip-th round (vk_partition_root_skip)

    // present targets (next row N1/N2)
    uint32_t *rgba8 = nullptr, *bgra8 = nullptr;
    uint32_t present_w = 0, present_h = 0;

    // Frames in flight (vk_ctx_frames_in_flight; vk_context.hip).  A slot owns a frame's surface: backbuffer, step image,
    // present targets, the stream its work runs on and the event that marks its end.  The CURRENT slot's surface lives in the
    // fields above (backbuffer, steps, rgba8, bgra8, present_w/h, stream) -- every other translation unit keeps addressing
    // those -- and is parked in its FrameSlot while another slot is current (frame_slot_switch).  fif_k == 1: one slot, on
    // whatever stream the context runs on.
    struct FrameSlot {
        void *backbuffer = nullptr;
        uint32_t *steps = nullptr, *rgba8 = nullptr, *bgra8 = nullptr;
        uint32_t present_w = 0, present_h = 0;
        hipStream_t stream = nullptr;  // fif_k > 1: the slot's stream, created with the ring at the highest stream priority (vk_ctx_frames_in_flight)
        hipEvent_t done = nullptr;     // recorded by vk_frame_end
        uint64_t id = 0;               // the frame the slot holds (0: none)
        bool ended = false;            // ... and whether vk_frame_end has recorded `done` for it
    } fif[VK_MAX_FRAMES_IN_FLIGHT];
    uint32_t fif_k = 1, fif_cur = 0;
    uint32_t fif_concurrent = 0;  // frames of the ring that may EXECUTE at once (vk_frame_begin); 0: k - 1 for k >= 3 -- one frame always waits, recorded, in its queue
    uint64_t fif_seq = 0;       // frames begun since the ring was sized: frame number n takes slot n % fif_k
    uint64_t fif_last_id = 0;   // ids handed out so far (never reused by a context)
    bool fif_open = false;      // between vk_frame_begin and vk_frame_end

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_open = false, timing_done = false;

    // multi-GPU: this context's RCCL communicator (vk_comm_init_rank / vk_group_create)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 0;
    bool comm_owned = false;
    bool in_group = false;  // a member of a vk_group: its communicator belongs to the group's world
};

// ---- errors --------------------------------------------------------------------------------------------
extern thread_local std::string g_create_err;  // errors of calls that have no context yet (vk_last_error(NULL))

inline int fail(vk_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? VK_ERR_OOM : VK_ERR_HIP,              \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
        }                                                                                      \
    } while (0)

// Frames in flight: how many of the ring's frames execute at once (vk_frame_begin holds the others back), and whether the launch being recorded
// belongs to a frame of a CROWDED ring -- three executing (k = 4) -- where the machine is full and the leaner kernel variants win.
inline uint32_t frames_concurrent(const vk_ctx *ctx) { return ctx->fif_concurrent ? ctx->fif_concurrent : (ctx->fif_k >= 3u ? ctx->fif_k - 1u : ctx->fif_k); }
inline bool frames_crowded(const vk_ctx *ctx) { return ctx->fif_open && ctx->fif_k > 1u && frames_concurrent(ctx) >= 3u; }

inline size_t px_bytes(int fmt) { return fmt == VK_OUT_RGBA16F ? 8 : 16; }
// a pixel of a partition's compact tiles: the backbuffer's pixel, or its three colour channels (VK_WIRE_RGB)
inline size_t wire_px_bytes(int fmt, int wire) { return wire == VK_WIRE_RGB ? px_bytes(fmt) / 4 * 3 : px_bytes(fmt); }

// ---- shared between translation units --------------------------------------------------------------------
int frames_drain(vk_ctx *ctx);   // vk_context.hip: wait for the work of every frame slot (one stream when fif_k == 1)
void free_volume(vk_ctx *ctx);   // vk_volume.hip
void comm_release(vk_ctx *ctx);  // vk_comm.hip

// vk_order.hip: screen-space geometry of a camera, the tile order and its device ring
void cull_rect_cam(const vk_ctx *ctx, const float *cam, int mode, int32_t r[4]);
void compute_tile_order_raw(const vk_ctx *ctx, const float *cam, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                            uint32_t *order, uint32_t *order_pos, uint32_t &order_active, int G);
int tile_order_update(vk_ctx *ctx, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts, bool need_device);
int order_ensure_device(vk_ctx *ctx);
int order_wait(vk_ctx *ctx);

// vk_post.hip
int present_targets(vk_ctx *ctx, uint32_t width, uint32_t height, bool also_bgra);

// vk_render.hip
int check_render(vk_ctx *ctx, int mode, const float *cam, float dt_scale, uint32_t ts, uint32_t rank, uint32_t nranks);
int dispatch_march(vk_ctx *ctx, int mode, const vk::LaunchDesc &L_in, uint32_t flags, const float *reach_cam);
uint32_t launch_flags(const vk_ctx *ctx, uint32_t render_flags, bool batch);

// the kernel-instantiating TUs: each launches on ctx->stream and returns; the caller checks hipGetLastError
void launch_cells(vk_ctx *ctx, const vk::LaunchDesc &L, const vk::VolumeDesc &V, uint32_t grid, bool count, bool skip, bool safe, int walk /* vk_march.hpp: WalkKind */);
void launch_staged(vk_ctx *ctx, const vk::LaunchDesc &L, const vk::VolumeDesc &V, uint32_t grid, bool count, const float *cam);
void launch_compute(vk_ctx *ctx, const vk::LaunchDesc &L, const vk::VolumeDesc &V, uint32_t grid, bool count, bool records, bool skip);
void launch_procedural(vk_ctx *ctx, const vk::LaunchDesc &L, uint32_t grid, bool count, float time, bool device_sine);
