// vk_context.hip -- context lifetime, uniforms, backbuffer, read-back, counters, timers and tuning knobs of the C-ABI
// (include/vokselis_hip.h).  gfx950 only; no CPU fallback: every entry point fails with VK_ERR_HIP / VK_ERR_NO_DEVICE when
// the HIP runtime or the device is unavailable.
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

thread_local std::string g_create_err;

// ---- frame slots (frames in flight) ------------------------------------------------------------------------
// The current slot's surface lives in the context's own fields; switching parks it and loads another slot's.
static void frame_slot_switch(vk_ctx *ctx, uint32_t to) {
    if (to == ctx->fif_cur) return;
    vk_ctx::FrameSlot &a = ctx->fif[ctx->fif_cur], &b = ctx->fif[to];
    a.backbuffer = ctx->backbuffer; a.steps = ctx->steps; a.rgba8 = ctx->rgba8; a.bgra8 = ctx->bgra8;
    a.present_w = ctx->present_w; a.present_h = ctx->present_h;
    ctx->backbuffer = b.backbuffer; ctx->steps = b.steps; ctx->rgba8 = b.rgba8; ctx->bgra8 = b.bgra8;
    ctx->present_w = b.present_w; ctx->present_h = b.present_h;
    b.backbuffer = nullptr; b.steps = nullptr; b.rgba8 = b.bgra8 = nullptr;
    if (ctx->fif_k > 1) ctx->stream = b.stream;
    ctx->fif_cur = to;
}

static vk_ctx::FrameSlot *frame_slot_of(vk_ctx *ctx, uint64_t id, uint32_t *index) {
    if (id == 0) return nullptr;
    for (uint32_t i = 0; i < ctx->fif_k; i++)
        if (ctx->fif[i].id == id) { if (index) *index = i; return &ctx->fif[i]; }
    return nullptr;
}

int frames_drain(vk_ctx *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->fif_k > 1)
        for (uint32_t i = 0; i < ctx->fif_k; i++)
            if (ctx->fif[i].stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->fif[i].stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

static void frame_slot_release(vk_ctx::FrameSlot &s, bool keep_stream) {  // a parked slot's device memory, event and stream
    if (s.backbuffer) (void)hipFree(s.backbuffer);
    if (s.steps) (void)hipFree(s.steps);
    if (s.rgba8) (void)hipFree(s.rgba8);
    if (s.bgra8) (void)hipFree(s.bgra8);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.stream && !keep_stream) (void)hipStreamDestroy(s.stream);
    s = vk_ctx::FrameSlot();
}

// (Re)allocate the CURRENT slot's backbuffer for the context's shape and clear it (HdrBackBuffer::new); its step image goes.
static int surface_alloc_current(vk_ctx *ctx) {
    if (ctx->backbuffer) (void)hipFree(ctx->backbuffer);
    if (ctx->steps) (void)hipFree(ctx->steps);
    ctx->backbuffer = nullptr;
    ctx->steps = nullptr;
    HIP_TRY(ctx, hipMalloc(&ctx->backbuffer, (size_t)ctx->width * ctx->height * px_bytes(ctx->out_format)));
    return vk_backbuffer_clear(ctx);
}

// Run `fn` with the frame's slot as the context's current surface (and stream), then restore the current one.
template <class F>
static int with_frame_slot(vk_ctx *ctx, uint64_t frame_id, const char *who, F fn) {
    if (!ctx) return VK_ERR_INVALID;
    uint32_t i = 0;
    vk_ctx::FrameSlot *s = frame_slot_of(ctx, frame_id, &i);
    if (!s) return fail(ctx, VK_ERR_INVALID, std::string(who) + (frame_id == 0 || frame_id > ctx->fif_last_id ? ": no such frame" : ": that frame's slot has been taken by a later frame"));
    if (!s->ended) return fail(ctx, VK_ERR_INVALID, std::string(who) + ": that frame is still open (vk_frame_end first)");
    const uint32_t cur = ctx->fif_cur;
    frame_slot_switch(ctx, i);
    const int rc = fn();
    frame_slot_switch(ctx, cur);
    return rc;
}

extern "C" {

int vk_abi_version(void) { return VK_ABI_VERSION; }

uint32_t vk_dispatch_optimal(uint32_t len, uint32_t subgroup_size) {
    if (subgroup_size == 0) return 0;
    uint32_t padded = (subgroup_size - len % subgroup_size) % subgroup_size;
    return (len + padded) / subgroup_size;
}

const char *vk_last_error(vk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int vk_ctx_create(int device_ordinal, vk_ctx **out) {
    if (!out) return fail(nullptr, VK_ERR_INVALID, "vk_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, VK_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, VK_ERR_INVALID, "vk_ctx_create: device ordinal out of range");
    vk_ctx *ctx = new (std::nothrow) vk_ctx();
    if (!ctx) return fail(nullptr, VK_ERR_OOM, "vk_ctx_create: host allocation failed");
    ctx->device = device_ordinal;
    auto bail = [&](hipError_t err, const char *what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(err);
        delete ctx;
        return fail(nullptr, VK_ERR_HIP, msg);
    };
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipGetDeviceProperties(&ctx->prop, device_ordinal)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string msg = std::string("device is ") + ctx->prop.gcnArchName + ", this library carries gfx950 code only";
        delete ctx;
        return fail(nullptr, VK_ERR_NO_DEVICE, msg);
    }
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    ctx->stream = ctx->own_stream;
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc(&ctx->counters, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMemset(ctx->counters, 0, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset(counters)");
    *out = ctx;
    return VK_OK;
}

int vk_ctx_destroy(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)frames_drain(ctx);
    frame_slot_switch(ctx, 0);  // (slot 0's surface is in the fields freed below)
    for (uint32_t i = 0; i < VK_MAX_FRAMES_IN_FLIGHT; i++) frame_slot_release(ctx->fif[i], false);
    if (ctx->fif_k > 1) ctx->stream = ctx->own_stream;  // (the ring's streams are gone)
    ctx->fif_k = 1;
    comm_release(ctx);
    free_volume(ctx);
    if (ctx->backbuffer) (void)hipFree(ctx->backbuffer);
    if (ctx->steps) (void)hipFree(ctx->steps);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->rgba8) (void)hipFree(ctx->rgba8);
    if (ctx->bgra8) (void)hipFree(ctx->bgra8);
    if (ctx->trace) (void)hipFree(ctx->trace);
    if (ctx->d_ring) (void)hipFree(ctx->d_ring);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
    for (auto &b : ctx->batch) { if (b.d) (void)hipFree(b.d); if (b.h) (void)hipHostFree(b.h); if (b.ev) (void)hipEventDestroy(b.ev); }
    for (auto &r : ctx->batch_retired) { (void)hipFree(r.first); (void)hipHostFree(r.second); }
    for (hipEvent_t e : ctx->ring_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->ring_guard_ev) (void)hipEventDestroy(ctx->ring_guard_ev);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return VK_OK;
}

int vk_ctx_set_stream(vk_ctx *ctx, void *hip_stream) {
    if (!ctx) return VK_ERR_INVALID;
    if (ctx->fif_k > 1) return fail(ctx, VK_ERR_INVALID, "vk_ctx_set_stream: a context with frames in flight runs its slots on streams of its own");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return VK_OK;
}

int vk_ctx_sync(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    return frames_drain(ctx);
}

int vk_device_info(vk_ctx *ctx, char *name, size_t name_cap, int *compute_units, int *arch_is_gfx950,
                   size_t *total_mem_bytes) {
    if (!ctx) return VK_ERR_INVALID;
    if (name && name_cap) std::snprintf(name, name_cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (arch_is_gfx950) *arch_is_gfx950 = std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) == 0;
    if (total_mem_bytes) *total_mem_bytes = ctx->prop.totalGlobalMem;
    return VK_OK;
}

// ---- uniforms ------------------------------------------------------------------------------------

int vk_set_uniform(vk_ctx *ctx, const void *blob48) {
    if (!ctx || !blob48) return fail(ctx, VK_ERR_INVALID, "vk_set_uniform: NULL argument");
    std::memcpy(ctx->uniform, blob48, 48);  // read by neither fs_main nor get_col2
    return VK_OK;
}

int vk_set_camera(vk_ctx *ctx, const void *blob144) {
    if (!ctx || !blob144) return fail(ctx, VK_ERR_INVALID, "vk_set_camera: NULL argument");
    std::memcpy(ctx->camera, blob144, 144);
    for (int i = 0; i < 36; i++)
        if (!std::isfinite(ctx->camera[i])) { ctx->have_camera = false; return fail(ctx, VK_ERR_INVALID, "camera blob has non-finite entries"); }
    ctx->have_camera = true;
    return VK_OK;
}

// ---- backbuffer ------------------------------------------------------------------------------------

int vk_backbuffer_resize(vk_ctx *ctx, uint32_t width, uint32_t height, int out_format) {
    if (!ctx) return VK_ERR_INVALID;
    if (width == 0 || height == 0 || width > 32768 || height > 32768) return fail(ctx, VK_ERR_INVALID, "backbuffer size must be in [1, 32768]");
    if (out_format != VK_OUT_RGBA32F && out_format != VK_OUT_RGBA16F) return fail(ctx, VK_ERR_INVALID, "unknown output format");
    if (ctx->fif_open) return fail(ctx, VK_ERR_INVALID, "vk_backbuffer_resize: a frame is open (vk_frame_end first)");
    int drc = frames_drain(ctx);
    if (drc) return drc;
    for (auto &b : ctx->batch) b.id = 0;  // batch ids held by the caller named tiles of the old shape
    ctx->batch_key.clear();
    for (auto &r : ctx->batch_retired) { (void)hipFree(r.first); (void)hipHostFree(r.second); }
    ctx->batch_retired.clear();
    ctx->width = width;
    ctx->height = height;
    ctx->out_format = out_format;
    // every slot of the ring takes the new shape; the frames they held are gone
    const uint32_t cur = ctx->fif_cur;
    int rc = VK_OK;
    for (uint32_t i = 0; i < ctx->fif_k && rc == VK_OK; i++) {
        frame_slot_switch(ctx, i);
        ctx->fif[i].id = 0; ctx->fif[i].ended = false;
        rc = surface_alloc_current(ctx);
    }
    frame_slot_switch(ctx, cur);
    if (rc != VK_OK) { ctx->width = ctx->height = 0; return rc; }
    return VK_OK;
}

// ---- frames in flight ------------------------------------------------------------------------------------
// (include/vokselis_hip.h: the reference's queue runs ahead of the GPU, src/lib.rs:178-194; a swapchain bounds how far)

int vk_ctx_frames_in_flight(vk_ctx *ctx, uint32_t k) {
    if (!ctx) return VK_ERR_INVALID;
    if (k == 0 || k > VK_MAX_FRAMES_IN_FLIGHT) return fail(ctx, VK_ERR_INVALID, "vk_ctx_frames_in_flight: k must be in [1, " + std::to_string(VK_MAX_FRAMES_IN_FLIGHT) + "]");
    if (ctx->fif_open) return fail(ctx, VK_ERR_INVALID, "vk_ctx_frames_in_flight: a frame is open (vk_frame_end first)");
    if (k > 1 && ctx->in_group) return fail(ctx, VK_ERR_INVALID, "vk_ctx_frames_in_flight: a member of a vk_group renders the group's batches on the group's streams");
    if (k > 1 && ctx->stream != ctx->own_stream && ctx->fif_k == 1)
        return fail(ctx, VK_ERR_INVALID, "vk_ctx_frames_in_flight: the context runs on a caller's stream (vk_ctx_set_stream(ctx, NULL) first)");
    int drc = frames_drain(ctx);
    if (drc) return drc;
    frame_slot_switch(ctx, 0);
    for (uint32_t i = 0; i < VK_MAX_FRAMES_IN_FLIGHT; i++) { ctx->fif[i].id = 0; ctx->fif[i].ended = false; }
    ctx->fif_seq = 0;
    for (int i = 0; i < 16; i++) { ctx->ring_use_stream[i] = nullptr; ctx->ring_use_frame[i] = 0; }  // (drained: no reader is left)
    for (uint32_t i = k; i < VK_MAX_FRAMES_IN_FLIGHT; i++) frame_slot_release(ctx->fif[i], false);  // a smaller ring
    auto one_slot = [&]() {  // one slot: on the context's own stream (or the caller's, vk_ctx_set_stream)
        if (ctx->fif[0].stream) { (void)hipStreamDestroy(ctx->fif[0].stream); ctx->fif[0].stream = nullptr; }
        if (ctx->fif_k > 1) ctx->stream = ctx->own_stream;
        ctx->fif_k = 1;
    };
    if (k == 1) { one_slot(); return VK_OK; }
    // The ring's streams.  The ROCm runtime multiplexes a process's streams onto a few hardware queues PER PRIORITY LEVEL (GPU_MAX_HW_QUEUES,
    // 4 by default), and two slots that land on one queue run one after the other -- in a process that already holds a handful of streams
    // (PyTorch's, other contexts') exactly that happened to default-priority slot streams (three in flight ran like two).  The slots
    // therefore take streams of the highest priority level, a queue pool of their own: VK_MAX_FRAMES_IN_FLIGHT of them fit it, and frames --
    // the latency-bound work of the process -- are not queued behind its other launches.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const uint32_t old_k = ctx->fif_k;
    int rc = VK_OK;
    for (uint32_t i = 0; i < k && rc == VK_OK; i++) {
        vk_ctx::FrameSlot &s = ctx->fif[i];
        if (!s.stream) {
            hipError_t e = hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio_greatest);
            if (e != hipSuccess) rc = fail(ctx, VK_ERR_HIP, std::string("vk_ctx_frames_in_flight: hipStreamCreateWithPriority: ") + hipGetErrorString(e));
        }
    }
    if (rc == VK_OK) {
        ctx->fif_k = k;  // (frame_slot_switch moves streams once the ring has more than one slot)
        ctx->stream = ctx->fif[0].stream;
        for (uint32_t i = std::max(old_k, 1u); i < k && rc == VK_OK; i++) {
            if (!ctx->width) break;  // (no backbuffer yet: vk_backbuffer_resize sizes every slot)
            frame_slot_switch(ctx, i);  // a new slot takes the backbuffer's current shape, cleared
            rc = surface_alloc_current(ctx);
            frame_slot_switch(ctx, 0);
        }
    }
    if (rc != VK_OK) {  // back to one slot: never a ring with a slot that has no surface
        (void)frames_drain(ctx);
        for (uint32_t i = 1; i < VK_MAX_FRAMES_IN_FLIGHT; i++) frame_slot_release(ctx->fif[i], false);
        ctx->fif_k = std::max(ctx->fif_k, 2u);  // (so that one_slot moves the context back to its own stream)
        one_slot();
        return rc;
    }
    return frames_drain(ctx);  // (the clears)
}

int vk_frame_begin(vk_ctx *ctx, uint64_t *frame_id) {
    if (!ctx) return VK_ERR_INVALID;
    if (ctx->fif_open) return fail(ctx, VK_ERR_INVALID, "vk_frame_begin: the previous frame is still open (vk_frame_end first)");
    if (!ctx->backbuffer) return fail(ctx, VK_ERR_INVALID, "vk_frame_begin: no backbuffer (vk_backbuffer_resize)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t slot = (uint32_t)(ctx->fif_seq % ctx->fif_k);
    vk_ctx::FrameSlot &s = ctx->fif[slot];
    // the swapchain's back-pressure (Surface::get_current_texture, src/context.rs:252): the frame that used this slot k frames ago has to be
    // through before the slot is written again.  This is also what bounds how far launches lag behind the host (the tile-order ring of
    // vk_order.hip holds 16 cameras: with at most VK_MAX_FRAMES_IN_FLIGHT frames in flight no slot of it is rewritten under a running launch).
    // One slot: its frames follow one another on one stream, whose order alone protects the surface -- no wait (ABI 4's behaviour).
    if (ctx->fif_k > 1 && s.id != 0 && s.ended) HIP_TRY(ctx, hipEventSynchronize(s.done));
    if (!s.done) HIP_TRY(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    frame_slot_switch(ctx, slot);
    ctx->fif_seq++;
    s.id = ++ctx->fif_last_id;
    s.ended = false;
    ctx->fif_open = true;
    // One frame of a ring of three or four is always RECORDED AND WAITING in its queue while the others execute: this frame's stream waits (on
    // the GPU) for the end of the frame k - 1 before it.  Single-frame launches dealt to the machine all together share it evenly, end together
    // and leave it to a lone frame while the host records the next ones (C2, k = 3, no limit: three kernels ran 48 % of the time, a lone one
    // 31 %, 0.089 ms per frame); with one frame held back the next starts the moment a frame ends, with no host latency in between
    // (0.080 at k = 3, 0.072 at k = 4 with the lean kernel; the xor frame 0.055 / 0.050; profiles/r06_frames_in_flight.txt).  fif_concurrent overrides k - 1.
    const uint32_t conc = frames_concurrent(ctx);
    if (ctx->fif_k > conc && s.id > conc) {
        const uint64_t before = s.id - conc;
        for (uint32_t i = 0; i < ctx->fif_k; i++)
            if (ctx->fif[i].id == before && ctx->fif[i].ended && i != slot) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->fif[i].done, 0));
    }
    if (frame_id) *frame_id = s.id;
    return VK_OK;
}

int vk_frame_end(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    if (!ctx->fif_open) return fail(ctx, VK_ERR_INVALID, "vk_frame_end without vk_frame_begin");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    vk_ctx::FrameSlot &s = ctx->fif[ctx->fif_cur];
    HIP_TRY(ctx, hipEventRecord(s.done, ctx->stream));
    s.ended = true;
    ctx->fif_open = false;
    return VK_OK;
}

int vk_frame_wait(vk_ctx *ctx, uint64_t frame_id) {
    if (!ctx) return VK_ERR_INVALID;
    if (frame_id == 0 || frame_id > ctx->fif_last_id) return fail(ctx, VK_ERR_INVALID, "vk_frame_wait: no such frame");
    vk_ctx::FrameSlot *s = frame_slot_of(ctx, frame_id, nullptr);
    if (!s) return VK_OK;  // its slot has been taken again: vk_frame_begin waited for it then
    if (!s->ended) return fail(ctx, VK_ERR_INVALID, "vk_frame_wait: that frame is still open (vk_frame_end first)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(s->done));
    return VK_OK;
}

int vk_frame_readback(vk_ctx *ctx, uint64_t frame_id, void *dst, size_t row_pitch_bytes) {
    return with_frame_slot(ctx, frame_id, "vk_frame_readback", [&] { return vk_readback(ctx, dst, row_pitch_bytes); });
}

int vk_frame_capture(vk_ctx *ctx, uint64_t frame_id, void *dst, size_t dst_bytes, uint32_t *out_width, uint32_t *out_height, uint32_t *out_padded_bytes_per_row) {
    return with_frame_slot(ctx, frame_id, "vk_frame_capture", [&] { return vk_capture_frame(ctx, dst, dst_bytes, out_width, out_height, out_padded_bytes_per_row); });
}

int vk_frame_info(vk_ctx *ctx, uint64_t frame_id, void **backbuffer, void **rgba8, int *complete) {
    return with_frame_slot(ctx, frame_id, "vk_frame_info", [&] {
        if (backbuffer) *backbuffer = ctx->backbuffer;
        if (rgba8) *rgba8 = ctx->rgba8;
        if (complete) { *complete = hipEventQuery(ctx->fif[ctx->fif_cur].done) == hipSuccess ? 1 : 0; (void)hipGetLastError(); }  // (hipErrorNotReady is an answer, not an error to find later)
        return (int)VK_OK;
    });
}

int vk_backbuffer_info(vk_ctx *ctx, uint32_t *width, uint32_t *height, int *out_format, void **device_ptr) {
    if (!ctx) return VK_ERR_INVALID;
    if (width) *width = ctx->width;
    if (height) *height = ctx->height;
    if (out_format) *out_format = ctx->out_format;
    if (device_ptr) *device_ptr = ctx->backbuffer;
    return VK_OK;
}

// Device buffers for hosts without another allocator (the C++ host, a plain C consumer): frame batches, gather buffers.
int vk_device_alloc(vk_ctx *ctx, size_t bytes, void **ptr) {
    if (!ctx || !ptr || bytes == 0) return fail(ctx, VK_ERR_INVALID, "vk_device_alloc: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *ptr = nullptr;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? VK_ERR_OOM : VK_ERR_HIP, std::string("vk_device_alloc: ") + hipGetErrorString(e));
    return VK_OK;
}

int vk_device_free(vk_ctx *ctx, void *ptr) {
    if (!ctx) return VK_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(ptr));
    return VK_OK;
}

int vk_device_download(vk_ctx *ctx, void *dst_host, const void *src_device, size_t bytes) {
    if (!ctx || !dst_host || !src_device) return fail(ctx, VK_ERR_INVALID, "vk_device_download: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

// ---- results ------------------------------------------------------------------------------------

int vk_readback(vk_ctx *ctx, void *dst, size_t row_pitch_bytes) {
    if (!ctx || !dst) return fail(ctx, VK_ERR_INVALID, "vk_readback: NULL argument");
    if (!ctx->backbuffer) return fail(ctx, VK_ERR_INVALID, "no backbuffer");
    const size_t row = (size_t)ctx->width * px_bytes(ctx->out_format);
    if (row_pitch_bytes < row) return fail(ctx, VK_ERR_INVALID, "row pitch smaller than a row");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy2DAsync(dst, row_pitch_bytes, ctx->backbuffer, row, row, ctx->height, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

int vk_step_counts(vk_ctx *ctx, uint64_t *s_ref, uint64_t *s_sampled) {
    if (!ctx) return VK_ERR_INVALID;
    unsigned long long h[2] = {0, 0};
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (s_ref) *s_ref = h[0];
    if (s_sampled) *s_sampled = h[1];
    return VK_OK;
}

int vk_simt_census(vk_ctx *ctx, uint64_t out[4]) {
    if (!ctx || !out) return fail(ctx, VK_ERR_INVALID, "vk_simt_census: NULL argument");
    unsigned long long h[8] = {0};
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; i++) out[i] = h[2 + i];
    return VK_OK;
}

int vk_debug_wave_trace(vk_ctx *ctx, int enable, uint64_t *out, size_t n_blocks) {
    if (!ctx) return VK_ERR_INVALID;
    ctx->want_trace = enable != 0;
    if (!out) return VK_OK;
    if (!ctx->trace || n_blocks > ctx->trace_blocks) return fail(ctx, VK_ERR_INVALID, "vk_debug_wave_trace: no trace of that size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->trace, n_blocks * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VK_OK;
}

int vk_debug_set_param(vk_ctx *ctx, const char *name, double value) {
    if (!ctx || !name) return VK_ERR_INVALID;
    const std::string n(name);
    if (n == "stage_cap_bytes") ctx->stage_cap_bytes = (uint32_t)value;          // LDS window of the staged march (next render)
    else if (n == "stage_slab_cells") ctx->stage_slab_cells = (uint32_t)value;   // cells per slab along the major axis (next render)
    else if (n == "trip_log_cap") { if (value < 0 || value > 4096 || ((uint32_t)value & 7u)) return fail(ctx, VK_ERR_INVALID, "trip_log_cap: a multiple of 8 up to 4096"); ctx->trip_log_cap = (uint32_t)value; }
    else if (n == "stage_group") ctx->stage_group = (uint32_t)value;             // staged march: windows shared by the four waves of a group (docs/archive/tools/staged_group.py)
    else if (n == "frame_runs") ctx->frame_runs = (uint32_t)value;               // batched launches: runs of consecutive frames per XCD (docs/archive/tools/frame_runs.py)
    else if (n == "stage_grow_every") ctx->stage_grow_every = (uint32_t)value;   // slab search growth period (next render)
    else if (n == "stage_row_pad") ctx->stage_row_pad = (uint32_t)value;          // odd row pitch of the staged window (next render)
    else if (n == "wave_prio") ctx->wave_prio = (uint32_t)value;
    else if (n == "fif_concurrent") ctx->fif_concurrent = (uint32_t)value;  // frames in flight: how many of the ring's frames may execute at once (0: all)
    else if (n == "order_never_inline") ctx->order_never_inline = (uint32_t)value;  // A/B: the tile order through the device table for every launch
    else if (n == "walk_cap") ctx->walk_cap = (uint32_t)value;
    else if (n == "walk_cap_all") ctx->walk_cap_all = (uint32_t)value;
    else if (n == "render_tile") ctx->render_tile = ((uint32_t)value & ~7u);
    else if (n == "pair_ring") ctx->pair_ring = (uint32_t)value;
    else if (n == "probe_ahead") ctx->probe_ahead = (uint32_t)value;
    else if (n == "pair_walk_min") ctx->pair_walk_min = (uint32_t)std::min(std::max(2.0, value), 1e6);
    else if (n == "order_rays") { ctx->order_rays = ctx->order_rays_batch = (uint32_t)std::min<double>(std::max<double>(value, 1), 8); ctx->batch_key.clear(); ctx->order_key.clear(); }
    else if (n == "naive_lds_pad") ctx->naive_lds_pad = (uint32_t)value;          // experiments: caps the cell kernels' waves per SIMD
    else if (n == "stage_copies_mask") ctx->stage_copies_mask = (uint32_t)value; // which brick copies to build (next upload)
    else return fail(ctx, VK_ERR_INVALID, "vk_debug_set_param: unknown parameter " + n);
    return VK_OK;
}

int vk_step_counts_reset(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters, 0, 8 * sizeof(unsigned long long), ctx->stream));
    return VK_OK;
}

int vk_readback_steps(vk_ctx *ctx, uint32_t *dst) {
    if (!ctx || !dst) return fail(ctx, VK_ERR_INVALID, "vk_readback_steps: NULL argument");
    if (!ctx->steps) return fail(ctx, VK_ERR_INVALID, "no VK_RENDER_COUNT launch since the last resize");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->steps, (size_t)ctx->width * ctx->height * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

int vk_timer_begin(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    ctx->timing_open = true;
    ctx->timing_done = false;
    return VK_OK;
}

int vk_timer_end(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    if (!ctx->timing_open) return fail(ctx, VK_ERR_INVALID, "vk_timer_end without vk_timer_begin");
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timing_open = false;
    ctx->timing_done = true;
    return VK_OK;
}

int vk_timer_elapsed_ms(vk_ctx *ctx, float *ms) {
    if (!ctx || !ms) return fail(ctx, VK_ERR_INVALID, "vk_timer_elapsed_ms: NULL argument");
    if (!ctx->timing_done) return fail(ctx, VK_ERR_INVALID, "no completed timer bracket");
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return VK_OK;
}


}  // extern "C"
