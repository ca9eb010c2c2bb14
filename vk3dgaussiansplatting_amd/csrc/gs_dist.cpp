// gs_dist.cpp -- the one exchange step of a multi-GPU frame behind the C-ABI: a gather of RGBA8 tile rows to the root
// rank over RCCL (xGMI inside a node).  No reference counterpart (the reference has no collective);
// frames shard by screen-tile rows (gs_set_tile_rows / gs_set_tile_rows_interleaved), the gaussians are
// replicated, and nothing but the finished rows crosses GPUs.
//
// One process per GPU.  A gather to one root is R - 1 point-to-point transfers, each over the peer's own xGMI link to
// the root (7 links x ~153 GB/s per GPU), so it is written as exactly that -- grouped ncclSend / ncclRecv -- rather than
// as a ring collective: at 4K a band of an 8-way shard is 4.1 MB, ~30 us on its link.  Contiguous bands land where they
// belong in the root's frame (no padding, no assembly pass: band p is received at pixel row 16 * edge[p]); rows dealt
// round-robin arrive packed and are re-ordered by R strided copies.
//
// gs_render_sharded_async keeps TWO frames in flight: the gather of frame f runs on a stream of its own behind an event,
// frame f + 1's kernels start at once into the other slot, and the assembled frame stays in the root's HBM until somebody
// asks for it (gs_sharded_frame / gs_sharded_read) -- the shape of the reference's frames in flight with the present
// decoupled from the recording (Renderer.cpp:297-404).
//
// RCCL is bound at gs_dist_init (dlopen of librccl.so.1), not at link time: a process that never shards a frame does
// not map RCCL at all, and a process that already holds an RCCL (PyTorch-ROCm wheels bundle one under the same SONAME)
// gets THAT copy instead of a second one -- the same one-runtime-per-process rule INTEGRATION.md describes for
// libamdhip64.  Nor is it a BUILD dependency: the eight entry points used here are declared below (their ABI has been
// stable since NCCL 2.7: an opaque communicator pointer, a 128-byte id passed by value, int enums), so the library
// builds on a ROCm install without the RCCL development headers.
#include "gs_ctx.h"
#include "gs_balance.h"

#include <dlfcn.h>

#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>

namespace {

// ---- the slice of <rccl/rccl.h> this file programs against ----
struct RcclUniqueId { char internal[128]; };
using RcclComm = void*;
using RcclResult = int;                 // ncclSuccess == 0
constexpr int kRcclUint8 = 1;           // ncclUint8 (ncclInt8 / ncclChar == 0)
static_assert(sizeof(RcclUniqueId) == GS_DIST_UNIQUE_ID_BYTES, "GS_DIST_UNIQUE_ID_BYTES is sizeof(ncclUniqueId)");

struct Rccl {
    void* handle = nullptr;
    RcclResult (*GetUniqueId)(RcclUniqueId*) = nullptr;
    RcclResult (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    RcclResult (*CommDestroy)(RcclComm) = nullptr;
    RcclResult (*GroupStart)() = nullptr;
    RcclResult (*GroupEnd)() = nullptr;
    RcclResult (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    RcclResult (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(RcclResult) = nullptr;
    std::string error;
};

void bind_rccl(Rccl& r) {
    // GS_RCCL_LIBRARY: a deployment's own build of RCCL, by path (a path with a slash is opened as that file even when a
    // library of the same SONAME is already mapped) -- and how the tests put tools/mock_rccl under a process that holds torch
    if (const char* path = std::getenv("GS_RCCL_LIBRARY")) {
        r.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) { r.error = std::string("GS_RCCL_LIBRARY: ") + dlerror(); return; }
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (r.handle) break;
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.handle) { r.error = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
    bool ok = true;
    auto bind = [&](auto& fn, const char* sym) {
        fn = reinterpret_cast<std::decay_t<decltype(fn)>>(dlsym(r.handle, sym));
        if (!fn) { ok = false; r.error = std::string("librccl: missing symbol ") + sym; }
    };
    bind(r.GetUniqueId, "ncclGetUniqueId");
    bind(r.CommInitRank, "ncclCommInitRank");
    bind(r.CommDestroy, "ncclCommDestroy");
    bind(r.GroupStart, "ncclGroupStart");
    bind(r.GroupEnd, "ncclGroupEnd");
    bind(r.Send, "ncclSend");
    bind(r.Recv, "ncclRecv");
    bind(r.GetErrorString, "ncclGetErrorString");
    if (!ok) { dlclose(r.handle); r.handle = nullptr; }
}

Rccl& rccl() {                       // bound once per process, whichever context asks first (contexts may live on other threads)
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, bind_rccl, std::ref(r));
    return r;
}

int fail(gs_ctx* c, int code, const std::string& msg) {
    if (c) c->last_error = msg;
    return code;
}

#define DIST_TRY(c, call)                                                                              \
    do {                                                                                               \
        const hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail((c), GS_ERR_HIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

size_t band_bytes(const gs_ctx* c, uint32_t rb, uint32_t re) {            // the pixel rows of tile rows [rb, re) that exist
    const uint32_t y0 = std::min(rb * 16u, c->height), y1 = std::min(re * 16u, c->height);
    return (size_t)(y1 - y0) * c->width * 4u;
}

bool rows_as_sharded(const gs_ctx* c) {
    return c->dist_rows_sig[0] == c->row_begin && c->dist_rows_sig[1] == c->row_end && c->dist_rows_sig[2] == c->row_stride &&
           c->dist_rows_sig[3] == c->first_row && c->dist_rows_sig[4] == (c->compact_out ? 1u : 0u);
}

void remember_rows(gs_ctx* c) {
    c->dist_rows_sig[0] = c->row_begin; c->dist_rows_sig[1] = c->row_end; c->dist_rows_sig[2] = c->row_stride;
    c->dist_rows_sig[3] = c->first_row; c->dist_rows_sig[4] = c->compact_out ? 1u : 0u;
}

int apply_band(gs_ctx* c) {                                                // contiguous / balanced: this rank's band from the edges
    const int rc = gs_set_tile_rows(c, c->dist_edges[(size_t)c->dist_rank], c->dist_edges[(size_t)c->dist_rank + 1u]);
    if (rc == GS_OK) remember_rows(c);
    return rc;
}

// The exchange of one slot, enqueued on `st` (which already waits for the slot's frame).
int enqueue_gather(gs_ctx* c, int slot, hipStream_t st) {
    Rccl& r = rccl();
    RcclComm comm = c->dist_comm;
    const bool root = c->dist_rank == 0;
    const uint32_t R = (uint32_t)c->dist_world;
    RcclResult rc = 0;
    if (c->dist_dealing == GS_ROWS_INTERLEAVED) {
        const size_t block = (size_t)16u * c->width * 4u;                  // one tile row of pixels
        if (root) {
            uint8_t* gathered = static_cast<uint8_t*>(c->dist_gathered[slot]);
            if (R > 1u) {
                rc = r.GroupStart();
                for (uint32_t p = 1; p < R && rc == 0; ++p)
                    rc = r.Recv(gathered + (size_t)p * c->dist_strip_bytes, c->dist_strip_bytes, kRcclUint8, (int)p, comm, st);
                const RcclResult rc_end = r.GroupEnd();
                if (rc == 0) rc = rc_end;
            }
            if (rc != 0) return fail(c, GS_ERR_HIP, std::string("gs_render_sharded: ") + r.GetErrorString(rc));
            // strip p, block k (16 pixel rows) -> tile row k * R + p of the frame: one strided copy per rank
            for (uint32_t p = 0; p < R; ++p) {
                const uint32_t owned = c->grid_h > p ? (c->grid_h - p + R - 1u) / R : 0u;
                if (!owned) continue;
                const uint8_t* src = p == 0u ? static_cast<uint8_t*>(c->dist_strip[slot]) : gathered + (size_t)p * c->dist_strip_bytes;
                DIST_TRY(c, hipMemcpy2DAsync(static_cast<uint8_t*>(c->dist_image[slot]) + (size_t)p * block, block * R, src, block, block,
                                             owned, hipMemcpyDeviceToDevice, st));
            }
        } else {
            rc = r.Send(c->dist_strip[slot], c->dist_strip_bytes, kRcclUint8, 0, comm, st);
            if (rc != 0) return fail(c, GS_ERR_HIP, std::string("gs_render_sharded: ") + r.GetErrorString(rc));
        }
        return GS_OK;
    }
    // contiguous / balanced bands: band p goes straight to its place in the root's frame; the root rendered its own in place
    if (root) {
        if (R > 1u) {
            rc = r.GroupStart();
            for (uint32_t p = 1; p < R && rc == 0; ++p) {
                const size_t bytes = band_bytes(c, c->dist_edges[p], c->dist_edges[p + 1u]);
                if (bytes) rc = r.Recv(static_cast<uint8_t*>(c->dist_image[slot]) + (size_t)c->dist_edges[p] * 16u * c->width * 4u, bytes,
                                       kRcclUint8, (int)p, comm, st);
            }
            const RcclResult rc_end = r.GroupEnd();
            if (rc == 0) rc = rc_end;
        }
    } else {
        const size_t bytes = band_bytes(c, c->row_begin, c->row_end);
        if (bytes) rc = r.Send(c->dist_strip[slot], bytes, kRcclUint8, 0, comm, st);
    }
    if (rc != 0) return fail(c, GS_ERR_HIP, std::string("gs_render_sharded: ") + r.GetErrorString(rc));
    return GS_OK;
}

int check_sharded(gs_ctx* c, const char* who) {
    if (!c->dist_comm || !c->dist_sharded) return fail(c, GS_ERR_INVALID, std::string(who) + ": gs_dist_init + gs_dist_shard_rows first");
    if (!rows_as_sharded(c))
        return fail(c, GS_ERR_INVALID, std::string(who) + ": the context's tile rows were changed after gs_dist_shard_rows (gs_set_tile_rows*): "
                                       "the buffers of the sharded frame belong to the rows dealt there -- call gs_dist_shard_rows again");
    return GS_OK;
}

int slot_of(gs_ctx* c, uint32_t which, const char* who) {
    if (which > 1u) { fail(c, GS_ERR_INVALID, std::string(who) + ": which must be 0 (the last sharded frame) or 1 (the one before)"); return -1; }
    const int slot = c->dist_recent[which];
    if (slot < 0) { fail(c, GS_ERR_NO_SCENE, std::string(who) + ": no such sharded frame yet"); return -1; }
    return slot;
}

}  // namespace

void gsi_dist_free_buffers(gs_ctx* c) {
    if (c->dist_stream) (void)hipStreamSynchronize(c->dist_stream);
    for (int k = 0; k < 2; ++k) {
        for (void** p : {&c->dist_strip[k], &c->dist_gathered[k], &c->dist_image[k]}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        c->dist_used[k] = false;
        c->dist_recent[k] = -1;
    }
    if (c->dist_xchg) { (void)hipFree(c->dist_xchg); c->dist_xchg = nullptr; }
    c->dist_strip_bytes = 0;
    c->dist_sharded = false;
    c->dist_next = 0;
    c->dist_edges.clear();
    c->dist_history.clear();
}

extern "C" {

int gs_dist_unique_id(void* id_out) {
    if (!id_out) return GS_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.handle) { gsi_set_create_error("gs_dist_unique_id: " + r.error); return GS_ERR_HIP; }   // gs_last_error(NULL)
    RcclUniqueId id;
    const RcclResult rc = r.GetUniqueId(&id);
    if (rc != 0) { gsi_set_create_error(std::string("gs_dist_unique_id: ") + r.GetErrorString(rc)); return GS_ERR_HIP; }
    std::memcpy(id_out, &id, sizeof(id));
    return GS_OK;
}

int gs_dist_init(gs_ctx* c, const void* unique_id, int rank, int world) {
    if (!c) return GS_ERR_INVALID;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(c, GS_ERR_INVALID, "gs_dist_init: bad rank / world / id");
    if (c->dist_comm) return fail(c, GS_ERR_INVALID, "gs_dist_init: already initialised (gs_dist_destroy first)");
    Rccl& r = rccl();
    if (!r.handle) return fail(c, GS_ERR_HIP, "gs_dist_init: " + r.error);
    DIST_TRY(c, hipSetDevice(c->device));
    RcclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    RcclComm comm = nullptr;
    const RcclResult rc = r.CommInitRank(&comm, world, id, rank);
    if (rc != 0) return fail(c, GS_ERR_HIP, std::string("gs_dist_init: ncclCommInitRank: ") + r.GetErrorString(rc));
    c->dist_comm = comm;
    c->dist_rank = rank;
    c->dist_world = world;
    return GS_OK;
}

int gs_gather_strips(gs_ctx* c, const void* strip_dev, void* gathered_dev, size_t bytes, int root) {
    if (!c) return GS_ERR_INVALID;
    if (!c->dist_comm) return fail(c, GS_ERR_INVALID, "gs_gather_strips: gs_dist_init not called");
    if (!strip_dev || bytes == 0 || root < 0 || root >= c->dist_world) return fail(c, GS_ERR_INVALID, "gs_gather_strips: bad argument");
    if (c->dist_rank == root && !gathered_dev) return fail(c, GS_ERR_INVALID, "gs_gather_strips: the root needs a destination");
    Rccl& r = rccl();
    RcclComm comm = c->dist_comm;
    DIST_TRY(c, hipSetDevice(c->device));
    RcclResult rc = 0;
    if (c->dist_rank == root) {
        uint8_t* dst = static_cast<uint8_t*>(gathered_dev);
        // the root's own strip never leaves the GPU; the peers' strips arrive over their own links, all in one group
        DIST_TRY(c, hipMemcpyAsync(dst + (size_t)root * bytes, strip_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
        if (c->dist_world > 1) {
            rc = r.GroupStart();
            for (int p = 0; p < c->dist_world && rc == 0; ++p)
                if (p != root) rc = r.Recv(dst + (size_t)p * bytes, bytes, kRcclUint8, p, comm, c->stream);
            const RcclResult rc_end = r.GroupEnd();
            if (rc == 0) rc = rc_end;
        }
    } else {
        rc = r.Send(strip_dev, bytes, kRcclUint8, root, comm, c->stream);
    }
    if (rc != 0) return fail(c, GS_ERR_HIP, std::string("gs_gather_strips: ") + r.GetErrorString(rc));
    return GS_OK;
}

int gs_dist_shard_rows(gs_ctx* c, uint32_t dealing) {
    if (!c) return GS_ERR_INVALID;
    if (!c->dist_comm) return fail(c, GS_ERR_INVALID, "gs_dist_shard_rows: gs_dist_init not called");
    if (!c->capacity) return fail(c, GS_ERR_NO_SCENE, "gs_dist_shard_rows: gs_set_resolution not called");
    if (dealing > GS_ROWS_BALANCED) return fail(c, GS_ERR_INVALID, "gs_dist_shard_rows: dealing must be GS_ROWS_CONTIGUOUS, _INTERLEAVED or _BALANCED");
    const uint32_t R = (uint32_t)c->dist_world, r = (uint32_t)c->dist_rank;
    DIST_TRY(c, hipSetDevice(c->device));
    if (c->stream) DIST_TRY(c, hipStreamSynchronize(c->stream));
    gsi_dist_free_buffers(c);
    c->dist_dealing = dealing;
    int rc;
    if (dealing == GS_ROWS_INTERLEAVED) {
        rc = gs_set_tile_rows_interleaved(c, r, R, 1u);
        if (rc == GS_OK) remember_rows(c);
    } else {
        // GS_ROWS_BALANCED starts from the same equal bands (no frame has been seen yet) -- ceil(Ty / R) rows each, what
        // dist.RowBalancer starts from, so that the two balancers walk the same trajectory; gs_dist_rebalance moves the edges
        c->dist_edges = gs::equal_row_edges(c->grid_h, R);
        rc = apply_band(c);
    }
    if (rc != GS_OK) return rc;
    if (!c->dist_stream) DIST_TRY(c, hipStreamCreateWithFlags(&c->dist_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        if (!c->dist_begin[k]) DIST_TRY(c, hipEventCreate(&c->dist_begin[k]));
        if (!c->dist_rendered[k]) DIST_TRY(c, hipEventCreate(&c->dist_rendered[k]));
        if (!c->dist_done[k]) DIST_TRY(c, hipEventCreateWithFlags(&c->dist_done[k], hipEventDisableTiming));
    }
    // every buffer holds whole frames' worth of tile rows (padded to 16-pixel rows): bands may grow under GS_ROWS_BALANCED
    // without a re-allocation, and 2 x 33 MB at 4K is nothing against 288 GB
    const size_t frame_bytes = (size_t)c->grid_h * 16u * c->width * 4u;
    const uint32_t per = (c->grid_h + R - 1u) / R;
    c->dist_strip_bytes = (size_t)per * 16u * c->width * 4u;
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        const bool needs_strip = dealing == GS_ROWS_INTERLEAVED || r != 0u;
        const size_t strip = dealing == GS_ROWS_INTERLEAVED ? c->dist_strip_bytes : frame_bytes;
        if (needs_strip) { e = hipMalloc(&c->dist_strip[k], strip); if (e == hipSuccess) e = hipMemset(c->dist_strip[k], 0, strip); }
        if (e == hipSuccess && r == 0u) { e = hipMalloc(&c->dist_image[k], frame_bytes); if (e == hipSuccess) e = hipMemset(c->dist_image[k], 0, frame_bytes); }
        if (e == hipSuccess && r == 0u && dealing == GS_ROWS_INTERLEAVED) e = hipMalloc(&c->dist_gathered[k], c->dist_strip_bytes * R);
    }
    if (e == hipSuccess && dealing == GS_ROWS_BALANCED) e = hipMalloc(&c->dist_xchg, (size_t)R * (c->grid_h + 1u) * sizeof(uint32_t));
    if (e != hipSuccess) { gsi_dist_free_buffers(c); return fail(c, GS_ERR_HIP, std::string("gs_dist_shard_rows: ") + hipGetErrorString(e)); }
    c->dist_sharded = true;
    return GS_OK;
}

int gs_dist_bands(const gs_ctx* c, uint32_t* edges_out, uint32_t count) {
    if (!c || !edges_out) return GS_ERR_INVALID;
    if (!c->dist_sharded || c->dist_dealing == GS_ROWS_INTERLEAVED || count != (uint32_t)c->dist_world + 1u) return GS_ERR_INVALID;
    for (uint32_t k = 0; k < count; ++k) edges_out[k] = c->dist_edges[k];
    return GS_OK;
}

int gs_render_sharded_async(gs_ctx* c, const float view[16], const float proj[16], const float cam_pos[3], uint32_t sh_mode) {
    if (!c) return GS_ERR_INVALID;
    if (int rc = check_sharded(c, "gs_render_sharded_async")) return rc;
    DIST_TRY(c, hipSetDevice(c->device));
    const int slot = c->dist_next;
    const bool root = c->dist_rank == 0;
    // the slot's buffers were last touched by the gather of the frame before the previous one
    hipError_t before = hipSuccess;                 // see below: no return between check_sharded and the exchange
    if (c->dist_used[slot]) before = hipStreamWaitEvent(c->stream, c->dist_done[slot], 0);
    { const hipError_t e = hipEventRecord(c->dist_begin[slot], c->stream); if (before == hipSuccess) before = e; }
    uint8_t* target;
    if (c->dist_dealing == GS_ROWS_INTERLEAVED) target = static_cast<uint8_t*>(c->dist_strip[slot]);      // packed rows (compact_output)
    else if (root) target = static_cast<uint8_t*>(c->dist_image[slot]);                                   // in place, real rows
    else target = static_cast<uint8_t*>(c->dist_strip[slot]) - (size_t)c->row_begin * 16u * c->width * 4u;   // band at the top of the strip
    const int rc_render = c->rows_owned ? gs_render_device_async(c, view, proj, cam_pos, sh_mode, target) : GS_OK;
    // a rank whose frame failed still takes part in the exchange (with whatever its rows hold) -- leaving now would leave
    // the other ranks waiting in theirs -- and reports its error afterwards.  The same holds for the event calls around the
    // exchange: from here on nothing returns before enqueue_gather has been called; the first HIP error is reported after it.
    const std::string render_error = rc_render < 0 ? c->last_error : std::string();
    hipError_t first_hip = before;
    auto note = [&](hipError_t e) { if (e != hipSuccess && first_hip == hipSuccess) first_hip = e; };
    note(hipEventRecord(c->dist_rendered[slot], c->stream));
    note(hipStreamWaitEvent(c->dist_stream, c->dist_rendered[slot], 0));
    const int rc_gather = enqueue_gather(c, slot, c->dist_stream);
    note(hipEventRecord(c->dist_done[slot], c->dist_stream));
    c->dist_used[slot] = true;
    c->dist_recent[1] = c->dist_recent[0];
    c->dist_recent[0] = slot;
    c->dist_next = slot ^ 1;
    if (rc_gather < 0) return rc_gather;
    if (first_hip != hipSuccess) { (void)hipGetLastError(); return fail(c, GS_ERR_HIP, std::string("gs_render_sharded_async: ") + hipGetErrorString(first_hip)); }
    if (rc_render < 0) return fail(c, rc_render, render_error);
    return rc_render;
}

int gs_sharded_frame(gs_ctx* c, uint32_t which, void** frame_dev) {
    if (!c || !frame_dev) return GS_ERR_INVALID;
    *frame_dev = nullptr;
    if (int rc = check_sharded(c, "gs_sharded_frame")) return rc;
    const int slot = slot_of(c, which, "gs_sharded_frame");
    if (slot < 0) return which > 1u ? GS_ERR_INVALID : GS_ERR_NO_SCENE;
    DIST_TRY(c, hipSetDevice(c->device));
    DIST_TRY(c, hipEventSynchronize(c->dist_done[slot]));
    if (c->dist_rank == 0) *frame_dev = c->dist_image[slot];
    return GS_OK;
}

int gs_sharded_read(gs_ctx* c, uint32_t which, uint8_t* rgba_out) {
    if (!c) return GS_ERR_INVALID;
    void* dev = nullptr;
    const int rc = gs_sharded_frame(c, which, &dev);
    if (rc != GS_OK) return rc;
    if (c->dist_rank != 0) return GS_OK;
    if (!rgba_out) return fail(c, GS_ERR_INVALID, "gs_sharded_read: rank 0 needs rgba_out");
    DIST_TRY(c, hipMemcpy(rgba_out, dev, (size_t)c->width * c->height * 4u, hipMemcpyDeviceToHost));
    return GS_OK;
}

int gs_render_sharded(gs_ctx* c, const float view[16], const float proj[16], const float cam_pos[3], uint32_t sh_mode,
                      uint8_t* rgba_out) {
    if (!c) return GS_ERR_INVALID;
    if (int rc = check_sharded(c, "gs_render_sharded")) return rc;
    if (c->dist_rank == 0 && !rgba_out) return fail(c, GS_ERR_INVALID, "gs_render_sharded: rank 0 needs rgba_out");
    const int rc_frame = gs_render_sharded_async(c, view, proj, cam_pos, sh_mode);
    const std::string frame_error = rc_frame < 0 ? c->last_error : std::string();
    // whatever the frame returned, the exchange was enqueued: wait for it before reporting
    const int slot = c->dist_recent[0];
    if (slot >= 0 && c->dist_used[slot]) {
        const int rc_read = gs_sharded_read(c, 0u, rgba_out);
        if (rc_frame >= 0 && rc_read < 0) return rc_read;
    }
    if (rc_frame < 0) { (void)hipStreamSynchronize(c->stream); return fail(c, rc_frame, frame_error); }
    // gs_get_timings: this rank's own rows (the gather is not part of the reference's buckets); GS_WARN_OVERFLOW as in gs_render
    return c->rows_owned ? gsi_finish_frame(c) : rc_frame;
}

int gs_dist_rebalance(gs_ctx* c, uint32_t* moved_out) {
    if (!c) return GS_ERR_INVALID;
    if (moved_out) *moved_out = 0u;
    if (int rc = check_sharded(c, "gs_dist_rebalance")) return rc;
    if (c->dist_dealing != GS_ROWS_BALANCED) return fail(c, GS_ERR_INVALID, "gs_dist_rebalance: the rows were not dealt with GS_ROWS_BALANCED");
    const int last = c->dist_recent[0];
    if (last < 0) return fail(c, GS_ERR_NO_SCENE, "gs_dist_rebalance: no sharded frame yet");
    Rccl& r = rccl();
    const uint32_t R = (uint32_t)c->dist_world, me = (uint32_t)c->dist_rank, ty = c->grid_h, words = ty + 1u;
    // from here to the exchange nothing returns: a rank that left on a HIP error of its own would leave the others waiting in
    // their Send / Recv group; the first such error is reported after the group
    hipError_t first_hip = hipSuccess;
    auto note = [&](hipError_t e) { if (e != hipSuccess && first_hip == hipSuccess) first_hip = e; };
    note(hipSetDevice(c->device));
    note(hipStreamSynchronize(c->stream));
    note(hipStreamSynchronize(c->dist_stream));
    // this rank's contribution: the sort elements of each of its tile rows (the last frame's tile ranges) and the GPU time
    // of its share (events around the frame; the gather is not in it)
    std::vector<uint32_t> ranges((size_t)c->grid_w * ty * 2u);
    note(hipMemcpy(ranges.data(), c->ranges, ranges.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> all((size_t)R * words, 0u);
    uint32_t* mine = &all[(size_t)me * words];
    for (uint32_t row = c->row_begin; row < c->row_end; ++row) {
        uint64_t sum = 0;
        for (uint32_t x = 0; x < c->grid_w; ++x) {
            const uint32_t* t = &ranges[((size_t)row * c->grid_w + x) * 2u];
            sum += t[1] - t[0];
        }
        mine[row] = (uint32_t)std::min<uint64_t>(sum, 0xFFFFFFFFull);
    }
    float ms = 0.0f;
    if (c->rows_owned && hipEventElapsedTime(&ms, c->dist_begin[last], c->dist_rendered[last]) != hipSuccess) { ms = 0.0f; (void)hipGetLastError(); }
    std::memcpy(&mine[ty], &ms, sizeof(ms));
    // all-gather as R (R - 1) small point-to-point transfers in one group (at most a few KB each)
    uint32_t* xchg = static_cast<uint32_t*>(c->dist_xchg);
    note(hipMemcpy(xchg + (size_t)me * words, mine, words * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (R > 1u) {
        RcclResult rc = r.GroupStart();
        for (uint32_t p = 0; p < R && rc == 0; ++p)
            if (p != me) rc = r.Send(xchg + (size_t)me * words, words * sizeof(uint32_t), kRcclUint8, (int)p, c->dist_comm, c->stream);
        for (uint32_t p = 0; p < R && rc == 0; ++p)
            if (p != me) rc = r.Recv(xchg + (size_t)p * words, words * sizeof(uint32_t), kRcclUint8, (int)p, c->dist_comm, c->stream);
        const RcclResult rc_end = r.GroupEnd();
        if (rc == 0) rc = rc_end;
        if (rc != 0) return fail(c, GS_ERR_HIP, std::string("gs_dist_rebalance: ") + r.GetErrorString(rc));
    }
    if (first_hip != hipSuccess) { (void)hipGetLastError(); return fail(c, GS_ERR_HIP, std::string("gs_dist_rebalance: ") + hipGetErrorString(first_hip)); }
    DIST_TRY(c, hipStreamSynchronize(c->stream));
    DIST_TRY(c, hipMemcpy(all.data(), xchg, all.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // dist.py: RowBalancer.update, statement for statement.  T_r = F + sum of weight(row): F = the intercept of the
    // least-squares line through the (elements, ms) pairs of the last epochs, inside [0, 0.8 min(T)]; weight(row) =
    // elements(row) x (T_r - F) / E_r (elements(row) when a rank has no time to report).  Identical inputs on every rank ->
    // identical edges, no second exchange.
    std::vector<double> elems(ty, 0.0), weights(ty, 0.0);
    std::vector<double> rank_ms(R, 0.0), totals(R, 0.0);
    // GS_REBALANCE_ELEMENTS_ONLY: cut by element counts alone -- for ranks whose share times say nothing about their shares
    // (several ranks on one GPU: the tests over tools/mock_rccl, a rehearsal); set on every rank or on none
    bool timed = std::getenv("GS_REBALANCE_ELEMENTS_ONLY") == nullptr;
    for (uint32_t p = 0; p < R; ++p) {
        float t;
        std::memcpy(&t, &all[(size_t)p * words + ty], sizeof(t));
        rank_ms[p] = (double)t;
        timed = timed && t > 0.0f;
        for (uint32_t row = c->dist_edges[p]; row < c->dist_edges[p + 1u]; ++row) {
            elems[row] = (double)all[(size_t)p * words + row];
            totals[p] += elems[row];
        }
    }
    weights = elems;
    double fixed = 0.0;
    if (timed) {
        for (uint32_t p = 0; p < R; ++p) c->dist_history.emplace_back(totals[p], rank_ms[p]);
        const size_t keep = (size_t)4u * R;                                 // RowBalancer.HISTORY_EPOCHS
        if (c->dist_history.size() > keep) c->dist_history.erase(c->dist_history.begin(), c->dist_history.end() - (std::ptrdiff_t)keep);
        const double n = (double)c->dist_history.size();
        double m_e = 0.0, m_t = 0.0, var = 0.0, cov = 0.0;
        for (const auto& h : c->dist_history) { m_e += h.first; m_t += h.second; }
        m_e /= n; m_t /= n;
        for (const auto& h : c->dist_history) { var += (h.first - m_e) * (h.first - m_e); cov += (h.first - m_e) * (h.second - m_t); }
        if (var > 0.0 && cov > 0.0) fixed = m_t - (cov / var) * m_e;
        if (!(fixed > 0.0)) fixed = 0.0;
        fixed = std::min(fixed, 0.8 * *std::min_element(rank_ms.begin(), rank_ms.end()));
        for (uint32_t p = 0; p < R; ++p) {
            const uint32_t b = c->dist_edges[p], e = c->dist_edges[p + 1u];
            for (uint32_t row = b; row < e; ++row)
                weights[row] = totals[p] > 0.0 ? elems[row] * ((rank_ms[p] - fixed) / totals[p]) : (rank_ms[p] - fixed) / (double)(e - b);
        }
    }
    auto cost = [&](const std::vector<uint32_t>& edges) {
        double worst = 0.0;
        for (uint32_t p = 0; p < R; ++p) {
            double s = 0.0;
            for (uint32_t row = edges[p]; row < edges[p + 1u]; ++row) s += weights[row];
            worst = std::max(worst, s);
        }
        return fixed + worst;
    };
    const std::vector<uint32_t> next = gs::balanced_edges(weights, R);
    // hysteresis: the bands (and the hipGraphs captured for them) move only for a predicted gain of 3 % on the slowest rank
    if (next == c->dist_edges || !(cost(next) <= (1.0 - 0.03) * cost(c->dist_edges))) return GS_OK;
    c->dist_edges = next;
    if (moved_out) *moved_out = 1u;
    return apply_band(c);
}

int gs_dist_destroy(gs_ctx* c) {
    if (!c) return GS_ERR_INVALID;
    if (!c->dist_comm) return GS_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    gsi_dist_free_buffers(c);
    for (int k = 0; k < 2; ++k)
        for (hipEvent_t* ev : {&c->dist_begin[k], &c->dist_rendered[k], &c->dist_done[k]}) { if (*ev) (void)hipEventDestroy(*ev); *ev = nullptr; }
    if (c->dist_stream) { (void)hipStreamDestroy(c->dist_stream); c->dist_stream = nullptr; }
    Rccl& r = rccl();
    const RcclResult rc = r.handle ? r.CommDestroy(c->dist_comm) : 0;
    c->dist_comm = nullptr;
    c->dist_rank = 0;
    c->dist_world = 1;
    if (rc != 0) return fail(c, GS_ERR_HIP, std::string("gs_dist_destroy: ") + r.GetErrorString(rc));
    return GS_OK;
}

}  // extern "C"
