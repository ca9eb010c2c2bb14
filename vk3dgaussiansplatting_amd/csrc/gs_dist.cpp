// gs_dist.cpp -- the one exchange step of a multi-GPU frame behind the C-ABI: a gather of RGBA8 tile strips to the root
// rank over RCCL (xGMI inside a node).  No reference counterpart (the reference has no collective);
// frames shard by screen-tile rows (gs_set_tile_rows / gs_set_tile_rows_interleaved), the gaussians are
// replicated, and nothing but the finished strips crosses GPUs.
//
// One process per GPU.  A gather of equal strips to one root is R - 1 point-to-point transfers, each over the peer's own
// xGMI link to the root (7 links x ~153 GB/s per GPU), so it is written as exactly that -- grouped ncclSend / ncclRecv
// on the context's stream -- rather than as a ring collective: at 4K a strip is 4.1 MB, ~30 us on its link.
//
// RCCL is bound at gs_dist_init (dlopen of librccl.so.1), not at link time: a process that never shards a frame does
// not map RCCL at all, and a process that already holds an RCCL (PyTorch-ROCm wheels bundle one under the same SONAME)
// gets THAT copy instead of a second one -- the same one-runtime-per-process rule INTEGRATION.md describes for
// libamdhip64.
#include "gs_ctx.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
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

static_assert(sizeof(ncclUniqueId) == GS_DIST_UNIQUE_ID_BYTES, "GS_DIST_UNIQUE_ID_BYTES is sizeof(ncclUniqueId)");

}  // namespace

extern "C" {

int gs_dist_unique_id(void* id_out) {
    if (!id_out) return GS_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.handle) { gsi_set_create_error("gs_dist_unique_id: " + r.error); return GS_ERR_HIP; }   // gs_last_error(NULL)
    ncclUniqueId id;
    const ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) { gsi_set_create_error(std::string("gs_dist_unique_id: ") + r.GetErrorString(rc)); return GS_ERR_HIP; }
    std::memcpy(id_out, &id, sizeof(id));
    return GS_OK;
}

int gs_dist_init(gs_ctx* c, const void* unique_id, int rank, int world) {
    if (!c) return GS_ERR_INVALID;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(c, GS_ERR_INVALID, "gs_dist_init: bad rank / world / id");
    if (c->dist_comm) return fail(c, GS_ERR_INVALID, "gs_dist_init: already initialised (gs_dist_destroy first)");
    Rccl& r = rccl();
    if (!r.handle) return fail(c, GS_ERR_HIP, "gs_dist_init: " + r.error);
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, GS_ERR_HIP, "gs_dist_init: hipSetDevice failed");
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r.CommInitRank(&comm, world, id, rank);
    if (rc != ncclSuccess) return fail(c, GS_ERR_HIP, std::string("gs_dist_init: ncclCommInitRank: ") + r.GetErrorString(rc));
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
    ncclComm_t comm = (ncclComm_t)c->dist_comm;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, GS_ERR_HIP, "gs_gather_strips: hipSetDevice failed");
    ncclResult_t rc = ncclSuccess;
    if (c->dist_rank == root) {
        uint8_t* dst = static_cast<uint8_t*>(gathered_dev);
        // the root's own strip never leaves the GPU; the peers' strips arrive over their own links, all in one group
        if (hipMemcpyAsync(dst + (size_t)root * bytes, strip_dev, bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
            return fail(c, GS_ERR_HIP, "gs_gather_strips: hipMemcpyAsync failed");
        if (c->dist_world > 1) {
            rc = r.GroupStart();
            for (int p = 0; p < c->dist_world && rc == ncclSuccess; ++p)
                if (p != root) rc = r.Recv(dst + (size_t)p * bytes, bytes, ncclUint8, p, comm, c->stream);
            const ncclResult_t rc_end = r.GroupEnd();
            if (rc == ncclSuccess) rc = rc_end;
        }
    } else {
        rc = r.Send(strip_dev, bytes, ncclUint8, root, comm, c->stream);
    }
    if (rc != ncclSuccess) return fail(c, GS_ERR_HIP, std::string("gs_gather_strips: ") + r.GetErrorString(rc));
    return GS_OK;
}

int gs_dist_shard_rows(gs_ctx* c, uint32_t interleaved) {
    if (!c) return GS_ERR_INVALID;
    if (!c->dist_comm) return fail(c, GS_ERR_INVALID, "gs_dist_shard_rows: gs_dist_init not called");
    if (!c->capacity) return fail(c, GS_ERR_NO_SCENE, "gs_dist_shard_rows: gs_set_resolution not called");
    const uint32_t R = (uint32_t)c->dist_world, r = (uint32_t)c->dist_rank;
    const uint32_t per = (c->grid_h + R - 1u) / R;          // tile rows of a (padded) strip: the same on every rank
    int rc;
    if (interleaved) rc = gs_set_tile_rows_interleaved(c, r, R, 1u);
    else rc = gs_set_tile_rows(c, std::min(r * per, c->grid_h), std::min((r + 1u) * per, c->grid_h));
    if (rc != GS_OK) return rc;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, GS_ERR_HIP, "gs_dist_shard_rows: hipSetDevice failed");
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void** p : {&c->dist_strip, &c->dist_gathered, &c->dist_image}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    c->dist_interleaved = interleaved != 0u;
    c->dist_strip_bytes = (size_t)per * 16u * c->width * 4u;
    hipError_t e = hipMalloc(&c->dist_strip, c->dist_strip_bytes);
    if (e == hipSuccess) e = hipMemset(c->dist_strip, 0, c->dist_strip_bytes);
    if (e == hipSuccess && r == 0u) e = hipMalloc(&c->dist_gathered, c->dist_strip_bytes * R);
    if (e == hipSuccess && r == 0u && interleaved) e = hipMalloc(&c->dist_image, c->dist_strip_bytes * R);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_dist_shard_rows: ") + hipGetErrorString(e));
    return GS_OK;
}

int gs_render_sharded(gs_ctx* c, const float view[16], const float proj[16], const float cam_pos[3], uint32_t sh_mode,
                      uint8_t* rgba_out) {
    if (!c) return GS_ERR_INVALID;
    if (!c->dist_comm || !c->dist_strip) return fail(c, GS_ERR_INVALID, "gs_render_sharded: gs_dist_init + gs_dist_shard_rows first");
    const bool root = c->dist_rank == 0;
    if (root && !rgba_out) return fail(c, GS_ERR_INVALID, "gs_render_sharded: rank 0 needs rgba_out");
    // a contiguous band addresses the real rows of the frame: hand the frame a pointer shifted up by the band's first
    // row, so that the band lands at the top of the strip; interleaved rows are written packed (compact_output)
    uint8_t* target = static_cast<uint8_t*>(c->dist_strip);
    if (!c->dist_interleaved) target -= (size_t)c->row_begin * 16u * c->width * 4u;
    int rc = c->rows_owned ? gs_render_device_async(c, view, proj, cam_pos, sh_mode, target) : GS_OK;
    // a rank whose frame failed still takes part in the exchange (with whatever its strip holds) -- leaving now would leave
    // the other ranks waiting in theirs -- and reports its error afterwards
    const std::string render_error = rc < 0 ? c->last_error : std::string();
    const int rc_render = rc;
    rc = gs_gather_strips(c, c->dist_strip, c->dist_gathered, c->dist_strip_bytes, 0);
    if (rc < 0) return rc;
    if (rc_render < 0) { (void)hipStreamSynchronize(c->stream); return fail(c, rc_render, render_error); }
    rc = rc_render;
    const size_t frame_bytes = (size_t)c->width * c->height * 4u;
    hipError_t e = hipSuccess;
    if (root && c->dist_interleaved) {
        // strip r, block k (16 pixel rows) -> tile row k * R + r of the frame: one strided copy per rank
        const size_t block = (size_t)16u * c->width * 4u;
        const uint32_t R = (uint32_t)c->dist_world;
        for (uint32_t r = 0; r < R && e == hipSuccess; ++r) {
            const uint32_t owned = c->grid_h > r ? (c->grid_h - r + R - 1u) / R : 0u;
            if (owned)
                e = hipMemcpy2DAsync(static_cast<uint8_t*>(c->dist_image) + (size_t)r * block, block * R,
                                     static_cast<uint8_t*>(c->dist_gathered) + (size_t)r * c->dist_strip_bytes, block, block, owned,
                                     hipMemcpyDeviceToDevice, c->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(rgba_out, c->dist_image, frame_bytes, hipMemcpyDeviceToHost, c->stream);
    } else if (root) {
        // contiguous bands of `per` tile rows each: the gathered strips ARE the frame, top to bottom (+ padding)
        e = hipMemcpyAsync(rgba_out, c->dist_gathered, frame_bytes, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_render_sharded: ") + hipGetErrorString(e));
    // gs_get_timings: this rank's own rows (the gather is not part of the reference's buckets); GS_WARN_OVERFLOW as in gs_render
    return c->rows_owned ? gsi_finish_frame(c) : rc;
}

int gs_dist_destroy(gs_ctx* c) {
    if (!c) return GS_ERR_INVALID;
    if (!c->dist_comm) return GS_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void** p : {&c->dist_strip, &c->dist_gathered, &c->dist_image}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    Rccl& r = rccl();
    const ncclResult_t rc = r.handle ? r.CommDestroy((ncclComm_t)c->dist_comm) : ncclSuccess;
    c->dist_comm = nullptr;
    c->dist_rank = 0;
    c->dist_world = 1;
    if (rc != ncclSuccess) return fail(c, GS_ERR_HIP, std::string("gs_dist_destroy: ") + r.GetErrorString(rc));
    return GS_OK;
}

}  // extern "C"
