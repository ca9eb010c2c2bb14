// gsplat.hpp -- header-only C++17 convenience layer over the C-ABI of gsplat.h, shaped like the
// reference's Renderer (Engine/Graphics/Renderer.h:154-161: init / initForScene / draw / cleanup) so
// that a maintainer of the reference can swap the class in.  No state beyond the
// gs_ctx handle; errors surface as status codes + lastError(), never as exceptions or aborts
// (the reference logs and continues, Engine/Dev/Log.cpp:40-43).
#pragma once

#include "gsplat.h"

#include <cstdint>
#include <string>
#include <vector>

namespace gsplat {

struct Timings : gs_timings {};   // init_sort_list_ms, radix_sort_ms, find_ranges_ms, render_ms, total_ms

class Renderer {
public:
    // Renderer.h:142-143: the averages skip the first WARMUP frames and are final after FRAMES more (the reference hard-codes
    // both; here they are constructor arguments so that a test or a short benchmark can use 3 + 5)
    static constexpr uint32_t WAIT_ELAPSED_WARMUP_FRAMES_FOR_AVG = 1000;
    static constexpr uint32_t WAIT_ELAPSED_FRAMES_FOR_AVG = 1000;

    Renderer(uint32_t width, uint32_t height, uint32_t warmupFramesForAvg = WAIT_ELAPSED_WARMUP_FRAMES_FOR_AVG,
             uint32_t framesForAvg = WAIT_ELAPSED_FRAMES_FOR_AVG)
        : width_(width), height_(height), warmupFrames_(warmupFramesForAvg), avgFrames_(framesForAvg) {}
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    ~Renderer() { cleanup(); }

    // Renderer::init (Renderer.cpp:688-694).  cfg == nullptr: the reference's constants, with the GPU timestamps on
    // (this class keeps the RECORD_GPU_TIMES averages; gs_default_config leaves them off like GfxSettings.h:7).
    int init(const gs_config* cfg = nullptr) {
        const int rc_clean = cleanup();
        if (rc_clean != GS_OK) return rc_clean;
        if (gs_api_version() != GS_API_VERSION) {   // the structs below are laid out as THIS header declares them
            error_ = "libgsplat_hip.so has API version " + std::to_string(gs_api_version()) + ", this header " +
                     std::to_string(GS_API_VERSION);
            return GS_ERR_INVALID;
        }
        gs_config def;
        if (!cfg) { gs_default_config(&def); def.record_timings = 1; cfg = &def; }
        const int rc = gs_create(cfg, &ctx_);
        if (rc != GS_OK) error_ = gs_last_error(nullptr);
        return rc;
    }

    // Renderer::initForScene (Renderer.cpp:712-756): records = ResourceManager::getGaussians(), 336 B each.
    int initForScene(const void* gaussianRecords, uint32_t numGaussians) {
        int rc = gs_upload_gaussians(ctx_, gaussianRecords, numGaussians);
        if (rc == GS_OK) rc = gs_set_resolution(ctx_, width_, height_);
        if (rc != GS_OK) error_ = gs_last_error(ctx_);
        resetAverages();
        return rc;
    }
    // Frame slot (GfxSettings::FRAMES_IN_FLIGHT, GfxSettings.h:15): render the scene `owner` uploaded, with this
    // renderer's own per-frame buffers and stream.  The arrays are reference-counted inside the library, so the two
    // renderers may be cleaned up or destructed in any order.
    int initForSceneSharedWith(Renderer& owner) {
        int rc = gs_share_scene(ctx_, owner.ctx_);
        if (rc == GS_OK) rc = gs_set_resolution(ctx_, width_, height_);
        if (rc != GS_OK) error_ = gs_last_error(ctx_);
        resetAverages();
        return rc;
    }
    int initForScenePly(const std::string& path) {   // ResourceManager::loadGaussians + initForScene
        int rc = gs_load_ply(ctx_, path.c_str());
        if (rc == GS_OK) rc = gs_set_resolution(ctx_, width_, height_);
        if (rc != GS_OK) error_ = rc == GS_ERR_IO || rc == GS_ERR_FORMAT ? gs_ply_last_error() : gs_last_error(ctx_);
        resetAverages();
        return rc;
    }

    // Renderer::draw (Renderer.cpp:297-515).  view/proj: column-major float[16] (glm::mat4 memory);
    // shMode 0/1/2 (Camera.h:7-12); rgbaOut: height*width*4 bytes, top row first.
    int draw(const float* view, const float* proj, const float* camPos, uint32_t shMode, uint8_t* rgbaOut) {
        return frameDone(gs_render(ctx_, view, proj, camPos, shMode, rgbaOut));
    }
    // The same frame without the sink: the image stays in HBM -- rgbaOutDevice (a device pointer of height*width*4 bytes) or,
    // with nullptr, the library's own framebuffer (gs_debug_read(GS_BUF_IMAGE)) -- like the reference's frame, which ends
    // in the swapchain image and is never copied to the host.
    int drawDevice(const float* view, const float* proj, const float* camPos, uint32_t shMode, void* rgbaOutDevice = nullptr) {
        return frameDone(gs_render_device(ctx_, view, proj, camPos, shMode, rgbaOutDevice));
    }

    // The frame on R GPUs, one process per GPU (no reference counterpart; INTEGRATION.md "N GPUs from the same C++ host"):
    // init(cfg with device_ordinal = rank) -> initDist(id from distUniqueId() on one rank, rank, world) -> initForScene ->
    // shardRows -> drawSharded on every rank every frame; rank 0 gets the whole frame in rgbaOut, the others pass nullptr.
    static int distUniqueId(void* id128) { return gs_dist_unique_id(id128); }
    int initDist(const void* id128, int rank, int world) {
        const int rc = gs_dist_init(ctx_, id128, rank, world);
        if (rc != GS_OK) error_ = gs_last_error(ctx_);
        return rc;
    }
    // dealing: GS_ROWS_CONTIGUOUS / GS_ROWS_INTERLEAVED / GS_ROWS_BALANCED (bool converts: false / true = the first two)
    int shardRows(uint32_t dealing = GS_ROWS_CONTIGUOUS) {
        const int rc = gs_dist_shard_rows(ctx_, dealing);
        if (rc != GS_OK) error_ = gs_last_error(ctx_);
        return rc;
    }
    // synchronous: frame + exchange + copy to rgbaOut (rank 0); the averages take this rank's own rows
    int drawSharded(const float* view, const float* proj, const float* camPos, uint32_t shMode, uint8_t* rgbaOut) {
        return frameDone(gs_render_sharded(ctx_, view, proj, camPos, shMode, rgbaOut));
    }
    // two frames in flight: returns after enqueueing; the assembled frame stays in rank 0's HBM (shardedFrame / shardedRead;
    // which = 0: the frame just enqueued, 1: the one before).  No timings (nothing waits).
    int drawShardedAsync(const float* view, const float* proj, const float* camPos, uint32_t shMode) {
        const int rc = gs_render_sharded_async(ctx_, view, proj, camPos, shMode);
        if (rc < 0) error_ = gs_last_error(ctx_);
        return rc;
    }
    int shardedFrame(uint32_t which, void** frameDevice) {
        const int rc = gs_sharded_frame(ctx_, which, frameDevice);
        if (rc < 0) error_ = gs_last_error(ctx_);
        return rc;
    }
    int shardedRead(uint32_t which, uint8_t* rgbaOut) {
        const int rc = gs_sharded_read(ctx_, which, rgbaOut);
        if (rc < 0) error_ = gs_last_error(ctx_);
        return rc;
    }
    // GS_ROWS_BALANCED, collective: move the band edges towards equal cost (gsplat.h); *moved = the edges changed
    int rebalance(bool* moved = nullptr) {
        uint32_t m = 0;
        const int rc = gs_dist_rebalance(ctx_, &m);
        if (rc < 0) error_ = gs_last_error(ctx_);
        if (moved) *moved = m != 0u;
        return rc;
    }

    // Renderer::cleanup (Renderer.cpp:230-270).  gs_destroy always frees the context (gsplat.h), so the handle is
    // dropped before the call and never touched afterwards.
    int cleanup() {
        gs_ctx* c = ctx_;
        ctx_ = nullptr;
        return c ? gs_destroy(c) : GS_OK;
    }

    // RECORD_CPU_TIMES figures of the last draw (Renderer.cpp:399-456)
    gs_host_timings hostTimings() const { gs_host_timings t{}; if (ctx_) gs_get_host_timings(ctx_, &t); return t; }

    const gs_timings& lastTimings() const { return last_; }
    // avgInitSortListMs, avgSortMs, avgFindRangesMs, avgRenderGaussiansMs, avgTotalGpuTimeMs (Renderer.h): running means
    // over every frame after the warm-up (Renderer.cpp:477-488); averagesComplete() is where the reference reports them
    // (ALERT_FINAL_AVERAGE, Renderer.cpp:500-510: warm-up + framesForAvg frames have elapsed)
    const double* averages() const { return avg_; }
    // avgWaitForFenceMs, avgRecordCommandBufferMs, avgPresentMs, avgCpuFrameTimeMs (RECORD_CPU_TIMES, Renderer.cpp:399-456)
    const double* hostAverages() const { return havg_; }
    bool averagesComplete() const { return elapsedFrames_ >= (uint64_t)warmupFrames_ + avgFrames_; }
    uint64_t elapsedFrames() const { return elapsedFrames_; }
    const std::string& lastError() const { return error_; }
    gs_ctx* handle() const { return ctx_; }

    static uint32_t getCeilPowTwo(uint32_t x) { uint32_t n = 1; while (n < x) n *= 2; return n; }   // Renderer.cpp:703-710
    uint32_t getNumTiles() const { return ((width_ + 15) / 16) * ((height_ + 15) / 16); }            // Renderer.cpp:696-701

private:
    void resetAverages() {
        elapsedFrames_ = 0;
        for (double& a : avg_) a = 0.0;
        for (double& a : havg_) a = 0.0;
    }
    // Renderer.cpp:458-497: after the frame has been waited for, read the timestamps and fold them into the running means
    int frameDone(int rc) {
        if (rc < 0) { error_ = gs_last_error(ctx_); return rc; }
        gs_get_timings(ctx_, &last_);
        if (elapsedFrames_ >= warmupFrames_) {
            const double t = 1.0 / double(elapsedFrames_ - warmupFrames_ + 1);
            const double v[5] = {last_.init_sort_list_ms, last_.radix_sort_ms, last_.find_ranges_ms, last_.render_ms, last_.total_ms};
            for (int i = 0; i < 5; ++i) avg_[i] = (1.0 - t) * avg_[i] + t * v[i];
            const gs_host_timings h = hostTimings();
            const double hv[4] = {h.wait_ms, h.record_ms, h.present_ms, h.cpu_frame_ms};
            for (int i = 0; i < 4; ++i) havg_[i] = (1.0 - t) * havg_[i] + t * hv[i];
        }
        ++elapsedFrames_;
        return rc;
    }

    gs_ctx* ctx_ = nullptr;
    uint32_t width_, height_;
    uint32_t warmupFrames_, avgFrames_;
    gs_timings last_{};
    double avg_[5] = {0, 0, 0, 0, 0};
    double havg_[4] = {0, 0, 0, 0};
    uint64_t elapsedFrames_ = 0;
    std::string error_;
};

}  // namespace gsplat
