// gs_api.cpp -- host side of libgsplat_hip.so: the C-ABI of include/gsplat.h over the HIP kernels.
// Mirrors the reference's frame orchestration (Engine/Graphics/Renderer.cpp, Subrenderer.cpp,
// Sort/RadixSort.cpp) with HIP streams/events in place of Vulkan command buffers/timestamps.
// There is NO CPU fallback: without a GPU every entry point that computes fails with
// GS_ERR_NO_DEVICE / GS_ERR_HIP.
#include "../../include/gsplat.h"
#include "gs_internal.h"
#include "gs_ctx.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace gs;

static thread_local std::string g_create_error;
void gsi_set_create_error(const std::string& msg) { g_create_error = msg; }

namespace {

int fail(gs_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->last_error = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail((ctx), GS_ERR_HIP,                                                       \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                      \
    } while (0)

template <typename T>
void free_dev(T*& p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

void free_scene(gs_ctx* c) {
    if (c->shared) {                  // drop this context's reference; the last one frees the arrays
        if (c->shared->refs.fetch_sub(1) == 1) {
            SceneBuffers& b = c->shared->b;
            free_dev(b.pos); free_dev(b.scale); free_dev(b.rot);
            free_dev(b.sh); free_dev(b.opacity); free_dev(b.sig2); free_dev(b.block_bounds);
            delete c->shared;
        }
        c->shared = nullptr;
    }
    c->scene = SceneBuffers{};
    free_dev(c->scratch.raster); free_dev(c->scratch.depth_key); free_dev(c->scratch.tiles_touched);
    free_dev(c->scratch.extents); free_dev(c->scratch.block_sums); free_dev(c->scratch.block_offsets);
    free_dev(c->scratch.help_list); free_dev(c->scratch.help_count); free_dev(c->scratch.help_slot);
    free_dev(c->scratch.wave_wrote);
    free_dev(c->scratch.band_list);
    free_dev(c->scratch.block_flags); free_dev(c->scratch.flag_offsets);
    free_dev(c->scratch.sorted_sums); free_dev(c->scratch.aux_params);
    c->n = 0;
}

void free_sort(SortBuffers& s) {
    for (int k = 0; k < 2; ++k) { free_dev(s.lo[k]); free_dev(s.hi[k]); free_dev(s.id[k]); }
    free_dev(s.table); free_dev(s.seg_sum); free_dev(s.params); free_dev(s.coarse);
    free_dev(s.fed[0]); s.fed[1] = s.fed[2] = nullptr;      // one allocation, three sets
}

void drop_sort_graph(gs_ctx* c) {
    if (c->sort_graph) { (void)hipGraphExecDestroy(c->sort_graph); c->sort_graph = nullptr; }
    if (c->presort_graph) { (void)hipGraphExecDestroy(c->presort_graph); c->presort_graph = nullptr; }
    if (c->chain_graph) { (void)hipGraphExecDestroy(c->chain_graph); c->chain_graph = nullptr; }
    c->sort_graph_failed = false;
}

void free_resolution(gs_ctx* c) {
    drop_sort_graph(c);
    free_sort(c->sort);
    free_dev(c->ranges); free_dev(c->tile_order); free_dev(c->framebuffer);
    // the strips of a sharded frame are sized by the resolution: gs_dist_shard_rows must be called again
    gsi_dist_free_buffers(c);
    c->capacity = 0; c->width = c->height = 0;
    c->have_frame = false;
    if (c->elems_note) *(volatile uint32_t*)c->elems_note = 0u;     // (free_resolution's callers have waited for the stream)
}

// Renderer.cpp:703-710
uint32_t ceil_pow2(uint32_t x) { uint32_t v = 1; while (v < x) v *= 2; return v; }

// RadixSort.cpp:7-16 + 203-204
uint32_t num_sort_bits_for(uint32_t num_tiles) {
    uint32_t x = num_tiles - 1u, bits = 0;
    for (int i = 31; i >= 0; --i) if ((x >> i) & 1u) { bits = (uint32_t)i + 1u; break; }
    return ((32u + bits + kRadixBits - 1u) / kRadixBits) * kRadixBits;
}

inline bool sorts_splat_first(uint32_t algo) { return algo == GS_SORT_RADIX4_SPLAT_FIRST || algo == GS_SORT_RADIX8_SPLAT_FIRST; }
inline uint32_t digit_bits_of(uint32_t algo) { return algo == GS_SORT_RADIX8 || algo == GS_SORT_RADIX8_SPLAT_FIRST ? 8u : (uint32_t)kRadixBits; }

int alloc_sort(gs_ctx* ctx, SortBuffers& s, uint32_t capacity, uint32_t digit_bits) {
    const size_t bytes = (size_t)capacity * sizeof(uint32_t);
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(ctx, hipMalloc((void**)&s.lo[k], bytes));
        HIP_TRY(ctx, hipMalloc((void**)&s.hi[k], bytes));
        HIP_TRY(ctx, hipMalloc((void**)&s.id[k], bytes));
    }
    if (digit_bits == 8u) {   // gs_sort8.hip: [groups][256] counts; segment counts + their scan
        const uint32_t max_groups = (capacity + kSort8TileSmall - 1) / kSort8TileSmall;   // the smaller of the two group sizes
        HIP_TRY(ctx, hipMalloc((void**)&s.table, (size_t)kBins8 * max_groups * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&s.seg_sum, (size_t)2 * kBins8 * kSegments * sizeof(uint32_t)));
    } else {
        const uint32_t max_groups = (capacity + kSortTile - 1) / kSortTile;
        HIP_TRY(ctx, hipMalloc((void**)&s.table, (size_t)kBins * max_groups * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&s.seg_sum, (size_t)kBins * kSegments * sizeof(uint32_t)));
        // fed counts: three rotating sets of [groups][16] rows (k_scatter<.., FED>); every row a pass reads was written or
        // cleared earlier in the same sort, so no initial clear
        const size_t set_words = (size_t)kBins * max_groups;
        HIP_TRY(ctx, hipMalloc((void**)&s.fed[0], 3 * set_words * sizeof(uint32_t)));
        s.fed[1] = s.fed[0] + set_words;
        s.fed[2] = s.fed[1] + set_words;
    }
    s.digit_bits = digit_bits;
    HIP_TRY(ctx, hipMalloc((void**)&s.params, sizeof(SortParams)));
    HIP_TRY(ctx, hipMemset(s.params, 0, sizeof(SortParams)));
    HIP_TRY(ctx, hipMalloc((void**)&s.coarse, (size_t)kMaxSortPasses * kBins * kCoarse * sizeof(uint32_t)));
    return GS_OK;
}

// the stand-alone 4-bit sorter knows its element count: fed counts (gs_sort.hip) for short lists
bool fed_for(const gs_ctx* c, uint32_t n) {
    if (digit_bits_of(c->cfg.sort_algorithm) != (uint32_t)kRadixBits) return false;
    return c->cfg.count_launches == GS_COUNT_FED ||
           (c->cfg.count_launches == GS_COUNT_AUTO && n <= kFedMaxGroups * (uint32_t)kSortTile);
}

int check_launch(gs_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, GS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return GS_OK;
}

// Renderer::recordCommandBuffer (Renderer.cpp:540-629): stage order + the 7 timestamp points.
int enqueue_frame(gs_ctx* c, const float* view, const float* proj, const float* cam_pos,
                  uint32_t sh_mode, uint8_t* out_dev) {
    if (!c->n) return fail(c, GS_ERR_NO_SCENE, "gs_render: no gaussians uploaded");
    if (!c->capacity) return fail(c, GS_ERR_NO_SCENE, "gs_render: gs_set_resolution not called");
    if (!view || !proj || !cam_pos) return fail(c, GS_ERR_INVALID, "gs_render: null camera argument");
    if (sh_mode > 2u) return fail(c, GS_ERR_INVALID, "gs_render: sh_mode must be 0, 1 or 2");
    FrameParams fp = make_frame_params(c, view, proj, cam_pos, sh_mode);
    if (!out_dev) fp.compact_out = 0u;   // the internal framebuffer is always a whole frame in real rows
    const bool tm = c->cfg.record_timings != 0;
    hipStream_t st = c->stream;

    c->unsorted_valid = false;
    if (tm) { HIP_TRY(c, hipEventRecord(c->ev[0], st)); HIP_TRY(c, hipEventRecord(c->ev[1], st)); }
    // computeInitSortList (Subrenderer.cpp:37-170): per-frame resets, then the dispatch.  Only the
    // ranges need clearing here: the 0xFF sentinel fill of both lists (Subrenderer.cpp:42-46,
    // RadixSort.cpp:676-692) is unobservable once every later stage runs over E instead of C.
    // (the ranges and the sort's coarse totals are cleared inside k_scan_blocks: no fill launches in a frame)
    fp.parity = (c->emit_parity ^= 1u);
    c->last_fp = fp;
    const bool splat_first = sorts_splat_first(c->cfg.sort_algorithm);
    const uint32_t digit = digit_bits_of(c->cfg.sort_algorithm);
    fp.splat_first = splat_first ? 1u : 0u;
    const bool bucket = c->cfg.sort_algorithm == GS_SORT_TILE_BUCKET;
    const bool per_pass_events = c->cfg.record_timings >= 2;
    const float tile_share = c->grid_h ? (float)c->rows_owned / (float)c->grid_h : 1.0f;
    const bool ordered = c->cfg.tile_order == GS_TILE_ORDER_LONGEST_FIRST;
    // one capture-or-replay of a run of radix passes (nothing executes during capture)
    auto radix_passes = [&](hipGraphExec_t& exec, int& result, auto&& launch) -> int {
        if (!per_pass_events && !exec && !c->sort_graph_failed) {
            hipGraph_t graph = nullptr;
            bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                result = launch(nullptr);
                ok = hipStreamEndCapture(st, &graph) == hipSuccess && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (!ok) { exec = nullptr; c->sort_graph_failed = true; (void)hipGetLastError(); }
        }
        if (!per_pass_events && exec) {
            if (hipGraphLaunch(exec, st) != hipSuccess) return -1;
            return result;
        }
        return launch(per_pass_events ? c->scatter_ev : nullptr);
    };
    launch_project(fp, c->scene, c->scratch, st);
    launch_scan_blocks(fp, c->scratch, c->sort.params, c->ranges, c->sort.coarse, st);
    bool ranges_done = false;
    if (splat_first) {
        // GS_SORT_RADIX4_SPLAT_FIRST: the eight passes over the depth word run on the list of emitting splats, the emit
        // walks that list, the tile-word passes finish.  InitSortList = project + lists + emit, RadixSort = all passes.
        // Nothing between the first scan and RenderGaussians depends on the camera: those kernels take a FrameParams
        // without it (and with a fixed helper counter, cleared by the first kernel of the chain), so their arguments
        // never change and the whole chain -- FindRanges included -- replays as ONE graph when no timers are asked for.
        FrameParams fps = fp;
        std::memset(fps.view, 0, sizeof(fps.view)); std::memset(fps.proj, 0, sizeof(fps.proj));
        std::memset(fps.cam_pos, 0, sizeof(fps.cam_pos));
        fps.sh_mode = 0u; fps.w_norm2 = 0.0f; fps.compact_out = 0u;
        fps.parity = 0u;
        auto depth_passes = [&](hipEvent_t* evs) {
            // the frame's own depth passes (shrinking depth words, payload as wide as the tile ids) over the splat list
            return launch_radix_sort(c->sort, c->n, 32u, st, evs, 0u, true, c->hi16, 1.0f, /*start*/ 1, /*coarse_pass*/ 0,
                                     c->scratch.aux_params, digit);
        };
        auto tile_passes = [&](hipEvent_t* evs) {
            return launch_radix_sort(c->sort, c->capacity, c->band_sort_bits, st, evs ? evs + 16 : nullptr, 32u, true, c->hi16,
                                     tile_share, /*start*/ 0, /*coarse_pass*/ 8, nullptr, digit);
        };
        if (!tm) {
            const int sorted = radix_passes(c->chain_graph, c->chain_result, [&](hipEvent_t*) {
                launch_splat_list(fps, c->scratch, c->sort, st);
                const int presorted = depth_passes(nullptr);
                launch_gather_sorted(fps, c->scratch, c->sort, presorted, st);
                launch_emit_sorted(fps, c->scratch, c->sort, presorted, st);
                const int si = tile_passes(nullptr);
                launch_find_ranges(fps, c->sort.hi[si], c->sort.params, c->ranges, st);
                if (ordered) launch_tile_order(fps, c->ranges, c->tile_order, st);
                return si;
            });
            if (sorted < 0) return fail(c, GS_ERR_HIP, "gs_render: the radix passes could not be enqueued (hipGraphLaunch failed or sort buffers of another digit width)");
            c->sorted_index = sorted;
            ranges_done = true;
        } else {
            launch_splat_list(fps, c->scratch, c->sort, st);
            if (int r = check_launch(c, "InitSortList")) return r;
            HIP_TRY(c, hipEventRecord(c->pre_ev[0], st));
            const int presorted = radix_passes(c->presort_graph, c->presort_result, depth_passes);
            if (presorted < 0) return fail(c, GS_ERR_HIP, "gs_render: the radix passes could not be enqueued (hipGraphLaunch failed or sort buffers of another digit width)");
            if (int r = check_launch(c, "RadixSort")) return r;
            HIP_TRY(c, hipEventRecord(c->pre_ev[1], st));
            launch_gather_sorted(fps, c->scratch, c->sort, presorted, st);
            launch_emit_sorted(fps, c->scratch, c->sort, presorted, st);
            if (int r = check_launch(c, "InitSortList")) return r;
            HIP_TRY(c, hipEventRecord(c->pre_ev[2], st));
            const int sorted = radix_passes(c->sort_graph, c->sort_graph_result, tile_passes);
            if (sorted < 0) return fail(c, GS_ERR_HIP, "gs_render: the radix passes could not be enqueued (hipGraphLaunch failed or sort buffers of another digit width)");
            c->sorted_index = sorted;
        }
    } else {
        launch_emit(fp, c->scratch, c->sort, st);
        if (int r = check_launch(c, "InitSortList")) return r;
        if (tm) HIP_TRY(c, hipEventRecord(c->ev[2], st));
        // gpuSort->computeSort (RadixSort.cpp:207-653).  The passes' arguments are fixed once resolution and tile rows are:
        // captured once, replayed as a hipGraph.  Without timers FindRanges (same property) rides in the same graph.
        // fed counts (gs_sort.hip): for the contractual sorter, when the list is short.  The host enqueues a frame without
        // knowing its element count, so GS_COUNT_AUTO goes by the count of a recent frame, which k_scan_blocks leaves in
        // a pinned host word (nothing waits for it; none yet: a Count launch per pass); a change of mind re-captures
        // the graph.  Speed only: the sorted list is the same either way and at any length.
        bool fed = false;
        if (c->cfg.sort_algorithm == GS_SORT_RADIX4) {
            const uint32_t limit = kFedMaxGroups * (uint32_t)kSortTile;
            const uint32_t note = c->elems_note ? *(volatile const uint32_t*)c->elems_note : 0u;   // count + 1 of a recent frame
            fed = c->cfg.count_launches == GS_COUNT_FED ||
                  (c->cfg.count_launches == GS_COUNT_AUTO && note != 0u &&
                   note - 1u <= (c->sort_fed ? limit : limit - limit / 16u));
        }
        if (fed != c->sort_fed) {
            if (c->sort_graph || c->chain_graph) HIP_TRY(c, hipStreamSynchronize(st));   // the graph may still be executing (rare: the mode flips)
            drop_sort_graph(c);
            c->sort_fed = fed;
        }
        auto all_passes = [&](hipEvent_t* evs) {
            return launch_radix_sort(c->sort, c->capacity, c->band_sort_bits, st, evs, bucket ? 32u : 0u, !bucket, c->hi16, tile_share,
                                     0, 0, nullptr, digit, fed);
        };
        if (!tm && !bucket) {
            const int sorted = radix_passes(c->chain_graph, c->chain_result, [&](hipEvent_t*) {
                const int si = all_passes(nullptr);
                launch_find_ranges(fp, c->sort.hi[si], c->sort.params, c->ranges, st);
                if (ordered) launch_tile_order(fp, c->ranges, c->tile_order, st);
                return si;
            });
            if (sorted < 0) return fail(c, GS_ERR_HIP, "gs_render: the radix passes could not be enqueued (hipGraphLaunch failed or sort buffers of another digit width)");
            c->sorted_index = sorted;
            ranges_done = true;
        } else {
            const int sorted = radix_passes(c->sort_graph, c->sort_graph_result, all_passes);
            if (sorted < 0) return fail(c, GS_ERR_HIP, "gs_render: the radix passes could not be enqueued (hipGraphLaunch failed or sort buffers of another digit width)");
            c->sorted_index = sorted;
        }
    }
    c->depth_dropped = !bucket;
    if (int r = check_launch(c, "RadixSort")) return r;
    if (tm) HIP_TRY(c, hipEventRecord(c->ev[3], st));
    // computeRanges (Subrenderer.cpp:172-216)
    if (!ranges_done) {
        launch_find_ranges(fp, c->sort.hi[c->sorted_index], c->sort.params, c->ranges, st);
        if (ordered) launch_tile_order(fp, c->ranges, c->tile_order, st);
    }
    if (int r = check_launch(c, "FindRanges")) return r;
    if (bucket) {
        // second half of the alternative sorter: per-tile depth sort (needs the ranges)
        const int si = c->sorted_index;
        if (tm) HIP_TRY(c, hipEventRecord(c->alt_ev[0], st));
        launch_tile_sort(fp, c->ranges, c->sort.lo[si], c->sort.id[si], c->sort.lo[si ^ 1], c->sort.id[si ^ 1], st,
                         c->helper_stream, c->fork_ev, c->join_ev);
        if (int r = check_launch(c, "TileSort")) return r;
        if (tm) HIP_TRY(c, hipEventRecord(c->alt_ev[1], st));
    }
    if (tm) HIP_TRY(c, hipEventRecord(c->ev[4], st));
    // computeRenderGaussians (Subrenderer.cpp:218-346)
    launch_render(fp, c->scratch.raster, c->sort.id[c->sorted_index], c->ranges, ordered ? c->tile_order : nullptr,
                  out_dev ? out_dev : c->framebuffer, c->cfg.render_mode, c->cfg.render_kernel, st);
    if (int r = check_launch(c, "RenderGaussians")) return r;
    if (tm) { HIP_TRY(c, hipEventRecord(c->ev[5], st)); HIP_TRY(c, hipEventRecord(c->ev[6], st)); }
    c->have_frame = true;
    return GS_OK;
}

double ms_since(HostClock::time_point t0) {
    return std::chrono::duration<double, std::milli>(HostClock::now() - t0).count();
}

// RECORD_CPU_TIMES (Renderer.cpp:299-314, 399-456): "CPU frame time" = entry of this draw - entry of the previous one
void host_frame_begin(gs_ctx* c) {
    const HostClock::time_point now = HostClock::now();
    c->host.cpu_frame_ms = c->have_entry ? (float)std::chrono::duration<double, std::milli>(now - c->last_entry).count() : 0.0f;
    c->last_entry = now;
    c->have_entry = true;
    c->host.wait_ms = 0.0f;
    c->host.present_ms = 0.0f;
}

// Renderer.cpp:458-475: wait, read the timestamps, compute the five buckets.
int finish_frame(gs_ctx* c) {
    const HostClock::time_point t_wait = HostClock::now();
    hipError_t sync_err = hipStreamSynchronize(c->stream);
    c->host.wait_ms = (float)ms_since(t_wait);       // the reference's waitForFences + waitIdle
    HIP_TRY(c, sync_err);
    SortParams sp{};
    HIP_TRY(c, hipMemcpy(&sp, c->sort.params, sizeof(sp), hipMemcpyDeviceToHost));
    gs_timings t{};
    if (c->cfg.record_timings) {
        if (sorts_splat_first(c->cfg.sort_algorithm)) {
            // project + lists | depth passes | gather + emit | tile-word passes
            float a = 0.0f, b = 0.0f, d = 0.0f, e = 0.0f;
            HIP_TRY(c, hipEventElapsedTime(&a, c->ev[1], c->pre_ev[0]));
            HIP_TRY(c, hipEventElapsedTime(&b, c->pre_ev[0], c->pre_ev[1]));
            HIP_TRY(c, hipEventElapsedTime(&d, c->pre_ev[1], c->pre_ev[2]));
            HIP_TRY(c, hipEventElapsedTime(&e, c->pre_ev[2], c->ev[3]));
            t.init_sort_list_ms = a + d;
            t.radix_sort_ms = b + e;
        } else {
            HIP_TRY(c, hipEventElapsedTime(&t.init_sort_list_ms, c->ev[1], c->ev[2]));
            HIP_TRY(c, hipEventElapsedTime(&t.radix_sort_ms, c->ev[2], c->ev[3]));
        }
        HIP_TRY(c, hipEventElapsedTime(&t.find_ranges_ms, c->ev[3], c->ev[4]));
        HIP_TRY(c, hipEventElapsedTime(&t.render_ms, c->ev[4], c->ev[5]));
        HIP_TRY(c, hipEventElapsedTime(&t.total_ms, c->ev[0], c->ev[6]));
        if (c->cfg.sort_algorithm == GS_SORT_TILE_BUCKET) {
            // the per-tile sort runs behind FindRanges: book it under RadixSort, not FindRanges
            float ranges_ms = 0.0f, tile_ms = 0.0f;
            HIP_TRY(c, hipEventElapsedTime(&ranges_ms, c->ev[3], c->alt_ev[0]));
            HIP_TRY(c, hipEventElapsedTime(&tile_ms, c->alt_ev[0], c->alt_ev[1]));
            t.find_ranges_ms = ranges_ms;
            t.radix_sort_ms += tile_ms;
        }
    }
    if (c->cfg.record_timings >= 2 && sorts_splat_first(c->cfg.sort_algorithm)) {
        // "full" = the eight depth passes over the SPLAT list, "tile" = the tile-word passes over the elements
        float sum_pre = 0.0f, sum_tile = 0.0f, bytes_pre = 0.0f;
        const uint32_t digit = digit_bits_of(c->cfg.sort_algorithm), n_pre = 32u / digit;
        const uint32_t n_tile = (c->band_sort_bits - 32u + digit - 1u) / digit;
        for (uint32_t k = 0; k < n_pre; ++k) {
            float ms = 0.0f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->scatter_ev[2 * k], c->scatter_ev[2 * k + 1]));
            sum_pre += ms;
            int lo_in, lo_out;
            scatter_depth_bytes(k * digit, 0u, true, &lo_in, &lo_out, digit);
            bytes_pre += (float)(lo_in + lo_out) + 2.0f * (c->hi16 ? 2.0f : 4.0f) + 8.0f;   // depth + count + splat, r + w
        }
        for (uint32_t k = 0; k < n_tile; ++k) {
            float ms = 0.0f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->scatter_ev[16 + 2 * k], c->scatter_ev[16 + 2 * k + 1]));
            sum_tile += ms;
        }
        t.scatter_ms_avg = sum_pre / (float)n_pre;
        t.scatter_launches = n_pre;
        t.scatter_bytes_per_elem = bytes_pre / (float)n_pre;   // per SPLAT of the list
        t.scatter_tile_ms_avg = n_tile ? sum_tile / (float)n_tile : 0.0f;
        t.scatter_tile_launches = n_tile;
        t.scatter_tile_bytes_per_elem = 2.0f * (c->hi16 ? 2.0f : 4.0f) + 8.0f;
    } else if (c->cfg.record_timings >= 2) {
        const uint32_t first_bit = c->cfg.sort_algorithm == GS_SORT_TILE_BUCKET ? 32u : 0u;
        const uint32_t digit = digit_bits_of(c->cfg.sort_algorithm);
        const uint32_t passes = (c->band_sort_bits - first_bit + digit - 1u) / digit;
        // launches that move all 24 bytes per element (k_scatter<true>) and the tile-word passes of the frame
        // path that leave the depth words behind (k_scatter<false>, 16 bytes per element) are averaged apart
        const bool bucket = c->cfg.sort_algorithm == GS_SORT_TILE_BUCKET;
        float sum_full = 0.0f, sum_tile = 0.0f, bytes_full = 0.0f, bytes_tile = 0.0f;
        uint32_t n_full = 0, n_tile = 0;
        for (uint32_t k = 0; k < passes; ++k) {
            float ms = 0.0f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->scatter_ev[2 * k], c->scatter_ev[2 * k + 1]));
            const uint32_t shift = first_bit + k * digit;
            int lo_in, lo_out;
            scatter_depth_bytes(shift, first_bit, !bucket, &lo_in, &lo_out, digit);
            const float moved = (float)(lo_in + lo_out) + 2.0f * (c->hi16 ? 2.0f : 4.0f) + 8.0f;   // depth + tile + id, r + w
            if (shift < 32u || bucket) { sum_full += ms; ++n_full; bytes_full += moved; }
            else { sum_tile += ms; ++n_tile; bytes_tile += moved; }
        }
        t.scatter_bytes_per_elem = n_full ? bytes_full / (float)n_full : 0.0f;
        t.scatter_tile_bytes_per_elem = n_tile ? bytes_tile / (float)n_tile : 0.0f;
        t.scatter_ms_avg = n_full ? sum_full / (float)n_full : 0.0f;
        t.scatter_launches = n_full;
        t.scatter_tile_ms_avg = n_tile ? sum_tile / (float)n_tile : 0.0f;
        t.scatter_tile_launches = n_tile;
    }
    t.num_sort_elements = sp.num_elems;
    t.overflowed = sp.overflow;
    t.emitted_elements = sp.counter;
    c->timings = t;
    return sp.overflow ? GS_WARN_OVERFLOW : GS_OK;
}

} // namespace

// gs_dist.cpp: the buckets and the element count of the frame gs_render_sharded has just waited for (hidden: not an export)
int gsi_finish_frame(gs_ctx* c) { return finish_frame(c); }

extern "C" {

uint32_t gs_api_version(void) { return GS_API_VERSION; }

int gs_runtime_versions(int* hip_build, int* hip_runtime, int* hip_driver) {
    if (hip_build) *hip_build = HIP_VERSION;
    int v = 0;
    if (hip_runtime) { if (hipRuntimeGetVersion(&v) != hipSuccess) return GS_ERR_HIP; *hip_runtime = v; }
    if (hip_driver) { if (hipDriverGetVersion(&v) != hipSuccess) return GS_ERR_HIP; *hip_driver = v; }
    return GS_OK;
}

void gs_default_config(gs_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (uint32_t)sizeof(*cfg);
    cfg->device_ordinal = 0;
    cfg->tile_size = 16;            // Renderer.h:146
    cfg->near_plane = 0.1f;         // Camera.cpp:4
    cfg->far_plane = 100.0f;        // Camera.cpp:5
    cfg->ndc_cull = 1.3f;           // Common.glsl:5
    cfg->in_view_limit = 0.8f;      // Common.glsl:9
    cfg->fov_y = 3.1415f * 0.5f;    // Common.glsl:2
    cfg->sort_algorithm = GS_SORT_RADIX4;
    cfg->render_mode = GS_RENDER_EXACT;
    cfg->record_timings = 0;        // RECORD_GPU_TIMES is commented out in the reference (GfxSettings.h:7)
    cfg->render_kernel = GS_RENDER_KERNEL_AUTO;
    cfg->tile_order = GS_TILE_ORDER_LONGEST_FIRST;
    cfg->count_launches = GS_COUNT_AUTO;
}

int gs_create(const gs_config* cfg_in, gs_ctx** out) {
    if (!out) return fail(nullptr, GS_ERR_INVALID, "gs_create: out is null");
    *out = nullptr;
    gs_config cfg;
    gs_default_config(&cfg);
    if (cfg_in) {
        // struct_size says how much of the struct the caller's header knows: fields beyond it keep their defaults, a
        // struct from a NEWER header than this library is refused instead of being read past what is understood
        if (cfg_in->struct_size < offsetof(gs_config, tile_order) + sizeof(uint32_t) || cfg_in->struct_size > sizeof(gs_config))
            return fail(nullptr, GS_ERR_INVALID, "gs_create: gs_config.struct_size does not match this library (call gs_default_config first; "
                                                 "compare GS_API_VERSION with gs_api_version())");
        std::memcpy(&cfg, cfg_in, cfg_in->struct_size);
        cfg.struct_size = (uint32_t)sizeof(cfg);
    }
    if (cfg.tile_size != 16) return fail(nullptr, GS_ERR_INVALID, "gs_create: only tile_size 16 is supported");
    if (cfg.sort_algorithm > GS_SORT_RADIX8_SPLAT_FIRST) return fail(nullptr, GS_ERR_INVALID, "gs_create: unknown sort_algorithm");
    if (cfg.render_mode > GS_RENDER_FAST) return fail(nullptr, GS_ERR_INVALID, "gs_create: unknown render_mode");
    if (cfg.render_kernel != GS_RENDER_KERNEL_AUTO && cfg.render_kernel != GS_RENDER_KERNEL_WAVE_1PX &&
        cfg.render_kernel != GS_RENDER_KERNEL_WAVE_2PX && cfg.render_kernel != GS_RENDER_KERNEL_WAVE_4PX &&
        cfg.render_kernel != GS_RENDER_KERNEL_WORKGROUP && cfg.render_kernel != GS_RENDER_KERNEL_WORKGROUP_8X8)
        return fail(nullptr, GS_ERR_INVALID, "gs_create: unknown render_kernel");
    if (cfg.tile_order > GS_TILE_ORDER_RASTER) return fail(nullptr, GS_ERR_INVALID, "gs_create: unknown tile_order");
    if (cfg.count_launches > GS_COUNT_FED) return fail(nullptr, GS_ERR_INVALID, "gs_create: unknown count_launches");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, GS_ERR_NO_DEVICE,
                    "gs_create: no HIP device available (this library has no CPU fallback)");
    if (cfg.device_ordinal < 0 || cfg.device_ordinal >= count)
        return fail(nullptr, GS_ERR_INVALID, "gs_create: device_ordinal out of range");
    gs_ctx* c = new (std::nothrow) gs_ctx();
    if (!c) return fail(nullptr, GS_ERR_INVALID, "gs_create: out of host memory");
    c->cfg = cfg;
    c->device = cfg.device_ordinal;
    if ((e = hipSetDevice(c->device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        std::string msg = std::string("gs_create: ") + hipGetErrorString(e);
        delete c;
        return fail(nullptr, GS_ERR_HIP, msg);
    }
    c->stream = c->own_stream;
    // a pinned host word the frames' k_scan_blocks writes their element count to (GS_COUNT_AUTO reads it when it enqueues
    // the next frame); doing without it is no error: the sort then keeps a Count launch per pass
    if (cfg.sort_algorithm == GS_SORT_RADIX4 && cfg.count_launches == GS_COUNT_AUTO) {
        void* note = nullptr;
        if (hipHostMalloc(&note, 64, hipHostMallocMapped) == hipSuccess && note) {
            c->elems_note = (uint32_t*)note;
            *c->elems_note = 0u;
        } else {
            (void)hipGetLastError();
        }
    }
    for (auto& ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) {
            std::string msg = std::string("gs_create: ") + hipGetErrorString(e);
            gs_destroy(c);
            return fail(nullptr, GS_ERR_HIP, msg);
        }
    for (auto& ev : c->alt_ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) {
            std::string msg = std::string("gs_create: ") + hipGetErrorString(e);
            gs_destroy(c);
            return fail(nullptr, GS_ERR_HIP, msg);
        }
    for (auto& ev : c->pre_ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) {
            std::string msg = std::string("gs_create: ") + hipGetErrorString(e);
            gs_destroy(c);
            return fail(nullptr, GS_ERR_HIP, msg);
        }
    if (cfg.sort_algorithm == GS_SORT_TILE_BUCKET) {
        if (init_tile_sort() != 0) {
            gs_destroy(c);
            return fail(nullptr, GS_ERR_HIP, "gs_create: cannot reserve 160 KB of LDS for the per-tile sort");
        }
        e = hipStreamCreateWithFlags(&c->helper_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            std::string msg = std::string("gs_create: ") + hipGetErrorString(e);
            gs_destroy(c);
            return fail(nullptr, GS_ERR_HIP, msg);
        }
    }
    if (cfg.record_timings >= 2)
        for (auto& ev : c->scatter_ev)
            if ((e = hipEventCreate(&ev)) != hipSuccess) {
                std::string msg = std::string("gs_create: ") + hipGetErrorString(e);
                gs_destroy(c);
                return fail(nullptr, GS_ERR_HIP, msg);
            }
    *out = c;
    return GS_OK;
}

int gs_destroy(gs_ctx* c) {
    if (!c) return GS_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->dist_comm) (void)gs_dist_destroy(c);
    free_resolution(c);
    free_scene(c);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->scatter_ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->alt_ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->pre_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    if (c->helper_stream) { (void)hipStreamSynchronize(c->helper_stream); (void)hipStreamDestroy(c->helper_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->elems_note) (void)hipHostFree(c->elems_note);
    delete c;
    return GS_OK;
}

const char* gs_last_error(const gs_ctx* c) { return c ? c->last_error.c_str() : g_create_error.c_str(); }

int gs_set_stream(gs_ctx* c, void* hip_stream) {
    if (!c) return GS_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return GS_OK;
}

// per-context, per-frame outputs of InitSortList's first kernel
static int alloc_scratch(gs_ctx* c, uint32_t n) {
    const size_t N = n;
    HIP_TRY(c, hipMalloc((void**)&c->scratch.raster, N * sizeof(SplatRaster)));
    HIP_TRY(c, hipMalloc((void**)&c->scratch.depth_key, N * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void**)&c->scratch.tiles_touched, N * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void**)&c->scratch.extents, N * sizeof(uint2)));
    c->num_blocks = (n + kProjThreads - 1) / kProjThreads;
    // k_scan_blocks reads/writes whole 16-byte groups up to 1024 * per entries: zero-padded
    const size_t padded = (size_t)c->num_blocks + 8192;
    HIP_TRY(c, hipMalloc((void**)&c->scratch.block_sums, padded * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void**)&c->scratch.block_offsets, padded * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->scratch.block_sums, 0, padded * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->scratch.block_offsets, 0, padded * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->scratch.raster, 0, N * sizeof(SplatRaster), c->stream));
    HIP_TRY(c, hipMalloc((void**)&c->scratch.wave_wrote, (size_t)c->num_blocks * 4));
    HIP_TRY(c, hipMemsetAsync(c->scratch.wave_wrote, 0, (size_t)c->num_blocks * 4, c->stream));
    HIP_TRY(c, hipMalloc((void**)&c->scratch.help_list, (size_t)kEmitHelpCap * sizeof(uint2)));
    HIP_TRY(c, hipMalloc((void**)&c->scratch.help_count, 4 * sizeof(uint32_t)));
    c->scratch.elems_note = nullptr;
    if (c->elems_note) {
        void* dev = nullptr;
        if (hipHostGetDevicePointer(&dev, c->elems_note, 0) == hipSuccess) c->scratch.elems_note = (uint32_t*)dev;
        else (void)hipGetLastError();
    }
    HIP_TRY(c, hipMalloc((void**)&c->scratch.band_list, (size_t)c->num_blocks * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void**)&c->scratch.help_slot, (size_t)c->num_blocks * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->scratch.help_count, 0, 4 * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->scratch.help_list, 0, (size_t)kEmitHelpCap * sizeof(uint2), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->scratch.help_slot, 0xFF, (size_t)c->num_blocks * sizeof(uint32_t), c->stream));
    c->emit_parity = 0;
    if (sorts_splat_first(c->cfg.sort_algorithm)) {
        HIP_TRY(c, hipMalloc((void**)&c->scratch.block_flags, padded * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void**)&c->scratch.flag_offsets, padded * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void**)&c->scratch.sorted_sums, padded * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void**)&c->scratch.aux_params, 2 * sizeof(SortParams)));
        HIP_TRY(c, hipMemsetAsync(c->scratch.block_flags, 0, padded * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->scratch.flag_offsets, 0, padded * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->scratch.sorted_sums, 0, padded * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->scratch.aux_params, 0, 2 * sizeof(SortParams), c->stream));
    }
    return GS_OK;
}

// Frames in flight (GfxSettings::FRAMES_IN_FLIGHT = 3, GfxSettings.h:15): several contexts render the same scene
// on their own streams with their own per-frame buffers; only the read-only gaussian arrays are shared.
int gs_share_scene(gs_ctx* c, gs_ctx* owner) {
    if (!c || !owner || c == owner) return GS_ERR_INVALID;
    if (!owner->n || !owner->shared) return fail(c, GS_ERR_NO_SCENE, "gs_share_scene: the owner has no gaussians uploaded");
    if (owner->device != c->device) return fail(c, GS_ERR_INVALID, "gs_share_scene: contexts are on different devices");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    SharedScene* sh = owner->shared;
    sh->refs.fetch_add(1);            // taken first: c may currently hold the same arrays
    free_resolution(c);
    free_scene(c);
    c->shared = sh;
    c->scene = sh->b;
    if (int r = alloc_scratch(c, sh->n)) { free_scene(c); return r; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n = sh->n;
    return GS_OK;
}

int gs_upload_gaussians(gs_ctx* c, const void* aos336, uint32_t n) {
    if (!c) return GS_ERR_INVALID;
    if (!aos336 || n == 0) return fail(c, GS_ERR_INVALID, "gs_upload_gaussians: empty input");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_resolution(c);   // capacity depends on n (Renderer.cpp:725)
    free_scene(c);        // contexts that share the previous arrays keep them
    c->shared = new (std::nothrow) SharedScene();
    if (!c->shared) return fail(c, GS_ERR_INVALID, "gs_upload_gaussians: out of host memory");
    c->shared->n = n;
    const size_t N = n;
    {
        SceneBuffers& b = c->shared->b;
        hipError_t e = hipMalloc((void**)&b.pos, 3 * N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&b.scale, 3 * N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&b.rot, 4 * N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&b.sh, 48 * N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&b.opacity, N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&b.sig2, N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&b.block_bounds, ((size_t)(n + 63u) / 64u + 4u) * 8 * sizeof(float));
        if (e != hipSuccess) {
            free_scene(c);
            return fail(c, GS_ERR_HIP, std::string("gs_upload_gaussians: ") + hipGetErrorString(e));
        }
        c->scene = b;
    }
    if (int r = alloc_scratch(c, n)) { free_scene(c); return r; }

    // AoS -> SoA on the device, through a bounded staging buffer
    const uint32_t chunk = n < (1u << 20) ? n : (1u << 20);
    float* staging = nullptr;
    if (hipMalloc((void**)&staging, (size_t)chunk * GS_GAUSSIAN_RECORD_BYTES) != hipSuccess) {
        free_scene(c);
        return fail(c, GS_ERR_HIP, "gs_upload_gaussians: cannot allocate the staging buffer");
    }
    const char* src = static_cast<const char*>(aos336);
    int rc = GS_OK;
    for (uint32_t first = 0; first < n && rc == GS_OK; first += chunk) {
        const uint32_t cnt = (n - first) < chunk ? (n - first) : chunk;
        hipError_t e = hipMemcpyAsync(staging, src + (size_t)first * GS_GAUSSIAN_RECORD_BYTES,
                                      (size_t)cnt * GS_GAUSSIAN_RECORD_BYTES, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            launch_aos_to_soa(staging, first, cnt, n, c->scene, c->stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // staging is reused
        if (e != hipSuccess) rc = fail(c, GS_ERR_HIP, std::string("gs_upload_gaussians: ") + hipGetErrorString(e));
    }
    (void)hipFree(staging);
    if (rc == GS_OK) {
        launch_block_bounds(n, c->scene, c->stream);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, GS_ERR_HIP, std::string("gs_upload_gaussians: ") + hipGetErrorString(e));
    }
    if (rc != GS_OK) { free_scene(c); return rc; }
    c->n = n;
    return GS_OK;
}

int gs_set_resolution(gs_ctx* c, uint32_t width, uint32_t height) {
    if (!c) return GS_ERR_INVALID;
    if (!c->n) return fail(c, GS_ERR_NO_SCENE, "gs_set_resolution: upload gaussians first");
    if (width == 0 || height == 0 || width > 65535u * 16u || height > 65535u * 16u)
        return fail(c, GS_ERR_INVALID, "gs_set_resolution: bad extent");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_resolution(c);
    const uint32_t gw = (width + kTile - 1) / kTile, gh = (height + kTile - 1) / kTile; // Renderer.cpp:696-701
    const uint64_t want = (uint64_t)c->n + 64ull * 16ull * gw * gh;                       // Renderer.cpp:725
    if (want > (1ull << 31)) return fail(c, GS_ERR_INVALID, "gs_set_resolution: sort list would exceed 2^31 elements");
    c->width = width; c->height = height; c->grid_w = gw; c->grid_h = gh;
    c->row_begin = 0; c->row_end = gh; c->row_stride = 1; c->first_row = 0; c->rows_owned = gh;
    c->compact_out = false;
    c->capacity = ceil_pow2((uint32_t)want);
    c->num_sort_bits = num_sort_bits_for(gw * gh);
    c->band_sort_bits = c->num_sort_bits;
    c->hi16 = (uint64_t)gw * gh <= 65535u;
    int rc = alloc_sort(c, c->sort, c->capacity, digit_bits_of(c->cfg.sort_algorithm));
    if (rc != GS_OK) { free_resolution(c); return rc; }
    // any failure from here on leaves the context without a resolution (capacity 0), never half set up
    hipError_t e = hipMalloc((void**)&c->ranges, ((size_t)gw * gh * 2 * sizeof(uint32_t) + 15) & ~(size_t)15);   // cleared 16 bytes at a time
    if (e == hipSuccess) e = hipMalloc((void**)&c->tile_order, tile_order_words(gw, gh) * sizeof(uint32_t));   // table + scratch of the two kernels
    if (e == hipSuccess) e = hipMalloc((void**)&c->framebuffer, (size_t)width * height * 4);
    if (e == hipSuccess) e = hipMemset(c->ranges, 0, (size_t)gw * gh * 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->framebuffer, 0, (size_t)width * height * 4);
    if (e != hipSuccess) {
        free_resolution(c);
        return fail(c, GS_ERR_HIP, std::string("gs_set_resolution: ") + hipGetErrorString(e));
    }
    return GS_OK;
}

static int apply_tile_rows(gs_ctx* c, uint32_t row_begin, uint32_t row_end, uint32_t stride, uint32_t phase,
                           bool compact_out) {
    c->row_begin = row_begin; c->row_end = row_end; c->row_stride = stride;
    c->first_row = row_begin + phase;
    c->rows_owned = c->first_row < row_end ? (row_end - c->first_row + stride - 1u) / stride : 0u;
    c->compact_out = compact_out;
    if (c->elems_note) {           // another band: another list length (GS_COUNT_AUTO learns it from the next frame)
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // no frame in flight may still write the old one
        *(volatile uint32_t*)c->elems_note = 0u;
    }
    const uint32_t owned_tiles = c->rows_owned * c->grid_w;
    c->band_sort_bits = num_sort_bits_for(owned_tiles ? owned_tiles : 1u);
    c->hi16 = owned_tiles <= 65535u;
    if (c->sort_graph || c->presort_graph || c->chain_graph) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // the graph may still be executing
        drop_sort_graph(c);
    }
    return GS_OK;
}

int gs_set_tile_rows(gs_ctx* c, uint32_t row_begin, uint32_t row_end) {
    if (!c) return GS_ERR_INVALID;
    if (!c->capacity) return fail(c, GS_ERR_NO_SCENE, "gs_set_tile_rows: gs_set_resolution not called");
    if (row_begin > row_end || row_end > c->grid_h)
        return fail(c, GS_ERR_INVALID, "gs_set_tile_rows: need row_begin <= row_end <= tiles_y");
    return apply_tile_rows(c, row_begin, row_end, 1u, 0u, false);
}

int gs_set_tile_rows_interleaved(gs_ctx* c, uint32_t phase, uint32_t stride, uint32_t compact_output) {
    if (!c) return GS_ERR_INVALID;
    if (!c->capacity) return fail(c, GS_ERR_NO_SCENE, "gs_set_tile_rows_interleaved: gs_set_resolution not called");
    if (stride == 0u || phase >= stride)
        return fail(c, GS_ERR_INVALID, "gs_set_tile_rows_interleaved: need 0 <= phase < stride");
    return apply_tile_rows(c, 0u, c->grid_h, stride, phase, compact_output != 0u);
}

int gs_get_scene_info(const gs_ctx* c, gs_scene_info* out) {
    if (!c || !out) return GS_ERR_INVALID;
    out->num_gaussians = c->n;
    out->width = c->width; out->height = c->height;
    out->tiles_x = c->grid_w; out->tiles_y = c->grid_h;
    out->capacity = c->capacity; out->num_sort_bits = c->num_sort_bits;
    out->row_begin = c->row_begin; out->row_end = c->row_end;
    out->tile_word_bytes = c->hi16 ? 2u : 4u;
    out->row_stride = c->row_stride; out->first_row = c->first_row; out->rows_owned = c->rows_owned;
    return GS_OK;
}

int gs_render_device_async(gs_ctx* c, const float view[16], const float proj[16],
                           const float cam_pos[3], uint32_t sh_mode, void* rgba_out_device) {
    if (!c) return GS_ERR_INVALID;
    host_frame_begin(c);
    HIP_TRY(c, hipSetDevice(c->device));
    const HostClock::time_point t_rec = HostClock::now();
    const int rc = enqueue_frame(c, view, proj, cam_pos, sh_mode, static_cast<uint8_t*>(rgba_out_device));
    c->host.record_ms = (float)ms_since(t_rec);      // recordCommandBuffer + submit
    return rc;
}

int gs_render_device(gs_ctx* c, const float view[16], const float proj[16], const float cam_pos[3],
                     uint32_t sh_mode, void* rgba_out_device) {
    int rc = gs_render_device_async(c, view, proj, cam_pos, sh_mode, rgba_out_device);
    if (rc != GS_OK) return rc;
    return finish_frame(c);
}

int gs_render(gs_ctx* c, const float view[16], const float proj[16], const float cam_pos[3],
              uint32_t sh_mode, uint8_t* rgba_out) {
    if (!c) return GS_ERR_INVALID;
    if (!rgba_out) return fail(c, GS_ERR_INVALID, "gs_render: rgba_out is null");
    int rc = gs_render_device_async(c, view, proj, cam_pos, sh_mode, nullptr);
    if (rc != GS_OK) return rc;
    rc = finish_frame(c);
    if (rc < 0) return rc;
    const HostClock::time_point t_present = HostClock::now();
    hipError_t e = hipMemcpy(rgba_out, c->framebuffer, (size_t)c->width * c->height * 4, hipMemcpyDeviceToHost);
    c->host.present_ms = (float)ms_since(t_present);   // where the reference presents, this sink copies the frame out
    HIP_TRY(c, e);
    return rc;
}

int gs_debug_init_sort_list(gs_ctx* c, const float view[16], const float proj[16],
                            const float cam_pos[3], uint32_t sh_mode) {
    if (!c) return GS_ERR_INVALID;
    if (!c->n || !c->capacity) return fail(c, GS_ERR_NO_SCENE, "gs_debug_init_sort_list: scene/resolution not set");
    if (!view || !proj || !cam_pos || sh_mode > 2u) return fail(c, GS_ERR_INVALID, "gs_debug_init_sort_list: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    FrameParams fp = make_frame_params(c, view, proj, cam_pos, sh_mode);
    fp.parity = (c->emit_parity ^= 1u);
    c->last_fp = fp;
    launch_project(fp, c->scene, c->scratch, c->stream);
    launch_scan_blocks(fp, c->scratch, c->sort.params, c->ranges, c->sort.coarse, c->stream);
    launch_emit(fp, c->scratch, c->sort, c->stream);
    if (int r = check_launch(c, "InitSortList")) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_frame = false;
    c->unsorted_valid = true;
    SortParams sp{};
    HIP_TRY(c, hipMemcpy(&sp, c->sort.params, sizeof(sp), hipMemcpyDeviceToHost));
    return sp.overflow ? GS_WARN_OVERFLOW : GS_OK;
}

int gs_synchronize(gs_ctx* c) {
    if (!c) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

int gs_get_timings(const gs_ctx* c, gs_timings* out) {
    if (!c || !out) return GS_ERR_INVALID;
    *out = c->timings;
    return GS_OK;
}

int gs_get_host_timings(const gs_ctx* c, gs_host_timings* out) {
    if (!c || !out) return GS_ERR_INVALID;
    *out = c->host;
    return GS_OK;
}

// The sort list holds compact tile ids (uint16 or uint32, FrameParams) -> the uint32 GLOBAL tile ids callers expect
static int read_tile_words(gs_ctx* c, const uint32_t* dev, uint32_t num_elems, void* dst, size_t bytes) {
    if (bytes > (size_t)num_elems * sizeof(uint32_t)) return fail(c, GS_ERR_INVALID, "gs_debug_read: size exceeds buffer");
    const size_t cnt = bytes / sizeof(uint32_t);
    uint32_t* out = static_cast<uint32_t*>(dst);
    if (c->hi16) {
        std::vector<uint16_t> h(num_elems);
        if (num_elems) HIP_TRY(c, hipMemcpy(h.data(), dev, (size_t)num_elems * sizeof(uint16_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; ++i) out[i] = h[i];
    } else if (cnt) {
        HIP_TRY(c, hipMemcpy(out, dev, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < cnt; ++i) {
        const uint32_t k = out[i] / c->grid_w, x = out[i] - k * c->grid_w;
        out[i] = (c->first_row + k * c->row_stride) * c->grid_w + x;
    }
    return GS_OK;
}

int gs_debug_read(gs_ctx* c, int which, void* dst, size_t bytes) {
    if (!c || !dst) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!c->have_frame && !c->unsorted_valid && which != GS_BUF_IMAGE) return fail(c, GS_ERR_NO_SCENE, "gs_debug_read: no frame rendered yet");
    SortParams sp{};
    HIP_TRY(c, hipMemcpy(&sp, c->sort.params, sizeof(sp), hipMemcpyDeviceToHost));
    const size_t e_bytes = (size_t)sp.num_elems * sizeof(uint32_t);
    const void* src = nullptr;
    size_t avail = 0;
    const int si = c->sorted_index;
    switch (which) {
        case GS_BUF_SORTED_TILE:
            return read_tile_words(c, c->sort.hi[si], sp.num_elems, dst, bytes);
        case GS_BUF_SORTED_DEPTH:
            if (c->depth_dropped && c->have_frame && !c->unsorted_valid) {
                // the frame path stops moving the depth words once they are sorted: rebuild them from the ids
                if (bytes > e_bytes) return fail(c, GS_ERR_INVALID, "gs_debug_read: size exceeds buffer");
                std::vector<uint32_t> ids(sp.num_elems), depth(c->n);
                if (sp.num_elems) HIP_TRY(c, hipMemcpy(ids.data(), c->sort.id[si], e_bytes, hipMemcpyDeviceToHost));
                HIP_TRY(c, hipMemcpy(depth.data(), c->scratch.depth_key, (size_t)c->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
                uint32_t* out = static_cast<uint32_t*>(dst);
                for (size_t i = 0; i < bytes / sizeof(uint32_t); ++i) out[i] = ids[i] < c->n ? depth[ids[i]] : 0u;
                return GS_OK;
            }
            src = c->sort.lo[si]; avail = e_bytes; break;
        case GS_BUF_SORTED_ID: src = c->sort.id[si]; avail = e_bytes; break;
        case GS_BUF_RANGES: src = c->ranges; avail = (size_t)c->grid_w * c->grid_h * 8; break;
        case GS_BUF_COUNT: {
            if (bytes > sizeof(uint64_t)) return fail(c, GS_ERR_INVALID, "gs_debug_read: size");
            std::memcpy(dst, &sp.counter, bytes);
            return GS_OK;
        }
        case GS_BUF_IMAGE: src = c->framebuffer; avail = (size_t)c->width * c->height * 4; break;
        case GS_BUF_UNSORTED_TILE:
        case GS_BUF_UNSORTED_DEPTH:
        case GS_BUF_UNSORTED_ID:
            // the list as emitted lives in ping-pong half 0 and is overwritten by the second pass
            if (!c->unsorted_valid)
                return fail(c, GS_ERR_INVALID, "gs_debug_read: unsorted list only valid after gs_debug_init_sort_list");
            if (which == GS_BUF_UNSORTED_TILE) return read_tile_words(c, c->sort.hi[0], sp.num_elems, dst, bytes);
            src = which == GS_BUF_UNSORTED_DEPTH ? c->sort.lo[0] : c->sort.id[0];
            avail = e_bytes;
            break;
        case GS_BUF_COLOR:
        case GS_BUF_COV: {
            const size_t need = (size_t)c->n * 4 * sizeof(float);
            if (bytes > need) return fail(c, GS_ERR_INVALID, "gs_debug_read: size exceeds buffer");
            std::vector<SplatRaster> host(c->n);
            HIP_TRY(c, hipMemcpy(host.data(), c->scratch.raster, (size_t)c->n * sizeof(SplatRaster), hipMemcpyDeviceToHost));
            // color.a as the reference stores it (InitSortList.comp:126) is the opacity itself; the record's copy is
            // already zeroed where the 2x2 determinant vanishes (RenderGaussians.comp:104), so it comes from the scene
            std::vector<float> opacity;
            std::vector<uint32_t> touched;
            // N6 (InitSortList.comp:124-127): the reference stores colour for EVERY splat that passes the culls; the frame
            // evaluates it for the emitting ones only, so the others are evaluated here, on demand, from the last frame's
            // camera (k_debug_colour) -- in a context that owns the whole grid.  A context that owns a subset of the tile rows
            // does not even project the splats that cannot reach its rows: there GS_BUF_COLOR stays what the frame stored.
            std::vector<float> on_demand;
            const bool whole_grid = c->row_begin == 0u && c->row_end == c->grid_h && c->row_stride == 1u;
            if (which == GS_BUF_COLOR) {
                opacity.resize(c->n); touched.resize(c->n);
                HIP_TRY(c, hipMemcpy(opacity.data(), c->scene.opacity, (size_t)c->n * sizeof(float), hipMemcpyDeviceToHost));
                HIP_TRY(c, hipMemcpy(touched.data(), c->scratch.tiles_touched, (size_t)c->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
                if (whole_grid) {
                    float* dev = nullptr;
                    HIP_TRY(c, hipMalloc((void**)&dev, need));
                    launch_debug_colour(c->last_fp, c->scene, c->scratch, dev, c->stream);
                    hipError_t e = hipGetLastError();
                    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                    on_demand.resize((size_t)c->n * 4);
                    if (e == hipSuccess) e = hipMemcpy(on_demand.data(), dev, need, hipMemcpyDeviceToHost);
                    (void)hipFree(dev);
                    HIP_TRY(c, e);
                }
            }
            // records of a wave (64 consecutive splats) that k_project did not store this frame -- wholly culled, or with
            // nothing to emit into this context's tile rows -- read back as zero, what a culled splat's scratch holds
            std::vector<uint8_t> wrote((size_t)c->num_blocks * 4);
            HIP_TRY(c, hipMemcpy(wrote.data(), c->scratch.wave_wrote, wrote.size(), hipMemcpyDeviceToHost));
            std::vector<float> outv((size_t)c->n * 4, 0.0f);
            for (uint32_t i = 0; i < c->n; ++i) {
                if (!wrote[i / 64u]) continue;
                const SplatRaster& r = host[i];
                float* o = &outv[(size_t)i * 4];
                if (which == GS_BUF_COLOR) {
                    if (touched[i]) { o[0] = r.r; o[1] = r.g; o[2] = r.b; o[3] = opacity[i]; }       // what the frame stored and blended
                    else if (!on_demand.empty() && on_demand[(size_t)i * 4 + 3] != 0.0f) {           // passed the culls, touched no tile
                        o[0] = on_demand[(size_t)i * 4]; o[1] = on_demand[(size_t)i * 4 + 1]; o[2] = on_demand[(size_t)i * 4 + 2]; o[3] = opacity[i];
                    }
                }
                else { o[0] = r.cx; o[1] = r.cy; o[2] = r.cz; o[3] = 0.0f; }
            }
            std::memcpy(dst, outv.data(), bytes);
            return GS_OK;
        }
        default: return fail(c, GS_ERR_INVALID, "gs_debug_read: unknown buffer id");
    }
    if (bytes > avail) return fail(c, GS_ERR_INVALID, "gs_debug_read: size exceeds buffer");
    if (bytes) HIP_TRY(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return GS_OK;
}

// Camera.cpp:7-48 over the glm 0.9.9.8 formulas (lookAtRH, perspectiveRH_ZO, normalize, cross).
int gs_camera_matrices(const float pos[3], float yaw, float pitch, float aspect, float near_plane,
                       float far_plane, float view[16], float proj[16]) {
    if (!pos || !view || !proj) return GS_ERR_INVALID;
    auto normalize3 = [](float v[3]) {
        const float d = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        const float inv = 1.0f / std::sqrt(d);
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
    };
    auto cross3 = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - b[1] * a[2];
        o[1] = a[2] * b[0] - b[2] * a[0];
        o[2] = a[0] * b[1] - b[0] * a[1];
    };
    auto dot3 = [](const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    float fwd[3] = {(float)(std::sin((double)yaw) * std::cos((double)pitch)), (float)std::sin((double)pitch),
                    (float)(std::cos((double)yaw) * std::cos((double)pitch))};   // Camera.cpp:10-16
    normalize3(fwd);
    const float up[3] = {0.0f, 1.0f, 0.0f};
    const float center[3] = {pos[0] + fwd[0], pos[1] + fwd[1], pos[2] + fwd[2]}; // Camera.cpp:34-38
    float f[3] = {center[0] - pos[0], center[1] - pos[1], center[2] - pos[2]};
    normalize3(f);
    float s[3], u[3];
    cross3(f, up, s);
    normalize3(s);
    cross3(s, f, u);
    std::memset(view, 0, 16 * sizeof(float));
    view[0] = s[0]; view[4] = s[1]; view[8] = s[2];
    view[1] = u[0]; view[5] = u[1]; view[9] = u[2];
    view[2] = -f[0]; view[6] = -f[1]; view[10] = -f[2];
    view[12] = -dot3(s, pos); view[13] = -dot3(u, pos); view[14] = dot3(f, pos);
    view[15] = 1.0f;
    const float fovy = 90.0f * 0.01745329251994329576923690768489f;                // Camera.cpp:42
    const float tan_half = std::tan(fovy / 2.0f);
    std::memset(proj, 0, 16 * sizeof(float));
    proj[0] = 1.0f / (aspect * tan_half);
    proj[5] = 1.0f / tan_half;
    proj[10] = far_plane / (near_plane - far_plane);
    proj[11] = -1.0f;
    proj[14] = -(far_plane * near_plane) / (far_plane - near_plane);
    return GS_OK;
}

int gs_sort_host(gs_ctx* c, uint32_t* tile, uint32_t* depth, uint32_t* id, uint32_t n,
                 uint32_t num_sort_bits) {
    if (!c || !tile || !depth || !id) return GS_ERR_INVALID;
    if (num_sort_bits == 0 || num_sort_bits > 64 || (num_sort_bits % kRadixBits) != 0)
        return fail(c, GS_ERR_INVALID, "gs_sort_host: num_sort_bits must be a multiple of 4 in [4,64]");
    if (n == 0) return GS_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    SortBuffers sb{};
    int rc = alloc_sort(c, sb, n, digit_bits_of(c->cfg.sort_algorithm));
    if (rc != GS_OK) { free_sort(sb); return rc; }
    const size_t bytes = (size_t)n * sizeof(uint32_t);
    hipError_t e = hipMemcpyAsync(sb.hi[0], tile, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sb.lo[0], depth, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sb.id[0], id, bytes, hipMemcpyHostToDevice, c->stream);
    int si = 0;
    if (e == hipSuccess) {
        launch_set_sort_params(sb.params, sb.coarse, n, c->stream);
        si = launch_radix_sort(sb, n, num_sort_bits, c->stream, nullptr, 0u, false, false, 1.0f, 0, 0, nullptr,
                               digit_bits_of(c->cfg.sort_algorithm), fed_for(c, n));
        e = hipGetLastError();
        if (si < 0) { free_sort(sb); return fail(c, GS_ERR_INVALID, "gs_sort_host: sort buffers of another digit width"); }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(tile, sb.hi[si], bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(depth, sb.lo[si], bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(id, sb.id[si], bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free_sort(sb);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_sort_host: ") + hipGetErrorString(e));
    return GS_OK;
}

int gs_sort_bench(gs_ctx* c, uint32_t n, uint32_t num_tiles, uint32_t iters, uint64_t seed,
                  float* ms_per_sort, uint32_t* sorted_ok) {
    if (!c || !ms_per_sort || n == 0 || num_tiles == 0 || iters == 0) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    SortBuffers sb{};
    int rc = alloc_sort(c, sb, n, digit_bits_of(c->cfg.sort_algorithm));
    if (rc != GS_OK) { free_sort(sb); return rc; }
    const uint32_t bits = num_sort_bits_for(num_tiles);
    uint32_t* bad = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void**)&bad, sizeof(uint32_t));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float total_ms = 0.0f;
    int si = 0;
    for (uint32_t it = 0; it < iters + 1 && e == hipSuccess; ++it) {   // iteration 0 = warm-up
        launch_fill_random_keys(sb.lo[0], sb.hi[0], sb.id[0], n, num_tiles, seed + it, c->stream);
        launch_set_sort_params(sb.params, sb.coarse, n, c->stream);
        e = hipEventRecord(e0, c->stream);
        if (e != hipSuccess) break;
        si = launch_radix_sort(sb, n, bits, c->stream, nullptr, 0u, false, false, 1.0f, 0, 0, nullptr,
                               digit_bits_of(c->cfg.sort_algorithm), fed_for(c, n));
        if (si < 0) { si = 0; e = hipErrorInvalidValue; break; }
        e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (it > 0) total_ms += ms;
    }
    uint32_t bad_host = 0;
    if (e == hipSuccess) e = hipMemsetAsync(bad, 0, sizeof(uint32_t), c->stream);
    if (e == hipSuccess) {
        launch_check_sorted(sb.lo[si], sb.hi[si], n, bad, c->stream);
        e = hipMemcpyAsync(&bad_host, bad, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (bad) (void)hipFree(bad);
    free_sort(sb);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_sort_bench: ") + hipGetErrorString(e));
    *ms_per_sort = total_ms / (float)iters;
    if (sorted_ok) *sorted_ok = bad_host == 0 ? 1u : 0u;
    return GS_OK;
}

int gs_membench(gs_ctx* c, int kind, size_t bytes, uint32_t blocks, uint32_t iters, float* gbps, float* ms_out) {
    if (!c || !gbps || bytes < 16 || iters == 0 || kind < 0 || kind > 12) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    bytes &= ~(size_t)15;
    if (blocks == 0) blocks = 2048;
    void *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&src, bytes + 65536);     // slack: the skewed scatter probes write a little past `bytes`
    if (e == hipSuccess) e = hipMalloc(&dst, bytes + 65536);
    if (e == hipSuccess) e = hipMemsetAsync(src, 0x5A, bytes, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dst, 0, bytes, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.0f;
    if (e == hipSuccess) {
        for (int w = 0; w < 3; ++w) launch_stream_probe(kind, src, dst, bytes, blocks, c->stream);
        e = hipEventRecord(e0, c->stream);
        for (uint32_t i = 0; i < iters; ++i) launch_stream_probe(kind, src, dst, bytes, blocks, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_membench: ") + hipGetErrorString(e));
    const double moved = (double)bytes * ((kind & 1) || kind >= 4 ? 2.0 : 1.0) * iters;
    *gbps = (float)(moved / (ms * 1e-3) / 1e9);
    if (ms_out) *ms_out = ms / (float)iters;
    return GS_OK;
}

} // extern "C"
