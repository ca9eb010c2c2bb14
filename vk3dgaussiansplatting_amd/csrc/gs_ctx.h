// gs_ctx.h -- the context behind the opaque gs_ctx handle of include/gsplat.h.  Internal: shared by gs_api.cpp and by
// the tuning probes of tools/probe (libgsplat_probe.so), which are built from this tree against the same layout.
#pragma once

#include "../../include/gsplat.h"
#include "gs_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

// The uploaded gaussian arrays (read-only on the path), reference-counted so that the contexts that render them
// (gs_share_scene: frame slots, tile-row bands) can be destroyed in any order.
struct SharedScene {
    gs::SceneBuffers b{};
    uint32_t n = 0;
    std::atomic<uint32_t> refs{1};
};

using HostClock = std::chrono::steady_clock;

struct gs_ctx {
    gs_config cfg{};
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[7] = {};
    hipEvent_t scatter_ev[32] = {};   // record_timings == 2: a pair per pass (<= 16 passes)
    hipEvent_t alt_ev[2] = {};        // GS_SORT_TILE_BUCKET: after FindRanges / after the per-tile sort
    hipStream_t helper_stream = nullptr;   // GS_SORT_TILE_BUCKET: big size classes run beside the small ones
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    std::string last_error;

    // scene
    uint32_t n = 0;
    gs::SceneBuffers scene{};
    gs::SplatScratch scratch{};
    uint32_t num_blocks = 0;
    uint32_t emit_parity = 0;     // FrameParams::parity of the last InitSortList launch
    SharedScene* shared = nullptr;    // owner of `scene`'s arrays (this context holds one reference)

    // resolution-dependent
    uint32_t width = 0, height = 0, grid_w = 0, grid_h = 0;
    // tile rows of this context: first_row + k * row_stride < row_end, k < rows_owned (FrameParams)
    uint32_t row_begin = 0, row_end = 0, row_stride = 1, first_row = 0, rows_owned = 0;
    bool compact_out = false;
    uint32_t capacity = 0, num_sort_bits = 0;
    // what the sort of the owned tiles runs over: compact tile ids, so ceil((32 + bits(T_owned - 1)) / 4) passes
    // (the reference's formula, RadixSort.cpp:203-204, for the context's own tile count)
    uint32_t band_sort_bits = 0;
    bool hi16 = false;   // the frame's sort list stores the compact tile ids as uint16 (at most 65535 owned tiles)
    gs::SortBuffers sort{};
    uint32_t* ranges = nullptr;
    uint32_t* tile_order = nullptr;   // [tiles] RenderGaussians' dispatch order (GS_TILE_ORDER_LONGEST_FIRST)
    uint8_t* framebuffer = nullptr;
    int sorted_index = 0;       // which ping-pong half holds the sorted list after the last frame
    // The radix passes of a frame (3 launches per pass, parameters fixed once resolution and band are) replayed
    // as one hipGraph launch: 36 launches -> 1 on the host side.  Built lazily, dropped when anything it baked in
    // changes.  Not used while per-Scatter events are recorded (record_timings == 2).
    hipGraphExec_t sort_graph = nullptr;
    hipGraphExec_t presort_graph = nullptr;   // GS_SORT_RADIX4_SPLAT_FIRST: the eight depth passes over the splat list
    hipGraphExec_t chain_graph = nullptr;     // ... without timers: everything from the splat list to FindRanges as one graph
    int chain_result = 0;
    hipEvent_t pre_ev[3] = {};        // ... after the splat list / after its passes / after the emit
    int sort_graph_result = 0, presort_result = 1;
    bool sort_graph_failed = false;
    uint32_t* elems_note = nullptr;   // pinned host word k_scan_blocks writes (element count + 1 of the latest list; 0: none
                                      // since the rows were set) -- what GS_COUNT_AUTO goes by
    bool sort_fed = false;        // the captured radix passes are the fed kind (k_scatter<.., FED>)
    bool depth_dropped = false;   // last frame's tile-word passes did not carry the depth words (see k_scatter)

    gs_timings timings{};
    gs_host_timings host{};           // RECORD_CPU_TIMES (Renderer.cpp:299-456)
    HostClock::time_point last_entry{};
    bool have_entry = false;
    bool have_frame = false;
    bool unsorted_valid = false;   // last thing run was gs_debug_init_sort_list
    gs::FrameParams last_fp{};     // of the last InitSortList launch (gs_debug_read(GS_BUF_COLOR) evaluates colours on demand)

    // gs_dist_init: the RCCL communicator of a multi-GPU frame (gs_dist.cpp)
    void* dist_comm = nullptr;
    int dist_rank = 0, dist_world = 1;
    // gs_dist_shard_rows: how the tile rows are dealt, and the buffers of a sharded frame -- two slots, so that the gather
    // of frame f (on dist_stream) runs beside the kernels of frame f + 1 (on the context's stream)
    uint32_t dist_dealing = 0;            // GS_ROWS_*
    bool dist_sharded = false;
    std::vector<uint32_t> dist_edges;     // contiguous / balanced: R + 1 band edges (tile rows), the same on every rank
    uint32_t dist_rows_sig[5] = {};       // row_begin, row_end, row_stride, first_row, compact_out the buffers were set up for
    void* dist_strip[2] = {};             // this rank's rows (not on the root of a contiguous dealing: it renders into the frame)
    void* dist_gathered[2] = {};          // root, interleaved rows: the ranks' packed strips before they are re-ordered
    void* dist_image[2] = {};             // root: the assembled frame, left in HBM
    size_t dist_strip_bytes = 0;          // interleaved: bytes of one rank's packed strip (the same on every rank)
    hipStream_t dist_stream = nullptr;
    hipEvent_t dist_begin[2] = {}, dist_rendered[2] = {}, dist_done[2] = {};
    bool dist_used[2] = {false, false};   // the slot's events have been recorded
    int dist_next = 0;                    // slot of the next gs_render_sharded_async
    int dist_recent[2] = {-1, -1};        // slots of the last and the last-but-one sharded frame
    void* dist_xchg = nullptr;            // gs_dist_rebalance: R x (tiles_y + 1) words
    std::vector<std::pair<double, double>> dist_history;   // ... (elements, ms) of every rank over the last epochs
};


// Text for gs_last_error(NULL) from translation units that have no context at hand (gs_dist.cpp); hidden: not an export.
__attribute__((visibility("hidden"))) void gsi_set_create_error(const std::string& msg);
// gs_dist.cpp: frees the buffers of a sharded frame (they are sized by the resolution); hidden: not an export.
__attribute__((visibility("hidden"))) void gsi_dist_free_buffers(gs_ctx* c);
// Renderer.cpp:458-475 for a frame enqueued with gs_render_device_async: wait, read the timestamps, fill gs_get_timings.
__attribute__((visibility("hidden"))) int gsi_finish_frame(gs_ctx* c);

namespace gs {

inline FrameParams make_frame_params(const gs_ctx* c, const float* view, const float* proj,
                              const float* cam_pos, uint32_t sh_mode) {
    FrameParams fp{};
    std::memcpy(fp.view, view, sizeof(fp.view));
    std::memcpy(fp.proj, proj, sizeof(fp.proj));
    std::memcpy(fp.cam_pos, cam_pos, sizeof(fp.cam_pos));
    fp.sh_mode = sh_mode;
    fp.width = c->width; fp.height = c->height;
    fp.grid_w = c->grid_w; fp.grid_h = c->grid_h;
    fp.row_begin = c->row_begin; fp.row_end = c->row_end;
    fp.row_stride = c->row_stride; fp.first_row = c->first_row; fp.rows_owned = c->rows_owned;
    fp.compact_out = c->compact_out ? 1u : 0u;
    fp.num_gaussians = c->n;
    fp.capacity = c->capacity;
    fp.near_plane = c->cfg.near_plane; fp.far_plane = c->cfg.far_plane;
    fp.ndc_cull = c->cfg.ndc_cull; fp.in_view_limit = c->cfg.in_view_limit;
    fp.tan_fov_y = (float)std::tan((double)(c->cfg.fov_y * 0.5f));   // Common.glsl:53, host-folded
    fp.hi16 = c->hi16 ? 1u : 0u;
    fp.parity = 0u;       // set by the InitSortList launch sites
    // |W|_2^2 <= min(trace, largest absolute row sum) of M = W^T W (Gershgorin); exactly 1 (+ rounding) for a rigid view
    double m[3][3], tr = 0.0, gersh = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            m[i][j] = 0.0;
            for (int k = 0; k < 3; ++k) m[i][j] += (double)view[i * 4 + k] * (double)view[j * 4 + k];
        }
    for (int i = 0; i < 3; ++i) {
        tr += m[i][i];
        gersh = std::max(gersh, std::fabs(m[i][0]) + std::fabs(m[i][1]) + std::fabs(m[i][2]));
    }
    fp.w_norm2 = (float)(std::min(tr, gersh) * (1.0 + 1e-5));
    return fp;
}

} // namespace gs
