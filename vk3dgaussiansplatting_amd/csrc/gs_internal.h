// gs_internal.h -- shared between the HIP kernels and the C-ABI host code of libgsplat_hip.so.
// gfx950 (MI355X / CDNA4) only: wave64, no other targets, no compatibility paths.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gs {

constexpr int kWave = 64;
constexpr int kTile = 16;              // TILE_SIZE, Common.glsl:12 / Renderer.h:146
constexpr int kRadixBits = 4;          // RS_BITS_PER_PASS, RadixSort.h:36
constexpr int kBins = 1 << kRadixBits; // RS_BIN_COUNT

// ---- radix-sort tiling -------------------------------------------------------------------
// One workgroup (256 threads = 4 waves) owns kSortTile consecutive keys of the current pass.
// The reference uses 64 keys per group (RS_WORK_GROUP_SIZE, RadixSort.h:38); a 2048-key group
// shrinks the histogram table 32x and makes every global access of a pass a >= 256-byte run.
// 8 keys per thread is the measured optimum on MI355X (12 and 16 keys per thread
// were 3-20 % slower) and what the 16-bit word paths are written for (8 keys per 16-byte load).
constexpr int kSortThreads = 256;
constexpr int kSortKeysPerThread = 8;
constexpr int kSortTile = kSortThreads * kSortKeysPerThread; // 2048 keys
constexpr int kSegments = 512;                // reduce segments = Count workgroups; each owns a contiguous run of groups
                                              // (measured: 512 beats 256 and 1024 at configs A-E)
constexpr int kCoarse = 64;                   // coarse reduce segments (kSegments / kCoarse segments each): second Reduce level
constexpr int kMaxSortPasses = 16;            // 64 key bits / 4
// Fed counts (gs_sort.hip, k_scatter<.., FED>): lists of at most this many groups may sort without per-pass Count
// launches -- every Scatter workgroup then sums the count rows of all groups itself (G x 64 bytes out of L2).
#ifndef GS_FED_MAX_GROUPS
#define GS_FED_MAX_GROUPS 1024    /* measured on MI355X (profiles/r06_fed_probe.txt): fed wins below ~1100 groups, by more the shorter the list */
#endif
constexpr uint32_t kFedMaxGroups = GS_FED_MAX_GROUPS;

// ---- GS_SORT_RADIX8*: the same sort with 8-bit digits (gs_sort8.hip) ------------------------
constexpr int kBins8 = 256;
#ifndef GS_SORT8_SMALL_BELOW
#define GS_SORT8_SMALL_BELOW 12000000   /* lists that cannot hold this many elements sort in 2048-key groups */
#endif
constexpr int kSort8Threads = 256;
constexpr int kSort8KeysPerThread = 16, kSort8KeysSmall = 8;
constexpr int kSort8Tile = kSort8Threads * kSort8KeysPerThread;   // keys per group: 4096 ...
constexpr int kSort8TileSmall = kSort8Threads * kSort8KeysSmall;  // ... or 2048 (launch_radix_sort8 chooses)
constexpr uint32_t kSort8SmallBelow = GS_SORT8_SMALL_BELOW;
static_assert(kBins8 <= kBins * kCoarse, "a pass's 256 digit totals live in its slab of SortBuffers::coarse");

// ---- InitSortList tiling -----------------------------------------------------------------
constexpr int kProjThreads = 256;      // splats per workgroup in project + emit

// Per-frame constants: CamUBO (ShaderStructs.h:37-41) + InitSortListPCD (7-12) + the shader
// #defines of Common.glsl:2-15.  Passed by value as a kernel argument.
struct FrameParams {
    float view[16];       // column-major
    float proj[16];
    float cam_pos[3];
    uint32_t sh_mode;     // integer, not float
    uint32_t width, height;
    uint32_t grid_w, grid_h;
    // Tile rows of this context (multi-GPU): rows first_row + k * row_stride < row_end, k = 0 .. rows_owned - 1, with
    // first_row = row_begin + row_phase.  One GPU: [0, grid_h), stride 1.  A contiguous band: [row_begin, row_end), stride 1.
    // Interleaved: every row_stride-th row of the whole grid starting at row_phase.  Inside a frame the sort list holds
    // COMPACT tile ids k * grid_w + x (k = index among the owned rows): same order as the global ids of the owned
    // tiles, no gaps, fewest key bits.
    uint32_t row_begin, row_end;
    uint32_t row_stride, first_row, rows_owned;
    uint32_t compact_out;         // image addressed in compact rows (pixel row of owned row k starts at 16 k): the strip
                                  // a rank contributes to the gather
    uint32_t num_gaussians;
    uint32_t capacity;
    float near_plane, far_plane;
    float ndc_cull, in_view_limit;
    float tan_fov_y;      // tan(FOV_Y*0.5f), folded on the host (Common.glsl:53)
    uint32_t hi16;        // sort list stores the compact tile ids as uint16 (at most 65535 owned tiles)
    float w_norm2;        // upper bound on the squared spectral norm of the upper-left 3x3 of view (host-folded, for the
                          // band bound; 1 for a rigid view matrix)
    uint32_t parity;      // InitSortList launches alternate between the two helper counters of SplatScratch
    uint32_t splat_first; // GS_SORT_RADIX4_SPLAT_FIRST: k_project also counts the emitting splats per block and leaves the
                          // helper records of k_emit to k_gather_sorted
};

// Device-side dispatch record: the role of RadixIndirectDispatch (ShaderStructs.h:45-57) +
// GaussianCullData (77-82).  Written by the scan kernel (IndirectSetup-equivalent), read by every
// later kernel, so the host never reads the element count back inside a frame.
struct SortParams {
    uint32_t num_elems;      // E' = min(counter, capacity)   (RadixSortIndirectSetup.comp:28)
    uint32_t num_groups;     // G  = ceil(E' / kSortTile)     (countSizeX)
    uint32_t groups_per_seg; // K  = ceil(G / kSegments)      (groups per reduce segment, <= 64)
    uint32_t overflow;       // counter > capacity
    uint64_t counter;        // un-truncated atomic-counter equivalent
    uint32_t pad[2];
};

// What RenderGaussians needs per splat, written once by the project kernel: the reference keeps color / covariance in
// the 336-byte record and redoes the screen position and the 2x2 inverse once per (tile, splat),
// RenderGaussians.comp:88-107; here that setup runs once per emitting splat -- same expressions, same operand order,
// so the same floats.  48 B, 16-aligned; RenderGaussians reads the first 36 bytes, the covariance rides along for
// gs_debug_read(GS_BUF_COV) at no extra traffic (the record is written in whole cache lines either way).
struct alignas(16) SplatRaster {
    float sx, sy;        // getScreenSpacePosition(...).xy                       RenderGaussians.comp:89-90
    float ix, iy, iz;    // gCovInv = (cz, -cy, cx) * (1 / det), 0 when det == 0  :94-107 (emitting splats only)
    float r, g, b;       // GaussianData.color.rgb                                :92
    float a;             // GaussianData.color.a, 0 when det == 0                 :92, :104
    float cx, cy, cz;    // GaussianData.covariance.xyz (raw, +0.3 dilation applied): every non-culled splat
};
static_assert(sizeof(SplatRaster) == 48, "SplatRaster must be 48 bytes");

// Device SoA image of the scene (converted once at upload from the 336-byte AoS records).
struct SceneBuffers {
    float* pos;      // [3][N] planes
    float* scale;    // [3][N]
    float* rot;      // [4][N]
    float* sh;       // [48][N], plane index = coeff*3 + channel
    float* opacity;  // [N]   shCoeffs[0].w
    float* sig2;     // [N]   upper bound of the largest eigenvalue of the 3-D covariance: |R|_F^2 * max(scale)^2,
                     //       computed at upload; lets a tile-row band skip far-away splats early (k_project)
    float* block_bounds; // [ceil(N / 64)][8]  per wave of 64 consecutive splats: min xyz, max xyz of their positions,
                     //       max sig2, pad -- a context with a subset of the tile rows drops a whole wave with one record
};

// Per-splat scratch of one frame.
struct SplatScratch {
    SplatRaster* raster;     // [N]
    uint32_t* depth_key;     // [N]
    uint32_t* tiles_touched; // [N]  0 for culled / zero-extent splats
    uint2* extents;          // [N]  packed u16: .x = minx | miny<<16, .y = maxx | maxy<<16 (row-clamped)
    uint8_t* wave_wrote;     // [4 * ceil(N/kProjThreads)]  1: this frame's k_project stored the raster records of the wave's
                             //       64 splats (else they are stale / zero: gs_debug_read zero-fills them)
    uint32_t* block_sums;    // [ceil(N/kProjThreads)]  tile counts per project workgroup
    uint32_t* block_offsets; // same size, exclusive scan
    // k_emit load balance: a project workgroup whose 256 splats emit more than kEmitSlice elements registers one
    // helper record {block, slice} per further slice; k_emit runs them as extra workgroups.
    uint2* help_list;        // [kEmitHelpCap]
    uint32_t* help_count;    // [4]: [p] helper records registered by the InitSortList launch of parity p, [2 + p] its band
                             // survivors (below); k_scan_blocks clears the pair of the next launch (FrameParams::parity)
    uint32_t* help_slot;     // [ceil(N/kProjThreads)]: first record of a heavy block, kEmitNoHelp if the list was full
    // A context that owns a subset of the tile rows: the project blocks k_band_cull could not reject, in arrival order,
    // block | skipped-waves mask << 28.
    uint32_t* band_list;     // [ceil(N/kProjThreads)]
    // GS_SORT_RADIX4_SPLAT_FIRST only (null otherwise)
    uint32_t* block_flags;   // [blocks, padded like block_sums]  emitting splats per project workgroup
    uint32_t* flag_offsets;  // same size, exclusive scan
    uint32_t* sorted_sums;   // [blocks, padded]  tile counts per 256 positions of the sorted splat list
    SortParams* aux_params;  // [2]: the splat list's dispatch record; scratch for the second scan
    uint32_t* elems_note;    // device alias of a pinned host word (or null): k_scan_blocks leaves the frame's element count + 1
                             // there, so that the host knows the length of recent lists without waiting for a frame
};
#ifndef GS_EMIT_SLICE
#define GS_EMIT_SLICE 4096
#endif
constexpr uint32_t kEmitSlice = GS_EMIT_SLICE;      // output elements per k_emit workgroup (a multiple of its 1024-element round)
constexpr uint32_t kEmitHelpCap = 65536;   // enough for 2^28 elements; beyond that the owner workgroup does the rest itself
constexpr uint32_t kEmitNoHelp = 0xFFFFFFFFu;
// helper workgroups of a k_emit launch = records k_project may register: the slices after the first number at most
// E / kEmitSlice <= capacity / kEmitSlice over all blocks (more only when the list overflows its capacity)
__host__ __device__ inline uint32_t emit_helpers(uint32_t capacity) {
    const uint32_t by_capacity = capacity / kEmitSlice + 1u;
    return by_capacity < kEmitHelpCap ? by_capacity : kEmitHelpCap;
}

struct SortBuffers {
    uint32_t *lo[2], *hi[2], *id[2]; // ping-pong: depth word, tile word, gaussian index; [capacity]
    uint32_t* table;                 // [16][G_max]  per-group digit counts (sumTable); 8-bit digits: [G8_max][256]
    uint32_t* seg_sum;               // [16][kSegments]  per-segment digit counts (reduce buffer); 8-bit digits:
                                     // [256][kSegments] counts, then [256][kSegments] scanned
    uint32_t* coarse;                // [kMaxSortPasses][16][kCoarse]  per-pass digit counts of the coarse segments
                                     // (8-bit digits: the first 256 words of a pass's slab = its digit totals)
    uint32_t* fed[3];                // 4-bit digits: [G_max][16] digit counts per group, three rotating sets of the fed
                                     // sort (k_scatter<.., FED>); null for 8-bit digits
    SortParams* params;
    uint32_t digit_bits;             // what alloc_sort sized table / seg_sum for (4 or 8): the launchers refuse the other width
};

// Bytes of the depth word a radix pass reads / writes per element (k_scatter<LO_IN, LO_OUT, HI16>); shared by the
// launcher and by the timing code that reports the bytes a pass moves.
inline void scatter_depth_bytes(uint32_t shift, uint32_t first_bit, bool drop_depth_payload, int* lo_in, int* lo_out,
                                uint32_t digit_bits = kRadixBits) {
    *lo_in = 4; *lo_out = 4;
    if (!drop_depth_payload) return;
    if (shift >= 32u) { *lo_in = 0; *lo_out = 0; return; }
    if (first_bit == 0u) {   // what the passes still to come read: nothing below bit shift + digit_bits
        *lo_in = shift >= 16u ? 2 : 4;
        *lo_out = shift + digit_bits >= 32u ? 0 : (shift + digit_bits >= 16u ? 2 : 4);
    }
}

// ---- launchers (each enqueues on `stream`, no host sync) ------------------------------------
void launch_project(const FrameParams& fp, const SceneBuffers& scene, const SplatScratch& sc,
                    hipStream_t stream);
// sums == nullptr: block_sums -> block_offsets, the frame's clears, the IndirectSetup record in `params`.
void launch_debug_colour(const FrameParams& fp, const SceneBuffers& scene, const SplatScratch& sc, float* out_rgba, hipStream_t stream);
void launch_scan_blocks(const FrameParams& fp, const SplatScratch& sc, SortParams* params,
                        uint32_t* ranges, uint32_t* coarse, hipStream_t stream);
void launch_emit(const FrameParams& fp, const SplatScratch& sc, const SortBuffers& sb,
                 hipStream_t stream);
// GS_SORT_RADIX4_SPLAT_FIRST (gs_project.hip): the (depth word | tile count | splat) list of the emitting splats into
// sort buffers [1] (lo, hi, id), with its dispatch record in sc.aux_params[0]; then, once that list is sorted by depth
// (buffers [sorted]), the sums of its tile counts + their scan; then the emit in depth order into buffers [0].
void launch_splat_list(const FrameParams& fp, const SplatScratch& sc, const SortBuffers& sb, hipStream_t stream);
void launch_gather_sorted(const FrameParams& fp, const SplatScratch& sc, const SortBuffers& sb, int sorted, hipStream_t stream);
void launch_emit_sorted(const FrameParams& fp, const SplatScratch& sc, const SortBuffers& sb, int sorted, hipStream_t stream);
// Sorts buffers [0] -> result index returned (0 or 1) after num_sort_bits/4 passes.
// scatter_events: optional 2*passes events recorded right before / after every Scatter launch.
// Passes run over key bits [first_bit, num_sort_bits) of tile << 32 | depth (in a frame the tile word is the compact
// tile id of FrameParams: a context that owns a subset of the tile rows sorts over fewer significant bits).
// drop_depth_payload: the tile-word passes (bits >= 32) do not carry the depth words (frame path only).
// hi16: the hi arrays hold 16-bit tile ids (frame path, at most 65535 owned tiles).
// start: the ping-pong buffer the list lies in; coarse_pass: first slab of sb.coarse to use (one per pass; two sorts
// in one frame must not share slabs); params: dispatch record of this list (default sb.params).
// fed: 4-bit digits only -- one Count launch for the whole sort, every Scatter feeds the next pass's counts (k_scatter<.., FED>);
// same output, meant for lists of at most kFedMaxGroups groups (correct, but slow, beyond).
// Returns -1 without launching anything when digit_bits is not the width sb was allocated for.
int launch_radix_sort(const SortBuffers& sb, uint32_t capacity, uint32_t num_sort_bits,
                      hipStream_t stream, hipEvent_t* scatter_events = nullptr, uint32_t first_bit = 0,
                      bool drop_depth_payload = false, bool hi16 = false, float share = 1.0f,
                      int start = 0, uint32_t coarse_pass = 0, const SortParams* params = nullptr,
                      uint32_t digit_bits = kRadixBits, bool fed = false);
// the same with 8-bit digits (gs_sort8.hip); launch_radix_sort forwards here when digit_bits == 8
int launch_radix_sort8(const SortBuffers& sb, uint32_t capacity, uint32_t num_sort_bits, hipStream_t stream,
                       hipEvent_t* scatter_events, uint32_t first_bit, bool drop_depth_payload, bool hi16, float share,
                       int start, uint32_t coarse_pass, const SortParams* params);
// GS_SORT_TILE_BUCKET: per-tile depth sort of the owned tiles (gs_tilesort.hip)
int init_tile_sort();
void launch_tile_sort(const FrameParams& fp, const uint32_t* ranges, uint32_t* lo, uint32_t* id,
                      uint32_t* lo_alt, uint32_t* id_alt, hipStream_t stream, hipStream_t helper = nullptr,
                      hipEvent_t fork = nullptr, hipEvent_t join = nullptr);
void launch_find_ranges(const FrameParams& fp, const uint32_t* sorted_tile, const SortParams* params,
                        uint32_t* ranges, hipStream_t stream);
// order[k] = the k-th tile RenderGaussians dispatches, as an index among the context's own tiles, longest list first
void launch_tile_order(const FrameParams& fp, const uint32_t* ranges, uint32_t* order, hipStream_t stream);
size_t tile_order_words(uint32_t grid_w, uint32_t grid_h);   // words `order` must hold: the table and the kernels' scratch
// order == nullptr: raster order
void launch_render(const FrameParams& fp, const SplatRaster* raster, const uint32_t* sorted_id,
                   const uint32_t* ranges, const uint32_t* order, uint8_t* rgba, uint32_t render_mode,
                   uint32_t render_kernel, hipStream_t stream);
void launch_aos_to_soa(const float* chunk, uint32_t first, uint32_t count, uint32_t n,
                       const SceneBuffers& s, hipStream_t stream);
void launch_block_bounds(uint32_t n, const SceneBuffers& s, hipStream_t stream);
void launch_stream_probe(int kind, const void* src, void* dst, size_t bytes, uint32_t blocks, hipStream_t stream);
// helpers for the stand-alone sorter entry points
void launch_set_sort_params(SortParams* params, uint32_t* coarse, uint32_t n, hipStream_t stream);
void launch_fill_random_keys(uint32_t* lo, uint32_t* hi, uint32_t* id, uint32_t n,
                             uint32_t num_tiles, uint64_t seed, hipStream_t stream);
void launch_check_sorted(const uint32_t* lo, const uint32_t* hi, uint32_t n, uint32_t* bad_count,
                         hipStream_t stream);

} // namespace gs
