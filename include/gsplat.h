/*
 * gsplat.h -- C-ABI of libgsplat_hip.so, the MI355X-native replacement for the per-frame splat
 * path of SiTronXD/vk3dGaussianSplatting (InitSortList -> 4-bit radix sort over 64-bit
 * tile|depth keys -> FindRanges -> RenderGaussians).
 *
 * The reference has no FFI: the seam this library replaces is the public surface of its
 * `Renderer` class plus the `GpuSort` plug-in interface.  Each entry point cites the reference
 * interface it stands in for; paths are relative to /root/reference/vkGaussianSplatting/.
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only; every function returns an int status
 * (0 = GS_OK, > 0 = warning, < 0 = error) and never aborts or throws (the reference reports
 * errors through a modal MessageBox and continues, Engine/Dev/Log.cpp:40-43); the caller owns
 * every host pointer for the duration of the call only; the library owns all device memory.
 * A gs_ctx is bound to one GPU and one HIP stream and is not thread-safe; independent contexts
 * may be used concurrently.  Matrices are column-major float[16] exactly as glm::mat4 lies in
 * memory (CamUBO, Engine/Graphics/ShaderStructs.h:37-41).
 */
#ifndef GSPLAT_H
#define GSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_API_VERSION 5   /* 5: gs_config grew count_launches (GS_COUNT_*).  2: gs_config grew tile_order; 3: gs_config starts with struct_size, gs_api_version(),
                              gs_runtime_versions(), gs_dist_* / gs_gather_strips; 4: GS_ROWS_BALANCED + gs_dist_rebalance /
                              gs_dist_bands, gs_render_sharded_async / gs_sharded_frame / gs_sharded_read (two sharded
                              frames in flight, the assembled frame left in HBM), GS_BUF_COLOR for every non-culled splat */

/* status codes */
#define GS_OK 0
#define GS_WARN_OVERFLOW 1      /* more sort elements than capacity; list truncated like InitSortList.comp:140-148 */
#define GS_ERR_INVALID (-1)     /* bad argument / call order */
#define GS_ERR_HIP (-2)         /* HIP runtime failure, see gs_last_error */
#define GS_ERR_NO_SCENE (-3)    /* gs_upload_gaussians / gs_set_resolution not called yet */
#define GS_ERR_IO (-4)          /* file cannot be opened (ResourceManager.cpp:169-173) */
#define GS_ERR_FORMAT (-5)      /* .ply header/property problem */
#define GS_ERR_NO_DEVICE (-6)   /* no usable GPU: the library has NO CPU fallback */

/* The reference's 336-byte record, Engine/Graphics/ShaderStructs.h:59-70:
 * position.xyz0, scale.xyz0 (already exp'd), rot (permuted unit quaternion, .x scalar part),
 * shCoeffs[16] ([0].w = sigmoid opacity), color, covariance (scratch, ignored on upload). */
#define GS_GAUSSIAN_RECORD_BYTES 336u

/* sort back-ends (the GpuSort seam, Engine/Graphics/Sort/GpuSort.h:8-22, selected at compile
 * time in the reference by GPU_SORT_ALGORITHM, Renderer.h:33) */
#define GS_SORT_RADIX4 0u       /* the contractual nine-stage 4-bit LSD radix sort over all key bits (default) */
#define GS_SORT_TILE_BUCKET 1u  /* alternative back-end, identical output: the global 4-bit passes sort by the tile word
                                   only, then every tile's run is depth-sorted inside LDS (csrc/gs_tilesort.hip) */
#define GS_SORT_RADIX4_SPLAT_FIRST 2u /* the same twelve 4-bit passes in another order, identical output: the depth word
                                     of a key is a property of the SPLAT, so the eight passes over it run on the
                                     (depth, splat) list of the emitting splats BEFORE a splat is replicated into its
                                     tiles; InitSortList's emit then walks the splats in depth order and the four
                                     tile-word passes -- stable -- finish the order by (tile, depth).  Moves a third of
                                     the bytes.  Not the default: the default keeps the reference's stage order. */
#define GS_SORT_RADIX8 3u       /* the A/B slot for an 8-bit-digit variant: the same stable LSD sort over all key bits with
                                   half the passes (Count, Scan, Scatter per pass; csrc/gs_sort8.hip), identical output */
#define GS_SORT_RADIX8_SPLAT_FIRST 4u /* GS_SORT_RADIX4_SPLAT_FIRST's stage order with the 8-bit passes */

/* render arithmetic */
#define GS_RENDER_EXACT 0u      /* bit-identical to the CPU oracle (no contraction, pinned exp) */
#define GS_RENDER_FAST 1u       /* fused multiply-adds + hardware exp2; <= 1 step per 8-bit channel */

typedef struct gs_ctx gs_ctx;

/* Replaces the compile-time constants of Renderer.h:145-147, RadixSort.h:36-39,
 * Camera.cpp:4-5 and Resources/Shaders/Common/Common.glsl:2-15. */
/* RenderGaussians launch shapes (all bit-identical in GS_RENDER_EXACT).  AUTO = GS_RENDER_KERNEL_WORKGROUP_8X8 (a workgroup
 * per tile, an 8 x 8 quadrant per wave) at every size. */
#define GS_RENDER_KERNEL_AUTO 0u
#define GS_RENDER_KERNEL_WAVE_1PX 1u   /* four independent waves per tile, 1 pixel per lane */
#define GS_RENDER_KERNEL_WAVE_2PX 2u   /* two independent waves per tile, 2 pixels per lane */
#define GS_RENDER_KERNEL_WAVE_4PX 4u   /* one wave per tile, 4 pixels per lane */
#define GS_RENDER_KERNEL_WORKGROUP 16u /* one 256-thread workgroup per tile, one pixel per lane (the reference's launch shape);
                                          its four waves walk the tile's list independently, each for a 16 x 4 strip */
#define GS_RENDER_KERNEL_WORKGROUP_8X8 17u /* the same with an 8 x 8 quadrant per wave */

/* Order in which RenderGaussians' tiles are dispatched (same pixels either way). */
#define GS_TILE_ORDER_LONGEST_FIRST 0u /* by list length, longest first: one small launch behind FindRanges (default) */
#define GS_TILE_ORDER_RASTER 1u        /* row-major, like the reference's dispatch (Subrenderer.cpp:330-333) */

/* The Count stage of the 4-bit radix sort (GS_SORT_RADIX4; RadixSortCount.comp:40-91): one launch per pass, or one launch per
 * SORT with every Scatter launch counting the next pass's digits of the keys it stores ("fed" counts: for short lists -- a
 * tile-row band of a multi-GPU frame, a small frame -- where a pass is two fixed launch latencies and little else).  Same
 * sorted list either way. */
#define GS_COUNT_AUTO 0u      /* fed when a recent frame held at most 1024 groups of 2048 elements = 2.1 M (default) */
#define GS_COUNT_PER_PASS 1u  /* always a Count launch per pass */
#define GS_COUNT_FED 2u       /* always fed (correct at every size; slow for long lists) */

typedef struct gs_config {
    uint32_t struct_size;     /* sizeof(gs_config) as the CALLER's header declares it (gs_default_config fills it in).
                                 gs_create copies that many bytes and keeps its defaults for fields the caller's
                                 header does not know yet; a struct larger than the library's is refused. */
    int32_t device_ordinal;   /* HIP device index */
    uint32_t tile_size;       /* 16; only 16 is supported (TILE_SIZE) */
    float near_plane;         /* 0.1f   Camera::NEAR_PLANE */
    float far_plane;          /* 100.0f Camera::FAR_PLANE */
    float ndc_cull;           /* 1.3f   CULLING_NDC_LIMIT */
    float in_view_limit;      /* 0.8f   IN_VIEW_LIMIT */
    float fov_y;              /* 3.1415f*0.5f FOV_Y (the covariance one, not the projection's) */
    uint32_t sort_algorithm;  /* GS_SORT_* */
    uint32_t render_mode;     /* GS_RENDER_* */
    uint32_t record_timings;  /* 0 (default, like the reference's commented-out RECORD_GPU_TIMES, GfxSettings.h:7) = no
                                 events; 1 = hipEvents at the reference's 7 timestamp points (Renderer.cpp:557-622);
                                 2 = additionally one event pair around every Scatter launch (roofline measurement) */
    uint32_t render_kernel;   /* GS_RENDER_KERNEL_*: how a tile maps to waves in RenderGaussians; same pixels either way */
    uint32_t tile_order;      /* GS_TILE_ORDER_* */
    uint32_t count_launches;  /* GS_COUNT_*: GS_SORT_RADIX4 frames and the stand-alone 4-bit sorter; ignored by the other sorters */
} gs_config;

/* The five buckets of Renderer.cpp:471-475 (ms) + the sort element count ("Elements To Sort"
 * in README.md:43-93).  InitSortList includes the per-frame clears, RadixSort includes the
 * IndirectSetup-equivalent, exactly like the reference's timestamp placement (Renderer.cpp:557-622).
 * With GS_SORT_RADIX4_SPLAT_FIRST the two stages interleave (project + splat list | depth passes | sums + emit |
 * tile-word passes): init_sort_list_ms and radix_sort_ms are the sums of their two halves, and with
 * record_timings == 2 scatter_* describe the depth passes over the SPLAT list (bytes per splat), scatter_tile_* the
 * tile-word passes over the elements. */
typedef struct gs_timings {
    float init_sort_list_ms;
    float radix_sort_ms;
    float find_ranges_ms;
    float render_ms;
    float total_ms;
    uint32_t num_sort_elements;   /* min(counter, capacity) */
    uint32_t overflowed;          /* counter > capacity this frame */
    uint64_t emitted_elements;    /* un-truncated counter */
    float scatter_ms_avg;         /* record_timings == 2: mean duration of one Scatter launch that moves key + payload
                                     (24 bytes per element; the dominant kernel) */
    uint32_t scatter_launches;    /* number of those launches in the frame (the 8 depth-word passes; all passes
                                     with GS_SORT_TILE_BUCKET) */
    float scatter_tile_ms_avg;    /* mean duration of a tile-word pass of the frame path, which leaves the already
                                     sorted depth words behind (16 bytes per element) */
    uint32_t scatter_tile_launches;
    float scatter_bytes_per_elem;       /* bytes per element really moved (read + written) by the launches above, mean */
    float scatter_tile_bytes_per_elem;
} gs_timings;

/* RECORD_CPU_TIMES (GfxSettings.h:6; Renderer.cpp:299-314, 343-352, 399-456): host-side milliseconds of the last
 * gs_render* call, always recorded (four clock reads per frame).  The reference's four figures map to:
 * waitForFence -> wait_ms (host blocked until the GPU has finished the frame; 0 for the async variant),
 * recordCommandBuffer -> record_ms (enqueueing the frame's launches), present -> present_ms (the copy of the frame to
 * the host in gs_render; 0 for the device variants), CPU frame time -> cpu_frame_ms (entry of this call minus entry of
 * the previous one). */
typedef struct gs_host_timings {
    float wait_ms;
    float record_ms;
    float present_ms;
    float cpu_frame_ms;
} gs_host_timings;

/* Scene-derived sizes: Renderer.cpp:696-701 (tiles), :725 (capacity), RadixSort.cpp:203-204 (bits). */
typedef struct gs_scene_info {
    uint32_t num_gaussians;
    uint32_t width, height;
    uint32_t tiles_x, tiles_y;
    uint32_t capacity;            /* C = ceilPow2(N + 64*16*T) */
    uint32_t num_sort_bits;       /* 4 * P */
    uint32_t row_begin, row_end;  /* tile-row band rendered by this context */
    uint32_t tile_word_bytes;     /* how the frame's sort list stores a tile id: 2 (uint16 index among the context's own
                                     tiles, at most 65535 of them) or 4; callers always see uint32 global ids */
    uint32_t row_stride;          /* the context renders tile rows first_row + k * row_stride < row_end, k < rows_owned */
    uint32_t first_row;
    uint32_t rows_owned;
} gs_scene_info;

/* buffers readable through gs_debug_read (state after the last gs_render*) */
#define GS_BUF_SORTED_TILE 0   /* uint32[num_sort_elements]  high half of the key   */
#define GS_BUF_SORTED_DEPTH 1  /* uint32[num_sort_elements]  low half of the key    */
#define GS_BUF_SORTED_ID 2     /* uint32[num_sort_elements]  gaussian index payload */
#define GS_BUF_RANGES 3        /* uint32[tiles][2]           {start,end} per tile (GaussianTileRangeData.xy) */
#define GS_BUF_COLOR 4         /* float[N][4]                GaussianData.color: rgb + opacity of EVERY splat that passed both culls
                                                             (InitSortList.comp:124-127), zero for the culled ones.  A frame evaluates
                                                             the colour of the splats that emit an element (nothing else is ever read);
                                                             the others are evaluated by this call, on demand, from the last frame's
                                                             camera.  A context that owns a SUBSET of the tile rows (gs_set_tile_rows*)
                                                             does not project splats that cannot reach its rows: there the buffer
                                                             holds the emitting splats only */
#define GS_BUF_COV 5           /* float[N][4]                GaussianData.covariance (w = 0) */
#define GS_BUF_COUNT 6         /* uint64[1]                  un-truncated element counter */
#define GS_BUF_UNSORTED_TILE 7 /* uint32[num_sort_elements]  list as emitted by InitSortList */
#define GS_BUF_UNSORTED_DEPTH 8
#define GS_BUF_UNSORTED_ID 9
#define GS_BUF_IMAGE 10        /* uint8[H][W][4]             internal framebuffer */

/* GS_API_VERSION of the library that is actually loaded: compare with the header's before anything else
 * (gsplat.hpp and the Python binding do). */
uint32_t gs_api_version(void);
/* HIP_VERSION the library was compiled against, and the HIP runtime / driver versions of the libamdhip64 the process
 * really bound (a process holds ONE HIP runtime -- INTEGRATION.md, "One HIP runtime per process"); any pointer may be NULL. */
int gs_runtime_versions(int* hip_build, int* hip_runtime, int* hip_driver);

void gs_default_config(gs_config* cfg);

/* Renderer::init (Renderer.cpp:688-694) + GpuSort::singleInitResources (RadixSort.cpp:23-142). */
int gs_create(const gs_config* cfg, gs_ctx** out);
/* Renderer::cleanup (Renderer.cpp:230-270) + RadixSort::cleanup (RadixSort.cpp:655-674).  Waits for the context's
 * stream, frees everything the context owns and the context itself, and returns GS_OK -- always: there is no failure
 * that leaves the handle alive, so it must not be used (not even for gs_last_error) once this has been called.
 * NULL is accepted.  A scene shared through gs_share_scene lives on while another context holds it. */
int gs_destroy(gs_ctx* ctx);
/* Text of the last error on ctx (ctx may be NULL: last gs_create failure). Log::error, Dev/Log.cpp:40-43. */
const char* gs_last_error(const gs_ctx* ctx);

/* Renderer::initForScene, gaussian upload (Renderer.cpp:712-724): n records of
 * GS_GAUSSIAN_RECORD_BYTES each, as ResourceManager::getGaussians() returns them
 * (ResourceManager.h:53).  Converted once to the device SoA layout. */
int gs_upload_gaussians(gs_ctx* ctx, const void* aos336, uint32_t n);
/* Frames in flight (the reference keeps GfxSettings::FRAMES_IN_FLIGHT = 3 command buffers, GfxSettings.h:15,
 * Renderer.cpp:304-310, 514): `ctx` renders the gaussians already uploaded to `owner` -- the read-only arrays
 * are shared, everything a frame writes (sort lists, ranges, raster records, image, stream, timings) stays per
 * context, so F contexts on F streams keep F frames in flight on one GPU.  Follow with gs_set_resolution.
 * The arrays are reference-counted inside the library: the contexts may be destroyed in any order, and a new
 * gs_upload_gaussians / gs_load_ply on one of them leaves the others rendering the arrays they hold. */
int gs_share_scene(gs_ctx* ctx, gs_ctx* owner);

/* ResourceManager::loadGaussians (ResourceManager.cpp:167-300): .ply with the INRIA property names -> records
 * (axis flips, exp, quaternion permutation, sigmoid, SH repack, Morton order) -> upload.  binary_little_endian is the
 * fast path (two sweeps over 16 MB chunks: 2.3 s for a 1.45 GB Garden-size file); binary_big_endian and ascii are
 * accepted; ascii is tokenised twice (chunked std::from_chars: about 0.3 GB of text per second over both sweeps).  The path
 * must name a seekable file. */
int gs_load_ply(gs_ctx* ctx, const char* path);
/* The same conversion without a context: writes up to max_records records to aos336_out (may be
 * NULL to query) and the record count to n_out. */
int gs_convert_ply(const char* path, void* aos336_out, uint32_t max_records, uint32_t* n_out);
/* Text of the last gs_load_ply / gs_convert_ply failure on this thread (happly throws instead,
 * happly.h:1089-1094). */
const char* gs_ply_last_error(void);

/* Frame output sink: where the reference's frame goes to the swapchain image
 * (RenderGaussians.comp:150 imageStore, presented by Renderer.cpp:341-397) a windowless host writes the
 * RGBA8 frame of gs_render to disk.  Format by extension: ".ppm" (binary P6, alpha dropped) or ".png"
 * (8-bit RGBA).  GS_ERR_IO when the file cannot be written, GS_ERR_INVALID for another extension. */
int gs_write_image(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height);

/* Swapchain extent -> tile grid, list capacity and pass count (Renderer.cpp:696-701, 725-755;
 * RadixSort::initForScene, RadixSort.cpp:144-205).  Must follow gs_upload_gaussians. */
int gs_set_resolution(gs_ctx* ctx, uint32_t width, uint32_t height);
/* Multi-GPU extension (no reference counterpart): this context emits/sorts/renders only tile rows
 * [row_begin,row_end) with GLOBAL tile ids, so keys, per-tile order and pixels equal the 1-GPU
 * result.  Default after gs_set_resolution: all rows. */
int gs_set_tile_rows(gs_ctx* ctx, uint32_t row_begin, uint32_t row_end);
/* Same extension, interleaved: this context owns tile rows phase, phase + stride, phase + 2 stride, ... of the whole
 * grid (rank r of R: phase = r, stride = R), which evens out the per-rank load when the splat density varies over the
 * height of the frame.  compact_output != 0: gs_render_device* address the image in compact rows -- the pixel rows of
 * the k-th owned tile row start at row 16 k -- i.e. they write the strip a rank contributes to the gather
 * (ceil(rows_owned) * 16 rows of `width` pixels); gs_render (host image) always writes the real rows. */
int gs_set_tile_rows_interleaved(gs_ctx* ctx, uint32_t phase, uint32_t stride, uint32_t compact_output);
int gs_get_scene_info(const gs_ctx* ctx, gs_scene_info* out);

/* Multi-GPU extension, the exchange step (no reference counterpart): one process per GPU, every
 * rank renders its tile rows into a strip (gs_render_device* with gs_set_tile_rows*), and the strips meet on the
 * root over RCCL -- point-to-point transfers over xGMI inside a node, enqueued on the context's stream behind the
 * frame that wrote the strip.  RCCL is bound at gs_dist_init (librccl.so.1 by name, or the copy the process
 * already holds; the environment variable GS_RCCL_LIBRARY names another build by path), not at link time.
 *   gs_dist_unique_id : ncclGetUniqueId -- call on ONE rank, hand the GS_DIST_UNIQUE_ID_BYTES bytes to the others by
 *                       any means (pipe, file, MPI, torch.distributed broadcast);
 *   gs_dist_init      : ncclCommInitRank for this context's GPU; collective over all `world` ranks;
 *   gs_gather_strips  : strip_dev = this rank's `bytes` bytes on the device (the SAME `bytes` and `root` on every
 *                       rank: strips are padded to one size); gathered_dev = world * bytes bytes on the root's device
 *                       (rank r's strip at offset r * bytes; ignored on other ranks, may be NULL there).  Every rank must
 *                       call it, in the same order as the others.  Asynchronous: ordered on the context's stream, wait
 *                       with gs_synchronize;
 *   gs_dist_destroy   : ncclCommDestroy (gs_destroy does it too). */
#define GS_DIST_UNIQUE_ID_BYTES 128
int gs_dist_unique_id(void* id_out);
int gs_dist_init(gs_ctx* ctx, const void* unique_id, int rank, int world);
int gs_gather_strips(gs_ctx* ctx, const void* strip_dev, void* gathered_dev, size_t bytes, int root);
int gs_dist_destroy(gs_ctx* ctx);
/* The same for a host that holds no device memory of its own -- Renderer::draw on R GPUs (Renderer.cpp:297-515):
 *   gs_dist_shard_rows : after gs_set_resolution + gs_dist_init (again after every gs_set_resolution, and after any
 *                        gs_set_tile_rows* of the caller's own: the buffers of the sharded frame belong to the rows dealt
 *                        HERE, and gs_render_sharded* refuse a context whose rows have been changed since).  Rank r of R
 *                        takes its tile rows --
 *                          GS_ROWS_CONTIGUOUS  a band of ceil(Ty / R) rows;
 *                          GS_ROWS_INTERLEAVED rows r, r + R, ...: even load whatever the scene, at the price of
 *                                              InitSortList's block cull (every rank's rows span the whole frame);
 *                          GS_ROWS_BALANCED    contiguous bands whose EDGES follow the scene: equal row counts at first,
 *                                              moved by gs_dist_rebalance towards equal cost --
 *                        and the library allocates the buffers of two sharded frames in flight (every rank its rows; rank 0
 *                        the assembled frames).
 *   gs_render_sharded_async : every rank calls it with the same camera: the rank's rows are rendered on the context's
 *                        stream; the exchange follows on a stream of the library's own behind an event, so it runs beside
 *                        the NEXT frame's kernels; bands land in place in rank 0's frame (interleaved rows: re-ordered by R
 *                        strided copies).  Returns after enqueueing; the assembled frame stays in rank 0's HBM.  A rank
 *                        whose own frame fails still takes part in the exchange before it returns its error, so the others
 *                        are not left waiting.
 *   gs_sharded_frame   : waits (host) until the exchange of the last (which = 0) or the last-but-one (which = 1) sharded
 *                        frame has finished; on rank 0 *frame_dev is the device pointer of that frame (height * width * 4
 *                        bytes, valid until the second gs_render_sharded* call from now), NULL on the other ranks.
 *   gs_sharded_read    : the same, and copies the frame to rgba_out (HOST) on rank 0.
 *   gs_render_sharded  : gs_render_sharded_async + gs_sharded_read(0) + the timings of this rank's own rows
 *                        (gs_get_timings, with record_timings on): the synchronous form.  rgba_out: HOST, height*width*4
 *                        bytes as in gs_render; ignored on the other ranks, may be NULL.  The frame is bit-identical to the
 *                        one-GPU frame of gs_render.
 *   gs_dist_rebalance  : GS_ROWS_BALANCED, collective (every rank, same order as the frames, e.g. every 32 frames): each
 *                        rank contributes the sort-element counts of its tile rows (the last frame's tile ranges) and the
 *                        GPU time of its share; after one exchange of R (R - 1) messages of Ty + 1 words every rank holds
 *                        the same vectors and derives the same new edges: weight(row) = elements(row) x (share time / share
 *                        elements of the rank that rendered it), edges at equal weight prefixes, moved only for a predicted
 *                        gain of 3 % on the slowest rank.  Waits for the frames in flight.  *moved_out (may be NULL) = 1
 *                        when the edges moved (the next frame re-captures its hipGraph).  With GS_REBALANCE_ELEMENTS_ONLY in the
 *                        environment (of every rank) the share times are ignored and the bands are cut by element counts
 *                        alone: for ranks whose times are not comparable, e.g. several ranks sharing one GPU.
 *   gs_dist_bands      : the R + 1 band edges (tile rows) of a contiguous or balanced dealing; count must be R + 1.
 * Collective safety: these calls pair with the other ranks' calls.  A rank that fails a PRECONDITION (no shard, rows changed
 * behind the library's back, a NULL rgba_out on rank 0, an allocation that failed in gs_dist_shard_rows) returns before the
 * exchange and its peers wait in theirs: check the status of gs_dist_shard_rows on every rank (e.g. exchange a flag over the
 * channel that carried the unique id) before the first frame.  A failure of the rank's own FRAME is collective-safe (above), and so
 * is a HIP error of the calls around the exchange (the event / copy calls of gs_render_sharded_async and gs_dist_rebalance once
 * the preconditions have passed): the rank still takes part in the exchange and reports its first error afterwards. */
#define GS_ROWS_CONTIGUOUS 0u
#define GS_ROWS_INTERLEAVED 1u
#define GS_ROWS_BALANCED 2u
int gs_dist_shard_rows(gs_ctx* ctx, uint32_t dealing);
int gs_render_sharded_async(gs_ctx* ctx, const float view[16], const float proj[16], const float cam_pos[3], uint32_t sh_mode);
int gs_sharded_frame(gs_ctx* ctx, uint32_t which, void** frame_dev);
int gs_sharded_read(gs_ctx* ctx, uint32_t which, uint8_t* rgba_out);
int gs_render_sharded(gs_ctx* ctx, const float view[16], const float proj[16], const float cam_pos[3],
                      uint32_t sh_mode, uint8_t* rgba_out);
int gs_dist_rebalance(gs_ctx* ctx, uint32_t* moved_out);
int gs_dist_bands(const gs_ctx* ctx, uint32_t* edges_out, uint32_t count);
/* The partition rule by itself (pure host arithmetic, no context, no GPU): world + 1 edges of contiguous bands over
 * tiles_y rows whose weights are as equal as whole rows allow -- edge r at the row boundary whose weight prefix is nearest
 * to r / world of the total, every band at least one row while there are rows to give; all-zero or non-finite weights
 * count as equal ones.  For hosts that run their own exchange (bench.py's torch path: dist.balanced_row_partition is the
 * same rule in Python). */
int gs_balance_rows(const double* row_weights, uint32_t tiles_y, uint32_t world, uint32_t* edges_out);

/* Renderer::draw (Renderer.cpp:297-515): updateUniformBuffer(view, proj) (:531-538), the push
 * constants of Subrenderer.cpp:152-160 (camPos, shMode as an INTEGER 0/1/2), then the recorded
 * frame (:540-629).  rgba_out: height*width*4 bytes on the HOST, row-major, top row first,
 * R,G,B,A with A = 255 (the R8G8B8A8_UNORM storage image of Swapchain.cpp:27).  Synchronous. */
int gs_render(gs_ctx* ctx, const float view[16], const float proj[16], const float cam_pos[3],
              uint32_t sh_mode, uint8_t* rgba_out);
/* Same frame, image left in HBM: rgba_out_device is a DEVICE pointer of height*width*4 bytes
 * (e.g. a torch tensor's data_ptr) or NULL for the internal framebuffer (GS_BUF_IMAGE).
 * Enqueued on the context's stream; returns after the frame has completed (timings valid). */
int gs_render_device(gs_ctx* ctx, const float view[16], const float proj[16],
                     const float cam_pos[3], uint32_t sh_mode, void* rgba_out_device);
/* As gs_render_device but returns right after enqueueing (no host sync, timings not updated):
 * the CPU never waits on the GPU inside a frame, like the reference's single vkQueueSubmit. */
int gs_render_device_async(gs_ctx* ctx, const float view[16], const float proj[16],
                           const float cam_pos[3], uint32_t sh_mode, void* rgba_out_device);
/* Wait for everything enqueued on the context's stream (vkDeviceWaitIdle, Renderer.cpp:459). */
int gs_synchronize(gs_ctx* ctx);

/* computeDiffs buckets of the last synchronous frame (Renderer.cpp:463-475). */
int gs_get_timings(const gs_ctx* ctx, gs_timings* out);
/* RECORD_CPU_TIMES figures of the last gs_render* call (Renderer.cpp:399-456). */
int gs_get_host_timings(const gs_ctx* ctx, gs_host_timings* out);

/* No reference counterpart (its buffers are only visible in a GPU debugger): copies one device
 * buffer of the last frame to dst; bytes must not exceed the buffer's size. */
int gs_debug_read(gs_ctx* ctx, int which, void* dst, size_t bytes);

/* Runs ONLY the InitSortList stage of a frame (project + count scan + emit) and waits; afterwards
 * GS_BUF_UNSORTED_*, GS_BUF_COLOR, GS_BUF_COV and GS_BUF_COUNT are readable (stage-level parity). */
int gs_debug_init_sort_list(gs_ctx* ctx, const float view[16], const float proj[16],
                            const float cam_pos[3], uint32_t sh_mode);

/* Use a caller-owned hipStream_t (passed as void*) instead of the context's own stream.  NULL switches back
 * to the context's own (non-blocking) stream: the legacy default stream has handle 0 and cannot be selected --
 * a caller that works on it must create a stream of its own and hand that over. */
int gs_set_stream(gs_ctx* ctx, void* hip_stream);

/* Camera::updateDirVectors + updateMatrices (Engine/Graphics/Camera.cpp:7-48): yaw/pitch/position
 * -> view (glm::lookAt) and proj (glm::perspective(radians(90), aspect, near, far), depth 0..1). */
int gs_camera_matrices(const float pos[3], float yaw, float pitch, float aspect, float near_plane,
                       float far_plane, float view_out[16], float proj_out[16]);

/* GpuSort seam used stand-alone (GpuSort.h:8-22: initForScene + gpuClearBuffers + computeSort on
 * caller data): sorts n (tile, depth, id) triples held in HOST arrays, stable, by the low
 * num_sort_bits of (tile<<32 | depth), with the same kernels the frame uses.  In place. */
int gs_sort_host(gs_ctx* ctx, uint32_t* tile, uint32_t* depth, uint32_t* id, uint32_t n,
                 uint32_t num_sort_bits);
/* Stress run for the sorter alone (BASELINE config E): n random triples generated on the device
 * (seeded), sorted `iters` times; returns the mean milliseconds of one full sort and verifies
 * sortedness on the device (sorted_ok = 1). */
int gs_sort_bench(gs_ctx* ctx, uint32_t n, uint32_t num_tiles, uint32_t iters, uint64_t seed,
                  float* ms_per_sort, uint32_t* sorted_ok);

/* Stream-bandwidth probe on the context's GPU (the "measured HBM roofline" denominator): kind 0 =
 * read with 16-byte loads, 1 = device-to-device copy with 16-byte accesses, 2 = read with 4-byte
 * loads, 3 = copy with 4-byte accesses, 4..7 = copy in the radix-scatter write pattern (three arrays,
 * tiles of 3072 / 6144 / 12288 / 49152 dwords written as 16 runs each), 10..12 = copies with four 16-byte loads in
 * flight per lane (10 plain, 11 non-temporal stores, 12 non-temporal loads and stores).  `bytes` per buffer; `blocks`
 * workgroups of 256 threads (0 = 2048).  Returns the mean over `iters` launches of bytes moved (read + written) per second. */
int gs_membench(gs_ctx* ctx, int kind, size_t bytes, uint32_t blocks, uint32_t iters, float* gbytes_per_s,
                float* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_H */
