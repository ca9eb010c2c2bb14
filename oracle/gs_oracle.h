/*
 * gs_oracle.h -- CPU restatement of the splat hot path of SiTronXD/vk3dGaussianSplatting.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported CPU baseline.  The product (libgsplat_hip.so) never links it.
 *
 * PARITY PINNING: the reference has no tests, golden vectors or CPU path, and its arithmetic
 * is GLSL compute (needs glslang + a Vulkan device; neither exists here), so the shader
 * restatement is "parity unpinned by the reference".  What IS pinned by reference code run
 * here (oracle/ref_fixtures.cpp, built against the reference's vendored glm + SMath.h into
 * oracle/_ref/, outputs committed under tests/golden/): camera view/projection matrices
 * (Camera.cpp:7-48) and Morton codes (SMath.h:10-34).  Everything else is pinned by the
 * reference's own synthetic scenes (TestSortScene.cpp:16-33 known-answer depth keys),
 * its host formulas (Renderer.cpp:725, RadixSort.cpp:203-204) and structural invariants.
 *
 * All paths below are relative to /root/reference/vkGaussianSplatting/; S/ abbreviates
 * Resources/Shaders/.
 *
 * Numeric contract (what "restatement" means where GLSL leaves freedom):
 *   - IEEE-754 binary32, round-to-nearest-even, NO fused contraction (-ffp-contract=off),
 *     operands combined left-to-right exactly as the GLSL source text associates them;
 *   - '/' is IEEE division, sqrt is correctly rounded, normalize(v) = v / sqrt(dot(v,v));
 *   - matrix products are evaluated as sum over k ascending of A[k][row]*B[col][k];
 *   - tan(FOV_Y*0.5f) is evaluated once on the host in double and rounded to float;
 *   - float->int / float->uint conversions truncate toward zero and SATURATE (GLSL leaves
 *     out-of-range undefined; NVIDIA and AMD hardware saturate), NaN -> 0;
 *   - exp(x) is pinned to gso_exp() below: exp2(x*log2(e)) with a degree-6 polynomial,
 *     max error 1.4 ulp -- inside GLSL's (3 + 2|x|) ULP allowance for exp() and
 *     reproducible bit-for-bit on any IEEE machine with fmaf (libm's expf is not);
 *   - rgba8 UNORM store = floor(clamp(c,0,1)*255 + 0.5).
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSO_FLOATS_PER_GAUSSIAN 84u /* 21 x vec4 = 336 B, Engine/Graphics/ShaderStructs.h:59-70 */

/* Per-frame + per-scene parameters: CamUBO (ShaderStructs.h:37-41), InitSortListPCD (7-12),
 * constants of S/Common/Common.glsl:2-15 and Engine/Graphics/Camera.cpp:4-5. */
typedef struct gso_params {
    float view[16];      /* column-major, m[col*4+row] */
    float proj[16];      /* column-major */
    float cam_pos[3];
    uint32_t sh_mode;    /* 0 all bands, 1 skip first, 2 only first (Camera.h:7-12) */
    uint32_t width, height;
    float near_plane;    /* 0.1f */
    float far_plane;     /* 100.0f */
    uint32_t tile_size;  /* 16 */
    float ndc_cull;      /* 1.3f  CULLING_NDC_LIMIT */
    float in_view_limit; /* 0.8f  IN_VIEW_LIMIT */
    float fov_y;         /* 3.1415f*0.5f  FOV_Y (NOT pi/2) */
    uint32_t row_begin;  /* tile-row band [row_begin,row_end) this rank emits; 0..gridH for 1 GPU */
    uint32_t row_end;
} gso_params;

/* Per-splat intermediate results of InitSortList (for stage-level parity checks). */
typedef struct gso_splat {
    uint32_t visible;            /* survived both culls (InitSortList.comp:92-101) */
    uint32_t depth_key;          /* getDepthKey (70-80) */
    uint32_t min_x, min_y, max_x, max_y; /* getGaussianTileExtents (47-68), max exclusive */
    float screen_x, screen_y;    /* getScreenSpacePosition (Common.glsl:80-89) */
} gso_splat;

void gso_default_params(gso_params* p, uint32_t width, uint32_t height);

/* Host formulas. */
uint32_t gso_num_tiles_x(uint32_t width, uint32_t tile);
uint32_t gso_num_tiles_y(uint32_t height, uint32_t tile);
uint32_t gso_ceil_pow2(uint32_t x);                            /* Renderer.cpp:703-710 */
uint32_t gso_capacity(uint32_t n, uint32_t num_tiles);         /* Renderer.cpp:725 */
uint32_t gso_num_sort_bits(uint32_t num_tiles);                /* RadixSort.cpp:203-204 */
float    gso_tan_half_fov(float fov_y);                        /* Common.glsl:53, host-folded */
float    gso_exp(float x);                                     /* pinned exp, see header */
/* number of floats x in [lo, hi] (every stride-th bit pattern; lo <= hi <= 0; plus +0, NaNs, -inf) for which the cheaper evaluation the HIP blend loop
 * uses differs from gso_exp(x) in any bit; *first_bad = the first such x */
uint64_t gso_exp_live_mismatches(float lo, float hi, uint32_t stride, float* first_bad);

/* Stage 1: InitSortList.comp:82-151 in the canonical (ascending splat index) order.
 * color/cov are [n][4]; entries of culled splats are left untouched (N6).  lists have
 * `capacity` entries and are filled with 0xFFFFFFFF first (Subrenderer.cpp:42-46).
 * Returns the un-truncated atomic counter value (may exceed capacity). */
uint64_t gso_init_sort_list(const gso_params* p, const float* aos, uint32_t n, uint32_t capacity,
                            float* color, float* cov, gso_splat* splats /* may be NULL */,
                            uint32_t* list_tile, uint32_t* list_depth, uint32_t* list_id);

/* Stage 2a: the sort's semantics -- stable sort by (tile<<32|depth) of the first e entries. */
void gso_sort_stable(uint32_t* tile, uint32_t* depth, uint32_t* id, uint32_t e);

/* Stage 2b: literal model of the six radix shaders (S/ComputeShaders/RadixSort/, all .comp files) with WORK_GROUP_SIZE=64,
 * pass loop and ping-pong as RadixSort.cpp:306-651.  `cap` entries per array; scratch arrays
 * (ping-pong) are allocated inside.  counter = atomic counter from stage 1.  Result is left in
 * tile/depth/id (like the caller-visible gaussiansSortListSBO after the swaps). */
void gso_radix_sort_literal(uint32_t* tile, uint32_t* depth, uint32_t* id, uint32_t cap,
                            uint64_t counter, uint32_t num_sort_bits);

/* Stage 3: FindRanges.comp:42-71.  ranges is [num_tiles][2], zeroed first
 * (Subrenderer.cpp:55-60).  literal != 0: loop over `n_threads` = capacity with the 0xFFFFFFFF
 * sentinel exactly like the shader (incl. quirk Q1 at i == capacity-1).  literal == 0: the
 * product's form -- loop over e = min(counter, capacity) valid entries, last end = e. */
void gso_find_ranges(const uint32_t* tile, uint32_t n_threads_or_e, uint32_t num_tiles,
                     uint32_t* ranges, int literal);

/* Stage 4: RenderGaussians.comp:56-152.  pos is read from the aos records; color/cov are the
 * stage-1 outputs.  rgba_out is height*width*4 bytes, row-major, top row first, A = 255.
 * Only tile rows [row_begin,row_end) of p are rendered (other pixels untouched). */
void gso_render(const gso_params* p, const float* aos, const float* color, const float* cov,
                const uint32_t* sorted_id, const uint32_t* ranges, uint8_t* rgba_out);

/* Stage 4 with libm expf() instead of gso_exp(): used by tests to show the pinned exp does not
 * move any 8-bit channel by more than 1 step. */
void gso_render_libm_exp(const gso_params* p, const float* aos, const float* color,
                         const float* cov, const uint32_t* sorted_id, const uint32_t* ranges,
                         uint8_t* rgba_out);

/* Whole frame (what bench.py's cpu_baseline times).  timings_ms[5] = init, sort, ranges,
 * render, total (same buckets as Renderer.cpp:471-475).  Returns min(counter, capacity). */
uint32_t gso_frame(const gso_params* p, const float* aos, uint32_t n, uint8_t* rgba_out,
                   double* timings_ms);

/* The same frame on `threads` host threads (1..GSO_MAX_THREADS): splats split across threads for
 * InitSortList (count, scan, emit in ascending splat index), a parallel stable LSD radix for the
 * sort, tile rows for RenderGaussians.  Same image and same E as gso_frame(). */
#define GSO_MAX_THREADS 256
uint32_t gso_frame_mt(const gso_params* p, const float* aos, uint32_t n, uint8_t* rgba_out,
                      double* timings_ms, uint32_t threads);

/* The stages of gso_frame_mt one by one, for full-size parity tests (same outputs as the
 * single-thread functions above; tests/test_oracle.py checks that). */
uint64_t gso_init_sort_list_mt(const gso_params* p, const float* aos, uint32_t n, uint32_t capacity,
                               float* color, float* cov, gso_splat* splats /* may be NULL */,
                               uint32_t* list_tile, uint32_t* list_depth, uint32_t* list_id,
                               uint32_t threads);
void gso_sort_stable_mt(uint32_t* tile, uint32_t* depth, uint32_t* id, uint32_t e, uint32_t threads);
void gso_render_mt(const gso_params* p, const float* aos, const float* color, const float* cov,
                   const uint32_t* sorted_id, const uint32_t* ranges, uint8_t* rgba_out,
                   uint32_t threads);

/* Camera (Camera.cpp:7-48 over glm 0.9.9.8 lookAtRH / perspectiveRH_ZO). */
void gso_camera_matrices(const float pos[3], float yaw, float pitch, float aspect,
                         float near_plane, float far_plane, float* view16, float* proj16);

/* Loader helpers (ResourceManager.cpp:229-297). */
uint32_t gso_morton(uint32_t x, uint32_t y, uint32_t z);       /* SMath.h:10-34 */

#ifdef __cplusplus
}
#endif
#endif
