// gs_render.hip -- FindRanges and RenderGaussians for gfx950.
//
// k_find_ranges: FindRanges.comp:42-71, launched over the E valid elements instead of the list
//   capacity (the reference runs 2^24 threads in 16-wide groups over mostly 0xFFFFFFFF sentinels,
//   Subrenderer.cpp:205-215); last end = E (differs from the reference only when the list
//   overflowed, quirk Q1 in SURVEY.md).
// k_render / k_render_wg: RenderGaussians.comp:56-152 in two launch shapes (gs_config.render_kernel; table of
//   measurements above launch_render).  k_render<PX>: independent wave64s, PX horizontally adjacent pixels per
//   lane (PX = 4: one wave per 16x16 tile), no workgroup barriers, one 16-byte RGBA8 store per lane.
//   k_render_wg: one 256-thread workgroup per tile, one pixel per lane (the shader's own shape), its four waves
//   each walking the list on their own for their four pixel rows.  Both gather the batch through the sorted id list
//   from the 48-byte SplatRaster records (screen position, inverse covariance, colour: set up once per splat by
//   k_project), prefetch one batch ahead of the blend loop, drop while staging the splats that provably touch no
//   pixel of the rectangle (conservative, unobservable), and use wave votes for the early-outs the reference lacks
//   (its `done` only zeroes `limit`, :111).  k_tile_classes + k_tile_scatter: the order in which the tiles are dispatched.
// GS_RENDER_EXACT evaluates every expression in the reference's order without contraction and with
// the pinned exp of oracle/gs_oracle.h => pixels bit-identical to the CPU oracle.
// GS_RENDER_FAST uses fused multiply-adds and the hardware exp2 (what a GLSL compiler is free to
// emit for the same source); <= 1 step per 8-bit channel against the oracle.
#include "gs_device_utils.h"
#include "gs_internal.h"

namespace gs {

// The sort list holds compact tile ids (FrameParams); ranges[] is indexed by the GLOBAL tile id.
struct TileMap { uint32_t grid_w, first_row, row_stride; };
__device__ __forceinline__ uint32_t global_tile(const TileMap& m, uint32_t c) {
    if (m.row_stride == 1u) return c + m.first_row * m.grid_w;
    const uint32_t k = c / m.grid_w;
    return (m.first_row + k * m.row_stride) * m.grid_w + (c - k * m.grid_w);
}

__global__ __launch_bounds__(256) void k_find_ranges(const uint32_t* __restrict__ tile,
                                                      const SortParams* __restrict__ params,
                                                      uint32_t* __restrict__ ranges, uint32_t hi16, TileMap map) {
    const uint32_t e = params->num_elems;
    // hi16: 16-bit tile ids (see k_scatter) -- eight elements per 16-byte load, else four
    const uint16_t* tile16 = reinterpret_cast<const uint16_t*>(tile);
    const uint32_t per = hi16 ? 8u : 4u;
    const uint32_t chunks = (e + per - 1u) / per;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < chunks; q += gridDim.x * blockDim.x) {
        const uint32_t i0 = q * per;
        uint32_t t[8];
        if (i0 + per - 1u < e) {
            if (hi16) {
                const uint4 v = *reinterpret_cast<const uint4*>(tile16 + i0);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(tile + i0);
                t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                t[k] = (uint32_t)k < per && i0 + k < e ? (hi16 ? (uint32_t)tile16[i0 + k] : tile[i0 + k]) : 0u;
        }
        uint32_t prev = i0 > 0 ? (hi16 ? (uint32_t)tile16[i0 - 1] : tile[i0 - 1]) : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t i = i0 + k;
            if ((uint32_t)k < per && i < e) {
                if (i == 0) {
                    ranges[global_tile(map, t[k]) * 2 + 0] = 0;               // FindRanges.comp:59-64
                } else if (prev != t[k]) {                                    // :48-58
                    ranges[global_tile(map, prev) * 2 + 1] = i;
                    ranges[global_tile(map, t[k]) * 2 + 0] = i;
                }
                if (i == e - 1) ranges[global_tile(map, t[k]) * 2 + 1] = e;   // end of the last tile = E
                prev = t[k];
            }
        }
    }
}

// pinned exp: identical operation sequence to gso_exp() in oracle/gs_oracle.c
__device__ __forceinline__ float exp_pinned(float x) {
    float t = x * 0x1.715476p+0f;
    t = t > -126.0f ? t : -126.0f;
    t = t < 126.0f ? t : 126.0f;
    const float n = __builtin_rintf(t);
    const float r = t - n;
    float p = 0x1.42059ap-13f;
    p = __builtin_fmaf(p, r, 0x1.5f3e12p-10f);
    p = __builtin_fmaf(p, r, 0x1.3b2d40p-7f);
    p = __builtin_fmaf(p, r, 0x1.c6aeeap-5f);
    p = __builtin_fmaf(p, r, 0x1.ebfbdcp-3f);
    p = __builtin_fmaf(p, r, 0x1.62e430p-1f);
    p = __builtin_fmaf(p, r, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

// The same pinned exp on two values at once: v_pk_mul / v_pk_add / v_pk_fma are IEEE per component, so
// each component goes through exactly the operation sequence of exp_pinned().
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f exp_pinned2(v2f x) {
    v2f t = x * (v2f){0x1.715476p+0f, 0x1.715476p+0f};
    t.x = t.x > -126.0f ? t.x : -126.0f;
    t.y = t.y > -126.0f ? t.y : -126.0f;
    t.x = t.x < 126.0f ? t.x : 126.0f;
    t.y = t.y < 126.0f ? t.y : 126.0f;
    const v2f n = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
    const v2f r = t - n;
    v2f p = {0x1.42059ap-13f, 0x1.42059ap-13f};
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.5f3e12p-10f, 0x1.5f3e12p-10f});
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.3b2d40p-7f, 0x1.3b2d40p-7f});
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.c6aeeap-5f, 0x1.c6aeeap-5f});
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.ebfbdcp-3f, 0x1.ebfbdcp-3f});
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.62e430p-1f, 0x1.62e430p-1f});
    p = __builtin_elementwise_fma(p, r, (v2f){1.0f, 1.0f});
    return (v2f){__builtin_ldexpf(p.x, (int)n.x), __builtin_ldexpf(p.y, (int)n.y)};
}

// exp_pinned2 for the blend loop, same bits with 14 instead of 22 instructions per pair.  What differs is only HOW the
// same values are produced:
//   * the two clamps of gso_exp (t > -126 ? t : -126; t < 126 ? t : 126) are one v_max_f32 per component: the upper
//     clamp cannot trigger where the result is used (a live lane has f <= 0, so t <= 0), and IEEE maxnum(NaN, -126) is
//     -126, which is what `NaN > -126 ? NaN : -126` gives;
//   * n = rint(t) as (t + 1.5 * 2^23) - 1.5 * 2^23 (round to nearest even, exact for |t| <= 126);
//   * ldexp(p, n) as an integer add of n to p's exponent field -- the low bits of t + 1.5 * 2^23 ARE n, and
//     (bits << 23) keeps exactly n << 23 (mod 2^32): p lies in [0.70, 1.42] and n >= -126, where n = -126 only comes
//     with r >= 0, i.e. p >= 1, so the exponent field never reaches 0 and the product is exact like ldexp's.
// Lanes whose exponent failed the f tests compute garbage here; blend_entry never looks at it.
__device__ __forceinline__ v2f exp_pinned2_live(v2f x) {
    v2f t = x * (v2f){0x1.715476p+0f, 0x1.715476p+0f};
    t.x = __builtin_fmaxf(t.x, -126.0f);
    t.y = __builtin_fmaxf(t.y, -126.0f);
    const v2f magic = {12582912.0f, 12582912.0f};
    const v2f tn = t + magic;
    const v2f n = tn - magic;
    const v2f r = t - n;
    v2f p = {0x1.42059ap-13f, 0x1.42059ap-13f};
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.5f3e12p-10f, 0x1.5f3e12p-10f});
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.3b2d40p-7f, 0x1.3b2d40p-7f});
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.c6aeeap-5f, 0x1.c6aeeap-5f});
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.ebfbdcp-3f, 0x1.ebfbdcp-3f});
    p = __builtin_elementwise_fma(p, r, (v2f){0x1.62e430p-1f, 0x1.62e430p-1f});
    p = __builtin_elementwise_fma(p, r, (v2f){1.0f, 1.0f});
    return (v2f){__uint_as_float(__float_as_uint(p.x) + (__float_as_uint(tn.x) << 23)),
                 __uint_as_float(__float_as_uint(p.y) + (__float_as_uint(tn.y) << 23))};
}

__device__ __forceinline__ float exp_pinned_live(float x) {            // the scalar form of exp_pinned2_live
    const float t = __builtin_fmaxf(x * 0x1.715476p+0f, -126.0f);
    const float tn = t + 12582912.0f;
    const float r = t - (tn - 12582912.0f);
    float p = 0x1.42059ap-13f;
    p = __builtin_fmaf(p, r, 0x1.5f3e12p-10f);
    p = __builtin_fmaf(p, r, 0x1.3b2d40p-7f);
    p = __builtin_fmaf(p, r, 0x1.c6aeeap-5f);
    p = __builtin_fmaf(p, r, 0x1.ebfbdcp-3f);
    p = __builtin_fmaf(p, r, 0x1.62e430p-1f);
    p = __builtin_fmaf(p, r, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(tn) << 23));
}
#ifdef GS_RENDER_EXP_CLAMPS
#define GS_EXP1 exp_pinned
#define GS_EXP2 exp_pinned2
#else
#define GS_EXP1 exp_pinned_live
#define GS_EXP2 exp_pinned2_live
#endif

struct Fetched {
    float4 a, b, c;   // raw SplatRaster
    bool valid;
};

__device__ __forceinline__ Fetched fetch_splat(const SplatRaster* __restrict__ raster,
                                               const uint32_t* __restrict__ sorted_id,
                                               uint32_t idx, uint32_t end) {
    Fetched f;
    f.valid = idx < end;
    if (f.valid) {
        const uint32_t gi = sorted_id[idx];                            // RenderGaussians.comp:88
        const float4* rp = reinterpret_cast<const float4*>(raster + gi);
        f.a = rp[0];
        f.b = rp[1];
        f.c = rp[2];
    } else {
        f.a = f.b = f.c = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return f;
}

// True when no pixel of the rectangle [x0, x0 + cols_m1] x [y0, y0 + rows_m1] can reach the exponent fthr for the
// conic (ix, iy, iz) centred at (sx, sy).  See stage_splat.
__device__ __forceinline__ bool rect_unreachable(float sx, float sy, float ix, float iy, float iz, float fthr,
                                                 float x0, float y0, float rows_m1, float cols_m1 = 15.0f) {
    const float u0 = sx - (x0 + cols_m1), u1 = sx - x0;                             // u range (u0 <= u1)
    const float v0 = y0 - sy, v1 = y0 + rows_m1 - sy;                               // v range
    bool reject = false;
    if (ix > 0.0f && iz > 0.0f && !(u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f)) {
        // hardware reciprocals (1 ulp): q is stationary at the minimiser, so its error enters q squared -- far below tol
        const float r_iz = __builtin_amdgcn_rcpf(iz), r_ix = __builtin_amdgcn_rcpf(ix);
        auto edge_u = [&](float ue) {   // min over v in [v0,v1] of q(ue, v)
            const float vs = fminf(fmaxf(-iy * ue * r_iz, v0), v1);
            return ix * ue * ue + 2.0f * iy * ue * vs + iz * vs * vs;
        };
        auto edge_v = [&](float ve) {   // min over u in [u0,u1] of q(u, ve)
            const float us = fminf(fmaxf(-iy * ve * r_ix, u0), u1);
            return ix * us * us + 2.0f * iy * us * ve + iz * ve * ve;
        };
        const float qmin = fminf(fminf(edge_u(u0), edge_u(u1)), fminf(edge_v(v0), edge_v(v1)));
        const float far_x = fmaxf(fabsf(u0), fabsf(u1)), far_y = fmaxf(fabsf(v0), fabsf(v1));
        const float mag = fabsf(ix) * far_x * far_x + fabsf(iz) * far_y * far_y + 2.0f * fabsf(iy) * far_x * far_y;
        const float tol = 0.01f + 8e-6f * mag;      // >> rounding of qmin here and of f in the pixel loop
        reject = (-0.5f * qmin + tol < fthr);
    }
    return reject;
}

// What is left of RenderGaussians.comp:86-108 per list entry now that k_project has done the per-splat setup (screen
// position, inverse 2x2 covariance, zero-determinant rule: SplatRaster): two things the shader does not have -- the
// exponent below which alpha < 1/255 is certain, and an exact rejection of splats that cannot touch the pixel
// rectangle [tile_x0, tile_x0 + 15] x [tile_y0, tile_y0 + rows_m1].  Returns false when the splat can be dropped;
// fthr goes to the slot beside alpha (nxt.c.y).
__device__ __forceinline__ bool stage_splat(Fetched& nxt, float tile_x0, float tile_y0, float rows_m1, float cols_m1 = 15.0f) {
    bool keep = false;
    if (nxt.valid) {
        const float sx = nxt.a.x, sy = nxt.a.y;
        const float ix = nxt.a.z, iy = nxt.a.w, iz = nxt.b.x;
        const float alpha0 = nxt.c.x;
        // Skip threshold: alpha = a*exp(f) < 1/255 (the `continue` of :127) is certain once
        // f < ln(1/(255 a)) - margin; the margin (0.01) dwarfs the errors of the fast log and of
        // the pinned exp (<= 2e-6 relative), so the test below never changes a result.  a <= 0 or
        // NaN gives +inf / NaN, i.e. "always skip" / "never skip", both exact.
        const float fthr = -__logf(255.0f * alpha0) - 0.01f;
        nxt.c.y = fthr;
        // Whole-rectangle rejection (the reference assigns tiles by a 3-sigma bounding box, so many list
        // entries touch no pixel of the tile).  With u = sx - px, v = py - sy the shader's exponent is
        // f = -q/2, q(u,v) = ix u^2 + 2 iy u v + iz v^2.  Over the pixel rectangle q is minimal either at the
        // centre (inside: keep) or on one of the four edges -- for a definite form and for an indefinite one
        // alike (no interior minimum then) -- where it is a 1-D parabola, convex because ix, iz > 0, with a
        // closed-form clamped minimiser.  If -q_min/2, widened by a generous bound on the fp32 error of the
        // per-pixel f, is below the skip threshold, every pixel would `continue` (:127): dropping the splat is
        // unobservable.
        keep = !rect_unreachable(sx, sy, ix, iy, iz, fthr, tile_x0, tile_y0, rows_m1, cols_m1);
    }
    return keep;
}

// STATS is a tuning-only instantiation (gs_debug_render_stats): per tile {list length, splats
// visited, splats with any pixel needing exp, clock ticks | entries staged before the tile was done, -, -, -}.
// The product launches STATS = false.
// PX = pixels per lane: 4 -> one wave per tile (16 rows x 4 lanes), 2 -> two waves per tile (each
// 8 rows x 8 lanes), 1 -> four waves per tile (each 4 rows x 16 lanes).  Waves of one tile are
// independent workgroups: each gathers, culls (against its own pixel rectangle) and blends on its own.
template <bool EXACT, int PX, bool STATS = false>
__global__ __launch_bounds__(64) void k_render(const FrameParams fp,
                                                const SplatRaster* __restrict__ raster,
                                                const uint32_t* __restrict__ sorted_id,
                                                const uint32_t* __restrict__ ranges,
                                                const uint32_t* __restrict__ order,
                                                uint32_t* __restrict__ rgba, uint4* __restrict__ stats = nullptr) {
    // LDS image of the current batch: {sx, sy, inv.x, inv.y}, {inv.z, r, g, b}, {a, skip threshold, -, -}
    __shared__ float4 s_batch[64][3];

    constexpr int WPT = 4 / PX;            // waves per tile
    constexpr int ROWS = kTile / WPT;      // pixel rows per wave
    constexpr int LPR = 64 / ROWS;         // lanes per row
    static_assert(LPR * PX == kTile, "lane layout");
    const int lane = threadIdx.x;
    const uint32_t tile_in_band = order ? order[blockIdx.x / WPT] : blockIdx.x / WPT;   // longest lists first (k_tile_classes, k_tile_scatter)
    const uint32_t sub = blockIdx.x % WPT;
    const uint32_t krow = tile_in_band / fp.grid_w;                    // index among this context's tile rows
    const uint32_t ty = fp.first_row + krow * fp.row_stride;
    const uint32_t tx = tile_in_band % fp.grid_w;
    const uint32_t tile_index = ty * fp.grid_w + tx;                   // :74-76
    const uint32_t start = ranges[tile_index * 2 + 0];                 // :77
    const uint32_t end = ranges[tile_index * 2 + 1];

    const uint32_t py = ty * kTile + sub * ROWS + (uint32_t)(lane / LPR);
    const uint32_t px0 = tx * kTile + (uint32_t)(lane % LPR) * PX;
    const float fpy = (float)py;                                       // integer pixel coords (R1)
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile + sub * ROWS);
    float fpx[PX];
    float col[PX][3];
    float T[PX];
    bool done[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        fpx[k] = (float)(px0 + k);
        col[k][0] = col[k][1] = col[k][2] = 0.0f;
        T[k] = 1.0f;
        done[k] = !(px0 + k < fp.width && py < fp.height);             // never stored (:147)
    }

    uint32_t st_visited = 0, st_need = 0, st_walked = 0;   // STATS: entries visited / needing an exp / staged before the tile was done
    const uint64_t st_t0 = STATS ? __builtin_amdgcn_s_memtime() : 0;
    Fetched nxt = fetch_splat(raster, sorted_id, start + lane, end);
    for (uint32_t i = start; i < end; i += 64) {                       // :81
        // :86-108, one list entry per lane (the per-splat setup itself is k_project's)
        const bool keep = stage_splat(nxt, tile_x0, tile_y0, (float)(ROWS - 1));
        const uint64_t kmask = __ballot(keep);
        const uint32_t n = (uint32_t)__popcll(kmask);
        if (STATS) st_walked = (i + 64u < end ? i + 64u : end) - start;
        if (keep) {
            const uint32_t slot = mbcnt(kmask);                        // order-preserving compaction
            s_batch[slot][0] = nxt.a;
            s_batch[slot][1] = nxt.b;
            s_batch[slot][2] = nxt.c;
        }
        __syncthreads();                                               // :109 (single wave)
        nxt = fetch_splat(raster, sorted_id, i + 64 + lane, end);      // prefetch next batch

        for (uint32_t j = 0; j < n; ++j) {                             // :112
            const float4 g0 = s_batch[j][0];
            const float4 g1 = s_batch[j][1];
            const float2 g2 = *reinterpret_cast<const float2*>(&s_batch[j][2]);
            const float ga = g2.x, fthr = g2.y;
            float ey = g0.y - fpy;                                     // :119
            ey = -ey;                                                  // :120
            float f[PX];
            bool need[PX];
            bool any_need = false;
            if constexpr (EXACT && PX == 4) {
                // two pixels per instruction (v_pk_mul_f32 / v_pk_add_f32 are IEEE per component, so the
                // operand order and every rounding are those of the scalar expression of :123)
                const float c_term = g1.x * ey * ey;
                const v2f sxv = {g0.x, g0.x}, ixv = {g0.z, g0.z}, iyv = {g0.w, g0.w};
                const v2f eyv = {ey, ey}, cv = {c_term, c_term}, mh = {-0.5f, -0.5f};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const v2f pxv = {fpx[2 * q], fpx[2 * q + 1]};
                    const v2f ex = sxv - pxv;
                    const v2f ff = mh * (ixv * ex * ex + cv) - iyv * ex * eyv;
                    f[2 * q] = ff.x;
                    f[2 * q + 1] = ff.y;
                }
            } else if constexpr (EXACT) {
                const float c_term = g1.x * ey * ey;                   // gCovInv.z * y * y
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    const float ex = g0.x - fpx[k];
                    f[k] = -0.5f * (g0.z * ex * ex + c_term) - g0.w * ex * ey;  // :123
                }
            } else {
                const float c_term = g1.x * ey * ey;
                const float b_term = g0.w * ey;
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    const float ex = g0.x - fpx[k];
                    const float q = __builtin_fmaf(g0.z * ex, ex, c_term);
                    f[k] = __builtin_fmaf(-0.5f, q, -(b_term * ex));
                }
            }
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                need[k] = !done[k] && !(f[k] > 0.0f) && !(f[k] < fthr);
                any_need |= need[k];
            }
            if (STATS) ++st_visited;
            if (!__any(any_need)) continue;                            // nobody can pass :127
            if (STATS) ++st_need;
            if constexpr (EXACT && PX == 4) {
                // blend two pixels per instruction where the ISA has a packed form
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k0 = 2 * q, k1 = 2 * q + 1;
                    if (!__any(need[k0] || need[k1])) continue;
                    const v2f alpha = (v2f){ga, ga} * GS_EXP2((v2f){f[k0], f[k1]});          // :124
                    const bool act0 = need[k0] && !(alpha.x < 1.0f / 255.0f);                // :127
                    const bool act1 = need[k1] && !(alpha.y < 1.0f / 255.0f);
                    const v2f Tv = {T[k0], T[k1]};
                    const v2f wgt = Tv * alpha;                                              // :131
                    const v2f cr = (v2f){col[k0][0], col[k1][0]} + wgt * (v2f){g1.y, g1.y};
                    const v2f cg = (v2f){col[k0][1], col[k1][1]} + wgt * (v2f){g1.z, g1.z};
                    const v2f cb = (v2f){col[k0][2], col[k1][2]} + wgt * (v2f){g1.w, g1.w};
                    col[k0][0] = act0 ? cr.x : col[k0][0];
                    col[k1][0] = act1 ? cr.y : col[k1][0];
                    col[k0][1] = act0 ? cg.x : col[k0][1];
                    col[k1][1] = act1 ? cg.y : col[k1][1];
                    col[k0][2] = act0 ? cb.x : col[k0][2];
                    col[k1][2] = act1 ? cb.y : col[k1][2];
                    const v2f next_t = Tv * ((v2f){1.0f, 1.0f} - alpha);                     // :133
                    const bool fin0 = act0 && next_t.x < 0.0001f;                            // :136-140
                    const bool fin1 = act1 && next_t.y < 0.0001f;
                    done[k0] = done[k0] || fin0;
                    done[k1] = done[k1] || fin1;
                    T[k0] = (act0 && !fin0) ? next_t.x : T[k0];                              // :142
                    T[k1] = (act1 && !fin1) ? next_t.y : T[k1];
                }
            } else
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                if (PX > 1 && !__any(need[k])) continue;
                float alpha;
                if constexpr (EXACT) alpha = ga * GS_EXP1(f[k]);       // :124
                else alpha = ga * __builtin_amdgcn_exp2f(f[k] * 0x1.715476p+0f);
                const bool act = need[k] && !(alpha < 1.0f / 255.0f);  // :127
                const float wgt = T[k] * alpha;                        // :131
                if constexpr (EXACT) {
                    col[k][0] = act ? col[k][0] + wgt * g1.y : col[k][0];
                    col[k][1] = act ? col[k][1] + wgt * g1.z : col[k][1];
                    col[k][2] = act ? col[k][2] + wgt * g1.w : col[k][2];
                } else {
                    const float w0 = act ? wgt : 0.0f;
                    col[k][0] = __builtin_fmaf(w0, g1.y, col[k][0]);
                    col[k][1] = __builtin_fmaf(w0, g1.z, col[k][1]);
                    col[k][2] = __builtin_fmaf(w0, g1.w, col[k][2]);
                }
                const float next_t = T[k] * (1.0f - alpha);            // :133
                const bool fin = act && next_t < 0.0001f;              // :136-140, colour already added
                done[k] = done[k] || fin;
                T[k] = (act && !fin) ? next_t : T[k];                  // :142
            }
            bool all_done = true;
#pragma unroll
            for (int k = 0; k < PX; ++k) all_done = all_done && done[k];
            if (__all(all_done)) goto finish;                          // whole-(sub)tile early-out
        }
        __syncthreads();                                               // :84
    }
finish:
    if (STATS && lane == 0) {
        stats[tile_index * 2] = make_uint4(end - start, st_visited, st_need, (uint32_t)(__builtin_amdgcn_s_memtime() - st_t0));
        stats[tile_index * 2 + 1] = make_uint4(st_walked, 0u, 0u, 0u);
    }
    // :147-151 clamp + RGBA8 UNORM store, A = 255
    uint32_t packed[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        uint32_t v = 0xFF000000u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = clampf(col[k][c], 0.0f, 1.0f);
            v |= (uint32_t)(x * 255.0f + 0.5f) << (8 * c);
        }
        packed[k] = v;
    }
    if (py < fp.height) {
        const uint32_t out_y = fp.compact_out ? krow * kTile + sub * ROWS + (uint32_t)(lane / LPR) : py;
        uint32_t* row = rgba + (size_t)out_y * fp.width;
        const bool vec_ok = (fp.width % PX) == 0u && px0 + (PX - 1) < fp.width;
        if (PX == 4 && vec_ok) {
            *reinterpret_cast<uint4*>(row + px0) = make_uint4(packed[0], packed[PX > 1 ? 1 : 0], packed[PX > 2 ? 2 : 0], packed[PX > 3 ? 3 : 0]);
        } else if (PX == 2 && vec_ok) {
            *reinterpret_cast<uint2*>(row + px0) = make_uint2(packed[0], packed[PX > 1 ? 1 : 0]);
        } else {
#pragma unroll
            for (int k = 0; k < PX; ++k)
                if (px0 + k < fp.width) row[px0 + k] = packed[k];
        }
    }
}

// Lane predicates of the workgroup kernel live as 64-bit wave masks in scalar registers: a comparison already leaves its
// result there, combining masks is scalar work that issues beside the vector instructions, and a select reads the mask
// directly (inverse ballot: no instruction).  Kept as per-lane bools the compiler parked `done` in a vector register
// and rebuilt masks from it every step (six vector instructions of ~90 per pair of entries).
__device__ __forceinline__ uint64_t mask_of(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ float sel(uint64_t m, float a, float b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b; }

// RenderGaussians.comp:127-142 for one pixel per lane and one list entry whose alpha is known: the `continue` on
// alpha < 1/255, the colour add, add-then-test transmittance.  `need` = lanes that are live and whose exponent passed the
// f tests.  FINITE (wave-uniform: every colour of the staged batch is finite): the colour add runs unconditionally
// with the weight forced to zero on the lanes that skip -- col + 0 * c == col bit for bit when c is finite (colours are
// max(x, 0), never negative) -- three selects less per entry; a batch with a non-finite colour takes the select form,
// where a skipped lane's colour is not touched at all, as in the shader.
template <bool EXACT, bool FINITE>
__device__ __forceinline__ void blend_entry(uint64_t need, float alpha, float cr, float cg, float cb, float& col0, float& col1,
                                            float& col2, float& T, uint64_t& done) {
    const uint64_t act = need & ~mask_of(alpha < 1.0f / 255.0f);      // :127
    const float wgt = T * alpha;                                       // :131
    if constexpr (!EXACT) {
        const float w0 = sel(act, wgt, 0.0f);
        col0 = __builtin_fmaf(w0, cr, col0);
        col1 = __builtin_fmaf(w0, cg, col1);
        col2 = __builtin_fmaf(w0, cb, col2);
    } else if constexpr (FINITE) {
        const float w0 = sel(act, wgt, 0.0f);
        col0 = col0 + w0 * cr;
        col1 = col1 + w0 * cg;
        col2 = col2 + w0 * cb;
    } else {
        col0 = sel(act, col0 + wgt * cr, col0);
        col1 = sel(act, col1 + wgt * cg, col1);
        col2 = sel(act, col2 + wgt * cb, col2);
    }
    const float next_t = T * (1.0f - alpha);                           // :133
    const uint64_t fin = act & mask_of(next_t < 0.0001f);              // :136-140, colour already added
    done |= fin;
    T = sel(act & ~fin, next_t, T);                                    // :142
}

// Two list entries (the pair `pair` of the staged batch; two == false when the batch ends on a single one: the second
// half of the pair's slots then holds stale values -- and a zero colour -- that nothing looks at) against one pixel per lane: their exponents,
// the pinned exp and alpha are evaluated side by side in the two halves of packed fp32 instructions (v_pk_mul /
// v_pk_add / v_pk_fma are IEEE per component, so each half goes through exactly the scalar operation sequence of
// RenderGaussians.comp:119-124), then the two blends run one after the other in list order (:127-142 is a chain
// through T).  `two` is wave-uniform.  The batch lies in LDS pair-interleaved -- {sx_a, sx_b, sy_a, sy_b}, {ix_a, ix_b,
// iy_a, iy_b}, {iz_a, iz_b, alpha_a, alpha_b}, {thr_a, thr_b, r_a, r_b}, {g_a, g_b, b_a, b_b} -- so that five 16-byte
// reads deliver every operand as the register pair a packed instruction wants (entry-major slots cost ten v_mov per
// step to build those pairs: 8 % of the loop).
constexpr int kPairQuads = 5;
template <bool EXACT, bool FINITE>
__device__ __forceinline__ void blend_pair(const float4* pairs, int pair, bool two, float fpx, float fpy,
                                           float& col0, float& col1, float& col2, float& T, uint64_t& done) {
    const float4* q = pairs + pair * kPairQuads;
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const v2f ex = (v2f){q0.x, q0.y} - (v2f){fpx, fpx};                          // :119
    const v2f ey = -((v2f){q0.z, q0.w} - (v2f){fpy, fpy});                       // :119-120
    const v2f ixv = {q1.x, q1.y}, iyv = {q1.z, q1.w}, izv = {q2.x, q2.y};
    v2f f;
    if constexpr (EXACT) {
        f = (v2f){-0.5f, -0.5f} * (ixv * ex * ex + izv * ey * ey) - iyv * ex * ey;   // :123
    } else {
        const v2f qq = __builtin_elementwise_fma(ixv * ex, ex, izv * ey * ey);
        f = __builtin_elementwise_fma((v2f){-0.5f, -0.5f}, qq, -(iyv * ey * ex));
    }
    const uint64_t live_a = mask_of(!(f.x > 0.0f)) & mask_of(!(f.x < q3.x));
    const uint64_t live_b = two ? mask_of(!(f.y > 0.0f)) & mask_of(!(f.y < q3.y)) : 0ull;
    if (((live_a | live_b) & ~done) == 0ull) return;                   // nobody in these rows can pass :127
    const float4 q4 = q[4];
    v2f alpha;
    if constexpr (EXACT) {
        alpha = (v2f){q2.z, q2.w} * GS_EXP2(f);                        // :124
    } else {
        const v2f t = f * (v2f){0x1.715476p+0f, 0x1.715476p+0f};
        alpha = (v2f){q2.z, q2.w} * (v2f){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    }
    blend_entry<EXACT, FINITE>(live_a & ~done, alpha.x, q3.z, q4.x, q4.z, col0, col1, col2, T, done);
    blend_entry<EXACT, FINITE>(live_b & ~done, alpha.y, q3.w, q4.y, q4.w, col0, col1, col2, T, done);
}

// One 256-thread workgroup per tile, one pixel per lane -- the reference's own launch shape (RenderGaussians.comp:
// local_size 16x16) -- but its four waves share nothing except the launch: wave w owns the pixel rows 4 w .. 4 w + 3 of
// the tile and walks the tile's list on its own in batches of 64 entries -- its own gather (prefetched one batch ahead),
// the exact rectangle test against its own 16 x 4 pixels, order-preserving compaction into its quarter of the LDS
// buffer, a dense blend loop over the survivors two entries per step (blend_pair) -- and leaves as soon as its 64
// pixels are done.  No workgroup barrier anywhere: a wave never waits for the wave with the most visits in a batch,
// which is what the shared 256-entry batch of the shader (and of this kernel until round 3) costs on long lists
// (profiles/r03_render_variants.txt: sharing the first 1, 2, 3, 4 or all batches loses 3 / 7 / 12 / 18 / 27 % on the
// capture-like cloud and 15 - 22 % on the uniform one, although every entry is then gathered four times).
// QUAD: wave w owns the 8 x 8 quadrant (w & 1, w >> 1) of the tile instead of the 16 x 4 strip w -- 20 % less perimeter,
// so fewer list entries survive a wave's rectangle test (gs_config.render_kernel = GS_RENDER_KERNEL_WORKGROUP_8X8).
template <bool EXACT, bool QUAD>
__global__ __launch_bounds__(256) void k_render_wg(const FrameParams fp,
                                                    const SplatRaster* __restrict__ raster,
                                                    const uint32_t* __restrict__ sorted_id,
                                                    const uint32_t* __restrict__ ranges,
                                                    const uint32_t* __restrict__ order,
                                                    uint32_t* __restrict__ rgba) {
    __shared__ float4 s_batch[4][32 * kPairQuads];   // per wave: 32 pairs of staged entries, pair-interleaved (blend_pair)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile_in_band = order ? order[blockIdx.x] : blockIdx.x;   // longest lists first (k_tile_classes, k_tile_scatter)
    const uint32_t krow = tile_in_band / fp.grid_w;                    // index among this context's tile rows
    const uint32_t ty = fp.first_row + krow * fp.row_stride;
    const uint32_t tx = tile_in_band % fp.grid_w;
    const uint32_t tile_index = ty * fp.grid_w + tx;                   // :74-76
    const uint32_t start = ranges[tile_index * 2 + 0];                 // :77
    const uint32_t end = ranges[tile_index * 2 + 1];
    // this lane's pixel inside the tile, and the first row / column of the wave's rectangle
    const uint32_t wy0 = QUAD ? (uint32_t)(wave >> 1) * 8u : (uint32_t)wave * 4u, wx0 = QUAD ? (uint32_t)(wave & 1) * 8u : 0u;
    const uint32_t ly = wy0 + (QUAD ? (uint32_t)(lane >> 3) : (uint32_t)(lane >> 4));
    const uint32_t py = ty * kTile + ly;
    const uint32_t px = tx * kTile + wx0 + (QUAD ? (uint32_t)(lane & 7) : (uint32_t)(lane & 15));
    const float fpx = (float)px, fpy = (float)py;                      // integer pixel coords (R1)
    const float tile_x0 = (float)(tx * kTile + wx0), wave_y0 = (float)(ty * kTile + wy0);
    constexpr float kRowsM1 = QUAD ? 7.0f : 3.0f, kColsM1 = QUAD ? 7.0f : 15.0f;

    float col0 = 0.0f, col1 = 0.0f, col2 = 0.0f, T = 1.0f;
    uint64_t done = mask_of(!(px < fp.width && py < fp.height));       // never stored (:147); a whole wave: 64 lanes
    if (done != ~0ull) {
        float4* wbatch = s_batch[wave];                                // this wave's quarter of the buffer
        Fetched nxt = fetch_splat(raster, sorted_id, start + lane, end);
        for (uint32_t i = start; i < end; i += 64) {                   // :81
            const bool keep = stage_splat(nxt, tile_x0, wave_y0, kRowsM1, kColsM1);
            const uint64_t kmask = __ballot(keep);
            const int n = (int)__popcll(kmask);
            // every staged colour finite (always, for a scene that is): the blend may add 0 * colour on skipping lanes
            const bool finite = mask_of(keep && !(fabsf(nxt.b.y) < INFINITY && fabsf(nxt.b.z) < INFINITY && fabsf(nxt.b.w) < INFINITY)) == 0ull;
            if (keep) {
                const uint32_t slot = mbcnt(kmask);                    // order-preserving compaction
                float* w = reinterpret_cast<float*>(wbatch + (slot >> 1) * kPairQuads) + (slot & 1u);
                w[0] = nxt.a.x; w[2] = nxt.a.y;                        // screen position
                w[4] = nxt.a.z; w[6] = nxt.a.w; w[8] = nxt.b.x;        // inverse 2x2 covariance
                w[10] = nxt.c.x; w[12] = nxt.c.y;                      // alpha, skip threshold
                w[14] = nxt.b.y; w[16] = nxt.b.z; w[18] = nxt.b.w;     // colour
                // a batch that ends on a single entry: the other half of its pair is never blended (`two` is false), but
                // the finite form multiplies its colour by a zero weight -- give it a colour that keeps 0 * c == 0
                // (the slots may hold anything, uninitialised LDS included)
                if ((n & 1) && slot == (uint32_t)n - 1u) { w[15] = 0.0f; w[17] = 0.0f; w[19] = 0.0f; }
            }
            // one wave writes and reads these slots and the DS operations of a wave execute in order: no barrier
            // instruction, only fences that keep the compiler from moving the reads below above the writes (:109)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            nxt = fetch_splat(raster, sorted_id, i + 64 + lane, end);  // prefetch next batch
            bool finished = false;
            if (finite) {
#pragma unroll 1
                for (int j = 0; j < n; j += 2) {                       // :112, two entries per step
                    blend_pair<EXACT, true>(wbatch, j >> 1, j + 1 < n, fpx, fpy, col0, col1, col2, T, done);
                    if (done == ~0ull) { finished = true; break; }     // this wave's rows are finished
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < n; j += 2) {
                    blend_pair<EXACT, false>(wbatch, j >> 1, j + 1 < n, fpx, fpy, col0, col1, col2, T, done);
                    if (done == ~0ull) { finished = true; break; }
                }
            }
            if (finished) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // ... nor the next batch's writes above these reads (:84)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    // :147-151 clamp + RGBA8 UNORM store, A = 255
    if (px < fp.width && py < fp.height) {
        uint32_t v = 0xFF000000u;
        v |= (uint32_t)(clampf(col0, 0.0f, 1.0f) * 255.0f + 0.5f);
        v |= (uint32_t)(clampf(col1, 0.0f, 1.0f) * 255.0f + 0.5f) << 8;
        v |= (uint32_t)(clampf(col2, 0.0f, 1.0f) * 255.0f + 0.5f) << 16;
        const uint32_t out_y = fp.compact_out ? krow * kTile + ly : py;
        rgba[(size_t)out_y * fp.width + px] = v;
    }
}

// Launch shape (gs_config.render_kernel; every shape gives the same pixels).  AUTO = the workgroup per tile with an
// 8 x 8 quadrant per wave, from these measurements on MI355X (round 4, exact mode, render bucket in ms, longest-first
// order; two pixels per lane with independent one-wave workgroups | workgroup per tile with 16 x 4 strips | with 8 x 8
// quadrants; tools/render_probe.py, profiles/r04_render_variants.txt):
//   8160 tiles (config C, uniform)        0.342 / 0.180 / 0.174
//   8160 tiles (config C-hard, capture)   0.831 / 0.534 / 0.500
//  32400 tiles (config D)                 0.557 / 0.537 / 0.516
// A quadrant has 20 % less perimeter than a strip, so fewer list entries survive a wave's rectangle test (3 - 7 % of
// the bucket); one launch slot per tile instead of two or four, and the blend loop takes two entries per step.
void launch_find_ranges(const FrameParams& fp, const uint32_t* sorted_tile, const SortParams* params,
                        uint32_t* ranges, hipStream_t stream) {
    uint32_t blocks = (fp.capacity / 4u + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0) blocks = 1;
    const TileMap map{fp.grid_w, fp.first_row, fp.row_stride};
    hipLaunchKernelGGL(k_find_ranges, dim3(blocks), dim3(256), 0, stream, sorted_tile, params, ranges, fp.hi16, map);
}

// Dispatch order of RenderGaussians' tiles: longest list first.  Workgroups are handed out in grid order and a tile's
// time grows with its list, so on a capture-like scene (a few tiles with tens of thousands of entries, most with
// hundreds) raster order leaves the heaviest tiles wherever they happen to lie and the launch ends on them; sorted by
// list length the tail is made of the shortest tiles.  The owned tiles are binned into 33 classes by bits(end - start),
// longest class first, any order inside a class (every order gives the same pixels).
// The LDS counters are fed per wave and class -- the lanes of a class elect a leader that adds their number -- not per
// tile (1024 threads on three or four addresses are slow).
__device__ __forceinline__ uint32_t class_slot(uint32_t cls, bool active, uint32_t* counters) {
    // For an active lane: the value of counters[cls] before this wave's lanes of that class, plus the lane's rank among
    // them.  The lanes of one class find each other with one ballot per class bit (as the radix Scatter finds equal
    // digits); the first lane of every class adds their number -- all classes of the wave in ONE atomic instruction.
    uint64_t same = __ballot(active);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const bool bit = (cls >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        same &= bit ? bal : ~bal;
    }
    const uint32_t rank = mbcnt(same);
    uint32_t base = 0u;
    if (active && rank == 0u) base = atomicAdd(&counters[cls], (uint32_t)__popcll(same));
    base = (uint32_t)__shfl((int)base, active ? __builtin_ctzll(same) : 0, 64);   // from the class leader (every lane takes part)
    return base + rank;
}

// Two launches over the tiles, 1024 per workgroup.  k_tile_classes: class and slot of every tile among the tiles of its
// class in its workgroup (arrival order), packed into scratch[c]; the workgroup's 33 class counts into wg_count.
// k_tile_scatter: every workgroup sums the counts of the others (first position of each class, longest class first,
// plus the tiles of that class in earlier workgroups) and places its tiles.  (One workgroup doing both took 10 us at
// 1920 x 1080 and 29 us at 4K: a single CU's instruction issue over 32,400 tiles.)
constexpr uint32_t kOrderTiles = 1024u;      // tiles per workgroup
__global__ __launch_bounds__(1024) void k_tile_classes(const uint32_t* __restrict__ ranges, uint32_t* __restrict__ scratch,
                                                        uint32_t* __restrict__ wg_count, uint32_t tiles, TileMap map) {
    __shared__ uint32_t s_count[33];
    if (threadIdx.x < 33) s_count[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t c = blockIdx.x * kOrderTiles + threadIdx.x;
    const bool active = c < tiles;                                       // whole waves vote in class_slot
    const uint2 r = active ? reinterpret_cast<const uint2*>(ranges)[global_tile(map, c)] : make_uint2(0u, 0u);
    const uint32_t len = r.y > r.x ? r.y - r.x : 0u;
    const uint32_t cls = len ? 32u - (uint32_t)__builtin_clz(len) : 0u;   // bits(len): 0 for an empty tile, else 1 .. 32
    const uint32_t slot = class_slot(cls, active, s_count);
    if (active) scratch[c] = cls | (slot << 6);
    __syncthreads();
    if (threadIdx.x < 33) wg_count[blockIdx.x * 33u + threadIdx.x] = s_count[threadIdx.x];
}

__global__ __launch_bounds__(1024) void k_tile_scatter(const uint32_t* __restrict__ scratch, const uint32_t* __restrict__ wg_count,
                                                        uint32_t* __restrict__ order, uint32_t tiles) {
    __shared__ uint32_t s_total[33], s_before[33], s_base[33];
    if (threadIdx.x < 33) { s_total[threadIdx.x] = 0u; s_before[threadIdx.x] = 0u; }
    __syncthreads();
    const uint32_t c = blockIdx.x * kOrderTiles + threadIdx.x;
    const uint32_t v = c < tiles ? scratch[c] : 0u;
    // class t = lane (33 of the 64 lanes of a wave), the sixteen waves take the workgroups b' = wave, wave + 16, ...
    {
        const uint32_t t = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        if (t < 33u) {
            uint32_t total = 0u, before = 0u;
            for (uint32_t b = wave; b < gridDim.x; b += 16u) {
                const uint32_t n = wg_count[b * 33u + t];
                total += n;
                before += b < blockIdx.x ? n : 0u;
            }
            if (total) atomicAdd(&s_total[t], total);
            if (before) atomicAdd(&s_before[t], before);
        }
    }
    __syncthreads();
    // first position of every class, longest class first: an inclusive DPP scan over the reversed totals (wave 0)
    if (threadIdx.x < 64) {
        const uint32_t n = threadIdx.x < 33 ? s_total[32 - threadIdx.x] : 0u;
        const uint32_t start = wave_inclusive_scan(n) - n;
        if (threadIdx.x < 33) s_base[32 - threadIdx.x] = start + s_before[32 - threadIdx.x];
    }
    __syncthreads();
    if (c < tiles) order[s_base[v & 63u] + (v >> 6)] = c;
}

// At most 1024 owned tiles (a 1/8 band of a 1920 x 1080 frame, BASELINE config A): one workgroup holds every count, so both
// steps are one launch -- the same classes, slots and class bases as the two kernels above (wgs = 1: nothing before us).
__global__ __launch_bounds__(1024) void k_tile_order_small(const uint32_t* __restrict__ ranges, uint32_t* __restrict__ order,
                                                            uint32_t tiles, TileMap map) {
    __shared__ uint32_t s_count[33], s_base[33];
    if (threadIdx.x < 33) s_count[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t c = threadIdx.x;
    const bool active = c < tiles;                                       // whole waves vote in class_slot
    const uint2 r = active ? reinterpret_cast<const uint2*>(ranges)[global_tile(map, c)] : make_uint2(0u, 0u);
    const uint32_t len = r.y > r.x ? r.y - r.x : 0u;
    const uint32_t cls = len ? 32u - (uint32_t)__builtin_clz(len) : 0u;
    const uint32_t slot = class_slot(cls, active, s_count);
    __syncthreads();
    if (threadIdx.x < 64) {                                              // longest class first, as k_tile_scatter
        const uint32_t n = threadIdx.x < 33 ? s_count[32 - threadIdx.x] : 0u;
        const uint32_t start = wave_inclusive_scan(n) - n;
        if (threadIdx.x < 33) s_base[32 - threadIdx.x] = start;
    }
    __syncthreads();
    if (active) order[s_base[cls] + slot] = c;
}

void launch_tile_order(const FrameParams& fp, const uint32_t* ranges, uint32_t* order, hipStream_t stream) {
    const uint32_t tiles = fp.rows_owned * fp.grid_w;
    if (tiles == 0) return;
    const TileMap map{fp.grid_w, fp.first_row, fp.row_stride};
    // order[0 .. tiles) is the table, then tiles words of scratch, then 33 counts per workgroup (tile_order_words below)
    const size_t all = (size_t)fp.grid_w * fp.grid_h;
    uint32_t* scratch = order + all;
    uint32_t* wg_count = order + 2 * all;
    const uint32_t wgs = (tiles + kOrderTiles - 1u) / kOrderTiles;
    if (wgs == 1u) {
        hipLaunchKernelGGL(k_tile_order_small, dim3(1), dim3(1024), 0, stream, ranges, order, tiles, map);
        return;
    }
    hipLaunchKernelGGL(k_tile_classes, dim3(wgs), dim3(1024), 0, stream, ranges, scratch, wg_count, tiles, map);
    hipLaunchKernelGGL(k_tile_scatter, dim3(wgs), dim3(1024), 0, stream, scratch, wg_count, order, tiles);
}
size_t tile_order_words(uint32_t grid_w, uint32_t grid_h) {
    const size_t all = (size_t)grid_w * grid_h;
    return 2 * all + ((all + kOrderTiles - 1u) / kOrderTiles) * 33u;
}

#ifdef GS_RENDER_STATS   // tools/probe only: the same kernel with its per-tile counters on
void launch_render_stats(const FrameParams& fp, const SplatRaster* raster, const uint32_t* sorted_id,
                         const uint32_t* ranges, uint8_t* rgba, uint4* stats, hipStream_t stream) {
    const uint32_t tiles = fp.rows_owned * fp.grid_w;
    if (tiles == 0) return;
    hipLaunchKernelGGL((k_render<true, 4, true>), dim3(tiles), dim3(64), 0, stream, fp, raster, sorted_id,
                       ranges, (const uint32_t*)nullptr, reinterpret_cast<uint32_t*>(rgba), stats);
}
#endif

void launch_render(const FrameParams& fp, const SplatRaster* raster, const uint32_t* sorted_id,
                   const uint32_t* ranges, const uint32_t* order, uint8_t* rgba, uint32_t render_mode,
                   uint32_t render_kernel, hipStream_t stream) {
    const uint32_t tiles = fp.rows_owned * fp.grid_w;
    if (tiles == 0) return;
    uint32_t* out = reinterpret_cast<uint32_t*>(rgba);
    const uint32_t px = render_kernel != 0u ? render_kernel : 17u;
#define GS_LAUNCH_RENDER(EXACT, PX)                                                                   \
    hipLaunchKernelGGL((k_render<EXACT, PX, false>), dim3(tiles * (4 / PX)), dim3(64), 0, stream, fp, \
                       raster, sorted_id, ranges, order, out, (uint4*)nullptr)
    if (px == 16 || px == 17) {   // workgroup-per-tile kernel: four 16 x 4 strips, or four 8 x 8 quadrants
#define GS_LAUNCH_WG(EXACT, QUAD) \
    hipLaunchKernelGGL((k_render_wg<EXACT, QUAD>), dim3(tiles), dim3(256), 0, stream, fp, raster, sorted_id, ranges, order, out)
        if (render_mode == 0u) { if (px == 17) GS_LAUNCH_WG(true, true); else GS_LAUNCH_WG(true, false); }
        else { if (px == 17) GS_LAUNCH_WG(false, true); else GS_LAUNCH_WG(false, false); }
#undef GS_LAUNCH_WG
    } else if (render_mode == 0u) {
        if (px == 1) GS_LAUNCH_RENDER(true, 1);
        else if (px == 2) GS_LAUNCH_RENDER(true, 2);
        else GS_LAUNCH_RENDER(true, 4);
    } else {
        if (px == 1) GS_LAUNCH_RENDER(false, 1);
        else if (px == 2) GS_LAUNCH_RENDER(false, 2);
        else GS_LAUNCH_RENDER(false, 4);
    }
#undef GS_LAUNCH_RENDER
}

} // namespace gs
