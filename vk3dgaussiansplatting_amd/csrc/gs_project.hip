// gs_project.hip -- InitSortList for gfx950, as three kernels:
//   k_project     : cull -> view/proj transform -> 2D covariance -> tile bbox -> SH colour
//                   (InitSortList.comp:82-127 + Common.glsl:17-170), one splat per lane, SoA loads
//   k_scan_blocks : exclusive scan of the per-workgroup tile counts + the IndirectSetup record
//                   (replaces the contended global atomicAdd of InitSortList.comp:131 and
//                   RadixSortIndirectSetup.comp:25-37)
//   k_emit        : one (tile, depth, id) element per overlapped tile (InitSortList.comp:132-150),
//                   load-balanced over OUTPUT elements, in the canonical deterministic order
//                   "ascending splat index, then row-major tile".
// Arithmetic follows the reference expression by expression, left to right, with contraction off
// (build flag -ffp-contract=off), IEEE division and sqrt: sort keys and tile extents are bit-exact
// against oracle/gs_oracle.c.  Paths are relative to /root/reference/vkGaussianSplatting/Resources/Shaders/.
#include <cstdlib>
#include "gs_device_utils.h"
#include "gs_internal.h"

namespace gs {

struct Mat3 { float m[3][3]; }; // column-major m[col][row], like GLSL mat3x3

// C = A*B, C[j][i] = (A[0][i]*B[j][0] + A[1][i]*B[j][1]) + A[2][i]*B[j][2]
__device__ __forceinline__ Mat3 mat3_mul(const Mat3& a, const Mat3& b) {
    Mat3 c;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = a.m[0][i] * b.m[j][0];
            acc = acc + a.m[1][i] * b.m[j][1];
            acc = acc + a.m[2][i] * b.m[j][2];
            c.m[j][i] = acc;
        }
    return c;
}
__device__ __forceinline__ Mat3 mat3_transpose(const Mat3& a) {
    Mat3 t;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) t.m[j][i] = a.m[i][j];
    return t;
}

// GLSL `M * v`, M column-major: ((M[0]*v.x + M[1]*v.y) + M[2]*v.z) + M[3]*v.w
__device__ __forceinline__ void mat4_mul_vec4(const float* m, float vx, float vy, float vz,
                                              float vw, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float acc = m[0 * 4 + r] * vx;
        acc = acc + m[1 * 4 + r] * vy;
        acc = acc + m[2 * 4 + r] * vz;
        acc = acc + m[3 * 4 + r] * vw;
        out[r] = acc;
    }
}

// Common/Common.glsl:17-30 (column-major constructor: col0, col1, col2)
__device__ __forceinline__ Mat3 get_rot_mat(float r, float x, float y, float z) {
    Mat3 m;
    m.m[0][0] = 1.0f - 2.0f * y * y - 2.0f * z * z;
    m.m[0][1] = 2.0f * x * y - 2.0f * r * z;
    m.m[0][2] = 2.0f * x * z + 2.0f * r * y;
    m.m[1][0] = 2.0f * x * y + 2.0f * r * z;
    m.m[1][1] = 1.0f - 2.0f * x * x - 2.0f * z * z;
    m.m[1][2] = 2.0f * y * z - 2.0f * r * x;
    m.m[2][0] = 2.0f * x * z - 2.0f * r * y;
    m.m[2][1] = 2.0f * y * z + 2.0f * r * x;
    m.m[2][2] = 1.0f - 2.0f * x * x - 2.0f * y * y;
    return m;
}

// Common/Common.glsl:32-78
__device__ __forceinline__ void get_covariance(const FrameParams& fp, const float scale[3],
                                               const float rot[4], const float pv_in[4],
                                               float cov[3]) {
    const float width = (float)fp.width, height = (float)fp.height;
    float pv[3] = {pv_in[0], pv_in[1], pv_in[2]};

    Mat3 rot_mat = get_rot_mat(rot[0], rot[1], rot[2], rot[3]);
    Mat3 scale_mat;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) scale_mat.m[j][i] = (i == j) ? scale[i] : 0.0f;
    Mat3 rs = mat3_mul(rot_mat, scale_mat);
    Mat3 sigma = mat3_mul(rs, mat3_transpose(rs));

    Mat3 w;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) w.m[c][r] = fp.view[c * 4 + r];

    const float tan_fov_y = fp.tan_fov_y;
    const float tan_fov_x = tan_fov_y * width / height;
    const float focal_x = width / (2.0f * tan_fov_x);
    const float focal_y = height / (2.0f * tan_fov_y);

    const float lim_x = tan_fov_x * fp.in_view_limit;
    const float lim_y = tan_fov_y * fp.in_view_limit;
    const float temp_x = pv[0] / pv[2];
    const float temp_y = pv[1] / pv[2];
    pv[0] = clampf(temp_x, -lim_x, lim_x) * pv[2];
    pv[1] = clampf(temp_y, -lim_y, lim_y) * pv[2];

    Mat3 j;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) j.m[c][r] = 0.0f;
    j.m[0][0] = focal_x / pv[2];
    j.m[1][1] = focal_y / pv[2];
    j.m[2][0] = -(focal_x * pv[0]) / (pv[2] * pv[2]);
    j.m[2][1] = -(focal_y * pv[1]) / (pv[2] * pv[2]);
    Mat3 jw = mat3_mul(j, w);
    Mat3 tmp = mat3_mul(jw, sigma);
    Mat3 sp = mat3_mul(tmp, mat3_transpose(jw));

    cov[0] = sp.m[0][0];
    cov[1] = sp.m[0][1];
    cov[2] = sp.m[1][1];
    cov[0] += 0.3f;
    cov[2] += 0.3f;
}

// Common/Common.glsl:94-138
__device__ __forceinline__ void sh_eval4(float dx, float dy, float dz, float sh[16]) {
    const float fX = -dx, fY = -dy, fZ = dz;
    float fC0, fC1, fS0, fS1, fTmpA, fTmpB, fTmpC;
    const float fZ2 = fZ * fZ;
    sh[0] = 0.2820947917738781f;
    sh[2] = 0.4886025119029199f * fZ;
    sh[6] = 0.9461746957575601f * fZ2 + -0.31539156525252f;
    sh[12] = fZ * (1.865881662950577f * fZ2 + -1.119528997770346f);
    fC0 = fX;
    fS0 = fY;
    fTmpA = -0.48860251190292f;
    sh[3] = fTmpA * fC0;
    sh[1] = fTmpA * fS0;
    fTmpB = -1.092548430592079f * fZ;
    sh[7] = fTmpB * fC0;
    sh[5] = fTmpB * fS0;
    fTmpC = -2.285228997322329f * fZ2 + 0.4570457994644658f;
    sh[13] = fTmpC * fC0;
    sh[11] = fTmpC * fS0;
    fC1 = fX * fC0 - fY * fS0;
    fS1 = fX * fS0 + fY * fC0;
    fTmpA = 0.5462742152960395f;
    sh[8] = fTmpA * fC1;
    sh[4] = fTmpA * fS1;
    fTmpB = 1.445305721320277f * fZ;
    sh[14] = fTmpB * fC1;
    sh[10] = fTmpB * fS1;
    fC0 = fX * fC1 - fY * fS1;
    fS0 = fX * fS1 + fY * fC1;
    fTmpC = -0.5900435899266435f;
    sh[15] = fTmpC * fC0;
    sh[9] = fTmpC * fS0;
}

// Number of owned tile rows (FrameParams: first_row + k * row_stride) below row y, i.e. the compact index of the first
// owned row >= y.
__device__ __forceinline__ int owned_rows_below(const FrameParams& fp, int y) {
    const int first = (int)fp.first_row;
    if (y <= first) return 0;
    return fp.row_stride == 1u ? y - first : (y - first + (int)fp.row_stride - 1) / (int)fp.row_stride;
}

__device__ __forceinline__ bool owns_every_row(const FrameParams& fp) {
    return fp.row_begin == 0u && fp.row_end == fp.grid_h && fp.row_stride == 1u;
}

// Conservative bound on a splat's radius in pixels from its view-space position and the upload-time bound sig2 on the
// largest eigenvalue of its 3-D covariance: radius = ceil(3 sqrt(lambda_max(Sigma'))) and lambda_max(Sigma') <=
// |J|_2^2 |W|_2^2 lambda_max(Sigma) + 0.3, with J, W as in getCovarianceMatrix (Common.glsl:49-69; tx, ty = the clamped
// x/z, y/z).  |J|_2^2 is the larger eigenvalue of J J^T = [[fx^2 (1 + tx^2), fx fy tx ty], [., fy^2 (1 + ty^2)]] / z^2,
// |W|_2^2 comes from the host (1 for a rigid view matrix).  Spectral norms: with Frobenius norms the radius came out
// up to 2.4 times too large and a 1/8 band kept a third of the blocks.  Widened by 2 % + 2 px.  Monotone in |tx|, |ty|,
// 1 / |vz| and sig2, which the box test relies on.
__device__ __forceinline__ float radius_bound(const FrameParams& fp, float tx, float ty, float vz, float sig2) {
    const float wdt = (float)fp.width, hgt = (float)fp.height;
    const float tfx = fp.tan_fov_y * wdt / hgt;
    const float fx = wdt / (2.0f * tfx), fy = hgt / (2.0f * fp.tan_fov_y);
    const float a = fx * fx * (1.0f + tx * tx), c = fy * fy * (1.0f + ty * ty), b = fx * fy * tx * ty;
    const float j2 = (0.5f * (a + c) + sqrtf(0.25f * (a - c) * (a - c) + b * b)) / (vz * vz);
    const float lam = j2 * fp.w_norm2 * sig2 * 1.02f + 0.31f;
    return 3.0f * sqrtf(lam) + 2.0f;
}

// True when no owned tile row intersects the pixel rows [ylo, yhi] (the callers have already widened the range by
// more than any rounding difference to the reference's own extents); a NaN anywhere gives false, i.e. the caller
// takes the normal path.
__device__ __forceinline__ bool misses_owned_rows(const FrameParams& fp, float ylo, float yhi) {
    if (!(ylo == ylo) || !(yhi == yhi) || !(ylo <= yhi)) return false;
    // rows as the reference cuts them (InitSortList.comp:61-64): int() truncates TOWARDS ZERO, so a splat whose lower
    // edge lies up to 16 px above the frame still lands in row 0
    const float lo_row = truncf(ylo * (1.0f / 16.0f)), hi_row = truncf(yhi * (1.0f / 16.0f)) + 1.0f;   // [lo, hi)
    const int y0 = lo_row < (float)fp.row_begin ? (int)fp.row_begin : (lo_row > 1e6f ? 1000000 : (int)lo_row);
    const int y1 = hi_row > (float)fp.row_end ? (int)fp.row_end : (hi_row < -1e6f ? -1000000 : (int)hi_row);
    if (y1 <= y0) return true;
    return owned_rows_below(fp, y1) - owned_rows_below(fp, y0) <= 0;
}

// k_emit walks at most kEmitSlice output elements per workgroup: a block of 256 splats (consecutive indices, or
// consecutive positions of the depth-sorted list) that emits more registers the slices after the first as helper
// records.  Whole workgroup; t is uniform.
__device__ __forceinline__ void register_emit_helpers(const FrameParams& fp, const SplatScratch& sc, uint32_t blk, uint32_t t) {
    if (t > kEmitSlice) {
        __shared__ uint32_t s_slot;
        const uint32_t extra = (t - 1u) / kEmitSlice;
        if (threadIdx.x == 0) {
            uint32_t slot = atomicAdd(&sc.help_count[fp.parity], extra);
            // no room (only when the element count overflows the list capacity): the owner does every slice
            if ((uint64_t)slot + extra > (uint64_t)emit_helpers(fp.capacity)) slot = kEmitNoHelp;
            sc.help_slot[blk] = slot;
            s_slot = slot;
        }
        __syncthreads();
        const uint32_t slot = s_slot;
        if (slot != kEmitNoHelp)
            for (uint32_t i = threadIdx.x; i < extra; i += kProjThreads) sc.help_list[slot + i] = make_uint2(blk, i + 1u);
    }
}

// The box around 64 consecutive splat positions (b0 = min xyz, max x; b1 = max yz, largest sig2; from the upload --
// the arrays are in Morton order, so it is small) against the owned tile rows.  If all eight corners are beyond the
// near plane, every splat's screen y lies between the corners' extremes and its radius is below the bound taken at the
// nearest corner depth; true when that range misses every owned row: none of the 64 splats emits anything.  A NaN /
// infinite corner, or one at or behind the near plane, keeps the wave.
__device__ __forceinline__ bool box_misses_owned_rows(const FrameParams& fp, const float4 b0, const float4 b1) {
    const float hgt = (float)fp.height;
    const float lim_x = fp.tan_fov_y * (float)fp.width / hgt * fp.in_view_limit, lim_y = fp.tan_fov_y * fp.in_view_limit;
    float ymin = 3.0e38f, ymax = -3.0e38f, zmin = 3.0e38f, txm = 0.0f, tym = 0.0f;
    bool in_front = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float vp[4], q[4];
        mat4_mul_vec4(fp.view, (c & 1) ? b0.w : b0.x, (c & 2) ? b1.x : b0.y, (c & 4) ? b1.y : b0.z, 1.0f, vp);
        mat4_mul_vec4(fp.proj, vp[0], vp[1], vp[2], vp[3], q);
        const float depth = -vp[2];
        // hardware reciprocals (1 ulp): far inside the 1 px / 0.01 % slack below
        const float sy = (1.0f - q[1] * __builtin_amdgcn_rcpf(q[3])) * 0.5f * hgt;
        in_front = in_front && depth > fp.near_plane && q[3] > 0.0f && (sy - sy == 0.0f);
        ymin = fminf(ymin, sy); ymax = fmaxf(ymax, sy); zmin = fminf(zmin, depth);
        // x/z and y/z are monotone along any segment in front of the camera: their extremes over the box are at corners
        const float rz = __builtin_amdgcn_rcpf(vp[2]);
        txm = fmaxf(txm, fabsf(vp[0] * rz)); tym = fmaxf(tym, fabsf(vp[1] * rz));
    }
    if (!in_front) return false;
    txm = fminf(txm * 1.0001f, lim_x); tym = fminf(tym * 1.0001f, lim_y);        // the clamp of Common.glsl:60-61
    const float rmax = radius_bound(fp, txm, tym, zmin, b1.z);
    // 1 px of slack for the rounding of the corner projections against the per-splat ones
    return misses_owned_rows(fp, ymin - rmax - 1.0f, ymax + rmax + 1.0f);
}

// First kernel of InitSortList in a context that owns a subset of the tile rows (multi-GPU): one THREAD per project
// block tests the boxes of its four waves and appends the blocks that may emit to band_list (with the mask of the
// waves that cannot), so that k_project runs over the survivors only.  A 1/8 band keeps 15-20 % of the blocks; testing
// inside k_project cost a workgroup launch, a dependent load and a barrier per rejected block (22.8 k of them at
// 5.8 M splats: 60 us of an 82 us launch).  A rejected block gets its zero block sum here and nothing else: k_emit
// leaves on a zero sum before it reads anything per splat.
constexpr int kCullThreads = 256;
__global__ __launch_bounds__(kCullThreads) void k_band_cull(const FrameParams fp, const SceneBuffers scene, const SplatScratch sc,
                                                            uint32_t num_blocks) {
    static_assert(kProjThreads / 64 == 4, "four wave records per project block");
    __shared__ uint32_t s_cnt[kCullThreads / 64];
    __shared__ uint32_t s_base;
    const uint32_t wrec = blockIdx.x * (uint32_t)kCullThreads + threadIdx.x;   // one thread per wave record, four per block
    const uint32_t b = wrec >> 2;
    const uint32_t n = fp.num_gaussians;
    bool skip = true;                                                 // a wave past the last splat
    if (b < num_blocks && (uint64_t)wrec * 64u < n) {
        const float4 b0 = reinterpret_cast<const float4*>(scene.block_bounds)[(size_t)wrec * 2 + 0];
        const float4 b1 = reinterpret_cast<const float4*>(scene.block_bounds)[(size_t)wrec * 2 + 1];
        skip = box_misses_owned_rows(fp, b0, b1);
    }
    const int lane = lane_id(), wave = wave_id();
    const uint32_t mask = (uint32_t)(__ballot(skip) >> (lane & ~3)) & 0xFu;     // the four waves of my block
    const bool leader = (lane & 3) == 0 && b < num_blocks;
    const bool survives = leader && mask != 0xFu;
    if (leader && !survives) {
        sc.block_sums[b] = 0u;
        if (fp.splat_first) sc.block_flags[b] = 0u;
        reinterpret_cast<uint32_t*>(sc.wave_wrote)[b] = 0u;           // none of its four waves writes records this frame
    }
    // one returning atomic per workgroup (on one address they complete at about 90 per microsecond)
    const uint64_t vote = __ballot(survives);
    if (lane == 0) s_cnt[wave] = (uint32_t)__builtin_popcountll(vote);
    __syncthreads();
    uint32_t before = 0u, total = 0u;
#pragma unroll
    for (int w = 0; w < kCullThreads / 64; ++w) {
        const uint32_t c = s_cnt[w];
        before += w < wave ? c : 0u;
        total += c;
    }
    if (total == 0u) return;
    if (threadIdx.x == 0) s_base = atomicAdd(&sc.help_count[2u + fp.parity], total);
    __syncthreads();
    if (survives)
        sc.band_list[s_base + before + (uint32_t)__builtin_popcountll(vote & ((1ull << lane) - 1ull))] = b | (mask << 28);
}

template <bool NT>
__device__ __forceinline__ float sh_load(const float* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// colour of a splat, InitSortList.comp:124-126 + Common.glsl:141-170 (k_project for the splats that emit, k_debug_colour for
// the others: one body, so both store the same bits)
template <bool NT>
__device__ __forceinline__ void splat_colour(const FrameParams& fp, const float* shp, const uint32_t n, const float px, const float py,
                                             const float pz, float res[3]) {
    const float ddx = px - fp.cam_pos[0], ddy = py - fp.cam_pos[1], ddz = pz - fp.cam_pos[2];
    const float len = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
    float basis[16];
    sh_eval4(ddx / len, ddy / len, ddz / len, basis);
    res[0] = 0.0f; res[1] = 0.0f; res[2] = 0.0f;
    if (fp.sh_mode == 0u) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                res[c] = res[c] + sh_load<NT>(shp + (size_t)(i * 3 + c) * n) * basis[i];
    } else if (fp.sh_mode == 1u) {
#pragma unroll
        for (int i = 1; i < 16; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                res[c] = res[c] + sh_load<NT>(shp + (size_t)(i * 3 + c) * n) * basis[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) res[c] = res[c] - 0.5f;
    } else if (fp.sh_mode == 2u) {
#pragma unroll
        for (int c = 0; c < 3; ++c) res[c] = res[c] + sh_load<NT>(shp + (size_t)c * n) * basis[0];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        res[c] = res[c] + 0.5f;
        res[c] = maxf(res[c], 0.0f);
    }
}

// Seven workgroups per CU: left to itself the compiler hoists the 48 SH loads and takes 117 VGPRs (four waves per SIMD);
// held to 72 it needs 65 without spilling, and the launch is 30 us shorter at config C (209 against 239 us).
#ifndef GS_PROJECT_MINBLOCKS
#define GS_PROJECT_MINBLOCKS 7
#endif
// NT_SH: the 48 spherical-harmonics planes (192 of the 236 bytes a splat's record takes) are loaded non-temporally -- a
// frame reads each of them once, and a scene larger than the 256 MiB Infinity Cache cannot keep them from one frame to
// the next anyway, while its other planes and the frame's lists can use the space (launch_project picks by scene size).
template <bool NT_SH>
__global__ __launch_bounds__(kProjThreads, GS_PROJECT_MINBLOCKS) void k_project(const FrameParams fp,
                                                           const SceneBuffers scene,
                                                           const SplatScratch sc, const uint32_t num_blocks) {
    __shared__ uint32_t s_wave_sum[kProjThreads / 64];
    __shared__ uint32_t s_wave_emits[kProjThreads / 64];
    __shared__ uint32_t s_wave_flags[kProjThreads / 64];
    // The 48-byte raster records of the workgroup's 256 splats are staged here and written out as one
    // contiguous 12 KB block of full cache lines: per-lane 16-byte stores at a 48-byte stride reached HBM
    // as 32-byte partial writes (WRITE_SIZE 355 MB per frame against 219 MB of payload, config C).
    __shared__ float4 s_raster[kProjThreads * 3];
    const uint32_t n = fp.num_gaussians;
    const bool band = !owns_every_row(fp);
    // One workgroup per block of 256 splats.  A contiguous band of tile rows: workgroup i takes the i-th block k_band_cull
    // kept (those beyond the count leave at once) and skips the waves it marked: only zero tile counts are written for
    // their splats (k_emit reads the counts of a workgroup that emits anything); RenderGaussians never sees their
    // records.  No loop over blocks here: as a loop body the kernel took 115 VGPRs instead of 65.
    const bool listed = band && fp.row_stride == 1u;    // interleaved rows: every block reaches some owned row
    uint32_t blk = blockIdx.x;
    bool wave_skip = false;
    if (listed) {
        if (blk >= sc.help_count[2u + fp.parity]) return;
        const uint32_t rec = sc.band_list[blk];
        blk = rec & 0x0FFFFFFFu;
        wave_skip = ((rec >> 28) >> wave_id()) & 1u;
    }
    const uint32_t g = blk * kProjThreads + threadIdx.x;
    uint32_t count = 0;
    bool kept = false;                                                          // passed both culls: its record is stored (N6)
    float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0, rec2 = rec0;   // culled splats: zero record

    if (g < n && !wave_skip) {
        const float px = scene.pos[g], py = scene.pos[(size_t)n + g], pz = scene.pos[2 * (size_t)n + g];
        float vp[4];
        mat4_mul_vec4(fp.view, px, py, pz, 1.0f, vp);                 // InitSortList.comp:93
        if (!(-vp[2] <= fp.near_plane)) {                            // :94
            float q[4];
            mat4_mul_vec4(fp.proj, vp[0], vp[1], vp[2], vp[3], q);    // :98
            const float ndc_x = q[0] / q[3];                          // :99
            const float ndc_y = q[1] / q[3];
            if (!(fabsf(ndc_x) > fp.ndc_cull || fabsf(ndc_y) > fp.ndc_cull)) { // :100
                // getDepthKey, :70-80 (float(MAX_UINT32) == 2^32)
                float nd = (-vp[2] - fp.near_plane) / (fp.far_plane - fp.near_plane);
                nd = clampf(nd, 0.0f, 1.0f);
                const uint32_t depth_key = f2u_sat(nd * 4294967296.0f);
                bool band_skip = false;

                // A context that owns a tile-row band can often tell from the position alone that a splat cannot
                // reach the band: radius = ceil(3 sqrt(lambda_max(Sigma'))) and lambda_max(Sigma') <=
                // |J|_2^2 |W|_2^2 lambda_max(Sigma) + 0.3 (spectral norms, radius_bound above), with lambda_max(Sigma) <=
                // sig2 from the upload and J, W as in getCovarianceMatrix (Common.glsl:49-69).  Widened by 2 % + 2 px; a
                // NaN or infinity anywhere makes the comparison false, i.e. the splat takes the normal path.
                if (band) {
                    const float hgt = (float)fp.height;
                    const float tfx = fp.tan_fov_y * (float)fp.width / hgt;
                    const float tx = clampf(vp[0] / vp[2], -tfx * fp.in_view_limit, tfx * fp.in_view_limit);
                    const float ty = clampf(vp[1] / vp[2], -fp.tan_fov_y * fp.in_view_limit, fp.tan_fov_y * fp.in_view_limit);
                    const float rmax = radius_bound(fp, tx, ty, vp[2], scene.sig2[g]);
                    const float sy_b = (1.0f - ndc_y) * 0.5f * hgt;          // screen y as below, to a rounding
                    band_skip = misses_owned_rows(fp, sy_b - rmax, sy_b + rmax);
                }
                if (!band_skip) {
                    float scale[3], rot[4], cov[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) scale[a] = scene.scale[a * (size_t)n + g];
#pragma unroll
                    for (int a = 0; a < 4; ++a) rot[a] = scene.rot[a * (size_t)n + g];
                    get_covariance(fp, scale, rot, vp, cov);              // :113-120

                    // getScreenSpacePosition, Common.glsl:80-89 (same proj*viewPos and division as above)
                    float sx = ndc_x, sy = -ndc_y;
                    sx = (sx + 1.0f) * 0.5f;
                    sy = (sy + 1.0f) * 0.5f;
                    sx = sx * (float)fp.width;
                    sy = sy * (float)fp.height;

                    // getGaussianTileExtents, InitSortList.comp:47-68
                    const float det = cov[0] * cov[2] - cov[1] * cov[1];
                    const float m = (cov[0] + cov[2]) * 0.5f;
                    const float lambda0 = m + sqrtf(maxf(m * m - det, 0.0f));
                    const float lambda1 = m - sqrtf(maxf(m * m - det, 0.0f));
                    const float radius = ceilf(3.0f * sqrtf(maxf(lambda0, lambda1)));
                    const int gw = (int)fp.grid_w, gh = (int)fp.grid_h;
                    const int min_x = clampi(f2i_sat((sx - radius) / 16.0f), 0, gw);
                    const int min_y = clampi(f2i_sat((sy - radius) / 16.0f), 0, gh);
                    int tx = f2i_sat((sx + radius) / 16.0f);
                    int ty = f2i_sat((sy + radius) / 16.0f);
                    const int max_x = clampi(tx == 2147483647 ? tx : tx + 1, 0, gw);
                    const int max_y = clampi(ty == 2147483647 ? ty : ty + 1, 0, gh);

                    // tile rows of this context (multi-GPU) as compact indices [k0, k1); identity for [0, grid_h)
                    int y0 = min_y > (int)fp.row_begin ? min_y : (int)fp.row_begin;
                    int y1 = max_y < (int)fp.row_end ? max_y : (int)fp.row_end;
                    if (y1 < y0) y1 = y0;
                    const int k0 = owned_rows_below(fp, y0), k1 = owned_rows_below(fp, y1);
                    count = (uint32_t)(max_x - min_x) * (uint32_t)(k1 - k0);   // :130

                    // :126-127.  The reference stores colour + covariance for every non-culled splat (N6).
                    // Colour is only ever read by RenderGaussians through the sorted list, so for a splat
                    // that emits no element here (off-screen inside the 1.3 NDC cull margin, or outside this
                    // context's tile-row band) the 192-byte SH read and the colour evaluation are skipped:
                    // unobservable in keys, ranges and pixels (SURVEY "F" list).
                    kept = true;
                    rec0 = make_float4(sx, sy, 0.0f, 0.0f);
                    rec2 = make_float4(0.0f, cov[0], cov[1], cov[2]);
                    if (count != 0u) {
                        // RenderGaussians.comp:94-107, once per splat instead of once per (tile, splat): the inverse of
                        // the 2x2 covariance (IEEE reciprocal, then three products) and the zero-determinant rule
                        float opacity = scene.opacity[g];
                        float inv_x = 0.0f, inv_y = 0.0f, inv_z = 0.0f;
                        if (det != 0.0f) {
                            const float det_inv = 1.0f / det;                  // :99
                            inv_x = cov[2] * det_inv;                          // :100
                            inv_y = -cov[1] * det_inv;
                            inv_z = cov[0] * det_inv;
                        } else {
                            opacity = 0.0f;                                    // :104
                        }
                        rec0.z = inv_x; rec0.w = inv_y;
                        rec2.x = opacity;
                        float res[3];
                        splat_colour<NT_SH>(fp, scene.sh + g, n, px, py, pz, res);
                        rec1 = make_float4(inv_z, res[0], res[1], res[2]);
                        sc.depth_key[g] = depth_key;
                        sc.extents[g] = make_uint2((uint32_t)min_x | ((uint32_t)k0 << 16),
                                                   (uint32_t)max_x | ((uint32_t)k1 << 16));
                    }
                }   // !band_skip
            }
        }
        sc.tiles_touched[g] = count;
    } else if (g < n) {
        sc.tiles_touched[g] = 0u;     // skipped wave: k_emit reads the counts of every splat of a workgroup that emits
    }

    s_raster[threadIdx.x * 3 + 0] = rec0;
    s_raster[threadIdx.x * 3 + 1] = rec1;
    s_raster[threadIdx.x * 3 + 2] = rec2;
    // per-workgroup total -> block_sums (input of the scan that replaces the atomic counter)
    const uint32_t wsum = wave_sum_to_lane63(count);
    const uint32_t wflags = (uint32_t)__popcll(__ballot(count != 0u));   // emitting splats (GS_SORT_RADIX4_SPLAT_FIRST)
    const bool wave_kept = __ballot(kept) != 0ull;
    if (lane_id() == 63) {
        s_wave_sum[wave_id()] = wsum;
        s_wave_flags[wave_id()] = wflags;
        // Which waves write their 3 KB of the block.  RenderGaussians reaches records only through the sorted list, i.e.
        // those of emitting splats; the reference also stores colour + covariance of every splat that passes the culls
        // (N6), readable here through gs_debug_read.  Full grid: a wave writes when any of its splats passed the culls
        // -- a wholly culled wave (a quarter of them at the benchmark pose; the arrays are in Morton order, so culled
        // splats come in runs) has nothing but zero records to store.  A context that owns a tile-row band: when any of
        // its splats emits into the band (most waves of a narrow band do not).  wave_wrote tells gs_debug_read which
        // records are this frame's (the others read back as zero, what a culled splat's scratch holds).
#ifndef GS_PROJECT_WRITE_ALL
#define GS_PROJECT_WRITE_ALL 0      /* tuning: 1 = a full-grid frame stores every wave's records, as before round 4 */
#endif
        const uint32_t wrote = (band ? wsum != 0u : (GS_PROJECT_WRITE_ALL || wave_kept)) ? 1u : 0u;
        s_wave_emits[wave_id()] = wrote;
        sc.wave_wrote[blk * (kProjThreads / 64) + wave_id()] = (uint8_t)wrote;
    }
    __syncthreads();
    {
        const uint32_t first = blk * kProjThreads;
        const uint32_t valid = (n - first) < (uint32_t)kProjThreads ? (n - first) : (uint32_t)kProjThreads;
        float4* out = reinterpret_cast<float4*>(sc.raster + first);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const uint32_t i = q * kProjThreads + threadIdx.x;
            if (i < valid * 3u && s_wave_emits[i / 192u]) out[i] = s_raster[i];
        }
    }
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kProjThreads / 64; ++w) t += s_wave_sum[w];
    if (threadIdx.x == 0) {
        sc.block_sums[blk] = t;
        if (fp.splat_first) sc.block_flags[blk] = s_wave_flags[0] + s_wave_flags[1] + s_wave_flags[2] + s_wave_flags[3];
    }
    // k_emit walks a workgroup's output range kEmitSlice elements per workgroup: a block that emits more (a few
    // hundred near splats hold half of a tile-row band's elements) registers the slices after the first as helper
    // records.  The order of the records depends on atomic arrival; what each one writes does not.  (With
    // GS_SORT_RADIX4_SPLAT_FIRST k_emit walks the splats in depth order: k_gather_sorted registers its blocks.)
    if (!fp.splat_first) register_emit_helpers(fp, sc, blk, t);
}

// One workgroup of 1024 threads: exclusive scan of block_sums (u64 running total so an overflowing
// scene is still counted correctly), then the IndirectSetup record.  Thread t owns the `per`
// consecutive entries [t*per, t*per+per) (16-byte loads; the arrays are zero-padded to 1024*per
// entries at upload): one pass to sum, one block scan, one pass to write -- no row-by-row carry chain.
// It also does the frame's clears (computeInitSortList's fills, Subrenderer.cpp:42-60): the tile ranges and the
// per-pass coarse digit totals of the sort -- two fill launches less per frame.
struct ScanJob { const uint32_t* sums; uint32_t* offsets; SortParams* params; uint32_t capacity;
                 uint32_t* host_note; };   // host-mapped word (or null): the element count + 1, for the host to read when it likes
constexpr uint32_t kScanClearWgs = 8;   // workgroups behind the scanning ones: the frame's clears, beside the scan
__global__ __launch_bounds__(1024) void k_scan_blocks(const ScanJob job0, const ScanJob job1, uint32_t jobs, uint32_t num_blocks,
                                                       uint4* __restrict__ zero_a, uint32_t n16_a,
                                                       uint4* __restrict__ zero_b, uint32_t n16_b,
                                                       uint32_t* __restrict__ next_help_count) {
    if (blockIdx.x >= jobs) {
        const uint32_t k = blockIdx.x - jobs;
        if (k == 0u && threadIdx.x == 0 && next_help_count) { next_help_count[0] = 0u; next_help_count[2] = 0u; }   // what the NEXT InitSortList launch adds to
        for (uint32_t i = k * 1024u + threadIdx.x; i < n16_a; i += kScanClearWgs * 1024u) zero_a[i] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t i = k * 1024u + threadIdx.x; i < n16_b; i += kScanClearWgs * 1024u) zero_b[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    // workgroup 0: the list of elements; workgroup 1, when asked for: a second, independent scan
    const ScanJob job = blockIdx.x == 0 ? job0 : job1;
    const uint32_t* __restrict__ block_sums = job.sums;
    uint32_t* __restrict__ block_offsets = job.offsets;
    SortParams* params = job.params;
    const uint32_t capacity = job.capacity;
    __shared__ uint64_t s_wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (((num_blocks + 1023u) / 1024u) + 3u) & ~3u;
    const uint4* in4 = reinterpret_cast<const uint4*>(block_sums + (size_t)tid * per);
    uint4* out4 = reinterpret_cast<uint4*>(block_offsets + (size_t)tid * per);
    auto sat = [](uint64_t x) { return (uint32_t)(x > 0xFFFFFFFFull ? 0xFFFFFFFFull : x); };
    constexpr uint32_t kHeld = 8;            // 16-byte groups a thread keeps in registers: up to 32 K blocks = 8.4 M splats
    const bool held = per / 4 <= kHeld;
    uint4 v[kHeld];
    uint64_t sum = 0;
    if (held) {
#pragma unroll
        for (uint32_t i = 0; i < kHeld; ++i) v[i] = i < per / 4 ? in4[i] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (uint32_t i = 0; i < kHeld; ++i) sum += (uint64_t)v[i].x + v[i].y + v[i].z + v[i].w;
    } else {
#pragma unroll 4
        for (uint32_t i = 0; i < per / 4; ++i) {
            const uint4 w = in4[i];
            sum += (uint64_t)w.x + w.y + w.z + w.w;
        }
    }
    const uint64_t inc = wave_inclusive_scan64(sum);
    if (lane == 63) s_wave_tot[wave] = inc;
    __syncthreads();
    uint64_t run = inc - sum;
    for (int w = 0; w < wave; ++w) run += s_wave_tot[w];
    if (tid == 1023) {
        const uint64_t counter = run + sum;
        const uint32_t e = counter < capacity ? (uint32_t)counter : capacity; // IndirectSetup.comp:28
        params->counter = counter;
        params->num_elems = e;
        params->num_groups = (e + kSortTile - 1) / kSortTile;
        params->groups_per_seg = (params->num_groups + kSegments - 1) / kSegments;
        params->overflow = counter > capacity ? 1u : 0u;
        // what GS_COUNT_AUTO goes by (gs_api.cpp): a fire-and-forget store into pinned host memory, nobody waits for it
        if (job.host_note) __hip_atomic_store(job.host_note, e + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (held) {
#pragma unroll
        for (uint32_t i = 0; i < kHeld; ++i) {
            uint4 o;
            o.x = sat(run); run += v[i].x;
            o.y = sat(run); run += v[i].y;
            o.z = sat(run); run += v[i].z;
            o.w = sat(run); run += v[i].w;
            if (i < per / 4) out4[i] = o;
        }
    } else {
#pragma unroll 4
        for (uint32_t i = 0; i < per / 4; ++i) {
            const uint4 w = in4[i];
            uint4 o;
            o.x = sat(run); run += w.x;
            o.y = sat(run); run += w.y;
            o.z = sat(run); run += w.z;
            o.w = sat(run); run += w.w;
            out4[i] = o;
        }
    }
}

// Emit: the splats [b*256, b*256+256) of project workgroup b own the output elements
// [block_offsets[b], +block_sums[b]).  Threads walk the OUTPUT range (coalesced stores); which splat
// owns an element is resolved per chunk in LDS from the scan of the 256 tile counts (see the loop).
// The owner workgroup of block b takes its first kEmitSlice elements; the further slices of heavy blocks are the helper
// records k_project registered, run by the first `helpers` workgroups of the same launch (those beyond the record
// count leave at once).  Without that split the launch lasted as long as its heaviest block: at 3840x2160 one
// block of near splats emits 105 k elements (102 rounds), and in a 1/8 tile-row band 186 blocks hold 55 % of the list.
constexpr int kEmitChunk = 4 * kProjThreads;   // output elements resolved per round of k_emit
static_assert(kEmitSlice % kEmitChunk == 0, "a slice is a whole number of rounds");

// SORTED (GS_SORT_RADIX4_SPLAT_FIRST): "splat g" is position g of the depth-sorted splat list -- its tile count rode
// through the depth passes as the payload, the splat index too, the extents are gathered through it -- and no depth
// word is written: the list is already in depth order, the tile-word passes that follow are stable and nothing later
// reads depth words.
template <bool SORTED>
__global__ __launch_bounds__(kProjThreads) void k_emit(const FrameParams fp, const SplatScratch sc,
                                                        uint32_t* __restrict__ out_lo,
                                                        uint32_t* __restrict__ out_hi,
                                                        uint32_t* __restrict__ out_id, uint32_t num_blocks,
                                                        uint32_t helpers, const uint32_t* __restrict__ sorted_ids,
                                                        const uint32_t* __restrict__ sorted_counts) {
    const uint32_t* __restrict__ sums = SORTED ? sc.sorted_sums : sc.block_sums;
    __shared__ uint32_t s_incl[kProjThreads];   // inclusive scan of tile counts
    __shared__ uint32_t s_wave_tot[kProjThreads / 64];
    __shared__ uint2 s_ext[kProjThreads];
    __shared__ uint32_t s_depth[kProjThreads];
    __shared__ uint32_t s_owner[kEmitChunk];
    __shared__ uint32_t s_wmax[kProjThreads / 64];
    const uint32_t n = SORTED ? sc.aux_params[0].num_elems : fp.num_gaussians;
    const int tid = threadIdx.x;
    // the helper workgroups come FIRST in the launch: the slices of the heaviest blocks start with the launch, and the
    // workgroups without a record are gone before the owners arrive
    uint32_t blk = blockIdx.x - helpers, slice = 0u;
    if (blockIdx.x < helpers) {
        const uint32_t h = blockIdx.x;
        if (h >= sc.help_count[fp.parity]) return;
        const uint2 rec = sc.help_list[h];
        blk = rec.x; slice = rec.y;
        // a registration that found no room still advanced the counter: records behind it are stale.  A record is
        // this frame's iff its block says so.
        if (blk >= num_blocks || slice == 0u) return;
        const uint32_t first = sc.help_slot[blk];
        if (first == kEmitNoHelp || h - first + 1u != slice) return;
        const uint32_t t = sums[blk];
        if (t <= kEmitSlice || slice > (t - 1u) / kEmitSlice) return;
    }
    const uint32_t g = blk * kProjThreads + tid;
    // every global read of the workgroup is issued up front (one memory latency instead of a chain);
    // extents/depth of splats that emit nothing are stale or uninitialised and never used
    const uint32_t total = sums[blk];
    // a context with a subset of the tile rows has mostly empty workgroups (some never wrote their per-splat
    // arrays, see k_project): look at the total first there
    if (!owns_every_row(fp) && total == 0) return;
    const uint32_t base = sc.block_offsets[blk];
    // the owner of a heavy block whose helper records did not fit the list walks every slice itself
    const bool alone = slice == 0u && total > kEmitSlice && sc.help_slot[blk] == kEmitNoHelp;
    uint32_t cnt = 0u, my_depth = 0u;              // SORTED: my_depth carries the splat index instead
    uint2 my_ext = make_uint2(0u, 0u);
    if (g < n) {
        if constexpr (SORTED) {
            my_depth = sorted_ids[g];
            cnt = fp.hi16 ? (uint32_t)reinterpret_cast<const uint16_t*>(sorted_counts)[g] : sorted_counts[g];
            my_ext = sc.extents[my_depth];           // the one gather of this path: 8 bytes per emitting splat
        } else {
            cnt = sc.tiles_touched[g];
            my_ext = sc.extents[g];
            my_depth = sc.depth_key[g];
        }
    }
    if (total == 0) return;
    if (base >= fp.capacity) return;             // whole block dropped (overflow, :143)
    s_ext[tid] = my_ext;
    s_depth[tid] = my_depth;
    const uint32_t inc = wave_inclusive_scan(cnt);
    if (lane_id() == 63) s_wave_tot[wave_id()] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave_id(); ++w) wbase += s_wave_tot[w];
    s_incl[tid] = wbase + inc;
    __syncthreads();

    const uint32_t g0 = blk * kProjThreads;
    const uint32_t my_excl = s_incl[tid] - cnt;
    const uint32_t c_begin = slice * kEmitSlice;
    const uint32_t c_end = alone || total - c_begin < kEmitSlice ? total : c_begin + kEmitSlice;
    // Owner of every output element: the list is cut into chunks of kEmitChunk elements; each splat marks the
    // first element it owns inside the chunk with its index + 1 and a prefix maximum spreads the marks (splat
    // index grows with the start offset), instead of one 8-step binary search per element.
    for (uint32_t c0 = c_begin; c0 < c_end; c0 += kEmitChunk) {
#pragma unroll
        for (int k = 0; k < kEmitChunk / kProjThreads; ++k) s_owner[k * kProjThreads + tid] = 0u;
        __syncthreads();
        if (cnt != 0u) {
            if (my_excl >= c0) {
                if (my_excl - c0 < (uint32_t)kEmitChunk) s_owner[my_excl - c0] = (uint32_t)tid + 1u;
            } else if (my_excl + cnt > c0) {
                s_owner[0] = (uint32_t)tid + 1u;                        // the splat that straddles the chunk start
            }
        }
        __syncthreads();
        uint4 o = *reinterpret_cast<const uint4*>(&s_owner[4 * tid]);   // thread t owns elements [4t, 4t + 4)
        o.y = o.y > o.x ? o.y : o.x;
        o.z = o.z > o.y ? o.z : o.y;
        o.w = o.w > o.z ? o.w : o.z;
        const uint32_t winc = wave_inclusive_max(o.w);
        if (lane_id() == 63) s_wmax[wave_id()] = winc;
        uint32_t prev = __shfl_up(winc, 1, 64);
        if (lane_id() == 0) prev = 0u;
        __syncthreads();
        for (int w = 0; w < wave_id(); ++w) prev = prev > s_wmax[w] ? prev : s_wmax[w];
        o.x = o.x > prev ? o.x : prev;
        o.y = o.y > prev ? o.y : prev;
        o.z = o.z > prev ? o.z : prev;
        o.w = o.w > prev ? o.w : prev;
        *reinterpret_cast<uint4*>(&s_owner[4 * tid]) = o;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kEmitChunk / kProjThreads; ++k) {
            const uint32_t jj = (uint32_t)(k * kProjThreads + tid);    // coalesced: consecutive lanes, consecutive slots
            const uint32_t j = c0 + jj;
            if (j < total) {
                const uint32_t s = s_owner[jj] - 1u;
                const uint32_t excl = s > 0u ? s_incl[s - 1u] : 0u;
                const uint32_t id_local = j - excl;
                const uint2 ext = s_ext[s];
                const uint32_t min_x = ext.x & 0xFFFFu, y0 = ext.x >> 16, max_x = ext.y & 0xFFFFu;
                const uint32_t wdt = max_x - min_x;
                // id_local / wdt (both < 2^24, exact in fp32) by reciprocal with a two-sided fix-up
                uint32_t ry = (uint32_t)((float)id_local * __builtin_amdgcn_rcpf((float)wdt));
                int32_t rx = (int32_t)(id_local - ry * wdt);
                if (rx < 0) { --ry; rx += (int32_t)wdt; }
                else if ((uint32_t)rx >= wdt) { ++ry; rx -= (int32_t)wdt; }
                // :137 with y = the y0-th + ry owned row: compact tile id (FrameParams), the global one on one GPU
                const uint32_t tile_key = (y0 + ry) * fp.grid_w + (min_x + (uint32_t)rx);
                const uint64_t out = (uint64_t)base + j;
                if (out < fp.capacity) {                                  // :143
                    if (fp.hi16) reinterpret_cast<uint16_t*>(out_hi)[out] = (uint16_t)tile_key;
                    else out_hi[out] = tile_key;
                    if constexpr (SORTED) out_id[out] = s_depth[s];
                    else { out_lo[out] = s_depth[s]; out_id[out] = g0 + s; }
                }
            }
        }
        __syncthreads();
    }
}

void launch_project(const FrameParams& fp, const SceneBuffers& scene, const SplatScratch& sc,
                    hipStream_t stream) {
    const uint32_t blocks = (fp.num_gaussians + kProjThreads - 1) / kProjThreads;
    if (blocks == 0) return;
    const bool listed = !(fp.row_begin == 0u && fp.row_end == fp.grid_h) && fp.row_stride == 1u;   // a contiguous band
    if (listed) hipLaunchKernelGGL(k_band_cull, dim3((blocks * 4u + (uint32_t)kCullThreads - 1u) / (uint32_t)kCullThreads), dim3(kCullThreads), 0, stream,
                                   fp, scene, sc, blocks);
#ifndef GS_PROJECT_NT_SH_ABOVE
#define GS_PROJECT_NT_SH_ABOVE ((size_t)256u << 20)       /* scene bytes above which the SH planes are loaded non-temporally */
#endif
    if ((size_t)fp.num_gaussians * 236u > GS_PROJECT_NT_SH_ABOVE)
        hipLaunchKernelGGL(k_project<true>, dim3(blocks), dim3(kProjThreads), 0, stream, fp, scene, sc, blocks);
    else
        hipLaunchKernelGGL(k_project<false>, dim3(blocks), dim3(kProjThreads), 0, stream, fp, scene, sc, blocks);
}

// gs_debug_read(GS_BUF_COLOR), full-grid contexts: the reference stores a colour for EVERY splat that passes the two culls
// (InitSortList.comp:124-127, before it knows whether the splat touches a tile; SURVEY "preserve" item N6); k_project
// evaluates it only for the splats that emit (nothing else is ever read by a frame).  This fills in the others on demand,
// from the last frame's camera: out[g] = (r, g, b, 1) for a splat whose record k_project stored (its covariance carries
// the +0.3 dilation, so .cx != 0 marks it) but which touched no tile, (0, 0, 0, 0) otherwise.
__global__ __launch_bounds__(256) void k_debug_colour(const FrameParams fp, const SceneBuffers scene, const SplatScratch sc, float4* out) {
    const uint32_t n = fp.num_gaussians;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n) return;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sc.wave_wrote[g >> 6] && sc.raster[g].cx != 0.0f && sc.tiles_touched[g] == 0u) {
        float res[3];
        splat_colour<false>(fp, scene.sh + g, n, scene.pos[g], scene.pos[(size_t)n + g], scene.pos[2 * (size_t)n + g], res);
        o = make_float4(res[0], res[1], res[2], 1.0f);
    }
    out[g] = o;
}

void launch_debug_colour(const FrameParams& fp, const SceneBuffers& scene, const SplatScratch& sc, float* out_rgba, hipStream_t stream) {
    const uint32_t blocks = (fp.num_gaussians + 255u) / 256u;
    if (blocks) hipLaunchKernelGGL(k_debug_colour, dim3(blocks), dim3(256), 0, stream, fp, scene, sc, reinterpret_cast<float4*>(out_rgba));
}

void launch_scan_blocks(const FrameParams& fp, const SplatScratch& sc, SortParams* params,
                        uint32_t* ranges, uint32_t* coarse, hipStream_t stream) {
    const uint32_t blocks = (fp.num_gaussians + kProjThreads - 1) / kProjThreads;
    // ranges: [grid_w * grid_h][2] uint32 (hipMalloc'd, so 16-byte aligned; padded to a multiple of 16 bytes by the caller)
    const uint32_t n16_ranges = (fp.grid_w * fp.grid_h * 2u + 3u) / 4u;
    const uint32_t n16_coarse = (uint32_t)(kMaxSortPasses * kBins * kCoarse) / 4u;
    const ScanJob elements{sc.block_sums, sc.block_offsets, params, fp.capacity, sc.elems_note};
    // GS_SORT_RADIX4_SPLAT_FIRST: the emitting splats are scanned beside the elements (a second workgroup)
    const ScanJob splats{sc.block_flags, sc.flag_offsets, sc.aux_params, fp.num_gaussians, nullptr};
    const uint32_t jobs = fp.splat_first ? 2u : 1u;
    hipLaunchKernelGGL(k_scan_blocks, dim3(jobs + kScanClearWgs), dim3(1024), 0, stream, elements, splats, jobs, blocks,
                       reinterpret_cast<uint4*>(ranges), n16_ranges, reinterpret_cast<uint4*>(coarse), n16_coarse,
                       sc.help_count + (fp.parity ^ 1u));
}

void launch_emit(const FrameParams& fp, const SplatScratch& sc, const SortBuffers& sb,
                 hipStream_t stream) {
    const uint32_t blocks = (fp.num_gaussians + kProjThreads - 1) / kProjThreads;
    if (blocks == 0) return;
    const uint32_t helpers = emit_helpers(fp.capacity);
    hipLaunchKernelGGL((k_emit<false>), dim3(helpers + blocks), dim3(kProjThreads), 0, stream, fp, sc, sb.lo[0],
                       sb.hi[0], sb.id[0], blocks, helpers, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
}

// ---- GS_SORT_RADIX4_SPLAT_FIRST ------------------------------------------------------------------------------------
// The list the eight depth passes sort: (depth word | tile count | splat) of every splat that emits anything, in splat
// order.  The tile count rides as the payload in the place of the tile word (16 bits wide when the tile ids are), so the
// passes are the frame's own depth passes -- same kernels, same shrinking depth words -- over a list a quarter as long.
// The truncation of InitSortList.comp:143 is applied here, where the reference applies it: an element is dropped when
// its position in SPLAT order is beyond the capacity, so every count is clipped against that position before
// anything is reordered (only an overflowing frame clips anything).
__global__ __launch_bounds__(kProjThreads) void k_splat_list(const FrameParams fp, const SplatScratch sc,
                                                              uint32_t* __restrict__ out_depth, uint32_t* __restrict__ out_count,
                                                              uint32_t* __restrict__ out_id) {
    __shared__ uint32_t s_cnt[kProjThreads / 64], s_flag[kProjThreads / 64];
    const uint32_t blk = blockIdx.x, g = blk * kProjThreads + threadIdx.x;
    // the helper counter k_sorted_sums adds to (sixteen launches later): this path always uses the one of parity 0
    if (blk == 0u && threadIdx.x == 0u) sc.help_count[fp.parity] = 0u;
    if (sc.block_sums[blk] == 0u) return;            // nothing emits (or k_band_cull rejected the block: stale arrays)
    const uint32_t cnt = g < fp.num_gaussians ? sc.tiles_touched[g] : 0u;
    const uint32_t dk = cnt != 0u ? sc.depth_key[g] : 0u;
    const uint64_t off0 = sc.block_offsets[blk];     // saturated at 2^32 - 1: beyond any capacity
    const uint32_t fpos0 = sc.flag_offsets[blk];
    const uint32_t inc = wave_inclusive_scan(cnt);
    const uint64_t fb = __ballot(cnt != 0u);
    if (lane_id() == 63) { s_cnt[wave_id()] = inc; s_flag[wave_id()] = (uint32_t)__popcll(fb); }
    __syncthreads();
    uint64_t off = off0 + (inc - cnt);
    uint32_t pos = fpos0 + (uint32_t)__popcll(fb & ((1ull << lane_id()) - 1ull));
    for (int w = 0; w < wave_id(); ++w) { off += s_cnt[w]; pos += s_flag[w]; }
    if (cnt != 0u) {
        const uint64_t room = off < fp.capacity ? (uint64_t)fp.capacity - off : 0ull;
        const uint32_t kept = room < cnt ? (uint32_t)room : cnt;
        out_depth[pos] = dk;
        if (fp.hi16) reinterpret_cast<uint16_t*>(out_count)[pos] = (uint16_t)kept;   // <= owned tiles <= 65535
        else out_count[pos] = kept;
        out_id[pos] = g;
    }
}

// Sums of the tile counts per 256 positions of the sorted list (input of the second scan) and k_emit's helper records.
// One WAVE per 256 positions (four counts per lane, one 8- or 16-byte load), eight of them per workgroup: nothing
// crosses waves.
constexpr int kSumsBlocksPerWg = 8;
__global__ __launch_bounds__(kProjThreads) void k_sorted_sums(const FrameParams fp, const SplatScratch sc,
                                                               const uint32_t* __restrict__ sorted_counts, uint32_t num_blocks) {
    const uint32_t ve = sc.aux_params[0].num_elems;
    const int lane = lane_id();
#pragma unroll
    for (int it = 0; it < kSumsBlocksPerWg / (kProjThreads / 64); ++it) {
        const uint32_t blk = blockIdx.x * kSumsBlocksPerWg + (uint32_t)(it * (kProjThreads / 64) + wave_id());
        if (blk >= num_blocks) return;
        const uint32_t i0 = blk * kProjThreads + 4u * (uint32_t)lane;   // four consecutive positions
        uint32_t c[4] = {0u, 0u, 0u, 0u};
        if (i0 + 3u < ve) {
            if (fp.hi16) {
                const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(sorted_counts) + i0);
                c[0] = v.x & 0xFFFFu; c[1] = v.x >> 16; c[2] = v.y & 0xFFFFu; c[3] = v.y >> 16;
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(sorted_counts + i0);
                c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < ve) c[k] = fp.hi16 ? (uint32_t)reinterpret_cast<const uint16_t*>(sorted_counts)[i0 + k] : sorted_counts[i0 + k];
        }
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)wave_sum_to_lane63(c[0] + c[1] + c[2] + c[3]), 63);
        if (lane == 0) sc.sorted_sums[blk] = t;
        // k_emit's helper records for a heavy block (register_emit_helpers, by one wave)
        if (t > kEmitSlice) {
            const uint32_t extra = (t - 1u) / kEmitSlice;
            uint32_t slot = 0u;
            if (lane == 0) {
                slot = atomicAdd(&sc.help_count[fp.parity], extra);
                if ((uint64_t)slot + extra > (uint64_t)emit_helpers(fp.capacity)) slot = kEmitNoHelp;
                sc.help_slot[blk] = slot;
            }
            slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
            if (slot != kEmitNoHelp)
                for (uint32_t i = (uint32_t)lane; i < extra; i += 64u) sc.help_list[slot + i] = make_uint2(blk, i + 1u);
        }
    }
}

void launch_splat_list(const FrameParams& fp, const SplatScratch& sc, const SortBuffers& sb, hipStream_t stream) {
    const uint32_t blocks = (fp.num_gaussians + kProjThreads - 1) / kProjThreads;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_splat_list, dim3(blocks), dim3(kProjThreads), 0, stream, fp, sc, sb.lo[1], sb.hi[1], sb.id[1]);
}

void launch_gather_sorted(const FrameParams& fp, const SplatScratch& sc, const SortBuffers& sb, int sorted, hipStream_t stream) {
    const uint32_t blocks = (fp.num_gaussians + kProjThreads - 1) / kProjThreads;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_sorted_sums, dim3((blocks + kSumsBlocksPerWg - 1) / kSumsBlocksPerWg), dim3(kProjThreads), 0, stream, fp, sc,
                       (const uint32_t*)sb.hi[sorted], blocks);
    // the offsets of the 256-position blocks of the sorted list (the dispatch record of the elements stays the first
    // scan's: same length, and it knows about an overflow)
    const ScanJob sorted_blocks{sc.sorted_sums, sc.block_offsets, sc.aux_params + 1, fp.capacity};
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, sorted_blocks, sorted_blocks, 1u, blocks,
                       (uint4*)nullptr, 0u, (uint4*)nullptr, 0u, (uint32_t*)nullptr);
}

void launch_emit_sorted(const FrameParams& fp, const SplatScratch& sc, const SortBuffers& sb, int sorted, hipStream_t stream) {
    const uint32_t blocks = (fp.num_gaussians + kProjThreads - 1) / kProjThreads;
    if (blocks == 0) return;
    const uint32_t helpers = emit_helpers(fp.capacity);
    hipLaunchKernelGGL((k_emit<true>), dim3(helpers + blocks), dim3(kProjThreads), 0, stream, fp, sc, (uint32_t*)nullptr,
                       sb.hi[0], sb.id[0], blocks, helpers, (const uint32_t*)sb.id[sorted], (const uint32_t*)sb.hi[sorted]);
}

} // namespace gs
