// ref_main_xcheck.cpp -- CROSS-CHECK, NOT A PIN.  Runs the main() bodies of three of the reference's compute shaders
// -- the text itself, cut out of the files where they lie under /root/reference by the Makefile, never copied into the
// repository -- as C++ over the reference's vendored glm 0.9.9.8:
//     ComputeShaders/InitSortList.comp     lines 45-151  (getGaussianTileExtents, getDepthKey, main: culls, colour and
//                                                          covariance store, atomic block reservation, emit loop)
//     ComputeShaders/FindRanges.comp       lines 30-71   (tryToWriteStart / tryToWriteEnd, main)
//     ComputeShaders/RenderGaussians.comp  lines 8-10, 47-54, 56-64 + 68-152 (group size, the shared batch, main:
//                                                          cooperative fetch + 2x2 inverse, blend loop, clamp +
//                                                          imageStore; lines 65-67 -- `aspectRatio` and `uv`, dead code
//                                                          that nothing reads -- are left out: `vec2(res.xy)` on a
//                                                          uvec2 needs a converting swizzle constructor glm lacks)
// together with Common/Common.glsl and Common/GaussiansStructs.glsl, #included unmodified.  What each shader file
// declares ABOVE those lines -- the `layout(...)` buffer / UBO / push-constant blocks -- is declared here as C++ objects
// with the members the shader declares.  tests/test_oracle.py compares oracle/gs_oracle.c, and tests/test_parity_gpu.py
// the HIP path, against the committed dump (tests/golden/ref_main_*.npz, tests/golden/make_main_xcheck.py).
//
// What this catches: a shared misreading of the control flow and data flow of the three main() bodies by the two
// restatements in this repository (cull predicates, emit order inside a splat, which range end a boundary writes, the
// batch loop, add-then-test transmittance, the `continue` conditions).  What it cannot do is pin GLSL arithmetic --
// this is glm under a C++ compiler, not a GLSL compiler on a GPU -- so DESIGN.md keeps saying "parity unpinned".
//
// Execution model: InitSortList and FindRanges have no barriers: their invocations run one after the other in ascending
// gl_GlobalInvocationID (for InitSortList that IS the canonical emission order of DESIGN.md section 2: atomicAdd returns
// the running sum).  RenderGaussians has barriers: every 16x16 workgroup runs as 256 host threads with a pthread
// barrier standing in for barrier().
//
// Glue, all of it here (the cut text is compiled unmodified):
//   * `using namespace glm`, GLM_FORCE_SWIZZLE, `#define inout`, `swizzle / scalar` as in ref_glsl_xcheck.cpp;
//   * gl_GlobalInvocationID / gl_LocalInvocationID (thread_local), barrier(), atomicAdd(), imageStore() into a float
//     buffer, `shared` -> `static`, `min(int, uint)` (GLSL converts the int implicitly, C++ templates do not);
//   * the numeric contract of oracle/gs_oracle.h where GLSL leaves the evaluation open, so that the dump can be compared
//     bit for bit: mat4 * vec4 summed left to right (glm associates (m0 x + m1 y) + (m2 z + m3 w)), normalize(v) =
//     v / sqrt(dot(v, v)) (glm: v * inversesqrt), exp() = the pinned exp2 polynomial of the oracle (libm's expf is not
//     reproducible on a GPU), UNORM8 store = floor(c * 255 + 0.5).
#define GLM_FORCE_RADIANS
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#define GLM_FORCE_QUAT_DATA_WXYZ
#define GLM_FORCE_SWIZZLE
#include <glm/glm.hpp>

#include <pthread.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace glm { namespace detail {
template <int N, typename T, qualifier Q, int E0, int E1, int E2, int E3>
vec<N, T, Q> operator/(const _swizzle<N, T, Q, E0, E1, E2, E3>& s, T f) { return s() / f; }
template <int N, typename T, qualifier Q, int E0, int E1, int E2, int E3>
_swizzle<N, T, Q, E0, E1, E2, E3>& operator/=(_swizzle<N, T, Q, E0, E1, E2, E3>& s, T f) { s = s() / f; return s; }
}}
using namespace glm;

// numeric contract (see the header): a non-template overload is preferred over glm's templates
inline vec4 operator*(const mat4& m, const vec4& v) {
    vec4 r;
    for (int k = 0; k < 4; ++k) {
        float acc = m[0][k] * v.x;
        acc = acc + m[1][k] * v.y;
        acc = acc + m[2][k] * v.z;
        acc = acc + m[3][k] * v.w;
        r[k] = acc;
    }
    return r;
}
inline vec3 normalize(const vec3& v) {
    const float len = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return vec3(v.x / len, v.y / len, v.z / len);
}

#define inout
#include "Common/Common.glsl"            // -I /root/reference/vkGaussianSplatting/Resources/Shaders
#include "Common/GaussiansStructs.glsl"

static thread_local uvec3 gl_GlobalInvocationID, gl_LocalInvocationID;
inline uint atomicAdd(uint& target, uint value) { const uint old = target; target += value; return old; }

// ---- InitSortList.comp: binding 0 CamUBO, 1 GaussiansBuffer, 2 GaussiansSortListBuffer, 3 GaussiansCullDataBuffer,
//      push constants (InitSortList.comp:12-43)
namespace init_sort_list {
struct { mat4 viewMat; mat4 projMat; } ubo;
struct { GaussianData* gaussians; } gaussiansBuffer;
struct { GaussianSortData* sortData; } listBuffer;
struct { GaussianCullData data; } cullData;
struct { vec4 clipPlanes; vec4 camPos; uvec4 resolution; } pc;
#define main shader_main
#include "initsortlist_45_151.inc"
#undef main
}

// ---- FindRanges.comp: binding 0 GaussiansSortListBuffer, 1 GaussiansRangesBuffer, push constants (FindRanges.comp:12-28)
namespace find_ranges {
struct { GaussianSortData* sortData; } listBuffer;
struct { GaussianTileRangeData* rangeData; } rangesBuffer;
struct { uvec4 data; } pc;
#define main shader_main
#include "findranges_30_71.inc"
#undef main
}

// ---- RenderGaussians.comp: binding 0 GaussiansBuffer, 1 GaussiansSortListBuffer, 2 GaussiansRangesBuffer, 3 CamUBO,
//      4 swapchainImage, push constants (RenderGaussians.comp:14-45)
namespace render_gaussians {
struct { GaussianData* gaussians; } gaussiansBuffer;
struct { GaussianSortData* sortData; } listBuffer;
struct { GaussianTileRangeData* rangeData; } rangesBuffer;
struct { mat4 viewMat; mat4 projMat; } ubo;
struct { uvec4 resolution; } pc;
struct Image { float* rgba; uint32_t width, height; } swapchainImage;
inline void imageStore(Image& img, ivec2 p, vec4 c) {
    if (p.x < 0 || p.y < 0 || (uint32_t)p.x >= img.width || (uint32_t)p.y >= img.height) return;   // guarded by :147 anyway
    float* o = img.rgba + ((size_t)p.y * img.width + (size_t)p.x) * 4;
    o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
}
static pthread_barrier_t wg_barrier;
inline void barrier() { pthread_barrier_wait(&wg_barrier); }
inline uint min(int a, uint b) { return (uint)a < b ? (uint)a : b; }   // min(ENTIRE_GROUP_SIZE, tileRange.y - i), :111
// the pinned exp of oracle/gs_oracle.c (gso_exp), operation for operation
inline float exp(float x) {
    float t = x * 0x1.715476p+0f;
    t = t > -126.0f ? t : -126.0f;
    t = t < 126.0f ? t : 126.0f;
    const float n = std::nearbyintf(t);
    const float r = t - n;
    float p = 0x1.42059ap-13f;
    p = std::fmaf(p, r, 0x1.5f3e12p-10f);
    p = std::fmaf(p, r, 0x1.3b2d40p-7f);
    p = std::fmaf(p, r, 0x1.c6aeeap-5f);
    p = std::fmaf(p, r, 0x1.ebfbdcp-3f);
    p = std::fmaf(p, r, 0x1.62e430p-1f);
    p = std::fmaf(p, r, 1.0f);
    return std::ldexp(p, (int)n);
}
#define shared static
#include "render_8_10.inc"
#include "render_47_54.inc"
#define main shader_main
#include "render_56_152.inc"
#undef main
#undef shared
}

static uint32_t ceil_pow2(uint32_t x) { uint32_t v = 1; while (v < x) v *= 2; return v; }   // Renderer.cpp:703-710

// input : u32 n, width, height, sh_mode; f32 view[16], proj[16], cam_pos[3]; f32 aos[n][84]
// output: u32 counter, capacity; f32 color[n][4], cov[n][4] (after InitSortList); u32 list[counter'][3] as emitted
//         (tile, depth, id; counter' = min(counter, capacity)); u32 sorted[counter'][3] (stable sort by (tile, depth) --
//         the sort itself is not reference code: any stable sort gives this order); u32 ranges[tiles][2];
//         f32 image[h][w][4] (what imageStore receives); u8 rgba[h][w][4] (UNORM8)
int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s input.bin output.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    uint32_t hdr[4];
    float view_f[16], proj_f[16], cam[3];
    if (std::fread(hdr, 4, 4, f) != 4 || std::fread(view_f, 4, 16, f) != 16 || std::fread(proj_f, 4, 16, f) != 16 ||
        std::fread(cam, 4, 3, f) != 3) return 1;
    const uint32_t n = hdr[0], width = hdr[1], height = hdr[2], sh_mode = hdr[3];
    std::vector<float> aos((size_t)n * 84);
    if (std::fread(aos.data(), 4, aos.size(), f) != aos.size()) return 1;
    std::fclose(f);
    static_assert(sizeof(GaussianData) == 336, "GaussianData is the reference's 336-byte record");
    std::vector<GaussianData> gaussians(n);
    std::memcpy(gaussians.data(), aos.data(), (size_t)n * 336);
    mat4 viewMat, projMat;
    std::memcpy(&viewMat[0][0], view_f, 64);   // glm is column-major like the UBO (Renderer.cpp:531-538)
    std::memcpy(&projMat[0][0], proj_f, 64);
    const uint32_t grid_w = (width + 15) / 16, grid_h = (height + 15) / 16, tiles = grid_w * grid_h;
    const uint32_t capacity = ceil_pow2(n + 64u * 16u * tiles);                    // Renderer.cpp:725

    // ---- InitSortList: fills as Subrenderer.cpp:42-60, push constants as Subrenderer.cpp:152-160, one invocation per gaussian
    std::vector<GaussianSortData> list(capacity);
    std::memset(list.data(), 0xFF, (size_t)capacity * sizeof(GaussianSortData));
    {
        using namespace init_sort_list;
        ubo.viewMat = viewMat; ubo.projMat = projMat;
        gaussiansBuffer.gaussians = gaussians.data();
        listBuffer.sortData = list.data();
        cullData.data.numGaussiansToRender = uvec4(0u, capacity, 0u, 0u);
        pc.clipPlanes = vec4(0.1f, 100.0f, (float)n, 0.0f);
        pc.camPos = vec4(cam[0], cam[1], cam[2], (float)sh_mode);
        pc.resolution = uvec4(width, height, 0u, 0u);
        const uint32_t groups = (n + 31u) / 32u;                                      // Subrenderer.cpp:167-169
        for (uint32_t g = 0; g < groups * 32u; ++g) {
            gl_GlobalInvocationID = uvec3(g, 0u, 0u);
            shader_main();
        }
    }
    const uint32_t counter = init_sort_list::cullData.data.numGaussiansToRender.x;
    const uint32_t e = counter < capacity ? counter : capacity;                      // RadixSortIndirectSetup.comp:28

    // ---- the sort: stable, by (tile, depth) -- not reference code; the unused tail keeps its 0xFFFFFFFF fill
    std::vector<GaussianSortData> sorted(list);
    {
        std::vector<uint32_t> idx(e);
        for (uint32_t i = 0; i < e; ++i) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            const uint64_t ka = ((uint64_t)list[a].data.x << 32) | list[a].data.y, kb = ((uint64_t)list[b].data.x << 32) | list[b].data.y;
            return ka < kb;
        });
        for (uint32_t i = 0; i < e; ++i) sorted[i] = list[idx[i]];
    }

    // ---- FindRanges over the list CAPACITY (Subrenderer.cpp:205, 213-215), ranges cleared (Subrenderer.cpp:56-60)
    std::vector<GaussianTileRangeData> ranges(tiles);
    std::memset(ranges.data(), 0, (size_t)tiles * sizeof(GaussianTileRangeData));
    {
        using namespace find_ranges;
        listBuffer.sortData = sorted.data();
        rangesBuffer.rangeData = ranges.data();
        pc.data = uvec4(capacity, 0u, 0u, 0u);
        const uint32_t groups = (capacity + 15u) / 16u;
        for (uint32_t g = 0; g < groups * 16u; ++g) {
            if (g >= capacity) break;                     // the reference's capacity is a multiple of 16: no thread beyond it
            gl_GlobalInvocationID = uvec3(g, 0u, 0u);
            shader_main();
        }
    }

    // ---- RenderGaussians: one workgroup per tile, 256 threads with a barrier
    std::vector<float> image((size_t)width * height * 4, 0.0f);
    {
        using namespace render_gaussians;
        gaussiansBuffer.gaussians = gaussians.data();
        listBuffer.sortData = sorted.data();
        rangesBuffer.rangeData = ranges.data();
        ubo.viewMat = viewMat; ubo.projMat = projMat;
        pc.resolution = uvec4(width, height, 0u, 0u);
        swapchainImage = Image{image.data(), width, height};
        pthread_barrier_init(&wg_barrier, nullptr, 256);
        for (uint32_t ty = 0; ty < grid_h; ++ty)
            for (uint32_t tx = 0; tx < grid_w; ++tx) {
                std::vector<std::thread> th;
                th.reserve(256);
                for (uint32_t l = 0; l < 256; ++l)
                    th.emplace_back([=] {
                        gl_LocalInvocationID = uvec3(l % 16u, l / 16u, 0u);
                        gl_GlobalInvocationID = uvec3(tx * 16u + l % 16u, ty * 16u + l / 16u, 0u);
                        shader_main();
                    });
                for (auto& t : th) t.join();
            }
        pthread_barrier_destroy(&wg_barrier);
    }

    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    std::fwrite(&counter, 4, 1, o);
    std::fwrite(&capacity, 4, 1, o);
    for (uint32_t g = 0; g < n; ++g) std::fwrite(&gaussians[g].color, 4, 4, o);
    for (uint32_t g = 0; g < n; ++g) std::fwrite(&gaussians[g].covariance, 4, 4, o);
    for (uint32_t i = 0; i < e; ++i) std::fwrite(&list[i].data, 4, 3, o);
    for (uint32_t i = 0; i < e; ++i) std::fwrite(&sorted[i].data, 4, 3, o);
    for (uint32_t t = 0; t < tiles; ++t) std::fwrite(&ranges[t].range, 4, 2, o);
    std::fwrite(image.data(), 4, image.size(), o);
    std::vector<uint8_t> rgba((size_t)width * height * 4);
    for (size_t i = 0; i < rgba.size(); ++i) rgba[i] = (uint8_t)(image[i] * 255.0f + 0.5f);   // R8G8B8A8_UNORM store (Swapchain.cpp:27)
    std::fwrite(rgba.data(), 1, rgba.size(), o);
    std::fclose(o);
    return 0;
}
