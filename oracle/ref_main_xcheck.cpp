// ref_main_xcheck.cpp -- CROSS-CHECK, NOT A PIN.  Runs the main() bodies of all nine of the reference's compute shaders
// on the path -- the text itself, cut out of the files where they lie under /root/reference by the Makefile, never
// copied into the repository -- as C++ over the reference's vendored glm 0.9.9.8:
//     ComputeShaders/InitSortList.comp     lines 45-151  (getGaussianTileExtents, getDepthKey, main: culls, colour and
//                                                          covariance store, atomic block reservation, emit loop)
//     ComputeShaders/FindRanges.comp       lines 30-71   (tryToWriteStart / tryToWriteEnd, main)
//     ComputeShaders/RenderGaussians.comp  lines 8-10, 47-54, 56-64 + 68-152 (group size, the shared batch, main:
//                                                          cooperative fetch + 2x2 inverse, blend loop, clamp +
//                                                          imageStore; lines 65-67 -- `aspectRatio` and `uv`, dead code
//                                                          that nothing reads -- are left out: `vec2(res.xy)` on a
//                                                          uvec2 needs a converting swizzle constructor glm lacks)
//     ComputeShaders/RadixSort/RadixSortIndirectSetup.comp 25-37, RadixSortCount.comp 37-91, RadixSortReduce.comp 31-72,
//     RadixSortScan.comp 25-71, RadixSortScanAdd.comp 31-66, RadixSortScatter.comp 46-171 (the `shared` arrays and
//     main of each), dispatched as RadixSort::computeSort does (Engine/Graphics/Sort/RadixSort.cpp:207-653)
// together with Common/Common.glsl, Common/CommonRadix.glsl and Common/GaussiansStructs.glsl, #included unmodified.  What each shader file
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
// the running sum).  RenderGaussians and the radix shaders have barriers: a workgroup is a set of fibers (ucontext) on
// one host thread, run round-robin from barrier to barrier (FiberGroup below): 256 per RenderGaussians tile, 64 per
// radix workgroup (RS_WORK_GROUP_SIZE, RadixSort.h:38; 1024 for Scan, :39), workgroups one after the other.
// subgroupAdd / subgroupExclusiveAdd / subgroupElect are emulated over subgroups of gl_SubgroupSize consecutive
// invocations with a scratch array between two workgroup barriers (every call site is in workgroup-uniform control
// flow).  The sort runs twice, with gl_SubgroupSize 32 and 64, and both results must equal a std::stable_sort by
// the low radixSortNumSortBits of tile << 32 | depth -- else the program fails.
//
// Glue, all of it here (the cut text is compiled unmodified):
//   * `using namespace glm`, GLM_FORCE_SWIZZLE, `#define inout`, `swizzle / scalar` as in ref_glsl_xcheck.cpp;
//   * gl_GlobalInvocationID / gl_LocalInvocationID / gl_WorkGroupID (set before an invocation resumes), barrier(),
//     atomicAdd(), imageStore() into a float
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

#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

namespace glm { namespace detail {
template <int N, typename T, qualifier Q, int E0, int E1, int E2, int E3>
vec<N, T, Q> operator/(const _swizzle<N, T, Q, E0, E1, E2, E3>& s, T f) { return s() / f; }
template <int N, typename T, qualifier Q, int E0, int E1, int E2, int E3>
_swizzle<N, T, Q, E0, E1, E2, E3>& operator/=(_swizzle<N, T, Q, E0, E1, E2, E3>& s, T f) { s = s() / f; return s; }
}}
using namespace glm;

// XCHECK_VARIANT selects what fills the freedom GLSL leaves ("parity envelope"):
//   0 (default, `ref_main_xcheck`)        the numeric contract of oracle/gs_oracle.h imposed by the overloads below --
//                                          the dump the fixtures ref_main_*.npz hold, compared bit for bit;
//   1 (`ref_main_xcheck_native`)          nothing imposed: glm's own mat4 * vec4 ((m0 x + m1 y) + (m2 z + m3 w)), glm's
//                                          normalize (v * inversesqrt(dot)), libm's expf;
//   2 (`ref_main_xcheck_gpu_like`)        as 1, with exp(x) = exp2f(x * log2(e)) -- the expansion GPU compilers use -- and
//                                          built with -ffp-contract=fast -mfma (every a*b+c the compiler can see is fused);
//   1 + -ffp-contract=fast -mfma          (`ref_main_xcheck_native_fma`).
// Variants 1 and 2 are other legal evaluations of the same text: tests/golden/make_envelope.py measures how far keys,
// tile extents, the sorted order and the pixels move between them and the contract.
#ifndef XCHECK_VARIANT
#define XCHECK_VARIANT 0
#endif
#if XCHECK_VARIANT == 0
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
#endif

#define inout
#include "Common/Common.glsl"            // -I /root/reference/vkGaussianSplatting/Resources/Shaders
#include "Common/GaussiansStructs.glsl"

static thread_local uvec3 gl_GlobalInvocationID, gl_LocalInvocationID, gl_WorkGroupID;
inline uint atomicAdd(uint& target, uint value) { const uint old = target; target += value; return old; }

// One workgroup = `size` fibers on this thread, run round-robin from barrier to barrier: invocation 0 up to its first
// barrier(), then invocation 1, ... -- lockstep by construction, deterministic, no host threads (every barrier() of
// the shaders on this path sits in workgroup-uniform control flow).
struct FiberGroup {
    static constexpr size_t kStack = 256 * 1024;
    static FiberGroup* active;
    std::vector<ucontext_t> ctx;
    std::vector<char> stacks, done;
    ucontext_t main_ctx;
    void (*job)() = nullptr;
    uint cur = 0;
    static void entry() { active->job(); active->done[active->cur] = 1; }          // then uc_link: back to run()
    void run(uint size, void (*fn)(), const std::function<void(uint)>& set_ids) {
        ctx.resize(size); done.assign(size, 0);
        if (stacks.size() < (size_t)size * kStack) stacks.resize((size_t)size * kStack);
        job = fn; active = this;
        for (uint l = 0; l < size; ++l) {
            getcontext(&ctx[l]);
            ctx[l].uc_stack.ss_sp = stacks.data() + (size_t)l * kStack;
            ctx[l].uc_stack.ss_size = kStack;
            ctx[l].uc_link = &main_ctx;
            makecontext(&ctx[l], entry, 0);
        }
        for (uint remaining = size; remaining;)
            for (uint l = 0; l < size; ++l)
                if (!done[l]) {
                    cur = l;
                    set_ids(l);
                    swapcontext(&main_ctx, &ctx[l]);
                    if (done[l]) --remaining;
                }
    }
    void barrier() { swapcontext(&ctx[cur], &main_ctx); }
};
FiberGroup* FiberGroup::active = nullptr;
static FiberGroup fibers;
inline void barrier() { fibers.barrier(); }

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
using ::barrier;
inline uint min(int a, uint b) { return (uint)a < b ? (uint)a : b; }   // min(ENTIRE_GROUP_SIZE, tileRange.y - i), :111
#if XCHECK_VARIANT == 1
inline float exp(float x) { return std::exp(x); }                       // libm's expf
#elif XCHECK_VARIANT == 2
inline float exp(float x) { return std::exp2(x * 0x1.715476p+0f); }     // v_exp_f32-style: exp2(x * log2 e)
#else
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
#endif
#define shared static
#include "render_8_10.inc"
#include "render_47_54.inc"
#define main shader_main
#include "render_56_152.inc"
#undef main
#undef shared
}

// ---- RadixSort/*.comp.  Each namespace declares what its file declares above the cut (buffers, push constants, the
//      specialization constant WORK_GROUP_SIZE with the value RadixSort.cpp:41-139 passes) and includes the rest.
#include "Common/CommonRadix.glsl"
namespace radix_rt {
static uint wg_size = 64u, gl_SubgroupSize = 32u;
static uint sg_scratch[1024];
using ::barrier;
inline uint subgroupExclusiveAdd(uint v) {
    const uint l = gl_LocalInvocationID.x;
    sg_scratch[l] = v;
    barrier();
    uint s = 0u;
    for (uint i = l - l % gl_SubgroupSize; i < l; ++i) s += sg_scratch[i];
    barrier();
    return s;
}
inline uint subgroupAdd(uint v) {
    const uint l = gl_LocalInvocationID.x, first = l - l % gl_SubgroupSize;
    sg_scratch[l] = v;
    barrier();
    uint s = 0u;
    for (uint i = first; i < first + gl_SubgroupSize && i < wg_size; ++i) s += sg_scratch[i];
    barrier();
    return s;
}
inline bool subgroupElect() { return gl_LocalInvocationID.x % gl_SubgroupSize == 0u; }   // every invocation is active
}
#define shared static
namespace radix_indirect_setup {     // RadixSortIndirectSetup.comp:8-23
using namespace radix_rt;
static const uint RS_WORK_GROUP_SIZE = 64u;
struct { GaussianCullData data; } cullData;
struct { RadixIndirectSetupData data; } indirectBuffer;
#define main shader_main
#include "radix_indirect_25_37.inc"
#undef main
}
namespace radix_count {              // RadixSortCount.comp:8-35
using namespace radix_rt;
static const uint WORK_GROUP_SIZE = 64u;
struct { RadixIndirectSetupData data; } indirectBuffer;
struct { GaussianSortData* sortData; } listBuffer;
struct { uvec4* buckets; } sumTable;
struct { uvec4 data; } pc;
#define main shader_main
#include "radix_count_37_91.inc"
#undef main
}
namespace radix_reduce {             // RadixSortReduce.comp:8-29
using namespace radix_rt;
static const uint WORK_GROUP_SIZE = 64u;
struct { RadixIndirectSetupData data; } indirectBuffer;
struct { uvec4* buckets; } sumTable;
struct { uvec4* data; } reduce;
#define main shader_main
#include "radix_reduce_31_72.inc"
#undef main
}
namespace radix_scan {               // RadixSortScan.comp:8-23
using namespace radix_rt;
static const uint WORK_GROUP_SIZE = 1024u;
struct { RadixIndirectSetupData data; } indirectBuffer;
struct { uvec4* data; } inputOutputBuffer;
#define main shader_main
#include "radix_scan_25_71.inc"
#undef main
}
namespace radix_scan_add {           // RadixSortScanAdd.comp:8-29
using namespace radix_rt;
static const uint WORK_GROUP_SIZE = 64u;
struct { RadixIndirectSetupData data; } indirectBuffer;
struct { uvec4* data; } reduceBuffer;
struct { uvec4* data; } sumTableBuffer;
#define main shader_main
#include "radix_scanadd_31_66.inc"
#undef main
}
namespace radix_scatter {            // RadixSortScatter.comp:11-44
using namespace radix_rt;
static const uint WORK_GROUP_SIZE = 64u;
struct { RadixIndirectSetupData data; } indirectBuffer;
struct { uvec4* data; } sumTableBuffer;
struct { GaussianSortData* sortData; } srcBuffer;
struct { GaussianSortData* sortData; } dstBuffer;
struct { uvec4 data; } pc;
#define main shader_main
#include "radix_scatter_46_171.inc"
#undef main
}
#undef shared

// vkCmdDispatch(workgroups, 1, 1) of a shader with local_size_x = size: one workgroup after the other
static void dispatch_1d(void (*fn)(), uint size, uint workgroups) {
    radix_rt::wg_size = size;
    for (uint wg = 0; wg < workgroups; ++wg)
        fibers.run(size, fn, [=](uint l) {
            gl_LocalInvocationID = uvec3(l, 0u, 0u);
            gl_WorkGroupID = uvec3(wg, 0u, 0u);
            gl_GlobalInvocationID = uvec3(wg * size + l, 0u, 0u);
        });
}

static uint32_t min_num_bits(uint32_t x) { uint32_t b = 0; while (x) { ++b; x >>= 1; } return b; }   // RadixSort.cpp:3-16

// RadixSort::initForScene + computeSort (RadixSort.cpp:144-205, 207-653) over list[0 .. capacity): returns the buffer
// the caller's list name refers to afterwards (the swap of :644-651).
static std::vector<GaussianSortData> reference_radix_sort(const std::vector<GaussianSortData>& list, uint32_t counter,
                                                          uint32_t capacity, uint32_t tiles, uint32_t subgroup_size) {
    const uint32_t WG = 64u, BINS = 16u;                                             // RadixSort.h:36-38
    const uint32_t max_count_groups = (capacity + WG - 1u) / WG;                     // :148-151
    const uint32_t max_reduce_blocks = (max_count_groups + WG - 1u) / WG;
    std::vector<uvec4> sum_table((size_t)max_count_groups * BINS, uvec4(0u)), reduce((size_t)max_reduce_blocks * BINS, uvec4(0u));
    std::vector<GaussianSortData> a(list), b(capacity);
    std::memset(b.data(), 0xFF, (size_t)capacity * sizeof(GaussianSortData));        // pingPongBuffer, :183-201
    const uint32_t sort_bits = ((32u + min_num_bits(tiles - 1u) + 3u) / 4u) * 4u;    // :203-204
    radix_rt::gl_SubgroupSize = subgroup_size;
    {   // IndirectSetup, :245-291
        using namespace radix_indirect_setup;
        cullData.data.numGaussiansToRender = uvec4(counter, capacity, 0u, 0u);
        dispatch_1d(shader_main, 1u, 1u);
    }
    const RadixIndirectSetupData ind = radix_indirect_setup::indirectBuffer.data;
    GaussianSortData *src = a.data(), *dst = b.data();
    for (uint32_t shift = 0u; shift < sort_bits; shift += 4u) {                       // :309
        radix_count::indirectBuffer.data = ind; radix_count::listBuffer.sortData = src;
        radix_count::sumTable.buckets = sum_table.data(); radix_count::pc.data = uvec4(shift, 0u, 0u, 0u);
        dispatch_1d(radix_count::shader_main, WG, ind.countSizeX);                    // :313-357
        radix_reduce::indirectBuffer.data = ind; radix_reduce::sumTable.buckets = sum_table.data();
        radix_reduce::reduce.data = reduce.data();
        dispatch_1d(radix_reduce::shader_main, WG, ind.reduceSizeX);                  // :386-424
        radix_scan::indirectBuffer.data = ind; radix_scan::inputOutputBuffer.data = reduce.data();
        dispatch_1d(radix_scan::shader_main, 1024u, 1u);                               // :437-466
        radix_scan_add::indirectBuffer.data = ind; radix_scan_add::reduceBuffer.data = reduce.data();
        radix_scan_add::sumTableBuffer.data = sum_table.data();
        dispatch_1d(radix_scan_add::shader_main, WG, ind.reduceSizeX);                // :496-534
        radix_scatter::indirectBuffer.data = ind; radix_scatter::sumTableBuffer.data = sum_table.data();
        radix_scatter::srcBuffer.sortData = src; radix_scatter::dstBuffer.sortData = dst;
        radix_scatter::pc.data = uvec4(shift, 0u, 0u, 0u);
        dispatch_1d(radix_scatter::shader_main, WG, ind.countSizeX);                  // :563-613
        std::swap(src, dst);                                                          // :638-641
    }
    return src == a.data() ? a : b;
}

static uint32_t ceil_pow2(uint32_t x) { uint32_t v = 1; while (v < x) v *= 2; return v; }   // Renderer.cpp:703-710

// input : u32 n, width, height, sh_mode; f32 view[16], proj[16], cam_pos[3]; f32 aos[n][84]
// output: u32 counter, capacity; f32 color[n][4], cov[n][4] (after InitSortList); u32 list[counter'][3] as emitted
//         (tile, depth, id; counter' = min(counter, capacity)); u32 sorted[counter'][3] (out of the reference's radix
//         shaders, checked against a stable sort); u32 ranges[tiles][2];
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

    // ---- the sort: the reference's six radix shaders, dispatched as RadixSort::computeSort does, once with subgroups of
    //      32 invocations and once with 64; both must give the stable order by the low sort bits of (tile, depth)
    std::vector<GaussianSortData> sorted = reference_radix_sort(list, counter, capacity, tiles, 32u);
    {
        const std::vector<GaussianSortData> sorted64 = reference_radix_sort(list, counter, capacity, tiles, 64u);
        const uint32_t sort_bits = ((32u + min_num_bits(tiles - 1u) + 3u) / 4u) * 4u;
        const uint64_t key_mask = sort_bits >= 64u ? ~0ull : ((1ull << sort_bits) - 1ull);
        std::vector<uint32_t> idx(e);
        for (uint32_t i = 0; i < e; ++i) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            const uint64_t ka = ((uint64_t)list[a].data.x << 32) | list[a].data.y, kb = ((uint64_t)list[b].data.x << 32) | list[b].data.y;
            return (ka & key_mask) < (kb & key_mask);
        });
        for (uint32_t i = 0; i < e; ++i) {
            const GaussianSortData& want = list[idx[i]];
            for (const GaussianSortData* got : {(const GaussianSortData*)&sorted[i], (const GaussianSortData*)&sorted64[i]})
                if (got->data.x != want.data.x || got->data.y != want.data.y || got->data.z != want.data.z) {
                    std::fprintf(stderr, "reference radix shaders: element %u differs from the stable order\n", i);
                    return 3;
                }
        }
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
        pc.resolution = uvec4(width, height, n, 0u);            // Subrenderer.cpp:317-323
        swapchainImage = Image{image.data(), width, height};
        for (uint32_t ty = 0; ty < grid_h; ++ty)
            for (uint32_t tx = 0; tx < grid_w; ++tx)
                fibers.run(256u, shader_main, [=](uint l) {
                    gl_LocalInvocationID = uvec3(l % 16u, l / 16u, 0u);
                    gl_GlobalInvocationID = uvec3(tx * 16u + l % 16u, ty * 16u + l / 16u, 0u);
                });
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
