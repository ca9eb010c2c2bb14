// ref_glsl_xcheck.cpp -- CROSS-CHECK, NOT A PIN.  Runs the text of the reference's own
// Resources/Shaders/Common/Common.glsl (#included where it lies under /root/reference, never copied)
// as C++ over the reference's vendored glm 0.9.9.8, and dumps what its four pure functions return
// for the splats of tests/golden/small_scene.npz:
//     getRotMat              Common.glsl:17-30
//     getCovarianceMatrix    Common.glsl:32-78
//     getScreenSpacePosition Common.glsl:80-89
//     getShColor/getShEval4  Common.glsl:94-170
// and, cut out of ComputeShaders/InitSortList.comp by the Makefile (see below),
//     getGaussianTileExtents InitSortList.comp:47-68
//     getDepthKey            InitSortList.comp:70-80
// tests/test_oracle.py compares oracle/gs_oracle.c against the committed dump.  What this catches is a
// shared MISREADING of the shader text by the two restatements in this repo (column-major constructors,
// matrix product order, swizzles, operand order); what it cannot do is pin GLSL's arithmetic: glm is a
// C++ library with one particular evaluation order, not a GLSL compiler, so DESIGN.md keeps saying
// "parity unpinned by the reference".
//
// Glue, all of it here (the GLSL file itself is compiled unmodified):
//   * `using namespace glm`, GLM_FORCE_SWIZZLE with -D_MSC_EXTENSIONS -fms-extensions (the reference is built by
//     MSVC, where glm's swizzle members are enabled the same way);
//   * `#define inout` (GLSL parameter qualifier; the array parameter decays to a pointer in C++);
//   * `swizzle / scalar` and `swizzle /= scalar`, which glm 0.9.9.8 lacks (it only has `*`): defined as
//     the component-wise IEEE division glm gives `vec / scalar`.
// Inputs that the shader computes OUTSIDE Common.glsl are handed in: the view-space position
// (InitSortList.comp:93) is evaluated here in the oracle's left-to-right order -- glm's own mat4*vec4
// associates (m0 x + m1 y) + (m2 z + m3 w) and is dumped beside it for information -- and the view
// direction (InitSortList.comp:124) as v / sqrt(dot(v, v)).
//
// Built by `make -C oracle ref` into oracle/_ref/ (git-ignored); run by tests/golden/make_glsl_xcheck.py.
#define GLM_FORCE_RADIANS
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#define GLM_FORCE_QUAT_DATA_WXYZ
#define GLM_FORCE_SWIZZLE
#include <glm/glm.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace glm { namespace detail {
template <int N, typename T, qualifier Q, int E0, int E1, int E2, int E3>
vec<N, T, Q> operator/(const _swizzle<N, T, Q, E0, E1, E2, E3>& s, T f) { return s() / f; }
template <int N, typename T, qualifier Q, int E0, int E1, int E2, int E3>
_swizzle<N, T, Q, E0, E1, E2, E3>& operator/=(_swizzle<N, T, Q, E0, E1, E2, E3>& s, T f) { s = s() / f; return s; }
}}
using namespace glm;
#define inout
#include "Common/Common.glsl"   // -I /root/reference/vkGaussianSplatting/Resources/Shaders

// The two helper functions of ComputeShaders/InitSortList.comp, getGaussianTileExtents (:47-68) and getDepthKey (:70-80),
// are pure apart from reading the camera UBO and the push constants.  The Makefile cuts exactly those lines out of the
// reference file into oracle/_ref/initsortlist_fns.inc (a build intermediate, never committed); the two data blocks
// they read are declared here with the members the shader declares (InitSortList.comp:13-17, 38-43).
struct { mat4 viewMat; mat4 projMat; } ubo;
struct { vec4 clipPlanes; vec4 camPos; uvec4 resolution; } pc;
#include "initsortlist_fns.inc"   // -I oracle/_ref

// input : u32 n, width, height; f32 view[16], proj[16], cam_pos[3]; f32 aos[n][84]
// output: f32 rot[n][9] (column-major), cov[n][3], screen[n][2], color[3 modes][n][3], viewpos_glm[n][4],
//         viewpos_in[n][4], f32 tan_half_fov (what glm folds `tan(FOV_Y * 0.5f)` to),
//         u32 extents[n][4] (getGaussianTileExtents on the cov above), u32 depth_key[n], u32 depth_key_defined[n]
//         (0 where the normalised depth reaches 1: `uint(1.0 * 2^32)` is out of range, undefined in GLSL and in C++)
int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s input.bin output.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    uint32_t hdr[3];
    float view_f[16], proj_f[16], cam[3];
    if (std::fread(hdr, 4, 3, f) != 3 || std::fread(view_f, 4, 16, f) != 16 || std::fread(proj_f, 4, 16, f) != 16 ||
        std::fread(cam, 4, 3, f) != 3) return 1;
    const uint32_t n = hdr[0];
    std::vector<float> aos((size_t)n * 84);
    if (std::fread(aos.data(), 4, aos.size(), f) != aos.size()) return 1;
    std::fclose(f);
    mat4 viewMat, projMat;
    std::memcpy(&viewMat[0][0], view_f, 64);   // glm is column-major like the UBO (Renderer.cpp:531-538)
    std::memcpy(&projMat[0][0], proj_f, 64);
    const float width = (float)hdr[1], height = (float)hdr[2];
    ubo.viewMat = viewMat; ubo.projMat = projMat;
    pc.clipPlanes = vec4(0.1f, 100.0f, (float)n, 0.0f);          // Subrenderer.cpp:152-160, Camera.cpp:4-5
    pc.camPos = vec4(cam[0], cam[1], cam[2], 0.0f);
    pc.resolution = uvec4(hdr[1], hdr[2], 0u, 0u);
    const ivec2 gridSize((int(pc.resolution.x) + TILE_SIZE - 1) / TILE_SIZE,          // InitSortList.comp:107-110
                         (int(pc.resolution.y) + TILE_SIZE - 1) / TILE_SIZE);
    std::vector<uint32_t> ext((size_t)n * 4), dkey(n), dkey_ok(n);

    std::vector<float> rot((size_t)n * 9), cov((size_t)n * 3), scr((size_t)n * 2), col((size_t)3 * n * 3),
        vpg((size_t)n * 4), vpi((size_t)n * 4);
    for (uint32_t g = 0; g < n; ++g) {
        const float* r = &aos[(size_t)g * 84];
        const vec3 worldPos(r[0], r[1], r[2]);
        // InitSortList.comp:93, left to right: ((M[0] x + M[1] y) + M[2] z) + M[3] w
        vec4 viewPos;
        for (int k = 0; k < 4; ++k) {
            float acc = viewMat[0][k] * worldPos.x;
            acc = acc + viewMat[1][k] * worldPos.y;
            acc = acc + viewMat[2][k] * worldPos.z;
            acc = acc + viewMat[3][k] * 1.0f;
            viewPos[k] = acc;
        }
        const vec4 viewPosGlm = viewMat * vec4(worldPos, 1.0f);
        const vec4 gRot(r[8], r[9], r[10], r[11]);
        const mat3x3 rm = getRotMat(gRot);
        const vec3 c = getCovarianceMatrix(width, height, vec3(r[4], r[5], r[6]), gRot, viewPos, viewMat);
        const vec4 s = getScreenSpacePosition(width, height, viewPos, projMat);
        const vec3 d = worldPos - vec3(cam[0], cam[1], cam[2]);
        const float len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        const vec3 dir(d.x / len, d.y / len, d.z / len);
        vec4 sh[16];
        for (int i = 0; i < 16; ++i) sh[i] = vec4(r[12 + 4 * i], r[13 + 4 * i], r[14 + 4 * i], r[15 + 4 * i]);
        for (int i = 0; i < 9; ++i) rot[(size_t)g * 9 + i] = rm[i / 3][i % 3];
        for (int i = 0; i < 3; ++i) cov[(size_t)g * 3 + i] = c[i];
        scr[(size_t)g * 2] = s.x; scr[(size_t)g * 2 + 1] = s.y;
        for (uint32_t mode = 0; mode < 3; ++mode) {
            const vec3 rgb = getShColor(dir, sh, mode);
            for (int i = 0; i < 3; ++i) col[((size_t)mode * n + g) * 3 + i] = rgb[i];
        }
        for (int i = 0; i < 4; ++i) { vpg[(size_t)g * 4 + i] = viewPosGlm[i]; vpi[(size_t)g * 4 + i] = viewPos[i]; }
        const uvec4 e = getGaussianTileExtents(g, viewPos, gridSize, c, width, height);   // InitSortList.comp:121
        for (int i = 0; i < 4; ++i) ext[(size_t)g * 4 + i] = e[i];
        const float nd = (-viewPos.z - pc.clipPlanes.x) / (pc.clipPlanes.y - pc.clipPlanes.x);
        dkey_ok[g] = (nd < 1.0f) ? 1u : 0u;                       // NaN and >= 1: the conversion below is undefined
        dkey[g] = dkey_ok[g] ? getDepthKey(viewPos.z) : 0u;       // InitSortList.comp:104
    }
    const float tan_half = tan(FOV_Y * 0.5f);   // glm::tan(float), as Common.glsl:53 reads under glm
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    std::fwrite(rot.data(), 4, rot.size(), o);
    std::fwrite(cov.data(), 4, cov.size(), o);
    std::fwrite(scr.data(), 4, scr.size(), o);
    std::fwrite(col.data(), 4, col.size(), o);
    std::fwrite(vpg.data(), 4, vpg.size(), o);
    std::fwrite(vpi.data(), 4, vpi.size(), o);
    std::fwrite(&tan_half, 4, 1, o);
    std::fwrite(ext.data(), 4, ext.size(), o);
    std::fwrite(dkey.data(), 4, dkey.size(), o);
    std::fwrite(dkey_ok.data(), 4, dkey_ok.size(), o);
    std::fclose(o);
    return 0;
}
