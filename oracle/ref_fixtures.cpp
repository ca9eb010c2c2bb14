// ref_fixtures.cpp -- driver that runs the REFERENCE's own vendored code (glm 0.9.9.8 and
// Engine/SMath.h, compiled where they lie under /root/reference, never copied) to generate the
// golden vectors committed under tests/golden/.  Built only in the authoring container by
// `make -C oracle ref` into oracle/_ref/ (git-ignored); /root/reference does not exist on the
// GPU box, where only the committed JSON fixtures are used.
//
// What it pins (the only parts of the path that are host C++ compilable without Vulkan/Win32):
//   * camera matrices: Camera::updateDirVectors / updateMatrices / recalculate -- the text of
//     Engine/Graphics/Camera.cpp:4-54, cut into a temporary directory by the Makefile (Camera.cpp as a file needs
//     pch.h / Window / GLFW) -- over glm::lookAt / perspective / normalize / cross with the project's defines
//     GLM_FORCE_RADIANS; GLM_FORCE_DEPTH_ZERO_TO_ONE; GLM_FORCE_QUAT_DATA_WXYZ (vkGaussianSplatting.vcxproj:50).
//     Glue: a class Camera with the members Camera.h:14-37 declares and a Window that answers getFramebufferSize /
//     getAspectRatio.  `sin(this->yaw)` on a float picks sinf under MSVC (its <cmath> puts the float overloads into
//     the global namespace) and may pick sin(double) elsewhere: the text is compiled both ways and the program fails
//     if the two disagree on any fixture pose;
//   * Morton codes: SMath::encodeZorderCurve (Engine/SMath.h:24-34);
//   * the host formulas that size the path: Renderer::getNumTiles / getCeilPowTwo (the text of Renderer.cpp:696-710)
//     and RadixSort::getMinNumBits (RadixSort.cpp:4-13), cut the same way, on classes that hold what they read
//     (swapchain extent, TILE_SIZE = 16 as Renderer.h:146); combined as Renderer.cpp:725 (list capacity) and
//     RadixSort.cpp:203-204 (sort bits) combine them;
//   * the reference's two synthetic scenes: TestSortScene::init (Scenes/TestSortScene.cpp:6-35) and
//     SimpleTestGaussiansScene::init (Scenes/SimpleTestGaussiansScene.cpp:5-30), their text on scene classes whose
//     camera records setPosition / setRotation and whose resource manager collects addGaussian; GaussianData{} with
//     its default initialisers comes from the reference's ShaderStructs.h, SMath::PI from the text of SMath.cpp:4-11
//     (3.141592f, not the float nearest to pi); rand() is MSVC's (seed 1: the reference never calls srand), which
//     only colours depend on.
#define GLM_FORCE_RADIANS
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#define GLM_FORCE_QUAT_DATA_WXYZ
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <SMath.h>  // -I /root/reference/vkGaussianSplatting/Engine, pulls <glm/glm.hpp>

struct Window {
    float aspect;
    void getFramebufferSize(int& w, int& h) const { w = 1600; h = 900; }
    float getAspectRatio() const { return aspect; }
};
enum class SphericalHarmonicsMode { ALL_BANDS };
#define GS_REF_CAMERA_CLASS                                                                                           \
    class Camera {                                                                                                    \
    public:                                                                                                           \
        glm::mat4 projectionMatrix, viewMatrix;                                                                       \
        glm::vec3 position, forwardDir, rightDir, upDir;                                                              \
        float yaw, pitch;                                                                                             \
        SphericalHarmonicsMode shMode;                                                                                \
        const Window* window;                                                                                         \
        void updateDirVectors();                                                                                      \
        void updateMatrices();                                                                                        \
        void recalculate();                                                                                           \
        const static float NEAR_PLANE;                                                                                \
        const static float FAR_PLANE;                                                                                 \
    };
namespace cam_as_compiled_here {   // unqualified sin / cos as this compiler resolves them
GS_REF_CAMERA_CLASS
#include "camera_cpp_4_54.inc"
}
namespace cam_float_overloads {    // ... and with the float overloads in scope, as under MSVC
using std::sin;
using std::cos;
GS_REF_CAMERA_CLASS
#include "camera_cpp_4_54.inc"
}

struct VkExtent2D { uint32_t width, height; };
struct Swapchain { VkExtent2D extent; const VkExtent2D& getVkExtent() const { return extent; } };
class Renderer {
public:
    static const uint32_t TILE_SIZE = 16;      // Renderer.h:146
    Swapchain swapchain;
    uint32_t getNumTiles() const;
    uint32_t getCeilPowTwo(uint32_t x) const;
};
#include "renderer_cpp_696_710.inc"
class RadixSort {
public:
    uint32_t getMinNumBits(uint32_t x) const;
};
#include "radixsort_cpp_4_13.inc"

#include <Graphics/ShaderStructs.h>   // GaussianData with its default member initialisers
#include "smath_cpp_4_11.inc"        // const float SMath::PI, SMath::roundToThreeDecimals
namespace scenes {
using Camera = cam_as_compiled_here::Camera;     // Camera::NEAR_PLANE / FAR_PLANE of Camera.cpp:4-5
static uint32_t msvc_rand_state = 1u;
static int rand() { msvc_rand_state = msvc_rand_state * 214013u + 2531011u; return (int)((msvc_rand_state >> 16) & 0x7fffu); }
struct SceneCamera {
    glm::vec3 position; float yaw = 0.0f, pitch = 0.0f;
    void init(const Window&) {}
    void setPosition(const glm::vec3& p) { position = p; }
    void setRotation(float y, float p) { yaw = y; pitch = p; }
    void update() {}
};
struct SceneResources {
    std::vector<GaussianData> gaussians;
    uint32_t addGaussian(const GaussianData& g) { gaussians.push_back(g); return (uint32_t)gaussians.size() - 1u; }
};
#define GS_REF_SCENE_CLASS(NAME)                                                                                      \
    class NAME {                                                                                                      \
    public:                                                                                                           \
        SceneCamera camera; SceneResources resources; Window window{16.0f / 9.0f};                                    \
        const Window& getWindow() const { return window; }                                                            \
        SceneResources& getResourceManager() { return resources; }                                                    \
        void init();                                                                                                  \
    };
GS_REF_SCENE_CLASS(TestSortScene)
#include "testsortscene_cpp_6_35.inc"
GS_REF_SCENE_CLASS(SimpleTestGaussiansScene)
#include "simplescene_cpp_5_30.inc"
}

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

static uint64_t sm_state = 0x1234567ull;
static uint64_t splitmix() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main() {
    struct Cam { const char* name; float px, py, pz, yaw, pitch, aspect; };
    // Scene poses: GardenScene.cpp:11-12, TrainScene.cpp:11-12, BicycleScene.cpp:11-12,
    // TestSortScene.cpp:11-12, SimpleTestGaussiansScene.cpp:11-12; aspects 16:9 and 4:3.
    const Cam cams[] = {
        // the scene files write these as double literals (the Makefile checks the text): converted as the calls convert them
        {"garden", (float)-0.620010, (float)0.189628, (float)2.271181, (float)2.971590, (float)-1.074159, 1920.0f / 1080.0f},
        {"train", (float)-2.857887, (float)0.188856, (float)1.048745, (float)1.361593, (float)0.005841, 1280.0f / 720.0f},
        {"bicycle", (float)0.945927, (float)-0.294418, (float)-0.181088, (float)-1.108407, (float)-0.324159, 1600.0f / 900.0f},
        {"testsort", 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1280.0f / 720.0f},
        {"simple", 0.0f, 0.0f, 2.0f, SMath::PI, 0.0f, 640.0f / 360.0f},     // SimpleTestGaussiansScene.cpp:11-12
        {"fourthirds", 1.5f, -2.25f, 0.75f, 0.7f, 0.3f, 128.0f / 96.0f},
    };
    std::printf("{\n \"cameras\": [\n");
    const int ncam = sizeof(cams) / sizeof(cams[0]);
    for (int c = 0; c < ncam; ++c) {
        const Cam& k = cams[c];
        const Window window{k.aspect};
        cam_as_compiled_here::Camera cam;
        cam.window = &window; cam.position = glm::vec3(k.px, k.py, k.pz); cam.yaw = k.yaw; cam.pitch = k.pitch;
        cam.recalculate();                                            // Camera.cpp:7-54
        cam_float_overloads::Camera camf;
        camf.window = &window; camf.position = cam.position; camf.yaw = k.yaw; camf.pitch = k.pitch;
        camf.recalculate();
        if (std::memcmp(&cam.viewMatrix, &camf.viewMatrix, 64) != 0 || std::memcmp(&cam.projectionMatrix, &camf.projectionMatrix, 64) != 0) {
            std::fprintf(stderr, "Camera.cpp: sin/cos overload resolution changes the matrices of pose %s\n", k.name);
            return 3;
        }
        const glm::mat4 view = cam.viewMatrix, proj = cam.projectionMatrix;
        std::printf("  {\"name\": \"%s\", \"pos\": [%u, %u, %u], \"yaw\": %u, \"pitch\": %u, \"aspect\": %u,\n",
                    k.name, bits(k.px), bits(k.py), bits(k.pz), bits(k.yaw), bits(k.pitch), bits(k.aspect));
        std::printf("   \"view\": [");
        for (int i = 0; i < 16; ++i) std::printf("%u%s", bits(view[i / 4][i % 4]), i < 15 ? ", " : "],\n");
        std::printf("   \"proj\": [");
        for (int i = 0; i < 16; ++i) std::printf("%u%s", bits(proj[i / 4][i % 4]), i < 15 ? ", " : "]}");
        std::printf("%s\n", c + 1 < ncam ? "," : "");
    }
    std::printf(" ],\n \"morton\": [\n");
    const int nm = 256;
    for (int i = 0; i < nm; ++i) {
        uint32_t x, y, z;
        if (i < 8) { x = (i & 1) ? 1023u : 0u; y = (i & 2) ? 1023u : 0u; z = (i & 4) ? 1023u : 0u; }
        else { x = splitmix() % 1024u; y = splitmix() % 1024u; z = splitmix() % 1024u; }
        uint32_t code = SMath::encodeZorderCurve(glm::uvec3(x, y, z));
        std::printf("  [%u, %u, %u, %u]%s\n", x, y, z, code, i + 1 < nm ? "," : "");
    }
    std::printf(" ],\n \"sizes\": [\n");
    // resolutions of the README tables and of the BASELINE configs, splat counts of its scenes and of the configs
    const uint32_t res[][2] = {{640, 360}, {1280, 720}, {1600, 900}, {1920, 1080}, {3840, 2160}, {200, 120}, {1, 1}, {17, 33}};
    const uint32_t counts[] = {1u, 600u, 100000u, 559263u, 1026508u, 4386142u, 5834784u, 50000000u};
    const int nres = sizeof(res) / sizeof(res[0]), ncnt = sizeof(counts) / sizeof(counts[0]);
    for (int r = 0; r < nres; ++r)
        for (int c = 0; c < ncnt; ++c) {
            Renderer renderer;
            renderer.swapchain.extent = VkExtent2D{res[r][0], res[r][1]};
            const uint32_t tiles = renderer.getNumTiles();
            const uint32_t capacity = renderer.getCeilPowTwo(counts[c] + 64 * 16 * tiles);              // Renderer.cpp:725
            const RadixSort sorter{};
            const uint32_t sort_bits = 32 + sorter.getMinNumBits(tiles - 1);                              // RadixSort.cpp:203
            const uint32_t num_sort_bits = uint32_t((sort_bits + 4 - 1) / 4) * 4;                        // RadixSort.cpp:204
            std::printf("  [%u, %u, %u, %u, %u, %u]%s\n", res[r][0], res[r][1], counts[c], tiles, capacity, num_sort_bits,
                        r + 1 < nres || c + 1 < ncnt ? "," : "");
        }
    std::printf(" ],\n \"scenes\": {\n");
    // pose [pos xyz, yaw, pitch] and, per gaussian, position xyz, scale xyzw, rot xyzw, shCoeffs[0] xyzw as float bits;
    // everything else of the 336-byte record must be zero
    auto dump = [](const char* name, const scenes::SceneCamera& cam, const std::vector<GaussianData>& gs, bool last) -> bool {
        std::printf("  \"%s\": {\"pose\": [%u, %u, %u, %u, %u], \"gaussians\": [\n", name, bits(cam.position.x), bits(cam.position.y),
                    bits(cam.position.z), bits(cam.yaw), bits(cam.pitch));
        for (size_t i = 0; i < gs.size(); ++i) {
            const float* f = reinterpret_cast<const float*>(&gs[i]);
            static_assert(sizeof(GaussianData) == 336, "the reference's record");
            for (int k = 16; k < 84; ++k) if (bits(f[k]) != 0u) return false;
            if (bits(f[3]) != 0u) return false;
            std::printf("   [");
            const int idx[15] = {0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
            for (int k = 0; k < 15; ++k) std::printf("%u%s", bits(f[idx[k]]), k < 14 ? ", " : "");
            std::printf("]%s\n", i + 1 < gs.size() ? "," : "");
        }
        std::printf("  ]}%s\n", last ? "" : ",");
        return true;
    };
    {
        scenes::msvc_rand_state = 1u;
        scenes::TestSortScene a; a.init();
        if (!dump("TestSortScene", a.camera, a.resources.gaussians, false)) return 4;
        scenes::msvc_rand_state = 1u;
        scenes::SimpleTestGaussiansScene b; b.init();
        if (!dump("SimpleTestGaussiansScene", b.camera, b.resources.gaussians, true)) return 4;
    }
    std::printf(" }\n}\n");
    return 0;
}
