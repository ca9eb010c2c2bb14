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
//     RadixSort.cpp:203-204 (sort bits) combine them.
#define GLM_FORCE_RADIANS
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#define GLM_FORCE_QUAT_DATA_WXYZ
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
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
        {"garden", -0.620010f, 0.189628f, 2.271181f, 2.971590f, -1.074159f, 1920.0f / 1080.0f},
        {"train", -2.857887f, 0.188856f, 1.048745f, 1.361593f, 0.005841f, 1280.0f / 720.0f},
        {"bicycle", 0.945927f, -0.294418f, -0.181088f, -1.108407f, -0.324159f, 1600.0f / 900.0f},
        {"testsort", 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1280.0f / 720.0f},
        {"simple", 0.0f, 0.0f, 2.0f, 3.14159265358979323846f, 0.0f, 640.0f / 360.0f},
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
    std::printf(" ]\n}\n");
    return 0;
}
