// ref_fixtures.cpp -- driver that runs the REFERENCE's own vendored code (glm 0.9.9.8 and
// Engine/SMath.h, compiled where they lie under /root/reference, never copied) to generate the
// golden vectors committed under tests/golden/.  Built only in the authoring container by
// `make -C oracle ref` into oracle/_ref/ (git-ignored); /root/reference does not exist on the
// GPU box, where only the committed JSON fixtures are used.
//
// What it pins (the only parts of the path that are host C++ compilable without Vulkan/Win32):
//   * camera matrices: glm::lookAt / glm::perspective / glm::normalize / glm::cross with the
//     project's defines GLM_FORCE_RADIANS; GLM_FORCE_DEPTH_ZERO_TO_ONE; GLM_FORCE_QUAT_DATA_WXYZ
//     (vkGaussianSplatting.vcxproj:50), called the way Engine/Graphics/Camera.cpp:7-48 calls them
//     (Camera.cpp itself needs Window/GLFW/pch.h and cannot be compiled here);
//   * Morton codes: SMath::encodeZorderCurve (Engine/SMath.h:24-34).
#define GLM_FORCE_RADIANS
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#define GLM_FORCE_QUAT_DATA_WXYZ
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <SMath.h>  // -I /root/reference/vkGaussianSplatting/Engine, pulls <glm/glm.hpp>

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
        glm::vec3 position(k.px, k.py, k.pz);
        // Camera.cpp:10-16
        glm::vec3 forwardDir((float)(sin((double)k.yaw) * cos((double)k.pitch)),
                             (float)sin((double)k.pitch),
                             (float)(cos((double)k.yaw) * cos((double)k.pitch)));
        forwardDir = glm::normalize(forwardDir);
        // Camera.cpp:34-46
        glm::mat4 view = glm::lookAt(position, position + forwardDir, glm::vec3(0.0f, 1.0f, 0.0f));
        glm::mat4 proj = glm::perspective(glm::radians(90.0f), k.aspect, 0.1f, 100.0f);
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
    std::printf(" ]\n}\n");
    return 0;
}
