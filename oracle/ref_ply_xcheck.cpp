// ref_ply_xcheck.cpp -- CROSS-CHECK of the .ply loader's CONVERSIONS against the reference's own code:
// runs ResourceManager::loadGaussians -- its text, Engine/ResourceManager.cpp:167-300, with the member template
// loadPlyProperty of Engine/ResourceManager.h:59-71, cut out where the files lie under /root/reference by the Makefile,
// never copied into the repository -- over the reference's own vendored happly.h (element / property containers), its
// glm 0.9.9.8, its Engine/SMath.h (encodeZorderCurve) and its Engine/Graphics/ShaderStructs.h (GaussianData), all
// #included unmodified.  Authoring container only; binaries land in the git-ignored oracle/_ref/.
//
// Glue: a class ResourceManager with the two members the function touches (ResourceManager.h itself needs the Vulkan
// wrappers, meshes and textures), Log::error / Log::write, and happly's file constructor: the vendored header only
// DECLARES PLYData(filename) and the parse functions -- their definitions sit in a Windows .lib -- so the constructor
// is defined here and fills the object through happly's own inline addElement / addProperty from a plain column table
// (u32 n, u32 k, k x {u32 len, name}, k x n floats).  What is exercised is therefore everything the reference does
// AFTER parsing -- position sign flips, exp of the scales, normalised and permuted quaternion, sigmoid opacity,
// channel-major f_rest -> per-coefficient RGB, and the Morton order with its `maxPos = numeric_limits<float>::min()`
// start value -- under the same compiler and libm as the product's loader; the PLY syntax itself is covered by the
// three-format test against a numpy restatement (tests/test_library.py::test_ply_conversion).
// The reference orders with an unstable std::sort on the Morton code alone: the scenes of tests/golden/make_ply_xcheck.py
// have no two splats with the same code, so the order is unique.
//
//   ref_ply_xcheck in.tbl out.bin      out: u32 count, then count x 84 floats (the 336-byte records, loaded order)
#define GLM_FORCE_RADIANS
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#define GLM_FORCE_QUAT_DATA_WXYZ
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include <happly.h>                    // -I /root/reference/vkGaussianSplatting/Linking/Include
#include <Graphics/ShaderStructs.h>    // -I /root/reference/vkGaussianSplatting/Engine: GaussianData, SMath.h, glm

// The virtual members of TypedProperty<float> that the vendored header only declares (file parsing and big-endian
// writing, defined in the .lib): needed for the vtable, never called here.
namespace happly {
template <> void TypedProperty<float>::reserve(size_t capacity) { data.reserve(capacity); }
template <> void TypedProperty<float>::parseNext(const std::vector<std::string>&, size_t&) { std::abort(); }
template <> void TypedProperty<float>::readNext(std::istream&) { std::abort(); }
template <> void TypedProperty<float>::readNextBigEndian(std::istream&) { std::abort(); }
template <> void TypedProperty<float>::writeDataBinaryBigEndian(std::ostream&, size_t) { std::abort(); }
}
// stands in for the definition in happlyRelease.lib: same object, filled from the column table
happly::PLYData::PLYData(const std::string& filename, bool) {
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + filename);
    uint32_t n = 0, k = 0;
    if (std::fread(&n, 4, 1, f) != 1 || std::fread(&k, 4, 1, f) != 1) throw std::runtime_error("short table");
    std::vector<std::string> names(k);
    for (auto& name : names) {
        uint32_t len = 0;
        if (std::fread(&len, 4, 1, f) != 1) throw std::runtime_error("short table");
        name.resize(len);
        if (len && std::fread(&name[0], 1, len, f) != len) throw std::runtime_error("short table");
    }
    addElement("vertex", n);
    for (const auto& name : names) {
        std::vector<float> col(n);
        if (n && std::fread(col.data(), 4, n, f) != n) throw std::runtime_error("short table");
        getElement("vertex").addProperty<float>(name, col);
    }
    std::fclose(f);
}

struct Log {
    static void error(const std::string& s) { std::fprintf(stderr, "[Log Error]: %s\n", s.c_str()); }
    static void write(const std::string& s) { std::fprintf(stderr, "[Log]: %s\n", s.c_str()); }
};

class ResourceManager {
public:
    std::vector<GaussianData> gaussians;
    template <typename T>
    void loadPlyProperty(happly::Element& element, const std::string& propertyStr, std::vector<T>& output);
    void loadGaussians(const std::string& filePath);
};

#include "resourcemanager_h_59_71.inc"
#include "resourcemanager_cpp_167_300.inc"

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s in.tbl out.bin\n", argv[0]); return 2; }
    static_assert(sizeof(GaussianData) == 336, "GaussianData is the reference's 336-byte record");
    ResourceManager rm;
    rm.loadGaussians(argv[1]);
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    const uint32_t n = (uint32_t)rm.gaussians.size();
    std::fwrite(&n, 4, 1, o);
    std::fwrite(rm.gaussians.data(), 336, n, o);
    std::fclose(o);
    return 0;
}
