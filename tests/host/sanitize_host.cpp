// Host-side code of the library (PLY reader + ResourceManager conversions, frame sinks, the band-cutting rule) built with
// -fsanitize=address,undefined and fed valid, truncated and corrupted inputs.  GPU sanitizers are not
// available on the pool, so this is where the parsers get their memory checking.  The device upload is
// replaced by a counter: nothing here touches HIP.
#include "../../include/gsplat.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <vector>

static uint32_t g_uploaded = 0;
extern "C" int gs_upload_gaussians(gs_ctx*, const void* aos, uint32_t n) {
    // touch every byte so that an undersized buffer is caught
    const unsigned char* p = static_cast<const unsigned char*>(aos);
    unsigned acc = 0;
    for (size_t i = 0; i < (size_t)n * GS_GAUSSIAN_RECORD_BYTES; ++i) acc += p[i];
    g_uploaded = n + (acc & 0u);
    return GS_OK;
}

static const char* kProps[] = {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"};

static std::string make_ply(uint32_t n, const char* format, uint64_t seed) {
    std::string s = "ply\nformat " + std::string(format) + " 1.0\nelement vertex " + std::to_string(n) + "\n";
    std::vector<std::string> names(kProps, kProps + 9);
    for (int i = 0; i < 45; ++i) names.push_back("f_rest_" + std::to_string(i));
    names.push_back("opacity");
    for (int i = 0; i < 3; ++i) names.push_back("scale_" + std::to_string(i));
    for (int i = 0; i < 4; ++i) names.push_back("rot_" + std::to_string(i));
    for (auto& nm : names) s += "property float " + nm + "\n";
    s += "end_header\n";
    const bool ascii = std::strcmp(format, "ascii") == 0;
    for (uint32_t i = 0; i < n; ++i)
        for (size_t k = 0; k < names.size(); ++k) {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            const float v = (float)((int64_t)(seed >> 40) - (1 << 23)) / (float)(1 << 21);
            if (ascii) { s += std::to_string(v); s += (k + 1 == names.size()) ? "\n" : " "; }
            else {
                unsigned char b[4]; std::memcpy(b, &v, 4);
                if (std::strcmp(format, "binary_big_endian") == 0) { std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
                s.append(reinterpret_cast<char*>(b), 4);
            }
        }
    return s;
}

static void write_file(const std::string& path, const std::string& data) {
    std::ofstream f(path, std::ios::binary);
    f.write(data.data(), (std::streamsize)data.size());
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const std::string path = dir + "/san.ply";
    int checked = 0;
    for (const char* fmt : {"binary_little_endian", "binary_big_endian", "ascii"}) {
        const std::string good = make_ply(37, fmt, 99);
        write_file(path, good);
        uint32_t n = 0;
        if (gs_convert_ply(path.c_str(), nullptr, 0, &n) != GS_OK || n != 37) { std::printf("query failed: %s\n", gs_ply_last_error()); return 1; }
        std::vector<unsigned char> rec((size_t)n * GS_GAUSSIAN_RECORD_BYTES);
        if (gs_convert_ply(path.c_str(), rec.data(), n, &n) != GS_OK) { std::printf("convert failed\n"); return 1; }
        // a caller buffer smaller than the file: must not write past max_records
        std::vector<unsigned char> small((size_t)5 * GS_GAUSSIAN_RECORD_BYTES);
        (void)gs_convert_ply(path.c_str(), small.data(), 5, &n);
        if (gs_load_ply(reinterpret_cast<gs_ctx*>(&g_uploaded), path.c_str()) != GS_OK || g_uploaded != 37) { std::printf("load failed\n"); return 1; }
        // truncations at every 97th byte and a few single-byte corruptions of the header
        for (size_t cut = 0; cut < good.size(); cut += 97) {
            write_file(path, good.substr(0, cut));
            (void)gs_convert_ply(path.c_str(), rec.data(), 37, &n);
            ++checked;
        }
        for (size_t pos = 0; pos < 200 && pos < good.size(); pos += 7) {
            std::string bad = good;
            bad[pos] = (char)(bad[pos] ^ 0x5A);
            write_file(path, bad);
            (void)gs_convert_ply(path.c_str(), rec.data(), 37, &n);
            ++checked;
        }
    }
    // absurd vertex counts / property lists
    write_file(path, "ply\nformat binary_little_endian 1.0\nelement vertex 4000000000\nproperty float x\nend_header\n");
    uint32_t n = 0;
    (void)gs_convert_ply(path.c_str(), nullptr, 0, &n);
    write_file(path, "ply\nformat ascii 1.0\nelement vertex -3\nproperty list uchar int vertex_indices\nend_header\n");
    (void)gs_convert_ply(path.c_str(), nullptr, 0, &n);
    write_file(path, "");
    (void)gs_convert_ply(path.c_str(), nullptr, 0, &n);

    // frame sinks
    std::vector<uint8_t> img((size_t)333 * 77 * 4);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)(i * 131u);
    if (gs_write_image((dir + "/f.png").c_str(), img.data(), 333, 77) != GS_OK) return 1;
    if (gs_write_image((dir + "/f.ppm").c_str(), img.data(), 333, 77) != GS_OK) return 1;
    if (gs_write_image((dir + "/big.png").c_str(), img.data(), 1, 77 * 333) != GS_OK) return 1;
    if (gs_write_image((dir + "/f.bmp").c_str(), img.data(), 333, 77) != GS_ERR_INVALID) return 1;
    if (gs_write_image(nullptr, img.data(), 1, 1) != GS_ERR_INVALID) return 1;
    // the band-cutting rule of a sharded frame (gs_balance.cpp: gs_dist_shard_rows / gs_dist_rebalance / gs_balance_rows): random,
    // degenerate and hostile weight vectors -- the edges must always be a partition of the rows, whatever the weights are
    {
        uint64_t s = 12345;
        auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
        int cuts = 0;
        for (int t = 0; t < 20000; ++t) {
            const uint32_t ty = rnd() % 300u, world = 1u + rnd() % 70u;
            std::vector<double> w(ty);
            for (double& x : w) {
                switch (t % 6) {
                    case 0: x = (double)(rnd() % 500000u); break;
                    case 1: x = 0.0; break;
                    case 2: x = (rnd() % 10u) ? 1.0 : 1e300; break;                                  // the sum overflows to infinity
                    case 3: x = (rnd() % 7u) ? (double)(rnd() % 100u) : std::numeric_limits<double>::quiet_NaN(); break;
                    case 4: x = -(double)(rnd() % 100u); break;                                     // negative weights: treated as equal rows
                    default: x = 1e-300 * (double)(rnd() % 3u); break;
                }
            }
            std::vector<uint32_t> e(world + 1u, 0xDEADBEEFu);
            if (gs_balance_rows(ty ? w.data() : nullptr, ty, world, e.data()) != GS_OK) { std::printf("gs_balance_rows failed\n"); return 1; }
            if (e[0] != 0u || e[world] != ty) { std::printf("edges do not span the rows (ty %u world %u)\n", ty, world); return 1; }
            for (uint32_t r = 0; r < world; ++r) {
                if (e[r] > e[r + 1u]) { std::printf("edges not monotone\n"); return 1; }
                if (ty >= world && e[r] == e[r + 1u]) { std::printf("a rank idles although there are rows to give (ty %u world %u)\n", ty, world); return 1; }
            }
            ++cuts;
        }
        uint32_t e2[3];
        if (gs_balance_rows(nullptr, 4, 2, e2) != GS_ERR_INVALID || gs_balance_rows(nullptr, 0, 0, e2) != GS_ERR_INVALID) return 1;
        std::printf("%d band cuts checked\n", cuts);
    }
    std::printf("sanitize_host ok: %d malformed inputs survived\n", checked);
    return 0;
}
