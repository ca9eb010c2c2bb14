// gs_ply.cpp -- .ply -> 336-byte GaussianData records, the input side of the hot path
// (ResourceManager::loadGaussians, Engine/ResourceManager.cpp:167-300).
//
// The reference parses with happly (nmwsharp/happly, version unpinned; only declarations are
// vendored, the bodies live in prebuilt MSVC .libs), so the parser here is our own reader of the
// public PLY format: ascii / binary_little_endian / binary_big_endian 1.0, scalar properties of
// the FIRST element looked up by name, float32 (or float64, narrowed) values.
// The conversions follow ResourceManager.cpp:229-297 line by line; one documented deviation:
// the Morton re-ordering uses a STABLE sort on precomputed codes instead of std::sort with an
// on-the-fly comparator (only changes which of two equal-code splats comes first).
#include "../../include/gsplat.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <limits>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

namespace {

thread_local std::string g_ply_error;

struct Prop { std::string name; std::string type; size_t size; bool is_list; };

size_t type_size(const std::string& t) {
    if (t == "char" || t == "int8" || t == "uchar" || t == "uint8") return 1;
    if (t == "short" || t == "int16" || t == "ushort" || t == "uint16") return 2;
    if (t == "int" || t == "int32" || t == "uint" || t == "uint32" || t == "float" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
bool is_f32(const std::string& t) { return t == "float" || t == "float32"; }
bool is_f64(const std::string& t) { return t == "double" || t == "float64"; }

// SMath.h:10-18
uint32_t morton_part_by2(uint32_t x) {
    x &= 0x000003ffu;
    x = (x ^ (x << 16)) & 0xff0000ffu;
    x = (x ^ (x << 8)) & 0x0300f00fu;
    x = (x ^ (x << 4)) & 0x030c30c3u;
    x = (x ^ (x << 2)) & 0x09249249u;
    return x;
}
uint32_t f2u(float v) {   // glm::uvec3(vec3): truncation; out-of-range pinned to saturate, NaN -> 0
    if (!(v == v) || v <= 0.0f) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

int convert(const char* path, std::vector<float>& records, uint32_t& n_out, std::string& err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = std::string("File cannot be found: ") + (path ? path : "(null)"); return GS_ERR_IO; } // ResourceManager.cpp:169-173
    std::string line;
    if (!std::getline(in, line) || line.substr(0, 3) != "ply") { err = "not a ply file"; return GS_ERR_FORMAT; }
    std::string format;
    size_t count = 0;
    bool in_first = false, seen_first = false, header_done = false;
    std::vector<Prop> props;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "format") { ss >> format; }
        else if (tok == "element") {
            std::string name; size_t cnt = 0;
            ss >> name >> cnt;
            if (!seen_first) { seen_first = true; in_first = true; count = cnt; }   // names[0], ResourceManager.cpp:178-179
            else in_first = false;
        } else if (tok == "property" && in_first) {
            Prop p; std::string t; ss >> t;
            if (t == "list") { err = "list property in the gaussian element is not supported"; return GS_ERR_FORMAT; }
            p.type = t; ss >> p.name; p.size = type_size(t); p.is_list = false;
            if (p.size == 0) { err = "unknown property type " + t; return GS_ERR_FORMAT; }
            props.push_back(p);
        } else if (tok == "end_header") { header_done = true; break; }
    }
    if (!header_done || !seen_first) { err = "ply header incomplete"; return GS_ERR_FORMAT; }
    const bool ascii = format == "ascii";
    const bool big = format == "binary_big_endian";
    if (!ascii && !big && format != "binary_little_endian") { err = "unsupported ply format " + format; return GS_ERR_FORMAT; }
    if (count > 0x7FFFFFFFull) { err = "too many gaussians"; return GS_ERR_FORMAT; }

    // the 62 named properties of ResourceManager.cpp:186-221
    std::vector<std::string> names = {"x", "y", "z", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1",
                                      "rot_2", "rot_3", "opacity", "f_dc_0", "f_dc_1", "f_dc_2"};
    for (int i = 0; i < 45; ++i) names.push_back("f_rest_" + std::to_string(i));
    std::vector<int> col(names.size(), -1);
    size_t row_bytes = 0;
    std::vector<size_t> offs(props.size());
    for (size_t i = 0; i < props.size(); ++i) { offs[i] = row_bytes; row_bytes += props[i].size; }
    for (size_t k = 0; k < names.size(); ++k) {
        for (size_t i = 0; i < props.size(); ++i)
            if (props[i].name == names[k]) { col[k] = (int)i; break; }
        if (col[k] < 0) { err = "ply is missing property " + names[k]; return GS_ERR_FORMAT; }
        if (!is_f32(props[col[k]].type) && !is_f64(props[col[k]].type)) {
            err = "property " + names[k] + " is not a float"; return GS_ERR_FORMAT;   // hasPropertyType<float>, ResourceManager.h:67
        }
    }

    const size_t n = count;
    // The header's element count is untrusted: before anything of that size is allocated, a binary file must really
    // hold count rows (an ascii row needs at least two bytes per property).
    {
        const std::streampos data_pos = in.tellg();
        in.seekg(0, std::ios::end);
        const std::streampos end_pos = in.tellg();
        in.seekg(data_pos);
        const uint64_t remaining = end_pos > data_pos ? (uint64_t)(end_pos - data_pos) : 0ull;
        const uint64_t need = (uint64_t)n * (ascii ? 2ull * props.size() : (uint64_t)row_bytes);
        if (!in || need > remaining) { err = "ply data truncated: the header announces more rows than the file holds"; return GS_ERR_FORMAT; }
    }
    std::vector<std::vector<float>> v(names.size(), std::vector<float>(n));
    if (ascii) {
        std::vector<double> row(props.size());
        for (size_t i = 0; i < n; ++i) {
            for (size_t p = 0; p < props.size(); ++p)
                if (!(in >> row[p])) { err = "ply data truncated"; return GS_ERR_FORMAT; }
            for (size_t k = 0; k < names.size(); ++k) v[k][i] = (float)row[col[k]];
        }
    } else {
        std::vector<unsigned char> buf(row_bytes);
        for (size_t i = 0; i < n; ++i) {
            if (!in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)row_bytes)) { err = "ply data truncated"; return GS_ERR_FORMAT; }
            for (size_t k = 0; k < names.size(); ++k) {
                const Prop& p = props[col[k]];
                unsigned char tmp[8];
                std::memcpy(tmp, buf.data() + offs[col[k]], p.size);
                if (big) std::reverse(tmp, tmp + p.size);
                if (p.size == 4) { float f; std::memcpy(&f, tmp, 4); v[k][i] = f; }
                else { double d; std::memcpy(&d, tmp, 8); v[k][i] = (float)d; }
            }
        }
    }

    // ResourceManager.cpp:223-282
    std::vector<float> rec(n * 84, 0.0f);
    float min_pos[3], max_pos[3];
    for (int a = 0; a < 3; ++a) { min_pos[a] = std::numeric_limits<float>::max(); max_pos[a] = std::numeric_limits<float>::min(); } // :225-226 (min() = smallest positive, as written)
    for (size_t i = 0; i < n; ++i) {
        float* g = &rec[i * 84];
        g[0] = v[0][i] * -1.0f; g[1] = v[1][i] * -1.0f; g[2] = v[2][i]; g[3] = 0.0f;          // :231-236
        g[4] = std::exp(v[3][i]); g[5] = std::exp(v[4][i]); g[6] = std::exp(v[5][i]); g[7] = 0.0f; // :237-242
        float r[4] = {v[6][i], v[7][i], v[8][i], v[9][i]};                                       // :244-249
        const float t0 = r[0] * r[0], t1 = r[1] * r[1], t2 = r[2] * r[2], t3 = r[3] * r[3];
        const float inv = 1.0f / std::sqrt((t0 + t1) + (t2 + t3));                              // glm::normalize(vec4), :250
        for (int a = 0; a < 4; ++a) r[a] = r[a] * inv;
        g[8] = -r[2]; g[9] = -r[3]; g[10] = r[0]; g[11] = -r[1];                                // :251-256
        g[12] = v[11][i]; g[13] = v[12][i]; g[14] = v[13][i];
        g[15] = 1.0f / (1.0f + std::exp(-v[10][i]));                                            // :259-264
        for (int c = 0; c < 15; ++c) {                                                          // :265-273
            g[16 + c * 4 + 0] = v[14 + c + 15 * 0][i];
            g[16 + c * 4 + 1] = v[14 + c + 15 * 1][i];
            g[16 + c * 4 + 2] = v[14 + c + 15 * 2][i];
            g[16 + c * 4 + 3] = 0.0f;
        }
        for (int a = 0; a < 3; ++a) { max_pos[a] = std::max(max_pos[a], g[a]); min_pos[a] = std::min(min_pos[a], g[a]); } // :275-281
    }
    // :283-297 Z-order sort for cache coherence
    const float delta[3] = {max_pos[0] - min_pos[0], max_pos[1] - min_pos[1], max_pos[2] - min_pos[2]};
    std::vector<uint32_t> code(n);
    for (size_t i = 0; i < n; ++i) {
        const float* g = &rec[i * 84];
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) q[a] = f2u((g[a] - min_pos[a]) / delta[a] * 1023.0f);
        code[i] = (morton_part_by2(q[2]) << 2) + (morton_part_by2(q[1]) << 1) + morton_part_by2(q[0]);
    }
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    records.resize(n * 84);
    for (size_t i = 0; i < n; ++i) std::memcpy(&records[i * 84], &rec[(size_t)order[i] * 84], 84 * sizeof(float));
    n_out = (uint32_t)n;
    return GS_OK;
}

} // namespace

extern "C" {

// nothing may leave an extern "C" function as an exception (std::bad_alloc / length_error on a hostile header)
static int convert_noexcept(const char* path, std::vector<float>& rec, uint32_t& n) {
    try {
        return convert(path, rec, n, g_ply_error);
    } catch (const std::exception& ex) {
        g_ply_error = std::string("ply conversion failed: ") + ex.what();
        return GS_ERR_FORMAT;
    } catch (...) {
        g_ply_error = "ply conversion failed";
        return GS_ERR_FORMAT;
    }
}

int gs_convert_ply(const char* path, void* aos336_out, uint32_t max_records, uint32_t* n_out) {
    if (!path || !n_out) return GS_ERR_INVALID;
    std::vector<float> rec;
    uint32_t n = 0;
    int rc = convert_noexcept(path, rec, n);
    if (rc != GS_OK) return rc;
    *n_out = n;
    if (aos336_out) {
        const uint32_t m = n < max_records ? n : max_records;
        if (m) std::memcpy(aos336_out, rec.data(), (size_t)m * GS_GAUSSIAN_RECORD_BYTES);
    }
    return GS_OK;
}

// declared in gs_api.cpp's translation unit via gsplat.h
int gs_load_ply(gs_ctx* ctx, const char* path) {
    if (!ctx || !path) return GS_ERR_INVALID;
    std::vector<float> rec;
    uint32_t n = 0;
    int rc = convert_noexcept(path, rec, n);
    if (rc != GS_OK) return rc;   // message retrievable with gs_ply_last_error()
    if (n == 0) return GS_ERR_FORMAT;
    return gs_upload_gaussians(ctx, rec.data(), n);
}

const char* gs_ply_last_error(void) { return g_ply_error.c_str(); }

} // extern "C"
