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
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
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

// What a .ply's first element looks like and where its rows start.
struct PlyHeader {
    std::string format;
    size_t count = 0, row_bytes = 0;
    std::vector<Prop> props;
    std::vector<size_t> offs;          // byte offset of every property inside a binary row
    int col[59];                       // property index of the 59 values the loader reads, in the order of kNames()
    bool ascii = false, big = false;
    std::streampos data_pos;
};

// the 59 named properties of ResourceManager.cpp:186-221
std::vector<std::string> value_names() {
    std::vector<std::string> names = {"x", "y", "z", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1",
                                      "rot_2", "rot_3", "opacity", "f_dc_0", "f_dc_1", "f_dc_2"};
    for (int i = 0; i < 45; ++i) names.push_back("f_rest_" + std::to_string(i));
    return names;
}

int read_header(std::ifstream& in, PlyHeader& h, std::string& err) {
    std::string line;
    if (!std::getline(in, line) || line.substr(0, 3) != "ply") { err = "not a ply file"; return GS_ERR_FORMAT; }
    bool in_first = false, seen_first = false, header_done = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "format") { ss >> h.format; }
        else if (tok == "element") {
            std::string name; size_t cnt = 0;
            ss >> name >> cnt;
            if (!seen_first) { seen_first = true; in_first = true; h.count = cnt; }   // names[0], ResourceManager.cpp:178-179
            else in_first = false;
        } else if (tok == "property" && in_first) {
            Prop p; std::string t; ss >> t;
            if (t == "list") { err = "list property in the gaussian element is not supported"; return GS_ERR_FORMAT; }
            p.type = t; ss >> p.name; p.size = type_size(t); p.is_list = false;
            if (p.size == 0) { err = "unknown property type " + t; return GS_ERR_FORMAT; }
            h.props.push_back(p);
        } else if (tok == "end_header") { header_done = true; break; }
    }
    if (!header_done || !seen_first) { err = "ply header incomplete"; return GS_ERR_FORMAT; }
    h.ascii = h.format == "ascii";
    h.big = h.format == "binary_big_endian";
    if (!h.ascii && !h.big && h.format != "binary_little_endian") { err = "unsupported ply format " + h.format; return GS_ERR_FORMAT; }
    if (h.count > 0x7FFFFFFFull) { err = "too many gaussians"; return GS_ERR_FORMAT; }
    const std::vector<std::string> names = value_names();
    h.offs.resize(h.props.size());
    for (size_t i = 0; i < h.props.size(); ++i) { h.offs[i] = h.row_bytes; h.row_bytes += h.props[i].size; }
    for (size_t k = 0; k < names.size(); ++k) {
        h.col[k] = -1;
        for (size_t i = 0; i < h.props.size(); ++i)
            if (h.props[i].name == names[k]) { h.col[k] = (int)i; break; }
        if (h.col[k] < 0) { err = "ply is missing property " + names[k]; return GS_ERR_FORMAT; }
        if (!is_f32(h.props[h.col[k]].type) && !is_f64(h.props[h.col[k]].type)) {
            err = "property " + names[k] + " is not a float"; return GS_ERR_FORMAT;   // hasPropertyType<float>, ResourceManager.h:67
        }
    }
    // The header's element count is untrusted: before anything of that size is allocated, a binary file must really
    // hold count rows (an ascii row needs at least two bytes per property).
    h.data_pos = in.tellg();
    in.seekg(0, std::ios::end);
    const std::streampos end_pos = in.tellg();
    in.seekg(h.data_pos);
    const uint64_t remaining = end_pos > h.data_pos ? (uint64_t)(end_pos - h.data_pos) : 0ull;
    const uint64_t need = (uint64_t)h.count * (h.ascii ? 2ull * h.props.size() : (uint64_t)h.row_bytes);
    if (!in || need > remaining) { err = "ply data truncated: the header announces more rows than the file holds"; return GS_ERR_FORMAT; }
    return GS_OK;
}

// Walks the rows of the first element once, front to back, in chunks of a few MB (a Garden-size file is 1.4 GB):
// fn(row index, the first `want` of the 59 values).  The converter makes two such sweeps (positions -> Morton order, then
// every row into its slot), so the file must be seekable, and an ASCII file is tokenised twice in full (chunked
// std::from_chars, not iostream extraction); trained models are written binary_little_endian, the format the sweeps are
// built for -- a row is then 59 floats copied out of a 16 MB chunk.
template <typename Fn>
int for_each_row(std::ifstream& in, const PlyHeader& h, int want, std::string& err, Fn&& fn) {
    in.clear();
    in.seekg(h.data_pos);
    float val[59];
    if (h.ascii) {
        // whitespace-separated numbers, any line structure: 16 MB chunks, tokens parsed in place with std::from_chars
        // (a token cut by the end of a chunk is carried over to the next one); a value is the double nearest to its
        // text, narrowed to float -- what `stream >> double` gives, at a fifth of its cost
        auto is_space = [](char ch) { return ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t' || ch == '\v' || ch == '\f'; };
        size_t cap = (size_t)16u << 20;
        if (const char* env = std::getenv("GS_PLY_ASCII_CHUNK")) cap = std::max<size_t>(64, (size_t)std::strtoull(env, nullptr, 10));   // tests: tiny chunks
        std::vector<char> buf(cap);
        std::vector<double> row(h.props.size());
        size_t have = 0, i = 0, p = 0;
        bool eof = false;
        while (i < h.count) {
            if (!eof) {
                in.read(buf.data() + have, (std::streamsize)(cap - have));
                const size_t got = (size_t)in.gcount();
                have += got;
                eof = got == 0;
            }
            size_t end = have;
            if (!eof) while (end > 0 && !is_space(buf[end - 1])) --end;      // the last token may continue in the next chunk
            if (end == 0 && !eof && have == cap) { err = "ply data: a number longer than a whole chunk"; return GS_ERR_FORMAT; }
            const char *c = buf.data(), *e = buf.data() + end;
            while (i < h.count) {
                while (c < e && is_space(*c)) ++c;
                if (c >= e) break;
                const char* t = c;
                while (c < e && !is_space(*c)) ++c;
                if (*t == '+') ++t;                                            // from_chars takes no leading plus
                double d = 0.0;
                const auto res = std::from_chars(t, c, d);
                if (res.ec != std::errc() || res.ptr != c) { err = "ply data: not a number: '" + std::string(t, c).substr(0, 40) + "'"; return GS_ERR_FORMAT; }
                row[p++] = d;
                if (p == h.props.size()) {
                    for (int k = 0; k < want; ++k) val[k] = (float)row[h.col[k]];
                    fn(i, val);
                    p = 0;
                    ++i;
                }
            }
            std::memmove(buf.data(), buf.data() + end, have - end);
            have -= end;
            if (eof && i < h.count && have == 0) { err = "ply data truncated"; return GS_ERR_FORMAT; }
            if (eof && i < h.count && end == 0) { err = "ply data truncated"; return GS_ERR_FORMAT; }
        }
        return GS_OK;
    }
    const size_t chunk_rows = std::max<size_t>(1, (16u << 20) / h.row_bytes);
    std::vector<unsigned char> buf(chunk_rows * h.row_bytes);
    for (size_t i0 = 0; i0 < h.count; i0 += chunk_rows) {
        const size_t rows = std::min(chunk_rows, h.count - i0);
        if (!in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(rows * h.row_bytes))) { err = "ply data truncated"; return GS_ERR_FORMAT; }
        for (size_t r = 0; r < rows; ++r) {
            const unsigned char* row = buf.data() + r * h.row_bytes;
            for (int k = 0; k < want; ++k) {
                const Prop& p = h.props[h.col[k]];
                unsigned char tmp[8];
                std::memcpy(tmp, row + h.offs[h.col[k]], p.size);
                if (h.big) std::reverse(tmp, tmp + p.size);
                if (p.size == 4) { float f; std::memcpy(&f, tmp, 4); val[k] = f; }
                else { double d; std::memcpy(&d, tmp, 8); val[k] = (float)d; }
            }
            fn(i0 + r, val);
        }
    }
    return GS_OK;
}

// ResourceManager.cpp:229-273 for one row: v = the 59 values in value_names() order, g = the 84-float record
inline void convert_row(const float* v, float* g) {
    g[0] = v[0] * -1.0f; g[1] = v[1] * -1.0f; g[2] = v[2]; g[3] = 0.0f;                        // :231-236
    g[4] = std::exp(v[3]); g[5] = std::exp(v[4]); g[6] = std::exp(v[5]); g[7] = 0.0f;          // :237-242
    float r[4] = {v[6], v[7], v[8], v[9]};                                                      // :244-249
    const float t0 = r[0] * r[0], t1 = r[1] * r[1], t2 = r[2] * r[2], t3 = r[3] * r[3];
    const float inv = 1.0f / std::sqrt((t0 + t1) + (t2 + t3));                                 // glm::normalize(vec4), :250
    for (int a = 0; a < 4; ++a) r[a] = r[a] * inv;
    g[8] = -r[2]; g[9] = -r[3]; g[10] = r[0]; g[11] = -r[1];                                   // :251-256
    g[12] = v[11]; g[13] = v[12]; g[14] = v[13];
    g[15] = 1.0f / (1.0f + std::exp(-v[10]));                                                  // :259-264
    for (int c = 0; c < 15; ++c) {                                                             // :265-273
        g[16 + c * 4 + 0] = v[14 + c + 15 * 0];
        g[16 + c * 4 + 1] = v[14 + c + 15 * 1];
        g[16 + c * 4 + 2] = v[14 + c + 15 * 2];
        g[16 + c * 4 + 3] = 0.0f;
    }
    for (int k = 76; k < 84; ++k) g[k] = 0.0f;                                                  // color, covariance: scratch
}

// Two sweeps over the file, nothing but the output held at full size (a Garden-30k .ply is 1.45 GB, its record array
// 1.96 GB; the first version kept 59 column vectors and two record arrays: 5.3 GB).
//   sweep 1: positions only -> bounds, Morton codes (ResourceManager.cpp:225-226, 275-297), the stable order;
//   sweep 2: every row converted (:229-273) straight into its slot of the caller's array.
// records_out == nullptr: only the count.  At most max_records records (the first ones of the Morton order) are written.
int convert(const char* path, float* records_out, size_t max_records, uint32_t& n_out, std::string& err,
            std::vector<float>* grow = nullptr) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = std::string("File cannot be found: ") + (path ? path : "(null)"); return GS_ERR_IO; } // ResourceManager.cpp:169-173
    PlyHeader h;
    if (int rc = read_header(in, h, err)) return rc;
    const size_t n = h.count;
    n_out = (uint32_t)n;
    if (grow) { grow->assign(n * 84, 0.0f); records_out = grow->data(); max_records = n; }
    if (!records_out || n == 0) return GS_OK;

    // sweep 1 (ResourceManager.cpp:223-227, 231-236, 275-281)
    std::vector<float> pos(n * 3);
    float min_pos[3], max_pos[3];
    for (int a = 0; a < 3; ++a) { min_pos[a] = std::numeric_limits<float>::max(); max_pos[a] = std::numeric_limits<float>::min(); } // :225-226 (min() = smallest positive, as written)
    if (int rc = for_each_row(in, h, 3, err, [&](size_t i, const float* v) {
            const float g[3] = {v[0] * -1.0f, v[1] * -1.0f, v[2]};
            for (int a = 0; a < 3; ++a) { pos[i * 3 + a] = g[a]; max_pos[a] = std::max(max_pos[a], g[a]); min_pos[a] = std::min(min_pos[a], g[a]); }
        })) return rc;
    // :283-297 Z-order sort for cache coherence: stable, on precomputed codes (three counting passes over the 30-bit code)
    const float delta[3] = {max_pos[0] - min_pos[0], max_pos[1] - min_pos[1], max_pos[2] - min_pos[2]};
    std::vector<uint32_t> code(n), order(n), tmp(n);
    for (size_t i = 0; i < n; ++i) {
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) q[a] = f2u((pos[i * 3 + a] - min_pos[a]) / delta[a] * 1023.0f);
        code[i] = (morton_part_by2(q[2]) << 2) + (morton_part_by2(q[1]) << 1) + morton_part_by2(q[0]);
    }
    { std::vector<float>().swap(pos); }
    std::iota(order.begin(), order.end(), 0u);
    for (int shift = 0; shift < 32; shift += 11) {                       // f2u saturates, so a code may use all 32 bits
        std::vector<size_t> cnt((size_t)1 << 11, 0);
        for (size_t i = 0; i < n; ++i) ++cnt[(code[order[i]] >> shift) & 2047u];
        size_t run = 0;
        for (size_t b = 0; b < cnt.size(); ++b) { const size_t c = cnt[b]; cnt[b] = run; run += c; }
        for (size_t i = 0; i < n; ++i) tmp[cnt[(code[order[i]] >> shift) & 2047u]++] = order[i];
        order.swap(tmp);
    }
    // order[k] = row that comes k-th; dest[row] = where it goes (tmp reused)
    std::vector<uint32_t>& dest = tmp;
    for (size_t k = 0; k < n; ++k) dest[order[k]] = (uint32_t)k;
    { std::vector<uint32_t>().swap(order); std::vector<uint32_t>().swap(code); }

    // sweep 2
    return for_each_row(in, h, 59, err, [&](size_t i, const float* v) {
        const size_t d = dest[i];
        if (d < max_records) convert_row(v, records_out + d * 84);
    });
}

} // namespace

extern "C" {

// nothing may leave an extern "C" function as an exception (std::bad_alloc / length_error on a hostile header)
static int convert_noexcept(const char* path, float* out, size_t max_records, uint32_t& n, std::vector<float>* grow) {
    try {
        return convert(path, out, max_records, n, g_ply_error, grow);
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
    uint32_t n = 0;
    const int rc = convert_noexcept(path, static_cast<float*>(aos336_out), aos336_out ? max_records : 0, n, nullptr);
    if (rc != GS_OK) return rc;
    *n_out = n;
    return GS_OK;
}

// declared in gs_api.cpp's translation unit via gsplat.h
int gs_load_ply(gs_ctx* ctx, const char* path) {
    if (!ctx || !path) return GS_ERR_INVALID;
    std::vector<float> rec;
    uint32_t n = 0;
    int rc = convert_noexcept(path, nullptr, 0, n, &rec);
    if (rc != GS_OK) return rc;   // message retrievable with gs_ply_last_error()
    if (n == 0) return GS_ERR_FORMAT;
    return gs_upload_gaussians(ctx, rec.data(), n);
}

const char* gs_ply_last_error(void) { return g_ply_error.c_str(); }

} // extern "C"
