// gs_image.cpp -- frame output sinks: the reference's frame ends in a swapchain image
// (imageStore in RenderGaussians.comp:150, presented by Renderer.cpp:341-397); a host that has no window
// writes the RGBA8 frame to disk instead.  Binary PPM (P6, alpha dropped) or PNG (8-bit RGBA, zlib stream
// of stored blocks -- no compression library needed, every decoder reads it).
#include "../../include/gsplat.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

uint32_t crc_table[256];
bool crc_ready = false;

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0xFFFFFFFFu, out.data() + at, out.size() - at) ^ 0xFFFFFFFFu);
}

bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i) {
        char a = s[s.size() - n + i], b = suffix[i];
        if (a >= 'A' && a <= 'Z') a = char(a - 'A' + 'a');
        if (a != b) return false;
    }
    return true;
}

int write_all(const char* path, const uint8_t* p, size_t n) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return GS_ERR_IO;
    const size_t w = std::fwrite(p, 1, n, f);
    const int rc = std::fclose(f);
    return (w == n && rc == 0) ? GS_OK : GS_ERR_IO;
}

} // namespace

extern "C" int gs_write_image(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba || width == 0 || height == 0) return GS_ERR_INVALID;
    const std::string name(path);
    const size_t npx = (size_t)width * height;
    if (ends_with(name, ".ppm")) {
        char head[64];
        const int hl = std::snprintf(head, sizeof head, "P6\n%u %u\n255\n", width, height);
        std::vector<uint8_t> out((size_t)hl + npx * 3);
        std::memcpy(out.data(), head, (size_t)hl);
        uint8_t* d = out.data() + hl;
        for (size_t i = 0; i < npx; ++i) {
            d[3 * i + 0] = rgba[4 * i + 0];
            d[3 * i + 1] = rgba[4 * i + 1];
            d[3 * i + 2] = rgba[4 * i + 2];
        }
        return write_all(path, out.data(), out.size());
    }
    if (!ends_with(name, ".png")) return GS_ERR_INVALID;

    // raw scanlines: filter byte 0 + RGBA
    const size_t stride = (size_t)width * 4 + 1;
    std::vector<uint8_t> raw(stride * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw[y * stride] = 0;
        std::memcpy(&raw[y * stride + 1], rgba + (size_t)y * width * 4, (size_t)width * 4);
    }
    // zlib stream: header 78 01, stored deflate blocks of <= 65535 bytes, adler32
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t len = raw.size() - off < 65535 ? raw.size() - off : 65535;
        const bool last = off + len == raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(uint8_t(len)); z.push_back(uint8_t(len >> 8));
        z.push_back(uint8_t(~len)); z.push_back(uint8_t((~len) >> 8));
        z.insert(z.end(), raw.begin() + (ptrdiff_t)off, raw.begin() + (ptrdiff_t)(off + len));
        for (size_t i = 0; i < len;) {                              // adler32, 5552-byte runs between the modulos
            const size_t run = len - i < 5552 ? len - i : 5552;
            for (size_t k = 0; k < run; ++k) { a += raw[off + i + k]; b += a; }
            a %= 65521u; b %= 65521u;
            i += run;
        }
        off += len;
    }
    put_be32(z, (b << 16) | a);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGBA
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    return write_all(path, out.data(), out.size());
}
