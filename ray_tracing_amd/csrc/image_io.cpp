// Image files for the present path.  The reference shows binding 8 by copying it into the
// swapchain image (vulkan.rs:1045-1101); a headless build needs the picture on disk instead.
// Two formats, both written without a compression library:
//   PNG  8 bits per channel from the displayed (gamma-corrected) image, deflate "stored" blocks;
//   EXR  32-bit float scan lines, uncompressed, for the linear accumulator.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"

namespace {

struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t run(uint32_t crc, const uint8_t* p, size_t n) const {
        for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
        return crc;
    }
};

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8) v.push_back(uint8_t(x >> s));
}

void put_chunk(std::vector<uint8_t>& out, const char tag[4], const std::vector<uint8_t>& body) {
    static const Crc32 crc;
    put_be32(out, uint32_t(body.size()));
    size_t start = out.size();
    out.insert(out.end(), tag, tag + 4);
    out.insert(out.end(), body.begin(), body.end());
    put_be32(out, crc.run(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu);
}

// zlib stream of deflate "stored" blocks (RFC 1950 / 1951 section 3.2.4)
std::vector<uint8_t> zlib_stored(const std::vector<uint8_t>& raw) {
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    size_t pos = 0;
    do {
        size_t n = raw.size() - pos;
        if (n > 65535) n = 65535;
        bool last = pos + n == raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(uint8_t(n & 0xFF));
        z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n & 0xFF));
        z.push_back(uint8_t((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    uint32_t a = 1, b = 0;  // Adler-32
    for (size_t i = 0; i < raw.size();) {
        size_t run = raw.size() - i;
        if (run > 5552) run = 5552;
        for (size_t k = 0; k < run; ++k) {
            a += raw[i + k];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
        i += run;
    }
    put_be32(z, (b << 16) | a);
    return z;
}

uint8_t to_byte(float v) {
    if (!(v > 0.0f)) return 0;  // NaN and negatives
    if (v >= 1.0f) return 255;
    return uint8_t(std::lrintf(v * 255.0f));
}

bool write_file(const char* path, const std::vector<uint8_t>& bytes) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

void put_le32(std::vector<uint8_t>& v, uint32_t x) {
    for (int s = 0; s < 32; s += 8) v.push_back(uint8_t(x >> s));
}
void put_le64(std::vector<uint8_t>& v, uint64_t x) {
    for (int s = 0; s < 64; s += 8) v.push_back(uint8_t(x >> s));
}
void put_str(std::vector<uint8_t>& v, const char* s) { v.insert(v.end(), s, s + std::strlen(s) + 1); }
void put_attr(std::vector<uint8_t>& v, const char* name, const char* type, const std::vector<uint8_t>& value) {
    put_str(v, name);
    put_str(v, type);
    put_le32(v, uint32_t(value.size()));
    v.insert(v.end(), value.begin(), value.end());
}
void put_f32(std::vector<uint8_t>& v, float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    put_le32(v, u);
}

}  // namespace

extern "C" {

int rt_save_png(const char* path, const float* rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba || width == 0 || height == 0) return RT_ERR_INVALID;
    std::vector<uint8_t> raw;
    raw.reserve(size_t(height) * (1 + size_t(width) * 3));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);  // filter type None
        const float* row = rgba + size_t(y) * width * 4;
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) raw.push_back(to_byte(row[x * 4 + c]));
    }
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8 bits, RGB, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(out, "IHDR", ihdr);
    std::vector<uint8_t> gama;
    put_be32(gama, 45455);  // the image was encoded with exponent 1/2.2 (shader/src/lib.rs:101-105)
    put_chunk(out, "gAMA", gama);
    put_chunk(out, "IDAT", zlib_stored(raw));
    put_chunk(out, "IEND", {});
    return write_file(path, out) ? RT_OK : RT_ERR_IO;
}

int rt_save_exr(const char* path, const float* rgba, uint32_t width, uint32_t height, float scale) {
    if (!path || !rgba || width == 0 || height == 0) return RT_ERR_INVALID;
    std::vector<uint8_t> out;
    put_le32(out, 20000630u);  // magic
    put_le32(out, 2u);         // version 2, single-part scan-line file
    {
        std::vector<uint8_t> ch;
        for (const char* name : {"B", "G", "R"}) {  // channels are stored in alphabetical order
            put_str(ch, name);
            put_le32(ch, 2u);  // FLOAT
            put_le32(ch, 0u);  // pLinear + reserved
            put_le32(ch, 1u);  // x sampling
            put_le32(ch, 1u);  // y sampling
        }
        ch.push_back(0);
        put_attr(out, "channels", "chlist", ch);
    }
    put_attr(out, "compression", "compression", {0});
    {
        std::vector<uint8_t> box;
        put_le32(box, 0);
        put_le32(box, 0);
        put_le32(box, width - 1);
        put_le32(box, height - 1);
        put_attr(out, "dataWindow", "box2i", box);
        put_attr(out, "displayWindow", "box2i", box);
    }
    put_attr(out, "lineOrder", "lineOrder", {0});
    {
        std::vector<uint8_t> v;
        put_f32(v, 1.0f);
        put_attr(out, "pixelAspectRatio", "float", v);
        std::vector<uint8_t> c;
        put_f32(c, 0.0f);
        put_f32(c, 0.0f);
        put_attr(out, "screenWindowCenter", "v2f", c);
        put_attr(out, "screenWindowWidth", "float", v);
    }
    out.push_back(0);  // end of header
    const uint64_t line_bytes = 8 + uint64_t(width) * 12;
    uint64_t first = out.size() + uint64_t(height) * 8;
    for (uint32_t y = 0; y < height; ++y) put_le64(out, first + y * line_bytes);
    out.reserve(out.size() + size_t(height * line_bytes));
    for (uint32_t y = 0; y < height; ++y) {
        put_le32(out, y);
        put_le32(out, width * 12);
        const float* row = rgba + size_t(y) * width * 4;
        for (int c = 2; c >= 0; --c)  // B, G, R planes of the line
            for (uint32_t x = 0; x < width; ++x) put_f32(out, row[x * 4 + c] * scale);
    }
    return write_file(path, out) ? RT_OK : RT_ERR_IO;
}

}  // extern "C"
