// Minimal PNG encoder for 8-bit RGB images -- the `image::save_buffer(path, data, w, h, Rgb8)` call of the
// reference's save_images (src/bin/sample/main.rs:118-125).  Dependency-free: the zlib stream uses stored
// (uncompressed) deflate blocks, filter type 0 on every scanline.
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "error.hpp"   // sdmi::Error, status codes

namespace sdmi {

static uint32_t crc_table_entry(uint32_t n) {
    uint32_t c = n;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    return c;
}

static uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) table[i] = crc_table_entry(i);
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}

static void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

static void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu);
}

void write_png_rgb8(const std::string& path, const uint8_t* rgb, int width, int height) {
    if (!rgb || width <= 0 || height <= 0) throw Error(SDMI_ERR_INVALID, "write_png: bad image");
    // raw scanlines: filter byte 0 + 3*width bytes
    const size_t stride = (size_t)3 * width;
    std::vector<uint8_t> raw;
    raw.reserve((stride + 1) * height);
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + y * stride, rgb + (y + 1) * stride);
    }
    // zlib container with stored blocks (<= 65535 bytes each) + adler32
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);   // BFINAL, BTYPE = 00
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; ++i) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    }
    put_be32(z, (b << 16) | a);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit, truecolour
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});

    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error(SDMI_ERR_IO, "write_png: cannot open " + path);
    const size_t w = std::fwrite(out.data(), 1, out.size(), f);
    const int rc = std::fclose(f);
    if (w != out.size() || rc != 0) throw Error(SDMI_ERR_IO, "write_png: short write to " + path);
}

}  // namespace sdmi
