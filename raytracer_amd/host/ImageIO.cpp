#include "ImageIO.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <vector>

namespace imageio {

namespace {

void putBE32(std::vector<unsigned char> &out, uint32_t v) {
    for (int shift = 24; shift >= 0; shift -= 8) out.push_back(static_cast<unsigned char>(v >> shift));
}

void putChunk(std::vector<unsigned char> &out, const char type[4], const unsigned char *body, size_t len) {
    putBE32(out, static_cast<uint32_t>(len));
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), body, body + len);
    putBE32(out, static_cast<uint32_t>(crc32(0L, &out[start], static_cast<uInt>(4 + len))));
}

bool writeAll(const std::string &path, const void *data, size_t bytes) {
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data, 1, bytes, f) == bytes;
    return std::fclose(f) == 0 && ok;
}

}  // namespace

bool WritePNG(const std::string &path, const uint8_t *rgba8, int width, int height) {
    if (!rgba8 || width <= 0 || height <= 0) return false;
    const size_t stride = static_cast<size_t>(width) * 4;
    std::vector<unsigned char> raw((stride + 1) * height);
    for (int y = 0; y < height; ++y) {
        unsigned char *row = &raw[(stride + 1) * y];
        row[0] = 0;   // filter: none
        std::memcpy(row + 1, rgba8 + stride * (height - 1 - y), stride);
    }
    uLongf packedLen = compressBound(static_cast<uLong>(raw.size()));
    std::vector<unsigned char> packed(packedLen);
    if (compress2(packed.data(), &packedLen, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK) return false;

    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    putBE32(ihdr, static_cast<uint32_t>(width));
    putBE32(ihdr, static_cast<uint32_t>(height));
    const unsigned char tail[5] = {8, 6, 0, 0, 0};   // 8 bits, RGBA, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    putChunk(out, "IHDR", ihdr.data(), ihdr.size());
    putChunk(out, "IDAT", packed.data(), packedLen);
    putChunk(out, "IEND", nullptr, 0);
    return writeAll(path, out.data(), out.size());
}

bool WritePFM(const std::string &path, const float *rgba, int width, int height, float scale) {
    if (!rgba || width <= 0 || height <= 0) return false;
    char header[64];
    const int headerLen = std::snprintf(header, sizeof header, "PF\n%d %d\n-1.0\n", width, height);
    std::vector<unsigned char> out(headerLen + static_cast<size_t>(width) * height * 3 * sizeof(float));
    std::memcpy(out.data(), header, headerLen);
    float *dst = reinterpret_cast<float *>(out.data() + headerLen);
    // The header is not a multiple of 4 bytes in general: fill through memcpy.
    std::vector<float> rgb(static_cast<size_t>(width) * height * 3);
    for (size_t p = 0; p < static_cast<size_t>(width) * height; ++p)
        for (int c = 0; c < 3; ++c) rgb[p * 3 + c] = rgba[p * 4 + c] * scale;
    std::memcpy(static_cast<void *>(dst), rgb.data(), rgb.size() * sizeof(float));
    return writeAll(path, out.data(), out.size());
}

}  // namespace imageio
