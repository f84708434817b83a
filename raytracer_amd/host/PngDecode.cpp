// PNG -> RGBA8: every colour type (grey, RGB, palette, grey+alpha, RGBA), bit depths 1, 2, 4, 8
// and 16, plain and Adam7-interlaced, tRNS transparency (palette alpha, grey and RGB colour keys).
// The reference gets texture images decoded to 8-bit RGBA by stb_image through tinygltf (reference
// src/MeshData.cpp:419-425, 497-528); samples below 8 bits are scaled to the full range (x255, x85,
// x17) and 16-bit samples keep their high byte, as that decoder's 8-bit interface does.  (For a
// 16-bit file the reference's own loader ends up with a texture without data, because it only
// converts byte and float pixels, src/MeshData.cpp:497-556; here such a file is usable.)
#include "ImageIO.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <vector>

namespace imageio {
namespace {

uint32_t be32(const unsigned char *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

// Undoes the per-row prediction filters of one (sub)image in place: rows of 1 filter byte +
// rowBytes, bpp = bytes per whole pixel (at least 1).  False on an unknown filter type.
bool unfilter(unsigned char *rows, size_t rowBytes, int height, size_t bpp) {
    for (int y = 0; y < height; ++y) {
        unsigned char *row = rows + (rowBytes + 1) * static_cast<size_t>(y);
        const unsigned char filter = row[0];
        unsigned char *cur = row + 1;
        const unsigned char *up = y > 0 ? cur - (rowBytes + 1) : nullptr;
        if (filter > 4) return false;
        for (size_t i = 0; i < rowBytes; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0;
            const int b = up ? up[i] : 0;
            const int c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: {
                    const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: break;
            }
            cur[i] = static_cast<unsigned char>(cur[i] + pred);
        }
    }
    return true;
}

struct Header {
    int width = 0, height = 0, bitDepth = 0, colorType = 0, interlace = 0, channels = 0;
};

// Sample `index` (counted in samples, not bytes) of an unfiltered row, at the file's bit depth.
inline unsigned sampleAt(const unsigned char *row, size_t index, int bitDepth) {
    switch (bitDepth) {
        case 16: return (unsigned(row[index * 2]) << 8) | row[index * 2 + 1];
        case 8: return row[index];
        default: {
            const size_t bit = index * static_cast<size_t>(bitDepth);
            return (row[bit >> 3] >> (8 - bitDepth - static_cast<int>(bit & 7))) & ((1u << bitDepth) - 1u);
        }
    }
}

struct Transparency {
    std::vector<unsigned char> paletteAlpha;
    bool hasKey = false;
    unsigned key[3] = {0, 0, 0};
};

// One pixel of an unfiltered row -> RGBA8.
void convertPixel(const Header &h, const unsigned char *row, size_t x, const std::vector<unsigned char> &palette, const Transparency &tr, unsigned char *out) {
    unsigned s[4] = {0, 0, 0, 0};
    for (int c = 0; c < h.channels; ++c) s[c] = sampleAt(row, x * static_cast<size_t>(h.channels) + static_cast<size_t>(c), h.bitDepth);
    if (h.colorType == 3) {
        const size_t k = s[0];
        const bool known = k * 3 + 2 < palette.size();
        out[0] = known ? palette[k * 3] : 0;
        out[1] = known ? palette[k * 3 + 1] : 0;
        out[2] = known ? palette[k * 3 + 2] : 0;
        out[3] = k < tr.paletteAlpha.size() ? tr.paletteAlpha[k] : 255;
        return;
    }
    // 8-bit value of a sample: low depths spread over the range, 16-bit keeps the high byte
    const unsigned scale = h.bitDepth == 1 ? 255u : h.bitDepth == 2 ? 85u : h.bitDepth == 4 ? 17u : 1u;
    auto to8 = [&](unsigned v) { return static_cast<unsigned char>(h.bitDepth == 16 ? v >> 8 : v * scale); };
    switch (h.colorType) {
        case 0:
            out[0] = out[1] = out[2] = to8(s[0]);
            out[3] = (tr.hasKey && s[0] == tr.key[0]) ? 0 : 255;
            break;
        case 2:
            out[0] = to8(s[0]); out[1] = to8(s[1]); out[2] = to8(s[2]);
            out[3] = (tr.hasKey && s[0] == tr.key[0] && s[1] == tr.key[1] && s[2] == tr.key[2]) ? 0 : 255;
            break;
        case 4:
            out[0] = out[1] = out[2] = to8(s[0]);
            out[3] = to8(s[1]);
            break;
        default:
            out[0] = to8(s[0]); out[1] = to8(s[1]); out[2] = to8(s[2]); out[3] = to8(s[3]);
            break;
    }
}

}  // namespace

bool DecodePngRGBA(const unsigned char *data, size_t size, int &width, int &height, std::vector<unsigned char> &rgba, std::string &why) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || std::memcmp(data, sig, 8) != 0) { why = "not a PNG stream"; return false; }
    size_t at = 8;
    Header h;
    bool haveHeader = false;
    std::vector<unsigned char> idat, palette, trns;
    while (at + 12 <= size) {
        const uint32_t len = be32(data + at);
        const unsigned char *type = data + at + 4, *body = data + at + 8;
        if (len > size - at - 12) { why = "truncated chunk"; return false; }
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            h.width = static_cast<int>(be32(body));
            h.height = static_cast<int>(be32(body + 4));
            h.bitDepth = body[8];
            h.colorType = body[9];
            h.interlace = body[12];
            if (be32(body) > 16384u || be32(body + 4) > 16384u) { why = "image larger than 16384 x 16384"; return false; }
            if (body[10] != 0 || body[11] != 0) { why = "unknown compression or filter method"; return false; }
            haveHeader = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        at += 12 + static_cast<size_t>(len);
    }
    if (!haveHeader || h.width <= 0 || h.height <= 0) { why = "no image header"; return false; }
    h.channels = h.colorType == 0 ? 1 : h.colorType == 2 ? 3 : h.colorType == 3 ? 1 : h.colorType == 4 ? 2 : h.colorType == 6 ? 4 : 0;
    const int d = h.bitDepth;
    const bool depthAllowed = h.colorType == 0   ? (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)
                              : h.colorType == 3 ? (d == 1 || d == 2 || d == 4 || d == 8)
                                                 : (d == 8 || d == 16);
    if (h.channels == 0 || !depthAllowed || h.interlace > 1) { why = "colour type, bit depth or interlace method outside the PNG standard"; return false; }

    Transparency tr;
    if (h.colorType == 3) {
        tr.paletteAlpha = trns;
    } else if (h.colorType == 0 && trns.size() >= 2) {
        tr.hasKey = true;
        tr.key[0] = (unsigned(trns[0]) << 8) | trns[1];
    } else if (h.colorType == 2 && trns.size() >= 6) {
        tr.hasKey = true;
        for (int c = 0; c < 3; ++c) tr.key[c] = (unsigned(trns[2 * c]) << 8) | trns[2 * c + 1];
    }

    // The seven Adam7 passes: origin and step of each pass's pixel lattice; a plain file is one pass over everything.
    static const int passX0[7] = {0, 4, 0, 2, 0, 1, 0}, passY0[7] = {0, 0, 4, 0, 2, 0, 1};
    static const int passDX[7] = {8, 8, 4, 4, 2, 2, 1}, passDY[7] = {8, 8, 8, 4, 4, 2, 2};
    const int passes = h.interlace ? 7 : 1;
    const size_t bitsPerPixel = static_cast<size_t>(h.channels) * static_cast<size_t>(d);
    const size_t bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
    struct Pass { int w, h, x0, y0, dx, dy; size_t rowBytes, offset; };
    Pass pass[7];
    size_t total = 0;
    for (int p = 0; p < passes; ++p) {
        Pass &q = pass[p];
        q.x0 = h.interlace ? passX0[p] : 0;
        q.y0 = h.interlace ? passY0[p] : 0;
        q.dx = h.interlace ? passDX[p] : 1;
        q.dy = h.interlace ? passDY[p] : 1;
        q.w = (h.width - q.x0 + q.dx - 1) / q.dx;
        q.h = (h.height - q.y0 + q.dy - 1) / q.dy;
        if (q.w <= 0 || q.h <= 0) q.w = q.h = 0;   // an empty pass has no bytes at all, not even filter bytes
        q.rowBytes = (static_cast<size_t>(q.w) * bitsPerPixel + 7) / 8;
        q.offset = total;
        total += (q.rowBytes + 1) * static_cast<size_t>(q.h);
    }
    // Deflate cannot expand beyond 1032 : 1, so a header that promises more than the data present could
    // inflate to is refused before anything of that size is allocated (a few header bytes must not cost gigabytes).
    if (total > (idat.size() + 16) * 1032) { why = "image data too short for the size the header gives"; return false; }
    std::vector<unsigned char> raw(total);
    uLongf rawLen = static_cast<uLongf>(raw.size());
    if (uncompress(raw.data(), &rawLen, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || rawLen != raw.size()) {
        why = "image data does not inflate to the size the header gives";
        return false;
    }
    rgba.assign(static_cast<size_t>(h.width) * h.height * 4, 0);
    for (int p = 0; p < passes; ++p) {
        const Pass &q = pass[p];
        if (q.h == 0) continue;
        if (!unfilter(raw.data() + q.offset, q.rowBytes, q.h, bpp)) { why = "unknown row filter"; return false; }
        for (int y = 0; y < q.h; ++y) {
            const unsigned char *row = raw.data() + q.offset + (q.rowBytes + 1) * static_cast<size_t>(y) + 1;
            unsigned char *outRow = &rgba[static_cast<size_t>(q.y0 + y * q.dy) * h.width * 4];
            for (int x = 0; x < q.w; ++x) convertPixel(h, row, static_cast<size_t>(x), palette, tr, outRow + static_cast<size_t>(q.x0 + x * q.dx) * 4);
        }
    }
    width = h.width;
    height = h.height;
    return true;
}

}  // namespace imageio
