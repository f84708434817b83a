// Baseline / extended-sequential JPEG (Huffman, 8-bit, 1 or 3 components, chroma subsampled
// by 1 or 2 in each direction, restart intervals) -> RGBA8.  The reference decodes textures
// with stb_image through tinygltf (reference src/MeshData.cpp:419-425, 558-580), which is not
// available; this decoder follows the same family of algorithms (integer IDCT of the IJG slow
// kind, triangle-filter chroma upsampling, fixed-point YCbCr conversion).  Progressive,
// arithmetic-coded, 12-bit and CMYK files are refused.
#include "ImageIO.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace imageio {
namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    // Canonical code tables: for each length, the first code, the first symbol index and the count.
    int firstCode[17] = {}, firstIndex[17] = {}, count[17] = {};
    unsigned char symbols[256] = {};
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocksW = 0, blocksH = 0;      // allocated size in blocks (whole MCUs)
    int dcPred = 0;
    std::vector<unsigned char> plane;  // blocksW*8 x blocksH*8 samples
};

struct BitReader {
    const unsigned char *p, *end;
    uint32_t bits = 0;
    int nbits = 0;
    bool hitMarker = false;

    void fill() {
        while (nbits <= 24) {
            int byte = 0;
            if (!hitMarker && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) {
                        p += 2;
                    } else {   // a marker: feed zeros from here on, the caller deals with it
                        hitMarker = true;
                        byte = 0;
                    }
                } else {
                    ++p;
                }
            }
            bits |= static_cast<uint32_t>(byte) << (24 - nbits);
            nbits += 8;
        }
    }
    int get(int n) {
        if (n == 0) return 0;
        if (nbits < n) fill();
        const int v = static_cast<int>(bits >> (32 - n));
        bits <<= n;
        nbits -= n;
        return v;
    }
    void reset() {
        bits = 0;
        nbits = 0;
        hitMarker = false;
    }
};

int decodeSymbol(BitReader &br, const Huffman &h) {
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
        code = (code << 1) | br.get(1);
        if (h.count[len] && code >= h.firstCode[len] && code < h.firstCode[len] + h.count[len])
            return h.symbols[h.firstIndex[len] + code - h.firstCode[len]];
    }
    return -1;
}

int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

inline unsigned char clamp255(int x) { return static_cast<unsigned char>(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// Integer inverse DCT, 12-bit fixed-point constants, columns then rows.
#define F2F(x) (static_cast<int>((x) * 4096 + 0.5))
#define IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                     \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;        \
    p2 = s2; p3 = s6;                                               \
    p1 = (p2 + p3) * F2F(0.5411961f);                               \
    t2 = p1 + p3 * F2F(-1.847759065f);                              \
    t3 = p1 + p2 * F2F(0.765366865f);                               \
    p2 = s0; p3 = s4;                                               \
    t0 = (p2 + p3) * 4096;                                          \
    t1 = (p2 - p3) * 4096;                                          \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;         \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                             \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;         \
    p5 = (p3 + p4) * F2F(1.175875602f);                             \
    t0 = t0 * F2F(0.298631336f);                                    \
    t1 = t1 * F2F(2.053119869f);                                    \
    t2 = t2 * F2F(3.072711026f);                                    \
    t3 = t3 * F2F(1.501321110f);                                    \
    p1 = p5 + p1 * F2F(-0.899976223f);                              \
    p2 = p5 + p2 * F2F(-2.562915447f);                              \
    p3 = p3 * F2F(-1.961570560f);                                   \
    p4 = p4 * F2F(-0.390180644f);                                   \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idctBlock(unsigned char *out, int stride, const short data[64]) {
    int val[64];
    for (int i = 0; i < 8; ++i) {
        const short *d = data + i;
        int *v = val + i;
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            const int dc = d[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; ++i) {
        const int *v = val + i * 8;
        unsigned char *o = out + i * stride;
        IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = clamp255((x0 + t3) >> 17); o[7] = clamp255((x0 - t3) >> 17);
        o[1] = clamp255((x1 + t2) >> 17); o[6] = clamp255((x1 - t2) >> 17);
        o[2] = clamp255((x2 + t1) >> 17); o[5] = clamp255((x2 - t1) >> 17);
        o[3] = clamp255((x3 + t0) >> 17); o[4] = clamp255((x3 - t0) >> 17);
    }
}
#undef IDCT_1D
#undef F2F

// Chroma rows at full resolution (triangle filter where a direction was subsampled by 2).
void upsampleRow(unsigned char *out, const unsigned char *nearRow, const unsigned char *farRow, int wIn, int hs, int vs) {
    if (hs == 1 && vs == 1) {
        std::memcpy(out, nearRow, static_cast<size_t>(wIn));
    } else if (hs == 1 && vs == 2) {
        for (int i = 0; i < wIn; ++i) out[i] = static_cast<unsigned char>((3 * nearRow[i] + farRow[i] + 2) >> 2);
    } else if (hs == 2 && vs == 1) {
        if (wIn == 1) { out[0] = out[1] = nearRow[0]; return; }
        out[0] = nearRow[0];
        out[1] = static_cast<unsigned char>((nearRow[0] * 3 + nearRow[1] + 2) >> 2);
        int i;
        for (i = 1; i < wIn - 1; ++i) {
            const int n = 3 * nearRow[i] + 2;
            out[i * 2 + 0] = static_cast<unsigned char>((n + nearRow[i - 1]) >> 2);
            out[i * 2 + 1] = static_cast<unsigned char>((n + nearRow[i + 1]) >> 2);
        }
        out[i * 2 + 0] = static_cast<unsigned char>((nearRow[wIn - 1] * 3 + nearRow[wIn - 2] + 2) >> 2);
        out[i * 2 + 1] = nearRow[wIn - 1];
    } else {   // 2 x 2
        int t1 = 3 * nearRow[0] + farRow[0];
        if (wIn == 1) { out[0] = out[1] = static_cast<unsigned char>((t1 + 2) >> 2); return; }
        out[0] = static_cast<unsigned char>((t1 + 2) >> 2);
        for (int i = 1; i < wIn; ++i) {
            const int t0 = t1;
            t1 = 3 * nearRow[i] + farRow[i];
            out[i * 2 - 1] = static_cast<unsigned char>((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = static_cast<unsigned char>((3 * t1 + t0 + 8) >> 4);
        }
        out[wIn * 2 - 1] = static_cast<unsigned char>((t1 + 2) >> 2);
    }
}

}  // namespace

bool DecodeJpegRGBA(const unsigned char *data, size_t size, int &width, int &height, std::vector<unsigned char> &rgba, std::string &why) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) { why = "not a JPEG stream"; return false; }
    uint16_t quant[4][64] = {};
    Huffman dc[4], ac[4];
    std::vector<Component> comps;
    int restartInterval = 0, hMax = 1, vMax = 1;
    bool haveFrame = false;
    size_t at = 2;
    while (at + 4 <= size) {
        if (data[at] != 0xFF) { why = "marker expected"; return false; }
        while (at < size && data[at] == 0xFF) ++at;   // fill bytes
        if (at >= size) break;
        const int marker = data[at++];
        if (marker == 0xD9) break;
        if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
        if (at + 2 > size) break;
        const size_t len = (static_cast<size_t>(data[at]) << 8) | data[at + 1];
        if (len < 2 || at + len > size) { why = "truncated segment"; return false; }
        const unsigned char *seg = data + at + 2;
        const size_t segLen = len - 2;
        if (marker == 0xDB) {                               // quantisation tables
            size_t o = 0;
            while (o < segLen) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                ++o;
                if (tq > 3 || o + (pq ? 128 : 64) > segLen) { why = "bad quantisation table"; return false; }
                for (int i = 0; i < 64; ++i) {
                    quant[tq][kZigzag[i]] = pq ? static_cast<uint16_t>((seg[o] << 8) | seg[o + 1]) : seg[o];
                    o += pq ? 2 : 1;
                }
            }
        } else if (marker == 0xC4) {                        // Huffman tables
            size_t o = 0;
            while (o + 17 <= segLen) {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                if (tc > 1 || th > 3) { why = "bad Huffman table"; return false; }
                Huffman &h = tc ? ac[th] : dc[th];
                int total = 0, code = 0;
                for (int l = 1; l <= 16; ++l) {
                    h.count[l] = seg[o + l];
                    h.firstIndex[l] = total;
                    h.firstCode[l] = code;
                    total += h.count[l];
                    code = (code + h.count[l]) << 1;
                }
                o += 17;
                if (total > 256 || o + total > segLen) { why = "bad Huffman table"; return false; }
                std::memcpy(h.symbols, seg + o, static_cast<size_t>(total));
                h.present = true;
                o += total;
            }
        } else if (marker == 0xC0 || marker == 0xC1) {      // baseline / extended sequential frame
            if (segLen < 6 || seg[0] != 8) { why = "only 8-bit samples are supported"; return false; }
            height = (seg[1] << 8) | seg[2];
            width = (seg[3] << 8) | seg[4];
            // a header may claim 65535 x 65535; textures beyond 16384 x 16384 are refused, not allocated
            if (width > 16384 || height > 16384) { why = "image larger than 16384 x 16384"; return false; }
            const int n = seg[5];
            if ((n != 1 && n != 3) || segLen < static_cast<size_t>(6 + 3 * n) || width <= 0 || height <= 0) {
                why = "only greyscale and YCbCr images are supported";
                return false;
            }
            comps.assign(static_cast<size_t>(n), Component());
            for (int i = 0; i < n; ++i) {
                Component &c = comps[static_cast<size_t>(i)];
                c.id = seg[6 + 3 * i];
                c.h = seg[7 + 3 * i] >> 4;
                c.v = seg[7 + 3 * i] & 15;
                c.tq = seg[8 + 3 * i];
                if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) { why = "unsupported sampling factors"; return false; }
                if (n == 1) c.h = c.v = 1;   // a single component is never interleaved: one block per MCU
                hMax = c.h > hMax ? c.h : hMax;
                vMax = c.v > vMax ? c.v : vMax;
            }
            haveFrame = true;
        } else if (marker == 0xC2 || (marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC)) {
            why = marker == 0xC2 ? "progressive JPEG is not supported" : "unsupported JPEG coding process";
            return false;
        } else if (marker == 0xDD) {
            if (segLen >= 2) restartInterval = (seg[0] << 8) | seg[1];
        } else if (marker == 0xDA) {                        // scan: all components interleaved
            if (!haveFrame) { why = "scan before frame header"; return false; }
            const int n = seg[0];
            if (n != static_cast<int>(comps.size()) || segLen < static_cast<size_t>(1 + 2 * n + 3)) {
                why = "only single-scan (interleaved) files are supported";
                return false;
            }
            for (int i = 0; i < n; ++i) {
                Component *c = nullptr;
                for (auto &k : comps)
                    if (k.id == seg[1 + 2 * i]) c = &k;
                if (!c) { why = "scan names an unknown component"; return false; }
                c->td = seg[2 + 2 * i] >> 4;
                c->ta = seg[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present) { why = "missing Huffman table"; return false; }
            }
            const int mcuW = 8 * hMax, mcuH = 8 * vMax;
            const int mcusX = (width + mcuW - 1) / mcuW, mcusY = (height + mcuH - 1) / mcuH;
            for (auto &c : comps) {
                c.blocksW = mcusX * c.h;
                c.blocksH = mcusY * c.v;
                c.plane.assign(static_cast<size_t>(c.blocksW) * 8 * c.blocksH * 8, 0);
                c.dcPred = 0;
            }
            BitReader br{data + at + len, data + size};
            int untilRestart = restartInterval;
            short block[64];
            for (int my = 0; my < mcusY; ++my) {
                for (int mx = 0; mx < mcusX; ++mx) {
                    for (auto &c : comps) {
                        for (int by = 0; by < c.v; ++by) {
                            for (int bx = 0; bx < c.h; ++bx) {
                                std::memset(block, 0, sizeof block);
                                const int t = decodeSymbol(br, dc[c.td]);
                                if (t < 0 || t > 11) { why = "corrupt entropy-coded data"; return false; }
                                c.dcPred += extend(br.get(t), t);
                                block[0] = static_cast<short>(c.dcPred * quant[c.tq][0]);
                                for (int k = 1; k < 64;) {
                                    const int rs = decodeSymbol(br, ac[c.ta]);
                                    if (rs < 0) { why = "corrupt entropy-coded data"; return false; }
                                    const int run = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (run != 15) break;   // end of block
                                        k += 16;
                                        continue;
                                    }
                                    k += run;
                                    if (k > 63) { why = "corrupt entropy-coded data"; return false; }
                                    const int z = kZigzag[k];
                                    block[z] = static_cast<short>(extend(br.get(sz), sz) * quant[c.tq][z]);
                                    ++k;
                                }
                                const int px = (mx * c.h + bx) * 8, py = (my * c.v + by) * 8;
                                idctBlock(&c.plane[static_cast<size_t>(py) * c.blocksW * 8 + px], c.blocksW * 8, block);
                            }
                        }
                    }
                    if (restartInterval && --untilRestart == 0 && !(my == mcusY - 1 && mx == mcusX - 1)) {
                        // Skip to the restart marker; predictions and the bit buffer start afresh.
                        const unsigned char *p = br.p;
                        while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
                        if (p + 1 >= br.end) { why = "restart marker missing"; return false; }
                        br.p = p + 2;
                        br.reset();
                        for (auto &c : comps) c.dcPred = 0;
                        untilRestart = restartInterval;
                    }
                }
            }
            // Colour conversion at full resolution.
            rgba.assign(static_cast<size_t>(width) * height * 4, 255);
            std::vector<unsigned char> rows[3];
            for (auto &r : rows) r.assign(static_cast<size_t>(mcusX) * mcuW + 16, 0);
            for (int y = 0; y < height; ++y) {
                for (size_t ci = 0; ci < comps.size(); ++ci) {
                    const Component &c = comps[ci];
                    const int hs = hMax / c.h, vs = vMax / c.v;
                    const int planeW = c.blocksW * 8, planeH = c.blocksH * 8;
                    const int wIn = (width + hs - 1) / hs;
                    int nearY = y, farY = y;
                    if (vs == 2) {   // rows y/2 and its neighbour on the side y lies towards
                        nearY = y >> 1;
                        farY = (y & 1) ? nearY + 1 : nearY - 1;
                        const int last = (height + 1) / 2 - 1;
                        if (farY < 0) farY = 0;
                        if (farY > last) farY = last;
                    }
                    if (nearY >= planeH) nearY = planeH - 1;
                    if (farY >= planeH) farY = planeH - 1;
                    upsampleRow(rows[ci].data(), &c.plane[static_cast<size_t>(nearY) * planeW], &c.plane[static_cast<size_t>(farY) * planeW], wIn, hs, vs);
                }
                unsigned char *out = &rgba[static_cast<size_t>(y) * width * 4];
                if (comps.size() == 1) {
                    for (int x = 0; x < width; ++x) out[x * 4] = out[x * 4 + 1] = out[x * 4 + 2] = rows[0][static_cast<size_t>(x)];
                } else {
                    for (int x = 0; x < width; ++x) {
                        const int yFixed = (rows[0][static_cast<size_t>(x)] << 20) + (1 << 19);
                        const int cb = rows[1][static_cast<size_t>(x)] - 128, cr = rows[2][static_cast<size_t>(x)] - 128;
                        const int r = yFixed + cr * (static_cast<int>(1.40200f * 4096.0f + 0.5f) << 8);
                        const int g = yFixed + cr * -(static_cast<int>(0.71414f * 4096.0f + 0.5f) << 8) +
                                      static_cast<int>((static_cast<unsigned>(cb * -(static_cast<int>(0.34414f * 4096.0f + 0.5f) << 8))) & 0xffff0000u);
                        const int b = yFixed + cb * (static_cast<int>(1.77200f * 4096.0f + 0.5f) << 8);
                        out[x * 4 + 0] = clamp255(r >> 20);
                        out[x * 4 + 1] = clamp255(g >> 20);
                        out[x * 4 + 2] = clamp255(b >> 20);
                    }
                }
            }
            return true;
        }
        at += len;
    }
    why = "no image data found";
    return false;
}

}  // namespace imageio
