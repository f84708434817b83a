// Huffman-coded 8-bit JPEG -> RGBA8: baseline, extended-sequential and progressive frames, 1 or 3
// components, chroma subsampled by 1 or 2 in each direction, restart intervals, interleaved and
// per-component scans.  The reference decodes textures with stb_image through tinygltf (reference
// src/MeshData.cpp:419-425, 558-580), which is not available; this decoder follows the same family
// of algorithms (integer IDCT of the IJG slow kind, triangle-filter chroma upsampling, fixed-point
// YCbCr conversion).  Arithmetic-coded, lossless, 12-bit and CMYK files are refused.
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
    int blocksW = 0, blocksH = 0;            // allocated size in blocks (whole MCUs)
    int blocksAcross = 0, blocksDown = 0;    // blocks that hold image samples: the extent of a scan of this component alone
    int dcPred = 0;
    std::vector<unsigned char> plane;        // blocksW*8 x blocksH*8 samples
    std::vector<short> coef;                 // blocksW x blocksH x 64 coefficients, kept when the file has several scans
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

// DC prediction of a component: running sum of differences.  A damaged file can run it past the
// range of int; it wraps then (defined for unsigned) instead of overflowing.
inline int addWrapped(int sum, int diff) { return static_cast<int>(static_cast<unsigned>(sum) + static_cast<unsigned>(diff)); }

inline unsigned char clamp255(long long x) { return static_cast<unsigned char>(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// Integer inverse DCT, 12-bit fixed-point constants, columns then rows.  The sums are kept in 64
// bits: the coefficients of a well-formed file stay far below 2^31 here, those of a damaged one need not.
#define F2F(x) (static_cast<int>((x) * 4096 + 0.5))
#define IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                     \
    long long t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;  \
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
            v[0] = static_cast<int>((x0 + t3) >> 10); v[56] = static_cast<int>((x0 - t3) >> 10);
            v[8] = static_cast<int>((x1 + t2) >> 10); v[48] = static_cast<int>((x1 - t2) >> 10);
            v[16] = static_cast<int>((x2 + t1) >> 10); v[40] = static_cast<int>((x2 - t1) >> 10);
            v[24] = static_cast<int>((x3 + t0) >> 10); v[32] = static_cast<int>((x3 - t0) >> 10);
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

namespace {

// One scan's parameters: spectral band [ss, se], successive-approximation bit positions (ah = the
// previous pass's position, 0 on a first pass; al = this pass's).
struct Scan {
    int ss = 0, se = 63, ah = 0, al = 0;
    int eobRun = 0;
};

// A whole block of a sequential scan: DC difference, then run/size pairs up to the end-of-block code.
bool decodeSequentialBlock(BitReader &br, const Huffman &dcTable, const Huffman &acTable, Component &c, short *block) {
    const int t = decodeSymbol(br, dcTable);
    if (t < 0 || t > 11) return false;
    c.dcPred = addWrapped(c.dcPred, extend(br.get(t), t));
    block[0] = static_cast<short>(c.dcPred);
    for (int k = 1; k < 64;) {
        const int rs = decodeSymbol(br, acTable);
        if (rs < 0) return false;
        const int run = rs >> 4, sz = rs & 15;
        if (sz == 0) {
            if (run != 15) break;   // end of block
            k += 16;
            continue;
        }
        k += run;
        if (k > 63) return false;
        block[kZigzag[k]] = static_cast<short>(extend(br.get(sz), sz));
        ++k;
    }
    return true;
}

// Progressive DC: the first pass carries the difference shifted down by al, later passes one more bit each.
bool decodeProgressiveDC(BitReader &br, const Huffman &dcTable, Component &c, const Scan &sc, short *block) {
    if (sc.ah == 0) {
        const int t = decodeSymbol(br, dcTable);
        if (t < 0 || t > 11) return false;
        c.dcPred = addWrapped(c.dcPred, extend(br.get(t), t));
        block[0] = static_cast<short>(static_cast<unsigned>(c.dcPred) << sc.al);
    } else if (br.get(1)) {
        block[0] = static_cast<short>(block[0] + (1 << sc.al));
    }
    return true;
}

// Progressive AC, first pass over a band: like the sequential codes, plus end-of-band runs that
// close this block's band and that of the following eobRun blocks.
bool decodeProgressiveACFirst(BitReader &br, const Huffman &acTable, Scan &sc, short *block) {
    if (sc.eobRun > 0) {
        --sc.eobRun;
        return true;
    }
    for (int k = sc.ss; k <= sc.se;) {
        const int rs = decodeSymbol(br, acTable);
        if (rs < 0) return false;
        const int run = rs >> 4, sz = rs & 15;
        if (sz == 0) {
            if (run < 15) {
                sc.eobRun = (1 << run) - 1;
                if (run) sc.eobRun += br.get(run);
                break;
            }
            k += 16;
            continue;
        }
        k += run;
        if (k > sc.se) return false;
        block[kZigzag[k]] = static_cast<short>(extend(br.get(sz), sz) * (1 << sc.al));
        ++k;
    }
    return true;
}

// One more bit for a coefficient that earlier passes already made non-zero.
inline void refineCoefficient(BitReader &br, short &value, int bit) {
    if (br.get(1) && (value & bit) == 0) value = static_cast<short>(value > 0 ? value + bit : value - bit);
}

// Progressive AC, refinement pass: each code places one new +-1 coefficient (at bit al) after a
// run of still-zero ones; every already non-zero coefficient passed on the way takes a correction bit.
bool decodeProgressiveACRefine(BitReader &br, const Huffman &acTable, Scan &sc, short *block) {
    const int bit = 1 << sc.al;
    int k = sc.ss;
    if (sc.eobRun > 0) {
        --sc.eobRun;
        for (; k <= sc.se; ++k) {
            short &value = block[kZigzag[k]];
            if (value != 0) refineCoefficient(br, value, bit);
        }
        return true;
    }
    while (k <= sc.se) {
        const int rs = decodeSymbol(br, acTable);
        if (rs < 0) return false;
        int run = rs >> 4;
        const int sz = rs & 15;
        int placed = 0;
        if (sz == 0) {
            if (run < 15) {   // end of band: the rest of this block only refines
                sc.eobRun = (1 << run) - 1;
                if (run) sc.eobRun += br.get(run);
                run = 64;
            }   // else: sixteen zeros, i.e. fifteen and a "new coefficient" of value zero
        } else {
            if (sz != 1) return false;
            placed = br.get(1) ? bit : -bit;
        }
        while (k <= sc.se) {
            short &value = block[kZigzag[k++]];
            if (value != 0) {
                refineCoefficient(br, value, bit);
            } else {
                if (run == 0) {
                    if (placed) value = static_cast<short>(placed);
                    break;
                }
                --run;
            }
        }
    }
    return true;
}

struct Decoder {
    const unsigned char *data;
    size_t size;
    std::string &why;
    uint16_t quant[4][64] = {};
    Huffman dc[4], ac[4];
    std::vector<Component> comps;
    int width = 0, height = 0;
    int restartInterval = 0, hMax = 1, vMax = 1, mcusX = 0, mcusY = 0;
    bool haveFrame = false, progressive = false, keepCoefficients = false, sawScan = false;

    Decoder(const unsigned char *d, size_t n, std::string &w) : data(d), size(n), why(w) {}

    bool fail(const char *message) {
        why = message;
        return false;
    }

    bool readQuantTables(const unsigned char *seg, size_t segLen) {
        size_t o = 0;
        while (o < segLen) {
            const int pq = seg[o] >> 4, tq = seg[o] & 15;
            ++o;
            if (tq > 3 || o + (pq ? 128 : 64) > segLen) return fail("bad quantisation table");
            for (int i = 0; i < 64; ++i) {
                quant[tq][kZigzag[i]] = pq ? static_cast<uint16_t>((seg[o] << 8) | seg[o + 1]) : seg[o];
                o += pq ? 2 : 1;
            }
        }
        return true;
    }

    bool readHuffmanTables(const unsigned char *seg, size_t segLen) {
        size_t o = 0;
        while (o + 17 <= segLen) {
            const int tc = seg[o] >> 4, th = seg[o] & 15;
            if (tc > 1 || th > 3) return fail("bad Huffman table");
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
            if (total > 256 || o + total > segLen) return fail("bad Huffman table");
            std::memcpy(h.symbols, seg + o, static_cast<size_t>(total));
            h.present = true;
            o += total;
        }
        return true;
    }

    bool readFrame(const unsigned char *seg, size_t segLen, bool isProgressive) {
        if (haveFrame) return fail("second frame header");
        if (segLen < 6 || seg[0] != 8) return fail("only 8-bit samples are supported");
        height = (seg[1] << 8) | seg[2];
        width = (seg[3] << 8) | seg[4];
        // a header may claim 65535 x 65535; textures beyond 16384 x 16384 are refused, not allocated
        if (width > 16384 || height > 16384) return fail("image larger than 16384 x 16384");
        // Every 8x8 block costs at least one bit of entropy-coded data, so a stream of `size` bytes cannot hold
        // more than size * 8 * 64 pixels: a short stream with a large header is refused before its planes are allocated.
        if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) > static_cast<uint64_t>(size) * 512u) return fail("stream too short for the size the header gives");
        const int n = seg[5];
        if ((n != 1 && n != 3) || segLen < static_cast<size_t>(6 + 3 * n) || width <= 0 || height <= 0)
            return fail("only greyscale and YCbCr images are supported");
        comps.assign(static_cast<size_t>(n), Component());
        for (int i = 0; i < n; ++i) {
            Component &c = comps[static_cast<size_t>(i)];
            c.id = seg[6 + 3 * i];
            c.h = seg[7 + 3 * i] >> 4;
            c.v = seg[7 + 3 * i] & 15;
            c.tq = seg[8 + 3 * i];
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return fail("unsupported sampling factors");
            if (n == 1) c.h = c.v = 1;   // a single component is never interleaved: one block per MCU
            hMax = c.h > hMax ? c.h : hMax;
            vMax = c.v > vMax ? c.v : vMax;
        }
        const int mcuW = 8 * hMax, mcuH = 8 * vMax;
        mcusX = (width + mcuW - 1) / mcuW;
        mcusY = (height + mcuH - 1) / mcuH;
        for (auto &c : comps) {
            c.blocksW = mcusX * c.h;
            c.blocksH = mcusY * c.v;
            c.blocksAcross = ((width * c.h + hMax - 1) / hMax + 7) / 8;
            c.blocksDown = ((height * c.v + vMax - 1) / vMax + 7) / 8;
            c.plane.assign(static_cast<size_t>(c.blocksW) * 8 * c.blocksH * 8, 0);
        }
        progressive = isProgressive;
        haveFrame = true;
        return true;
    }

    void storeBlock(Component &c, int bx, int by, const short *coefficients) {
        short block[64];
        for (int z = 0; z < 64; ++z) block[z] = static_cast<short>(coefficients[z] * quant[c.tq][z]);
        idctBlock(&c.plane[static_cast<size_t>(by) * 8 * c.blocksW * 8 + static_cast<size_t>(bx) * 8], c.blocksW * 8, block);
    }

    // Decodes the entropy-coded segment that starts at `from`; returns where it stopped reading.
    bool readScan(const unsigned char *seg, size_t segLen, const unsigned char *from, const unsigned char *&stoppedAt) {
        if (!haveFrame) return fail("scan before frame header");
        const int n = segLen ? seg[0] : 0;
        if (n < 1 || n > static_cast<int>(comps.size()) || segLen < static_cast<size_t>(1 + 2 * n + 3)) return fail("bad scan header");
        std::vector<Component *> inScan;
        for (int i = 0; i < n; ++i) {
            Component *c = nullptr;
            for (auto &k : comps)
                if (k.id == seg[1 + 2 * i]) c = &k;
            if (!c) return fail("scan names an unknown component");
            for (Component *earlier : inScan)
                if (earlier == c) return fail("scan names a component twice");
            c->td = seg[2 + 2 * i] >> 4;
            c->ta = seg[2 + 2 * i] & 15;
            if (c->td > 3 || c->ta > 3) return fail("bad Huffman table selector");
            inScan.push_back(c);
        }
        Scan sc;
        sc.ss = seg[1 + 2 * n];
        sc.se = seg[2 + 2 * n];
        sc.ah = seg[3 + 2 * n] >> 4;
        sc.al = seg[3 + 2 * n] & 15;
        if (progressive) {
            if (sc.ss > 63 || sc.se > 63 || sc.ss > sc.se || sc.ah > 13 || sc.al > 13) return fail("bad progressive scan parameters");
            if ((sc.ss == 0) != (sc.se == 0)) return fail("bad progressive scan parameters");   // DC and AC never share a scan
            if (sc.ss > 0 && n != 1) return fail("bad progressive scan parameters");            // AC scans cover one component
        } else {
            sc.ss = 0;
            sc.se = 63;
            sc.ah = sc.al = 0;
        }
        for (Component *c : inScan) {
            const bool needsDC = sc.ss == 0 && sc.ah == 0, needsAC = sc.se > 0;
            if ((needsDC && !dc[c->td].present) || (needsAC && !ac[c->ta].present)) return fail("missing Huffman table");
        }
        // A sequential file whose first scan carries every component goes straight to samples (and
        // ends the decode); anything else collects coefficients until the end of the image.
        const bool straightToSamples = !progressive && !sawScan && n == static_cast<int>(comps.size());
        if (!straightToSamples && !keepCoefficients) {
            keepCoefficients = true;
            for (auto &c : comps) c.coef.assign(static_cast<size_t>(c.blocksW) * c.blocksH * 64, 0);
        }
        sawScan = true;
        for (Component *c : inScan) c->dcPred = 0;

        BitReader br{from, data + size};
        const bool alone = n == 1;
        const int unitsX = alone ? inScan[0]->blocksAcross : mcusX, unitsY = alone ? inScan[0]->blocksDown : mcusY;
        int untilRestart = restartInterval;
        short direct[64];
        for (int uy = 0; uy < unitsY; ++uy) {
            for (int ux = 0; ux < unitsX; ++ux) {
                for (Component *c : inScan) {
                    const int across = alone ? 1 : c->h, down = alone ? 1 : c->v;
                    for (int j = 0; j < down; ++j) {
                        for (int i = 0; i < across; ++i) {
                            const int bx = ux * across + i, by = uy * down + j;
                            short *block = direct;
                            if (keepCoefficients) block = &c->coef[(static_cast<size_t>(by) * c->blocksW + bx) * 64];
                            else std::memset(direct, 0, sizeof direct);
                            bool ok;
                            if (!progressive) ok = decodeSequentialBlock(br, dc[c->td], ac[c->ta], *c, block);
                            else if (sc.ss == 0) ok = decodeProgressiveDC(br, dc[c->td], *c, sc, block);
                            else if (sc.ah == 0) ok = decodeProgressiveACFirst(br, ac[c->ta], sc, block);
                            else ok = decodeProgressiveACRefine(br, ac[c->ta], sc, block);
                            if (!ok) return fail("corrupt entropy-coded data");
                            if (!keepCoefficients) storeBlock(*c, bx, by, block);
                        }
                    }
                }
                if (restartInterval && --untilRestart == 0 && !(uy == unitsY - 1 && ux == unitsX - 1)) {
                    // Skip to the restart marker; predictions, end-of-band runs and the bit buffer start afresh.
                    const unsigned char *p = br.p;
                    while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
                    if (p + 1 >= br.end) return fail("restart marker missing");
                    br.p = p + 2;
                    br.reset();
                    for (Component *c : inScan) c->dcPred = 0;
                    sc.eobRun = 0;
                    untilRestart = restartInterval;
                }
            }
        }
        stoppedAt = br.p;
        return true;
    }

    // Samples of every component from the collected coefficients.
    void finishCoefficients() {
        for (auto &c : comps)
            for (int by = 0; by < c.blocksH; ++by)
                for (int bx = 0; bx < c.blocksW; ++bx) storeBlock(c, bx, by, &c.coef[(static_cast<size_t>(by) * c.blocksW + bx) * 64]);
    }

    // Colour conversion at full resolution.
    void toRGBA(std::vector<unsigned char> &rgba) const {
        const int mcuW = 8 * hMax;
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
    }

    bool run(std::vector<unsigned char> &rgba) {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG stream");
        size_t at = 2;
        while (at + 2 <= size) {
            if (data[at] != 0xFF) return fail("marker expected");
            while (at < size && data[at] == 0xFF) ++at;   // fill bytes
            if (at >= size) break;
            const int marker = data[at++];
            if (marker == 0xD9) break;
            if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
            if (at + 2 > size) break;
            const size_t len = (static_cast<size_t>(data[at]) << 8) | data[at + 1];
            if (len < 2 || at + len > size) return fail("truncated segment");
            const unsigned char *seg = data + at + 2;
            const size_t segLen = len - 2;
            at += len;
            if (marker == 0xDB) {
                if (!readQuantTables(seg, segLen)) return false;
            } else if (marker == 0xC4) {
                if (!readHuffmanTables(seg, segLen)) return false;
            } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {
                if (!readFrame(seg, segLen, marker == 0xC2)) return false;
            } else if (marker >= 0xC3 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC) {
                return fail("unsupported JPEG coding process");
            } else if (marker == 0xDD) {
                if (segLen >= 2) restartInterval = (seg[0] << 8) | seg[1];
            } else if (marker == 0xDA) {
                const unsigned char *stoppedAt = nullptr;
                if (!readScan(seg, segLen, data + at, stoppedAt)) return false;
                if (!keepCoefficients) break;   // the one scan of a sequential file: samples are in place
                // Next marker after the entropy-coded bytes (stuffed zeros and restart markers belong to them).
                size_t p = static_cast<size_t>(stoppedAt - data);
                while (p + 1 < size && !(data[p] == 0xFF && data[p + 1] != 0x00 && data[p + 1] != 0xFF && !(data[p + 1] >= 0xD0 && data[p + 1] <= 0xD7))) ++p;
                at = p;
            }
        }
        if (!sawScan) return fail("no image data found");
        if (keepCoefficients) finishCoefficients();
        toRGBA(rgba);
        return true;
    }
};

}  // namespace

bool DecodeJpegRGBA(const unsigned char *data, size_t size, int &width, int &height, std::vector<unsigned char> &rgba, std::string &why) {
    Decoder decoder(data, size, why);
    if (!decoder.run(rgba)) return false;
    width = decoder.width;
    height = decoder.height;
    return true;
}

}  // namespace imageio
