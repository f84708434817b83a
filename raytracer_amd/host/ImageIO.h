// Image output for rendered frames: 8-bit PNG of the display pass and float PFM of the
// progressive mean.  The reference shows frames only in its window (src/main.cpp:36-69) and
// has no file output beyond the loader's debug BMP dump (src/MeshData.cpp:330-415); these
// writers give a headless render somewhere to go.
#pragma once

#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>

namespace imageio {

// rgba8: width*height*4 bytes, row 0 = bottom row (the frame layout of the backend); the file
// stores rows top-down.  Returns false if the file cannot be written.
bool WritePNG(const std::string &path, const uint8_t *rgba8, int width, int height);

// rgba: width*height*4 floats, row 0 = bottom row (PFM's own order); writes the three colour
// channels scaled by `scale` (1/frames turns a sum buffer into the mean) as little-endian PFM.
bool WritePFM(const std::string &path, const float *rgba, int width, int height, float scale);

// Huffman-coded 8-bit JPEG (baseline, extended-sequential or progressive; greyscale or YCbCr)
// -> RGBA8, rows top-down.  On failure `why` says what is unsupported or broken.
bool DecodeJpegRGBA(const unsigned char *data, size_t size, int &width, int &height, std::vector<unsigned char> &rgba, std::string &why);

// PNG (any colour type and bit depth of the standard, plain or Adam7-interlaced, tRNS
// transparency) -> RGBA8, rows top-down; 16-bit samples keep their high byte.
bool DecodePngRGBA(const unsigned char *data, size_t size, int &width, int &height, std::vector<unsigned char> &rgba, std::string &why);

}  // namespace imageio
