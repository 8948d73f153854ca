// Host-side helpers of the I/O either side of the animation loop (reference basicsr/demo.py:166-185 reads the driving clip, :222 writes the
// result).  No device code: this translation unit only rides in libsmx.so so that the Python host layer reaches it through the same ctypes handle --
// and, called through ctypes, it runs WITHOUT the interpreter lock: a thread pool of decoders / encoders scales with the host's cores
// (driver.LazyFrames, png.decode_png / encode_png).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "smx.h"

// PNG scanline reconstruction (RFC 2083 section 6: filter types 0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth), 8 bits per sample.
// raw: h rows of (1 filter byte + stride bytes); out: h rows of stride bytes; bpp: bytes per pixel (1 | 2 | 3 | 4).
extern "C" int smx_png_unfilter_u8(const uint8_t* raw, int h, int stride, int bpp, uint8_t* out) {
  if (!raw || !out || h <= 0 || stride <= 0 || bpp < 1 || bpp > 4) return SMX_EINVAL;
  const uint8_t* prev = nullptr;
  for (int y = 0; y < h; ++y) {
    const uint8_t ft = raw[(size_t)y * (stride + 1)];
    const uint8_t* line = raw + (size_t)y * (stride + 1) + 1;
    uint8_t* cur = out + (size_t)y * stride;
    switch (ft) {
      case 0: memcpy(cur, line, stride); break;
      case 1:
        for (int x = 0; x < stride; ++x) cur[x] = (uint8_t)(line[x] + (x >= bpp ? cur[x - bpp] : 0));
        break;
      case 2:
        for (int x = 0; x < stride; ++x) cur[x] = (uint8_t)(line[x] + (prev ? prev[x] : 0));
        break;
      case 3:
        for (int x = 0; x < stride; ++x) {
          const int a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0;
          cur[x] = (uint8_t)(line[x] + ((a + b) >> 1));
        }
        break;
      case 4:
        for (int x = 0; x < stride; ++x) {
          const int a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0, c = (prev && x >= bpp) ? prev[x - bpp] : 0;
          const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
          cur[x] = (uint8_t)(line[x] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)));
        }
        break;
      default: return SMX_EINVAL;
    }
    prev = cur;
  }
  return SMX_OK;
}
