// RGBA8 PNG encoder (replaces Image::save / save_image, libpng.cpp:73-107, which needs libpng's headers;
// the image has only zlib).  8 bits per channel, colour type 6, no interlace, filter 0 on every row --
// the settings the reference passes to png_set_IHDR (libpng.cpp:87-93).
#include <cstdio>
#include <cstring>
#include <vector>
#include <zlib.h>

#include "host_scene.h"

namespace {
void put32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(FILE* f, const char* tag, const std::vector<unsigned char>& data)
{
  std::vector<unsigned char> hdr;
  put32(hdr, (uint32_t)data.size());
  fwrite(hdr.data(), 1, 4, f);
  uint32_t crc = crc32(0L, (const Bytef*)tag, 4);
  if (!data.empty()) crc = crc32(crc, data.data(), (uInt)data.size());
  fwrite(tag, 1, 4, f);
  if (!data.empty()) fwrite(data.data(), 1, data.size(), f);
  std::vector<unsigned char> c; put32(c, crc);
  fwrite(c.data(), 1, 4, f);
}
} // namespace

extern "C" int mirt_write_png(const char* path, const uint8_t* rgba, int width, int height)
{
  if (!path || !rgba || width <= 0 || height <= 0) { mirt::set_error("mirt_write_png: bad argument"); return MIRT_ERR_ARG; }
  const size_t row = (size_t)width * 4;
  std::vector<unsigned char> raw((row + 1) * (size_t)height);
  for (int y = 0; y < height; ++y) {
    raw[(row + 1) * y] = 0;
    memcpy(&raw[(row + 1) * y + 1], rgba + row * y, row);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<unsigned char> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), Z_DEFAULT_COMPRESSION) != Z_OK) {
    mirt::set_error("mirt_write_png: zlib compress2 failed"); return MIRT_ERR_IO;
  }
  z.resize(zlen);
  FILE* f = fopen(path, "wb");
  if (!f) { mirt::set_error(std::string("mirt_write_png: cannot open ") + path); return MIRT_ERR_IO; }
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  fwrite(sig, 1, 8, f);
  std::vector<unsigned char> ihdr;
  put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
  ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk(f, "IHDR", ihdr);
  chunk(f, "IDAT", z);
  chunk(f, "IEND", std::vector<unsigned char>());
  fclose(f);
  return MIRT_OK;
}
