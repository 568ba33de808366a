// Image files of Application::screenshot (Application.cpp:2231-2335). The reference hands the pixels to DevIL
// (ilTexImage + ilSaveImage, third party, not in this image); these writers produce the same two formats directly:
//   * 8-bit RGB PNG: one IDAT chunk of stored (uncompressed) deflate blocks — valid for every PNG reader, no zlib;
//   * Radiance .hdr: "#?RADIANCE" header, 32-bit RGBE pixels in flat (non-run-length) scanlines, the alpha channel of
//     the RGBA32F buffer is dropped as the format has none.
// Both store the top row first; the renderer's buffers have row 0 at the bottom (bottomUp).
#include "image_files.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace twk {

namespace {

uint32_t crcTable[256];
bool     crcReady = false;

uint32_t crc32(uint32_t crc, const unsigned char* data, size_t n)
{
  if (!crcReady)
  {
    for (uint32_t i = 0; i < 256; ++i)
    {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
      crcTable[i] = c;
    }
    crcReady = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = crcTable[(crc ^ data[i]) & 0xffu] ^ (crc >> 8);
  return ~crc;
}

void put32(std::vector<unsigned char>& v, uint32_t x)
{
  v.push_back((unsigned char) (x >> 24)); v.push_back((unsigned char) (x >> 16)); v.push_back((unsigned char) (x >> 8)); v.push_back((unsigned char) x);
}

bool writeChunk(FILE* f, const char type[4], const std::vector<unsigned char>& payload)
{
  std::vector<unsigned char> head;
  put32(head, (uint32_t) payload.size());
  head.insert(head.end(), type, type + 4);
  uint32_t crc = crc32(0u, reinterpret_cast<const unsigned char*>(type), 4);
  if (!payload.empty()) crc = crc32(crc, payload.data(), payload.size());
  std::vector<unsigned char> tail;
  put32(tail, crc);
  if (fwrite(head.data(), 1, head.size(), f) != head.size()) return false;
  if (!payload.empty() && fwrite(payload.data(), 1, payload.size(), f) != payload.size()) return false;
  return fwrite(tail.data(), 1, tail.size(), f) == tail.size();
}

} // namespace

bool writePngRgb8(const std::string& path, int width, int height, const unsigned char* rgb8, bool bottomUp, std::string& error)
{
  if (width <= 0 || height <= 0 || !rgb8) { error = "writePngRgb8: empty image"; return false; }
  // raw scanlines: filter byte 0 + RGB
  const size_t rowBytes = (size_t) width * 3;
  std::vector<unsigned char> raw;
  raw.reserve(((size_t) rowBytes + 1) * height);
  for (int y = 0; y < height; ++y)
  {
    const unsigned char* row = rgb8 + rowBytes * (size_t) (bottomUp ? height - 1 - y : y);
    raw.push_back(0);
    raw.insert(raw.end(), row, row + rowBytes);
  }
  // zlib stream: header, stored blocks of at most 65535 bytes, Adler-32 of the raw data
  std::vector<unsigned char> z;
  z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
  z.push_back(0x78); z.push_back(0x01);
  size_t pos = 0;
  while (pos < raw.size())
  {
    const size_t n = (raw.size() - pos < 65535) ? raw.size() - pos : 65535;
    z.push_back((pos + n == raw.size()) ? 1 : 0);
    z.push_back((unsigned char) (n & 0xff)); z.push_back((unsigned char) (n >> 8));
    z.push_back((unsigned char) (~n & 0xff)); z.push_back((unsigned char) ((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  }
  uint32_t a = 1, b = 0;
  for (size_t i = 0; i < raw.size(); ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
  put32(z, (b << 16) | a);

  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { error = "writePngRgb8: cannot open " + path; return false; }
  static const unsigned char signature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<unsigned char> ihdr;
  put32(ihdr, (uint32_t) width); put32(ihdr, (uint32_t) height);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); // 8 bit, RGB, deflate, no filter, no interlace
  bool ok = fwrite(signature, 1, 8, f) == 8;
  ok = ok && writeChunk(f, "IHDR", ihdr);
  ok = ok && writeChunk(f, "IDAT", z);
  ok = ok && writeChunk(f, "IEND", std::vector<unsigned char>());
  ok = (fclose(f) == 0) && ok;
  if (!ok) error = "writePngRgb8: write to " + path + " failed";
  return ok;
}

// Ward's float → RGBE: shared exponent of the largest component, mantissas truncated.
void floatToRgbe(float r, float g, float b, unsigned char rgbe[4])
{
  float v = r; if (g > v) v = g; if (b > v) v = b;
  if (!(v >= 1.0e-32f)) { rgbe[0] = rgbe[1] = rgbe[2] = rgbe[3] = 0; return; } // also NaN
  if (v > 1.0e38f) v = 1.0e38f; // inf: the largest exponent
  int e = 0;
  const float scale = frexpf(v, &e) * 256.0f / v;
  const float rs = r * scale, gs = g * scale, bs = b * scale;
  rgbe[0] = (unsigned char) (rs > 0.0f ? (rs < 255.0f ? rs : 255.0f) : 0.0f);
  rgbe[1] = (unsigned char) (gs > 0.0f ? (gs < 255.0f ? gs : 255.0f) : 0.0f);
  rgbe[2] = (unsigned char) (bs > 0.0f ? (bs < 255.0f ? bs : 255.0f) : 0.0f);
  rgbe[3] = (unsigned char) (e + 128);
}

bool writeHdrRgba32f(const std::string& path, int width, int height, const float* rgba, bool bottomUp, std::string& error)
{
  if (width <= 0 || height <= 0 || !rgba) { error = "writeHdrRgba32f: empty image"; return false; }
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { error = "writeHdrRgba32f: cannot open " + path; return false; }
  bool ok = fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", height, width) > 0;
  std::vector<unsigned char> row((size_t) width * 4);
  for (int y = 0; y < height && ok; ++y)
  {
    const float* src = rgba + (size_t) 4 * width * (size_t) (bottomUp ? height - 1 - y : y);
    for (int x = 0; x < width; ++x) floatToRgbe(src[4 * x + 0], src[4 * x + 1], src[4 * x + 2], &row[(size_t) 4 * x]);
    ok = fwrite(row.data(), 1, row.size(), f) == row.size();
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) error = "writeHdrRgba32f: write to " + path + " failed";
  return ok;
}

} // namespace twk
