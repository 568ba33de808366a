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
#include <algorithm>
#include <vector>

#include <zlib.h> // inflate for the PNG reader

namespace twk {

namespace {

uint32_t crcTable[256];
bool     crcReady = false;

uint32_t chunkCrc(uint32_t crc, const unsigned char* data, size_t n)
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
  uint32_t crc = chunkCrc(0u, reinterpret_cast<const unsigned char*>(type), 4);
  if (!payload.empty()) crc = chunkCrc(crc, payload.data(), payload.size());
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

// ---- readers ------------------------------------------------------------------------------------
// Picture::load (Picture.cpp:231-560) hands the file to DevIL with IL_ORIGIN_LOWER_LEFT and Texture::create*
// (Texture.cpp:933-1042,1300-1377) expands every format to four channels — luminance → (L, L, L, 1), luminance-alpha →
// (L, L, L, A), RGB → (R, G, B, 1) — with integer formats read as normalised floats. The readers below produce exactly
// that RGBA32F, row 0 = BOTTOM row, for the file types the reference's scenes use besides JPEG: PNG (zlib inflate;
// 1-16 bit, all colour types, not interlaced), Radiance .hdr (run-length and flat scanlines) and PFM.
namespace twk {

namespace {

bool readFile(const std::string& path, std::vector<unsigned char>& data)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  data.resize(n > 0 ? (size_t) n : 0);
  const bool ok = data.empty() || fread(data.data(), 1, data.size(), f) == data.size();
  fclose(f);
  return ok;
}

uint32_t get32(const unsigned char* p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : ((pb <= pc) ? b : c);
}

bool decodePng(const std::vector<unsigned char>& file, int& width, int& height, std::vector<float>& rgba, std::string& error)
{
  size_t pos = 8;
  int depth = 0, colour = 0, interlace = 0;
  std::vector<unsigned char> idat, palette, paletteAlpha;
  bool haveHeader = false, done = false;
  while (!done && pos + 12 <= file.size())
  {
    const uint32_t n = get32(&file[pos]);
    if (pos + 12 + (size_t) n > file.size()) { error = "PNG: truncated chunk"; return false; }
    const unsigned char* type = &file[pos + 4];
    const unsigned char* body = &file[pos + 8];
    if (chunkCrc(0u, type, 4 + (size_t) n) != get32(body + n)) { error = "PNG: chunk checksum mismatch"; return false; }
    if (!memcmp(type, "IHDR", 4) && n == 13)
    {
      width = (int) get32(body); height = (int) get32(body + 4);
      depth = body[8]; colour = body[9]; interlace = body[12];
      haveHeader = true;
    }
    else if (!memcmp(type, "PLTE", 4)) palette.assign(body, body + n);
    else if (!memcmp(type, "tRNS", 4)) paletteAlpha.assign(body, body + n);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + n);
    else if (!memcmp(type, "IEND", 4)) done = true;
    pos += 12 + (size_t) n;
  }
  if (!haveHeader || width <= 0 || height <= 0) { error = "PNG: no header"; return false; }
  if (interlace != 0) { error = "PNG: Adam7 interlaced files are not supported"; return false; }
  int channels = 0;
  switch (colour) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; default: error = "PNG: bad colour type"; return false; }
  const bool depthOk = (colour == 0) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                     : (colour == 3) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
  if (!depthOk) { error = "PNG: bad bit depth"; return false; }
  if (colour == 3 && palette.size() < 3) { error = "PNG: palette missing"; return false; }

  const size_t rowBytes = ((size_t) width * channels * depth + 7) / 8;
  std::vector<unsigned char> raw((rowBytes + 1) * (size_t) height);
  uLongf rawSize = (uLongf) raw.size();
  if (uncompress(raw.data(), &rawSize, idat.data(), (uLong) idat.size()) != Z_OK || rawSize != raw.size()) { error = "PNG: inflate failed"; return false; }

  // undo the scanline filters in place
  const size_t bpp = (size_t) std::max(1, channels * depth / 8);
  for (int y = 0; y < height; ++y)
  {
    unsigned char* row = &raw[(rowBytes + 1) * (size_t) y + 1];
    const unsigned char* up = (y > 0) ? row - (rowBytes + 1) : nullptr;
    const int filter = row[-1];
    for (size_t i = 0; i < rowBytes; ++i)
    {
      const int a = (i >= bpp) ? row[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int v = row[i];
      switch (filter) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, c); break; default: error = "PNG: bad filter"; return false; }
      row[i] = (unsigned char) v;
    }
  }

  rgba.assign((size_t) width * height * 4, 1.0f);
  const float maxValue = (float) ((1u << depth) - 1u); // normalised-float read: value / (2^depth - 1)
  for (int y = 0; y < height; ++y)
  {
    const unsigned char* row = &raw[(rowBytes + 1) * (size_t) y + 1];
    float* dst = &rgba[(size_t) 4 * width * (size_t) (height - 1 - y)]; // file is top-down, the buffer bottom-up
    for (int x = 0; x < width; ++x)
    {
      unsigned int s[4] = {0, 0, 0, 0};
      for (int c = 0; c < channels; ++c)
      {
        const size_t k = (size_t) x * channels + c;
        if (depth == 16)     s[c] = ((unsigned int) row[2 * k] << 8) | row[2 * k + 1];
        else if (depth == 8) s[c] = row[k];
        else                 s[c] = (row[k * depth / 8] >> (8 - depth - (k * depth) % 8)) & ((1u << depth) - 1u);
      }
      float* px = dst + 4 * x;
      if (colour == 3)
      {
        const unsigned int idx = (s[0] * 3 + 2 < palette.size()) ? s[0] : 0;
        px[0] = palette[idx * 3] / 255.0f; px[1] = palette[idx * 3 + 1] / 255.0f; px[2] = palette[idx * 3 + 2] / 255.0f;
        px[3] = (idx < paletteAlpha.size()) ? paletteAlpha[idx] / 255.0f : 1.0f;
      }
      else if (colour == 0) { px[0] = px[1] = px[2] = (float) s[0] / maxValue; }
      else if (colour == 4) { px[0] = px[1] = px[2] = (float) s[0] / maxValue; px[3] = (float) s[1] / maxValue; }
      else { px[0] = (float) s[0] / maxValue; px[1] = (float) s[1] / maxValue; px[2] = (float) s[2] / maxValue; if (colour == 6) px[3] = (float) s[3] / maxValue; }
    }
  }
  return true;
}

bool decodeHdr(const std::vector<unsigned char>& file, int& width, int& height, std::vector<float>& rgba, std::string& error)
{
  // header lines up to the empty line, then the resolution line
  size_t pos = 0;
  auto line = [&](std::string& out) { out.clear(); while (pos < file.size() && file[pos] != '\n') out.push_back((char) file[pos++]); if (pos < file.size()) ++pos; };
  std::string s;
  line(s);
  if (s.compare(0, 2, "#?") != 0) { error = "HDR: missing #? signature"; return false; }
  for (;;) { if (pos >= file.size()) { error = "HDR: truncated header"; return false; } line(s); if (s.empty()) break; }
  line(s);
  if (sscanf(s.c_str(), "-Y %d +X %d", &height, &width) != 2 || width <= 0 || height <= 0) { error = "HDR: only the standard orientation '-Y h +X w' is supported"; return false; }

  rgba.assign((size_t) width * height * 4, 1.0f);
  std::vector<unsigned char> scan((size_t) width * 4);
  for (int y = 0; y < height; ++y)
  {
    bool rle = false;
    if (width >= 8 && width <= 32767 && pos + 4 <= file.size() && file[pos] == 2 && file[pos + 1] == 2 && !(file[pos + 2] & 0x80))
    {
      if ((((int) file[pos + 2] << 8) | file[pos + 3]) != width) { error = "HDR: scanline width mismatch"; return false; }
      rle = true; pos += 4;
    }
    if (rle)
    {
      for (int c = 0; c < 4; ++c)
      {
        int x = 0;
        while (x < width)
        {
          if (pos >= file.size()) { error = "HDR: truncated scanline"; return false; }
          int count = file[pos++];
          if (count > 128)
          {
            count -= 128;
            if (x + count > width || pos >= file.size()) { error = "HDR: bad run"; return false; }
            const unsigned char v = file[pos++];
            for (int i = 0; i < count; ++i) scan[(size_t) 4 * (x++) + c] = v;
          }
          else
          {
            if (count == 0 || x + count > width || pos + (size_t) count > file.size()) { error = "HDR: bad literal run"; return false; }
            for (int i = 0; i < count; ++i) scan[(size_t) 4 * (x++) + c] = file[pos++];
          }
        }
      }
    }
    else
    {
      if (pos + scan.size() > file.size()) { error = "HDR: truncated pixel data"; return false; }
      memcpy(scan.data(), &file[pos], scan.size());
      pos += scan.size();
    }
    float* dst = &rgba[(size_t) 4 * width * (size_t) (height - 1 - y)];
    for (int x = 0; x < width; ++x)
    {
      const unsigned char* p = &scan[(size_t) 4 * x];
      const float f = (p[3] == 0) ? 0.0f : ldexpf(1.0f, (int) p[3] - (128 + 8)); // mantissa * 2^(e - 136), the common RGBE readers' convention
      dst[4 * x + 0] = p[0] * f; dst[4 * x + 1] = p[1] * f; dst[4 * x + 2] = p[2] * f;
    }
  }
  return true;
}

bool decodePfm(const std::vector<unsigned char>& file, int& width, int& height, std::vector<float>& rgba, std::string& error)
{
  int channels = (file.size() > 1 && file[1] == 'F') ? 3 : 1;
  int consumed = 0;
  float scaleField = 0.0f;
  const std::string head(reinterpret_cast<const char*>(file.data()), std::min<size_t>(file.size(), 128));
  if (sscanf(head.c_str() + 2, "%d %d %f%n", &width, &height, &scaleField, &consumed) != 3 || width <= 0 || height <= 0 || scaleField == 0.0f) { error = "PFM: bad header"; return false; }
  size_t pos = 2 + (size_t) consumed + 1; // one whitespace byte after the scale
  const size_t need = (size_t) width * height * channels * 4;
  if (pos + need > file.size()) { error = "PFM: truncated pixel data"; return false; }
  const bool littleEndian = scaleField < 0.0f;
  rgba.assign((size_t) width * height * 4, 1.0f);
  for (size_t i = 0; i < (size_t) width * height; ++i) // PFM rows run bottom to top, like the buffer
  {
    for (int c = 0; c < channels; ++c)
    {
      unsigned char b[4];
      memcpy(b, &file[pos + (i * channels + c) * 4], 4);
      if (!littleEndian) { std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
      float v; memcpy(&v, b, 4);
      if (channels == 1) { rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = v; } else rgba[4 * i + c] = v;
    }
  }
  return true;
}

} // namespace

bool loadImageRgba32f(const std::string& path, int& width, int& height, std::vector<float>& rgba, std::string& error)
{
  std::vector<unsigned char> file;
  if (!readFile(path, file)) { error = "cannot read " + path; return false; }
  static const unsigned char pngSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  bool ok = false;
  if (file.size() >= 8 && !memcmp(file.data(), pngSignature, 8)) ok = decodePng(file, width, height, rgba, error);
  else if (file.size() >= 2 && file[0] == '#' && file[1] == '?') ok = decodeHdr(file, width, height, rgba, error);
  else if (file.size() >= 2 && file[0] == 'P' && (file[1] == 'F' || file[1] == 'f')) ok = decodePfm(file, width, height, rgba, error);
  else if (file.size() >= 2 && file[0] == 0xff && file[1] == 0xd8) error = "JPEG decoding is not part of this build (convert the picture to PNG)";
  else error = "unknown image format (PNG, Radiance HDR and PFM are supported)";
  if (!ok) error = path + ": " + error;
  return ok;
}

} // namespace twk
