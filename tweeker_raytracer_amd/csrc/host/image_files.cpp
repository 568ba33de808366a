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

bool loadJpegRgba32f(const std::vector<unsigned char>& file, int& width, int& height, std::vector<float>& rgba, std::string& error);

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
  // IHDR is untrusted: deflate expands at most ~1032:1, so a picture whose raw size exceeds that bound of its IDAT
  // bytes cannot be valid — refuse before allocating width x height from the header alone
  if ((rowBytes + 1) * (size_t) height > idat.size() * 1032 + 65536 || (size_t) width * (size_t) height > ((size_t) 1 << 31))
  { error = "PNG: image dimensions do not fit the compressed data"; return false; }
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

  // untrusted header: a run-length scanline of w pixels takes at least 4 + 4 * ceil(w / 127) * 2 bytes, a flat one 4 w —
  // a picture with more rows than remaining bytes / 8 cannot be complete
  if ((size_t) height > (file.size() - pos) / 8 + 1 || (size_t) width > ((size_t) 1 << 24)) { error = "HDR: resolution does not fit the file"; return false; }
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
  else if (file.size() >= 2 && file[0] == 0xff && file[1] == 0xd8) ok = loadJpegRgba32f(file, width, height, rgba, error);
  else error = "unknown image format (PNG, baseline JPEG, Radiance HDR and PFM are supported)";
  if (!ok) error = path + ": " + error;
  return ok;
}

} // namespace twk

// ---- baseline JPEG -------------------------------------------------------------------------------
// Application::createPictures loads "./NVIDIA_Logo.jpg" through DevIL, i.e. through libjpeg with its default
// decompression parameters. This reader restates that decode path for baseline (SOF0 / SOF1 Huffman, 8-bit) files:
// the accurate integer inverse DCT (jidctint.c "islow"), "fancy" triangle-filter chroma upsampling for 2x1 and 2x2
// subsampling (jdsample.c) and the fixed-point YCbCr → RGB tables (jdcolor.c), so that the texels are the ones a
// libjpeg-based loader hands out, byte for byte (tests compare against Pillow's libjpeg). Progressive and arithmetic
// coded files are reported as unsupported.
namespace twk {

namespace {

struct JpegHuffman
{
  unsigned char bits[17] = {0};
  unsigned char values[256] = {0};
  int mincode[17], maxcode[18], valptr[17];
  bool present = false;
  void build()
  {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l)
    {
      valptr[l] = k; mincode[l] = code;
      code += bits[l]; k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};

struct JpegComponent
{
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int blocksPerLine = 0, blocksPerColumn = 0; // padded to whole MCUs
  int width = 0, height = 0;                  // real (downsampled) size
  std::vector<unsigned char> samples;         // blocksPerLine*8 x blocksPerColumn*8
  int dcPred = 0;
};

struct JpegBitReader
{
  const unsigned char* data; size_t size, pos;
  unsigned int buffer = 0; int count = 0; bool marker = false;
  int bit()
  {
    if (count == 0)
    {
      unsigned int b = 0;
      if (!marker && pos < size)
      {
        b = data[pos++];
        if (b == 0xff)
        {
          const unsigned int next = (pos < size) ? data[pos] : 0xd9;
          if (next == 0) ++pos;                  // stuffed zero
          else { marker = true; --pos; b = 0; }  // a marker ends the entropy-coded segment: feed zeros
        }
      }
      buffer = b; count = 8;
    }
    --count;
    return (int) ((buffer >> count) & 1u);
  }
  int receive(int n) { int v = 0; for (int i = 0; i < n; ++i) v = (v << 1) | bit(); return v; }
  void reset() { count = 0; buffer = 0; marker = false; }
};

int jpegDecodeSymbol(JpegBitReader& br, const JpegHuffman& h)
{
  int code = 0;
  for (int l = 1; l <= 16; ++l)
  {
    code = (code << 1) | br.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.values[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}

int jpegExtend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

const unsigned char jpegZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline int jpegDescale(long x, int n) { return (int) ((x + (1L << (n - 1))) >> n); }
inline unsigned char jpegClamp(int v) { return (unsigned char) (v < 0 ? 0 : (v > 255 ? 255 : v)); }

// jidctint.c jpeg_idct_islow: coefficients in natural order, already multiplied by the quantisation table.
void jpegIdctIslow(const int* in, unsigned char* out, int stride)
{
  const int CONST_BITS = 13, PASS1_BITS = 2;
  const long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
             F2053 = 16819, F2562 = 20995, F3072 = 25172;
  long ws[64];
  for (int pass = 0; pass < 2; ++pass)
  {
    for (int i = 0; i < 8; ++i)
    {
      long s[8];
      for (int k = 0; k < 8; ++k) s[k] = (pass == 0) ? (long) in[8 * k + i] : ws[8 * i + k];
      long z2 = s[2], z3 = s[6];
      long z1 = (z2 + z3) * F0541;
      long tmp2 = z1 + z3 * (-F1847);
      long tmp3 = z1 + z2 * F0765;
      long tmp0 = (s[0] + s[4]) << CONST_BITS;
      long tmp1 = (s[0] - s[4]) << CONST_BITS;
      const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = s[7]; tmp1 = s[5]; tmp2 = s[3]; tmp3 = s[1];
      z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
      const long z5 = (z3 + z4) * F1175;
      tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
      z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
      z3 += z5; z4 += z5;
      tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
      const long r[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
      if (pass == 0) for (int k = 0; k < 8; ++k) ws[8 * k + i] = jpegDescale(r[k], CONST_BITS - PASS1_BITS);
      else           for (int k = 0; k < 8; ++k) out[stride * i + k] = jpegClamp(jpegDescale(r[k], CONST_BITS + PASS1_BITS + 3) + 128);
    }
  }
}

bool decodeJpeg(const std::vector<unsigned char>& file, int& width, int& height, std::vector<float>& rgba, std::string& error)
{
  unsigned short qt[4][64] = {{0}};
  JpegHuffman dcTables[4], acTables[4];
  std::vector<JpegComponent> comps;
  int restartInterval = 0, hmax = 1, vmax = 1, adobeTransform = -1;
  bool haveFrame = false;
  size_t pos = 2;
  auto need = [&](size_t n) { return pos + n <= file.size(); };
  for (;;)
  {
    if (!need(2) || file[pos] != 0xff) { error = "JPEG: marker expected"; return false; }
    while (need(1) && file[pos] == 0xff) ++pos; // fill bytes
    if (!need(1)) { error = "JPEG: truncated"; return false; }
    const int m = file[pos++];
    if (m == 0xd9) { error = "JPEG: no scan"; return false; }
    if (!need(2)) { error = "JPEG: truncated"; return false; }
    const size_t len = ((size_t) file[pos] << 8) | file[pos + 1];
    if (len < 2 || !need(len)) { error = "JPEG: bad segment length"; return false; }
    const unsigned char* seg = &file[pos + 2];
    const size_t n = len - 2;
    if (m == 0xdb) // DQT
    {
      size_t k = 0;
      while (k < n)
      {
        const int pq = seg[k] >> 4, tq = seg[k] & 15; ++k;
        if (tq > 3 || k + (pq ? 128 : 64) > n) { error = "JPEG: bad quantisation table"; return false; }
        for (int i = 0; i < 64; ++i) { qt[tq][jpegZigzag[i]] = pq ? (unsigned short) ((seg[k] << 8) | seg[k + 1]) : seg[k]; k += pq ? 2 : 1; }
      }
    }
    else if (m == 0xc4) // DHT
    {
      size_t k = 0;
      while (k + 17 <= n)
      {
        const int tc = seg[k] >> 4, th = seg[k] & 15; ++k;
        if (tc > 1 || th > 3) { error = "JPEG: bad Huffman table id"; return false; }
        JpegHuffman& h = tc ? acTables[th] : dcTables[th];
        int total = 0;
        for (int l = 1; l <= 16; ++l) { h.bits[l] = seg[k++]; total += h.bits[l]; }
        if (total > 256 || k + (size_t) total > n) { error = "JPEG: bad Huffman table"; return false; }
        for (int i = 0; i < total; ++i) h.values[i] = seg[k++];
        h.build(); h.present = true;
      }
    }
    else if (m == 0xc0 || m == 0xc1) // SOF0 / SOF1: baseline / extended sequential, Huffman
    {
      if (n < 6 || seg[0] != 8) { error = "JPEG: only 8-bit samples are supported"; return false; }
      height = (seg[1] << 8) | seg[2]; width = (seg[3] << 8) | seg[4];
      const int nc = seg[5];
      if (width <= 0 || height <= 0 || (nc != 1 && nc != 3) || n < 6 + 3 * (size_t) nc) { error = "JPEG: unsupported frame (greyscale and three-component files only)"; return false; }
      comps.resize((size_t) nc);
      for (int c = 0; c < nc; ++c)
      {
        comps[(size_t) c].id = seg[6 + 3 * c]; comps[(size_t) c].h = seg[7 + 3 * c] >> 4; comps[(size_t) c].v = seg[7 + 3 * c] & 15; comps[(size_t) c].tq = seg[8 + 3 * c] & 3;
        hmax = std::max(hmax, comps[(size_t) c].h); vmax = std::max(vmax, comps[(size_t) c].v);
      }
      haveFrame = true;
    }
    else if (m == 0xc2 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) { error = "JPEG: progressive, lossless and arithmetic-coded files are not supported (baseline only)"; return false; }
    else if (m == 0xdd && n >= 2) restartInterval = (seg[0] << 8) | seg[1];
    else if (m == 0xee && n >= 12 && !memcmp(seg, "Adobe", 5)) adobeTransform = seg[11];
    else if (m == 0xda) // SOS
    {
      if (!haveFrame) { error = "JPEG: scan before frame"; return false; }
      const int ns = seg[0];
      if (ns != (int) comps.size() || n < 1 + 2 * (size_t) ns + 3) { error = "JPEG: only single-scan (interleaved) files are supported"; return false; }
      for (int i = 0; i < ns; ++i)
      {
        JpegComponent* c = nullptr;
        for (JpegComponent& k : comps) if (k.id == seg[1 + 2 * i]) c = &k;
        if (!c) { error = "JPEG: scan names an unknown component"; return false; }
        c->td = seg[2 + 2 * i] >> 4; c->ta = seg[2 + 2 * i] & 15;
        if (c->td > 3 || c->ta > 3 || !dcTables[c->td].present || !acTables[c->ta].present) { error = "JPEG: missing Huffman table"; return false; }
      }
      pos += len;
      break;
    }
    pos += len;
  }

  // component geometry
  if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; } // a single-component scan is not interleaved
  for (const JpegComponent& c : comps)
  {
    const bool full = (c.h == hmax && c.v == vmax), h2v1 = (c.h * 2 == hmax && c.v == vmax), h2v2 = (c.h * 2 == hmax && c.v * 2 == vmax);
    if (c.h < 1 || c.v < 1 || !(full || h2v1 || h2v2)) { error = "JPEG: unsupported chroma subsampling (1x1, 2x1 and 2x2 only)"; return false; }
  }
  const int mcuW = 8 * hmax, mcuH = 8 * vmax;
  const int mcusX = (width + mcuW - 1) / mcuW, mcusY = (height + mcuH - 1) / mcuH;
  for (JpegComponent& c : comps)
  {
    c.blocksPerLine = mcusX * c.h; c.blocksPerColumn = mcusY * c.v;
    c.width = (width * c.h + hmax - 1) / hmax; c.height = (height * c.v + vmax - 1) / vmax;
    c.samples.assign((size_t) c.blocksPerLine * 8 * c.blocksPerColumn * 8, 0);
  }

  // entropy-coded data
  JpegBitReader br{file.data(), file.size(), pos};
  int restartsLeft = restartInterval;
  for (int my = 0; my < mcusY; ++my)
  {
    for (int mx = 0; mx < mcusX; ++mx)
    {
      if (restartInterval && restartsLeft == 0)
      {
        // byte-align, expect RSTn
        br.reset();
        while (br.pos + 1 < br.size && !(br.data[br.pos] == 0xff && br.data[br.pos + 1] >= 0xd0 && br.data[br.pos + 1] <= 0xd7)) ++br.pos;
        if (br.pos + 1 >= br.size) { error = "JPEG: restart marker missing"; return false; }
        br.pos += 2;
        for (JpegComponent& c : comps) c.dcPred = 0;
        restartsLeft = restartInterval;
      }
      for (JpegComponent& c : comps)
      {
        for (int by = 0; by < c.v; ++by)
        {
          for (int bx = 0; bx < c.h; ++bx)
          {
            int coef[64] = {0};
            const int s = jpegDecodeSymbol(br, dcTables[c.td]);
            if (s < 0 || s > 11) { error = "JPEG: corrupt data (DC)"; return false; }
            c.dcPred += jpegExtend(br.receive(s), s);
            coef[0] = c.dcPred * qt[c.tq][0];
            for (int k = 1; k < 64;)
            {
              const int rs = jpegDecodeSymbol(br, acTables[c.ta]);
              if (rs < 0) { error = "JPEG: corrupt data (AC)"; return false; }
              const int r = rs >> 4, sz = rs & 15;
              if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
              k += r;
              if (k > 63) { error = "JPEG: corrupt data (run)"; return false; }
              coef[jpegZigzag[k]] = jpegExtend(br.receive(sz), sz) * qt[c.tq][jpegZigzag[k]];
              ++k;
            }
            const int blockX = mx * c.h + bx, blockY = my * c.v + by;
            jpegIdctIslow(coef, &c.samples[((size_t) blockY * 8) * ((size_t) c.blocksPerLine * 8) + (size_t) blockX * 8], c.blocksPerLine * 8);
          }
        }
      }
      if (restartInterval) --restartsLeft;
    }
  }

  // upsample the chroma planes to full resolution (jdsample.c, do_fancy_upsampling)
  auto plane = [&](const JpegComponent& c, std::vector<unsigned char>& out)
  {
    out.assign((size_t) width * height, 0);
    const int stride = c.blocksPerLine * 8;
    auto row = [&](int y) { y = std::max(0, std::min(c.height - 1, y)); return &c.samples[(size_t) y * stride]; };
    const int hs = hmax / c.h, vs = vmax / c.v;
    if (hs == 1 && vs == 1)
    {
      for (int y = 0; y < height; ++y) memcpy(&out[(size_t) y * width], row(y), (size_t) width);
    }
    else if (hs == 2 && vs == 1) // h2v1_fancy_upsample
    {
      const int n = c.width;
      std::vector<unsigned char> line((size_t) 2 * n + 2);
      for (int y = 0; y < height; ++y)
      {
        const unsigned char* in = row(y);
        if (n == 1) { line[0] = line[1] = in[0]; }
        else
        {
          line[0] = in[0]; line[1] = (unsigned char) ((in[0] * 3 + in[1] + 2) >> 2);
          for (int i = 1; i < n - 1; ++i)
          {
            line[2 * i]     = (unsigned char) ((in[i] * 3 + in[i - 1] + 1) >> 2);
            line[2 * i + 1] = (unsigned char) ((in[i] * 3 + in[i + 1] + 2) >> 2);
          }
          line[2 * n - 2] = (unsigned char) ((in[n - 1] * 3 + in[n - 2] + 1) >> 2); line[2 * n - 1] = in[n - 1];
        }
        memcpy(&out[(size_t) y * width], line.data(), (size_t) width);
      }
    }
    else // h2v2_fancy_upsample
    {
      const int n = c.width;
      std::vector<unsigned char> line((size_t) 2 * n + 2);
      for (int y = 0; y < height; ++y)
      {
        const int iy = y / 2;
        const unsigned char* in0 = row(iy);
        const unsigned char* in1 = row((y & 1) ? iy + 1 : iy - 1); // nearer neighbour row: above for the upper output row
        if (n == 1)
        {
          const int sum = in0[0] * 3 + in1[0];
          line[0] = (unsigned char) ((sum * 4 + 8) >> 4); line[1] = (unsigned char) ((sum * 4 + 7) >> 4);
        }
        else
        {
          int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
          line[0] = (unsigned char) ((thiscol * 4 + 8) >> 4); line[1] = (unsigned char) ((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol; thiscol = nextcol;
          for (int i = 1; i < n - 1; ++i)
          {
            nextcol = in0[i + 1] * 3 + in1[i + 1];
            line[2 * i]     = (unsigned char) ((thiscol * 3 + lastcol + 8) >> 4);
            line[2 * i + 1] = (unsigned char) ((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
          }
          line[2 * n - 2] = (unsigned char) ((thiscol * 3 + lastcol + 8) >> 4); line[2 * n - 1] = (unsigned char) ((thiscol * 4 + 7) >> 4);
        }
        memcpy(&out[(size_t) y * width], line.data(), (size_t) width);
      }
    }
  };

  rgba.assign((size_t) width * height * 4, 1.0f);
  std::vector<unsigned char> p0, p1, p2;
  plane(comps[0], p0);
  if (comps.size() == 3) { plane(comps[1], p1); plane(comps[2], p2); }
  for (int y = 0; y < height; ++y)
  {
    float* dst = &rgba[(size_t) 4 * width * (size_t) (height - 1 - y)];
    for (int x = 0; x < width; ++x)
    {
      const size_t i = (size_t) y * width + x;
      int r, g, b;
      if (comps.size() == 1) { r = g = b = p0[i]; }
      else if (adobeTransform == 0) { r = p0[i]; g = p1[i]; b = p2[i]; }
      else // jdcolor.c ycc_rgb_convert, 16-bit fixed point
      {
        const int yy = p0[i], cb = p1[i] - 128, cr = p2[i] - 128;
        r = jpegClamp(yy + (int) ((91881L * cr + 32768L) >> 16));
        g = jpegClamp(yy + (int) ((-22554L * cb + 32768L + -46802L * cr) >> 16));
        b = jpegClamp(yy + (int) ((116130L * cb + 32768L) >> 16));
      }
      dst[4 * x + 0] = (float) r / 255.0f; dst[4 * x + 1] = (float) g / 255.0f; dst[4 * x + 2] = (float) b / 255.0f;
    }
  }
  return true;
}

} // namespace

bool loadJpegRgba32f(const std::vector<unsigned char>& file, int& width, int& height, std::vector<float>& rgba, std::string& error)
{
  return decodeJpeg(file, width, height, rgba, error);
}

} // namespace twk
