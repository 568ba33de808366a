// PNG / Radiance HDR writers for the screenshot path (Application.cpp:2231-2335), see image_files.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace twk {

bool writePngRgb8(const std::string& path, int width, int height, const unsigned char* rgb8, bool bottomUp, std::string& error);
bool writeHdrRgba32f(const std::string& path, int width, int height, const float* rgba, bool bottomUp, std::string& error);
// PNG / Radiance HDR / PFM → RGBA32F, row 0 = bottom row (what Picture::load + Texture::create* hand to the device).
bool loadImageRgba32f(const std::string& path, int& width, int& height, std::vector<float>& rgba, std::string& error);
void floatToRgbe(float r, float g, float b, unsigned char rgbe[4]);

} // namespace twk
