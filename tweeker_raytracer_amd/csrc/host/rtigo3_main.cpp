// rtigo3_hip: command-line front end over libtweeker_hip.so for the batch path of the reference's rtigo3
//   rtigo3 -s system.txt -d scene.txt -m 1
// (main.cpp:169-172 → Application::benchmark, Application.cpp:491-531): render samplesSqrt² iterations, wait for the
// device, print "<iterations> / <seconds> = <fps> fps", store the tonemapped screenshot. Options as Options.cpp:44-156.
// The interactive mode (-m 0: GLFW window, imgui) needs a display and is not part of this build.
//
// Multi-GPU: `strategy` > 0 in the system description renders with every visible device selected by `devicesMask`
// (Raytracer.cpp:60-123), each device its checkerboard share into a local buffer (DeviceMultiGPULocalCopy.cpp), then
// ONE peer copy per device and ONE compositor launch on the first device. TWK_CLI_VIRTUAL_DEVICES=N shares one
// physical GPU between N handles (testing on a single-GPU machine).
#include "tweeker_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Options // Options.cpp:33-38 defaults
{
  int width = 512, height = 512, mode = 0;
  std::string system, scene;
};

void printUsage(const std::string& argv0)
{
  std::cerr << "\nUsage: " << argv0 << " [options]\n"
            << "App Options:\n"
               "   ? | help | --help       Print this usage message and exit.\n"
               "  -w | --width <int>       Width of the client window  (512), unused without a display\n"
               "  -h | --height <int>      Height of the client window (512), unused without a display\n"
               "  -m | --mode <int>        0 = interactive (not available in this build), 1 == benchmark (0)\n"
               "  -s | --system <filename> Filename for system options (empty).\n"
               "  -d | --desc   <filename> Filename for scene description (empty).\n"
            << std::endl;
}

bool parseCommandLine(int argc, char* argv[], Options& o)
{
  for (int i = 1; i < argc; ++i)
  {
    const std::string arg(argv[i]);
    if (arg == "?" || arg == "help" || arg == "--help") { printUsage(argv[0]); return false; }
    const bool known = arg == "-w" || arg == "--width" || arg == "-h" || arg == "--height" || arg == "-m" || arg == "--mode" ||
                       arg == "-s" || arg == "--system" || arg == "-d" || arg == "--desc";
    if (!known) { std::cerr << "Unknown option '" << arg << "'\n"; printUsage(argv[0]); return false; }
    if (i == argc - 1) { std::cerr << "Option '" << arg << "' requires additional argument.\n"; printUsage(argv[0]); return false; }
    const char* value = argv[++i];
    if      (arg == "-w" || arg == "--width")  o.width  = atoi(value);
    else if (arg == "-h" || arg == "--height") o.height = atoi(value);
    else if (arg == "-m" || arg == "--mode")   o.mode   = atoi(value);
    else if (arg == "-s" || arg == "--system") o.system = value;
    else                                       o.scene  = value;
  }
  return true;
}

// ≙ Application::createPictures (Application.cpp:679-699) + Raytracer::initTextures: the two hard-coded material
// pictures and, for miss 2, the environment map named by "envMap". A picture that cannot be read is reported and
// skipped (materials that ask for it then render untextured); a progressive JPEG is not decodable here, so the
// albedo picture is also looked up as ./NVIDIA_Logo.png.
struct PictureFile { int slot; std::vector<std::string> candidates; };

bool loadPicture(const PictureFile& picture, int& width, int& height, std::vector<float>& rgba)
{
  for (const std::string& path : picture.candidates)
  {
    if (twk_load_image(path.c_str(), &width, &height, nullptr, 0) != TWK_SUCCESS) continue;
    rgba.resize((size_t) width * height * 4);
    if (twk_load_image(path.c_str(), &width, &height, rgba.data(), rgba.size()) == TWK_SUCCESS)
    {
      std::cerr << "INFO: picture " << path << " " << width << " x " << height << std::endl;
      return true;
    }
  }
  std::cerr << "WARNING: picture " << picture.candidates.front() << " not loaded: " << twk_last_error() << std::endl;
  return false;
}

#define TWK_OK(call) do { if ((call) != TWK_SUCCESS) { std::cerr << "ERROR: " << #call << ": " << twk_last_error() << std::endl; return 1; } } while (0)
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::cerr << "ERROR: " << #call << ": " << hipGetErrorString(e_) << std::endl; return 1; } } while (0)

} // namespace

int main(int argc, char* argv[])
{
  Options options;
  if (!parseCommandLine(argc, argv, options)) return 1;
  if (options.system.empty() || options.scene.empty())
  {
    std::cerr << "ERROR: system (-s) and scene (-d) description files are required.\n";
    printUsage(argv[0]);
    return 1;
  }
  if (std::max(0, options.mode) != 1)
  {
    std::cerr << "ERROR: mode 0 (interactive) needs a GLFW window; this build runs the benchmark mode only: -m 1\n";
    return 1;
  }

  TwkApp app = nullptr;
  TWK_OK(twk_app_create(&app, options.system.c_str(), options.scene.c_str()));
  TwkAppInfo info;
  TWK_OK(twk_app_info(app, &info));

  // device selection (Raytracer.cpp:60-123): strategy 0 = first visible device, otherwise all visible devices in the mask
  int visible = 0;
  TWK_OK(twk_device_count(&visible));
  std::vector<int> ordinals;
  const char* virtualDevices = getenv("TWK_CLI_VIRTUAL_DEVICES");
  if (info.strategy == 0) ordinals.push_back(0);
  else if (virtualDevices && atoi(virtualDevices) > 0) ordinals.assign((size_t) std::min(32, atoi(virtualDevices)), 0);
  else
  {
    for (int d = 0; d < visible && d < 32; ++d) if (info.devicesMask & (1 << d)) ordinals.push_back(d);
  }
  if (ordinals.empty()) { std::cerr << "ERROR: no device selected by devicesMask " << info.devicesMask << " (" << visible << " visible)\n"; return 1; }
  const int count = (int) ordinals.size();

  std::vector<TwkDevice> devices((size_t) count, nullptr);
  TwkDeviceState state;
  TWK_OK(twk_app_get_state(app, &state));
  for (int i = 0; i < count; ++i) TWK_OK(twk_device_create(&devices[(size_t) i], ordinals[(size_t) i], i, count, info.miss));

  std::vector<PictureFile> pictures = {{TWK_TEXTURE_ALBEDO, {"./NVIDIA_Logo.jpg", "./NVIDIA_Logo.png"}}, {TWK_TEXTURE_CUTOUT, {"./slots_alpha.png"}}};
  char environment[4096];
  TWK_OK(twk_app_get_environment(app, environment, sizeof(environment)));
  if (info.miss == 2 && environment[0] != 0) pictures.push_back({TWK_TEXTURE_ENVIRONMENT, {environment}});
  for (const PictureFile& picture : pictures)
  {
    int w = 0, h = 0;
    std::vector<float> rgba;
    if (!loadPicture(picture, w, h, rgba)) continue;
    for (int i = 0; i < count; ++i) TWK_OK(twk_init_texture(devices[(size_t) i], picture.slot, rgba.data(), w, h));
  }
  for (int i = 0; i < count; ++i) TWK_OK(twk_app_init_device(app, devices[(size_t) i]));

  // Buffer strategy (Raytracer.cpp:125-176 picks the Device flavour): 1 = zero copy — one pinned host frame mapped into
  // every device (DeviceMultiGPUZeroCopy.cpp:106-118); 2 = peer access — one frame on the first device, written by its
  // peers (DeviceMultiGPUPeerAccess.cpp:110-158); 3 = local copy — packed tile buffers + compositor.
  void* sharedFrame = nullptr;
  const size_t frameBytes = (size_t) info.resolution[0] * info.resolution[1] * 16;
  if (count > 1 && info.strategy == 1)
  {
    HIP_OK(hipHostMalloc(&sharedFrame, frameBytes, hipHostMallocPortable | hipHostMallocMapped));
    memset(sharedFrame, 0, frameBytes);
  }
  else if (count > 1 && info.strategy == 2)
  {
    HIP_OK(hipSetDevice(ordinals[0]));
    HIP_OK(hipMalloc(&sharedFrame, frameBytes));
    HIP_OK(hipMemset(sharedFrame, 0, frameBytes));
    for (int i = 1; i < count; ++i)
    {
      if (ordinals[(size_t) i] == ordinals[0]) continue;
      int canAccess = 0;
      HIP_OK(hipDeviceCanAccessPeer(&canAccess, ordinals[(size_t) i], ordinals[0]));
      if (!canAccess) { std::cerr << "ERROR: device " << ordinals[(size_t) i] << " cannot access the frame on device " << ordinals[0] << " (strategy 2 needs one peer-to-peer island)\n"; return 1; }
      HIP_OK(hipSetDevice(ordinals[(size_t) i]));
      const hipError_t e = hipDeviceEnablePeerAccess(ordinals[0], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { std::cerr << "ERROR: hipDeviceEnablePeerAccess: " << hipGetErrorString(e) << std::endl; return 1; }
      (void) hipGetLastError();
    }
  }
  if (sharedFrame) for (int i = 0; i < count; ++i) TWK_OK(twk_set_shared_frame(devices[(size_t) i], sharedFrame, frameBytes));
  std::cerr << "INFO: " << count << " device(s), " << info.resolution[0] << " x " << info.resolution[1] << ", "
            << info.samplesSqrt * info.samplesSqrt << " spp, " << info.numInstances << " instances" << std::endl;

  // Application::benchmark (Application.cpp:491-513)
  const unsigned int spp = (unsigned int) (info.samplesSqrt * info.samplesSqrt);
  const auto start = std::chrono::steady_clock::now();
  unsigned int iterationIndex = 0;
  while (iterationIndex < spp)
  {
    for (int i = 0; i < count; ++i) TWK_OK(twk_launch(devices[(size_t) i], iterationIndex));
    ++iterationIndex;
  }
  for (int i = 0; i < count; ++i) TWK_OK(twk_sync(devices[(size_t) i]));
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
  const double fps = double(iterationIndex) / seconds;
  {
    std::ostringstream stream;
    stream.precision(3);
    stream << std::fixed << iterationIndex << " / " << seconds << " = " << fps << " fps";
    std::cout << stream.str() << std::endl;
  }

  // screenshot(true) (Application.cpp:525,2231-2335)
  const int width = info.resolution[0], height = info.resolution[1];
  const size_t numPixels = (size_t) width * height;
  TwkTonemapper tonemapper;
  TWK_OK(twk_app_get_tonemapper(app, &tonemapper));
  std::vector<unsigned char> rgb8(numPixels * 3);
  if (count == 1)
  {
    TWK_OK(twk_tonemap(devices[0], &tonemapper, nullptr, numPixels, rgb8.data()));
  }
  else if (sharedFrame)
  {
    TWK_OK(twk_tonemap(devices[0], &tonemapper, sharedFrame, numPixels, rgb8.data())); // every device wrote its pixels straight into the frame
  }
  else
  {
    int launchWidth = 0;
    TWK_OK(twk_get_launch_width(devices[0], &launchWidth));
    const size_t tileBytes = (size_t) launchWidth * height * 16;
    void* tiles = nullptr; void* full = nullptr;
    HIP_OK(hipSetDevice(ordinals[0]));
    HIP_OK(hipMalloc(&tiles, tileBytes * count));
    HIP_OK(hipMalloc(&full, numPixels * 16));
    for (int i = 0; i < count; ++i)
    {
      void* src = nullptr; size_t bytes = 0;
      TWK_OK(twk_get_output_device_pointer(devices[(size_t) i], &src, &bytes));
      HIP_OK(hipMemcpyPeer(static_cast<char*>(tiles) + tileBytes * i, ordinals[0], src, ordinals[(size_t) i], tileBytes));
    }
    // device-to-device copies return before they have finished and the handle's stream is non-blocking: wait here,
    // or the compositor reads tiles that are still in flight
    HIP_OK(hipDeviceSynchronize());
    TWK_OK(twk_compositor(devices[0], tiles, full));
    TWK_OK(twk_tonemap(devices[0], &tonemapper, full, numPixels, rgb8.data()));
    HIP_OK(hipFree(tiles)); HIP_OK(hipFree(full));
  }
  char path[4096];
  TWK_OK(twk_app_screenshot_path(app, 1, path, sizeof(path)));
  TWK_OK(twk_write_png_rgb8(path, width, height, rgb8.data(), 1));
  std::cout << path << std::endl;

  for (int i = 0; i < count; ++i) TWK_OK(twk_device_destroy(devices[(size_t) i]));
  if (sharedFrame) { if (info.strategy == 1) (void) hipHostFree(sharedFrame); else (void) hipFree(sharedFrame); }
  TWK_OK(twk_app_destroy(app));
  return 0;
}
