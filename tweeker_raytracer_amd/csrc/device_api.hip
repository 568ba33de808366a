// The per-GPU renderer behind the C ABI of include/tweeker_hip.h (≙ rtigo3's Device family,
// reference apps/rtigo3/src/Device.cpp + DeviceSingleGPU.cpp + DeviceMultiGPULocalCopy.cpp).
// Host code here only moves data and enqueues kernels; nothing is ever computed on the CPU in place
// of a kernel. Without a HIP device every entry point that needs one fails.
#include "device_types.h"
#include "bvh_build.h"
#include "error_state.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------
static thread_local std::string g_lastError;

int twkSetError(int code, const std::string& message)
{
  g_lastError = message;
  return code;
}

namespace twk {
void launchTrace(const LaunchParams& p, int depth, bool count, bool primary, int gridBlocks, hipStream_t stream);
void launchTraceQuery(const LaunchParams& p, const float* rays, unsigned int numRays, int anyHit, float* tBetaGamma, int* ids, int gridBlocks, hipStream_t stream);
void launchGenerate(const LaunchParams& p, hipStream_t stream);
void launchShade(const LaunchParams& p, int depth, bool primary, int gridBlocks, hipStream_t stream);
void launchTileEntries(const LaunchParams& p, const float4* topTable, int tilesX, int tilesY, int4* out, hipStream_t stream);
void launchAccumulate(const LaunchParams& p, hipStream_t stream);
void launchCompositor(const float4* tiles, float4* output, int width, int height, int launchWidth, int deviceCount,
                      int tileSizeX, int tileShiftX, int tileShiftY, hipStream_t stream);
void launchMathTap(int op, const float* x, const float* y, float* out, size_t n, hipStream_t stream);
void launchStreamCopy(const float4* src, float4* dst, size_t n, hipStream_t stream);
void launchGatherProbeFill(float4* table, size_t count, unsigned int lines, hipStream_t stream);
void launchGatherProbe(const float4* table, unsigned int lines, int steps, float* out, int gridBlocks, hipStream_t stream);
void launchTonemap(const float4* hdr, unsigned char* ldr, size_t numPixels, const TwkTonemapper& tm, hipStream_t stream);
}

using namespace twk;

#define HIP_TRY(call)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return twkSetError((e_ == hipErrorOutOfMemory) ? TWK_ERROR_OUT_OF_MEMORY : TWK_ERROR_HIP,    \
                         std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

static_assert(sizeof(TwkCameraDefinition) == 48, "CameraDefinition is 48 B (camera_definition.h:34-40)");
static_assert(sizeof(TwkLightDefinition) == 80, "LightDefinition is 80 B (light_definition.h:42-59)");
static_assert(sizeof(TwkTriangleAttributes) == 48, "TriangleAttributes is 48 B (vertex_attributes.h:34-40)");
static_assert(sizeof(DevLight) == sizeof(TwkLightDefinition), "device light layout");
static_assert(sizeof(DevMaterial) == 64, "MaterialDefinition is 64 B (material_definition.h:37-56)");
static_assert(sizeof(DevInstance) == 128, "instance record");
static_assert(sizeof(BvhNode) == 64, "BVH2 node");

struct GeometryHost
{
  std::vector<TwkTriangleAttributes> attributes;
  std::vector<unsigned int>          indices;
  unsigned int attributeBase = 0, indexBase = 0;
  int triangleBase = 0, nodeBase = 0, numTriangles = 0;
  float rootBounds[6];
};

struct InstanceHost
{
  int   geometry;
  float transform[12];
  int   material, light;
};

struct TimedLaunch { hipEvent_t start, stop; int kind; };
// What a table of primary-ray entry points (trace_kernels.hip tileEntryKernel) was built for: camera bits, frame, tree, tile
// distribution. Compared with memcmp (as floats, TWK_NODE_CACHED | 1 rounded to TWK_NODE_CACHED | 0, a build serial
// stopped counting at 2^24, and -0.0 compared equal to +0.0).
struct TileEntriesKey
{
  uint32_t camera[12];
  int32_t  resolution[2], topRoot, topRoot2, launchWidth, deviceCount, deviceIndex, tileSize[2], valid;
  uint64_t buildSerial;
};
#define TWK_MAX_LANES 4
#define TWK_LANE_QUEUE_PAD (TWK_QUEUE_SEGMENTS * 1024)                       // slots a lane's queue arrays may use beyond its path count (segment gaps: K x (255 + 512) at most)
#define TWK_STREAM_PAD ((size_t) TWK_MAX_LANES * (TWK_LANE_QUEUE_PAD + 1024)) // ... of all lanes + the rounding of their shares
#define TWK_STATS_WORDS 192 // device words of TwkLaunchStats: [0, 24) traversal + shade totals, [24, 96) the shade phases (shade_device.h PhaseScope); twice: the time view's scratch copy
#define TWK_COUNTER_WORDS (TWK_COUNTERS_PER_DEPTH * (TWK_MAX_DEPTH + 2)) // one lane's counter block

struct TwkDevice_t
{
  int ordinal = 0, index = 0, count = 1, miss = 1;
  int numCUs = 256;
  hipStream_t stream = nullptr;

  TwkDeviceState state;
  bool stateSet = false;
  int  launchWidth = 1;

  std::vector<TwkCameraDefinition> cameras;
  std::vector<TwkLightDefinition>  lights;
  std::vector<DevMaterial>         materials;
  std::vector<GeometryHost>        geometries;
  std::vector<InstanceHost>        instances;
  bool built = false;
  bool twoLevel = true;                 // some instance is entered through the top level (else the soup is the whole scene)
  int  flattenMaxTriangles = TWK_FLATTEN_TRIANGLES, flattenMaxReferences = TWK_FLATTEN_REFERENCES;
  int  maxInstanceMaterial = -1, maxInstanceLight = -1; // largest indices the built scene's instances use
  TwkBuildInfo buildInfo;

  // device memory
  float* d_camera = nullptr;
  DevLight* d_lights = nullptr;
  DevMaterial* d_materials = nullptr; int materialCapacity = 0;
  float* d_attributes = nullptr; unsigned int* d_indices = nullptr;
  BvhNode* d_nodes = nullptr; BvhNode* d_wideNodes = nullptr /* build-time only: full-precision wide nodes, freed once quantised */; float4* d_wideQ = nullptr; float4* d_triangles = nullptr; float4* d_shadeTriangles = nullptr; DevInstance* d_instances = nullptr;
  bool directSmallLeaves = true; /* TWK_DIRECT_SMALL_LEAVES=0: A/B */ bool costedCuts = true; /* TWK_COSTED_CUTS=0: A/B */ bool fusedPrimary = true; /* TWK_FUSED_PRIMARY=0: A/B */ bool tileEntries = true; /* TWK_TILE_ENTRIES=0: A/B */ bool wideRoot = true; /* TWK_WIDE_ROOT=0: A/B */ int wideRoot1 = 0, wideRoot2 = TWK_BVH_SENTINEL; size_t wideNodesTotal = 0;
  int4* d_tileEntries = nullptr; size_t tileEntriesCapacity = 0; TileEntriesKey tileEntriesKey = {}; uint64_t buildSerial = 0; // entry points of the primary rays (trace_kernels.hip tileEntryKernel) and what they were built for
  float4* d_topNodes = nullptr; float4* d_topNodes7 = nullptr; bool topCache = true; int traceWavesForced = 0 /* TWK_TRACE_WAVES_RUNTIME: 6 or 7, 0 = by scene */; // TWK_TOP_CACHE=0 turns the LDS top-of-tree cache off (A/B)
  float4* d_texels[3] = {nullptr, nullptr, nullptr};
  float* d_envCDF_U = nullptr; float* d_envCDF_V = nullptr;
  int tlasRoot = 0;
  size_t totalNodes = 0, totalTriangles = 0;

  // per-resolution streams
  int allocatedPixels = 0;
  void* d_streamBlock = nullptr; // one allocation carved into the SoA streams
  float4* d_outputInternal = nullptr;
  float4* d_outputExternal = nullptr; size_t outputExternalBytes = 0;
  bool outputFrame = false; // the external buffer is a shared full frame (twk_set_shared_frame)
  unsigned int* d_counters = nullptr;
  unsigned long long* d_stats = nullptr;
  unsigned int* h_dropped = nullptr; unsigned int* d_dropped = nullptr; // pinned + device-mapped: LaunchParams::droppedPushes
  int* d_spill = nullptr; size_t spillLanes = 0;
  bool packedQueue = true; // TWK_PACKED_QUEUE=0: A/B
  int shadeSort = 1;       // TWK_SHADE_SORT=0: slot order (A/B); 1: class order in every launch but the first of a pass; 2: in the first too
  float4* d_firstHit = nullptr; int* d_firstHitInstance = nullptr;
  // denoiser AOVs (Optix7Gui raygeneration.cu:125-164): per-path values of a pass and their running means per launch index
  bool aovEnabled = false; int shaderVariant = TWK_SHADERS_RTIGO3;
  bool nextEventEstimation = true, debugExceptions = false; // twk_set_next_event_estimation / twk_set_debug_exceptions (≙ shaders/config.h:50-56)
  bool timeView = false; float* d_pathTime = nullptr; int timePaths = 0; // twk_set_time_view
  float4* d_pathAlbedo = nullptr; float4* d_pathNormal = nullptr; int aovPaths = 0;
  float4* d_aovAlbedo = nullptr; float4* d_aovNormal = nullptr; int aovPixels = 0;
  bool captureFirstHits = false;
  bool statsEnabled = false;
  bool profileEnabled = false;
  std::vector<TimedLaunch> timed; size_t timedUsed = 0;
  float profileMs[TWK_KERNEL_COUNT] = {0, 0, 0, 0, 0};
  int   profileLaunches[TWK_KERNEL_COUNT] = {0, 0, 0, 0, 0};
  // Deferred launches: twk_launch only records the iteration; consecutive iterations are rendered together as one
  // wavefront pass of up to batchMax samples per pixel when the batch is full or anything observes the device.
  int   batchMax = 64;
  size_t streamBudgetBytes = 0; // TWK_STREAM_BUDGET_MB: cap on the path streams of one pass (0 = device memory is the limit)
  unsigned int pendingFirst = 0;
  int   pendingCount = 0;
  int   allocatedPaths = 0;

  // Pass lanes: a wavefront pass may be cut into up to TWK_MAX_LANES independent sub-passes over disjoint path ranges, each a
  // chain of generate / trace / shade launches on a stream of its own (lane 0 = `stream`), joined before the accumulate
  // kernel. Small passes are bound by the dependent-launch chain and the longest ray of each launch, not by throughput:
  // two chains side by side fill each other's gaps (renderPass).
  hipStream_t laneStream[TWK_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t  laneDone[TWK_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t  laneFork = nullptr;
  int   lanesForced = 0; // TWK_PASS_LANES: 0 = choose by pass size
  int   laneTraceWaves = 0; // TWK_LANE_TRACE_WAVES: trace blocks per CU of each lane (0 = TWK_TRACE_WAVES / lanes)

  std::vector<std::vector<char>> hostScene; // twk_debug_snapshot_scene: host copies of the scene arrays

  LaunchParams params;
  BvhBuilder builder;
};

static int flushPending(TwkDevice dev);

static int activate(TwkDevice dev, const char* where, bool flush = true)
{
  if (!dev) return twkSetError(TWK_ERROR_INVALID_VALUE, std::string(where) + ": NULL device handle");
  HIP_TRY(hipSetDevice(dev->ordinal));
  if (flush && dev->pendingCount > 0) return flushPending(dev); // recorded launches run before anything else touches the device
  return TWK_SUCCESS;
}

template<typename T> static void freeDevice(T*& p) { if (p) { (void) hipFree(p); p = nullptr; } }

// Scratch device allocation of one call; freed on every return path.
template<typename T> struct ScopedDeviceBuffer
{
  T* ptr = nullptr;
  ~ScopedDeviceBuffer() { if (ptr) (void) hipFree(ptr); }
  hipError_t allocate(size_t count) { return hipMalloc(&ptr, count * sizeof(T)); }
};

// Inverse of a row-major 3x4 affine matrix in double, rounded once (OptiX derives the same matrix for
// optixGetInstanceInverseTransformFromHandle, closesthit.cu:49-52).
static void invertAffine(const float m[12], float inv[12])
{
  const double a00 = m[0], a01 = m[1], a02 = m[2],  t0 = m[3];
  const double a10 = m[4], a11 = m[5], a12 = m[6],  t1 = m[7];
  const double a20 = m[8], a21 = m[9], a22 = m[10], t2 = m[11];
  const double c00 = a11 * a22 - a12 * a21;
  const double c01 = a12 * a20 - a10 * a22;
  const double c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double r = 1.0 / det;
  const double i00 = c00 * r, i01 = (a02 * a21 - a01 * a22) * r, i02 = (a01 * a12 - a02 * a11) * r;
  const double i10 = c01 * r, i11 = (a00 * a22 - a02 * a20) * r, i12 = (a02 * a10 - a00 * a12) * r;
  const double i20 = c02 * r, i21 = (a01 * a20 - a00 * a21) * r, i22 = (a00 * a11 - a01 * a10) * r;
  inv[0] = (float) i00; inv[1] = (float) i01; inv[2]  = (float) i02; inv[3]  = (float) -(i00 * t0 + i01 * t1 + i02 * t2);
  inv[4] = (float) i10; inv[5] = (float) i11; inv[6]  = (float) i12; inv[7]  = (float) -(i10 * t0 + i11 * t1 + i12 * t2);
  inv[8] = (float) i20; inv[9] = (float) i21; inv[10] = (float) i22; inv[11] = (float) -(i20 * t0 + i21 * t1 + i22 * t2);
}

// MaterialGUI → MaterialDefinition, Device.cpp:1022-1050
static DevMaterial convertMaterial(const TwkMaterialGUI& g)
{
  DevMaterial m;
  memset(&m, 0, sizeof(m));
  m.textureAlbedo = g.useAlbedoTexture ? (TWK_TEXTURE_ALBEDO + 1) : 0;
  m.textureCutout = g.useCutoutTexture ? (TWK_TEXTURE_CUTOUT + 1) : 0;
  m.roughness[0] = g.roughness[0]; m.roughness[1] = g.roughness[1];
  m.indexBSDF = g.indexBSDF;
  m.albedo[0] = g.albedo[0]; m.albedo[1] = g.albedo[1]; m.albedo[2] = g.albedo[2];
  m.absorption[0] = m.absorption[1] = m.absorption[2] = 0.0f;
  if (0.0f < g.absorptionScale)
  {
    const float x = -logf(fmax(0.0001f, g.absorptionColor[0]));
    const float y = -logf(fmax(0.0001f, g.absorptionColor[1]));
    const float z = -logf(fmax(0.0001f, g.absorptionColor[2]));
    m.absorption[0] = x * g.absorptionScale; m.absorption[1] = y * g.absorptionScale; m.absorption[2] = z * g.absorptionScale;
  }
  m.ior   = g.ior;
  m.flags = g.thinwalled ? TWK_FLAG_THINWALLED : 0u;
  return m;
}

static int calculateShift(int size) // Device.cpp:1172-1189
{
  int s = 0;
  while (s < 32 && (size & (1 << s)) == 0) ++s;
  return s;
}

static void refreshParams(TwkDevice dev)
{
  LaunchParams& p = dev->params;
  p.nodes = dev->d_nodes; p.wideQ = dev->d_wideQ; p.triangles = dev->d_triangles; p.shadeTriangles = dev->d_shadeTriangles; p.instances = dev->d_instances;
  p.attributes = dev->d_attributes; p.indices = dev->d_indices;
  p.materials = dev->d_materials; p.lights = dev->d_lights; p.camera = dev->d_camera;
  p.tlasRoot = dev->tlasRoot;
  p.topNodes = dev->d_topNodes; p.topNodes7 = dev->d_topNodes7;
  p.topRoot = dev->topCache ? (TWK_NODE_CACHED | 0) : dev->wideRoot1;
  p.topRoot2 = (dev->wideRoot2 == TWK_BVH_SENTINEL) ? TWK_BVH_SENTINEL : (dev->topCache ? (TWK_NODE_CACHED | 1) : dev->wideRoot2);
  p.twoLevel = dev->twoLevel ? 1 : 0;
  p.numInstances = (int) dev->instances.size();
  p.numLights = (int) dev->lights.size();
  p.numMaterials = (int) dev->materials.size();
  p.miss = dev->miss;
  p.hasCutout = 0; p.hasAlbedoTexture = 0;
  for (const DevMaterial& m : dev->materials) { if (m.textureCutout != 0) p.hasCutout = 1; if (m.textureAlbedo != 0) p.hasAlbedoTexture = 1; }
  // seven trace blocks per CU where the variant that fits them applies (device_types.h TWK_TRACE_WAVES7)
  p.traceWaves = (!dev->twoLevel && (!p.hasCutout || TWK_TRACE_CUTOUT_SEVEN) && dev->totalNodes <= (size_t) TWK_TRACE_WAVES7_MAX_NODES) ? TWK_TRACE_WAVES7 : (p.hasCutout ? (dev->twoLevel ? TWK_TRACE_WAVES_CUTOUT_OTHER : TWK_TRACE_WAVES) : TWK_TRACE_WAVES);
  if (dev->traceWavesForced == TWK_TRACE_WAVES || (dev->traceWavesForced == TWK_TRACE_WAVES7 && !dev->twoLevel)) p.traceWaves = dev->traceWavesForced;
  p.envCDF_U = dev->d_envCDF_U; p.envCDF_V = dev->d_envCDF_V;
  for (int k = 0; k < 2; ++k)
  {
    p.resolution[k]  = dev->state.resolution[k];
    p.tileSize[k]    = dev->state.tileSize[k];
    p.tileShift[k]   = calculateShift(dev->state.tileSize[k]);
    p.pathLengths[k] = dev->state.pathLengths[k];
  }
  p.deviceCount = dev->count; p.deviceIndex = dev->index; p.distribution = dev->state.distribution;
  p.launchWidth = dev->launchWidth;
  p.lensShader  = dev->state.lensShader;
  p.sceneEpsilon = dev->state.epsilonFactor * SCENE_EPSILON_SCALE;
  p.envRotation  = dev->state.envRotation;
  p.numPixels = dev->launchWidth * dev->state.resolution[1];
  p.batchCount = 1;
  p.numPaths = p.numPixels;
  p.queueStride = TWK_QUEUE_STRIDE(p.numPaths);
  p.pathBase = 0;
  p.output = dev->d_outputExternal ? dev->d_outputExternal : dev->d_outputInternal;
  p.outputFrame = (dev->d_outputExternal && dev->outputFrame) ? 1 : 0;
  p.counters = dev->d_counters;
  // the time view runs the measurement builds of the kernels, which tally: into a scratch block unless statistics are on
  p.stats = dev->statsEnabled ? dev->d_stats : (dev->timeView ? dev->d_stats + TWK_STATS_WORDS / 2 : nullptr);
  p.pathTime = dev->timeView ? dev->d_pathTime : nullptr; p.clockScale = dev->state.clockFactor * 1.0e-9f; // Device.h:350 CLOCK_FACTOR_SCALE
  p.shaderVariant = dev->shaderVariant; p.shadeSort = dev->shadeSort;
  p.nextEventEstimation = dev->nextEventEstimation ? 1 : 0; p.debugExceptions = dev->debugExceptions ? 1 : 0;
  p.pathAlbedo = dev->aovEnabled ? dev->d_pathAlbedo : nullptr; p.pathNormal = dev->aovEnabled ? dev->d_pathNormal : nullptr;
  p.aovAlbedo  = dev->aovEnabled ? dev->d_aovAlbedo : nullptr;  p.aovNormal  = dev->aovEnabled ? dev->d_aovNormal : nullptr;
  p.firstHit = dev->captureFirstHits ? dev->d_firstHit : nullptr;
  p.firstHitInstance = dev->captureFirstHits ? dev->d_firstHitInstance : nullptr;
  p.traceStackSpill = dev->d_spill;
  p.droppedPushes = dev->d_dropped;
  p.packedQueue = 0; // renderPass decides per pass
}

static int traceGridBlocks(TwkDevice dev) { return dev->numCUs * TWK_TRACE_WAVES7; } // the larger of the two persistent grids (sizes the spill stacks); a launch uses numCUs x params.traceWaves

// `samples`: samples per pixel the next wavefront pass carries; the path streams grow to what passes actually need
// (a 64-sample pass of a 1920x1080 frame takes 46 GB, a handle that renders two iterations takes 1.4 GB).
static int ensureStreams(TwkDevice dev, int samples = 1)
{
  const size_t wantPixels = (size_t) dev->launchWidth * (size_t) dev->state.resolution[1];
  const size_t wantPaths  = wantPixels * (size_t) (samples > 1 ? samples : 1);
  if (wantPaths >= ((size_t) 1 << 31)) // paths and queue slots are 32-bit indices; reported like a failed allocation so that the pass is cut in halves
    return twkSetError(TWK_ERROR_OUT_OF_MEMORY, "a pass of " + std::to_string(wantPaths) + " paths exceeds the 2^31 path indices of a wavefront pass");
  const int numPixels = (int) wantPixels;
  const int numPaths  = (int) wantPaths;
  if (numPixels > dev->allocatedPixels || dev->d_outputInternal == nullptr)
  {
    freeDevice(dev->d_outputInternal);
    freeDevice(dev->d_firstHit); freeDevice(dev->d_firstHitInstance);
    const size_t n = (size_t) numPixels;
    HIP_TRY(hipMalloc(&dev->d_outputInternal, n * sizeof(float4)));
    HIP_TRY(hipMemsetAsync(dev->d_outputInternal, 0, n * sizeof(float4), dev->stream));
    HIP_TRY(hipMalloc(&dev->d_firstHit, n * sizeof(float4)));
    HIP_TRY(hipMalloc(&dev->d_firstHitInstance, n * sizeof(int)));
    dev->allocatedPixels = numPixels;
  }
  if (numPaths > dev->allocatedPaths || dev->d_streamBlock == nullptr)
  {
    freeDevice(dev->d_streamBlock);
    const size_t n = (size_t) numPaths;
    // float4 streams: rayOrg[2], rayDir[2], rayThroughput[2], hitRecord, shadowOrg, shadowDir, shadowPending, radiance, volumeStack[4] = 15
    // 8-byte: raySeedFlags[2]; 4-byte: rayPixel[2], hitInstance, shadowPixel, overflowSlots[2]   → 280 bytes per path
    // (+ TWK_STREAM_PAD slots per stream: the segments of a queue leave gaps between them, device_types.h "queue segments")
    const size_t bytes = (n + TWK_STREAM_PAD) * (15 * sizeof(float4) + 2 * sizeof(uint2) + 6 * sizeof(unsigned int)) + 4096;
    if (dev->streamBudgetBytes != 0 && n * (15 * sizeof(float4) + 2 * sizeof(uint2) + 6 * sizeof(unsigned int)) > dev->streamBudgetBytes) // (the budget is for what grows with the pass, not for the fixed padding)
      return twkSetError(TWK_ERROR_OUT_OF_MEMORY, "path streams of " + std::to_string(bytes >> 20) + " MiB exceed TWK_STREAM_BUDGET_MB");
    HIP_TRY(hipMalloc(&dev->d_streamBlock, bytes));
    dev->allocatedPaths = numPaths;
  }
  if (dev->timeView && dev->allocatedPaths > dev->timePaths)
  {
    freeDevice(dev->d_pathTime); dev->timePaths = 0;
    HIP_TRY(hipMalloc(&dev->d_pathTime, (size_t) dev->allocatedPaths * sizeof(float)));
    dev->timePaths = dev->allocatedPaths;
  }
  if (dev->aovEnabled)
  {
    if (dev->allocatedPaths > dev->aovPaths)
    {
      freeDevice(dev->d_pathAlbedo); freeDevice(dev->d_pathNormal); dev->aovPaths = 0;
      HIP_TRY(hipMalloc(&dev->d_pathAlbedo, (size_t) dev->allocatedPaths * sizeof(float4)));
      HIP_TRY(hipMalloc(&dev->d_pathNormal, (size_t) dev->allocatedPaths * sizeof(float4)));
      dev->aovPaths = dev->allocatedPaths;
    }
    if (dev->allocatedPixels > dev->aovPixels)
    {
      freeDevice(dev->d_aovAlbedo); freeDevice(dev->d_aovNormal); dev->aovPixels = 0;
      const size_t bytes = (size_t) dev->allocatedPixels * sizeof(float4);
      HIP_TRY(hipMalloc(&dev->d_aovAlbedo, bytes));
      HIP_TRY(hipMalloc(&dev->d_aovNormal, bytes));
      HIP_TRY(hipMemsetAsync(dev->d_aovAlbedo, 0, bytes, dev->stream));
      HIP_TRY(hipMemsetAsync(dev->d_aovNormal, 0, bytes, dev->stream));
      dev->aovPixels = dev->allocatedPixels;
    }
  }
  if (!dev->d_counters) HIP_TRY(hipMalloc(&dev->d_counters, sizeof(unsigned int) * TWK_COUNTER_WORDS * TWK_MAX_LANES));
  if (!dev->d_stats) { HIP_TRY(hipMalloc(&dev->d_stats, sizeof(unsigned long long) * TWK_STATS_WORDS)); HIP_TRY(hipMemsetAsync(dev->d_stats, 0, sizeof(unsigned long long) * TWK_STATS_WORDS, dev->stream)); } // TwkLaunchStats words (24 + the shade phases' 3 x 24) + a scratch block of the same size for the time view
  if (!dev->h_dropped)
  {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&dev->h_dropped), sizeof(unsigned int), hipHostMallocMapped));
    *dev->h_dropped = 0u;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev->d_dropped), dev->h_dropped, 0));
  }
  // one full grid of per-lane spill stacks (the lanes of a pass share it); two only under the experiments-only
  // TWK_LANE_TRACE_WAVES knob, which may give every lane more than its share of a grid (264 MB per handle otherwise)
  const size_t lanes = (size_t) (dev->laneTraceWaves > 0 ? 2 : 1) * traceGridBlocks(dev) * TWK_TRACE_BLOCK;
  if (lanes > dev->spillLanes)
  {
    freeDevice(dev->d_spill);
    HIP_TRY(hipMalloc(&dev->d_spill, lanes * TWK_TRACE_STACK_SPILL * sizeof(int)));
    dev->spillLanes = lanes;
  }

  // carve the block
  LaunchParams& p = dev->params;
  const size_t n = (size_t) dev->allocatedPaths;
  char* base = static_cast<char*>(dev->d_streamBlock);
  auto take = [&](size_t elemBytes) { char* r = base; base += (n + TWK_STREAM_PAD) * elemBytes; return r; };
  p.rayOrg[0] = (float4*) take(16); p.rayOrg[1] = (float4*) take(16);
  p.rayDir[0] = (float4*) take(16); p.rayDir[1] = (float4*) take(16);
  p.hitRecord = (float4*) take(16);
  p.shadowOrg = (float4*) take(16); p.shadowDir = (float4*) take(16); p.shadowPending = (float4*) take(16);
  p.rayThroughput[0] = (float4*) take(16); p.rayThroughput[1] = (float4*) take(16); p.pathRadiance = (float4*) take(16);
  p.volumeStack = (float4*) take(64);
  p.raySeedFlags[0] = (uint2*) take(8); p.raySeedFlags[1] = (uint2*) take(8);
  p.rayPixel[0] = (unsigned int*) take(4); p.rayPixel[1] = (unsigned int*) take(4);
  p.hitInstance = (int*) take(4);
  p.shadowPixel = (unsigned int*) take(4);
  p.overflowSlots = (unsigned int*) take(8);
  return TWK_SUCCESS;
}

static void timedLaunchBegin(TwkDevice dev, int kind, hipStream_t stream)
{
  if (!dev->profileEnabled) return;
  if (dev->timedUsed == dev->timed.size())
  {
    TimedLaunch t; t.kind = kind;
    if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess) { dev->profileEnabled = false; return; }
    dev->timed.push_back(t);
  }
  dev->timed[dev->timedUsed].kind = kind;
  (void) hipEventRecord(dev->timed[dev->timedUsed].start, stream);
}

static void timedLaunchEnd(TwkDevice dev, hipStream_t stream)
{
  if (!dev->profileEnabled) return;
  (void) hipEventRecord(dev->timed[dev->timedUsed].stop, stream);
  dev->timedUsed++;
}

static int collectTimed(TwkDevice dev)
{
  if (dev->timedUsed == 0) return TWK_SUCCESS;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  for (size_t i = 0; i < dev->timedUsed; ++i)
  {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, dev->timed[i].start, dev->timed[i].stop));
    dev->profileMs[dev->timed[i].kind] += ms;
    dev->profileLaunches[dev->timed[i].kind] += 1;
  }
  dev->timedUsed = 0;
  return TWK_SUCCESS;
}

// After a stream synchronisation: did any traversal lose a push (trace_device.h TWK_PUSH)? Cannot happen on a scene
// twk_build accepted; if it does, the image is wrong and the caller must hear about it, statistics on or off.
static int checkDroppedPushes(TwkDevice dev, const char* where)
{
  if (!dev->h_dropped || *dev->h_dropped == 0u) return TWK_SUCCESS;
  const unsigned int n = *dev->h_dropped;
  *dev->h_dropped = 0u;
  return twkSetError(TWK_ERROR_INVALID_STATE, std::string(where) + ": " + std::to_string(n) + " traversal stack pushes were dropped (tree deeper than the " +
                     std::to_string(TWK_TRACE_STACK_LDS + TWK_TRACE_STACK_SPILL) + "-entry stacks): the image is incomplete");
}

// Spherical environment CDFs + integral, Texture.cpp:1499-1645.
static float gaussianFilter(const float* rgba, unsigned int width, unsigned int height, unsigned int x, unsigned int y)
{
  const unsigned int left   = (0 < x)          ? x - 1 : width - 1;
  const unsigned int right  = (x < width - 1)  ? x + 1 : 0;
  const unsigned int bottom = (0 < y)          ? y - 1 : y;
  const unsigned int top    = (y < height - 1) ? y + 1 : y;
  auto sum3 = [&](unsigned int xx, unsigned int yy) { const float* q = rgba + ((size_t) width * yy + xx) * 4; return q[0] + q[1] + q[2]; };
  float intensity = sum3(x, y) * 0.619347f;
  float f = sum3(x, bottom);
  f += sum3(left, y);
  f += sum3(right, y);
  f += sum3(x, top);
  intensity += f * 0.0838195f;
  f  = sum3(left, bottom);
  f += sum3(right, bottom);
  f += sum3(left, top);
  f += sum3(right, top);
  intensity += f * 0.0113437f;
  return intensity / 3.0f;
}

static void calculateSphericalCDF(const float* rgba, unsigned int width, unsigned int height,
                                  std::vector<float>& cdfU, std::vector<float>& cdfV, float& integralOut)
{
  std::vector<float> funcU((size_t) width * height), funcV(height + 1);
  float sum = 0.0f;
  for (unsigned int y = 0; y < height; ++y)
  {
    const float sinTheta = float(sin(M_PI * (double(y) + 0.5) / double(height)));
    for (unsigned int x = 0; x < width; ++x)
    {
      const float value = gaussianFilter(rgba, width, height, x, y);
      funcU[(size_t) y * width + x] = value * sinTheta;
      const float* q = rgba + ((size_t) y * width + x) * 4;
      const float intensity = (q[0] + q[1] + q[2]) / 3.0f;
      sum += intensity * sinTheta;
    }
  }
  integralOut = sum * 2.0f * kPi * kPi / float(width * height);

  cdfU.assign((size_t) (width + 1) * height, 0.0f);
  cdfV.assign(height + 1, 0.0f);
  for (unsigned int y = 0; y < height; ++y)
  {
    const size_t row = (size_t) y * (width + 1);
    cdfU[row] = 0.0f;
    for (unsigned int x = 1; x <= width; ++x) cdfU[row + x] = cdfU[row + x - 1] + funcU[(size_t) y * width + x - 1];
    const float integral = cdfU[row + width];
    funcV[y] = integral;
    if (integral != 0.0f) { for (unsigned int x = 1; x <= width; ++x) cdfU[row + x] /= integral; }
    else                  { for (unsigned int x = 1; x <= width; ++x) cdfU[row + x] = float(x) / float(width); }
  }
  cdfV[0] = 0.0f;
  for (unsigned int y = 1; y <= height; ++y) cdfV[y] = cdfV[y - 1] + funcV[y - 1];
  const float integral = cdfV[height];
  if (integral != 0.0f) { for (unsigned int y = 1; y <= height; ++y) cdfV[y] /= integral; }
  else                  { for (unsigned int y = 1; y <= height; ++y) cdfV[y] = float(y) / float(height); }
}

static int renderPass(TwkDevice dev, unsigned int firstIteration, int count);

static int flushPending(TwkDevice dev)
{
  // The streams of a pass take 280 bytes per pixel and iteration. When they do not fit in device memory the pass is
  // cut in halves until they do (the image does not depend on how iterations are grouped) and the handle keeps the
  // smaller limit; only an allocation failure for a single iteration is reported.
  unsigned int first = dev->pendingFirst;
  int left = dev->pendingCount;
  dev->pendingCount = 0;
  while (left > 0)
  {
    int count = (left < dev->batchMax) ? left : dev->batchMax;
    if (count < 1) count = 1;
    for (;;)
    {
      const int rc = ensureStreams(dev, count);
      if (rc == TWK_SUCCESS) break;
      (void) hipGetLastError(); // the failed allocation must not surface again at the next launch check
      if (rc != TWK_ERROR_OUT_OF_MEMORY || count == 1) return rc;
      count = (count + 1) / 2;
      dev->batchMax = count;
    }
    const int rc = renderPass(dev, first, count);
    if (rc) return rc;
    first += (unsigned int) count;
    left -= count;
  }
  return TWK_SUCCESS;
}

// How many lanes a pass of `numPaths` paths is cut into (TWK_PASS_LANES forces a count). Measured on C2, DESIGN.md 2.
static int chooseLanes(TwkDevice dev, int numPaths)
{
  int lanes = 1;
  if (dev->lanesForced > 0) lanes = dev->lanesForced;
  else if (numPaths <= TWK_LANES2_MAX_PATHS) lanes = 2;
  if (dev->captureFirstHits) lanes = 1; // debug capture indexes by launch index
  // twk_profile_enable: the per-kind sums of twk_profile_get are sums of launch durations; the launches of two lanes overlap
  // in time, so their sum would be about twice the wall time of the kind. A profiled pass runs as ONE lane.
  if (dev->profileEnabled) lanes = 1;
  while (lanes > 1 && numPaths / lanes < 4096) --lanes;
  return std::min(lanes, TWK_MAX_LANES);
}

// The launch parameters of lane `lane` of `lanes`: paths [base, base + count) of the pass, every per-path and per-slot
// stream offset to the lane's own range, a counter block and a slice of the traversal spill stacks of its own.
static LaunchParams laneParams(TwkDevice dev, const LaunchParams& p, int lane, int lanes, int traceBlocks)
{
  if (lanes == 1) return p;
  LaunchParams q = p;
  const size_t total = (size_t) p.numPaths;
  const size_t share = ((total + lanes - 1) / lanes + 1023) & ~(size_t) 1023;
  const size_t base = std::min(total, share * lane), count = std::min(total - base, share);
  q.pathBase = (int) base; q.numPaths = (int) count;
  q.queueStride = TWK_QUEUE_STRIDE(count);
  // the queue arrays of a lane reach beyond its path count (the gaps between a queue's segments): their bases leave room for that
  const size_t queueBase = (size_t) lane * (share + TWK_LANE_QUEUE_PAD);
  for (int k = 0; k < 2; ++k)
  {
    q.rayOrg[k] += queueBase; q.rayDir[k] += queueBase; q.rayPixel[k] += queueBase; q.rayThroughput[k] += queueBase; q.raySeedFlags[k] += queueBase;
  }
  q.hitRecord += base; q.hitInstance += base;
  q.shadowOrg += queueBase; q.shadowDir += queueBase; q.shadowPixel += queueBase; q.shadowPending += queueBase;
  q.pathRadiance += base; q.overflowSlots += 2 * base;
  q.volumeStack += 4 * base; // [4][numPaths of the lane], indexed level * numPaths + path (shade_device.h)
  if (q.pathAlbedo) { q.pathAlbedo += base; q.pathNormal += base; }
  if (q.pathTime) q.pathTime += base;
  q.counters = dev->d_counters + (size_t) lane * TWK_COUNTER_WORDS;
  q.traceStackSpill = dev->d_spill + (size_t) lane * traceBlocks * TWK_TRACE_BLOCK * TWK_TRACE_STACK_SPILL;
  return q;
}

// Runs iterations [firstIteration, firstIteration + count) as one wavefront pass; the streams are allocated.
static int renderPass(TwkDevice dev, unsigned int firstIteration, int count)
{
  refreshParams(dev);
  LaunchParams& p = dev->params;
  p.iterationIndex = firstIteration;
  p.batchCount = count;
  p.numPaths = p.numPixels * p.batchCount;
  p.pathBase = 0;
  p.queueStride = TWK_QUEUE_STRIDE(p.numPaths);

  // launch index + path flags and the LCG state in the constant words of the queued rays (device_types.h LaunchParams::packedQueue)
  p.packedQueue = (dev->packedQueue && !p.hasCutout && (unsigned int) p.numPaths <= TWK_PACKED_PIXEL_MASK) ? 1 : 0;
  const int maxDepth = dev->state.pathLengths[1];
  const int lanes = chooseLanes(dev, p.numPaths);
  // every block of every lane's persistent trace kernel resident at once: the lanes share the CUs' block slots
  int traceWaves = std::max(1, p.traceWaves / lanes);
  if (lanes > 1 && dev->laneTraceWaves > 0) traceWaves = std::min(dev->laneTraceWaves, 2 * TWK_TRACE_WAVES / lanes); // TWK_LANE_TRACE_WAVES (experiments; the spill stacks hold two full grids)
  const int traceGrid = dev->numCUs * traceWaves;

  // Every bounce runs as a per-depth trace / shade launch pair over compacted queues (the persistent tail kernel for the deep
  // bounces, rounds 1-4, lives in tools/experiments/r04_tail_kernel.patch: no faster at any launch size measured).
  const int wavefrontDepth = maxDepth;

  // Primary rays are computed by the first traversal and the first shade launch instead of being written by generateKernel and
  // read back (shade_kernels.hip "primary rays") — unless the paths have no bounce to be shaded in.
  const bool fusedPrimary = dev->fusedPrimary && wavefrontDepth >= 1;
  // ... and start at their tile's entry points (trace_kernels.hip tileEntryKernel): pinhole camera; launch index = pixel, or a
  // tile distribution whose tiles are whole entry tiles.
  // The lists depend on camera, frame and tree; rebuilt (one small kernel) when any of those changed since they were made.
  p.tileEntries = nullptr; p.tilesX = 0;
  const bool distributed = p.distribution && 1 < p.deviceCount;
  const bool tilesAlign = !distributed || (p.tileSize[0] % TWK_ENTRY_TILE == 0 && p.tileSize[1] % TWK_ENTRY_TILE == 0); // a distribution tile = whole entry tiles
  if (fusedPrimary && dev->tileEntries && p.lensShader == 0 && tilesAlign && (distributed || p.launchWidth == p.resolution[0]) && !dev->cameras.empty())
  {
    const int tilesX = (p.launchWidth + TWK_ENTRY_TILE - 1) / TWK_ENTRY_TILE, tilesY = (p.resolution[1] + TWK_ENTRY_TILE - 1) / TWK_ENTRY_TILE;
    const size_t need = (size_t) tilesX * tilesY * 2;
    TileEntriesKey key;
    memset(&key, 0, sizeof(key)); // padding bytes too: the keys are compared with memcmp
    memcpy(key.camera, &dev->cameras[0], sizeof(key.camera));
    key.resolution[0] = p.resolution[0]; key.resolution[1] = p.resolution[1]; key.topRoot = p.topRoot; key.topRoot2 = p.topRoot2;
    key.launchWidth = p.launchWidth; key.deviceCount = distributed ? p.deviceCount : 1; key.deviceIndex = p.deviceIndex;
    key.tileSize[0] = p.tileSize[0]; key.tileSize[1] = p.tileSize[1]; key.valid = 1; key.buildSerial = dev->buildSerial;
    if (need > dev->tileEntriesCapacity)
    {
      freeDevice(dev->d_tileEntries); dev->tileEntriesCapacity = 0; memset(&dev->tileEntriesKey, 0, sizeof(dev->tileEntriesKey));
      HIP_TRY(hipMalloc(&dev->d_tileEntries, sizeof(int4) * need));
      dev->tileEntriesCapacity = need;
    }
    if (memcmp(&key, &dev->tileEntriesKey, sizeof(key)) != 0)
    {
      // the table the PRIMARY build of the traversal kernel caches: TWK_PRIMARY_SIX -> the six-block build's
      const float4* topTable = (TWK_PRIMARY_SIX || p.traceWaves != TWK_TRACE_WAVES7) ? p.topNodes : p.topNodes7;
      launchTileEntries(p, topTable, tilesX, tilesY, dev->d_tileEntries, dev->stream);
      HIP_TRY(hipGetLastError());
      dev->tileEntriesKey = key;
    }
    p.tileEntries = dev->d_tileEntries; p.tilesX = tilesX;
  }

  if (p.pathTime != nullptr) HIP_TRY(hipMemsetAsync(p.pathTime, 0, sizeof(float) * (size_t) p.numPaths, dev->stream)); // before the fork: every lane's launches are behind it

  if (lanes > 1)
  {
    if (!dev->laneFork) HIP_TRY(hipEventCreateWithFlags(&dev->laneFork, hipEventDisableTiming));
    for (int k = 1; k < lanes; ++k)
    {
      if (!dev->laneStream[k]) HIP_TRY(hipStreamCreateWithFlags(&dev->laneStream[k], hipStreamNonBlocking));
      if (!dev->laneDone[k]) HIP_TRY(hipEventCreateWithFlags(&dev->laneDone[k], hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(dev->laneFork, dev->stream)); // everything the handle's stream holds so far (uploads, the previous pass's accumulate)
  }
  // The chains are enqueued side by side, step by step (lane 0's kernel k, lane 1's kernel k, ...): enqueued one whole chain
  // after the other, the second lane would start a chain's worth of host launch time behind the first.
  LaunchParams laneP[TWK_MAX_LANES];
  hipStream_t laneS[TWK_MAX_LANES];
  int shadeGrid[TWK_MAX_LANES];
  int active = 0;
  for (int lane = 0; lane < lanes; ++lane)
  {
    const LaunchParams q = laneParams(dev, p, lane, lanes, traceGrid);
    if (q.numPaths <= 0) continue;
    hipStream_t stream = (lane == 0) ? dev->stream : dev->laneStream[lane];
    if (lane > 0) HIP_TRY(hipStreamWaitEvent(stream, dev->laneFork, 0));
    HIP_TRY(hipMemsetAsync(q.counters, 0, sizeof(unsigned int) * TWK_COUNTER_WORDS, stream));
    // Many more blocks than are resident at once (4 per CU): a block that has finished its windows makes room for the next,
    // which evens out what the blocks' windows cost. Measured on C2 (shade ms/step): 4 / 8 / 16 / 64 / 256 / 2048 blocks per
    // CU = 0.310 / 0.309 / 0.305 / 0.294 / 0.290 / 0.298.
    int grid = (q.numPaths + TWK_SHADE_BLOCK - 1) / TWK_SHADE_BLOCK;
    if (grid > dev->numCUs * TWK_SHADE_BLOCKS_PER_CU) grid = dev->numCUs * TWK_SHADE_BLOCKS_PER_CU;
    laneP[active] = q; laneS[active] = stream; shadeGrid[active] = grid; ++active;
  }
  for (int k = 0; k < active; ++k)
  {
    if (fusedPrimary) { const unsigned int numPaths = (unsigned int) laneP[k].numPaths; HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) laneP[k].counters, (int) numPaths, 1, laneS[k])); continue; } // length of queue 0
    timedLaunchBegin(dev, TWK_KERNEL_GENERATE, laneS[k]); launchGenerate(laneP[k], laneS[k]); timedLaunchEnd(dev, laneS[k]);
  }
  for (int depth = 0; depth < wavefrontDepth; ++depth)
  {
    const bool primary = fusedPrimary && depth == 0;
    // (the PRIMARY build of the traversal kernel needs more registers than seven blocks per CU leave: six at most)
    const int primaryWaves = p.hasCutout ? TWK_TRACE_WAVES_CUTOUT_OTHER : (p.twoLevel ? TWK_TRACE_WAVES_PRIMARY_TWO_LEVEL : TWK_TRACE_WAVES_PRIMARY);
    const int grid = (primary && TWK_PRIMARY_SIX) ? std::min(traceGrid, dev->numCUs * std::max(1, primaryWaves / lanes)) : traceGrid;
    for (int k = 0; k < active; ++k) { timedLaunchBegin(dev, TWK_KERNEL_TRACE, laneS[k]); launchTrace(laneP[k], depth, dev->statsEnabled || dev->timeView, primary, grid, laneS[k]); timedLaunchEnd(dev, laneS[k]); }
    for (int k = 0; k < active; ++k) { timedLaunchBegin(dev, TWK_KERNEL_SHADE, laneS[k]); launchShade(laneP[k], depth, primary, shadeGrid[k], laneS[k]); timedLaunchEnd(dev, laneS[k]); }
  }
  if (maxDepth > 0)
  {
    // closest hits of queue `wavefrontDepth` (empty when wavefrontDepth == maxDepth) + the shadow rays of the last shade
    for (int k = 0; k < active; ++k) { timedLaunchBegin(dev, TWK_KERNEL_TRACE, laneS[k]); launchTrace(laneP[k], wavefrontDepth, dev->statsEnabled || dev->timeView, false, traceGrid, laneS[k]); timedLaunchEnd(dev, laneS[k]); }
  }
  for (int k = 0; k < active; ++k)
  {
    if (laneS[k] == dev->stream) continue;
    const int lane = (int) (std::find(dev->laneStream, dev->laneStream + TWK_MAX_LANES, laneS[k]) - dev->laneStream);
    HIP_TRY(hipEventRecord(dev->laneDone[lane], laneS[k]));
    HIP_TRY(hipStreamWaitEvent(dev->stream, dev->laneDone[lane], 0));
  }
  // the running mean folds the samples of the pass in iteration order over ALL lanes' paths: after the join, on the handle's stream
  timedLaunchBegin(dev, TWK_KERNEL_ACCUM, dev->stream); launchAccumulate(p, dev->stream); timedLaunchEnd(dev, dev->stream);
  HIP_TRY(hipGetLastError());
  return TWK_SUCCESS;
}

// =============================================================================================
extern "C" {

const char* twk_last_error(void) { return g_lastError.c_str(); }
int twk_abi_version(void) { return TWK_ABI_VERSION; }

int twk_device_count(int* count)
try
{
  if (!count) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_device_count: NULL argument");
  *count = 0;
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return twkSetError(TWK_ERROR_NO_DEVICE, std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
  *count = n;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_device_count")

int twk_device_create(TwkDevice* out, int ordinal, int index, int count, int miss)
try
{
  if (!out) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_device_create: NULL out pointer");
  *out = nullptr;
  if (count < 1 || index < 0 || index >= count) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_device_create: need 0 <= index < count");
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return twkSetError(TWK_ERROR_NO_DEVICE, std::string("twk_device_create: no HIP device available (") + ((e != hipSuccess) ? hipGetErrorString(e) : "device count 0") + "); this library has no CPU path");
  if (ordinal < 0 || ordinal >= n) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_device_create: device ordinal out of range");

  TwkDevice_t* dev = new TwkDevice_t();
  dev->ordinal = ordinal; dev->index = index; dev->count = count; dev->miss = miss;
  memset(&dev->params, 0, sizeof(dev->params));
  memset(&dev->state, 0, sizeof(dev->state));
  // Device.cpp:282-303 defaults
  dev->state.resolution[0] = dev->state.resolution[1] = 1;
  dev->state.tileSize[0] = dev->state.tileSize[1] = 8;
  dev->state.pathLengths[0] = 2; dev->state.pathLengths[1] = 5;
  dev->state.epsilonFactor = 500.0f;
  dev->state.clockFactor = 1000.0f;

  hipError_t err = hipSetDevice(ordinal);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking);
  hipDeviceProp_t prop;
  if (err == hipSuccess) err = hipGetDeviceProperties(&prop, ordinal);
  if (err != hipSuccess)
  {
    delete dev;
    return twkSetError(TWK_ERROR_NO_DEVICE, std::string("twk_device_create: ") + hipGetErrorString(err));
  }
  dev->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("TWK_PASS_LANES")) dev->lanesForced = std::max(0, std::min(TWK_MAX_LANES, atoi(e)));
  if (const char* e = getenv("TWK_LANE_TRACE_WAVES")) dev->laneTraceWaves = std::max(0, atoi(e));
  if (const char* e = getenv("TWK_TOP_CACHE")) dev->topCache = (atoi(e) != 0);
  if (const char* e = getenv("TWK_DIRECT_SMALL_LEAVES")) dev->directSmallLeaves = (atoi(e) != 0);
  if (const char* e = getenv("TWK_COSTED_CUTS")) dev->costedCuts = (atoi(e) != 0);
  if (const char* e = getenv("TWK_FUSED_PRIMARY")) dev->fusedPrimary = (atoi(e) != 0);
  if (const char* e = getenv("TWK_TILE_ENTRIES")) dev->tileEntries = (atoi(e) != 0);
  if (const char* e = getenv("TWK_WIDE_ROOT")) dev->wideRoot = (atoi(e) != 0);
  if (const char* e = getenv("TWK_PACKED_QUEUE")) dev->packedQueue = (atoi(e) != 0);
  if (const char* e = getenv("TWK_SHADE_SORT")) dev->shadeSort = atoi(e);
  if (const char* e = getenv("TWK_TRACE_WAVES_RUNTIME")) dev->traceWavesForced = atoi(e); // A/B: 6 or 7 blocks per CU of the persistent trace kernel
  if (const char* e = getenv("TWK_BUILD_QUALITY")) dev->builder.setQuality(atoi(e)); // A/B: 0 LBVH, 1 binned SAH (default)
  memset(&dev->buildInfo, 0, sizeof(dev->buildInfo));
  if (const char* e = getenv("TWK_STREAM_BUDGET_MB")) { const long long mb = atoll(e); dev->streamBudgetBytes = (mb > 0) ? (size_t) mb << 20 : 0; }
  if (const char* e = getenv("TWK_BATCH")) { const int b = atoi(e); dev->batchMax = (b < 1) ? 1 : ((b > 64) ? 64 : b); }
  *out = dev;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_device_create")

int twk_device_destroy(TwkDevice dev)
try
{
  if (!dev) return TWK_SUCCESS;
  dev->pendingCount = 0; // recorded but never observed launches are dropped
  (void) hipSetDevice(dev->ordinal);
  if (dev->stream) (void) hipStreamSynchronize(dev->stream);
  for (TimedLaunch& t : dev->timed) { (void) hipEventDestroy(t.start); (void) hipEventDestroy(t.stop); }
  freeDevice(dev->d_camera); freeDevice(dev->d_lights); freeDevice(dev->d_materials);
  freeDevice(dev->d_attributes); freeDevice(dev->d_indices);
  freeDevice(dev->d_nodes); freeDevice(dev->d_wideNodes); freeDevice(dev->d_wideQ); freeDevice(dev->d_triangles); freeDevice(dev->d_shadeTriangles); freeDevice(dev->d_instances);
  for (int k = 0; k < 3; ++k) freeDevice(dev->d_texels[k]);
  freeDevice(dev->d_envCDF_U); freeDevice(dev->d_envCDF_V); freeDevice(dev->d_topNodes); freeDevice(dev->d_topNodes7); freeDevice(dev->d_tileEntries);
  freeDevice(dev->d_streamBlock); freeDevice(dev->d_outputInternal);
  freeDevice(dev->d_counters); freeDevice(dev->d_stats); freeDevice(dev->d_spill); freeDevice(dev->d_pathTime);
  if (dev->h_dropped) { (void) hipHostFree(dev->h_dropped); dev->h_dropped = nullptr; dev->d_dropped = nullptr; }
  freeDevice(dev->d_firstHit); freeDevice(dev->d_firstHitInstance);
  freeDevice(dev->d_pathAlbedo); freeDevice(dev->d_pathNormal); freeDevice(dev->d_aovAlbedo); freeDevice(dev->d_aovNormal);
  dev->builder.release();
  for (int k = 1; k < TWK_MAX_LANES; ++k)
  {
    if (dev->laneStream[k]) (void) hipStreamDestroy(dev->laneStream[k]);
    if (dev->laneDone[k]) (void) hipEventDestroy(dev->laneDone[k]);
  }
  if (dev->laneFork) (void) hipEventDestroy(dev->laneFork);
  if (dev->stream) (void) hipStreamDestroy(dev->stream);
  delete dev;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_device_destroy")

int twk_set_state(TwkDevice dev, const TwkDeviceState* s)
try
{
  int rc = activate(dev, "twk_set_state"); if (rc) return rc;
  if (!s) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_state: NULL state");
  if (s->resolution[0] < 1 || s->resolution[1] < 1) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_state: resolution must be >= 1");
  if (s->tileSize[0] < 1 || s->tileSize[1] < 1 || (s->tileSize[0] & (s->tileSize[0] - 1)) || (s->tileSize[1] & (s->tileSize[1] - 1)))
    return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_state: tileSize must be a power of two");
  if (s->pathLengths[1] < 0 || s->pathLengths[1] > TWK_MAX_DEPTH) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_state: pathLengths.y must be in [0, 64]");
  if (s->lensShader < 0 || s->lensShader > 2) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_state: lensShader must be 0..2");
  HIP_TRY(hipStreamSynchronize(dev->stream)); // Device.cpp:1194-1195
  dev->state = *s;
  dev->stateSet = true;
  if (s->distribution && 1 < dev->count)
  {
    // DeviceMultiGPULocalCopy.cpp:84-97
    const int width = (s->resolution[0] + dev->count - 1) / dev->count;
    const int mask  = s->tileSize[0] - 1;
    dev->launchWidth = (width + mask) & ~mask;
  }
  else dev->launchWidth = s->resolution[0];
  const size_t needBytes = (size_t) (dev->outputFrame ? s->resolution[0] : dev->launchWidth) * s->resolution[1] * sizeof(float4);
  if (dev->d_outputExternal && dev->outputExternalBytes < needBytes)
  {
    dev->d_outputExternal = nullptr; dev->outputExternalBytes = 0; dev->outputFrame = false; // too small for the new state: fall back to the internal buffer
  }
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_state")

int twk_init_cameras(TwkDevice dev, const TwkCameraDefinition* c, int count)
try
{
  int rc = activate(dev, "twk_init_cameras"); if (rc) return rc;
  if (!c || count < 1) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_init_cameras: at least one camera is required");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->cameras.assign(c, c + count);
  if (!dev->d_camera) HIP_TRY(hipMalloc(&dev->d_camera, sizeof(TwkCameraDefinition)));
  HIP_TRY(hipMemcpyAsync(dev->d_camera, dev->cameras.data(), sizeof(TwkCameraDefinition), hipMemcpyHostToDevice, dev->stream)); // the lens shaders read cameraDefinitions[0]
  return TWK_SUCCESS;
}
TWK_CATCH("twk_init_cameras")

int twk_update_camera(TwkDevice dev, int idCamera, const TwkCameraDefinition* c)
try
{
  int rc = activate(dev, "twk_update_camera"); if (rc) return rc;
  if (!c || idCamera < 0 || idCamera >= (int) dev->cameras.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_update_camera: bad camera id");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->cameras[idCamera] = *c;
  if (idCamera == 0) HIP_TRY(hipMemcpyAsync(dev->d_camera, dev->cameras.data(), sizeof(TwkCameraDefinition), hipMemcpyHostToDevice, dev->stream));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_update_camera")

int twk_init_lights(TwkDevice dev, const TwkLightDefinition* l, int count)
try
{
  int rc = activate(dev, "twk_init_lights"); if (rc) return rc;
  if (count < 0 || (count > 0 && !l)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_init_lights: bad arguments");
  if (dev->built && count <= dev->maxInstanceLight)
    return twkSetError(TWK_ERROR_INVALID_STATE, "twk_init_lights: the built scene has an instance with light index " + std::to_string(dev->maxInstanceLight) + "; " + std::to_string(count) + " lights would leave it dangling (twk_clear_scene first)");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  freeDevice(dev->d_lights);
  dev->lights.assign(l, l + count);
  if (count > 0)
  {
    HIP_TRY(hipMalloc(&dev->d_lights, sizeof(DevLight) * count));
    HIP_TRY(hipMemcpyAsync(dev->d_lights, dev->lights.data(), sizeof(DevLight) * count, hipMemcpyHostToDevice, dev->stream));
  }
  return TWK_SUCCESS;
}
TWK_CATCH("twk_init_lights")

int twk_update_light(TwkDevice dev, int idLight, const TwkLightDefinition* l)
try
{
  int rc = activate(dev, "twk_update_light"); if (rc) return rc;
  if (!l || idLight < 0 || idLight >= (int) dev->lights.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_update_light: bad light id");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->lights[idLight] = *l;
  HIP_TRY(hipMemcpyAsync(dev->d_lights + idLight, &dev->lights[idLight], sizeof(DevLight), hipMemcpyHostToDevice, dev->stream));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_update_light")

int twk_init_materials(TwkDevice dev, const TwkMaterialGUI* m, int count)
try
{
  int rc = activate(dev, "twk_init_materials"); if (rc) return rc;
  if (!m || count < 1) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_init_materials: at least one material is required");
  if (dev->built && count <= dev->maxInstanceMaterial)
    return twkSetError(TWK_ERROR_INVALID_STATE, "twk_init_materials: the built scene has an instance with material index " + std::to_string(dev->maxInstanceMaterial) + "; " + std::to_string(count) + " materials would leave it dangling (twk_clear_scene first)");
  for (int i = 0; i < count; ++i)
  {
    if (m[i].indexBSDF < 0 || m[i].indexBSDF > 4) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_init_materials: indexBSDF out of range");
  }
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->materials.resize(count);
  for (int i = 0; i < count; ++i) dev->materials[i] = convertMaterial(m[i]);
  if (count > dev->materialCapacity)
  {
    freeDevice(dev->d_materials);
    HIP_TRY(hipMalloc(&dev->d_materials, sizeof(DevMaterial) * count));
    dev->materialCapacity = count;
  }
  HIP_TRY(hipMemcpyAsync(dev->d_materials, dev->materials.data(), sizeof(DevMaterial) * count, hipMemcpyHostToDevice, dev->stream));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_init_materials")

int twk_update_material(TwkDevice dev, int idMaterial, const TwkMaterialGUI* m)
try
{
  int rc = activate(dev, "twk_update_material"); if (rc) return rc;
  if (!m || idMaterial < 0 || idMaterial >= (int) dev->materials.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_update_material: bad material id");
  if (m->indexBSDF < 0 || m->indexBSDF > 4) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_update_material: indexBSDF out of range");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->materials[idMaterial] = convertMaterial(*m);
  HIP_TRY(hipMemcpyAsync(dev->d_materials + idMaterial, &dev->materials[idMaterial], sizeof(DevMaterial), hipMemcpyHostToDevice, dev->stream));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_update_material")

int twk_init_texture(TwkDevice dev, int slot, const float* rgba, int width, int height)
try
{
  int rc = activate(dev, "twk_init_texture"); if (rc) return rc;
  if (slot < 0 || slot > 2 || !rgba || width < 1 || height < 1) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_init_texture: bad arguments");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  freeDevice(dev->d_texels[slot]);
  const size_t bytes = sizeof(float4) * (size_t) width * height;
  HIP_TRY(hipMalloc(&dev->d_texels[slot], bytes));
  HIP_TRY(hipMemcpy(dev->d_texels[slot], rgba, bytes, hipMemcpyHostToDevice));
  DevTexture& t = dev->params.textures[slot];
  t.texels = dev->d_texels[slot]; t.width = width; t.height = height; t.clampV = (slot == TWK_TEXTURE_ENVIRONMENT) ? 1 : 0; t.pad = 0;
  if (slot == TWK_TEXTURE_ENVIRONMENT)
  {
    std::vector<float> cdfU, cdfV; float integral = 1.0f;
    calculateSphericalCDF(rgba, (unsigned int) width, (unsigned int) height, cdfU, cdfV, integral);
    freeDevice(dev->d_envCDF_U); freeDevice(dev->d_envCDF_V);
    HIP_TRY(hipMalloc(&dev->d_envCDF_U, sizeof(float) * cdfU.size()));
    HIP_TRY(hipMalloc(&dev->d_envCDF_V, sizeof(float) * cdfV.size()));
    HIP_TRY(hipMemcpy(dev->d_envCDF_U, cdfU.data(), sizeof(float) * cdfU.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dev->d_envCDF_V, cdfV.data(), sizeof(float) * cdfV.size(), hipMemcpyHostToDevice));
    dev->params.envWidth = (unsigned int) width; dev->params.envHeight = (unsigned int) height; dev->params.envIntegral = integral;
  }
  return TWK_SUCCESS;
}
TWK_CATCH("twk_init_texture")

int twk_clear_scene(TwkDevice dev)
try
{
  int rc = activate(dev, "twk_clear_scene"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->geometries.clear(); dev->instances.clear(); dev->built = false;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_clear_scene")

int twk_add_geometry(TwkDevice dev, const TwkTriangleAttributes* attributes, size_t numAttributes,
                     const unsigned int* indices, size_t numIndices, int* idGeometry)
try
{
  int rc = activate(dev, "twk_add_geometry"); if (rc) return rc;
  if (!attributes || !indices || numAttributes == 0 || numIndices == 0 || (numIndices % 3) != 0)
    return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_add_geometry: need attributes and a non-empty multiple of three indices");
  // a leaf reference holds a 28-bit triangle slot (device_types.h BvhNode): refuse here what twk_build could not address
  if (numIndices / 3 >= ((size_t) 1 << 28) || numAttributes >= ((size_t) 1 << 32))
    return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_add_geometry: more than 2^28 - 1 triangles (or 2^32 - 1 vertices) in one geometry");
  for (size_t i = 0; i < numIndices; ++i)
    if (indices[i] >= numAttributes) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_add_geometry: index out of range");
  GeometryHost g;
  g.attributes.assign(attributes, attributes + numAttributes);
  g.indices.assign(indices, indices + numIndices);
  g.numTriangles = (int) (numIndices / 3);
  dev->geometries.push_back(std::move(g));
  dev->built = false;
  if (idGeometry) *idGeometry = (int) dev->geometries.size() - 1;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_add_geometry")

int twk_add_instance(TwkDevice dev, int idGeometry, const float transform[12], int idMaterial, int idLight, int* idInstance)
try
{
  int rc = activate(dev, "twk_add_instance"); if (rc) return rc;
  if (!transform || idGeometry < 0 || idGeometry >= (int) dev->geometries.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_add_instance: bad geometry id");
  if (idMaterial < 0) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_add_instance: an instance needs a material (Device.cpp:1429)");
  InstanceHost inst;
  inst.geometry = idGeometry; inst.material = idMaterial; inst.light = idLight;
  memcpy(inst.transform, transform, sizeof(float) * 12);
  dev->instances.push_back(inst);
  dev->built = false;
  if (idInstance) *idInstance = (int) dev->instances.size() - 1;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_add_instance")

int twk_set_flatten_policy(TwkDevice dev, int maxTriangles, int maxReferences)
try
{
  int rc = activate(dev, "twk_set_flatten_policy"); if (rc) return rc;
  if (maxTriangles < 0 || maxReferences < 0) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_flatten_policy: limits must be >= 0");
  dev->flattenMaxTriangles = maxTriangles; dev->flattenMaxReferences = maxReferences;
  dev->built = false;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_flatten_policy")

int twk_set_build_quality(TwkDevice dev, int quality)
try
{
  int rc = activate(dev, "twk_set_build_quality"); if (rc) return rc;
  if (quality != TWK_BUILD_LBVH && quality != TWK_BUILD_SAH) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_build_quality: unknown quality");
  dev->builder.setQuality(quality);
  dev->built = false;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_build_quality")

int twk_get_build_info(TwkDevice dev, TwkBuildInfo* info)
try
{
  if (!dev || !info) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_get_build_info: NULL argument");
  if (!dev->built) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_get_build_info: twk_build has not been called");
  refreshParams(dev); // the traversal kernel variant depends on the materials as they are now
  dev->buildInfo.traceBlocksPerCU = (uint64_t) dev->params.traceWaves;
  *info = dev->buildInfo;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_get_build_info")

int twk_build(TwkDevice dev)
try
{
  int rc = activate(dev, "twk_build"); if (rc) return rc;
  if (dev->geometries.empty() || dev->instances.empty()) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_build: the scene has no geometry or no instance");
  dev->built = false; // until this build has succeeded: a failure below leaves no half-built scene to launch on
  const auto buildStart = std::chrono::steady_clock::now();
  TwkBuildInfo info;
  memset(&info, 0, sizeof(info));
  info.quality = dev->builder.quality();
  int maxMaterial = -1, maxLight = -1;
  for (const InstanceHost& inst : dev->instances)
  {
    if (inst.material >= (int) dev->materials.size()) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_build: instance material index beyond twk_init_materials");
    if (inst.light >= (int) dev->lights.size()) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_build: instance light index beyond twk_init_lights");
    if (inst.material > maxMaterial) maxMaterial = inst.material;
    if (inst.light > maxLight) maxLight = inst.light;
  }
  HIP_TRY(hipStreamSynchronize(dev->stream));

  // Which instances are flattened (include/tweeker_hip.h twk_set_flatten_policy): those of tiny geometries and those
  // whose geometry is referenced so rarely that instancing saves no memory worth the per-ray instance entry (ray
  // transform, per-instance Woop constants, exit step). A flattened instance gets world-space triangle slots and an
  // LBVH of its own whose root is spliced into the top level as an inner node: traversal walks from the top level
  // straight into it with the untransformed ray.
  const int numInstances = (int) dev->instances.size();
  std::vector<int> references(dev->geometries.size(), 0);
  for (const InstanceHost& inst : dev->instances) references[inst.geometry]++;
  std::vector<char> flattened(numInstances, 0), needsBlas(dev->geometries.size(), 0);
  int numEntered = 0, maxFlatTriangles = 0;
  for (int i = 0; i < numInstances; ++i)
  {
    const int g = dev->instances[i].geometry;
    flattened[i] = (dev->geometries[g].numTriangles <= dev->flattenMaxTriangles) || (references[g] <= dev->flattenMaxReferences);
    if (flattened[i]) maxFlatTriangles = std::max(maxFlatTriangles, dev->geometries[g].numTriangles);
    else { needsBlas[g] = 1; ++numEntered; }
  }

  // shared attribute / index arrays and the node / triangle budgets: one bottom level per geometry that is still
  // entered through an instance, one world-space tree per flattened instance, the top level
  size_t numAttr = 0, numIdx = 0, numTris = 0, numNodes = 0;
  for (size_t k = 0; k < dev->geometries.size(); ++k)
  {
    GeometryHost& g = dev->geometries[k];
    g.attributeBase = (unsigned int) numAttr; g.indexBase = (unsigned int) numIdx;
    g.triangleBase = (int) numTris; g.nodeBase = (int) numNodes;
    numAttr += g.attributes.size(); numIdx += g.indices.size();
    if (needsBlas[k]) { numTris += (size_t) g.numTriangles; numNodes += (size_t) ((g.numTriangles > 1) ? g.numTriangles - 1 : 1); }
  }
  std::vector<int> flatTriangleBase(numInstances, -1), flatNodeBase(numInstances, -1);
  for (int i = 0; i < numInstances; ++i)
  {
    if (!flattened[i]) continue;
    const int n = dev->geometries[dev->instances[i].geometry].numTriangles;
    flatTriangleBase[i] = (int) numTris; flatNodeBase[i] = (int) numNodes;
    numTris += (size_t) n; numNodes += (size_t) ((n > 1) ? n - 1 : 1);
  }
  const int tlasBase = (int) numNodes;
  numNodes += (size_t) ((numInstances > 1) ? numInstances - 1 : 1);
  if (numTris >= ((size_t) 1 << 28) || numAttr >= ((size_t) 1 << 31) || numIdx >= ((size_t) 1 << 31))
    return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_build: " + std::to_string(numTris) + " triangle slots; a leaf reference holds 28 bits of slot index");

  freeDevice(dev->d_attributes); freeDevice(dev->d_indices); freeDevice(dev->d_nodes); freeDevice(dev->d_wideNodes); freeDevice(dev->d_wideQ); freeDevice(dev->d_triangles); freeDevice(dev->d_shadeTriangles); freeDevice(dev->d_instances);
  HIP_TRY(hipMalloc(&dev->d_attributes, sizeof(TwkTriangleAttributes) * numAttr));
  HIP_TRY(hipMalloc(&dev->d_indices, sizeof(unsigned int) * numIdx));
  HIP_TRY(hipMalloc(&dev->d_nodes, sizeof(BvhNode) * numNodes));
  HIP_TRY(hipMalloc(&dev->d_wideNodes, sizeof(BvhNode) * 2 * (numNodes + 2))); // + the two nodes of an 8-wide root (wideRootKernel)
  HIP_TRY(hipMalloc(&dev->d_wideQ, sizeof(float4) * 4 * (numNodes + 2)));
  HIP_TRY(hipMalloc(&dev->d_triangles, sizeof(float4) * 3 * numTris));
  HIP_TRY(hipMalloc(&dev->d_shadeTriangles, sizeof(float4) * TWK_SHADE_RECORD * numTris));
  HIP_TRY(hipMalloc(&dev->d_instances, sizeof(DevInstance) * numInstances));
  for (const GeometryHost& g : dev->geometries)
  {
    HIP_TRY(hipMemcpyAsync(dev->d_attributes + 12 * (size_t) g.attributeBase, g.attributes.data(), sizeof(TwkTriangleAttributes) * g.attributes.size(), hipMemcpyHostToDevice, dev->stream));
    HIP_TRY(hipMemcpyAsync(dev->d_indices + g.indexBase, g.indices.data(), sizeof(unsigned int) * g.indices.size(), hipMemcpyHostToDevice, dev->stream));
  }

  if (const char* e = getenv("TWK_MAX_LEAF")) dev->builder.setMaxLeaf(atoi(e)); // tuning knob, default 2 triangles per leaf
  ScopedDeviceBuffer<float> nodeCost; // expected wide-node visits below each node: what the wide nodes' cuts are chosen by (bvh_build.hip refitKernel)
  if (dev->costedCuts) HIP_TRY(nodeCost.allocate(numNodes));
  struct NodeCostScope { BvhBuilder& b; ~NodeCostScope() { b.setNodeCost(nullptr); } } nodeCostScope{dev->builder}; // the array does not outlive this call
  dev->builder.setNodeCost(nodeCost.ptr);
  int maxEnteredHeight = 0, maxFlatHeight = 0, topHeight = 0; // binary-tree heights: what a traversal stack may have to hold
  // bottom level: one LBVH per entered geometry, shared by all of its instances (Device.cpp:1339 caches the GAS per Triangles id)
  for (size_t k = 0; k < dev->geometries.size(); ++k)
  {
    GeometryHost& g = dev->geometries[k];
    if (!needsBlas[k]) continue;
    HIP_TRY(dev->builder.buildTriangles(dev->stream, dev->d_attributes + 12 * (size_t) g.attributeBase, dev->d_indices + g.indexBase, g.numTriangles,
                                        dev->d_nodes + g.nodeBase, dev->d_wideNodes + 2 * (size_t) g.nodeBase, g.nodeBase, dev->d_triangles, dev->d_shadeTriangles, g.triangleBase, g.rootBounds));
    info.sahInnerCost += dev->builder.lastSahInner(); info.sahLeafCost += dev->builder.lastSahLeaf(); info.trees += 1;
    maxEnteredHeight = std::max(maxEnteredHeight, dev->builder.lastHeight());
  }

  // instance records (shading reads them for every hit, flattened or not)
  std::vector<DevInstance> records(numInstances);
  for (int i = 0; i < numInstances; ++i)
  {
    const InstanceHost& inst = dev->instances[i];
    const GeometryHost& g = dev->geometries[inst.geometry];
    DevInstance& r = records[i];
    memset(&r, 0, sizeof(r));
    memcpy(r.objectToWorld, inst.transform, sizeof(float) * 12);
    invertAffine(inst.transform, r.worldToObject);
    r.blasRoot = flattened[i] ? flatNodeBase[i] : g.nodeBase; r.material = inst.material; r.light = inst.light;
    r.triangleFirst = flattened[i] ? flatTriangleBase[i] : g.triangleBase; r.triangleCount = g.numTriangles;
    r.attributeBase = g.attributeBase; r.indexBase = g.indexBase; r.geometry = inst.geometry;
  }
  HIP_TRY(hipMemcpyAsync(dev->d_instances, records.data(), sizeof(DevInstance) * numInstances, hipMemcpyHostToDevice, dev->stream));

  // world-space trees of the flattened instances + the world boxes of all instances
  std::vector<float4> boxLo(numInstances), boxHi(numInstances);
  std::vector<int> leafPayload(numInstances);
  ScopedDeviceBuffer<int4> soup;
  if (maxFlatTriangles > 0) HIP_TRY(soup.allocate((size_t) maxFlatTriangles));
  for (int i = 0; i < numInstances; ++i)
  {
    const InstanceHost& inst = dev->instances[i];
    const GeometryHost& g = dev->geometries[inst.geometry];
    if (flattened[i])
    {
      float bounds[6];
      dev->builder.soupDescriptors(dev->stream, soup.ptr, 0, g.numTriangles, i, (int) g.attributeBase, (int) g.indexBase);
      HIP_TRY(hipGetLastError());
      HIP_TRY(dev->builder.buildTriangles(dev->stream, dev->d_attributes, dev->d_indices, g.numTriangles,
                                          dev->d_nodes + flatNodeBase[i], dev->d_wideNodes + 2 * (size_t) flatNodeBase[i], flatNodeBase[i],
                                          dev->d_triangles, dev->d_shadeTriangles, flatTriangleBase[i], bounds, soup.ptr, dev->d_instances));
      info.sahInnerCost += dev->builder.lastSahInner(); info.sahLeafCost += dev->builder.lastSahLeaf(); info.trees += 1;
      maxFlatHeight = std::max(maxFlatHeight, dev->builder.lastHeight());
      boxLo[i] = make_float4(bounds[0], bounds[1], bounds[2], 0.0f);
      boxHi[i] = make_float4(bounds[3], bounds[4], bounds[5], 0.0f);
      leafPayload[i] = ~flatNodeBase[i]; // child reference ~payload = the instance's root node: an inner reference
      // A flattened instance of no more triangles than a leaf holds (a wall, the area light: two triangles) IS a leaf of the
      // top level: its slots are referenced directly instead of through a one-node tree of two single-triangle leaves —
      // one node visit and one leaf step less for every ray that crosses its box (C2: six of the eight instances).
      if (dev->directSmallLeaves && g.numTriangles <= dev->builder.maxLeaf() && g.numTriangles <= 4)
      {
        leafPayload[i] = flatTriangleBase[i] | ((g.numTriangles - 1) << 28) | TWK_LEAF_WORLD;
        info.directLeafInstances += 1;
      }
      continue;
    }
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int corner = 0; corner < 8; ++corner)
    {
      const float x = g.rootBounds[(corner & 1) ? 3 : 0], y = g.rootBounds[(corner & 2) ? 4 : 1], z = g.rootBounds[(corner & 4) ? 5 : 2];
      const float* m = inst.transform;
      const float w[3] = { m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11] };
      for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], w[k]); hi[k] = fmaxf(hi[k], w[k]); }
    }
    for (int k = 0; k < 3; ++k)
    {
      // world box of an object-space box: pad for the rounding of the transform in both directions
      const float e = 1.0e-5f * fmaxf(1.0f, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
      lo[k] -= e; hi[k] += e;
    }
    boxLo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    boxHi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    leafPayload[i] = i;
  }
  if (numInstances == 1 && flattened[0]) dev->tlasRoot = flatNodeBase[0]; // the one world-space tree IS the scene
  else
  {
    HIP_TRY(dev->builder.buildInstances(dev->stream, boxLo.data(), boxHi.data(), leafPayload.data(), numInstances, dev->d_nodes + tlasBase, dev->d_wideNodes + 2 * (size_t) tlasBase, tlasBase));
    dev->tlasRoot = tlasBase;
    topHeight = dev->builder.lastHeight();
  }
  // Deepest stack a single-ray traversal can need (trace_device.h traverse(): at most one push per inner node on the path,
  // plus the sentinel of an instance entry): the top level, then either a spliced world-space tree or an entered
  // geometry's tree. The persistent kernel hands rays that outgrow its LDS stack to that traversal, whose stack holds
  // TWK_TRACE_STACK_LDS + TWK_TRACE_STACK_SPILL entries; a scene beyond that would lose subtrees silently, so it is refused.
  const int traversalDepth = topHeight + std::max(maxFlatHeight, (numEntered > 0) ? 1 + maxEnteredHeight : 0);
  int depthLimit = TWK_TRACE_STACK_LDS + TWK_TRACE_STACK_SPILL - 2;
  if (const char* e = getenv("TWK_MAX_TRAVERSAL_DEPTH")) depthLimit = std::min(depthLimit, atoi(e)); // test hook: a lower limit only
  if (traversalDepth > depthLimit)
  {
    dev->built = false;
    return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_build: the acceleration structure is " + std::to_string(traversalDepth) + " levels deep (top " + std::to_string(topHeight) +
                       ", flattened trees " + std::to_string(maxFlatHeight) + ", entered geometries " + std::to_string(maxEnteredHeight) + "); the traversal stacks hold " +
                       std::to_string(TWK_TRACE_STACK_LDS + TWK_TRACE_STACK_SPILL) + " entries" + (dev->builder.quality() == TWK_BUILD_SAH ? " (try twk_set_build_quality(TWK_BUILD_LBVH))" : ""));
  }
  // the persistent trace kernel reads the quantised copy of the wide nodes; the full-precision ones were scratch
  // the root as two wide nodes where that pays (bvh_build.hip wideRootKernel)
  dev->wideRoot1 = dev->tlasRoot; dev->wideRoot2 = TWK_BVH_SENTINEL; dev->wideNodesTotal = numNodes;
  if (dev->wideRoot)
  {
    ScopedDeviceBuffer<int> result;
    HIP_TRY(result.allocate(1));
    launchWideRoot(dev->d_wideNodes, dev->tlasRoot, (int) numNodes, result.ptr, dev->stream);
    int has = 0;
    HIP_TRY(hipMemcpyAsync(&has, result.ptr, sizeof(int), hipMemcpyDeviceToHost, dev->stream));
    HIP_TRY(hipStreamSynchronize(dev->stream));
    if (has) { dev->wideRoot1 = (int) numNodes; dev->wideRoot2 = (int) numNodes + 1; dev->wideNodesTotal = numNodes + 2; }
  }
  launchQuantizeWide(dev->d_wideNodes, dev->d_wideQ, (int) dev->wideNodesTotal, dev->stream);
  if (!dev->d_topNodes) HIP_TRY(hipMalloc(&dev->d_topNodes, sizeof(float4) * 4 * TWK_TOP_NODES));
  if (!dev->d_topNodes7) HIP_TRY(hipMalloc(&dev->d_topNodes7, sizeof(float4) * 4 * TWK_TOP_NODES7));
  launchTopCache(dev->d_wideQ, dev->wideRoot1, dev->wideRoot2, dev->d_topNodes, TWK_TOP_NODES, dev->stream);
  launchTopCache(dev->d_wideQ, dev->wideRoot1, dev->wideRoot2, dev->d_topNodes7, TWK_TOP_NODES7, dev->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(dev->stream));
  freeDevice(dev->d_wideNodes);

  info.maxTraversalDepth = (uint64_t) traversalDepth;
  info.triangleSlots = numTris; info.nodes = numNodes; info.instances = (uint64_t) numInstances; info.flattenedInstances = (uint64_t) (numInstances - numEntered);
  info.buildMilliseconds = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - buildStart).count();
  dev->buildInfo = info;
  dev->twoLevel = (numEntered > 0);
  dev->maxInstanceMaterial = maxMaterial; dev->maxInstanceLight = maxLight;
  dev->totalNodes = numNodes; dev->totalTriangles = numTris;
  dev->built = true; ++dev->buildSerial;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_build")

int twk_launch(TwkDevice dev, unsigned int iterationIndex)
try
{
  int rc = activate(dev, "twk_launch", false); if (rc) return rc;
  if (!dev->stateSet) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_launch: twk_set_state has not been called (samplesSqrt 0, Device.cpp:293)");
  if (!dev->built) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_launch: twk_build has not been called");
  if (dev->cameras.empty() || dev->materials.empty()) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_launch: cameras and materials are required");
  if (dev->miss == 2 && dev->d_texels[TWK_TEXTURE_ENVIRONMENT] == nullptr) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_launch: miss 2 needs an environment texture");
  for (const DevMaterial& m : dev->materials)
  {
    if (m.textureAlbedo && dev->d_texels[TWK_TEXTURE_ALBEDO] == nullptr) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_launch: a material uses the albedo texture but none was uploaded");
    if (m.textureCutout && dev->d_texels[TWK_TEXTURE_CUTOUT] == nullptr) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_launch: a material uses the cutout texture but none was uploaded");
  }

  // Asynchronous like optixLaunch: the iteration is recorded; consecutive iterations are rendered together (up to
  // batchMax samples per pixel per wavefront pass). Results are identical to one pass per iteration.
  const int limit = dev->captureFirstHits ? 1 : (dev->batchMax > 1 ? dev->batchMax : 1);
  if (dev->pendingCount > 0 && (iterationIndex != dev->pendingFirst + (unsigned int) dev->pendingCount || dev->pendingCount >= limit))
  {
    if ((rc = flushPending(dev))) return rc;
  }
  if (dev->pendingCount == 0) dev->pendingFirst = iterationIndex;
  dev->pendingCount++;
  if (dev->pendingCount >= limit) return flushPending(dev);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_launch")

int twk_set_launch_batch(TwkDevice dev, int iterations)
try
{
  int rc = activate(dev, "twk_set_launch_batch"); if (rc) return rc;
  if (iterations < 1 || iterations > 64) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_launch_batch: 1..64 iterations per pass");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->batchMax = iterations;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_launch_batch")

int twk_reserve_launch_batch(TwkDevice dev, int iterations)
try
{
  int rc = activate(dev, "twk_reserve_launch_batch"); if (rc) return rc;
  if (iterations < 1 || iterations > 64) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_reserve_launch_batch: 1..64 iterations per pass");
  if (!dev->stateSet) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_reserve_launch_batch: twk_set_state first");
  return ensureStreams(dev, iterations);
}
TWK_CATCH("twk_reserve_launch_batch")

int twk_sync(TwkDevice dev)
try
{
  int rc = activate(dev, "twk_sync"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  return checkDroppedPushes(dev, "twk_sync");
}
TWK_CATCH("twk_sync")

int twk_get_launch_width(TwkDevice dev, int* launchWidth)
try
{
  if (!dev || !launchWidth) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_get_launch_width: NULL argument");
  *launchWidth = dev->launchWidth;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_get_launch_width")

int twk_read_output(TwkDevice dev, float* rgbaHost, size_t numFloats)
try
{
  int rc = activate(dev, "twk_read_output"); if (rc) return rc;
  if (!rgbaHost) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_read_output: NULL buffer");
  const size_t n = (size_t) ((dev->d_outputExternal && dev->outputFrame) ? dev->state.resolution[0] : dev->launchWidth) * dev->state.resolution[1];
  if (numFloats != n * 4) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_read_output: buffer must hold launchWidth*height*4 floats (width*height*4 with a shared frame)");
  const float4* src = dev->d_outputExternal ? dev->d_outputExternal : dev->d_outputInternal;
  if (!src) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_read_output: nothing has been rendered");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  HIP_TRY(hipMemcpy(rgbaHost, src, n * sizeof(float4), hipMemcpyDeviceToHost));
  return checkDroppedPushes(dev, "twk_read_output");
}
TWK_CATCH("twk_read_output")

int twk_set_shader_variant(TwkDevice dev, int variant)
try
{
  int rc = activate(dev, "twk_set_shader_variant"); if (rc) return rc;
  if (variant != TWK_SHADERS_RTIGO3 && variant != TWK_SHADERS_OPTIX7GUI) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_shader_variant: unknown variant");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->shaderVariant = variant;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_shader_variant")

int twk_enable_aov(TwkDevice dev, int enable)
try
{
  int rc = activate(dev, "twk_enable_aov"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->aovEnabled = (enable != 0);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_enable_aov")

int twk_set_time_view(TwkDevice dev, int enable)
try
{
  int rc = activate(dev, "twk_set_time_view"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->timeView = (enable != 0);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_time_view")

int twk_set_next_event_estimation(TwkDevice dev, int enable)
try
{
  int rc = activate(dev, "twk_set_next_event_estimation"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->nextEventEstimation = (enable != 0);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_next_event_estimation")

int twk_set_debug_exceptions(TwkDevice dev, int enable)
try
{
  int rc = activate(dev, "twk_set_debug_exceptions"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->debugExceptions = (enable != 0);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_debug_exceptions")

int twk_read_aov(TwkDevice dev, int which, float* rgbaHost, size_t numFloats)
try
{
  int rc = activate(dev, "twk_read_aov"); if (rc) return rc;
  if (!rgbaHost || (which != TWK_AOV_ALBEDO && which != TWK_AOV_NORMAL)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_read_aov: bad arguments");
  const size_t n = (size_t) dev->launchWidth * dev->state.resolution[1];
  if (numFloats != n * 4) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_read_aov: buffer must hold launchWidth*height*4 floats");
  const float4* src = (which == TWK_AOV_ALBEDO) ? dev->d_aovAlbedo : dev->d_aovNormal;
  if (!dev->aovEnabled || !src || (size_t) dev->aovPixels < n) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_read_aov: nothing has been rendered with twk_enable_aov(1)");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  HIP_TRY(hipMemcpy(rgbaHost, src, n * sizeof(float4), hipMemcpyDeviceToHost));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_read_aov")

int twk_get_output_device_pointer(TwkDevice dev, void** dptr, size_t* bytes)
try
{
  int rc = activate(dev, "twk_get_output_device_pointer"); if (rc) return rc;
  if (!dptr) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_get_output_device_pointer: NULL argument");
  if (!dev->stateSet) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_get_output_device_pointer: twk_set_state first");
  if ((rc = ensureStreams(dev))) return rc;
  *dptr = dev->d_outputExternal ? dev->d_outputExternal : dev->d_outputInternal;
  if (bytes) *bytes = (size_t) dev->launchWidth * dev->state.resolution[1] * sizeof(float4);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_get_output_device_pointer")

int twk_set_output_device_pointer(TwkDevice dev, void* dptr, size_t bytes)
try
{
  int rc = activate(dev, "twk_set_output_device_pointer"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  if (dptr == nullptr) { dev->d_outputExternal = nullptr; dev->outputExternalBytes = 0; dev->outputFrame = false; return TWK_SUCCESS; }
  if (!dev->stateSet) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_set_output_device_pointer: twk_set_state first");
  if (bytes < (size_t) dev->launchWidth * dev->state.resolution[1] * sizeof(float4)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_output_device_pointer: buffer smaller than launchWidth*height*16 bytes");
  dev->d_outputExternal = static_cast<float4*>(dptr); dev->outputExternalBytes = bytes; dev->outputFrame = false;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_output_device_pointer")

int twk_set_shared_frame(TwkDevice dev, void* frame, size_t bytes)
try
{
  int rc = activate(dev, "twk_set_shared_frame"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  if (frame == nullptr) { dev->d_outputExternal = nullptr; dev->outputExternalBytes = 0; dev->outputFrame = false; return TWK_SUCCESS; }
  if (!dev->stateSet) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_set_shared_frame: twk_set_state first");
  if (bytes < (size_t) dev->state.resolution[0] * dev->state.resolution[1] * sizeof(float4)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_set_shared_frame: buffer smaller than width*height*16 bytes");
  dev->d_outputExternal = static_cast<float4*>(frame); dev->outputExternalBytes = bytes; dev->outputFrame = true;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_set_shared_frame")

int twk_compositor(TwkDevice dev, const void* tiles, void* output)
try
{
  int rc = activate(dev, "twk_compositor"); if (rc) return rc;
  if (!tiles || !output) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_compositor: NULL buffer");
  if (!dev->stateSet) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_compositor: twk_set_state first");
  launchCompositor(static_cast<const float4*>(tiles), static_cast<float4*>(output), dev->state.resolution[0], dev->state.resolution[1],
                   dev->launchWidth, dev->count, dev->state.tileSize[0], calculateShift(dev->state.tileSize[0]), calculateShift(dev->state.tileSize[1]), dev->stream);
  HIP_TRY(hipGetLastError());
  return TWK_SUCCESS;
}
TWK_CATCH("twk_compositor")

int twk_tonemap(TwkDevice dev, const TwkTonemapper* tm, const void* rgbaDevice, size_t numPixels, unsigned char* rgb8Host)
try
{
  int rc = activate(dev, "twk_tonemap"); if (rc) return rc;
  if (!tm || !rgb8Host) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_tonemap: NULL argument");
  if (!(tm->gamma > 0.0f) || !(tm->whitePoint > 0.0f)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_tonemap: gamma and whitePoint must be positive");
  const float4* src = static_cast<const float4*>(rgbaDevice);
  if (!src)
  {
    src = dev->d_outputExternal ? dev->d_outputExternal : dev->d_outputInternal;
    if (!src || !dev->stateSet) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_tonemap: nothing has been rendered");
    if (numPixels != (size_t) dev->launchWidth * dev->state.resolution[1]) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_tonemap: numPixels must be launchWidth*height for the handle's own buffer");
  }
  if (numPixels == 0) return TWK_SUCCESS;
  ScopedDeviceBuffer<unsigned char> ldr;
  HIP_TRY(ldr.allocate(numPixels * 3));
  launchTonemap(src, ldr.ptr, numPixels, *tm, dev->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(rgb8Host, ldr.ptr, numPixels * 3, hipMemcpyDeviceToHost, dev->stream));
  HIP_TRY(hipStreamSynchronize(dev->stream));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_tonemap")

// ---- measurement ------------------------------------------------------------------------------
int twk_profile_enable(TwkDevice dev, int enable)
try
{
  int rc = activate(dev, "twk_profile_enable"); if (rc) return rc;
  if ((rc = collectTimed(dev))) return rc;
  dev->profileEnabled = (enable != 0);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_profile_enable")

int twk_profile_reset(TwkDevice dev)
try
{
  int rc = activate(dev, "twk_profile_reset"); if (rc) return rc;
  if ((rc = collectTimed(dev))) return rc;
  for (int k = 0; k < TWK_KERNEL_COUNT; ++k) { dev->profileMs[k] = 0.0f; dev->profileLaunches[k] = 0; }
  return TWK_SUCCESS;
}
TWK_CATCH("twk_profile_reset")

int twk_profile_get(TwkDevice dev, float ms[TWK_KERNEL_COUNT], int launches[TWK_KERNEL_COUNT])
try
{
  int rc = activate(dev, "twk_profile_get"); if (rc) return rc;
  if ((rc = collectTimed(dev))) return rc;
  for (int k = 0; k < TWK_KERNEL_COUNT; ++k) { if (ms) ms[k] = dev->profileMs[k]; if (launches) launches[k] = dev->profileLaunches[k]; }
  return TWK_SUCCESS;
}
TWK_CATCH("twk_profile_get")

int twk_stats_enable(TwkDevice dev, int enable)
try
{
  int rc = activate(dev, "twk_stats_enable"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->statsEnabled = (enable != 0);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_stats_enable")

int twk_stats_get(TwkDevice dev, TwkLaunchStats* stats, int reset)
try
{
  int rc = activate(dev, "twk_stats_get"); if (rc) return rc;
  if (!stats) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_stats_get: NULL argument");
  memset(stats, 0, sizeof(*stats));
  if (!dev->d_stats) return TWK_SUCCESS;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  unsigned long long h[TWK_STATS_WORDS / 2];
  HIP_TRY(hipMemcpy(h, dev->d_stats, sizeof(h), hipMemcpyDeviceToHost));
  stats->radianceRays = h[0]; stats->shadowRays = h[1]; stats->nodesVisited = h[2]; stats->trianglesTested = h[3];
  stats->instancesEntered = h[4]; stats->shadedHits = h[5]; stats->missed = h[6]; stats->maxNodesPerRay = h[7];
  stats->overflowRays = h[12]; // tailRays .. tailInstancesEntered (words 8-11): the tail kernel is not part of this build, they stay 0
  stats->nodeWaveSteps = h[13]; stats->triangleWaveSteps = h[14]; stats->leafWaveSteps = h[15];
  stats->cachedNodesVisited = h[16]; stats->droppedStackPushes = dev->h_dropped ? *dev->h_dropped : 0u;
  for (int i = 0; i < 6; ++i) stats->waveCycles[i] = h[18 + i];
  const int TWK_SHADE_PHASES = TWK_SHADE_PHASE_COUNT; static_assert(24 + 3 * TWK_SHADE_PHASE_COUNT <= TWK_STATS_WORDS / 2, "shade phase words"); // shade_device.h asserts TWK_SHADE_PHASES == TWK_SHADE_PHASE_COUNT
  for (int i = 0; i < TWK_SHADE_PHASES; ++i) { stats->shadePhaseWaveSteps[i] = h[24 + i]; stats->shadePhaseLanes[i] = h[24 + TWK_SHADE_PHASES + i]; stats->shadePhaseCycles[i] = h[24 + 2 * TWK_SHADE_PHASES + i]; }
  if (reset) HIP_TRY(hipMemset(dev->d_stats, 0, sizeof(h)));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_stats_get")

int twk_stream_peak_gbps(TwkDevice dev, size_t bytes, int repeats, float* gbps)
try
{
  int rc = activate(dev, "twk_stream_peak_gbps"); if (rc) return rc;
  if (!gbps || bytes < 4096 || repeats < 1) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_stream_peak_gbps: bad arguments");
  const size_t n = (bytes / sizeof(float4)) & ~(size_t) 1023; // whole 16 KiB pieces of the copy kernel
  float4 *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc(&a, n * sizeof(float4)));
  if (hipMalloc(&b, n * sizeof(float4)) != hipSuccess) { (void) hipFree(a); return twkSetError(TWK_ERROR_OUT_OF_MEMORY, "twk_stream_peak_gbps: out of memory"); }
  (void) hipMemsetAsync(a, 0, n * sizeof(float4), dev->stream);
  hipEvent_t e0, e1;
  (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  launchStreamCopy(a, b, n, dev->stream); // warm-up
  (void) hipEventRecord(e0, dev->stream);
  for (int i = 0; i < repeats; ++i) launchStreamCopy(a, b, n, dev->stream);
  (void) hipEventRecord(e1, dev->stream);
  hipError_t e = hipStreamSynchronize(dev->stream);
  float ms = 0.0f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
  (void) hipFree(a); (void) hipFree(b);
  if (e != hipSuccess) return twkSetError(TWK_ERROR_HIP, std::string("twk_stream_peak_gbps: ") + hipGetErrorString(e));
  *gbps = (float) (2.0 * (double) (n * sizeof(float4)) * repeats / ((double) ms * 1.0e-3) / 1.0e9);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_stream_peak_gbps")

int twk_gather_peak(TwkDevice dev, size_t tableBytes, float* gigaLaneLoadsPerSecond)
try
{
  int rc = activate(dev, "twk_gather_peak"); if (rc) return rc;
  if (!gigaLaneLoadsPerSecond || tableBytes < 128 * 1024 || tableBytes > ((size_t) 1 << 36)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_gather_peak: bad arguments");
  const unsigned int lines = (unsigned int) (tableBytes / 128);
  ScopedDeviceBuffer<float4> table; ScopedDeviceBuffer<float> out;
  HIP_TRY(table.allocate((size_t) lines * 8));
  HIP_TRY(out.allocate(1));
  launchGatherProbeFill(table.ptr, (size_t) lines * 8, lines, dev->stream);
  const int blocks = dev->numCUs * 6, steps = 1000; // 6 waves per SIMD, as the traversal kernel runs
  launchGatherProbe(table.ptr, lines, 50, out.ptr, blocks, dev->stream); // warm-up: table into the caches
  hipEvent_t e0, e1;
  (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  (void) hipEventRecord(e0, dev->stream);
  launchGatherProbe(table.ptr, lines, steps, out.ptr, blocks, dev->stream);
  (void) hipEventRecord(e1, dev->stream);
  hipError_t e = hipStreamSynchronize(dev->stream);
  float ms = 0.0f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
  if (e != hipSuccess) return twkSetError(TWK_ERROR_HIP, std::string("twk_gather_peak: ") + hipGetErrorString(e));
  *gigaLaneLoadsPerSecond = (float) ((double) blocks * 256.0 * steps * 8.0 / ((double) ms * 1.0e-3) / 1.0e9);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_gather_peak")

// ---- parity taps ------------------------------------------------------------------------------
int twk_debug_capture(TwkDevice dev, int enable)
try
{
  int rc = activate(dev, "twk_debug_capture"); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(dev->stream));
  dev->captureFirstHits = (enable != 0);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_debug_capture")

int twk_debug_read_first_hits(TwkDevice dev, float* tBetaGamma, int* instPrim, size_t numPixels)
try
{
  int rc = activate(dev, "twk_debug_read_first_hits"); if (rc) return rc;
  const size_t n = (size_t) dev->launchWidth * dev->state.resolution[1];
  if (!tBetaGamma || !instPrim || numPixels != n) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_debug_read_first_hits: size mismatch");
  if (!dev->d_firstHit) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_debug_read_first_hits: nothing captured");
  HIP_TRY(hipStreamSynchronize(dev->stream));
  std::vector<float4> h(n); std::vector<int> inst(n);
  HIP_TRY(hipMemcpy(h.data(), dev->d_firstHit, n * sizeof(float4), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(inst.data(), dev->d_firstHitInstance, n * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i)
  {
    tBetaGamma[3 * i] = h[i].x; tBetaGamma[3 * i + 1] = h[i].y; tBetaGamma[3 * i + 2] = h[i].z;
    instPrim[2 * i] = inst[i];
    int prim; memcpy(&prim, &h[i].w, 4);
    instPrim[2 * i + 1] = (inst[i] < 0) ? -1 : prim;
  }
  return TWK_SUCCESS;
}
TWK_CATCH("twk_debug_read_first_hits")

int twk_trace_rays(TwkDevice dev, const float* rays, size_t numRays, int anyHit, float* tBetaGamma, int* ids)
try
{
  int rc = activate(dev, "twk_trace_rays"); if (rc) return rc;
  if (!dev->built) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_trace_rays: twk_build has not been called");
  if (!rays || !tBetaGamma || !ids) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_trace_rays: NULL buffer");
  if (numRays == 0) return TWK_SUCCESS;
  if (!dev->stateSet) { dev->launchWidth = 1; }
  if ((rc = ensureStreams(dev))) return rc;
  refreshParams(dev);
  ScopedDeviceBuffer<float> d_rays, d_out; ScopedDeviceBuffer<int> d_ids;
  HIP_TRY(d_rays.allocate(numRays * 8));
  HIP_TRY(d_out.allocate(numRays * 3));
  HIP_TRY(d_ids.allocate(numRays * 2));
  HIP_TRY(hipMemcpyAsync(d_rays.ptr, rays, numRays * 8 * sizeof(float), hipMemcpyHostToDevice, dev->stream));
  int grid = (int) ((numRays + TWK_TRACE_BLOCK - 1) / TWK_TRACE_BLOCK);
  if (grid > dev->numCUs * TWK_TRACE_WAVES) grid = dev->numCUs * TWK_TRACE_WAVES;
  launchTraceQuery(dev->params, d_rays.ptr, (unsigned int) numRays, anyHit, d_out.ptr, d_ids.ptr, grid, dev->stream);
  HIP_TRY(hipMemcpyAsync(tBetaGamma, d_out.ptr, numRays * 3 * sizeof(float), hipMemcpyDeviceToHost, dev->stream));
  HIP_TRY(hipMemcpyAsync(ids, d_ids.ptr, numRays * 2 * sizeof(int), hipMemcpyDeviceToHost, dev->stream));
  HIP_TRY(hipStreamSynchronize(dev->stream));
  return checkDroppedPushes(dev, "twk_trace_rays");
}
TWK_CATCH("twk_trace_rays")

int twk_debug_trace_queue(TwkDevice dev, const float* closestRays, size_t numClosest, const float* shadowRays, size_t numShadow,
                          float* tBetaGammaSlot, int* instance, int* occluded)
try
{
  int rc = activate(dev, "twk_debug_trace_queue"); if (rc) return rc;
  if (!dev->built) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_debug_trace_queue: twk_build has not been called");
  if ((numClosest && (!closestRays || !tBetaGammaSlot || !instance)) || (numShadow && (!shadowRays || !occluded))) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_debug_trace_queue: NULL buffer");
  for (const DevMaterial& m : dev->materials) if (m.textureCutout) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_debug_trace_queue: geometric query only, not for scenes with cutout opacity");
  const size_t n = std::max(numClosest, numShadow);
  if (n == 0) return TWK_SUCCESS;
  if (n >= ((size_t) 1 << 30)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_debug_trace_queue: too many rays");
  if (!dev->stateSet) { dev->launchWidth = 1; }
  const size_t pixels = (size_t) dev->launchWidth * (size_t) dev->state.resolution[1];
  if ((rc = ensureStreams(dev, (int) std::min<size_t>((n + pixels - 1) / pixels, (size_t) 1 << 30)))) return rc;
  if ((size_t) dev->allocatedPaths < n) return twkSetError(TWK_ERROR_OUT_OF_MEMORY, "twk_debug_trace_queue: path streams too small");
  refreshParams(dev);
  LaunchParams p = dev->params;
  p.numPaths = dev->allocatedPaths; p.batchCount = 1; p.firstHit = nullptr; p.firstHitInstance = nullptr; p.pathTime = nullptr;
  p.stats = dev->statsEnabled ? dev->d_stats : nullptr; // twk_stats_enable: visit counts and the number of rays that overflowed the LDS stack (tests/test_gpu_big_scenes.py)
  // the rays of one bounce: radiance rays in queue 1, the shadow rays "emitted by shade 0" in the shadow queue
  std::vector<float4> org(n), dir(n);
  std::vector<unsigned int> index(n);
  for (size_t i = 0; i < n; ++i) index[i] = (unsigned int) i;
  auto split = [&](const float* rays, size_t count)
  {
    for (size_t i = 0; i < count; ++i)
    {
      const float* r = rays + 8 * i;
      org[i] = make_float4(r[0], r[1], r[2], r[3]); dir[i] = make_float4(r[4], r[5], r[6], r[7]);
    }
  };
  HIP_TRY(hipStreamSynchronize(dev->stream));
  HIP_TRY(hipMemset(dev->d_counters, 0, sizeof(unsigned int) * TWK_COUNTER_WORDS));
  if (numClosest)
  {
    split(closestRays, numClosest);
    HIP_TRY(hipMemcpy(p.rayOrg[1], org.data(), numClosest * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.rayDir[1], dir.data(), numClosest * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.rayPixel[1], index.data(), numClosest * sizeof(unsigned int), hipMemcpyHostToDevice));
    const unsigned int c = (unsigned int) numClosest;
    HIP_TRY(hipMemcpy(dev->d_counters + 1 * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_CLOSEST, &c, sizeof(c), hipMemcpyHostToDevice)); // everything in segment 0
  }
  if (numShadow)
  {
    split(shadowRays, numShadow);
    std::vector<float4> pending(numShadow, make_float4(1.0f, 0.0f, 0.0f, 0.0f));
    HIP_TRY(hipMemcpy(p.shadowOrg, org.data(), numShadow * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.shadowDir, dir.data(), numShadow * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.shadowPixel, index.data(), numShadow * sizeof(unsigned int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.shadowPending, pending.data(), numShadow * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(p.pathRadiance, 0, numShadow * sizeof(float4)));
    const unsigned int c = (unsigned int) numShadow;
    HIP_TRY(hipMemcpy(dev->d_counters + 0 * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_SHADOW, &c, sizeof(c), hipMemcpyHostToDevice));
  }
  launchTrace(p, 1, dev->statsEnabled, false, dev->numCUs * p.traceWaves, dev->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(dev->stream));
  if ((rc = checkDroppedPushes(dev, "twk_debug_trace_queue"))) return rc;
  if (numClosest)
  {
    HIP_TRY(hipMemcpy(tBetaGammaSlot, p.hitRecord, numClosest * sizeof(float4), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(instance, p.hitInstance, numClosest * sizeof(int), hipMemcpyDeviceToHost));
  }
  if (numShadow)
  {
    std::vector<float4> radiance(numShadow);
    HIP_TRY(hipMemcpy(radiance.data(), p.pathRadiance, numShadow * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < numShadow; ++i) occluded[i] = (radiance[i].x == 0.0f) ? 1 : 0; // an unoccluded shadow ray adds its pending contribution (1, 0, 0)
  }
  return TWK_SUCCESS;
}
TWK_CATCH("twk_debug_trace_queue")

int twk_debug_read_acceleration(TwkDevice dev, TwkAccelerationInfo* info, void* wideNodes, void* triangles, void* instances)
try
{
  int rc = activate(dev, "twk_debug_read_acceleration"); if (rc) return rc;
  if (!info) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_debug_read_acceleration: NULL info");
  if (!dev->built) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_debug_read_acceleration: twk_build has not been called");
  info->root = dev->wideRoot1; info->twoLevel = dev->twoLevel ? 1 : 0;
  info->root2 = (dev->wideRoot2 == TWK_BVH_SENTINEL) ? -1 : dev->wideRoot2; info->nodeFloats = 16;
  info->numNodes = (uint64_t) dev->wideNodesTotal; info->numTriangleSlots = dev->totalTriangles; info->numInstances = dev->instances.size(); // 4-ary nodes: the binary nodes' + the two of an 8-wide root
  HIP_TRY(hipStreamSynchronize(dev->stream));
  if (wideNodes) HIP_TRY(hipMemcpy(wideNodes, dev->d_wideQ, sizeof(float) * (size_t) info->nodeFloats * info->numNodes, hipMemcpyDeviceToHost));
  if (triangles) HIP_TRY(hipMemcpy(triangles, dev->d_triangles, sizeof(float4) * 3 * dev->totalTriangles, hipMemcpyDeviceToHost));
  if (instances) HIP_TRY(hipMemcpy(instances, dev->d_instances, sizeof(DevInstance) * dev->instances.size(), hipMemcpyDeviceToHost));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_debug_read_acceleration")

int twk_debug_snapshot_scene(TwkDevice dev, void* launchParams, size_t paramsBytes)
try
{
  int rc = activate(dev, "twk_debug_snapshot_scene"); if (rc) return rc;
  if (!dev->built) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_debug_snapshot_scene: twk_build has not been called");
  if (!launchParams || paramsBytes != sizeof(LaunchParams)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_debug_snapshot_scene: paramsBytes must be sizeof(LaunchParams) = " + std::to_string(sizeof(LaunchParams)));
  HIP_TRY(hipStreamSynchronize(dev->stream));
  refreshParams(dev);
  LaunchParams q = dev->params;
  dev->hostScene.clear();
  rc = TWK_SUCCESS;
  auto host = [&](const void* devicePointer, size_t bytes) -> const void*
  {
    if (!devicePointer || bytes == 0) return nullptr;
    dev->hostScene.emplace_back(bytes);
    if (hipMemcpy(dev->hostScene.back().data(), devicePointer, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = TWK_ERROR_HIP;
    return dev->hostScene.back().data();
  };
  q.nodes          = static_cast<const BvhNode*>(host(dev->d_nodes, sizeof(BvhNode) * dev->totalNodes));
  q.wideQ          = static_cast<const float4*>(host(dev->d_wideQ, sizeof(float4) * 4 * dev->wideNodesTotal));
  q.topNodes       = static_cast<const float4*>(host(dev->d_topNodes, sizeof(float4) * 4 * TWK_TOP_NODES));
  q.topNodes7      = static_cast<const float4*>(host(dev->d_topNodes7, sizeof(float4) * 4 * TWK_TOP_NODES7));
  q.triangles      = static_cast<const float4*>(host(dev->d_triangles, sizeof(float4) * 3 * dev->totalTriangles));
  q.shadeTriangles = static_cast<const float4*>(host(dev->d_shadeTriangles, sizeof(float4) * TWK_SHADE_RECORD * dev->totalTriangles));
  q.instances      = static_cast<const DevInstance*>(host(dev->d_instances, sizeof(DevInstance) * dev->instances.size()));
  q.materials      = static_cast<const DevMaterial*>(host(dev->d_materials, sizeof(DevMaterial) * dev->materials.size()));
  q.lights         = static_cast<const DevLight*>(host(dev->d_lights, sizeof(DevLight) * dev->lights.size()));
  q.camera         = static_cast<const float*>(host(dev->d_camera, sizeof(float) * 12));
  q.attributes = nullptr; q.indices = nullptr; // build input only
  for (int k = 0; k < 3; ++k)
    q.textures[k].texels = static_cast<const float4*>(host(dev->d_texels[k], sizeof(float4) * (size_t) q.textures[k].width * (size_t) q.textures[k].height));
  q.envCDF_U = static_cast<const float*>(host(dev->d_envCDF_U, sizeof(float) * ((size_t) q.envWidth + 1) * q.envHeight));
  q.envCDF_V = static_cast<const float*>(host(dev->d_envCDF_V, sizeof(float) * ((size_t) q.envHeight + 1)));
  if (rc) return twkSetError(rc, "twk_debug_snapshot_scene: device-to-host copy failed");
  // streams, counters, outputs: the host build allocates its own
  q.tileEntries = nullptr; q.tilesX = 0;
  for (int k = 0; k < 2; ++k) { q.rayOrg[k] = nullptr; q.rayDir[k] = nullptr; q.rayPixel[k] = nullptr; q.rayThroughput[k] = nullptr; q.raySeedFlags[k] = nullptr; }
  q.hitRecord = nullptr; q.hitInstance = nullptr; q.shadowOrg = nullptr; q.shadowDir = nullptr; q.shadowPixel = nullptr; q.shadowPending = nullptr;
  q.pathRadiance = nullptr; q.volumeStack = nullptr; q.pathAlbedo = nullptr; q.pathNormal = nullptr; q.aovAlbedo = nullptr; q.aovNormal = nullptr;
  q.output = nullptr; q.counters = nullptr; q.stats = nullptr; q.firstHit = nullptr; q.firstHitInstance = nullptr; q.traceStackSpill = nullptr;
  q.overflowSlots = nullptr; q.droppedPushes = nullptr;
  memcpy(launchParams, &q, sizeof(q));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_debug_snapshot_scene")

int twk_debug_math(TwkDevice dev, int op, const float* x, const float* y, float* out, size_t n)
try
{
  int rc = activate(dev, "twk_debug_math"); if (rc) return rc;
  if (op < 0 || op > 9 || !x || !out || ((op == 3 || op == 9) && !y)) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_debug_math: bad arguments");
  if (n == 0) return TWK_SUCCESS;
  ScopedDeviceBuffer<float> dx, dy, dout;
  HIP_TRY(dx.allocate(n));
  HIP_TRY(dy.allocate(n));
  HIP_TRY(dout.allocate(n));
  HIP_TRY(hipMemcpyAsync(dx.ptr, x, n * sizeof(float), hipMemcpyHostToDevice, dev->stream));
  HIP_TRY(hipMemcpyAsync(dy.ptr, y ? y : x, n * sizeof(float), hipMemcpyHostToDevice, dev->stream));
  launchMathTap(op, dx.ptr, dy.ptr, dout.ptr, n, dev->stream);
  HIP_TRY(hipMemcpyAsync(out, dout.ptr, n * sizeof(float), hipMemcpyDeviceToHost, dev->stream));
  HIP_TRY(hipStreamSynchronize(dev->stream));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_debug_math")

} // extern "C"
