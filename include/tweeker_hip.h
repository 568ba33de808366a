/*
 * tweeker_hip.h — C ABI of the MI355X-native path-tracing hot path.
 *
 * This is the drop-in seam that stands where rtigo3 talks to libnvoptix.so.1:
 *   - the run-time loaded OptiX function table      (reference apps/rtigo3/src/Device.cpp:504-536)
 *   - the per-GPU `Device` object wrapping it       (reference apps/rtigo3/inc/Device.h:292-404)
 * Everything is plain C: opaque handle, POD structs, pointers and sizes. No C++/torch types.
 * Every call returns 0 on success or a TwkResult error code; twk_last_error() returns the text
 * (≙ OptixResult/CUresult + CheckMacros.h:38-80 which throw std::runtime_error; a C ABI never throws).
 *
 * A handle is NOT thread safe; one handle per GPU; all work of a handle is enqueued on its own
 * non-blocking HIP stream (≙ Device.cpp:255) and calls of different handles may be interleaved
 * from one host thread.
 *
 * The library never computes on the CPU: without a usable HIP device every compute entry point
 * fails with TWK_ERROR_NO_DEVICE.
 */
#ifndef TWEEKER_HIP_H
#define TWEEKER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: TwkLaunchStats grew by waveCycles[6] (twk_stats_get writes sizeof(TwkLaunchStats) bytes) and
 * twk_debug_read_acceleration hands out the 64-byte quantised wide nodes instead of 128-byte ones. */
/* 8: TwkBuildInfo grew by wide8Nodes / wide8Levels; TwkAccelerationInfo.reserved became nodeFloats — twk_debug_read_acceleration
 * hands out the compressed 8-ary nodes (20 floats each) where the persistent kernel walks those. */
/* 9: TwkLaunchStats grew by the shade kernel's per-phase tallies; twk_set_next_event_estimation, twk_set_debug_exceptions. */
#define TWK_ABI_VERSION 9

typedef enum TwkResult
{
  TWK_SUCCESS              = 0,
  TWK_ERROR_INVALID_VALUE  = 1,
  TWK_ERROR_NO_DEVICE      = 2, /* no HIP device / HIP runtime failure at create */
  TWK_ERROR_HIP            = 3, /* a HIP call failed, text in twk_last_error() */
  TWK_ERROR_INVALID_STATE  = 4, /* call order violated (e.g. launch before build) */
  TWK_ERROR_OUT_OF_MEMORY  = 5,
  TWK_ERROR_IO             = 6, /* scene/system description file problems */
  TWK_ERROR_PARSE          = 7
} TwkResult;

/* ---- POD layouts shared with the reference (sizes asserted in the implementation) ---------- */

/* ≙ CameraDefinition, reference shaders/camera_definition.h:34-40 (48 B) */
typedef struct TwkCameraDefinition
{
  float P[3];
  float U[3];
  float V[3];
  float W[3];
} TwkCameraDefinition;

/* ≙ LightType, reference shaders/light_definition.h:34-40 */
enum { TWK_LIGHT_ENVIRONMENT = 0, TWK_LIGHT_PARALLELOGRAM = 1 };

/* ≙ LightDefinition, reference shaders/light_definition.h:42-59 (80 B) */
typedef struct TwkLightDefinition
{
  int   type;
  float position[3];
  float vecU[3];
  float vecV[3];
  float normal[3];
  float area;
  float emission[3];
  float unused0, unused1, unused2;
} TwkLightDefinition;

/* ≙ FunctionIndex, reference shaders/function_indices.h:51-60 */
enum
{
  TWK_INDEX_BRDF_DIFFUSE   = 0,
  TWK_INDEX_BRDF_SPECULAR  = 1,
  TWK_INDEX_BSDF_SPECULAR  = 2,
  TWK_INDEX_BRDF_GGX_SMITH = 3,
  TWK_INDEX_BSDF_GGX_SMITH = 4
};

/* ≙ LensShader, reference shaders/function_indices.h:42-49 */
enum { TWK_LENS_SHADER_PINHOLE = 0, TWK_LENS_SHADER_FISHEYE = 1, TWK_LENS_SHADER_SPHERE = 2 };

/* ≙ MaterialGUI without the name, reference inc/MaterialGUI.h:39-52. The device-side
 * MaterialDefinition (absorption coefficient, FLAG_THINWALLED) is derived inside
 * twk_init_materials exactly as Device::initMaterials does (Device.cpp:1022-1050). */
typedef struct TwkMaterialGUI
{
  int   indexBSDF;
  float albedo[3];
  float absorptionColor[3];
  float absorptionScale;
  float ior;
  int   thinwalled;
  int   useAlbedoTexture;
  int   useCutoutTexture;
  float roughness[2];
} TwkMaterialGUI;

/* ≙ TriangleAttributes, reference shaders/vertex_attributes.h:34-40 (48 B) */
typedef struct TwkTriangleAttributes
{
  float vertex[3];
  float tangent[3];
  float normal[3];
  float texcoord[3];
} TwkTriangleAttributes;

/* ≙ DeviceState, reference inc/Device.h:277-290 */
typedef struct TwkDeviceState
{
  int   resolution[2];
  int   tileSize[2];     /* power of two */
  int   pathLengths[2];  /* .x = min length before Russian roulette, .y = max length */
  int   distribution;    /* 1: checkerboard tile distribution over deviceCount devices */
  int   samplesSqrt;
  int   lensShader;
  float epsilonFactor;   /* sceneEpsilon = epsilonFactor * 1e-7 (config.h:42) */
  float envRotation;
  float clockFactor;     /* accepted, unused (USE_TIME_VIEW is 0 in the reference build) */
} TwkDeviceState;

/* Flattening (build option of twk_build, ≙ the accelBuildOptions of Device.cpp:1383-1389): an instance is FLATTENED
 * when its geometry has at most `maxTriangles` triangles (walls, light quads) or is referenced by at most
 * `maxReferences` instances (instancing saves no memory worth a per-ray instance entry). Flattened instances are
 * intersected in WORLD space: their vertices are transformed once at twk_build (row-major 3x4 object-to-world,
 * m0*x + m1*y + m2*z + m3 in fp32, as transformPoint closesthit.cu:88-98 evaluates it) and tested against the
 * untransformed ray, all in one world-space BVH; every other instance is entered through the top level with the ray
 * taken through the inverse transform (≙ IAS→GAS descent, Device.cpp:1427-1445). t, beta, gamma name the same
 * quantities either way; which one applies is part of the traversal contract the CPU oracle restates
 * (oracle/orc_trace.h), so both sides take the same policy. (0, 0) = pure two-level. */
#define TWK_FLATTEN_TRIANGLES  4
#define TWK_FLATTEN_REFERENCES 2

/* Texture slots ≙ the three hard-wired textures of Device::initTextures (Device.cpp:911-942). */
enum { TWK_TEXTURE_ALBEDO = 0, TWK_TEXTURE_CUTOUT = 1, TWK_TEXTURE_ENVIRONMENT = 2 };

/* Per-launch work counters (device side, exact integers). */
typedef struct TwkLaunchStats
{
  uint64_t radianceRays;    /* closest-hit rays traced */
  uint64_t shadowRays;      /* any-hit rays traced */
  uint64_t nodesVisited;    /* 4-ary wide nodes visited (64 B quantised records), both ray kinds */
  uint64_t trianglesTested; /* triangle records fetched (48 B each) */
  uint64_t instancesEntered;
  uint64_t shadedHits;
  uint64_t missed;
  uint64_t maxNodesPerRay;  /* longest single traversal (inner-node visits) seen by the wavefront trace kernel */
  uint64_t tailRays;        /* reserved (0): the persistent tail kernel of rounds 1-4 is an experiment patch now (tools/experiments/) */
  uint64_t tailNodesVisited;
  uint64_t tailTrianglesTested;
  uint64_t tailInstancesEntered;
  uint64_t overflowRays;    /* rays whose LDS traversal stack overflowed and were re-traced with the HBM-backed stack */
  uint64_t nodeWaveSteps;     /* wave-level iterations of the node step: lane occupancy there = nodesVisited / (64 * nodeWaveSteps) */
  uint64_t triangleWaveSteps; /* wave-level iterations of the triangle test */
  uint64_t leafWaveSteps;     /* wave-level executions of the leaf / instance entry / instance exit step */
  uint64_t cachedNodesVisited; /* of nodesVisited: wide nodes served from the LDS top-of-tree cache, not from memory */
  uint64_t droppedStackPushes; /* single-ray fallback traversal: pushes beyond its LDS + HBM stack (a truncated traversal); 0 on every scene tried */
  /* Where the waves of the persistent traversal kernel spend their time: shader-clock cycles (s_memtime, waits included)
   * summed over all waves, per phase of the kernel's outer loop — [0] refill (ray fetch), [1] node loop, [2] leaf /
   * instance step, [3] triangle loop, [4] pop + result write, [5] whole kernel. */
  uint64_t waveCycles[6];
  /* ABI 9. Where the waves of the shade kernel spend their instructions and their time, per phase of the shading of a path segment
   * (TWK_SHADE_PHASE_*): how often a wave ran the phase, with how many of its 64 lanes (lane occupancy of the phase = lanes /
   * (64 x wave steps)), and for how many shader-clock cycles (waits included). */
  uint64_t shadePhaseWaveSteps[24];
  uint64_t shadePhaseLanes[24];
  uint64_t shadePhaseCycles[24];
} TwkLaunchStats;
#define TWK_SHADE_PHASE_COUNT 24
/* index into TwkLaunchStats::shadePhase*: */
enum
{
  TWK_SHADE_PHASE_PATH = 0,          /* the whole shading of a segment */
  TWK_SHADE_PHASE_VOLUME_FETCH = 1,  /* volume stack top of a path inside a medium */
  TWK_SHADE_PHASE_MISS = 2,          /* miss programs (miss.cu) */
  TWK_SHADE_PHASE_HIT_RECORD = 3,    /* instance, shading record, material; normals; front face (closesthit.cu:126-186) */
  TWK_SHADE_PHASE_TANGENT = 4,       /* GGX materials: tangent */
  TWK_SHADE_PHASE_TEXCOORD = 5,      /* textured materials */
  TWK_SHADE_PHASE_LIGHT_HIT = 6,     /* implicit light hit (closesthit.cu:192-222) */
  TWK_SHADE_PHASE_BSDF_DIFFUSE = 7, TWK_SHADE_PHASE_BSDF_MIRROR = 8, TWK_SHADE_PHASE_BSDF_GLASS = 9,
  TWK_SHADE_PHASE_BSDF_GGX = 10, TWK_SHADE_PHASE_BSDF_GGX_GLASS = 11, /* the five sample callables */
  TWK_SHADE_PHASE_NEE_SAMPLE = 12,   /* draws + light sampler (closesthit.cu:252-264) */
  TWK_SHADE_PHASE_NEE_EVAL = 13,     /* BSDF eval + contribution (closesthit.cu:266-299) */
  TWK_SHADE_PHASE_RADIANCE = 14,     /* read-modify-write of the path's radiance */
  TWK_SHADE_PHASE_TAIL = 15,         /* integrator loop tail (raygeneration.cu:91-146) */
  TWK_SHADE_PHASE_VOLUME_PUSH = 16,  /* glass transmission: volume stack push / pop */
  TWK_SHADE_PHASE_AOV = 17,          /* denoiser AOV writes */
  TWK_SHADE_PHASE_KERNEL_LOAD = 18,  /* wait for the queue slot's streams */
  TWK_SHADE_PHASE_KERNEL_APPEND = 19,/* queue appends: ballots, barriers, the block's atomic, the stores (lanes = appending lanes) */
  TWK_SHADE_PHASE_KERNEL_ITERATION = 20, /* one block iteration of the kernel, per wave (lanes = lanes with a queue slot) */
  TWK_SHADE_PHASE_APPEND_BARRIER1 = 21,  /* of the append: from its start to behind the first barrier = the wait for the block's slowest wave */
  TWK_SHADE_PHASE_APPEND_ATOMIC = 22,    /* the round trip of the block's returning atomic, per issuing lane */
  TWK_SHADE_PHASE_APPEND_BARRIER2 = 23   /* from the first barrier to behind the second: stream requests, the atomic, the wait for it */
};

/* Accumulated device time per kernel class since twk_profile_reset (profiling mode only). */
enum
{
  TWK_KERNEL_GENERATE = 0,
  TWK_KERNEL_TRACE    = 1,
  TWK_KERNEL_SHADE    = 2,
  TWK_KERNEL_ACCUM    = 3,
  TWK_KERNEL_TAIL     = 4,  /* reserved: never launched by this build */
  TWK_KERNEL_COUNT    = 5
};

typedef struct TwkDevice_t* TwkDevice;

/* ---- seam 1/2: the per-GPU renderer ------------------------------------------------------- */

const char* twk_last_error(void);
int twk_abi_version(void);
int twk_device_count(int* count);

/* ≙ Device::Device(strategy, ordinal, index, count, miss, ...) — Device.cpp:222-317.
 * ordinal: HIP device ordinal. index/count: position in the set of rendering devices
 * (tile distribution). miss: 0 = black, 1 = constant white env, 2 = spherical HDR env
 * (selects the miss program like Device.cpp:660-672). */
int twk_device_create(TwkDevice* out, int ordinal, int index, int count, int miss);
int twk_device_destroy(TwkDevice dev); /* ≙ Device::~Device, Device.cpp:320-358 */

int twk_set_state(TwkDevice dev, const TwkDeviceState* state);                  /* ≙ Device::setState      Device.cpp:1192-1256 */
int twk_init_cameras(TwkDevice dev, const TwkCameraDefinition* c, int count);   /* ≙ Device::initCameras   Device.cpp:944-968 */
int twk_init_lights(TwkDevice dev, const TwkLightDefinition* l, int count);     /* ≙ Device::initLights    Device.cpp:970-1000 */
int twk_init_materials(TwkDevice dev, const TwkMaterialGUI* m, int count);      /* ≙ Device::initMaterials Device.cpp:1002-1056 */
int twk_update_camera(TwkDevice dev, int idCamera, const TwkCameraDefinition* c);   /* ≙ Device::updateCamera   Device.cpp:1083-1096 */
int twk_update_light(TwkDevice dev, int idLight, const TwkLightDefinition* l);      /* ≙ Device::updateLight    Device.cpp:1098-1110 */
int twk_update_material(TwkDevice dev, int idMaterial, const TwkMaterialGUI* m);    /* ≙ Device::updateMaterial Device.cpp:1112-1168 */

/* ≙ Device::initTextures (Device.cpp:911-942). Texels are RGBA32F, row 0 = v 0 (origin lower left),
 * bilinear, normalized coordinates; wrap in u and v except the environment which clamps v
 * (Texture.cpp:668-693,1353). For TWK_TEXTURE_ENVIRONMENT the spherical CDFs and the integral
 * are computed like Texture::calculateSphericalCDF (Texture.cpp:1500-1645). */
int twk_init_texture(TwkDevice dev, int slot, const float* rgba, int width, int height);

/* Scene ≙ Device::initScene → traverseNode (Device.cpp:1058-1080,1283-1331), flattened by the caller. */
int twk_add_geometry(TwkDevice dev, const TwkTriangleAttributes* attributes, size_t numAttributes,
                     const unsigned int* indices, size_t numIndices, int* idGeometry); /* ≙ createGeometry (GAS) Device.cpp:1333-1425 */
int twk_add_instance(TwkDevice dev, int idGeometry, const float transform[12],
                     int idMaterial, int idLight, int* idInstance);                    /* ≙ createInstance Device.cpp:1427-1445 + hit record :1492-1532 */
int twk_build(TwkDevice dev);                                                          /* ≙ createTLAS + createHitGroupRecords Device.cpp:1448-1532 */
int twk_clear_scene(TwkDevice dev);
/* Acceleration-structure quality of the next twk_build (≙ the buildFlags of accelBuildOptions, Device.cpp:1383-1389):
 * TWK_BUILD_LBVH — Morton codes + radix tree, the fastest build; TWK_BUILD_SAH (default) — binned surface-area-heuristic
 * top-down splits, fewer node visits per ray. Hit records do not depend on it (closest hit is order independent). */
enum { TWK_BUILD_LBVH = 0, TWK_BUILD_SAH = 1 };
int twk_set_build_quality(TwkDevice dev, int quality);

/* What the last twk_build produced. SAH cost terms: over every bottom-level / flattened-instance tree, the sum of
 * half-area(node) / half-area(root of its tree) over the inner nodes that survive the leaf collapse (sahInnerCost)
 * and of half-area(leaf) / half-area(root) x triangles over its leaves (sahLeafCost): expected binary-node visits and
 * triangle tests of a random ray that hits the root box, summed over `trees` trees. */
typedef struct TwkBuildInfo
{
  int      quality;
  int      trees;
  double   sahInnerCost;
  double   sahLeafCost;
  double   buildMilliseconds; /* host wall time of twk_build, uploads included */
  uint64_t triangleSlots, nodes, instances, flattenedInstances;
  uint64_t maxTraversalDepth; /* ABI 4: binary-tree levels of the deepest root-to-leaf path (top level + the deepest tree below it); twk_build refuses a scene deeper than the traversal stacks */
  uint64_t directLeafInstances; /* ABI 5: flattened instances of at most a leaf's triangles that ARE leaves of the top level (no tree of their own is visited) */
  uint64_t traceBlocksPerCU;    /* ABI 5: resident blocks per CU of the persistent traversal kernel for this scene with the materials as they are now: 7 (every instance flattened, at most 1 M nodes — with or without cutout opacity), 5 (two-level with cutout opacity), else 6 */
  uint64_t wide8Nodes;          /* ABI 8: reserved, 0 — the compressed 8-ary nodes of round 4 lost on every scene and are an experiment patch now (tools/experiments/r04_wide8_nodes.patch) */
  uint64_t wide8Levels;         /* ABI 8: reserved, 0 */
} TwkBuildInfo;
int twk_get_build_info(TwkDevice dev, TwkBuildInfo* info);

/* Flattening policy of the next twk_build (defaults TWK_FLATTEN_TRIANGLES, TWK_FLATTEN_REFERENCES; see there). */
int twk_set_flatten_policy(TwkDevice dev, int maxTriangles, int maxReferences);

/* ≙ Device*::render(iterationIndex, buffer) → optixLaunch(pipeline, stream, d_sys, 192, &sbt, W, H, 1)
 * (DeviceSingleGPU.cpp:104-182; multi-GPU: DeviceMultiGPULocalCopy.cpp:104-190).
 * Asynchronous on the handle's stream. One call = one sample per pixel of this device's share:
 * the full W×H frame (distribution 0) or the launchWidth×H checkerboard tile set (distribution 1,
 * raygeneration.cu:152-164,259-344). The accumulation buffer holds the running mean
 * (raygeneration.cu:246-253), RGBA32F, alpha 1. */
int twk_launch(TwkDevice dev, unsigned int iterationIndex);
int twk_sync(TwkDevice dev);                                  /* ≙ Device::synchronizeStream */
/* twk_launch is asynchronous and deferred: consecutive iteration indices are rendered together, up to `iterations`
 * samples per pixel per wavefront pass (1..64, default 64; 280 bytes of path streams per sample and pixel), as soon
 * as the batch is full or any other call observes the device. The image is bit-identical to one pass per iteration; 1 restores strict one-launch-per-call behaviour. */
int twk_set_launch_batch(TwkDevice dev, int iterations);
/* The path streams of a pass are allocated on demand and grow with the largest pass seen; this allocates them up
 * front for passes of `iterations` samples per pixel, so that no allocation falls into a timed or interactive loop. */
int twk_reserve_launch_batch(TwkDevice dev, int iterations);

/* The two apps the hot path serves differ in ONE rule of __closesthit__radiance: rtigo3 ends a path on a light only
 * when its lit side is hit and lets a back-face hit fall through to the light's BSDF (apps/rtigo3/shaders/closesthit.cu:192-222);
 * Optix7Gui (intro_07's app) ends the path on either side, black on the back face (apps/Optix7Gui/shaders/closesthit.cu:189-226). */
enum { TWK_SHADERS_RTIGO3 = 0, TWK_SHADERS_OPTIX7GUI = 1 };
int twk_set_shader_variant(TwkDevice dev, int variant);

/* Denoiser AOVs of Optix7Gui's integrator (apps/Optix7Gui/shaders/raygeneration.cu:125-164,239-262), the input the
 * OptiX AI denoiser is fed with (the denoiser itself is closed third-party code and not part of this library):
 * TWK_AOV_ALBEDO: throughput-attenuated albedo of the first diffuse or light event, clamped to [0, 1], alpha 1;
 * TWK_AOV_NORMAL: shading normal of the primary hit in right-handed camera space, renormalised running mean, w 0.
 * Both accumulate like the radiance (running mean, skipped with it when a sample is NaN). Layout as twk_read_output. */
enum { TWK_AOV_ALBEDO = 0, TWK_AOV_NORMAL = 1 };
int twk_enable_aov(TwkDevice dev, int enable);
int twk_read_aov(TwkDevice dev, int which, float* rgbaHost, size_t numFloats);

/* ABI 6. Time view, ≙ the reference's compile-time USE_TIME_VIEW (apps/rtigo3/shaders/config.h:60, raygeneration.cu:169-171,
 * 231-244, Device.h:350): while enabled the ALPHA of the accumulation buffer is not 1 but the running mean of
 * (shader-clock cycles the sample's lanes spent in traversal and shading) x TwkDeviceState::clockFactor x 1e-9 — what
 * rtigo3's rasteriser maps through its colour ramp. RGB is unchanged, bit for bit. The reference counts one thread's
 * clock() from ray generation to the write; a wavefront path has no single thread, so the cycles of its lanes in every
 * traversal and shade launch are summed (lanes of a wave wait for each other in both designs). Measurement builds of the
 * kernels run while it is on (as with twk_stats_enable): it is a diagnostic view, not a fast path. */
int twk_set_time_view(TwkDevice dev, int enable);

/* ABI 9. ≙ the reference's compile-time lighting switch USE_NEXT_EVENT_ESTIMATION (shaders/config.h:50-52), a run-time switch
 * here: 1 (default) = next-event estimation per path vertex with power-heuristic MIS; 0 = brute-force path tracing — no light
 * sample, no shadow ray (closesthit.cu:250-304), implicit light and environment hits unweighted (closesthit.cu:202-214,
 * miss.cu:62-68,92-106). Both estimate the same image; the reference keeps the switch "to compare lighting results". */
int twk_set_next_event_estimation(TwkDevice dev, int enable);
/* ABI 9. ≙ USE_DEBUG_EXCEPTIONS of the ray generation program (config.h:54-56, raygeneration.cu:205-218): 1 = a sample that is
 * NaN / infinite / negative is accumulated as super red / green / blue (1e6) instead of NaN samples being dropped; 0 (default). */
int twk_set_debug_exceptions(TwkDevice dev, int enable);

/* Output. With distribution 0 the buffer is W×H (≙ outputBuffer); with distribution 1 it is the
 * packed launchWidth×H local tile buffer (≙ texelBuffer, DeviceMultiGPULocalCopy.cpp:109-172). */
int twk_get_launch_width(TwkDevice dev, int* launchWidth);    /* ≙ m_launchWidth, DeviceMultiGPULocalCopy.cpp:84-97 */
int twk_read_output(TwkDevice dev, float* rgbaHost, size_t numFloats); /* ≙ getOutputBufferHost, sync D2H */
int twk_get_output_device_pointer(TwkDevice dev, void** dptr, size_t* bytes);
/* Let the caller own the accumulation buffer (device memory of ≥ launchWidth*H*16 B, e.g. a
 * torch tensor used as RCCL send buffer). Pass NULL to return to the internal buffer. */
int twk_set_output_device_pointer(TwkDevice dev, void* dptr, size_t bytes);

/* The reference's two other multi-GPU buffer strategies (≙ DeviceMultiGPUZeroCopy.cpp:106-118: one pinned host buffer
 * mapped into every device; DeviceMultiGPUPeerAccess.cpp:110-158: one buffer on the first device, written by its peers):
 * every device accumulates straight into ONE shared W x H RGBA32F frame at the pixel its launch index maps to
 * (raygeneration.cu:175-183,229), no packed tile buffers, no compositor. `frame` must be addressable from this device
 * (hipHostMalloc(..., hipHostMallocPortable | hipHostMallocMapped), or device memory with peer access enabled);
 * devices write disjoint pixels. NULL returns to the internal packed buffer. twk_read_output then returns the frame. */
int twk_set_shared_frame(TwkDevice dev, void* frame, size_t bytes);

/* ≙ DeviceMultiGPULocalCopy::compositor + compositor.cu:38-64, for all source devices in one kernel:
 * `tiles` is the gathered [deviceCount][H][launchWidth] RGBA32F block (device memory, rank order),
 * `output` the full W×H RGBA32F image (device memory). Runs on this handle's stream. */
int twk_compositor(TwkDevice dev, const void* tiles, void* output);

/* ≙ TonemapperGUI (inc/TonemapperGUI.h:34-43), same field order. Neutral defaults: gamma 1, whitePoint 1,
 * colorBalance 1 1 1, burnHighlights 1, crushBlacks 0, saturation 1, brightness 1 (Application.cpp:111-120). */
typedef struct TwkTonemapper
{
  float gamma;
  float whitePoint;
  float colorBalance[3];
  float burnHighlights;
  float crushBlacks;
  float saturation;
  float brightness;
} TwkTonemapper;

/* ≙ the tonemapper loop of Application::screenshot (Application.cpp:2259-2297), as the device kernel its authors
 * ask for there: RGBA32F → RGB8, pixel i at rgb8Host[3*i..3*i+2], same row order as the input.
 * rgbaDevice NULL: the handle's own accumulation buffer (numPixels must be launchWidth*height); otherwise any
 * device buffer of numPixels float4 (e.g. the composited multi-GPU image). Synchronises the handle's stream. */
int twk_tonemap(TwkDevice dev, const TwkTonemapper* tm, const void* rgbaDevice, size_t numPixels, unsigned char* rgb8Host);

/* ---- measurement -------------------------------------------------------------------------- */
int twk_profile_enable(TwkDevice dev, int enable);   /* hipEvent pair around every kernel launch */
int twk_profile_reset(TwkDevice dev);
int twk_profile_get(TwkDevice dev, float msPerKernelClass[TWK_KERNEL_COUNT], int launchesPerKernelClass[TWK_KERNEL_COUNT]);
int twk_stats_enable(TwkDevice dev, int enable);     /* counting kernel variants (not for timed runs) */
int twk_stats_get(TwkDevice dev, TwkLaunchStats* stats, int reset);
int twk_stream_peak_gbps(TwkDevice dev, size_t bytes, int repeats, float* gbps); /* float4 copy kernel */
/* Divergent-gather ceiling: every lane of every wave chases its own chain of 128-byte lines through a table of
 * tableBytes and reads each line as eight 16-byte loads (the access pattern of a wide-node fetch); returns giga
 * lane-loads (16 B each) per second. With a table the size of the scene this is the memory-side ceiling of traversal. */
int twk_gather_peak(TwkDevice dev, size_t tableBytes, float* gigaLaneLoadsPerSecond);

/* ---- debugging / parity taps (stage-level SoA read-back after one launch) ------------------ */
/* First-bounce hit record per pixel of the last launch: t, beta, gamma, instance, primitive.
 * Requires twk_debug_capture(dev, 1) before the launch. prim/inst = -1 on miss. */
int twk_debug_capture(TwkDevice dev, int enable);
int twk_debug_read_first_hits(TwkDevice dev, float* tBetaGamma /*3 per px*/, int* instPrim /*2 per px*/, size_t numPixels);

/* Closest-hit / any-hit query of arbitrary rays through the device BVH (≙ optixTrace contract,
 * raygeneration.cu:84-89, closesthit.cu:281-286). rays: 8 floats each (o.xyz, tmin, d.xyz, tmax).
 * out: t, beta, gamma per ray; ids: instance, primitive (or -1). anyHit != 0: ids[0] = 1 if occluded.
 * Geometric query: cutout opacity is not applied here. */
int twk_trace_rays(TwkDevice dev, const float* rays, size_t numRays, int anyHit, float* tBetaGamma, int* ids);

/* The same query through the PERSISTENT traversal kernel of the hot path, as ONE bounce's launch sees it: closestRays go
 * into the radiance ray queue, shadowRays into the shadow queue (either may be empty; 8 floats per ray as above).
 * tBetaGammaSlot: t, beta, gamma and the bits of the hit triangle's slot per closest ray (the hit record shade reads;
 * twk_debug_read_acceleration maps a slot to its primitive); instance: -1 on a miss; occluded: 1 / 0 per shadow ray.
 * Overwrites the handle's path streams; not for scenes with cutout opacity. */
int twk_debug_trace_queue(TwkDevice dev, const float* closestRays, size_t numClosest, const float* shadowRays, size_t numShadow,
                          float* tBetaGammaSlot, int* instance, int* occluded);

/* Read-back of the acceleration structure twk_build produced, for the same-BVH host walker of the test tooling
 * (oracle/same_bvh_walk.cpp: visit counts and a one-core traversal rate on exactly the tree the kernels walk).
 * Two-call protocol: with NULL buffers only `info` is filled. wideNodes: numNodes x 64 B, the quantised 4-ary nodes the
 * persistent kernel walks (four float4: origin.xyz, cell.x | cell.y, cell.z, qlo.x, qlo.y | qlo.z, qhi.x, qhi.y, qhi.z |
 * four references; q words hold one byte per child, child box = origin + q * cell, an unused entry has the inverted box
 * lo 255 / hi 0; reference >= 0 inner node, < 0 leaf with payload ~ref = instance index, or first slot | (count - 1) << 28
 * [| 0x40000000 for world-space slots]);
 * triangles: numTriangleSlots x 48 B (three float4: vertex, .w = primitive index / instance index / 0);
 * instances: numInstances x 128 B (world-to-object 3x4, BVH root, ..., see csrc/device_types.h DevInstance). */
typedef struct TwkAccelerationInfo
{
  int      root;      /* node index traversal starts at */
  int      twoLevel;  /* 0: every instance is flattened, no instance reference occurs */
  uint64_t numNodes, numTriangleSlots, numInstances;
  int      root2;     /* ABI 7: the second wide node of an 8-wide root (a ray starts at `root` with `root2` on its stack), -1: none */
  int      nodeFloats; /* ABI 8: floats per node record: 16 = quantised 4-ary node (64 B), 20 = compressed 8-ary node (80 B, root = node 0, csrc/device_types.h) */
} TwkAccelerationInfo;
int twk_debug_read_acceleration(TwkDevice dev, TwkAccelerationInfo* info, void* wideNodes, void* triangles, void* instances);

/* Host copy of everything the kernels read of the scene, for the host build of the kernels (oracle/host_kernels.cpp: the
 * north_star's "single-threaded C++ CPU fallback of the same kernels", test infrastructure like the oracle): writes the
 * handle's launch parameters (csrc/device_types.h LaunchParams, `paramsBytes` must equal its size) with every SCENE pointer
 * (binary nodes, triangle slots, shading records, instances, materials, lights, camera, textures, environment tables)
 * replaced by a pointer into host memory owned by the handle (valid until the next twk_build / twk_debug_snapshot_scene /
 * twk_device_destroy); the path streams, counters and output pointers are null. Nothing in the product reads it back. */
int twk_debug_snapshot_scene(TwkDevice dev, void* launchParams, size_t paramsBytes);

/* Unit taps of the device math used by the shaders (bit-exact parity with the oracle):
 * op 0 sin, 1 cos, 2 exp, 3 atan2(x[i], y[i]), 4 acos, 5 atan, 6 sqrt, 7 1/x, 8 log, 9 pow(x[i], y[i]). */
int twk_debug_math(TwkDevice dev, int op, const float* x, const float* y, float* out, size_t n);

/* ---- host scene layer (rtigo3 Application: description files, meshes, camera) -------------- */
typedef struct TwkApp_t* TwkApp;

/* ≙ Application::loadSystemDescription (Application.cpp:1046-1299) + createCameras/createLights
 * (:562-677) + loadSceneDescription (:1397-1878). */
int twk_app_create(TwkApp* out, const char* systemDescriptionFile, const char* sceneDescriptionFile);
int twk_app_create_from_strings(TwkApp* out, const char* systemDescription, const char* sceneDescription);
int twk_app_destroy(TwkApp app);

typedef struct TwkAppInfo
{
  int   strategy, devicesMask, light, miss, lensShader, samplesSqrt;
  int   resolution[2], tileSize[2], pathLengths[2];
  float epsilonFactor, envRotation, clockFactor;
  float center[3], phi, theta, fov, distance;
  int   numCameras, numLights, numMaterials, numGeometries, numInstances;
  int   shaderVariant; /* "shaderVariant" of the system description (grammar extension): TWK_SHADERS_*; applied by twk_app_init_device */
  int   nextEventEstimation; /* ABI 9: "nextEventEstimation 0|1" (grammar extension ≙ USE_NEXT_EVENT_ESTIMATION), default 1; applied by twk_app_init_device */
  int   debugExceptions;     /* ABI 9: "debugExceptions 0|1" (grammar extension ≙ USE_DEBUG_EXCEPTIONS), default 0 */
} TwkAppInfo;

int twk_app_info(TwkApp app, TwkAppInfo* info);
int twk_app_set_resolution(TwkApp app, int width, int height); /* re-derives the camera frustum (aspect) */
int twk_app_get_state(TwkApp app, TwkDeviceState* state);
int twk_app_get_cameras(TwkApp app, TwkCameraDefinition* out, int capacity);
int twk_app_get_lights(TwkApp app, TwkLightDefinition* out, int capacity);
int twk_app_get_materials(TwkApp app, TwkMaterialGUI* out, int capacity);
int twk_app_get_geometry_sizes(TwkApp app, int idGeometry, size_t* numAttributes, size_t* numIndices);
int twk_app_get_geometry(TwkApp app, int idGeometry, TwkTriangleAttributes* attributes, unsigned int* indices);
/* Flattened instance list in traverseNode order (Device.cpp:1283-1331). */
int twk_app_get_instance(TwkApp app, int idInstance, int* idGeometry, float transform[12], int* idMaterial, int* idLight);
/* Runs the reference's init sequence on a device: setState, initCameras, initLights, initMaterials,
 * initScene (Application.cpp:303,328-332). */
int twk_app_init_device(TwkApp app, TwkDevice dev);
/* ≙ the text Application::saveSystemDescription writes (Application.cpp:1300-1345): the current settings in the
 * loader's grammar. Two-call protocol: out == NULL returns the length (without the terminator) in *length. */
int twk_app_system_description(TwkApp app, char* out, size_t capacity, size_t* length);
/* Tonemapper settings of the system description ("gamma", "colorBalance", "whitePoint", "burnHighlights",
 * "crushBlacks", "saturation", "brightness", Application.cpp:1244-1292). */
int twk_app_get_tonemapper(TwkApp app, TwkTonemapper* tm);
/* ≙ the file name Application::screenshot builds (Application.cpp:2235-2239, getDateTime :1927-2010):
 * <prefixScreenshot>_<spp>spp_<YYYMMDD_HHMMSS_mmm>.png|.hdr (tm_year and tm_mon as the reference prints them). */
int twk_app_screenshot_path(TwkApp app, int tonemap, char* out, size_t capacity);

/* Image files written by Application::screenshot through DevIL (Application.cpp:2251-2320), without DevIL:
 * 8-bit RGB PNG (stored deflate blocks) and Radiance RGBE .hdr (flat scanlines). `bottomUp` != 0: row 0 of the
 * buffer is the BOTTOM row of the picture (the renderer's convention, IL_ORIGIN_LOWER_LEFT). */
/* ≙ Picture::load + the format expansion of Texture::create* (Picture.cpp:231-560, Texture.cpp:933-1042): decode an
 * image file to RGBA32F, row 0 = bottom row, ready for twk_init_texture. PNG, baseline JPEG (libjpeg's default
 * decode path, byte-exact), Radiance .hdr, PFM — DevIL is not available. Two-call protocol: with rgba == NULL only width/height are returned; otherwise
 * capacityFloats must be >= width*height*4. */
int twk_load_image(const char* path, int* width, int* height, float* rgba, size_t capacityFloats);
/* File name given with "envMap" in the system description (Application.cpp:1151-1156), "" if none. */
int twk_app_get_environment(TwkApp app, char* out, size_t capacity);
int twk_write_png_rgb8(const char* path, int width, int height, const unsigned char* rgb8, int bottomUp);
int twk_write_hdr_rgba32f(const char* path, int width, int height, const float* rgba, int bottomUp);

/* Stand-alone host helpers (≙ sg::Triangles::create*, Camera::getFrustum, calculateTileShift). */
int twk_mesh_plane(unsigned int tessU, unsigned int tessV, unsigned int upAxis, TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx);
int twk_mesh_box(TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx);
int twk_mesh_sphere(unsigned int tessU, unsigned int tessV, float radius, float maxTheta, TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx);
int twk_mesh_torus(unsigned int tessU, unsigned int tessV, float innerRadius, float outerRadius, TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx);
int twk_mesh_parallelogram(const float position[3], const float vecU[3], const float vecV[3], const float normal[3], TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx);
int twk_camera_frustum(const float center[3], float phi, float theta, float fov, float distance, float aspect, TwkCameraDefinition* out); /* ≙ Camera::getFrustum Camera.cpp:187-216 */
/* Tile map ≙ distribute() raygeneration.cu:152-164: launch column → pixel column. */
int twk_tile_column(int launchX, int launchY, const int tileSize[2], int deviceCount, int deviceIndex, int* pixelX);
int twk_launch_width(int width, int tileSizeX, int deviceCount, int* launchWidth); /* ≙ DeviceMultiGPULocalCopy.cpp:84-97 */
/* Tokenise description text like Parser::getNextToken (Parser.cpp:72-148): writes "<type> <token>\n" per token
 * (type 1 = identifier, 2 = value) into out (NUL terminated), returns the count in numTokens. */
int twk_parse_tokens(const char* text, char* out, size_t capacity, int* numTokens);

#ifdef __cplusplus
}
#endif

#endif /* TWEEKER_HIP_H */
