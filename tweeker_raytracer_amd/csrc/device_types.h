// Device-side data layout of the wavefront renderer (all buffers live in HBM, SoA, 16-byte records).
//
// Where OptiX keeps one 128-byte PerRayData per thread in local memory (shaders/per_ray_data.h:84-114)
// and recurses raygen → trace → closesthit → trace(shadow), this renderer streams fixed-size records
// between kernels:
//   ray queue      32 B  (origin.xyz, tmin | direction.xyz, tmax) + 4 B launch index
//   hit record     16 B  (t, beta, gamma, triangle slot) + 4 B instance
//   path state     in the ray queue (per slot): throughput+pdf 16 B, seed+flags 8 B — streamed with the ray;
//                  per path: radiance 16 B, volume stack 64 B (touched only by additions / glass transmission)
//   shadow queue   ray 32 B + 4 B launch index + pending contribution 16 B
//   BVH2 node      64 B  (two child boxes + two child references)
//   triangle       48 B  (three float4: vertex, w of the first = primitive index)
#pragma once
#include "device_math.h"
#include "../../include/tweeker_hip.h" // TWK_SHADERS_*, TWK_FLATTEN_*

namespace twk {

// shaders/config.h:38-46
static const float RT_DEFAULT_MAX      = 1.e27f;
static const float SCENE_EPSILON_SCALE = 1.0e-7f;
static const float DENOMINATOR_EPSILON = 1.0e-6f;

// shaders/per_ray_data.h:39-71
#define TWK_MATERIAL_STACK_EMPTY (-1)
#define TWK_MATERIAL_STACK_FIRST 0
#define TWK_MATERIAL_STACK_LAST  3
#define TWK_MATERIAL_STACK_SIZE  4
#define TWK_FLAG_HIT          0x00000001u
#define TWK_FLAG_SHADOW       0x00000002u
#define TWK_FLAG_DIFFUSE      0x00000004u
#define TWK_FLAG_LIGHT        0x00000008u // light or environment "hit" (Optix7Gui per_ray_data.h:49: 0x4 there, where DIFFUSE is 0x8; only tested together with DIFFUSE)
#define TWK_FLAG_FRONTFACE    0x00000010u
#define TWK_FLAG_THINWALLED   0x00000020u
#define TWK_FLAG_TRANSMISSION 0x00000100u
#define TWK_FLAG_VOLUME       0x00001000u
#define TWK_FLAG_ALBEDO       0x10000000u // the path has written its denoiser albedo (Optix7Gui per_ray_data.h:67), persistent
#define TWK_FLAG_TERMINATE    0x80000000u
#define TWK_FLAG_CLEAR_MASK   (TWK_FLAG_DIFFUSE | TWK_FLAG_ALBEDO) // rtigo3 per_ray_data.h:71 keeps DIFFUSE; Optix7Gui :76 also ALBEDO (never set without AOVs)

// Path word packed next to the seed: bit 2 = FLAG_DIFFUSE of the last interaction (the only flag that
// survives FLAG_CLEAR_MASK, raygeneration.cu:66), bits 8-15 depth, bits 16-18 volume stack index + 1.
#define TWK_PATH_DEPTH_SHIFT 8
#define TWK_PATH_STACK_SHIFT 16
// Packed queue records (LaunchParams::packedQueue): origin.w = launch index (27 bits) | FLAG_DIFFUSE << 27 | FLAG_ALBEDO << 28 |
// (volume stack index + 1) << 29; direction.w = the LCG state.
#define TWK_PACKED_PIXEL_BITS 27
#define TWK_PACKED_PIXEL_MASK ((1u << TWK_PACKED_PIXEL_BITS) - 1u)

// ≙ MaterialDefinition (shaders/material_definition.h:37-56), 64 B
struct DevMaterial
{
  float albedo[3];     float ior;
  float absorption[3]; unsigned int flags;
  float roughness[2];  int indexBSDF; int textureAlbedo; // texture: 0 = none, else slot + 1
  int   textureCutout; int pad0, pad1, pad2;
};

// ≙ LightDefinition (shaders/light_definition.h:42-59), 80 B
struct DevLight
{
  int   type;
  float position[3];
  float vecU[3];
  float vecV[3];
  float normal[3];
  float area;
  float emission[3];
  float unused0, unused1, unused2;
};

// ≙ OptixInstance + hit-group record (src/Device.cpp:1427-1445,1492-1532), 128 B.
// The first 64 bytes are everything traversal needs (four 16-byte loads); shading reads the rest.
struct DevInstance
{
  float worldToObject[12];
  int   blasRoot;      // node index of the geometry's BVH root in the shared node array
  int   triangleFirst; // first triangle slot of the geometry (flattened instance: of its own world-space slots)
  int   triangleCount; // <= TWK_FLATTEN_TRIANGLES: the instance is flattened, traversal never enters it (see TWK_LEAF_WORLD)
  int   geometry;
  float objectToWorld[12];
  int   material;
  int   light;         // < 0: not a light
  unsigned int attributeBase; // first TriangleAttributes of the geometry in the shared attribute array
  unsigned int indexBase;     // first index (uint) of the geometry in the shared index array
};

// FLATTENED instances (include/tweeker_hip.h twk_set_flatten_policy: tiny geometries — planes, the light quad — and
// geometries referenced so rarely that instancing saves nothing): the instance's triangles are written in WORLD space
// into slots of its own (vertex .w of the second float4 = instance index) with an LBVH of its own, whose root is
// spliced into the top level as an inner node. Its leaves carry TWK_LEAF_WORLD in the payload and are tested with
// the world-space ray — no ray transform, no per-instance Woop constants, no descent, no exit step.
#define TWK_LEAF_WORLD 0x40000000
#define TWK_SHADE_RECORD 8 // float4 per shading record

struct DevTexture
{
  const float4* texels;
  int width, height;
  int clampV;
  int pad;
};

// BVH2 node, 64 B. Child reference: >= 0 inner node index; < 0 leaf, payload = ~ref:
// bottom level: bits 0-27 first triangle slot in the reordered triangle array, bits 28-29 triangle count - 1;
// top level: instance index; world-space tree of a flattened instance: TWK_LEAF_WORLD | the same slot-range packing.
struct __attribute__((aligned(16))) BvhNode
{
  float lo0[3]; float hi0x;
  float hi0yz[2]; float lo1xy[2];
  float lo1z; float hi1[3];
  int   child0, child1, pad0, pad1;
};

#define TWK_BVH_SENTINEL 0x7fffffff

// Quantised wide node of the persistent trace kernel, 64 bytes = four float4 (bvh_build.hip quantizeWideKernel):
//   [0] origin.xyz, cell.x      origin = lower corner of the union of the child boxes, cell = a power of two per axis
//   [1] cell.y, cell.z, qlo.x, qlo.y     q words: one byte per child (child k in bits 8k..8k+7)
//   [2] qlo.z, qhi.x, qhi.y, qhi.z       child box = origin + q * cell, rounded outwards (lo down, hi up)
//   [3] the four child references; an unused entry has the inverted box lo = 255, hi = 0 (its near planes lie behind
//       its far planes for every ray) and repeats the first child's reference
// Why 64 instead of the 128 bytes of full-precision boxes: half the lane loads per visit (a CU's vector memory path takes
// ONE divergent 16-byte lane address per clock, tools/gather_probe2.hip) and half the node array (scenes of millions of
// triangles are not cache-resident), and — what decides it on the cache-resident scenes, where the kernel is bound by
// vector-instruction issue — the four children of a plane family share a word, so near and far planes are picked by the
// ray's sign for all four at once (trace_kernels.hip). The boxes only cull, so their precision is free to trade: the grid
// is 1/254 of the node's extent per axis, 1 % more node visits and 10 % more triangle tests on C2.

// Top-of-tree cache of the persistent trace kernel: the first TWK_TOP_NODES wide nodes in breadth-first order from the
// root (every ray visits them) are copied into LDS by each block. A reference TWK_NODE_CACHED | slot names a cached
// node; only the cached copies and LaunchParams::topRoot carry such references, the node arrays in HBM never do.
// The same fetch from LDS (80-byte slot stride: 16 consecutive slots fall into disjoint bank windows) costs a fraction
// of the lane-address slots. Measured on C2 with 128-byte nodes: 16 slots -8.7 % kernel time, 32 slots a further -1.2 %.
#define TWK_NODE_CACHED 0x20000000
#ifndef TWK_TOP_NODES
#define TWK_TOP_NODES 64
#endif
#define TWK_TOP_STRIDE 5 // float4 per cached node (4 used)

// Everything a kernel needs; passed by value (≙ SystemData, shaders/system_data.h:40-90).
struct LaunchParams
{
  // scene
  const BvhNode*     nodes;          // binary nodes (single-ray traversal: query kernel, overflow fallback, tail kernel)
  const float4*      topNodes;       // TWK_TOP_NODES x 4 float4: the cached top of the tree (device_types.h TWK_NODE_CACHED), built by twk_build
  const float4*      topNodes7;      // the same for the seven-blocks-per-CU variant: TWK_TOP_NODES7 nodes, references among THEM rewritten
  int                topRoot;        // reference the persistent kernel starts at: TWK_NODE_CACHED | 0, or tlasRoot when the cache is off
  int                topRoot2;       // the second node of an 8-wide root (bvh_build.hip wideRootKernel), on every ray's stack at its start; TWK_BVH_SENTINEL: none
  const float4*      wideQ;          // quantised 4-ary nodes, 64 bytes = 4 float4 per inner node index (persistent trace kernel; layout above)
  const float4*      triangles;      // 3 per triangle slot: the vertices, .w of the first = primitive id, of the second = instance (world-space slots)
  const float4*      shadeTriangles; // TWK_SHADE_RECORD (8) per triangle slot, 128 B: geometric normal + the three vertices' normals | tangents | texcoords (bvh_build.hip emitTrianglesKernel)
  const DevInstance* instances;
  const float*       attributes;     // 12 floats per vertex
  const unsigned int* indices;
  const DevMaterial* materials;
  const DevLight*    lights;
  const float*       camera;         // P, U, V, W
  DevTexture         textures[3];
  const float*       envCDF_U;
  const float*       envCDF_V;
  int   tlasRoot;
  int   traceWaves;     // blocks per CU of the persistent trace kernel: TWK_TRACE_WAVES, or TWK_TRACE_WAVES7 (flattened scene, no cutout, small: see TWK_TRACE_WAVES7)
  int   twoLevel;       // 0: every instance is flattened — the BVH is one world-space tree (top level + spliced instance trees) and no kernel ever enters an instance
  int   numInstances;
  int   numLights;
  int   numMaterials;   // (the launch parameters are read with scalar loads: their layout shows in the kernels' code)
  int   miss;
  int   hasCutout;      // some material has cutout opacity: trace kernels run the stochastic any-hit candidate loop
  int   hasAlbedoTexture; // some material multiplies its albedo with the albedo texture: selects the shade kernel variant
  unsigned int envWidth, envHeight;
  float envIntegral, envRotation;

  // state
  int   resolution[2];
  int   tileSize[2];
  int   tileShift[2];
  int   pathLengths[2];
  int   deviceCount, deviceIndex, distribution;
  int   launchWidth;
  int   lensShader;
  unsigned int iterationIndex;
  float sceneEpsilon;

  // streams
  float4* rayOrg[2];     // origin.xyz, tmin   — two queues, ping-pong per bounce
  float4* rayDir[2];     // direction.xyz, tmax
  unsigned int* rayPixel[2];
  float4* hitRecord;     // t, beta, gamma, triangle slot (bits)
  int*    hitInstance;
  float4* shadowOrg;
  float4* shadowDir;
  unsigned int* shadowPixel;
  float4* shadowPending; // throughput * NEE contribution
  float4* rayThroughput[2]; // per queue slot, next to the ray: xyz throughput, w pdf of the last BSDF sample
  uint2*  raySeedFlags[2];  // per queue slot: LCG state, path word (FLAG_DIFFUSE, volume-stack index)
  float4* pathRadiance;
  float4* volumeStack;    // [4][numPaths]
  float4* pathAlbedo;     // denoiser AOVs (Optix7Gui raygeneration.cu:125-164), per path, nullptr when off: albedo of the first diffuse / light event
  float4* pathNormal;     //   camera-space shading normal of the primary hit
  float*  pathTime;       // time view (≙ USE_TIME_VIEW, raygeneration.cu:169-171,231-244), per path, nullptr when off: shader-clock cycles the path's lanes spent in traversal and shading
  float   clockScale;     // clockFactor * 1e-9 (Device.h:350 CLOCK_FACTOR_SCALE): cycles -> the alpha the colour ramp reads
  float4* aovAlbedo;      // their running means per launch index (raygeneration.cu:239-262)
  float4* aovNormal;
  int     shaderVariant;  // TWK_SHADERS_RTIGO3 / TWK_SHADERS_OPTIX7GUI (include/tweeker_hip.h)
  int     shadeSort;      // shadeKernel shades the slots of a block's window in class order (shade_kernels.hip "class-coherent execution"): 1 = in every launch but the first of a pass (default), 2 = in the first too, 0 = slot order (TWK_SHADE_SORT)
  int     nextEventEstimation; // ≙ USE_NEXT_EVENT_ESTIMATION (shaders/config.h:50-52), a run-time switch here (twk_set_next_event_estimation): 0 = brute-force path tracing, no light sampling, no MIS weights
  int     debugExceptions;     // ≙ USE_DEBUG_EXCEPTIONS (config.h:54-56; raygeneration.cu:205-218): NaN / Inf / negative samples become super red / green / blue instead of NaN being dropped
  int     outputFrame;    // 1: `output` is a shared full W x H frame addressed by absolute pixel (ZeroCopy / PeerAccess strategies), 0: this device's packed launchWidth x H buffer
  float4* output;         // running mean, RGBA32F
  unsigned int* counters; // see CounterSlot
  unsigned long long* stats; // TwkLaunchStats as 7 u64, or nullptr
  float4* firstHit;       // debug capture (t, beta, gamma, prim) or nullptr
  int*    firstHitInstance;
  int*    traceStackSpill; // per persistent lane overflow stack
  unsigned int* overflowSlots; // queue slots of rays that overflowed the LDS stack (2 per launch index)
  int     numPixels;      // launchWidth * height = launch indices of ONE sample per pixel
  int     numPaths;       // numPixels * batchCount: paths of this wavefront pass, path = sample * numPixels + launch index
  int     batchCount;     // iterations rendered together (iterationIndex .. iterationIndex + batchCount - 1)
  int     pathBase;       // first path of the pass this launch's streams start at (a pass cut into lanes, device_api.hip renderPass); 0 otherwise
  // Entry points of the primary rays (trace_kernels.hip tileEntryKernel), nullptr when off: per tile of TWK_ENTRY_TILE x
  // TWK_ENTRY_TILE launch indices two int4 = (count, ref 0..6): the subtrees a ray through that tile can reach, nearest first.
  const int4* tileEntries;
  int     tilesX;
  // 1: the queues shadeKernel writes (every depth >= 1) carry launch index + path flags in the .w of the origin and the LCG state in
  // the .w of the direction — the tmin / tmax of a continuation ray are the constants sceneEpsilon / RT_DEFAULT_MAX — and the
  // rayPixel / raySeedFlags streams are neither written nor read: 12 bytes less per path and bounce on both sides (TWK_PACKED_*).
  // Scenes without cutout opacity (its candidate loop keeps a per-ray tmin in the record), passes of fewer than 2^27 paths.
  int     packedQueue;
  unsigned int queueStride; // slots between the starts of two queue segments (TWK_QUEUE_STRIDE of the pass's or the lane's path count)
  unsigned int* droppedPushes; // pinned host word (device-mapped): pushes the single-ray traversal could not store (trace_device.h TWK_PUSH); stays 0 on every scene twk_build accepts
};

// Queue segments (round 5). shadeKernel's time was the number of returning atomics on ONE counter word divided by the rate a word
// sustains (87.8 per microsecond: profiles/r05_shade_diagnosis.md 7). So a queue is TWK_QUEUE_SEGMENTS regions of the same arrays,
// each with a counter word of its own: the block that shades window w (256 slots) appends to segment w mod K, at
// k * queueStride + the old value of that segment's counter. A consumer walks the segments one after the other: its VIRTUAL slot
// v in [0, total) is record physicalSlot(v) — the hit records it writes / reads stay indexed by v. Queue 0 (primary rays, or
// generateKernel's) and twk_debug_trace_queue's input are one segment: everything in segment 0.
#ifndef TWK_QUEUE_SEGMENTS
#define TWK_QUEUE_SEGMENTS 4
#endif
// Counter block layout (unsigned int each), zeroed once per launch. Per depth d (0..maxDepth), K = TWK_QUEUE_SEGMENTS:
// [0, K) rays in segment k of queue d, [K, 2K) shadow rays emitted by shade d in segment k, [2K] chunk tickets of trace launch d
// (second half of a long queue), [2K + 1] rays of trace launch d that overflowed the LDS stack.
// The counters of a queue's segments lie TWK_COUNTER_SEGMENT_STRIDE words apart (their own cache lines / memory channels).
#ifndef TWK_COUNTER_SEGMENT_STRIDE
#define TWK_COUNTER_SEGMENT_STRIDE 64 // 256 bytes: measured 1 (one line for all) / 16 / 1024 words: shade 0.233 / 0.2235 / 0.2207 ms per step (exact), 0.230 / 0.173 / 0.172 (native-math build)
#endif
#define TWK_COUNTERS_PER_DEPTH (2 * TWK_QUEUE_SEGMENTS * TWK_COUNTER_SEGMENT_STRIDE + 2)
#define TWK_COUNTER_CLOSEST  0
#define TWK_COUNTER_SHADOW   (TWK_QUEUE_SEGMENTS * TWK_COUNTER_SEGMENT_STRIDE)
#define TWK_COUNTER_TICKET   (2 * TWK_QUEUE_SEGMENTS * TWK_COUNTER_SEGMENT_STRIDE)
#define TWK_COUNTER_OVERFLOW (2 * TWK_QUEUE_SEGMENTS * TWK_COUNTER_SEGMENT_STRIDE + 1)
// Slots between the starts of two segments for a pass (or lane) of numPaths paths: segment k receives from at most
// ceil(windows / K) windows of at most 256 rays each.
#define TWK_QUEUE_STRIDE(numPaths) ((((unsigned int) (numPaths) / TWK_QUEUE_SEGMENTS + 255u) & ~255u) + 512u)

struct QueueSegments { unsigned int first[TWK_QUEUE_SEGMENTS]; unsigned int total; }; // virtual slot of each segment's first ray; rays in all

TWK_HD QueueSegments queueSegments(const unsigned int* counts)
{
  QueueSegments s; unsigned int sum = 0u;
  for (int k = 0; k < TWK_QUEUE_SEGMENTS; ++k) { s.first[k] = sum; sum += counts[k * TWK_COUNTER_SEGMENT_STRIDE]; }
  s.total = sum;
  return s;
}
TWK_HD QueueSegments noSegments() { QueueSegments s; for (int k = 0; k < TWK_QUEUE_SEGMENTS; ++k) s.first[k] = 0u; s.total = 0u; return s; }
TWK_HD unsigned int physicalSlot(const QueueSegments& s, unsigned int stride, unsigned int v)
{
  unsigned int k = 0u, f = 0u;
  for (int j = 1; j < TWK_QUEUE_SEGMENTS; ++j) { const bool in = v >= s.first[j]; k += in ? 1u : 0u; f = in ? s.first[j] : f; }
  return k * stride + (v - f);
}
#define TWK_MAX_DEPTH 64

#ifndef TWK_TRACE_STACK_LDS
#define TWK_TRACE_STACK_LDS   20  // entries per lane in LDS
#endif
// Blocks of the persistent trace kernel per CU (= waves per SIMD, the grid is numCUs x this). LDS per block: 21 KiB of
// traversal stacks + 5 KiB top-of-tree cache = 26 KiB, six of them in a CU's 160 KiB. Measured on C2 with every block
// resident: 4 / 5 / 6 blocks per CU = 0.94 / 0.88 / 0.84 ms per step.
#ifndef TWK_TRACE_WAVES
#define TWK_TRACE_WAVES 6
#endif
// The variant of the persistent trace kernel for scenes whose instances are all flattened, without cutout opacity, needs 70
// VGPRs: SEVEN blocks per CU fit the registers, and the LDS too with a 19-entry stack (the cliff on the shipped scenes is
// between 18 and 19: 0.673 against 0.536 ms/step at 18) and a 32-node top-of-tree cache (64 -> 32 costs nothing measurable):
// 7 x (20 KiB + 2.5 KiB) = 157.5 of 160 KiB. Measured, 6 -> 7 blocks per CU: C2 trace 0.536 -> 0.518 ms/step, whole frame
// +1.7 % (64 iterations) / +2.7 % (20), C4 geometry +2.0 %, a C5 rank's share +3.7 %. The two-level and cutout variants
// need 76..91 VGPRs and spill at 72 (C4 instances -11 %, C3 -16 %), and scenes of millions of triangles lose 2-6 % (the
// smaller cache and stack matter there), so those keep six blocks, a 20-entry stack and 64 cached nodes.
#ifndef TWK_TRACE_CUTOUT_SEVEN
#define TWK_TRACE_CUTOUT_SEVEN 1 // a switch, not a count: 1 = the flattened build with cutout opacity runs the seven-block form like the one without (71-80 VGPRs since round 4's restart from the ray record; 19-entry stack, 32 cached nodes; scenes of at most TWK_TRACE_WAVES7_MAX_NODES nodes), 0 = TWK_TRACE_WAVES blocks per CU
#endif
#ifndef TWK_TRACE_WAVES_CUTOUT_OTHER
#define TWK_TRACE_WAVES_CUTOUT_OTHER 5 // ... of its PRIMARY and two-level builds (93-94 VGPRs; at six 36-100 bytes of scratch)
#endif
#ifndef TWK_TRACE_WAVES_PRIMARY
#define TWK_TRACE_WAVES_PRIMARY TWK_TRACE_WAVES // blocks per CU of the PRIMARY builds (flattened: 80 VGPRs, 2 spilled at six)
#endif
#ifndef TWK_TRACE_WAVES_PRIMARY_TWO_LEVEL
#define TWK_TRACE_WAVES_PRIMARY_TWO_LEVEL 5 // ... of the two-level PRIMARY build (16 spilled at six: C4 instances +1 % at five; the flattened one: no difference)
#endif
#define TWK_TRACE_WAVES7      7
#define TWK_TRACE_STACK_LDS7  19
#define TWK_TOP_NODES7        32
#ifndef TWK_TRACE_WAVES7_MAX_NODES
#define TWK_TRACE_WAVES7_MAX_NODES 1000000 // binary nodes (= triangle slots - 1); measured on the Cornell room: +2.7 % at 64 k, +2.3 % at 258 k, +2.9 % at 977 k, -2.5 % at 2.0 M
#endif
#ifndef TWK_ENTRY_TILE
#define TWK_ENTRY_TILE 8        // launch indices per side of a primary-ray entry tile
#endif
#define TWK_ENTRY_REFS 7        // references per tile at most (with the count: two int4)
#ifndef TWK_PRIMARY_SIX
#define TWK_PRIMARY_SIX 1 // the PRIMARY build of the trace kernel (shade_kernels.hip "primary rays") runs six blocks per CU (2 registers spilled), not seven (16)
#endif
#define TWK_TRACE_STACK_SPILL 72  // further entries per lane in HBM
#define TWK_TRACE_BLOCK       256
#define TWK_SHADE_BLOCKS_PER_CU 128 // grid of shadeKernel = numCUs x this at most (device_api.hip)
#ifndef TWK_SHADE_BLOCK
#define TWK_SHADE_BLOCK       256  // threads per shadeKernel block: queue appends are aggregated per block. Measured (ms/step of shade): 128 → 0.48 (atomics), 256 → 0.33, 512 → 0.34 (waves wait at the block's barriers for its slowest wave), 1024 → 0.37
#endif
// Passes of at most this many paths are cut into two lanes (device_api.hip chooseLanes); measured on C2, DESIGN.md 2.
#ifndef TWK_LANES2_MAX_PATHS
#define TWK_LANES2_MAX_PATHS 21000000 // passes of at most this many paths run as two lanes. Round 5's final kernels, C2, one against two lanes: 2.1 M paths 1 138 / 1 211 Msamples/s, 4.1 M 1 646 / 1 735, 10.4 M 2 304 / 2 375, 20.7 M 2 760 / 2 766, 29 M 2 962 / 2 863 (a C5 rank's share of 20 iterations, 20.7 M: 2 746 / 2 718). Until then 30 M
#endif
#ifndef TWK_TRACE_SMALL_CHUNK
#define TWK_TRACE_SMALL_CHUNK 64   // queue slots per chunk of a SHORT queue (fewer than TWK_TRACE_TAIL_MIN long chunks per wave), 0 = round 2's one contiguous share per wave
#endif
#ifndef TWK_TRACE_CHUNK
#define TWK_TRACE_CHUNK       512  // queue slots per chunk of the persistent trace kernel's interleaved assignment (trace_kernels.hip)
#endif

} // namespace twk
