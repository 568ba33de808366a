// Ray generation, shading and accumulation kernels of the wavefront renderer.
//
// Reference programs restated per stage (apps/rtigo3/shaders/):
//   generateKernel   __raygen__path_tracer up to the first optixTrace      raygeneration.cu:167-203, lens_shader.cu
//   shadeKernel      miss programs miss.cu:41-109, __closesthit__radiance closesthit.cu:126-305, the BSDF /
//                    light callables (bxdf_*.cu, light_sample.cu) and the integrator loop body after the
//                    trace raygeneration.cu:91-146
//   accumulateKernel NaN filter + running mean                              raygeneration.cu:222-253
//   compositorKernel compositor.cu:38-64 for all source devices at once
// Per-path RNG draw order is the reference's: jitter rng2; per bounce the BSDF's draws, then NEE rng2
// [+ rng when more than one light], then Russian roulette rng.
#include "shade_device.h"
#include "../../include/tweeker_hip.h"

namespace twk {
// One thread per path = (sample, launch index): seed, jitter, lens shader, path state reset, primary ray into
// queue 0. A pass renders batchCount consecutive iterations at once (path = sample * numPixels + launch index, so a
// wave still covers 64 neighbouring pixels of one sample); sample s uses iterationIndex + s exactly as a launch of
// its own would.
// The seed index is the absolute pixel W*y + x for every device count (for one device this IS the
// reference's formula raygeneration.cu:191; for several it makes tiled == single-device bit for bit).
__global__ void __launch_bounds__(256) generateKernel(LaunchParams p)
{
  const unsigned int index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= (unsigned int) p.numPaths) return;
  generatePath(p, index);
}

// ---------------------------------------------------------------------------------------------
// One thread per ray of queue (depth & 1): shadePath(), then append the continuation ray — with the path's throughput,
// pdf, RNG state and flags, which travel in the queue next to the ray so that every access of a bounce is a coalesced
// stream (indexed by path they were 16-byte gathers after the first compaction) — to queue ((depth + 1) & 1) and the
// shadow ray (with the pending contribution) to the shadow queue.
//
// Queue appends are aggregated per BLOCK (TWK_SHADE_BLOCK threads): every wave counts its appenders with a ballot, the
// block sums the wave counts through LDS and one lane issues ONE returning atomic per queue per block iteration. With one atomic per
// wave the two counter words saw ~110 k returning atomics per step and the kernel sat in s_waitcnt for 87 % of its
// wave-cycles (a single word sustains ~90 atomics/us on this chip: MI355X_MICROARCH "dequeue").
// Waves per SIMD the register allocation aims at (launch bounds), per kernel variant. The variant without the spherical
// environment needs 107 VGPRs unconstrained and fits 96 — five waves per SIMD — WITHOUT scratch (the texture variant
// with 16 bytes of it); the environment variants spill 28 / 72 bytes at 96 and stay at four waves (113 VGPRs, 36 bytes).
// Measured, whole frame, 4 -> 5 waves: C2 2 439 -> 2 525, C4 geometry 4 173 -> 4 309, C4 instances 3 712 -> 3 757, a C5
// rank's share 2 253 -> 2 317 Msamples/s; C3 (environment + textures) 3 869 -> 3 764, hence the split. Six waves (80
// VGPRs) spill 68..136 bytes in every variant. (Round 2 forced 96 registers on the one-for-all kernel: 0.290 -> 0.350 ms
// per step; what changed is that the variant no longer carries the environment sampler's and the texture fetch's live state.)
#ifndef TWK_SHADE_WAVES
#define TWK_SHADE_WAVES 5
#endif
#ifndef TWK_SHADE_WAVES_ENV
#define TWK_SHADE_WAVES_ENV 4
#endif

// The streams of one queue slot, as loaded (the fetch is issued one block iteration ahead, see shadeKernel).
struct ShadeInput
{
  float4 ro, rd, hit, throughputPdf;
  uint2 seedFlags;
  unsigned int pixel;
  int instanceIndex;
  bool inRange;
};

// Packed queue records (device_types.h LaunchParams::packedQueue): the path word <-> the five bits next to the launch index.
TWK_D unsigned int packPathWord(unsigned int pixel, unsigned int word)
{
  return pixel | (((word >> 2) & 1u) << 27) | (((word >> 28) & 1u) << 28) | (((word >> TWK_PATH_STACK_SHIFT) & 7u) << 29);
}
TWK_D unsigned int unpackPathWord(unsigned int packed)
{
  return (((packed >> 27) & 1u) << 2) | (((packed >> 28) & 1u) << 28) | ((packed >> 29) << TWK_PATH_STACK_SHIFT);
}
static_assert(TWK_FLAG_DIFFUSE == (1u << 2) && TWK_FLAG_ALBEDO == (1u << 28) && TWK_PACKED_PIXEL_BITS == 27, "packPathWord's bit positions");

// PRIMARY ("primary rays" below): queue 0 was never written; the slot's ray and path state are computed.
// packed: the queue was written by shadeKernel in the packed form (device_types.h LaunchParams::packedQueue).
// slot: the VIRTUAL slot (hit records are indexed by it); the queued ray and its path state sit at physicalSlot(slot)
// (device_types.h "queue segments").
template<bool PRIMARY>
TWK_D void loadShadeInput(const LaunchParams& p, int q, unsigned int slot, const QueueSegments& segments, bool packed, ShadeInput& in)
{
  const unsigned int numRays = segments.total;
  in.inRange = slot < numRays;
  if (PRIMARY)
  {
    if (in.inRange)
    {
      const PrimaryRay ray = primaryRay(p, slot);
      in.ro = make_float4(ray.origin.x, ray.origin.y, ray.origin.z, p.sceneEpsilon);
      in.rd = make_float4(ray.direction.x, ray.direction.y, ray.direction.z, ray.active ? RT_DEFAULT_MAX : -1.0f);
      in.pixel = slot;
      in.hit = p.hitRecord[slot];
      in.instanceIndex = p.hitInstance[slot];
      in.throughputPdf = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
      // a scene with cutout opacity: the first traversal stored the seed in queue 0 and its opacity tests drew from it
      in.seedFlags = p.hasCutout ? p.raySeedFlags[0][slot] : make_uint2(ray.seed, 0u);
    }
    return;
  }
  if (in.inRange)
  {
    const unsigned int record = physicalSlot(segments, p.queueStride, slot);
    in.ro = p.rayOrg[q][record];
    in.rd = p.rayDir[q][record];
    in.hit = p.hitRecord[slot];
    in.instanceIndex = p.hitInstance[slot];
    in.throughputPdf = p.rayThroughput[q][record];
    if (packed)
    {
      const unsigned int word = __float_as_uint(in.ro.w);
      in.pixel = word & TWK_PACKED_PIXEL_MASK;
      in.seedFlags = make_uint2(__float_as_uint(in.rd.w), unpackPathWord(word));
      in.ro.w = p.sceneEpsilon; in.rd.w = RT_DEFAULT_MAX; // what a continuation ray's record holds there otherwise
    }
    else
    {
      in.pixel = p.rayPixel[q][record];
      in.seedFlags = p.raySeedFlags[q][record];
    }
  }
}

// Block barrier that orders LDS only. __syncthreads() also drains the wave's outstanding global loads and stores
// (s_waitcnt vmcnt(0)), which is exactly what the two barriers of the append must not do: see shadeKernel.
TWK_D void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// Class-coherent execution (round 5). The shading of a segment branches on what was hit — nothing, a light, one of five BSDFs —
// and a wave pays for every branch any of its lanes takes: on C2 the GGX wall's tangent frame, sample and eval ran in 53 % of the
// wave iterations with 10 % of the lanes (profiles/r05a_shade_phases_c2_b20.txt). So the block sorts its window of TWK_SHADE_BLOCK
// queue slots by class before it shades them — a counting sort through LDS: one returning LDS add per thread on its class
// counter, its place = the threads of smaller classes + its rank — and the threads EXCHANGE what they loaded: every thread reads
// its slot's streams in slot order exactly as without the sort (coalesced, requested a block iteration ahead), stores them at its
// place of the exchange buffer, and takes over the record of the slot it shades. Paths do not depend on which thread shades them:
// images stay bit-identical (tests/test_gpu_pass_variants.py; TWK_SHADE_SORT=0 restores slot order).
// History: a sort that LOADED in sorted order (hit instance first, then the streams of slot window + perm[t]) was built in rounds
// 2, 3 and 5: -38 % vector instructions, lanes per instruction 0.37 -> 0.65, and never faster — until round 5 the kernel stood on
// the returning atomic of one counter word (profiles/r05_shade_diagnosis.md 7), and after that the dependent second fetch cost a
// memory round trip per window, which most blocks cannot hide (the grid gives a block one or two windows).
#define TWK_SHADE_CLASSES 8
#ifndef TWK_SHADE_SORT_TABLE_BYTES
#define TWK_SHADE_SORT_TABLE_BYTES 8192 // table budget of the builds that carry the exchange buffer (20 KiB): five blocks per CU = 145 of 160 KiB
#endif
static_assert((TWK_SHADE_BLOCK & (TWK_SHADE_BLOCK - 1)) == 0, "the sorted window wraps with a mask");

// Order of the classes in a sorted window: Lambert, the class with most lanes, in front; GGX, the rare expensive one (its wave
// takes 2.4 x a Lambert wave's cycles on C2 and is what a block iteration waits for), at the end next to the classes that end the
// path or only reflect. (Measured against the order miss, light, mirror, glass, rough glass, Lambert, GGX: no difference.)
TWK_D unsigned int shadeClass(const ShadeTables& tables, int instanceIndex, bool active)
{
  if (!active) return 7u;                           // beyond the queue, or an inactive launch index
  if (instanceIndex < 0) return 3u;                 // miss program
  const DevInstance& inst = tables.instances[instanceIndex];
  if (inst.light >= 0) return 4u;                   // light geometry
  const int bsdf = tables.materials[inst.material].indexBSDF;
  return (bsdf == 1) ? 2u : (bsdf == 2) ? 1u : (bsdf == 4) ? 5u : (bsdf == 3) ? 6u : 0u; // mirror, glass, rough glass, GGX; Lambert (default)
}
// What a thread hands over: the streams of one queue slot as loaded (ShadeInput), 80 bytes, as four 16-byte rows + four words.
struct ShadeExchange
{
  float4 ro[TWK_SHADE_BLOCK], rd[TWK_SHADE_BLOCK], hit[TWK_SHADE_BLOCK], throughputPdf[TWK_SHADE_BLOCK];
  uint4  words[TWK_SHADE_BLOCK]; // seed, flags, pixel (all ones: no slot), instance
};
// rotate: the sorted window starts at thread `rotate` and wraps — a multiple of 64 that differs from block to block and from
// window to window. Without it wave 3 of EVERY block would shade the last classes (Lambert, GGX: the expensive ones) and wave 0 the
// misses; a block's waves sit on the four SIMDs of its CU in order, so one SIMD of every CU would do most of the chip's shading.
TWK_D void sortExchange(const ShadeTables& tables, unsigned int* classCount, ShadeExchange& x, unsigned int rotate, ShadeInput& in)
{
  const bool active = in.inRange && in.rd.w >= 0.0f;
  const unsigned int key = shadeClass(tables, in.instanceIndex, active);
  const unsigned int rank = atomicAdd(classCount + key, 1u); // the order inside a class is the order the adds arrive in: no result depends on it
  ldsBarrier();
  const uint4 lo = *reinterpret_cast<const uint4*>(classCount), hi = *reinterpret_cast<const uint4*>(classCount + 4);
  const unsigned int counts[TWK_SHADE_CLASSES] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  unsigned int place = rank + rotate;
#pragma unroll
  for (unsigned int c = 0; c + 1 < TWK_SHADE_CLASSES; ++c) place += (c < key) ? counts[c] : 0u;
  place &= TWK_SHADE_BLOCK - 1u;
  x.ro[place] = in.ro; x.rd[place] = in.rd; x.hit[place] = in.hit; x.throughputPdf[place] = in.throughputPdf;
  x.words[place] = make_uint4(in.seedFlags.x, in.seedFlags.y, in.inRange ? in.pixel : 0xFFFFFFFFu, (unsigned int) in.instanceIndex);
  ldsBarrier();
  if (threadIdx.x < TWK_SHADE_CLASSES) classCount[threadIdx.x] = 0u; // every thread has read the counts; the next window counts behind two more barriers
  in.ro = x.ro[threadIdx.x]; in.rd = x.rd[threadIdx.x]; in.hit = x.hit[threadIdx.x]; in.throughputPdf = x.throughputPdf[threadIdx.x];
  const uint4 w = x.words[threadIdx.x];
  in.seedFlags = make_uint2(w.x, w.y); in.pixel = w.z; in.inRange = (w.z != 0xFFFFFFFFu); in.instanceIndex = (int) w.w;
}

// Primary rays. A pass used to start with generateKernel writing queue 0 — ray, pixel, throughput, seed and the black radiance
// of every path, 76 bytes each, 157 MB per C2 iteration at the HBM write rate — for the first traversal and the first shade
// launch to read back. Both now COMPUTE the primary ray of their slot (shade_device.h primaryRay: 60 integer operations for
// the seed, two draws, the lens shader) — PRIMARY variants of traceKernel and shadeKernel, launched at depth 0 — and the first
// shade launch writes the path's radiance instead of adding to it. A scene with cutout opacity keeps ONE word of queue 0: the
// seed, stored by the first traversal (its opacity tests draw from it) and read by the first shade launch. device_api.hip
// renderPass keeps generateKernel for paths without any bounce.
#ifndef TWK_SHADE_LDS_TABLES
#define TWK_SHADE_LDS_TABLES 1
#endif
#ifndef TWK_SHADE_TABLE_BYTES
#define TWK_SHADE_TABLE_BYTES 20480 // per block; five blocks per CU
#endif
#ifndef TWK_SHADE_WAVES_PRIMARY
#define TWK_SHADE_WAVES_PRIMARY 4 // the PRIMARY variant carries the ray generation: 13 registers spilled at five waves
#endif
// MEASURE: the measurement build — the time view (twk_set_time_view: every path's shading cycles are added to its time word) and
// twk_stats_enable (per-phase wave executions, lanes and cycles of shadePath: shade_device.h PhaseScope).
// TWK_SHADE_EXTRA_FMA=N (experiment builds): N more dependent v_fma_f32 per shaded path — the issue-sensitivity probe of
// profiles/r05_shade_fma_sensitivity.md.
#ifndef TWK_SHADE_EXTRA_FMA
#define TWK_SHADE_EXTRA_FMA 0
#endif
#ifndef TWK_PROBE_EXTRA_ATOMICS
#define TWK_PROBE_EXTRA_ATOMICS 0
#endif
// SORT: the block shades its window in class order (above; LDS_TABLES builds only). The measurement builds carry the exchange
// buffer too and take LaunchParams::shadeSort at run time.
template<bool ENV, bool TEX, bool PRIMARY, bool LDS_TABLES, bool MEASURE, bool SORT>
__global__ void __launch_bounds__(TWK_SHADE_BLOCK, ENV ? TWK_SHADE_WAVES_ENV : (PRIMARY ? TWK_SHADE_WAVES_PRIMARY : TWK_SHADE_WAVES)) shadeKernel(LaunchParams p, int depth)
{
  // Double-buffered by block iteration: iteration i + 2 rewrites what i used only after every thread has passed a barrier
  // of iteration i + 1, so no third barrier per iteration is needed.
  // The grid is sized for the pass's path count (the host does not know a queue's length): at the deep bounces most blocks have no
  // window. They leave HERE, before the table copy and its barrier — each used to hold a block slot for a global round trip: with 28 k
  // of 32 k blocks idle (1 M rays) that was half of the launch (round 5). The block's first window is requested at once, so that
  // its streams fly while the tables are copied.
  const QueueSegments segments = queueSegments(&p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_CLOSEST]);
  const unsigned int numRays = segments.total;
  if (blockIdx.x * blockDim.x >= numRays) return;
  const int q = depth & 1, qn = q ^ 1;
  const bool packedIn = p.packedQueue != 0 && depth > 0, packedOut = p.packedQueue != 0; // queue 0 is computed (PRIMARY) or written by generateKernel
  ShadeInput in = {}; // a thread beyond the queue hands its (empty) record to the exchange all the same
  loadShadeInput<PRIMARY>(p, q, blockIdx.x * blockDim.x + threadIdx.x, segments, packedIn, in);

  __shared__ unsigned int waveCount[2][2][TWK_SHADE_BLOCK / 64];
  __shared__ unsigned int blockBase[2][2];
  __shared__ unsigned int phaseWords[MEASURE ? 3 * TWK_SHADE_PHASES : 1];
  unsigned int* const phaseLds = (MEASURE && p.stats != nullptr) ? phaseWords : nullptr;
  if (MEASURE) { if (threadIdx.x < 3 * TWK_SHADE_PHASES) phaseWords[threadIdx.x] = 0u; __syncthreads(); }
  const bool measurePhases = MEASURE && p.stats != nullptr; // time view alone: only the path time
  constexpr bool EXCHANGE = LDS_TABLES && (SORT || MEASURE);
  const bool sorted = LDS_TABLES && (MEASURE ? (p.shadeSort == 2 || (p.shadeSort == 1 && !PRIMARY)) : SORT); // as launchShade chooses for the plain builds
  __shared__ __attribute__((aligned(16))) unsigned int classCount[TWK_SHADE_CLASSES]; // class sort: threads of the window per class (zero between uses)
  __shared__ float4 exchangeStorage[EXCHANGE ? sizeof(ShadeExchange) / 16 : 1];
  if (EXCHANGE && threadIdx.x < TWK_SHADE_CLASSES) classCount[threadIdx.x] = 0u; // the table copy's barrier is behind this

  // Instance, material and light records in LDS (scenes whose tables fit): a hit reads ~16 float4 of them, each one divergent
  // lane address for the CU's vector memory path, which takes one per clock — the kernel's bound (rocprofv3: 0.9 lane
  // addresses per clock and CU) — while LDS serves a wave's read of a few distinct records in 8..32 clocks
  // (profiles/r02a_gather_probe2.txt).
  ShadeTables tables;
  tables.instances = p.instances; tables.materials = p.materials; tables.lights = p.lights;
  __shared__ float4 tableStorage[LDS_TABLES ? (EXCHANGE ? TWK_SHADE_SORT_TABLE_BYTES : TWK_SHADE_TABLE_BYTES) / 16 : 1];
  if (LDS_TABLES)
  {
    // launchShade has checked that the three tables fit. The pointers are LDS pointers at compile time: the reads of the
    // records become ds_read (through a pointer chosen at run time they stay flat loads, which the vector memory path
    // processes at the same one lane address per clock whether they end in LDS or not: measured, no gain at all).
    const unsigned int nI = (unsigned int) p.numInstances * (sizeof(DevInstance) / 16), nM = (unsigned int) p.numMaterials * (sizeof(DevMaterial) / 16), nL = (unsigned int) p.numLights * (sizeof(DevLight) / 16);
    const float4* gI = reinterpret_cast<const float4*>(p.instances); const float4* gM = reinterpret_cast<const float4*>(p.materials); const float4* gL = reinterpret_cast<const float4*>(p.lights);
    for (unsigned int i = threadIdx.x; i < nI; i += TWK_SHADE_BLOCK) tableStorage[i] = gI[i];
    for (unsigned int i = threadIdx.x; i < nM; i += TWK_SHADE_BLOCK) tableStorage[nI + i] = gM[i];
    for (unsigned int i = threadIdx.x; i < nL; i += TWK_SHADE_BLOCK) tableStorage[nI + nM + i] = gL[i];
    tables.instances = reinterpret_cast<const DevInstance*>(tableStorage);
    tables.materials = reinterpret_cast<const DevMaterial*>(tableStorage + nI);
    tables.lights    = reinterpret_cast<const DevLight*>(tableStorage + nI + nM);
    __syncthreads();
  }
  unsigned int statHit = 0, statMiss = 0;

  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long laneBelow = (1ull << lane) - 1ull;

  // A block iteration is a chain of waits — the slot's streams, the shading record, the slowest wave at the
  // append's first barrier, the returning atomic — so the chain is kept short: the streams of the NEXT iteration's slot are
  // requested between the two barriers of the append — they fly while the block waits for its returning atomic — and nothing
  // waits for the appended records to be written.
  unsigned int buffer = 0u;

  // block-uniform trip count: every thread reaches both barriers of every iteration
  for (unsigned int base = blockIdx.x * blockDim.x; base < numRays; base += gridDim.x * blockDim.x)
  {
    ShadeOutput out;
    out.alive = false; out.wantShadow = false;
    // this window's segment of the two queues it appends to (device_types.h "queue segments"): a counter word of its own
    const unsigned int segment = (base / TWK_SHADE_BLOCK) % TWK_QUEUE_SEGMENTS;
    unsigned int* const nextCount   = &p.counters[(depth + 1) * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_CLOSEST + segment * TWK_COUNTER_SEGMENT_STRIDE];
    unsigned int* const shadowCount = &p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_SHADOW + segment * TWK_COUNTER_SEGMENT_STRIDE];
    const unsigned int clockBegin = MEASURE ? (unsigned int) __builtin_readcyclecounter() : 0u;
    const unsigned int iterationBegin = clockBegin;
    if (EXCHANGE && sorted) sortExchange(tables, classCount, *reinterpret_cast<ShadeExchange*>(exchangeStorage), ((blockIdx.x + base / (gridDim.x * blockDim.x)) & 3u) << 6, in);
    const unsigned int slotLanes = MEASURE ? (unsigned int) __popcll(__ballot(in.inRange)) : 0u; // lanes of this wave with a queue slot in THIS window (`in` holds the next window's by the time the tallies are written)
    const unsigned int pixel = in.pixel;
    if (in.inRange && in.rd.w >= 0.0f) // else: beyond the queue, or an inactive launch index (tile column beyond the image)
    {
      if (measurePhases)
      {
        // the wait for the slot's streams, requested one block iteration ahead: their first use
        PhaseScope<MEASURE> phase(phaseLds, SP_KERNEL_LOAD);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      out.throughputPdf = in.throughputPdf;
      out.seedFlags     = in.seedFlags;
      if (measurePhases) shadePath<ENV, TEX, PRIMARY, MEASURE>(p, tables, depth, pixel, in.ro, in.rd, in.hit, in.instanceIndex, out, phaseLds);
      else               shadePath<ENV, TEX, PRIMARY, false>(p, tables, depth, pixel, in.ro, in.rd, in.hit, in.instanceIndex, out);
#if TWK_SHADE_EXTRA_FMA
      {
        float acc = out.throughputPdf.x;
        const float a = p.sceneEpsilon + 1.0f, b = p.clockScale;
#pragma unroll
        for (int k = 0; k < TWK_SHADE_EXTRA_FMA; ++k) acc = __builtin_fmaf(acc, a, b);
        if (acc == 12345.678f) out.nextPos.x = acc; // keeps the chain alive; never true for finite inputs of this size
      }
#endif
      if (p.stats != nullptr) { if (in.instanceIndex < 0) ++statMiss; else ++statHit; }
      if (MEASURE && p.pathTime != nullptr) atomicAdd(&p.pathTime[pixel], float((unsigned int) __builtin_readcyclecounter() - clockBegin)); // the shadow ray of this path may be adding its traversal time meanwhile: atomic
    }
    else if (PRIMARY && in.inRange) p.pathRadiance[pixel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // inactive launch index: weight 0, as generateKernel leaves it

    const unsigned int appendBegin = MEASURE ? (unsigned int) __builtin_readcyclecounter() : 0u;
    const unsigned long long shadowMask = __ballot(out.wantShadow);
    const unsigned long long nextMask   = __ballot(out.alive);
    if (lane == 0)
    {
      waveCount[buffer][0][wave] = (unsigned int) __popcll(shadowMask);
      waveCount[buffer][1][wave] = (unsigned int) __popcll(nextMask);
    }
    ldsBarrier();
    const unsigned int afterBarrier1 = MEASURE ? (unsigned int) __builtin_readcyclecounter() : 0u;
    // (hipcc still waits for part of these right here — it copies one component of the hit record to another register
    // behind the loads; pinning the record at its first use makes that worse, every component then gets such a copy)
    loadShadeInput<PRIMARY>(p, q, base + gridDim.x * blockDim.x + threadIdx.x, segments, packedIn, in);
#if TWK_PROBE_EXTRA_ATOMICS // timing probe (profiles/r05_shade_diagnosis.md 7): that many more returning atomics per counter word and block iteration, adding a zero the compiler cannot see
    if (threadIdx.x >= 2 && threadIdx.x < 2 + 2 * TWK_PROBE_EXTRA_ATOMICS)
      blockBase[buffer ^ 1u][threadIdx.x & 1u] += atomicAdd((threadIdx.x & 1u) ? nextCount : shadowCount, (unsigned int) p.numPaths >> 31) & 0u;
#endif
    if (threadIdx.x < 2)
    {
      unsigned int total = 0;
      for (unsigned int w = 0; w < TWK_SHADE_BLOCK / 64; ++w) total += waveCount[buffer][threadIdx.x][w];
      const unsigned int atomicBegin = MEASURE ? (unsigned int) __builtin_readcyclecounter() : 0u;
      blockBase[buffer][threadIdx.x] = (total != 0u) ? atomicAdd((threadIdx.x == 0) ? shadowCount : nextCount, total) : 0u;
      if (measurePhases && total != 0u)
      {
        // the round trip of the block's returning atomic, as the issuing lane sees it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(&phaseWords[SP_APPEND_ATOMIC], 1u); atomicAdd(&phaseWords[TWK_SHADE_PHASES + SP_APPEND_ATOMIC], 1u);
        atomicAdd(&phaseWords[2 * TWK_SHADE_PHASES + SP_APPEND_ATOMIC], (unsigned int) __builtin_readcyclecounter() - atomicBegin);
      }
    }
    ldsBarrier();
    if (measurePhases && lane == 0)
    {
      // per wave: how long it stood at the first barrier (for the block's slowest wave) and at the second (for the atomic)
      const unsigned int now = (unsigned int) __builtin_readcyclecounter();
      atomicAdd(&phaseWords[SP_APPEND_BARRIER1], 1u); atomicAdd(&phaseWords[TWK_SHADE_PHASES + SP_APPEND_BARRIER1], 64u);
      atomicAdd(&phaseWords[2 * TWK_SHADE_PHASES + SP_APPEND_BARRIER1], afterBarrier1 - appendBegin);
      atomicAdd(&phaseWords[SP_APPEND_BARRIER2], 1u); atomicAdd(&phaseWords[TWK_SHADE_PHASES + SP_APPEND_BARRIER2], 64u);
      atomicAdd(&phaseWords[2 * TWK_SHADE_PHASES + SP_APPEND_BARRIER2], now - afterBarrier1);
    }
    unsigned int shadowOffset = segment * p.queueStride + blockBase[buffer][0], nextOffset = segment * p.queueStride + blockBase[buffer][1];
    for (unsigned int w = 0; w < wave; ++w) { shadowOffset += waveCount[buffer][0][w]; nextOffset += waveCount[buffer][1][w]; }

    if (out.wantShadow)
    {
      const unsigned int s = shadowOffset + (unsigned int) __popcll(shadowMask & laneBelow);
      p.shadowOrg[s]     = make_float4(out.nextPos.x, out.nextPos.y, out.nextPos.z, p.sceneEpsilon);
      p.shadowDir[s]     = make_float4(out.shadowDir.x, out.shadowDir.y, out.shadowDir.z, out.shadowTmax);
      p.shadowPixel[s]   = pixel;
      p.shadowPending[s] = make_float4(out.pending.x, out.pending.y, out.pending.z, __uint_as_float(out.shadowSeed));
    }
    if (out.alive)
    {
      const unsigned int n = nextOffset + (unsigned int) __popcll(nextMask & laneBelow);
      if (packedOut)
      {
        // launch index + path flags and the LCG state ride in the record's two constant words: 12 bytes less written here, 12
        // less read by the next shade launch (the big shade launches sit at the HBM rate)
        p.rayOrg[qn][n] = make_float4(out.nextPos.x, out.nextPos.y, out.nextPos.z, __uint_as_float(packPathWord(pixel, out.seedFlags.y)));
        p.rayDir[qn][n] = make_float4(out.nextDir.x, out.nextDir.y, out.nextDir.z, __uint_as_float(out.seedFlags.x));
      }
      else
      {
        p.rayOrg[qn][n]   = make_float4(out.nextPos.x, out.nextPos.y, out.nextPos.z, p.sceneEpsilon);
        p.rayDir[qn][n]   = make_float4(out.nextDir.x, out.nextDir.y, out.nextDir.z, RT_DEFAULT_MAX);
        p.rayPixel[qn][n] = pixel;
        p.raySeedFlags[qn][n]  = out.seedFlags;
      }
      p.rayThroughput[qn][n] = out.throughputPdf;
    }
    buffer ^= 1u;
    if (measurePhases && lane == 0)
    {
      // per wave (every wave of the block runs every iteration: the barriers): the append, and the whole iteration
      const unsigned int now = (unsigned int) __builtin_readcyclecounter();
      atomicAdd(&phaseWords[SP_KERNEL_APPEND], 1u); atomicAdd(&phaseWords[TWK_SHADE_PHASES + SP_KERNEL_APPEND], (unsigned int) __popcll(shadowMask | nextMask));
      atomicAdd(&phaseWords[2 * TWK_SHADE_PHASES + SP_KERNEL_APPEND], now - appendBegin);
      atomicAdd(&phaseWords[SP_KERNEL_ITERATION], 1u); atomicAdd(&phaseWords[TWK_SHADE_PHASES + SP_KERNEL_ITERATION], slotLanes);
      atomicAdd(&phaseWords[2 * TWK_SHADE_PHASES + SP_KERNEL_ITERATION], now - iterationBegin);
    }
  }
  if (measurePhases)
  {
    __syncthreads();
    // stats[24 ..): wave executions, lanes, cycles per phase (device_api.hip twk_stats_get)
    if (threadIdx.x < 3 * TWK_SHADE_PHASES && phaseWords[threadIdx.x] != 0u) atomicAdd(&p.stats[24 + threadIdx.x], (unsigned long long) phaseWords[threadIdx.x]);
  }

  if (p.stats != nullptr)
  {
    // one atomic per wave, not per path
    for (int offset = 32; offset > 0; offset >>= 1)
    {
      statHit  += __shfl_down(statHit, offset);
      statMiss += __shfl_down(statMiss, offset);
    }
    if ((threadIdx.x & 63) == 0)
    {
      if (statHit)  atomicAdd(&p.stats[5], (unsigned long long) statHit);
      if (statMiss) atomicAdd(&p.stats[6], (unsigned long long) statMiss);
    }
  }
}

// raygeneration.cu:222-253: drop NaN samples, running mean into the RGBA32F buffer, alpha 1. The samples of a batch
// are folded in iteration order, one lerp each, so the float result equals batchCount separate launches.
__global__ void __launch_bounds__(256) accumulateKernel(LaunchParams p)
{
  const unsigned int index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= (unsigned int) p.numPixels) return;
  accumulateLaunchIndex(p, index);
}

// compositor.cu:38-64 for every source device in one launch: tiles is [deviceCount][H][launchWidth].
__global__ void __launch_bounds__(256) compositorKernel(const float4* __restrict__ tiles, float4* __restrict__ output,
                                                         int width, int height, int launchWidth, int deviceCount,
                                                         int tileSizeX, int tileShiftX, int tileShiftY)
{
  const unsigned int xLaunch = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned int yLaunch = blockIdx.y;
  const unsigned int device  = blockIdx.z;
  if (xLaunch >= (unsigned int) launchWidth || yLaunch >= (unsigned int) height) return;
  const unsigned int xBlock = xLaunch >> tileShiftX;
  const unsigned int yBlock = yLaunch >> tileShiftY;
  const unsigned int xTile  = xBlock * deviceCount + ((device + yBlock) % deviceCount);
  const unsigned int xPixel = xTile * tileSizeX + (xLaunch & (tileSizeX - 1));
  if (xPixel < (unsigned int) width)
  {
    output[(size_t) yLaunch * width + xPixel] = tiles[((size_t) device * height + yLaunch) * launchWidth + xLaunch];
  }
}

__global__ void mathTapKernel(int op, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, size_t n)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
  {
    float r = 0.0f;
    switch (op)
    {
      case 0: r = sinP(x[i]); break;
      case 1: r = cosP(x[i]); break;
      case 2: r = expP(x[i]); break;
      case 3: r = atan2P(x[i], y[i]); break;
      case 4: r = acosP(x[i]); break;
      case 5: r = atanP(x[i]); break;
      case 6: r = sqrtf(x[i]); break;
      case 7: r = 1.0f / x[i]; break;
      case 8: r = logP(x[i]); break;
      case 9: r = powP(x[i], y[i]); break;
    }
    out[i] = r;
  }
}

// Tonemapper of Application::screenshot (Application.cpp:2259-2297, the loop its authors mark "PERF Add a native CUDA
// kernel doing this"; same operator as the GLSL display shader Rasterizer.cpp:553-578): white point, colour balance,
// burn highlights, saturation, crush blacks, gamma, then truncation to 8 bits. One pixel's three bytes per thread
// iteration; pow is device_math.h powP (fixed algorithm, the oracle evaluates the same).
struct TonemapConstants { float invGamma, invWhitePoint, burnHighlights, crushBlacks, saturation; float colorBalance[3]; };

TWK_D V3 pow3(const V3& v, float e) { return v3(powP(v.x, e), powP(v.y, e), powP(v.z, e)); }
TWK_D V3 max3(const V3& v, float lo) { return v3(fmaxf(lo, v.x), fmaxf(lo, v.y), fmaxf(lo, v.z)); }
TWK_D float saturateP(float v) { return fmaxf(0.0f, fminf(v, 1.0f)); } // clamp(), vector_math.h:148-151

__global__ void tonemapKernel(const float4* __restrict__ hdr, unsigned char* __restrict__ ldr, size_t numPixels, TonemapConstants c)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < numPixels; i += (size_t) gridDim.x * blockDim.x)
  {
    const V3 hdrColor = v3(hdr[i]);
    V3 ldrColor = v3(c.invWhitePoint * c.colorBalance[0], c.invWhitePoint * c.colorBalance[1], c.invWhitePoint * c.colorBalance[2]) * hdrColor; // :2275
    ldrColor = ldrColor * v3((ldrColor.x * c.burnHighlights + 1.0f) / (ldrColor.x + 1.0f),
                             (ldrColor.y * c.burnHighlights + 1.0f) / (ldrColor.y + 1.0f),
                             (ldrColor.z * c.burnHighlights + 1.0f) / (ldrColor.z + 1.0f));                                             // :2276
    float luminance = dot(ldrColor, v3(0.3f, 0.59f, 0.11f));
    ldrColor = lerp(v3(luminance), ldrColor, c.saturation);
    ldrColor = max3(ldrColor, 0.0f);
    luminance = dot(ldrColor, v3(0.3f, 0.59f, 0.11f));
    if (luminance < 1.0f)
    {
      const V3 crushed = pow3(ldrColor, c.crushBlacks);
      ldrColor = lerp(crushed, ldrColor, sqrtf(luminance));
      ldrColor = max3(ldrColor, 0.0f);
    }
    ldrColor = pow3(ldrColor, c.invGamma);
    ldr[3 * i + 0] = (unsigned char) (saturateP(ldrColor.x) * 255.0f);
    ldr[3 * i + 1] = (unsigned char) (saturateP(ldrColor.y) * 255.0f);
    ldr[3 * i + 2] = (unsigned char) (saturateP(ldrColor.z) * 255.0f);
  }
}

// Stream-copy peak (measurement only, twk_stream_peak_gbps): ONE float4 per thread, one block per 4 KiB piece, no loop.
// tools/probes/stream_copy_probe.hip (profiles/r03a_stream_copy_probe.txt) on 1 GiB + 1 GiB / 4 GiB + 4 GiB buffers: this form
// 6.18 / 6.24 TB/s = the guide's 6.29 for a float4 copy; round 2's form (4 pieces in flight per lane, 4096 blocks) 5.4-5.6;
// grid-stride loops of 1..8 pieces at 8..64 blocks per CU 5.2-5.8, with non-temporal loads AND stores 5.8-6.1;
// hipMemcpyAsync 4.8-5.1. (256 MiB buffers read 6.4-7.0: the Infinity Cache, not HBM.)
__global__ void __launch_bounds__(256) streamCopyKernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// Divergent-gather ceiling of the chip (measurement only, twk_gather_peak): every lane walks its own pseudo-random
// chain of 128-byte lines through a table and reads the whole line as eight 16-byte loads — the access pattern of a
// wide-node fetch. A CU takes one divergent lane address per clock, whatever the occupancy (MI355X: ~610 G
// lane-loads/s from an L2-resident table), and THAT, not HBM, is the memory-side ceiling of traversal over a scene
// that lives in the caches.
__global__ void __launch_bounds__(256) gatherProbeKernel(const float4* __restrict__ table, unsigned int lines, int steps, float* out)
{
  unsigned int line = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % lines;
  float acc = 0.0f;
  for (int s = 0; s < steps; ++s)
  {
    const float4* p = table + (size_t) line * 8;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y + v[k].z;
    line = (__float_as_uint(v[0].w) + threadIdx.x) % lines; // dependent chain, like child references
  }
  if (acc == 12345.678f) out[0] = acc;
}

__global__ void gatherProbeFillKernel(float4* table, size_t count, unsigned int lines)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < count; i += (size_t) gridDim.x * blockDim.x)
  {
    unsigned int x = (unsigned int) i * 1664525u + 1013904223u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    table[i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(x % lines));
  }
}

void launchGatherProbeFill(float4* table, size_t count, unsigned int lines, hipStream_t stream)
{
  hipLaunchKernelGGL(gatherProbeFillKernel, dim3(2048), dim3(256), 0, stream, table, count, lines);
}
void launchGatherProbe(const float4* table, unsigned int lines, int steps, float* out, int gridBlocks, hipStream_t stream)
{
  hipLaunchKernelGGL(gatherProbeKernel, dim3(gridBlocks), dim3(256), 0, stream, table, lines, steps, out);
}

void launchGenerate(const LaunchParams& p, hipStream_t stream)
{
  hipLaunchKernelGGL(generateKernel, dim3((p.numPaths + 255) / 256), dim3(256), 0, stream, p);
}
template<bool PRIMARY, bool LDS_TABLES, bool MEASURE, bool SORT>
static void launchShadeVariant(const LaunchParams& p, int depth, int gridBlocks, hipStream_t stream)
{
  // the variant without what the scene does not have (shade_device.h shadePath): spherical environment, albedo textures
  const bool env = (p.miss == 2), tex = (p.hasAlbedoTexture != 0);
  if (env && tex)  hipLaunchKernelGGL((shadeKernel<true, true, PRIMARY, LDS_TABLES, MEASURE, SORT>),  dim3(gridBlocks), dim3(TWK_SHADE_BLOCK), 0, stream, p, depth);
  else if (env)    hipLaunchKernelGGL((shadeKernel<true, false, PRIMARY, LDS_TABLES, MEASURE, SORT>), dim3(gridBlocks), dim3(TWK_SHADE_BLOCK), 0, stream, p, depth);
  else if (tex)    hipLaunchKernelGGL((shadeKernel<false, true, PRIMARY, LDS_TABLES, MEASURE, SORT>), dim3(gridBlocks), dim3(TWK_SHADE_BLOCK), 0, stream, p, depth);
  else             hipLaunchKernelGGL((shadeKernel<false, false, PRIMARY, LDS_TABLES, MEASURE, SORT>), dim3(gridBlocks), dim3(TWK_SHADE_BLOCK), 0, stream, p, depth);
}
// primary: depth 0 of a pass whose generateKernel was skipped ("primary rays" above)
void launchShade(const LaunchParams& p, int depth, bool primary, int gridBlocks, hipStream_t stream)
{
  const size_t tableBytes = (size_t) p.numInstances * sizeof(DevInstance) + (size_t) p.numMaterials * sizeof(DevMaterial) + (size_t) p.numLights * sizeof(DevLight);
  const bool lds = TWK_SHADE_LDS_TABLES && tableBytes <= (size_t) TWK_SHADE_TABLE_BYTES;
  const bool ldsSort = TWK_SHADE_LDS_TABLES && tableBytes <= (size_t) TWK_SHADE_SORT_TABLE_BYTES; // the builds with the exchange buffer hold smaller tables
  if (p.pathTime != nullptr || p.stats != nullptr) // time view, statistics: the measurement builds
  {
    if (primary) { if (ldsSort) launchShadeVariant<true, true, true, false>(p, depth, gridBlocks, stream);  else launchShadeVariant<true, false, true, false>(p, depth, gridBlocks, stream); }
    else         { if (ldsSort) launchShadeVariant<false, true, true, false>(p, depth, gridBlocks, stream); else launchShadeVariant<false, false, true, false>(p, depth, gridBlocks, stream); }
    return;
  }
  if (p.shadeSort && ldsSort && !(p.shadeSort == 1 && primary)) // the first launch of a pass: neighbouring pixels hit alike, the sort only costs (shade -2.5 %); TWK_SHADE_SORT=2 sorts it too
  {
    if (primary) launchShadeVariant<true, true, false, true>(p, depth, gridBlocks, stream); else launchShadeVariant<false, true, false, true>(p, depth, gridBlocks, stream);
    return;
  }
  if (primary) { if (lds) launchShadeVariant<true, true, false, false>(p, depth, gridBlocks, stream);  else launchShadeVariant<true, false, false, false>(p, depth, gridBlocks, stream); }
  else         { if (lds) launchShadeVariant<false, true, false, false>(p, depth, gridBlocks, stream); else launchShadeVariant<false, false, false, false>(p, depth, gridBlocks, stream); }
}
void launchAccumulate(const LaunchParams& p, hipStream_t stream)
{
  hipLaunchKernelGGL(accumulateKernel, dim3((p.numPixels + 255) / 256), dim3(256), 0, stream, p);
}
void launchCompositor(const float4* tiles, float4* output, int width, int height, int launchWidth, int deviceCount,
                      int tileSizeX, int tileShiftX, int tileShiftY, hipStream_t stream)
{
  hipLaunchKernelGGL(compositorKernel, dim3((launchWidth + 255) / 256, height, deviceCount), dim3(256), 0, stream,
                     tiles, output, width, height, launchWidth, deviceCount, tileSizeX, tileShiftX, tileShiftY);
}
void launchMathTap(int op, const float* x, const float* y, float* out, size_t n, hipStream_t stream)
{
  hipLaunchKernelGGL(mathTapKernel, dim3(1024), dim3(256), 0, stream, op, x, y, out, n);
}
void launchTonemap(const float4* hdr, unsigned char* ldr, size_t numPixels, const TwkTonemapper& tm, hipStream_t stream)
{
  TonemapConstants c;
  c.invGamma       = 1.0f / tm.gamma;
  c.invWhitePoint  = tm.brightness / tm.whitePoint;
  c.burnHighlights = tm.burnHighlights;
  c.crushBlacks    = tm.crushBlacks + tm.crushBlacks + 1.0f;
  c.saturation     = tm.saturation;
  c.colorBalance[0] = tm.colorBalance[0]; c.colorBalance[1] = tm.colorBalance[1]; c.colorBalance[2] = tm.colorBalance[2];
  hipLaunchKernelGGL(tonemapKernel, dim3(2048), dim3(256), 0, stream, hdr, ldr, numPixels, c);
}
void launchStreamCopy(const float4* src, float4* dst, size_t n, hipStream_t stream)
{
  hipLaunchKernelGGL(streamCopyKernel, dim3((unsigned int) ((n + 255) / 256)), dim3(256), 0, stream, src, dst, n);
}

} // namespace twk
