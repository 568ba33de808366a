// Ray traversal kernels (stand where optixTrace stood: raygeneration.cu:84-89 radiance rays,
// closesthit.cu:281-286 shadow rays; the traversal itself lives in closed libnvoptix.so.1).
//
// One persistent launch per bounce serves BOTH ray kinds: the closest-hit rays of bounce k+1 and the
// any-hit shadow rays emitted by the shading of bounce k. A wave takes chunks of consecutive queue slots (the first
// half of the queue interleaved statically, the second half in chunk-sized tickets) and hands them to its lanes as
// they fall idle. Each lane walks the two-level BVH with its own stack — quantised 4-ary wide nodes, TWK_TRACE_STACK_LDS entries in
// LDS laid out [entry][lane] (bank = lane, conflict free); a ray whose stack would overflow is handed to
// traceOverflowKernel, whose single-ray traverse() over the binary nodes continues the stack in HBM. Instances are
// entered by transforming the ray into object space (t is preserved), exactly what an OptiX IAS→GAS descent does;
// instances of tiny geometries (<= TWK_FLATTEN_TRIANGLES triangles: walls, light quads) were flattened into
// world-space triangle slots at twk_build and are tested right at their top-level leaf with the world-space ray.
//
// Triangle test: watertight algorithm of Woop, Benthin, Wald (JCGT 2013), single precision with the
// double fallback on zero edge functions, no fused multiply-add. Ties in t go to the smaller
// (instance, primitive) pair, so the result does not depend on traversal order.
#pragma once
#include "trace_device.h"
#include "shade_device.h" // tex2D, rng, tea for the cutout-opacity test

namespace twk {

// Cutout opacity (anyhit.cu:46-80 radiance, :94-132 shadow): stochastic alpha test of ONE candidate hit. Candidates
// are visited closest-first; radiance rays draw from the path's seed, shadow rays from the stream forked when the ray
// was emitted (shadowPending.w, see shadePath). Returns true when the candidate is ignored; the caller then restarts
// the traversal strictly behind it. The new tmin is written back to the ray record so that a re-trace by
// traceOverflowKernel continues behind the same candidate.
// primary: a ray of the fused first launch (PRIMARY builds): queue 0 holds its seed (stored at the refill) but no ray record —
// the new tmin stays in the caller's register and is handed to traceOverflowKernel through the hit record (see there).
// record: the physical index of the ray's record in its queue (device_types.h "queue segments").
TWK_D bool cutoutIgnoresCandidate(const LaunchParams& p, const TraceResult& res, bool isShadow, int q, unsigned int record, bool primary = false)
{
  const DevInstance& inst = p.instances[res.instance];
  const DevMaterial& material = p.materials[inst.material];
  if (material.textureCutout == 0) return false;
  const float4* sv = p.shadeTriangles + TWK_SHADE_RECORD * (size_t) res.triangleSlot;
  const float4 s5 = sv[5], s6 = sv[6], s7 = sv[7];
  const float alpha = 1.0f - res.beta - res.gamma;
  const V3 texcoord = v3(s5.y, s5.z, s5.w) * alpha + v3(s6.x, s6.y, s6.z) * res.beta + v3(s6.w, s7.x, s7.y) * res.gamma;
  const float opacity = intensity(v3(tex2D(p.textures[1], texcoord.x, texcoord.y)));
  if (!(opacity < 1.0f)) return false;
  float draw;
  if (isShadow)
  {
    float4 pend = p.shadowPending[record];
    unsigned int seed = __float_as_uint(pend.w);
    draw = rng(seed);
    pend.w = __uint_as_float(seed);
    p.shadowPending[record] = pend;
  }
  else
  {
    uint2 sf = p.raySeedFlags[q][record];
    draw = rng(sf.x);
    p.raySeedFlags[q][record] = sf;
  }
  if (!(opacity <= draw)) return false;
  if (isShadow) p.shadowOrg[record].w = res.t; else if (!primary) p.rayOrg[q][record].w = res.t;
  return true;
}

// Persistent traversal launch for bounce `depth`: slots [0, numClosest) are the radiance rays of queue
// (depth & 1), slots [numClosest, numClosest + numShadow) the shadow rays emitted by shade(depth - 1).
//
// Structure (persistent threads with per-lane refill, after Aila & Laine 2009, re-tiled for 64-wide waves):
//   * a wave owns a pool of consecutive queue slots, one chunk at a time: its interleaved static chunks first, then
//     tickets of one chunk from the depth's counter word (see "Wave-uniform pool" below);
//   * every lane carries one ray; when fewer than TWK_TRACE_REFILL lanes still hold a ray the wave leaves the
//     traversal loop and hands fresh slots from its pool to the idle lanes (ballot + prefix popcount, no atomics) —
//     ray lengths on this workload range from 3 to 100+ node visits, and without refill the wave idles on its
//     slowest lane (measured: 10.5 of 64 lanes active per VALU instruction);
//   * "while-while": all lanes first descend inner nodes together, then handle their leaf / instance entry /
//     instance exit once, so a wave does not pay for three code paths per step.
#ifndef TWK_TRACE_REFILL
#define TWK_TRACE_REFILL 52
#endif
#ifndef TWK_TRACE_REFILL_PRIMARY
#define TWK_TRACE_REFILL_PRIMARY 32 // the PRIMARY build computes its rays at the refill: fewer, fuller refills
#endif
// The node loop of a round ends once fewer than NUM/DEN of the lanes that entered it are still at an inner node.
#ifndef TWK_TRACE_TAIL_DEN
#define TWK_TRACE_TAIL_DEN 2
#endif
#ifndef TWK_TRACE_TAIL_MIN
#define TWK_TRACE_TAIL_MIN 2u // queues shorter than this many chunks per wave are dealt statically throughout
#endif
#ifndef TWK_TRACE_NODE_NUM
#define TWK_TRACE_NODE_NUM 1
#endif
#ifndef TWK_TRACE_NODE_DEN
#define TWK_TRACE_NODE_DEN 2
#endif

// TWO_LEVEL = false: every instance of the scene is flattened (device_types.h TWK_LEAF_WORLD) — one world-space tree,
// every leaf a triangle range; the instance entry / exit code is compiled out.
// W7: the seven-blocks-per-CU build of the kernel (device_types.h TWK_TRACE_WAVES7): a 19-entry LDS stack, a 32-node cache.
// PRIMARY: depth 0 of a pass without generateKernel — the lane computes the primary ray of its slot instead of fetching it
// (shade_kernels.hip "primary rays"). With CUTOUT the seed is stored in queue 0 for the opacity draws.
// (The builds over compressed 8-ary nodes, round 4, lost on every scene and live in tools/experiments/r04_wide8_nodes.patch.)
template<bool COUNT, bool CUTOUT, bool TWO_LEVEL, bool W7, bool PRIMARY>
__global__ void __launch_bounds__(TWK_TRACE_BLOCK, W7 ? TWK_TRACE_WAVES7 : (CUTOUT ? ((PRIMARY || TWO_LEVEL) ? TWK_TRACE_WAVES_CUTOUT_OTHER : TWK_TRACE_WAVES) : ((PRIMARY && TWO_LEVEL) ? TWK_TRACE_WAVES_PRIMARY_TWO_LEVEL : (PRIMARY ? TWK_TRACE_WAVES_PRIMARY : TWK_TRACE_WAVES)))) // blocks per CU = waves per SIMD: device_types.h
traceKernel(LaunchParams p, int depth)
{
  constexpr int STACK_LDS = W7 ? TWK_TRACE_STACK_LDS7 : TWK_TRACE_STACK_LDS;
  constexpr int TOP_NODES = W7 ? TWK_TOP_NODES7 : TWK_TOP_NODES;
  constexpr int STACK_ROWS = STACK_LDS + 1; // + 1 dummy row, see the node step
  __shared__ int stackStorage[STACK_ROWS * TWK_TRACE_BLOCK];
  __shared__ float4 topCache[TOP_NODES * TWK_TOP_STRIDE];         // device_types.h TWK_NODE_CACHED / TWK_TOP8_NODES
  int* ldsStack = stackStorage + threadIdx.x;
  const int stride = TWK_TRACE_BLOCK;
  {
    const float4* topSource = W7 ? p.topNodes7 : p.topNodes;
    for (int i = threadIdx.x; i < TOP_NODES * 4; i += TWK_TRACE_BLOCK) topCache[(i >> 2) * TWK_TOP_STRIDE + (i & 3)] = topSource[i];
  }
  __syncthreads();

  // the two queues of this launch, each in segments (device_types.h "queue segments"): a slot of the launch is a VIRTUAL slot
  const QueueSegments closestSegments = queueSegments(&p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_CLOSEST]);
  const QueueSegments shadowSegments  = (depth > 0) ? queueSegments(&p.counters[(depth - 1) * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_SHADOW]) : noSegments();
  const unsigned int numClosest = closestSegments.total, numShadow = shadowSegments.total;
  const unsigned int total = numClosest + numShadow;
  // physical record of a slot in its queue; PRIMARY: queue 0 is the identity
#define TWK_RECORD(slot_) (((slot_) < numClosest) ? (PRIMARY ? (slot_) : physicalSlot(closestSegments, p.queueStride, (slot_))) : physicalSlot(shadowSegments, p.queueStride, (slot_) - numClosest))

  const int q = depth & 1;
  const bool packed = !CUTOUT && !PRIMARY && p.packedQueue != 0 && depth > 0; // device_types.h LaunchParams::packedQueue: the closest-hit rays' .w words are not tmin / tmax
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned long long laneBelow = (1ull << lane) - 1ull;

  unsigned int nodeCount = 0, triCount = 0, instCount = 0, closestCount = 0, shadowCount = 0, maxSteps = 0, cachedCount = 0;
  unsigned int nodeWaveSteps = 0, triWaveSteps = 0, leafWaveSteps = 0; // COUNT: wave-level iterations, tallied by the first active lane (lane occupancy = lane count / (64 * wave steps))
#define TWK_WAVE_STEP(counter) if (COUNT) { if (lane == (unsigned int) (__ffsll((long long) __ballot(true)) - 1)) ++(counter); }
  // COUNT: wave time per phase of the outer loop (TwkLaunchStats::waveCycles), shader clock, wave-uniform
  unsigned long long phaseCycles[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned long long phaseMark = COUNT ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long kernelStart = phaseMark;
#define TWK_PHASE_END(k) if (COUNT) { const unsigned long long now_ = __builtin_readcyclecounter(); phaseCycles[k] += now_ - phaseMark; phaseMark = now_; }

  // Wave-uniform pool of queue slots. The queue is cut into chunks of TWK_TRACE_CHUNK rays. The first half goes out
  // statically and interleaved — chunk c belongs to wave c mod numWaves, so every wave samples the whole queue (rays of
  // one image region cost alike, and regions differ: sky against geometry) — and touches no counter; the second half
  // goes out in tickets of one chunk from the depth's counter word, so the waves that run ahead (the CUs do not all
  // run alike) take what is left and all finish together. A short queue (deep bounces, or one iteration per pass) is
  // spread over all waves statically, one chunk of 16.. rays each.
  // (What each way of dealing the queue measured: DESIGN.md 4.1, first bullet.)
  const unsigned int numWaves = gridDim.x * (TWK_TRACE_BLOCK / 64);
  const unsigned int waveId   = blockIdx.x * (TWK_TRACE_BLOCK / 64) + (threadIdx.x >> 6);
  unsigned int ticketSize = TWK_TRACE_CHUNK;
  const bool longQueue = total >= numWaves * ticketSize * TWK_TRACE_TAIL_MIN;
#if TWK_TRACE_SMALL_CHUNK
  // A short queue (deep bounces; a pass of one or a few iterations) goes out in chunks of ONE wave-load, all static and
  // interleaved like the static half of a long one: chunk c belongs to wave c mod numWaves (full waves on few SIMDs instead of
  // quarter-filled waves on all of them; DESIGN.md 4.1 "short queues").
  if (!longQueue) ticketSize = TWK_TRACE_SMALL_CHUNK;
#else
  if (total < numWaves * ticketSize) ticketSize = min((unsigned int) TWK_TRACE_CHUNK, max(16u, ((total + numWaves - 1u) / numWaves + 15u) & ~15u));
#endif
  unsigned int nextChunk = waveId * ticketSize;
  if (nextChunk >= total) return; // nothing for this wave
  unsigned int poolBase = 0u, poolCount = 0u;
#if TWK_TRACE_TAIL_DEN
  unsigned int* ticket = &p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_TICKET];
  const unsigned int staticEnd = longQueue ? ((total - total / TWK_TRACE_TAIL_DEN) / (numWaves * ticketSize)) * (numWaves * ticketSize) : total;
#else
  const unsigned int staticEnd = total;
#endif
  bool exhausted = false;

  // per-lane ray state
  // Lane flags live in ONE vector register and are changed with and/or. As separate bools they sit in scalar
  // lane masks, and every divergent region that ends merges each of them with three scalar instructions — the leaf
  // step and the triangle loop were mostly such merges (as many scalar as vector instructions in the whole kernel).
  enum : unsigned int
  {
    ST_HAS_RAY    = 1u,  // the lane carries an unfinished ray
    ST_ANY_HIT    = 2u,  // first accepted hit ends the ray (shadow rays, unless the scene has cutout materials)
    ST_DONE       = 4u,  // the ray completed in this round and its result is not yet written
    ST_SHADOW     = 8u,  // slot belongs to the shadow queue
    ST_RETRACE    = 16u, // LDS stack overflow: the ray is re-traced by traceOverflowKernel with the spilling traverse()
    ST_OVERFLOWED = 32u, // set while the overflowed ray is handed over (nothing is written for it here)
    ST_CLOCKED    = 64u  // COUNT builds: rayClock holds the ray's elapsed cycles (stopped in the round it completed in), not its start
  };
  unsigned int state = 0u;
  unsigned int slot = 0;
  V3 org = v3(0.0f), dir = v3(0.0f);
  float tmin = 0.0f;
  TraceResult res; res.t = 0.0f; res.beta = 0.0f; res.gamma = 0.0f; res.instance = -1; res.primitive = -1; res.triangleSlot = -1;
  TraceRay ray; ray.o = v3(0.0f); ray.d = v3(0.0f); ray.id = v3(0.0f);
  WoopConstants woop; woop.perm = 0u; woop.Sx = 0.0f; woop.Sy = 0.0f; woop.Sz = 0.0f;
  int currentInstance = -1, sp = 0, node = TWK_BVH_SENTINEL;
  unsigned int guard = 0;
  unsigned int rayClock = 0; // COUNT, time view: shader clock when this lane took its ray; once the ray has completed (ST_CLOCKED): the cycles it took

  for (;;)
  {
    // ---- the results of the rays that completed since the last refill ----------------------------
    // Written HERE, once per pass of the outer loop, not in the round a ray completes in: a lane that has finished its ray takes
    // no new one before the next refill anyway, so it loses nothing by waiting — and the wave runs this code (two stores; for a
    // shadow ray two dependent fetches and a read-modify-write of the path's radiance) once with every finished lane active
    // instead of in nearly every round with one to three (round 4).
    if (state & ST_DONE)
    {
      // time view: the ray's cycles were stopped in the round it completed in, not here — a finished lane's wait for the rest of
      // its wave is not the path's time (the reference brackets the trace calls of ONE thread with clock(), raygeneration.cu:169-244)
      const unsigned int rayCycles = COUNT ? ((state & ST_CLOCKED) ? rayClock : (unsigned int) __builtin_readcyclecounter() - rayClock) : 0u;
      state &= ~(ST_DONE | ST_CLOCKED);
      const bool isShadow = (state & ST_SHADOW) != 0u;
      if (state & ST_RETRACE)
      {
        // LDS stack overflow: hand the ray to traceOverflowKernel (spilling single-ray traversal), which runs
        // right behind this launch; nothing is written for it here.
        state = (state & ~ST_RETRACE) | ST_OVERFLOWED;
        const unsigned int k = atomicAdd(&p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_OVERFLOW], 1u);
        p.overflowSlots[k] = slot;
        if (PRIMARY && CUTOUT) p.hitRecord[slot] = make_float4(tmin, 0.0f, 0.0f, 0.0f); // where the re-trace continues: behind the candidates ignored so far
      }
      if (COUNT) maxSteps = max(maxSteps, guard);
      const bool ignoredCandidate = CUTOUT && !(state & ST_OVERFLOWED) && res.instance >= 0 &&
                                    cutoutIgnoresCandidate(p, res, isShadow, q, TWK_RECORD(slot), PRIMARY);
      if (ignoredCandidate)
      {
        // continue strictly behind the ignored candidate: the traversal starts again with tmin = its distance. The ray comes from
        // its record again (PRIMARY: is computed again) — kept in registers for this, origin and direction cost the cutout builds
        // 14 VGPRs and their sixth block per CU (rounds 2-3: 93-95 VGPRs, five blocks; round 4: 79, six).
        tmin = res.t;
        float4 o, d;
        if (PRIMARY)
        {
          const PrimaryRay pr = primaryRay(p, slot);
          o = make_float4(pr.origin.x, pr.origin.y, pr.origin.z, 0.0f);
          d = make_float4(pr.direction.x, pr.direction.y, pr.direction.z, RT_DEFAULT_MAX);
        }
        else
        {
          const unsigned int record = TWK_RECORD(slot);
          o = isShadow ? p.shadowOrg[record] : p.rayOrg[q][record];
          d = isShadow ? p.shadowDir[record] : p.rayDir[q][record];
        }
        org = v3(o); dir = v3(d);
        res.t = d.w; res.beta = 0.0f; res.gamma = 0.0f; res.instance = -1; res.primitive = -1; res.triangleSlot = -1;
        setupRay(ray, org, dir);
        woopSetup(dir, woop);
        if (COUNT) rayClock = (unsigned int) __builtin_readcyclecounter() - rayCycles; // the clock runs on behind the ignored candidate
        currentInstance = -1; sp = 0; node = p.topRoot; guard = 0;
        if (p.topRoot2 != TWK_BVH_SENTINEL) { ldsStack[0] = p.topRoot2; sp = 1; }
        state |= ST_HAS_RAY;
      }
      else if (state & ST_OVERFLOWED) { state &= ~ST_OVERFLOWED; }
      else if (!isShadow)
      {
        p.hitRecord[slot]   = make_float4(res.t, res.beta, res.gamma, __int_as_float(res.triangleSlot));
        p.hitInstance[slot] = res.instance;
        if (COUNT) ++closestCount;
        if (COUNT && p.pathTime != nullptr) atomicAdd(&p.pathTime[PRIMARY ? slot : (packed ? (__float_as_uint(p.rayOrg[q][TWK_RECORD(slot)].w) & TWK_PACKED_PIXEL_MASK) : p.rayPixel[q][TWK_RECORD(slot)])], float(rayCycles)); // time view: the lane's cycles from taking the ray to its completion
        if (p.firstHit != nullptr && depth == 0)
        {
          const unsigned int pixel = PRIMARY ? slot : p.rayPixel[q][TWK_RECORD(slot)];
          p.firstHit[pixel] = make_float4(res.t, res.beta, res.gamma, __int_as_float(res.primitive));
          p.firstHitInstance[pixel] = res.instance;
        }
      }
      else
      {
        if (COUNT) ++shadowCount;
        if (COUNT && p.pathTime != nullptr) atomicAdd(&p.pathTime[p.shadowPixel[TWK_RECORD(slot)]], float(rayCycles));
        if (res.instance < 0)
        {
          // visible: add the pending next-event contribution (closesthit.cu:288-299, raygeneration.cu:100)
          const unsigned int s = TWK_RECORD(slot);
          const unsigned int pixel = p.shadowPixel[s];
          const float4 c = p.shadowPending[s];
          float4 r = p.pathRadiance[pixel];
          r.x += c.x; r.y += c.y; r.z += c.z;
          p.pathRadiance[pixel] = r;
        }
      }
    }

    // ---- refill idle lanes from the wave's pool ------------------------------------------------------
    // The wave waits here for the ray records it fetches; the other waves of the SIMD cover that wait (prefetch variants that
    // were built and measured: DESIGN.md 4.1 "ray prefetch").
    {
      const unsigned long long idle = __ballot(!(state & ST_HAS_RAY));
      if (idle != 0ull && !exhausted)
      {
        if (poolCount == 0u)
        {
          if (nextChunk < staticEnd) { poolBase = nextChunk; poolCount = min(ticketSize, staticEnd - nextChunk); nextChunk += numWaves * ticketSize; }
#if TWK_TRACE_TAIL_DEN
          else if (staticEnd < total)
          {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(ticket, ticketSize);
            base = __builtin_amdgcn_readfirstlane(base) + staticEnd;
            if (base >= total) exhausted = true;
            else { poolBase = base; poolCount = min(ticketSize, total - base); }
          }
#endif
          else exhausted = true;
        }
        if (poolCount != 0u)
        {
          const unsigned int rank = (unsigned int) __popcll(idle & laneBelow);
          const unsigned int take = min(poolCount, (unsigned int) __popcll(idle));
          if (!(state & ST_HAS_RAY) && rank < take)
          {
            slot = poolBase + rank;
            float4 o, d;
            int4 entryA = make_int4(0, 0, 0, 0), entryB = make_int4(0, 0, 0, 0);
            if (PRIMARY)
            {
              const PrimaryRay pr = primaryRay(p, slot);
              o = make_float4(pr.origin.x, pr.origin.y, pr.origin.z, p.sceneEpsilon);
              d = make_float4(pr.direction.x, pr.direction.y, pr.direction.z, pr.active ? RT_DEFAULT_MAX : -1.0f);
              state = ST_HAS_RAY;
              if (CUTOUT) p.raySeedFlags[0][slot] = make_uint2(pr.seed, 0u); // the opacity test of this segment draws from the seed in the queue (cutoutIgnoresCandidate), and shade(0) takes it from there
              if (p.tileEntries != nullptr)
              {
                const unsigned int launchIndex = (slot + (unsigned int) p.pathBase) % (unsigned int) p.numPixels;
                const unsigned int lx = launchIndex % (unsigned int) p.launchWidth, ly = launchIndex / (unsigned int) p.launchWidth;
                const int4* tile = p.tileEntries + 2 * ((size_t) (ly / TWK_ENTRY_TILE) * (unsigned int) p.tilesX + lx / TWK_ENTRY_TILE);
                entryA = tile[0]; entryB = tile[1];
              }
            }
            else if (slot < numClosest)
            {
              const unsigned int record = physicalSlot(closestSegments, p.queueStride, slot);
              o = p.rayOrg[q][record]; d = p.rayDir[q][record]; state = ST_HAS_RAY;
              if (packed) { o.w = p.sceneEpsilon; d.w = RT_DEFAULT_MAX; }
            }
            else
            {
              const unsigned int record = physicalSlot(shadowSegments, p.queueStride, slot - numClosest);
              o = p.shadowOrg[record]; d = p.shadowDir[record]; state = ST_HAS_RAY | ST_SHADOW | (CUTOUT ? 0u : ST_ANY_HIT);
            }
            org = v3(o); dir = v3(d); tmin = o.w;
            res.t = d.w; res.beta = 0.0f; res.gamma = 0.0f; res.instance = -1; res.primitive = -1; res.triangleSlot = -1;
            setupRay(ray, org, dir);
            woopSetup(dir, woop); // world-space constants: flattened instances are tested without entering anything
            currentInstance = -1; sp = 0; guard = 0;
            node = p.topRoot;
            if (p.topRoot2 != TWK_BVH_SENTINEL) { ldsStack[0] = p.topRoot2; sp = 1; } // the root's second node (wideRootKernel)
            if (COUNT) rayClock = (unsigned int) __builtin_readcyclecounter();
            if (PRIMARY && entryA.x > 0)
            {
              sp = 0; // the tile's list was opened from both nodes of the root
              // the tile's entry points instead of the root: the first goes next, the others wait on the stack, nearest on top
              node = entryA.y;
              const int count = entryA.x;
              if (count > 6) { ldsStack[sp * stride] = entryB.w; ++sp; }
              if (count > 5) { ldsStack[sp * stride] = entryB.z; ++sp; }
              if (count > 4) { ldsStack[sp * stride] = entryB.y; ++sp; }
              if (count > 3) { ldsStack[sp * stride] = entryB.x; ++sp; }
              if (count > 2) { ldsStack[sp * stride] = entryA.w; ++sp; }
              if (count > 1) { ldsStack[sp * stride] = entryA.z; ++sp; }
            }
          }
          poolBase += take; poolCount -= take;
        }
      }
      if (__ballot((state & ST_HAS_RAY) != 0u) == 0ull)
      {
        if (exhausted) break;
        continue; // the pool was empty: the next chunk was taken just now
      }
    }
    TWK_PHASE_END(0)
    // ---- traverse until enough lanes have finished to be worth a refill ------------------------------
    for (;;)
    {
      const int roundActive = __popcll(__ballot((state & ST_HAS_RAY) != 0u));
      // All lanes descend inner nodes. Kept flat on purpose: the stack lives in LDS only here, push and pop are
      // straight-line predicated code (nested LDS/HBM stack selects compiled to ~70 scalar branch instructions
      // per node). A lane whose stack would overflow abandons this traversal and re-traces its ray with the
      // spilling traverse() (cold path, not taken on the LBVHs of the shipped scenes).
      // (a lane without a ray holds node = TWK_BVH_SENTINEL: one comparison decides who steps)
      while ((unsigned int) node < (unsigned int) TWK_BVH_SENTINEL)
      {
        {
          // one WIDE node = the four grandchildren of binary node `node` (two levels of the binary tree per round of
          // loads), 64 bytes: child boxes as 8-bit grid coordinates of the node's own box (device_types.h "quantised wide
          // node") — four 16-byte lane loads instead of eight.
          float4 n0, n1, n2, n3;
          // hipcc merges the two branches into ONE set of flat_load instructions on a selected generic pointer, and that is the
          // faster form: forcing ds_read for the cached lanes and global_load for the others (empty asm pins in
          // both branches) serialises two waits per step for a wave whose lanes are on both sides — 0.794 -> 0.878 ms/step.
          if (node & TWK_NODE_CACHED)
          {
            const float4* w = topCache + (node & 0xff) * TWK_TOP_STRIDE; // the top of the tree, from LDS
            if (COUNT) ++cachedCount;
            n0 = w[0]; n1 = w[1]; n2 = w[2]; n3 = w[3];
          }
          else
          {
            const float4* w = p.wideQ + 4 * (size_t) node;
            n0 = w[0]; n1 = w[1]; n2 = w[2]; n3 = w[3];
          }
          ++guard;
          if (COUNT) ++nodeCount;
          TWK_WAVE_STEP(nodeWaveSteps)
          int r0 = __float_as_int(n3.x), r1 = __float_as_int(n3.y), r2 = __float_as_int(n3.z), r3 = __float_as_int(n3.w);
          // Pin the references here: left alone, hipcc fetches them with separate loads AFTER the box tests — more
          // dependent L2 round trips per traversal step.
          asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
          // plane distance of grid coordinate q on one axis: (origin + q * cell - o) / d = q * (cell / d) + (origin / d - o / d)
          const float ax = n0.w * ray.id.x, ay = n1.x * ray.id.y, az = n1.y * ray.id.z;
          const float bx = __builtin_fmaf(n0.x, ray.id.x, -ray.ood.x), by = __builtin_fmaf(n0.y, ray.id.y, -ray.ood.y), bz = __builtin_fmaf(n0.z, ray.id.z, -ray.ood.z);
          // Near and far plane of each axis by the sign of the ray direction, chosen ONCE for the four children (their grid
          // coordinates share a word) — six selects instead of a min and a max per plane pair, 24 of them. (Vector
          // instructions other than fma / mul / add cost 1.6 x an fma on this chip, tools/probes/valu_issue_probe.hip, and
          // the node step is bound by their issue.) An unused entry has an inverted box (bvh_build.hip quantizeWideKernel):
          // its near planes lie behind its far planes for every ray.
          const bool negX = ray.id.x < 0.0f, negY = ray.id.y < 0.0f, negZ = ray.id.z < 0.0f;
          const unsigned int qlx = __float_as_uint(n1.z), qly = __float_as_uint(n1.w), qlz = __float_as_uint(n2.x);
          const unsigned int qhx = __float_as_uint(n2.y), qhy = __float_as_uint(n2.z), qhz = __float_as_uint(n2.w);
          const unsigned int qnx = negX ? qhx : qlx, qfx = negX ? qlx : qhx;
          const unsigned int qny = negY ? qhy : qly, qfy = negY ? qly : qhy;
          const unsigned int qnz = negZ ? qhz : qlz, qfz = negZ ? qlz : qhz;
          float t0, t1, t2, t3;
  #define TWK_Q(word, k) ((float) (((word) >> (8 * (k))) & 0xffu)) /* v_cvt_f32_ubyte<k> */
  #define TWK_SLAB(k, tk) slabTestGrid(ax, ay, az, bx, by, bz, TWK_Q(qnx, k), TWK_Q(qny, k), TWK_Q(qnz, k), TWK_Q(qfx, k), TWK_Q(qfy, k), TWK_Q(qfz, k), tmin, res.t, tk)
          const bool h0 = TWK_SLAB(0, t0);
          const bool h1 = TWK_SLAB(1, t1);
          const bool h2 = TWK_SLAB(2, t2);
          const bool h3 = TWK_SLAB(3, t3);
  #undef TWK_SLAB
  #undef TWK_Q
          const float inf = __uint_as_float(0x7f800000u);
          t0 = h0 ? t0 : inf; t1 = h1 ? t1 : inf; t2 = h2 ? t2 : inf; t3 = h3 ? t3 : inf;
          // sort the four (entry distance, reference) pairs, misses last: 5 compare-exchanges
  #define TWK_CE(ta, ra, tb, rb) { const bool sw = (tb) < (ta); const float tl = sw ? (tb) : (ta); (tb) = sw ? (ta) : (tb); (ta) = tl; const int rl = sw ? (rb) : (ra); (rb) = sw ? (ra) : (rb); (ra) = rl; }
          TWK_CE(t0, r0, t1, r1) TWK_CE(t2, r2, t3, r3) TWK_CE(t0, r0, t2, r2) TWK_CE(t1, r1, t3, r3) TWK_CE(t1, r1, t2, r2)
  #undef TWK_CE
          const int hits = (int) h0 + (int) h1 + (int) h2 + (int) h3;
          bool stop = (guard > (1u << 22));
          bool overflow = false;
          if (hits > 0)
          {
            // nearest child next, the others pushed far-to-near; row STACK_LDS of the LDS stack is a dummy
            // that absorbs the unconditional stores once the stack is full
            node = r0;
            ldsStack[min(sp, STACK_LDS) * stride] = r3; overflow |= (hits > 3) & (sp >= STACK_LDS); sp += (hits > 3);
            ldsStack[min(sp, STACK_LDS) * stride] = r2; overflow |= (hits > 2) & (sp >= STACK_LDS); sp += (hits > 2);
            ldsStack[min(sp, STACK_LDS) * stride] = r1; overflow |= (hits > 1) & (sp >= STACK_LDS); sp += (hits > 1);
          }
          else
          {
            stop = stop || (sp == 0);
            sp = max(sp - 1, 0);
            node = ldsStack[sp * stride];
          }
          if (overflow) state |= ST_RETRACE;
          if (stop | overflow) { state = (state & ~ST_HAS_RAY) | ST_DONE; node = TWK_BVH_SENTINEL; }
        }
        // Leave the node loop once most lanes are parked at a leaf: the stragglers resume in the next round
        // together with the lanes that come back from their leaf, instead of running at a few lanes per wave.
        if (__popcll(__ballot((unsigned int) node < (unsigned int) TWK_BVH_SENTINEL)) * TWK_TRACE_NODE_DEN < roundActive * TWK_TRACE_NODE_NUM) break;
      }

      TWK_PHASE_END(1)
      // one leaf / instance-entry / instance-exit step per lane; lanes that left the node loop early are still
      // at an inner node and must NOT take this path (their `node` is not a leaf reference)
      unsigned int pop = 0u; // lane flag kept in a vector register, like `state`
      // Triangle range this lane tests in this round: the slots of a bottom-level leaf (object space, inside an
      // instance) or the world-space slots of a flattened instance at the top level. ONE triangle phase serves both:
      // `woop` and `ray.o` always belong to the space the lane is in.
      int triFirst = 0, triLast = -1;
      if ((state & ST_HAS_RAY) && !((unsigned int) node < (unsigned int) TWK_BVH_SENTINEL))
      {
        TWK_WAVE_STEP(leafWaveSteps)
        if (TWO_LEVEL && node == TWK_BVH_SENTINEL)
        {
          // leaving an instance: the world-space Woop constants saved at entry come back from the stack
          sp -= 4;
          woop.perm = (unsigned int) ldsStack[sp * stride];
          woop.Sx = __int_as_float(ldsStack[(sp + 1) * stride]);
          woop.Sy = __int_as_float(ldsStack[(sp + 2) * stride]);
          woop.Sz = __int_as_float(ldsStack[(sp + 3) * stride]);
          setupRay(ray, org, dir); // back to the world-space ray
          currentInstance = -1;
          pop = 1u;
        }
        else
        {
          const int payload = ~node;
          if (TWO_LEVEL && currentInstance < 0 && !(payload & TWK_LEAF_WORLD))
          {
            const float4* rec = reinterpret_cast<const float4*>(p.instances + payload);
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
            if (COUNT) ++instCount;
            if (sp + 5 > STACK_LDS) state = (state & ~ST_HAS_RAY) | ST_DONE | ST_RETRACE;
            else
            {
              // below the sentinel: the world-space Woop constants, restored when the sentinel is popped
              ldsStack[sp * stride] = (int) woop.perm;
              ldsStack[(sp + 1) * stride] = __float_as_int(woop.Sx);
              ldsStack[(sp + 2) * stride] = __float_as_int(woop.Sy);
              ldsStack[(sp + 3) * stride] = __float_as_int(woop.Sz);
              ldsStack[(sp + 4) * stride] = TWK_BVH_SENTINEL;
              sp += 5;
              float m[12];
              m[0] = r0.x; m[1] = r0.y; m[2] = r0.z; m[3] = r0.w;
              m[4] = r1.x; m[5] = r1.y; m[6] = r1.z; m[7] = r1.w;
              m[8] = r2.x; m[9] = r2.y; m[10] = r2.z; m[11] = r2.w;
              const V3 objOrg = transformPoint(m, org);
              const V3 objDir = transformVector(m, dir);
              woopSetup(objDir, woop);
              setupRay(ray, objOrg, objDir);
              currentInstance = payload;
              node = __float_as_int(r3.x);
            }
          }
          else
          {
            // a leaf of 1..4 consecutive triangle slots (bvh_build.hip: at most TWK_MAX_LEAF, default 2; a flattened instance: all of its triangles)
            triFirst = payload & 0x0fffffff; triLast = triFirst + ((payload >> 28) & 3);
            pop = 1u;
          }
        }

      }

      TWK_PHASE_END(2)
      // ---- triangle phase -----------------------------------------------------------------------------------------
#define TWK_MERGE_HIT(hit_, t_, beta_, gamma_, inst_, prim_, ts_)                                                              \
      {                                                                                                                        \
        const bool closer = (hit_) & (((t_) < res.t) |                                                                         \
                                      (((t_) == res.t) & (res.instance >= 0) &                                                 \
                                       (((inst_) < res.instance) | (((inst_) == res.instance) & ((prim_) < res.primitive))))); \
        res.t = closer ? (t_) : res.t; res.beta = closer ? (beta_) : res.beta; res.gamma = closer ? (gamma_) : res.gamma;      \
        res.instance = closer ? (inst_) : res.instance; res.primitive = closer ? (prim_) : res.primitive;                      \
        res.triangleSlot = closer ? (ts_) : res.triangleSlot;                                                                  \
        if (closer & ((state & ST_ANY_HIT) != 0u))                                                                             \
        {                                                                                                                      \
          pop = 0u; state = (state & ~ST_HAS_RAY) | ST_DONE; triLast = -1;                                                     \
        }                                                                                                                      \
      }
      // (Postponed leaves, slots stored by component, pair fetch, handing a leaf's second triangle to an idle lane: all built,
      // measured and not kept — DESIGN.md 4.1.)
      for (int ts = triFirst; ts <= triLast; ++ts)
      {
        const float4* tri = p.triangles + 3 * (size_t) ts;
        const float4 a = tri[0], b = tri[1], c = tri[2];
        if (COUNT) ++triCount;
        TWK_WAVE_STEP(triWaveSteps)
        float t, beta, gamma;
        const bool hit = woopIntersect(woop, ray.o, v3(a), v3(b), v3(c), tmin, t, beta, gamma);
        const int prim = __float_as_int(a.w);
        const int triInstance = (TWO_LEVEL && currentInstance >= 0) ? currentInstance : __float_as_int(b.w); // world-space slots carry their instance
        TWK_MERGE_HIT(hit, t, beta, gamma, triInstance, prim, ts)
      }
#undef TWK_MERGE_HIT
      TWK_PHASE_END(3)

      if (pop)
      {
        if (sp == 0) state = (state & ~ST_HAS_RAY) | ST_DONE;
        else { --sp; node = ldsStack[sp * stride]; }
      }
      if (COUNT) { if ((state & (ST_DONE | ST_CLOCKED)) == ST_DONE) { rayClock = (unsigned int) __builtin_readcyclecounter() - rayClock; state |= ST_CLOCKED; } }
      TWK_PHASE_END(4)
      if (!(state & ST_HAS_RAY)) node = TWK_BVH_SENTINEL; // what the node loop's condition relies on
      const unsigned long long active = __ballot((state & ST_HAS_RAY) != 0u);
      if (active == 0ull) break;
      if (!exhausted && __popcll(active) < min(PRIMARY ? TWK_TRACE_REFILL_PRIMARY : TWK_TRACE_REFILL, (int) ticketSize)) break;
    }
  }

  if (COUNT)
  {
    const unsigned long long kernelEnd = __builtin_readcyclecounter(); // before the counters' own atomics, which queue up behind each other
    atomicAdd(&p.stats[0], (unsigned long long) closestCount);
    atomicAdd(&p.stats[1], (unsigned long long) shadowCount);
    atomicAdd(&p.stats[2], (unsigned long long) nodeCount);
    atomicAdd(&p.stats[3], (unsigned long long) triCount);
    atomicAdd(&p.stats[4], (unsigned long long) instCount);
    atomicMax(&p.stats[7], (unsigned long long) maxSteps);
    if (nodeWaveSteps) atomicAdd(&p.stats[13], (unsigned long long) nodeWaveSteps);
    if (triWaveSteps)  atomicAdd(&p.stats[14], (unsigned long long) triWaveSteps);
    if (leafWaveSteps) atomicAdd(&p.stats[15], (unsigned long long) leafWaveSteps);
    if (cachedCount)   atomicAdd(&p.stats[16], (unsigned long long) cachedCount);
    if (lane == 0)
    {
      for (int k = 0; k < 5; ++k) atomicAdd(&p.stats[18 + k], phaseCycles[k]);
      atomicAdd(&p.stats[23], kernelEnd - kernelStart);
    }
  }
#undef TWK_WAVE_STEP
#undef TWK_PHASE_END
#undef TWK_RECORD
}

} // namespace twk
