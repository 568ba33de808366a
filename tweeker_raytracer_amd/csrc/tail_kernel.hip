// Tail kernel: the deep bounces of a launch in ONE persistent kernel.
//
// The wavefront bounces (trace → shade per depth) are efficient while the queues are long. From bounce 3 on the
// Cornell-box queues hold 5–15 % of the pixels and every per-depth launch pair is bound by its slowest ray, not by
// throughput (measured: 7 launches of 170–380 us for ~0.1 M rays each). Here each lane adopts one surviving path and
// walks it to its end — shade, shadow ray, next closest hit, shade, … — with the same shadePath() and traverse()
// the wavefront kernels use, so the arithmetic, the RNG draw order and the order of the radiance additions per pixel
// are unchanged (and the image stays bit-identical to the all-wavefront schedule and to the oracle).
// A lane whose path ended adopts the next queue slot from its wave's pool (one atomic ticket per 64 slots).
//
// Input: queue (depth0 & 1) with the hit records of traceKernel(depth0) already in place; that launch also resolved
// the shadow rays of bounce depth0 - 1, so no radiance addition of an earlier bounce can race with this kernel.
#include "trace_device.h"
#include "shade_device.h"

namespace twk {

template<bool COUNT>
__global__ void __launch_bounds__(TWK_TRACE_BLOCK)
tailKernel(LaunchParams p, int depth0)
{
  __shared__ int stackStorage[TWK_TRACE_STACK_LDS * TWK_TRACE_BLOCK];
  int* ldsStack = stackStorage + threadIdx.x;
  int* spill = p.traceStackSpill + (size_t) (blockIdx.x * blockDim.x + threadIdx.x) * TWK_TRACE_STACK_SPILL;

  const unsigned int numPaths = p.counters[depth0 * TWK_COUNTERS_PER_DEPTH + 0];
  unsigned int* ticket = &p.counters[(TWK_MAX_DEPTH + 1) * TWK_COUNTERS_PER_DEPTH + 2];
  const int q = depth0 & 1;
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned long long laneBelow = (1ull << lane) - 1ull;

  unsigned int rays = 0, nodeCount = 0, triCount = 0, instCount = 0, statHit = 0, statMiss = 0;

  unsigned int poolBase = 0, poolCount = 0;
  bool exhausted = false;

  bool has = false, haveHit = false;
  unsigned int pixel = 0;
  float4 ro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rd = ro, hit = ro, throughputPdf = ro;
  uint2 seedFlags = make_uint2(0u, 0u);
  int inst = -1, depth = depth0;

  for (;;)
  {
    // adopt new paths
    const unsigned long long idle = __ballot(!has);
    if (idle != 0ull && !exhausted)
    {
      if (poolCount == 0u)
      {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(ticket, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= numPaths) exhausted = true;
        else { poolBase = base; poolCount = min(64u, numPaths - base); }
      }
      if (poolCount != 0u)
      {
        const unsigned int rank = (unsigned int) __popcll(idle & laneBelow);
        const unsigned int take = min(poolCount, (unsigned int) __popcll(idle));
        if (!has && rank < take)
        {
          const unsigned int slot = poolBase + rank;
          ro = p.rayOrg[q][slot];
          rd = p.rayDir[q][slot];
          pixel = p.rayPixel[q][slot];
          hit = p.hitRecord[slot];
          inst = p.hitInstance[slot];
          throughputPdf = p.rayThroughput[q][slot];
          seedFlags = p.raySeedFlags[q][slot];
          depth = depth0;
          haveHit = true;
          has = (rd.w >= 0.0f); // inactive launch indices (tile padding) never get this deep, but stay safe
        }
        poolBase += take; poolCount -= take;
      }
    }
    if (__ballot(has) == 0ull)
    {
      if (exhausted) break;
      continue;
    }

    if (has)
    {
      if (!haveHit)
      {
        TraceResult res;
        traverse<COUNT>(p, v3(ro), v3(rd), ro.w, rd.w, false, ldsStack, spill, res, nodeCount, triCount, instCount);
        hit = make_float4(res.t, res.beta, res.gamma, __int_as_float(res.triangleSlot));
        inst = res.instance;
        if (COUNT) ++rays;
      }

      ShadeOutput out;
      out.throughputPdf = throughputPdf; out.seedFlags = seedFlags;
      shadePath(p, depth, pixel, ro, rd, hit, inst, out);
      if (COUNT) { if (inst < 0) ++statMiss; else ++statHit; }

      if (out.wantShadow)
      {
        TraceResult res;
        traverse<COUNT>(p, out.nextPos, out.shadowDir, p.sceneEpsilon, out.shadowTmax, true, ldsStack, spill, res, nodeCount, triCount, instCount);
        if (COUNT) ++rays;
        if (res.instance < 0)
        {
          float4 r = p.pathRadiance[pixel];
          r.x += out.pending.x; r.y += out.pending.y; r.z += out.pending.z;
          p.pathRadiance[pixel] = r;
        }
      }

      if (out.alive)
      {
        ro = make_float4(out.nextPos.x, out.nextPos.y, out.nextPos.z, p.sceneEpsilon);
        rd = make_float4(out.nextDir.x, out.nextDir.y, out.nextDir.z, RT_DEFAULT_MAX);
        throughputPdf = out.throughputPdf; seedFlags = out.seedFlags;
        ++depth;
        haveHit = false;
      }
      else has = false;
    }
  }

  if (COUNT)
  {
    atomicAdd(&p.stats[8],  (unsigned long long) rays);
    atomicAdd(&p.stats[9],  (unsigned long long) nodeCount);
    atomicAdd(&p.stats[10], (unsigned long long) triCount);
    atomicAdd(&p.stats[11], (unsigned long long) instCount);
    atomicAdd(&p.stats[5],  (unsigned long long) statHit);
    atomicAdd(&p.stats[6],  (unsigned long long) statMiss);
  }
}

void launchTail(const LaunchParams& p, int depth0, bool count, int gridBlocks, hipStream_t stream)
{
  if (count) hipLaunchKernelGGL(tailKernel<true>,  dim3(gridBlocks), dim3(TWK_TRACE_BLOCK), 0, stream, p, depth0);
  else       hipLaunchKernelGGL(tailKernel<false>, dim3(gridBlocks), dim3(TWK_TRACE_BLOCK), 0, stream, p, depth0);
}

} // namespace twk
