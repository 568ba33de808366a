// ORACLE-SIDE TEST INFRASTRUCTURE — not part of the product, never linked into libtweeker_hip.so.
//
// Host build of the product's kernels: BASELINE.json north_star's "single-threaded C++ CPU fallback of the same kernels
// timed on the host cores". The SAME source the GPU runs — tweeker_raytracer_amd/csrc/shade_device.h (generatePath,
// shadePath with every BSDF / light / miss program, accumulateLaunchIndex), trace_device.h (traverse: the single-ray
// two-level traversal of the overflow, query and tail kernels, slab test, watertight triangle test), device_math.h —
// compiled by g++ with -ffp-contract=off (host_kernels_shim.h supplies the dozen device intrinsics), driven as the same
// wavefront: generate -> [trace -> shade] x depth -> trace -> accumulate over the same SoA queues, on the BVH the DEVICE
// built (twk_debug_snapshot_scene). What is not the same source is what cannot exist on a CPU: the persistent kernel's
// wave-level ray dealing and while-while loop (every ray is walked by traverse() instead — same hits by construction of
// the tie rule, tests/test_gpu_host_kernels.py holds the images equal bit for bit) and the block-aggregated queue
// appends (a plain sequential append: another queue order, the same paths).
// Used by tests (-m gpu: it needs a device-built scene) and by bench.py's cpu_baseline leg, never by the product.
#include "host_kernels_shim.h"
#include "../tweeker_raytracer_amd/csrc/shade_device.h"
#include "../tweeker_raytracer_amd/csrc/trace_device.h"

#include <chrono>
#include <vector>

using namespace twk;

namespace {

template<typename T> T* carve(std::vector<char>& pool, size_t& offset, size_t count)
{
  offset = (offset + 15) & ~(size_t) 15;
  T* p = reinterpret_cast<T*>(pool.data() + offset);
  offset += count * sizeof(T);
  return p;
}

} // namespace

extern "C" {

// launchParams: what twk_debug_snapshot_scene wrote. Renders iterations [firstIteration, firstIteration + batch) as ONE
// wavefront pass on the calling thread into `output` (launchWidth x height float4 running mean, read and written like
// the device's accumulation buffer). counts (may be null): radiance rays, shadow rays, node visits, triangle tests,
// instance entries, shaded segments. Returns 0, or 1 for bad arguments / a scene the host build does not cover.
int hostk_render(const void* launchParams, size_t paramsBytes, unsigned int firstIteration, int batch, float* output, double* seconds, unsigned long long* counts)
{
  if (!launchParams || paramsBytes != sizeof(LaunchParams) || !output || batch < 1) return 1;
  LaunchParams p;
  memcpy(&p, launchParams, sizeof(p));
  if (p.hasCutout) return 1; // the stochastic any-hit candidate loop lives in trace_kernels.hip (device only)
  const size_t numPixels = (size_t) p.numPixels, n = numPixels * (size_t) batch;
  const int maxDepth = p.pathLengths[1];

  std::vector<char> pool(n * (15 * sizeof(float4) + 2 * sizeof(uint2) + 6 * sizeof(unsigned int)) + 4096 + sizeof(unsigned int) * TWK_COUNTERS_PER_DEPTH * (TWK_MAX_DEPTH + 2));
  size_t off = 0;
  for (int k = 0; k < 2; ++k)
  {
    p.rayOrg[k] = carve<float4>(pool, off, n); p.rayDir[k] = carve<float4>(pool, off, n); p.rayThroughput[k] = carve<float4>(pool, off, n);
    p.raySeedFlags[k] = carve<uint2>(pool, off, n); p.rayPixel[k] = carve<unsigned int>(pool, off, n);
  }
  p.hitRecord = carve<float4>(pool, off, n); p.hitInstance = carve<int>(pool, off, n);
  p.shadowOrg = carve<float4>(pool, off, n); p.shadowDir = carve<float4>(pool, off, n); p.shadowPending = carve<float4>(pool, off, n);
  p.shadowPixel = carve<unsigned int>(pool, off, n);
  p.pathRadiance = carve<float4>(pool, off, n);
  p.volumeStack = carve<float4>(pool, off, 4 * n);
  p.counters = carve<unsigned int>(pool, off, TWK_COUNTERS_PER_DEPTH * (TWK_MAX_DEPTH + 2));
  memset(p.counters, 0, sizeof(unsigned int) * TWK_COUNTERS_PER_DEPTH * (TWK_MAX_DEPTH + 2));
  unsigned int dropped = 0;
  p.droppedPushes = &dropped;
  p.output = reinterpret_cast<float4*>(output);
  p.outputFrame = 0;
  p.iterationIndex = firstIteration; p.batchCount = batch; p.numPaths = (int) n; p.pathBase = 0;

  std::vector<int> ldsStack((size_t) TWK_TRACE_STACK_LDS * TWK_TRACE_BLOCK, 0), spill(TWK_TRACE_STACK_SPILL, 0);
  unsigned int nodeCount = 0, triCount = 0, instCount = 0;
  unsigned long long radianceRays = 0, shadowRays = 0, nodes = 0, tris = 0, insts = 0, shaded = 0;
  const auto t0 = std::chrono::steady_clock::now();

  // generateKernel
  for (size_t index = 0; index < n; ++index) generatePath(p, (unsigned int) index);

  auto trace = [&](int depth)
  {
    // traceKernel's job for bounce `depth`: closest hits of queue depth & 1, then the shadow rays shade(depth - 1) emitted
    // (the host build appends in slot order to ONE segment of each queue — segment 0: device_types.h "queue segments")
    const unsigned int numClosest = p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_CLOSEST];
    const unsigned int numShadow = (depth > 0) ? p.counters[(depth - 1) * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_SHADOW] : 0u;
    const int q = depth & 1;
    for (unsigned int slot = 0; slot < numClosest; ++slot)
    {
      const float4 o = p.rayOrg[q][slot], d = p.rayDir[q][slot];
      TraceResult res;
      nodeCount = triCount = instCount = 0;
      traverse<true>(p, v3(o), v3(d), o.w, d.w, false, ldsStack.data(), spill.data(), res, nodeCount, triCount, instCount);
      nodes += nodeCount; tris += triCount; insts += instCount; ++radianceRays;
      p.hitRecord[slot] = make_float4(res.t, res.beta, res.gamma, __int_as_float(res.triangleSlot));
      p.hitInstance[slot] = res.instance;
    }
    for (unsigned int s = 0; s < numShadow; ++s)
    {
      const float4 o = p.shadowOrg[s], d = p.shadowDir[s];
      TraceResult res;
      nodeCount = triCount = instCount = 0;
      traverse<true>(p, v3(o), v3(d), o.w, d.w, true, ldsStack.data(), spill.data(), res, nodeCount, triCount, instCount);
      nodes += nodeCount; tris += triCount; insts += instCount; ++shadowRays;
      if (res.instance < 0)
      {
        // visible: add the pending next-event contribution (trace_kernels.hip, closesthit.cu:288-299)
        const unsigned int pixel = p.shadowPixel[s];
        const float4 c = p.shadowPending[s];
        float4 r = p.pathRadiance[pixel];
        r.x += c.x; r.y += c.y; r.z += c.z;
        p.pathRadiance[pixel] = r;
      }
    }
  };

  for (int depth = 0; depth < maxDepth; ++depth)
  {
    trace(depth);
    // shadeKernel: shadePath per queue slot, continuation and shadow rays appended in slot order
    const unsigned int numRays = p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_CLOSEST];
    const int q = depth & 1, qn = q ^ 1;
    unsigned int& nextCount = p.counters[(depth + 1) * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_CLOSEST];
    unsigned int& shadowCount = p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_SHADOW];
    for (unsigned int slot = 0; slot < numRays; ++slot)
    {
      const float4 ro = p.rayOrg[q][slot], rd = p.rayDir[q][slot];
      if (!(rd.w >= 0.0f)) continue; // an inactive launch index (tile column beyond the image)
      const unsigned int pixel = p.rayPixel[q][slot];
      ShadeOutput out;
      out.alive = false; out.wantShadow = false;
      out.throughputPdf = p.rayThroughput[q][slot];
      out.seedFlags = p.raySeedFlags[q][slot];
      shadePath<true, true>(p, depth, pixel, ro, rd, p.hitRecord[slot], p.hitInstance[slot], out);
      ++shaded;
      if (out.wantShadow)
      {
        const unsigned int s = shadowCount++;
        p.shadowOrg[s]     = make_float4(out.nextPos.x, out.nextPos.y, out.nextPos.z, p.sceneEpsilon);
        p.shadowDir[s]     = make_float4(out.shadowDir.x, out.shadowDir.y, out.shadowDir.z, out.shadowTmax);
        p.shadowPixel[s]   = pixel;
        p.shadowPending[s] = make_float4(out.pending.x, out.pending.y, out.pending.z, __uint_as_float(out.shadowSeed));
      }
      if (out.alive)
      {
        const unsigned int k = nextCount++;
        p.rayOrg[qn][k]   = make_float4(out.nextPos.x, out.nextPos.y, out.nextPos.z, p.sceneEpsilon);
        p.rayDir[qn][k]   = make_float4(out.nextDir.x, out.nextDir.y, out.nextDir.z, RT_DEFAULT_MAX);
        p.rayPixel[qn][k] = pixel;
        p.rayThroughput[qn][k] = out.throughputPdf;
        p.raySeedFlags[qn][k]  = out.seedFlags;
      }
    }
  }
  if (maxDepth > 0) trace(maxDepth); // the shadow rays of the last shade

  // accumulateKernel
  for (size_t index = 0; index < numPixels; ++index) accumulateLaunchIndex(p, (unsigned int) index);

  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (counts) { counts[0] = radianceRays; counts[1] = shadowRays; counts[2] = nodes; counts[3] = tris; counts[4] = insts; counts[5] = shaded; }
  return dropped ? 1 : 0;
}

size_t hostk_params_bytes(void) { return sizeof(LaunchParams); }

} // extern "C"
