// Ray traversal kernels (stand where optixTrace stood: raygeneration.cu:84-89 radiance rays,
// closesthit.cu:281-286 shadow rays; the traversal itself lives in closed libnvoptix.so.1).
//
// One persistent launch per bounce serves BOTH ray kinds: the closest-hit rays of bounce k+1 and the
// any-hit shadow rays emitted by the shading of bounce k. Waves pull 64 consecutive queue slots with
// one atomic ticket per wave, so ray and hit records are read and written as full 1-KiB coalesced
// wave accesses (two 16-byte loads per lane). Each lane walks the two-level BVH2 with its own stack:
// the first TWK_TRACE_STACK_LDS entries in LDS laid out [entry][lane] (bank = lane, conflict free),
// overflow in a per-lane HBM segment. Instances are entered by transforming the ray into object
// space (t is preserved), exactly what an OptiX IAS→GAS descent does.
//
// Triangle test: watertight algorithm of Woop, Benthin, Wald (JCGT 2013), single precision with the
// double fallback on zero edge functions, no fused multiply-add. Ties in t go to the smaller
// (instance, primitive) pair, so the result does not depend on traversal order.
#include "device_types.h"

namespace twk {

struct TraceRay
{
  V3 o, d, id; // origin, direction, guarded reciprocal direction
};

TWK_D float guardedReciprocal(float d)
{
  // Parallel-to-slab rays: a huge finite reciprocal keeps 0 * inf = NaN out of the slab test and
  // makes "origin on the slab plane" count as inside (conservative).
  return (fabsf(d) >= 1.0e-20f) ? 1.0f / d : copysignf(1.0e20f, d);
}

TWK_D void setupRay(TraceRay& r, const V3& o, const V3& d)
{
  r.o = o; r.d = d;
  r.id = v3(guardedReciprocal(d.x), guardedReciprocal(d.y), guardedReciprocal(d.z));
}

// Conservative slab test of one child box. Returns entry distance, or a value > tfar limit on miss.
TWK_D bool slabTest(const TraceRay& r, float lox, float loy, float loz, float hix, float hiy, float hiz, float tmin, float tmax, float& tnear)
{
  const float x0 = (lox - r.o.x) * r.id.x, x1 = (hix - r.o.x) * r.id.x;
  const float y0 = (loy - r.o.y) * r.id.y, y1 = (hiy - r.o.y) * r.id.y;
  const float z0 = (loz - r.o.z) * r.id.z, z1 = (hiz - r.o.z) * r.id.z;
  const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
  const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
  tnear = tn;
  return tn * 0.9999995f <= tf * 1.0000005f;
}

struct WoopConstants
{
  int   kx, ky, kz;
  float Sx, Sy, Sz;
};

TWK_D float component(const V3& v, int k) { return (k == 0) ? v.x : ((k == 1) ? v.y : v.z); }

TWK_D void woopSetup(const V3& d, WoopConstants& w)
{
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  int kx = (kz == 2) ? 0 : kz + 1;
  int ky = (kx == 2) ? 0 : kx + 1;
  const float dz = component(d, kz);
  if (dz < 0.0f) { const int s = kx; kx = ky; ky = s; }
  w.kx = kx; w.ky = ky; w.kz = kz;
  w.Sx = component(d, kx) / dz;
  w.Sy = component(d, ky) / dz;
  w.Sz = 1.0f / dz;
}

TWK_D bool woopIntersect(const WoopConstants& w, const V3& o, const V3& p0, const V3& p1, const V3& p2,
                         float tmin, float& t, float& beta, float& gamma)
{
  const V3 A = p0 - o, B = p1 - o, C = p2 - o;
  const float Akx = component(A, w.kx), Aky = component(A, w.ky), Akz = component(A, w.kz);
  const float Bkx = component(B, w.kx), Bky = component(B, w.ky), Bkz = component(B, w.kz);
  const float Ckx = component(C, w.kx), Cky = component(C, w.ky), Ckz = component(C, w.kz);

  const float Ax = Akx - w.Sx * Akz, Ay = Aky - w.Sy * Akz;
  const float Bx = Bkx - w.Sx * Bkz, By = Bky - w.Sy * Bkz;
  const float Cx = Ckx - w.Sx * Ckz, Cy = Cky - w.Sy * Ckz;

  float U = Cx * By - Cy * Bx;
  float V = Ax * Cy - Ay * Cx;
  float W = Bx * Ay - By * Ax;

  if (U == 0.0f || V == 0.0f || W == 0.0f)
  {
    const double CxBy = (double) Cx * (double) By, CyBx = (double) Cy * (double) Bx;
    U = (float) (CxBy - CyBx);
    const double AxCy = (double) Ax * (double) Cy, AyCx = (double) Ay * (double) Cx;
    V = (float) (AxCy - AyCx);
    const double BxAy = (double) Bx * (double) Ay, ByAx = (double) By * (double) Ax;
    W = (float) (BxAy - ByAx);
  }

  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;

  const float det = U + V + W;
  if (det == 0.0f) return false;

  const float Az = w.Sz * Akz, Bz = w.Sz * Bkz, Cz = w.Sz * Ckz;
  const float T = U * Az + V * Bz + W * Cz;
  const float rcpDet = 1.0f / det;
  const float tt = T * rcpDet;
  if (!(tt > tmin)) return false;

  t     = tt;
  beta  = V * rcpDet;
  gamma = W * rcpDet;
  return true;
}

struct TraceResult
{
  float t, beta, gamma;
  int   instance, primitive;
};

// COUNT: tally node / triangle / instance visits (measurement builds only).
template<bool COUNT>
TWK_D void traverse(const LaunchParams& p, const V3& org, const V3& dir, float tmin, float tmax, bool anyHit,
                    int* ldsStack /* [entry * blockDim + tid] base at tid */, int* spill, TraceResult& res,
                    unsigned int& nodeCount, unsigned int& triCount, unsigned int& instCount)
{
  const int stride = TWK_TRACE_BLOCK;
  res.t = tmax; res.beta = 0.0f; res.gamma = 0.0f; res.instance = -1; res.primitive = -1;

  TraceRay ray;
  setupRay(ray, org, dir);
  WoopConstants woop;
  V3 objOrg = org;
  int currentInstance = -1;

  int sp = 0;
  int node = p.tlasRoot;
  unsigned int guard = 0; // a well-formed tree never gets near this; keeps a corrupted one from hanging the GPU

#define TWK_PUSH(v) do { if (sp < TWK_TRACE_STACK_LDS) ldsStack[sp * stride] = (v); else if (sp < TWK_TRACE_STACK_LDS + TWK_TRACE_STACK_SPILL) spill[sp - TWK_TRACE_STACK_LDS] = (v); ++sp; } while (0)
#define TWK_POP(v)  do { --sp; (v) = (sp < TWK_TRACE_STACK_LDS) ? ldsStack[sp * stride] : ((sp < TWK_TRACE_STACK_LDS + TWK_TRACE_STACK_SPILL) ? spill[sp - TWK_TRACE_STACK_LDS] : TWK_BVH_SENTINEL); } while (0)

  for (;;)
  {
    if (++guard > (1u << 22)) break;
    if (node == TWK_BVH_SENTINEL)
    {
      // leaving an instance: back to the world-space ray
      setupRay(ray, org, dir);
      currentInstance = -1;
      if (sp == 0) break;
      TWK_POP(node);
      continue;
    }

    if (node >= 0)
    {
      const float4* n = reinterpret_cast<const float4*>(p.nodes + node);
      const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
      if (COUNT) ++nodeCount;
      float t0, t1;
      const bool h0 = slabTest(ray, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, tmin, res.t, t0);
      const bool h1 = slabTest(ray, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, tmin, res.t, t1);
      const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
      if (h0 && h1)
      {
        const bool firstIs0 = (t0 <= t1);
        TWK_PUSH(firstIs0 ? c1 : c0);
        node = firstIs0 ? c0 : c1;
      }
      else if (h0) node = c0;
      else if (h1) node = c1;
      else
      {
        if (sp == 0) break;
        TWK_POP(node);
      }
      continue;
    }

    // leaf
    const int payload = ~node;
    if (currentInstance < 0)
    {
      // top level: enter the instance
      const DevInstance* inst = p.instances + payload;
      if (COUNT) ++instCount;
      float m[12];
      const float4* mw = reinterpret_cast<const float4*>(inst->worldToObject);
      const float4 r0 = mw[0], r1 = mw[1], r2 = mw[2];
      m[0] = r0.x; m[1] = r0.y; m[2] = r0.z; m[3] = r0.w;
      m[4] = r1.x; m[5] = r1.y; m[6] = r1.z; m[7] = r1.w;
      m[8] = r2.x; m[9] = r2.y; m[10] = r2.z; m[11] = r2.w;
      objOrg = transformPoint(m, org);
      const V3 objDir = transformVector(m, dir);
      setupRay(ray, objOrg, objDir);
      woopSetup(objDir, woop);
      currentInstance = payload;
      TWK_PUSH(TWK_BVH_SENTINEL);
      node = inst->blasRoot;
      continue;
    }

    // bottom level: one triangle slot
    {
      const float4* tri = p.triangles + 3 * (size_t) payload;
      const float4 a = tri[0], b = tri[1], c = tri[2];
      if (COUNT) ++triCount;
      float t, beta, gamma;
      if (woopIntersect(woop, objOrg, v3(a), v3(b), v3(c), tmin, t, beta, gamma))
      {
        const int prim = __float_as_int(a.w);
        const bool closer = (t < res.t) ||
                            (t == res.t && res.instance >= 0 &&
                             (currentInstance < res.instance || (currentInstance == res.instance && prim < res.primitive)));
        if (closer)
        {
          res.t = t; res.beta = beta; res.gamma = gamma; res.instance = currentInstance; res.primitive = prim;
          if (anyHit) break;
        }
      }
      if (sp == 0) break;
      TWK_POP(node);
    }
  }
#undef TWK_PUSH
#undef TWK_POP
}

// Persistent traversal launch for bounce `depth`: slots [0, numClosest) are the radiance rays of queue
// (depth & 1), slots [numClosest, numClosest + numShadow) the shadow rays emitted by shade(depth - 1).
template<bool COUNT>
__global__ void __launch_bounds__(TWK_TRACE_BLOCK)
traceKernel(LaunchParams p, int depth)
{
  __shared__ int stackStorage[TWK_TRACE_STACK_LDS * TWK_TRACE_BLOCK];
  int* ldsStack = stackStorage + threadIdx.x;
  int* spill = p.traceStackSpill + (size_t) (blockIdx.x * blockDim.x + threadIdx.x) * TWK_TRACE_STACK_SPILL;

  const unsigned int numClosest = p.counters[depth * TWK_COUNTERS_PER_DEPTH + 0];
  const unsigned int numShadow  = (depth > 0) ? p.counters[(depth - 1) * TWK_COUNTERS_PER_DEPTH + 1] : 0u;
  const unsigned int total = numClosest + numShadow;
  unsigned int* ticket = &p.counters[depth * TWK_COUNTERS_PER_DEPTH + 2];

  const int q = depth & 1;
  const unsigned int lane = threadIdx.x & 63u;

  unsigned int nodeCount = 0, triCount = 0, instCount = 0, closestCount = 0, shadowCount = 0;

  for (;;)
  {
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(ticket, 64u);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base >= total) break;

    const unsigned int slot = base + lane;
    if (slot < total)
    {
      if (slot < numClosest)
      {
        const float4 o = p.rayOrg[q][slot];
        const float4 d = p.rayDir[q][slot];
        TraceResult res;
        traverse<COUNT>(p, v3(o), v3(d), o.w, d.w, false, ldsStack, spill, res, nodeCount, triCount, instCount);
        p.hitRecord[slot]   = make_float4(res.t, res.beta, res.gamma, __int_as_float(res.primitive));
        p.hitInstance[slot] = res.instance;
        if (COUNT) ++closestCount;
        if (p.firstHit != nullptr && depth == 0)
        {
          const unsigned int pixel = p.rayPixel[q][slot];
          p.firstHit[pixel] = make_float4(res.t, res.beta, res.gamma, __int_as_float(res.primitive));
          p.firstHitInstance[pixel] = res.instance;
        }
      }
      else
      {
        const unsigned int s = slot - numClosest;
        const float4 o = p.shadowOrg[s];
        const float4 d = p.shadowDir[s];
        TraceResult res;
        traverse<COUNT>(p, v3(o), v3(d), o.w, d.w, true, ldsStack, spill, res, nodeCount, triCount, instCount);
        if (COUNT) ++shadowCount;
        if (res.instance < 0)
        {
          // visible: add the pending next-event contribution (closesthit.cu:288-299, raygeneration.cu:100)
          const unsigned int pixel = p.shadowPixel[s];
          const float4 c = p.shadowPending[s];
          float4 r = p.pathRadiance[pixel];
          r.x += c.x; r.y += c.y; r.z += c.z;
          p.pathRadiance[pixel] = r;
        }
      }
    }
  }

  if (COUNT)
  {
    atomicAdd(&p.stats[0], (unsigned long long) closestCount);
    atomicAdd(&p.stats[1], (unsigned long long) shadowCount);
    atomicAdd(&p.stats[2], (unsigned long long) nodeCount);
    atomicAdd(&p.stats[3], (unsigned long long) triCount);
    atomicAdd(&p.stats[4], (unsigned long long) instCount);
  }
}

// Stand-alone query kernel for twk_trace_rays (parity taps): rays 8 floats each.
__global__ void __launch_bounds__(TWK_TRACE_BLOCK)
traceQueryKernel(LaunchParams p, const float* __restrict__ rays, unsigned int numRays, int anyHit,
                 float* __restrict__ tBetaGamma, int* __restrict__ ids)
{
  __shared__ int stackStorage[TWK_TRACE_STACK_LDS * TWK_TRACE_BLOCK];
  int* ldsStack = stackStorage + threadIdx.x;
  int* spill = p.traceStackSpill + (size_t) (blockIdx.x * blockDim.x + threadIdx.x) * TWK_TRACE_STACK_SPILL;
  unsigned int n0 = 0, n1 = 0, n2 = 0;
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < numRays; i += gridDim.x * blockDim.x)
  {
    const float* r = rays + 8 * (size_t) i;
    TraceResult res;
    traverse<false>(p, v3(r[0], r[1], r[2]), v3(r[4], r[5], r[6]), r[3], r[7], anyHit != 0, ldsStack, spill, res, n0, n1, n2);
    if (anyHit)
    {
      tBetaGamma[3 * i] = 0.0f; tBetaGamma[3 * i + 1] = 0.0f; tBetaGamma[3 * i + 2] = 0.0f;
      ids[2 * i] = (res.instance >= 0) ? 1 : 0; ids[2 * i + 1] = -1;
    }
    else
    {
      tBetaGamma[3 * i] = res.t; tBetaGamma[3 * i + 1] = res.beta; tBetaGamma[3 * i + 2] = res.gamma;
      ids[2 * i] = res.instance; ids[2 * i + 1] = res.primitive;
    }
  }
}

void launchTrace(const LaunchParams& p, int depth, bool count, int gridBlocks, hipStream_t stream)
{
  if (count) hipLaunchKernelGGL(traceKernel<true>,  dim3(gridBlocks), dim3(TWK_TRACE_BLOCK), 0, stream, p, depth);
  else       hipLaunchKernelGGL(traceKernel<false>, dim3(gridBlocks), dim3(TWK_TRACE_BLOCK), 0, stream, p, depth);
}

void launchTraceQuery(const LaunchParams& p, const float* rays, unsigned int numRays, int anyHit, float* tBetaGamma, int* ids, int gridBlocks, hipStream_t stream)
{
  hipLaunchKernelGGL(traceQueryKernel, dim3(gridBlocks), dim3(TWK_TRACE_BLOCK), 0, stream, p, rays, numRays, anyHit, tBetaGamma, ids);
}

} // namespace twk
