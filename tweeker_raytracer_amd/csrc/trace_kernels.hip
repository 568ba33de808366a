// Launchers of the persistent traversal kernel (trace_persistent.h), the kernel that re-traces the rays whose LDS stack
// overflowed, the stand-alone query kernel and the entry points of the primary rays.
#include "trace_persistent.h"

namespace twk {

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

// Rays whose traversal overflowed the LDS stack of the persistent kernel (none on the shipped scenes): traced again
// with the single-ray traversal whose stack continues in HBM. Launched behind every traceKernel; exits at once when
// the list is empty.
template<bool COUNT, bool CUTOUT, bool PRIMARY>
__global__ void __launch_bounds__(TWK_TRACE_BLOCK)
traceOverflowKernel(LaunchParams p, int depth)
{
  __shared__ int stackStorage[TWK_TRACE_STACK_LDS * TWK_TRACE_BLOCK];
  const unsigned int count = p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_OVERFLOW];
  if (count == 0u) return;
  int* ldsStack = stackStorage + threadIdx.x;
  int* spill = p.traceStackSpill + (size_t) (blockIdx.x * blockDim.x + threadIdx.x) * TWK_TRACE_STACK_SPILL;
  const QueueSegments closestSegments = queueSegments(&p.counters[depth * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_CLOSEST]);
  const QueueSegments shadowSegments  = (depth > 0) ? queueSegments(&p.counters[(depth - 1) * TWK_COUNTERS_PER_DEPTH + TWK_COUNTER_SHADOW]) : noSegments();
  const unsigned int numClosest = closestSegments.total;
  const int q = depth & 1;
  const bool packed = !CUTOUT && !PRIMARY && p.packedQueue != 0 && depth > 0; // as in traceKernel
  unsigned int nodeCount = 0, triCount = 0, instCount = 0;
  for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
  {
    const unsigned int slot = p.overflowSlots[k];
    const bool isShadow = !(slot < numClosest);
    // the ray's record in its queue (device_types.h "queue segments"); the hit record stays indexed by the launch's slot
    const unsigned int record = isShadow ? physicalSlot(shadowSegments, p.queueStride, slot - numClosest) : (PRIMARY ? slot : physicalSlot(closestSegments, p.queueStride, slot));
    float4 o, d;
    if (PRIMARY)
    {
      const PrimaryRay pr = primaryRay(p, slot);
      o = make_float4(pr.origin.x, pr.origin.y, pr.origin.z, p.sceneEpsilon);
      d = make_float4(pr.direction.x, pr.direction.y, pr.direction.z, pr.active ? RT_DEFAULT_MAX : -1.0f);
    }
    else
    {
      o = isShadow ? p.shadowOrg[record] : p.rayOrg[q][record];
      d = isShadow ? p.shadowDir[record] : p.rayDir[q][record];
    }
    const unsigned int packedPixel = __float_as_uint(o.w) & TWK_PACKED_PIXEL_MASK;
    if (packed && !isShadow) { o.w = p.sceneEpsilon; d.w = RT_DEFAULT_MAX; }
    float tmin = (PRIMARY && CUTOUT) ? p.hitRecord[slot].x : o.w; // carries the distance of the last ignored cutout candidate, if any
    const unsigned int rayClock = COUNT ? (unsigned int) __builtin_readcyclecounter() : 0u; // time view: this lane's cycles for the re-trace
    TraceResult res;
    for (;;)
    {
      traverse<COUNT>(p, v3(o), v3(d), tmin, d.w, isShadow && !CUTOUT, ldsStack, spill, res, nodeCount, triCount, instCount);
      if (!(CUTOUT && res.instance >= 0 && cutoutIgnoresCandidate(p, res, isShadow, q, record, PRIMARY))) break;
      tmin = res.t;
    }
    if (COUNT && p.pathTime != nullptr)
      atomicAdd(&p.pathTime[isShadow ? p.shadowPixel[record] : (PRIMARY ? slot : (packed ? packedPixel : p.rayPixel[q][record]))], float((unsigned int) __builtin_readcyclecounter() - rayClock));
    if (!isShadow)
    {
      p.hitRecord[slot]   = make_float4(res.t, res.beta, res.gamma, __int_as_float(res.triangleSlot));
      p.hitInstance[slot] = res.instance;
      if (p.firstHit != nullptr && depth == 0)
      {
        const unsigned int pixel = PRIMARY ? slot : p.rayPixel[q][record];
        p.firstHit[pixel] = make_float4(res.t, res.beta, res.gamma, __int_as_float(res.primitive));
        p.firstHitInstance[pixel] = res.instance;
      }
    }
    else if (res.instance < 0)
    {
      const unsigned int sIdx = record;
      const unsigned int pixel = p.shadowPixel[sIdx];
      const float4 c = p.shadowPending[sIdx];
      float4 r = p.pathRadiance[pixel];
      r.x += c.x; r.y += c.y; r.z += c.z;
      p.pathRadiance[pixel] = r;
    }
  }
  if (COUNT)
  {
    // the persistent kernel already counted these rays and its partial visits; add the re-trace's visits
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&p.stats[12], (unsigned long long) count);
    atomicAdd(&p.stats[2], (unsigned long long) nodeCount);
    atomicAdd(&p.stats[3], (unsigned long long) triCount);
    atomicAdd(&p.stats[4], (unsigned long long) instCount);
  }
}

template<bool COUNT, bool CUTOUT, bool TWO_LEVEL, bool W7, bool PRIMARY>
static void launchTraceVariant(const LaunchParams& p, int depth, int gridBlocks, hipStream_t stream)
{
  const int overflowBlocks = gridBlocks < 64 ? gridBlocks : 64; // lanes index the same per-lane spill segments
  hipLaunchKernelGGL((traceKernel<COUNT, CUTOUT, TWO_LEVEL, W7, PRIMARY>), dim3(gridBlocks), dim3(TWK_TRACE_BLOCK), 0, stream, p, depth);
  hipLaunchKernelGGL((traceOverflowKernel<COUNT, CUTOUT, PRIMARY>), dim3(overflowBlocks), dim3(TWK_TRACE_BLOCK), 0, stream, p, depth);
}

template<bool COUNT, bool PRIMARY>
static void launchTraceOpaque(const LaunchParams& p, int depth, int gridBlocks, hipStream_t stream)
{
  if (p.twoLevel)                                                      launchTraceVariant<COUNT, false, true,  false, PRIMARY>(p, depth, gridBlocks, stream);
  else if (p.traceWaves == TWK_TRACE_WAVES7 && !(PRIMARY && TWK_PRIMARY_SIX)) launchTraceVariant<COUNT, false, false, true,  PRIMARY>(p, depth, gridBlocks, stream);
  else                                         launchTraceVariant<COUNT, false, false, false, PRIMARY>(p, depth, gridBlocks, stream);
}

// gridBlocks must be numCUs x p.traceWaves (or a lane's share of it): every block of the persistent kernel resident at once.
// primary: depth 0 of a pass whose generateKernel was skipped.

void launchTrace(const LaunchParams& p, int depth, bool count, bool primary, int gridBlocks, hipStream_t stream)
{
  if (!p.hasCutout)
  {
    if (primary) { if (count) launchTraceOpaque<true, true>(p, depth, gridBlocks, stream);  else launchTraceOpaque<false, true>(p, depth, gridBlocks, stream); }
    else         { if (count) launchTraceOpaque<true, false>(p, depth, gridBlocks, stream); else launchTraceOpaque<false, false>(p, depth, gridBlocks, stream); }
    return;
  }
  if (primary)
  {
    if (p.twoLevel) { if (count) launchTraceVariant<true, true, true,  false, true>(p, depth, gridBlocks, stream); else launchTraceVariant<false, true, true,  false, true>(p, depth, gridBlocks, stream); }
    else            { if (count) launchTraceVariant<true, true, false, false, true>(p, depth, gridBlocks, stream); else launchTraceVariant<false, true, false, false, true>(p, depth, gridBlocks, stream); }
    return;
  }
  if (p.twoLevel) { if (count) launchTraceVariant<true, true, true,  false, false>(p, depth, gridBlocks, stream); else launchTraceVariant<false, true, true,  false, false>(p, depth, gridBlocks, stream); }
  else if (p.traceWaves == TWK_TRACE_WAVES7) { if (count) launchTraceVariant<true, true, false, true, false>(p, depth, gridBlocks, stream); else launchTraceVariant<false, true, false, true, false>(p, depth, gridBlocks, stream); } // the flattened cutout build fits seven blocks since round 4 (71 VGPRs)
  else            { if (count) launchTraceVariant<true, true, false, false, false>(p, depth, gridBlocks, stream); else launchTraceVariant<false, true, false, false, false>(p, depth, gridBlocks, stream); }
}

// ---------------------------------------------------------------------------------------------
// Entry points of the primary rays. All primary rays of a pinhole camera leave one point, and those of a tile of 8 x 8
// launch indices stay inside the pyramid through the tile's corners whatever their jitter: a subtree whose box lies outside
// one of the pyramid's four side planes cannot be hit by any of them. One thread per tile opens the tree from the root —
// the inner entry that is nearest along the tile's axis first, children outside the pyramid dropped — for as long as the
// list fits TWK_ENTRY_REFS references, and leaves them sorted nearest first. traceKernel<.., PRIMARY> starts a ray at its
// tile's list instead of at the root: a ray that looks at a wall starts at the wall's two triangles, one that looks at the
// sky of an open scene at nothing. What a ray can hit is unchanged (only subtrees NO ray of the tile reaches are skipped;
// the boxes tested are the quantised ones the traversal itself tests, the pyramid is widened by half a pixel), the order
// of visits is not part of the result (ties in t go to the smaller instance, primitive).
#ifndef TWK_ENTRY_MAX_LEAVES
#define TWK_ENTRY_MAX_LEAVES 1
#endif
__global__ void __launch_bounds__(64) tileEntryKernel(LaunchParams p, const float4* __restrict__ topTable, int tilesX, int tilesY, int4* __restrict__ out)
{
  const int tile = blockIdx.x * blockDim.x + threadIdx.x;
  if (tile >= tilesX * tilesY) return;
  const int tx = tile % tilesX, ty = tile / tilesX;
  const float* cam = p.camera;
  const V3 P = v3(cam[0], cam[1], cam[2]), U = v3(cam[3], cam[4], cam[5]), V = v3(cam[6], cam[7], cam[8]), W = v3(cam[9], cam[10], cam[11]);
  const float screenX = float(p.resolution[0]), screenY = float(p.resolution[1]);
  // the tile's pixels: its launch indices themselves, or — a device of a tile distribution — where distribute() puts them
  // (the host has checked that a distribution tile is a whole number of entry tiles: one entry tile = one pixel square)
  unsigned int column = (unsigned int) (tx * TWK_ENTRY_TILE);
  if (p.distribution && 1 < p.deviceCount) column = distribute(p, column, (unsigned int) (ty * TWK_ENTRY_TILE));
  if (column >= (unsigned int) p.resolution[0])
  {
    out[2 * (size_t) tile] = make_int4(0, 0, 0, 0); out[2 * (size_t) tile + 1] = make_int4(0, 0, 0, 0); // launch indices without a pixel
    return;
  }
  const float x0 = float(column) - 0.5f, x1 = fminf(float(column + TWK_ENTRY_TILE), screenX) + 0.5f;
  const float y0 = float(ty * TWK_ENTRY_TILE) - 0.5f, y1 = fminf(float(ty * TWK_ENTRY_TILE + TWK_ENTRY_TILE), screenY) + 0.5f;
  auto direction = [&](float fx, float fy) { return U * ((fx / screenX) * 2.0f - 1.0f) + V * ((fy / screenY) * 2.0f - 1.0f) + W; };
  const V3 d00 = direction(x0, y0), d10 = direction(x1, y0), d01 = direction(x0, y1), d11 = direction(x1, y1);
  const V3 axis = normalize(d00 + d10 + d01 + d11);
  // inward normals of the four side planes (all through P): oriented by the opposite corner
  V3 plane[4] = { cross(d00, d10), cross(d10, d11), cross(d11, d01), cross(d01, d00) };
  const V3 opposite[4] = { d11, d01, d00, d10 };
  for (int k = 0; k < 4; ++k)
  {
    if (dot(plane[k], opposite[k]) < 0.0f) plane[k] = -plane[k];
    plane[k] = normalize(plane[k]);
  }
  // box [lo, hi] outside the pyramid? (its corner furthest along a plane's inward normal is still behind that plane, by more
  // than rounding can account for)
  auto outside = [&](const V3& lo, const V3& hi) -> bool
  {
    const float extent = fmaxf(fmaxf(fabsf(lo.x - P.x), fabsf(hi.x - P.x)), fmaxf(fmaxf(fabsf(lo.y - P.y), fabsf(hi.y - P.y)), fmaxf(fabsf(lo.z - P.z), fabsf(hi.z - P.z))));
    const float slack = 1.0e-4f * extent;
    for (int k = 0; k < 4; ++k)
    {
      const V3 n = plane[k];
      const V3 q = v3(n.x > 0.0f ? hi.x : lo.x, n.y > 0.0f ? hi.y : lo.y, n.z > 0.0f ? hi.z : lo.z);
      if (dot(n, q - P) < -slack) return true;
    }
    const V3 q = v3(axis.x > 0.0f ? hi.x : lo.x, axis.y > 0.0f ? hi.y : lo.y, axis.z > 0.0f ? hi.z : lo.z);
    return dot(axis, q - P) < -slack; // entirely behind the camera
  };

  int   ref[TWK_ENTRY_REFS + 4];
  float key[TWK_ENTRY_REFS + 4]; // distance of the box centre along the tile's axis: the order of the list
  bool  closed[TWK_ENTRY_REFS + 4]; // an inner entry whose children did not fit: stays one reference
  int n = 1;
  ref[0] = p.topRoot; key[0] = 0.0f; closed[0] = false;
  if (p.topRoot2 != TWK_BVH_SENTINEL) { ref[1] = p.topRoot2; key[1] = 0.0f; closed[1] = false; n = 2; } // both nodes of an 8-wide root
  for (int round = 0; round < 64; ++round)
  {
    // the nearest inner entry that still fits when opened
    int pick = -1;
    for (int i = 0; i < n; ++i)
      if ((unsigned int) ref[i] < (unsigned int) TWK_BVH_SENTINEL && !closed[i] && (pick < 0 || key[i] < key[pick])) pick = i;
    if (pick < 0) break;
    const int node = ref[pick];
    const float4* w = (node & TWK_NODE_CACHED) ? topTable + 4 * (size_t) (node & 0xff) : p.wideQ + 4 * (size_t) node;
    const float4 n0 = w[0], n1 = w[1], n2 = w[2], n3 = w[3];
    const unsigned int qlx = __float_as_uint(n1.z), qly = __float_as_uint(n1.w), qlz = __float_as_uint(n2.x);
    const unsigned int qhx = __float_as_uint(n2.y), qhy = __float_as_uint(n2.z), qhz = __float_as_uint(n2.w);
    const int childRef[4] = { __float_as_int(n3.x), __float_as_int(n3.y), __float_as_int(n3.z), __float_as_int(n3.w) };
    int   keepRef[4];
    float keepKey[4];
    int m = 0;
    for (int k = 0; k < 4; ++k)
    {
      const float lx = float((qlx >> (8 * k)) & 0xffu), ly = float((qly >> (8 * k)) & 0xffu), lz = float((qlz >> (8 * k)) & 0xffu);
      const float hx = float((qhx >> (8 * k)) & 0xffu), hy = float((qhy >> (8 * k)) & 0xffu), hz = float((qhz >> (8 * k)) & 0xffu);
      if (lx > hx) continue; // unused entry (inverted box)
      // the planes the traversal tests: origin + q * cell, widened by one cell of rounding room
      const V3 lo = v3(__builtin_fmaf(lx - 1.0f, n0.w, n0.x), __builtin_fmaf(ly - 1.0f, n1.x, n0.y), __builtin_fmaf(lz - 1.0f, n1.y, n0.z));
      const V3 hi = v3(__builtin_fmaf(hx + 1.0f, n0.w, n0.x), __builtin_fmaf(hy + 1.0f, n1.x, n0.y), __builtin_fmaf(hz + 1.0f, n1.y, n0.z));
      if (outside(lo, hi)) continue;
      keepRef[m] = childRef[k];
      keepKey[m] = dot(axis, (lo + hi) * 0.5f - P);
      ++m;
    }
    int leaves = 0;
    for (int k = 0; k < m; ++k) leaves += (keepRef[k] < 0) ? 1 : 0;
    // Every entry of the list is visited by every ray of the tile, hit or not; below a node that stays closed only the
    // children a ray's own box test lets through are. Opening a node of several leaves trades one node visit for a triangle
    // test per leaf and ray: not worth it from TWK_ENTRY_MAX_LEAVES leaves on.
    if (n - 1 + m > TWK_ENTRY_REFS || leaves > TWK_ENTRY_MAX_LEAVES) { closed[pick] = true; continue; } // stays one reference; the next nearest is looked at
    ref[pick] = ref[n - 1]; key[pick] = key[n - 1]; closed[pick] = closed[n - 1]; --n;
    for (int k = 0; k < m; ++k) { ref[n] = keepRef[k]; key[n] = keepKey[k]; closed[n] = false; ++n; }
  }
  // nearest first
  for (int i = 1; i < n; ++i)
  {
    const int r = ref[i]; const float k = key[i];
    int j = i - 1;
    while (j >= 0 && key[j] > k) { ref[j + 1] = ref[j]; key[j + 1] = key[j]; --j; }
    ref[j + 1] = r; key[j + 1] = k;
  }
  // n == 0: nothing of the scene in this tile's pyramid — the kernel still wants a reference: an empty list means "start at the root"
  int4 a = make_int4(n, 0, 0, 0), b = make_int4(0, 0, 0, 0);
  if (n > 0) a.y = ref[0];
  if (n > 1) a.z = ref[1];
  if (n > 2) a.w = ref[2];
  if (n > 3) b.x = ref[3];
  if (n > 4) b.y = ref[4];
  if (n > 5) b.z = ref[5];
  if (n > 6) b.w = ref[6];
  out[2 * (size_t) tile] = a; out[2 * (size_t) tile + 1] = b;
}

void launchTileEntries(const LaunchParams& p, const float4* topTable, int tilesX, int tilesY, int4* out, hipStream_t stream)
{
  const int tiles = tilesX * tilesY;
  hipLaunchKernelGGL(tileEntryKernel, dim3((tiles + 63) / 64), dim3(64), 0, stream, p, topTable, tilesX, tilesY, out);
}

void launchTraceQuery(const LaunchParams& p, const float* rays, unsigned int numRays, int anyHit, float* tBetaGamma, int* ids, int gridBlocks, hipStream_t stream)
{
  hipLaunchKernelGGL(traceQueryKernel, dim3(gridBlocks), dim3(TWK_TRACE_BLOCK), 0, stream, p, rays, numRays, anyHit, tBetaGamma, ids);
}

} // namespace twk
