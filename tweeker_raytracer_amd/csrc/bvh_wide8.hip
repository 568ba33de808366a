// Compressed 8-ary nodes over the binary BVH of a flattened scene (device_types.h "compressed 8-ary node"; stands where
// optixAccelBuild stood, src/Device.cpp:1362-1407,1456-1486 — the traversal structure is this build's own, OptiX's is closed).
//
// Collapse, top-down and level-synchronous on the device: a wide node starts as the two children of a binary node and opens
// its inner entry of largest surface area until it holds eight entries or only leaves (Wald et al. 2008). Per level:
// count the inner children and leaf triangles of every node, exclusive scan (rocPRIM), then emit — children get consecutive
// node indices in slot order (a node's children are ONE reference: childBase + a mask), leaf triangles consecutive triangle
// slots in slot order (the triangle arrays are permuted afterwards, permuteSlotsKernel). No atomics: the same binary tree
// gives the same nodes byte for byte. Levels are numbered one after the other, so the array is in breadth-first order and
// its first nodes are the top of the tree (cached in LDS by the kernel).
//
// Slot assignment (Ylitie, Karras, Laine 2017, section 3.2 "octant-based traversal order"): entry c goes to the slot s that
// maximises dot(centroid_c - centre, d_s), d_s = ((-1)^bit0, (-1)^bit1, (-1)^bit2), greedily over all pairs.
#include "device_types.h"
#include "bvh_build.h"

#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

namespace twk {

namespace {

struct Entry8
{
  float4 lo, hi;
  int    ref;
};

TWK_D float halfArea8(const float4& lo, const float4& hi)
{
  const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  return dx * dy + dy * dz + dz * dx;
}

// The two children of binary node `node` (device_types.h BvhNode); a child that can never be hit (singleLeafKernel's second
// child: an infinite box) is left out.
TWK_D int childrenOf(const BvhNode* __restrict__ nodes, int node, Entry8* out)
{
  const float4* c = reinterpret_cast<const float4*>(nodes + node);
  const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
  const float inf = __uint_as_float(0x7f800000u);
  int n = 0;
  if (!(c0.x == inf)) { out[n].lo = make_float4(c0.x, c0.y, c0.z, 0.0f); out[n].hi = make_float4(c0.w, c1.x, c1.y, 0.0f); out[n].ref = __float_as_int(c3.x); ++n; }
  if (!(c1.z == inf)) { out[n].lo = make_float4(c1.z, c1.w, c2.x, 0.0f); out[n].hi = make_float4(c2.y, c2.z, c2.w, 0.0f); out[n].ref = __float_as_int(c3.y); ++n; }
  return n;
}

// The entries of the wide node that stands for binary node `node`: deterministic, so the counting and the emitting pass see
// the same list.
TWK_D int gatherEntries(const BvhNode* __restrict__ nodes, int node, Entry8* e)
{
  int n = childrenOf(nodes, node, e);
  for (;;)
  {
    if (n >= 8) break;
    int pick = -1; float pickArea = -1.0f;
    for (int k = 0; k < n; ++k)
    {
      if (e[k].ref < 0) continue;
      const float a = halfArea8(e[k].lo, e[k].hi);
      if (a > pickArea) { pickArea = a; pick = k; }
    }
    if (pick < 0) break;
    Entry8 c[2];
    const int m = childrenOf(nodes, e[pick].ref, c);
    if (m == 0) { e[pick] = e[n - 1]; --n; continue; } // cannot happen in a built tree
    e[pick] = c[0];
    if (m > 1) { e[n] = c[1]; ++n; }
  }
  return n;
}

TWK_D int leafTriangles(int ref) { return (((~ref) >> 28) & 3) + 1; }

__global__ void wide8CountKernel(const BvhNode* __restrict__ nodes, const int* __restrict__ queue, int count, unsigned long long* __restrict__ counts)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Entry8 e[8];
  const int n = gatherEntries(nodes, queue[i], e);
  unsigned int inner = 0, triangles = 0;
  for (int k = 0; k < n; ++k) { if (e[k].ref >= 0) ++inner; else triangles += (unsigned int) leafTriangles(e[k].ref); }
  counts[i] = ((unsigned long long) inner << 32) | triangles;
}

__global__ void wide8TotalKernel(const unsigned long long* __restrict__ counts, const unsigned long long* __restrict__ offsets, int count, unsigned long long* total)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) total[0] = offsets[count - 1] + counts[count - 1];
}

// queue[start + i] = the binary node wide node start + i stands for; the children of this level's nodes are numbered from
// nextStart, their leaf triangles from triStart, both in the order of the exclusive scan `offsets` (inner << 32 | triangles).
__global__ void wide8EmitKernel(const BvhNode* __restrict__ nodes, int* __restrict__ queue, int start, int count, int nextStart, int triStart,
                                const unsigned long long* __restrict__ offsets, float4* __restrict__ out, int* __restrict__ slotMap)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Entry8 e[8];
  const int n = gatherEntries(nodes, queue[start + i], e);
  const float inf = __uint_as_float(0x7f800000u);

  float org[3] = {inf, inf, inf}, top[3] = {-inf, -inf, -inf};
  for (int k = 0; k < n; ++k)
  {
    org[0] = fminf(org[0], e[k].lo.x); org[1] = fminf(org[1], e[k].lo.y); org[2] = fminf(org[2], e[k].lo.z);
    top[0] = fmaxf(top[0], e[k].hi.x); top[1] = fmaxf(top[1], e[k].hi.y); top[2] = fmaxf(top[2], e[k].hi.z);
  }
  if (n == 0) { org[0] = org[1] = org[2] = 0.0f; top[0] = top[1] = top[2] = 0.0f; }

  // slot of every entry: greedily the (entry, slot) pair with the largest dot(centroid - centre, diagonal of the slot)
  int slotOf[8], entryAt[8];
  for (int k = 0; k < 8; ++k) { slotOf[k] = -1; entryAt[k] = -1; }
  {
    const float cx = 0.5f * (org[0] + top[0]), cy = 0.5f * (org[1] + top[1]), cz = 0.5f * (org[2] + top[2]);
    for (int round = 0; round < n; ++round)
    {
      int bestEntry = -1, bestSlot = -1; float best = -inf;
      for (int k = 0; k < n; ++k)
      {
        if (slotOf[k] >= 0) continue;
        const float dx = 0.5f * (e[k].lo.x + e[k].hi.x) - cx, dy = 0.5f * (e[k].lo.y + e[k].hi.y) - cy, dz = 0.5f * (e[k].lo.z + e[k].hi.z) - cz;
        for (int s = 0; s < 8; ++s)
        {
          if (entryAt[s] >= 0) continue;
          const float cost = ((s & 1) ? -dx : dx) + ((s & 2) ? -dy : dy) + ((s & 4) ? -dz : dz);
          if (cost > best) { best = cost; bestEntry = k; bestSlot = s; }
        }
      }
      slotOf[bestEntry] = bestSlot; entryAt[bestSlot] = bestEntry;
    }
  }

  // quantisation grid: per axis the smallest power of two with extent / cell < 254 (bvh_build.hip quantizeWideKernel)
  float cell[3];
  unsigned int expo[3];
  for (int c = 0; c < 3; ++c)
  {
    const float extent = fmaxf(top[c] - org[c], 0.0f);
    int ex = -125;
    if (extent > 0.0f && extent < inf) { frexpf(extent * (1.0f / 254.0f), &ex); ex = max(-125, min(126, ex)); }
    cell[c] = ldexpf(1.0f, ex);
    expo[c] = (unsigned int) ex & 0xffu; // a signed byte
  }

  const unsigned long long offset = offsets[i];
  const int childBase = nextStart + (int) (offset >> 32);
  const int triBase   = triStart + (int) (offset & 0xffffffffull);
  unsigned int imask = 0, metaLo = 0, metaHi = 0;
  unsigned int qlo[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}}, qhi[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
  int innerRank = 0, triOffset = 0;
  for (int s = 0; s < 8; ++s)
  {
    const int word = s >> 2, shift = 8 * (s & 3);
    const int k = entryAt[s];
    if (k < 0)
    {
      for (int c = 0; c < 3; ++c) qlo[c][word] |= 255u << shift; // empty slot: inverted box, never entered
      continue;
    }
    const float lo[3] = {e[k].lo.x, e[k].lo.y, e[k].lo.z}, hi[3] = {e[k].hi.x, e[k].hi.y, e[k].hi.z};
    for (int c = 0; c < 3; ++c)
    {
      int ql = (int) fminf(fmaxf(floorf((lo[c] - org[c]) / cell[c]), 0.0f), 255.0f);
      int qh = (int) fminf(fmaxf(ceilf((hi[c] - org[c]) / cell[c]), 0.0f), 255.0f);
      // against the float expression the traversal evaluates: origin + q * cell
      if (ql > 0 && __builtin_fmaf((float) ql, cell[c], org[c]) > lo[c]) --ql;
      if (qh < 255 && __builtin_fmaf((float) qh, cell[c], org[c]) < hi[c]) ++qh;
      qlo[c][word] |= (unsigned int) ql << shift;
      qhi[c][word] |= (unsigned int) qh << shift;
    }
    if (e[k].ref >= 0)
    {
      imask |= 1u << s;
      queue[childBase + innerRank] = e[k].ref;
      ++innerRank;
    }
    else
    {
      const int payload = ~e[k].ref;
      const int first = payload & 0x0fffffff, tris = ((payload >> 28) & 3) + 1;
      const unsigned int meta = (unsigned int) triOffset | ((unsigned int) (tris - 1) << 5);
      if (word == 0) metaLo |= meta << shift; else metaHi |= meta << shift;
      for (int j = 0; j < tris; ++j) slotMap[first + j] = triBase + triOffset + j;
      triOffset += tris;
    }
  }
  float4* o = out + TWK_WIDE8_FLOAT4 * (size_t) (start + i);
  o[0] = make_float4(org[0], org[1], org[2], __uint_as_float(expo[0] | (expo[1] << 8) | (expo[2] << 16) | (imask << 24)));
  o[1] = make_float4(__int_as_float(childBase), __int_as_float(triBase), __uint_as_float(metaLo), __uint_as_float(metaHi));
  o[2] = make_float4(__uint_as_float(qlo[0][0]), __uint_as_float(qlo[0][1]), __uint_as_float(qlo[1][0]), __uint_as_float(qlo[1][1]));
  o[3] = make_float4(__uint_as_float(qlo[2][0]), __uint_as_float(qlo[2][1]), __uint_as_float(qhi[0][0]), __uint_as_float(qhi[0][1]));
  o[4] = make_float4(__uint_as_float(qhi[1][0]), __uint_as_float(qhi[1][1]), __uint_as_float(qhi[2][0]), __uint_as_float(qhi[2][1]));
}

// Triangle slots and shading records into the order the 8-ary nodes address them in.
__global__ void permuteSlotsKernel(const int* __restrict__ slotMap, int count, const float4* __restrict__ triangles, const float4* __restrict__ shade,
                                   float4* __restrict__ outTriangles, float4* __restrict__ outShade)
{
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= count) return;
  const int to = slotMap[slot];
  for (int k = 0; k < 3; ++k) outTriangles[3 * (size_t) to + k] = triangles[3 * (size_t) slot + k];
  for (int k = 0; k < TWK_SHADE_RECORD; ++k) outShade[TWK_SHADE_RECORD * (size_t) to + k] = shade[TWK_SHADE_RECORD * (size_t) slot + k];
}

TWK_D int remapLeaf(int ref, const int* __restrict__ slotMap)
{
  if (ref >= 0 || ref == TWK_BVH_SENTINEL) return ref;
  const int payload = ~ref;
  if (!(payload & TWK_LEAF_WORLD)) return ref; // an instance reference, or the never-hit child of a one-leaf tree
  return ~((payload & ~0x0fffffff) | slotMap[payload & 0x0fffffff]);
}

// The leaf references of the binary nodes (single-ray traversal) and of the quantised 4-ary nodes follow the triangles.
__global__ void remapLeafRefsKernel(BvhNode* __restrict__ nodes, int numNodes, float4* __restrict__ wideQ, int numWide, const int* __restrict__ slotMap)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < numNodes)
  {
    float4* p = reinterpret_cast<float4*>(nodes + i) + 3;
    float4 r = *p;
    r.x = __int_as_float(remapLeaf(__float_as_int(r.x), slotMap));
    r.y = __int_as_float(remapLeaf(__float_as_int(r.y), slotMap));
    *p = r;
  }
  if (i < numWide)
  {
    float4 r = wideQ[4 * (size_t) i + 3];
    r.x = __int_as_float(remapLeaf(__float_as_int(r.x), slotMap));
    r.y = __int_as_float(remapLeaf(__float_as_int(r.y), slotMap));
    r.z = __int_as_float(remapLeaf(__float_as_int(r.z), slotMap));
    r.w = __int_as_float(remapLeaf(__float_as_int(r.w), slotMap));
    wideQ[4 * (size_t) i + 3] = r;
  }
}

} // namespace

#define W8_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)

// nodes: the scene's binary nodes (absolute references), root: the binary node the scene starts at, numTriangles: its
// triangle slots (all of them reachable from root: a flattened scene). On success *outNodes (hipMalloc, 5 float4 per node)
// holds *outCount nodes in breadth-first order, *outSlotMap (hipMalloc, numTriangles ints) the new slot of every old one,
// *outLevels the depth of the 8-ary tree.
hipError_t buildWide8(hipStream_t stream, const BvhNode* nodes, int numBinaryNodes, int root, int numTriangles,
                      float4** outNodes, int* outCount, int** outSlotMap, int* outLevels)
{
  *outNodes = nullptr; *outCount = 0; *outSlotMap = nullptr; *outLevels = 0;
  const size_t capacity = (size_t) numBinaryNodes + 1; // every wide node stands for at least one binary node of its own
  float4* wide = nullptr; int* queue = nullptr; int* slotMap = nullptr;
  unsigned long long *counts = nullptr, *offsets = nullptr, *total = nullptr; void* scanTemp = nullptr;
  auto cleanup = [&]() {
    void* p[] = { wide, queue, slotMap, counts, offsets, total, scanTemp };
    for (void* q : p) if (q) (void) hipFree(q);
  };
  W8_CHECK(hipMalloc(&wide, sizeof(float4) * TWK_WIDE8_FLOAT4 * capacity));
  W8_CHECK(hipMalloc(&queue, sizeof(int) * capacity));
  W8_CHECK(hipMalloc(&slotMap, sizeof(int) * (size_t) (numTriangles > 0 ? numTriangles : 1)));
  W8_CHECK(hipMalloc(&counts, sizeof(unsigned long long) * capacity));
  W8_CHECK(hipMalloc(&offsets, sizeof(unsigned long long) * capacity));
  W8_CHECK(hipMalloc(&total, sizeof(unsigned long long)));
  size_t scanBytes = 0;
  W8_CHECK(rocprim::exclusive_scan(nullptr, scanBytes, counts, offsets, 0ull, capacity, rocprim::plus<unsigned long long>(), stream));
  W8_CHECK(hipMalloc(&scanTemp, scanBytes > 0 ? scanBytes : 16));
  W8_CHECK(hipMemsetAsync(slotMap, 0xff, sizeof(int) * (size_t) (numTriangles > 0 ? numTriangles : 1), stream)); // -1: a slot no node reached (checked by the caller's count)
  W8_CHECK(hipMemcpyAsync(queue, &root, sizeof(int), hipMemcpyHostToDevice, stream));

  int start = 0, count = 1, triStart = 0, levels = 0;
  while (count > 0)
  {
    if ((size_t) start + (size_t) count > capacity) { cleanup(); return hipErrorInvalidValue; }
    const int grid = (count + 127) / 128;
    hipLaunchKernelGGL(wide8CountKernel, dim3(grid), dim3(128), 0, stream, nodes, queue + start, count, counts);
    size_t bytes = scanBytes;
    W8_CHECK(rocprim::exclusive_scan(scanTemp, bytes, counts, offsets, 0ull, (size_t) count, rocprim::plus<unsigned long long>(), stream));
    hipLaunchKernelGGL(wide8TotalKernel, dim3(1), dim3(1), 0, stream, counts, offsets, count, total);
    unsigned long long sum = 0;
    W8_CHECK(hipMemcpyAsync(&sum, total, sizeof(sum), hipMemcpyDeviceToHost, stream));
    W8_CHECK(hipStreamSynchronize(stream));
    const int inner = (int) (sum >> 32), triangles = (int) (sum & 0xffffffffull);
    if ((size_t) start + (size_t) count + (size_t) inner > capacity || triStart + triangles > numTriangles) { cleanup(); return hipErrorInvalidValue; }
    hipLaunchKernelGGL(wide8EmitKernel, dim3(grid), dim3(128), 0, stream, nodes, queue, start, count, start + count, triStart, offsets, wide, slotMap);
    W8_CHECK(hipGetLastError());
    start += count; count = inner; triStart += triangles; ++levels;
  }
  if (triStart != numTriangles) { cleanup(); return hipErrorInvalidValue; } // a triangle slot the tree does not reach: not a flattened scene
  // the nodes at their exact size
  float4* exact = nullptr;
  W8_CHECK(hipMalloc(&exact, sizeof(float4) * TWK_WIDE8_FLOAT4 * (size_t) start));
  hipError_t e = hipMemcpyAsync(exact, wide, sizeof(float4) * TWK_WIDE8_FLOAT4 * (size_t) start, hipMemcpyDeviceToDevice, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { (void) hipFree(exact); cleanup(); return e; }
  *outNodes = exact; *outCount = start; *outSlotMap = slotMap; *outLevels = levels;
  slotMap = nullptr; // handed over
  cleanup();
  return hipSuccess;
}

void launchPermuteSlots(const int* slotMap, int count, const float4* triangles, const float4* shade, float4* outTriangles, float4* outShade, hipStream_t stream)
{
  if (count > 0) hipLaunchKernelGGL(permuteSlotsKernel, dim3((count + 255) / 256), dim3(256), 0, stream, slotMap, count, triangles, shade, outTriangles, outShade);
}

void launchRemapLeafRefs(BvhNode* nodes, int numNodes, float4* wideQ, int numWide, const int* slotMap, hipStream_t stream)
{
  const int n = numNodes > numWide ? numNodes : numWide;
  if (n > 0) hipLaunchKernelGGL(remapLeafRefsKernel, dim3((n + 255) / 256), dim3(256), 0, stream, nodes, numNodes, wideQ, numWide, slotMap);
}

} // namespace twk
