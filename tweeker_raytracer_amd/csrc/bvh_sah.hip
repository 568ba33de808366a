// Binned-SAH top-down topology builder (the "SAH" half of north_star's "LBVH/SAH BVH built on-device"; stands where
// optixAccelBuild stood, src/Device.cpp:1362-1407 GAS / :1456-1486 IAS). It replaces the Morton sort + Karras radix
// tree of bvh_build.hip as the source of the tree TOPOLOGY and primitive ORDER and hands over exactly the arrays
// that builder's refit consumes (left / right / parents / position ranges / primitive id per sorted position), so
// box refit, leaf collapse, wide-node emission and triangle-slot emission are shared by both qualities.
//
// Level-synchronous: every node that still has to be split owns a contiguous range of positions. Per level, for the
// LARGE nodes (more than SAH_SMALL primitives):
//   bounds    centroid bounds per node                       (atomic min / max, wave-aggregated when a wave is in one node)
//   bin       16 bins per axis: primitive count + box        (atomics)
//   select    one thread per node: sweep the 3 x 15 split planes, cost = area(L) n(L) + area(R) n(R); allocate the
//             children (inner-node indices from one counter; a full binary tree over n leaves has n - 1 of them)
//   partition primitives move to their side of the split inside the node's range
// Nodes of at most SAH_SMALL primitives are finished by ONE thread each with an exact sweep over all three axes
// (insertion sort of <= 8 centroids), which removes the bottom levels — most of the nodes — from the level loop.
// The tree goes down to single primitives like the radix tree does; the refit collapses subtrees of <= maxLeaf
// positions into leaves. Nothing here is on the timed path (twk_build).
#include "device_types.h"
#include "bvh_build.h"

#include <algorithm>

namespace twk {

#define SAH_BINS 16
#define SAH_SMALL 8
#define SAH_BIN_WORDS 7 // count + lo.xyz + hi.xyz (ordered-uint floats)

struct SahActive { int node, first, count, pad; };
struct SahSplit  { int axis, bin, leftCount, first, slotL, slotR, pad0, pad1; };

TWK_D unsigned int sahOrdered(float f)
{
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
TWK_D float sahFromOrdered(unsigned int u)
{
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
TWK_D float sahHalfArea(float dx, float dy, float dz) { return dx * dy + dy * dz + dz * dx; }

TWK_D V3 sahCentroid(const float4& lo, const float4& hi) { return v3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)); }

// Bin of a centroid coordinate inside [cmin, cmax]; the SAME expression in the bin and the partition pass.
TWK_D int sahBin(float c, float cmin, float cmax)
{
  const float extent = cmax - cmin;
  if (!(extent > 0.0f)) return 0;
  const int b = (int) ((c - cmin) * ((float) SAH_BINS / extent));
  return min(max(b, 0), SAH_BINS - 1);
}

__global__ void sahInitKernel(int count, int* __restrict__ order, int* __restrict__ slotOf, int large)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  order[i] = i;
  slotOf[i] = large ? 0 : -1;
}

// cb: 6 words per active node (min xyz, max xyz as ordered uints); bins: 3 * SAH_BINS * SAH_BIN_WORDS words; fill: 2 words.
__global__ void sahClearKernel(int numActive, unsigned int* __restrict__ cb, unsigned int* __restrict__ bins, unsigned int* __restrict__ fill)
{
  const int perNode = 6 + 3 * SAH_BINS * SAH_BIN_WORDS + 2;
  const long long total = (long long) numActive * perNode;
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x)
  {
    const int k = (int) (i / perNode), w = (int) (i % perNode);
    if (w < 6) cb[6 * (size_t) k + w] = (w < 3) ? 0xffffffffu : 0u;
    else if (w < 6 + 3 * SAH_BINS * SAH_BIN_WORDS)
    {
      const int b = w - 6, word = b % SAH_BIN_WORDS;
      bins[(size_t) k * 3 * SAH_BINS * SAH_BIN_WORDS + b] = (word == 0) ? 0u : ((word < 4) ? 0xffffffffu : 0u);
    }
    else fill[2 * (size_t) k + (w - 6 - 3 * SAH_BINS * SAH_BIN_WORDS)] = 0u;
  }
}

__global__ void sahBoundsKernel(int count, const int* __restrict__ order, const int* __restrict__ slotOf,
                                const float4* __restrict__ primLo, const float4* __restrict__ primHi, unsigned int* __restrict__ cb)
{
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = (pos < count) ? slotOf[pos] : -1;
  V3 c = v3(0.0f);
  if (k >= 0)
  {
    const int prim = order[pos];
    c = sahCentroid(primLo[prim], primHi[prim]);
  }
  // a wave whose lanes all sit in one node reduces first: the top levels would otherwise send every primitive's six
  // atomics to the same six words
  const int k0 = __builtin_amdgcn_readfirstlane(k);
  if (__ballot(k != k0) == 0ull)
  {
    if (k0 < 0) return;
    float mnx = c.x, mny = c.y, mnz = c.z, mxx = c.x, mxy = c.y, mxz = c.z;
    for (int offset = 32; offset > 0; offset >>= 1)
    {
      mnx = fminf(mnx, __shfl_xor(mnx, offset)); mny = fminf(mny, __shfl_xor(mny, offset)); mnz = fminf(mnz, __shfl_xor(mnz, offset));
      mxx = fmaxf(mxx, __shfl_xor(mxx, offset)); mxy = fmaxf(mxy, __shfl_xor(mxy, offset)); mxz = fmaxf(mxz, __shfl_xor(mxz, offset));
    }
    if ((threadIdx.x & 63) == 0)
    {
      unsigned int* w = cb + 6 * (size_t) k0;
      atomicMin(&w[0], sahOrdered(mnx)); atomicMin(&w[1], sahOrdered(mny)); atomicMin(&w[2], sahOrdered(mnz));
      atomicMax(&w[3], sahOrdered(mxx)); atomicMax(&w[4], sahOrdered(mxy)); atomicMax(&w[5], sahOrdered(mxz));
    }
    return;
  }
  if (k < 0) return;
  unsigned int* w = cb + 6 * (size_t) k;
  atomicMin(&w[0], sahOrdered(c.x)); atomicMin(&w[1], sahOrdered(c.y)); atomicMin(&w[2], sahOrdered(c.z));
  atomicMax(&w[3], sahOrdered(c.x)); atomicMax(&w[4], sahOrdered(c.y)); atomicMax(&w[5], sahOrdered(c.z));
}

__global__ void __launch_bounds__(256) sahBinKernel(int count, const int* __restrict__ order, const int* __restrict__ slotOf,
                             const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                             const unsigned int* __restrict__ cb, unsigned int* __restrict__ bins)
{
  // A block whose 256 positions all lie in ONE node (every block of the top levels) bins into LDS first and sends one
  // atomic per touched word: the top levels would otherwise send every primitive's 21 atomics to the same 336 words.
  __shared__ unsigned int local[3 * SAH_BINS * SAH_BIN_WORDS];
  __shared__ int uniform;
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = (pos < count) ? slotOf[pos] : -1;
  const int kFirst = slotOf[min(blockIdx.x * blockDim.x, (unsigned int) (count - 1))];
  if (threadIdx.x == 0) uniform = 1;
  for (int w = threadIdx.x; w < 3 * SAH_BINS * SAH_BIN_WORDS; w += blockDim.x)
  {
    const int word = w % SAH_BIN_WORDS;
    local[w] = (word == 0) ? 0u : ((word < 4) ? 0xffffffffu : 0u);
  }
  __syncthreads();
  if (pos < count && k != kFirst) uniform = 0;
  __syncthreads();
  const bool useLocal = (uniform != 0) && (kFirst >= 0);
  if (k >= 0)
  {
    const int prim = order[pos];
    const float4 lo = primLo[prim], hi = primHi[prim];
    const V3 c = sahCentroid(lo, hi);
    const unsigned int* w = cb + 6 * (size_t) k;
    const float cc[3] = {c.x, c.y, c.z};
    for (int axis = 0; axis < 3; ++axis)
    {
      const int b = sahBin(cc[axis], sahFromOrdered(w[axis]), sahFromOrdered(w[3 + axis]));
      unsigned int* bin = useLocal ? local + ((size_t) axis * SAH_BINS + b) * SAH_BIN_WORDS
                                   : bins + ((size_t) k * 3 * SAH_BINS + (size_t) axis * SAH_BINS + b) * SAH_BIN_WORDS;
      atomicAdd(&bin[0], 1u);
      atomicMin(&bin[1], sahOrdered(lo.x)); atomicMin(&bin[2], sahOrdered(lo.y)); atomicMin(&bin[3], sahOrdered(lo.z));
      atomicMax(&bin[4], sahOrdered(hi.x)); atomicMax(&bin[5], sahOrdered(hi.y)); atomicMax(&bin[6], sahOrdered(hi.z));
    }
  }
  if (!useLocal) return; // block-uniform: no thread leaves before the barrier below
  __syncthreads();
  unsigned int* nodeBins = bins + (size_t) kFirst * 3 * SAH_BINS * SAH_BIN_WORDS;
  for (int w = threadIdx.x; w < 3 * SAH_BINS * SAH_BIN_WORDS; w += blockDim.x)
  {
    const int word = w % SAH_BIN_WORDS;
    const unsigned int v = local[w];
    if (word == 0) { if (v != 0u) atomicAdd(&nodeBins[w], v); }
    else if (word < 4) { if (v != 0xffffffffu) atomicMin(&nodeBins[w], v); }
    else { if (v != 0u) atomicMax(&nodeBins[w], v); }
  }
}

// Registers a child range: a leaf reference for one primitive, else a fresh inner node that goes to the small list
// (finished by sahSmallKernel) or to the next level's active list. Returns the child reference; slot = active slot or -1.
TWK_D int sahMakeChild(int parent, int first, int count, int* nodeCounter, int* innerParent, int* leafParent, int2* range,
                       SahActive* nextActive, int* nextCount, SahActive* smallList, int* smallCount, int& slot)
{
  slot = -1;
  if (count == 1) { leafParent[first] = parent; return ~first; }
  const int c = atomicAdd(nodeCounter, 1);
  innerParent[c] = parent;
  range[c] = make_int2(first, count);
  SahActive a; a.node = c; a.first = first; a.count = count; a.pad = 0;
  if (count <= SAH_SMALL) smallList[atomicAdd(smallCount, 1)] = a;
  else { slot = atomicAdd(nextCount, 1); nextActive[slot] = a; }
  return c;
}

__global__ void sahSelectKernel(int numActive, const SahActive* __restrict__ active, const unsigned int* __restrict__ bins, int forceMiddle,
                                SahSplit* __restrict__ split, int* nodeCounter, int* __restrict__ left, int* __restrict__ right,
                                int* __restrict__ innerParent, int* __restrict__ leafParent, int2* __restrict__ range,
                                SahActive* __restrict__ nextActive, int* nextCount, SahActive* __restrict__ smallList, int* smallCount)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= numActive) return;
  const SahActive a = active[k];
  const unsigned int* nodeBins = bins + (size_t) k * 3 * SAH_BINS * SAH_BIN_WORDS;

  float bestCost = __uint_as_float(0x7f800000u);
  int bestAxis = -1, bestBin = 0, bestLeft = 0;
  if (!forceMiddle)
  {
    for (int axis = 0; axis < 3; ++axis)
    {
      const unsigned int* b = nodeBins + (size_t) axis * SAH_BINS * SAH_BIN_WORDS;
      // suffix: area and count of bins [s, SAH_BINS)
      float rightArea[SAH_BINS]; int rightCount[SAH_BINS];
      float lx = __uint_as_float(0x7f800000u), ly = lx, lz = lx, hx = -lx, hy = -lx, hz = -lx;
      int n = 0;
      for (int s = SAH_BINS - 1; s >= 1; --s)
      {
        const unsigned int* w = b + (size_t) s * SAH_BIN_WORDS;
        if (w[0] != 0u)
        {
          n += (int) w[0];
          lx = fminf(lx, sahFromOrdered(w[1])); ly = fminf(ly, sahFromOrdered(w[2])); lz = fminf(lz, sahFromOrdered(w[3]));
          hx = fmaxf(hx, sahFromOrdered(w[4])); hy = fmaxf(hy, sahFromOrdered(w[5])); hz = fmaxf(hz, sahFromOrdered(w[6]));
        }
        rightCount[s] = n;
        rightArea[s] = (n > 0) ? sahHalfArea(hx - lx, hy - ly, hz - lz) : 0.0f;
      }
      lx = __uint_as_float(0x7f800000u); ly = lx; lz = lx; hx = -lx; hy = -lx; hz = -lx;
      n = 0;
      for (int s = 1; s < SAH_BINS; ++s)
      {
        const unsigned int* w = b + (size_t) (s - 1) * SAH_BIN_WORDS;
        if (w[0] != 0u)
        {
          n += (int) w[0];
          lx = fminf(lx, sahFromOrdered(w[1])); ly = fminf(ly, sahFromOrdered(w[2])); lz = fminf(lz, sahFromOrdered(w[3]));
          hx = fmaxf(hx, sahFromOrdered(w[4])); hy = fmaxf(hy, sahFromOrdered(w[5])); hz = fmaxf(hz, sahFromOrdered(w[6]));
        }
        if (n == 0 || rightCount[s] == 0) continue;
        const float cost = sahHalfArea(hx - lx, hy - ly, hz - lz) * (float) n + rightArea[s] * (float) rightCount[s];
        if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = s; bestLeft = n; }
      }
    }
  }
  if (bestAxis < 0) { bestBin = -1; bestLeft = a.count / 2; } // all centroids in one bin on every axis (or a forced level): cut the range in the middle

  SahSplit sp;
  sp.axis = bestAxis; sp.bin = bestBin; sp.leftCount = bestLeft; sp.first = a.first; sp.pad0 = sp.pad1 = 0;
  const int l = sahMakeChild(a.node, a.first, bestLeft, nodeCounter, innerParent, leafParent, range, nextActive, nextCount, smallList, smallCount, sp.slotL);
  const int r = sahMakeChild(a.node, a.first + bestLeft, a.count - bestLeft, nodeCounter, innerParent, leafParent, range, nextActive, nextCount, smallList, smallCount, sp.slotR);
  left[a.node] = l; right[a.node] = r;
  split[k] = sp;
}

__global__ void sahPartitionKernel(int count, const int* __restrict__ order, const int* __restrict__ slotOf,
                                   const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                                   const unsigned int* __restrict__ cb, const SahSplit* __restrict__ split, unsigned int* __restrict__ fill,
                                   int* __restrict__ orderNext, int* __restrict__ slotNext)
{
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= count) return;
  const int k = slotOf[pos];
  const int prim = order[pos];
  if (k < 0) { orderNext[pos] = prim; slotNext[pos] = -1; return; } // finished range: stays where it is
  const SahSplit sp = split[k];
  bool toLeft;
  int target;
  if (sp.bin < 0)
  {
    toLeft = (pos - sp.first) < sp.leftCount; // positional cut, nothing moves
    target = pos;
  }
  else
  {
    const V3 c = sahCentroid(primLo[prim], primHi[prim]);
    const float cc = (sp.axis == 0) ? c.x : ((sp.axis == 1) ? c.y : c.z);
    const unsigned int* w = cb + 6 * (size_t) k;
    toLeft = sahBin(cc, sahFromOrdered(w[sp.axis]), sahFromOrdered(w[3 + sp.axis])) < sp.bin;
    // one atomic per wave and side when the whole wave sits in one node (every wave of the top levels), else one per lane
    const unsigned long long active = __ballot(true);
    const int k0 = __builtin_amdgcn_readfirstlane(k);
    unsigned int rank;
    if (__ballot(k != k0) == 0ull)
    {
      const unsigned long long leftMask = __ballot(toLeft), mine = toLeft ? leftMask : (active & ~leftMask);
      const unsigned int lane = threadIdx.x & 63u;
      const int leader = __ffsll((long long) mine) - 1;
      unsigned int base = 0u;
      if ((int) lane == leader) base = atomicAdd(&fill[2 * (size_t) k + (toLeft ? 0 : 1)], (unsigned int) __popcll(mine));
      base = __shfl(base, leader);
      rank = base + (unsigned int) __popcll(mine & ((1ull << lane) - 1ull));
    }
    else rank = atomicAdd(&fill[2 * (size_t) k + (toLeft ? 0 : 1)], 1u);
    target = sp.first + (toLeft ? 0 : sp.leftCount) + (int) rank;
  }
  orderNext[target] = prim;
  slotNext[target] = toLeft ? sp.slotL : sp.slotR;
}

// One thread finishes a node of 2..SAH_SMALL primitives: exact SAH sweep over the three axes at every split.
__global__ void sahSmallKernel(int numSmall, const SahActive* __restrict__ smallList, int* __restrict__ order,
                               const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                               int* nodeCounter, int* __restrict__ left, int* __restrict__ right,
                               int* __restrict__ innerParent, int* __restrict__ leafParent, int2* __restrict__ range)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= numSmall) return;
  int stackNode[SAH_SMALL], stackFirst[SAH_SMALL], stackCount[SAH_SMALL];
  int sp = 0;
  stackNode[0] = smallList[t].node; stackFirst[0] = smallList[t].first; stackCount[0] = smallList[t].count; sp = 1;
  while (sp > 0)
  {
    --sp;
    const int node = stackNode[sp], first = stackFirst[sp], count = stackCount[sp];
    int ids[SAH_SMALL];
    float4 lo[SAH_SMALL], hi[SAH_SMALL];
    for (int i = 0; i < count; ++i) { ids[i] = order[first + i]; lo[i] = primLo[ids[i]]; hi[i] = primHi[ids[i]]; }

    int bestPerm[SAH_SMALL];
    for (int i = 0; i < count; ++i) bestPerm[i] = i;
    int bestLeft = count / 2;
    float bestCost = __uint_as_float(0x7f800000u);
    if (count > 2)
    {
      for (int axis = 0; axis < 3; ++axis)
      {
        int perm[SAH_SMALL]; float key[SAH_SMALL];
        for (int i = 0; i < count; ++i)
        {
          const float c = (axis == 0) ? (lo[i].x + hi[i].x) : ((axis == 1) ? (lo[i].y + hi[i].y) : (lo[i].z + hi[i].z));
          int j = i;
          while (j > 0 && key[j - 1] > c) { key[j] = key[j - 1]; perm[j] = perm[j - 1]; --j; }
          key[j] = c; perm[j] = i;
        }
        float rightArea[SAH_SMALL];
        float lx = __uint_as_float(0x7f800000u), ly = lx, lz = lx, hx = -lx, hy = -lx, hz = -lx;
        for (int i = count - 1; i >= 1; --i)
        {
          const int p = perm[i];
          lx = fminf(lx, lo[p].x); ly = fminf(ly, lo[p].y); lz = fminf(lz, lo[p].z);
          hx = fmaxf(hx, hi[p].x); hy = fmaxf(hy, hi[p].y); hz = fmaxf(hz, hi[p].z);
          rightArea[i] = sahHalfArea(hx - lx, hy - ly, hz - lz);
        }
        lx = __uint_as_float(0x7f800000u); ly = lx; lz = lx; hx = -lx; hy = -lx; hz = -lx;
        for (int i = 1; i < count; ++i)
        {
          const int p = perm[i - 1];
          lx = fminf(lx, lo[p].x); ly = fminf(ly, lo[p].y); lz = fminf(lz, lo[p].z);
          hx = fmaxf(hx, hi[p].x); hy = fmaxf(hy, hi[p].y); hz = fmaxf(hz, hi[p].z);
          const float cost = sahHalfArea(hx - lx, hy - ly, hz - lz) * (float) i + rightArea[i] * (float) (count - i);
          if (cost < bestCost)
          {
            bestCost = cost; bestLeft = i;
            for (int q = 0; q < count; ++q) bestPerm[q] = perm[q];
          }
        }
      }
    }
    for (int i = 0; i < count; ++i) order[first + i] = ids[bestPerm[i]];

    int refs[2];
    const int childFirst[2] = {first, first + bestLeft}, childCount[2] = {bestLeft, count - bestLeft};
    for (int side = 0; side < 2; ++side)
    {
      if (childCount[side] == 1) { leafParent[childFirst[side]] = node; refs[side] = ~childFirst[side]; }
      else
      {
        const int c = atomicAdd(nodeCounter, 1);
        innerParent[c] = node;
        range[c] = make_int2(childFirst[side], childCount[side]);
        refs[side] = c;
        stackNode[sp] = c; stackFirst[sp] = childFirst[side]; stackCount[sp] = childCount[side]; ++sp;
      }
    }
    left[node] = refs[0]; right[node] = refs[1];
  }
}

__global__ void sahKeysKernel(int count, const int* __restrict__ order, unsigned long long* __restrict__ keys)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) keys[i] = (unsigned long long) (unsigned int) order[i];
}

__global__ void sahRootKernel(int count, int* nodeCounter, int* innerParent, int2* range, SahActive* active, SahActive* smallList, int* counts)
{
  // counts: [0] active count of level 0, [1] next level's count, [2] small count
  innerParent[0] = -1;
  range[0] = make_int2(0, count);
  *nodeCounter = 1;
  SahActive a; a.node = 0; a.first = 0; a.count = count; a.pad = 0;
  counts[0] = 0; counts[1] = 0; counts[2] = 0;
  if (count <= SAH_SMALL) { smallList[0] = a; counts[2] = 1; }
  else { active[0] = a; counts[0] = 1; }
}

// SAH cost of the tree the refit produced (measurement): sum over the nodes that survive the leaf collapse of
// half-area(node) / half-area(root), inner nodes and leaf primitives separately (cost = Cn * inner + Ct * leaf).
__global__ void sahCostKernel(int count, int maxLeaf, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ innerParent,
                              const int2* __restrict__ range, const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi,
                              const unsigned long long* __restrict__ keys, const float4* __restrict__ primLo, const float4* __restrict__ primHi, double* out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count - 1) return;
  const float rootArea = sahHalfArea(nodeHi[0].x - nodeLo[0].x, nodeHi[0].y - nodeLo[0].y, nodeHi[0].z - nodeLo[0].z);
  if (!(rootArea > 0.0f)) return;
  const int2 rg = range[i];
  const int parent = innerParent[i];
  const bool collapsed = (rg.y <= maxLeaf);
  if (collapsed && parent >= 0 && range[parent].y <= maxLeaf) return; // inside a collapsed subtree: not part of the final tree
  const float area = sahHalfArea(nodeHi[i].x - nodeLo[i].x, nodeHi[i].y - nodeLo[i].y, nodeHi[i].z - nodeLo[i].z) / rootArea;
  if (collapsed && parent >= 0) { atomicAdd(&out[1], (double) area * rg.y); return; } // a leaf of rg.y triangles
  atomicAdd(&out[0], (double) area);
  const int refs[2] = {left[i], right[i]};
  for (int side = 0; side < 2; ++side)
  {
    if (refs[side] >= 0) continue;
    const unsigned int prim = (unsigned int) (keys[~refs[side]] & 0xffffffffull); // a single-primitive leaf right below an inner node
    const float4 lo = primLo[prim], hi = primHi[prim];
    atomicAdd(&out[1], (double) (sahHalfArea(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z) / rootArea));
  }
}

#define SAH_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

hipError_t BvhBuilder::reserveSah(int count)
{
  if (count <= m_sahCapacity) return hipSuccess;
  releaseSah();
  const size_t n = (size_t) count;
  const size_t maxLarge = n / (SAH_SMALL + 1) + 64, maxSmall = n / 2 + 64;
  SAH_CHECK(hipMalloc(&m_sahOrder[0], sizeof(int) * n));  SAH_CHECK(hipMalloc(&m_sahOrder[1], sizeof(int) * n));
  SAH_CHECK(hipMalloc(&m_sahSlot[0], sizeof(int) * n));   SAH_CHECK(hipMalloc(&m_sahSlot[1], sizeof(int) * n));
  SAH_CHECK(hipMalloc(&m_sahActive[0], sizeof(SahActive) * maxLarge)); SAH_CHECK(hipMalloc(&m_sahActive[1], sizeof(SahActive) * maxLarge));
  SAH_CHECK(hipMalloc(&m_sahSmall, sizeof(SahActive) * maxSmall));
  SAH_CHECK(hipMalloc(&m_sahSplit, sizeof(SahSplit) * maxLarge));
  SAH_CHECK(hipMalloc(&m_sahCb, sizeof(unsigned int) * 6 * maxLarge));
  SAH_CHECK(hipMalloc(&m_sahBins, sizeof(unsigned int) * 3 * SAH_BINS * SAH_BIN_WORDS * maxLarge));
  SAH_CHECK(hipMalloc(&m_sahFill, sizeof(unsigned int) * 2 * maxLarge));
  SAH_CHECK(hipMalloc(&m_sahCounters, sizeof(int) * 8));
  SAH_CHECK(hipMalloc(&m_sahCost, sizeof(double) * 2));
  m_sahCapacity = count;
  return hipSuccess;
}

void BvhBuilder::releaseSah()
{
  void* p[] = { m_sahOrder[0], m_sahOrder[1], m_sahSlot[0], m_sahSlot[1], m_sahActive[0], m_sahActive[1], m_sahSmall, m_sahSplit, m_sahCb, m_sahBins, m_sahFill, m_sahCounters, m_sahCost };
  for (void* q : p) if (q) (void) hipFree(q);
  m_sahOrder[0] = m_sahOrder[1] = m_sahSlot[0] = m_sahSlot[1] = nullptr;
  m_sahActive[0] = m_sahActive[1] = nullptr; m_sahSmall = nullptr; m_sahSplit = nullptr;
  m_sahCb = m_sahBins = m_sahFill = nullptr; m_sahCounters = nullptr; m_sahCost = nullptr;
  m_sahCapacity = 0;
}

// Topology + order for `count` >= 2 primitives whose boxes are in m_primLo / m_primHi: fills m_keysOut, m_left, m_right,
// m_innerParent, m_leafParent, m_range exactly as the Morton / radix-tree path does.
hipError_t BvhBuilder::buildSahTopology(hipStream_t stream, int count)
{
  SAH_CHECK(reserveSah(count));
  const int block = 256, grid = (count + block - 1) / block;
  SahActive* active[2] = { static_cast<SahActive*>(m_sahActive[0]), static_cast<SahActive*>(m_sahActive[1]) };
  SahActive* smallList = static_cast<SahActive*>(m_sahSmall);
  SahSplit* split = static_cast<SahSplit*>(m_sahSplit);
  int* nodeCounter = m_sahCounters + 4;
  // counters: [0] current level's active count, [1] next level's, [2] small nodes, [4] inner nodes allocated
  hipLaunchKernelGGL(sahRootKernel, dim3(1), dim3(1), 0, stream, count, nodeCounter, m_innerParent, m_range, active[0], smallList, m_sahCounters);
  hipLaunchKernelGGL(sahInitKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[0], m_sahSlot[0], count > SAH_SMALL ? 1 : 0);
  int cur = 0;
  int numActive = (count > SAH_SMALL) ? 1 : 0;
  for (int level = 0; numActive > 0 && level < 40 + 34; ++level)
  {
    const int next = cur ^ 1;
    // A degenerate input that SAH keeps peeling one primitive off must not grow a tree deeper than the traversal stacks
    // (trace_device.h: 20 LDS + 72 HBM entries): from level 40 on ranges are halved by position, which ends every range
    // within log2(count) further levels. (Scenes of millions of triangles finish their SAH levels in 25-35.) twk_build
    // measures the height that results and refuses a scene whose top + bottom height exceeds the stack capacity.
    const int forceMiddle = (level >= 40) ? 1 : 0;
    const long long clearWords = (long long) numActive * (6 + 3 * SAH_BINS * SAH_BIN_WORDS + 2); // 344 words per node: beyond int at ~6.2 M active nodes
    hipLaunchKernelGGL(sahClearKernel, dim3((unsigned int) std::min<long long>(4096, (clearWords + 255) / 256)), dim3(256), 0, stream, numActive, m_sahCb, m_sahBins, m_sahFill);
    hipLaunchKernelGGL(sahBoundsKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[cur], m_sahSlot[cur], m_primLo, m_primHi, m_sahCb);
    hipLaunchKernelGGL(sahBinKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[cur], m_sahSlot[cur], m_primLo, m_primHi, m_sahCb, m_sahBins);
    hipLaunchKernelGGL(sahSelectKernel, dim3((numActive + 63) / 64), dim3(64), 0, stream, numActive, active[cur], m_sahBins, forceMiddle, split, nodeCounter,
                       m_left, m_right, m_innerParent, m_leafParent, m_range, active[next], m_sahCounters + 1, smallList, m_sahCounters + 2);
    hipLaunchKernelGGL(sahPartitionKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[cur], m_sahSlot[cur], m_primLo, m_primHi, m_sahCb, split, m_sahFill,
                       m_sahOrder[next], m_sahSlot[next]);
    SAH_CHECK(hipGetLastError());
    int nextCount = 0;
    SAH_CHECK(hipMemcpyAsync(&nextCount, m_sahCounters + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
    SAH_CHECK(hipMemsetAsync(m_sahCounters + 1, 0, sizeof(int), stream));
    SAH_CHECK(hipStreamSynchronize(stream));
    numActive = nextCount;
    cur = next;
  }
  if (numActive > 0) return hipErrorUnknown; // cannot happen: forced halving ends every range within 32 further levels
  int numSmall = 0;
  SAH_CHECK(hipMemcpyAsync(&numSmall, m_sahCounters + 2, sizeof(int), hipMemcpyDeviceToHost, stream));
  SAH_CHECK(hipStreamSynchronize(stream));
  if (numSmall > 0)
    hipLaunchKernelGGL(sahSmallKernel, dim3((numSmall + 63) / 64), dim3(64), 0, stream, numSmall, smallList, m_sahOrder[cur], m_primLo, m_primHi,
                       nodeCounter, m_left, m_right, m_innerParent, m_leafParent, m_range);
  hipLaunchKernelGGL(sahKeysKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[cur], m_keysOut);
  return hipGetLastError();
}

// After the refit: SAH cost terms of the tree just built (count >= 2), added to the builder's totals.
hipError_t BvhBuilder::accumulateSahCost(hipStream_t stream, int count)
{
  if (count < 2) return hipSuccess;
  SAH_CHECK(reserveSah(1));
  SAH_CHECK(hipMemsetAsync(m_sahCost, 0, sizeof(double) * 2, stream));
  hipLaunchKernelGGL(sahCostKernel, dim3((count + 255) / 256), dim3(256), 0, stream, count, m_maxLeaf, m_left, m_right, m_innerParent, m_range, m_nodeLo, m_nodeHi,
                     m_keysOut, m_primLo, m_primHi, m_sahCost);
  double h[2] = {0.0, 0.0};
  SAH_CHECK(hipMemcpyAsync(h, m_sahCost, sizeof(h), hipMemcpyDeviceToHost, stream));
  SAH_CHECK(hipStreamSynchronize(stream));
  m_lastSahInner = h[0]; m_lastSahLeaf = h[1];
  return hipSuccess;
}

} // namespace twk
