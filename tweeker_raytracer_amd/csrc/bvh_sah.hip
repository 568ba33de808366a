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
//   select    one thread per node: sweep the 3 x 15 split planes, cost = area(L) n(L) + area(R) n(R), and say what each
//             child will be (leaf / small / large)
//   assign    exclusive scan of the per-node child counts, then one thread per node gives its children their inner-node
//             indices (a full binary tree over n leaves has n - 1 of them) and their slots in the next level's lists
//   partition a STABLE partition inside the node's range: exclusive scan of the "goes left" flags over all positions,
//             rank = prefix(position) - prefix(first position of the node)
// No atomic decides an index or a position (round 2 ranked primitives with atomics on a fill counter and took node
// indices from an atomic counter: topology, slot order, visit counts and the LDS top-of-tree cache then differed from
// run to run — the hit records never did): the same input gives the same tree, bit for bit (test_gpu_edge_cases.py).
// Nodes of at most SAH_SMALL primitives are finished by ONE thread each with an exact sweep over all three axes
// (insertion sort of <= 8 centroids), which removes the bottom levels — most of the nodes — from the level loop.
// The tree goes down to single primitives like the radix tree does; the refit collapses subtrees of <= maxLeaf
// positions into leaves. Nothing here is on the timed path (twk_build).
#include "device_types.h"
#include "bvh_build.h"

#include <algorithm>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace twk {

#define SAH_BINS 16
#define SAH_SMALL 8
#define SAH_BIN_WORDS 7 // count + lo.xyz + hi.xyz (ordered-uint floats)

struct SahActive { int node, first, count, pad; };
struct SahSplit  { int axis, bin, leftCount, first, slotL, slotR, pad0, pad1; };
// children a split creates: inner nodes, entries of the next level's active list (large), entries of the small list
struct SahCounts { int inner, large, small, pad; };
struct SahCountsPlus { TWK_HD SahCounts operator()(const SahCounts& a, const SahCounts& b) const { SahCounts r; r.inner = a.inner + b.inner; r.large = a.large + b.large; r.small = a.small + b.small; r.pad = 0; return r; } };

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

// cb: 6 words per active node (min xyz, max xyz as ordered uints); bins: 3 * SAH_BINS * SAH_BIN_WORDS words.
__global__ void sahClearKernel(int numActive, unsigned int* __restrict__ cb, unsigned int* __restrict__ bins)
{
  const int perNode = 6 + 3 * SAH_BINS * SAH_BIN_WORDS;
  const long long total = (long long) numActive * perNode;
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x)
  {
    const int k = (int) (i / perNode), w = (int) (i % perNode);
    if (w < 6) cb[6 * (size_t) k + w] = (w < 3) ? 0xffffffffu : 0u;
    else
    {
      const int b = w - 6, word = b % SAH_BIN_WORDS;
      bins[(size_t) k * 3 * SAH_BINS * SAH_BIN_WORDS + b] = (word == 0) ? 0u : ((word < 4) ? 0xffffffffu : 0u);
    }
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

// What a child range becomes: 0 a leaf reference (one primitive), 1 an inner node finished by sahSmallKernel, 2 an inner
// node split again on the next level.
TWK_D int sahChildKind(int count) { return (count == 1) ? 0 : ((count <= SAH_SMALL) ? 1 : 2); }

__global__ void sahSelectKernel(int numActive, const SahActive* __restrict__ active, const unsigned int* __restrict__ bins, int forceMiddle,
                                SahSplit* __restrict__ split, SahCounts* __restrict__ childCounts)
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
  sp.axis = bestAxis; sp.bin = bestBin; sp.leftCount = bestLeft; sp.first = a.first; sp.slotL = -1; sp.slotR = -1; sp.pad0 = sp.pad1 = 0;
  split[k] = sp;
  const int kindL = sahChildKind(bestLeft), kindR = sahChildKind(a.count - bestLeft);
  SahCounts c;
  c.inner = (kindL != 0) + (kindR != 0); c.large = (kindL == 2) + (kindR == 2); c.small = (kindL == 1) + (kindR == 1); c.pad = 0;
  childCounts[k] = c;
}

// One thread per active node, after the exclusive scan of childCounts: its children get their inner-node indices and their
// slots in the next level's active list / the small list — left child first — from the scanned offsets, i.e. from the
// node's position in the (deterministic) active list and nothing else. counters: [0] inner nodes allocated so far,
// [1] small nodes so far (both as of the START of this level; sahAdvanceKernel moves them on).
__global__ void sahAssignKernel(int numActive, const SahActive* __restrict__ active, SahSplit* __restrict__ split, const SahCounts* __restrict__ offsets,
                                const int* __restrict__ counters, int* __restrict__ left, int* __restrict__ right,
                                int* __restrict__ innerParent, int* __restrict__ leafParent, int2* __restrict__ range,
                                SahActive* __restrict__ nextActive, SahActive* __restrict__ smallList)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= numActive) return;
  const SahActive a = active[k];
  SahSplit sp = split[k];
  const SahCounts off = offsets[k];
  int inner = counters[0] + off.inner, large = off.large, small = counters[1] + off.small;
  int refs[2];
  const int childFirst[2] = {a.first, a.first + sp.leftCount}, childCount[2] = {sp.leftCount, a.count - sp.leftCount};
  int slots[2] = {-1, -1};
  for (int side = 0; side < 2; ++side)
  {
    const int kind = sahChildKind(childCount[side]);
    if (kind == 0) { leafParent[childFirst[side]] = a.node; refs[side] = ~childFirst[side]; continue; }
    const int c = inner++;
    innerParent[c] = a.node;
    range[c] = make_int2(childFirst[side], childCount[side]);
    SahActive child; child.node = c; child.first = childFirst[side]; child.count = childCount[side]; child.pad = 0;
    if (kind == 1) smallList[small++] = child;
    else { slots[side] = large; nextActive[large++] = child; }
    refs[side] = c;
  }
  left[a.node] = refs[0]; right[a.node] = refs[1];
  sp.slotL = slots[0]; sp.slotR = slots[1];
  split[k] = sp;
}

// Moves the running totals on by this level's sums (last offset + last count) and leaves the next level's active count
// where the host reads it. counters: [0] inner nodes, [1] small nodes, [2] next level's active count.
__global__ void sahAdvanceKernel(int numActive, const SahCounts* __restrict__ childCounts, const SahCounts* __restrict__ offsets, int* counters)
{
  const SahCounts c = childCounts[numActive - 1], o = offsets[numActive - 1];
  counters[0] += o.inner + c.inner;
  counters[1] += o.small + c.small;
  counters[2] = o.large + c.large;
}

// "Goes to the left child" per position (0 for positions outside every active node): the input of the partition's scan.
__global__ void sahFlagKernel(int count, const int* __restrict__ order, const int* __restrict__ slotOf,
                              const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                              const unsigned int* __restrict__ cb, const SahSplit* __restrict__ split, int* __restrict__ flags)
{
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= count) return;
  const int k = slotOf[pos];
  int toLeft = 0;
  if (k >= 0)
  {
    const SahSplit sp = split[k];
    if (sp.bin < 0) toLeft = (pos - sp.first) < sp.leftCount; // positional cut
    else
    {
      const int prim = order[pos];
      const V3 c = sahCentroid(primLo[prim], primHi[prim]);
      const float cc = (sp.axis == 0) ? c.x : ((sp.axis == 1) ? c.y : c.z);
      const unsigned int* w = cb + 6 * (size_t) k;
      toLeft = sahBin(cc, sahFromOrdered(w[sp.axis]), sahFromOrdered(w[3 + sp.axis])) < sp.bin;
    }
  }
  flags[pos] = toLeft;
}

// Stable partition: a primitive keeps its order among the primitives that go the same way. prefix = exclusive scan of flags.
__global__ void sahPartitionKernel(int count, const int* __restrict__ order, const int* __restrict__ slotOf,
                                   const SahSplit* __restrict__ split, const int* __restrict__ flags, const int* __restrict__ prefix,
                                   int* __restrict__ orderNext, int* __restrict__ slotNext)
{
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= count) return;
  const int k = slotOf[pos];
  const int prim = order[pos];
  if (k < 0) { orderNext[pos] = prim; slotNext[pos] = -1; return; } // finished range: stays where it is
  const SahSplit sp = split[k];
  const bool toLeft = flags[pos] != 0;
  const int leftBefore = prefix[pos] - prefix[sp.first];         // primitives of this node before `pos` that go left
  const int target = toLeft ? sp.first + leftBefore : sp.first + sp.leftCount + (pos - sp.first - leftBefore);
  orderNext[target] = prim;
  slotNext[target] = toLeft ? sp.slotL : sp.slotR;
}

// One thread finishes a node of 2..SAH_SMALL primitives: exact SAH sweep over the three axes at every split.
// A node of c primitives has c - 1 inner nodes below and including itself, so it creates c - 2 new ones: the exclusive
// scan of (count - 2) over the small list gives every thread the first index of a block of its own.
__global__ void sahSmallNeedKernel(int numSmall, const SahActive* __restrict__ smallList, int* __restrict__ need)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < numSmall) need[t] = smallList[t].count - 2;
}

__global__ void sahSmallKernel(int numSmall, const SahActive* __restrict__ smallList, int* __restrict__ order,
                               const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                               const int* __restrict__ counters, const int* __restrict__ nodeBase, int* __restrict__ left, int* __restrict__ right,
                               int* __restrict__ innerParent, int* __restrict__ leafParent, int2* __restrict__ range)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= numSmall) return;
  int nextNode = counters[0] + nodeBase[t]; // inner nodes allocated by the level loop + this thread's block
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
        const int c = nextNode++;
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

__global__ void sahRootKernel(int count, int* innerParent, int2* range, SahActive* active, SahActive* smallList, int* counters)
{
  // counters: [0] inner nodes allocated, [1] small nodes, [2] next level's active count
  innerParent[0] = -1;
  range[0] = make_int2(0, count);
  SahActive a; a.node = 0; a.first = 0; a.count = count; a.pad = 0;
  counters[0] = 1; counters[1] = 0; counters[2] = 0;
  if (count <= SAH_SMALL) { smallList[0] = a; counters[1] = 1; }
  else active[0] = a;
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
  // scans: "goes left" flags and their prefix over the positions (also reused for the small list's node blocks: maxSmall <= n),
  // child counts and their prefix over the active nodes
  SAH_CHECK(hipMalloc(&m_sahFlags, sizeof(int) * (n + 64))); SAH_CHECK(hipMalloc(&m_sahPrefix, sizeof(int) * (n + 64)));
  SAH_CHECK(hipMalloc(&m_sahChildCounts, sizeof(SahCounts) * maxLarge)); SAH_CHECK(hipMalloc(&m_sahChildOffsets, sizeof(SahCounts) * maxLarge));
  size_t bytesInt = 0, bytesCounts = 0;
  SAH_CHECK(rocprim::exclusive_scan(nullptr, bytesInt, m_sahFlags, m_sahPrefix, 0, n, rocprim::plus<int>(), (hipStream_t) 0));
  SAH_CHECK(rocprim::exclusive_scan(nullptr, bytesCounts, static_cast<SahCounts*>(m_sahChildCounts), static_cast<SahCounts*>(m_sahChildOffsets),
                                    SahCounts{0, 0, 0, 0}, maxLarge, SahCountsPlus(), (hipStream_t) 0));
  m_sahScanBytes = std::max(bytesInt, bytesCounts);
  SAH_CHECK(hipMalloc(&m_sahScanTemp, m_sahScanBytes > 0 ? m_sahScanBytes : 16));
  SAH_CHECK(hipMalloc(&m_sahCounters, sizeof(int) * 8));
  SAH_CHECK(hipMalloc(&m_sahCost, sizeof(double) * 2));
  m_sahCapacity = count;
  return hipSuccess;
}

void BvhBuilder::releaseSah()
{
  void* p[] = { m_sahOrder[0], m_sahOrder[1], m_sahSlot[0], m_sahSlot[1], m_sahActive[0], m_sahActive[1], m_sahSmall, m_sahSplit, m_sahCb, m_sahBins,
                m_sahFlags, m_sahPrefix, m_sahChildCounts, m_sahChildOffsets, m_sahScanTemp, m_sahCounters, m_sahCost };
  for (void* q : p) if (q) (void) hipFree(q);
  m_sahOrder[0] = m_sahOrder[1] = m_sahSlot[0] = m_sahSlot[1] = nullptr;
  m_sahActive[0] = m_sahActive[1] = nullptr; m_sahSmall = nullptr; m_sahSplit = nullptr;
  m_sahCb = m_sahBins = nullptr; m_sahFlags = m_sahPrefix = nullptr; m_sahChildCounts = m_sahChildOffsets = nullptr; m_sahScanTemp = nullptr; m_sahScanBytes = 0;
  m_sahCounters = nullptr; m_sahCost = nullptr;
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
  SahCounts* childCounts = static_cast<SahCounts*>(m_sahChildCounts);
  SahCounts* childOffsets = static_cast<SahCounts*>(m_sahChildOffsets);
  // counters: [0] inner nodes allocated, [1] small nodes, [2] next level's active count
  hipLaunchKernelGGL(sahRootKernel, dim3(1), dim3(1), 0, stream, count, m_innerParent, m_range, active[0], smallList, m_sahCounters);
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
    const long long clearWords = (long long) numActive * (6 + 3 * SAH_BINS * SAH_BIN_WORDS); // 342 words per node: beyond int at ~6.2 M active nodes
    const int nodeGrid = (numActive + 63) / 64;
    hipLaunchKernelGGL(sahClearKernel, dim3((unsigned int) std::min<long long>(4096, (clearWords + 255) / 256)), dim3(256), 0, stream, numActive, m_sahCb, m_sahBins);
    hipLaunchKernelGGL(sahBoundsKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[cur], m_sahSlot[cur], m_primLo, m_primHi, m_sahCb);
    hipLaunchKernelGGL(sahBinKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[cur], m_sahSlot[cur], m_primLo, m_primHi, m_sahCb, m_sahBins);
    hipLaunchKernelGGL(sahSelectKernel, dim3(nodeGrid), dim3(64), 0, stream, numActive, active[cur], m_sahBins, forceMiddle, split, childCounts);
    size_t bytes = m_sahScanBytes;
    SAH_CHECK(rocprim::exclusive_scan(m_sahScanTemp, bytes, childCounts, childOffsets, SahCounts{0, 0, 0, 0}, (size_t) numActive, SahCountsPlus(), stream));
    hipLaunchKernelGGL(sahAssignKernel, dim3(nodeGrid), dim3(64), 0, stream, numActive, active[cur], split, childOffsets, m_sahCounters,
                       m_left, m_right, m_innerParent, m_leafParent, m_range, active[next], smallList);
    hipLaunchKernelGGL(sahAdvanceKernel, dim3(1), dim3(1), 0, stream, numActive, childCounts, childOffsets, m_sahCounters);
    hipLaunchKernelGGL(sahFlagKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[cur], m_sahSlot[cur], m_primLo, m_primHi, m_sahCb, split, m_sahFlags);
    bytes = m_sahScanBytes;
    SAH_CHECK(rocprim::exclusive_scan(m_sahScanTemp, bytes, m_sahFlags, m_sahPrefix, 0, (size_t) count, rocprim::plus<int>(), stream));
    hipLaunchKernelGGL(sahPartitionKernel, dim3(grid), dim3(block), 0, stream, count, m_sahOrder[cur], m_sahSlot[cur], split, m_sahFlags, m_sahPrefix,
                       m_sahOrder[next], m_sahSlot[next]);
    SAH_CHECK(hipGetLastError());
    int nextCount = 0;
    SAH_CHECK(hipMemcpyAsync(&nextCount, m_sahCounters + 2, sizeof(int), hipMemcpyDeviceToHost, stream));
    SAH_CHECK(hipStreamSynchronize(stream));
    numActive = nextCount;
    cur = next;
  }
  if (numActive > 0) return hipErrorUnknown; // cannot happen: forced halving ends every range within 32 further levels
  int numSmall = 0;
  SAH_CHECK(hipMemcpyAsync(&numSmall, m_sahCounters + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
  SAH_CHECK(hipStreamSynchronize(stream));
  if (numSmall > 0)
  {
    const int smallGrid = (numSmall + 63) / 64;
    hipLaunchKernelGGL(sahSmallNeedKernel, dim3(smallGrid), dim3(64), 0, stream, numSmall, smallList, m_sahFlags);
    size_t bytes = m_sahScanBytes;
    SAH_CHECK(rocprim::exclusive_scan(m_sahScanTemp, bytes, m_sahFlags, m_sahPrefix, 0, (size_t) numSmall, rocprim::plus<int>(), stream));
    hipLaunchKernelGGL(sahSmallKernel, dim3(smallGrid), dim3(64), 0, stream, numSmall, smallList, m_sahOrder[cur], m_primLo, m_primHi,
                       m_sahCounters, m_sahPrefix, m_left, m_right, m_innerParent, m_leafParent, m_range);
  }
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
