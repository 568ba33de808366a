// On-device LBVH builder (stands where optixAccelBuild stood: GAS src/Device.cpp:1362-1407, IAS :1456-1486).
//
// Pipeline, all on the GPU: primitive boxes → 30-bit Morton code of the box centre → 64-bit key
// (code << 32 | primitive index) → radix sort (rocPRIM) → Karras 2012 binary radix tree (one thread per
// inner node) → bottom-up box refit with one atomic ticket per inner node → 64-byte BVH2 nodes that
// store the boxes of both children (one node fetch decides both descents).
// The same builder serves both levels: triangles of one geometry (bottom level, object space) and
// instance boxes (top level, world space).
#include "device_types.h"
#include "bvh_build.h"

#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

namespace twk {

// Order-preserving float <-> uint mapping for atomicMin/atomicMax on floats.
TWK_D unsigned int orderedBits(float f)
{
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
TWK_D float fromOrderedBits(unsigned int u)
{
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void initBoundsKernel(unsigned int* bounds)
{
  if (threadIdx.x < 3) bounds[threadIdx.x] = 0xffffffffu;      // min
  else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;          // max
}

// Soup descriptor of one flattened triangle: x = instance, y = first vertex of its geometry in the shared attribute
// array, z = first index of the TRIANGLE in the shared index array, w = primitive index inside the geometry.
// Vertex c of a soup triangle, in world space: the geometry's object-space position through the instance's
// object-to-world matrix — m0*x + m1*y + m2*z + m3 in fp32, the expression transformPoint (closesthit.cu:88-98) and
// the oracle evaluate.
TWK_D void soupVertices(const int4& d, const float* __restrict__ attributes, const unsigned int* __restrict__ indices,
                        const DevInstance* __restrict__ instances, const float*& a, const float*& b, const float*& c, V3& wa, V3& wb, V3& wc)
{
  const unsigned int i0 = indices[d.z], i1 = indices[d.z + 1], i2 = indices[d.z + 2];
  a = attributes + 12 * ((size_t) d.y + i0);
  b = attributes + 12 * ((size_t) d.y + i1);
  c = attributes + 12 * ((size_t) d.y + i2);
  const float* m = instances[d.x].objectToWorld;
  wa = transformPoint(m, v3(a[0], a[1], a[2]));
  wb = transformPoint(m, v3(b[0], b[1], b[2]));
  wc = transformPoint(m, v3(c[0], c[1], c[2]));
}

// Triangle boxes of one geometry (object space), or of the world-space soup of the flattened instances (soup != nullptr:
// attributes / indices are then the shared arrays). attributes: 12 floats per vertex (position first), indices: 3 per triangle.
__global__ void triangleBoxesKernel(const float* __restrict__ attributes, const unsigned int* __restrict__ indices,
                                    const int4* __restrict__ soup, const DevInstance* __restrict__ instances,
                                    int count, float4* __restrict__ primLo, float4* __restrict__ primHi, unsigned int* bounds)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  V3 va, vb, vc;
  if (soup != nullptr)
  {
    const float *a, *b, *c;
    soupVertices(soup[i], attributes, indices, instances, a, b, c, va, vb, vc);
  }
  else
  {
    const unsigned int i0 = indices[3 * i], i1 = indices[3 * i + 1], i2 = indices[3 * i + 2];
    const float* a = attributes + 12 * (size_t) i0;
    const float* b = attributes + 12 * (size_t) i1;
    const float* c = attributes + 12 * (size_t) i2;
    va = v3(a[0], a[1], a[2]); vb = v3(b[0], b[1], b[2]); vc = v3(c[0], c[1], c[2]);
  }
  const float lo[3] = { fminf(fminf(va.x, vb.x), vc.x), fminf(fminf(va.y, vb.y), vc.y), fminf(fminf(va.z, vb.z), vc.z) };
  const float hi[3] = { fmaxf(fmaxf(va.x, vb.x), vc.x), fmaxf(fmaxf(va.y, vb.y), vc.y), fmaxf(fmaxf(va.z, vb.z), vc.z) };
  primLo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
  primHi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
  for (int k = 0; k < 3; ++k)
  {
    atomicMin(&bounds[k], orderedBits(lo[k]));
    atomicMax(&bounds[3 + k], orderedBits(hi[k]));
  }
}

// Soup descriptors of one flattened instance: triangle k of its geometry → soup primitive first + k.
__global__ void soupDescriptorsKernel(int4* __restrict__ soup, int first, int count, int instance, int attributeBase, int indexBase)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  soup[first + k] = make_int4(instance, attributeBase, indexBase + 3 * k, k);
}

__global__ void boxBoundsKernel(const float4* __restrict__ primLo, const float4* __restrict__ primHi, int count, unsigned int* bounds)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float4 lo = primLo[i], hi = primHi[i];
  atomicMin(&bounds[0], orderedBits(lo.x)); atomicMin(&bounds[1], orderedBits(lo.y)); atomicMin(&bounds[2], orderedBits(lo.z));
  atomicMax(&bounds[3], orderedBits(hi.x)); atomicMax(&bounds[4], orderedBits(hi.y)); atomicMax(&bounds[5], orderedBits(hi.z));
}

TWK_D unsigned int expandBits10(unsigned int v)
{
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ void mortonKeysKernel(const float4* __restrict__ primLo, const float4* __restrict__ primHi, int count,
                                 const unsigned int* __restrict__ bounds, unsigned long long* __restrict__ keys)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float4 lo = primLo[i], hi = primHi[i];
  const float c[3] = { 0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z) };
  unsigned int q[3];
  for (int k = 0; k < 3; ++k)
  {
    const float bl = fromOrderedBits(bounds[k]);
    const float bh = fromOrderedBits(bounds[3 + k]);
    const float ext = bh - bl;
    float n = (ext > 0.0f) ? (c[k] - bl) / ext : 0.0f;
    n = fminf(fmaxf(n * 1024.0f, 0.0f), 1023.0f);
    q[k] = (unsigned int) n;
  }
  const unsigned int code = (expandBits10(q[0]) << 2) | (expandBits10(q[1]) << 1) | expandBits10(q[2]);
  keys[i] = ((unsigned long long) code << 32) | (unsigned int) i;
}

// Length of the common prefix of keys i and j, -1 when j is out of range (Karras 2012, section 4).
TWK_D int commonPrefix(const unsigned long long* __restrict__ keys, int count, int i, int j)
{
  if (j < 0 || j >= count) return -1;
  return __clzll((long long) (keys[i] ^ keys[j])); // keys are unique (index in the low word)
}

// One thread per inner node. childRef encoding while building: >= 0 inner node, < 0 leaf ~position.
__global__ void radixTreeKernel(const unsigned long long* __restrict__ keys, int count,
                                int* __restrict__ left, int* __restrict__ right,
                                int* __restrict__ innerParent, int* __restrict__ leafParent,
                                int2* __restrict__ range)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count - 1) return;

  const int d = (commonPrefix(keys, count, i, i + 1) - commonPrefix(keys, count, i, i - 1)) >= 0 ? 1 : -1;
  const int deltaMin = commonPrefix(keys, count, i, i - d);
  int lmax = 2;
  while (commonPrefix(keys, count, i, i + lmax * d) > deltaMin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
  {
    if (commonPrefix(keys, count, i, i + (l + t) * d) > deltaMin) l += t;
  }
  const int j = i + l * d;
  const int deltaNode = commonPrefix(keys, count, i, j);
  int s = 0;
  int t = l;
  do
  {
    t = (t + 1) >> 1;
    if (commonPrefix(keys, count, i, i + (s + t) * d) > deltaNode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + min(d, 0);

  const int lo = min(i, j), hi = max(i, j);
  const int leftRef  = (lo == gamma)     ? ~gamma       : gamma;
  const int rightRef = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  left[i]  = leftRef;
  right[i] = rightRef;
  range[i] = make_int2(lo, hi - lo + 1); // sorted positions covered by this node: first, count
  if (leftRef  >= 0) innerParent[leftRef]  = i; else leafParent[~leftRef]  = i;
  if (rightRef >= 0) innerParent[rightRef] = i; else leafParent[~rightRef] = i;
  if (i == 0) innerParent[0] = -1;
}

TWK_D void padBox(float4& lo, float4& hi)
{
  // The watertight triangle test reports t with an ABSOLUTE error that scales with the triangle's size and with the
  // magnitude of its coordinates (the edge functions are differences of products of vertex-minus-origin terms), not
  // with t: a ray that starts 2e-4 above a 16-unit floor triangle gets a t that is off by 3e-7, 1e-3 of its value —
  // measured: with axis-tight boxes (a flat floor: pad 1e-30 in y) the slab interval [3.52942e-4, 3.52942e-4] was
  // culled against a current best of 3.52938e-4 although the triangle test would have returned 3.5266e-4. Every
  // primitive box therefore grows on ALL axes by 2^-17 (128 ulp) of (largest coordinate magnitude + box diagonal):
  // two orders of magnitude above the observed error, 0.03 % of a 0.03-unit triangle. Errors that scale with the
  // distance travelled are covered by the relative widening of the slab test itself (trace_device.h).
  const float k = 7.6293945e-6f; // 2^-17
  const float m = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))), fmaxf(fabsf(lo.z), fabsf(hi.z)));
  const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  const float e = k * (m + sqrtf(dx * dx + dy * dy + dz * dz)) + 1.0e-30f;
  lo.x -= e; lo.y -= e; lo.z -= e;
  hi.x += e; hi.y += e; hi.z += e;
}

TWK_D void writeNode(BvhNode* node, const float4& lo0, const float4& hi0, const float4& lo1, const float4& hi1, int c0, int c1)
{
  float4* p = reinterpret_cast<float4*>(node);
  p[0] = make_float4(lo0.x, lo0.y, lo0.z, hi0.x);
  p[1] = make_float4(hi0.y, hi0.z, lo1.x, lo1.y);
  p[2] = make_float4(lo1.z, hi1.x, hi1.y, hi1.z);
  p[3] = make_float4(__int_as_float(c0), __int_as_float(c1), 0.0f, 0.0f);
}

// Wide (4-ary) node of inner node i = the children of its children, 128 bytes (layout at writeWideNode): a traversal step over a wide node crosses two levels of the binary tree
// with one round of loads instead of two dependent ones. A child that is a leaf occupies one entry; unused entries
// get an empty box. Wide nodes share the index space of the binary nodes (every inner node has one; only those at
// even depth below the root are ever visited).
struct WideEntry { float4 lo, hi; int ref; };
// (Round 3, measured and not kept: opening twice the inner entry with the largest surface area instead of taking the four
// grandchildren — node visits per ray 7.64 -> 7.43 on C2, triangle tests 3.07 -> 3.28, C2 -7 %: profiles/r03y_wide_collapse_greedy.txt.)
TWK_D float wideHalfArea(const float4& lo, const float4& hi)
{
  const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  return dx * dy + dy * dz + dz * dx;
}

TWK_D void emptyEntry(WideEntry& e)
{
  const float inf = __uint_as_float(0x7f800000u);
  // lo = hi = +inf: every plane distance is +inf or -inf on both sides, so the slab interval is empty for any ray.
  // (An inverted box lo > hi does NOT work: the slab test orders the two plane distances itself.)
  e.lo = make_float4(inf, inf, inf, 0.0f); e.hi = make_float4(inf, inf, inf, 0.0f); e.ref = ~0;
}

// Expands child `ref` (box lo/hi) into one entry (leaf) or the two entries of its own children (inner node).
TWK_D void expandChild(const BvhNode* outNodes, int nodeBase, int ref, const float4& lo, const float4& hi, WideEntry* e, int& n)
{
  if (ref < 0) { e[n].lo = lo; e[n].hi = hi; e[n].ref = ref; ++n; return; }
  const float4* c = reinterpret_cast<const float4*>(outNodes + (ref - nodeBase));
  const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
  e[n].lo = make_float4(c0.x, c0.y, c0.z, 0.0f); e[n].hi = make_float4(c0.w, c1.x, c1.y, 0.0f); e[n].ref = __float_as_int(c3.x); ++n;
  e[n].lo = make_float4(c1.z, c1.w, c2.x, 0.0f); e[n].hi = make_float4(c2.y, c2.z, c2.w, 0.0f); e[n].ref = __float_as_int(c3.y); ++n;
}

// Wide node layout, 8 float4: (lo_k.xyz, ref_k) and (hi_k.xyz, unused) for k = 0..3, with the four references in
// the .w of the FIRST four float4 — (lo0,ref0) (hi0,ref1) (lo1,ref2) (hi1,ref3) (lo2,-) (hi2,-) (lo3,-) (hi3,-) —
// so that they arrive with the box data the first slab tests need. (With the references in a float4 of their own
// the compiler sank those loads behind the box tests: two extra dependent L2 round trips per traversal step.)
TWK_D void writeWideNode(BvhNode* wide, const WideEntry* e)
{
  float4* p = reinterpret_cast<float4*>(wide);
  p[0] = make_float4(e[0].lo.x, e[0].lo.y, e[0].lo.z, __int_as_float(e[0].ref));
  p[1] = make_float4(e[0].hi.x, e[0].hi.y, e[0].hi.z, __int_as_float(e[1].ref));
  p[2] = make_float4(e[1].lo.x, e[1].lo.y, e[1].lo.z, __int_as_float(e[2].ref));
  p[3] = make_float4(e[1].hi.x, e[1].hi.y, e[1].hi.z, __int_as_float(e[3].ref));
  p[4] = make_float4(e[2].lo.x, e[2].lo.y, e[2].lo.z, 0.0f);
  p[5] = make_float4(e[2].hi.x, e[2].hi.y, e[2].hi.z, 0.0f);
  p[6] = make_float4(e[3].lo.x, e[3].lo.y, e[3].lo.z, 0.0f);
  p[7] = make_float4(e[3].hi.x, e[3].hi.y, e[3].hi.z, 0.0f);
}

// One thread per leaf walks up; the second thread to arrive at an inner node (ticket == 1) owns it.
// leafMode 0 (triangles): leaf reference = ~(first slot | (count - 1) << 28 | leafFlag) with first = leafBase + sorted
// position; a child subtree that covers at most maxLeaf sorted positions is referenced as ONE leaf (its slots are
// contiguous because the leaves of a radix tree are in key order), which removes the bottom levels of the tree.
// leafFlag = TWK_LEAF_WORLD for the world-space soup of the flattened instances (device_types.h), else 0.
// leafMode 1 (top level): child reference = ~leafPayload[primitive index]: payload = instance index gives a leaf;
// payload = ~(root node of the soup) gives an INNER reference — the soup's tree becomes a subtree of the top level.
__global__ void refitKernel(const unsigned long long* __restrict__ keys, int count,
                            const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                            const int* __restrict__ left, const int* __restrict__ right,
                            const int* __restrict__ innerParent, const int* __restrict__ leafParent,
                            const int2* __restrict__ range,
                            unsigned int* __restrict__ tickets, float4* nodeLo, float4* nodeHi,
                            BvhNode* outNodes, BvhNode* __restrict__ outWide, int nodeBase, int leafMode, int leafBase, int maxLeaf,
                            const int* __restrict__ leafPayload, int leafFlag, float* nodeCost /* indexed by absolute node index; nullptr: every wide node = the four grandchildren */)
{
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= count) return;

  int node = leafParent[leaf];
  while (node >= 0)
  {
    __threadfence();
    const unsigned int ticket = atomicAdd(&tickets[node], 1u);
    if (ticket == 0) return; // the sibling subtree is not finished; its thread will continue
    __threadfence();

    const int l = left[node], r = right[node];
    float4 lo0, hi0, lo1, hi1;
    int c0, c1;
    int height0 = 0, height1 = 0; // height of the FINAL binary tree below each child (a collapsed subtree is a leaf: 0), kept in nodeLo[].w
    if (l < 0)
    {
      const unsigned int prim = (unsigned int) (keys[~l] & 0xffffffffull);
      lo0 = primLo[prim]; hi0 = primHi[prim]; padBox(lo0, hi0);
      c0 = (leafMode == 0) ? ~((leafBase + ~l) | leafFlag) : ~leafPayload[prim];
    }
    else
    {
      lo0 = nodeLo[l]; hi0 = nodeHi[l];
      const int2 rg = range[l];
      c0 = (leafMode == 0 && rg.y <= maxLeaf) ? ~((leafBase + rg.x) | ((rg.y - 1) << 28) | leafFlag) : nodeBase + l;
      if (c0 >= 0) height0 = __float_as_int(lo0.w);
    }
    if (r < 0)
    {
      const unsigned int prim = (unsigned int) (keys[~r] & 0xffffffffull);
      lo1 = primLo[prim]; hi1 = primHi[prim]; padBox(lo1, hi1);
      c1 = (leafMode == 0) ? ~((leafBase + ~r) | leafFlag) : ~leafPayload[prim];
    }
    else
    {
      lo1 = nodeLo[r]; hi1 = nodeHi[r];
      const int2 rg = range[r];
      c1 = (leafMode == 0 && rg.y <= maxLeaf) ? ~((leafBase + rg.x) | ((rg.y - 1) << 28) | leafFlag) : nodeBase + r;
      if (c1 >= 0) height1 = __float_as_int(lo1.w);
    }
    writeNode(&outNodes[node], lo0, hi0, lo1, hi1, c0, c1);
    {
      WideEntry e[4];
      int n = 0;
      if (nodeCost != nullptr)
      {
        // Which of the two children give way to their own children in this node's wide node: the cut with the fewest
        // expected wide-node visits below this node, N(i) = 1 + sum over the inner entries e of area(e) / area(i) * N(e)
        // (leaf entries are reached with the same probability under every cut). The binary nodes and the N of everything
        // below are final: this walk is bottom-up.
        WideEntry child[2], grand[2][2];
        child[0].lo = lo0; child[0].hi = hi0; child[0].ref = c0;
        child[1].lo = lo1; child[1].hi = hi1; child[1].ref = c1;
        const float areaNode = wideHalfArea(make_float4(fminf(lo0.x, lo1.x), fminf(lo0.y, lo1.y), fminf(lo0.z, lo1.z), 0.0f),
                                            make_float4(fmaxf(hi0.x, hi1.x), fmaxf(hi0.y, hi1.y), fmaxf(hi0.z, hi1.z), 0.0f));
        const float scale = (areaNode > 0.0f) ? 1.0f / areaNode : 0.0f;
        float keep[2], open[2]; // expected visits below child k when it stays one entry / when its children take its place
        for (int k = 0; k < 2; ++k)
        {
          keep[k] = 0.0f; open[k] = 0.0f;
          if (child[k].ref < 0) continue;
          keep[k] = fminf(wideHalfArea(child[k].lo, child[k].hi) * scale, 1.0f) * nodeCost[child[k].ref];
          int m = 0;
          expandChild(outNodes, nodeBase, child[k].ref, child[k].lo, child[k].hi, grand[k], m);
          for (int g = 0; g < 2; ++g)
            if (grand[k][g].ref >= 0) open[k] += fminf(wideHalfArea(grand[k][g].lo, grand[k][g].hi) * scale, 1.0f) * nodeCost[grand[k][g].ref];
        }
        float best = 0.0f;
        int bestMask = -1;
        for (int mask = 3; mask >= 0; --mask) // ties: the wider node
        {
          if (((mask & 1) && c0 < 0) || ((mask & 2) && c1 < 0)) continue;
          const float cost = 1.0f + ((mask & 1) ? open[0] : keep[0]) + ((mask & 2) ? open[1] : keep[1]);
          if (bestMask < 0 || cost < best) { best = cost; bestMask = mask; }
        }
        for (int k = 0; k < 2; ++k)
        {
          if (bestMask & (1 << k)) { e[n++] = grand[k][0]; e[n++] = grand[k][1]; }
          else e[n++] = child[k];
        }
        nodeCost[nodeBase + node] = best;
      }
      else
      {
        expandChild(outNodes, nodeBase, c0, lo0, hi0, e, n);
        expandChild(outNodes, nodeBase, c1, lo1, hi1, e, n);
      }
      for (; n < 4; ++n) emptyEntry(e[n]);
      writeWideNode(&outWide[2 * node], e);
    }
    nodeLo[node] = make_float4(fminf(lo0.x, lo1.x), fminf(lo0.y, lo1.y), fminf(lo0.z, lo1.z), __int_as_float(1 + max(height0, height1)));
    nodeHi[node] = make_float4(fmaxf(hi0.x, hi1.x), fmaxf(hi0.y, hi1.y), fmaxf(hi0.z, hi1.z), 0.0f);
    node = innerParent[node];
  }
}

// A single primitive has no inner node: emit one node whose second child can never be hit.
__global__ void singleLeafKernel(const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                                 BvhNode* outNodes, BvhNode* outWide, float4* nodeLo, float4* nodeHi, int leafMode, int leafBase,
                                 const int* __restrict__ leafPayload, int leafFlag, float* nodeCost)
{
  if (nodeCost != nullptr) nodeCost[0] = 1.0f;
  const int leafRef = (leafMode == 0) ? ~(leafBase | leafFlag) : ~leafPayload[0];
  float4 lo = primLo[0], hi = primHi[0];
  padBox(lo, hi);
  const float inf = __uint_as_float(0x7f800000u);
  const float4 elo = make_float4(inf, inf, inf, 0.0f), ehi = make_float4(inf, inf, inf, 0.0f); // never hit, see emptyEntry()
  writeNode(&outNodes[0], lo, hi, elo, ehi, leafRef, ~0);
  WideEntry e[4];
  e[0].lo = lo; e[0].hi = hi; e[0].ref = leafRef;
  emptyEntry(e[1]); emptyEntry(e[2]); emptyEntry(e[3]);
  writeWideNode(outWide, e);
  lo.w = __int_as_float(1); // height of this one-node tree (refitKernel keeps heights in nodeLo[].w)
  nodeLo[0] = lo; nodeHi[0] = hi;
}

// Triangle slots in leaf order: three float4 per slot, w of the first = primitive index. The same pass writes the
// 128-byte shading record of the slot (one cache line, eight float4), ordered by who needs what, so that shading does
// not chase indices → three 48-byte vertex records and fetches only the part its material uses — every fetch here is
// one divergent lane address, and a CU takes one of those per clock (tools/gather_probe.hip):
//   [0..2]  geometric normal cross(v1 - v0, v2 - v0) as closesthit.cu:150 computes it (same float operations, done
//           once here), then the three vertex normals                                   — every hit
//   [3..5.x] the three vertex tangents                                                  — GGX materials (TBN) only
//   [5.y..7.y] the three vertex texture coordinates                                     — textured materials only
__global__ void emitTrianglesKernel(const float* __restrict__ attributes, const unsigned int* __restrict__ indices,
                                    const int4* __restrict__ soup, const DevInstance* __restrict__ instances,
                                    const unsigned long long* __restrict__ keys, int count, float4* __restrict__ triangles,
                                    float4* __restrict__ shadeTriangles)
{
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= count) return;
  const unsigned int prim = (unsigned int) (keys[slot] & 0xffffffffull);
  const float *a, *b, *c;
  if (soup != nullptr)
  {
    // world-space positions for traversal, tagged with primitive and instance; the shading record below stays in
    // object space (shading transforms normals / tangents with the instance matrices as for any other hit)
    const int4 d = soup[prim];
    V3 wa, wb, wc;
    soupVertices(d, attributes, indices, instances, a, b, c, wa, wb, wc);
    triangles[3 * (size_t) slot + 0] = make_float4(wa.x, wa.y, wa.z, __int_as_float(d.w));
    triangles[3 * (size_t) slot + 1] = make_float4(wb.x, wb.y, wb.z, __int_as_float(d.x));
    triangles[3 * (size_t) slot + 2] = make_float4(wc.x, wc.y, wc.z, 0.0f);
  }
  else
  {
    const unsigned int i0 = indices[3 * prim], i1 = indices[3 * prim + 1], i2 = indices[3 * prim + 2];
    a = attributes + 12 * (size_t) i0;
    b = attributes + 12 * (size_t) i1;
    c = attributes + 12 * (size_t) i2;
    triangles[3 * (size_t) slot + 0] = make_float4(a[0], a[1], a[2], __uint_as_float(prim));
    triangles[3 * (size_t) slot + 1] = make_float4(b[0], b[1], b[2], 0.0f);
    triangles[3 * (size_t) slot + 2] = make_float4(c[0], c[1], c[2], 0.0f);
  }
  const V3 v0 = v3(a[0], a[1], a[2]), v1 = v3(b[0], b[1], b[2]), v2 = v3(c[0], c[1], c[2]);
  const V3 ng = cross(v1 - v0, v2 - v0);
  float4* out = shadeTriangles + TWK_SHADE_RECORD * (size_t) slot;
  out[0] = make_float4(ng.x, ng.y, ng.z, a[6]);
  out[1] = make_float4(a[7], a[8], b[6], b[7]);
  out[2] = make_float4(b[8], c[6], c[7], c[8]);
  out[3] = make_float4(a[3], a[4], a[5], b[3]);
  out[4] = make_float4(b[4], b[5], c[3], c[4]);
  out[5] = make_float4(c[5], a[9], a[10], a[11]);
  out[6] = make_float4(b[9], b[10], b[11], c[9]);
  out[7] = make_float4(c[10], c[11], 0.0f, 0.0f);
}

// Quantised copy of the wide nodes (device_types.h "quantised wide node"), one thread per inner node index. The grid
// cell of an axis is the smallest power of two with extent / cell < 254; lo planes are rounded down and hi planes up,
// and each is checked against the float expression the traversal evaluates (origin + q * cell).
__global__ void quantizeWideKernel(const BvhNode* __restrict__ wide, float4* __restrict__ out, int count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float4* w = reinterpret_cast<const float4*>(wide + 2 * (size_t) i);
  const float inf = __uint_as_float(0x7f800000u);
  float lo[4][3], hi[4][3];
  int ref[4];
  bool valid[4];
  float org[3] = {inf, inf, inf}, top[3] = {-inf, -inf, -inf};
  for (int k = 0; k < 4; ++k)
  {
    const float4 a = w[2 * k], b = w[2 * k + 1];
    lo[k][0] = a.x; lo[k][1] = a.y; lo[k][2] = a.z; hi[k][0] = b.x; hi[k][1] = b.y; hi[k][2] = b.z;
    ref[k] = __float_as_int(w[k].w);
    valid[k] = !(a.x == inf && b.x == inf); // emptyEntry
    if (valid[k]) for (int c = 0; c < 3; ++c) { org[c] = fminf(org[c], lo[k][c]); top[c] = fmaxf(top[c], hi[k][c]); }
  }
  float cell[3];
  unsigned int qlo[3] = {0u, 0u, 0u}, qhi[3] = {0u, 0u, 0u};
  for (int c = 0; c < 3; ++c)
  {
    if (!(org[c] < inf)) org[c] = 0.0f; // no valid child (never visited)
    const float extent = fmaxf(top[c] - org[c], 0.0f);
    int e = -125;
    if (extent > 0.0f && extent < inf) { frexpf(extent * (1.0f / 254.0f), &e); e = max(-125, min(126, e)); }
    cell[c] = ldexpf(1.0f, e);
    for (int k = 0; k < 4; ++k)
    {
      if (!valid[k]) { qlo[c] |= 255u << (8 * k); continue; } // unused entry: inverted box (lo = 255, hi = 0), never entered
      int ql = (int) fminf(fmaxf(floorf((lo[k][c] - org[c]) / cell[c]), 0.0f), 255.0f);
      int qh = (int) fminf(fmaxf(ceilf((hi[k][c] - org[c]) / cell[c]), 0.0f), 255.0f);
      if (ql > 0 && __builtin_fmaf((float) ql, cell[c], org[c]) > lo[k][c]) --ql;
      if (qh < 255 && __builtin_fmaf((float) qh, cell[c], org[c]) < hi[k][c]) ++qh;
      qlo[c] |= (unsigned int) ql << (8 * k);
      qhi[c] |= (unsigned int) qh << (8 * k);
    }
  }
  float4* o = out + 4 * (size_t) i;
  o[0] = make_float4(org[0], org[1], org[2], cell[0]);
  o[1] = make_float4(cell[1], cell[2], __uint_as_float(qlo[0]), __uint_as_float(qlo[1]));
  o[2] = make_float4(__uint_as_float(qlo[2]), __uint_as_float(qhi[0]), __uint_as_float(qhi[1]), __uint_as_float(qhi[2]));
  // an unused entry repeats the reference of the first child: should a degenerate node (no extent on any axis) let a ray
  // into it after all, the ray walks a real subtree twice instead of following a wild reference
  int first = 0;
  while (first < 3 && !valid[first]) ++first;
  o[3] = make_float4(__int_as_float(valid[0] ? ref[0] : ref[first]), __int_as_float(valid[1] ? ref[1] : ref[first]),
                     __int_as_float(valid[2] ? ref[2] : ref[first]), __int_as_float(valid[3] ? ref[3] : ref[first]));
}

void launchQuantizeWide(const BvhNode* wide, float4* out, int count, hipStream_t stream)
{
  if (count > 0) hipLaunchKernelGGL(quantizeWideKernel, dim3((count + 255) / 256), dim3(256), 0, stream, wide, out, count);
}

// Top-of-tree cache (device_types.h TWK_NODE_CACHED): breadth-first from the root over the quantised wide nodes, the
// first TWK_TOP_NODES inner nodes; references among them become TWK_NODE_CACHED | slot. One block: thread 0 walks the
// queue, then one thread per slot copies its node and rewrites its references.
__global__ void topCacheKernel(const float4* __restrict__ wideQ, int root, int root2, float4* __restrict__ top, int numSlots)
{
  __shared__ int queue[TWK_TOP_NODES];
  __shared__ int count;
  if (threadIdx.x == 0)
  {
    int n = 1;
    queue[0] = root;
    if (root2 != TWK_BVH_SENTINEL) queue[n++] = root2; // the second half of an 8-wide root: slot 1
    for (int i = 0; i < n; ++i)
    {
      const float4* w = wideQ + 4 * (size_t) queue[i];
      const float4 refs = w[3];
      const int r[4] = {__float_as_int(refs.x), __float_as_int(refs.y), __float_as_int(refs.z), __float_as_int(refs.w)};
      const unsigned int qlx = __float_as_uint(w[1].z), qhx = __float_as_uint(w[2].y);
      for (int k = 0; k < 4; ++k)
      {
        const bool unused = ((qlx >> (8 * k)) & 0xffu) > ((qhx >> (8 * k)) & 0xffu); // inverted box
        if (!unused && r[k] >= 0 && r[k] != TWK_BVH_SENTINEL && n < numSlots) queue[n++] = r[k];
      }
    }
    count = n;
  }
  __syncthreads();
  const int n = count;
  for (int i = threadIdx.x; i < numSlots; i += blockDim.x)
  {
    float4* out = top + 4 * i;
    if (i >= n) { for (int k = 0; k < 4; ++k) out[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); continue; } // slot never referenced
    const float4* w = wideQ + 4 * (size_t) queue[i];
    for (int k = 0; k < 3; ++k) out[k] = w[k];
    const float4 refs = w[3];
    int r[4] = {__float_as_int(refs.x), __float_as_int(refs.y), __float_as_int(refs.z), __float_as_int(refs.w)};
    for (int k = 0; k < 4; ++k)
      if (r[k] >= 0 && r[k] != TWK_BVH_SENTINEL)
        for (int j = 0; j < n; ++j) if (queue[j] == r[k]) { r[k] = TWK_NODE_CACHED | j; break; }
    out[3] = make_float4(__int_as_float(r[0]), __int_as_float(r[1]), __int_as_float(r[2]), __int_as_float(r[3]));
  }
}

void launchTopCache(const float4* wideQ, int root, int root2, float4* top, int numSlots, hipStream_t stream)
{
  hipLaunchKernelGGL(topCacheKernel, dim3(1), dim3(64), 0, stream, wideQ, root, root2, top, numSlots < TWK_TOP_NODES ? numSlots : TWK_TOP_NODES);
}

// An 8-wide root as TWO wide nodes. Every ray takes the node step of the root and, with probability ~1 each, of the large
// nodes right below it (Cornell box: root, {floor + spheres}, {three walls}: 2.9 steps). The entries of the root's wide node
// are opened — largest surface area first — while the list fits eight; if what was opened would have been entered by more
// than one ray in one on average (sum of area ratios > 1: the price of the second node step every ray now takes), the list
// is written as two full-precision wide nodes at `firstNew` and `firstNew + 1`, inner nodes in the first (a ray starts there
// and finds the second on its stack), and result[0] = 1. The two are quantised with all other nodes afterwards.
__global__ void wideRootKernel(BvhNode* wide, int root, int firstNew, int* result)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float inf = __uint_as_float(0x7f800000u);
  WideEntry e[12];
  int n = 0;
  auto entriesOf = [&](int node, WideEntry* out) -> int
  {
    const float4* p = reinterpret_cast<const float4*>(wide + 2 * (size_t) node);
    const int refs[4] = { __float_as_int(p[0].w), __float_as_int(p[1].w), __float_as_int(p[2].w), __float_as_int(p[3].w) };
    int m = 0;
    for (int k = 0; k < 4; ++k)
    {
      const float4 lo = p[2 * k], hi = p[2 * k + 1];
      if (lo.x == inf && hi.x == inf) continue; // emptyEntry
      out[m].lo = make_float4(lo.x, lo.y, lo.z, 0.0f); out[m].hi = make_float4(hi.x, hi.y, hi.z, 0.0f); out[m].ref = refs[k];
      ++m;
    }
    return m;
  };
  n = entriesOf(root, e);
  float4 sceneLo = make_float4(inf, inf, inf, 0.0f), sceneHi = make_float4(-inf, -inf, -inf, 0.0f);
  for (int k = 0; k < n; ++k)
  {
    sceneLo = make_float4(fminf(sceneLo.x, e[k].lo.x), fminf(sceneLo.y, e[k].lo.y), fminf(sceneLo.z, e[k].lo.z), 0.0f);
    sceneHi = make_float4(fmaxf(sceneHi.x, e[k].hi.x), fmaxf(sceneHi.y, e[k].hi.y), fmaxf(sceneHi.z, e[k].hi.z), 0.0f);
  }
  const float sceneArea = fmaxf(wideHalfArea(sceneLo, sceneHi), 1.0e-30f);
  float opened = 0.0f; // expected node steps per ray the opened entries stood for
  bool closed[12] = { false, false, false, false, false, false, false, false, false, false, false, false };
  for (int round = 0; round < 16; ++round)
  {
    int pick = -1; float pickArea = -1.0f;
    for (int k = 0; k < n; ++k)
    {
      if (closed[k] || e[k].ref < 0 || e[k].ref == TWK_BVH_SENTINEL) continue;
      const float a = wideHalfArea(e[k].lo, e[k].hi);
      if (a > pickArea) { pickArea = a; pick = k; }
    }
    if (pick < 0) break;
    WideEntry c[4];
    const int m = entriesOf(e[pick].ref, c);
    if (n - 1 + m > 8) { closed[pick] = true; continue; }
    opened += fminf(pickArea / sceneArea, 1.0f);
    e[pick] = e[n - 1]; closed[pick] = closed[n - 1]; --n;
    for (int k = 0; k < m; ++k) { e[n] = c[k]; closed[n] = false; ++n; }
  }
  if (n <= 4 || !(opened > 1.0f)) { result[0] = 0; return; }
  // inner nodes first (stable)
  for (int i = 1; i < n; ++i)
  {
    if (e[i].ref < 0) continue;
    int j = i;
    while (j > 0 && e[j - 1].ref < 0) { const WideEntry t = e[j]; e[j] = e[j - 1]; e[j - 1] = t; --j; }
  }
  WideEntry first[4], second[4];
  for (int k = 0; k < 4; ++k) { first[k] = e[k]; if (4 + k < n) second[k] = e[4 + k]; else emptyEntry(second[k]); }
  writeWideNode(wide + 2 * (size_t) firstNew, first);
  writeWideNode(wide + 2 * (size_t) (firstNew + 1), second);
  result[0] = 1;
}

void launchWideRoot(BvhNode* wide, int root, int firstNew, int* result, hipStream_t stream)
{
  hipLaunchKernelGGL(wideRootKernel, dim3(1), dim3(1), 0, stream, wide, root, firstNew, result);
}

#define BVH_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

hipError_t BvhBuilder::reserve(int count)
{
  if (count <= m_capacity) return hipSuccess;
  release();
  const size_t n = (size_t) count;
  BVH_CHECK(hipMalloc(&m_primLo, sizeof(float4) * n));
  BVH_CHECK(hipMalloc(&m_primHi, sizeof(float4) * n));
  BVH_CHECK(hipMalloc(&m_keysIn, sizeof(unsigned long long) * n));
  BVH_CHECK(hipMalloc(&m_keysOut, sizeof(unsigned long long) * n));
  BVH_CHECK(hipMalloc(&m_left, sizeof(int) * n));
  BVH_CHECK(hipMalloc(&m_right, sizeof(int) * n));
  BVH_CHECK(hipMalloc(&m_innerParent, sizeof(int) * n));
  BVH_CHECK(hipMalloc(&m_leafParent, sizeof(int) * n));
  BVH_CHECK(hipMalloc(&m_range, sizeof(int2) * n));
  BVH_CHECK(hipMalloc(&m_tickets, sizeof(unsigned int) * n));
  BVH_CHECK(hipMalloc(&m_nodeLo, sizeof(float4) * n));
  BVH_CHECK(hipMalloc(&m_nodeHi, sizeof(float4) * n));
  BVH_CHECK(hipMalloc(&m_bounds, sizeof(unsigned int) * 8));
  BVH_CHECK(hipMalloc(&m_leafPayload, sizeof(int) * n));
  m_sortBytes = 0;
  BVH_CHECK(rocprim::radix_sort_keys(nullptr, m_sortBytes, m_keysIn, m_keysOut, n, 0, 64, (hipStream_t) 0));
  BVH_CHECK(hipMalloc(&m_sortTemp, m_sortBytes > 0 ? m_sortBytes : 16));
  m_capacity = count;
  return hipSuccess;
}

void BvhBuilder::release()
{
  void* p[] = { m_primLo, m_primHi, m_keysIn, m_keysOut, m_left, m_right, m_innerParent, m_leafParent, m_range, m_tickets, m_nodeLo, m_nodeHi, m_bounds, m_sortTemp, m_leafPayload };
  for (void* q : p) if (q) (void) hipFree(q);
  m_primLo = m_primHi = m_nodeLo = m_nodeHi = nullptr;
  m_keysIn = m_keysOut = nullptr;
  m_left = m_right = m_innerParent = m_leafParent = nullptr; m_range = nullptr;
  m_tickets = nullptr; m_bounds = nullptr; m_sortTemp = nullptr; m_leafPayload = nullptr;
  m_capacity = 0;
  releaseSah();
}

hipError_t BvhBuilder::buildFromBoxes(hipStream_t stream, int count, BvhNode* outNodes, BvhNode* outWide, int nodeBase, int leafMode, int leafBase, int leafFlag)
{
  const int block = 256;
  const int grid  = (count + block - 1) / block;
  if (count == 1)
  {
    hipLaunchKernelGGL(singleLeafKernel, dim3(1), dim3(1), 0, stream, m_primLo, m_primHi, outNodes, outWide, m_nodeLo, m_nodeHi, leafMode, leafBase, m_leafPayload, leafFlag, m_nodeCost ? m_nodeCost + nodeBase : nullptr);
    // keysOut[0] must still name primitive 0 for emitTriangles
    BVH_CHECK(hipMemsetAsync(m_keysOut, 0, sizeof(unsigned long long), stream));
    return hipGetLastError();
  }
  if (m_quality == 1)
  {
    BVH_CHECK(buildSahTopology(stream, count)); // binned-SAH splits: same arrays as the radix tree below
  }
  else
  {
    hipLaunchKernelGGL(mortonKeysKernel, dim3(grid), dim3(block), 0, stream, m_primLo, m_primHi, count, m_bounds, m_keysIn);
    BVH_CHECK(rocprim::radix_sort_keys(m_sortTemp, m_sortBytes, m_keysIn, m_keysOut, (size_t) count, 0, 64, stream));
    hipLaunchKernelGGL(radixTreeKernel, dim3(grid), dim3(block), 0, stream, m_keysOut, count, m_left, m_right, m_innerParent, m_leafParent, m_range);
  }
  BVH_CHECK(hipMemsetAsync(m_tickets, 0, sizeof(unsigned int) * count, stream));
  hipLaunchKernelGGL(refitKernel, dim3(grid), dim3(block), 0, stream, m_keysOut, count, m_primLo, m_primHi, m_left, m_right,
                     m_innerParent, m_leafParent, m_range, m_tickets, m_nodeLo, m_nodeHi, outNodes, outWide, nodeBase, leafMode, leafBase, m_maxLeaf, m_leafPayload, leafFlag, m_nodeCost);
  BVH_CHECK(hipGetLastError());
  return (leafMode == 0) ? accumulateSahCost(stream, count) : hipSuccess;
}

hipError_t BvhBuilder::buildTriangles(hipStream_t stream, const float* attributes, const unsigned int* indices, int numTriangles,
                                      BvhNode* outNodes, BvhNode* outWide, int nodeBase, float4* outTriangles, float4* outShadeTriangles, int triangleBase, float rootBounds[6],
                                      const int4* soup, const DevInstance* instances)
{
  BVH_CHECK(reserve(numTriangles));
  const int block = 256;
  const int grid  = (numTriangles + block - 1) / block;
  hipLaunchKernelGGL(initBoundsKernel, dim3(1), dim3(64), 0, stream, m_bounds);
  hipLaunchKernelGGL(triangleBoxesKernel, dim3(grid), dim3(block), 0, stream, attributes, indices, soup, instances, numTriangles, m_primLo, m_primHi, m_bounds);
  BVH_CHECK(buildFromBoxes(stream, numTriangles, outNodes, outWide, nodeBase, 0, triangleBase, soup ? TWK_LEAF_WORLD : 0));
  hipLaunchKernelGGL(emitTrianglesKernel, dim3(grid), dim3(block), 0, stream, attributes, indices, soup, instances, m_keysOut, numTriangles, outTriangles + 3 * (size_t) triangleBase, outShadeTriangles + TWK_SHADE_RECORD * (size_t) triangleBase);
  BVH_CHECK(hipGetLastError());
  float4 lo, hi;
  BVH_CHECK(hipMemcpyAsync(&lo, m_nodeLo, sizeof(float4), hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipMemcpyAsync(&hi, m_nodeHi, sizeof(float4), hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipStreamSynchronize(stream));
  rootBounds[0] = lo.x; rootBounds[1] = lo.y; rootBounds[2] = lo.z;
  rootBounds[3] = hi.x; rootBounds[4] = hi.y; rootBounds[5] = hi.z;
  { int h; memcpy(&h, &lo.w, sizeof(h)); m_lastHeight = h; }
  return hipSuccess;
}

void BvhBuilder::soupDescriptors(hipStream_t stream, int4* soup, int first, int count, int instance, int attributeBase, int indexBase)
{
  hipLaunchKernelGGL(soupDescriptorsKernel, dim3((count + 255) / 256), dim3(256), 0, stream, soup, first, count, instance, attributeBase, indexBase);
}

hipError_t BvhBuilder::buildInstances(hipStream_t stream, const float4* hostLo, const float4* hostHi, const int* hostLeafPayload, int numInstances, BvhNode* outNodes, BvhNode* outWide, int nodeBase)
{
  BVH_CHECK(reserve(numInstances));
  const int block = 256;
  const int grid  = (numInstances + block - 1) / block;
  BVH_CHECK(hipMemcpyAsync(m_primLo, hostLo, sizeof(float4) * numInstances, hipMemcpyHostToDevice, stream));
  BVH_CHECK(hipMemcpyAsync(m_primHi, hostHi, sizeof(float4) * numInstances, hipMemcpyHostToDevice, stream));
  BVH_CHECK(hipMemcpyAsync(m_leafPayload, hostLeafPayload, sizeof(int) * numInstances, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(initBoundsKernel, dim3(1), dim3(64), 0, stream, m_bounds);
  hipLaunchKernelGGL(boxBoundsKernel, dim3(grid), dim3(block), 0, stream, m_primLo, m_primHi, numInstances, m_bounds);
  BVH_CHECK(buildFromBoxes(stream, numInstances, outNodes, outWide, nodeBase, 1, 0, 0));
  float4 lo;
  BVH_CHECK(hipMemcpyAsync(&lo, m_nodeLo, sizeof(float4), hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipStreamSynchronize(stream)); // hostLo/hostHi may go away
  { int h; memcpy(&h, &lo.w, sizeof(h)); m_lastHeight = h; }
  return hipSuccess;
}

} // namespace twk
