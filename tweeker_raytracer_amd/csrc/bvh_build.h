#pragma once
#include "device_types.h"

namespace twk {

// Quantised copy of `count` wide nodes (128 B each at wide[2 * i]) into out[4 * i] (64 B each, device_types.h).
void launchQuantizeWide(const BvhNode* wide, float4* out, int count, hipStream_t stream);

// Copies the first numSlots (<= TWK_TOP_NODES) quantised wide nodes (breadth-first from `root`) into top[numSlots * 4] with the
// references among them rewritten to TWK_NODE_CACHED | slot (device_types.h).
void launchTopCache(const float4* wideQ, int root, int root2, float4* top, int numSlots, hipStream_t stream);
// The root as two wide nodes of up to eight entries together (bvh_build.hip wideRootKernel): written to the full-precision wide
// nodes firstNew, firstNew + 1 when it pays; result[0] (device) says whether.
void launchWideRoot(BvhNode* wide, int root, int firstNew, int* result, hipStream_t stream);

// Scratch-owning LBVH builder, reused for every geometry (bottom level) and for the instance level.
class BvhBuilder
{
public:
  ~BvhBuilder() { release(); }

  // outWide receives, per inner node, the 128-byte wide node (two BvhNode halves) at outWide[2 * localIndex].
  // Bottom level over the triangles of one geometry. Writes max(1, numTriangles - 1) nodes at
  // outNodes[0..] whose inner references are nodeBase-relative absolutes and numTriangles triangle
  // slots at outTriangles[3 * triangleBase ..] (+ the 128-byte shading records at outShadeTriangles[TWK_SHADE_RECORD * triangleBase ..]). rootBounds receives the (padded) object-space box.
  // soup != nullptr: the world-space soup of the flattened instances instead — numTriangles descriptors (bvh_build.hip
  // soupVertices), attributes / indices are the SHARED arrays, leaves carry TWK_LEAF_WORLD, rootBounds is a world box.
  hipError_t buildTriangles(hipStream_t stream, const float* attributes, const unsigned int* indices, int numTriangles,
                            BvhNode* outNodes, BvhNode* outWide, int nodeBase, float4* outTriangles, float4* outShadeTriangles, int triangleBase, float rootBounds[6],
                            const int4* soup = nullptr, const DevInstance* instances = nullptr);
  // Descriptors of one flattened instance: its geometry's triangles 0..count-1 become soup primitives first..first+count-1.
  void soupDescriptors(hipStream_t stream, int4* soup, int first, int count, int instance, int attributeBase, int indexBase);

  // Top level over boxes given on the host. Child reference of primitive k = ~hostLeafPayload[k]: an instance index
  // gives a leaf, ~(root node of the soup) splices the soup's tree in as a subtree.
  hipError_t buildInstances(hipStream_t stream, const float4* hostLo, const float4* hostHi, const int* hostLeafPayload, int numInstances, BvhNode* outNodes, BvhNode* outWide, int nodeBase);

  void release();
  // 0: Morton codes + Karras radix tree (LBVH); 1: binned-SAH top-down (bvh_sah.hip). Same refit / emission either way.
  void setQuality(int q) { m_quality = (q != 0) ? 1 : 0; }
  int  quality() const { return m_quality; }
  // SAH cost terms of the tree built last (half-area relative to the root, summed over the final tree's inner nodes /
  // over its leaf primitives), measured after every build.
  double lastSahInner() const { return m_lastSahInner; }
  double lastSahLeaf() const { return m_lastSahLeaf; }
  // Height of the binary tree built last (inner nodes on the longest root-to-leaf path, after the leaf collapse): what a
  // single-ray traversal may have to keep on its stack.
  int    lastHeight() const { return m_lastHeight; }
  void setMaxLeaf(int n) { m_maxLeaf = (n < 1) ? 1 : ((n > 4) ? 4 : n); } // count - 1 takes two bits of a leaf reference
  int    maxLeaf() const { return m_maxLeaf; }
  // Expected wide-node visits per node (bvh_build.hip refitKernel), one float per node of the scene's node array, absolute
  // index: with it every wide node takes the cheaper of its possible cuts; nullptr: the four grandchildren.
  void   setNodeCost(float* absolute) { m_nodeCost = absolute; }

private:
  hipError_t reserve(int count);
  hipError_t buildFromBoxes(hipStream_t stream, int count, BvhNode* outNodes, BvhNode* outWide, int nodeBase, int leafMode, int leafBase, int leafFlag);
  hipError_t reserveSah(int count);
  void releaseSah();
  hipError_t buildSahTopology(hipStream_t stream, int count);
  hipError_t accumulateSahCost(hipStream_t stream, int count);

  int m_capacity = 0;
  float4* m_primLo = nullptr; float4* m_primHi = nullptr;
  unsigned long long* m_keysIn = nullptr; unsigned long long* m_keysOut = nullptr;
  int* m_left = nullptr; int* m_right = nullptr; int* m_innerParent = nullptr; int* m_leafParent = nullptr;
  int2* m_range = nullptr;
  int m_maxLeaf = 2; // triangles per bottom-level leaf (1..4); measured on C2: 1 → 1156, 2 → 1215, 3 → 1172, 4 → 1106 Msamples/s
  unsigned int* m_tickets = nullptr;
  float4* m_nodeLo = nullptr; float4* m_nodeHi = nullptr;
  unsigned int* m_bounds = nullptr;
  int* m_leafPayload = nullptr; // top level: leaf payload per instance
  // binned-SAH builder scratch (bvh_sah.hip)
  int  m_quality = 1;
  int  m_sahCapacity = 0;
  int* m_sahOrder[2] = {nullptr, nullptr}; int* m_sahSlot[2] = {nullptr, nullptr};
  void* m_sahActive[2] = {nullptr, nullptr}; void* m_sahSmall = nullptr; void* m_sahSplit = nullptr;
  unsigned int* m_sahCb = nullptr; unsigned int* m_sahBins = nullptr;
  int* m_sahFlags = nullptr; int* m_sahPrefix = nullptr; void* m_sahChildCounts = nullptr; void* m_sahChildOffsets = nullptr;
  void* m_sahScanTemp = nullptr; size_t m_sahScanBytes = 0;
  int* m_sahCounters = nullptr; double* m_sahCost = nullptr;
  double m_lastSahInner = 0.0, m_lastSahLeaf = 0.0;
  int    m_lastHeight = 0;
  float* m_nodeCost = nullptr;
  void* m_sortTemp = nullptr; size_t m_sortBytes = 0;
};

} // namespace twk
