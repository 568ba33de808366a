// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// optixTrace stand-in. The reference has NO source for this half: BVH build, traversal and the
// ray-triangle test live in NVIDIA's closed libnvoptix.so.1 (OptiX 7.0.0, loaded at
// src/Device.cpp:519-535) — PARITY UNPINNED at that boundary. What is pinned is the observable
// contract, restated here from the call sites:
//   * radiance ray: closest hit in (tmin, tmax) = (sceneEpsilon, 1e27), all instances visible, no
//     face culling                                   — shaders/raygeneration.cu:84-89, Device.cpp:1378,1438
//   * shadow ray: any hit in (eps, dist - eps)       — shaders/closesthit.cu:281-286, anyhit.cu:84-91
//   * geometry: float3 positions at stride 48 B, uint3 indices — Device.cpp:1362-1381
//   * instance: row-major 3x4 object→world, ray taken to object space, t preserved — Device.cpp:1434
//   * barycentrics (beta, gamma) weight vertex 1 and 2 — shaders/closesthit.cu:142-147
//   * flattened instances (include/tweeker_hip.h twk_set_flatten_policy: geometry with at most maxTriangles
//     triangles, or referenced by at most maxReferences instances): intersected in WORLD space — the vertices are
//     taken to world space once (row-major 3x4, m0*x + m1*y + m2*z + m3 in fp32, as transformPoint
//     closesthit.cu:88-98 evaluates it) and tested against the untransformed ray. t, beta, gamma are the same
//     quantities as in object space (t is preserved by the affine map); which space the closed OptiX traversal rounds
//     in is not observable from the reference, so the policy is part of this build's definition of the contract.
// The triangle test is the published watertight algorithm of Woop, Benthin, Wald (JCGT 2013) in
// single precision with the double-precision fallback on zero edge functions, every operation
// spelled out so that the HIP kernel can be compared bit for bit. Ties in t resolve to the smaller
// (instance, primitive) pair so that brute force, this file's BVH and the device BVH agree.
//
// Two search modes: brute force over every triangle of every instance (the definition), and a
// median-split BVH per geometry with a conservative slab test (must return identical results;
// tests/test_oracle_trace.py checks that).
#pragma once
#include "orc_types.h"
#include "../include/tweeker_hip.h" // TWK_FLATTEN_TRIANGLES
#include <algorithm>
#include <cfloat>

namespace orc {

struct Hit
{
  float t;
  float beta, gamma;
  int   instance; // -1 = miss
  int   primitive;
};

struct TraceCounters
{
  uint64_t rays = 0, boxTests = 0, triTests = 0;
};

// Per-thread tally of the counters above (orc_render_rect_threads runs rows on several threads).
inline TraceCounters& traceTally() { static thread_local TraceCounters t{}; return t; }

struct BvhNode // oracle-private layout, not the device layout
{
  float lo[3], hi[3];
  int   left;  // internal: index of left child, right = left + 1 ... stored explicitly below
  int   right;
  int   first, count; // leaf when count > 0
};

struct Geometry
{
  std::vector<TriangleAttributes> attributes;
  std::vector<unsigned int>       indices;
  std::vector<BvhNode>            nodes;
  std::vector<int>                order; // primitive ids in leaf order
};

struct Instance
{
  int   geometry;
  float objectToWorld[12];
  float worldToObject[12];
  int   material;
  int   light;
  float lo[3], hi[3]; // world AABB of the transformed vertices
  int   worldGeometry = -1; // flattened instance: index into Scene::worldGeometries (its vertices in world space + a BVH over them)
};

// Inverse of a row-major 3x4 affine matrix, evaluated in double and rounded once.
// (OptiX derives this internally for optixGetInstanceInverseTransformFromHandle — closesthit.cu:49-52.)
static inline void invertAffine(const float m[12], float inv[12])
{
  const double a00 = m[0], a01 = m[1], a02 = m[2],  t0 = m[3];
  const double a10 = m[4], a11 = m[5], a12 = m[6],  t1 = m[7];
  const double a20 = m[8], a21 = m[9], a22 = m[10], t2 = m[11];
  const double c00 = a11 * a22 - a12 * a21;
  const double c01 = a12 * a20 - a10 * a22;
  const double c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double r = 1.0 / det;
  const double i00 = c00 * r, i01 = (a02 * a21 - a01 * a22) * r, i02 = (a01 * a12 - a02 * a11) * r;
  const double i10 = c01 * r, i11 = (a00 * a22 - a02 * a20) * r, i12 = (a02 * a10 - a00 * a12) * r;
  const double i20 = c02 * r, i21 = (a01 * a20 - a00 * a21) * r, i22 = (a00 * a11 - a01 * a10) * r;
  inv[0] = (float) i00; inv[1] = (float) i01; inv[2]  = (float) i02; inv[3]  = (float) -(i00 * t0 + i01 * t1 + i02 * t2);
  inv[4] = (float) i10; inv[5] = (float) i11; inv[6]  = (float) i12; inv[7]  = (float) -(i10 * t0 + i11 * t1 + i12 * t2);
  inv[8] = (float) i20; inv[9] = (float) i21; inv[10] = (float) i22; inv[11] = (float) -(i20 * t0 + i21 * t1 + i22 * t2);
}

// closesthit.cu:88-98 (transformPoint) / :101-111 (transformVector) on a float[12]
static inline float3 xfmPoint(const float* m, const float3& v)
{
  return { m[0] * v.x + m[1] * v.y + m[2]  * v.z + m[3],
           m[4] * v.x + m[5] * v.y + m[6]  * v.z + m[7],
           m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] };
}
static inline float3 xfmVector(const float* m, const float3& v)
{
  return { m[0] * v.x + m[1] * v.y + m[2]  * v.z,
           m[4] * v.x + m[5] * v.y + m[6]  * v.z,
           m[8] * v.x + m[9] * v.y + m[10] * v.z };
}

// Per (ray, instance) constants of the watertight test.
struct WoopRay
{
  float o[3], d[3];
  int   kx, ky, kz;
  float Sx, Sy, Sz;
};

static inline void woopSetup(const float3& o, const float3& d, WoopRay& r)
{
  r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z;
  r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z;
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  int kx = (kz == 2) ? 0 : kz + 1;
  int ky = (kx == 2) ? 0 : kx + 1;
  if (r.d[kz] < 0.0f) { const int s = kx; kx = ky; ky = s; }
  r.kx = kx; r.ky = ky; r.kz = kz;
  r.Sx = r.d[kx] / r.d[kz];
  r.Sy = r.d[ky] / r.d[kz];
  r.Sz = 1.0f / r.d[kz];
}

// Returns true and fills t/beta/gamma when the triangle is hit with tmin < t < tmax.
static inline bool woopIntersect(const WoopRay& r, const float3& p0, const float3& p1, const float3& p2,
                                 float tmin, float tmax, float& t, float& beta, float& gamma)
{
  const float v0[3] = {p0.x, p0.y, p0.z}, v1[3] = {p1.x, p1.y, p1.z}, v2[3] = {p2.x, p2.y, p2.z};
  const float Akx = v0[r.kx] - r.o[r.kx], Aky = v0[r.ky] - r.o[r.ky], Akz = v0[r.kz] - r.o[r.kz];
  const float Bkx = v1[r.kx] - r.o[r.kx], Bky = v1[r.ky] - r.o[r.ky], Bkz = v1[r.kz] - r.o[r.kz];
  const float Ckx = v2[r.kx] - r.o[r.kx], Cky = v2[r.ky] - r.o[r.ky], Ckz = v2[r.kz] - r.o[r.kz];

  const float Ax = Akx - r.Sx * Akz, Ay = Aky - r.Sy * Akz;
  const float Bx = Bkx - r.Sx * Bkz, By = Bky - r.Sy * Bkz;
  const float Cx = Ckx - r.Sx * Ckz, Cy = Cky - r.Sy * Ckz;

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

  const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
  const float T = U * Az + V * Bz + W * Cz;
  const float rcpDet = 1.0f / det;
  const float tt = T * rcpDet;
  if (!(tt > tmin && tt < tmax)) return false;

  t     = tt;
  beta  = V * rcpDet;
  gamma = W * rcpDet;
  return true;
}

class Scene
{
public:
  std::vector<Geometry> geometries;
  std::vector<Instance> instances;
  std::vector<Geometry> worldGeometries; // one per flattened instance, see prepare()
  int  flattenMaxTriangles = TWK_FLATTEN_TRIANGLES, flattenMaxReferences = TWK_FLATTEN_REFERENCES;
  bool dirty = true;
  bool useBvh = true;
  mutable TraceCounters counters; // totals; the hot paths tally into traceTally() (thread local) and the render entry points merge

  int addGeometry(const TriangleAttributes* attr, size_t numAttr, const unsigned int* idx, size_t numIdx)
  {
    Geometry g;
    g.attributes.assign(attr, attr + numAttr);
    g.indices.assign(idx, idx + numIdx);
    geometries.push_back(std::move(g));
    buildBvh(geometries.back());
    return (int) geometries.size() - 1;
  }

  int addInstance(int geometry, const float m[12], int material, int light)
  {
    Instance inst;
    inst.geometry = geometry;
    for (int i = 0; i < 12; ++i) inst.objectToWorld[i] = m[i];
    invertAffine(m, inst.worldToObject);
    inst.material = material;
    inst.light = light;
    for (int k = 0; k < 3; ++k) { inst.lo[k] = FLT_MAX; inst.hi[k] = -FLT_MAX; }
    const Geometry& g = geometries[geometry];
    for (unsigned int i : g.indices)
    {
      const float3 w = xfmPoint(m, g.attributes[i].vertex);
      const float c[3] = {w.x, w.y, w.z};
      for (int k = 0; k < 3; ++k) { inst.lo[k] = std::min(inst.lo[k], c[k]); inst.hi[k] = std::max(inst.hi[k], c[k]); }
    }
    // Pad: the object-space test runs on a transformed ray, rounding differs from the world-space box.
    for (int k = 0; k < 3; ++k)
    {
      const float e = 1.0e-4f * std::max(1.0f, std::max(fabsf(inst.lo[k]), fabsf(inst.hi[k])));
      inst.lo[k] -= e; inst.hi[k] += e;
    }
    instances.push_back(inst);
    dirty = true;
    return (int) instances.size() - 1;
  }

  void clear() { geometries.clear(); instances.clear(); worldGeometries.clear(); dirty = true; }

  void setFlattenPolicy(int maxTriangles, int maxReferences) { flattenMaxTriangles = maxTriangles; flattenMaxReferences = maxReferences; dirty = true; }

  // Decides which instances are flattened (twk_set_flatten_policy's rule) and gives each of them its world-space
  // vertices. Called by the entry points before anything is traced.
  void prepare()
  {
    if (!dirty) return;
    worldGeometries.clear();
    std::vector<int> references(geometries.size(), 0);
    for (const Instance& inst : instances) references[inst.geometry]++;
    for (Instance& inst : instances)
    {
      const Geometry& g = geometries[inst.geometry];
      inst.worldGeometry = -1;
      if ((int) (g.indices.size() / 3) > flattenMaxTriangles && references[inst.geometry] > flattenMaxReferences) continue;
      Geometry w;
      w.attributes = g.attributes;
      w.indices = g.indices;
      for (TriangleAttributes& a : w.attributes) a.vertex = xfmPoint(inst.objectToWorld, a.vertex);
      buildBvh(w);
      inst.worldGeometry = (int) worldGeometries.size();
      worldGeometries.push_back(std::move(w));
    }
    dirty = false;
  }

  // Closest hit (anyHit == false) or first-found occlusion (anyHit == true).
  Hit trace(const float3& origin, const float3& direction, float tmin, float tmax, bool anyHit) const
  {
    Hit best; best.t = tmax; best.beta = best.gamma = 0.0f; best.instance = -1; best.primitive = -1;
    traceTally().rays++;
    WoopRay wrWorld; woopSetup(origin, direction, wrWorld);
    for (int ii = 0; ii < (int) instances.size(); ++ii)
    {
      const Instance& inst = instances[ii];
      if (useBvh && !slab(inst.lo, inst.hi, origin, direction, tmin, best.t)) continue;
      if (inst.worldGeometry >= 0)
      {
        // flattened instance: world-space vertices against the untransformed ray (see the header)
        const Geometry& w = worldGeometries[inst.worldGeometry];
        if (useBvh) traverse(w, ii, wrWorld, origin, direction, tmin, anyHit, best);
        else
        {
          const int numPrims = (int) (w.indices.size() / 3);
          for (int p = 0; p < numPrims; ++p) test(w, ii, p, wrWorld, tmin, best);
        }
        if (anyHit && best.instance >= 0) return best;
        continue;
      }
      const float3 o = xfmPoint(inst.worldToObject, origin);
      const float3 d = xfmVector(inst.worldToObject, direction);
      WoopRay wr; woopSetup(o, d, wr);
      const Geometry& g = geometries[inst.geometry];
      if (useBvh) traverse(g, ii, wr, o, d, tmin, anyHit, best);
      else
      {
        const int numPrims = (int) (g.indices.size() / 3);
        for (int p = 0; p < numPrims; ++p) test(g, ii, p, wr, tmin, best);
      }
      if (anyHit && best.instance >= 0) return best;
    }
    return best;
  }

private:
  void test(const Geometry& g, int ii, int p, const WoopRay& wr, float tmin, Hit& best) const
  {
    traceTally().triTests++;
    const unsigned int i0 = g.indices[3 * p], i1 = g.indices[3 * p + 1], i2 = g.indices[3 * p + 2];
    float t, b, c;
    // tmax passed as +inf-like bound; the commit rule below implements "t < best, ties to the smaller id".
    if (!woopIntersect(wr, g.attributes[i0].vertex, g.attributes[i1].vertex, g.attributes[i2].vertex, tmin, RT_DEFAULT_MAX * 2.0f, t, b, c)) return;
    const bool closer = (t < best.t) ||
                        (t == best.t && best.instance >= 0 && (ii < best.instance || (ii == best.instance && p < best.primitive)));
    if (closer) { best.t = t; best.beta = b; best.gamma = c; best.instance = ii; best.primitive = p; }
  }

  // Conservative slab test: true when [tnear, tfar] may overlap (tmin, tbest].
  bool slab(const float lo[3], const float hi[3], const float3& o, const float3& d, float tmin, float tbest) const
  {
    traceTally().boxTests++;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    float tn = tmin, tf = tbest;
    for (int k = 0; k < 3; ++k)
    {
      if (dd[k] == 0.0f)
      {
        const float pad = 1.0e-5f * std::max(1.0f, std::max(fabsf(lo[k]), fabsf(hi[k])));
        if (oo[k] < lo[k] - pad || oo[k] > hi[k] + pad) return false;
        continue;
      }
      const double inv = 1.0 / (double) dd[k];
      double t0 = ((double) lo[k] - (double) oo[k]) * inv;
      double t1 = ((double) hi[k] - (double) oo[k]) * inv;
      if (t0 > t1) std::swap(t0, t1);
      t0 -= 1.0e-6 * std::max(1.0, fabs(t0));
      t1 += 1.0e-6 * std::max(1.0, fabs(t1));
      if (t0 > tn) tn = (float) std::min(t0, (double) FLT_MAX);
      if (t1 < tf) tf = (float) std::max(t1, (double) -FLT_MAX);
      if (tn > tf) return false;
    }
    return true;
  }

  void traverse(const Geometry& g, int ii, const WoopRay& wr, const float3& o, const float3& d, float tmin, bool anyHit, Hit& best) const
  {
    if (g.nodes.empty()) return;
    int stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp > 0)
    {
      const BvhNode& n = g.nodes[stack[--sp]];
      if (!slab(n.lo, n.hi, o, d, tmin, best.t)) continue;
      if (n.count > 0)
      {
        for (int k = 0; k < n.count; ++k) test(g, ii, g.order[n.first + k], wr, tmin, best);
        if (anyHit && best.instance >= 0) return;
      }
      else { stack[sp++] = n.right; stack[sp++] = n.left; }
    }
  }

  static void buildBvh(Geometry& g)
  {
    const int numPrims = (int) (g.indices.size() / 3);
    g.nodes.clear(); g.order.resize(numPrims);
    if (numPrims == 0) return;
    std::vector<float> lo(3 * numPrims), hi(3 * numPrims), ce(3 * numPrims);
    for (int p = 0; p < numPrims; ++p)
    {
      g.order[p] = p;
      for (int k = 0; k < 3; ++k) { lo[3 * p + k] = FLT_MAX; hi[3 * p + k] = -FLT_MAX; }
      for (int v = 0; v < 3; ++v)
      {
        const float3& q = g.attributes[g.indices[3 * p + v]].vertex;
        const float c[3] = {q.x, q.y, q.z};
        for (int k = 0; k < 3; ++k) { lo[3 * p + k] = std::min(lo[3 * p + k], c[k]); hi[3 * p + k] = std::max(hi[3 * p + k], c[k]); }
      }
      for (int k = 0; k < 3; ++k) ce[3 * p + k] = 0.5f * (lo[3 * p + k] + hi[3 * p + k]);
      // The triangle test's t carries an absolute error that scales with the triangle's size and coordinates, not
      // with t (a ray that starts 2e-4 above a 16-unit floor triangle gets t off by 3e-7): grow every primitive box on
      // all axes by 2^-17 of (largest coordinate magnitude + diagonal) so that this BVH never culls what brute force
      // — the definition — would report.
      float m = 0.0f, d2 = 0.0f;
      for (int k = 0; k < 3; ++k)
      {
        m = std::max(m, std::max(fabsf(lo[3 * p + k]), fabsf(hi[3 * p + k])));
        const float d = hi[3 * p + k] - lo[3 * p + k];
        d2 += d * d;
      }
      const float e = 7.6293945e-6f * (m + sqrtf(d2)) + 1.0e-30f;
      for (int k = 0; k < 3; ++k) { lo[3 * p + k] -= e; hi[3 * p + k] += e; }
    }
    g.nodes.reserve(2 * numPrims);
    struct Job { int node, first, count; };
    std::vector<Job> jobs;
    g.nodes.push_back(BvhNode());
    jobs.push_back({0, 0, numPrims});
    while (!jobs.empty())
    {
      const Job job = jobs.back(); jobs.pop_back();
      BvhNode n;
      for (int k = 0; k < 3; ++k) { n.lo[k] = FLT_MAX; n.hi[k] = -FLT_MAX; }
      float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (int i = job.first; i < job.first + job.count; ++i)
      {
        const int p = g.order[i];
        for (int k = 0; k < 3; ++k)
        {
          n.lo[k] = std::min(n.lo[k], lo[3 * p + k]); n.hi[k] = std::max(n.hi[k], hi[3 * p + k]);
          clo[k] = std::min(clo[k], ce[3 * p + k]);   chi[k] = std::max(chi[k], ce[3 * p + k]);
        }
      }
      n.left = n.right = -1; n.first = job.first; n.count = 0;
      int axis = 0;
      if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
      if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
      if (job.count <= 4 || !(chi[axis] > clo[axis]))
      {
        n.count = job.count;
      }
      else
      {
        const int mid = job.first + job.count / 2;
        std::nth_element(g.order.begin() + job.first, g.order.begin() + mid, g.order.begin() + job.first + job.count,
                         [&](int a, int b) { return ce[3 * a + axis] < ce[3 * b + axis] || (ce[3 * a + axis] == ce[3 * b + axis] && a < b); });
        n.left  = (int) g.nodes.size(); g.nodes.push_back(BvhNode());
        n.right = (int) g.nodes.size(); g.nodes.push_back(BvhNode());
        jobs.push_back({n.left, job.first, mid - job.first});
        jobs.push_back({n.right, mid, job.first + job.count - mid});
      }
      g.nodes[job.node] = n;
    }
  }
};

} // namespace orc
