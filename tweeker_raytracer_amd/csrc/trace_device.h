// Device functions of ray traversal, shared by the traversal kernels (trace_persistent.h, trace_kernels.hip) and the host build of the kernels: conservative slab test,
// watertight triangle test (Woop, Benthin, Wald, JCGT 2013), and the single-ray two-level traversal whose stack
// continues from LDS into HBM. See trace_kernels.hip for the contract these stand in for (optixTrace).
#pragma once
#include "device_types.h"

namespace twk {


struct TraceRay
{
  V3 o, d, id, ood; // origin, direction, guarded reciprocal direction, origin * reciprocal direction
};

TWK_D float guardedReciprocal(float d)
{
  // Parallel-to-slab rays: a huge finite reciprocal keeps 0 * inf = NaN out of the slab test and
  // makes "origin on the slab plane" count as inside (conservative).
  // v_rcp_f32 (1 ulp) instead of the correctly rounded division: this reciprocal only feeds the culling test, whose
  // 2.5e-6 relative widening dwarfs the error; the divisions that are part of the RESULT (Woop shear constants,
  // 1 / det) stay IEEE.
  return (fabsf(d) >= 1.0e-20f) ? __builtin_amdgcn_rcpf(d) : copysignf(1.0e20f, d);
}

TWK_D void setupRay(TraceRay& r, const V3& o, const V3& d)
{
  r.o = o; r.d = d;
  r.id  = v3(guardedReciprocal(d.x), guardedReciprocal(d.y), guardedReciprocal(d.z));
  r.ood = v3(o.x * r.id.x, o.y * r.id.y, o.z * r.id.z);
}

// Conservative slab test of one child box: plane distances as one fused multiply-add each (the box test only
// culls, its rounding is not part of the result; the 2.5e-6 relative widening covers fma-vs-exact differences,
// and boxes are padded at build time). Returns the entry distance for near/far ordering.
TWK_D bool slabTest(const TraceRay& r, float lox, float loy, float loz, float hix, float hiy, float hiz, float tmin, float tmax, float& tnear)
{
  const float x0 = __builtin_fmaf(lox, r.id.x, -r.ood.x), x1 = __builtin_fmaf(hix, r.id.x, -r.ood.x);
  const float y0 = __builtin_fmaf(loy, r.id.y, -r.ood.y), y1 = __builtin_fmaf(hiy, r.id.y, -r.ood.y);
  const float z0 = __builtin_fmaf(loz, r.id.z, -r.ood.z), z1 = __builtin_fmaf(hiz, r.id.z, -r.ood.z);
  const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
  const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
  tnear = tn;
  return tn * 0.9999975f <= tf * 1.0000025f;
}

// The same test on a quantised wide node (device_types.h): the planes of a child box are grid coordinates q of the
// node's own box, plane distance = q * a + b with a = cell / d and b = (origin - o) / d per axis; qn / qf are the
// coordinates of the plane the ray meets first / last on each axis (chosen by the caller from the sign of d). Three
// roundings instead of one per plane; the build pads every box by 2^-17 of the scene scale (bvh_build.hip padBox),
// which dwarfs them.
TWK_D bool slabTestGrid(float ax, float ay, float az, float bx, float by, float bz,
                        float qnx, float qny, float qnz, float qfx, float qfy, float qfz, float tmin, float tmax, float& tnear)
{
  const float tn = fmaxf(fmaxf(__builtin_fmaf(qnx, ax, bx), __builtin_fmaf(qny, ay, by)), fmaxf(__builtin_fmaf(qnz, az, bz), tmin));
  const float tf = fminf(fminf(__builtin_fmaf(qfx, ax, bx), __builtin_fmaf(qfy, ay, by)), fminf(__builtin_fmaf(qfz, az, bz), tmax));
  tnear = tn;
  // the same widening as slabTest's tn * (1 - 2.5e-6) <= tf * (1 + 2.5e-6), as ONE product: tn >= tmin >= 0 here, so dividing by
  // the left factor keeps the direction, and 1.0000051 > (1 + 2.5e-6) / (1 - 2.5e-6) keeps it conservative (four multiplies
  // fewer per node step of the issue-bound traversal kernel)
  return tn <= tf * 1.0000051f;
}

// Woop-Benthin-Wald ray constants. The axis permutation (kx, ky, kz) is a cyclic shift of (x, y, z) chosen by the
// dominant direction axis kz, with kx and ky exchanged when d[kz] < 0; it is kept as three flag bits in one register
// (lane flags held as bools live in scalar masks and cost scalar merge instructions at the end of every divergent
// region) and applied with selects. (Indexing the components with integer kx / ky / kz — `k == 0 ? x : k == 1 ? y : z` — is turned into a
// switch by the compiler and then into ~20 scalar exec-mask instructions and two branches PER COMPONENT: the triangle
// test was 490 instructions long, most of them scalar.)
struct WoopConstants
{
  unsigned int perm; // bit 0: kz == 0, bit 1: kz == 1
  float Sx, Sy, Sz;
};

struct WoopPermutation { bool zIsX, zIsY; };
TWK_D WoopPermutation woopFlags(unsigned int perm) { WoopPermutation f; f.zIsX = (perm & 1u) != 0u; f.zIsY = (perm & 2u) != 0u; return f; }

// (v[kx], v[ky], v[kz]) of the cyclic permutation described by w.
TWK_D void woopPermute(const WoopPermutation& w, const V3& v, float& vx, float& vy, float& vz)
{
  vx = w.zIsX ? v.y : (w.zIsY ? v.z : v.x);
  vy = w.zIsX ? v.z : (w.zIsY ? v.x : v.y);
  vz = w.zIsX ? v.x : (w.zIsY ? v.y : v.z);
}

// The paper (and the CPU restatement the parity tests compare with) exchanges kx and ky when d[kz] < 0 to keep the winding of the sheared
// triangle. Nothing here culls by winding, and the exchange changes no result bit: with x and y exchanged every edge
// function is the exact negation (U = Cx*By - Cy*Bx becomes Cy*Bx - Cx*By, also in the double-precision fallback), so
// det and T change sign together and t = T / det, beta = V / det, gamma = W / det and every sign test come out the
// same. It is left out: six selects per triangle test (selects cost 1.6 x an fma, tools/probes/valu_issue_probe.hip).
TWK_D void woopSetup(const V3& d, WoopConstants& w)
{
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  // kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2)
  WoopPermutation f;
  f.zIsX = (ax > ay) & (ax > az);
  f.zIsY = !(ax > ay) & (ay > az);
  float dx, dy, dz;
  woopPermute(f, d, dx, dy, dz);
  w.perm = (f.zIsX ? 1u : 0u) | (f.zIsY ? 2u : 0u);
  w.Sx = dx / dz;
  w.Sy = dy / dz;
  w.Sz = 1.0f / dz;
}

// Straight-line: every lane runs the whole test and the verdict is one flag (a wave leaves early only when all of its
// lanes fail, which with ~20 active lanes practically never happens, while each early return costs exec-mask
// bookkeeping for everyone). The arithmetic of an accepted hit is unchanged.
TWK_D bool woopIntersect(const WoopConstants& w, const V3& o, const V3& p0, const V3& p1, const V3& p2,
                         float tmin, float& t, float& beta, float& gamma)
{
  const V3 A = p0 - o, B = p1 - o, C = p2 - o;
  const WoopPermutation f = woopFlags(w.perm);
  float Akx, Aky, Akz, Bkx, Bky, Bkz, Ckx, Cky, Ckz;
  woopPermute(f, A, Akx, Aky, Akz);
  woopPermute(f, B, Bkx, Bky, Bkz);
  woopPermute(f, C, Ckx, Cky, Ckz);

  const float Ax = Akx - w.Sx * Akz, Ay = Aky - w.Sy * Akz;
  const float Bx = Bkx - w.Sx * Bkz, By = Bky - w.Sy * Bkz;
  const float Cx = Ckx - w.Sx * Ckz, Cy = Cky - w.Sy * Ckz;

  float U = Cx * By - Cy * Bx;
  float V = Ax * Cy - Ay * Cx;
  float W = Bx * Ay - By * Ax;

  if ((U == 0.0f) | (V == 0.0f) | (W == 0.0f)) // rare: an edge function vanished in float, decide it in double
  {
    const double CxBy = (double) Cx * (double) By, CyBx = (double) Cy * (double) Bx;
    U = (float) (CxBy - CyBx);
    const double AxCy = (double) Ax * (double) Cy, AyCx = (double) Ay * (double) Cx;
    V = (float) (AxCy - AyCx);
    const double BxAy = (double) Bx * (double) Ay, ByAx = (double) By * (double) Ax;
    W = (float) (BxAy - ByAx);
  }

  const bool mixedSigns = ((U < 0.0f) | (V < 0.0f) | (W < 0.0f)) & ((U > 0.0f) | (V > 0.0f) | (W > 0.0f));
  const float det = U + V + W;
  const float Az = w.Sz * Akz, Bz = w.Sz * Bkz, Cz = w.Sz * Ckz;
  const float T = U * Az + V * Bz + W * Cz;
  const float rcpDet = 1.0f / det;
  const float tt = T * rcpDet;
  t     = tt;
  beta  = V * rcpDet;
  gamma = W * rcpDet;
  return !mixedSigns & (det != 0.0f) & (tt > tmin);
}

struct TraceResult
{
  float t, beta, gamma;
  int   instance, primitive;
  int   triangleSlot; // slot of the hit triangle in the leaf-ordered arrays (what shading indexes)
};

// COUNT: tally node / triangle / instance visits (measurement builds only).
template<bool COUNT>
TWK_D void traverse(const LaunchParams& p, const V3& org, const V3& dir, float tmin, float tmax, bool anyHit,
                    int* ldsStack /* [entry * blockDim + tid] base at tid */, int* spill, TraceResult& res,
                    unsigned int& nodeCount, unsigned int& triCount, unsigned int& instCount)
{
  const int stride = TWK_TRACE_BLOCK;
  res.t = tmax; res.beta = 0.0f; res.gamma = 0.0f; res.instance = -1; res.primitive = -1; res.triangleSlot = -1;

  TraceRay ray;
  setupRay(ray, org, dir);
  WoopConstants woopWorld, woop; // world-space constants (flattened instances), current space
  woopSetup(dir, woopWorld);
  woop = woopWorld;
  V3 objOrg = org;
  int currentInstance = -1;

  int sp = 0;
  int node = p.tlasRoot;
  unsigned int guard = 0; // a well-formed tree never gets near this; keeps a corrupted one from hanging the GPU

  // A push beyond LDS + HBM capacity (20 + 72 entries of a binary traversal: a tree more than ~90 levels deep along one
  // path) cannot be stored; the pop then yields the sentinel and the subtree is lost. twk_build refuses scenes that deep
  // (device_api.hip: traversalDepth), so this cannot happen on a built scene; should it ever, the push is counted in a
  // word of pinned host memory whether or not statistics are on, and the next twk_sync / twk_read_output fails.
#define TWK_PUSH(v) do { if (sp < TWK_TRACE_STACK_LDS) ldsStack[sp * stride] = (v); else if (sp < TWK_TRACE_STACK_LDS + TWK_TRACE_STACK_SPILL) spill[sp - TWK_TRACE_STACK_LDS] = (v); else __hip_atomic_fetch_add(p.droppedPushes, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ++sp; } while (0)
#define TWK_POP(v)  do { --sp; (v) = (sp < TWK_TRACE_STACK_LDS) ? ldsStack[sp * stride] : ((sp < TWK_TRACE_STACK_LDS + TWK_TRACE_STACK_SPILL) ? spill[sp - TWK_TRACE_STACK_LDS] : TWK_BVH_SENTINEL); } while (0)

  for (;;)
  {
    if (++guard > (1u << 22)) break;
    if (node == TWK_BVH_SENTINEL)
    {
      // leaving an instance: back to the world-space ray
      setupRay(ray, org, dir);
      woop = woopWorld; objOrg = org;
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
    if (currentInstance < 0 && !(payload & TWK_LEAF_WORLD))
    {
      // top level: enter the instance
      const float4* rec = reinterpret_cast<const float4*>(p.instances + payload);
      const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
      if (COUNT) ++instCount;
      float m[12];
      m[0] = r0.x; m[1] = r0.y; m[2] = r0.z; m[3] = r0.w;
      m[4] = r1.x; m[5] = r1.y; m[6] = r1.z; m[7] = r1.w;
      m[8] = r2.x; m[9] = r2.y; m[10] = r2.z; m[11] = r2.w;
      objOrg = transformPoint(m, org);
      const V3 objDir = transformVector(m, dir);
      woopSetup(objDir, woop);
      setupRay(ray, objOrg, objDir);
      currentInstance = payload;
      TWK_PUSH(TWK_BVH_SENTINEL);
      node = __float_as_int(r3.x);
      continue;
    }

    // a leaf of 1..4 consecutive triangle slots: of the current instance's geometry (object space), or the world-space
    // slots of a flattened instance at the top level (instance index in the slot)
    {
      const int first = payload & 0x0fffffff, last = first + ((payload >> 28) & 3);
      bool stop = false;
      for (int slot = first; slot <= last; ++slot)
      {
        const float4* tri = p.triangles + 3 * (size_t) slot;
        const float4 a = tri[0], b = tri[1], c = tri[2];
        if (COUNT) ++triCount;
        float t, beta, gamma;
        if (woopIntersect(woop, objOrg, v3(a), v3(b), v3(c), tmin, t, beta, gamma))
        {
          const int prim = __float_as_int(a.w);
          const int triInstance = (currentInstance >= 0) ? currentInstance : __float_as_int(b.w);
          const bool closer = (t < res.t) ||
                              (t == res.t && res.instance >= 0 &&
                               (triInstance < res.instance || (triInstance == res.instance && prim < res.primitive)));
          if (closer)
          {
            res.t = t; res.beta = beta; res.gamma = gamma; res.instance = triInstance; res.primitive = prim; res.triangleSlot = slot;
            if (anyHit) { stop = true; break; }
          }
        }
      }
      if (stop || sp == 0) break;
      TWK_POP(node);
    }
  }
#undef TWK_PUSH
#undef TWK_POP
}

} // namespace twk
