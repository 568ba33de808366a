// ORACLE-SIDE TEST INFRASTRUCTURE — not part of the product, never linked into libtweeker_hip.so.
//
// Same-BVH host walker (north_star: "a single-threaded C++ CPU fallback of the same kernels timed on the host cores";
// "per-ray visit counts taken from the CPU fallback running the same BVH on the same rays"): walks the
// acceleration structure the DEVICE built (read back through twk_debug_read_acceleration) with the per-ray algorithm
// of the persistent traversal kernel (tweeker_raytracer_amd/csrc/trace_kernels.hip): quantised 4-ary wide nodes (64 B,
// child boxes on the 8-bit grid of the node's own box, decoded with the kernel's float expressions), the four entry
// distances sorted by the same five compare-exchanges, nearest child next and the others pushed far to near,
// flattened world-space leaves tested with the untransformed ray, instances entered through the world-to-object
// matrix, the watertight Woop-Benthin-Wald triangle test with ties to the smaller (instance, primitive). It returns
// hit records — which must equal the device's bit for bit — and the exact visit counts, and is timed by bench.py on
// ONE host core as the CPU traversal baseline. The only arithmetic that differs from the device is the reciprocal
// of the culling test (v_rcp_f32 there, 1.0f / d here): it can move a borderline slab decision, i.e. the visit counts
// by a few in a million, never a hit.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline int asInt(float f) { int i; memcpy(&i, &f, 4); return i; }

const int SENTINEL = 0x7fffffff, LEAF_WORLD = 0x40000000;

struct Ray { V3 o, d, id, ood; };
inline float guardedReciprocal(float d) { return (fabsf(d) >= 1.0e-20f) ? 1.0f / d : copysignf(1.0e20f, d); }
inline void setupRay(Ray& r, const V3& o, const V3& d)
{
  r.o = o; r.d = d;
  r.id = v3(guardedReciprocal(d.x), guardedReciprocal(d.y), guardedReciprocal(d.z));
  r.ood = v3(o.x * r.id.x, o.y * r.id.y, o.z * r.id.z);
}
// trace_device.h slabTestGrid: plane distance of grid coordinate q = q * a + b, a = cell / d, b = (origin - o) / d;
// qn / qf = coordinates of the plane the ray meets first / last on each axis
inline bool slabTestGrid(const float a[3], const float b[3], const float qn[3], const float qf[3], float tmin, float tmax, float& tnear)
{
  const float tn = fmaxf(fmaxf(fmaf(qn[0], a[0], b[0]), fmaf(qn[1], a[1], b[1])), fmaxf(fmaf(qn[2], a[2], b[2]), tmin));
  const float tf = fminf(fminf(fmaf(qf[0], a[0], b[0]), fmaf(qf[1], a[1], b[1])), fminf(fmaf(qf[2], a[2], b[2]), tmax));
  tnear = tn;
  return tn <= tf * 1.0000051f; // trace_device.h slabTestGrid: the widening as one product
}
inline unsigned int asUint(float f) { unsigned int i; memcpy(&i, &f, 4); return i; }

struct Woop { int kx, ky, kz; float Sx, Sy, Sz; };
inline void woopSetup(const V3& d, Woop& w)
{
  const float dd[3] = {d.x, d.y, d.z};
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  int kx = (kz == 2) ? 0 : kz + 1;
  int ky = (kx == 2) ? 0 : kx + 1;
  if (dd[kz] < 0.0f) { const int s = kx; kx = ky; ky = s; }
  w.kx = kx; w.ky = ky; w.kz = kz;
  w.Sx = dd[kx] / dd[kz]; w.Sy = dd[ky] / dd[kz]; w.Sz = 1.0f / dd[kz];
}
inline bool woopIntersect(const Woop& w, const V3& o, const float* p0, const float* p1, const float* p2, float tmin, float& t, float& beta, float& gamma)
{
  const float oo[3] = {o.x, o.y, o.z};
  const float Akx = p0[w.kx] - oo[w.kx], Aky = p0[w.ky] - oo[w.ky], Akz = p0[w.kz] - oo[w.kz];
  const float Bkx = p1[w.kx] - oo[w.kx], Bky = p1[w.ky] - oo[w.ky], Bkz = p1[w.kz] - oo[w.kz];
  const float Ckx = p2[w.kx] - oo[w.kx], Cky = p2[w.ky] - oo[w.ky], Ckz = p2[w.kz] - oo[w.kz];
  const float Ax = Akx - w.Sx * Akz, Ay = Aky - w.Sy * Akz;
  const float Bx = Bkx - w.Sx * Bkz, By = Bky - w.Sy * Bkz;
  const float Cx = Ckx - w.Sx * Ckz, Cy = Cky - w.Sy * Ckz;
  float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
  if (U == 0.0f || V == 0.0f || W == 0.0f)
  {
    U = (float) ((double) Cx * (double) By - (double) Cy * (double) Bx);
    V = (float) ((double) Ax * (double) Cy - (double) Ay * (double) Cx);
    W = (float) ((double) Bx * (double) Ay - (double) By * (double) Ax);
  }
  const bool mixed = (U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f);
  const float det = U + V + W;
  const float Az = w.Sz * Akz, Bz = w.Sz * Bkz, Cz = w.Sz * Ckz;
  const float T = U * Az + V * Bz + W * Cz;
  const float rcpDet = 1.0f / det;
  t = T * rcpDet; beta = V * rcpDet; gamma = W * rcpDet;
  return !mixed && det != 0.0f && t > tmin;
}

} // namespace

extern "C" {

// rays: 8 floats each (o.xyz, tmin, d.xyz, tmax). out: t, beta, gamma per ray; ids: instance, primitive (-1 on a miss; any
// hit: ids[0] = 1 if occluded). counts: [0] wide nodes visited, [1] triangles tested, [2] instances entered, [3] deepest stack.
// root2 >= 0: the second node of an 8-wide root — a ray starts at `root` with `root2` on its stack (bvh_build.hip wideRootKernel).
int orc_walk_same_bvh(const float* wideNodes, int root, int root2, const float* triangles, const float* instances,
                      const float* rays, uint64_t numRays, int anyHit, float* tBetaGamma, int* ids, uint64_t counts[4])
{
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  std::vector<int> stack(4096);
  for (uint64_t i = 0; i < numRays; ++i)
  {
    const float* r = rays + 8 * i;
    const V3 org = v3(r[0], r[1], r[2]), dir = v3(r[4], r[5], r[6]);
    const float tmin = r[3];
    float bestT = r[7], bestBeta = 0.0f, bestGamma = 0.0f;
    int bestInstance = -1, bestPrimitive = -1;
    Ray ray; setupRay(ray, org, dir);
    Woop woopWorld; woopSetup(dir, woopWorld);
    Woop woop = woopWorld;
    int currentInstance = -1, node = root;
    size_t sp = 0;
    if (root2 >= 0) stack[sp++] = root2;
    bool done = false;
    uint64_t guard = 0;
    while (!done && ++guard < (1ull << 26))
    {
      if (node >= 0 && node != SENTINEL)
      {
        const float* w = wideNodes + 16 * (size_t) node;
        ++counts[0];
        float t[4]; int ref[4]; bool hit[4];
        int hits = 0;
        const float a[3] = {w[3] * ray.id.x, w[4] * ray.id.y, w[5] * ray.id.z};
        const float b[3] = {fmaf(w[0], ray.id.x, -ray.ood.x), fmaf(w[1], ray.id.y, -ray.ood.y), fmaf(w[2], ray.id.z, -ray.ood.z)};
        const unsigned int qlo[3] = {asUint(w[6]), asUint(w[7]), asUint(w[8])}, qhi[3] = {asUint(w[9]), asUint(w[10]), asUint(w[11])};
        const float idir[3] = {ray.id.x, ray.id.y, ray.id.z};
        for (int k = 0; k < 4; ++k)
        {
          ref[k] = asInt(w[12 + k]);
          float qn[3], qf[3];
          for (int c = 0; c < 3; ++c)
          {
            const float ql = (float) ((qlo[c] >> (8 * k)) & 0xffu), qh = (float) ((qhi[c] >> (8 * k)) & 0xffu);
            qn[c] = (idir[c] < 0.0f) ? qh : ql; qf[c] = (idir[c] < 0.0f) ? ql : qh;
          }
          hit[k] = slabTestGrid(a, b, qn, qf, tmin, bestT, t[k]); // an unused entry has an inverted box: never entered
          if (!hit[k]) t[k] = INFINITY;
          hits += hit[k] ? 1 : 0;
        }
        auto ce = [&](int a, int b) { if (t[b] < t[a]) { const float tt = t[a]; t[a] = t[b]; t[b] = tt; const int rr = ref[a]; ref[a] = ref[b]; ref[b] = rr; } };
        ce(0, 1); ce(2, 3); ce(0, 2); ce(1, 3); ce(1, 2);
        if (hits > 0)
        {
          node = ref[0];
          if (hits > 3) stack[sp++] = ref[3];
          if (hits > 2) stack[sp++] = ref[2];
          if (hits > 1) stack[sp++] = ref[1];
          if (sp + 8 > stack.size()) stack.resize(stack.size() * 2);
          if (sp > counts[3]) counts[3] = sp;
        }
        else
        {
          if (sp == 0) done = true; else node = stack[--sp];
        }
        continue;
      }
      bool pop = false;
      if (node == SENTINEL)
      {
        setupRay(ray, org, dir); woop = woopWorld; currentInstance = -1;
        pop = true;
      }
      else
      {
        const int payload = ~node;
        if (currentInstance < 0 && !(payload & LEAF_WORLD))
        {
          const float* m = instances + 32 * (size_t) payload; // world-to-object 3x4, then the BVH root
          ++counts[2];
          const V3 oo = v3(m[0] * org.x + m[1] * org.y + m[2] * org.z + m[3], m[4] * org.x + m[5] * org.y + m[6] * org.z + m[7], m[8] * org.x + m[9] * org.y + m[10] * org.z + m[11]);
          const V3 od = v3(m[0] * dir.x + m[1] * dir.y + m[2] * dir.z, m[4] * dir.x + m[5] * dir.y + m[6] * dir.z, m[8] * dir.x + m[9] * dir.y + m[10] * dir.z);
          woopSetup(od, woop);
          setupRay(ray, oo, od);
          currentInstance = payload;
          stack[sp++] = SENTINEL;
          node = asInt(m[12]);
        }
        else
        {
          const int first = payload & 0x0fffffff, last = first + ((payload >> 28) & 3);
          for (int ts = first; ts <= last; ++ts)
          {
            const float* tri = triangles + 12 * (size_t) ts;
            ++counts[1];
            float t, beta, gamma;
            if (!woopIntersect(woop, ray.o, tri, tri + 4, tri + 8, tmin, t, beta, gamma)) continue;
            const int prim = asInt(tri[3]);
            const int inst = (currentInstance >= 0) ? currentInstance : asInt(tri[7]);
            const bool closer = (t < bestT) || (t == bestT && bestInstance >= 0 && (inst < bestInstance || (inst == bestInstance && prim < bestPrimitive)));
            if (closer) { bestT = t; bestBeta = beta; bestGamma = gamma; bestInstance = inst; bestPrimitive = prim; if (anyHit) { done = true; break; } }
          }
          pop = true;
        }
      }
      if (pop && !done) { if (sp == 0) done = true; else node = stack[--sp]; }
    }
    if (anyHit)
    {
      tBetaGamma[3 * i] = tBetaGamma[3 * i + 1] = tBetaGamma[3 * i + 2] = 0.0f;
      ids[2 * i] = (bestInstance >= 0) ? 1 : 0; ids[2 * i + 1] = -1;
    }
    else
    {
      tBetaGamma[3 * i] = bestT; tBetaGamma[3 * i + 1] = bestBeta; tBetaGamma[3 * i + 2] = bestGamma;
      ids[2 * i] = bestInstance; ids[2 * i + 1] = bestPrimitive;
    }
  }
  return 0;
}


} // extern "C"
