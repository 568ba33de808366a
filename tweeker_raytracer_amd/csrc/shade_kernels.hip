// Ray generation, shading and accumulation kernels of the wavefront renderer.
//
// Reference programs restated per stage (apps/rtigo3/shaders/):
//   generateKernel   __raygen__path_tracer up to the first optixTrace      raygeneration.cu:167-203, lens_shader.cu
//   shadeKernel      miss programs miss.cu:41-109, __closesthit__radiance closesthit.cu:126-305, the BSDF /
//                    light callables (bxdf_*.cu, light_sample.cu) and the integrator loop body after the
//                    trace raygeneration.cu:91-146
//   accumulateKernel NaN filter + running mean                              raygeneration.cu:222-253
//   compositorKernel compositor.cu:38-64 for all source devices at once
// Per-path RNG draw order is the reference's: jitter rng2; per bounce the BSDF's draws, then NEE rng2
// [+ rng when more than one light], then Russian roulette rng.
#include "device_types.h"

namespace twk {

// shaders/random_number_generators.h:40-53
template<unsigned int N>
TWK_D unsigned int tea(const unsigned int val0, const unsigned int val1)
{
  unsigned int v0 = val0;
  unsigned int v1 = val1;
  unsigned int s0 = 0;
  for (unsigned int n = 0; n < N; ++n)
  {
    s0 += 0x9e3779b9;
    v0 += ((v1 << 4) + 0xA341316C) ^ (v1 + s0) ^ ((v1 >> 5) + 0xC8013EA4);
    v1 += ((v0 << 4) + 0xAD90777D) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7E95761E);
  }
  return v0;
}

// shaders/random_number_generators.h:56-62
TWK_D float rng(unsigned int& previous)
{
  previous = previous * 1664525u + 1013904223u;
  return float(previous & 0x00FFFFFF) / float(0x01000000u);
}

struct PathPrd // the part of PerRayData (per_ray_data.h:84-114) a bounce works on, in registers
{
  V3 pos, wo, wi;
  V3 radiance, f_over_pdf, sigma_t;
  float4 absorption_ior;
  float iorX, iorY;
  float distance, pdf;
  unsigned int flags, seed;
};

struct SurfaceState // per_ray_data.h:74-81
{
  V3 normalGeo, tangent, normal, texcoord, albedo;
};

TWK_D float4 tex2D(const DevTexture& tex, float u, float v)
{
  const int W = tex.width, H = tex.height;
  u = u - floorf(u);
  const float xB = u * float(W) - 0.5f;
  const float xf = floorf(xB);
  const float a  = xB - xf;
  int i0 = (int) xf; int i1 = i0 + 1;
  i0 = ((i0 % W) + W) % W; i1 = ((i1 % W) + W) % W;
  float b; int j0, j1;
  if (tex.clampV)
  {
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    const float yB = v * float(H) - 0.5f;
    const float yf = floorf(yB);
    b = yB - yf;
    j0 = min(max((int) yf, 0), H - 1); j1 = min(max((int) yf + 1, 0), H - 1);
  }
  else
  {
    v = v - floorf(v);
    const float yB = v * float(H) - 0.5f;
    const float yf = floorf(yB);
    b = yB - yf;
    j0 = (int) yf; j1 = j0 + 1;
    j0 = ((j0 % H) + H) % H; j1 = ((j1 % H) + H) % H;
  }
  const float4 t00 = tex.texels[(size_t) j0 * W + i0];
  const float4 t10 = tex.texels[(size_t) j0 * W + i1];
  const float4 t01 = tex.texels[(size_t) j1 * W + i0];
  const float4 t11 = tex.texels[(size_t) j1 * W + i1];
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  float4 r;
  r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
  r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
  r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
  r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
  return r;
}

// ---------------------------------------------------------------------------------------------
// raygeneration.cu:152-164
TWK_D unsigned int distribute(const LaunchParams& p, unsigned int x, unsigned int y)
{
  const unsigned int xBlock = x >> p.tileShift[0];
  const unsigned int yBlock = y >> p.tileShift[1];
  const unsigned int xTile = xBlock * p.deviceCount + ((p.deviceIndex + yBlock) % p.deviceCount);
  return xTile * p.tileSize[0] + (x & (p.tileSize[0] - 1));
}

// One thread per launch index: seed, jitter, lens shader, path state reset, primary ray into queue 0.
// The seed index is the absolute pixel W*y + x for every device count (for one device this IS the
// reference's formula raygeneration.cu:191; for several it makes tiled == single-device bit for bit).
__global__ void __launch_bounds__(256) generateKernel(LaunchParams p)
{
  const unsigned int index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= (unsigned int) p.numPixels) return;
  const unsigned int lx = index % (unsigned int) p.launchWidth;
  const unsigned int ly = index / (unsigned int) p.launchWidth;

  unsigned int launchColumn = lx;
  bool active = true;
  if (p.distribution && 1 < p.deviceCount)
  {
    launchColumn = distribute(p, lx, ly);
    active = launchColumn < (unsigned int) p.resolution[0];
  }

  V3 origin = v3(0.0f), direction = v3(0.0f, 0.0f, 1.0f);
  unsigned int seed = 0;
  if (active)
  {
    seed = tea<4>((unsigned int) p.resolution[0] * ly + launchColumn, p.iterationIndex);

    const float screenX = float(p.resolution[0]), screenY = float(p.resolution[1]);
    const float pixelX  = float(launchColumn),    pixelY  = float(ly);
    const float sampleX = rng(seed);
    const float sampleY = rng(seed);

    const float* cam = p.camera;
    const V3 P = v3(cam[0], cam[1], cam[2]), U = v3(cam[3], cam[4], cam[5]), V = v3(cam[6], cam[7], cam[8]), W = v3(cam[9], cam[10], cam[11]);

    if (p.lensShader == 1) // lens_shader.cu:55-73 fisheye
    {
      const float fx = pixelX + sampleX, fy = pixelY + sampleY;
      const float cx = screenX * 0.5f, cy = screenY * 0.5f;
      const float inv = 1.0f / sqrtf(cx * cx + cy * cy);
      const float uvx = (fx - cx) * inv, uvy = (fy - cy) * inv;
      const float z = cosP(sqrtf(uvx * uvx + uvy * uvy) * 0.7071067812f * 0.5f * kPi);
      const V3 Un = normalize(U), Vn = normalize(V), Wn = normalize(W);
      origin = P;
      direction = normalize(uvx * Un + uvy * Vn + z * Wn);
    }
    else if (p.lensShader == 2) // lens_shader.cu:76-99 sphere
    {
      const float uvx = (pixelX + sampleX) / screenX, uvy = (pixelY + sampleY) / screenY;
      const float phi   = uvx * 2.0f * kPi;
      const float theta = uvy * kPi;
      const float sinTheta = sinP(theta);
      const V3 v = v3(-sinP(phi) * sinTheta, -cosP(theta), -cosP(phi) * sinTheta);
      const V3 Un = normalize(U), Vn = normalize(V), Wn = normalize(W);
      origin = P;
      direction = normalize(v.x * Un + v.y * Vn + v.z * Wn);
    }
    else // lens_shader.cu:40-52 pinhole
    {
      const float ndcX = ((pixelX + sampleX) / screenX) * 2.0f - 1.0f;
      const float ndcY = ((pixelY + sampleY) / screenY) * 2.0f - 1.0f;
      origin = P;
      direction = normalize(U * ndcX + V * ndcY + W);
    }
  }

  // integrator prologue (raygeneration.cu:53-62): black radiance, unit throughput, empty volume stack
  p.pathRadiance[index]   = make_float4(0.0f, 0.0f, 0.0f, active ? 1.0f : 0.0f);
  p.pathThroughput[index] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
  p.pathSeedFlags[index]  = make_uint2(seed, 0u);

  // Inactive launch indices (tile columns beyond the image) still own a slot so that queue 0 is the
  // identity mapping; they carry tmax < tmin and never hit anything, and shade drops them.
  p.rayOrg[0][index]   = make_float4(origin.x, origin.y, origin.z, p.sceneEpsilon);
  p.rayDir[0][index]   = make_float4(direction.x, direction.y, direction.z, active ? RT_DEFAULT_MAX : -1.0f);
  p.rayPixel[0][index] = index;
  if (index == 0) p.counters[0] = (unsigned int) p.numPixels;
}

// ---------------------------------------------------------------------------------------------
// BSDFs

// bxdf_diffuse.cu:39-47
TWK_D void alignVector(const V3& axis, V3& w)
{
  const float s = copysignf(1.0f, axis.z);
  w.z *= s;
  const V3 h = v3(axis.x, axis.y, axis.z + s);
  const float k = dot(w, h) / (1.0f + fabsf(axis.z));
  w = k * h - w;
}

// bxdf_diffuse.cu:49-63
TWK_D void unitSquareToCosineHemisphere(float sx, float sy, const V3& axis, V3& w, float& pdf)
{
  const float theta = 2.0f * kPi * sx;
  const float r = sqrtf(sy);
  w.x = r * cosP(theta);
  w.y = r * sinP(theta);
  w.z = 1.0f - w.x * w.x - w.y * w.y;
  w.z = (0.0f < w.z) ? sqrtf(w.z) : 0.0f;
  pdf = w.z * kInvPi;
  alignVector(axis, w);
}

// shader_common.h:47-77
TWK_D bool refract(V3& r, const V3& i, const V3& n, const float ior)
{
  V3 nn = n;
  float negNdotV = dot(i, nn);
  float eta;
  if (negNdotV > 0.0f) { eta = ior; nn = -n; negNdotV = -negNdotV; }
  else                 { eta = 1.f / ior; }
  const float k = 1.f - eta * eta * (1.f - negNdotV * negNdotV);
  if (k < 0.0f) { r = v3(0.f); return false; }
  r = normalize(eta * i - (eta * negNdotV + sqrtf(k)) * nn);
  return true;
}

// bxdf_specular.cu:42-67
TWK_D float evaluateFresnelDielectric(const float et, const float cosIn)
{
  const float cosi = fabsf(cosIn);
  float sint = 1.0f - cosi * cosi;
  sint = (0.0f < sint) ? sqrtf(sint) / et : 0.0f;
  if (1.0f < sint) return 1.0f;
  float cost = 1.0f - sint * sint;
  cost = (0.0f < cost) ? sqrtf(cost) : 0.0f;
  const float et_cosi = et * cosi;
  const float et_cost = et * cost;
  const float rPerpendicular = (cosi - et_cost) / (cosi + et_cost);
  const float rParallel      = (et_cosi - cost) / (et_cosi + cost);
  const float result = (rParallel * rParallel + rPerpendicular * rPerpendicular) * 0.5f;
  return (result <= 1.0f) ? result : 1.0f;
}

// shader_common.h:119-125 + :133-143
struct TangentSpace
{
  V3 tangent, bitangent, normal;
  TWK_D TangentSpace(const V3& tangentReference, const V3& n)
  {
    normal    = n;
    bitangent = normalize(cross(normal, tangentReference));
    tangent   = cross(bitangent, normal);
  }
  TWK_D V3 toLocal(const V3& q) const { return v3(dot(q, tangent), dot(q, bitangent), dot(q, normal)); }
  TWK_D V3 toWorld(const V3& q) const { return q.x * tangent + q.y * bitangent + q.z * normal; }
};

// bxdf_ggx_smith.cu:74-94
TWK_D void distribution_d_pdf(const float ax, const float ay, const V3& wm, float& d, float& pdf)
{
  d = 0.0f; pdf = 0.0f;
  if (DENOMINATOR_EPSILON < wm.z)
  {
    const float cosThetaSqr = wm.z * wm.z;
    const float tanThetaSqr = (1.0f - cosThetaSqr) / cosThetaSqr;
    const float phiM    = atan2P(wm.y, wm.x);
    const float cosPhiM = cosP(phiM);
    const float sinPhiM = sinP(phiM);
    const float term = 1.0f + tanThetaSqr * ((cosPhiM * cosPhiM) / (ax * ax) + (sinPhiM * sinPhiM) / (ay * ay));
    d   = 1.0f / (kPi * ax * ay * cosThetaSqr * cosThetaSqr * term * term);
    pdf = d * wm.z;
  }
}

// bxdf_ggx_smith.cu:96-106
TWK_D V3 distribution_sample(const float ax, const float ay, const float u1, const float u2)
{
  const float theta    = atanP(ay * sqrtf(u1) / sqrtf(1.0f - u1));
  const float phi      = 2.0f * kPi * u2;
  const float sinTheta = sinP(theta);
  return normalize(v3(cosP(phi) * sinTheta * ax / ay, sinP(phi) * sinTheta, cosP(theta)));
}

// bxdf_ggx_smith.cu:109-125
TWK_D float smith_G1(const float alpha, const V3& w, const V3& wm)
{
  const float w_wm = dot(w, wm);
  if (w_wm * w.z <= 0.0f) return 0.0f;
  const float cosThetaSqr = w.z * w.z;
  const float sinThetaSqr = 1.0f - cosThetaSqr;
  const float tanThetaSqr = (0.0f < sinThetaSqr) ? sinThetaSqr / cosThetaSqr : 0.0f;
  const float invASqr = alpha * alpha * tanThetaSqr;
  return 2.0f / (1.0f + sqrtf(1.0f + invASqr));
}

// bxdf_ggx_smith.cu:150-165
TWK_D float distribution_G(const float ax, const float ay, const V3& wo, const V3& wi, const V3& wm)
{
  float phi   = atan2P(wo.y, wo.x);
  float c     = cosP(phi);
  float s     = sinP(phi);
  float alpha = sqrtf(c * c * ax * ax + s * s * ay * ay);
  const float g = smith_G1(alpha, wo, wm);
  phi   = atan2P(wi.y, wi.x);
  c     = cosP(phi);
  s     = sinP(phi);
  alpha = sqrtf(c * c * ax * ax + s * s * ay * ay);
  return g * smith_G1(alpha, wi, wm);
}

// Sample callables, dispatched on MaterialDefinition::indexBSDF (closesthit.cu:246-248).
TWK_D void sampleBsdf(const DevMaterial& material, const SurfaceState& state, PathPrd& prd)
{
  switch (material.indexBSDF)
  {
    default:
    case 0: // bxdf_diffuse.cu:67-86
    {
      const float sx = rng(prd.seed);
      const float sy = rng(prd.seed);
      unitSquareToCosineHemisphere(sx, sy, state.normal, prd.wi, prd.pdf);
      if (prd.pdf <= 0.0f || dot(prd.wi, state.normalGeo) <= 0.0f) { prd.flags |= TWK_FLAG_TERMINATE; return; }
      prd.f_over_pdf = state.albedo;
      prd.flags |= TWK_FLAG_DIFFUSE;
      return;
    }
    case 1: // bxdf_specular.cu:71-83
    {
      prd.wi = reflect(-prd.wo, state.normal);
      if (dot(prd.wi, state.normalGeo) <= 0.0f) { prd.flags |= TWK_FLAG_TERMINATE; return; }
      prd.f_over_pdf = state.albedo;
      prd.pdf        = 1.0f;
      return;
    }
    case 2: // bxdf_specular.cu:94-134
    {
      prd.absorption_ior = make_float4(material.absorption[0], material.absorption[1], material.absorption[2], material.ior);
      const float eta = (prd.flags & (TWK_FLAG_FRONTFACE | TWK_FLAG_THINWALLED)) ? prd.absorption_ior.w / prd.iorX : prd.iorY / prd.absorption_ior.w;
      const V3 R = reflect(-prd.wo, state.normal);
      float reflective = 1.0f;
      if (refract(prd.wi, -prd.wo, state.normal, eta))
      {
        if (prd.flags & TWK_FLAG_THINWALLED) prd.wi = -prd.wo;
        reflective = evaluateFresnelDielectric(eta, dot(prd.wo, state.normal));
      }
      const float pseudo = rng(prd.seed);
      if (pseudo < reflective) prd.wi = R;
      else if (!(prd.flags & TWK_FLAG_THINWALLED)) prd.flags |= TWK_FLAG_TRANSMISSION;
      prd.f_over_pdf = state.albedo;
      prd.pdf        = 1.0f;
      return;
    }
    case 3: // bxdf_ggx_smith.cu:169-222
    {
      const float sx = rng(prd.seed);
      const float sy = rng(prd.seed);
      const V3 wm = distribution_sample(material.roughness[0], material.roughness[1], sx, sy);
      const TangentSpace ts(state.tangent, state.normal);
      const V3 wh = ts.toWorld(wm);
      prd.wi = reflect(-prd.wo, wh);
      if (dot(prd.wi, state.normalGeo) <= 0.0f) { prd.flags |= TWK_FLAG_TERMINATE; return; }
      const V3 wo = ts.toLocal(prd.wo);
      const V3 wi = ts.toLocal(prd.wi);
      const float wi_wh = dot(prd.wi, wh);
      if (wo.z <= 0.0f || wi.z <= 0.0f || wi_wh <= 0.0f) { prd.flags |= TWK_FLAG_TERMINATE; return; }
      float D, PDF;
      distribution_d_pdf(material.roughness[0], material.roughness[1], wm, D, PDF);
      if (PDF <= 0.0f) { prd.flags |= TWK_FLAG_TERMINATE; return; }
      const float G = distribution_G(material.roughness[0], material.roughness[1], wo, wi, wm);
      prd.pdf = PDF / (4.0f * wi_wh);
      prd.f_over_pdf = state.albedo * (G * D * wi_wh / (PDF * wo.z));
      prd.flags |= TWK_FLAG_DIFFUSE;
      return;
    }
    case 4: // bxdf_ggx_smith.cu:265-319
    {
      prd.absorption_ior = make_float4(material.absorption[0], material.absorption[1], material.absorption[2], material.ior);
      const float eta = (prd.flags & (TWK_FLAG_FRONTFACE | TWK_FLAG_THINWALLED)) ? prd.absorption_ior.w / prd.iorX : prd.iorY / prd.absorption_ior.w;
      const float sx = rng(prd.seed);
      const float sy = rng(prd.seed);
      const V3 wm = distribution_sample(material.roughness[0], material.roughness[1], sx, sy);
      const TangentSpace ts(state.tangent, state.normal);
      const V3 wh = ts.toWorld(wm);
      const V3 R = reflect(-prd.wo, wh);
      float reflective = 1.0f;
      if (refract(prd.wi, -prd.wo, wh, eta))
      {
        if (prd.flags & TWK_FLAG_THINWALLED) prd.wi = reflect(R, state.normal);
        reflective = evaluateFresnelDielectric(eta, dot(prd.wo, wh));
      }
      const float pseudo = rng(prd.seed);
      if (pseudo < reflective) prd.wi = R;
      else if (!(prd.flags & TWK_FLAG_THINWALLED)) prd.flags |= TWK_FLAG_TRANSMISSION;
      prd.f_over_pdf = state.albedo;
      prd.pdf        = 1.0f;
      return;
    }
  }
}

// Eval callables (closesthit.cu:271): f in xyz, pdf in w.
TWK_D float4 evalBsdf(const DevMaterial& material, const SurfaceState& state, const PathPrd& prd, const V3& wiL)
{
  if (material.indexBSDF == 0) // bxdf_diffuse.cu:89-96
  {
    const V3 f = state.albedo * kInvPi;
    const float pdf = fmaxf(0.0f, dot(wiL, state.normal) * kInvPi);
    return make_float4(f.x, f.y, f.z, pdf);
  }
  if (material.indexBSDF == 3) // bxdf_ggx_smith.cu:226-261
  {
    const TangentSpace ts(state.tangent, state.normal);
    const V3 wo = ts.toLocal(prd.wo);
    const V3 wi = ts.toLocal(wiL);
    if (wo.z <= 0.0f || wi.z <= 0.0f) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    V3 wm = wo + wi;
    if (isNull(wm)) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    wm = normalize(wm);
    float D, PDF;
    distribution_d_pdf(material.roughness[0], material.roughness[1], wm, D, PDF);
    const float G = distribution_G(material.roughness[0], material.roughness[1], wo, wi, wm);
    const V3 f = state.albedo * (D * G / (4.0f * wo.z * wi.z));
    const float pdf = PDF / (4.0f * dot(wi, wm));
    return make_float4(f.x, f.y, f.z, pdf);
  }
  return make_float4(0.0f, 0.0f, 0.0f, 0.0f); // bxdf_specular.cu:87-90, shared by every specular BSDF
}

// ---------------------------------------------------------------------------------------------
// Light sampling (light_sample.cu). Returns pdf (0 = unusable).
struct LightSampleD
{
  V3 direction, emission;
  float distance, pdf;
};

TWK_D void sampleLight(const LaunchParams& p, int index, const V3& point, float sx, float sy, LightSampleD& ls)
{
  const DevLight& light = p.lights[index];
  if (light.type == 1) // light_sample.cu:156-177 parallelogram
  {
    ls.pdf = 0.0f;
    const V3 position = v3(light.position[0], light.position[1], light.position[2]) +
                        v3(light.vecU[0], light.vecU[1], light.vecU[2]) * sx +
                        v3(light.vecV[0], light.vecV[1], light.vecV[2]) * sy;
    ls.direction = position - point;
    ls.distance  = length(ls.direction);
    ls.emission  = v3(0.0f);
    if (DENOMINATOR_EPSILON < ls.distance)
    {
      ls.direction = ls.direction / ls.distance;
      const float cosTheta = dot(-ls.direction, v3(light.normal[0], light.normal[1], light.normal[2]));
      if (DENOMINATOR_EPSILON < cosTheta)
      {
        ls.emission = v3(light.emission[0], light.emission[1], light.emission[2]) * float(p.numLights);
        ls.pdf      = (ls.distance * ls.distance) / (light.area * cosTheta);
      }
    }
    return;
  }
  if (p.miss == 2) // light_sample.cu:67-153 importance-sampled spherical environment
  {
    const unsigned int sizeV = p.envHeight;
    unsigned int ilo = 0, ihi = sizeV;
    const float* cdfV = p.envCDF_V;
    while (ilo != ihi - 1)
    {
      const unsigned int i = (ilo + ihi) >> 1;
      if (sy < cdfV[i]) ihi = i; else ilo = i;
    }
    const unsigned int vIdx = ilo;
    const unsigned int sizeU = p.envWidth;
    ilo = 0; ihi = sizeU;
    const float* cdfU = &p.envCDF_U[(size_t) vIdx * (sizeU + 1)];
    while (ilo != ihi - 1)
    {
      const unsigned int i = (ilo + ihi) >> 1;
      if (sx < cdfU[i]) ihi = i; else ilo = i;
    }
    const unsigned int uIdx = ilo;
    const float cdfLowerU = cdfU[uIdx], cdfUpperU = cdfU[uIdx + 1];
    const float du = (sx - cdfLowerU) / (cdfUpperU - cdfLowerU);
    const float cdfLowerV = cdfV[vIdx], cdfUpperV = cdfV[vIdx + 1];
    const float dv = (sy - cdfLowerV) / (cdfUpperV - cdfLowerV);
    const float u = (float(uIdx) + du) / float(sizeU);
    const float v = (float(vIdx) + dv) / float(sizeV);
    const float phi   = (u - p.envRotation) * 2.0f * kPi;
    const float theta = v * kPi;
    const float sinTheta = sinP(theta);
    ls.direction = v3(-sinP(phi) * sinTheta, -cosP(theta), cosP(phi) * sinTheta);
    ls.distance  = RT_DEFAULT_MAX;
    const V3 emission = v3(tex2D(p.textures[2], u, v));
    ls.emission = emission * float(p.numLights);
    ls.pdf = intensity(emission) / p.envIntegral;
    return;
  }
  // light_sample.cu:40-65 constant environment
  {
    V3 d;
    d.z = 1.0f - 2.0f * sx;
    float r = 1.0f - d.z * d.z;
    r = (0.0f < r) ? sqrtf(r) : 0.0f;
    const float phi = sy * 2.0f * kPi;
    d.x = r * cosP(phi);
    d.y = r * sinP(phi);
    ls.direction = d;
    ls.pdf = 0.25f * kInvPi;
    ls.distance = RT_DEFAULT_MAX;
    ls.emission = v3(float(p.numLights));
  }
}

// Queue append: one atomic per wave for all lanes that call it together (ballot + prefix popcount).
TWK_D unsigned int waveAppend(unsigned int* counter)
{
  const unsigned long long mask = __ballot(1);
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int rank = __popcll(mask & ((1ull << lane) - 1ull));
  const int leader = __ffsll((long long) mask) - 1;
  unsigned int base = 0;
  if ((int) lane == leader) base = atomicAdd(counter, (unsigned int) __popcll(mask));
  base = __shfl(base, leader);
  return base + rank;
}

// ---------------------------------------------------------------------------------------------
// One thread per ray of queue (depth & 1): miss or closest-hit shading, next-event estimation,
// then the integrator's loop tail. Appends the continuation ray to queue ((depth + 1) & 1) and the
// shadow ray (with the pending contribution) to the shadow queue.
__global__ void __launch_bounds__(256) shadeKernel(LaunchParams p, int depth)
{
  const unsigned int numRays = p.counters[depth * TWK_COUNTERS_PER_DEPTH + 0];
  const int q = depth & 1, qn = q ^ 1;
  unsigned int* nextCount   = &p.counters[(depth + 1) * TWK_COUNTERS_PER_DEPTH + 0];
  unsigned int* shadowCount = &p.counters[depth * TWK_COUNTERS_PER_DEPTH + 1];
  unsigned int statHit = 0, statMiss = 0;

  for (unsigned int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < numRays; slot += gridDim.x * blockDim.x)
  {
    const float4 ro = p.rayOrg[q][slot];
    const float4 rd = p.rayDir[q][slot];
    if (rd.w < 0.0f) continue; // inactive launch index (tile column beyond the image)
    const unsigned int pixel = p.rayPixel[q][slot];
    const float4 hit = p.hitRecord[slot];
    const int instanceIndex = p.hitInstance[slot];

    const float4 tp = p.pathThroughput[pixel];
    const uint2  sf = p.pathSeedFlags[pixel];
    V3 throughput = v3(tp.x, tp.y, tp.z);

    PathPrd prd;
    prd.pos = v3(ro); prd.wi = v3(rd);
    prd.pdf = tp.w;
    prd.seed = sf.x;
    int stackIdx = (int) ((sf.y >> TWK_PATH_STACK_SHIFT) & 7u) - 1;

    // raygeneration.cu:63-78: per-segment reset + volume state
    prd.wo       = -prd.wi;
    prd.iorX     = 1.0f; prd.iorY = 1.0f;
    prd.distance = RT_DEFAULT_MAX;
    prd.flags    = sf.y & TWK_FLAG_CLEAR_MASK;
    prd.sigma_t  = v3(0.0f);
    prd.absorption_ior = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    prd.f_over_pdf = v3(0.0f);
    prd.radiance = v3(0.0f);
    if (TWK_MATERIAL_STACK_FIRST <= stackIdx)
    {
      prd.flags |= TWK_FLAG_VOLUME;
      const float4 top = p.volumeStack[(size_t) stackIdx * p.numPixels + pixel];
      prd.sigma_t = v3(top);
      prd.iorX    = top.w;
      if (TWK_MATERIAL_STACK_FIRST <= stackIdx - 1)
      {
        prd.iorY = p.volumeStack[(size_t) (stackIdx - 1) * p.numPixels + pixel].w;
      }
    }

    bool wantShadow = false;
    V3 shadowDir = v3(0.0f), contribution = v3(0.0f);
    float shadowTmax = 0.0f;

    if (instanceIndex < 0)
    {
      // ---- miss programs, miss.cu
      if (p.miss == 0) { prd.radiance = v3(0.0f); }                                   // :41-52
      else if (p.miss == 2)                                                            // :75-109
      {
        const V3 R = prd.wi;
        const float u     = (atan2P(R.x, -R.z) + kPi) * 0.5f * kInvPi + p.envRotation;
        const float theta = acosP(-R.y);
        const float v     = theta * kInvPi;
        const V3 emission = v3(tex2D(p.textures[2], u, v));
        float weightMIS = 1.0f;
        if (prd.flags & TWK_FLAG_DIFFUSE)
        {
          const float pdfLight = intensity(emission) / p.envIntegral;
          weightMIS = powerHeuristic(prd.pdf, pdfLight);
        }
        prd.radiance = emission * weightMIS;
      }
      else                                                                             // :54-73
      {
        const float weightMIS = (prd.flags & TWK_FLAG_DIFFUSE) ? powerHeuristic(prd.pdf, 0.25f * kInvPi) : 1.0f;
        prd.radiance = v3(weightMIS);
      }
      prd.flags |= TWK_FLAG_TERMINATE;
    }
    else
    {
      // ---- __closesthit__radiance, closesthit.cu:126-305
      const DevInstance& inst = p.instances[instanceIndex];
      const unsigned int prim = (unsigned int) __float_as_int(hit.w);
      const unsigned int* tri = p.indices + inst.indexBase + 3 * (size_t) prim;
      const float* a0 = p.attributes + 12 * (size_t) (inst.attributeBase + tri[0]);
      const float* a1 = p.attributes + 12 * (size_t) (inst.attributeBase + tri[1]);
      const float* a2 = p.attributes + 12 * (size_t) (inst.attributeBase + tri[2]);

      const float beta = hit.y, gamma = hit.z;
      const float alpha = 1.0f - beta - gamma;

      const V3 v0 = v3(a0[0], a0[1], a0[2]), v1 = v3(a1[0], a1[1], a1[2]), v2 = v3(a2[0], a2[1], a2[2]);
      const V3 ng = cross(v1 - v0, v2 - v0);
      const V3 tg = v3(a0[3], a0[4], a0[5]) * alpha + v3(a1[3], a1[4], a1[5]) * beta + v3(a2[3], a2[4], a2[5]) * gamma;
      const V3 ns = v3(a0[6], a0[7], a0[8]) * alpha + v3(a1[6], a1[7], a1[8]) * beta + v3(a2[6], a2[7], a2[8]) * gamma;

      SurfaceState state;
      state.texcoord = v3(a0[9], a0[10], a0[11]) * alpha + v3(a1[9], a1[10], a1[11]) * beta + v3(a2[9], a2[10], a2[11]) * gamma;

      state.normalGeo = normalize(transformNormal(inst.worldToObject, ng));
      state.tangent   = normalize(transformVector(inst.objectToWorld, tg));
      state.normal    = normalize(transformNormal(inst.worldToObject, ns));

      prd.distance = hit.x;
      prd.pos = prd.pos + prd.wi * prd.distance;

      prd.flags |= (0.0f <= dot(prd.wo, state.normalGeo)) ? TWK_FLAG_FRONTFACE : 0u;
      if ((prd.flags & TWK_FLAG_FRONTFACE) == 0)
      {
        state.normalGeo = -state.normalGeo;
        state.tangent   = -state.tangent;
        state.normal    = -state.normal;
      }

      bool lightHit = false;
      if (0 <= inst.light && (prd.flags & TWK_FLAG_FRONTFACE))
      {
        const float cosTheta = dot(prd.wo, state.normalGeo);
        if (DENOMINATOR_EPSILON < cosTheta)
        {
          const DevLight& light = p.lights[inst.light];
          V3 emission = v3(light.emission[0], light.emission[1], light.emission[2]);
          const float lightPdf = (prd.distance * prd.distance) / (light.area * cosTheta);
          if ((prd.flags & TWK_FLAG_DIFFUSE) && DENOMINATOR_EPSILON < lightPdf)
          {
            emission = emission * powerHeuristic(prd.pdf, lightPdf);
          }
          prd.radiance = emission;
          prd.flags |= TWK_FLAG_TERMINATE;
          lightHit = true;
        }
      }

      if (!lightHit)
      {
        prd.f_over_pdf = v3(0.0f);
        prd.pdf        = 0.0f;

        const DevMaterial& material = p.materials[inst.material];
        state.albedo = v3(material.albedo[0], material.albedo[1], material.albedo[2]);
        if (material.textureAlbedo != 0)
        {
          const V3 texColor = v3(tex2D(p.textures[0], state.texcoord.x, state.texcoord.y));
          state.albedo = state.albedo * texColor;
        }

        prd.flags = (prd.flags & ~TWK_FLAG_DIFFUSE) | TWK_FLAG_HIT | material.flags;

        sampleBsdf(material, state, prd);

        const int numLights = p.numLights;
        if ((prd.flags & TWK_FLAG_DIFFUSE) && 0 < numLights)
        {
          const float sx = rng(prd.seed);
          const float sy = rng(prd.seed);
          const int lightIndex = (1 < numLights) ? min(max(static_cast<int>(floorf(rng(prd.seed) * numLights)), 0), numLights - 1) : 0;

          LightSampleD ls;
          sampleLight(p, lightIndex, prd.pos, sx, sy, ls);

          if (0.0f < ls.pdf)
          {
            const float4 bsdf_pdf = evalBsdf(material, state, prd, ls.direction);
            const V3 bsdf = v3(bsdf_pdf);
            if (0.0f < bsdf_pdf.w && isNotNull(bsdf))
            {
              // The shadow ray is traced by the next trace launch; the contribution it would add when
              // unoccluded is computed now (closesthit.cu:288-299).
              V3 emission = ls.emission;
              if (prd.flags & TWK_FLAG_VOLUME)
              {
                emission = emission * exp3(-ls.distance * prd.sigma_t);
              }
              const float weightMis = powerHeuristic(ls.pdf, bsdf_pdf.w);
              contribution = bsdf * emission * (weightMis * dot(ls.direction, state.normal) / ls.pdf);
              shadowDir  = ls.direction;
              shadowTmax = ls.distance - p.sceneEpsilon;
              wantShadow = true;
            }
          }
        }
      }
    }

    // ---- integrator loop tail, raygeneration.cu:91-146
    if (prd.flags & TWK_FLAG_VOLUME)
    {
      throughput = throughput * exp3(-prd.distance * prd.sigma_t);
    }

    if (!wantShadow)
    {
      // emission / environment (or nothing): radiance += throughput * prd.radiance
      float4 r = p.pathRadiance[pixel];
      const V3 add = throughput * prd.radiance;
      r.x += add.x; r.y += add.y; r.z += add.z;
      p.pathRadiance[pixel] = r;
    }
    else
    {
      const unsigned int s = waveAppend(shadowCount);
      const V3 pending = throughput * contribution;
      p.shadowOrg[s]     = make_float4(prd.pos.x, prd.pos.y, prd.pos.z, p.sceneEpsilon);
      p.shadowDir[s]     = make_float4(shadowDir.x, shadowDir.y, shadowDir.z, shadowTmax);
      p.shadowPixel[s]   = pixel;
      p.shadowPending[s] = make_float4(pending.x, pending.y, pending.z, 0.0f);
    }

    bool alive = !((prd.flags & TWK_FLAG_TERMINATE) || prd.pdf <= 0.0f || isNull(prd.f_over_pdf));
    if (alive)
    {
      throughput = throughput * prd.f_over_pdf;
      if (p.pathLengths[0] <= depth)
      {
        const float probability = maxComponent(throughput);
        if (probability < rng(prd.seed)) alive = false;
        else throughput = throughput / probability;
      }
    }
    if (alive)
    {
      if ((prd.flags & (TWK_FLAG_THINWALLED | TWK_FLAG_TRANSMISSION)) == TWK_FLAG_TRANSMISSION)
      {
        if (prd.flags & TWK_FLAG_FRONTFACE)
        {
          stackIdx = min(stackIdx + 1, TWK_MATERIAL_STACK_LAST);
          p.volumeStack[(size_t) stackIdx * p.numPixels + pixel] = prd.absorption_ior;
        }
        else
        {
          stackIdx = max(stackIdx - 1, TWK_MATERIAL_STACK_EMPTY);
        }
      }
      alive = (depth + 1 < p.pathLengths[1]);
    }

    if (alive)
    {
      p.pathThroughput[pixel] = make_float4(throughput.x, throughput.y, throughput.z, prd.pdf);
      p.pathSeedFlags[pixel]  = make_uint2(prd.seed, (prd.flags & TWK_FLAG_CLEAR_MASK) | ((unsigned int) (stackIdx + 1) << TWK_PATH_STACK_SHIFT));
      const unsigned int n = waveAppend(nextCount);
      p.rayOrg[qn][n]   = make_float4(prd.pos.x, prd.pos.y, prd.pos.z, p.sceneEpsilon);
      p.rayDir[qn][n]   = make_float4(prd.wi.x, prd.wi.y, prd.wi.z, RT_DEFAULT_MAX);
      p.rayPixel[qn][n] = pixel;
    }

    if (p.stats != nullptr)
    {
      if (instanceIndex < 0) ++statMiss; else ++statHit;
    }
  }

  if (p.stats != nullptr)
  {
    // one atomic per wave, not per path
    for (int offset = 32; offset > 0; offset >>= 1)
    {
      statHit  += __shfl_down(statHit, offset);
      statMiss += __shfl_down(statMiss, offset);
    }
    if ((threadIdx.x & 63) == 0)
    {
      if (statHit)  atomicAdd(&p.stats[5], (unsigned long long) statHit);
      if (statMiss) atomicAdd(&p.stats[6], (unsigned long long) statMiss);
    }
  }
}

// raygeneration.cu:222-253: drop NaN samples, running mean into the RGBA32F buffer, alpha 1.
__global__ void __launch_bounds__(256) accumulateKernel(LaunchParams p)
{
  const unsigned int index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= (unsigned int) p.numPixels) return;
  const float4 r = p.pathRadiance[index];
  if (r.w == 0.0f) return; // launch index outside the image (tile padding): never written, like the early return at raygeneration.cu:180-183
  V3 radiance = v3(r.x, r.y, r.z);
  if (!(isnan(radiance.x) || isnan(radiance.y) || isnan(radiance.z)))
  {
    if (0 < p.iterationIndex)
    {
      const float4 dst = p.output[index];
      radiance = lerp(v3(dst.x, dst.y, dst.z), radiance, 1.0f / float(p.iterationIndex + 1));
    }
    p.output[index] = make_float4(radiance.x, radiance.y, radiance.z, 1.0f);
  }
}

// compositor.cu:38-64 for every source device in one launch: tiles is [deviceCount][H][launchWidth].
__global__ void __launch_bounds__(256) compositorKernel(const float4* __restrict__ tiles, float4* __restrict__ output,
                                                         int width, int height, int launchWidth, int deviceCount,
                                                         int tileSizeX, int tileShiftX, int tileShiftY)
{
  const unsigned int xLaunch = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned int yLaunch = blockIdx.y;
  const unsigned int device  = blockIdx.z;
  if (xLaunch >= (unsigned int) launchWidth || yLaunch >= (unsigned int) height) return;
  const unsigned int xBlock = xLaunch >> tileShiftX;
  const unsigned int yBlock = yLaunch >> tileShiftY;
  const unsigned int xTile  = xBlock * deviceCount + ((device + yBlock) % deviceCount);
  const unsigned int xPixel = xTile * tileSizeX + (xLaunch & (tileSizeX - 1));
  if (xPixel < (unsigned int) width)
  {
    output[(size_t) yLaunch * width + xPixel] = tiles[((size_t) device * height + yLaunch) * launchWidth + xLaunch];
  }
}

__global__ void mathTapKernel(int op, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, size_t n)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
  {
    float r = 0.0f;
    switch (op)
    {
      case 0: r = sinP(x[i]); break;
      case 1: r = cosP(x[i]); break;
      case 2: r = expP(x[i]); break;
      case 3: r = atan2P(x[i], y[i]); break;
      case 4: r = acosP(x[i]); break;
      case 5: r = atanP(x[i]); break;
      case 6: r = sqrtf(x[i]); break;
      case 7: r = 1.0f / x[i]; break;
    }
    out[i] = r;
  }
}

__global__ void streamCopyKernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
}

void launchGenerate(const LaunchParams& p, hipStream_t stream)
{
  hipLaunchKernelGGL(generateKernel, dim3((p.numPixels + 255) / 256), dim3(256), 0, stream, p);
}
void launchShade(const LaunchParams& p, int depth, int gridBlocks, hipStream_t stream)
{
  hipLaunchKernelGGL(shadeKernel, dim3(gridBlocks), dim3(256), 0, stream, p, depth);
}
void launchAccumulate(const LaunchParams& p, hipStream_t stream)
{
  hipLaunchKernelGGL(accumulateKernel, dim3((p.numPixels + 255) / 256), dim3(256), 0, stream, p);
}
void launchCompositor(const float4* tiles, float4* output, int width, int height, int launchWidth, int deviceCount,
                      int tileSizeX, int tileShiftX, int tileShiftY, hipStream_t stream)
{
  hipLaunchKernelGGL(compositorKernel, dim3((launchWidth + 255) / 256, height, deviceCount), dim3(256), 0, stream,
                     tiles, output, width, height, launchWidth, deviceCount, tileSizeX, tileShiftX, tileShiftY);
}
void launchMathTap(int op, const float* x, const float* y, float* out, size_t n, hipStream_t stream)
{
  hipLaunchKernelGGL(mathTapKernel, dim3(1024), dim3(256), 0, stream, op, x, y, out, n);
}
void launchStreamCopy(const float4* src, float4* dst, size_t n, hipStream_t stream)
{
  hipLaunchKernelGGL(streamCopyKernel, dim3(2048), dim3(256), 0, stream, src, dst, n);
}

} // namespace twk
