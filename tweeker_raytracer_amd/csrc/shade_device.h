// Device functions of the shading stage (shade_kernels.hip; trace_persistent.h takes tex2D / rng / tea from here).
//
// Reference programs restated (apps/rtigo3/shaders/): tea/rng random_number_generators.h:40-78, the BSDF callables
// bxdf_diffuse.cu / bxdf_specular.cu / bxdf_ggx_smith.cu, light sampling light_sample.cu, miss programs miss.cu:41-109,
// __closesthit__radiance closesthit.cu:126-305 and the integrator loop body raygeneration.cu:63-146.
// Per-path RNG draw order is the reference's: jitter rng2; per bounce the BSDF's draws, then NEE rng2
// [+ rng when more than one light], then Russian roulette rng.
#pragma once
#ifndef TWK_PROBE_GGX_AS_LAMBERT
#define TWK_PROBE_GGX_AS_LAMBERT 0
#endif
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

// Measurement builds of shadeKernel (MEASURE: twk_stats_enable, time view): for every phase of shadePath — how often a wave ran it,
// with how many of its lanes, and for how many shader-clock cycles (waits included). Tallied by the phase's first active lane in
// LDS words of the block ([0, N) wave executions, [N, 2N) lanes, [2N, 3N) cycles), flushed once per block to
// TwkLaunchStats::shadePhase*. Lane occupancy of a phase = lanes / (64 x wave executions). Compiled out of every other build.
#define TWK_SHADE_PHASES 24
static_assert(TWK_SHADE_PHASES == TWK_SHADE_PHASE_COUNT, "include/tweeker_hip.h TwkLaunchStats::shadePhase*");
enum ShadePhase
{
  SP_PATH = 0,        // the whole of shadePath
  SP_VOLUME_FETCH,    // volume stack top of a path inside a medium
  SP_MISS,            // miss programs
  SP_HIT_RECORD,      // instance + shading record + material, normals, front face
  SP_TANGENT,         // GGX only: tangent fetch + transform
  SP_TEXCOORD,        // textured materials only
  SP_LIGHT_HIT,       // emission + MIS weight of an implicit light hit
  SP_BSDF_DIFFUSE, SP_BSDF_MIRROR, SP_BSDF_GLASS, SP_BSDF_GGX, SP_BSDF_GGX_GLASS, // the five sample callables
  SP_NEE_SAMPLE,      // draws + light sampler
  SP_NEE_EVAL,        // BSDF eval + contribution of a usable light sample
  SP_RADIANCE,        // read-modify-write of the path's radiance word
  SP_TAIL,            // integrator loop tail: absorption, Russian roulette, volume stack, outputs
  SP_VOLUME_PUSH,     // glass transmission: push / pop of the volume stack
  SP_AOV,             // denoiser AOV writes
  SP_KERNEL_LOAD,     // shadeKernel: wait for the slot's streams (first use of the prefetched registers)
  SP_KERNEL_APPEND,   // shadeKernel: ballots, the block's two barriers and returning atomic, the queue writes
  SP_KERNEL_ITERATION,// shadeKernel: one block iteration, everything included
  SP_APPEND_BARRIER1, // ... of the append: from its start to behind the first barrier (the wait for the block's slowest wave)
  SP_APPEND_ATOMIC,   // ... the round trip of the block's returning atomic (per issuing lane)
  SP_APPEND_BARRIER2  // ... from the first barrier to behind the second (stream requests, the atomic, the wait for it)
};
template<bool MEASURE> struct PhaseScope
{
  TWK_D PhaseScope(unsigned int*, int) {}
  TWK_D void end() {}
};
#if defined(__HIPCC__)
template<> struct PhaseScope<true>
{
  unsigned int* word;
  unsigned int t0;
  TWK_D PhaseScope(unsigned int* lds, int phase)
  {
    const unsigned long long mask = __ballot(1);
    word = nullptr;
    if ((threadIdx.x & 63u) == (unsigned int) (__ffsll((long long) mask) - 1))
    {
      word = lds + phase;
      atomicAdd(word, 1u);
      atomicAdd(word + TWK_SHADE_PHASES, (unsigned int) __popcll(mask));
    }
    t0 = (unsigned int) __builtin_readcyclecounter();
  }
  TWK_D void end() // a phase that ends before its scope does
  {
    const unsigned int now = (unsigned int) __builtin_readcyclecounter();
    if (word != nullptr) { atomicAdd(word + 2 * TWK_SHADE_PHASES, now - t0); word = nullptr; }
  }
  TWK_D ~PhaseScope() { end(); }
};
#endif

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
template<bool MEASURE = false>
TWK_D void sampleBsdf(const DevMaterial& material, const SurfaceState& state, PathPrd& prd, unsigned int* phaseLds = nullptr)
{
#if TWK_PROBE_GGX_AS_LAMBERT // timing probe (wrong image): what the kernel costs without its rare expensive class
  switch (material.indexBSDF == 3 ? 0 : material.indexBSDF)
#else
  switch (material.indexBSDF)
#endif
  {
    default:
    case 0: // bxdf_diffuse.cu:67-86
    {
      PhaseScope<MEASURE> phase(phaseLds, SP_BSDF_DIFFUSE);
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
      PhaseScope<MEASURE> phase(phaseLds, SP_BSDF_MIRROR);
      prd.wi = reflect(-prd.wo, state.normal);
      if (dot(prd.wi, state.normalGeo) <= 0.0f) { prd.flags |= TWK_FLAG_TERMINATE; return; }
      prd.f_over_pdf = state.albedo;
      prd.pdf        = 1.0f;
      return;
    }
    case 2: // bxdf_specular.cu:94-134
    {
      PhaseScope<MEASURE> phase(phaseLds, SP_BSDF_GLASS);
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
      PhaseScope<MEASURE> phase(phaseLds, SP_BSDF_GGX);
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
      PhaseScope<MEASURE> phase(phaseLds, SP_BSDF_GGX_GLASS);
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
#if TWK_PROBE_GGX_AS_LAMBERT
  if (material.indexBSDF == 0 || material.indexBSDF == 3)
#else
  if (material.indexBSDF == 0) // bxdf_diffuse.cu:89-96
#endif
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

// ENV: the scene has the importance-sampled spherical environment (miss 2); compiled out of the kernel variant of scenes without it.
// Where shading reads the instance / material / light records: the scene's arrays, or a block's copies of them in LDS
// (shade_kernels.hip shadeKernel).
struct ShadeTables
{
  const DevInstance* instances;
  const DevMaterial* materials;
  const DevLight*    lights;
};

template<bool ENV>
TWK_D void sampleLight(const LaunchParams& p, const ShadeTables& tables, int index, const V3& point, float sx, float sy, LightSampleD& ls)
{
  const DevLight& light = tables.lights[index];
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
  if (ENV && p.miss == 2) // light_sample.cu:67-153 importance-sampled spherical environment
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
// Shading of one path segment (shadeKernel; the host build of the kernels, oracle/host_kernels.cpp):
// miss or closest-hit shading, next-event estimation, then the integrator's loop tail. Throughput, pdf, RNG state
// and flags come in and go out through `out` (they travel in the ray queue); radiance and the volume stack are
// per-path arrays; the continuation ray and the shadow ray are returned in registers.
struct ShadeOutput
{
  bool  alive;       // the path continues with (nextPos, nextDir)
  bool  wantShadow;  // a shadow ray (pos → shadowDir, tmax shadowTmax) decides whether `pending` is added
  V3    nextPos, nextDir;
  V3    shadowDir;
  float shadowTmax;
  V3    pending;     // throughput * MIS-weighted next-event contribution
  float4 throughputPdf; // in: the path's throughput.xyz and the pdf of its last BSDF sample; out (alive): the same after this bounce
  uint2  seedFlags;     // in / out: LCG state, path word (TWK_PATH_* bits)
  unsigned int shadowSeed; // cutout scenes: RNG stream of the shadow ray's any-hit draws, forked from the path's seed
};

//
// ENV / TEX: kernel variants (shade_kernels.hip launchShade). A scene without the spherical environment map (miss != 2) and
// without an albedo texture on any material runs a variant with both compiled out: 107 instead of 113 VGPRs, no scratch
// (the full kernel spills 9 dwords), 30 instead of 36 KB of code; measured on C2: shade 0.290 -> 0.273 ms per step.
// PRIMARY: the first segment of a path whose generateKernel was skipped (shade_kernels.hip "primary rays"): this call owns the
// path's radiance and AOV words and initialises them (raygeneration.cu:53-62) instead of adding to them.
// MEASURE: the measurement build (PhaseScope above); phaseLds = the block's tally words.
template<bool ENV = true, bool TEX = true, bool PRIMARY = false, bool MEASURE = false>
TWK_D void shadePath(const LaunchParams& p, const ShadeTables& tables, int depth, unsigned int pixel, const float4& ro, const float4& rd,
                     const float4& hit, int instanceIndex, ShadeOutput& out, unsigned int* phaseLds = nullptr)
{
  PhaseScope<MEASURE> phasePath(phaseLds, SP_PATH);
  if (PRIMARY && p.pathAlbedo != nullptr)
  {
    p.pathAlbedo[pixel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // Optix7Gui raygeneration.cu:66-71: black, null vector
    p.pathNormal[pixel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  const float4 tp = out.throughputPdf;
  const uint2  sf = out.seedFlags;
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
    PhaseScope<MEASURE> phase(phaseLds, SP_VOLUME_FETCH);
    prd.flags |= TWK_FLAG_VOLUME;
    const float4 top = p.volumeStack[(size_t) stackIdx * p.numPaths + pixel];
    prd.sigma_t = v3(top);
    prd.iorX    = top.w;
    if (TWK_MATERIAL_STACK_FIRST <= stackIdx - 1)
    {
      prd.iorY = p.volumeStack[(size_t) (stackIdx - 1) * p.numPaths + pixel].w;
    }
  }

  bool wantShadow = false;
  V3 shadowDir = v3(0.0f), contribution = v3(0.0f);
  float shadowTmax = 0.0f;
  V3 aovAlbedo = v3(0.0f), aovNormal = v3(0.0f); // prd.albedo / prd.normal of Optix7Gui's PerRayData (per_ray_data.h:111-114)

  if (instanceIndex < 0)
  {
    PhaseScope<MEASURE> phase(phaseLds, SP_MISS);
    // ---- miss programs, miss.cu
    if (p.miss == 0) { prd.radiance = v3(0.0f); }                                   // :41-52 (albedo 0)
    else if (ENV && p.miss == 2)                                                     // :75-109
    {
      const V3 R = prd.wi;
      const float u     = (atan2P(R.x, -R.z) + kPi) * 0.5f * kInvPi + p.envRotation;
      const float theta = acosP(-R.y);
      const float v     = theta * kInvPi;
      const V3 emission = v3(tex2D(p.textures[2], u, v));
      float weightMIS = 1.0f;
      if (p.nextEventEstimation && (prd.flags & TWK_FLAG_DIFFUSE))                   // miss.cu:92-106
      {
        const float pdfLight = intensity(emission) / p.envIntegral;
        weightMIS = powerHeuristic(prd.pdf, pdfLight);
      }
      prd.radiance = emission * weightMIS;
      aovAlbedo = emission;                                                          // Optix7Gui miss.cu:105-107
    }
    else                                                                             // :54-73
    {
      const float weightMIS = (p.nextEventEstimation && (prd.flags & TWK_FLAG_DIFFUSE)) ? powerHeuristic(prd.pdf, 0.25f * kInvPi) : 1.0f; // miss.cu:62-69
      prd.radiance = v3(weightMIS);
      aovAlbedo = v3(1.0f);                                                          // Optix7Gui miss.cu:67-69
    }
    prd.flags |= TWK_FLAG_LIGHT | TWK_FLAG_TERMINATE;
  }
  else
  {
    // ---- __closesthit__radiance, closesthit.cu:126-305
    PhaseScope<MEASURE> phaseHit(phaseLds, SP_HIT_RECORD);
    const DevInstance& inst = tables.instances[instanceIndex];
    // The slot's shading record (bvh_build.hip emitTrianglesKernel): geometric normal and vertex normals for every
    // hit; tangents and texture coordinates are fetched only by the materials that read them (the tangent feeds the
    // GGX tangent space only, closesthit.cu:161 computes it for all) — each fetch is one divergent lane address.
    const float4* sv = p.shadeTriangles + TWK_SHADE_RECORD * (size_t) __float_as_int(hit.w);
    const float4 s0 = sv[0], s1 = sv[1], s2 = sv[2];
    const DevMaterial& material = tables.materials[inst.material];
#if TWK_PROBE_GGX_AS_LAMBERT
    const bool needTangent  = material.indexBSDF >= 4;
#else
    const bool needTangent  = material.indexBSDF >= 3;
#endif
    const bool needTexcoord = TEX && material.textureAlbedo != 0;

    const float beta = hit.y, gamma = hit.z;
    const float alpha = 1.0f - beta - gamma;

    const V3 ng = v3(s0.x, s0.y, s0.z);                                               // closesthit.cu:150
    const V3 ns = v3(s0.w, s1.x, s1.y) * alpha + v3(s1.z, s1.w, s2.x) * beta + v3(s2.y, s2.z, s2.w) * gamma;

    SurfaceState state;
    state.texcoord = v3(0.0f);
    state.tangent  = v3(0.0f);
    if (needTangent)
    {
      PhaseScope<MEASURE> phase(phaseLds, SP_TANGENT);
      const float4 s3 = sv[3], s4 = sv[4], s5 = sv[5];
      const V3 tg = v3(s3.x, s3.y, s3.z) * alpha + v3(s3.w, s4.x, s4.y) * beta + v3(s4.z, s4.w, s5.x) * gamma;
      state.tangent = normalize(transformVector(inst.objectToWorld, tg));
    }
    if (needTexcoord)
    {
      PhaseScope<MEASURE> phase(phaseLds, SP_TEXCOORD);
      const float4 s5 = sv[5], s6 = sv[6], s7 = sv[7];
      state.texcoord = v3(s5.y, s5.z, s5.w) * alpha + v3(s6.x, s6.y, s6.z) * beta + v3(s6.w, s7.x, s7.y) * gamma;
    }

    state.normalGeo = normalize(transformNormal(inst.worldToObject, ng));
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

    aovNormal = state.normal; // Optix7Gui closesthit.cu:183-185: the normal on the side the ray looks at
    phaseHit.end();

    bool lightHit = false;
    if (0 <= inst.light && p.shaderVariant == TWK_SHADERS_OPTIX7GUI)
    {
      PhaseScope<MEASURE> phase(phaseLds, SP_LIGHT_HIT);
      // Optix7Gui closesthit.cu:189-226: a light ends the path whichever side is hit; black on the back face and edge-on
      V3 emission = v3(0.0f);
      const float cosTheta = dot(prd.wo, state.normalGeo);
      if ((prd.flags & TWK_FLAG_FRONTFACE) && DENOMINATOR_EPSILON < cosTheta)
      {
        const DevLight& light = tables.lights[inst.light];
        emission = v3(light.emission[0], light.emission[1], light.emission[2]);
        if (p.nextEventEstimation)                                                   // Optix7Gui closesthit.cu:202-214
        {
          const float lightPdf = (prd.distance * prd.distance) / (light.area * cosTheta);
          if ((prd.flags & TWK_FLAG_DIFFUSE) && DENOMINATOR_EPSILON < lightPdf)
          {
            emission = emission * powerHeuristic(prd.pdf, lightPdf);
          }
        }
      }
      prd.radiance = emission;
      aovAlbedo = emission;
      prd.flags |= TWK_FLAG_HIT | TWK_FLAG_LIGHT | TWK_FLAG_TERMINATE;
      lightHit = true;
    }
    else if (0 <= inst.light && (prd.flags & TWK_FLAG_FRONTFACE))
    {
      PhaseScope<MEASURE> phase(phaseLds, SP_LIGHT_HIT);
      // rtigo3 closesthit.cu:192-222: only the lit side ends the path, a back-face hit falls through to the light's (black specular) BSDF
      const float cosTheta = dot(prd.wo, state.normalGeo);
      if (DENOMINATOR_EPSILON < cosTheta)
      {
        const DevLight& light = tables.lights[inst.light];
        V3 emission = v3(light.emission[0], light.emission[1], light.emission[2]);
        if (p.nextEventEstimation)                                                   // closesthit.cu:202-214
        {
          const float lightPdf = (prd.distance * prd.distance) / (light.area * cosTheta);
          if ((prd.flags & TWK_FLAG_DIFFUSE) && DENOMINATOR_EPSILON < lightPdf)
          {
            emission = emission * powerHeuristic(prd.pdf, lightPdf);
          }
        }
        prd.radiance = emission;
        aovAlbedo = emission;
        prd.flags |= TWK_FLAG_LIGHT | TWK_FLAG_TERMINATE;
        lightHit = true;
      }
    }

    if (!lightHit)
    {
      prd.f_over_pdf = v3(0.0f);
      prd.pdf        = 0.0f;

      state.albedo = v3(material.albedo[0], material.albedo[1], material.albedo[2]);
      if (TEX && material.textureAlbedo != 0)
      {
        const V3 texColor = v3(tex2D(p.textures[0], state.texcoord.x, state.texcoord.y));
        state.albedo = state.albedo * texColor;
      }
      aovAlbedo = state.albedo;                                                      // Optix7Gui closesthit.cu:247-249

      prd.flags = (prd.flags & ~TWK_FLAG_DIFFUSE) | TWK_FLAG_HIT | material.flags;

      sampleBsdf<MEASURE>(material, state, prd, phaseLds);

      const int numLights = p.numLights;
      if (p.nextEventEstimation && (prd.flags & TWK_FLAG_DIFFUSE) && 0 < numLights) // closesthit.cu:250-304
      {
        PhaseScope<MEASURE> phaseSample(phaseLds, SP_NEE_SAMPLE);
        const float sx = rng(prd.seed);
        const float sy = rng(prd.seed);
        const int lightIndex = (1 < numLights) ? min(max(static_cast<int>(floorf(rng(prd.seed) * numLights)), 0), numLights - 1) : 0;

        LightSampleD ls;
        sampleLight<ENV>(p, tables, lightIndex, prd.pos, sx, sy, ls);
        phaseSample.end();

        if (0.0f < ls.pdf)
        {
          PhaseScope<MEASURE> phase(phaseLds, SP_NEE_EVAL);
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
  PhaseScope<MEASURE> phaseTail(phaseLds, SP_TAIL);
  if (prd.flags & TWK_FLAG_VOLUME)
  {
    throughput = throughput * exp3(-prd.distance * prd.sigma_t);
  }


  unsigned int albedoWritten = 0u;
  if (p.pathAlbedo != nullptr)
  {
    PhaseScope<MEASURE> phase(phaseLds, SP_AOV);
    // Denoiser AOVs, Optix7Gui raygeneration.cu:125-164: the albedo of the first diffuse or light event, attenuated by
    // the throughput up to it (after this segment's absorption, before this bounce's BSDF weight); the shading normal
    // of the primary hit in a right-handed camera space.
    if (!(prd.flags & TWK_FLAG_ALBEDO) && (prd.flags & (TWK_FLAG_DIFFUSE | TWK_FLAG_LIGHT)))
    {
      const V3 a = throughput * aovAlbedo;
      p.pathAlbedo[pixel] = make_float4(fmaxf(0.0f, fminf(a.x, 1.0f)), fmaxf(0.0f, fminf(a.y, 1.0f)), fmaxf(0.0f, fminf(a.z, 1.0f)), 1.0f); // clamp(), vector_math.h:148-151
      albedoWritten = TWK_FLAG_ALBEDO;
    }
    if (depth == 0 && instanceIndex >= 0)
    {
      const float* cam = p.camera;
      const V3 U = normalize(v3(cam[3], cam[4], cam[5])), V = normalize(v3(cam[6], cam[7], cam[8])), W = normalize(v3(cam[9], cam[10], cam[11]));
      p.pathNormal[pixel] = make_float4(dot(aovNormal, U), dot(aovNormal, V), -dot(aovNormal, W), 0.0f);
    }
  }

  out.wantShadow = wantShadow;
  out.shadowDir = shadowDir; out.shadowTmax = shadowTmax;
  out.pending = throughput * contribution;
  out.shadowSeed = (wantShadow && p.hasCutout) ? tea<2>(prd.seed, 0x53484457u /* 'SHDW' */) : 0u;
  if (PRIMARY)
  {
    // black radiance (raygeneration.cu:55) + what this segment adds: 0.0f + x is x, bit for bit, unless x is -0.0f
    const V3 add = wantShadow ? v3(0.0f) : throughput * prd.radiance;
    p.pathRadiance[pixel] = make_float4(0.0f + add.x, 0.0f + add.y, 0.0f + add.z, 1.0f);
  }
  else if (!wantShadow)
  {
    // emission / environment (or nothing): radiance += throughput * prd.radiance. A segment that adds exactly zero (a specular
    // bounce, a surface without emission) leaves the word alone: the sum starts at +0 and only takes non-negative addends,
    // so x + 0 is x bit for bit (a NaN addend is not zero and is added).
    const V3 add = throughput * prd.radiance;
    if (add.x != 0.0f || add.y != 0.0f || add.z != 0.0f)
    {
      PhaseScope<MEASURE> phase(phaseLds, SP_RADIANCE);
      float4 r = p.pathRadiance[pixel];
      r.x += add.x; r.y += add.y; r.z += add.z;
      p.pathRadiance[pixel] = r;
    }
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
      PhaseScope<MEASURE> phase(phaseLds, SP_VOLUME_PUSH);
      if (prd.flags & TWK_FLAG_FRONTFACE)
      {
        stackIdx = min(stackIdx + 1, TWK_MATERIAL_STACK_LAST);
        p.volumeStack[(size_t) stackIdx * p.numPaths + pixel] = prd.absorption_ior;
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
    out.throughputPdf = make_float4(throughput.x, throughput.y, throughput.z, prd.pdf);
    out.seedFlags     = make_uint2(prd.seed, ((prd.flags | albedoWritten) & TWK_FLAG_CLEAR_MASK) | ((unsigned int) (stackIdx + 1) << TWK_PATH_STACK_SHIFT));
  }
  out.alive = alive;
  out.nextPos = prd.pos; out.nextDir = prd.wi;
}

// The same with the records read from the scene's arrays (host build of the kernels).
template<bool ENV = true, bool TEX = true, bool PRIMARY = false>
TWK_D void shadePath(const LaunchParams& p, int depth, unsigned int pixel, const float4& ro, const float4& rd,
                     const float4& hit, int instanceIndex, ShadeOutput& out)
{
  ShadeTables tables;
  tables.instances = p.instances; tables.materials = p.materials; tables.lights = p.lights;
  shadePath<ENV, TEX, PRIMARY>(p, tables, depth, pixel, ro, rd, hit, instanceIndex, out);
}

// ---------------------------------------------------------------------------------------------
// The primary ray of path `index` of the launch (raygeneration.cu:167-203 up to the first optixTrace, lens_shader.cu): seed,
// jitter, lens shader. generateKernel stores it; on the fused path (shade_kernels.hip "primary rays") the first traversal and
// the first shade launch each compute it instead of passing it through memory — the same function, the same bits.
struct PrimaryRay { V3 origin, direction; unsigned int seed; bool active; };

TWK_D PrimaryRay primaryRay(const LaunchParams& p, const unsigned int index)
{
  const unsigned int path = index + (unsigned int) p.pathBase; // path of the whole pass; `index` counts within this launch's lane
  const unsigned int sampleIndex = path / (unsigned int) p.numPixels;
  const unsigned int launchIndex = path - sampleIndex * (unsigned int) p.numPixels;
  const unsigned int lx = launchIndex % (unsigned int) p.launchWidth;
  const unsigned int ly = launchIndex / (unsigned int) p.launchWidth;

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
    seed = tea<4>((unsigned int) p.resolution[0] * ly + launchColumn, p.iterationIndex + sampleIndex);

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

  PrimaryRay r;
  r.origin = origin; r.direction = direction; r.seed = seed; r.active = active;
  return r;
}

// Body of generateKernel for path `index` of the launch (shade_kernels.hip; also what the host build of the kernels runs:
// oracle/host_kernels.cpp).
TWK_D void generatePath(const LaunchParams& p, const unsigned int index)
{
  const PrimaryRay ray = primaryRay(p, index);
  const V3 origin = ray.origin, direction = ray.direction;
  const unsigned int seed = ray.seed;
  const bool active = ray.active;

  // integrator prologue (raygeneration.cu:53-62): black radiance, unit throughput, empty volume stack
  p.pathRadiance[index]   = make_float4(0.0f, 0.0f, 0.0f, active ? 1.0f : 0.0f);
  if (p.pathAlbedo != nullptr)
  {
    p.pathAlbedo[index] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // Optix7Gui raygeneration.cu:66-71: black, null vector
    p.pathNormal[index] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  p.rayThroughput[0][index] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
  p.raySeedFlags[0][index]  = make_uint2(seed, 0u);

  // Inactive launch indices (tile columns beyond the image) still own a slot so that queue 0 is the
  // identity mapping; they carry tmax < tmin and never hit anything, and shade drops them.
  p.rayOrg[0][index]   = make_float4(origin.x, origin.y, origin.z, p.sceneEpsilon);
  p.rayDir[0][index]   = make_float4(direction.x, direction.y, direction.z, active ? RT_DEFAULT_MAX : -1.0f);
  p.rayPixel[0][index] = index;
  if (index == 0) p.counters[0] = (unsigned int) p.numPaths;
}

// Body of accumulateKernel for launch index `index` (raygeneration.cu:222-253).
TWK_D void accumulateLaunchIndex(const LaunchParams& p, const unsigned int index)
{
  const bool aov = (p.aovAlbedo != nullptr);
  // Where this launch index accumulates: its slot of the packed tile buffer (single device, LocalCopy), or — shared
  // frame of the ZeroCopy / PeerAccess strategies — the pixel it maps to, as __raygen__path_tracer addresses
  // sysData.outputBuffer (raygeneration.cu:175-183,229): index = y * W + distribute(launch index)
  size_t outIndex = index;
  if (p.outputFrame)
  {
    const unsigned int lx = index % (unsigned int) p.launchWidth, ly = index / (unsigned int) p.launchWidth;
    const unsigned int column = (p.distribution && 1 < p.deviceCount) ? distribute(p, lx, ly) : lx;
    if (column >= (unsigned int) p.resolution[0]) return;
    outIndex = (size_t) ly * (unsigned int) p.resolution[0] + column;
  }
  float4 dst = p.output[outIndex];
  float4 dstAlbedo = aov ? p.aovAlbedo[index] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4 dstNormal = aov ? p.aovNormal[index] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  bool touched = false;
  for (int s = 0; s < p.batchCount; ++s)
  {
    const size_t path = (size_t) s * p.numPixels + index;
    const float4 r = p.pathRadiance[path];
    if (r.w == 0.0f) continue; // launch index outside the image (tile padding): never written, like the early return at raygeneration.cu:180-183
    V3 radiance = v3(r.x, r.y, r.z);
    bool keep = !(isnan(radiance.x) || isnan(radiance.y) || isnan(radiance.z));
    if (p.debugExceptions) // raygeneration.cu:205-218: numerical errors in false colours, and every sample is accumulated
    {
      if (!keep)                                                             radiance = v3(1000000.0f, 0.0f, 0.0f); // super red
      else if (isinf(radiance.x) || isinf(radiance.y) || isinf(radiance.z)) radiance = v3(0.0f, 1000000.0f, 0.0f); // super green
      else if (radiance.x < 0.0f || radiance.y < 0.0f || radiance.z < 0.0f) radiance = v3(0.0f, 0.0f, 1000000.0f); // super blue
      keep = true;
    }
    if (keep)
    {
      const unsigned int iteration = p.iterationIndex + (unsigned int) s;
      V3 albedo = v3(0.0f), normal = v3(0.0f);
      if (aov) { albedo = v3(p.pathAlbedo[path]); normal = v3(p.pathNormal[path]); }
      // time view (raygeneration.cu:231-244): alpha = the sample's clock cycles * clockScale, accumulated like the radiance
      float alpha = (p.pathTime != nullptr) ? p.pathTime[path] * p.clockScale : 1.0f;
      if (0 < iteration)
      {
        const float t = 1.0f / float(iteration + 1);
        radiance = lerp(v3(dst.x, dst.y, dst.z), radiance, t);
        if (p.pathTime != nullptr) alpha = dst.w + t * (alpha - dst.w); // lerp(dst, result, t), fourth component
        if (aov)
        {
          // Optix7Gui raygeneration.cu:243-252: same running mean; the mean normal is renormalised unless it vanished
          albedo = lerp(v3(dstAlbedo), albedo, t);
          normal = lerp(v3(dstNormal), normal, t);
          if (isNotNull(normal)) normal = normalize(normal);
        }
      }
      dst = make_float4(radiance.x, radiance.y, radiance.z, alpha);
      dstAlbedo = make_float4(albedo.x, albedo.y, albedo.z, 1.0f);
      dstNormal = make_float4(normal.x, normal.y, normal.z, 0.0f);
      touched = true;
    }
  }
  if (touched)
  {
    p.output[outIndex] = dst;
    if (aov) { p.aovAlbedo[index] = dstAlbedo; p.aovNormal[index] = dstNormal; }
  }
}

} // namespace twk
