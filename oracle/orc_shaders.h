// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// CPU restatement of the reference's device programs (apps/rtigo3/shaders/*.cu), one function per
// program, same control flow, same float evaluation order, same RNG draw order. Each function cites
// the file:line it follows. The OptiX megakernel structure is kept on purpose (one PerRayData per
// sample, recursion through trace()), so this oracle does NOT share structure or code with the
// wavefront HIP renderer it checks.
//
// Pinning status: the reference holds no tests, golden images or known-answer vectors
// (SURVEY.md §4, §8c). The helper layer (tea/rng, refract, TBN, vector maths, mesh generators,
// camera frustum, tile maps) is validated bit for bit against the reference's own sources compiled
// in oracle/_ref (see oracle/Makefile, tests/test_oracle_vs_ref.py); the .cu programs themselves
// #include <optix.h>, which this image lacks, so they cannot be compiled here and the BSDF /
// integrator restatement below is "parity unpinned" beyond review against the cited lines.
#pragma once
#include "orc_types.h"
#include "orc_trace.h"

namespace orc {

// ---------------------------------------------------------------------------------------------
// shaders/random_number_generators.h:40-53
template<unsigned int N>
static inline unsigned int tea(const unsigned int val0, const unsigned int val1)
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
static inline float rng(unsigned int& previous)
{
  previous = previous * 1664525u + 1013904223u;
  return float(previous & 0x00FFFFFF) / float(0x01000000u);
}

// shaders/random_number_generators.h:65-78
static inline float2 rng2(unsigned int& previous)
{
  float2 s;
  previous = previous * 1664525u + 1013904223u;
  s.x = float(previous & 0x00FFFFFF) / float(0x01000000u);
  previous = previous * 1664525u + 1013904223u;
  s.y = float(previous & 0x00FFFFFF) / float(0x01000000u);
  return s;
}

// ---------------------------------------------------------------------------------------------
// shaders/shader_common.h:47-77
static inline bool refract(float3& r, float3 const& i, float3 const& n, const float ior)
{
  float3 nn = n;
  float negNdotV = dot(i, nn);
  float eta;
  if (negNdotV > 0.0f)
  {
    eta = ior;
    nn = -n;
    negNdotV = -negNdotV;
  }
  else
  {
    eta = 1.f / ior;
  }
  const float k = 1.f - eta * eta * (1.f - negNdotV * negNdotV);
  if (k < 0.0f)
  {
    r = make_float3(0.f);
    return false;
  }
  else
  {
    r = normalize(eta * i - (eta * negNdotV + sqrtf(k)) * nn);
    return true;
  }
}

// shaders/shader_common.h:82-148 (only the constructor the BSDFs use, :119-125)
struct TBN
{
  TBN(const float3& tangent_reference, const float3& n) : normal(n)
  {
    bitangent = normalize(cross(normal, tangent_reference));
    tangent   = cross(bitangent, normal);
  }
  float3 transformToLocal(const float3& p) const { return make_float3(dot(p, tangent), dot(p, bitangent), dot(p, normal)); }
  float3 transformToWorld(const float3& p) const { return p.x * tangent + p.y * bitangent + p.z * normal; }
  float3 tangent, bitangent, normal;
};

// shader_common.h:157-160, :172-187
static inline float intensity(const float3& rgb) { return (rgb.x + rgb.y + rgb.z) * 0.3333333333f; }
static inline bool isNull(const float3& v) { return (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f); }
static inline bool isNotNull(const float3& v) { return (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f); }
static inline float powerHeuristic(const float a, const float b) { const float t = a * a; return t / (t + b * b); }

// ---------------------------------------------------------------------------------------------
// tex2D<float4> on an RGBA32F image: normalized coordinates, bilinear, wrap (v clamps for the
// environment) — src/Texture.cpp:668-693,1353. The hardware's 9-bit weight quantisation is not modelled.
static inline float4 tex2D(const Texture& tex, float u, float v)
{
  const int W = tex.width, H = tex.height;
  u = u - floorf(u);
  const float xB = u * float(W) - 0.5f;
  const float xf = floorf(xB);
  const float a  = xB - xf;
  int i0 = (int) xf; int i1 = i0 + 1;
  i0 = ((i0 % W) + W) % W; i1 = ((i1 % W) + W) % W;
  float yB; int j0, j1; float b;
  if (tex.clampV)
  {
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    yB = v * float(H) - 0.5f;
    const float yf = floorf(yB);
    b = yB - yf;
    j0 = clampi((int) yf, 0, H - 1); j1 = clampi((int) yf + 1, 0, H - 1);
  }
  else
  {
    v = v - floorf(v);
    yB = v * float(H) - 0.5f;
    const float yf = floorf(yB);
    b = yB - yf;
    j0 = (int) yf; j1 = j0 + 1;
    j0 = ((j0 % H) + H) % H; j1 = ((j1 % H) + H) % H;
  }
  const float4& t00 = tex.texels[(size_t) j0 * W + i0];
  const float4& t10 = tex.texels[(size_t) j0 * W + i1];
  const float4& t01 = tex.texels[(size_t) j1 * W + i0];
  const float4& t11 = tex.texels[(size_t) j1 * W + i1];
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  float4 r;
  r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
  r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
  r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
  r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
  return r;
}

// ---------------------------------------------------------------------------------------------
// Lens shaders — shaders/lens_shader.cu
// :40-52
static inline void lens_pinhole(const SystemData& sysData, const float2 screen, const float2 pixel, const float2 sample, float3& origin, float3& direction)
{
  const float2 fragment = pixel + sample;
  const float2 ndc      = (fragment / screen) * 2.0f - 1.0f;
  const CameraDefinition camera = sysData.cameraDefinitions[0];
  origin    = camera.P;
  direction = normalize(camera.U * ndc.x + camera.V * ndc.y + camera.W);
}

// :55-73
static inline void lens_fisheye(const SystemData& sysData, const float2 screen, const float2 pixel, const float2 sample, float3& origin, float3& direction)
{
  const float2 fragment = pixel + sample;
  const float2 center = screen * 0.5f;
  const float2 uv     = (fragment - center) / length(center);
  const float z       = pm_cosf(length(uv) * 0.7071067812f * 0.5f * M_PIf_);
  const CameraDefinition camera = sysData.cameraDefinitions[0];
  const float3 U = normalize(camera.U);
  const float3 V = normalize(camera.V);
  const float3 W = normalize(camera.W);
  origin    = camera.P;
  direction = normalize(uv.x * U + uv.y * V + z * W);
}

// :76-99
static inline void lens_sphere(const SystemData& sysData, const float2 screen, const float2 pixel, const float2 sample, float3& origin, float3& direction)
{
  const float2 uv = (pixel + sample) / screen;
  const float phi   = uv.x * 2.0f * M_PIf_;
  const float theta = uv.y * M_PIf_;
  const float sinTheta = pm_sinf(theta);
  const float3 v = make_float3(-pm_sinf(phi) * sinTheta, -pm_cosf(theta), -pm_cosf(phi) * sinTheta);
  const CameraDefinition camera = sysData.cameraDefinitions[0];
  const float3 U = normalize(camera.U);
  const float3 V = normalize(camera.V);
  const float3 W = normalize(camera.W);
  origin    = camera.P;
  direction = normalize(v.x * U + v.y * V + v.z * W);
}

// ---------------------------------------------------------------------------------------------
// Light sampling — shaders/light_sample.cu
// :40-51
static inline void unitSquareToSphere(const float u, const float v, float3& p, float& pdf)
{
  p.z = 1.0f - 2.0f * u;
  float r = 1.0f - p.z * p.z;
  r = (0.0f < r) ? sqrtf(r) : 0.0f;
  const float phi = v * 2.0f * M_PIf_;
  p.x = r * pm_cosf(phi);
  p.y = r * pm_sinf(phi);
  pdf = 0.25f * M_1_PIf_;
}

// :55-65
static inline void light_env_constant(const SystemData& sysData, float3 const& point, const float2 sample, LightSample& lightSample)
{
  (void) point;
  unitSquareToSphere(sample.x, sample.y, lightSample.direction, lightSample.pdf);
  lightSample.distance = RT_DEFAULT_MAX;
  lightSample.emission = make_float3(float(sysData.numLights));
}

// :67-153
static inline void light_env_sphere(const SystemData& sysData, float3 const& point, const float2 sample, LightSample& lightSample)
{
  (void) point;
  const unsigned int sizeV = sysData.envHeight;
  unsigned int ilo = 0;
  unsigned int ihi = sizeV;
  const float* cdfV = sysData.envCDF_V.data();
  while (ilo != ihi - 1)
  {
    const unsigned int i = (ilo + ihi) >> 1;
    if (sample.y < cdfV[i]) ihi = i; else ilo = i;
  }
  const unsigned int vIdx = ilo;
  const unsigned int sizeU = sysData.envWidth;
  ilo = 0;
  ihi = sizeU;
  const float* cdfU = &sysData.envCDF_U[(size_t) vIdx * (sizeU + 1)];
  while (ilo != ihi - 1)
  {
    const unsigned int i = (ilo + ihi) >> 1;
    if (sample.x < cdfU[i]) ihi = i; else ilo = i;
  }
  const unsigned int uIdx = ilo;
  const float cdfLowerU = cdfU[uIdx];
  const float cdfUpperU = cdfU[uIdx + 1];
  const float du = (sample.x - cdfLowerU) / (cdfUpperU - cdfLowerU);
  const float cdfLowerV = cdfV[vIdx];
  const float cdfUpperV = cdfV[vIdx + 1];
  const float dv = (sample.y - cdfLowerV) / (cdfUpperV - cdfLowerV);
  const float u = (float(uIdx) + du) / float(sizeU);
  const float v = (float(vIdx) + dv) / float(sizeV);
  const float phi   = (u - sysData.envRotation) * 2.0f * M_PIf_;
  const float theta = v * M_PIf_;
  const float sinTheta = pm_sinf(theta);
  lightSample.direction = make_float3(-pm_sinf(phi) * sinTheta, -pm_cosf(theta), pm_cosf(phi) * sinTheta);
  lightSample.distance = RT_DEFAULT_MAX;
  const float3 emission = make_float3(tex2D(sysData.textures[2], u, v));
  lightSample.emission = emission * float(sysData.numLights);
  lightSample.pdf = intensity(emission) / sysData.envIntegral;
}

// :156-177
static inline void light_parallelogram(const SystemData& sysData, float3 const& point, const float2 sample, LightSample& lightSample)
{
  lightSample.pdf = 0.0f;
  LightDefinition const& light = sysData.lightDefinitions[lightSample.index];
  lightSample.position  = light.position + light.vecU * sample.x + light.vecV * sample.y;
  lightSample.direction = lightSample.position - point;
  lightSample.distance  = length(lightSample.direction);
  if (DENOMINATOR_EPSILON < lightSample.distance)
  {
    lightSample.direction /= lightSample.distance;
    const float cosTheta = dot(-lightSample.direction, light.normal);
    if (DENOMINATOR_EPSILON < cosTheta)
    {
      lightSample.emission = light.emission * float(sysData.numLights);
      lightSample.pdf      = (lightSample.distance * lightSample.distance) / (light.area * cosTheta);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BSDFs — shaders/bxdf_diffuse.cu
// :39-47
static inline void alignVector(float3 const& axis, float3& w)
{
  const float s = copysignf(1.0f, axis.z);
  w.z *= s;
  const float3 h = make_float3(axis.x, axis.y, axis.z + s);
  const float  k = dot(w, h) / (1.0f + fabsf(axis.z));
  w = k * h - w;
}

// :49-63
static inline void unitSquareToCosineHemisphere(const float2 sample, float3 const& axis, float3& w, float& pdf)
{
  const float theta = 2.0f * M_PIf_ * sample.x;
  const float r = sqrtf(sample.y);
  w.x = r * pm_cosf(theta);
  w.y = r * pm_sinf(theta);
  w.z = 1.0f - w.x * w.x - w.y * w.y;
  w.z = (0.0f < w.z) ? sqrtf(w.z) : 0.0f;
  pdf = w.z * M_1_PIf_;
  alignVector(axis, w);
}

// :67-86
static inline void sample_brdf_diffuse(MaterialDefinition const& material, State const& state, PerRayData* prd)
{
  (void) material;
  unitSquareToCosineHemisphere(rng2(prd->seed), state.normal, prd->wi, prd->pdf);
  if (prd->pdf <= 0.0f || dot(prd->wi, state.normalGeo) <= 0.0f)
  {
    prd->flags |= FLAG_TERMINATE;
    return;
  }
  prd->f_over_pdf = state.albedo;
  prd->flags |= FLAG_DIFFUSE;
}

// :89-96
static inline float4 eval_brdf_diffuse(MaterialDefinition const& material, State const& state, PerRayData* const prd, float3 const& wiL)
{
  (void) material; (void) prd;
  const float3 f   = state.albedo * M_1_PIf_;
  const float  pdf = fmaxf(0.0f, dot(wiL, state.normal) * M_1_PIf_);
  return make_float4(f, pdf);
}

// shaders/bxdf_specular.cu:42-67 (identical copy at bxdf_ggx_smith.cu:44-69)
static inline float evaluateFresnelDielectric(const float et, const float cosIn)
{
  const float cosi = fabsf(cosIn);
  float sint = 1.0f - cosi * cosi;
  sint = (0.0f < sint) ? sqrtf(sint) / et : 0.0f;
  if (1.0f < sint)
  {
    return 1.0f;
  }
  float cost = 1.0f - sint * sint;
  cost = (0.0f < cost) ? sqrtf(cost) : 0.0f;
  const float et_cosi = et * cosi;
  const float et_cost = et * cost;
  const float rPerpendicular = (cosi - et_cost) / (cosi + et_cost);
  const float rParallel      = (et_cosi - cost) / (et_cosi + cost);
  const float result = (rParallel * rParallel + rPerpendicular * rPerpendicular) * 0.5f;
  return (result <= 1.0f) ? result : 1.0f;
}

// shaders/bxdf_specular.cu:71-83
static inline void sample_brdf_specular(MaterialDefinition const& material, State const& state, PerRayData* prd)
{
  (void) material;
  prd->wi = reflect(-prd->wo, state.normal);
  if (dot(prd->wi, state.normalGeo) <= 0.0f)
  {
    prd->flags |= FLAG_TERMINATE;
    return;
  }
  prd->f_over_pdf = state.albedo;
  prd->pdf        = 1.0f;
}

// shaders/bxdf_specular.cu:87-90 — shared by every specular eval (src/Device.cpp:744-748,768-772)
static inline float4 eval_brdf_specular(MaterialDefinition const&, State const&, PerRayData* const, float3 const&)
{
  return make_float4(0.0f);
}

// shaders/bxdf_specular.cu:94-134
static inline void sample_bsdf_specular(MaterialDefinition const& material, State const& state, PerRayData* prd)
{
  prd->absorption_ior = make_float4(material.absorption, material.ior);
  const float eta = (prd->flags & (FLAG_FRONTFACE | FLAG_THINWALLED))
                    ? prd->absorption_ior.w / prd->ior.x
                    : prd->ior.y / prd->absorption_ior.w;
  const float3 R = reflect(-prd->wo, state.normal);
  float reflective = 1.0f;
  if (refract(prd->wi, -prd->wo, state.normal, eta))
  {
    if (prd->flags & FLAG_THINWALLED)
    {
      prd->wi = -prd->wo;
    }
    reflective = evaluateFresnelDielectric(eta, dot(prd->wo, state.normal));
  }
  const float pseudo = rng(prd->seed);
  if (pseudo < reflective)
  {
    prd->wi = R;
  }
  else if (!(prd->flags & FLAG_THINWALLED))
  {
    prd->flags |= FLAG_TRANSMISSION;
  }
  prd->f_over_pdf = state.albedo;
  prd->pdf        = 1.0f;
}

// shaders/bxdf_ggx_smith.cu:74-94
static inline float2 distribution_d_pdf(const float ax, const float ay, float3 const& wm)
{
  if (DENOMINATOR_EPSILON < wm.z)
  {
    const float cosThetaSqr = wm.z * wm.z;
    const float tanThetaSqr = (1.0f - cosThetaSqr) / cosThetaSqr;
    const float phiM    = pm_atan2f(wm.y, wm.x);
    const float cosPhiM = pm_cosf(phiM);
    const float sinPhiM = pm_sinf(phiM);
    const float term = 1.0f + tanThetaSqr * ((cosPhiM * cosPhiM) / (ax * ax) + (sinPhiM * sinPhiM) / (ay * ay));
    const float d   = 1.0f / (M_PIf_ * ax * ay * cosThetaSqr * cosThetaSqr * term * term);
    const float pdf = d * wm.z;
    return make_float2(d, pdf);
  }
  return make_float2(0.0f);
}

// :96-106
static inline float3 distribution_sample(const float ax, const float ay, const float u1, const float u2)
{
  const float theta    = pm_atanf(ay * sqrtf(u1) / sqrtf(1.0f - u1));
  const float phi      = 2.0f * M_PIf_ * u2;
  const float sinTheta = pm_sinf(theta);
  return normalize(make_float3(pm_cosf(phi) * sinTheta * ax / ay, pm_sinf(phi) * sinTheta, pm_cosf(theta)));
}

// :109-125
static inline float smith_G1(const float alpha, float3 const& w, float3 const& wm)
{
  const float w_wm = dot(w, wm);
  if (w_wm * w.z <= 0.0f)
  {
    return 0.0f;
  }
  const float cosThetaSqr = w.z * w.z;
  const float sinThetaSqr = 1.0f - cosThetaSqr;
  const float tanThetaSqr = (0.0f < sinThetaSqr) ? sinThetaSqr / cosThetaSqr : 0.0f;
  const float invASqr = alpha * alpha * tanThetaSqr;
  return 2.0f / (1.0f + sqrtf(1.0f + invASqr));
}

// :150-165
static inline float distribution_G(const float ax, const float ay, float3 const& wo, float3 const& wi, float3 const& wm)
{
  float phi   = pm_atan2f(wo.y, wo.x);
  float c     = pm_cosf(phi);
  float s     = pm_sinf(phi);
  float alpha = sqrtf(c * c * ax * ax + s * s * ay * ay);
  const float g = smith_G1(alpha, wo, wm);
  phi   = pm_atan2f(wi.y, wi.x);
  c     = pm_cosf(phi);
  s     = pm_sinf(phi);
  alpha = sqrtf(c * c * ax * ax + s * s * ay * ay);
  return g * smith_G1(alpha, wi, wm);
}

// :169-222
static inline void sample_brdf_ggx_smith(MaterialDefinition const& material, State const& state, PerRayData* prd)
{
  const float2 sample = rng2(prd->seed);
  const float3 wm = distribution_sample(material.roughness.x, material.roughness.y, sample.x, sample.y);
  const TBN tangentSpace(state.tangent, state.normal);
  const float3 wh = tangentSpace.transformToWorld(wm);
  prd->wi = reflect(-prd->wo, wh);
  if (dot(prd->wi, state.normalGeo) <= 0.0f)
  {
    prd->flags |= FLAG_TERMINATE;
    return;
  }
  const float3 wo = tangentSpace.transformToLocal(prd->wo);
  const float3 wi = tangentSpace.transformToLocal(prd->wi);
  const float wi_wh = dot(prd->wi, wh);
  if (wo.z <= 0.0f || wi.z <= 0.0f || wi_wh <= 0.0f)
  {
    prd->flags |= FLAG_TERMINATE;
    return;
  }
  const float2 D_PDF = distribution_d_pdf(material.roughness.x, material.roughness.y, wm);
  if (D_PDF.y <= 0.0f)
  {
    prd->flags |= FLAG_TERMINATE;
    return;
  }
  const float G = distribution_G(material.roughness.x, material.roughness.y, wo, wi, wm);
  prd->pdf = D_PDF.y / (4.0f * wi_wh);
  prd->f_over_pdf = state.albedo * (G * D_PDF.x * wi_wh / (D_PDF.y * wo.z));
  prd->flags |= FLAG_DIFFUSE;
}

// :226-261
static inline float4 eval_brdf_ggx_smith(MaterialDefinition const& material, State const& state, PerRayData* const prd, float3 const& wiL)
{
  const TBN tangentSpace(state.tangent, state.normal);
  const float3 wo = tangentSpace.transformToLocal(prd->wo);
  const float3 wi = tangentSpace.transformToLocal(wiL);
  if (wo.z <= 0.0f || wi.z <= 0.0f)
  {
    return make_float4(0.0f);
  }
  float3 wm = wo + wi;
  if (isNull(wm))
  {
    return make_float4(0.0f);
  }
  wm = normalize(wm);
  const float2 D_PDF = distribution_d_pdf(material.roughness.x, material.roughness.y, wm);
  const float G = distribution_G(material.roughness.x, material.roughness.y, wo, wi, wm);
  const float3 f = state.albedo * (D_PDF.x * G / (4.0f * wo.z * wi.z));
  const float pdf = D_PDF.y / (4.0f * dot(wi, wm));
  return make_float4(f, pdf);
}

// :265-319
static inline void sample_bsdf_ggx_smith(MaterialDefinition const& material, State const& state, PerRayData* prd)
{
  prd->absorption_ior = make_float4(material.absorption, material.ior);
  const float eta = (prd->flags & (FLAG_FRONTFACE | FLAG_THINWALLED))
                  ? prd->absorption_ior.w / prd->ior.x
                  : prd->ior.y / prd->absorption_ior.w;
  const float2 sample = rng2(prd->seed);
  const float3 wm = distribution_sample(material.roughness.x, material.roughness.y, sample.x, sample.y);
  const TBN tangentSpace(state.tangent, state.normal);
  const float3 wh = tangentSpace.transformToWorld(wm);
  const float3 R = reflect(-prd->wo, wh);
  float reflective = 1.0f;
  if (refract(prd->wi, -prd->wo, wh, eta))
  {
    if (prd->flags & FLAG_THINWALLED)
    {
      prd->wi = reflect(R, state.normal);
    }
    reflective = evaluateFresnelDielectric(eta, dot(prd->wo, wh));
  }
  const float pseudo = rng(prd->seed);
  if (pseudo < reflective)
  {
    prd->wi = R;
  }
  else if (!(prd->flags & FLAG_THINWALLED))
  {
    prd->flags |= FLAG_TRANSMISSION;
  }
  prd->f_over_pdf = state.albedo;
  prd->pdf        = 1.0f;
}

// Callable tables ≙ SBT direct callables in ProgramGroupId order (inc/Device.h:216-232).
static inline void callLens(const SystemData& s, int idx, const float2 screen, const float2 pixel, const float2 sample, float3& o, float3& d)
{
  switch (idx)
  {
    default:
    case 0: lens_pinhole(s, screen, pixel, sample, o, d); break;
    case 1: lens_fisheye(s, screen, pixel, sample, o, d); break;
    case 2: lens_sphere(s, screen, pixel, sample, o, d); break;
  }
}

// PGID_LIGHT_ENV is env_constant or env_sphere depending on the miss shader (src/Device.cpp:704-716).
static inline void callLight(const SystemData& s, int lightType, float3 const& point, const float2 sample, LightSample& ls)
{
  if (lightType == LIGHT_PARALLELOGRAM) light_parallelogram(s, point, sample, ls);
  else if (s.miss == 2)                 light_env_sphere(s, point, sample, ls);
  else                                  light_env_constant(s, point, sample, ls);
}

static inline void callBsdfSample(int indexBSDF, MaterialDefinition const& m, State const& st, PerRayData* prd)
{
  switch (indexBSDF)
  {
    default:
    case INDEX_BRDF_DIFFUSE:   sample_brdf_diffuse(m, st, prd); break;
    case INDEX_BRDF_SPECULAR:  sample_brdf_specular(m, st, prd); break;
    case INDEX_BSDF_SPECULAR:  sample_bsdf_specular(m, st, prd); break;
    case INDEX_BRDF_GGX_SMITH: sample_brdf_ggx_smith(m, st, prd); break;
    case INDEX_BSDF_GGX_SMITH: sample_bsdf_ggx_smith(m, st, prd); break;
  }
}

static inline float4 callBsdfEval(int indexBSDF, MaterialDefinition const& m, State const& st, PerRayData* prd, float3 const& wiL)
{
  switch (indexBSDF)
  {
    default:
    case INDEX_BRDF_DIFFUSE:   return eval_brdf_diffuse(m, st, prd, wiL);
    case INDEX_BRDF_GGX_SMITH: return eval_brdf_ggx_smith(m, st, prd, wiL);
    case INDEX_BRDF_SPECULAR:
    case INDEX_BSDF_SPECULAR:
    case INDEX_BSDF_GGX_SMITH: return eval_brdf_specular(m, st, prd, wiL);
  }
}

} // namespace orc
