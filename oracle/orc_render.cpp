// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header). Never linked into the product.
//
// Ray generation, integrator, closest-hit, any-hit and miss programs of the reference restated as a
// plain per-sample CPU loop, plus the C entry points the tests and bench.py's cpu_baseline leg call.
// The C entry points take the same inputs as include/tweeker_hip.h so one scene description can be
// fed to both sides.
#include "orc_shaders.h"
#include "../include/tweeker_hip.h"

#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <thread>
#include <cmath>

// The reference's compile-time switches (apps/rtigo3/shaders/config.h:50-60), resolved at compile time here too: the Makefile
// builds liboracle.so with the defaults and liboracle_nee0.so / liboracle_dbgexc.so with one of them flipped
// (orc.Oracle(nee=False) / orc.Oracle(debugExceptions=True)); the product has them as run-time switches
// (twk_set_next_event_estimation, twk_set_debug_exceptions).
#ifndef USE_NEXT_EVENT_ESTIMATION
#define USE_NEXT_EVENT_ESTIMATION 1
#endif
#ifndef USE_DEBUG_EXCEPTIONS
#define USE_DEBUG_EXCEPTIONS 0
#endif

namespace orc {

struct Oracle
{
  SystemData sys;
  Scene      scene;
  std::vector<float4> output;     // RGBA32F running mean, W*H (distribution 0) or launchWidth*H (distribution 1)
  std::vector<Hit>    firstHits;  // per launch index, filled by the last render when captureFirstHits
  bool captureFirstHits = false;
  int  launchWidth = 1;
  int  shaderVariant = TWK_SHADERS_RTIGO3; // which app's __closesthit__radiance rule for light hits (include/tweeker_hip.h)
  bool aov = false;                        // Optix7Gui's denoiser AOVs (raygeneration.cu:125-164,239-262)
  std::vector<float4> aovAlbedo, aovNormal;
  uint64_t radianceRays = 0, shadowRays = 0, samples = 0;
};

struct RayTally { uint64_t radiance = 0, shadow = 0, samples = 0; };
static RayTally& rayTally() { static thread_local RayTally t; return t; }
static std::mutex g_tallyMutex;

// Adds this thread's tallies to the handle's totals (end of a render call / of a worker thread).
static void mergeTallies(Oracle& o)
{
  std::lock_guard<std::mutex> lock(g_tallyMutex);
  RayTally& r = rayTally();
  o.radianceRays += r.radiance; o.shadowRays += r.shadow; o.samples += r.samples;
  r = RayTally();
  TraceCounters& t = traceTally();
  o.scene.counters.rays += t.rays; o.scene.counters.boxTests += t.boxTests; o.scene.counters.triTests += t.triTests;
  t = TraceCounters();
}

// ---------------------------------------------------------------------------------------------
// Hit context ≙ what the OptiX intrinsics hand to the hit programs.
struct HitContext
{
  const Instance* inst;
  const Geometry* geom;
  int   primitive;
  float beta, gamma;
  float tmax;
};

// closesthit.cu:114-123
static inline float3 transformNormal(const float* m, float3 const& v)
{
  float3 r;
  r.x = m[0] * v.x + m[4] * v.y + m[8]  * v.z;
  r.y = m[1] * v.x + m[5] * v.y + m[9]  * v.z;
  r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z;
  return r;
}

// anyhit.cu:46-80 __anyhit__radiance_cutout, restated statement for statement: returns true where the program calls
// optixIgnoreIntersection() (the candidate is skipped), false where it falls off its end (the candidate is accepted).
// `seed` is thePrd->seed of the ray's payload.
static inline bool anyhitRadianceCutout(const Oracle& o, const HitContext& hc, unsigned int& seed)
{
  const SystemData& sysData = o.sys;
  const Geometry& g = *hc.geom;

  MaterialDefinition const& material = sysData.materialDefinitions[hc.inst->material];

  if (material.textureCutout != 0)
  {
    const unsigned int* tri = &g.indices[3 * (size_t) hc.primitive];

    const float2 theBarycentrics = make_float2(hc.beta, hc.gamma);

    const float  alpha = 1.0f - theBarycentrics.x - theBarycentrics.y;

    const float3 texcoord = g.attributes[tri[0]].texcoord * alpha +
                            g.attributes[tri[1]].texcoord * theBarycentrics.x +
                            g.attributes[tri[2]].texcoord * theBarycentrics.y;

    const float opacity = intensity(make_float3(tex2D(sysData.textures[1], texcoord.x, texcoord.y)));

    if (opacity < 1.0f && opacity <= rng(seed))
    {
      return true;
    }
  }
  return false;
}

// anyhit.cu:94-132 __anyhit__shadow_cutout: true = optixIgnoreIntersection(), false = FLAG_SHADOW + optixTerminateRay()
// (the caller sets the flag). `seed` is the stream the shadow ray draws from (see "Cutout opacity" below).
static inline bool anyhitShadowCutout(const Oracle& o, const HitContext& hc, unsigned int& seed)
{
  const SystemData& sysData = o.sys;
  const Geometry& g = *hc.geom;

  MaterialDefinition const& material = sysData.materialDefinitions[hc.inst->material];

  float opacity = 1.0f;

  if (material.textureCutout != 0)
  {
    const unsigned int* tri = &g.indices[3 * (size_t) hc.primitive];

    const float2 theBarycentrics = make_float2(hc.beta, hc.gamma);
    const float  alpha = 1.0f - theBarycentrics.x - theBarycentrics.y;

    const float3 texcoord = g.attributes[tri[0]].texcoord * alpha +
                            g.attributes[tri[1]].texcoord * theBarycentrics.x +
                            g.attributes[tri[2]].texcoord * theBarycentrics.y;

    opacity = intensity(make_float3(tex2D(sysData.textures[1], texcoord.x, texcoord.y)));
  }

  if (opacity < 1.0f && opacity <= rng(seed))
  {
    return true;
  }
  else
  {
    return false;
  }
}

// Cutout opacity — what is pinned and what is a choice of this build (SURVEY.md §7 "RNG-stream fidelity"):
// OptiX invokes the any-hit program once per candidate intersection in TRAVERSAL order, which is implementation
// defined inside libnvoptix, and every invocation draws from the path's LCG (anyhit.cu:75,122). Here candidates are
// visited closest-first (re-trace strictly behind an ignored candidate), each exactly once:
//   * radiance rays draw from the path's seed, before the closest-hit program runs — as in the reference;
//   * shadow rays draw from a stream FORKED from the path's seed when the shadow ray is emitted
//     (tea<2>(seed, 'SHDW')), so that the visibility test does not feed back into the path's later draws. The
//     reference couples them (the Russian-roulette draw follows the shadow ray's any-hit draws); decoupling keeps
//     the shadow ray of bounce k independent of the path's continuation, which the wavefront schedule relies on.
//     Statistically equivalent, not sample-identical to an OptiX run (which is not reproducible across OptiX
//     versions either).
static const unsigned int SHADOW_FORK = 0x53484457u; // 'SHDW'

// Debug tap (orc_debug_path): the rays one pixel's sample traces, in call order: o.xyz, tmin, d.xyz, tmax, kind (0 radiance, 1 shadow).
static thread_local std::vector<float>* g_pathLog = nullptr;
static inline void logRay(const float3& org, const float3& dir, float tmin, float tmax, float kind)
{
  if (!g_pathLog) return;
  const float r[9] = {org.x, org.y, org.z, tmin, dir.x, dir.y, dir.z, tmax, kind};
  g_pathLog->insert(g_pathLog->end(), r, r + 9);
}

static Hit traceRadiance(Oracle& o, PerRayData* prd, const float3& org, const float3& dir, float tmin, float tmax)
{
  rayTally().radiance++;
  logRay(org, dir, tmin, tmax, 0.0f);
  float lo = tmin;
  for (;;)
  {
    Hit h = o.scene.trace(org, dir, lo, tmax, false);
    if (h.instance < 0) return h;
    const Instance& inst = o.scene.instances[h.instance];
    HitContext hc = { &inst, &o.scene.geometries[inst.geometry], h.primitive, h.beta, h.gamma, h.t };
    if (!anyhitRadianceCutout(o, hc, prd->seed)) return h; // anyhit.cu:46-80
    lo = h.t; // continue strictly behind the ignored candidate
  }
}

static bool traceShadow(Oracle& o, PerRayData* prd, const float3& org, const float3& dir, float tmin, float tmax)
{
  rayTally().shadow++;
  logRay(org, dir, tmin, tmax, 1.0f);
  bool anyCutout = false;
  for (const MaterialDefinition& m : o.sys.materialDefinitions)
    if (m.textureCutout != 0) { anyCutout = true; break; }
  if (!anyCutout)
  {
    return o.scene.trace(org, dir, tmin, tmax, true).instance >= 0; // anyhit.cu:84-91
  }
  unsigned int shadowSeed = tea<2>(prd->seed, SHADOW_FORK);
  float lo = tmin;
  for (;;)
  {
    Hit h = o.scene.trace(org, dir, lo, tmax, false);
    if (h.instance < 0) return false;
    const Instance& inst = o.scene.instances[h.instance];
    HitContext hc = { &inst, &o.scene.geometries[inst.geometry], h.primitive, h.beta, h.gamma, h.t };
    if (!anyhitShadowCutout(o, hc, shadowSeed)) return true; // anyhit.cu:94-132
    lo = h.t;
  }
}

// ---------------------------------------------------------------------------------------------
// miss.cu:41-52 / :54-73 / :75-109
static void missProgram(const Oracle& o, PerRayData* thePrd)
{
  const SystemData& sysData = o.sys;
  switch (sysData.miss)
  {
    case 0:
      thePrd->radiance = make_float3(0.0f);
      thePrd->albedo   = make_float3(0.0f); // Optix7Gui miss.cu:47-50 (FLAG_LIGHT too)
      thePrd->flags |= FLAG_LIGHT | FLAG_TERMINATE;
      break;
    default:
    case 1:
    {
#if USE_NEXT_EVENT_ESTIMATION
      const float weightMIS = (thePrd->flags & FLAG_DIFFUSE) ? powerHeuristic(thePrd->pdf, 0.25f * M_1_PIf_) : 1.0f;
      thePrd->radiance = make_float3(weightMIS);
#else
      thePrd->radiance = make_float3(1.0f);
#endif
      thePrd->albedo   = make_float3(1.0f); // Optix7Gui miss.cu:67-71
      thePrd->flags |= FLAG_LIGHT | FLAG_TERMINATE;
      break;
    }
    case 2:
    {
      const float3 R = thePrd->wi;
      const float u     = (pm_atan2f(R.x, -R.z) + M_PIf_) * 0.5f * M_1_PIf_ + sysData.envRotation;
      const float theta = pm_acosf(-R.y);
      const float v     = theta * M_1_PIf_;
      const float3 emission = make_float3(tex2D(sysData.textures[2], u, v));
#if USE_NEXT_EVENT_ESTIMATION
      float weightMIS = 1.0f;
      if (thePrd->flags & FLAG_DIFFUSE)
      {
        const float pdfLight = intensity(emission) / sysData.envIntegral;
        weightMIS = powerHeuristic(thePrd->pdf, pdfLight);
      }
      thePrd->radiance = emission * weightMIS;
#else
      thePrd->radiance = emission;
#endif
      thePrd->albedo   = emission;          // Optix7Gui miss.cu:105-109
      thePrd->flags |= FLAG_LIGHT | FLAG_TERMINATE;
      break;
    }
  }
}

// closesthit.cu:126-305
static void closesthitRadiance(Oracle& o, const HitContext& hc, PerRayData* thePrd)
{
  const SystemData& sysData = o.sys;
  const Geometry& g = *hc.geom;
  const unsigned int* tri = &g.indices[3 * (size_t) hc.primitive];

  TriangleAttributes const& attr0 = g.attributes[tri[0]];
  TriangleAttributes const& attr1 = g.attributes[tri[1]];
  TriangleAttributes const& attr2 = g.attributes[tri[2]];

  const float2 theBarycentrics = make_float2(hc.beta, hc.gamma);
  const float  alpha = 1.0f - theBarycentrics.x - theBarycentrics.y;

  const float3 ng = cross(attr1.vertex - attr0.vertex, attr2.vertex - attr0.vertex);
  const float3 tg = attr0.tangent * alpha + attr1.tangent * theBarycentrics.x + attr2.tangent * theBarycentrics.y;
  const float3 ns = attr0.normal  * alpha + attr1.normal  * theBarycentrics.x + attr2.normal  * theBarycentrics.y;

  State state;
  state.texcoord = attr0.texcoord * alpha + attr1.texcoord * theBarycentrics.x + attr2.texcoord * theBarycentrics.y;

  const float* objectToWorld = hc.inst->objectToWorld;
  const float* worldToObject = hc.inst->worldToObject;

  state.normalGeo = normalize(transformNormal(worldToObject, ng));
  state.tangent   = normalize(xfmVector(objectToWorld, tg));
  state.normal    = normalize(transformNormal(worldToObject, ns));

  thePrd->distance = hc.tmax;
  thePrd->pos = thePrd->pos + thePrd->wi * thePrd->distance;

  thePrd->flags |= (0.0f <= dot(thePrd->wo, state.normalGeo)) ? FLAG_FRONTFACE : 0;

  if ((thePrd->flags & FLAG_FRONTFACE) == 0)
  {
    state.normalGeo = -state.normalGeo;
    state.tangent   = -state.tangent;
    state.normal    = -state.normal;
  }

  thePrd->normal = state.normal; // Optix7Gui closesthit.cu:183-185
  thePrd->radiance = make_float3(0.0f);

  if (0 <= hc.inst->light && o.shaderVariant == TWK_SHADERS_OPTIX7GUI)
  {
    // apps/Optix7Gui/shaders/closesthit.cu:189-226: a light ends the path whichever side is hit, black on the back face
    float3 emission = make_float3(0.0f);
    const float cosTheta = dot(thePrd->wo, state.normalGeo);
    if ((thePrd->flags & FLAG_FRONTFACE) && DENOMINATOR_EPSILON < cosTheta)
    {
      LightDefinition const& light = sysData.lightDefinitions[hc.inst->light];
      emission = light.emission;
#if USE_NEXT_EVENT_ESTIMATION
      const float lightPdf = (thePrd->distance * thePrd->distance) / (light.area * cosTheta);
      if ((thePrd->flags & FLAG_DIFFUSE) && DENOMINATOR_EPSILON < lightPdf)
      {
        emission *= powerHeuristic(thePrd->pdf, lightPdf);
      }
#endif
    }
    thePrd->radiance = emission;
    thePrd->albedo   = emission;
    thePrd->flags |= (FLAG_HIT | FLAG_LIGHT | FLAG_TERMINATE);
    return;
  }

  if (0 <= hc.inst->light && (thePrd->flags & FLAG_FRONTFACE)) // apps/rtigo3/shaders/closesthit.cu:192-222
  {
    const float cosTheta = dot(thePrd->wo, state.normalGeo);
    if (DENOMINATOR_EPSILON < cosTheta)
    {
      LightDefinition const& light = sysData.lightDefinitions[hc.inst->light];
      float3 emission = light.emission;
#if USE_NEXT_EVENT_ESTIMATION
      const float lightPdf = (thePrd->distance * thePrd->distance) / (light.area * cosTheta);
      if ((thePrd->flags & FLAG_DIFFUSE) && DENOMINATOR_EPSILON < lightPdf)
      {
        emission *= powerHeuristic(thePrd->pdf, lightPdf);
      }
#endif
      thePrd->radiance = emission;
      thePrd->albedo   = emission;
      thePrd->flags |= FLAG_LIGHT | FLAG_TERMINATE;
      return;
    }
  }

  thePrd->f_over_pdf = make_float3(0.0f);
  thePrd->pdf        = 0.0f;

  MaterialDefinition const& material = sysData.materialDefinitions[hc.inst->material];

  state.albedo = material.albedo;

  if (material.textureAlbedo != 0)
  {
    const float3 texColor = make_float3(tex2D(sysData.textures[0], state.texcoord.x, state.texcoord.y));
    state.albedo *= texColor;
  }
  thePrd->albedo = state.albedo; // Optix7Gui closesthit.cu:247-249

  thePrd->flags = (thePrd->flags & ~FLAG_DIFFUSE) | FLAG_HIT | material.flags;

  callBsdfSample(material.indexBSDF, material, state, thePrd);

#if USE_NEXT_EVENT_ESTIMATION
  const int numLights = sysData.numLights;
  if ((thePrd->flags & FLAG_DIFFUSE) && 0 < numLights)
  {
    const float2 sample = rng2(thePrd->seed);

    LightSample lightSample;
    lightSample.index = (1 < numLights) ? clampi(static_cast<int>(floorf(rng(thePrd->seed) * numLights)), 0, numLights - 1) : 0;

    callLight(sysData, sysData.lightDefinitions[lightSample.index].type, thePrd->pos, sample, lightSample);

    if (0.0f < lightSample.pdf)
    {
      const float4 bsdf_pdf = callBsdfEval(material.indexBSDF, material, state, thePrd, lightSample.direction);

      if (0.0f < bsdf_pdf.w && isNotNull(make_float3(bsdf_pdf)))
      {
        const bool shadowed = traceShadow(o, thePrd, thePrd->pos, lightSample.direction,
                                          sysData.sceneEpsilon, lightSample.distance - sysData.sceneEpsilon);
        if (shadowed) thePrd->flags |= FLAG_SHADOW;

        if ((thePrd->flags & FLAG_SHADOW) == 0)
        {
          if (thePrd->flags & FLAG_VOLUME)
          {
            lightSample.emission *= expf3(-lightSample.distance * thePrd->sigma_t);
          }
          const float weightMis = powerHeuristic(lightSample.pdf, bsdf_pdf.w);
          thePrd->radiance += make_float3(bsdf_pdf) * lightSample.emission * (weightMis * dot(lightSample.direction, state.normal) / lightSample.pdf);
        }
      }
    }
  }
#endif
}

// raygeneration.cu:42-149
static float3 integrator(Oracle& o, PerRayData& prd, Hit* firstHit, float3& albedo, float3& normal)
{
  const SystemData& sysData = o.sys;
  albedo = make_float3(0.0f);     // Optix7Gui raygeneration.cu:66-71
  normal = make_float3(0.0f);
  prd.normal = make_float3(0.0f);
  float4 absorptionStack[MATERIAL_STACK_SIZE];
  int stackIdx = MATERIAL_STACK_EMPTY;
  int depth = 0;
  float3 radiance   = make_float3(0.0f);
  float3 throughput = make_float3(1.0f);

  prd.absorption_ior = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
  prd.sigma_t        = make_float3(0.0f);
  prd.flags          = 0;

  while (depth < sysData.pathLengths.y)
  {
    prd.wo        = -prd.wi;
    prd.ior       = make_float2(1.0f);
    prd.distance  = RT_DEFAULT_MAX;
    prd.flags    &= FLAG_CLEAR_MASK;

    if (MATERIAL_STACK_FIRST <= stackIdx)
    {
      prd.flags  |= FLAG_VOLUME;
      prd.sigma_t = make_float3(absorptionStack[stackIdx]);
      prd.ior.x   = absorptionStack[stackIdx].w;
      if (MATERIAL_STACK_FIRST <= stackIdx - 1)
      {
        prd.ior.y = absorptionStack[stackIdx - 1].w;
      }
    }

    const Hit h = traceRadiance(o, &prd, prd.pos, prd.wi, sysData.sceneEpsilon, prd.distance);
    if (depth == 0 && firstHit) *firstHit = h;
    if (h.instance < 0)
    {
      missProgram(o, &prd);
    }
    else
    {
      const Instance& inst = o.scene.instances[h.instance];
      HitContext hc = { &inst, &o.scene.geometries[inst.geometry], h.primitive, h.beta, h.gamma, h.t };
      closesthitRadiance(o, hc, &prd);
    }

    if (prd.flags & FLAG_VOLUME)
    {
      throughput *= expf3(-prd.distance * prd.sigma_t);
    }

    radiance += throughput * prd.radiance;

    if (o.aov)
    {
      // Optix7Gui raygeneration.cu:125-164
      if (!(prd.flags & FLAG_ALBEDO) && (prd.flags & (FLAG_DIFFUSE | FLAG_LIGHT)))
      {
        const float3 a = throughput * prd.albedo;
        albedo = make_float3(clampf(a.x, 0.0f, 1.0f), clampf(a.y, 0.0f, 1.0f), clampf(a.z, 0.0f, 1.0f));
        prd.flags |= FLAG_ALBEDO;
      }
      if (depth == 0 && h.instance >= 0) // FLAG_HIT of Optix7Gui's closest hit: any hit, lights included
      {
        const CameraDefinition& cam = sysData.cameraDefinitions[0];
        normal = make_float3( dot(prd.normal, normalize(cam.U)),
                              dot(prd.normal, normalize(cam.V)),
                             -dot(prd.normal, normalize(cam.W)));
      }
    }

    if ((prd.flags & FLAG_TERMINATE) || prd.pdf <= 0.0f || isNull(prd.f_over_pdf))
    {
      break;
    }

    throughput *= prd.f_over_pdf;

    if (sysData.pathLengths.x <= depth)
    {
      const float probability = fmaxf3(throughput);
      if (probability < rng(prd.seed))
      {
        break;
      }
      throughput /= probability;
    }

    if ((prd.flags & (FLAG_THINWALLED | FLAG_TRANSMISSION)) == FLAG_TRANSMISSION)
    {
      if (prd.flags & FLAG_FRONTFACE)
      {
        stackIdx = std::min(stackIdx + 1, (int) MATERIAL_STACK_LAST);
        absorptionStack[stackIdx] = prd.absorption_ior;
      }
      else
      {
        stackIdx = std::max(stackIdx - 1, (int) MATERIAL_STACK_EMPTY);
      }
    }

    ++depth;
  }
  return radiance;
}

// raygeneration.cu:152-164
static inline unsigned int distribute(const SystemData& sysData, const unsigned int x, const unsigned int y)
{
  const unsigned int xBlock = x >> sysData.tileShift.x;
  const unsigned int yBlock = y >> sysData.tileShift.y;
  const unsigned int xTile = xBlock * sysData.deviceCount + ((sysData.deviceIndex + yBlock) % sysData.deviceCount);
  return xTile * sysData.tileSize.x + (x & (sysData.tileSize.x - 1));
}

// raygeneration.cu:167-256 (single buffer) and :259-344 (local copy): one launch index.
// Seed: the reference uses launchDim.x * y + launchColumn * deviceCount + deviceIndex (:191), which
// for one device is W*y + x. This build seeds by absolute pixel W*y + launchColumn for every device
// count so that a tiled render equals the single-device render sample for sample.
static void raygenPathTracer(Oracle& o, unsigned int lx, unsigned int ly)
{
  const SystemData& sysData = o.sys;
  unsigned int launchColumn = lx;
  const bool tiled = (sysData.distribution && 1 < sysData.deviceCount);
  if (tiled)
  {
    launchColumn = distribute(sysData, lx, ly);
    if ((unsigned int) sysData.resolution.x <= launchColumn) return;
  }

  PerRayData prd;
  memset(&prd, 0, sizeof(prd));

  const unsigned int seedIndex = (unsigned int) sysData.resolution.x * ly + launchColumn;
  prd.seed = tea<4>(seedIndex, (unsigned int) sysData.iterationIndex);

  const float2 screen = make_float2(float(sysData.resolution.x), float(sysData.resolution.y));
  const float2 pixel  = make_float2(float(launchColumn), float(ly));
  const float2 sample = rng2(prd.seed);

  callLens(sysData, sysData.lensShader, screen, pixel, sample, prd.pos, prd.wi);

  const unsigned int index = tiled ? (ly * (unsigned int) o.launchWidth + lx) : (ly * (unsigned int) sysData.resolution.x + launchColumn);

  float3 albedo, normal;
  float3 radiance = integrator(o, prd, o.captureFirstHits ? &o.firstHits[index] : nullptr, albedo, normal);
  rayTally().samples++;

#if USE_DEBUG_EXCEPTIONS
  if (std::isnan(radiance.x) || std::isnan(radiance.y) || std::isnan(radiance.z))
  {
    radiance = make_float3(1000000.0f, 0.0f, 0.0f);
  }
  else if (std::isinf(radiance.x) || std::isinf(radiance.y) || std::isinf(radiance.z))
  {
    radiance = make_float3(0.0f, 1000000.0f, 0.0f);
  }
  else if (radiance.x < 0.0f || radiance.y < 0.0f || radiance.z < 0.0f)
  {
    radiance = make_float3(0.0f, 0.0f, 1000000.0f);
  }
#else
  if (!(std::isnan(radiance.x) || std::isnan(radiance.y) || std::isnan(radiance.z)))
#endif
  {
    if (0 < sysData.iterationIndex)
    {
      const float t = 1.0f / float(sysData.iterationIndex + 1);
      const float4 dst = o.output[index];
      radiance = lerp(make_float3(dst), radiance, t);
      if (o.aov) // Optix7Gui raygeneration.cu:243-252
      {
        albedo = lerp(make_float3(o.aovAlbedo[index]), albedo, t);
        normal = lerp(make_float3(o.aovNormal[index]), normal, t);
        if (isNotNull(normal)) normal = normalize(normal);
      }
    }
    o.output[index] = make_float4(radiance, 1.0f);
    if (o.aov)
    {
      o.aovAlbedo[index] = make_float4(albedo, 1.0f);
      o.aovNormal[index] = make_float4(normal, 0.0f);
    }
  }
}

// src/Device.cpp:1172-1189
static int2 calculateTileShift(const int2 tileSize)
{
  int xShift = 0;
  while (xShift < 32 && (tileSize.x & (1 << xShift)) == 0) ++xShift;
  int yShift = 0;
  while (yShift < 32 && (tileSize.y & (1 << yShift)) == 0) ++yShift;
  return {xShift, yShift};
}

// src/Texture.cpp:1499-1537 — 3x3 Gaussian (sigma 0.5), repeat in x, clamp in y.
static float gaussianFilter(const float* rgba, unsigned int width, unsigned int height, unsigned int x, unsigned int y)
{
  unsigned int left   = (0 < x)          ? x - 1 : width - 1;
  unsigned int right  = (x < width - 1)  ? x + 1 : 0;
  unsigned int bottom = (0 < y)          ? y - 1 : y;
  unsigned int top    = (y < height - 1) ? y + 1 : y;

  const float *p = rgba + (width * y + x) * 4;
  float intensity = (p[0] + p[1] + p[2]) * 0.619347f;

  p = rgba + (width * bottom + x) * 4;
  float f = p[0] + p[1] + p[2];
  p = rgba + (width * y + left) * 4;
  f += p[0] + p[1] + p[2];
  p = rgba + (width * y + right) * 4;
  f += p[0] + p[1] + p[2];
  p = rgba + (width * top + x) * 4;
  f += p[0] + p[1] + p[2];
  intensity += f * 0.0838195f;

  p = rgba + (width * bottom + left) * 4;
  f  = p[0] + p[1] + p[2];
  p = rgba + (width * bottom + right) * 4;
  f += p[0] + p[1] + p[2];
  p = rgba + (width * top + left) * 4;
  f += p[0] + p[1] + p[2];
  p = rgba + (width * top + right) * 4;
  f += p[0] + p[1] + p[2];
  intensity += f * 0.0113437f;

  return intensity / 3.0f;
}

// src/Texture.cpp:1542-1645 — CDFs for the importance-sampled spherical environment + its integral.
static void calculateSphericalCDF(Oracle& o)
{
  const Texture& tex = o.sys.textures[2];
  const unsigned int m_width = (unsigned int) tex.width, m_height = (unsigned int) tex.height;
  const float* rgba = reinterpret_cast<const float*>(tex.texels.data());
  std::vector<float> funcU((size_t) m_width * m_height), funcV(m_height + 1);

  float sum = 0.0f;
  for (unsigned int y = 0; y < m_height; ++y)
  {
    float sinTheta = float(sin(M_PI * (double(y) + 0.5) / double(m_height)));
    for (unsigned int x = 0; x < m_width; ++x)
    {
      const float value = gaussianFilter(rgba, m_width, m_height, x, y);
      funcU[y * m_width + x] = value * sinTheta;
      const float *p = rgba + (y * m_width + x) * 4;
      const float intensity = (p[0] + p[1] + p[2]) / 3.0f;
      sum += intensity * sinTheta;
    }
  }
  o.sys.envIntegral = sum * 2.0f * M_PIf_ * M_PIf_ / float(m_width * m_height);

  o.sys.envCDF_U.assign((size_t) (m_width + 1) * m_height, 0.0f);
  o.sys.envCDF_V.assign(m_height + 1, 0.0f);
  float* cdfU = o.sys.envCDF_U.data();
  float* cdfV = o.sys.envCDF_V.data();

  for (unsigned int y = 0; y < m_height; ++y)
  {
    unsigned int row = y * (m_width + 1);
    cdfU[row + 0] = 0.0f;
    for (unsigned int x = 1; x <= m_width; ++x)
    {
      unsigned int i = row + x;
      cdfU[i] = cdfU[i - 1] + funcU[y * m_width + x - 1];
    }
    const float integral = cdfU[row + m_width];
    funcV[y] = integral;
    if (integral != 0.0f)
    {
      for (unsigned int x = 1; x <= m_width; ++x) cdfU[row + x] /= integral;
    }
    else
    {
      for (unsigned int x = 1; x <= m_width; ++x) cdfU[row + x] = float(x) / float(m_width);
    }
  }

  cdfV[0] = 0.0f;
  for (unsigned int y = 1; y <= m_height; ++y) cdfV[y] = cdfV[y - 1] + funcV[y - 1];
  const float integral = cdfV[m_height];
  if (integral != 0.0f)
  {
    for (unsigned int y = 1; y <= m_height; ++y) cdfV[y] /= integral;
  }
  else
  {
    for (unsigned int y = 1; y <= m_height; ++y) cdfV[y] = float(y) / float(m_height);
  }
  o.sys.envWidth = m_width; o.sys.envHeight = m_height;
}

} // namespace orc

// =============================================================================================
// C entry points (tests / bench cpu_baseline only)
using namespace orc;

static std::string g_error;

extern "C" {

typedef struct Oracle* OrcHandle;

const char* orc_last_error(void) { return g_error.c_str(); }

// miss ≙ Device ctor argument (src/Device.cpp:222-227); index/count ≙ deviceIndex/deviceCount.
int orc_create(OrcHandle* out, int index, int count, int miss)
{
  Oracle* o = new Oracle();
  o->sys.deviceIndex = index; o->sys.deviceCount = count; o->sys.miss = miss;
  *out = o;
  return 0;
}

int orc_destroy(OrcHandle o) { delete o; return 0; }

// src/Device.cpp:1192-1256 + DeviceMultiGPULocalCopy.cpp:84-97
int orc_set_state(OrcHandle o, const TwkDeviceState* s)
{
  o->sys.resolution   = {s->resolution[0], s->resolution[1]};
  o->sys.tileSize     = {s->tileSize[0], s->tileSize[1]};
  o->sys.tileShift    = calculateTileShift(o->sys.tileSize);
  o->sys.pathLengths  = {s->pathLengths[0], s->pathLengths[1]};
  o->sys.distribution = s->distribution;
  o->sys.samplesSqrt  = s->samplesSqrt;
  o->sys.lensShader   = s->lensShader;
  o->sys.sceneEpsilon = s->epsilonFactor * SCENE_EPSILON_SCALE;
  o->sys.envRotation  = s->envRotation;
  const bool tiled = (s->distribution && 1 < o->sys.deviceCount);
  if (tiled)
  {
    const int width = (s->resolution[0] + o->sys.deviceCount - 1) / o->sys.deviceCount;
    const int mask  = s->tileSize[0] - 1;
    o->launchWidth = (width + mask) & ~mask;
  }
  else o->launchWidth = s->resolution[0];
  o->output.assign((size_t) o->launchWidth * s->resolution[1], make_float4(0.0f));
  o->firstHits.assign(o->output.size(), Hit{0, 0, 0, -1, -1});
  if (o->aov) { o->aovAlbedo.assign(o->output.size(), make_float4(0.0f)); o->aovNormal = o->aovAlbedo; }
  return 0;
}

int orc_init_cameras(OrcHandle o, const TwkCameraDefinition* c, int count)
{
  o->sys.cameraDefinitions.resize(count);
  for (int i = 0; i < count; ++i)
  {
    CameraDefinition& d = o->sys.cameraDefinitions[i];
    d.P = make_float3(c[i].P[0], c[i].P[1], c[i].P[2]);
    d.U = make_float3(c[i].U[0], c[i].U[1], c[i].U[2]);
    d.V = make_float3(c[i].V[0], c[i].V[1], c[i].V[2]);
    d.W = make_float3(c[i].W[0], c[i].W[1], c[i].W[2]);
  }
  return 0;
}

int orc_init_lights(OrcHandle o, const TwkLightDefinition* l, int count)
{
  o->sys.lightDefinitions.resize(count);
  for (int i = 0; i < count; ++i)
  {
    LightDefinition& d = o->sys.lightDefinitions[i];
    d.type     = l[i].type;
    d.position = make_float3(l[i].position[0], l[i].position[1], l[i].position[2]);
    d.vecU     = make_float3(l[i].vecU[0], l[i].vecU[1], l[i].vecU[2]);
    d.vecV     = make_float3(l[i].vecV[0], l[i].vecV[1], l[i].vecV[2]);
    d.normal   = make_float3(l[i].normal[0], l[i].normal[1], l[i].normal[2]);
    d.area     = l[i].area;
    d.emission = make_float3(l[i].emission[0], l[i].emission[1], l[i].emission[2]);
  }
  o->sys.numLights = count;
  return 0;
}

// src/Device.cpp:1022-1050
int orc_init_materials(OrcHandle o, const TwkMaterialGUI* m, int count)
{
  o->sys.materialDefinitions.resize(count);
  for (int i = 0; i < count; ++i)
  {
    MaterialDefinition& material = o->sys.materialDefinitions[i];
    material.textureAlbedo = m[i].useAlbedoTexture ? 1 : 0;
    material.textureCutout = m[i].useCutoutTexture ? 2 : 0;
    material.roughness     = make_float2(m[i].roughness[0], m[i].roughness[1]);
    material.indexBSDF     = m[i].indexBSDF;
    material.albedo        = make_float3(m[i].albedo[0], m[i].albedo[1], m[i].albedo[2]);
    material.absorption    = make_float3(0.0f);
    if (0.0f < m[i].absorptionScale)
    {
      const float x = -logf(fmax(0.0001f, m[i].absorptionColor[0]));
      const float y = -logf(fmax(0.0001f, m[i].absorptionColor[1]));
      const float z = -logf(fmax(0.0001f, m[i].absorptionColor[2]));
      material.absorption = make_float3(x, y, z) * m[i].absorptionScale;
    }
    material.ior   = m[i].ior;
    material.flags = (m[i].thinwalled) ? FLAG_THINWALLED : 0;
  }
  return 0;
}

int orc_init_texture(OrcHandle o, int slot, const float* rgba, int width, int height)
{
  if (slot < 0 || slot > 2 || width <= 0 || height <= 0) { g_error = "orc_init_texture: bad arguments"; return 1; }
  Texture& t = o->sys.textures[slot];
  t.width = width; t.height = height; t.clampV = (slot == 2);
  t.texels.resize((size_t) width * height);
  memcpy(t.texels.data(), rgba, sizeof(float) * 4 * (size_t) width * height);
  if (slot == 2) calculateSphericalCDF(*o);
  return 0;
}

int orc_get_env_tables(OrcHandle o, float* cdfU, float* cdfV, float* integral)
{
  if (cdfU) memcpy(cdfU, o->sys.envCDF_U.data(), sizeof(float) * o->sys.envCDF_U.size());
  if (cdfV) memcpy(cdfV, o->sys.envCDF_V.data(), sizeof(float) * o->sys.envCDF_V.size());
  if (integral) *integral = o->sys.envIntegral;
  return 0;
}

int orc_add_geometry(OrcHandle o, const TwkTriangleAttributes* attributes, size_t numAttributes,
                     const unsigned int* indices, size_t numIndices, int* idGeometry)
{
  static_assert(sizeof(TwkTriangleAttributes) == sizeof(TriangleAttributes), "layout");
  const int id = o->scene.addGeometry(reinterpret_cast<const TriangleAttributes*>(attributes), numAttributes, indices, numIndices);
  if (idGeometry) *idGeometry = id;
  return 0;
}

int orc_add_instance(OrcHandle o, int idGeometry, const float transform[12], int idMaterial, int idLight, int* idInstance)
{
  if (idGeometry < 0 || idGeometry >= (int) o->scene.geometries.size()) { g_error = "orc_add_instance: bad geometry"; return 1; }
  const int id = o->scene.addInstance(idGeometry, transform, idMaterial, idLight);
  if (idInstance) *idInstance = id;
  return 0;
}

int orc_clear_scene(OrcHandle o) { o->scene.clear(); return 0; }

// 0 = brute force over all triangles (the definition), 1 = oracle BVH (same results, faster)
int orc_set_trace_mode(OrcHandle o, int useBvh) { o->scene.useBvh = (useBvh != 0); return 0; }
int orc_set_flatten_policy(OrcHandle o, int maxTriangles, int maxReferences) { o->scene.setFlattenPolicy(maxTriangles, maxReferences); return 0; } // ≙ twk_set_flatten_policy
int orc_set_shader_variant(OrcHandle o, int variant) { o->shaderVariant = variant; return 0; } // ≙ twk_set_shader_variant
int orc_enable_aov(OrcHandle o, int enable)                                                   // ≙ twk_enable_aov
{
  o->aov = (enable != 0);
  if (o->aov) { o->aovAlbedo.assign(o->output.size(), make_float4(0.0f)); o->aovNormal = o->aovAlbedo; }
  return 0;
}
int orc_read_aov(OrcHandle o, int which, float* rgba, size_t numFloats)                       // ≙ twk_read_aov
{
  const std::vector<float4>& src = (which == TWK_AOV_ALBEDO) ? o->aovAlbedo : o->aovNormal;
  if (!o->aov || numFloats != src.size() * 4) { g_error = "orc_read_aov: AOVs off or size mismatch"; return 1; }
  memcpy(rgba, src.data(), sizeof(float) * numFloats);
  return 0;
}
int orc_capture_first_hits(OrcHandle o, int enable) { o->captureFirstHits = (enable != 0); return 0; }

int orc_get_launch_width(OrcHandle o, int* w) { *w = o->launchWidth; return 0; }

// One sample per pixel of the launch rectangle [x0,x1) x [y0,y1) in launch coordinates
// (≙ one optixLaunch restricted to a window; pixels are independent so a window equals the same
// pixels of a full launch).
int orc_render_rect(OrcHandle o, unsigned int iterationIndex, int x0, int y0, int x1, int y1)
{
  if (o->sys.cameraDefinitions.empty() || o->sys.materialDefinitions.empty()) { g_error = "orc_render: cameras/materials missing"; return 4; }
  o->scene.prepare();
  o->sys.iterationIndex = (int) iterationIndex;
  x0 = std::max(x0, 0); y0 = std::max(y0, 0);
  x1 = std::min(x1, o->launchWidth); y1 = std::min(y1, o->sys.resolution.y);
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x)
      raygenPathTracer(*o, (unsigned int) x, (unsigned int) y);
  mergeTallies(*o);
  return 0;
}

// Same rows on `threads` host threads (pixels are independent; each thread takes whole rows from a shared counter).
// For the CPU baseline of bench.py; the image is the one the single-threaded call produces.
int orc_render_rect_threads(OrcHandle o, unsigned int iterationIndex, int x0, int y0, int x1, int y1, int threads)
{
  if (o->sys.cameraDefinitions.empty() || o->sys.materialDefinitions.empty()) { g_error = "orc_render: cameras/materials missing"; return 4; }
  if (o->captureFirstHits || threads <= 1) return orc_render_rect(o, iterationIndex, x0, y0, x1, y1);
  o->scene.prepare();
  o->sys.iterationIndex = (int) iterationIndex;
  x0 = std::max(x0, 0); y0 = std::max(y0, 0);
  x1 = std::min(x1, o->launchWidth); y1 = std::min(y1, o->sys.resolution.y);
  std::atomic<int> nextRow(y0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
  {
    pool.emplace_back([&]()
    {
      for (int y = nextRow.fetch_add(1); y < y1; y = nextRow.fetch_add(1))
        for (int x = x0; x < x1; ++x)
          raygenPathTracer(*o, (unsigned int) x, (unsigned int) y);
      mergeTallies(*o);
    });
  }
  for (std::thread& t : pool) t.join();
  return 0;
}

// Debug tap: the rays the sample (x, y, iterationIndex) traces — 9 floats each (o.xyz, tmin, d.xyz, tmax, kind) —
// without touching the accumulation buffer's other pixels. Returns the number of rays in *numRays.
int orc_debug_path(OrcHandle o, unsigned int iterationIndex, int x, int y, float* rays, int capacity, int* numRays)
{
  o->scene.prepare();
  o->sys.iterationIndex = (int) iterationIndex;
  std::vector<float> log;
  // the sample must not touch the image: the one element raygenPathTracer writes is put back (copying the whole frame, as
  // this did, cost 8 ms per call on a 1920x1080 frame)
  const bool tiled = (o->sys.distribution && 1 < o->sys.deviceCount);
  const size_t index = tiled ? ((size_t) y * (size_t) o->launchWidth + (size_t) x) : ((size_t) y * (size_t) o->sys.resolution.x + (size_t) x);
  if (index >= o->output.size()) { g_error = "orc_debug_path: pixel outside the frame"; return 1; }
  const float4 saved = o->output[index];
  g_pathLog = &log;
  raygenPathTracer(*o, (unsigned int) x, (unsigned int) y);
  g_pathLog = nullptr;
  o->output[index] = saved;
  mergeTallies(*o);
  const int n = (int) (log.size() / 9);
  *numRays = n;
  for (int i = 0; i < n && i < capacity; ++i) memcpy(rays + 9 * i, log.data() + 9 * i, sizeof(float) * 9);
  return 0;
}

int orc_render(OrcHandle o, unsigned int iterationIndex)
{
  return orc_render_rect(o, iterationIndex, 0, 0, o->launchWidth, o->sys.resolution.y);
}

int orc_read_output(OrcHandle o, float* rgba, size_t numFloats)
{
  if (numFloats != o->output.size() * 4) { g_error = "orc_read_output: size mismatch"; return 1; }
  memcpy(rgba, o->output.data(), sizeof(float) * numFloats);
  return 0;
}

int orc_read_first_hits(OrcHandle o, float* tBetaGamma, int* instPrim, size_t numPixels)
{
  if (numPixels != o->firstHits.size()) { g_error = "orc_read_first_hits: size mismatch"; return 1; }
  for (size_t i = 0; i < numPixels; ++i)
  {
    const Hit& h = o->firstHits[i];
    tBetaGamma[3 * i] = h.t; tBetaGamma[3 * i + 1] = h.beta; tBetaGamma[3 * i + 2] = h.gamma;
    instPrim[2 * i] = h.instance; instPrim[2 * i + 1] = h.primitive;
  }
  return 0;
}

int orc_get_counters(OrcHandle o, uint64_t out[6])
{
  out[0] = o->radianceRays; out[1] = o->shadowRays; out[2] = o->samples;
  out[3] = o->scene.counters.rays; out[4] = o->scene.counters.boxTests; out[5] = o->scene.counters.triTests;
  return 0;
}

// optixTrace contract on arbitrary rays (8 floats: o.xyz, tmin, d.xyz, tmax).
int orc_trace_rays(OrcHandle o, const float* rays, size_t numRays, int anyHit, float* tBetaGamma, int* ids)
{
  o->scene.prepare();
  for (size_t i = 0; i < numRays; ++i)
  {
    const float* r = rays + 8 * i;
    const Hit h = o->scene.trace(make_float3(r[0], r[1], r[2]), make_float3(r[4], r[5], r[6]), r[3], r[7], anyHit != 0);
    if (anyHit)
    {
      ids[2 * i] = (h.instance >= 0) ? 1 : 0; ids[2 * i + 1] = -1;
      tBetaGamma[3 * i] = tBetaGamma[3 * i + 1] = tBetaGamma[3 * i + 2] = 0.0f;
    }
    else
    {
      tBetaGamma[3 * i] = h.t; tBetaGamma[3 * i + 1] = h.beta; tBetaGamma[3 * i + 2] = h.gamma;
      ids[2 * i] = h.instance; ids[2 * i + 1] = h.primitive;
    }
  }
  mergeTallies(*o);
  return 0;
}

// ---- tonemapper ------------------------------------------------------------------------------
// Application::screenshot, tonemap branch (Application.cpp:2253-2297), the loop over idx restated statement for
// statement (the reference's x / y loops are flattened into one index; float3 overloads: orc_vec.h).
// tm = gamma, whitePoint, colorBalance[3], burnHighlights, crushBlacks, saturation, brightness (TonemapperGUI.h:34-43).
struct uchar3 { unsigned char x, y, z; };
static inline uchar3 make_uchar3(unsigned char x, unsigned char y, unsigned char z) { return {x, y, z}; }
struct TonemapperGUI { float gamma, whitePoint, colorBalance[3], burnHighlights, crushBlacks, saturation, brightness; };

static void screenshotTonemap(const TonemapperGUI& m_tonemapperGUI, const float4* bufferHost, size_t numPixels, uchar3* dst)
{
  const float  invGamma       = 1.0f / m_tonemapperGUI.gamma;
  const float3 colorBalance   = make_float3(m_tonemapperGUI.colorBalance[0], m_tonemapperGUI.colorBalance[1], m_tonemapperGUI.colorBalance[2]);
  const float  invWhitePoint  = m_tonemapperGUI.brightness / m_tonemapperGUI.whitePoint;
  const float  burnHighlights = m_tonemapperGUI.burnHighlights;
  const float  crushBlacks    = m_tonemapperGUI.crushBlacks + m_tonemapperGUI.crushBlacks + 1.0f;
  const float  saturation     = m_tonemapperGUI.saturation;

  for (size_t idx = 0; idx < numPixels; ++idx)
  {
    float3 hdrColor = make_float3(bufferHost[idx]);
    float3 ldrColor = invWhitePoint * colorBalance * hdrColor;
    ldrColor       *= ((ldrColor * burnHighlights) + 1.0f) / (ldrColor + 1.0f);

    float luminance = dot(ldrColor, make_float3(0.3f, 0.59f, 0.11f));
    ldrColor = lerp(make_float3(luminance), ldrColor, saturation);
    ldrColor = fmaxf3v(make_float3(0.0f), ldrColor);

    luminance = dot(ldrColor, make_float3(0.3f, 0.59f, 0.11f));
    if (luminance < 1.0f)
    {
      const float3 crushed = powf3(ldrColor, crushBlacks);
      ldrColor = lerp(crushed, ldrColor, sqrtf(luminance));
      ldrColor = fmaxf3v(make_float3(0.0f), ldrColor);
    }
    ldrColor = clamp3(powf3(ldrColor, invGamma), 0.0f, 1.0f);

    dst[idx] = make_uchar3((unsigned char) (ldrColor.x * 255.0f),
                           (unsigned char) (ldrColor.y * 255.0f),
                           (unsigned char) (ldrColor.z * 255.0f));
  }
}

int orc_tonemap(const float* tm, const float* rgba, size_t numPixels, unsigned char* rgb8)
{
  TonemapperGUI gui;
  gui.gamma = tm[0]; gui.whitePoint = tm[1];
  gui.colorBalance[0] = tm[2]; gui.colorBalance[1] = tm[3]; gui.colorBalance[2] = tm[4];
  gui.burnHighlights = tm[5]; gui.crushBlacks = tm[6]; gui.saturation = tm[7]; gui.brightness = tm[8];
  static_assert(sizeof(uchar3) == 3 && sizeof(float4) == 16, "packed pixels");
  screenshotTonemap(gui, reinterpret_cast<const float4*>(rgba), numPixels, reinterpret_cast<uchar3*>(rgb8));
  return 0;
}

// ---- compositor ------------------------------------------------------------------------------
// compositor.cu:38-64, one launch index (xLaunch, yLaunch) of the kernel; compositor_data.h:34-48 for the arguments.
struct CompositorData
{
  const float4* tileBuffer;
  float4*       outputBuffer;
  int2 resolution;
  int2 tileSize;
  int2 tileShift;
  int  launchWidth;
  int  deviceCount;
  int  deviceIndex;
};

static void compositor(const CompositorData* args, const unsigned int xLaunch, const unsigned int yLaunch)
{
  if (yLaunch < (unsigned int) args->resolution.y)
  {
    const unsigned int xBlock = xLaunch >> args->tileShift.x;
    const unsigned int yBlock = yLaunch >> args->tileShift.y;

    const unsigned int xTile = xBlock * args->deviceCount + ((args->deviceIndex + yBlock) % args->deviceCount);

    const unsigned int xPixel = xTile * args->tileSize.x + (xLaunch & (args->tileSize.x - 1));

    if (xPixel < (unsigned int) args->resolution.x)
    {
      const float4 *src = args->tileBuffer;
      float4       *dst = args->outputBuffer;

      dst[yLaunch * args->resolution.x + xPixel] = src[yLaunch * args->launchWidth + xLaunch];
    }
  }
}

// tiles: [deviceCount][height][launchWidth] float4, the gathered tile buffers; output: [height][width] float4.
// One compositor pass per source device, as DeviceMultiGPULocalCopy::compositor runs them (DeviceMultiGPULocalCopy.cpp:279-337).
int orc_compositor(const float* tiles, float* output, int width, int height, int launchWidth, int deviceCount, int tileX, int tileY)
{
  for (int device = 0; device < deviceCount; ++device)
  {
    CompositorData args;
    args.tileBuffer   = reinterpret_cast<const float4*>(tiles) + (size_t) device * height * launchWidth;
    args.outputBuffer = reinterpret_cast<float4*>(output);
    args.resolution   = {width, height};
    args.tileSize     = {tileX, tileY};
    args.tileShift    = calculateTileShift(args.tileSize);
    args.launchWidth  = launchWidth;
    args.deviceCount  = deviceCount;
    args.deviceIndex  = device;
    for (unsigned int y = 0; y < (unsigned int) height; ++y)
      for (unsigned int x = 0; x < (unsigned int) launchWidth; ++x)
        compositor(&args, x, y);
  }
  return 0;
}

// ---- unit taps for known-answer tests -------------------------------------------------------
// op 0 sin, 1 cos, 2 exp, 3 atan2(x, y), 4 acos, 5 atan, 6 sqrt, 7 1/x, 8 log, 9 pow(x, y)
int orc_math(int op, const float* x, const float* y, float* out, size_t n)
{
  for (size_t i = 0; i < n; ++i)
  {
    switch (op)
    {
      case 0: out[i] = pm_sinf(x[i]); break;
      case 1: out[i] = pm_cosf(x[i]); break;
      case 2: out[i] = pm_expf(x[i]); break;
      case 3: out[i] = pm_atan2f(x[i], y[i]); break;
      case 4: out[i] = pm_acosf(x[i]); break;
      case 5: out[i] = pm_atanf(x[i]); break;
      case 6: out[i] = sqrtf(x[i]); break;
      case 7: out[i] = 1.0f / x[i]; break;
      case 8: out[i] = pm_logf(x[i]); break;
      case 9: out[i] = pm_powf(x[i], y[i]); break;
      default: return 1;
    }
  }
  return 0;
}

unsigned int orc_tea4(unsigned int v0, unsigned int v1) { return tea<4>(v0, v1); }
float orc_rng(unsigned int* seed) { return rng(*seed); }

int orc_refract(const float i[3], const float n[3], float ior, float r[3])
{
  float3 rr;
  const bool ok = refract(rr, make_float3(i[0], i[1], i[2]), make_float3(n[0], n[1], n[2]), ior);
  r[0] = rr.x; r[1] = rr.y; r[2] = rr.z;
  return ok ? 1 : 0;
}

// BSDF unit taps for the CPU property tests (tests/test_oracle_properties.py): many samples / evaluations of one BSDF at
// a surface with shading normal = geometric normal `n` and tangent `t`, seen from `wo`, from the front (FLAG_FRONTFACE)
// in vacuum. params: albedo.xyz, roughness.xy, ior. sample out per draw: wi.xyz, f_over_pdf.xyz, pdf, flags.
static void tapSetup(const float* params, const float n[3], const float t[3], const float wo[3], MaterialDefinition& m, State& st, PerRayData& prd)
{
  memset(&m, 0, sizeof(m)); memset(&st, 0, sizeof(st)); memset(&prd, 0, sizeof(prd));
  m.albedo = make_float3(params[0], params[1], params[2]); m.roughness = make_float2(params[3], params[4]); m.ior = params[5];
  st.normalGeo = st.normal = make_float3(n[0], n[1], n[2]); st.tangent = make_float3(t[0], t[1], t[2]); st.albedo = m.albedo;
  prd.wo = make_float3(wo[0], wo[1], wo[2]);
  prd.ior = make_float2(1.0f);
  prd.absorption_ior = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
  prd.flags = FLAG_FRONTFACE;
}

int orc_prop_bsdf_sample(int indexBSDF, const float* params, const float n[3], const float t[3], const float wo[3], unsigned int seed, int count, float* out8)
{
  MaterialDefinition m; State st; PerRayData prd;
  for (int i = 0; i < count; ++i)
  {
    tapSetup(params, n, t, wo, m, st, prd);
    prd.seed = seed;
    callBsdfSample(indexBSDF, m, st, &prd);
    seed = prd.seed;
    float* o = out8 + 8 * (size_t) i;
    o[0] = prd.wi.x; o[1] = prd.wi.y; o[2] = prd.wi.z; o[3] = prd.f_over_pdf.x; o[4] = prd.f_over_pdf.y; o[5] = prd.f_over_pdf.z; o[6] = prd.pdf;
    memcpy(&o[7], &prd.flags, 4);
  }
  return 0;
}

int orc_prop_bsdf_eval(int indexBSDF, const float* params, const float n[3], const float t[3], const float wo[3], const float* wi3, int count, float* out4)
{
  MaterialDefinition m; State st; PerRayData prd;
  tapSetup(params, n, t, wo, m, st, prd);
  for (int i = 0; i < count; ++i)
  {
    const float4 r = callBsdfEval(indexBSDF, m, st, &prd, make_float3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]));
    out4[4 * i] = r.x; out4[4 * i + 1] = r.y; out4[4 * i + 2] = r.z; out4[4 * i + 3] = r.w;
  }
  return 0;
}

float orc_fresnel_dielectric(float et, float cosIn) { return evaluateFresnelDielectric(et, cosIn); }

int orc_tbn(const float tangentRef[3], const float n[3], float out9[9])
{
  TBN t(make_float3(tangentRef[0], tangentRef[1], tangentRef[2]), make_float3(n[0], n[1], n[2]));
  out9[0] = t.tangent.x; out9[1] = t.tangent.y; out9[2] = t.tangent.z;
  out9[3] = t.bitangent.x; out9[4] = t.bitangent.y; out9[5] = t.bitangent.z;
  out9[6] = t.normal.x; out9[7] = t.normal.y; out9[8] = t.normal.z;
  return 0;
}

int orc_vec3(int op, const float a[3], const float b[3], float out[3])
{
  const float3 A = make_float3(a[0], a[1], a[2]), B = make_float3(b[0], b[1], b[2]);
  float3 r = make_float3(0.0f);
  switch (op)
  {
    case 0: r = normalize(A); break;
    case 1: r = reflect(A, B); break;
    case 2: r = cross(A, B); break;
    case 3: r.x = dot(A, B); break;
    case 4: r.x = length(A); break;
    case 5: r.x = powerHeuristic(a[0], b[0]); r.y = intensity(A); break;
    default: return 1;
  }
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
  return 0;
}

// BSDF sample tap: in/out PerRayData fields packed as floats.
// io layout: wo[3], flags(bits), seed(bits), ior[2] | out: wi[3], f_over_pdf[3], pdf, flags, seed, absorption_ior[4]
int orc_bsdf_sample(int indexBSDF, const TwkMaterialGUI* mg, const float normalGeo[3], const float tangent[3], const float normal[3],
                    const float albedo[3], const float wo[3], unsigned int flagsIn, unsigned int seedIn, const float ior[2],
                    float out[16])
{
  Oracle tmp; orc_init_materials(&tmp, mg, 1);
  MaterialDefinition m = tmp.sys.materialDefinitions[0];
  State st;
  st.normalGeo = make_float3(normalGeo[0], normalGeo[1], normalGeo[2]);
  st.tangent   = make_float3(tangent[0], tangent[1], tangent[2]);
  st.normal    = make_float3(normal[0], normal[1], normal[2]);
  st.texcoord  = make_float3(0.0f);
  st.albedo    = make_float3(albedo[0], albedo[1], albedo[2]);
  PerRayData prd; memset(&prd, 0, sizeof(prd));
  prd.wo = make_float3(wo[0], wo[1], wo[2]);
  prd.flags = flagsIn; prd.seed = seedIn; prd.ior = make_float2(ior[0], ior[1]);
  callBsdfSample(indexBSDF, m, st, &prd);
  out[0] = prd.wi.x; out[1] = prd.wi.y; out[2] = prd.wi.z;
  out[3] = prd.f_over_pdf.x; out[4] = prd.f_over_pdf.y; out[5] = prd.f_over_pdf.z;
  out[6] = prd.pdf; out[7] = bits2f(prd.flags); out[8] = bits2f(prd.seed);
  out[9] = prd.absorption_ior.x; out[10] = prd.absorption_ior.y; out[11] = prd.absorption_ior.z; out[12] = prd.absorption_ior.w;
  return 0;
}

int orc_bsdf_eval(int indexBSDF, const TwkMaterialGUI* mg, const float normalGeo[3], const float tangent[3], const float normal[3],
                  const float albedo[3], const float wo[3], const float wiL[3], float out[4])
{
  Oracle tmp; orc_init_materials(&tmp, mg, 1);
  MaterialDefinition m = tmp.sys.materialDefinitions[0];
  State st;
  st.normalGeo = make_float3(normalGeo[0], normalGeo[1], normalGeo[2]);
  st.tangent   = make_float3(tangent[0], tangent[1], tangent[2]);
  st.normal    = make_float3(normal[0], normal[1], normal[2]);
  st.texcoord  = make_float3(0.0f);
  st.albedo    = make_float3(albedo[0], albedo[1], albedo[2]);
  PerRayData prd; memset(&prd, 0, sizeof(prd));
  prd.wo = make_float3(wo[0], wo[1], wo[2]);
  const float4 r = callBsdfEval(indexBSDF, m, st, &prd, make_float3(wiL[0], wiL[1], wiL[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
  return 0;
}

} // extern "C"
