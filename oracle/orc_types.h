// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// Data contract of the reference's device programs, restated (reference paths relative to apps/rtigo3/).
#pragma once
#include "orc_vec.h"
#include <vector>
#include <cstdint>

namespace orc {

// shaders/config.h:38-48
static const float RT_DEFAULT_MAX         = 1.e27f;
static const float SCENE_EPSILON_SCALE    = 1.0e-7f;
static const float DENOMINATOR_EPSILON    = 1.0e-6f;
static const float MICROFACET_MIN_ROUGHNESS = 0.0014142f;

static const float M_PIf_   = 3.14159265358979323846f; // vector_math.h M_PIf
static const float M_1_PIf_ = 0.318309886183790671538f; // vector_math.h M_1_PIf

// shaders/per_ray_data.h:39-71
enum
{
  MATERIAL_STACK_EMPTY = -1,
  MATERIAL_STACK_FIRST = 0,
  MATERIAL_STACK_LAST  = 3,
  MATERIAL_STACK_SIZE  = 4
};
static const unsigned int FLAG_HIT          = 0x00000001u;
static const unsigned int FLAG_SHADOW       = 0x00000002u;
static const unsigned int FLAG_DIFFUSE      = 0x00000004u;
static const unsigned int FLAG_LIGHT        = 0x00000008u; // Optix7Gui per_ray_data.h:49 (0x4 there, DIFFUSE 0x8: only ever tested together)
static const unsigned int FLAG_FRONTFACE    = 0x00000010u;
static const unsigned int FLAG_THINWALLED   = 0x00000020u;
static const unsigned int FLAG_TRANSMISSION = 0x00000100u;
static const unsigned int FLAG_VOLUME       = 0x00001000u;
static const unsigned int FLAG_ALBEDO       = 0x10000000u; // Optix7Gui per_ray_data.h:67
static const unsigned int FLAG_TERMINATE    = 0x80000000u;
static const unsigned int FLAG_CLEAR_MASK   = FLAG_DIFFUSE | FLAG_ALBEDO; // rtigo3 per_ray_data.h:71: DIFFUSE; Optix7Gui :76: DIFFUSE | ALBEDO (ALBEDO is never set without AOVs)

// shaders/function_indices.h:34-60
enum { NUM_LENS_SHADERS = 3, NUM_LIGHT_TYPES = 2 };
enum { INDEX_BRDF_DIFFUSE = 0, INDEX_BRDF_SPECULAR = 1, INDEX_BSDF_SPECULAR = 2, INDEX_BRDF_GGX_SMITH = 3, INDEX_BSDF_GGX_SMITH = 4 };
enum { LIGHT_ENVIRONMENT = 0, LIGHT_PARALLELOGRAM = 1 };

// shaders/per_ray_data.h:74-81
struct State
{
  float3 normalGeo;
  float3 tangent;
  float3 normal;
  float3 texcoord;
  float3 albedo;
};

// shaders/per_ray_data.h:84-114
struct PerRayData
{
  float4 absorption_ior;
  float2 ior;
  float3 pos;
  float  distance;
  float3 wo;
  float3 wi;
  float3 radiance;
  unsigned int flags;
  float3 f_over_pdf;
  float  pdf;
  float3 sigma_t;
  float  opacity;
  unsigned int seed;
  float3 albedo;  // Optix7Gui per_ray_data.h:111: albedo for the denoiser's albedo buffer
  float3 normal;  // Optix7Gui per_ray_data.h:114: shading normal for the denoiser's normal buffer
};

// shaders/material_definition.h:37-56 (texture objects become slot indices, 0 = none, else slot+1)
struct MaterialDefinition
{
  int    textureAlbedo;
  int    textureCutout;
  float2 roughness;
  int    indexBSDF;
  float3 albedo;
  float3 absorption;
  float  ior;
  unsigned int flags;
};

// shaders/light_definition.h:42-69
struct LightDefinition
{
  int    type;
  float3 position;
  float3 vecU;
  float3 vecV;
  float3 normal;
  float  area;
  float3 emission;
};

struct LightSample
{
  float3 position;
  int    index;
  float3 direction;
  float  distance;
  float3 emission;
  float  pdf;
};

// shaders/camera_definition.h:34-40
struct CameraDefinition { float3 P, U, V, W; };

// shaders/vertex_attributes.h:34-40
struct TriangleAttributes { float3 vertex, tangent, normal, texcoord; };

// RGBA32F texture sampled like the reference's CUDA texture objects: normalized coordinates, bilinear,
// wrap (clamp in v for the environment) — src/Texture.cpp:668-693,1353.
struct Texture
{
  int width = 0, height = 0;
  bool clampV = false;
  std::vector<float4> texels;
};

// shaders/system_data.h:40-90 (device pointers become host containers)
struct SystemData
{
  std::vector<CameraDefinition>   cameraDefinitions;
  std::vector<LightDefinition>    lightDefinitions;
  std::vector<MaterialDefinition> materialDefinitions;
  Texture textures[3]; // albedo, cutout, environment
  std::vector<float> envCDF_U, envCDF_V;
  int2  resolution   = {1, 1};
  int2  tileSize     = {8, 8};
  int2  tileShift    = {3, 3};
  int2  pathLengths  = {2, 5};
  int   deviceCount  = 1;
  int   deviceIndex  = 0;
  int   distribution = 0;
  int   iterationIndex = 0;
  int   samplesSqrt  = 0;
  float sceneEpsilon = 500.0f * SCENE_EPSILON_SCALE;
  int   lensShader   = 0;
  int   numLights    = 0;
  unsigned int envWidth = 0, envHeight = 0;
  float envIntegral = 1.0f;
  float envRotation = 0.0f;
  int   miss = 1;
};

} // namespace orc
