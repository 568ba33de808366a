// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver around the part of the reference that compiles in this image WITHOUT any
// stand-in: the host scene code (SceneGraph/Box/Plane/Sphere/Torus/Parallelogram/Camera/Parser.cpp,
// dp/math) and the header-only device helpers that do not include <optix.h>
// (shaders/random_number_generators.h, shader_common.h, vector_math.h). The reference sources are
// compiled where they lie under /root/reference (see oracle/Makefile, target _ref); nothing of them
// is copied into this repository. The shading programs (shaders/*.cu) include <optix.h>, which the
// image lacks, so they are unbuildable here and are NOT part of this library.
//
// Used by tests/test_oracle_vs_ref.py to pin the oracle's helper layer and the product's host scene
// layer, and by tests/golden/make_golden.py to write the committed fixtures.
#include <cuda_runtime.h>

#include "shaders/config.h"
#include "shaders/vector_math.h"
#include "shaders/shader_common.h"
#include "shaders/random_number_generators.h"
#include "shaders/vertex_attributes.h"
#include "shaders/camera_definition.h"
#include "shaders/light_definition.h"
#include "shaders/function_indices.h"
#include "inc/TonemapperGUI.h"
#include <cstddef>
#include "inc/SceneGraph.h"
#include "inc/Camera.h"
#include "inc/Parser.h"
#include "dp/math/Matmnt.h"
#include "dp/math/Quatt.h"

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

static int copyMesh(sg::Triangles& t, float* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
{
  static_assert(sizeof(TriangleAttributes) == 48, "TriangleAttributes layout");
  *numAttr = t.getAttributes().size();
  *numIdx  = t.getIndices().size();
  if (attr) memcpy(attr, t.getAttributes().data(), sizeof(TriangleAttributes) * *numAttr);
  if (idx)  memcpy(idx, t.getIndices().data(), sizeof(unsigned int) * *numIdx);
  return 0;
}

extern "C" {

int ref_mesh_plane(unsigned int tessU, unsigned int tessV, unsigned int upAxis, float* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
{
  sg::Triangles t(0); t.createPlane(tessU, tessV, upAxis); return copyMesh(t, attr, numAttr, idx, numIdx);
}
int ref_mesh_box(float* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
{
  sg::Triangles t(0); t.createBox(); return copyMesh(t, attr, numAttr, idx, numIdx);
}
int ref_mesh_sphere(unsigned int tessU, unsigned int tessV, float radius, float maxTheta, float* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
{
  sg::Triangles t(0); t.createSphere(tessU, tessV, radius, maxTheta); return copyMesh(t, attr, numAttr, idx, numIdx);
}
int ref_mesh_torus(unsigned int tessU, unsigned int tessV, float innerRadius, float outerRadius, float* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
{
  sg::Triangles t(0); t.createTorus(tessU, tessV, innerRadius, outerRadius); return copyMesh(t, attr, numAttr, idx, numIdx);
}
int ref_mesh_parallelogram(const float p[3], const float u[3], const float v[3], const float n[3], float* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
{
  sg::Triangles t(0);
  t.createParallelogram(make_float3(p[0], p[1], p[2]), make_float3(u[0], u[1], u[2]), make_float3(v[0], v[1], v[2]), make_float3(n[0], n[1], n[2]));
  return copyMesh(t, attr, numAttr, idx, numIdx);
}

// Camera::getFrustum through the public members the system description loader sets (Application.cpp:1205-1235).
int ref_camera_frustum(const float center[3], float phi, float theta, float fov, float distance, int width, int height, float out12[12])
{
  Camera c;
  c.m_center = make_float3(center[0], center[1], center[2]);
  c.m_phi = phi; c.m_theta = theta; c.m_fov = fov; c.m_distance = distance;
  c.setResolution(width, height);
  float3 p, u, v, w;
  c.getFrustum(p, u, v, w, true);
  out12[0] = p.x; out12[1] = p.y; out12[2]  = p.z;
  out12[3] = u.x; out12[4] = u.y; out12[5]  = u.z;
  out12[6] = v.x; out12[7] = v.y; out12[8]  = v.z;
  out12[9] = w.x; out12[10] = w.y; out12[11] = w.z;
  return 0;
}

unsigned int ref_tea4(unsigned int v0, unsigned int v1) { return tea<4>(v0, v1); }
float ref_rng(unsigned int* seed) { return rng(*seed); }
void ref_rng2(unsigned int* seed, float out[2]) { const float2 s = rng2(*seed); out[0] = s.x; out[1] = s.y; }

int ref_refract(const float i[3], const float n[3], float ior, float r[3])
{
  float3 rr;
  const bool ok = refract(rr, make_float3(i[0], i[1], i[2]), make_float3(n[0], n[1], n[2]), ior);
  r[0] = rr.x; r[1] = rr.y; r[2] = rr.z;
  return ok ? 1 : 0;
}

int ref_tbn(const float tangentRef[3], const float n[3], float out9[9])
{
  TBN t(make_float3(tangentRef[0], tangentRef[1], tangentRef[2]), make_float3(n[0], n[1], n[2]));
  out9[0] = t.tangent.x; out9[1] = t.tangent.y; out9[2] = t.tangent.z;
  out9[3] = t.bitangent.x; out9[4] = t.bitangent.y; out9[5] = t.bitangent.z;
  out9[6] = t.normal.x; out9[7] = t.normal.y; out9[8] = t.normal.z;
  return 0;
}

// same op codes as orc_vec3
int ref_vec3(int op, const float a[3], const float b[3], float out[3])
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

// Tokenise a description file with the reference Parser. Output: one line per token "<type> <text>\n".
int ref_parse_tokens(const char* filename, char* out, size_t capacity)
{
  Parser parser;
  if (!parser.load(std::string(filename))) return -1;
  std::string result, token;
  ParserTokenType t;
  while ((t = parser.getNextToken(token)) != PTT_EOF)
  {
    result += std::to_string((int) t) + " " + token + "\n";
    if (t == PTT_UNKNOWN) break;
  }
  if (result.size() + 1 > capacity) return -2;
  memcpy(out, result.c_str(), result.size() + 1);
  return (int) result.size();
}

// The loader's transform stack maths (Application.cpp:1612-1690) on dp::math: ops is a list of
// (kind, a, b, c, d): kind 0 rotate(axis a,b,c; degrees d), 1 scale(a,b,c), 2 translate(a,b,c).
// Output: the 3x4 row-major object→world matrix appendInstance would emit (Application.cpp:1353-1359).
int ref_transform_stack(const float* ops, int numOps, float trafo[12])
{
  dp::math::Mat44f curMatrix(dp::math::cIdentity44f);
  for (int i = 0; i < numOps; ++i)
  {
    const float* o = ops + 5 * i;
    const int kind = (int) o[0];
    if (kind == 0)
    {
      dp::math::Vec3f axis;
      axis[0] = o[1]; axis[1] = o[2]; axis[2] = o[3];
      axis.normalize();
      const float angle = dp::math::degToRad(o[4]);
      dp::math::Quatf rotation(axis, angle);
      dp::math::Mat44f matrix(rotation, dp::math::Vec3f(0.0f, 0.0f, 0.0f));
      curMatrix *= matrix;
    }
    else if (kind == 1)
    {
      dp::math::Mat44f scaling(dp::math::cIdentity44f);
      scaling[0][0] = o[1]; scaling[1][1] = o[2]; scaling[2][2] = o[3];
      curMatrix *= scaling;
    }
    else
    {
      dp::math::Mat44f translation(dp::math::cIdentity44f);
      translation[3][0] = o[1]; translation[3][1] = o[2]; translation[3][2] = o[3];
      curMatrix *= translation;
    }
  }
  const float t[12] =
  {
    curMatrix[0][0], curMatrix[1][0], curMatrix[2][0], curMatrix[3][0],
    curMatrix[0][1], curMatrix[1][1], curMatrix[2][1], curMatrix[3][1],
    curMatrix[0][2], curMatrix[1][2], curMatrix[2][2], curMatrix[3][2]
  };
  memcpy(trafo, t, sizeof(t));
  return 0;
}


// Layout of the reference's plain structs that cross the replaced boundary, as its own headers define them:
// out[] = size, then the offset of every member in declaration order. which: 0 CameraDefinition
// (shaders/camera_definition.h), 1 LightDefinition (shaders/light_definition.h), 2 TriangleAttributes
// (shaders/vertex_attributes.h), 3 TonemapperGUI (inc/TonemapperGUI.h). Returns the number of values written.
int ref_struct_layout(int which, int* out, int capacity)
{
  int n = 0;
#define PUT(v) do { if (n < capacity) out[n] = (int) (v); ++n; } while (0)
  switch (which)
  {
    case 0:
      PUT(sizeof(CameraDefinition));
      PUT(offsetof(CameraDefinition, P)); PUT(offsetof(CameraDefinition, U)); PUT(offsetof(CameraDefinition, V)); PUT(offsetof(CameraDefinition, W));
      break;
    case 1:
      PUT(sizeof(LightDefinition));
      PUT(offsetof(LightDefinition, type)); PUT(offsetof(LightDefinition, position)); PUT(offsetof(LightDefinition, vecU));
      PUT(offsetof(LightDefinition, vecV)); PUT(offsetof(LightDefinition, normal)); PUT(offsetof(LightDefinition, area));
      PUT(offsetof(LightDefinition, emission)); PUT(offsetof(LightDefinition, unused0)); PUT(offsetof(LightDefinition, unused1)); PUT(offsetof(LightDefinition, unused2));
      break;
    case 2:
      PUT(sizeof(TriangleAttributes));
      PUT(offsetof(TriangleAttributes, vertex)); PUT(offsetof(TriangleAttributes, tangent)); PUT(offsetof(TriangleAttributes, normal)); PUT(offsetof(TriangleAttributes, texcoord));
      break;
    case 3:
      PUT(sizeof(TonemapperGUI));
      PUT(offsetof(TonemapperGUI, gamma)); PUT(offsetof(TonemapperGUI, whitePoint)); PUT(offsetof(TonemapperGUI, colorBalance));
      PUT(offsetof(TonemapperGUI, burnHighlights)); PUT(offsetof(TonemapperGUI, crushBlacks)); PUT(offsetof(TonemapperGUI, saturation)); PUT(offsetof(TonemapperGUI, brightness));
      break;
    default: return -1;
  }
#undef PUT
  return n;
}

// FunctionIndex / LightType values the C ABI mirrors as plain ints (shaders/function_indices.h, light_definition.h).
int ref_enum_values(int* out, int capacity)
{
  const int v[] = {INDEX_BRDF_DIFFUSE, INDEX_BRDF_SPECULAR, INDEX_BSDF_SPECULAR, INDEX_BRDF_GGX_SMITH, INDEX_BSDF_GGX_SMITH,
                   LIGHT_ENVIRONMENT, LIGHT_PARALLELOGRAM};
  const int n = (int) (sizeof(v) / sizeof(v[0]));
  for (int i = 0; i < n && i < capacity; ++i) out[i] = v[i];
  return n;
}

} // extern "C"
