#include "application.h"
#include "description_parser.h"
#include "transform_stack.h"

#include <algorithm>
#include <cstdlib>
#include <sstream>

namespace twk {

Application::Application() {}

static bool readFloat(DescriptionParser& p, float& v)
{
  std::string t;
  if (p.nextToken(t) != TOKEN_VAL) return false;
  v = (float) atof(t.c_str());
  return true;
}

static bool readInt(DescriptionParser& p, int& v)
{
  std::string t;
  if (p.nextToken(t) != TOKEN_VAL) return false;
  v = atoi(t.c_str());
  return true;
}

static std::string where(const char* what, DescriptionParser& p, const std::string& key)
{
  std::ostringstream s;
  s << what << " line " << p.line() << ": '" << key << "' expects numeric arguments";
  return s.str();
}

bool Application::loadSystemDescription(const std::string& text, std::string& error)
{
  DescriptionParser parser;
  parser.loadString(text);

  std::string token;
  TokenType type;
  while ((type = parser.nextToken(token)) != TOKEN_EOF)
  {
    if (type == TOKEN_UNKNOWN) { error = "system description: unknown token type"; return false; }
    if (type != TOKEN_ID) continue; // stray values are skipped like the reference does

    const std::string key = token;
    bool ok = true;
    float f[4];
    int   i[2];

    if (key == "strategy")
    {
      ok = readInt(parser, i[0]);
      if (ok) { if (0 <= i[0] && i[0] < 4) strategy = i[0]; else warnings.push_back("invalid renderer strategy, using 0"); }
    }
    else if (key == "devicesMask") { ok = readInt(parser, devicesMask); }
    else if (key == "interop")     { ok = readInt(parser, interop); } // kept for saveSystemDescription; no OpenGL interop in this build
    else if (key == "present")     { ok = readInt(parser, i[0]); if (ok) present = (i[0] != 0); }
    else if (key == "resolution")
    {
      ok = readInt(parser, i[0]) && readInt(parser, i[1]);
      if (ok) { resolution[0] = std::max(1, i[0]); resolution[1] = std::max(1, i[1]); }
    }
    else if (key == "tileSize")
    {
      ok = readInt(parser, i[0]) && readInt(parser, i[1]);
      if (ok)
      {
        tileSize[0] = std::max(1, i[0]); tileSize[1] = std::max(1, i[1]);
        if (tileSize[0] & (tileSize[0] - 1)) { warnings.push_back("tileSize.x is not a power of two, using 8"); tileSize[0] = 8; }
        if (tileSize[1] & (tileSize[1] - 1)) { warnings.push_back("tileSize.y is not a power of two, using 8"); tileSize[1] = 8; }
      }
    }
    else if (key == "samplesSqrt") { ok = readInt(parser, i[0]); if (ok) samplesSqrt = std::max(1, i[0]); }
    else if (key == "miss")        { ok = readInt(parser, miss); }
    else if (key == "envMap")      { ok = (parser.restOfLine(environment) == TOKEN_ID); }
    else if (key == "envRotation") { ok = readFloat(parser, envRotation); }
    else if (key == "clockFactor") { ok = readFloat(parser, clockFactor); }
    else if (key == "light")       { ok = readInt(parser, i[0]); if (ok) light = std::min(2, std::max(0, i[0])); }
    else if (key == "pathLengths") { ok = readInt(parser, pathLengths[0]) && readInt(parser, pathLengths[1]); }
    else if (key == "epsilonFactor") { ok = readFloat(parser, epsilonFactor); }
    else if (key == "lensShader")
    {
      ok = readInt(parser, i[0]);
      if (ok) lensShader = (i[0] < 0 || 2 < i[0]) ? 0 : i[0];
    }
    else if (key == "center")
    {
      ok = readFloat(parser, f[0]) && readFloat(parser, f[1]) && readFloat(parser, f[2]);
      if (ok) { camera.center[0] = f[0]; camera.center[1] = f[1]; camera.center[2] = f[2]; }
    }
    else if (key == "camera")
    {
      ok = readFloat(parser, f[0]) && readFloat(parser, f[1]) && readFloat(parser, f[2]) && readFloat(parser, f[3]);
      if (ok) { camera.phi = f[0]; camera.theta = f[1]; camera.fov = f[2]; camera.distance = f[3]; }
    }
    else if (key == "prefixScreenshot") { ok = (parser.restOfLine(prefixScreenshot) == TOKEN_ID); }
    // Extension of the grammar (not in the reference): which app's closest-hit rule for light hits the scene was authored
    // for — 0 rtigo3, 1 Optix7Gui (intro_07's app ends a path on a light's back face, closesthit.cu:189-226).
    else if (key == "shaderVariant") { ok = readInt(parser, i[0]); if (ok) shaderVariant = (i[0] == 1) ? 1 : 0; }
    // Extensions of the grammar for the reference's two compile-time switches of shaders/config.h:50-56, which a system
    // description cannot reach there (they need a rebuild): USE_NEXT_EVENT_ESTIMATION and USE_DEBUG_EXCEPTIONS.
    else if (key == "nextEventEstimation") { ok = readInt(parser, i[0]); if (ok) nextEventEstimation = (i[0] != 0) ? 1 : 0; }
    else if (key == "debugExceptions")     { ok = readInt(parser, i[0]); if (ok) debugExceptions = (i[0] != 0) ? 1 : 0; }
    // tonemapper settings (Application.cpp:1244-1292), consumed by twk_tonemap / screenshot
    else if (key == "gamma")          { ok = readFloat(parser, tonemapper.gamma); }
    else if (key == "whitePoint")     { ok = readFloat(parser, tonemapper.whitePoint); }
    else if (key == "burnHighlights") { ok = readFloat(parser, tonemapper.burnHighlights); }
    else if (key == "crushBlacks")    { ok = readFloat(parser, tonemapper.crushBlacks); }
    else if (key == "saturation")     { ok = readFloat(parser, tonemapper.saturation); }
    else if (key == "brightness")     { ok = readFloat(parser, tonemapper.brightness); }
    else if (key == "colorBalance")
    {
      ok = readFloat(parser, tonemapper.colorBalance[0]) && readFloat(parser, tonemapper.colorBalance[1]) && readFloat(parser, tonemapper.colorBalance[2]);
    }
    else
    {
      warnings.push_back("unknown system option name: " + key);
    }

    if (!ok) { error = where("system description", parser, key); return false; }
  }

  camera.setResolution(resolution[0], resolution[1]); // Application.cpp:207
  return true;
}

// ≙ Application::saveSystemDescription (Application.cpp:1300-1345): the current settings in the grammar the loader
// reads, same keys in the same order, numbers through operator<< like the reference.
std::string Application::systemDescription() const
{
  std::ostringstream d;
  d << "strategy " << strategy << std::endl;
  d << "devicesMask " << devicesMask << std::endl;
  d << "interop " << interop << std::endl;
  d << "present " << (present ? "1" : "0") << std::endl;
  d << "resolution " << resolution[0] << " " << resolution[1] << std::endl;
  d << "tileSize " << tileSize[0] << " " << tileSize[1] << std::endl;
  d << "samplesSqrt " << samplesSqrt << std::endl;
  d << "miss " << miss << std::endl;
  if (!environment.empty()) d << "envMap " << environment << std::endl;
  d << "envRotation " << envRotation << std::endl;
  d << "clockFactor " << clockFactor << std::endl;
  d << "light " << light << std::endl;
  d << "pathLengths " << pathLengths[0] << " " << pathLengths[1] << std::endl;
  d << "epsilonFactor " << epsilonFactor << std::endl;
  d << "lensShader " << lensShader << std::endl;
  if (shaderVariant != 0) d << "shaderVariant " << shaderVariant << std::endl;
  if (nextEventEstimation != 1) d << "nextEventEstimation " << nextEventEstimation << std::endl;
  if (debugExceptions != 0) d << "debugExceptions " << debugExceptions << std::endl;
  d << "center " << camera.center[0] << " " << camera.center[1] << " " << camera.center[2] << std::endl;
  d << "camera " << camera.phi << " " << camera.theta << " " << camera.fov << " " << camera.distance << std::endl;
  if (!prefixScreenshot.empty()) d << "prefixScreenshot " << prefixScreenshot << std::endl;
  d << "gamma " << tonemapper.gamma << std::endl;
  d << "colorBalance " << tonemapper.colorBalance[0] << " " << tonemapper.colorBalance[1] << " " << tonemapper.colorBalance[2] << std::endl;
  d << "whitePoint " << tonemapper.whitePoint << std::endl;
  d << "burnHighlights " << tonemapper.burnHighlights << std::endl;
  d << "crushBlacks " << tonemapper.crushBlacks << std::endl;
  d << "saturation " << tonemapper.saturation << std::endl;
  d << "brightness " << tonemapper.brightness << std::endl;
  return d.str();
}

void Application::setResolution(int w, int h)
{
  resolution[0] = std::max(1, w);
  resolution[1] = std::max(1, h);
  camera.setResolution(resolution[0], resolution[1]);
  if (!cameras.empty()) cameras[0] = camera.frustum();
}

TwkDeviceState Application::deviceState() const
{
  TwkDeviceState s;
  s.resolution[0] = resolution[0];   s.resolution[1] = resolution[1];
  s.tileSize[0] = tileSize[0];       s.tileSize[1] = tileSize[1];
  s.pathLengths[0] = pathLengths[0]; s.pathLengths[1] = pathLengths[1];
  s.distribution  = (strategy == 0) ? 0 : 1; // full frames on one device, tiled across devices otherwise (Application.cpp:223-245)
  s.samplesSqrt   = samplesSqrt;
  s.lensShader    = lensShader;
  s.epsilonFactor = epsilonFactor;
  s.envRotation   = envRotation;
  s.clockFactor   = clockFactor;
  return s;
}

void Application::createCameras()
{
  cameras.clear();
  cameras.push_back(camera.frustum());
}

void Application::createLights()
{
  lights.clear();

  TwkLightDefinition l;
  l.type = TWK_LIGHT_ENVIRONMENT;
  l.position[0] = 0.0f; l.position[1] = 0.0f; l.position[2] = 0.0f;
  l.vecU[0] = 1.0f; l.vecU[1] = 0.0f; l.vecU[2] = 0.0f;
  l.vecV[0] = 0.0f; l.vecV[1] = 1.0f; l.vecV[2] = 0.0f;
  l.normal[0] = 0.0f; l.normal[1] = 0.0f; l.normal[2] = 1.0f;
  l.area = 1.0f;
  l.emission[0] = l.emission[1] = l.emission[2] = 1.0f;
  l.unused0 = l.unused1 = l.unused2 = 0.0f;

  if (miss == 1 || miss == 2) // the environment is always light 0
  {
    l.type = TWK_LIGHT_ENVIRONMENT;
    l.area = 4.0f * 3.14159265358979323846f;
    lights.push_back(l);
  }

  if (light != 1 && light != 2) return;

  const int indexLight = (int) lights.size();

  // light 1: 1x1 at y = 1.95 over a 2x2x2 box; light 2: 4x4 at y = 4 (Application.cpp:611-635)
  const float size = (light == 1) ? 1.0f : 4.0f;
  l.type = TWK_LIGHT_PARALLELOGRAM;
  l.position[0] = (light == 1) ? -0.5f : -2.0f;
  l.position[1] = (light == 1) ? 1.95f : 4.0f;
  l.position[2] = (light == 1) ? -0.5f : -2.0f;
  l.vecU[0] = size; l.vecU[1] = 0.0f; l.vecU[2] = 0.0f;
  l.vecV[0] = 0.0f; l.vecV[1] = 0.0f; l.vecV[2] = size;
  // normal = cross(vecU, vecV); area = |normal|; normal /= area (vector_math.h:580-589,526-530)
  const float nx = l.vecU[1] * l.vecV[2] - l.vecU[2] * l.vecV[1];
  const float ny = l.vecU[2] * l.vecV[0] - l.vecU[0] * l.vecV[2];
  const float nz = l.vecU[0] * l.vecV[1] - l.vecU[1] * l.vecV[0];
  l.area = sqrtf(nx * nx + ny * ny + nz * nz);
  const float inv = 1.0f / l.area;
  l.normal[0] = nx * inv; l.normal[1] = ny * inv; l.normal[2] = nz * inv;
  l.emission[0] = l.emission[1] = l.emission[2] = 10.0f;
  lights.push_back(l);

  // Black, thin-walled specular material for the light geometry (Application.cpp:640-659)
  TwkMaterialGUI m;
  m.indexBSDF = TWK_INDEX_BRDF_SPECULAR;
  m.albedo[0] = m.albedo[1] = m.albedo[2] = 0.0f;
  m.absorptionColor[0] = m.absorptionColor[1] = m.absorptionColor[2] = 1.0f;
  m.absorptionScale = 0.0f;
  m.ior = 1.5f;
  m.thinwalled = 1;
  m.useAlbedoTexture = 0;
  m.useCutoutTexture = 0;
  m.roughness[0] = m.roughness[1] = 0.1f;
  const int indexMaterial = (int) materials.size();
  materials.push_back(m);
  materialNames.push_back("rtigo3_area_light");
  m_materialReferences["rtigo3_area_light"] = indexMaterial;

  const unsigned int idGeometry = (unsigned int) geometries.size();
  m_geometryKeys["rtigo3_area_light"] = idGeometry;
  std::shared_ptr<TrianglesNode> geometry = std::make_shared<TrianglesNode>(idGeometry);
  makeParallelogram(geometry->mesh, l.position, l.vecU, l.vecV, l.normal);
  geometries.push_back(geometry);

  std::shared_ptr<InstanceNode> instance = std::make_shared<InstanceNode>(); // identity transform
  instance->child    = geometry;
  instance->material = indexMaterial;
  instance->light    = indexLight;
  m_scene->children.push_back(instance);
}

void Application::appendInstance(std::shared_ptr<TrianglesNode> geometry, const float trafo[12], const std::string& reference)
{
  std::shared_ptr<InstanceNode> instance = std::make_shared<InstanceNode>();
  for (int i = 0; i < 12; ++i) instance->transform[i] = trafo[i];
  instance->child = geometry;

  int indexMaterial = -1;
  std::map<std::string, int>::const_iterator it = m_materialReferences.find(reference);
  if (it != m_materialReferences.end())
  {
    indexMaterial = it->second;
  }
  else
  {
    warnings.push_back("no material found for " + reference + ", trying default");
    it = m_materialReferences.find("default");
    if (it != m_materialReferences.end()) indexMaterial = it->second;
    else warnings.push_back("no default material found");
  }
  instance->material = indexMaterial;
  m_scene->children.push_back(instance);
}

bool Application::loadSceneDescription(const std::string& text, std::string& error)
{
  DescriptionParser parser;
  parser.loadString(text);

  TransformStack xform;

  // Current material state (Application.cpp:1425-1431)
  float albedo[3]          = {1.0f, 1.0f, 1.0f};
  float roughness[2]       = {0.1f, 0.1f};
  float absorptionColor[3] = {1.0f, 1.0f, 1.0f};
  float absorptionScale    = 0.0f;
  float ior                = 1.5f;
  bool  thinwalled         = false;

  std::string token;
  TokenType type;
  while ((type = parser.nextToken(token)) != TOKEN_EOF)
  {
    if (type == TOKEN_UNKNOWN) { error = "scene description: unknown token type"; return false; }
    if (type != TOKEN_ID) continue;

    const std::string key = token;
    bool ok = true;
    float f[4];
    int   n[3];

    if (key == "albedo")               { ok = readFloat(parser, albedo[0]) && readFloat(parser, albedo[1]) && readFloat(parser, albedo[2]); }
    else if (key == "roughness")       { ok = readFloat(parser, roughness[0]) && readFloat(parser, roughness[1]); }
    else if (key == "absorption")      { ok = readFloat(parser, absorptionColor[0]) && readFloat(parser, absorptionColor[1]) && readFloat(parser, absorptionColor[2]); }
    else if (key == "absorptionScale") { ok = readFloat(parser, absorptionScale); }
    else if (key == "ior")             { ok = readFloat(parser, ior); }
    else if (key == "thinwalled")      { ok = readInt(parser, n[0]); if (ok) thinwalled = (n[0] != 0); }
    else if (key == "material")
    {
      std::string reference, bsdf;
      parser.nextToken(reference); // any token type is accepted as a name, the last duplicate wins
      parser.nextToken(bsdf);

      TwkMaterialGUI m;
      m.indexBSDF = TWK_INDEX_BRDF_DIFFUSE;
      if      (bsdf == "brdf_diffuse")   m.indexBSDF = TWK_INDEX_BRDF_DIFFUSE;
      else if (bsdf == "brdf_specular")  m.indexBSDF = TWK_INDEX_BRDF_SPECULAR;
      else if (bsdf == "bsdf_specular")  m.indexBSDF = TWK_INDEX_BSDF_SPECULAR;
      else if (bsdf == "brdf_ggx_smith") m.indexBSDF = TWK_INDEX_BRDF_GGX_SMITH;
      else if (bsdf == "bsdf_ggx_smith") m.indexBSDF = TWK_INDEX_BSDF_GGX_SMITH;
      else warnings.push_back("unknown material " + bsdf);

      for (int k = 0; k < 3; ++k) { m.albedo[k] = albedo[k]; m.absorptionColor[k] = absorptionColor[k]; }
      m.roughness[0] = roughness[0]; m.roughness[1] = roughness[1];
      m.absorptionScale = absorptionScale;
      m.ior = ior;
      m.thinwalled = thinwalled ? 1 : 0;
      m.useAlbedoTexture = 0;
      m.useCutoutTexture = 0;

      const int indexMaterial = (int) materials.size();
      materials.push_back(m);
      materialNames.push_back(reference);
      m_materialReferences[reference] = indexMaterial;
    }
    else if (key == "identity") { xform.identity(); }
    else if (key == "push")     { xform.push(); }
    else if (key == "pop")      { if (!xform.pop()) warnings.push_back("pop on empty stack, resetting to identity"); }
    else if (key == "rotate")
    {
      ok = readFloat(parser, f[0]) && readFloat(parser, f[1]) && readFloat(parser, f[2]) && readFloat(parser, f[3]);
      if (ok) xform.rotate(f[0], f[1], f[2], f[3]);
    }
    else if (key == "scale")
    {
      ok = readFloat(parser, f[0]) && readFloat(parser, f[1]) && readFloat(parser, f[2]);
      if (ok) xform.scale(f[0], f[1], f[2]);
    }
    else if (key == "translate")
    {
      ok = readFloat(parser, f[0]) && readFloat(parser, f[1]) && readFloat(parser, f[2]);
      if (ok) xform.translate(f[0], f[1], f[2]);
    }
    else if (key == "model")
    {
      std::string kind;
      if (parser.nextToken(kind) != TOKEN_ID) { error = "scene description: 'model' expects a model type"; return false; }

      std::ostringstream geometryKey;
      std::string reference;
      std::shared_ptr<TrianglesNode> geometry;
      float trafo[12];
      xform.current().toAffine3x4(trafo);

      // Identical procedural meshes are shared through a string key → instancing (Application.cpp:1705-1723).
      auto lookup = [&](const std::string& k) -> bool
      {
        std::map<std::string, unsigned int>::const_iterator it = m_geometryKeys.find(k);
        if (it != m_geometryKeys.end()) { geometry = geometries[it->second]; return true; }
        const unsigned int id = (unsigned int) geometries.size();
        m_geometryKeys[k] = id;
        geometry = std::make_shared<TrianglesNode>(id);
        geometries.push_back(geometry);
        return false;
      };

      if (kind == "plane")
      {
        ok = readInt(parser, n[0]) && readInt(parser, n[1]) && readInt(parser, n[2]);
        if (ok)
        {
          parser.nextToken(reference);
          geometryKey << "plane_" << (unsigned int) n[0] << "_" << (unsigned int) n[1] << "_" << (unsigned int) n[2];
          if (!lookup(geometryKey.str())) makePlane(geometry->mesh, (unsigned int) n[0], (unsigned int) n[1], (unsigned int) n[2]);
          appendInstance(geometry, trafo, reference);
        }
      }
      else if (kind == "box")
      {
        parser.nextToken(reference);
        if (!lookup("box_1_1")) makeBox(geometry->mesh);
        appendInstance(geometry, trafo, reference);
      }
      else if (kind == "sphere")
      {
        ok = readInt(parser, n[0]) && readInt(parser, n[1]) && readFloat(parser, f[0]);
        if (ok)
        {
          parser.nextToken(reference);
          geometryKey << "sphere_" << (unsigned int) n[0] << "_" << (unsigned int) n[1] << "_" << f[0];
          if (!lookup(geometryKey.str())) makeSphere(geometry->mesh, (unsigned int) n[0], (unsigned int) n[1], 1.0f, f[0] * 3.14159265358979323846f);
          appendInstance(geometry, trafo, reference);
        }
      }
      else if (kind == "torus")
      {
        ok = readInt(parser, n[0]) && readInt(parser, n[1]) && readFloat(parser, f[0]) && readFloat(parser, f[1]);
        if (ok)
        {
          parser.nextToken(reference);
          geometryKey << "torus_" << (unsigned int) n[0] << "_" << (unsigned int) n[1] << "_" << f[0] << "_" << f[1];
          if (!lookup(geometryKey.str())) makeTorus(geometry->mesh, (unsigned int) n[0], (unsigned int) n[1], f[0], f[1]);
          appendInstance(geometry, trafo, reference);
        }
      }
      else if (kind == "assimp")
      {
        std::string path;
        parser.restOfLine(path);
        warnings.push_back("model assimp " + path + " skipped: mesh file import is not part of this build");
      }
      else
      {
        warnings.push_back("unknown model type " + kind);
      }
    }
    else
    {
      warnings.push_back("unknown token " + key + " ignored");
    }

    if (!ok) { error = where("scene description", parser, key); return false; }
  }
  return true;
}

// m = a * b for row-major 3x4 affine matrices (Device.cpp:1265-1281)
static void multiplyAffine(float* m, const float* a, const float* b)
{
  for (int r = 0; r < 3; ++r)
  {
    const float a0 = a[4 * r], a1 = a[4 * r + 1], a2 = a[4 * r + 2], a3 = a[4 * r + 3];
    m[4 * r + 0] = a0 * b[0] + a1 * b[4] + a2 * b[8];
    m[4 * r + 1] = a0 * b[1] + a1 * b[5] + a2 * b[9];
    m[4 * r + 2] = a0 * b[2] + a1 * b[6] + a2 * b[10];
    m[4 * r + 3] = a0 * b[3] + a1 * b[7] + a2 * b[11] + a3;
  }
}

void Application::flatten(const std::shared_ptr<SceneNode>& node, const float matrix[12], int material, int light)
{
  switch (node->kind)
  {
    case SceneNode::GROUP:
    {
      const GroupNode* group = static_cast<const GroupNode*>(node.get());
      for (const std::shared_ptr<InstanceNode>& child : group->children) flatten(child, matrix, material, light);
      break;
    }
    case SceneNode::INSTANCE:
    {
      const InstanceNode* instance = static_cast<const InstanceNode*>(node.get());
      float trafo[12];
      multiplyAffine(trafo, matrix, instance->transform);
      // the last non-negative material / light along the path wins
      if (0 <= instance->material) material = instance->material;
      if (0 <= instance->light)    light    = instance->light;
      if (instance->child) flatten(instance->child, trafo, material, light);
      break;
    }
    case SceneNode::TRIANGLES:
    {
      const TrianglesNode* triangles = static_cast<const TrianglesNode*>(node.get());
      FlatInstance fi;
      fi.geometry = (int) triangles->id;
      for (int i = 0; i < 12; ++i) fi.transform[i] = matrix[i];
      fi.material = material;
      fi.light    = light;
      instances.push_back(fi);
      break;
    }
  }
}

bool Application::buildScene(const std::string& sceneText, std::string& error)
{
  m_scene = std::make_shared<GroupNode>();
  materials.clear(); materialNames.clear(); geometries.clear(); instances.clear();
  m_materialReferences.clear(); m_geometryKeys.clear();

  createCameras();
  createLights();
  if (!loadSceneDescription(sceneText, error)) return false;

  const float identity[12] = {1, 0, 0, 0,  0, 1, 0, 0,  0, 0, 1, 0};
  flatten(m_scene, identity, -1, -1);

  for (const FlatInstance& fi : instances)
  {
    if (fi.material < 0) { error = "scene description: an instance has no material"; return false; }
  }
  if (materials.empty()) { error = "scene description: no materials"; return false; }
  return true;
}

} // namespace twk
