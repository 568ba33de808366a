// C entry points of the host scene layer declared in include/tweeker_hip.h (twk_app_*, twk_mesh_*,
// twk_camera_frustum, twk_tile_column, twk_launch_width). Pure host code: usable without a GPU.
#include "application.h"
#include "description_parser.h"
#include "image_files.h"
#include "../error_state.h"

#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <vector>

using namespace twk;

struct TwkApp_t
{
  Application app;
};

static bool readTextFile(const char* path, std::string& out)
{
  std::ifstream in(path, std::ios::binary);
  if (!in) return false;
  std::stringstream ss;
  ss << in.rdbuf();
  out = ss.str();
  return true;
}

static int copyMeshOut(const TriangleMesh& mesh, TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
{
  if (!numAttr || !numIdx) return twkSetError(TWK_ERROR_INVALID_VALUE, "mesh: numAttr/numIdx must not be NULL");
  // Two-call protocol: with NULL arrays only the sizes are returned; otherwise *numAttr/*numIdx are capacities.
  if (attr) { if (*numAttr < mesh.attributes.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "mesh: attribute capacity too small"); memcpy(attr, mesh.attributes.data(), sizeof(TwkTriangleAttributes) * mesh.attributes.size()); }
  if (idx)  { if (*numIdx < mesh.indices.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "mesh: index capacity too small"); memcpy(idx, mesh.indices.data(), sizeof(unsigned int) * mesh.indices.size()); }
  *numAttr = mesh.attributes.size();
  *numIdx  = mesh.indices.size();
  return TWK_SUCCESS;
}

extern "C" {

int twk_app_create_from_strings(TwkApp* out, const char* systemDescription, const char* sceneDescription)
try
{
  if (!out || !systemDescription || !sceneDescription) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_create: NULL argument");
  *out = nullptr;
  TwkApp_t* a = new TwkApp_t();
  std::string error;
  if (!a->app.loadSystemDescription(systemDescription, error)) { delete a; return twkSetError(TWK_ERROR_PARSE, error); }
  if (!a->app.buildScene(sceneDescription, error))              { delete a; return twkSetError(TWK_ERROR_PARSE, error); }
  *out = a;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_create_from_strings")

int twk_app_create(TwkApp* out, const char* systemDescriptionFile, const char* sceneDescriptionFile)
try
{
  if (!out || !systemDescriptionFile || !sceneDescriptionFile) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_create: NULL argument");
  *out = nullptr;
  std::string sys, scene;
  if (!readTextFile(systemDescriptionFile, sys))  return twkSetError(TWK_ERROR_IO, std::string("failed to open system description file ") + systemDescriptionFile);
  if (!readTextFile(sceneDescriptionFile, scene)) return twkSetError(TWK_ERROR_IO, std::string("failed to open scene description file ") + sceneDescriptionFile);
  return twk_app_create_from_strings(out, sys.c_str(), scene.c_str());
}
TWK_CATCH("twk_app_create")

int twk_app_destroy(TwkApp app) { delete app; return TWK_SUCCESS; }

int twk_app_info(TwkApp app, TwkAppInfo* info)
try
{
  if (!app || !info) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_info: NULL argument");
  const Application& a = app->app;
  info->strategy = a.strategy; info->devicesMask = a.devicesMask; info->light = a.light; info->miss = a.miss;
  info->lensShader = a.lensShader; info->samplesSqrt = a.samplesSqrt;
  for (int k = 0; k < 2; ++k) { info->resolution[k] = a.resolution[k]; info->tileSize[k] = a.tileSize[k]; info->pathLengths[k] = a.pathLengths[k]; }
  info->epsilonFactor = a.epsilonFactor; info->envRotation = a.envRotation; info->clockFactor = a.clockFactor;
  for (int k = 0; k < 3; ++k) info->center[k] = a.camera.center[k];
  info->phi = a.camera.phi; info->theta = a.camera.theta; info->fov = a.camera.fov; info->distance = a.camera.distance;
  info->numCameras = (int) a.cameras.size(); info->numLights = (int) a.lights.size(); info->numMaterials = (int) a.materials.size();
  info->numGeometries = (int) a.geometries.size(); info->numInstances = (int) a.instances.size();
  info->shaderVariant = a.shaderVariant; info->nextEventEstimation = a.nextEventEstimation; info->debugExceptions = a.debugExceptions;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_info")

int twk_app_set_resolution(TwkApp app, int width, int height)
try
{
  if (!app) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_set_resolution: NULL app");
  app->app.setResolution(width, height);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_set_resolution")

int twk_app_get_state(TwkApp app, TwkDeviceState* state)
try
{
  if (!app || !state) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_state: NULL argument");
  *state = app->app.deviceState();
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_state")

int twk_app_get_cameras(TwkApp app, TwkCameraDefinition* out, int capacity)
try
{
  if (!app || !out || capacity < (int) app->app.cameras.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_cameras: bad arguments");
  memcpy(out, app->app.cameras.data(), sizeof(TwkCameraDefinition) * app->app.cameras.size());
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_cameras")

int twk_app_get_lights(TwkApp app, TwkLightDefinition* out, int capacity)
try
{
  if (!app || capacity < (int) app->app.lights.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_lights: bad arguments");
  if (!app->app.lights.empty()) memcpy(out, app->app.lights.data(), sizeof(TwkLightDefinition) * app->app.lights.size());
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_lights")

int twk_app_get_materials(TwkApp app, TwkMaterialGUI* out, int capacity)
try
{
  if (!app || !out || capacity < (int) app->app.materials.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_materials: bad arguments");
  memcpy(out, app->app.materials.data(), sizeof(TwkMaterialGUI) * app->app.materials.size());
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_materials")

int twk_app_get_geometry_sizes(TwkApp app, int idGeometry, size_t* numAttributes, size_t* numIndices)
try
{
  if (!app || idGeometry < 0 || idGeometry >= (int) app->app.geometries.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_geometry_sizes: bad geometry id");
  *numAttributes = app->app.geometries[idGeometry]->mesh.attributes.size();
  *numIndices    = app->app.geometries[idGeometry]->mesh.indices.size();
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_geometry_sizes")

int twk_app_get_geometry(TwkApp app, int idGeometry, TwkTriangleAttributes* attributes, unsigned int* indices)
try
{
  if (!app || idGeometry < 0 || idGeometry >= (int) app->app.geometries.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_geometry: bad geometry id");
  const TriangleMesh& mesh = app->app.geometries[idGeometry]->mesh;
  if (attributes) memcpy(attributes, mesh.attributes.data(), sizeof(TwkTriangleAttributes) * mesh.attributes.size());
  if (indices)    memcpy(indices, mesh.indices.data(), sizeof(unsigned int) * mesh.indices.size());
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_geometry")

int twk_app_get_instance(TwkApp app, int idInstance, int* idGeometry, float transform[12], int* idMaterial, int* idLight)
try
{
  if (!app || idInstance < 0 || idInstance >= (int) app->app.instances.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_instance: bad instance id");
  const FlatInstance& fi = app->app.instances[idInstance];
  if (idGeometry) *idGeometry = fi.geometry;
  if (transform)  memcpy(transform, fi.transform, sizeof(float) * 12);
  if (idMaterial) *idMaterial = fi.material;
  if (idLight)    *idLight = fi.light;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_instance")

// ≙ Application.cpp:303 (initState) and :328-332 (initCameras/Lights/Materials/Scene)
int twk_app_init_device(TwkApp app, TwkDevice dev)
try
{
  if (!app || !dev) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_init_device: NULL argument");
  const Application& a = app->app;
  int rc;
  TwkDeviceState state = a.deviceState();
  if ((rc = twk_set_state(dev, &state))) return rc;
  if ((rc = twk_set_shader_variant(dev, a.shaderVariant))) return rc;
  if ((rc = twk_set_next_event_estimation(dev, a.nextEventEstimation))) return rc;
  if ((rc = twk_set_debug_exceptions(dev, a.debugExceptions))) return rc;
  if ((rc = twk_init_cameras(dev, a.cameras.data(), (int) a.cameras.size()))) return rc;
  if ((rc = twk_init_lights(dev, a.lights.data(), (int) a.lights.size()))) return rc;
  if ((rc = twk_init_materials(dev, a.materials.data(), (int) a.materials.size()))) return rc;
  if ((rc = twk_clear_scene(dev))) return rc;
  for (const std::shared_ptr<TrianglesNode>& g : a.geometries)
  {
    int id = -1;
    if ((rc = twk_add_geometry(dev, g->mesh.attributes.data(), g->mesh.attributes.size(), g->mesh.indices.data(), g->mesh.indices.size(), &id))) return rc;
    if (id != (int) g->id) return twkSetError(TWK_ERROR_INVALID_STATE, "twk_app_init_device: geometry ids out of order");
  }
  for (const FlatInstance& fi : a.instances)
  {
    if ((rc = twk_add_instance(dev, fi.geometry, fi.transform, fi.material, fi.light, nullptr))) return rc;
  }
  return twk_build(dev);
}
TWK_CATCH("twk_app_init_device")

int twk_app_system_description(TwkApp app, char* out, size_t capacity, size_t* length)
try
{
  if (!app) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_system_description: NULL argument");
  const std::string s = app->app.systemDescription();
  if (length) *length = s.size();
  if (!out) return TWK_SUCCESS;
  if (s.size() + 1 > capacity) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_system_description: buffer too small");
  memcpy(out, s.c_str(), s.size() + 1);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_system_description")

int twk_app_get_tonemapper(TwkApp app, TwkTonemapper* tm)
try
{
  if (!app || !tm) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_tonemapper: NULL argument");
  *tm = app->app.tonemapper;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_tonemapper")

// ≙ Application::screenshot's path (Application.cpp:2235-2239,2256,2303) with getDateTime's Linux branch (:1927-2008),
// which prints tm_year (years since 1900) and tm_mon (0-based) as they are and "000" for the milliseconds.
int twk_app_screenshot_path(TwkApp app, int tonemap, char* out, size_t capacity)
try
{
  if (!app || !out) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_screenshot_path: NULL argument");
  time_t rawtime;
  time(&rawtime);
  struct tm ts;
  localtime_r(&rawtime, &ts);
  std::ostringstream path;
  const int spp = app->app.samplesSqrt * app->app.samplesSqrt;
  path << app->app.prefixScreenshot << "_" << spp << "spp_";
  path << ts.tm_year;
  if (ts.tm_mon < 10) path << '0';
  path << ts.tm_mon;
  if (ts.tm_mday < 10) path << '0';
  path << ts.tm_mday << '_';
  if (ts.tm_hour < 10) path << '0';
  path << ts.tm_hour;
  if (ts.tm_min < 10) path << '0';
  path << ts.tm_min;
  if (ts.tm_sec < 10) path << '0';
  path << ts.tm_sec << '_' << "000";
  path << (tonemap ? ".png" : ".hdr");
  const std::string s = path.str();
  if (s.size() + 1 > capacity) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_screenshot_path: buffer too small");
  memcpy(out, s.c_str(), s.size() + 1);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_screenshot_path")

int twk_load_image(const char* path, int* width, int* height, float* rgba, size_t capacityFloats)
try
{
  if (!path || !width || !height) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_load_image: NULL argument");
  int w = 0, h = 0;
  std::vector<float> pixels;
  std::string error;
  if (!loadImageRgba32f(path, w, h, pixels, error)) return twkSetError(TWK_ERROR_IO, "twk_load_image: " + error);
  *width = w; *height = h;
  if (!rgba) return TWK_SUCCESS;
  if (capacityFloats < pixels.size()) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_load_image: buffer smaller than width*height*4 floats");
  memcpy(rgba, pixels.data(), pixels.size() * sizeof(float));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_load_image")

int twk_app_get_environment(TwkApp app, char* out, size_t capacity)
try
{
  if (!app || !out || capacity == 0) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_environment: NULL argument");
  const std::string& s = app->app.environment;
  if (s.size() + 1 > capacity) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_app_get_environment: buffer too small");
  memcpy(out, s.c_str(), s.size() + 1);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_app_get_environment")

int twk_write_png_rgb8(const char* path, int width, int height, const unsigned char* rgb8, int bottomUp)
try
{
  if (!path || !rgb8 || width <= 0 || height <= 0) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_write_png_rgb8: bad arguments");
  std::string error;
  if (!writePngRgb8(path, width, height, rgb8, bottomUp != 0, error)) return twkSetError(TWK_ERROR_IO, error);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_write_png_rgb8")

int twk_write_hdr_rgba32f(const char* path, int width, int height, const float* rgba, int bottomUp)
try
{
  if (!path || !rgba || width <= 0 || height <= 0) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_write_hdr_rgba32f: bad arguments");
  std::string error;
  if (!writeHdrRgba32f(path, width, height, rgba, bottomUp != 0, error)) return twkSetError(TWK_ERROR_IO, error);
  return TWK_SUCCESS;
}
TWK_CATCH("twk_write_hdr_rgba32f")

int twk_mesh_plane(unsigned int tessU, unsigned int tessV, unsigned int upAxis, TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
try
{
  if (tessU < 1 || tessV < 1 || upAxis > 2) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_mesh_plane: tessellation must be >= 1, upAxis 0..2");
  TriangleMesh m; makePlane(m, tessU, tessV, upAxis); return copyMeshOut(m, attr, numAttr, idx, numIdx);
}
TWK_CATCH("twk_mesh_plane")

int twk_mesh_box(TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
try
{
  TriangleMesh m; makeBox(m); return copyMeshOut(m, attr, numAttr, idx, numIdx);
}
TWK_CATCH("twk_mesh_box")

int twk_mesh_sphere(unsigned int tessU, unsigned int tessV, float radius, float maxTheta, TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
try
{
  if (tessU < 3 || tessV < 3) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_mesh_sphere: tessellation must be >= 3");
  TriangleMesh m; makeSphere(m, tessU, tessV, radius, maxTheta); return copyMeshOut(m, attr, numAttr, idx, numIdx);
}
TWK_CATCH("twk_mesh_sphere")

int twk_mesh_torus(unsigned int tessU, unsigned int tessV, float innerRadius, float outerRadius, TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
try
{
  if (tessU < 3 || tessV < 3) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_mesh_torus: tessellation must be >= 3");
  TriangleMesh m; makeTorus(m, tessU, tessV, innerRadius, outerRadius); return copyMeshOut(m, attr, numAttr, idx, numIdx);
}
TWK_CATCH("twk_mesh_torus")

int twk_mesh_parallelogram(const float position[3], const float vecU[3], const float vecV[3], const float normal[3], TwkTriangleAttributes* attr, size_t* numAttr, unsigned int* idx, size_t* numIdx)
try
{
  TriangleMesh m; makeParallelogram(m, position, vecU, vecV, normal); return copyMeshOut(m, attr, numAttr, idx, numIdx);
}
TWK_CATCH("twk_mesh_parallelogram")

int twk_camera_frustum(const float center[3], float phi, float theta, float fov, float distance, float aspect, TwkCameraDefinition* out)
try
{
  if (!center || !out) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_camera_frustum: NULL argument");
  OrbitCamera c;
  c.center[0] = center[0]; c.center[1] = center[1]; c.center[2] = center[2];
  c.phi = phi; c.theta = theta; c.fov = fov; c.distance = distance; c.aspect = aspect;
  *out = c.frustum();
  return TWK_SUCCESS;
}
TWK_CATCH("twk_camera_frustum")

// ≙ calculateTileShift (Device.cpp:1172-1189) + distribute() (raygeneration.cu:152-164)
int twk_tile_column(int launchX, int launchY, const int tileSize[2], int deviceCount, int deviceIndex, int* pixelX)
try
{
  if (!tileSize || !pixelX || deviceCount < 1 || tileSize[0] < 1 || tileSize[1] < 1 ||
      (tileSize[0] & (tileSize[0] - 1)) || (tileSize[1] & (tileSize[1] - 1)))
    return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_tile_column: tile size must be a power of two, deviceCount >= 1");
  int xShift = 0; while ((tileSize[0] & (1 << xShift)) == 0) ++xShift;
  int yShift = 0; while ((tileSize[1] & (1 << yShift)) == 0) ++yShift;
  const unsigned int xBlock = (unsigned int) launchX >> xShift;
  const unsigned int yBlock = (unsigned int) launchY >> yShift;
  const unsigned int xTile  = xBlock * deviceCount + ((deviceIndex + yBlock) % deviceCount);
  *pixelX = (int) (xTile * tileSize[0] + ((unsigned int) launchX & (tileSize[0] - 1)));
  return TWK_SUCCESS;
}
TWK_CATCH("twk_tile_column")

int twk_launch_width(int width, int tileSizeX, int deviceCount, int* launchWidth)
try
{
  if (!launchWidth || deviceCount < 1 || tileSizeX < 1) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_launch_width: bad arguments");
  const int w    = (width + deviceCount - 1) / deviceCount;
  const int mask = tileSizeX - 1;
  *launchWidth = (w + mask) & ~mask;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_launch_width")

int twk_parse_tokens(const char* text, char* out, size_t capacity, int* numTokens)
try
{
  if (!text || !out || !numTokens) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_parse_tokens: NULL argument");
  DescriptionParser parser;
  parser.loadString(text);
  std::string result, token;
  int count = 0;
  TokenType t;
  while ((t = parser.nextToken(token)) != TOKEN_EOF)
  {
    result += std::to_string((int) t) + " " + token + "\n";
    ++count;
    if (t == TOKEN_UNKNOWN) break;
  }
  if (result.size() + 1 > capacity) return twkSetError(TWK_ERROR_INVALID_VALUE, "twk_parse_tokens: output buffer too small");
  memcpy(out, result.c_str(), result.size() + 1);
  *numTokens = count;
  return TWK_SUCCESS;
}
TWK_CATCH("twk_parse_tokens")

} // extern "C"
