// Host scene layer: what rtigo3's Application builds before it hands the scene to the per-GPU Device.
//   system description  ≙ Application::loadSystemDescription   reference src/Application.cpp:1046-1299
//   lights + light mesh ≙ Application::createLights            reference src/Application.cpp:572-677
//   scene description   ≙ Application::loadSceneDescription    reference src/Application.cpp:1397-1878
//   instance flattening ≙ Device::traverseNode/multiplyMatrix  reference src/Device.cpp:1265-1331
// No GUI, no image loading, no assimp (out of scope); "model assimp" lines are skipped
// with a warning.
#pragma once
#include "../../../include/tweeker_hip.h"
#include "orbit_camera.h"
#include "triangle_meshes.h"

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace twk {

// Minimal scene graph with the reference's node kinds (inc/SceneGraph.h:47-133).
struct SceneNode
{
  enum Kind { GROUP, INSTANCE, TRIANGLES } kind;
  explicit SceneNode(Kind k) : kind(k) {}
  virtual ~SceneNode() {}
};

struct TrianglesNode : SceneNode
{
  explicit TrianglesNode(unsigned int id_) : SceneNode(TRIANGLES), id(id_) {}
  unsigned int id;
  TriangleMesh mesh;
};

struct InstanceNode : SceneNode
{
  InstanceNode() : SceneNode(INSTANCE)
  {
    for (int i = 0; i < 12; ++i) transform[i] = 0.0f;
    transform[0] = transform[5] = transform[10] = 1.0f;
  }
  int   material = -1;
  int   light    = -1;
  float transform[12];
  std::shared_ptr<SceneNode> child;
};

struct GroupNode : SceneNode
{
  GroupNode() : SceneNode(GROUP) {}
  std::vector<std::shared_ptr<InstanceNode>> children;
};

struct FlatInstance
{
  int   geometry;
  float transform[12];
  int   material;
  int   light;
};

class Application
{
public:
  Application();

  bool loadSystemDescription(const std::string& text, std::string& error);
  bool loadSceneDescription(const std::string& text, std::string& error);
  // Runs createCameras/createLights, parses the scene and flattens it. Call once after the system description.
  bool buildScene(const std::string& sceneText, std::string& error);

  void setResolution(int w, int h);
  std::string systemDescription() const;
  TwkDeviceState deviceState() const;

  // system options (Application.cpp:55-75,105-120 defaults)
  int   strategy      = 0;
  int   devicesMask   = 255;
  int   interop       = 0;
  bool  present       = false;
  int   light         = 0;
  int   miss          = 1;
  int   lensShader    = 0;
  int   nextEventEstimation = 1; // grammar extension "nextEventEstimation": ≙ USE_NEXT_EVENT_ESTIMATION (shaders/config.h:50-52)
  int   debugExceptions = 0;     // grammar extension "debugExceptions": ≙ USE_DEBUG_EXCEPTIONS (config.h:54-56)
  int   shaderVariant = 0; // grammar extension "shaderVariant": 0 rtigo3, 1 Optix7Gui light-hit rule (include/tweeker_hip.h TWK_SHADERS_*)
  int   samplesSqrt   = 1;
  int   resolution[2] = {1, 1};
  int   tileSize[2]   = {8, 8};
  int   pathLengths[2] = {0, 2};
  float epsilonFactor = 500.0f;
  float envRotation   = 0.0f;
  float clockFactor   = 1000.0f;
  std::string environment;
  std::string prefixScreenshot = "./img";
  TwkTonemapper tonemapper = {1.0f, 1.0f, {1.0f, 1.0f, 1.0f}, 1.0f, 0.0f, 1.0f, 1.0f}; // neutral (Application.cpp:111-120)
  OrbitCamera camera;

  std::vector<TwkCameraDefinition> cameras;
  std::vector<TwkLightDefinition>  lights;
  std::vector<TwkMaterialGUI>      materials;
  std::vector<std::string>         materialNames;
  std::vector<std::shared_ptr<TrianglesNode>> geometries;
  std::vector<FlatInstance>        instances;
  std::vector<std::string>         warnings;

private:
  void createCameras();
  void createLights();
  void appendInstance(std::shared_ptr<TrianglesNode> geometry, const float trafo[12], const std::string& reference);
  void flatten(const std::shared_ptr<SceneNode>& node, const float matrix[12], int material, int light);

  std::shared_ptr<GroupNode>          m_scene;
  std::map<std::string, int>          m_materialReferences;
  std::map<std::string, unsigned int> m_geometryKeys;
};

} // namespace twk
