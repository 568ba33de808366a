// Procedural triangle meshes of rtigo3's scene description ("model plane|box|sphere|torus", area light
// parallelogram). Vertex order, index order and float arithmetic follow the reference generators so the
// acceleration structure sees identical input:
//   plane          apps/rtigo3/src/Plane.cpp:37-134
//   box            apps/rtigo3/src/Box.cpp:37-183
//   sphere         apps/rtigo3/src/Sphere.cpp:37-104
//   torus          apps/rtigo3/src/Torus.cpp:49-109
//   parallelogram  apps/rtigo3/src/Parallelogram.cpp:46-80
#pragma once
#include "../../../include/tweeker_hip.h"
#include <vector>

namespace twk {

struct TriangleMesh
{
  std::vector<TwkTriangleAttributes> attributes;
  std::vector<unsigned int>          indices;
};

void makePlane(TriangleMesh& mesh, unsigned int tessU, unsigned int tessV, unsigned int upAxis);
void makeBox(TriangleMesh& mesh);
void makeSphere(TriangleMesh& mesh, unsigned int tessU, unsigned int tessV, float radius, float maxTheta);
void makeTorus(TriangleMesh& mesh, unsigned int tessU, unsigned int tessV, float innerRadius, float outerRadius);
void makeParallelogram(TriangleMesh& mesh, const float position[3], const float vecU[3], const float vecV[3], const float normal[3]);

} // namespace twk
