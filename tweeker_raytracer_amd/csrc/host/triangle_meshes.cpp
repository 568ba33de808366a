#include "triangle_meshes.h"

#include <cmath>

namespace twk {

static const float kPi = 3.14159265358979323846f; // M_PIf

static inline TwkTriangleAttributes vertexRecord(float px, float py, float pz,
                                                 float tx, float ty, float tz,
                                                 float nx, float ny, float nz,
                                                 float u, float v)
{
  TwkTriangleAttributes a;
  a.vertex[0] = px;  a.vertex[1] = py;  a.vertex[2] = pz;
  a.tangent[0] = tx; a.tangent[1] = ty; a.tangent[2] = tz;
  a.normal[0] = nx;  a.normal[1] = ny;  a.normal[2] = nz;
  a.texcoord[0] = u; a.texcoord[1] = v; a.texcoord[2] = 0.0f;
  return a;
}

// Two triangles per grid cell: (ll, lr, ur) and (ur, ul, ll).
static void gridIndices(std::vector<unsigned int>& indices, unsigned int cellsU, unsigned int cellsV, unsigned int stride)
{
  indices.reserve(indices.size() + 6u * cellsU * cellsV);
  for (unsigned int j = 0; j < cellsV; ++j)
  {
    for (unsigned int i = 0; i < cellsU; ++i)
    {
      const unsigned int ll = j * stride + i;
      const unsigned int lr = ll + 1;
      const unsigned int ul = (j + 1) * stride + i;
      const unsigned int ur = ul + 1;
      indices.push_back(ll); indices.push_back(lr); indices.push_back(ur);
      indices.push_back(ur); indices.push_back(ul); indices.push_back(ll);
    }
  }
}

void makePlane(TriangleMesh& mesh, unsigned int tessU, unsigned int tessV, unsigned int upAxis)
{
  mesh.attributes.clear();
  mesh.indices.clear();

  const float uTile = 2.0f / float(tessU);
  const float vTile = 2.0f / float(tessV);

  if (upAxis <= 2)
  {
    for (unsigned int j = 0; j <= tessV; ++j)
    {
      const float v = float(j) * vTile;
      for (unsigned int i = 0; i <= tessU; ++i)
      {
        const float u = float(i) * uTile;
        switch (upAxis)
        {
          case 0: // +x normal, yz plane, corner (0,-1,1)
            mesh.attributes.push_back(vertexRecord(0.0f + 0.0f, -1.0f + v, 1.0f + -u,  0.0f, 0.0f, -1.0f,  1.0f, 0.0f, 0.0f,  u * 0.5f, v * 0.5f));
            break;
          case 1: // +y normal, xz plane, corner (-1,0,1)
            mesh.attributes.push_back(vertexRecord(-1.0f + u, 0.0f + 0.0f, 1.0f + -v,  1.0f, 0.0f, 0.0f,  0.0f, 1.0f, 0.0f,  u * 0.5f, v * 0.5f));
            break;
          case 2: // +z normal, xy plane, corner (-1,-1,0)
            mesh.attributes.push_back(vertexRecord(-1.0f + u, -1.0f + v, 0.0f + 0.0f,  1.0f, 0.0f, 0.0f,  0.0f, 0.0f, 1.0f,  u * 0.5f, v * 0.5f));
            break;
        }
      }
    }
  }
  gridIndices(mesh.indices, tessU, tessV, tessU + 1);
}

void makeBox(TriangleMesh& mesh)
{
  mesh.attributes.clear();
  mesh.indices.clear();

  // Per face: tangent, normal and the four corners counter-clockwise seen from outside,
  // texcoords (0,0) (1,0) (1,1) (0,1). Face order: left, right, back, front, bottom, top.
  struct Face { float t[3]; float n[3]; float c[4][3]; };
  static const Face faces[6] =
  {
    { { 0, 0,  1}, {-1,  0,  0}, { {-1, -1, -1}, {-1, -1,  1}, {-1,  1,  1}, {-1,  1, -1} } },
    { { 0, 0, -1}, { 1,  0,  0}, { { 1, -1,  1}, { 1, -1, -1}, { 1,  1, -1}, { 1,  1,  1} } },
    { {-1, 0,  0}, { 0,  0, -1}, { { 1, -1, -1}, {-1, -1, -1}, {-1,  1, -1}, { 1,  1, -1} } },
    { { 1, 0,  0}, { 0,  0,  1}, { {-1, -1,  1}, { 1, -1,  1}, { 1,  1,  1}, {-1,  1,  1} } },
    { { 1, 0,  0}, { 0, -1,  0}, { {-1, -1, -1}, { 1, -1, -1}, { 1, -1,  1}, {-1, -1,  1} } },
    { { 1, 0,  0}, { 0,  1,  0}, { {-1,  1,  1}, { 1,  1,  1}, { 1,  1, -1}, {-1,  1, -1} } }
  };
  static const float uv[4][2] = { {0, 0}, {1, 0}, {1, 1}, {0, 1} };

  for (unsigned int f = 0; f < 6; ++f)
  {
    for (unsigned int k = 0; k < 4; ++k)
    {
      const Face& F = faces[f];
      mesh.attributes.push_back(vertexRecord(F.c[k][0], F.c[k][1], F.c[k][2], F.t[0], F.t[1], F.t[2], F.n[0], F.n[1], F.n[2], uv[k][0], uv[k][1]));
    }
    const unsigned int base = f * 4;
    mesh.indices.push_back(base);     mesh.indices.push_back(base + 1); mesh.indices.push_back(base + 2);
    mesh.indices.push_back(base + 2); mesh.indices.push_back(base + 3); mesh.indices.push_back(base);
  }
}

void makeSphere(TriangleMesh& mesh, unsigned int tessU, unsigned int tessV, float radius, float maxTheta)
{
  mesh.attributes.clear();
  mesh.indices.clear();
  mesh.attributes.reserve((tessU + 1) * tessV);

  const float phiStep   = 2.0f * kPi / (float) tessU;
  const float thetaStep = maxTheta / (float) (tessV - 1);

  // Rings from the south pole upwards; the seam column is duplicated with different texcoords.
  for (unsigned int lat = 0; lat < tessV; ++lat)
  {
    const float theta    = (float) lat * thetaStep;
    const float sinTheta = sinf(theta);
    const float cosTheta = cosf(theta);
    const float texv     = (float) lat / (float) (tessV - 1);

    for (unsigned int lon = 0; lon <= tessU; ++lon)
    {
      const float phi    = (float) lon * phiStep;
      const float sinPhi = sinf(phi);
      const float cosPhi = cosf(phi);
      const float texu   = (float) lon / (float) tessU;

      const float nx = cosPhi * sinTheta;
      const float ny = -cosTheta;
      const float nz = -sinPhi * sinTheta;
      mesh.attributes.push_back(vertexRecord(nx * radius, ny * radius, nz * radius,  -sinPhi, 0.0f, -cosPhi,  nx, ny, nz,  texu, texv));
    }
  }
  gridIndices(mesh.indices, tessU, tessV - 1, tessU + 1);
}

void makeTorus(TriangleMesh& mesh, unsigned int tessU, unsigned int tessV, float innerRadius, float outerRadius)
{
  mesh.attributes.clear();
  mesh.indices.clear();
  mesh.attributes.reserve((tessU + 1) * (tessV + 1));

  const float u = (float) tessU;
  const float v = (float) tessV;
  const float phiStep   = 2.0f * kPi / u;
  const float thetaStep = 2.0f * kPi / v;

  for (unsigned int lat = 0; lat <= tessV; ++lat)
  {
    const float theta    = (float) lat * thetaStep;
    const float sinTheta = sinf(theta);
    const float cosTheta = cosf(theta);
    const float ring     = innerRadius + outerRadius * cosTheta;

    for (unsigned int lon = 0; lon <= tessU; ++lon)
    {
      const float phi    = (float) lon * phiStep;
      const float sinPhi = sinf(phi);
      const float cosPhi = cosf(phi);
      mesh.attributes.push_back(vertexRecord(ring * cosPhi, outerRadius * sinTheta, ring * -sinPhi,
                                             -sinPhi, 0.0f, -cosPhi,
                                             cosPhi * cosTheta, sinTheta, -sinPhi * cosTheta,
                                             (float) lon / u, (float) lat / v));
    }
  }
  gridIndices(mesh.indices, tessU, tessV, tessU + 1);
}

void makeParallelogram(TriangleMesh& mesh, const float position[3], const float vecU[3], const float vecV[3], const float normal[3])
{
  mesh.attributes.clear();
  mesh.indices.clear();

  // tangent = normalize(vecU), reference vector_math.h:592-596 (multiply by the reciprocal length)
  const float invLen = 1.0f / sqrtf(vecU[0] * vecU[0] + vecU[1] * vecU[1] + vecU[2] * vecU[2]);
  const float t[3] = { vecU[0] * invLen, vecU[1] * invLen, vecU[2] * invLen };

  const float corner[4][3] =
  {
    { position[0],                         position[1],                         position[2] },
    { position[0] + vecU[0],               position[1] + vecU[1],               position[2] + vecU[2] },
    { position[0] + vecU[0] + vecV[0],     position[1] + vecU[1] + vecV[1],     position[2] + vecU[2] + vecV[2] },
    { position[0] + vecV[0],               position[1] + vecV[1],               position[2] + vecV[2] }
  };
  static const float uv[4][2] = { {0, 0}, {1, 0}, {1, 1}, {0, 1} };
  for (int k = 0; k < 4; ++k)
  {
    mesh.attributes.push_back(vertexRecord(corner[k][0], corner[k][1], corner[k][2], t[0], t[1], t[2], normal[0], normal[1], normal[2], uv[k][0], uv[k][1]));
  }
  const unsigned int idx[6] = {0, 1, 2, 2, 3, 0};
  mesh.indices.assign(idx, idx + 6);
}

} // namespace twk
