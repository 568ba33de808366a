// Transform state of the scene description ("identity/push/pop/rotate/scale/translate").
// rtigo3 keeps a row-vector 4x4 matrix (nvpro-pipeline dp::math), post-multiplies every new transform
// (curMatrix *= M, reference Application.cpp:1612-1690) and emits the transposed upper 3x4 block per
// instance (Application.cpp:1353-1359). The float arithmetic below follows dp::math so the instance
// matrices are identical: axis normalised in float (dp/math/Vecnt.h:822-830), degrees→radians with a
// float PI/180 (dp/math/math.h:52,91-94), quaternion from axis/angle (dp/math/Quatt.h:326-333),
// quaternion→3x3 (dp/math/Matmnt.h:1099-1111), 4x4 product as an in-order accumulation from zero
// (dp/math/Matmnt.h:1005-1020).
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace twk {

struct Matrix44
{
  float m[4][4];

  static Matrix44 identity()
  {
    Matrix44 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.0f : 0.0f;
    return r;
  }

  Matrix44 times(const Matrix44& b) const
  {
    Matrix44 r;
    for (int i = 0; i < 4; ++i)
    {
      for (int j = 0; j < 4; ++j)
      {
        float sum = 0.0f;
        for (int l = 0; l < 4; ++l) sum += m[i][l] * b.m[l][j];
        r.m[i][j] = sum;
      }
    }
    return r;
  }

  // Row-major 3x4 object→world matrix as the renderer consumes it (column-vector convention).
  void toAffine3x4(float t[12]) const
  {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) t[4 * r + c] = m[c][r];
  }
};

class TransformStack
{
public:
  TransformStack() : m_current(Matrix44::identity()) {}

  void identity() { m_current = Matrix44::identity(); }
  void push() { m_stack.push_back(m_current); }
  bool pop()
  {
    if (m_stack.empty()) { m_current = Matrix44::identity(); return false; } // Application.cpp:1602-1608
    m_current = m_stack.back();
    m_stack.pop_back();
    return true;
  }

  void rotate(float ax, float ay, float az, float degrees)
  {
    // axis.normalize(): single precision, in-order sum of squares, divide when the length exceeds epsilon
    // (dp/math/Vecnt.h:822-830; the reference-built helper in oracle/_ref pins that the float path is taken)
    float x0 = ax, y0 = ay, z0 = az;
    float sq = 0.0f; sq += x0 * x0; sq += y0 * y0; sq += z0 * z0;
    const float norm = sqrtf(sq);
    if (std::numeric_limits<float>::epsilon() < norm) { x0 /= norm; y0 /= norm; z0 /= norm; }

    const float PI = (float) (4 * atan(1.0));
    const float angle = degrees * (PI / 180);

    // Quatt(axis, angle): sin / cos of the half angle are evaluated in double and rounded (dp/math/Quatt.h:326-333
    // calls the C functions with a float argument)
    const float s = (float) sin((double) (0.5f * angle));
    const float x = x0 * s, y = y0 * s, z = z0 * s;
    const float w = (float) cos((double) (0.5f * angle));

    Matrix44 r = Matrix44::identity();
    r.m[0][0] = 1 - 2 * (y * y + z * z); r.m[0][1] = 2 * (x * y + z * w);     r.m[0][2] = 2 * (x * z - y * w);
    r.m[1][0] = 2 * (x * y - z * w);     r.m[1][1] = 1 - 2 * (x * x + z * z); r.m[1][2] = 2 * (y * z + x * w);
    r.m[2][0] = 2 * (x * z + y * w);     r.m[2][1] = 2 * (y * z - x * w);     r.m[2][2] = 1 - 2 * (x * x + y * y);
    m_current = m_current.times(r);
  }

  void scale(float sx, float sy, float sz)
  {
    Matrix44 s = Matrix44::identity();
    s.m[0][0] = sx; s.m[1][1] = sy; s.m[2][2] = sz;
    m_current = m_current.times(s);
  }

  void translate(float tx, float ty, float tz)
  {
    Matrix44 t = Matrix44::identity();
    t.m[3][0] = tx; t.m[3][1] = ty; t.m[3][2] = tz;
    m_current = m_current.times(t);
  }

  const Matrix44& current() const { return m_current; }

private:
  Matrix44 m_current;
  std::vector<Matrix44> m_stack;
};

} // namespace twk
