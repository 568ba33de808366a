// Orbit camera → pinhole frustum (P, U, V, W).
// Follows reference apps/rtigo3/src/Camera.cpp:187-216 (Camera::getFrustum): phi/theta in [0,1] are
// fractions of the full longitude/latitude range, fov in degrees, the image plane spans U (scaled by the
// aspect ratio) and V at unit distance along W.
#pragma once
#include "../../../include/tweeker_hip.h"
#include <cmath>

namespace twk {

struct OrbitCamera
{
  float center[3] = {0.0f, 0.0f, 0.0f};
  float distance  = 10.0f; // reference Camera.cpp:39-42 defaults
  float phi       = 0.75f;
  float theta     = 0.6f;
  float fov       = 60.0f;
  float aspect    = 1.0f;

  void setResolution(int w, int h) // Camera.cpp:69-79
  {
    const int ww = (0 < w) ? w : 1;
    const int hh = (0 < h) ? h : 1;
    aspect = float(ww) / float(hh);
  }

  TwkCameraDefinition frustum() const
  {
    const float kPi = 3.14159265358979323846f;
    const float cosPhi   = cosf(phi * 2.0f * kPi);
    const float sinPhi   = sinf(phi * 2.0f * kPi);
    const float cosTheta = cosf(theta * kPi);
    const float sinTheta = sinf(theta * kPi);

    const float n[3] = { cosPhi * sinTheta, -cosTheta, -sinPhi * sinTheta };
    const float tanFovHalf = tanf((fov * 0.5f) * kPi / 180.0f);

    TwkCameraDefinition c;
    c.P[0] = center[0] + distance * n[0];
    c.P[1] = center[1] + distance * n[1];
    c.P[2] = center[2] + distance * n[2];
    // U = aspect * (-sinPhi, 0, -cosPhi) * tanFovHalf, evaluated left to right
    c.U[0] = (aspect * -sinPhi) * tanFovHalf;
    c.U[1] = (aspect * 0.0f)    * tanFovHalf;
    c.U[2] = (aspect * -cosPhi) * tanFovHalf;
    c.V[0] = (cosTheta * cosPhi)  * tanFovHalf;
    c.V[1] = sinTheta             * tanFovHalf;
    c.V[2] = (cosTheta * -sinPhi) * tanFovHalf;
    c.W[0] = -n[0]; c.W[1] = -n[1]; c.W[2] = -n[2];
    return c;
  }
};

} // namespace twk
