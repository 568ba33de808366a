// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path; nothing under tweeker_raytracer_amd/
// may include, link or call this. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline use it.
//
// Portable single-precision elementary functions.
//
// Why they exist: the reference's shaders call sinf/cosf/expf/atan2f/acosf/atanf
// (bxdf_diffuse.cu:52-56, light_sample.cu:46-48,133-138, bxdf_ggx_smith.cu:82-84,99-104,152-163,
// miss.cu:84-85, raygeneration.cu:98) which nvcc compiled with --use_fast_math
// (apps/rtigo3/CMakeLists.txt:165-184). Neither glibc's nor ROCm's libm reproduces those bits, and
// they do not reproduce each other. To make "oracle vs HIP" a BIT-EXACT comparison, both sides
// evaluate the same published algorithm — Cephes single precision (S. Moshier, sinf.c/cosf.c/expf.c/
// atanf.c/asinf.c) — with every operation written out (no FMA contraction: build with
// -ffp-contract=off). tests/test_oracle_math.py bounds the distance to libm (≤ 2 ulp on the
// argument ranges the shaders use), so the choice stays inside any stated radiance tolerance.
// ORC_USE_LIBM switches the oracle to glibc for that comparison.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace orc {

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

#ifdef ORC_USE_LIBM
static inline float pm_sinf(float x) { return ::sinf(x); }
static inline float pm_cosf(float x) { return ::cosf(x); }
static inline float pm_expf(float x) { return ::expf(x); }
static inline float pm_atanf(float x) { return ::atanf(x); }
static inline float pm_atan2f(float y, float x) { return ::atan2f(y, x); }
static inline float pm_acosf(float x) { return ::acosf(x); }
static inline float pm_logf(float x) { return ::logf(x); }
static inline float pm_powf(float x, float y) { return ::powf(x, y); }
#else

// Cephes sinf/cosf: octant reduction with a 3-part pi/4, degree-7/8 minimax polynomials.
// Valid for |x| < 8192; the shaders stay within [-2pi, 4pi].
static inline void pm_reduce(float ax, float& r, int& j)
{
  j = (int) (ax * 1.27323954473516f); // 4/pi
  j = (j + 1) & ~1;                   // round up to even: map to octant pairs
  const float y = (float) j;
  r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}

static inline float pm_sinpoly(float r)
{
  const float z = r * r;
  return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
}

static inline float pm_cospoly(float r)
{
  const float z = r * r;
  return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}

static inline float pm_sinf(float x)
{
  float sign = 1.0f;
  float ax = x;
  if (x < 0.0f) { sign = -1.0f; ax = -x; }
  float r; int j;
  pm_reduce(ax, r, j);
  j &= 7;
  if (j > 3) { sign = -sign; j -= 4; }
  const float y = (j == 2) ? pm_cospoly(r) : pm_sinpoly(r);
  return sign * y;
}

static inline float pm_cosf(float x)
{
  float sign = 1.0f;
  const float ax = (x < 0.0f) ? -x : x;
  float r; int j;
  pm_reduce(ax, r, j);
  j &= 7;
  if (j > 3) { sign = -sign; j -= 4; }
  if (j > 1) { sign = -sign; }
  const float y = (j == 2) ? pm_sinpoly(r) : pm_cospoly(r);
  return sign * y;
}

// Cephes expf: x = n ln2 + r, degree-5 polynomial, scale by 2^n through the exponent field.
static inline float pm_expf(float x)
{
  if (x > 88.0f)  return bits2f(0x7f800000u);
  if (x < -87.0f) return 0.0f; // results below the normal range are flushed; absorption never needs them
  float z = floorf(1.44269504088896341f * x + 0.5f);
  const int n = (int) z;
  x = x - z * 0.693359375f;
  x = x - z * -2.12194440e-4f;
  z = x * x;
  float p = ((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f;
  p = p * z + x + 1.0f;
  return p * bits2f((uint32_t) (n + 127) << 23);
}

// Cephes atanf.
static inline float pm_atanf(float xx)
{
  float sign = 1.0f;
  float x = xx;
  if (xx < 0.0f) { sign = -1.0f; x = -xx; }
  float y;
  if (x > 2.414213562373095f)       { y = 1.5707963267948966192f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
  else                              { y = 0.0f; }
  const float z = x * x;
  y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
  return sign * y;
}

// Cephes atan2f(y, x) quadrant logic on top of pm_atanf.
static inline float pm_atan2f(float y, float x)
{
  const float PIF = 3.14159265358979323846f;
  const float PIO2F = 1.5707963267948966192f;
  if (x == 0.0f)
  {
    if (y > 0.0f) return PIO2F;
    if (y < 0.0f) return -PIO2F;
    return 0.0f;
  }
  if (y == 0.0f)
  {
    return (x < 0.0f) ? PIF : 0.0f;
  }
  float w = 0.0f;
  if (x < 0.0f) w = (y < 0.0f) ? -PIF : PIF;
  return w + pm_atanf(y / x);
}

// Cephes asinf / acosf.
static inline float pm_asinf(float xx)
{
  float sign = 1.0f;
  float a = xx;
  if (xx < 0.0f) { sign = -1.0f; a = -xx; }
  if (a > 1.0f) return bits2f(0x7fc00000u);
  float x, z;
  int flag;
  if (a > 0.5f) { z = 0.5f * (1.0f - a); x = sqrtf(z); flag = 1; }
  else          { x = a; z = x * x; flag = 0; }
  z = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
  if (flag != 0) { z = z + z; z = 1.5707963267948966192f - z; }
  return sign * z;
}

static inline float pm_acosf(float x)
{
  if (x < -1.0f || x > 1.0f) return bits2f(0x7fc00000u);
  if (x < -0.5f) return 3.14159265358979323846f - 2.0f * pm_asinf(sqrtf(0.5f * (1.0f + x)));
  if (x > 0.5f)  return 2.0f * pm_asinf(sqrtf(0.5f * (1.0f - x)));
  return 1.5707963267948966192f - pm_asinf(x);
}
// Cephes logf (tonemapper, Application.cpp:2283-2287): frexp split, mantissa folded to [sqrt(1/2), sqrt(2)),
// polynomial of degree 9; subnormal arguments are treated as zero.
static inline float pm_logf(float xx)
{
  if (xx < 0.0f) return bits2f(0x7fc00000u);
  if (xx < 1.17549435e-38f) return bits2f(0xff800000u);
  if (xx > 3.40282347e+38f) return xx;
  const uint32_t u = f2bits(xx);
  int e = (int) (u >> 23) - 126;
  float x = bits2f((u & 0x007fffffu) | 0x3f000000u); // [0.5, 1)
  if (x < 0.707106781186547524f) { e = e - 1; x = x + x - 1.0f; }
  else { x = x - 1.0f; }
  const float z = x * x;
  float y = 7.0376836292e-2f;
  y = y * x - 1.1514610310e-1f;
  y = y * x + 1.1676998740e-1f;
  y = y * x - 1.2420140846e-1f;
  y = y * x + 1.4249322787e-1f;
  y = y * x - 1.6668057665e-1f;
  y = y * x + 2.0000714765e-1f;
  y = y * x - 2.4999993993e-1f;
  y = y * x + 3.3333331174e-1f;
  y = y * x * z;
  const float fe = (float) e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  float r = x + y;
  r = r + 0.693359375f * fe;
  return r;
}

// pow restricted to what the tonemapper needs (bases >= 0): exp(y log x), 0^y = 0 for y > 0, x^0 = 1.
static inline float pm_powf(float x, float y)
{
  if (y == 0.0f) return 1.0f;
  if (y == 1.0f) return x; // exact like libm's powf: neutral tonemapper settings leave the colour unchanged
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : bits2f(0x7f800000u);
  if (!(x > 0.0f)) return bits2f(0x7fc00000u);
  return pm_expf(y * pm_logf(x));
}
#endif // ORC_USE_LIBM

} // namespace orc
