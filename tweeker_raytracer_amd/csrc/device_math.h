// Single-precision elementary functions and small-vector helpers of the HIP kernels.
//
// The reference's shaders call sinf/cosf/expf/atan2f/acosf/atanf (e.g. bxdf_diffuse.cu:52-56,
// light_sample.cu:46-48, bxdf_ggx_smith.cu:82-104, miss.cu:84-85, raygeneration.cu:98), compiled
// by nvcc with --use_fast_math. To get results that are reproducible bit for bit on any IEEE-754
// machine (and therefore checkable against a CPU oracle with ==), the kernels evaluate a fixed,
// published algorithm instead of a vendor libm: Cephes single precision (S. Moshier): octant-reduced
// sin/cos, range-reduced exp, three-interval atan, asin via sqrt identity. All kernels are built with
// -ffp-contract=off so that no multiply-add is fused behind the source's back; sqrt and division are
// the correctly rounded IEEE operations (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TWK_HD __host__ __device__ __forceinline__
#define TWK_D  __device__ __forceinline__

// TWK_NATIVE_MATH=1 — the OPT-IN approximate build (libtweeker_hip_fast.so, csrc/Makefile): the shading kernels take gfx950's
// native v_sin_f32 / v_cos_f32 / v_exp_f32 here, and their translation unit is compiled with approximate division and square
// root (v_rcp_f32, v_sqrt_f32), flushed denormals and fused multiply-adds. It is what the reference itself ran: its device code
// was built with --use_fast_math (apps/rtigo3/CMakeLists.txt:165-184). Images are then no longer bit-identical to the oracle;
// tests/test_gpu_native_math.py holds them to SURVEY 8(d)'s tolerance (relative RMSE <= 2 %). The default build is exact.
#ifndef TWK_NATIVE_MATH
#define TWK_NATIVE_MATH 0
#endif
#if TWK_NATIVE_MATH && defined(__HIP_DEVICE_COMPILE__)
#define TWK_NATIVE_DEVICE 1
#else
#define TWK_NATIVE_DEVICE 0
#endif

namespace twk {

static const float kPi      = 3.14159265358979323846f;  // M_PIf   (vector_math.h)
static const float kInvPi   = 0.318309886183790671538f; // M_1_PIf (vector_math.h)

TWK_HD float    asFloat(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
TWK_HD uint32_t asUint(float f)     { union { uint32_t u; float f; } c; c.f = f; return c.u; }

// --- sin / cos -------------------------------------------------------------------------------
TWK_HD void octantReduce(float ax, float& r, int& j)
{
  j = (int) (ax * 1.27323954473516f);
  j = (j + 1) & ~1;
  const float y = (float) j;
  r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}

TWK_HD float sinKernel(float r)
{
  const float z = r * r;
  return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
}

TWK_HD float cosKernel(float r)
{
  const float z = r * r;
  return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}

TWK_HD float sinP(float x)
{
#if TWK_NATIVE_DEVICE
  return __builtin_amdgcn_sinf(x * 0.15915494309189535f); // v_sin_f32 takes revolutions; arguments here lie within a few turns
#endif
  float sign = 1.0f;
  float ax = x;
  if (x < 0.0f) { sign = -1.0f; ax = -x; }
  float r; int j;
  octantReduce(ax, r, j);
  j &= 7;
  if (j > 3) { sign = -sign; j -= 4; }
  const float y = (j == 2) ? cosKernel(r) : sinKernel(r);
  return sign * y;
}

TWK_HD float cosP(float x)
{
#if TWK_NATIVE_DEVICE
  return __builtin_amdgcn_cosf(x * 0.15915494309189535f);
#endif
  float sign = 1.0f;
  const float ax = (x < 0.0f) ? -x : x;
  float r; int j;
  octantReduce(ax, r, j);
  j &= 7;
  if (j > 3) { sign = -sign; j -= 4; }
  if (j > 1) { sign = -sign; }
  const float y = (j == 2) ? sinKernel(r) : cosKernel(r);
  return sign * y;
}

// --- exp -------------------------------------------------------------------------------------
TWK_HD float expP(float x)
{
#if TWK_NATIVE_DEVICE
  return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); // v_exp_f32
#endif
  if (x > 88.0f)  return asFloat(0x7f800000u);
  if (x < -87.0f) return 0.0f;
  float z = floorf(1.44269504088896341f * x + 0.5f);
  const int n = (int) z;
  x = x - z * 0.693359375f;
  x = x - z * -2.12194440e-4f;
  z = x * x;
  float p = ((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f;
  p = p * z + x + 1.0f;
  return p * asFloat((uint32_t) (n + 127) << 23);
}

// --- log / pow (tonemapper only: Application.cpp:2283-2287 calls powf) -----------------------
// Cephes logf: x = m 2^e with m in [sqrt(1/2), sqrt(2)), degree-9 polynomial in m - 1. Arguments below the normal
// range count as zero (-inf), negative ones give NaN.
TWK_HD float logP(float x)
{
  if (x < 0.0f) return asFloat(0x7fc00000u);
  if (x < 1.17549435e-38f) return asFloat(0xff800000u);
  if (x > 3.40282347e+38f) return x; // +inf
  const uint32_t bits = asUint(x);
  int e = (int) (bits >> 23) - 126;                         // frexp: x = m 2^e, m in [0.5, 1)
  float m = asFloat((bits & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
  else                           { m = m - 1.0f; }
  const float z = m * m;
  float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
              + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
              + 3.3333331174e-1f) * m * z;
  const float fe = (float) e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  float r = m + y;
  r = r + 0.693359375f * fe;
  return r;
}

// pow for the tonemapper's non-negative bases: exp(y log x); 0^y = 0 for y > 0, x^0 = 1.
TWK_HD float powP(float x, float y)
{
  if (y == 0.0f) return 1.0f;
  if (y == 1.0f) return x; // exact, as libm: the neutral tonemapper (gamma 1, crushBlacks 0) is the identity
  if (!(x > 0.0f)) return (x == 0.0f && y > 0.0f) ? 0.0f : ((x == 0.0f) ? asFloat(0x7f800000u) : asFloat(0x7fc00000u));
  return expP(y * logP(x));
}

// --- atan / atan2 ----------------------------------------------------------------------------
TWK_HD float atanP(float xx)
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

TWK_HD float atan2P(float y, float x)
{
  const float PIO2F = 1.5707963267948966192f;
  if (x == 0.0f)
  {
    if (y > 0.0f) return PIO2F;
    if (y < 0.0f) return -PIO2F;
    return 0.0f;
  }
  if (y == 0.0f)
  {
    return (x < 0.0f) ? kPi : 0.0f;
  }
  float w = 0.0f;
  if (x < 0.0f) w = (y < 0.0f) ? -kPi : kPi;
  return w + atanP(y / x);
}

// --- asin / acos -----------------------------------------------------------------------------
TWK_HD float asinP(float xx)
{
  float sign = 1.0f;
  float a = xx;
  if (xx < 0.0f) { sign = -1.0f; a = -xx; }
  if (a > 1.0f) return asFloat(0x7fc00000u);
  float x, z;
  int flag;
  if (a > 0.5f) { z = 0.5f * (1.0f - a); x = sqrtf(z); flag = 1; }
  else          { x = a; z = x * x; flag = 0; }
  z = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
  if (flag != 0) { z = z + z; z = 1.5707963267948966192f - z; }
  return sign * z;
}

TWK_HD float acosP(float x)
{
  if (x < -1.0f || x > 1.0f) return asFloat(0x7fc00000u);
  if (x < -0.5f) return kPi - 2.0f * asinP(sqrtf(0.5f * (1.0f + x)));
  if (x > 0.5f)  return 2.0f * asinP(sqrtf(0.5f * (1.0f - x)));
  return 1.5707963267948966192f - asinP(x);
}

// --- float3 with the reference's evaluation order (shaders/vector_math.h) ---------------------
struct V3 { float x, y, z; };

TWK_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
TWK_HD V3 v3(float s) { V3 r; r.x = s; r.y = s; r.z = s; return r; }
TWK_HD V3 v3(const float4& f) { V3 r; r.x = f.x; r.y = f.y; r.z = f.z; return r; }
TWK_HD V3 operator-(const V3& a) { return v3(-a.x, -a.y, -a.z); }
TWK_HD V3 operator+(const V3& a, const V3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
TWK_HD V3 operator-(const V3& a, const V3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
TWK_HD V3 operator*(const V3& a, const V3& b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
TWK_HD V3 operator*(const V3& a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
TWK_HD V3 operator*(float s, const V3& a) { return v3(a.x * s, a.y * s, a.z * s); }
TWK_HD V3 operator/(const V3& a, float s) { const float inv = 1.0f / s; return v3(a.x * inv, a.y * inv, a.z * inv); } // :520-530
TWK_HD float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                                // :574-577
TWK_HD V3 cross(const V3& a, const V3& b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); } // :580-583
TWK_HD float length(const V3& v) { return sqrtf(dot(v, v)); }                                                            // :586-589
TWK_HD V3 normalize(const V3& v) { const float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }                    // :592-596
TWK_HD V3 reflect(const V3& i, const V3& n) { return i - (2.0f * n) * dot(n, i); }                                       // :605-608
TWK_HD V3 lerp(const V3& a, const V3& b, float t) { return a + t * (b - a); }                                            // :547-550
TWK_HD float maxComponent(const V3& a) { return fmaxf(fmaxf(a.x, a.y), a.z); }                                           // :442-445
TWK_HD V3 exp3(const V3& v) { return v3(expP(v.x), expP(v.y), expP(v.z)); }                                              // :620-623
TWK_HD bool isNull(const V3& v) { return v.x == 0.0f && v.y == 0.0f && v.z == 0.0f; }      // shader_common.h:172-175
TWK_HD bool isNotNull(const V3& v) { return v.x != 0.0f || v.y != 0.0f || v.z != 0.0f; }   // shader_common.h:177-180
TWK_HD float intensity(const V3& c) { return (c.x + c.y + c.z) * 0.3333333333f; }          // shader_common.h:157-160
TWK_HD float powerHeuristic(float a, float b) { const float t = a * a; return t / (t + b * b); } // shader_common.h:183-187

// Row-major 3x4 affine matrix helpers (closesthit.cu:88-123)
TWK_HD V3 transformPoint(const float* m, const V3& v)
{
  return v3(m[0] * v.x + m[1] * v.y + m[2]  * v.z + m[3],
            m[4] * v.x + m[5] * v.y + m[6]  * v.z + m[7],
            m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]);
}
TWK_HD V3 transformVector(const float* m, const V3& v)
{
  return v3(m[0] * v.x + m[1] * v.y + m[2]  * v.z,
            m[4] * v.x + m[5] * v.y + m[6]  * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// inverse matrix applied as inverse transpose
TWK_HD V3 transformNormal(const float* m, const V3& v)
{
  return v3(m[0] * v.x + m[4] * v.y + m[8]  * v.z,
            m[1] * v.x + m[5] * v.y + m[9]  * v.z,
            m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

} // namespace twk
