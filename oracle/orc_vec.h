// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// Vector helpers restating the evaluation order of the reference's shaders/vector_math.h so that every
// float expression rounds the way the reference source is written (left to right, no contraction).
#pragma once
#include "orc_math.h"

namespace orc {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2   { int x, y; };
struct uint2  { unsigned int x, y; };

static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float2 make_float2(float s) { return {s, s}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float3 make_float3(float s) { return {s, s, s}; }
static inline float3 make_float3(const float4& v) { return {v.x, v.y, v.z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline float4 make_float4(const float3& v, float w) { return {v.x, v.y, v.z, w}; }
static inline float4 make_float4(float s) { return {s, s, s, s}; }

// vector_math.h:296-330 (float2), :450-545 (float3)
static inline float2 operator+(const float2& a, const float2& b) { return {a.x + b.x, a.y + b.y}; }
static inline float2 operator-(const float2& a, const float2& b) { return {a.x - b.x, a.y - b.y}; }
static inline float2 operator-(const float2& a, float b) { return {a.x - b, a.y - b}; }
static inline float2 operator*(const float2& a, float s) { return {a.x * s, a.y * s}; }
static inline float2 operator/(const float2& a, const float2& b) { return {a.x / b.x, a.y / b.y}; }
// vector_math.h float2 operator/(float2, float): multiplies by the reciprocal (see :279-288)
static inline float2 operator/(const float2& a, float s) { const float inv = 1.0f / s; return {a.x * inv, a.y * inv}; }
static inline float  dot(const float2& a, const float2& b) { return a.x * b.x + a.y * b.y; }
static inline float  length(const float2& v) { return sqrtf(dot(v, v)); }

static inline float3 operator-(const float3& a) { return {-a.x, -a.y, -a.z}; }
static inline float3 operator+(const float3& a, const float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float3 operator-(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float3 operator*(const float3& a, const float3& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline float3 operator*(const float3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float3 operator*(float s, const float3& a) { return {s * a.x, s * a.y, s * a.z}; }
// vector_math.h:520-545: float3 / float multiplies by the reciprocal
static inline float3 operator/(const float3& a, float s) { const float inv = 1.0f / s; return {a.x * inv, a.y * inv, a.z * inv}; }
static inline float3& operator+=(float3& a, const float3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
static inline float3& operator*=(float3& a, const float3& b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
static inline float3& operator*=(float3& a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
static inline float3& operator/=(float3& a, float s) { const float inv = 1.0f / s; a.x *= inv; a.y *= inv; a.z *= inv; return a; }

// vector_math.h:574-577
static inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vector_math.h:580-583
static inline float3 cross(const float3& a, const float3& b)
{
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// vector_math.h:586-589
static inline float length(const float3& v) { return sqrtf(dot(v, v)); }
// vector_math.h:592-596
static inline float3 normalize(const float3& v) { const float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
// vector_math.h:605-608: i - 2.0f * n * dot(n, i)  == i - ((2.0f * n) * dot(n, i))
static inline float3 reflect(const float3& i, const float3& n) { return i - (2.0f * n) * dot(n, i); }
// vector_math.h:547-550
static inline float3 lerp(const float3& a, const float3& b, float t) { return a + t * (b - a); }
// vector_math.h:442-445
static inline float fmaxf3(const float3& a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
// vector_math.h:620-623 (with the portable expf)
static inline float3 expf3(const float3& v) { return {pm_expf(v.x), pm_expf(v.y), pm_expf(v.z)}; }

static inline float clampf(float v, float lo, float hi) { return fmaxf(lo, fminf(v, hi)); } // vector_math.h:148-151
// the float3 overloads the screenshot tonemapper uses (Application.cpp:2275-2289): vector_math.h:455-458 (float3 + float),
// :526-529 (float3 / float3), :438-441 (fmaxf), :562-565 (clamp), :626-629 (powf, with the portable pm_powf)
static inline float3 operator+(const float3& a, float b) { return {a.x + b, a.y + b, a.z + b}; }
static inline float3 operator/(const float3& a, const float3& b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline float3 fmaxf3v(const float3& a, const float3& b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
static inline float3 clamp3(const float3& v, float a, float b) { return {clampf(v.x, a, b), clampf(v.y, a, b), clampf(v.z, a, b)}; }
static inline float3 powf3(const float3& v, float e) { return {pm_powf(v.x, e), pm_powf(v.y, e), pm_powf(v.z, e)}; }
static inline int clampi(int v, int lo, int hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

} // namespace orc
