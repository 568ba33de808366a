// ORACLE-SIDE TEST INFRASTRUCTURE — host stand-ins for the handful of device intrinsics the product's kernel headers use,
// so that tweeker_raytracer_amd/csrc/{device_math,device_types,shade_device,trace_device}.h compile with g++ as they are
// (HIP's own headers already make __device__ / __forceinline__ / float4 / make_float4 ... host constructs under g++).
// Only oracle/host_kernels.cpp includes this.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

static inline int          __float_as_int(float f)          { int i; memcpy(&i, &f, 4); return i; }
static inline float        __int_as_float(int i)            { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned int __float_as_uint(float f)         { unsigned int u; memcpy(&u, &f, 4); return u; }
static inline float        __uint_as_float(unsigned int u)  { float f; memcpy(&f, &u, 4); return f; }
// v_rcp_f32 is a 1-ulp approximation that only feeds the conservative box culling (trace_device.h guardedReciprocal):
// the exact reciprocal here can move a borderline slab decision (visit counts by a few in a million), never a hit.
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
// one "lane": wave intrinsics of functions the host build never calls (waveAppend) still have to parse
static inline unsigned long long __ballot(int predicate) { return predicate ? 1ull : 0ull; }
template<typename T> static inline T __shfl(T v, int) { return v; }
static const struct { unsigned int x, y, z; } threadIdx = {0u, 0u, 0u};
static inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { const unsigned int old = *p; *p = old + v; return old; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long old = *p; *p = old + v; return old; }
#ifndef __HIP_MEMORY_SCOPE_SYSTEM
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
#define __hip_atomic_fetch_add(ptr, value, order, scope) ((*(ptr)) += (value))
using std::isnan;
using std::isinf;
using std::min;
using std::max;
