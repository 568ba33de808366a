// Does the hardware fp32 atomic add (global_atomic_add_f32, no return) of gfx950 keep subnormals and round to nearest
// even, i.e. is `atomic x += c` bit-identical to the VALU `x = x + c`? Build: hipcc -O3 --offload-arch=gfx950 -o atomic_denorm_probe atomic_denorm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
__global__ void addKernel(float* x, const float* c, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) unsafeAtomicAdd(&x[i], c[i]); }
__global__ void refKernel(float* x, const float* c, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] + c[i]; }
int main()
{
  const int n = 1 << 20;
  std::vector<float> x(n), c(n), a(n), r(n);
  std::mt19937 rng(1234);
  for (int i = 0; i < n; ++i)
  {
    unsigned int bx = rng(), bc = rng();
    const int kind = i & 7;
    if (kind == 0) { bx &= 0x807fffffu; bc &= 0x807fffffu; }              // both subnormal
    else if (kind == 1) { bx = 0u; bc &= 0x807fffffu; }                     // 0 + subnormal
    else if (kind == 2) { bx = (bx & 0x807fffffu) | 0x00800000u; bc &= 0x807fffffu; } // smallest normals + subnormal
    else if (kind == 3) { bx = (bx & 0x007fffffu) | 0x3f800000u; bc = (bc & 0x007fffffu) | 0x33000000u; } // rounding ties region
    else { bx = (bx & 0x807fffffu) | ((60u + (bx >> 23) % 130u) << 23); bc = (bc & 0x807fffffu) | ((60u + (bc >> 23) % 130u) << 23); }
    memcpy(&x[i], &bx, 4); memcpy(&c[i], &bc, 4);
  }
  float *dx, *dc, *dr;
  hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dr, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dr, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
  addKernel<<<n / 256, 256>>>(dx, dc, n); refKernel<<<n / 256, 256>>>(dr, dc, n);
  hipMemcpy(a.data(), dx, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
  int badVsValu = 0, badVsHost = 0;
  for (int i = 0; i < n; ++i)
  {
    const float h = x[i] + c[i];
    if (memcmp(&a[i], &r[i], 4)) { if (badVsValu < 5) printf("atomic != valu at %d: x %a c %a atomic %a valu %a\n", i, x[i], c[i], a[i], r[i]); ++badVsValu; }
    if (memcmp(&a[i], &h, 4)) ++badVsHost;
  }
  printf("n %d  atomic != VALU add: %d  atomic != host add: %d\n", n, badVsValu, badVsHost);
  return 0;
}
