// Follow-up to gather_probe.hip: what does a wave-instruction of 16-byte lane loads cost when the lanes' addresses
// COINCIDE or fall into a handful of lines (the top of a BVH, which every ray visits) instead of 64 different lines?
// and what does the same fetch cost from LDS? Reports wave-instructions per microsecond per CU and lane-loads/clk/CU.
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe2.hip -o build/gather_probe2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// DISTINCT: number of distinct 128-byte lines the 64 lanes of a wave touch per step (1, 2, 5, 16, 64)
template <int DISTINCT>
__global__ void __launch_bounds__(256) gatherKernel(const float4* __restrict__ table, unsigned int lines, int steps, float* out)
{
  const unsigned int lane = threadIdx.x & 63u;
  unsigned int line = (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) * 2654435761u % lines; // wave-uniform start
  float acc = 0.0f;
  for (int s = 0; s < steps; ++s)
  {
    const unsigned int mine = (line + (lane % DISTINCT) * 977u) % lines;
    const float4* p = table + (size_t) mine * 8;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y + v[k].z;
    line = __builtin_amdgcn_readfirstlane(__float_as_uint(v[0].w)) % lines; // dependent chain, wave-uniform
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int DISTINCT>
__global__ void __launch_bounds__(256) ldsKernel(const float4* __restrict__ table, unsigned int lines, int steps, float* out)
{
  __shared__ float4 cache[32 * 8]; // 32 nodes of 128 B
  for (int i = threadIdx.x; i < 32 * 8; i += 256) cache[i] = table[i];
  __syncthreads();
  const unsigned int lane = threadIdx.x & 63u;
  unsigned int line = threadIdx.x >> 6;
  float acc = 0.0f;
  for (int s = 0; s < steps; ++s)
  {
    const unsigned int mine = (line + (lane % DISTINCT) * 7u) & 31u;
    const float4* p = cache + mine * 8;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y + v[k].z;
    line = (__float_as_uint(v[0].w) + s) & 31u;
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <typename K>
static void run(const char* name, int distinct, K kernel, const float4* table, unsigned int lines, float* out, int cus, double mhz)
{
  const int blocks = cus * 6, steps = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  kernel<<<blocks, 256>>>(table, lines, 50, out);
  hipEventRecord(a);
  kernel<<<blocks, 256>>>(table, lines, steps, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  const double laneLoads = (double) blocks * 256 * steps * 8;
  const double clkPerCu = ms * 1e-3 * mhz * 1e6;
  printf("%-6s distinct lines per wave %2d : %7.3f ms  %8.2f Glane-loads/s  %5.2f lane-loads/clk/CU  clk per wave-instruction per CU %6.1f\n",
         name, distinct, ms, laneLoads / ms * 1e-6, laneLoads / cus / clkPerCu, clkPerCu / ((double) blocks / cus * 4 * steps * 8));
}

int main()
{
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1000.0;
  printf("%s: %d CUs, %.0f MHz\n", prop.name, cus, mhz);
  float* out; hipMalloc(&out, 4);
  const unsigned int lines = (unsigned int) (8.0e6 / 128);
  std::vector<float4> h((size_t) lines * 8);
  unsigned int x = 12345u;
  for (size_t i = 0; i < h.size(); ++i)
  {
    x = x * 1664525u + 1013904223u;
    unsigned int r = (x >> 4) % lines;
    float w; memcpy(&w, &r, 4);
    h[i] = make_float4(0.f, 0.f, 0.f, w);
  }
  float4* table; hipMalloc(&table, h.size() * 16);
  hipMemcpy(table, h.data(), h.size() * 16, hipMemcpyHostToDevice);
  run("global", 1,  gatherKernel<1>,  table, lines, out, cus, mhz);
  run("global", 2,  gatherKernel<2>,  table, lines, out, cus, mhz);
  run("global", 5,  gatherKernel<5>,  table, lines, out, cus, mhz);
  run("global", 16, gatherKernel<16>, table, lines, out, cus, mhz);
  run("global", 64, gatherKernel<64>, table, lines, out, cus, mhz);
  run("lds", 1,  ldsKernel<1>,  table, lines, out, cus, mhz);
  run("lds", 2,  ldsKernel<2>,  table, lines, out, cus, mhz);
  run("lds", 5,  ldsKernel<5>,  table, lines, out, cus, mhz);
  run("lds", 16, ldsKernel<16>, table, lines, out, cus, mhz);
  run("lds", 32, ldsKernel<32>, table, lines, out, cus, mhz);
  hipFree(table);
  return 0;
}
