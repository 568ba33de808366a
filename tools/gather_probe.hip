// Divergent-gather probe: what does one CU's vector memory path sustain when every lane of a wave reads its own
// 128-byte line (the access pattern of BVH traversal)? Each lane walks a pseudo-random chain of lines inside a
// table that fits the L2/Infinity Cache and reads LOADS x 16 bytes of each line. Reports lane-loads/clk/CU.
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o build/gather_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// MASK: which lanes take part (the others idle, as diverged lanes do in traversal): 0 all, 1 even lanes, 2 lanes 0..31 of the wave,
// 3 one lane in four, 4 lanes 0..15.
template <int LOADS, int MASK>
__global__ void __launch_bounds__(256) gatherKernel(const float4* __restrict__ table, unsigned int lines, int steps, float* out)
{
  const unsigned int lane = threadIdx.x & 63u;
  if (MASK == 1 && (lane & 1u)) return;
  if (MASK == 2 && lane >= 32u) return;
  if (MASK == 3 && (lane & 3u)) return;
  if (MASK == 4 && lane >= 16u) return;
  unsigned int line = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % lines;
  float acc = 0.0f;
  for (int s = 0; s < steps; ++s)
  {
    const float4* p = table + (size_t) line * 8;
    float4 v[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) v[k] = p[k];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) acc += v[k].x + v[k].y + v[k].z;
    line = (__float_as_uint(v[0].w) + threadIdx.x) % lines; // dependent chain, like child references
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int LOADS, int MASK = 0>
static void run(const float4* table, unsigned int lines, float* out, int cus, int wavesPerSimd, double mhz)
{
  const int blocks = cus * wavesPerSimd, steps = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  gatherKernel<LOADS, MASK><<<blocks, 256>>>(table, lines, 50, out);
  hipEventRecord(a);
  gatherKernel<LOADS, MASK><<<blocks, 256>>>(table, lines, steps, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  const double active = (MASK == 0) ? 1.0 : ((MASK == 1 || MASK == 2) ? 0.5 : 0.25);
  const double laneLoads = (double) blocks * 256 * steps * LOADS * active;
  const double clkPerCu = ms * 1e-3 * mhz * 1e6;
  printf("table %6.1f MB  mask %d  waves/SIMD %d  loads/line %d : %7.3f ms  %7.2f Glane-loads/s  %5.2f lane-loads/clk/CU  %6.1f GB/s useful  step latency %6.1f ns\n",
         lines * 128.0 / 1e6, MASK, wavesPerSimd, LOADS, ms, laneLoads / ms * 1e-6, laneLoads / cus / clkPerCu, laneLoads * 16 / ms * 1e-6, ms * 1e6 / steps);
}

int main()
{
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1000.0;
  printf("%s: %d CUs, %.0f MHz\n", prop.name, cus, mhz);
  float* out; hipMalloc(&out, 4);
  for (double mb : {2.0, 8.0, 64.0})
  {
    const unsigned int lines = (unsigned int) (mb * 1e6 / 128);
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
    for (int w : {2, 6, 8})
    {
      run<1>(table, lines, out, cus, w, mhz);
      run<2>(table, lines, out, cus, w, mhz);
      run<4>(table, lines, out, cus, w, mhz);
      run<8>(table, lines, out, cus, w, mhz);
    }
    if (mb == 2.0)
    {
      run<8, 1>(table, lines, out, cus, 6, mhz);
      run<8, 2>(table, lines, out, cus, 6, mhz);
      run<8, 3>(table, lines, out, cus, 6, mhz);
      run<8, 4>(table, lines, out, cus, 6, mhz);
    }
    hipFree(table);
  }
  return 0;
}
