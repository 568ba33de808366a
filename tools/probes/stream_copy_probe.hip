// Which float4 copy kernel reaches the chip's HBM stream rate (MI355X_MICROARCH.md: 6.29 TB/s measured for a float4 copy)?
// twk_stream_peak_gbps is the denominator of two roofline fractions of bench.py; round 2's kernel measured 5.4-5.8 TB/s.
// Variants: pieces per block iteration (loads in flight per lane), grid size, non-temporal loads / stores, buffer size.
// Build: hipcc -O3 --offload-arch=gfx950 -o stream_copy_probe stream_copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template<int UNROLL, bool NT_LOAD, bool NT_STORE>
__global__ void __launch_bounds__(256) copyKernel(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n)
{
  const size_t stride = (size_t) gridDim.x * 256 * UNROLL;
  for (size_t i = (size_t) blockIdx.x * 256 * UNROLL + threadIdx.x; i + (UNROLL - 1) * 256 < n; i += stride)
  {
    v4f v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) v[k] = NT_LOAD ? __builtin_nontemporal_load(src + i + k * 256) : src[i + k * 256];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) { if (NT_STORE) __builtin_nontemporal_store(v[k], dst + i + k * 256); else dst[i + k * 256] = v[k]; }
  }
}
__global__ void fill(v4f* p, size_t n)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
  { unsigned int x = (unsigned int) i * 2654435761u; p[i] = v4f{(float) (x & 0xffff), (float) (x >> 16), 1.5f, (float) i}; }
}
template<int UNROLL, bool NTL, bool NTS>
void run(const char* name, const v4f* a, v4f* b, size_t n, int grid)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  copyKernel<UNROLL, NTL, NTS><<<grid, 256>>>(a, b, n);
  const int reps = 10;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) copyKernel<UNROLL, NTL, NTS><<<grid, 256>>>(a, b, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s unroll %d grid %6d  %7.1f MiB  %8.1f GB/s (read + write)\n", name, UNROLL, grid, n * 16.0 / 1048576.0, 2.0 * n * 16.0 * reps / (ms * 1e-3) / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}
int main()
{
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  for (size_t mib : {256, 1024, 4096})
  {
    const size_t n = mib * 1048576 / 16;
    v4f *a, *b;
    if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    fill<<<4096, 256>>>(a, n); hipDeviceSynchronize();
    for (int perCu : {8, 16, 32, 64})
    {
      const int grid = cus * perCu;
      run<1, false, false>("plain", a, b, n, grid);
      run<2, false, false>("plain", a, b, n, grid);
      run<4, false, false>("plain", a, b, n, grid);
      run<8, false, false>("plain", a, b, n, grid);
      run<4, true,  false>("nt load", a, b, n, grid);
      run<4, false, true >("nt store", a, b, n, grid);
      run<4, true,  true >("nt load + nt store", a, b, n, grid);
      run<8, true,  true >("nt load + nt store", a, b, n, grid);
    }
    // one piece per block, no loop
    run<4, false, false>("one pass grid", a, b, n, (int) (n / 1024));
    run<1, false, false>("one pass grid", a, b, n, (int) (n / 256));
    run<4, true, true>("one pass grid nt", a, b, n, (int) (n / 1024));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.1f MiB  %8.1f GB/s (read + write)\n", "hipMemcpyAsync D2D", n * 16.0 / 1048576.0, 2.0 * n * 16.0 * 10 / (ms * 1e-3) / 1e9);
    hipFree(a); hipFree(b);
  }
  return 0;
}
