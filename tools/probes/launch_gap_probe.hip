// How long does the GPU sit between two DEPENDENT kernels of one stream — as plain launches and as a replayed hipGraph?
// ("graph the pass, report inter-kernel gaps before / after".) A chain of 22 kernels like a
// wavefront pass's (11 trace + 10 shade + accumulate), each a streaming read-modify-write of `bytes` bytes so that the
// end-of-kernel cache write-back has something to do; gap = (chain time - sum of kernel times alone) / 21.
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/launch_gap_probe.hip -o build/launch_gap_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void touch(float4* a, size_t n, float k)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
  {
    float4 v = a[i]; v.x += k; v.y += k; v.z += k; v.w += k; a[i] = v;
  }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
  hipStream_t stream; CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  const int chain = 22, repeats = 20;
  for (size_t mb : {0ul, 1ul, 64ul, 512ul})
  {
    const size_t n = mb ? (mb << 20) / sizeof(float4) : 64;
    float4* a; CHECK(hipMalloc(&a, n * sizeof(float4))); CHECK(hipMemset(a, 0, n * sizeof(float4)));
    const int grid = mb ? 256 * 8 : 1;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // one kernel alone
    float alone = 0.0f;
    for (int r = 0; r < repeats + 2; ++r)
    {
      CHECK(hipEventRecord(e0, stream));
      hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, stream, a, n, 1.0f);
      CHECK(hipEventRecord(e1, stream)); CHECK(hipStreamSynchronize(stream));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) alone += ms;
    }
    alone /= repeats;
    // the chain, plain launches
    float plain = 0.0f; double plainWall = 0.0;
    for (int r = 0; r < repeats + 2; ++r)
    {
      const auto t0 = std::chrono::steady_clock::now();
      CHECK(hipEventRecord(e0, stream));
      for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, stream, a, n, 1.0f);
      CHECK(hipEventRecord(e1, stream)); CHECK(hipStreamSynchronize(stream));
      const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) { plain += ms; plainWall += wall; }
    }
    plain /= repeats; plainWall /= repeats;
    // the chain as a graph
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
    for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, stream, a, n, 1.0f);
    CHECK(hipStreamEndCapture(stream, &graph));
    const auto ti = std::chrono::steady_clock::now();
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const double instantiateMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti).count();
    float graphed = 0.0f; double graphWall = 0.0;
    for (int r = 0; r < repeats + 2; ++r)
    {
      const auto t0 = std::chrono::steady_clock::now();
      CHECK(hipEventRecord(e0, stream));
      CHECK(hipGraphLaunch(exec, stream));
      CHECK(hipEventRecord(e1, stream)); CHECK(hipStreamSynchronize(stream));
      const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) { graphed += ms; graphWall += wall; }
    }
    graphed /= repeats; graphWall /= repeats;
    printf("{\"buffer_MB\": %zu, \"kernel_alone_us\": %.1f, \"chain\": %d, \"plain_chain_us\": %.1f, \"plain_gap_us\": %.2f, \"plain_wall_us\": %.1f, "
           "\"graph_chain_us\": %.1f, \"graph_gap_us\": %.2f, \"graph_wall_us\": %.1f, \"graph_instantiate_us\": %.1f}\n",
           mb, alone * 1e3, chain, plain * 1e3, (plain - chain * alone) * 1e3 / (chain - 1), plainWall * 1e3,
           graphed * 1e3, (graphed - chain * alone) * 1e3 / (chain - 1), graphWall * 1e3, instantiateMs * 1e3);
    (void) hipGraphExecDestroy(exec); (void) hipGraphDestroy(graph); (void) hipFree(a);
  }
  return 0;
}
