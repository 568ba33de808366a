// Issue rate of the vector instructions the traversal kernel is made of, per SIMD, with W waves per SIMD interleaved.
// Each wave runs ITER x 32 independent instances of one instruction; reported: SIMD cycles per wave-instruction
// (= time x clock / (instructions per wave x W)), assuming 2.4 GHz. Build: hipcc -O3 --offload-arch=gfx950 -o valu_issue_probe valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R8(x) x x x x x x x x
#define R32(x) R8(x) R8(x) R8(x) R8(x)
template<int OP> __global__ void __launch_bounds__(256) probe(float* out, int iters, float seed)
{
  float a = seed + threadIdx.x, b = seed * 2.0f, c = seed * 3.0f, d = 1.0f, e = 2.0f, f = 3.0f, g = 4.0f, h = 5.0f;
  unsigned int u = threadIdx.x * 2654435761u;
  for (int i = 0; i < iters; ++i)
  {
    if (OP == 0) asm volatile(R8("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
    if (OP == 1) asm volatile(R8("v_min_f32 %0, %4, %0\n v_max_f32 %1, %5, %1\n v_min_f32 %2, %4, %2\n v_max_f32 %3, %5, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
    if (OP == 2) asm volatile(R8("v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %1, %5, %1, vcc\n v_cndmask_b32 %2, %4, %2, vcc\n v_cndmask_b32 %3, %5, %3, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
    if (OP == 3) asm volatile(R8("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %5\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
    if (OP == 4) asm volatile(R8("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u));
    if (OP == 5) asm volatile(R8("v_max3_f32 %0, %4, %5, %0\n v_min3_f32 %1, %4, %5, %1\n v_max3_f32 %2, %4, %5, %2\n v_min3_f32 %3, %4, %5, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
    if (OP == 6) asm volatile(R8("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n") : "+v"(*(double*) &a), "+v"(*(double*) &c) : "v"(*(double*) &e));
    if (OP == 7) asm volatile(R8("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %5, %1, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %5, %3, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
    if (OP == 8) asm volatile(R8("v_add_u32 %0, %4, %0\n v_lshl_or_b32 %1, %1, 3, %5\n v_and_b32 %2, %4, %2\n v_add_u32 %3, %5, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
    if (OP == 10) asm volatile(R8("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u), "v"(f));
    if (OP == 9) asm volatile(R8("v_fma_f32 %0, %4, %5, %0\n v_min_f32 %1, %0, %1\n v_max_f32 %2, %1, %2\n v_cndmask_b32 %3, %2, %3, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc"); // dependent chain
  }
  if (a + b + c + d == 12345.678f) out[0] = a;
}
template<int OP> void run(const char* name, int wavesPerSimd, int cus, float* dOut)
{
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP><<<cus * wavesPerSimd, 256>>>(dOut, 10, 1.0f);
  hipEventRecord(e0);
  probe<OP><<<cus * wavesPerSimd, 256>>>(dOut, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double instrPerWave = (double) iters * 32.0;
  printf("%-34s W=%d  %.2f SIMD cycles per wave-instruction\n", name, wavesPerSimd, ms * 1e-3 * 2.4e9 / (instrPerWave * wavesPerSimd));
}
int main()
{
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* dOut; hipMalloc(&dOut, 64);
  for (int w : {1, 2, 4, 6})
  {
    run<0>("v_fma_f32", w, cus, dOut); run<1>("v_min/max_f32", w, cus, dOut); run<2>("v_cndmask_b32 (vcc)", w, cus, dOut);
    run<3>("v_cmp_lt_f32 -> vcc", w, cus, dOut); run<4>("v_cvt_f32_ubyteN", w, cus, dOut); run<5>("v_max3/min3_f32", w, cus, dOut);
    run<6>("v_pk_mul_f32", w, cus, dOut); run<7>("v_cmp + v_cndmask pairs", w, cus, dOut); run<8>("int add/lshl_or/and", w, cus, dOut);
    run<9>("dependent fma>min>max>cndmask", w, cus, dOut); run<10>("v_fma_mix_f32 (f16 src0)", w, cus, dOut);
  }
  return 0;
}
