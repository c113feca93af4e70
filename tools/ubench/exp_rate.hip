// v_exp_f32 vs v_exp_f16 (and packed-convert variants) issue rate on gfx950: hipcc --offload-arch=gfx950 -O3 exp_rate.hip -o exp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(float* out, int iters) {
  float a[8]; for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f - 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      else if (MODE == 1) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      else if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
      else if (MODE == 3) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a[i]));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  float* d; hipMalloc(&d, 1 << 24);
  const int iters = 20000, threads = 64 * 4 * waves_per_simd, blocks = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100);
  hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double inst = (double)iters * 8 * waves_per_simd;      // wave-instructions per SIMD
  printf("%-28s %d wave(s)/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 4; w *= 2) { run<0>("v_exp_f32", w); run<1>("v_exp_f16", w); run<3>("v_exp_f16 op_sel hi", w); run<2>("v_fma_f32", w); }
  return 0;
}
