// Does the transcendental pipe of gfx950 run beside the plain VALU / the MFMA pipe, or do they share the issue slot?
// hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix ; prints cycles (at 2.4 GHz) per GROUP of instructions per SIMD.
//   mode 0: 8 x v_exp_f32                 mode 1: 8 x v_fma_f32              mode 2: 8 x (v_exp_f32 ; v_fma_f32)
//   mode 3: 8 x (v_exp_f32 ; 2 v_fma_f32) mode 4: 4 x mfma 32x32x16         mode 5: 4 x (mfma ; 2 v_exp_f32)
//   mode 6: 4 x (mfma ; 4 v_exp_f32)      mode 7: 4 x (mfma ; 4 v_fma_f32)   mode 8: 4 x (mfma ; 2 v_exp ; 2 v_fma)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE> __global__ __launch_bounds__(1024) void k(float* out, int iters) {
  float a[8], b[8], c[8];
  for (int i = 0; i < 8; ++i) { a[i] = (float)(threadIdx.x + i) * 1e-3f - 0.5f; b[i] = a[i] * 0.5f; c[i] = a[i] * 0.25f; }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  bf16x8 x, y; for (int e = 0; e < 8; ++e) { x[e] = (short)(threadIdx.x + e); y[e] = (short)(threadIdx.x * 3 + e); }
  for (int it = 0; it < iters; ++it) {
    if (MODE <= 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0 || MODE >= 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (MODE >= 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[i]));
        if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(c[i]));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x), "v"(y));
        if (MODE == 5 || MODE == 6 || MODE == 8) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[2 * j])); asm volatile("v_exp_f32 %0, %0" : "+v"(a[2 * j + 1])); }
        if (MODE == 6) { asm volatile("v_exp_f32 %0, %0" : "+v"(b[2 * j])); asm volatile("v_exp_f32 %0, %0" : "+v"(b[2 * j + 1])); }
        if (MODE == 7) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[2 * j])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[2 * j + 1])); }
        if (MODE == 7 || MODE == 8) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[2 * j])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b[2 * j + 1])); }
      }
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i] + c[i];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  float* d; hipMalloc(&d, 1 << 24);
  const int iters = 10000, threads = 64 * 4 * waves_per_simd, blocks = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100);
  hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const int groups = MODE <= 3 ? 8 : 4;
  const double g = (double)iters * groups * waves_per_simd;      // groups per SIMD
  printf("%-44s %d wave(s)/SIMD: %.1f cycles per group per SIMD\n", name, waves_per_simd, ms * 1e6 / g * 2.4);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 4; w++) {
    if (w == 3) continue;
    run<0>("v_exp_f32", w); run<1>("v_fma_f32", w); run<2>("v_exp_f32 + v_fma_f32", w); run<3>("v_exp_f32 + 2 v_fma_f32", w);
    run<4>("mfma 32x32x16", w); run<5>("mfma + 2 v_exp", w); run<6>("mfma + 4 v_exp", w); run<7>("mfma + 4 v_fma", w); run<8>("mfma + 2 v_exp + 2 v_fma", w);
  }
  return 0;
}
