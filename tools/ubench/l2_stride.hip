// Does the ROW PITCH of a GEMM operand decide what a CU takes in from L2?  An LDS-DMA piece of the implicit GEMM is 8 rows x 128 B: lane = (row, 16-B chunk), the rows
// one operand pitch apart (pitch = K x 2 bytes: 2560 B for the C = 1280 layers).  If the XCD's L2 channels interleave on low address bits, a pitch that is a multiple of
// 2^k lines folds all rows of all pieces of a K step onto a few channels.
//   hipcc --offload-arch=gfx950 -O3 l2_stride.hip -o l2_stride ; ./l2_stride
// Every workgroup (one per CU, 4 loader waves) walks K steps: per step 28 pieces = 224 rows x 128 B at the step's column, rows `pitch` bytes apart (the 64 x 160 tile's
// stage), 3 steps in flight; workgroups read one of 8 row panels of a buffer that fits the XCD's L2.  Prints GB/s per CU per pitch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void bufdma16(const void* base, void* lds_wave_base, unsigned voff, unsigned soff) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFF0u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// mfma_waves > 0: waves 4 .. 7 run bare MFMA loops on random register operands beside the four loader waves (the consumers of igemm_pc.h without their LDS reads);
// clk[0..1]: shader clock ticks and 100 MHz real-time ticks of workgroup 0's loader wave 0 over the timed loop -> the clock the chip holds
__global__ __launch_bounds__(512) void k(const char* __restrict__ buf, unsigned* __restrict__ sink, int pitch, int ksteps, int iters, int panels, int mfma_per_step,
                                         unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (wid >= 4) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(((lane * 37 + e * 11 + wid) % 255) / 127.0f - 1.0f); b[e] = (__bf16)(((lane * 29 + e * 7) % 251) / 125.0f - 1.0f); }
    f32x4 acc[10];
    for (int j = 0; j < 10; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters * ksteps; ++it) {
      for (int m = 0; m < mfma_per_step; m += 5) {          // (static accumulator indices: a runtime index would put the array in scratch)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[5 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[5 + j], 0, 0, 0);
        m += 5;
      }
      asm volatile("s_barrier" ::: "memory");
    }
    float t = 0.f;
    for (int j = 0; j < 10; ++j) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (t == 1.2345f) sink[1] = 1;
    return;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int lrow = lane >> 3;
  // panels > 0: workgroup b reads panel b % panels (with round-robin placement over the 8 XCDs and panels = 8: ONE panel per XCD); panels < 0: every XCD cycles through
  // -panels panels of its own ((b / 8) % -panels): the XCD's L2 working set is -panels x 224 rows x pitch
  const unsigned panel = (panels > 0 ? (blockIdx.x % panels) : ((blockIdx.x & 7) * (unsigned)(-panels) + (blockIdx.x >> 3) % (unsigned)(-panels))) * 224u * (unsigned)pitch;
  unsigned voff[7];
  for (int j = 0; j < 7; ++j) voff[j] = panel + (unsigned)((wid + 4 * j) * 8 + lrow) * (unsigned)pitch + (unsigned)((lane & 7) ^ lrow) * 16u;
  for (int it = 0; it < iters; ++it) {
    for (int ks = 0; ks < ksteps; ++ks) {
      char* dst = smem + (ks % 4) * 28672;
      const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)ks * 128u);
#pragma unroll
      for (int j = 0; j < 7; ++j) bufdma16(buf, dst + (wid + 4 * j) * 1024, voff[j], so);
      asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      if (mfma_per_step >= 0) asm volatile("s_barrier" ::: "memory");      // one barrier per K step with the MFMA waves, as in the producer / consumer kernel
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
  if (*(const unsigned*)(smem + threadIdx.x * 16) == 0x12345u) sink[0] = 1;
}
int main() {
  char* buf; unsigned* sink; unsigned long long* clk;
  hipMalloc(&clk, 64);
  const size_t cap = (size_t)64 << 20;
  hipMalloc(&buf, cap); hipMalloc(&sink, 256); hipMemset(buf, 1, cap);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 28672);
  const int pitches[] = {640, 1280, 1408, 2560, 2688, 5120, 12800, 23040, 23168};
  for (int pitch : pitches) {
    int ksteps = pitch / 128; if (ksteps > 40) ksteps = 40;          // columns walked: the row's K range (capped)
    for (int panels : {1, 8}) {
      if ((size_t)panels * 224 * pitch > cap) continue;
      const int iters = 400 / (ksteps / 5 + 1) + 4;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 4 * 28672, 0, buf, sink, pitch, ksteps, 2, panels, -1, clk);
      hipEventRecord(a); hipLaunchKernelGGL(k, dim3(256), dim3(256), 4 * 28672, 0, buf, sink, pitch, ksteps, iters, panels, -1, clk); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double bytes_cu = (double)iters * ksteps * 28672.0;
      printf("pitch %6d B (%3d lines, %s) %d panel(s) x %4d KB: %7.1f GB/s per CU  %5.1f B/clk/CU   %.2f us per 20 K steps\n", pitch, pitch / 128, (pitch / 128) % 2 ? "odd " : "even", panels,
             224 * pitch / 1024, bytes_cu / (ms * 1e-3) * 1e-9, bytes_cu / (ms * 1e-3) / 2.4e9, ms * 1e3 / (iters * ksteps) * 20);
    }
  }
  // L2 capacity: the XCD's working set (per-XCD panels of 560 KB at pitch 2560), loaders only and with 20 MFMAs per wave and K step beside them
  printf("\nper-XCD working set (pitch 2560): panels per XCD -> GB/s per CU loaders only | with 20 MFMAs per wave and step\n");
  for (int pp : {1, 2, 4, 5, 6, 7, 8, 10}) {
    double r[2];
    for (int v = 0; v < 2; ++v) {
      const int pitch = 2560, ksteps = 20, iters = 300;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k, dim3(256), dim3(v ? 512 : 256), 4 * 28672, 0, buf, sink, pitch, ksteps, 20, -pp, v ? 20 : -1, clk);
      hipEventRecord(a); hipLaunchKernelGGL(k, dim3(256), dim3(v ? 512 : 256), 4 * 28672, 0, buf, sink, pitch, ksteps, iters, -pp, v ? 20 : -1, clk); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      r[v] = 28672.0 / (ms * 1e3 / (iters * ksteps)) * 1e-3;
    }
    printf("  %2d panels per XCD (%4.1f MB): %6.1f | %6.1f GB/s per CU   (%.3f | %.3f us per K step)\n", pp, pp * 224 * 2560 / 1048576.0, r[0], r[1], 28.672 / r[0], 28.672 / r[1]);
  }
  // the same loader stream (pitch 2560: K = 1280) with four MFMA waves beside it, 0 .. 40 MFMAs (16x16x32 bf16, random operands) per wave and K step, one barrier per step:
  // what does the matrix pipe's load do to the piece rate, and what clock does the chip hold?  (20 MFMAs per wave and step = the 64 x 160 tile)
  printf("\nloaders + MFMA waves, pitch 2560, 8 panels: MFMAs per wave and K step -> us per K step, GB/s per CU, shader clock (s_memtime / s_memrealtime)\n");
  for (int mf : {0, 10, 20, 30, 40, 60}) {
    const int pitch = 2560, ksteps = 20, iters = 400, panels = 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 4 * 28672, 0, buf, sink, pitch, ksteps, iters, panels, mf, clk);      // warm: let the clock settle
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(256), dim3(512), 4 * 28672, 0, buf, sink, pitch, ksteps, iters, panels, mf, clk); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double step_us = ms * 1e3 / (iters * ksteps), ghz = (double)h[0] / ((double)h[1] * 10.0) ;      // ticks per 10 ns
    printf("  %2d MFMAs: %.3f us per K step  %6.1f GB/s per CU  clock %.2f GHz  -> %4.0f cycles per K step (MFMA pipe %3d cycles, 28 pieces)\n", mf, step_us, 28672.0 / step_us * 1e-3, ghz,
           step_us * ghz * 1e3, mf * 16);
  }
  return 0;
}
