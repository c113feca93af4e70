// What does one CU take in from its XCD's L2 per clock, and does the path matter?  (the one-workgroup-per-CU GEMMs of the
// 16 x 16 / 8 x 8 maps "cost the same with every load dropped" -- dropped = zero-record descriptors, which still occupy the TA -> LDS path.)
//   hipcc --offload-arch=gfx950 -O3 l2_ingest.hip -o l2_ingest ; ./l2_ingest
// Every workgroup (one per CU, 256 CUs) streams an L2-resident 2 MB buffer over and over; modes:
//   0  LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), DEPTH pieces in flight per wave
//   1  coalesced global_load_dwordx4 to registers (1 KiB per wave-instruction), DEPTH loads in flight per wave
//   2  16 rows x 64 B per wave-instruction to registers (an MFMA A fragment read straight from a row-major [M][K] activation, row pitch 2560 B)
//   3  one piece of mode 0 + one load of mode 1 alternating (two paths at once)
// Prints GB/s per CU and bytes per clock per CU at 2.4 GHz for 4 / 8 / 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void bufdma16(const void* base, void* lds_wave_base, unsigned voff, unsigned soff) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFF0u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

constexpr unsigned BUF = 2u << 20;   // bytes; every XCD's L2 (4 MiB) keeps its own copy

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ buf, unsigned* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), nw = blockDim.x >> 6;
  char* lds = smem + wid * (DEPTH * 1024);
  // each wave walks its own stripe of the buffer, workgroups start at different places
  unsigned pos = ((blockIdx.x * 37u + wid * 5u) * 16384u) % BUF;
  u32x4 r[DEPTH];
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) r[d] = u32x4{0, 0, 0, 0};
  const unsigned lane_off = MODE == 2 ? (unsigned)((lane & 15) * 2560 + (lane >> 4) * 16) : (unsigned)lane * 16u;
  const unsigned step = MODE == 2 ? 64u : 1024u;          // mode 2: next 32-deep k block of the same 16 rows
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned so = __builtin_amdgcn_readfirstlane(pos);
      if (MODE == 0 || (MODE == 3 && (d & 1) == 0)) bufdma16(buf, lds + d * 1024, lane_off, so);
      else { acc ^= r[d]; r[d] = *(const u32x4*)(buf + so + lane_off); }
      pos += step * nw; if (pos >= BUF - 65536u) pos -= BUF - 65536u;
    }
    if (MODE == 0) wait_vmcnt<DEPTH / 2>();
  }
  wait_vmcnt<0>();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) acc ^= r[d];
  if (MODE == 0 || MODE == 3) { __syncthreads(); acc ^= *(const u32x4*)(smem + threadIdx.x * 16); }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = acc[0];
}

template <int MODE, int DEPTH> void run(const char* name, const char* buf, unsigned* sink) {
  for (int waves = 4; waves <= 16; waves *= 2) {
    const int threads = waves * 64, blocks = 256, iters = 2000;
    const size_t lds = (size_t)waves * DEPTH * 1024;
    if (lds > 160 * 1024) continue;                          // (16 waves x 16 pieces would need 256 KB of LDS)
    auto kf = k<MODE, DEPTH>;
    hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kf, dim3(blocks), dim3(threads), lds, 0, buf, sink, 50);
    hipEventRecord(a); hipLaunchKernelGGL(kf, dim3(blocks), dim3(threads), lds, 0, buf, sink, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes_cu = (double)iters * DEPTH * 1024.0 * waves;
    printf("%-46s depth %2d  %2d waves/CU: %7.1f GB/s per CU  %5.1f B/clk/CU  (%.1f TB/s chip)\n", name, DEPTH, waves, bytes_cu / (ms * 1e-3) * 1e-9,
           bytes_cu / (ms * 1e-3) / 2.4e9, bytes_cu * 256 / (ms * 1e-3) * 1e-12);
  }
}
int main() {
  char* buf; unsigned* sink;
  hipMalloc(&buf, BUF + (1 << 20)); hipMalloc(&sink, 256);
  hipMemset(buf, 1, BUF + (1 << 20));
  run<0, 4>("LDS-DMA pieces", buf, sink); run<0, 8>("LDS-DMA pieces", buf, sink); run<0, 16>("LDS-DMA pieces", buf, sink);
  run<1, 4>("global_load_dwordx4 -> VGPR (coalesced)", buf, sink); run<1, 8>("global_load_dwordx4 -> VGPR (coalesced)", buf, sink); run<1, 16>("global_load_dwordx4 -> VGPR (coalesced)", buf, sink);
  run<2, 8>("16 rows x 64 B -> VGPR (A fragment, row-major)", buf, sink); run<2, 16>("16 rows x 64 B -> VGPR (A fragment, row-major)", buf, sink);
  run<3, 8>("LDS-DMA + VGPR loads alternating", buf, sink); run<3, 16>("LDS-DMA + VGPR loads alternating", buf, sink);
  return 0;
}
