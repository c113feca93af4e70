// What does the SHAPE of an epilogue's store burst cost?  The 160-wide igemm tiles and the row-panel kernels leave their outputs as 40-byte row chunks per lane (a lane owns 20
// consecutive bf16 channels of one pixel: 16 + 16 + 8-byte stores, four lanes of a pixel = 160 contiguous bytes, 16 pixel rows per wave instruction): every instruction touches
// 32 cache lines partially.  Round 6's stamps of qkv_chain_kernel showed its four 21 MB store bursts at ~4.4 TB/s.  This writes the same [32768][320] bf16 array (21 MB) from
// 256 workgroups x 8 waves, all at once, in four shapes:
//   0  the kernels' shape: wave (mh, nq) of workgroup b: rows 128 b + 64 mh + px + 16 i, bytes 160 nq + 40 q .. + 40 as 16 + 16 + 8
//   1  the same rows and columns, but a wave writes whole 160-byte row segments with consecutive lanes on consecutive 16-byte pieces (what an LDS transpose inside the wave would give)
//   2  whole 640-byte rows: consecutive lanes on consecutive 16-byte pieces of a row, a wave instruction = 1.6 rows (what an LDS transpose across the four column waves would give)
//   3  the 256-wide tiles' shape: lane owns 32 bytes (two 16-byte stores), four lanes of a pixel one 128-byte line, 16 rows per instruction (80 % of the bytes: columns 0 .. 511)
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern ; ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef u32x4 __attribute__((aligned(8))) u32x4_a8;
__global__ __launch_bounds__(512) void k(char* __restrict__ out, int shape, int passes, long long pass_bytes) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, mh = wid >> 2, nq = wid & 3, q = lane >> 4, px = lane & 15;
  const u32x4 v = {(unsigned)lane, (unsigned)wid, blockIdx.x, 7u};
  for (int ps = 0; ps < passes; ++ps) {
    char* o = out + (long long)ps * pass_bytes;
    if (shape == 0) {
      for (int i = 0; i < 4; ++i) {
        char* p = o + (long long)(128 * blockIdx.x + 64 * mh + px + 16 * i) * 640 + 160 * nq + 40 * q;
        *(u32x4_a8*)p = v; *(u32x4_a8*)(p + 16) = v; *(u32x2*)(p + 32) = u32x2{v[0], v[1]};
      }
    } else if (shape == 1) {            // 64 rows x 160 B per wave = 640 pieces of 16 B: piece id = it * 64 + lane, row = id / 10, piece = id % 10
      for (int it = 0; it < 10; ++it) {
        const int id = it * 64 + lane, row = id / 10, pc = id - row * 10;
        *(u32x4*)(o + (long long)(128 * blockIdx.x + 64 * mh + row) * 640 + 160 * nq + 16 * pc) = v;
      }
    } else if (shape == 2) {            // 128 rows x 640 B per workgroup = 5120 pieces, 640 per wave, contiguous
      for (int it = 0; it < 10; ++it) {
        const int id = (wid * 10 + it) * 64 + lane;
        *(u32x4*)(o + (long long)128 * blockIdx.x * 640 + 16LL * id) = v;
      }
    } else {                            // the 256-wide tiles' shape: a lane owns 32 bytes (two 16-byte stores), the four lanes of a pixel one 128-byte line; columns 0 .. 511 of each row only
      for (int i = 0; i < 4; ++i) {
        char* p = o + (long long)(128 * blockIdx.x + 64 * mh + px + 16 * i) * 640 + 128 * nq + 32 * q;
        *(u32x4*)p = v; *(u32x4*)(p + 16) = v;
      }
    }
  }
}
int main() {
  const long long pass_bytes = 32768LL * 640;
  const int passes = 4;
  char* buf; hipMalloc(&buf, pass_bytes * passes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int shape = 0; shape < 4; ++shape) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, buf, shape, passes, pass_bytes);
    hipEventRecord(a, 0);
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, buf, shape, passes, pass_bytes);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters;
    const double bytes = (shape == 3 ? 0.8 : 1.0) * pass_bytes * passes;
    printf("shape %d: %.1f us per launch of %d passes = %.2f TB/s (incl. the launch boundary and its L2 write-back)\n", shape, us, passes, bytes / (us * 1e-6) / 1e12);
  }
  // one pass per launch: the boundary's share
  for (int shape = 0; shape < 3; ++shape) {
    hipEventRecord(a, 0);
    for (int it = 0; it < 40; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, buf, shape, 1, pass_bytes);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("shape %d, one 21 MB pass per launch: %.1f us per launch\n", shape, ms * 1e3 / 40);
  }
  return 0;
}
