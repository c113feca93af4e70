// Stand-alone bf16 GEMM on the 256 x 256 x 64, 8-wave, 8-phase schedule (cdna_hip_programming.md "The 256^2 8-phase
// template"), written from that description: the measuring stick for igemm8p (agenda_amd/csrc/igemm8p.h), which is this
// main loop with the im2col gather on the A side and the register epilogue behind it.
//
//   C[M][N] = A[M][K] . B[N][K]^T      (A = pixels x channels, B = weight rows; both K-contiguous bf16; C bf16)
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm8p gemm8p.hip      Run: ./gemm8p [M N K iters]   (./gemm8p q = quick set)
// Measured on this pool (random uniform [-1, 1) operands): 1203 - 1232 TFLOP/s at 4096^3 / 8192^3.  Schedule variants, one box, same run
// (4096^3 / 8192^3): base 1227 / 1231; -DV_ORDER=1 (MFMAs k-innermost) 1183 / 1234; -DV_ORDER=2 (row tiles innermost) 1196 / 1223;
// -DV_STAGE_FIRST (LDS-DMA ahead of the fragment reads) 1163 / 1223; -DV_NOPRIO (no s_setprio) 1229 / 1235: all within +-3 %.
//
// Geometry: 512 threads = 8 waves as 2 (pixel rows) x 4 (channels); wave (wr, wc) owns pixels wr*128 .. +127 and channels
// wc*64 .. +63 of the tile.  One K tile (64 deep) is four 16 KiB half-tiles in LDS: A half mh = the 64-pixel sub-blocks
// {wr*128 + mh*64 ..} of both wave rows, B half nh = the 16-row MFMA tiles j = 2nh, 2nh + 1 of all four wave columns.  Two K
// tiles are resident (128 KiB).  A phase = {fragment reads of one half-tile | stage one half-tile of a later K tile by LDS-DMA}
// -> barrier -> 16 MFMAs (one quadrant of the wave's accumulators over the whole K tile) -> barrier; the two wave rows run the
// same program one barrier apart, so on every SIMD one wave computes while its partner reads and stages.  vmcnt is waited
// for once per K tile (phase 4), counted: three half-tiles stay in flight across it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstring>
#include <type_traits>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define DEV __device__ __forceinline__

DEV void bufdma16(const void* base, void* lds_wave_base, unsigned voff, unsigned soff, unsigned nrec) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nrec, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
DEV int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
DEV unsigned pack_bf2(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}

#define HALF 16384
#define KT_BYTES 65536            // one K tile: A0 A1 B0 B1
template <int N> DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
template <int N> DEV void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory"); }
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                        int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int tiles_n = N >> 8;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n, tm = bid / tiles_n;
  const int m0 = tm << 8, n0 = tn << 8;
  const int nk = K >> 6;

  // ---- LDS-DMA sources.  A piece = one wave instruction = 8 LDS rows of 128 B; this wave issues pieces wid and wid + 8 of every
  // half-tile.  Lane: row lrow of the piece, position lane & 7 holds logical 16-B chunk (lane & 7) ^ (row & 7).
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  unsigned avoff[2][2], bvoff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int R = (wid + 8 * i) * 8 + lrow;                        // LDS row of the half-tile
      const int pix = (R >> 6) * 128 + h * 64 + (R & 63);             // A half h: wave row R >> 6, its pixels h*64 ..
      avoff[h][i] = (unsigned)(((long long)(m0 + pix) * K + lchunk * 8) * 2);
      const int wco = R >> 5, jj = (R >> 4) & 1, rho = R & 15;        // B half h: wave column, MFMA tile 2h + jj, fragment row rho = 4q + r
      const int col = wco * 64 + (rho >> 2) * 16 + 4 * (2 * h + jj) + (rho & 3);
      bvoff[h][i] = (unsigned)(((long long)(n0 + col) * K + lchunk * 8) * 2);
    }
  constexpr unsigned LIVE = 0x7FFFFFF0u;
  auto stageA = [&](int buf, int h, int t) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)t * 128u), nr = t < nk ? LIVE : 0u;
    char* d = smem + buf * KT_BYTES + h * HALF + wid * 1024;
    bufdma16(A, d, avoff[h][0], so, nr);
    bufdma16(A, d + 8192, avoff[h][1], so, nr);
  };
  auto stageB = [&](int buf, int h, int t) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)t * 128u), nr = t < nk ? LIVE : 0u;
    char* d = smem + buf * KT_BYTES + 2 * HALF + h * HALF + wid * 1024;
    bufdma16(B, d, bvoff[h][0], so, nr);
    bufdma16(B, d + 8192, bvoff[h][1], so, nr);
  };

  // ---- fragment read addresses (16x16x32 operand: lane (q, rho) reads row rho, k chunk kk*4 + q; swizzle key = row & 7 = rho & 7)
  const int q = lane >> 4, rho = lane & 15;
  const int sw0 = ((q ^ (rho & 7)) << 4);
  const int xoff = (wr * 64 + rho) * 128 + sw0;                       // + ii*2048 ; kk = 1: ^ 64
  const int woff = 2 * HALF + (wc * 32 + rho) * 128 + sw0;            // + jj*2048

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[4][2], wf[2][2][2];

  auto readW = [&](int buf, int h) {
    const char* s = smem + buf * KT_BYTES + h * HALF;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      wf[h][jj][0] = *(const bf16x8*)(s + jj * 2048 + woff);
      wf[h][jj][1] = *(const bf16x8*)(s + jj * 2048 + (woff ^ 64));
    }
  };
  auto readX = [&](int buf, int h) {
    const char* s = smem + buf * KT_BYTES + h * HALF;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      xf[ii][0] = *(const bf16x8*)(s + ii * 2048 + xoff);
      xf[ii][1] = *(const bf16x8*)(s + ii * 2048 + (xoff ^ 64));
    }
  };
  auto mma = [&](auto mh_, auto nh_) {
    constexpr int mh = decltype(mh_)::value, nh = decltype(nh_)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
#if defined(V_ORDER) && V_ORDER == 1
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#elif defined(V_ORDER) && V_ORDER == 2
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#else
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#endif
          acc[mh * 4 + ii][nh * 2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nh][jj][kk], xf[ii][kk], acc[mh * 4 + ii][nh * 2 + jj], 0, 0, 0);
#ifndef V_NOPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  // one K tile (tile t, resident in buffer b): four phases
  auto ktile = [&](auto b_, int t) {
    constexpr int b = decltype(b_)::value, o = b ^ 1;
    // P1: W half 0 + X half 0 of this tile; stage A half 1 of tile t + 1
#ifdef V_STAGE_FIRST
    stageA(o, 1, t + 1); __builtin_amdgcn_sched_barrier(0);
    readW(b, 0); __builtin_amdgcn_sched_barrier(0); readX(b, 0);
#else
    readW(b, 0); __builtin_amdgcn_sched_barrier(0); readX(b, 0);
    stageA(o, 1, t + 1);
#endif
    wait_lgkm<8>();                                  // the four W reads (issued first) have returned: B[b][0] may be restaged next phase
    BAR(); wait_lgkm<0>(); __builtin_amdgcn_sched_barrier(0);
    mma(I0{}, I0{});
    BAR();
    // P2: W half 1; stage B half 0 of tile t + 2
#ifdef V_STAGE_FIRST
    stageB(b, 0, t + 2); __builtin_amdgcn_sched_barrier(0);
    readW(b, 1);
#else
    readW(b, 1);
    stageB(b, 0, t + 2);
#endif
    BAR(); wait_lgkm<0>(); __builtin_amdgcn_sched_barrier(0);
    mma(I0{}, I1{});
    BAR();
    // P3: X half 1; stage A half 0 of tile t + 2
#ifdef V_STAGE_FIRST
    stageA(b, 0, t + 2); __builtin_amdgcn_sched_barrier(0);
    readX(b, 1);
#else
    readX(b, 1);
    stageA(b, 0, t + 2);
#endif
    BAR(); wait_lgkm<0>(); __builtin_amdgcn_sched_barrier(0);
    mma(I1{}, I0{});
    BAR();
    // P4: stage B half 1 of tile t + 2; everything older than the last three half-tiles has landed -> tile t + 1 is complete
    stageB(b, 1, t + 2);
    wait_vm<6>();
    BAR();
    mma(I1{}, I1{});
    BAR();
  };

  // prologue: tile 0 (four half-tiles), then B0, A0, B1 of tile 1
  stageB(0, 0, 0); stageA(0, 0, 0); stageB(0, 1, 0); stageA(0, 1, 0);
  stageB(1, 0, 1); stageA(1, 0, 1); stageB(1, 1, 1);
  wait_vm<6>();
  BAR();
  if (wr == 1) BAR();                                // the second wave row runs one barrier behind the first
  int t = 0;
  for (; t + 1 < nk; t += 2) { ktile(I0{}, t); ktile(I1{}, t + 1); }
  if (t < nk) ktile(I0{}, t);
  if (wr == 0) BAR();
  wait_vm<0>();                                      // dead tail pieces still write zeros into LDS

  // epilogue: lane (q, px) holds channels wc*64 + q*16 .. +15 of pixel wr*128 + i*16 + px
  const int px = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + px;
    bf16_t* op = C + (long long)m * N + n0 + wc * 64 + q * 16;
    u32x4 p0, p1;
    p0[0] = pack_bf2(acc[i][0][0], acc[i][0][1]); p0[1] = pack_bf2(acc[i][0][2], acc[i][0][3]);
    p0[2] = pack_bf2(acc[i][1][0], acc[i][1][1]); p0[3] = pack_bf2(acc[i][1][2], acc[i][1][3]);
    p1[0] = pack_bf2(acc[i][2][0], acc[i][2][1]); p1[1] = pack_bf2(acc[i][2][2], acc[i][2][3]);
    p1[2] = pack_bf2(acc[i][3][0], acc[i][3][1]); p1[3] = pack_bf2(acc[i][3][2], acc[i][3][3]);
    *(u32x4*)op = p0; *(u32x4*)(op + 8) = p1;
  }
}

// ------------------------------------------------------------------------------------------------------------------
static inline float bf2f(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static inline float urand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (float)((rng_s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }

static int run(int M, int N, int K, int iters, bool full_check) {
  if (M % 256 || N % 256 || K % 64 || K < 128) { fprintf(stderr, "bad shape\n"); return 2; }
  std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
  for (auto& v : hA) v = f2bf(urand());
  for (auto& v : hB) v = f2bf(urand());
  bf16_t *dA, *dB, *dC;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, hC.size() * 2));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xFF, hC.size() * 2));
  const int lds = 2 * KT_BYTES;
  CK(hipFuncSetAttribute((const void*)gemm8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = (M / 256) * (N / 256);
  hipLaunchKernelGGL(gemm8p_kernel, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
  // reference: every element (small shapes) or 4096 sampled elements, fp64
  double maxerr = 0; long long bad = 0, checked = 0;
  auto check = [&](int m, int n) {
    double s = 0; for (int k = 0; k < K; ++k) s += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
    const double got = bf2f(hC[(size_t)m * N + n]), err = fabs(got - s), tol = 0.01 * fabs(s) + 0.02 * sqrt((double)K) * 0.05;
    if (err > maxerr) maxerr = err;
    if (!(err <= tol)) { if (bad < 5) fprintf(stderr, "  mismatch (%d,%d): got %g want %g\n", m, n, got, s); ++bad; }
    ++checked;
  };
  if (full_check) { for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) check(m, n); }
  else for (int i = 0; i < 4096; ++i) check((int)((urand() * 0.5f + 0.5f) * (M - 1)), (int)((urand() * 0.5f + 0.5f) * (N - 1)));
  printf("M=%d N=%d K=%d: checked %lld elements, %lld bad, max abs err %.4g\n", M, N, K, checked, bad, maxerr);
  if (iters > 0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm8p_kernel, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm8p_kernel, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("M=%d N=%d K=%d: %.1f us per launch, %.0f TFLOP/s (random uniform [-1,1) operands)\n", M, N, K, us, tf);
  }
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc >= 4) return run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 20, false);
  int rc = 0;
  if (argc == 2) {                         // quick: one correctness shape + the two square timings
    rc |= run(512, 768, 192, 0, true); rc |= run(4096, 4096, 4096, 50, false); rc |= run(8192, 8192, 8192, 10, false); rc |= run(32768, 2560, 320, 50, false);
    printf(rc ? "FAILED\n" : "ALL OK\n"); return rc;
  }
  rc |= run(256, 256, 128, 0, true);       // nk = 2
  rc |= run(512, 768, 192, 0, true);       // odd nk, N tiles not a multiple of 8
  rc |= run(256, 512, 832, 0, true);       // nk = 13
  rc |= run(4096, 4096, 4096, 50, false);
  rc |= run(8192, 8192, 8192, 10, false);
  rc |= run(32768, 512, 2880, 50, false);  // SD-1.5 L0 3x3-conv-like K
  rc |= run(32768, 2560, 320, 50, false);  // GEGLU-like short K
  printf(rc ? "FAILED\n" : "ALL OK\n");
  return rc;
}
